#!/bin/bash
# A variant of the library that differs from the tree's build in a few files only:
#   scratch/build_variant2.sh <name> "<files without .hip>" <flags...>   ->  honeybadgermpc_amd/lib/libhbmpc_hip_<name>.so
# (the named files are compiled with the extra flags, every other object is the main build's: run csrc/build.sh first)
set -e
NAME="$1"; FILES="$2"; shift 2
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/honeybadgermpc_amd/csrc"
OBJ="$SRC/.obj_$NAME"
mkdir -p "$OBJ"
objs=""
pids=""
for src in "$SRC"/*.hip; do
  b="$(basename "${src%.hip}")"
  if [[ " $FILES " == *" $b "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=default -Wno-unused-result "$@" -c "$src" -o "$OBJ/$b.o" &
    pids="$pids $!"
    objs="$objs $OBJ/$b.o"
  else
    objs="$objs $SRC/.obj/$b.o"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/honeybadgermpc_amd/lib/libhbmpc_hip_$NAME.so" $objs
echo "built libhbmpc_hip_$NAME.so"
