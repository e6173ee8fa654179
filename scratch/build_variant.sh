#!/bin/bash
# Builds a variant of the library with extra compiler flags, for A/B runs through HBMPC_HIP_LIB:
#   scratch/build_variant.sh timing -DHB_MM8_TIMING   ->  honeybadgermpc_amd/lib/libhbmpc_hip_timing.so
set -e
NAME="$1"; shift
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/honeybadgermpc_amd/csrc"
OBJ="$SRC/.obj_$NAME"
mkdir -p "$OBJ"
python3 "$SRC/gen_fused.py" > /dev/null
python3 "$SRC/gen_mm8.py" > /dev/null
python3 "$SRC/gen_mm8w.py" > /dev/null
pids=""
for src in "$SRC"/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=default -Wno-unused-result "$@" -c "$src" -o "$OBJ/$(basename "${src%.hip}").o" &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/honeybadgermpc_amd/lib/libhbmpc_hip_$NAME.so" "$OBJ"/*.o
echo "built libhbmpc_hip_$NAME.so"
