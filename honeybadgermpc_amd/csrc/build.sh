#!/bin/bash
# Builds libhbmpc_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/.obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=default -Wno-unused-result"
python3 "$HERE/gen_fused.py" > /dev/null
python3 "$HERE/gen_mm8.py" > /dev/null
python3 "$HERE/gen_mm8w.py" > /dev/null
objs=""
pids=""
for src in "$HERE"/*.hip; do
  obj="$HERE/.obj/$(basename "${src%.hip}").o"
  objs="$objs $obj"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ -n "$(find "$HERE" "$HERE/../../include" -maxdepth 1 \( -name '*.hpp' -o -name '*.h' -o -name '*.inc' \) -newer "$obj" 2>/dev/null)" ]; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libhbmpc_hip.so" $objs
echo "built $OUT/libhbmpc_hip.so"
# CPython helper for list[int] <-> limb marshalling (plumbing only)
PYINC="$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
if [ ! -f "$OUT/_hbmarshal.so" ] || [ "$HERE/hb_pymarshal.c" -nt "$OUT/_hbmarshal.so" ]; then
  gcc -O2 -shared -fPIC -I"$PYINC" "$HERE/hb_pymarshal.c" -o "$OUT/_hbmarshal.so"
fi
