#!/bin/bash
# A variant of the library whose k_mm8w passes are generated with extra environment (gen_mm8w.py's HB_GEN_MM8W_* knobs), for A/B
# timing through HBMPC_HIP_LIB; only hb_mfma_wide.hip is recompiled, the other objects come from the regular build:
#   scratch/build_wide_variant.sh spread2 HB_GEN_MM8W_SPREAD=2    ->  honeybadgermpc_amd/lib/libhbmpc_hip_spread2.so
set -e
NAME="$1"; shift
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/honeybadgermpc_amd/csrc"
TMP="$SRC/.obj_wv_$NAME"
rm -rf "$TMP"; mkdir -p "$TMP"
cp "$SRC"/*.hip "$SRC"/*.hpp "$SRC"/*.inc "$SRC"/gen_mm8w.py "$TMP"/
( cd "$TMP" && env "$@" python3 gen_mm8w.py > /dev/null )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=default -Wno-unused-result -I"$ROOT/include" -I"$SRC" -c "$TMP/hb_mfma_wide.hip" -o "$TMP/hb_mfma_wide.o"
objs=""
for o in "$SRC"/.obj/*.o; do [ "$(basename "$o")" = hb_mfma_wide.o ] || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/honeybadgermpc_amd/lib/libhbmpc_hip_$NAME.so" $objs "$TMP/hb_mfma_wide.o"
rm -rf "$TMP"
echo "built libhbmpc_hip_$NAME.so"
