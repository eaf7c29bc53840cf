#!/bin/bash
# Developer loop: recompile ONE translation unit of the library with build.sh's flags and relink (the other objects must exist from a
# full build.sh run), then the ISA checks.  Usage: tools/devbuild.sh <tu> [extra hipcc flags]      e.g. tools/devbuild.sh stream_path
# AC_OUT=<lib.so> AC_TUOBJ=<tu.o>: build a side library (timing variants) without touching the product's object or library.
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
here="$root/audiocodecs_amd/csrc"; obj="$here/build"; out="${AC_OUT:-$root/audiocodecs_amd/libaudiocodecs_amd.so}"
tu="$1"; shift
tuobj="${AC_TUOBJ:-$obj/$tu.o}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-strict-aliasing -fPIC -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=65536 -I"$root/include" "$@" -save-temps=obj -c "$here/$tu.hip" -o "$tuobj"
objs=()
for t in core mimi_path dac_path wavtok_path stream_path ac_api; do if [ "$t" = "$tu" ]; then objs+=("$tuobj"); else objs+=("$obj/$t.o"); fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out.tmp" "${objs[@]}"
bash "$here/check_isa.sh" "$out.tmp" && mv "$out.tmp" "$out"
python3 "$root/tools/mfma_branch_hazard.py" "$(dirname "$tuobj")/$tu-hip-amdgcn-amd-amdhsa-gfx950.s" | tail -1
