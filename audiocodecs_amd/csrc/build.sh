#!/bin/bash
# Build libaudiocodecs_amd.so for gfx950 (MI355X) in-tree.  Usage: build.sh [extra hipcc flags]
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../libaudiocodecs_amd.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -fno-slp-vectorize: the SLP vectoriser turns the stem's scalar fp32 FMAs (enc_front.h) into v_pk_fma_f32 with op_sel
# broadcasts, and THAT code returned wrong values in lanes 48..63 of one FMA group per ~100 chunks whenever a second wave shared the
# SIMD (run-to-run different; never with one wave per SIMD, never without the packed FMAs -- profiles/r3_pk_fma_hazard.md).
# The flag costs the other kernels nothing measurable (19.2 -> 19.5 ms per step, inside the box-to-box spread).
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared -Wall -Wno-unused-function \
    -I"$here/../../include" "$@" -o "$out" "$here/ac_api.hip"
echo "built $out"
