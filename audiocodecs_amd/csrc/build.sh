#!/bin/bash
# Build libaudiocodecs_amd.so for gfx950 (MI355X) in-tree.  Usage: build.sh [extra hipcc flags]
# Six translation units (core.h has the map), compiled in parallel, linked into one shared library.
# -fno-slp-vectorize: the SLP vectoriser turns the stem's scalar fp32 FMAs (enc_front.h) into v_pk_fma_f32 with op_sel
# broadcasts, and THAT code returned wrong values in lanes 48..63 of one FMA group per ~100 chunks whenever a second wave shared the
# SIMD (run-to-run different; never with one wave per SIMD, never without the packed FMAs -- profiles/r3_pk_fma_hazard.md).
# The flag costs the other kernels nothing measurable (19.2 -> 19.5 ms per step, inside the box-to-box spread).  The stem itself no
# longer depends on it (fma_pinned, split16.h) and check_isa.sh disassembles the result: a packed fp32 FMA or scratch in the fused
# chains fails the build.
# -fno-strict-aliasing: the fused chains view one LDS region through differently typed pointers (fp16 planes, fp32 rows, dec_tail.h /
# enc_front.h); the hand-overs carry compiler fences, and with this flag correctness does not rest on their placement.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="${AC_OUT:-$here/../libaudiocodecs_amd.so}"     # AC_OUT / AC_OBJ: developer builds beside the product (timing variants)
obj="${AC_OBJ:-$here/build}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -pragma-unroll-threshold: the staged epilogue's pass loops (tap_gemm6.h, `#pragma unroll` over the wave's column tiles and row passes) hold
# the paired store loop twice; at the default 16 k the unroller gives up on them, the accumulator index becomes dynamic and the tile goes
# to scratch (check_isa.sh fails the build on that).
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-strict-aliasing -fPIC -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=65536 -I"$here/../../include" "$@")
# The structural scans at the end read hipcc's device assembly; what they accept was validated against ONE compiler.  A different hipcc
# may schedule around the requests it cannot see (tap_gemm8.h) differently: fail loudly, BEFORE compiling, instead of trusting stale scans
# (AC_ALLOW_HIPCC=1 builds anyway, e.g. to re-validate: run the GPU bit-identity tests, then update the version here).
HIPCC_VALIDATED="7.2.26015"
hipcc_ver="$("$HIPCC" --version | sed -n 's/^HIP version: *//p' | head -1)"
echo "hipcc: HIP version $hipcc_ver (scans validated with $HIPCC_VALIDATED*)"
case "$hipcc_ver" in
    "$HIPCC_VALIDATED"*) ;;
    *) if [ "${AC_ALLOW_HIPCC:-0}" != "1" ]; then
           echo "build.sh: hipcc $hipcc_ver differs from the version the ISA scans were validated with ($HIPCC_VALIDATED): re-validate (GPU bit-identity tests) and update HIPCC_VALIDATED, or set AC_ALLOW_HIPCC=1" >&2
           exit 1
       fi ;;
esac
mkdir -p "$obj" "$obj/dev"
pids=()
TUS=(core mimi_path dac_path wavtok_path stream_path ac_api)
# -save-temps=obj: the device assembly of every translation unit stays beside its object for tools/mfma_branch_hazard.py (below)
for tu in "${TUS[@]}"; do
    "$HIPCC" "${FLAGS[@]}" -save-temps=obj -c "$here/$tu.hip" -o "$obj/$tu.o" &
    pids+=($!)
done
# The DEVELOPER library (split16.h AC_DEV_MODE): the same sources with -DAC_DEVELOPER for the two translation units that act on the
# timing / fault-injection words (the fused blocks and the LSTM in core.hip, the switch table in ac_api.hip); the other objects are shared.
# tests/test_fault_injection_gpu.py and tools/experiments load it through AUDIOCODECS_AMD_LIB; the product never does.
for tu in core ac_api; do
    "$HIPCC" "${FLAGS[@]}" -DAC_DEVELOPER -c "$here/$tu.hip" -o "$obj/dev/$tu.o" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
# Link to a temporary path, run EVERY gate on it, install only what passed: a rejected build (scratch in a kernel: unsafe under hipGraph
# replay; a miscounted vmcnt: silent stale-LDS results) must not be left where _native.py would load it.
tmp_out="$out.tmp.$$"
dev_out="$(dirname "$out")/libaudiocodecs_amd_dev.so"
trap 'rm -f "$tmp_out" "$dev_out.tmp.$$"' EXIT
objs=(); for tu in "${TUS[@]}"; do objs+=("$obj/$tu.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$tmp_out" "${objs[@]}"
bash "$here/check_isa.sh" "$tmp_out"
asm=()
for tu in "${TUS[@]}"; do asm+=("$obj/$tu-hip-amdgcn-amd-amdhsa-gfx950.s"); done    # (explicit list: no stale .s of removed translation units)
# Round 4: hipcc's hazard recogniser left an MFMA -> taken branch -> v_accvgpr_read of the MFMA's result without wait states in one
# version of rvq16.h's tile loop (run-to-run different tokens; profiles/r4_variants.md).  The scan fails the build if any kernel
# reads an MFMA result across a branch closer than the matrix pipe needs.
scan_log="$obj/mfma_branch_hazard.log"
if ! python3 "$here/../../tools/mfma_branch_hazard.py" "${asm[@]}" > "$scan_log" 2>&1; then cat "$scan_log"; exit 1; fi
tail -1 "$scan_log"
# Round 5: tap_gemm8's counted wait (s_waitcnt vmcnt(A_SLOTS)) is only right while hipcc emits exactly A_SLOTS buffer loads behind the
# LDS-DMA requests of every stage, on every path (tap_gemm8.h; the race of profiles/r4_tapgemm8.md section 2.1 passed every test).
scan_log="$obj/tap8_pipeline_scan.log"
if ! python3 "$here/../../tools/tap8_pipeline_scan.py" "${asm[@]}" > "$scan_log" 2>&1; then cat "$scan_log"; exit 1; fi
tail -1 "$scan_log"
# Round 6: the product carries no developer switch
if strings "$tmp_out" | grep -q "AC_RB6_DBG\|AC_LSTM_DBG"; then echo "build.sh: the product library reads a developer environment switch" >&2; exit 1; fi
mv "$tmp_out" "$out"
echo "built $out"
if [ -z "${AC_OUT:-}" ]; then      # (side builds of the product with extra flags do not rebuild the developer library)
    devobjs=(); for tu in "${TUS[@]}"; do if [ "$tu" = core ] || [ "$tu" = ac_api ]; then devobjs+=("$obj/dev/$tu.o"); else devobjs+=("$obj/$tu.o"); fi; done
    "$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$dev_out.tmp.$$" "${devobjs[@]}"
    bash "$here/check_isa.sh" "$dev_out.tmp.$$" > /dev/null
    mv "$dev_out.tmp.$$" "$dev_out"
    echo "built $dev_out (developer switches: fault injection, timing modes)"
fi
