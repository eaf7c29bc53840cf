#!/bin/bash
# ISA checks on the built library (called by build.sh; fails the build).  Usage: check_isa.sh <libaudiocodecs_amd.so>
#  1. no v_pk_fma_f32 in enc_front_kernel / dec_tail_kernel: packed fp32 FMAs with op_sel broadcasts behind LDS reads returned wrong
#     values in lanes 48..63 when a second wave shared the SIMD (profiles/r3_pk_fma_hazard.md; mechanism not established).  The stem's
#     FMAs are pinned in source (fma_pinned, split16.h) and the library is built with -fno-slp-vectorize; this check is what notices
#     when a compiler upgrade or an extra flag brings them back.
#  1b. in NO kernel a packed fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) with an `op_sel:[..1..]` operand, i.e. one that
#     feeds the HIGH half of a source pair to the low result lane: that is the form that faults (tools/ubench/pk_fma_lds.hip,
#     profiles/r4_pk_fma_repro.md: wrong values in lanes 48..63 whenever MFMAs run on the SIMD and a second wave shares it; the
#     low-half broadcasts `op_sel_hi:[..0..]` the library's scale multiplies compile to never fault).
#  2. no scratch in ANY kernel: a kernel that uses scratch cannot be replayed from a hipGraph once the runtime has resized its scratch
#     buffer, and every entry point of the library may be captured.
set -euo pipefail
so="$1"
objdump=/opt/rocm/lib/llvm/bin/llvm-objdump
readelf=/opt/rocm/lib/llvm/bin/llvm-readelf
tmp="$(mktemp -d)"
trap 'rm -rf "$tmp"' EXIT
cp "$so" "$tmp/lib.so"
(cd "$tmp" && "$objdump" --offloading lib.so >/dev/null 2>&1)     # writes lib.so.N.hipv4-amdgcn-amd-amdhsa--gfx950 beside the copy
bad=0
found=0
for co in "$tmp"/lib.so.*gfx950; do
    [ -e "$co" ] || continue
    "$objdump" -d "$co" > "$tmp/dis.s"
    for k in enc_front_kernel dec_tail_kernel; do
        n=$(awk -v k="$k" '/^[0-9a-f]+ <.*>:/{name=$2} /v_pk_fma_f32/{if (index(name, k)) c++} END{print c+0}' "$tmp/dis.s")
        if grep -q "<.*$k.*>:" "$tmp/dis.s"; then found=$((found+1)); fi
        if [ "$n" != "0" ]; then echo "check_isa: $n v_pk_fma_f32 in $k (profiles/r3_pk_fma_hazard.md)"; bad=1; fi
    done
    n=$(grep -E "v_pk_(fma|mul|add)_f32" "$tmp/dis.s" | grep -c "op_sel:\[" || true)
    if [ "$n" != "0" ]; then
        echo "check_isa: $n packed fp32 instructions select the HIGH half of a source (op_sel) -- profiles/r4_pk_fma_repro.md:"
        awk '/^[0-9a-f]+ <.*>:/{name=$2} /v_pk_(fma|mul|add)_f32/ && /op_sel:\[/{c[name]++} END{for (k in c) print "   ", c[k], k}' "$tmp/dis.s"
        bad=1
    fi
    # .private_segment_fixed_size of EVERY kernel from the code object's metadata notes: every entry point is capturable (DESIGN.md
    # section 1), so every kernel can end up in a hipGraph (round 4: three tap_gemm6 arrangements carried 12 - 32 bytes of spill)
    "$readelf" --notes "$co" > "$tmp/notes.txt" 2>/dev/null || true
    awk '/\.private_segment_fixed_size:/{ps=$2} /\.symbol:/{if (ps != 0) print ps, $2; ps=0}' "$tmp/notes.txt" > "$tmp/scratch.txt"
    if [ -s "$tmp/scratch.txt" ]; then
        while read -r bytes sym; do echo "check_isa: $(echo "$sym" | c++filt | cut -c1-140) uses $bytes bytes of scratch"; done < "$tmp/scratch.txt"
        bad=1
    fi
done
if [ "$found" -lt 2 ]; then echo "check_isa: enc_front_kernel / dec_tail_kernel not found in $so"; exit 1; fi
[ "$bad" = "0" ] && echo "check_isa: ok (no packed fp32 FMAs in the fused chains, no scratch and no high-half-selecting packed fp32 instruction in any kernel)"
exit $bad
