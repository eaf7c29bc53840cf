#!/bin/bash
# Developer lens: one kernel's device assembly out of a hipcc -save-temps .s file, with its resource lines and instruction mix.
# Usage: tools/kstat.sh <file.s> <mangled-name substring> [out.s]
S="$1"; K="$2"; OUT="${3:-/tmp/kstat_$K.s}"
awk -v k="$K" '/^_Z[A-Za-z0-9_]*:/ { on = index($0, k) > 0 } on { print } on && /\.end_amdhsa_kernel/ { exit }' "$S" > "$OUT"
echo "$(wc -l < "$OUT") lines -> $OUT"
grep -E "^; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|SGPRBlocks|NumSgprs)" "$OUT" | tr '\n' ' '; echo
for i in v_mfma ds_read ds_write buffer_load buffer_store global_load v_exp s_waitcnt s_barrier v_accvgpr v_cndmask v_fma_mix v_cvt_pk v_pk_mul s_cbranch v_mov; do
    echo -n "$i=$(grep -cE "^\s+$i" "$OUT") "
done
echo
echo "instructions: $(grep -cE '^\s+[vsdbg][a-z_0-9]+( |$)' "$OUT")  valu: $(grep -cE '^\s+v_' "$OUT")  salu: $(grep -cE '^\s+s_' "$OUT")"
