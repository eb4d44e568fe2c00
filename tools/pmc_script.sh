#!/bin/bash
# rocprofv3 --pmc passes over any script (tools/pmc_script.sh <out.json> <script relative to the repo> [args]); as tools/pmc_layer.sh: (each counter set in its own pass, --kernel-trace only), summarised by
# tools/pmc_summary.py.      tools/pmc_layer.sh <out.json> [profile_layer.py arguments]
out=$1; shift; script=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=/tmp/pmc_layer_$$
rm -rf "$d"; mkdir -p "$d"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"; do
    i=$((i + 1))
    (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$d/p$i" -o out -- python "$root/$script" "$@" > "$d/log$i.txt" 2>&1) || tail -5 "$d/log$i.txt"
done
python "$root/tools/pmc_summary.py" "$d" "$out"
