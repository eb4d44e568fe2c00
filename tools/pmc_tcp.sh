#!/bin/bash
# Vector-memory path counters (TCP / TA / TD) of tools/profile_layer.py --bn, one rocprofv3 --pmc pass per counter pair;
# summary -> gpurun_out/pmc_tcp.json.      tools/pmc_tcp.sh
root=$(cd "$(dirname "$0")/.." && pwd)
d=/tmp/pmc2; rm -rf $d; mkdir -p $d
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_BUSY_avr" "TD_TD_BUSY_sum TCP_GATE_EN1_sum" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM" "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d/p$i -o out -- python $root/tools/profile_layer.py --bn > $d/log$i.txt 2>&1) || tail -3 $d/log$i.txt
done
python $root/tools/pmc_summary.py $d $root/gpurun_out/pmc_tcp.json
