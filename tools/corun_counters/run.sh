#!/bin/bash
# usage: tools/corun_counters/run.sh <outdir>   -- co-run, march alone and back to back
# (COUNTER_GROUPS="A B;C D" overrides the counters: names separated by spaces, groups by ';')
cd "$(dirname "$0")/../.." || exit 1
R=$PWD
OUT=${1:-$R/gpurun_out/corun_counters}
mkdir -p "$OUT"
GROUPS_=${COUNTER_GROUPS:-"SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY;SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE;TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum;TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_READ_sum;TCC_HIT_sum TCC_MISS_sum;GRBM_GUI_ACTIVE"}
for mode in corun march back_to_back; do
  ROCP_TOOL_LIBRARIES=$R/tools/corun_counters/libcorun_counters.so AVR_COUNTER_LOG=$OUT/$mode.log \
    AVR_COUNTER_GROUPS="$GROUPS_" AVR_COUNTER_WINDOW_MS=200 \
    timeout -k 10 120 python3 tools/corun_counters/steady.py --mode $mode --seconds 9 --out $OUT/$mode.json > $OUT/$mode.out 2> $OUT/$mode.err || { tail -5 $OUT/$mode.err; exit 1; }
  python3 tools/corun_counters/summarise.py $OUT/$mode.json $OUT/$mode.log | tee $OUT/$mode.txt
done
