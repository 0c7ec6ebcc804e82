#!/bin/bash
# Collect PMC counters for the wavefront kernels in separate passes (each pass = one short bench run).
# usage: tools/pmc_passes.sh <outdir> [spp]
set -u
OUT=${1:-gpurun_out/pmc}
SPP=${2:-16}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$ROOT/$OUT"
cd /tmp
i=0
for CNT in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SALU" \
  "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PERF_SEL_TOTAL_READ" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_UTCL1_REQUEST_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$ROOT/$OUT/pass$i" -- python3 "$ROOT/bench.py" --spp "$SPP" --steps 1 --warmup 0 --no-cpu-baseline > "$ROOT/$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$ROOT/$OUT/pass$i.log"; }
done
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT" > "$ROOT/$OUT/summary.txt" 2>&1
cat "$ROOT/$OUT/summary.txt"
