#!/bin/bash
# Round-2 starting point: counter evidence for the ablation variants of the 64x64 LDS-DMA kernel on
# the conv / linear micro-benchmark.  One rocprofv3 --pmc pass per counter group and per MAA_DBG mask (separate runs,
# kernel-trace only -- never combined with other trace domains).  Usage (through gpurun): bash scripts/gpu_pmc_ablate.sh [tag]
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/pmc_ablate_$tag.txt
: > $out
rocprofv3 --list-avail > gpurun_out/pmc_list_avail_$tag.txt 2>&1 || true
groups=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY"
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
  "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY"
  "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
  "GRBM_GUI_ACTIVE FETCH_SIZE"
)
for dbg in 0 1 2 3; do
  for g in "${groups[@]}"; do
    d=gpurun_out/pmc_abl_${tag}_$dbg
    rm -rf $d
    MAA_FORCE_CFG=2 MAA_DBG=$dbg timeout 120 rocprofv3 --kernel-trace --pmc $g -d $d -o pmc -- python scripts/conv_bench.py bf16x3 child > $d.log 2>&1
    echo "== MAA_DBG=$dbg  counters: $g" >> $out
    python scripts/pmc_summary.py $d/pmc_results.db 6 2>&1 | grep -v "^# columns" >> $out
    rm -rf $d $d.log
  done
done
tail -80 $out
