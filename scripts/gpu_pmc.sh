#!/bin/bash
# PMC passes (own runs, kernel-trace only) over a short eager workload.  Usage: bash scripts/gpu_pmc.sh <tag> <precision>
tag=${1:-x}; prec=${2:-bf16x3}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_WAVES" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_${tag}_$i -o pmc -- python scripts/pmc_workload.py $prec 4 > gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "== pass $i: $set" >> gpurun_out/pmc_$tag.txt
  python scripts/pmc_summary.py gpurun_out/pmc_${tag}_$i/pmc_results.db 12 >> gpurun_out/pmc_$tag.txt 2>&1
  rm -rf gpurun_out/pmc_${tag}_$i
done
cat gpurun_out/pmc_$tag.txt | cut -c1-400
