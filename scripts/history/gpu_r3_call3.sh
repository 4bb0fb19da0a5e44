#!/bin/bash
# Round 3, third GPU call: ping-pong engines v2 (lean memory phase, 1x1 form): tests, micro-benchmarks, ablations, SQ counters.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pp.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -12 | tee gpurun_out/r3_call3_pp_tests_tail.txt
timeout 900 python scripts/pp_bench.py bf16x3 2>&1 | tee gpurun_out/r3_pp_bench.txt
timeout 600 python scripts/pp_ablate.py 2>&1 | tee gpurun_out/r3_pp_ablate.txt
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc_pp_$i -o pmc -- python $GRAFT_REPO_ROOT/scripts/pp_pmc_workload.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_pp_$i.log 2>&1
  echo "== pass $i: $set" >> $GRAFT_REPO_ROOT/gpurun_out/r3_pp_pmc_sq_counters.txt
  python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_pp_$i/pmc_results.db 12 | grep -v "^# columns" >> $GRAFT_REPO_ROOT/gpurun_out/r3_pp_pmc_sq_counters.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_pp_$i
done
cut -c1-330 $GRAFT_REPO_ROOT/gpurun_out/r3_pp_pmc_sq_counters.txt
