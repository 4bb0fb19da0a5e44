#!/bin/bash
# Round 2, first GPU call: correctness of the wide-tile / split-K engine, its sweep against the round-1 engines, the new
# boundary tests, a first bench line and the SQ counters of the old and new dominant kernels.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== dma2 op tests"; timeout 900 python -m pytest tests/test_gpu_dma2.py -q -x --timeout 600 2>&1 | tail -15
echo "== sweep"; timeout 900 python scripts/dma2_sweep.py full > gpurun_out/r2_dma2_sweep.txt 2>&1; tail -45 gpurun_out/r2_dma2_sweep.txt
echo "== ablate"; timeout 400 python scripts/dma2_sweep.py ablate > gpurun_out/r2_dma2_ablate.txt 2>&1; tail -40 gpurun_out/r2_dma2_ablate.txt
echo "== engines bit-identical + batch invariance + end-to-end gates (bf16x3)"
timeout 900 python -m pytest tests/test_gpu_precision.py -q --timeout 800 -k "dma_engine or batch_invariance or end_to_end" 2>&1 | tail -8
echo "== boundary + config-2 tests"; timeout 1500 python -m pytest tests/test_gpu_tools.py tests/test_gpu_config2.py -q --timeout 1200 2>&1 | tail -15
echo "== bench (default policy)"; timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_call1.json 2> gpurun_out/r2_bench_call1.err; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r2_bench_call1.json")); r = d["roofline"]
    print("VALUE", d["value"], d["ms_per_step"]); print({k: r[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "launches")})
    print(r["all_igemm"]); print(list(r["kernel_time_ms"].items())[:14])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r2_bench_call1.err").read()[-1500:])
PY
echo "== bench (round-1 engines)"; MAA_DMA2=off timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('VALUE r1 engines', d['value'], d['ms_per_step'])"
echo "== SQ counters"; cd /tmp
for tag in new old; do
  if [ $tag = old ]; then export MAA_DMA2=off; else unset MAA_DMA2; fi
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_WAVES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /root/repo/gpurun_out/pmc_${tag}_$i -o pmc -- python /root/repo/scripts/pmc_workload.py bf16x3 2 > /root/repo/gpurun_out/pmc_${tag}_$i.log 2>&1
    echo "== $tag pass $i: $set" >> /root/repo/gpurun_out/r2_pmc_sq_$tag.txt
    python /root/repo/scripts/pmc_summary.py /root/repo/gpurun_out/pmc_${tag}_$i/pmc_results.db 10 >> /root/repo/gpurun_out/r2_pmc_sq_$tag.txt 2>&1
    rm -rf /root/repo/gpurun_out/pmc_${tag}_$i
  done
  cut -c1-330 /root/repo/gpurun_out/r2_pmc_sq_$tag.txt
done
