#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 evidence for the two secondary workloads -- BASELINE configs[2] (HiFi-GAN
# 64 x 1024 frames) and configs[4] on one GPU (inpaint + image-to-audio): kernel trace + stats of the bench command itself,
# then HBM-side FETCH_SIZE / WRITE_SIZE in their own passes (kernel trace only, as the MI355X guide prescribes).
# Usage: bash scripts/gpu_profile_secondary.sh <tag> [precision]
tag=${1:-r3}; prec=${2:-bf16x3}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for wl in ${WORKLOADS:-hifigan64 mixed}; do
  if [ $wl = mixed ]; then steps="--steps 1 --warmup 1"; pmc_args="--steps 1 --warmup 0 --ddim-steps 4 --no-graph"; else steps="--steps 2 --warmup 1"; pmc_args="--steps 1 --warmup 0"; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_$wl -o bench -- python bench.py --workload $wl $steps --precision $prec --no-cpu-baseline > gpurun_out/${tag}_${wl}_bench_under_rocprof.json 2> gpurun_out/${tag}_${wl}_prof.err
  python scripts/prof_summary.py gpurun_out/prof_${tag}_$wl/bench_results.db > gpurun_out/${tag}_${wl}_kernel_stats.txt
  rm -rf gpurun_out/prof_${tag}_$wl
  out=gpurun_out/${tag}_${wl}_pmc_fetch_write.txt
  echo "# rocprofv3 --kernel-trace --pmc <set> -- python bench.py --workload $wl $pmc_args --no-roofline --no-cpu-baseline   (sums over the launches of each kernel; FETCH_SIZE / WRITE_SIZE in KiB-units as rocprofv3 reports them; on gfx950 FETCH_SIZE counts half the bytes of wide streaming reads -- double it)" > $out
  for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
    n=$(echo $set | cut -d_ -f1)
    timeout 600 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_${tag}_${wl}_$n -o pmc -- python bench.py --workload $wl $pmc_args --precision $prec --no-roofline --no-cpu-baseline > gpurun_out/pmc_${tag}_${wl}_$n.log 2>&1
    echo "== $set" >> $out
    python scripts/pmc_summary.py gpurun_out/pmc_${tag}_${wl}_$n/pmc_results.db 80 | grep -v "^# columns" >> $out 2>&1
    rm -rf gpurun_out/pmc_${tag}_${wl}_$n
  done
  # third pass: matrix-pipe occupancy; then the per-launch table bench.py's `secondary` rooflines read (profiles/pmc_traffic.json)
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_${tag}_${wl}_SQ -o pmc -- python bench.py --workload $wl $pmc_args --precision $prec --no-roofline --no-cpu-baseline > gpurun_out/pmc_${tag}_${wl}_SQ.log 2>&1
  echo "== SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" >> $out
  python scripts/pmc_summary.py gpurun_out/pmc_${tag}_${wl}_SQ/pmc_results.db 80 | grep -v "^# columns" >> $out 2>&1
  rm -rf gpurun_out/pmc_${tag}_${wl}_SQ
  if [ $wl = mixed ]; then units=4; else units=1; fi
  python scripts/pmc_traffic_json.py $out $prec $units profiles/pmc_traffic.json --section $wl && cp profiles/pmc_traffic.json gpurun_out/${tag}_pmc_traffic.json
  head -12 gpurun_out/${tag}_${wl}_kernel_stats.txt | cut -c1-170
done
