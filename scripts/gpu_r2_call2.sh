#!/bin/bash
# Round 2, second GPU call: L2 prefetch / deeper stages / 128x320 tile variants of the wide-tile engine: correctness, sweep,
# in-pipeline A/B of policies.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== dma2 op tests"; timeout 900 python -m pytest tests/test_gpu_dma2.py -q -x --timeout 600 2>&1 | tail -8
echo "== sweep"; timeout 1200 python scripts/dma2_sweep.py full > gpurun_out/r2_dma2_sweep2.txt 2>&1; tail -12 gpurun_out/r2_dma2_sweep2.txt
echo "== in-pipeline A/B (20 DDIM steps + decode, ms per batch)"
Y=MAA_DMA2; X=MAA_DMA2_N320
timeout 900 python scripts/dma2_inpipe.py \
  "r1=$Y=off" \
  "default=" \
  "y:t0ns2S4=$Y=0,2,0,4,1024 $X=off" \
  "y:t0ns2S4pf=$Y=0,2,0,4,1024,1 $X=off" \
  "y:t0ns2S2pf=$Y=0,2,0,2,1024,1 $X=off" \
  "y:t0ns3S2pf=$Y=0,3,0,2,1024,1 $X=off" \
  "y:t0ns4S2pf=$Y=0,4,0,2,1024,1 $X=off" \
  "y:t0p4S2=$Y=0,4,1,2,1024 $X=off" \
  "y:t0p4S2pf=$Y=0,4,1,2,1024,1 $X=off" \
  "y:t0p5S2=$Y=0,5,1,2,1024 $X=off" \
  "y:t0p4S4pf=$Y=0,4,1,4,1024,1 $X=off" \
  "y:t1p3S2pf=$Y=1,3,1,2,1024,1 $X=off" \
  "x:t2S2pf=$Y=off $X=2,2,0,2,1024,1" \
  "x:t2S3pf=$Y=off $X=2,2,0,3,1024,1" \
  "x:t2S2=$Y=off $X=2,2,0,2,1024" \
  "x:t0ns2S2pf=$Y=off $X=0,2,0,2,1024,1" \
  "x:t0p4S2pf=$Y=off $X=0,4,1,2,1024,1" \
  "x:t1p3S3pf=$Y=off $X=1,3,1,3,1024,1" \
  "xy:a=$Y=0,2,0,4,1024,1 $X=2,2,0,2,1024,1" \
  "xy:b=$Y=0,4,1,2,1024,1 $X=2,2,0,3,1024,1" \
  "xy:c=$Y=0,4,1,2,1024,1 $X=2,2,0,2,1024,1" \
  "xy:all640=$Y=0,4,1,2,640,1 $X=2,2,0,2,1024,1" \
  "r1 again=$Y=off" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_dma2_inpipe.txt
