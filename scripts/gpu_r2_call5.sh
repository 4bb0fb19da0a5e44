#!/bin/bash
# Round 2, fifth GPU call: persistent-workgroup rewrite of the wide-tile engine (now also 64x64 / 128x64 tiles).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== dma2 op tests"; timeout 1200 python -m pytest tests/test_gpu_dma2.py -q -x --timeout 900 2>&1 | tail -6
echo "== engines bit-identical, batch invariance, end-to-end gates, Inpaint.inference"
timeout 1200 python -m pytest tests/test_gpu_precision.py tests/test_gpu_tools.py -q --timeout 900 -k "dma_engine or batch_invariance or end_to_end or files_in_files_out" 2>&1 | tail -6
echo "== sweep (quick)"; timeout 900 python scripts/dma2_sweep.py quick > gpurun_out/r2_dma2_sweep3.txt 2>&1; grep -v "^  \[" gpurun_out/r2_dma2_sweep3.txt | tail -22
echo "== in-pipeline probes"
A=MAA_DMA2
timeout 900 python scripts/dma2_inpipe.py \
  "default=" \
  "no persistence=MAA_DMA2_PERSIST=0" \
  "shortK:64x64 ns4=$A=3,4,0,1,0,2047" \
  "shortK:64x64 ns2=$A=3,2,0,1,0,2047" \
  "shortK:64x64 ns4 pipe=$A=3,4,1,1,0,2047" \
  "shortK:128x64 ns3=$A=4,3,0,1,0,2047" \
  "shortK:128x64 ns4 pipe=$A=4,4,1,1,0,2047" \
  "shortK:64x64 ns4 + geglu 128x128 ns2=$A=3,4,0,1,0,2047 MAA_DMA2_N2560=0,2,0,1,0 MAA_DMA2_N5120=0,2,0,1,0" \
  "shortK:64x64 ns4 + geglu 128x128 p4=$A=3,4,0,1,0,2047 MAA_DMA2_N2560=0,4,1,1,0 MAA_DMA2_N5120=0,4,1,1,0" \
  "shortK:64x64 ns4 nopersist=$A=3,4,0,1,0,2047 MAA_DMA2_PERSIST=0" \
  "r1 engines=$A=off" \
  "default again=" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_dma2_inpipe4.txt
