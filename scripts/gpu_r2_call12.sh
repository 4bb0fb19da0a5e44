#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
SK=256,2047
bash scripts/inflight_ab.sh gpurun_out/r2_inflight_policy_ab.txt \
 "default|MAA_X=0" \
 "n320 128x320 S1|MAA_DMA2_N320=2,2,0,1,$SK" \
 "n320 128x320 S2|MAA_DMA2_N320=2,2,0,2,$SK" \
 "n640 128x128 p4 S1|MAA_DMA2_N640=0,4,1,1,$SK" \
 "n640 128x128 ns2 S1|MAA_DMA2_N640=0,2,0,1,$SK" \
 "qkv 128x128 ns2 S1|MAA_DMA2_N960=0,2,0,1,$SK MAA_DMA2_N1920=0,2,0,1,$SK" \
 "qkv 256x128 p3 S1|MAA_DMA2_N960=1,3,1,1,$SK MAA_DMA2_N1920=1,3,1,1,$SK" \
 "all three|MAA_DMA2_N320=2,2,0,1,$SK MAA_DMA2_N640=0,2,0,1,$SK MAA_DMA2_N960=0,2,0,1,$SK MAA_DMA2_N1920=0,2,0,1,$SK" \
 "longK ns2|MAA_DMA2=0,2,0,2,2048 MAA_DMA2_N320=2,2,0,2,2048" \
 "longK ns3p|MAA_DMA2=0,3,1,2,2048 MAA_DMA2_N320=2,2,0,2,2048" \
 "r1 stages 2,2,2|MAA_DMA_NS=2,2,2" \
 "default again|MAA_X=0"
echo "== shapes"; timeout 600 python scripts/shape_profile.py 5 bf16x3 > gpurun_out/r2_shapes_bf16x3.txt 2> gpurun_out/r2_shapes.err; tail -3 gpurun_out/r2_shapes.err
grep -A14 "== vae\|== vocoder" gpurun_out/r2_shapes_bf16x3.txt | cut -c1-150; grep -A40 "== unet" gpurun_out/r2_shapes_bf16x3.txt | cut -c1-150
