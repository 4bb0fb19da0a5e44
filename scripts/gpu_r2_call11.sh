#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== shapes"; timeout 600 python scripts/shape_profile.py 3 bf16x3 > gpurun_out/r2_shapes_bf16x3.txt 2> gpurun_out/r2_shapes.err; tail -5 gpurun_out/r2_shapes.err
grep -A14 "== vae\|== vocoder" gpurun_out/r2_shapes_bf16x3.txt | cut -c1-150; grep -A34 "== unet" gpurun_out/r2_shapes_bf16x3.txt | cut -c1-150
echo "== short-K ablation: columns M12480 N320 K=64,128,320,640,1280 | M3120 N640 K=128,640,2560 | qkv 320->960, 640->1920"
out=gpurun_out/r2_shortk_ablation.txt; : > $out
for ns in 2 4; do for dbg in 0 1 2 3; do
  MAA_DMA2=off MAA_DBG=$dbg MAA_DMA_NS=$ns timeout 120 python scripts/shortk_ablate.py 2>/dev/null >> $out
done; done
cat $out
