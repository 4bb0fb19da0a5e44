#!/bin/bash
# First GPU call of round 2: validate and A/B the experimental kernels that were written after round 1's GPU budget was
# spent (MAA_DMA_LEAN: igemm_dma_lean.hip; MAA_DMA_PIPE: igemm_dma_pipe_kernel), then the counter passes.
#   bash scripts/gpu_round2_first.sh            (through gpurun, ~4 GPU-minutes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== correctness of the lean kernel (bit-identical to the register engine over a UNet forward + bf16x3 end-to-end gates)"
MAA_DMA_LEAN=1 MAA_DMA_NOWIDEN=1 python -m pytest tests/test_gpu_precision.py -m gpu -q --timeout 600 -k "dma_engine or (end_to_end and bf16x3) or bf16x3_ddim" 2>&1 | tail -3
echo "== micro-benchmark (us per launch; columns as scripts/conv_bench.py)"
for e in "MAA_FORCE_CFG=2" "MAA_FORCE_CFG=2 MAA_DMA_LEAN=1"; do echo "-- $e"; env $e python scripts/conv_bench.py bf16x3 child 2>&1 | tail -1; done
echo "== in-pipeline A/B (20 DDIM steps + decode)"
bash scripts/gpu_knobs.sh bf16x3 "X=1" "MAA_DMA_LEAN=1" "MAA_DMA_LEAN=1 MAA_DMA_NOWIDEN=1" "X=2"
