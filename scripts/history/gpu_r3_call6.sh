#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_precision.py -m gpu -q --timeout 300 -p no:cacheprovider -k "attention" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_encoders.py -m gpu -q --timeout 600 -p no:cacheprovider -k "unet or clap_text or openclip" 2>&1 | tail -6
timeout 300 python scripts/attn_bench.py bf16x3 2>&1 | tee gpurun_out/r3_attn_bench.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/pmc_attn -o pmc -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py bf16x3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_attn/pmc_results.db 8 | grep -v "^# columns" | cut -c1-300 | tee $GRAFT_REPO_ROOT/gpurun_out/r3_attn_pmc_lds.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_attn
