#!/bin/bash
# Round-5 validation, third call (what is left of the GPU budget, ~100 s): the drop-in tool chains SERIALLY.  In the second call eight xdist
# workers each ran the CPU oracle on all host cores at once; the heavy tool tests (37 - 41 s each alone) ran into the 150-s per-test limit.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 104 python -m pytest "tests/test_gpu_tools.py::test_T2A_txt2audio_matches_oracle_chain[bf16x3]" tests/test_gpu_tools.py::test_I2A_img2audio_matches_oracle_chain tests/test_gpu_tools.py::test_Inpaint_inference_mel_matches_oracle_chain tests/test_gpu_tools.py::test_T2A_inference_writes_a_wav_file -m gpu -v --timeout 100 -p no:cacheprovider 2>&1 | grep -E "PASSED|FAILED|ERROR|passed|failed|Error" | tee gpurun_out/r5_gpu_tests_third_call_tools_serial.txt
cp gpurun_out/parity.jsonl gpurun_out/r5_parity_third_call.jsonl 2>/dev/null
