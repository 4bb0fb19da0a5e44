#!/bin/bash
# Round 3, first GPU call: the new parity tests (N4 on the device, configs[4] at its benched shape, graph cache, f0=None),
# then the rocprofv3 evidence for configs[2] / configs[4].
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1200 python -m pytest tests/test_gpu_clap.py tests/test_gpu_config5.py tests/test_gpu_nsf.py tests/test_gpu_tools.py "tests/test_gpu_models.py::test_ddim_step_graph_is_kept_across_calls_and_invalidated_by_its_inputs" "tests/test_gpu_models.py::test_ddim_graph_replay_is_bit_identical" -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/r3_call1_tests_tail.txt
bash scripts/gpu_profile_secondary.sh r3 bf16x3
