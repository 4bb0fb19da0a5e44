"""A few launches of one UNet 3x3 convolution (640 -> 640 at 5x39, batch 16) and of one GEGLU-sized linear on the ping-pong
engines, for rocprofv3 --pmc passes.  python scripts/pp_pmc_workload.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiogpt_amd.backend import Context  # noqa: E402

ctx = Context("cuda:0", precision="bf16x3")
ctx.op_bench_conv(16, 5, 39, 640, 640, 9, True, 10)
ctx.op_bench_conv(16, 10, 78, 320, 320, 9, True, 10)
ctx.op_bench_conv(16, 10, 78, 320, 2560, 1, True, 10)
ctx.op_bench_conv(16, 5, 39, 640, 640, 1, True, 10)
