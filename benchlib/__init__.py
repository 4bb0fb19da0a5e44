"""The pieces of bench.py (repo root): workload constants, the CPU baseline, the box record, the roofline attachments, the
stdout line, and the secondary workloads.  `python bench.py` is the only entry point; its contract is in its docstring."""
