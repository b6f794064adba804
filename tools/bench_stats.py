"""Row-statistics kernel alone on the workload's two shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from walkgpt_amd import ops

for M, D in [(32768, 768), (8200, 1024), (32768, 1280)]:
    x = torch.randn(M, D, device="cuda:0").bfloat16()
    for _ in range(5):
        ops.row_stats(x, 1e-6)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100):
        ops.row_stats(x, 1e-6)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 10
    print("M=%d D=%d  %.1f us  %.2f TB/s" % (M, D, us, M * D * 2 / us / 1e6), flush=True)
