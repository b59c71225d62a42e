"""Time the bf16 LayerNorm kernel on the DistilBERT bench shape ([32768, 768]).  usage: python tools/ln_bf16_bench.py [M C]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "music-generation-emotion-adaptive_amd"))
import torch
from mgea import ops

M, C = (int(x) for x in sys.argv[1:3]) if len(sys.argv) >= 3 else (32768, 768)
x = torch.randn(M, C, device="cuda").bfloat16()
w, b = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
for _ in range(5):
    ops.layernorm_bf16(x, w, b, 1e-12)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 100
e0.record()
for _ in range(reps):
    ops.layernorm_bf16(x, w, b, 1e-12)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print(f"layernorm_bf16 [{M}, {C}]: {us:6.1f} us  {M * C * 4 / us / 1e6:5.2f} TB/s of read + write")
