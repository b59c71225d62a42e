import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "music-generation-emotion-adaptive_amd"))
import torch
from mgea import ops
M = 32768
for N in (2304, 768):
    for K in (768, 1536, 3072, 6144):
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        b = torch.randn(N, device="cuda")
        f = lambda: ops.gemm_bf16(a, w, b, None)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        tiles = (M // 256) * (N // 256); rounds = -(-tiles // 256)
        print(f"N={N:5d} K={K:5d}: {us:8.1f} us {2*M*N*K/us/1e6:7.1f} TFLOP/s  rounds {rounds}  us/round {us/rounds:6.1f}  us per K-tile per round {us/rounds/(K//64):5.2f}", flush=True)
