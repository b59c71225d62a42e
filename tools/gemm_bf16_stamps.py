#!/usr/bin/env python3
"""Where a tile of the persistent bf16 GEMM spends its time: in-kernel 100 MHz stamps (thread 0 of every workgroup) from the
tools-only build of the library (tools/build_stamps.sh -> tools/libmgea_hip_stamps.so).  Per unit of a workgroup:
  0 unit start | 1 K-tile 0 landed, K loop starts | 2 K loop done | 3 next unit's prologue issued | 4+2k pass k staged in LDS |
  5+2k pass k read out, its global stores issued
Prints, per DistilBERT shape, the median over workgroups of every interval of units 0..2 in us.
  MGEA_LIB_PATH=tools/libmgea_hip_stamps.so python3 tools/gemm_bf16_stamps.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MGEA_LIB_PATH", os.path.join(ROOT, "tools", "libmgea_hip_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import numpy as np
import torch
from mgea import _lib, ops
lib = _lib.load()
_lib.tune_set("bf16_gemm_tail", 1)          # half units last: units 0.. are whole tiles in every workgroup
lib.mgea_dbg_set_ph_stamps.restype = C.c_int
lib.mgea_dbg_set_ph_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros(256, 64, dtype=torch.int64, device="cuda")
assert lib.mgea_dbg_set_ph_stamps(C.c_void_p(stamps.data_ptr())) == 0
lib.mgea_dbg_set_ph_cycles.restype = C.c_int
lib.mgea_dbg_set_ph_cycles.argtypes = [C.c_void_p]
cycles = torch.zeros(256, 64, dtype=torch.int64, device="cuda")     # s_memtime at the same points: the clock a span was run at
assert lib.mgea_dbg_set_ph_cycles(C.c_void_p(cycles.data_ptr())) == 0
PH = int(os.environ.get("MGEA_BF16_GEMM_PHASES", "0") or 0)
if PH: print(f"bf16_gemm_phases = {PH}")
lib.mgea_dbg_set_ph_same_tile.restype = C.c_int
lib.mgea_dbg_set_ph_same_tile.argtypes = [C.c_int]
SAME = len(sys.argv) > 1 and sys.argv[1] == "same_tile"      # ablation: every unit loads tile 0's operands (L2-resident; wrong results)
assert lib.mgea_dbg_set_ph_same_tile(1 if SAME else 0) == 0
if "_a" in os.path.basename(os.environ["MGEA_LIB_PATH"]):
    print(f"ABLATION build {os.path.basename(os.environ['MGEA_LIB_PATH'])} (bits: 1 no LDS-DMA after the prologue, 2 no ds_reads, 4 no MFMAs; whole-tile K loop): "
          "wrong results, timings only")
if SAME:
    print("ABLATION same_tile: every unit loads the operands of tile 0 (they never leave L2); results are wrong, timings only")
M = 32768
NAMES = {0: "start", 1: "K0 landed", 2: "K loop done", 3: "next prologue issued"}
for name, N, K, epi in [("qkv", 2304, 768, 3), ("out", 768, 768, 5), ("fc1", 3072, 768, 4), ("fc2", 768, 3072, 5), ("qkv0", 2304, 768, 0)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda").bfloat16()
    st = torch.stack([torch.randn(M, device="cuda") * 0.1, torch.rand(M, device="cuda") + 0.5], 1).contiguous()
    kw = {}
    if epi in (3, 4):
        kw = dict(ln=dict(rowstat=st, c1=w.float().sum(1)), gelu=epi == 4)
    elif epi == 5:
        kw = dict(res=r, ln=dict(rowstat=st, g=torch.ones(N, device="cuda"), b=torch.zeros(N, device="cuda"), stats=torch.zeros(M, N // 256, 2, device="cuda")))
    kw["out"] = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(int(os.environ.get('STAMPS_WARM', '3000'))): ops.gemm_bf16(a, w, b, **kw)      # the clock settles under sustained load only
    torch.cuda.synchronize()
    stamps.zero_(); cycles.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_bf16(a, w, b, **kw); e1.record(); torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64) / 100.0      # us
    t0 = s[:, 0].min()
    print(f"== {name} N={N} K={K} epi {epi}: launch {e0.elapsed_time(e1) * 1e3:.1f} us; kernel span by stamps {s.max() - t0:.1f} us; "
          f"workgroup start skew {np.percentile(s[:, 0] - t0, 50):.2f} / {np.percentile(s[:, 0] - t0, 99):.2f} us (median / p99)")
    cy = cycles.cpu().numpy().astype(np.float64)
    ok = (s[:, 2] > s[:, 1]) & (cy[:, 2] > cy[:, 1])
    if ok.any():   # shader clock over the first unit's K loop and over the whole kernel, median over workgroups
        last = np.array([np.flatnonzero(s[i] > 0).max() for i in range(s.shape[0])])
        whole = np.array([(cy[i, last[i]] - cy[i, 0]) / max(s[i, last[i]] - s[i, 0], 1e-9) for i in range(s.shape[0])])
        print(f"  in-kernel clock: K loop of unit 0 {np.median((cy[ok, 2] - cy[ok, 1]) / (s[ok, 2] - s[ok, 1])) / 1e3:.3f} GHz, whole kernel {np.median(whole) / 1e3:.3f} GHz")
    for u in range(3):
        base = u * 12
        if not (s[:, base] > 0).any():
            break
        wg = s[:, base] > 0
        cols = [c for c in range(12) if (s[wg, base + c] > 0).all()]
        line = []
        for c0, c1 in zip(cols[:-1], cols[1:]):
            d = s[wg, base + c1] - s[wg, base + c0]
            lab = NAMES.get(c1, f"pass {(c1 - 4) // 2} {'staged' if c1 % 2 == 0 else 'stored'}")
            line.append(f"{lab} +{np.median(d):.2f}")
        nxt = (u + 1) * 12
        if (s[wg, nxt] > 0).all():
            line.append(f"-> next unit +{np.median(s[wg, nxt] - s[wg, base + cols[-1]]):.2f}")
        print(f"  unit {u} ({int(wg.sum())} workgroups, unit span {np.median(s[wg, base + cols[-1]] - s[wg, base]):.2f} us): " + " | ".join(line))
