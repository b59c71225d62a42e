"""Thin torch-tensor wrappers over the op-level C entry points (mgea_op_* in include/mgea.h).
Used by the parity tests to check single kernels; the engines do not go through these."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import SamplerConfig, check, ptr, stream_ptr


def _dev(t: torch.Tensor) -> torch.Tensor:
    if t.device.type != "cuda":
        raise RuntimeError("mgea ops need ROCm device tensors; there is no CPU path")
    return t.contiguous()


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, split_k: int = 0) -> torch.Tensor:
    """a [M,K] @ w[N,K]^T (+ bias): exact-fp32 MFMA GEMM with deterministic split-K."""
    lib = _lib.load()
    a, w = _dev(a.float()), _dev(w.float())
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({M}x{K} and {w.shape[1]}x{N})")
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    slabs = split_k if split_k > 0 else (32 if M <= 64 else 1)
    ws = torch.empty(max(1, lib.mgea_op_gemm_workspace_floats(M, N, slabs)), dtype=torch.float32, device=a.device)
    b = None if bias is None else _dev(bias.float())
    check(lib.mgea_op_gemm_f32(ptr(a), ptr(w), ptr(b), ptr(out), M, N, K, split_k, ptr(ws), stream_ptr()))
    return out


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    lib = _lib.load()
    x, w, b = _dev(x.float()), _dev(w.float()), _dev(b.float())
    M, Cd = x.shape
    y = torch.empty_like(x)
    check(lib.mgea_op_layernorm(ptr(x), ptr(w), ptr(b), ptr(y), M, Cd, float(eps), stream_ptr()))
    return y


def attention(qkv: torch.Tensor, n_head: int, lens: Optional[torch.Tensor] = None,
              mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """qkv [B,T,3C] -> [B,T,C]; non-causal, keys valid per lens / mask."""
    lib = _lib.load()
    qkv = _dev(qkv.float())
    B, T, C3 = qkv.shape
    Cd = C3 // 3
    out = torch.empty(B, T, Cd, dtype=torch.float32, device=qkv.device)
    l32 = None if lens is None else _dev(lens.to(torch.int32))
    m32 = None if mask is None else _dev(mask.to(torch.int32))
    check(lib.mgea_op_attention_f32(ptr(qkv), ptr(l32), ptr(m32), ptr(out), B, T, n_head, Cd // n_head, stream_ptr()))
    return out


def gemm_bf16(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
              gelu: bool = False, ln: Optional[dict] = None, info: Optional[list] = None, out: Optional[torch.Tensor] = None):
    """bf16 perf-mode GEMM: (a [M,K] bf16) @ (w [N,K] bf16)^T + bias (fp32) [+GELU | +res (bf16)] -> bf16.
    ln (persistent 256 x 256 kernel only) selects a LayerNorm-folding epilogue of the big-batch DistilBERT pipeline (mgea.h):
      dict(rowstat=[M,2], c1=[N])                      -> epi 3 / 4 (gelu): rstd (a w'^T - mean c1) + bias, bias = c2
      dict(rowstat=[M,2], g=[N], b=[N], stats=True)    -> epi 5: a w^T + bias + LN(res); returns (out, per-tile (sum, M2) [M, N/256, 2])
    info: a list that receives [kernel, half_tiles] of the launch."""
    lib = _lib.load()
    a, w = _dev(a.to(torch.bfloat16)), _dev(w.to(torch.bfloat16))
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    b = None if bias is None else _dev(bias.float())
    r = None if res is None else _dev(res.to(torch.bfloat16))
    epi = 2 if res is not None else (1 if gelu else 0)
    rowstat = c1 = g = be = stats = None
    if ln is not None:
        rowstat = _dev(ln["rowstat"].float())
        if "c1" in ln:
            epi, c1 = (4 if gelu else 3), _dev(ln["c1"].float())
        else:
            epi, g, be = 5, _dev(ln["g"].float()), _dev(ln["b"].float())
            if isinstance(ln.get("stats"), torch.Tensor):
                stats = ln["stats"]                                   # caller's [M, N/256, 2] fp32 buffer (benchmarks: no allocation per call)
            elif ln.get("stats"):
                stats = torch.zeros(M, N // 256, 2, dtype=torch.float32, device=a.device)
    io = (C.c_int32 * 2)(-1, -1)
    check(lib.mgea_op_gemm_bf16_ln(ptr(a), ptr(w), ptr(b), ptr(r), ptr(out), M, N, K, epi, ptr(rowstat), ptr(c1), ptr(g), ptr(be),
                                   ptr(stats), io, stream_ptr()))
    if info is not None:
        info[:] = [int(io[0]), int(io[1])]
    return (out, stats) if stats is not None else out


def ln_rowstat(part: torch.Tensor, C_: int, eps: float) -> torch.Tensor:
    """per-tile (sum, M2) [M, n_part, 2] -> (mean, rstd) [M, 2] of rows of C_ columns."""
    lib = _lib.load()
    part = _dev(part.float())
    M, n_part = part.shape[0], part.shape[1]
    out = torch.empty(M, 2, dtype=torch.float32, device=part.device)
    check(lib.mgea_op_ln_rowstat(ptr(part), ptr(out), M, n_part, int(C_), float(eps), stream_ptr()))
    return out


def fold_ln_bf16(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias: torch.Tensor):
    """-> (bf16(W diag(gamma)) [N,K], c1 [N] = row sums of the rounded product, c2 [N] = bias + W beta)."""
    lib = _lib.load()
    w, gamma, beta, bias = _dev(w.float()), _dev(gamma.float()), _dev(beta.float()), _dev(bias.float())
    N, K = w.shape
    wf = torch.empty(N, K, dtype=torch.bfloat16, device=w.device)
    c1 = torch.empty(N, dtype=torch.float32, device=w.device)
    c2 = torch.empty(N, dtype=torch.float32, device=w.device)
    check(lib.mgea_op_fold_ln_bf16(ptr(w), ptr(gamma), ptr(beta), ptr(bias), N, K, ptr(wf), ptr(c1), ptr(c2), stream_ptr()))
    return wf, c1, c2


def attention_bf16(qkv: torch.Tensor, n_head: int, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    qkv = _dev(qkv.to(torch.bfloat16))
    B, T, C3 = qkv.shape
    Cd = C3 // 3
    out = torch.empty(B, T, Cd, dtype=torch.bfloat16, device=qkv.device)
    m32 = None if mask is None else _dev(mask.to(torch.int32))
    check(lib.mgea_op_attention_bf16(ptr(qkv), ptr(m32), ptr(out), B, T, n_head, Cd // n_head, stream_ptr()))
    return out


def layernorm_bf16(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    lib = _lib.load()
    x = _dev(x.to(torch.bfloat16))
    y = torch.empty_like(x)
    check(lib.mgea_op_layernorm_bf16(ptr(x), ptr(_dev(w.float())), ptr(_dev(b.float())), ptr(y), x.shape[0], x.shape[1],
                                     float(eps), stream_ptr()))
    return y


def sample(logits: torch.Tensor, temperature=1.0, top_k=50, top_p=None, seed=0, step=0, want_probs=False):
    lib = _lib.load()
    logits = _dev(logits.float())
    B, V = logits.shape
    s = SamplerConfig(temperature=float(temperature), top_k=int(top_k) if top_k else 0,
                      top_p=float(top_p) if top_p else 0.0, eos_id=-1, seed=int(seed))
    ids = torch.empty(B, dtype=torch.int32, device=logits.device)
    probs = torch.empty(B, V, dtype=torch.float32, device=logits.device) if want_probs else None
    check(lib.mgea_op_sample(ptr(logits), B, V, C.byref(s), int(step), ptr(ids), ptr(probs), stream_ptr()))
    return (ids, probs) if want_probs else ids


def tile_weights(w: torch.Tensor) -> torch.Tensor:
    """W [N,K] row-major -> the fragment-ordered layout the skinny GEMM reads (rows padded to 32)."""
    lib = _lib.load()
    w = _dev(w.float())
    N, K = w.shape
    out = torch.empty(lib.mgea_op_tiled_weight_floats(N, K), dtype=torch.float32, device=w.device)
    check(lib.mgea_op_tile_weights(ptr(w), N, K, ptr(out), stream_ptr()))
    return out


def tile_rows(x: torch.Tensor) -> torch.Tensor:
    """[M<=512, N] row-major -> k-tiled activation buffer (whole 64-row groups of 64 * N floats; rows >= M are zero)."""
    lib = _lib.load()
    x = _dev(x.float())
    M, N = x.shape
    out = torch.zeros((M + 63) // 64 * 64 * N, dtype=torch.float32, device=x.device)
    check(lib.mgea_op_tile_rows(ptr(x), ptr(out), M, N, 1, stream_ptr()))
    return out


def untile_rows(t: torch.Tensor, M: int, N: int) -> torch.Tensor:
    lib = _lib.load()
    out = torch.empty(M, N, dtype=torch.float32, device=t.device)
    check(lib.mgea_op_tile_rows(ptr(_dev(t)), ptr(out), M, N, 0, stream_ptr()))
    return out


def fold_ln(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias: torch.Tensor):
    """LayerNorm folded into the matrix it feeds -> (tiled gamma * W, c1, c2); see mgea_op_fold_ln."""
    lib = _lib.load()
    w, gamma, beta, bias = _dev(w.float()), _dev(gamma.float()), _dev(beta.float()), _dev(bias.float())
    N, K = w.shape
    wt = torch.empty(lib.mgea_op_tiled_weight_floats(N, K), dtype=torch.float32, device=w.device)
    c1 = torch.empty(N, dtype=torch.float32, device=w.device)
    c2 = torch.empty(N, dtype=torch.float32, device=w.device)
    check(lib.mgea_op_fold_ln(ptr(w), ptr(gamma), ptr(beta), ptr(bias), N, K, ptr(wt), ptr(c1), ptr(c2), stream_ptr()))
    return wt, c1, c2


def skinny(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, *, residual: Optional[torch.Tensor] = None,
           act: int = 0, ln: Optional[tuple] = None, dbg: int = 0):
    """The fused decode-step GEMM on row-major inputs (tiling / folding done here): epilogue `residual` (returns
    (residual + a @ w^T + bias, per-16-column (mean, M2) statistics)) or activation `act` (0 none, 1 GELU,
    2 ReLU).  ln = (gamma, beta, stats [M, K/16, 2]) applies LayerNorm to `a` from 16-column partial statistics."""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    at = tile_rows(a)
    c1 = st = None
    n_part = 0
    if ln is not None:
        wt, c1, b = fold_ln(w, ln[0], ln[1], bias)
        st = _dev(ln[2].float())
        n_part = st.shape[1]
    else:
        wt, b = tile_weights(w), _dev(bias.float())
    stats_out = torch.zeros(max(64, M) * (N // 16) * 2 + 4096, dtype=torch.float32, device=a.device)
    if residual is not None:
        out = tile_rows(residual)
        check(lib.mgea_op_skinny(1, ptr(at), ptr(wt), ptr(b), ptr(c1), ptr(st), n_part, 16, ptr(out), ptr(stats_out),
                                 M, N, K, 0, dbg, stream_ptr()))
        return untile_rows(out, M, N), stats_out[:M * (N // 16) * 2].view(M, N // 16, 2)
    out = torch.zeros((M + 63) // 64 * 64 * N, dtype=torch.float32, device=a.device)
    check(lib.mgea_op_skinny(2, ptr(at), ptr(wt), ptr(b), ptr(c1), ptr(st), n_part, 16, ptr(out), ptr(stats_out),
                             M, N, K, act, dbg, stream_ptr()))
    return untile_rows(out, M, N)


def head(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, want_logits: bool = True):
    """The LM head of a decode step (api_cache.py:105) on row-major inputs: returns (logits [M, N] or None, argmax [M] int64 merged from
    the per-workgroup (max, argmax) partials the kernel leaves for the greedy tail, partial count P).  Which kernel runs -- the balanced
    one-round kernel of csrc/head_gemm.hip or the generic skinny kernel -- follows the library's own routing (switch head_balanced)."""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    at, wt, b = tile_rows(a), tile_weights(w), _dev(bias.float())
    P = int(lib.mgea_op_skinny_logits_partials(M, N, K))
    part = torch.full((2 * 64 * P + 64,), float("nan"), dtype=torch.float32, device=a.device)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=a.device) if want_logits else None
    check(lib.mgea_op_skinny(3, ptr(at), ptr(wt), ptr(b), None, None, 0, 16, ptr(out), ptr(part), M, N, K, 0, 0, stream_ptr()))
    val = part[: 64 * P].view(64, P)[:M]
    idx = part[64 * P: 2 * 64 * P].view(torch.int32).view(64, P)[:M].long()
    best = val.max(1, keepdim=True).values
    cand = torch.where(val == best, idx, torch.full_like(idx, 2 ** 31 - 1))
    return out, cand.min(1).values, P
