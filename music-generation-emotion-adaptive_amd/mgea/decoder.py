"""Host side of the MI355X decoder engine: weight arena, handle lifetime, torch tensor plumbing.

PyTorch is used only for device memory, streams and (in mgea.dist) the RCCL broadcast; all
arithmetic happens in libmgea_hip.so.  Mirrors GPTWithKV / sample_kvcache of the reference
(api_cache.py:76-106, 159-184) -- see generate_music/generate.py for the drop-in names.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import re
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import DecoderConfig, SamplerConfig, check, ptr

_LAYER_TENSORS = ["ln1.weight", "ln1.bias", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                  "attn.out_proj.bias", "ln2.weight", "ln2.bias", "mlp.0.weight", "mlp.0.bias", "mlp.2.weight",
                  "mlp.2.bias"]


def remap_state_dict(old_sd: Dict) -> Dict:
    """Training-checkpoint names -> model names, same mapping as the reference's
    remap_state_dict (api_cache.py:118-134): emb->tok_emb, pos->pos_emb, fc->head,
    tr.layers.N.{self_attn,norm1,norm2,linear1,linear2} -> layers.N.{attn,ln1,ln2,mlp.0,mlp.2}."""
    table = [(r"^emb\.weight$", "tok_emb.weight"), (r"^pos$", "pos_emb"), (r"^fc\.", "head."),
             (r"^tr\.layers\.(\d+)\.self_attn", r"layers.\1.attn"), (r"^tr\.layers\.(\d+)\.norm1", r"layers.\1.ln1"),
             (r"^tr\.layers\.(\d+)\.norm2", r"layers.\1.ln2"), (r"^tr\.layers\.(\d+)\.linear1", r"layers.\1.mlp.0"),
             (r"^tr\.layers\.(\d+)\.linear2", r"layers.\1.mlp.2")]
    new_sd = {}
    for k, v in old_sd.items():
        k2 = k
        for pat, rep in table:
            k2 = re.sub(pat, rep, k2)
        new_sd[k2] = v
    return new_sd


def geometry_from_state_dict(sd: Dict) -> Dict[str, int]:
    """Infer (n_layer, seq_len, d_model, vocab, d_ff) from tensor shapes like api_cache.py:31-37."""
    sd = remap_state_dict(sd)
    n_layer = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
    seq_len, d_model = (int(x) for x in sd["pos_emb"].shape)
    return dict(n_layer=n_layer, seq_len=seq_len, d_model=d_model, vocab=int(sd["tok_emb.weight"].shape[0]),
                d_ff=int(sd["layers.0.mlp.0.weight"].shape[0]))


def _as_f32(t, device):
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t))
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def arena_layout(geometry: Dict[str, int], n_head: int = 8):
    """(offsets, total_floats) of the canonical weight arena for a geometry -- what a rank that
    only RECEIVES the RCCL broadcast needs to size its buffer."""
    lib = _lib.load()
    cfg = DecoderConfig(vocab=geometry["vocab"], seq_len=geometry["seq_len"], d_model=geometry["d_model"],
                        n_head=n_head, n_layer=geometry["n_layer"], d_ff=geometry["d_ff"], max_batch=1, max_ctx=1,
                        dtype=_lib.DTYPE_F32, block_mode=0, pos_mode=0, ln_eps=1e-5)
    n, total = C.c_int32(0), C.c_int64(0)
    check(lib.mgea_decoder_arena_layout(C.byref(cfg), None, C.byref(n), C.byref(total)))
    offs = (C.c_int64 * n.value)()
    check(lib.mgea_decoder_arena_layout(C.byref(cfg), offs, C.byref(n), C.byref(total)))
    return list(offs), total.value


# training-checkpoint tensor-name suffixes of the matrices the fp16 mode stores in fp16 (everything else stays fp32)
F16_ROUNDED_KEYS = ("self_attn.in_proj_weight", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight", "fc.weight",
                    "attn.in_proj_weight", "attn.out_proj.weight", "mlp.0.weight", "mlp.2.weight", "head.weight")


class DecoderEngine:
    """One native decoder handle on one GPU."""

    def __init__(self, state_dict: Optional[Dict], n_head: int = 8, max_batch: int = 64, max_ctx: Optional[int] = None,
                 device="cuda:0", block_mode: str = "kv", pos_mode: str = "reference", geometry: Optional[Dict] = None,
                 arena: Optional[torch.Tensor] = None, ln_eps: float = 1e-5, dtype: str = "f32"):
        """dtype "f32": the parity mode (fp32 storage, exact-fp32 MFMA; bit-exact greedy ids against the reference).
        dtype "f16": the perf mode of BASELINE configs[4] -- the five projection-matrix kinds and the KV pages are stored
        in fp16 (half the bytes a decode step streams), accumulation / residual stream / LayerNorm / softmax / logits
        stay fp32.  The model it serves is exactly the reference with those matrices rounded to fp16 (F16_ROUNDED_KEYS);
        what differs from an fp32 run of THAT model is only activation and KV rounding (tests/test_gpu_f16.py)."""
        if dtype not in ("f32", "f16"):
            raise ValueError("decoder dtype must be 'f32' or 'f16'")
        self.dtype = dtype
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DecoderEngine needs a ROCm device ('cuda:N'); there is no CPU path")
        geo = dict(geometry) if geometry is not None else geometry_from_state_dict(state_dict)
        self.vocab, self.seq_len, self.d_model = geo["vocab"], geo["seq_len"], geo["d_model"]
        self.n_layer, self.d_ff, self.n_head = geo["n_layer"], geo["d_ff"], int(n_head)
        self.max_batch = int(max_batch)
        self.max_ctx = int(max_ctx if max_ctx is not None else self.seq_len)
        self.cfg = DecoderConfig(vocab=self.vocab, seq_len=self.seq_len, d_model=self.d_model, n_head=self.n_head,
                                 n_layer=self.n_layer, d_ff=self.d_ff, max_batch=self.max_batch, max_ctx=self.max_ctx,
                                 dtype=_lib.DTYPE_F16 if dtype == "f16" else _lib.DTYPE_F32,
                                 block_mode=_lib.BLOCK_PRELN_GELU if block_mode == "kv" else _lib.BLOCK_POSTLN_RELU,
                                 pos_mode=_lib.POS_REFERENCE if pos_mode == "reference" else _lib.POS_ABSOLUTE,
                                 ln_eps=ln_eps)
        n = C.c_int32(0)
        total = C.c_int64(0)
        check(self.lib.mgea_decoder_arena_layout(C.byref(self.cfg), None, C.byref(n), C.byref(total)))
        offs = (C.c_int64 * n.value)()
        check(self.lib.mgea_decoder_arena_layout(C.byref(self.cfg), offs, C.byref(n), C.byref(total)))
        self.offsets = list(offs)
        self.arena_floats = total.value
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        if arena is not None:
            if arena.numel() != self.arena_floats or arena.dtype != torch.float32 or arena.device != self.device:
                raise ValueError("arena tensor has the wrong size / dtype / device")
            self.arena = arena
        else:
            self.arena = self.pack_arena(state_dict, self.cfg_dict(), self.offsets, self.arena_floats, self.device)
        torch.cuda.synchronize(self.device)
        h = C.c_void_p(0)
        check(self.lib.mgea_decoder_create(C.byref(self.cfg), ptr(self.arena), C.byref(h)))
        self.h = h
        self._cur_batch = 0
        self._epoch = 0  # bumps whenever the native cache is reset (guards stale `presents`)
        self._len = 0

    def cfg_dict(self):
        return dict(vocab=self.vocab, seq_len=self.seq_len, d_model=self.d_model, n_layer=self.n_layer, d_ff=self.d_ff)

    @staticmethod
    def tensor_names(n_layer: int) -> List[str]:
        names = ["tok_emb.weight", "pos_emb"]
        for i in range(n_layer):
            names += [f"layers.{i}.{t}" for t in _LAYER_TENSORS]
        return names + ["head.weight", "head.bias"]

    @staticmethod
    def pack_arena(state_dict, geo, offsets, total, device) -> torch.Tensor:
        """Copy every tensor into its slot of the single fp32 arena (canonical order of mgea.h)."""
        sd = remap_state_dict(state_dict)
        names = DecoderEngine.tensor_names(geo["n_layer"])
        missing = [k for k in names if k not in sd]
        if missing:
            raise KeyError(f"Missing key(s) in state_dict: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        arena = torch.zeros(total, dtype=torch.float32, device=device)
        for name, off in zip(names, offsets):
            t = _as_f32(sd[name], device).reshape(-1)
            arena[off:off + t.numel()].copy_(t)
        return arena

    # ------------------------------------------------------------------ lifetime
    def refresh_weights(self):
        """Call after rewriting `self.arena` in place: the engine keeps a decode-layout copy of the matrices."""
        with self._on_stream():
            check(self.lib.mgea_decoder_refresh_weights(self.h, self._sp()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.mgea_decoder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    @contextlib.contextmanager
    def _on_stream(self):
        """Run on the engine's own stream (hipGraph capture is illegal on the null stream),
        ordered after / before the caller's current stream."""
        outer = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(outer)
        with torch.cuda.stream(self.stream):
            yield
        outer.wait_stream(self.stream)

    def _sp(self):
        return C.c_void_p(self.stream.cuda_stream)

    def _check_ids(self, ids: torch.Tensor) -> bool:
        """Token ids outside the vocabulary: nn.Embedding raises IndexError in the reference (api_cache.py:99).
        A HOST tensor is checked here, before the upload, at no GPU cost.  A DEVICE tensor is not read back (that
        would put a host sync in front of every call of the reference's own loop, api_cache.py:166-168): the
        kernels clamp such ids and set a sticky device flag, see id_errors().  Returns True if it checked."""
        if ids.is_cuda:
            return False
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.vocab):
            raise IndexError("index out of range in self")  # what nn.Embedding raises on CPU
        return True

    def id_errors(self, raise_error: bool = True) -> int:
        """Read and clear the engine's sticky device flags (ONE stream sync): bit 0 = some token id handed over as a
        device tensor since the last call was outside the vocabulary (and was clamped)."""
        flags = C.c_int32(0)
        with self._on_stream():
            check(self.lib.mgea_decoder_error_flags(self.h, C.byref(flags), self._sp()))
        if raise_error and (flags.value & 1):
            raise IndexError("index out of range in self")
        return flags.value

    @staticmethod
    def sampler(temperature=1.0, top_k: Optional[int] = 50, top_p: Optional[float] = None, eos_id: int = -1,
                seed: int = 0) -> SamplerConfig:
        return SamplerConfig(temperature=float(temperature), top_k=int(top_k) if top_k else 0,
                             top_p=float(top_p) if top_p else 0.0, eos_id=int(eos_id), seed=int(seed) & (2 ** 64 - 1))

    # ------------------------------------------------------------------ model(idx, past) surface
    def reset(self, batch: int, max_len: Optional[int] = None):
        with self._on_stream():
            check(self.lib.mgea_decoder_reset(self.h, int(batch), int(max_len or self.max_ctx), self._sp()))
        self._cur_batch = int(batch)
        self._epoch += 1
        self._len = 0

    def forward(self, idx: torch.Tensor, lens: Optional[torch.Tensor] = None, want_logits: bool = True):
        """Append idx [B,T] to the cache and run the blocks (GPTWithKV.forward, api_cache.py:87-106)."""
        if idx.dim() != 2:
            raise RuntimeError("idx must be [B, T]")
        B, T = idx.shape
        self._check_ids(idx)
        with self._on_stream():
            ids32 = idx.to(device=self.device, dtype=torch.int32).contiguous()
            lens32 = None if lens is None else lens.to(device=self.device, dtype=torch.int32).contiguous()
            logits = torch.empty(B, T, self.vocab, dtype=torch.float32, device=self.device) if want_logits else None
            check(self.lib.mgea_decoder_forward(self.h, ptr(ids32), ptr(lens32), B, T, ptr(logits), self._sp()))
        self._len += T
        return logits

    def step(self, ids_in: Optional[torch.Tensor], sampler: SamplerConfig, want_logits: bool = False):
        with self._on_stream():
            B = self._batch()
            ids32 = None if ids_in is None else ids_in.to(device=self.device, dtype=torch.int32).contiguous()
            out = torch.empty(B, dtype=torch.int32, device=self.device)
            logits = torch.empty(B, self.vocab, dtype=torch.float32, device=self.device) if want_logits else None
            check(self.lib.mgea_decoder_step(self.h, ptr(ids32), C.byref(sampler), ptr(out), ptr(logits), self._sp()))
        self._len += 1
        return out, logits

    def _batch(self):
        return self._cur_batch

    def context_lengths(self) -> torch.Tensor:
        with self._on_stream():
            out = torch.empty(self._cur_batch, dtype=torch.int32, device=self.device)
            check(self.lib.mgea_decoder_context_lengths(self.h, ptr(out), self._sp()))
        return out

    # ------------------------------------------------------------------ sample_kvcache surface
    def generate(self, prompts, n_steps: int, temperature: float = 1.0, top_k: Optional[int] = 50,
                 top_p: Optional[float] = None, eos_id: int = -1, seed: int = 0, check_ids: bool = True) -> torch.Tensor:
        """Batched sample_kvcache (api_cache.py:159-184).  prompts: list of id lists (ragged ok) or
        an int tensor [B, Tp].  Returns int32 [B, n_steps] of generated ids (-1 after a row's EOS).
        check_ids: prompts given as a DEVICE tensor are range-checked through the device flag once the
        generation has been enqueued (one sync at the end, which the caller's read of the ids needs anyway);
        False skips even that and leaves the flag for id_errors()."""
        if isinstance(prompts, torch.Tensor):
            ids = prompts.to(torch.int32)
            lens = None
        else:
            B = len(prompts)
            Tp = max(len(p) for p in prompts)
            if min(len(p) for p in prompts) < 1:
                raise ValueError("empty prompt")
            ids = torch.zeros(B, Tp, dtype=torch.int32)
            for b, p in enumerate(prompts):
                ids[b, :len(p)] = torch.tensor(list(p), dtype=torch.int32)
            lens = None if all(len(p) == Tp for p in prompts) else torch.tensor([len(p) for p in prompts], dtype=torch.int32)
        B, Tp = ids.shape
        checked = self._check_ids(ids)
        samp = self.sampler(temperature, top_k, top_p, eos_id, seed)
        with self._on_stream():
            ids = ids.to(self.device).contiguous()
            lens = None if lens is None else lens.to(self.device).contiguous()
            out = torch.empty(B, max(n_steps, 1), dtype=torch.int32, device=self.device)
            check(self.lib.mgea_decoder_generate(self.h, ptr(ids), ptr(lens), B, Tp, int(n_steps), C.byref(samp), ptr(out),
                                                 self._sp()))
        self._cur_batch = B
        self._epoch += 1
        if check_ids and not checked:
            self.id_errors()
        return out[:, :n_steps]

    def reset_and_prefill(self, idx: torch.Tensor, lens=None, want_logits=True, max_len=None):
        self.reset(idx.shape[0], max_len)
        return self.forward(idx, lens, want_logits)

    PROFILE_CLASSES = ("gemm", "rowop", "attn_paged", "attn_dense", "sample")

    def profile(self, stride: int):
        """Time every `stride`-th decode step of generate() eagerly with HIP events (0 = off)."""
        check(self.lib.mgea_decoder_profile(self.h, int(stride)))

    def profile_read(self):
        ms = (C.c_double * 8)()
        n = (C.c_int64 * 8)()
        check(self.lib.mgea_decoder_profile_read(self.h, ms, n, 8))
        return {k: dict(ms=ms[i], launches=n[i]) for i, k in enumerate(self.PROFILE_CLASSES)}

    def stats(self):
        out = (C.c_int64 * 8)()
        check(self.lib.mgea_decoder_stats(self.h, out))
        return dict(graph_nodes=out[0], graph_replays=out[1], graph_instantiates=out[2], graphs_cached=out[4], prefill16_forwards=out[5])
