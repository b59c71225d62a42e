"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the reference's
KV-cache MIDI-token decoder.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinned against golden vectors produced by the reference's own classes
(tests/golden/make_golden.py -> tests/golden/decoder_*.npz; tests/test_oracle_decoder.py).

What is restated (reference file:line):
  * block      : api_cache.py:51-74   pre-LN, nn.MultiheadAttention with NO mask, exact-erf GELU MLP
  * model      : api_cache.py:87-106  tok_emb[idx] + pos_emb[:T]  (T == 1 on every decode step,
                                      so every generated token gets position row 0)
  * key remap  : api_cache.py:118-134 training-checkpoint names -> model names
  * sampler    : api_cache.py:159-184 prefill (logits dropped), re-feed of the last prompt token,
                                      /temperature, top-k additive -1e10 mask, softmax, multinomial
  * no-KV twin : generate_music/generate.py:25-35,46-61  post-LN ReLU nn.TransformerEncoder,
                                      full recompute with true positions

Deliberate difference from the reference's *mechanics* (not its results): the reference caches
ln1(x) and re-projects K/V for the whole past each step (api_cache.py:62-68); this restatement
caches the projected K/V.  The projection is row-wise, so the numbers are identical up to fp32
summation order (checked against the golden logits to 1e-5).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _t(a) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.detach().to(torch.float32).cpu().contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class DecoderRef:
    def __init__(self, state_dict: Dict[str, "np.ndarray | torch.Tensor"], n_head: int = 8):
        sd = {k: _t(v) for k, v in state_dict.items()}
        self.tok_emb = sd["emb.weight"]
        self.pos_emb = sd["pos"]
        self.head_w, self.head_b = sd["fc.weight"], sd["fc.bias"]
        self.V, self.C = self.tok_emb.shape
        self.seq_len = self.pos_emb.shape[0]
        self.H = n_head
        assert self.C % n_head == 0
        self.dh = self.C // n_head
        idx = sorted({int(k.split(".")[2]) for k in sd if k.startswith("tr.layers.")})
        self.layers = []
        for i in idx:
            p = f"tr.layers.{i}."
            self.layers.append(dict(
                in_w=sd[p + "self_attn.in_proj_weight"], in_b=sd[p + "self_attn.in_proj_bias"],
                out_w=sd[p + "self_attn.out_proj.weight"], out_b=sd[p + "self_attn.out_proj.bias"],
                l1_w=sd[p + "linear1.weight"], l1_b=sd[p + "linear1.bias"],
                l2_w=sd[p + "linear2.weight"], l2_b=sd[p + "linear2.bias"],
                n1_w=sd[p + "norm1.weight"], n1_b=sd[p + "norm1.bias"],
                n2_w=sd[p + "norm2.weight"], n2_b=sd[p + "norm2.bias"]))
        self.NL = len(self.layers)

    # ---------------------------------------------------------------- KV-cache model
    def _attend(self, q, k, v, key_valid):
        """q [B,Tq,C], k/v [B,Tk,C], key_valid [B,Tk] bool or None -> [B,Tq,C]; no causal mask."""
        B, Tq, C = q.shape
        Tk = k.shape[1]
        H, dh = self.H, self.dh
        q = q.view(B, Tq, H, dh).transpose(1, 2)
        k = k.view(B, Tk, H, dh).transpose(1, 2)
        v = v.view(B, Tk, H, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
        if key_valid is not None:
            s = s.masked_fill(~key_valid[:, None, None, :], float("-inf"))
        p = torch.softmax(s, dim=-1)
        return (p @ v).transpose(1, 2).reshape(B, Tq, C)

    @torch.no_grad()
    def forward(self, idx: torch.Tensor, cache: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None,
                cache_valid: Optional[torch.Tensor] = None, new_valid: Optional[torch.Tensor] = None):
        """idx [B,T] int64.  cache: per layer (K, V) each [B,Tp,C] of *projected* keys/values.
        cache_valid [B,Tp] / new_valid [B,T]: which cached / new positions are real tokens
        (padding of ragged prompts is masked out so each row equals its solo run).
        Returns logits [B,T,V], new cache, new validity."""
        B, T = idx.shape
        C = self.C
        if T > self.seq_len:
            raise RuntimeError(f"T={T} exceeds the position table ({self.seq_len} rows) (api_cache.py:99)")
        x = self.tok_emb[idx] + self.pos_emb[:T]
        if cache is None:
            cache = [None] * self.NL
        if new_valid is None:
            new_valid = torch.ones(B, T, dtype=torch.bool)
        if cache[0] is not None and cache_valid is None:
            cache_valid = torch.ones(B, cache[0][0].shape[1], dtype=torch.bool)
        valid = new_valid if cache[0] is None else torch.cat([cache_valid, new_valid], 1)
        mask = None if bool(valid.all()) else valid
        presents = []
        for L, past in zip(self.layers, cache):
            xn = F.layer_norm(x, (C,), L["n1_w"], L["n1_b"], 1e-5)
            qkv = xn @ L["in_w"].t() + L["in_b"]
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
            if past is not None:
                k = torch.cat([past[0], k], 1)
                v = torch.cat([past[1], v], 1)
            presents.append((k, v))
            a = self._attend(q, k, v, mask)
            x = x + (a @ L["out_w"].t() + L["out_b"])
            h = F.layer_norm(x, (C,), L["n2_w"], L["n2_b"], 1e-5)
            h = F.gelu(h @ L["l1_w"].t() + L["l1_b"])          # exact erf GELU (nn.GELU default)
            x = x + (h @ L["l2_w"].t() + L["l2_b"])
        logits = x @ self.head_w.t() + self.head_b
        return logits, presents, valid

    # ---------------------------------------------------------------- sampler pieces
    @staticmethod
    def masked_probs(logits: torch.Tensor, temperature: float = 1.0, top_k: Optional[int] = 50,
                     top_p: Optional[float] = None) -> torch.Tensor:
        """api_cache.py:169-177: logits/temperature, additive -1e10 outside the top-k, softmax.
        `top_p` (nucleus) is NOT in the reference: build-defined, parity unpinned --
        keep the smallest prefix of the descending-sorted distribution whose mass reaches top_p
        (always at least one token), renormalise."""
        lg = logits / temperature
        if top_k is not None:
            k = min(int(top_k), lg.shape[-1])
            _, idxs = lg.topk(k)
            mask = torch.full_like(lg, -1e10)
            mask.scatter_(-1, idxs, 0.0)
            lg = lg + mask
        probs = torch.softmax(lg, dim=-1)
        if top_p is not None:
            sp, si = probs.sort(dim=-1, descending=True)
            cum = sp.cumsum(-1)
            keep = (cum - sp) < top_p
            sp = sp * keep
            probs = torch.zeros_like(probs).scatter_(-1, si, sp)
            probs = probs / probs.sum(-1, keepdim=True)
        return probs

    @torch.no_grad()
    def generate_greedy(self, prompts: Sequence[Sequence[int]], n_steps: int,
                        return_logits: bool = False):
        """Batched restatement of sample_kvcache(..., temperature=1, top_k=1) (api_cache.py:159-184).
        Rows may have different prompt lengths; each row's result equals the reference run on
        that prompt alone.  Returns list of id lists (prompt + n_steps ids) [and step logits]."""
        B = len(prompts)
        lens = [len(p) for p in prompts]
        Tp = max(lens)
        idx = torch.zeros(B, Tp, dtype=torch.long)
        valid = torch.zeros(B, Tp, dtype=torch.bool)
        for b, p in enumerate(prompts):
            idx[b, :len(p)] = torch.tensor(list(p))
            valid[b, :len(p)] = True
        _, cache, cvalid = self.forward(idx, None, None, valid)     # prefill, logits dropped (:163)
        last = torch.tensor([p[-1] for p in prompts]).view(B, 1)     # re-feed last prompt token (:167)
        outs = [list(p) for p in prompts]
        step_logits = []
        for _ in range(n_steps):
            logits, cache, cvalid = self.forward(last, cache, cvalid, None)
            lg = logits[:, -1, :]
            if return_logits:
                step_logits.append(lg.clone())
            last = lg.argmax(-1, keepdim=True)
            for b in range(B):
                outs[b].append(int(last[b, 0]))
        if return_logits:
            return outs, torch.stack(step_logits, 1)
        return outs

    # ---------------------------------------------------------------- no-KV twin (generate.py)
    @torch.no_grad()
    def forward_twin(self, idx: torch.Tensor) -> torch.Tensor:
        """generate_music/generate.py:25-35: nn.TransformerEncoder defaults = post-LN, ReLU,
        no mask; `pos` there has seq_len-1 rows but is added as pos[:T] all the same."""
        B, T = idx.shape
        C = self.C
        x = self.tok_emb[idx] + self.pos_emb[:T]
        for L in self.layers:
            qkv = x @ L["in_w"].t() + L["in_b"]
            a = self._attend(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], None)
            x = F.layer_norm(x + (a @ L["out_w"].t() + L["out_b"]), (C,), L["n1_w"], L["n1_b"], 1e-5)
            h = torch.relu(x @ L["l1_w"].t() + L["l1_b"])
            x = F.layer_norm(x + (h @ L["l2_w"].t() + L["l2_b"]), (C,), L["n2_w"], L["n2_b"], 1e-5)
        return x @ self.head_w.t() + self.head_b

    @torch.no_grad()
    def generate_greedy_twin(self, prompt: Sequence[int], n_steps: int) -> List[int]:
        ids = torch.tensor(list(prompt)).view(1, -1)
        for _ in range(n_steps):
            nxt = self.forward_twin(ids)[:, -1, :].argmax(-1, keepdim=True)
            ids = torch.cat([ids, nxt], 1)
        return ids[0].tolist()
