"""Drop-in for the decoder half of the reference: the names `api_cache.py` defines for itself
(`GPTWithKV`, `remap_state_dict`, `sample_kvcache`, `encode`, `decode`, `closest_bpm_token`,
`normalize_key_signature`, `FAMILY_TO_INSTRUMENTS`, `note_re`; api_cache.py:76-184) and the ones
`generate_music/generate.py` defines (`GPT`, `sample`; generate.py:25-61), plus the north-star name
`generate_sequence`.  Every tensor operation runs in libmgea_hip.so on the MI355X; importing this
module has no side effects (the reference's files load a checkpoint at import time).

Replacing api_cache.py:39-138,159-184 by

    from generate_music.generate import *          # GPTWithKV, remap_state_dict, sample_kvcache, ...
    model, tok2id, id2tok, SEQ_LEN, D_MODEL = load_checkpoint(CKPT)

leaves its endpoint (api_cache.py:186-243) untouched.

Note on the small host helpers: `FAMILY_TO_INSTRUMENTS`, `note_re`, `encode`, `decode`, `closest_bpm_token` and
`normalize_key_signature` below are the drop-in's constant table, one regex and four one-to-seven-line functions whose exact
behaviour (keys, error classes, string formats) IS the contract with api_cache.py:140-157 -- they are restated line for line
from there on purpose (each cites its lines); everything that computes (the model, the sampler loop) is this repo's own design.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence

import torch

from mgea.decoder import DecoderEngine, geometry_from_state_dict, remap_state_dict  # noqa: F401

__all__ = ["GPTWithKV", "GPT", "remap_state_dict", "load_checkpoint", "set_vocab", "encode", "decode",
           "closest_bpm_token", "normalize_key_signature", "FAMILY_TO_INSTRUMENTS", "note_re", "sample_kvcache",
           "generate_sequence", "sample", "tok2id", "id2tok"]

# module globals like the reference's (api_cache.py:34-35); filled by load_checkpoint / set_vocab
tok2id: Dict[str, int] = {}
id2tok: Dict[int, str] = {}
model = None  # generate.py's `sample(prompt, ...)` uses a module-level model

FAMILY_TO_INSTRUMENTS = {          # api_cache.py:152-156
    "Strings": ["Violin"],
    "Piano": ["Acoustic Grand Piano"],
    "Woodwind": ["Flute"],
}
note_re = re.compile(r"\[NOTE\] \[PITCH:(.+?)\] \[START:(.+?)\] \[END:(.+?)\] \[DURATION:(.+?)\]")  # api_cache.py:157

_DEFAULT_DEVICE = "cuda:0"


def set_vocab(vocab: Dict[str, int]) -> None:
    """Install the token<->id maps (api_cache.py:34-35)."""
    tok2id.clear()
    tok2id.update(vocab)
    id2tok.clear()
    id2tok.update({i: t for t, i in vocab.items()})


def encode(tokens):            # api_cache.py:140
    return torch.tensor([tok2id[t] for t in tokens])


def decode(ids):               # api_cache.py:141
    return [id2tok[int(i)] for i in ids]


def closest_bpm_token(val):    # api_cache.py:142-144 (ValueError from min() when no [BPM] token exists)
    bpm_toks = [t for t in tok2id if t.startswith("[BPM]")]
    return min(bpm_toks, key=lambda s: abs(float(s.split()[-1]) - val))


def normalize_key_signature(key_string):   # api_cache.py:145-151
    key_string = key_string.replace("♭", "-").replace("♯", "#")
    parts = key_string.strip().split()
    if len(parts) == 2:
        key, scale = parts
        return f"[KEY_SIGNATURE] {key} {scale.lower()}"
    return f"[KEY_SIGNATURE] {key_string}"


class _KVState(list):
    """What `model(idx, past)` returns as `presents`: the cache itself lives in the native engine's
    KV pages; this token only proves the caller continues the most recent sequence.  It is a list of
    n_layer entries so code that zips it with the layers (api_cache.py:101) still works."""

    def __init__(self, n_layer, epoch, length):
        super().__init__([None] * n_layer)
        self.epoch, self.length = epoch, length


class GPTWithKV:
    """Same constructor / call surface as the reference class (api_cache.py:76-106), backed by a
    native decoder handle.  Weights arrive through load_state_dict (names after remap_state_dict)."""

    block_mode = "kv"

    def __init__(self, vocab_size, seq_len, d_model, n_head, n_layer, max_batch: int = 8,
                 max_ctx: Optional[int] = None, device: str = _DEFAULT_DEVICE, dtype: str = "f32"):
        self.vocab_size, self.seq_len, self.d_model = int(vocab_size), int(seq_len), int(d_model)
        self.n_head, self.n_layer = int(n_head), int(n_layer)
        self.max_batch, self.max_ctx, self.device = int(max_batch), max_ctx, device
        self.dtype = dtype   # "f32" = the reference's arithmetic (parity mode); "f16" = fp16 matrices + KV (mgea.decoder)
        self.engine: Optional[DecoderEngine] = None
        self._epoch = -1

    # -- nn.Module look-alikes ----------------------------------------------------------------
    def load_state_dict(self, sd: Dict, strict: bool = True):
        sd = remap_state_dict(sd)
        geo = geometry_from_state_dict(sd)
        want = dict(vocab=self.vocab_size, seq_len=self.seq_len, d_model=self.d_model, n_layer=self.n_layer)
        for k, v in want.items():
            if geo[k] != v:   # what nn.Module.load_state_dict reports as a size mismatch
                raise RuntimeError(f"Error(s) in loading state_dict: size mismatch for {k}: checkpoint {geo[k]}, model {v}")
        if self.engine is not None:
            self.engine.close()
        max_ctx = self.max_ctx if self.max_ctx is not None else max(self.seq_len, 1)
        self.engine = DecoderEngine(sd, n_head=self.n_head, max_batch=self.max_batch, max_ctx=max_ctx,
                                    device=self.device, block_mode=self.block_mode, dtype=self.dtype)
        return "<All keys matched successfully>"

    def eval(self):
        return self

    def to(self, device=None, *a, **k):
        return self  # the model lives on its MI355X; `device="cpu"` of the reference call is accepted and ignored

    def _need(self) -> DecoderEngine:
        if self.engine is None:
            raise RuntimeError("GPTWithKV has no weights: call load_state_dict first")
        return self.engine

    def __call__(self, idx: torch.Tensor, past_kv=None):
        """logits [B,T,V] (fp32, on the GPU) and an opaque `presents` to pass back as past_kv."""
        eng = self._need()
        B, T = idx.shape
        fresh = past_kv is None or (not isinstance(past_kv, _KVState) and all(p is None for p in past_kv))
        if fresh:   # api_cache.py:96-97: past_kv=None means an empty cache
            eng.reset(B)
            self._epoch = eng._epoch
        else:
            if not isinstance(past_kv, _KVState) or past_kv.epoch != self._epoch or past_kv.length != eng._len:
                raise RuntimeError("past_kv is not the most recent `presents` of this model "
                                   "(the KV cache lives in the native engine and only grows)")
        logits = eng.forward(idx, None, want_logits=True)
        return logits, _KVState(self.n_layer, self._epoch, eng._len)

    forward = __call__


class GPT(GPTWithKV):
    """The no-cache twin of generate_music/generate.py:25-35 (post-LN, ReLU nn.TransformerEncoder,
    full recompute, no mask): `model(x)` returns logits only.  Takes the training checkpoint as is."""

    block_mode = "twin"

    def __init__(self, vocab, seq_len, d_model, n_head=4, n_layer=2, **kw):
        # generate.py stores seq_len-1 position rows (generate.py:18,29)
        super().__init__(vocab, seq_len - 1, d_model, n_head, n_layer, **kw)

    def __call__(self, x: torch.Tensor):
        eng = self._need()
        return eng.reset_and_prefill(x, None, want_logits=True)

    forward = __call__


def load_checkpoint(path, n_head: int = 8, device: str = _DEFAULT_DEVICE, max_batch: int = 8,
                    max_ctx: Optional[int] = None):
    """api_cache.py:26-37,108-138 in one call: torch.load(weights_only=True), geometry from tensor
    shapes, GPTWithKV + remapped weights.  Returns (model, tok2id, id2tok, SEQ_LEN, D_MODEL) and
    installs the vocabulary in this module."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    geo = geometry_from_state_dict(ckpt["model"])
    set_vocab(ckpt["vocab"])
    m = GPTWithKV(vocab_size=len(tok2id), seq_len=geo["seq_len"], d_model=geo["d_model"], n_head=n_head,
                  n_layer=geo["n_layer"], max_batch=max_batch, max_ctx=max_ctx, device=device)
    m.load_state_dict(remap_state_dict(ckpt["model"]))
    global model
    model = m
    return m, tok2id, id2tok, geo["seq_len"], geo["d_model"]


def _as_model(model_or_weights, n_head=8, device=_DEFAULT_DEVICE) -> GPTWithKV:
    if isinstance(model_or_weights, GPTWithKV):
        return model_or_weights
    if isinstance(model_or_weights, dict):
        sd = model_or_weights.get("model", model_or_weights)
        if "vocab" in model_or_weights and not tok2id:
            set_vocab(model_or_weights["vocab"])
        geo = geometry_from_state_dict(sd)
        m = GPTWithKV(geo["vocab"], geo["seq_len"], geo["d_model"], n_head, geo["n_layer"], device=device)
        m.load_state_dict(sd)
        return m
    raise TypeError("expected a GPTWithKV or a state dict / checkpoint dict")


def _draw_seed() -> int:
    # torch.manual_seed(s) therefore makes a sampled generation reproducible, like the reference
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def sample_kvcache(model, prompt: Sequence[str], max_len=512, temperature=1.0, top_k=50, device="cpu",
                   top_p: Optional[float] = None, seed: Optional[int] = None) -> List[str]:
    """api_cache.py:159-184 with the same signature: prefill the prompt (logits dropped), then up to
    max_len - len(prompt) steps of /temperature, top-k mask, softmax, one multinomial draw; stops
    after [END_SEQUENCE].  Runs as one native generate() call (prefill + hipGraph-replayed decode
    steps).  `device` is accepted for compatibility; the work happens on the model's MI355X.
    top_k=1 is exactly greedy; other settings match torch.multinomial in distribution only."""
    m = _as_model(model)
    eng = m._need()
    ids = [tok2id[t] for t in prompt]          # KeyError for an unknown token, like api_cache.py:162
    n_steps = int(max_len) - len(ids)
    if n_steps <= 0:
        return [id2tok[i] for i in ids]
    if len(ids) + n_steps > eng.max_ctx:
        raise RuntimeError(f"max_len={max_len} exceeds the engine's reserved context {eng.max_ctx}")
    eos = tok2id.get("[END_SEQUENCE]", -1)
    out = eng.generate([ids], n_steps, temperature=temperature, top_k=top_k, top_p=top_p, eos_id=eos,
                       seed=_draw_seed() if seed is None else seed)
    gen = [int(i) for i in out[0].cpu().tolist() if i >= 0]
    return [id2tok[i] for i in ids + gen]


def generate_sequence(model_or_weights, prompt: Sequence[str], max_len=512, temperature=1.0, top_k=50,
                      device=_DEFAULT_DEVICE, top_p: Optional[float] = None, seed: Optional[int] = None,
                      n_head: int = 8) -> List[str]:
    """North-star name (BASELINE.json): sample_kvcache on a model object or a weights/checkpoint dict."""
    return sample_kvcache(_as_model(model_or_weights, n_head, device), prompt, max_len, temperature, top_k,
                          device, top_p, seed)


def generate_batch(model, prompts: Sequence[Sequence[str]], max_len=512, temperature=1.0, top_k=50,
                   top_p: Optional[float] = None, seed: Optional[int] = None) -> List[List[str]]:
    """New surface: many prompts in one batch (ragged lengths allowed); every row equals the
    reference run on that prompt alone (greedy) -- rows are independent."""
    m = _as_model(model)
    eng = m._need()
    ids = [[tok2id[t] for t in p] for p in prompts]
    n_steps = int(max_len) - max(len(p) for p in ids)
    eos = tok2id.get("[END_SEQUENCE]", -1)
    out = eng.generate(ids, max(n_steps, 0), temperature=temperature, top_k=top_k, top_p=top_p, eos_id=eos,
                       seed=_draw_seed() if seed is None else seed).cpu().tolist()
    return [[id2tok[i] for i in p + [g for g in row if g >= 0]] for p, row in zip(ids, out)]


def sample(prompt: Sequence[str], max_len=512, temperature=1.0, top_k=50, device="cpu") -> List[str]:
    """generate_music/generate.py:46-61: same sampler over the module-level `model`.  With a
    GPTWithKV model this is sample_kvcache; with the post-LN twin `GPT` it recomputes the whole
    sequence every step exactly like the reference script (no cache is valid for that network)."""
    if model is None:
        raise RuntimeError("no module-level model: call load_checkpoint() or set generate.model")
    if not isinstance(model, GPT):
        return sample_kvcache(model, prompt, max_len, temperature, top_k, device)
    from mgea import ops
    ids = encode(prompt).unsqueeze(0)
    eos = tok2id.get("[END_SEQUENCE]", -1)
    for step in range(max_len - len(prompt)):
        logits = model(ids)[:, -1, :]
        nxt = ops.sample(logits, temperature, top_k, None, seed=_draw_seed(), step=step).cpu().long().view(1, 1)
        ids = torch.cat([ids, nxt], dim=1)
        if int(nxt) == eos:
            break
    return decode(ids.squeeze(0))
