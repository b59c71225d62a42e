#!/usr/bin/env python3
"""Benchmark of the MI355X hot path on BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU, RCCL weight broadcast over xGMI.  Under a launcher (torch.distributed.run sets RANK /
  WORLD_SIZE) this process IS a rank; without one it starts its own N ranks as a child job before touching the GPU
  (mgea/launch.py) and exits with that job's code.  WORLD_SIZE != N is an error, never a silent smaller run.

A "step" = one pass of the hot path over one batch of synthetic input = one greedy generation of
configs[2]: Decoder-S (6L / 512d / 8H, V = 8324, random weights), batch 64 per GPU, 5-token
prompts decoded to a total length of 1024 (1019 decode steps, sample_kvcache semantics incl. the
re-fed last prompt token).  value = generated MIDI tokens / s over all ranks, inputs and weights
resident in HBM, timed with barrier + torch.cuda.synchronize() on both sides, max over ranks.
Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` for the dominant
kernel (paged decode attention; HIP events on the launch stream) and `cpu_baseline` (the oracle =
CPU port of the reference semantics, timed on this host's cores).  Under "extra": DistilBERT prompts/s (the
other half of BASELINE.json's metric; f32 parity mode and bf16), the [64, 1024] decoder prefill against the MFMA
peak, BASELINE configs[4] (12L / 768d, top-p 0.9, 2048 tokens, fp16 storage) on one GPU, the headline generation
in fp16 storage and at B = 256.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "music-generation-emotion-adaptive_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
PMC_FILES = ("r4_pmc_fetch_size_attn.json", "r3_pmc_fetch_size_attn.json", "r2_pmc_fetch_size_full.json", "r1_v6_pmc_fetch_size_full.json")   # newest committed FETCH_SIZE pass first

DEC = dict(vocab=8324, seq_len=1024, d_model=512, n_layer=6, d_ff=2048)   # train/train_large2.py:10-12,23-28
N_HEAD = 8                                                                 # api_cache.py:112
BERT = dict(vocab=30522, max_pos=512, dim=768, n_layers=6, hidden=3072, num_labels=28)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="prompts per GPU")
    ap.add_argument("--prompt-len", type=int, default=5)
    ap.add_argument("--total-len", type=int, default=1024)
    ap.add_argument("--profile-stride", type=int, default=16, help="HIP-event profile every n-th decode step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-bert", action="store_true", help="skip the DistilBERT extra")
    ap.add_argument("--no-extra", action="store_true", help="skip every extra (prefill, Decoder-L, fp16, B=256, DistilBERT)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dry-run", action="store_true",
                    help="launch path only: ranks rendezvous, broadcast a small arena and print the line; no GPU work (CPU test)")
    return ap.parse_args()


def host_threads():
    """Threads for the CPU baseline: the cores this process may run on, capped at the GPU box's
    per-GPU CPU share (16) -- oversubscribing a 256-core host from a 16-core share is slower."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("MGEA_CPU_THREADS", "16"))))


def cpu_baseline_decoder(sd, prompts, budget_s):
    """Oracle (CPU port of api_cache.py semantics with a projected-KV cache) on the same prompts, timed on this host's cores.
    Bounded sample in TWO context windows, because a decode step gets slower as the cache grows and the GPU figure is an average
    over contexts 6..1024: (a) the real start of the generation (prompt 5, decode steps for ~60 % of the budget), (b) the middle
    (a 507-token prefix = the prompts followed by synthetic ids, prefilled untimed, then decode steps for the rest).  `value` is
    tokens / s over the timed decode steps of both windows together."""
    from mgea import synth
    from oracle.decoder_ref import DecoderRef
    torch.set_num_threads(host_threads())
    ref = DecoderRef(sd, n_head=N_HEAD)
    idx = torch.tensor(prompts)
    B = idx.shape[0]

    def window(prefix, seconds, max_steps):
        _, cache, valid = ref.forward(prefix)
        last = prefix[:, -1:]
        n, t0 = 0, time.perf_counter()
        while True:
            logits, cache, valid = ref.forward(last, cache, valid)
            last = logits[:, -1, :].argmax(-1, keepdim=True)
            n += 1
            if time.perf_counter() - t0 > seconds or n >= max_steps:
                break
        return n, time.perf_counter() - t0

    n_a, t_a = window(idx, 0.6 * budget_s, 400)
    mid = torch.cat([idx, torch.from_numpy(synth.integers(3, "cpu_mid", (B, 502), 0, DEC["vocab"]))], 1)
    n_b, t_b = window(mid, 0.4 * budget_s, 200)
    return dict(value=B * (n_a + n_b) / (t_a + t_b), unit="tokens/s", cores=torch.get_num_threads(), kind="port",
                windows=[dict(ctx_from=idx.shape[1] + 1, ctx_to=idx.shape[1] + n_a, tokens_per_sec=B * n_a / t_a),
                         dict(ctx_from=mid.shape[1] + 1, ctx_to=mid.shape[1] + n_b, tokens_per_sec=B * n_b / t_b)],
                sample=f"oracle/decoder_ref.py (torch CPU fp32), B={B}: {n_a} decode steps from the 5-token prompts (ctx {idx.shape[1] + 1}.."
                       f"{idx.shape[1] + n_a}, {t_a:.1f} s) + {n_b} decode steps behind a 507-token prefix (ctx {mid.shape[1] + 1}.."
                       f"{mid.shape[1] + n_b}, {t_b:.1f} s; its prefill untimed)")


def bert_extra(device, steps, warmup, with_cpu, dtype="f32", sd=None, ad=None, ref_logits=None, keep_logits=False):
    from mgea import synth
    from mgea.bert import BertEngine
    B, S = 256, 128
    if sd is None:
        sd = synth.distilbert_state_dict(41, BERT["vocab"], BERT["max_pos"], BERT["dim"], BERT["n_layers"], BERT["hidden"])
        ad = synth.lora_adapter(41, BERT["dim"], BERT["n_layers"])
    eng = BertEngine(sd, n_heads=12, adapter=ad, max_tokens=B * S, device=device, dtype=dtype)
    ids, mask = synth.bert_inputs(2, B, S, BERT["vocab"])
    ids, mask = torch.from_numpy(ids).to(device), torch.from_numpy(mask).to(device)
    check = None
    if ref_logits is not None:
        # the timed configuration is checked before it is timed: all 256 rows against the f32 parity engine's logits on the same ids
        lg, am = eng.forward(ids, mask)
        d = (lg - ref_logits).abs().max(1).values
        srt = ref_logits.sort(1).values
        decided = (srt[:, -1] - srt[:, -2]) > 0.10
        agree = bool((am[decided].long() == ref_logits.argmax(1)[decided]).all())
        check = dict(vs="f32 engine, same ids, 256 rows", max_abs_logit_diff=float(d.max()), mean_row_max=float(d.mean()), tolerance=0.05,
                     rows_with_top2_gap_over_0p10=int(decided.sum()), labels_equal_on_those=agree, kernels=eng.stats())
        assert float(d.max()) < 0.05 and agree, f"bf16 DistilBERT logits off the f32 engine: {check}"
    from mgea import _lib

    def timed(full_last):
        old = _lib.tune_set("bert_full_last_layer", full_last)
        try:
            for _ in range(max(1, warmup)):
                eng.forward(ids, mask)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.forward(ids, mask)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps, eng.stats()["last_layer_cls_only"]
        finally:
            _lib.tune_set("bert_full_last_layer", old)
    D, FF, L = BERT["dim"], BERT["hidden"], BERT["n_layers"]
    peak = 2500.0 if dtype == "bf16" else 157.3
    # every position of every layer (what the reference computes: transformers DistilBERT, then hidden_state[:, 0])
    flops_all = 2 * B * S * L * (4 * D * D + 2 * D * FF) + 4 * B * S * S * D * L + 2 * B * (D * D + 28 * D)
    # the engine's default: of the LAST layer only K | V for every position; its query, attention, out-projection and FFN for the B
    # [CLS] rows the classifier reads.  FLOPs actually executed:
    flops_cls = flops_all - 2 * B * (S - 1) * (2 * D * D + 2 * D * FF) - 4 * B * (S - 1) * S * D
    dt_all, was_cls = timed(1)
    assert not was_cls
    dt, was_cls = timed(0)
    assert was_cls
    out = dict(metric="distilbert_prompts_per_sec", value=B / dt, unit="prompts/s", ms_per_batch=dt * 1e3,
               dtype=dtype, workload="DistilBERT-base(+LoRA merged) classifier B=256 S=128 padded rows, random weights; last layer: K | V of every "
                                     "position, query / attention / out-proj / FFN for the 256 [CLS] rows the classifier reads (same logits)",
               roofline=dict(bound="mfma", achieved=flops_cls / dt / 1e12, peak=peak, unit="TFLOP/s", frac=flops_cls / dt / 1e12 / peak, traffic=None,
                             flops_executed=flops_cls, flops_every_position=flops_all,
                             note="FLOPs actually EXECUTED / wall time vs the dense MFMA peak of the dtype"),
               every_position=dict(value=B / dt_all, unit="prompts/s", ms_per_batch=dt_all * 1e3,
                                   roofline=dict(bound="mfma", achieved=flops_all / dt_all / 1e12, peak=peak, unit="TFLOP/s",
                                                 frac=flops_all / dt_all / 1e12 / peak, traffic=None),
                                   note="switch bert_full_last_layer = 1: every position of the last layer computed and all but one per sequence "
                                        "discarded, as the reference does; the figure of rounds 1-2"))
    if True:
        # PACKED rows (round 4): the real tokens of the batch back to back -- the tokenizer pads to the longest prompt (inference.py:16), 44 % of
        # this batch is padding.  Reported BESIDE the padded figures (the padded every-position number stays the roofline evidence); FLOPs counted
        # are those executed on the real tokens.  The packed ids / positions / offsets are the input format (resident before the clock starts),
        # as the padded ids + mask are for the other legs; logits checked against the padded forward before timing.
        pk = BertEngine.pack(ids.cpu(), mask.cpu())
        pk_dev = tuple(t.to(device) if isinstance(t, torch.Tensor) else t for t in pk)
        lg_pad = eng.forward(ids, mask)[0]
        lg_pk = eng.forward_packed(*pk_dev)[0]
        rows = eng.stats()["rows"]
        d_pk = float((lg_pk - lg_pad).abs().max())
        assert rows == int(mask.sum()) and d_pk < (5e-3 if dtype == "bf16" else 1e-4), f"packed DistilBERT forward differs from the padded one by {d_pk}"
        for _ in range(max(1, warmup)):
            eng.forward_packed(*pk_dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.forward_packed(*pk_dev)
        torch.cuda.synchronize()
        dt_p = (time.perf_counter() - t0) / steps
        lens = mask.sum(1).double().cpu()
        n_real = float(lens.sum())
        flops_pk = (2 * n_real * (L - 1) * (4 * D * D + 2 * D * FF) + 2 * n_real * 2 * D * D + 4 * float((lens * lens).sum()) * D * (L - 1)
                    + 2 * B * (2 * D * D + 2 * D * FF) + 4 * n_real * D + 2 * B * (D * D + 28 * D))
        out["packed"] = dict(value=B / dt_p, unit="prompts/s", ms_per_batch=dt_p * 1e3, rows=rows, rows_padded=B * S,
                             max_abs_logit_diff_vs_padded=d_pk,
                             roofline=dict(bound="mfma", achieved=flops_pk / dt_p / 1e12, peak=peak, unit="TFLOP/s", frac=flops_pk / dt_p / 1e12 / peak,
                                           traffic=None, flops_executed=flops_pk),
                             note="mgea_bert_forward_packed: every GEMM, LayerNorm statistic and attention tile on the real tokens only (same logits "
                                  "as the padded call); an algorithmic cut, reported beside the padded figures, not instead of them")
    if with_cpu:
        from oracle.distilbert_ref import DistilBertRef
        ref = DistilBertRef(sd, 12, ad)
        n = 32
        t0 = time.perf_counter()
        ref.forward(ids[:n].cpu().long(), mask[:n].cpu().long())
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=n / dtc, unit="prompts/s", cores=torch.get_num_threads(), kind="port",
                                   sample=f"oracle/distilbert_ref.py, first {n} rows of the batch, one forward, {dtc:.1f} s")
    # the reference's own serving case for this model: ONE text per request (api_cache.py:189 -> inference.predict, B = 1), here a
    # 32-token prompt through the same engine (forward + device argmax, ids already on the device)
    i1, m1 = ids[:1, :32].contiguous(), torch.ones(1, 32, dtype=mask.dtype, device=device)
    for _ in range(5):
        eng.forward(i1, m1)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        eng.forward(i1, m1)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out["single_prompt"] = dict(batch=1, tokens=32, ms_per_request=sorted(ts)[len(ts) // 2] * 1e3,
                                note="one 32-token prompt per request, median of 30 (api_cache.py:189: the endpoint classifies one text)")
    if check is not None:
        out["parity_check"] = check
    if keep_logits:
        out["_logits"] = eng.forward(ids, mask)[0].clone()
    eng.close()
    return out


def b1_extra(arena, device, TL=1024, Tp=5):
    """The reference's actual serving case (api_cache.py:204: B = 1, top_k = 50, a fresh seed per request) and the only
    figure the reference publishes for this path (0.29 ms per token with its KV cache on an RTX A4000, 6L / 512d,
    seq 512): one warm-up request, then two timed 1019-step requests with new seeds through the cached step graph."""
    from mgea.decoder import DecoderEngine
    eng = DecoderEngine(None, n_head=N_HEAD, max_batch=1, max_ctx=TL, device=device, geometry=DEC, arena=arena)
    p = [[1, 2, 3, 4, 5][:Tp]]
    n = TL - Tp
    eng.generate(p, n, temperature=1.0, top_k=50, seed=1)
    torch.cuda.synchronize()
    i0 = eng.stats()["graph_instantiates"]
    ts = []
    for seed in (101, 102):
        t0 = time.perf_counter()
        out = eng.generate(p, n, temperature=1.0, top_k=50, seed=seed).cpu()       # incl. reset, prefill and the D2H copy of the ids
        ts.append(time.perf_counter() - t0)
    assert int(out.min()) >= 0 and int(out.max()) < DEC["vocab"]
    st = eng.stats()                                   # before the greedy leg below, which captures its own (greedy) step graph
    eng.generate(p, n, temperature=1.0, top_k=1).cpu()
    t0 = time.perf_counter()
    eng.generate(p, n, temperature=1.0, top_k=1).cpu()
    tg = time.perf_counter() - t0
    eng.close()
    dt = min(ts)
    return {"metric": "decoder_latency_per_token", "value": dt / n * 1e3, "unit": "ms/token", "higher_is_better": False, "batch": 1,
            "tokens_per_sec": n / dt, "us_per_step": dt / n * 1e6, "ms_per_request": [round(t * 1e3, 2) for t in ts],
            "us_per_step_greedy": tg / n * 1e6, "sampler": {"temperature": 1.0, "top_k": 50},
            "graph_instantiations_during_timed_requests": st["graph_instantiates"] - i0, "graph_nodes": st["graph_nodes"],
            "published_reference": {"value": 0.29, "unit": "ms/token", "hardware": "RTX A4000", "source": "BASELINE.md (paper, KV cache, B=1)"},
            "vs_published": 0.29 / (dt / n * 1e3),
            "workload": f"Decoder-S B=1 prompt {Tp} total_len {TL}, top_k=50 multinomial, fresh seed per request (api_cache.py:204), random weights"}


def large_batch_extra(arena, device, B, Tp, TL):
    from mgea import synth
    from mgea.decoder import DecoderEngine
    eng = DecoderEngine(None, n_head=N_HEAD, max_batch=B, max_ctx=TL, device=device, geometry=DEC, arena=arena)
    prompts = torch.from_numpy(synth.integers(9, "prompts", (B, Tp), 0, DEC["vocab"])).to(device=device, dtype=torch.int32)
    eng.generate(prompts, TL - Tp, temperature=1.0, top_k=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.generate(prompts, TL - Tp, temperature=1.0, top_k=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nodes = eng.stats()["graph_nodes"]
    eng.close()
    C_, NL, V = DEC["d_model"], DEC["n_layer"], DEC["vocab"]
    p_step = NL * (12 * C_ * C_ + 13 * C_) + V * C_ + V
    step_bytes = sum(p_step * 4 + B * NL * 2 * C_ * 4 * (Tp + i + 1 + 1) for i in range(TL - Tp))   # as the headline's whole_step_hbm_frac
    return {"metric": "midi_tokens_per_sec", "value": B * (TL - Tp) / dt, "unit": "tokens/s", "batch": B, "ms_per_generation": dt * 1e3,
            "whole_step_hbm_frac": step_bytes / dt / 1e9 / HBM_PEAK_GBS,
            "graph_nodes": nodes, "note": "same model, prompts of the same shape, greedy; not the BASELINE configuration (B = 64)"}


def whole_step_bytes(geo, B, Tp, n_steps, e_w, e_kv):
    """per step the projection / head parameters once (e_w bytes each) + every cached K/V element read once
    and the new token's K/V written (e_kv bytes each); ctx includes the duplicated last prompt token."""
    C_, NL, V = geo["d_model"], geo["n_layer"], geo["vocab"]
    p_step = NL * (12 * C_ * C_ + 13 * C_) + V * C_ + V
    return sum(p_step * e_w + B * NL * 2 * C_ * e_kv * (Tp + i + 1 + 1) for i in range(n_steps))


def prefill_extra(arena, device, B=64, T=1024, reps=5, dtype="f32"):
    """north_star: ">= 50 % MFMA-roofline on prefill".  Decoder-S [64, 1024] non-causal prefill with the logits of every
    position (api_cache.py:87-106 returns them; SURVEY §8d: 3.86 TFLOP = dense 3.03 + attention 0.82)."""
    from mgea import synth
    from mgea.decoder import DecoderEngine
    eng = DecoderEngine(None, n_head=N_HEAD, max_batch=B, max_ctx=T, device=device, geometry=DEC, arena=arena, dtype=dtype)
    ids = torch.from_numpy(synth.integers(1, "prefill", (B, T), 0, DEC["vocab"])).to(device=device, dtype=torch.int32)
    peak = 157.3 if dtype == "f32" else 2500.0
    out = {}
    C_, NL, V = DEC["d_model"], DEC["n_layer"], DEC["vocab"]
    for name, want_logits in (("with_logits", True), ("cache_fill_only", False)):
        for _ in range(2):              # the GPU has just idled through the CPU baseline: let clocks and allocator settle
            lg = eng.reset_and_prefill(ids, want_logits=want_logits)
        torch.cuda.synchronize()
        times = []
        for _ in range(reps):           # each repetition timed on its own; the median is robust against a one-off allocator stall
            del lg
            t0 = time.perf_counter()
            lg = eng.reset_and_prefill(ids, want_logits=want_logits)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        del lg
        dt = sorted(times)[len(times) // 2]
        flops = 2 * B * T * (NL * 12 * C_ * C_ + (V * C_ if want_logits else 0)) + 4 * B * T * T * C_ * NL
        if not want_logits:
            # a forward whose logits are dropped (the prompt prefill of sample_kvcache, api_cache.py:163) ends once the last block's
            # K | V are in the cache: its query projection, attention, out-projection and MLP (10 of its 12 C^2 and its T^2 term)
            # are not executed, and are not counted
            flops -= 2 * B * T * (10 if dtype == "f16" else 9) * C_ * C_ + 4 * B * T * T * C_   # (the exact-fp32 path still projects that block's query)
        out[name] = dict(ms=dt * 1e3, tokens_per_sec=B * T / dt, tflops=flops / dt / 1e12, algorithmic_tflop=flops / 1e12,
                         frac_of_mfma_peak=flops / dt / 1e12 / peak, ms_each=[round(t * 1e3, 2) for t in times])
        if not want_logits:
            out[name]["note"] = "logits dropped: the last block stops at its K | V (FLOPs executed; switch decoder_prefill_full = 1 runs the whole block)"
    st = eng.stats()
    eng.close()
    w = out["with_logits"]
    note = ("exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): the f32 matrix peak is 157.3 TFLOP/s, 1/16 of the bf16 one" if dtype == "f32" else
            "fp16 perf mode: v_mfma_f32_16x16x32_f16 GEMMs (persistent 256 x 256 kernel, LayerNorm folded), fp16 flash attention, fp32-output "
            "head; against the dense f16 MFMA peak")
    return dict(metric="decoder_prefill_tokens_per_sec", value=w["tokens_per_sec"], unit="tokens/s", ms=w["ms"], dtype=dtype,
                workload=f"Decoder-S non-causal prefill, ids [{B}, {T}], logits for every position, empty cache, random weights",
                roofline=dict(bound="mfma", achieved=w["tflops"], peak=peak, unit="TFLOP/s", frac=w["frac_of_mfma_peak"], traffic=None, note=note),
                cache_fill_only=out["cache_fill_only"], prefill16_forwards=st.get("prefill16_forwards", 0))


DEC_L = dict(vocab=8324, seq_len=2048, d_model=768, n_layer=12, d_ff=3072)   # BASELINE configs[4]; 12 heads x 64


def decoder_gen_extra(device, geo, n_head, B, Tp, TL, dtype, sampler, arena=None, seed=5, note=""):
    """One warm-up + one timed generation of a decoder configuration; whole-step algorithmic bytes vs the HBM peak."""
    from mgea import synth
    from mgea.decoder import DecoderEngine
    sd = None if arena is not None else synth.decoder_state_dict(seed, geo["vocab"], geo["seq_len"], geo["d_model"], geo["n_layer"])
    eng = DecoderEngine(sd, n_head=n_head, max_batch=B, max_ctx=TL, device=device, geometry=geo if arena is not None else None,
                        arena=arena, dtype=dtype)
    prompts = torch.from_numpy(synth.integers(9, "prompts", (B, Tp), 0, geo["vocab"])).to(device=device, dtype=torch.int32)
    eng.generate(prompts, TL - Tp, **sampler)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = eng.generate(prompts, TL - Tp, **sampler)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.stats()
    assert int(out.min()) >= 0 and int(out.max()) < geo["vocab"]
    eng.close()
    e = 2 if dtype == "f16" else 4
    nbytes = whole_step_bytes(geo, B, Tp, TL - Tp, e, e)
    return {"metric": "midi_tokens_per_sec", "value": B * (TL - Tp) / dt, "unit": "tokens/s", "dtype": dtype, "batch": B,
            "ms_per_generation": dt * 1e3, "us_per_step": dt / (TL - Tp) * 1e6,
            "whole_step_hbm_frac": nbytes / dt / 1e9 / HBM_PEAK_GBS, "algorithmic_gb": nbytes / 1e9,
            "graph_nodes": st["graph_nodes"], "sampler": {k: v for k, v in sampler.items() if k != "seed"},
            "workload": f"{geo['n_layer']}L/{geo['d_model']}d/{n_head}H V={geo['vocab']} B={B} prompt {Tp} total_len {TL}, random weights",
            "note": note}


def dry_run(args, backend):
    """The launch path without a GPU: rendezvous, one arena broadcast, barrier, max-reduce, one JSON line."""
    import torch.distributed as dist
    from mgea import dist as mdist
    from mgea import synth
    rank, world, _ = mdist.init_from_env(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        assert dist.get_world_size() == args.gpus
    n = 1 << 16
    arena = torch.from_numpy(synth.uniform(9, "arena", (n,))) if rank == 0 else torch.zeros(n)
    mdist.broadcast_arena(arena, 0)
    ok = bool(np.array_equal(arena.numpy(), synth.uniform(9, "arena", (n,))))
    if world > 1:
        dist.barrier()
    per_rank = mdist.all_gather_floats(1000.0 + rank, "cpu")
    dt = mdist.all_reduce_max(1e-3 * (rank + 1), "cpu")
    assert ok and abs(dt - 1e-3 * world) < 1e-12 and per_rank == [1000.0 + r for r in range(world)]
    # the prompt shards of the real run (BASELINE configs[3]: 64 prompts per GPU, 512 over 8): every rank reports the slice it would take
    gb = 64 * world
    mine = mdist.shard_rows(gb, rank, world)
    firsts = [int(v) for v in mdist.all_gather_floats(float(mine.start), "cpu")]
    counts = [int(v) for v in mdist.all_gather_floats(float(len(mine)), "cpu")]
    assert counts == [64] * world and firsts == [64 * r for r in range(world)]
    if rank == 0:
        print(json.dumps({"metric": "midi_tokens_per_sec", "value": 0.0, "unit": "tokens/s", "n_gpus": world, "steps": 0, "warmup": 0,
                          "dry_run": True, "backend": dist.get_backend() if world > 1 else None, "scaling": "weak",
                          "per_rank_tokens_per_sec": per_rank, "rows_per_rank": counts, "first_row_per_rank": firsts,
                          "config": {"workload": "launch-path rehearsal: no GPU work", "global_batch": gb, "parallelism": parallelism_label(world, backend, n * 4, 0.0)}}))
    if world > 1:
        dist.destroy_process_group()


def parallelism_label(world, backend, nbytes, seconds):
    lib = "RCCL" if backend == "nccl" else backend
    if world == 1:
        return "1 GPU (no collective)"
    return f"dp{world} replicas, one {lib} weight broadcast ({nbytes / 1e6:.1f} MB, {seconds * 1e3:.1f} ms), no collective in decode"


def main():
    args = parse()
    backend = os.environ.get("MGEA_DIST_BACKEND", "nccl")
    from mgea import launch
    if args.gpus > 1 and not launch.launched_by_a_launcher():
        # no launcher: start the N ranks ourselves, BEFORE anything initialises the GPU in this process
        # (devices counted from sysfs: no HIP or torch device query in the parent)
        ndev = launch.count_gpus()
        raise SystemExit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, backend, None if args.dry_run else ndev))
    if args.dry_run:
        return dry_run(args, backend)
    from mgea import dist as mdist
    from mgea import synth
    from mgea.decoder import DecoderEngine, arena_layout
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    # production: "nccl" (= RCCL over xGMI), one rank per GPU.  MGEA_DIST_BACKEND=gloo rehearses the
    # N > 1 flow on a box with fewer GPUs than ranks (ranks then share devices round-robin).
    ndev = torch.cuda.device_count()
    if backend != "nccl":
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev))
    rank, world, local = mdist.init_from_env(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the job must have exactly one rank per GPU")
    if world > 1:
        assert dist.get_world_size() == args.gpus
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    B, Tp, TL = args.batch, args.prompt_len, args.total_len
    n_steps = TL - Tp
    # ---- weights: rank 0 packs the arena, one RCCL broadcast moves it (xGMI), replicas afterwards
    offs, total = arena_layout(DEC, N_HEAD)
    sd = None
    if rank == 0:
        sd = synth.decoder_state_dict(0, DEC["vocab"], DEC["seq_len"], DEC["d_model"], DEC["n_layer"])
        arena = DecoderEngine.pack_arena(sd, DEC, offs, total, device)
    else:
        arena = torch.empty(total, dtype=torch.float32, device=device)
    t_b0 = time.perf_counter()
    mdist.broadcast_arena(arena, 0)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t_b0
    eng = DecoderEngine(None, n_head=N_HEAD, max_batch=B, max_ctx=TL, device=device, geometry=DEC, arena=arena)
    prompts = synth.integers(1 + rank, "prompts", (B, Tp), 0, DEC["vocab"])
    prompts_dev = torch.from_numpy(prompts).to(device=device, dtype=torch.int32)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        eng.generate(prompts_dev, n_steps, temperature=1.0, top_k=1)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = eng.generate(prompts_dev, n_steps, temperature=1.0, top_k=1)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_rank = mdist.all_gather_floats(args.steps * B * n_steps / dt, device)   # each rank's own tokens/s: a straggler shows in the line
    dt = mdist.all_reduce_max(dt, device)
    tokens = world * args.steps * B * n_steps
    value = tokens / dt

    # ---- roofline of the dominant kernel: paged decode attention, HIP events on the launch stream
    roof = None
    if rank == 0 and args.profile_stride > 0:
        eng.profile(args.profile_stride)
        eng.generate(prompts_dev, n_steps, temperature=1.0, top_k=1)
        prof = eng.profile_read()
        eng.profile(0)
        st = args.profile_stride
        prof_steps = [i for i in range(n_steps) if i % st == st // 2]
        dh = DEC["d_model"] // N_HEAD
        # algorithmic bytes of one launch: K and V of every cached token read once (ctx includes the
        # token appended by this step and the duplicated last prompt token), fp32
        bytes_total = sum(B * N_HEAD * 2 * (Tp + i + 1) * dh * 4 for i in prof_steps) * DEC["n_layer"]
        a = prof["attn_paged"]
        if a["launches"] > 0 and a["ms"] > 0:
            ach = bytes_total / (a["ms"] * 1e-3) / 1e9
            # HBM bytes per launch: NOT measured in this run (PMC counters need their own rocprofv3 pass, and rocprofv3
            # --pmc aborts under hipGraph replay on this stack: profiles/README.md).  It is the constant of the committed
            # PMC pass over this command's eager twin, named as such: mean FETCH_SIZE [KB] x 1024 x 2 (gfx950 counts
            # half of a wide coalesced stream, MI355X_MICROARCH.md HBM section).
            traffic, traffic_source = None, None
            for name in PMC_FILES:
                pmc = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(pmc):
                    continue
                rec = json.load(open(pmc))
                for k, v in rec.items():
                    if "attn_paged_kernel" in k and isinstance(v, dict):
                        traffic = v["mean"] * 1024 * 2
                        traffic_source = (f"committed PMC constant profiles/{name} @ {rec.get('_commit', 'unknown')}: {rec.get('_command', 'rocprofv3 --pmc FETCH_SIZE')}, "
                                          f"mean of {v.get('n', v.get('dispatches', 'unknown'))} launches")
                break
            # cross-check from the committed rocprofv3 --kernel-trace --stats pass of this command (profiles/): the kernel's own average
            # duration, without the ~2.5 us of dispatch that an event pair around a single launch includes (all 1019 x 6 launches of a
            # generation there, every 16th step here: the same mean context)
            trace = None
            kst = os.path.join(ROOT, "profiles", "r4_bench_kernel_stats.csv")
            if os.path.exists(kst):
                import csv
                for r in csv.DictReader(open(kst)):
                    if "attn_paged_kernel<64, false" in r["Name"]:      # <64, false, false> since the SPLIT parameter (round 4)
                        t_us = float(r["AverageNs"]) / 1e3
                        alg = B * N_HEAD * 2 * (Tp + (n_steps - 1) / 2 + 1) * dh * 4     # mean over all steps of the generation
                        trace = dict(avg_launch_us=t_us, launches=int(r["Calls"]), achieved=alg / (t_us * 1e-6) / 1e9,
                                     frac=alg / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS, source="profiles/r4_bench_kernel_stats.csv (committed pass, not this run)")
            roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, kernel_trace=trace,
                        traffic=traffic, traffic_source=traffic_source, kernel="attn_paged_kernel<64>", launches=a["launches"],
                        avg_launch_us=a["ms"] * 1e3 / a["launches"],
                        algorithmic_bytes_per_launch=bytes_total / a["launches"],
                        step_breakdown_ms={k: round(v["ms"] / max(1, len(prof_steps)), 4) for k, v in prof.items()})

    line = None
    if rank == 0:
        gen = out.cpu()
        assert int(gen.min()) >= 0 and int(gen.max()) < DEC["vocab"]
        # whole-step algorithmic bytes: weights once per step + KV read + KV write
        step_bytes = whole_step_bytes(DEC, B, Tp, n_steps, 4, 4)
        line = {
            "metric": "midi_tokens_per_sec", "value": value, "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Decoder-S 6L/512d/8H V=8324 greedy decode (sample_kvcache semantics), "
                                   f"B={B}/GPU, prompt {Tp}, total_len {TL} ({n_steps} decode steps), random weights",
                       "global_batch": B * world, "seq_len": TL,
                       "parallelism": parallelism_label(world, backend, total * 4, t_bcast)},
            "tokens_per_sec_per_gpu": value / world,
            "per_rank_tokens_per_sec": [round(v, 1) for v in per_rank],
            "whole_step_hbm_frac": step_bytes * args.steps / (dt) / 1e9 / HBM_PEAK_GBS,
            "graph": eng.stats(),
            "roofline": roof,
        }
        if not args.no_cpu and world == 1:      # the CPU baseline and the extras are N = 1 legs (rank 0 would stall the others)
            line["cpu_baseline"] = cpu_baseline_decoder(sd, prompts.tolist(), args.cpu_seconds)
        extra = {}
        if world == 1 and not args.no_extra:
            eng.close()
            eng = None
            extra["decoder_prefill"] = prefill_extra(arena, device)
            extra["decoder_prefill_f16"] = prefill_extra(arena, device, dtype="f16")
        if not args.no_bert and not args.no_extra and world == 1:
            extra["distilbert"] = bert_extra(device, max(2, args.steps), 1, not args.no_cpu, keep_logits=True)
            extra["distilbert_bf16"] = bert_extra(device, max(10, args.steps), 2, False, dtype="bf16", ref_logits=extra["distilbert"].pop("_logits"))
        if world == 1 and not args.no_extra:
            # BASELINE configs[4] as written: fp16 storage, top-p 0.9, 2048 tokens, captured step graph (one GPU's share)
            top_p = dict(temperature=1.0, top_k=None, top_p=0.9, seed=1)
            extra["decoder_L"] = decoder_gen_extra(device, DEC_L, 12, 64, 5, 2048, "f16", top_p,
                                                   note="BASELINE configs[4] per GPU: fp16 weights + KV, fp32 accumulate; sampled (ids are not a parity claim)")
            extra["decoder_L_f32"] = decoder_gen_extra(device, DEC_L, 12, 64, 5, 2048, "f32", top_p, note="same, fp32 parity-mode storage")
            extra["decoder_S_f16"] = decoder_gen_extra(device, DEC, N_HEAD, B, Tp, TL, "f16", dict(temperature=1.0, top_k=1), arena=arena,
                                                       note="the headline generation in fp16 storage (perf mode, not the parity mode)")
            # informational: the headline generation at 4x the batch per GPU (still the fused 32-launch step)
            extra["decoder_batch256"] = large_batch_extra(arena, device, 256, Tp, TL)
            extra["decoder_b1"] = b1_extra(arena, device)
        if extra:
            line["extra"] = extra
    if eng is not None:
        eng.close()
    barrier()
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
