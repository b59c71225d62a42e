#!/usr/bin/env python3
"""Benchmark of the MI355X hot path on BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL weight broadcast over xGMI)

A "step" = one pass of the hot path over one batch of synthetic input = one greedy generation of
configs[2]: Decoder-S (6L / 512d / 8H, V = 8324, random weights), batch 64 per GPU, 5-token
prompts decoded to a total length of 1024 (1019 decode steps, sample_kvcache semantics incl. the
re-fed last prompt token).  value = generated MIDI tokens / s over all ranks, inputs and weights
resident in HBM, timed with barrier + torch.cuda.synchronize() on both sides, max over ranks.
Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` for the dominant
kernel (paged decode attention; HIP events on the launch stream) and `cpu_baseline` (the oracle =
CPU port of the reference semantics, timed on this host's cores).  DistilBERT prompts/s (the
other half of BASELINE.json's metric) is reported under "extra".
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
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
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
    """Oracle (CPU port of api_cache.py semantics with a projected-KV cache) on the same prompts;
    bounded sample: decode steps until ~budget_s of CPU work."""
    from oracle.decoder_ref import DecoderRef
    torch.set_num_threads(host_threads())
    ref = DecoderRef(sd, n_head=N_HEAD)
    idx = torch.tensor(prompts)
    B = idx.shape[0]
    t0 = time.perf_counter()
    _, cache, valid = ref.forward(idx)
    last = idx[:, -1:]
    n = 0
    while True:
        logits, cache, valid = ref.forward(last, cache, valid)
        last = logits[:, -1, :].argmax(-1, keepdim=True)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 512:
            break
    dt = time.perf_counter() - t0
    return dict(value=B * n / dt, unit="tokens/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle/decoder_ref.py (torch CPU fp32), B={B}, prompt {idx.shape[1]}, first {n} decode "
                       f"steps (ctx <= {idx.shape[1] + n}), {dt:.1f} s")


def bert_extra(device, steps, warmup, with_cpu, dtype="f32", sd=None, ad=None):
    from mgea import synth
    from mgea.bert import BertEngine
    B, S = 256, 128
    if sd is None:
        sd = synth.distilbert_state_dict(41, BERT["vocab"], BERT["max_pos"], BERT["dim"], BERT["n_layers"], BERT["hidden"])
        ad = synth.lora_adapter(41, BERT["dim"], BERT["n_layers"])
    eng = BertEngine(sd, n_heads=12, adapter=ad, max_tokens=B * S, device=device, dtype=dtype)
    ids, mask = synth.bert_inputs(2, B, S, BERT["vocab"])
    ids, mask = torch.from_numpy(ids).to(device), torch.from_numpy(mask).to(device)
    for _ in range(max(1, warmup)):
        eng.forward(ids, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward(ids, mask)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    D, FF, L = BERT["dim"], BERT["hidden"], BERT["n_layers"]
    flops = 2 * B * S * L * (4 * D * D + 2 * D * FF) + 4 * B * S * S * D * L + 2 * B * (D * D + 28 * D)
    out = dict(metric="distilbert_prompts_per_sec", value=B / dt, unit="prompts/s", ms_per_batch=dt * 1e3,
               dtype=dtype, workload="DistilBERT-base(+LoRA merged) classifier B=256 S=128 padded rows, random weights",
               roofline=dict(bound="mfma", achieved=flops / dt / 1e12, peak=2500.0 if dtype == "bf16" else 157.3,
                             unit="TFLOP/s", frac=flops / dt / 1e12 / (2500.0 if dtype == "bf16" else 157.3), traffic=None,
                             note="whole-forward algorithmic FLOPs / wall time vs the dense MFMA peak of the dtype"))
    if with_cpu:
        from oracle.distilbert_ref import DistilBertRef
        ref = DistilBertRef(sd, 12, ad)
        n = 32
        t0 = time.perf_counter()
        ref.forward(ids[:n].cpu().long(), mask[:n].cpu().long())
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=n / dtc, unit="prompts/s", cores=torch.get_num_threads(), kind="port",
                                   sample=f"oracle/distilbert_ref.py, first {n} rows of the batch, one forward, {dtc:.1f} s")
    eng.close()
    return out


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


def main():
    args = parse()
    from mgea import dist as mdist
    from mgea import synth
    from mgea.decoder import DecoderEngine, arena_layout
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    # production: "nccl" (= RCCL over xGMI), one rank per GPU.  MGEA_DIST_BACKEND=gloo rehearses the
    # N > 1 flow on a box with fewer GPUs than ranks (ranks then share devices round-robin).
    backend = os.environ.get("MGEA_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend != "nccl":
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev))
    rank, world, local = mdist.init_from_env(backend)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
            # HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE run of this command's
            # eager twin (profiles/README.md): mean FETCH_SIZE x 1024 x 2 (gfx950 half-count correction)
            traffic, traffic_note = None, None
            pmc = os.path.join(ROOT, "profiles", "r1_v6_pmc_fetch_size_full.json")
            if os.path.exists(pmc):
                for k, v in json.load(open(pmc)).items():
                    if "attn_paged_kernel" in k:
                        traffic = v["mean"] * 1024 * 2
                        traffic_note = ("mean over all 6114 attention launches of the same generation run eagerly "
                                        "(MGEA_DECODER_NOGRAPH=1: rocprofv3 --pmc crashes under hipGraph replay); "
                                        "the 3 % above the algorithmic bytes is the whole-page K read of the first, speculative tile at contexts <= 256 and line granularity")
            roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                        traffic=traffic, traffic_note=traffic_note, kernel="attn_paged_kernel<64>", launches=a["launches"],
                        avg_launch_us=a["ms"] * 1e3 / a["launches"],
                        algorithmic_bytes_per_launch=bytes_total / a["launches"],
                        step_breakdown_ms={k: round(v["ms"] / max(1, len(prof_steps)), 4) for k, v in prof.items()})

    line = None
    if rank == 0:
        gen = out.cpu()
        assert int(gen.min()) >= 0 and int(gen.max()) < DEC["vocab"]
        # whole-step algorithmic bytes: weights once per step + KV read + KV write
        C_, NL, V = DEC["d_model"], DEC["n_layer"], DEC["vocab"]
        p_step = NL * (12 * C_ * C_ + 13 * C_) + V * C_ + V
        step_bytes = sum(p_step * 4 + B * NL * 2 * C_ * 4 * (Tp + i + 1 + 1) for i in range(n_steps))
        line = {
            "metric": "midi_tokens_per_sec", "value": value, "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Decoder-S 6L/512d/8H V=8324 greedy decode (sample_kvcache semantics), "
                                   f"B={B}/GPU, prompt {Tp}, total_len {TL} ({n_steps} decode steps), random weights",
                       "global_batch": B * world, "seq_len": TL,
                       "parallelism": f"dp{world} replicas, one RCCL weight broadcast ({total * 4 / 1e6:.1f} MB, "
                                      f"{t_bcast * 1e3:.1f} ms), no collective in decode"},
            "tokens_per_sec_per_gpu": value / world,
            "whole_step_hbm_frac": step_bytes * args.steps / (dt) / 1e9 / HBM_PEAK_GBS,
            "graph": eng.stats(),
            "roofline": roof,
        }
        if not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline_decoder(sd, prompts.tolist(), args.cpu_seconds)
        if not args.no_bert:
            line["extra"] = {"distilbert": bert_extra(device, max(2, args.steps), 1, not args.no_cpu),
                             "distilbert_bf16": bert_extra(device, max(3, args.steps), 2, False, dtype="bf16")}
            if world == 1:   # informational: the same generation at 4x the batch per GPU (still the fused 32-launch step)
                line["extra"]["decoder_batch256"] = large_batch_extra(arena, device, 256, Tp, TL)
    eng.close()
    barrier()
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
