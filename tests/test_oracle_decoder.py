"""Pins oracle/decoder_ref.py against the golden vectors produced by the reference's own classes
(tests/golden/make_golden.py; reference api_cache.py:39-106,159-184, generate.py:25-61)."""
import hashlib

import numpy as np
import pytest
import torch

from mgea import synth
from oracle.decoder_ref import DecoderRef


def build(g):
    seed, vocab, seq_len, d_model, n_head, n_layer = (int(x) for x in g["cfg"])
    sd = synth.decoder_state_dict(seed, vocab, seq_len, d_model, n_layer)
    return DecoderRef(sd, n_head=n_head)


def prompts_of(g):
    n = sum(1 for k in g.files if k.startswith("prompt"))
    return [g[f"prompt{i}"].tolist() for i in range(n)]


@pytest.mark.parametrize("tag", ["tiny", "tiny8h"])
def test_prefill_and_step_logits(golden, tag):
    g = golden("decoder_" + tag)
    ref = build(g)
    for i, p in enumerate(prompts_of(g)):
        logits, cache, valid = ref.forward(torch.tensor([p]))
        np.testing.assert_allclose(logits[0].numpy(), g[f"prefill_logits{i}"], atol=1e-5, rtol=0)
        n_tf = g[f"step_logits{i}"].shape[0]
        ids, sl = ref.generate_greedy([p], n_tf, return_logits=True)
        np.testing.assert_allclose(sl[0].numpy(), g[f"step_logits{i}"], atol=1e-5, rtol=0)


@pytest.mark.parametrize("tag", ["tiny", "tiny8h", "S"])
def test_greedy_ids_bit_exact_solo_and_batched(golden, tag):
    g = golden("decoder_" + tag)
    ref = build(g)
    prompts = prompts_of(g)
    n_steps = len(g["greedy0"]) - len(prompts[0])
    # batched, ragged prompt lengths: every row must equal the reference's solo run
    outs = ref.generate_greedy(prompts, n_steps)
    for i, p in enumerate(prompts):
        want = g[f"greedy{i}"]
        assert outs[i] == want.tolist(), f"row {i} diverged"
        sha = hashlib.sha256(np.array(outs[i], dtype="<i8").tobytes()).digest()
        assert sha == g[f"greedy_sha{i}"].tobytes()
    solo = ref.generate_greedy([prompts[0]], 8)
    assert solo[0] == g["greedy0"][: len(prompts[0]) + 8].tolist()


def test_real_shape_step_logits(golden):
    g = golden("decoder_S")
    ref = build(g)
    prompts = prompts_of(g)
    _, sl = ref.generate_greedy(prompts[:2], 6, return_logits=True)
    for i in range(2):
        np.testing.assert_allclose(sl[i, :, :64].numpy(), g[f"step_logits_head{i}"][:6], atol=2e-5, rtol=0)
        np.testing.assert_allclose(sl[i].max(-1).values.numpy(), g[f"step_logits_max{i}"][:6], atol=2e-5, rtol=0)


def test_topk_masked_probs(golden):
    g = golden("decoder_tiny")
    ref = build(g)
    p = prompts_of(g)[0]
    _, sl = ref.generate_greedy([p], 1, return_logits=True)
    probs = DecoderRef.masked_probs(sl[:, 0], temperature=0.8, top_k=50)
    want = g["topk_probs_T0.8_k50"]
    np.testing.assert_allclose(probs[0].numpy(), want, atol=1e-6, rtol=0)
    assert int((probs[0] > 0).sum()) == 50
    # top_k=1 is exactly greedy: one-hot on the argmax
    one = DecoderRef.masked_probs(sl[:, 0], 1.0, 1)
    assert int(one.argmax()) == int(sl[0, 0].argmax()) and float(one.max()) == 1.0


def test_top_p_is_a_nucleus():
    lg = torch.log(torch.tensor([[0.5, 0.3, 0.15, 0.05]]))
    pr = DecoderRef.masked_probs(lg, 1.0, None, top_p=0.7)
    np.testing.assert_allclose(pr[0].numpy(), [0.625, 0.375, 0, 0], atol=1e-6)
    pr = DecoderRef.masked_probs(lg, 1.0, None, top_p=0.01)
    np.testing.assert_allclose(pr[0].numpy(), [1, 0, 0, 0], atol=1e-6)


@pytest.mark.parametrize("tag", ["tiny", "tiny8h"])
def test_no_kv_twin(golden, tag):
    g = golden("decoder_" + tag)
    ref = build(g)
    p = prompts_of(g)[1]
    tl = ref.forward_twin(torch.tensor([p]))
    np.testing.assert_allclose(tl[0].numpy(), g["twin_logits1"], atol=1e-5, rtol=0)
    n = len(g["twin_greedy1"]) - len(p)
    assert ref.generate_greedy_twin(p, n) == g["twin_greedy1"].tolist()


def test_prompt_longer_than_position_table_raises(golden):
    g = golden("decoder_tiny")
    ref = build(g)
    with pytest.raises(RuntimeError):
        ref.forward(torch.zeros(1, ref.seq_len + 1, dtype=torch.long))
