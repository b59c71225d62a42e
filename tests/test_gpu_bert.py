"""Parity of the HIP DistilBERT(+LoRA) forward against the golden logits of the `transformers`
class (bit-exact argmax labels, logits within 1e-3)."""
import numpy as np
import pytest
import torch

from mgea import synth

pytestmark = pytest.mark.gpu


def make(g, with_adapter=True):
    from mgea.bert import BertEngine
    seed, vocab, max_pos, dim, n_heads, n_layers, hidden, batch, seq = (int(x) for x in g["cfg"])
    sd = synth.distilbert_state_dict(seed, vocab, max_pos, dim, n_layers, hidden)
    ad = synth.lora_adapter(seed, dim, n_layers) if with_adapter else None
    return BertEngine(sd, n_heads=n_heads, adapter=ad, max_tokens=batch * seq)


@pytest.mark.parametrize("tag", ["tiny", "base"])
@pytest.mark.parametrize("full_last", [0, 1])
def test_logits_and_labels_vs_golden(golden, tag, full_last, tune):
    """Both forms of the last layer against the transformers-generated golden: [CLS] rows only (the engine's default: K | V of every
    position, query / attention / out-proj / FFN for the B rows the classifier reads) and every position (as the reference computes it)."""
    tune("bert_full_last_layer", full_last)
    g = golden("distilbert_" + tag)
    eng = make(g)
    logits, amax = eng.forward(torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    assert eng.stats()["last_layer_cls_only"] == (not full_last)
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], atol=1e-3, rtol=0)
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 1e-4
    assert amax.cpu().tolist() == g["argmax"].tolist()


def test_solo_row_without_padding(golden):
    g = golden("distilbert_tiny")
    eng = make(g)
    n = int(g["mask"][1].sum())
    logits, _ = eng.forward(torch.from_numpy(g["ids"][1:2, :n]), None)
    np.testing.assert_allclose(logits.cpu().numpy(), g["solo_row1_logits"], atol=1e-4, rtol=0)


def test_lora_changes_the_result(golden):
    g = golden("distilbert_tiny")
    a, _ = make(g, True).forward(torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    b, _ = make(g, False).forward(torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    assert float((a - b).abs().max()) > 1e-3


def test_sequence_longer_than_position_table(golden):
    g = golden("distilbert_tiny")
    eng = make(g)
    with pytest.raises(RuntimeError):
        eng.forward(torch.zeros(1, 33, dtype=torch.long), None)


@pytest.mark.parametrize("dim,n_heads,S", [(128, 4, 200), (128, 2, 77), (256, 4, 513), (64, 2, 4)])
def test_last_layer_on_cls_rows_only_equals_every_position(dim, n_heads, S, tune):
    """The [CLS]-rows-only last layer against the same engine computing every position: head_dim 32 and 64, sequence lengths that are
    not a multiple of the 64-key chunks of the one-query attention kernel (and one longer than a chunk's eight-fold), ragged key
    masks.  Same function of the inputs; the GEMMs of 5 rows instead of 5 x S take another split of K, so equality is to fp32
    rounding, not bitwise."""
    from mgea.bert import BertEngine
    B, vocab, L = 5, 500, 2
    sd = synth.distilbert_state_dict(17, vocab, 520, dim, L, 4 * dim)
    ids = torch.from_numpy(synth.integers(17, "ids", (B, S), 0, vocab))
    mask = torch.ones(B, S, dtype=torch.int64)
    for b in range(1, B):
        mask[b, max(1, S - (S // 6) * b):] = 0
    out = {}
    for full_last in (0, 1):
        tune("bert_full_last_layer", full_last)
        eng = BertEngine(sd, n_heads=n_heads, max_tokens=B * S)
        logits, amax = eng.forward(ids, mask)
        assert eng.stats()["last_layer_cls_only"] == (not full_last)
        out[full_last] = (logits.cpu().numpy(), amax.cpu().numpy())
        eng.close()
    assert np.abs(out[0][0] - out[1][0]).max() < 2e-5
    assert (out[0][1] == out[1][1]).all()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_device_resident_ids_outside_the_vocabulary_are_reported(dtype, tune):
    """nn.Embedding raises IndexError for an id outside the vocabulary (emotion_analysis/inference.py:16-17 -> transformers).  Host
    tensors are checked before the upload; DEVICE tensors are not read back (no host sync in front of a forward): the embedding kernels
    clamp such ids and set the engine's sticky flag, which id_errors() turns into the same IndexError (the check had been
    dropped without a replacement).  Both embedding kernels: the fp32 one and, on a bf16 engine forced onto its 16-bit path, the bf16 one."""
    from mgea.bert import BertEngine
    sd = synth.distilbert_state_dict(5, 100, 128, 128, 2, 512)
    eng = BertEngine(sd, n_heads=2, max_tokens=1024, dtype=dtype)
    B, S = (8, 128) if dtype == "bf16" else (2, 8)       # bf16 engines take their 16-bit kernels from 512 tokens on
    good = torch.randint(1, 100, (B, S), dtype=torch.int32)
    with pytest.raises(IndexError):
        eng.forward(torch.full((1, 4), 100))               # host tensor: refused before any GPU work
    eng.forward(good.cuda())
    assert eng.id_errors(raise_error=False) == 0
    bad = good.clone()
    bad[B - 1, 3] = 100
    eng.forward(bad.cuda())                                # no exception yet: nothing was synchronised
    with pytest.raises(IndexError):
        eng.id_errors()
    assert eng.id_errors(raise_error=False) == 0            # reading clears the flag
    bad[0, 0] = -1
    eng.forward(bad.cuda())
    assert eng.id_errors(raise_error=False) == 1
    eng.close()


@pytest.mark.parametrize("tag", ["tiny", "base"])
def test_packed_forward_of_the_f32_engine_vs_golden_and_padded(golden, tag):
    """The PACKED forward (mgea_bert_forward_packed) on the parity engine: the golden fixture's ragged rows, given as real tokens back to
    back, against the transformers-generated golden logits (1e-4, labels exact) and against the padded call of the same engine
    (fp32 summation noise: the attention cuts a sequence's keys into the same 64-key tiles in both forms, the GEMMs cut the rows into
    different 128-row tiles of identical per-element arithmetic -> observed exactly equal).  Both forms of the last layer."""
    from mgea import _lib
    from mgea.bert import BertEngine
    g = golden("distilbert_" + tag)
    seed, vocab, max_pos, dim, n_heads, n_layers, hidden, gb, seq = (int(x) for x in g["cfg"])
    sd = synth.distilbert_state_dict(seed, vocab, max_pos, dim, n_layers, hidden)
    ad = synth.lora_adapter(seed, dim, n_layers)
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    eng = BertEngine(sd, n_heads=n_heads, adapter=ad, max_tokens=gb * seq, dtype="f32")
    for full_last in (0, 1):
        old = _lib.tune_set("bert_full_last_layer", full_last)
        try:
            padded, _ = eng.forward(ids, mask)
            pk = BertEngine.pack(ids, mask)
            assert pk is not None and pk[0].numel() == int(mask.sum()) < ids.numel()
            logits, amax = eng.forward_packed(*pk)
            assert eng.stats()["rows"] == int(mask.sum())
            auto, _ = eng.forward_auto(ids, mask)
            assert eng.stats()["rows"] == int(mask.sum()) and torch.equal(auto, logits)
        finally:
            _lib.tune_set("bert_full_last_layer", old)
        d_gold = np.abs(logits.cpu().numpy() - g["logits"]).max()
        d_pad = float((logits - padded).abs().max())
        print(f"[f32 packed {tag}, full_last={full_last}] vs golden {d_gold:.2e}, vs padded {d_pad:.2e}")
        assert d_gold < 1e-4 and d_pad < 2e-5
        assert amax.cpu().tolist() == g["argmax"].tolist()
    eng.close()
