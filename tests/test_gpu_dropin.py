"""The reference's own call surface on top of the HIP engines: generate_music.generate
(GPTWithKV / sample_kvcache / sample / generate_sequence) and emotion_analysis.inference
(predict & co.), checked against golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from mgea import synth

pytestmark = pytest.mark.gpu


def setup_decoder(g, cls_name="GPTWithKV"):
    import generate_music.generate as gen
    seed, vocab, seq_len, d_model, n_head, n_layer = (int(x) for x in g["cfg"])
    sd = synth.decoder_state_dict(seed, vocab, seq_len, d_model, n_layer)
    gen.set_vocab(synth.decoder_vocab(vocab))
    if cls_name == "GPTWithKV":
        m = gen.GPTWithKV(vocab_size=vocab, seq_len=seq_len, d_model=d_model, n_head=n_head, n_layer=n_layer)
        assert m.load_state_dict(gen.remap_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})) == "<All keys matched successfully>"
    else:
        m = gen.GPT(vocab, seq_len + 1, d_model, n_head=n_head, n_layer=n_layer)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    return gen, m, sd


def test_sample_kvcache_greedy_tokens_equal_reference(golden):
    g = golden("decoder_tiny")
    gen, m, _ = setup_decoder(g)
    for i in range(3):
        ids = g[f"prompt{i}"].tolist()
        want = [gen.id2tok[j] for j in g[f"greedy{i}"].tolist()]
        toks = gen.sample_kvcache(m, [gen.id2tok[j] for j in ids], max_len=len(want), temperature=1.0, top_k=1, device="cpu")
        assert toks == want
        assert gen.generate_sequence(m, [gen.id2tok[j] for j in ids], max_len=len(want), top_k=1) == want
    with pytest.raises(KeyError):
        gen.sample_kvcache(m, ["[NO SUCH TOKEN]"], max_len=8)


def test_model_call_surface_like_the_reference_loop(golden):
    """logits, past = model(ids); then model(last, past) per step (api_cache.py:163-168)."""
    g = golden("decoder_tiny")
    gen, m, _ = setup_decoder(g)
    ids = torch.tensor(g["prompt1"]).unsqueeze(0)
    logits, past = m(ids)
    np.testing.assert_allclose(logits[0].cpu().numpy(), g["prefill_logits1"], atol=1e-3, rtol=0)
    assert len(past) == m.n_layer
    generated = ids
    for s in range(4):
        lg, past = m(generated[:, -1:], past)
        np.testing.assert_allclose(lg[0, -1].cpu().numpy(), g["step_logits1"][s], atol=1e-3, rtol=0)
        generated = torch.cat([generated, lg[:, -1].argmax(-1, keepdim=True).cpu()], 1)
    assert generated[0].tolist() == g["greedy1"][: generated.shape[1]].tolist()
    stale = past
    _, past2 = m(generated[:, -1:], past)
    with pytest.raises(RuntimeError):
        m(generated[:, -1:], stale)          # an old `presents` cannot rewind the native cache


@pytest.mark.parametrize("tag", ["tiny", "tiny8h", "S"])
def test_twin_gpt_and_sample(golden, tag):
    """generate_music/generate.py:25-61 behind its own names (GPT, sample), at d_model 128, 256 and the Decoder-S shape."""
    g = golden("decoder_" + tag)
    gen, m, _ = setup_decoder(g, "GPT")
    p = g["prompt1"].tolist()
    logits = m(torch.tensor([p]))
    want_lg = g["twin_logits1"]
    np.testing.assert_allclose(logits[0].cpu().numpy()[:, : want_lg.shape[1]], want_lg, atol=1e-3, rtol=0)
    gen.model = m
    want = [gen.id2tok[j] for j in g["twin_greedy1"].tolist()]
    assert gen.sample([gen.id2tok[j] for j in p], max_len=len(want), temperature=1.0, top_k=1) == want


def test_generate_batch_rows_equal_solo_runs(golden):
    g = golden("decoder_tiny")
    gen, m, _ = setup_decoder(g)
    prompts = [[gen.id2tok[j] for j in g[f"prompt{i}"].tolist()] for i in range(3)]
    n = 20
    rows = gen.generate_batch(m, prompts, max_len=max(len(p) for p in prompts) + n, top_k=1)
    for i, p in enumerate(prompts):
        assert rows[i] == [gen.id2tok[j] for j in g[f"greedy{i}"].tolist()][: len(p) + n]


def test_config0_plumbing_label_to_tokens(golden):
    """BASELINE config 0 on the box: label -> EATS -> control-token prompt -> greedy tokens == oracle."""
    import random
    from emotion_analysis import EATS
    from oracle.decoder_ref import DecoderRef
    g = golden("decoder_tiny8h")
    gen, m, sd = setup_decoder(g)
    random.seed(3)
    mapping = EATS.get_music_params("admiration")
    instruments = [i for fam in mapping["all_families"] for i in gen.FAMILY_TO_INSTRUMENTS.get(fam, [])]
    prompt = ["[START_SEQUENCE]", gen.closest_bpm_token(mapping["bpm"]), gen.normalize_key_signature(mapping["key"])] + \
             [f"[INSTRUMENT] {i}" for i in instruments]
    toks = gen.sample_kvcache(m, prompt, max_len=len(prompt) + 24, temperature=1.0, top_k=1)
    want = DecoderRef(sd, int(g["cfg"][4])).generate_greedy([[gen.tok2id[t] for t in prompt]], 24)[0]
    assert [gen.tok2id[t] for t in toks] == want


@pytest.fixture(scope="module")
def emotion():
    from emotion_analysis import inference
    from mgea.bert import BertEngine
    from mgea.tokenizer import WordPieceTokenizer
    from oracle.distilbert_ref import DistilBertRef
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"w{i}" for i in range(60)] + \
            "i am walking down a road and see rainbow it is sunny . love life so happy sad angry ! ? , the".split()
    vocab = {w: i for i, w in enumerate(dict.fromkeys(words))}
    sd = synth.distilbert_state_dict(51, len(vocab), 64, 128, 2, 512)
    ad = synth.lora_adapter(51, 128, 2)
    tok = WordPieceTokenizer(vocab)
    eng = BertEngine(sd, n_heads=2, adapter=ad, max_tokens=8 * 64)
    inference.configure(tok, eng)
    return inference, tok, DistilBertRef(sd, 2, ad)


def test_inference_functions_match_oracle(emotion):
    from emotion_analysis.config import ID2LABEL
    inference, tok, ref = emotion
    text = "i am walking down a road and i see a rainbow and it is sunny. i love life."
    enc = tok(text)
    logits = ref.forward(enc["input_ids"], enc["attention_mask"])
    probs = torch.softmax(logits, 1)[0]
    assert inference.predict(text) == ID2LABEL[int(logits.argmax())] == inference.classify(text)
    allp = inference.predict_all_labels(text)
    assert list(allp) == [ID2LABEL[i] for i in range(28)]
    np.testing.assert_allclose(list(allp.values()), [round(float(p), 4) for p in probs], atol=1.1e-4)
    top = inference.predict_top_k_labels(text, k=3)
    assert [t[0] for t in top] == [ID2LABEL[int(i)] for i in probs.topk(3).indices] and top[0][1] >= top[1][1] >= top[2][1]
    thr = inference.predict_labels_above_threshold(text, threshold=0.03)
    assert [t[0] for t in thr] == [ID2LABEL[i] for i, p in enumerate(probs) if float(p) > 0.03]
    texts = [text, "so sad !", "happy ? i am angry"]
    enc = tok(texts)
    want = ref.forward(enc["input_ids"], enc["attention_mask"]).argmax(1).tolist()
    assert inference.classify(texts) == [ID2LABEL[i] for i in want]
    assert inference.classify(enc["input_ids"], enc["attention_mask"]) == [ID2LABEL[i] for i in want]
    trace = inference.analyze_emotion_transitions("I am happy. It is sunny!")
    assert [s for s, _ in trace] == ["I am happy.", "It is sunny!"]
