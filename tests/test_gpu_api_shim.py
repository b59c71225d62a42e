"""POST /generate through FastAPI's TestClient on top of the HIP engines (BASELINE configs[0] plumbing):
text -> emotion label -> control-token prompt -> greedy MIDI tokens -> MIDI bytes."""
import random

import pytest
import torch

from mgea import synth

pytestmark = pytest.mark.gpu


def test_generate_endpoint_end_to_end(golden):
    pytest.importorskip("httpx")
    from fastapi.testclient import TestClient
    import generate_music.generate as gen
    from api_shim import create_app
    from emotion_analysis import inference
    from mgea.bert import BertEngine
    from mgea.tokenizer import WordPieceTokenizer
    from oracle.decoder_ref import DecoderRef

    # decoder with the tiny8h fixture weights / vocabulary
    g = golden("decoder_tiny8h")
    seed, vocab, seq_len, d_model, n_head, n_layer = (int(x) for x in g["cfg"])
    sd = synth.decoder_state_dict(seed, vocab, seq_len, d_model, n_layer)
    gen.set_vocab(synth.decoder_vocab(vocab))
    model = gen.GPTWithKV(vocab, seq_len, d_model, n_head, n_layer)
    model.load_state_dict(gen.remap_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}))
    # classifier with a synthetic vocabulary
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + "i am walking down a road and see rainbow it is sunny . love life".split()
    vmap = {w: i for i, w in enumerate(dict.fromkeys(words))}
    bsd = synth.distilbert_state_dict(61, len(vmap), 64, 128, 2, 512)
    inference.configure(WordPieceTokenizer(vmap), BertEngine(bsd, n_heads=2, adapter=synth.lora_adapter(61, 128, 2), max_tokens=64))

    app = create_app(model, seq_len=32, temperature=1.0, top_k=1)
    client = TestClient(app)
    text = "i am walking down a road and i see a rainbow. i love life."
    kw = {"data": {"prompt": text}} if app.state.prompt_in == "form" else {"params": {"prompt": text}}
    random.seed(11)
    r = client.post("/generate", **kw)
    assert r.status_code == 200 and r.headers["content-type"].startswith("audio/midi")
    assert r.content[:4] == b"MThd" and int(r.headers["x-generated-tokens"]) == 32
    label = r.headers["x-emotion"]
    assert label == inference.predict("i am walking down a road and i see a rainbow. i love life.")
    # same request path by hand -> oracle greedy ids
    from emotion_analysis import EATS
    random.seed(11)
    mapping = EATS.get_music_params(label)
    instruments = [i for fam in mapping["all_families"] for i in gen.FAMILY_TO_INSTRUMENTS.get(fam, [])]
    prompt = ["[START_SEQUENCE]", gen.closest_bpm_token(mapping["bpm"]), gen.normalize_key_signature(mapping["key"])] + \
             [f"[INSTRUMENT] {i}" for i in instruments]
    toks = gen.sample_kvcache(model, prompt, max_len=32, top_k=1)
    want = DecoderRef(sd, n_head).generate_greedy([[gen.tok2id[t] for t in prompt]], 32 - len(prompt))[0]
    assert [gen.tok2id[t] for t in toks] == want
    assert client.post("/generate").status_code == 422          # FastAPI's own validation, as in the reference
