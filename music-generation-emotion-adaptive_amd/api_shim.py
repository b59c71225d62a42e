"""The reference's serving entry point on top of the MI355X hot path.

`create_app()` builds a FastAPI app whose `POST /generate` body follows api_cache.py:186-243 step by
step -- `inference.predict` -> `EATS.get_music_params` -> `closest_bpm_token` /
`normalize_key_signature` / `FAMILY_TO_INSTRUMENTS` -> `sample_kvcache` -> note tokens -> MIDI -- using
this repo's drop-in modules.  The only differences are the ones the offline box forces: the MIDI file
is written by generate_music.midi (pretty_midi absent) and returned directly as `audio/midi` instead
of being rendered to WAV by FluidSynth (midi2audio and the SoundFont are absent; rendering is outside
the accelerated path).
"""
from __future__ import annotations


def create_app(model, seq_len: int, temperature: float = 1.0, top_k: int = 50):
    from fastapi import FastAPI, Form
    from fastapi.middleware.cors import CORSMiddleware
    from fastapi.responses import Response

    import generate_music.generate as gen
    from emotion_analysis import EATS, inference
    from generate_music.midi import tokens_to_midi

    app = FastAPI()
    app.add_middleware(CORSMiddleware, allow_origins=["*"], allow_methods=["*"], allow_headers=["*"])
    try:   # the reference reads a multipart form field (api_cache.py:187); that needs python-multipart
        import multipart  # noqa: F401
        prompt_param = Form(...)
    except ImportError:   # absent on the offline box: same endpoint, `prompt` as a query parameter
        from fastapi import Query
        prompt_param = Query(...)
    app.state.prompt_in = "form" if prompt_param.__class__.__name__ == "Form" else "query"

    @app.post("/generate")
    def generate_music(prompt: str = prompt_param):
        label = inference.predict(prompt)                                     # api_cache.py:189
        mapping = EATS.get_music_params(label)                                # :190
        bpm_tok = gen.closest_bpm_token(mapping["bpm"])                       # :194
        key = gen.normalize_key_signature(mapping["key"])                     # :195
        instruments = []
        for fam in mapping["all_families"]:                                   # :196-198
            instruments.extend(gen.FAMILY_TO_INSTRUMENTS.get(fam, []))
        gen_prompt = ["[START_SEQUENCE]", bpm_tok, key] + [f"[INSTRUMENT] {i}" for i in instruments]   # :203
        tokens = gen.sample_kvcache(model, gen_prompt, max_len=seq_len, temperature=temperature, top_k=top_k,
                                    device="cpu")                             # :204
        midi = tokens_to_midi(tokens)                                         # :208-221 (+ pm.write)
        return Response(content=midi, media_type="audio/midi",
                        headers={"X-Emotion": label, "X-Prompt-Tokens": str(len(gen_prompt)),
                                 "X-Generated-Tokens": str(len(tokens))})

    return app
