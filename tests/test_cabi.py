"""The C-ABI library loads and exports every symbol include/mgea.h declares; host-only entry
points work without a GPU; engines refuse to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from mgea import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mgea.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mgea_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in mgea.h but not exported by libmgea_hip.so"
        assert s in _lib.PROTOTYPES, f"{s} has no ctypes prototype in mgea/_lib.py"
    assert set(_lib.PROTOTYPES) <= set(syms), "ctypes binds symbols the header does not declare"


def test_arena_layouts_are_host_side():
    lib = _lib.load()
    cfg = _lib.DecoderConfig(vocab=8324, seq_len=1024, d_model=512, n_head=8, n_layer=6, d_ff=2048, max_batch=64,
                             max_ctx=1024, dtype=0, block_mode=0, pos_mode=0, ln_eps=1e-5)
    n, total = C.c_int32(0), C.c_int64(0)
    assert lib.mgea_decoder_arena_layout(C.byref(cfg), None, C.byref(n), C.byref(total)) == 0
    assert n.value == 2 + 12 * 6 + 2
    params = 8324 * 512 * 2 + 1024 * 512 + 6 * (12 * 512 * 512 + 13 * 512) + 8324
    assert params <= total.value <= params + 64 * n.value      # 256-byte alignment padding only
    offs = (C.c_int64 * n.value)()
    assert lib.mgea_decoder_arena_layout(C.byref(cfg), offs, C.byref(n), C.byref(total)) == 0
    assert list(offs) == sorted(offs) and all(o % 64 == 0 for o in offs)
    bcfg = _lib.BertConfig(vocab=30522, max_pos=512, dim=768, n_heads=12, n_layers=6, hidden=3072, num_labels=28,
                           max_tokens=128, dtype=0, ln_eps=1e-12)
    assert lib.mgea_bert_arena_layout(C.byref(bcfg), None, C.byref(n), C.byref(total)) == 0
    assert n.value == 4 + 12 * 6 + 4 and total.value >= 66_955_000


def test_bad_config_is_einval_with_message():
    lib = _lib.load()
    cfg = _lib.DecoderConfig(vocab=100, seq_len=16, d_model=100, n_head=8, n_layer=1, d_ff=400, max_batch=1,
                             max_ctx=16, dtype=0, block_mode=0, pos_mode=0, ln_eps=1e-5)
    n, total = C.c_int32(0), C.c_int64(0)
    assert lib.mgea_decoder_arena_layout(C.byref(cfg), None, C.byref(n), C.byref(total)) == _lib.EINVAL
    assert "divisible" in _lib.last_error()
    with pytest.raises(RuntimeError):
        _lib.check(_lib.EINVAL)
    with pytest.raises(ValueError):
        _lib.check(_lib.ECAPACITY)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_a_gpu():
    from mgea import synth
    from mgea.decoder import DecoderEngine
    assert _lib.load().mgea_device_count() < 0
    with pytest.raises(Exception):
        DecoderEngine(synth.decoder_state_dict(1, 64, 16, 128, 1), n_head=2, device="cuda:0")
    with pytest.raises(RuntimeError):
        DecoderEngine(synth.decoder_state_dict(1, 64, 16, 128, 1), n_head=2, device="cpu")


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "music-generation-emotion-adaptive_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f), encoding="utf-8").read()
                assert "import oracle" not in txt and "from oracle" not in txt, f"{f} imports the oracle"


def test_tiled_weight_size_is_host_arithmetic():
    """mgea_op_tiled_weight_floats runs without a GPU: rows padded to a multiple of 32, K unchanged."""
    from mgea import _lib
    lib = _lib.load()
    assert lib.mgea_op_tiled_weight_floats(8324, 512) == 8352 * 512
    assert lib.mgea_op_tiled_weight_floats(512, 2048) == 512 * 2048
    assert lib.mgea_op_tiled_weight_floats(1, 32) == 32 * 32


def test_switch_table_is_host_side_and_restorable():
    """mgea_tune_set / mgea_tune_get (the library's A/B and test switches) work without a GPU; unknown names are EINVAL with a
    message; nothing reads the environment after load (setting MGEA_* now changes nothing)."""
    import os
    from mgea import _lib
    lib = _lib.load()
    v = C.c_int32(-1)
    assert lib.mgea_tune_get(b"bf16_gemm_tail", C.byref(v)) == 0 and v.value in (0, 1, 2)
    old = _lib.tune_set("bf16_gemm_tail", 0)
    try:
        assert _lib.tune_get("bf16_gemm_tail") == 0
        os.environ["MGEA_BF16_GEMM_TAIL"] = "1"
        assert _lib.tune_get("bf16_gemm_tail") == 0           # the environment was read once, when the library was loaded
    finally:
        os.environ.pop("MGEA_BF16_GEMM_TAIL", None)
        _lib.tune_set("bf16_gemm_tail", old)
    assert _lib.tune_get("bf16_gemm_tail") == old
    assert lib.mgea_tune_set(b"no_such_switch", 1) == _lib.EINVAL and "no_such_switch" in _lib.last_error()
    for name in ("bf16_gemm_tile", "bf16_gemm_small", "bf16_gemm_phases", "bert_bf16_nofold", "bert_full_last_layer", "decoder_prefill_full", "decoder_unfused", "decoder_nogemv", "decoder_nograph",
                 "decoder_prefill16"):
        assert lib.mgea_tune_get(name.encode(), C.byref(v)) == 0


def test_no_getenv_on_native_launch_paths():
    """the only getenv in csrc/ is the one-time table initialiser in capi.hip."""
    csrc = os.path.join(ROOT, "music-generation-emotion-adaptive_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            for i, line in enumerate(open(os.path.join(csrc, f), encoding="utf-8"), 1):
                if "getenv" in line and not line.lstrip().startswith("//"):
                    hits.append((f, i))
    assert [h[0] for h in hits] == ["capi.hip"], hits
