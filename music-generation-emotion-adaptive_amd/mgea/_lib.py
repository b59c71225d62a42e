"""ctypes binding of the C ABI in include/mgea.h (libmgea_hip.so, built by csrc/Makefile).

There is NO CPU fallback: if the shared library is missing, or no HIP device is visible when an
engine is created, this raises -- the product path never routes through oracle/ or PyTorch math.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MGEA_LIB_PATH: tools/ only -- an instrumented build of the same library, e.g. tools/libmgea_hip_stamps.so)
LIB_PATH = os.environ.get("MGEA_LIB_PATH") or os.path.join(_HERE, "libmgea_hip.so")

OK, EINVAL, ENOMEM, EHIP, ECAPACITY, ENODEVICE = 0, -1, -2, -3, -4, -5
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
BLOCK_PRELN_GELU, BLOCK_POSTLN_RELU = 0, 1
POS_REFERENCE, POS_ABSOLUTE = 0, 1
KV_PAGE_TOKENS = 64


class MgeaError(RuntimeError):
    """A HIP / allocation failure inside the native library."""


class DecoderConfig(C.Structure):
    _fields_ = [("vocab", C.c_int32), ("seq_len", C.c_int32), ("d_model", C.c_int32), ("n_head", C.c_int32),
                ("n_layer", C.c_int32), ("d_ff", C.c_int32), ("max_batch", C.c_int32), ("max_ctx", C.c_int32),
                ("dtype", C.c_int32), ("block_mode", C.c_int32), ("pos_mode", C.c_int32), ("ln_eps", C.c_float)]


class SamplerConfig(C.Structure):
    _fields_ = [("temperature", C.c_float), ("top_k", C.c_int32), ("top_p", C.c_float), ("eos_id", C.c_int32),
                ("seed", C.c_uint64)]


class BertConfig(C.Structure):
    _fields_ = [("vocab", C.c_int32), ("max_pos", C.c_int32), ("dim", C.c_int32), ("n_heads", C.c_int32),
                ("n_layers", C.c_int32), ("hidden", C.c_int32), ("num_labels", C.c_int32),
                ("max_tokens", C.c_int32), ("dtype", C.c_int32), ("ln_eps", C.c_float)]


_P = C.c_void_p
_I32, _I64, _F = C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol include/mgea.h declares
PROTOTYPES = {
    "mgea_last_error": (C.c_char_p, []),
    "mgea_version": (C.c_int, []),
    "mgea_device_count": (C.c_int, []),
    "mgea_tune_set": (C.c_int, [C.c_char_p, _I32]),
    "mgea_tune_get": (C.c_int, [C.c_char_p, C.POINTER(_I32)]),
    "mgea_decoder_arena_layout": (C.c_int, [C.POINTER(DecoderConfig), C.POINTER(_I64), C.POINTER(_I32), C.POINTER(_I64)]),
    "mgea_decoder_create": (C.c_int, [C.POINTER(DecoderConfig), _P, C.POINTER(_P)]),
    "mgea_decoder_destroy": (C.c_int, [_P]),
    "mgea_decoder_refresh_weights": (C.c_int, [_P, _P]),
    "mgea_decoder_reset": (C.c_int, [_P, _I32, _I32, _P]),
    "mgea_decoder_forward": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P]),
    "mgea_decoder_step": (C.c_int, [_P, _P, C.POINTER(SamplerConfig), _P, _P, _P]),
    "mgea_decoder_generate": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, C.POINTER(SamplerConfig), _P, _P]),
    "mgea_decoder_context_lengths": (C.c_int, [_P, _P, _P]),
    "mgea_decoder_stats": (C.c_int, [_P, C.POINTER(_I64)]),
    "mgea_decoder_error_flags": (C.c_int, [_P, C.POINTER(_I32), _P]),
    "mgea_bert_error_flags": (C.c_int, [_P, C.POINTER(_I32), _P]),
    "mgea_decoder_profile": (C.c_int, [_P, _I32]),
    "mgea_decoder_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_I64), _I32]),
    "mgea_bert_arena_layout": (C.c_int, [C.POINTER(BertConfig), C.POINTER(_I64), C.POINTER(_I32), C.POINTER(_I64)]),
    "mgea_bert_create": (C.c_int, [C.POINTER(BertConfig), _P, C.POINTER(_P)]),
    "mgea_bert_destroy": (C.c_int, [_P]),
    "mgea_bert_forward": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P, _P]),
    "mgea_bert_forward_packed": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "mgea_bert_stats": (C.c_int, [_P, C.POINTER(_I64)]),
    "mgea_lora_merge": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _F, _P]),
    "mgea_op_gemm_workspace_floats": (_I64, [_I32, _I32, _I32]),
    "mgea_op_gemm_f32": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "mgea_op_layernorm": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _F, _P]),
    "mgea_op_attention_f32": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "mgea_op_f32_to_bf16": (C.c_int, [_P, _P, _I64, _P]),
    "mgea_op_gemm_bf16": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "mgea_op_gemm_bf16_ln": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, C.POINTER(_I32), _P]),
    "mgea_op_ln_rowstat": (C.c_int, [_P, _P, _I32, _I32, _I32, _F, _P]),
    "mgea_op_fold_ln_bf16": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _P, _P, _P, _P]),
    "mgea_op_attention_bf16": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "mgea_op_layernorm_bf16": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _F, _P]),
    "mgea_op_tiled_weight_floats": (C.c_int64, [_I32, _I32]),
    "mgea_op_tile_weights": (C.c_int, [_P, _I32, _I32, _P, _P]),
    "mgea_op_tile_rows": (C.c_int, [_P, _P, _I32, _I32, _I32, _P]),
    "mgea_op_fold_ln": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _P, _P, _P, _P]),
    "mgea_op_skinny": (C.c_int, [_I32, _P, _P, _P, _P, _P, _I32, _I32, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "mgea_op_skinny_logits_partials": (C.c_int, [_I32, _I32, _I32]),
    "mgea_op_sample": (C.c_int, [_P, _I32, _I32, C.POINTER(SamplerConfig), _I64, _P, _P, _P]),
}

_lib = None


def load():
    """Load libmgea_hip.so (after torch, so both share torch's libamdhip64.so.7 by SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C music-generation-emotion-adaptive_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads the HIP runtime the extension must share)
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def tune_get(name: str) -> int:
    v = _I32(0)
    check(load().mgea_tune_get(name.encode(), C.byref(v)))
    return int(v.value)


def tune_set(name: str, value: int) -> int:
    """Set an A/B switch of the native library (tools/README.md); returns the previous value."""
    old = tune_get(name)
    check(load().mgea_tune_set(name.encode(), int(value)))
    return old


def last_error() -> str:
    return (load().mgea_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> int:
    """Map status codes to the exception classes the reference would raise."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == EINVAL:
        # the reference's shape failures are torch RuntimeErrors (api_cache.py:99)
        raise RuntimeError(msg)
    if rc == ECAPACITY:
        raise ValueError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    raise MgeaError(f"[{rc}] {msg}")


def ptr(t) -> C.c_void_p:
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def stream_ptr(stream=None) -> C.c_void_p:
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)
