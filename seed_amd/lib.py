"""ctypes binding of libseedmi.so (the C-ABI boundary declared in include/seedmi.h).

The product path has NO fallback: if the shared library is missing or the device is not a gfx950 the
import / first call raises.  PyTorch is used only as the owner of device memory and streams
(``tensor.data_ptr()``, ``torch.cuda.current_stream().cuda_stream``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEEDMI_LIB_PATH") or os.path.join(_HERE, "libseedmi.so")     # (override: A/B builds of the library)

ABI_VERSION = 4            # == SEEDMI_ABI_VERSION of include/seedmi.h; load() refuses a library built from another header

EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL, EPI_BIAS_TANH, EPI_SWIGLU, EPI_PATCH_EMBED, EPI_RELU = range(8)

_vp = C.c_void_p
_i = C.c_int


class VitLayer(C.Structure):
    _fields_ = [(n, _vp) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ln2_w", "ln2_b",
                                    "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                                    "qkv_wg", "qkv_cs", "qkv_bf", "fc1_wg", "fc1_cs", "fc1_bf")]     # LayerNorm fold (optional)


class GemmExt(C.Structure):
    _fields_ = [("ln_stats", _vp), ("ln_colsum", _vp), ("bias_f32", _vp), ("stats_out", _vp), ("stats_ld", _i),
                ("stats_by_tile", _i), ("ln_planes", _i), ("ln_ld", _i), ("ln_cols", _i), ("ln_eps", C.c_float)]


class QfLayer(C.Structure):
    _fields_ = ([(n, _vp) for n in ("qkv_w", "qkv_b", "ao_w", "ao_b", "ao_ln_w", "ao_ln_b")] + [("has_cross", _i)] +
                [(n, _vp) for n in ("cq_w", "cq_b", "ckv_w", "ckv_b", "co_w", "co_b", "co_ln_w", "co_ln_b",
                                    "ffn_w1", "ffn_b1", "ffn_w2", "ffn_b2", "ffn_ln_w", "ffn_ln_b")])


class TokenizerWeights(C.Structure):
    _fields_ = ([(n, _i) for n in ("img_size", "patch", "vit_dim", "vit_depth", "vit_heads", "vit_ffn", "qf_dim",
                                   "qf_layers", "qf_heads", "qf_ffn", "n_query", "n_embed", "code_dim", "kpad")] +
                [("patch_w", _vp), ("patch_b", _vp), ("pos_embed", _vp), ("cls_pos0", _vp),
                 ("vit", C.POINTER(VitLayer)), ("ln_vision_w", _vp), ("ln_vision_b", _vp), ("query_ln", _vp),
                 ("qf", C.POINTER(QfLayer)), ("head_w0", _vp), ("head_b0", _vp), ("head_w1", _vp), ("head_b1", _vp),
                 ("codebook", _vp), ("code_sqnorm", _vp)])


class DetokWeights(C.Structure):
    _fields_ = ([(n, _i) for n in ("n_embed", "code_dim", "code_pad", "dim", "heads", "ffn", "depth", "n_query", "down1", "down2",
                                   "down3", "out_dim")] +
                [(n, _vp) for n in ("codebook_pad", "dec_w0", "dec_b0", "dec_w1", "dec_b1", "pos_embed_image")] +
                [("blocks", C.POINTER(VitLayer))] +
                [(n, _vp) for n in ("down_w0", "down_w1", "down_w2", "distill_w", "distill_b")])


class TokenizerTaps(C.Structure):
    _fields_ = [("image_embeds", _vp), ("qformer_out", _vp), ("z", _vp)]


class ForkJoin(C.Structure):
    _fields_ = [("side_stream", _vp * 3), ("fork_event", _vp), ("join_event", _vp * 3), ("n_side", _i)]


class LlamaLayer(C.Structure):
    _fields_ = [(n, _vp) for n in ("ln1_w", "qkv_w", "o_w", "ln2_w", "gate_up_w", "down_w", "k_cache", "v_cache",
                                    "qkv_wp", "o_wp", "gate_up_wp", "down_wp")]


class LlamaWeights(C.Structure):
    _fields_ = ([(n, _i) for n in ("hidden", "layers", "heads", "ffn", "vocab", "vocab_pad", "max_pos", "tmax",
                                   "batch_cap")] + [("rms_eps", C.c_float)] +
                [("embed", _vp), ("layer", C.POINTER(LlamaLayer)), ("norm_w", _vp), ("lm_head", _vp),
                 ("cos_t", _vp), ("sin_t", _vp), ("lm_head_p", _vp), ("norm_folded", _i)])


# name -> (restype, argtypes); must list every symbol include/seedmi.h declares (tests check the export table)
SIGNATURES = {
    "seedmi_version": (_i, []),
    "seedmi_compute_dtype": (_i, []),
    "seedmi_last_error": (C.c_char_p, []),
    "seedmi_check_device": (_i, []),
    "seedmi_set_option": (_i, [C.c_char_p, _i]),
    "seedmi_gemm_bf16": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp]),
    "seedmi_gemm_workspace_bytes": (C.c_size_t, []),
    "seedmi_gemm_bf16_ext": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, C.POINTER(GemmExt), _vp, C.c_size_t,
                                  _vp]),
    "seedmi_gemm_tile_stats_supported": (_i, [_i, _i]),
    "seedmi_layernorm_stats_bf16": (_i, [_vp, _i, _i, _i, C.c_float, _vp, _vp]),
    "seedmi_layernorm_stats_finalize": (_i, [_vp, _i, _i, _i, _i, C.c_float, _vp, _vp]),
    "seedmi_gemm_bf16_ws": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, C.c_size_t, _vp]),
    "seedmi_layernorm_bf16": (_i, [_vp, _i, _vp, _vp, C.c_float, _vp, _i, _i, _i, _vp]),
    "seedmi_rmsnorm_bf16": (_i, [_vp, _i, _vp, C.c_float, _vp, _i, _i, _i, _vp]),
    "seedmi_im2col_patch": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "seedmi_fill_rows": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "seedmi_attention_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, C.c_float, _i, _i, _vp]),
    "seedmi_vq_code_sqnorm": (_i, [_vp, _vp, _i, _i, _vp]),
    "seedmi_vq_argmin_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "seedmi_vq_head_argmin_bf16": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "seedmi_embed_rows": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "seedmi_rope_kv_append": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "seedmi_add_i32": (_i, [_vp, _i, _vp]),
    "seedmi_add_i32_vec": (_i, [_vp, _vp, _i, _vp]),
    "seedmi_llama_decode_slots": (_i, [C.POINTER(LlamaWeights), _vp, _vp, _i, _vp, _i, _vp, C.c_size_t, _vp]),
    "seedmi_llama_attention_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, C.c_float, _i, _vp, _vp]),
    "seedmi_llama_decode_attention_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, C.c_float, _i, _vp,
                                                _i, _vp]),
    "seedmi_gemm_skinny_bf16": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp]),
    "seedmi_pack_skinny_weights_bytes": (C.c_size_t, [_i, _i]),
    "seedmi_pack_skinny_weights": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "seedmi_gemm_skinny_norm_bf16": (_i, [_i, _i, _i, _vp, _i, _vp, C.c_float, _vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "seedmi_gemm_skinny_workspace_bytes": (C.c_size_t, []),
    "seedmi_gemm_skinny_ws_status": (_i, [_vp, C.c_size_t, _vp]),
    "seedmi_gemm_skinny_workspace_init": (_i, [_vp, C.c_size_t, _vp]),
    "seedmi_llama_workspace_init": (_i, [_vp, C.c_size_t, _vp]),
    "seedmi_llama_decode_status": (_i, [C.POINTER(LlamaWeights), _i, _vp, C.c_size_t, _vp]),
    "seedmi_gemm_skinny_norm_ws_bf16": (_i, [_i, _i, _i, _vp, _i, _vp, C.c_float, _vp, _i, _i, _vp, _i, _i, _vp, _vp, C.c_size_t, _vp]),
    "seedmi_pack_activations_bf16": (_i, [_vp, _i, _vp, _i, _i, _vp]),
    "seedmi_gemm_skinny_packed_bf16": (_i, [_i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp]),
    "seedmi_rmsnorm_packed_bf16": (_i, [_vp, _i, _vp, C.c_float, _vp, _i, _i, _vp]),
    "seedmi_tokenize_workspace_bytes": (C.c_size_t, [C.POINTER(TokenizerWeights), _i]),
    "seedmi_tokenize": (_i, [C.POINTER(TokenizerWeights), _vp, _i, _i, _vp, C.POINTER(TokenizerTaps), _vp,
                             C.c_size_t, _vp]),
    "seedmi_tokenize_fj": (_i, [C.POINTER(TokenizerWeights), _vp, _i, _i, _vp, C.POINTER(TokenizerTaps), _vp,
                                C.c_size_t, C.POINTER(ForkJoin), _vp]),
    "seedmi_sample_token_bf16": (_i, [_vp, _i, _i, _i, C.c_float, C.c_float, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "seedmi_preprocess_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "seedmi_preprocess_image_u8": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        _vp, _i, _vp, _vp, C.c_size_t, _vp]),
    "seedmi_detokenize_workspace_bytes": (C.c_size_t, [C.POINTER(DetokWeights), _i]),
    "seedmi_detokenize": (_i, [C.POINTER(DetokWeights), _vp, _i, _vp, _vp, _vp, C.c_size_t, _vp]),
    "seedmi_llama_workspace_bytes": (C.c_size_t, [C.POINTER(LlamaWeights), _i, _i]),
    "seedmi_llama_forward": (_i, [C.POINTER(LlamaWeights), _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, C.c_size_t, _vp]),
    "seedmi_llama_forward_ex": (_i, [C.POINTER(LlamaWeights), _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, C.c_size_t, _vp]),
    "seedmi_llama_forward_io": (_i, [C.POINTER(LlamaWeights), _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, C.c_size_t,
                                     _vp]),
}


class SeedmiError(RuntimeError):
    pass


_lib = None
_lib_f16 = None
LIB_PATH_F16 = os.environ.get("SEEDMI_LIB_PATH_F16", os.path.join(_HERE, "libseedmi_f16.so"))


def _is_f16(dtype) -> bool:
    """torch.float16 / "fp16" / "float16" select the fp16 build; None, torch.bfloat16, "bf16" the default one."""
    if dtype is None:
        return False
    name = str(dtype).replace("torch.", "")
    if name in ("float16", "fp16", "half"):
        return True
    if name in ("bfloat16", "bf16"):
        return False
    raise SeedmiError(f"no build of the library computes in {dtype}: the 16-bit element is bfloat16 (libseedmi.so) or float16 (libseedmi_f16.so)")


def load(dtype=None):
    """Load the library that computes in ``dtype`` (default / torch.bfloat16: libseedmi.so; torch.float16: libseedmi_f16.so, the same
    sources built with -DSEEDMI_F16) and bind every declared symbol.  Raises if the library is not built."""
    global _lib, _lib_f16
    f16 = _is_f16(dtype)
    if f16 and _lib_f16 is not None:
        return _lib_f16
    if not f16 and _lib is not None:
        return _lib
    path = LIB_PATH_F16 if f16 else LIB_PATH
    lib = _load_path(path, 1 if f16 else 0)
    if f16:
        _lib_f16 = lib
    else:
        _lib = lib
    return lib


def _load_path(LIB_PATH, want_dtype):
    if not os.path.exists(LIB_PATH):
        raise SeedmiError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU / PyTorch fallback for the hot path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    got = lib.seedmi_version()
    if got != ABI_VERSION:
        raise SeedmiError(f"{LIB_PATH} reports C-ABI version {got}, this binding was written for {ABI_VERSION} (include/seedmi.h): "
                          "rebuild the library (`python -m seed_amd.build --force`); argument lists and struct layouts differ between versions")
    if lib.seedmi_compute_dtype() != want_dtype:
        raise SeedmiError(f"{LIB_PATH} computes in {'fp16' if lib.seedmi_compute_dtype() else 'bf16'}, expected {'fp16' if want_dtype else 'bf16'}")
    # tuning knobs for experiments: SEEDMI_OPTIONS="gemm_persist=0,gemm_group_m=4" (same keys as seedmi_set_option)
    for kv in filter(None, os.environ.get("SEEDMI_OPTIONS", "").split(",")):
        key, _, val = kv.partition("=")
        if lib.seedmi_set_option(key.strip().encode(), int(val)) != 0:
            raise SeedmiError(f"SEEDMI_OPTIONS: bad option {kv!r}: {lib.seedmi_last_error().decode(errors='replace')}")
    return lib


def check(rc, what="", lib=None):
    """``lib``: the library the failing call was made through (each build keeps its own error string); default: the bf16 build."""
    if rc != 0:
        msg = (lib or load()).seedmi_last_error().decode(errors="replace")
        raise SeedmiError(f"{what} failed with code {rc}: {msg}")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return C.c_void_p(0 if t is None else t.data_ptr())
