"""ctypes binding of ``libveritasfi_hip.so`` (C ABI: ``include/veritasfi_hip.h``).

There is NO fallback: if the library is missing or fails to load, every entry point raises.  Build it
with ``python -m veritasfi_amd.build`` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VF_LIB_PATH") or os.path.join(_HERE, "lib", "libveritasfi_hip.so")   # VF_LIB_PATH: A/B builds

VF_OK = 0
VF_DTYPE_F32, VF_DTYPE_F16, VF_DTYPE_FP8_E4M3 = 0, 1, 2

c_i32, c_i64, c_f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
p_i32, p_i64, p_f32 = ctypes.POINTER(c_i32), ctypes.POINTER(c_i64), ctypes.POINTER(c_f32)
vp = ctypes.c_void_p


class EncoderConfig(ctypes.Structure):
    _fields_ = [
        ("vocab", c_i32), ("hidden", c_i32), ("layers", c_i32), ("heads", c_i32), ("ffn", c_i32), ("max_pos", c_i32),
        ("type_vocab", c_i32), ("roberta_pad_idx", c_i32), ("pooling", c_i32), ("normalize", c_i32), ("head", c_i32),
        ("ln_eps", c_f32),
    ]


class DecoderConfig(ctypes.Structure):
    _fields_ = [
        ("vocab", c_i32), ("hidden", c_i32), ("layers", c_i32), ("heads", c_i32), ("kv_heads", c_i32), ("head_dim", c_i32),
        ("ffn", c_i32), ("rope_theta", c_f32), ("rms_eps", c_f32), ("qk_norm", c_i32), ("pooling", c_i32),
        ("normalize", c_i32), ("head", c_i32), ("act", c_i32), ("norm_plus_one", c_i32), ("embed_scale", c_f32),
    ]


class VitConfig(ctypes.Structure):
    _fields_ = [
        ("image", c_i32), ("patch", c_i32), ("channels", c_i32), ("hidden", c_i32), ("layers", c_i32), ("heads", c_i32),
        ("ffn", c_i32), ("proj_dim", c_i32), ("act", c_i32), ("normalize", c_i32), ("ln_eps", c_f32),
    ]


class ClipTextConfig(ctypes.Structure):
    _fields_ = [
        ("vocab", c_i32), ("max_pos", c_i32), ("hidden", c_i32), ("layers", c_i32), ("heads", c_i32), ("ffn", c_i32),
        ("proj_dim", c_i32), ("act", c_i32), ("eos_token_id", c_i32), ("normalize", c_i32), ("ln_eps", c_f32),
    ]


class SearchStats(ctypes.Structure):
    _fields_ = [
        ("path", c_i64), ("n_queries", c_i64), ("candidates", c_i64), ("max_candidates", c_i64),
        ("uncertified", c_i64), ("overflowed", c_i64), ("exact_reruns", c_i64), ("wide_launches", c_i64), ("wide_queries", c_i64),
        ("aux_cus", c_i64), ("scans_overlap", c_i64), ("scan_kernel", c_i64), ("reserved", c_i64 * 4),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"}


# every symbol include/veritasfi_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "vf_version": (ctypes.c_int, []),
    "vf_last_error": (ctypes.c_char_p, []),
    "vf_device_count": (ctypes.c_int, [p_i32]),
    "vf_index_create": (ctypes.c_int, [ctypes.POINTER(vp), vp, c_i64, c_i32, c_i32, c_i32, c_i64]),
    "vf_index_create_device": (ctypes.c_int, [ctypes.POINTER(vp), vp, c_i64, c_i32, c_i32, c_i32, c_i64]),
    "vf_index_create_sharded": (ctypes.c_int, [ctypes.POINTER(vp), vp, c_i64, c_i32, c_i32, p_i32, c_i32]),
    "vf_index_create_sharded_from_file": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.c_char_p, p_i32, c_i32]),
    "vf_index_group": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.POINTER(vp), c_i32]),
    "vf_index_shards": (ctypes.c_int, [vp, p_i32, p_i32, c_i32]),
    "vf_index_peer_access": (ctypes.c_int, [vp, p_i32, c_i32, p_i32]),
    "vf_corpus_file_info": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "vf_index_create_from_file": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.c_char_p, c_i64, c_i64, c_i32, c_i64]),
    "vf_index_search": (ctypes.c_int, [vp, vp, c_i32, c_i32, vp, vp]),
    "vf_index_search_device": (ctypes.c_int, [vp, vp, c_i32, c_i32, vp, vp, vp]),
    "vf_index_slots": (ctypes.c_int, [vp, p_i32]),
    "vf_index_search_begin": (ctypes.c_int, [vp, c_i32, vp, c_i32, c_i32, vp, vp, vp]),
    "vf_index_search_end": (ctypes.c_int, [vp, c_i32]),
    "vf_index_info": (ctypes.c_int, [vp, p_i64, p_i32, p_i32, p_i32]),
    "vf_index_stats": (ctypes.c_int, [vp, ctypes.POINTER(SearchStats)]),
    "vf_index_set_option": (ctypes.c_int, [vp, ctypes.c_char_p, c_i64]),
    "vf_index_profile": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_double), p_i64, ctypes.POINTER(ctypes.c_double), p_i64]),
    "vf_index_profile_span": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_double), p_i64]),
    "vf_index_debug_read": (ctypes.c_int, [vp, c_i32, vp, c_i64]),
    "vf_index_destroy": (ctypes.c_int, [vp]),
    "vf_cosine_matrix": (ctypes.c_int, [vp, c_i32, c_i32, vp, c_i32]),
    "vf_cosine_matrix_rows": (ctypes.c_int, [vp, vp, c_i32, vp]),
    "vf_cosine_matrix_rows_mixed": (ctypes.c_int, [vp, vp, c_i32, vp, c_i32, vp]),
    "vf_cosine_scores": (ctypes.c_int, [vp, c_i32, vp, c_i64, c_i32, vp, c_i32]),
    "vf_merge_topk_device": (ctypes.c_int, [vp, vp, c_i32, c_i32, c_i32, vp, vp, c_i32, vp]),
    "vf_merge_topk_packed_device": (ctypes.c_int, [vp, c_i32, c_i32, c_i32, vp, vp, c_i32, vp]),
    "vf_fuse_rank": (ctypes.c_int, [vp, vp, c_i32, vp, vp, c_i32]),
    "vf_encoder_weight_sizes": (ctypes.c_int, [ctypes.POINTER(EncoderConfig), p_i64, p_i64]),
    "vf_encoder_create": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.POINTER(EncoderConfig), vp, c_i64, vp, c_i64, c_i32]),
    "vf_encoder_forward": (ctypes.c_int, [vp, vp, vp, vp, c_i32, c_i32, c_i32, vp]),
    "vf_encoder_forward_pooled": (ctypes.c_int, [vp, vp, vp, vp, c_i32, c_i32, c_i32, c_i32, c_i32, vp]),
    "vf_encoder_forward_hidden": (ctypes.c_int, [vp, vp, vp, vp, c_i32, c_i32, vp]),
    "vf_encoder_info": (ctypes.c_int, [vp, ctypes.POINTER(EncoderConfig)]),
    "vf_encoder_destroy": (ctypes.c_int, [vp]),
    "vf_reranker_create": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.POINTER(EncoderConfig), vp, c_i64, vp, c_i64, c_i32]),
    "vf_reranker_score": (ctypes.c_int, [vp, vp, vp, vp, c_i32, c_i32, vp]),
    "vf_reranker_destroy": (ctypes.c_int, [vp]),
    "vf_decoder_weight_sizes": (ctypes.c_int, [vp, p_i64, p_i64]),
    "vf_decoder_create": (ctypes.c_int, [ctypes.POINTER(vp), vp, vp, c_i64, vp, c_i64, c_i32]),
    "vf_decoder_forward": (ctypes.c_int, [vp, vp, vp, c_i32, c_i32, c_i32, vp]),
    "vf_decoder_forward_pooled": (ctypes.c_int, [vp, vp, vp, c_i32, c_i32, c_i32, c_i32, vp]),
    "vf_decoder_forward_hidden": (ctypes.c_int, [vp, vp, vp, c_i32, c_i32, vp]),
    "vf_decoder_destroy": (ctypes.c_int, [vp]),
    "vf_vit_weight_sizes": (ctypes.c_int, [ctypes.POINTER(VitConfig), p_i64, p_i64]),
    "vf_vit_create": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.POINTER(VitConfig), vp, c_i64, vp, c_i64, c_i32]),
    "vf_vit_forward": (ctypes.c_int, [vp, vp, c_i32, vp]),
    "vf_vit_forward_u8": (ctypes.c_int, [vp, vp, vp, vp, c_i32, vp]),
    "vf_vit_destroy": (ctypes.c_int, [vp]),
    "vf_clip_text_weight_sizes": (ctypes.c_int, [ctypes.POINTER(ClipTextConfig), p_i64, p_i64]),
    "vf_clip_text_create": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.POINTER(ClipTextConfig), vp, c_i64, vp, c_i64, c_i32]),
    "vf_clip_text_forward": (ctypes.c_int, [vp, vp, vp, c_i32, c_i32, vp]),
    "vf_clip_text_destroy": (ctypes.c_int, [vp]),
}

_lib = None
_lock = threading.Lock()


def lib():
    """Load the HIP library (once).  Raises if it is not built -- there is no CPU path."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: build it with `python -m veritasfi_amd.build` "
                        "(hipcc --offload-arch=gfx950). veritasfi_amd has no CPU fallback.")
                # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (same SONAME as
                # /opt/rocm's).  Load torch FIRST so our library binds to that copy; the other order
                # leaves torch unable to see the GPU, and device pointers / streams are shared anyway.
                try:
                    import torch  # noqa: F401
                except ImportError:
                    pass
                L = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(L, name)  # AttributeError if the ABI and this table drift apart
                    fn.restype = res
                    fn.argtypes = args
                _lib = L
    return _lib


def loaded() -> bool:
    """The library has been loaded in this process (an index, a model handle or a small dense op was created)."""
    return _lib is not None


def last_error() -> str:
    msg = lib().vf_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str = "") -> None:
    """Map a VF_E* code to RuntimeError(vf_last_error()), as SURVEY 8b prescribes."""
    if rc != VF_OK:
        raise RuntimeError(f"{what or 'veritasfi_hip'} failed (rc={rc}): {last_error()}")
