"""ctypes binding of libkosmosx_hip.so (C ABI declared in include/kosmosx_hip.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing or a call
fails, a RuntimeError is raised (after logging, mirroring the reference's log-and-re-raise
convention, /root/reference/kosmosx/model.py:233-235).
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libkosmosx_hip.so"
_lib = None
ABI_VERSION = 7   # KX_ABI_VERSION of include/kosmosx_hip.h

KX_PREC_BF16, KX_PREC_F32, KX_PREC_BF16X3, KX_PREC_F16C, KX_PREC_F16, KX_PREC_F32W24, KX_PREC_F32W16, KX_PREC_F16CHL = 0, 1, 2, 3, 4, 5, 6, 7
KX_F32, KX_BF16, KX_BF16X3, KX_F16C, KX_F16, KX_F16P, KX_F16HL = 0, 1, 2, 3, 4, 5, 6
KX_ACT_NONE, KX_ACT_GELU, KX_ACT_QUICK_GELU = 0, 1, 2
KX_ATTN_FULL, KX_ATTN_CAUSAL = 0, 1
KX_ACT_RELU, KX_ACT_SWISH = 5, 6
ACTS = {"none": KX_ACT_NONE, "gelu": KX_ACT_GELU, "quick_gelu": KX_ACT_QUICK_GELU, "relu": KX_ACT_RELU, "swish": KX_ACT_SWISH}
PRECS = {"bf16": KX_PREC_BF16, "fp32": KX_PREC_F32, "bf16x3": KX_PREC_BF16X3, "f16c": KX_PREC_F16C, "f16": KX_PREC_F16}
# model-level modes: the stage precisions above plus "mixed" — the error-budgeted mix (CLIP tower in plain fp16, Perceiver
# and decoder in f16c; kx_precision in include/kosmosx_hip.h, tools/precision_study.py --budget)
MODEL_PRECS = list(PRECS) + ["mixed"]
# internal pack name -> kx_precision: "w24" = fp32 operands rounded to 24 bits + 3-byte streaming copies (the decode step of f16c / mixed)
PACK_PRECS = dict(PRECS, w24=KX_PREC_F32, w16=KX_PREC_F32)     # w16: block-scaled int16 streaming copies (2.125 B / weight)


def stage_precision(prec: str, stage: str, widths=()) -> str:
    """The arithmetic a stage ("vit", "perceiver", "decoder") runs in under the model-level mode `prec`.
    `widths`: the K extents of the stage's GEMMs.  f16c operand rows are built from 128-element fp8 blocks, so a stage
    whose widths are multiples of 64 but not of 128 (dim = 192, 320, ...) cannot run it; under the DEFAULT mode ("mixed")
    such a stage runs bf16x3 — the other arithmetic inside the 1e-3 bound — instead of failing at the first forward
    (ADVICE r2).  An explicit "f16c" still raises (model._operand_f16c names the remedy)."""
    if prec == "mixed":
        if stage == "vit":
            return "f16"
        return "bf16x3" if any(int(k) % 128 for k in widths) else "f16c"
    return prec

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [("A", vp), ("lda", i64), ("W", vp), ("ldw", i64), ("C", vp), ("ldc", i64), ("cdt", i32),
                ("bias", vp), ("residual", vp), ("ldr", i64), ("M", i64), ("N", i64), ("K", i64),
                ("act", i32), ("qscale", f32), ("qcols", i64),
                ("xq_cs", vp), ("xq_ss", vp), ("xk_cs", vp), ("xk_ss", vp), ("xpos_T", i64), ("xpos_dim", i64),
                ("prec", i32), ("tile", i32), ("row_stats", vp), ("colsum", vp), ("stats_out", vp),
                ("splitk_ws", vp), ("splitk_ws_bytes", C.c_size_t), ("splitk", i32),
                ("ln_gamma", vp), ("ln_beta", vp), ("ln_eps", f32),
                ("stats_partials", vp), ("stats_in_nseg", i64), ("stats_in_seg", i64), ("stats_eps", f32),
                ("stats_out_seg", i32),
                ("ln_out", vp), ("ln_out_dt", i32), ("ln_out_gamma", vp), ("ln_out_beta", vp), ("ln_out_eps", f32),
                ("w_scale", vp), ("ln_operand_out", vp), ("ln_operand_dt", i32), ("ln_operand_stats", vp),
                ("w_tiled", i32), ("ksplit", i32), ("C2", vp), ("residual2", vp), ("a_add", vp),
                ("pair_ws", vp), ("pair_ws_bytes", C.c_size_t), ("row_stats_scratch", vp), ("f16c_corr", i32),
                ("splitk_flags", vp)]


class AttnArgs(C.Structure):
    _fields_ = [("q", vp), ("q_batch_stride", i64), ("q_row_stride", i64),
                ("k", vp), ("v", vp), ("kv_batch_stride", i64), ("kv_row_stride", i64),
                ("out", vp), ("out_batch_stride", i64), ("out_row_stride", i64), ("odt", i32),
                ("B", i64), ("H", i64), ("Tq", i64), ("Tk", i64), ("mask", i32), ("prec", i32), ("stats_out", vp),
                ("lse_out", vp), ("dropout_p", f32), ("dropout_site", i32), ("dropout_seed", C.c_uint64)]


class VitLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_g", "ln2_b",
                                  "w1", "b1", "w2", "b2", "wqkv_f", "bqkv_f", "wqkv_colsum", "w1_f", "b1_f", "w1_colsum")]


class _SizedWeights(C.Structure):
    """Weights structs open with the caller's sizeof() of the struct and of its per-layer element (header: "Binding
    safety"); the mirrors fill them in at construction so a mirror that falls behind the header fails with
    KX_ERR_INVALID_ARG instead of being walked with the wrong stride."""
    _layer_cls = None

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_bytes, self.layer_bytes = C.sizeof(type(self)), C.sizeof(self._layer_cls)


class VitWeights(_SizedWeights):
    _layer_cls = VitLayer
    _fields_ = [("struct_bytes", C.c_uint32), ("layer_bytes", C.c_uint32), ("image", i32), ("patch", i32), ("dim", i32), ("heads", i32), ("ffn", i32), ("layers", i32),
                ("act", i32), ("eps", f32), ("kpad", i32),
                ("wpatch", vp), ("cls", vp), ("pos", vp), ("pre_g", vp), ("pre_b", vp),
                ("layer", C.POINTER(VitLayer))]


class PerceiverLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("nm_g", "nm_b", "nl_g", "nl_b", "wq", "wkv", "wout", "ff_g", "ff_b", "w1", "w2")]


class PerceiverWeights(_SizedWeights):
    _layer_cls = PerceiverLayer
    _fields_ = [("struct_bytes", C.c_uint32), ("layer_bytes", C.c_uint32), ("dim", i32), ("depth", i32), ("heads", i32), ("latents", i32), ("ff_mult", i32), ("out_dim", i32),
                ("eps", f32), ("latents_p", vp), ("media_pos", vp), ("layer", C.POINTER(PerceiverLayer)),
                ("norm_g", vp), ("norm_b", vp), ("wproj", vp)]


class DecoderLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("sa_g", "sa_b", "wqkv", "bqkv", "wo", "bo", "wo_colsum", "fl_g", "fl_b",
                                  "w1", "b1", "w2", "b2", "w2_colsum", "wqkv_f", "bqkv_f", "wqkv_colsum",
                                  "w1_f", "b1_f", "w1_colsum", "wqkv_t", "wo_t", "w1_t", "w2_t")]


class DecoderWeights(_SizedWeights):
    _layer_cls = DecoderLayer
    _fields_ = [("struct_bytes", C.c_uint32), ("layer_bytes", C.c_uint32), ("layers", i32), ("dim", i32), ("heads", i32), ("ffn", i32), ("vocab", i32), ("act", i32),
                ("subln", i32), ("xpos", i32), ("eps", f32), ("layer", C.POINTER(DecoderLayer)),
                ("ln_g", vp), ("ln_b", vp), ("wout", vp), ("wout_f", vp), ("bout_f", vp), ("wout_colsum", vp),
                ("wout_t", vp)]


class ProfRecord(C.Structure):
    _fields_ = [("kind", i32), ("reserved", i32), ("a", i64), ("b", i64), ("c", i64), ("ms", f32),
                ("reserved2", f32)]


class ResamplePlan(C.Structure):
    _fields_ = [("crop", i32), ("hk", i32), ("vk", i32), ("y_first", i32), ("rows_needed", i32), ("x_first", i32),
                ("span_px", i32), ("hbounds", vp), ("hcoef", vp), ("vbounds", vp), ("vcoef", vp)]


KERNEL_KINDS = ["gemm_bf16_128x128", "gemm_bf16_64x64", "gemm_f32_128x128", "gemm_f32_64x64", "layernorm",
                "attn_bf16", "attn_f32", "embed", "misc", "gemm_bf16_160x128", "gemm_bf16_256x128_phased",
                "gemm_bf16_256x256_phased",
                "gemm_f16c_128x128", "gemm_f16c_64x64", "gemm_f16c_160x128", "gemm_f16c_256x128_phased", "gemm_f16c_256x256_phased",
                "gemm_f16_128x128", "gemm_f16_64x64", "gemm_f16_160x128", "gemm_f16_256x128_phased", "gemm_f16_256x256_phased",
                "attn_f16", "attn_f16s"]


# every symbol include/kosmosx_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "kx_version": (C.c_int, []),
    "kx_last_error": (C.c_int, [C.c_char_p, C.c_size_t]),
    "kx_struct_bytes": (C.c_size_t, [i32]),
    "kx_layernorm": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, i64, i64, f32, i64, i64, i64, vp]),
    "kx_gemm": (C.c_int, [C.POINTER(GemmArgs), vp]),
    "kx_pair_split_errors": (C.c_int, [C.POINTER(C.c_uint)]),
    "kx_attention": (C.c_int, [C.POINTER(AttnArgs), vp]),
    "kx_row_stats_finalize": (C.c_int, [vp, i64, i64, i64, f32, vp, vp]),
    "kx_embed_splice": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, i64, vp]),
    "kx_token_range": (C.c_int, [vp, i64, vp, vp]),
    "kx_set_tuning": (C.c_int, [C.c_int, C.c_int]),
    "kx_prof_enable": (C.c_int, [C.c_int]),
    "kx_prof_collect": (C.c_int, [C.POINTER(ProfRecord), C.c_int]),
    "kx_vit_workspace_bytes": (C.c_size_t, [C.POINTER(VitWeights), i64, i32]),
    "kx_vit_forward": (C.c_int, [C.POINTER(VitWeights), vp, i64, vp, vp, C.c_size_t, i32, vp]),
    "kx_perceiver_workspace_bytes": (C.c_size_t, [C.POINTER(PerceiverWeights), i64, i64, i32]),
    "kx_perceiver_forward": (C.c_int, [C.POINTER(PerceiverWeights), vp, i64, i64, vp, vp, vp, C.c_size_t, i32, vp]),
    "kx_decoder_workspace_bytes": (C.c_size_t, [C.POINTER(DecoderWeights), i64, i64, i32]),
    "kx_decoder_prefill": (C.c_int, [C.POINTER(DecoderWeights), vp, i64, i64, vp, vp, vp, vp, vp, i32, vp, vp, i64, vp,
                                     C.c_size_t, i32, vp]),
    "kx_decoder_decode_step": (C.c_int, [C.POINTER(DecoderWeights), vp, i64, i64, vp, vp, vp, vp, vp, vp, i64, vp, i32,
                                         vp, C.c_size_t, i32, vp]),
    "kx_attention_decode": (C.c_int, [vp, vp, vp, vp, i32, vp, i64, i64, i64, i64, i32, vp]),
    "kx_decoder_forward": (C.c_int, [C.POINTER(DecoderWeights), vp, i64, i64, vp, vp, vp, vp, vp, i32, vp,
                                     C.c_size_t, i32, vp]),
    "kx_clip_preprocess_workspace_bytes": (C.c_size_t, [i64, i32, i32]),
    "kx_clip_preprocess": (C.c_int, [vp, i64, i32, i32, i64, i64, C.POINTER(ResamplePlan), vp, vp, vp, vp,
                                     C.c_size_t, vp]),
    "kx_token_splice": (C.c_int, [vp, i64, i64, i64, i64, i64, i64, vp, vp, vp]),
    "kx_transpose": (C.c_int, [vp, vp, i64, i64, i64, i64, i32, vp]),
    "kx_to_operand": (C.c_int, [vp, vp, i64, i64, i64, i64, i32, i32, vp]),
    "kx_to_operand_pair_workspace_bytes": (C.c_size_t, [i64, i64]),
    "kx_to_operand_pair": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, i64, vp, vp, C.c_size_t, vp]),
    "kx_gelu_backward_operand_pair": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, i64, i64, vp, vp, C.c_size_t, vp]),
    "kx_colsum_workspace_bytes": (C.c_size_t, [i64, i64]),
    "kx_colsum": (C.c_int, [vp, i64, i64, i64, vp, i32, vp, C.c_size_t, vp]),
    "kx_layernorm_backward_workspace_bytes": (C.c_size_t, [i64, i64]),
    "kx_layernorm_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, i64, f32, vp, C.c_size_t, vp]),
    "kx_gelu_layernorm": (C.c_int, [vp, vp, vp, vp, C.c_int, i64, i64, f32, vp]),
    "kx_gelu_layernorm_backward_supported": (C.c_int, [i64]),
    "kx_gelu_layernorm_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, i64, f32, vp, C.c_size_t, vp]),
    "kx_gelu_forward": (C.c_int, [vp, vp, i64, vp]),
    "kx_gelu_backward": (C.c_int, [vp, vp, vp, i64, vp]),
    "kx_cross_entropy": (C.c_int, [vp, i64, i64, i64, vp, f32, vp, vp, i64, vp]),
    "kx_reduce_sum": (C.c_int, [vp, i64, i32, vp, i32, vp, C.c_size_t, vp]),
    "kx_xpos_backward": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, vp, f32, vp]),
    "kx_embed_backward": (C.c_int, [vp, vp, i64, i64, i64, i64, i64, vp, vp, vp]),
    "kx_adamw": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, vp, f32, vp]),
    "kx_lion": (C.c_int, [vp, vp, vp, i64, f32, f32, f32, f32, vp, f32, vp]),
    "kx_attention_backward_dropout": (C.c_int, [vp] * 10 + [i64] * 7 + [i32, f32, C.c_uint64, i32, vp]),
    "kx_attention_backward_dropout_bf16": (C.c_int, [vp] * 3 + [i32] + [vp] * 7 + [i64] * 7 + [i32, f32, C.c_uint64, i32, vp]),
    "kx_dropout": (C.c_int, [vp, vp, vp, i64, f32, C.c_uint64, i32, vp]),
    "kx_dropout_mask": (C.c_int, [vp, i64, f32, C.c_uint64, i32, vp]),
    "kx_quick_gelu_forward": (C.c_int, [vp, vp, i64, vp]),
    "kx_quick_gelu_backward": (C.c_int, [vp, vp, vp, i64, vp]),
    "kx_add_rowvec": (C.c_int, [vp, vp, vp, i64, i64, vp]),
    "kx_patchify": (C.c_int, [vp, vp, i64, i32, i32, i32, i32, vp]),
    "kx_vit_assemble": (C.c_int, [vp, vp, vp, vp, i64, i32, i32, vp]),
    "kx_attention_backward": (C.c_int, [vp, vp, vp, i32] + [vp] * 7 + [i64] * 7 + [i32, i32, vp]),
}


STRUCT_IDS = [GemmArgs, AttnArgs, VitLayer, VitWeights, PerceiverLayer, PerceiverWeights, DecoderLayer, DecoderWeights,
              ResamplePlan, ProfRecord]            # index = kx_struct_id


def lib_path() -> Path:
    return Path(os.environ.get("KOSMOSX_HIP_LIB", str(_LIB_PATH)))


def load():
    """Load the HIP library or fail loudly (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        msg = (f"libkosmosx_hip.so not found at {path}: build it with `python kosmos-x_amd/build.py` "
               "(hipcc --offload-arch=gfx950). The Kosmos-X MI355X path has no CPU fallback.")
        logging.error(msg)
        raise RuntimeError(msg)
    try:
        lib = C.CDLL(str(path))
    except OSError as e:  # missing ROCm runtime etc.
        logging.error(f"Failed to load {path}: {e}")
        raise RuntimeError(f"Failed to load {path}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.kx_version() != ABI_VERSION:
        raise RuntimeError(f"libkosmosx_hip.so ABI version {lib.kx_version()} != {ABI_VERSION}")
    for sid, cls in enumerate(STRUCT_IDS):             # kx_struct_id order
        if lib.kx_struct_bytes(sid) != C.sizeof(cls):
            raise RuntimeError(f"ctypes mirror {cls.__name__} is {C.sizeof(cls)} bytes, the library's struct "
                               f"{lib.kx_struct_bytes(sid)}: _hip.py is out of date with include/kosmosx_hip.h")
    # measurement hook: KOSMOSX_TUNING="14=1,4=8" applies kx_set_tuning(key, value) pairs once at load, so that a whole bench.py
    # run can be A/B-ed against a kernel variant without code changes (defaults — all keys 0 — are the shipped configuration)
    for kv in filter(None, os.environ.get("KOSMOSX_TUNING", "").split(",")):
        k, v = kv.split("=")
        if lib.kx_set_tuning(int(k), int(v)) != 0:
            raise RuntimeError(f"KOSMOSX_TUNING: kx_set_tuning({k}, {v}) refused")
        logging.warning(f"KOSMOSX_TUNING: tuning key {k} = {v} (not the shipped configuration)")
    _lib = lib
    return lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().kx_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc: int, what: str):
    if rc != 0:
        msg = f"{what} failed (kx_status {rc}): {last_error()}"
        logging.error(msg)
        raise RuntimeError(msg)


def prof_enable(on: bool):
    load().kx_prof_enable(int(on))


def prof_collect(max_records: int = 1 << 16) -> list:
    """[(kind_name, a, b, c, ms)] for every launch since prof_enable(True)."""
    buf = (ProfRecord * max_records)()
    n = load().kx_prof_collect(buf, max_records)
    return [(KERNEL_KINDS[buf[i].kind], buf[i].a, buf[i].b, buf[i].c, buf[i].ms) for i in range(n)]


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()
