"""MI355X-native drop-in for ``kosmosx.model`` of kyegomez/Kosmos-X.

Same module path, class names, constructor signatures, attribute names and state_dict key
namespace as the reference (/root/reference/kosmosx/model.py:132-320), but every
tensor operation of the forward pass runs in hand-written gfx950 HIP kernels behind the C ABI of
``libkosmosx_hip.so`` (include/kosmosx_hip.h).  PyTorch is used for parameter storage, device
memory and the HIP stream only.  There is no CPU fallback: calling ``forward`` with CPU tensors or
without the built library raises.

Third-party classes the reference imports into this module's namespace (``Decoder``,
``DecoderConfig``, ``PositionalEmbedding``, ``PerceiverResampler``) are re-created here as
parameter containers with the upstream attribute/key names, so ``from kosmosx.model import
Decoder`` (/root/reference/train.py:44) keeps working.
"""
from __future__ import annotations

import ctypes as C
import logging
import math
import os

import torch
import torch.nn as nn

from . import _hip as H
from .config import DecoderConfig, KosmosConfig, PerceiverConfig, Switches, VitConfig

# the reference configures the root logger at import (/root/reference/kosmosx/model.py:8-10)
if not os.environ.get("KOSMOSX_NO_LOGGING_CONFIG"):
    logging.basicConfig(level=logging.DEBUG, format="%(asctime)s - %(levelname)s - %(message)s")

__all__ = ["Kosmos", "KosmosLanguage", "KosmosTokenizer", "Decoder", "DecoderConfig", "PositionalEmbedding",
           "PerceiverResampler", "CLIPVisionTower"]


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _prec_dtype(prec: str):
    return torch.float32 if prec in ("fp32", "w24", "w16") else torch.float16 if prec == "f16" else torch.bfloat16


def _operand(t: torch.Tensor, prec: str) -> torch.Tensor:
    """A weight matrix [N,K] as the GEMM operand of `prec`: bf16 / fp32 cast, or for "bf16x3" the split rows
    [hi(K) | lo(K) | hi(K)] in bf16 (hi = bf16(w), lo = bf16(w - hi)) that pair with [hi | hi | lo] activation rows —
    see kx_precision in include/kosmosx_hip.h."""
    t = t.detach()
    if prec == "f16c":
        return _operand_f16c(t.float())
    if prec == "w24":       # fp32 rounded to 16 significant bits (low mantissa byte zero): what the 24-bit streaming planes hold
        from .ops import round_to_24_bits
        return round_to_24_bits(t).contiguous()
    if prec == "w16":       # fp32 values (float)q * scale of the block-scaled int16 representation (the 16-bit streaming planes)
        from .ops import quantize_block16
        q, sc, wq = quantize_block16(t.float())
        wq._kx_w16_parts = (q, sc)          # the planes are built from these on the first decode step; they live and die with wq
        return wq
    if prec != "bf16x3":
        return t.to(_prec_dtype(prec)).contiguous()
    f = t.float()
    hi = f.to(torch.bfloat16)
    lo = (f - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo, hi], dim=1).contiguous()


def _operand_f16c(f: torch.Tensor) -> torch.Tensor:
    """[N,K] fp32 -> the packed KX_F16C weight matrix (kx_precision in include/kosmosx_hip.h): N rows of 4K bytes
    [h = fp16(w) | r = fp8((w - h) * 2^(s+11)) | e = fp8(w * 2^s)] followed by the N E8M0 scale bytes 127 - s, where s
    is the row's exponent (max|w| * 2^s in (64, 128]).  Returned as a flat uint8 tensor."""
    N, K = f.shape
    if K % 128:
        raise ValueError(f"f16c operands need K % 128 == 0 (K={K}): the DEFAULT precision 'mixed' runs such a stage in bf16x3 "
                         f"(fp32 when decoding incrementally) by itself (_hip.stage_precision); an explicit 'f16c' does not fall "
                         f"back — choose 'mixed', 'bf16x3' or 'fp32' for this model (model.precision / KOSMOSX_PRECISION)")
    amax = f.abs().amax(dim=1).clamp_min(2.0 ** -100)
    sexp = (7 - torch.ceil(torch.log2(amax))).clamp(-100, 100)          # integer-valued
    sc = torch.exp2(sexp)[:, None]
    h = f.clamp(-65504.0, 65504.0).to(torch.float16)         # saturates like every fp16 operand conversion (csrc: clamp_f16)
    r = ((f - h.float()) * sc * 2048.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    e = (f * sc).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    rows = torch.cat([h.view(torch.uint8).reshape(N, 2 * K), r.view(torch.uint8), e.view(torch.uint8)], dim=1)
    scale = (127 - sexp).to(torch.uint8)
    pad = (-N) % 16
    if pad:
        scale = torch.cat([scale, torch.full((pad,), 127, dtype=torch.uint8, device=f.device)])
    return torch.cat([rows.reshape(-1), scale]).contiguous()


def _operand_colsum(wp: torch.Tensor, prec: str, shape=None) -> torch.Tensor:
    """Sum over k of the values a packed operand row represents (what the folded-LayerNorm epilogue subtracts)."""
    if prec == "f16c":      # the value the GEMM reconstructs is the fp32 weight to 2^-15: h + 2^-(s+11) r
        N, K = shape
        rows = wp[: N * 4 * K].view(N, 4 * K)
        h = rows[:, : 2 * K].contiguous().view(torch.float16).float()
        r = rows[:, 2 * K: 3 * K].contiguous().view(torch.float8_e4m3fn).float()
        sexp = 127.0 - wp[N * 4 * K: N * 4 * K + N].float()
        return (h + r * torch.exp2(-(sexp + 11.0))[:, None]).sum(1)
    if prec != "bf16x3":
        return wp.float().sum(1)
    K = wp.shape[1] // 3
    return (wp[:, :K].float() + wp[:, K:2 * K].float()).sum(1)


FOLD_PRECS = ("bf16", "f16", "f16c")      # stage precisions whose pre-LayerNorms are folded into qkv / fc1 / logits


def _fold_pre_ln(stage: str = "") -> bool:
    """Opt-in (KOSMOSX_FOLD_PRE_LN=1).  Measured at B = 32 (HISTORY.md §4.2b): the 95 LayerNorm launches it removes cost
    1.5 ms, the residual epilogues' second store + lane exchanges, the consumers' row-statistics loads and the 96
    statistics-finalize launches it adds cost 2.3 ms — a 2 % loss on the headline and on C3, so it ships off."""
    v = os.environ.get("KOSMOSX_FOLD_PRE_LN", "0")        # "1": every stage; "vit" / "decoder": that stage only (A/B)
    return v == "1" or (v != "0" and v == stage)


def _fold_ln_linear(ln_w, ln_b, w, b, prec: str):
    """LayerNorm(gamma, beta) followed by Linear(W, b), folded (kx_decoder_layer in include/kosmosx_hip.h):
    W' = gamma ⊙ W as the operand of `prec`, b' = W·beta + b (fp32), colsum[n] = Σ_k W'[n,k] of the packed values."""
    wf, g_, b_ = w.detach().float(), ln_w.detach().float(), ln_b.detach().float()
    wp = _operand(wf * g_[None, :], prec)
    bias = wf @ b_ + (b.detach().float() if b is not None else 0.0)
    return wp, bias.contiguous(), _operand_colsum(wp, prec, tuple(wf.shape)).contiguous()


def _default_precision() -> str:
    # "mixed": the fastest arithmetic that holds the north star's 1e-3 on the logits (see kx_precision in the header)
    p = os.environ.get("KOSMOSX_PRECISION", "mixed")
    if p not in H.MODEL_PRECS:
        raise ValueError(f"KOSMOSX_PRECISION must be one of {H.MODEL_PRECS}")
    return p


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        msg = (f"{what} is on {t.device}: the Kosmos-X MI355X path runs on HIP devices only "
               "(no CPU fallback); move the model and inputs to 'cuda'")
        logging.error(msg)
        raise RuntimeError(msg)


class _Workspace:
    """Grow-only device scratch shared by the three stages (they run back to back on one stream).
    One buffer per HIP stream: forwards issued on different streams (micro-batch overlap) never share scratch."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes: int, device) -> torch.Tensor:
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = self.bufs[key] = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        return buf


class _PackedMixin:
    """Caches the operand-dtype copies / fused layouts of a module's weights per (device, precision).
    Invalidated when the module is moved (``_apply``) or a state_dict is loaded; call
    ``invalidate_packed()`` after mutating parameters in place."""

    def _packed_init(self):
        self._stream_src = {}
        self._packed = {}
        self._pack_gen = 0      # bumped on every invalidation: captured hipGraphs (raw pointers into `_packed`) check it

    def _drop_packed(self):
        self._packed = {}
        self._stream_src = {}
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1

    def invalidate_packed(self):
        self._drop_packed()
        for m in self.children():
            if isinstance(m, _PackedMixin):
                m.invalidate_packed()

    def _apply(self, fn, *a, **k):  # .to() / .cuda() / .float() ...
        out = super()._apply(fn, *a, **k)
        self._drop_packed()
        return out

    def _load_from_state_dict(self, *a, **k):
        self._drop_packed()
        return super()._load_from_state_dict(*a, **k)


_PINNED_RANGE = {}


def _begin_token_id_check(tokens: torch.Tensor, vocab: int):
    """IndexError for ids outside [0, vocab) — what F.embedding raises on the reference's CPU path; the kernels only
    clamp (memory safety).  One tiny reduction kernel + a 16-byte copy to pinned host memory are ENQUEUED here; the
    returned callable waits for that copy (an event, not the stream) and raises.  The caller enqueues its own launches
    in between, so the read-back does not leave the GPU idle: in the decode step the host stays one step ahead of the
    device instead of draining the stream every token.  Skipped while a hipGraph is being captured (the graphed forward
    validates the live inputs before it replays)."""
    if torch.cuda.is_current_stream_capturing():
        return lambda: None
    dev = tokens.device
    mm = torch.empty(2, dtype=torch.int64, device=dev)
    H.check(H.load().kx_token_range(tokens.data_ptr(), tokens.numel(), mm.data_ptr(), _stream()), "kx_token_range")
    pool = _PINNED_RANGE.setdefault(dev, [])         # free list of (pinned buffer, event); a check that is never finished
    slot = pool.pop() if pool else (torch.empty(2, dtype=torch.int64).pin_memory(), torch.cuda.Event())   # (an exception
    host, ev = slot                                  # between begin and finish) just drops its pair: nothing stays "busy"
    host.copy_(mm, non_blocking=True)
    ev.record(torch.cuda.current_stream(dev))

    def finish():
        ev.synchronize()
        lo, hi = host.tolist()
        if len(pool) < 8:
            pool.append(slot)
        if lo < 0 or hi >= vocab:
            msg = f"index out of range in self: token id {hi if hi >= vocab else lo} outside the {vocab}-row embedding table"
            logging.error(msg)
            raise IndexError(msg)
    return finish


def _validate_token_ids(tokens: torch.Tensor, vocab: int):
    """The check above, finished at once."""
    _begin_token_id_check(tokens, vocab)()


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------
# CLIP ViT-L/14 vision tower (HF CLIPVisionTransformer key namespace)
# ------------------------------------------------------------------------------------------------
class _CLIPEmbeddings(nn.Module):
    def __init__(self, c: VitConfig):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.empty(c.dim))
        self.patch_embedding = nn.Conv2d(3, c.dim, c.patch, c.patch, bias=False)
        self.position_embedding = nn.Embedding(c.tokens, c.dim)


class _CLIPAttention(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(d, d) for _ in range(4))


class _CLIPMLP(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(d, f), nn.Linear(f, d)


class _CLIPEncoderLayer(nn.Module):
    def __init__(self, c: VitConfig):
        super().__init__()
        self.self_attn = _CLIPAttention(c.dim)
        self.layer_norm1 = nn.LayerNorm(c.dim, eps=c.eps)
        self.mlp = _CLIPMLP(c.dim, c.ffn)
        self.layer_norm2 = nn.LayerNorm(c.dim, eps=c.eps)


class _CLIPEncoder(nn.Module):
    def __init__(self, c: VitConfig):
        super().__init__()
        self.layers = nn.ModuleList([_CLIPEncoderLayer(c) for _ in range(c.layers)])


class CLIPVisionTower(_PackedMixin, nn.Module):
    """``CLIPModel.from_pretrained(...).vision_model`` stand-in
    (/root/reference/kosmosx/model.py:154-156).  ``tower(pixel_values=x)["last_hidden_state"]``."""

    def __init__(self, c: VitConfig):
        super().__init__()
        self._packed_init()
        self.cfg = c
        self.embeddings = _CLIPEmbeddings(c)
        self.pre_layrnorm = nn.LayerNorm(c.dim, eps=c.eps)  # (sic) HF key name
        self.encoder = _CLIPEncoder(c)
        self.post_layernorm = nn.LayerNorm(c.dim, eps=c.eps)

    def _pack(self, prec: str):
        dev = self.pre_layrnorm.weight.device
        key = (dev, prec)
        if key in self._packed:
            return self._packed[key]
        c, dt = self.cfg, _prec_dtype(prec)
        keep = []

        def op(t):  # GEMM operand in the compute dtype
            t = _operand(t, prec); keep.append(t); return t.data_ptr()

        def v(t):   # fp32 vector / table
            t = _f32(t); keep.append(t); return t.data_ptr()

        kreal = 3 * c.patch * c.patch
        kpad = (kreal + 127) // 128 * 128      # whole 128-byte K-tiles in every operand format
        wp = torch.zeros((c.dim, kpad), dtype=torch.float32, device=dev)
        wp[:, :kreal] = self.embeddings.patch_embedding.weight.detach().reshape(c.dim, kreal)
        layers = (H.VitLayer * c.layers)()
        for i, L in enumerate(self.encoder.layers):
            a = L.self_attn
            wqkv = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0)
            bqkv = torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0)
            e = layers[i]
            e.ln1_g, e.ln1_b = v(L.layer_norm1.weight), v(L.layer_norm1.bias)
            e.wqkv, e.bqkv = op(wqkv), v(bqkv)
            e.wo, e.bo = op(a.out_proj.weight), v(a.out_proj.bias)
            e.ln2_g, e.ln2_b = v(L.layer_norm2.weight), v(L.layer_norm2.bias)
            e.w1, e.b1 = op(L.mlp.fc1.weight), v(L.mlp.fc1.bias)
            e.w2, e.b2 = op(L.mlp.fc2.weight), v(L.mlp.fc2.bias)
            if prec in FOLD_PRECS and _fold_pre_ln("vit"):      # layer_norm1 -> qkv, layer_norm2 -> fc1
                for dst, ln, wt, bt in (("wqkv", L.layer_norm1, wqkv, bqkv), ("w1", L.layer_norm2, L.mlp.fc1.weight, L.mlp.fc1.bias)):
                    t3 = _fold_ln_linear(ln.weight, ln.bias, wt, bt, prec)
                    keep.extend(t3)
                    setattr(e, dst + "_f", t3[0].data_ptr())
                    setattr(e, "b" + dst[1:] + "_f", t3[1].data_ptr())
                    setattr(e, dst + "_colsum", t3[2].data_ptr())
        w = H.VitWeights()
        w.image, w.patch, w.dim, w.heads, w.ffn, w.layers = c.image, c.patch, c.dim, c.heads, c.ffn, c.layers
        w.act, w.eps, w.kpad = H.ACTS[c.act], c.eps, kpad
        w.wpatch = op(wp)
        w.cls, w.pos = v(self.embeddings.class_embedding), v(self.embeddings.position_embedding.weight)
        w.pre_g, w.pre_b = v(self.pre_layrnorm.weight), v(self.pre_layrnorm.bias)
        w.layer = C.cast(layers, C.POINTER(H.VitLayer))
        self._packed[key] = (w, layers, keep)
        return self._packed[key]

    def run(self, pixels: torch.Tensor, prec: str, ws: _Workspace) -> torch.Tensor:
        prec = H.stage_precision(prec, "vit")
        """pixels [B,3,H,W] any real dtype (HF casts to the weight dtype; SURVEY H2) -> [B,tokens,dim] fp32."""
        _require_cuda(pixels, "images")
        c = self.cfg
        if pixels.dim() != 4 or pixels.shape[1] != 3 or pixels.shape[2] != c.image or pixels.shape[3] != c.image:
            raise ValueError(f"Input image size ({pixels.shape[-2]}*{pixels.shape[-1]}) doesn't match model "
                             f"({c.image}*{c.image}).")
        w, _, _ = self._pack(prec)
        lib = H.load()
        px = pixels.to(torch.float32).contiguous()
        B = px.shape[0]
        out = torch.empty((B, c.tokens, c.dim), dtype=torch.float32, device=px.device)
        need = lib.kx_vit_workspace_bytes(C.byref(w), B, H.PRECS[prec])
        buf = ws.get(need, px.device)
        H.check(lib.kx_vit_forward(C.byref(w), px.data_ptr(), B, out.data_ptr(), buf.data_ptr(), buf.numel(),
                                   H.PRECS[prec], _stream()), "kx_vit_forward")
        return out

    def forward(self, pixel_values: torch.Tensor = None, **kwargs):
        from . import ops
        h = self.run(pixel_values, getattr(self, "precision", _default_precision()), _Workspace())
        pooled = ops.layernorm(h[:, 0, :].contiguous(), _f32(self.post_layernorm.weight),
                               _f32(self.post_layernorm.bias), self.cfg.eps)
        return {"last_hidden_state": h, "pooler_output": pooled}


# ------------------------------------------------------------------------------------------------
# PerceiverResampler (lucidrains flamingo-pytorch key namespace)
# ------------------------------------------------------------------------------------------------
class _PerceiverAttention(nn.Module):
    def __init__(self, dim, dim_head, heads, eps):
        super().__init__()
        inner = dim_head * heads
        self.norm_media = nn.LayerNorm(dim, eps=eps)
        self.norm_latents = nn.LayerNorm(dim, eps=eps)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)


def _feed_forward(dim, mult, eps):
    return nn.Sequential(nn.LayerNorm(dim, eps=eps), nn.Linear(dim, dim * mult, bias=False), nn.GELU(),
                         nn.Linear(dim * mult, dim, bias=False))


class PerceiverResampler(_PackedMixin, nn.Module):
    """flamingo_pytorch.PerceiverResampler stand-in (/root/reference/kosmosx/model.py:196-203)."""

    def __init__(self, *, dim, depth, dim_head=64, heads=8, num_latents=64, num_media_embeds=4, ff_mult=4,
                 eps=1e-5, switches: Switches | None = None):
        super().__init__()
        self._packed_init()
        if dim_head != 64:
            raise ValueError("the gfx950 attention kernels are specialised for dim_head == 64")
        self.cfg = PerceiverConfig(dim, depth, dim_head, heads, num_latents, num_media_embeds, ff_mult, eps)
        self.switches = switches or Switches()
        self.latents = nn.Parameter(torch.empty(num_latents, dim))
        self.media_pos_emb = nn.Parameter(torch.empty(num_media_embeds, 1, dim))
        self.layers = nn.ModuleList([
            nn.ModuleList([_PerceiverAttention(dim, dim_head, heads, eps), _feed_forward(dim, ff_mult, eps)])
            for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=eps)

    def _gemm_widths(self):
        c = self.cfg
        return (c.dim, c.dim_head * c.heads, c.dim * c.ff_mult)

    def _pack(self, prec: str, image_proj: torch.Tensor | None):
        dev = self.latents.device
        key = (dev, prec, None if image_proj is None else image_proj.data_ptr())
        if key in self._packed:
            return self._packed[key]
        c, dt = self.cfg, _prec_dtype(prec)
        if not self.switches.u6_media_pos_first_only:
            raise NotImplementedError("per-token media_pos_emb is not a behaviour of the reference path")
        keep = []

        def op(t):
            t = _operand(t, prec); keep.append(t); return t.data_ptr()

        def v(t):
            t = _f32(t); keep.append(t); return t.data_ptr()

        inner = c.heads * c.dim_head
        layers = (H.PerceiverLayer * c.depth)()
        for i, (att, ff) in enumerate(self.layers):
            e = layers[i]
            e.nm_g, e.nm_b = v(att.norm_media.weight), v(att.norm_media.bias)
            e.nl_g, e.nl_b = v(att.norm_latents.weight), v(att.norm_latents.bias)
            wkv = att.to_kv.weight.detach()
            if not self.switches.u6_kv_k_first:
                wkv = torch.cat([wkv[inner:], wkv[:inner]], 0)
            e.wq, e.wkv, e.wout = op(att.to_q.weight), op(wkv), op(att.to_out.weight)
            e.ff_g, e.ff_b = v(ff[0].weight), v(ff[0].bias)
            e.w1, e.w2 = op(ff[1].weight), op(ff[3].weight)
        w = H.PerceiverWeights()
        w.dim, w.depth, w.heads, w.latents, w.ff_mult, w.eps = c.dim, c.depth, c.heads, c.latents, c.ff_mult, c.eps
        w.out_dim = 0 if image_proj is None else image_proj.shape[0]
        w.latents_p, w.media_pos = v(self.latents), v(self.media_pos_emb[0, 0])
        w.layer = C.cast(layers, C.POINTER(H.PerceiverLayer))
        w.norm_g, w.norm_b = v(self.norm.weight), v(self.norm.bias)
        w.wproj = 0 if image_proj is None else op(image_proj)
        self._packed[key] = (w, layers, keep)
        return self._packed[key]

    def run(self, x: torch.Tensor, prec: str, ws: _Workspace, image_proj: torch.Tensor | None = None,
            want_latents: bool = False):
        """x [B,m,dim] fp32 -> (projected [B,latents,out_dim] or None, latents [B,latents,dim] or None)."""
        _require_cuda(x, "media")
        prec = H.stage_precision(prec, "perceiver", self._gemm_widths())
        w, _, _ = self._pack(prec, image_proj)
        lib = H.load()
        x = x.to(torch.float32).contiguous()
        B, m, _ = x.shape
        c = self.cfg
        out = (torch.empty((B, c.latents, image_proj.shape[0]), dtype=torch.float32, device=x.device)
               if image_proj is not None else None)
        lat = torch.empty((B, c.latents, c.dim), dtype=torch.float32, device=x.device) if want_latents else None
        need = lib.kx_perceiver_workspace_bytes(C.byref(w), B, m, H.PRECS[prec])
        buf = ws.get(need, x.device)
        H.check(lib.kx_perceiver_forward(C.byref(w), x.data_ptr(), B, m, H.ptr(out), H.ptr(lat), buf.data_ptr(),
                                         buf.numel(), H.PRECS[prec], _stream()), "kx_perceiver_forward")
        return out, lat

    def forward(self, x: torch.Tensor):
        if x.dim() == 4:
            if x.shape[1] != 1:
                raise NotImplementedError("only a single media time-step is on the Kosmos-X path")
            x = x[:, 0]
        _, lat = self.run(x, getattr(self, "precision", _default_precision()), _Workspace(), None, True)
        return lat[:, None]  # b 1 n d, as upstream; the caller squeezes (/root/reference/kosmosx/model.py:231)


# ------------------------------------------------------------------------------------------------
# torchscale Decoder (Magneto sub-LN, XPos, multiway "A" branch) key namespace
# ------------------------------------------------------------------------------------------------
class PositionalEmbedding(nn.Embedding):
    """torchscale.component.embedding.PositionalEmbedding: learned table, fairseq positions start at 2."""

    def forward(self, x, positions=None, **kwargs):
        raise RuntimeError("PositionalEmbedding is a parameter container here; positions are added inside "
                           "the fused kx_embed_splice kernel")


class MultiwayNetwork(nn.Module):
    """torchscale MultiwayNetwork with split_position == -1: forward == A.  The B copy is dead weight on
    this path and is never allocated on the device.  state_dict round trip: a checkpoint whose B tensors differ from A
    (torchscale initialises B = deepcopy(A), training can move them apart) keeps them — parked on the host in
    `_b_store`, re-emitted unchanged on save; without stored tensors B is emitted as an alias of A."""

    def __init__(self, module: nn.Module):
        super().__init__()
        self.A = module
        self.split_position = -1
        self._b_store = {}
        self._register_state_dict_hook(MultiwayNetwork._emit_b)
        self._register_load_state_dict_pre_hook(MultiwayNetwork._keep_b, with_module=True)

    @staticmethod
    def _emit_b(module, state_dict, prefix, local_metadata):
        for k in [k for k in state_dict if k.startswith(prefix + "A.")]:
            suffix = k[len(prefix) + 2:]
            state_dict[prefix + "B." + suffix] = module._b_store.get(suffix, state_dict[k])

    @staticmethod
    def _keep_b(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        store = {}
        for k in [k for k in state_dict if k.startswith(prefix + "B.")]:
            suffix = k[len(prefix) + 2:]
            v, a = state_dict.pop(k), state_dict.get(prefix + "A." + suffix)
            if a is None or v.shape != a.shape or not torch.equal(v.detach().cpu(), a.detach().cpu()):
                store[suffix] = v.detach().to("cpu").clone()
        module._b_store = store


def _mw(multiway: bool, module: nn.Module) -> nn.Module:
    return MultiwayNetwork(module) if multiway else module


def _a(module: nn.Module) -> nn.Module:
    return module.A if isinstance(module, MultiwayNetwork) else module


class XPOS(nn.Module):
    def __init__(self, head_dim, scale_base=512):
        super().__init__()
        self.head_dim, self.scale_base = head_dim, scale_base
        self.register_buffer("scale", (torch.arange(0, head_dim, 2) + 0.4 * head_dim) / (1.4 * head_dim))

    def _closed_form(self, first_abs: int, n_rows: int, min_pos: int, downscale: bool):
        """Rows for absolute positions a = first_abs .. first_abs + n_rows - 1, from the DEFINITION rather than from
        torchscale's tensor program (the CPU test reference restates that program; this is a separate derivation so that
        a slip in either shows up as a difference, tests/test_abi.py::test_xpos_tables_closed_form):
            table[a, j] = cos|sin(theta) * zeta_j ** (+-(a + min_pos) / scale_base),   zeta_j = (2j + 0.4 hd) / (1.4 hd)
        The scale is evaluated in float64 as exp(e * log(zeta_j)).  The angle keeps the reference's fp32 semantics —
        theta = fp32(a) * fp32(1 / 10000 ** (j / (hd/2))), what torchscale's fixed_pos_embedding feeds sin/cos: at
        a ~ 2000 the fp32 product is 1e-4 rad away from the real-number angle, and the reference's logits include that
        — and cos / sin of that fp32 angle are taken in float64 and rounded once."""
        import numpy as np
        hd, half = self.head_dim, self.head_dim // 2
        j = np.arange(half, dtype=np.float64)
        log_zeta = np.log((2.0 * j + 0.4 * hd) / (1.4 * hd))
        a = np.arange(first_abs, first_abs + n_rows, dtype=np.float64)
        e = (a + float(min_pos)) / float(self.scale_base)
        scale = np.exp((-e if downscale else e)[:, None] * log_zeta[None, :]).astype(np.float32)
        inv_freq = (1.0 / (10000 ** (torch.arange(0, half) / half))).numpy()              # fp32, the definition itself
        theta = (a.astype(np.float32)[:, None] * inv_freq[None, :]).astype(np.float64)     # fp32 product, exactly
        cos, sin = np.cos(theta).astype(np.float32), np.sin(theta).astype(np.float32)
        return torch.from_numpy(cos * scale).contiguous(), torch.from_numpy(sin * scale).contiguous()

    def tables_centred(self, n_pos: int, centre_len: int, downscale: bool = False):
        """Rows for absolute positions 0..n_pos-1 with the centring constant of a `centre_len`-token sequence
        (min_pos = -(centre_len)//2).  For n_pos == centre_len this is `tables(centre_len)`; incremental decoding
        keeps the prefill's centring for every later position (the constant cancels in q·k)."""
        return self._closed_form(0, n_pos, -(centre_len) // 2, downscale)

    def tables(self, length: int, offset: int = 0, downscale: bool = False):
        """(cos*scale, sin*scale) [length, head_dim/2] fp32 — what torchscale's XPOS.forward + fixed_pos_embedding
        produce for a `length`-token call at `offset` (rows = absolute positions offset .. offset+length-1, exponent
        centred with min_pos = -(length + offset) // 2, Python floor division)."""
        return self._closed_form(offset, length, -(length + offset) // 2, downscale)


class MultiheadAttention(nn.Module):
    def __init__(self, args: DecoderConfig):
        super().__init__()
        d, mw = args.decoder_embed_dim, args.multiway
        self.embed_dim, self.num_heads = d, args.decoder_attention_heads
        self.head_dim = d // self.num_heads
        self.scaling = self.head_dim ** -0.5
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (_mw(mw, nn.Linear(d, d, bias=True)) for _ in range(4))
        self.inner_attn_ln = _mw(mw, nn.LayerNorm(d, eps=args.layernorm_eps)) if args.subln else None
        self.xpos = XPOS(self.head_dim, args.xpos_scale_base) if args.xpos_rel_pos else None


class FeedForwardNetwork(nn.Module):
    def __init__(self, args: DecoderConfig):
        super().__init__()
        d, f = args.decoder_embed_dim, args.decoder_ffn_embed_dim
        self.fc1, self.fc2 = nn.Linear(d, f), nn.Linear(f, d)
        self.ffn_layernorm = nn.LayerNorm(f, eps=args.layernorm_eps) if args.subln else None


class DecoderLayer(nn.Module):
    def __init__(self, args: DecoderConfig):
        super().__init__()
        d, mw = args.decoder_embed_dim, args.multiway
        self.self_attn = MultiheadAttention(args)
        self.self_attn_layer_norm = _mw(mw, nn.LayerNorm(d, eps=args.layernorm_eps))
        self.ffn = _mw(mw, FeedForwardNetwork(args))
        self.final_layer_norm = _mw(mw, nn.LayerNorm(d, eps=args.layernorm_eps))
        self.alpha = 1.0


def _warn_train_mode(module: nn.Module) -> None:
    """SURVEY H1 / VERDICT r4 missing #5: the reference builds its modules with dropout = attention_dropout = 0.1
    (/root/reference/kosmosx/model.py:175,177) and example.py never calls .eval(), so ITS forward is stochastic in train mode.
    This forward is the deterministic (eval) arithmetic whatever `self.training` says — the train-mode arithmetic lives in
    LanguageModelTrainer / KosmosTrainer(train_mode=True).  Say so once per module instead of silently returning eval numbers."""
    if module.training and not getattr(module, "_kx_warned_train_mode", False):
        module._kx_warned_train_mode = True
        logging.warning(f"{type(module).__name__}.forward called in train mode: this path runs the deterministic (eval) forward — "
                        "the reference's dropout / attention_dropout (0.1) are NOT applied here.  Call .eval() to make that "
                        "explicit, or use kosmosx.training.*Trainer(train_mode=True) for the stochastic training arithmetic.")



class Decoder(_PackedMixin, nn.Module):
    """torchscale.architecture.decoder.Decoder stand-in with the ``passed_x`` patch
    (/root/reference/README.md:179-193)."""

    def __init__(self, args: DecoderConfig, embed_tokens=None, embed_positions=None, output_projection=None,
                 switches: Switches | None = None, **kwargs):
        super().__init__()
        self._packed_init()
        if args.decoder_embed_dim != 64 * args.decoder_attention_heads:
            raise ValueError("the gfx950 attention kernels are specialised for head_dim == 64")
        if args.activation_fn not in ("gelu", "relu", "swish"):
            # torchscale's get_activation_fn knows exactly these three (the reference's ctor smoke builds "relu" and "swish"
            # models: /root/reference/tests/test_kosmos_lang.py:17-66).  gelu is the reference path and the tuned one; relu /
            # swish run the generic 128 x 128 GEMM kernel for fc1 (kx_act in the header) — forward only
            raise NotImplementedError(f"activation_fn={args.activation_fn!r}: torchscale knows 'gelu', 'relu' and 'swish'")
        self.args = args
        self.switches = switches or Switches()
        self.embed_scale = 1.0 if args.no_scale_embedding else math.sqrt(args.decoder_embed_dim)
        self.embed_tokens, self.embed_positions, self.output_projection = embed_tokens, embed_positions, output_projection
        self.layers = nn.ModuleList([DecoderLayer(args) for _ in range(args.decoder_layers)])
        self.num_layers = len(self.layers)
        self.layer_norm = nn.LayerNorm(args.decoder_embed_dim, eps=args.layernorm_eps)  # subln => normalize_before
        self._xpos_cache = {}
        self._ws = _Workspace()

    def _gemm_widths(self):
        return (self.args.decoder_embed_dim, self.args.decoder_ffn_embed_dim)

    # -- weights ----------------------------------------------------------------------------------
    def _pack(self, prec: str):
        dev = self.layer_norm.weight.device
        key = (dev, prec)
        if key in self._packed:
            return self._packed[key]
        a, dt = self.args, _prec_dtype(prec)
        keep = []

        def op(t):
            t = _operand(t, prec); keep.append(t); return t.data_ptr()

        def v(t):
            t = _f32(t); keep.append(t); return t.data_ptr()

        stream_src = []     # (layer index or -1, field, packed operand): what _pack_decode_tiles re-tiles for the decode step

        def fold(ln, lin):
            """Fold a sub-LayerNorm into the Linear that consumes it (see kx_decoder_layer in the header):
            W' = γ ⊙ W cast to the operand dtype, b' = W·β + b, colsum = Σ_k W'[n,k] of the cast values."""
            wf, g_, b_ = lin.weight.detach().float(), ln.weight.detach().float(), ln.bias.detach().float()
            wp = _operand(wf * g_[None, :], prec)
            shp = tuple(wf.shape)
            keep.append(wp)
            return wp.data_ptr(), v(wf @ b_ + lin.bias.detach().float()), v(_operand_colsum(wp, prec, shp))

        def src(i, field, ptr):   # remember the packed operand behind `ptr` (bf16 / fp32: the decode step's precisions)
            if prec in ("bf16", "fp32", "w24", "w16"):
                stream_src.append((i, field, next(t for t in reversed(keep) if t.data_ptr() == ptr)))

        layers = (H.DecoderLayer * self.num_layers)()
        for i, L in enumerate(self.layers):
            sa, ffn = L.self_attn, _a(L.ffn)
            q, k, vv, o = _a(sa.q_proj), _a(sa.k_proj), _a(sa.v_proj), _a(sa.out_proj)
            e = layers[i]
            e.sa_g, e.sa_b = v(_a(L.self_attn_layer_norm).weight), v(_a(L.self_attn_layer_norm).bias)
            e.wqkv = op(torch.cat([q.weight, k.weight, vv.weight], 0))
            src(i, "wqkv_t", e.wqkv)
            e.bqkv = v(torch.cat([q.bias, k.bias, vv.bias], 0))
            if a.subln:
                e.wo, e.bo, e.wo_colsum = fold(_a(sa.inner_attn_ln), o)
                e.w2, e.b2, e.w2_colsum = fold(ffn.ffn_layernorm, ffn.fc2)
            else:
                e.wo, e.bo = op(o.weight), v(o.bias)
                e.w2, e.b2 = op(ffn.fc2.weight), v(ffn.fc2.bias)
            e.fl_g, e.fl_b = v(_a(L.final_layer_norm).weight), v(_a(L.final_layer_norm).bias)
            e.w1, e.b1 = op(ffn.fc1.weight), v(ffn.fc1.bias)
            src(i, "wo_t", e.wo); src(i, "w2_t", e.w2); src(i, "w1_t", e.w1)
            if prec in FOLD_PRECS and _fold_pre_ln("decoder"):      # self_attn_layer_norm -> qkv, final_layer_norm -> fc1
                sl, fl = _a(L.self_attn_layer_norm), _a(L.final_layer_norm)
                t3 = _fold_ln_linear(sl.weight, sl.bias, torch.cat([q.weight, k.weight, vv.weight], 0),
                                     torch.cat([q.bias, k.bias, vv.bias], 0), prec)
                keep.extend(t3)
                e.wqkv_f, e.bqkv_f, e.wqkv_colsum = (t.data_ptr() for t in t3)
                t3 = _fold_ln_linear(fl.weight, fl.bias, ffn.fc1.weight, ffn.fc1.bias, prec)
                keep.extend(t3)
                e.w1_f, e.b1_f, e.w1_colsum = (t.data_ptr() for t in t3)
        w = H.DecoderWeights()
        w.layers, w.dim, w.heads, w.ffn = self.num_layers, a.decoder_embed_dim, a.decoder_attention_heads, a.decoder_ffn_embed_dim
        w.vocab, w.act = self.output_projection.weight.shape[0], H.ACTS[a.activation_fn]
        w.subln, w.xpos, w.eps = int(a.subln), int(a.xpos_rel_pos), a.layernorm_eps
        w.layer = C.cast(layers, C.POINTER(H.DecoderLayer))
        w.ln_g, w.ln_b = v(self.layer_norm.weight), v(self.layer_norm.bias)
        w.wout = op(self.output_projection.weight)
        src(-1, "wout_t", w.wout)
        if prec in FOLD_PRECS and _fold_pre_ln("decoder"):          # decoder.layer_norm -> output_projection
            t3 = _fold_ln_linear(self.layer_norm.weight, self.layer_norm.bias, self.output_projection.weight, None, prec)
            keep.extend(t3)
            w.wout_f, w.bout_f, w.wout_colsum = (t.data_ptr() for t in t3)
        emb, pos = _f32(self.embed_tokens.weight), _f32(self.embed_positions.weight)
        keep += [emb, pos]
        self._packed[key] = (w, layers, keep, emb, pos)
        self._stream_src[key] = stream_src
        return self._packed[key]

    def _pack_decode_tiles(self, prec: str):
        """The decode step streams every weight once: a second copy of the bf16 operands in the STREAMING layout
        (kx_gemm_args.w_tiled: one contiguous 1 KB block per wave instruction instead of 16 row segments of 64 B; 4.2-5.0 vs
        3.2-3.7 TB/s) is made on the first decode step and hung into the packed weight structs (+2.6 GB at full size).
        KOSMOSX_DECODE_TILED=0 keeps the row-major operands."""
        key = (self.layer_norm.weight.device, prec)
        todo = self._stream_src.pop(key, None)
        if not todo or os.environ.get("KOSMOSX_DECODE_TILED", "1") == "0":
            return
        from .ops import quantize_block16, tile_weight_rows, tile_weight_rows_w16, tile_weight_rows_w24
        w, layers, keep, _, _ = self._packed[key]
        for i, field, t in todo:
            if t.shape[1] % 32:
                if prec in ("w24", "w16"):
                    raise ValueError("the compressed streaming planes need K % 32 == 0")   # (decoder widths are multiples of 64)
                continue
            if prec == "w16":       # t holds (float)q * scale; q and scale were kept when it was packed
                q, sc = getattr(t, "_kx_w16_parts", (None, None))
                if q is None:
                    q, sc, wq = quantize_block16(t)
                    if not torch.equal(wq, t):
                        raise RuntimeError("w16 operand without its (q, scale) parts")
                else:
                    del t._kx_w16_parts                     # 2 B / weight: only the planes are needed from here on
                tt = tile_weight_rows_w16(q, sc)
            else:
                tt = tile_weight_rows_w24(t) if prec == "w24" else tile_weight_rows(t)
            keep.append(tt)
            setattr(w if i < 0 else layers[i], field, tt.data_ptr())

    def _xpos_tables(self, T: int, device):
        key = (T, device)
        if key not in self._xpos_cache:
            x = self.layers[0].self_attn.xpos
            qc, qs = x.tables(T, 0, False)
            kc, ks = x.tables(T, 0, True)
            self._xpos_cache[key] = tuple(t.to(device) for t in (qc, qs, kc, ks))
        return self._xpos_cache[key]

    # -- stages -----------------------------------------------------------------------------------
    def embed(self, tokens: torch.Tensor | None, prec: str, img: torch.Tensor | None = None, splice_at: int = 2,
              alias: bool | None = None, pos_offset: int = 0, defer_check: bool = False) -> torch.Tensor:
        """Fused forward_embedding/cat/forward_embedding of /root/reference/kosmosx/model.py:238-244
        (img given) or the single forward_embedding of :319 (img None).  The token-id range check is read back after the
        embedding launch; with ``defer_check`` the caller finishes it (``self._finish_check()``) after enqueuing more."""
        prec = H.stage_precision(prec, "decoder", self._gemm_widths())
        _, _, _, emb, pos = self._pack(prec)
        lib = H.load()
        check = None
        if tokens is not None:
            _require_cuda(tokens, "text_tokens")
            if tokens.dtype != torch.int64:
                tokens = tokens.long()  # F.embedding accepts int32/int64 indices
            tokens = tokens.contiguous()
            B, Tt = tokens.shape
            if getattr(self, "validate_token_ids", True) and tokens.numel():
                check = _begin_token_id_check(tokens, emb.shape[0])
        else:
            B, Tt = img.shape[0], 0
        n_img = 0 if img is None else img.shape[1]
        d = emb.shape[1]
        out = torch.empty((B, Tt + n_img, d), dtype=torch.float32, device=emb.device)
        alias = self.switches.u1_inplace_alias if alias is None else alias
        rc = lib.kx_embed_splice(H.ptr(tokens), emb.data_ptr(), pos.data_ptr(), H.ptr(img), out.data_ptr(), B, Tt,
                                 n_img, d, emb.shape[0], pos.shape[0], splice_at, int(alias), pos_offset, _stream())
        if rc == 1 and "out of range" in H.last_error():
            msg = H.last_error()
            logging.error(msg)
            raise IndexError("index out of range in self: " + msg)  # what F.embedding raises upstream
        H.check(rc, "kx_embed_splice")
        self._pending_check = check
        if not defer_check:
            self._finish_check()
        return out

    def _finish_check(self):
        check, self._pending_check = getattr(self, "_pending_check", None), None
        if check is not None:
            check()

    def run(self, x: torch.Tensor, prec: str, logits_dtype=torch.float32) -> torch.Tensor:
        """x [B,T,dim] fp32 residual stream (CONSUMED) -> logits [B,T,vocab]."""
        prec = H.stage_precision(prec, "decoder", self._gemm_widths())
        w, _, _, _, _ = self._pack(prec)
        lib = H.load()
        B, T, _ = x.shape
        tabs = self._xpos_tables(T, x.device) if self.args.xpos_rel_pos else (None,) * 4
        logits = torch.empty((B, T, w.vocab), dtype=logits_dtype, device=x.device)
        need = lib.kx_decoder_workspace_bytes(C.byref(w), B, T, H.PRECS[prec])
        buf = self._ws.get(need, x.device)
        ldt = H.KX_F32 if logits_dtype == torch.float32 else H.KX_BF16
        H.check(lib.kx_decoder_forward(C.byref(w), x.data_ptr(), B, T, *(H.ptr(t) for t in tabs), logits.data_ptr(),
                                       ldt, buf.data_ptr(), buf.numel(), H.PRECS[prec], _stream()),
                "kx_decoder_forward")
        return logits

    # -- incremental decoding ---------------------------------------------------
    def _forward_incremental(self, tokens, state: dict, passed_x, prec: str) -> torch.Tensor:
        """torchscale's incremental_state protocol: the first call runs the whole prefix and fills the KV cache,
        every later call is given the token history (only its last token and its length are used, as upstream's
        `tokens[:, -1:]`) and appends one position.  ``state`` is an opaque dict owned by the caller."""
        if self.args.activation_fn != "gelu":
            raise NotImplementedError(f"incremental decoding with activation_fn={self.args.activation_fn!r}: the weight-streaming "
                                      "decode kernels offer gelu only (kx_act in include/kosmosx_hip.h)")
        model_prec = prec
        prec = H.stage_precision(prec, "decoder", self._gemm_widths())
        if prec == "bf16x3":
            # the cache kernels are offered in bf16, fp32 and f16c.  Under the default mode the bf16x3 stage fallback (widths
            # that are not multiples of 128) decodes in fp32 — the other arithmetic inside the tolerance that
            # the prefill / step kernels implement; an explicit bf16x3 is refused here, not by a C error three calls down.
            if model_prec != "mixed":
                raise ValueError("incremental decoding is offered in bf16, fp32, f16c and mixed; precision 'bf16x3' has no "
                                 "KV-cache kernels (use 'fp32' for the same accuracy class)")
            prec = "fp32"
        w, _, _, emb, pos = self._pack(prec)
        lib = H.load()
        D, L = self.args.decoder_embed_dim, self.num_layers
        if "len" not in state:                                             # ---- first step: prefill ----
            if passed_x is not None:
                _require_cuda(passed_x, "passed_x")
                x = passed_x.to(torch.float32).clone(memory_format=torch.contiguous_format)
            else:
                x = self.embed(tokens, prec)
            B, T, _ = x.shape
            Tmax = int(state.get("max_len", pos.shape[0] - 2))
            if T > Tmax:
                raise IndexError(f"index out of range in self: {T} tokens exceed the {Tmax}-row cache")
            dt = torch.float32 if prec in ("fp32", "f16c") else _prec_dtype(prec)   # q/k/v-typed cache
            nh = D // 64                                       # opaque to the caller; the kernels keep [heads][Tmax][64] per sequence
            state["kcache"] = torch.empty((L, B, nh, Tmax, 64), dtype=dt, device=x.device)
            state["vcache"] = torch.empty((L, B, nh, Tmax, 64), dtype=dt, device=x.device)
            xp = self.layers[0].self_attn.xpos
            if xp is not None:
                q = xp.tables_centred(Tmax, T, False)
                k = xp.tables_centred(Tmax, T, True)
                state["xpos"] = tuple(t.to(x.device) for t in (*q, *k))
            else:
                state["xpos"] = (None,) * 4
            logits = torch.empty((B, T, w.vocab), dtype=torch.float32, device=x.device)
            need = lib.kx_decoder_workspace_bytes(C.byref(w), B, T, H.PRECS[prec])
            buf = self._ws.get(need, x.device)
            H.check(lib.kx_decoder_prefill(C.byref(w), x.data_ptr(), B, T, *(H.ptr(t) for t in state["xpos"]),
                                           logits.data_ptr(), H.KX_F32, state["kcache"].data_ptr(),
                                           state["vcache"].data_ptr(), Tmax, buf.data_ptr(), buf.numel(),
                                           H.PRECS[prec], _stream()), "kx_decoder_prefill")
            state.update(len=T, max_len=Tmax, batch=B, prec=prec)
            return logits
        t, Tmax, B = state["len"], state["max_len"], state["batch"]           # ---- later steps: one token ----
        if state["prec"] != prec:
            raise RuntimeError("precision changed between incremental steps")
        if t >= Tmax or t + 2 >= pos.shape[0]:
            raise IndexError(f"index out of range in self: position {t + 2} exceeds the table / cache")
        # The arithmetic of a decode step.  A step is weight streaming — what it costs is bytes — and an f16c weight row is
        # 4 bytes per value, exactly what the fp32 weight is: the f16c step therefore runs on the fp32 operands with the
        # exact-f32 MFMA (same bytes, no compensation needed; its q/k/v cache is fp32 already), i.e. every precision that
        # holds the north star's tolerance decodes with fp32 products.  KOSMOSX_DECODE_EXACT=0 keeps the f16c tile GEMMs (A/B).
        # ... and since the f16c weights themselves carry 15-16 bits, the weights of that step are rounded to 16 significant
        # bits and streamed as THREE bytes each ("w24": kx_gemm_args.w_tiled = 2; the activations and the products stay fp32).
        # KOSMOSX_DECODE_EXACT=fp32 streams the full fp32 weights (4 bytes), =0 keeps the f16c tile GEMMs.
        exact = os.environ.get("KOSMOSX_DECODE_EXACT", "1")
        # the compressed forms (w24 / w16) exist to be STREAMED: with more than 16 sequences no streaming launch exists and
        # they would run plain fp32 GEMMs on rounded weights — accuracy lost, no byte saved, and a second fp32 pack held
        #  — so those steps keep the f16c tile GEMMs the prefill ran.  (KOSMOSX_DECODE_TILED=0 is the A/B switch
        # that runs the SAME compressed weights from their row-major fp32 operands: bit-identical to the planes, test-only.)
        streams = B <= 16
        sprec = prec if (prec != "f16c" or exact == "0" or not (streams or exact == "fp32")) else (
            {"fp32": "fp32", "w24": "w24"}.get(exact, "w16"))
        if sprec != prec:
            w = self._pack(sprec)[0]
        if sprec in ("bf16", "fp32", "w24", "w16") and B <= 16:
            self._pack_decode_tiles(sprec)                 # first decode step: the streaming copy of the weights
        if passed_x is not None:
            _require_cuda(passed_x, "passed_x")
            x = passed_x[:, -1:].to(torch.float32).clone(memory_format=torch.contiguous_format)
        else:
            x = self.embed(tokens[:, -1:], prec, pos_offset=t, defer_check=True)
        if x.shape[0] != B:
            raise ValueError("batch size changed between incremental steps")
        rows = tuple(None if tb is None else tb[t] for tb in state["xpos"])   # views: row t of each [Tmax, 32] table
        logits = torch.empty((B, 1, w.vocab), dtype=torch.float32, device=x.device)
        # (w24 without streaming copies — KOSMOSX_DECODE_TILED=0, more than 16 sequences — is plain fp32 on the rounded operands)
        pid = ({"w24": H.KX_PREC_F32W24, "w16": H.KX_PREC_F32W16}[sprec] if (sprec in ("w24", "w16") and bool(w.wout_t))
               else H.PACK_PRECS[sprec])
        need = lib.kx_decoder_workspace_bytes(C.byref(w), B, 1, pid)
        buf = self._ws.get(need, x.device)
        H.check(lib.kx_decoder_decode_step(C.byref(w), x.data_ptr(), B, t, *(H.ptr(r) for r in rows),
                                           state["kcache"].data_ptr(), state["vcache"].data_ptr(), Tmax,
                                           logits.data_ptr(), H.KX_F32, buf.data_ptr(), buf.numel(), pid,
                                           _stream()), "kx_decoder_decode_step")
        self._finish_check()                               # an IndexError leaves the state where it was (row t is rewritten)
        state["len"] = t + 1
        return logits

    # -- torchscale-compatible surface ------------------------------------------------------------
    def forward_embedding(self, tokens, token_embedding=None, incremental_state=None):
        if incremental_state is not None:
            raise NotImplementedError("incremental decoding is not on the reference forward path (SURVEY §8f)")
        prec = getattr(self, "precision", _default_precision())
        if token_embedding is None:
            x = self.embed(tokens, prec)
            return x, (x if self.switches.u1_inplace_alias else None)
        _require_cuda(token_embedding, "token_embedding")
        x = self.embed(None, prec, img=token_embedding.to(torch.float32).contiguous(), splice_at=0)
        return x, x

    def forward(self, prev_output_tokens, self_attn_padding_mask=None, encoder_out=None, incremental_state=None,
                features_only=False, return_all_hiddens=False, token_embeddings=None, **kwargs):
        if encoder_out is not None or self_attn_padding_mask is not None or features_only:
            raise NotImplementedError("only the decoder-only logits path of the reference is built")
        prec = getattr(self, "precision", _default_precision())
        passed_x = kwargs.get("passed_x", None)
        if incremental_state is not None:
            return self._forward_incremental(prev_output_tokens, incremental_state, passed_x, prec), \
                {"inner_states": None, "l_aux": None, "attn": None}
        if passed_x is None:
            x, _ = self.forward_embedding(prev_output_tokens, token_embeddings)
        else:
            _require_cuda(passed_x, "passed_x")
            x = passed_x.to(torch.float32).clone(memory_format=torch.contiguous_format)  # run() consumes it
        return self.run(x, prec), {"inner_states": None, "l_aux": None, "attn": None}


# ------------------------------------------------------------------------------------------------
# initialisation (a1): the reference's distributions, seeded
# ------------------------------------------------------------------------------------------------
def _normal(t, std, g):
    with torch.no_grad():
        t.normal_(0.0, std, generator=g)


def _uniform(t, bound, g):
    with torch.no_grad():
        t.uniform_(-bound, bound, generator=g)


def _xavier_uniform(t, gain, g):
    fan_out, fan_in = t.shape[0], t.shape[1]
    _uniform(t, gain * math.sqrt(6.0 / (fan_in + fan_out)), g)


def _linear_default(lin: nn.Linear, g):  # nn.Linear.reset_parameters: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(in))
    b = 1.0 / math.sqrt(lin.in_features)
    _uniform(lin.weight, b, g)
    if lin.bias is not None:
        _uniform(lin.bias, b, g)


def _ln_default(ln: nn.LayerNorm):
    with torch.no_grad():
        ln.weight.fill_(1.0)
        ln.bias.zero_()


def init_vit_(m: CLIPVisionTower, g):
    """HF CLIPPreTrainedModel._init_weights (factor 1, initializer_range 0.02).  The reference loads
    pretrained weights instead (/root/reference/kosmosx/model.py:154-156); there is no network here."""
    c = m.cfg
    _normal(m.embeddings.class_embedding, c.dim ** -0.5, g)
    _normal(m.embeddings.patch_embedding.weight, 0.02, g)
    _normal(m.embeddings.position_embedding.weight, 0.02, g)
    in_std = (c.dim ** -0.5) * ((2 * c.layers) ** -0.5)
    out_std = c.dim ** -0.5
    fc_std = (2 * c.dim) ** -0.5
    for L in m.encoder.layers:
        for p in (L.self_attn.q_proj, L.self_attn.k_proj, L.self_attn.v_proj):
            _normal(p.weight, in_std, g)
        _normal(L.self_attn.out_proj.weight, out_std, g)
        _normal(L.mlp.fc1.weight, fc_std, g)
        _normal(L.mlp.fc2.weight, in_std, g)
        with torch.no_grad():
            for p in (L.self_attn.q_proj, L.self_attn.k_proj, L.self_attn.v_proj, L.self_attn.out_proj, L.mlp.fc1,
                      L.mlp.fc2):
                p.bias.zero_()
        _ln_default(L.layer_norm1); _ln_default(L.layer_norm2)
    _ln_default(m.pre_layrnorm); _ln_default(m.post_layernorm)


def init_perceiver_(m: PerceiverResampler, g):
    _normal(m.latents, 1.0, g)
    _normal(m.media_pos_emb, 1.0, g)
    for att, ff in m.layers:
        _ln_default(att.norm_media); _ln_default(att.norm_latents)
        for lin in (att.to_q, att.to_kv, att.to_out, ff[1], ff[3]):
            _linear_default(lin, g)
        _ln_default(ff[0])
    _ln_default(m.norm)


def init_decoder_(m: Decoder, g):
    """torchscale MultiheadAttention.reset_parameters / FeedForwardNetwork.reset_parameters and the subln
    init scale sqrt(log(2*layers)) applied to fc1, fc2, out_proj, v_proj (weights and biases)."""
    a = m.args
    init_scale = math.sqrt(math.log(a.decoder_layers * 2)) if a.subln else 1.0
    for L in m.layers:
        sa, ffn = L.self_attn, _a(L.ffn)
        for p in (sa.q_proj, sa.k_proj, sa.v_proj):
            _xavier_uniform(_a(p).weight, 1 / math.sqrt(2), g)
            _uniform(_a(p).bias, 1.0 / math.sqrt(a.decoder_embed_dim), g)
        _xavier_uniform(_a(sa.out_proj).weight, 1.0, g)
        with torch.no_grad():
            _a(sa.out_proj).bias.zero_()
        _linear_default(ffn.fc1, g); _linear_default(ffn.fc2, g)
        with torch.no_grad():
            for p in (_a(sa.v_proj), _a(sa.out_proj), ffn.fc1, ffn.fc2):
                p.weight.mul_(init_scale)
                p.bias.mul_(init_scale)
        for ln in (_a(L.self_attn_layer_norm), _a(L.final_layer_norm)):
            _ln_default(ln)
        if a.subln:
            _ln_default(_a(sa.inner_attn_ln)); _ln_default(ffn.ffn_layernorm)
    _ln_default(m.layer_norm)


def init_embeddings_(embed: nn.Embedding, pos: nn.Embedding, g, padding_idx=1):
    _xavier_uniform(embed.weight, 1.0, g)   # bitsandbytes Embedding.reset_parameters
    _normal(pos.weight, 1.0, g)             # nn.Embedding default
    with torch.no_grad():
        embed.weight[padding_idx].zero_()
        pos.weight[padding_idx].zero_()


def perturb_(module: nn.Module, g, amount: float):
    """Test helper: make every LayerNorm gain/bias and every Linear bias non-trivial so that parity tests
    cannot pass with a forgotten bias or affine term."""
    with torch.no_grad():
        for mod in module.modules():
            if isinstance(mod, nn.LayerNorm):
                mod.weight.add_(torch.randn(mod.weight.shape, generator=g) * amount)
                mod.bias.add_(torch.randn(mod.bias.shape, generator=g) * amount)
            elif isinstance(mod, nn.Linear) and mod.bias is not None:
                mod.bias.add_(torch.randn(mod.bias.shape, generator=g) * amount * 0.5)


# ------------------------------------------------------------------------------------------------
# the drop-in classes
# ------------------------------------------------------------------------------------------------
class Kosmos(nn.Module):
    """The main Kosmos model class (/root/reference/kosmosx/model.py:132-253).

    Attributes (same names as the reference): clip_model, embed, embed_positions, output_projection,
    config, decoder, perceive, image_proj.
    """

    def __init__(self):
        super().__init__()
        self._build(KosmosConfig(decoder=DecoderConfig(vocab_size=64007)), seed=None, switches=None)

    @classmethod
    def _from_config(cls, cfg: KosmosConfig, seed: int | None = None, switches: Switches | None = None,
                     perturb: float = 0.0):
        """Reduced-size / seeded construction for tests and benchmarks (``Kosmos()`` itself takes no
        arguments, as in the reference)."""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self._build(cfg, seed, switches, perturb)
        return self

    def _build(self, cfg: KosmosConfig, seed, switches, perturb: float = 0.0):
        self.cfg = cfg
        self.switches = switches or Switches()
        self.precision = _default_precision()
        d = cfg.decoder.decoder_embed_dim
        try:
            self.clip_model = CLIPVisionTower(cfg.vit)
        except Exception as e:
            logging.error(f"Failed to initialize CLIP model: {e}")
            raise
        self.embed = nn.Embedding(cfg.vocab, d, padding_idx=cfg.padding_idx)   # bitsandbytes Embedding == F.embedding
        self.embed_positions = PositionalEmbedding(cfg.max_positions, d, cfg.padding_idx)
        self.output_projection = nn.Linear(d, cfg.vocab, bias=False)
        self.config = cfg.decoder
        try:
            self.decoder = Decoder(self.config, embed_tokens=self.embed, embed_positions=self.embed_positions,
                                   output_projection=self.output_projection, switches=self.switches)
        except Exception as e:
            logging.error(f"Failed to initialize Decoder: {e}")
            raise
        p = cfg.perceiver
        self.perceive = PerceiverResampler(dim=p.dim, depth=p.depth, dim_head=p.dim_head, heads=p.heads,
                                           num_latents=p.latents, num_media_embeds=p.media_embeds,
                                           ff_mult=p.ff_mult, eps=p.eps, switches=self.switches)
        self.image_proj = nn.Linear(p.dim, d, bias=False)
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        init_vit_(self.clip_model, g)
        init_perceiver_(self.perceive, g)
        init_embeddings_(self.embed, self.embed_positions, g, cfg.padding_idx)
        _normal(self.output_projection.weight, d ** -0.5, g)   # /root/reference/kosmosx/model.py:166-167
        init_decoder_(self.decoder, g)
        _normal(self.image_proj.weight, d ** -0.5, g)          # :205-206
        if perturb:
            perturb_(self, g, perturb)
        self._ws = _Workspace()
        self.use_hip_graphs = os.environ.get("KOSMOSX_HIP_GRAPHS", "0") == "1"
        self._graphs = {}

    def invalidate_packed(self):
        self._graphs = {}
        for m in (self.clip_model, self.perceive, self.decoder):
            m.invalidate_packed()

    def _apply(self, fn, *a, **k):
        self._graphs = {}
        return super()._apply(fn, *a, **k)

    def forward(self, text_tokens: torch.Tensor, images: torch.Tensor, **kwargs):
        """text_tokens [B,Tt] integer ids, images [B,3,224,224] any real dtype
        -> logits [B, Tt+64, vocab] fp32.  kwargs are ignored, as in the reference."""
        if not isinstance(text_tokens, torch.Tensor) or not isinstance(images, torch.Tensor):
            raise TypeError("text_tokens and images must be instances of torch.Tensor")
        _warn_train_mode(self)
        if self.use_hip_graphs and text_tokens.is_cuda and images.is_cuda and text_tokens.dim() == 2:
            return self._forward_graphed(text_tokens, images)
        return self._forward_impl(text_tokens, images)

    def _forward_graphed(self, text_tokens, images):
        """Replay the ~420 kernel launches of one forward as a single hipGraph (the library never allocates or
        synchronises, so the whole launch sequence is capturable).  Worth it when the forward is launch-bound
        (small batches): one graph per (batch, text length, precision), inputs copied into static buffers."""
        # one graph per issuing stream as well: graphs replayed concurrently on different streams (independent
        # requests) must not share static buffers or scratch — each capture runs on its own stream, which is what keys
        # the library workspace (_Workspace), and allocates from its own graph memory pool
        key = (tuple(text_tokens.shape), tuple(images.shape), images.dtype, self.precision, text_tokens.device,
               torch.cuda.current_stream(text_tokens.device).cuda_stream)
        ent = self._graphs.get(key)
        gen = (self.clip_model._pack_gen, self.perceive._pack_gen, self.decoder._pack_gen)
        if ent is not None and ent[5] != gen:       # weights re-packed since capture (load_state_dict, .to(), an
            ent = None                              # in-place update + invalidate_packed): the graph's pointers are stale
        if getattr(self.decoder, "validate_token_ids", True) and text_tokens.numel():
            _validate_token_ids(text_tokens.long().contiguous(), self.embed.weight.shape[0])
        if ent is None:
            self._forward_impl(text_tokens, images)          # warm-up: packs weights, sizes workspaces, uploads tables
            torch.cuda.synchronize(text_tokens.device)
            s_tok, s_img = text_tokens.clone(), images.clone()
            g = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream(device=text_tokens.device)
            with torch.cuda.graph(g, stream=cap):
                s_out = self._forward_impl(s_tok, s_img)
            gen = (self.clip_model._pack_gen, self.perceive._pack_gen, self.decoder._pack_gen)
            ent = self._graphs[key] = (g, s_tok, s_img, s_out, cap, gen)
        g, s_tok, s_img, s_out = ent[:4]
        s_tok.copy_(text_tokens)
        s_img.copy_(images)
        g.replay()
        return s_out.clone()

    def _forward_impl(self, text_tokens: torch.Tensor, images: torch.Tensor):
        prec = self.precision
        try:
            img = self.clip_model.run(images, prec, self._ws)                       # model.py:230
            img, _ = self.perceive.run(img, prec, self._ws, self.image_proj.weight)  # :231-232
        except Exception as e:
            logging.error(f"Failed during image processing: {e}")
            raise
        try:
            if text_tokens.dim() != 2 or text_tokens.shape[0] != img.shape[0]:
                raise ValueError(f"text_tokens must be [batch, seq] with batch {img.shape[0]}, got "
                                 f"{tuple(text_tokens.shape)}")
            model_input = self.decoder.embed(text_tokens, prec, img=img)            # :238-244
        except Exception as e:
            logging.error(f"Failed during text processing: {e}")
            raise
        try:
            return self.decoder.run(model_input, prec, getattr(self, "logits_dtype", torch.float32))   # :250
        except Exception as e:
            logging.error(f"Failed during model forward pass: {e}")
            raise


class KosmosLanguage(nn.Module):
    """Text-only variant (/root/reference/kosmosx/model.py:256-320)."""

    def __init__(self, vocab_size: int = 64007, dim: int = 2048, depth: int = 24, ffn_dim: int = 8192,
                 dropout: float = 0.1, multiway: bool = True, decoder_heads: int = 32, activation_fn: str = "gelu",
                 subln: bool = True, alibi_pos_bias: bool = True, alibi_num_heads: int = 16,
                 xpos_rel_pos: bool = True, max_rel_pos: int = 2048, *args, **kwargs):
        super().__init__()
        seed = kwargs.pop("_seed", None)
        perturb = kwargs.pop("_perturb", 0.0)
        self.precision = _default_precision()
        self.embed = nn.Embedding(vocab_size, dim, padding_idx=1)
        self.embed_positions = PositionalEmbedding(kwargs.pop("_max_positions", dim), dim, 1)  # (dim, dim, 1) upstream
        self.output_projection = nn.Linear(dim, vocab_size, bias=False)
        self.config = DecoderConfig(decoder_layers=depth, decoder_embed_dim=dim, decoder_ffn_embed_dim=ffn_dim,
                                    decoder_attention_heads=decoder_heads, dropout=dropout,
                                    activation_fn=activation_fn, attention_dropout=dropout, vocab_size=vocab_size,
                                    subln=subln, xpos_rel_pos=xpos_rel_pos, multiway=multiway,
                                    max_rel_pos=max_rel_pos, alibi_pos_bias=alibi_pos_bias,
                                    alibi_num_heads=alibi_num_heads)
        self.decoder = Decoder(self.config, embed_tokens=self.embed, embed_positions=self.embed_positions,
                               output_projection=self.output_projection)
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        init_embeddings_(self.embed, self.embed_positions, g, 1)
        _linear_default(self.output_projection, g)     # plain nn.Linear init upstream (:282)
        init_decoder_(self.decoder, g)
        if perturb:
            perturb_(self, g, perturb)

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        if not isinstance(x, torch.Tensor):
            raise TypeError("x must be an instance of torch.Tensor")
        _warn_train_mode(self)
        inc = kwargs.get("incremental_state", None)
        if inc is not None:     # torchscale's incremental protocol; kwargs reach forward_embedding upstream
            return self.decoder._forward_incremental(x, inc, None, self.precision)
        model_input = self.decoder.embed(x, self.precision)         # :319
        return self.decoder.run(model_input, self.precision)        # :320


class KosmosTokenizer:
    """The reference's host pre-processing (/root/reference/kosmosx/model.py:23-129) with its tensor half on the
    device: `tokenize_images` = CLIP resize / centre-crop / rescale / normalize in
    csrc/kx_preprocess.hip, bit-identical to the HF CLIPProcessor; `tokenize_texts` / `tokenize` = the
    `<s> <image> </image> text` id splice and attention mask in `kx_token_splice`.

    The text tokenizer itself (GPT-NeoX vocabulary files) is host string processing and stays the HF object: pass one
    in (`tokenizer=`), or leave it None to load "EleutherAI/gpt-neox-20b" exactly as the reference does (:41-48) —
    which needs the hub or a local cache and otherwise fails with the reference's log-and-re-raise (:49-51).  The
    image half needs no files: the CLIP processor constants are built in."""

    def __init__(self, tokenizer=None, device=None):
        try:
            if tokenizer is None:
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(
                    "EleutherAI/gpt-neox-20b", additional_special_tokens=["<image>", "</image>"], eos_token="<eos>",
                    pad_token="<pad>", extra_ids=0, model_max_length=8192)
            self.tokenizer = tokenizer
        except Exception as e:
            logging.error(f"Failed to initialize KosmosTokenizer: {e}")
            raise
        self.device = torch.device(device) if device is not None else None
        self.im_idx, self.im_end_idx = self.tokenizer.convert_tokens_to_ids(["<image>", "</image>"])

    def _dev(self):
        return self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())

    def tokenize_texts(self, texts):
        """-> (text_tokens [B,L+2] with the image tokens spliced in after <s>, only_text_tokens [B,L]), on the device."""
        try:
            ids = self.tokenizer(texts, return_tensors="pt", padding=True, truncation=True).input_ids
            ids = ids.to(self._dev(), dtype=torch.int64)
            from .preprocess import token_splice
            tok, self._last_mask = token_splice(ids, self.im_idx, self.im_end_idx, self.tokenizer.pad_token_id)
            return tok, ids
        except Exception as e:
            logging.error(f"Failed to tokenize texts: {e}")
            raise

    def tokenize_images(self, images):
        """-> pixel_values float32 [B,3,224,224] on the device (CLIPProcessor(images=...).pixel_values)."""
        try:
            from .preprocess import clip_preprocess
            return clip_preprocess(images, device=self._dev())
        except Exception as e:
            logging.error(f"Failed to tokenize images: {e}")
            raise

    def tokenize(self, sample):
        """{"text_tokens", "images", "labels", "attention_mask"} as the reference builds them (:106-129)."""
        try:
            text_tokens, only_text_tokens = self.tokenize_texts(sample["target_text"])
            return {
                "text_tokens": text_tokens,
                "images": self.tokenize_images(sample["image"]),
                "labels": only_text_tokens,
                "attention_mask": self._last_mask,      # [64 ones | text_tokens != pad], from the same kernel
            }
        except Exception as e:
            logging.error(f"Failed to tokenize sample: {e}")
            raise


