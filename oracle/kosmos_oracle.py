"""CPU oracle for the Kosmos-X multimodal forward path (TEST INFRASTRUCTURE, not product).

This file is a plain-PyTorch fp32 *restatement* of the arithmetic reached by
``kosmosx.model.Kosmos.forward`` (/root/reference/kosmosx/model.py:208-253) and
``KosmosLanguage.forward`` (/root/reference/kosmosx/model.py:310-320).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product package (``kosmos-x_amd/``) never does.

PARITY STATUS: **parity unpinned at the third-party boundary** (SURVEY.md §8c).  The reference
repo holds no arithmetic and no golden vectors of its own; the arithmetic lives in four
third-party packages, of which only ``transformers`` is importable here:

* ViT-L/14 vision tower  -> follows HF ``modeling_clip.py`` (CLIPVisionEmbeddings.forward,
  CLIPEncoderLayer.forward, eager_attention_forward, CLIPVisionTransformer.forward);
  PINNED: tests/test_oracle_crosscheck.py diffs it against the installed HF module.
* PerceiverResampler     -> restates lucidrains ``flamingo_pytorch/flamingo_pytorch.py``
  (PerceiverAttention.forward / PerceiverResampler.forward), package absent, unpinned version
  (/root/reference/requirements.txt:4); attention PINNED against HF IdeficsPerceiverAttention.
* Decoder (sub-LN, XPos) -> restates microsoft ``torchscale`` (architecture/decoder.py,
  component/multihead_attention.py, component/xpos_relative_position.py,
  component/feedforward_network.py, component/embedding.py) with the ``passed_x`` patch of
  /root/reference/README.md:179-193; package absent, unpinned (/root/reference/requirements.txt:17).
  Block structure PINNED against HF Kosmos2TextBlock, rotary half of XPos PINNED against HF GPT-J.
* bitsandbytes Embedding -> plain F.embedding forward.

Every point that could not be corroborated inside this container is an explicit switch in
``Switches`` (U1, U3b, U5, U6, U7 of SURVEY.md §8c) with the recalled upstream behaviour as default.

Weights are a flat ``dict[str, Tensor]`` keyed by the reference's state_dict names
(SURVEY.md §8b), multiway "A" branch only (U7).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class VitCfg:
    """CLIP ViT-L/14 (/root/reference/kosmosx/model.py:154-156)."""
    image: int = 224
    patch: int = 14
    dim: int = 1024
    heads: int = 16
    ffn: int = 4096
    layers: int = 24
    eps: float = 1e-5
    act: str = "gelu"  # U5: "gelu" | "quick_gelu"

    @property
    def tokens(self) -> int:
        return (self.image // self.patch) ** 2 + 1


@dataclass
class PerceiverCfg:
    """PerceiverResampler(dim=1024, depth=2, dim_head=64, heads=8, num_latents=64,
    num_media_embeds=257) (/root/reference/kosmosx/model.py:196-203)."""
    dim: int = 1024
    depth: int = 2
    dim_head: int = 64
    heads: int = 8
    latents: int = 64
    media_embeds: int = 257
    ff_mult: int = 4
    eps: float = 1e-5


@dataclass
class DecoderCfg:
    """DecoderConfig(...) (/root/reference/kosmosx/model.py:170-183)."""
    layers: int = 24
    dim: int = 2048
    ffn: int = 8192
    heads: int = 32
    vocab: int = 32002
    max_pos: int = 2048  # PositionalEmbedding(2048, 2048, 1) (/root/reference/kosmosx/model.py:164)
    eps: float = 1e-5
    xpos_scale_base: int = 512
    subln: bool = True
    xpos: bool = True
    act: str = "gelu"


@dataclass
class KosmosCfg:
    vit: VitCfg = field(default_factory=VitCfg)
    perceiver: PerceiverCfg = field(default_factory=PerceiverCfg)
    decoder: DecoderCfg = field(default_factory=DecoderCfg)


def tiny_cfg() -> KosmosCfg:
    """Reduced configuration that exercises every code path (head_dim stays 64 everywhere,
    which is what the HIP attention kernels are specialised for)."""
    return KosmosCfg(
        vit=VitCfg(image=56, patch=14, dim=128, heads=2, ffn=256, layers=2),
        perceiver=PerceiverCfg(dim=128, depth=2, dim_head=64, heads=2, latents=8, media_embeds=17),
        decoder=DecoderCfg(layers=2, dim=256, ffn=512, heads=4, vocab=1002, max_pos=64),
    )


@dataclass
class Switches:
    """Unverifiable points of SURVEY.md §8c; defaults = recalled upstream behaviour."""
    u1_inplace_alias: bool = True   # forward_embedding()[1] already contains positions
    u3b_xpos_scale: bool = True     # zeta-scale half of XPos (False => plain rotary)
    u6_media_pos_first_only: bool = True  # media_pos_emb[:1] broadcast
    u6_kv_k_first: bool = True      # to_kv chunk order: k first, v second
    emulate_bf16: bool = False      # round every matmul operand to bf16 (predicts the GPU bf16 path)


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
# The dtype of the places where the reference casts explicitly (HF: pixel_values.to(weight dtype); torchscale / HF: softmax in
# fp32; torchscale FFN: activation in fp32).  float32 = the reference.  `working_dtype(torch.float64)` together with float64
# weights evaluates the SAME algorithm in double precision: the yardstick that separates an fp32 implementation's
# summation-order noise from an algorithmic difference (bench.py cpu_baseline.fp32_vs_float64, tests/test_model_gpu.py).
_WORK = [torch.float32]


class working_dtype:
    def __init__(self, dt):
        self.dt = dt

    def __enter__(self):
        self.prev, _WORK[0] = _WORK[0], self.dt

    def __exit__(self, *a):
        _WORK[0] = self.prev


def _r(x: torch.Tensor, sw: Switches) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32) if sw.emulate_bf16 else x


def linear(x, w, b, sw: Switches):
    """nn.Linear: y = x W^T + b, W is [out, in]."""
    y = _r(x, sw) @ _r(w, sw).t()
    return y if b is None else y + b


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def act_fn(x, name: str):
    if name == "gelu":
        return F.gelu(x)                      # erf form
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)   # HF activations.QuickGELUActivation
    if name == "relu":                        # torchscale component/feedforward_network.py get_activation_fn: F.relu
        return F.relu(x)
    if name == "swish":                       # ... and F.silu
        return F.silu(x)
    raise ValueError(name)


# --------------------------------------------------------------------------------------
# a3: CLIP ViT-L/14 vision tower (HF modeling_clip.py)
# --------------------------------------------------------------------------------------
def vit_forward(w: dict, pixels: torch.Tensor, cfg: VitCfg, sw: Switches, prefix="clip_model.") -> torch.Tensor:
    """[B,3,H,W] any real dtype -> last_hidden_state [B, tokens, dim] (no post_layernorm on the
    sequence: HF CLIPVisionTransformer.forward)."""
    p = prefix
    x = pixels.to(_WORK[0])  # HF: pixel_values.to(dtype=target_dtype)
    B = x.shape[0]
    pw = w[p + "embeddings.patch_embedding.weight"]
    if sw.emulate_bf16:
        pe = F.conv2d(_r(x, sw), _r(pw, sw), None, stride=cfg.patch)
    else:
        pe = F.conv2d(x, pw, None, stride=cfg.patch)
    pe = pe.flatten(2).transpose(1, 2)                                   # [B, P, dim]
    cls = w[p + "embeddings.class_embedding"].expand(B, 1, -1)
    h = torch.cat([cls, pe], dim=1) + w[p + "embeddings.position_embedding.weight"][None]
    h = layer_norm(h, w[p + "pre_layrnorm.weight"], w[p + "pre_layrnorm.bias"], cfg.eps)
    hd = cfg.dim // cfg.heads
    for i in range(cfg.layers):
        q_ = f"{p}encoder.layers.{i}."
        r = h
        y = layer_norm(h, w[q_ + "layer_norm1.weight"], w[q_ + "layer_norm1.bias"], cfg.eps)
        q = linear(y, w[q_ + "self_attn.q_proj.weight"], w[q_ + "self_attn.q_proj.bias"], sw)
        k = linear(y, w[q_ + "self_attn.k_proj.weight"], w[q_ + "self_attn.k_proj.bias"], sw)
        v = linear(y, w[q_ + "self_attn.v_proj.weight"], w[q_ + "self_attn.v_proj.bias"], sw)
        S = y.shape[1]
        q = q.view(B, S, cfg.heads, hd).transpose(1, 2)
        k = k.view(B, S, cfg.heads, hd).transpose(1, 2)
        v = v.view(B, S, cfg.heads, hd).transpose(1, 2)
        a = (_r(q, sw) @ _r(k, sw).transpose(-1, -2)) * (hd ** -0.5)
        a = F.softmax(a, dim=-1, dtype=_WORK[0])
        o = (_r(a, sw) @ _r(v, sw)).transpose(1, 2).reshape(B, S, cfg.dim)
        o = linear(o, w[q_ + "self_attn.out_proj.weight"], w[q_ + "self_attn.out_proj.bias"], sw)
        h = r + o
        r = h
        y = layer_norm(h, w[q_ + "layer_norm2.weight"], w[q_ + "layer_norm2.bias"], cfg.eps)
        y = linear(y, w[q_ + "mlp.fc1.weight"], w[q_ + "mlp.fc1.bias"], sw)
        y = act_fn(y, cfg.act)
        y = linear(y, w[q_ + "mlp.fc2.weight"], w[q_ + "mlp.fc2.bias"], sw)
        h = r + y
    return h


# --------------------------------------------------------------------------------------
# a4: PerceiverResampler (flamingo_pytorch, recalled)
# --------------------------------------------------------------------------------------
def perceiver_forward(w: dict, x: torch.Tensor, cfg: PerceiverCfg, sw: Switches, prefix="perceive.") -> torch.Tensor:
    """[B, m, dim] -> [B, 1, latents, dim] (the caller squeezes dim 1,
    /root/reference/kosmosx/model.py:231)."""
    p = prefix
    B = x.shape[0]
    x = x[:, None]                                                       # b n d -> b 1 n d
    times = x.shape[1]
    mpe = w[p + "media_pos_emb"]                                         # [num_media_embeds, 1, dim]
    x = x + (mpe[:times] if sw.u6_media_pos_first_only else mpe[: x.shape[2]].transpose(0, 1))
    lat = w[p + "latents"][None, None].expand(B, times, -1, -1)
    H, dh = cfg.heads, cfg.dim_head
    inner = H * dh
    for i in range(cfg.depth):
        a_ = f"{p}layers.{i}.0."
        f_ = f"{p}layers.{i}.1."
        xn = layer_norm(x, w[a_ + "norm_media.weight"], w[a_ + "norm_media.bias"], cfg.eps)
        ln = layer_norm(lat, w[a_ + "norm_latents.weight"], w[a_ + "norm_latents.bias"], cfg.eps)
        q = linear(ln, w[a_ + "to_q.weight"], None, sw)
        kv_in = torch.cat((xn, ln), dim=-2)                              # keys include the latents
        kv = linear(kv_in, w[a_ + "to_kv.weight"], None, sw)
        k, v = kv.chunk(2, dim=-1) if sw.u6_kv_k_first else kv.chunk(2, dim=-1)[::-1]

        def split(t):  # b t n (h d) -> b h t n d
            b, t_, n, _ = t.shape
            return t.view(b, t_, n, H, dh).permute(0, 3, 1, 2, 4)

        q, k, v = split(q), split(k), split(v)
        q = q * (dh ** -0.5)
        sim = _r(q, sw) @ _r(k, sw).transpose(-1, -2)
        sim = sim - sim.amax(dim=-1, keepdim=True)
        attn = sim.softmax(dim=-1)
        o = _r(attn, sw) @ _r(v, sw)                                     # b h t n d
        o = o.permute(0, 2, 3, 1, 4).reshape(B, times, -1, inner)
        lat = linear(o, w[a_ + "to_out.weight"], None, sw) + lat
        y = layer_norm(lat, w[f_ + "0.weight"], w[f_ + "0.bias"], cfg.eps)
        y = linear(y, w[f_ + "1.weight"], None, sw)
        y = F.gelu(y)
        y = linear(y, w[f_ + "3.weight"], None, sw)
        lat = y + lat
    return layer_norm(lat, w[p + "norm.weight"], w[p + "norm.bias"], cfg.eps)


# --------------------------------------------------------------------------------------
# a11: XPos (torchscale component/xpos_relative_position.py, recalled)
# --------------------------------------------------------------------------------------
def xpos_tables(length: int, head_dim: int, scale_base: int, offset: int = 0, downscale: bool = False,
                use_scale: bool = True):
    """Returns (cos*scale, sin*scale), each [length, head_dim/2], fp32 — the two tables
    ``apply_rotary_pos_emb`` multiplies with after ``duplicate_interleave``."""
    half = head_dim // 2
    zeta = (torch.arange(0, head_dim, 2, dtype=torch.float32) + 0.4 * head_dim) / (1.4 * head_dim)
    min_pos = -(length + offset) // 2          # Python floor division of the negated sum
    max_pos = length + offset + min_pos
    expo = torch.arange(min_pos, max_pos, 1).to(zeta).div(scale_base)[:, None]
    scale = zeta ** expo                                                 # [L+offset, half]
    seq_len = scale.shape[0]
    inv_freq = 1.0 / (10000 ** (torch.arange(0, half) / half))
    sinusoid = torch.einsum("i , j -> i j", torch.arange(0, seq_len, dtype=torch.float), inv_freq).to(scale)
    sin, cos = torch.sin(sinusoid), torch.cos(sinusoid)
    if scale.shape[0] > length:
        scale, sin, cos = scale[-length:], sin[-length:], cos[-length:]
    if not use_scale:
        scale = torch.ones_like(scale)
    if downscale:
        scale = 1 / scale
    return cos * scale, sin * scale


def _dup_interleave(m):  # [L, half] -> [L, 2*half] with each entry repeated twice
    return m.view(-1, 1).repeat(1, 2).view(m.shape[0], -1)


def _rotate_every_two(x):
    x1, x2 = x[:, :, ::2], x[:, :, 1::2]
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def apply_xpos(x, cs, ss):
    """x [BH, L, hd]; cs/ss from xpos_tables."""
    return (x * _dup_interleave(cs)) + (_rotate_every_two(x) * _dup_interleave(ss))


# --------------------------------------------------------------------------------------
# a9-a13: Decoder (torchscale, recalled) with the passed_x patch
# --------------------------------------------------------------------------------------
def decoder_layer(w: dict, x: torch.Tensor, i: int, cfg: DecoderCfg, sw: Switches, prefix="decoder.",
                  mw=".A", drop: dict | None = None):
    """drop (training mode; None = eval): {site: keep/(1-p) tensor} at torchscale's dropout_module calls —
    site 1+3i: the attention probabilities [B*H, T, T] (MultiheadAttention, attention_dropout), 2+3i: the attention
    block's output before the residual add, 3+3i: fc2's output before the residual add (DecoderLayer / FeedForwardNetwork,
    dropout).  Site 0 (after the embedding) is applied by the caller."""
    p = f"{prefix}layers.{i}."
    B, T, D = x.shape
    H = cfg.heads
    hd = D // H
    r = x
    y = layer_norm(x, w[p + f"self_attn_layer_norm{mw}.weight"], w[p + f"self_attn_layer_norm{mw}.bias"], cfg.eps)
    q = linear(y, w[p + f"self_attn.q_proj{mw}.weight"], w[p + f"self_attn.q_proj{mw}.bias"], sw)
    k = linear(y, w[p + f"self_attn.k_proj{mw}.weight"], w[p + f"self_attn.k_proj{mw}.bias"], sw)
    v = linear(y, w[p + f"self_attn.v_proj{mw}.weight"], w[p + f"self_attn.v_proj{mw}.bias"], sw)
    q = q * (hd ** -0.5)
    q = q.view(B, T, H, hd).transpose(1, 2).reshape(B * H, T, hd)
    k = k.view(B, T, H, hd).transpose(1, 2).reshape(B * H, T, hd)
    v = v.view(B, T, H, hd).transpose(1, 2).reshape(B * H, T, hd)
    if cfg.xpos:
        kc, ks = xpos_tables(T, hd, cfg.xpos_scale_base, 0, True, sw.u3b_xpos_scale)
        qc, qs = xpos_tables(T, hd, cfg.xpos_scale_base, 0, False, sw.u3b_xpos_scale)
        k = apply_xpos(k, kc, ks)
        q = apply_xpos(q, qc, qs)
    a = torch.bmm(_r(q, sw), _r(k, sw).transpose(1, 2))
    mask = torch.triu(torch.zeros([T, T]).to(_WORK[0]).fill_(float("-inf")), 1)
    a = torch.nan_to_num(a) + mask[None]
    a = F.softmax(a, dim=-1, dtype=_WORK[0])
    if drop is not None and (1 + 3 * i) in drop:
        a = a * drop[1 + 3 * i]
    o = torch.bmm(_r(a, sw), _r(v, sw))
    o = o.transpose(0, 1).reshape(T, B, D).transpose(0, 1)
    if cfg.subln:
        o = layer_norm(o, w[p + f"self_attn.inner_attn_ln{mw}.weight"], w[p + f"self_attn.inner_attn_ln{mw}.bias"], cfg.eps)
    o = linear(o, w[p + f"self_attn.out_proj{mw}.weight"], w[p + f"self_attn.out_proj{mw}.bias"], sw)
    if drop is not None and (2 + 3 * i) in drop:
        o = o * drop[2 + 3 * i]
    x = r * 1.0 + o                                                      # residual_connection, alpha = 1
    r = x
    y = layer_norm(x, w[p + f"final_layer_norm{mw}.weight"], w[p + f"final_layer_norm{mw}.bias"], cfg.eps)
    y = y.reshape(-1, D)
    y = linear(y, w[p + f"ffn{mw}.fc1.weight"], w[p + f"ffn{mw}.fc1.bias"], sw)
    y = act_fn(y.to(_WORK[0]), cfg.act)
    if cfg.subln:
        y = layer_norm(y, w[p + f"ffn{mw}.ffn_layernorm.weight"], w[p + f"ffn{mw}.ffn_layernorm.bias"], cfg.eps)
    y = linear(y, w[p + f"ffn{mw}.fc2.weight"], w[p + f"ffn{mw}.fc2.bias"], sw).view(B, T, D)
    if drop is not None and (3 + 3 * i) in drop:
        y = y * drop[3 + 3 * i]
    return r * 1.0 + y


def decoder_forward(w: dict, x: torch.Tensor, cfg: DecoderCfg, sw: Switches, prefix="decoder.",
                    features_only: bool = False, drop: dict | None = None, mw: str = ".A") -> torch.Tensor:
    """Decoder.forward(..., passed_x=x)[0]: 24 layers, final LayerNorm, output_projection.
    mw = ".A": multiway=True wraps every LayerNorm / Linear / FFN in MultiwayNetwork, whose keys carry the branch name and
    whose forward is the A branch (split_position == -1); mw = "": multiway=False, plain modules."""
    for i in range(cfg.layers):
        x = decoder_layer(w, x, i, cfg, sw, prefix, mw=mw, drop=drop)
    x = layer_norm(x, w[prefix + "layer_norm.weight"], w[prefix + "layer_norm.bias"], cfg.eps)
    if features_only:
        return x
    return linear(x, w["output_projection.weight"], None, sw)


def decoder_incremental(w: dict, x: torch.Tensor, cfg: DecoderCfg, sw: Switches, prefix="decoder.", mw=".A",
                        first: int | None = None) -> torch.Tensor:
    """torchscale's incremental_state path, restated (MultiheadAttention.forward with prev_key/prev_value,
    Decoder.forward's `is_first_step`): the first `first` positions are processed as one block (causal mask, XPos
    offset 0), every later position one at a time with the cache of PRE-XPos keys: k of the whole cache is re-rotated
    with offset 0 (min_pos = -(src_len)//2) and q with offset src_len-1; no mask.  Returns logits for all positions.
    x [B,T,D] = embedded inputs (forward_embedding already applied)."""
    B, T, D = x.shape
    H = cfg.heads
    hd = D // H
    first = T if first is None else first
    caches = [dict() for _ in range(cfg.layers)]
    outs = []
    steps = [(0, first)] + [(t, t + 1) for t in range(first, T)]
    for (lo, hi) in steps:
        h = x[:, lo:hi]
        n = hi - lo
        for i in range(cfg.layers):
            p = f"{prefix}layers.{i}."
            r = h
            y = layer_norm(h, w[p + f"self_attn_layer_norm{mw}.weight"], w[p + f"self_attn_layer_norm{mw}.bias"], cfg.eps)
            q = linear(y, w[p + f"self_attn.q_proj{mw}.weight"], w[p + f"self_attn.q_proj{mw}.bias"], sw) * (hd ** -0.5)
            k = linear(y, w[p + f"self_attn.k_proj{mw}.weight"], w[p + f"self_attn.k_proj{mw}.bias"], sw)
            v = linear(y, w[p + f"self_attn.v_proj{mw}.weight"], w[p + f"self_attn.v_proj{mw}.bias"], sw)
            sp = lambda t_: t_.view(B, n, H, hd).transpose(1, 2).reshape(B * H, n, hd)
            q, k, v = sp(q), sp(k), sp(v)
            c = caches[i]
            if "k" in c:
                k, v = torch.cat([c["k"], k], 1), torch.cat([c["v"], v], 1)
            c["k"], c["v"] = k, v
            src = k.shape[1]
            if cfg.xpos:
                off = src - 1 if lo > 0 else 0
                kc, ks = xpos_tables(src, hd, cfg.xpos_scale_base, 0, True, sw.u3b_xpos_scale)
                qc, qs = xpos_tables(n, hd, cfg.xpos_scale_base, off, False, sw.u3b_xpos_scale)
                k = apply_xpos(k, kc, ks)
                q = apply_xpos(q, qc, qs)
            a = torch.bmm(_r(q, sw), _r(k, sw).transpose(1, 2))
            if lo == 0:
                a = torch.nan_to_num(a) + torch.triu(torch.zeros([n, n]).fill_(float("-inf")), 1)[None]
            a = F.softmax(a, dim=-1, dtype=_WORK[0])
            o = torch.bmm(_r(a, sw), _r(v, sw)).transpose(0, 1).reshape(n, B, D).transpose(0, 1)
            if cfg.subln:
                o = layer_norm(o, w[p + f"self_attn.inner_attn_ln{mw}.weight"], w[p + f"self_attn.inner_attn_ln{mw}.bias"], cfg.eps)
            h = r + linear(o, w[p + f"self_attn.out_proj{mw}.weight"], w[p + f"self_attn.out_proj{mw}.bias"], sw)
            r = h
            y = layer_norm(h, w[p + f"final_layer_norm{mw}.weight"], w[p + f"final_layer_norm{mw}.bias"], cfg.eps)
            y = act_fn(linear(y, w[p + f"ffn{mw}.fc1.weight"], w[p + f"ffn{mw}.fc1.bias"], sw), cfg.act)
            if cfg.subln:
                y = layer_norm(y, w[p + f"ffn{mw}.ffn_layernorm.weight"], w[p + f"ffn{mw}.ffn_layernorm.bias"], cfg.eps)
            h = r + linear(y, w[p + f"ffn{mw}.fc2.weight"], w[p + f"ffn{mw}.fc2.bias"], sw)
        h = layer_norm(h, w[prefix + "layer_norm.weight"], w[prefix + "layer_norm.bias"], cfg.eps)
        outs.append(linear(h, w["output_projection.weight"], None, sw))
    return torch.cat(outs, dim=1)


def positions_for(T: int) -> torch.Tensor:
    """PositionalEmbedding.forward: fairseq convention, positions start at 2."""
    return torch.arange(2, T + 2).long()


def forward_embedding_tokens(w: dict, tokens: torch.Tensor, cfg: DecoderCfg):
    """Decoder.forward_embedding(tokens) -> (x, embed). embed_scale = 1 (no_scale_embedding=True)."""
    T = tokens.shape[1]
    if T + 2 > cfg.max_pos:
        raise IndexError(f"position {T + 1} out of range for a {cfg.max_pos}-row table (SURVEY H3)")
    pos = w["embed_positions.weight"][positions_for(T)][None]
    tok = F.embedding(tokens, w["embed.weight"])
    embed = 1.0 * tok
    x = embed + pos
    return x, embed


def kosmos_forward(w: dict, text_tokens: torch.Tensor, images: torch.Tensor, cfg: KosmosCfg,
                   sw: Switches | None = None, stages: dict | None = None) -> torch.Tensor:
    """Kosmos.forward (/root/reference/kosmosx/model.py:208-253), eval mode.
    ``stages`` (optional dict) receives the intermediate tensors for per-stage parity tests."""
    sw = sw or Switches()
    with torch.no_grad():
        img = vit_forward(w, images, cfg.vit, sw)                        # :230
        if stages is not None:
            stages["vit"] = img
        img = perceiver_forward(w, img, cfg.perceiver, sw).squeeze(1)    # :231
        if stages is not None:
            stages["perceiver"] = img
        img = linear(img, w["image_proj.weight"], None, sw)              # :232
        if stages is not None:
            stages["image_proj"] = img
        x, embed = forward_embedding_tokens(w, text_tokens, cfg.decoder)  # :238, takes [1]
        first = x if sw.u1_inplace_alias else embed                      # U1: x += positions aliases embed
        mi = torch.cat([first[:, 0:2], img, first[:, 2:]], dim=1)        # :239-241
        T = mi.shape[1]
        if T + 2 > cfg.decoder.max_pos:
            raise IndexError("sequence exceeds the position table (SURVEY H3)")
        mi = 1.0 * mi + w["embed_positions.weight"][positions_for(T)][None]  # :242-244, [0]
        if stages is not None:
            stages["embed"] = mi
        return decoder_forward(w, mi, cfg.decoder, sw)                   # :250


def kosmos_language_forward(w: dict, tokens: torch.Tensor, cfg: DecoderCfg, sw: Switches | None = None, mw: str = ".A"):
    """KosmosLanguage.forward (/root/reference/kosmosx/model.py:319-320), eval mode."""
    sw = sw or Switches()
    with torch.no_grad():
        x, _ = forward_embedding_tokens(w, tokens, cfg)
        return decoder_forward(w, x, cfg, sw, mw=mw)
