"""Second, independent restatement of the Kosmos-X forward path in plain NumPy (float64, explicit
per-head loops, hand-written LayerNorm / softmax / GELU / XPos) — TEST INFRASTRUCTURE.

It exists to guard the primary oracle (oracle/kosmos_oracle.py, PyTorch ops) against a transcription
slip: the two are written in different styles from the same specification
and must agree to rounding (tests/test_oracle_numpy.py).  Small configurations only: it is slow.
Reference call sites: /root/reference/kosmosx/model.py:230-250 (Kosmos.forward), :319-320.
PARITY STATUS: unpinned at the third-party boundary, same as the primary oracle.
"""
from __future__ import annotations

import math

import numpy as np

_erf = np.vectorize(math.erf)


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def _gelu(x):
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def _act(x, name):
    return _gelu(x) if name == "gelu" else x / (1.0 + np.exp(-1.702 * x))


def _softmax(s):
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    return e / e.sum(-1, keepdims=True)


def _w(w, k):
    return w[k].detach().cpu().double().numpy() if hasattr(w[k], "detach") else np.asarray(w[k], np.float64)


def vit(w, px, c, p="clip_model."):
    px = np.asarray(px, np.float64)
    B = px.shape[0]
    G, ps, D = c.image // c.patch, c.patch, c.dim
    wp = _w(w, p + "embeddings.patch_embedding.weight").reshape(D, -1)
    x = np.zeros((B, G * G + 1, D))
    for b in range(B):
        x[b, 0] = _w(w, p + "embeddings.class_embedding")
        for py in range(G):
            for qx in range(G):
                patch = px[b, :, py * ps:(py + 1) * ps, qx * ps:(qx + 1) * ps].reshape(-1)
                x[b, 1 + py * G + qx] = wp @ patch
    x = x + _w(w, p + "embeddings.position_embedding.weight")[None]
    x = _ln(x, _w(w, p + "pre_layrnorm.weight"), _w(w, p + "pre_layrnorm.bias"), c.eps)
    H, hd = c.heads, D // c.heads
    for i in range(c.layers):
        q_ = f"{p}encoder.layers.{i}."
        y = _ln(x, _w(w, q_ + "layer_norm1.weight"), _w(w, q_ + "layer_norm1.bias"), c.eps)
        q = y @ _w(w, q_ + "self_attn.q_proj.weight").T + _w(w, q_ + "self_attn.q_proj.bias")
        k = y @ _w(w, q_ + "self_attn.k_proj.weight").T + _w(w, q_ + "self_attn.k_proj.bias")
        v = y @ _w(w, q_ + "self_attn.v_proj.weight").T + _w(w, q_ + "self_attn.v_proj.bias")
        o = np.zeros_like(y)
        for h in range(H):
            sl = slice(h * hd, (h + 1) * hd)
            for b in range(B):
                a = _softmax(q[b, :, sl] @ k[b, :, sl].T / math.sqrt(hd))
                o[b, :, sl] = a @ v[b, :, sl]
        x = x + o @ _w(w, q_ + "self_attn.out_proj.weight").T + _w(w, q_ + "self_attn.out_proj.bias")
        y = _ln(x, _w(w, q_ + "layer_norm2.weight"), _w(w, q_ + "layer_norm2.bias"), c.eps)
        y = _act(y @ _w(w, q_ + "mlp.fc1.weight").T + _w(w, q_ + "mlp.fc1.bias"), c.act)
        x = x + y @ _w(w, q_ + "mlp.fc2.weight").T + _w(w, q_ + "mlp.fc2.bias")
    return x


def perceiver(w, x, c, p="perceive."):
    B = x.shape[0]
    x = x + _w(w, p + "media_pos_emb")[0, 0]                     # media_pos_emb[:1] broadcast to every token
    lat = np.broadcast_to(_w(w, p + "latents"), (B,) + _w(w, p + "latents").shape).copy()
    H, dh = c.heads, c.dim_head
    inner = H * dh
    for i in range(c.depth):
        a_, f_ = f"{p}layers.{i}.0.", f"{p}layers.{i}.1."
        xn = _ln(x, _w(w, a_ + "norm_media.weight"), _w(w, a_ + "norm_media.bias"), c.eps)
        ln = _ln(lat, _w(w, a_ + "norm_latents.weight"), _w(w, a_ + "norm_latents.bias"), c.eps)
        q = ln @ _w(w, a_ + "to_q.weight").T
        kv = np.concatenate([xn, ln], axis=1) @ _w(w, a_ + "to_kv.weight").T
        k, v = kv[..., :inner], kv[..., inner:]
        o = np.zeros((B, lat.shape[1], inner))
        for b in range(B):
            for h in range(H):
                sl = slice(h * dh, (h + 1) * dh)
                o[b, :, sl] = _softmax((q[b, :, sl] * dh ** -0.5) @ k[b, :, sl].T) @ v[b, :, sl]
        lat = lat + o @ _w(w, a_ + "to_out.weight").T
        y = _ln(lat, _w(w, f_ + "0.weight"), _w(w, f_ + "0.bias"), c.eps)
        lat = lat + _gelu(y @ _w(w, f_ + "1.weight").T) @ _w(w, f_ + "3.weight").T
    return _ln(lat, _w(w, p + "norm.weight"), _w(w, p + "norm.bias"), c.eps)


def xpos(x, downscale, scale_base=512):
    """x [T, hd] for one head: rotate pair (2j, 2j+1) by p*10000^(-j/(hd/2)), scale by zeta_j^(+-(min_pos+p)/base)."""
    T, hd = x.shape
    half = hd // 2
    min_pos = -((T + 1) // 2) if T % 2 else -(T // 2)            # == Python's  -(T) // 2
    y = np.empty_like(x)
    for p in range(T):
        for j in range(half):
            zeta = (2 * j + 0.4 * hd) / (1.4 * hd)
            s = zeta ** ((min_pos + p) / scale_base)
            if downscale:
                s = 1.0 / s
            th = p * 10000.0 ** (-j / half)
            c_, s_ = math.cos(th) * s, math.sin(th) * s
            a, b = x[p, 2 * j], x[p, 2 * j + 1]
            y[p, 2 * j] = a * c_ - b * s_
            y[p, 2 * j + 1] = b * c_ + a * s_
    return y


def decoder(w, x, c, p="decoder.", mw=".A"):
    B, T, D = x.shape
    H, hd = c.heads, D // c.heads
    for i in range(c.layers):
        l_ = f"{p}layers.{i}."
        y = _ln(x, _w(w, l_ + f"self_attn_layer_norm{mw}.weight"), _w(w, l_ + f"self_attn_layer_norm{mw}.bias"), c.eps)
        q = (y @ _w(w, l_ + f"self_attn.q_proj{mw}.weight").T + _w(w, l_ + f"self_attn.q_proj{mw}.bias")) * hd ** -0.5
        k = y @ _w(w, l_ + f"self_attn.k_proj{mw}.weight").T + _w(w, l_ + f"self_attn.k_proj{mw}.bias")
        v = y @ _w(w, l_ + f"self_attn.v_proj{mw}.weight").T + _w(w, l_ + f"self_attn.v_proj{mw}.bias")
        o = np.zeros_like(y)
        for b in range(B):
            for h in range(H):
                sl = slice(h * hd, (h + 1) * hd)
                qq, kk = q[b, :, sl], k[b, :, sl]
                if c.xpos:
                    qq, kk = xpos(qq, False, c.xpos_scale_base), xpos(kk, True, c.xpos_scale_base)
                s = qq @ kk.T
                s[np.triu_indices(T, 1)] = -np.inf
                o[b, :, sl] = _softmax(s) @ v[b, :, sl]
        if c.subln:
            o = _ln(o, _w(w, l_ + f"self_attn.inner_attn_ln{mw}.weight"), _w(w, l_ + f"self_attn.inner_attn_ln{mw}.bias"), c.eps)
        x = x + o @ _w(w, l_ + f"self_attn.out_proj{mw}.weight").T + _w(w, l_ + f"self_attn.out_proj{mw}.bias")
        y = _ln(x, _w(w, l_ + f"final_layer_norm{mw}.weight"), _w(w, l_ + f"final_layer_norm{mw}.bias"), c.eps)
        y = _gelu(y @ _w(w, l_ + f"ffn{mw}.fc1.weight").T + _w(w, l_ + f"ffn{mw}.fc1.bias"))
        if c.subln:
            y = _ln(y, _w(w, l_ + f"ffn{mw}.ffn_layernorm.weight"), _w(w, l_ + f"ffn{mw}.ffn_layernorm.bias"), c.eps)
        x = x + y @ _w(w, l_ + f"ffn{mw}.fc2.weight").T + _w(w, l_ + f"ffn{mw}.fc2.bias")
    x = _ln(x, _w(w, p + "layer_norm.weight"), _w(w, p + "layer_norm.bias"), c.eps)
    return x @ _w(w, "output_projection.weight").T


def kosmos(w, tokens, images, cfg, u1_alias=True):
    tokens = np.asarray(tokens)
    B, Tt = tokens.shape
    img = perceiver(w, vit(w, images, cfg.vit), cfg.perceiver) @ _w(w, "image_proj.weight").T
    n = img.shape[1]
    emb, pos = _w(w, "embed.weight"), _w(w, "embed_positions.weight")
    T = Tt + n
    x = np.zeros((B, T, emb.shape[1]))
    for b in range(B):
        for t in range(T):
            if 2 <= t < 2 + n:
                row = img[b, t - 2]
            else:
                tt = t if t < 2 else t - n
                row = emb[tokens[b, tt]] + (pos[2 + tt] if u1_alias else 0.0)
            x[b, t] = row + pos[2 + t]
    return decoder(w, x, cfg.decoder)


def kosmos_language(w, tokens, dcfg):
    tokens = np.asarray(tokens)
    emb, pos = _w(w, "embed.weight"), _w(w, "embed_positions.weight")
    x = emb[tokens] + pos[2:2 + tokens.shape[1]][None]
    return decoder(w, x, dcfg)
