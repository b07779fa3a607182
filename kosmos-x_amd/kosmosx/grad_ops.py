"""Thin wrappers over the backward / optimizer entry points of libkosmosx_hip.so (training step).
fp32 CUDA tensors in, fp32 CUDA tensors out; no CPU fallback (the library call fails loudly on anything else)."""
from __future__ import annotations

import torch

from . import _hip as H
from .ops import _need_cuda, _stream


def _ws(nbytes: int, dev) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 4096), dtype=torch.uint8, device=dev)


def transpose(x: torch.Tensor, pad_to: int = 1) -> torch.Tensor:
    """[R,C] -> [C,Rp] (fp32 or bf16), Rp = R rounded up to a multiple of pad_to, the padding columns zero —
    the K dimension of a GEMM operand has to be a multiple of 32 (fp32) / 64 (bf16)."""
    _need_cuda(x)
    R, Cc = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    out = (torch.zeros if Rp != R else torch.empty)((Cc, Rp), dtype=x.dtype, device=x.device)
    dt = H.KX_F32 if x.dtype == torch.float32 else H.KX_BF16
    H.check(H.load().kx_transpose(H.ptr(x), H.ptr(out), R, Cc, x.stride(0), Rp, dt, _stream()), "kx_transpose")
    return out


FMT = {"bf16": 1, "bf16x3_act": 2, "bf16x3_w": 3}


def to_operand(x: torch.Tensor, fmt: str, transpose_: bool = False) -> torch.Tensor:
    """fp32 [R,C] -> bf16 GEMM operand of (x^T if transpose_ else x): K padded to a multiple of 64 with zeros, rows
    [K] (bf16) or [3K] (bf16x3: activation rows hi|hi|lo, weight rows hi|lo|hi)."""
    _need_cuda(x)
    R, Cc = x.shape
    orows, ocols = (Cc, R) if transpose_ else (R, Cc)
    kp = (ocols + 63) // 64 * 64
    out = torch.empty((orows, kp if fmt == "bf16" else 3 * kp), dtype=torch.bfloat16, device=x.device)
    H.check(H.load().kx_to_operand(H.ptr(x), H.ptr(out), R, Cc, x.stride(0), kp, int(transpose_), FMT[fmt], _stream()),
            "kx_to_operand")
    return out


def to_operand_pair(x: torch.Tensor, straight: bool = True, transposed: bool = True, colsum_out: torch.Tensor | None = None):
    """fp32 [R,C] -> (bf16 operand rows [R, Cp], bf16 rows of x^T [C, Rp]) in one pass (Cp / Rp = C / R rounded up to 64,
    padding zero); an output that is not asked for is None.  colsum_out [C] fp32: also the column sums (a bias gradient)."""
    _need_cuda(x, colsum_out)
    R, Cc = x.shape
    kp, kpt = (Cc + 63) // 64 * 64, (R + 63) // 64 * 64
    a = torch.empty((R, kp), dtype=torch.bfloat16, device=x.device) if straight else None
    t = torch.empty((Cc, kpt), dtype=torch.bfloat16, device=x.device) if transposed else None
    lib = H.load()
    ws, n = None, 0
    if colsum_out is not None:
        n = lib.kx_to_operand_pair_workspace_bytes(R, Cc)
        ws = _ws(n, x.device)
    H.check(lib.kx_to_operand_pair(H.ptr(x), H.ptr(a), H.ptr(t), R, Cc, x.stride(0), kp, kpt, H.ptr(colsum_out), H.ptr(ws),
                                   ws.numel() if ws is not None else 0, _stream()), "kx_to_operand_pair")
    return a, t


def gelu_backward_pair(pre: torch.Tensor, dg: torch.Tensor, colsum_out: torch.Tensor | None = None):
    """dpre = dg * gelu'(pre) as the bf16 operand pair of to_operand_pair (and its column sums), in one pass over pre and dg."""
    _need_cuda(pre, dg, colsum_out)
    if pre.shape != dg.shape or pre.stride(0) != dg.stride(0):
        raise ValueError("gelu_backward_pair: pre and dg must share shape and row pitch")
    R, Cc = dg.shape
    kp, kpt = (Cc + 63) // 64 * 64, (R + 63) // 64 * 64
    a = torch.empty((R, kp), dtype=torch.bfloat16, device=dg.device)
    t = torch.empty((Cc, kpt), dtype=torch.bfloat16, device=dg.device)
    lib = H.load()
    ws, n = None, 0
    if colsum_out is not None:
        n = lib.kx_to_operand_pair_workspace_bytes(R, Cc)
        ws = _ws(n, dg.device)
    H.check(lib.kx_gelu_backward_operand_pair(H.ptr(dg), H.ptr(pre), H.ptr(a), H.ptr(t), R, Cc, dg.stride(0), kp, kpt,
                                              H.ptr(colsum_out), H.ptr(ws), ws.numel() if ws is not None else 0, _stream()),
            "kx_gelu_backward_operand_pair")
    return a, t


def gelu(pre: torch.Tensor) -> torch.Tensor:
    _need_cuda(pre)
    out = torch.empty_like(pre)
    H.check(H.load().kx_gelu_forward(H.ptr(pre), H.ptr(out), pre.numel(), _stream()), "kx_gelu_forward")
    return out


def colsum(x: torch.Tensor, out: torch.Tensor | None = None, accumulate: bool = False) -> torch.Tensor:
    _need_cuda(x, out)
    R, Cc = x.shape
    if out is None:
        out = torch.empty(Cc, dtype=torch.float32, device=x.device)
    lib = H.load()
    n = lib.kx_colsum_workspace_bytes(R, Cc)
    ws = _ws(n, x.device)
    H.check(lib.kx_colsum(H.ptr(x), R, Cc, x.stride(0), H.ptr(out), int(accumulate), H.ptr(ws), ws.numel(), _stream()),
            "kx_colsum")
    return out


def layernorm_backward(x, gamma, dy, eps=1e-5, dres=None, want_param_grads=True, dgamma_out=None, dbeta_out=None):
    """-> (dx, dgamma, dbeta); dres (optional) is added to dx (gradient arriving through the residual branch)."""
    _need_cuda(x, gamma, dy, dres, dgamma_out, dbeta_out)
    R, Cc = x.shape
    dx = torch.empty_like(x)
    dg = (dgamma_out if dgamma_out is not None else torch.empty(Cc, dtype=torch.float32, device=x.device)) if want_param_grads else None
    db = (dbeta_out if dbeta_out is not None else torch.empty(Cc, dtype=torch.float32, device=x.device)) if want_param_grads else None
    lib = H.load()
    ws = _ws(lib.kx_layernorm_backward_workspace_bytes(R, Cc), x.device)
    H.check(lib.kx_layernorm_backward(H.ptr(x), H.ptr(gamma), H.ptr(dy), H.ptr(dres), H.ptr(dx), H.ptr(dg), H.ptr(db), R, Cc,
                                      float(eps), H.ptr(ws), ws.numel(), _stream()), "kx_layernorm_backward")
    return dx, dg, db


def gelu_layernorm(pre, gamma, beta, eps=1e-5, out_dtype=torch.float32):
    """LayerNorm(gelu(pre)) * gamma + beta without the activation in memory (fp32 or bf16 rows)."""
    _need_cuda(pre, gamma, beta)
    R, Cc = pre.shape
    out = torch.empty((R, Cc), dtype=out_dtype, device=pre.device)
    H.check(H.load().kx_gelu_layernorm(H.ptr(pre), H.ptr(gamma), H.ptr(beta), H.ptr(out),
                                       H.KX_BF16 if out_dtype == torch.bfloat16 else H.KX_F32, R, Cc, float(eps), _stream()),
            "kx_gelu_layernorm")
    return out


def gelu_layernorm_backward(pre, gamma, dy, eps=1e-5, dres=None, dgamma_out=None, dbeta_out=None):
    """layernorm_backward with x = gelu(pre) rebuilt on load (dx = the gradient at the activation); widths the one-pass kernel
    is not built for write the activation first."""
    _need_cuda(pre, gamma, dy, dres, dgamma_out, dbeta_out)
    R, Cc = pre.shape
    lib = H.load()
    if not lib.kx_gelu_layernorm_backward_supported(Cc):
        return layernorm_backward(gelu(pre), gamma, dy, eps, dres=dres, dgamma_out=dgamma_out, dbeta_out=dbeta_out)
    dx = torch.empty_like(pre)
    dg = dgamma_out if dgamma_out is not None else torch.empty(Cc, dtype=torch.float32, device=pre.device)
    db = dbeta_out if dbeta_out is not None else torch.empty(Cc, dtype=torch.float32, device=pre.device)
    ws = _ws(lib.kx_layernorm_backward_workspace_bytes(R, Cc), pre.device)
    H.check(lib.kx_gelu_layernorm_backward(H.ptr(pre), H.ptr(gamma), H.ptr(dy), H.ptr(dres), H.ptr(dx), H.ptr(dg), H.ptr(db), R,
                                           Cc, float(eps), H.ptr(ws), ws.numel(), _stream()), "kx_gelu_layernorm_backward")
    return dx, dg, db


def gelu_backward(pre, dg):
    _need_cuda(pre, dg)
    out = torch.empty_like(pre)
    H.check(H.load().kx_gelu_backward(H.ptr(pre), H.ptr(dg), H.ptr(out), pre.numel(), _stream()), "kx_gelu_backward")
    return out


def cross_entropy(logits, target, scale, want_grad=True):
    """logits [M,V] fp32, target [M] int64 -> (loss_rows [M], dlogits [M,V] or None)."""
    _need_cuda(logits, target)
    M, V = logits.shape
    loss = torch.empty(M, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    H.check(H.load().kx_cross_entropy(H.ptr(logits), M, V, logits.stride(0), H.ptr(target), float(scale), H.ptr(loss),
                                      H.ptr(dl), V, _stream()), "kx_cross_entropy")
    return loss, dl


def reduce_sum(x, squares=False, out=None, accumulate=False):
    _need_cuda(x, out)
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = _ws(4096, x.device)
    H.check(H.load().kx_reduce_sum(H.ptr(x), x.numel(), int(squares), H.ptr(out), int(accumulate), H.ptr(ws), ws.numel(),
                                   _stream()), "kx_reduce_sum")
    return out


def quick_gelu(pre: torch.Tensor) -> torch.Tensor:
    """HF QuickGELUActivation on a kept pre-activation (CLIP's MLP)."""
    _need_cuda(pre)
    out = torch.empty_like(pre)
    H.check(H.load().kx_quick_gelu_forward(H.ptr(pre), H.ptr(out), pre.numel(), _stream()), "kx_quick_gelu_forward")
    return out


def quick_gelu_backward(pre, dg):
    _need_cuda(pre, dg)
    out = torch.empty_like(pre)
    H.check(H.load().kx_quick_gelu_backward(H.ptr(pre), H.ptr(dg), H.ptr(out), pre.numel(), _stream()), "kx_quick_gelu_backward")
    return out


def add_rowvec(x: torch.Tensor, vec: torch.Tensor) -> torch.Tensor:
    """x [rows, cols] + vec [cols] (every row)."""
    _need_cuda(x, vec)
    out = torch.empty_like(x)
    H.check(H.load().kx_add_rowvec(H.ptr(x), H.ptr(vec), H.ptr(out), x.shape[0], x.shape[1], _stream()), "kx_add_rowvec")
    return out


def patchify(pixels: torch.Tensor, patch: int, kpad: int, bf16: bool = False) -> torch.Tensor:
    """pixels [B,3,S,S] fp32 -> patch rows [B*(S/patch)^2, kpad] (columns as Conv2d.weight.flatten(1), zero padded)."""
    _need_cuda(pixels)
    B, _, S, _ = pixels.shape
    rows = B * (S // patch) ** 2
    out = torch.empty((rows, kpad), dtype=torch.bfloat16 if bf16 else torch.float32, device=pixels.device)
    H.check(H.load().kx_patchify(H.ptr(pixels), H.ptr(out), B, S, patch, kpad, H.KX_PREC_BF16 if bf16 else H.KX_PREC_F32,
                                 _stream()), "kx_patchify")
    return out


def vit_assemble(patch_out: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, B: int) -> torch.Tensor:
    """cat(class_embedding, patch_out[b]) + position_embedding -> [B, tokens, dim] fp32."""
    _need_cuda(patch_out, cls, pos)
    tokens, dim = pos.shape
    x = torch.empty((B, tokens, dim), dtype=torch.float32, device=pos.device)
    H.check(H.load().kx_vit_assemble(H.ptr(patch_out), H.ptr(cls), H.ptr(pos), H.ptr(x), B, tokens, dim, _stream()),
            "kx_vit_assemble")
    return x


def xpos_backward_(dqkv, D, T, tables, qscale):
    """In place on the fused [M,3D] gradient: undo XPos on the q and k blocks, apply the q scale."""
    _need_cuda(dqkv)
    M = dqkv.shape[0]
    t = [H.ptr(x) for x in tables] if tables is not None else [0, 0, 0, 0]
    H.check(H.load().kx_xpos_backward(H.ptr(dqkv), M, D, T, *t, float(qscale), _stream()), "kx_xpos_backward")
    return dqkv


def embed_backward(tokens, dx, vocab, max_pos, pos_offset=0, out_embed=None, out_pos=None):
    """tokens [B,T] int64, dx [B,T,d] -> (dembed [vocab,d], dpos [max_pos,d]; rows of dpos outside [2, 2+T) are zero)."""
    _need_cuda(tokens, dx, out_embed, out_pos)
    B, T, d = dx.shape
    de = out_embed if out_embed is not None else torch.empty((vocab, d), dtype=torch.float32, device=dx.device)
    dp = out_pos.zero_() if out_pos is not None else torch.zeros((max_pos, d), dtype=torch.float32, device=dx.device)
    H.check(H.load().kx_embed_backward(H.ptr(tokens), H.ptr(dx), B, T, d, vocab, pos_offset, H.ptr(de), H.ptr(dp), _stream()),
            "kx_embed_backward")
    return de, dp


def adamw_(param, grad, m, v, step, lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0, grad_norm_sq=None, max_norm=1.0):
    _need_cuda(param, grad, m, v, grad_norm_sq)
    H.check(H.load().kx_adamw(H.ptr(param), H.ptr(grad), H.ptr(m), H.ptr(v), param.numel(), float(lr), float(betas[0]),
                              float(betas[1]), float(eps), float(weight_decay), int(step), H.ptr(grad_norm_sq),
                              float(max_norm), _stream()), "kx_adamw")


def lion_(param, grad, m, lr, betas=(0.9, 0.99), weight_decay=0.0, grad_norm_sq=None, max_norm=1.0):
    """lion_pytorch.Lion.step on flat fp32 tensors (the optimizer /root/reference/train.py:547-556 asks for)."""
    _need_cuda(param, grad, m, grad_norm_sq)
    H.check(H.load().kx_lion(H.ptr(param), H.ptr(grad), H.ptr(m), param.numel(), float(lr), float(betas[0]), float(betas[1]),
                             float(weight_decay), H.ptr(grad_norm_sq), float(max_norm), _stream()), "kx_lion")


def dropout(x: torch.Tensor, p: float, seed: int, site: int, residual: torch.Tensor | None = None) -> torch.Tensor:
    """(residual +) keep * x / (1 - p) with the Philox mask of (seed, site); on a gradient: the dropout backward."""
    _need_cuda(x, residual)
    y = torch.empty_like(x)
    H.check(H.load().kx_dropout(H.ptr(x), H.ptr(residual), H.ptr(y), x.numel(), float(p), int(seed), int(site), _stream()),
            "kx_dropout")
    return y


def dropout_mask(n: int, p: float, seed: int, site: int, device) -> torch.Tensor:
    """The keep mask (uint8 [n]) of (seed, site): test hook for the CPU autograd reference."""
    m = torch.empty(n, dtype=torch.uint8, device=device)
    H.check(H.load().kx_dropout_mask(H.ptr(m), n, float(p), int(seed), int(site), _stream()), "kx_dropout_mask")
    return m


def attention_backward(qkv, out, dout, lse, B, T, Hh, causal=True, bf16_products=False, dropout=None):
    """qkv [B*T, 3D] (q pre-scaled and XPos-rotated; fp32, or bf16 with bf16_products), out/dout [B,T,D] fp32, lse [B,H,T]
    -> dqkv [B*T, 3D] fp32."""
    _need_cuda(qkv, out, dout, lse)
    D = Hh * 64
    if qkv.dtype == torch.bfloat16 and not bf16_products:
        raise TypeError("bf16 q/k/v need bf16_products=True")
    dqkv = torch.empty(qkv.shape, dtype=torch.float32, device=qkv.device)
    delta = torch.empty((B, Hh, T), dtype=torch.float32, device=qkv.device)
    es = qkv.element_size()
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + D * es, qkv.data_ptr() + 2 * D * es
    dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + D * 4, dqkv.data_ptr() + 2 * D * 4
    if dropout is not None and bf16_products:      # train mode on the matrix-core passes (bf16 products)
        H.check(H.load().kx_attention_backward_dropout_bf16(q, k, v, H.KX_BF16 if qkv.dtype == torch.bfloat16 else H.KX_F32,
                                                            H.ptr(out), H.ptr(dout), H.ptr(lse), dq, dk, dv, H.ptr(delta), B, Hh,
                                                            T, 3 * D, T * 3 * D, D, T * D,
                                                            H.KX_ATTN_CAUSAL if causal else H.KX_ATTN_FULL, float(dropout[0]),
                                                            int(dropout[1]), int(dropout[2]), _stream()),
                "kx_attention_backward_dropout_bf16")
        return dqkv
    if dropout is not None:                        # (p, seed, site) of the forward's attention dropout
        if qkv.dtype != torch.float32:
            raise TypeError("attention dropout with fp32 products runs on fp32 q/k/v")
        H.check(H.load().kx_attention_backward_dropout(q, k, v, H.ptr(out), H.ptr(dout), H.ptr(lse), dq, dk, dv, H.ptr(delta), B,
                                                       Hh, T, 3 * D, T * 3 * D, D, T * D,
                                                       H.KX_ATTN_CAUSAL if causal else H.KX_ATTN_FULL, float(dropout[0]),
                                                       int(dropout[1]), int(dropout[2]), _stream()),
                "kx_attention_backward_dropout")
        return dqkv
    H.check(H.load().kx_attention_backward(q, k, v, H.KX_BF16 if qkv.dtype == torch.bfloat16 else H.KX_F32, H.ptr(out),
                                           H.ptr(dout), H.ptr(lse), dq, dk, dv, H.ptr(delta), B, Hh, T, 3 * D, T * 3 * D, D,
                                           T * D, H.KX_ATTN_CAUSAL if causal else H.KX_ATTN_FULL,
                                           H.KX_PREC_BF16 if bf16_products else H.KX_PREC_F32, _stream()),
            "kx_attention_backward")
    return dqkv
