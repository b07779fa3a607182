"""Thin torch-tensor wrappers over the primitive C-ABI ops of libkosmosx_hip.so.

PyTorch is plumbing here (device memory + the current HIP stream); the arithmetic runs in the
hand-written gfx950 kernels.  Every wrapper requires CUDA tensors and raises otherwise.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _hip as H


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("kosmosx ops run on the MI355X HIP path only: tensor is not on a CUDA/HIP device")


def _cdt(dtype) -> int:
    if dtype == torch.float32:
        return H.KX_F32
    if dtype == torch.bfloat16:
        return H.KX_BF16
    if dtype == torch.float16:
        return H.KX_F16
    raise TypeError(f"unsupported dtype {dtype}")


def pack_f16c_rows(x: torch.Tensor) -> torch.Tensor:
    """[rows, K] fp32 -> KX_F16C activation rows [rows, 4K] uint8: [fp16(x) | fp8(x) | fp8((x - fp16(x)) * 2^11)]
    (torch conversions; the device producers write the same bytes — kx_precision in include/kosmosx_hip.h)."""
    x = x.float()
    h = x.clamp(-65504.0, 65504.0).to(torch.float16)     # the fp16 piece saturates (csrc/kx_common.h: clamp_f16)
    e = x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    r = ((x - h.float()) * 2048.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return torch.cat([h.view(torch.uint8).reshape(x.shape[0], -1), e.view(torch.uint8), r.view(torch.uint8)], dim=1).contiguous()


def unpack_f16c_rows(rows: torch.Tensor, K: int):
    """KX_F16C activation rows [rows, 4K] uint8 -> (h, e, r) as fp32 tensors [rows, K] (r still carries its 2^11)."""
    h = rows[:, : 2 * K].contiguous().view(torch.float16).float()
    e = rows[:, 2 * K: 3 * K].contiguous().view(torch.float8_e4m3fn).float()
    r = rows[:, 3 * K: 4 * K].contiguous().view(torch.float8_e4m3fn).float()
    return h, e, r


def layernorm(x, gamma, beta, eps=1e-5, out_dtype=torch.float32, pre_add=None, out=None,
              rows_per_group=None, out_group_stride=0, out_row_offset=0, x3=False, f16c=False):
    """x [rows, cols] fp32 -> LN(x (+pre_add)) * gamma + beta.  x3: KX_BF16X3 rows [hi | hi | lo] ([rows, 3*cols] bf16);
    f16c: KX_F16C rows ([rows, 4*cols] uint8)."""
    _need_cuda(x, gamma, beta, pre_add, out)
    rows, cols = x.shape
    if out is None and f16c:
        out = torch.empty((rows, 4 * cols), dtype=torch.uint8, device=x.device)
    if out is None:
        out = torch.empty((rows, 3 * cols if x3 else cols), dtype=torch.bfloat16 if x3 else out_dtype, device=x.device)
    rc = H.load().kx_layernorm(H.ptr(x), H.ptr(pre_add), H.ptr(gamma), H.ptr(beta), H.ptr(out),
                               H.KX_F16C if f16c else H.KX_BF16X3 if x3 else _cdt(out.dtype),
                               rows, cols, float(eps), rows_per_group or rows, out_group_stride, out_row_offset,
                               _stream())
    H.check(rc, "kx_layernorm")
    return out


def gemm(a, w, bias=None, residual=None, act="none", out_dtype=torch.float32, qscale=1.0, qcols=0,
         xpos=None, xpos_dim=0, tile=0, out=None, row_stats=None, colsum=None, stats_out=None, splitk_ws=None,
         splitk=0, ln=None, stats_partials=None, stats_in_seg=64, stats_eps=1e-5, stats_out_seg=0, out_x3=False,
         ln_out=None, ln_operand=None, w_tiled_rows=0, ksplit=0, out2=None, residual2=None, a_add=None, out_pieces=False,
         a_pieces=False, pair_ws=None, splitk_flags=None):
    """epilogue(a [M,K] @ w[N,K]^T).  a/w both bf16 or both fp32.  xpos = (xq_cs, xq_ss, xk_cs, xk_ss) [T,32].
    tile=16 (weight streaming, bf16, M <= 16) extras: ln = (gamma, beta, eps) with `a` the raw fp32 rows;
    stats_partials [M,nseg,2] instead of row_stats; stats_out_seg=16.
    ln_out = (gamma, beta, eps, dtype) with split-K scratch on a skinny problem: also returns LayerNorm(out) (the
    row-owning reduce kernel); the call then returns (out, ln)."""
    _need_cuda(a, w, bias, residual, out)
    M, K = a.shape
    N = w_tiled_rows or w.shape[0]                 # w_tiled_rows = N: `w` is tile_weight_rows(W) (tile 16 only)
    prec = H.KX_PREC_BF16 if w.dtype == torch.bfloat16 else H.KX_PREC_F16 if w.dtype == torch.float16 else H.KX_PREC_F32
    if a.dtype != w.dtype and ln is None and w.dtype != torch.uint8:
        raise TypeError("gemm operands must share a dtype")
    if out is None:
        out = torch.empty((M, 3 * N if out_x3 else N), dtype=torch.bfloat16 if out_x3 else out_dtype, device=a.device)
    g = H.GemmArgs()
    g.A, g.lda, g.W, g.ldw = H.ptr(a), a.stride(0), H.ptr(w), (K if w_tiled_rows else w.stride(0))
    # uint8 tiles: 24-bit planes (tile_weight_rows_w24, 1536-byte blocks) or block-scaled 16-bit weights (_w16, 1088-byte blocks)
    g.w_tiled = ((3 if w.shape[-1] == 1088 else 2) if w.dtype == torch.uint8 else 1) if w_tiled_rows else 0
    g.C, g.ldc, g.cdt = H.ptr(out), out.stride(0), (H.KX_BF16X3 if out_x3 else H.KX_F16P if out_pieces else _cdt(out.dtype))
    if a_pieces:                                   # `a` holds KX_F16P rows (f16_pieces_rows, or a producer's out_pieces output)
        assert g.w_tiled == 3, "a_pieces rides on the block-scaled 16-bit planes"
        g.w_tiled = 4
    g.bias, g.residual, g.ldr = H.ptr(bias), H.ptr(residual), (residual.stride(0) if residual is not None else 0)
    g.M, g.N, g.K = M, N, K
    g.act, g.qscale, g.qcols = H.ACTS[act], float(qscale), qcols
    if xpos is not None:
        g.xq_cs, g.xq_ss, g.xk_cs, g.xk_ss = (H.ptr(t) for t in xpos)
        g.xpos_T, g.xpos_dim = xpos[0].shape[0], xpos_dim
    g.prec, g.tile = prec, tile
    g.row_stats, g.colsum, g.stats_out = H.ptr(row_stats), H.ptr(colsum), H.ptr(stats_out)
    if splitk_ws is not None:
        g.splitk_ws, g.splitk_ws_bytes, g.splitk = H.ptr(splitk_ws), splitk_ws.numel() * splitk_ws.element_size(), splitk
    if pair_ws is not None:         # scratch of the 256x256 kernel's pair split (first 4 KB zero: pair_scratch())
        g.pair_ws, g.pair_ws_bytes = H.ptr(pair_ws), pair_ws.numel() * pair_ws.element_size()
    if splitk_flags is not None:    # >= 2 x CUs int32 words (splitk_flags()): the split-K launch reduces its partials itself
        g.splitk_flags = H.ptr(splitk_flags)
    if ln is not None:
        g.ln_gamma, g.ln_beta, g.ln_eps = H.ptr(ln[0]), H.ptr(ln[1]), float(ln[2])
    if stats_partials is not None:
        g.stats_partials, g.stats_in_nseg = H.ptr(stats_partials), stats_partials.shape[1]
        g.stats_in_seg, g.stats_eps = stats_in_seg, float(stats_eps)
        if row_stats is not None and tile != 16:      # tile kernels: `row_stats` is the [M, 2] OUTPUT scratch of the finalize pass (ABI 7)
            g.row_stats, g.row_stats_scratch = None, H.ptr(row_stats)
    g.stats_out_seg = stats_out_seg
    # tile 16, the residual stream as a pair (kx_gemm_args.ksplit): out2 receives part 1's product
    g.ksplit, g.C2, g.residual2, g.a_add = ksplit, H.ptr(out2), H.ptr(residual2), H.ptr(a_add)
    lnt = None
    if ln_out is not None:
        lnt = torch.empty((M, N), dtype=ln_out[3], device=a.device)
        g.ln_out, g.ln_out_dt = H.ptr(lnt), _cdt(ln_out[3])
        g.ln_out_gamma, g.ln_out_beta, g.ln_out_eps = H.ptr(ln_out[0]), H.ptr(ln_out[1]), float(ln_out[2])
    lop = None
    if ln_operand is not None:      # dtype of the operand copy: torch.bfloat16 / torch.float16 / "f16c"
        lop = _ln_operand_buffers(g, ln_operand, M, N, a.device)
    H.check(H.load().kx_gemm(C.byref(g), _stream()), "kx_gemm")
    if lop is not None:
        return (out,) + lop
    return out if lnt is None else (out, lnt)


def _ln_operand_buffers(g, dt, M, N, device):
    """Folded pre-LayerNorm producer outputs: (operand copy of the finished rows, partial statistics [M, N/64, 2])."""
    if dt == "f16c":
        cp, g.ln_operand_dt = torch.empty((M, 4 * N), dtype=torch.uint8, device=device), H.KX_F16C
    else:
        cp, g.ln_operand_dt = torch.empty((M, N), dtype=dt, device=device), _cdt(dt)
    st = torch.zeros((M, N // 64, 2), dtype=torch.float32, device=device)
    g.ln_operand_out, g.ln_operand_stats = H.ptr(cp), H.ptr(st)
    return cp, st


def tile_weight_rows(w: torch.Tensor) -> torch.Tensor:
    """[N, K] bf16 or fp32 (K % 32 == 0) -> the streaming layout of kx_gemm_args.w_tiled: [ceil(N/16), K/ks, 64, e] with
    e = 16 bytes of values (8 bf16 / 4 fp32) and ks = 4e the k-step; piece l of block (p, c) = row 16p + (l & 15), columns
    ks*c + e*(l >> 4) .. +e-1 — one contiguous 1 KB block per wave load; rows past N are zero.  A copy (torch data movement)."""
    N, K = w.shape
    assert K % 32 == 0 and w.dtype in (torch.bfloat16, torch.float32)
    e = 16 // w.element_size()
    ks = 4 * e
    Np = (N + 15) // 16 * 16
    if Np != N:
        w = torch.cat([w, torch.zeros((Np - N, K), dtype=w.dtype, device=w.device)], 0)
    # [p, i, c, g, e] -> [p, c, g, i, e]: piece index l = g * 16 + i
    return w.view(Np // 16, 16, K // ks, 4, e).permute(0, 2, 3, 1, 4).contiguous().view(Np // 16, K // ks, 64, e)


def round_to_24_bits(w: torch.Tensor) -> torch.Tensor:
    """fp32 -> the nearest (ties to even) fp32 value whose low mantissa byte is zero: 16 significant bits, what the 24-bit
    weight planes of kx_gemm_args.w_tiled = 2 can hold."""
    b = w.detach().float().contiguous().view(torch.int32)
    b = (b + 0x7F + ((b >> 8) & 1)) & ~0xFF
    return b.view(torch.float32)


def tile_weight_rows_w24(w: torch.Tensor) -> torch.Tensor:
    """[N, K] fp32 whose values fit 24 bits (round_to_24_bits), K % 32 == 0 -> kx_gemm_args.w_tiled = 2 planes, uint8
    [ceil(N/16), K/32, 1536]: per block of 16 rows x 32 columns, 64 pieces of 16 B (piece l = row 16p + (l & 15): the bf16
    halves of columns 32c + 4(l >> 4) .. +3, then 32c + 16 + 4(l >> 4) .. +3) followed by 64 pieces of 8 B (their third bytes)."""
    N, K = w.shape
    assert K % 32 == 0 and w.dtype == torch.float32
    bits = w.contiguous().view(torch.int32)
    assert int((bits & 0xFF).abs().max()) == 0, "values must be rounded to 24 bits first (round_to_24_bits)"
    Np = (N + 15) // 16 * 16
    if Np != N:
        bits = torch.cat([bits, torch.zeros((Np - N, K), dtype=torch.int32, device=w.device)], 0)
    # [p, i, c, half, g, j] -> [p, c, g, i, half, j]: piece l = g * 16 + i holds (half, j) = 8 values
    v = bits.view(Np // 16, 16, K // 32, 2, 4, 4).permute(0, 2, 4, 1, 3, 5).contiguous()
    hi = ((v >> 16) & 0xFFFF).to(torch.int16).view(Np // 16, K // 32, 64 * 8).view(torch.uint8)      # little-endian halves
    lo = ((v >> 8) & 0xFF).to(torch.uint8).view(Np // 16, K // 32, 64 * 8)
    return torch.cat([hi.view(Np // 16, K // 32, 1024), lo], dim=2).contiguous()


def quantize_block16(w: torch.Tensor):
    """fp32 [N, K] (K % 32 == 0) -> (q int16 [N, K], scale fp32 [N, K/32], wq fp32 [N, K]): per row and block of 32 columns
    scale = max|w| / 32767 (1 for an all-zero block), q = round(w / scale), wq = q * scale — the values the block-scaled 16-bit
    planes of kx_gemm_args.w_tiled = 3 hold, as the kernel rebuilds them ((float)q * scale, one fp32 rounding)."""
    N, K = w.shape
    assert K % 32 == 0
    wb = w.detach().float().reshape(N, K // 32, 32)
    amax = wb.abs().amax(-1)
    scale = torch.where(amax > 0, amax / 32767.0, torch.ones_like(amax))
    q = torch.round(wb / scale[..., None]).clamp(-32767, 32767).to(torch.int16)
    wq = (q.float() * scale[..., None]).reshape(N, K).contiguous()
    return q.reshape(N, K).contiguous(), scale.contiguous(), wq


def tile_weight_rows_w16(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """(q int16 [N, K], scale fp32 [N, K/32]) -> kx_gemm_args.w_tiled = 3 planes, uint8 [ceil(N/16), K/32, 1088]: per block
    64 pieces of 16 B (piece l = row 16p + (l & 15): columns 32c + 4(l >> 4) .. +3, then 32c + 16 + 4(l >> 4) .. +3) followed
    by the block's 16 row scales."""
    N, K = q.shape
    assert K % 32 == 0 and q.dtype == torch.int16 and scale.shape == (N, K // 32)
    Np = (N + 15) // 16 * 16
    if Np != N:
        q = torch.cat([q, torch.zeros((Np - N, K), dtype=q.dtype, device=q.device)], 0)
        scale = torch.cat([scale, torch.ones((Np - N, K // 32), dtype=scale.dtype, device=scale.device)], 0)
    v = q.view(Np // 16, 16, K // 32, 2, 4, 4).permute(0, 2, 4, 1, 3, 5).contiguous()          # [p, c, g, i, half, j]
    qb = v.view(Np // 16, K // 32, 64 * 8).view(torch.uint8).view(Np // 16, K // 32, 1024)
    sb = scale.float().view(Np // 16, 16, K // 32).permute(0, 2, 1).contiguous().view(torch.uint8).view(Np // 16, K // 32, 64)
    return torch.cat([qb, sb], dim=2).contiguous()


def f16_pieces_rows(x: torch.Tensor) -> torch.Tensor:
    """fp32 [M, K] (K % 32 == 0) -> KX_F16P rows as an fp32-typed [M, K] tensor: value = hi + lo, hi = fp16(x) toward zero
    (saturating at 65504), lo = fp16(x - hi); per 32 values [hi: 4 chunks of (4g..4g+3, 16+4g..16+4g+3) | lo: the same]."""
    M, K = x.shape
    assert K % 32 == 0 and x.dtype == torch.float32
    xc = x.clamp(-65504.0, 65504.0)
    h = xc.to(torch.float16)
    over = h.float().abs() > xc.abs()                                 # round-to-nearest went away from zero: one step back
    h = torch.where(over, (h.view(torch.int16) - 1).view(torch.float16), h)
    lo = (x - h.float()).to(torch.float16)

    def order(t):                                                     # [M, K/32, half, g, j] -> [M, K/32, g, half, j]
        return t.reshape(M, K // 32, 2, 4, 4).permute(0, 1, 3, 2, 4).reshape(M, K // 32, 32)
    return torch.cat([order(h), order(lo)], dim=2).contiguous().view(torch.float32).reshape(M, K)


def f16_pieces_values(rows: torch.Tensor) -> torch.Tensor:
    """KX_F16P rows (fp32-typed [M, K]) -> the fp32 values hi + lo they stand for."""
    M, K = rows.shape
    p = rows.contiguous().view(torch.float16).reshape(M, K // 32, 2, 4, 2, 4).float()      # [M, blk, piece, g, half, j]
    v = p[:, :, 0] + p[:, :, 1]
    return v.permute(0, 1, 3, 2, 4).reshape(M, K)


def gemm_f16c(a_rows, w_packed, N, K, bias=None, residual=None, act="none", out_f16c=False, qscale=1.0, qcols=0,
              xpos=None, xpos_dim=0, tile=0, row_stats=None, colsum=None, stats_out=None, splitk_ws=None, splitk=0,
              ln_operand=None, pair_ws=None, out_hilo=False, stats_partials=None, stats_in_seg=64, stats_eps=1e-5, corr="both",
              splitk_flags=None):
    """KX_PREC_F16C GEMM: a_rows [M, 4K] uint8 (KX_F16C activation rows), w_packed = the flat packed weight matrix
    (N rows of 4K bytes + N scale bytes, model._operand_f16c).  Output fp32 [M, N] or KX_F16C rows [M, 4N] uint8.
    out_hilo (with xpos): the fp32-shaped output holds KX_F16HL head slots — [64 fp16 hi | 64 fp16 lo] of 2^8 x per 64 columns."""
    _need_cuda(a_rows, w_packed, bias, residual)
    M = a_rows.shape[0]
    assert a_rows.dtype == torch.uint8 and a_rows.shape[1] == 4 * K and w_packed.numel() >= N * 4 * K + N
    out = (torch.empty((M, 4 * N), dtype=torch.uint8, device=a_rows.device) if out_f16c else
           torch.empty((M, N), dtype=torch.float32, device=a_rows.device))
    g = H.GemmArgs()
    g.A, g.lda, g.W, g.ldw = H.ptr(a_rows), a_rows.stride(0) // 2, H.ptr(w_packed), 2 * K
    g.C, g.ldc, g.cdt = H.ptr(out), (2 * N if out_f16c else out.stride(0)), (H.KX_F16C if out_f16c else H.KX_F16HL if out_hilo else H.KX_F32)
    g.w_scale = w_packed.data_ptr() + N * 4 * K
    g.bias, g.residual, g.ldr = H.ptr(bias), H.ptr(residual), (residual.stride(0) if residual is not None else 0)
    g.M, g.N, g.K = M, N, K
    g.act, g.qscale, g.qcols = H.ACTS[act], float(qscale), qcols
    if xpos is not None:
        g.xq_cs, g.xq_ss, g.xk_cs, g.xk_ss = (H.ptr(t) for t in xpos)
        g.xpos_T, g.xpos_dim = xpos[0].shape[0], xpos_dim
    g.prec, g.tile = H.KX_PREC_F16C, tile
    g.row_stats, g.colsum, g.stats_out = H.ptr(row_stats), H.ptr(colsum), H.ptr(stats_out)
    if stats_partials is not None:      # with row_stats (the OUTPUT scratch, kx_gemm_args.row_stats_scratch): finalised by kx_gemm's own pass
        g.stats_partials, g.stats_in_nseg = H.ptr(stats_partials), stats_partials.shape[1]
        g.stats_in_seg, g.stats_eps = stats_in_seg, float(stats_eps)
        if row_stats is not None:
            g.row_stats, g.row_stats_scratch = None, H.ptr(row_stats)
    g.f16c_corr = {"both": 0, "weight": 1, "act": 2, "none": 3}[corr]
    if splitk_ws is not None:
        g.splitk_ws, g.splitk_ws_bytes, g.splitk = H.ptr(splitk_ws), splitk_ws.numel() * splitk_ws.element_size(), splitk
    if pair_ws is not None:
        g.pair_ws, g.pair_ws_bytes = H.ptr(pair_ws), pair_ws.numel() * pair_ws.element_size()
    if splitk_flags is not None:
        g.splitk_flags = H.ptr(splitk_flags)
    lop = _ln_operand_buffers(g, ln_operand, M, N, a_rows.device) if ln_operand is not None else None
    H.check(H.load().kx_gemm(C.byref(g), _stream()), "kx_gemm")
    return out if lop is None else (out,) + lop


def pair_scratch(device="cuda", workgroups=256):
    """kx_gemm_args.pair_ws: 4 KB of hand-off words (zero now, zero again after every completed call) + one 128 KB slab per
    workgroup of the pair split."""
    return torch.zeros(4096 + workgroups * 131072, dtype=torch.uint8, device=device)


def set_objective(name: str) -> None:
    """Scheduling objective of the library's automatic kernel choice (kx_set_tuning key 18): "latency" (default; every launch
    chosen to finish soonest alone on the chip — one step at a time) or "throughput" (the caller keeps two or more steps in flight
    on separate streams: fewest CU-microseconds per launch).  Results are bit-identical either way."""
    v = {"latency": 0, "throughput": 1}[name]
    if H.load().kx_set_tuning(18, v) != 0:
        raise RuntimeError("kx_set_tuning(18) refused")


def splitk_flags(device="cuda"):
    """kx_gemm_args.splitk_flags: one word per workgroup of an in-launch split-K reduction (2 per CU), cleared once."""
    return torch.zeros(1024, dtype=torch.int32, device=device)


def pair_split_errors() -> int:
    """kx_pair_split_errors: 0 when every pair-split hand-off since the last call met its partner, else 1 + the index of the
    last workgroup whose bounded poll gave up (the word is cleared).  Synchronises the device: diagnostics only."""
    import ctypes
    w = ctypes.c_uint(0)
    H.check(H.load().kx_pair_split_errors(ctypes.byref(w)), "kx_pair_split_errors")
    return int(w.value)


def row_stats_finalize(partials, seg_size, eps=1e-5):
    """partials [rows, nseg, 2] (sum, M2 about the segment mean) -> [rows, 2] (mean, rstd)."""
    _need_cuda(partials)
    rows, nseg, _ = partials.shape
    out = torch.empty((rows, 2), dtype=torch.float32, device=partials.device)
    H.check(H.load().kx_row_stats_finalize(H.ptr(partials), rows, nseg, seg_size, float(eps), H.ptr(out), _stream()),
            "kx_row_stats_finalize")
    return out


def attention(q, k, v, causal=False, out_dtype=None, stats_out=None, out_x3=False, lse_out=None, f16c=False,
              out_f16c=False, dropout=None, hilo=False):
    """q [B,Tq,H,64], k/v [B,Tk,H,64] (any row/batch strides, last two dims contiguous) -> [B,Tq,H*64]
    (out_x3, fp32 inputs only: KX_BF16X3 rows [hi | hi | lo], [B,Tq,3*H*64] bf16).
    f16c (fp32 inputs): the KX_PREC_F16C kernel — split fp16 (hi, lo) products; out_f16c: KX_F16C rows [B,Tq,4*H*64] uint8."""
    _need_cuda(q, k, v)
    B, Tq, Hh, hd = q.shape
    Tk = k.shape[1]
    assert hd == 64 and q.stride(3) == 1 and q.stride(2) == 64 and k.stride(2) == 64 and v.stride(2) == 64
    assert k.stride(0) == v.stride(0) and k.stride(1) == v.stride(1)
    prec = (H.KX_PREC_BF16 if q.dtype == torch.bfloat16 else H.KX_PREC_F16 if q.dtype == torch.float16
            else (H.KX_PREC_F16CHL if hilo else H.KX_PREC_F16C if f16c or out_f16c else H.KX_PREC_F32))   # hilo: fp32-typed KX_F16HL rows
    out = (torch.empty((B, Tq, 4 * Hh * 64), dtype=torch.uint8, device=q.device) if out_f16c else
           torch.empty((B, Tq, 3 * Hh * 64), dtype=torch.bfloat16, device=q.device) if out_x3 else
           torch.empty((B, Tq, Hh * 64), dtype=out_dtype or q.dtype, device=q.device))
    a = H.AttnArgs()
    a.q, a.q_batch_stride, a.q_row_stride = H.ptr(q), q.stride(0), q.stride(1)
    a.k, a.v, a.kv_batch_stride, a.kv_row_stride = H.ptr(k), H.ptr(v), k.stride(0), k.stride(1)
    a.out, a.out_batch_stride, a.out_row_stride = H.ptr(out), out.stride(0), out.stride(1)
    if out_f16c:                                   # strides count 2-byte units
        a.out_batch_stride, a.out_row_stride = out.stride(0) // 2, out.stride(1) // 2
    a.odt = H.KX_F16C if out_f16c else H.KX_BF16X3 if out_x3 else _cdt(out.dtype)
    a.B, a.H, a.Tq, a.Tk = B, Hh, Tq, Tk
    a.mask, a.prec = (H.KX_ATTN_CAUSAL if causal else H.KX_ATTN_FULL), prec
    a.stats_out = H.ptr(stats_out)
    a.lse_out = H.ptr(lse_out)
    if dropout is not None:                        # (p, seed, site): attention dropout of the training step, fp32 only
        a.dropout_p, a.dropout_seed, a.dropout_site = float(dropout[0]), int(dropout[1]), int(dropout[2])
    H.check(H.load().kx_attention(C.byref(a), _stream()), "kx_attention")
    return out


def embed_splice(tokens, embed, pos, img=None, u1_alias=True, splice_at=2, pos_offset=0):
    """Decoder input assembly (see kx_embed_splice in include/kosmosx_hip.h)."""
    _need_cuda(tokens, embed, pos, img)
    if tokens is not None:
        B, Tt = tokens.shape
    else:
        B, Tt = img.shape[0], 0
    n_img = 0 if img is None else img.shape[1]
    d = embed.shape[1]
    out = torch.empty((B, Tt + n_img, d), dtype=torch.float32, device=embed.device)
    rc = H.load().kx_embed_splice(H.ptr(tokens), H.ptr(embed), H.ptr(pos), H.ptr(img), H.ptr(out), B, Tt, n_img, d,
                                  embed.shape[0], pos.shape[0], splice_at, int(u1_alias), pos_offset, _stream())
    H.check(rc, "kx_embed_splice")
    return out
