"""Backward / optimizer kernels against torch autograd on the CPU, fp32, through the C ABI."""
import math

import pytest
import torch

from kosmosx import grad_ops as G
from kosmosx import ops
from helpers import rel_err
from oracle import kosmos_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _g(seed):
    return torch.Generator().manual_seed(seed)


# bf16 shapes that are multiples of 8 take the 16-byte kernel (partial 64 x 64 tiles at the edges included)
@pytest.mark.parametrize("shape,dt", [((70, 130), torch.float32), ((257, 64), torch.bfloat16), ((1, 5), torch.float32),
                                      ((512, 2048), torch.bfloat16), ((72, 200), torch.bfloat16), ((8, 8), torch.bfloat16),
                                      ((136, 64), torch.bfloat16)])
def test_transpose(shape, dt):
    x = torch.randn(*shape, generator=_g(1)).to(dt)
    assert torch.equal(G.transpose(x.to(DEV)).cpu(), x.t().contiguous())


@pytest.mark.parametrize("shape", [(1, 7), (700, 130), (5000, 64)])
def test_colsum(shape):
    x = torch.randn(*shape, generator=_g(2))
    out = G.colsum(x.to(DEV))
    assert rel_err(out, x.double().sum(0).float()) < 1e-5
    out2 = G.colsum(x.to(DEV), out=out, accumulate=True)
    assert rel_err(out2, 2 * x.double().sum(0).float()) < 1e-5
    assert torch.equal(G.colsum(x.to(DEV)), G.colsum(x.to(DEV)))           # deterministic


# (114, 2048), (4096, 2048), (300, 8192), (77, 6144), (50, 3072), (1000, 1024), (2051, 4096): the one-pass kernel (cols a
# multiple of 1024: rows in registers, dgamma / dbeta partials carried across a workgroup's rows); the others the two-kernel form
@pytest.mark.parametrize("rows,cols", [(3, 64), (114, 2048), (1000, 256), (4096, 2048), (300, 8192), (77, 6144), (50, 3072),
                                       (1000, 1024), (2051, 4096), (1, 2048), (40, 5120)])
def test_layernorm_backward(rows, cols):
    g = _g(3)
    x = (torch.randn(rows, cols, generator=g) * 2 + 0.5).requires_grad_()
    gam, bet = torch.randn(cols, generator=g).requires_grad_(), torch.randn(cols, generator=g).requires_grad_()
    dy, dres = torch.randn(rows, cols, generator=g), torch.randn(rows, cols, generator=g)
    y = torch.nn.functional.layer_norm(x, (cols,), gam, bet, 1e-5)
    (y * dy).sum().backward()
    dx, dg, db = G.layernorm_backward(x.detach().to(DEV), gam.detach().to(DEV), dy.to(DEV), dres=dres.to(DEV))
    assert rel_err(dx, x.grad + dres) < 2e-5
    assert rel_err(dg, gam.grad) < 2e-5 and rel_err(db, bet.grad) < 2e-5
    dx2, dg2, db2 = G.layernorm_backward(x.detach().to(DEV), gam.detach().to(DEV), dy.to(DEV), dres=dres.to(DEV))
    assert torch.equal(dx, dx2) and torch.equal(dg, dg2) and torch.equal(db, db2)        # fixed summation order
    dx3, dg3, db3 = G.layernorm_backward(x.detach().to(DEV), gam.detach().to(DEV), dy.to(DEV), want_param_grads=False)
    assert dg3 is None and db3 is None and rel_err(dx3, x.grad) < 2e-5


@pytest.mark.parametrize("rows,cols", [(300, 8192), (50, 2048), (77, 6144), (33, 512), (20, 1024), (9, 4100)])
def test_gelu_layernorm_forward_and_backward_rebuild_the_activation(rows, cols):
    """kx_gelu_layernorm / kx_gelu_layernorm_backward == the LayerNorm kernels on a written gelu(pre), bit for bit (the same
    erf GELU on load): the training step's FFN keeps the pre-activation only."""
    g = _g(21 + cols)
    pre = (torch.randn(rows, cols, generator=g) * 2).to(DEV)
    gam, bet = torch.randn(cols, generator=g).to(DEV), torch.randn(cols, generator=g).to(DEV)
    dy = torch.randn(rows, cols, generator=g).to(DEV)
    act = G.gelu(pre)
    for dt in (torch.float32, torch.bfloat16):
        assert torch.equal(G.gelu_layernorm(pre, gam, bet, 1e-5, out_dtype=dt), ops.layernorm(act, gam, bet, 1e-5, out_dtype=dt))
    dx, dg, db = G.gelu_layernorm_backward(pre, gam, dy)
    rx, rg, rb = G.layernorm_backward(act, gam, dy)
    assert torch.equal(dx, rx) and torch.equal(dg, rg) and torch.equal(db, rb)
    ref = torch.nn.functional.layer_norm(torch.nn.functional.gelu(pre.cpu()), (cols,), gam.cpu(), bet.cpu(), 1e-5)
    assert rel_err(G.gelu_layernorm(pre, gam, bet, 1e-5), ref) < 2e-5


def test_layernorm_backward_rows_far_from_zero():
    """Row means of 1e3 with unit spread: the second moment is taken on centred values (no E[x^2] - mean^2 cancellation)."""
    g = _g(33)
    rows, cols = 600, 2048
    x = (torch.randn(rows, cols, generator=g) + 1000.0 * torch.randn(rows, 1, generator=g)).requires_grad_()
    gam, bet = torch.randn(cols, generator=g).requires_grad_(), torch.randn(cols, generator=g).requires_grad_()
    dy = torch.randn(rows, cols, generator=g)
    xd, gd, bd = x.detach().double().requires_grad_(), gam.detach().double().requires_grad_(), bet.detach().double().requires_grad_()
    (torch.nn.functional.layer_norm(xd, (cols,), gd, bd, 1e-5) * dy.double()).sum().backward()
    dx, dg, db = G.layernorm_backward(x.detach().to(DEV), gam.detach().to(DEV), dy.to(DEV))
    assert rel_err(dx, xd.grad.float()) < 1e-3          # fp32 centring of |x| ~ 1e3 leaves ~1e-4 of the unit spread
    assert rel_err(dg, gd.grad.float()) < 1e-3 and rel_err(db, bd.grad.float()) < 2e-5


@pytest.mark.parametrize("n,off", [(1, 0), (1023, 1), (4096 * 37 + 5, 3), (3_000_001, 0), (70_000_003, 2)])
def test_reduce_sum_vector_loads(n, off):
    """kx_reduce_sum reads 16 bytes per lane, four loads in flight: unaligned heads and ragged tails go one by one."""
    buf = torch.randn(n + off, generator=_g(5))
    x = buf.to(DEV)[off:]
    for sq in (False, True):
        ref = (buf[off:].double() ** 2).sum() if sq else buf[off:].double().sum()
        got = G.reduce_sum(x, squares=sq)
        assert abs(float(got) - float(ref)) <= 2e-6 * float((buf[off:].double().abs() ** (2 if sq else 1)).sum())
        assert torch.equal(got, G.reduce_sum(x, squares=sq))


def test_gelu_backward_and_cross_entropy_and_sum():
    g = _g(4)
    pre = (torch.randn(50, 300, generator=g) * 2).requires_grad_()
    dg = torch.randn(50, 300, generator=g)
    (torch.nn.functional.gelu(pre) * dg).sum().backward()
    assert rel_err(G.gelu_backward(pre.detach().to(DEV), dg.to(DEV)), pre.grad) < 1e-5
    assert rel_err(G.gelu(pre.detach().to(DEV)), torch.nn.functional.gelu(pre.detach())) < 1e-6
    odd = pre.detach().flatten()[:14999].to(DEV)                    # n % 4 != 0: the one-value-per-lane kernels
    assert torch.equal(G.gelu(odd), G.gelu(pre.detach().to(DEV)).flatten()[:14999])
    assert torch.equal(G.gelu_backward(odd, dg.flatten()[:14999].to(DEV)),
                       G.gelu_backward(pre.detach().to(DEV), dg.to(DEV)).flatten()[:14999])
    logits = (torch.randn(37, 1002, generator=g) * 3).requires_grad_()
    tgt = torch.randint(0, 1002, (37,), generator=g)
    tgt[5] = -100                                                   # ignore_index
    loss = torch.nn.functional.cross_entropy(logits, tgt, ignore_index=-100, reduction="sum")
    (loss / 36).backward()
    lr, dl = G.cross_entropy(logits.detach().to(DEV), tgt.to(DEV), 1.0 / 36)
    assert abs(float(G.reduce_sum(lr)) - float(loss.detach())) < 1e-3 * abs(float(loss.detach()))
    assert rel_err(dl, logits.grad) < 1e-5 and float(lr[5]) == 0.0
    x = torch.randn(100003, generator=g)
    assert abs(float(G.reduce_sum(x.to(DEV), squares=True)) - float((x.double() ** 2).sum())) < 1e-2


def test_xpos_backward():
    g = _g(5)
    B, T, Hh = 2, 9, 2
    D = Hh * 64
    raw = torch.randn(B * T, 3 * D, generator=g).requires_grad_()
    qc, qs = O.xpos_tables(T, 64, 512, 0, False)
    kc, ks = O.xpos_tables(T, 64, 512, 0, True)
    q = (raw[:, :D] * 0.125).view(B, T, Hh, 64).transpose(1, 2).reshape(B * Hh, T, 64)
    k = raw[:, D:2 * D].view(B, T, Hh, 64).transpose(1, 2).reshape(B * Hh, T, 64)
    q2 = O.apply_xpos(q, qc, qs).view(B, Hh, T, 64).transpose(1, 2).reshape(B * T, D)
    k2 = O.apply_xpos(k, kc, ks).view(B, Hh, T, 64).transpose(1, 2).reshape(B * T, D)
    outp = torch.cat([q2, k2, raw[:, 2 * D:]], 1)
    dy = torch.randn(B * T, 3 * D, generator=g)
    (outp * dy).sum().backward()
    tabs = [t.contiguous().to(DEV) for t in (qc, qs, kc, ks)]
    got = G.xpos_backward_(dy.clone().to(DEV), D, T, tabs, 0.125)
    assert rel_err(got, raw.grad) < 1e-5


def test_embed_backward_and_adamw():
    g = _g(6)
    B, T, d, V, P = 3, 7, 64, 50, 32
    tok = torch.randint(0, V, (B, T), generator=g)
    emb = torch.randn(V, d, generator=g).requires_grad_()
    pos = torch.randn(P, d, generator=g).requires_grad_()
    x = emb[tok] + pos[2:2 + T][None]
    dx = torch.randn(B, T, d, generator=g)
    (x * dx).sum().backward()
    de, dp = G.embed_backward(tok.to(DEV), dx.to(DEV), V, P)
    assert rel_err(de, emb.grad) < 1e-5 and rel_err(dp, pos.grad) < 1e-5
    # AdamW against torch.optim.AdamW over 3 steps, with clip_grad_norm_(1.0)
    p0 = torch.randn(1000, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    p = p0.clone().to(DEV); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g) * 3
        ref.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        gd = gr.to(DEV)
        G.adamw_(p, gd, m, v, step, 1e-2, (0.9, 0.95), 1e-8, 0.1, grad_norm_sq=G.reduce_sum(gd, squares=True), max_norm=1.0)
        assert rel_err(p, ref.detach()) < 1e-5, step
    # Lion (lion_pytorch.Lion.step restated in torch: the package is absent; /root/reference/train.py:547-556 selects it)
    pr = p0.clone(); mr = torch.zeros_like(pr)
    p = p0.clone().to(DEV); m = torch.zeros_like(p)
    lr, b1, b2, wd = 1e-3, 0.9, 0.95, 0.1
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g) * 3
        clip = min(1.0, 1.0 / (float(gr.norm()) + 1e-6))
        gc = gr * clip
        pr = pr * (1 - lr * wd)
        pr = pr - lr * torch.sign(mr * b1 + gc * (1 - b1))
        mr = mr * b2 + gc * (1 - b2)
        gd = gr.to(DEV)
        G.lion_(p, gd, m, lr, (b1, b2), wd, grad_norm_sq=G.reduce_sum(gd, squares=True), max_norm=1.0)
        assert rel_err(p, pr) < 1e-6 and rel_err(m, mr) < 1e-5, step


@pytest.mark.parametrize("B,Hh,T,causal", [(2, 2, 9, True), (1, 3, 114, True), (2, 1, 130, False), (1, 2, 200, True)])
def test_attention_backward(B, Hh, T, causal):
    g = _g(7 + T)
    D = Hh * 64
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 0.5).requires_grad_()
    q, k, v = (qkv[:, i * D:(i + 1) * D].reshape(B, T, Hh, 64).transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.triu(torch.full((T, T), float("-inf")), 1)
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, D)
    do = torch.randn(B, T, D, generator=g)
    (o * do).sum().backward()
    qd = qkv.detach().to(DEV)
    q3, k3, v3 = (qd[:, i * D:(i + 1) * D].unflatten(0, (B, T)).unflatten(2, (Hh, 64)) for i in range(3))
    lse = torch.empty(B, Hh, T, device=DEV)
    og = ops.attention(q3, k3, v3, causal, lse_out=lse)
    assert rel_err(og, o.detach()) < 2e-5
    ref_lse = torch.logsumexp(s.detach(), -1)
    assert rel_err(lse, ref_lse) < 1e-5
    dqkv = G.attention_backward(qd, og, do.to(DEV), lse, B, T, Hh, causal)
    assert rel_err(dqkv, qkv.grad) < 3e-5
    from kosmosx import _hip
    _hip.load().kx_set_tuning(2, 1)                      # the first version: LDS-tiled VALU passes
    try:
        first = G.attention_backward(qd, og, do.to(DEV), lse, B, T, Hh, causal)
    finally:
        _hip.load().kx_set_tuning(2, 0)
    assert rel_err(first, qkv.grad) < 3e-5
    # bf16 products (mixed-precision training): fp32 statistics, operands rounded on the way in
    mixed = G.attention_backward(qd, og, do.to(DEV), lse, B, T, Hh, causal, bf16_products=True)
    rms = float((mixed.cpu() - qkv.grad).pow(2).mean().sqrt() / qkv.grad.pow(2).mean().sqrt())
    assert rms < 1.5e-2 and rel_err(mixed, qkv.grad) < 0.2, rms
    # ... and with q/k/v stored in bf16: the bf16 flash forward supplies out and the log-sum-exp
    qb = qd.to(torch.bfloat16)
    qb3, kb3, vb3 = (qb[:, i * D:(i + 1) * D].unflatten(0, (B, T)).unflatten(2, (Hh, 64)) for i in range(3))
    lse16 = torch.empty(B, Hh, T, device=DEV)
    o16 = ops.attention(qb3, kb3, vb3, causal, out_dtype=torch.float32, lse_out=lse16)
    assert rel_err(lse16, ref_lse) < 2e-2 and rel_err(o16, o.detach()) < 3e-2
    stored = G.attention_backward(qb, o16, do.to(DEV), lse16, B, T, Hh, causal, bf16_products=True)
    rms = float((stored.cpu() - qkv.grad).pow(2).mean().sqrt() / qkv.grad.pow(2).mean().sqrt())
    assert rms < 2e-2, rms


@pytest.mark.parametrize("shape", [(70, 132), (64, 64), (300, 2048), (129, 8), (512, 8192)])
def test_gelu_backward_folded_into_the_operand_pair(shape):
    """kx_gelu_backward_operand_pair == kx_gelu_backward followed by kx_to_operand_pair, bit for bit (operands and the
    bias column sums): the training step's FFN backward never writes the fp32 gradient of the pre-activation."""
    g = _g(11 + sum(shape))
    pre, dg = (torch.randn(*shape, generator=g) * 2).to(DEV), torch.randn(*shape, generator=g).to(DEV)
    bias_a, bias_b = torch.empty(shape[1], device=DEV), torch.empty(shape[1], device=DEV)
    a, t = G.gelu_backward_pair(pre, dg, colsum_out=bias_a)
    ra, rt = G.to_operand_pair(G.gelu_backward(pre, dg), colsum_out=bias_b)
    assert torch.equal(a, ra) and torch.equal(t, rt) and torch.equal(bias_a, bias_b)


@pytest.mark.parametrize("shape", [(70, 132), (64, 64), (5, 260), (300, 2048), (129, 8)])
def test_to_operand_pair_matches_the_two_single_conversions(shape):
    """One pass over an fp32 matrix -> bf16 operand rows and the rows of its transpose, both zero-padded to 64."""
    x = torch.randn(*shape, generator=_g(7 + sum(shape))).to(DEV)
    a, t = G.to_operand_pair(x)
    assert torch.equal(a, G.to_operand(x, "bf16", False)) and torch.equal(t, G.to_operand(x, "bf16", True))
    cs = torch.empty(shape[1], dtype=torch.float32, device=DEV)
    a1, t1 = G.to_operand_pair(x, colsum_out=cs)                                  # + the column sums (bias gradient)
    assert torch.equal(a1, a) and torch.equal(t1, t)
    torch.testing.assert_close(cs.cpu(), x.cpu().double().sum(0).float(), rtol=2e-6, atol=2e-5)
    a2, none = G.to_operand_pair(x, transposed=False)
    none2, t2 = G.to_operand_pair(x, straight=False)
    assert none is None and none2 is None and torch.equal(a2, a) and torch.equal(t2, t)
    big = torch.randn(shape[0], shape[1] + 12, generator=_g(3)).to(DEV)           # a row-strided source view
    v = big[:, :shape[1]]
    a3, t3 = G.to_operand_pair(v)
    assert torch.equal(a3, G.to_operand(v, "bf16", False)) and torch.equal(t3, G.to_operand(v, "bf16", True))


@pytest.mark.parametrize("shape", [(70, 130), (64, 64), (5, 257), (300, 2048)])
@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("fmt", ["bf16", "bf16x3_act", "bf16x3_w"])
def test_to_operand_formats(shape, transpose, fmt):
    """fp32 matrix -> GEMM operand: optional transpose, K zero-padded to 64, bf16 or the two bf16x3 row layouts."""
    x = torch.randn(*shape, generator=_g(sum(shape)))
    o = x.t().contiguous() if transpose else x
    kp = (o.shape[1] + 63) // 64 * 64
    pad = torch.zeros(o.shape[0], kp)
    pad[:, :o.shape[1]] = o
    hi = pad.to(torch.bfloat16)
    lo = (pad - hi.float()).to(torch.bfloat16)
    ref = {"bf16": hi, "bf16x3_act": torch.cat([hi, hi, lo], 1), "bf16x3_w": torch.cat([hi, lo, hi], 1)}[fmt]
    got = G.to_operand(x.to(DEV), fmt, transpose)
    assert torch.equal(got.cpu(), ref)
    sub = G.to_operand(x.to(DEV)[:, 1:], fmt, transpose)          # a column-sliced (unaligned) source view
    o2 = x[:, 1:].t().contiguous() if transpose else x[:, 1:]
    assert torch.equal(sub.cpu()[:, :o2.shape[1]], o2.to(torch.bfloat16))
