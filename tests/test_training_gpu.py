"""Training step on the device against autograd + torch.optim.AdamW on the CPU oracle: the loss,
EVERY parameter gradient, and the parameters after two clipped AdamW steps."""
import math

import pytest
import torch

from kosmosx.model import KosmosLanguage
from kosmosx.training import LanguageModelTrainer
from oracle import kosmos_oracle as O
from oracle import train_oracle as TO
from helpers import oracle_weights, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tiny_lm(seed=1):
    return KosmosLanguage(vocab_size=1002, dim=256, depth=2, ffn_dim=512, decoder_heads=4, _seed=seed, _perturb=0.1,
                          _max_positions=128).eval()


def _leaf_weights(lm):
    w = {k: v.clone().requires_grad_() for k, v in oracle_weights(lm).items()}
    if "decoder.embed_tokens.weight" in w:            # tied alias of embed.weight: one leaf
        w.pop("decoder.embed_tokens.weight")
    for k in [k for k in w if k.startswith("decoder.embed_positions") or k.startswith("decoder.output_projection")]:
        w.pop(k)
    return w


@pytest.mark.parametrize("B,T", [(3, 20), (2, 70), (1, 2), (1, 126)])
def test_loss_and_every_gradient_match_autograd(B, T):
    lm = _tiny_lm()
    tok = torch.randint(2, 1002, (B, T), generator=torch.Generator().manual_seed(B))
    tok[0, min(3, T - 1)] = 1                                   # a padding token in the inputs / targets
    cfg = O.DecoderCfg(layers=2, dim=256, ffn=512, heads=4, vocab=1002, max_pos=128)
    w = _leaf_weights(lm)
    ref_loss = TO.lm_loss(w, tok, cfg)
    TO.backward(ref_loss, w)
    tr = LanguageModelTrainer(lm.to(DEV))
    loss = tr.step(tok.to(DEV), apply_update=False)
    assert abs(float(loss) - float(ref_loss.detach())) < 2e-5 * abs(float(ref_loss.detach())), (float(loss), float(ref_loss.detach()))
    names = dict(lm.named_parameters()).keys()
    checked, errs = 0, {}
    for name in names:
        if ".B." in name:
            continue
        g_ref = w[name].grad
        assert name in tr.grads, name
        # (B=1, T=2: the one predicting position attends to a single key, so dq = dk = 0 exactly — absolute floor)
        e = float((tr.grads[name].cpu() - g_ref).abs().max() / (g_ref.pow(2).mean().sqrt() + 1e-3))
        errs[name] = e
        checked += 1
    bad = {k: v for k, v in errs.items() if not v < 2e-4}
    assert not bad, bad
    assert checked >= 2 * 18 + 5


def test_two_clipped_adamw_steps_match_torch():
    lm = _tiny_lm(seed=2)
    cfg = O.DecoderCfg(layers=2, dim=256, ffn=512, heads=4, vocab=1002, max_pos=128)
    w = _leaf_weights(lm)
    opt = TO.make_optimizer(w, lr=1e-3)
    tr = LanguageModelTrainer(lm.to(DEV), lr=1e-3)
    g = torch.Generator().manual_seed(9)
    for step in range(2):
        tok = torch.randint(2, 1002, (2, 24), generator=g)
        ref_loss = TO.train_step(w, opt, tok, cfg)
        loss = tr.step(tok.to(DEV))
        assert abs(float(loss) - float(ref_loss)) < 1e-4 * abs(float(ref_loss)), step
    params = dict(lm.named_parameters())
    # Adam's first steps move every element by ~lr * sign(g): an element whose gradient is rounding noise around zero
    # can legitimately step the other way, so the bound is RMS-based, with the worst element capped by 2 * lr * steps.
    for n in w:
        if n not in params:
            continue
        d = params[n].detach().cpu() - w[n].detach()
        assert float(d.pow(2).mean().sqrt() / w[n].detach().pow(2).mean().sqrt()) < 2e-5, n
        assert float(d.abs().max()) <= 2.1 * 1e-3 * 2, n
    # and the updated model still runs the inference path (packed copies were invalidated)
    out = lm(tok.to(DEV))
    assert out.shape == (2, 24, 1002) and torch.isfinite(out).all()


@pytest.mark.parametrize("prec,rms_tol,max_tol", [("bf16x3", 2e-4, 2e-3), ("bf16", 3e-2, 3e-1)])
def test_mixed_precision_gradients(prec, rms_tol, max_tol):
    """bf16 MFMA arithmetic on fp32 master weights: bf16x3 (split operands) keeps fp32-class gradients, plain bf16 the
    usual mixed-precision error.  Every parameter gradient vs autograd (fp32) on the CPU oracle."""
    lm = _tiny_lm(seed=3)
    tok = torch.randint(2, 1002, (2, 40), generator=torch.Generator().manual_seed(5))
    cfg = O.DecoderCfg(layers=2, dim=256, ffn=512, heads=4, vocab=1002, max_pos=128)
    w = _leaf_weights(lm)
    ref_loss = TO.lm_loss(w, tok, cfg)
    TO.backward(ref_loss, w)
    tr = LanguageModelTrainer(lm.to(DEV), precision=prec)
    loss = tr.step(tok.to(DEV), apply_update=False)
    assert abs(float(loss) - float(ref_loss.detach())) < (1e-4 if prec == "bf16x3" else 5e-3) * abs(float(ref_loss.detach()))
    worst_rms, worst_max = 0.0, 0.0
    for name in dict(lm.named_parameters()):
        if ".B." in name:
            continue
        g, r = tr.grads[name].float().cpu(), w[name].grad
        rms = float((g - r).pow(2).mean().sqrt() / (r.pow(2).mean().sqrt() + 1e-30))
        worst_rms, worst_max = max(worst_rms, rms), max(worst_max, rel_err(g, r))
    print(f"{prec}: worst gradient error rms {worst_rms:.2e}, max/rms {worst_max:.2e}")
    assert worst_rms < rms_tol and worst_max < max_tol


def test_sharded_update_over_rccl_single_rank_matches_local_update():
    """The data-parallel exchange (reduce-scatter, norm all-reduce, all-gather) driven through RCCL with one rank:
    same parameters as the local update, bit for bit (sum over one rank is the identity)."""
    import socket
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    tok = torch.randint(2, 1002, (2, 24), generator=torch.Generator().manual_seed(11)).to(DEV)
    a, b = _tiny_lm(seed=4).to(DEV), _tiny_lm(seed=4).to(DEV)
    ta = LanguageModelTrainer(a, lr=1e-3, precision="bf16")
    la = [float(ta.step(tok)) for _ in range(2)]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        tb = LanguageModelTrainer(b, lr=1e-3, precision="bf16", force_collectives=True)
        lb = [float(tb.step(tok)) for _ in range(2)]
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert la == lb
    assert torch.equal(ta.flat_p, tb.flat_p)


def test_full_size_gradient_families_match_autograd():
    """parity of gradient families at the REAL size (24 layers, 2048-d, vocab 32002; B=1, T=24 so the
    CPU autograd reference stays at seconds): first / last layer fc2, q | k | v, the sub-LayerNorms, the output projection
    and the embedding rows the batch touches."""
    lm = KosmosLanguage(vocab_size=32002, dim=2048, _seed=7, _perturb=0.05).eval()
    tok = torch.randint(2, 32002, (1, 24), generator=torch.Generator().manual_seed(8))
    cfg = O.DecoderCfg(vocab=32002)
    w = _leaf_weights(lm)
    ref_loss = TO.lm_loss(w, tok, cfg)
    TO.backward(ref_loss, w)
    tr = LanguageModelTrainer(lm.to(DEV))
    loss = tr.step(tok.to(DEV), apply_update=False)
    assert abs(float(loss) - float(ref_loss.detach())) < 2e-5 * abs(float(ref_loss.detach()))
    fam = ["output_projection.weight", "embed.weight", "embed_positions.weight", "decoder.layer_norm.weight"]
    for li in (0, 11, 23):
        p = f"decoder.layers.{li}."
        fam += [p + "ffn.A.fc2.weight", p + "ffn.A.fc1.weight", p + "self_attn.q_proj.A.weight", p + "self_attn.k_proj.A.weight",
                p + "self_attn.v_proj.A.weight", p + "self_attn.out_proj.A.weight", p + "self_attn.inner_attn_ln.A.weight",
                p + "ffn.A.ffn_layernorm.weight", p + "self_attn_layer_norm.A.bias"]
    worst = 0.0
    for name in fam:
        g, r = tr.grads[name].cpu(), w[name].grad
        if name == "embed.weight":                        # sparse: the rows of the batch's tokens (the rest is exactly zero)
            rows = tok.unique()
            other = g.clone()
            other[rows] = 0
            assert float(other.abs().max()) == 0.0
            g, r = g[rows], r[rows]
        elif name == "embed_positions.weight":
            g, r = g[2:26], r[2:26]
        e = float((g - r).abs().max() / (r.pow(2).mean().sqrt() + 1e-6))
        worst = max(worst, e)
        assert e < 1e-3, (name, e)
    print(f"24L/2048d gradient families vs autograd: worst max|d|/rms = {worst:.2e}")


def _two_rank_trainer_worker(rank, world, port, q, zero_stage=1):
    """One data-parallel rank of the WHOLE trainer (forward, backward, reduce-scatter, clip, AdamW slice, all-gather);
    both ranks share the box's one GPU, so the process group is gloo (RCCL refuses two ranks per device)."""
    import os
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    for p in (str(root), str(root / "kosmos-x_amd"), str(root / "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KOSMOSX_NO_LOGGING_CONFIG="1")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        lm = _tiny_lm(seed=2).to(DEV)                       # same seed on every rank = replicated start
        tr = LanguageModelTrainer(lm, lr=1e-3, zero_stage=zero_stage)
        n_params = sum(p.numel() for p in lm.parameters()) if zero_stage == 1 else sum(math.prod(s) for s in tr._shapes.values())
        if zero_stage == 1:
            assert tr.zero.world == world and tr.m.numel() * world == tr.flat_p.numel()  # 1/world of the moments per rank
        else:                                               # 1/world of parameters, gradients and both moments per rank
            for t in (tr.shard_p, tr.shard_g, tr.m, tr.v):
                assert n_params / world <= t.numel() <= n_params / world + 16 * (1 + len(lm.decoder.layers))
            assert all(p.numel() == 0 for p in lm.parameters())
        g = torch.Generator().manual_seed(9)
        losses = []
        for _ in range(2):
            tok = torch.randint(2, 1002, (2 * world, 24), generator=g)
            losses.append(float(tr.step(tok[2 * rank:2 * rank + 2].to(DEV))))
        if zero_stage == 3:
            assert not tr._pfull and not tr._gfull and all(p.numel() == 0 for p in lm.parameters())   # nothing full is left
            tr.gather_parameters()
        torch.cuda.synchronize()
        q.put((rank, losses, {n: p.detach().cpu().numpy() for n, p in lm.named_parameters()}))   # by value
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("zero_stage", [1, 3])
def test_whole_trainer_on_two_ranks_matches_single_process_autograd(zero_stage):
    """LanguageModelTrainer.step on two data-parallel ranks (each half of the batch) against
    single-process autograd + torch.optim.AdamW on the full batch: the mean of the rank losses, and every parameter after
    two clipped steps — identical on both ranks.  zero_stage 3: parameters, gradients and moments sharded
    (config/zero3.json), a layer's weights gathered for its forward and again for its recompute + backward."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_trainer_worker, args=(r, 2, port, q, zero_stage)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, losses, params = q.get(timeout=600)
        got[r] = (losses, {n: torch.from_numpy(v) for n, v in params.items()})
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    lm = _tiny_lm(seed=2)
    cfg = O.DecoderCfg(layers=2, dim=256, ffn=512, heads=4, vocab=1002, max_pos=128)
    w = _leaf_weights(lm)
    opt = TO.make_optimizer(w, lr=1e-3)
    g = torch.Generator().manual_seed(9)
    for step in range(2):
        tok = torch.randint(2, 1002, (4, 24), generator=g)
        ref_loss = float(TO.train_step(w, opt, tok, cfg))
        mean = 0.5 * (got[0][0][step] + got[1][0][step])     # equal shards: the mean of the rank means
        assert abs(mean - ref_loss) < 1e-4 * abs(ref_loss), (step, mean, ref_loss)
    for n in w:
        if n not in got[0][1]:
            continue
        assert torch.equal(got[0][1][n], got[1][1][n]), n    # the all-gather leaves every rank with the same parameters
        d = got[0][1][n] - w[n].detach()
        assert float(d.pow(2).mean().sqrt() / w[n].detach().pow(2).mean().sqrt()) < 2e-5, n
        assert float(d.abs().max()) <= 2.1 * 1e-3 * 2, n


def test_zero_stage_3_single_rank_is_bitwise_the_replicated_step():
    """Stage 3 on one rank runs the same kernels on the same values (gather = copy, reduce-scatter = copy): identical
    losses and parameters after two steps, with activation recompute on both sides."""
    tok = torch.randint(2, 1002, (2, 24), generator=torch.Generator().manual_seed(11)).to(DEV)
    a, b = _tiny_lm(seed=4).to(DEV), _tiny_lm(seed=4).to(DEV)
    ta = LanguageModelTrainer(a, lr=1e-3, precision="bf16", checkpoint_activations=True)
    tb = LanguageModelTrainer(b, lr=1e-3, precision="bf16", zero_stage=3)
    assert [float(ta.step(tok)) for _ in range(2)] == [float(tb.step(tok)) for _ in range(2)]
    tb.gather_parameters()
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    assert all(torch.equal(pa[n], pb[n]) for n in pa)
    out = b(tok)                                           # gathered parameters serve the inference path
    assert torch.isfinite(out).all()
    assert abs(float(tb.step(tok)) - float(ta.step(tok))) == 0.0      # and the next step shards them again


def test_dropout_ops_masks_are_a_function_of_seed_site_and_index():
    """kx_dropout / kx_dropout_mask / attention dropout: Philox4x32-10 masks — the exported mask is the one the kernels
    apply, the keep rate is 1 - p, another site or seed gives another mask, and the attention forward / backward with the
    mask equal autograd through softmax * mask."""
    from kosmosx import grad_ops as G, ops
    x = torch.randn(4096 * 64, generator=torch.Generator().manual_seed(0)).to(DEV) + 3.0
    keep = G.dropout_mask(x.numel(), 0.1, 1234, 7, DEV)
    y = G.dropout(x, 0.1, 1234, 7)
    assert torch.equal(y, torch.where(keep.bool(), x * (1.0 / (1.0 - 0.1)), torch.zeros_like(x)))
    assert abs(float(keep.float().mean()) - 0.9) < 3e-3
    assert not torch.equal(keep, G.dropout_mask(x.numel(), 0.1, 1234, 8, DEV))
    assert not torch.equal(keep, G.dropout_mask(x.numel(), 0.1, 1235, 7, DEV))
    assert torch.equal(keep[:1000], G.dropout_mask(1000, 0.1, 1234, 7, DEV))          # a function of the index, not of n
    r = torch.randn_like(x)
    assert torch.equal(G.dropout(x, 0.1, 1234, 7, residual=r), r + y)
    # attention: B=2, H=3, T=70 (partial tiles), causal, p = 0.25
    B, Hh, T, D = 2, 3, 70, 192
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 0.5).to(DEV)
    dout = torch.randn(B, T, D, generator=g).to(DEV)
    q3, k3, v3 = (qkv[:, i * D:(i + 1) * D].unflatten(0, (B, T)).unflatten(2, (Hh, 64)) for i in range(3))
    lse = torch.empty(B, Hh, T, device=DEV)
    out = ops.attention(q3, k3, v3, True, out_dtype=torch.float32, lse_out=lse, dropout=(0.25, 99, 4))
    mask = G.dropout_mask(B * Hh * T * T, 0.25, 99, 4, DEV).view(B, Hh, T, T).float().cpu() / 0.75
    qr = qkv.detach().cpu().clone().requires_grad_()
    qq, kk, vv = (qr[:, i * D:(i + 1) * D].view(B, T, Hh, 64).transpose(1, 2) for i in range(3))
    sc = qq @ kk.transpose(-1, -2) + torch.triu(torch.full((T, T), float("-inf")), 1)
    pr = torch.softmax(sc, -1)
    ref = ((pr * mask) @ vv).transpose(1, 2).reshape(B, T, D)
    assert float((out.cpu() - ref.detach()).abs().max()) < 2e-5
    assert float((lse.cpu() - torch.logsumexp(sc, -1).detach()).abs().max()) < 2e-5
    ref.backward(dout.cpu())
    dqkv = G.attention_backward(qkv, out, dout, lse, B, T, Hh, True, dropout=(0.25, 99, 4))
    assert float((dqkv.cpu() - qr.grad).abs().max() / qr.grad.pow(2).mean().sqrt()) < 1e-4


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("T", [72, 132, 320, 70, 114, 131])
def test_attention_dropout_on_the_matrix_core_kernels(T, causal):
    """Train mode in bf16: the flash forward and both backward passes carry the Philox mask themselves (one block per
    query and four keys — two when T % 4 != 0 (70, 114 = the multimodal decoder's length, 131): a row of the mask then
    starts inside a block; the dK/dV pass transposes keep bits inside lane quads).  Reference: fp64 autograd over the SAME
    bf16-rounded q/k/v with the exported mask — what is left is the rounding of P / dS / dO to bf16, far below what one
    wrong mask bit costs (>= 1.33 * v / T on an output row)."""
    from kosmosx import grad_ops as G, ops
    B, Hh, D, p = 2, 3, 192, 0.25
    g = torch.Generator().manual_seed(7 + T)
    qkv16 = (torch.randn(B * T, 3 * D, generator=g) * 0.7).to(torch.bfloat16)
    dout = torch.randn(B, T, D, generator=g)
    qb = qkv16.to(DEV)
    q3, k3, v3 = (qb[:, i * D:(i + 1) * D].unflatten(0, (B, T)).unflatten(2, (Hh, 64)) for i in range(3))
    lse = torch.empty(B, Hh, T, device=DEV)
    out = ops.attention(q3, k3, v3, causal, out_dtype=torch.float32, lse_out=lse, dropout=(p, 4321, 9))
    mask = G.dropout_mask(B * Hh * T * T, p, 4321, 9, DEV).view(B, Hh, T, T).double().cpu() / (1.0 - p)
    qr = qkv16.double().requires_grad_()
    qq, kk, vv = (qr[:, i * D:(i + 1) * D].view(B, T, Hh, 64).transpose(1, 2) for i in range(3))
    sc = qq @ kk.transpose(-1, -2)
    if causal:
        sc = sc + torch.triu(torch.full((T, T), float("-inf"), dtype=torch.float64), 1)
    ref = ((torch.softmax(sc, -1) * mask) @ vv).transpose(1, 2).reshape(B, T, D)
    assert rel_err(out, ref.detach().float()) < 1e-2                       # one wrong bit: >= 2.5e-2 at T = 132
    assert float((lse.cpu().double() - torch.logsumexp(sc, -1).detach()).abs().max()) < 2e-2
    ref.backward(dout.double())
    for src in (qb, qb.float()):                                           # q/k/v as stored in bf16, or fp32 rounded on the way in
        dqkv = G.attention_backward(src, out, dout.to(DEV), lse, B, T, Hh, causal, bf16_products=True, dropout=(p, 4321, 9))
        for i, name in enumerate("qkv"):
            got, want = dqkv[:, i * D:(i + 1) * D].cpu().double(), qr.grad[:, i * D:(i + 1) * D]
            rms = float((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
            assert rms < 1.2e-2, (name, rms)
    # the same launch twice: masks are a function of (seed, site, index), nothing else
    assert torch.equal(out, ops.attention(q3, k3, v3, causal, out_dtype=torch.float32, lse_out=lse, dropout=(p, 4321, 9)))


@pytest.mark.parametrize("T", [32, 30])
def test_train_mode_in_bf16_matches_autograd_with_the_same_masks(T):
    """precision="bf16", train_mode=True: the matrix-core attention kernels with the mask inside (T = 30: rows of the mask
    that straddle Philox blocks); every gradient against autograd over the oracle with the exported masks at the
    mixed-precision tolerance of test_mixed_precision_gradients."""
    lm = _tiny_lm(seed=8)
    cfg = O.DecoderCfg(layers=2, dim=256, ffn=512, heads=4, vocab=1002, max_pos=128)
    w = _leaf_weights(lm)
    tr = LanguageModelTrainer(lm.to(DEV), precision="bf16", train_mode=True, dropout_seed=11)
    tok = torch.randint(2, 1002, (2, T), generator=torch.Generator().manual_seed(13))
    drop = {k: v.cpu() for k, v in tr.dropout_masks(2, T).items()}
    ref = TO.lm_loss(w, tok, cfg, drop=drop)
    TO.backward(ref, w)
    loss = tr.step(tok.to(DEV), apply_update=False)
    assert abs(float(loss) - float(ref.detach())) < 5e-3 * abs(float(ref.detach()))
    worst = 0.0
    for name in dict(lm.named_parameters()):
        if ".B." in name:
            continue
        gq, r = tr.grads[name].float().cpu(), w[name].grad
        worst = max(worst, float((gq - r).pow(2).mean().sqrt() / (r.pow(2).mean().sqrt() + 1e-30)))
    assert worst < 3e-2, worst


@pytest.mark.parametrize("zero_stage", [1, 3])
def test_train_mode_dropout_matches_autograd_with_the_same_masks(zero_stage):
    """train_mode=True (the reference's model.train(): dropout = attention_dropout = 0.1): the loss, every gradient and the
    parameters after two steps against autograd over the oracle given the masks the kernels draw (dropout_masks())."""
    lm = _tiny_lm(seed=6)
    cfg = O.DecoderCfg(layers=2, dim=256, ffn=512, heads=4, vocab=1002, max_pos=128)
    w = _leaf_weights(lm)
    opt = TO.make_optimizer(w, lr=1e-3)
    tr = LanguageModelTrainer(lm.to(DEV), lr=1e-3, train_mode=True, dropout_seed=5, zero_stage=zero_stage)
    g = torch.Generator().manual_seed(12)
    seen = []
    for step in range(2):
        tok = torch.randint(2, 1002, (2, 30), generator=g)
        drop = {k: v.cpu() for k, v in tr.dropout_masks(2, 30).items()}
        assert set(drop) == {0, 1, 2, 3, 4, 5, 6}
        seen.append(drop[2])
        opt.zero_grad()
        ref = TO.lm_loss(w, tok, cfg, drop=drop)
        TO.backward(ref, w)
        loss = tr.step(tok.to(DEV), apply_update=False)
        assert abs(float(loss) - float(ref.detach())) < 2e-5 * abs(float(ref.detach())), (step, float(loss), float(ref.detach()))
        if zero_stage == 1:
            for name in dict(lm.named_parameters()):
                if ".B." in name:
                    continue
                e = float((tr.grads[name].cpu() - w[name].grad).abs().max() / (w[name].grad.pow(2).mean().sqrt() + 1e-3))
                assert e < 3e-4, (step, name, e)
        torch.nn.utils.clip_grad_norm_(list(w.values()), 1.0)
        opt.step()
        tr._update()
    assert not torch.equal(seen[0], seen[1])               # a new mask every step
    tr.gather_parameters()
    params = dict(lm.named_parameters())
    for n in w:
        if n in params:
            d = params[n].detach().cpu() - w[n].detach()
            assert float(d.pow(2).mean().sqrt() / w[n].detach().pow(2).mean().sqrt()) < 2e-5, n


def test_trainer_argument_errors():
    lm = _tiny_lm().to(DEV)
    tr = LanguageModelTrainer(lm)
    with pytest.raises(TypeError):
        tr.step(torch.zeros(2, 8, dtype=torch.long))                 # CPU tensor: no fallback
    with pytest.raises(ValueError):
        tr.step(torch.zeros(2, 1, dtype=torch.long, device=DEV))     # nothing to predict
    with pytest.raises(IndexError):
        tr.step(torch.zeros(1, 127, dtype=torch.long, device=DEV))   # 127 + 2 > the 128-row position table
    with pytest.raises(ValueError):
        LanguageModelTrainer(lm, precision="fp8")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        LanguageModelTrainer(_tiny_lm())


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "bf16"])
def test_repeated_steps_overfit_a_fixed_batch(prec):
    """Property: the step trains.  40 steps on one fixed batch drive the loss from ~ln(V) to well below half of it."""
    lm = _tiny_lm(seed=6).to(DEV)
    tr = LanguageModelTrainer(lm, lr=3e-3, precision=prec)
    tok = torch.randint(2, 1002, (4, 32), generator=torch.Generator().manual_seed(12)).to(DEV)
    losses = [float(tr.step(tok)) for _ in range(40)]
    assert 6.0 < losses[0] < 8.0                       # ln(1002) = 6.9 at initialisation
    assert losses[-1] < 0.4 * losses[0], (losses[0], losses[-1])
    assert all(torch.isfinite(p).all() for p in lm.parameters())


def test_activation_checkpointing_gives_identical_gradients():
    """Recomputing each layer's forward before its backward is the same arithmetic: bit-identical loss and gradients."""
    tok = torch.randint(2, 1002, (2, 40), generator=torch.Generator().manual_seed(13)).to(DEV)
    a, b = _tiny_lm(seed=7).to(DEV), _tiny_lm(seed=7).to(DEV)
    ta = LanguageModelTrainer(a, precision="bf16")
    tb = LanguageModelTrainer(b, precision="bf16", checkpoint_activations=True)
    la, lb = ta.step(tok, apply_update=False), tb.step(tok, apply_update=False)
    assert float(la) == float(lb) and torch.equal(ta.flat_g, tb.flat_g)
