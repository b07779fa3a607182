"""Training step of the multimodal model `Kosmos()` on the device (/root/reference/train.py:521) against
autograd over the CPU forward oracle: the loss, EVERY parameter gradient (tower, resampler, image_proj, decoder,
embeddings), and the parameters after two clipped AdamW steps."""
import pytest
import torch

from kosmosx.model import Kosmos
from kosmosx.training import KosmosTrainer
from oracle import train_oracle as TO
from helpers import oracle_cfg, oracle_switches, oracle_weights, rel_err, tiny_config

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _leaves(m):
    w = {k: v.clone().requires_grad_() for k, v in oracle_weights(m).items() if v.is_floating_point()}
    for k in [k for k in w if k.startswith("decoder.embed_tokens") or k.startswith("decoder.embed_positions")
              or k.startswith("decoder.output_projection")]:
        w.pop(k)                                          # aliases of embed / embed_positions / output_projection
    return w


def _batch(cfg, B, Tt, seed):
    g = torch.Generator().manual_seed(seed)
    tok = torch.randint(2, cfg.vocab, (B, Tt), generator=g)
    tok[0, min(3, Tt - 1)] = 1                            # a padding token among inputs and targets
    img = torch.randn(B, 3, cfg.vit.image, cfg.vit.image, generator=g)
    return tok, img


@pytest.mark.parametrize("B,Tt,act", [(2, 9, "gelu"), (3, 3, "quick_gelu"), (1, 30, "gelu")])
def test_multimodal_loss_and_every_gradient_match_autograd(B, Tt, act):
    cfg = tiny_config()
    cfg.vit.act = act                                     # laion's config says gelu, HF's class default quick_gelu
    m = Kosmos._from_config(cfg, seed=3, perturb=0.1).eval()
    tok, img = _batch(cfg, B, Tt, 10 + B)
    w = _leaves(m)
    ref = TO.mm_loss(w, tok, img, oracle_cfg(cfg), oracle_switches(m.switches))
    TO.backward(ref, w)
    tr = KosmosTrainer(m.to(DEV))
    loss = tr.step(tok.to(DEV), img.to(DEV), apply_update=False)
    assert abs(float(loss) - float(ref.detach())) < 2e-5 * abs(float(ref.detach())), (float(loss), float(ref.detach()))
    errs, checked = {}, 0
    for name in dict(m.named_parameters()):
        if ".B." in name:
            continue
        g = tr.grads[name].cpu()
        if w[name].grad is None:                          # post_layernorm: not on the path (HF last_hidden_state)
            assert "post_layernorm" in name and float(g.abs().max()) == 0.0, name
            continue
        r = w[name].grad.reshape(g.shape)
        errs[name] = float((g - r).abs().max() / (r.pow(2).mean().sqrt() + 1e-3))
        checked += 1
    bad = {k: v for k, v in errs.items() if not v < 3e-4}
    assert not bad, bad
    assert checked >= 2 * 16 + 2 * 10 + 2 * 18 + 12, checked
    print(f"multimodal step B={B} Tt={Tt}: {checked} gradients, worst max|d|/rms = {max(errs.values()):.2e}")


def test_two_multimodal_adamw_steps_match_torch():
    cfg = tiny_config()
    m = Kosmos._from_config(cfg, seed=4, perturb=0.1).eval()
    w = _leaves(m)
    opt = TO.make_optimizer(w, lr=1e-3)
    tr = KosmosTrainer(m.to(DEV), lr=1e-3)
    for step in range(2):
        tok, img = _batch(cfg, 2, 12, 20 + step)
        ref = TO.train_step(w, opt, tok, oracle_cfg(cfg), images=img, sw=oracle_switches(m.switches))
        loss = tr.step(tok.to(DEV), img.to(DEV))
        assert abs(float(loss) - float(ref)) < 1e-4 * abs(float(ref)), step
    params = dict(m.named_parameters())
    for n in w:
        if n not in params or w[n].grad is None:
            continue
        d = params[n].detach().cpu() - w[n].detach().reshape(params[n].shape)
        assert float(d.pow(2).mean().sqrt() / (w[n].detach().pow(2).mean().sqrt() + 1e-12)) < 3e-5, n
        assert float(d.abs().max()) <= 2.1 * 1e-3 * 2, n
    out = m(tok.to(DEV), img.to(DEV))                      # the inference path sees the updated weights
    assert out.shape == (2, 12 + cfg.perceiver.latents, cfg.vocab) and torch.isfinite(out).all()


def test_bf16_products_multimodal_gradients():
    cfg = tiny_config()
    m = Kosmos._from_config(cfg, seed=5, perturb=0.1).eval()
    tok, img = _batch(cfg, 2, 10, 31)
    w = _leaves(m)
    ref = TO.mm_loss(w, tok, img, oracle_cfg(cfg), oracle_switches(m.switches))
    TO.backward(ref, w)
    for prec, rms_tol in (("bf16x3", 3e-4), ("bf16", 4e-2)):
        tr = KosmosTrainer(m.to(DEV), precision=prec)
        loss = tr.step(tok.to(DEV), img.to(DEV), apply_update=False)
        assert abs(float(loss) - float(ref.detach())) < (1e-4 if prec == "bf16x3" else 5e-3) * abs(float(ref.detach()))
        worst = 0.0
        for name in dict(m.named_parameters()):
            if ".B." in name or w[name].grad is None:
                continue
            g, r = tr.grads[name].float().cpu(), w[name].grad.reshape(tr.grads[name].shape)
            # (+1e-5: the tower's k_proj.bias gradient is analytically zero — softmax ignores a per-query constant — so both
            # sides hold rounding noise there)
            worst = max(worst, float((g - r).pow(2).mean().sqrt() / (r.pow(2).mean().sqrt() + 1e-5)))
        print(f"multimodal {prec}: worst gradient rms error {worst:.2e}")
        assert worst < rms_tol, (prec, worst)


def test_zero_stage_3_multimodal_step_is_bitwise_the_replicated_step():
    cfg = tiny_config()
    tok, img = _batch(cfg, 2, 10, 41)
    a = Kosmos._from_config(cfg, seed=7, perturb=0.1).eval().to(DEV)
    b = Kosmos._from_config(cfg, seed=7, perturb=0.1).eval().to(DEV)
    ta = KosmosTrainer(a, lr=1e-3, checkpoint_activations=True)
    tb = KosmosTrainer(b, lr=1e-3, zero_stage=3)
    assert float(ta.step(tok.to(DEV), img.to(DEV))) == float(tb.step(tok.to(DEV), img.to(DEV)))
    tb.gather_parameters()
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    assert all(torch.equal(pa[n], pb[n]) for n in pa)


def test_multimodal_train_mode_dropout_matches_autograd_with_the_same_masks():
    cfg = tiny_config()
    m = Kosmos._from_config(cfg, seed=8, perturb=0.1).eval()
    tok, img = _batch(cfg, 2, 9, 51)
    w = _leaves(m)
    tr = KosmosTrainer(m.to(DEV), train_mode=True, dropout_seed=3)
    T = 9 + cfg.perceiver.latents
    drop = {k: v.cpu() for k, v in tr.dropout_masks(2, T).items()}
    ref = TO.mm_loss(w, tok, img, oracle_cfg(cfg), oracle_switches(m.switches), drop=drop)
    TO.backward(ref, w)
    loss = tr.step(tok.to(DEV), img.to(DEV), apply_update=False)
    assert abs(float(loss) - float(ref.detach())) < 2e-5 * abs(float(ref.detach()))
    worst = 0.0
    for name in dict(m.named_parameters()):
        if ".B." in name or w[name].grad is None:
            continue
        g, r = tr.grads[name].cpu(), w[name].grad.reshape(tr.grads[name].shape)
        e = float((g - r).abs().max() / (r.pow(2).mean().sqrt() + 1e-3))
        worst = max(worst, e)
        assert e < 3e-4, (name, e)
    print(f"multimodal train-mode step: worst gradient max|d|/rms = {worst:.2e}")


def test_trainer_argument_errors():
    cfg = tiny_config()
    m = Kosmos._from_config(cfg, seed=6).eval().to(DEV)
    tr = KosmosTrainer(m)
    tok, img = _batch(cfg, 2, 6, 1)
    with pytest.raises(TypeError):
        tr.step(tok.to(DEV), img)                          # CPU images: no fallback
    with pytest.raises(ValueError):
        tr.step(tok[:, :2].to(DEV), img.to(DEV))           # nothing after the image block
    with pytest.raises(IndexError):
        tr.step(torch.randint(2, 100, (2, 60)).to(DEV), img.to(DEV))    # 60 + 8 + 2 > 64 positions
