import sys
sys.path[:0] = [".", "kosmos-x_amd", "tests"]
import torch
from kosmosx.model import Kosmos
from kosmosx.training import KosmosTrainer
from oracle import train_oracle as TO
from helpers import oracle_cfg, oracle_switches, oracle_weights, tiny_config
cfg = tiny_config()
import os
m = Kosmos._from_config(cfg, seed=int(os.environ.get("SEED", 3)), perturb=0.1).eval()
g = torch.Generator().manual_seed(int(os.environ.get("GSEED", 12)))
B, Tt = 2, int(os.environ.get("TT", 9))
tok = torch.randint(2, cfg.vocab, (B, Tt), generator=g)
img = torch.randn(B, 3, 56, 56, generator=g)
w = {k: v.clone().requires_grad_() for k, v in oracle_weights(m).items() if v.is_floating_point()}
for k in [k for k in w if k.startswith(("decoder.embed_tokens", "decoder.embed_positions", "decoder.output_projection"))]:
    w.pop(k)
ref = TO.mm_loss(w, tok, img, oracle_cfg(cfg), oracle_switches(m.switches))
TO.backward(ref, w)
tr = KosmosTrainer(m.to("cuda"), precision=sys.argv[1] if len(sys.argv) > 1 else "fp32")
loss = tr.step(tok.cuda(), img.cuda(), apply_update=False)
print(float(loss), float(ref))
for name in dict(m.named_parameters()):
    if ".B." in name or w[name].grad is None:
        continue
    gg = tr.grads[name].cpu(); r = w[name].grad.reshape(gg.shape)
    e = float((gg - r).abs().max() / (r.pow(2).mean().sqrt() + 1e-3))
    if e > float(sys.argv[2] if len(sys.argv) > 2 else 3e-4):
        print(f"{name:70s} {e:.3e}  |g|rms {float(gg.pow(2).mean().sqrt()):.3e} ref {float(r.pow(2).mean().sqrt()):.3e}")
