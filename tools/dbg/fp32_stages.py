"""Where does the HIP fp32 path sit relative to the float64 evaluation of the oracle, stage by stage (full size, B=1)?"""
import sys
sys.path[:0] = [".", "kosmos-x_amd", "tests"]
import torch
from kosmosx.model import Kosmos
from oracle import kosmos_oracle as O
from helpers import oracle_cfg, oracle_weights
m = Kosmos._from_config(__import__("kosmosx.config", fromlist=["KosmosConfig"]).KosmosConfig(), seed=0, perturb=0.05).eval()
cfg = oracle_cfg(m.cfg)
g = torch.Generator().manual_seed(0)
tok = torch.randint(0, m.cfg.vocab, (1, 50), generator=g)
img = torch.randn(1, 3, 224, 224, generator=g)
w = oracle_weights(m)
s32, s64 = {}, {}
r32 = O.kosmos_forward(w, tok, img, cfg, O.Switches(), stages=s32)
with O.working_dtype(torch.float64):
    r64 = O.kosmos_forward({k: (v.double() if v.is_floating_point() else v) for k, v in w.items()}, tok, img.double(), cfg, O.Switches(), stages=s64)
s32["logits"], s64["logits"] = r32, r64
m = m.to("cuda"); m.precision = "fp32"
with torch.no_grad():
    v = m.clip_model.run(img.cuda(), "fp32", m._ws)
    im, lat = m.perceive.run(v, "fp32", m._ws, m.image_proj.weight, want_latents=True)
    x = m.decoder.embed(tok.cuda(), "fp32", img=im)
    lg = m.decoder.run(x.clone(), "fp32")
hip = {"vit": v, "perceiver": lat, "image_proj": im, "embed": x, "logits": lg}
def rel(a, b): return float((a.double().cpu().reshape(b.shape) - b).abs().max() / b.pow(2).mean().sqrt())
for k in ("vit", "perceiver", "image_proj", "embed", "logits"):
    print(f"{k:11s} HIP fp32 vs f64 {rel(hip[k], s64[k]):.3e}   CPU fp32 vs f64 {rel(s32[k], s64[k]):.3e}")
# decoder alone from the float64 embed: isolates the decoder's own noise
with torch.no_grad():
    lg2 = m.decoder.run(s64["embed"].float().cuda().contiguous(), "fp32")
print("decoder from exact input: HIP", rel(lg2, s64["logits"]))
# the op-by-op training forward (unfolded LayerNorms, separate GELU) on the same exact input
from kosmosx.training import KosmosTrainer
tr = KosmosTrainer(m)
o = tr._make_ops()
lg3, _ = tr._decoder_forward(o, s64["embed"].float().cuda().reshape(114, 2048).contiguous(), 1, 114)
print("op-by-op decoder (unfolded LN) from exact input: HIP", rel(lg3[:, :32002], s64["logits"]))
