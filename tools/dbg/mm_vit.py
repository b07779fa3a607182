import sys
sys.path[:0] = [".", "kosmos-x_amd", "tests"]
import torch, torch.nn.functional as F
from kosmosx.model import Kosmos
from kosmosx.training import KosmosTrainer
from kosmosx import ops, grad_ops as G
from oracle import kosmos_oracle as O
from helpers import oracle_cfg, oracle_switches, oracle_weights, tiny_config
cfg = tiny_config()
m = Kosmos._from_config(cfg, seed=3, perturb=0.1).eval()
g = torch.Generator().manual_seed(12)
B = 2
img = torch.randn(B, 3, 56, 56, generator=g)
w = oracle_weights(m); p = "clip_model."
pe_ref = F.conv2d(img, w[p + "embeddings.patch_embedding.weight"], None, stride=14).flatten(2).transpose(1, 2)
h_ref = torch.cat([w[p + "embeddings.class_embedding"].expand(B, 1, -1), pe_ref], 1) + w[p + "embeddings.position_embedding.weight"][None]
x_ref = F.layer_norm(h_ref, (128,), w[p + "pre_layrnorm.weight"], w[p + "pre_layrnorm.bias"], 1e-5)
def rel(a, b): return float((a.cpu().float() - b).abs().max() / b.pow(2).mean().sqrt())
m = m.to("cuda"); tw = m.clip_model; c = tw.cfg
kraw = 588; kpad = 608
patches = G.patchify(img.cuda(), 14, kpad)
pr = F.unfold(img, 14, stride=14).transpose(1, 2).reshape(B * 16, 588)
print("patches", rel(patches[:, :588], pr), float(patches[:, 588:].abs().max()))
wpe = torch.zeros((128, kpad), device="cuda"); wpe[:, :kraw] = tw.embeddings.patch_embedding.weight.detach().flatten(1)
pe = ops.gemm(patches, wpe)
print("pe", rel(pe.view(B, 16, 128), pe_ref))
h0 = G.vit_assemble(pe, tw.embeddings.class_embedding.detach(), tw.embeddings.position_embedding.weight.detach(), B)
print("h0", rel(h0, h_ref))
x = ops.layernorm(h0.reshape(B * 17, 128), tw.pre_layrnorm.weight.detach(), tw.pre_layrnorm.bias.detach(), 1e-5)
print("x", rel(x.view(B, 17, 128), x_ref))
# layer 0 pieces
L = tw.encoder.layers[0]; sa = L.self_attn; q_ = p + "encoder.layers.0."
y_ref = F.layer_norm(x_ref, (128,), w[q_ + "layer_norm1.weight"], w[q_ + "layer_norm1.bias"], 1e-5)
y1 = ops.layernorm(x, L.layer_norm1.weight.detach(), L.layer_norm1.bias.detach(), 1e-5)
print("y1", rel(y1.view(B, 17, 128), y_ref))
wqkv = torch.cat([sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight], 0).detach(); bqkv = torch.cat([sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias], 0).detach()
qkv = ops.gemm(y1, wqkv, bqkv, qscale=0.125, qcols=128)
qr = (y_ref @ w[q_ + "self_attn.q_proj.weight"].t() + w[q_ + "self_attn.q_proj.bias"]) * 0.125
kr = y_ref @ w[q_ + "self_attn.k_proj.weight"].t() + w[q_ + "self_attn.k_proj.bias"]
vr = y_ref @ w[q_ + "self_attn.v_proj.weight"].t() + w[q_ + "self_attn.v_proj.bias"]
print("q", rel(qkv[:, :128].view(B, 17, 128), qr), "k", rel(qkv[:, 128:256].view(B, 17, 128), kr), "v", rel(qkv[:, 256:].view(B, 17, 128), vr))
q3, k3, v3 = (qkv[:, i * 128:(i + 1) * 128].unflatten(0, (B, 17)).unflatten(2, (2, 64)) for i in range(3))
lse = torch.empty((B, 2, 17), device="cuda")
att = ops.attention(q3, k3, v3, False, out_dtype=torch.float32, lse_out=lse)
def sp(t): return t.view(B, 17, 2, 64).transpose(1, 2)
a = torch.softmax(sp(qr) @ sp(kr).transpose(-1, -2), -1)
ar = (a @ sp(vr)).transpose(1, 2).reshape(B, 17, 128)
print("att", rel(att, ar), "lse", rel(lse, torch.logsumexp(sp(qr) @ sp(kr).transpose(-1, -2), -1)))
att2 = ops.attention(q3, k3, v3, False, out_dtype=torch.float32)
print("att (no lse)", rel(att2, ar))
