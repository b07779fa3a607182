import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/kosmos-x_amd']
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx.model import KosmosLanguage
from kosmosx.training import LanguageModelTrainer
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
tb = [torch.randint(2, 32002, (8, 512), generator=g).to(dev) for _ in range(4)]
lm = KosmosLanguage(vocab_size=32002, dim=2048, depth=4, _seed=0).eval().to(dev)
w0 = lm.decoder.layers[0].ffn.fc1.weight.detach().clone() if hasattr(lm.decoder.layers[0].ffn, "fc1") else None
tr = LanguageModelTrainer(lm, precision="bf16")
print("det  ", [round(float(tr.step(t)), 5) for t in tb])
p = dict(lm.named_parameters())
k = [n for n in p if n.endswith("fc1.weight")][0]
print("lm param changed by trainer 1:", bool((p[k].detach() != w0).any()) if w0 is not None else None)
del tr
tr = LanguageModelTrainer(lm, precision="bf16", train_mode=True, dropout_seed=1234)
print("p", tr.p_drop, tr.p_attn)
print("train", [round(float(tr.step(t)), 5) for t in tb])
lm2 = KosmosLanguage(vocab_size=32002, dim=2048, depth=4, _seed=0).eval().to(dev)
tr2 = LanguageModelTrainer(lm2, precision="bf16", train_mode=True, dropout_seed=1234)
print("train fresh", [round(float(tr2.step(t)), 5) for t in tb])
