"""Soak run of the training step at full size: N steps over a few fixed synthetic batches (the loss must fall and stay
finite, memory must not grow), in the chosen precision.  Prints one JSON line."""
import argparse, json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx.model import KosmosLanguage
from kosmosx.training import LanguageModelTrainer, cosine_schedule_with_warmup

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=120)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--lr", type=float, default=3e-4)
a = ap.parse_args()
dev = torch.device("cuda", 0)
lm = KosmosLanguage(vocab_size=32002, dim=2048, _seed=0).eval().to(dev)
tr = LanguageModelTrainer(lm, lr=a.lr, precision=a.precision)
g = torch.Generator().manual_seed(0)
batches = [torch.randint(2, 32002, (8, 256), generator=g).to(dev) for _ in range(4)]
losses, mem = [], []
t0 = time.perf_counter()
for step in range(a.steps):
    tr.lr = a.lr * cosine_schedule_with_warmup(step, max(1, a.steps // 100), a.steps)
    losses.append(float(tr.step(batches[step % 4])))
    if step % 20 == 0:
        mem.append(round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
torch.cuda.synchronize()
finite = all(torch.isfinite(p).all().item() for p in lm.parameters())
print(json.dumps({"precision": a.precision, "steps": a.steps, "seconds": round(time.perf_counter() - t0, 1),
                  "loss_first": round(losses[0], 4), "loss_every_20": [round(l, 3) for l in losses[::20]],
                  "loss_last": round(losses[-1], 4), "all_parameters_finite": finite, "peak_GiB_every_20": mem}))
