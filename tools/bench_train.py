"""Training step of the text decoder (BASELINE configs[4]): tokens/s of forward + backward + clip +
AdamW on synthetic token batches, full-size 24L/2048d decoder, with the kernel-class breakdown and a bounded CPU sample
of the oracle's step.  One GPU:  python tools/bench_train.py --precision bf16
N GPUs of a node (one process per GPU, RCCL; every rank its own batch shard, ZeRO-1-style sharded AdamW):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_train.py --precision bf16
tokens/s is the whole-job figure (all ranks' tokens / the slowest rank's time), weak scaling (fixed batch per GPU)."""
import argparse, json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd"), str(ROOT / "tests")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import _hip
from kosmosx.model import KosmosLanguage
from kosmosx.training import LanguageModelTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--seq", type=int, default=512)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--layers", type=int, default=24)
ap.add_argument("--cpu-seconds", type=float, default=20.0)
ap.add_argument("--checkpoint", action="store_true", help="keep only layer inputs, recompute each layer before its backward")
ap.add_argument("--train-mode", action="store_true", help="the reference's model.train(): dropout = attention_dropout = 0.1 "
                "(/root/reference/train.py:642, kosmosx/model.py:175-177), Philox masks; default: the deterministic step")
ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16"], help="arithmetic of the matrix products")
ap.add_argument("--tune", default="", help="A/B: kx_set_tuning key=value pairs, e.g. 13=1 (no pair split of the 256x256 GEMM kernel)")
a = ap.parse_args()
for kv in filter(None, a.tune.split(",")):
    from kosmosx import _hip
    _hip.load().kx_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
import torch.distributed as dist
world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
force = os.environ.get("KOSMOSX_FORCE_DIST") == "1"          # single-rank run of the RCCL path
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
if world > 1 or force:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
lm = KosmosLanguage(vocab_size=32002, dim=2048, depth=a.layers, _seed=0).eval().to(dev)      # same seed: replicated weights
tr = LanguageModelTrainer(lm, precision=a.precision, force_collectives=force, checkpoint_activations=a.checkpoint,
                          train_mode=a.train_mode, dropout_seed=1234)
g = torch.Generator().manual_seed(1000 + rank)                                               # per-rank batch shard
batches = [torch.randint(2, 32002, (a.batch, a.seq), generator=g).to(dev) for _ in range(a.warmup + a.steps + 1)]
losses = []
for i in range(a.warmup):
    losses.append(float(tr.step(batches[i])))
torch.cuda.synchronize()
if world > 1 or force:
    dist.barrier()
t0 = time.perf_counter()
for i in range(a.warmup, a.warmup + a.steps):
    loss = tr.step(batches[i])
torch.cuda.synchronize()
if world > 1 or force:
    dist.barrier()
dt = (time.perf_counter() - t0) / a.steps
if world > 1:
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
losses.append(float(loss))
recs = []
if world == 1 and not force:                 # the instrumented step (per-launch events) runs on a single-GPU job only
    _hip.prof_enable(True)
    tr.step(batches[-1])
    torch.cuda.synchronize()
    recs = _hip.prof_collect()
    _hip.prof_enable(False)
MISC = {20: "transpose", 21: "colsum", 22: "gelu_bwd", 23: "cross_entropy", 24: "reduce_sum", 25: "xpos_bwd", 26: "adamw",
        27: "gelu_fwd", 28: "to_operand", 3: "stats_finalize", 0: "rows_bcast"}
agg = {}
for kind, x, y, z, ms in recs:
    if kind == "misc":
        kind = "misc:" + MISC.get(int(z), str(z))
    elif kind == "attn_f32":
        kind = "attn_f32_bwd" if z < 0 else "attn_f32_fwd"
    elif kind == "layernorm":
        kind = "layernorm_bwd" if z == 1 else "layernorm_fwd"
    elif kind == "embed":
        kind = "embed_bwd" if z == 1 else "embed_fwd"
    e = agg.setdefault(kind, [0, 0.0, 0.0]); e[0] += 1; e[1] += ms
    if "gemm" in str(kind):
        e[2] += 2.0 * x * y * z
nparams = sum(p.numel() for p in lm.parameters())
tokens = a.batch * a.seq * world
flops = 6.0 * (nparams - 32002 * 2048 - lm.embed_positions.weight.numel()) * tokens   # matmul parameters x 6 (fwd + 2x bwd)
res = {"workload": f"KosmosLanguage train step (fwd+bwd+clip+AdamW), {a.layers}L/2048d, B={a.batch} T={a.seq}, {a.precision} products on fp32 master weights"
                   + (f", train mode (dropout {tr.p_drop}, attention dropout {tr.p_attn})" if a.train_mode else ""),
       "n_gpus": world, "scaling": "weak", "ms_per_step": round(dt * 1e3, 1), "tokens_per_s": round(tokens / dt, 1), "losses": [round(l, 4) for l in losses],
       "approx_model_tflops": round(flops / dt / 1e12, 1),
       "kernels_ms": {k: {"n": v[0], "ms": round(v[1], 1), **({"tflops": round(v[2] / v[1] / 1e9, 1)} if v[2] else {})}
                      for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}
# bounded CPU sample: the oracle's step (autograd + torch.optim.AdamW) on one short sequence of the same model
if a.cpu_seconds > 0 and rank == 0 and world == 1:
    from oracle import kosmos_oracle as O
    from oracle import train_oracle as TO
    from helpers import oracle_weights
    cfg = O.DecoderCfg(layers=a.layers, vocab=32002)
    w = {k: v.clone().requires_grad_() for k, v in oracle_weights(lm).items()
         if not (k.startswith("decoder.embed_") or k.startswith("decoder.output_projection"))}
    opt = TO.make_optimizer(w)
    ctok = batches[0][:1, :128].cpu()
    n, t_cpu = 0, 0.0
    while t_cpu < a.cpu_seconds and n < 8:
        t1 = time.perf_counter(); TO.train_step(w, opt, ctok, cfg); t_cpu += time.perf_counter() - t1; n += 1
    res["cpu_baseline"] = {"value": round(n * 128 / t_cpu, 1), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                           "sample": f"{n} x train step on 1x128 tokens, fp32 torch autograd + AdamW on the oracle, {t_cpu:.1f} s"}
if rank == 0:
    print(json.dumps(res))
if world > 1 or force:
    dist.barrier()
    dist.destroy_process_group()
