"""The N > 1 code paths over RCCL itself, with device tensors, on the one GPU a test box has (until
round 3 the `direct` gather, ZeRO stage 3's all-gather / reduce-scatter and stage 1's reduce-scatter + all-gather had only
ever run over gloo with host-staged copies; /root/reference/train.py:709 is the reference's only collective site,
config/zero3.json:26-45 its sharding).  A process group of ONE rank on backend "nccl" (= RCCL on ROCm) with
`force=True`: every collective is entered with device tensors on RCCL's own streams; with one rank each is an identity,
so results are checked bit for bit.  The 2- and 3-rank semantics are covered on CPU by tests/test_distributed.py.

Runs in a spawned process so the pytest process itself never initialises torch.distributed."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(port, q):
    try:
        for p in (str(ROOT), str(ROOT / "kosmos-x_amd"), str(ROOT / "tests")):
            sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KOSMOSX_NO_LOGGING_CONFIG="1",
                          HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        import torch.distributed as dist
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from kosmosx.parallel import LogitsGatherer, Zero3Layout, ZeroShardedOptimizer
        res = {"backend": dist.get_backend()}

        # ---- the logits exchange: both schedules, bf16 wire straight from the tensor, slots recycled, side stream ----
        x = torch.randn(6, 114, 1002, device=dev)
        for algo in ("all_gather", "direct"):
            ga = LogitsGatherer(wire_dtype=None, force=True, algo=algo, slots=2)
            outs = []
            for k in range(3):
                o = ga.gather(x + k, total=6)
                ga.wait()
                outs.append(o.clone())
            torch.cuda.synchronize()
            res[f"gather_{algo}"] = all(torch.equal(o, x + k) for k, o in enumerate(outs)) and ga.last_algo.startswith(algo)
            gb = LogitsGatherer(wire_dtype=torch.bfloat16, force=True, algo=algo)
            o = gb.gather(x)
            gb.wait()
            torch.cuda.synchronize()
            res[f"gather_{algo}_bf16"] = o.dtype == torch.bfloat16 and torch.equal(o, x.to(torch.bfloat16))

        # ---- ZeRO stage 3 layout: all-gather of a group's slices, reduce-scatter of its gradients, the norm all-reduce ----
        groups = [[("a", 10, True), ("b", 7, False)], [("c", 33, True), ("d", 1, False), ("e", 5, False)], [("f", 4, False)]]
        z = Zero3Layout(groups, force=True)
        shard_p = torch.arange(z.shard_total, dtype=torch.float32, device=dev)
        ok = True
        for g in range(3):
            full = z.gather(g, shard_p)
            ok = ok and full.is_cuda and torch.equal(full, shard_p[z.shard_slice(g)])
        shard_g = torch.zeros(z.shard_total, device=dev)
        for g in range(3):
            z.scatter_grad(g, shard_p[z.shard_slice(g)] * 2 + 1, shard_g)
        ok = ok and torch.equal(shard_g, shard_p * 2 + 1)
        t = torch.tensor([3.5], device=dev)
        ok = ok and float(z.all_reduce_scalar(t)) == 3.5
        res["zero3_layout"] = bool(ok)

        # ---- ZeRO stage 1: reduce-scatter(gradients) + all-reduce(norm) + all-gather(parameters) ----
        zo = ZeroShardedOptimizer(total=1000, n_decay=600)
        p = torch.randn(zo.padded, device=dev)
        p[1000:] = 0
        g_ = torch.randn(zo.padded, device=dev)
        g_[1000:] = 0
        p0 = p.clone()
        mom, var = torch.zeros(zo.shard, device=dev), torch.zeros(zo.shard, device=dev)
        seen = []

        def adamw(pp, gg, mm, vv, decayed, gsq):
            seen.append((pp.numel(), decayed))
            pp.sub_(0.5 * gg)

        gsq = zo.step(p, g_, mom, var, adamw, lambda t_: t_.pow(2).sum().reshape(1), force=True)
        torch.cuda.synchronize()
        res["zero1_step"] = (torch.equal(p[:1000], (p0 - 0.5 * g_)[:1000]) and seen == [(600, True), (400, False)]
                             and abs(float(gsq) - float(g_.pow(2).sum())) < 1e-3 * float(gsq))

        # ---- the whole trainer with its collectives forced: bitwise the un-forced step (stage 1 and stage 3) ----
        from kosmosx.model import KosmosLanguage
        from kosmosx.training import LanguageModelTrainer
        tok = torch.randint(2, 302, (2, 24), generator=torch.Generator().manual_seed(3)).to(dev)
        for stage in (1, 3):
            finals = []
            for force in (False, True):
                lm = KosmosLanguage(vocab_size=302, dim=128, depth=2, ffn_dim=256, decoder_heads=2, _seed=4, _perturb=0.1,
                                    _max_positions=64).to(dev)
                tr = LanguageModelTrainer(lm, precision="fp32", zero_stage=stage, force_collectives=force)
                losses = [float(tr.step(tok)) for _ in range(2)]
                if stage == 3:
                    tr.gather_parameters()
                finals.append((losses, {n: v.detach().clone() for n, v in lm.named_parameters()}))
            (l0, p0_), (l1, p1_) = finals
            res[f"trainer_stage{stage}"] = l0 == l1 and all(torch.equal(p0_[n], p1_[n]) for n in p0_)
        dist.barrier()
        dist.destroy_process_group()
        q.put(res)
    except Exception as e:      # surface the worker's failure in the parent's assertion
        import traceback
        q.put({"error": f"{e!r}\n{traceback.format_exc()}"})


def test_collectives_run_over_rccl_with_device_tensors_on_one_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(120)
    assert "error" not in res, res.get("error")
    assert res.pop("backend") == "nccl"
    bad = [k for k, v in res.items() if v is not True]
    assert not bad, (bad, res)
    assert set(res) == {"gather_all_gather", "gather_all_gather_bf16", "gather_direct", "gather_direct_bf16", "zero3_layout",
                        "zero1_step", "trainer_stage1", "trainer_stage3"}
