"""Data-parallel path on CPU (gloo, world_size 2): contiguous batch sharding + the logits all-gather reproduce the
single-process result exactly.  The per-rank compute is the CPU oracle here (no GPU in this container); on the
GPU box the same plumbing wraps Kosmos.forward over RCCL (bench.py)."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

def _np(o):
    """Queue payloads go by VALUE: torch.multiprocessing hands tensors over as shared-memory handles served by the producer,
    and a worker that has left its barrier and exited before the parent opened them made q.get() fail (FileNotFoundError,
    seen once in ~40 runs).  NumPy arrays are pickled whole."""
    if isinstance(o, torch.Tensor):
        return ("__t__", o.detach().cpu().float().numpy() if o.dtype == torch.bfloat16 else o.detach().cpu().numpy(), str(o.dtype))
    if isinstance(o, (list, tuple)):
        return type(o)(_np(x) for x in o)
    if isinstance(o, dict):
        return {k: _np(v) for k, v in o.items()}
    return o


def _pt(o):
    if isinstance(o, tuple) and len(o) == 3 and isinstance(o[0], str) and o[0] == "__t__":
        t = torch.from_numpy(o[1])
        return t.to(torch.bfloat16) if o[2] == "torch.bfloat16" else t
    if isinstance(o, (list, tuple)):
        return type(o)(_pt(x) for x in o)
    if isinstance(o, dict):
        return {k: _pt(v) for k, v in o.items()}
    return o


ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, wire_bf16, q):
    for p in (str(ROOT), str(ROOT / "kosmos-x_amd"), str(ROOT / "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KOSMOSX_NO_LOGGING_CONFIG="1")
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import oracle_cfg, oracle_weights, tiny_config
    from kosmosx.model import Kosmos
    from kosmosx.parallel import LogitsGatherer, shard_range
    from oracle import kosmos_oracle as O
    m = Kosmos._from_config(tiny_config(), seed=0, perturb=0.1).eval()     # same seed => replicated weights
    w, cfg = oracle_weights(m), oracle_cfg(m.cfg)
    g = torch.Generator().manual_seed(7)
    tok = torch.randint(0, m.cfg.vocab, (total, 9), generator=g)
    img = torch.randn(total, 3, 56, 56, generator=g)
    lo, hi = shard_range(total, rank, world)
    local = O.kosmos_forward(w, tok[lo:hi], img[lo:hi], cfg)
    gathered = LogitsGatherer(wire_dtype=torch.bfloat16 if wire_bf16 else None).gather(local)
    if rank == 0:
        full = O.kosmos_forward(w, tok, img, cfg)
        q.put(_np((gathered.float(), full)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wire_bf16", [False, True])
def test_sharded_forward_plus_allgather_equals_full_batch(wire_bf16):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 4, wire_bf16, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, full = _pt(q.get(timeout=240))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert gathered.shape == full.shape
    if wire_bf16:
        assert torch.equal(gathered, full.to(torch.bfloat16).float()) or (gathered - full).abs().max() < 2e-2
    else:
        assert (gathered - full).abs().max() < 2e-5      # row independence: only BLAS blocking differs with batch


def test_shard_range_partitions_the_batch():
    from kosmosx.parallel import shard_range
    for total in (1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


# ---------------------------------------------------------------------------------------------
# Training step across ranks: ZeRO-1-style sharded AdamW == the single-process optimizer on the
# averaged gradient.  The collectives and the slice arithmetic are the product's (kosmosx.parallel); the elementwise
# arithmetic injected here is torch on the CPU (on the GPU it is kx_adamw / kx_reduce_sum).
# ---------------------------------------------------------------------------------------------
def _ref_adamw(lr, betas, eps, wd, max_norm, step):
    def adamw(p, g, m, v, decayed, gsq):
        clip = min(1.0, max_norm / (float(gsq.sqrt()) + 1e-6))
        g = g * clip
        p.mul_(1 - lr * (wd if decayed else 0.0))
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
        p.addcdiv_(m, v.sqrt() / (bc2 ** 0.5) + eps, value=-lr / bc1)
    return adamw


def _zero_worker(rank, world, port, q):
    for p in (str(ROOT), str(ROOT / "kosmos-x_amd"), str(ROOT / "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KOSMOSX_NO_LOGGING_CONFIG="1")
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kosmosx.parallel import ZeroShardedOptimizer
    total, n_decay = 1003, 700                                         # not divisible by the world size
    z = ZeroShardedOptimizer(total, n_decay)
    g0 = torch.Generator().manual_seed(3)
    flat_p = torch.zeros(z.padded); flat_p[:total] = torch.randn(total, generator=g0)      # replicated start
    m, v = torch.zeros(z.shard), torch.zeros(z.shard)
    per_rank = [torch.randn(total, generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
    for step in (1, 2):
        flat_g = torch.zeros(z.padded)
        flat_g[:total] = per_rank[rank] * step / world                 # each rank's gradient already carries 1/world
        z.step(flat_p, flat_g, m, v, _ref_adamw(1e-2, (0.9, 0.95), 1e-8, 0.1, 1.0, step), lambda x: (x * x).sum().reshape(1))
    if rank == 0:
        q.put(_np(flat_p[:total].clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_zero_sharded_adamw_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = _pt(q.get(timeout=240))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process: torch.optim.AdamW on the averaged gradient with clip_grad_norm_
    total, n_decay, world = 1003, 700, 2
    p0 = torch.randn(total, generator=torch.Generator().manual_seed(3))
    pa, pb = torch.nn.Parameter(p0[:n_decay].clone()), torch.nn.Parameter(p0[n_decay:].clone())
    opt = torch.optim.AdamW([{"params": [pa], "weight_decay": 0.1}, {"params": [pb], "weight_decay": 0.0}], lr=1e-2,
                            betas=(0.9, 0.95), eps=1e-8)
    per_rank = [torch.randn(total, generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
    for step in (1, 2):
        gavg = sum(per_rank) * step / world
        pa.grad, pb.grad = gavg[:n_decay].clone(), gavg[n_decay:].clone()
        torch.nn.utils.clip_grad_norm_([pa, pb], 1.0)
        opt.step()
    ref = torch.cat([pa.detach(), pb.detach()])
    assert (got - ref).abs().max() < 1e-6


def test_zero_shard_regions_cover_the_parameters_once():
    from kosmosx.parallel import ZeroShardedOptimizer
    for total, n_decay in ((1003, 700), (16, 16), (10, 0), (4097, 1)):
        for world in (1, 2, 3, 8):
            seen = torch.zeros(total, dtype=torch.int32)
            for rank in range(world):
                z = ZeroShardedOptimizer.__new__(ZeroShardedOptimizer)
                z.group, z.world, z.rank, z.total, z.n_decay = None, world, rank, total, n_decay
                z.shard = ((total + world - 1) // world + 3) // 4 * 4
                z.padded, z.lo, z.hi = z.shard * world, rank * z.shard, (rank + 1) * z.shard
                for a, b, dec in z.regions():
                    seen[a:b] += 1
                    assert (b <= n_decay) if dec else (a >= n_decay)
            assert int(seen.min()) == 1 and int(seen.max()) == 1


# ---------------------------------------------------------------------------------------------
# LogitsGatherer variants: the direct (grouped send/recv, one xGMI link per peer) schedule and ragged shards
# ---------------------------------------------------------------------------------------------
def _gather_worker(rank, world, port, algo, total, q):
    for p in (str(ROOT), str(ROOT / "kosmos-x_amd"), str(ROOT / "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KOSMOSX_NO_LOGGING_CONFIG="1")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kosmosx.parallel import LogitsGatherer, shard_range
    full = torch.arange(total * 3 * 5, dtype=torch.float32).reshape(total, 3, 5)
    lo, hi = shard_range(total, rank, world)
    ga = LogitsGatherer(wire_dtype=None, algo=algo, slots=3)
    outs = [ga.gather(full[lo:hi] + k).clone() for k in range(4)]     # more gathers than slots: buffers are recycled
    ga.wait()
    if rank == 0:
        q.put(_np(outs))
    dist.barrier()
    dist.destroy_process_group()


def _gather_auto_worker(rank, world, port, q):
    for p in (str(ROOT), str(ROOT / "kosmos-x_amd"), str(ROOT / "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KOSMOSX_NO_LOGGING_CONFIG="1")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kosmosx.parallel import LogitsGatherer, shard_range
    ga = LogitsGatherer(wire_dtype=None, slots=2)                     # the default: algo="auto"
    ga.DIRECT_MIN_BYTES = 4096                                        # (instance override: the rule, not 8 MiB of test data)
    res = []
    for total, cols in ((7, 8), (7, 512)):                            # ragged: 3 + 2 + 2 rows; 96-byte rows, then 6 KB rows
        full = torch.arange(total * 3 * cols, dtype=torch.float32).reshape(total, 3, cols)
        lo, hi = shard_range(total, rank, world)
        out = ga.gather(full[lo:hi], total=total)
        ga.wait()
        res.append((ga.last_algo, bool(torch.equal(out, full))))
    if rank == 0:
        q.put(_np(res))
    dist.barrier()
    dist.destroy_process_group()


def _gather_auto_empty_worker(rank, world, port, q):
    for p in (str(ROOT), str(ROOT / "kosmos-x_amd"), str(ROOT / "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KOSMOSX_NO_LOGGING_CONFIG="1")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kosmosx.parallel import LogitsGatherer, shard_range
    ga = LogitsGatherer(wire_dtype=None, slots=2)                     # algo="auto"
    ga.DIRECT_MIN_BYTES = 4096
    total, cols = 2, 2048                                             # 2 rows over 3 ranks: rank 2 holds nothing; 24 KB rows
    full = torch.arange(total * 3 * cols, dtype=torch.float32).reshape(total, 3, cols)
    lo, hi = shard_range(total, rank, world)
    out = ga.gather(full[lo:hi], total=total)
    ga.wait()
    q.put(_np([(rank, ga.last_algo, bool(torch.equal(out, full)))]))
    dist.barrier()
    dist.destroy_process_group()


def test_logits_gatherer_auto_with_an_empty_shard_takes_one_branch_on_every_rank():
    """the auto rule sized the message from the LOCAL tensor, so a rank with a zero-row shard (total < world)
    chose all_gather while its peers chose direct — different collectives, a hang.  Every rank must report the same
    algorithm and the gathered batch must be whole."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_auto_empty_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [_pt(q.get(timeout=240))[0] for _ in range(3)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len({a for _, a, _ in res}) == 1 and all(ok for _, _, ok in res), res
    assert res[0][1] == "all_gather [auto]", res                      # an empty shard rules the grouped send / recv schedule out


def test_logits_gatherer_auto_picks_by_message_size():
    """small messages take RCCL's all_gather, bandwidth-bound ones (the 233 MB logits shard) the
    one-link-per-peer schedule when more than two ranks exchange; the decision uses the largest shard, identical on all ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_auto_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = _pt(q.get(timeout=240))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == ("all_gather [auto]", True) and res[1] == ("direct [auto]", True), res


@pytest.mark.parametrize("algo", ["all_gather", "direct"])
@pytest.mark.parametrize("world,total", [(2, 8), (2, 7), (3, 8)])
def test_logits_gatherer_algorithms_and_ragged_shards(algo, world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, algo, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = _pt(q.get(timeout=240))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    full = torch.arange(total * 3 * 5, dtype=torch.float32).reshape(total, 3, 5)
    for k, o in enumerate(outs):
        assert torch.equal(o, full + k), (algo, world, total, k)


def _gather_tail_worker(rank, world, port, algo, mode, q):
    """ONE gatherer fed a full batch and then a ragged tail batch (totals 8 then 7, then 8 again).  With the
    old per-local-size cache rank 0 (4 rows both times at world 2: 4+4, 4+3) skipped the size exchange that rank 1 entered."""
    for p in (str(ROOT), str(ROOT / "kosmos-x_amd"), str(ROOT / "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KOSMOSX_NO_LOGGING_CONFIG="1")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kosmosx.parallel import LogitsGatherer, shard_range
    ga = LogitsGatherer(wire_dtype=None, algo=algo, slots=2)
    outs = []
    for k, total in enumerate((8, 7, 8, 5)):
        full = torch.arange(total * 2 * 3, dtype=torch.float32).reshape(total, 2, 3) + 100 * k
        lo, hi = shard_range(total, rank, world)
        kw = {"total": total} if mode == "total" else ({"sizes": [b - a for a, b in (shard_range(total, r, world) for r in range(world))]}
                                                       if mode == "sizes" else {})
        outs.append(ga.gather(full[lo:hi], **kw).clone())
    ga.wait()
    bad = None
    try:
        ga.gather(torch.zeros(3, 2, 3), total=8)                       # 3 rows is not this rank's share of 8: loud, no collective
    except ValueError as e:
        bad = str(e)
    if rank == 0:
        q.put(_np((outs, bad)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["all_gather", "direct"])
@pytest.mark.parametrize("mode", ["total", "sizes", "exchange"])
def test_logits_gatherer_ragged_tail_batch_through_one_gatherer(algo, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_tail_worker, args=(r, 2, port, algo, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs, bad = _pt(q.get(timeout=240))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for k, (total, o) in enumerate(zip((8, 7, 8, 5), outs)):
        assert torch.equal(o, torch.arange(total * 2 * 3, dtype=torch.float32).reshape(total, 2, 3) + 100 * k), (algo, mode, k)
    assert bad is not None and "shard_range" in bad


def _subgroup_worker(rank, world, port, q):
    """direct schedule inside a NON-default process group: P2P peers must be global ranks."""
    for p in (str(ROOT), str(ROOT / "kosmos-x_amd"), str(ROOT / "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KOSMOSX_NO_LOGGING_CONFIG="1")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kosmosx.parallel import LogitsGatherer, shard_range
    grp = dist.new_group([1, 2])                                        # group-local ranks 0, 1 = global ranks 1, 2
    out = None
    if rank in (1, 2):
        ga = LogitsGatherer(group=grp, wire_dtype=None, algo="direct", slots=2)
        full = torch.arange(5 * 4, dtype=torch.float32).reshape(5, 4)
        lo, hi = shard_range(5, ga.rank, 2)
        out = ga.gather(full[lo:hi], total=5).clone()
        ga.wait()
    if rank == 1:
        q.put(_np(out))
    dist.barrier()
    dist.destroy_process_group()


def test_logits_gatherer_direct_in_a_subgroup():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    out = _pt(q.get(timeout=240))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert torch.equal(out, torch.arange(20, dtype=torch.float32).reshape(5, 4))


def _zero3_worker(rank, world, port, q):
    for p in (str(ROOT), str(ROOT / "kosmos-x_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KOSMOSX_NO_LOGGING_CONFIG="1")
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kosmosx.parallel import Zero3Layout
    groups = [[("a", 10, True), ("b", 7, False)], [("c", 33, True), ("d", 1, False), ("e", 5, False)], [("f", 4, False)]]
    z = Zero3Layout(groups)
    full = [torch.arange(z.padded[g], dtype=torch.float32) + 1000 * g for g in range(3)]     # the replicated start
    for g in range(3):
        full[g][z.total[g]:] = 0
    shard_p = torch.cat([full[g][rank * z.shard[g]:(rank + 1) * z.shard[g]] for g in range(3)])
    assert shard_p.numel() == z.shard_total
    ok = all(torch.equal(z.gather(g, shard_p), full[g]) for g in range(3))                     # all-gather restores a group
    shard_g = torch.zeros(z.shard_total)
    for g in range(3):
        z.scatter_grad(g, full[g] * (rank + 1), shard_g)                                       # SUM over ranks: x (1 + 2)
    ok = ok and all(torch.equal(shard_g[z.shard_slice(g)], 3 * full[g][rank * z.shard[g]:(rank + 1) * z.shard[g]]) for g in range(3))
    regs = []
    for lo, hi, decayed in z.regions():                    # shard-buffer coordinates -> (group, element range inside the group)
        g = max(i for i in range(3) if z.goff[i] <= lo)
        assert hi <= z.goff[g] + z.shard[g]
        regs.append((g, lo - z.goff[g] + rank * z.shard[g], hi - z.goff[g] + rank * z.shard[g], decayed))
    q.put(_np((rank, ok, regs, z.n_decay, z.total)))
    dist.barrier()
    dist.destroy_process_group()


def test_zero3_layout_gathers_scatters_and_covers_every_parameter_once():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero3_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [_pt(q.get(timeout=240)) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    covered = {}
    for rank, ok, regs, n_decay, total in got:
        assert ok
        for g, lo, hi, decayed in regs:
            for e in range(lo, hi):
                assert (g, e) not in covered and e < total[g] and decayed == (e < n_decay[g])
                covered[(g, e)] = rank
    assert len(covered) == 17 + 39 + 4                                                # every real element, once
