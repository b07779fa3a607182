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
        q.put((gathered.float(), full))
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
    gathered, full = q.get(timeout=240)
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
