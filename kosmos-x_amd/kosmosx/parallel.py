"""Data-parallel plumbing for the forward path: one process per GPU, batch rows sharded contiguously,
replicated weights, and ONE exchange step — the all-gather of logits over RCCL/xGMI
(BASELINE.json north_star).  The forward itself needs no collective: every op of
``Kosmos.forward`` is per-sample (/root/reference/kosmosx/model.py:230-250).

``torch.distributed`` backend "nccl" is RCCL on ROCm; on CPU the same code runs over gloo (tests).
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous split of ``total`` batch rows: rank r takes [lo, hi).  Remainder rows go to the first ranks."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class LogitsGatherer:
    """All-gather of per-rank logits ``[b_r, T, V]`` into ``[sum(b_r), T, V]`` (rank-major = global batch order when the
    batch was split with ``shard_range``).

    * On HIP devices the exchange is issued on a side stream so that the gather of step k overlaps the compute of step
      k+1; the gathered tensor of step k is safe to read after ``wait()`` and stays valid until ``slots - 1`` further
      gathers have been issued (``slots`` output buffers are cycled; use pipeline depth + 1).
    * Payload: xGMI is per-link bound (7 links x ~153 GB/s), so bytes matter.  ``wire_dtype=None`` sends ``local`` as it
      is — the intended use is a model that already emits bf16 logits straight from the logits GEMM's epilogue
      (``model.logits_dtype = torch.bfloat16``), so no cast kernel runs; a dtype here casts first.
    * ``algo="all_gather"``: one ``all_gather_into_tensor`` (RCCL picks ring / tree).  ``algo="direct"``: world - 1
      grouped send / recv pairs, every peer over its own xGMI link at once — the fully-connected schedule SURVEY 8e
      computes at ~1/7 of a ring's time for this message (233 MB per rank in bf16 at 32 samples per GPU).
      ``algo="auto"`` (default) decides per call from the message: ``direct`` when more than two ranks exchange at least
      ``DIRECT_MIN_BYTES`` per rank (bandwidth-bound: a ring moves world - 1 shards over ONE link per rank, the direct
      schedule one shard over each of world - 1 links), RCCL's ``all_gather`` below that (latency-bound: RCCL's tuned
      small-message protocols) and for two ranks (one link either way).  The decision uses the LARGEST shard of the size
      list, which every rank holds identically, so all ranks take the same branch.
    * Ragged shards (global batch not divisible by world): rows are padded to the largest shard for ``all_gather`` and
      the padding is dropped; ``direct`` sends exact sizes.  The per-rank row counts come from the caller whenever it
      knows them — ``gather(local, total=global_batch)`` (the ``shard_range`` split: no collective, no host sync) or
      ``gather(local, sizes=[...])`` — and are otherwise exchanged on EVERY call (a 1-element all_gather): a cache keyed
      on this rank's own row count let ranks disagree about whether to enter that collective on a ragged tail batch
      (world 2, batches of 10 then 9 — rank 0 keeps 5 rows and skipped it, rank 1 entered it alone).
    """

    def __init__(self, group=None, wire_dtype: torch.dtype | None = torch.bfloat16, overlap: bool = True,
                 force: bool = False, algo: str = "auto", slots: int = 3):
        if algo not in ("auto", "all_gather", "direct"):
            raise ValueError("algo must be 'auto', 'all_gather' or 'direct'")
        self.group = group
        self.force = force          # run the collective even with a single rank (exercises the RCCL path in tests)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.wire_dtype = wire_dtype
        self.overlap = overlap
        self.algo = algo
        self._stream = None
        self._pending = []   # (event, send buffer) kept alive until wait()
        self._slot = 0
        self._out = [None] * max(2, int(slots))
        self.last_algo = None                                   # what the last gather() actually ran (bench line)

    DIRECT_MIN_BYTES = 8 << 20

    def _pick(self, wire, sizes) -> str:
        if self.algo != "auto":
            return self.algo
        # bytes per row from the TRAILING dims, which every rank shares: a rank with a zero-row shard must not compute 0
        # here and enter a different collective than its peers (shard_range leaves empty shards when total < world)
        row_bytes = wire.element_size() * math.prod(wire.shape[1:])
        if min(sizes) == 0:                 # `direct` would post sends / receives of empty slices: take the library collective
            return "all_gather"
        return "direct" if self.world > 2 and max(sizes) * row_bytes >= self.DIRECT_MIN_BYTES else "all_gather"

    def _shard_sizes(self, rows: int, device, total=None, sizes=None) -> list:
        """Rows of every rank.  Every rank must take the same branch: `total` / `sizes` are collective-free, the
        fallback is a collective that all ranks enter on every call."""
        if sizes is not None:
            sizes = [int(n) for n in sizes]
            if len(sizes) != self.world or sizes[self.rank] != rows:
                raise ValueError(f"sizes {sizes} do not describe rank {self.rank}'s {rows} rows in a world of {self.world}")
            return sizes
        if total is not None:
            sizes = [hi - lo for lo, hi in (shard_range(int(total), r, self.world) for r in range(self.world))]
            if sizes[self.rank] != rows:
                raise ValueError(f"rank {self.rank} holds {rows} rows; shard_range({total}, {self.rank}, {self.world}) "
                                 f"gives {sizes[self.rank]}")
            return sizes
        t = torch.tensor([rows], dtype=torch.int64, device=device)
        allr = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(allr, t, group=self.group)
        return [int(x.item()) for x in allr]

    def _peer(self, r: int) -> int:
        """P2POp peers are GLOBAL ranks; self.rank / the staggered schedule are group-local."""
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def _exchange(self, out, wire, sizes, algo):
        if algo == "direct" and self.world == 1:           # force=True on one rank: the P2P machinery against itself
            ops = [dist.P2POp(dist.isend, wire, self._peer(0), self.group),
                   dist.P2POp(dist.irecv, out, self._peer(0), self.group)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            return out
        if algo == "direct" and self.world > 1:
            offs = [0]
            for n in sizes:
                offs.append(offs[-1] + n)
            out[offs[self.rank]:offs[self.rank + 1]].copy_(wire)
            ops = []
            for d in range(1, self.world):                      # peer order staggered by rank: every link busy at once
                dst, src = (self.rank + d) % self.world, (self.rank - d) % self.world
                ops.append(dist.P2POp(dist.isend, wire, self._peer(dst), self.group))
                ops.append(dist.P2POp(dist.irecv, out[offs[src]:offs[src + 1]], self._peer(src), self.group))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            return out
        if len(set(sizes)) == 1:
            dist.all_gather_into_tensor(out, wire, group=self.group)
            return out
        mx = max(sizes)                                         # ragged: pad to the largest shard, drop the padding
        padded = wire if wire.shape[0] == mx else torch.cat([wire, wire.new_zeros((mx - wire.shape[0],) + tuple(wire.shape[1:]))])
        full = wire.new_empty((self.world * mx,) + tuple(wire.shape[1:]))
        dist.all_gather_into_tensor(full, padded.contiguous(), group=self.group)
        lo = 0
        for r, n in enumerate(sizes):
            out[lo:lo + n].copy_(full[r * mx:r * mx + n])
            lo += n
        return out

    def gather(self, local: torch.Tensor, total: int | None = None, sizes=None) -> torch.Tensor:
        """total: the global batch that was split with shard_range (preferred); sizes: explicit rows per rank."""
        if self.world == 1 and not self.force:
            self.last_algo = "none (one rank)"
            return local
        wire = local if (self.wire_dtype is None or local.dtype == self.wire_dtype) else local.to(self.wire_dtype)
        wire = wire.contiguous()
        sizes = self._shard_sizes(wire.shape[0], wire.device, total, sizes) if self.world > 1 else [wire.shape[0]]
        algo = self._pick(wire, sizes)
        self.last_algo = (algo if self.world > 1 else f"{algo} (forced on one rank)") + (" [auto]" if self.algo == "auto" else "")
        shape = (sum(sizes),) + tuple(wire.shape[1:])
        slot = self._slot
        self._slot = (self._slot + 1) % len(self._out)
        out = self._out[slot]
        if out is None or out.shape != shape or out.dtype != wire.dtype or out.device != wire.device:
            out = self._out[slot] = torch.empty(shape, dtype=wire.dtype, device=wire.device)
        if wire.is_cuda and self.overlap:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=wire.device)
            self._stream.wait_stream(torch.cuda.current_stream(wire.device))
            with torch.cuda.stream(self._stream):
                self._exchange(out, wire, sizes, algo)
                ev = torch.cuda.Event()
                ev.record(self._stream)
            wire.record_stream(self._stream)
            self._pending.append((ev, wire))
        else:
            self._exchange(out, wire, sizes, algo)
        return out

    def wait(self):
        """Make every gather issued so far visible to the current stream of the device that issued it."""
        for ev, wire in self._pending:
            torch.cuda.current_stream(wire.device).wait_event(ev)
        self._pending.clear()


class ZeroShardedOptimizer:
    """ZeRO-stage-1-style data parallelism for the training step (BASELINE configs[4]: the
    reference's config/zero3.json shards parameters, gradients and optimizer state; this first step shards the
    optimizer state and the update, the two thirds of the memory that matter for AdamW): every rank holds the whole
    flat fp32 parameter and gradient buffers and 1/world of the Adam moments.  One step is

        reduce-scatter(gradients, SUM)   — each rank receives the sum of its slice (the loss scale carries 1/world)
        all-reduce(|slice|^2)            — the global gradient norm for clip_grad_norm_
        AdamW on the slice               — decay region / no-decay region of the flat layout
        all-gather(parameters)           — every rank ends the step with identical parameters

    i.e. exactly two bandwidth collectives of the parameter size per step, both of the ring-friendly kind xGMI
    wants, and no collective inside forward or backward.  The arithmetic is injected (`adamw`, `sumsq`): HIP kernels
    in the product (kosmosx/training.py), torch reference ops in the gloo CPU test.

    Layout contract: flat buffers of `padded` = world * shard floats; [0, n_decay) are the weight-decayed
    parameters, [n_decay, total) the others, [total, padded) zero padding.
    """

    def __init__(self, total: int, n_decay: int, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.total, self.n_decay = total, n_decay
        self.shard = (total + self.world - 1) // self.world
        self.shard = (self.shard + 3) // 4 * 4                    # keep 16-byte aligned slices
        self.padded = self.shard * self.world
        self.lo, self.hi = self.rank * self.shard, (self.rank + 1) * self.shard

    def regions(self):
        """[(start, stop, decayed?)] of this rank's slice, in flat coordinates (padding excluded)."""
        out = []
        a, b = self.lo, min(self.hi, self.n_decay)
        if b > a:
            out.append((a, b, True))
        a, b = max(self.lo, self.n_decay), min(self.hi, self.total)
        if b > a:
            out.append((a, b, False))
        return out

    def step(self, flat_p, flat_g, m, v, adamw, sumsq, force: bool = False):
        """flat_p / flat_g: [padded]; m / v: [shard] (this rank's moments).  adamw(p, g, m, v, decayed, gnorm_sq) updates
        in place; sumsq(x) -> 1-element tensor.  Returns the global squared gradient norm (1-element tensor)."""
        distributed = self.world > 1 or (force and dist.is_initialized())
        # gloo with device tensors (several ranks sharing one GPU: the whole-trainer test on a 1-GPU box): the collectives
        # run on host copies.  RCCL (the product path) takes the device tensors as they are.
        staged = distributed and flat_g.is_cuda and dist.get_backend(self.group) == "gloo"
        if distributed:
            src = flat_g.cpu() if staged else flat_g
            g_shard = torch.empty(self.shard, dtype=flat_g.dtype, device=src.device)
            dist.reduce_scatter_tensor(g_shard, src, op=dist.ReduceOp.SUM, group=self.group)
            g_shard = g_shard.to(flat_g.device)
        else:
            g_shard = flat_g[self.lo:self.hi]
        gsq = sumsq(g_shard)
        if distributed:
            if staged:
                h = gsq.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                gsq.copy_(h)
            else:
                dist.all_reduce(gsq, op=dist.ReduceOp.SUM, group=self.group)
        for a, b, decayed in self.regions():
            s = slice(a - self.lo, b - self.lo)
            adamw(flat_p[a:b], g_shard[s], m[s], v[s], decayed, gsq)
        if distributed:
            if staged:
                full = torch.empty(self.padded, dtype=flat_p.dtype)
                dist.all_gather_into_tensor(full, flat_p[self.lo:self.hi].cpu(), group=self.group)
                flat_p.copy_(full)
            else:
                dist.all_gather_into_tensor(flat_p, flat_p[self.lo:self.hi].clone(), group=self.group)
        return gsq


class Zero3Layout:
    """ZeRO-stage-3 layout of the training step (BASELINE configs[4]; /root/reference/config/zero3.json:26-45 shards
    parameters, gradients and optimizer state): the parameters are cut into GROUPS (one per decoder layer + one for the
    rest); every group's flat fp32 buffer [decayed | others | padding] is cut into `world` equal slices and a rank keeps
    slice `rank` of EVERY group — its master parameters, gradients and both Adam moments are 1/world of the model.  A group
    exists in full only while it is used:

        acquire(g)   all-gather the group's slices into a scratch buffer (the layer's weights for forward / recompute)
        release(g)   reduce-scatter the group's full gradient buffer (SUM; the loss scale carries 1/world) into the
                     rank's gradient slice and drop both scratch buffers

    so a step moves each parameter three times (gather for forward, gather for the recompute before backward, reduce-
    scatter of its gradient) — DeepSpeed stage 3's traffic — and the optimizer needs no collective but the norm.
    The collectives are injected like ZeroShardedOptimizer's (device tensors on RCCL, host copies on gloo)."""

    def __init__(self, groups, group=None, force: bool = False):
        """groups: list of lists of (name, numel, decayed) in buffer order (decayed entries first within a group).
        force: run the collectives even with a single rank (exercises the RCCL path on a one-GPU box)."""
        self.group = group
        self.force = bool(force) and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.offset, self.gof = {}, {}            # name -> offset inside its group's full buffer / group index
        self.total, self.n_decay, self.shard, self.padded, self.goff = [], [], [], [], []
        off_shard = 0
        for gi, items in enumerate(groups):
            off, nd, seen_other = 0, 0, False
            for name, numel, decayed in items:
                if decayed:
                    assert not seen_other, "decayed parameters come first inside a group"
                    nd += numel
                else:
                    seen_other = True
                self.offset[name], self.gof[name] = off, gi
                off += numel
            sh = (off + self.world - 1) // self.world
            sh = (sh + 3) // 4 * 4                       # 16-byte aligned slices
            self.total.append(off); self.n_decay.append(nd); self.shard.append(sh); self.padded.append(sh * self.world)
            self.goff.append(off_shard)
            off_shard += sh
        self.shard_total = off_shard

    def shard_slice(self, gi):
        """This rank's slice of group gi inside its shard buffers."""
        return slice(self.goff[gi], self.goff[gi] + self.shard[gi])

    def regions(self):
        """[(start, stop, decayed?)] of the rank's shard buffer (padding excluded)."""
        out = []
        for gi in range(len(self.total)):
            lo, hi = self.rank * self.shard[gi], (self.rank + 1) * self.shard[gi]          # group coordinates
            base = self.goff[gi] - lo
            a, b = lo, min(hi, self.n_decay[gi])
            if b > a:
                out.append((base + a, base + b, True))
            a, b = max(lo, self.n_decay[gi]), min(hi, self.total[gi])
            if b > a:
                out.append((base + a, base + b, False))
        return out

    def _staged(self, t):
        return (self.world > 1 or self.force) and t.is_cuda and dist.get_backend(self.group) == "gloo"

    def gather(self, gi, shard_p):
        """-> the group's full parameter buffer [padded] from every rank's slice."""
        mine = shard_p[self.shard_slice(gi)]
        if self.world == 1 and not self.force:
            return mine.clone()
        if self._staged(mine):
            full = torch.empty(self.padded[gi], dtype=mine.dtype)
            dist.all_gather_into_tensor(full, mine.cpu(), group=self.group)
            return full.to(mine.device)
        full = torch.empty(self.padded[gi], dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(full, mine.contiguous(), group=self.group)
        return full

    def scatter_grad(self, gi, full_g, shard_g):
        """reduce-scatter(SUM) of the group's full gradient buffer into this rank's gradient slice."""
        dst = shard_g[self.shard_slice(gi)]
        if self.world == 1 and not self.force:
            dst.copy_(full_g)
        elif self._staged(full_g):
            out = torch.empty(self.shard[gi], dtype=full_g.dtype)
            dist.reduce_scatter_tensor(out, full_g.cpu(), op=dist.ReduceOp.SUM, group=self.group)
            dst.copy_(out)
        else:
            dist.reduce_scatter_tensor(dst, full_g, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce_scalar(self, t):
        if self.world == 1 and not self.force:
            return t
        if self._staged(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t
