"""Data-parallel plumbing for the forward path: one process per GPU, batch rows sharded contiguously,
replicated weights, and ONE exchange step — the all-gather of logits over RCCL/xGMI
(BASELINE.json north_star).  The forward itself needs no collective: every op of
``Kosmos.forward`` is per-sample (/root/reference/kosmosx/model.py:230-250).

``torch.distributed`` backend "nccl" is RCCL on ROCm; on CPU the same code runs over gloo (tests).
"""
from __future__ import annotations

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
    """All-gather of per-rank logits ``[b, T, V]`` into ``[world*b, T, V]`` (rank-major = global batch order
    when the batch was split with ``shard_range`` into equal shards).

    On HIP devices the collective is issued on a side stream so that the gather of step k overlaps the
    compute of step k+1 (the gathered tensor of step k is safe to read after ``wait()``); payload dtype is
    configurable because xGMI is per-link bound (7 links x ~153 GB/s): bf16 halves the bytes on the wire.
    """

    def __init__(self, group=None, wire_dtype: torch.dtype | None = torch.bfloat16, overlap: bool = True,
                 force: bool = False):
        self.group = group
        self.force = force          # run the collective even with a single rank (exercises the RCCL path in tests)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.wire_dtype = wire_dtype
        self.overlap = overlap
        self._stream = None
        self._pending = []   # (event, send buffer) kept alive until wait()
        self._slot = 0
        self._out = [None, None]

    def gather(self, local: torch.Tensor) -> torch.Tensor:
        if self.world == 1 and not self.force:
            return local
        wire = local if self.wire_dtype is None else local.to(self.wire_dtype)
        wire = wire.contiguous()
        shape = (self.world * wire.shape[0],) + tuple(wire.shape[1:])
        slot = self._slot
        self._slot ^= 1
        out = self._out[slot]
        if out is None or out.shape != shape or out.dtype != wire.dtype or out.device != wire.device:
            out = self._out[slot] = torch.empty(shape, dtype=wire.dtype, device=wire.device)
        if wire.is_cuda and self.overlap:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=wire.device)
            self._stream.wait_stream(torch.cuda.current_stream(wire.device))
            with torch.cuda.stream(self._stream):
                dist.all_gather_into_tensor(out, wire, group=self.group)
                ev = torch.cuda.Event()
                ev.record(self._stream)
            wire.record_stream(self._stream)
            self._pending.append((ev, wire))
        else:
            dist.all_gather_into_tensor(out, wire, group=self.group)
        return out

    def wait(self):
        """Make every gather issued so far visible to the current stream."""
        for ev, _ in self._pending:
            torch.cuda.current_stream().wait_event(ev)
        self._pending.clear()
