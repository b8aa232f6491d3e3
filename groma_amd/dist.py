"""Multi-GPU data parallelism of the forward path: every image is independent
(the reference evaluates the same way: DistributedSampler at groma/eval/eval_rec.py:80-83), so the image batch is
sharded across one process per GPU with a full model replica each and NO data-path collective inside the forward;
the only exchange is one small all-gather of the per-image results per batch (region logits = last-position logits
over <r0..r99>, generated ids, boxes), replacing the reference's three scalar reduces (eval_rec.py:122-124).
Backend: torch.distributed ("nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests)."""
import os

import torch


def shard_range(global_batch, world, rank):
    """Contiguous image shard [lo, hi) of rank `rank`; the remainder goes to the first ranks."""
    q, r = divmod(global_batch, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def init(backend=None, device=None, single_process=False):
    """Join the process group the launcher described (RANK / WORLD_SIZE / MASTER_* in the environment, as torchrun and
    bench.py's own launcher export them).  A missing RANK or WORLD_SIZE raises: N processes that each silently became
    "rank 0 of 1" would report single-GPU numbers as if scaling had worked.  single_process=True is the explicit request for a
    one-rank group (smoke test of the collective path on one GPU)."""
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    if single_process:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    missing = [k for k in ("RANK", "WORLD_SIZE") if k not in os.environ]
    if missing:
        raise RuntimeError(f"groma_amd.dist.init: {', '.join(missing)} not set -- launch one process per GPU with "
                           f"torch.distributed.run (or `python bench.py --gpus N`, which does), or pass single_process=True")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, **kw)
    return dist


def count_ranks(device):
    """all-reduce of ones over the default group: how many ranks actually take part (1 without a process group)"""
    import torch.distributed as dist
    if not dist.is_initialized():
        return 1
    t = torch.ones((1,), dtype=torch.float32, device=device)
    dist.all_reduce(t)
    return int(round(float(t.item())))


def all_gather_rows(local, counts=None):
    """All-gather a [b_local, ...] tensor along dim 0 (ragged shard sizes allowed via `counts` = per-rank rows).
    One collective; KB-scale messages -> latency-bound on xGMI, not link-bound."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if counts is None or len(set(counts)) == 1:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx: r * mx + c] for r, c in enumerate(counts)])


def region_logits(logits, box_idx_token_ids):
    """last-position logits restricted to the <r_i> vocabulary slice (SURVEY §8 'Region logits')."""
    r0 = box_idx_token_ids[0]
    return logits[:, -1, r0:r0 + len(box_idx_token_ids)].float().contiguous()


class ShardedJob:
    """The per-rank driver of an image-sharded job -- what bench.py runs, and what tests/test_dist_gloo.py runs under gloo.

    One process per GPU, a full replica each; `rows` images of the global batch live on this rank (weak scaling: the same
    count everywhere; strong scaling: shard_range of a fixed global batch, possibly ragged).  A step = local forward of the
    shard + ONE all-gather of the per-image result rows into a preallocated buffer (no allocation, no host sync inside the
    timed region).  Timing follows the driver's contract: barrier + device sync on both sides, MAX over ranks."""

    def __init__(self, device, row_shape, dtype, global_batch=None, rows_per_rank=None):
        import torch.distributed as dist
        self.dist = dist if dist.is_initialized() else None
        self.world = dist.get_world_size() if self.dist else 1
        self.rank = dist.get_rank() if self.dist else 0
        self.device = torch.device(device)
        if global_batch is not None:   # strong scaling: a fixed global batch split over the ranks
            self.counts = [shard_range(global_batch, self.world, r)[1] - shard_range(global_batch, self.world, r)[0]
                           for r in range(self.world)]
            self.lo, self.hi = shard_range(global_batch, self.world, self.rank)
        else:                          # weak scaling: rows_per_rank images on every rank
            self.counts = [int(rows_per_rank)] * self.world
            self.lo, self.hi = self.rank * rows_per_rank, (self.rank + 1) * rows_per_rank
        self.rows = self.counts[self.rank]
        self.global_batch = sum(self.counts)
        self.max_rows = max(self.counts)
        self.row_shape = tuple(row_shape)
        self._send = torch.zeros((self.max_rows,) + self.row_shape, dtype=dtype, device=self.device)
        self._recv = torch.zeros((self.world * self.max_rows,) + self.row_shape, dtype=dtype, device=self.device)

    def exchange(self, local_rows):
        """local_rows [rows, *row_shape] -> view [global_batch, *row_shape] of every rank's rows, in image order."""
        if self.dist is None:
            return local_rows
        if local_rows.shape[0] == self.max_rows and len(set(self.counts)) == 1:
            self.dist.all_gather_into_tensor(self._recv, local_rows.contiguous())
            return self._recv
        self._send[: local_rows.shape[0]].copy_(local_rows)
        self.dist.all_gather_into_tensor(self._recv, self._send)
        return torch.cat([self._recv[r * self.max_rows: r * self.max_rows + c] for r, c in enumerate(self.counts)])

    def barrier(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        if self.dist is not None:
            self.dist.barrier()
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)

    def timed(self, step_fn, warmup, steps, after_warmup=None):
        """W untimed + exactly K timed calls of step_fn(i); returns seconds, MAX over ranks.  The slowest and the fastest rank's
        own time are kept in `last_elapsed_max` / `last_elapsed_min` (load imbalance across ranks); `after_warmup()` runs
        between the warm-up and the first barrier (e.g. to snapshot counters that must not move inside the timed region)."""
        import time
        for i in range(warmup):
            step_fn(i)
        if after_warmup is not None:
            after_warmup()
        self.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step_fn(warmup + i)
        self.barrier()
        elapsed = time.perf_counter() - t0
        self.last_elapsed_min = self.last_elapsed_max = elapsed
        if self.dist is not None:
            t = torch.tensor([elapsed, -elapsed], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t[0].item())
            self.last_elapsed_max, self.last_elapsed_min = elapsed, -float(t[1].item())
        return elapsed
