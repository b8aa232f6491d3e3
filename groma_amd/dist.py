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


def init(backend=None, device=None):
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, **kw)
    return dist


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
