"""world_size-2 gloo run of the multi-GPU exchange step used by bench.py: shard, compute locally,
one all-gather of the per-image rows."""
import os

import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from groma_amd import dist as gdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = 5  # ragged: rank 0 gets 3 images, rank 1 gets 2
    lo, hi = gdist.shard_range(B, world, rank)
    full = torch.arange(B * 100, dtype=torch.float32).view(B, 100)
    local = full[lo:hi] * 2.0  # "forward" of the local shard
    counts = [gdist.shard_range(B, world, r)[1] - gdist.shard_range(B, world, r)[0] for r in range(world)]
    got = gdist.all_gather_rows(local, counts)
    even = gdist.all_gather_rows(torch.full((2, 4), float(rank)))
    # eval counters (groma/eval/eval_rec.py:122-124: three scalar reduces) -> one all-reduce of the RecMeter state
    from groma_amd import evalkit
    m = evalkit.RecMeter(0.5)
    tok = list(range(32014, 32114))
    seq = torch.tensor([[5, 6, tok[0]], [5, 6, 7]]) if rank == 0 else torch.tensor([[5, 6, tok[1]]])
    box = [torch.tensor([[0.5, 0.5, 0.2, 0.2], [0.1, 0.1, 0.1, 0.1]])] * seq.shape[0]
    gt = [torch.tensor([[0.5, 0.5, 0.2, 0.2]])] * seq.shape[0]
    m.update(seq, 2, box, gt, tok)
    summ = m.summary()
    # bench.py's own per-step driver (groma_amd.dist.ShardedJob): weak shard (even) and strong shard (ragged 5 over 2)
    ok_job = True
    for kw in (dict(rows_per_rank=3), dict(global_batch=5)):
        job = gdist.ShardedJob("cpu", (100,), torch.float32, **kw)
        ref = torch.arange(job.global_batch * 100, dtype=torch.float32).view(job.global_batch, 100)
        calls = []

        def step(i, job=job, ref=ref, calls=calls):
            calls.append(i)
            return job.exchange(ref[job.lo:job.hi] * (i + 1.0))
        elapsed = job.timed(step, warmup=1, steps=3)
        ok_job = ok_job and calls == [0, 1, 2, 3] and elapsed > 0 and job.rows == job.hi - job.lo
        ok_job = ok_job and torch.equal(step(4), ref * 5.0) and sum(job.counts) == job.global_batch
    q.put((rank, ok_job and torch.equal(got, full * 2.0) and summ["count"] == 3 and abs(summ["iou@0.5 accu"] - 1 / 3) < 1e-12
           and abs(summ["missing percentage"] - 1 / 3) < 1e-12, even[:, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_all_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    assert all(ev == [0.0, 0.0, 1.0, 1.0] for _, _, ev in res)
