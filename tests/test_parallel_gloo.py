"""world_size-2 gloo tests (CPU) of the data-parallel path: the flat gradient bucket, the
single all-reduce per step and the parameter broadcast.  The kernels are not
involved: gradients are fabricated, what is tested is the collective plumbing bench.py uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import small_cfg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    from nvp_amd import parallel
    from nvp_amd.modules import NVP
    r, w, _ = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # deliberately different init per rank
    cfg = small_cfg(F=2, T=4, X=5, Y=5, n_levels=4)
    model = NVP(out_features=3, encoding_config=cfg)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(rank)                                 # rank-dependent parameters
    parallel.broadcast_parameters(model, src=0)
    params = parallel.unique_parameters(model)
    chk = torch.stack([p.detach().double().sum() for p in params])
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    assert all(torch.equal(g, gathered[0]) for g in gathered), "broadcast did not replicate the parameters"

    bucket = parallel.GradBucket(params)
    n_el = sum(p.numel() for p in params)
    assert bucket.flat.numel() == n_el
    # the shared SirenNet (net == wrapper.net) must appear once
    assert len({id(p) for p in params}) == len(params) == 4 + 14
    # .grad tensors alias the flat buffer
    for p in params:
        p.grad.fill_(float(rank + 1))
    assert bool((bucket.flat == rank + 1).all()) and bucket.consistent()
    bucket.all_reduce_mean()
    want = sum(range(1, world + 1)) / world
    assert torch.allclose(bucket.flat, torch.full_like(bucket.flat, want))
    assert all(torch.allclose(p.grad, torch.full_like(p.grad, want)) for p in params)

    # zero_() keeps the aliasing; an optimizer-style zero_grad(set_to_none=True) is repaired
    bucket.zero_()
    assert float(bucket.flat.abs().sum()) == 0 and bucket.consistent()
    for p in params:
        p.grad = torch.full_like(p, float(10 * (rank + 1)))    # autograd replaced the tensors
    assert not bucket.consistent()
    bucket.all_reduce_mean()
    want = sum(10 * (k + 1) for k in range(world)) / world
    assert bucket.consistent() and torch.allclose(bucket.flat, torch.full_like(bucket.flat, want))

    # early (asynchronous) all-reduce of the grid range + the remainder afterwards == one full all-reduce
    early = [model.keyframes_xy.params, model.keyframes_yt.params, model.keyframes_xt.params, model.sparse_grid.embeddings]
    b2 = parallel.GradBucket(params, early=early)
    assert b2._early_range == (0, sum(p.numel() for p in early))        # the grids are the first, contiguous params
    for k, p in enumerate(params):
        p.grad.fill_(float((rank + 1) * (k + 1)))
    b2.start_early()                         # grids in flight ...
    b2.start_early()                         # (idempotent)
    b2.all_reduce_mean()                     # ... MLP range reduced, early work joined, everything scaled
    for k, p in enumerate(params):
        want_k = sum((r + 1) * (k + 1) for r in range(world)) / world
        assert torch.allclose(p.grad, torch.full_like(p.grad, want_k)), k
    # early range reduced on stale memory (grads replaced afterwards) must still give the right answer
    for p in params:
        p.grad.zero_()
    b2.start_early()
    for k, p in enumerate(params):
        p.grad = torch.full_like(p, float((rank + 1) * (k + 2)))
    b2.all_reduce_mean()
    for k, p in enumerate(params):
        want_k = sum((r + 1) * (k + 2) for r in range(world)) / world
        assert torch.allclose(p.grad, torch.full_like(p.grad, want_k)), k
    # chunked early reduce + step schedule: pieces cross parameter boundaries, every element is scheduled exactly once,
    # and after all waits the buffer holds the SUM (the optimizer applies 1/world)
    b3 = parallel.GradBucket(params, early=early, chunk_elems=1000)
    assert len(b3.early_chunks()) > 4 and b3.early_chunks()[0][0] == 0 and b3.early_chunks()[-1][1] == b3._early_range[1]
    for k, p in enumerate(params):
        p.grad.fill_(float((rank + 1) * (k + 3)))
    b3.start_early()
    sched = b3.step_schedule()
    seen = {id(p): torch.zeros(p.numel(), dtype=torch.int32) for p in params}
    for wait, items in sched:
        wait()
        for p, a, b_ in items:
            assert 0 <= a < b_ <= p.numel()
            seen[id(p)][a:b_] += 1
            off = b3._offsets[[id(q) for q in params].index(id(p))]
            k = [id(q) for q in params].index(id(p))
            want_sum = float(sum((r + 1) * (k + 3) for r in range(world)))
            assert bool((b3.flat[off + a:off + b_] == want_sum).all()), "piece scheduled before its reduce completed"
    assert all(bool((v == 1).all()) for v in seen.values()), "every element must be scheduled exactly once"
    assert b3._early_work is None and b3.step_schedule() is not None      # a second call reduces everything again (one piece)
    bucket.attach()

    # identical AdamW steps from identical averaged grads keep the replicas bit-identical
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=1e-3)
    opt.step()
    chk = torch.stack([p.detach().double().sum() for p in params])
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_grad_bucket_allreduce_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, "ok"), (1, "ok")]


def test_single_process_bucket_is_a_noop_collective():
    from nvp_amd import parallel
    lin = torch.nn.Linear(3, 2)
    b = parallel.GradBucket(parallel.unique_parameters(lin))
    lin.weight.grad.fill_(2.0)
    b.all_reduce_mean()                 # no process group: must not touch the values
    assert float(b.flat.sum()) == 2.0 * 6
