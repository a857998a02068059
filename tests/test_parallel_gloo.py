"""gloo tests (CPU) of the data-parallel path at world 2, 4 and 8 (BASELINE.json configs[4] is the 8-GPU run): the flat gradient
bucket, the all-reduce per step, the parameter broadcast, and the three exchange schemes.  The kernels are not
involved: gradients are fabricated, what is tested is the collective plumbing bench.py uses - bucket aliasing, early + remainder
== full all-reduce, the repair paths, piece / shard boundaries at flat sizes that are NOT multiples of 64 * world, `first=`
pieces, sharded / all_to_all == replicated, per-rank checkpoints."""
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


WORLDS = [2, 4, 8]


def _spawn(target, world, *args):
    """`world` ranks of `target(rank, world, port, q, *args)`; every rank must exit 0 and report ok."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=420)
    for p in procs:
        if p.is_alive():
            p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5) for _ in range(world)) == [(r, "ok") for r in range(world)]


def _worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    torch.set_num_threads(1)                             # eight ranks share this container's eight cores
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
    b2.start_early()                         # backward is not writing into the bucket (no sink()): must refuse to run on stale memory
    assert b2._early_work is None
    b2.sink()                                # what train_step does before backward: the bucket views ARE the gradient storage
    b2.start_early()                         # grids in flight ...
    assert b2._early_work is not None and len(b2._early_work) >= 1
    b2.start_early()                         # (idempotent)
    b2.all_reduce_mean()                     # ... MLP range reduced, early work joined, everything scaled
    for k, p in enumerate(params):
        want_k = sum((r + 1) * (k + 1) for r in range(world)) / world
        assert torch.allclose(p.grad, torch.full_like(p.grad, want_k)), k
    # without the sink an "early" call is a no-op and replaced grads are copied back and reduced exactly once
    for p in params:
        p.grad.zero_()
    b2.start_early()
    for k, p in enumerate(params):
        p.grad = torch.full_like(p, float((rank + 1) * (k + 2)))
    b2.all_reduce_mean()
    for k, p in enumerate(params):
        want_k = sum((r + 1) * (k + 2) for r in range(world)) / world
        assert torch.allclose(p.grad, torch.full_like(p.grad, want_k)), k
    # ADVICE r1 (parallel.py repair path): backward wrote through the sink, the early collective is already rewriting the
    # bucket, and autograd CLONED instead of adopting the views - the clones of the early range may hold half-reduced
    # garbage.  The early range must come from the bucket memory (reduced exactly once), the rest from the copies.
    for k, v in enumerate(b2.views):
        v.fill_(float((rank + 1) * (k + 5)))          # what the kernels wrote
    b2.sink()
    b2.start_early()
    early_ids = {id(p) for p in early}
    for k, p in enumerate(params):
        if id(p) in early_ids:
            p.grad = torch.full_like(p, 1e30)         # a clone taken while the collective was in flight: garbage
        else:
            p.grad = torch.full_like(p, float((rank + 1) * (k + 5)))     # a valid clone of a range no collective has touched
    assert not b2.consistent()
    b2.all_reduce_mean()
    assert b2.consistent()
    for k, p in enumerate(params):
        want_k = sum((r + 1) * (k + 5) for r in range(world)) / world
        assert torch.allclose(p.grad, torch.full_like(p.grad, want_k)), (k, float(p.grad.flatten()[0]), want_k)
    # the same scenario through step_schedule (the optimizer path): falls back to the repair, nothing reduced twice
    for k, v in enumerate(b2.views):
        v.fill_(float((rank + 1) * (k + 6)))
    b2.sink()
    b2.start_early()
    for k, p in enumerate(params):
        p.grad = torch.full_like(p, 1e30 if id(p) in early_ids else float((rank + 1) * (k + 6)))
    assert b2.step_schedule() is None
    for k, p in enumerate(params):
        want_k = float(sum((r + 1) * (k + 6) for r in range(world)))       # step_schedule leaves SUMS
        assert torch.allclose(p.grad, torch.full_like(p.grad, want_k)), k
    # chunked early reduce + step schedule: pieces cross parameter boundaries, every element is scheduled exactly once,
    # and after all waits the buffer holds the SUM (the optimizer applies 1/world)
    b3 = parallel.GradBucket(params, early=early, chunk_elems=1000)
    assert len(b3.early_chunks()) > 4 and b3.early_chunks()[0][0] == 0 and b3.early_chunks()[-1][1] == b3._early_range[1]
    for k, p in enumerate(params):
        p.grad.fill_(float((rank + 1) * (k + 3)))
    b3.sink()
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


@pytest.mark.parametrize("world", WORLDS)
def test_grad_bucket_allreduce(world):
    _spawn(_worker, world)


def test_single_process_bucket_is_a_noop_collective():
    from nvp_amd import parallel
    lin = torch.nn.Linear(3, 2)
    b = parallel.GradBucket(parallel.unique_parameters(lin))
    lin.weight.grad.fill_(2.0)
    b.all_reduce_mean()                 # no process group: must not touch the values
    assert float(b.flat.sum()) == 2.0 * 6


# ------------------------------------------------------------------------------------------------------------
# ZeRO-1 (parallel.ShardedAdamW): reduce-scatter -> AdamW on the own shard -> all-gather of the parameters
# ------------------------------------------------------------------------------------------------------------
def _cpu_adamw_update(p, g, m, v, lr, b1, b2, eps, wd, step, grad_scale):
    """torch.optim.AdamW's update rule (training.py:13) on flat slices - the CHECKER for the sharded plumbing (the product's
    update is the HIP kernel nvp_adamw_step; element-wise, so slicing cannot change a single bit)."""
    import math
    gs = g * grad_scale
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(gs, alpha=1 - b1)
    v.mul_(b2).addcmul_(gs, gs, value=1 - b2)
    denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))


def _sharded_worker(rank, world, port, q, algo):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    torch.set_num_threads(1)
    import io
    from nvp_amd import parallel
    from nvp_amd.modules import NVP
    parallel.init_distributed(backend="gloo")
    cfg = small_cfg(F=2, T=16, X=20, Y=20, n_levels=4)       # sparse grid of 12 800 elements: several 3000-element pieces lie wholly inside it
    unit = parallel.ShardedAdamW.alignment(world)
    STEPS, CKPT_AT = 5, 2

    def build():
        torch.manual_seed(7)
        return NVP(out_features=3, encoding_config=cfg)

    def fake_grads(views, it, exact=True):
        """What backward would write through the sink: rank- and step-dependent.  exact: integers in [-1024, 1024] times a power of
        two that depends on (step, tensor) but NOT on the rank - any summation order over <= 8 ranks gives the same fp32 sum, so
        the three exchange schemes (gloo's all-reduce order, its reduce-scatter order, the rank-order local sum of the all_to_all
        form) must agree BIT FOR BIT at every world size.  exact=False: plain floats - the schemes then differ by fp32 summation
        order only (checked to a tolerance)."""
        g = torch.Generator().manual_seed(1000 * it + rank)
        ge = torch.Generator().manual_seed(77 * it)
        for v in views:
            e = float(torch.randint(-20, 1, (1,), generator=ge))
            if exact:
                v.copy_(torch.randint(-1024, 1025, v.shape, generator=g).float() * 2.0 ** e)
            else:
                v.copy_(torch.randn(v.shape, generator=g) * 2.0 ** e)

    def grids_of(m):
        return [m.keyframes_xy.params, m.keyframes_yt.params, m.keyframes_xt.params, m.sparse_grid.embeddings]

    def make_sharded():
        m = build()
        ps = parallel.unique_parameters(m)
        b = parallel.GradBucket(ps, early=grids_of(m), chunk_elems=3000, pad_to=unit)
        o = parallel.ShardedAdamW(b, lr=1e-2, weight_decay=1e-3, algo=algo, update=_cpu_adamw_update, first=[m.sparse_grid.embeddings])
        return m, ps, b, o, torch.optim.lr_scheduler.CosineAnnealingLR(o, T_max=STEPS, eta_min=1e-5)

    # ---- replicated reference: all-reduce (SUM), every rank applies the whole update with grad_scale = 1/world
    m_rep = build()
    p_rep = parallel.unique_parameters(m_rep)
    b_rep = parallel.GradBucket(p_rep, early=grids_of(m_rep), chunk_elems=3000)
    flat_p = torch.cat([p.detach().reshape(-1) for p in p_rep]).clone()
    m1, v1 = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
    # ---- sharded
    m_sh, p_sh, b_sh, opt, sched = make_sharded()
    # layout at this world size: the flat size is NOT a multiple of 64 * world (zero padding at the end), every piece splits into
    # `world` equal 256-B aligned shards, piece boundaries fall INSIDE the early (grid) range and inside parameters, the end of
    # the early range is not a piece boundary (its tail travels with the remainder piece)
    assert b_sh.numel % unit != 0 and b_sh.padded % unit == 0 and 0 < b_sh.padded - b_sh.numel < unit
    assert len(opt.pieces) >= 3 and opt.n_early >= 2 and opt.pieces[0][0] == 0 and opt.pieces[-1][1] == b_sh.padded
    assert all(opt.pieces[i][1] == opt.pieces[i + 1][0] for i in range(len(opt.pieces) - 1))
    assert all((b - a) % unit == 0 and (b - a) > 0 for a, b in opt.pieces)
    e_lo, e_hi = b_sh._early_range
    assert e_hi % unit != 0 and opt.pieces[opt.n_early - 1][1] == e_hi // unit * unit < e_hi
    inner = {a for a, _ in opt.pieces[1:opt.n_early]}
    assert inner and not inner & set(b_sh._offsets), "early piece boundaries should cut through parameters"
    for (a, b), (lo, hi, off) in zip(opt.pieces, opt.shards):
        assert hi - lo == (b - a) // world and lo == a + rank * (hi - lo) and lo % 64 == 0
    # the pieces that only hold the sparse grid can be exchanged before the dense planes' gradients exist
    emb_lo = b_sh._offsets[3]
    assert opt._first_pieces and all(opt.pieces[i][0] >= emb_lo for i in opt._first_pieces) and 0 not in opt._first_pieces
    assert opt.exp_avg.numel() * world == b_sh.padded                      # optimizer state is 1/world of the flat vector
    assert all(p.data_ptr() == opt.pflat[o:o + 1].data_ptr() for p, o in zip(p_sh, b_sh._offsets))    # parameters re-homed, values kept
    assert torch.equal(torch.cat([p.detach().reshape(-1) for p in p_sh]), flat_p)

    def sharded_step(m, ps, b, o, it, exact=True):
        if it % 2 == 0:
            # SPARSE_READY_HOOK: only the sparse grid's gradient exists yet - its pieces go out; the dense planes' gradients are
            # written AFTERWARDS (stale values in their range must not have been sent)
            for v in b.views[:3]:
                v.fill_(float("nan"))
            b.sink()
            g_all = [torch.empty_like(v) for v in b.views]
            fake_grads(g_all, it, exact)
            b.views[3].copy_(g_all[3])
            o.start_first()
            o.start_first()                         # (idempotent)
            for v, gsrc in zip(b.views, g_all):
                if v is not b.views[3]:
                    v.copy_(gsrc)
        else:
            fake_grads(b.views, it, exact)
            b.sink()
        o.start_early()                             # GRIDS_READY_HOOK: the (remaining) early pieces' reduce-scatter is in flight
        if it == 3:                                 # the MLP range lost its views (autograd cloned): repaired, exchanged once
            for p in ps[4:]:
                p.grad = p.grad.clone()
            assert not b.consistent()
        o.step()

    lr_now = 1e-2
    ckpt, after_ckpt = None, {}
    for it in range(1, STEPS + 1):
        exact = it < STEPS                            # the last step: plain float gradients (summation order differs between the schemes)
        # replicated
        fake_grads(b_rep.views, it, exact)
        b_rep.sink(); b_rep.start_early()
        assert b_rep.consistent()
        for wait, _ in b_rep.step_schedule():
            wait()
        _cpu_adamw_update(flat_p, b_rep.flat, m1, v1, lr_now, 0.9, 0.999, 1e-8, 1e-3, it, 1.0 / world)
        # sharded: same local gradients
        assert abs(opt.param_groups[0]["lr"] - lr_now) < 1e-15
        sharded_step(m_sh, p_sh, b_sh, opt, it, exact)
        sched.step()
        lr_now = opt.param_groups[0]["lr"]
        got = torch.cat([p.detach().reshape(-1) for p in p_sh])
        if exact:
            assert torch.equal(got, flat_p), f"step {it}: sharded parameters differ from the replicated AdamW path (max {float((got - flat_p).abs().max())})"
        else:
            # fp32 summation order: a gradient sum off by ulps moves m / (sqrt(v) + eps) by ~1e-6 of O(1): |dp| <= lr * 1e-5
            assert float((got - flat_p).abs().max()) <= 1e-2 * 1e-5, float((got - flat_p).abs().max())
        # replicas bit-identical after EVERY step, exact sums or not (every rank receives the same gathered shards)
        chk = torch.stack([opt.pflat.double().sum(), opt.pflat.double().abs().sum()])
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        assert all(torch.equal(g, gathered[0]) for g in gathered), f"step {it}: replicas diverged"
        if it == CKPT_AT:
            buf = io.BytesIO()                        # what training.py:65-68 does, per rank (the moments exist nowhere else)
            torch.save({"model": m_sh.state_dict(), "optimizer": opt.state_dict(), "scheduler": sched.state_dict()}, buf)
            ckpt = buf.getvalue()
        if it > CKPT_AT:
            after_ckpt[it] = got.clone()
        if it == STEPS - 1:
            # the moments of my shard equal the replicated moments of the same elements (exact sums so far: bit for bit)
            for (lo, hi, off) in opt.shards:
                n_real = max(0, min(hi, b_sh.numel) - lo)
                assert torch.equal(opt.exp_avg[off:off + n_real], m1[lo:lo + n_real]) and torch.equal(opt.exp_avg_sq[off:off + n_real], v1[lo:lo + n_real])
    # ---- checkpoint round trip at this world size: a fresh model + optimizer per rank, resumed from its own file, continues bit for bit
    m2, p2, b2, opt2, sched2 = make_sharded()
    ck = torch.load(io.BytesIO(ckpt), weights_only=False)
    m2.load_state_dict(ck["model"])
    opt2.load_state_dict(ck["optimizer"])
    sched2.load_state_dict(ck["scheduler"])
    assert opt2.steps_done == CKPT_AT and ck["optimizer"]["sharded"]["world"] == world and ck["optimizer"]["sharded"]["rank"] == rank
    assert all(p.data_ptr() == opt2.pflat[o:o + 1].data_ptr() for p, o in zip(p2, b2._offsets)), "load_state_dict must keep the parameters re-homed"
    for it in range(CKPT_AT + 1, STEPS + 1):
        sharded_step(m2, p2, b2, opt2, it, it < STEPS)
        sched2.step()
        assert torch.equal(torch.cat([p.detach().reshape(-1) for p in p2]), after_ckpt[it]), f"resumed run differs at step {it}"
    # ... and a state saved under another layout is refused loudly: another rank's file, and a world-2 run's file at this world size
    other = torch.load(io.BytesIO(ckpt), weights_only=False)["optimizer"]
    other["sharded"]["rank"] = (rank + 1) % world
    with pytest.raises(ValueError, match="rank"):
        opt2.load_state_dict(other)
    if world != 2:
        w2 = torch.load(io.BytesIO(ckpt), weights_only=False)["optimizer"]
        u2 = parallel.ShardedAdamW.alignment(2)
        pad2 = (b_sh.numel + u2 - 1) // u2 * u2
        w2["sharded"].update(world=2, rank=rank % 2, padded=pad2)       # what a world-2 run of the same model would have written
        with pytest.raises(ValueError, match="world"):
            opt2.load_state_dict(w2)
        w2["sharded"]["world"] = world                                  # even relabelled, its padding / pieces do not fit
        w2["sharded"]["rank"] = rank
        if pad2 != b_sh.padded:
            with pytest.raises(ValueError, match="padded"):
                opt2.load_state_dict(w2)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("algo", ["reduce_scatter", "all_to_all"])
def test_sharded_adamw_matches_replicated(algo, world):
    """reduce-scatter -> sharded AdamW -> all-gather against the all-reduce + full-AdamW path at world 2, 4 and 8 (configs[4] is the
    8-GPU run), over 5 steps with a cosine schedule, incl. a step whose MLP gradients lost their bucket views and sparse-first steps;
    both exchange algorithms (RCCL-style reduce_scatter, one-hop all_to_all + rank-order local sum).  Steps 1-4 use gradients whose
    cross-rank sums are exact in fp32 (integers times a rank-independent power of two): the schemes must then agree BIT FOR BIT
    whatever order a backend sums in; step 5 uses plain floats: replicas still bit-identical to each other, sharded vs replicated to
    |dp| <= lr * 1e-5 (fp32 summation order).  Per-rank checkpoint after step 2, resumed in a fresh optimizer: bit-identical
    continuation; a state written under another rank or world size is refused."""
    _spawn(_sharded_worker, world, algo)


def test_sharded_adamw_single_process_has_no_cpu_update():
    """Without a process group ShardedAdamW is a plain flat AdamW in pieces; its default update is the HIP kernel and
    refuses CPU tensors (no CPU path in the product)."""
    from nvp_amd import parallel
    lin = torch.nn.Linear(5, 3)
    b = parallel.GradBucket(parallel.unique_parameters(lin), pad_to=parallel.ShardedAdamW.alignment(1))
    opt = parallel.ShardedAdamW(b, lr=1e-2)
    lin.weight.grad.fill_(1.0)
    with pytest.raises(RuntimeError, match="HIP device"):
        opt.step()
    w0 = lin.weight.detach().clone()
    opt2 = parallel.ShardedAdamW(b, lr=1e-2, weight_decay=0.0, update=_cpu_adamw_update)
    lin.weight.grad.fill_(1.0); lin.bias.grad.fill_(0.0)
    opt2.step()
    assert torch.allclose(lin.weight, w0 - 1e-2, atol=1e-6)             # first Adam step: -lr * sign(g)


def test_sharded_adamw_state_dict_round_trip():
    """the reference checkpoints optim.state_dict() (training.py:65-68, 90-93).  ShardedAdamW keeps its moments and
    its step counter outside Optimizer.state, so state_dict() / load_state_dict() carry them explicitly: a run resumed from
    (parameters, optimizer state, scheduler state) continues bit for bit; a replicated state or another layout is refused."""
    from nvp_amd import parallel

    def build():
        torch.manual_seed(3)
        lin = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5))
        b = parallel.GradBucket(parallel.unique_parameters(lin), chunk_elems=256, early=[lin[0].weight], pad_to=parallel.ShardedAdamW.alignment(1))
        opt = parallel.ShardedAdamW(b, lr=1e-2, weight_decay=1e-3, update=_cpu_adamw_update)
        return lin, b, opt, torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=6, eta_min=1e-5)

    def run(lin, b, opt, sched, its):
        for it in its:
            g = torch.Generator().manual_seed(50 + it)
            for v in b.views:
                v.copy_(torch.randn(v.shape, generator=g))
            b.sink()
            opt.step()
            sched.step()

    lin, b, opt, sched = build()
    run(lin, b, opt, sched, range(3))
    ck = {"model": {k: v.clone() for k, v in lin.state_dict().items()}, "optimizer": opt.state_dict(), "scheduler": sched.state_dict()}
    assert ck["optimizer"]["sharded"]["steps_done"] == 3 and float(ck["optimizer"]["sharded"]["exp_avg"].abs().sum()) > 0
    import io
    buf = io.BytesIO()
    torch.save(ck, buf)                                   # what training.py:65-68 does with it
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    run(lin, b, opt, sched, range(3, 6))
    want = [p.detach().clone() for p in lin.parameters()]

    lin2, b2, opt2, sched2 = build()
    lin2.load_state_dict(ck["model"])
    opt2.load_state_dict(ck["optimizer"])
    sched2.load_state_dict(ck["scheduler"])
    assert opt2.steps_done == 3 and abs(opt2.param_groups[0]["lr"] - ck["optimizer"]["param_groups"][0]["lr"]) == 0
    assert all(p.data_ptr() == opt2.pflat[o:o + 1].data_ptr() for p, o in zip(b2.params, b2._offsets)), "load_state_dict must keep the parameters re-homed"
    run(lin2, b2, opt2, sched2, range(3, 6))
    for a, c in zip(want, lin2.parameters()):
        assert torch.equal(a, c), "resumed run differs from the uninterrupted one"
    # a replicated optimizer's state has no sharded moments; another layout is refused loudly
    with pytest.raises(ValueError, match="sharded"):
        opt2.load_state_dict(torch.optim.AdamW(lin2.parameters()).state_dict())
    bad = opt.state_dict()
    bad["sharded"]["world"] = 2
    with pytest.raises(ValueError, match="world"):
        opt2.load_state_dict(bad)
