"""Pixel-batch data parallelism: one process per GPU, every rank draws its own i.i.d. pixel batch,
parameters are replicated, and after backward the flat fp32 gradient (543 MB for nvp_s) is summed across ranks over
RCCL / xGMI (backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).  Weak scaling: the per-GPU batch stays
N = 1 245 184, the global batch is world * N.

Two exchange schemes share the flat buffers of `GradBucket`:

* replicated (`GradBucket.step_schedule` + `optim.AdamW(schedule=...)`): all-reduce, every rank runs the whole AdamW.
  The grid range (99.9 % of the bytes) is reduced early, asynchronously and in pieces underneath the dW GEMMs, and AdamW
  updates piece i while piece i+1 is on the wire.
* sharded (`ShardedAdamW`, ZeRO-1): the same pieces are REDUCE-SCATTERED (each rank receives the sum of 1/world of every
  piece), each rank runs AdamW on its shard only (optimizer traffic and state / world), and the updated parameters are
  ALL-GATHERED in place into a flat parameter buffer the module's parameters are views of.  Same bytes on the wire as an
  all-reduce (which is a reduce-scatter + all-gather), but the all-gather of piece i overlaps the update of piece i+1 and
  the 3.8 GB optimizer pass shrinks to 3.8 / world GB.  xGMI is point-to-point (7 links per GPU): besides RCCL's own
  reduce-scatter the one-hop direct form is available (`algo="all_to_all"`): every rank sends shard j of its piece
  straight to rank j over their private link and sums the world received shards locally.

The flat buffers are persistent: `GradBucket` points every parameter's `.grad` at a view of one contiguous tensor, so
a collective is a single call on a slice of it and no per-step flatten/unflatten copy exists.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Iterable, List, Optional

import torch
import torch.distributed as dist

# ShardedAdamW: update + all-gather the grid shards on a side stream, underneath the dW GEMMs (0: everything on the compute stream)
EARLY_UPDATE = os.environ.get("NVP_DP_EARLY_UPDATE", "1") != "0"


def _force() -> bool:
    """NVP_DP_FORCE_COLLECTIVES=1: run the collective code paths even with ONE rank (a single-GPU box can then exercise
    the real RCCL calls - init, barrier, chunked asynchronous collectives on slices of the flat buffer, waits)."""
    return os.environ.get("NVP_DP_FORCE_COLLECTIVES", "0") == "1"


def _multi() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _force())


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _force()) and not dist.is_initialized():
        if backend is None:
            # NVP_DIST_BACKEND=gloo: smoke-test the multi-rank code paths with several ranks on ONE GPU (RCCL refuses that)
            backend = os.environ.get("NVP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def unique_parameters(module: torch.nn.Module) -> List[torch.nn.Parameter]:
    """NVP registers its SirenNet twice (net and wrapper.net); parameters() already dedups."""
    return [p for p in module.parameters() if p.requires_grad]


class GradBucket:
    """All gradients of a module as views into one flat fp32 buffer + the per-step all-reduce."""

    def __init__(self, params: Iterable[torch.nn.Parameter], early: Optional[Iterable[torch.nn.Parameter]] = None,
                 chunk_elems: int = 32 * 1024 * 1024, pad_to: int = 1):
        """`early`: parameters whose gradients are complete before the rest of backward has run (NVP's four
        grids: 99.9 % of the bytes).  If they occupy one contiguous range of the flat buffer, their exchange
        can be started early and asynchronously (`start_early`) and overlaps the remaining backward kernels.
        `pad_to`: the flat buffer's length is rounded up to a multiple of this (zeros; ShardedAdamW needs equal shards)."""
        self.params = list(params)
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.padded = (self.numel + pad_to - 1) // pad_to * pad_to
        self._flat_all = torch.zeros(self.padded, device=dev, dtype=torch.float32)
        self.flat = self._flat_all[:self.numel]
        self.views = []
        off = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("GradBucket expects fp32 parameters on one device")
            v = self.flat[off:off + p.numel()].view_as(p)
            self.views.append(v)
            off += p.numel()
        self.attach()
        self._early_range = None
        self._early_work = None          # list of (lo, hi, work) once start_early() has run
        self._sink_armed = False         # sink() handed the views to backward for the step in progress
        self.chunk_elems = int(chunk_elems)          # the early range is reduced in pieces of <= this many elements (128 MB)
        self._offsets = []
        o = 0
        for p in self.params:
            self._offsets.append(o)
            o += p.numel()
        if early is not None:
            ids = {id(p) for p in early}
            offs, o = [], 0
            for p in self.params:
                if id(p) in ids:
                    offs.append((o, o + p.numel()))
                o += p.numel()
            if offs and len(offs) == len(ids):
                lo, hi = min(a for a, _ in offs), max(b for _, b in offs)
                if sum(b - a for a, b in offs) == hi - lo:            # contiguous: nothing else in between
                    self._early_range = (lo, hi)

    def early_chunks(self) -> list:
        """[(lo, hi)] pieces of the early range, each <= chunk_elems elements."""
        if self._early_range is None:
            return []
        lo, hi = self._early_range
        step = max(1, self.chunk_elems)
        return [(a, min(a + step, hi)) for a in range(lo, hi, step)]

    def start_early(self) -> None:
        """Asynchronous all-reduces of the early range, piece by piece (call once its gradients are enqueued on the
        current stream; torch.distributed orders the collectives after that work).  Pieces let the optimizer update
        the parameters of piece i while piece i+1 is still on the wire (`step_schedule`).  Only runs when backward is
        writing into the bucket (`sink()` armed for this step): otherwise the memory it would reduce is stale.
        No-op without a group."""
        if self._early_range is None or self._early_work is not None or not self._sink_armed:
            return
        if _multi():
            self._early_work = [(a, b, dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))
                                for a, b in self.early_chunks()]

    def _join_early(self) -> None:
        if self._early_work is not None:
            for _, _, w in self._early_work:
                w.wait()
            self._early_work = None

    def params_in(self, lo: int, hi: int) -> list:
        """[(param, a, b)]: the sub-ranges [a, b) (in elements of each parameter) that flat range [lo, hi) covers."""
        out = []
        for p, o in zip(self.params, self._offsets):
            a, b = max(lo, o), min(hi, o + p.numel())
            if a < b:
                out.append((p, a - o, b - o))
        return out

    def _repair(self) -> Optional[tuple]:
        """Some `.grad` is not the bucket view any more (autograd cloned instead of adopting the tensor backward returned,
        or something replaced it).  Make the bucket authoritative again WITHOUT double counting:

        * the range an early collective already ran on was written by backward through the sink (start_early refuses to run
          otherwise), so after joining, the bucket memory of that range IS the cross-rank sum; a clone autograd may have
          taken meanwhile can hold half-reduced data (it was copied on the compute stream while the collective rewrote the
          memory) and is simply dropped - it must neither be copied back nor reduced again;
        * everything else is copied from `.grad` into the bucket and still has to be reduced.
        Returns the flat range that is already reduced (or None)."""
        done = None
        if self._early_work is not None:
            self._join_early()
            done = self._early_range
        for p, v, o in zip(self.params, self.views, self._offsets):
            inside = done is not None and done[0] <= o and o + p.numel() <= done[1]
            if inside:
                continue                               # bucket memory is the reduced gradient
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
        self.attach()
        return done

    def step_schedule(self) -> Optional[list]:
        """For an optimizer that can update sub-ranges (nvp_amd.optim.AdamW.step(schedule=...)): reduce what is not
        in flight yet, and return [(wait, [(param, a, b), ...]), ...] in completion order - wait() makes the
        current stream wait for that piece's collective.  Gradients stay SUMS (the caller passes 1/world as
        grad_scale).  Returns None for a single process (nothing to wait for)."""
        self._sink_armed = False
        if not _multi():
            return None
        if not self.consistent():
            self.all_reduce(scale=False)            # repair path: everything reduced (exactly once) synchronously
            return None
        sched = []
        if self._early_work is None:
            w = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
            return [(w.wait, self.params_in(0, self.numel))]
        lo, hi = self._early_range
        rest = []
        if lo > 0:
            rest.append((0, lo, dist.all_reduce(self.flat[:lo], op=dist.ReduceOp.SUM, async_op=True)))
        if hi < self.numel:
            rest.append((hi, self.numel, dist.all_reduce(self.flat[hi:], op=dist.ReduceOp.SUM, async_op=True)))
        for a, b, w in self._early_work + rest:      # collectives complete in issue order
            sched.append((w.wait, self.params_in(a, b)))
        self._early_work = None
        return sched

    def attach(self) -> None:
        """(Re)bind .grad to the bucket views (after zero_grad(set_to_none=True) or a rebuild)."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero_(self) -> None:
        self.flat.zero_()
        self.attach()

    def sink(self) -> dict:
        """{param.data_ptr(): bucket view} for nvp_amd.functional.StepHooks.grad_sink: backward then writes each
        gradient directly into the flat buffer (every element exactly once - no zero-fill needed)."""
        self._sink_armed = True
        return {p.data_ptr(): v for p, v in zip(self.params, self.views)}

    def detach_grads(self) -> None:
        """Drop .grad so autograd adopts the tensors backward returns (the bucket views) without a copy."""
        for p in self.params:
            p.grad = None

    def consistent(self) -> bool:
        return all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.views))

    @staticmethod
    def world_size() -> int:
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def all_reduce_mean(self) -> None:
        """One collective over the whole gradient, then 1/world; no-op for a single process."""
        self.all_reduce(scale=True)

    def all_reduce(self, scale: bool = True) -> None:
        """SUM all-reduce of the flat gradient.  scale=False leaves the sum (the caller folds 1/world into
        the optimizer kernel: nvp_amd.optim.AdamW.step(grad_scale=...))."""
        self._sink_armed = False
        done = None
        if not self.consistent():
            done = self._repair()
        if _multi():
            if self._early_work is not None:
                # the big range is already in flight (overlapping the dW GEMMs): reduce the rest, then join
                done = self._early_range
                self._reduce_outside(done)
                self._join_early()
            elif done is not None:
                self._reduce_outside(done)            # repaired: the early range is already summed, reduce only the rest
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)       # one collective over the whole gradient
            if scale:
                self.flat.mul_(1.0 / dist.get_world_size())
        else:
            self._join_early()

    def _reduce_outside(self, rng: tuple) -> None:
        lo, hi = rng
        if lo > 0:
            dist.all_reduce(self.flat[:lo], op=dist.ReduceOp.SUM)
        if hi < self.numel:
            dist.all_reduce(self.flat[hi:], op=dist.ReduceOp.SUM)


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Make every rank start from rank `src`'s parameters (replicated-parameter DP)."""
    if _multi():
        for p in unique_parameters(module):
            dist.broadcast(p.data, src=src)


# ------------------------------------------------------------------------------------------------------------
# ZeRO-1: reduce-scatter -> AdamW on the own shard -> all-gather of the parameters
# ------------------------------------------------------------------------------------------------------------
def _hip_adamw_update(p, g, m, v, lr, b1, b2, eps, wd, step, grad_scale) -> None:
    """AdamW on four flat fp32 device slices of equal length in one launch of nvp_adamw_step (no CPU path)."""
    from . import _lib
    lib = _lib.load()
    seg = _lib.AdamwSeg(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), p.numel())
    arr = (_lib.AdamwSeg * 1)(seg)
    _lib.check(lib.nvp_adamw_step(arr, 1, float(lr), float(b1), float(b2), float(eps), float(wd), int(step), float(grad_scale),
                                  _lib.stream_ptr()), "nvp_adamw_step")


class ShardedAdamW(torch.optim.Optimizer):
    """The reference's AdamW (training.py:13) with the state and the update sharded over the data-parallel ranks.

    * `bucket` must have been built with pad_to=ShardedAdamW.alignment(world): every piece then splits into `world`
      equal, 256-B aligned shards.
    * The module's parameters are re-homed: `p.data` becomes a view of one flat parameter buffer (values preserved), so
      that the updated shards can be all-gathered in place.
    * `param_groups[0]['lr']` is what CosineAnnealingLR drives, exactly as with torch.optim.AdamW; `state` is sharded
      (`exp_avg`, `exp_avg_sq` of this rank's 1/world of the flat parameter vector) and `step` is one global counter.
    * step(): for every piece in completion order: wait for its reduce-scatter, update the own shard (gradient SUM times
      1/world inside the kernel), start the asynchronous all-gather of that piece's parameters; all gathers are joined on
      the current stream before step() returns (stream-ordered waits, no host sync).
    `update` is the flat-slice AdamW kernel launcher (default: nvp_adamw_step on the HIP device).  The world-2 gloo test
    passes a CPU restatement of torch.optim.AdamW's rule instead - tests own their checker, the product has no CPU path."""

    @staticmethod
    def alignment(world: int) -> int:
        return 64 * max(world, 1)

    def __init__(self, bucket: GradBucket, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 algo: str = "reduce_scatter", update: Optional[Callable] = None, first: Optional[Iterable[torch.nn.Parameter]] = None,
                 emulate_world: int = 0):
        """`emulate_world` (MEASUREMENT ONLY, single process without a group; bench.py's `dp_floor`): lay the pieces and shards out as rank 0
        of a world of that size and update only that shard - the per-GPU compute of an N-rank step (gradient route + AdamW on 1/N of the
        parameters) without any exchange.  The other (N-1)/N of the parameters are NOT updated: not a training mode."""
        if algo not in ("reduce_scatter", "all_to_all"):
            raise ValueError("algo must be 'reduce_scatter' or 'all_to_all'")
        super().__init__(bucket.params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.bucket = bucket
        self.algo = algo
        self.update = update or _hip_adamw_update
        self.world = dist.get_world_size() if _multi() else 1
        self.rank = dist.get_rank() if _multi() else 0
        if emulate_world > 1 and (self.world > 1 or _multi()):
            raise ValueError("emulate_world is a single-process measurement aid")
        self.layout_world = int(emulate_world) if emulate_world > 1 else self.world
        unit = self.alignment(self.layout_world)
        if bucket.padded % unit:
            raise ValueError(f"GradBucket must be padded to a multiple of {unit} elements (pad_to=ShardedAdamW.alignment(world))")
        dev = bucket.flat.device
        # ---- flat parameter buffer; the parameters become views of it
        self.pflat = torch.zeros(bucket.padded, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(bucket.params, bucket._offsets):
                self.pflat[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.pflat[o:o + p.numel()].view_as(p)
        # ---- pieces: the early range in chunks whose boundaries are multiples of `unit`, then one remainder piece
        self.pieces = []
        e_hi = 0
        if bucket._early_range is not None and bucket._early_range[0] == 0:
            e_hi = bucket._early_range[1] // unit * unit
            step = max(unit, bucket.chunk_elems // unit * unit)
            self.pieces = [(a, min(a + step, e_hi)) for a in range(0, e_hi, step)]
        self.n_early = len(self.pieces)
        if e_hi < bucket.padded:
            self.pieces.append((e_hi, bucket.padded))
        # ---- this rank's shard of every piece, and where it lives in the (sharded) moment buffers
        self.shards, off = [], 0
        for a, b in self.pieces:
            s = (b - a) // self.layout_world
            self.shards.append((a + self.rank * s, a + (self.rank + 1) * s, off))
            off += s
        self.exp_avg = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(off, device=dev, dtype=torch.float32)
        self.steps_done = 0
        self._scratch = torch.empty(bucket.padded, device=dev, dtype=torch.float32) if (algo == "all_to_all" and (self.world > 1 or _force())) else None
        self._early_work = None          # [(piece index, wait)] in launch order, once start_first / start_early have run
        self._early_event = None
        self._first_event = None
        self._first_started = False
        self._early_started = False
        self._side = None
        # `first`: early parameters whose gradients are complete BEFORE the other early ones (NVP: the sparse grid, which the
        # scatter produces first).  The early pieces that lie wholly inside them can be exchanged from start_first() on.
        self._first_pieces = []
        if first is not None:
            ids = {id(p) for p in first}
            rng = [(o, o + p.numel()) for p, o in zip(bucket.params, bucket._offsets) if id(p) in ids]
            if rng and len(rng) == len(ids):
                f_lo, f_hi = min(a for a, _ in rng), max(e for _, e in rng)
                if sum(e - a for a, e in rng) == f_hi - f_lo:
                    self._first_pieces = [i for i in range(self.n_early) if self.pieces[i][0] >= f_lo and self.pieces[i][1] <= f_hi]

    # -- checkpointing (the reference saves optim.state_dict() next to the model: training.py:65-68, 90-93)
    def state_dict(self):
        """torch's layout ({'state', 'param_groups'}: lr, betas, eps, weight_decay - what CosineAnnealingLR drives) plus a
        `sharded` entry with THIS RANK's share of the optimizer state: the AdamW moments of its shards, the global step count
        (bias correction) and the piece / shard layout they refer to.  Every rank saves its own file (the moments exist
        nowhere else); load_state_dict refuses a state whose layout (world size, rank, pieces) differs from this optimizer's."""
        sd = super().state_dict()
        sd["sharded"] = {"world": self.world, "rank": self.rank, "algo": self.algo, "padded": int(self.bucket.padded),
                         "pieces": [tuple(int(v) for v in pc) for pc in self.pieces],
                         "shards": [tuple(int(v) for v in sh) for sh in self.shards],
                         "steps_done": int(self.steps_done),
                         "exp_avg": self.exp_avg.detach().clone(), "exp_avg_sq": self.exp_avg_sq.detach().clone()}
        return sd

    def load_state_dict(self, state_dict) -> None:
        sh = state_dict.get("sharded")
        if sh is None:
            raise ValueError("not a ShardedAdamW state_dict: no 'sharded' entry (a replicated torch.optim.AdamW / nvp_amd.optim.AdamW "
                             "state holds whole-tensor moments and cannot be loaded into a sharded optimizer)")
        mine = {"world": self.world, "rank": self.rank, "padded": int(self.bucket.padded),
                "pieces": [tuple(int(v) for v in pc) for pc in self.pieces], "shards": [tuple(int(v) for v in s_) for s_ in self.shards]}
        for k, v in mine.items():
            got = sh[k] if k in ("world", "rank", "padded") else [tuple(int(x) for x in t) for t in sh[k]]
            if got != v:
                raise ValueError(f"ShardedAdamW.load_state_dict: the saved state's {k} ({got if k in ('world', 'rank', 'padded') else '...'}) "
                                 f"does not match this optimizer's ({v if k in ('world', 'rank', 'padded') else '...'}): moments are sharded "
                                 "per rank and per piece layout")
        if sh["exp_avg"].numel() != self.exp_avg.numel() or sh["exp_avg_sq"].numel() != self.exp_avg_sq.numel():
            raise ValueError("ShardedAdamW.load_state_dict: moment buffers have the wrong length")
        super().load_state_dict({"state": state_dict.get("state", {}), "param_groups": state_dict["param_groups"]})
        with torch.no_grad():
            self.exp_avg.copy_(sh["exp_avg"].to(self.exp_avg.device))
            self.exp_avg_sq.copy_(sh["exp_avg_sq"].to(self.exp_avg_sq.device))
        self.steps_done = int(sh["steps_done"])

    # -- exchange of one piece: after wait(), flat[own shard] holds the cross-rank SUM of that shard
    def _reduce_piece(self, i: int):
        a, b = self.pieces[i]
        lo, hi, _ = self.shards[i]
        g = self.bucket._flat_all
        if self.world == 1 and not _force():
            return lambda: None
        if self.algo == "reduce_scatter":
            w = dist.reduce_scatter_tensor(g[lo:hi], g[a:b], op=dist.ReduceOp.SUM, async_op=True)    # in place: output = input + rank * count
            return w.wait
        recv = self._scratch[a:b]
        w = dist.all_to_all_single(recv, g[a:b], async_op=True)       # shard j of my piece -> rank j, one hop over our private link

        def fin():
            w.wait()
            torch.sum(recv.view(self.world, hi - lo), dim=0, out=g[lo:hi])          # ranks summed in rank order: deterministic
        return fin

    def start_first(self) -> None:
        """Start the exchange of the early pieces that only hold `first` parameters; call once THEIR gradients are enqueued
        (functional.StepHooks.sparse_ready).  Optional: start_early() picks up whatever has not been started."""
        if self._first_started or self._early_started or not self.bucket._sink_armed or not self._first_pieces:
            return
        self._first_started = True
        self._early_work = [(i, self._reduce_piece(i)) for i in self._first_pieces]
        if self.pflat.is_cuda:
            self._first_event = torch.cuda.Event()
            self._first_event.record()              # the `first` gradients are complete at this point of the compute stream

    def start_early(self) -> None:
        """Start the exchange of the (remaining) early grid pieces; call once their gradients are enqueued (functional.StepHooks.grids_ready)."""
        if self._early_started or not self.bucket._sink_armed:
            return
        self._early_started = True
        started = {i for i, _ in (self._early_work or [])}
        self._early_work = (self._early_work or []) + [(i, self._reduce_piece(i)) for i in range(self.n_early) if i not in started]
        if self.pflat.is_cuda:
            self._early_event = torch.cuda.Event()
            self._early_event.record()              # every grid gradient is complete at this point of the compute stream

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise RuntimeError("ShardedAdamW.step does not take a closure")
        b = self.bucket
        b._sink_armed = False
        early = list(self._early_work or [])
        first_ev, early_ev = self._first_event, self._early_event
        self._early_work, self._first_event, self._early_event = None, None, None
        self._first_started = self._early_started = False
        if not b.consistent():
            # a gradient did not land in the bucket.  Pieces already exchanged were written through the sink and are valid
            # (see GradBucket._repair): bucket memory stands inside them; everything else is copied back before it is exchanged.
            done = sorted(self.pieces[i] for i, _ in early)
            for p, v, o in zip(b.params, b.views, b._offsets):
                if p.grad is not None and p.grad.data_ptr() == v.data_ptr():
                    continue
                flat_v = v.reshape(-1)
                flat_g = None if p.grad is None else p.grad.reshape(-1)
                pos = o
                for a, e in done + [(o + p.numel(), o + p.numel())]:
                    hi = min(max(a, o), o + p.numel())         # [pos, hi): not covered by an exchanged piece
                    if hi > pos:
                        if flat_g is None:
                            flat_v[pos - o:hi - o].zero_()
                        else:
                            flat_v[pos - o:hi - o].copy_(flat_g[pos - o:hi - o])
                    pos = max(pos, min(e, o + p.numel()))
            b.attach()
        group = self.param_groups[0]
        b1, b2 = group["betas"]
        self.steps_done += 1
        gathers = []

        def finish_piece(i, wait):
            wait()
            lo, hi, off = self.shards[i]
            self.update(self.pflat[lo:hi], b._flat_all[lo:hi], self.exp_avg[off:off + hi - lo], self.exp_avg_sq[off:off + hi - lo],
                        group["lr"], b1, b2, group["eps"], group["weight_decay"], self.steps_done, 1.0 / self.world)
            if self.world > 1 or _force():
                a, e = self.pieces[i]
                gathers.append(dist.all_gather_into_tensor(self.pflat[a:e], self.pflat[lo:hi], async_op=True))   # in place

        # Early (grid) pieces: update + parameter all-gather on a SIDE stream that only waits for the pieces' own exchange, not for
        # the dW GEMMs still running on the compute stream (nothing after the scatter reads a grid parameter, and the MLP
        # parameters the dW kernels do read sit in the remainder piece).  The grid shards' AdamW and most of their all-gather then
        # run underneath the dW GEMMs instead of after them.  NVP_DP_EARLY_UPDATE=0: everything on the compute stream, in order.
        side = None
        if early and EARLY_UPDATE and self.pflat.is_cuda and early_ev is not None:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.pflat.device)
            side = self._side
            n_first = len(self._first_pieces) if first_ev is not None else 0
            side.wait_event(first_ev if n_first else early_ev)
            with torch.cuda.stream(side):
                for k, (i, wait) in enumerate(early):
                    if k == n_first and n_first:
                        side.wait_event(early_ev)       # from here on the pieces hold the later (dense-plane) gradients
                    finish_piece(i, wait)
        else:
            for i, wait in early:
                finish_piece(i, wait)
        started = {i for i, _ in early}
        for i in range(len(self.pieces)):
            if i not in started:
                finish_piece(i, self._reduce_piece(i))
        if side is not None:
            torch.cuda.current_stream(self.pflat.device).wait_stream(side)
        for w in gathers:
            w.wait()
        return None
