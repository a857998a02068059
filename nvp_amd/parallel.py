"""Pixel-batch data parallelism: one process per GPU, every rank draws its
own i.i.d. pixel batch, parameters are replicated, and after backward ONE all-reduce
(RCCL over xGMI; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests) sums a single
flat fp32 gradient buffer, which is then scaled by 1/world.  Weak scaling: the per-GPU
batch stays N = 1 245 184, the global batch is world * N.

The flat buffer is persistent: `GradBucket` points every parameter's `.grad` at a view of
one contiguous tensor, so the collective really is a single call on 543 MB (nvp_s) and no
per-step flatten/unflatten copy exists.  autograd accumulates in place into those views
(`zero_grad(set_to_none=False)` semantics are provided by `GradBucket.zero_()`).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def _force() -> bool:
    """NVP_DP_FORCE_COLLECTIVES=1: run the collective code paths even with ONE rank (a single-GPU box can then exercise
    the real RCCL calls - init, barrier, chunked asynchronous all-reduces on slices of the flat buffer, waits)."""
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
            backend = "nccl" if torch.cuda.is_available() else "gloo"
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
                 chunk_elems: int = 32 * 1024 * 1024):
        """`early`: parameters whose gradients are complete before the rest of backward has run (NVP's four
        grids: 99.9 % of the bytes).  If they occupy one contiguous range of the flat buffer, their all-reduce
        can be started early and asynchronously (`start_early`) and overlaps the remaining backward kernels."""
        self.params = list(params)
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, device=dev, dtype=torch.float32)
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
        the parameters of piece i while piece i+1 is still on the wire (`step_schedule`).  No-op without a group."""
        if self._early_range is None or self._early_work is not None:
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

    def step_schedule(self) -> Optional[list]:
        """For an optimizer that can update sub-ranges (nvp_amd.optim.AdamW.step(schedule=...)): reduce what is not
        in flight yet, and return [(wait, [(param, a, b), ...]), ...] in completion order - wait() makes the
        current stream wait for that piece's collective.  Gradients stay SUMS (the caller passes 1/world as
        grad_scale).  Returns None for a single process (nothing to wait for)."""
        if not _multi():
            return None
        if not self.consistent():
            self.all_reduce(scale=False)            # repair path: everything reduced synchronously
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
        """{param.data_ptr(): bucket view} for nvp_amd.functional.GRAD_SINK: backward then writes each
        gradient directly into the flat buffer (every element exactly once - no zero-fill needed)."""
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
        if not self.consistent():
            # a grad tensor was replaced (e.g. zero_grad(set_to_none=True)): copy back into the bucket.
            # An early all-reduce that already ran on stale bucket memory is joined and discarded: the
            # copy below restores the local gradients and the full all-reduce redoes the sum.
            self._join_early()
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
            self.attach()
        if _multi():
            if self._early_work is not None:
                # the big range is already in flight (overlapping the dW GEMMs): reduce the rest, then join
                lo, hi = self._early_range
                if lo > 0:
                    dist.all_reduce(self.flat[:lo], op=dist.ReduceOp.SUM)
                if hi < self.numel:
                    dist.all_reduce(self.flat[hi:], op=dist.ReduceOp.SUM)
                self._join_early()
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)       # one collective over the whole gradient
            if scale:
                self.flat.mul_(1.0 / dist.get_world_size())


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Make every rank start from rank `src`'s parameters (replicated-parameter DP)."""
    if _multi():
        for p in unique_parameters(module):
            dist.broadcast(p.data, src=src)
