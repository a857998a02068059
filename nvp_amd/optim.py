"""AdamW on the HIP path: the reference's `torch.optim.AdamW(lr=lr, params=model.parameters(),
weight_decay=0.001)` (training.py:13) with the whole step - every parameter tensor - in ONE kernel launch
(`nvp_adamw_step`, nvp_amd/csrc/optim.hip) instead of torch's multi-tensor chunks.

It is a regular `torch.optim.Optimizer`: `param_groups[i]['lr']` is what `CosineAnnealingLR` drives
(training.py:14), `state[p]` holds `step`, `exp_avg`, `exp_avg_sq` under torch's names, `zero_grad()` works.
Same update rule as torch.optim.AdamW (decoupled weight decay, bias-corrected, no amsgrad/maximize).
There is no CPU path: parameters must live on a HIP device.

Early / fused updates (harness.train_step on one GPU: `early_update`, `fused_peek` / `fused_commit`) step the grids DURING backward:
the sparse grid (NVP_FUSED_SPARSE_ADAMW, default on) AND the three keyframe planes (NVP_FUSED_DENSE_ADAMW, default on) are updated
inside the scatter's flushes - after backward their `.grad` is None and their parameters are already stepped (a grad-norm log or a
clipping hook on AFTER_BACKWARD sees no gradient for them); the remaining grids, if any, follow on a side stream (NVP_EARLY_ADAMW).
That is only equivalent to the reference's loop when every backward is followed by exactly one `step()` on unmodified gradients:
loops that clip or accumulate gradients, inspect `.grad` of the grids, skip steps or recover from exceptions must run with
NVP_FUSED_ADAMW=0 (one switch for all of the above; = NVP_EARLY_ADAMW=0 NVP_FUSED_SPARSE_ADAMW=0 NVP_FUSED_DENSE_ADAMW=0) - then this
class is a plain one-launch AdamW.  An iteration whose step() never ran is reported by `begin_step()` (RuntimeWarning,
`unfinished_iterations`).
"""
from __future__ import annotations

from typing import Iterable

import torch

from . import _lib


class AdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._early_done = set()        # id(param) of the parameters early_update() has already stepped in the current iteration
        self._side = None               # side stream of early_update()
        self._by_ptr = None

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def begin_step(self) -> None:
        """Start of an iteration: forget the early updates of an earlier iteration whose step() never ran (an exception between
        backward and step would otherwise make the next iteration skip those parameters' early update silently)."""
        if self._early_done:
            # The previous iteration applied early / fused updates (their parameters and moments ARE stepped) but its step() never
            # ran - backward raised half-way, or the caller skipped step().  Nothing can be rolled back; say so instead of letting the
            # step counts of those parameters drift ahead of the others silently.
            import warnings
            self.unfinished_iterations = getattr(self, "unfinished_iterations", 0) + 1
            warnings.warn(f"nvp_amd.optim.AdamW: {len(self._early_done)} parameter tensor(s) were updated early in an iteration whose step() never "
                          "ran (the sparse grid and the keyframe planes are stepped inside the scatter's flushes); they are one optimizer step ahead of the "
                          "rest (set NVP_FUSED_ADAMW=0 for loops that skip or recover from failed steps, clip or accumulate gradients or read the grids' .grad)", RuntimeWarning, stacklevel=2)
        self._early_done.clear()

    @torch.no_grad()
    def early_update(self, pairs, grad_scale: float = 1.0) -> None:
        """AdamW step of some parameters NOW, before backward has finished: `pairs` = [(parameter tensor, gradient tensor)] whose
        gradients are complete on the current stream (functional.EARLY_GRADS_HOOK: NVP's grids, 99.7 % of the parameters, are
        done after the scatter while the dW GEMMs - which read no grid parameter - still have 1.7 ms to run).  The update runs on a
        side stream; step() skips what was updated here and joins the stream.  Same kernel, same scalars, same result as step()."""
        # parameter lookup by storage address; rebuilt whenever an address went stale (p.data re-homed into a flat buffer,
        # `enc.params = ...` rebinds are new Parameter objects and never reach this optimizer)
        if self._by_ptr is None or any(p.data_ptr() != k for k, (p, _) in self._by_ptr.items()):
            self._by_ptr = {p.data_ptr(): (p, g) for g in self.param_groups for p in g["params"]}
        lib = _lib.load()
        dev = pairs[0][0].device
        if self._side is None:
            self._side = _lib.side_stream(dev)
        ready = torch.cuda.Event()
        ready.record()                                        # the gradients are complete at this point of the compute stream
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            for t, g in pairs:
                ent = self._by_ptr.get(t.data_ptr())
                if ent is None or id(ent[0]) in self._early_done:
                    continue
                p, group = ent
                if p.grad is not None:
                    continue            # autograd will ACCUMULATE into the existing .grad: `g` alone is not the gradient; step() handles it
                g.record_stream(self._side)      # read here on the side stream; if autograd clones instead of adopting it, it is freed on the compute stream
                st = self._state_of(p)
                st["step"] += 1
                for x in (p, g, st["exp_avg"], st["exp_avg_sq"]):
                    _lib.ptr(x)
                if g.numel() != p.numel():
                    raise RuntimeError("early_update: gradient and parameter sizes differ")
                seg = (_lib.AdamwSeg * 1)(_lib.AdamwSeg(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()))
                b1, b2 = group["betas"]
                _lib.check(lib.nvp_adamw_step(seg, 1, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                              float(group["weight_decay"]), int(st["step"]), float(grad_scale), _lib.stream_ptr()),
                           "nvp_adamw_step")
                self._early_done.add(id(p))

    @torch.no_grad()
    def fused_peek(self, t: torch.Tensor):
        """For a kernel that applies this optimizer's step to the parameter with storage `t` ITSELF (the scatter's flush:
        nvp_encode_bwd_sparse_adamw): the state tensors and the scalars of the step that is due, WITHOUT advancing anything -
        fused_commit() does that once the kernel has been enqueued.  None when the parameter is unknown to this optimizer, was
        already stepped in this iteration, or already has a .grad (autograd would accumulate: the kernel's gradient alone is not
        the gradient)."""
        if self._by_ptr is None or any(p.data_ptr() != k for k, (p, _) in self._by_ptr.items()):
            self._by_ptr = {p.data_ptr(): (p, g) for g in self.param_groups for p in g["params"]}
        ent = self._by_ptr.get(t.data_ptr())
        if ent is None or id(ent[0]) in self._early_done or ent[0].grad is not None:
            return None
        p, group = ent
        st = self._state_of(p)
        for x in (p, st["exp_avg"], st["exp_avg_sq"]):
            _lib.ptr(x)
        b1, b2 = group["betas"]
        return {"param": p, "exp_avg": st["exp_avg"], "exp_avg_sq": st["exp_avg_sq"], "lr": float(group["lr"]), "beta1": float(b1),
                "beta2": float(b2), "eps": float(group["eps"]), "weight_decay": float(group["weight_decay"]), "step": int(st["step"]) + 1}

    def fused_commit(self, t: torch.Tensor) -> None:
        """The kernel that took fused_peek()'s state has been enqueued on the compute stream: count the step, skip the parameter in step()."""
        p, _ = self._by_ptr[t.data_ptr()]
        self._state_of(p)["step"] += 1
        self._early_done.add(id(p))

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, schedule=None):
        """One AdamW update.  `grad_scale` multiplies every gradient inside the kernel (data parallel: 1/world
        after a SUM all-reduce of the flat gradient, which saves a separate scaling pass over 543 MB).
        `schedule` (parallel.GradBucket.step_schedule()): [(wait, [(param, a, b), ...]), ...] - for each entry
        wait() is called (the current stream then waits for that piece of the gradient all-reduce) and elements
        [a, b) of those parameters are updated, so the update of one piece overlaps the collective of the next.
        Parameters (or parts) the schedule does not mention are updated last."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        info = {}                                   # id(param) -> (param, group, step, covered ranges)
        early, self._early_done = self._early_done, set()
        if early and self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)      # the early updates are part of this step
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None or id(p) in early:          # early_update() already stepped it in this iteration
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("nvp_amd.optim.AdamW does not support sparse gradients")
                st = self._state_of(p)
                st["step"] += 1
                for t in (p, p.grad, st["exp_avg"], st["exp_avg_sq"]):
                    _lib.ptr(t)                     # loud errors for CPU / non-contiguous tensors
                info[id(p)] = (p, group, int(st["step"]), [])

        def launch(items):
            """items: [(param, a, b)] -> one kernel launch per (group, step) combination"""
            buckets = {}
            for p, a, b in items:
                ent = info.get(id(p))
                if ent is None or a >= b:
                    continue
                _, group, step, cov = ent
                cov.append((a, b))
                st = self.state[p]
                seg = _lib.AdamwSeg(p.data_ptr() + 4 * a, p.grad.data_ptr() + 4 * a, st["exp_avg"].data_ptr() + 4 * a,
                                    st["exp_avg_sq"].data_ptr() + 4 * a, b - a)
                buckets.setdefault((id(group), step), (group, step, []))[2].append(seg)
            for group, step, segs in buckets.values():
                b1, b2 = group["betas"]
                arr = (_lib.AdamwSeg * len(segs))(*segs)
                _lib.check(lib.nvp_adamw_step(arr, len(segs), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                              float(group["weight_decay"]), step, float(grad_scale), _lib.stream_ptr()),
                           "nvp_adamw_step")

        for wait, items in (schedule or []):
            wait()
            launch(items)
        # whatever the schedule did not cover (everything, without a schedule)
        rest = []
        for p, group, step, cov in info.values():
            pos = 0
            for a, b in sorted(cov):
                if a > pos:
                    rest.append((p, pos, a))
                pos = max(pos, b)
            if pos < p.numel():
                rest.append((p, pos, p.numel()))
        launch(rest)
        return loss
