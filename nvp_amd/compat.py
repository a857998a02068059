"""Expose nvp_amd's modules under the reference's own top-level import names so that
`experiment_scripts/train_video.py` / `eval.py` (which do `import modules`, and whose
`modules.py` does `import tinycudann, modulation` / `from sparsegrid import SparseGrid`)
pick up the MI355X implementation unmodified.  Call `install()` before those imports
(e.g. from a sitecustomize.py or a two-line launcher); see INTEGRATION.md section A."""
import sys

_STOCK_ADAMW = None


def install(optimizer: bool = False) -> None:
    """optimizer=True (opt-in): `torch.optim.AdamW(...)` as the reference calls it (training.py:13: lr, params, weight_decay)
    returns nvp_amd.optim.AdamW - the same update rule in ONE launch over all tensors (nvp_adamw_step) - whenever every parameter
    is an fp32 HIP tensor and no option outside that rule is requested; anything else gets the stock class.  On MI355X the stock
    default (multi-tensor `foreach`) needs 2.5 ms per step for NVP's 136 M parameters, nvp_adamw_step 0.65 ms."""
    from . import modulation, modules, sparsegrid, tinycudann
    sys.modules["modules"] = modules
    sys.modules["modulation"] = modulation
    sys.modules["sparsegrid"] = sparsegrid
    sys.modules["tinycudann"] = tinycudann
    if optimizer:
        install_optimizer()


def install_optimizer() -> None:
    import torch
    global _STOCK_ADAMW
    if _STOCK_ADAMW is not None:
        return
    stock = _STOCK_ADAMW = torch.optim.AdamW
    from .optim import AdamW as NvpAdamW

    class _Routing(type(stock)):
        # isinstance(opt, torch.optim.AdamW) stays true for whatever the routed call returned - the one-launch optimizer or a plain stock
        # instance (LR-scheduler wrappers, checkpoint loaders, Lightning)
        def __instancecheck__(cls, obj):
            return isinstance(obj, (NvpAdamW, stock)) or type.__instancecheck__(cls, obj)

    class AdamW(stock, metaclass=_Routing):
        """torch.optim.AdamW, routed to nvp_amd.optim.AdamW for fp32 HIP parameters (nvp_amd.compat.install_optimizer).  A SUBCLASS of the
        stock class, so subclassing it, isinstance / issubclass checks and __name__ lookups keep working - but never instantiated itself:
        the call returns either the one-launch optimizer or a PLAIN instance of the stock class (CPU parameters, param-group dicts,
        amsgrad, ...), which pickles and deep-copies like any stock optimizer.  `foreach` / `fused` are accepted and ignored on the
        routed path (the routed step is one launch over all tensors already)."""

        def __new__(cls, params=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **kw):
            if cls is AdamW and params is not None:                     # (subclasses of this class keep the stock behaviour)
                plist = list(params)
                plain = all(torch.is_tensor(p) for p in plist)          # (param groups given as dicts keep the stock class)
                ok = (plain and plist and all(p.is_cuda and p.dtype == torch.float32 for p in plist)
                      and not any(kw.get(k) for k in ("amsgrad", "maximize", "capturable", "differentiable")) and isinstance(lr, float))
                if ok:
                    return NvpAdamW(plist, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
                return stock(plist, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
            # neither return value above is an instance of cls, so type.__call__ does not run __init__ on it again
            return object.__new__(cls)                                  # subclasses; copyreg's argument-less reconstruction (pickle, deepcopy)

    AdamW.__name__ = AdamW.__qualname__ = "AdamW"
    AdamW.__module__ = stock.__module__
    torch.optim.AdamW = AdamW


def uninstall_optimizer() -> None:
    import torch
    global _STOCK_ADAMW
    if _STOCK_ADAMW is not None:
        torch.optim.AdamW = _STOCK_ADAMW
        _STOCK_ADAMW = None
