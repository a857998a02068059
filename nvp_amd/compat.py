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

    def AdamW(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **kw):
        from .optim import AdamW as NvpAdamW
        plist = list(params)
        plain = all(torch.is_tensor(p) for p in plist)              # (param groups given as dicts keep the stock class)
        ok = plain and plist and all(p.is_cuda and p.dtype == torch.float32 for p in plist) and not any(kw.get(k) for k in ("amsgrad", "maximize", "capturable", "differentiable"))
        if ok and isinstance(lr, float):
            return NvpAdamW(plist, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        return stock(plist, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)

    AdamW.__doc__ = "torch.optim.AdamW, routed to nvp_amd.optim.AdamW for fp32 HIP parameters (nvp_amd.compat.install_optimizer)"
    torch.optim.AdamW = AdamW


def uninstall_optimizer() -> None:
    import torch
    global _STOCK_ADAMW
    if _STOCK_ADAMW is not None:
        torch.optim.AdamW = _STOCK_ADAMW
        _STOCK_ADAMW = None
