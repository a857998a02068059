"""Helpers shared by the GPU parity tests."""
import torch

import nvp_oracle as O


def _load_state_into(model, sd):
    """copy oracle-keyed tensors into an nvp_amd NVP module"""
    with torch.no_grad():
        for k, v in sd.items():
            obj = model
            parts = k.split(".")
            for p in parts[:-1]:
                obj = obj[int(p)] if p.isdigit() else getattr(obj, p)
            getattr(obj, parts[-1]).copy_(v)


def _grad_of(model, key):
    obj = model
    parts = key.split(".")
    for p in parts[:-1]:
        obj = obj[int(p)] if p.isdigit() else getattr(obj, p)
    return getattr(obj, parts[-1]).grad


def _away_from_kinks(coords, sd, cfg, n, margin=1e-4):
    """Keep the first n candidate pixels whose three LeakyReLU inputs all satisfy |p| > margin.
    At p ~ 0 the slope jumps 0.01 -> 1, so a 1-ulp difference in p (MFMA vs MKL summation
    order) flips that unit's gradient; such pixels say nothing about kernel correctness."""
    with torch.no_grad():
        lat = O.nvp_latent(coords, sd, cfg)
        pre = O.modulator_preacts(lat, [sd[f"wrapper.modulator.layers.{k}.0.weight"] for k in range(3)],
                                  [sd[f"wrapper.modulator.layers.{k}.0.bias"] for k in range(3)])
        ok = torch.stack([p.abs().min(dim=1).values for p in pre]).min(dim=0).values > margin
    idx = torch.nonzero(ok).flatten()[:n]
    assert idx.numel() == n, "not enough candidates away from the LeakyReLU kinks"
    return idx


