"""Modulated SIREN MLP (reference modulation.py:20-168): Sine, Siren, SirenNet, Modulator,
SirenWrapper with the reference's constructor signatures, attribute names, state_dict keys
and init distributions, drawn in the same order from torch's global RNG.

The arithmetic of `SirenWrapper.forward` (modulator + modulated SIREN, forward and
backward) is ONE fused set of fp32-MFMA kernels in libnvp_hip.so (nvp_mlp_*): that is the
only entry NVP uses (modules.py:81).  `Modulator.forward` on its own returns the three
modulation vectors from the same kernel (inference only); the other stand-alone forwards
(`Siren`, `SirenNet` with externally supplied mods) are not part of NVP's path and raise.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import _lib as L
from .functional import ModulatedSiren


def exists(val):
    return val is not None


class Sine(nn.Module):
    def __init__(self, w0=1.):
        super().__init__()
        self.w0 = w0

    def forward(self, x):
        raise NotImplementedError("Sine is fused into the nvp_mlp_* kernels; call SirenWrapper.forward")


class Siren(nn.Module):
    """One SIREN layer: parameters + init only (reference modulation.py:30-56)."""

    def __init__(self, dim_in, dim_out, w0=1., c=6., is_first=False, use_bias=True, activation=None):
        super().__init__()
        if not use_bias:
            raise NotImplementedError("use_bias=False is outside NVP's path")
        self.dim_in = dim_in
        self.is_first = is_first
        self.w0 = w0
        weight = torch.zeros(dim_out, dim_in)
        bias = torch.zeros(dim_out)
        # U(-1/dim_in, 1/dim_in) for the first layer, U(+-sqrt(c/dim_in)/w0) otherwise; weight then bias
        w_std = (1 / dim_in) if is_first else (math.sqrt(c / dim_in) / w0)
        weight.uniform_(-w_std, w_std)
        bias.uniform_(-w_std, w_std)
        self.weight = nn.Parameter(weight)
        self.bias = nn.Parameter(bias)
        self.activation = Sine(w0) if activation is None else activation

    def forward(self, x):
        raise NotImplementedError("Siren layers are fused into the nvp_mlp_* kernels; call SirenWrapper.forward")


class SirenNet(nn.Module):
    def __init__(self, dim_in, dim_hidden, dim_out, num_layers, w0=1., w0_initial=30., use_bias=True, final_activation=None):
        super().__init__()
        self.num_layers = num_layers
        self.dim_hidden = dim_hidden
        self.dim_in = dim_in
        self.dim_out = dim_out
        self.w0 = w0
        self.w0_initial = w0_initial
        self.layers = nn.ModuleList([])
        for ind in range(num_layers):
            first = ind == 0
            self.layers.append(Siren(dim_in=dim_in if first else dim_hidden, dim_out=dim_hidden,
                                     w0=w0_initial if first else w0, use_bias=use_bias, is_first=first))
        if exists(final_activation) and not isinstance(final_activation, nn.Identity):
            raise NotImplementedError("only the Identity tail used by NVP is implemented")
        self.last_layer = Siren(dim_in=dim_hidden, dim_out=dim_out, w0=w0, use_bias=use_bias, activation=nn.Identity())

    def kernel_compatible(self) -> bool:
        return (self.dim_in == 1 and self.dim_hidden == L.HIDDEN and self.dim_out == 3 and self.num_layers == 3
                and self.w0 == 1. and self.w0_initial == 30.)

    def tensors(self):
        out = []
        for layer in self.layers:
            out += [layer.weight, layer.bias]
        return out + [self.last_layer.weight, self.last_layer.bias]

    def forward(self, x, mods=None):
        raise NotImplementedError("SirenNet with externally supplied mods is not on NVP's path; "
                                  "call SirenWrapper.forward(coords, latent)")


def init_weights_normal(m):
    if type(m) == nn.Linear:
        if hasattr(m, 'weight'):
            nn.init.kaiming_normal_(m.weight, a=0.0, nonlinearity='relu', mode='fan_in')


class Modulator(nn.Module):
    def __init__(self, dim_in, dim_hidden, num_layers):
        super().__init__()
        self.dim_in = dim_in
        self.dim_hidden = dim_hidden
        self.num_layers = num_layers
        self.layers = nn.ModuleList([])
        for ind in range(num_layers):
            dim = dim_in if ind == 0 else (dim_hidden + dim_in)
            self.layers.append(nn.Sequential(nn.Linear(dim, dim_hidden), nn.LeakyReLU()))
        self.weight_init = init_weights_normal
        self.layers.apply(self.weight_init)

    def tensors(self):
        out = []
        for layer in self.layers:
            out += [layer[0].weight, layer[0].bias]
        return out

    def forward(self, z):
        raise NotImplementedError("the Modulator is fused with the SIREN in the nvp_mlp_* kernels; "
                                  "call SirenWrapper.forward(coords, latent)")


class SirenWrapper(nn.Module):
    def __init__(self, net, latent_dim=None):
        super().__init__()
        self.net = net
        self.modulator = None
        if exists(latent_dim):
            self.modulator = Modulator(dim_in=latent_dim, dim_hidden=net.dim_hidden, num_layers=net.num_layers)
        self.latent_dim = latent_dim

    def mlp_tensors(self):
        """The 14 tensors in the canonical kernel order (modulator, SIREN, last layer)."""
        if self.modulator is None or not self.net.kernel_compatible() or self.modulator.dim_hidden != L.HIDDEN:
            raise NotImplementedError("the HIP kernels are specialised for NVP's network: dim_in=1, "
                                      "128 hidden units, 3 layers, w0=(30,1,1), RGB out, with a modulator")
        return self.modulator.tensors() + self.net.tensors()

    def forward(self, coords, latent=None):
        modulate = exists(self.modulator)
        assert not (modulate ^ exists(latent)), 'latent vector must be only supplied if `latent_dim` was passed in on instantiation'
        return ModulatedSiren.apply(latent, coords, torch.is_grad_enabled(), *self.mlp_tensors())
