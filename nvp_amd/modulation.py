"""Modulated SIREN MLP (reference modulation.py:20-168): Sine, Siren, SirenNet, Modulator,
SirenWrapper with the reference's constructor signatures, attribute names, state_dict keys
and init distributions, drawn in the same order from torch's global RNG.

The arithmetic of `SirenWrapper.forward` (modulator + modulated SIREN, forward and
backward) is ONE fused set of MFMA kernels in libnvp_hip.so (nvp_mlp_*): that is the only
entry NVP uses (modules.py:81).

The stand-alone pieces - `Sine.forward`, `Siren.forward`, `SirenNet.forward(x, mods)`,
`Modulator.forward(z)` (modulation.py:24-25, 53-56, 83-92, 112-121) - are NOT on NVP's path
(nothing in the reference calls them except through SirenWrapper).  They are provided for
API completeness as plain device-side library ops (rocBLAS `F.linear`, element-wise ATen
kernels) on HIP tensors, differentiable through autograd; like everything else in this
package they refuse CPU tensors.  `tests/test_gpu_parity.py::test_standalone_modulation_*`
checks them, and the fused kernel's own modulator outputs (`modulator_streams`), against
the reference goldens' mod0..2.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import _lib as L
from .functional import ModulatedSiren


def exists(val):
    return val is not None


def _on_device(x: torch.Tensor, what: str) -> None:
    if not x.is_cuda:
        raise RuntimeError(f"{what}: nvp_amd runs on a HIP device only (got a CPU tensor); there is no CPU path")


class Sine(nn.Module):
    def __init__(self, w0=1.):
        super().__init__()
        self.w0 = w0

    def forward(self, x):
        _on_device(x, "Sine.forward")
        return torch.sin(x * self.w0)


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
        """act(x W^T + b) - stand-alone (un-fused) evaluation of one layer, library GEMM + ATen activation."""
        _on_device(x, "Siren.forward")
        return self.activation(nn.functional.linear(x, self.weight, self.bias))


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
        """Stand-alone evaluation with externally supplied modulation vectors (one per hidden layer, or None):
        each sine layer's output is multiplied by its `mod` before the next layer; Identity tail."""
        _on_device(x, "SirenNet.forward")
        if not isinstance(mods, tuple):
            mods = (mods,) * self.num_layers
        if len(mods) != self.num_layers:
            raise RuntimeError(f"expected {self.num_layers} modulation tensors, got {len(mods)}")
        for k, layer in enumerate(self.layers):
            x = layer(x)
            if exists(mods[k]):
                x = x * mods[k]            # the reference multiplies in place (`x *= mod`); same values
        return self.last_layer(x)


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
        """(h0, h1, h2): h0 = lrelu(W0 z + b0), h_k = lrelu(W_k [h_{k-1}; z] + b_k) - stand-alone evaluation
        (library GEMMs).  The fused kernels compute the same vectors inside nvp_mlp_fwd."""
        _on_device(z, "Modulator.forward")
        hiddens, x = [], z
        for layer in self.layers:
            h = layer(x)
            hiddens.append(h)
            x = torch.cat((h, z), dim=1)
        return tuple(hiddens)


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

    @torch.no_grad()
    def modulator_streams(self, coords, latent):
        """Debug / test hook: the activations the FUSED forward kernel saves for backward, as row-major [N,128]
        tensors {'h0','h1','h2' (Modulator.forward's outputs), 'q1','q2' (pre-sine SIREN activations)}."""
        from .functional import modulated_siren_streams
        return modulated_siren_streams(latent, coords, self.mlp_tensors())
