"""CPU tests: the oracle against the golden vectors captured from the real reference
(oracle/make_golden.py), and the host-side restatements the kernels rely on."""
import math
import os

import numpy as np
import pytest
import torch

import nvp_oracle as O
from conftest import GOLDEN, small_cfg


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def _eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("tag", ["a", "b"])
def test_sparse_forward_and_grad_match_reference(tag):
    g = _load(f"sparse_{tag}.npz")
    emb = torch.from_numpy(g["emb"]).requires_grad_(True)
    coords = torch.from_numpy(g["coords"])
    out = O.sparse_grid_forward(emb, coords)
    assert _eq(out.detach().numpy(), g["out"])            # pure copy: bit-exact
    (out ** 2).sum().backward()
    np.testing.assert_allclose(emb.grad.numpy(), g["dE"], rtol=1e-5, atol=1e-5)
    inter = O.sparse_grid_forward_inter(emb.detach(), coords)
    assert _eq(inter.numpy(), g["out_inter"])
    assert np.isnan(g["out_inter"]).any()                  # the t == 1 quirk is pinned


def test_sparse_upsample_matches_reference():
    g = _load("sparse_upsample.npz")
    out = O.sparse_grid_forward(torch.from_numpy(g["emb"]), torch.from_numpy(g["coords"]), upsample=True)
    assert _eq(out.numpy(), g["out"])


@pytest.mark.parametrize("D", [114, 228])
def test_mlp_matches_reference(D):
    g = _load(f"mlp_d{D}.npz")
    sd = {k[2:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith("p:")}
    lat = torch.from_numpy(g["latent"]).requires_grad_(True)
    steps = torch.from_numpy(g["steps"])
    out = O.mlp_forward(lat, steps, sd)
    assert _eq(out.detach().numpy(), g["out"])
    mods = O.modulator_forward(lat, [sd[f"wrapper.modulator.layers.{k}.0.weight"] for k in range(3)],
                               [sd[f"wrapper.modulator.layers.{k}.0.bias"] for k in range(3)])
    for i, m in enumerate(mods):
        assert _eq(m.detach().numpy(), g[f"mod{i}"])
    loss = O.image_mse(out.reshape(1, -1, 3), torch.from_numpy(g["gt"]))
    assert _eq(loss.detach().numpy(), g["loss"])
    loss.backward()
    for k, v in sd.items():
        np.testing.assert_allclose(v.grad.numpy(), g["g:" + k], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(lat.grad.numpy(), g["dlatent"], rtol=1e-5, atol=1e-9)


def test_e2e_minus_keyframes_and_trajectory():
    g = _load("e2e_minus_kf.npz")
    Fd, T, X, Y, n = (int(v) for v in g["dims"])
    emb = torch.from_numpy(g["sparse_grid.embeddings"]).clone().requires_grad_(True)
    sd = {k[2:]: torch.from_numpy(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith("p:")}
    coords, kf, steps = (torch.from_numpy(g[k]) for k in ("coords", "kf", "steps"))
    gt = O.normalise_gt(torch.from_numpy(g["gt_u8"]))
    params = [emb] + [sd[k] for k in O.STATE_KEYS_MLP]
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.001)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=3, eta_min=1e-5)
    losses = []
    for it in range(3):
        lat = torch.cat((kf, O.sparse_grid_forward(emb, coords)), dim=1)
        out = O.mlp_forward(lat, steps, sd).reshape(1, n, 3)
        loss = O.image_mse(out, gt)
        opt.zero_grad()
        loss.backward()
        if it == 0:
            assert _eq(out.detach().numpy(), g["out"])
            np.testing.assert_allclose(emb.grad.numpy(), g["g:sparse_grid.embeddings"], rtol=1e-5, atol=1e-9)
        opt.step()
        sched.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, _load("traj3.npz")["losses"], rtol=1e-6)


def test_dense_grid_geometry_matches_reference_arithmetic():
    cfg = small_cfg()["2d_encoding_xy"]
    scales, ress, offs = O.dense_grid_levels(cfg)
    # SURVEY section 2 [probe] resolutions and the 4 616 112-cell total that reproduces the README's 0.901 bpp
    assert ress == [16, 22, 30, 40, 54, 72, 97, 131, 177, 239, 322, 435, 587, 792, 1069, 1443]
    assert offs[-1] == 4616112
    assert O.dense_grid_n_params(cfg) == 9232224
    assert all(abs(s - (16 * 1.35 ** l - 1)) < 1e-3 for l, s in enumerate(scales))


def test_dense_grid_interpolates_and_partitions_unity():
    cfg = {"n_levels": 4, "n_features_per_level": 2, "base_resolution": 16, "per_level_scale": 1.35}
    n = O.dense_grid_n_params(cfg)
    x = torch.rand(257, 2)
    ones = O.dense_grid_2d(torch.ones(n), x, cfg)
    np.testing.assert_allclose(ones.numpy(), 1.0, atol=2e-7)      # bilinear weights sum to 1
    # a cell-aligned input reproduces that cell's value exactly: x = (i - 0.5)/scale  ->  pos = i
    scales, ress, offs = O.dense_grid_levels(cfg)
    P = torch.randn(n)
    xi = torch.tensor([[(3 - 0.5) / scales[0], (5 - 0.5) / scales[0]]])
    out = O.dense_grid_2d(P, xi, cfg)
    cell = 3 + 5 * ress[0]
    np.testing.assert_allclose(out[0, :2].numpy(), P.reshape(-1, 2)[cell].numpy(), rtol=1e-5, atol=1e-6)


def test_sampler_restates_reference_order_and_ranges():
    gen = torch.Generator().manual_seed(0)
    ti, pi, coords, steps = O.sample_batch(16, 64, 64, 1000, gen)
    gen2 = torch.Generator().manual_seed(0)
    assert torch.equal(ti, torch.randint(0, 16, (1000,), generator=gen2))        # temporal first
    assert torch.equal(pi, torch.randint(0, 64 * 64, (1000,), generator=gen2))   # then spatial
    assert float(coords.min()) >= 0 and float(coords.max()) <= 1
    mg = O.get_mgrid_2d(64, 64)
    assert torch.equal(coords[:, 1:], mg[pi])
    assert math.isclose(float(steps.min()), 0.5 / 16, rel_tol=0, abs_tol=1e-6) or float(steps.min()) > 0.5 / 16


def test_branch_free_sincos_constants_are_accurate():
    """Host emulation of nvp_sincos (nvp_amd/csrc/nvp_common.h) with the same constants."""
    f = np.float32

    def fma(a, b, c):
        return f(np.float64(a) * np.float64(b) + np.float64(c))

    x = ((np.random.default_rng(0).random(400000) * 2 - 1) * 1000).astype(f)
    n = np.rint(x * f(0.636619747)).astype(f)
    r = fma(n, f(-1.57079637e+00), x)
    r = fma(n, f(4.37113883e-08), r)
    r = fma(n, f(1.71512451e-15), r)
    q = n.astype(np.int64)
    r2 = (r * r).astype(f)
    ps = fma(r2, f(-1.9515295891e-4), f(8.3321608736e-3)); ps = fma(ps, r2, f(-1.6666654611e-1)); ps = fma((ps * r2).astype(f), r, r)
    pc = fma(r2, f(2.443315711809948e-5), f(-1.388731625493765e-3)); pc = fma(pc, r2, f(4.166664568298827e-2))
    pc = fma((pc * r2).astype(f), r2, fma(r2, f(-0.5), f(1.0)))
    s0 = np.where(q & 1, pc, ps); c0 = np.where(q & 1, ps, pc)
    sn = np.where(q & 2, -s0, s0); cs = np.where((q + 1) & 2, -c0, c0)
    assert np.abs(sn - np.sin(x.astype(np.float64))).max() < 2e-7
    assert np.abs(cs - np.cos(x.astype(np.float64))).max() < 2e-7


def test_sin_only_polynomial_is_accurate():
    """Host emulation of nvp_sin (forward pass / dW staging): reduction by pi + one degree-9 odd polynomial."""
    f = np.float32

    def fma(a, b, c):
        return f(np.float64(a) * np.float64(b) + np.float64(c))

    x = ((np.random.default_rng(1).random(400000) * 2 - 1) * 1000).astype(f)
    n = np.rint(x * f(0.318309886)).astype(f)
    r = fma(n, f(-3.14159274), x)
    r = fma(n, f(8.74227766e-08), r)
    r2 = (r * r).astype(f)
    p = fma(r2, f(2.6000545605e-06), f(-1.9806615092e-04)); p = fma(p, r2, f(8.3330172897e-03)); p = fma(p, r2, f(-1.6666657096e-01))
    sn = fma((p * r2).astype(f), r, r)
    sn = np.where(n.astype(np.int64) & 1, -sn, sn)
    assert np.abs(sn - np.sin(x.astype(np.float64))).max() < 2e-7


def test_sincos_pi_reduction_is_accurate():
    """Host emulation of the default nvp_sincos (NVP_SINCOS_PI): one reduction by pi, degree-9 sine and degree-10 cosine."""
    f = np.float32

    def fma(a, b, c):
        return f(np.float64(a) * np.float64(b) + np.float64(c))

    x = ((np.random.default_rng(2).random(400000) * 2 - 1) * 1000).astype(f)
    n = np.rint(x * f(0.318309886)).astype(f)
    r = fma(n, f(-3.14159274), x)
    r = fma(n, f(8.74227766e-08), r)
    r2 = (r * r).astype(f)
    p = fma(r2, f(2.6000545605e-06), f(-1.9806615092e-04)); p = fma(p, r2, f(8.3330172897e-03)); p = fma(p, r2, f(-1.6666657096e-01))
    sn = fma((p * r2).astype(f), r, r)
    c = fma(r2, f(-2.6077104766e-07), f(2.4761886211e-05)); c = fma(c, r2, f(-1.3888403507e-03)); c = fma(c, r2, f(4.1666640728e-02))
    c = fma(c, r2, f(-4.9999999550e-01)); cs = fma(c, r2, f(1.0))
    odd = n.astype(np.int64) & 1
    sn = np.where(odd, -sn, sn); cs = np.where(odd, -cs, cs)
    assert np.abs(sn - np.sin(x.astype(np.float64))).max() < 2e-7
    assert np.abs(cs - np.cos(x.astype(np.float64))).max() < 2e-7


def test_upsample2x_index_and_weight_rule_matches_aten():
    """The index / weight rule the HIP x2 pre-upsample and its gather adjoint are written from (nvp_amd/csrc/encode.hip: up_src, up_weight;
    sparsegrid.py:26-34 = F.interpolate(scale_factor=2, mode='bilinear')), restated on the host and checked against ATen: forward to
    1-ulp-level association differences, adjoint weights against autograd.  Odd sizes and a one-cell axis included."""
    import numpy as np
    f32 = np.float32

    def up_src(u, R):
        src = f32(max(f32(0.5) * (f32(u) + f32(0.5)) - f32(0.5), 0))
        i0 = int(src)
        l1 = f32(src - f32(i0))
        return i0, min(i0 + 1, R - 1), f32(1) - l1, l1

    def up_weight(u, c, R):
        if u < 0 or u >= 2 * R:
            return f32(0)
        i0, i1, l0, l1 = up_src(u, R)
        return (l0 if i0 == c else f32(0)) + (l1 if i1 == c else f32(0))

    g = torch.Generator().manual_seed(3)
    for (T, X, Y, Fd) in ((2, 5, 7, 2), (1, 1, 4, 4), (2, 6, 3, 1)):
        emb = torch.randn((T, X, Y, Fd), generator=g, requires_grad=True)
        ref = torch.nn.functional.interpolate(emb.permute(3, 0, 1, 2), scale_factor=2, mode='bilinear').permute(1, 2, 3, 0)
        e = emb.detach().numpy()
        out = np.zeros((T, 2 * X, 2 * Y, Fd), f32)
        for ux in range(2 * X):
            i0, i1, l0, l1 = up_src(ux, X)
            for uy in range(2 * Y):
                j0, j1, m0, m1 = up_src(uy, Y)
                out[:, ux, uy, :] = l0 * (m0 * e[:, i0, j0, :] + m1 * e[:, i0, j1, :]) + l1 * (m0 * e[:, i1, j0, :] + m1 * e[:, i1, j1, :])
        assert np.abs(out - ref.detach().numpy()).max() <= 1e-6 * max(1.0, float(ref.abs().max()))
        d = torch.randn(ref.shape, generator=g)
        (want,) = torch.autograd.grad(ref, emb, d)
        dn, got = d.numpy(), np.zeros((T, X, Y, Fd), f32)
        for x in range(X):
            for y in range(Y):
                for a in range(-1, 3):
                    wx = up_weight(2 * x + a, x, X)
                    for b in range(-1, 3):
                        wy = up_weight(2 * y + b, y, Y)
                        if wx != 0 and wy != 0:
                            got[:, x, y, :] += (wx * wy) * dn[:, 2 * x + a, 2 * y + b, :]
        assert np.abs(got - want.numpy()).max() <= 1e-5 * max(1.0, float(want.abs().max()))
