#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference, and pin the oracle against it.

Runs only in the build container (needs /root/reference).  It imports the reference's
sparsegrid.py, modulation.py and loss_functions.py (pure Python/torch, importable on
CPU), feeds them seeded inputs, asserts that oracle/nvp_oracle.py reproduces their
outputs bit-for-bit on CPU, and stores inputs + expected outputs as small fixtures.
Nothing of the reference (source or bytecode) is written to the repo - only data.

    python oracle/make_golden.py            # rewrites tests/golden/

Fixture inventory:
  sparse_{a,b}.npz      SparseGrid fwd + dE for loss=(out^2).sum(), edge-case coords
  sparse_inter_{a,b}    forward_inter on the same inputs (NaN rows at t==1 pinned)
  sparse_upsample.npz   upsample=True small case
  mlp_d{114,228}.npz    Modulator/SirenNet/SirenWrapper fwd, all param grads + dLatent
  e2e_minus_kf.npz      [stand-in keyframe features | SparseGrid] -> wrapper -> mse -> grads
  traj3.npz             3 AdamW + cosine steps of the e2e case (loss values)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

sys.path.insert(0, HERE)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import nvp_oracle as O  # noqa: E402
import sparsegrid as ref_sparsegrid  # noqa: E402  (reference)
import modulation as ref_modulation  # noqa: E402  (reference)
import loss_functions as ref_loss  # noqa: E402  (reference)


def bit_equal(a, b):
    a = a.detach().contiguous()
    b = b.detach().contiguous()
    if a.shape != b.shape:
        return False
    return bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())


def edge_coords(n, T, X, Y, gen):
    """Random coords in [0,1] seeded with exact 0, 1, .5-boundary and lattice points."""
    c = torch.rand((n, 3), generator=gen)
    specials = [0.0, 1.0, 0.5]
    k = 0
    for v in specials:
        for d in range(3):
            c[k, d] = v
            k += 1
    for d, res in enumerate((T, X, Y)):
        for i in range(res):
            if k >= n:
                break
            c[k, d] = i / (res - 1)               # lattice point
            k += 1
            if k < n:
                c[k, d] = (i + 0.5) / (res - 1)   # exact .5 rounding boundary (may exceed 1 -> clamp)
                c[k, d] = min(float(c[k, d]), 1.0)
                k += 1
    c[n - 1] = torch.tensor([1.0, 1.0, 1.0])
    c[n - 2] = torch.tensor([0.0, 0.0, 0.0])
    return c


def gen_sparse(tag, T, X, Y, Fd, n, seed):
    gen = torch.Generator().manual_seed(seed)
    m = ref_sparsegrid.SparseGrid(level_dim=Fd, x_resolution=X, y_resolution=Y, t_resolution=T, upsample=False)
    with torch.no_grad():
        m.embeddings.copy_(torch.randn(m.embeddings.shape, generator=gen))
    coords = edge_coords(n, T, X, Y, gen)
    out = m(coords)
    (out ** 2).sum().backward()
    dE = m.embeddings.grad.clone()

    emb = m.embeddings.detach().clone().requires_grad_(True)
    o2 = O.sparse_grid_forward(emb, coords)
    assert bit_equal(o2, out), f"oracle sparse fwd != reference ({tag})"
    (o2 ** 2).sum().backward()
    assert torch.allclose(emb.grad, dE, rtol=1e-5, atol=1e-5), f"oracle sparse bwd != reference ({tag})"

    with torch.no_grad():
        inter = m.forward_inter(coords)
        inter2 = O.sparse_grid_forward_inter(m.embeddings.detach(), coords)
    assert bit_equal(inter2, inter), f"oracle forward_inter != reference ({tag})"
    assert torch.isnan(inter).any(), "expected NaN rows at t == 1 (SURVEY R7)"

    np.savez_compressed(os.path.join(OUT, f"sparse_{tag}.npz"),
                        emb=m.embeddings.detach().numpy(), coords=coords.numpy(),
                        out=out.detach().numpy(), dE=dE.numpy(), out_inter=inter.numpy())


def gen_sparse_upsample(seed):
    gen = torch.Generator().manual_seed(seed)
    T, X, Y, Fd, n = 3, 6, 5, 2, 128
    m = ref_sparsegrid.SparseGrid(level_dim=Fd, x_resolution=X, y_resolution=Y, t_resolution=T, upsample=True)
    with torch.no_grad():
        m.embeddings.copy_(torch.randn(m.embeddings.shape, generator=gen))
    coords = edge_coords(n, T, 2 * X, 2 * Y, gen)
    with torch.no_grad():
        out = m(coords)
        o2 = O.sparse_grid_forward(m.embeddings.detach(), coords, upsample=True)
    assert bit_equal(o2, out), "oracle sparse upsample fwd != reference"
    np.savez_compressed(os.path.join(OUT, "sparse_upsample.npz"),
                        emb=m.embeddings.detach().numpy(), coords=coords.numpy(), out=out.numpy())


def build_ref_mlp(D, seed):
    torch.manual_seed(seed)
    net = ref_modulation.SirenNet(dim_in=1, dim_hidden=128, dim_out=3, num_layers=3, w0_initial=30.)
    wrapper = ref_modulation.SirenWrapper(net, latent_dim=D)
    return net, wrapper


def mlp_state(net, wrapper):
    sd = {}
    for k in range(3):
        sd[f"wrapper.modulator.layers.{k}.0.weight"] = wrapper.modulator.layers[k][0].weight
        sd[f"wrapper.modulator.layers.{k}.0.bias"] = wrapper.modulator.layers[k][0].bias
        sd[f"net.layers.{k}.weight"] = net.layers[k].weight
        sd[f"net.layers.{k}.bias"] = net.layers[k].bias
    sd["net.last_layer.weight"] = net.last_layer.weight
    sd["net.last_layer.bias"] = net.last_layer.bias
    return sd


def gen_mlp(D, n, seed):
    net, wrapper = build_ref_mlp(D, seed)
    gen = torch.Generator().manual_seed(seed + 1)
    # latent at a realistic-but-nontrivial scale so every path carries signal
    latent = (torch.randn((n, D), generator=gen) * 0.5).requires_grad_(True)
    T = 16
    ti = torch.randint(0, T, (n,), generator=gen)
    steps = torch.linspace(0.5 / T, 1 - 0.5 / T, T)[ti].reshape(n, 1)
    gt = torch.rand((1, n, 3), generator=gen) * 2 - 1

    mods = wrapper.modulator(latent)
    out = wrapper(coords=steps, latent=latent)
    loss = ref_loss.image_mse(None, {"model_out": out.reshape(1, n, 3)}, {"img": gt})["img_loss"]
    loss.backward()

    sd = mlp_state(net, wrapper)
    # oracle check (bit-identical forward, same autograd graph shape -> same grads)
    sd2 = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    lat2 = latent.detach().clone().requires_grad_(True)
    out2 = O.mlp_forward(lat2, steps, sd2)
    assert bit_equal(out2, out), f"oracle mlp fwd != reference (D={D})"
    mods2 = O.modulator_forward(lat2, [sd2[f"wrapper.modulator.layers.{k}.0.weight"] for k in range(3)],
                                [sd2[f"wrapper.modulator.layers.{k}.0.bias"] for k in range(3)])
    for a, b in zip(mods, mods2):
        assert bit_equal(a, b)
    loss2 = O.image_mse(out2.reshape(1, n, 3), gt)
    assert bit_equal(loss2, loss)
    loss2.backward()
    for k in sd:
        assert torch.allclose(sd2[k].grad, sd[k].grad, rtol=1e-5, atol=1e-9), f"grad {k}"
    assert torch.allclose(lat2.grad, latent.grad, rtol=1e-5, atol=1e-9)

    blob = {"latent": latent.detach().numpy(), "steps": steps.numpy(), "gt": gt.numpy(),
            "out": out.detach().numpy(), "loss": loss.detach().numpy(),
            "dlatent": latent.grad.numpy()}
    for i, m_ in enumerate(mods):
        blob[f"mod{i}"] = m_.detach().numpy()
    for k, v in sd.items():
        blob["p:" + k] = v.detach().numpy()
        blob["g:" + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, f"mlp_d{D}.npz"), **blob)


def gen_e2e(seed):
    """NVP minus keyframes: latent = [seeded stand-in for the 48F keyframe columns | SparseGrid out]
    -> wrapper -> image_mse -> grads, then a 3-step AdamW + cosine trajectory (reference
    training.py:13-14,73-76 ordering: zero_grad, backward, step, sched.step)."""
    Fd, T, X, Y, n = 2, 8, 9, 7, 256
    D = 57 * Fd
    gen = torch.Generator().manual_seed(seed)
    net, wrapper = build_ref_mlp(D, seed)
    grid = ref_sparsegrid.SparseGrid(level_dim=Fd, x_resolution=X, y_resolution=Y, t_resolution=T, upsample=False)
    with torch.no_grad():
        grid.embeddings.copy_(torch.randn(grid.embeddings.shape, generator=gen) * 0.3)
    coords = edge_coords(n, T, X, Y, gen)
    kf = (torch.randn((n, 48 * Fd), generator=gen) * 0.3).requires_grad_(True)
    ti = torch.randint(0, T, (n,), generator=gen)
    steps = torch.linspace(0.5 / T, 1 - 0.5 / T, T)[ti].reshape(n, 1)
    gt_u8 = torch.randint(0, 256, (1, n, 3), generator=gen, dtype=torch.uint8)
    gt = (gt_u8.float() - 127.5) / 127.5

    params = [grid.embeddings] + list(wrapper.parameters())   # wrapper.net == net (shared)
    init = {"sparse_grid.embeddings": grid.embeddings.detach().clone().numpy()}
    sd = mlp_state(net, wrapper)
    for k, v in sd.items():
        init["p:" + k] = v.detach().clone().numpy()

    optim = torch.optim.AdamW(lr=1e-2, params=params, weight_decay=0.001)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=3, eta_min=1e-5)
    losses, first = [], None
    for it in range(3):
        latent = torch.cat((kf, grid(coords)), dim=1)
        out = wrapper(coords=steps, latent=latent).reshape(1, n, 3)
        loss = ref_loss.image_mse(None, {"model_out": out}, {"img": gt})["img_loss"]
        optim.zero_grad()
        if kf.grad is not None:
            kf.grad = None
        loss.backward()
        if it == 0:
            first = {"out": out.detach().numpy().copy(), "dkf": kf.grad.numpy().copy(),
                     "g:sparse_grid.embeddings": grid.embeddings.grad.numpy().copy()}
            for k, v in sd.items():
                first["g:" + k] = v.grad.numpy().copy()
        optim.step()
        sched.step()
        losses.append(float(loss))

    # oracle replay of step 0 (bit-identical forward)
    sd0 = {k[2:]: torch.from_numpy(v) for k, v in init.items() if k.startswith("p:")}
    emb0 = torch.from_numpy(init["sparse_grid.embeddings"])
    lat0 = torch.cat((kf.detach(), O.sparse_grid_forward(emb0, coords)), dim=1)
    out0 = O.mlp_forward(lat0, steps, sd0).reshape(1, n, 3)
    assert bit_equal(out0, torch.from_numpy(first["out"])), "oracle e2e fwd != reference"

    blob = dict(init)
    blob.update(first)
    blob.update({"coords": coords.numpy(), "kf": kf.detach().numpy(), "steps": steps.numpy(),
                 "gt_u8": gt_u8.numpy(), "dims": np.array([Fd, T, X, Y, n])})
    np.savez_compressed(os.path.join(OUT, "e2e_minus_kf.npz"), **blob)
    np.savez_compressed(os.path.join(OUT, "traj3.npz"), losses=np.array(losses, dtype=np.float64))


def gen_init(seed):
    """Init-stream parity: the reference's constructors consume torch's global RNG in a fixed
    order (SparseGrid, SirenNet layers weight->bias, last layer, Modulator Linear defaults,
    then kaiming_normal_).  Store a fingerprint of every tensor for a given seed."""
    torch.manual_seed(seed)
    grid = ref_sparsegrid.SparseGrid(level_dim=2, x_resolution=9, y_resolution=7, t_resolution=8, upsample=False)
    net = ref_modulation.SirenNet(dim_in=1, dim_hidden=128, dim_out=3, num_layers=3, w0_initial=30.)
    wrapper = ref_modulation.SirenWrapper(net, latent_dim=114)
    sd = mlp_state(net, wrapper)
    sd["sparse_grid.embeddings"] = grid.embeddings
    blob = {}
    for k, v in sd.items():
        f = v.detach().flatten()
        blob["head:" + k] = f[:8].numpy().copy()
        blob["sum:" + k] = np.array(f.double().sum().item())
        blob["absmax:" + k] = np.array(f.abs().max().item())
    np.savez_compressed(os.path.join(OUT, f"init_seed{seed}.npz"), **blob)


def gen_quant(seed):
    """eval.py's quantize_keyframes / quantize_sparse_grid are plain functions at the top of a script that
    cannot be imported (argparse + missing packages at module level).  Their two `def`s are extracted from
    the reference file with `ast` and executed here, in the build container only, to capture golden
    outputs; `.cuda()` is neutralised because this container has no GPU."""
    import ast
    import math
    from torch import nn
    src = open(os.path.join(REF, "experiment_scripts", "eval.py")).read()
    tree = ast.parse(src)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("quantize_keyframes", "quantize_sparse_grid")]
    assert len(fns) == 2
    env = {"torch": torch, "nn": nn, "math": math, "unit_multiplier": 2.0 ** 8 - 1.0}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "eval.py", "exec"), env)
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        gen = torch.Generator().manual_seed(seed)
        cfg = {"n_levels": 5, "n_features_per_level": 2, "per_level_scale": 1.35, "base_resolution": 16}
        n = O.dense_grid_n_params(cfg)
        kf = torch.randn(n, generator=gen) * 0.1
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            kq = env["quantize_keyframes"](kf.clone(), cfg).detach()
            sg = torch.randn((5, 6, 7, 2), generator=gen) * 0.05
            sq = env["quantize_sparse_grid"](sg.clone(), {"n_features_per_level": 2}).detach()
    finally:
        torch.Tensor.cuda = orig
    np.savez_compressed(os.path.join(OUT, "quant.npz"), kf=kf.numpy(), kf_q=kq.numpy(), sg=sg.numpy(), sg_q=sq.numpy(),
                        n_levels=np.array(5))


def gen_export(seed):
    """compression.py's compress_keyframes / compress_sparse_grid write u8 images with cv2.imwrite.
    Their two `def`s are extracted with `ast` (the script itself cannot be imported: argparse, cv2, configargparse)
    and executed here with a stand-in `cv2.imwrite` that records (path, array) instead of encoding a PNG; the
    decode side of eval_compression.py (img / 255 * (max - min) + min, per dim and level) is applied to the recorded
    images to capture the round-trip values."""
    import ast
    import math
    import types
    src = open(os.path.join(REF, "experiment_scripts", "compression.py")).read()
    tree = ast.parse(src)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("compress_keyframes", "compress_sparse_grid")]
    assert len(fns) == 2
    written = []
    cv2 = types.SimpleNamespace(imwrite=lambda path, arr: written.append((path, np.array(arr))))
    fake_os = types.SimpleNamespace(path=os.path, makedirs=lambda *a, **k: None)
    env = {"torch": torch, "np": np, "math": math, "cv2": cv2, "os": fake_os, "unit_multiplier": 2.0 ** 8 - 1.0}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "compression.py", "exec"), env)
    gen = torch.Generator().manual_seed(seed)
    cfg = {"n_levels": 5, "n_features_per_level": 2, "per_level_scale": 1.35, "base_resolution": 16}
    n = O.dense_grid_n_params(cfg)
    kf = torch.randn(n, generator=gen) * 0.1
    sg = torch.randn((5, 6, 7, 2), generator=gen) * 0.05
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env["compress_keyframes"](kf.clone(), cfg, "kf")
        n_kf = len(written)
        env["compress_sparse_grid"](sg.clone(), {"n_features_per_level": 2}, "sg")
    blob = {"kf": kf.numpy(), "sg": sg.numpy(), "n_levels": np.array(5)}
    for path, arr in written[:n_kf]:                       # kf/dim{d}/{level:02d}.png
        d, lvl = int(path.split("dim")[1].split("/")[0]), int(os.path.basename(path)[:2])
        blob[f"kf_d{d}_l{lvl}"] = arr
    for path, arr in written[n_kf:]:                       # sg/dim{d}/{frame:05d}.png
        d, fr = int(path.split("dim")[1].split("/")[0]), int(os.path.basename(path)[:5])
        blob[f"sg_d{d}_f{fr}"] = arr
    np.savez_compressed(os.path.join(OUT, "export.npz"), **blob)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)          # deterministic reductions while generating
    gen_sparse("a", 4, 7, 5, 2, 512, seed=11)
    gen_sparse("b", 16, 12, 10, 4, 512, seed=12)
    gen_sparse_upsample(seed=13)
    gen_mlp(114, 256, seed=21)
    gen_mlp(228, 256, seed=22)
    gen_e2e(seed=31)
    gen_init(seed=123)
    gen_quant(seed=41)
    gen_export(seed=51)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"golden fixtures written to {OUT} ({tot / 1024:.0f} KiB); oracle == reference on all cases")


if __name__ == "__main__":
    main()
