"""CPU oracle for the NVP per-coordinate encoding path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32 PyTorch on the CPU, the arithmetic of the
reference hot path.  It is the checker for the
HIP kernels and the "port" CPU baseline in bench.py.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the
product package (nvp_amd/) never does.

Pinning status
--------------
* R4-R13 (SparseGrid, Modulator, SirenNet, image_mse): PINNED.  oracle/make_golden.py
  imports the reference's sparsegrid.py / modulation.py / loss_functions.py in the
  build container, checks this restatement against them bit-for-bit on CPU and
  writes the golden vectors in tests/golden/.
* R1-R3 (the 2D multi-resolution DenseGrid = tinycudann.Encoding): PARITY UNPINNED.
  The implementation lives in an un-vendored, un-pinned tiny-cuda-nn fork
  (reference README.md:30-32) that is absent from /root/reference.  What the
  reference itself pins (level resolutions/offsets eval.py:28-35, feature-innermost
  cell-major layout eval.py:41 / compression.py:71-72, no per-level padding
  compression.py:77, fp32 modules.py:15, width L*F modules.py:42-44) is honoured;
  the interpolation follows the published tiny-cuda-nn GridEncoding (Dense grid,
  Linear interpolation) algorithm as restated in `dense_grid_2d` - by default in upstream's
  arithmetic (fmaf position, fma-chain blend, fp32 exp2f level scale); the plain two-rounding
  restatement and a clamping border are selectable per encoding_config ("variant", "border").

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

HIDDEN = 128
LRELU_SLOPE = 0.01          # nn.LeakyReLU() default, reference modulation.py:106
W0_FIRST = 30.0             # reference modules.py:37 (w0_initial)


# --------------------------------------------------------------------------------------
# R1: level geometry of the 2D DenseGrid ("learnable keyframes")
# --------------------------------------------------------------------------------------
def _f32(x: float) -> float:
    return float(torch.tensor(x, dtype=torch.float64).to(torch.float32))


def dense_grid_variant(cfg: dict) -> Tuple[bool, bool, bool, str]:
    """(pos_fma, interp_fma, clamp, scale_mode) of an encoding_config (same keys as the product's
    nvp_amd._lib.grid_variant, restated here; the oracle never imports the product):
    "variant": "tcnn" (default: published tiny-cuda-nn arithmetic) | "two_rounding"; "border": "wrap" | "clamp";
    "scale_mode": "fp32_exp2" | "double"."""
    variant = cfg.get("variant", "tcnn")
    assert variant in ("tcnn", "two_rounding")
    border = cfg.get("border", "wrap")
    assert border in ("wrap", "clamp")
    fma = variant == "tcnn"
    scale_mode = cfg.get("scale_mode", "fp32_exp2" if fma else "double")
    assert scale_mode in ("fp32_exp2", "double")
    return fma, fma, border == "clamp", scale_mode


def dense_grid_levels(cfg: dict) -> Tuple[List[float], List[int], List[int]]:
    """Per-level (scale, resolution, cell offset) of a tcnn DenseGrid.

    Resolutions / offsets follow the reference's own restatement of the level geometry,
    experiment_scripts/eval.py:28-35 (== compression.py:26-33), in double on the host as eval.py does:
        a = exp(l * log(per_level_scale)) * base - 1 ;  res = ceil(a) + 1 ; offset += res**2
    `scale` (the value every kernel multiplies with): scale_mode "double" = `a` rounded once to fp32;
    "fp32_exp2" (default) = published tiny-cuda-nn's grid_scale(), exp2f(l * log2f(pls)) * base - 1.0f with every
    step rounded to fp32 [upstream, from memory - parity unpinned].
    """
    n_levels = int(cfg["n_levels"])
    base = float(cfg.get("base_resolution", 16))
    pls = float(cfg["per_level_scale"])
    scale_mode = dense_grid_variant(cfg)[3]
    log2_pls = _f32(math.log2(_f32(pls)))
    scales, ress, offs = [], [], []
    total = 0
    for lvl in range(n_levels):
        a = math.exp(lvl * math.log(pls)) * base - 1.0
        res = int(math.ceil(a) + 1)
        if scale_mode == "fp32_exp2":
            e = _f32(2.0 ** _f32(float(lvl) * log2_pls))
            sc = _f32(_f32(e * _f32(base)) - 1.0)
            assert int(math.ceil(sc) + 1) == res
        else:
            sc = _f32(a)
        scales.append(sc)
        ress.append(res)
        offs.append(total)
        total += res * res
    offs.append(total)
    return scales, ress, offs


def dense_grid_n_params(cfg: dict) -> int:
    """Length of the flat fp32 `params` vector: sum(res_l^2) * F, no padding
    (reference compression.py:77 `check == 0`; SURVEY section 6 bpp cross-check)."""
    return dense_grid_levels(cfg)[2][-1] * int(cfg["n_features_per_level"])


# --------------------------------------------------------------------------------------
# R2: DenseGrid forward  (tinycudann.Encoding.__call__, call sites modules.py:65-67)
# --------------------------------------------------------------------------------------
def dense_grid_2d(params: torch.Tensor, x: torch.Tensor, cfg: dict) -> torch.Tensor:
    """x [N,2] in [0,1] -> [N, n_levels*F].  PARITY UNPINNED (see module docstring).

    Published tiny-cuda-nn GridEncoding semantics, Dense grid / Linear interpolation [upstream, from memory]:
      pos  = fmaf(scale_l, x, 0.5f)             (variant "two_rounding": fp32 multiply, then fp32 add)
      i    = floor(pos); w = pos - i
      out  = fma(w_c, P_l[cell(i+c)], out) over the 4 corners c in {0,1}^2, out starting at 0,
             w_c = prod_d (c_d ? w_d : 1-w_d)   (variant "two_rounding": out + fl(w_c * P))
      cell = (ix + iy*res_l) mod res_l^2        (dim 0 is the fast axis; upper border wraps; border "clamp":
                                                 ix+1, iy+1 clamp to res_l - 1 instead)
    corner order (0,0),(1,0),(0,1),(1,1).  The fused multiply-adds are emulated in float64: the product of two fp32
    values is exact in float64 and the sum is rounded to fp32 once (exact fma except for double-rounding ties, ~2^-29).
    Parameter layout: level-major, cell-major inside a level, feature innermost
    (reference eval.py:41, compression.py:51-58,71-72).
    """
    Fdim = int(cfg["n_features_per_level"])
    scales, ress, offs = dense_grid_levels(cfg)
    pos_fma, interp_fma, clamp, _ = dense_grid_variant(cfg)
    P = params.reshape(-1, Fdim)
    outs = []
    x0 = x[:, 0].to(torch.float32)
    x1 = x[:, 1].to(torch.float32)
    for scale, res, off in zip(scales, ress, offs[:-1]):
        s = torch.tensor(scale, dtype=torch.float32)
        if pos_fma:
            p0 = (x0.double() * s.double() + 0.5).float()
            p1 = (x1.double() * s.double() + 0.5).float()
        else:
            p0 = x0 * s + 0.5
            p1 = x1 * s + 0.5
        f0 = torch.floor(p0)
        f1 = torch.floor(p1)
        w0 = p0 - f0
        w1 = p1 - f1
        i0 = f0.to(torch.int64)
        i1 = f1.to(torch.int64)
        acc = None
        for c1 in (0, 1):
            for c0 in (0, 1):
                wt = (w0 if c0 else (1.0 - w0)) * (w1 if c1 else (1.0 - w1))
                if clamp:
                    cell = torch.clamp(i0 + c0, 0, res - 1) + torch.clamp(i1 + c1, 0, res - 1) * res
                else:
                    cell = ((i0 + c0) + (i1 + c1) * res) % (res * res)
                v = P[off + cell]
                if P.dtype == torch.float64:
                    # float64 parameters: the "exact" evaluation used as the yardstick for gradient accuracy - cell
                    # selection and corner weights stay the fp32 arithmetic above, blend and everything after it run in float64
                    term = wt.double().unsqueeze(1) * v
                    acc = term if acc is None else acc + term
                elif interp_fma:
                    prod = wt.double().unsqueeze(1) * v.double()
                    acc = prod.float() if acc is None else (prod + acc.double()).float()
                else:
                    term = wt.unsqueeze(1) * v
                    acc = term if acc is None else acc + term
        outs.append(acc)
    return torch.cat(outs, dim=1)


# --------------------------------------------------------------------------------------
# R5 / R7: SparseGrid forward  (reference sparsegrid.py:23-72 and :76-156)
# --------------------------------------------------------------------------------------
def _nearest_index(c: torch.Tensor, res: int) -> torch.Tensor:
    """clamp(int64(fp32((res-1)*c) + 0.5), 0, res-1): multiply and add are rounded
    separately, conversion truncates toward zero (reference sparsegrid.py:43-46)."""
    f = (res - 1) * c
    return torch.clamp((f + 0.5).to(torch.int64), 0, res - 1)


def _patch_indices(inputs: torch.Tensor, x_res: int, y_res: int):
    """[N,9] x- and y-indices of the clamped 3x3 neighbourhood, ordered i (x offset,
    outer) then j (y offset, inner) exactly like the double loop at sparsegrid.py:61-64."""
    xi = _nearest_index(inputs[:, 1], x_res)
    yi = _nearest_index(inputs[:, 2], y_res)
    d = torch.tensor([-1, 0, 1], dtype=torch.int64)
    vx = torch.clamp(xi[:, None] + d[None, :], 0, x_res - 1)      # [N,3]
    vy = torch.clamp(yi[:, None] + d[None, :], 0, y_res - 1)      # [N,3]
    vx9 = vx[:, :, None].expand(-1, 3, 3).reshape(-1, 9)
    vy9 = vy[:, None, :].expand(-1, 3, 3).reshape(-1, 9)
    return vx9, vy9


def _maybe_upsample(emb: torch.Tensor, upsample: bool) -> torch.Tensor:
    """x2 bilinear upsample of the (x,y) axes of the whole grid
    (reference sparsegrid.py:26-34: 4-D interpolate over [dim, T, X, Y])."""
    if not upsample:
        return emb
    t = emb.permute(3, 0, 1, 2)
    t = F.interpolate(t, scale_factor=2, mode="bilinear")
    return t.permute(1, 2, 3, 0)


def sparse_grid_forward(emb: torch.Tensor, inputs: torch.Tensor, upsample: bool = False) -> torch.Tensor:
    """emb [T,X,Y,F], inputs [N,3]=(t,x,y) -> [N,9F]: nearest-t, clamped 3x3 (x,y)
    neighbourhood copy, no interpolation weights (reference sparsegrid.py:42-72)."""
    emb = _maybe_upsample(emb, upsample)
    T, X, Y, Fd = emb.shape
    ti = _nearest_index(inputs[:, 0], T)
    vx9, vy9 = _patch_indices(inputs, X, Y)
    return emb[ti[:, None], vx9, vy9, :].reshape(inputs.shape[0], 9 * Fd)


def sparse_grid_forward_inter(emb: torch.Tensor, inputs: torch.Tensor, upsample: bool = False) -> torch.Tensor:
    """t-linear variant used by eval.py --t_interp (reference sparsegrid.py:76-156).

    tf = (T-1)*t; lo = trunc(tf); hi = clamp(trunc(tf+1)); uc = tf-lo; lc = hi-tf;
    uc' = uc/(uc+lc); lc' = lc/(uc'+lc)      <- the second line uses the UPDATED uc (:108-109)
    out = lc'*patch(E[lo]) + uc'*patch(E[hi]).  At tf == T-1, hi == lo -> 0/0 -> NaN row
    (reference behaviour; pinned by the golden vectors).
    """
    emb = _maybe_upsample(emb, upsample)
    T, X, Y, Fd = emb.shape
    tf = (T - 1) * inputs[:, 0]
    lo = tf.to(torch.int64)
    hi = torch.clamp((tf + 1).to(torch.int64), 0, T - 1)
    uc = tf - lo
    lc = hi - tf
    uc = uc / (uc + lc)
    lc = lc / (uc + lc)
    vx9, vy9 = _patch_indices(inputs, X, Y)
    N = inputs.shape[0]
    lo_feat = emb[lo[:, None], vx9, vy9, :].reshape(N, 9 * Fd) * lc[:, None]
    hi_feat = emb[hi[:, None], vx9, vy9, :].reshape(N, 9 * Fd) * uc[:, None]
    return lo_feat + hi_feat


# --------------------------------------------------------------------------------------
# R8-R10: Modulator + modulated SIREN (reference modulation.py:96-121, 60-92, 138-145)
# --------------------------------------------------------------------------------------
def modulator_forward(z: torch.Tensor, Ws: Sequence[torch.Tensor], bs: Sequence[torch.Tensor]):
    """h0 = lrelu(z W0^T + b0); h_k = lrelu([h_{k-1}, z] W_k^T + b_k)
    (skip-concat puts the hidden first, the latent last: modulation.py:119)."""
    hs = []
    x = z
    for W, b in zip(Ws, bs):
        h = F.leaky_relu(F.linear(x, W, b), LRELU_SLOPE)
        hs.append(h)
        x = torch.cat((h, z), dim=1)
    return tuple(hs)


def modulator_preacts(z: torch.Tensor, Ws: Sequence[torch.Tensor], bs: Sequence[torch.Tensor]):
    """The three LeakyReLU inputs (same arithmetic as modulator_forward).  Tests use them to
    keep gradient comparisons away from the kink at 0, where d/dp jumps from 0.01 to 1 and a
    1-ulp difference in p (summation order) legitimately changes the gradient."""
    pre = []
    x = z
    for W, b in zip(Ws, bs):
        p = F.linear(x, W, b)
        pre.append(p)
        x = torch.cat((F.leaky_relu(p, LRELU_SLOPE), z), dim=1)
    return tuple(pre)


def siren_forward(s: torch.Tensor, mods, Ws: Sequence[torch.Tensor], bs: Sequence[torch.Tensor],
                  W_last: torch.Tensor, b_last: torch.Tensor, w0_first: float = W0_FIRST) -> torch.Tensor:
    """x_k = sin(w0_k * (x_{k-1} W_k^T + b_k)) * mod_k, w0 = (30, 1, 1); rgb = x W_last^T + b_last
    (modulation.py:53-56 Siren.forward, :24-25 Sine, :83-92 SirenNet.forward; Identity tail)."""
    x = s.to(Ws[0].dtype)                 # (float64 parameters: the yardstick evaluation, see dense_grid_2d)
    for k, (W, b) in enumerate(zip(Ws, bs)):
        w0 = w0_first if k == 0 else 1.0
        x = torch.sin(w0 * F.linear(x, W, b))
        x = x * mods[k]
    return F.linear(x, W_last, b_last)


STATE_KEYS_MLP = (
    [f"wrapper.modulator.layers.{k}.0.{p}" for k in range(3) for p in ("weight", "bias")]
    + [f"net.layers.{k}.{p}" for k in range(3) for p in ("weight", "bias")]
    + ["net.last_layer.weight", "net.last_layer.bias"]
)


def mlp_forward(latent: torch.Tensor, steps: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """SirenWrapper.forward(coords=steps [N,1], latent [N,D]) -> [N,3] (modulation.py:138-145)."""
    mW = [sd[f"wrapper.modulator.layers.{k}.0.weight"] for k in range(3)]
    mb = [sd[f"wrapper.modulator.layers.{k}.0.bias"] for k in range(3)]
    sW = [sd[f"net.layers.{k}.weight"] for k in range(3)]
    sb = [sd[f"net.layers.{k}.bias"] for k in range(3)]
    mods = modulator_forward(latent, mW, mb)
    return siren_forward(steps, mods, sW, sb, sd["net.last_layer.weight"], sd["net.last_layer.bias"])


# --------------------------------------------------------------------------------------
# R11: NVP.forward (reference modules.py:51-84)
# --------------------------------------------------------------------------------------
def nvp_latent(coords: torch.Tensor, sd: Dict[str, torch.Tensor], cfg: dict, temporal_interp: bool = False):
    """coords [N,3]=(t,x,y) -> latent [N, 3*16F + 9F]; column order xy, yt, xt, sparse
    (modules.py:61-69,78).  Plane inputs: xy<-(x,y)=cols[1,2]; xt<-(t,x)=cols[0,1];
    yt<-(t,y)=cols[0,2] (modules.py:61-63)."""
    xy = dense_grid_2d(sd["keyframes_xy.params"], coords[:, [1, 2]], cfg["2d_encoding_xy"])
    xt = dense_grid_2d(sd["keyframes_xt.params"], coords[:, [0, 1]], cfg["2d_encoding_xt"])
    yt = dense_grid_2d(sd["keyframes_yt.params"], coords[:, [0, 2]], cfg["2d_encoding_yt"])
    up = bool(cfg["3d_encoding"].get("upsample", False))
    if temporal_interp:
        sp = sparse_grid_forward_inter(sd["sparse_grid.embeddings"], coords, up)
    else:
        sp = sparse_grid_forward(sd["sparse_grid.embeddings"], coords, up)
    return torch.cat((xy, yt, xt, sp), dim=1)


def nvp_forward(all_coords: torch.Tensor, temporal_steps: torch.Tensor, sd: Dict[str, torch.Tensor],
                cfg: dict, temporal_interp: bool = False) -> torch.Tensor:
    """model_input {'all_coords': [b,t,3], 'temporal_steps': [b,t]} -> model_out [b,t,3]."""
    b, t = temporal_steps.shape[0], temporal_steps.shape[1]
    steps = temporal_steps.reshape(b * t, -1)
    coords = all_coords.reshape(-1, 3)
    latent = nvp_latent(coords, sd, cfg, temporal_interp)
    latent = latent.to(sd["net.last_layer.weight"].dtype)
    return mlp_forward(latent, steps, sd).reshape(b, t, 3)


def image_mse(out: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """mean((out-gt)^2) over every element (reference loss_functions.py:1-3)."""
    return ((out - gt) ** 2).mean()


# --------------------------------------------------------------------------------------
# Parameter construction with the reference's init distributions (R1, R4, R8, R9)
# --------------------------------------------------------------------------------------
def latent_dim(cfg: dict) -> int:
    d = 0
    for k in ("2d_encoding_xy", "2d_encoding_yt", "2d_encoding_xt"):
        d += int(cfg[k]["n_levels"]) * int(cfg[k]["n_features_per_level"])
    return d + 9 * int(cfg["3d_encoding"]["n_features_per_level"])


def init_state(cfg: dict, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Fresh parameters with the reference's distributions (not its RNG streams):
    grids U(-1e-4,1e-4) (sparsegrid.py:19-21; tcnn default); modulator weight
    kaiming_normal(fan_in, relu) + nn.Linear default bias (modulation.py:151-154, :103);
    SIREN layer 0 U(+-1/dim_in), others U(+-sqrt(6/dim_in)/w0) incl. last (modulation.py:44-51)."""
    g = torch.Generator().manual_seed(seed)
    D = latent_dim(cfg)
    sd: Dict[str, torch.Tensor] = {}

    def uni(shape, a):
        return (torch.rand(shape, generator=g, dtype=dtype) * 2 - 1) * a

    for k in ("xy", "yt", "xt"):
        sd[f"keyframes_{k}.params"] = uni((dense_grid_n_params(cfg[f"2d_encoding_{k}"]),), 1e-4)
    c3 = cfg["3d_encoding"]
    sd["sparse_grid.embeddings"] = uni(
        (int(c3["t_resolution"]), int(c3["x_resolution"]), int(c3["y_resolution"]),
         int(c3["n_features_per_level"])), 1e-4)
    H = int(cfg["network"]["n_neurons"])
    for k in range(3):
        fan_in = D if k == 0 else H + D
        sd[f"wrapper.modulator.layers.{k}.0.weight"] = torch.randn((H, fan_in), generator=g, dtype=dtype) * math.sqrt(2.0 / fan_in)
        sd[f"wrapper.modulator.layers.{k}.0.bias"] = uni((H,), 1.0 / math.sqrt(fan_in))
    for k in range(3):
        din = 1 if k == 0 else H
        std = (1.0 / din) if k == 0 else math.sqrt(6.0 / din)
        sd[f"net.layers.{k}.weight"] = uni((H, din), std)
        sd[f"net.layers.{k}.bias"] = uni((H,), std)
    std = math.sqrt(6.0 / H)
    sd["net.last_layer.weight"] = uni((3, H), std)
    sd["net.last_layer.bias"] = uni((3,), std)
    return sd


# --------------------------------------------------------------------------------------
# Row H: sampler + one optimisation step, as the reference harness does them
# --------------------------------------------------------------------------------------
def get_mgrid_2d(H: int, W: int) -> torch.Tensor:
    """[H*W, 2] (row/(H-1), col/(W-1)), row-major (reference dataio.py:11-20,29)."""
    r = torch.arange(H, dtype=torch.float32) / (H - 1)
    c = torch.arange(W, dtype=torch.float32) / (W - 1)
    return torch.stack(torch.meshgrid(r, c, indexing="ij"), dim=-1).reshape(-1, 2)


def sample_batch(T: int, H: int, W: int, n: int, gen: torch.Generator):
    """Reference sampler order (dataio.py:106-118): temporal randint first, then spatial.
    Returns (ti, pi, all_coords [n,3], temporal_steps [n])."""
    ti = torch.randint(0, T, (n,), generator=gen)
    pi = torch.randint(0, H * W, (n,), generator=gen)
    tcoord = torch.linspace(0, 1, T)[ti]
    half_dt = 0.5 / T
    tstep = torch.linspace(half_dt, 1 - half_dt, T)[ti]
    row = torch.div(pi, W, rounding_mode="floor").to(torch.float32) / (H - 1)
    col = (pi % W).to(torch.float32) / (W - 1)
    coords = torch.stack((tcoord, row, col), dim=1)
    return ti, pi, coords, tstep


def normalise_gt(u8: torch.Tensor) -> torch.Tensor:
    """(x - 127.5)/127.5 in fp32 (reference training.py:47-48)."""
    return (u8.to(torch.float32) - 127.5) / 127.5
