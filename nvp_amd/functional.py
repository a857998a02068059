"""torch.autograd.Function wrappers around the HIP kernels (autograd contract).

Every function enqueues hand-written gfx950 kernels from libnvp_hip.so on the current
HIP stream; PyTorch only owns the memory and the autograd tape.  Backward returns dense
fp32 gradients with the parameters' shapes (what torch.optim.AdamW consumes), and `None`
for coordinates / temporal steps, exactly like the reference's autograd graph.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import List, Sequence, Tuple

import torch

from . import _lib as L


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class KernelTimer:
    """Optional HIP-event timer around individual kernel launches (bench.py roofline leg).
    Events are recorded on torch's current stream, which is the stream every launch uses."""

    def __init__(self):
        self.spans = {}

    def run(self, name, fn, *args):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        self.spans.setdefault(name, []).append((a, b))
        return rc

    def summary(self):
        """name -> (mean ms, count); call after a device synchronize."""
        return {k: (sum(a.elapsed_time(b) for a, b in v) / len(v), len(v)) for k, v in self.spans.items()}

    def reset(self):
        self.spans = {}


TIMER = None      # set to a KernelTimer() to time launches
DW_SIDE_STREAM = os.environ.get("NVP_DW_SIDE_STREAM", "0") == "1"      # experiment: dW GEMMs concurrent with the grid scatter
# NVPFused: the small kernels that depend on the coordinates or on the parameters only (the scatter's keys / sorts / sparse row table,
# the weight packing for forward and backward) run on a side stream underneath the gather kernel (NVP_SCATTER_PRESORT=0: in line)
SIDE_WORK = os.environ.get("NVP_SCATTER_PRESORT", "1") != "0"
_SIDE_STREAMS = {}          # device index -> the side stream of that device (one process normally drives one GPU)


def _side_stream(dev: torch.device) -> "torch.cuda.Stream":
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(idx)
    if st is None:
        st = _SIDE_STREAMS[idx] = L.side_stream(dev)
    return st

class StepHooks:
    """Per-call hooks of ONE NVPFused forward/backward pair.  The caller (harness.train_step) creates one, hands it to the model as
    `model_input['nvp_hooks']` and may fill it in any time before backward runs; backward reads it from its own autograd context,
    so two models, two threads or two overlapping steps in one process never see each other's hooks (autograd runs backward on
    its own worker thread: neither module globals nor thread-locals would be safe).

      grad_sink     {param.data_ptr(): preallocated tensor} (data parallelism): backward writes that parameter's gradient straight
                    into the tensor (a view of the flat exchange bucket) and returns it - no zero-fill / accumulate / flatten pass;
      sparse_ready  callback() fired as soon as the sparse grid's gradient alone has been enqueued (only when the scatter runs
                    sparse-first: y-sorted batches with the level-major hand-over); the dense planes follow;
      grids_ready   callback() fired as soon as all four grid gradients have been enqueued (before the dW GEMMs): data parallelism
                    starts their exchange there (parallel.GradBucket / ShardedAdamW);
      early_grads   callback([(parameter tensor, gradient tensor)]) fired as soon as those grid gradients have been enqueued - the
                    sparse grid's first, then the three planes' (optim.AdamW.early_update runs their AdamW on a side stream,
                    underneath the rest of the scatter and the dW GEMMs).  Only meaningful when each grid feeds ONE NVPFused call
                    per step;
      fused_sparse  an optimizer with fused_peek(tensor) / fused_commit(tensor) (optim.AdamW; one GPU, no grad_sink): the sparse
                    grid's AdamW step is applied INSIDE the scatter's flush (nvp_encode_bwd_sparse_adamw) - its gradient is never
                    written to HBM, backward returns None for it.  Falls back to the gradient + early_grads route when the
                    optimizer declines (peek returns None) or the kernel does not support the layout;
      loss_gt       uint8 ground truth [.., N, 3] of THIS batch, set BEFORE the forward call by a caller that promises the loss is
                    image_mse(model_out, (gt - 127.5) / 127.5) seeded with gradient 1 (harness.train_step does): forward then runs the
                    tile-fused step kernel (nvp_encode_mlp_fwd_bwd: forward, the loss gradient and the backward chain per 32-pixel tile
                    in one launch; bit-identical streams and gradients) and backward starts at the scatter.  Ignored where the kernel
                    does not apply (nvp_l, unsorted batches, data parallel sinks are fine);
      fused_dense   the same for the three dense planes (nvp_encode_bwd_dense_adamw: the update runs in band_kernel's /
                    slab_reduce_kernel's flushes).  With both set, no optimizer launch is left for the four grids."""
    __slots__ = ("grad_sink", "sparse_ready", "grids_ready", "early_grads", "fused_sparse", "fused_dense", "loss_gt", "packed_cache")

    def __init__(self, grad_sink=None, sparse_ready=None, grids_ready=None, early_grads=None, fused_sparse=None, fused_dense=None, loss_gt=None):
        self.grad_sink, self.sparse_ready, self.grids_ready, self.early_grads = grad_sink, sparse_ready, grids_ready, early_grads
        self.fused_sparse, self.fused_dense = fused_sparse, fused_dense
        self.loss_gt = loss_gt
        self.packed_cache = None      # inference only: a dict the caller keeps for as long as the parameters do not change (one frame / one
                                      # evaluation under no_grad): the packed forward weights are built once and reused by every slice

    def clear(self) -> None:
        self.grad_sink = self.sparse_ready = self.grids_ready = self.early_grads = self.fused_sparse = self.fused_dense = self.loss_gt = None


_NO_HOOKS = StepHooks()

# Batches that do not arrive sorted by their y coordinate (the reference's own sampler, dataio.py:104-120) are
# put into that order inside NVPFused for the duration of the step and the RGB rows are returned in the
# caller's order: the grid gathers get row locality (encode 1.07 -> 0.75 ms at N = 1.2 M) and the scatter skips a
# sort.  Only worth it for training-size batches; 0 disables.
AUTO_SORT_MIN = int(os.environ.get("NVP_AUTO_SORT_MIN", "65536"))
ROW_ORDER = os.environ.get("NVP_ROW_ORDER", "nvp")            # "torch": torch.argsort of y instead of nvp_order_by_rows (A/B)

# model_input['sorted_by_y'] is a promise by the caller (nvp_amd's own sampler makes it): the gradient scatter then skips
# its y radix sort and binary-searches row starts in the batch as delivered, so a wrong hint gives silently wrong keyframe
# gradients.  NVP_CHECK_SORTED=1 verifies the promise on every call (one pass over N floats + a host sync): for debugging
# and for tests; off by default because the sync would serialise the training loop.
CHECK_SORTED = os.environ.get("NVP_CHECK_SORTED", "0") == "1"

# NVP.forward in ONE launch: the grid lookups run inside the forward MLP's waves (nvp_encode_mlp_fwd; config_nvp_s-sized latents).
# NVP_FUSED_FWD=0 keeps the two-kernel path (gather kernel -> latent in HBM -> MLP kernel); RGB and gradients are bit-identical.
FUSED_FWD = os.environ.get("NVP_FUSED_FWD", "1") != "0"
# harness.train_step: forward, loss gradient and backward chain of a tile in ONE launch (nvp_encode_mlp_fwd_bwd; StepHooks.loss_gt).  EXPERIMENT:
# bit-identical, measured 0.3 ms SLOWER than the two kernels; the entry point exists in libnvp_hip_experiments.so only
# (NVP_HIP_LIB=.../libnvp_hip_experiments.so NVP_TILE_FUSED=1)
TILE_FUSED = os.environ.get("NVP_TILE_FUSED", "0") == "1"

# For y-sorted batches the backward chain hands the xy / yt planes' latent gradients to the scatter in its own level-major
# layout (nvp_encode_bwd_prepare / NVP_DZ_PLANES_READY): 2/3 of the scatter's permute pass disappear.  NVP_DZ_LEVEL_MAJOR=0 keeps
# the row-major hand-over for every plane (bit-identical gradients either way).
DZ_LEVEL_MAJOR = os.environ.get("NVP_DZ_LEVEL_MAJOR", "1") != "0"

def _row_order(lib, coords: torch.Tensor, n: int, lv_xy, lv_yt) -> torch.Tensor:
    """Permutation that puts a caller-order batch into non-decreasing grid-row order at every level of the xy / yt planes (what
    `y_sorted` promises the scatter): one stable counting sort on a 13-bit row key (nvp_order_by_rows) - no library sort on the
    reference-surface path.  Exotic level geometries (key space > 12 288) fall back to a stable argsort of y."""
    ws_bytes = lib.nvp_order_by_rows_workspace_bytes(n, C.byref(lv_xy), C.byref(lv_yt))
    if ws_bytes >= 0 and ROW_ORDER != "torch":
        ws = torch.empty(max(int(ws_bytes), 4), device=coords.device, dtype=torch.uint8)
        order = torch.empty(n, device=coords.device, dtype=torch.int64)
        rc = lib.nvp_order_by_rows(L.ptr(coords), L.ptr(order, torch.int64), n, C.byref(lv_xy), C.byref(lv_yt), L.ptr(ws, torch.uint8), ws.numel(), L.stream_ptr())
        if rc != L.ERR_UNSUPPORTED:
            L.check(rc, "nvp_order_by_rows")
            return order
    return torch.argsort(coords[:, 2], stable=True)


def _grad_buffer(param: torch.Tensor, sink=None) -> torch.Tensor:
    if sink is not None:
        t = sink.get(param.data_ptr())
        if t is not None and t.shape == param.shape and t.is_contiguous() and t.dtype == torch.float32:
            return t.detach()      # fresh alias (use_count 1) so autograd adopts it as .grad without a clone
    return torch.empty_like(param)


def _call(name, fn, *args):
    if TIMER is not None:
        return TIMER.run(name, fn, *args)
    return fn(*args)


def dw_chunks(n: int) -> int:
    """Number of pixel chunks of the split-K weight-gradient GEMMs."""
    cap = int(os.environ.get("NVP_DW_CHUNKS", "256"))
    return max(1, min(cap, L.ntiles(n) // 8))


# --------------------------------------------------------------------------------------
# tinycudann.Encoding  (R2 / R3)
# --------------------------------------------------------------------------------------
class DenseGrid2D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, params: torch.Tensor, levels: L.Levels) -> torch.Tensor:
        lib = L.load()
        x = _f32c(x)
        params = _f32c(params)
        if x.dim() != 2 or x.shape[1] != 2:
            raise RuntimeError(f"Encoding expects [N,2] inputs, got {tuple(x.shape)}")
        if params.numel() != L.levels_n_params(levels):
            raise RuntimeError("params has the wrong length for this encoding_config")
        n = x.shape[0]
        out = torch.empty((n, levels.n_levels * levels.n_features), device=x.device, dtype=torch.float32)
        L.check(lib.nvp_dense2d_fwd(L.ptr(params), L.ptr(x), L.ptr(out), n, C.byref(levels), L.stream_ptr()), "nvp_dense2d_fwd")
        ctx.levels = levels
        ctx.save_for_backward(x, params)
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        lib = L.load()
        x, params = ctx.saved_tensors
        dparams = torch.zeros_like(params)
        dout = _f32c(dout)
        L.check(lib.nvp_dense2d_bwd(L.ptr(x), L.ptr(dout), L.ptr(dparams), x.shape[0], C.byref(ctx.levels), L.stream_ptr()),
                "nvp_dense2d_bwd")
        return None, dparams, None


# --------------------------------------------------------------------------------------
# SparseGrid  (R5 / R6 / R7)
# --------------------------------------------------------------------------------------
def _sparse_shape(emb: torch.Tensor) -> L.SparseShape:
    if emb.dim() != 4:
        raise RuntimeError("embeddings must be [T, X, Y, F]")
    T, X, Y, F = emb.shape
    return L.SparseShape(T, X, Y, F)


class SparseUpsample2x(torch.autograd.Function):
    """SparseGrid(upsample=True): emb [T,X,Y,F] -> [T,2X,2Y,F], the reference's permute -> F.interpolate(scale_factor=2, 'bilinear') ->
    permute (sparsegrid.py:26-34) as ONE HIP pass (nvp_sparse_upsample2x_fwd); backward is its adjoint as a deterministic gather."""

    @staticmethod
    def forward(ctx, emb: torch.Tensor) -> torch.Tensor:
        lib = L.load()
        emb = _f32c(emb)
        sh = _sparse_shape(emb)
        out = torch.empty((sh.t_res, 2 * sh.x_res, 2 * sh.y_res, sh.n_features), device=emb.device, dtype=torch.float32)
        L.check(lib.nvp_sparse_upsample2x_fwd(L.ptr(emb), L.ptr(out), C.byref(sh), L.stream_ptr()), "nvp_sparse_upsample2x_fwd")
        ctx.sh = sh
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        lib = L.load()
        sh = ctx.sh
        dout = _f32c(dout)
        demb = torch.empty((sh.t_res, sh.x_res, sh.y_res, sh.n_features), device=dout.device, dtype=torch.float32)
        L.check(lib.nvp_sparse_upsample2x_bwd(L.ptr(dout), L.ptr(demb), C.byref(sh), L.stream_ptr()), "nvp_sparse_upsample2x_bwd")
        return demb


class SparseGrid3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs: torch.Tensor, emb: torch.Tensor, inter: bool) -> torch.Tensor:
        lib = L.load()
        inputs = _f32c(inputs)
        emb = _f32c(emb)
        if inputs.dim() != 2 or inputs.shape[1] != 3:
            raise RuntimeError(f"SparseGrid expects [N,3] inputs (t,x,y), got {tuple(inputs.shape)}")
        sh = _sparse_shape(emb)
        n = inputs.shape[0]
        out = torch.empty((n, 9 * sh.n_features), device=inputs.device, dtype=torch.float32)
        fn = lib.nvp_sparse3x3_inter_fwd if inter else lib.nvp_sparse3x3_fwd
        L.check(fn(L.ptr(emb), L.ptr(inputs), L.ptr(out), n, C.byref(sh), L.stream_ptr()), "nvp_sparse3x3_fwd")
        ctx.inter = inter
        ctx.sh = sh
        ctx.save_for_backward(inputs, emb)
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        if ctx.inter:
            raise NotImplementedError("forward_inter is an inference-only path (reference eval.py --t_interp)")
        lib = L.load()
        inputs, emb = ctx.saved_tensors
        demb = torch.zeros_like(emb)
        dout = _f32c(dout)
        L.check(lib.nvp_sparse3x3_bwd(L.ptr(inputs), L.ptr(dout), L.ptr(demb), inputs.shape[0], C.byref(ctx.sh), L.stream_ptr()),
                "nvp_sparse3x3_bwd")
        return None, demb, None


# --------------------------------------------------------------------------------------
# MLP core shared by SirenWrapper and the fused NVP function
# --------------------------------------------------------------------------------------
def _mlp_forward(zt: torch.Tensor, steps: torch.Tensor, mlp: Sequence[torch.Tensor], n: int, d: int, save: bool, packed=None):
    """`packed`: (tensor, event) of weights already packed on another stream (NVPFused packs underneath the gather kernel)."""
    lib = L.load()
    dev = zt.device
    if n == 0:
        return torch.empty((0, 3), device=dev, dtype=torch.float32), None
    stream = L.stream_ptr()
    pstruct = L.mlp_params_struct(mlp)
    if packed is not None:
        packed, ev = packed
        if ev is not None:
            torch.cuda.current_stream(dev).wait_event(ev)
    else:
        packed = torch.empty(lib.nvp_packed_fwd_floats(d), device=dev, dtype=torch.float32)
        L.check(lib.nvp_mlp_pack_fwd(C.byref(pstruct), L.ptr(packed), d, stream), "nvp_mlp_pack_fwd")
    rgb = torch.empty((n, 3), device=dev, dtype=torch.float32)
    saved = torch.empty((5, L.ntiles(n), L.HIDDEN, L.TILE), device=dev, dtype=torch.float32) if save else None
    L.check(_call("nvp_mlp_fwd", lib.nvp_mlp_fwd, L.ptr(zt), L.ptr(steps), C.byref(pstruct), L.ptr(packed), L.ptr(rgb), L.ptr(saved), n, d, stream),
            "nvp_mlp_fwd")
    return rgb, saved


def _mlp_backward(drgb: torch.Tensor, steps: torch.Tensor, zt: torch.Tensor, saved: torch.Tensor,
                  mlp: Sequence[torch.Tensor], n: int, d: int, between=None, lm=None, packed=None, sink=None, chain=None) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """dX chain (+ latent gradient), then the dW GEMMs.  `between(dz_rows)`, if given, runs after the
    dX kernels are enqueued and before the dW kernels: the fused NVP path uses it to enqueue the grid
    scatter (which only needs dz) first, so its gradients can be all-reduced underneath the dW GEMMs."""
    lib = L.load()
    dev = zt.device
    stream = L.stream_ptr()
    nt = L.ntiles(n)
    pstruct = L.mlp_params_struct(mlp)
    drgb = _f32c(drgb)
    if chain is not None:           # (dy, dz_rows) of the tile-fused step kernel (forward): the chain has run
        dy, dz_rows = chain
    else:
        if packed is not None:          # (tensor, event): packed during forward on the side stream
            packed, ev = packed
            torch.cuda.current_stream(dev).wait_event(ev)
        else:
            packed = torch.empty(lib.nvp_packed_bwd_floats(d), device=dev, dtype=torch.float32)
            L.check(lib.nvp_mlp_pack_bwd(C.byref(pstruct), L.ptr(packed), d, stream), "nvp_mlp_pack_bwd")
        dy = torch.empty((6, nt, L.HIDDEN, L.TILE), device=dev, dtype=torch.float32)
        dz_rows = torch.empty((nt * L.TILE, lib.nvp_dz_stride(d)), device=dev, dtype=torch.float32)
        # lm (optional, fused NVP path with y-sorted batches): the scatter's level-major buffers for the xy / yt planes
        L.check(_call("nvp_mlp_bwd_dx", lib.nvp_mlp_bwd_dx, L.ptr(drgb), L.ptr(steps), L.ptr(saved), C.byref(pstruct), L.ptr(packed),
                                   L.ptr(dy), L.ptr(dz_rows), C.byref(lm) if lm is not None else None, n, d, stream), "nvp_mlp_bwd_dx")
    grads = [_grad_buffer(t, sink) for t in mlp]
    gstruct = L.mlp_params_struct(grads)
    nch = dw_chunks(n)
    partials = torch.empty(lib.nvp_dw_partial_floats(d, nch), device=dev, dtype=torch.float32)
    if DW_SIDE_STREAM and between is not None:
        # experiment (NVP_DW_SIDE_STREAM=1): the dW GEMMs on a second stream, concurrent with the grid scatter
        side = _side_stream(dev)
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            L.check(lib.nvp_mlp_bwd_dw(L.ptr(drgb), L.ptr(steps), L.ptr(zt), L.ptr(saved), L.ptr(dy), C.byref(pstruct),
                                       L.ptr(partials), nch, C.byref(gstruct), n, d, L.stream_ptr()), "nvp_mlp_bwd_dw")
        between(dz_rows)
        main.wait_stream(side)
        return dz_rows, grads
    if between is not None:
        between(dz_rows)
    L.check(_call("nvp_mlp_bwd_dw", lib.nvp_mlp_bwd_dw, L.ptr(drgb), L.ptr(steps), L.ptr(zt), L.ptr(saved), L.ptr(dy), C.byref(pstruct),
                               L.ptr(partials), nch, C.byref(gstruct), n, d, stream), "nvp_mlp_bwd_dw")
    return dz_rows, grads


def _packed_fwd_cached(lib, cache, pstruct, d: int, dev) -> torch.Tensor:
    """The forward weight pack on the current stream; `cache` (StepHooks.packed_cache, inference only) keeps it for the caller's evaluation."""
    packed = cache.get(("fwd", d, dev)) if cache is not None else None
    if packed is None:
        packed = torch.empty(lib.nvp_packed_fwd_floats(d), device=dev, dtype=torch.float32)
        L.check(lib.nvp_mlp_pack_fwd(C.byref(pstruct), L.ptr(packed), d, L.stream_ptr()), "nvp_mlp_pack_fwd")
        if cache is not None:
            cache[("fwd", d, dev)] = packed
    return packed


def _check_mlp(mlp: Sequence[torch.Tensor], d: int) -> None:
    H = L.HIDDEN
    want = [(H, d), (H,), (H, H + d), (H,), (H, H + d), (H,), (H, 1), (H,), (H, H), (H,), (H, H), (H,), (3, H), (3,)]
    if len(mlp) != 14:
        raise RuntimeError("expected 14 MLP tensors")
    for t, w in zip(mlp, want):
        if tuple(t.shape) != w:
            raise RuntimeError(f"MLP tensor has shape {tuple(t.shape)}, kernels are specialised for {w} "
                               "(n_neurons=128, n_hidden_layers=3, dim_out=3)")


class ModulatedSiren(torch.autograd.Function):
    """SirenWrapper.forward(coords=steps [N,1], latent [N,D]) -> [N,3]  (R8-R10, R12)."""

    @staticmethod
    def forward(ctx, latent: torch.Tensor, steps: torch.Tensor, grad_mode: bool, *mlp: torch.Tensor) -> torch.Tensor:
        lib = L.load()
        latent = _f32c(latent)
        n, d = latent.shape
        steps = _f32c(steps).reshape(-1)
        if steps.numel() != n:
            raise RuntimeError("coords and latent disagree on the batch size")
        mlp = [_f32c(t) for t in mlp]
        _check_mlp(mlp, d)
        rows = lib.nvp_latent_rows(d)
        zt = torch.empty((L.ntiles(n), rows, L.TILE), device=latent.device, dtype=torch.float32)
        if n:
            L.check(lib.nvp_rows_to_ptm(L.ptr(latent), L.ptr(zt), n, d, rows, L.stream_ptr()), "nvp_rows_to_ptm")
        need_grad = bool(grad_mode) and any(ctx.needs_input_grad)
        rgb, saved = _mlp_forward(zt, steps, mlp, n, d, save=need_grad)
        ctx.n, ctx.d, ctx.rows = n, d, rows
        if need_grad:
            if saved is None:
                saved = torch.empty(0, device=latent.device)
            ctx.save_for_backward(zt, steps, saved, *mlp)
        return rgb

    @staticmethod
    def backward(ctx, drgb: torch.Tensor):
        lib = L.load()
        zt, steps, saved, *mlp = ctx.saved_tensors
        n, d = ctx.n, ctx.d
        if n == 0:
            return (torch.zeros((0, d), device=zt.device), None, None, *[torch.zeros_like(t) for t in mlp])
        dz_rows, grads = _mlp_backward(drgb, steps, zt, saved, mlp, n, d)
        return (dz_rows[:n, :d].contiguous(), None, None, *grads)


def modulated_siren_streams(latent: torch.Tensor, steps: torch.Tensor, mlp: Sequence[torch.Tensor]) -> dict:
    """Run the fused forward kernel with its save path on and return what it saved, converted from PTM4 to row-major
    [N,128]: h0,h1,h2 (the Modulator outputs, modulation.py:112-121) and q1,q2 (pre-sine SIREN activations)."""
    lib = L.load()
    latent = _f32c(latent)
    n, d = latent.shape
    steps = _f32c(steps).reshape(-1)
    mlp = [_f32c(t) for t in mlp]
    _check_mlp(mlp, d)
    rows = lib.nvp_latent_rows(d)
    zt = torch.empty((L.ntiles(n), rows, L.TILE), device=latent.device, dtype=torch.float32)
    L.check(lib.nvp_rows_to_ptm(L.ptr(latent), L.ptr(zt), n, d, rows, L.stream_ptr()), "nvp_rows_to_ptm")
    _, saved = _mlp_forward(zt, steps, mlp, n, d, save=True)
    out = {}
    for k, name in enumerate(("h0", "h1", "h2", "q1", "q2")):
        r = torch.empty((n, L.HIDDEN), device=latent.device, dtype=torch.float32)
        L.check(lib.nvp_ptm_to_rows(L.ptr(saved[k]), L.ptr(r), n, L.HIDDEN, L.HIDDEN, L.stream_ptr()), "nvp_ptm_to_rows")
        out[name] = r
    return out


def _scatter_ws_bytes(lib, n, lv_xy, lv_yt, lv_xt, sh) -> int:
    b = lib.nvp_encode_bwd_workspace_bytes(n, C.byref(lv_xy), C.byref(lv_yt), C.byref(lv_xt), C.byref(sh))
    if b < 0:
        raise L.NvpHipError("nvp_encode_bwd_workspace_bytes failed")
    return b


def materialises_nothing(model, temporal_interp: bool) -> bool:
    """True when a no-grad forward of `model` writes no per-pixel tensor besides the RGB (the fused forward takes the call): then a whole
    frame can go through one call whatever its size (harness.render_frame); otherwise the call writes the latent [n, D] and callers bound n."""
    if not FUSED_FWD:
        return False
    sg = model.sparse_grid
    # the shape from the module's attributes: sg._grid() would run the whole x2 upsample pass (and allocate [T, 2X, 2Y, F]) only to be measured
    T_, X_, Y_, F_ = (int(v) for v in sg.embeddings.shape)
    up = 2 if getattr(sg, "upsample", False) else 1
    sh = L.SparseShape(T_, X_ * up, Y_ * up, F_)
    return int(L.load().nvp_encode_mlp_fwd_supported(C.byref(model.keyframes_xy.levels), C.byref(model.keyframes_yt.levels),
                                                      C.byref(model.keyframes_xt.levels), C.byref(sh))) == 1      # (2: fused, but the latent tensor is its workspace)


class NVPFused(torch.autograd.Function):
    """NVP.forward hot path (R11): coords [N,3], steps [N] -> rgb [N,3] in four kernels
    (encode -> pack -> MLP), the latent only ever exists in the MFMA-friendly PTM layout."""

    @staticmethod
    def forward(ctx, coords, steps, kf_xy, kf_yt, kf_xt, emb, lv_xy, lv_yt, lv_xt, temporal_interp, grad_mode, y_sorted, hooks, *mlp):
        lib = L.load()
        coords = _f32c(coords)
        steps = _f32c(steps).reshape(-1)
        kf_xy, kf_yt, kf_xt, emb = _f32c(kf_xy), _f32c(kf_yt), _f32c(kf_xt), _f32c(emb)
        mlp = [_f32c(t) for t in mlp]
        n = coords.shape[0]
        sh = _sparse_shape(emb)
        for kf, lv in ((kf_xy, lv_xy), (kf_yt, lv_yt), (kf_xt, lv_xt)):
            if kf.numel() != L.levels_n_params(lv):
                raise RuntimeError("keyframe params have the wrong length for their encoding_config")
        d = sum(lv.n_levels * lv.n_features for lv in (lv_xy, lv_yt, lv_xt)) + 9 * sh.n_features
        _check_mlp(mlp, d)
        rows = lib.nvp_latent_rows(d)
        dev = coords.device
        need_grad = bool(grad_mode) and any(ctx.needs_input_grad)
        if y_sorted and CHECK_SORTED and n > 1 and not bool((coords[1:, 2] >= coords[:-1, 2]).all()):
            raise RuntimeError("model_input['sorted_by_y'] is set but all_coords[..., 2] is not non-decreasing "
                               "(the gradient scatter would produce wrong keyframe gradients)")
        order = None
        if need_grad and not y_sorted and not temporal_interp and AUTO_SORT_MIN > 0 and n >= AUTO_SORT_MIN:
            order = _row_order(lib, coords, n, lv_xy, lv_yt)
            coords, steps, y_sorted = coords[order], steps[order], True
        # (SparseGrid.forward_inter - eval.py --t_interp - has inference kernels only: under grad mode it takes the two-kernel path and raises below)
        sup = int(lib.nvp_encode_mlp_fwd_supported(C.byref(lv_xy), C.byref(lv_yt), C.byref(lv_xt), C.byref(sh))) if (bool(n) and FUSED_FWD and not (temporal_interp and need_grad)) else 0
        fused = sup > 0
        fused_parks_rows = sup == 2        # config_nvp_l-sized latents: the rows beyond the wave's LDS tile pass through the latent tensor even without a backward pass
        # the latent tensor: an intermediate of the two-kernel path; with the fused forward it only exists for the backward pass (or as that workspace)
        zt = torch.empty((L.ntiles(n), rows, L.TILE), device=dev, dtype=torch.float32) if (need_grad or not fused or fused_parks_rows) else None
        # Scatter workspace, allocated here when a backward pass will follow: (a) for y-sorted batches the backward chain writes the
        # xy / yt planes' latent gradients straight into the scatter's level-major buffers, (b) everything the scatter derives from
        # the COORDINATES alone (sort keys, orders, the sparse row table: a dozen small latency-bound kernels, ~0.26 ms back to
        # back) is started NOW on a side stream, underneath the gather kernel, and the scatter later only waits for its event.
        # The side stream also packs the MLP weights (forward layout now, backward layout for later): ~0.1 ms of small kernels that
        # depend on the parameters only.
        ctx.ws = ctx.presorted = ctx.packed_bwd = ctx.chain = None
        packed_fwd = None
        bwd_follows = need_grad and n and not temporal_interp
        # the tile-fused training step (forward + loss gradient + backward chain in one launch): the caller handed the ground truth over
        tile_fused = bool(TILE_FUSED and hooks is not None and hooks.loss_gt is not None and fused and bwd_follows and y_sorted and order is None
                          and SIDE_WORK and DZ_LEVEL_MAJOR and lib.nvp_dz_lm_supported(d) and L.has_entry("nvp_encode_mlp_fwd_bwd")
                          and lib.nvp_encode_mlp_fwd_bwd_supported(C.byref(lv_xy), C.byref(lv_yt), C.byref(lv_xt), C.byref(sh)))
        if tile_fused:
            gt_u8 = hooks.loss_gt.reshape(-1, 3)
            if gt_u8.dtype != torch.uint8 or gt_u8.shape[0] != n or not gt_u8.is_contiguous() or gt_u8.device != dev:
                tile_fused = False
        # the side stream is only worth its hand-over (two event markers on the compute queue + one deferred-free event per buffer)
        # when something runs on it: the scatter's presort / backward pack (a backward pass follows) or the forward pack underneath
        # the gather kernel of the two-kernel path.  Fused inference (eval: 100 slices per frame) packs in line, below.
        inf_cache = hooks.packed_cache if (hooks is not None and not need_grad) else None      # inference: one weight pack per evaluation
        if n and SIDE_WORK and (bwd_follows or (not fused and inf_cache is None)):
            L.ptr(coords)                      # CPU tensors are refused here, before any stream is touched (no CPU path)
            side = _side_stream(dev)
            pstruct = L.mlp_params_struct(mlp)
            pk_f = torch.empty(lib.nvp_packed_fwd_floats(d), device=dev, dtype=torch.float32)
            pk_b = torch.empty(lib.nvp_packed_bwd_floats(d), device=dev, dtype=torch.float32) if bwd_follows else None
            if bwd_follows:                    # the scatter workspace too, BEFORE the one hand-over to the side stream (every
                ws_bytes = _scatter_ws_bytes(lib, n, lv_xy, lv_yt, lv_xt, sh)      # wait_stream is an event marker on the compute queue)
                ctx.ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
            side.wait_stream(torch.cuda.current_stream(dev))     # allocations, parameters and coordinates are ordered on the compute stream
            # Lifetimes: these buffers come from the COMPUTE stream's allocator pool but are written (and the coordinates /
            # parameters read) on the side stream.  If the autograd graph is dropped without a backward pass (a validation
            # loss under grad mode, an exception between forward and backward), nothing ever makes the compute stream wait
            # for the side stream again - record_stream keeps the allocator from handing the blocks out while the side
            # stream may still be using them.
            pk_f.record_stream(side)
            coords.record_stream(side)
            for t in mlp:
                t.record_stream(side)
            if pk_b is not None:
                pk_b.record_stream(side)
            if fused:
                # one-launch forward: nothing can run underneath this pack (the forward kernel needs it, and it needs the optimizer's
                # last update) - on the side stream it would only add two cross-queue hand-overs to the step boundary
                L.check(lib.nvp_mlp_pack_fwd(C.byref(pstruct), L.ptr(pk_f), d, L.stream_ptr()), "nvp_mlp_pack_fwd")
                packed_fwd = (pk_f, None)
            else:
                with torch.cuda.stream(side):                  # two-kernel forward: underneath the gather kernel
                    L.check(lib.nvp_mlp_pack_fwd(C.byref(pstruct), L.ptr(pk_f), d, L.stream_ptr()), "nvp_mlp_pack_fwd")
                    ev = torch.cuda.Event()
                    ev.record()
                    packed_fwd = (pk_f, ev)
        if bwd_follows:
            L.ptr(coords)
            lvs = (lv_xy, lv_yt, lv_xt)
            if ctx.ws is None:
                ws_bytes = _scatter_ws_bytes(lib, n, lv_xy, lv_yt, lv_xt, sh)
                ctx.ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
            ws_bytes = ctx.ws.numel()
            bflags = L.COORDS_SORTED_BY_Y if y_sorted else 0
            if DZ_LEVEL_MAJOR and y_sorted and lib.nvp_dz_lm_supported(d):
                bflags |= L.DZ_PLANES_READY
            ctx.bflags = bflags
            if SIDE_WORK:                      # (bwd_follows: the hand-over above has run, `side` / `pstruct` / `pk_b` exist)
                ctx.ws.record_stream(side)
                with torch.cuda.stream(side):
                    if tile_fused:             # the fused step kernel needs the backward pack up front: it goes first on the side stream
                        L.check(lib.nvp_mlp_pack_bwd(C.byref(pstruct), L.ptr(pk_b), d, L.stream_ptr()), "nvp_mlp_pack_bwd")
                        ev = torch.cuda.Event()
                        ev.record()
                        ctx.packed_bwd = (pk_b, ev)
                    L.check(lib.nvp_encode_bwd_presort(L.ptr(coords), n, C.byref(lvs[0]), C.byref(lvs[1]), C.byref(lvs[2]), C.byref(sh),
                                                       L.ptr(ctx.ws, torch.uint8), ws_bytes, bflags, L.stream_ptr()), "nvp_encode_bwd_presort")
                    ctx.presorted = torch.cuda.Event()
                    ctx.presorted.record()
                    if not tile_fused:
                        L.check(lib.nvp_mlp_pack_bwd(C.byref(pstruct), L.ptr(pk_b), d, L.stream_ptr()), "nvp_mlp_pack_bwd")
                        ev = torch.cuda.Event()
                        ev.record()
                        ctx.packed_bwd = (pk_b, ev)
        if tile_fused:
            # forward + image_mse gradient + backward chain, tile by tile, in one launch
            pstruct = L.mlp_params_struct(mlp)
            packed, _ = packed_fwd
            pk_b, ev_b = ctx.packed_bwd
            torch.cuda.current_stream(dev).wait_event(ev_b)
            ws_bytes = ctx.ws.numel()
            lm = L.ScatterLm()
            L.check(lib.nvp_encode_bwd_prepare(n, C.byref(lv_xy), C.byref(lv_yt), C.byref(lv_xt), C.byref(sh),
                                               L.ptr(ctx.ws, torch.uint8), ws_bytes, C.byref(lm), L.stream_ptr()), "nvp_encode_bwd_prepare")
            nt = L.ntiles(n)
            rgb = torch.empty((n, 3), device=dev, dtype=torch.float32)
            saved = torch.empty((5, nt, L.HIDDEN, L.TILE), device=dev, dtype=torch.float32)
            dy = torch.empty((6, nt, L.HIDDEN, L.TILE), device=dev, dtype=torch.float32)
            dz_rows = torch.empty((nt * L.TILE, lib.nvp_dz_stride(d)), device=dev, dtype=torch.float32)
            L.check(_call("nvp_encode_mlp_fwd_bwd", lib.nvp_encode_mlp_fwd_bwd, L.ptr(coords), L.ptr(steps), L.ptr(gt_u8, torch.uint8), L.ptr(kf_xy), L.ptr(kf_yt),
                          L.ptr(kf_xt), L.ptr(emb), C.byref(pstruct), L.ptr(packed), L.ptr(pk_b), L.ptr(rgb), L.ptr(saved), L.ptr(zt), L.ptr(dy), L.ptr(dz_rows),
                          C.byref(lm), n, C.byref(lv_xy), C.byref(lv_yt), C.byref(lv_xt), C.byref(sh), L.stream_ptr()), "nvp_encode_mlp_fwd_bwd")
            ctx.chain = (dy, dz_rows, lm, gt_u8)
            ctx.packed_bwd = None
        elif fused:
            # one launch: every wave of the forward MLP gathers its own latent tile into LDS; the latent tensor is written only
            # when a backward pass will read it (dW GEMMs)
            pstruct = L.mlp_params_struct(mlp)
            if packed_fwd is not None:
                packed, ev = packed_fwd
                if ev is not None:
                    torch.cuda.current_stream(dev).wait_event(ev)
            else:
                packed = _packed_fwd_cached(lib, inf_cache, pstruct, d, dev)
            rgb = torch.empty((n, 3), device=dev, dtype=torch.float32)
            saved = torch.empty((5, L.ntiles(n), L.HIDDEN, L.TILE), device=dev, dtype=torch.float32) if need_grad else None
            L.check(_call("nvp_encode_mlp_fwd", lib.nvp_encode_mlp_fwd, L.ptr(coords), L.ptr(steps), L.ptr(kf_xy), L.ptr(kf_yt), L.ptr(kf_xt), L.ptr(emb),
                          C.byref(pstruct), L.ptr(packed), L.ptr(rgb), L.ptr(saved), L.ptr(zt) if (need_grad or fused_parks_rows) else None, n,
                          C.byref(lv_xy), C.byref(lv_yt), C.byref(lv_xt), C.byref(sh), 1 if temporal_interp else 0, L.stream_ptr()), "nvp_encode_mlp_fwd")
        else:
            if n:
                L.check(_call("nvp_encode_fwd", lib.nvp_encode_fwd, L.ptr(coords), L.ptr(kf_xy), L.ptr(kf_yt), L.ptr(kf_xt), L.ptr(emb), L.ptr(zt), n,
                                           C.byref(lv_xy), C.byref(lv_yt), C.byref(lv_xt), C.byref(sh),
                                           1 if temporal_interp else 0, L.COORDS_SORTED_BY_Y if y_sorted else 0, L.stream_ptr()), "nvp_encode_fwd")
            if packed_fwd is None and inf_cache is not None and n:
                packed_fwd = (_packed_fwd_cached(lib, inf_cache, L.mlp_params_struct(mlp), d, dev), None)
            rgb, saved = _mlp_forward(zt, steps, mlp, n, d, save=need_grad, packed=packed_fwd)
        if need_grad:
            if temporal_interp:
                raise NotImplementedError("temporal_interp=True is an inference-only path (reference eval.py --t_interp)")
            ctx.n, ctx.d = n, d
            ctx.hooks = hooks
            ctx.order = order
            ctx.flags = L.COORDS_SORTED_BY_Y if y_sorted else 0
            ctx.lv = (lv_xy, lv_yt, lv_xt)
            ctx.sh = sh
            if saved is None:
                saved = torch.empty(0, device=dev)
            ctx.save_for_backward(coords, steps, zt, saved, kf_xy, kf_yt, kf_xt, emb, *mlp)
        if order is not None:                       # rows back in the caller's order
            out = torch.empty_like(rgb)
            out[order] = rgb
            return out
        return rgb

    @staticmethod
    def backward(ctx, drgb):
        lib = L.load()
        coords, steps, zt, saved, kf_xy, kf_yt, kf_xt, emb, *mlp = ctx.saved_tensors
        n, d = ctx.n, ctx.d
        if ctx.order is not None:
            drgb = drgb[ctx.order]
        if n == 0:
            z = [torch.zeros_like(t) for t in (kf_xy, kf_yt, kf_xt, emb)]
            return (None, None, *z, None, None, None, None, None, None, None, *[torch.zeros_like(t) for t in mlp])
        lv = ctx.lv
        # every gradient element is written exactly once by the sorted-band scatter: no zero-fill
        hk = ctx.hooks if ctx.hooks is not None else _NO_HOOKS
        sink = hk.grad_sink
        # one GPU: the grids' optimizer steps inside the scatter's flushes (no gradient tensors for them at all)
        fused_st, fused_dn = None, None
        split_ok = bool(ctx.bflags & L.DZ_PLANES_READY)
        if hk.fused_sparse is not None and sink is None and split_ok and ctx.needs_input_grad[5]:       # [5]: emb
            fused_st = hk.fused_sparse.fused_peek(emb)
        if hk.fused_dense is not None and sink is None and split_ok and all(ctx.needs_input_grad[2:5]):  # [2:5]: the three planes
            fused_dn = [hk.fused_dense.fused_peek(t) for t in (kf_xy, kf_yt, kf_xt)]
            if any(f is None for f in fused_dn) or len({(f["lr"], f["beta1"], f["beta2"], f["eps"], f["weight_decay"]) for f in fused_dn}) != 1:
                fused_dn = None             # the kernel takes one set of hyper-parameters for the three planes
        d_xy = d_yt = d_xt = None
        if fused_dn is None:
            d_xy, d_yt, d_xt = (_grad_buffer(t, sink) for t in (kf_xy, kf_yt, kf_xt))
        d_emb = _grad_buffer(emb, sink) if fused_st is None else None

        # scatter workspace: allocated (and its coordinate-only part started) in forward
        ws, flags, presorted, lm = ctx.ws, ctx.bflags, ctx.presorted, None
        ctx.ws = ctx.presorted = None
        if ws is None:          # a second backward through the same graph (retain_graph): fresh workspace, nothing presorted
            ws_bytes = lib.nvp_encode_bwd_workspace_bytes(n, C.byref(lv[0]), C.byref(lv[1]), C.byref(lv[2]), C.byref(ctx.sh))
            if ws_bytes < 0:
                raise L.NvpHipError("nvp_encode_bwd_workspace_bytes failed")
            ws = torch.empty(ws_bytes, device=coords.device, dtype=torch.uint8)
        ws_bytes = ws.numel()
        chain, ctx.chain = ctx.chain, None          # the tile-fused step kernel already ran the backward chain in forward
        if chain is not None:
            lm = chain[2]
        elif flags & L.DZ_PLANES_READY:
            lm = L.ScatterLm()
            L.check(lib.nvp_encode_bwd_prepare(n, C.byref(lv[0]), C.byref(lv[1]), C.byref(lv[2]), C.byref(ctx.sh),
                                               L.ptr(ws, torch.uint8), ws_bytes, C.byref(lm), L.stream_ptr()), "nvp_encode_bwd_prepare")
        if presorted is not None:
            flags |= L.SCATTER_PRESORTED

        def scatter_call(fl):
            L.check(_call("nvp_encode_bwd", lib.nvp_encode_bwd, L.ptr(coords), L.ptr(dz_rows_ref[0]), dz_rows_ref[0].shape[1],
                          L.ptr(d_xy), L.ptr(d_yt), L.ptr(d_xt), L.ptr(d_emb), n,
                          C.byref(lv[0]), C.byref(lv[1]), C.byref(lv[2]), C.byref(ctx.sh),
                          L.ptr(ws, torch.uint8), ws_bytes, fl, L.stream_ptr()),
                    "nvp_encode_bwd")

        dz_rows_ref = [None]
        packed_bwd, ctx.packed_bwd = ctx.packed_bwd, None

        def scatter(dz_rows):
            dz_rows_ref[0] = dz_rows
            # the presort's event was recorded on the side stream BEFORE the backward pack's, which the compute stream has waited
            # for in front of the chain kernel: a second wait would only be one more marker on the compute queue
            if presorted is not None and packed_bwd is None:
                torch.cuda.current_stream(coords.device).wait_event(presorted)
            nonlocal d_emb, fused_st, fused_dn, d_xy, d_yt, d_xt
            if not (split_ok and (fused_st is not None or fused_dn is not None or hk.sparse_ready is not None or hk.early_grads is not None)):
                scatter_call(flags)         # one call: sparse grid and planes
                if hk.early_grads is not None:
                    hk.early_grads([(emb, d_emb), (kf_xy, d_xy), (kf_yt, d_yt), (kf_xt, d_xt)])
            else:
                # split scatter.  The sparse grid (80 % of the gradient bytes) first: fused with its AdamW step, or scattered and handed
                # on - to the data-parallel exchange, or to the optimizer - while the dense planes are still being scattered
                if fused_st is not None:
                    f = fused_st
                    rc = _call("nvp_encode_bwd", lib.nvp_encode_bwd_sparse_adamw, L.ptr(coords), L.ptr(dz_rows), dz_rows.shape[1], n,
                               C.byref(lv[0]), C.byref(lv[1]), C.byref(lv[2]), C.byref(ctx.sh), L.ptr(ws, torch.uint8), ws_bytes, flags,
                               L.ptr(emb), L.ptr(f["exp_avg"]), L.ptr(f["exp_avg_sq"]), f["lr"], f["beta1"], f["beta2"], f["eps"],
                               f["weight_decay"], f["step"], L.stream_ptr())
                    if rc == L.ERR_UNSUPPORTED:             # layout the fused flush does not take (nothing enqueued): the gradient route after all
                        fused_st = None
                        d_emb = _grad_buffer(emb, sink)
                    else:
                        L.check(rc, "nvp_encode_bwd_sparse_adamw")
                        hk.fused_sparse.fused_commit(emb)
                if fused_st is None:
                    scatter_call(flags | L.SCATTER_SPARSE_ONLY)
                    if hk.sparse_ready is not None:
                        hk.sparse_ready()
                    if hk.early_grads is not None:
                        hk.early_grads([(emb, d_emb)])
                # then the three dense planes (the sparse call above ran the presort if nobody had)
                if fused_dn is not None:
                    f0 = fused_dn[0]
                    arr = lambda key: (C.c_void_p * 3)(*[L.ptr(f[key]) for f in fused_dn])      # noqa: E731
                    steps3 = (C.c_int64 * 3)(*[f["step"] for f in fused_dn])
                    rc = _call("nvp_encode_bwd", lib.nvp_encode_bwd_dense_adamw, L.ptr(coords), L.ptr(dz_rows), dz_rows.shape[1], n,
                               C.byref(lv[0]), C.byref(lv[1]), C.byref(lv[2]), C.byref(ctx.sh), L.ptr(ws, torch.uint8), ws_bytes,
                               flags | L.SCATTER_PRESORTED, arr("param"), arr("exp_avg"), arr("exp_avg_sq"), f0["lr"], f0["beta1"], f0["beta2"],
                               f0["eps"], f0["weight_decay"], steps3, L.stream_ptr())
                    if rc == L.ERR_UNSUPPORTED:
                        fused_dn = None
                        d_xy, d_yt, d_xt = (_grad_buffer(t, sink) for t in (kf_xy, kf_yt, kf_xt))
                    else:
                        L.check(rc, "nvp_encode_bwd_dense_adamw")
                        for t in (kf_xy, kf_yt, kf_xt):
                            hk.fused_dense.fused_commit(t)
                if fused_dn is None:
                    scatter_call(flags | L.SCATTER_DENSE_ONLY | L.SCATTER_PRESORTED)
                    if hk.early_grads is not None:
                        hk.early_grads([(kf_xy, d_xy), (kf_yt, d_yt), (kf_xt, d_xt)])
            if hk.grids_ready is not None:
                hk.grids_ready()            # e.g. start the (async) all-reduce of the grid gradients

        # order: dX chain -> grid scatter (needs only dz) -> dW GEMMs (independent of the scatter)
        _, grads = _mlp_backward(drgb, steps, zt, saved, mlp, n, d, between=scatter, lm=lm, packed=packed_bwd, sink=sink,
                                 chain=None if chain is None else chain[:2])
        return (None, None, d_xy, d_yt, d_xt, d_emb, None, None, None, None, None, None, None, *grads)
