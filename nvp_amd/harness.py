"""Own counterpart of the reference's training harness: the sampler of
dataio.py:78-120 with the u8 video resident in HBM, image_mse (loss_functions.py:1-3 after
training.py:47-48) fused with its gradient, and one optimisation step in the reference's
order (training.py:13-14,50-76): forward, loss, zero_grad, backward, step, sched.step.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L

N_SAMPLES = 1245184          # reference dataio.py:91

# Optional callable fired inside train_step right after loss.backward() has returned (everything of backward is enqueued,
# nothing of the gradient exchange / optimizer yet): bench.py records a HIP event there to time the exposed exchange.
AFTER_BACKWARD_HOOK = None
# one GPU, nvp_amd.optim.AdamW: update the grids on a side stream as soon as the scatter has produced their gradients (0: after backward)
# NVP_FUSED_ADAMW=0: ONE kill switch for every update that happens before step() (early side-stream update, sparse and dense flushes) -
# for loops that clip / accumulate gradients, read the grids' .grad, skip steps or recover from failed ones (nvp_amd/optim.py)
_ALL_FUSED = os.environ.get("NVP_FUSED_ADAMW", "1") != "0"
EARLY_ADAMW = _ALL_FUSED and os.environ.get("NVP_EARLY_ADAMW", "1") != "0"
# one GPU, nvp_amd.optim.AdamW, y-sorted batches: the sparse grid's AdamW step is applied by the scatter kernel's flush
# (nvp_encode_bwd_sparse_adamw; bit-identical parameters; 0: gradient tensor + early_update)
FUSED_SPARSE_ADAMW = _ALL_FUSED and os.environ.get("NVP_FUSED_SPARSE_ADAMW", "1") != "0"
# ... and the three dense planes' by band_kernel / slab_reduce_kernel (nvp_encode_bwd_dense_adamw; bit-identical; 0: gradient tensors + early_update)
FUSED_DENSE_ADAMW = _ALL_FUSED and os.environ.get("NVP_FUSED_DENSE_ADAMW", "1") != "0"
SAMPLER_SORT = os.environ.get("NVP_SAMPLER_SORT", "nvp")        # "torch": torch.argsort for the sampler's column order
RECORD_STREAM = os.environ.get("NVP_SAMPLER_RECORD_STREAM", "0") == "1"     # DeviceVideo(prefetch=True): record_stream on every batch tensor


class ImageMSEU8(torch.autograd.Function):
    """mean((out - (gt_u8-127.5)/127.5)^2) with d/dout produced in the same pass (R13)."""

    @staticmethod
    def forward(ctx, out: torch.Tensor, gt_u8: torch.Tensor) -> torch.Tensor:
        lib = L.load()
        o = out.contiguous()
        n = o.numel() // 3
        if gt_u8.dtype != torch.uint8 or gt_u8.numel() != o.numel():
            raise RuntimeError("gt must be uint8 with the same number of elements as the model output")
        gt_u8 = gt_u8.contiguous()
        loss_sum = torch.zeros(1, device=o.device, dtype=torch.float32)
        drgb = torch.empty_like(o) if ctx.needs_input_grad[0] else None
        if n:
            L.check(lib.nvp_mse_u8(L.ptr(o), L.ptr(gt_u8, torch.uint8), L.ptr(drgb), L.ptr(loss_sum), n, L.stream_ptr()), "nvp_mse_u8")
        ctx.save_for_backward(drgb if drgb is not None else torch.empty(0, device=o.device))
        return (loss_sum / max(o.numel(), 1)).reshape(())

    @staticmethod
    def backward(ctx, g):
        (drgb,) = ctx.saved_tensors
        # train_step seeds backward with a cached device scalar 1.0 (unit_gradient): multiplying by it would be an exact no-op that
        # costs a fill, a 30-MB element-wise pass and two launch boundaries per step
        if g.dim() == 0 and g.data_ptr() == _UNIT.get(g.device, (None, 0))[1]:
            return drgb.detach(), None          # a fresh alias: an in-place op on the returned gradient cannot reach the saved tensor's version counter
        return drgb * g, None


_UNIT = {}          # device -> (the scalar tensor 1.0 on it, its data_ptr)


def unit_gradient(dev: torch.device) -> torch.Tensor:
    """A cached fp32 scalar 1.0 on `dev`: `loss.backward(gradient=unit_gradient(dev))` is `loss.backward()` without the per-step
    ones_like fill, and lets ImageMSEU8.backward recognise the seed (by address) and skip the multiplication by one."""
    ent = _UNIT.get(dev)
    if ent is None:
        t = torch.ones((), device=dev, dtype=torch.float32)
        ent = _UNIT[dev] = (t, t.data_ptr())
    return ent[0]          # NEVER modify it in place (it is recognised by address, not by value): use it as a backward seed only


def image_mse_u8(model_out: torch.Tensor, gt_u8: torch.Tensor) -> torch.Tensor:
    return ImageMSEU8.apply(model_out, gt_u8)


class DeviceVideo:
    """u8 video [T, H, W, 3] kept on the device + the reference's per-step random sampler."""

    def __init__(self, video_u8: torch.Tensor, n_samples: int = N_SAMPLES, seed: int = 0, sort_by_y: bool = True, prefetch: bool = False):
        if video_u8.dtype != torch.uint8 or video_u8.dim() != 4 or video_u8.shape[-1] != 3:
            raise ValueError("video must be uint8 [T, H, W, 3]")
        self.video = video_u8.contiguous()
        self.T, self.H, self.W = (int(v) for v in video_u8.shape[:3])
        self.n = int(n_samples)
        self.sort_by_y = bool(sort_by_y)
        dev = video_u8.device
        half_dt = 0.5 / self.T
        # dataio.py:93-99: modulation input and temporal coordinate tables
        self.tstep_tab = torch.linspace(half_dt, 1 - half_dt, self.T).to(dev)
        self.tcoord_tab = torch.linspace(0, 1, self.T).to(dev)
        self.gen = torch.Generator(device=dev).manual_seed(seed)
        # prefetch: the NEXT batch is drawn on a side stream while the current step's kernels run (the sampler is a chain of
        # small latency-bound kernels - two randint, a 16-bit argsort, a 3-byte gather - that depends on nothing the step
        # computes); same draws, same order of batches as without it.  Worth 0.05 ms of a 7.5-ms step on MI355X (the sampler's
        # ~0.2 ms of small kernels hide underneath the gather / MLP kernels, which slow down by 0.06 ms): bench.py and train.py turn it on
        self._side = L.side_stream(dev) if (prefetch and video_u8.is_cuda) else None
        self._next = None

    def sample(self) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
        if self._side is None:
            return self._draw()
        cur = torch.cuda.current_stream()
        if self._next is None:
            self._issue()
        batch, ev = self._next
        cur.wait_event(ev)
        if RECORD_STREAM:                   # opt-in: consumers on OTHER streams than the one current here (see the contract below)
            for d_ in batch:
                for t_ in d_.values():
                    if torch.is_tensor(t_):
                        t_.record_stream(cur)
        # CONTRACT (single consumer stream): a prefetched batch must be consumed on the stream that is current at this call, and
        # the next sample() must be issued from that same stream - then the block of a freed batch can only be reused by a draw
        # that was enqueued behind everything that read it.  A caller that reads a batch on another stream (its own copy / eval
        # stream) sets NVP_SAMPLER_RECORD_STREAM=1, or passes prefetch=False.
        # The batch tensors come from the SIDE stream's allocator pool and are consumed on this stream.  No record_stream (each
        # would cost an event marker on the compute queue when the tensor is freed - ~11 us of queue time apiece): a freed batch
        # can only be handed out again by a later _draw, every _draw is enqueued behind `side.wait_stream(compute stream)` in
        # _issue, and whoever freed the batch had enqueued its last use before that.
        self._issue()
        return batch

    def _issue(self) -> None:
        # the video and the coordinate tables may still be being produced on the caller's stream (procedural_video, a H2D copy)
        self._side.wait_stream(torch.cuda.current_stream(self.video.device))
        with torch.cuda.stream(self._side):
            batch = self._draw()
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._next = (batch, ev)

    def _draw(self) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
        """-> ({'all_coords': [1,N,3], 'temporal_steps': [1,N]}, {'img': uint8 [1,N,3]})
        randint order as dataio.py:106-107 (temporal indices first, then spatial)."""
        lib = L.load()
        dev = self.video.device
        n = self.n
        ti = torch.randint(0, self.T, (n,), device=dev, generator=self.gen)
        pi = torch.randint(0, self.H * self.W, (n,), device=dev, generator=self.gen)
        order = None
        if self.sort_by_y:
            # Same i.i.d. draws as the reference, delivered in ascending-y (= image column) order.  The MSE
            # is permutation-invariant; the order gives the grid gathers row locality and lets the
            # gradient scatter skip one sort (NVP_COORDS_SORTED_BY_Y).  The gather kernel applies the permutation.
            # 16-bit keys halve the radix passes; wider frames than 32 767 columns keep 32-bit keys (int16 would wrap)
            # a stable counting sort on the column key (nvp_sample_order_by_column: three small launches; the vendor sort needed six);
            # NVP_SAMPLER_SORT=torch keeps torch.argsort (same order: its radix sort is stable too)
            order = None
            if SAMPLER_SORT != "torch":
                ws_bytes = lib.nvp_sample_order_workspace_bytes(n, self.W)
                ws = torch.empty(max(int(ws_bytes), 4), device=dev, dtype=torch.uint8)
                order = torch.empty(n, device=dev, dtype=torch.int64)
                rc = lib.nvp_sample_order_by_column(L.ptr(pi, torch.int64), L.ptr(order, torch.int64), n, self.W, L.ptr(ws, torch.uint8), ws.numel(), L.stream_ptr())
                if rc == L.ERR_UNSUPPORTED:
                    order = None
                else:
                    L.check(rc, "nvp_sample_order_by_column")
            if order is None:
                order = torch.argsort((pi % self.W).to(torch.int16 if self.W <= 32767 else torch.int32), stable=True)
        coords = torch.empty((n, 3), device=dev, dtype=torch.float32)
        steps = torch.empty((n,), device=dev, dtype=torch.float32)
        gt = torch.empty((n, 3), device=dev, dtype=torch.uint8)
        L.check(lib.nvp_sample_gather(L.ptr(self.video, torch.uint8), L.ptr(ti, torch.int64), L.ptr(pi, torch.int64),
                                      L.ptr(order, torch.int64) if order is not None else None, L.ptr(self.tcoord_tab), L.ptr(self.tstep_tab), L.ptr(coords), L.ptr(steps),
                                      L.ptr(gt, torch.uint8), n, self.T, self.H, self.W, L.stream_ptr()), "nvp_sample_gather")
        mi = {"all_coords": coords.unsqueeze(0), "temporal_steps": steps.unsqueeze(0)}
        if self.sort_by_y:
            mi["sorted_by_y"] = True
        return (mi, {"img": gt.unsqueeze(0)})

    def frame_batch(self, frame: int, lo: int, hi: int):
        """Whole-frame evaluation slice (eval.py:219-239): pixels [lo, hi) of one frame."""
        dev = self.video.device
        p = torch.arange(lo, hi, device=dev)
        row = torch.div(p, self.W, rounding_mode="floor").float() / (self.H - 1)
        col = (p % self.W).float() / (self.W - 1)
        t = self.tcoord_tab[frame].expand(hi - lo)
        coords = torch.stack((t, row, col), dim=1).unsqueeze(0)
        steps = self.tstep_tab[frame].expand(1, hi - lo)
        return {"all_coords": coords, "temporal_steps": steps}, self.video[frame].reshape(-1, 3)[lo:hi]


def make_optimizer(model: torch.nn.Module, total_steps: int, lr: float = 1e-2):
    """AdamW(lr, weight_decay=1e-3) + CosineAnnealingLR(T_max=steps, eta_min=1e-5): training.py:13-14."""
    # same update rule as the reference's torch.optim.AdamW.  On a HIP device the whole step is one launch of
    # nvp_adamw_step (K13: 3.8 GB/step, one pass); CPU modules (host-logic tests only) get torch's own AdamW.
    params = list(model.parameters())
    if params and all(p.is_cuda for p in params):
        from .optim import AdamW
        opt = AdamW(params, lr=lr, weight_decay=0.001)
    else:
        opt = torch.optim.AdamW(lr=lr, params=params, weight_decay=0.001)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=total_steps, eta_min=1e-5)
    return opt, sched


def train_step(model, opt, sched, model_input, gt, bucket=None) -> torch.Tensor:
    """One iteration in the reference's order (training.py:50-76). Returns the (device) loss.
    `bucket` (parallel.GradBucket) turns on data parallelism: backward writes the gradients into one flat buffer whose
    grid range is exchanged early and asynchronously; the optimizer decides how:
      * parallel.ShardedAdamW - reduce-scatter, AdamW on the own shard, all-gather of the parameters (ZeRO-1);
      * optim.AdamW           - all-reduce in pieces, every rank updates everything (1/world inside the kernel);
      * anything else         - one blocking all-reduce + mean, then opt.step()."""
    from . import functional
    from .optim import AdamW
    from .parallel import ShardedAdamW
    sharded = isinstance(opt, ShardedAdamW)
    if sharded:
        bucket = opt.bucket
    hooks = functional.StepHooks()                  # this step's own hooks: read by this call's backward only (re-entrant)
    if functional.TILE_FUSED and gt["img"].dtype == torch.uint8:
        hooks.loss_gt = gt["img"]                   # the loss below IS image_mse on this ground truth, seeded with 1: forward may run the chain too
    out = model(dict(model_input, nvp_hooks=hooks))["model_out"]
    loss = image_mse_u8(out, gt["img"])
    if bucket is not None:
        bucket.detach_grads()                       # == zero_grad(set_to_none=True)
        hooks.grad_sink = bucket.sink()             # backward writes straight into the flat buffer
        hooks.grids_ready = opt.start_early if sharded else bucket.start_early   # grid grads exchanged underneath the dW GEMMs
        hooks.sparse_ready = opt.start_first if sharded else None                 # ... the sparse grid's even earlier
    else:
        opt.zero_grad()
        if EARLY_ADAMW and isinstance(opt, AdamW):
            opt.begin_step()                        # forget early updates of an iteration whose step() never ran (exception)
            hooks.early_grads = opt.early_update    # the grids' AdamW underneath the rest of backward (one GPU)
            if FUSED_SPARSE_ADAMW:
                hooks.fused_sparse = opt            # ... and the sparse grid's INSIDE the scatter's flush: its gradient never reaches HBM
            if FUSED_DENSE_ADAMW:
                hooks.fused_dense = opt             # ... the three planes' too: no optimizer launch is left for the grids
    try:
        loss.backward(gradient=unit_gradient(loss.device) if loss.is_cuda else None)
    finally:
        hooks.clear()
    if AFTER_BACKWARD_HOOK is not None:
        AFTER_BACKWARD_HOOK()
    if sharded:
        opt.step()
    elif bucket is not None and isinstance(opt, AdamW):
        # SUM all-reduce in pieces (the grid pieces are already in flight); AdamW updates each piece as its collective
        # completes, 1/world applied inside the kernel
        opt.step(grad_scale=1.0 / bucket.world_size(), schedule=bucket.step_schedule())
    else:
        if bucket is not None:
            bucket.all_reduce_mean()                # copies back only if some grad did not land in the bucket
        opt.step()
    sched.step()
    return loss.detach()


def make_dp(model: torch.nn.Module, total_steps: int, mode: str = "sharded", algo: str = "reduce_scatter", lr: float = 1e-2,
            early: bool = True):
    """Data-parallel optimizer set-up: (opt, sched, bucket).  mode "sharded" = ZeRO-1 (parallel.ShardedAdamW), "replicated" =
    all-reduce + full AdamW on every rank (optim.AdamW with the chunked schedule).  Hyper-parameters: training.py:13-14."""
    from . import parallel
    from .optim import AdamW
    params = parallel.unique_parameters(model)
    grids = [model.keyframes_xy.params, model.keyframes_yt.params, model.keyframes_xt.params, model.sparse_grid.embeddings] if early else None
    if mode == "sharded":
        world = parallel.GradBucket.world_size()
        bucket = parallel.GradBucket(params, early=grids, pad_to=parallel.ShardedAdamW.alignment(world))
        opt = parallel.ShardedAdamW(bucket, lr=lr, weight_decay=0.001, algo=algo, first=[model.sparse_grid.embeddings] if early else None)
    elif mode == "replicated":
        bucket = parallel.GradBucket(params, early=grids)
        opt = AdamW(params, lr=lr, weight_decay=0.001)
    else:
        raise ValueError("mode must be 'sharded' or 'replicated'")
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=total_steps, eta_min=1e-5)
    return opt, sched, bucket


def train_psnr(loss: torch.Tensor) -> float:
    """10*log10(4/mse): signal range 2 (training.py:58)."""
    return 10.0 * math.log10(4.0 / float(loss))


_LATTICE = {}          # (device, H', W') -> [H'W', 3] float32 with columns 1, 2 = (row, col) lattice coordinates; column 0 is the frame's t
_LATTICE_MAX_BYTES = 1 << 30       # cached lattices are evicted (oldest first) beyond this: a 4K x --s_interp 2 lattice alone is 400 MB
# pixels per model call when a call MATERIALISES per-pixel state (the two-kernel forward of nvp_l-sized latents and forward_inter write the
# latent [n, D] and the MLP's buffers): what the reference's 100-slice loop (eval.py:233-239) exists to bound.  The fused no-grad forward
# (functional.fused_forward_supported) materialises nothing and takes the frame in one call.
MAX_PIXELS_PER_CALL = int(os.environ.get("NVP_EVAL_MAX_PIXELS", str(1 << 22)))


@torch.no_grad()
def render_frame(model, f: int, org_nframes: int, resolution, nframes: Optional[int] = None, temporal_interp: bool = False,
                 n_slice: int = 100, hooks=None, literal_slices: bool = False) -> torch.Tensor:
    """One output frame of the reference's inference loop (eval.py:219-245): [H', W', 3] on the device, in [0, 1].

    resolution = (H', W') of the QUERY lattice (eval.py:201-204: the video's resolution times --s_interp);
    nframes = number of output frames (eval.py:205-210: the video's frame count times --t_interp);
    org_nframes = the video's own frame count, which fixes half_dt (eval.py:224).
    Frame f: temporal coordinate linspace(0,1,nframes)[f], modulation input linspace(half_dt, 1-half_dt, nframes)[f];
    pixels are evaluated in n_slice slices of int(H'W'/n_slice) (eval.py:233-239: a remainder beyond the last slice stays
    0, i.e. 0.5 after the (img+1)/2 map - reference behaviour); temporal_interp routes the sparse grid through
    forward_inter (modules.py:72-73), whose t == 1 rows are NaN (the last frame of a --t_interp run; clamp keeps NaN).

    The reference slices a frame to bound the memory of its materialised activations; this path materialises none under no_grad, and
    every pixel is computed independently of its neighbours, so the n_slice * int(H'W'/n_slice) pixels the slices cover are evaluated in
    ONE model call - bit-identical to the sliced loop (tests/test_gpu_parity.py::test_eval_drivers...), pixels beyond the last slice
    still stay 0.  literal_slices=True runs the reference's loop slice by slice (bench.py --mode eval reports both)."""
    dev = next(model.parameters()).device
    nframes = org_nframes if nframes is None else nframes
    Hq, Wq = int(resolution[0]), int(resolution[1])
    total = Hq * Wq
    # the (row, col) lattice of a resolution is the same for every frame and every slice: built once per (device, H', W') with the
    # reference's arithmetic (dataio.get_mgrid: k / (side - 1)), then sliced - a slice costs the model call and one copy, not seven
    # element-wise launches (the reference's 100-slice loop is launch-bound on a GPU this fast)
    key = (dev, Hq, Wq)
    lat = _LATTICE.get(key)
    if lat is None:
        rows = torch.arange(Hq, device=dev, dtype=torch.float32) / max(Hq - 1, 1)
        cols = torch.arange(Wq, device=dev, dtype=torch.float32) / max(Wq - 1, 1)      # dataio.get_mgrid: k / (side - 1)
        p = torch.arange(total, device=dev)
        lat = torch.empty((total, 3), device=dev, dtype=torch.float32)
        lat[:, 1] = rows[torch.div(p, Wq, rounding_mode="floor")]
        lat[:, 2] = cols[p % Wq]
        _LATTICE[key] = lat
        while len(_LATTICE) > 1 and sum(t.numel() * 4 for t in _LATTICE.values()) > _LATTICE_MAX_BYTES:
            _LATTICE.pop(next(iter(_LATTICE)))               # oldest first; the newest always stays
    half_dt = 0.5 / org_nframes
    tstep = torch.linspace(half_dt, 1 - half_dt, nframes)[f].item()
    tcoord = torch.linspace(0, 1, nframes)[f].item()
    coords_all = lat.clone()
    coords_all[:, 0] = tcoord
    steps_all = torch.full((1, total), tstep, device=dev, dtype=torch.float32)
    out = torch.zeros((total, 3), device=dev, dtype=torch.float32)
    split = int(total / n_slice)
    from . import functional
    if hooks is None:                   # the parameters cannot change inside this call (no_grad): the slices share one weight pack
        hooks = functional.StepHooks()
        hooks.packed_cache = {}
    spans = [(i * split, (i + 1) * split) for i in range(n_slice if split > 0 else 0)]
    if not literal_slices and spans:
        # the same pixels in one call - or, where a call materialises per-pixel state, in chunks of <= MAX_PIXELS_PER_CALL (bit-identical:
        # every pixel is independent), so a 4K or --s_interp lattice cannot ask for tens of GB of latent
        covered = split * n_slice
        one_call = functional.materialises_nothing(model, temporal_interp)
        step = covered if one_call else max(min(covered, MAX_PIXELS_PER_CALL), 1)
        spans = [(lo, min(lo + step, covered)) for lo in range(0, covered, step)]
    for lo, hi in spans:
        out[lo:hi] = model({"all_coords": coords_all[lo:hi].unsqueeze(0), "temporal_steps": steps_all[:, lo:hi], "nvp_hooks": hooks},
                           temporal_interp=temporal_interp)["model_out"].reshape(-1, 3)
    return torch.clamp((out.reshape(Hq, Wq, 3) + 1) / 2, 0, 1)


@torch.no_grad()
def eval_psnr(model, video: DeviceVideo, frames=None, n_slice: int = 100, s_interp: int = -1, t_interp: int = -1, on_frame=None,
              literal_slices: bool = False):
    """The reference's evaluation driver (eval.py:201-263) on a device-resident video.

    Plain run (s_interp == t_interp == -1): per-frame PSNR on [0,1] - (img+1)/2, clamp, vs u8/255 - averaged over
    `frames` (default: all), returned as a float.  With --s_interp k the query lattice is k times finer in both image
    axes; with --t_interp k there are k times as many output frames and the sparse grid is read through forward_inter.
    Like the reference, interpolated runs compute no PSNR (there is no ground truth): the frames are handed to
    `on_frame(f, img[H',W',3])` and None is returned."""
    res = (video.H, video.W)
    nframes = video.T
    temporal_interp = False
    if s_interp != -1:
        res = (video.H * s_interp, video.W * s_interp)
    if t_interp != -1:
        temporal_interp = True
        nframes = video.T * t_interp
    plain = s_interp == -1 and t_interp == -1
    psnrs = []
    from . import functional
    hooks = functional.StepHooks()
    hooks.packed_cache = {}                                 # one weight pack for the whole evaluation: nothing updates the model in here
    for f in (range(nframes) if frames is None else frames):
        img = render_frame(model, f, video.T, res, nframes, temporal_interp, n_slice, hooks=hooks, literal_slices=literal_slices)
        if on_frame is not None:
            on_frame(f, img)
        if plain:
            gt = video.video[f].float() / 255.0
            psnrs.append(10.0 * math.log10(1.0 / float(((img - gt) ** 2).mean())))
    return (sum(psnrs) / len(psnrs)) if (plain and psnrs) else None


def eval_slices(total_pixels: int, want: int = 100) -> int:
    """The reference evaluates a frame in Nslice = 100 slices of int(total/100) pixels (eval.py:233-239) and leaves a
    remainder unrendered; UVG-HD and 4K frame sizes divide by 100.  For other sizes pick the largest slice count <= want
    that divides the frame, so that every pixel is rendered."""
    for k in range(min(want, total_pixels), 0, -1):
        if total_pixels % k == 0:
            return k
    return 1


def load_video(path: str, frames: int, height: int = 0, width: int = 0) -> torch.Tensor:
    """uint8 [T, H, W, 3] on the host, the reference's loader restated (dataio.py:33-66):
      * `*.npy`            - np.load, first `frames` frames;
      * a directory        - sorted(glob('*.png'))[:frames] through PIL (what the reference's README extracts UVG into);
      * `*.yuv`            - raw planar 8-bit 4:2:0 (how UVG is distributed), needs height / width; converted with the
                             BT.709 limited-range matrix and nearest chroma up-sampling (an ffmpeg-free approximation of
                             the reference's `ffmpeg -i x.yuv f%05d.png` step; use the PNG route for exact parity)."""
    import glob
    import os
    import numpy as np
    if path.endswith(".npy"):
        v = np.load(path, mmap_mode="r")[:frames]
        return torch.from_numpy(np.ascontiguousarray(v)).to(torch.uint8)
    if os.path.isdir(path):
        from PIL import Image
        files = sorted(glob.glob(os.path.join(path, "*.png")))[:frames]
        if not files:
            raise FileNotFoundError(f"no *.png frames under {path}")
        first = np.array(Image.open(files[0]).convert("RGB"))
        out = np.zeros((len(files),) + first.shape, dtype=np.uint8)
        for i, f in enumerate(files):
            out[i] = np.array(Image.open(f).convert("RGB"))
        return torch.from_numpy(out)
    if path.endswith(".yuv"):
        if height <= 0 or width <= 0:
            raise ValueError("raw .yuv needs --height and --width")
        fsz = height * width * 3 // 2
        n = min(frames, os.path.getsize(path) // fsz)
        out = np.zeros((n, height, width, 3), dtype=np.uint8)
        with open(path, "rb") as fh:
            for i in range(n):
                buf = np.frombuffer(fh.read(fsz), dtype=np.uint8)
                y = buf[:height * width].reshape(height, width).astype(np.float32)
                u = buf[height * width:height * width * 5 // 4].reshape(height // 2, width // 2).astype(np.float32)
                v = buf[height * width * 5 // 4:].reshape(height // 2, width // 2).astype(np.float32)
                u = np.repeat(np.repeat(u, 2, 0), 2, 1) - 128.0
                v = np.repeat(np.repeat(v, 2, 0), 2, 1) - 128.0
                yy = (y - 16.0) * (255.0 / 219.0)
                r = yy + 1.5748 * v * (255.0 / 224.0)
                g = yy - (0.1873 * u + 0.4681 * v) * (255.0 / 224.0)
                b = yy + 1.8556 * u * (255.0 / 224.0)
                out[i] = np.clip(np.stack((r, g, b), -1) + 0.5, 0, 255).astype(np.uint8)
        return torch.from_numpy(out)
    raise ValueError(f"unsupported video source {path!r}: expected .npy, a PNG directory or .yuv")


def _pink_field(ch: int, h: int, w: int, alpha: float, gen: torch.Generator, device, r0: float = 0.0) -> torch.Tensor:
    """[ch, h, w] fields whose amplitude spectrum falls as 1 / f^alpha (power 1 / f^(2 alpha): alpha = 1 is the classic
    natural-image spectrum), zero mean, unit standard deviation per channel."""
    noise = torch.randn((ch, h, w), generator=gen, device=device)
    fy = torch.fft.fftfreq(h, device=device)[:, None]
    fx = torch.fft.rfftfreq(w, device=device)[None, :]
    r = torch.sqrt(fx * fx + fy * fy)
    filt = 1.0 / torch.clamp(r + r0, min=1.0 / max(h, w)) ** alpha
    filt[0, 0] = 0.0
    out = torch.fft.irfft2(torch.fft.rfft2(noise) * filt, s=(h, w))
    return (out - out.mean(dim=(1, 2), keepdim=True)) / out.std(dim=(1, 2), keepdim=True)


def natural_video(T: int, H: int, W: int, device, seed: int = 0, grain: float = 1.25, alpha: float = 1.15, n_objects: int = 6) -> torch.Tensor:
    """Synthetic clip with NATURAL-IMAGE STATISTICS - a stand-in for the UVG clips the reference's README table is measured
    on (README.md:92-100), which are not shipped and cannot be fetched.  NOT UVG: numbers measured on it only say how the
    encoder behaves on content with a 1/f spectrum, edges, motion and sensor noise instead of a handful of sinusoids.

      * scene: a texture larger than the frame with a 1/f^alpha amplitude spectrum (luma + weaker, smoother chroma), cut by
        the level sets of a smoother field into regions of different brightness (occlusion-like edges);
      * global motion: the camera pans (two incommensurate sinusoids) and breathes (+-2 % zoom) over the scene, sub-pixel
        bilinear resampling per frame;
      * local motion: `n_objects` soft-edged discs with their own texture and brightness cross the frame on curved paths;
      * grain: independent Gaussian noise per frame, pixel and channel (`grain` 8-bit levels: caps the PSNR any encoder can
        reach at about 20 log10(255 / grain) dB = 44.6 dB for 1.5).
    Deterministic for a (device type, seed).  uint8 [T, H, W, 3] on `device`."""
    import torch.nn.functional as Fn
    device = torch.device(device)
    gen = torch.Generator(device=device).manual_seed(seed)
    pad_y, pad_x = max(2, int(0.12 * H)), max(2, int(0.12 * W))
    Hs, Ws = H + 2 * pad_y, W + 2 * pad_x
    luma = _pink_field(1, Hs, Ws, alpha, gen, device)
    chroma = _pink_field(2, Hs, Ws, alpha + 0.4, gen, device)
    region = _pink_field(1, Hs, Ws, 1.8, gen, device)[0]
    luma = 0.55 * luma + 0.9 * (region > 0.35).float() - 0.7 * (region < -0.6).float() + 0.35 * (region.abs() < 0.08).float()
    luma = (luma - luma.mean()) / luma.std()
    c1, c2 = 0.07 * chroma[0], 0.07 * chroma[1]
    y = 0.45 + 0.17 * luma[0]
    scene = torch.stack((y + 1.402 * c2, y - 0.344 * c1 - 0.714 * c2, y + 1.772 * c1), dim=0)[None]       # [1,3,Hs,Ws]
    obj_tex = _pink_field(3, Hs, Ws, alpha, gen, device)[None] * 0.16
    # object parameters (host side, from a CPU generator so that they do not depend on the device's RNG stream)
    hg = torch.Generator().manual_seed(seed + 77)
    ob = torch.rand((n_objects, 8), generator=hg)
    ys = torch.linspace(-1, 1, H, device=device)[:, None].expand(H, W)
    xs = torch.linspace(-1, 1, W, device=device)[None, :].expand(H, W)
    py = torch.arange(H, device=device, dtype=torch.float32)[:, None]
    px = torch.arange(W, device=device, dtype=torch.float32)[None, :]
    out = torch.empty((T, H, W, 3), device=device, dtype=torch.uint8)
    for f in range(T):
        t = f / max(T - 1, 1)
        # camera: pan inside the padding, slow zoom
        oy = 0.8 * pad_y * math.sin(2 * math.pi * 0.45 * t + 1.0)
        ox = 0.8 * pad_x * math.sin(2 * math.pi * 0.70 * t)
        zoom = 1.0 + 0.02 * math.sin(2 * math.pi * 0.9 * t + 0.5)
        gy = (ys * (H / 2) / zoom + oy) / (Hs / 2)
        gx = (xs * (W / 2) / zoom + ox) / (Ws / 2)
        img = Fn.grid_sample(scene, torch.stack((gx, gy), dim=-1)[None], mode="bilinear", padding_mode="border", align_corners=False)[0]
        for k in range(n_objects):
            o = ob[k]
            rad = (0.04 + 0.10 * float(o[0])) * H
            cy = (0.15 + 0.7 * float(o[1])) * H + 0.25 * H * math.sin(2 * math.pi * (0.3 + float(o[2])) * t + 6.28 * float(o[3]))
            cx = (-0.1 + 1.2 * ((float(o[4]) + (0.4 + 0.8 * float(o[5])) * t) % 1.0)) * W
            d = torch.sqrt((py - cy) ** 2 + (px - cx) ** 2)
            a = torch.clamp((rad - d) / 1.5 + 0.5, 0, 1)[None]                                  # soft 1.5-px edge
            # the object's texture moves with it: sample obj_tex at (pixel - centre) + a per-object offset
            ty = ((py - cy) + (0.2 + 0.6 * float(o[6])) * Hs - Hs / 2) / (Hs / 2)
            tx = ((px - cx) + (0.2 + 0.6 * float(o[7])) * Ws - Ws / 2) / (Ws / 2)
            tex = Fn.grid_sample(obj_tex, torch.stack((tx.expand(H, W), ty.expand(H, W)), dim=-1)[None], mode="bilinear",
                                 padding_mode="border", align_corners=False)[0]
            col = torch.tensor([0.25 + 0.5 * float(o[(k + c) % 8]) for c in range(3)], device=device)[:, None, None]
            img = img * (1 - a) + (col + tex) * a
        img = img * 255.0 + grain * torch.randn((3, H, W), generator=gen, device=device)
        out[f] = torch.clamp(img + 0.5, 0, 255).to(torch.uint8).permute(1, 2, 0)
    return out


def procedural_video(T: int, H: int, W: int, device, seed: int = 0) -> torch.Tensor:
    """Deterministic smooth moving pattern (stand-in for UVG frames, which are not shipped)."""
    g = torch.Generator().manual_seed(seed)
    ph = torch.rand(6, generator=g) * 6.28
    t = torch.linspace(0, 1, T, device=device)[:, None, None]
    y = torch.linspace(0, 1, H, device=device)[None, :, None]
    x = torch.linspace(0, 1, W, device=device)[None, None, :]
    chans = []
    for c in range(3):
        v = (torch.sin(6.28 * (3 + c) * x + 4 * t + ph[c]) * torch.cos(6.28 * (2 + c) * y - 3 * t + ph[3 + c])
             + 0.5 * torch.sin(25 * (x - 0.3 * t) * (y + 0.2)))
        chans.append(((v / 1.5 * 0.5 + 0.5).clamp(0, 1) * 255).to(torch.uint8))
    return torch.stack(chans, dim=-1).contiguous()
