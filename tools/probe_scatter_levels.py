#!/usr/bin/env python3
"""Where does the dense planes' scatter spend its time?  Times nvp_encode_bwd (dense planes only, presorted) at N = 1 245 184 for keyframe
grids with the first L levels of config_nvp_s (L = 4, 8, 10, 12, 14, 16): the increments are the cost of each level group in band_kernel /
slab_reduce (the sparse grid is excluded through NVP_SCATTER_DENSE_ONLY)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nvp_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
lib = L.load()
n = bench.N_PX
g = torch.Generator(device=dev).manual_seed(0)
coords = torch.rand((n, 3), device=dev, generator=g)
coords[:, 2] = torch.randint(0, 1920, (n,), device=dev, generator=g).float() / 1919
coords = coords[torch.argsort(coords[:, 2], stable=True)].contiguous()
sh = L.SparseShape(600, 300, 300, 2)
for nl in (4, 8, 10, 12, 14, 16):
    cfg = dict(bench.CONFIG_NVP_S["2d_encoding_xy"], n_levels=nl)
    lv = L.make_levels(cfg)
    d = 3 * nl * 2 + 18
    stride = (d + 3) // 4 * 4
    dz = torch.randn((n, stride), device=dev, generator=g) * 1e-3
    ws = torch.empty(lib.nvp_encode_bwd_workspace_bytes(n, C.byref(lv), C.byref(lv), C.byref(lv), C.byref(sh)), device=dev, dtype=torch.uint8)
    grads = [torch.empty(L.levels_n_params(lv), device=dev) for _ in range(3)]
    demb = torch.empty((600, 300, 300, 2), device=dev)
    flags = L.COORDS_SORTED_BY_Y
    L.check(lib.nvp_encode_bwd_presort(L.ptr(coords), n, C.byref(lv), C.byref(lv), C.byref(lv), C.byref(sh), L.ptr(ws, torch.uint8), ws.numel(), flags, L.stream_ptr()), "presort")
    def run():
        L.check(lib.nvp_encode_bwd(L.ptr(coords), L.ptr(dz), stride, L.ptr(grads[0]), L.ptr(grads[1]), L.ptr(grads[2]), L.ptr(demb), n,
                                   C.byref(lv), C.byref(lv), C.byref(lv), C.byref(sh), L.ptr(ws, torch.uint8), ws.numel(), flags | L.SCATTER_PRESORTED, L.stream_ptr()), "bwd")
    for _ in range(3):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        run()
    b.record()
    torch.cuda.synchronize()
    print(f"levels 0..{nl - 1}: whole scatter (planes + sparse grid) {a.elapsed_time(b) / 10:.3f} ms; finest res {lv.res[nl - 1]}")
