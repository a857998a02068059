"""Time nvp_mlp_fwd / nvp_mlp_bwd_dx variants (tools/ablate.sh) directly through the C ABI."""
import ctypes as C, os, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nvp_amd import _lib as L
dev = torch.device("cuda:0")
n, d = 1245184, 114
nt = L.ntiles(n)
torch.manual_seed(0)
rows = (d + 3) // 4 * 4
zt = torch.randn((nt, rows, 32), device=dev) * 0.3
steps = torch.rand(n, device=dev)
H = 128
shapes = [(H, d), (H,), (H, H + d), (H,), (H, H + d), (H,), (H, 1), (H,), (H, H), (H,), (H, H), (H,), (3, H), (3,)]
mlp = [torch.randn(s, device=dev) * 0.1 for s in shapes]
ps = L.mlp_params_struct(mlp)
rgb = torch.empty((n, 3), device=dev)
drgb = torch.randn((n, 3), device=dev) * 1e-3
saved = torch.empty((5, nt, H, 32), device=dev)
dy = torch.empty((6, nt, H, 32), device=dev)
xs = torch.empty((3, nt, H, 32), device=dev)
dz = torch.empty((nt * 32, (d + 3) // 4 * 4), device=dev)
stream = torch.cuda.current_stream().cuda_stream
vp = lambda t: C.c_void_p(t.data_ptr())
def timeit(fn, reps=5):
    for _ in range(2): assert fn() == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for path in sorted(glob.glob(os.path.join(ROOT, "tools", "bin", "libmlp_*.so"))):
    lib = C.CDLL(path)
    lib.nvp_packed_fwd_floats.restype = C.c_int64; lib.nvp_packed_bwd_floats.restype = C.c_int64
    pf = torch.empty(lib.nvp_packed_fwd_floats(C.c_int32(d)), device=dev)
    pb = torch.empty(lib.nvp_packed_bwd_floats(C.c_int32(d)), device=dev)
    lib.nvp_mlp_pack_fwd(C.byref(ps), vp(pf), C.c_int32(d), C.c_void_p(stream))
    lib.nvp_mlp_pack_bwd(C.byref(ps), vp(pb), C.c_int32(d), C.c_void_p(stream))
    f = lambda: lib.nvp_mlp_fwd(vp(zt), vp(steps), C.byref(ps), vp(pf), vp(rgb), vp(saved), C.c_int64(n), C.c_int32(d), C.c_void_p(stream))
    g = lambda: lib.nvp_mlp_bwd_dx(vp(drgb), vp(steps), vp(saved), C.byref(ps), vp(pb), vp(dy), vp(dz), None, C.c_int64(n), C.c_int32(d), C.c_void_p(stream))
    lib.nvp_dw_partial_floats.restype = C.c_int64
    if os.environ.get("NVP_LOOP_STAGE"):          # tools/power_probe.sh: loop one stage for a few seconds
        import time
        nchl = 256
        partl = torch.empty(lib.nvp_dw_partial_floats(C.c_int32(d), C.c_int32(nchl)), device=dev)
        gradsl = [torch.empty_like(t) for t in mlp]; gsl = L.mlp_params_struct(gradsl)
        hl = lambda: lib.nvp_mlp_bwd_dw(vp(drgb), vp(steps), vp(zt), vp(saved), vp(dy), C.byref(ps), vp(partl), C.c_int32(nchl), C.byref(gsl), C.c_int64(n), C.c_int32(d), C.c_void_p(stream))
        fn = {"fwd": f, "bwd": g, "dw": hl}[os.environ["NVP_LOOP_STAGE"]]
        f(); g(); torch.cuda.synchronize()
        t0 = time.time(); it = 0
        while time.time() - t0 < float(os.environ.get("NVP_LOOP_SECONDS", "8")):
            for _ in range(50): fn()
            torch.cuda.synchronize(); it += 50
        print(f"stage {os.environ['NVP_LOOP_STAGE']}: {it} launches, {(time.time() - t0) / it * 1e3:.3f} ms each")
        sys.exit(0)
    grads = [torch.empty_like(t) for t in mlp]
    gs = L.mlp_params_struct(grads)
    for nch in [int(v) for v in os.environ.get("NCH", "256").split(",")]:
        part = torch.empty(lib.nvp_dw_partial_floats(C.c_int32(d), C.c_int32(nch)), device=dev)
        hdw = lambda: lib.nvp_mlp_bwd_dw(vp(drgb), vp(steps), vp(zt), vp(saved), vp(dy), C.byref(ps), vp(part), C.c_int32(nch), C.byref(gs), C.c_int64(n), C.c_int32(d), C.c_void_p(stream))
        res = []
        for rep in range(3):          # interleaved repeats: same box, same process
            res.append((timeit(f), timeit(g), timeit(hdw)))
        tf, tb, tw = (min(r[i] for r in res) for i in range(3))
        print(f"{os.path.basename(path):18s} nch {nch:4d} fwd {tf:6.3f} ms {219648 * n / tf / 1e9:6.1f} TF | bwd_dx+dz {tb:6.3f} ms {219392 * n / tb / 1e9:6.1f} TF | dw {tw:6.3f} ms {219648 * n / tw / 1e9:6.1f} TF   all: {[tuple(round(x, 3) for x in r) for r in res]}", flush=True)
