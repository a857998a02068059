"""Time nvp_encode_bwd for the libnvp_*.so variants built by tools/ablate_scatter.sh (same box, same process)."""
import ctypes as C, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from nvp_amd import _lib as L
dev = torch.device("cuda:0")
n = 1245184
torch.manual_seed(0)
cfg = bench.CONFIG_NVP_S
lv = [L.make_levels(cfg[k]) for k in ("2d_encoding_xy", "2d_encoding_yt", "2d_encoding_xt")]
sh = L.SparseShape(600, 300, 300, 2)
coords = torch.rand((n, 3), device=dev)
coords = coords[torch.argsort(coords[:, 2])].contiguous()
dz = torch.randn((n, 116), device=dev) * 1e-4
g = [torch.empty(L.levels_n_params(v), device=dev) for v in lv]
demb = torch.empty((600, 300, 300, 2), device=dev)
stream = torch.cuda.current_stream().cuda_stream
vp = lambda t: C.c_void_p(t.data_ptr())
ref = None
for path in sorted(glob.glob(os.path.join(ROOT, "tools", "bin", "libnvp_*.so"))):
    lib = C.CDLL(path)
    lib.nvp_encode_bwd_workspace_bytes.restype = C.c_int64
    wsb = lib.nvp_encode_bwd_workspace_bytes(C.c_int64(n), C.byref(lv[0]), C.byref(lv[1]), C.byref(lv[2]), C.byref(sh))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    for flags in (0, 1):
        f = lambda: lib.nvp_encode_bwd(vp(coords), vp(dz), C.c_int32(116), vp(g[0]), vp(g[1]), vp(g[2]), vp(demb), C.c_int64(n),
                                       C.byref(lv[0]), C.byref(lv[1]), C.byref(lv[2]), C.byref(sh), vp(ws), C.c_int64(wsb), C.c_int32(flags), C.c_void_p(stream))
        for _ in range(2): assert f() == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): f()
        b.record(); torch.cuda.synchronize()
        chk = float(g[0].double().sum() + g[2].double().sum() + demb.double().sum())
        if ref is None: ref = chk
        print(f"{os.path.basename(path):22s} flags={flags} {a.elapsed_time(b)/5:6.3f} ms  checksum diff {abs(chk-ref):.3e}", flush=True)
