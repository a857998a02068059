"""Time nvp_mlp_fwd variants (tools/ablate.sh) directly through the C ABI."""
import ctypes as C, os, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nvp_amd import _lib as L
dev = torch.device("cuda:0")
n, d = 1245184, 114
nt = L.ntiles(n)
torch.manual_seed(0)
zt = torch.randn((nt, (d + 3) // 4 * 4, 32), device=dev) * 0.3
steps = torch.rand(n, device=dev)
H = 128
shapes = [(H, d), (H,), (H, H + d), (H,), (H, H + d), (H,), (H, 1), (H,), (H, H), (H,), (H, H), (H,), (3, H), (3,)]
mlp = [torch.randn(s, device=dev) * 0.1 for s in shapes]
ps = L.mlp_params_struct(mlp)
rgb = torch.empty((n, 3), device=dev)
saved = torch.empty((5, nt, H, 32), device=dev)
stream = torch.cuda.current_stream().cuda_stream
for path in sorted(glob.glob(os.path.join(ROOT, "tools", "bin", "libfwd_*.so"))):
    lib = C.CDLL(path)
    lib.nvp_packed_fwd_floats.restype = C.c_int64
    packed = torch.empty(lib.nvp_packed_fwd_floats(C.c_int32(d)), device=dev)
    lib.nvp_mlp_pack_fwd(C.byref(ps), C.c_void_p(packed.data_ptr()), C.c_int32(d), C.c_void_p(stream))
    for label, sv in (("save", saved), ("nosave", None)):
        args = (C.c_void_p(zt.data_ptr()), C.c_void_p(steps.data_ptr()), C.byref(ps), C.c_void_p(packed.data_ptr()),
                C.c_void_p(rgb.data_ptr()), C.c_void_p(sv.data_ptr() if sv is not None else None), C.c_int64(n), C.c_int32(d), C.c_void_p(stream))
        for _ in range(2): assert lib.nvp_mlp_fwd(*args) == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): lib.nvp_mlp_fwd(*args)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        print(f"{os.path.basename(path):22s} {label:7s} {ms:7.3f} ms  {219648 * n / ms / 1e9:6.1f} TF", flush=True)
