"""Host emulation of the operand splits the MLP kernels run on the matrix cores (nvp_amd/csrc/mlp_b3.h):

  fp32 chain : sequential float32 fma accumulation (what an fp32 MFMA / the CPU oracle does)
  bf16 x 3   : x = hi + mid + lo in bf16, six products per k-step of 16, fp32 accumulation
  fp16 x 2   : x * 2^e = hi + lo in fp16 (power-of-two scale per weight matrix / per pixel), three products

Each MFMA is modelled as an exact sum of its 16 products added to the fp32 accumulator with one rounding (pessimistic).
Errors are against a float64 dot product.  Usage: python tools/split_accuracy.py > profiles/r02_probe_f16x2_split.txt"""
import numpy as np

rng = np.random.default_rng(0)


def bf16(x):
    x = np.asarray(x, np.float32)
    b = x.view(np.uint32).astype(np.uint64)
    return ((b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split_bf3(x):
    h = bf16(x)
    r = (x - h).astype(np.float32)
    m = bf16(r)
    return h, m, bf16((r - m).astype(np.float32))


def split_f16(x, s):
    t = (x * s).astype(np.float32)
    h = t.astype(np.float16)
    r = (t - h.astype(np.float32)).astype(np.float32)
    return h.astype(np.float32), r.astype(np.float16).astype(np.float32)


def pow2_scale(m):          # 2^e with m 2^e in [2^13, 2^14)
    return 2.0 ** (13 - np.floor(np.log2(m)))


def run(K, n=20000, wscale=0.1, ascale=1.0, dist="normal", bias=True):
    w = (rng.uniform(-1, 1, (n, K)) * wscale).astype(np.float32)
    a = ((rng.standard_normal((n, K)) if dist == "normal" else rng.uniform(-1, 1, (n, K))) * ascale).astype(np.float32)
    ref = (w.astype(np.float64) * a).sum(1)
    c = np.zeros(n, np.float64)
    for k in range(K):
        c = (c + w[:, k].astype(np.float64) * a[:, k]).astype(np.float32).astype(np.float64)
    e32 = np.abs(c - ref)
    f = lambda x, y: x.astype(np.float64) * y

    def chain(parts_w, parts_a, order, scale=1.0):
        acc = np.zeros(n, np.float32)
        for k in range(0, K, 16):
            sl = slice(k, k + 16)
            for (i, j) in order:
                acc = (acc.astype(np.float64) + f(parts_w[i][:, sl], parts_a[j][:, sl]).sum(1)).astype(np.float32)
        return np.abs((acc.astype(np.float64) / scale).astype(np.float32) - ref)

    eb3 = chain(split_bf3(w), split_bf3(a), [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)])
    sw = pow2_scale(np.abs(w).max())
    amax = np.abs(a).max(1, keepdims=True)
    sa = pow2_scale(np.maximum(amax, 1.0) if bias else np.maximum(amax, 2.0 ** -50))
    eh2 = chain(split_f16(w, sw), split_f16(a, sa), [(1, 0), (0, 1), (0, 0)], sw * sa[:, 0])
    rms = lambda e: np.sqrt((e ** 2).mean())
    print(f"K={K:3d} |w|<={wscale:<6g} act~{ascale:<6g}{dist[0]} bias-clamp={int(bias)} mean|ref|={np.abs(ref).mean():.3g}   max / rms abs error:  "
          f"fp32 chain {e32.max():.2e} / {rms(e32):.2e}   bf16x3 {eb3.max():.2e} / {rms(eb3):.2e}   fp16x2 {eh2.max():.2e} / {rms(eh2):.2e}")


if __name__ == "__main__":
    print(__doc__.split("Usage")[0])
    run(128)
    run(242)
    run(128, wscale=1.0, dist="uniform")
    run(128, wscale=0.007, ascale=1e-4)
    run(128, wscale=0.1, ascale=1e-7, bias=True)
    run(128, wscale=0.1, ascale=1e-7, bias=False)
    run(128, wscale=0.05, ascale=3e-9, bias=False)
