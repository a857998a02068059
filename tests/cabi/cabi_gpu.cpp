// A C++ caller of the drop-in boundary WITHOUT torch and without Python: device memory from the HIP runtime, the entry points
// of include/nvp_hip.h through dlopen, results against plain host loops that restate sparsegrid.py:23-72 (SparseGrid.forward: nearest
// index = trunc(fl((res - 1) c) + 0.5), clamped 3 x 3 copy; its autograd: index_put_(accumulate)).  Bit-exact: the forward is a copy, the
// backward sums exactly representable values.  tests/test_gpu_parity.py builds and runs this on the GPU box.
//   hipcc -I include tests/cabi/cabi_gpu.cpp -ldl -o cabi_gpu && ./cabi_gpu nvp_amd/csrc/libnvp_hip.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "nvp_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s (line %d)\n", hipGetErrorString(e_), __LINE__); return 10; } } while (0)

static int nearest(float c, int res) {
    volatile float f = (float)(res - 1) * c;          // separately rounded product and sum (the reference's two torch ops)
    volatile float g = f + 0.5f;
    int i = (int)g;
    return i < 0 ? 0 : (i > res - 1 ? res - 1 : i);
}

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: cabi_gpu LIB\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 3; }
    typedef int (*fwd_t)(const float*, const float*, float*, int64_t, const nvp_sparse_shape*, void*);
    typedef int (*bwd_t)(const float*, const float*, float*, int64_t, const nvp_sparse_shape*, void*);
    fwd_t fwd = (fwd_t)dlsym(h, "nvp_sparse3x3_fwd");
    bwd_t bwd = (bwd_t)dlsym(h, "nvp_sparse3x3_bwd");
    if (!fwd || !bwd) { printf("missing symbol\n"); return 4; }
    const nvp_sparse_shape sh = {6, 11, 9, 2};
    const int F = sh.n_features, n = 1000 + 37;
    const size_t cells = (size_t)sh.t_res * sh.x_res * sh.y_res * F;
    std::vector<float> emb(cells), coords((size_t)n * 3), out((size_t)n * 9 * F), want(out.size()), dout(out.size()), demb(cells), dwant(cells, 0.f);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : emb) v = rnd() - 0.5f;
    for (auto& v : coords) v = rnd();
    for (int k = 0; k < 9; ++k) coords[k] = k % 2 ? 1.0f : 0.0f;                        // exact borders
    for (auto& v : dout) v = (float)((int)(rnd() * 64) - 32) * 0.125f;                  // multiples of 1/8: every sum is exact
    for (int i = 0; i < n; ++i) {
        const int t = nearest(coords[3 * i], sh.t_res), x = nearest(coords[3 * i + 1], sh.x_res), y = nearest(coords[3 * i + 2], sh.y_res);
        for (int a = -1; a <= 1; ++a)
            for (int b = -1; b <= 1; ++b) {
                const int vx = x + a < 0 ? 0 : (x + a > sh.x_res - 1 ? sh.x_res - 1 : x + a), vy = y + b < 0 ? 0 : (y + b > sh.y_res - 1 ? sh.y_res - 1 : y + b);
                for (int f = 0; f < F; ++f) {
                    const size_t cell = (((size_t)t * sh.x_res + vx) * sh.y_res + vy) * F + f, col = (size_t)i * 9 * F + (3 * (a + 1) + (b + 1)) * F + f;
                    want[col] = emb[cell];
                    dwant[cell] += dout[col];
                }
            }
    }
    float *d_emb, *d_coords, *d_out, *d_dout, *d_demb;
    CK(hipMalloc(&d_emb, cells * 4)); CK(hipMalloc(&d_coords, coords.size() * 4)); CK(hipMalloc(&d_out, out.size() * 4));
    CK(hipMalloc(&d_dout, out.size() * 4)); CK(hipMalloc(&d_demb, cells * 4));
    CK(hipMemcpy(d_emb, emb.data(), cells * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_coords, coords.data(), coords.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dout, dout.data(), out.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(d_demb, 0, cells * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    int rc = fwd(d_emb, d_coords, d_out, n, &sh, st);
    if (rc) { printf("nvp_sparse3x3_fwd rc %d\n", rc); return 5; }
    rc = bwd(d_coords, d_dout, d_demb, n, &sh, st);
    if (rc) { printf("nvp_sparse3x3_bwd rc %d\n", rc); return 6; }
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(demb.data(), d_demb, cells * 4, hipMemcpyDeviceToHost));
    const bool f_ok = !memcmp(out.data(), want.data(), out.size() * 4), b_ok = !memcmp(demb.data(), dwant.data(), cells * 4);
    const int rc_null = fwd(nullptr, d_coords, d_out, n, &sh, st);
    printf("forward %s, backward %s, NULL argument -> %d\n", f_ok ? "bit-exact" : "DIFFERS", b_ok ? "bit-exact" : "DIFFERS", rc_null);
    return (f_ok && b_ok && rc_null == NVP_ERR_BADARG) ? 0 : 7;
}
