/* A plain C99 caller of the drop-in boundary ("C-ABI layer underneath, so kernels are testable without torch"):
 * include/nvp_hip.h compiles as C, libnvp_hip.so is loaded with dlopen (as a cgo / JNI / ctypes binding would), and entry points that
 * decide on the host - version string, latent geometry, argument validation - are called.  No GPU needed: tests/test_host.py runs this.
 *   gcc -std=c99 -I include tests/cabi/cabi_host.c -ldl -o cabi_host && ./cabi_host nvp_amd/csrc/libnvp_hip.so */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "nvp_hip.h"

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: cabi_host LIB\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    const char* (*version)(void) = (const char* (*)(void))dlsym(h, "nvp_version");
    int32_t (*rows)(int32_t) = (int32_t (*)(int32_t))dlsym(h, "nvp_latent_rows");
    int (*sparse_fwd)(const float*, const float*, float*, int64_t, const nvp_sparse_shape*, void*) =
        (int (*)(const float*, const float*, float*, int64_t, const nvp_sparse_shape*, void*))dlsym(h, "nvp_sparse3x3_fwd");
    if (!version || !rows || !sparse_fwd) { fprintf(stderr, "missing symbol\n"); return 4; }
    const nvp_sparse_shape sh = {8, 9, 7, 2};
    const int rc_null = sparse_fwd(NULL, NULL, NULL, 16, &sh, NULL);          /* argument errors are decided before anything is enqueued */
    const int rc_empty = sparse_fwd(NULL, NULL, NULL, 0, &sh, NULL);
    printf("%s|rows114=%d|rows228=%d|null=%d|empty=%d|badarg=%d\n", version(), (int)rows(114), (int)rows(228), rc_null, rc_empty, (int)NVP_ERR_BADARG);
    return 0;
}
