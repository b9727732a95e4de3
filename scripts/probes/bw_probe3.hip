// bw_probe3.hip -- do cache-policy bits (nt / sc0 / sc1) on global accesses keep a stream OUT of the 256 MB
// Infinity Cache of gfx950, so that a second, plainly accessed set stays resident?
//   set A: SA MB, read+write with plain accesses (the "fields")
//   set B: SB MB, read+write with the policy under test (the "accumulators")
// per repetition: kernel(A), kernel(B); reported: average time of kernel(A) and kernel(B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));

#define DEF_KERNEL(NAME, LMOD, SMOD)                                                                      \
    __global__ void NAME(v4 *__restrict__ a, size_t n) {                                                  \
        size_t stride = (size_t)gridDim.x * blockDim.x;                                                   \
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;                                         \
        for (; i + 3 * stride < n; i += 4 * stride) {                                                     \
            v4 x0, x1, x2, x3;                                                                            \
            v4 *p0 = a + i, *p1 = a + i + stride, *p2 = a + i + 2 * stride, *p3 = a + i + 3 * stride;     \
            asm volatile("global_load_dwordx4 %0, %4, off " LMOD "\n"                                     \
                         "global_load_dwordx4 %1, %5, off " LMOD "\n"                                     \
                         "global_load_dwordx4 %2, %6, off " LMOD "\n"                                     \
                         "global_load_dwordx4 %3, %7, off " LMOD "\n"                                     \
                         "s_waitcnt vmcnt(0)"                                                             \
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)                                     \
                         : "v"(p0), "v"(p1), "v"(p2), "v"(p3)                                             \
                         : "memory");                                                                     \
            x0 = x0 * 1.0001f + 1.0f; x1 = x1 * 1.0001f + 1.0f; x2 = x2 * 1.0001f + 1.0f; x3 = x3 * 1.0001f + 1.0f; \
            asm volatile("global_store_dwordx4 %0, %4, off " SMOD "\n"                                    \
                         "global_store_dwordx4 %1, %5, off " SMOD "\n"                                    \
                         "global_store_dwordx4 %2, %6, off " SMOD "\n"                                    \
                         "global_store_dwordx4 %3, %7, off " SMOD "\n"                                    \
                         :: "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(x0), "v"(x1), "v"(x2), "v"(x3)        \
                         : "memory");                                                                     \
        }                                                                                                 \
    }

DEF_KERNEL(k_plain, "", "")
DEF_KERNEL(k_nt, "nt", "nt")
DEF_KERNEL(k_sc1, "sc1", "sc1")
DEF_KERNEL(k_sc0sc1, "sc0 sc1", "sc0 sc1")
DEF_KERNEL(k_ntsc1, "sc1 nt", "sc1 nt")
DEF_KERNEL(k_all, "sc0 sc1 nt", "sc0 sc1 nt")
DEF_KERNEL(k_ntsc0, "sc0 nt", "sc0 nt")

typedef void (*kern_t)(v4 *, size_t);

int main(int argc, char **argv) {
    size_t SA = (argc > 1 ? atoi(argv[1]) : 180), SB = (argc > 2 ? atoi(argv[2]) : 120);
    int reps = 50;
    v4 *A, *B;
    CK(hipMalloc((void **)&A, SA << 20)); CK(hipMemset(A, 0, SA << 20));
    CK(hipMalloc((void **)&B, (SB ? SB : 1) << 20)); CK(hipMemset(B, 0, (SB ? SB : 1) << 20));
    size_t nA = (SA << 20) / 16, nB = (SB << 20) / 16;
    struct { const char *n; kern_t k; } var[] = {{"plain", k_plain}, {"nt", k_nt}, {"sc1", k_sc1}, {"sc0 sc1", k_sc0sc1}, {"sc1 nt", k_ntsc1}, {"sc0 nt", k_ntsc0}, {"sc0 sc1 nt", k_all}};
    hipEvent_t ev[3];
    for (auto &x : ev) CK(hipEventCreate(&x));
    printf("set A %zu MB plain, set B %zu MB with policy under test; r+w streams, us per kernel (TB/s)\n", SA, SB);
    for (auto &v : var) {
        double ta = 0, tb = 0;
        for (int r = -5; r < reps; r++) {
            CK(hipEventRecord(ev[0], 0));
            hipLaunchKernelGGL(k_plain, dim3((unsigned)(nA / 1024)), dim3(256), 0, 0, A, nA + 1);
            CK(hipEventRecord(ev[1], 0));
            if (SB) hipLaunchKernelGGL(v.k, dim3((unsigned)(nB / 1024)), dim3(256), 0, 0, B, nB + 1);
            CK(hipEventRecord(ev[2], 0));
            CK(hipEventSynchronize(ev[2]));
            float m1, m2;
            CK(hipEventElapsedTime(&m1, ev[0], ev[1])); CK(hipEventElapsedTime(&m2, ev[1], ev[2]));
            if (r >= 0) { ta += m1; tb += m2; }
        }
        ta = ta * 1e3 / reps; tb = tb * 1e3 / reps;
        printf("  B policy %-12s: A %8.2f us (%5.2f TB/s)   B %8.2f us (%5.2f TB/s)\n", v.n, ta, 2.0 * (SA << 20) / ta / 1e6, tb,
               SB ? 2.0 * (SB << 20) / tb / 1e6 : 0.0);
    }
    return 0;
}
