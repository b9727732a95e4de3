// param_maps.hip -- the parameterisation maps of the callers of the operator boundary, fused: user parameters on the
// physical (nz, nx) grid -> replicate padding -> mask blend with the reference model -> (Lambda [MPa], Mu [MPa], Den) on the
// padded grid in ONE launch, and the whole chain rule back (Lame derivatives, mask, transpose of the padding) in one
// launch over the grid plus a small one over its rim.
//
// Replaces, for HIP-resident tensors, the ~20 elementwise torch kernels over 3 x 9 MB that the reference's modules issue per
// iteration on the CPU (DAS_Waveform_Inversion/Ops/FWI/FWI_ops.py:116-127 FWI, :194-204 FWI_Lame_Den, :256-266 FWI_IP_IS_Den,
// :319-330 FWI_Vp_Vs_IP, :381-393 FWI_Vp_Vs_IS; padding fwi_utils.py:31-44 with nz == nz_orig, i.e. the identity resize):
//
//     P_pad  = replicate_pad(P)                                   (fwi_utils.py:40-43)
//     P_m    = Mask * P_pad + (1 - Mask) * P_ref                   (FWI_ops.py:120-122)
//     Lambda, Mu, Den = map_kind(A_m, B_m, C_m)                    (FWI_ops.py:124-125, ...)
//
// Same float32 operation order as the torch expressions (the library is built with -ffp-contract=off), divisions are
// IEEE divisions, so the forward map equals the CPU torch result bit for bit; the backward sums the padding strips in a
// fixed order (torch's replication_pad backward adds in another one: equal to rounding).
#include <hip/hip_runtime.h>

#include "param_maps.hpp"

namespace sepfwi {

namespace {

struct Triple {
    float a, b, c;
};

// (A_m, B_m, C_m) -> (Lambda, Mu, Den), float32, the reference's expression order
__device__ __forceinline__ Triple map_fwd(int kind, float a, float b, float c) {
    Triple o;
    switch (kind) {
        case PARAM_VP_VS_DEN:  // FWI_ops.py:124-125
            o.a = (a * a - 2.0f * (b * b)) * c / 1e6f;
            o.b = (b * b) * c / 1e6f;
            o.c = c;
            break;
        case PARAM_LAM_MU_DEN:  // FWI_ops.py:204
            o.a = a;
            o.b = b;
            o.c = c;
            break;
        case PARAM_IP_IS_DEN:  // FWI_ops.py:261-262
            o.a = (a * a - 2.0f * (b * b)) / c;
            o.b = (b * b) / c;
            o.c = c;
            break;
        case PARAM_VP_VS_IP:  // FWI_ops.py:326-328   (a = Vp, b = Vs, c = IP)
            o.a = c * a - 2.0f * c / a * (b * b);
            o.b = c / a * (b * b);
            o.c = c / a;
            break;
        default:  // PARAM_VP_VS_IS, FWI_ops.py:389-391   (a = Vp, b = Vs, c = IS)
            o.a = c / b * (a * a) - 2.0f * c * b;
            o.b = c * b;
            o.c = c / b;
            break;
    }
    return o;
}

// (dL/dLambda, dL/dMu, dL/dDen) at a padded cell -> its contribution to (dL/dA_m, dL/dB_m, dL/dC_m)
__device__ __forceinline__ Triple map_bwd(int kind, float a, float b, float c, float gl, float gm, float gd) {
    Triple o;
    switch (kind) {
        case PARAM_VP_VS_DEN:
            o.a = gl * (2.0f * a * c / 1e6f);
            o.b = gl * (-4.0f * b * c / 1e6f) + gm * (2.0f * b * c / 1e6f);
            o.c = gl * ((a * a - 2.0f * (b * b)) / 1e6f) + gm * ((b * b) / 1e6f) + gd;
            break;
        case PARAM_LAM_MU_DEN:
            o.a = gl;
            o.b = gm;
            o.c = gd;
            break;
        case PARAM_IP_IS_DEN:
            o.a = gl * (2.0f * a / c);
            o.b = gl * (-4.0f * b / c) + gm * (2.0f * b / c);
            o.c = -gl * ((a * a - 2.0f * (b * b)) / (c * c)) - gm * ((b * b) / (c * c)) + gd;
            break;
        case PARAM_VP_VS_IP: {
            const float r = b / a;  // Vs / Vp
            o.a = gl * (c + 2.0f * c * r * r) - gm * (c * r * r) - gd * (c / (a * a));
            o.b = gl * (-4.0f * c * r) + gm * (2.0f * c * r);
            o.c = gl * (a - 2.0f * b * r) + gm * (b * r) + gd / a;
            break;
        }
        default: {  // PARAM_VP_VS_IS
            const float r = a / b;  // Vp / Vs
            o.a = gl * (2.0f * c * r);
            o.b = gl * (-c * r * r - 2.0f * c) + gm * c - gd * (c / (b * b));
            o.c = gl * (a * r - 2.0f * b) + gm * b + gd / b;
            break;
        }
    }
    return o;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace

// one thread per PADDED cell
__global__ void k_param_fwd(int kind, int nz, int nx, int nPml, int nzp, int nxp, const float *__restrict__ A,
                            const float *__restrict__ B, const float *__restrict__ C, const float *__restrict__ A_ref,
                            const float *__restrict__ B_ref, const float *__restrict__ C_ref, const float *__restrict__ Mask,
                            float *__restrict__ Lam, float *__restrict__ Mu, float *__restrict__ Den) {
    const int X = blockIdx.x * blockDim.x + threadIdx.x, Z = blockIdx.y * blockDim.y + threadIdx.y;
    if (X >= nxp || Z >= nzp) return;
    const size_t o = (size_t)Z * nxp + X;
    const size_t s = (size_t)clampi(Z - nPml, 0, nz - 1) * nx + clampi(X - nPml, 0, nx - 1);  // replicate padding
    const float m = Mask[o], w = 1.0f - m;
    const Triple r = map_fwd(kind, m * A[s] + w * A_ref[o], m * B[s] + w * B_ref[o], m * C[s] + w * C_ref[o]);
    Lam[o] = r.a;
    Mu[o] = r.b;
    Den[o] = r.c;
}

// one thread per PHYSICAL cell: the padded cell that is the cell itself
__global__ void k_param_bwd(int kind, int nz, int nx, int nPml, int nzp, int nxp, const float *__restrict__ A,
                            const float *__restrict__ B, const float *__restrict__ C, const float *__restrict__ A_ref,
                            const float *__restrict__ B_ref, const float *__restrict__ C_ref, const float *__restrict__ Mask,
                            const float *__restrict__ gLam, const float *__restrict__ gMu, const float *__restrict__ gDen,
                            float *__restrict__ gA, float *__restrict__ gB, float *__restrict__ gC) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, z = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= nx || z >= nz) return;
    const size_t s = (size_t)z * nx + x, o = (size_t)(z + nPml) * nxp + (x + nPml);
    const float m = Mask[o], w = 1.0f - m;
    const Triple d = map_bwd(kind, m * A[s] + w * A_ref[o], m * B[s] + w * B_ref[o], m * C[s] + w * C_ref[o], gLam[o], gMu[o], gDen[o]);
    gA[s] = m * d.a;
    gB[s] = m * d.b;
    gC[s] = m * d.c;
}

// Transpose of the replicate padding: one WAVE per cell of the physical grid's rim adds the padded cells that replicate
// it (a strip of the padding for an edge cell, a rectangle for the four corners; its own cell is done above).  Lanes
// stride over the cells row-major, then a shuffle tree: a fixed summation order, no atomics.
__global__ void k_param_bwd_rim(int kind, int nz, int nx, int nPml, int nzp, int nxp, const float *__restrict__ A,
                                const float *__restrict__ B, const float *__restrict__ C, const float *__restrict__ A_ref,
                                const float *__restrict__ B_ref, const float *__restrict__ C_ref, const float *__restrict__ Mask,
                                const float *__restrict__ gLam, const float *__restrict__ gMu, const float *__restrict__ gDen,
                                float *__restrict__ gA, float *__restrict__ gB, float *__restrict__ gC) {
    const int lane = threadIdx.x & 63;
    int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // rim cell: top row, bottom row, then the two columns
    int z, x;
    if (r < nx) { z = 0; x = r; }
    else if (r < 2 * nx) { z = nz - 1; x = r - nx; }
    else if (r < 2 * nx + (nz - 2)) { z = r - 2 * nx + 1; x = 0; }
    else if (r < 2 * nx + 2 * (nz - 2)) { z = r - 2 * nx - (nz - 2) + 1; x = nx - 1; }
    else return;
    if (nz == 1 && r >= nx) return;  // a one-row grid has one rim row
    if (nx == 1 && r >= 2 * nx && r >= 2 * nx + (nz - 2)) return;
    const size_t s = (size_t)z * nx + x;
    const int Z0 = (z == 0) ? 0 : z + nPml, Z1 = (z == nz - 1) ? nzp - 1 : z + nPml;
    const int X0 = (x == 0) ? 0 : x + nPml, X1 = (x == nx - 1) ? nxp - 1 : x + nPml;
    const int w_ = X1 - X0 + 1, cnt = (Z1 - Z0 + 1) * w_;
    const float a0 = A[s], b0 = B[s], c0 = C[s];
    float ga = 0.0f, gb = 0.0f, gc = 0.0f;
    for (int k = lane; k < cnt; k += 64) {
        const int Z = Z0 + k / w_, X = X0 + k % w_;
        if (Z == z + nPml && X == x + nPml) continue;  // the cell itself
        const size_t o = (size_t)Z * nxp + X;
        const float m = Mask[o];
        if (m == 0.0f) continue;
        const float w = 1.0f - m;
        const Triple d = map_bwd(kind, m * a0 + w * A_ref[o], m * b0 + w * B_ref[o], m * c0 + w * C_ref[o], gLam[o], gMu[o], gDen[o]);
        ga += m * d.a;
        gb += m * d.b;
        gc += m * d.c;
    }
    for (int off = 32; off > 0; off >>= 1) {
        ga += __shfl_down(ga, off, 64);
        gb += __shfl_down(gb, off, 64);
        gc += __shfl_down(gc, off, 64);
    }
    if (lane == 0) {
        gA[s] += ga;
        gB[s] += gb;
        gC[s] += gc;
    }
}

void launch_param_fwd(hipStream_t st, int kind, int nz, int nx, int nPml, int nPad, const float *A, const float *B, const float *C,
                      const float *A_ref, const float *B_ref, const float *C_ref, const float *Mask, float *Lam, float *Mu,
                      float *Den) {
    const int nzp = nz + 2 * nPml + nPad, nxp = nx + 2 * nPml;
    hipLaunchKernelGGL(k_param_fwd, dim3((nxp + 63) / 64, (nzp + 3) / 4), dim3(64, 4), 0, st, kind, nz, nx, nPml, nzp, nxp, A, B, C,
                       A_ref, B_ref, C_ref, Mask, Lam, Mu, Den);
}

void launch_param_bwd(hipStream_t st, int kind, int nz, int nx, int nPml, int nPad, const float *A, const float *B, const float *C,
                      const float *A_ref, const float *B_ref, const float *C_ref, const float *Mask, const float *gLam,
                      const float *gMu, const float *gDen, float *gA, float *gB, float *gC) {
    const int nzp = nz + 2 * nPml + nPad, nxp = nx + 2 * nPml;
    hipLaunchKernelGGL(k_param_bwd, dim3((nx + 63) / 64, (nz + 3) / 4), dim3(64, 4), 0, st, kind, nz, nx, nPml, nzp, nxp, A, B, C,
                       A_ref, B_ref, C_ref, Mask, gLam, gMu, gDen, gA, gB, gC);
    const int rim = 2 * nx + 2 * (nz > 2 ? nz - 2 : 0);
    hipLaunchKernelGGL(k_param_bwd_rim, dim3((rim + 3) / 4), dim3(256), 0, st, kind, nz, nx, nPml, nzp, nxp, A, B, C, A_ref, B_ref,
                       C_ref, Mask, gLam, gMu, gDen, gA, gB, gC);
}

}  // namespace sepfwi
