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

// ---- rock-physics maps (phi, cc, sw) -> (Lambda [MPa], Mu [MPa], Den): FWI_ops.py:451-497 (Voigt-Reuss-Hill) and :567-611
// (Biot-Gassmann on a consolidation-parameter frame).  Constants as the reference writes them (double products rounded to
// float32 where torch multiplies a float32 tensor by a Python scalar); every operation is the float32 operation of the torch
// expression, in its order.  The chain rule is reverse-mode differentiation of exactly that operation list (what autograd does),
// including its forms of the quotient rule (-g (a/b)/b), the reciprocal rule (-g r r) and the square-root rule (g / (2 r)).
constexpr float RK_KQ = (float)(37.00 * 1e9), RK_KC = (float)(21.00 * 1e9), RK_KW = (float)(2.25 * 1e9), RK_KH = (float)(0.04 * 1e9);
constexpr float RK_MUQ = (float)(44.00 * 1e9), RK_MUC = (float)(10.00 * 1e9);
constexpr float RK_RHOQ = (float)(2.65 * 1e3), RK_RHOC = (float)(2.55 * 1e3), RK_RHOW = (float)(1.00 * 1e3), RK_RHOH = (float)(0.10 * 1e3);
constexpr float RK_CS = 20.0f, RK_CS15 = (float)(1.5 * 20.0), RK_23 = (float)(2. / 3.);

struct VrhMid {  // intermediates of the VRH map that its chain rule needs
    float omp, omc, oms, ks_v, kf_v, r1, r2, kr, ms, rho_f, rho_s;
};
__device__ __forceinline__ Triple vrh_fwd(float a, float b, float c, VrhMid &q) {
    q.omp = 1.0f - a;
    q.omc = 1.0f - b;
    q.oms = 1.0f - c;
    q.ks_v = RK_KC * b + RK_KQ * q.omc;
    q.kf_v = RK_KW * c + RK_KH * q.oms;
    const float kv = q.omp * q.ks_v + a * q.kf_v;                        // FWI_ops.py:463
    q.r1 = b / RK_KC + q.omc / RK_KQ;
    q.r2 = c / RK_KW + q.oms / RK_KH;
    const float kr1 = q.omp * q.r1 + a * q.r2;                          // :464
    q.kr = 1.0f / kr1;
    const float k = 0.5f * (kv + q.kr);
    q.ms = RK_MUC * b + RK_MUQ * q.omc;
    const float mu0 = 0.5f * (q.omp * q.ms + 0.0f);                     // Reuss shear modulus is zero (:468-472)
    q.rho_f = RK_RHOW * c + RK_RHOH * q.oms;
    q.rho_s = RK_RHOC * b + RK_RHOQ * q.omc;
    Triple o;
    o.c = q.rho_f * a + q.rho_s * q.omp;
    o.a = (k - RK_23 * mu0) / 1e6f;                                     // :484-487
    o.b = mu0 / 1e6f;
    return o;
}
__device__ __forceinline__ Triple vrh_bwd(float a, float b, float c, float gl, float gm, float gd) {
    VrhMid q;
    (void)vrh_fwd(a, b, c, q);
    const float gLam0 = gl / 1e6f;
    const float gMu0 = gm / 1e6f + -(gLam0 * RK_23);
    const float gk = gLam0;
    const float gmuv = 0.5f * gMu0;
    const float gkv = 0.5f * gk, gkr = 0.5f * gk;
    const float gkr1 = -gkr * (q.kr * q.kr);
    float gomp = gkv * q.ks_v, ga = gkv * q.kf_v;
    const float gks_v = gkv * q.omp, gkf_v = gkv * a;
    float gb = gks_v * RK_KC, gomc = gks_v * RK_KQ;
    float gc = gkf_v * RK_KW, goms = gkf_v * RK_KH;
    gomp += gkr1 * q.r1;
    ga += gkr1 * q.r2;
    const float gr1 = gkr1 * q.omp, gr2 = gkr1 * a;
    gb += gr1 / RK_KC;
    gomc += gr1 / RK_KQ;
    gc += gr2 / RK_KW;
    goms += gr2 / RK_KH;
    gomp += gmuv * q.ms;
    const float gms = gmuv * q.omp;
    gb += gms * RK_MUC;
    gomc += gms * RK_MUQ;
    const float grho_f = gd * a, grho_s = gd * q.omp;
    ga += gd * q.rho_f;
    gomp += gd * q.rho_s;
    gc += grho_f * RK_RHOW;
    goms += grho_f * RK_RHOH;
    gb += grho_s * RK_RHOC;
    gomc += grho_s * RK_RHOQ;
    Triple o;
    o.a = ga - gomp;
    o.b = gb - gomc;
    o.c = gc - goms;
    return o;
}

struct GasMid {
    float omp, omc, oms, rho_f, k_f, k_s, mu_s, rho_s, d1, q1, k_d, d2, q2, mu_d, e1, e2, e3, e4, e5, e12, dd, denom, u2, u3, u5, num, k_u,
        rho, nA, A, vp, Bq, vs, vp2, vs2, w;
};
__device__ __forceinline__ Triple gas_fwd(float a, float b, float c, GasMid &q) {
    q.oms = 1.0f - c;
    q.omc = 1.0f - b;
    q.omp = 1.0f - a;
    q.rho_f = RK_RHOW * c + RK_RHOH * q.oms;                            // FWI_ops.py:585-589
    q.k_f = RK_KW * c + RK_KH * q.oms;
    q.k_s = RK_KC * b + RK_KQ * q.omc;
    q.mu_s = RK_MUC * b + RK_MUQ * q.omc;
    q.rho_s = RK_RHOC * b + RK_RHOQ * q.omc;
    q.d1 = 1.0f + RK_CS * a;
    q.q1 = q.omp / q.d1;
    q.k_d = q.k_s * q.q1;                                               // :591
    q.d2 = 1.0f + RK_CS15 * a;
    q.q2 = q.omp / q.d2;
    q.mu_d = q.mu_s * q.q2;                                             // :592
    q.e1 = q.omp / a;
    q.e2 = q.k_f / q.k_s;
    q.e3 = q.k_s - q.k_s * a;
    q.e4 = q.k_d / q.e3;
    q.e5 = 1.0f - q.e4;
    q.e12 = q.e1 * q.e2;
    const float Delta = q.e12 * q.e5;                                   // :594
    q.dd = 1.0f + Delta;
    q.denom = a * q.dd;                                                 // :596
    q.u2 = 1.0f + a;
    q.u3 = q.k_d / q.k_s;
    q.u5 = 1.0f - q.u2 * q.u3;
    q.num = a * q.k_d + q.u5 * q.k_f;
    q.k_u = q.num / q.denom;                                            // :598
    q.rho = q.rho_f * a + q.rho_s * q.omp;                              // :602
    q.nA = q.k_u + 0.75f * q.mu_d;
    q.A = q.nA / q.rho;
    q.vp = sqrtf(q.A);                                                  // :603 (0.75, as the reference has it)
    q.Bq = q.mu_d / q.rho;
    q.vs = sqrtf(q.Bq);
    q.vp2 = q.vp * q.vp;
    q.vs2 = q.vs * q.vs;
    q.w = q.vp2 - q.vs2 * 2.0f;
    Triple o;
    o.a = q.rho * q.w / 1e6f;                                           // :609-611
    o.b = q.rho * q.vs2 / 1e6f;
    o.c = q.rho;
    return o;
}
__device__ __forceinline__ Triple gas_bwd(float a, float b, float c, float gl, float gm, float gd) {
    GasMid q;
    (void)gas_fwd(a, b, c, q);
    const float gL0 = gl / 1e6f, gM0 = gm / 1e6f;
    float grho = gL0 * q.w;
    const float gw = gL0 * q.rho;
    const float gvp2 = gw, gvs2 = -gw * 2.0f;
    grho += gM0 * q.vs2;
    const float gvs2b = gM0 * q.rho;
    grho += gd;
    const float gvp = gvp2 * (2.0f * q.vp);
    const float gvs = gvs2 * (2.0f * q.vs) + gvs2b * (2.0f * q.vs);
    const float gA = gvp / (2.0f * q.vp);
    const float gnA = gA / q.rho;
    grho += -gA * (q.A / q.rho);
    const float gk_u = gnA;
    float gmu_d = gnA * 0.75f;
    const float gBq = gvs / (2.0f * q.vs);
    gmu_d += gBq / q.rho;
    grho += -gBq * (q.Bq / q.rho);
    const float grho_f = grho * a, grho_s = grho * q.omp;
    float ga = grho * q.rho_f, gomp = grho * q.rho_s;
    const float gnum = gk_u / q.denom;
    const float gdenom = -gk_u * (q.k_u / q.denom);
    const float gu5 = gnum * q.k_f;
    float gk_f = gnum * q.u5;
    const float gu4 = -gu5;
    const float gu2 = gu4 * q.u3, gu3 = gu4 * q.u2;
    float gk_d = gu3 / q.k_s;
    float gk_s = -gu3 * (q.u3 / q.k_s);
    ga += gu2;
    ga += gnum * q.k_d;
    gk_d += gnum * a;
    ga += gdenom * q.dd;
    const float gDelta = gdenom * a;
    const float ge12 = gDelta * q.e5, ge5 = gDelta * q.e12;
    const float ge1 = ge12 * q.e2, ge2 = ge12 * q.e1;
    const float ge4 = -ge5;
    gk_d += ge4 / q.e3;
    const float ge3 = -ge4 * (q.e4 / q.e3);
    gk_s += ge3;
    const float gksa = -ge3;
    gk_s += gksa * a;
    ga += gksa * q.k_s;
    gk_f += ge2 / q.k_s;
    gk_s += -ge2 * (q.e2 / q.k_s);
    gomp += ge1 / a;
    ga += -ge1 * (q.e1 / a);
    const float gmu_s = gmu_d * q.q2, gq2 = gmu_d * q.mu_s;
    gomp += gq2 / q.d2;
    ga += (-gq2 * (q.q2 / q.d2)) * RK_CS15;
    gk_s += gk_d * q.q1;
    const float gq1 = gk_d * q.k_s;
    gomp += gq1 / q.d1;
    ga += (-gq1 * (q.q1 / q.d1)) * RK_CS;
    float gb = grho_s * RK_RHOC, gomc = grho_s * RK_RHOQ;
    gb += gmu_s * RK_MUC;
    gomc += gmu_s * RK_MUQ;
    gb += gk_s * RK_KC;
    gomc += gk_s * RK_KQ;
    float gc = gk_f * RK_KW, goms = gk_f * RK_KH;
    gc += grho_f * RK_RHOW;
    goms += grho_f * RK_RHOH;
    Triple o;
    o.a = ga - gomp;
    o.b = gb - gomc;
    o.c = gc - goms;
    return o;
}

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
        case PARAM_VP_VS_IS:  // FWI_ops.py:389-391   (a = Vp, b = Vs, c = IS)
            o.a = c / b * (a * a) - 2.0f * c * b;
            o.b = c * b;
            o.c = c / b;
            break;
        case PARAM_ROCK_VRH: {  // (a, b, c) = (phi, cc, sw)
            VrhMid q;
            o = vrh_fwd(a, b, c, q);
            break;
        }
        default: {  // PARAM_ROCK_GASSMANN
            GasMid q;
            o = gas_fwd(a, b, c, q);
            break;
        }
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
        case PARAM_VP_VS_IS: {
            const float r = a / b;  // Vp / Vs
            o.a = gl * (2.0f * c * r);
            o.b = gl * (-c * r * r - 2.0f * c) + gm * c - gd * (c / (b * b));
            o.c = gl * (a * r - 2.0f * b) + gm * b + gd / b;
            break;
        }
        case PARAM_ROCK_VRH:
            o = vrh_bwd(a, b, c, gl, gm, gd);
            break;
        default:  // PARAM_ROCK_GASSMANN
            o = gas_bwd(a, b, c, gl, gm, gd);
            break;
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
