// device_common.hpp -- device-side helpers shared by the gfx950 kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include "fwi_types.hpp"

namespace sepfwi {
namespace dev {

constexpr float C1 = 9.0f / 8.0f;   // el_stress.cu:42
constexpr float C2 = 1.0f / 24.0f;  // el_stress.cu:43
// backward-staggered first derivative D-:  (c1 (f0 - fm1) - c2 (fp1 - fm2)) / h
__device__ __forceinline__ float dminus(float fm2, float fm1, float f0, float fp1, float rh) {
    return (C1 * (f0 - fm1) - C2 * (fp1 - fm2)) * rh;
}
// forward-staggered first derivative D+:   (c1 (fp1 - f0) - c2 (fp2 - fm1)) / h
__device__ __forceinline__ float dplus(float fm1, float f0, float fp1, float fp2, float rh) {
    return (C1 * (fp1 - f0) - C2 * (fp2 - fm1)) * rh;
}

// Slot of cell (z,x) in the packed boundary frame, or -1.  The frame is the 5-cell-thick ring
// rows/cols [nPml-2, nPml+2] U [n-nPml-3, n-nPml+1] of the reference (utilities.cu:362-392) without
// its duplicated corners:  [top 5 rows][bottom 5 rows][middle rows: 5 left + 5 right cells].
__device__ __forceinline__ int frame_slot(const Grid &g, int z, int x) {
    const int zf = z - (g.nPml - 2), xf = x - (g.nPml - 2);
    if (zf < 0 || zf >= g.nzBnd || xf < 0 || xf >= g.nxBnd) return -1;
    if (zf < 5) return zf * g.nxBnd + xf;
    if (zf >= g.nzBnd - 5) return (5 + zf - (g.nzBnd - 5)) * g.nxBnd + xf;
    const int base = 10 * g.nxBnd + (zf - 5) * 10;
    if (xf < 5) return base + xf;
    if (xf >= g.nxBnd - 5) return base + 5 + (xf - (g.nxBnd - 5));
    return -1;
}

__device__ __forceinline__ bool in_pml_z(const Grid &g, int z) { return z < g.nPml || z > g.nzc - g.nPml - 1; }

// ---------------------------------------------------------------------------------------------
// Layouts of the big arrays.  A propagation state (forward or adjoint) is five wavefields of n floats each; the imaging
// accumulators are five arrays of n floats.  Two layouts of the same 5 n floats (option "pair", one bit per group):
//   P = false   planar:   [vz | vx | szz | sxx | sxz]                 [lam | mu | xz | a | b]
//   P = true    the members every kernel taps at IDENTICAL offsets interleaved, so that one 8-byte load per lane brings
//               both:     [(vz,vx) x n | (szz,sxx) x n | sxz]         [(lam,mu) x n | xz | (a,b) x n]
// The kernel bodies are written once in terms of the pair accessors v(i), s(i), lm(i), ab(i); with the planar layout a
// pair access is two 4-byte loads of which the compiler drops the one whose component is not used.  Pure layout: the
// arithmetic, its order and therefore every result bit are the same in both.
// ---------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool P>
struct F5 {
    float *b;
    size_t n;
    __device__ __forceinline__ f2 v(size_t i) const {
        if constexpr (P) return ((const f2 *)b)[i];
        else { f2 r; r.x = b[i]; r.y = b[n + i]; return r; }
    }
    __device__ __forceinline__ f2 s(size_t i) const {
        if constexpr (P) return ((const f2 *)(b + 2 * n))[i];
        else { f2 r; r.x = b[2 * n + i]; r.y = b[3 * n + i]; return r; }
    }
    __device__ __forceinline__ float sxz(size_t i) const { return b[4 * n + i]; }
    __device__ __forceinline__ void set_v(size_t i, float vz, float vx) const {
        if constexpr (P) { f2 r; r.x = vz; r.y = vx; ((f2 *)b)[i] = r; }
        else { b[i] = vz; b[n + i] = vx; }
    }
    __device__ __forceinline__ void set_s(size_t i, float szz, float sxx) const {
        if constexpr (P) { f2 r; r.x = szz; r.y = sxx; ((f2 *)(b + 2 * n))[i] = r; }
        else { b[2 * n + i] = szz; b[3 * n + i] = sxx; }
    }
    __device__ __forceinline__ void set_sxz(size_t i, float x) const { b[4 * n + i] = x; }
    // addresses of single members (atomics of k_inject)
    __device__ __forceinline__ float *p_vz(size_t i) const { return P ? b + 2 * i : b + i; }
    __device__ __forceinline__ float *p_vx(size_t i) const { return P ? b + 2 * i + 1 : b + n + i; }
};

// Imaging accumulators (fwi_types.hpp ImgAcc).  NT: non-temporal loads / stores -- every accumulator cell is touched exactly
// once per time step (option "acc_nt").
template <bool P>
struct Acc5 {
    float *b;
    size_t n;
    int nt;
    __device__ __forceinline__ static float ld(const float *p, int nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
    __device__ __forceinline__ static f2 ld2(const f2 *p, int nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
    __device__ __forceinline__ static void st(float *p, float x, int nt) { if (nt) __builtin_nontemporal_store(x, p); else *p = x; }
    __device__ __forceinline__ static void st2(f2 *p, f2 x, int nt) { if (nt) __builtin_nontemporal_store(x, p); else *p = x; }
    __device__ __forceinline__ f2 lm(size_t i) const {
        if constexpr (P) return ld2((const f2 *)b + i, nt);
        else { f2 r; r.x = ld(b + i, nt); r.y = ld(b + n + i, nt); return r; }
    }
    __device__ __forceinline__ float xz(size_t i) const { return ld(b + 2 * n + i, nt); }
    __device__ __forceinline__ f2 ab(size_t i) const {
        if constexpr (P) return ld2((const f2 *)(b + 3 * n) + i, nt);
        else { f2 r; r.x = ld(b + 3 * n + i, nt); r.y = ld(b + 4 * n + i, nt); return r; }
    }
    __device__ __forceinline__ void set_lm(size_t i, float lam, float mu) const {
        if constexpr (P) { f2 r; r.x = lam; r.y = mu; st2((f2 *)b + i, r, nt); }
        else { st(b + i, lam, nt); st(b + n + i, mu, nt); }
    }
    __device__ __forceinline__ void set_xz(size_t i, float x) const { st(b + 2 * n + i, x, nt); }
    __device__ __forceinline__ void set_ab(size_t i, float a, float bb) const {
        if constexpr (P) { f2 r; r.x = a; r.y = bb; st2((f2 *)(b + 3 * n) + i, r, nt); }
        else { st(b + 3 * n + i, a, nt); st(b + 4 * n + i, bb, nt); }
    }
};


}  // namespace dev
}  // namespace sepfwi
