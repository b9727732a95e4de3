// kernels_device.hpp -- cell / tile helpers, accumulator and memory-scope policies of the field kernels
// Part of the ONE translation unit kernels.hip (included there, inside namespace sepfwi): the kernels share their bodies as
// inline functions, and every kernel structure must compile them identically (bit-identical results).

using namespace dev;

namespace {

constexpr int BX = 64;              // threads along x  (one wave)
constexpr int MAXT = 1024;          // block = 64 x bz threads, bz in {1..16} (run-time option "bz")

struct Cell {
    int z, x;
    size_t i;  // z*pitch + x
};

// Tile (and, in batched launches, shot) of this block.  Blocks are dealt round-robin to the 8 XCDs (blockIdx % 8 shares an
// L2); with xcd_remap the logical order gives each XCD a contiguous run of logical indices, so z-halo rows are re-read
// from the SAME L2 instead of once per XCD.  Batched launches (g.nb shots in one grid) order the pairs either shot-major
// (all tiles of shot 0, then shot 1, ...) or, shot_fastest, tile-major: the nb shots of one tile are dispatched back to
// back on one XCD, so the media coefficients of the tile (the same for every shot) are fetched from the fabric once
// and hit that XCD's L2 for the other shots.
__device__ __forceinline__ Cell my_cell(const Grid &g, int *shot = nullptr, int block = -1 /* default: blockIdx.x */) {
    Cell c;
    int t = block < 0 ? (int)blockIdx.x : block;
    const int ntile = g.gx * g.gy;
    const int nb = shot ? g.nb : 1;
    if (g.xcd_remap) {
        const int per = (ntile * nb + 7) >> 3;
        t = (t & 7) * per + (t >> 3);
    }
    if (shot) {
        int sh;
        if (g.shot_fastest) {
            const int q = t / nb;
            sh = t - q * nb;
            t = q;
        } else {
            sh = t / ntile;
            t -= sh * ntile;
            if (sh >= nb) {  // surplus block of the remapped numbering
                sh = nb - 1;
                t = ntile;
            }
        }
        *shot = sh;
    }
    const int ty = t / g.gx, tx = t - ty * g.gx;
    c.x = tx * BX + (threadIdx.x & (BX - 1));
    // row is wave-uniform: keep it in an SGPR so the z-profile loads and PML tests are scalar
    c.z = __builtin_amdgcn_readfirstlane(ty * g.bz + (int)(threadIdx.x >> 6));
    if (ty >= g.gy) c.z = g.nz + 1;  // surplus block of the remapped numbering: out of range
    c.i = (size_t)c.z * (size_t)g.pitch + (size_t)c.x;
    return c;
}

// 4-point harmonic mean of mu at the staggered corner (z+1/2, x+1/2): aveMuInit, utilities.cu:124-137.  amu_fly: rebuilt
// from mu (three neighbour taps that hit the cache) instead of streaming a second array; single precision with the
// hardware reciprocal (<= 1 ulp each), i.e. within 4e-7 of the reference's double-precision value.  While the option is
// on, k_model_prep stores exactly THIS value in md.ave_mu as well, so kernels that read the array (the backward ones,
// by default) and kernels that rebuild it see the same bits and reverse-time reconstruction cancels as before.  A zero
// mu gives 1/0 = inf -> 4/inf = 0, the reference's fluid rule.  Valid on [2, n-3]^2 (every cell the kernels update).
__device__ __forceinline__ float ave_mu_at(const Grid &g, const Media &md, size_t i, float mu0) {
    if (g.amu_fly) {
        const float s = (__builtin_amdgcn_rcpf(mu0) + __builtin_amdgcn_rcpf(md.mu[i + g.pitch])) +
                        (__builtin_amdgcn_rcpf(md.mu[i + 1]) + __builtin_amdgcn_rcpf(md.mu[i + g.pitch + 1]));
        return 4.0f * __builtin_amdgcn_rcpf(s);
    }
    return md.ave_mu[i];
}


// Imaging accumulators behind an accessor, so that the same bodies serve the per-step launches (accumulators in HBM, AccG)
// and the persistent time loop (accumulators of the workgroup's own tile in LDS, AccT below).
enum { ACC_LAM = 0, ACC_MU = 1, ACC_XZ = 2, ACC_A = 3, ACC_B = 4 };
template <int K>
__device__ __forceinline__ float *acc_array(const ImgAcc &a) {
    return K == ACC_LAM ? a.lam : K == ACC_MU ? a.mu : K == ACC_XZ ? a.xz : K == ACC_A ? a.a : a.b;
}
struct AccG {
    ImgAcc p;
    template <int K> __device__ __forceinline__ float ld(size_t i) const { return acc_array<K>(p)[i]; }
    template <int K> __device__ __forceinline__ void st(size_t i, float v) const { acc_array<K>(p)[i] = v; }
};
typedef __attribute__((address_space(3))) float lds_float;
// MASK bit K set: accumulator K of this lane's cell lives in LDS at cell[rank of K among the set bits * stride]
template <int MASK>
struct AccT {
    ImgAcc p;
    lds_float *cell;  // this lane's slot of the current row segment
    int stride;       // floats between two LDS-resident accumulator arrays of the tile
    template <int K> __device__ __forceinline__ float ld(size_t i) const {
        if constexpr ((MASK >> K) & 1) return cell[__builtin_popcount(MASK & ((1 << K) - 1)) * stride];
        else return acc_array<K>(p)[i];
    }
    template <int K> __device__ __forceinline__ void st(size_t i, float v) const {
        if constexpr ((MASK >> K) & 1) cell[__builtin_popcount(MASK & ((1 << K) - 1)) * stride] = v;
        else acc_array<K>(p)[i] = v;
    }
};

// How the backward bodies touch the wavefields, the adjoint fields and the C-PML memories.  MemPlain: ordinary loads / stores
// (every per-step launch; inside the persistent loop every row segment whose stencils stay within one XCD's band of rows).
// MemAgent: agent-scope accesses (`sc1`: loads bypass the vector L1 and are served coherently, stores are written through) for
// the persistent loop's segments next to another XCD's band -- the L2s of different XCDs are not coherent with each other.
struct MemPlain {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
struct MemAgent {
    static __device__ __forceinline__ float ld(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ void st(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};

}  // namespace
