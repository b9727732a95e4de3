// kernels_aux.hpp -- receivers, injection, residual, transpose, model preparation, gradient finalisation
// Part of the ONE translation unit kernels.hip (included there, inside namespace sepfwi): the kernels share their bodies as
// inline functions, and every kernel structure must compile them identically (bit-identical results).

// ---------------------------------------------------------------------------------------------
// receivers.  Seismograms are kept time-major on the device: d[comp][it][rec]  (coalesced for a
// horizontal fibre); they are transposed to the reference's [rec][it] files only on export.
// comps bit mask: 1 pressure, 2 vx, 4 vz, 8 ett.
// ---------------------------------------------------------------------------------------------
__global__ void k_record(Grid g, Fields f, int nrec, const int *__restrict__ rec_idx /* z*pitch+x */,
                         float *__restrict__ d_pr, float *__restrict__ d_vx, float *__restrict__ d_vz,
                         float *__restrict__ d_ett, int comps, int fiber, const float *__restrict__ sens) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    const int i = rec_idx[r];
    if (comps & 1) d_pr[r] = f.szz[i] + f.sxx[i];
    const float vx = f.vx[i];
    if (comps & 2) d_vx[r] = vx;
    const float vz = f.vz[i];
    if (comps & 4) d_vz[r] = vz;
    if (!(comps & 8)) return;
    if (sens) {
        // directional channel: ett = s0 exx + s3 ezz + s1 exz (MOD/elasticSolver.py:266-276), strains as one-cell differences in
        // units of "strain x dx" like recording_exx (the z-differences carry dx/dz)
        const float k = g.dx * g.rdz;
        const float exx = vx - f.vx[i - 1];
        const float ezz = (vz - f.vz[i - g.pitch]) * k;
        const float exz = 0.5f * ((f.vx[i + g.pitch] - vx) * k + (f.vz[i + 1] - vz));
        d_ett[r] = sens[3 * r] * exx + sens[3 * r + 1] * ezz + sens[3 * r + 2] * exz;
        return;
    }
    // axial strain over one cell, not divided by the spacing (utilities.cu:600-601): exx for a horizontal fibre,
    // ezz (recording_ezz, utilities.cu:620-629) for a vertical one
    d_ett[r] = fiber ? vz - f.vz[i - g.pitch] : vx - f.vx[i - 1];
}

// res_injection_exx: vx_adj(z,x) += r ; vx_adj(z,x-1) -= r.  Adjacent channels share cells, so the
// two statements are applied through float atomics (the reference's plain +=/-= is racy there,
// utilities.cu:613-614).  Atomic order only permutes a few adds per cell.  With `sens`: the transpose of the
// directional channel above.
__global__ void k_inject(Fields adj, int nrec, const int *__restrict__ rec_idx, const float *__restrict__ res_t,
                         int down /* 0: horizontal fibre, else the pitch: vertical fibre (res_injection_ezz, utilities.cu:632-641) */,
                         const float *__restrict__ sens, int pitch, float dx_dz) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    const int i = rec_idx[r];
    const float v = res_t[r];
    if (sens) {
        const float a = sens[3 * r] * v, b = sens[3 * r + 1] * dx_dz * v, c = 0.5f * sens[3 * r + 2] * v;
        atomicAdd(&adj.vx[i], a);
        atomicAdd(&adj.vx[i - 1], -a);
        atomicAdd(&adj.vz[i], b);
        atomicAdd(&adj.vz[i - pitch], -b);
        atomicAdd(&adj.vx[i + pitch], c * dx_dz);
        atomicAdd(&adj.vx[i], -(c * dx_dz));
        atomicAdd(&adj.vz[i + 1], c);
        atomicAdd(&adj.vz[i], -c);
    } else if (down) {
        atomicAdd(&adj.vz[i], v);
        atomicAdd(&adj.vz[i - down], -v);
    } else {
        atomicAdd(&adj.vx[i], v);
        atomicAdd(&adj.vx[i - 1], -v);
    }
}

// Batched forms (one launch for every shot of a batch, blockIdx.y = shot; ShotDev table): on grids small enough for the batched schedule
// a launch per shot and time step for the receivers alone made a vertical fibre three times slower than a horizontal one.
__device__ __forceinline__ void record_one(const Grid &g, const Fields &f, int i, int r, float *d_pr, float *d_vx, float *d_vz, float *d_ett, int comps, int fiber,
                                           const float *sens) {
    if (comps & 1) d_pr[r] = f.szz[i] + f.sxx[i];
    const float vx = f.vx[i];
    if (comps & 2) d_vx[r] = vx;
    const float vz = f.vz[i];
    if (comps & 4) d_vz[r] = vz;
    if (!(comps & 8)) return;
    if (sens) {
        const float k = g.dx * g.rdz;
        const float exx = vx - f.vx[i - 1];
        const float ezz = (vz - f.vz[i - g.pitch]) * k;
        const float exz = 0.5f * ((f.vx[i + g.pitch] - vx) * k + (f.vz[i + 1] - vz));
        d_ett[r] = sens[3 * r] * exx + sens[3 * r + 1] * ezz + sens[3 * r + 2] * exz;
        return;
    }
    d_ett[r] = fiber ? vz - f.vz[i - g.pitch] : vx - f.vx[i - 1];
}
// seismogram column `column` of every shot of the batch whose channels are NOT sampled inside k_stress (comps bit 16)
__global__ void k_record_batch(Grid g, const ShotDev *__restrict__ shots, size_t n, size_t data_len, int column) {
    const ShotDev &s = shots[blockIdx.y];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if ((s.comps & 16) || r >= s.nrec) return;
    const Fields f{s.fields, s.fields + n, s.fields + 2 * n, s.fields + 3 * n, s.fields + 4 * n};
    float *col = s.syn + (size_t)column * (size_t)s.nrec;
    record_one(g, f, s.rec[r], r, col, col + data_len, col + 2 * data_len, col + 3 * data_len, s.comps, g.fiber, s.sens);
}
// residual column `it` of every shot of the batch whose channels are not a fused line (lr_n == 0), as k_inject
__global__ void k_inject_batch(Grid g, const ShotDev *__restrict__ shots, size_t n, int it) {
    const ShotDev &s = shots[blockIdx.y];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (s.lr_n != 0 || r >= s.nrec) return;
    float *avz = s.adj, *avx = s.adj + n;
    const int i = s.rec[r], P = g.pitch;
    const float v = s.res[(size_t)it * (size_t)s.nrec + r], dx_dz = g.dx * g.rdz;
    if (s.sens) {
        const float a = s.sens[3 * r] * v, b = s.sens[3 * r + 1] * dx_dz * v, c = 0.5f * s.sens[3 * r + 2] * v;
        atomicAdd(&avx[i], a);
        atomicAdd(&avx[i - 1], -a);
        atomicAdd(&avz[i], b);
        atomicAdd(&avz[i - P], -b);
        atomicAdd(&avx[i + P], c * dx_dz);
        atomicAdd(&avx[i], -(c * dx_dz));
        atomicAdd(&avz[i + 1], c);
        atomicAdd(&avz[i], -c);
    } else if (g.fiber) {
        atomicAdd(&avz[i], v);
        atomicAdd(&avz[i - P], -v);
    } else {
        atomicAdd(&avx[i], v);
        atomicAdd(&avx[i - 1], -v);
    }
}

// The adjoint source of the persistent loop for general receivers (inject_plan.hpp): one value per target cell and time step,
//   val[it][t] = sum_e w_e res[it][rec_e]   over the target's entries in channel order
// -- what k_inject's atomics add to that cell in that step, in a fixed order.  One thread per (target, time step).
__global__ void k_inject_values(const float *__restrict__ res, int nrec, const int *__restrict__ tgt_start, const int *__restrict__ ent_rec,
                                const float *__restrict__ ent_w, int ntgt, float *__restrict__ val) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, it = blockIdx.y;
    if (t >= ntgt) return;
    const float *r = res + (size_t)it * (size_t)nrec;
    float s = 0.0f;
    for (int e = tgt_start[t]; e < tgt_start[t + 1]; e++) s += ent_w[e] * r[ent_rec[e]];
    val[(size_t)it * (size_t)ntgt + t] = s;
}

// residual r = obs - syn (time sample 0 forced to 0) and sum r^2, all time-major [it][rec].
// One double partial per block -> atomicAdd(double).
__global__ void k_residual(const float *__restrict__ obs, const float *__restrict__ syn, float *__restrict__ res,
                           int nrec, long long n, double *__restrict__ sumsq) {
    double s = 0.0;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        float r = (k < nrec) ? 0.0f : (obs[k] - syn[k]);  // first time sample: utilities.cu:159-163
        res[k] = r;
        s += (double)r * (double)r;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double part[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) part[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) t += part[k];
        atomicAdd(sumsq, t);
    }
}

// [rows][cols] -> [cols][rows] through a padded LDS tile (used for seismogram import/export).
__global__ void k_transpose(const float *__restrict__ in, float *__restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    int c = blockIdx.x * 32 + threadIdx.x, r0 = blockIdx.y * 32;
    for (int k = threadIdx.y; k < 32; k += blockDim.y) {
        int r = r0 + k;
        if (r < rows && c < cols) tile[k][threadIdx.x] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    int orow0 = blockIdx.x * 32, oc = blockIdx.y * 32 + threadIdx.x;
    for (int k = threadIdx.y; k < 32; k += blockDim.y) {
        int orow = orow0 + k;
        if (orow < cols && oc < rows) out[(size_t)orow * rows + oc] = tile[threadIdx.x][k];
    }
}

// ---------------------------------------------------------------------------------------------
// media preparation: boundary arrays (nz, nx) dense [MPa] -> internal pitched arrays [Pa] + averages
// ---------------------------------------------------------------------------------------------
__global__ void k_model_prep(Grid g, const float *__restrict__ Lam_in, const float *__restrict__ Mu_in,
                             const float *__restrict__ Den_in, float *__restrict__ lam, float *__restrict__ mu,
                             float *__restrict__ ave_mu, float *__restrict__ byc_a, float *__restrict__ byc_b,
                             float *__restrict__ rho, unsigned int *__restrict__ cp2_max_bits, int amu_fly) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int z = blockIdx.y * blockDim.y + threadIdx.y;
    float cp2 = 0.0f;
    if (x < g.nx && z < g.nz) {
        const size_t si = (size_t)z * g.nx + x;
        const float L = (float)((double)Lam_in[si] * 1e6);  // libCUFD.cu:73-74
        const float M = (float)((double)Mu_in[si] * 1e6);
        const float D = Den_in[si];
        cp2 = (float)(((double)L + 2.0 * (double)M) / (double)D);  // velInit, utilities.cu:119-120 (squared)
        if (z < g.nzc) {
            const size_t i = (size_t)z * g.pitch + x;
            lam[i] = L;
            mu[i] = M;
            rho[i] = D;
            float am = 0.0f, ba = 1.0f / 1000.0f, bb = 1.0f / 1000.0f;  // Model.cu:67,72-73
            // averages exist on [2, n-3] of the FULL padded grid (utilities.cu:129,146); rows >= nzc are
            // never read by any kernel.
            if (z >= 2 && z <= g.nz - 3 && x >= 2 && x <= g.nx - 3) {
                const double a = M;
                const double b = (double)Mu_in[si + g.nx] * 1e6;
                const double c = (double)Mu_in[si + 1] * 1e6;
                const double d = (double)Mu_in[si + g.nx + 1] * 1e6;
                const float bf = (float)b, cf = (float)c, df = (float)d;
                if (amu_fly)  // the value the stress kernels rebuild on the fly (ave_mu_at): ONE definition of the average per session
                    am = 4.0f * __builtin_amdgcn_rcpf((__builtin_amdgcn_rcpf(M) + __builtin_amdgcn_rcpf(bf)) +
                                                      (__builtin_amdgcn_rcpf(cf) + __builtin_amdgcn_rcpf(df)));
                else if (!(M == 0.0f || bf == 0.0f || cf == 0.0f || df == 0.0f))
                    am = (float)(4.0 / (1.0 / a + 1.0 / (double)bf + 1.0 / (double)cf + 1.0 / (double)df));
                ba = (float)(2.0 / (double)(Den_in[si + g.nx] + D));
                bb = (float)(2.0 / (double)(Den_in[si + 1] + D));
            }
            ave_mu[i] = am;
            byc_a[i] = ba;
            byc_b[i] = bb;
        }
    }
    // max over the whole padded grid for the Courant guard (utilities.cu:225-232).  Cp^2 > 0, so the
    // float bit pattern orders like the value.
    for (int off = 32; off > 0; off >>= 1) cp2 = fmaxf(cp2, __shfl_down(cp2, off, 64));
    if ((threadIdx.y * blockDim.x + threadIdx.x) % 64 == 0 && cp2 > 0.0f) atomicMax(cp2_max_bits, __float_as_uint(cp2));
}

// ---------------------------------------------------------------------------------------------
// gradient finalisation (once per call): gather form of the reference's sprays, written straight
// into the boundary layout (nz, nx) dense -- rows >= nzc are zero.
//   el_stress.cu:108-123 :  gLam = MEGA*acc.lam ; gMu = MEGA*acc.mu + sum_p S(p)/mu(z,x)^2 over the
//       staggered points p in {(z,x),(z-1,x),(z,x-1),(z-1,x-1)} that sprayed onto (z,x)
//   el_velocity.cu:101-110: gDen = A(z,x)+B(z,x)+A(z-1,x)+B(z,x-1), A = acc.a*(-byc_a^2/2), ...
// including the reference's edge tests (the x+1 spray is unconditional).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float xz_weight(const Grid &g, const Media &md, const ImgAcc &acc, int pz, int px) {
    if (pz < g.nPml || pz > g.zmax || px < g.nPml || px > g.xmax) return 0.0f;
    const size_t p = (size_t)pz * g.pitch + px;
    const float am = md.ave_mu[p];
    if (am == 0.0f) return 0.0f;
    const double h = 1.0 / (double)md.mu[p] + 1.0 / (double)md.mu[p + g.pitch] + 1.0 / (double)md.mu[p + 1] +
                     1.0 / (double)md.mu[p + g.pitch + 1];
    return (float)((double)(acc.xz[p] * am) / h * 1e6);
}

__global__ void k_finalize_gradients(Grid g, Media md, ImgAcc acc, float *__restrict__ gLam, float *__restrict__ gMu,
                                     float *__restrict__ gDen) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int z = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= g.nx || z >= g.nz) return;
    const size_t o = (size_t)z * g.nx + x;
    float gl = 0.0f, gm = 0.0f, gd = 0.0f;
    // targets can lie one column right of the interior (always-true x test of the reference)
    if (z >= g.nPml && z <= g.zmax && x >= g.nPml && x <= g.xmax + 1) {
        const size_t i = (size_t)z * g.pitch + x;
        const bool inside = (x <= g.xmax);
        if (inside) {
            gl = (float)((double)acc.lam[i] * 1e6);
            gm = (float)((double)acc.mu[i] * 1e6);
        }
        // A fluid cell (mu = 0) makes every corner around it a zero-average one, which sprays nothing (el_stress.cu:112: the spray
        // is inside `if (ave_Mu != 0)`); its 1/mu^2 = inf must not meet those zero weights (inf * 0 = NaN).
        const double rmu2 = 1.0 / ((double)md.mu[i] * (double)md.mu[i]);
        auto spray = [&](int pz, int px) {
            const float s = xz_weight(g, md, acc, pz, px);
            if (s != 0.0f) gm += (float)(rmu2 * (double)s);
        };
        spray(z, x);          // own corner
        spray(z - 1, x);      // sprayed down (z+1<=zmax holds: target z<=zmax)
        spray(z, x - 1);      // sprayed right, unconditional
        if (inside) spray(z - 1, x - 1);
        // density
        auto A = [&](int pz, int px) -> float {
            if (pz < g.nPml || pz > g.zmax || px < g.nPml || px > g.xmax) return 0.0f;
            const size_t p = (size_t)pz * g.pitch + px;
            const double b = md.byc_a[p];
            return (float)((double)acc.a[p] * (-(b * b) / 2.0));
        };
        auto B = [&](int pz, int px) -> float {
            if (pz < g.nPml || pz > g.zmax || px < g.nPml || px > g.xmax) return 0.0f;
            const size_t p = (size_t)pz * g.pitch + px;
            const double b = md.byc_b[p];
            return (float)((double)acc.b[p] * (-(b * b) / 2.0));
        };
        gd = A(z, x) + B(z, x) + A(z - 1, x) + B(z, x - 1);
    }
    gLam[o] = gl;
    gMu[o] = gm;
    gDen[o] = gd;
}
