/*
 * numba_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle + timed CPU baseline; never on the
 * product path).
 *
 * Plain-C float64 restatement of the reference's CPU propagator
 * DAS_Waveform_Modeling/src/elasticSolver.py (class elasticSolver, Numba-jitted kernels):
 *   update_velocity   elasticSolver.py:310-345
 *   update_stress     elasticSolver.py:348-386
 *   forward_it        elasticSolver.py:185-305  (time loop, sponge, source, recorders)
 *   damping profile   elasticSolver.py:74-79
 * Arrays are [i = x][j = z] row-major (z contiguous), exactly as the NumPy arrays of the
 * reference.  Same loop nests and operation order; build with -ffp-contract=off.
 *
 * PARITY PINNING: checked against golden traces produced by importing the reference module
 * itself in the build container (scripts/make_golden_numba.py -> tests/golden/numba_*.npz).
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define A(a, i, j) a[(size_t)(i) * (size_t)nz + (size_t)(j)]

/* elasticSolver.py:74-79 / :389-403 : sin^2 sponge, multiplicative over the four edges */
void onb_damp_profile(double *damp, int nx, int nz, int ndamp)
{
    for (size_t k = 0; k < (size_t)nx * (size_t)nz; k++) damp[k] = 1.0;
    for (int i = 0; i < ndamp; i++) {
        double s = sin(M_PI / 2 * i / ndamp);
        double w = s * s;
        for (int j = 0; j < nz; j++) { A(damp, i, j) *= w; }
        for (int j = 0; j < nz; j++) { A(damp, nx - i - 1, j) *= w; }
        for (int k = 0; k < nx; k++) { A(damp, k, i) *= w; }
        for (int k = 0; k < nx; k++) { A(damp, k, nz - i - 1) *= w; }
    }
}

/* elasticSolver.py:310-345 */
void onb_update_velocity(double *vx, double *vz, const double *sxx, const double *szz, const double *sxz,
                         int nx, int nz, double dx, double dz, double dt, const double *rho)
{
    const double c1 = 9.0 / 8.0, c2 = 1.0 / 24.0;
    for (int i = 2; i < nx - 2; i++) {
        for (int j = 2; j < nz - 2; j++) {
            double rhox = 0.5 * (A(rho, i, j) + A(rho, i + 1, j));
            double rhoz = 0.5 * (A(rho, i, j) + A(rho, i, j + 1));
            double szz_z = (c1 * (A(szz, i, j + 1) - A(szz, i, j)) - c2 * (A(szz, i, j + 2) - A(szz, i, j - 1))) / dz;
            double sxz_x = (c1 * (A(sxz, i, j) - A(sxz, i - 1, j)) - c2 * (A(sxz, i + 1, j) - A(sxz, i - 2, j))) / dx;
            double sxz_z = (c1 * (A(sxz, i, j) - A(sxz, i, j - 1)) - c2 * (A(sxz, i, j + 1) - A(sxz, i, j - 2))) / dz;
            double sxx_x = (c1 * (A(sxx, i + 1, j) - A(sxx, i, j)) - c2 * (A(sxx, i + 2, j) - A(sxx, i - 1, j))) / dx;
            A(vx, i, j) += (sxz_z + sxx_x) * dt / rhoz;
            A(vz, i, j) += (szz_z + sxz_x) * dt / rhox;
        }
    }
}

/* elasticSolver.py:348-386 */
void onb_update_stress(const double *vx, const double *vz, double *sxx, double *szz, double *sxz,
                       int nx, int nz, double dx, double dz, double dt, const double *lam, const double *mu)
{
    const double c1 = 9.0 / 8.0, c2 = 1.0 / 24.0;
    for (int i = 2; i < nx - 2; i++) {
        for (int j = 2; j < nz - 2; j++) {
            double muxz;
            if (A(mu, i, j) != 0.0 && A(mu, i + 1, j) != 0.0 && A(mu, i, j + 1) != 0.0 && A(mu, i + 1, j + 1) != 0.0)
                muxz = 4.0 / (1 / A(mu, i, j) + 1 / A(mu, i + 1, j) + 1 / A(mu, i, j + 1) + 1 / A(mu, i + 1, j + 1));
            else
                muxz = 0.0;
            double vzz = (c1 * (A(vz, i, j) - A(vz, i, j - 1)) - c2 * (A(vz, i, j + 1) - A(vz, i, j - 2))) / dz;
            double vxx = (c1 * (A(vx, i, j) - A(vx, i - 1, j)) - c2 * (A(vx, i + 1, j) - A(vx, i - 2, j))) / dx;
            double vxz = (c1 * (A(vx, i, j + 1) - A(vx, i, j)) - c2 * (A(vx, i, j + 2) - A(vx, i, j - 1))) / dz;
            double vzx = (c1 * (A(vz, i + 1, j) - A(vz, i, j)) - c2 * (A(vz, i + 2, j) - A(vz, i - 1, j))) / dx;
            A(szz, i, j) += ((A(lam, i, j) + 2 * A(mu, i, j)) * vzz + A(lam, i, j) * vxx) * dt;
            A(sxx, i, j) += (A(lam, i, j) * vzz + (A(lam, i, j) + 2 * A(mu, i, j)) * vxx) * dt;
            A(sxz, i, j) += (vxz + vzx) * muxz * dt;
        }
    }
}

/*
 * elasticSolver.py:185-305 (forward_it) for one shot.
 * nx, nz include the 2*ndamp sponge cells; lam/mu/rho are the padded (nx,nz) arrays
 * (elasticSolver.py:45-47,60-61); grids are already shifted by +ndamp (:82-84).
 * geo_* : vx, vz, pr records  [geo_num][nt];   das_* : exx, ezz, exz, ett  [das_num][nt].
 * das_sens: [das_num][6] (exx, exy, exz, eyy, eyz, ezz) -- ett uses columns 0, 3, 1 exactly as
 * the reference does (:276).
 */
void onb_forward_shot(int nx, int nz, int ndamp, double dx, double dz, double dt, int nt,
                      const double *lam, const double *mu, const double *rho, const double *stf,
                      int src_ix, int src_iz,
                      int geo_num, const int *geo_ix, const int *geo_iz,
                      int das_num, const int *das_ix, const int *das_iz, const double *das_sens,
                      double *geoVx, double *geoVz, double *geoPr,
                      double *dasExx, double *dasEzz, double *dasExz, double *dasEtt)
{
    size_t n = (size_t)nx * (size_t)nz;
    double *vx = (double *)calloc(5 * n, sizeof(double));
    double *vz = vx + n, *sxx = vx + 2 * n, *szz = vx + 3 * n, *sxz = vx + 4 * n;
    double *damp = (double *)malloc(n * sizeof(double));
    onb_damp_profile(damp, nx, nz, ndamp);
    for (int it = 0; it < nt; it++) {
        onb_update_velocity(vx, vz, sxx, szz, sxz, nx, nz, dx, dz, dt, rho);
        for (size_t k = 0; k < n; k++) { vx[k] *= damp[k]; vz[k] *= damp[k]; }
        onb_update_stress(vx, vz, sxx, szz, sxz, nx, nz, dx, dz, dt, lam, mu);
        for (size_t k = 0; k < n; k++) { sxx[k] *= damp[k]; szz[k] *= damp[k]; sxz[k] *= damp[k]; }
        A(sxx, src_ix, src_iz) += stf[it] * dt / 2.0;
        A(szz, src_ix, src_iz) += stf[it] * dt / 2.0;
        for (int r = 0; r < geo_num; r++) {
            size_t o = (size_t)r * (size_t)nt + (size_t)it;
            geoVx[o] = A(vx, geo_ix[r], geo_iz[r]);
            geoVz[o] = A(vz, geo_ix[r], geo_iz[r]);
            geoPr[o] = (A(sxx, geo_ix[r], geo_iz[r]) + A(szz, geo_ix[r], geo_iz[r])) * 0.5;
        }
        for (int r = 0; r < das_num; r++) {
            size_t o = (size_t)r * (size_t)nt + (size_t)it;
            int i = das_ix[r], j = das_iz[r];
            dasExx[o] = (A(vx, i, j) - A(vx, i - 1, j)) / dx;
            dasEzz[o] = (A(vz, i, j) - A(vz, i, j - 1)) / dz;
            dasExz[o] = 0.5 * ((A(vx, i, j + 1) - A(vx, i, j)) / dz + (A(vz, i + 1, j) - A(vz, i, j)) / dx);
            dasEtt[o] = das_sens[r * 6 + 0] * dasExx[o] + das_sens[r * 6 + 3] * dasEzz[o] + das_sens[r * 6 + 1] * dasExz[o];
        }
    }
    free(vx); free(damp);
}
