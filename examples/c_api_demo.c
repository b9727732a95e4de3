/* c_api_demo.c -- the C ABI of libsepfwi.so used from plain C (no Python, no torch): the calls a maintainer of the reference
 * would place where Src/Torch_Fwi.cpp:31,86,132 call cufd().  Host pointers in, host pointers out.
 *
 *   gcc -O2 -I include examples/c_api_demo.c -L sep-2023_amd -lsepfwi -Wl,-rpath,$PWD/sep-2023_amd -lm -o /tmp/c_api_demo
 *   /tmp/c_api_demo /tmp/c_api_demo_work
 *
 * Homogeneous 60x80 model + one stiffer box, 2 shots, a DAS line of 60 channels, 300 time steps: writes the two JSON files
 * of the reference's schema (fwi_utils.py:46-124), generates "observed" data from the true model (calc_id 2), evaluates
 * misfit and gradients of the background model (calc_id 1) and checks the misfit against the calc_id 0 entry point. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "sepfwi.h"

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc_ = (call);                                                    \
        if (rc_ != SEPFWI_OK) {                                              \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sepfwi_last_error()); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

int main(int argc, char **argv) {
    const char *work = argc > 1 ? argv[1] : "/tmp/c_api_demo_work";
    const int nz = 60, nx = 80, nPml = 10, nSteps = 300, nShots = 2, nrec = 60;
    const int nPad = 32 - (nz + 2 * nPml) % 32; /* fwi_utils / Main-001:35 */
    const int nzp = nz + 2 * nPml + nPad, nxp = nx + 2 * nPml;
    const float dz = 10.f, dt = 1e-3f, f0 = 25.f;
    char para[512], survey[512], data[512], cmd[600];
    snprintf(para, sizeof para, "%s/para_file.json", work);
    snprintf(survey, sizeof survey, "%s/survey_file.json", work);
    snprintf(data, sizeof data, "%s/Data", work);
    snprintf(cmd, sizeof cmd, "mkdir -p %s", data);
    if (system(cmd) != 0) return 1;

    FILE *fp = fopen(para, "w");
    if (!fp) return 1;
    fprintf(fp, "{\"nz\": %d, \"nx\": %d, \"dz\": %g, \"dx\": %g, \"nSteps\": %d, \"dt\": %g, \"f0\": %g, \"nPoints_pml\": %d, \"nPad\": %d, "
                "\"survey_fname\": \"%s\", \"data_dir_name\": \"%s\"}\n",
            nzp, nxp, dz, dz, nSteps, dt, f0, nPml, nPad, survey, data);
    fclose(fp);
    fp = fopen(survey, "w");
    if (!fp) return 1;
    fprintf(fp, "{\"nShots\": %d", nShots);
    for (int s = 0; s < nShots; s++) {
        fprintf(fp, ", \"shot%d\": {\"z_src\": 2, \"x_src\": %d, \"nrec\": %d, \"z_rec\": [", s, 20 + 40 * s, nrec);
        for (int r = 0; r < nrec; r++) fprintf(fp, "%s%d", r ? ", " : "", nz - 8);
        fprintf(fp, "], \"x_rec\": [");
        for (int r = 0; r < nrec; r++) fprintf(fp, "%s%d", r ? ", " : "", 10 + r);
        fprintf(fp, "]}");
    }
    fprintf(fp, "}\n");
    fclose(fp);

    const size_t n = (size_t)nzp * nxp;
    float *lam = malloc(n * 4), *mu = malloc(n * 4), *den = malloc(n * 4), *lam_t = malloc(n * 4);
    float *gL = malloc(n * 4), *gM = malloc(n * 4), *gD = malloc(n * 4);
    float *stf = calloc((size_t)nShots * nSteps, 4), *gS = calloc((size_t)nShots * nSteps, 4);
    for (size_t i = 0; i < n; i++) { /* Vp 3000, Vs 1732, rho 2400 -> Lame parameters in MPa (FWI_ops.py:124-125) */
        const double vp = 3000.0, vs = 3000.0 / 1.732, rho = 2400.0;
        mu[i] = (float)(vs * vs * rho / 1e6);
        lam[i] = (float)((vp * vp - 2 * vs * vs) * rho / 1e6);
        den[i] = (float)rho;
        lam_t[i] = lam[i];
    }
    for (int z = nPml + 25; z < nPml + 35; z++)
        for (int x = nPml + 35; x < nPml + 45; x++) lam_t[(size_t)z * nxp + x] *= 1.10f; /* the anomaly to be imaged */
    for (int s = 0; s < nShots; s++)
        for (int it = 0; it < nSteps; it++) { /* Ricker x 1e7, fwi_utils.py:127-140 */
            const double e = M_PI * M_PI * f0 * f0, t = it * dt - 1.2 / f0;
            stf[(size_t)s * nSteps + it] = (float)((1.0 - 2.0 * e * t * t) * exp(-e * t * t) * 1e7);
        }
    const int ids[2] = {0, 1};
    float misfit = -1.f, misfit0 = -1.f;

    printf("libsepfwi version %d, %d HIP device(s)\n", sepfwi_version(), sepfwi_device_count());
    CHECK(sepfwi_cufd(&misfit, NULL, NULL, NULL, NULL, lam_t, mu, den, stf, SEPFWI_CALC_OBSERVE, 0, nShots, ids, para));
    CHECK(sepfwi_cufd(&misfit, gL, gM, gD, gS, lam, mu, den, stf, SEPFWI_CALC_GRADIENT, 0, nShots, ids, para));
    CHECK(sepfwi_cufd(&misfit0, NULL, NULL, NULL, NULL, lam, mu, den, stf, SEPFWI_CALC_MISFIT, 0, nShots, ids, para));
    double gmax = 0;
    size_t imax = 0;
    for (size_t i = 0; i < n; i++)
        if (fabs(gL[i]) > gmax) { gmax = fabs(gL[i]); imax = i; }
    printf("misfit %.6e (calc_id 1)  %.6e (calc_id 0)   max |dJ/dLambda| %.4e at (z=%zu, x=%zu)\n", misfit, misfit0, gmax, imax / nxp - nPml,
           imax % nxp - nPml);
    sepfwi_stats st;
    CHECK(sepfwi_get_stats(para, 0, &st));
    printf("last call: %.2f ms, %lld launches, %.1f MB on the device\n", st.total_ms, st.launches, st.device_bytes / 1e6);
    /* an unknown file is an error code and a message, not exit(1) as in the reference (Src/utilities.cu:12-16) */
    if (sepfwi_cufd(&misfit, NULL, NULL, NULL, NULL, lam, mu, den, stf, SEPFWI_CALC_MISFIT, 0, nShots, ids, "/nonexistent.json") == SEPFWI_OK) return 1;
    printf("expected failure reported: %s\n", sepfwi_last_error());
    sepfwi_release_all();
    const int ok = misfit > 0 && fabs(misfit - misfit0) <= 1e-6 * misfit && gmax > 0;
    printf(ok ? "OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}
