// CPU-only check of the injection plan behind the persistent backward loop's general receivers (csrc/inject_plan.cpp), built with
// -fsanitize=address,undefined by tests/test_sanitizers.py.  For random channel sets (scattered, repeated, strided, vertical,
// directional) the plan -- residual folded per target cell, then one add per owning lane found through lookup / lane mask / popcount --
// must leave in the adjoint velocities exactly what the channel-by-channel adds of res_injection_exx / _ezz (Src/utilities.cu:605-641)
// and of the directional transpose leave there, and every table must be consistent.
//   argv[1] = seed, argv[2] = number of random cases
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../sep-2023_amd/csrc/inject_plan.hpp"

using namespace sepfwi;

static int check(std::mt19937 &rng) {
    const int nzc = 8 + (int)(rng() % 60), nx = 10 + (int)(rng() % 300), nseg = (nx + 63) / 64;
    const int kind = (int)(rng() % 4);  // 0 horizontal scattered, 1 vertical, 2 directional, 3 horizontal strided line with repeats
    const int nrec = 1 + (int)(rng() % 90);
    const float dx_dz = 0.5f + (float)(rng() % 100) / 80.0f;
    std::vector<int> zr(nrec), xr(nrec);
    std::vector<float> sens;
    for (int r = 0; r < nrec; r++) {
        zr[r] = 1 + (int)(rng() % (nzc - 2));
        xr[r] = 1 + (int)(rng() % (nx - 2));
        if (kind == 3) {
            zr[r] = nzc / 2;
            xr[r] = 1 + (3 * r) % (nx - 2);
        }
    }
    if (kind == 2)
        for (int k = 0; k < 3 * nrec; k++) sens.push_back((float)((int)(rng() % 2001) - 1000) / 1000.0f);
    InjectPlan p;
    try {
        p = make_inject_plan(nrec, zr.data(), xr.data(), kind == 2 ? sens.data() : nullptr, kind == 1, dx_dz, nzc, nx);
    } catch (...) {
        return 1;
    }
    if ((int)p.lookup.size() != nzc * nseg || (int)p.tgt_start.size() != p.ntgt + 1 || p.ent_rec.size() != p.ent_w.size() ||
        p.tgt_start.back() != (int)p.ent_rec.size())
        return 2;
    std::vector<float> res(nrec);
    for (float &v : res) v = (float)((int)(rng() % 20001) - 10000) / 7.0f;
    // reference: channel by channel onto zero fields (k_inject's adds in channel order)
    std::vector<float> vx((size_t)nzc * nx, 0.0f), vz((size_t)nzc * nx, 0.0f), gx(vx), gz(vz);
    auto at = [&](std::vector<float> &f, int z, int x) -> float & { return f[(size_t)z * nx + x]; };
    for (int r = 0; r < nrec; r++) {
        const int z = zr[r], x = xr[r];
        const float v = res[r];
        if (kind == 2) {
            const float a = sens[3 * r], b = sens[3 * r + 1] * dx_dz, c = 0.5f * sens[3 * r + 2];
            at(vx, z, x) += a * v; at(vx, z, x - 1) += -a * v; at(vz, z, x) += b * v; at(vz, z - 1, x) += -b * v;
            at(vx, z + 1, x) += (c * dx_dz) * v; at(vx, z, x) += -(c * dx_dz) * v; at(vz, z, x + 1) += c * v; at(vz, z, x) += -c * v;
        } else if (kind == 1) {
            at(vz, z, x) += v; at(vz, z - 1, x) += -v;
        } else {
            at(vx, z, x) += v; at(vx, z, x - 1) += -v;
        }
    }
    // the plan: fold, then one add per owning lane
    std::vector<float> val(p.ntgt);
    for (int t = 0; t < p.ntgt; t++) {
        if (p.tgt_start[t] >= p.tgt_start[t + 1]) return 3;  // a target without entries
        float s = 0.0f;
        for (int e = p.tgt_start[t]; e < p.tgt_start[t + 1]; e++) {
            if (p.ent_rec[e] < 0 || p.ent_rec[e] >= nrec) return 4;
            if (e > p.tgt_start[t] && p.ent_rec[e] < p.ent_rec[e - 1]) return 5;  // channel order inside a target
            s += p.ent_w[e] * res[p.ent_rec[e]];
        }
        val[t] = s;
    }
    std::vector<char> used(p.ntgt, 0);
    for (int z = 0; z < nzc; z++)
        for (int xs = 0; xs < nseg; xs++) {
            const int slot = p.lookup[(size_t)z * nseg + xs];
            if (slot < 0) continue;
            if (slot >= (int)p.segs.size()) return 6;
            const InjSeg &q = p.segs[slot];
            if (q.mask[0] == 0ull && q.mask[1] == 0ull) return 7;
            for (int lane = 0; lane < 64; lane++)
                for (int f = 0; f < 2; f++)
                    if ((q.mask[f] >> lane) & 1ull) {
                        const int t = q.base[f] + __builtin_popcountll(q.mask[f] & ((1ull << lane) - 1ull));
                        const int x = xs * 64 + lane;
                        if (t < 0 || t >= p.ntgt || used[t] || x >= nx) return 8;
                        used[t] = 1;
                        at(f ? gz : gx, z, x) += val[t];
                    }
        }
    for (int t = 0; t < p.ntgt; t++)
        if (!used[t]) return 9;
    for (size_t k = 0; k < vx.size(); k++)
        if (vx[k] != gx[k] || vz[k] != gz[k]) return 10;
    return 0;
}

int main(int argc, char **argv) {
    const int seed = argc > 1 ? atoi(argv[1]) : 1, n = argc > 2 ? atoi(argv[2]) : 200;
    std::mt19937 rng(seed);
    for (int k = 0; k < n; k++) {
        const int rc = check(rng);
        if (rc) {
            printf("FAIL case %d code %d\n", k, rc);
            return 1;
        }
    }
    // a channel that reaches outside the grid is refused, not written past the tables
    const int z0[1] = {0}, x0[1] = {0};
    bool threw = false;
    try {
        (void)make_inject_plan(1, z0, x0, nullptr, false, 1.0f, 8, 8);
    } catch (...) {
        threw = true;
    }
    if (!threw) {
        printf("FAIL: out-of-grid channel accepted\n");
        return 1;
    }
    printf("OK %d cases\n", n);
    return 0;
}
