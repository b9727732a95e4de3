// CPU-only check of the tiling behind the persistent backward loop (csrc/persist_plan.cpp), built with -fsanitize=address,undefined
// by tests/test_sanitizers.py: every row segment owned exactly once, tiles balanced to +- 1, edge segments first and flagged
// exactly when a stencil leaves the tile, XBAND exactly when it leaves the band, neighbour lists complete and symmetric.
//   argv[1] = seed, argv[2] = number of random grids
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

#include "../../sep-2023_amd/csrc/persist_plan.hpp"

using namespace sepfwi;

static int check(int nzc, int nseg, int nwg, int nband, int sw) {
    PersistPlan p;
    const std::string why = make_persist_plan(nzc, nseg, nwg, nband, sw, &p);
    if (!why.empty()) return why.find("neighbours") != std::string::npos ? 2 : -1;  // refused (legitimately: too many neighbours) or bad arguments
    std::vector<int> seen((size_t)nzc * nseg, 0);
    int mn = 1 << 30, mx = 0;
    std::vector<std::set<int>> nbs(nwg);
    for (int t = 0; t < nwg; t++) {
        const TileHdr &h = p.hdr[t];
        mn = std::min(mn, h.n_seg);
        mx = std::max(mx, h.n_seg);
        if (h.n_seg > p.cap || h.n_edge > h.n_seg || h.n_nb > kPlanMaxNb) return 10;
        for (int k = 0; k < h.n_nb; k++) nbs[t].insert(h.nb[k]);
        for (int j = 0; j < h.n_seg; j++) {
            const uint32_t d = p.seg[(size_t)t * p.cap + j];
            const int z = (int)(d & 0xffffu), xs = (int)((d >> 16) & 0xffu);
            if (z >= nzc || xs >= nseg || p.owner[(size_t)z * nseg + xs] != t) return 11;
            seen[(size_t)z * nseg + xs]++;
            bool edge = false, xband = false;
            const int dz[6] = {-2, -1, 1, 2, 0, 0}, dx[6] = {0, 0, 0, 0, -1, 1};
            for (int q = 0; q < 6; q++) {
                const int zz = z + dz[q], xx = xs + dx[q];
                if (zz < 0 || zz >= nzc || xx < 0 || xx >= nseg) continue;
                const int o = p.owner[(size_t)zz * nseg + xx];
                if (o == t) continue;
                edge = true;
                if (o / p.per_band != t / p.per_band) xband = true;
                if (!nbs[t].count(o)) return 12;  // a tile it exchanges halos with is missing from its list
            }
            if (edge != ((d & kSegEdge) != 0) || xband != ((d & kSegXband) != 0)) return 13;
            if (edge != (j < h.n_edge)) return 14;  // edge segments first, nothing else among them
        }
    }
    for (int v : seen)
        if (v != 1) return 15;
    for (int t = 0; t < nwg; t++)
        for (int o : nbs[t])
            if (o < 0 || o >= nwg || !nbs[o].count(t)) return 16;  // symmetric
    // balanced inside every band (bands differ by at most one row of segments)
    for (int b = 0; b < nband; b++) {
        int lo = 1 << 30, hi = 0;
        for (int t = b * p.per_band; t < (b + 1) * p.per_band; t++) {
            lo = std::min(lo, p.hdr[t].n_seg);
            hi = std::max(hi, p.hdr[t].n_seg);
        }
        if (hi - lo > 1) return 17;
    }
    return 0;
}

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const int n = argc > 2 ? atoi(argv[2]) : 200;
    std::mt19937 rng(seed);
    int rc = check(1064, 33, 512, 8, 3);  // the headline grid as shipped
    if (rc != 0) { printf("FAIL headline plan: %d\n", rc); return 1; }
    rc = check(1064, 33, 256, 8, 4);
    if (rc != 0) { printf("FAIL headline plan (1 per CU): %d\n", rc); return 1; }
    int ok = 0, refused = 0;
    for (int it = 0; it < n; it++) {
        const int nband = 1 << (rng() % 4), per = 1 + (int)(rng() % 70), nwg = nband * per;
        const int nzc = 1 + (int)(rng() % 700), nseg = 1 + (int)(rng() % 60), sw = 1 + (int)(rng() % 9);
        rc = check(nzc, nseg, nwg, nband, sw);
        if (rc == 0) ok++;
        else if (rc == 2) refused++;
        else { printf("FAIL nzc %d nseg %d nwg %d nband %d strip %d: %d\n", nzc, nseg, nwg, nband, sw, rc); return 1; }
    }
    if (check(70000, 3, 8, 8, 1) != -1 || check(10, 300, 8, 8, 1) != -1 || check(10, 10, 9, 8, 1) != -1) { printf("FAIL: bad arguments accepted\n"); return 1; }
    printf("OK %d plans checked, %d refused for more than %d neighbours\n", ok, refused, kPlanMaxNb);
    return 0;
}
