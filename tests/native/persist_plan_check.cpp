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

static int check_quiet(const PersistPlan &p);

// cost = nullptr: plain tiling (tiles balanced by count); else by cost (PlanCost: the C-PML strips weigh more)
static int check(int nzc, int nseg, int nwg, int nband, int sw, const PlanCost *cost = nullptr) {
    PersistPlan p;
    const std::string why = cost ? make_persist_plan(nzc, nseg, nwg, nband, sw, &p, true, *cost) : make_persist_plan(nzc, nseg, nwg, nband, sw, &p);
    if (!why.empty())  // refused (legitimately: too many neighbours, more tiles than segments) or bad arguments
        return (why.find("neighbours") != std::string::npos || why.find("without") != std::string::npos) ? 2 : -1;
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
    // balanced inside every band (bands differ by at most one row of segments): by count, or by cost to within two segments
    auto seg_cost = [&](int z, int xs) -> long long {
        long long w = 100;
        if (cost && cost->nx > 0) {
            const int x0 = xs * 64, x1 = std::min(cost->nx, x0 + 64) - 1;
            const int in_layers = std::max(0, std::min(x1, cost->npml - 1) - x0 + 1) + std::max(0, x1 - std::max(x0, cost->nx - cost->npml) + 1);
            if (in_layers > 0) w = w * (in_layers == x1 - x0 + 1 ? cost->w_xpure : cost->w_xpml) / 100;
            if (z < cost->npml || z > nzc - cost->npml - 1) w = w * cost->w_zpml / 100;
        }
        return std::max(1LL, w);
    };
    for (int b = 0; b < nband; b++) {
        long long lo = 1LL << 60, hi = 0, wmax = 0;
        for (int t = b * p.per_band; t < (b + 1) * p.per_band; t++) {
            long long c = 0;
            for (int j = 0; j < p.hdr[t].n_seg; j++) {
                const uint32_t d = p.seg[(size_t)t * p.cap + j];
                const long long w = cost ? seg_cost((int)(d & 0xffffu), (int)((d >> 16) & 0xffu)) : 1;
                c += w;
                wmax = std::max(wmax, w);
            }
            lo = std::min(lo, c);
            hi = std::max(hi, c);
        }
        if (hi - lo > (cost ? 2 * wmax : 1)) return 17;
    }
    return check_quiet(p);
}

// Quiet-segment neighbour table (make_quiet_neighbours): byte k of a segment's word = position in the tile of its k-th stencil
// neighbour, 0xfe when another tile owns it, 0xff outside the grid.
static int check_quiet(const PersistPlan &p) {
    const std::vector<unsigned long long> nb = make_quiet_neighbours(p);
    if (p.cap > 253) return nb.empty() ? 0 : 30;
    if (nb.size() != (size_t)p.nwg * p.cap) return 31;
    const int dz[6] = {-2, -1, 1, 2, 0, 0}, dx[6] = {0, 0, 0, 0, -1, 1};
    for (int t = 0; t < p.nwg; t++)
        for (int j = 0; j < p.hdr[t].n_seg; j++) {
            const uint32_t d = p.seg[(size_t)t * p.cap + j];
            const int z = (int)(d & 0xffffu), xs = (int)((d >> 16) & 0xffu);
            const unsigned long long w = nb[(size_t)t * p.cap + j];
            if ((w >> 48) != 0xffffull) return 32;
            for (int q = 0; q < 6; q++) {
                const int zz = z + dz[q], xx = xs + dx[q];
                const unsigned int code = (unsigned int)((w >> (8 * q)) & 0xffull);
                if (zz < 0 || zz >= p.nzc || xx < 0 || xx >= p.nseg) {
                    if (code != 0xffu) return 33;
                } else if (p.owner[(size_t)zz * p.nseg + xx] != t) {
                    if (code != 0xfeu) return 34;
                } else {
                    if ((int)code >= p.hdr[t].n_seg) return 35;
                    const uint32_t e = p.seg[(size_t)t * p.cap + code];
                    if ((int)(e & 0xffffu) != zz || (int)((e >> 16) & 0xffu) != xx) return 36;
                }
            }
        }
    return 0;
}

// Several shots in one launch: every (shot, row, segment column) owned exactly once, descriptors within range, tile sizes as the
// plan of the stacked grid gave them (make_persist_plan_multishot only rewrites the descriptors).
static int check_multishot(int nzc, int nshot, int nseg, int nwg, int nband, int sw) {
    PersistPlan p, q;
    PlanCost pc;
    pc.nx = nseg * 64 - 7;
    pc.npml = std::min(10, nzc / 3);
    pc.w_xpml = pc.w_xpure = 150;
    pc.w_zpml = 115;
    const std::string why = make_persist_plan_multishot(nzc, nshot, nseg, nwg, nband, sw, &p, true, pc);
    if (!why.empty()) return (why.find("neighbours") != std::string::npos || why.find("without") != std::string::npos) ? 2 : -1;
    pc.period = nzc;
    if (!make_persist_plan(nzc * nshot, nseg, nwg, nband, sw, &q, true, pc).empty()) return 20;
    std::vector<int> seen((size_t)nzc * nshot * nseg, 0);
    for (int t = 0; t < nwg; t++) {
        if (p.hdr[t].n_seg != q.hdr[t].n_seg || p.hdr[t].n_edge != q.hdr[t].n_edge || p.hdr[t].n_nb != q.hdr[t].n_nb) return 21;
        for (int j = 0; j < p.hdr[t].n_seg; j++) {
            const uint32_t d = p.seg[(size_t)t * p.cap + j], dv = q.seg[(size_t)t * q.cap + j];
            const int z = (int)(d & 0xffffu), xs = (int)((d >> 16) & 0xffu), sh = (int)(d >> 26);
            if (z >= nzc || xs >= nseg || sh >= nshot) return 22;
            if ((int)(dv & 0xffffu) != sh * nzc + z || ((d ^ dv) & 0x03ff0000u) != 0u) return 23;  // same virtual row, same column and flags
            seen[((size_t)sh * nzc + z) * nseg + xs]++;
        }
    }
    for (int v : seen)
        if (v != 1) return 24;
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
        PlanCost pc;
        int bw[8];
        const bool weighted = (rng() % 2) != 0;
        if (weighted) {
            pc.nx = std::max(1, nseg * 64 - (int)(rng() % 64));
            pc.npml = (int)(rng() % 70);
            pc.w_xpml = 25 + (int)(rng() % 300);
            pc.w_zpml = 25 + (int)(rng() % 300);
            pc.w_xpure = 25 + (int)(rng() % 300);
            for (int b = 0; b < 8; b++) bw[b] = 60 + (int)(rng() % 100);
            pc.band_w = (rng() % 2) ? bw : nullptr;
        }
        rc = check(nzc, nseg, nwg, nband, sw, weighted ? &pc : nullptr);
        if (rc == 0) ok++;
        else if (rc == 2) refused++;
        else { printf("FAIL nzc %d nseg %d nwg %d nband %d strip %d: %d\n", nzc, nseg, nwg, nband, sw, rc); return 1; }
    }
    // the reference's own experiments: 19 shots of 165 rows x 5 segment columns in one launch; and random stacks
    rc = check_multishot(165, 19, 5, 512, 8, 3);
    if (rc != 0) { printf("FAIL notebook-sized multi-shot plan: %d\n", rc); return 1; }
    for (int it = 0; it < n / 4; it++) {
        const int nzc = 8 + (int)(rng() % 300), nshot = 1 + (int)(rng() % 40), nseg = 1 + (int)(rng() % 12), per = 1 + (int)(rng() % 64), sw = 1 + (int)(rng() % 5);
        rc = check_multishot(nzc, nshot, nseg, 8 * per, 8, sw);
        if (rc != 0 && rc != 2) { printf("FAIL multi-shot nzc %d nshot %d nseg %d nwg %d strip %d: %d\n", nzc, nshot, nseg, 8 * per, sw, rc); return 1; }
    }
    if (check_multishot(100, 65, 4, 64, 8, 2) != -1 || check_multishot(3000, 30, 4, 64, 8, 2) != -1) { printf("FAIL: too many shots / rows accepted\n"); return 1; }
    if (check(70000, 3, 8, 8, 1) != -1 || check(10, 300, 8, 8, 1) != -1 || check(10, 10, 9, 8, 1) != -1) { printf("FAIL: bad arguments accepted\n"); return 1; }
    printf("OK %d plans checked, %d refused for more than %d neighbours\n", ok, refused, kPlanMaxNb);
    return 0;
}
