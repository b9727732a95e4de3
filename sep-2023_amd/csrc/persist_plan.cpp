// persist_plan.cpp -- see persist_plan.hpp.  Pure host code (no HIP): unit-tested on the CPU (tests/native/).
#include "persist_plan.hpp"

#include <algorithm>

namespace sepfwi {

std::string make_persist_plan(int nzc, int nseg, int nwg, int nband, int strip_w, PersistPlan *out, bool edge_first, const PlanCost &cost) {
    PersistPlan &p = *out;
    p = PersistPlan{};
    if (nzc < 1 || nseg < 1 || nwg < 1 || nband < 1 || strip_w < 1) return "persist plan: bad arguments";
    if (nzc > 65535 || nseg > 255) return "persist plan: grid too large for the 16 + 8 bit segment descriptor";
    if (nwg % nband) return "persist plan: workgroups not a multiple of the bands";
    p.nwg = nwg;
    p.nband = nband;
    p.per_band = nwg / nband;
    p.nzc = nzc;
    p.nseg = nseg;
    p.strip_w = strip_w;
    p.owner.assign((size_t)nzc * nseg, -1);
    std::vector<std::vector<uint32_t>> tiles(nwg);
    // cost of a segment in percent of a plain one (100 everywhere without weights: the runs are then equal in length)
    auto seg_cost = [&](int z, int xs) -> long long {
        long long w = 100;
        if (cost.nx > 0) {
            const int x0 = xs * 64, x1 = std::min(cost.nx, x0 + 64) - 1;
            const int in_layers = std::max(0, std::min(x1, cost.npml - 1) - x0 + 1) + std::max(0, x1 - std::max(x0, cost.nx - cost.npml) + 1);
            if (in_layers > 0) w = w * (in_layers == x1 - x0 + 1 ? cost.w_xpure : cost.w_xpml) / 100;
            const int zl = cost.period > 0 ? z % cost.period : z, nzl = cost.period > 0 ? cost.period : nzc;
            if (zl < cost.npml || zl > nzl - cost.npml - 1) w = w * cost.w_zpml / 100;
        }
        return std::max(1LL, w);
    };
    // rows per band: equal, or in inverse proportion to the band weights
    std::vector<int> row0(nband + 1, 0);
    {
        double tot = 0.0;
        for (int b = 0; b < nband; b++) tot += 100.0 / (double)((cost.band_w && cost.band_w[b] > 0) ? cost.band_w[b] : 100);
        double acc = 0.0;
        for (int b = 0; b < nband; b++) {
            row0[b] = (int)(acc / tot * nzc + 0.5);
            acc += 100.0 / (double)((cost.band_w && cost.band_w[b] > 0) ? cost.band_w[b] : 100);
        }
        row0[nband] = nzc;
        for (int b = 0; b < nband; b++)
            if (row0[b + 1] <= row0[b]) return "persist plan: a band without rows";
    }
    for (int b = 0; b < nband; b++) {
        const int r0 = row0[b], r1 = row0[b + 1];
        long long total = 0;
        for (int z = r0; z < r1; z++)
            for (int xs = 0; xs < nseg; xs++) total += seg_cost(z, xs);
        long long k = 0;  // cost of the band's strip-by-strip sequence before this segment
        int strip = 0;
        for (int s0 = 0; s0 < nseg; s0 += strip_w, strip++) {
            const int s1 = std::min(nseg, s0 + strip_w);
            for (int rr = 0; rr < r1 - r0; rr++) {
                const int z = ((strip & 1) && cost.snake) ? r1 - 1 - rr : r0 + rr;  // alternate direction: a run that crosses strips stays compact
                for (int xs = s0; xs < s1; xs++) {
                    const long long w = seg_cost(z, xs);
                    const int t = b * p.per_band + (int)std::min<long long>(p.per_band - 1, (k + w / 2) * p.per_band / total);
                    k += w;
                    p.owner[(size_t)z * nseg + xs] = t;
                    tiles[t].push_back((uint32_t)z | ((uint32_t)xs << 16));
                }
            }
        }
    }
    for (int t = 0; t < nwg; t++)
        if (tiles[t].empty()) return "persist plan: a tile without segments";
    p.hdr.assign(nwg, TileHdr{});
    size_t cap = 1;
    for (int t = 0; t < nwg; t++) cap = std::max(cap, tiles[t].size());
    p.cap = (int)cap;
    p.seg.assign((size_t)nwg * cap, 0u);
    for (int t = 0; t < nwg; t++) {
        TileHdr &h = p.hdr[t];
        std::vector<uint32_t> edge, inner;
        std::vector<int> nbs;
        const int band = t / p.per_band;
        for (uint32_t d : tiles[t]) {
            const int z = (int)(d & 0xffffu), xs = (int)(d >> 16);
            bool is_edge = false, xband = false;
            auto look = [&](int zz, int xx) {
                if (zz < 0 || zz >= nzc || xx < 0 || xx >= nseg) return;
                const int o = p.owner[(size_t)zz * nseg + xx];
                if (o == t) return;
                is_edge = true;
                if (o / p.per_band != band) xband = true;
                if (std::find(nbs.begin(), nbs.end(), o) == nbs.end()) nbs.push_back(o);
            };
            look(z - 2, xs); look(z - 1, xs); look(z + 1, xs); look(z + 2, xs); look(z, xs - 1); look(z, xs + 1);
            const uint32_t flagged = d | (is_edge ? kSegEdge : 0u) | (xband ? kSegXband : 0u);
            ((is_edge || !edge_first) ? edge : inner).push_back(flagged);
        }
        if ((int)nbs.size() > kPlanMaxNb) return "persist plan: a tile has more than " + std::to_string(kPlanMaxNb) + " neighbours";
        h.n_edge = (int)edge.size();  // natural order: every segment counts as an edge segment -- the tile publishes a phase when all of it is done
        h.n_seg = (int)(edge.size() + inner.size());
        h.n_nb = (int)nbs.size();
        std::sort(nbs.begin(), nbs.end());
        for (int k = 0; k < kPlanMaxNb; k++) h.nb[k] = k < h.n_nb ? nbs[k] : -1;
        uint32_t *dst = p.seg.data() + (size_t)t * cap;
        std::copy(edge.begin(), edge.end(), dst);
        std::copy(inner.begin(), inner.end(), dst + edge.size());
    }
    return "";
}

std::vector<unsigned long long> make_quiet_neighbours(const PersistPlan &p) {
    std::vector<unsigned long long> out;
    if (p.cap > 253) return out;
    std::vector<int> pos((size_t)p.nzc * p.nseg, -1);  // (row, column) -> position in its tile
    for (int t = 0; t < p.nwg; t++)
        for (int j = 0; j < p.hdr[t].n_seg; j++) {
            const uint32_t d = p.seg[(size_t)t * p.cap + j];
            pos[(size_t)(d & 0xffffu) * p.nseg + ((d >> 16) & 0xffu)] = j;
        }
    out.assign((size_t)p.nwg * p.cap, ~0ull);
    const int dz[6] = {-2, -1, 1, 2, 0, 0}, dx[6] = {0, 0, 0, 0, -1, 1};
    for (int t = 0; t < p.nwg; t++)
        for (int j = 0; j < p.hdr[t].n_seg; j++) {
            const uint32_t d = p.seg[(size_t)t * p.cap + j];
            const int z = (int)(d & 0xffffu), xs = (int)((d >> 16) & 0xffu);
            unsigned long long w = ~0ull;  // bytes 6, 7 stay 0xff
            for (int q = 0; q < 6; q++) {
                const int zz = z + dz[q], xx = xs + dx[q];
                unsigned long long code = 0xffull;
                if (zz >= 0 && zz < p.nzc && xx >= 0 && xx < p.nseg)
                    code = p.owner[(size_t)zz * p.nseg + xx] == t ? (unsigned long long)pos[(size_t)zz * p.nseg + xx] : 0xfeull;
                w = (w & ~(0xffull << (8 * q))) | (code << (8 * q));
            }
            out[(size_t)t * p.cap + j] = w;
        }
    return out;
}

std::string make_persist_plan_multishot(int nzc, int nshot, int nseg, int nwg, int nband, int strip_w, PersistPlan *out, bool edge_first,
                                        const PlanCost &cost) {
    if (nshot < 1 || nshot > kPlanMaxShots) return "persist plan: more than " + std::to_string(kPlanMaxShots) + " shots in one launch";
    if ((long long)nzc * nshot > 65535) return "persist plan: the stacked shots have more rows than the descriptor holds";
    PlanCost c = cost;
    c.period = nzc;
    const std::string why = make_persist_plan(nzc * nshot, nseg, nwg, nband, strip_w, out, edge_first, c);
    if (!why.empty()) return why;
    for (uint32_t &d : out->seg) {
        const uint32_t zv = d & 0xffffu;
        d = (d & ~0xffffu) | (zv % (uint32_t)nzc) | ((zv / (uint32_t)nzc) << 26);
    }
    return "";
}

}  // namespace sepfwi
