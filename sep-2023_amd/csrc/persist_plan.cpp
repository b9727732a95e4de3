// persist_plan.cpp -- see persist_plan.hpp.  Pure host code (no HIP): unit-tested on the CPU (tests/native/).
#include "persist_plan.hpp"

#include <algorithm>

namespace sepfwi {

std::string make_persist_plan(int nzc, int nseg, int nwg, int nband, int strip_w, PersistPlan *out, bool edge_first) {
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
    for (int b = 0; b < nband; b++) {
        const int r0 = (int)((long long)nzc * b / nband), r1 = (int)((long long)nzc * (b + 1) / nband);
        const long long len = (long long)(r1 - r0) * nseg;
        long long k = 0;  // position in the band's strip-by-strip sequence
        int strip = 0;
        for (int s0 = 0; s0 < nseg; s0 += strip_w, strip++) {
            const int s1 = std::min(nseg, s0 + strip_w);
            for (int rr = 0; rr < r1 - r0; rr++) {
                const int z = (strip & 1) ? r1 - 1 - rr : r0 + rr;  // alternate direction: a run that crosses strips stays compact
                for (int xs = s0; xs < s1; xs++, k++) {
                    const int t = b * p.per_band + (int)(k * p.per_band / len);
                    p.owner[(size_t)z * nseg + xs] = t;
                    tiles[t].push_back((uint32_t)z | ((uint32_t)xs << 16));
                }
            }
        }
    }
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

}  // namespace sepfwi
