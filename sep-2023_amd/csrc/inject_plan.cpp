// inject_plan.cpp -- see inject_plan.hpp.  Pure host code (no HIP): unit-tested on the CPU (tests/native/).
#include "inject_plan.hpp"

#include <algorithm>
#include <stdexcept>

namespace sepfwi {

namespace {
struct Add {
    int z, x, field, rec;  // field 0: vx, 1: vz
    float w;
};
}  // namespace

InjectPlan make_inject_plan(int nrec, const int *z_rec, const int *x_rec, const float *sens, bool vertical, float dx_dz, int nzc, int nx) {
    InjectPlan p;
    const int nseg = (nx + 63) / 64;
    p.lookup.assign((size_t)nzc * nseg, -1);
    std::vector<Add> adds;
    adds.reserve((size_t)nrec * (sens ? 8 : 2));
    for (int r = 0; r < nrec; r++) {
        const int z = z_rec[r], x = x_rec[r];
        if (sens) {  // the transpose of ett = s_xx exx + s_zz ezz + s_xz exz (k_record; MOD/elasticSolver.py:266-276)
            const float a = sens[3 * r], b = sens[3 * r + 1] * dx_dz, c = 0.5f * sens[3 * r + 2];
            adds.push_back({z, x, 0, r, a});
            adds.push_back({z, x - 1, 0, r, -a});
            adds.push_back({z, x, 1, r, b});
            adds.push_back({z - 1, x, 1, r, -b});
            adds.push_back({z + 1, x, 0, r, c * dx_dz});
            adds.push_back({z, x, 0, r, -(c * dx_dz)});
            adds.push_back({z, x + 1, 1, r, c});
            adds.push_back({z, x, 1, r, -c});
        } else if (vertical) {  // res_injection_ezz, utilities.cu:632-641
            adds.push_back({z, x, 1, r, 1.0f});
            adds.push_back({z - 1, x, 1, r, -1.0f});
        } else {  // res_injection_exx, utilities.cu:605-615
            adds.push_back({z, x, 0, r, 1.0f});
            adds.push_back({z, x - 1, 0, r, -1.0f});
        }
    }
    for (const Add &a : adds)
        if (a.z < 0 || a.z >= nzc || a.x < 0 || a.x >= nx) throw std::invalid_argument("inject plan: a channel reaches outside the grid");
    // sort by row segment, field, cell; entries of one target stay in channel order (stable)
    auto key = [&](const Add &a) { return (((long long)a.z * nseg + (a.x >> 6)) * 2 + a.field) * 64 + (a.x & 63); };
    std::stable_sort(adds.begin(), adds.end(), [&](const Add &u, const Add &v) { return key(u) < key(v); });
    long long prev = -1;
    for (const Add &a : adds) {
        const long long k = key(a);
        if (k != prev) {  // a new target
            const int sidx = a.z * nseg + (a.x >> 6);
            if (p.lookup[sidx] < 0) {
                p.lookup[sidx] = (int)p.segs.size();
                p.segs.push_back(InjSeg{{0, 0}, {0, 0}, {0ull, 0ull}});
            }
            InjSeg &s = p.segs[p.lookup[sidx]];
            if (s.mask[a.field] == 0ull) s.base[a.field] = p.ntgt;
            s.mask[a.field] |= 1ull << (a.x & 63);
            p.tgt_start.push_back((int)p.ent_rec.size());
            p.ntgt++;
            prev = k;
        }
        p.ent_rec.push_back(a.rec);
        p.ent_w.push_back(a.w);
    }
    p.tgt_start.push_back((int)p.ent_rec.size());
    return p;
}

}  // namespace sepfwi
