// inject_plan.hpp -- host-side plan of the adjoint-source injection of ONE shot for the persistent backward loop.
//
// res_injection_exx / _ezz (Src/utilities.cu:605-641) and their directional generalisation add, per channel r and time step, the
// residual r[it][r] with weights +-w to a handful of adjoint-velocity cells; neighbouring channels share cells.  The per-step path
// does that with a launch of its own between the two halves of a backward step (k_inject, float atomics).  Inside the persistent
// loop every cell has ONE owner, so the adds are done by the lane that owns the cell, right after its adjoint-velocity update:
//   * TARGETS: the distinct (field, cell) pairs that receive anything, sorted by row segment, field, lane;
//   * per target the list of (channel, weight) ENTRIES, in channel order -- a small kernel folds the shot's residual [it][rec]
//     into one value per target and time step before the pass starts (k_inject_values, kernels_aux.hpp), summed in that order,
//     so the result does not depend on any scheduling (the atomics' order does);
//   * per row segment that holds targets one InjSeg: for vx and for vz the lanes that receive a value (64-bit mask) and the index
//     of the first of them in the target list -- lane l reads target base + popcount(mask below l);
//   * a lookup from the global row-segment index z * nseg + xs to the InjSeg (-1: nothing to inject there).
// A horizontal line of consecutive channels does not need any of this (it is folded into the adjoint-velocity body: LineRec).
#pragma once
#include <cstdint>
#include <vector>

#include "fwi_types.hpp"

namespace sepfwi {

struct InjectPlan {
    int ntgt = 0;                  // distinct (field, cell) targets
    std::vector<int> lookup;       // [nzc * nseg] -> index into segs, or -1
    std::vector<InjSeg> segs;
    std::vector<int> tgt_start;    // [ntgt + 1] -> entries
    std::vector<int> ent_rec;      // channel of the entry
    std::vector<float> ent_w;      // weight of the entry
};

// z_rec / x_rec: the shot's channels in padded grid coordinates; sens: null or nrec x 3 (s_xx, s_zz, s_xz); vertical: the fibre runs
// along z (ett = ezz); dx_dz = dx / dz (the z-differences of a directional channel carry it, k_record / k_inject).
InjectPlan make_inject_plan(int nrec, const int *z_rec, const int *x_rec, const float *sens, bool vertical, float dx_dz, int nzc, int nx);

}  // namespace sepfwi
