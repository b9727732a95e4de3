// persist_plan.hpp -- host-side tiling of the grid for the persistent backward time loop (kernels.hip, k_bwd_persist).
//
// The unit of work is a ROW SEGMENT: 64 consecutive columns of one row (one wave, one 256-B line per array).  The grid's
// segments are dealt to `nwg` workgroups that stay resident for a whole backward pass; a workgroup owns the same segments in
// every time step (its imaging accumulators live in LDS).  Layout: `nband` bands of rows (one per XCD, so that tiles that
// exchange halos share an L2 except across the nband - 1 band edges); inside a band the segments are ordered strip by strip
// (strips `strip_w` segments wide, walked top-down / bottom-up alternately) and that sequence is cut into equal runs, one per
// workgroup -- every tile has the same number of segments +- 1 whatever the grid size.
//
// A segment is an EDGE segment when a stencil centred in it reaches a segment of another tile (rows z +- 1, z +- 2 of the same
// segment column, or the two neighbouring segment columns of the same row); edge segments are processed first in every phase
// and their completion is what a tile publishes to its neighbours.  XBAND marks edge segments whose partner lies in another
// band (another XCD's L2).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace sepfwi {

constexpr int kPlanMaxNb = 28;
constexpr uint32_t kSegEdge = 1u << 24, kSegXband = 1u << 25;

struct TileHdr {  // device-readable
    int n_edge, n_seg, n_nb;
    int nb[kPlanMaxNb];  // tiles this tile exchanges halos with
    int pad[1];          // 128 bytes: one header per cache line
};
static_assert(sizeof(TileHdr) == 128, "TileHdr is one 128-byte line");

struct PersistPlan {
    int nwg = 0, nband = 0, per_band = 0, cap = 0;  // cap: slots per tile in `seg` (>= the largest tile)
    int nzc = 0, nseg = 0, strip_w = 0;
    std::vector<uint32_t> seg;  // [nwg][cap]: z | segment column << 16 | flags; edge segments first
    std::vector<TileHdr> hdr;   // [nwg]
    std::vector<int> owner;     // [nzc][nseg] -> tile
};

// Builds the plan; returns an empty string, or why the grid cannot be tiled this way (e.g. a tile with more than kPlanMaxNb
// neighbours).  Rows [0, nzc) x segment columns [0, nseg).  edge_first = false keeps the strip order inside a tile (better cache locality) and
// lets n_edge = n_seg: such a tile publishes a phase only when it is through with all of it.
// Cost weights (optional): a row segment that straddles the edge of an x C-PML layer (columns x < npml or x > nx - npml - 1: its
// waves run the reverse-time bodies AND the absorbing-layer branches of the adjoint ones) counts w_xpml percent of a plain one, a
// segment wholly inside a layer w_xpure percent, rows inside the z layers w_zpml percent, band b's rows band_w[b] percent (null: 100): the sequence
// is cut into runs of equal COST, so the tiles that own the absorbing strips get fewer segments (they execute the C-PML branches:
// about twice the loads) and every tile takes the same time per phase -- tiles wait for their neighbours every phase, the slowest
// one sets the pace of all (profiles/r05_pk_trace.txt).
struct PlanCost {
    int nx = 0, npml = 0;            // grid width in cells, layer thickness; nx = 0: no weighting
    int w_xpml = 100, w_xpure = 100, w_zpml = 100;  // percent
    const int *band_w = nullptr;     // [nband] percent, or null
    bool snake = true;               // strips walked top-down / bottom-up alternately (false: all top-down -- an experiment, -DSEPFWI_PROBES)
    int period = 0;                  // > 0: the rows are `period`-row grids stacked on each other (several shots in one launch): the z layers repeat
};
std::string make_persist_plan(int nzc, int nseg, int nwg, int nband, int strip_w, PersistPlan *out, bool edge_first = true,
                              const PlanCost &cost = PlanCost());

// Quiet row segments inside the loop (k_bwd_persist<.., QS>): per tile and row segment (in the order of `seg`) the positions in the tile
// of the six row segments its stencils reach -- rows z-2, z-1, z+1, z+2 of its column, then columns xs-1 and xs+1 of its row -- one
// byte each in a 64-bit word (byte k = neighbour k): 0xff no such segment (outside the grid), 0xfe a segment of another tile.
// Empty when a tile has more than 253 row segments (the quiet variant of the loop is then not used).
std::vector<unsigned long long> make_quiet_neighbours(const PersistPlan &p);

// Several shots in one launch (k_bwd_persist<.., MS>): the plan of the VIRTUAL grid of nshot grids of nzc rows stacked on each other
// (make_persist_plan with nzc * nshot rows and cost.period = nzc; stencils never cross a shot's first or last two rows, so the
// neighbour relations the plan finds there only over-synchronise), its descriptors rewritten to  row inside the shot | segment
// column << 16 | flags | shot << 26.  Empty string, or why not (more than 64 shots, more rows than the descriptor holds).
constexpr int kPlanMaxShots = 64;
std::string make_persist_plan_multishot(int nzc, int nshot, int nseg, int nwg, int nband, int strip_w, PersistPlan *out, bool edge_first = true,
                                        const PlanCost &cost = PlanCost());

}  // namespace sepfwi
