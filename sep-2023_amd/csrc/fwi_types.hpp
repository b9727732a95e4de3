// fwi_types.hpp -- shared host/device plain structs of the propagator.
//
// HBM layout (see DESIGN.md "Data layout"): every 2-D array is row-major with x fastest, i.e. the
// operator boundary's own (nz, nx) orientation (FWI_ops.py:124-127) -- the reference transposes to
// z-fastest on the host every call (Src/libCUFD.cu:71-77,718-724); here no transpose is needed.
// Rows are padded to `pitch` floats (a multiple of 64 -> every row starts on a 256-B line) and only
// the nzc = nz - nPad rows that are ever computed are stored (Appendix A-5 of SURVEY.md: the nPad
// bottom rows are dead).
#pragma once
#include <cstddef>
#include <cstdint>

namespace sepfwi {

struct Grid {
    int nzc;    // stored/computed rows = nz - nPad
    int nx;     // padded nx (columns)
    int pitch;  // floats per stored row
    int nz;     // padded nz of the boundary arrays (nzc + nPad)
    int nPml;
    int zmax;   // last interior row    = nzc - 1 - nPml   (el_stress.cu:92)
    int xmax;   // last interior column = nx  - 1 - nPml
    int nSteps;
    float dt;
    float dt_img;    // time weight of the imaging condition in this launch: dt (every step, the reference), k dt on every k-th step and 0
                     // (imaging skipped) on the others with option img_every = k
    float rdz, rdx;  // 1/dz, 1/dx
    float dz, dx;
    // boundary frame geometry (Boundary.cu:17-27): rows/cols [nPml-2, nPml-2+n?Bnd)
    int nzBnd, nxBnd;  // nzc - 2 nPml + 4, nx - 2 nPml + 4
    int frame_len;     // floats per field per time step in OUR packing (no duplicated corners)
    // launch tiling of the field kernels (filled by the launchers): gx x gy tiles of 64 columns x bz rows,
    // optionally renumbered so that each XCD (blockIdx % 8) owns a contiguous band of tiles
    int gx, gy, bz, xcd_remap;
    int fiber;    // DAS fibre direction: 0 horizontal (exx), 1 vertical (ezz)
    int rk_lazy;  // 1: adjoint kernels read 1/K only inside the layers
    int rho_fly;  // 1: buoyancy averages recomputed from the density in the velocity-type kernels
    int amu_fly;  // 1: the 4-point harmonic mean of mu recomputed from mu in the stress-type kernels
    int nb, shot_fastest;  // batched launches: shots per grid and the order of (tile, shot) in the block index
    int nblk;     // blocks of the launch in the logical numbering (my_cell)
    int qr;       // quiet-skipping kernels: consecutive rows per wave (a quiet wave costs its dispatch whatever it skips)
    int qzw, qn;  // quiet-segment bit maps (option quiet_skip, Fields::q): 32-bit words per segment column, words per map
};

// Five wavefields (or their adjoint twins), each nzc*pitch floats.
struct Fields {
    float *vz, *vx, *szz, *sxx, *sxz;
    // Option quiet_skip (null: off): two bit maps of Grid::qn words each, one bit per ROW SEGMENT (64 columns of one row) -- set once
    // the velocity-type arrays (vz, vx and the C-PML memories written with them) resp. the stress-type arrays of the segment may hold
    // a non-zero value.  A segment whose own bit and the bits within reach of the stencil are all clear is left as it is: every
    // value the update would read is +0, and it would store +0 again.  Layout: column xs + 1 of Grid::qzw words, bit z + 2.
    unsigned int *q = nullptr;
};

// Eight C-PML memory variables.  Forward run: psi of the forward fields; backward run: reused as
// the adjoint memory variables exactly as the reference does (libCUFD.cu:508-515).
struct PmlMem {
    float *dvz_dz, *dvz_dx, *dvx_dz, *dvx_dx;      // updated by the stress kernels
    float *dszz_dz, *dsxz_dx, *dsxz_dz, *dsxx_dx;  // updated by the velocity kernels
};

// Media in internal layout, Pa.
struct Media {
    const float *lam, *mu, *ave_mu, *byc_a, *byc_b;
    const float *rho;  // density itself: the hot kernels rebuild byc_a / byc_b from it (option rho_fly) instead of streaming both
};

// 1-D C-PML profiles; z arrays have nzc entries, x arrays nx entries.  rK = 1/K.
struct PmlCoef {
    const float *a_z, *b_z, *rK_z, *a_zh, *b_zh, *rK_zh;
    const float *a_x, *b_x, *rK_x, *a_xh, *b_xh, *rK_xh;
};

// Imaging accumulators (race-free gather form of el_stress.cu:108-123 / el_velocity.cu:101-110):
//   lam : sum_t -(szz_a+sxx_a)(dvz_dz+dvx_dx) dt          at (z,x)
//   mu  : sum_t -2(szz_a dvz_dz + sxx_a dvx_dx) dt          at (z,x)
//   xz  : sum_t -sxz_a (dvx_dz+dvz_dx) dt                    at the staggered corner (z+1/2,x+1/2)
//   a   : sum_t -vz_a (dszz_dz+dsxz_dx) dt                   at the vz point
//   b   : sum_t -vx_a (dsxz_dz+dsxx_dx) dt                   at the vx point
// The constant-in-time factors (MEGA, mu-harmonic weights, -byc^2/2) and the 4-/2-point spray are
// applied once per call by k_finalize_gradients.
struct ImgAcc {
    float *lam, *mu, *xz, *a, *b;
};

// Receivers that form a horizontal line (the usual horizontal DAS fibre): channel r sits at (z, x0 + r).
// Lets the field kernels sample / inject at their own cells with no extra launch and no lookup.
struct LineRec {
    int z, x0, n;                // n == 0: survey is not a line (separate k_record / k_inject launches)
    float *d_vx, *d_vz, *d_ett;  // forward: this step's seismogram columns (each may be null)
    const float *res;            // backward: this step's residual column
};

// Device-side description of one shot of a BATCHED launch (grid.y = shot of the batch): the per-shot pointers and
// scalars the reference passes as kernel arguments.  Per-step quantities (boundary-frame block, seismogram column,
// residual column, source amplitude) are derived in the kernel from the step index, so all launches of a time loop
// have the same argument list except `it`.
struct ShotDev {
    float *fields;        // vz, vx, szz, sxx, sxz (stride n)
    float *mem;           // forward: 8 C-PML memory variables (stride n)
    float *frame;         // boundary frames [nSteps][5][frame_len] or null
    float *syn;           // seismograms [4][data_len], time-major
    const float *stf;     // tapered source time function, nSteps floats
    float *bmem;          // backward: 8 memory variables, 5 adjoint fields, 5 accumulators (stride n)
    float *adj;
    float *acc;
    const float *res;     // axial-strain residual [it][rec]
    float *stf_grad;      // nSteps floats
    int z_src, x_src;
    int lr_z, lr_x0, lr_n;  // horizontal line of channels (lr_n == 0: separate k_record / k_inject launches)
    int comps, nrec;
    float src_rxz;
    unsigned int *quiet;  // null, or the shot's four quiet-segment maps: forward velocity, forward stress, adjoint velocity, adjoint stress
    const int *rec;       // the shot's channels as flat cell indices (k_record_batch / k_inject_batch: shots whose channels are not a fused line)
    const float *sens;    // null, or nrec x 3 directional sensitivities
};

// Adjoint-source injection inside the persistent backward loop for receivers that are not a fused horizontal line (inject_plan.hpp):
// per row segment that holds target cells, the lanes that receive a value and where their values start in the per-time-step list.
struct InjSeg {  // 32 bytes
    int base[2];                 // index of the segment's first target: vx, vz
    int pad[2];
    unsigned long long mask[2];  // lanes (cells of the segment) that receive a value: vx, vz
};
struct InjArgs {
    const unsigned char *tile_has;  // [tiles] 1: the tile owns a row segment with targets (the others never look anything up)
    const int *lookup;     // [nzc * nseg] row segment -> InjSeg index or -1
    const InjSeg *segs;
    const float *val;      // [nSteps][ntgt]: the residual folded per target cell (k_inject_values)
    int ntgt, nseg;
};

// k_bwd_persist<.., MS = true>: several shots in one launch.  Shot k of the launch has its fields at s.fields + k state_stride, its
// backward-pass arrays (memories, adjoint fields, accumulators) at + k bwd_stride, its boundary frames, residual, source trace and
// source gradient likewise; its scalars are shots[k].
struct MultiShot {
    const ShotDev *shots;
    size_t state_stride, bwd_stride, frame_stride, res_stride;  // floats
    int nshot;
};

// k_bwd_persist<.., QS = true>: quiet row segments inside the loop (option quiet_skip).
struct QuietArgs {
    const unsigned int *maps;       // the shot's forward maps as the forward pass left them: velocities at +0, stresses at +Grid::qn
    const unsigned long long *nbr;  // [tiles][cap]: per row segment the positions IN THE TILE of the six row segments its stencils reach
                                    // (rows z-2, z-1, z+1, z+2; columns xs-1, xs+1), one byte each: 0xff none (outside the grid), 0xfe another tile's
};

constexpr unsigned int kPersistGo = 1, kPersistAbortResidency = 2, kPersistAbortPlacement = 3;  // start rendezvous of k_bwd_persist

// Argument block of the persistent backward time loop (k_bwd_persist), passed BY VALUE: pointers that arrive in the kernel-argument
// segment are known to be global, so the bodies compile to global_load / global_store with graded waits (through a pointer to this
// block they were flat accesses with full drains: profiles/EXPERIMENTS.md #35k).  One shot, time steps it_hi ... it_lo.
struct PersistArgs {
    ShotDev s;
    const float *media, *cz;  // media bundle (stride n), C-PML profile bundle
    size_t n;
    int it_hi, it_lo;
    float src_scale;
    int img_every;
    int nband, per_band;  // tiles: nband row bands (one per XCD) x per_band tiles
    int cap;              // slots per tile in `seg`, and row segments per LDS-resident accumulator array
    const uint32_t *seg;  // [tiles][cap] segment descriptors (persist_plan.hpp)
    const struct TileHdr *hdr;
    unsigned int *flags;     // [tiles] x 32 words (one 128-B line each): phases of the pass whose edge part is complete
    unsigned int *band_xcc;  // [nband]: XCC_ID the band's workgroups run on (0xffffffff before the first reports); [8] workgroups arrived,
                             // [9] the start rendezvous' decision (kPersist*)
    int *err;                // 0, 1 a wait inside the pass timed out
    int phase0;              // phases of the pass done by earlier launches
    int nosync;              // -DSEPFWI_PROBES builds, timing experiments only: no waits, no flags, no agent-scope accesses (results wrong)
    int lock;                // -DSEPFWI_PROBES builds, timing experiments only: phases interleaved (option pk_lock)
    int prio;                // 1: wave priorities interleave the CU's two workgroups (kernels.hip)
    const InjArgs *injp;     // k_bwd_persist<LMASK, true>: general receivers -- the tables of the shot's adjoint source, in device memory (else null)
    MultiShot ms;            // k_bwd_persist<LMASK, false, true>: the shots of the launch (else unused)
    QuietArgs q;             // k_bwd_persist<LMASK, false, false, true>: quiet row segments (else unused)
};

struct Frame {  // boundary-saving storage, one block of 5*frame_len floats per time step
    float *buf;  // [nSteps][5][frame_len]  order: szz, sxz, sxx, vz, vx (Boundary.cu:57-80)
};

}  // namespace sepfwi
