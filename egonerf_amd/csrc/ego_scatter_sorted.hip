// libegonerf_hip.so, part 9: the table-gradient scatter without atomics - bit-reproducible (SURVEY 5 "sorted-segment mode", VERDICT r04
// item 3).  Backward of F.grid_sample in compute_densityfeature / compute_appfeature (models/EgoNeRF.py:291-347, :349-413) under
// train.py:312-314, for the tuned table shapes (16 density / 48 appearance components).
//
// k_vm_scatter (ego_train.inc) walks rays and sends one float-atomic line per (cell run, tap): ~16 M atomic line requests per 8192 x 256
// step at the L2's ~21 G/s plus ~0.95 ms of cell bookkeeping, and a sum whose order changes from run to run.  Here the step's samples are
// binned by texel CELL once (three stable LSD radix sorts of 18-bit keys, 9 bits per pass, all three sorts in the same launches:
// k_radix_hist / k_radix_scan / k_radix_scatter below), and every gradient texel is then written exactly once from sums taken in a
// fixed order:
//
//   ego_scatter_sort      : coords -> three permutations + cell start offsets (needs only the forward's coordinates: it runs on the side
//                           stream next to the dumping shade forward, and both fields share it)
//       sort 0: key (grid, phi cell, r cell)     -> cells of plane 1 (x = r, y = phi), ranges of line 0 (phi)
//       sort 1: key (grid, r cell, theta cell)   -> cells of plane 0 (x = r, y = theta), ranges of line 2 (r)
//       sort 2: key (grid, theta cell, phi cell) -> cells of plane 2 (x = theta, y = phi), ranges of line 1 (theta)
//     a cell = the unclamped west tap index + 1 (0 .. n); a sample whose two taps of an axis are both out of range has no gradient
//     through that axis and sorts behind everything else
//   k_sorted_plane        : one 16-lane group per cell (lane = channel of a 64-byte line, as in k_vm_scatter), four cells per wave: the
//                           cell's samples (contiguous, in ascending sample order: the sort is stable) are added in that order into four
//                           corner sums in registers, which go to a cell buffer
//   k_sorted_plane_final  : texel = the four corner sums of its four neighbouring cells, added in a fixed order, one store
//   k_sorted_line         : a line cell holds thousands of samples: fixed 256-sample sub-blocks of its range, one wave each -> partial sums
//   k_sorted_line_final   : texel = its two cells' partials, added in sub-block order
//
// No zero fill of the gradient tables is needed (every texel is written), no atomics, and two runs return the same bits.
//
// Round 6: ONE pass over the gradient activations.  The line kernels above read every sample's d a second time
// (1.2 GB of dv in another random order) plus four random co-plane taps per sample: 0.68 ms and 4.6 GB of fetches per step.  A plane
// cell's samples all use the SAME four plane texels, so the line gradient of the plane's own line (the THIRD axis, the one that is not
// in the sort key) costs four multiply-adds per sample next to the plane's - but its samples land in arbitrary line texels, so the sums
// cannot be taken in a fixed order.  They are taken in an order-INDEPENDENT arithmetic instead: every contribution is converted to a
// 64-bit fixed-point integer (one power-of-two unit per table, derived from max |d| x max |plane texel|, both computed on the device:
// >= 40 bits below the largest possible contribution - an fp32 sum keeps 24) and added with integer LDS atomics into a per-workgroup
// table of the line; integer addition is associative, so any order gives the same bits.  k_sorted_walk = the plane walk + those
// four multiply-adds + two ds_add_u64 per channel lane and sample; k_fused_line_final adds the workgroups' tables and converts once.
// Lines too long for the LDS are cut into BLOCKS by the sort keys (a workgroup's table holds one block's window), so every line rides
// along on every shipped grid; the separate line kernels stay as the fall-back (EGO_SORTED_LINES=separate forces them for every line,
// EGO_SORTED_WALK=0 the whole round-5 form).  Late in the round the appearance walk also took the basis gradient along (BAS: it holds
// plane value x line value, one bf16 MFMA per iteration multiplies it with the sample's feature-slot gradients): no v dump, no d(basis) pass.

#include "ego_device.h"
#include "ego_host.h"
#include <atomic>
#include <stdlib.h>

namespace {

constexpr int SUB = 256;       // samples per line sub-block
constexpr int RBITS = 9, RADIX = 1 << RBITS, RTILE = 4096;   // radix sort: digit bits, buckets, elements per workgroup tile (256 threads x 16)
constexpr int CMAX = 48;       // channels of the widest field (appearance)
constexpr int FUSED_MAX_WG = 320;   // workgroups of a fused launch (one per CU: 256 on MI355X; room for a larger part)

// sort s: major / minor axis of its key (0 r, 1 theta, 2 phi), the plane whose cells it bins and the line whose ranges it bins
__host__ __device__ constexpr int sort_major(int s) { return s == 0 ? 2 : s == 1 ? 0 : 1; }
__host__ __device__ constexpr int sort_minor(int s) { return s == 0 ? 0 : s == 1 ? 1 : 2; }
__host__ __device__ constexpr int sort_plane(int s) { return s == 0 ? 1 : s == 1 ? 0 : 2; }
__host__ __device__ constexpr int sort_line(int s) { return s == 0 ? 0 : s == 1 ? 2 : 1; }

struct SortGeom {
  int32_t res[3];      // N_r, N_theta, N_phi
  int64_t M;
  uint32_t K[3];       // keys per sort = 2 (n_major + 1) (n_minor + 1); key K = "no gradient through this pair of axes"
  uint32_t LC[3];      // line cells per sort = 2 (n_major + 1)
  uint32_t nsub_max;   // upper bound of the line sub-blocks of one sort
  int bits;            // key bits (covers max K)
  // byte offsets into the workspace
  int64_t perm[3], start[3], suboff[3], scratch, total;
  int64_t stepsum[3], steps[3];   // the walk's step list (round 6): per cell the number of 16-sample steps before it; the steps themselves
  int64_t costsum[3];             // per cell the COST of the steps before it (what the walk is dealt by)
  // line blocks (round 6): sort s's key carries, above its two plane axes, the BLOCK of the sample's cell along the third axis (nb[s]
  // blocks of bs[s] cells), so that a workgroup of the walk needs only one block's texels of the fused line in LDS
  uint32_t nb[3], bs[3], kc[3];   // kc = cells of one (grid, block): (n_major + 1) (n_minor + 1)
  int64_t step_cap[3];            // entries of steps[s]
  // sort-phase view of the scratch region
  int64_t keys_in[3], k1[3], v1[3], k2[3], hist[3], scanpart[3];
  uint32_t nblocks;    // radix tiles (RTILE elements each)
  int passes;          // ceil(bits / 9)
  // scatter-phase view of the scratch region
  int64_t cellbuf[3], linepart[3];
  int64_t fx, fpart;   // fused form: the fixed-point scale block, the per-workgroup integer line tables
  int64_t bpart;       // the walk's per-wave shares of d(basis): [FUSED_MAX_WG x WALK_NW_BAS][6][64][4] floats
  int64_t fpart_stride;   // entries (8 bytes each) per workgroup table
  bool dense_cells;    // cell buffer indexed by cell (K <= M) or by the cell's first sorted position (K > M: at most M cells hold samples)
  uint32_t cell_slots[3];
};

inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

// ---- launch plan of the walk --------------------------------------------------------------------------------------------------------
// Sort s serves plane sort_plane(s) AND the line of the axis that is not in its key (line index sort_plane(s)).  A workgroup keeps ONE
// line block's window of that line's integer table for ONE grid in LDS: (bs + 1) x C x 8 bytes beside the waves' records; the sort keys
// carry the block (line_blocks / make_geom: the headline grid [150, 172, 516] needs 1 / 1 / 3 blocks for theta / phi / r at C = 48 -
// the r line alone would be 198 KB).  Only a line whose block window does not fit even in WALK_MAX_BLOCKS blocks keeps the two-pass form.
constexpr int FUSED_LDS_LIMIT = 160 * 1024 - 1024;
constexpr int WALK_MAX_BLOCKS = 16, WALK_MAX_SEG = 6 * WALK_MAX_BLOCKS;   // line blocks per sort; (sort, grid, block) segments of a launch
constexpr int WALK_NW_APP = 12, WALK_NW_DENS = 16, WALK_NW_BAS = 8;   // waves per workgroup: 144+ VGPRs at 48 channels (three waves per SIMD), ~110 at 16
struct FusedPlan {
  int nw;
  bool do_line[3];   // per sort
  int entries_max;   // 8-byte table entries per workgroup (0: no fused line at all)
  int lds_bytes;
};
__host__ __device__ constexpr int fused_line_axis(int s) { return 2 - (s == 0 ? 1 : s == 1 ? 0 : 2); }   // vm_line_ax(sort_plane(s))
inline int walk_wave_bytes(int C) { return 12 * 64 * 4 + (C / 16) * 4 * 64 * 4; }   // sizeof(WalkLds<C / 16>)

// the round-5 form (lockstep plane kernel + line kernels) stays reachable for A/B: EGO_SORTED_WALK=0; EGO_SORTED_LINES=separate keeps the
// two-pass LINES under the walk's planes.  Both are read per call: a sort and the scatters that use it must see the same values.
bool walk_wanted() {
  const char* e = getenv("EGO_SORTED_WALK");
  return !(e && e[0] == '0');
}
bool lines_separate() {
  const char* e = getenv("EGO_SORTED_LINES");
  return e && e[0] == 's';
}

// blocks of the third axis (n texels, n + 1 cells): the fewest whose window of bs + 1 texels x 48 channels x 8 bytes fits the LDS
// beside the 48-channel walk's wave records (the 16-channel walk has more room)
inline void line_blocks(int n, uint32_t* nb, uint32_t* bs) {
  const int room = FUSED_LDS_LIMIT - WALK_NW_APP * walk_wave_bytes(CMAX) - 1024;
  uint32_t k = 1;
  while (k < WALK_MAX_BLOCKS && ((int64_t)((n + 1 + k - 1) / k) + 1) * CMAX * 8 > room) ++k;
  *nb = k; *bs = (uint32_t)(n + 1 + k - 1) / k;
}

SortGeom make_geom(const int32_t res[3], int64_t M) {
  SortGeom G{};
  G.M = M;
  uint32_t kmax = 0, lcmax = 0;
  for (int a = 0; a < 3; ++a) G.res[a] = res[a];
  bool blocked = walk_wanted() && !lines_separate();
  if (blocked) {   // all or nothing: a line that does not fit even in WALK_MAX_BLOCKS blocks needs the two-pass kernels, which read the plain key layout
    const int room = FUSED_LDS_LIMIT - WALK_NW_APP * walk_wave_bytes(CMAX) - 1024;
    for (int s = 0; s < 3; ++s) {
      uint32_t nb_, bs_;
      line_blocks(res[fused_line_axis(s)], &nb_, &bs_);
      if ((int64_t)(bs_ + 1) * CMAX * 8 > room) blocked = false;
    }
  }
  for (int s = 0; s < 3; ++s) {
    const uint32_t nmaj = (uint32_t)res[sort_major(s)] + 1, nmin = (uint32_t)res[sort_minor(s)] + 1;
    G.nb[s] = 1; G.bs[s] = (uint32_t)res[fused_line_axis(s)] + 1;
    if (blocked) line_blocks(res[fused_line_axis(s)], &G.nb[s], &G.bs[s]);
    G.kc[s] = nmaj * nmin;
    G.K[s] = 2u * G.nb[s] * nmaj * nmin;
    G.LC[s] = 2u * nmaj;
    kmax = G.K[s] > kmax ? G.K[s] : kmax;
    lcmax = G.LC[s] > lcmax ? G.LC[s] : lcmax;
  }
  G.bits = 1;
  while ((1ull << G.bits) <= kmax) ++G.bits;      // keys 0 .. K inclusive
  G.nsub_max = (uint32_t)(M / SUB) + lcmax + 1;
  int64_t o = 0;
  for (int s = 0; s < 3; ++s) { G.perm[s] = o; o = align256(o + 4 * M); }
  for (int s = 0; s < 3; ++s) { G.start[s] = o; o = align256(o + 4 * ((int64_t)G.K[s] + 2)); }
  for (int s = 0; s < 3; ++s) { G.suboff[s] = o; o = align256(o + 4 * ((int64_t)G.LC[s] + 1)); }
  for (int s = 0; s < 3; ++s) { G.stepsum[s] = o; o = align256(o + 4 * ((int64_t)G.K[s] + 2)); }
  for (int s = 0; s < 3; ++s) { G.costsum[s] = o; o = align256(o + 4 * ((int64_t)G.K[s] + 2)); }
  for (int s = 0; s < 3; ++s) {
    // a cell of n samples takes ceil(n / 16) steps: at most M / 16 + one per cell that holds samples
    G.step_cap[s] = M / 16 + (M < (int64_t)G.K[s] ? M : (int64_t)G.K[s]) + 1;
    G.steps[s] = o; o = align256(o + 16 * G.step_cap[s]);
  }
  G.scratch = o;
  // sort phase
  int64_t a = o;
  G.passes = (G.bits + RBITS - 1) / RBITS;
  G.nblocks = (uint32_t)((M + RTILE - 1) / RTILE);
  for (int s = 0; s < 3; ++s) { G.keys_in[s] = a; a = align256(a + 4 * M); }
  for (int s = 0; s < 3; ++s) { G.k1[s] = a; a = align256(a + 4 * M); }
  for (int s = 0; s < 3; ++s) { G.v1[s] = a; a = align256(a + 4 * M); }
  for (int s = 0; s < 3; ++s) { G.k2[s] = a; a = align256(a + 4 * M); }
  for (int s = 0; s < 3; ++s) { G.hist[s] = a; a = align256(a + 4 * ((int64_t)RADIX * G.nblocks + RADIX)); }   // + the digit totals
  for (int s = 0; s < 3; ++s) { G.scanpart[s] = a; a = align256(a + 4 * 2 * ((int64_t)G.K[s] / 4096 + 2)); }   // k_step_scan's per-workgroup totals (steps, cost)
  // scatter phase
  int64_t b = o;
  // cell buffer: only cells that hold samples are ever written or read - at most min(K, M) of them (K x 4 x 48 floats was
  // 1.2 GB on the [300, 346, 1036] grid whatever the batch)
  G.dense_cells = (int64_t)kmax <= M;
  for (int s = 0; s < 3; ++s) {
    G.cell_slots[s] = G.dense_cells ? G.K[s] : (uint32_t)M;   // sparse: slot = the cell's first sorted position (< M, distinct per non-empty cell)
    G.cellbuf[s] = b; b = align256(b + 4 * (int64_t)G.cell_slots[s] * 4 * CMAX);
  }
  for (int s = 0; s < 3; ++s) { G.linepart[s] = b; b = align256(b + 4 * (int64_t)G.nsub_max * 2 * CMAX); }
  // fused form (shares the line-partial region's place in time, not its bytes: both forms are sized so that either can run)
  G.fx = b; b = align256(b + 256 + 4 * 7 * 128 + 4 * (WALK_MAX_SEG + 1));   // FxScale + k_fx_absmax's per-workgroup maxima + the walk's deal
  {
    // a workgroup's line table: one block's window of the third axis (bs + 1 texels) x the widest field's channels
    int64_t emax = 0;
    for (int s = 0; s < 3; ++s) {
      const int64_t e = (int64_t)(G.bs[s] + 1) * CMAX;
      const int room = FUSED_LDS_LIMIT - WALK_NW_APP * walk_wave_bytes(CMAX) - 1024;
      if (e * 8 <= room && e > emax) emax = e;
    }
    G.fpart_stride = emax;
    G.fpart = b; b = align256(b + 8 * G.fpart_stride * FUSED_MAX_WG);
  }
  G.bpart = b; b = align256(b + (int64_t)FUSED_MAX_WG * WALK_NW_BAS * 6 * 256 * 4);
  G.total = a > b ? a : b;
  return G;
}

// which lines the walk of a C-channel field takes along: a sort's line rides along when one block's window of it fits the LDS
inline FusedPlan fused_plan(const SortGeom& G, int C) {
  FusedPlan P{};
  P.nw = C > 16 ? WALK_NW_APP : WALK_NW_DENS;
  const int fixed = P.nw * walk_wave_bytes(C) + 1024;   // + the deal table
  const int room = FUSED_LDS_LIMIT - fixed;
  for (int s = 0; s < 3; ++s) {
    const int64_t bytes = (int64_t)(G.bs[s] + 1) * C * 8;
    P.do_line[s] = !lines_separate() && bytes <= room;
    if (P.do_line[s] && (int)(bytes / 8) > P.entries_max) P.entries_max = (int)(bytes / 8);
  }
  P.lds_bytes = fixed + P.entries_max * 8;
  return P;
}

// the unclamped west tap index + 1 (0 .. n) with lin_setup's arithmetic, or -1 when both taps are out of range
__device__ __forceinline__ int cell_of(float xhat, int n) {
  const float ix = __fmul_rn(__fadd_rn(xhat, 1.0f), 0.5f * (float)(n - 1));
  const float flc = fminf(fmaxf(floorf(ix), -2.0f), (float)n);
  const int i0 = (int)flc;
  return (i0 < -1 || i0 > n - 1) ? -1 : i0 + 1;
}

struct KeyArgs {
  uint32_t K[3], nb[3], bs[3];
};
// key of sort s = ((grid * nb + block of the third axis' cell) * (n_major + 1) + major cell) * (n_minor + 1) + minor cell; a sample
// without a gradient through the third axis (both taps out of range: its line weights are 0) goes to block 0
__global__ void k_sort_keys(const float* __restrict__ coords, int64_t M, int nr, int nth, int nph, KeyArgs A,
                            uint32_t* __restrict__ k0, uint32_t* __restrict__ k1, uint32_t* __restrict__ k2) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const f32x4 cc = ((const f32x4*)coords)[m];
  const uint32_t g = cc.w != 0.f ? 1u : 0u;
  const int cr = cell_of(cc.x, nr), cth = cell_of(cc.y, nth), cph = cell_of(cc.z, nph);
  const uint32_t b0 = cth < 0 ? 0u : (uint32_t)cth / A.bs[0], b1 = cph < 0 ? 0u : (uint32_t)cph / A.bs[1], b2 = cr < 0 ? 0u : (uint32_t)cr / A.bs[2];
  k0[m] = (cph < 0 || cr < 0) ? A.K[0] : ((g * A.nb[0] + b0) * (uint32_t)(nph + 1) + (uint32_t)cph) * (uint32_t)(nr + 1) + (uint32_t)cr;
  k1[m] = (cr < 0 || cth < 0) ? A.K[1] : ((g * A.nb[1] + b1) * (uint32_t)(nr + 1) + (uint32_t)cr) * (uint32_t)(nth + 1) + (uint32_t)cth;
  k2[m] = (cth < 0 || cph < 0) ? A.K[2] : ((g * A.nb[2] + b2) * (uint32_t)(nth + 1) + (uint32_t)cth) * (uint32_t)(nph + 1) + (uint32_t)cph;
}

// ---- stable LSD radix sort of (key, sample index), 9 bits per pass, the three sorts side by side (blockIdx.y) -----------------------
// Plain kernels (no look-back between workgroups, no library state): the whole sort is graph-capturable and bit-reproducible.  A pass =
//   k_radix_hist    : per 4096-element tile, the digit histogram (LDS integer atomics) -> hist[digit][tile]
//   k_radix_scan    : per digit, exclusive scan of its per-tile counts + the digit's total (the digit bases are a 512-value scan that every
//                     scatter workgroup does for itself): where each tile's run of each digit starts
//   k_radix_scatter : the tile again: every element's rank among the EARLIER elements of its digit (wave w owns elements [1024 w, 1024 w +
//                     1024) of the tile and walks them 64 at a time in order; inside an iteration the equal-digit lanes are found with 9
//                     ballots) -> stable position -> (key, value) stored
struct RadixArgs {
  const uint32_t* kin[3];
  const uint32_t* vin[3];    // nullptr: the value is the element's index (first pass)
  uint32_t* kout[3];
  uint32_t* vout[3];
  uint32_t* hist[3];
  uint32_t* dsum[3];         // [RADIX] per-digit totals of the pass
  int64_t M;
  uint32_t nblocks;
  int shift;
};

__global__ __launch_bounds__(256) void k_radix_hist(RadixArgs A) {
  __shared__ uint32_t h[RADIX];
  const int s = blockIdx.y, t = threadIdx.x;
  for (int i = t; i < RADIX; i += 256) h[i] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RTILE;
#pragma unroll 4
  for (int j = 0; j < RTILE / 256; ++j) {
    const int64_t idx = base + j * 256 + t;
    if (idx < A.M) atomicAdd(&h[(A.kin[s][idx] >> A.shift) & (RADIX - 1)], 1u);
  }
  __syncthreads();
  for (int i = t; i < RADIX; i += 256) A.hist[s][(int64_t)i * A.nblocks + blockIdx.x] = h[i];
}

// exclusive scan of a workgroup's 256 values (one per thread); returns the thread's prefix, *total = the sum
__device__ __forceinline__ uint32_t block_excl_scan256(uint32_t v, uint32_t* wsum /* [4] shared */, uint32_t* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t u = __shfl_up(inc, d, 64);
    if (lane >= d) inc += u;
  }
  __syncthreads();   // wsum may still be read from a previous call
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t before = 0;
  for (int w = 0; w < wv; ++w) before += wsum[w];
  *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return before + inc - v;
}

// one workgroup per (digit, sort): in-place exclusive scan of the digit's per-tile counts; the digit's total goes to dsum[digit]
// (k_radix_scatter turns the 512 totals into digit bases itself)
__global__ __launch_bounds__(256) void k_radix_scan(RadixArgs A) {
  __shared__ uint32_t wsum[4];
  uint32_t* h = A.hist[blockIdx.y] + (int64_t)blockIdx.x * A.nblocks;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < A.nblocks; base += 256) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < A.nblocks ? h[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_excl_scan256(v, wsum, &total);
    if (i < A.nblocks) h[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) A.dsum[blockIdx.y][blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void k_radix_scatter(RadixArgs A) {
  __shared__ uint32_t wcnt[4][RADIX];
  __shared__ uint32_t lbase[RADIX], gbase[RADIX];   // where a digit's run starts inside the sorted tile / in the output
  __shared__ uint32_t skey[RTILE], sval[RTILE];     // the tile in sorted order: the output is then written in runs, not element by element
  __shared__ uint32_t wsum[4];
  const int s = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int i = t; i < 4 * RADIX; i += 256) (&wcnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t tbase = (int64_t)blockIdx.x * RTILE, wbase = tbase + w * (RTILE / 4);
  uint32_t key[RTILE / 256];
#pragma unroll
  for (int it = 0; it < RTILE / 256; ++it) {
    const int64_t idx = wbase + it * 64 + lane;
    key[it] = idx < A.M ? A.kin[s][idx] : 0u;
    if (idx < A.M) atomicAdd(&wcnt[w][(key[it] >> A.shift) & (RADIX - 1)], 1u);
  }
  __syncthreads();
  {   // two digits per thread (2 t, 2 t + 1): the tile's digit starts, and the digit bases from the 512 digit totals of the pass
    const uint32_t c0 = wcnt[0][2 * t] + wcnt[1][2 * t] + wcnt[2][2 * t] + wcnt[3][2 * t];
    const uint32_t c1 = wcnt[0][2 * t + 1] + wcnt[1][2 * t + 1] + wcnt[2][2 * t + 1] + wcnt[3][2 * t + 1];
    uint32_t total;
    const uint32_t ex = block_excl_scan256(c0 + c1, wsum, &total);
    lbase[2 * t] = ex; lbase[2 * t + 1] = ex + c0;
    const uint32_t d0 = A.dsum[s][2 * t], d1 = A.dsum[s][2 * t + 1];
    const uint32_t exg = block_excl_scan256(d0 + d1, wsum, &total);
    gbase[2 * t] = exg + A.hist[s][(int64_t)(2 * t) * A.nblocks + blockIdx.x];
    gbase[2 * t + 1] = exg + d0 + A.hist[s][(int64_t)(2 * t + 1) * A.nblocks + blockIdx.x];
  }
  __syncthreads();
  for (int d = t; d < RADIX; d += 256) {   // counts -> where wave w's run of digit d starts in the sorted tile
    uint32_t run = lbase[d];
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { const uint32_t c = wcnt[ww][d]; wcnt[ww][d] = run; run += c; }
  }
  __syncthreads();
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
  for (int it = 0; it < RTILE / 256; ++it) {
    const int64_t idx = wbase + it * 64 + lane;
    const bool ok = idx < A.M;
    const uint32_t d = (key[it] >> A.shift) & (RADIX - 1);
    unsigned long long m = __ballot(ok);
#pragma unroll
    for (int b = 0; b < RBITS; ++b) {
      const unsigned long long bal = __ballot(ok && ((d >> b) & 1u));
      m &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t rank = (uint32_t)__popcll(m & lt);
    uint32_t off = 0;
    if (ok) off = wcnt[w][d];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (ok && rank == 0) wcnt[w][d] = off + (uint32_t)__popcll(m);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (ok) {
      skey[off + rank] = key[it];
      sval[off + rank] = A.vin[s] ? A.vin[s][idx] : (uint32_t)idx;
    }
  }
  __syncthreads();
  const int n_tile = (int)(A.M - tbase < RTILE ? A.M - tbase : RTILE);
  for (int i = t; i < n_tile; i += 256) {
    const uint32_t k = skey[i], d = (k >> A.shift) & (RADIX - 1);
    const uint32_t pos = gbase[d] + ((uint32_t)i - lbase[d]);
    A.kout[s][pos] = k;
    A.vout[s][pos] = sval[i];
  }
}

// start[k] = first sorted position whose key is >= k, k = 0 .. K + 1 (start[K] = the number of samples with a gradient)
struct StartArgs {
  const uint32_t* sorted[3];
  uint32_t* start[3];
  uint32_t* suboff[3];
  uint32_t* stepsum[3];
  uint32_t* costsum[3];
  uint4* steps[3];
  uint32_t* scanpart[3];   // k_step_scan's per-workgroup totals
  uint32_t K[3], LC[3], nmin1[3];
  int64_t M;
};

__global__ void k_cell_starts(StartArgs A) {
  const uint32_t* __restrict__ sorted = A.sorted[blockIdx.y];
  uint32_t* __restrict__ start = A.start[blockIdx.y];
  const uint32_t K = A.K[blockIdx.y];
  const int64_t M = A.M;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > K + 1) return;
  int64_t lo = 0, hi = M;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted[mid] < k) lo = mid + 1; else hi = mid;
  }
  start[k] = (uint32_t)lo;
}

// suboff[lc] = number of 256-sample sub-blocks of the line cells before lc (exclusive scan; suboff[LC] = total); one workgroup
__global__ __launch_bounds__(1024) void k_line_suboff(StartArgs A) {
  const uint32_t* __restrict__ start = A.start[blockIdx.y];
  uint32_t* __restrict__ suboff = A.suboff[blockIdx.y];
  const uint32_t LC = A.LC[blockIdx.y], nmin1 = A.nmin1[blockIdx.y];
  __shared__ uint32_t wsum[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < LC; base += 1024) {
    const uint32_t lc = base + t;
    uint32_t n = 0;
    if (lc < LC) n = (start[(lc + 1) * nmin1] - start[lc * nmin1] + SUB - 1) / SUB;
    uint32_t inc = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t v = __shfl_up(inc, d, 64);
      if (lane >= d) inc += v;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wv; ++w) before += wsum[w];
    uint32_t all = 0;
    for (int w = 0; w < 16; ++w) all += wsum[w];
    if (lc < LC) suboff[lc] = carry + before + inc - n;
    carry += all;
    __syncthreads();
  }
  if (t == 0) suboff[LC] = carry;
}

// ---- the walk's step list -------------------------------------------------------------------------------------------------------------
// A STEP = up to 16 consecutive sorted samples of one cell (what a 16-lane group of k_sorted_walk handles at a time).
// stepsum[k] = number of steps of the cells before k (k = 0 .. K; the cells of grid 0 come first), one workgroup per sort;
// steps[j] = {first sorted position, cell, samples | first step of its cell << 8 | last << 9, 0}.  With the list the walk is dealt in
// EQUAL numbers of steps per group, whatever the cells' sizes - a first version that dealt cells in chunks had its slowest wave at 3.5 x
// the mean (tools/sorted_probe.py PROBE_PROF on a -DEGO_WALK_PROF build).
// cost of a cell's steps: a step = its fixed part (prefetches, set-up, record: 0.6 of an iteration's time, measured with
// -DEGO_WALK_PROF: 3.2 k against 5.3 k clocks in the 48-channel walk) + ceil(samples / 4) stage-2 iterations (U = 4 samples per group)
__device__ __forceinline__ uint32_t cell_cost(uint32_t n) {   // in fifths of an iteration: a step's fixed part = 3, an iteration = 5
  const uint32_t full = n / 16u, r = n % 16u;
  return full * 23u + (r ? 3u + 5u * ((r + 3u) / 4u) : 0u);
}

// stepsum / costsum = exclusive prefix sums over the sort's cells, in three launches: per 4096-cell workgroup the local prefixes + its
// total (PHASE 0), the totals' scan by one workgroup (1), the offsets added (2).  blockIdx.y = 0 steps, 1 cost; z = sort.  (A
// single-workgroup loop took 0.1 ms.)
template <int PHASE>
__global__ __launch_bounds__(1024) void k_step_scan(StartArgs A) {
  const int s = blockIdx.z;
  const uint32_t* __restrict__ start = A.start[s];
  const bool cost = blockIdx.y == 1;
  uint32_t* __restrict__ out = cost ? A.costsum[s] : A.stepsum[s];
  uint32_t* __restrict__ part = A.scanpart[s] + (cost ? (A.K[s] / 4096u + 2u) : 0u);
  const uint32_t K = A.K[s], nblk = (K + 4095u) / 4096u;
  __shared__ uint32_t wsum[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (PHASE == 1) {   // one workgroup: exclusive scan of the workgroup totals, the grand total behind them
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblk; base += 1024) {
      const uint32_t i = base + (uint32_t)t;
      const uint32_t v = i < nblk ? part[i] : 0u;
      uint32_t inc = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d, 64); if (lane >= d) inc += u; }
      if (lane == 63) wsum[wv] = inc;
      __syncthreads();
      uint32_t before = 0, all = 0;
      for (int w = 0; w < 16; ++w) { if (w < wv) before += wsum[w]; all += wsum[w]; }
      if (i < nblk) part[i] = carry + before + inc - v;
      carry += all;
      __syncthreads();
    }
    if (t == 0) { out[K] = carry; out[K + 1] = carry; }
    return;
  }
  if (blockIdx.x >= nblk) return;
  const uint32_t k0 = blockIdx.x * 4096u + 4u * (uint32_t)t;
  if (PHASE == 2) {
    const uint32_t off = part[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (k0 + i < K) out[k0 + i] += off;
    return;
  }
  uint32_t n[4], own = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t k = k0 + i;
    const uint32_t ns = k < K ? start[k + 1] - start[k] : 0u;
    n[i] = cost ? cell_cost(ns) : (ns + 15u) / 16u;
    own += n[i];
  }
  uint32_t inc = own;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d, 64); if (lane >= d) inc += u; }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (int w = 0; w < 16; ++w) { if (w < wv) before += wsum[w]; all += wsum[w]; }
  uint32_t run = before + inc - own;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (k0 + i < K) out[k0 + i] = run;
    run += n[i];
  }
  if (t == 0) part[blockIdx.x] = all;
}

__global__ void k_step_fill(StartArgs A) {
  const int s = blockIdx.y;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= A.K[s]) return;
  const uint32_t c = k;
  const uint32_t a = A.start[s][c], n = A.start[s][c + 1] - a;
  if (!n) return;
  uint4* out = A.steps[s] + A.stepsum[s][k];
  const uint32_t nb = (n + 15u) / 16u;
  for (uint32_t b = 0; b < nb; ++b) {
    const uint32_t cnt = min(16u, n - 16u * b);
    out[b] = uint4{a + 16u * b, c, cnt | (b == 0 ? 256u : 0u) | (b + 1 == nb ? 512u : 0u), 0u};
  }
}

struct GradTables {
  float* plane[2][3];
  float* line[2][3];
};

struct SortedArgs {
  DevField F;
  GradTables G;
  const float* coords;   // [M][4]
  const float* d;        // DENS: dfeat [M]; else k_shade_bwd's blocked dv
  const uint32_t* perm[3];
  const uint32_t* start[3];
  const uint32_t* suboff[3];
  const uint32_t* stepsum[3];
  const uint32_t* costsum[3];
  const uint4* steps[3];
  float* cellbuf[3];
  float* linepart[3];
  uint32_t K[3], LC[3];
  int dense_cells;
  uint32_t nb[3], bs[3], kc[3];   // line blocks of the sort keys (SortGeom)
  int line_mask;   // bit s: k_sorted_line / k_sorted_line_final take the line of sort s (the walk takes the others)
};

// where cell k of sort s keeps its four corner sums (only called for cells that hold samples)
__device__ __forceinline__ int64_t cell_slot(const SortedArgs& A, int s, uint32_t k) { return A.dense_cells ? (int64_t)k : (int64_t)A.start[s][k]; }

// fold the four 16-lane groups of a wave: lanes 0..15 end up with (g0 + g1) + (g2 + g3)
__device__ __forceinline__ float fold_groups(float v) {
  v += __shfl_down(v, 16, 64);
  v += __shfl_down(v, 32, 64);
  return v;
}

// ---- planes: one wave per cell ----------------------------------------------------------------------------------------------------
// Both reductions run in two stages per batch of 64 sorted samples.  Stage 1, lane = sample: permutation entry -> coordinates -> tap
// weights and offsets, once per sample (not once per channel lane), written to the wave's LDS record (structure of arrays: conflict-free
// writes, broadcast reads).  Stage 2, lane = channel: the four 16-lane groups take samples t, t + 1, t + 2, t + 3, U steps are in flight
// at a time and their loads are issued back to back - the first form of these kernels chased perm -> coords -> taps once per sample and
// was bound by that latency chain (0.66 / 0.65 ms for the appearance planes / lines).
constexpr int REC_F = 12;   // dwords per sample record
struct WaveRec {
  uint32_t f[REC_F][64];
};
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One 16-lane group per cell, four cells per wave (a cell holds 12 - 40 samples on the barbershop grid at 8192 x 256: a wave per cell
// spent its life waiting for three dependent round trips with 4 096 waves in flight; four cells per wave share them).  Stage 1: lane
// 16 q + j sets up sample j of the current 16-sample batch of cell q; stage 2: group q walks its own batch in order, U samples in flight.
template <int C, bool DENS, int S_>
__device__ __forceinline__ void sorted_plane(const SortedArgs& A, WaveRec& R) {
#pragma clang fp contract(fast)
  constexpr int NL = C / 16, I = sort_plane(S_), U = 4;
  constexpr int AX = vm_plane_x(I), AY = vm_plane_y(I), AL = vm_line_ax(I);
  const int lane = threadIdx.x & 63, c16 = lane & 15, q = lane >> 4;
  const uint32_t k_raw = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 4u + (uint32_t)q;
  const bool cell_ok = k_raw < A.K[S_];
  const uint32_t k = cell_ok ? k_raw : A.K[S_] - 1;
  const uint32_t a = A.start[S_][k], b = cell_ok ? A.start[S_][k + 1] : a;
  const uint32_t n_mine = b - a;
  uint32_t n_max = max(n_mine, (uint32_t)__shfl_xor((int)n_mine, 16, 64));
  n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, 32, 64));
  if (n_max == 0) return;   // four empty cells: nothing written, k_sorted_plane_final does not read them
  const int nmin1 = A.F.res[sort_minor(S_)] + 1, nmaj1 = A.F.res[sort_major(S_)] + 1;
  const int cmin = (int)(k % (uint32_t)nmin1), t_ = (int)(k / (uint32_t)nmin1);
  const int cmaj = t_ % nmaj1, g = t_ / nmaj1;
  const int cX = sort_major(S_) == AX ? cmaj : cmin, cY = sort_major(S_) == AY ? cmaj : cmin;
  const int W = A.F.res[AX], H = A.F.res[AY], NLn = A.F.res[AL];
  const int x0 = max(cX - 1, 0), x1 = min(cX, W - 1), y0 = max(cY - 1, 0), y1 = min(cY, H - 1);   // the clamped tap indices of lin_setup
  const int oP[4] = {(y0 * W + x0) * C, (y0 * W + x1) * C, (y1 * W + x0) * C, (y1 * W + x1) * C};
  const float* P = (g ? A.F.plane[1][I] : A.F.plane[0][I]) + c16;
  const float* L = (g ? A.F.line[1][I] : A.F.line[0][I]) + c16;
  float acc[NL][4];
#pragma unroll
  for (int i = 0; i < NL; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
  for (uint32_t off = 0; off < n_max; off += 16) {
    const int cnt = off < n_mine ? (int)min(16u, n_mine - off) : 0;   // this group's samples in the batch
    {   // stage 1: lane 16 q + j = sample j of cell q's batch (clamped to a valid entry; unused records are never read as data)
      const uint32_t last = n_mine ? b - 1 : (uint32_t)(A.start[S_][A.K[S_]] ? A.start[S_][A.K[S_]] - 1 : 0);
      const uint32_t pidx = min(a + off + (uint32_t)c16, last);
      const uint32_t m = A.perm[S_][pidx];
      const f32x4 cc = ((const f32x4*)A.coords)[m];
      const float ax[3] = {cc.x, cc.y, cc.z};
      const Lin1 X = lin_setup(ax[AX], W), Y = lin_setup(ax[AY], H), Ln = lin_setup(ax[AL], NLn);
      R.f[0][lane] = m;
      R.f[1][lane] = __float_as_uint(__fmul_rn(Y.w0, X.w0)); R.f[2][lane] = __float_as_uint(__fmul_rn(Y.w0, X.w1));
      R.f[3][lane] = __float_as_uint(__fmul_rn(Y.w1, X.w0)); R.f[4][lane] = __float_as_uint(__fmul_rn(Y.w1, X.w1));
      R.f[5][lane] = (uint32_t)(Ln.i0 * C); R.f[6][lane] = (uint32_t)(Ln.i1 * C);
      R.f[7][lane] = __float_as_uint(Ln.w0); R.f[8][lane] = __float_as_uint(Ln.w1);
      if (DENS) R.f[9][lane] = __float_as_uint(A.d[m]);
    }
    wave_sync();
    int cnt_max = max(cnt, __shfl_xor(cnt, 16, 64));
    cnt_max = max(cnt_max, __shfl_xor(cnt_max, 32, 64));
    for (int t0 = 0; t0 < cnt_max; t0 += U) {   // stage 2: lane = channel of group q's sample t
      float w4[U][4], lw[U][2], di[U][NL], l0[U][NL], l1[U][NL], pt[U][4];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u;
        ok[u] = t < cnt;
        const int tt = 16 * q + (ok[u] ? t : 0);
        const int64_t m = R.f[0][tt];
#pragma unroll
        for (int c = 0; c < 4; ++c) w4[u][c] = __uint_as_float(R.f[1 + c][tt]);
        const int oL0 = (int)R.f[5][tt], oL1 = (int)R.f[6][tt];
        lw[u][0] = __uint_as_float(R.f[7][tt]); lw[u][1] = __uint_as_float(R.f[8][tt]);
#pragma unroll
        for (int i = 0; i < NL; ++i) { l0[u][i] = L[oL0 + 16 * i]; l1[u][i] = L[oL1 + 16 * i]; }
        if (DENS) {
          di[u][0] = __uint_as_float(R.f[9][tt]);
#pragma unroll
          for (int c = 0; c < 4; ++c) pt[u][c] = P[oP[c]];
        } else {
#pragma unroll
          for (int i = 0; i < NL; ++i) di[u][i] = A.d[(m >> 5) * (32 * 3 * C) + (I * NL + i) * 512 + (m & 31) * 16 + c16];   // k_shade_bwd's blocked dv
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float lv[NL], dd[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) lv[i] = l0[u][i] * lw[u][0] + l1[u][i] * lw[u][1];
        if (DENS) {
          // relu per plane (EgoNeRF.py:340,346): the gradient passes where this plane's sum over channels is positive
          float dot = (pt[u][0] * w4[u][0] + pt[u][1] * w4[u][1] + pt[u][2] * w4[u][2] + pt[u][3] * w4[u][3]) * lv[0];
#pragma unroll
          for (int sh = 8; sh >= 1; sh >>= 1) dot += __shfl_xor(dot, sh, 16);
          dd[0] = (ok[u] && dot > 0.f) ? di[u][0] : 0.f;
        } else {
#pragma unroll
          for (int i = 0; i < NL; ++i) dd[i] = ok[u] ? di[u][i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const float gp = dd[i] * lv[i];
          acc[i][0] += gp * w4[u][0]; acc[i][1] += gp * w4[u][1]; acc[i][2] += gp * w4[u][2]; acc[i][3] += gp * w4[u][3];
        }
      }
    }
    wave_sync();   // the next batch overwrites the record
  }
  if (n_mine) {
    float* out = A.cellbuf[S_] + cell_slot(A, S_, k) * 4 * C + c16;
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) out[t * C + 16 * i] = acc[i][t];
  }
}

template <int C, bool DENS>
__global__ __launch_bounds__(256) void k_sorted_plane(SortedArgs A) {
  __shared__ WaveRec rec[4];
  WaveRec& R = rec[threadIdx.x >> 6];
  if (blockIdx.y == 0) sorted_plane<C, DENS, 0>(A, R);
  else if (blockIdx.y == 1) sorted_plane<C, DENS, 1>(A, R);
  else sorted_plane<C, DENS, 2>(A, R);
}

// texel (g, ty, tx) of plane sort_plane(s): corner 0 (y0, x0) of cell (ty + 1, tx + 1), corner 1 (y0, x1) of cell (ty + 1, tx), corner 2
// (y1, x0) of cell (ty, tx + 1), corner 3 (y1, x1) of cell (ty, tx); thread = (texel, 4 channels)
template <int C, int S_>
__device__ __forceinline__ void sorted_plane_final(const SortedArgs& A) {
  constexpr int I = sort_plane(S_), AX = vm_plane_x(I), AY = vm_plane_y(I), Q = C / 4;
  const int W = A.F.res[AX], H = A.F.res[AY];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)2 * H * W * Q) return;
  const int c4 = (int)(idx % Q);
  const int64_t tex = idx / Q;
  const int tx = (int)(tex % W), ty = (int)((tex / W) % H), g = (int)(tex / ((int64_t)W * H));
  const int nmin1 = A.F.res[sort_minor(S_)] + 1, nmaj1 = A.F.res[sort_major(S_)] + 1;
  // a texel's cell exists once per line block (nb = 1 in the round-5 key layout): blocks are added in order, then the four corners
  f32x4 sum[4];
  const uint32_t nbk = A.nb[S_];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int cy = ty + (t < 2 ? 1 : 0), cx = tx + ((t & 1) ? 0 : 1);
    const int cmaj = sort_major(S_) == AX ? cx : cy, cmin = sort_major(S_) == AX ? cy : cx;
    sum[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (uint32_t bk = 0; bk < nbk; ++bk) {
      const uint32_t k = (((uint32_t)g * nbk + bk) * (uint32_t)nmaj1 + (uint32_t)cmaj) * (uint32_t)nmin1 + (uint32_t)cmin;
      if (A.start[S_][k + 1] != A.start[S_][k]) {
        const f32x4 v = *(const f32x4*)(A.cellbuf[S_] + (cell_slot(A, S_, k) * 4 + t) * C + 4 * c4);
        sum[t] = bk ? sum[t] + v : v;
      }
    }
  }
  const f32x4 r = (sum[0] + sum[1]) + (sum[2] + sum[3]);
  *(f32x4*)((g ? A.G.plane[1][I] : A.G.plane[0][I]) + ((int64_t)ty * W + tx) * C + 4 * c4) = r;
}

template <int C>
__global__ void k_sorted_plane_final(SortedArgs A) {
  if (blockIdx.y == 0) sorted_plane_final<C, 0>(A);
  else if (blockIdx.y == 1) sorted_plane_final<C, 1>(A);
  else sorted_plane_final<C, 2>(A);
}

// ---- lines: one wave per 256-sample sub-block of a line cell's range ------------------------------------------------------------------
template <int C, bool DENS, int S_>
__device__ __forceinline__ void sorted_line(const SortedArgs& A, WaveRec& R) {
#pragma clang fp contract(fast)
  constexpr int NL = C / 16, J = sort_line(S_), U = 4;     // line J, its co-plane J
  constexpr int AX = vm_plane_x(J), AY = vm_plane_y(J), AL = vm_line_ax(J);
  static_assert(AL == sort_major(S_), "the line's axis is the major key of its sort");
  const int lane = threadIdx.x & 63, c16 = lane & 15, q = lane >> 4;
  const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t LC = A.LC[S_];
  if (w >= A.suboff[S_][LC]) return;
  // line cell of sub-block w: the last lc with suboff[lc] <= w (empty cells have suboff[lc] == suboff[lc + 1] and are never selected)
  uint32_t lo = 0, hi = LC;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (A.suboff[S_][mid] <= w) lo = mid; else hi = mid;
  }
  const uint32_t lc = lo, j = w - A.suboff[S_][lc];
  const uint32_t nmin1 = (uint32_t)A.F.res[sort_minor(S_)] + 1;
  const uint32_t r0 = A.start[S_][lc * nmin1] + j * SUB, rend = A.start[S_][(lc + 1) * nmin1];
  const uint32_t r1 = r0 + SUB < rend ? r0 + SUB : rend;
  const int g = (int)(lc / ((uint32_t)A.F.res[AL] + 1));
  const int W = A.F.res[AX], H = A.F.res[AY], NLn = A.F.res[AL];
  const float* P = (g ? A.F.plane[1][J] : A.F.plane[0][J]) + c16;
  const float* L = (g ? A.F.line[1][J] : A.F.line[0][J]) + c16;
  float acc[NL][2];
#pragma unroll
  for (int i = 0; i < NL; ++i) acc[i][0] = acc[i][1] = 0.f;
  for (uint32_t base = r0; base < r1; base += 64) {
    const int cnt = (int)min(64u, r1 - base);
    {   // stage 1: lane = sample
      const uint32_t pidx = min(base + (uint32_t)lane, r1 - 1);
      const uint32_t m = A.perm[S_][pidx];
      const f32x4 cc = ((const f32x4*)A.coords)[m];
      const float ax[3] = {cc.x, cc.y, cc.z};
      const Lin1 X = lin_setup(ax[AX], W), Y = lin_setup(ax[AY], H), Ln = lin_setup(ax[AL], NLn);
      R.f[0][lane] = m;
      R.f[1][lane] = __float_as_uint(__fmul_rn(Y.w0, X.w0)); R.f[2][lane] = __float_as_uint(__fmul_rn(Y.w0, X.w1));
      R.f[3][lane] = __float_as_uint(__fmul_rn(Y.w1, X.w0)); R.f[4][lane] = __float_as_uint(__fmul_rn(Y.w1, X.w1));
      R.f[5][lane] = (uint32_t)((Y.i0 * W + X.i0) * C); R.f[6][lane] = (uint32_t)((Y.i0 * W + X.i1) * C);
      R.f[7][lane] = (uint32_t)((Y.i1 * W + X.i0) * C); R.f[8][lane] = (uint32_t)((Y.i1 * W + X.i1) * C);
      R.f[9][lane] = __float_as_uint(Ln.w0); R.f[10][lane] = __float_as_uint(Ln.w1);
      if (DENS) R.f[11][lane] = __float_as_uint(A.d[m]);
    }
    wave_sync();
    // the line's own two taps are the same for the whole range (one line cell): clamped indices of lin_setup for cell c = lc % (n + 1)
    const int cL = (int)(lc % ((uint32_t)NLn + 1));
    const int oL0 = max(cL - 1, 0) * C, oL1 = min(cL, NLn - 1) * C;
    for (int t0 = 0; t0 < cnt; t0 += 4 * U) {   // stage 2: lane = channel
      float w4[U][4], lw[U][2], di[U][NL], pt[U][NL][4];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + 4 * u + q;
        ok[u] = t < cnt;
        const int tt = ok[u] ? t : cnt - 1;
        const int64_t m = R.f[0][tt];
        int o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { w4[u][c] = __uint_as_float(R.f[1 + c][tt]); o[c] = (int)R.f[5 + c][tt]; }
        lw[u][0] = __uint_as_float(R.f[9][tt]); lw[u][1] = __uint_as_float(R.f[10][tt]);
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) pt[u][i][c] = P[o[c] + 16 * i];
        if (DENS) {
          di[u][0] = __uint_as_float(R.f[11][tt]);
        } else {
#pragma unroll
          for (int i = 0; i < NL; ++i) di[u][i] = A.d[(m >> 5) * (32 * 3 * C) + (J * NL + i) * 512 + (m & 31) * 16 + c16];
        }
      }
      float lt0 = 0.f, lt1 = 0.f;
      if (DENS) { lt0 = L[oL0]; lt1 = L[oL1]; }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float pv[NL], dd[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) pv[i] = pt[u][i][0] * w4[u][0] + pt[u][i][1] * w4[u][1] + pt[u][i][2] * w4[u][2] + pt[u][i][3] * w4[u][3];
        if (DENS) {
          float dot = pv[0] * (lt0 * lw[u][0] + lt1 * lw[u][1]);
#pragma unroll
          for (int sh = 8; sh >= 1; sh >>= 1) dot += __shfl_xor(dot, sh, 16);
          dd[0] = (ok[u] && dot > 0.f) ? di[u][0] : 0.f;
        } else {
#pragma unroll
          for (int i = 0; i < NL; ++i) dd[i] = ok[u] ? di[u][i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const float gl = dd[i] * pv[i];
          acc[i][0] += gl * lw[u][0]; acc[i][1] += gl * lw[u][1];
        }
      }
    }
    wave_sync();
  }
  float* out = A.linepart[S_] + (int64_t)w * 2 * C + c16;
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float v = fold_groups(acc[i][t]);
      if (q == 0) out[t * C + 16 * i] = v;
    }
}

template <int C, bool DENS>
__global__ __launch_bounds__(256) void k_sorted_line(SortedArgs A) {
  __shared__ WaveRec rec[4];
  WaveRec& R = rec[threadIdx.x >> 6];
  if (!((A.line_mask >> blockIdx.y) & 1)) return;
  if (blockIdx.y == 0) sorted_line<C, DENS, 0>(A, R);
  else if (blockIdx.y == 1) sorted_line<C, DENS, 1>(A, R);
  else sorted_line<C, DENS, 2>(A, R);
}

// line texel (g, t): west tap (weight w0) of the samples of cell t + 1, east tap (w1) of cell t; partials added in sub-block order
template <int C, int S_>
__device__ __forceinline__ void sorted_line_final(const SortedArgs& A) {
  constexpr int J = sort_line(S_), AL = vm_line_ax(J);
  const int n = A.F.res[AL];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * n * C) return;
  const int ch = idx % C, t = (idx / C) % n, g = idx / (C * n);
  // a cell has 8 - 30 sub-blocks: the partials are FETCHED eight at a time (independent loads) and ADDED one after the other in sub-block
  // order - the same sum, bit for bit, as a one-load-at-a-time loop, without its chain of dependent memory round trips
  auto ordered_sum = [&](uint32_t lc, int off) {
    float s = 0.f;
    const uint32_t w1 = A.suboff[S_][lc + 1];
    const float* base = A.linepart[S_] + off + ch;
    for (uint32_t w = A.suboff[S_][lc]; w < w1; w += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = base[(int64_t)(w + k < w1 ? w + k : w1 - 1) * 2 * C];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (w + k < w1) s += v[k];
    }
    return s;
  };
  const float s0 = ordered_sum((uint32_t)g * (uint32_t)(n + 1) + (uint32_t)t + 1, 0);
  const float s1 = ordered_sum((uint32_t)g * (uint32_t)(n + 1) + (uint32_t)t, C);
  (g ? A.G.line[1][J] : A.G.line[0][J])[t * C + ch] = s0 + s1;
}

template <int C>
__global__ void k_sorted_line_final(SortedArgs A) {
  if (!((A.line_mask >> blockIdx.y) & 1)) return;
  if (blockIdx.y == 0) sorted_line_final<C, 0>(A);
  else if (blockIdx.y == 1) sorted_line_final<C, 1>(A);
  else sorted_line_final<C, 2>(A);
}

// =====================================================================================================================================
// Fused form (round 6): planes + the line of the third axis in one pass over d
// =====================================================================================================================================
// Fixed-point unit of the line sums of one scatter call.  |contribution| = |d| |plane value| |line weight| < 2^(eD + eP + 1) with
// max |d| < 2^eD and max |plane texel| < 2^eP (the interpolated plane value is a convex combination of texels; the + 1 covers its
// rounding).  A texel adds at most M contributions, so with unit 2^k, k = eD + eP + 1 - nbits, nbits = min(50, 62 - ceil(log2 M)), the
// integer sums stay below 2^62.  Conversion: t = fma((double)gl, (double)lw, 1.5 * 2^(52 + k)) rounds the EXACT product to a multiple
// of 2^k (round to nearest even), and bits(t) - bits(1.5 * 2^(52 + k)) is that multiple as a signed integer (|q| < 2^51).
struct FxScale {
  uint32_t pmax_bits[3];   // max |texel| of plane I over both grids (float bits; NaN / Inf compare above every finite value)
  uint32_t dmax_bits;      // max |d|
  uint32_t poison;         // a non-finite input: the line gradients are NaN (as a float sum's would be)
  uint32_t pad_[3];
  double magic[3];         // per plane / line index I
  double lsb[3];
};

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
  return v;
}

// max |x| over n floats (n a multiple of 4, 16-byte aligned) as float bits -> atomicMax(out)
__device__ __forceinline__ void absmax_range(const float* __restrict__ x, int64_t n, uint32_t* out) {
  uint32_t m = 0;
  const int64_t n4 = n >> 2;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {   // four independent loads in flight
    const u4 v0 = ((const u4*)x)[i], v1 = ((const u4*)x)[i + stride], v2 = ((const u4*)x)[i + 2 * stride], v3 = ((const u4*)x)[i + 3 * stride];
    m = max(m, max(max(v0.x & 0x7fffffffu, v0.y & 0x7fffffffu), max(v0.z & 0x7fffffffu, v0.w & 0x7fffffffu)));
    m = max(m, max(max(v1.x & 0x7fffffffu, v1.y & 0x7fffffffu), max(v1.z & 0x7fffffffu, v1.w & 0x7fffffffu)));
    m = max(m, max(max(v2.x & 0x7fffffffu, v2.y & 0x7fffffffu), max(v2.z & 0x7fffffffu, v2.w & 0x7fffffffu)));
    m = max(m, max(max(v3.x & 0x7fffffffu, v3.y & 0x7fffffffu), max(v3.z & 0x7fffffffu, v3.w & 0x7fffffffu)));
  }
  for (; i < n4; i += stride) {
    const u4 v = ((const u4*)x)[i];
    m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
  // one value per workgroup, stored: thousands of atomicMax on ONE address serialise at ~150 ns each (a 0.4 ms kernel for 14 MB)
  __shared__ uint32_t wm[4];
  m = wave_max_u32(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) *out = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
  __syncthreads();
}

constexpr int FX_BLOCKS = 128;   // workgroups per array of k_fx_absmax; their maxima go to FxScale-adjacent slots [7][FX_BLOCKS]
struct AbsmaxArgs {
  uint32_t* part;      // [7][FX_BLOCKS]: per-workgroup maxima (6 planes, d), reduced by k_fx_setup
  const float* plane[2][3];
  int64_t n_plane[3];
  const float* d;      // nullptr: dmax comes from the caller
  int64_t n_d;         // floats of d that are all valid (whole 32-sample tiles of a blocked dv; M for dfeat)
  int32_t tail_rows;   // blocked dv only: valid samples of the last, partial tile (its other rows are uninitialised memory)
  int32_t tail_block;  // floats per tile (32 x 3 C)
  FxScale* fx;
};

// blockIdx.y = 0..5: plane (g, I); 6: d
__global__ __launch_bounds__(256) void k_fx_absmax(AbsmaxArgs A) {
  const int y = blockIdx.y;
  uint32_t* out = A.part + y * FX_BLOCKS + blockIdx.x;
  if (y < 6) {
    absmax_range(A.plane[y / 3][y % 3], A.n_plane[y % 3], out);
  } else if (A.d) {
    absmax_range(A.d, A.n_d, out);
    if (A.tail_rows && blockIdx.x == 0) {   // [plane * 3 + line][sample j][16]: rows j < tail_rows
      uint32_t m = 0;
      for (int i = threadIdx.x; i < A.tail_block; i += blockDim.x)
        if (((i >> 4) & 31) < A.tail_rows) m = max(m, __float_as_uint(A.d[A.n_d + i]) & 0x7fffffffu);
      m = wave_max_u32(m);
      if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);   // (four waves, one workgroup)
    }
  } else if (threadIdx.x == 0) {
    *out = 0u;
  }
}

__global__ void k_fx_setup(FxScale* fx, const uint32_t* part, int64_t M, const float* dmax_ext, const float* basis0 = nullptr,
                           const float* basis1 = nullptr) {
  // RDV (the walk re-derives dv = B^T dfe): dmax_ext is max |dfe|, and |dv| <= max |dfe| x the largest column sum of |B| (27 x 144 per grid)
  float colmax = 0.f;
  if (basis0) {
    for (int c = threadIdx.x; c < 2 * 144; c += 64) {
      const float* B = c < 144 ? basis0 : basis1;
      float sum = 0.f;
      for (int f = 0; f < 27; ++f) sum += fabsf(B[f * 144 + c % 144]);
      colmax = fmaxf(colmax, sum);   // (a NaN weight: fmaxf drops it - the plane / dfe maxima still poison what it touches, and the walk's result is NaN wherever it is used)
    }
    colmax = __uint_as_float(wave_max_u32(__float_as_uint(colmax)));
  }
  // one wave: the workgroups' maxima -> the four scalars
  for (int y = 0; y < 7; ++y) {
    uint32_t m = 0;
    for (int i = threadIdx.x; i < FX_BLOCKS; i += 64) m = max(m, part[y * FX_BLOCKS + i]);
    m = wave_max_u32(m);
    if (threadIdx.x == 0) {
      if (y < 6) fx->pmax_bits[y % 3] = y < 3 ? m : max(fx->pmax_bits[y % 3], m);
      else fx->dmax_bits = m;
    }
  }
  if (threadIdx.x) return;
  uint32_t db = fx->dmax_bits;
  if (dmax_ext) db = __float_as_uint(*dmax_ext) & 0x7fffffffu;
  if (basis0 && db < 0x7f800000u) {
    const float bound = __uint_as_float(db) * colmax * 1.01f;   // (+ 1 %: the walk's fp16 three-term products are within 2^-20 of exact)
    db = bound < 3.0e38f ? __float_as_uint(bound) : 0x7f800000u;
  }
  int lg = 1;
  while (((int64_t)1 << lg) < M && lg < 40) ++lg;
  const int nbits = min(50, 62 - lg);
  uint32_t poison = (db >= 0x7f800000u) ? 1u : 0u;
  for (int i = 0; i < 3; ++i) {
    const uint32_t pb = fx->pmax_bits[i];
    if (pb >= 0x7f800000u) poison = 1u;
    // x < 2^(e - 126) for the biased exponent e of x (normal or subnormal: e = 0 -> x < 2^-126)
    const int eD = (int)(db >> 23) - 126, eP = (int)(pb >> 23) - 126;
    int k = eD + eP + 1 - nbits;
    k = max(k, -1000);                    // (all-zero inputs: any unit will do)
    fx->magic[i] = ldexp(1.5, 52 + k);
    fx->lsb[i] = ldexp(1.0, k);
  }
  fx->poison = poison;
}

// first k in [lo, hi] with a[k] >= target, given a[hi] >= target; 64-ary search by one wave (three dependent round trips for 2^18 keys)
__device__ __forceinline__ uint32_t wave_lower_bound(const uint32_t* __restrict__ a, uint32_t lo, uint32_t hi, uint32_t target) {
  const uint32_t lane = threadIdx.x & 63;
  while (hi > lo) {
    const uint32_t span = hi - lo, step = (span + 63) / 64;
    const uint32_t idx = lo + lane * step;
    const bool ge = idx < hi ? a[idx] >= target : true;
    const unsigned long long m = __ballot(ge);
    if (m == 0ull) { lo = lo + 63 * step + 1; continue; }
    const uint32_t f = (uint32_t)__ffsll((long long)m) - 1u;
    hi = min(hi, lo + f * step);
    if (f > 0) lo = lo + (f - 1) * step + 1;
  }
  return lo;
}

// ---- the walk ----------------------------------------------------------------------------------------------------------------------
// Round 5's plane kernel gave each wave four cells and walked them in lockstep: 16 sample slots per cell and batch whatever the cells
// held (75 % of a batch's time went to its chain of dependent round trips: cell starts -> permutation -> coordinates -> d and taps x 4),
// a workgroup per 16 cells.  Here
//   * the unit of work is a STEP of the step list (k_step_fill): up to 16 samples of one cell, for one 16-lane group; every group of the
//     launch gets the same number of consecutive steps (cut at cell boundaries), so nobody waits for a larger cell or a fuller chunk;
//   * the four groups of a wave walk their own steps independently;
//   * step t + 3's list entry, step t + 2's permutation entries and step t + 1's coordinates are fetched at the top of step t, and the
//     four plane texels of a cell go to LDS by the load itself when its first step begins (global_load_lds: no registers, no
//     compiler-inserted wait), so a step's only exposed round trips are its own 16 / U gathers of d;
//   * where the line table of the plane's third axis fits the LDS (fused_plan), the line gradient is taken in the same step (fixed
//     point, see above); the workgroup's table goes to `part` at the end.
// -DEGO_WALK_PROF: per-phase cycle counts of the walk (s_memtime), summed over all waves into g_walk_prof (tools/sorted_probe.py PROBE_PROF)
#ifdef EGO_WALK_PROF
__device__ unsigned long long g_walk_prof[16];
__device__ unsigned long long g_walk_span[4];   // -, -, slowest wave, -
__device__ uint32_t g_walk_wave[4096 * 4];   // per wave of the last launch: busy ticks, steps, iterations, pair
#define WPROF_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define WPROF_ADD(i, x) prof[i] += (x)
#else
#define WPROF_T(v)
#define WPROF_ADD(i, x)
#endif
template <int NL>
struct WalkLds {           // per wave
  WaveRec rec;
  float pt[NL * 4][64];    // [channel group * 4 + corner][lane]: the four texels of the lane's group's current cell
};

struct FusedArgs {
  SortedArgs A;
  const FxScale* fx;
  unsigned long long* part;   // [workgroup][part_stride]: the workgroups' integer line tables
  uint32_t part_stride;
  int32_t* deal;              // [WALK_MAX_SEG + 1] the launch's deal of workgroups to segments (written by workgroup 0, read by k_fused_line_final)
  int32_t nwg;                // workgroups of the launch
  int8_t do_line[3];          // sort s also takes the gradient of line sort_plane(s)
  const float* dfe;           // BAS (48 channels): ego_shade_backward's feature-slot gradients [M][32]
  float* bpart;               // BAS: per wave [2 slot tiles][3 channel groups][64 lanes][4]: its share of d(basis)
  const float* basis[2];      // RDV: nn.Linear(144 -> 27).weight [27][144] of each grid: the walk re-derives dv = B^T dfe itself
  int32_t dbg;                // experiments (EGO_FUSED_DBG = 16 (n + 1)): segment n alone (timing only: the other segments' gradients are not written)
};

// SEGMENT = (sort, grid, line block): a contiguous range of kc[s] cells of the sort's key order.  Workgroups are dealt to the segments in
// proportion to the COST of their steps (cell_cost; x 6 / 5 where the line rides along: measured); every segment that has steps gets
// at least one.  Wave 0 of every workgroup computes the same deal into LDS; workgroup 0 also leaves it in memory for k_fused_line_final.
struct WalkDealLds {
  int32_t off[WALK_MAX_SEG + 1];   // segment i owns workgroups [off[i], off[i + 1])
  uint32_t w[WALK_MAX_SEG];
};
__device__ __forceinline__ int seg_base(const SortedArgs& A, int s) { return s == 0 ? 0 : s == 1 ? 2 * (int)A.nb[0] : 2 * (int)(A.nb[0] + A.nb[1]); }

__device__ __forceinline__ void walk_deal(const FusedArgs& F, WalkDealLds* D) {
  const SortedArgs& A = F.A;
  const int nseg = 2 * (int)(A.nb[0] + A.nb[1] + A.nb[2]);
  if (threadIdx.x < 64) {
    for (int i = threadIdx.x; i < nseg; i += 64) {
      const int s = i >= seg_base(A, 2) ? 2 : i >= seg_base(A, 1) ? 1 : 0;
      const uint32_t gb = (uint32_t)(i - seg_base(A, s));                 // g * nb + b
      const uint32_t kb = gb * A.kc[s];
      const uint32_t c = A.costsum[s][kb + A.kc[s]] - A.costsum[s][kb];
      D->w[i] = c;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t W = 0;
    int nz = 0;
    for (int i = 0; i < nseg; ++i) {
      const int s = i >= seg_base(A, 2) ? 2 : i >= seg_base(A, 1) ? 1 : 0;
      W += (uint64_t)D->w[i] * (F.do_line[s] ? 6u : 5u);
      nz += D->w[i] != 0;
    }
    int off = 0;
    uint64_t acc = 0;
    for (int i = 0; i < nseg; ++i) {
      D->off[i] = off;
      if (D->w[i]) {
        const int s = i >= seg_base(A, 2) ? 2 : i >= seg_base(A, 1) ? 1 : 0;
        acc += (uint64_t)D->w[i] * (F.do_line[s] ? 6u : 5u);
        --nz;
        int end = (int)(((uint64_t)F.nwg * acc + W / 2) / W);
        end = max(end, off + 1);
        end = min(end, F.nwg - nz);
        off = end;
      }
    }
    D->off[nseg] = off;
    if (blockIdx.x == 0)
      for (int i = 0; i <= nseg; ++i) F.deal[i] = D->off[i];
  }
  __syncthreads();
}

// sum over the 16 lanes of a row on the DPP path (every lane gets the total): quad_perm xor 1, xor 2, row_half_mirror, row_mirror
__device__ __forceinline__ float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xb1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4e, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));   // row_mirror
  return v;
}

// x = hi + lo + O(2^-17 |x|) as two bf16: hi = the top 16 bits of x as they are (truncation - the residual x - hi is exact in fp32 and
// carries what was cut), lo = that residual rounded to its top 16 bits (+ 0x8000).  Four VALU per value (and, sub, add, half a v_perm per
// packed pair and term) where rounding both terms to nearest even takes ten: the walk is bound by VALU issue.
typedef short walk_s4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void walk_split4(const float x[4], walk_s4& hi, walk_s4& lo) {
  uint32_t xb[4], lb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    xb[e] = __float_as_uint(x[e]);
    lb[e] = __float_as_uint(__fsub_rn(x[e], __uint_as_float(xb[e] & 0xffff0000u))) + 0x8000u;   // the residual rounded, not cut: ~17 bits in all, unbiased
  }
  // v_perm_b32: {top half of the second value, top half of the first}
  const u32x2_w ph = {__builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u), __builtin_amdgcn_perm(xb[3], xb[2], 0x07060302u)};
  const u32x2_w pl = {__builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u), __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u)};
  hi = __builtin_bit_cast(walk_s4, ph);
  lo = __builtin_bit_cast(walk_s4, pl);
}

// x (already scaled into fp16's normal range: |x| < 2^13) = hi + lo + O(2^-21 |x|) as two fp16: hi rounded to nearest, lo = the exact fp32
// residual cut to fp16 - the operand form of k_shade_bwd's data-gradient chain (ego_train.inc: split8_rn), here for v_mfma_f32_16x16x16_f16
typedef _Float16 walk_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void walk_split4_f16(const float x[4], walk_h4& hi, walk_h4& lo) {
  typedef float f2v __attribute__((ext_vector_type(2)));
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  uint32_t hw[2], lw[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    const h2v hp = __builtin_convertvector(f2v{a, b}, h2v);
    hw[p] = __builtin_bit_cast(uint32_t, hp);
    lw[p] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__fsub_rn(a, (float)hp[0]), __fsub_rn(b, (float)hp[1])));
  }
  hi = __builtin_bit_cast(walk_h4, u32x2_w{hw[0], hw[1]});
  lo = __builtin_bit_cast(walk_h4, u32x2_w{lw[0], lw[1]});
}
// the power of two that takes amax into [2^12, 2^13) and its inverse (1 for zero, subnormal and non-finite amax), as ego_train.inc's sample_scale
__device__ __forceinline__ float walk_pow2_scale(float amax, float& inv) {
  const int e = (__float_as_int(amax) >> 23) & 0xff;
  const int k = (e == 0 || e > 254) ? 0 : 139 - e;
  const int kc = k > 100 ? 100 : (k < -100 ? -100 : k);
  inv = __int_as_float((127 - kc) << 23);
  return __int_as_float((127 + kc) << 23);
}

template <int C, bool DENS, int S_, int NW, int U, bool BAS = false, bool RDV = false>
__device__ __forceinline__ void sorted_walk(const FusedArgs& F, const int gb, const int wg0, const int wg1, unsigned long long* __restrict__ tab,
                                            WalkLds<C / 16>* wl, const bool do_line) {
#pragma clang fp contract(fast)
  constexpr int NL = C / 16, I = sort_plane(S_);
  constexpr int AX = vm_plane_x(I), AY = vm_plane_y(I), AL = vm_line_ax(I);
  static_assert(AL != sort_major(S_) && AL != sort_minor(S_), "the fused line is the one whose axis is not in the sort key");
  const SortedArgs& A = F.A;
  WalkLds<NL>& W = wl[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63, c16 = lane & 15, q = lane >> 4;
  const int Wd = A.F.res[AX], H = A.F.res[AY], NLn = A.F.res[AL];
  // the segment: grid g, line block blk -> kc cells from kbase; the block's window of the line: texels [lbase, lbase + lsize)
  const int g = gb / (int)A.nb[S_], blk = gb % (int)A.nb[S_];
  const int lbase = max(blk * (int)A.bs[S_] - 1, 0), lsize = min((blk + 1) * (int)A.bs[S_] - 1, NLn - 1) - lbase + 1;
  const int entries = do_line ? lsize * C : 0;
  for (int i = threadIdx.x; i < entries; i += NW * 64) tab[i] = 0ull;
  // a group that never starts a cell (an empty share) multiplies these texels by its zero weights: they must be finite, not what LDS held
#pragma unroll
  for (int i = 0; i < NL * 4; ++i) W.pt[i][lane] = 0.f;
  const uint32_t K = A.K[S_], Kh = A.kc[S_], kbase = (uint32_t)gb * Kh;
  const uint32_t T0 = A.costsum[S_][kbase], T1 = A.costsum[S_][kbase + Kh], Tall = A.stepsum[S_][K];
  // this group's steps: an equal share of the segment's COST, cut at the cell boundaries at or behind the nominal cuts
  const uint32_t nb = (uint32_t)(wg1 - wg0), bl = blockIdx.x - (uint32_t)wg0;
  const uint32_t ng = nb * NW * 4u, gi = (bl * NW + (uint32_t)(threadIdx.x >> 6)) * 4u + (uint32_t)q;
  uint32_t js, je;
  {
    const uint32_t* __restrict__ ss = A.costsum[S_];
    const uint32_t tl = T0 + (uint32_t)((uint64_t)(T1 - T0) * gi / ng), th = T0 + (uint32_t)((uint64_t)(T1 - T0) * (gi + 1) / ng);
    // two 16-ary searches side by side: first cell k in [kbase, kbase + Kh] with costsum[k] >= target (true at kbase + Kh: = T1 >= target)
    uint32_t lo[2] = {kbase, kbase}, hi[2] = {kbase + Kh, kbase + Kh};
    const uint32_t tg[2] = {tl, th};
    while (__ballot(hi[0] > lo[0] || hi[1] > lo[1]) != 0ull) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t span = hi[e] - lo[e], step = (span + 15) / 16;
        const uint32_t idx = lo[e] + (uint32_t)c16 * step;
        const bool ge = (span == 0 || idx >= hi[e]) ? true : ss[idx] >= tg[e];
        const uint32_t m16 = (uint32_t)(__ballot(ge) >> (16 * q)) & 0xffffu;
        if (span) {
          if (m16 == 0) { lo[e] = lo[e] + 15 * step + 1; }
          else {
            const uint32_t f = (uint32_t)__ffs((int)m16) - 1u;
            hi[e] = min(hi[e], lo[e] + f * step);
            if (f > 0) lo[e] = lo[e] + (f - 1) * step + 1;
          }
        }
      }
    }
    js = A.stepsum[S_][lo[0]]; je = A.stepsum[S_][lo[1]];
  }
  // wave-uniform table bases; the lane's channel (c16) goes into the 32-bit element offsets, so a tap address is `scalar base + one VGPR`
  // instead of a 64-bit per-lane pointer sum (the 48-channel walk is VALU-bound: ~470 instructions per iteration, three waves per SIMD)
  const float* __restrict__ Pg = g ? A.F.plane[1][I] : A.F.plane[0][I];
  const float* __restrict__ L = g ? A.F.line[1][I] : A.F.line[0][I];
  const float* __restrict__ Dv = A.d;
  // (BYTE offsets: `base + zext(u32)` is what selects the scalar-base form of global_load; an element index would need a 64-bit shift)
  // (`imm`: a constant added in 64 bits, behind the zero-extension - it lands in the instruction's offset field; added to the 32-bit offset it
  // cost a VALU add per load, because that sum may wrap where the address must not)
  auto at = [](const float* base, uint32_t byte_off, int imm = 0) -> float { return *(const float*)((const char*)base + byte_off + imm); };
  const int nmin1 = A.F.res[sort_minor(S_)] + 1;
  const double magic = F.fx->magic[I];
  const uint32_t magic_hi = (uint32_t)((unsigned long long)__double_as_longlong(magic) >> 32);
  const uint32_t* __restrict__ perm = A.perm[S_];
  const f32x4* __restrict__ coords4 = (const f32x4*)A.coords;
  const uint4* __restrict__ steps = A.steps[S_];
  __syncthreads();   // the table is zero before anyone adds to it
#ifdef EGO_WALK_PROF
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_wave0 = __builtin_amdgcn_s_memtime();
#endif
  if (Tall) {
    const uint32_t jlast = Tall - 1;   // every index below is clamped to a step that exists
    auto entry = [&](uint32_t j) -> uint4 { return steps[min(j, jlast)]; };
    auto pos = [&](const uint4& e) -> uint32_t { return e.x + min((uint32_t)c16, (e.z & 0xffu) - 1u); };
    uint32_t j = js;
    uint4 e0 = entry(j), e1 = entry(j + 1), e2 = entry(j + 2), e3;
    uint32_t m0 = perm[pos(e0)], m1 = perm[pos(e1)], m2;
    f32x4 cc0 = coords4[m0], cc1;
    float d0 = DENS ? A.d[m0] : 0.f, d1 = 0.f;   // dfeat of the step's samples travels with their coordinates
    float acc[NL][4];
#pragma unroll
    for (int i = 0; i < NL; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
    // BAS: this wave's share of d(basis)[slot][channel of plane I] = sum_samples dfe[s][slot] v[s][channel], v = plane value x line
    // value - the walk has both in registers, so the forward need not dump v (576 B per sample) for a weight-gradient pass to read.
    // v_mfma_f32_16x16x16_bf16: K = the 16 samples of one iteration (group q's sample u is k = 4 q + u: what lane 16 q + c holds for
    // both operands), A = dfe^T (row = slot 16 mt + c16), B = v (column = channel 16 i + c16), bf16 hi / lo split, three terms.
    f32x4 bacc[BAS ? 2 : 1][BAS ? NL : 1];
    if constexpr (BAS) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < NL; ++i) bacc[mt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // RDV: dv = B_g^T dfe is a linear map of the 27 feature-slot gradients the walk loads anyway: it is re-derived here, 16 samples x 48
    // channels per iteration, instead of being written by k_shade_bwd (576 B per sample) and gathered back.  v_mfma_f32_16x16x16_f16,
    // M = the iteration's 16 samples (row 4 q' + u'), K = 16 slots (two k-blocks), N = channel: lane 16 q + c supplies A[row c][slots
    // 16 kb + 4 q ..] = four feature-slot gradients of sample (c / 4, c % 4) and B[slots 16 kb + 4 q ..][channel 16 i + c] (constants of
    // the launch, split once), and receives D[rows 4 q ..][channel] = dv of ITS OWN group's four samples.  Arithmetic = k_shade_bwd's for
    // the same product: operands scaled by a power of two per sample (A) / per channel lane (B) into fp16's normal range, fp16 hi + lo,
    // three terms, fp32 accumulation (~2^-21 per product; a bf16 split's 2^-16 showed at 2e-5 of the largest texel gradient).
    walk_h4 bfh[RDV ? 2 : 1][RDV ? NL : 1], bfl[RDV ? 2 : 1][RDV ? NL : 1];
    float inv_b = 1.f;
    if constexpr (RDV) {
      const float* Bg = g ? F.basis[1] : F.basis[0];
      float w[2][NL][4], am = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e, f = 2 * r + kb;   // dfe column 16 kb + r holds feature 2 r + kb (slot r of lane half kb), r <= 13
            w[kb][i][e] = (r <= 13 && f < 27) ? Bg[f * (3 * C) + I * C + 16 * i + c16] : 0.f;
            am = fmaxf(am, fabsf(w[kb][i][e]));
          }
      am = fmaxf(am, __shfl_xor(am, 16, 64));
      am = fmaxf(am, __shfl_xor(am, 32, 64));   // the channel lane's column maximum: the four lanes that supply a column agree on its scale
      const float sb = walk_pow2_scale(am, inv_b);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < NL; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) w[kb][i][e] *= sb;
          walk_split4_f16(w[kb][i], bfh[kb][i], bfl[kb][i]);
        }
    }
    while (__ballot(j < je) != 0ull) {
      WPROF_T(ts0);
      e3 = entry(j + 3);          // step t + 3's entry, t + 2's permutation entries, t + 1's coordinates: in flight under this step
      m2 = perm[pos(e2)];
      cc1 = coords4[m1];
      if (DENS) d1 = A.d[m1];
      const bool act = j < je;
      const int cnt = act ? (int)(e0.z & 0xffu) : 0;
      const uint32_t k = e0.y;
      if (act && (e0.z & 256u)) {   // the cell's first step: its four plane texels -> LDS (clamped tap indices of lin_setup)
        const uint32_t kl = k - kbase;
        const int cmin = (int)(kl % (uint32_t)nmin1), cmaj = (int)(kl / (uint32_t)nmin1);
        const int cX = sort_major(S_) == AX ? cmaj : cmin, cY = sort_major(S_) == AY ? cmaj : cmin;
        const int x0 = max(cX - 1, 0), x1 = min(cX, Wd - 1), y0 = max(cY - 1, 0), y1 = min(cY, H - 1);
        const uint32_t oP[4] = {(uint32_t)((y0 * Wd + x0) * C + c16), (uint32_t)((y0 * Wd + x1) * C + c16), (uint32_t)((y1 * Wd + x0) * C + c16),
                                (uint32_t)((y1 * Wd + x1) * C + c16)};
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Pg + (oP[c] + 16u * i)),
                                             (__attribute__((address_space(3))) void*)&W.pt[i * 4 + c][0], 4, 0, 0);
      }
      {   // stage 1: lane 16 q + i = sample i of group q's step
        const float ax[3] = {cc0.x, cc0.y, cc0.z};
        const Lin1 X = lin_setup(ax[AX], Wd), Y = lin_setup(ax[AY], H), Ln = lin_setup(ax[AL], NLn);
        WaveRec& R = W.rec;
        // a slot past the step's samples repeats the last sample (pos()) with ALL WEIGHTS ZERO: its plane value, line value and every
        // product with them are exact zeros, so stage 2 masks nothing per channel (its d / dv is the repeated sample's: finite)
        const bool live = c16 < cnt;
        R.f[0][lane] = m0;
        R.f[1][lane] = live ? __float_as_uint(__fmul_rn(Y.w0, X.w0)) : 0u; R.f[2][lane] = live ? __float_as_uint(__fmul_rn(Y.w0, X.w1)) : 0u;
        R.f[3][lane] = live ? __float_as_uint(__fmul_rn(Y.w1, X.w0)) : 0u; R.f[4][lane] = live ? __float_as_uint(__fmul_rn(Y.w1, X.w1)) : 0u;
        // per tap: the byte offset of its texel row in the line table, and of its row in the block's LDS window (a tap outside the
        // window has weight 0) - once per sample here instead of once per channel lane in stage 2
        R.f[5][lane] = (uint32_t)(Ln.i0 * C) * 4u; R.f[6][lane] = (uint32_t)(Ln.i1 * C) * 4u;
        R.f[7][lane] = live ? __float_as_uint(Ln.w0) : 0u; R.f[8][lane] = live ? __float_as_uint(Ln.w1) : 0u;
        if (DENS) R.f[9][lane] = live ? __float_as_uint(d0) : 0u;
        R.f[10][lane] = (uint32_t)(min(max(Ln.i0 - lbase, 0), lsize - 1) * C) * 8u;
        R.f[11][lane] = (uint32_t)(min(max(Ln.i1 - lbase, 0), lsize - 1) * C) * 8u;
      }
      wave_sync();
      int cnt_max = max(cnt, __shfl_xor(cnt, 16, 64));
      cnt_max = __builtin_amdgcn_readfirstlane(max(cnt_max, __shfl_xor(cnt_max, 32, 64)));
      float pt[NL][4];
      WPROF_T(ts1); WPROF_ADD(0, ts1 - ts0); WPROF_ADD(5, 1);
      for (int t0 = 0; t0 < cnt_max; t0 += U) {   // stage 2: lane = channel of group q's sample t
        WPROF_T(ti0); WPROF_ADD(6, 1);
        const WaveRec& R = W.rec;
        float w4[U][4], lw[U][2], di[U][NL], l0[U][NL], l1[U][NL];
        float fa[BAS ? 2 : 1][4], vv[BAS ? NL : 1][4];
        static_assert(!BAS || U == 4, "one MFMA per iteration: 4 groups x 4 samples = K 16");
        uint32_t tL0[U], tL1[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int t = t0 + u;   // (<= 15: t0 is a multiple of U = 4 below cnt_max <= 16)
          ok[u] = t < cnt;
          const int tt = 16 * q + t;
          const uint32_t m = R.f[0][tt];
#pragma unroll
          for (int c = 0; c < 4; ++c) w4[u][c] = __uint_as_float(R.f[1 + c][tt]);
          lw[u][0] = __uint_as_float(R.f[7][tt]); lw[u][1] = __uint_as_float(R.f[8][tt]);
          tL0[u] = R.f[10][tt]; tL1[u] = R.f[11][tt];
          const uint32_t oL0 = R.f[5][tt] + (uint32_t)c16 * 4u, oL1 = R.f[6][tt] + (uint32_t)c16 * 4u;
#pragma unroll
          for (int i = 0; i < NL; ++i) { l0[u][i] = at(L, oL0, 64 * i); l1[u][i] = at(L, oL1, 64 * i); }
          if (DENS) {
            di[u][0] = __uint_as_float(R.f[9][tt]);
          } else {
            // k_shade_bwd's blocked dv; 32-bit element offsets (byte offsets below 2^32: the launcher takes this path only below 2^30 elements)
            if constexpr (!RDV) {
              const uint32_t od = ((m >> 5) * (uint32_t)(32 * 3 * C) + (uint32_t)(I * NL) * 512u + (m & 31u) * 16u + (uint32_t)c16) * 4u;
#pragma unroll
              for (int i = 0; i < NL; ++i) di[u][i] = at(Dv, od, 2048 * i);
            }
            if constexpr (BAS) {
              const uint32_t of = (m * 32u + (uint32_t)c16) * 4u;
              fa[0][u] = at(F.dfe, of); fa[1][u] = at(F.dfe, of, 64);
            }
          }
        }
        f32x4 ra[RDV ? 2 : 1];
        bool rok = false;
        if constexpr (RDV) {   // A operand: row c16 = sample (group c16 / 4, slot t0 + c16 % 4)
          const int gq = c16 >> 2, tr = t0 + (c16 & 3);
          rok = true;   // (a slot past the step's samples: the repeated sample's row, multiplied by zero weights below)
          const uint32_t mr = R.f[0][16 * gq + tr];
          const f32x4* pa = (const f32x4*)((const char*)F.dfe + (mr * 32u + 4u * (uint32_t)q) * 4u);
          ra[0] = pa[0]; ra[1] = pa[4];   // slots 4 q .. 4 q + 3 of k-block 0 (dfe columns 0 .. 15) and of k-block 1 (columns 16 .. 31)
        }
        WPROF_T(ti1); WPROF_ADD(1, ti1 - ti0);
#ifdef EGO_WALK_PROF
        __builtin_amdgcn_s_waitcnt(0x0f70);
        WPROF_T(ti2); WPROF_ADD(2, ti2 - ti1);
#endif
        if (t0 == 0) {
          // the texels of a cell that started in this step were sent to LDS ahead of the loads above; vector memory returns in order
          // and vmcnt counts the LDS-bound loads too, so after this wait they have landed (the compiler knows of no dependency)
          __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
          asm volatile("" ::: "memory");
#pragma unroll
          for (int i = 0; i < NL; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) pt[i][c] = W.pt[i * 4 + c][lane];
        }
        if constexpr (RDV) {
          float a8[2][4] = {{ra[0].x, ra[0].y, ra[0].z, ra[0].w}, {ra[1].x, ra[1].y, ra[1].z, ra[1].w}};
          float am = 0.f;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a8[kb][e] = rok ? a8[kb][e] : 0.f; am = fmaxf(am, fabsf(a8[kb][e])); }
          am = fmaxf(am, __shfl_xor(am, 16, 64));
          am = fmaxf(am, __shfl_xor(am, 32, 64));   // the sample's (row c16's) largest slot gradient
          float inv_a;
          const float sa = walk_pow2_scale(am, inv_a);
          walk_h4 ah[2], al[2];
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a8[kb][e] *= sa;
            walk_split4_f16(a8[kb], ah[kb], al[kb]);
          }
          float inv_d[4];   // rows 4 q + r belong to this group's samples: their scales sit in the lanes that supplied those rows
#pragma unroll
          for (int r = 0; r < 4; ++r) inv_d[r] = __shfl(inv_a, 4 * q + r, 64) * inv_b;
#pragma unroll
          for (int i = 0; i < NL; ++i) {
            f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
              d4 = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[kb], bfh[kb][i], d4, 0, 0, 0);
              d4 = __builtin_amdgcn_mfma_f32_16x16x16f16(al[kb], bfh[kb][i], d4, 0, 0, 0);
              d4 = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[kb], bfl[kb][i], d4, 0, 0, 0);
            }
            di[0][i] = d4.x * inv_d[0]; di[1][i] = d4.y * inv_d[1]; di[2][i] = d4.z * inv_d[2]; di[3][i] = d4.w * inv_d[3];
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float lv[NL], pv[NL], dd[NL];
#pragma unroll
          for (int i = 0; i < NL; ++i) {
            lv[i] = l0[u][i] * lw[u][0] + l1[u][i] * lw[u][1];
            pv[i] = pt[i][0] * w4[u][0] + pt[i][1] * w4[u][1] + pt[i][2] * w4[u][2] + pt[i][3] * w4[u][3];
          }
          if (DENS) {
            // relu per plane (EgoNeRF.py:340,346): the gradient passes where this plane's sum over channels is positive
            const float dot = row_sum16(pv[0] * lv[0]);
            dd[0] = dot > 0.f ? di[u][0] : 0.f;
          } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) dd[i] = di[u][i];
          }
#pragma unroll
          for (int i = 0; i < NL; ++i) {
            const float gp = dd[i] * lv[i];
            acc[i][0] += gp * w4[u][0]; acc[i][1] += gp * w4[u][1]; acc[i][2] += gp * w4[u][2]; acc[i][3] += gp * w4[u][3];
            if constexpr (BAS) vv[i][u] = __fmul_rn(pv[i], lv[i]);
          }
          if (do_line) {
            const double lw0 = (double)lw[u][0], lw1 = (double)lw[u][1];
            unsigned long long* t0p = (unsigned long long*)((char*)tab + tL0[u]) + c16;
            unsigned long long* t1p = (unsigned long long*)((char*)tab + tL1[u]) + c16;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
              const double gl = (double)__fmul_rn(dd[i], pv[i]);
              // bits(x + magic) - bits(magic); magic = 1.5 x 2^(52 + k) has no bit in its low dword: one 32-bit subtraction
              const unsigned long long b0 = (unsigned long long)__double_as_longlong(__fma_rn(gl, lw0, magic));
              const unsigned long long b1 = (unsigned long long)__double_as_longlong(__fma_rn(gl, lw1, magic));
              uint32_t h0 = (uint32_t)(b0 >> 32) - magic_hi, h1 = (uint32_t)(b1 >> 32) - magic_hi;
              asm("" : "+v"(h0), "+v"(h1));   // (keeps the compiler from widening this back into a 64-bit subtraction with a zero low half)
              const unsigned long long q0 = ((unsigned long long)h0 << 32) | (uint32_t)b0, q1 = ((unsigned long long)h1 << 32) | (uint32_t)b1;
              if (ok[u]) {
                __hip_atomic_fetch_add(t0p + 16 * i, (unsigned long long)q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(t1p + 16 * i, (unsigned long long)q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          }
        }
        if constexpr (BAS) {
          walk_s4 ah[2], al[2], bh[NL], bl[NL];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) walk_split4(fa[mt], ah[mt], al[mt]);
#pragma unroll
          for (int i = 0; i < NL; ++i) walk_split4(vv[i], bh[i], bl[i]);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < NL; ++i) {
              bacc[mt][i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[mt], bh[i], bacc[mt][i], 0, 0, 0);
              bacc[mt][i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al[mt], bh[i], bacc[mt][i], 0, 0, 0);
              bacc[mt][i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[mt], bl[i], bacc[mt][i], 0, 0, 0);
            }
        }
      }
      WPROF_T(ts2); WPROF_ADD(3, ts2 - ts1);
      wave_sync();   // the next step overwrites the record (and, for a group that starts a cell, its texels)
      if (act && (e0.z & 512u)) {    // the cell's last step: its four corner sums - one writer per cell, its samples added in ascending order
        const int64_t slot = A.dense_cells ? (int64_t)k : (int64_t)A.start[S_][k];
        float* out = A.cellbuf[S_] + slot * 4 * C + c16;
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
          for (int t = 0; t < 4; ++t) { out[t * C + 16 * i] = acc[i][t]; acc[i][t] = 0.f; }
      }
      ++j;
      e0 = e1; e1 = e2; e2 = e3;
      m0 = m1; cc0 = cc1; m1 = m2; d0 = d1;
      WPROF_T(ts3); WPROF_ADD(4, ts3 - ts2);
    }
    if constexpr (BAS) {   // lane 16 q + c16 holds D[slot 16 mt + 4 q + r][channel 16 i + c16], r = 0 .. 3
      f32x4* o = (f32x4*)F.bpart + ((int64_t)blockIdx.x * NW + (threadIdx.x >> 6)) * (2 * NL * 64) + lane;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < NL; ++i) o[(mt * NL + i) * 64] = bacc[mt][i];
    }
  } else if (BAS) {
    f32x4* o = (f32x4*)F.bpart + ((int64_t)blockIdx.x * NW + (threadIdx.x >> 6)) * (2 * NL * 64) + lane;
#pragma unroll
    for (int e = 0; e < 2 * NL; ++e) o[e * 64] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#ifdef EGO_WALK_PROF
  if (lane == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_walk_prof[i + (DENS ? 0 : 8)], prof[i]);
    const unsigned long long busy = __builtin_amdgcn_s_memtime() - t_wave0;
    atomicMax(&g_walk_span[2], busy);
    const uint32_t wid = blockIdx.x * NW + (threadIdx.x >> 6);
    if (wid < 4096) { g_walk_wave[4 * wid] = (uint32_t)busy; g_walk_wave[4 * wid + 1] = (uint32_t)prof[5]; g_walk_wave[4 * wid + 2] = (uint32_t)prof[6]; g_walk_wave[4 * wid + 3] = 2 * S_ + g; }

  }
#endif
  __syncthreads();
  if (do_line) {
    unsigned long long* out = F.part + (int64_t)blockIdx.x * F.part_stride;
    for (int i = threadIdx.x; i < entries; i += NW * 64) out[i] = tab[i];
  }
}

// NW waves per workgroup, U samples in flight per 16-lane group
template <int C, bool DENS, int NW, int U, bool BAS = false, bool RDV = false>
__global__ __launch_bounds__(NW * 64) void k_sorted_walk(FusedArgs F) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
  typedef WalkLds<C / 16> WL;
  WL* wl = (WL*)fused_lds;
  WalkDealLds* D = (WalkDealLds*)(fused_lds + NW * sizeof(WL));
  unsigned long long* tab2 = (unsigned long long*)(fused_lds + NW * sizeof(WL) + 1024);
  static_assert(sizeof(WalkDealLds) <= 1024, "deal table");
  walk_deal(F, D);
  const SortedArgs& A = F.A;
  const int nseg = 2 * (int)(A.nb[0] + A.nb[1] + A.nb[2]);
  if ((int)blockIdx.x >= D->off[nseg]) return;
  int seg = 0;
  for (int i = 1; i < nseg; ++i) seg += (int)blockIdx.x >= D->off[i] ? 1 : 0;   // (empty segments own no workgroup: off[i] == off[i + 1])
  // read from LDS, so the compiler takes them for per-lane values: as scalars the table bases stay in SGPRs (global_load with a scalar base
  // and an immediate offset instead of a 64-bit VALU address sum per load) and the loops over steps / iterations branch on scalars
  seg = __builtin_amdgcn_readfirstlane(seg);
  const int wg0 = __builtin_amdgcn_readfirstlane(D->off[seg]), wg1 = __builtin_amdgcn_readfirstlane(D->off[seg + 1]);
  const int s = seg >= seg_base(A, 2) ? 2 : seg >= seg_base(A, 1) ? 1 : 0;
  const int gb = seg - seg_base(A, s);
  if (F.dbg >= 16 && (F.dbg >> 4) - 1 != seg) return;   // experiments: one segment alone (timing only)
  const bool do_line = F.do_line[s] != 0;
  if (s == 0) sorted_walk<C, DENS, 0, NW, U, BAS, RDV>(F, gb, wg0, wg1, tab2, wl, do_line);
  else if (s == 1) sorted_walk<C, DENS, 1, NW, U, BAS, RDV>(F, gb, wg0, wg1, tab2, wl, do_line);
  else sorted_walk<C, DENS, 2, NW, U, BAS, RDV>(F, gb, wg0, wg1, tab2, wl, do_line);
}

// d(basis) [2 grids][32 slots][144 = plane x 48 + channel] = the waves' partial products (k_sorted_walk<.., BAS>) added in workgroup / wave
// order: bit-reproducible.  Plane I's 48 columns come from the sort that walks plane I; slot = ego_shade_backward's dfe column.
template <int NW>
__global__ __launch_bounds__(256) void k_basis_reduce(FusedArgs F, float* __restrict__ G, int ldg) {
  // one wave per output element: lane L adds the partials L, L + 64, ... of the element's (workgroup, wave) list in that order, the 64
  // lane sums are added as a fixed tree - the same bits every run (a thread per element walked ~700 partials alone: 60 us)
  const SortedArgs& A = F.A;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (idx >= 2 * 32 * 144) return;
  const int col = idx % 144, slot = (idx / 144) % 32, g = idx / (144 * 32);
  const int I = col / 48, i = (col % 48) / 16, c16 = col % 16, mt = slot / 16, q = (slot % 16) / 4, r = slot % 4;
  const int s = I == 1 ? 0 : I == 0 ? 1 : 2;   // sort_plane(s) == I
  const int nbk = (int)A.nb[s];
  const int64_t elem = (int64_t)(mt * 3 + i) * 256 + (16 * q + c16) * 4 + r;
  float sum = 0.f;
  for (int blk = 0; blk < nbk; ++blk) {
    const int seg = seg_base(A, s) + g * nbk + blk;
    const int b0 = F.deal[seg], n = (F.deal[seg + 1] - b0) * NW;
    for (int t = lane; t < n; t += 64) sum += F.bpart[((int64_t)b0 * NW + t) * (6 * 256) + elem];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
  if (lane == 0) G[(32 * g + slot) * ldg + col] = sum;
}

// line texel (g, t, ch) of line I = sort_plane(s): the integer sums of the workgroups that served the (one or two) blocks whose window
// holds the texel, converted once
template <int C>
__global__ void k_fused_line_final(FusedArgs F) {
  const SortedArgs& A = F.A;
  const int s = blockIdx.y, I = s == 0 ? 1 : s == 1 ? 0 : 2, AL = 2 - I;
  if (!F.do_line[s]) return;
  const int n = A.F.res[AL];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * n * C) return;
  const int ch = idx % C, t = (idx / C) % n, g = idx / (C * n);
  const int nbk = (int)A.nb[s], bsz = (int)A.bs[s];
  long long sum = 0;
  for (int blk = 0; blk < nbk; ++blk) {
    const int lbase = max(blk * bsz - 1, 0), lsize = min((blk + 1) * bsz - 1, n - 1) - lbase + 1;
    if (t < lbase || t >= lbase + lsize) continue;
    const int seg = seg_base(A, s) + g * nbk + blk;
    const int b0 = F.deal[seg], b1 = F.deal[seg + 1];
    const unsigned long long* p = F.part + (t - lbase) * C + ch;
    for (int b = b0; b < b1; ++b) sum += (long long)p[(int64_t)b * F.part_stride];
  }
  const float v = F.fx->poison ? __uint_as_float(0x7fc00000u) : (float)((double)sum * F.fx->lsb[I]);
  (g ? A.G.line[1][I] : A.G.line[0][I])[t * C + ch] = v;
}

int fill_args(const ego_vm_field& f, const ego_vm_grad* grad, const float* coords, const float* d, const SortGeom& G, void* ws, SortedArgs* a, const char* who) {
  if (!grad) return ego_fail(EGO_E_BADARG, "%s: null gradient struct", who);
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i) {
      if (!f.plane[g][i] || !f.line[g][i] || !grad->plane[g][i] || !grad->line[g][i]) return ego_fail(EGO_E_BADARG, "%s: null table", who);
      a->G.plane[g][i] = grad->plane[g][i];
      a->G.line[g][i] = grad->line[g][i];
    }
  a->F = make_field(f);
  a->coords = coords; a->d = d;
  char* base = (char*)ws;
  for (int s = 0; s < 3; ++s) {
    a->perm[s] = (const uint32_t*)(base + G.perm[s]);
    a->start[s] = (const uint32_t*)(base + G.start[s]);
    a->suboff[s] = (const uint32_t*)(base + G.suboff[s]);
    a->stepsum[s] = (const uint32_t*)(base + G.stepsum[s]);
    a->costsum[s] = (const uint32_t*)(base + G.costsum[s]);
    a->steps[s] = (const uint4*)(base + G.steps[s]);
    a->cellbuf[s] = (float*)(base + G.cellbuf[s]);
    a->linepart[s] = (float*)(base + G.linepart[s]);
    a->K[s] = G.K[s]; a->LC[s] = G.LC[s];
  }
  a->dense_cells = G.dense_cells ? 1 : 0;
  for (int s = 0; s < 3; ++s) { a->nb[s] = G.nb[s]; a->bs[s] = G.bs[s]; a->kc[s] = G.kc[s]; }
  a->line_mask = 7;
  return EGO_OK;
}

template <int C, bool DENS>
int launch_sorted(const SortedArgs& a, const SortGeom& G, hipStream_t st) {
  uint32_t kmax = 0;
  int64_t texmax = 0, linemax = 0;
  for (int s = 0; s < 3; ++s) {
    kmax = G.K[s] > kmax ? G.K[s] : kmax;
    const int I = sort_plane(s);
    const int64_t tex = (int64_t)2 * G.res[I == 2 ? 1 : 0] * G.res[I == 0 ? 1 : 2] * (C / 4);   // plane I: x axis, y axis (vm_plane_x / _y)
    texmax = tex > texmax ? tex : texmax;
    const int64_t ln = (int64_t)2 * G.res[sort_major(s)] * C;
    linemax = ln > linemax ? ln : linemax;
  }
  k_sorted_plane<C, DENS><<<dim3((kmax + 15) / 16, 3), 256, 0, st>>>(a);   // 4 cells per wave, 4 waves per workgroup
  if (int e = ego_launch_status("k_sorted_plane")) return e;
  k_sorted_line<C, DENS><<<dim3((G.nsub_max + 3) / 4, 3), 256, 0, st>>>(a);
  if (int e = ego_launch_status("k_sorted_line")) return e;
  k_sorted_plane_final<C><<<dim3((unsigned)((texmax + 255) / 256), 3), 256, 0, st>>>(a);
  if (int e = ego_launch_status("k_sorted_plane_final")) return e;
  k_sorted_line_final<C><<<dim3((unsigned)((linemax + 255) / 256), 3), 256, 0, st>>>(a);
  return ego_launch_status("k_sorted_line_final");
}

int device_cus() {
  static std::atomic<int> cus_of[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  int cus = cus_of[dev].load(std::memory_order_relaxed);
  if (!cus) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cus_of[dev].store(cus, std::memory_order_relaxed);
  }
  return cus;
}

template <int C, bool DENS, int NW, int U, bool BAS = false, bool RDV = false>
int launch_walk_nw(const FusedArgs& F, int wg_total, int lds_bytes, hipStream_t st) {
  static std::atomic<int> attr_set{0};
  if (attr_set.load(std::memory_order_relaxed) < lds_bytes) {
    if (const hipError_t e = hipFuncSetAttribute((const void*)k_sorted_walk<C, DENS, NW, U, BAS, RDV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes))
      return ego_fail((int)e, "k_sorted_walk: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
    attr_set.store(lds_bytes, std::memory_order_relaxed);
  }
  k_sorted_walk<C, DENS, NW, U, BAS, RDV><<<wg_total, NW * 64, lds_bytes, st>>>(F);
  return ego_launch_status("k_sorted_walk");
}

template <int C, bool DENS>
int launch_walk(SortedArgs a, const SortGeom& G, char* base, int64_t M, const float* dmax_ext, hipStream_t st, const float* dfe = nullptr,
                float* gbasis = nullptr, int ldg = 0, const float* const* basis = nullptr) {
  static_assert(sizeof(WalkLds<C / 16>) == 12 * 64 * 4 + (C / 16) * 4 * 64 * 4, "walk_wave_bytes() mirrors WalkLds");
  const FusedPlan P = fused_plan(G, C);
  FxScale* fx = (FxScale*)(base + G.fx);
  if (const hipError_t e = hipMemsetAsync(fx, 0, 256, st)) return ego_fail((int)e, "scatter_sorted: hipMemsetAsync failed: %s", hipGetErrorString(e));
  const bool any_line = P.do_line[0] || P.do_line[1] || P.do_line[2];
  if (any_line) {
    AbsmaxArgs ab{};
    for (int g = 0; g < 2; ++g)
      for (int i = 0; i < 3; ++i) ab.plane[g][i] = a.F.plane[g][i];
    for (int i = 0; i < 3; ++i) ab.n_plane[i] = (int64_t)G.res[i == 2 ? 1 : 0] * G.res[i == 0 ? 1 : 2] * C;
    ab.fx = fx;
    ab.part = (uint32_t*)(base + G.fx + 256);
    if (!dmax_ext) {
      ab.d = a.d;
      if (DENS) { ab.n_d = M; }
      else { ab.n_d = (M / 32) * (32 * 3 * C); ab.tail_rows = (int32_t)(M % 32); ab.tail_block = 32 * 3 * C; }
    }
    k_fx_absmax<<<dim3(FX_BLOCKS, 7), 256, 0, st>>>(ab);
    if (int e = ego_launch_status("k_fx_absmax")) return e;
    k_fx_setup<<<1, 64, 0, st>>>(fx, ab.part, M, dmax_ext, basis ? basis[0] : nullptr, basis ? basis[1] : nullptr);
    if (int e = ego_launch_status("k_fx_setup")) return e;
  }
  FusedArgs F{};
  F.fx = fx;
  F.part = (unsigned long long*)(base + G.fpart);
  F.part_stride = (uint32_t)G.fpart_stride;
  F.deal = (int32_t*)(base + G.fx + 256 + 4 * 7 * 128);
  if (const char* e = getenv("EGO_FUSED_DBG")) F.dbg = atoi(e);
  // one workgroup per CU; the kernel deals them to the (sort, grid) pairs in proportion to their steps (walk_deal)
  int wg_total = device_cus();
  if (wg_total > FUSED_MAX_WG) wg_total = FUSED_MAX_WG;
  const int nseg = 2 * (int)(G.nb[0] + G.nb[1] + G.nb[2]);
  if (wg_total < nseg) wg_total = nseg;   // (<= WALK_MAX_SEG = 96 <= FUSED_MAX_WG)
  F.nwg = wg_total;
  const int off = wg_total;
  for (int s = 0; s < 3; ++s) F.do_line[s] = P.do_line[s] ? 1 : 0;
  // the lines the walk does not take keep the two-pass form
  int mask = 0;
  for (int s = 0; s < 3; ++s)
    if (!P.do_line[s]) { const int Ln = sort_plane(s); mask |= 1 << (Ln == 0 ? 0 : Ln == 1 ? 2 : 1); }   // the sort whose major key is line Ln's axis
  a.line_mask = mask;
  F.A = a;
  const bool bas = !DENS && dfe && gbasis;
  if (bas) {   // 8 waves per workgroup: the basis product's accumulators and operands take the walk past 170 VGPRs (the 48-channel walk runs as fast at two waves per SIMD as at three)
    F.dfe = dfe;
    F.bpart = (float*)(base + G.bpart);
    if constexpr (!DENS) {
      if (basis) {
        F.basis[0] = basis[0]; F.basis[1] = basis[1];
        if (int e = launch_walk_nw<C, DENS, WALK_NW_BAS, 4, true, true>(F, off, P.lds_bytes, st)) return e;
      } else if (int e = launch_walk_nw<C, DENS, WALK_NW_BAS, 4, true>(F, off, P.lds_bytes, st)) return e;
      k_basis_reduce<WALK_NW_BAS><<<(2 * 32 * 144 + 3) / 4, 256, 0, st>>>(F, gbasis, ldg);
      if (int e = ego_launch_status("k_basis_reduce")) return e;
    }
  } else {
    constexpr int NW = C > 16 ? WALK_NW_APP : WALK_NW_DENS;
    if (int e = launch_walk_nw<C, DENS, NW, 4>(F, off, P.lds_bytes, st)) return e;
  }
  if (mask) {
    k_sorted_line<C, DENS><<<dim3((G.nsub_max + 3) / 4, 3), 256, 0, st>>>(a);
    if (int e = ego_launch_status("k_sorted_line")) return e;
  }
  int64_t texmax = 0, linemax = 0;
  for (int s = 0; s < 3; ++s) {
    const int I = sort_plane(s);
    const int64_t tex = (int64_t)2 * G.res[I == 2 ? 1 : 0] * G.res[I == 0 ? 1 : 2] * (C / 4);
    texmax = tex > texmax ? tex : texmax;
    const int64_t ln = (int64_t)2 * G.res[s] * C;
    linemax = ln > linemax ? ln : linemax;
  }
  k_sorted_plane_final<C><<<dim3((unsigned)((texmax + 255) / 256), 3), 256, 0, st>>>(a);
  if (int e = ego_launch_status("k_sorted_plane_final")) return e;
  if (mask) {
    k_sorted_line_final<C><<<dim3((unsigned)((linemax + 255) / 256), 3), 256, 0, st>>>(a);
    if (int e = ego_launch_status("k_sorted_line_final")) return e;
  }
  if (any_line) {
    k_fused_line_final<C><<<dim3((unsigned)((linemax + 255) / 256), 3), 256, 0, st>>>(F);
    if (int e = ego_launch_status("k_fused_line_final")) return e;
  }
  return EGO_OK;
}

// an empty batch: no sample, so every gradient texel is 0 - and "every texel is written" holds for it too (the tables
// come from torch.empty in the sorted mode)
int zero_tables(const ego_vm_field& f, const ego_vm_grad* grad, int C, hipStream_t st, const char* who) {
  if (!grad) return ego_fail(EGO_E_BADARG, "%s: null gradient struct", who);
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i) {
      if (!grad->plane[g][i] || !grad->line[g][i]) return ego_fail(EGO_E_BADARG, "%s: null table", who);
      const size_t np = (size_t)f.res[i == 2 ? 1 : 0] * f.res[i == 0 ? 1 : 2] * C * 4, nl = (size_t)f.res[2 - i] * C * 4;
      if (const hipError_t e = hipMemsetAsync(grad->plane[g][i], 0, np, st)) return ego_fail((int)e, "%s: hipMemsetAsync failed: %s", who, hipGetErrorString(e));
      if (const hipError_t e = hipMemsetAsync(grad->line[g][i], 0, nl, st)) return ego_fail((int)e, "%s: hipMemsetAsync failed: %s", who, hipGetErrorString(e));
    }
  return EGO_OK;
}

int check_sizes(const ego_scene* sc, int64_t N, int32_t S, const char* who) {
  if (!sc) return ego_fail(EGO_E_BADARG, "%s: null scene", who);
  if (!(N >= 0 && S >= 1 && N * (int64_t)S < (1ll << 31))) return ego_fail(EGO_E_BADARG, "%s: bad size (N * S must be below 2^31)", who);
  for (int a = 0; a < 3; ++a)
    if (sc->density.res[a] < 2 || sc->density.res[a] > 4096) return ego_fail(EGO_E_BADARG, "%s: table resolution out of range [2, 4096]", who);
  return EGO_OK;
}

}  // namespace

#ifdef EGO_WALK_PROF
extern "C" int ego_debug_walk_prof(unsigned long long* out16) {   // experiment builds only: read and clear the counters
  unsigned long long z[16] = {};
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_walk_prof), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out16 + 16, HIP_SYMBOL(g_walk_span), 32) != hipSuccess) return -1;
  unsigned long long sp[4] = {0ull, 0ull, 0ull, 0ull};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_walk_span), sp, 32) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_walk_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

#ifdef EGO_WALK_PROF
extern "C" int ego_debug_walk_waves(uint32_t* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_wave), 4096 * 16) == hipSuccess ? 0 : -1; }
#endif

extern "C" {

int64_t ego_scatter_sorted_workspace_bytes(const ego_scene* sc, int64_t N, int32_t S) {
  if (check_sizes(sc, N, S, "scatter_sorted_workspace_bytes")) return -1;
  return make_geom(sc->density.res, N * (int64_t)S > 0 ? N * (int64_t)S : 1).total;
}

int ego_scatter_sort(const ego_scene* sc, const float* coords, int64_t N, int32_t S, void* workspace, int64_t workspace_bytes, void* stream) {
  EGO_TRACE("ego_scatter_sort");
  if (int e = check_sizes(sc, N, S, "scatter_sort")) return e;
  if (N == 0) return EGO_OK;
  EGO_REQUIRE(coords && workspace && ((uintptr_t)workspace & 255) == 0, "scatter_sort: null argument or workspace not 256-byte aligned");
  for (int a = 0; a < 3; ++a)
    EGO_REQUIRE(sc->app.res[a] == sc->density.res[a], "scatter_sort: the density and appearance fields must share one resolution (they do: EgoNeRF.py:102-122)");
  const int64_t M = N * (int64_t)S;
  const SortGeom G = make_geom(sc->density.res, M);
  if (workspace_bytes < G.total) return ego_fail(EGO_E_BADARG, "scatter_sort: workspace too small (%lld < %lld bytes)", (long long)workspace_bytes, (long long)G.total);
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace;
  uint32_t* kin[3] = {(uint32_t*)(base + G.keys_in[0]), (uint32_t*)(base + G.keys_in[1]), (uint32_t*)(base + G.keys_in[2])};
  KeyArgs ka{};
  for (int s = 0; s < 3; ++s) { ka.K[s] = G.K[s]; ka.nb[s] = G.nb[s]; ka.bs[s] = G.bs[s]; }
  k_sort_keys<<<(unsigned)((M + 255) / 256), 256, 0, st>>>(coords, M, G.res[0], G.res[1], G.res[2], ka, kin[0], kin[1], kin[2]);
  if (int e = ego_launch_status("k_sort_keys")) return e;
  // LSD passes ping-pong between (k1, v1) and (k2, perm); the last pass lands in (k2, perm)
  for (int p = 0; p < G.passes; ++p) {
    RadixArgs r{};
    const bool to2 = ((G.passes - 1 - p) & 1) == 0;
    for (int s = 0; s < 3; ++s) {
      r.kin[s] = p == 0 ? kin[s] : (const uint32_t*)(base + (to2 ? G.k1[s] : G.k2[s]));
      r.vin[s] = p == 0 ? nullptr : (const uint32_t*)(base + (to2 ? G.v1[s] : G.perm[s]));
      r.kout[s] = (uint32_t*)(base + (to2 ? G.k2[s] : G.k1[s]));
      r.vout[s] = (uint32_t*)(base + (to2 ? G.perm[s] : G.v1[s]));
      r.hist[s] = (uint32_t*)(base + G.hist[s]);
      r.dsum[s] = r.hist[s] + (int64_t)RADIX * G.nblocks;
    }
    r.M = M; r.nblocks = G.nblocks; r.shift = p * RBITS;
    k_radix_hist<<<dim3(G.nblocks, 3), 256, 0, st>>>(r);
    if (int e = ego_launch_status("k_radix_hist")) return e;
    k_radix_scan<<<dim3(RADIX, 3), 256, 0, st>>>(r);
    if (int e = ego_launch_status("k_radix_scan")) return e;
    k_radix_scatter<<<dim3(G.nblocks, 3), 256, 0, st>>>(r);
    if (int e = ego_launch_status("k_radix_scatter")) return e;
  }
  StartArgs sa{};
  uint32_t kmax = 0;
  for (int s = 0; s < 3; ++s) {
    sa.sorted[s] = (const uint32_t*)(base + G.k2[s]);
    sa.start[s] = (uint32_t*)(base + G.start[s]);
    sa.suboff[s] = (uint32_t*)(base + G.suboff[s]);
    sa.stepsum[s] = (uint32_t*)(base + G.stepsum[s]);
    sa.costsum[s] = (uint32_t*)(base + G.costsum[s]);
    sa.steps[s] = (uint4*)(base + G.steps[s]);
    sa.K[s] = G.K[s]; sa.LC[s] = G.LC[s]; sa.nmin1[s] = (uint32_t)G.res[sort_minor(s)] + 1;
    kmax = G.K[s] > kmax ? G.K[s] : kmax;
  }
  sa.M = M;
  k_cell_starts<<<dim3((kmax + 2 + 255) / 256, 3), 256, 0, st>>>(sa);
  if (int e = ego_launch_status("k_cell_starts")) return e;
  k_line_suboff<<<dim3(1, 3), 1024, 0, st>>>(sa);
  if (int e = ego_launch_status("k_line_suboff")) return e;
  for (int s = 0; s < 3; ++s) sa.scanpart[s] = (uint32_t*)(base + G.scanpart[s]);
  {
    const unsigned nblk = (kmax + 4095) / 4096;
    k_step_scan<0><<<dim3(nblk, 2, 3), 1024, 0, st>>>(sa);
    if (int e = ego_launch_status("k_step_scan<0>")) return e;
    k_step_scan<1><<<dim3(1, 2, 3), 1024, 0, st>>>(sa);
    if (int e = ego_launch_status("k_step_scan<1>")) return e;
    k_step_scan<2><<<dim3(nblk, 2, 3), 1024, 0, st>>>(sa);
    if (int e = ego_launch_status("k_step_scan<2>")) return e;
  }
  k_step_fill<<<dim3((kmax + 255) / 256, 3), 256, 0, st>>>(sa);
  if (int e = ego_launch_status("k_step_fill")) return e;
  return EGO_OK;
}

int ego_scatter_density_sorted(const ego_scene* sc, const ego_vm_grad* gdensity, const float* coords, const float* dfeat, int64_t N, int32_t S,
                               void* workspace, int64_t workspace_bytes, void* stream) {
  EGO_TRACE("ego_scatter_density_sorted");
  if (int e = check_sizes(sc, N, S, "scatter_density_sorted")) return e;
  if (sc->density.n_comp != 16) return ego_fail(EGO_E_UNSUPPORTED, "scatter_density_sorted: n_comp %d (supported: 16)", sc->density.n_comp);
  if (N == 0) return zero_tables(sc->density, gdensity, 16, (hipStream_t)stream, "scatter_density_sorted");
  EGO_REQUIRE(coords && dfeat && workspace, "scatter_density_sorted: null argument");
  const SortGeom G = make_geom(sc->density.res, N * (int64_t)S);
  if (workspace_bytes < G.total) return ego_fail(EGO_E_BADARG, "scatter_density_sorted: workspace too small");
  SortedArgs a{};
  if (int e = fill_args(sc->density, gdensity, coords, dfeat, G, workspace, &a, "scatter_density_sorted")) return e;
  if (walk_wanted()) return launch_walk<16, true>(a, G, (char*)workspace, N * (int64_t)S, nullptr, (hipStream_t)stream);
  return launch_sorted<16, true>(a, G, (hipStream_t)stream);
}

int ego_scatter_app_sorted(const ego_scene* sc, const ego_vm_grad* gapp, const float* coords, const float* dv, const float* dv_absmax,
                           const float* dfe, float* gbasis, int32_t ldg, int64_t N, int32_t S, void* workspace, int64_t workspace_bytes, void* stream) {
  EGO_TRACE("ego_scatter_app_sorted");
  if (int e = check_sizes(sc, N, S, "scatter_app_sorted")) return e;
  if (sc->app.n_comp != 48) return ego_fail(EGO_E_UNSUPPORTED, "scatter_app_sorted: n_comp %d (supported: 48)", sc->app.n_comp);
  if (N == 0) return zero_tables(sc->app, gapp, 48, (hipStream_t)stream, "scatter_app_sorted");
  EGO_REQUIRE(coords && workspace, "scatter_app_sorted: null argument");
  const bool rdv = dv == nullptr;   // v16: dv re-derived in the walk from dfe and the scene's basis matrices
  if (rdv) EGO_REQUIRE(dfe && gbasis && dv_absmax && sc->basis[0] && sc->basis[1],
                       "scatter_app_sorted: dv == NULL asks the walk to re-derive it: dfe, gbasis, dv_absmax (= max |dfe|) and sc->basis are needed");
  const SortGeom G = make_geom(sc->app.res, N * (int64_t)S);
  if (workspace_bytes < G.total) return ego_fail(EGO_E_BADARG, "scatter_app_sorted: workspace too small");
  SortedArgs a{};
  if (int e = fill_args(sc->app, gapp, coords, dv, G, workspace, &a, "scatter_app_sorted")) return e;
  if (walk_wanted() && (N * (int64_t)S + 31) / 32 * 32 * 144 < (1ll << 30))   // (the walk addresses dv with 32-bit element offsets)
    return launch_walk<48, false>(a, G, (char*)workspace, N * (int64_t)S, dv_absmax, (hipStream_t)stream, dfe, gbasis, ldg, rdv ? sc->basis : nullptr);
  if (dfe || gbasis) return ego_fail(EGO_E_UNSUPPORTED, "scatter_app_sorted: d(basis) rides along only in the walk form (EGO_SORTED_WALK=0 or a batch above 2^30 dv elements was asked for)");
  return launch_sorted<48, false>(a, G, (hipStream_t)stream);
}

}  // extern "C"
