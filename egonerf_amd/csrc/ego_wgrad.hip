// Weight gradients of the shade backward: G[ca][cb] += sum over samples m of A[m][ca] * B[m][cb] for tall-skinny fp32
// matrices (M = all samples of the batch, ca <= 128, cb <= 160) — d(W2) = dh2^T h1, d(W1) = dh1^T x, d(W3) = do^T h2,
// d(basis_g) = dfe_g^T v (train.py's loss.backward() through nn.Linear).  The contraction index is the row index, so a
// workgroup stages 32 rows at a time through LDS *transposed* ([column][32 samples]) — the layout the MFMA operands need —
// split into bf16 hi + lo (bf16 keeps the fp32 exponent: gradients of 1e-8 need no scaling), and accumulates its share of
// the rows in registers with three v_mfma_f32_32x32x16_bf16 per product (hi*hi + lo*hi + hi*lo: ~17 significand bits per
// operand, rounded to nearest so the error does not accumulate over the 10^6 terms; compared at 2e-4).  A designated column of B can be replaced by ones, which
// puts the bias gradient (the column sums of A) into the same pass.  One pass over the dumps, HBM-bound.
#include "ego_device.h"
#include "ego_host.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_ROW = 72;  // bytes per LDS row: 32 bf16 samples (64 B) + 8 B padding (2-way instead of 16-way bank conflicts)

struct WgradArgs {
  const float* A;
  const float* a_scale;  // A layout 2: per-row power of two
  const float* B;
  float* G;
  int64_t M;
  int32_t lda, ca, ldb, cb, ones_col, ldg, steps_per_wg;
  const float* fe;       // k_wgrad_h<.., XSYN>: the forward's feature-slot dump and the rays: B = the MLP input x, re-derived
  const float* rays;
  int32_t S;
  float* Gpart;          // deterministic mode: [workgroup][32 CAB][32 CBB] partial products, summed in workgroup order by k_wgrad_reduce
};

// fp32 -> bf16 bits in the high half, round to nearest even (finite inputs)
__device__ __forceinline__ uint32_t bf16_rn(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}

// v = hi + lo + O(2^-18 |v|): hi = bf16(v), lo = bf16(v - hi) (the residual is exact in fp32); rounding to nearest keeps the
// error unbiased — truncation would lose up to 2^-15 |v|, always towards zero, which does not average out over the samples
__device__ __forceinline__ void split_bf16_pair(float v0, float v1, uint32_t& hi, uint32_t& lo) {
  const uint32_t h0 = bf16_rn(v0), h1 = bf16_rn(v1);
  const float r0 = __fsub_rn(v0, __uint_as_float(h0)), r1 = __fsub_rn(v1, __uint_as_float(h1));
  hi = (h0 >> 16) | h1;
  lo = (bf16_rn(r0) >> 16) | bf16_rn(r1);
}

// One matrix tile = rows [row0, row0 + 32) x columns [0, 32 NB), staged to LDS as [term][column][32 samples].  A thread owns
// (column quad, sample pair) items: two float4 global loads (fetch, issued a whole step ahead of their use), then eight
// packed bf16-pair stores (commit).
// BLK: 0 row-major fp32, 1 tile-blocked fp32, 2 tile-blocked fp16 in the shade kernels' operand order ([tile][k-step s][lane =
// 32 h + sample][8 halves], element e = logical column 8 (2 s + e / 4) + 4 h + e % 4; cx / 16 k-steps in memory, the tile's further
// column blocks are zero): ego_shade_backward's dh2 / dh1 with value = half * scale[row], the forward's x / h1 / h2 dumps with scale
// = NULL (the halves are the values),
// 3 row-major fp32 [M][32] standing for 64 logical columns: row m fills columns [32 g, 32 g + 32) with g = (scale[4 m + 3] != 0),
// the other 32 are zero (ego_shade_backward's dfe with `scale` = the forward's coords [M][4]; NB == 2)
template <int NB, bool VEC, int BLK>
struct Tile {
  static constexpr int QUADS = NB * 8;                       // column quads per row
  static constexpr int ITEMS = BLK == 2 ? (NB + 3) / 4 : (16 * QUADS + 255) / 256;     // items per thread
  f32x4 v0[ITEMS], v1[ITEMS];
  float s0, s1;  // BLK == 2: the two samples' scales

  // item -> (column quad, sample pair).  Row-major source: consecutive lanes take consecutive quads of one row (coalesced
  // rows); tile-blocked source (the shade kernels' dump layout, [tile][quad pair][32 h + j][4]): consecutive lanes take
  // consecutive sample pairs of one quad (512 contiguous bytes per 16 lanes, and conflict-free LDS stores)
  static __device__ __forceinline__ void item(int idx, int& cq, int& sp) {
    if (BLK == 1 || BLK == 2) { sp = idx & 15; cq = idx >> 4; } else { cq = idx % QUADS; sp = idx / QUADS; }
  }

  __device__ __forceinline__ void fetch(const float* __restrict__ X, const float* __restrict__ scale, int ldx, int cx, int64_t row0, int64_t M) {
    if (BLK == 2) {  // item = (k-step, lane half, sample pair): 32 contiguous bytes = the 8 halves of two neighbouring lanes
      static_assert(BLK != 2 || NB == 4 || NB == 5, "the fp16 layouts are the 128-column dumps (+ a block for the ones column) and the 160-column x dump");
      const int hw = (threadIdx.x >> 4) & 1, sp = threadIdx.x & 15, steps_mem = cx >> 4;
      const int64_t r0 = row0 + 2 * sp;
      const u32x4 z = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const int s = (threadIdx.x >> 5) + 8 * it;
        const u32x4* p = (const u32x4*)X + ((row0 >> 5) * steps_mem + (s < steps_mem ? s : 0)) * 64 + hw * 32 + 2 * sp;
        v0[it] = __builtin_bit_cast(f32x4, (r0 < M && s < steps_mem) ? p[0] : z);
        v1[it] = __builtin_bit_cast(f32x4, (r0 + 1 < M && s < steps_mem) ? p[1] : z);
      }
      s0 = r0 < M ? (scale ? scale[r0] : 1.f) : 0.f;
      s1 = r0 + 1 < M ? (scale ? scale[r0 + 1] : 1.f) : 0.f;
      return;
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = threadIdx.x + 256 * it;
      int cq, sp;
      item(idx, cq, sp);
      const int64_t r0 = row0 + 2 * sp;
      const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
      v0[it] = z; v1[it] = z;
      if (idx < 16 * QUADS && 4 * cq < cx) {
        if (BLK == 3) {
          static_assert(BLK != 3 || NB == 2, "the grid-routed layout is the 2 x 32-column feature gradient");
          const float* p = X + r0 * 32 + 4 * (cq & 7);
          if (r0 < M && (scale[r0 * 4 + 3] != 0.f) == ((cq >> 3) != 0)) v0[it] = *(const f32x4*)p;
          if (r0 + 1 < M && (scale[(r0 + 1) * 4 + 3] != 0.f) == ((cq >> 3) != 0)) v1[it] = *(const f32x4*)(p + 32);
        } else if (BLK) {  // row0 is a multiple of 32: tile = row0 / 32; logical column quad cq = 2 q + h
          const float* p = X + row0 * ldx + (cq >> 1) * 256 + ((cq & 1) * 32 + 2 * sp) * 4;
          if (r0 < M) v0[it] = *(const f32x4*)p;
          if (r0 + 1 < M) v1[it] = *(const f32x4*)(p + 4);
        } else if (VEC) {  // ldx and cx are multiples of 4 (checked on the host)
          if (r0 < M) v0[it] = *(const f32x4*)(X + r0 * ldx + 4 * cq);
          if (r0 + 1 < M) v1[it] = *(const f32x4*)(X + (r0 + 1) * ldx + 4 * cq);
        } else {    // ragged / unaligned rows (the 3-column d(output) matrix)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (4 * cq + e < cx) {
              if (r0 < M) v0[it][e] = X[r0 * ldx + 4 * cq + e];
              if (r0 + 1 < M) v1[it][e] = X[(r0 + 1) * ldx + 4 * cq + e];
            }
        }
      }
    }
  }

  __device__ __forceinline__ void commit(int ones_col, int64_t row0, int64_t M, uint8_t* __restrict__ lds) const {
    constexpr int COLS = NB * 32;
    if (BLK == 2) {
      typedef _Float16 h8v __attribute__((ext_vector_type(8)));
      const int hw = (threadIdx.x >> 4) & 1, sp = threadIdx.x & 15;
      const int64_t r0 = row0 + 2 * sp;
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const int s = (threadIdx.x >> 5) + 8 * it;
        if (s >= 2 * NB) break;   // k-steps of the tile (memory holds the first cx / 16 of them, fetch left the others zero)
        const h8v a8 = __builtin_bit_cast(h8v, v0[it]), b8 = __builtin_bit_cast(h8v, v1[it]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int col = (2 * s + (e >> 2)) * 8 + hw * 4 + (e & 3);
          float a = __fmul_rn((float)a8[e], s0), b = __fmul_rn((float)b8[e], s1);
          if (col == ones_col) { a = r0 < M ? 1.f : 0.f; b = r0 + 1 < M ? 1.f : 0.f; }
          uint32_t hi, lo;  // half * power of two is exact in fp32; its 11 significant bits split into 8 (hi) + 3 (lo) exactly
          split_bf16_pair(a, b, hi, lo);
          *(uint32_t*)(lds + col * WG_ROW + sp * 4) = hi;
          *(uint32_t*)(lds + (COLS + col) * WG_ROW + sp * 4) = lo;
        }
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = threadIdx.x + 256 * it;
      if (idx >= 16 * QUADS) break;
      int cq, sp;
      item(idx, cq, sp);
      const int64_t r0 = row0 + 2 * sp;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 4 * cq + e;
        float a = v0[it][e], b = v1[it][e];
        if (col == ones_col) { a = r0 < M ? 1.f : 0.f; b = r0 + 1 < M ? 1.f : 0.f; }
        uint32_t hi, lo;
        split_bf16_pair(a, b, hi, lo);
        *(uint32_t*)(lds + col * WG_ROW + sp * 4) = hi;
        *(uint32_t*)(lds + (COLS + col) * WG_ROW + sp * 4) = lo;
      }
    }
  }
};

__device__ __forceinline__ bf8 frag(const uint8_t* row, int ks, int kb) {
  const u32x2* p = (const u32x2*)(row + ks * 32 + kb * 16);
  const u32x2 a = p[0], b = p[1];
  return __builtin_bit_cast(bf8, u32x4{a.x, a.y, b.x, b.y});
}

// 2-3 workgroups per CU hide the row fetches of one behind the multiplies of the others: cap the registers (left alone the
// compiler takes 272 for the 4 x 5 shape, i.e. one workgroup per CU)
template <int CAB, int CBB, bool AVEC, int ABLK, int BBLK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CAB * CBB >= 20 ? 2 : 3, CAB * CBB >= 20 ? 2 : 3))) void k_wgrad(WgradArgs P) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[(CAB + CBB) * 32 * 2 * WG_ROW];
  uint8_t* la = lds;
  uint8_t* lb = lds + CAB * 32 * 2 * WG_ROW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kb = lane >> 5;
  // the CAB x CBB output blocks (32 x 32 each) go to the four waves: a whole block row each when there are four rows (the
  // A fragment is then shared by the wave's blocks), round-robin (b = wave + 4 k) otherwise
  constexpr int NBLK = CAB * CBB, MAXB = (NBLK + 3) / 4;
  f32x16 acc[MAXB];
#pragma unroll
  for (int k = 0; k < MAXB; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  const int64_t step0 = (int64_t)blockIdx.x * P.steps_per_wg;
  Tile<CAB, AVEC, ABLK> ta;
  Tile<CBB, true, BBLK> tb;
  ta.fetch(P.A, P.a_scale, P.lda, P.ca, step0 * 32, P.M);
  tb.fetch(P.B, nullptr, P.ldb, P.cb, step0 * 32, P.M);
  for (int st = 0; st < P.steps_per_wg; ++st) {
    const int64_t row0 = (step0 + st) * 32;
    if (row0 >= P.M) break;  // uniform over the workgroup
    ta.commit(-1, row0, P.M, la);
    tb.commit(P.ones_col, row0, P.M, lb);
    __syncthreads();
    if (st + 1 < P.steps_per_wg) {  // next step's rows travel while this step multiplies
      ta.fetch(P.A, P.a_scale, P.lda, P.ca, row0 + 32, P.M);
      tb.fetch(P.B, nullptr, P.ldb, P.cb, row0 + 32, P.M);
    }
#pragma unroll
    for (int k = 0; k < MAXB; ++k) {
      const int b = CAB == 4 ? wave * CBB + k : wave + 4 * k;
      if (b < NBLK) {
        const int mt = b / CBB, nt = b % CBB;
        const uint8_t* ra = la + (32 * mt + i) * WG_ROW;
        const uint8_t* rb = lb + (32 * nt + i) * WG_ROW;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf8 ah = frag(ra, ks, kb), al = frag(ra + CAB * 32 * WG_ROW, ks, kb);
          const bf8 bh = frag(rb, ks, kb), bl = frag(rb + CBB * 32 * WG_ROW, ks, kb);
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[k], 0, 0, 0);
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[k], 0, 0, 0);
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[k], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // D[row][col]: lane = col + 32 * ((row >> 2) & 1), register (row & 3) + 4 * (row >> 3)
#pragma unroll
  for (int k = 0; k < MAXB; ++k) {
    const int b = CAB == 4 ? wave * CBB + k : wave + 4 * k;
    if (b < NBLK) {
      const int mt = b / CBB, nt = b % CBB;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kb, col = 32 * nt + i;
        const float v = acc[k][r];
        if (P.Gpart) P.Gpart[((int64_t)blockIdx.x * (32 * CAB) + row) * (32 * CBB) + col] = v;
        else if (v != 0.f) unsafeAtomicAdd(P.G + (int64_t)row * P.ldg + col, v);
      }
    }
  }
}

// ---- both operands as halves (the training step's two large products: d(W2) = dh2^T h1, d(W1) = dh1^T x) ------------------------------
// A = ego_shade_backward's dh (fp16 payload x one power of two per sample), B = the forward's h1 / x dump (halves).  A product of two
// halves is exact in fp32, so ONE v_mfma_f32_32x32x16_f16 does what the bf16 hi/lo form needs three instructions and a split of both
// operands for (~240 VALU instructions per thread and 32-row step: those passes were bound by their commit -> barrier -> MFMA -> barrier
// chain, 11 k clocks per step against 1 k of matrix work, not by bytes).  The per-sample scale runs along K, so it cannot be applied
// to the accumulator; within a step (32 samples = one tile) the largest scale `ref` is taken out instead: A's row m (the payload, whose
// largest entry lies in [2^12, 2^13)) is multiplied by inv_m / ref <= 1, a power of two - exact until an entry drops below 2^-14, i.e.
// 26 binades of head room for a sample's largest entries (consecutive samples of a ray differ by many orders of magnitude in their
// gradients; scaling B's O(1) activations instead flushed them after ~10 binades and cost 4 % of a small batch's dW1) -, the step's
// product lands in a scratch accumulator and acc += ref * scratch.
typedef _Float16 h8w __attribute__((ext_vector_type(8)));
typedef _Float16 h2w __attribute__((ext_vector_type(2)));

__device__ __forceinline__ h8w frag_h(const uint8_t* row, int ks, int kb) {
  const u32x2* p = (const u32x2*)(row + ks * 32 + kb * 16);
  const u32x2 a = p[0], b = p[1];
  return __builtin_bit_cast(h8w, u32x4{a.x, a.y, b.x, b.y});
}

struct HTile {   // one operand's rows [row0, row0 + 32): up to two (k-step, lane half, sample pair) items per thread
  u32x4 v0[2], v1[2];
  __device__ __forceinline__ void fetch(const void* __restrict__ X, int steps_mem, int items, int64_t row0, int64_t M) {
    const int hw = (threadIdx.x >> 4) & 1, sp = threadIdx.x & 15;
    const int64_t r0 = row0 + 2 * sp;
    const u32x4 z = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if (it >= items) break;
      const int s = (threadIdx.x >> 5) + 8 * it;
      const u32x4* p = (const u32x4*)X + ((row0 >> 5) * steps_mem + (s < steps_mem ? s : 0)) * 64 + hw * 32 + 2 * sp;
      v0[it] = (r0 < M && s < steps_mem) ? p[0] : z;
      v1[it] = (r0 + 1 < M && s < steps_mem) ? p[1] : z;
    }
  }
  // [column][32 samples] halves; ratio2 = the two samples' inv / ref as packed halves (SCALE: A), ones_col as in the bf16 form (B)
  template <bool SCALE>
  __device__ __forceinline__ void commit(int steps, int items, int ones_col, uint32_t ratio2, uint8_t* __restrict__ lds) const {
    const int hw = (threadIdx.x >> 4) & 1, sp = threadIdx.x & 15;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if (it >= items) break;
      const int s = (threadIdx.x >> 5) + 8 * it;
      if (s >= steps) break;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = (2 * s + (e >> 2)) * 8 + hw * 4 + (e & 3);
        const uint32_t a = v0[it][e >> 1], b = v1[it][e >> 1];
        uint32_t word = (e & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));   // (sample 2 sp, sample 2 sp + 1)
        if (col == ones_col) word = 0x3c003c00u;   // (1.0, 1.0); rows beyond M meet zero rows of A
        if (SCALE) word = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2w, word) * __builtin_bit_cast(h2w, ratio2));
        *(uint32_t*)(lds + col * WG_ROW + sp * 4) = word;
      }
    }
  }
};

// ---- B = the layer-1 input x, re-derived instead of read back (round 5) ----------------------------------------------------------------
// x is a pure function of the 14 feature slots of a lane half and the ray's view direction (tensorBase.py:68-75: [features, viewdirs,
// PE(features), PE(viewdirs)] in the shade kernels' K order: slot r -> columns 5 r .. 5 r + 4 = f, sin f, sin 2f, cos f, cos 2f; then the
// eight view values of the half; then two zeros).  The training forward used to dump it as 160 halves per sample (0.67 GB written, 0.67 GB
// read back per 8192 x 256 step) only for d(W1) = dh1^T x; the feature slots are dumped anyway (128 B per sample, the shade backward needs
// them), so this fetch builds the same 8 halves per (k-step, lane half, sample) with the forward's own instructions (sincos_x_2x_hw,
// round-to-nearest pack): bit-identical operands, 192 B per sample less to read and 320 B less to write.
__device__ __forceinline__ u32x4 pack8_rn_w(const float x[8]) {
  u32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    o[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2v{x[2 * q], x[2 * q + 1]}, h2v));
  }
  return o;
}

// One k-step (compile-time STEP) of a sample's lane half hw, in two parts so that the loads travel while the previous step multiplies:
// x_load = the (at most three) feature slots the step's eight values come from (fe_lane = the half's slot dump, &fe[(tile * 4) * 64 + lane]
// as floats, quad q 256 floats further on); x_make = the eight values from them and the ray's direction, packed as the forward packs them.
constexpr int NSLOT_W = 14;
template <int STEP>
__device__ __forceinline__ void x_load(const float* __restrict__ fe_lane, float f[3]) {
  constexpr int r_first = (STEP * 8) / 5;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    constexpr int dummy = 0; (void)dummy;
    const int r = r_first + k;
    f[k] = (r < NSLOT_W && r * 5 <= STEP * 8 + 7) ? fe_lane[(r >> 2) * 256 + (r & 3)] : 0.f;
  }
}
template <int STEP>
__device__ __forceinline__ u32x4 x_make(const float f[3], const float dir[3], int hw) {
  constexpr int r_first = (STEP * 8) / 5;
  float xs[8], s1[3], c1[3], s2[3], c2[3], vw[8];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (r_first + k < NSLOT_W && (r_first + k) * 5 <= STEP * 8 + 7) sincos_x_2x_hw(f[k], s1[k], c1[k], s2[k], c2[k]);
  if (STEP >= 8) {   // the view values live in the last two steps
    float sa0, ca0, sb0, cb0, sa1, ca1, sb1, cb1, sa2, ca2, sb2, cb2;
    sincos_x_2x_hw(dir[0], sa0, ca0, sb0, cb0);
    sincos_x_2x_hw(dir[1], sa1, ca1, sb1, cb1);
    sincos_x_2x_hw(dir[2], sa2, ca2, sb2, cb2);
    vw[0] = hw ? sb2 : dir[0]; vw[1] = hw ? ca0 : dir[1]; vw[2] = hw ? cb0 : dir[2]; vw[3] = hw ? ca1 : sa0;
    vw[4] = hw ? cb1 : sb0; vw[5] = hw ? ca2 : sa1; vw[6] = hw ? cb2 : sb1; vw[7] = hw ? 0.f : sa2;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kk = STEP * 8 + e;
    float x;
    if (kk < 5 * NSLOT_W) {
      const int k = kk / 5 - r_first, kind = kk % 5;
      x = kind == 0 ? f[k] : (kind == 1 ? s1[k] : (kind == 2 ? s2[k] : (kind == 3 ? c1[k] : c2[k])));
    } else if (kk < 5 * NSLOT_W + 8) {
      x = vw[kk - 5 * NSLOT_W];
    } else {
      x = 0.f;
    }
    xs[e] = x;
  }
  return pack8_rn_w(xs);
}
#define EGO_X_CASES(CALL)                                                                                                        \
  switch (s) {                                                                                                                  \
    case 0: CALL(0); break; case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; \
    case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break; case 9: CALL(9); break; \
    default: break;                                                                                                             \
  }
__device__ __forceinline__ void x_load_any(int s, const float* __restrict__ fe_lane, float f[3]) {
  f[0] = f[1] = f[2] = 0.f;
#define EGO_X_CALL(N) x_load<N>(fe_lane, f)
  EGO_X_CASES(EGO_X_CALL)
#undef EGO_X_CALL
}
__device__ __forceinline__ u32x4 x_make_any(int s, const float f[3], const float dir[3], int hw) {
  u32x4 o = u32x4{0u, 0u, 0u, 0u};
#define EGO_X_CALL(N) o = x_make<N>(f, dir, hw)
  EGO_X_CASES(EGO_X_CALL)
#undef EGO_X_CALL
  return o;
}
#undef EGO_X_CASES

template <int CBB, bool XSYN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_wgrad_h(WgradArgs P) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[(4 + CBB) * 32 * WG_ROW];
  uint8_t* la = lds;
  uint8_t* lb = lds + 4 * 32 * WG_ROW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kb = lane >> 5;
  f32x16 acc[CBB];
#pragma unroll
  for (int k = 0; k < CBB; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  const int64_t step0 = (int64_t)blockIdx.x * P.steps_per_wg;
  const int steps_b = P.cb >> 4;   // k-steps of B in memory; the tile's further steps (ones / padding columns) are fetched as zeros
  HTile ta, tb;
  float x_f[3][3], x_dir[3];   // XSYN: the feature slots of the thread's three (k-step, half, sample) items and its ray's direction
  bool x_in = false;
  u32x4 xb[3];                 // ... and the items themselves, made from them behind the previous step's MFMAs (off the barrier chain)
  auto make_x = [&]() {
#pragma unroll
    for (int t = 0; t < 3; ++t) xb[t] = x_in ? x_make_any(wave + 4 * t, x_f[t], x_dir, kb) : u32x4{0u, 0u, 0u, 0u};
  };
  float inv0, inv1;   // the thread's two samples' scales (rows row0 + 2 sp, + 1)
  auto fetch = [&](int64_t row0) {
    ta.fetch(P.A, 8, 1, row0, P.M);
    if (XSYN) {
      // items = (k-step s = wave + 4 t, lane half hw = lane >> 5, sample row0 + (lane & 31)): the k-step is WAVE-uniform (no divergence in
      // x_step_any's switch - the first form, with HTile's item order, executed two cases per wave and cost the step 0.15 ms)
      const uint32_t m = (uint32_t)row0 + (uint32_t)i;
      const bool in = (int64_t)m < P.M;
      // (the prefetch of a workgroup's last step may name a tile behind the last one: clamped, its rows are `!in` and never used)
      const int64_t tile = (row0 < P.M ? row0 : P.M - 1) >> 5;
      const float* fe_l = P.fe + (tile * 4 * 64 + lane) * 4;   // this lane's slots: quad q 256 floats further on
      const float* dir = P.rays + (int64_t)((in ? m : 0u) / (uint32_t)P.S) * 6 + 3;
      x_in = in;
      x_dir[0] = dir[0]; x_dir[1] = dir[1]; x_dir[2] = dir[2];
#pragma unroll
      for (int t = 0; t < 3; ++t) x_load_any(wave + 4 * t, fe_l, x_f[t]);   // loads only: the values are made at commit time
    } else tb.fetch(P.B, steps_b, 2, row0, P.M);
    const int64_t r0 = row0 + 2 * (threadIdx.x & 15);
    inv0 = r0 < P.M ? P.a_scale[r0] : 0.f;
    inv1 = r0 + 1 < P.M ? P.a_scale[r0 + 1] : 0.f;
  };
  fetch(step0 * 32);
  if (XSYN) make_x();
  for (int st = 0; st < P.steps_per_wg; ++st) {
    const int64_t row0 = (step0 + st) * 32;
    if (row0 >= P.M) break;  // uniform over the workgroup
    // the step's largest scale: 16 consecutive lanes hold the 32 rows
    float ref = fmaxf(inv0, inv1);
#pragma unroll
    for (int sh = 1; sh < 16; sh <<= 1) ref = fmaxf(ref, __shfl_xor(ref, sh, 16));
    const float rinv = ref > 0.f ? 1.f / ref : 0.f;   // powers of two: exact
    const h2w ratio = h2w{(_Float16)(inv0 * rinv), (_Float16)(inv1 * rinv)};
    ta.commit<true>(8, 1, -1, __builtin_bit_cast(uint32_t, ratio), la);
    if (XSYN) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int s = wave + 4 * t;
        if (s >= 2 * CBB) break;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int col = (2 * s + (e >> 2)) * 8 + kb * 4 + (e & 3);
          uint16_t hv = (uint16_t)(xb[t][e >> 1] >> ((e & 1) * 16));
          if (col == P.ones_col) hv = 0x3c00u;   // 1.0; rows beyond M meet zero rows of A
          *(uint16_t*)(lb + col * WG_ROW + i * 2) = hv;
        }
      }
    } else tb.commit<false>(2 * CBB, 2, P.ones_col, 0u, lb);
    __syncthreads();
    if (st + 1 < P.steps_per_wg) fetch(row0 + 32);   // next step's rows travel while this step multiplies
    const uint8_t* ra = la + (32 * wave + i) * WG_ROW;
    const h8w a0 = frag_h(ra, 0, kb), a1 = frag_h(ra, 1, kb);
#pragma unroll
    for (int nt = 0; nt < CBB; ++nt) {
      const uint8_t* rb = lb + (32 * nt + i) * WG_ROW;
      f32x16 t;
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = 0.f;
      t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, frag_h(rb, 0, kb), t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, frag_h(rb, 1, kb), t, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = fmaf(ref, t[r], acc[nt][r]);
    }
    if (XSYN && st + 1 < P.steps_per_wg) make_x();   // the next step's items, while this step's MFMAs drain
    __syncthreads();
  }
#pragma unroll
  for (int nt = 0; nt < CBB; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kb, col = 32 * nt + i;
      const float v = acc[nt][r];
      if (P.Gpart) P.Gpart[((int64_t)blockIdx.x * 128 + row) * (32 * CBB) + col] = v;
      else if (v != 0.f) unsafeAtomicAdd(P.G + (int64_t)row * P.ldg + col, v);
    }
}

// ---- d(W3) = do^T h2 (+ d(b3) = column sums of do): three rows -----------------------------------------------------------------------------
// A = d(output) [M][3] fp32, B = the forward's h2 dump (halves, 128 columns, operand order).  The generic kernel above runs this product as a
// 32 x 160 MFMA tile with 29 empty rows and a bf16 split of both operands: commit -> barrier -> MFMA -> barrier per 32 samples, 0.20 ms for
// 0.56 GB.  It is 2.1 M x 384 multiply-adds - nothing for the VALU: thread (k-step s, lane) keeps the eight columns its 16-byte item holds
// ([tile][s][lane = 32 h + j][8 halves]: the same eight columns of sample j in every tile) as 3 x 8 fp32 sums over the workgroup's tiles in
// order, the 32 lanes of a half are added as a tree at the end.  fp32 products of exact halves; streaming, no LDS staging, no barrier in the loop.
__global__ __launch_bounds__(512) void k_wgrad3(WgradArgs P) {
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  const int s = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
  const int64_t tile0 = (int64_t)blockIdx.x * P.steps_per_wg, n_tiles = (P.M + 31) >> 5;
  float acc[3][8], accb[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
  const u32x4* B = (const u32x4*)P.B;
  auto fetch = [&](int64_t tile, u32x4& x, float d[3]) {
    const int64_t t = tile < n_tiles ? tile : n_tiles - 1, m = t * 32 + j;
    const bool in = tile < n_tiles && m < P.M;
    const int64_t mc = in ? m : P.M - 1;
    const u32x4 xl = B[(t * 8 + s) * 64 + lane];
    x = in ? xl : u32x4{0u, 0u, 0u, 0u};   // rows behind M may hold anything (0 x NaN)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = P.A[mc * P.lda + c];
      d[c] = in ? v : 0.f;
    }
  };
  // two tiles in flight behind the two being added (one 16-byte load per lane and tile: a single tile ahead left the HBM queue short)
  u32x4 x[2], xn[2];
  float d[2][3], dn[2][3];
  fetch(tile0, x[0], d[0]);
  fetch(tile0 + 1, x[1], d[1]);
  for (int st = 0; st < P.steps_per_wg; st += 2) {
    const int64_t tile = tile0 + st;
    if (tile >= n_tiles) break;   // uniform over the workgroup
    const bool more = st + 2 < P.steps_per_wg;
    if (more) { fetch(tile + 2, xn[0], dn[0]); fetch(tile + 3, xn[1], dn[1]); }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (st + k >= P.steps_per_wg) break;   // odd step count: the second slot of the last pair belongs to the next workgroup
      const h8v hv = __builtin_bit_cast(h8v, x[k]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[c][e] = fmaf(d[k][c], (float)hv[e], acc[c][e]);
        accb[c] += d[k][c];
      }
    }
    if (more) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        x[k] = xn[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) d[k][c] = dn[k][c];
      }
    }
  }
  // the 32 samples of the half: balanced tree (fixed order)
#pragma unroll
  for (int sh = 1; sh <= 16; sh <<= 1) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[c][e] += __shfl_xor(acc[c][e], sh, 64);
      accb[c] += __shfl_xor(accb[c], sh, 64);
    }
  }
  if (j != 0) return;
  constexpr int LD = 160;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = (2 * s + (e >> 2)) * 8 + h * 4 + (e & 3);
      if (P.Gpart) P.Gpart[((int64_t)blockIdx.x * 3 + c) * LD + col] = acc[c][e];
      else if (acc[c][e] != 0.f) unsafeAtomicAdd(P.G + (int64_t)c * P.ldg + col, acc[c][e]);
    }
    if (s == 0 && h == 0) {   // the remaining columns of the padded row: the ones column (bias gradient), zeros elsewhere
      for (int col = 128; col < LD; ++col) {
        const float v = col == P.ones_col ? accb[c] : 0.f;
        if (P.Gpart) P.Gpart[((int64_t)blockIdx.x * 3 + c) * LD + col] = v;
        else if (v != 0.f) unsafeAtomicAdd(P.G + (int64_t)c * P.ldg + col, v);
      }
    }
  }
}

// deterministic mode: G[row][col] = sum over workgroups, in workgroup order (sixteen contiguous segments, then the segments in order),
// of the partial products; overwrites G's [rows][ld] block (no zero fill needed)
__global__ __launch_bounds__(1024) void k_wgrad_reduce(const float* __restrict__ part, int n_wg, int rows, int ld, float* __restrict__ G, int ldg) {
  __shared__ float q[16][64];
  const int l = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + l;
  const int E = rows * ld;
  const int nq = (n_wg + 15) / 16, w0 = s * nq, w1 = min(w0 + nq, n_wg);
  float sum = 0.f;
  if (e < E) {
    const float* p = part + e;
#pragma unroll 8
    for (int w = w0; w < w1; ++w) sum += p[(int64_t)w * E];
  }
  q[s][l] = sum;
  __syncthreads();
  if (s == 0 && e < E) {
    float t = q[0][l];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += q[k][l];
    G[(int64_t)(e / ld) * ldg + e % ld] = t;
  }
}

inline int reduce_partials(const WgradArgs& p, unsigned n_wg, int rows, int ld, hipStream_t st) {
  k_wgrad_reduce<<<(unsigned)((rows * ld + 63) / 64), 1024, 0, st>>>(p.Gpart, (int)n_wg, rows, ld, p.G, p.ldg);
  return ego_launch_status("k_wgrad_reduce");
}

template <int CBB, bool XSYN = false>
int launch_h(const WgradArgs& a, hipStream_t st) {
  WgradArgs p = a;
  const int64_t steps = (a.M + 31) / 32;
  const int64_t cap = a.Gpart ? 512 : 1024;   // ordered mode: one resident round of workgroups (2 per CU), half the partial blocks to store and add
  const int64_t wgs = steps < cap ? steps : cap;
  p.steps_per_wg = (int32_t)((steps + wgs - 1) / wgs);
  const unsigned n_wg = (unsigned)((steps + p.steps_per_wg - 1) / p.steps_per_wg);
  k_wgrad_h<CBB, XSYN><<<n_wg, 256, 0, st>>>(p);
  if (int e = ego_launch_status("k_wgrad_h")) return e;
  return p.Gpart ? reduce_partials(p, n_wg, 128, 32 * CBB, st) : EGO_OK;
}

int launch_w3(const WgradArgs& a, hipStream_t st) {
  WgradArgs p = a;
  const int64_t steps = (a.M + 31) / 32;
  const int64_t wgs = steps < 512 ? steps : 512;
  p.steps_per_wg = (int32_t)((steps + wgs - 1) / wgs);
  const unsigned n_wg = (unsigned)((steps + p.steps_per_wg - 1) / p.steps_per_wg);
  k_wgrad3<<<n_wg, 512, 0, st>>>(p);
  if (int e = ego_launch_status("k_wgrad3")) return e;
  return p.Gpart ? reduce_partials(p, n_wg, 3, 160, st) : EGO_OK;
}

template <int CAB, int CBB, bool AVEC, int ABLK, int BBLK>
int launch(const WgradArgs& a, hipStream_t st) {
  WgradArgs p = a;
  const int64_t steps = (a.M + 31) / 32;
  const int64_t wgs = steps < 768 ? steps : 768;
  p.steps_per_wg = (int32_t)((steps + wgs - 1) / wgs);
  const unsigned n_wg = (unsigned)((steps + p.steps_per_wg - 1) / p.steps_per_wg);
  k_wgrad<CAB, CBB, AVEC, ABLK, BBLK><<<n_wg, 256, 0, st>>>(p);
  if (int e = ego_launch_status("k_wgrad")) return e;
  return p.Gpart ? reduce_partials(p, n_wg, 32 * CAB, 32 * CBB, st) : EGO_OK;
}

}  // namespace

extern "C" {

int ego_weight_grad_x(const void* dh1, const float* dh_scale, const float* fe, const float* rays, int32_t S, int32_t ones_col, int64_t M, float* G,
                      int32_t ldg, float* partial, int64_t partial_floats, void* stream) {
  EGO_TRACE("ego_weight_grad_x");
  EGO_REQUIRE(M >= 0 && S >= 1 && ones_col < 160 && ldg >= 160, "weight_grad_x: bad size (ldg >= 160, ones_col < 160)");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(dh1 && dh_scale && fe && rays && G && ((uintptr_t)dh1 & 15) == 0 && ((uintptr_t)fe & 15) == 0, "weight_grad_x: null or unaligned argument");
  EGO_REQUIRE(!partial || partial_floats >= (int64_t)1024 * 128 * 160, "weight_grad_x: partial buffer smaller than ego_weight_grad_partial_floats()");
  WgradArgs a{(const float*)dh1, dh_scale, nullptr, G, M, 128, 128, 160, 160, ones_col, ldg, 0, fe, rays, S, partial};
  return launch_h<5, true>(a, (hipStream_t)stream);
}

int64_t ego_weight_grad_partial_floats(void) { return (int64_t)1024 * 128 * 160; }   // workgroups x rows x padded columns, the largest product

int ego_weight_grad(const void* A, int32_t lda, int32_t ca, int32_t a_layout, const float* a_scale, const void* B, int32_t ldb, int32_t cb,
                    int32_t b_layout, int32_t ones_col, int64_t M, float* G, int32_t ldg, void* stream) {
  return ego_weight_grad_det(A, lda, ca, a_layout, a_scale, B, ldb, cb, b_layout, ones_col, M, G, ldg, nullptr, 0, stream);
}

int ego_weight_grad_det(const void* A, int32_t lda, int32_t ca, int32_t a_layout, const float* a_scale, const void* B, int32_t ldb, int32_t cb,
                        int32_t b_layout, int32_t ones_col, int64_t M, float* G, int32_t ldg, float* partial, int64_t partial_floats, void* stream) {
  EGO_TRACE("ego_weight_grad");
  EGO_REQUIRE(!partial || partial_floats >= ego_weight_grad_partial_floats(), "weight_grad: partial buffer smaller than ego_weight_grad_partial_floats()");
  EGO_REQUIRE(a_layout >= 0 && a_layout <= 3, "weight_grad: a_layout must be 0 (row-major), 1 (blocked fp32), 2 (blocked scaled fp16) or 3 (grid-routed)");
  EGO_REQUIRE(b_layout >= 0 && b_layout <= 2, "weight_grad: b_layout must be 0 (row-major), 1 (blocked fp32) or 2 (blocked fp16)");
  EGO_REQUIRE(b_layout != 2 || ((cb & 15) == 0 && ldb == cb), "weight_grad: the fp16 B layout holds whole k-steps (cb a multiple of 16, ldb = cb)");
  EGO_REQUIRE(M >= 0 && ca >= 1 && ca <= 128 && cb >= 1 && cb <= 160 && (lda >= ca || a_layout == 3) && ldb >= cb && ones_col < 160,
              "weight_grad: bad size (ca <= 128, cb <= 160)");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(A && B && G, "weight_grad: null argument");
  EGO_REQUIRE(a_layout != 2 || (a_scale && ca == 128 && lda == 128 && ((uintptr_t)A & 15) == 0),
              "weight_grad: the scaled-fp16 layout needs a_scale, 128 columns and a 16-byte aligned A");
  EGO_REQUIRE(a_layout != 3 || (a_scale && ca == 64 && lda == 32 && ((uintptr_t)A & 15) == 0),
              "weight_grad: the grid-routed layout needs a_scale (= coords [M][4]), ca = 64, lda = 32 and a 16-byte aligned A");
  const int cab = (ca + 31) / 32;
  const int cbb = ((ones_col >= cb ? ones_col + 1 : cb) + 31) / 32;
  EGO_REQUIRE(ldg >= 32 * cbb, "weight_grad: ldg must cover the padded column blocks");
  EGO_REQUIRE((ldb & 3) == 0 && (cb & 3) == 0 && ((uintptr_t)B & 15) == 0, "weight_grad: B rows must be 16-byte aligned, cb a multiple of 4");
  const bool avec = (lda & 3) == 0 && (ca & 3) == 0 && ((uintptr_t)A & 15) == 0;
  WgradArgs a{(const float*)A, a_scale, (const float*)B, G, M, lda, ca, ldb, cb, ones_col, ldg, 0, nullptr, nullptr, 0, partial};
  const hipStream_t st = (hipStream_t)stream;
  // instantiations: the training step's four products, their all-fp32 forms and the all-row-major forms of the same shapes
  // (the training step's four products: dh2^T h1, dh1^T x: scaled-fp16 A, fp16 B; do^T h2: ragged row-major A, fp16 B; dfe^T v: grid-routed A,
  // blocked fp32 B)
  const int key = (cab * 8 + cbb) * 16 + a_layout * 4 + b_layout;
  switch (key) {
    case (1 * 8 + 5) * 16 + 0: return avec ? launch<1, 5, true, 0, 0>(a, st) : launch<1, 5, false, 0, 0>(a, st);
    case (1 * 8 + 5) * 16 + 1: return avec ? launch<1, 5, true, 0, 1>(a, st) : launch<1, 5, false, 0, 1>(a, st);
    case (1 * 8 + 5) * 16 + 2:
      if (ca == 3 && cb == 128 && lda == 3 && (ones_col < 0 || ones_col >= 128) && ((uintptr_t)B & 15) == 0 && !getenv("EGO_WGRAD3_MFMA")) return launch_w3(a, st);
      return avec ? launch<1, 5, true, 0, 2>(a, st) : launch<1, 5, false, 0, 2>(a, st);
    case (2 * 8 + 5) * 16 + 0: if (avec) return launch<2, 5, true, 0, 0>(a, st); break;
    case (2 * 8 + 5) * 16 + 1: if (avec) return launch<2, 5, true, 0, 1>(a, st); break;
    case (2 * 8 + 5) * 16 + 13: return launch<2, 5, true, 3, 1>(a, st);
    case (2 * 8 + 5) * 16 + 14: return launch<2, 5, true, 3, 2>(a, st);
    case (4 * 8 + 5) * 16 + 0: if (avec) return launch<4, 5, true, 0, 0>(a, st); break;
    case (4 * 8 + 5) * 16 + 5: if (avec) return launch<4, 5, true, 1, 1>(a, st); break;
    case (4 * 8 + 5) * 16 + 9: return launch<4, 5, true, 2, 1>(a, st);
    case (4 * 8 + 5) * 16 + 10: return getenv("EGO_WGRAD_BF16") ? launch<4, 5, true, 2, 2>(a, st) : launch_h<5>(a, st);   // both operands halves: the fp16 MFMA form
    case (4 * 8 + 4) * 16 + 0: if (avec) return launch<4, 4, true, 0, 0>(a, st); break;
    default: break;
  }
  return ego_fail(EGO_E_UNSUPPORTED, "weight_grad: no instantiation for %d x %d column blocks%s (A layout %d, B layout %d)", cab, cbb,
                  avec ? "" : " with unaligned A rows", a_layout, b_layout);
}

}  // extern "C"
