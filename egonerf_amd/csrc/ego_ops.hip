// libegonerf_hip.so, part 1: the separately callable stages (rows A-E, I, J of SURVEY 8a) and the
// fused marching / compositing kernels.  gfx950 only.
#include <atomic>
#include <dlfcn.h>
#include <stdlib.h>
#include "ego_device.h"
#include "ego_host.h"
#include "ego_generic.h"

// =============================================================================================
// Row A  — sample schedule -> points      models/EgoNeRF.py:56-87
// =============================================================================================
__device__ __forceinline__ float sched_z(const float* __restrict__ r_sched, const float* __restrict__ jitter,
                                         int64_t ray, int s, int S, float near_) {
  float r = r_sched[s];
  if (jitter) {
    const float step = (s < S - 1) ? __fsub_rn(r_sched[s + 1], r) : __fsub_rn(r, r_sched[S - 2]);
    r = __fadd_rn(r, __fmul_rn(step, jitter[ray * S + s]));
  }
  return __fadd_rn(near_, r);
}

__global__ void k_sample_ray_exp(const float* __restrict__ rays, const float* __restrict__ r_sched,
                                 const float* __restrict__ jitter, float near_, int64_t N, int S,
                                 float* __restrict__ xyz, float* __restrict__ z_out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * S) return;
  const int64_t ray = idx / S;
  const int s = (int)(idx - ray * S);
  const float z = sched_z(r_sched, jitter, ray, s, S, near_);
  if (z_out) z_out[idx] = z;
  if (xyz) {
    const float* R = rays + ray * 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) xyz[idx * 3 + k] = __fadd_rn(R[k], __fmul_rn(R[3 + k], z));
  }
}

// =============================================================================================
// Rows B, C
// =============================================================================================
__global__ void k_from_cartesian(DevCoords c, const float* __restrict__ xyz, int64_t M, float* __restrict__ c7) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const YinYang y = yinyang_from_xyz(xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], c);
  float* o = c7 + i * 7;
  const int b = y.yang ? 3 : 0, nb = y.yang ? 0 : 3;
  o[b] = y.r; o[b + 1] = y.th; o[b + 2] = y.ph;
  o[nb] = 0.f; o[nb + 1] = 0.f; o[nb + 2] = 0.f;
  o[6] = y.yang ? 1.f : 0.f;
}

__global__ void k_normalize_coord(DevCoords c, const float* __restrict__ c7, int64_t M, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float* p = c7 + i * 7;
  float* o = out + i * 7;
#pragma unroll
  for (int b = 0; b < 6; b += 3) {
    o[b] = normalize_r(p[b], c.r_lut, c.n_lut, c.n_r);
    o[b + 1] = normalize_ang(p[b + 1], c.th_near, c.th_inv);
    o[b + 2] = normalize_ang(p[b + 2], c.ph_near, c.ph_inv);
  }
  o[6] = p[6];
}

// =============================================================================================
// Row D / D' — density feature: sum_i relu(sum_c P_ic * L_ic)      models/EgoNeRF.py:291-347, 232-289
// lane = sample; one bilinear tap = C contiguous floats (C/4 x 16-byte loads).
// =============================================================================================
template <int C>
__device__ __forceinline__ float density_lookup(const DevField& F, int g, float a_r, float a_th, float a_ph) {
#pragma clang fp contract(fast)  // the library is built with -ffp-contract=off; interpolation may use FMAs
  const VMTaps t = vm_setup(a_r, a_th, a_ph, F.res);
  float feat = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const Lin1 X = t.ax[vm_plane_x(i)], Y = t.ax[vm_plane_y(i)], Ln = t.ax[vm_line_ax(i)];
    const int W = F.res[vm_plane_x(i)];
    const float* P = g ? F.plane[1][i] : F.plane[0][i];
    const float* L = g ? F.line[1][i] : F.line[0][i];
    const f32x4* p00 = (const f32x4*)(P + ((int64_t)Y.i0 * W + X.i0) * C);
    const f32x4* p01 = (const f32x4*)(P + ((int64_t)Y.i0 * W + X.i1) * C);
    const f32x4* p10 = (const f32x4*)(P + ((int64_t)Y.i1 * W + X.i0) * C);
    const f32x4* p11 = (const f32x4*)(P + ((int64_t)Y.i1 * W + X.i1) * C);
    const f32x4* l0 = (const f32x4*)(L + (int64_t)Ln.i0 * C);
    const f32x4* l1 = (const f32x4*)(L + (int64_t)Ln.i1 * C);
    const float w00 = __fmul_rn(Y.w0, X.w0), w01 = __fmul_rn(Y.w0, X.w1);
    const float w10 = __fmul_rn(Y.w1, X.w0), w11 = __fmul_rn(Y.w1, X.w1);
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
      const f32x4 pv = p00[q] * w00 + p01[q] * w01 + p10[q] * w10 + p11[q] * w11;
      const f32x4 lv = l0[q] * Ln.w0 + l1[q] * Ln.w1;
      const f32x4 m = pv * lv;
      dot += (m.x + m.y) + (m.z + m.w);
    }
    feat += fmaxf(dot, 0.f);
  }
  return feat;
}

template <int C>
__global__ void k_density_feature(DevField F, const float* __restrict__ c7n, int64_t M, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float* p = c7n + i * 7;
  const int g = (p[6] == 0.f) ? 0 : 1;
  const int b = g ? 3 : 0;
  out[i] = density_lookup<C>(F, g, p[b], p[b + 1], p[b + 2]);
}

// =============================================================================================
// Row E
// =============================================================================================
__global__ void k_feature2density(const float* __restrict__ f, int64_t M, int softplus, float shift,
                                  float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  out[i] = softplus ? softplus_shift(f[i], shift) : fmaxf(f[i], 0.f);
}

// one wave per ray; transmittance = exclusive product of (1 - alpha + 1e-10)   tensorBase.py:22-27
__global__ void k_raw2alpha(const float* __restrict__ sigma, const float* __restrict__ dist, int64_t N, int S,
                            float* __restrict__ alpha, float* __restrict__ weight, float* __restrict__ bg) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (ray >= N) return;
  float carry = 1.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool ok = s < S;
    const float a = ok ? alpha_from(sigma[ray * S + s] * dist[ray * S + s]) : 0.f;
    const float t = ok ? __fadd_rn(__fsub_rn(1.f, a), 1e-10f) : 1.f;
    const float inc = wave_scan_mul(t, lane);
    float exc = __shfl_up(inc, 1, 64);
    if (lane == 0) exc = 1.f;
    const float T = carry * exc;
    if (ok) {
      if (alpha) alpha[ray * S + s] = a;
      if (weight) weight[ray * S + s] = a * T;
    }
    carry *= __shfl(inc, 63, 64);
  }
  if (bg && lane == 0) bg[ray] = carry;
}

// =============================================================================================
// Row M — occupancy lookup: trilinear F.grid_sample (align_corners, zeros) on a {0,1} volume [N_phi][N_theta][N_r]
// models/EgoNeRF.py:11-24
// =============================================================================================
struct DevOcc {
  const uint8_t* vol;  // [2][res2][res1][res0] or null
  int32_t res[3];
  const uint8_t* cell; // optional [2][res2-1][res1-1][res0-1]: OR of the 8 corner voxels of every cell (ego_scene.occ_cell)
};

__device__ __forceinline__ float occ_sample(const DevOcc& O, int g, float a_r, float a_th, float a_ph) {
  const Lin1 X = lin_setup(a_r, O.res[0]), Y = lin_setup(a_th, O.res[1]), Z = lin_setup(a_ph, O.res[2]);
  const uint8_t* V = O.vol + (int64_t)g * O.res[0] * O.res[1] * O.res[2];
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int ix = (k & 1) ? X.i1 : X.i0, iy = (k & 2) ? Y.i1 : Y.i0, iz = (k & 4) ? Z.i1 : Z.i0;
    const float w = ((k & 1) ? X.w1 : X.w0) * ((k & 2) ? Y.w1 : Y.w0) * ((k & 4) ? Z.w1 : Z.w0);
    v += w * (float)V[((int64_t)iz * O.res[1] + iy) * O.res[0] + ix];
  }
  return v;
}

// "mask value > 0" (tensorBase.py:464-478) for the march: a sample strictly inside a cell (all six axis weights > 0, all taps in
// range) has a positive trilinear value iff any of the cell's eight corner voxels is set - one byte of the per-cell OR volume instead
// of eight dependent byte loads and the interpolation; samples on a lattice plane / outside the volume take the exact evaluation
__device__ __forceinline__ bool occ_occupied(const DevOcc& O, int g, float a_r, float a_th, float a_ph) {
  if (O.cell) {
    const Lin1 X = lin_setup(a_r, O.res[0]), Y = lin_setup(a_th, O.res[1]), Z = lin_setup(a_ph, O.res[2]);
    const bool interior = fminf(fminf(fminf(X.w0, X.w1), fminf(Y.w0, Y.w1)), fminf(Z.w0, Z.w1)) > 0.f;
    if (interior) {
      const int64_t c0 = O.res[0] - 1, c1 = O.res[1] - 1, c2 = O.res[2] - 1;
      return O.cell[(((int64_t)g * c2 + Z.i0) * c1 + Y.i0) * c0 + X.i0] != 0;
    }
  }
  return occ_sample(O, g, a_r, a_th, a_ph) > 0.f;
}

__global__ void k_alpha_mask_sample(DevOcc O, const float* __restrict__ c7n, int64_t M, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float* p = c7n + i * 7;
  const int g = (p[6] == 0.f) ? 0 : 1;
  const int b = g ? 3 : 0;
  out[i] = occ_sample(O, g, p[b], p[b + 1], p[b + 2]);
}

// =============================================================================================
// Fused: rows A+B+C+D+E for one ray per wave.  models/EgoNeRF.py:507-529 (coarse) / 544-553 (fine)
//
// A pass = 64 consecutive samples of the ray, in three phases that hand data over through the wave's own LDS slab:
//   A  lane = sample: distance, point, yin-yang coordinates, radius normalisation (O(1) knot estimate from the LUT's own linear
//      run / exponential tail, checked against the four knots around it, binary search only if the check fails), the three axis
//      tap set-ups -> byte offsets (grid table offsets folded in) + weights to LDS (6 x 16 B per sample)
//   B  lane = (team t = lane >> 2, part p = lane & 3), four rounds of 16 samples x three planes: a team reads its sample's set-up
//      (6 x ds_read_b128 per round: no per-lane recomputation of the set-up, no cross-lane bpermute, no table select), gathers
//      the 18 taps as whole 64-byte lines (part p = one float4 of each) in a two-deep software pipeline (loads of stage k + 1
//      issued before the arithmetic of stage k), reduces inside the quad with DPP; its leader leaves the feature in LDS
//   C  lane = sample: softplus, alpha, DPP transmittance scan with carry, weights, tile flags, outputs
// Round 2's form did phase B with four rounds of five __shfl hand-overs (ds_bpermute + wait each), two more per plane for the
// channel reduction and six in the scan, and recomputed the tap set-up in each of a team's four lanes: 1 256 VALU instructions and
// 66 LDS-queue round trips per pass against 1 000 and 32 here.  Measured (tools/march_timing.py, us old -> new): 4096 x 512
// 106.1 -> 105.2, 16384 x 128 on the pooled tables 106.0 -> 99.7, 16384 x 256 with explicit distances 207.3 -> 183.2; with the
// tap loads replaced by register values 52 us: the kernel is co-limited by the vector L1's 16 clk per load instruction (72 loads
// per pass = 64 us at 16 waves per CU) and VALU issue, which overlap only partly.  Three waves per SIMD (167 VGPRs): at four the
// two 24-register load buffers spill (92 B of scratch per lane, 31 MB per launch through L2: 124 us).  DESIGN.md 4.4.
// =============================================================================================
struct LutHints {   // derived per workgroup from the LUT itself (so any monotone LUT works: the estimate is only a starting point)
  float r0;         // first knot spacing (lut[1] - lut[0])
  float lin_end;    // last knot of the arithmetic run
  float inv_l2r;    // 1 / log2(ratio) of the exponential tail
  float shift;      // tail knots: shift + r0' * ratio^k
  float inv_r0;
  float tail_k0;    // index offset of the tail estimate
};

// searchsorted(lut, r, right) clamped to [1, n - 1] (coordinates.py:125-127) without the 10-step dependent search: estimate the
// knot from the closed forms, read the four knots around it, accept if they bracket r, else search.  Bit-identical to normalize_r.
__device__ __forceinline__ float normalize_r_fast(float r, const float* lut, int n_lut, int n_r, const LutHints& H) {
  float kf;
  if (r < H.lin_end) kf = r * H.inv_r0;
  else kf = fmaf(__log2f(fmaxf(r - H.shift, 1e-30f)), H.inv_l2r, H.tail_k0);
  kf = fminf(fmaxf(kf, 0.f), (float)(n_lut - 1));
  const int c0 = (int)kf + 1;                      // estimate of lo = #{i : lut[i] <= r}
  // knots c0 - 2 .. c0 + 1; virtual knots below index 0 are <= r, beyond the end are > r
  int lo = c0 - 2;
  bool ok = r == r;
#pragma unroll
  for (int d = -2; d <= 1; ++d) {
    const int j = c0 + d;
    const bool in = (unsigned)j < (unsigned)n_lut;
    const float v = lut[in ? j : 0];
    const bool le = in ? (v <= r) : (j < 0);
    lo += le ? 1 : 0;
    if (d == -2) ok = ok && le;                    // the window's first knot must not exceed r
    if (d == 1) ok = ok && (!le || j >= n_lut - 1);  // and its last one must (unless it is the LUT's end)
  }
  if (!ok) {   // rare: estimate off by more than a knot (or NaN): the reference's search
    int l = 0, h = n_lut;
    while (l < h) {
      const int mid = (l + h) >> 1;
      if (!(lut[mid] > r)) l = mid + 1; else h = mid;
    }
    lo = l;
  }
  const int k_out = lo < 1 ? 1 : (lo > n_lut - 1 ? n_lut - 1 : lo);
  const int k_in = k_out - 1;
  const float g0 = lut[k_in], g1 = lut[k_out];
  const float frac = __fdiv_rn(__fsub_rn(r, g0), __fsub_rn(g1, g0));
  const float v = __fdiv_rn(__fadd_rn((float)k_in, frac), (float)n_r);
  return __fsub_rn(__fmul_rn(v, 2.0f), 1.0f);
}

// NSPLIT (1, 2): a ray's passes are dealt to NSPLIT waves of the workgroup (segments of consecutive passes).  One wave per ray
// leaves the launch quantised by wave slots: 3 waves per SIMD = 3 072 slots, so 4 096 rays take a full round and a third of one, 106 us
// where 3 072 rays take 76 and the sustained rate (19.6 us per 1 024 rays) would give 80 (tools/march_timing.py sweep).  Shorter waves
// pack the slots better: 4 096 x 512 with two waves per ray 98 us.  Each wave pays the workgroup prologue (LUT, hints) and the hand-over,
// so it only pays with few rounds of slots and long rays (four waves per ray: 109 us; 16 384 x 128 split in two: 101 -> 118 us); wave
// priorities by phase (+7 %) and a staggered start of the waves of a SIMD (+-0) were measured and are not in the kernel.  Segment 0 streams as before (it knows the transmittance in front of it); a later segment evaluates its
// passes with the transmittance unknown, parks (alpha, in-pass exclusive product) per sample and the pass totals in LDS, and turns
// them into weights once its predecessor has published the carry - with the same multiplications in the same order as the
// one-wave form, so z / alpha / weight / bg / tile flags are bit-identical for every NSPLIT.  One documented difference:
// `coords_out` of passes that the one-wave form stops evaluating (transmittance exactly 0 in front of them, no per-sample alpha asked
// for) holds zeros there and the real coordinates in a deferred segment, which cannot know yet; those samples have weight 0, their
// tiles are never shaded, and tests/test_hip_march_split.py compares coordinates where a colour is read.
constexpr int MARCH_MAXP = 4;   // passes per deferred segment (LDS parking space)
template <int C, bool OCC, int NSPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_march_density(DevCoords c, DevField F, const float* __restrict__ rays,
                                                       int64_t N, int S, const float* __restrict__ z_in,
                                                       const float* __restrict__ r_sched,
                                                       const float* __restrict__ jitter, float near_,
                                                       int softplus, float shift, float dscale,
                                                       float* __restrict__ z_out, float* __restrict__ alpha,
                                                       int alpha_stride, float* __restrict__ weight,
                                                       float* __restrict__ bg, float* __restrict__ coords_out,
                                                       float* __restrict__ sigma_out, DevOcc occ, float term_eps,
                                                       float shade_above, uint8_t* __restrict__ tile_active) {
  static_assert(C == 16, "one texel = one 64-byte line = the team's four float4 parts");
  __shared__ float lut[1024];
  __shared__ __attribute__((aligned(16))) f32x4 stage[4][6][64];   // per wave: the pass's tap set-ups
  __shared__ float fres[4][64];                                      // per wave: the pass's density features
  __shared__ float park[NSPLIT > 1 ? 4 : 1][NSPLIT > 1 ? MARCH_MAXP : 1][2][64];   // deferred segments: alpha, exclusive product
  __shared__ float park_tot[4][MARCH_MAXP];
  __shared__ float carry_slot[4];
  __shared__ int s_nlin;
  __shared__ LutHints s_hints;
  for (int i = threadIdx.x; i < c.n_lut; i += blockDim.x) lut[i] = c.r_lut[i];
  if (threadIdx.x == 0) s_nlin = c.n_lut - 1;
  __syncthreads();
  {
    // arithmetic run = the leading intervals equal to lut[1] - lut[0] up to float32 rounding of k * r0 (the reference replaces
    // every shell thinner than r0 by a step of r0): the first interval that is not ends it
    const float r0 = lut[1] - lut[0];
    int first_other = c.n_lut - 1;
    for (int i = threadIdx.x; i + 1 < c.n_lut; i += blockDim.x)
      if (!(fabsf((lut[i + 1] - lut[i]) - r0) <= r0 * 1e-3f)) first_other = min(first_other, i);
    if (first_other < c.n_lut - 1) atomicMin(&s_nlin, first_other);
    __syncthreads();
    if (threadIdx.x == 0) {
      const int n = c.n_lut, nl = s_nlin;
      LutHints H;
      H.r0 = r0; H.inv_r0 = 1.f / r0; H.lin_end = lut[nl];
      // tail: lut[k] = shift + a * ratio^k: ratio from the last three knots, shift from the last two
      float ratio = 2.f, sh = 0.f;
      if (n - nl >= 4) {
        const float a = lut[n - 3], b = lut[n - 2], cc = lut[n - 1];
        ratio = (cc - b) / (b - a);
        sh = b - (cc - b) / (ratio - 1.f);
      }
      H.shift = sh;
      H.inv_l2r = 1.f / __log2f(fmaxf(ratio, 1.0000001f));
      // kf = log2(r - shift) * inv_l2r + tail_k0 must give kf = k for r = lut[k]: calibrate on the last knot
      H.tail_k0 = (float)(n - 1) - __log2f(fmaxf(lut[n - 1] - sh, 1e-30f)) * H.inv_l2r;
      if (n - nl < 4) { H.lin_end = 3.0e38f; }   // (almost) all linear: never take the tail form
      s_hints = H;
    }
    __syncthreads();
  }
  LutHints H;   // wave-uniform: keep the six values in SGPRs (read from LDS they would occupy six VGPRs for the whole kernel)
  {
    const auto sf = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    H.r0 = sf(s_hints.r0); H.lin_end = sf(s_hints.lin_end); H.inv_l2r = sf(s_hints.inv_l2r); H.shift = sf(s_hints.shift);
    H.inv_r0 = sf(s_hints.inv_r0); H.tail_k0 = sf(s_hints.tail_k0);
  }
  // the wave index is wave-uniform: readfirstlane puts it - and with it the ray index and every per-ray base address - into SGPRs
  const int lane0 = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int seg = NSPLIT > 1 ? (wv & (NSPLIT - 1)) : 0;
  const int64_t ray_raw = (int64_t)blockIdx.x * (4 / NSPLIT) + wv / NSPLIT;
  const bool live = ray_raw < N;
  if (NSPLIT == 1 && !live) return;            // split launches meet at workgroup barriers below: nobody leaves early
  const int64_t ray = live ? ray_raw : N - 1;
  const int n_pass = (S + 63) >> 6, per_seg = (n_pass + NSPLIT - 1) / NSPLIT;
  const int pass0 = seg * per_seg, pass1 = live ? min(n_pass, pass0 + per_seg) : pass0;
  const bool defer = NSPLIT > 1 && seg > 0;    // wave-uniform
  const float* R = rays + ray * 6;
  const float ox = R[0], oy = R[1], oz = R[2], dx = R[3], dy = R[4], dz = R[5];
  const int nr64 = F.res[0] * (C * 4), nth64 = F.res[1] * (C * 4);
  float carry = 1.f;
  // Exact early termination: once the transmittance in front of a pass is exactly 0 (fp32 underflow behind opaque samples), every
  // remaining weight is a * 0 = 0 and bg stays 0, so the rest of the ray only needs its distances and zero weights - unless the
  // caller wants per-sample alpha / sigma, which are independent of what lies in front.
  const bool may_stop = !alpha && !sigma_out && !defer;
  // weights, tile flags and the carry of one pass from (alpha, in-pass exclusive product, pass total) and the carry in front of it
  auto emit = [&](int s0, int lane, float a, float exc, float tot) {
    const int s = min(s0 + lane, S - 1);
    const bool ok = (s0 + lane) < S;
    const float T = carry * exc;
    const float wgt = (term_eps > 0.f && T < term_eps) ? 0.f : a * T;  // early termination (opt-in)
    if (tile_active) {
      // 32-sample shade tiles are cut from the flat [N*S] order: lanes 0-31 / 32-63 of this pass are (parts of) tiles
      // a tile is shaded iff it holds a sample whose colour is read: weight > shade_above (0, or rayMarch_weight_thres)
      const unsigned long long nz = __ballot(ok && wgt > shade_above);
      const int64_t o = ray * S + s;
      if ((S & 31) == 0) {
        // whole tiles per pass half: written unconditionally (0 or 1), so the caller need not clear the flags first
        if (ok && (lane & 31) == 0) tile_active[o >> 5] = (nz >> (lane & 32) & 0xffffffffull) != 0ull ? 1 : 0;
      } else if (ok && (nz >> (lane & 32) & 0xffffffffull) != 0ull && ((lane & 31) == 0 || (o & 31) == 0)) {
        tile_active[o >> 5] = 1;  // tiles straddle rays: flags are pre-zeroed by the caller and only set here
      }
    }
    if (ok && weight) weight[ray * S + s] = wgt;
    carry *= tot;
  };
  for (int s0 = pass0 * 64; s0 < pass1 * 64; s0 += 64) {
    // lane-derived values (team, part, LDS slab addresses, ...) are recomputed per pass from an opaque copy of the lane index:
    // hoisted out of the loop they cost a dozen VGPRs for the whole kernel, which then spill around the 48-register load buffers
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int team = lane >> 2, part = lane & 3;
    const int s = min(s0 + lane, S - 1);
    const bool ok = (s0 + lane) < S;
    // z[s] and its right neighbour (left neighbour for the last sample: dists repeat the last interval)
    const int sn = (s < S - 1) ? s + 1 : s - 1;
    float z, zn;
    if (z_in) {
      z = z_in[ray * S + s];
      zn = z_in[ray * S + sn];
    } else {
      z = sched_z(r_sched, jitter, ray, s, S, near_);
      zn = sched_z(r_sched, jitter, ray, sn, S, near_);
    }
    if (may_stop && carry == 0.f) {  // wave-uniform
      if (ok) {
        const int64_t o = ray * S + s;
        if (z_out) z_out[o] = z;
        if (coords_out) ((f32x4*)coords_out)[o] = f32x4{0.f, 0.f, 0.f, 0.f};  // never shaded (tile flags stay 0); defined values anyway
        if (weight) weight[o] = 0.f;
        if (tile_active && (S & 31) == 0 && (lane & 31) == 0) tile_active[o >> 5] = 0;
      }
      continue;
    }
    // ---- phase A: lane = sample -----------------------------------------------------------------------------------
    const float dist = (s < S - 1) ? __fsub_rn(zn, z) : __fsub_rn(z, zn);
    const float px = __fadd_rn(ox, __fmul_rn(dx, z)), py = __fadd_rn(oy, __fmul_rn(dy, z)),
                pz = __fadd_rn(oz, __fmul_rn(dz, z));
    const YinYang y = yinyang_from_xyz(px, py, pz, c);
    const float a_r = normalize_r_fast(y.r, lut, c.n_lut, c.n_r, H);
    const float a_th = normalize_ang(y.th, c.th_near, c.th_inv);
    const float a_ph = normalize_ang(y.ph, c.ph_near, c.ph_inv);
    // occupancy mask (opt-in): unoccupied samples keep sigma = 0
    const bool occupied = !OCC || !occ.vol || occ_occupied(occ, y.yang, a_r, a_th, a_ph);
    // a pass whose 64 samples are ALL masked out skips the set-up hand-over and the whole gather (wave-uniform: the software
    // pipeline of phase B is either run completely or not at all, so its wait counts stay exact); sigma is 0 for every lane then
    const bool pass_occupied = !OCC || __ballot(occupied) != 0ull;
    if (pass_occupied) {
    {
      // byte offsets of every tap row / line of the sample's grid (the grid's table offsets folded in here, once per sample, so
      // that phase B needs no per-lane table select) + column offsets + the six axis weights
      const Lin1 Rr = lin_setup(a_r, F.res[0]), Tt = lin_setup(a_th, F.res[1]), Pp = lin_setup(a_ph, F.res[2]);
      const auto fu = [](uint32_t v) { return __uint_as_float(v); };
      const bool yg = y.yang != 0;
      const uint32_t pb0 = yg ? F.poff[1][0] : F.poff[0][0], pb1 = yg ? F.poff[1][1] : F.poff[0][1], pb2 = yg ? F.poff[1][2] : F.poff[0][2];
      const uint32_t lb0 = yg ? F.loff[1][0] : F.loff[0][0], lb1 = yg ? F.loff[1][1] : F.loff[0][1], lb2 = yg ? F.loff[1][2] : F.loff[0][2];
      constexpr uint32_t TX = C * 4;   // bytes per texel
      stage[wv][0][lane] = f32x4{fu(Rr.i0 * TX), fu(Rr.i1 * TX), fu(Tt.i0 * TX), fu(Tt.i1 * TX)};                             // columns: r | theta
      stage[wv][1][lane] = f32x4{fu(pb0 + Tt.i0 * nr64), fu(pb0 + Tt.i1 * nr64), fu(pb1 + Pp.i0 * nr64), fu(pb1 + Pp.i1 * nr64)};  // rows: plane 0 | plane 1
      stage[wv][2][lane] = f32x4{fu(pb2 + Pp.i0 * nth64), fu(pb2 + Pp.i1 * nth64), fu(lb0 + Pp.i0 * TX), fu(lb0 + Pp.i1 * TX)};   // rows: plane 2 | line 0 (phi)
      stage[wv][3][lane] = f32x4{fu(lb1 + Tt.i0 * TX), fu(lb1 + Tt.i1 * TX), fu(lb2 + Rr.i0 * TX), fu(lb2 + Rr.i1 * TX)};         // line 1 (theta) | line 2 (r)
      stage[wv][4][lane] = f32x4{Rr.w0, Rr.w1, Tt.w0, Tt.w1};
      stage[wv][5][lane] = f32x4{Pp.w0, Pp.w1, fu(occupied ? 1u : 0u), 0.f};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- phase B: lane = (team, part), four rounds of 16 samples x three planes = 12 stages of six 16-byte loads -------------
    // Software pipeline: the loads of stage k + 1 are issued before the arithmetic of stage k (two 6-register-quad buffers), so
    // six to twelve loads stay in flight instead of the queue draining once per plane; a round's set-up (6 x ds_read_b128) is
    // fetched while the previous round's last plane computes.
    {
#pragma clang fp contract(fast)
      f32x4 raw[2][6];
      float wgt6[2][6];
      f32x4 q[6];
      float feat = 0.f;
      const uint32_t p16 = 16u * (uint32_t)part;
      const auto ui = [](float v) { return __float_as_uint(v); };
      auto fetch_setup = [&](int rd, f32x4* dst) {
        const int src = 16 * rd + team;
#pragma unroll
        for (int k = 0; k < 6; ++k) dst[k] = stage[wv][k][src];
      };
      auto issue = [&](const f32x4* Q, int i, f32x4* r6, float* w6) {
        // plane i: columns (x axis), rows (y axis, table offset included), line taps; weights of x / y / line axis
        const uint32_t c0 = ui(i == 2 ? Q[0].z : Q[0].x) + p16, c1 = ui(i == 2 ? Q[0].w : Q[0].y) + p16;
        const uint32_t r0 = ui(i == 0 ? Q[1].x : (i == 1 ? Q[1].z : Q[2].x)), r1 = ui(i == 0 ? Q[1].y : (i == 1 ? Q[1].w : Q[2].y));
        const uint32_t l0 = ui(i == 0 ? Q[2].z : (i == 1 ? Q[3].x : Q[3].z)) + p16, l1 = ui(i == 0 ? Q[2].w : (i == 1 ? Q[3].y : Q[3].w)) + p16;
        const float xw0 = i == 2 ? Q[4].z : Q[4].x, xw1 = i == 2 ? Q[4].w : Q[4].y;
        const float yw0 = i == 0 ? Q[4].z : Q[5].x, yw1 = i == 0 ? Q[4].w : Q[5].y;
        const float lw0 = i == 0 ? Q[5].x : (i == 1 ? Q[4].z : Q[4].x), lw1 = i == 0 ? Q[5].y : (i == 1 ? Q[4].w : Q[4].y);
        r6[0] = *(const f32x4*)(F.base + (r0 + c0)); r6[1] = *(const f32x4*)(F.base + (r0 + c1));
        r6[2] = *(const f32x4*)(F.base + (r1 + c0)); r6[3] = *(const f32x4*)(F.base + (r1 + c1));
        r6[4] = *(const f32x4*)(F.base + l0); r6[5] = *(const f32x4*)(F.base + l1);
        w6[0] = __fmul_rn(yw0, xw0); w6[1] = __fmul_rn(yw0, xw1); w6[2] = __fmul_rn(yw1, xw0); w6[3] = __fmul_rn(yw1, xw1);
        w6[4] = lw0; w6[5] = lw1;
        // every weight in a register of its own: a packed multiply then broadcasts the LOW half of its operand pair; the high-half
        // broadcast form (op_sel without op_sel_hi) is the code-generation feature DESIGN.md 5.1 ties to non-reproducible results
#pragma unroll
        for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(w6[k]));
      };
      auto finish = [&](const f32x4* r6, const float* w6) {
        const f32x4 pv = r6[0] * w6[0] + r6[1] * w6[1] + r6[2] * w6[2] + r6[3] * w6[3];
        const f32x4 lv = r6[4] * w6[4] + r6[5] * w6[5];
        const f32x4 m = pv * lv;
        float dot = (m.x + m.y) + (m.z + m.w);
        // the team's four lanes are one DPP quad: two quad_perm adds (no LDS round trip, unlike __shfl_xor's ds_bpermute)
        dot += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(dot), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
        dot += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(dot), 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, true));
        feat += fmaxf(dot, 0.f);         // relu per plane (EgoNeRF.py:340,346)
      };
      fetch_setup(0, q);
      issue(q, 0, raw[0], wgt6[0]);
      bool masked = OCC && !(__float_as_int(q[5].z) & 1), masked_next = false;
#pragma unroll
      for (int st = 0; st < 12; ++st) {
        const int rd = st / 3, i = st % 3;
        if (st + 1 < 12) issue(q, (st + 1) % 3, raw[(st + 1) & 1], wgt6[(st + 1) & 1]);
        if (i == 1 && rd + 1 < 4) {   // the set-up is dead once the round's last plane has been issued: fetch the next round's
          fetch_setup(rd + 1, q);
          masked_next = OCC && !(__float_as_int(q[5].z) & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        finish(raw[st & 1], wgt6[st & 1]);
        if (i == 2) {
          if (masked) feat = 0.f;   // sample masked out: sigma stays 0 (feature unused)
          if (part == 0) fres[wv][16 * rd + team] = feat;
          feat = 0.f;
          masked = masked_next;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }  // pass_occupied
    // ---- phase C: lane = sample ------------------------------------------------------------------------------------------
    const float f = pass_occupied ? fres[wv][lane] : 0.f;
    float sg = 0.f;
    if (occupied) sg = softplus ? softplus_shift(f, shift) : fmaxf(f, 0.f);
    const float a = ok ? alpha_from(sg * __fmul_rn(dist, dscale)) : 0.f;
    const float t = ok ? __fadd_rn(__fsub_rn(1.f, a), 1e-10f) : 1.f;
    const float inc = wave_scan_mul(t, lane);
    float exc = __shfl_up(inc, 1, 64);
    if (lane == 0) exc = 1.f;
    const float tot = __shfl(inc, 63, 64);
    if (ok) {
      const int64_t o = ray * S + s;
      if (z_out) z_out[o] = z;
      if (coords_out) ((f32x4*)coords_out)[o] = f32x4{a_r, a_th, a_ph, y.yang ? 1.f : 0.f};
      if (sigma_out) sigma_out[o] = sg;
      if (alpha) alpha[ray * alpha_stride + s] = a;
    }
    if (!defer) {
      emit(s0, lane, a, exc, tot);
    } else {
      const int kp = (s0 >> 6) - pass0;
      park[NSPLIT > 1 ? wv : 0][NSPLIT > 1 ? kp : 0][0][lane] = a;
      park[NSPLIT > 1 ? wv : 0][NSPLIT > 1 ? kp : 0][1][lane] = exc;
      if (lane == 0) park_tot[wv][kp] = tot;
    }
  }
  if (NSPLIT > 1) {
    // the carry travels down the ray's segments: segment k turns its parked passes into weights once segment k - 1 has published
    if (!defer && lane0 == 0) carry_slot[wv] = carry;
#pragma unroll
    for (int k = 1; k < NSPLIT; ++k) {
      __syncthreads();
      if (seg == k && live) {
        carry = carry_slot[wv - 1];
        for (int kp = 0; kp < pass1 - pass0; ++kp) {
          int lane = lane0;
          asm volatile("" : "+v"(lane));
          emit((pass0 + kp) * 64, lane, park[NSPLIT > 1 ? wv : 0][NSPLIT > 1 ? kp : 0][0][lane], park[NSPLIT > 1 ? wv : 0][NSPLIT > 1 ? kp : 0][1][lane], park_tot[wv][kp]);
        }
        if (lane0 == 0) carry_slot[wv] = carry;
      }
    }
    if (live && seg == 0 && alpha && lane0 < alpha_stride - S) alpha[ray * alpha_stride + S + lane0] = 1.f;
    if (live && seg == NSPLIT - 1 && bg && lane0 == 0) bg[ray] = carry;
    return;
  }
  // with an environment map the reference appends a column of ones to alpha (EgoNeRF.py:587)
  if (alpha && lane0 < alpha_stride - S) alpha[ray * alpha_stride + S + lane0] = 1.f;
  if (bg && lane0 == 0) bg[ray] = carry;
}

// =============================================================================================
// Row J — environment map     models/envmap.py:6-14, 26-34
// =============================================================================================
__global__ void k_envmap(const float* __restrict__ em, int h, const float* __restrict__ dirs, int64_t N,
                         float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float o[3];
  envmap_lookup(em, h, dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2], o);
  out[i * 3] = o[0]; out[i * 3 + 1] = o[1]; out[i * 3 + 2] = o[2];
}

// backward of bg_weight * sigmoid(bilinear(emission)) into d(emission); thread per ray, 12 float atomics
__global__ void k_envmap_bwd(int h, const float* __restrict__ dirs, int dstride, const float* __restrict__ g_rgb,
                             const float* __restrict__ rgb_raw, const float* __restrict__ bgw,
                             const float* __restrict__ env_map, int64_t N, float* __restrict__ g_em) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float dx = dirs[i * dstride], dy = dirs[i * dstride + 1], dz = dirs[i * dstride + 2];
  const float nrm = fmaxf(__fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz))), 1e-12f);
  const float nx = __fdiv_rn(dx, nrm), ny = __fdiv_rn(dy, nrm), nz = __fdiv_rn(dz, nrm);
  const float u = __fmul_rn(__fadd_rn(nz, 1.f), 0.5f);
  const float v = __fdiv_rn(__fadd_rn(atan2f(ny, nx), 3.14159265358979323846f), 6.28318530717958647692f);
  const Lin1 X = lin_setup(__fsub_rn(__fmul_rn(u, 2.f), 1.f), h);
  const Lin1 Y = lin_setup(__fsub_rn(__fmul_rn(v, 2.f), 1.f), 2 * h);
  const float b = bgw[i];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float raw = rgb_raw[i * 3 + ch];
    const float g = (raw >= 0.f && raw <= 1.f) ? g_rgb[i * 3 + ch] : 0.f;  // clamp(0,1) backward
    const float e = env_map[i * 3 + ch];
    const float ge = g * b * e * (1.f - e);
    if (ge == 0.f) continue;
    float* E = g_em + (int64_t)ch * 2 * h * h;
    unsafeAtomicAdd(E + (int64_t)Y.i0 * h + X.i0, ge * Y.w0 * X.w0);
    unsafeAtomicAdd(E + (int64_t)Y.i0 * h + X.i1, ge * Y.w0 * X.w1);
    unsafeAtomicAdd(E + (int64_t)Y.i1 * h + X.i0, ge * Y.w1 * X.w0);
    unsafeAtomicAdd(E + (int64_t)Y.i1 * h + X.i1, ge * Y.w1 * X.w1);
  }
}

// Row G alternative — SHRender (models/tensorBase.py:30-34 + models/sh.py:87-112, degree 2): rgb_c = relu(sum_k Y_k(d) f[9c+k] + 0.5)
__global__ void k_sh_render(const float* __restrict__ dirs, const float* __restrict__ feat, int64_t M, float* __restrict__ rgb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float x = dirs[i * 3], y = dirs[i * 3 + 1], z = dirs[i * 3 + 2];
  const float xx = __fmul_rn(x, x), yy = __fmul_rn(y, y), zz = __fmul_rn(z, z);
  const float xy = __fmul_rn(x, y), yz = __fmul_rn(y, z), xz = __fmul_rn(x, z);
  float Y[9];
  Y[0] = 0.28209479177387814f;
  Y[1] = __fmul_rn(-0.4886025119029199f, y);
  Y[2] = __fmul_rn(0.4886025119029199f, z);
  Y[3] = __fmul_rn(-0.4886025119029199f, x);
  Y[4] = __fmul_rn(1.0925484305920792f, xy);
  Y[5] = __fmul_rn(-1.0925484305920792f, yz);
  Y[6] = __fmul_rn(0.31539156525252005f, __fsub_rn(__fsub_rn(__fmul_rn(2.0f, zz), xx), yy));
  Y[7] = __fmul_rn(-1.0925484305920792f, xz);
  Y[8] = __fmul_rn(0.5462742152960396f, __fsub_rn(xx, yy));
  const float* f = feat + i * 27;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) s = __fadd_rn(s, __fmul_rn(Y[k], f[9 * c + k]));
    rgb[i * 3 + c] = fmaxf(__fadd_rn(s, 0.5f), 0.f);
  }
}

// =============================================================================================
// Row H — compositing, one wave per ray      models/EgoNeRF.py:579-598
// =============================================================================================
__global__ void k_composite(const float* __restrict__ em, int em_h, const float* __restrict__ rays,
                            const float* __restrict__ z, const float* __restrict__ weight,
                            const float* __restrict__ bgw, const float* __restrict__ rgb, int64_t N, int S,
                            float* __restrict__ rgb_map, float* __restrict__ depth, float* __restrict__ bg_map,
                            float* __restrict__ env_map, float* __restrict__ rgb_raw, float shade_above) {
  const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
  const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (ray >= N) return;
  // Summation order = the folded shade kernel's (ego_shade_composite, csrc/ego_shade.hip): lane j of a half adds samples j, j + 32, j + 64,
  // ... in that order (separately rounded products), then the 32 lanes of the half are added as a balanced tree.  The two forms of
  // ego_render_forward therefore return the same bits, and a batch renders identically whichever form its size selects.  Half 0 sums
  // the colours, half 1 the depth; both sum the weights.
  float acc = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, dp = 0.f;
  for (int s = j; s < S; s += 32) {
    const int64_t o = ray * S + s;
    const float w = weight[o];
    acc += w;
    if (half) {
      dp += w * z[o];
    } else if (w > shade_above) {
      // colour only from samples above the threshold (tensorBase.py:482-487; 0 without one: weights are >= 0, and tiles
      // skipped by ego_shade - mask / early termination / all below the threshold - never wrote their rgb)
      cr += w * rgb[o * 3];
      cg += w * rgb[o * 3 + 1];
      cb += w * rgb[o * 3 + 2];
    }
  }
#pragma unroll
  for (int d = 1; d <= 16; d <<= 1) {
    acc += __shfl_xor(acc, d, 64); cr += __shfl_xor(cr, d, 64); cg += __shfl_xor(cg, d, 64); cb += __shfl_xor(cb, d, 64); dp += __shfl_xor(dp, d, 64);
  }
  dp = __shfl(dp, 32, 64);
  if (lane != 0) return;
  const float* R = rays + ray * 6;
  if (em) {
    float e[3];
    envmap_lookup(em, em_h, R[3], R[4], R[5], e);
    const float b = bgw[ray];
    const float bx = b * e[0], by = b * e[1], bz = b * e[2];
    cr += bx; cg += by; cb += bz;
    if (bg_map) { bg_map[ray * 3] = bx; bg_map[ray * 3 + 1] = by; bg_map[ray * 3 + 2] = bz; }
    if (env_map) { env_map[ray * 3] = e[0]; env_map[ray * 3 + 1] = e[1]; env_map[ray * 3 + 2] = e[2]; }
  }
  if (rgb_raw) { rgb_raw[ray * 3] = cr; rgb_raw[ray * 3 + 1] = cg; rgb_raw[ray * 3 + 2] = cb; }
  rgb_map[ray * 3] = fminf(fmaxf(cr, 0.f), 1.f);
  rgb_map[ray * 3 + 1] = fminf(fmaxf(cg, 0.f), 1.f);
  rgb_map[ray * 3 + 2] = fminf(fmaxf(cb, 0.f), 1.f);
  if (depth) depth[ray] = dp + (1.f - acc) * R[5];  // (1-acc) * d_z: reference quirk, EgoNeRF.py:598
}

// =============================================================================================
// Row I — inverse-CDF resampling + sort, one workgroup (256 threads) per ray
// dataLoader/ray_utils.py:156-187 and models/EgoNeRF.py:532-542
// =============================================================================================
#define PDF_MAX 2048
__global__ __launch_bounds__(256) void k_sample_pdf_merge(const float* __restrict__ z, const float* __restrict__ weight,
                                                          const float* __restrict__ u_in, int Sc, int n_fine,
                                                          int use_coarse, float* __restrict__ z_out,
                                                          float* __restrict__ z_new_out) {
  __shared__ float cdf[PDF_MAX];   // [Sc-1] entries: 0, cumsum(pdf)
  __shared__ float keys[PDF_MAX];  // sort buffer
  __shared__ double red[256];
  const int64_t ray = blockIdx.x;
  const int tid = threadIdx.x;
  const float* zr = z + ray * Sc;
  const float* wr = weight + ray * Sc;
  const int nb = Sc - 1;  // bins = midpoints z_mid[0..Sc-2]
  const int nw = Sc - 2;  // pdf entries = weight[1..Sc-2]
  // sum(w + 1e-5) and the cdf are accumulated in double and rounded to float per element, like ATen's CPU
  // sum/cumsum (acc_type<float> = double): the inverse CDF is discontinuous at u == 1 when the last bin is
  // thinner than 1e-5, so the rounding of cdf[-1] is observable
  double part = 0.0;
  for (int i = tid; i < nw; i += 256) part += (double)__fadd_rn(wr[1 + i], 1e-5f);
  red[tid] = part;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (tid < d) red[tid] += red[tid + d];
    __syncthreads();
  }
  const float total = (float)red[0];
  for (int i = tid; i < nw; i += 256) keys[i] = __fdiv_rn(__fadd_rn(wr[1 + i], 1e-5f), total);
  __syncthreads();
  if (tid < 64) {
    double carry = 0.0;
    for (int s0 = 0; s0 < nw; s0 += 64) {
      const int i = s0 + tid;
      double v = (i < nw) ? (double)keys[i] : 0.0;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(v, d, 64);
        if (tid >= d) v += o;
      }
      if (i < nw) cdf[i + 1] = (float)(carry + v);
      carry += __shfl(v, 63, 64);
    }
    if (tid == 0) cdf[0] = 0.f;
  }
  __syncthreads();
  // inverse CDF
  const int n_out = use_coarse ? Sc + n_fine : n_fine;
  for (int j = tid; j < n_fine; j += 256) {
    float u;
    if (u_in) u = u_in[ray * n_fine + j];
    else {  // torch.linspace(0, 1, n) in float32: symmetric evaluation from both ends
      const float step = __fdiv_rn(1.f, (float)(n_fine - 1));
      u = (n_fine == 1) ? 0.f : (j < n_fine / 2 ? __fmul_rn(step, (float)j) : __fsub_rn(1.f, __fmul_rn(step, (float)(n_fine - 1 - j))));
    }
    int lo = 0, hi = nb;  // searchsorted right over cdf[0..nb-1]
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (!(cdf[mid] > u)) lo = mid + 1; else hi = mid;
    }
    const int below = max(lo - 1, 0), above = min(lo, nb - 1);
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = 0.5f * __fadd_rn(zr[below + 1], zr[below]);
    const float b1 = 0.5f * __fadd_rn(zr[above + 1], zr[above]);
    float den = __fsub_rn(c1, c0);
    if (den < 1e-5f) den = 1.f;
    const float t = __fdiv_rn(__fsub_rn(u, c0), den);
    const float zs = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));
    keys[(use_coarse ? Sc : 0) + j] = zs;
    if (z_new_out) z_new_out[ray * n_fine + j] = zs;
  }
  if (use_coarse)
    for (int i = tid; i < Sc; i += 256) keys[i] = zr[i];
  __syncthreads();
  // In eval mode u is a linspace, so the fine samples come out non-decreasing, and the coarse schedule always is: the sort is then
  // a merge of two sorted runs — every key finds its output slot with one binary search in the other run (stable: coarse keys
  // before equal fine ones) — instead of 36 barrier-separated bitonic stages.  Sortedness is checked, not assumed (training
  // draws random u; rounding may invert neighbours by an ulp): any inversion falls back to the bitonic network.
  const int base_f = use_coarse ? Sc : 0;
  bool inv = false;
  for (int j = tid; j + 1 < n_fine; j += 256) inv |= keys[base_f + j] > keys[base_f + j + 1];
  if (use_coarse)
    for (int i = tid; i + 1 < Sc; i += 256) inv |= keys[i] > keys[i + 1];
  if (!__syncthreads_or(inv)) {
    if (use_coarse) {
      float* outk = cdf;  // the cdf is no longer needed
      for (int i = tid; i < Sc; i += 256) {
        const float v = keys[i];
        int lo = 0, hi = n_fine;  // number of fine keys < v
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[Sc + mid] < v) lo = mid + 1; else hi = mid; }
        outk[i + lo] = v;
      }
      for (int j = tid; j < n_fine; j += 256) {
        const float v = keys[Sc + j];
        int lo = 0, hi = Sc;      // number of coarse keys <= v
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] <= v) lo = mid + 1; else hi = mid; }
        outk[j + lo] = v;
      }
      __syncthreads();
      for (int i = tid; i < n_out; i += 256) z_out[ray * n_out + i] = outk[i];
    } else {
      for (int i = tid; i < n_out; i += 256) z_out[ray * n_out + i] = keys[i];
    }
    return;
  }
  // bitonic sort of n_out keys padded to a power of two with +inf
  int P = 1;
  while (P < n_out) P <<= 1;
  for (int i = n_out + tid; i < P; i += 256) keys[i] = __int_as_float(0x7f800000);
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const float a = keys[i], b = keys[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[l] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < n_out; i += 256) z_out[ray * n_out + i] = keys[i];
}

// The same, one WAVE per ray (four rays per workgroup) for Sc, n_fine <= 256 - every shipped configuration (128 + 128): no
// workgroup barriers (the workgroup form has ~15 on its path and leaves half of its 256 threads without an item at Sc = 128), four
// times as many rays in flight per CU.  Same arithmetic in the same order wherever the order is observable: the cdf is the same
// 64-wide double scan with carry, the inverse CDF and the rank merge are the loops above with a stride of 64.  The one difference is
// the grouping of the double-precision sum of the pdf normaliser (per-lane partials + butterfly instead of 256 partials + tree):
// it is rounded to float once, so the two agree unless the double sum lands within 2^-29 ulp of a float rounding boundary.  When the
// fine samples come out unsorted (random u in training) only THEY are sorted (bitonic in LDS, wave-synchronous) and then merged
// with the coarse run; equal keys are equal values, so the result is the sort's.
constexpr int PDFW_MAX = 256;
__global__ __launch_bounds__(256) void k_sample_pdf_merge_w(const float* __restrict__ z, const float* __restrict__ weight,
                                                            const float* __restrict__ u_in, int64_t N, int Sc, int n_fine,
                                                            int use_coarse, float* __restrict__ z_out,
                                                            float* __restrict__ z_new_out) {
  __shared__ float s_cdf[4][PDFW_MAX];
  __shared__ float s_keys[4][2 * PDFW_MAX];
  __shared__ float s_out[4][2 * PDFW_MAX];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
  if (ray >= N) return;
  float* cdf = s_cdf[wv];
  float* keys = s_keys[wv];
  float* outk = s_out[wv];
  const float* zr = z + ray * Sc;
  const float* wr = weight + ray * Sc;
  const int nb = Sc - 1, nw = Sc - 2;
  const auto wsync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  double part = 0.0;
  for (int i = lane; i < nw; i += 64) part += (double)__fadd_rn(wr[1 + i], 1e-5f);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
  const float total = (float)part;
  double carry = 0.0;
  for (int s0 = 0; s0 < nw; s0 += 64) {
    const int i = s0 + lane;
    double v = (i < nw) ? (double)__fdiv_rn(__fadd_rn(wr[1 + i], 1e-5f), total) : 0.0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double o = __shfl_up(v, d, 64);
      if (lane >= d) v += o;
    }
    if (i < nw) cdf[i + 1] = (float)(carry + v);
    carry += __shfl(v, 63, 64);
  }
  if (lane == 0) cdf[0] = 0.f;
  wsync();
  const int n_out = use_coarse ? Sc + n_fine : n_fine;
  const int base_f = use_coarse ? Sc : 0;
  for (int j = lane; j < n_fine; j += 64) {
    float u;
    if (u_in) u = u_in[ray * n_fine + j];
    else {
      const float step = __fdiv_rn(1.f, (float)(n_fine - 1));
      u = (n_fine == 1) ? 0.f : (j < n_fine / 2 ? __fmul_rn(step, (float)j) : __fsub_rn(1.f, __fmul_rn(step, (float)(n_fine - 1 - j))));
    }
    int lo = 0, hi = nb;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (!(cdf[mid] > u)) lo = mid + 1; else hi = mid;
    }
    const int below = max(lo - 1, 0), above = min(lo, nb - 1);
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = 0.5f * __fadd_rn(zr[below + 1], zr[below]);
    const float b1 = 0.5f * __fadd_rn(zr[above + 1], zr[above]);
    float den = __fsub_rn(c1, c0);
    if (den < 1e-5f) den = 1.f;
    const float t = __fdiv_rn(__fsub_rn(u, c0), den);
    const float zs = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));
    keys[base_f + j] = zs;
    if (z_new_out) z_new_out[ray * n_fine + j] = zs;
  }
  if (use_coarse)
    for (int i = lane; i < Sc; i += 64) keys[i] = zr[i];
  wsync();
  bool inv_f = false, inv_c = false;
  for (int j = lane; j + 1 < n_fine; j += 64) inv_f |= keys[base_f + j] > keys[base_f + j + 1];
  if (use_coarse)
    for (int i = lane; i + 1 < Sc; i += 64) inv_c |= keys[i] > keys[i + 1];
  const bool any_f = __ballot(inv_f) != 0ull, any_c = __ballot(inv_c) != 0ull;
  // sort a run of n keys at keys[base ..) in place: bitonic network over the next power of two (padding +inf lives in outk)
  const auto sort_run = [&](int base, int n) {
    int P = 1;
    while (P < n) P <<= 1;
    float* buf = outk;   // scratch; P <= 2 * PDFW_MAX
    for (int i = lane; i < P; i += 64) buf[i] = i < n ? keys[base + i] : __int_as_float(0x7f800000);
    wsync();
    for (int k = 2; k <= P; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = lane; i < P; i += 64) {
          const int l = i ^ j;
          if (l > i) {
            const float a = buf[i], b = buf[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { buf[i] = b; buf[l] = a; }
          }
        }
        wsync();
      }
    for (int i = lane; i < n; i += 64) keys[base + i] = buf[i];
    wsync();
  };
  if (any_c) {            // never seen (the coarse schedule is monotone); kept for safety: sort everything
    sort_run(0, n_out);
    for (int i = lane; i < n_out; i += 64) z_out[ray * n_out + i] = keys[i];
    return;
  }
  if (any_f) sort_run(base_f, n_fine);
  if (use_coarse) {
    for (int i = lane; i < Sc; i += 64) {
      const float v = keys[i];
      int lo = 0, hi = n_fine;  // number of fine keys < v
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[Sc + mid] < v) lo = mid + 1; else hi = mid; }
      outk[i + lo] = v;
    }
    for (int j = lane; j < n_fine; j += 64) {
      const float v = keys[Sc + j];
      int lo = 0, hi = Sc;      // number of coarse keys <= v
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] <= v) lo = mid + 1; else hi = mid; }
      outk[j + lo] = v;
    }
    wsync();
    for (int i = lane; i < n_out; i += 64) z_out[ray * n_out + i] = outk[i];
  } else {
    for (int i = lane; i < n_out; i += 64) z_out[ray * n_out + i] = keys[i];
  }
}

// =============================================================================================
// update_coarse_sigma_grid — 2x average pooling of a channel-last table    models/EgoNeRF.py:124-131
// =============================================================================================
__global__ void k_avgpool(const float* __restrict__ src, int H, int W, int C, float* __restrict__ dst) {
  const int Ho = H / 2, Wo = (W == 1) ? 1 : W / 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Ho * Wo * C) return;
  const int ch = (int)(idx % C);
  const int x = (int)((idx / C) % Wo);
  const int y = (int)(idx / ((int64_t)C * Wo));
  if (W == 1) {
    dst[idx] = (src[((int64_t)2 * y) * C + ch] + src[((int64_t)2 * y + 1) * C + ch]) * 0.5f;
  } else {
    const float* r0 = src + ((int64_t)(2 * y) * W + 2 * x) * C + ch;
    const float* r1 = src + ((int64_t)(2 * y + 1) * W + 2 * x) * C + ch;
    dst[idx] = (((r0[0] + r0[C]) + r1[0]) + r1[C]) * 0.25f;
  }
}

// all tables of a field in one launch (update_coarse_sigma_grid runs after every training step: 12 launches of ~5 us otherwise)
struct PoolJobs {
  const float* src[12];
  float* dst[12];
  int32_t H[12], W[12];
  int32_t C, n;
};

__global__ void k_avgpool_many(PoolJobs J) {
  const int jb = blockIdx.y;
  const int H = J.H[jb], W = J.W[jb], C = J.C;
  const int Ho = H / 2, Wo = (W == 1) ? 1 : W / 2;
  const float* __restrict__ src = J.src[jb];
  float* __restrict__ dst = J.dst[jb];
  const int64_t total = (int64_t)Ho * Wo * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % C);
    const int x = (int)((idx / C) % Wo);
    const int y = (int)(idx / ((int64_t)C * Wo));
    if (W == 1) {
      dst[idx] = (src[((int64_t)2 * y) * C + ch] + src[((int64_t)2 * y + 1) * C + ch]) * 0.5f;
    } else {
      const float* r0 = src + ((int64_t)(2 * y) * W + 2 * x) * C + ch;
      const float* r1 = src + ((int64_t)(2 * y + 1) * W + 2 * x) * C + ch;
      dst[idx] = (((r0[0] + r0[C]) + r1[0]) + r1[C]) * 0.25f;
    }
  }
}

// =============================================================================================
// Equirectangular camera rays on the device   dataLoader/ray_utils.py:24-40 (directions), :85-113 (pose)
// =============================================================================================
struct Pose34 { float m[12]; };

__global__ void k_erp_rays(int H, int W, int row0, int n_rows, Pose34 c2w, int normalize, float* __restrict__ rays) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_rows * W) return;
  const int col = (int)(idx % W), row = row0 + (int)(idx / W);
  const float i = (float)col + 0.5f, j = (float)row + 0.5f;
  const float phi = __fmul_rn(__fsub_rn(1.f, __fdiv_rn(__fmul_rn(2.f, i), (float)W)), 3.14159265358979323846f);
  const float theta = __fdiv_rn(__fmul_rn(__fsub_rn(1.f, __fdiv_rn(__fmul_rn(2.f, j), (float)H)), 3.14159265358979323846f), 2.f);
  const float ct = cosf(theta);
  float d0 = -ct * sinf(phi), d1 = sinf(theta), d2 = -ct * cosf(phi);
  if (normalize) {  // directions / torch.norm(directions, dim=-1): sqrt of the sum of squares, then three divisions
    const float n = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)));
    d0 = __fdiv_rn(d0, n); d1 = __fdiv_rn(d1, n); d2 = __fdiv_rn(d2, n);
  }
  float* o = rays + idx * 6;
  o[0] = c2w.m[3]; o[1] = c2w.m[7]; o[2] = c2w.m[11];
#pragma unroll
  for (int r = 0; r < 3; ++r) o[3 + r] = (d0 * c2w.m[4 * r] + d1 * c2w.m[4 * r + 1]) + d2 * c2w.m[4 * r + 2];
}

// =============================================================================================
// C ABI
// =============================================================================================
static inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

// roctx hooks of EGO_TRACE (ego_host.h): resolved once, on the first traced call
const EgoRoctx* ego_roctx() {
  static const EgoRoctx* const hooks = []() -> const EgoRoctx* {
    const char* e = getenv("EGO_ROCTX");
    if (!e || !*e || *e == '0') return nullptr;
    static EgoRoctx h{};
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* so = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
      if (!so) continue;
      h.push = (int (*)(const char*))dlsym(so, "roctxRangePushA");
      h.pop = (int (*)())dlsym(so, "roctxRangePop");
      if (h.push && h.pop) return &h;
    }
    fprintf(stderr, "libegonerf_hip: EGO_ROCTX is set but no roctx library could be loaded; tracing ranges are off\n");
    return nullptr;
  }();
  return hooks;
}

extern "C" {

int ego_abi_version(void) { return EGO_ABI_VERSION; }
const char* ego_last_error(void) { return ego_err_buf(); }

int ego_sample_ray_exp(const float* rays, const float* r_sched, const float* jitter, float near_, int64_t N, int32_t S,
                       float* xyz, float* z, void* stream) {
  EGO_TRACE("ego_sample_ray_exp");
  EGO_REQUIRE(rays && r_sched && N >= 0 && S >= 2, "sample_ray_exp: null input or S < 2");
  if (N == 0) return EGO_OK;
  k_sample_ray_exp<<<nblk(N * S, 256), 256, 0, (hipStream_t)stream>>>(rays, r_sched, jitter, near_, N, S, xyz, z);
  return ego_launch_status("k_sample_ray_exp");
}

int ego_erp_rays(int32_t H, int32_t W, int32_t row0, int32_t n_rows, const float* c2w, int32_t normalize, float* rays,
                 void* stream) {
  EGO_TRACE("ego_erp_rays");
  EGO_REQUIRE(H >= 1 && W >= 1 && row0 >= 0 && n_rows >= 0 && row0 + n_rows <= H, "erp_rays: bad image window");
  if (n_rows == 0) return EGO_OK;
  EGO_REQUIRE(c2w && rays, "erp_rays: null argument");
  Pose34 p;
  for (int i = 0; i < 12; ++i) p.m[i] = c2w[i];
  k_erp_rays<<<nblk((int64_t)n_rows * W, 256), 256, 0, (hipStream_t)stream>>>(H, W, row0, n_rows, p, normalize, rays);
  return ego_launch_status("k_erp_rays");
}

// ---- device -> mapped host memory, by a kernel of a chosen (small) footprint --------------------------------------------------------
struct CopyOutArgs {
  const float* src[EGO_COPY_OUT_MAX];
  float* dst[EGO_COPY_OUT_MAX];
  int64_t n[EGO_COPY_OUT_MAX];   // floats
  int32_t count;
};

__global__ __launch_bounds__(256) void k_copy_out(CopyOutArgs A) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  for (int t = 0; t < A.count; ++t) {
    const float* __restrict__ s = A.src[t];
    float* __restrict__ d = A.dst[t];
    const int64_t n = A.n[t];
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
      const int64_t n4 = n >> 2;
      for (int64_t i = tid; i < n4; i += nth) __builtin_nontemporal_store(((const f32x4*)s)[i], (f32x4*)d + i);
      for (int64_t i = (n4 << 2) + tid; i < n; i += nth) d[i] = s[i];
    } else {
      for (int64_t i = tid; i < n; i += nth) d[i] = s[i];
    }
  }
}

int ego_copy_out(int32_t count, const float* const* src, float* const* dst, const int64_t* n_floats, int32_t workgroups, void* stream) {
  EGO_TRACE("ego_copy_out");
  EGO_REQUIRE(count >= 0 && count <= EGO_COPY_OUT_MAX && workgroups >= 1 && workgroups <= 65535, "copy_out: count / workgroups out of range");
  if (count == 0) return EGO_OK;
  EGO_REQUIRE(src && dst && n_floats, "copy_out: null argument");
  CopyOutArgs a{};
  int64_t total = 0;
  for (int t = 0; t < count; ++t) {
    EGO_REQUIRE(n_floats[t] >= 0 && (n_floats[t] == 0 || (src[t] && dst[t])), "copy_out: null buffer or negative size");
    a.src[t] = src[t]; a.dst[t] = dst[t]; a.n[t] = n_floats[t];
    total += n_floats[t];
  }
  a.count = count;
  if (total == 0) return EGO_OK;
  k_copy_out<<<workgroups, 256, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_copy_out");
}

int ego_from_cartesian(const ego_scene* sc, const float* xyz, int64_t M, float* c7, void* stream) {
  EGO_TRACE("ego_from_cartesian");
  EGO_REQUIRE(M >= 0, "from_cartesian: M < 0");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(sc && xyz && c7, "from_cartesian: null argument");
  k_from_cartesian<<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_coords(*sc), xyz, M, c7);
  return ego_launch_status("k_from_cartesian");
}

int ego_normalize_coord(const ego_scene* sc, const float* c7, int64_t M, float* c7n, void* stream) {
  EGO_TRACE("ego_normalize_coord");
  EGO_REQUIRE(M >= 0, "normalize_coord: M < 0");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(sc && c7 && c7n && sc->r_lut, "normalize_coord: null argument");
  k_normalize_coord<<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_coords(*sc), c7, M, c7n);
  return ego_launch_status("k_normalize_coord");
}

static DevOcc make_occ(const ego_scene& sc, int coarse) {
  DevOcc o;
  o.vol = coarse ? nullptr : sc.occ;  // the coarse (proposal) pass always sees the full field
  o.res[0] = sc.occ_res[0]; o.res[1] = sc.occ_res[1]; o.res[2] = sc.occ_res[2];
  o.cell = (o.vol && o.res[0] >= 2 && o.res[1] >= 2 && o.res[2] >= 2) ? sc.occ_cell : nullptr;
  return o;
}

static int check_field(const ego_vm_field& f, const char* what) {
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i)
      if (!f.plane[g][i] || !f.line[g][i]) return ego_fail(EGO_E_BADARG, "%s: null table pointer", what);
  if (f.res[0] < 2 || f.res[1] < 2 || f.res[2] < 2) return ego_fail(EGO_E_BADARG, "%s: resolution < 2", what);
  if (!ego_field_is_compact(f, 4))
    return ego_fail(EGO_E_BADARG, "%s: the 12 tables of a field must lie within 4 GB of each other (allocate them from one buffer): "
                                  "the gathers address taps as base + 32-bit offset", what);
  return EGO_OK;
}

int ego_density_feature(const ego_scene* sc, const float* c7n, int64_t M, int32_t coarse, float* out, void* stream) {
  EGO_TRACE("ego_density_feature");
  EGO_REQUIRE(M >= 0, "density_feature: M < 0");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(sc && c7n && out, "density_feature: null argument");
  const ego_vm_field& f = coarse ? sc->density_coarse : sc->density;
  if (int e = check_field(f, "density_feature")) return e;
  if (f.n_comp == 16)
    k_density_feature<16><<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_field(f), c7n, M, out);
  else if (f.n_comp == 8)
    k_density_feature<8><<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_field(f), c7n, M, out);
  else if (f.n_comp == 4)
    k_density_feature<4><<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_field(f), c7n, M, out);
  else if (f.n_comp == 12)
    k_density_feature<12><<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_field(f), c7n, M, out);
  else if (f.n_comp == 24)
    k_density_feature<24><<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_field(f), c7n, M, out);
  else if (f.n_comp == 32)
    k_density_feature<32><<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_field(f), c7n, M, out);
  else if (f.n_comp == 48)
    k_density_feature<48><<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_field(f), c7n, M, out);
  else
    return ego_fail(EGO_E_UNSUPPORTED, "density_feature: n_comp %d (supported: 4, 8, 12, 16, 24, 32, 48)", f.n_comp);
  return ego_launch_status("k_density_feature");
}

int ego_feature2density(const ego_scene* sc, const float* feat, int64_t M, float* sigma, void* stream) {
  EGO_TRACE("ego_feature2density");
  EGO_REQUIRE(M >= 0, "feature2density: M < 0");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(sc && feat && sigma, "feature2density: null argument");
  k_feature2density<<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(feat, M, sc->act_softplus, sc->density_shift, sigma);
  return ego_launch_status("k_feature2density");
}

int ego_raw2alpha(const float* sigma, const float* dist, int64_t N, int32_t S, float* alpha, float* weight,
                  float* bg_weight, void* stream) {
  EGO_TRACE("ego_raw2alpha");
  EGO_REQUIRE(N >= 0 && S >= 1, "raw2alpha: bad size");
  if (N == 0) return EGO_OK;
  EGO_REQUIRE(sigma && dist, "raw2alpha: null argument");
  k_raw2alpha<<<nblk(N, 4), 256, 0, (hipStream_t)stream>>>(sigma, dist, N, S, alpha, weight, bg_weight);
  return ego_launch_status("k_raw2alpha");
}

int ego_sample_pdf_merge(const float* z, const float* weight, const float* u, int64_t N, int32_t Sc, int32_t n_fine,
                         int32_t use_coarse, float* z_out, float* z_new_out, void* stream) {
  EGO_TRACE("ego_sample_pdf_merge");
  EGO_REQUIRE(z && weight && z_out && N >= 0, "sample_pdf_merge: null argument");
  EGO_REQUIRE(Sc >= 3 && n_fine >= 1 && Sc + n_fine <= PDF_MAX, "sample_pdf_merge: need 3 <= Sc, Sc + n_fine <= 2048");
  if (N == 0) return EGO_OK;
  if (Sc <= PDFW_MAX && n_fine <= PDFW_MAX)   // every shipped configuration: one wave per ray
    k_sample_pdf_merge_w<<<(unsigned)((N + 3) / 4), 256, 0, (hipStream_t)stream>>>(z, weight, u, N, Sc, n_fine, use_coarse, z_out, z_new_out);
  else
    k_sample_pdf_merge<<<(unsigned)N, 256, 0, (hipStream_t)stream>>>(z, weight, u, Sc, n_fine, use_coarse, z_out, z_new_out);
  return ego_launch_status("k_sample_pdf_merge");
}

int ego_envmap_radiance(const ego_scene* sc, const float* dirs, int64_t N, float* out, void* stream) {
  EGO_TRACE("ego_envmap_radiance");
  EGO_REQUIRE(N >= 0, "envmap_radiance: N < 0");
  if (N == 0) return EGO_OK;
  EGO_REQUIRE(sc && dirs && out && sc->envmap && sc->envmap_h >= 2, "envmap_radiance: no envmap / null argument");
  k_envmap<<<nblk(N, 256), 256, 0, (hipStream_t)stream>>>(sc->envmap, sc->envmap_h, dirs, N, out);
  return ego_launch_status("k_envmap");
}

int ego_envmap_backward(const ego_scene* sc, const float* dirs, int32_t dir_stride, const float* g_rgb, const float* rgb_raw, const float* bg_weight,
                        const float* env_map, int64_t N, float* g_emission, void* stream) {
  EGO_TRACE("ego_envmap_backward");
  EGO_REQUIRE(N >= 0, "envmap_backward: N < 0");
  if (N == 0) return EGO_OK;
  EGO_REQUIRE(sc && dirs && dir_stride >= 3 && g_rgb && rgb_raw && bg_weight && env_map && g_emission && sc->envmap_h >= 2,
              "envmap_backward: no envmap / null argument");
  k_envmap_bwd<<<nblk(N, 256), 256, 0, (hipStream_t)stream>>>(sc->envmap_h, dirs, dir_stride, g_rgb, rgb_raw, bg_weight, env_map, N, g_emission);
  return ego_launch_status("k_envmap_bwd");
}

int ego_sh_render(const float* viewdirs, const float* features, int64_t M, float* rgb, void* stream) {
  EGO_TRACE("ego_sh_render");
  EGO_REQUIRE(M >= 0, "sh_render: M < 0");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(viewdirs && features && rgb, "sh_render: null argument");
  k_sh_render<<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(viewdirs, features, M, rgb);
  return ego_launch_status("k_sh_render");
}

int ego_alpha_mask_sample(const ego_scene* sc, const float* c7n, int64_t M, float* out, void* stream) {
  EGO_TRACE("ego_alpha_mask_sample");
  EGO_REQUIRE(M >= 0, "alpha_mask_sample: M < 0");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(sc && c7n && out && sc->occ && sc->occ_res[0] >= 2 && sc->occ_res[1] >= 2 && sc->occ_res[2] >= 2,
              "alpha_mask_sample: no occupancy volume / null argument");
  k_alpha_mask_sample<<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_occ(*sc, 0), c7n, M, out);
  return ego_launch_status("k_alpha_mask_sample");
}

int ego_avgpool_table(const float* src, int32_t H, int32_t W, int32_t C, float* dst, void* stream) {
  EGO_TRACE("ego_avgpool_table");
  EGO_REQUIRE(src && dst && H >= 2 && W >= 1 && C >= 1, "avgpool_table: bad argument");
  const int64_t n = (int64_t)(H / 2) * (W == 1 ? 1 : W / 2) * C;
  k_avgpool<<<nblk(n, 256), 256, 0, (hipStream_t)stream>>>(src, H, W, C, dst);
  return ego_launch_status("k_avgpool");
}

int ego_avgpool_field(const ego_vm_field* src, const ego_vm_field* dst, void* stream) {
  EGO_TRACE("ego_avgpool_field");
  EGO_REQUIRE(src && dst && src->n_comp >= 1 && dst->n_comp == src->n_comp, "avgpool_field: null field or component counts differ");
  PoolJobs J{};
  J.C = src->n_comp; J.n = 12;
  int64_t most = 0;
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i) {
      // plane i: [res[y axis]][res[x axis]][C]; line i: [res[line axis]][C] (vm_plane_x / vm_plane_y / vm_line_ax of ego_device.h)
      const int px = i == 2 ? 1 : 0, py = i == 0 ? 1 : 2, la = 2 - i;
      EGO_REQUIRE(src->plane[g][i] && src->line[g][i] && dst->plane[g][i] && dst->line[g][i], "avgpool_field: null table");
      EGO_REQUIRE(dst->res[0] == src->res[0] / 2 && dst->res[1] == src->res[1] / 2 && dst->res[2] == src->res[2] / 2 && dst->res[0] >= 1 &&
                  dst->res[1] >= 1 && dst->res[2] >= 1, "avgpool_field: dst.res must be src.res / 2");
      const int a = g * 6 + i, b = g * 6 + 3 + i;
      J.src[a] = src->plane[g][i]; J.dst[a] = (float*)dst->plane[g][i]; J.H[a] = src->res[py]; J.W[a] = src->res[px];
      J.src[b] = src->line[g][i]; J.dst[b] = (float*)dst->line[g][i]; J.H[b] = src->res[la]; J.W[b] = 1;
      const int64_t n = (int64_t)(J.H[a] / 2) * (J.W[a] / 2) * J.C;
      most = n > most ? n : most;
    }
  const unsigned bx = nblk(most, 256) < 1024u ? nblk(most, 256) : 1024u;
  k_avgpool_many<<<dim3(bx ? bx : 1u, 12), 256, 0, (hipStream_t)stream>>>(J);
  return ego_launch_status("k_avgpool_many");
}

int ego_march_density(const ego_scene* sc, const float* rays, int64_t N, int32_t S, const float* z_in,
                      const float* r_sched, const float* jitter, float near_, int32_t coarse, float* z_out,
                      float* alpha, int32_t alpha_stride, float* weight, float* bg_weight, float* coords_out, float* sigma_out,
                      uint8_t* tile_active, void* stream) {
  EGO_TRACE("ego_march_density");
  EGO_REQUIRE(sc && rays && N >= 0 && S >= 2, "march_density: null argument or S < 2");
  if (alpha_stride == 0) alpha_stride = S;
  EGO_REQUIRE(alpha_stride >= S && alpha_stride <= S + 64, "march_density: alpha_stride must be in [S, S+64]");
  EGO_REQUIRE(z_in || r_sched, "march_density: need z_in or r_sched");
  EGO_REQUIRE(sc->r_lut && sc->n_r_lut >= 2 && sc->n_r_lut <= 1024, "march_density: r_lut missing or > 1024 entries");
  const ego_vm_field& f = (coarse & 1) ? sc->density_coarse : sc->density;
  if (int e = check_field(f, "march_density")) return e;
  if (N == 0) return EGO_OK;
  const DevOcc o = make_occ(*sc, coarse & 1);
  if (f.n_comp != 16)   // any other component count: the compatibility kernel (csrc/ego_generic.hip)
    return ego_generic_march(sc, f, (coarse & 2) != 0, rays, N, S, z_in, r_sched, jitter, near_, o.vol, z_out, alpha, alpha_stride, weight, bg_weight,
                             coords_out, sigma_out, tile_active, stream);
  // Rays split over two waves when one wave per ray would leave the launch quantised by wave slots (see k_march_density): fewer than
  // two rounds of slots and at least two passes per half (at most MARCH_MAXP in the deferred half).
  const int n_pass = (S + 63) / 64;
  int nsplit = 1;
  {
    // wave slots (3 per SIMD) of the CURRENT device, queried once per device; racing first calls store the same value
    static std::atomic<int> slots_of[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    int slots = slots_of[dev].load(std::memory_order_relaxed);
    if (!slots) {
      int cus = 0;
      slots = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus * 12 : 3072;
      slots_of[dev].store(slots, std::memory_order_relaxed);
    }
    if ((double)N / slots < 2.0 && n_pass >= 4 && (n_pass + 1) / 2 <= MARCH_MAXP) nsplit = 2;
    // experiments and tests/test_hip_march_split.py (which flips it between calls of one process, hence read per launch: a getenv is
    // ~50 ns against a 5 us launch): EGO_MARCH_SPLIT forces 1 or 2 where the shape allows it
    if (const char* e = getenv("EGO_MARCH_SPLIT")) {
      const int forced = atoi(e);
      if (forced == 1 || (forced == 2 && n_pass >= 2 && (n_pass + 1) / 2 <= MARCH_MAXP)) nsplit = forced;
    }
  }
#define EGO_LAUNCH_MARCH(OCC, NS)                                                                                                       \
  k_march_density<16, OCC, NS><<<nblk(N, 4 / NS), 256, 0, (hipStream_t)stream>>>(                                                      \
      make_coords(*sc, (coarse & 2) != 0), make_field(f), rays, N, S, z_in, r_sched, jitter, near_, sc->act_softplus, sc->density_shift, \
      sc->distance_scale, z_out, alpha, alpha_stride, weight, bg_weight, coords_out, sigma_out, o, sc->term_eps,                          \
      fmaxf(sc->weight_thres, 0.f), tile_active)
  if (o.vol) {
    if (nsplit == 2) EGO_LAUNCH_MARCH(true, 2); else EGO_LAUNCH_MARCH(true, 1);
  } else {
    if (nsplit == 2) EGO_LAUNCH_MARCH(false, 2); else EGO_LAUNCH_MARCH(false, 1);
  }
#undef EGO_LAUNCH_MARCH
  return ego_launch_status("k_march_density");
}

int ego_composite(const ego_scene* sc, const float* rays, const float* z, const float* weight, const float* bg_weight,
                  const float* rgb, int64_t N, int32_t S, float* rgb_map, float* depth, float* bg_map, float* env_map,
                  float* rgb_raw, void* stream) {
  EGO_TRACE("ego_composite");
  EGO_REQUIRE(sc && rays && z && weight && rgb && rgb_map && N >= 0 && S >= 1, "composite: null argument");
  EGO_REQUIRE(!sc->envmap || bg_weight, "composite: envmap needs bg_weight");
  if (N == 0) return EGO_OK;
  k_composite<<<nblk(N, 4), 256, 0, (hipStream_t)stream>>>(sc->envmap, sc->envmap_h, rays, z, weight, bg_weight, rgb, N, S,
                                                          rgb_map, depth, bg_map, env_map, rgb_raw, fmaxf(sc->weight_thres, 0.f));
  return ego_launch_status("k_composite");
}

}  // extern "C"
