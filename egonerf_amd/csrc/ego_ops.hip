// libegonerf_hip.so, part 1: the separately callable stages (rows A-E, I, J of SURVEY 8a) and the
// fused marching / compositing kernels.  gfx950 only.
#include "ego_device.h"
#include "ego_host.h"

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

// Density feature by 4-lane teams (C == 16: one texel = one 64-B line = the team's four float4 parts): lane = 16 p + s serves
// sample slot s with part p, so a load instruction touches 16 whole lines instead of 64 quarter-used ones.  Returns the
// feature of the team's sample in all four of its lanes (relu per plane, EgoNeRF.py:340,346).
__device__ __forceinline__ float density_team16(const DevField& F, int g, float a_r, float a_th, float a_ph, int p) {
#pragma clang fp contract(fast)
  constexpr int C = 16;
  const VMTaps t = vm_setup(a_r, a_th, a_ph, F.res);
  float feat = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const Lin1 X = t.ax[vm_plane_x(i)], Y = t.ax[vm_plane_y(i)], Ln = t.ax[vm_line_ax(i)];
    const int W = F.res[vm_plane_x(i)];
    // compact addressing (DevField): scalar base + 32-bit byte offset per tap (a table set spans < 4 GB)
    const uint32_t pb = (g ? F.poff[1][i] : F.poff[0][i]) + 16u * (uint32_t)p;
    const uint32_t lb = (g ? F.loff[1][i] : F.loff[0][i]) + 16u * (uint32_t)p;
    const uint32_t r0 = pb + (uint32_t)(Y.i0 * W) * (C * 4), r1 = pb + (uint32_t)(Y.i1 * W) * (C * 4);
    const uint32_t c0 = (uint32_t)X.i0 * (C * 4), c1 = (uint32_t)X.i1 * (C * 4);
    const f32x4 t00 = *(const f32x4*)(F.base + (r0 + c0)), t01 = *(const f32x4*)(F.base + (r0 + c1));
    const f32x4 t10 = *(const f32x4*)(F.base + (r1 + c0)), t11 = *(const f32x4*)(F.base + (r1 + c1));
    const f32x4 u0 = *(const f32x4*)(F.base + (lb + (uint32_t)Ln.i0 * (C * 4))), u1 = *(const f32x4*)(F.base + (lb + (uint32_t)Ln.i1 * (C * 4)));
    const float w00 = __fmul_rn(Y.w0, X.w0), w01 = __fmul_rn(Y.w0, X.w1);
    const float w10 = __fmul_rn(Y.w1, X.w0), w11 = __fmul_rn(Y.w1, X.w1);
    const f32x4 pv = t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11;
    const f32x4 lv = u0 * Ln.w0 + u1 * Ln.w1;
    const f32x4 m = pv * lv;
    float dot = (m.x + m.y) + (m.z + m.w);
    dot += __shfl_xor(dot, 1, 64);   // the team's four lanes are neighbours: two quad-local exchanges
    dot += __shfl_xor(dot, 2, 64);
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
    const float a = ok ? __fsub_rn(1.f, expf(-sigma[ray * S + s] * dist[ray * S + s])) : 0.f;
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
// =============================================================================================
template <int C>
__global__ __launch_bounds__(256) void k_march_density(DevCoords c, DevField F, const float* __restrict__ rays,
                                                       int64_t N, int S, const float* __restrict__ z_in,
                                                       const float* __restrict__ r_sched,
                                                       const float* __restrict__ jitter, float near_,
                                                       int softplus, float shift, float dscale,
                                                       float* __restrict__ z_out, float* __restrict__ alpha,
                                                       int alpha_stride, float* __restrict__ weight,
                                                       float* __restrict__ bg, float* __restrict__ coords_out,
                                                       float* __restrict__ sigma_out, DevOcc occ, float term_eps,
                                                       float shade_above, uint8_t* __restrict__ tile_active) {
  __shared__ float lut[1024];
  for (int i = threadIdx.x; i < c.n_lut; i += blockDim.x) lut[i] = c.r_lut[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (ray >= N) return;
  const float* R = rays + ray * 6;
  const float ox = R[0], oy = R[1], oz = R[2], dx = R[3], dy = R[4], dz = R[5];
  float carry = 1.f;
  // Exact early termination: once the transmittance in front of a pass is exactly 0 (fp32 underflow behind opaque samples), every
  // remaining weight is a * 0 = 0 and bg stays 0, so the rest of the ray only needs its distances and zero weights - unless the
  // caller wants per-sample alpha / sigma, which are independent of what lies in front.
  const bool may_stop = !alpha && !sigma_out;
  for (int s0 = 0; s0 < S; s0 += 64) {
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
    const float dist = (s < S - 1) ? __fsub_rn(zn, z) : __fsub_rn(z, zn);
    const float px = __fadd_rn(ox, __fmul_rn(dx, z)), py = __fadd_rn(oy, __fmul_rn(dy, z)),
                pz = __fadd_rn(oz, __fmul_rn(dz, z));
    const YinYang y = yinyang_from_xyz(px, py, pz, c);
    const float a_r = normalize_r(y.r, lut, c.n_lut, c.n_r);
    const float a_th = normalize_ang(y.th, c.th_near, c.th_inv);
    const float a_ph = normalize_ang(y.ph, c.ph_near, c.ph_inv);
    // occupancy mask (opt-in): unoccupied samples keep sigma = 0 and skip the 18-tap gather
    const bool occupied = !occ.vol || occ_sample(occ, y.yang, a_r, a_th, a_ph) > 0.f;
    float sg = 0.f;
    if (C == 16) {
      // four rounds of 16 samples: round rd serves samples 16 rd .. 16 rd + 15 of this pass; lane 16 p + s keeps round p's result
      float f = 0.f;
      const int yg = y.yang ? 1 : 0;
#pragma unroll
      for (int rd = 0; rd < 4; ++rd) {
        // team = four NEIGHBOURING lanes (4 t .. 4 t + 3) serving sample 16 rd + t with parts 0..3: the 16 lanes of a row read 4
        // lines (not 16 quarter-lines), consecutive samples of the ray - which mostly share their texel lines - sit in one row,
        // and the channel reduction is quad-local (teams of lanes 16 apart: 0.109 vs 0.105 ms, removed)
        const int src = 16 * rd + (lane >> 2);
        const float tr = __shfl(a_r, src, 64), tt = __shfl(a_th, src, 64), tp = __shfl(a_ph, src, 64);
        const int tg = __shfl(yg, src, 64);
        const bool tocc = __shfl((int)occupied, src, 64) != 0;
        float d = 0.f;
        if (__ballot(tocc) != 0ull) d = density_team16(F, tg, tr, tt, tp, lane & 3);
        const float got = __shfl(d, 4 * (lane & 15), 64);   // sample 16 rd + (lane & 15) was served by team lane & 15
        if ((lane >> 4) == rd) f = got;
      }
      if (occupied) sg = softplus ? softplus_shift(f, shift) : fmaxf(f, 0.f);
    } else if (occupied) {
      const float f = density_lookup<C>(F, y.yang, a_r, a_th, a_ph);
      sg = softplus ? softplus_shift(f, shift) : fmaxf(f, 0.f);
    }
    const float a = ok ? __fsub_rn(1.f, expf(-sg * __fmul_rn(dist, dscale))) : 0.f;
    const float t = ok ? __fadd_rn(__fsub_rn(1.f, a), 1e-10f) : 1.f;
    const float inc = wave_scan_mul(t, lane);
    float exc = __shfl_up(inc, 1, 64);
    if (lane == 0) exc = 1.f;
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
    if (ok) {
      const int64_t o = ray * S + s;
      if (z_out) z_out[o] = z;
      if (coords_out) ((f32x4*)coords_out)[o] = f32x4{a_r, a_th, a_ph, y.yang ? 1.f : 0.f};
      if (sigma_out) sigma_out[o] = sg;
      if (alpha) alpha[ray * alpha_stride + s] = a;
      if (weight) weight[o] = wgt;
    }
    carry *= __shfl(inc, 63, 64);
  }
  // with an environment map the reference appends a column of ones to alpha (EgoNeRF.py:587)
  if (alpha && lane < alpha_stride - S) alpha[ray * alpha_stride + S + lane] = 1.f;
  if (bg && lane == 0) bg[ray] = carry;
}

// =============================================================================================
// Row J — environment map     models/envmap.py:6-14, 26-34
// =============================================================================================
__device__ __forceinline__ void envmap_lookup(const float* __restrict__ em, int h, float dx, float dy, float dz,
                                              float out[3]) {
  const float nrm = fmaxf(__fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz))), 1e-12f);
  const float nx = __fdiv_rn(dx, nrm), ny = __fdiv_rn(dy, nrm), nz = __fdiv_rn(dz, nrm);
  const float u = __fmul_rn(__fadd_rn(nz, 1.f), 0.5f);
  const float v = __fdiv_rn(__fadd_rn(atan2f(ny, nx), 3.14159265358979323846f), 6.28318530717958647692f);
  const Lin1 X = lin_setup(__fsub_rn(__fmul_rn(u, 2.f), 1.f), h);       // u indexes the h-wide axis
  const Lin1 Y = lin_setup(__fsub_rn(__fmul_rn(v, 2.f), 1.f), 2 * h);   // v indexes the 2h axis
  const float w00 = __fmul_rn(Y.w0, X.w0), w01 = __fmul_rn(Y.w0, X.w1), w10 = __fmul_rn(Y.w1, X.w0),
              w11 = __fmul_rn(Y.w1, X.w1);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float* E = em + (int64_t)ch * 2 * h * h;
    const float s = E[(int64_t)Y.i0 * h + X.i0] * w00 + E[(int64_t)Y.i0 * h + X.i1] * w01 +
                    E[(int64_t)Y.i1 * h + X.i0] * w10 + E[(int64_t)Y.i1 * h + X.i1] * w11;
    out[ch] = sigmoidf(s);
  }
}

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
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (ray >= N) return;
  float acc = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, dp = 0.f;
  for (int s = lane; s < S; s += 64) {
    const int64_t o = ray * S + s;
    const float w = weight[o];
    acc += w;
    dp += w * z[o];
    // colour only from samples above the threshold (tensorBase.py:482-487; 0 without one: weights are >= 0, and tiles
    // skipped by ego_shade — mask / early termination / all below the threshold — never wrote their rgb)
    if (w > shade_above) {
      cr += w * rgb[o * 3];
      cg += w * rgb[o * 3 + 1];
      cb += w * rgb[o * 3 + 2];
    }
  }
  acc = wave_sum(acc); cr = wave_sum(cr); cg = wave_sum(cg); cb = wave_sum(cb); dp = wave_sum(dp);
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

extern "C" {

int ego_abi_version(void) { return EGO_ABI_VERSION; }
const char* ego_last_error(void) { return ego_err_buf(); }

int ego_sample_ray_exp(const float* rays, const float* r_sched, const float* jitter, float near_, int64_t N, int32_t S,
                       float* xyz, float* z, void* stream) {
  EGO_REQUIRE(rays && r_sched && N >= 0 && S >= 2, "sample_ray_exp: null input or S < 2");
  if (N == 0) return EGO_OK;
  k_sample_ray_exp<<<nblk(N * S, 256), 256, 0, (hipStream_t)stream>>>(rays, r_sched, jitter, near_, N, S, xyz, z);
  return ego_launch_status("k_sample_ray_exp");
}

int ego_erp_rays(int32_t H, int32_t W, int32_t row0, int32_t n_rows, const float* c2w, int32_t normalize, float* rays,
                 void* stream) {
  EGO_REQUIRE(H >= 1 && W >= 1 && row0 >= 0 && n_rows >= 0 && row0 + n_rows <= H, "erp_rays: bad image window");
  if (n_rows == 0) return EGO_OK;
  EGO_REQUIRE(c2w && rays, "erp_rays: null argument");
  Pose34 p;
  for (int i = 0; i < 12; ++i) p.m[i] = c2w[i];
  k_erp_rays<<<nblk((int64_t)n_rows * W, 256), 256, 0, (hipStream_t)stream>>>(H, W, row0, n_rows, p, normalize, rays);
  return ego_launch_status("k_erp_rays");
}

int ego_from_cartesian(const ego_scene* sc, const float* xyz, int64_t M, float* c7, void* stream) {
  EGO_REQUIRE(M >= 0, "from_cartesian: M < 0");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(sc && xyz && c7, "from_cartesian: null argument");
  k_from_cartesian<<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_coords(*sc), xyz, M, c7);
  return ego_launch_status("k_from_cartesian");
}

int ego_normalize_coord(const ego_scene* sc, const float* c7, int64_t M, float* c7n, void* stream) {
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
  EGO_REQUIRE(M >= 0, "density_feature: M < 0");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(sc && c7n && out, "density_feature: null argument");
  const ego_vm_field& f = coarse ? sc->density_coarse : sc->density;
  if (int e = check_field(f, "density_feature")) return e;
  if (f.n_comp == 16)
    k_density_feature<16><<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_field(f), c7n, M, out);
  else if (f.n_comp == 8)
    k_density_feature<8><<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_field(f), c7n, M, out);
  else
    return ego_fail(EGO_E_UNSUPPORTED, "density_feature: n_comp %d (supported: 8, 16)", f.n_comp);
  return ego_launch_status("k_density_feature");
}

int ego_feature2density(const ego_scene* sc, const float* feat, int64_t M, float* sigma, void* stream) {
  EGO_REQUIRE(M >= 0, "feature2density: M < 0");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(sc && feat && sigma, "feature2density: null argument");
  k_feature2density<<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(feat, M, sc->act_softplus, sc->density_shift, sigma);
  return ego_launch_status("k_feature2density");
}

int ego_raw2alpha(const float* sigma, const float* dist, int64_t N, int32_t S, float* alpha, float* weight,
                  float* bg_weight, void* stream) {
  EGO_REQUIRE(N >= 0 && S >= 1, "raw2alpha: bad size");
  if (N == 0) return EGO_OK;
  EGO_REQUIRE(sigma && dist, "raw2alpha: null argument");
  k_raw2alpha<<<nblk(N, 4), 256, 0, (hipStream_t)stream>>>(sigma, dist, N, S, alpha, weight, bg_weight);
  return ego_launch_status("k_raw2alpha");
}

int ego_sample_pdf_merge(const float* z, const float* weight, const float* u, int64_t N, int32_t Sc, int32_t n_fine,
                         int32_t use_coarse, float* z_out, float* z_new_out, void* stream) {
  EGO_REQUIRE(z && weight && z_out && N >= 0, "sample_pdf_merge: null argument");
  EGO_REQUIRE(Sc >= 3 && n_fine >= 1 && Sc + n_fine <= PDF_MAX, "sample_pdf_merge: need 3 <= Sc, Sc + n_fine <= 2048");
  if (N == 0) return EGO_OK;
  k_sample_pdf_merge<<<(unsigned)N, 256, 0, (hipStream_t)stream>>>(z, weight, u, Sc, n_fine, use_coarse, z_out, z_new_out);
  return ego_launch_status("k_sample_pdf_merge");
}

int ego_envmap_radiance(const ego_scene* sc, const float* dirs, int64_t N, float* out, void* stream) {
  EGO_REQUIRE(N >= 0, "envmap_radiance: N < 0");
  if (N == 0) return EGO_OK;
  EGO_REQUIRE(sc && dirs && out && sc->envmap && sc->envmap_h >= 2, "envmap_radiance: no envmap / null argument");
  k_envmap<<<nblk(N, 256), 256, 0, (hipStream_t)stream>>>(sc->envmap, sc->envmap_h, dirs, N, out);
  return ego_launch_status("k_envmap");
}

int ego_envmap_backward(const ego_scene* sc, const float* dirs, int32_t dir_stride, const float* g_rgb, const float* rgb_raw, const float* bg_weight,
                        const float* env_map, int64_t N, float* g_emission, void* stream) {
  EGO_REQUIRE(N >= 0, "envmap_backward: N < 0");
  if (N == 0) return EGO_OK;
  EGO_REQUIRE(sc && dirs && dir_stride >= 3 && g_rgb && rgb_raw && bg_weight && env_map && g_emission && sc->envmap_h >= 2,
              "envmap_backward: no envmap / null argument");
  k_envmap_bwd<<<nblk(N, 256), 256, 0, (hipStream_t)stream>>>(sc->envmap_h, dirs, dir_stride, g_rgb, rgb_raw, bg_weight, env_map, N, g_emission);
  return ego_launch_status("k_envmap_bwd");
}

int ego_sh_render(const float* viewdirs, const float* features, int64_t M, float* rgb, void* stream) {
  EGO_REQUIRE(M >= 0, "sh_render: M < 0");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(viewdirs && features && rgb, "sh_render: null argument");
  k_sh_render<<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(viewdirs, features, M, rgb);
  return ego_launch_status("k_sh_render");
}

int ego_alpha_mask_sample(const ego_scene* sc, const float* c7n, int64_t M, float* out, void* stream) {
  EGO_REQUIRE(M >= 0, "alpha_mask_sample: M < 0");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(sc && c7n && out && sc->occ && sc->occ_res[0] >= 2 && sc->occ_res[1] >= 2 && sc->occ_res[2] >= 2,
              "alpha_mask_sample: no occupancy volume / null argument");
  k_alpha_mask_sample<<<nblk(M, 256), 256, 0, (hipStream_t)stream>>>(make_occ(*sc, 0), c7n, M, out);
  return ego_launch_status("k_alpha_mask_sample");
}

int ego_avgpool_table(const float* src, int32_t H, int32_t W, int32_t C, float* dst, void* stream) {
  EGO_REQUIRE(src && dst && H >= 2 && W >= 1 && C >= 1, "avgpool_table: bad argument");
  const int64_t n = (int64_t)(H / 2) * (W == 1 ? 1 : W / 2) * C;
  k_avgpool<<<nblk(n, 256), 256, 0, (hipStream_t)stream>>>(src, H, W, C, dst);
  return ego_launch_status("k_avgpool");
}

int ego_march_density(const ego_scene* sc, const float* rays, int64_t N, int32_t S, const float* z_in,
                      const float* r_sched, const float* jitter, float near_, int32_t coarse, float* z_out,
                      float* alpha, int32_t alpha_stride, float* weight, float* bg_weight, float* coords_out, float* sigma_out,
                      uint8_t* tile_active, void* stream) {
  EGO_REQUIRE(sc && rays && N >= 0 && S >= 2, "march_density: null argument or S < 2");
  if (alpha_stride == 0) alpha_stride = S;
  EGO_REQUIRE(alpha_stride >= S && alpha_stride <= S + 64, "march_density: alpha_stride must be in [S, S+64]");
  EGO_REQUIRE(z_in || r_sched, "march_density: need z_in or r_sched");
  EGO_REQUIRE(sc->r_lut && sc->n_r_lut >= 2 && sc->n_r_lut <= 1024, "march_density: r_lut missing or > 1024 entries");
  const ego_vm_field& f = (coarse & 1) ? sc->density_coarse : sc->density;
  if (int e = check_field(f, "march_density")) return e;
  if (N == 0) return EGO_OK;
  if (f.n_comp != 16) return ego_fail(EGO_E_UNSUPPORTED, "march_density: n_comp %d (supported: 16)", f.n_comp);
  k_march_density<16><<<nblk(N, 4), 256, 0, (hipStream_t)stream>>>(
      make_coords(*sc, (coarse & 2) != 0), make_field(f), rays, N, S, z_in, r_sched, jitter, near_, sc->act_softplus, sc->density_shift,
      sc->distance_scale, z_out, alpha, alpha_stride, weight, bg_weight, coords_out, sigma_out, make_occ(*sc, coarse & 1), sc->term_eps,
      fmaxf(sc->weight_thres, 0.f), tile_active);
  return ego_launch_status("k_march_density");
}

int ego_composite(const ego_scene* sc, const float* rays, const float* z, const float* weight, const float* bg_weight,
                  const float* rgb, int64_t N, int32_t S, float* rgb_map, float* depth, float* bg_map, float* env_map,
                  float* rgb_raw, void* stream) {
  EGO_REQUIRE(sc && rays && z && weight && rgb && rgb_map && N >= 0 && S >= 1, "composite: null argument");
  EGO_REQUIRE(!sc->envmap || bg_weight, "composite: envmap needs bg_weight");
  if (N == 0) return EGO_OK;
  k_composite<<<nblk(N, 4), 256, 0, (hipStream_t)stream>>>(sc->envmap, sc->envmap_h, rays, z, weight, bg_weight, rgb, N, S,
                                                          rgb_map, depth, bg_map, env_map, rgb_raw, fmaxf(sc->weight_thres, 0.f));
  return ego_launch_status("k_composite");
}

}  // extern "C"
