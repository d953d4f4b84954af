// Device helpers shared by every kernel of libegonerf_hip.so (gfx950 only, wave64).
// Arithmetic follows the reference's ATen op sequence rounding-by-rounding where that is cheap
// (explicit __f*_rn so the compiler does not contract into FMAs the CPU path does not use).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/egonerf_hip.h"

#define EGO_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// POD copies of the scene that travel in kernel arguments (SGPR-resident, no pointer chasing).
// ---------------------------------------------------------------------------------------------
struct DevField {
  const float* plane[2][3];
  const float* line[2][3];
  int32_t res[3];  // N_r, N_theta, N_phi
  // compact addressing for the forward gathers: the 12 tables of a field lie within 4 GB of `base` (the lowest of them), so a tap
  // is `base + 32-bit byte offset`: one scalar base register pair + one VGPR per address (global_load saddr form) instead of a
  // per-lane 64-bit pointer select and 64-bit adds.  ego_field_is_compact() checks it on the host; the backward kernels keep
  // using the pointer arrays.
  const char* base;
  uint32_t poff[2][3];
  uint32_t loff[2][3];
};

struct DevCoords {
  float cx, cy, cz;
  float th_near, ph_near, th_inv, ph_inv;
  const float* r_lut;
  int32_t n_lut;  // N_r + 1
  int32_t n_r;
};

__host__ inline DevField make_field(const ego_vm_field& f) {
  DevField d;
  uintptr_t lo = ~(uintptr_t)0;
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i) {
      d.plane[g][i] = f.plane[g][i];
      d.line[g][i] = f.line[g][i];
      if (f.plane[g][i] && (uintptr_t)f.plane[g][i] < lo) lo = (uintptr_t)f.plane[g][i];
      if (f.line[g][i] && (uintptr_t)f.line[g][i] < lo) lo = (uintptr_t)f.line[g][i];
    }
  for (int i = 0; i < 3; ++i) d.res[i] = f.res[i];
  d.base = (const char*)lo;
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i) {
      d.poff[g][i] = (uint32_t)((uintptr_t)f.plane[g][i] - lo);
      d.loff[g][i] = (uint32_t)((uintptr_t)f.line[g][i] - lo);
    }
  return d;
}

// true iff every table of the field (elem_bytes per element, n_comp channels) ends within 4 GB of the lowest table address, i.e.
// DevField's 32-bit byte offsets address all of it.  The host layer allocates the tables of a field from one buffer.
__host__ inline bool ego_field_is_compact(const ego_vm_field& f, int elem_bytes) {
  uintptr_t lo = ~(uintptr_t)0, hi = 0;
  const int ax_x[3] = {0, 0, 1}, ax_y[3] = {1, 2, 2}, ax_l[3] = {2, 1, 0};
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i) {
      if (!f.plane[g][i] || !f.line[g][i]) return false;
      const uintptr_t p = (uintptr_t)f.plane[g][i], l = (uintptr_t)f.line[g][i];
      const uintptr_t pe = p + (uintptr_t)f.res[ax_x[i]] * f.res[ax_y[i]] * f.n_comp * elem_bytes;
      const uintptr_t le = l + (uintptr_t)f.res[ax_l[i]] * f.n_comp * elem_bytes;
      lo = p < lo ? p : lo; lo = l < lo ? l : lo;
      hi = pe > hi ? pe : hi; hi = le > hi ? le : hi;
    }
  return hi - lo < ((uintptr_t)1 << 32);
}

__host__ inline DevCoords make_coords(const ego_scene& s, bool fine_pass = false) {
  DevCoords c;
  c.cx = s.center[0]; c.cy = s.center[1]; c.cz = s.center[2];
  c.th_near = s.ang_near[0]; c.ph_near = s.ang_near[1];
  c.th_inv = s.ang_inv[0]; c.ph_inv = s.ang_inv[1];
  c.r_lut = s.r_lut; c.n_lut = s.n_r_lut; c.n_r = s.n_r;
  if (fine_pass && s.r_lut_fine) { c.r_lut = s.r_lut_fine; c.n_lut = s.n_r_lut_fine; c.n_r = s.n_r_fine; }
  return c;
}

// ---------------------------------------------------------------------------------------------
// Row B: Cartesian -> (r, theta, phi, is_yang)   models/coordinates.py:468-498
// ---------------------------------------------------------------------------------------------
struct YinYang {
  float r, th, ph;
  int yang;
};

#define EGO_PI_4 0.78539816339744830962f       // float32(pi/4): thresholds compare in float32
#define EGO_3PI_4 2.35619449019234492885f

__device__ __forceinline__ float nan_to_zero(float v) { return (v != v) ? 0.0f : v; }

__device__ __forceinline__ YinYang yinyang_from_xyz(float x, float y, float z, const DevCoords& c) {
  const float dx = __fsub_rn(x, c.cx), dy = __fsub_rn(y, c.cy), dz = __fsub_rn(z, c.cz);
  const float r = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  const float th_n = nan_to_zero(acosf(__fdiv_rn(dz, r)));
  const float ph_n = atan2f(dy, dx);
  const bool yin = (EGO_PI_4 <= th_n) && (th_n <= EGO_3PI_4) && (-EGO_3PI_4 <= ph_n) && (ph_n <= EGO_3PI_4);
  YinYang o;
  o.r = r;
  if (yin) {
    o.th = th_n; o.ph = ph_n; o.yang = 0;
  } else {
    o.th = nan_to_zero(acosf(__fdiv_rn(dy, r)));
    o.ph = atan2f(dz, -dx);
    o.yang = 1;
  }
  return o;
}

// ---------------------------------------------------------------------------------------------
// Row C: normalisation   models/coordinates.py:442-466, :110-131, :156
// `lut` may point to LDS or global memory.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float normalize_r(float r, const float* lut, int n_lut, int n_r) {
  // searchsorted(side='right'): first index with lut[i] > r; NaN sorts last like torch.
  int lo = 0, hi = n_lut;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (!(lut[mid] > r)) lo = mid + 1; else hi = mid;
  }
  int k_out = lo < 1 ? 1 : (lo > n_lut - 1 ? n_lut - 1 : lo);
  const int k_in = k_out - 1;
  const float g0 = lut[k_in], g1 = lut[k_out];
  const float frac = __fdiv_rn(__fsub_rn(r, g0), __fsub_rn(g1, g0));
  const float v = __fdiv_rn(__fadd_rn((float)k_in, frac), (float)n_r);
  return __fsub_rn(__fmul_rn(v, 2.0f), 1.0f);
}

__device__ __forceinline__ float normalize_ang(float a, float near_, float inv) {
  return __fsub_rn(__fmul_rn(__fmul_rn(__fsub_rn(a, near_), inv), 2.0f), 1.0f);
}

// ---------------------------------------------------------------------------------------------
// Bilinear / linear tap set-up with align_corners=True, zero padding (F.grid_sample semantics,
// ATen GridSamplerKernel: ix = (x+1)*((W-1)/2); west weight = 1-fx).
// ---------------------------------------------------------------------------------------------
struct Lin1 {
  int i0, i1;    // clamped indices (always addressable)
  float w0, w1;  // weights with the out-of-range taps zeroed
};

__device__ __forceinline__ Lin1 lin_setup(float xhat, int n) {
  Lin1 t;
  const float ix = __fmul_rn(__fadd_rn(xhat, 1.0f), 0.5f * (float)(n - 1));
  const float fl = floorf(ix);
  const float f = __fsub_rn(ix, fl);
  // ix may be NaN/huge for degenerate points: keep the int conversion defined and the taps masked (fmaxf(NaN, -2) = -2, so
  // a NaN lands on i0 = -2, i1 = -1: both out of range).  One unsigned compare per tap: 0 <= i < n.
  const float flc = fminf(fmaxf(fl, -2.0f), (float)n);
  const int i0 = (int)flc, i1 = i0 + 1;
  t.w0 = ((unsigned)i0 < (unsigned)n) ? __fsub_rn(1.0f, f) : 0.0f;
  t.w1 = ((unsigned)i1 < (unsigned)n) ? f : 0.0f;
  t.i0 = min(max(i0, 0), n - 1);
  t.i1 = min(max(i1, 0), n - 1);
  return t;
}

// The three (plane, line) lookups of one grid address these axes (matMode / vecMode, EgoNeRF.py:30-33):
//   i=0: plane x=r (W=N_r), y=theta (H=N_theta); line = phi
//   i=1: plane x=r,         y=phi   (H=N_phi);   line = theta
//   i=2: plane x=theta (W=N_theta), y=phi;       line = r
struct VMTaps {
  Lin1 ax[3];  // per axis: 0 r, 1 theta, 2 phi
};

__device__ __forceinline__ VMTaps vm_setup(float a_r, float a_th, float a_ph, const int32_t res[3]) {
  VMTaps t;
  t.ax[0] = lin_setup(a_r, res[0]);
  t.ax[1] = lin_setup(a_th, res[1]);
  t.ax[2] = lin_setup(a_ph, res[2]);
  return t;
}

__device__ __forceinline__ constexpr int vm_plane_x(int i) { return i == 2 ? 1 : 0; }
__device__ __forceinline__ constexpr int vm_plane_y(int i) { return i == 0 ? 1 : 2; }
__device__ __forceinline__ constexpr int vm_line_ax(int i) { return 2 - i; }

// ---------------------------------------------------------------------------------------------
// Row E scalars
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_shift(float f, float shift) {
  const float x = __fadd_rn(f, shift);
  return x > 20.0f ? x : log1pf(expf(x));  // torch softplus, threshold 20
}

// relu as an integer max: one v_max_i32 (fmaxf costs a canonicalising v_max_f32 first); -0.0 and negatives -> +0.0
__device__ __forceinline__ float relu_f(float x) {
  return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0));
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// raw2alpha (tensorBase.py:22-27: alpha = 1 - exp(-sigma * dist)) evaluated as -expm1(-x): the correctly rounded value of the
// reference's expression.  The literal float32 form loses ~6e-8 ABSOLUTE in the subtraction (6e-6 relative at alpha = 0.01), in the
// reference as much as here but with a different exp and therefore a different error; the inverse CDF of the resampling pass
// amplifies exactly those weight errors (steep tiny grids: 1e-4 in RGB from 6e-8 in the coarse weights).  With the accurate form
// this side contributes nothing, so the distance to the reference is the reference's own rounding error instead of the sum of two.
__device__ __forceinline__ float alpha_from(float x) { return -expm1f(-x); }

// sin and cos of one argument: Cody-Waite reduction by pi/2 (two FMAs) + Cephes minimax polynomials on
// [-pi/4, pi/4].  ~1 ulp for |x| < 1e4, branch-free (the libm versions drag a Payne-Hanek slow path
// into every call site, which matters when 68 of them are inlined between MFMAs).
__device__ __forceinline__ void sincos_f32(float x, float& s_out, float& c_out) {
  const float k = rintf(x * 0.63661977236758134308f);
  float r = fmaf(k, -1.57079637050628662109375f, x);      // float32(pi/2)
  r = fmaf(k, 4.37113900018624283e-8f, r);                // pi/2 - float32(pi/2) = -4.371139e-8
  const float r2 = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f),
                        r2 * r2, fmaf(-0.5f, r2, 1.0f));
  const int q = (int)k;
  const float s = (q & 1) ? pc : ps;
  const float c = (q & 1) ? ps : pc;
  s_out = (q & 2) ? -s : s;
  c_out = ((q + 1) & 2) ? -c : c;
}

// sin/cos of x and 2x (the two positional-encoding frequencies): one reduction + double-angle identities.
// sin 2x = 2 s c, cos 2x = 1 - 2 s^2: adds <= 2 ulp to the ~1 ulp of sincos_f32.
__device__ __forceinline__ void sincos_x_2x(float x, float& s1, float& c1, float& s2, float& c2) {
  sincos_f32(x, s1, c1);
  const float t = s1 + s1;
  s2 = t * c1;
  c2 = fmaf(-t, s1, 1.0f);
}

// Same outputs through the hardware transcendentals (v_sin_f32 / v_cos_f32 take revolutions): 6 VALU issues instead
// of 27.  Measured on gfx950 (tools/sincos_probe.hip): max |err| 1.4e-7 for |x| < 2 (the feature range), 8e-7 at |x| = 12
// (the x / 2pi product rounds at |x| * 6e-8); the double angle doubles it.  Used by the f16x3 shade kernel only.
__device__ __forceinline__ void sincos_x_2x_hw(float x, float& s1, float& c1, float& s2, float& c2) {
  const float t = x * 0.15915494309189533577f;
  s1 = __builtin_amdgcn_sinf(t);
  c1 = __builtin_amdgcn_cosf(t);
  const float d = s1 + s1;
  s2 = d * c1;
  c2 = fmaf(-d, s1, 1.0f);
}

// wave64 inclusive multiplicative scan on the DPP path (no LDS round trips; __shfl_up compiles to ds_bpermute_b32 + s_waitcnt):
// row_shr 1, 2, 4, 8 inside the 16-lane rows, then row_bcast:15 into rows 1 / 3 and row_bcast:31 into rows 2 / 3.  Lanes without a
// source keep `old` = 1.0, the identity, so no per-step select is needed.
__device__ __forceinline__ float wave_scan_mul(float v, int /*lane*/) {
  const int one = __float_as_int(1.0f);
#define EGO_SCAN_STEP(ctrl, rmask) v *= __int_as_float(__builtin_amdgcn_update_dpp(one, __float_as_int(v), ctrl, rmask, 0xf, false))
  EGO_SCAN_STEP(0x111, 0xf);  // row_shr:1
  EGO_SCAN_STEP(0x112, 0xf);  // row_shr:2
  EGO_SCAN_STEP(0x114, 0xf);  // row_shr:4
  EGO_SCAN_STEP(0x118, 0xf);  // row_shr:8
  EGO_SCAN_STEP(0x142, 0xa);  // row_bcast:15 -> rows 1, 3
  EGO_SCAN_STEP(0x143, 0xc);  // row_bcast:31 -> rows 2, 3
#undef EGO_SCAN_STEP
  return v;
}

// Row J - environment map (models/envmap.py:6-14, 26-34): bilinear lookup of the sigmoid-ed emission in direction (dx, dy, dz)
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
