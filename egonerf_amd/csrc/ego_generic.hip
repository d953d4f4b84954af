// libegonerf_hip.so, part 7: the render path for model shapes OTHER than the one every shipped config resolves to
// (opt.py:87-100 lets a user set n_lamb_sigma / n_lamb_sh, data_dim_color, featureC, view_pe, fea_pe).  The MFMA kernels of
// ego_shade.hip bake 48 appearance components, app_dim 27, a 150 -> 128 -> 128 -> 3 MLP with two encoding frequencies and 16
// density components into their register layouts; these kernels take the shape as runtime numbers (hidden width as a template
// parameter) and compute in plain fp32 with the reference's operation order, so that any reference checkpoint renders on the
// device.  They are a compatibility path, an order of magnitude slower than the tuned one (forward AND, since round 4, backward); the
// host layer picks them exactly when ego_shape_is_tuned() is false.
//
//   k_march_generic : rows A-E for C density components (multiple of 4, <= 48), lane = sample, wave per ray
//   k_shade_generic : rows F, G (EgoNeRF.py:349-413, tensorBase.py:54-78): appearance gather + per-grid basis + positional
//                     encoding + MLP_Fea for n_comp <= 48 (multiple of 4), app_dim <= 32, featureC in {64, 128}, view_pe, fea_pe <= 8
#include "ego_device.h"
#include "ego_host.h"
#include "ego_generic.h"

namespace {

struct GenShadeArgs {
  DevField F;
  const float* gp;       // generic packed weights (ego_generic_pack)
  const float* rays;     // SHADE: [N][6]
  const float* coords;   // SHADE: [M][4]
  const float* c7n;      // APP: [M][7]
  const float* feat;     // MLP: [M][app_dim]
  const float* dirs;     // MLP: [M][3]
  float* out;
  const uint8_t* tile_active;
  int64_t M;
  int32_t S, app_dim, n_comp, in_c, view_pe, fea_pe;
  int32_t head;          // EGO_HEAD_MLP_FEA: features -> MLP -> sigmoid; EGO_HEAD_RGB: colour = features 0..2 (RGBRender, tensorBase.py:37-39)
  // training forward (DUMP): row-major per-sample activations for the backward pass / the weight-gradient products
  float* dump_x;   // [M][ldx]: the MLP input row in the reference's column order (tensorBase.py:68-75)
  float* dump_h1;  // [M][ldh]: relu(h1)
  float* dump_h2;  // [M][ldh]: relu(h2)
  float* dump_v;   // [M][ldv]: plane x line products, column = plane * C + channel
  int32_t ldx, ldh, ldv;
};

// packed layout (floats): W1T [in_c][HID] | b1 [HID] | W2T [HID][HID] | b2 [HID] | W3 [3][HID] | b3 [4] | basisT [2][3 C][32] | W1 | W2 | basis [2][32][3 C]
struct GenLayout {
  int64_t w1t, b1, w2t, b2, w3, b3, basis, w1n, w2n, basisn, total;
};
__host__ __device__ inline GenLayout gen_layout(int in_c, int hid, int n_comp) {
  GenLayout L;
  int64_t o = 0;
  L.w1t = o; o += (int64_t)in_c * hid;
  L.b1 = o; o += hid;
  L.w2t = o; o += (int64_t)hid * hid;
  L.b2 = o; o += hid;
  L.w3 = o; o += 3 * hid;
  L.b3 = o; o += 4;
  L.basis = o; o += 2 * 3 * (int64_t)n_comp * 32;
  // the same matrices the other way round, for the backward's products on the matrix pipe (B operands with the OUTPUT index contiguous):
  // W1 [HID][in_c], W2 [HID][HID] (the reference's own layouts), basis [2][32][3 C] (rows >= app_dim zero)
  L.w1n = o; o += (int64_t)hid * in_c;
  L.w2n = o; o += (int64_t)hid * hid;
  L.basisn = o; o += 2 * 32 * 3 * (int64_t)n_comp;
  L.total = o;
  return L;
}

__global__ void k_generic_pack(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                               const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
                               const float* __restrict__ basis_yin, const float* __restrict__ basis_yang, int in_c, int hid, int n_comp,
                               int app_dim, float* __restrict__ out) {
  const GenLayout L = gen_layout(in_c, hid, n_comp);
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= L.total) return;
  float v = 0.f;
  if (idx < L.b1) { const int k = (int)(idx / hid), j = (int)(idx % hid); v = w1[(int64_t)j * in_c + k]; }
  else if (idx < L.w2t) v = b1[idx - L.b1];
  else if (idx < L.b2) { const int64_t e = idx - L.w2t; const int k = (int)(e / hid), j = (int)(e % hid); v = w2[(int64_t)j * hid + k]; }
  else if (idx < L.w3) v = b2[idx - L.b2];
  else if (idx < L.b3) v = w3[idx - L.w3];
  else if (idx < L.basis) { const int c = (int)(idx - L.b3); v = (c < 3 && b3) ? b3[c] : 0.f; }   // b3 == null: EGO_HEAD_RGB (hid = in_c = 0)
  else if (idx < L.w1n) {
    const int64_t e = idx - L.basis;
    const int f = (int)(e & 31), col = (int)((e >> 5) % (3 * n_comp)), g = (int)((e >> 5) / (3 * n_comp));
    if (f < app_dim) v = (g ? basis_yang : basis_yin)[(int64_t)f * (3 * n_comp) + col];
  }
  else if (idx < L.w2n) v = w1[idx - L.w1n];
  else if (idx < L.basisn) v = w2[idx - L.w2n];
  else {
    const int64_t e = idx - L.basisn;
    const int ncol = 3 * n_comp, col = (int)(e % ncol), f = (int)((e / ncol) & 31), g = (int)(e / ((int64_t)32 * ncol));
    if (f < app_dim) v = (g ? basis_yang : basis_yin)[(int64_t)f * ncol + col];
  }
  out[idx] = v;
}

enum { G_SHADE = 0, G_APP = 1, G_MLP = 2 };
#ifndef EGO_GENERIC_MFMA
#define EGO_GENERIC_MFMA 1
#endif

// n iterations of body(it, b) with the NB values b = loadb(it) of iteration it + PF already in flight: one wave per SIMD has nobody else
// to hide a global load behind (a loop that loads, waits and multiplies spent 3 000 clocks per 512 clocks of MFMA)
template <int PF, int NB, class LB, class BD>
__device__ __forceinline__ void gen_pipelined(int n, LB loadb, BD body) {
  if (n <= 0) return;
  float bq[PF][NB];
#pragma unroll
  for (int p = 0; p < PF; ++p) loadb(p < n ? p : n - 1, bq[p]);
  for (int i0 = 0; i0 < n; i0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int it = i0 + p;
      if (it < n) {
        float b[NB];
#pragma unroll
        for (int e = 0; e < NB; ++e) b[e] = bq[p][e];
        const int nx = it + PF;
        loadb(nx < n ? nx : n - 1, bq[p]);
        __builtin_amdgcn_sched_barrier(0);   // (the machine scheduler sinks the loads back to their uses otherwise: `s_waitcnt vmcnt(0)` at the loop's top)
        body(it, b);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// dv element (sample ms, column c = plane * C + channel): row-major [M][ldv], or (ldv == 0; C == 48) the tuned scatters' blocked layout
// [tile of 32 samples][plane * 3 + 16-channel line][sample][16] (include/egonerf_hip.h, ego_shade_backward)
__device__ __forceinline__ int64_t gen_dv_index(int64_t ms, int c, int ldv) {
  return ldv ? ms * ldv + c : (ms >> 5) * (32 * 144) + (int64_t)(c >> 4) * 512 + (ms & 31) * 16 + (c & 15);
}

__device__ __forceinline__ void wave_sync_g() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
typedef __attribute__((address_space(4))) float gen_cfloat;

// torch.sin / torch.cos of the positional encodings (tensorBase.py:10-19), both from ONE argument reduction, branch-free: the reduction
// by pi/2 runs in float64 (x - k pi/2 with a two-term pi/2: exact to the last bit of the float32 remainder while |x| < ~1e15 - libm's
// float32 route needs a Payne-Hanek slow path for that, and sinf + cosf inlined 2 x 32 times per frequency were two thirds of this
// kernel's instructions), the Cephes minimax polynomials of sincos_f32 (ego_device.h) on [-pi/4, pi/4] follow in float32: ~1 ulp.
__device__ __forceinline__ void gen_sincos(float x, float& s_out, float& c_out) {
  const double xd = (double)x;
  const double kd = rint(xd * 0.63661977236758134308);
  double rd = fma(kd, -1.57079632679489655800e+00, xd);      // double(pi/2)
  rd = fma(kd, -6.12323399573676603587e-17, rd);             // pi/2 - double(pi/2)
  const float r = (float)rd, r2 = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f), r2 * r2, fmaf(-0.5f, r2, 1.0f));
  const int q = (int)(kd - 4.0 * floor(kd * 0.25));          // k mod 4 in 0 .. 3, whatever the size of k
  const float s = (q & 1) ? pc : ps, c = (q & 1) ? ps : pc;
  s_out = (q & 2) ? -s : s;
  c_out = ((q + 1) & 2) ? -c : c;
}

// one wave = 64 samples (lane = sample); 2 waves per workgroup; per wave an LDS slab [HID][64] that first stages chunks of the MLP
// input and then holds relu(h1)
template <int HID, int MODE, bool DUMP = false>
__global__ __launch_bounds__(128) void k_shade_generic(GenShadeArgs A) {
  // the weight blob through the CONSTANT address space: the dumping variant's stores may alias a plain global pointer as far as the
  // compiler can tell, the uniform weight rows then come through VECTOR loads (128 VGPRs per row) instead of scalar ones and the kernel
  // spills 434 registers (restrict-qualified locals did not convince it); nothing writes the blob while a kernel reads it
  const gen_cfloat* gp = (const gen_cfloat*)A.gp;
  float* dump_x = A.dump_x;
  float* dump_h1 = A.dump_h1;
  float* dump_h2 = A.dump_h2;
  float* dump_v = A.dump_v;
  float* outp = A.out;
  __shared__ float slab[2][HID][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float (*sl)[64] = slab[wv];
  const GenLayout L = gen_layout(A.in_c, A.head == EGO_HEAD_RGB ? 0 : HID, A.n_comp);   // no MLP block in the RGB head's blob
  const int64_t n_units = (A.M + 63) >> 6;
  for (int64_t unit = (int64_t)blockIdx.x * 2 + wv; unit < n_units; unit += (int64_t)gridDim.x * 2) {
    if (MODE == G_SHADE && A.tile_active) {   // two 32-sample tiles per unit (the last unit of a ragged M may hold one): skip when neither is read
      const int64_t n_tiles = (A.M + 31) >> 5, t1 = 2 * unit + 1;
      if (!A.tile_active[2 * unit] && !(t1 < n_tiles && A.tile_active[t1])) continue;
    }
    const int64_t m_raw = unit * 64 + lane;
    const bool valid = m_raw < A.M;
    const int64_t m = valid ? m_raw : A.M - 1;
    float feat[32];
#pragma unroll
    for (int f = 0; f < 32; ++f) feat[f] = 0.f;
    float vd[3] = {0.f, 0.f, 0.f};
#if EGO_GENERIC_MFMA
    const int l31 = lane & 31, kk = lane >> 5;
    const float* gw = A.gp;
    // slab element (row, sample): the column is swizzled with the row so that the 32 lanes of a D tile (32 rows, one sample) hit 32 banks
    auto SL = [&](int row, int col) -> float& { return sl[row][col ^ (row & 31)]; };
#endif
    if (MODE == G_MLP) {
#pragma unroll
      for (int f = 0; f < 32; ++f)
        if (f < A.app_dim) feat[f] = A.feat[m * A.app_dim + f];
      vd[0] = A.dirs[m * 3]; vd[1] = A.dirs[m * 3 + 1]; vd[2] = A.dirs[m * 3 + 2];
    } else {
      // ---- rows F: appearance gather (EgoNeRF.py:349-413) + this sample's grid's basis (EgoNeRF.py:99-100) ----
      float a[3];
      int g;
      if (MODE == G_APP) {
        const float* p = A.c7n + m * 7;
        g = (p[6] == 0.f) ? 0 : 1;
        const int b = g ? 3 : 0;
        a[0] = p[b]; a[1] = p[b + 1]; a[2] = p[b + 2];
      } else {
        const f32x4 cc = ((const f32x4*)A.coords)[m];
        a[0] = cc.x; a[1] = cc.y; a[2] = cc.z; g = cc.w != 0.f;
        const uint32_t ray = (uint32_t)(m / A.S);
        const float* R = A.rays + (int64_t)ray * 6;
        vd[0] = R[3]; vd[1] = R[4]; vd[2] = R[5];
      }
      const VMTaps t = vm_setup(a[0], a[1], a[2], A.F.res);
      const int C = A.n_comp;
#if EGO_GENERIC_MFMA
      // features = basis_g (plane value x line value): the unit's [64 samples][3 C] products go through the slab a plane at a time and
      // meet basisT [g][column][32 features] on the matrix pipe; a sample's row of A is zero for the grid it does not belong to, so both
      // grids accumulate into the same tiles (the lane = sample form took 2 x 32 scalar weights and 32 selects per column and sample)
      f32x16 accf[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accf[mt][r] = 0.f;
      const unsigned long long gmask = __ballot(g != 0);
      const bool any_yin = ~gmask != 0ull, any_yang = gmask != 0ull;
      const bool yang0 = (gmask >> l31) & 1ull, yang1 = (gmask >> (32 + l31)) & 1ull;   // grid of samples l31 and 32 + l31
#endif
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const Lin1 X = t.ax[vm_plane_x(i)], Y = t.ax[vm_plane_y(i)], Ln = t.ax[vm_line_ax(i)];
        const int W = A.F.res[vm_plane_x(i)];
        const float* P = g ? A.F.plane[1][i] : A.F.plane[0][i];
        const float* Lp = g ? A.F.line[1][i] : A.F.line[0][i];
        const float* p00 = P + ((int64_t)Y.i0 * W + X.i0) * C;
        const float* p01 = P + ((int64_t)Y.i0 * W + X.i1) * C;
        const float* p10 = P + ((int64_t)Y.i1 * W + X.i0) * C;
        const float* p11 = P + ((int64_t)Y.i1 * W + X.i1) * C;
        const float* l0 = Lp + (int64_t)Ln.i0 * C;
        const float* l1 = Lp + (int64_t)Ln.i1 * C;
        const float w00 = __fmul_rn(Y.w0, X.w0), w01 = __fmul_rn(Y.w0, X.w1), w10 = __fmul_rn(Y.w1, X.w0), w11 = __fmul_rn(Y.w1, X.w1);
        for (int c4 = 0; c4 < C; c4 += 4) {
          const f32x4 pv = *(const f32x4*)(p00 + c4) * w00 + *(const f32x4*)(p01 + c4) * w01 + *(const f32x4*)(p10 + c4) * w10 +
                           *(const f32x4*)(p11 + c4) * w11;
          const f32x4 lv = *(const f32x4*)(l0 + c4) * Ln.w0 + *(const f32x4*)(l1 + c4) * Ln.w1;
          const f32x4 pr = pv * lv;
          if (DUMP && valid) *(f32x4*)(dump_v + m * A.ldv + i * C + c4) = pr;
#if EGO_GENERIC_MFMA
#pragma unroll
          for (int e = 0; e < 4; ++e) SL(c4 + e, lane) = pr[e];
#else
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // basisT [g][col][32]: uniform addresses -> scalar loads of both grids' rows, selected per lane
            const gen_cfloat* b0 = gp + L.basis + (int64_t)(i * C + c4 + e) * 32;
            const gen_cfloat* b1 = b0 + (int64_t)3 * C * 32;
#pragma unroll
            for (int f = 0; f < 32; ++f)
              if (f < A.app_dim) feat[f] = fmaf(g ? b1[f] : b0[f], pr[e], feat[f]);
          }
#endif
        }
#if EGO_GENERIC_MFMA
        wave_sync_g();
        gen_pipelined<4, 2>((C + 1) / 2,
          [&](int it, float (&bq)[2]) {
            const int tt = 2 * it + kk, tc = tt < C ? tt : C - 1;
            const float* bw = gw + L.basis + (int64_t)(i * C + tc) * 32 + l31;
            bq[0] = any_yin ? bw[0] : 0.f;
            bq[1] = any_yang ? bw[(int64_t)3 * C * 32] : 0.f;
          },
          [&](int it, const float (&bq)[2]) {
            const int tt = 2 * it + kk;
            const bool ok = tt < C;
            const int tc = ok ? tt : C - 1;
            const float x0 = ok ? SL(tc, l31) : 0.f, x1 = ok ? SL(tc, 32 + l31) : 0.f;
            if (any_yin) {
              accf[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(yang0 ? 0.f : x0, bq[0], accf[0], 0, 0, 0);
              accf[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(yang1 ? 0.f : x1, bq[0], accf[1], 0, 0, 0);
            }
            if (any_yang) {
              accf[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(yang0 ? x0 : 0.f, bq[1], accf[0], 0, 0, 0);
              accf[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(yang1 ? x1 : 0.f, bq[1], accf[1], 0, 0, 0);
            }
          });
        wave_sync_g();
#endif
      }
#if EGO_GENERIC_MFMA
      // D register r of lane l = sample 32 mt + 8 (r / 4) + 4 (l / 32) + r % 4, feature l % 32: back to lane = sample through the slab
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) SL(l31, 32 * mt + 8 * (r >> 2) + 4 * kk + (r & 3)) = accf[mt][r];
      wave_sync_g();
#pragma unroll
      for (int f = 0; f < 32; ++f)
        if (f < A.app_dim) feat[f] = SL(f, lane);
      wave_sync_g();
#endif
    }
    if (MODE == G_APP) {
      if (valid) {
#pragma unroll
        for (int f = 0; f < 32; ++f)
          if (f < A.app_dim) outp[m * A.app_dim + f] = feat[f];
      }
      continue;
    }
    if (A.head == EGO_HEAD_RGB) {   // RGBRender (tensorBase.py:37-39): the colour IS the (3-channel) appearance feature; no sigmoid, no clamp
      if (valid) {
        float* op = outp + m * 3;
        op[0] = feat[0]; op[1] = feat[1]; op[2] = feat[2];
      }
      continue;
    }
    // ---- row G: mlp_in = [features, viewdirs, PE(features), PE(viewdirs)] (tensorBase.py:68-75) -> Linear relu Linear relu Linear ----
#if EGO_GENERIC_MFMA
    // Both hidden layers on the matrix pipe in fp32 (v_mfma_f32_32x32x2_f32: products and sums in fp32, as the reference's): the unit's 64
    // samples x HID hidden units are 2 x HID / 32 tiles of 16 accumulators; A = the staged inputs from the slab (lane = (sample l % 32,
    // k = l / 32)), B = two rows of W^T by coalesced VECTOR loads (lane = (k, hidden unit l % 32)).  The lane = sample form it replaces
    // fed every FMA an SGPR weight: 128 weights per input row through a 104-SGPR file, every row two or three dependent scalar-load ->
    // FMA phases with one wave per SIMD to hide them (5-10 % of the fp32 rate; kept as -DEGO_GENERIC_MFMA=0).
    constexpr int NT = HID / 32;
    f32x16 acc[2][NT];
    auto bias = [&](int64_t off) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bv = gw[off + 32 * nt + l31];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = bv;
      }
    };
    // n staged inputs (slab rows off .. off + n - 1) times rows wrow0 + t * stride of the [rows][HID] matrix at `wbase`
    auto matmul = [&](int n, int off, int64_t wbase, int wrow0, int stride) {
      wave_sync_g();   // the slab rows were written by lane = sample
      gen_pipelined<2, NT>((n + 1) / 2,
        [&](int it, float (&bq)[NT]) {
          const int tt = 2 * it + kk, tc = tt < n ? tt : n - 1;
          const float* w = gw + wbase + (int64_t)(wrow0 + tc * stride) * HID + l31;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bq[nt] = w[32 * nt];
        },
        [&](int it, const float (&bq)[NT]) {
          const int tt = 2 * it + kk;
          const bool ok = tt < n;
          const int tc = ok ? tt : n - 1;
          float av[2];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) { const float x = SL(off + tc, 32 * mt + l31); av[mt] = ok ? x : 0.f; }
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bq[nt], acc[mt][nt], 0, 0, 0);
        });
      wave_sync_g();   // ... and will be overwritten by the next chunk
    };
    // relu of the accumulators -> slab [hidden][sample] (+ the row-major dump: a register's 32 lanes are 32 consecutive hidden units of
    // one sample - 128-byte stores); D register r of lane l = sample 32 mt + 8 (r / 4) + 4 (l / 32) + r % 4, hidden unit 32 nt + l % 32
    const bool full = unit * 64 + 64 <= A.M;
    auto relu_out = [&](float* dump) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int smp = 32 * mt + 8 * (r >> 2) + 4 * kk + (r & 3), hu = 32 * nt + l31;
            const float hv = fmaxf(acc[mt][nt][r], 0.f);
            SL(hu, smp) = hv;
            if (DUMP) {
              const int64_t ms = unit * 64 + smp;
              if (full || ms < A.M) dump[ms * A.ldh + hu] = hv;   // (`full`, a scalar, spares 128 exec-mask branches: the stores then leave in one batch)
            }
          }
    };
    auto consume = [&](int n, int row0, int stride, int off = 0) {
      if (DUMP && valid)
        for (int tt = 0; tt < n; ++tt) dump_x[m * A.ldx + row0 + tt * stride] = SL(off + tt, lane);
      matmul(n, off, L.w1t, row0, stride);
    };
    bias(L.b1);
    const int D = A.app_dim;
    // chunk 1: the raw features and the view direction (rows 0 .. D + 2)
#pragma unroll
    for (int f = 0; f < 32; ++f)
      if (f < D) SL(f, lane) = feat[f];
    SL(D, lane) = vd[0]; SL(D + 1, lane) = vd[1]; SL(D + 2, lane) = vd[2];
    consume(D + 3, 0, 1);
    // PE(features): element-major, frequency-minor; all sines (rows base + f * fea_pe + q), then all cosines
    {
      const int base_s = D + 3, base_c = base_s + D * A.fea_pe;
      float fr = 1.f;
      for (int q = 0; q < A.fea_pe; ++q, fr *= 2.f) {
        // sine and cosine of one argument from ONE reduction (gen_sincos), the cosines parked in rows D .. 2 D - 1 of the slab (2 D <= 64 <= HID)
#pragma unroll
        for (int f = 0; f < 32; ++f)
          if (f < D) {
            float sv, cv;
            gen_sincos(__fmul_rn(feat[f], fr), sv, cv);
            SL(f, lane) = sv; SL(D + f, lane) = cv;
          }
        consume(D, base_s + q, A.fea_pe);
        consume(D, base_c + q, A.fea_pe, D);
      }
      const int vbase_s = base_c + D * A.fea_pe, vbase_c = vbase_s + 3 * A.view_pe;
      fr = 1.f;
      for (int q = 0; q < A.view_pe; ++q, fr *= 2.f) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          float sv, cv;
          gen_sincos(__fmul_rn(vd[d], fr), sv, cv);
          SL(d, lane) = sv; SL(3 + d, lane) = cv;
        }
        consume(3, vbase_s + q, A.view_pe);
        consume(3, vbase_c + q, A.view_pe, 3);
      }
    }
    relu_out(dump_h1);
    bias(L.b2);
    matmul(HID, 0, L.w2t, 0, 1);
    relu_out(dump_h2);
    wave_sync_g();
    const gen_cfloat* w3 = gp + L.w3;
    const gen_cfloat* b3 = gp + L.b3;
    float o[3] = {b3[0], b3[1], b3[2]};
    for (int j = 0; j < HID; ++j) {   // three outputs: lane = sample again, the weights as scalars
      const float hv = SL(j, lane);
      o[0] = fmaf(w3[j], hv, o[0]); o[1] = fmaf(w3[HID + j], hv, o[1]); o[2] = fmaf(w3[2 * HID + j], hv, o[2]);
    }
    if (valid) {
      float* op = outp + m * 3;
      op[0] = sigmoidf(o[0]); op[1] = sigmoidf(o[1]); op[2] = sigmoidf(o[2]);
    }
    wave_sync_g();   // the next unit stages into the slab
  }
}
#else
    float h[HID];
    {
      const gen_cfloat* b1 = gp + L.b1;
#pragma unroll
      for (int j = 0; j < HID; ++j) h[j] = b1[j];
    }
    // consume n staged inputs: input t of the chunk sits in sl[t][lane] and multiplies row (row0 + t * stride) of W1^T
    auto consume = [&](int n, int row0, int stride, int off = 0) {
      for (int tt = 0; tt < n; ++tt) {
        const float xk = sl[off + tt][lane];
        if (DUMP && valid) dump_x[m * A.ldx + row0 + tt * stride] = xk;
        const gen_cfloat* wrow = gp + L.w1t + (int64_t)(row0 + tt * stride) * HID;   // uniform: scalar loads
#pragma unroll
        for (int j = 0; j < HID; ++j) h[j] = fmaf(wrow[j], xk, h[j]);
      }
    };
    const int D = A.app_dim;
    // chunk 1: the raw features and the view direction (rows 0 .. D + 2)
#pragma unroll
    for (int f = 0; f < 32; ++f)
      if (f < D) sl[f][lane] = feat[f];
    sl[D][lane] = vd[0]; sl[D + 1][lane] = vd[1]; sl[D + 2][lane] = vd[2];
    consume(D + 3, 0, 1);
    // PE(features): element-major, frequency-minor; all sines (rows base + f * fea_pe + q), then all cosines
    {
      const int base_s = D + 3, base_c = base_s + D * A.fea_pe;
      float fr = 1.f;
      for (int q = 0; q < A.fea_pe; ++q, fr *= 2.f) {
        // sine and cosine of one argument from ONE reduction (gen_sincos), the cosines parked in rows D .. 2 D - 1 of the slab
        // (2 D <= 64 <= HID): libm's sinf + cosf inlined 2 x 32 times per frequency were 2/3 of this kernel's instructions
#pragma unroll
        for (int f = 0; f < 32; ++f)
          if (f < D) {
            float sv, cv;
            gen_sincos(__fmul_rn(feat[f], fr), sv, cv);
            sl[f][lane] = sv; sl[D + f][lane] = cv;
          }
        consume(D, base_s + q, A.fea_pe);
        consume(D, base_c + q, A.fea_pe, D);
      }
      const int vbase_s = base_c + D * A.fea_pe, vbase_c = vbase_s + 3 * A.view_pe;
      fr = 1.f;
      for (int q = 0; q < A.view_pe; ++q, fr *= 2.f) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          float sv, cv;
          gen_sincos(__fmul_rn(vd[d], fr), sv, cv);
          sl[d][lane] = sv; sl[3 + d][lane] = cv;
        }
        consume(3, vbase_s + q, A.view_pe);
        consume(3, vbase_c + q, A.view_pe, 3);
      }
    }
    // relu(h1) -> LDS, layer 2 with the hidden index as the (runtime) loop
#pragma unroll
    for (int j = 0; j < HID; ++j) {
      sl[j][lane] = fmaxf(h[j], 0.f);
      if (DUMP && valid) dump_h1[m * A.ldh + j] = fmaxf(h[j], 0.f);
    }
    {
      const gen_cfloat* b2 = gp + L.b2;
#pragma unroll
      for (int j = 0; j < HID; ++j) h[j] = b2[j];
      for (int k = 0; k < HID; ++k) {
        const float xk = sl[k][lane];
        const gen_cfloat* wrow = gp + L.w2t + (int64_t)k * HID;
#pragma unroll
        for (int j = 0; j < HID; ++j) h[j] = fmaf(wrow[j], xk, h[j]);
      }
    }
    const gen_cfloat* w3 = gp + L.w3;
    const gen_cfloat* b3 = gp + L.b3;
    float o[3] = {b3[0], b3[1], b3[2]};
#pragma unroll
    for (int j = 0; j < HID; ++j) {
      const float hv = fmaxf(h[j], 0.f);
      if (DUMP && valid) dump_h2[m * A.ldh + j] = hv;
      o[0] = fmaf(w3[j], hv, o[0]); o[1] = fmaf(w3[HID + j], hv, o[1]); o[2] = fmaf(w3[2 * HID + j], hv, o[2]);
    }
    if (valid) {
      float* op = outp + m * 3;
      op[0] = sigmoidf(o[0]); op[1] = sigmoidf(o[1]); op[2] = sigmoidf(o[2]);
    }
  }
}
#endif

// ---- training of the other model shapes: backward of rows F, G and of the VM lookups, plain fp32, lane = sample --------------------
// The data-gradient chain of one sample: do = dL/d(pre-sigmoid) -> dh2 = relu'(h2) W3^T do -> dh1 = relu'(h1) W2^T dh2 -> dx = W1^T dh1
// -> feature gradients through the encodings (d sin(f w)/df = w cos(f w), taken from the dumped cosines) -> dv = B_g^T dfe.  The
// weight gradients are plain A^T B products over the row-major buffers written here and by the dumping forward (ego_weight_grad).
struct GenBwdArgs {
  const float* gp;
  const float* coords;  // [M][4]
  float* dc;            // in: dL/d rgb_sample [M][3]; out: dL/d(pre-sigmoid)
  const float* rgb;     // [M][3] the forward's colours
  const float* x;       // dumps of the forward
  const float* h1;
  const float* h2;
  float* dh2;           // out [M][HID]
  float* dh1;           // out [M][HID]
  float* dfe;           // out [M][64]: columns [32 g, 32 g + 32) of the sample's grid g, the other half zero
  float* dv;            // out [M][ldv]
  int64_t M;
  int32_t app_dim, n_comp, in_c, view_pe, fea_pe, ldx, ldh, ldv;
  int32_t head;
};

#if EGO_GENERIC_MFMA
// The data-gradient chain on the matrix pipe (fp32 MFMA, as the forward): one wave = 64 samples, one wave per workgroup, slab rows
// [0, HID): the A operand of the running product ([k][sample]), [HID, HID + 32): one 32-column tile of dx, [HID + 32, HID + 64): the
// feature gradients being summed.  B operands from the natural-order copies in the blob (output index contiguous: coalesced loads); the
// relu masks read h2 / h1 and write dh2 / dh1 in the D layout, where a register's 32 lanes are 32 consecutive hidden units of one sample.
template <int HID>
__global__ __launch_bounds__(64) void k_shade_generic_bwd(GenBwdArgs A) {
  __shared__ float slab[HID + 64][64];
  const int lane = threadIdx.x & 63, l31 = lane & 31, kk = lane >> 5;
  float (*sl)[64] = slab;
  auto SL = [&](int row, int col) -> float& { return sl[row][col ^ (row & 31)]; };
  const GenLayout L = gen_layout(A.in_c, A.head == EGO_HEAD_RGB ? 0 : HID, A.n_comp);
  const int64_t n_units = (A.M + 63) >> 6;
  const int D = A.app_dim;
  const float* gw = A.gp;
  constexpr int NT = HID / 32;
  for (int64_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
    const int64_t m_raw = unit * 64 + lane;
    const bool valid = m_raw < A.M;
    const int64_t m = valid ? m_raw : A.M - 1;
    const int g = ((const f32x4*)A.coords)[m].w != 0.f;
    const int ncol = 3 * A.n_comp;
    if (A.head == EGO_HEAD_RGB) {   // colour = features: dfe = dL/d rgb_sample, then dv = B_g^T dfe (basisT [g][col][32])
      float dfe[3] = {A.dc[m * 3], A.dc[m * 3 + 1], A.dc[m * 3 + 2]};
      if (valid) {
#pragma unroll
        for (int f = 0; f < 32; ++f) {
          A.dfe[m * 64 + 32 * g + f] = f < 3 ? dfe[f] : 0.f;
          A.dfe[m * 64 + 32 * (1 - g) + f] = 0.f;
        }
      }
      for (int col = 0; col < ncol; ++col) {
        const float* b0 = A.gp + L.basis + (int64_t)col * 32;
        const float* b1 = b0 + (int64_t)ncol * 32;
        float s = 0.f;
#pragma unroll
        for (int f = 0; f < 3; ++f) s = fmaf(g ? b1[f] : b0[f], dfe[f], s);
        if (valid) A.dv[gen_dv_index(m, col, A.ldv)] = s;
      }
      continue;
    }
    // acc[mt][nt] (+)= slab rows [off, off + n) x rows of the [..][ld] matrix at wbase, columns col0 + 32 nt + l % 32 (zero beyond ncols)
    auto matmul = [&](f32x16 (&acc)[2][NT], int nts, int n, int off, int64_t wbase, int ld, int col0, int ncols) {
      wave_sync_g();
      gen_pipelined<2, NT>((n + 1) / 2,
        [&](int it, float (&bq)[NT]) {
          const int tt = 2 * it + kk, tc = tt < n ? tt : n - 1;
          const float* w = gw + wbase + (int64_t)tc * ld;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int c = col0 + 32 * nt + l31;
            const float x = w[c < ncols ? c : ncols - 1];
            bq[nt] = (nt < nts && c < ncols) ? x : 0.f;
          }
        },
        [&](int it, const float (&bq)[NT]) {
          const int tt = 2 * it + kk;
          const bool ok = tt < n;
          const int tc = ok ? tt : n - 1;
          float av[2];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) { const float x = SL(off + tc, 32 * mt + l31); av[mt] = ok ? x : 0.f; }
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              if (nt < nts) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bq[nt], acc[mt][nt], 0, 0, 0);
        });
      wave_sync_g();
    };
    const bool full = unit * 64 + 64 <= A.M;
    auto zero = [&](f32x16 (&acc)[2][NT]) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    };
    // relu'(h) mask in the D layout, the masked gradient out (row-major [M][HID]) and into slab rows [0, HID) as the next A operand
    auto mask_out = [&](f32x16 (&acc)[2][NT], const float* h, float* out) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          // a tile's 16 loads first, then its 16 stores: `out` may alias `h` as far as the compiler knows, and load, store, load, store
          // in program order is one exposed round trip per element (6 of this kernel's 10 ms)
          const int hu = 32 * nt + l31;
          float hv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t ms = unit * 64 + 32 * mt + 8 * (r >> 2) + 4 * kk + (r & 3), mc = (full || ms < A.M) ? ms : A.M - 1;
            hv[r] = h[mc * A.ldh + hu];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int smp = 32 * mt + 8 * (r >> 2) + 4 * kk + (r & 3);
            const int64_t ms = unit * 64 + smp;
            const float v = hv[r] > 0.f ? acc[mt][nt][r] : 0.f;      // threshold backward of torch.nn.ReLU
            if (full || ms < A.M) out[ms * HID + hu] = v;
            SL(hu, smp) = v;
          }
        }
    };
    float d_o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float y = A.rgb[m * 3 + c];
      d_o[c] = A.dc[m * 3 + c] * y * (1.f - y);       // sigmoid'
    }
    if (valid) { A.dc[m * 3] = d_o[0]; A.dc[m * 3 + 1] = d_o[1]; A.dc[m * 3 + 2] = d_o[2]; }
    SL(0, lane) = d_o[0]; SL(1, lane) = d_o[1]; SL(2, lane) = d_o[2];
    f32x16 acc[2][NT];
    zero(acc);
    matmul(acc, NT, 3, 0, L.w3, HID, 0, HID);                 // dh2 (before the mask) = d_o W3, W3 [3][HID]
    mask_out(acc, A.h2, A.dh2);
    zero(acc);
    matmul(acc, NT, HID, 0, L.w2n, HID, 0, HID);              // dh1 (before the mask) = dh2 W2, W2 [j][k]
    mask_out(acc, A.h1, A.dh1);
    // dx = dh1 W1 a 32-column tile at a time, each tile folded into the feature gradients through the encodings' derivatives
    // (d sin(f w) = w cos(f w) df, d cos(f w) = -w sin(f w) df, sines and cosines taken from the dumped x)
#pragma unroll
    for (int f = 0; f < 32; ++f) SL(HID + 32 + f, lane) = 0.f;
    const int base_s = D + 3, base_c = base_s + D * A.fea_pe, vbase_s = base_c + D * A.fea_pe;
    const float* xr = A.x + m * A.ldx;
    for (int c0 = 0; c0 < vbase_s; c0 += 32 * NT) {           // (the view-direction encodings behind vbase_s carry no gradient)
      zero(acc);
      const int nts = (vbase_s - c0 + 31) / 32 < NT ? (vbase_s - c0 + 31) / 32 : NT;
      matmul(acc, nts, HID, 0, L.w1n, A.in_c, c0, A.in_c);    // NT tiles per pass: two MFMAs per k-pair do not cover the loads' latency
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int b0 = c0 + 32 * nt;
        if (b0 >= vbase_s) break;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) SL(HID + l31, 32 * mt + 8 * (r >> 2) + 4 * kk + (r & 3)) = acc[mt][nt][r];
        wave_sync_g();
        const int b1 = b0 + 32 < vbase_s ? b0 + 32 : vbase_s;
        for (int t = b0; t < b1; ++t) {                        // lane = sample; t is uniform
          int f;
          float coef;
          if (t < D) { f = t; coef = 1.f; }
          else if (t < base_s) continue;                        // the view direction itself
          else if (t < base_c) { const int e = t - base_s, q = e % A.fea_pe; f = e / A.fea_pe; coef = ldexpf(xr[t + D * A.fea_pe], q); }
          else { const int e = t - base_c, q = e % A.fea_pe; f = e / A.fea_pe; coef = -ldexpf(xr[t - D * A.fea_pe], q); }
          SL(HID + 32 + f, lane) += coef * SL(HID + (t - b0), lane);
        }
        wave_sync_g();
      }
    }
    if (valid) {
#pragma unroll
      for (int f = 0; f < 32; ++f) {
        A.dfe[m * 64 + 32 * g + f] = SL(HID + 32 + f, lane);
        A.dfe[m * 64 + 32 * (1 - g) + f] = 0.f;
      }
    }
    // dv = dfe basis_g, basis [g][32][3 C]: a sample's row of A is zero for the grid it does not belong to
    const unsigned long long gmask = __ballot(g != 0);
    const bool any_yin = ~gmask != 0ull, any_yang = gmask != 0ull;
    const bool yang0 = (gmask >> l31) & 1ull, yang1 = (gmask >> (32 + l31)) & 1ull;
    wave_sync_g();
    for (int c0 = 0; c0 < ncol; c0 += 32 * NT) {
      zero(acc);
      gen_pipelined<2, 2 * NT>(16,
        [&](int it, float (&bq)[2 * NT]) {
          const float* bw = gw + L.basisn + (int64_t)(2 * it + kk) * ncol;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int c = c0 + 32 * nt + l31;
            const bool okc = c < ncol;
            const int cc = okc ? c : ncol - 1;
            const float y0 = any_yin ? bw[cc] : 0.f, y1 = any_yang ? bw[(int64_t)32 * ncol + cc] : 0.f;
            bq[2 * nt] = okc ? y0 : 0.f; bq[2 * nt + 1] = okc ? y1 : 0.f;
          }
        },
        [&](int it, const float (&bq)[2 * NT]) {
          const int tt = 2 * it + kk;
          const float x0 = SL(HID + 32 + tt, l31), x1 = SL(HID + 32 + tt, 32 + l31);
          const float a00 = yang0 ? 0.f : x0, a01 = yang1 ? 0.f : x1, a10 = yang0 ? x0 : 0.f, a11 = yang1 ? x1 : 0.f;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (c0 + 32 * nt >= ncol) break;
            if (any_yin) {
              acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, bq[2 * nt], acc[0][nt], 0, 0, 0);
              acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, bq[2 * nt], acc[1][nt], 0, 0, 0);
            }
            if (any_yang) {
              acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, bq[2 * nt + 1], acc[0][nt], 0, 0, 0);
              acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, bq[2 * nt + 1], acc[1][nt], 0, 0, 0);
            }
          }
        });
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = c0 + 32 * nt + l31;
        if (c0 + 32 * nt >= ncol) break;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t ms = unit * 64 + 32 * mt + 8 * (r >> 2) + 4 * kk + (r & 3);
            if (c < ncol && (full || ms < A.M)) A.dv[gen_dv_index(ms, c, A.ldv)] = acc[mt][nt][r];
          }
      }
    }
    wave_sync_g();   // the next unit stages into the slab
  }
}
#else
template <int HID>
__global__ __launch_bounds__(128) void k_shade_generic_bwd(GenBwdArgs A) {
  __shared__ float slab[2][HID][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float (*sl)[64] = slab[wv];
  const GenLayout L = gen_layout(A.in_c, A.head == EGO_HEAD_RGB ? 0 : HID, A.n_comp);
  const int64_t n_units = (A.M + 63) >> 6;
  const int D = A.app_dim;
  for (int64_t unit = (int64_t)blockIdx.x * 2 + wv; unit < n_units; unit += (int64_t)gridDim.x * 2) {
    const int64_t m_raw = unit * 64 + lane;
    const bool valid = m_raw < A.M;
    const int64_t m = valid ? m_raw : A.M - 1;
    const int g = ((const f32x4*)A.coords)[m].w != 0.f;
    if (A.head == EGO_HEAD_RGB) {   // colour = features: dfe = dL/d rgb_sample, then dv = B_g^T dfe as below (basisT [g][col][32])
      float dfe[3] = {A.dc[m * 3], A.dc[m * 3 + 1], A.dc[m * 3 + 2]};
      if (valid) {
#pragma unroll
        for (int f = 0; f < 32; ++f) {
          A.dfe[m * 64 + 32 * g + f] = f < 3 ? dfe[f] : 0.f;
          A.dfe[m * 64 + 32 * (1 - g) + f] = 0.f;
        }
      }
      const int ncol = 3 * A.n_comp;
      for (int col = 0; col < ncol; ++col) {
        const float* b0 = A.gp + L.basis + (int64_t)col * 32;
        const float* b1 = b0 + (int64_t)ncol * 32;
        float s = 0.f;
#pragma unroll
        for (int f = 0; f < 3; ++f) s = fmaf(g ? b1[f] : b0[f], dfe[f], s);
        if (valid) A.dv[gen_dv_index(m, col, A.ldv)] = s;
      }
      continue;
    }
    float d_o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float y = A.rgb[m * 3 + c];
      d_o[c] = A.dc[m * 3 + c] * y * (1.f - y);       // sigmoid'
    }
    if (valid) { A.dc[m * 3] = d_o[0]; A.dc[m * 3 + 1] = d_o[1]; A.dc[m * 3 + 2] = d_o[2]; }
    float dh[HID];
    const float* w3 = A.gp + L.w3;
#pragma unroll
    for (int j = 0; j < HID; ++j) {
      const float s = w3[j] * d_o[0] + w3[HID + j] * d_o[1] + w3[2 * HID + j] * d_o[2];
      dh[j] = A.h2[m * A.ldh + j] > 0.f ? s : 0.f;    // threshold backward of torch.nn.ReLU
      if (valid) A.dh2[m * HID + j] = dh[j];
    }
    // dh1[k] = relu'(h1[k]) sum_j W2[j][k] dh2[j]; W2T row k holds W2[:, k]
    for (int k = 0; k < HID; ++k) {
      const float* wrow = A.gp + L.w2t + (int64_t)k * HID;   // uniform: scalar loads
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < HID; ++j) s = fmaf(wrow[j], dh[j], s);
      s = A.h1[m * A.ldh + k] > 0.f ? s : 0.f;
      sl[k][lane] = s;
      if (valid) A.dh1[m * HID + k] = s;
    }
#pragma unroll
    for (int j = 0; j < HID; ++j) dh[j] = sl[j][lane];      // dh1 in registers
    auto dx = [&](int t) {                                   // dL/d(MLP input t) = sum_k W1[k][t] dh1[k]; W1T row t holds W1[:, t]
      const float* wrow = A.gp + L.w1t + (int64_t)t * HID;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < HID; ++j) s = fmaf(wrow[j], dh[j], s);
      return s;
    };
    const int base_s = D + 3, base_c = base_s + D * A.fea_pe;
    const float* xr = A.x + m * A.ldx;
    for (int f = 0; f < D; ++f) {
      float acc = dx(f);
      float fr = 1.f;
      for (int q = 0; q < A.fea_pe; ++q, fr *= 2.f) {
        const int ts = base_s + f * A.fea_pe + q, tc = base_c + f * A.fea_pe + q;
        acc += fr * (xr[tc] * dx(ts) - xr[ts] * dx(tc));    // d sin(f w) = w cos(f w) df, d cos(f w) = -w sin(f w) df
      }
      sl[f][lane] = acc;
    }
    float dfe[32];
#pragma unroll
    for (int f = 0; f < 32; ++f) dfe[f] = f < D ? sl[f][lane] : 0.f;
    if (valid) {
#pragma unroll
      for (int f = 0; f < 32; ++f) {
        A.dfe[m * 64 + 32 * g + f] = dfe[f];
        A.dfe[m * 64 + 32 * (1 - g) + f] = 0.f;
      }
    }
    // dv[col] = sum_f basis_g[f][col] dfe[f]; basisT [g][col][32]
    const int ncol = 3 * A.n_comp;
    for (int col = 0; col < ncol; ++col) {
      const float* b0 = A.gp + L.basis + (int64_t)col * 32;
      const float* b1 = b0 + (int64_t)ncol * 32;
      float s = 0.f;
#pragma unroll
      for (int f = 0; f < 32; ++f) s = fmaf(g ? b1[f] : b0[f], dfe[f], s);
      if (valid) A.dv[gen_dv_index(m, col, A.ldv)] = s;
    }
  }
}
#endif

// backward of the VM lookups for any component count (multiple of 4): thread = (sample, plane); float atomics per tap and channel
struct GenScatterArgs {
  DevField F;
  float* gplane[2][3];
  float* gline[2][3];
  const float* coords;   // [M][4]
  const float* d;        // ldd == 0: dfeat [M] (density: relu per plane, EgoNeRF.py:340,346); else dv [M][ldd], column = plane * C + channel
  int64_t M;
  int32_t C, ldd;
};

// sum over the 16 lanes of a row on the DPP path (every lane gets the total)
__device__ __forceinline__ float gen_row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xb1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4e, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));   // row_mirror
  return v;
}

// A 16-lane group per (sample, plane), lane = channel (16 at a time): every gather and every atomic of a group is one 64-byte line, as in
// the tuned k_vm_scatter (ego_train.inc) but without its run merging.  (Rounds 2-5 had a THREAD per (sample, plane): 64 lanes = 64 texels
// = 64 lines per atomic instruction, 105 ms for the shipped tables at 8192 x 256 - 71 % of a training step of the other heads.)
__global__ __launch_bounds__(256) void k_scatter_generic(GenScatterArgs A) {
#pragma clang fp contract(fast)
  const int c16 = threadIdx.x & 15, C = A.C;
  const bool dens = A.ldd == 0;
  const int64_t n_items = A.M * 3, n_groups = (int64_t)gridDim.x * 16;
  for (int64_t item = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; item < n_items; item += n_groups) {
    const int64_t m = item / 3;
    const int i = (int)(item - m * 3);
    float ds = 0.f;
    if (dens) {
      ds = A.d[m];
      if (ds == 0.f) continue;
    }
    const f32x4 cc = ((const f32x4*)A.coords)[m];
    const int g = cc.w != 0.f;
    const VMTaps t = vm_setup(cc.x, cc.y, cc.z, A.F.res);
    const Lin1 X = t.ax[i == 2 ? 1 : 0], Y = t.ax[i == 0 ? 1 : 2], Ln = t.ax[2 - i];
    const int W = A.F.res[i == 2 ? 1 : 0];
    const float* P = g ? A.F.plane[1][i] : A.F.plane[0][i];
    const float* Lp = g ? A.F.line[1][i] : A.F.line[0][i];
    float* GP = g ? A.gplane[1][i] : A.gplane[0][i];
    float* GL = g ? A.gline[1][i] : A.gline[0][i];
    const int64_t o00 = ((int64_t)Y.i0 * W + X.i0) * C, o01 = ((int64_t)Y.i0 * W + X.i1) * C, o10 = ((int64_t)Y.i1 * W + X.i0) * C,
                  o11 = ((int64_t)Y.i1 * W + X.i1) * C, ol0 = (int64_t)Ln.i0 * C, ol1 = (int64_t)Ln.i1 * C;
    const float w00 = __fmul_rn(Y.w0, X.w0), w01 = __fmul_rn(Y.w0, X.w1), w10 = __fmul_rn(Y.w1, X.w0), w11 = __fmul_rn(Y.w1, X.w1);
    if (dens) {   // relu per plane (EgoNeRF.py:340,346): the gradient passes where this plane's sum over channels is positive
      float dot = 0.f;
      for (int c0 = 0; c0 < C; c0 += 16) {
        const int ch = c0 + c16;
        if (ch < C) {
          const float pv = P[o00 + ch] * w00 + P[o01 + ch] * w01 + P[o10 + ch] * w10 + P[o11 + ch] * w11;
          const float lv = Lp[ol0 + ch] * Ln.w0 + Lp[ol1 + ch] * Ln.w1;
          dot += pv * lv;
        }
      }
      dot = gen_row_sum16(dot);
      if (!(dot > 0.f)) continue;
    }
    for (int c0 = 0; c0 < C; c0 += 16) {
      const int ch = c0 + c16;
      if (ch >= C) continue;
      const float pv = P[o00 + ch] * w00 + P[o01 + ch] * w01 + P[o10 + ch] * w10 + P[o11 + ch] * w11;
      const float lv = Lp[ol0 + ch] * Ln.w0 + Lp[ol1 + ch] * Ln.w1;
      const float di = dens ? ds : A.d[m * A.ldd + i * C + ch];
      const float gp = di * lv, gl = di * pv;
      if (gp != 0.f) {
        if (w00 != 0.f) unsafeAtomicAdd(GP + o00 + ch, gp * w00);
        if (w01 != 0.f) unsafeAtomicAdd(GP + o01 + ch, gp * w01);
        if (w10 != 0.f) unsafeAtomicAdd(GP + o10 + ch, gp * w10);
        if (w11 != 0.f) unsafeAtomicAdd(GP + o11 + ch, gp * w11);
      }
      if (gl != 0.f) {
        if (Ln.w0 != 0.f) unsafeAtomicAdd(GL + ol0 + ch, gl * Ln.w0);
        if (Ln.w1 != 0.f) unsafeAtomicAdd(GL + ol1 + ch, gl * Ln.w1);
      }
    }
  }
}

// ---- generic march: the round-2 structure without teams (lane = sample gathers its own taps), any C % 4 == 0 -----------------
struct GenMarchArgs {
  DevCoords c;
  DevField F;
  const float* rays; const float* z_in; const float* r_sched; const float* jitter;
  float* z_out; float* alpha; float* weight; float* bg; float* coords_out; float* sigma_out; uint8_t* tile_active;
  const uint8_t* occ; int32_t occ_res[3];
  int64_t N;
  int32_t S, C, alpha_stride, softplus;
  float near_, shift, dscale, term_eps, shade_above;
};

__device__ __forceinline__ float gen_occ(const GenMarchArgs& A, int g, float a_r, float a_th, float a_ph) {
  const Lin1 X = lin_setup(a_r, A.occ_res[0]), Y = lin_setup(a_th, A.occ_res[1]), Z = lin_setup(a_ph, A.occ_res[2]);
  const uint8_t* V = A.occ + (int64_t)g * A.occ_res[0] * A.occ_res[1] * A.occ_res[2];
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int ix = (k & 1) ? X.i1 : X.i0, iy = (k & 2) ? Y.i1 : Y.i0, iz = (k & 4) ? Z.i1 : Z.i0;
    const float w = ((k & 1) ? X.w1 : X.w0) * ((k & 2) ? Y.w1 : Y.w0) * ((k & 4) ? Z.w1 : Z.w0);
    v += w * (float)V[((int64_t)iz * A.occ_res[1] + iy) * A.occ_res[0] + ix];
  }
  return v;
}

__global__ __launch_bounds__(256) void k_march_generic(GenMarchArgs A) {
  __shared__ float lut[1024];
  for (int i = threadIdx.x; i < A.c.n_lut; i += blockDim.x) lut[i] = A.c.r_lut[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= A.N) return;
  const float* R = A.rays + ray * 6;
  const float ox = R[0], oy = R[1], oz = R[2], dx = R[3], dy = R[4], dz = R[5];
  const int S = A.S, C = A.C;
  float carry = 1.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = min(s0 + lane, S - 1);
    const bool ok = (s0 + lane) < S;
    const int sn = (s < S - 1) ? s + 1 : s - 1;
    float z, zn;
    if (A.z_in) { z = A.z_in[ray * S + s]; zn = A.z_in[ray * S + sn]; }
    else {
      const auto zz = [&](int k) {
        float r = A.r_sched[k];
        if (A.jitter) {
          const float step = (k < S - 1) ? __fsub_rn(A.r_sched[k + 1], r) : __fsub_rn(r, A.r_sched[S - 2]);
          r = __fadd_rn(r, __fmul_rn(step, A.jitter[ray * S + k]));
        }
        return __fadd_rn(A.near_, r);
      };
      z = zz(s); zn = zz(sn);
    }
    const float dist = (s < S - 1) ? __fsub_rn(zn, z) : __fsub_rn(z, zn);
    const float px = __fadd_rn(ox, __fmul_rn(dx, z)), py = __fadd_rn(oy, __fmul_rn(dy, z)), pz = __fadd_rn(oz, __fmul_rn(dz, z));
    const YinYang y = yinyang_from_xyz(px, py, pz, A.c);
    const float a_r = normalize_r(y.r, lut, A.c.n_lut, A.c.n_r);
    const float a_th = normalize_ang(y.th, A.c.th_near, A.c.th_inv);
    const float a_ph = normalize_ang(y.ph, A.c.ph_near, A.c.ph_inv);
    const bool occupied = !A.occ || gen_occ(A, y.yang, a_r, a_th, a_ph) > 0.f;
    float sg = 0.f;
    if (occupied) {
#pragma clang fp contract(fast)
      const VMTaps t = vm_setup(a_r, a_th, a_ph, A.F.res);
      const int g = y.yang;
      float feat = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const Lin1 X = t.ax[vm_plane_x(i)], Y = t.ax[vm_plane_y(i)], Ln = t.ax[vm_line_ax(i)];
        const int W = A.F.res[vm_plane_x(i)];
        const float* P = g ? A.F.plane[1][i] : A.F.plane[0][i];
        const float* Lp = g ? A.F.line[1][i] : A.F.line[0][i];
        const float* p00 = P + ((int64_t)Y.i0 * W + X.i0) * C;
        const float* p01 = P + ((int64_t)Y.i0 * W + X.i1) * C;
        const float* p10 = P + ((int64_t)Y.i1 * W + X.i0) * C;
        const float* p11 = P + ((int64_t)Y.i1 * W + X.i1) * C;
        const float* l0 = Lp + (int64_t)Ln.i0 * C;
        const float* l1 = Lp + (int64_t)Ln.i1 * C;
        const float w00 = __fmul_rn(Y.w0, X.w0), w01 = __fmul_rn(Y.w0, X.w1), w10 = __fmul_rn(Y.w1, X.w0), w11 = __fmul_rn(Y.w1, X.w1);
        float dot = 0.f;
        for (int c4 = 0; c4 < C; c4 += 4) {
          const f32x4 pv = *(const f32x4*)(p00 + c4) * w00 + *(const f32x4*)(p01 + c4) * w01 + *(const f32x4*)(p10 + c4) * w10 +
                           *(const f32x4*)(p11 + c4) * w11;
          const f32x4 lv = *(const f32x4*)(l0 + c4) * Ln.w0 + *(const f32x4*)(l1 + c4) * Ln.w1;
          const f32x4 mm = pv * lv;
          dot += (mm.x + mm.y) + (mm.z + mm.w);
        }
        feat += fmaxf(dot, 0.f);
      }
      sg = A.softplus ? softplus_shift(feat, A.shift) : fmaxf(feat, 0.f);
    }
    const float a = ok ? alpha_from(sg * __fmul_rn(dist, A.dscale)) : 0.f;
    const float tt = ok ? __fadd_rn(__fsub_rn(1.f, a), 1e-10f) : 1.f;
    const float inc = wave_scan_mul(tt, lane);
    float exc = __shfl_up(inc, 1, 64);
    if (lane == 0) exc = 1.f;
    const float T = carry * exc;
    const float wgt = (A.term_eps > 0.f && T < A.term_eps) ? 0.f : a * T;
    if (A.tile_active) {
      const unsigned long long nz = __ballot(ok && wgt > A.shade_above);
      const int64_t o = ray * S + s;
      if ((S & 31) == 0) {
        if (ok && (lane & 31) == 0) A.tile_active[o >> 5] = (nz >> (lane & 32) & 0xffffffffull) != 0ull ? 1 : 0;
      } else if (ok && (nz >> (lane & 32) & 0xffffffffull) != 0ull && ((lane & 31) == 0 || (o & 31) == 0)) {
        A.tile_active[o >> 5] = 1;
      }
    }
    if (ok) {
      const int64_t o = ray * S + s;
      if (A.z_out) A.z_out[o] = z;
      if (A.coords_out) ((f32x4*)A.coords_out)[o] = f32x4{a_r, a_th, a_ph, y.yang ? 1.f : 0.f};
      if (A.sigma_out) A.sigma_out[o] = sg;
      if (A.alpha) A.alpha[ray * A.alpha_stride + s] = a;
      if (A.weight) A.weight[o] = wgt;
    }
    carry *= __shfl(inc, 63, 64);
  }
  if (A.alpha && lane < A.alpha_stride - S) A.alpha[ray * A.alpha_stride + S + lane] = 1.f;
  if (A.bg && lane == 0) A.bg[ray] = carry;
}

int check_generic_shape(const ego_scene* sc, const char* who, bool need_tables, bool need_mlp, bool need_packed = true) {
  if (!sc) return ego_fail(EGO_E_BADARG, "%s: null scene", who);
  if (sc->app_dim < 1 || sc->app_dim > 32) return ego_fail(EGO_E_UNSUPPORTED, "%s: app_dim %d (supported: 1..32)", who, sc->app_dim);
  const int C = sc->app.n_comp;
  if (C < 4 || C > 48 || (C & 3)) return ego_fail(EGO_E_UNSUPPORTED, "%s: appearance n_comp %d (supported: multiples of 4 up to 48)", who, C);
  if (sc->head != EGO_HEAD_MLP_FEA && sc->head != EGO_HEAD_RGB) return ego_fail(EGO_E_BADARG, "%s: scene.head %d (EGO_HEAD_MLP_FEA or EGO_HEAD_RGB)", who, sc->head);
  if (sc->head == EGO_HEAD_RGB) {   // tensorBase.py:194-196: `assert self.app_dim == 3`; no MLP
    if (sc->app_dim != 3) return ego_fail(EGO_E_BADARG, "%s: the RGB head needs app_dim == 3 (got %d)", who, sc->app_dim);
    if (sc->mlp_in != 0 || sc->mlp_hidden != 0) return ego_fail(EGO_E_BADARG, "%s: the RGB head has no MLP (mlp_in / mlp_hidden must be 0)", who);
    need_mlp = false;
  } else if (sc->mlp_hidden != 64 && sc->mlp_hidden != 128) return ego_fail(EGO_E_UNSUPPORTED, "%s: featureC %d (supported: 64, 128)", who, sc->mlp_hidden);
  if (need_tables) {
    for (int g = 0; g < 2; ++g)
      for (int i = 0; i < 3; ++i)
        if (!sc->app.plane[g][i] || !sc->app.line[g][i]) return ego_fail(EGO_E_BADARG, "%s: null appearance table", who);
    if (sc->app.res[0] < 2 || sc->app.res[1] < 2 || sc->app.res[2] < 2) return ego_fail(EGO_E_BADARG, "%s: appearance resolution < 2", who);
  }
  if (need_mlp) {
    if (sc->view_pe < 0 || sc->view_pe > 8 || sc->fea_pe < 0 || sc->fea_pe > 8)
      return ego_fail(EGO_E_UNSUPPORTED, "%s: view_pe %d / fea_pe %d (supported: 0..8)", who, sc->view_pe, sc->fea_pe);
    const int in_c = 2 * sc->view_pe * 3 + 2 * sc->fea_pe * sc->app_dim + 3 + sc->app_dim;
    if (sc->mlp_in != in_c) return ego_fail(EGO_E_BADARG, "%s: mlp_in %d does not match the encoding widths (%d)", who, sc->mlp_in, in_c);
  }
  if (need_packed && !sc->packed) return ego_fail(EGO_E_BADARG, "%s: scene.packed is null (call ego_pack_mlp first)", who);
  return EGO_OK;
}

}  // namespace

bool ego_shape_is_tuned(const ego_scene* sc) {
  return sc->head == EGO_HEAD_MLP_FEA && sc->app_dim == 27 && sc->app.n_comp == 48 && sc->mlp_in == 150 && sc->mlp_hidden == 128 && sc->view_pe == 2 &&
         sc->fea_pe == 2;
}

int64_t ego_generic_packed_floats(const ego_scene* sc) {
  return gen_layout(sc->mlp_in, sc->mlp_hidden, sc->app.n_comp).total;
}

int ego_generic_pack(const ego_scene* sc, float* out, void* stream) {
  if (int e = check_generic_shape(sc, "pack_mlp", false, true, false)) return e;
  const int64_t n = ego_generic_packed_floats(sc);
  const bool mlp = sc->head != EGO_HEAD_RGB;   // the RGB head's blob is [4 zeros | basisT]
  k_generic_pack<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(sc->mlp_w[0], sc->mlp_b[0], sc->mlp_w[1], sc->mlp_b[1], sc->mlp_w[2],
                                                                              mlp ? sc->mlp_b[2] : nullptr, sc->basis[0], sc->basis[1], sc->mlp_in, sc->mlp_hidden,
                                                                              sc->app.n_comp, sc->app_dim, out);
  return ego_launch_status("k_generic_pack");
}

template <int MODE>
static int launch_shade(const ego_scene* sc, GenShadeArgs& a, hipStream_t st) {
  const int64_t units = (a.M + 63) >> 6;
  const unsigned grid = (unsigned)((units + 1) / 2 < 2048 ? (units + 1) / 2 : 2048);
  if (sc->mlp_hidden != 128) k_shade_generic<64, MODE><<<grid, 128, 0, st>>>(a);   // 64, or 0 = the RGB head (leaves before the MLP)
  else k_shade_generic<128, MODE><<<grid, 128, 0, st>>>(a);
  return ego_launch_status("k_shade_generic");
}

static void fill_common(const ego_scene* sc, GenShadeArgs& a) {
  a.F = make_field(sc->app);
  a.gp = sc->packed;
  a.app_dim = sc->app_dim; a.n_comp = sc->app.n_comp; a.in_c = sc->mlp_in; a.view_pe = sc->view_pe; a.fea_pe = sc->fea_pe; a.head = sc->head;
}

int ego_generic_shade(const ego_scene* sc, const float* rays, const float* coords, int64_t N, int32_t S, float* rgb, const uint8_t* tile_active,
                      void* stream) {
  if (int e = check_generic_shape(sc, "shade", true, true)) return e;
  if (!coords) return ego_fail(EGO_E_BADARG, "shade: coords (from ego_march_density) is required for this model shape");
  GenShadeArgs a{};
  fill_common(sc, a);
  a.rays = rays; a.coords = coords; a.out = rgb; a.tile_active = tile_active; a.M = N * (int64_t)S; a.S = S;
  return launch_shade<G_SHADE>(sc, a, (hipStream_t)stream);
}

int ego_generic_app_feature(const ego_scene* sc, const float* c7n, int64_t M, float* out, void* stream) {
  if (int e = check_generic_shape(sc, "app_feature", true, false)) return e;
  GenShadeArgs a{};
  fill_common(sc, a);
  a.c7n = c7n; a.out = out; a.M = M; a.S = 1;
  return launch_shade<G_APP>(sc, a, (hipStream_t)stream);
}

int ego_generic_mlp_fea(const ego_scene* sc, const float* viewdirs, const float* feat, int64_t M, float* rgb, void* stream) {
  if (int e = check_generic_shape(sc, "mlp_fea", false, true)) return e;
  if (sc->head == EGO_HEAD_RGB) return ego_fail(EGO_E_UNSUPPORTED, "mlp_fea: the RGB head has no MLP (RGBRender returns the features)");
  GenShadeArgs a{};
  fill_common(sc, a);
  a.feat = feat; a.dirs = viewdirs; a.out = rgb; a.M = M; a.S = 1;
  return launch_shade<G_MLP>(sc, a, (hipStream_t)stream);
}

int ego_generic_march(const ego_scene* sc, const ego_vm_field& f, bool fine_lut, const float* rays, int64_t N, int32_t S, const float* z_in,
                      const float* r_sched, const float* jitter, float near_, const uint8_t* occ, float* z_out, float* alpha, int32_t alpha_stride,
                      float* weight, float* bg_weight, float* coords_out, float* sigma_out, uint8_t* tile_active, void* stream) {
  const int C = f.n_comp;
  if (C < 4 || C > 48 || (C & 3)) return ego_fail(EGO_E_UNSUPPORTED, "march_density: n_comp %d (supported: multiples of 4 up to 48)", C);
  GenMarchArgs a{};
  a.c = make_coords(*sc, fine_lut); a.F = make_field(f);
  a.rays = rays; a.z_in = z_in; a.r_sched = r_sched; a.jitter = jitter; a.z_out = z_out; a.alpha = alpha; a.weight = weight; a.bg = bg_weight;
  a.coords_out = coords_out; a.sigma_out = sigma_out; a.tile_active = tile_active; a.occ = occ;
  a.occ_res[0] = sc->occ_res[0]; a.occ_res[1] = sc->occ_res[1]; a.occ_res[2] = sc->occ_res[2];
  a.N = N; a.S = S; a.C = C; a.alpha_stride = alpha_stride; a.softplus = sc->act_softplus;
  a.near_ = near_; a.shift = sc->density_shift; a.dscale = sc->distance_scale; a.term_eps = sc->term_eps; a.shade_above = fmaxf(sc->weight_thres, 0.f);
  k_march_generic<<<(unsigned)((N + 3) / 4), 256, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_march_generic");
}

// ---- training entry points for the other model shapes (declared in include/egonerf_hip.h) ---------------------------------------
extern "C" {

int ego_shade_train_generic(const ego_scene* sc, const float* rays, const float* coords, int64_t N, int32_t S, float* rgb, float* x, int32_t ldx,
                            float* h1, float* h2, int32_t ldh, float* v, int32_t ldv, void* stream) {
  EGO_TRACE("ego_shade_train_generic");
  EGO_REQUIRE(N >= 0 && S >= 1 && N * (int64_t)S < (1ll << 31), "shade_train_generic: bad size");
  if (N == 0) return EGO_OK;
  if (int e = check_generic_shape(sc, "shade_train_generic", true, true)) return e;
  const bool rgb_head = sc->head == EGO_HEAD_RGB;   // no MLP: x / h1 / h2 are not written and may be null
  EGO_REQUIRE(rays && coords && rgb && v && (rgb_head || (x && h1 && h2)), "shade_train_generic: null argument");
  EGO_REQUIRE(ldx >= sc->mlp_in && ldh >= sc->mlp_hidden && ldv >= 3 * sc->app.n_comp && (ldv & 3) == 0 && ((uintptr_t)v & 15) == 0,
              "shade_train_generic: leading dimensions too small (or v not 16-byte aligned / ldv not a multiple of 4)");
  GenShadeArgs a{};
  fill_common(sc, a);
  a.rays = rays; a.coords = coords; a.out = rgb; a.M = N * (int64_t)S; a.S = S;
  a.dump_x = x; a.dump_h1 = h1; a.dump_h2 = h2; a.dump_v = v; a.ldx = ldx; a.ldh = ldh; a.ldv = ldv;
  const int64_t units = (a.M + 63) >> 6;
  const unsigned grid = (unsigned)((units + 1) / 2 < 2048 ? (units + 1) / 2 : 2048);
  if (sc->mlp_hidden != 128) k_shade_generic<64, G_SHADE, true><<<grid, 128, 0, (hipStream_t)stream>>>(a);
  else k_shade_generic<128, G_SHADE, true><<<grid, 128, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_shade_generic<DUMP>");
}

int ego_shade_backward_generic(const ego_scene* sc, const float* coords, float* dc, const float* rgb, const float* x, int32_t ldx, const float* h1,
                               const float* h2, int32_t ldh, float* dh2, float* dh1, float* dfe64, float* dv, int32_t ldv, int64_t N, int32_t S,
                               void* stream) {
  EGO_TRACE("ego_shade_backward_generic");
  EGO_REQUIRE(N >= 0 && S >= 1 && N * (int64_t)S < (1ll << 31), "shade_backward_generic: bad size");
  if (N == 0) return EGO_OK;
  if (int e = check_generic_shape(sc, "shade_backward_generic", false, true)) return e;
  EGO_REQUIRE(coords && dc && dfe64 && dv && (sc->head == EGO_HEAD_RGB || (rgb && x && h1 && h2 && dh2 && dh1)), "shade_backward_generic: null argument");
  EGO_REQUIRE(ldx >= sc->mlp_in && ldh >= sc->mlp_hidden && (ldv >= 3 * sc->app.n_comp || (ldv == 0 && sc->app.n_comp == 48)),
              "shade_backward_generic: leading dimensions too small (ldv == 0 = the blocked dv of the tuned scatters: 48 components only)");
  GenBwdArgs a{};
  a.gp = sc->packed; a.coords = coords; a.dc = dc; a.rgb = rgb; a.x = x; a.h1 = h1; a.h2 = h2; a.dh2 = dh2; a.dh1 = dh1; a.dfe = dfe64; a.dv = dv;
  a.M = N * (int64_t)S; a.app_dim = sc->app_dim; a.n_comp = sc->app.n_comp; a.in_c = sc->mlp_in; a.view_pe = sc->view_pe; a.fea_pe = sc->fea_pe;
  a.ldx = ldx; a.ldh = ldh; a.ldv = ldv; a.head = sc->head;
  const int64_t units = (a.M + 63) >> 6;
#if EGO_GENERIC_MFMA
  const unsigned grid = (unsigned)(units < 4096 ? units : 4096);   // one wave per workgroup (its slab: 32 / 48 KB)
  if (sc->mlp_hidden != 128) k_shade_generic_bwd<64><<<grid, 64, 0, (hipStream_t)stream>>>(a);
  else k_shade_generic_bwd<128><<<grid, 64, 0, (hipStream_t)stream>>>(a);
#else
  const unsigned grid = (unsigned)((units + 1) / 2 < 2048 ? (units + 1) / 2 : 2048);
  if (sc->mlp_hidden != 128) k_shade_generic_bwd<64><<<grid, 128, 0, (hipStream_t)stream>>>(a);
  else k_shade_generic_bwd<128><<<grid, 128, 0, (hipStream_t)stream>>>(a);
#endif
  return ego_launch_status("k_shade_generic_bwd");
}

int ego_scatter_generic(const ego_vm_field* field, const ego_vm_grad* grad, const float* coords, const float* d, int32_t ldd, int64_t N, int32_t S,
                        void* stream) {
  EGO_TRACE("ego_scatter_generic");
  EGO_REQUIRE(field && grad && N >= 0 && S >= 1 && ldd >= 0, "scatter_generic: null argument or bad size");
  if (N == 0) return EGO_OK;
  EGO_REQUIRE(coords && d, "scatter_generic: null argument");
  const int C = field->n_comp;
  if (C < 4 || C > 48 || (C & 3)) return ego_fail(EGO_E_UNSUPPORTED, "scatter_generic: n_comp %d (supported: multiples of 4 up to 48)", C);
  EGO_REQUIRE(ldd == 0 || (ldd >= 3 * C && (ldd & 3) == 0 && ((uintptr_t)d & 15) == 0), "scatter_generic: ldd must be 0 (density) or >= 3 C, a multiple of 4");
  GenScatterArgs a{};
  a.F = make_field(*field);
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i) {
      EGO_REQUIRE(field->plane[g][i] && field->line[g][i] && grad->plane[g][i] && grad->line[g][i], "scatter_generic: null table");
      a.gplane[g][i] = grad->plane[g][i]; a.gline[g][i] = grad->line[g][i];
    }
  a.coords = coords; a.d = d; a.M = N * (int64_t)S; a.C = C; a.ldd = ldd;
  const int64_t n = a.M * 3, blocks = (n + 15) / 16;   // 16 groups of 16 lanes per workgroup
  k_scatter_generic<<<(unsigned)(blocks < (1 << 20) ? (blocks > 0 ? blocks : 1) : (1 << 20)), 256, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_scatter_generic");
}

}  // extern "C"
