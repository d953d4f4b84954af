// Probe (round 4): can the two correction terms of the shade kernel's split products run on the fp6 (e2m3) path of
// v_mfma_scale_f32_32x32x64_f8f6f4 instead of the fp8 one?
//  1. v_cvt_scalef32_2xpk16_fp6_f32 / v_cvt_scalef32_pk32_fp6_f16: element order of the 32 six-bit results, scale, rounding, saturation.
//  2. the MFMA's operand layout with fp6 data (32 K values of a lane = 192 bits in 6 registers) and its block scales.
//  3. matrix-pipe time: fp8 vs fp6 vs fp4 vs the fp16 32x32x16 instruction, 1 and 2 waves per SIMD.
//  4. issue cost of the conversions next to v_cvt_pk_fp8_f32.
// Build: hipcc --offload-arch=gfx950 -O2 tools/fp6_probe.hip -o /tmp/fp6_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

static float e2m3_decode(int c) {
  const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
  const float f = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
  return s ? -f : f;
}

__global__ void k_cvt(const float* x, unsigned* o, float sc) {
  v16f a, b;
  v32h hh;
  for (int i = 0; i < 16; ++i) { a[i] = x[i]; b[i] = x[16 + i]; }
  for (int i = 0; i < 32; ++i) hh[i] = (_Float16)x[i];
  const auto r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, sc);
  const auto r2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hh, sc);
  if (threadIdx.x == 0)
    for (int i = 0; i < 6; ++i) { o[i] = r[i]; o[8 + i] = r2[i]; }
}

// A [32][64] codes (row, k), B [64][32] codes (k, col); candidate layout: lane l holds row / col l % 32, k = 32 (l / 32) + e, element e
// at bits [6 e, 6 e + 6) of the lane's 192-bit operand
__global__ void k_mfma6(const uint8_t* A, const uint8_t* B, float* D, int scale_a, int scale_b) {
  const int l = threadIdx.x, rc = l & 31, kb = l >> 5;
  uint32_t aw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int e = 0; e < 32; ++e) {
    const int k = 32 * kb + e, bit = 6 * e, w = bit >> 5, sh = bit & 31;
    const uint32_t ca = A[rc * 64 + k] & 63u, cb = B[k * 32 + rc] & 63u;
    aw[w] |= ca << sh; bw[w] |= cb << sh;
    if (sh > 26) { aw[w + 1] |= ca >> (32 - sh); bw[w + 1] |= cb >> (32 - sh); }
  }
  v8i a, b;
  for (int w = 0; w < 8; ++w) { a[w] = (int)aw[w]; b[w] = (int)bw[w]; }
  v16f acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 2, 2, 0, scale_a, 0, scale_b);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

// per-lane block scales: lane l passes its own E8M0 byte
__global__ void k_mfma6_lanescale(const uint8_t* A, const uint8_t* B, float* D, const int* sa, const int* sb) {
  const int l = threadIdx.x, rc = l & 31, kb = l >> 5;
  uint32_t aw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int e = 0; e < 32; ++e) {
    const int k = 32 * kb + e, bit = 6 * e, w = bit >> 5, sh = bit & 31;
    const uint32_t ca = A[rc * 64 + k] & 63u, cb = B[k * 32 + rc] & 63u;
    aw[w] |= ca << sh; bw[w] |= cb << sh;
    if (sh > 26) { aw[w + 1] |= ca >> (32 - sh); bw[w + 1] |= cb >> (32 - sh); }
  }
  v8i a, b;
  for (int w = 0; w < 8; ++w) { a[w] = (int)aw[w]; b[w] = (int)bw[w]; }
  v16f acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 2, 2, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

template <int FMT>  // 0 fp8, 2 fp6, 4 fp4, 16: fp16 32x32x16
__global__ void k_time(float* out, int iters) {
  v8i a, b;
  for (int w = 0; w < 8; ++w) { a[w] = threadIdx.x * 17 + w; b[w] = threadIdx.x * 29 + w; }
  h8 ah, bh;
  for (int w = 0; w < 8; ++w) { ah[w] = (_Float16)(threadIdx.x + w); bh[w] = (_Float16)(w * 0.5f); }
  v16f acc[4];
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (FMT == 16) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[q], 0, 0, 0);
      else acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[q], FMT, FMT, 0, 127, 0, 127);
    }
  }
  float s = 0.f;
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// conversion issue cost: KIND 0 = 16 x v_cvt_pk_fp8_f32 (32 values), 1 = one 2xpk16_fp6_f32 (32 values), 2 = one pk32_fp6_f16, 3 = 16 v_max3_f32
template <int KIND>
__global__ void k_cvt_time(float* out, int iters, float sc) {
  v16f a, b;
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.01f + i; b[i] = threadIdx.x * 0.02f - i; }
  v32h hh;
  for (int i = 0; i < 32; ++i) hh[i] = (_Float16)(threadIdx.x * 0.01f + i);
  unsigned accu = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      if (KIND == 0) {
        int w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          w[i] = __builtin_amdgcn_cvt_pk_fp8_f32(a[2 * i], a[2 * i + 1], (int)accu, false);
          w[i] = __builtin_amdgcn_cvt_pk_fp8_f32(b[2 * i], b[2 * i + 1], w[i], true);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) accu ^= (unsigned)w[i];
      } else if (KIND == 1) {
        const auto r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, sc);
        accu ^= r[0] ^ r[5];
      } else if (KIND == 2) {
        const auto r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hh, sc);
        accu ^= r[0] ^ r[5];
      } else {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 2) m = fmaxf(fmaxf(fabsf(a[i]), fabsf(a[i + 1])), m);
#pragma unroll
        for (int i = 0; i < 16; i += 2) m = fmaxf(fmaxf(fabsf(b[i]), fabsf(b[i + 1])), m);
        accu ^= __float_as_uint(m);
      }
      asm volatile("" : "+v"(a), "+v"(b));
      asm volatile("" : "+v"(hh));
    }
  }
  if (accu == 0x12345u) out[threadIdx.x] = 1.f;
}

template <typename F>
static float time_ms(F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  // ---- 1. conversions ----
  float hx[32];
  for (int i = 0; i < 32; ++i) hx[i] = e2m3_decode(i);  // src0[i] = code i, src1[i] = code 16 + i
  float* dx; unsigned* dout;
  (void)hipMalloc(&dx, 256); (void)hipMalloc(&dout, 256);
  (void)hipMemcpy(dx, hx, 128, hipMemcpyHostToDevice);
  for (float sc : {1.0f, 2.0f, 0.5f}) {
    k_cvt<<<1, 64>>>(dx, dout, sc);
    unsigned ho[16];
    (void)hipMemcpy(ho, dout, 64, hipMemcpyDeviceToHost);
    printf("scale %g\n  2xpk16_fp6_f32 codes by output element:", sc);
    for (int e = 0; e < 32; ++e) {
      const int bit = 6 * e, w = bit >> 5, sh = bit & 31;
      uint64_t v = ho[w] | ((uint64_t)(w + 1 < 6 ? ho[w + 1] : 0) << 32);
      printf(" %d", (int)((v >> sh) & 63));
    }
    printf("\n  pk32_fp6_f16   codes by output element:");
    for (int e = 0; e < 32; ++e) {
      const int bit = 6 * e, w = bit >> 5, sh = bit & 31;
      uint64_t v = ho[8 + w] | ((uint64_t)(w + 1 < 6 ? ho[8 + w + 1] : 0) << 32);
      printf(" %d", (int)((v >> sh) & 63));
    }
    printf("\n");
  }
  {  // rounding / saturation / sign
    const float vals[32] = {0.0624f, 0.0626f, 0.19f, 1.06f, 1.07f, 7.4f, 7.6f, 8.0f, 100.f, -0.3f, -7.7f, 1e-9f, 3.1f, 3.4f, 0.99f, -1.49f};
    (void)hipMemcpy(dx, vals, 128, hipMemcpyHostToDevice);
    k_cvt<<<1, 64>>>(dx, dout, 1.0f);
    unsigned ho[16];
    (void)hipMemcpy(ho, dout, 64, hipMemcpyDeviceToHost);
    printf("rounding (2xpk16, scale 1), listing every output element with its decoded value:\n ");
    for (int e = 0; e < 32; ++e) {
      const int bit = 6 * e, w = bit >> 5, sh = bit & 31;
      uint64_t v = ho[w] | ((uint64_t)(w + 1 < 6 ? ho[w + 1] : 0) << 32);
      printf(" [%d]%g", e, e2m3_decode((int)((v >> sh) & 63)));
    }
    printf("\n  inputs:");
    for (int i = 0; i < 16; ++i) printf(" %g", vals[i]);
    printf("\n");
  }
  // ---- 2. MFMA layout ----
  {
    std::vector<uint8_t> hA(32 * 64), hB(64 * 32);
    std::vector<float> fA(32 * 64), fB(64 * 32);
    uint32_t s = 777;
    for (auto i = 0u; i < hA.size(); ++i) { s = s * 1664525u + 1013904223u; hA[i] = (s >> 24) & 63; fA[i] = e2m3_decode(hA[i]); }
    for (auto i = 0u; i < hB.size(); ++i) { s = s * 1664525u + 1013904223u; hB[i] = (s >> 24) & 63; fB[i] = e2m3_decode(hB[i]); }
    uint8_t *dA, *dB; float* dD; int *dsa, *dsb;
    (void)hipMalloc(&dA, hA.size()); (void)hipMalloc(&dB, hB.size()); (void)hipMalloc(&dD, 4096);
    (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256);
    (void)hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB.data(), hB.size(), hipMemcpyHostToDevice);
    for (int cfg = 0; cfg < 3; ++cfg) {
      const int sa = cfg == 1 ? 116 : 127, sb = cfg == 2 ? 130 : 127;
      k_mfma6<<<1, 64>>>(dA, dB, dD, sa, sb);
      float hD[1024];
      (void)hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
      const float es = ldexpf(1.0f, (sa - 127) + (sb - 127));
      double maxerr = 0, maxref = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double ref = 0;
          for (int k = 0; k < 64; ++k) ref += (double)fA[i * 64 + k] * fB[k * 32 + j];
          maxerr = fmax(maxerr, fabs(hD[i * 32 + j] - ref * es)); maxref = fmax(maxref, fabs(ref * es));
        }
      printf("mfma fp6 x fp6 sequential-bit layout, scale_a %d scale_b %d: max |D - ref| = %g (max |ref| %g)\n", sa, sb, maxerr, maxref);
    }
    int hsa[64], hsb[64];
    for (int l = 0; l < 64; ++l) { hsa[l] = 120 + (l % 7); hsb[l] = 125 + (l % 5); }
    (void)hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    k_mfma6_lanescale<<<1, 64>>>(dA, dB, dD, dsa, dsb);
    float hD[1024];
    (void)hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int k = 0; k < 64; ++k) {
          const int la = i + 32 * (k >> 5), lb = j + 32 * (k >> 5);
          ref += (double)fA[i * 64 + k] * fB[k * 32 + j] * ldexp(1.0, hsa[la] - 127 + hsb[lb] - 127);
        }
        maxerr = fmax(maxerr, fabs(hD[i * 32 + j] - ref)); maxref = fmax(maxref, fabs(ref));
      }
    printf("mfma fp6 per-lane scales (lane l scales its own 32 K values): max |D - ref| = %g (max |ref| %g)\n", maxerr, maxref);
  }
  // ---- 3. matrix pipe time ----
  float* dout2;
  (void)hipMalloc(&dout2, 4096);
  const int iters = 20000;
  for (int wv = 1; wv <= 2; ++wv) {
    const int threads = 256 * wv, blocks = 256;
    const double n = (double)iters * 4;  // MFMAs per wave
    const float t16 = time_ms([&] { k_time<16><<<blocks, threads>>>(dout2, iters); });
    const float t8 = time_ms([&] { k_time<0><<<blocks, threads>>>(dout2, iters); });
    const float t6 = time_ms([&] { k_time<2><<<blocks, threads>>>(dout2, iters); });
    const float t4 = time_ms([&] { k_time<4><<<blocks, threads>>>(dout2, iters); });
    printf("%d wave(s) per SIMD: ns per MFMA per wave: f16 32x32x16 %.1f | fp8 32x32x64 %.1f | fp6 %.1f | fp4 %.1f\n", wv, t16 * 1e6 / n, t8 * 1e6 / n,
           t6 * 1e6 / n, t4 * 1e6 / n);
  }
  // ---- 4. conversion issue cost (per 32 values) ----
  for (int wv = 1; wv <= 2; ++wv) {
    const int threads = 256 * wv, blocks = 256, it2 = 20000;
    const double n = (double)it2 * 4;
    const float c0 = time_ms([&] { k_cvt_time<0><<<blocks, threads>>>(dout2, it2, 1.0f); });
    const float c1 = time_ms([&] { k_cvt_time<1><<<blocks, threads>>>(dout2, it2, 1.0f); });
    const float c2 = time_ms([&] { k_cvt_time<2><<<blocks, threads>>>(dout2, it2, 1.0f); });
    const float c3 = time_ms([&] { k_cvt_time<3><<<blocks, threads>>>(dout2, it2, 1.0f); });
    printf("%d wave(s) per SIMD: ns per 32 values per wave: 16 x cvt_pk_fp8_f32 %.1f | 2xpk16_fp6_f32 %.1f | pk32_fp6_f16 %.1f | 32-value |max| %.1f\n", wv,
           c0 * 1e6 / n, c1 * 1e6 / n, c2 * 1e6 / n, c3 * 1e6 / n);
  }
  return 0;
}
