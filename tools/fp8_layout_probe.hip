// Probe (round 2): operand layout, scale semantics and conversions for the fp8 correction terms of the shade kernel.
//  1. v_cvt_pk_fp8_f32: encoding (OCP e4m3fn: 1.0 = 0x38), rounding, saturation, subnormals; word select.
//  2. v_cvt_scalef32_pk_fp8_f32: what the scale operand does.
//  3. v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 operands: A / B lane layout (asymmetric integer data, compared with the CPU),
//     E8M0 scale bytes and opsel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v2s __attribute__((ext_vector_type(2)));

__global__ void k_cvt(const float* x, int n, uint32_t* o, float sc, int ovfl) {
  const int i = threadIdx.x;
  if (i >= n) return;
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);  // MODE.FP16_OVFL: saturate instead of NaN / inf
  int r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], -x[i], r, false);
  o[i] = (uint32_t)r;
  v2s q = {0, 0};
  q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(q, x[i], -x[i], sc, false);
  o[64 + i] = (uint32_t)__builtin_bit_cast(int, q);
}

// A [32][64] (row i, k), B [64][32] (k, col j) as bytes; candidate layout: lane l holds row/col l % 32 and k = 32 * (l / 32) + 0..31,
// byte b of the 32-byte operand = k offset b.
__global__ void k_mfma(const uint8_t* A, const uint8_t* B, float* D, int scale_a, int scale_b) {
  const int l = threadIdx.x, rc = l & 31, kb = l >> 5;
  v8i a, b;
  for (int w = 0; w < 8; ++w) {
    uint32_t av = 0, bv = 0;
    for (int e = 0; e < 4; ++e) {
      const int k = 32 * kb + 4 * w + e;
      av |= (uint32_t)A[rc * 64 + k] << (8 * e);
      bv |= (uint32_t)B[k * 32 + rc] << (8 * e);
    }
    a[w] = (int)av; b[w] = (int)bv;
  }
  v16f acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, scale_a, 0, scale_b);
  // C/D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

static float e4m3_decode(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  if (e == 15 && m == 7) f = NAN;
  return s ? -f : f;
}

int main() {
  float hx[64];
  const float vals[] = {1.0f, 0.5f, 1.0625f, 1.1875f, 448.0f, 500.0f, 1e6f, 0.015625f, 0.001953125f, 0.0009765625f, 0.0029296875f, 3.3f, 0.3f, 17.0f, 240.0f, 0.0f};
  const int n = 16;
  for (int i = 0; i < n; ++i) hx[i] = vals[i];
  float* dx; uint32_t* dout;
  (void)hipMalloc(&dx, 256); (void)hipMalloc(&dout, 1024);
  (void)hipMemcpy(dx, hx, 64, hipMemcpyHostToDevice);
  for (float sc : {1.0f, 2.0f, 0.5f, 2048.0f, 1.0f / 2048.0f}) {
    k_cvt<<<1, 64>>>(dx, n, dout, sc, 0);
    uint32_t ho[128];
    (void)hipMemcpy(ho, dout, 512, hipMemcpyDeviceToHost);
    printf("scale operand %g:\n", sc);
    for (int i = 0; i < n; ++i)
      printf("  x=%-12g cvt_pk: %02x (%g) / %02x   scalef32: %02x (%g)\n", hx[i], ho[i] & 255, e4m3_decode(ho[i] & 255), (ho[i] >> 8) & 255,
             ho[64 + i] & 255, e4m3_decode(ho[64 + i] & 255));
    if (sc != 1.0f) continue;
  }
  {
    k_cvt<<<1, 64>>>(dx, n, dout, 1.0f, 1);
    uint32_t ho[128];
    (void)hipMemcpy(ho, dout, 512, hipMemcpyDeviceToHost);
    printf("with MODE.FP16_OVFL = 1:\n");
    for (int i = 0; i < n; ++i)
      if (hx[i] > 400.f) printf("  x=%-12g cvt_pk: %02x (%g)   scalef32: %02x (%g)\n", hx[i], ho[i] & 255, e4m3_decode(ho[i] & 255), ho[64 + i] & 255, e4m3_decode(ho[64 + i] & 255));
  }
  // layout check with exactly representable small integers
  uint8_t hA[32 * 64], hB[64 * 32];
  float fA[32 * 64], fB[64 * 32];
  const uint8_t codes[] = {0x00, 0x38, 0x40, 0x44, 0x48, 0xb8, 0xc0, 0x30};  // 0, 1, 2, 3, 4, -1, -2, 0.5
  uint32_t s = 12345;
  for (int i = 0; i < 32 * 64; ++i) { s = s * 1664525u + 1013904223u; hA[i] = codes[(s >> 24) & 7]; fA[i] = e4m3_decode(hA[i]); }
  for (int i = 0; i < 64 * 32; ++i) { s = s * 1664525u + 1013904223u; hB[i] = codes[(s >> 24) & 7]; fB[i] = e4m3_decode(hB[i]); }
  uint8_t *dA, *dB; float* dD;
  (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dD, 4096);
  (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int sa = cfg == 0 ? 127 : cfg == 1 ? 116 : cfg == 2 ? 127 : (116 | (127 << 8)), sb = cfg == 2 ? 120 : 127;
    k_mfma<<<1, 64>>>(dA, dB, dD, sa, sb);
    float hD[1024];
    (void)hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    const float expect_scale = ldexpf(1.0f, ((sa & 255) - 127) + ((sb & 255) - 127));
    double maxerr = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        float ref = 0;
        for (int k = 0; k < 64; ++k) ref += fA[i * 64 + k] * fB[k * 32 + j];
        maxerr = fmax(maxerr, fabs(hD[i * 32 + j] - ref * expect_scale));
      }
    printf("mfma_scale 32x32x64 fp8, scale_a %#x scale_b %#x: max |D - A.B * 2^(sa+sb-254)| = %g (D[0][0] = %g)\n", sa, sb, maxerr, hD[0]);
  }
  return 0;
}
