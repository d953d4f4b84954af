// Probe for the next round's idea: how long does one v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3, K = 64) take next to
// one v_mfma_f32_32x32x16_f16 (K = 16) on MI355X, alone and with two waves per SIMD?  And a numeric sanity check of the operand
// layout: all-ones operands with unit scales must give K in every accumulator element.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void k_rate(float* out, int reps) {
  v16f acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  v8i a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838; b8[i] = 0x38383838; }  // e4m3 1.0 = 0x38
  h8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)1.0f; bh[i] = (_Float16)1.0f; }
  const int one = 127;  // E8M0 scale 2^0
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int k = 0; k < (KIND == 2 ? 24 : 16); ++k) {
      if (KIND == 0) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[k & 3], 0, 0, 0);
      else if (KIND == 1) acc[k & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[k & 3], 0, 0, 0, one, 0, one);
      else if (KIND == 2) {  // the f16f8 kernel's pattern: per pair of k-steps 4 + 4 fp16 MFMAs, then 4 fp8 ones, on 4 accumulators
        if ((k % 12) < 8) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[k & 3], 0, 0, 0);
        else acc[k & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[k & 3], 0, 0, 0, one, 0, one);
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
float run(float* out, int waves_per_simd) {
  const int reps = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k_rate<KIND><<<256, 256 * waves_per_simd>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k_rate<KIND><<<256, 256 * waves_per_simd>>>(out, reps);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / (reps * (KIND == 2 ? 24.0f : 16.0f));  // ns per instruction per wave
}

__global__ void k_check(float* out) {
  v16f acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  v8i a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838; b8[i] = 0x38383838; }
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc, 0, 0, 0, 127, 0, 127);
  for (int r = 0; r < 16; ++r) out[threadIdx.x * 16 + r] = acc[r];
}

int main() {
  float* out;
  (void)hipMalloc(&out, 512 * 256 * 4);
  for (int w = 1; w <= 2; ++w) {
    const float f16 = run<0>(out, w), f8 = run<1>(out, w), mix = run<2>(out, w);
    printf("waves/SIMD %d: mixed pattern (8 fp16 + 4 fp8 per 12, 4 accumulators): %.2f ns per instruction; sum of parts %.2f ns\n", w, mix,
           (8 * f16 + 4 * f8) / 12);
    printf("waves/SIMD %d: v_mfma_f32_32x32x16_f16 %.2f ns, v_mfma_scale_f32_32x32x64_f8f6f4 (fp8) %.2f ns per instruction per wave -> fp8 / f16 = %.2f for 4x the K\n",
           w, f16, f8, f8 / f16);
  }
  k_check<<<1, 64>>>(out);
  float h[1024];
  (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  float mn = 1e30f, mx = -1e30f;
  for (int i = 0; i < 1024; ++i) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; }
  printf("all-ones fp8 operands, unit scales: accumulator min %.1f max %.1f (expected 64)\n", mn, mx);
  return 0;
}
