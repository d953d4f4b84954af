// Probe: do VALU instructions overlap with MFMAs on one SIMD of MI355X?  Each wave runs REP x (NM independent
// v_mfma_f32_32x32x16_f16 + NV independent v_fma_f32); time for (NM, NV) vs (NM, 0) and (0, NV).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NV, bool FINE>
__global__ __launch_bounds__(512) void k_probe(float* out, int reps) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  const float c0 = 1.0001f, c1 = 0.0001f;
  for (int it = 0; it < reps; ++it) {
    if (FINE) {  // one MFMA, then NV / NM VALU, repeated
#pragma unroll
      for (int k = 0; k < NM; ++k) {
        acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV / (NM ? NM : 1); ++q) v[q & 7] = __builtin_fmaf(v[q & 7], c0, c1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
        for (int k = 0; k < NM / 4; ++k) acc[(k) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[(k) & 3], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NV / 4; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], c0, c1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, bool FINE = false>
float run(float* out, int waves_per_simd) {
  const int reps = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 64 * 4 * waves_per_simd;  // one workgroup per CU
  k_probe<NM, NV, FINE><<<256, threads>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_probe<NM, NV, FINE><<<256, threads>>>(out, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / reps;  // ns per iteration
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  for (int w = 1; w <= 2; ++w) {
    printf("waves/SIMD %d   [ns per iteration of one wave-loop body; clk = ns * 2.1 approx]\n", w);
    printf("  16 MFMA          : %8.1f\n", run<16, 0>(out, w));
    printf("  128 VALU         : %8.1f\n", run<0, 128>(out, w));
    printf("  16 MFMA + 128 V  : %8.1f\n", run<16, 128>(out, w));
    printf("  16 MFMA + 64 V   : %8.1f\n", run<16, 64>(out, w));
    printf("  16 MFMA + 32 V   : %8.1f\n", run<16, 32>(out, w));
    printf("  fine 1 MFMA : 8 V (x16): %8.1f\n", run<16, 128, true>(out, w));
    printf("  fine 1 MFMA : 4 V (x16): %8.1f\n", run<16, 64, true>(out, w));
    printf("  fine 1 MFMA : 2 V (x16): %8.1f\n", run<16, 32, true>(out, w));
  }
  return 0;
}
