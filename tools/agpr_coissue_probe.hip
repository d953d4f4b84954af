// Probe (round 2): does it matter for VALU / MFMA overlap whether the MFMA accumulators live in ArchVGPRs or in AccVGPRs?
// Each wave runs REPS x 16 x (one v_mfma_f32_32x32x16_f16 on one of 4 accumulators + NV independent v_fma_f32), with the
// accumulators pinned to "v" or "a" registers through inline-asm constraints.  Compare with the MFMA-only and VALU-only loops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool AGPR, int NV, bool WITH_MFMA, int VOP = 0>
__global__ __launch_bounds__(512) void k_probe(float* out, int reps) {
  f32x16 acc0, acc1, acc2, acc3;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  float v[8];
  unsigned long long w[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.01f + i; w[i] = threadIdx.x * 0x100000001ull + i; }
  const float c0 = 1.0001f, c1 = 0.0001f;
#define ONE(ACC)                                                                                              \
  if (WITH_MFMA) {                                                                                            \
    if (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(a), "v"(b));            \
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));                 \
  }                                                                                                           \
  _Pragma("unroll") for (int q = 0; q < NV; ++q) {                                                            \
    if (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(c0), "v"(c1));               \
    if (VOP == 1) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[q & 7]) : "v"(c0));                  \
    if (VOP == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[q & 7]));                               \
    if (VOP == 3) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(v[q & 7]) : "v"(c0), "v"(c1));            \
    if (VOP == 4) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(v[q & 7]) : "v"(c0), "v"(c1)); \
    if (VOP == 5) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[q & 7]) : "v"(c0));                            \
  }
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { ONE(acc0) ONE(acc1) ONE(acc2) ONE(acc3) }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  for (int i = 0; i < 8; ++i) s += v[i] + (float)(w[i] & 0xffff);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool AGPR, int NV, bool WITH_MFMA, int VOP = 0>
float run(float* out, int waves_per_simd) {
  const int reps = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int threads = 64 * 4 * waves_per_simd;
  k_probe<AGPR, NV, WITH_MFMA, VOP><<<256, threads>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k_probe<AGPR, NV, WITH_MFMA, VOP><<<256, threads>>>(out, reps);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / reps;  // ns per loop body (16 MFMA + 16 NV VALU)
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  for (int w = 1; w <= 2; ++w) {
    printf("waves/SIMD %d  [ns per body of 16 MFMA + 16 x NV v_fma]\n", w);
    printf("  MFMA only            acc in VGPR %7.1f   AGPR %7.1f\n", run<false, 0, true>(out, w), run<true, 0, true>(out, w));
    printf("  VALU only  NV=4 %7.1f   NV=8 %7.1f\n", run<false, 4, false>(out, w), run<false, 8, false>(out, w));
    printf("  MFMA + 4 VALU each   acc in VGPR %7.1f   AGPR %7.1f\n", run<false, 4, true>(out, w), run<true, 4, true>(out, w));
    printf("  MFMA + 8 VALU each   acc in VGPR %7.1f   AGPR %7.1f\n", run<false, 8, true>(out, w), run<true, 8, true>(out, w));
    const char* names[6] = {"v_fma_f32", "v_cvt_pkrtz_f16_f32", "v_pk_fma_f32", "v_cvt_pk_fp8_f32", "v_fma_mix_f32", "v_max_i32"};
#define ROW(OP) printf("  %-22s x8 per MFMA: VALU only %7.1f  with MFMA %7.1f   | x14: VALU only %7.1f  with MFMA %7.1f\n", names[OP], \
        run<false, 8, false, OP>(out, w), run<false, 8, true, OP>(out, w), run<false, 14, false, OP>(out, w), run<false, 14, true, OP>(out, w));
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5)
  }
  return 0;
}
