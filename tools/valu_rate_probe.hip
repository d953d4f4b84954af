// Probe: issue cost of the VALU instructions the shade kernel is made of, on one SIMD of MI355X (gfx950).
// One wave per SIMD (256 threads per workgroup, one workgroup per CU) runs REPS x 64 instructions of one opcode over 8
// independent register chains; the time per instruction is reported relative to v_fma_f32.  Build and run:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate_probe tools/valu_rate_probe.hip && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

enum Op {
  FMA, PK_FMA, PK_MUL, PK_ADD, MUL_LO_U32, MUL_U32_U24, MAD_U32_U24, LSHL_ADD_U64, LSHL_ADD_U32, ADD_U32, CNDMASK, CVT_PKRTZ,
  FMA_MIX, MAX_I32, SIN, EXP, RCP, CVT_F32_F16, PERMLANE32_SWAP, PERMLANE16_SWAP, MOV, DOT2_F32_F16, MAD_U64_U32, ASHRREV, MED3,
  CNDMASK_E64, CNDMASK_IND, CMP_VCC, CMP_SGPR, ADD_CO, FLOOR, CVT_I32, AND_B32, MOV_DPP, FMA_SGPR, CNDMASK_CONST, V_MAX_F32, FMAC, CNDMASK_E64_VCC, PAIR_VCC, PAIR_SGPR, CVT_PK_FP8, CVT_SCALE_PK_FP8, CVT_PK_F16, CVT_PK_FP8_HI,
  N_OPS
};
static const char* NAMES[N_OPS] = {
  "v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_lshl_add_u64",
  "v_lshl_add_u32", "v_add_u32", "v_cndmask_b32", "v_cvt_pkrtz_f16_f32", "v_fma_mix_f32", "v_max_i32", "v_sin_f32", "v_exp_f32",
  "v_rcp_f32", "v_cvt_f32_f16", "v_permlane32_swap", "v_permlane16_swap", "v_mov_b32", "v_dot2c_f32_f16", "v_mad_u64_u32",
  "v_ashrrev_i32", "v_med3_f32", "v_cndmask_b32 e64 s[10:11]", "v_cndmask_b32 (dst!=src)", "v_cmp_lt_f32 vcc",
  "v_cmp_lt_f32 s[10:11]", "v_add_co_u32 vcc", "v_floor_f32", "v_cvt_i32_f32", "v_and_b32", "v_mov_b32 dpp row_shr:1", "v_fma_f32 with SGPR src",
  "v_cndmask_b32 0, 1.0, vcc", "v_max_f32", "v_fmac_f32", "v_cndmask_b32_e64 v,v,vcc", "cmp+cndmask via vcc (2)", "cmp+cndmask via s[10:11] (2)", "v_cvt_pk_fp8_f32", "v_cvt_scalef32_pk_fp8_f32",
  "v_cvt_pk_f16_f32", "v_cvt_pk_fp8_f32 op_sel hi"};

template <int OP>
__device__ __forceinline__ void one(float& a, float& b, uint64_t& w, float c0, float c1) {
  if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c0), "v"(c1));
  if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w));
  if (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(w));
  if (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(w));
  if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(c0));
  if (OP == MUL_U32_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(c0));
  if (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(c0), "v"(c1));
  if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(w));
  if (OP == LSHL_ADD_U32) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a) : "v"(c0));
  if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(c0));
  if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(c0));
  if (OP == CVT_PKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a) : "v"(c0));
  if (OP == FMA_MIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a) : "v"(c0), "v"(c1));
  if (OP == MAX_I32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(c0));
  if (OP == SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(a));
  if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
  if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
  if (OP == CVT_F32_F16) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a));
  if (OP == PERMLANE32_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  if (OP == PERMLANE16_SWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(c0));
  if (OP == DOT2_F32_F16) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a) : "v"(c0), "v"(c1));
  if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w) : "v"(c0), "v"(c1) : "vcc");
  if (OP == ASHRREV) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a));
  if (OP == CNDMASK_E64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a) : "v"(c0));
  if (OP == CNDMASK_IND) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(c0), "v"(c1));
  if (OP == CMP_VCC) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a), "v"(c0) : "vcc");
  if (OP == CMP_SGPR) asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1" : : "v"(a), "v"(c0) : "s10", "s11");
  if (OP == ADD_CO) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a) : "v"(c0) : "vcc");
  if (OP == FLOOR) asm volatile("v_floor_f32 %0, %0" : "+v"(a));
  if (OP == CVT_I32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a));
  if (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(c0));
  if (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));
  if (OP == FMA_SGPR) asm volatile("v_fma_f32 %0, %0, s10, %1" : "+v"(a) : "v"(c1));
  if (OP == CNDMASK_CONST) asm volatile("v_cndmask_b32_e64 %0, 0, 1.0, vcc" : "=v"(a));
  if (OP == V_MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(c0));
  if (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(c0), "v"(c1));
  if (OP == CNDMASK_E64_VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a) : "v"(c0));
  if (OP == PAIR_VCC) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(c0), "v"(c1) : "vcc");
  if (OP == PAIR_SGPR) asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %2, s[10:11]" : "+v"(a) : "v"(c0), "v"(c1) : "s10", "s11");
  if (OP == CVT_PK_FP8) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(a) : "v"(c0), "v"(c1));
  if (OP == CVT_SCALE_PK_FP8) asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(a) : "v"(c0), "v"(c1), "v"(b));
  if (OP == CVT_PK_F16) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a) : "v"(c0));
  if (OP == CVT_PK_FP8_HI) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(a) : "v"(c0), "v"(c1));
  if (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c0), "v"(c1));
}

template <int OP>
__global__ __launch_bounds__(512) void k_probe(float* out, int reps) {
  float a[8], b[8];
  uint64_t w[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = threadIdx.x * 0.01f + i;
    b[i] = threadIdx.x * 0.02f - i;
    w[i] = (uint64_t)threadIdx.x * 0x100000001ull + i;
  }
  const float c0 = 1.0001f, c1 = 0.0001f;
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int k = 0; k < 64; ++k) one<OP>(a[k & 7], b[k & 7], w[k & 7], c0, c1);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i] + b[i] + (float)(w[i] & 0xffff);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
float run(float* out, int waves_per_simd) {
  const int reps = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int threads = 64 * 4 * waves_per_simd;
  k_probe<OP><<<256, threads>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k_probe<OP><<<256, threads>>>(out, reps);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / (reps * 64.0f);  // ns per instruction per wave
}

template <int OP>
void all(float* out, float* res, int w) {
  res[OP] = run<OP>(out, w);
  if constexpr (OP + 1 < N_OPS) all<OP + 1>(out, res, w);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 512 * 256 * 4);
  float res[N_OPS], res2[N_OPS];
  all<0>(out, res, 1);
  all<0>(out, res, 1);  // second pass: clocks settled
  all<0>(out, res2, 2);
  printf("%-32s %10s %8s %14s %8s\n", "instruction", "ns/instr", "x v_fma", "2 waves/SIMD", "x v_fma");
  for (int i = 0; i < N_OPS; ++i)
    printf("%-32s %10.3f %8.2f %14.3f %8.2f\n", NAMES[i], res[i], res[i] / res[FMA], res2[i], res2[i] / res2[FMA]);
  return 0;
}
