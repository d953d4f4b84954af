// Probe: two waves per SIMD; one wave class runs checked packed-fp32 math (v_pk_mul_f32 / v_pk_fma_f32 with op_sel, as in the
// gather's interpolation), the other a stream of cross-lane operations (DPP quad_perm / v_permlane16_swap / ds_swizzle / ds_bpermute),
// each checking its own results.  ROLE_SWAP exchanges which waves (0-3 or 4-7) do what.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int XL, bool SWAP_ROLES, bool BOTH>
__global__ __launch_bounds__(512) void k_probe(unsigned* errors, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool math_wave = BOTH ? true : ((wave >= 4) == SWAP_ROLES);
  const bool xl_wave = BOTH ? true : !math_wave;
  unsigned err_math = 0, err_xl = 0;
  for (int it = 0; it < reps; ++it) {
    if (math_wave) {
      // 12 packed products like team_finish: pv = t00*w00 + t01*w01 + t10*w10 + t11*w11; lv = u0*a + u1*b; m = pv*lv
      f2 t00 = {(float)(lane + it), (float)(lane * 2 + 1)}, t01 = {(float)(lane * 3 + 2), (float)(it % 7)};
      f2 t10 = {(float)(lane % 5 + 1), (float)(lane % 3 + 2)}, t11 = {(float)((it + lane) % 11), 3.f};
      f2 u0 = {2.f, (float)(lane % 4)}, u1 = {(float)(it % 3), 1.f};
      f2 w = {0.5f, 0.25f}, w2 = {2.f, 4.f}, ab = {1.f, 2.f};
      asm volatile("" : "+v"(t00), "+v"(t01), "+v"(t10), "+v"(t11), "+v"(u0), "+v"(u1), "+v"(w), "+v"(w2), "+v"(ab));
      f2 pv, lv, m;
      asm volatile("v_pk_mul_f32 %0, %3, %7 op_sel_hi:[1,0]\n v_pk_fma_f32 %0, %4, %7, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                   "v_pk_fma_f32 %0, %5, %8, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %0, %6, %8, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                   "v_pk_mul_f32 %1, %9, %11 op_sel_hi:[1,0]\n v_pk_fma_f32 %1, %10, %11, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                   "v_pk_mul_f32 %2, %0, %1"
                   : "=&v"(pv), "=&v"(lv), "=&v"(m)
                   : "v"(t00), "v"(t01), "v"(t10), "v"(t11), "v"(w), "v"(w2), "v"(u0), "v"(u1), "v"(ab));
      const float e0 = (t00.x * w.x + t01.x * w.y + t10.x * w2.x + t11.x * w2.y) * (u0.x * ab.x + u1.x * ab.y);
      const float e1 = (t00.y * w.x + t01.y * w.y + t10.y * w2.x + t11.y * w2.y) * (u0.y * ab.x + u1.y * ab.y);
      if (m.x != e0 || m.y != e1) ++err_math;
    }
    if (xl_wave) {
      float x = (float)(lane * 3 + it + 1000 * wave), y;
      float ey = (float)((lane ^ 1) * 3 + it + 1000 * wave);
      asm volatile("" : "+v"(x));
      if (XL == 0) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1" : "=&v"(y) : "v"(x));
      else if (XL == 1) {
        float x2 = x + 0.5f;
        asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1\n s_nop 1" : "+v"(x), "+v"(x2));
        const int row = lane >> 4;
        y = x; ey = (row & 1) ? (float)((lane - 16) * 3 + it + 1000 * wave) + 0.5f : (float)(lane * 3 + it + 1000 * wave);
      } else if (XL == 2) y = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x041F));
      else y = __shfl_xor(x, 1, 64);
      if (y != ey) ++err_xl;
    }
  }
  if (err_math) { atomicAdd(&errors[lane >> 4], err_math); atomicAdd(&errors[8 + (wave >= 4)], 1u); }
  if (err_xl) { atomicAdd(&errors[4 + (lane >> 4)], err_xl); atomicAdd(&errors[10 + (wave >= 4)], 1u); }
}

template <int XL, bool SWAP_ROLES, bool BOTH>
void run(unsigned* d_err, const char* name) {
  (void)hipMemset(d_err, 0, 12 * sizeof(unsigned));
  k_probe<XL, SWAP_ROLES, BOTH><<<256, 512>>>(d_err, 100000);
  (void)hipDeviceSynchronize();
  unsigned h[12];
  (void)hipMemcpy(h, d_err, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-64s math errors by row %u %u %u %u (waves 0-3/4-7: %u/%u)  cross-lane errors by row %u %u %u %u (%u/%u)\n", name, h[0], h[1], h[2], h[3],
         h[8], h[9], h[4], h[5], h[6], h[7], h[10], h[11]);
}

int main() {
  unsigned* d_err;
  (void)hipMalloc(&d_err, 12 * sizeof(unsigned));
  run<0, false, false>(d_err, "pk math in waves 0-3, DPP in waves 4-7");
  run<0, true, false>(d_err, "DPP in waves 0-3, pk math in waves 4-7");
  run<0, false, true>(d_err, "all waves: pk math + DPP");
  run<1, false, false>(d_err, "pk math in waves 0-3, permlane16_swap in waves 4-7");
  run<1, true, false>(d_err, "permlane16_swap in waves 0-3, pk math in waves 4-7");
  run<1, false, true>(d_err, "all waves: pk math + permlane16_swap");
  run<2, false, true>(d_err, "all waves: pk math + ds_swizzle");
  run<3, false, true>(d_err, "all waves: pk math + ds_bpermute");
  return 0;
}
