// Probe: packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32, also with op_sel broadcast) on HIGH register numbers of a ~236-VGPR
// allocation, two waves per SIMD in lockstep: the second wave's operands then live in the upper half of the 512-entry file.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(512) void k_probe(unsigned* errors, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned err = 0;
  for (int it = 0; it < reps; ++it) {
    float a0 = (float)(lane + it % 13), a1 = (float)(lane * 2 + 1), b0 = (float)(lane % 5 + 1), b1 = (float)(lane % 3 + 2);
    float c0 = (float)((it + lane) % 11), c1 = 3.f, w0 = 0.5f + wave, w1 = 0.25f + 2 * wave + (it & 3);  // wave-specific weights
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(w0), "+v"(w1));
    float r0, r1;
    if (MODE == 0) {  // high registers
      asm volatile("v_mov_b32 v220, %2\n v_mov_b32 v221, %3\n v_mov_b32 v222, %4\n v_mov_b32 v223, %5\n v_mov_b32 v224, %6\n v_mov_b32 v225, %7\n"
                   "v_mov_b32 v226, %8\n v_mov_b32 v227, %9\n s_nop 1\n"
                   "v_pk_mul_f32 v[228:229], v[220:221], v[226:227] op_sel_hi:[1,0]\n"     // a * w0
                   "v_pk_fma_f32 v[228:229], v[222:223], v[226:227], v[228:229] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"  // + b * w1
                   "v_pk_mul_f32 v[230:231], v[228:229], v[224:225]\n"                    // * c
                   "s_nop 1\n v_mov_b32 %0, v230\n v_mov_b32 %1, v231"
                   : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1), "v"(w0), "v"(w1)
                   : "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231");
    } else if (MODE == 2) {  // the failing kernels' form: v_pk_fma_f32 D, W, X, D op_sel:[1,0,0] (both results use W's odd register)
      asm volatile("v_mov_b32 v220, %2\n v_mov_b32 v221, %3\n v_mov_b32 v222, %4\n v_mov_b32 v223, %5\n v_mov_b32 v224, %6\n v_mov_b32 v225, %7\n"
                   "v_mov_b32 v226, %8\n v_mov_b32 v227, %9\n s_nop 1\n"
                   "v_pk_mul_f32 v[228:229], v[220:221], v[226:227] op_sel_hi:[1,0]\n"
                   "v_pk_fma_f32 v[228:229], v[226:227], v[222:223], v[228:229] op_sel:[1,0,0]\n"
                   "v_pk_mul_f32 v[230:231], v[228:229], v[224:225]\n"
                   "s_nop 1\n v_mov_b32 %0, v230\n v_mov_b32 %1, v231"
                   : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1), "v"(w0), "v"(w1)
                   : "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231");
    } else {          // same on low registers, same allocation size
      asm volatile("v_mov_b32 v20, %2\n v_mov_b32 v21, %3\n v_mov_b32 v22, %4\n v_mov_b32 v23, %5\n v_mov_b32 v24, %6\n v_mov_b32 v25, %7\n"
                   "v_mov_b32 v26, %8\n v_mov_b32 v27, %9\n s_nop 1\n"
                   "v_pk_mul_f32 v[28:29], v[20:21], v[26:27] op_sel_hi:[1,0]\n"
                   "v_pk_fma_f32 v[28:29], v[22:23], v[26:27], v[28:29] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                   "v_pk_mul_f32 v[30:31], v[28:29], v[24:25]\n"
                   "s_nop 1\n v_mov_b32 %0, v30\n v_mov_b32 %1, v31\n v_mov_b32 v231, 0"
                   : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1), "v"(w0), "v"(w1)
                   : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v231");
    }
    const float e0 = (a0 * w0 + b0 * w1) * c0, e1 = (a1 * w0 + b1 * w1) * c1;
    if (r0 != e0 || r1 != e1) ++err;
  }
  if (err) { atomicAdd(&errors[lane >> 4], err); atomicAdd(&errors[4 + (wave >= 4)], 1u); }
}

template <int MODE>
void run(unsigned* d_err, const char* name) {
  (void)hipMemset(d_err, 0, 32);
  k_probe<MODE><<<256, 512>>>(d_err, 200000);
  (void)hipDeviceSynchronize();
  unsigned h[8];
  (void)hipMemcpy(h, d_err, 32, hipMemcpyDeviceToHost);
  printf("%-50s wrong results by 16-lane row: %u %u %u %u   lanes with errors in waves 0-3 / 4-7: %u / %u\n", name, h[0], h[1], h[2], h[3], h[4], h[5]);
}
int main() {
  unsigned* d_err;
  (void)hipMalloc(&d_err, 32);
  run<0>(d_err, "packed fp32 math on v220..v231");
  run<1>(d_err, "packed fp32 math on v20..v31 (232-VGPR allocation)");
  run<2>(d_err, "v_pk_fma_f32 D, W, X, D op_sel:[1,0,0] on v220..v231");
  return 0;
}
