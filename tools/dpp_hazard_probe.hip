// Probe: is a cross-lane VALU operation (v_mov_b32_dpp quad_perm, v_permlane16_swap, v_permlane32_swap) reliable on gfx950 when
// its source was produced shortly before by another VALU instruction, with a second wave competing for the SIMD?
// Every wave runs REPS x { x = producer(x-ish); gap; y = crosslane(x); check y against the value computed from lane ids }.
//   hipcc --offload-arch=gfx950 -O3 -o tools/dpp_hazard_probe.bin tools/dpp_hazard_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

enum Consumer { DPP_QUAD = 0, PERM16 = 1, PERM32 = 2 };
enum Producer { P_CNDMASK = 0, P_PKMUL = 1, P_ADD = 2 };

template <int CONS, int PROD, int GAP, bool NOPGAP, bool BUSY>
__global__ __launch_bounds__(512) void k_probe(unsigned* errors, int reps) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  unsigned err = 0;
  // the "other" waves of a BUSY launch just hammer the VALU / matrix pipe
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  if (BUSY && (wave & 1)) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(i * 0.5f); }
    float v = lane;
    for (int it = 0; it < reps; ++it) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      for (int k = 0; k < 8; ++k) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    }
    float s = v;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) errors[0] += 1;
    return;
  }
  const unsigned long long even_mask = 0x5555555555555555ull;
  for (int it = 0; it < reps; ++it) {
    // inputs depend on the iteration so nothing is loop invariant
    float a = (float)(lane * 3 + it), b = (float)(lane * 5 - it);
    float c0 = a, c1 = b;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c0), "+v"(c1));
    float x, x2;
    float expect_x, expect_x2;
    if (PROD == P_CNDMASK) {
      asm volatile("v_cndmask_b32_e64 %0, %2, %3, %4\n v_cndmask_b32_e64 %1, %3, %2, %4" : "=&v"(x), "=&v"(x2) : "v"(a), "v"(b), "s"(even_mask));
      // lanes with mask bit set take src1
      expect_x = (lane & 1) ? a : b; expect_x2 = (lane & 1) ? b : a;
    } else if (PROD == P_PKMUL) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 in = {a, b}, w = {2.0f, 3.0f}, out;
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(out) : "v"(in), "v"(w));
      x = out.x; x2 = out.y;
      expect_x = a * 2.0f; expect_x2 = b * 3.0f;
    } else {
      asm volatile("v_add_f32 %0, %2, %3\n v_sub_f32 %1, %2, %3" : "=&v"(x), "=&v"(x2) : "v"(a), "v"(b));
      expect_x = a + b; expect_x2 = a - b;
    }
    // gap between producer and consumer
    if (NOPGAP) {
      if (GAP == 1) asm volatile("s_nop 0" : "+v"(x), "+v"(x2));
      if (GAP == 2) asm volatile("s_nop 1" : "+v"(x), "+v"(x2));
      if (GAP == 4) asm volatile("s_nop 3" : "+v"(x), "+v"(x2));
      if (GAP == 8) asm volatile("s_nop 7" : "+v"(x), "+v"(x2));
    } else {
#pragma unroll
      for (int k = 0; k < GAP; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c0) : "v"(lane), "v"(x), "v"(x2));
    }
    float y, y2;
    float ey, ey2;
    // what every lane would have produced, recomputed for the source lane
    auto val = [&](int l, int which) -> float {
      const float aa = (float)(l * 3 + it), bb = (float)(l * 5 - it);
      if (PROD == P_CNDMASK) return which == 0 ? ((l & 1) ? aa : bb) : ((l & 1) ? bb : aa);
      if (PROD == P_PKMUL) return which == 0 ? aa * 2.0f : bb * 3.0f;
      return which == 0 ? aa + bb : aa - bb;
    };
    if (CONS == DPP_QUAD) {
      asm volatile("s_nop 1\n v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1"
                   : "=&v"(y), "=&v"(y2) : "v"(x), "v"(x2));
      ey = val(lane ^ 1, 0); ey2 = val(lane ^ 1, 1);
    } else if (CONS == PERM16) {
      y = x; y2 = x2;
      asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1\n s_nop 1" : "+v"(y), "+v"(y2));
      const int row = lane >> 4;
      // odd rows of vdst <-> even rows of src0
      ey = (row & 1) ? val(lane - 16, 1) : val(lane, 0);
      ey2 = (row & 1) ? val(lane, 1) : val(lane + 16, 0);
    } else {
      y = x; y2 = x2;
      asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1\n s_nop 1" : "+v"(y), "+v"(y2));
      ey = lane < 32 ? val(lane, 0) : val(lane - 32, 1);
      ey2 = lane < 32 ? val(lane + 32, 0) : val(lane, 1);
    }
    (void)expect_x; (void)expect_x2;
    if (y != ey) ++err;
    if (y2 != ey2) ++err;
    if (c0 == -1.f) ++err;
  }
  if (err) atomicAdd(&errors[1 + (lane >> 4)], err);
}

template <int CONS, int PROD, int GAP, bool NOPGAP, bool BUSY>
void run(unsigned* d_err, const char* name) {
  (void)hipMemset(d_err, 0, 8 * sizeof(unsigned));
  k_probe<CONS, PROD, GAP, NOPGAP, BUSY><<<256, 512>>>(d_err, 20000);
  (void)hipDeviceSynchronize();
  unsigned h[8];
  (void)hipMemcpy(h, d_err, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s errors by row: %u %u %u %u\n", name, h[1], h[2], h[3], h[4]);
}


// ---- same wave: one MFMA in flight, DELAY idle cycles, then the cross-lane operation on unrelated registers -------------------
template <int CONS, int DELAY, int NMFMA>
__global__ __launch_bounds__(512) void k_mfma_then(unsigned* errors, int reps) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const int lane = threadIdx.x & 63;
  unsigned err = 0;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(i * 0.5f); }
  for (int it = 0; it < reps; ++it) {
    float x = (float)(lane * 3 + it), x2 = (float)(lane * 5 - it);
    asm volatile("" : "+v"(x), "+v"(x2));
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7" : "+v"(x), "+v"(x2));  // producers long done
#pragma unroll
    for (int k = 0; k < NMFMA; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    asm volatile("" : "+v"(acc), "+v"(x), "+v"(x2));
#pragma unroll
    for (int k = 0; k < DELAY; ++k) asm volatile("s_nop 3" : "+v"(x), "+v"(x2));
    float y, y2, ey, ey2;
    auto val = [&](int l, int which) -> float { return which == 0 ? (float)(l * 3 + it) : (float)(l * 5 - it); };
    if (CONS == DPP_QUAD) {
      asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1"
                   : "=&v"(y), "=&v"(y2) : "v"(x), "v"(x2));
      ey = val(lane ^ 1, 0); ey2 = val(lane ^ 1, 1);
    } else if (CONS == PERM16) {
      y = x; y2 = x2;
      asm volatile("v_permlane16_swap_b32 %0, %1\n s_nop 1" : "+v"(y), "+v"(y2));
      const int row = lane >> 4;
      ey = (row & 1) ? val(lane - 16, 1) : val(lane, 0);
      ey2 = (row & 1) ? val(lane, 1) : val(lane + 16, 0);
    } else {
      y = x; y2 = x2;
      asm volatile("v_permlane32_swap_b32 %0, %1\n s_nop 1" : "+v"(y), "+v"(y2));
      ey = lane < 32 ? val(lane, 0) : val(lane - 32, 1);
      ey2 = lane < 32 ? val(lane + 32, 0) : val(lane, 1);
    }
    if (y != ey) ++err;
    if (y2 != ey2) ++err;
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 12345.678f) ++err;
  if (err) atomicAdd(&errors[1 + (lane >> 4)], err);
}

template <int CONS, int DELAY, int NMFMA>
void run2(unsigned* d_err, const char* name) {
  (void)hipMemset(d_err, 0, 8 * sizeof(unsigned));
  k_mfma_then<CONS, DELAY, NMFMA><<<256, 512>>>(d_err, 20000);
  (void)hipDeviceSynchronize();
  unsigned h[8];
  (void)hipMemcpy(h, d_err, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s errors by row: %u %u %u %u\n", name, h[1], h[2], h[3], h[4]);
}
#define RUN2(C, D, N) run2<C, D, N>(d_err, #C " " #D " x4 cycles after " #N " MFMA (same wave)")
template <int CONS, int D>
void sweep(unsigned* d_err) {
  RUN2(CONS, D, 1); RUN2(CONS, D, 3);
  if constexpr (D < 24) sweep<CONS, D + 2>(d_err);
}

// ---- write-after-read: overwrite the source of a cross-lane op / the B operand of a queued (dependent) MFMA right after issue ----
template <int MODE, int GAPI>
__global__ __launch_bounds__(512) void k_war(unsigned* errors, int reps) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const int lane = threadIdx.x & 63;
  unsigned err = 0;
  for (int it = 0; it < reps; ++it) {
    if (MODE == 0) {  // DPP source overwritten by the next VALU instruction
      float x = (float)(lane * 3 + it), junk = -7777.f, y;
      asm volatile("" : "+v"(x), "+v"(junk));
      asm volatile("s_nop 3\n v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32 %1, %2\n s_nop 1"
                   : "=&v"(y), "+v"(x) : "v"(junk));
      if (y != (float)((lane ^ 1) * 3 + it)) ++err;
    } else {  // dependent MFMA chain, B of the last one overwritten GAPI instructions after it was issued
      h8 a, b, junk;
      for (int i = 0; i < 8; ++i) { a[i] = (_Float16)((lane + i + it) % 5); b[i] = (_Float16)((lane * 3 + i + it) % 7); junk[i] = (_Float16)99.f; }
      f32x16 acc, ref;
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; ref[r] = 0.f; }
      asm volatile("" : "+v"(a), "+v"(b), "+v"(junk), "+v"(acc), "+v"(ref));
      // reference: the same three products with nothing disturbed
      ref = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, ref, 0, 0, 0);
      ref = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, ref, 0, 0, 0);
      ref = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, ref, 0, 0, 0);
      asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7" : "+v"(ref));
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const u4 bw = __builtin_bit_cast(u4, b);
      const unsigned jw = 0x56305630u;  // two fp16 99.0
      if (GAPI == 0)
        asm volatile("v_mov_b32 v100, %2\n v_mov_b32 v101, %3\n v_mov_b32 v102, %4\n v_mov_b32 v103, %5\n s_nop 7\n"
                     "v_mfma_f32_32x32x16_f16 %0, %1, v[100:103], %0\n v_mfma_f32_32x32x16_f16 %0, %1, v[100:103], %0\n"
                     "v_mfma_f32_32x32x16_f16 %0, %1, v[100:103], %0\n"
                     "v_mov_b32 v100, %6\n v_mov_b32 v101, %6\n v_mov_b32 v102, %6\n v_mov_b32 v103, %6"
                     : "+v"(acc) : "v"(a), "v"(bw.x), "v"(bw.y), "v"(bw.z), "v"(bw.w), "v"(jw) : "v100", "v101", "v102", "v103");
      else
        asm volatile("v_mov_b32 v100, %2\n v_mov_b32 v101, %3\n v_mov_b32 v102, %4\n v_mov_b32 v103, %5\n s_nop 7\n"
                     "v_mfma_f32_32x32x16_f16 %0, %1, v[100:103], %0\n v_mfma_f32_32x32x16_f16 %0, %1, v[100:103], %0\n"
                     "v_mfma_f32_32x32x16_f16 %0, %1, v[100:103], %0\n s_nop 1\n"
                     "v_mov_b32 v100, %6\n v_mov_b32 v101, %6\n v_mov_b32 v102, %6\n v_mov_b32 v103, %6"
                     : "+v"(acc) : "v"(a), "v"(bw.x), "v"(bw.y), "v"(bw.z), "v"(bw.w), "v"(jw) : "v100", "v101", "v102", "v103");
      asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7" : "+v"(acc));
      for (int r = 0; r < 16; ++r) if (acc[r] != ref[r]) { ++err; break; }
    }
  }
  if (err) atomicAdd(&errors[1 + (lane >> 4)], err);
}
template <int MODE, int GAPI>
void run3(unsigned* d_err, const char* name) {
  (void)hipMemset(d_err, 0, 8 * sizeof(unsigned));
  k_war<MODE, GAPI><<<256, 512>>>(d_err, 5000);
  (void)hipDeviceSynchronize();
  unsigned h[8];
  (void)hipMemcpy(h, d_err, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s errors by row: %u %u %u %u\n", name, h[1], h[2], h[3], h[4]);
}

// ---- big-register kernel: > 128 VGPRs per wave, so the second wave of a SIMD sits in the upper half of the 512-entry file ------
template <int CONS>
__global__ __launch_bounds__(512) void k_bigreg(unsigned* errors, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned err = 0;
  for (int it = 0; it < reps; ++it) {
    float x = (float)(lane * 3 + it), x2 = (float)(lane * 5 - it), y, y2, ey, ey2;
    asm volatile("" : "+v"(x), "+v"(x2));
    auto val = [&](int l, int which) -> float { return which == 0 ? (float)(l * 3 + it) * 2.0f : (float)(l * 5 - it) * 2.0f; };
    if (CONS == DPP_QUAD) {
      asm volatile("v_mul_f32 v230, 2.0, %2\n v_mul_f32 v231, 2.0, %3\n s_nop 1\n"
                   "v_mov_b32_dpp v232, v230 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp v233, v231 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                   "v_mov_b32 %0, v232\n v_mov_b32 %1, v233"
                   : "=v"(y), "=v"(y2) : "v"(x), "v"(x2) : "v230", "v231", "v232", "v233");
      ey = val(lane ^ 1, 0); ey2 = val(lane ^ 1, 1);
    } else {
      asm volatile("v_mul_f32 v230, 2.0, %2\n v_mul_f32 v231, 2.0, %3\n s_nop 1\n"
                   "v_permlane16_swap_b32 v230, v231\n s_nop 1\n v_mov_b32 %0, v230\n v_mov_b32 %1, v231"
                   : "=v"(y), "=v"(y2) : "v"(x), "v"(x2) : "v230", "v231");
      const int row = lane >> 4;
      ey = (row & 1) ? val(lane - 16, 1) : val(lane, 0);
      ey2 = (row & 1) ? val(lane, 1) : val(lane + 16, 0);
    }
    if (y != ey) ++err;
    if (y2 != ey2) ++err;
  }
  if (err) { atomicAdd(&errors[1 + (lane >> 4)], err); atomicAdd(&errors[5 + (wave >= 4)], 1u); }
}
template <int CONS>
void run4(unsigned* d_err, const char* name) {
  (void)hipMemset(d_err, 0, 8 * sizeof(unsigned));
  k_bigreg<CONS><<<256, 512>>>(d_err, 20000);
  (void)hipDeviceSynchronize();
  unsigned h[8];
  (void)hipMemcpy(h, d_err, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s errors by row: %u %u %u %u   lanes with errors in waves 0-3 / 4-7: %u / %u\n", name, h[1], h[2], h[3], h[4], h[5], h[6]);
}

// ---- two waves per SIMD running the SAME code on the SAME VGPR numbers with DIFFERENT data; idle cycles between the producer and
// the cross-lane consumer give the other wave room to issue its own producer in between --------------------------------------------
template <int CONS, int NOPS, bool PRIO>
__global__ __launch_bounds__(512) void k_samecode(unsigned* errors, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
  unsigned err = 0;
  const float wbase = 1000.f * (float)(wave + 8 * blockIdx.x % 64);
  for (int it = 0; it < reps; ++it) {
    float x = (float)(lane * 3 + it) + wbase, x2 = (float)(lane * 5 - it) + wbase, y, y2, ey, ey2;
    asm volatile("" : "+v"(x), "+v"(x2));
    auto val = [&](int l, int which) -> float { return which == 0 ? ((float)(l * 3 + it) + wbase) * 2.0f : ((float)(l * 5 - it) + wbase) * 2.0f; };
#define NOPSTR(n) (n == 0 ? "" : "")
    if (CONS == DPP_QUAD) {
      if (NOPS == 0)
        asm volatile("v_mul_f32 v40, 2.0, %2\n v_mul_f32 v41, 2.0, %3\n s_nop 1\n"
                     "v_mov_b32_dpp v42, v40 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp v43, v41 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32 %0, v42\n v_mov_b32 %1, v43"
                     : "=v"(y), "=v"(y2) : "v"(x), "v"(x2) : "v40", "v41", "v42", "v43");
      else
        asm volatile("v_mul_f32 v40, 2.0, %2\n v_mul_f32 v41, 2.0, %3\n s_nop 7\n s_nop 7\n"
                     "v_mov_b32_dpp v42, v40 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp v43, v41 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32 %0, v42\n v_mov_b32 %1, v43"
                     : "=v"(y), "=v"(y2) : "v"(x), "v"(x2) : "v40", "v41", "v42", "v43");
      ey = val(lane ^ 1, 0); ey2 = val(lane ^ 1, 1);
    } else {
      if (NOPS == 0)
        asm volatile("v_mul_f32 v40, 2.0, %2\n v_mul_f32 v41, 2.0, %3\n s_nop 1\n v_permlane16_swap_b32 v40, v41\n s_nop 1\n v_mov_b32 %0, v40\n v_mov_b32 %1, v41"
                     : "=v"(y), "=v"(y2) : "v"(x), "v"(x2) : "v40", "v41");
      else
        asm volatile("v_mul_f32 v40, 2.0, %2\n v_mul_f32 v41, 2.0, %3\n s_nop 7\n s_nop 7\n v_permlane16_swap_b32 v40, v41\n s_nop 1\n v_mov_b32 %0, v40\n v_mov_b32 %1, v41"
                     : "=v"(y), "=v"(y2) : "v"(x), "v"(x2) : "v40", "v41");
      const int row = lane >> 4;
      ey = (row & 1) ? val(lane - 16, 1) : val(lane, 0);
      ey2 = (row & 1) ? val(lane, 1) : val(lane + 16, 0);
    }
    if (y != ey) ++err;
    if (y2 != ey2) ++err;
  }
  if (err) { atomicAdd(&errors[1 + (lane >> 4)], err); atomicAdd(&errors[5 + (wave >= 4)], 1u); }
}
template <int CONS, int NOPS, bool PRIO>
void run5(unsigned* d_err, const char* name) {
  (void)hipMemset(d_err, 0, 8 * sizeof(unsigned));
  k_samecode<CONS, NOPS, PRIO><<<256, 512>>>(d_err, 50000);
  (void)hipDeviceSynchronize();
  unsigned h[8];
  (void)hipMemcpy(h, d_err, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s errors by row: %u %u %u %u   lanes with errors in waves 0-3 / 4-7: %u / %u\n", name, h[1], h[2], h[3], h[4], h[5], h[6]);
}

template <int CONS>
__global__ __launch_bounds__(512) void k_straddle(unsigned* errors, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned err = 0;
  const float wbase = 1000.f * (float)wave;
  for (int it = 0; it < reps; ++it) {
    float x = (float)(lane * 3 + it) + wbase, x2 = (float)(lane * 5 - it) + wbase, y, y2, ey, ey2;
    asm volatile("" : "+v"(x), "+v"(x2));
    auto val = [&](int l, int which) -> float { return which == 0 ? ((float)(l * 3 + it) + wbase) * 2.0f : ((float)(l * 5 - it) + wbase) * 2.0f; };
    if (CONS == DPP_QUAD) {
      asm volatile("v_mul_f32 v2, 2.0, %2\n v_mul_f32 v3, 2.0, %3\n s_nop 1\n"
                   "v_mov_b32_dpp v232, v2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp v233, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                   "v_mov_b32 %0, v232\n v_mov_b32 %1, v233"
                   : "=v"(y), "=v"(y2) : "v"(x), "v"(x2) : "v2", "v3", "v232", "v233");
      ey = val(lane ^ 1, 0); ey2 = val(lane ^ 1, 1);
    } else {
      asm volatile("v_mul_f32 v2, 2.0, %2\n v_mul_f32 v231, 2.0, %3\n s_nop 1\n"
                   "v_permlane16_swap_b32 v2, v231\n s_nop 1\n v_mov_b32 %0, v2\n v_mov_b32 %1, v231"
                   : "=v"(y), "=v"(y2) : "v"(x), "v"(x2) : "v2", "v231");
      const int row = lane >> 4;
      ey = (row & 1) ? val(lane - 16, 1) : val(lane, 0);
      ey2 = (row & 1) ? val(lane, 1) : val(lane + 16, 0);
    }
    if (y != ey) ++err;
    if (y2 != ey2) ++err;
  }
  if (err) { atomicAdd(&errors[1 + (lane >> 4)], err); atomicAdd(&errors[5 + (wave >= 4)], 1u); }
}
template <int CONS>
void run6(unsigned* d_err, const char* name) {
  (void)hipMemset(d_err, 0, 8 * sizeof(unsigned));
  k_straddle<CONS><<<256, 512>>>(d_err, 50000);
  (void)hipDeviceSynchronize();
  unsigned h[8];
  (void)hipMemcpy(h, d_err, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s errors by row: %u %u %u %u   lanes with errors in waves 0-3 / 4-7: %u / %u\n", name, h[1], h[2], h[3], h[4], h[5], h[6]);
}

#define RUN(C, P, G, N, B) run<C, P, G, N, B>(d_err, #C " after " #P " gap " #G " nop=" #N " busy=" #B)

int main() {
  unsigned* d_err;
  (void)hipMalloc(&d_err, 8 * sizeof(unsigned));
  RUN(DPP_QUAD, P_CNDMASK, 0, false, false); RUN(DPP_QUAD, P_CNDMASK, 0, false, true);
  RUN(DPP_QUAD, P_CNDMASK, 2, false, true);  RUN(DPP_QUAD, P_CNDMASK, 4, true, true);
  RUN(DPP_QUAD, P_PKMUL, 0, false, true);    RUN(DPP_QUAD, P_PKMUL, 2, false, true);
  RUN(DPP_QUAD, P_ADD, 0, false, true);      RUN(DPP_QUAD, P_ADD, 8, true, true);
  RUN(PERM16, P_CNDMASK, 0, false, false);   RUN(PERM16, P_CNDMASK, 0, false, true);
  RUN(PERM16, P_PKMUL, 0, false, true);      RUN(PERM16, P_PKMUL, 2, false, true);
  RUN(PERM16, P_PKMUL, 8, true, true);       RUN(PERM16, P_ADD, 0, false, true);
  RUN(PERM32, P_PKMUL, 0, false, true);      RUN(PERM32, P_CNDMASK, 0, false, true);
  run5<DPP_QUAD, 0, false>(d_err, "same code, wave-specific data: DPP, no idle cycles");
  run5<DPP_QUAD, 1, false>(d_err, "same code, wave-specific data: DPP, 16 idle cycles");
  run5<DPP_QUAD, 1, true>(d_err, "same code, wave-specific data: DPP, 16 idle, s_setprio");
  run5<PERM16, 0, false>(d_err, "same code, wave-specific data: permlane16_swap, no idle");
  run5<PERM16, 1, false>(d_err, "same code, wave-specific data: permlane16_swap, 16 idle");
  run5<PERM16, 1, true>(d_err, "same code, wave-specific data: permlane16_swap, 16 idle, prio");
  run6<DPP_QUAD>(d_err, "DPP v2 -> v232 (operands either side of the wave base)");
  run6<PERM16>(d_err, "permlane16_swap v2 <-> v231");
  run4<DPP_QUAD>(d_err, "DPP quad_perm on v230.. (234 VGPRs per wave)");
  run4<PERM16>(d_err, "permlane16_swap on v230/v231 (234 VGPRs per wave)");
  run3<0, 0>(d_err, "WAR: DPP source overwritten by the next instruction");
  run3<1, 0>(d_err, "WAR: B (first VGPR) of a queued dependent MFMA overwritten at once");
  run3<1, 2>(d_err, "WAR: same, 2 wait states later");
  return 0;
}
