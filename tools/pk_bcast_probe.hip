// Probe: the gather's interpolation with the weights kept as float2 pairs, so that the packed products take the second element by
// a HIGH-half broadcast (v_pk_fma_f32 ... op_sel:[1,0,0]) - the instruction form every non-reproducible build of the shade kernels
// contained - fed by 18 dwordx4 loads per round, two waves per SIMD, ~230 VGPRs.  Each wave computes 12 products
// twice per iteration, once with paired weights and once with scalar weights, and compares.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool BIGREG>
__global__ __launch_bounds__(512) void k_probe(const float* __restrict__ buf, unsigned* errors, int reps, unsigned n_lines) {
#pragma clang fp contract(fast)
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned wid = blockIdx.x * 8 + wave;
  unsigned err = 0;
  for (int it = 0; it < reps; ++it) {
    f32x4 raw[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      const unsigned who = (lane >> 1) & 15;
      const unsigned line = (((who * 97u + (unsigned)k * 31u + (unsigned)it * 131u + wid * 17u) * 2654435761u) >> 7) % n_lines;
      raw[k] = *(const f32x4*)(buf + (uint64_t)line * 16 + ((lane >> 5) + 2 * (lane & 1)) * 4);
    }
    // weights: pairs built by vector arithmetic (what the SLP vectoriser makes of w00..w11)
    f32x2 yw = {0.25f + 0.001f * (lane & 7) + wave, 0.75f - 0.001f * (lane & 7)};
    f32x2 xw = {0.4f + 0.002f * (it & 15), 0.6f - 0.002f * (it & 15) + 0.5f * wave};
    f32x2 lw = {0.3f + wave, 0.7f};
    asm volatile("" : "+v"(yw), "+v"(xw), "+v"(lw));
    const f32x2 w0 = xw * yw.x, w1 = xw * yw.y;   // {w00, w01}, {w10, w11}
    if (BIGREG) asm volatile("v_mov_b32 v235, 0" ::: "v235");
    float ref[12], got[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const f32x4 pv = raw[i] * w0.x + raw[3 + i] * w0.y + raw[6 + i] * w1.x + raw[9 + i] * w1.y;
      const f32x4 lv = raw[12 + i] * lw.x + raw[15 + i] * lw.y;
      const f32x4 m = pv * lv;
      got[4 * i] = m.x; got[4 * i + 1] = m.y; got[4 * i + 2] = m.z; got[4 * i + 3] = m.w;
    }
    float s00 = w0.x, s01 = w0.y, s10 = w1.x, s11 = w1.y, l0 = lw.x, l1 = lw.y;
    asm volatile("" : "+v"(s00)); asm volatile("" : "+v"(s01)); asm volatile("" : "+v"(s10)); asm volatile("" : "+v"(s11));
    asm volatile("" : "+v"(l0)); asm volatile("" : "+v"(l1));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const f32x4 pv = raw[i] * s00 + raw[3 + i] * s01 + raw[6 + i] * s10 + raw[9 + i] * s11;
      const f32x4 lv = raw[12 + i] * l0 + raw[15 + i] * l1;
      const f32x4 m = pv * lv;
      ref[4 * i] = m.x; ref[4 * i + 1] = m.y; ref[4 * i + 2] = m.z; ref[4 * i + 3] = m.w;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) if (got[k] != ref[k]) ++err;
  }
  if (err) { atomicAdd(&errors[lane >> 4], err); atomicAdd(&errors[4 + (wave >= 4)], 1u); }
}

int main() {
  const unsigned n = 1u << 24;
  float* h = new float[n];
  unsigned s = 1;
  for (unsigned i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) / 16777216.0f) - 0.5f; }
  float* buf; unsigned* d_err;
  (void)hipMalloc(&buf, (size_t)n * 4); (void)hipMalloc(&d_err, 32);
  (void)hipMemcpy(buf, h, (size_t)n * 4, hipMemcpyHostToDevice);
  for (int big = 0; big < 2; ++big) {
    (void)hipMemset(d_err, 0, 32);
    if (big) k_probe<true><<<256, 512>>>(buf, d_err, 4000, n / 16); else k_probe<false><<<256, 512>>>(buf, d_err, 4000, n / 16);
    (void)hipDeviceSynchronize();
    unsigned e[8];
    (void)hipMemcpy(e, d_err, 32, hipMemcpyDeviceToHost);
    printf("paired-weight vs scalar-weight interpolation, %s: differing products by 16-lane row %u %u %u %u, lanes in waves 0-3 / 4-7: %u / %u\n",
           big ? "236-VGPR allocation" : "small allocation", e[0], e[1], e[2], e[3], e[4], e[5]);
  }
  return 0;
}
