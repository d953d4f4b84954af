// Probe: v_cmp -> SGPR lane mask -> v_cndmask chains (the shape of the bilinear tap set-up) with two waves per SIMD in lockstep.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

struct Lin { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lin lin_setup(float xhat, int n) {
  Lin t;
  const float ix = __fmul_rn(__fadd_rn(xhat, 1.0f), 0.5f * (float)(n - 1));
  const float fl = floorf(ix);
  const float f = __fsub_rn(ix, fl);
  const float flc = fminf(fmaxf(fl, -2.0f), (float)n);
  const int i0 = (int)flc, i1 = i0 + 1;
  t.w0 = ((unsigned)i0 < (unsigned)n) ? __fsub_rn(1.0f, f) : 0.0f;
  t.w1 = ((unsigned)i1 < (unsigned)n) ? f : 0.0f;
  t.i0 = min(max(i0, 0), n - 1);
  t.i1 = min(max(i1, 0), n - 1);
  return t;
}

template <bool BIGREG>
__global__ __launch_bounds__(512) void k_probe(const float* xs, float* out, unsigned* errors, int reps, int n0, int n1, int n2) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned err = 0;
  const float* x = xs + (size_t)(blockIdx.x * 8 + wave) * 64 * 6 + lane * 6;
  float ref[12];
  for (int it = 0; it < reps; ++it) {
    float a[6];
    for (int k = 0; k < 6; ++k) a[k] = x[k];
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));
    const Lin A0 = lin_setup(a[0], n0), A1 = lin_setup(a[1], n1), A2 = lin_setup(a[2], n2);
    const Lin B0 = lin_setup(a[3], n0), B1 = lin_setup(a[4], n1), B2 = lin_setup(a[5], n2);
    if (BIGREG) asm volatile("v_mov_b32 v231, 0" ::: "v231");
    const float r[12] = {A0.w0 * (A0.i0 + 1), A0.w1 * (A0.i1 + 1), A1.w0 * (A1.i0 + 1), A1.w1 * (A1.i1 + 1), A2.w0 * (A2.i0 + 1), A2.w1 * (A2.i1 + 1),
                         B0.w0 * (B0.i0 + 1), B0.w1 * (B0.i1 + 1), B1.w0 * (B1.i0 + 1), B1.w1 * (B1.i1 + 1), B2.w0 * (B2.i0 + 1), B2.w1 * (B2.i1 + 1)};
    if (it == 0) { for (int k = 0; k < 12; ++k) ref[k] = r[k]; }
    else { for (int k = 0; k < 12; ++k) if (r[k] != ref[k]) { ++err; break; } }
  }
  for (int k = 0; k < 12; ++k) out[((size_t)(blockIdx.x * 8 + wave) * 64 + lane) * 12 + k] = ref[k];
  if (err) { atomicAdd(&errors[lane >> 4], err); atomicAdd(&errors[4 + (wave >= 4)], 1u); }
}

int main() {
  const int nw = 256 * 8;
  float* h = new float[nw * 64 * 6];
  unsigned s = 12345;
  for (int i = 0; i < nw * 64 * 6; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) / 16777216.0f) * 2.6f - 1.3f; }
  float *xs, *out; unsigned* d_err;
  (void)hipMalloc(&xs, nw * 64 * 6 * 4); (void)hipMalloc(&out, nw * 64 * 12 * 4); (void)hipMalloc(&d_err, 32);
  (void)hipMemcpy(xs, h, nw * 64 * 6 * 4, hipMemcpyHostToDevice);
  for (int big = 0; big < 2; ++big) {
    (void)hipMemset(d_err, 0, 32);
    if (big) k_probe<true><<<256, 512>>>(xs, out, d_err, 20000, 10, 10, 30);
    else k_probe<false><<<256, 512>>>(xs, out, d_err, 20000, 10, 10, 30);
    (void)hipDeviceSynchronize();
    unsigned e[8];
    (void)hipMemcpy(e, d_err, 32, hipMemcpyDeviceToHost);
    printf("tap set-up x6 per iteration, %s: iterations differing from the first by row: %u %u %u %u   lanes in waves 0-3 / 4-7: %u / %u\n",
           big ? "232 VGPRs" : "small allocation", e[0], e[1], e[2], e[3], e[4], e[5]);
  }
  return 0;
}
