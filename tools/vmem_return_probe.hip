// Probe: integrity of vector-memory returns with two waves per SIMD, ~232 VGPRs per wave and 18 dwordx4 loads in flight per wave
// (the gather's shape).  buf[i] = i; every wave loads 18 x 16 B per lane from pseudo-random 64-byte-aligned places, waits, and checks
// all 72 registers.  hipcc --offload-arch=gfx950 -O3 -o tools/vmem_return_probe.bin tools/vmem_return_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool BIGREG, bool TEAM>
__global__ __launch_bounds__(512) void k_probe(const float* buf, unsigned* errors, int reps, unsigned n_lines) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned wid = blockIdx.x * 8 + wave;
  unsigned err = 0;
  for (int it = 0; it < reps; ++it) {
    f32x4 raw[18];
    unsigned line[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      // TEAM: the four lanes {2t, 2t+1, 2t+32, 2t+33} read the four quads of one line (as the gather does); else one line per lane
      const unsigned who = TEAM ? ((lane >> 1) & 15) : lane;
      line[k] = (((who * 97u + (unsigned)k * 31u + (unsigned)it * 131u + wid * 17u) * 2654435761u) >> 7) % n_lines;
      const unsigned quad = TEAM ? ((lane >> 5) + 2 * (lane & 1)) : (k & 3);
      raw[k] = *(const f32x4*)(buf + (uint64_t)line[k] * 16 + quad * 4);
    }
    if (BIGREG) asm volatile("v_mov_b32 v231, 0" ::: "v231");  // forces a ~232-VGPR allocation (2 waves per SIMD, second one high)
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      const unsigned quad = TEAM ? ((lane >> 5) + 2 * (lane & 1)) : (k & 3);
      const float e = (float)(line[k] * 16 + quad * 4);
      if (raw[k].x != e || raw[k].y != e + 1 || raw[k].z != e + 2 || raw[k].w != e + 3) ++err;
    }
  }
  if (err) { atomicAdd(&errors[lane >> 4], err); atomicAdd(&errors[4 + (wave >= 4)], 1u); }
}

template <bool BIGREG, bool TEAM>
void run(const float* buf, unsigned* d_err, unsigned n_lines, const char* name) {
  (void)hipMemset(d_err, 0, 8 * sizeof(unsigned));
  k_probe<BIGREG, TEAM><<<256, 512>>>(buf, d_err, 3000, n_lines);
  (void)hipDeviceSynchronize();
  unsigned h[8];
  (void)hipMemcpy(h, d_err, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-52s wrong registers by 16-lane row: %u %u %u %u   lanes with errors in waves 0-3 / 4-7: %u / %u\n", name, h[0], h[1], h[2], h[3], h[4], h[5]);
}

int main() {
  const unsigned n = 1u << 24;  // 64 MB of floats, exact as integers
  float* h = new float[n];
  for (unsigned i = 0; i < n; ++i) h[i] = (float)i;
  float* buf; unsigned* d_err;
  (void)hipMalloc(&buf, (size_t)n * sizeof(float));
  (void)hipMalloc(&d_err, 8 * sizeof(unsigned));
  (void)hipMemcpy(buf, h, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
  run<false, false>(buf, d_err, n / 16, "18 loads in flight, small allocation, lane lines");
  run<true, false>(buf, d_err, n / 16, "18 loads in flight, 232 VGPRs, lane lines");
  run<true, true>(buf, d_err, n / 16, "18 loads in flight, 232 VGPRs, 4-lane teams");
  return 0;
}
