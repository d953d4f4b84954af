// Probe: float-atomic throughput on MI355X as a function of how a wave's 64 lanes spread over cache lines.
// mode k: each wave instruction touches 64/k lines with k consecutive dwords each (k = 1, 2, 4, 16, 64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_atomic(float* tab, size_t n_lines, int k, int iters, unsigned seed) {
  const int lane = threadIdx.x & 63;
  unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ ((threadIdx.x >> 6) * 40503u);
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    // group = lane / k picks the line, lane % k the dword; lines are random but shared by the k lanes of a group
    unsigned gs = s ^ ((lane / k) * 2246822519u);
    gs ^= gs >> 15; gs *= 2654435761u; gs ^= gs >> 13;
    const size_t line = gs % n_lines;
    const int dw = (k >= 16) ? (lane % 16) : (lane % k) * (16 / k);
    const size_t extra = (k == 64) ? (size_t)(lane / 16) : 0;  // k = 64: four consecutive lines
    unsafeAtomicAdd(tab + (line + extra) * 16 + dw, 1.0f);
  }
}
int main() {
  const size_t n_lines = (74u << 20) / 64;
  float* tab;
  hipMalloc(&tab, (n_lines + 4) * 64);
  hipMemset(tab, 0, (n_lines + 4) * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 256, blocks = 2048, threads = 256;
  for (int k : {1, 2, 4, 16, 64}) {
    k_atomic<<<blocks, threads>>>(tab, n_lines, k, 8, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_atomic<<<blocks, threads>>>(tab, n_lines, k, iters, 7u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * threads * iters;
    printf("k=%2d lanes/line: %.3f ms, %.1f G atomics/s, %.1f G line-requests/s\n", k, ms, n / ms * 1e-6, n / (k > 16 ? 16 : k) / ms * 1e-6);
  }
  return 0;
}
