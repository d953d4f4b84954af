// Does a per-sample-contiguous dv layout make the sorted scatter's random gather cheaper?  2.1 M samples x 3 planes, each a 16-lane group
// reading 3 x 64 B: layout A = k_shade_bwd's blocked dv ([tile of 32][9 chunks][32 samples][16]: the three lines 2 KB apart), layout B =
// [sample][144] (192 contiguous bytes).  Random permutation of the samples (the worst case of a cell-sorted order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
template <int LAYOUT>
__global__ void k_gather(const float* __restrict__ dv, const unsigned* __restrict__ perm, long M, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, c16 = lane & 15, q = lane >> 4;
  const long g = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + q;   // one group = 16 consecutive sorted positions x plane blockIdx.y
  const int I = blockIdx.y;
  float acc = 0.f;
  for (int t = 0; t < 16; t += 4) {
    float v[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long p = g * 16 + t + u;
      const long m = p < M ? perm[p] : 0;
#pragma unroll
      for (int i = 0; i < 3; ++i)
        v[u][i] = LAYOUT == 0 ? dv[(m >> 5) * 4608 + (I * 3 + i) * 512 + (m & 31) * 16 + c16] : dv[m * 144 + I * 48 + i * 16 + c16];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u][0] + v[u][1] + v[u][2];
  }
  if (acc == 12345.f) out[0] = acc;
}
int main() {
  const long M = 8192L * 256;
  std::vector<unsigned> perm(M);
  for (long i = 0; i < M; ++i) perm[i] = (unsigned)i;
  std::mt19937 rng(1);
  float* dv; unsigned* dperm; float* out;
  hipMalloc(&dv, M * 144 * 4); hipMalloc(&dperm, M * 4); hipMalloc(&out, 4);
  hipMemset(dv, 0, M * 144 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {   // 0: random samples; 1: runs of 4 consecutive samples at random places; 2: sequential
    if (mode == 0) std::shuffle(perm.begin(), perm.end(), rng);
    if (mode == 1) { std::vector<unsigned> r(M / 4); for (long i = 0; i < M / 4; ++i) r[i] = (unsigned)i; std::shuffle(r.begin(), r.end(), rng);
                     for (long i = 0; i < M / 4; ++i) for (int k = 0; k < 4; ++k) perm[4 * i + k] = 4 * r[i] + k; }
    if (mode == 2) for (long i = 0; i < M; ++i) perm[i] = (unsigned)i;
    hipMemcpy(dperm, perm.data(), M * 4, hipMemcpyHostToDevice);
    for (int layout = 0; layout < 2; ++layout) {
      const dim3 grid((unsigned)((M / 16 + 15) / 16), 3);
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (layout == 0) k_gather<0><<<grid, 256>>>(dv, dperm, M, out); else k_gather<1><<<grid, 256>>>(dv, dperm, M, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
      }
      printf("order %d (%s) layout %c: %.3f ms  (%.2f TB/s of useful bytes)\n", mode, mode == 0 ? "random samples" : mode == 1 ? "random runs of 4" : "sequential",
             layout ? 'B' : 'A', best, M * 576.0 / best / 1e9);
    }
  }
  return 0;
}
