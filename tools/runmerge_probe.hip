// micro-test of run_after / run_sum (egonerf_amd/csrc/ego_train.inc) against a CPU model
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
template <int WIDTH>
__device__ __forceinline__ int run_after(int key, bool active, int lane) {
  const int nkey = __shfl_down(key, 1, WIDTH);
  const bool nact = __shfl_down((int)active, 1, WIDTH) != 0;
  const bool same_next = active && nact && nkey == key && ((lane & (WIDTH - 1)) != WIDTH - 1);
  const unsigned long long m = __ballot(same_next);
  const unsigned long long shifted = ~(m >> lane);
  return __builtin_ctzll(shifted);
}
template <int WIDTH>
__device__ __forceinline__ float run_sum(float v, int after) {
#pragma unroll
  for (int d = 1; d < WIDTH; d <<= 1) {
    const float t = __shfl_down(v, d, WIDTH);
    if (after >= d) v += t;
  }
  return v;
}
template <int WIDTH>
__global__ void kern(const int* key, const float* val, int* after, float* sum, int* head) {
  const int lane = threadIdx.x;
  const int a = run_after<WIDTH>(key[lane], true, lane);
  after[lane] = a;
  sum[lane] = run_sum<WIDTH>(val[lane], a);
  head[lane] = ((lane & (WIDTH - 1)) == 0) || __shfl_up(a, 1, WIDTH) == 0;
}
int main() {
  int hk[64], ha[64], hh[64]; float hv[64], hs[64];
  int *dk, *da, *dh; float *dv, *ds;
  hipMalloc(&dk, 256); hipMalloc(&da, 256); hipMalloc(&dh, 256); hipMalloc(&dv, 256); hipMalloc(&ds, 256);
  srand(3);
  for (int trial = 0; trial < 4; ++trial) {
    int k = 0;
    for (int i = 0; i < 64; ++i) { if (rand() % 3 == 0) k++; hk[i] = k; hv[i] = (float)(rand() % 7 + 1); }
    hipMemcpy(dk, hk, 256, hipMemcpyHostToDevice); hipMemcpy(dv, hv, 256, hipMemcpyHostToDevice);
    const int W = (trial & 1) ? 32 : 64;
    if (W == 64) kern<64><<<1, 64>>>(dk, dv, da, ds, dh); else kern<32><<<1, 64>>>(dk, dv, da, ds, dh);
    hipMemcpy(ha, da, 256, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, 256, hipMemcpyDeviceToHost); hipMemcpy(hh, dh, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
      int a = 0; while (i + a + 1 < 64 && ((i + a + 1) / W == i / W) && hk[i + a + 1] == hk[i]) a++;
      float s = 0; for (int j = i; j <= i + a; ++j) s += hv[j];
      int head = (i % W == 0) || hk[i - 1] != hk[i];
      if (a != ha[i] || s != hs[i] || head != hh[i]) { if (bad < 5) printf("  lane %d: after %d/%d sum %g/%g head %d/%d\n", i, ha[i], a, hs[i], s, hh[i], head); bad++; }
    }
    printf("trial %d width %d: %d mismatches\n", trial, W, bad);
  }
  return 0;
}
