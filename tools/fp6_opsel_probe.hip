#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int OA, int OB>
__global__ void k(float* D, int sa, int sb) {
  v8i a, b;
  // e2m3 code 8 = 1.0 in every element: 6-bit pattern 001000 repeated
  unsigned w[6] = {0,0,0,0,0,0};
  for (int e = 0; e < 32; ++e) { int bit = 6*e, wi = bit>>5, sh = bit&31; w[wi] |= 8u << sh; if (sh > 26) w[wi+1] |= 8u >> (32-sh); }
  for (int i = 0; i < 6; ++i) { a[i] = w[i]; b[i] = w[i]; }
  a[6]=a[7]=b[6]=b[7]=0;
  v16f acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 2, 2, OA, sa, OB, sb);
  if (threadIdx.x == 0) D[0] = acc[0];
}
int main() {
  float* d; hipMalloc(&d, 16);
  const int sa = 127 | (120 << 8) | (125 << 16) | (130u << 24), sb = 127 | (126 << 8) | (124 << 16) | (131u << 24);
  float h;
#define RUN(OA, OB) k<OA, OB><<<1, 64>>>(d, sa, sb); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("opsel_a %d opsel_b %d: D = %g = 64 * 2^%g\n", OA, OB, h, log2(h / 64.0));
  RUN(0,0) RUN(1,0) RUN(2,0) RUN(3,0) RUN(0,1) RUN(0,2) RUN(0,3)
  printf("bytes a: 127 120 125 130 (exp 0 -7 -2 +3); b: 127 126 124 131 (0 -1 -3 +4)\n");
}
