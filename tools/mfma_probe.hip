// Hardware probe (gfx950): (1) does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs?  (2) which k does
// element e of lane-half hh feed (A and B)?  Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe(const float* A /*[32][16]*/, const float* B /*[16][32]*/, int map, float* D /*[32][32]*/) {
  const int l = threadIdx.x, hh = l >> 5, ij = l & 31;
  h8 a, b;
  for (int e = 0; e < 8; ++e) {
    const int k = map == 0 ? 8 * hh + e : 4 * hh + (e & 3) + 8 * (e >> 2);
    a[e] = (_Float16)A[ij * 16 + k];
    b[e] = (_Float16)B[k * 32 + ij];
  }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + ij] = c[r];
}

int main() {
  float hA[512], hB[512], hD[1024], *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  for (int map = 0; map < 2; ++map) {
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7 + 3) % 11 - 5); hB[i] = (float)((i * 5 + 1) % 13 - 6); }
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, map, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j];
      err = fmax(err, fabs(s - hD[i * 32 + j]));
    }
    printf("layout map %d (0: k=8*hh+e, 1: k=4*hh+(e&3)+8*(e>>2)): max err %g  (any consistent A/B map is exact for GEMM)\n", map, err);
  }
  // subnormal test: A = 2^-20 (fp16 subnormal), B = 1
  for (int i = 0; i < 512; ++i) { hA[i] = ldexpf(1.f, -20); hB[i] = 1.f; }
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, 0, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  printf("subnormal A=2^-20 x B=1, K=16: D = %g (honoured: %g, flushed: 0)\n", hD[0], 16 * ldexp(1.0, -20));
  for (int i = 0; i < 512; ++i) { hA[i] = ldexpf(1.f, -20); hB[i] = ldexpf(1.f, -20); }
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, 0, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  printf("subnormal x subnormal: D = %g (expected %g)\n", hD[0], 16 * ldexp(1.0, -40));
  return 0;
}
