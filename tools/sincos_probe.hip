// accuracy of the hardware transcendental path (v_sin_f32 / v_cos_f32 on x / 2pi) vs the library, for PE arguments
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const float* x, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float t = x[i] * 0.15915494309189533577f;  // revolutions
  s[i] = __builtin_amdgcn_sinf(t);
  c[i] = __builtin_amdgcn_cosf(t);
}
int main() {
  const int n = 1 << 20;
  float *hx = new float[n], *hs = new float[n], *hc = new float[n], *dx, *ds, *dc;
  for (int i = 0; i < n; ++i) hx[i] = -12.f + 24.f * (float)i / n;
  hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, ds, dc, n);
  hipMemcpy(hs, ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, n * 4, hipMemcpyDeviceToHost);
  double es = 0, ec = 0, es1 = 0;
  for (int i = 0; i < n; ++i) {
    es = fmax(es, fabs(hs[i] - sin((double)hx[i]))); ec = fmax(ec, fabs(hc[i] - cos((double)hx[i])));
    if (fabs(hx[i]) < 2.0) es1 = fmax(es1, fabs(hs[i] - sin((double)hx[i])));
  }
  printf("v_sin/v_cos on [-12,12]: max |err| sin %.3e cos %.3e ; sin on [-2,2] %.3e\n", es, ec, es1);
  return 0;
}
