// Probe for DESIGN.md 5.1 (round 3): packed fp32 instructions that broadcast the HIGH half of a register pair, executed while the
// same wave and the co-resident wave of the SIMD have gather-style global_load_dwordx4 in flight (the structure of the failing
// gather: loads of round k + 1 issued, then the interpolation of round k).  Every wave checks (a) the packed results against
// scalar arithmetic and (b) every loaded dword against the value its address must hold.  Two 256-VGPR waves per SIMD.
//   FORM 0: high-half broadcast (v_pk_fma_f32 D, W, X, D op_sel:[1,0,0]; v_pk_mul_f32 D, X, W op_sel:[0,1])
//   FORM 1: low-half broadcast control (op_sel_hi:[0,1,1] / [1,0])
//   PAD  1: two s_nop 7 in front of every packed instruction (made the real kernel fail in every call)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline float cell_value(uint32_t idx, int c) {
  uint32_t h = idx * 2654435761u + (uint32_t)c * 40503u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return (float)(h & 0xffff) * (1.0f / 256.0f) + 1.0f;   // exactly representable, never zero
}

template <int FORM, int PAD>
__device__ __forceinline__ f32x2 pk_fma_w(f32x2 d, f32x2 w, f32x2 x) {   // d + w.{hi|lo} * x
  if (PAD) asm volatile("s_nop 7\n s_nop 7");
  if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(d) : "v"(w), "v"(x));
  else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(d) : "v"(w), "v"(x));
  return d;
}
template <int FORM, int PAD>
__device__ __forceinline__ f32x2 pk_mul_w(f32x2 x, f32x2 w) {   // x * w.{hi|lo}
  f32x2 d;
  if (PAD) asm volatile("s_nop 7\n s_nop 7");
  if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(x), "v"(w));
  else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(w));
  return d;
}

constexpr int NLOAD = 12;

template <int FORM, int PAD>
__global__ __launch_bounds__(512) void k_probe(const f32x4* __restrict__ src, uint32_t n_cells, unsigned* errors, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  asm volatile("" ::: "v255");   // 256-VGPR allocation: two waves per SIMD, the second one in the upper half of the file
  unsigned err_pk = 0, err_ld = 0;
  f32x4 cur[NLOAD], nxt[NLOAD];
  uint32_t cidx[NLOAD], nidx[NLOAD];
  uint32_t seed = (blockIdx.x * 8 + wave) * 64 + lane;
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) { cidx[k] = (seed * 977u + k * 131071u) % n_cells; cur[k] = src[cidx[k]]; }
  for (int it = 0; it < reps; ++it) {
    // round it + 1: issue the loads (team-like: 4 neighbouring lanes read one 64-byte line)
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      nidx[k] = (((seed >> 2) * 7919u + (uint32_t)it * 104729u + k * 15485863u) % (n_cells / 4)) * 4 + (lane & 3);
      nxt[k] = src[nidx[k]];
    }
    // round it: interpolate with the weight pair, loads in flight
    const float w0 = 0.25f + (float)((it + lane) & 7) * 0.125f, w1 = 1.0f - w0 * 0.5f;
    f32x2 w = {w0, w1};
    asm volatile("" : "+v"(w));
#pragma unroll
    for (int k = 0; k + 1 < NLOAD; k += 2) {
      const f32x2 a0 = {cur[k].x, cur[k].y}, a1 = {cur[k].z, cur[k].w}, b0 = {cur[k + 1].x, cur[k + 1].y}, b1 = {cur[k + 1].z, cur[k + 1].w};
      f32x2 p0 = pk_mul_w<FORM, PAD>(a0, w), p1 = pk_mul_w<FORM, PAD>(a1, w);
      p0 = pk_fma_w<FORM, PAD>(p0, w, b0);
      p1 = pk_fma_w<FORM, PAD>(p1, w, b1);
      const float ws = FORM == 0 ? w1 : w0;
      const float e0 = fmaf(ws, cur[k + 1].x, cur[k].x * ws), e1 = fmaf(ws, cur[k + 1].y, cur[k].y * ws);
      const float e2 = fmaf(ws, cur[k + 1].z, cur[k].z * ws), e3 = fmaf(ws, cur[k + 1].w, cur[k].w * ws);
      if (p0.x != e0 || p0.y != e1 || p1.x != e2 || p1.y != e3) ++err_pk;
    }
    // verify what round it loaded (before it is replaced), then rotate
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      if (cur[k].x != cell_value(cidx[k], 0) || cur[k].y != cell_value(cidx[k], 1) || cur[k].z != cell_value(cidx[k], 2) ||
          cur[k].w != cell_value(cidx[k], 3)) ++err_ld;
      cur[k] = nxt[k]; cidx[k] = nidx[k];
    }
  }
  if (err_pk) { atomicAdd(&errors[0 + (wave >= 4)], 1u); atomicAdd(&errors[4 + (lane >> 4)], err_pk); }
  if (err_ld) { atomicAdd(&errors[2 + (wave >= 4)], 1u); atomicAdd(&errors[8 + (lane >> 4)], err_ld); }
}

template <int FORM, int PAD>
void run(const f32x4* src, uint32_t n, unsigned* d_err, const char* name) {
  (void)hipMemset(d_err, 0, 64);
  k_probe<FORM, PAD><<<256, 512>>>(src, n, d_err, 20000);
  (void)hipDeviceSynchronize();
  unsigned h[16];
  (void)hipMemcpy(h, d_err, 64, hipMemcpyDeviceToHost);
  printf("%-34s lanes with wrong PACKED results waves 0-3 / 4-7: %u / %u (by 16-lane row %u %u %u %u) | wrong LOADED data: %u / %u (by row %u %u %u %u)\n",
         name, h[0], h[1], h[4], h[5], h[6], h[7], h[2], h[3], h[8], h[9], h[10], h[11]);
}

int main() {
  const uint32_t n = 1u << 20;   // 16 MB of float4 cells: L2 hits and misses
  std::vector<f32x4> host(n);
  for (uint32_t i = 0; i < n; ++i) host[i] = f32x4{cell_value(i, 0), cell_value(i, 1), cell_value(i, 2), cell_value(i, 3)};
  f32x4* src; unsigned* d_err;
  (void)hipMalloc(&src, n * sizeof(f32x4)); (void)hipMalloc(&d_err, 64);
  (void)hipMemcpy(src, host.data(), n * sizeof(f32x4), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0>(src, n, d_err, "high-half broadcast");
    run<0, 1>(src, n, d_err, "high-half broadcast + s_nop pad");
    run<1, 0>(src, n, d_err, "low-half broadcast (control)");
    run<1, 1>(src, n, d_err, "low-half broadcast + s_nop pad");
  }
  hipError_t e = hipGetLastError();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
