// Probe: does a global_load read its address VGPRs at issue for ALL lanes?  Each wave issues a dwordx4 load from addresses held in
// a VGPR pair and overwrites that pair with another valid address in the very next instruction(s); the loaded data must belong to
// the ORIGINAL address.  hipcc --offload-arch=gfx950 -O3 -o tools/vmem_war_probe.bin tools/vmem_war_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(512) void k_probe(const float* buf, unsigned* errors, int reps, int n_elems) {
  const int lane = threadIdx.x & 63;
  unsigned err = 0;
  const int wid = blockIdx.x * 8 + (threadIdx.x >> 6);
  for (int it = 0; it < reps; ++it) {
    // element index (multiple of 4), scattered so that the loads miss/hit differently
    const unsigned e0 = (((unsigned)(lane * 37 + it * 101 + wid * 13) * 2654435761u) >> 8) % (unsigned)(n_elems / 4 - 4096) * 4u;
    const float* p = buf + e0;
    const float* other = buf + ((e0 + 4096u * 4u) % (unsigned)(n_elems - 8));
    f32x4 d;
    uint64_t a = (uint64_t)p, b = (uint64_t)other;
    if (VARIANT == 0) {  // overwrite by a 64-bit move immediately
      asm volatile("global_load_dwordx4 %0, %1, off\n v_mov_b64 %1, %2\n s_waitcnt vmcnt(0)"
                   : "=&v"(d), "+v"(a) : "v"(b) : "memory");
    } else if (VARIANT == 1) {  // overwrite by a 64-bit add (as the gather's v_lshl_add_u64)
      asm volatile("global_load_dwordx4 %0, %1, off\n v_lshl_add_u64 %1, %2, 0, 0\n s_waitcnt vmcnt(0)"
                   : "=&v"(d), "+v"(a) : "v"(b) : "memory");
    } else if (VARIANT == 2) {  // two loads from the same pair with offsets, then overwrite (the gather's pattern)
      f32x4 d2;
      asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:64\n v_lshl_add_u64 %2, %3, 0, 0\n s_waitcnt vmcnt(0)"
                   : "=&v"(d), "=&v"(d2), "+v"(a) : "v"(b) : "memory");
      if (d2.x != (float)(e0 + 16)) ++err;
    } else {  // destination overlaps the address pair (fixed registers)
      asm volatile("v_mov_b64 v[100:101], %1\n s_nop 4\n global_load_dwordx4 v[100:103], v[100:101], off\n s_waitcnt vmcnt(0)\n"
                   "v_mov_b32 %0, v100" : "=v"(d.x) : "v"(a) : "memory", "v100", "v101", "v102", "v103");
      d.y = (float)(e0 + 1); d.z = (float)(e0 + 2); d.w = (float)(e0 + 3);
    }
    if (d.x != (float)e0 || d.y != (float)(e0 + 1) || d.z != (float)(e0 + 2) || d.w != (float)(e0 + 3)) ++err;
  }
  if (err) atomicAdd(&errors[lane >> 4], err);
}

template <int VARIANT>
void run(const float* buf, unsigned* d_err, int n, const char* name) {
  (void)hipMemset(d_err, 0, 4 * sizeof(unsigned));
  k_probe<VARIANT><<<256, 512>>>(buf, d_err, 4000, n);
  (void)hipDeviceSynchronize();
  unsigned h[4];
  (void)hipMemcpy(h, d_err, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-70s errors by 16-lane row: %u %u %u %u\n", name, h[0], h[1], h[2], h[3]);
}

// ---- read-after-write: the address pair is produced by the instruction(s) right before the load (as in the gather: v_mul_lo_u32,
// v_add_u32, v_lshl_add_u64 -> global_load); a stale pair would fetch the PREVIOUS iteration's element -------------------------------
template <int VARIANT>
__global__ __launch_bounds__(512) void k_raw(const float* buf, unsigned* errors, int reps, int n_elems) {
  const int lane = threadIdx.x & 63;
  unsigned err = 0;
  const int wid = blockIdx.x * 8 + (threadIdx.x >> 6);
  const uint64_t base = (uint64_t)buf;
  for (int it = 0; it < reps; ++it) {
    const unsigned idx = (((unsigned)(lane * 37 + it * 101 + wid * 13) * 2654435761u) >> 8) % (unsigned)(n_elems / 48 - 8);  // "texel" index
    unsigned texel = idx, mul = 48;
    asm volatile("" : "+v"(texel), "+v"(mul));
    float d0;
    if (VARIANT == 0) {        // offset = texel * 48 floats; address = base + (offset << 2); load at once
      asm volatile("v_mul_lo_u32 v200, %1, %2\n v_mov_b32 v201, 0\n v_lshl_add_u64 v[202:203], v[200:201], 2, %3\n"
                   "global_load_dwordx4 v[204:207], v[202:203], off\n s_waitcnt vmcnt(0)\n v_mov_b32 %0, v204"
                   : "=v"(d0) : "v"(texel), "v"(mul), "v"(base) : "memory", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207");
    } else if (VARIANT == 1) { // same with three loads at offsets 0 / 64 / 128 (one texel = three lines) and the next address computed meanwhile
      asm volatile("v_mul_lo_u32 v200, %1, %2\n v_mov_b32 v201, 0\n v_lshl_add_u64 v[202:203], v[200:201], 2, %3\n"
                   "global_load_dwordx4 v[204:207], v[202:203], off\n global_load_dwordx4 v[208:211], v[202:203], off offset:64\n"
                   "v_add_u32 v200, 48, v200\n global_load_dwordx4 v[212:215], v[202:203], off offset:128\n"
                   "v_lshl_add_u64 v[202:203], v[200:201], 2, %3\n s_waitcnt vmcnt(0)\n v_mov_b32 %0, v204\n"
                   : "=v"(d0) : "v"(texel), "v"(mul), "v"(base)
                   : "memory", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215");
    } else {                   // 64-bit add through v_add_co / v_addc
      asm volatile("v_mul_lo_u32 v200, %1, %2\n v_lshlrev_b32 v200, 2, v200\n v_add_co_u32 v202, vcc, %L3, v200\n"
                   "v_addc_co_u32 v203, vcc, 0, %H3, vcc\n global_load_dwordx4 v[204:207], v[202:203], off\n s_waitcnt vmcnt(0)\n v_mov_b32 %0, v204"
                   : "=v"(d0) : "v"(texel), "v"(mul), "v"(base) : "memory", "vcc", "v200", "v202", "v203", "v204", "v205", "v206", "v207");
    }
    if (d0 != (float)(idx * 48)) ++err;
  }
  if (err) { atomicAdd(&errors[lane >> 4], err); }
}
template <int VARIANT>
void run_raw(const float* buf, unsigned* d_err, int n, const char* name) {
  (void)hipMemset(d_err, 0, 4 * sizeof(unsigned));
  k_raw<VARIANT><<<256, 512>>>(buf, d_err, 8000, n);
  (void)hipDeviceSynchronize();
  unsigned h[4];
  (void)hipMemcpy(h, d_err, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-70s errors by 16-lane row: %u %u %u %u\n", name, h[0], h[1], h[2], h[3]);
}

int main() {
  const int n = 1 << 24;
  float* h = new float[n];
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float* buf; unsigned* d_err;
  (void)hipMalloc(&buf, n * sizeof(float));
  (void)hipMalloc(&d_err, 4 * sizeof(unsigned));
  (void)hipMemcpy(buf, h, n * sizeof(float), hipMemcpyHostToDevice);
  run<0>(buf, d_err, n, "address pair overwritten by 2 x v_mov right after the load");
  run<1>(buf, d_err, n, "address pair overwritten by v_lshl_add_u64 right after the load");
  run<2>(buf, d_err, n, "two loads (offset 0 / 64) then v_lshl_add_u64 on the pair");
  run<3>(buf, d_err, n, "load whose destination overlaps its address pair");
  run_raw<0>(buf, d_err, n, "RAW: v_mul_lo_u32, v_lshl_add_u64 then the load at once");
  run_raw<1>(buf, d_err, n, "RAW: three loads per address, next address computed between them");
  return 0;
}
