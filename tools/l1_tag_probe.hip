// Probe (round 2): what does a wave64 global_load_dwordx4 cost in the vector L1 as a function of how many distinct 64-byte and
// 128-byte lines its 64 lanes touch?  All addresses hit a 16 KB L1-resident buffer; 8 waves per CU issue REPS loads each.
//   pattern 0: one 64-B segment per 4-lane team, 16 segments in 16 different 128-B lines
//   pattern 1: 16 segments forming 8 aligned 128-B pairs (team 2k and 2k+1 share a 128-B line)
//   pattern 2: 16 segments, consecutive (one contiguous 1 KB block)
//   pattern 3: every lane in a different 64-B line (64 lines)
//   pattern 4: every lane the same 16 bytes (broadcast)
//   pattern 5: 8 segments of 128 B contiguous, one per 8-lane team, in 8 different 256-B blocks
//   pattern 6: 16 teams but only 4 distinct segments (teams 4k..4k+3 read the same 64 B)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k(const char* buf, float* out, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned off;
  const int team = lane >> 2, p = lane & 3;
  if (PAT == 0) off = team * 256 + p * 16;
  if (PAT == 1) off = (team >> 1) * 256 + (team & 1) * 64 + p * 16;
  if (PAT == 2) off = lane * 16;
  if (PAT == 3) off = lane * 64;
  if (PAT == 4) off = 0;
  if (PAT == 5) off = (lane >> 3) * 256 + (lane & 7) * 16;
  if (PAT == 6) off = (team >> 2) * 256 + p * 16;
  off = (off + wave * 4096 * 0) & 16383;
  f32x4 acc = {0, 0, 0, 0};
  unsigned rot = 0;
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x4 v = *(const f32x4*)(buf + ((off + rot) & 16383));
      acc += v;
      rot += 4096 + 256;  // walk the 16 KB buffer so that successive loads are not the very same lines
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int PAT>
float run(const char* buf, float* out) {
  const int reps = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<PAT><<<256, 512>>>(buf, out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<PAT><<<256, 512>>>(buf, out, reps);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / (reps * 8.0f * 8.0f);  // ns per load instruction per CU (8 waves x 8 loads per rep)
}

int main() {
  char* buf; float* out;
  (void)hipMalloc(&buf, 1 << 20); (void)hipMemset(buf, 0, 1 << 20);
  (void)hipMalloc(&out, 256 * 512 * 4);
  const char* names[7] = {"16 x 64 B in 16 different 128-B lines", "16 x 64 B as 8 aligned 128-B pairs", "1 KB contiguous", "64 lanes in 64 different 64-B lines",
                          "all lanes the same 16 B", "8 x 128 B contiguous (8-lane teams)", "16 teams, 4 distinct 64-B segments"};
  float r[7] = {run<0>(buf, out), run<1>(buf, out), run<2>(buf, out), run<3>(buf, out), run<4>(buf, out), run<5>(buf, out), run<6>(buf, out)};
  for (int rep = 0; rep < 1; ++rep) { r[0] = run<0>(buf, out); r[1] = run<1>(buf, out); r[2] = run<2>(buf, out); r[3] = run<3>(buf, out); r[4] = run<4>(buf, out); r[5] = run<5>(buf, out); r[6] = run<6>(buf, out); }
  for (int i = 0; i < 7; ++i) printf("%-44s %7.2f ns per wave-load per CU  (~%5.1f clk at 2.1 GHz)\n", names[i], r[i], r[i] * 2.1f);
  return 0;
}
