// Probe (round 2): does an exec-masked wave64 global_load_dwordx4 cost the vector L1 less?  Same set-up as l1_tag_probe.hip (all
// hits in a 16 KB buffer, 8 waves per CU, every 4-lane team reads one 64-B segment in its own 128-B line); only a subset of the
// teams is active.  If the cost follows the active lanes, loads of a sample whose texel cell equals the previous sample's can be
// skipped per lane (the registers keep the old texels) and the L1 time of the gathers falls with them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(1024) void k(const char* buf, float* out, int reps) {
  const int lane = threadIdx.x & 63;
  const int team = lane >> 2, p = lane & 3;
  bool active = true;
  if (PAT == 1) active = team < 8;
  if (PAT == 2) active = (team & 1) == 0;
  if (PAT == 3) active = team < 4;
  if (PAT == 4) active = (team & 3) == 0;
  if (PAT == 5) active = team == 0;
  if (PAT == 6) active = (lane & 1) == 0;   // half of every team
  unsigned off = team * 256 + p * 16;
  f32x4 acc = {0, 0, 0, 0};
  unsigned rot = 0;
  if (active) {
    for (int it = 0; it < reps; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 v = *(const f32x4*)(buf + ((off + rot) & 16383));
        acc += v;
        rot += 4096 + 256;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int PAT>
float run(const char* buf, float* out, int threads = 512) {
  const int reps = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<PAT><<<256, threads>>>(buf, out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<PAT><<<256, threads>>>(buf, out, reps);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / (reps * 8.0f * (threads / 64.0f));
}

int main() {
  char* buf; float* out;
  (void)hipMalloc(&buf, 1 << 20); (void)hipMemset(buf, 0, 1 << 20);
  (void)hipMalloc(&out, 256 * 1024 * 4);
  const char* names[7] = {"64 lanes active (16 teams)", "teams 0-7 active (32 lanes)", "even teams active (32 lanes)", "teams 0-3 active (16 lanes)",
                          "every 4th team active (16 lanes)", "one team active (4 lanes)", "even lanes active (half of every team)"};
  float r[7];
  for (int rep = 0; rep < 2; ++rep) { r[0] = run<0>(buf, out); r[1] = run<1>(buf, out); r[2] = run<2>(buf, out); r[3] = run<3>(buf, out); r[4] = run<4>(buf, out); r[5] = run<5>(buf, out); r[6] = run<6>(buf, out); }
  for (int i = 0; i < 7; ++i) printf("%-44s %7.2f ns per wave-load per CU  (~%5.1f clk at 2.1 GHz)\n", names[i], r[i], r[i] * 2.1f);
  // occupancy sweep: a throughput limit gives the same ns per load per CU at any wave count, a latency limit scales with 1 / waves
  for (int threads = 128; threads <= 1024; threads *= 2)
    printf("waves per CU %2d: all lanes %6.2f ns, 16 lanes %6.2f ns, 4 lanes %6.2f ns per wave-load per CU\n", threads / 64, run<0>(buf, out, threads),
           run<3>(buf, out, threads), run<5>(buf, out, threads));
  return 0;
}
