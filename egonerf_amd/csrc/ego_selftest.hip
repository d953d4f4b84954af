// libegonerf_hip.so, part 8: run-time self-test of the team-gather kernel family.
//
// Builds of ego_shade.hip that contain packed fp32 instructions broadcasting the HIGH dword of a register pair returned one
// gather round's products wrong in 0.3 % ... 100 % of the calls on MI355X, only with two waves co-running on a SIMD, and the
// mechanism is not understood.  The build fences it off (register pinning in the source, -fno-slp-vectorize, an ISA check of the
// shipped code objects, a compiler pin); this is the run-time fence: ego_selftest() runs the SHIPPED kernels of that family
// (ego_app_feature = k_shade_h<APP>, ego_shade in both fp16 arithmetics = the rolling gather) on a fixed synthetic tile set
// `reps` times through the public entry points and bit-compares every result with the first one.  The forward path has no
// atomics, so any difference is the fault (or another one like it).  The host layer calls it once per process and device from
// egonerf_amd/_lib.py and refuses to continue on a mismatch.
//
// The library allocates nothing: the caller hands in ego_selftest_workspace_bytes() bytes of device memory.
#include "ego_host.h"

namespace {

constexpr int NR = 10, NTH = 10, NPH = 30, C = 48, APP_DIM = 27, MLP_IN = 150, HID = 128;
constexpr int M_APP = 512;             // one workgroup of 8 waves, two tiles per wave: tools/flaky_probe.py's shape
constexpr int M_APP2 = 512 * 24;       // 24 workgroups
constexpr int N_RAYS = 64, S = 64;     // fused kernel: 4096 samples = 128 tiles
constexpr int LUT_N = NR + 1;

struct Layout {
  int64_t plane[2][3], line[2][3], basis[2], w[3], b[3], lut, c7_in, c7_sc, c7_big, rays, z, coords, packed, out_ref, out, counter, total;
};

__host__ int64_t align64(int64_t floats) { return (floats + 63) & ~(int64_t)63; }   // 256-byte alignment in floats

__host__ Layout make_layout() {
  Layout L{};
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += align64(n); return at; };
  const int px[3] = {NR, NR, NTH}, py[3] = {NTH, NPH, NPH}, ln[3] = {NPH, NTH, NR};
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i) { L.plane[g][i] = take((int64_t)px[i] * py[i] * C); L.line[g][i] = take((int64_t)ln[i] * C); }
  for (int g = 0; g < 2; ++g) L.basis[g] = take(APP_DIM * 3 * C);
  L.w[0] = take(HID * MLP_IN); L.w[1] = take(HID * HID); L.w[2] = take(3 * HID);
  L.b[0] = take(HID); L.b[1] = take(HID); L.b[2] = take(4);
  L.lut = take(LUT_N);
  L.c7_in = take(M_APP * 7); L.c7_sc = take(M_APP * 7); L.c7_big = take(M_APP2 * 7);
  L.rays = take(N_RAYS * 6); L.z = take(N_RAYS * S); L.coords = take(N_RAYS * S * 4);
  L.packed = take(ego_packed_floats());
  L.out_ref = take((int64_t)M_APP2 * APP_DIM); L.out = take((int64_t)M_APP2 * APP_DIM);
  L.counter = take(64);
  L.total = o;
  return L;
}

__device__ inline uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ inline float unit(uint32_t seed, uint32_t i) { return (float)(mix(seed * 0x9e3779b9u + i) >> 8) * (1.0f / 16777216.0f); }  // [0, 1)

// one fill kernel for the whole workspace: region r = [lo, hi) gets kind-specific values
struct Fill { int64_t lo, n; int kind; float a, b; };
struct Fills { Fill f[40]; int n; };

__global__ void k_selftest_fill(float* ws, Fills F) {
  for (int r = 0; r < F.n; ++r) {
    const Fill f = F.f[r];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < f.n; i += (int64_t)gridDim.x * blockDim.x) {
      float v;
      const uint32_t u = (uint32_t)i;
      switch (f.kind) {
        case 0: v = f.a * (unit(r + 1, u) - 0.5f); break;                                   // uniform in [-a/2, a/2)
        case 1: {                                                                             // c7n rows: 6 coords in [a, b), col 6 = grid flag
          const int col = (int)(i % 7);
          v = col == 6 ? (unit(r + 1, u) > 0.5f ? 1.f : 0.f) : f.a + (f.b - f.a) * unit(r + 1, u);
          if (col < 6) {   // the other grid's three columns are zero in the reference's 7-vector
            const bool yang = unit(r + 1, u - col + 6) > 0.5f;
            if ((col < 3) == yang) v = 0.f;
          }
          break;
        }
        case 2: {                                                                             // coords [M][4]: 3 in [a, b), flag
          const int col = (int)(i & 3);
          v = col == 3 ? (unit(r + 1, u >> 6) > 0.7f ? 1.f : 0.f) : f.a + (f.b - f.a) * unit(r + 1, u);   // the flag changes every 16 samples: mixed tiles
          break;
        }
        case 3: v = f.a + (f.b - f.a) * (float)i / (float)(f.n - 1); break;                   // monotone LUT
        default: v = 0.f;
      }
      ws[f.lo + i] = v;
    }
  }
}

// one workgroup per call: counter += 1 iff any word of the two results differs (launches are serial on the stream: no atomic)
__global__ __launch_bounds__(1024) void k_selftest_compare(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, int64_t n, uint32_t* counter) {
  int bad = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) bad |= a[i] != b[i];
  const int any = __syncthreads_or(bad);
  if (threadIdx.x == 0 && any) counter[0] += 1u;
}

}  // namespace

extern "C" {

int64_t ego_selftest_workspace_bytes(void) { return make_layout().total * 4; }

int ego_selftest(void* workspace, int64_t workspace_bytes, int32_t reps, int32_t* mismatching_calls, void* stream_) {
  EGO_TRACE("ego_selftest");
  EGO_REQUIRE(workspace && mismatching_calls && reps >= 1, "selftest: null argument or reps < 1");
  const Layout L = make_layout();
  EGO_REQUIRE(workspace_bytes >= L.total * 4 && ((uintptr_t)workspace & 255) == 0, "selftest: workspace too small (ego_selftest_workspace_bytes) or not 256-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  float* ws = (float*)workspace;
  Fills F{};
  auto add = [&](int64_t lo, int64_t n, int kind, float a, float b) { F.f[F.n++] = Fill{lo, n, kind, a, b}; };
  const int px[3] = {NR, NR, NTH}, py[3] = {NTH, NPH, NPH}, ln[3] = {NPH, NTH, NR};
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i) { add(L.plane[g][i], (int64_t)px[i] * py[i] * C, 0, 2.f, 0.f); add(L.line[g][i], (int64_t)ln[i] * C, 0, 2.f, 0.f); }
  for (int g = 0; g < 2; ++g) add(L.basis[g], APP_DIM * 3 * C, 0, 0.3f, 0.f);
  add(L.w[0], HID * MLP_IN, 0, 0.3f, 0.f); add(L.w[1], HID * HID, 0, 0.3f, 0.f); add(L.w[2], 3 * HID, 0, 0.3f, 0.f);
  add(L.b[0], HID, 0, 0.2f, 0.f); add(L.b[1], HID, 0, 0.2f, 0.f); add(L.b[2], 4, 0, 0.2f, 0.f);
  add(L.lut, LUT_N, 3, 0.f, 20.f);
  add(L.c7_in, M_APP * 7, 1, -0.999f, 0.999f);     // in range
  add(L.c7_sc, M_APP * 7, 1, -1.3f, 1.3f);         // scattered (out-of-range taps -> zero padding)
  add(L.c7_big, (int64_t)M_APP2 * 7, 1, -1.05f, 1.05f);
  add(L.rays, N_RAYS * 6, 0, 2.f, 0.f);
  add(L.z, N_RAYS * S, 0, 1.f, 0.f);
  add(L.coords, N_RAYS * S * 4, 2, -1.02f, 1.02f);
  k_selftest_fill<<<256, 256, 0, stream>>>(ws, F);
  if (int e = ego_launch_status("k_selftest_fill")) return e;
  if (hipError_t e = hipMemsetAsync(ws + L.counter, 0, 256, stream)) return ego_fail((int)e, "selftest: memset failed");

  ego_scene sc{};
  sc.ang_near[0] = 0.785398f; sc.ang_near[1] = -2.356194f; sc.ang_inv[0] = 0.636620f; sc.ang_inv[1] = 0.212207f;
  sc.r_lut = ws + L.lut; sc.n_r_lut = LUT_N; sc.n_r = NR;
  sc.act_softplus = 1; sc.density_shift = -8.f; sc.distance_scale = 25.f;
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i) { sc.app.plane[g][i] = ws + L.plane[g][i]; sc.app.line[g][i] = ws + L.line[g][i]; }
  sc.app.n_comp = C; sc.app.res[0] = NR; sc.app.res[1] = NTH; sc.app.res[2] = NPH;
  sc.basis[0] = ws + L.basis[0]; sc.basis[1] = ws + L.basis[1]; sc.app_dim = APP_DIM;
  for (int k = 0; k < 3; ++k) { sc.mlp_w[k] = ws + L.w[k]; sc.mlp_b[k] = ws + L.b[k]; }
  sc.mlp_in = MLP_IN; sc.mlp_hidden = HID; sc.view_pe = 2; sc.fea_pe = 2;
  sc.weight_thres = -1.f;
  if (int e = ego_pack_mlp(&sc, ws + L.packed, stream)) return e;
  sc.packed = ws + L.packed;

  float* ref = ws + L.out_ref;
  float* out = ws + L.out;
  uint32_t* counters = (uint32_t*)(ws + L.counter);
  int slot = 0;
  auto soak = [&](auto&& launch, int64_t n_out, int n_reps) -> int {
    if (int e = launch(ref)) return e;
    for (int r = 0; r < n_reps; ++r) {
      if (int e = launch(out)) return e;
      k_selftest_compare<<<1, 1024, 0, stream>>>((const uint32_t*)ref, (const uint32_t*)out, n_out, counters + slot);
      if (int e = ego_launch_status("k_selftest_compare")) return e;
    }
    ++slot;
    return EGO_OK;
  };
  // the stand-alone gather (round-by-round load buffer), f16x3 kernel family: in-range, scattered, and a 24-workgroup launch
  sc.mlp_precision = EGO_PREC_F16X3;
  if (int e = soak([&](float* o) { return ego_app_feature(&sc, ws + L.c7_in, M_APP, o, stream); }, (int64_t)M_APP * APP_DIM, reps)) return e;
  if (int e = soak([&](float* o) { return ego_app_feature(&sc, ws + L.c7_sc, M_APP, o, stream); }, (int64_t)M_APP * APP_DIM, reps)) return e;
  if (int e = soak([&](float* o) { return ego_app_feature(&sc, ws + L.c7_big, M_APP2, o, stream); }, (int64_t)M_APP2 * APP_DIM, (reps + 7) / 8)) return e;
  // the fused kernels (rolling load buffer) in both fp16 arithmetics
  const int precs[3] = {EGO_PREC_F16X3, EGO_PREC_F16F8, EGO_PREC_F16F6};
  for (int p = 0; p < 3; ++p) {
    sc.mlp_precision = precs[p];
    if (int e = soak([&](float* o) { return ego_shade(&sc, ws + L.rays, ws + L.z, ws + L.coords, N_RAYS, S, o, nullptr, nullptr, stream); },
                     (int64_t)N_RAYS * S * 3, (reps + 1) / 2)) return e;
  }
  uint32_t host[8] = {0};
  if (hipError_t e = hipMemcpyAsync(host, counters, sizeof(host), hipMemcpyDeviceToHost, stream)) return ego_fail((int)e, "selftest: copy failed");
  if (hipError_t e = hipStreamSynchronize(stream)) return ego_fail((int)e, "selftest: %s", hipGetErrorString(e));
  int32_t total = 0;
  for (int k = 0; k < slot; ++k) total += (int32_t)host[k];
  *mismatching_calls = total;
  if (total)
    return ego_fail(EGO_E_UNSUPPORTED, "selftest: the gather kernels returned different bits on identical inputs (stage in-range %u, scattered %u, "
                    "24 workgroups %u, fused f16x3 %u, fused f16f8 %u, fused f16f6 %u mismatching calls): this build has the reproducibility fault of "
                    "DESIGN.md 5.1 on this device - rebuild with the pinned compiler / flags (egonerf_amd/build.py)", host[0], host[1], host[2], host[3], host[4], host[5]);
  return EGO_OK;
}

}  // extern "C"
