// libegonerf_hip.so, part 3: EgoNeRF.forward as one call (models/EgoNeRF.py:491-602) — a fixed
// sequence of launches on the caller's stream, no host synchronisation, no allocation.
#include "ego_host.h"

namespace {
struct Plan {
  int64_t zc, wc, zf, w, bg, rgb, crd, act, total;  // float offsets
  int32_t S_out;
};

inline int64_t align64(int64_t v) { return (v + 63) & ~int64_t(63); }

Plan make_plan(int64_t N, const ego_render_args* a) {
  Plan p{};
  const int64_t Sc = a->n_coarse;
  p.S_out = a->resampling ? (a->use_coarse_sample ? a->n_coarse + a->n_fine : a->n_fine) : a->n_coarse;
  int64_t o = 0;
  p.zc = o; o = align64(o + N * Sc);
  p.wc = o; o = align64(o + (a->resampling ? N * Sc : 0));
  p.zf = o; o = align64(o + (a->resampling ? N * (int64_t)p.S_out : 0));
  p.w = o; o = align64(o + N * (int64_t)p.S_out);
  p.bg = o; o = align64(o + N);
  p.rgb = o; o = align64(o + N * (int64_t)p.S_out * 3);
  p.crd = o; o = align64(o + N * (int64_t)p.S_out * 4);
  p.act = o; o = align64(o + (N * (int64_t)p.S_out / 32 + 1 + 3) / 4);  // tile flags (bytes)
  p.total = o;
  return p;
}
}  // namespace

extern "C" {

/* lets a binding verify its struct mirrors: which = 0 ego_scene, 1 ego_render_args, 2 ego_vm_field, 3 ego_adam_tensor */
int64_t ego_sizeof(int32_t which) {
  return which == 0 ? (int64_t)sizeof(ego_scene) : which == 1 ? (int64_t)sizeof(ego_render_args) : which == 2 ? (int64_t)sizeof(ego_vm_field)
       : which == 3 ? (int64_t)sizeof(ego_adam_tensor) : which == 4 ? (int64_t)sizeof(ego_shade_dump) : -1;
}

int64_t ego_render_workspace_bytes(int64_t N, const ego_render_args* args) {
  if (!args || N < 0 || args->n_coarse < 2) return -1;
  return make_plan(N, args).total * (int64_t)sizeof(float);
}

int ego_render_forward(const ego_scene* sc, const ego_render_args* a, const float* rays, int64_t N, void* workspace,
                       float* rgb_map, float* depth, float* alpha, float* bg_map, float* env_map, void* stream) {
  EGO_TRACE("ego_render_forward");
  EGO_REQUIRE(N >= 0, "render_forward: N < 0");
  if (N == 0) return EGO_OK;
  EGO_REQUIRE(sc && a && rays && workspace && rgb_map, "render_forward: null argument");
  EGO_REQUIRE((a->r_sched || a->z_coarse) && a->n_coarse >= 2, "render_forward: r_sched / z_coarse missing or n_coarse < 2");
  EGO_REQUIRE(!a->resampling || a->n_fine >= 1, "render_forward: resampling needs n_fine >= 1");
  const Plan p = make_plan(N, a);
  float* ws = (float*)workspace;
  const int32_t S = p.S_out;
  const int32_t astride = sc->envmap ? S + 1 : S;
  int e;
  const float* z;
  // tile-level skipping only when a mask, early termination or the weight threshold is switched on (otherwise every sample is shaded)
  uint8_t* act = (sc->occ || sc->term_eps > 0.f || sc->weight_thres >= 0.f) ? (uint8_t*)(ws + p.act) : nullptr;
  if (act && (S & 31) != 0) {  // with whole tiles per ray the march writes every flag itself (0 or 1)
    const hipError_t me = hipMemsetAsync(act, 0, (size_t)(N * (int64_t)S / 32 + 1), (hipStream_t)stream);
    if (me != hipSuccess) return ego_fail((int)me, "render_forward: hipMemsetAsync failed: %s", hipGetErrorString(me));
  }
  const float* zc_in = a->z_coarse;  // explicit first-pass distances (exp_sampling=False) or NULL (schedule + jitter)
  if (a->resampling) {
    // coarse pass on the pooled tables -> weights -> inverse-CDF samples merged into the coarse schedule
    if ((e = ego_march_density(sc, rays, N, a->n_coarse, zc_in, zc_in ? nullptr : a->r_sched, zc_in ? nullptr : a->jitter, a->near_, 1,
                               zc_in ? nullptr : ws + p.zc, nullptr, 0, ws + p.wc, nullptr, nullptr, nullptr, nullptr, stream))) return e;
    if ((e = ego_sample_pdf_merge(zc_in ? zc_in : ws + p.zc, ws + p.wc, a->u, N, a->n_coarse, a->n_fine, a->use_coarse_sample, ws + p.zf,
                                  nullptr, stream))) return e;
    if ((e = ego_march_density(sc, rays, N, S, ws + p.zf, nullptr, nullptr, a->near_, 2 /* fine pass: full tables, fine LUT */, nullptr, alpha, astride, ws + p.w,
                               ws + p.bg, ws + p.crd, nullptr, act, stream))) return e;
    z = ws + p.zf;
  } else {
    if ((e = ego_march_density(sc, rays, N, S, zc_in, zc_in ? nullptr : a->r_sched, zc_in ? nullptr : a->jitter, a->near_, 0,
                               zc_in ? nullptr : ws + p.zc, alpha, astride, ws + p.w, ws + p.bg, ws + p.crd, nullptr, act, stream))) return e;
    z = zc_in ? zc_in : ws + p.zc;
  }
  if (a->marched)
    if (const hipError_t ee = hipEventRecord((hipEvent_t)a->marched, (hipStream_t)stream))
      return ego_fail((int)ee, "render_forward: hipEventRecord(marched) failed: %s", hipGetErrorString(ee));
  // one launch for shading + compositing wherever it applies and its ray-granular deal of the work is balanced (ego_render_forward_folds):
  // since the folded kernel loads a plane's basis fragments ahead of the next plane's taps (it has the registers for that, the two-launch
  // kernel does not) it is 3.5 % faster than ego_shade, 2.8 % at step level (round 5; before that it lost by 0.25 %).  EGO_RENDER_FOLD=0
  // keeps the two launches, =1 folds whenever the scene allows it.
  if (ego_render_forward_folds(sc, N, S))
    return ego_shade_composite(sc, rays, z, ws + p.crd, ws + p.w, ws + p.bg, N, S, act, rgb_map, depth, bg_map, env_map, stream);
  if ((e = ego_shade(sc, rays, z, ws + p.crd, N, S, ws + p.rgb, nullptr, act, stream))) return e;
  return ego_composite(sc, rays, z, ws + p.w, ws + p.bg, ws + p.rgb, N, S, rgb_map, depth, bg_map, env_map, nullptr, stream);
}

}  // extern "C"
