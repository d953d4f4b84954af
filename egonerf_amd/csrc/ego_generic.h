// Internal (not part of the C ABI): the any-shape compatibility kernels of ego_generic.hip, called by the ABI entry points of
// ego_shade.hip / ego_ops.hip when the scene's shape is not the tuned one.
#pragma once
#include "../../include/egonerf_hip.h"

bool ego_shape_is_tuned(const ego_scene* sc);   // app_dim 27, 48 appearance components, MLP_Fea 150 -> 128 -> 128 -> 3 with view_pe = fea_pe = 2
int64_t ego_generic_packed_floats(const ego_scene* sc);
int ego_generic_pack(const ego_scene* sc, float* out, void* stream);
int ego_generic_shade(const ego_scene* sc, const float* rays, const float* coords, int64_t N, int32_t S, float* rgb, const uint8_t* tile_active,
                      void* stream);
int ego_generic_app_feature(const ego_scene* sc, const float* c7n, int64_t M, float* out, void* stream);
int ego_generic_mlp_fea(const ego_scene* sc, const float* viewdirs, const float* feat, int64_t M, float* rgb, void* stream);
int ego_generic_march(const ego_scene* sc, const ego_vm_field& f, bool fine_lut, const float* rays, int64_t N, int32_t S, const float* z_in,
                      const float* r_sched, const float* jitter, float near_, const uint8_t* occ, float* z_out, float* alpha, int32_t alpha_stride,
                      float* weight, float* bg_weight, float* coords_out, float* sigma_out, uint8_t* tile_active, void* stream);
