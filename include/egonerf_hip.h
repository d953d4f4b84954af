/*
 * egonerf_hip.h — C ABI of libegonerf_hip.so: EgoNeRF's volume-rendering hot path on MI355X (gfx950).
 *
 * Plain pointers and sizes only (no torch types).  Every pointer marked "dev" is device memory owned
 * by the caller; the library allocates nothing, keeps no global mutable state and launches every
 * kernel on the hipStream_t handed in (`stream`, may be NULL = default stream).  All entry points
 * return 0 on success or a negative EGO_E_* / positive hipError_t code; ego_last_error() returns a
 * thread-local message.  The reference signals errors with Python exceptions / assert
 * (models/EgoNeRF.py:504,522); the Python host layer turns non-zero codes into RuntimeError.
 *
 * Reference interface replaced by each entry point (paths relative to the reference repo):
 *   ego_erp_rays            get_ray_directions_360 + get_rays      dataLoader/ray_utils.py:24-40, :85-113
 *   ego_sample_ray_exp      EgoNeRF.sample_ray_exp                models/EgoNeRF.py:56-87
 *   ego_from_cartesian      YinYangSphericalCoords.from_cartesian  models/coordinates.py:468-498
 *   ego_normalize_coord     YinYangSphericalCoords.normalize_coord models/coordinates.py:442-466 (+ :110-131)
 *   ego_density_feature     EgoNeRF.compute_densityfeature         models/EgoNeRF.py:291-347
 *                           EgoNeRF.compute_coarse_densityfeature  models/EgoNeRF.py:232-289 (coarse=1)
 *   ego_app_feature         EgoNeRF.compute_appfeature             models/EgoNeRF.py:349-413
 *   ego_feature2density     TensorBase.feature2density             models/tensorBase.py:415-419
 *   ego_raw2alpha           raw2alpha                              models/tensorBase.py:22-27
 *   ego_mlp_fea             MLPRender_Fea.forward                  models/tensorBase.py:54-78
 *   ego_sample_pdf_merge    sample_pdf + sort(cat(z, z_fine))      dataLoader/ray_utils.py:156-187, models/EgoNeRF.py:532-542
 *   ego_envmap_radiance     EnvironmentMap.get_radiance            models/envmap.py:6-14,26-34
 *   ego_avgpool_tables      EgoNeRF.update_coarse_sigma_grid       models/EgoNeRF.py:124-133
 *   ego_pack_mlp            (new) weight re-layout for the MFMA kernels; no reference counterpart
 *   ego_march_density       EgoNeRF.forward lines 507-529 / 544-553 (sampling -> sigma -> alpha,w)   models/EgoNeRF.py
 *   ego_shade               EgoNeRF.forward lines 555-556 / 571-572 (app feature -> renderModule)   models/EgoNeRF.py
 *   ego_composite           EgoNeRF.forward lines 579-598 (acc, rgb_map, envmap, clamp, depth)       models/EgoNeRF.py
 *   ego_render_forward      EgoNeRF.forward (whole call)                                          models/EgoNeRF.py:491-602
 *
 * Table layout in HBM ("channel-last"): a reference plane parameter (1,C,H,W) is stored as [H][W][C]
 * and a line parameter (1,C,L,1) as [L][C], fp32, so one bilinear tap is one contiguous C*4-byte read
 * (64 B for the 16 density components, 192 B for the 48 appearance components).  torch's
 * channels_last memory format of the reference-shaped tensor is exactly this layout.
 */
#ifndef EGONERF_HIP_H
#define EGONERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGO_ABI_VERSION 17

enum { EGO_PREC_F16X3 = 0, EGO_PREC_F32 = 1, EGO_PREC_F16F8 = 2, EGO_PREC_F16F6 = 3 };
/* ego_scene.head: the appearance head TensorBase.init_render_func selected (models/tensorBase.py:186-200) */
enum { EGO_HEAD_MLP_FEA = 0, EGO_HEAD_RGB = 1 };

enum {
  EGO_OK = 0,
  EGO_E_BADARG = -1,   /* null pointer / bad size / unsupported configuration */
  EGO_E_UNSUPPORTED = -2,
};

/* One VM-decomposed field of both grids.  grid index 0 = yin, 1 = yang; i = 0..2 follow the
 * reference's matMode [[0,1],[0,2],[1,2]] / vecMode [2,1,0] over axes (r, theta, phi):
 *   plane[g][0]: [N_theta][N_r][C]   line[g][0]: [N_phi][C]
 *   plane[g][1]: [N_phi][N_r][C]     line[g][1]: [N_theta][C]
 *   plane[g][2]: [N_phi][N_theta][C] line[g][2]: [N_r][C]                                      */
/* The 12 tables of one field must lie within 4 GB of each other (allocate them from one buffer, as the Python host layer
 * does): the forward gathers address every tap as lowest-table-address + 32-bit byte offset (scalar base + one VGPR per
 * address).  Entry points return EGO_E_BADARG for a field that does not. */
typedef struct ego_vm_field {
  const float* plane[2][3]; /* dev */
  const float* line[2][3];  /* dev */
  int32_t n_comp;           /* C: 16 (density) or 48 (appearance) in every shipped config */
  int32_t res[3];           /* (N_r, N_theta, N_phi) of THIS field (coarse field = full / 2) */
} ego_vm_field;

typedef struct ego_scene {
  /* coordinates (models/coordinates.py:76,500-505): all float32 values computed by the host exactly
   * as the reference computes them */
  float center[3];
  float ang_near[2]; /* theta, phi lower bounds: pi/4, -3pi/4 */
  float ang_inv[2];  /* 1/(far-near) per angle */
  const float* r_lut; /* dev [n_r_lut] = reference_r_grid, n_r_lut = N_r + 1 */
  int32_t n_r_lut;
  int32_t n_r;        /* N_r used in the final division (coordinates.py:156) */
  /* density activation (tensorBase.py:415-419) and compositing scale (opt.py:144) */
  int32_t act_softplus; /* 1 softplus, 0 relu */
  float density_shift;
  float distance_scale;
  /* tables */
  ego_vm_field density;
  ego_vm_field density_coarse;
  ego_vm_field app;
  const float* basis[2]; /* dev [app_dim][3*C_app] row-major = nn.Linear.weight, per grid */
  int32_t app_dim;       /* 27 */
  /* MLP_Fea (tensorBase.py:54-78): reference-layout weights (row-major [out][in]) ... */
  const float* mlp_w[3]; /* dev [128][150], [128][128], [3][128] */
  const float* mlp_b[3]; /* dev */
  int32_t mlp_in, mlp_hidden, view_pe, fea_pe;
  /* ... and the MFMA re-layout of the same weights produced by ego_pack_mlp (dev, EGO_PACKED_FLOATS) */
  const float* packed;
  /* environment map (models/envmap.py): emission [3][2h][h] or NULL */
  const float* envmap;
  int32_t envmap_h;
  /* arithmetic of the basis/MLP matrix products: EGO_PREC_F16X3 (0) = three fp16 MFMAs per product (hi*hi + lo*hi + hi*lo,
   * fp32 accumulate, ~2^-21 relative: fp32-grade); EGO_PREC_F32 = fp32-input MFMA; EGO_PREC_F16F8 = layers 1 and 2 of the MLP
   * with the main term in fp16 and both correction terms in one block-scaled fp8 (e4m3) MFMA per pair of k-steps (~2^-16
   * relative per product; composited max |d RGB| ~8e-6, i.e. > 10x inside the 1e-4 bar; inference only, the training forward
   * keeps the fp16 split); EGO_PREC_F16F6 = the same split with the correction terms on the fp6 (e2m3) path of the block-scaled
   * MFMA, two instructions per group of four k-steps, one power-of-two block scale per lane and 32 K values - static for the
   * weights, taken from the largest activation of the block at run time (so no fixed range to saturate); per-sample error 1.1x
   * F16F8's, shade kernel 3-4 % faster */
  int32_t mlp_precision;
  /* Opt-in skipping (EgoNeRF.forward itself evaluates every sample; these follow TensorBase.forward's mask semantics,
   * models/tensorBase.py:464-478, and YinYangAlphaGridMask, models/EgoNeRF.py:11-24):
   * occ: two {0,1} byte volumes [grid][N_phi][N_theta][N_r] (occ_res = N_r, N_theta, N_phi) or NULL.  A sample whose
   * trilinear mask value (align_corners, zero padding) is <= 0 gets sigma = 0.
   * term_eps > 0: samples whose incoming transmittance T is below term_eps get weight 0 (|d rgb| <= term_eps). */
  const uint8_t* occ;
  int32_t occ_res[3];
  float term_eps;
  /* Optional half-precision copy of the appearance tables (inference only): same shapes/layout as `app`, elements are
   * IEEE binary16 (the pointer type is nominal).  app_f16 != 0 makes ego_shade / ego_app_feature gather from it
   * (halves the bytes through the vector L1; interpolation and everything after stay fp32).  Measured effect on RGB in
   * DESIGN.md. */
  ego_vm_field app16;
  int32_t app_f16;
  /* Radius LUT of the fine pass after resampling, when it differs from r_lut: with the plain exponential grid
   * (interval_th = False, coordinates.py:132-155) the first pass normalises r on the `downsample=2` grid (EgoNeRF.py:524) and
   * the fine pass on the full one (EgoNeRF.py:546).  NULL / 0: the fine pass uses r_lut. */
  int32_t n_r_lut_fine;
  const float* r_lut_fine;
  int32_t n_r_fine;
  /* Appearance skip of TensorBase.forward (models/tensorBase.py:482-487, `rayMarch_weight_thres`): < 0 = off (every sample is
   * shaded); >= 0: a sample whose weight is <= weight_thres contributes colour 0 — its weight still counts in acc / depth — and
   * 32-sample tiles without any sample above the threshold are not shaded.  weight_thres = 0 is EXACT (EgoNeRF.forward adds
   * w * rgb = 0 for those samples, models/EgoNeRF.py:583) and is what the host layer sets by default. */
  float weight_thres;
  /* Optional accelerator for `occ` (NULL = off): [grid][N_phi - 1][N_theta - 1][N_r - 1] bytes, 1 iff any of the cell's eight corner
   * voxels is set.  The march then decides "mask value > 0" for a sample strictly inside a cell from this one byte (identical result:
   * all eight trilinear weights are positive there) and keeps the eight-tap evaluation for samples on lattice planes / outside. */
  const uint8_t* occ_cell;
  /* Appearance head (models/tensorBase.py:186-200).  EGO_HEAD_MLP_FEA (0): MLPRender_Fea (tensorBase.py:54-78); shadingMode 'MLP'
   * (MLPRender, tensorBase.py:107-129) is the same network with fea_pe = 0, i.e. mlp_in = 3 + 6 view_pe + app_dim.  EGO_HEAD_RGB (1):
   * RGBRender (tensorBase.py:37-39): the per-sample colour is the 3-channel appearance feature itself (no MLP, no sigmoid; app_dim must be
   * 3, mlp_in = mlp_hidden = 0, mlp_w / mlp_b unused).  'MLP_PE' cannot run in EgoNeRF.forward (it is handed 7-column coordinates for a
   * 3-column encoding), 'SH' crashes there too (its stage op is ego_sh_render). */
  int32_t head;
} ego_scene;

/* number of floats ego_pack_mlp writes: the packed weight blob used by ego_shade / ego_mlp_fea / ego_app_feature
 * (fp32 fragment layout, the fp16-split layout, then the basis fragments in the K order of the fp16-table gather) */
int64_t ego_packed_floats(void);
/* the same for the shape of `sc` (app_dim, app.n_comp, mlp_in, mlp_hidden, view_pe, fea_pe): ego_packed_floats() for the tuned shape
 * every shipped config resolves to (27 / 48 / 150 / 128 / 2 / 2), the fp32 layout of the any-shape compatibility kernels otherwise
 * (opt.py:87-100 lets a user choose n_lamb_sh, data_dim_color, featureC, view_pe, fea_pe; supported: n_comp a multiple of 4 up to
 * 48, app_dim <= 32, featureC 64 or 128, view_pe / fea_pe <= 8; density n_comp a multiple of 4 up to 48).  Those kernels compute
 * in plain fp32 and are about an order of magnitude slower than the tuned path; they train through ego_shade_train_generic /
 * ego_shade_backward_generic / ego_scatter_generic (row-major buffers; ego_shade's `dump` argument is for the tuned shape only). */
int64_t ego_packed_floats_scene(const ego_scene* sc);

int ego_abi_version(void);
const char* ego_last_error(void);
/* sizeof the ABI structs as compiled (0 ego_scene, 1 ego_render_args, 2 ego_vm_field, 3 ego_adam_tensor, 4 ego_shade_dump): lets a foreign-
 * language binding verify its struct mirrors at load time */
int64_t ego_sizeof(int32_t which);

/* Run-time self-test of the team-gather kernel family (no reference counterpart).  Builds of these kernels that contain
 * packed fp32 instructions broadcasting the high dword of a register pair returned different bits call after call on MI355X; the build
 * fences that off and this entry point checks the SHIPPED code on the device at hand: ego_app_feature and ego_shade (f16x3, f16f8, f16f6) run
 * `reps` times on a synthetic scene laid out in `workspace` (dev, ego_selftest_workspace_bytes() bytes, 256-byte aligned) and every
 * result is bit-compared with the first.  Synchronises `stream`.  Returns EGO_OK and *mismatching_calls = 0, or EGO_E_UNSUPPORTED with
 * the number of calls whose bits differed (host pointer).  The host layer calls it once per process and device (EGO_SKIP_SELFTEST=1 opts out). */
int64_t ego_selftest_workspace_bytes(void);
int ego_selftest(void* workspace, int64_t workspace_bytes, int32_t reps, int32_t* mismatching_calls, void* stream);

/* ---- separately callable stages (back the reference's public methods; parity-tested one by one) ---- */

/* rays [N][6] (o,d); r_sched [S] = radial offsets (host-built, float32); jitter [N][S] in [0,1) or NULL.
 * z = near + r (+ step*jitter); xyz = o + d*z.  Outputs xyz [N][S][3], z [N][S] (either may be NULL). */
int ego_sample_ray_exp(const float* rays, const float* r_sched, const float* jitter, float near_, int64_t N, int32_t S,
                       float* xyz, float* z, void* stream);

/* Equirectangular rays of rows [row0, row0+n_rows) of an H x W panorama for the camera-to-world pose c2w (HOST pointer,
 * 3x4 row-major): get_ray_directions_360 + get_rays (dataLoader/ray_utils.py:24-40, :85-113).  normalize != 0 divides the
 * camera-space direction by its norm first, as both ERP datasets do before get_rays (dataset_egocentric_video.py:57-58,
 * dataset_omniblender.py:42-43).  rays [n_rows*W][6] dev. */
int ego_erp_rays(int32_t H, int32_t W, int32_t row0, int32_t n_rows, const float* c2w, int32_t normalize, float* rays,
                 void* stream);

int ego_from_cartesian(const ego_scene* sc, const float* xyz, int64_t M, float* c7, void* stream);
int ego_normalize_coord(const ego_scene* sc, const float* c7, int64_t M, float* c7n, void* stream);

/* c7n [M][7] normalised yin-yang coordinates; out [M].  coarse != 0 uses sc->density_coarse. */
int ego_density_feature(const ego_scene* sc, const float* c7n, int64_t M, int32_t coarse, float* out, void* stream);
/* out [M][app_dim] */
int ego_app_feature(const ego_scene* sc, const float* c7n, int64_t M, float* out, void* stream);
int ego_feature2density(const ego_scene* sc, const float* feat, int64_t M, float* sigma, void* stream);
/* sigma, dist [N][S] (dist already multiplied by distance_scale, like the reference call site);
 * alpha, weight [N][S]; bg_weight [N].  tensorBase.py:22-27.  alpha = 1 - exp(-sigma * dist) is evaluated as -expm1(-sigma * dist),
 * the correctly rounded value of that expression (the literal float32 form loses ~6e-8 absolute in the subtraction; the fused
 * march does the same) */
int ego_raw2alpha(const float* sigma, const float* dist, int64_t N, int32_t S, float* alpha, float* weight,
                  float* bg_weight, void* stream);
/* viewdirs [M][3], feat [M][app_dim] -> rgb [M][3] */
int ego_mlp_fea(const ego_scene* sc, const float* viewdirs, const float* feat, int64_t M, float* rgb, void* stream);
/* models/tensorBase.py:30-34 SHRender with models/sh.py:87-112 (degree-2 real spherical harmonics):
 * rgb[m][c] = relu(sum_k Y_k(viewdirs[m]) * features[m][9c + k] + 0.5), features [M][27]. */
int ego_sh_render(const float* viewdirs, const float* features, int64_t M, float* rgb, void* stream);
/* z [N][Sc] coarse distances, weight [N][Sc] coarse weights (bins = midpoints of z, pdf = weight[1:-1]),
 * u [N][n_fine] or NULL (= linspace(0,1,n_fine), eval mode).  use_coarse != 0: z_out [N][Sc+n_fine] =
 * sort(cat(z, z_new)); else z_out [N][n_fine] = sort(z_new).  z_new_out [N][n_fine] optional (unsorted). */
int ego_sample_pdf_merge(const float* z, const float* weight, const float* u, int64_t N, int32_t Sc, int32_t n_fine,
                         int32_t use_coarse, float* z_out, float* z_new_out, void* stream);
int ego_envmap_radiance(const ego_scene* sc, const float* dirs, int64_t N, float* out, void* stream);
/* 2x average pooling of one channel-last plane [H][W][C] -> [H/2][W/2][C] (W==1: line [H][C] -> [H/2][C]) */
int ego_avgpool_table(const float* src, int32_t H, int32_t W, int32_t C, float* dst, void* stream);
/* the same for all 12 tables of a field in one launch (EgoNeRF.py:124-133 update_coarse_sigma_grid, called after every training
 * step when resampling): dst->res must be src->res / 2 per axis, dst's tables allocated by the caller */
int ego_avgpool_field(const ego_vm_field* src, const ego_vm_field* dst, void* stream);
/* 8-tap occupancy lookup of YinYangAlphaGridMask.sample_alpha (models/EgoNeRF.py:19-24): c7n [M][7] -> out [M] (the
 * trilinear mask value; > 0 means occupied). */
int ego_alpha_mask_sample(const ego_scene* sc, const float* c7n, int64_t M, float* out, void* stream);
/* reference-layout basis/MLP weights in `sc` -> packed blob (dev, ego_packed_floats_scene(sc) floats) */
int ego_pack_mlp(const ego_scene* sc, float* packed_out, void* stream);
/* the same with a choice of what is written: for_training != 0 writes only the regions the differentiable path reads (the fp32 and
 * fp16-split fragment layouts and the basis fragments) and leaves the f16f8 / f16f6 inference images of the blob untouched - a training
 * step re-packs after every optimiser step and never reads them (a memset and three kernels saved per step).  A scene packed this way
 * must be shaded with EGO_PREC_F16X3 or EGO_PREC_F32.  ego_pack_mlp(sc, out, st) == ego_pack_mlp_for(sc, out, 0, st). */
int ego_pack_mlp_for(const ego_scene* sc, float* packed_out, int32_t for_training, void* stream);
/* The fp32 compatibility layout (what ego_packed_floats_scene / ego_pack_mlp produce for a non-tuned shape) for ANY supported shape,
 * the tuned one included: a scene whose `packed` points at this blob can be trained through ego_shade_train_generic /
 * ego_shade_backward_generic / ego_scatter_generic, i.e. with fp32 activations, fp32 dumps and fp32 weight-gradient operands like the
 * reference's autograd (train.py:312-314) - the parity mode beside the tuned training path, whose x / h1 / h2 dumps and dh1 / dh2 are
 * halves (weight gradients carry ~2^-12 relative error per operand, activations above 65504 are not representable). */
int64_t ego_packed_floats_compat(const ego_scene* sc);
int ego_pack_mlp_compat(const ego_scene* sc, float* packed_out, void* stream);

/* ---- the fused hot path ------------------------------------------------------------------------ */

/* Sampling -> yin-yang coords -> density lookup -> sigma -> alpha, transmittance scan.
 * z_in [N][S] explicit sample distances, or NULL: z = near + r_sched[s] (+ jitter).  coarse bit 0 selects the
 * pooled tables, bit 1 the fine-pass radius LUT (ego_scene.r_lut_fine, if set).  Outputs (any may be NULL): z_out [N][S], alpha [N][alpha_stride] (alpha_stride 0 = S;
 * columns S.. are filled with 1, the reference's trailing ones column when an envmap is present,
 * EgoNeRF.py:587), weight [N][S], bg_weight [N], coords_out [N][S][4] = normalised (r, theta, phi) of the sample's
 * grid + is_yang flag (what ego_shade needs; saves it the acos/atan2/LUT search), sigma_out [N][S] (kept for the
 * backward pass), tile_active [ceil(N*S/32)] bytes: 1 for every 32-sample tile that holds a weight above the shading
 * threshold, 0 otherwise (lets ego_shade skip tiles that are fully masked / terminated).  When S is a multiple of 32 every
 * flag is written; otherwise tiles straddle rays, flags are only ever set to 1 and the caller must zero them first. */
int ego_march_density(const ego_scene* sc, const float* rays, int64_t N, int32_t S, const float* z_in,
                      const float* r_sched, const float* jitter, float near_, int32_t coarse, float* z_out,
                      float* alpha, int32_t alpha_stride, float* weight, float* bg_weight, float* coords_out, float* sigma_out,
                      uint8_t* tile_active, void* stream);

/* Per-sample activations the training forward keeps for the backward pass (all dev).  Logical matrices x [M][160] (layer-1
 * inputs), h1, h2 [M][128] (post-ReLU hidden activations), v [M][144] (plane x line products); element kk of lane half h
 * is logical column 8 (kk / 4) + 4 h + kk % 4 (ego_train_layout() maps columns to the reference's inputs / units).  Storage
 * is tile-blocked, lane-major, whole tiles (each buffer needs ceil(M / 32) * 32 rows):
 *   v: fp32 [tile = m / 32][quad pair q = column / 8][lane = 32 h + m % 32][4 floats] (float offset tile * 32 * width + q * 256 +
 *      (32 h + m % 32) * 4 + c);
 *   x, h1, h2: IEEE halves, rounded to nearest, in the MFMA operand order of the kernels themselves: [tile][k-step s][lane][8], element
 *      e = logical column 8 (2 s + e / 4) + 4 h + e % 4 - they only feed the weight-gradient sums (sum over samples of dh * x: the
 *      2^-12 rounding is unbiased and averages out; pinned by the gradient goldens), at half the bytes of fp32 (round 4: the training
 *      step is bound by its dump traffic).  What the DATA gradients need of the forward stays exact: the ReLU masks (relu_bits) and
 *      the feature slots in fp32 (fe).
 * ego_weight_grad reads these layouts directly (a_layout / b_layout).  relu_bits [ceil(M / 32)][2 (h1, h2)][64 lanes][2] uint32 holds the ReLU masks of
 * the two hidden layers (bit 16 (t & 1) + r of word t >> 1 of lane 32 h + m % 32 <=> unit 32 t + slot_row(r, h) of sample m is
 * > 0): all the shade backward needs of h1 / h2 (autograd of torch.nn.ReLU, tensorBase.py:68-71), 32 B instead of 1 KB per sample. */
typedef struct ego_shade_dump {
  uint16_t* x;   /* optional (NULL: not written): only d(W1) reads it, and ego_weight_grad_x re-derives it from `fe` and the rays */
  uint16_t* h1;
  uint16_t* h2;
  float* v;      /* optional since v15 (NULL: not written): only d(basis) reads it, and ego_scatter_app_sorted(dfe, gbasis) takes that product along */
  uint32_t* relu_bits;
  float* fe;   /* [ceil(M / 32)][4][64 lanes][4] fp32: the 16 feature slots of lane 32 h + m % 32 (basis output, slot r = feature 2 r + h);
                * the backward re-derives the encodings' sines and cosines from them with the forward's own instructions */
} ego_shade_dump;

/* Static facts about the shade kernel that implements `precision` (EGO_PREC_*), for roofline accounting by a caller that times it
 * (bench.py): out[0] samples per tile (one wave-level unit of work), [1] fp16 / fp32-input MFMA instructions per tile of samples
 * that lie in one grid, [2] block-scaled fp8 / fp6 MFMA instructions per tile, [3] flop per instruction of [1], [4] flop per
 * instruction of [2], [5] f32 -> fp8 conversion instructions (F16F6: 32-value fp6 conversions) per lane and tile, [6] algorithmic flop per sample (basis +
 * MLP_Fea of tensorBase.py:54-78: 2 (144 x 27 + 150 x 128 + 128 x 128 + 128 x 3)), [7] gathered appearance tap bytes per
 * sample (EgoNeRF.py:349-413: 3 x (4 + 2) taps x 48 channels x 4 B).  n must be 8.  The counts are computed from the same
 * constants the kernel's loops run over. */
int ego_shade_kernel_info(int32_t precision, int32_t* out, int32_t n);

/* Appearance lookup -> basis -> positional encoding -> MLP for every sample: rgb [N][S][3].
 * z [N][S] sample distances (from ego_march_density); coords [N][S][4] optional (ego_march_density's coords_out),
 * NULL = recompute the yin-yang coordinates from rays and z. */
int ego_shade(const ego_scene* sc, const float* rays, const float* z, const float* coords, int64_t N, int32_t S, float* rgb,
              const ego_shade_dump* dump /* NULL for inference */, const uint8_t* tile_active /* NULL = shade every tile */,
              void* stream);

/* ego_shade + ego_composite as ONE launch (rows F, G, H, J; EgoNeRF.py:555-598): the shade kernel's waves own whole rays and finish the
 * pixel in their epilogue, no per-sample colours are written.  For the tuned model shape with fp32 tables, a split-precision arithmetic
 * (not EGO_PREC_F32), weight_thres <= 0 and S a multiple of 32 (EGO_E_UNSUPPORTED otherwise: call ego_shade + ego_composite); coords
 * is required.  Sums run in a different order than ego_composite's (per lane over the ray's tiles, then across 32 lanes): equal to it
 * within fp32 rounding of the sums, not bit for bit.  3.5 % faster than ego_shade alone since round 5 (the folded kernel has the registers
 * to load a plane's basis fragments ahead of the next plane's taps), so ego_render_forward takes it by default where it applies and
 * where its ray-granular deal of the work is balanced (ego_render_forward_folds below); it also saves the [N][S][3] colour buffer. */
int ego_shade_composite(const ego_scene* sc, const float* rays, const float* z, const float* coords, const float* weight,
                        const float* bg_weight /* NULL without an envmap */, int64_t N, int32_t S,
                        const uint8_t* tile_active /* NULL = shade every tile */, float* rgb_map, float* depth /* may be NULL */,
                        float* bg_map /* may be NULL */, float* env_map /* may be NULL */, void* stream);

/* 1 iff ego_render_forward shades and composites N rays x S samples of this scene in ONE launch (ego_shade_composite): the scene
 * qualifies (above) and whole rays divide evenly enough over the kernel's waves (critical path within 3 % of the tile-granular
 * ego_shade's: 4096 x 512 and 16384 x 256 fold, 333 or 4097 rays do not).  Environment: EGO_RENDER_FOLD=0 never folds, =1 folds
 * whenever the scene qualifies.  No device work. */
int32_t ego_render_forward_folds(const ego_scene* sc, int64_t N, int32_t S);

/* acc, rgb_map (+ envmap background), clamp, depth (+ (1-acc)*d_z quirk, EgoNeRF.py:598).
 * Outputs rgb_map [N][3], depth [N]; bg_map/env_map [N][3] written only when sc->envmap != NULL (may be NULL);
 * rgb_raw [N][3] optional = rgb_map before the clamp (the backward pass needs the clamp mask). */
int ego_composite(const ego_scene* sc, const float* rays, const float* z, const float* weight, const float* bg_weight,
                  const float* rgb, int64_t N, int32_t S, float* rgb_map, float* depth, float* bg_map, float* env_map,
                  float* rgb_raw, void* stream);

/* ---- training step: backward of the path (autograd of EgoNeRF.forward, train.py:312-314) ---------
 * Table gradients are accumulated with float atomics straight into channel-last gradient tables (same layout as the
 * parameters; the caller zeroes them).  Weight gradients of the basis / MLP are left to the caller as plain GEMMs over
 * the per-sample buffers these kernels write (lane order; ego_train_layout gives the column maps). */
typedef struct ego_vm_grad {
  float* plane[2][3]; /* dev, [H][W][C] like the parameter */
  float* line[2][3];
} ego_vm_grad;

int64_t ego_train_packed_floats(void);
/* transposed fp16-split fragments of W2, W1, the basis matrices and W3 for the data-gradient chain */
int ego_pack_train(const ego_scene* sc, float* out, void* stream);
/* which = 0: x column [160] -> MLP input column | 1: hidden column [128] -> unit | 2: dfe column [32] -> feature |
 * 3: v column [144] -> basis input column; -1 marks padding.  Host memory. */
int ego_train_layout(int32_t which, int32_t* out, int32_t n);
/* compositing backward (autograd of EgoNeRF.py:579-593 and tensorBase.py:22-27,415-419).
 * g_rgb [N][3] = dL/d rgb_map, rgb_raw = rgb_map before the clamp, env_map [N][3] or NULL.  alpha has row stride
 * alpha_stride (S, or S+1 when the envmap's ones column is appended); g_alpha = dL/d alpha with the same stride or NULL
 * (train.py:306-309 ray_entropy_loss).  depth_map carries no gradient in the reference (computed under no_grad,
 * EgoNeRF.py:595-598).  Writes dc [N][S][3] = dL/d rgb_sample and dfeat [N][S] = dL/d(density feature) (the per-plane relu
 * mask of EgoNeRF.py:340,346 is applied by ego_scatter_density). */
int ego_march_backward(const ego_scene* sc, const float* z, const float* alpha, int32_t alpha_stride, const float* weight,
                       const float* sigma, const float* bg_weight, const float* rgb, const float* g_rgb, const float* g_alpha,
                       const float* rgb_raw, const float* env_map, int64_t N, int32_t S, float* dc, float* dfeat, void* stream);
/* shade backward: dc [N][S][3] in = dL/d rgb_sample, out = dL/d(pre-sigmoid).  Writes dh2, dh1 (logical [M][128]), dfe [M][32]
 * (the feature-slot gradients of the sample's OWN grid; ego_weight_grad's a_layout 3 routes them by coords.w) and dv = dL/d(plane x line products).  dh2 / dh1 are stored as SCALED fp16 in the order the kernel's own
 * matrix products consume them: [tile = m / 32][k-step s = 0..7][lane = 32 h + m % 32][8 halves], element e = logical column
 * 8 (2 s + e / 4) + 4 h + e % 4 (the column numbering of the forward's dumps), value = half * dh_scale[row][m] with dh_scale
 * [2][ceil(M / 32) * 32] per-sample powers of two (row 0: dh2, row 1: dh1) chosen so that a sample's largest magnitude lies in
 * [2^12, 2^13): 11 significant bits, rounded to nearest, half the bytes of fp32 (ego_weight_grad's a_layout 2 reads it).  dv is
 * [tile][plane * 3 + line][sample][16 channels] fp32 (it feeds single table texels, where fp16 rounding would show); all three
 * need ceil(M / 32) * 32 rows.  Reads fwd->fe and fwd->relu_bits only. */
int ego_shade_backward(const ego_scene* sc, const float* train_packed, const float* coords, float* dc, const float* rgb,
                       const ego_shade_dump* fwd, uint16_t* dh2, uint16_t* dh1, float* dh_scale, float* dfe, float* dv, float* dv_absmax,
                       int64_t N, int32_t S, void* stream);
/* dv_absmax (v15; dev, one float, may be NULL): receives max |dv| over the valid samples (NaN if a sample's gradient is not finite) - the
 * bound ego_scatter_app_sorted derives the fixed-point unit of its line sums from, so that it need not read dv once more to find it. */
/* backward of the VM lookups (autograd of F.grid_sample in EgoNeRF.py:291-347 / :349-413): accumulates into the gradient
 * tables (same channel-last layout as the parameters).  coords [N][S][4] = the forward's normalised (r, theta, phi, grid). */
int ego_scatter_density(const ego_scene* sc, const ego_vm_grad* gdensity, const float* coords, const float* dfeat, int64_t N, int32_t S,
                        void* stream);
int ego_scatter_app(const ego_scene* sc, const ego_vm_grad* gapp, const float* coords, const float* dv, int64_t N, int32_t S,
                    void* stream);

/* ---- the same table gradients without atomics: bit-reproducible (csrc/ego_scatter_sorted.hip) ----
 * ego_scatter_density / ego_scatter_app add with float atomics: their sums depend on the order the hardware serves them.  The three
 * entry points below bin the step's samples by texel cell once (three stable radix sorts of the forward's coordinates; both fields share
 * them) and then write every gradient texel exactly once from sums taken in a fixed order: two calls return the same bits, no zero fill
 * of `grad` is needed, and there is no atomic traffic (they are also faster: DESIGN.md 4.2).  Same mathematics as F.grid_sample's
 * backward in compute_densityfeature / compute_appfeature (models/EgoNeRF.py:291-347, :349-413) under train.py:312-314; same argument
 * meaning as ego_scatter_density / ego_scatter_app.  `workspace` (dev, 256-byte aligned, ego_scatter_sorted_workspace_bytes(sc, N, S)
 * bytes; -1 = bad arguments) carries the sort from ego_scatter_sort to the two scatters and holds their scratch: calls that share a
 * workspace must be ordered on one stream (or by events).  The density and appearance fields must have one resolution. */
int64_t ego_scatter_sorted_workspace_bytes(const ego_scene* sc, int64_t N, int32_t S);
int ego_scatter_sort(const ego_scene* sc, const float* coords, int64_t N, int32_t S, void* workspace, int64_t workspace_bytes, void* stream);
int ego_scatter_density_sorted(const ego_scene* sc, const ego_vm_grad* gdensity, const float* coords, const float* dfeat, int64_t N, int32_t S,
                               void* workspace, int64_t workspace_bytes, void* stream);
int ego_scatter_app_sorted(const ego_scene* sc, const ego_vm_grad* gapp, const float* coords, const float* dv, const float* dv_absmax,
                           const float* dfe, float* gbasis, int32_t ldg, int64_t N, int32_t S, void* workspace, int64_t workspace_bytes,
                           void* stream); /* dv: ego_shade_backward's blocked layout */
/* dfe / gbasis (v15; both or neither; NULL = as before): the basis gradient rides along as well.  d(basis_mat)[slot][plane x 48 + channel] =
 * sum over samples of dfe[s][slot] x (plane value x line value)[s][channel] - and the walk has both factors of that product in registers,
 * so the forward need not dump v (ego_shade_dump.v = NULL: 576 B per sample less to write) for ego_weight_grad to read it back.  dfe =
 * ego_shade_backward's [M][32] (the sample's own grid's slots); gbasis [64][ldg >= 144] receives row 32 g + slot, column plane x 48 +
 * channel (reference channel order), written, not accumulated; bf16 hi / lo split MFMA (three terms, ~16 significand bits per operand), per-wave partial products added
 * in a fixed order (bit-reproducible).  Only in the walk form (EGO_E_UNSUPPORTED with EGO_SORTED_WALK=0). */
/* dv == NULL (v16; needs dfe, gbasis, dv_absmax and sc->basis): the walk re-derives dv = basis_g^T dfe for its own plane's 48 channels (the
 * 27 feature-slot gradients of a sample are 128 B where its dv is 576 B) in ego_shade_backward's own arithmetic for that product (operands scaled
 * by a power of two per sample / per channel, fp16 hi + lo, three MFMA terms, fp32 accumulation), so ego_shade_backward
 * need not write dv at all (pass it dv = NULL: its dv_absmax then carries max |dfe|, which is what this call expects in that case - the
 * fixed-point unit comes from max |dfe| x the largest column sum of |basis|).  A sample's dv differs from ego_shade_backward's by the
 * rounding of that arithmetic (~2^-21 per product; the fp32 summation order over the 27 slots differs).  The padding columns of dfe (slots
 * 14, 15 of either half and feature 27) meet zero weights here and must be finite, as ego_shade_backward writes them (zeros). */
/* v15: ONE pass over dfeat / dv.  The gradient of a plane and of the line it is multiplied with (the table of the axis that is not in the
 * plane) come out of the same walk over the plane's cells: a cell's samples share the four plane texels, so the line's contribution of a
 * sample is four multiply-adds away - but it lands in an arbitrary line texel.  Those sums are therefore taken in 64-bit FIXED POINT
 * (integer LDS atomics per workgroup, integer partial tables added at the end: integer addition is associative, any order returns the same
 * bits).  The unit is one power of two per table: 2^k with k = e(max |d|) + e(max |plane texel|) + 1 - nbits, nbits = min(50, 62 -
 * ceil(log2(N S))), i.e. >= 40 bits below the largest possible contribution at 2^21 samples (an fp32 sum keeps 24 bits below its running
 * value; a contribution more than ~2^-40 below the table's largest possible one is rounded to the unit).  max |plane texel| is taken by
 * the call itself; max |dfeat| too; dv_absmax (dev, one float, may be NULL = computed here by one more read of dv) is ego_shade_backward's.
 * Non-finite inputs give NaN line gradients.  Line tables too large for the LDS (axis > ~1 200 texels) and EGO_SORTED_LINES=separate take
 * the two-pass form of v14 (float partial sums of fixed 256-sample sub-blocks, added in sub-block order).  An empty batch (N = 0)
 * zero-fills the tables. */
/* d(envmap.emission) [3][2h][h] += backward of bg_weight * sigmoid(bilinear(emission, dir)) (envmap.py:26-34,
 * EgoNeRF.py:588-590).  dirs = N directions dir_stride floats apart (rays + 3 with stride 6, or a packed [N][3]);
 * env_map = the forward's radiance [N][3]; the clamp mask is taken from rgb_raw (pass values in [0,1] for none). */
int ego_envmap_backward(const ego_scene* sc, const float* dirs, int32_t dir_stride, const float* g_rgb, const float* rgb_raw, const float* bg_weight,
                        const float* env_map, int64_t N, float* g_emission, void* stream);
/* Weight gradients of nn.Linear layers over all samples: G [32 ceil(ca/32)][ldg] += A^T B for A (M rows, ca <= 128 columns
 * used of lda) and B (M rows, cb <= 160 of ldb), G accumulated (zero it first).  a_layout 0: row-major fp32 | 1: fp32 in the
 * shade kernels' dump layout [tile = m / 32][quad pair q][lane = 32 h + m % 32][4] with logical column 8 q + 4 h + c
 * (ceil(M / 32) * 32 rows allocated) | 2: ego_shade_backward's scaled-fp16 layout (ca = lda = 128, a_scale [M] = that matrix's
 * row of dh_scale) | 3: row-major fp32 [M][32] standing for 64 logical columns, row m filling columns [32 g, 32 g + 32) with
 * g = (a_scale[4 m + 3] != 0) (ego_shade_backward's dfe; ca = 64, lda = 32, a_scale = the forward's coords [M][4]); a_scale NULL for
 * layouts 0 / 1.  b_layout 0: row-major fp32 | 1: fp32 in the dump layout (ego_shade_dump.v) | 2: halves in the dump layout of
 * ego_shade_dump's x / h1 / h2 ([tile][k-step][lane][8]; cb a multiple of 16, ldb = cb).  ones_col >= 0 replaces that column
 * of B by ones (it may lie beyond cb, < 160), which yields the bias gradient = column sums of A in G[:, ones_col].  bf16 hi/lo
 * split MFMA, ~17 significand bits per operand. */
int ego_weight_grad(const void* A, int32_t lda, int32_t ca, int32_t a_layout, const float* a_scale, const void* B, int32_t ldb,
                    int32_t cb, int32_t b_layout, int32_t ones_col, int64_t M, float* G, int32_t ldg, void* stream);
/* The same product, bit-reproducible: every workgroup stores its partial product block to `partial` (dev, ego_weight_grad_partial_floats()
 * floats) and a second kernel adds the blocks in workgroup order and STORES the result (G needs no zero fill; rows [0, 32 ceil(ca/32)) x
 * columns [0, 32 ceil(cols/32)) of G are overwritten - EXCEPT on the d(W3) fast path (ca == 3, cb == 128, b_layout 2: the VALU kernel
 * k_wgrad3), which writes rows [0, 3) x columns [0, 160) only and leaves the padding rows 3..31 as they were: zero G once if those rows
 * are read).  ego_weight_grad adds with float atomics: its sums depend on the order the
 * hardware serves them (differences in the last bits from run to run).  partial = NULL is ego_weight_grad. */
int64_t ego_weight_grad_partial_floats(void);
/* d(W1) and d(b1) of the tuned head WITHOUT the x dump: G [128][ldg >= 160] = dh1^T [x | 1] where x - the MLP input of
 * tensorBase.py:68-75 in the shade kernels' column order - is re-derived per sample from the forward's feature-slot dump
 * (ego_shade_dump.fe) and the ray's view direction (rays [N][6], sample m belongs to ray m / S) with the forward's own instructions and
 * rounding, i.e. the operands are bit-identical to ego_weight_grad(dh1, .., a_layout 2, x dump, b_layout 2, ones_col) and
 * ego_shade_dump.x may be NULL in the training forward (320 B per sample less to write, 192 B less to read).  dh1 / dh_scale as
 * ego_shade_backward writes them; ones_col = the zero-padding column that yields the bias gradient; partial as in ego_weight_grad_det
 * (NULL = float atomics into a zeroed G). */
int ego_weight_grad_x(const void* dh1, const float* dh_scale, const float* fe, const float* rays, int32_t S, int32_t ones_col, int64_t M, float* G,
                      int32_t ldg, float* partial, int64_t partial_floats, void* stream);
int ego_weight_grad_det(const void* A, int32_t lda, int32_t ca, int32_t a_layout, const float* a_scale, const void* B, int32_t ldb, int32_t cb,
                        int32_t b_layout, int32_t ones_col, int64_t M, float* G, int32_t ldg, float* partial, int64_t partial_floats,
                        void* stream);

/* ---- training of model shapes other than the tuned one (opt.py:87-100 lets a user choose n_lamb_sigma / n_lamb_sh, data_dim_color,
 * featureC, view_pe, fea_pe; supported shapes as for ego_packed_floats_scene).  Plain fp32 compatibility kernels over ROW-MAJOR
 * per-sample buffers, about an order of magnitude slower than the tuned path; ego_march_density / ego_march_backward /
 * ego_composite / ego_weight_grad (a_layout 0, b_blocked 0, 160-column chunks) serve every shape.  Autograd of EgoNeRF.py:349-413 and
 * tensorBase.py:54-78, as train.py:312-314 runs it. ---- */
/* ego_shade for training: rgb [N][S][3] plus the activations the backward needs: x [M][ldx] = the MLP input row in the reference's
 * column order (tensorBase.py:68-75; columns >= mlp_in are not written: zero the buffer), h1 / h2 [M][ldh] = post-ReLU hidden
 * activations, v [M][ldv] = plane x line products (column = plane * n_comp + channel; 16-byte aligned, ldv % 4 == 0). */
int ego_shade_train_generic(const ego_scene* sc, const float* rays, const float* coords, int64_t N, int32_t S, float* rgb, float* x, int32_t ldx,
                            float* h1, float* h2, int32_t ldh, float* v, int32_t ldv, void* stream);
/* dc [M][3] in = dL/d rgb_sample, out = dL/d(pre-sigmoid).  Writes dh2, dh1 [M][mlp_hidden], dfe64 [M][64] (the feature gradients of
 * the sample's own grid g in columns [32 g, 32 g + 32), zeros in the other half: dfe64^T v gives both basis gradients in one
 * product) and dv [M][ldv] = dL/d(plane x line products).  ldv == 0 (48 appearance components only): dv in ego_shade_backward's blocked layout
 * ([ceil(M / 32)][9][32][16] floats), so that the tuned scatters (ego_scatter_app_sorted / ego_scatter_app) serve a model whose HEAD has
 * another shape but whose tables have the shipped one. */
int ego_shade_backward_generic(const ego_scene* sc, const float* coords, float* dc, const float* rgb, const float* x, int32_t ldx, const float* h1,
                               const float* h2, int32_t ldh, float* dh2, float* dh1, float* dfe64, float* dv, int32_t ldv, int64_t N, int32_t S,
                               void* stream);
/* backward of the VM lookups of `field` (any n_comp that is a multiple of 4 up to 48) into `grad` (channel-last like the parameters,
 * accumulated with float atomics).  ldd == 0: d = dfeat [M] from ego_march_backward (density: the per-plane relu mask of
 * EgoNeRF.py:340,346 is applied here); ldd > 0: d = dv [M][ldd] from ego_shade_backward_generic (appearance). */
int ego_scatter_generic(const ego_vm_field* field, const ego_vm_grad* grad, const float* coords, const float* d, int32_t ldd, int64_t N, int32_t S,
                        void* stream);

/* ---- training-step table ops (train.py:245-330).  Tables are channel-last [H][W][C].  `value` (device double, may be
 * NULL) and `grad` (device, same layout as the table, may be NULL) are ACCUMULATED into, so one buffer collects a whole
 * regulariser and gradients add onto the render's. ---- */
/* utils.py:155-171 TVLoss on one plane: value += scale * 2 (sum dH^2 / (C (H-1) W) + sum dW^2 / (C H (W-1)));
 * EgoNeRF.py:214-228 call it with scale = 1e-2 per plane. */
int ego_tv_plane(const float* table, int32_t C, int32_t H, int32_t W, float scale, double* value, float* grad, void* stream);
/* EgoNeRF.py:206-212 density_L1 term of one table: value += scale * mean |x|. */
int ego_l1_table(const float* table, int64_t n, float scale, double* value, float* grad, void* stream);
/* EgoNeRF.py:189-201 vectorDiffs term of one line table [n][C]: value += scale * mean |off-diagonal of L^T L|. */
int ego_line_ortho(const float* line, int32_t C, int32_t n, float scale, double* value, float* grad, void* stream);
/* utils.py:175-183 ray_entropy_loss over alpha [N][S] (row stride `stride`): value += mean_ray H(alpha / (sum alpha + 1e-10)),
 * g_alpha (same stride, WRITTEN) = d value / d alpha. */
int ego_ray_entropy(const float* alpha, int64_t N, int32_t S, int32_t stride, double* value, float* g_alpha, void* stream);
/* coordinates.py:27-39 and :226-266 (up_sampling_VM): bilinear, align_corners=True, zero-padded resample of a table at
 * per-axis normalised positions xs [W2], ys [H2] (device) -> dst [H2][W2][C]. */
int ego_resample_table(const float* src, int32_t C, int32_t H, int32_t W, const float* xs, const float* ys, int32_t H2, int32_t W2,
                       float* dst, void* stream);
/* torch.optim.Adam step (train.py:182,311-313; no weight decay / amsgrad) over `count` tensors (host array of device
 * pointers), each with its own lr (train.py:328-329 decays lr per group); `step` counts from 1. */
typedef struct ego_adam_tensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t n;
  float lr;
  int32_t reserved;
} ego_adam_tensor;
int ego_adam_step(const ego_adam_tensor* tensors, int32_t count, float beta1, float beta2, float eps, int32_t step, void* stream);
/* The same update with the step count and the learning-rate schedule ON THE DEVICE, so that a hipGraph capture of the whole
 * training step (forward + backward + this + ego_avgpool_field) can be replayed: clock = device double[4] {t, lr scale, scratch,
 * scratch}, initialised by the caller to {0, 1, 0, 0}.  Every call does t += 1, uses lr_i * scale / (1 - beta1^t) and
 * sqrt(1 - beta2^t), then scale *= lr_factor (train.py:328-329 multiplies every group's lr by lr_factor after each step). */
int ego_adam_step_graph(const ego_adam_tensor* tensors, int32_t count, float beta1, float beta2, float eps, double lr_factor,
                        double* clock, void* stream);

/* utils.py:104-152 rgb_ssim (renderer.py:160) on device images [H][W][3]: 'valid' separable Gaussian window in float64 over
 * float32 inputs.  sum (device double, may be NULL) accumulates the sum of the SSIM map — the caller divides by
 * (H-fs+1)(W-fs+1)3; ssim_map [(H-fs+1)][(W-fs+1)][3] may be NULL. */
int ego_rgb_ssim(const float* img0, const float* img1, int32_t H, int32_t W, double max_val, int32_t filter_size, double filter_sigma,
                 double k1, double k2, double* sum, float* ssim_map, void* stream);

/* v17: up to EGO_COPY_OUT_MAX float arrays device -> MAPPED host memory (hipHostMalloc / torch pinned memory: the device writes it through the
 * same pointer) in ONE launch of `workgroups` x 256 threads.  For renderer.py:39-53's `.cpu().numpy()` of every chunk's outputs: the runtime's
 * own device -> host copy is a kernel that fills the chip - next to it the following chunk's march took 260 us instead of 97, its shade 570
 * instead of 450 (tools/handover_timeline.sh) - while 8 MB over the host link need no more than a few dozen workgroups.  Ordered on `stream`
 * like any kernel; the host may read `dst` once the stream (or an event behind the call) has completed. */
#define EGO_COPY_OUT_MAX 8
int ego_copy_out(int32_t count, const float* const* src, float* const* dst, const int64_t* n_floats, int32_t workgroups, void* stream);

typedef struct ego_render_args {
  int32_t n_coarse, n_fine;
  int32_t resampling, use_coarse_sample;
  const float* r_sched; /* dev [n_coarse] */
  const float* jitter;  /* dev [N][n_coarse] or NULL (eval) */
  const float* u;       /* dev [N][n_fine] or NULL (eval: linspace) */
  float near_;
  int32_t reserved;
  const float* z_coarse; /* dev [N][n_coarse] explicit distances of the first pass, or NULL.  Overrides r_sched / jitter: the
                          * exp_sampling=False path, TensorBase.sample_ray (tensorBase.py:308-327), whose per-ray schedule the
                          * host computes */
  void* marched;         /* v17: hipEvent_t or NULL.  Recorded on `stream` behind the LAST march launch, ahead of the shade: lets a caller
                          * start copies of the PREVIOUS call's outputs under this call's shade kernel (bound by instruction issue) instead
                          * of under its march (bound by the latency of its gathers: next to the runtime's copy kernel it took 260 us
                          * instead of 97, tools/handover_timeline.sh) */
} ego_render_args;

/* Whole EgoNeRF.forward for N rays.  S_out = n_coarse (no resampling) | n_coarse+n_fine | n_fine.
 * workspace: dev scratch of ego_render_workspace_bytes(N, args) bytes.
 * Outputs: rgb_map [N][3], depth [N], alpha [N][S_out (+1 with envmap)], bg_map/env_map [N][3] (envmap only). */
int64_t ego_render_workspace_bytes(int64_t N, const ego_render_args* args);
int ego_render_forward(const ego_scene* sc, const ego_render_args* args, const float* rays, int64_t N, void* workspace,
                       float* rgb_map, float* depth, float* alpha, float* bg_map, float* env_map, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EGONERF_HIP_H */
