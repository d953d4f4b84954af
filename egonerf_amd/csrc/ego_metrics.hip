// Evaluation metric next to the render (renderer.py:153-163): rgb_ssim (utils.py:104-152) on device images.
#include "ego_device.h"
#include "ego_host.h"

namespace {

constexpr int SSIM_MAX_FS = 15;
constexpr int SSIM_TILE = 16;

struct SsimArgs {
  const float* img0;
  const float* img1;  // [H][W][3]
  int H, W, fs;
  double filt[SSIM_MAX_FS];
  double c1, c2;
  double* sum;      // accumulates the sum of the map
  float* map;       // [(H-fs+1)][(W-fs+1)][3] or null
};

// one 16x16-pixel output tile per workgroup; both input tiles staged in LDS.  The Gaussian window is separable and the reference applies
// it that way (rows, then columns, in float64 over float32 images and float32 squares / products): per channel the five row-filtered
// moment images of the tile go to LDS once ((16 + fs - 1) x 16 values each) and every output takes its column sums from there - the same
// additions in the same order as filtering each output's own fs x fs window (the round-3 form: 5 x 121 + 55 float64 multiply-adds per
// output instead of 5 x (11 x 26 / 16 + 11)), so the map is bit-identical to it.
__global__ __launch_bounds__(256) void k_rgb_ssim(SsimArgs A) {
  constexpr int TIN = SSIM_TILE + SSIM_MAX_FS - 1;
  __shared__ float t0[TIN * TIN * 3], t1[TIN * TIN * 3];
  __shared__ double hrow[5][TIN][SSIM_TILE];
  __shared__ double red[256];
  const int Ho = A.H - A.fs + 1, Wo = A.W - A.fs + 1;
  const int y0 = blockIdx.y * SSIM_TILE, x0 = blockIdx.x * SSIM_TILE;
  const int tin = SSIM_TILE + A.fs - 1;
  for (int e = threadIdx.x; e < tin * tin * 3; e += 256) {
    const int c = e % 3, px = (e / 3) % tin, py = e / (3 * tin);
    const int yy = y0 + py, xx = x0 + px;
    const bool in = yy < A.H && xx < A.W;
    const int64_t o = ((int64_t)yy * A.W + xx) * 3 + c;
    t0[e] = in ? A.img0[o] : 0.f;
    t1[e] = in ? A.img1[o] : 0.f;
  }
  __syncthreads();
  double part = 0.0;
  const int opy = threadIdx.x / SSIM_TILE, opx = threadIdx.x % SSIM_TILE;   // the thread's output pixel of the tile
  for (int c = 0; c < 3; ++c) {
    for (int e = threadIdx.x; e < tin * SSIM_TILE; e += 256) {   // rows: (input row py, output column px)
      const int py = e / SSIM_TILE, px = e % SSIM_TILE;
      double r0 = 0, r1 = 0, r00 = 0, r11 = 0, r01 = 0;
      for (int dx = 0; dx < A.fs; ++dx) {
        const int i = (py * tin + (px + dx)) * 3 + c;
        const float a = t0[i], b = t1[i];
        const double w = A.filt[dx];
        r0 += w * (double)a; r1 += w * (double)b;
        r00 += w * (double)__fmul_rn(a, a); r11 += w * (double)__fmul_rn(b, b); r01 += w * (double)__fmul_rn(a, b);
      }
      hrow[0][py][px] = r0; hrow[1][py][px] = r1; hrow[2][py][px] = r00; hrow[3][py][px] = r11; hrow[4][py][px] = r01;
    }
    __syncthreads();
    if (y0 + opy < Ho && x0 + opx < Wo) {   // columns
      double mu0 = 0, mu1 = 0, s00 = 0, s11 = 0, s01 = 0;
      for (int dy = 0; dy < A.fs; ++dy) {
        const double w = A.filt[dy];
        mu0 += w * hrow[0][opy + dy][opx]; mu1 += w * hrow[1][opy + dy][opx];
        s00 += w * hrow[2][opy + dy][opx]; s11 += w * hrow[3][opy + dy][opx]; s01 += w * hrow[4][opy + dy][opx];
      }
      const double mu00 = mu0 * mu0, mu11 = mu1 * mu1, mu01 = mu0 * mu1;
      double sg00 = s00 - mu00, sg11 = s11 - mu11, sg01 = s01 - mu01;
      sg00 = sg00 > 0.0 ? sg00 : 0.0;
      sg11 = sg11 > 0.0 ? sg11 : 0.0;
      const double lim = sqrt(sg00 * sg11), mag = fabs(sg01) < lim ? fabs(sg01) : lim;
      sg01 = sg01 > 0.0 ? mag : (sg01 < 0.0 ? -mag : 0.0);
      const double v = ((2.0 * mu01 + A.c1) * (2.0 * sg01 + A.c2)) / ((mu00 + mu11 + A.c1) * (sg00 + sg11 + A.c2));
      part += v;
      if (A.map) A.map[((int64_t)(y0 + opy) * Wo + (x0 + opx)) * 3 + c] = (float)v;
    }
    __syncthreads();   // the next channel overwrites the row sums
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0 && A.sum) atomicAdd(A.sum, red[0]);
}

}  // namespace

extern "C" {

int ego_rgb_ssim(const float* img0, const float* img1, int32_t H, int32_t W, double max_val, int32_t filter_size, double filter_sigma,
                 double k1, double k2, double* sum, float* ssim_map, void* stream) {
  EGO_TRACE("ego_rgb_ssim");
  EGO_REQUIRE(filter_size >= 1 && filter_size <= SSIM_MAX_FS && H >= filter_size && W >= filter_size && filter_sigma > 0.0,
              "rgb_ssim: bad size (image must cover the filter, filter_size <= 15)");
  EGO_REQUIRE(img0 && img1 && (sum || ssim_map), "rgb_ssim: null argument");
  SsimArgs a{};
  a.img0 = img0; a.img1 = img1; a.H = H; a.W = W; a.fs = filter_size; a.sum = sum; a.map = ssim_map;
  // utils.py:117-121
  const int hw = filter_size / 2;
  const double shift = (2 * hw - filter_size + 1) / 2.0;
  double tot = 0.0;
  for (int i = 0; i < filter_size; ++i) {
    const double f = ((double)(i - hw) + shift) / filter_sigma;
    a.filt[i] = exp(-0.5 * f * f);
    tot += a.filt[i];
  }
  for (int i = 0; i < filter_size; ++i) a.filt[i] /= tot;
  a.c1 = (k1 * max_val) * (k1 * max_val);
  a.c2 = (k2 * max_val) * (k2 * max_val);
  const int Ho = H - filter_size + 1, Wo = W - filter_size + 1;
  dim3 grid((Wo + SSIM_TILE - 1) / SSIM_TILE, (Ho + SSIM_TILE - 1) / SSIM_TILE);
  k_rgb_ssim<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_rgb_ssim");
}

}  // extern "C"
