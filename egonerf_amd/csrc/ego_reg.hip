// Training-step table ops of the EgoNeRF path that sit next to the render (train.py:245-330): the TV / L1 / ortho
// regularisers with their gradients fused into the same pass, the ray-entropy loss, coarse-to-fine table resampling and
// a multi-tensor Adam.  All tables are channel-last ([H][W][C] fp32), all kernels are HBM-bound streaming passes.
#include "ego_device.h"
#include "ego_host.h"

namespace {

__device__ __forceinline__ double block_sum_256(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (tid < d) red[tid] += red[tid + d];
    __syncthreads();
  }
  return red[0];
}

// utils.py:155-171 (TVLoss) on one plane, value and gradient in one pass.  ah = scale*2/count_h, aw = scale*2/count_w.
__global__ __launch_bounds__(256) void k_tv_plane(const float* __restrict__ x, int C, int H, int W, float ah, float aw,
                                                  double* __restrict__ value, float* __restrict__ grad) {
  __shared__ double red[256];
  const int64_t n = (int64_t)H * W * C;
  const int64_t rowstride = (int64_t)W * C;
  double part = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t pix = i / C;
    const int xx = (int)(pix % W), yy = (int)(pix / W);
    const float v = x[i];
    const float dh = (yy + 1 < H) ? x[i + rowstride] - v : 0.f;
    const float dw = (xx + 1 < W) ? x[i + C] - v : 0.f;
    part += (double)(dh * dh * ah + dw * dw * aw);
    if (grad) {
      const float bh = (yy > 0) ? v - x[i - rowstride] : 0.f;
      const float bw = (xx > 0) ? v - x[i - C] : 0.f;
      grad[i] += 2.f * (ah * (bh - dh) + aw * (bw - dw));
    }
  }
  const double tot = block_sum_256(part, red);
  if (threadIdx.x == 0 && value) atomicAdd(value, tot);
}

// EgoNeRF.py:206-212 (density_L1) on one table: value += scale * mean|x|, grad += scale * sign(x) / n
__global__ __launch_bounds__(256) void k_l1(const float* __restrict__ x, int64_t n, float a, double* __restrict__ value,
                                            float* __restrict__ grad) {
  __shared__ double red[256];
  double part = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    part += (double)fabsf(v);
    if (grad) grad[i] += a * (float)((v > 0.f) - (v < 0.f));
  }
  const double tot = block_sum_256(part, red);
  if (threadIdx.x == 0 && value) atomicAdd(value, tot * (double)a);
}

// EgoNeRF.py:189-201 (vectorDiffs) on one line table L [n][C]: G = L^T L, value += scale * mean |offdiag G|,
// grad[k][i] += scale * 2 / (C (C-1)) * sum_{j != i} sign(G_ij) L[k][j].  One workgroup per table.
__global__ __launch_bounds__(256) void k_line_ortho(const float* __restrict__ L, int C, int n, float scale,
                                                    double* __restrict__ value, float* __restrict__ grad) {
  __shared__ float G[64 * 64];
  __shared__ double red[256];
  const int tid = threadIdx.x;
  for (int e = tid; e < C * C; e += 256) {
    const int i = e / C, j = e % C;
    float acc = 0.f;
    for (int k = 0; k < n; ++k) acc += L[k * C + i] * L[k * C + j];
    G[e] = acc;
  }
  __syncthreads();
  double part = 0.0;
  for (int e = tid; e < C * C; e += 256)
    if (e / C != e % C) part += (double)fabsf(G[e]);
  const float a = scale / (float)(C * (C - 1));
  const double tot = block_sum_256(part, red);
  if (tid == 0 && value) atomicAdd(value, tot * (double)a);
  if (!grad) return;
  for (int e = tid; e < n * C; e += 256) {
    const int k = e / C, i = e % C;
    float acc = 0.f;
    for (int j = 0; j < C; ++j) {
      const float g = G[i * C + j];
      const float s = (j == i) ? 0.f : (float)((g > 0.f) - (g < 0.f));
      acc += s * L[k * C + j];
    }
    grad[e] += 2.f * a * acc;
  }
}

// utils.py:175-183 (ray_entropy_loss), one wave per ray: value += H_ray / N, g_alpha = dH/dalpha / N
__global__ __launch_bounds__(256) void k_ray_entropy(const float* __restrict__ alpha, int64_t N, int S, int stride,
                                                     double* __restrict__ value, float* __restrict__ g_alpha) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= N) return;
  const float* a = alpha + ray * stride;
  float sum = 0.f;
  for (int s = lane; s < S; s += 64) sum += a[s];
  const float A = wave_sum(sum) + 1e-10f;
  const float inv_ln2 = 1.44269504088896340736f;
  float ent = 0.f, ph = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float p = a[s] / A;
    const float lg = log2f(p + 1e-10f);
    ent -= p * lg;
    ph += p * -(lg + p / (p + 1e-10f) * inv_ln2);
  }
  ent = wave_sum(ent);
  ph = wave_sum(ph);
  const float invN = 1.f / (float)N;
  if (g_alpha)
    for (int s = lane; s < S; s += 64) {
      const float p = a[s] / A;
      const float h = -(log2f(p + 1e-10f) + p / (p + 1e-10f) * inv_ln2);
      g_alpha[ray * stride + s] = (h - ph) / A * invN;
    }
  if (lane == 0 && value) atomicAdd(value, (double)ent * (double)invN);
}

// coordinates.py:27-39 / :226-266: bilinear (align_corners, zero padding) resample of a channel-last table at per-axis
// normalised coordinates xs [W2], ys [H2]
__global__ void k_resample_table(const float* __restrict__ src, int C, int H, int W, const float* __restrict__ xs,
                                 const float* __restrict__ ys, int H2, int W2, float* __restrict__ dst) {
#pragma clang fp contract(fast)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)H2 * W2 * C) return;
  const int c = (int)(i % C);
  const int64_t pix = i / C;
  const int x2 = (int)(pix % W2), y2 = (int)(pix / W2);
  const Lin1 X = lin_setup(xs[x2], W), Y = lin_setup(ys[y2], H);
  const float* r0 = src + ((int64_t)Y.i0 * W) * C + c;
  const float* r1 = src + ((int64_t)Y.i1 * W) * C + c;
  dst[i] = (r0[(int64_t)X.i0 * C] * X.w0 + r0[(int64_t)X.i1 * C] * X.w1) * Y.w0 +
           (r1[(int64_t)X.i0 * C] * X.w0 + r1[(int64_t)X.i1 * C] * X.w1) * Y.w1;
}

// torch.optim.Adam (train.py:182, betas (0.9, 0.99)) over up to ADAM_MAX tensors per launch; 1024 elements per block
constexpr int ADAM_MAX = 40;
struct AdamBatch {
  float* p[ADAM_MAX];
  const float* g[ADAM_MAX];
  float* m[ADAM_MAX];
  float* v[ADAM_MAX];
  int64_t n[ADAM_MAX];
  int32_t chunk0[ADAM_MAX + 1];  // first block of each tensor
  float step_size[ADAM_MAX];     // lr / (1 - beta1^t)
  int32_t count;
  float beta1, beta2, eps, bc2_sqrt;
  const double* clock;           // optional device clock (ego_adam_step_graph): [2] = lr scale / (1 - beta1^t), [3] = sqrt(1 - beta2^t)
};

// The step count and the learning-rate scale live on the device, so that a captured hipGraph of the training step advances them
// by itself: clock = {t, lr scale, lr scale / (1 - beta1^t), sqrt(1 - beta2^t)}.  One thread, double precision like the host path.
__global__ void k_adam_clock(double* clock, double beta1, double beta2, double lr_factor) {
  const double t = clock[0] + 1.0;
  clock[0] = t;
  clock[2] = clock[1] / (1.0 - pow(beta1, t));
  clock[3] = sqrt(1.0 - pow(beta2, t));
  clock[1] *= lr_factor;   // train.py:328-329: the decay follows the step
}

__global__ __launch_bounds__(256) void k_adam(AdamBatch B) {
  int t = 0;
  while (t + 1 < B.count && (int)blockIdx.x >= B.chunk0[t + 1]) ++t;
  const int64_t base = (int64_t)(blockIdx.x - B.chunk0[t]) * 1024;
  float* __restrict__ p = B.p[t];
  const float* __restrict__ g = B.g[t];
  float* __restrict__ m = B.m[t];
  float* __restrict__ v = B.v[t];
  const int64_t n = B.n[t];
  const float ss = B.clock ? (float)((double)B.step_size[t] * B.clock[2]) : B.step_size[t];
  const float bc2_sqrt = B.clock ? (float)B.clock[3] : B.bc2_sqrt;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    if (i >= n) break;
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - B.beta1);           // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * B.beta2 + (1.f - B.beta2) * gi * gi;     // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + B.eps;
    p[i] = p[i] - ss * (mi / denom);
  }
}

inline unsigned stream_blocks(int64_t n) {
  const int64_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" {

int ego_tv_plane(const float* table, int32_t C, int32_t H, int32_t W, float scale, double* value, float* grad, void* stream) {
  EGO_TRACE("ego_tv_plane");
  EGO_REQUIRE(C >= 1 && H >= 2 && W >= 2, "tv_plane: bad size");
  EGO_REQUIRE(table && (value || grad), "tv_plane: null argument");
  const float ah = scale * 2.f / ((float)C * (float)(H - 1) * (float)W);
  const float aw = scale * 2.f / ((float)C * (float)H * (float)(W - 1));
  k_tv_plane<<<stream_blocks((int64_t)C * H * W), 256, 0, (hipStream_t)stream>>>(table, C, H, W, ah, aw, value, grad);
  return ego_launch_status("k_tv_plane");
}

int ego_l1_table(const float* table, int64_t n, float scale, double* value, float* grad, void* stream) {
  EGO_TRACE("ego_l1_table");
  EGO_REQUIRE(n >= 1, "l1_table: bad size");
  EGO_REQUIRE(table && (value || grad), "l1_table: null argument");
  k_l1<<<stream_blocks(n), 256, 0, (hipStream_t)stream>>>(table, n, scale / (float)n, value, grad);
  return ego_launch_status("k_l1");
}

int ego_line_ortho(const float* line, int32_t C, int32_t n, float scale, double* value, float* grad, void* stream) {
  EGO_TRACE("ego_line_ortho");
  EGO_REQUIRE(C >= 2 && C <= 64 && n >= 1, "line_ortho: bad size (2 <= n_comp <= 64)");
  EGO_REQUIRE(line && (value || grad), "line_ortho: null argument");
  k_line_ortho<<<1, 256, 0, (hipStream_t)stream>>>(line, C, n, scale, value, grad);
  return ego_launch_status("k_line_ortho");
}

int ego_ray_entropy(const float* alpha, int64_t N, int32_t S, int32_t stride, double* value, float* g_alpha, void* stream) {
  EGO_TRACE("ego_ray_entropy");
  EGO_REQUIRE(N >= 0 && S >= 1 && stride >= S, "ray_entropy: bad size");
  if (N == 0) return EGO_OK;
  EGO_REQUIRE(alpha && (value || g_alpha), "ray_entropy: null argument");
  k_ray_entropy<<<(unsigned)((N + 3) / 4), 256, 0, (hipStream_t)stream>>>(alpha, N, S, stride, value, g_alpha);
  return ego_launch_status("k_ray_entropy");
}

int ego_resample_table(const float* src, int32_t C, int32_t H, int32_t W, const float* xs, const float* ys, int32_t H2, int32_t W2,
                       float* dst, void* stream) {
  EGO_TRACE("ego_resample_table");
  EGO_REQUIRE(C >= 1 && H >= 1 && W >= 1 && H2 >= 1 && W2 >= 1, "resample_table: bad size");
  EGO_REQUIRE(src && xs && ys && dst, "resample_table: null argument");
  const int64_t n = (int64_t)H2 * W2 * C;
  k_resample_table<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, C, H, W, xs, ys, H2, W2, dst);
  return ego_launch_status("k_resample_table");
}

static int adam_launch(const ego_adam_tensor* tensors, int32_t count, float beta1, float beta2, float eps, double bc1, double bc2,
                       const double* clock, void* stream) {
  for (int first = 0; first < count; first += ADAM_MAX) {
    AdamBatch b{};
    b.count = count - first < ADAM_MAX ? count - first : ADAM_MAX;
    b.beta1 = beta1; b.beta2 = beta2; b.eps = eps; b.bc2_sqrt = (float)sqrt(bc2); b.clock = clock;
    int blocks = 0;
    for (int i = 0; i < b.count; ++i) {
      const ego_adam_tensor& t = tensors[first + i];
      EGO_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq && t.n >= 0, "adam_step: null tensor / negative size");
      b.p[i] = t.param; b.g[i] = t.grad; b.m[i] = t.exp_avg; b.v[i] = t.exp_avg_sq; b.n[i] = t.n;
      b.step_size[i] = (float)((double)t.lr / bc1);
      b.chunk0[i] = blocks;
      blocks += (int)((t.n + 1023) / 1024);
    }
    b.chunk0[b.count] = blocks;
    if (blocks == 0) continue;
    k_adam<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(b);
    if (int e = ego_launch_status("k_adam")) return e;
  }
  return EGO_OK;
}

int ego_adam_step(const ego_adam_tensor* tensors, int32_t count, float beta1, float beta2, float eps, int32_t step, void* stream) {
  EGO_TRACE("ego_adam_step");
  EGO_REQUIRE(count >= 0 && step >= 1, "adam_step: bad count / step (steps count from 1)");
  if (count == 0) return EGO_OK;
  EGO_REQUIRE(tensors, "adam_step: null argument");
  return adam_launch(tensors, count, beta1, beta2, eps, 1.0 - pow((double)beta1, (double)step), 1.0 - pow((double)beta2, (double)step), nullptr, stream);
}

int ego_adam_step_graph(const ego_adam_tensor* tensors, int32_t count, float beta1, float beta2, float eps, double lr_factor, double* clock,
                        void* stream) {
  EGO_TRACE("ego_adam_step_graph");
  EGO_REQUIRE(count >= 0 && lr_factor > 0.0, "adam_step_graph: bad count / lr_factor");
  if (count == 0) return EGO_OK;
  EGO_REQUIRE(tensors && clock, "adam_step_graph: null argument");
  k_adam_clock<<<1, 1, 0, (hipStream_t)stream>>>(clock, (double)beta1, (double)beta2, lr_factor);
  if (int e = ego_launch_status("k_adam_clock")) return e;
  return adam_launch(tensors, count, beta1, beta2, eps, 1.0, 0.0, clock, stream);
}

}  // extern "C"
