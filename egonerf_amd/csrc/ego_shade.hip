// libegonerf_hip.so, part 2: appearance lookup -> basis -> positional encoding -> MLP_Fea, on the
// fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, so parity with the reference's
// fp32 Linear layers is rounding-order only).
//
// Work mapping (wave64): a wave owns a tile of 32 consecutive samples; lane l serves sample
// j = l & 31 and "half" h = l >> 5.  In D = A*B (32x32x2): A[i][k] comes from lane (i, k) = (l&31, l>>5),
// B[k][j] from lane (j, k) = (l&31, l>>5), and D[i][j] lands in lane j + 32*((i>>2)&1), register
// r = (i&3) + 4*(i>>3).  So with samples on the N axis a lane only ever supplies / receives values of
// ITS OWN sample: the gathered products feed the basis MFMAs straight from registers, the basis
// output feeds layer 1, layer 1 feeds layer 2 — no cross-lane traffic, no LDS round trip for
// activations.  The price is a fixed K-order per lane half, which ego_pack_mlp bakes into the weights:
//   * lane half h gathers appearance channels [24h, 24h+24) of each of the 3 planes  (72 k-steps)
//   * basis rows are permuted so half h receives features f = 2r + h in register r   (14 slots)
//   * layer-1 k-steps: 14 slots x (f, sin f, sin 2f, cos f, cos 2f) + 8 view slots + 2 zero pads = 80
//     (slot-major, so each encoding is produced right before the MFMAs that consume it)
//   * layer-2 k-step m*16 + r consumes hidden unit m*32 + (r&3) + 8*(r>>2) + 4h
// Layer 3 (128 -> 3) runs on the VALU from the layer-2 accumulators + one xor-32 exchange.
//
// W1/W2/W3/biases (150.5 KB packed) live in LDS for the whole persistent workgroup (8 waves = 2 per
// SIMD, so one wave's gather overlaps the other's MFMA chain); the basis fragments stream from L2.
#include "ego_device.h"
#include "ego_host.h"
#include "variants.h"
#include "ego_generic.h"

namespace {

constexpr int APP_C = 48;      // appearance components per plane
constexpr int APP_HALF = 24;   // channels gathered by one lane half
constexpr int APP_DIM = 27;
constexpr int HID = 128;
constexpr int NSLOT = 14;      // feature slots per lane half
constexpr int KS_BASIS = 72;
constexpr int KS1 = 80;
constexpr int KS2 = 64;
constexpr int MLP_IN = 150;

constexpr int OFF_W1 = 0;                           // [KS1/4][4 m][64 lanes][4]
constexpr int OFF_W2 = OFF_W1 + KS1 * 4 * 64;       // [KS2/4][4 m][64][4]
constexpr int OFF_B1 = OFF_W2 + KS2 * 4 * 64;       // [4 m][2 h][16 r]
constexpr int OFF_B2 = OFF_B1 + 128;
constexpr int OFF_W3 = OFF_B2 + 128;                // [4 m][2 h][16 r][4 (c0,c1,c2,0)]
constexpr int OFF_B3 = OFF_W3 + 512;                // [4]
constexpr int LDS_W_FLOATS = OFF_B3 + 4;            // 37636 floats = 150544 B
constexpr int OFF_BASIS = LDS_W_FLOATS;             // [2 g][KS_BASIS/4][64][4]
constexpr int PACKED_FLOATS = OFF_BASIS + 2 * (KS_BASIS / 4) * 64 * 4;
constexpr int LUT_MAX = 1024;

__host__ __device__ constexpr int slot_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// appearance channel (0..143, plane-major) gathered by lane half h as its kk-th product: the halves interleave at
// 16-byte granularity (half h owns float4 quads 2i+h of a texel), so the two lanes of a sample always read the
// same 64-byte line in a given load instruction -> half as many L1 tag lookups as a [0,24) / [24,48) split.
// K order of the f16x3 kernel's gather (gather_team4 below): samples are gathered by 4-lane teams (part p reads quad
// 4i+p of line i, so a team reads whole 64-byte lines) and transposed into the 2-lanes-per-sample MFMA layout with
// v_permlane16_swap + v_permlane32_swap.  Lane half h ends up with the parts p = h (first 12 products of a plane) and
// p = h + 2 (next 12): product kk = plane*24 + half*12 + i*4 + c is channel plane*48 + 16i + 4(h + 2 half) + c.
// Activation dumps of the training forward / backward (logical [M][2 K] matrices: x 160, h1/h2/dh1/dh2 128, v 144 columns).
// Logical column of element kk of lane half h: dump_col(kk, h) (the float4 quads of the two halves interleave).  Storage is
// tile-blocked and lane-major: [tile = m / 32][quad pair q = kk / 4][lane = 32 h + (m % 32)][4 floats], so every store / load
// instruction of the shade kernels moves 1 KB of contiguous memory (row-major rows gave 32-byte pieces 640 B apart and
// 2.5 TB/s; this layout 1.88 -> 1.37 ms for the dumping forward).  Buffers hold ceil(M / 32) * 32 rows.
__host__ __device__ constexpr int64_t dump_off(int64_t tile, int width, int q, int h, int j) {
  return tile * (32 * (int64_t)width) + q * 256 + (h * 32 + j) * 4;
}
__host__ __device__ constexpr int dump_col(int kk, int h) { return (kk >> 2) * 8 + h * 4 + (kk & 3); }

__host__ __device__ constexpr int app_channel_g(int kk, int h) {
  return (kk / APP_HALF) * APP_C + (((kk % APP_HALF) % 12) / 4) * 16 + 4 * (h + 2 * ((kk % APP_HALF) / 12)) + (kk % 4);
}

// (fp32-MFMA kernel k_shade: lane half h owns float4 quads 2i+h)
__host__ __device__ constexpr int app_channel(int kk, int h) {
  return (kk / APP_HALF) * APP_C + ((kk % APP_HALF) / 4) * 8 + 4 * h + (kk % 4);
}

// reference MLP input column held by X register kk of lane half h (-1: zero weight)
__device__ int x_channel(int kk, int h) {
  if (kk < 5 * NSLOT) {
    const int kind = kk % 5, r = kk / 5, f = 2 * r + h;
    if (f >= APP_DIM) return -1;
    const int pe0 = APP_DIM + 3;               // 30: sin block of the feature PE
    const int pe1 = pe0 + 2 * APP_DIM;         // 84: cos block
    switch (kind) {
      case 0: return f;
      case 1: return pe0 + 2 * f;
      case 2: return pe0 + 2 * f + 1;
      case 3: return pe1 + 2 * f;
      default: return pe1 + 2 * f + 1;
    }
  }
  if (kk < 5 * NSLOT + 8) {
    const int t = (kk - 5 * NSLOT) + 8 * h;
    // d0 d1 d2 | sin d0, sin 2d0, sin d1, sin 2d1, sin d2, sin 2d2 | cos ... | pad
    return t < 3 ? APP_DIM + t : (t < 15 ? 138 + (t - 3) : -1);
  }
  return -1;
}

__global__ void k_pack_mlp(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                           const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
                           const float* __restrict__ basis_yin, const float* __restrict__ basis_yang,
                           float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= PACKED_FLOATS) return;
  float v = 0.f;
  if (idx < OFF_W2) {  // W1 fragments
    const int j = idx & 3, lane = (idx >> 2) & 63, m = (idx >> 8) & 3, kk4 = idx >> 10;
    const int ch = x_channel(kk4 * 4 + j, lane >> 5);
    if (ch >= 0) v = w1[(m * 32 + (lane & 31)) * MLP_IN + ch];
  } else if (idx < OFF_B1) {  // W2 fragments
    const int e = idx - OFF_W2;
    const int j = e & 3, lane = (e >> 2) & 63, m2 = (e >> 8) & 3, kk4 = e >> 10;
    const int kk = kk4 * 4 + j;
    v = w2[(m2 * 32 + (lane & 31)) * HID + (kk >> 4) * 32 + slot_row(kk & 15, lane >> 5)];
  } else if (idx < OFF_W3) {  // biases of layers 1, 2 in accumulator layout
    const int e = (idx - OFF_B1) & 127;
    const float* b = (idx < OFF_B2) ? b1 : b2;
    v = b[(e >> 5) * 32 + slot_row(e & 15, (e >> 4) & 1)];
  } else if (idx < OFF_B3) {  // W3 in accumulator layout
    const int e = idx - OFF_W3;
    const int c = e & 3, r = (e >> 2) & 15, h = (e >> 6) & 1, m = e >> 7;
    if (c < 3) v = w3[c * HID + m * 32 + slot_row(r, h)];
  } else if (idx < OFF_BASIS) {
    const int c = idx - OFF_B3;
    if (c < 3) v = b3[c];
  } else {  // basis fragments
    const int e = idx - OFF_BASIS;
    const int j = e & 3, lane = (e >> 2) & 63, kk4 = (e >> 8) % (KS_BASIS / 4), g = e / (KS_BASIS / 4 * 256);
    const int i = lane & 31, h = lane >> 5, kk = kk4 * 4 + j;
    const int rh = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3), f = 2 * r + rh;  // feature delivered to tile row i
    if (r < NSLOT && f < APP_DIM) {
      const int col = app_channel(kk, h);
      v = (g ? basis_yang : basis_yin)[f * (3 * APP_C) + col];
    }
  }
  out[idx] = v;
}

// ---- fp16-split ("f16x3") weight layout ------------------------------------------------------------------
// Same regions/offsets as the fp32 blob, but the three matrix regions hold fp16 pairs: w = hi + lo with
// hi = fp16(w), lo = fp16(w - hi) (22 significant bits).  Fragment order [k-step][m-tile][term hi|lo][lane][8 k]:
// one ds_read_b128 / global_load_dwordx4 per (step, tile, term) per lane, conflict-free.  A lane's 8 k of a step
// are its values 8*step .. 8*step+7 in the same per-half K order as the fp32 layout.
constexpr int KH1 = KS1 / 8, KH2 = KS2 / 8, KHB = KS_BASIS / 8;
constexpr int BASIS16_FLOATS_C = 2 * KHB * 2 * 64 * 4;  // third blob region (basis fragments for the fp16-table gather)

__device__ inline void split_weight(float w, _Float16& hi, _Float16& lo) {
  hi = (_Float16)w;
  lo = (_Float16)(w - (float)hi);
}

__global__ void k_pack_mlp_h(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                             const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
                             const float* __restrict__ basis_yin, const float* __restrict__ basis_yang,
                             const float* __restrict__ f32_blob, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one 32-bit slot = two fp16
  if (idx >= PACKED_FLOATS) return;
  if (idx >= OFF_B1 && idx < OFF_BASIS) { out[idx] = f32_blob[idx]; return; }  // biases, W3, b3 stay fp32
  _Float16 pr[2];
  for (int p = 0; p < 2; ++p) {
    float w = 0.f;
    int term;
    if (idx < OFF_W2) {
      const int hidx = (idx - OFF_W1) * 2 + p;
      const int e = hidx & 7, lane = (hidx >> 3) & 63, step = hidx >> 12, mt = (hidx >> 10) & 3;
      term = (hidx >> 9) & 1;
      const int ch = x_channel(step * 8 + e, lane >> 5);
      if (ch >= 0) w = w1[(mt * 32 + (lane & 31)) * MLP_IN + ch];
    } else if (idx < OFF_B1) {
      const int hidx = (idx - OFF_W2) * 2 + p;
      const int e = hidx & 7, lane = (hidx >> 3) & 63, step = hidx >> 12, mt = (hidx >> 10) & 3;
      term = (hidx >> 9) & 1;
      const int kk = step * 8 + e;
      w = w2[(mt * 32 + (lane & 31)) * HID + (kk >> 4) * 32 + slot_row(kk & 15, lane >> 5)];
    } else {
      const int hidx = (idx - OFF_BASIS) * 2 + p;
      const int e = hidx & 7, lane = (hidx >> 3) & 63, sg = hidx >> 10;  // sg = g * KHB + step
      term = (hidx >> 9) & 1;
      const int g = sg / KHB, kk = (sg % KHB) * 8 + e;
      const int i = lane & 31, h = lane >> 5;
      const int rh = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3), f = 2 * r + rh;
      if (r < NSLOT && f < APP_DIM) {
        const int col = app_channel_g(kk, h);
        w = (g ? basis_yang : basis_yin)[f * (3 * APP_C) + col];
      }
    }
    _Float16 hi, lo;
    split_weight(w, hi, lo);
    pr[p] = term ? lo : hi;
  }
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  h2v v = {pr[0], pr[1]};
  out[idx] = __builtin_bit_cast(float, v);
}

// basis fragments for the fp16-table gather: same [g][step][term][lane][8] order, columns follow app_channel_f16
__host__ __device__ constexpr int app_channel_f16_fwd(int kk, int h) {
  return (kk / APP_HALF) * APP_C + ((kk % APP_HALF) / 8) * 16 + 8 * h + (kk % 8);
}

__global__ void k_pack_basis16(const float* __restrict__ basis_yin, const float* __restrict__ basis_yang, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * KHB * 2 * 64 * 4) return;
  _Float16 pr[2];
  for (int p = 0; p < 2; ++p) {
    const int hidx = idx * 2 + p;
    const int e = hidx & 7, lane = (hidx >> 3) & 63, term = (hidx >> 9) & 1, sg = hidx >> 10;
    const int g = sg / KHB, kk = (sg % KHB) * 8 + e;
    const int i = lane & 31, h = lane >> 5;
    const int rh = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3), f = 2 * r + rh;
    float w = 0.f;
    if (r < NSLOT && f < APP_DIM) w = (g ? basis_yang : basis_yin)[f * (3 * APP_C) + app_channel_f16_fwd(kk, h)];
    _Float16 hi, lo;
    split_weight(w, hi, lo);
    pr[p] = term ? lo : hi;
  }
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  h2v v = {pr[0], pr[1]};
  out[idx] = __builtin_bit_cast(float, v);
}

// ---- fp16 main term + fp8 correction terms ("f16f8") weight layout ----------------------------------------------------
// w*x = w_hi*x_hi (one v_mfma_f32_32x32x16_f16 per k-step, as in f16x3) + [w_lo*x_hi + w_hi*x_lo] on the block-scaled fp8 path:
// ONE v_mfma_scale_f32_32x32x64_f8f6f4 per PAIR of k-steps carries both correction terms (K = 64 = 2 terms x 2 steps x 8 values
// x 2 lane halves), at ~1.9x the matrix-pipe time of one fp16 instruction instead of 4x.  The corrections are ~2^-11 of the main
// term, so e4m3's 4 significant bits leave ~2^-16 relative error per product (measured: DESIGN.md 4.1, profiles/r02/precision_sweep.json).
// Operand bytes of a lane (row i = lane & 31 of m-tile mt, half h = lane >> 5; probed layout: byte b = k offset b of the lane's
// 32-wide K block, tools/fp8_layout_probe.hip):
//   A: [0..7] e4m3(w_lo * 2^11) step 2p | [8..15] same, step 2p+1 | [16..23] e4m3(w_hi) step 2p | [24..31] e4m3(w_hi) step 2p+1
//   B: [0..7] e4m3(x)           step 2p | [8..15] same, step 2p+1 | [16..23] e4m3(x_lo * 2^11) step 2p | [24..31] ..., step 2p+1
// and the instruction's E8M0 block scale of A is 2^-11 (exponent byte 116), of B 2^0 (127).
// Region layout (32-bit slots, same OFF_W1 / OFF_W2 extents as the other layouts, so the LDS image keeps its size):
//   per layer: hi fragments [step][m-tile][lane][8 halves], then fp8 fragments [pair][m-tile][part 0|1][lane][16 bytes]
constexpr int F8_HI1 = KH1 * 4 * 64 * 4;          // 32-bit slots of layer 1's hi part (10240)
constexpr int F8_HI2 = KH2 * 4 * 64 * 4;          // layer 2 (8192)
constexpr int F8_FLOATS = OFF_B1;                 // W1 + W2 regions only; biases / W3 come from the f16x3 blob

__device__ inline uint32_t e4m3_byte(float v) {
  return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xffu;
}

__global__ void k_pack_mlp_f8(const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one 32-bit slot
  if (idx >= F8_FLOATS) return;
  const bool l2 = idx >= OFF_W2;
  const int e0 = l2 ? idx - OFF_W2 : idx;
  const int hi_slots = l2 ? F8_HI2 : F8_HI1;
  auto weight = [&](int step, int e, int lane, int mt) -> float {
    if (!l2) {
      const int ch = x_channel(step * 8 + e, lane >> 5);
      return ch >= 0 ? w1[(mt * 32 + (lane & 31)) * MLP_IN + ch] : 0.f;
    }
    const int kk = step * 8 + e;
    return w2[(mt * 32 + (lane & 31)) * HID + (kk >> 4) * 32 + slot_row(kk & 15, lane >> 5)];
  };
  uint32_t word = 0;
  if (e0 < hi_slots) {  // [step][mt][lane][8 halves]
    _Float16 pr[2];
    for (int p = 0; p < 2; ++p) {
      const int hidx = e0 * 2 + p;
      const int e = hidx & 7, lane = (hidx >> 3) & 63, mt = (hidx >> 9) & 3, step = hidx >> 11;
      _Float16 hi, lo;
      split_weight(weight(step, e, lane, mt), hi, lo);
      pr[p] = hi;
    }
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    h2v v = {pr[0], pr[1]};
    word = __builtin_bit_cast(uint32_t, v);
  } else {  // [pair][mt][part][lane][16 bytes]
    for (int b = 0; b < 4; ++b) {
      const int bidx = (e0 - hi_slots) * 4 + b;
      const int byte = bidx & 15, lane = (bidx >> 4) & 63, part = (bidx >> 10) & 1, mt = (bidx >> 11) & 3, pair = bidx >> 13;
      const int pos = part * 16 + byte;                    // byte of the 32-byte operand
      const int step = 2 * pair + ((pos >> 3) & 1), e = pos & 7;
      const float w = weight(step, e, lane, mt);
      _Float16 hi, lo;
      split_weight(w, hi, lo);
      const float v = pos < 16 ? (w - (float)hi) * 2048.0f : (float)hi;
      word |= e4m3_byte(v) << (8 * b);
    }
  }
  out[idx] = __builtin_bit_cast(float, word);
}

// ---- fp16 main term + fp6 correction terms ("f16f6") weight layout ------------------------------------------------------------
// Same split as f16f8, with the two correction terms on the fp6 (e2m3) path of v_mfma_scale_f32_32x32x64_f8f6f4, which runs at
// 1.18x the time of one fp16 32x32x16 instruction where the fp8 path takes 2.0x, and whose operands one conversion instruction
// produces for 32 values at a time (v_cvt_scalef32_pk32_fp6_f16 / v_cvt_scalef32_2xpk16_fp6_f32: 64 clk per 32 values against
// 16 x ~10 clk on the fp8 path; tools/fp6_probe.hip, profiles/r04/fp6_probe.txt).  e2m3 spans 6 binades only, so every block of 32
// K values of a lane carries its own power-of-two scale (the instruction's E8M0 block scale is per lane): static for the weights
// (from the block's largest magnitude), dynamic for the activations (exponent of the largest |x| of the lane's 32 values).
// A GROUP is 4 k-steps = 32 values of a lane half; per group and m-tile two fp6 MFMAs: term 0 = w_lo * x_hi, term 1 = w_hi * x_lo.
// Element e of a lane's 192-bit operand sits at bits [6e, 6e + 6); it carries the lane's K value f6_value(layer, group, term, e):
//   term 0 (B from pk32_fp6_f16 of the four steps' packed halves): value 32 g + e
//   term 1 (B from 2xpk16_fp6_f32 of the residuals, which interleaves its two 16-value sources): even e -> 32 g + e / 2, odd e -> 32 g + 16 + e / 2
//   layer 1's last group holds only steps 8, 9: both terms come from 2xpk16(values, zeros): even e -> 64 + e / 2, odd e -> none
// Scale bytes: with eb = biased exponent of the block's largest |x| (clamped to >= 14) the conversions divide by 2^(eb - 129) (x) and
// 2^(eb - 140) (residual, i.e. 2^-11 further down) and the MFMAs pass eb itself as B's block scale; the 2^-2 / 2^-13 that this
// overstates is folded into A's static scale byte, which is biased(weight block scale) - 2 (term 0) / - 13 (term 1).
// Image layout (32-bit slots; it replaces the first OFF_B1 slots of the LDS image, biases / W3 behind it stay where they are):
//   F6I_HI1 / F6I_HI2: fp16 hi fragments [step][m-tile][lane][8 halves] of layers 1 / 2 (as f16f8)
//   F6I_Q1 / F6I_Q2:   the fp6 operands [group][m-tile][quad 0..2][lane][4]: the two terms' 6 + 6 dwords of a lane as three 16-byte
//                      pieces: term 0 dwords 0-3 | term 1 dwords 0-3 | term 0 dwords 4-5, term 1 dwords 4-5.  (Separate 16 + 8 byte
//                      pieces per term made the compiler pair the 8-byte reads of neighbouring fragments and copy the halves apart,
//                      +42 v_mov per tile; term 0's six dwords followed by term 1's, +81: it does not coalesce a 12-register tuple.)
//   F6I_SC:            scale bytes [group (layer 1's three, then layer 2's two)][lane][2 dwords]: byte mt of dword t = block scale of
//                      (term t, m-tile mt) - the MFMA's op_sel picks the byte
// The kernel addresses LDS as `per-lane base + 16-bit immediate` with one opaque base per 64 KB window (f6_bases): left to itself
// the compiler spends a v_add on every read beyond 64 KB (62 more per tile than f16f8).
constexpr int G6_1 = 3, G6_2 = 2;                 // groups per layer
constexpr int F6I_HI1 = 0, F6I_HI2 = F6I_HI1 + F8_HI1;
constexpr int F6I_Q1 = F6I_HI2 + F8_HI2, F6I_Q2 = F6I_Q1 + G6_1 * 4 * 3 * 256;
constexpr int F6I_SC = F6I_Q2 + G6_2 * 4 * 3 * 256;
constexpr int F6I_END = F6I_SC + (G6_1 + G6_2) * 128;
constexpr int F6_FLOATS = OFF_B1;
static_assert(F6I_END <= F6_FLOATS, "f16f6 image must fit the W1 / W2 part of the LDS image");
static_assert(F6I_SC * 4 < 3 * 65536, "16-byte-stride part of the f16f6 image: three 64 KB windows");

__host__ __device__ constexpr int f6_value(int layer2, int grp, int term, int e) {
  if (!layer2 && grp == G6_1 - 1) return (e & 1) ? -1 : 64 + (e >> 1);
  if (term == 0) return 32 * grp + e;
  return 32 * grp + ((e & 1) ? 16 + (e >> 1) : (e >> 1));
}

// e2m3 code of v (already divided by its block scale): round to nearest even, saturating at 7.5
__device__ inline uint32_t e2m3_code(float v) {
  const uint32_t s = v < 0.f ? 32u : 0u;
  const float a = fminf(fabsf(v), 7.5f);
  if (a < 1.0f) return s | (uint32_t)rintf(a * 8.0f);     // subnormal step 1/8; 8 = the code of 1.0
  const int e = a < 2.0f ? 0 : (a < 4.0f ? 1 : 2);
  int m = (int)rintf(ldexpf(a, 3 - e));                    // 8..16
  int ee = e;
  if (m == 16) { m = 8; ee = e + 1; }
  if (ee > 2) return s | 31u;
  return s | (uint32_t)(((ee + 1) << 3) | (m - 8));
}

__device__ inline float mlp_weight_k(const float* __restrict__ w1, const float* __restrict__ w2, bool l2, int k, int lane, int mt) {
  if (!l2) {
    const int ch = k < KS1 ? x_channel(k, lane >> 5) : -1;
    return ch >= 0 ? w1[(mt * 32 + (lane & 31)) * MLP_IN + ch] : 0.f;
  }
  return w2[(mt * 32 + (lane & 31)) * HID + (k >> 4) * 32 + slot_row(k & 15, lane >> 5)];
}

// the fp16 hi fragments of both layers: one thread per 32-bit slot
__global__ void k_pack_mlp_f6(const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t* o = (uint32_t*)out;
  const int n_hi = F8_HI1 + F8_HI2;
  if (idx < n_hi) {  // [step][mt][lane][8 halves]
    const bool l2 = idx >= F8_HI1;
    const int e0 = l2 ? idx - F8_HI1 : idx;
    _Float16 pr[2];
    for (int p = 0; p < 2; ++p) {
      const int hidx = e0 * 2 + p;
      const int e = hidx & 7, lane = (hidx >> 3) & 63, mt = (hidx >> 9) & 3, step = hidx >> 11;
      _Float16 hi, lo;
      split_weight(mlp_weight_k(w1, w2, l2, step * 8 + e, lane, mt), hi, lo);
      pr[p] = hi;
    }
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    h2v v = {pr[0], pr[1]};
    o[(l2 ? F6I_HI2 : F6I_HI1) + e0] = __builtin_bit_cast(uint32_t, v);
  }
}

// the fp6 operands: one 32-lane group per (layer / group, m-tile, lane, term), lane e = element e (the one-thread-per-operand form took
// 29 us - two dependent chains of 32 scattered weight reads per thread - on every training iteration, whose weights change every step)
__global__ void k_pack_mlp_f6_frag(const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ out) {
  uint32_t* o = (uint32_t*)out;
  const int item = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5), e = threadIdx.x & 31;
  if (item >= (G6_1 + G6_2) * 4 * 64 * 2) return;
  const int term = item & 1, lane = (item >> 1) & 63, mt = (item >> 7) & 3, gg = item >> 9;
  const bool l2 = gg >= G6_1;
  const int grp = l2 ? gg - G6_1 : gg;
  const int k = f6_value(l2, grp, term, e);
  float w = 0.f;
  if (k >= 0) {
    _Float16 hi, lo;
    const float wf = mlp_weight_k(w1, w2, l2, k, lane, mt);
    split_weight(wf, hi, lo);
    w = term == 0 ? wf - (float)hi : (float)hi;   // term 0 carries the exact fp32 residual of the weight, term 1 a copy of its fp16 part
  }
  float amax = fabsf(w);
#pragma unroll
  for (int sh = 1; sh < 32; sh <<= 1) amax = fmaxf(amax, __shfl_xor(amax, sh, 32));
  int E = amax > 0.f ? ilogbf(amax) : -100;
  if (E < -100) E = -100;
  if (ldexpf(amax, 2 - E) > 7.75f) E += 1;          // the largest element would saturate: one binade up
  const float inv = ldexpf(1.0f, 2 - E);            // 1 / block scale, block scale = 2^(E - 2)
  const uint32_t c = e2m3_code(w * inv);
  const int bit = 6 * e, wi = bit >> 5, sh = bit & 31;
  uint32_t mine = 0;                                // after the reduction lane d < 6 holds dword d of the 192-bit operand
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    uint32_t v = (d == wi ? c << sh : 0u) | ((d == wi + 1 && sh > 26) ? c >> (32 - sh) : 0u);
#pragma unroll
    for (int x = 1; x < 32; x <<= 1) v |= __shfl_xor(v, x, 32);
    if (e == d) mine = v;
  }
  uint32_t* q = o + (l2 ? F6I_Q2 : F6I_Q1) + (grp * 4 + mt) * 768 + lane * 4;
  if (e < 4) q[term * 256 + e] = mine;
  else if (e < 6) q[512 + 2 * term + (e - 4)] = mine;
  if (e == 0) {
    const int byte = (E - 2) + 127 - (term == 0 ? 2 : 13);
    ((uint8_t*)(o + F6I_SC + gg * 128))[lane * 8 + term * 4 + mt] = (uint8_t)(byte < 0 ? 0 : (byte > 254 ? 254 : byte));
  }
}

enum { MODE_SHADE = 0, MODE_APP = 1, MODE_MLP = 2 };

struct ShadeArgs {
  DevCoords c;
  DevField F;
  const float* packed;
  const float* rays;  // MODE_SHADE: [N][6]
  const float* z;     // MODE_SHADE: [N][S]
  const float* coords;  // MODE_SHADE, optional: [N][S][4] = (r^, theta^, phi^, is_yang) from ego_march_density
  const float* c7n;   // MODE_APP: [M][7]
  const float* feat;  // MODE_MLP: [M][27]
  const float* dirs;  // MODE_MLP: [M][3]
  float* out;         // rgb [M][3] or feat [M][27]
  uint16_t* dump_x;   // training forward (MODE_SHADE, f16x3): activations kept for the backward pass, lane-contiguous
  uint16_t* dump_h1;  //   [M][160] X, [M][128] relu(H1), [M][128] relu(H2) as halves [tile][k-step][lane][8]; [M][144] v fp32 (include/egonerf_hip.h)
  uint16_t* dump_h2;
  float* dump_v;
  float* dump_fe;       // [tile][quad 0..3][lane][4]: the lane's 16 feature slots (basis output), fp32: the backward re-derives the encodings from them
  uint32_t* dump_bits;  // [tile][layer 0: h1, 1: h2][lane][2]: bit 16 (mt & 1) + r of word mt >> 1 <=> post-ReLU unit (mt, r) of the lane > 0
  const uint8_t* tile_active;  // optional [n_tiles]: 0 = every weight of the tile is zero, skip it
  // FOLD (ego_shade_composite): compositing in the kernel's epilogue instead of per-sample colours
  const float* comp_w;    // [N][S] weights
  const float* comp_bg;   // [N] background weight (with an envmap)
  const float* envmap;    // emission or null
  int32_t envmap_h;
  float* rgb_map;         // [N][3]
  float* depth;           // [N] or null
  float* bg_map;          // [N][3] or null
  float* env_map;         // [N][3] or null
  int64_t M;
  int32_t S;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// The gather/basis phase runs in 6 stages (plane ST>>1, channel quads 3*(ST&1)..+2 of this lane's 24-channel half):
// 18 x 16-byte loads in flight per lane per stage, 12 products, 12 basis k-steps.
constexpr int STQ = 3;                 // float4 quads per stage
constexpr int STV = STQ * 4;           // products per stage

template <int ST>
__device__ __forceinline__ void gather_stage(const DevField& F, const VMTaps& t, int g, int h, float v[STV]) {
#pragma clang fp contract(fast)  // built with -ffp-contract=off; interpolation may use FMAs
  constexpr int I = ST >> 1, Q0 = STQ * (ST & 1);
  const Lin1 X = t.ax[vm_plane_x(I)], Y = t.ax[vm_plane_y(I)], Ln = t.ax[vm_line_ax(I)];
  const int W = F.res[vm_plane_x(I)];
  const float* P = (g ? F.plane[1][I] : F.plane[0][I]) + 4 * h + 8 * Q0;
  const float* L = (g ? F.line[1][I] : F.line[0][I]) + 4 * h + 8 * Q0;
  const f32x4* p00 = (const f32x4*)(P + (Y.i0 * W + X.i0) * APP_C);
  const f32x4* p01 = (const f32x4*)(P + (Y.i0 * W + X.i1) * APP_C);
  const f32x4* p10 = (const f32x4*)(P + (Y.i1 * W + X.i0) * APP_C);
  const f32x4* p11 = (const f32x4*)(P + (Y.i1 * W + X.i1) * APP_C);
  const f32x4* l0 = (const f32x4*)(L + Ln.i0 * APP_C);
  const f32x4* l1 = (const f32x4*)(L + Ln.i1 * APP_C);
  const float w00 = __fmul_rn(Y.w0, X.w0), w01 = __fmul_rn(Y.w0, X.w1);
  const float w10 = __fmul_rn(Y.w1, X.w0), w11 = __fmul_rn(Y.w1, X.w1);
#pragma unroll
  for (int q = 0; q < STQ; ++q) {
    const f32x4 pv = p00[2 * q] * w00 + p01[2 * q] * w01 + p10[2 * q] * w10 + p11[2 * q] * w11;
    const f32x4 lv = l0[2 * q] * Ln.w0 + l1[2 * q] * Ln.w1;
    const f32x4 m = pv * lv;
    v[q * 4 + 0] = m.x; v[q * 4 + 1] = m.y; v[q * 4 + 2] = m.z; v[q * 4 + 3] = m.w;
  }
}

// basis A fragments of stage ST for grid gsel (streamed from L2, fetched one stage ahead of their MFMAs)
template <int ST>
__device__ __forceinline__ void basis_load(const f32x4* __restrict__ BAS, int lane, int gsel, f32x4 a[STQ]) {
#pragma unroll
  for (int q = 0; q < STQ; ++q) a[q] = BAS[(gsel * (KS_BASIS / 4) + ST * STQ + q) * 64 + lane];
}

__device__ __forceinline__ void basis_mfma(const f32x4 a[STQ], const float v[STV], bool keep, f32x16& fe) {
#pragma unroll
  for (int q = 0; q < STQ; ++q) {
    fe = MFMA(a[q].x, keep ? v[q * 4 + 0] : 0.f, fe);
    fe = MFMA(a[q].y, keep ? v[q * 4 + 1] : 0.f, fe);
    fe = MFMA(a[q].z, keep ? v[q * 4 + 2] : 0.f, fe);
    fe = MFMA(a[q].w, keep ? v[q * 4 + 3] : 0.f, fe);
  }
}

// a wave that straddles the yin/yang border (rare: ~0.8 crossings per 512-sample ray) runs the second weight set
// with the first grid's samples zeroed on the B side (a lane only feeds its own output column)
template <int ST>
__device__ __forceinline__ void basis_mixed(const f32x4* __restrict__ BAS, int lane, int g, const float v[STV], f32x16& fe) {
  f32x4 a[STQ];
  basis_load<ST>(BAS, lane, 1, a);
  basis_mfma(a, v, g != 0, fe);
}

template <int MODE>
__global__ __launch_bounds__(512) void k_shade(ShadeArgs A) {
  __shared__ __attribute__((aligned(16))) float lds[(MODE == MODE_APP ? 0 : LDS_W_FLOATS) + (MODE == MODE_MLP ? 4 : LUT_MAX)];
  float* lut = lds + (MODE == MODE_APP ? 0 : LDS_W_FLOATS);
  if (MODE != MODE_APP) {
    const f32x4* src = (const f32x4*)A.packed;
    f32x4* dst = (f32x4*)lds;
    for (int i = threadIdx.x; i < LDS_W_FLOATS / 4; i += 512) dst[i] = src[i];
  }
  if (MODE == MODE_SHADE)
    for (int i = threadIdx.x; i < A.c.n_lut; i += 512) lut[i] = A.c.r_lut[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  // waves w and w+4 share a SIMD and would otherwise run their gather and MFMA phases in lockstep (equal sharing
  // of the matrix pipe keeps them synchronised); a static priority for one of them staggers the phases
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  const int64_t n_tiles = (A.M + 31) >> 5;
  const f32x4* W1 = (const f32x4*)(lds + OFF_W1);
  const f32x4* W2 = (const f32x4*)(lds + OFF_W2);
  const f32x4* B1 = (const f32x4*)(lds + OFF_B1);
  const f32x4* B2 = (const f32x4*)(lds + OFF_B2);
  const f32x4* W3 = (const f32x4*)(lds + OFF_W3);
  const f32x4* BAS = (const f32x4*)(A.packed + OFF_BASIS);

  for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 8) {
    if (MODE == MODE_SHADE && A.tile_active && !A.tile_active[tile]) continue;
    // LDS weights are loop-invariant: without an opaque per-iteration index LICM hoists all 240
    // ds_read_b128 out of the tile loop and spills them to scratch
    int lw = lane;
    asm volatile("" : "+v"(lw));
    const int hw = lw >> 5;
    const int64_t m_raw = tile * 32 + j;
    const bool valid = m_raw < A.M;
    const int64_t m = valid ? m_raw : A.M - 1;

    f32x16 fe;  // basis output: register r holds feature 2r + h
    float vd0 = 0.f, vd1 = 0.f, vd2 = 0.f;
    if (MODE == MODE_MLP) {
      const float* fp = A.feat + m * APP_DIM;
#pragma unroll
      for (int r = 0; r < 16; ++r) fe[r] = (r < NSLOT && 2 * r + h < APP_DIM) ? fp[min(2 * r + h, APP_DIM - 1)] : 0.f;
      vd0 = A.dirs[m * 3]; vd1 = A.dirs[m * 3 + 1]; vd2 = A.dirs[m * 3 + 2];
    } else {
      float a_r, a_th, a_ph;
      int g;
      if (MODE == MODE_APP) {
        const float* p = A.c7n + m * 7;
        g = (p[6] == 0.f) ? 0 : 1;
        const int b = g ? 3 : 0;
        a_r = p[b]; a_th = p[b + 1]; a_ph = p[b + 2];
      } else {
        const uint32_t ray = (uint32_t)m / (uint32_t)A.S;
        const float* R = A.rays + (int64_t)ray * 6;
        const float zz = A.z[m];
        vd0 = R[3]; vd1 = R[4]; vd2 = R[5];
        const float px = __fadd_rn(R[0], __fmul_rn(vd0, zz)), py = __fadd_rn(R[1], __fmul_rn(vd1, zz)),
                    pz = __fadd_rn(R[2], __fmul_rn(vd2, zz));
        const YinYang y = yinyang_from_xyz(px, py, pz, A.c);
        g = y.yang;
        a_r = normalize_r(y.r, lut, A.c.n_lut, A.c.n_r);
        a_th = normalize_ang(y.th, A.c.th_near, A.c.th_inv);
        a_ph = normalize_ang(y.ph, A.c.ph_near, A.c.ph_inv);
      }
      const bool any_yin = __ballot(g == 0) != 0ull, any_yang = __ballot(g != 0) != 0ull;
      const VMTaps taps = vm_setup(a_r, a_th, a_ph, A.F.res);
#pragma unroll
      for (int r = 0; r < 16; ++r) fe[r] = 0.f;
      // software pipeline over the 6 stages; sched_barriers keep the scheduler from hoisting every load to the top
      {
        const bool mixed = any_yin && any_yang;
        const int gu = any_yin ? 0 : 1;              // weight set of the first pass
        const bool keep = !mixed || g == 0;          // first pass keeps yin lanes only when the wave is mixed
        float vA[STV], vB[STV];
        f32x4 aA[STQ], aB[STQ];
        basis_load<0>(BAS, lane, gu, aA);
        gather_stage<0>(A.F, taps, g, h, vA);
        __builtin_amdgcn_sched_barrier(0);
#define EGO_STAGE(ST, vN, aN, vP, aP)                         \
        basis_load<ST>(BAS, lane, gu, aN);                    \
        gather_stage<ST>(A.F, taps, g, h, vN);                \
        basis_mfma(aP, vP, keep, fe);                         \
        if (mixed) basis_mixed<ST - 1>(BAS, lane, g, vP, fe); \
        __builtin_amdgcn_sched_barrier(0);
        EGO_STAGE(1, vB, aB, vA, aA)
        EGO_STAGE(2, vA, aA, vB, aB)
        EGO_STAGE(3, vB, aB, vA, aA)
        EGO_STAGE(4, vA, aA, vB, aB)
        EGO_STAGE(5, vB, aB, vA, aA)
#undef EGO_STAGE
        basis_mfma(aB, vB, keep, fe);
        if (mixed) basis_mixed<5>(BAS, lane, g, vB, fe);
      }
    }

    if (MODE == MODE_APP) {
      if (valid) {
        float* o = A.out + m * APP_DIM;
#pragma unroll
        for (int r = 0; r < NSLOT; ++r)
          if (2 * r + h < APP_DIM) o[2 * r + h] = fe[r];
      }
      continue;
    }

    // ---- view-direction slots (3 raw + 12 encodings, 8 per lane half) -------------------------------
    float vw[8];
    {
      float sa0, ca0, sb0, cb0, sa1, ca1, sb1, cb1, sa2, ca2, sb2, cb2;
      sincos_f32(vd0, sa0, ca0); sincos_f32(__fmul_rn(vd0, 2.f), sb0, cb0);
      sincos_f32(vd1, sa1, ca1); sincos_f32(__fmul_rn(vd1, 2.f), sb1, cb1);
      sincos_f32(vd2, sa2, ca2); sincos_f32(__fmul_rn(vd2, 2.f), sb2, cb2);
      // half 0: d0 d1 d2 sin(d0) sin(2d0) sin(d1) sin(2d1) sin(d2) | half 1: sin(2d2) cos(d0) cos(2d0) ... cos(2d2) 0
      vw[0] = h ? sb2 : vd0; vw[1] = h ? ca0 : vd1; vw[2] = h ? cb0 : vd2; vw[3] = h ? ca1 : sa0;
      vw[4] = h ? cb1 : sb0; vw[5] = h ? ca2 : sa1; vw[6] = h ? cb2 : sb1; vw[7] = h ? 0.f : sa2;
    }

    // ---- layer 1: 150 -> 128.  K-outer: four accumulators (one per 32-unit M-tile) stay live, the
    // B operand of k-step kk is produced right before it is consumed -------------------------------
    f32x16 H[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = B1[(mt * 2 + hw) * 4 + q];
        H[mt][q * 4 + 0] = b.x; H[mt][q * 4 + 1] = b.y; H[mt][q * 4 + 2] = b.z; H[mt][q * 4 + 3] = b.w;
      }
    {
      f32x4 a[4];
      float s1 = 0.f, c1 = 0.f, s2 = 0.f, c2 = 0.f;
#pragma unroll
      for (int kk = 0; kk < KS1; ++kk) {
        if ((kk & 3) == 0) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) a[mt] = W1[((kk >> 2) * 4 + mt) * 64 + lw];
        }
        float x;
        if (kk < 5 * NSLOT) {
          const int r = kk / 5, kind = kk % 5;
          if (kind == 0) {
            sincos_f32(fe[r], s1, c1);
            sincos_f32(__fmul_rn(fe[r], 2.f), s2, c2);
          }
          x = kind == 0 ? fe[r] : (kind == 1 ? s1 : (kind == 2 ? s2 : (kind == 3 ? c1 : c2)));
        } else if (kk < 5 * NSLOT + 8) {
          x = vw[kk - 5 * NSLOT];
        } else {
          x = 0.f;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) H[mt] = MFMA(a[mt][kk & 3], x, H[mt]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) H[mt][r] = relu_f(H[mt][r]);

    // ---- layer 2: 128 -> 128 (K-outer again), then layer 3: 128 -> 3 on the VALU -----------------------
    f32x16 G[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = B2[(mt * 2 + hw) * 4 + q];
        G[mt][q * 4 + 0] = b.x; G[mt][q * 4 + 1] = b.y; G[mt][q * 4 + 2] = b.z; G[mt][q * 4 + 3] = b.w;
      }
    {
      f32x4 a[4];
#pragma unroll
      for (int kk = 0; kk < KS2; ++kk) {
        if ((kk & 3) == 0) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) a[mt] = W2[((kk >> 2) * 4 + mt) * 64 + lw];
        }
        const float x = H[kk >> 4][kk & 15];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) G[mt] = MFMA(a[mt][kk & 3], x, G[mt]);
      }
    }
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const f32x4 w = W3[(mt * 2 + hw) * 16 + r];
        const float hv = relu_f(G[mt][r]);
        o0 = fmaf(hv, w.x, o0); o1 = fmaf(hv, w.y, o1); o2 = fmaf(hv, w.z, o2);
      }
    o0 += __shfl_xor(o0, 32, 64);
    o1 += __shfl_xor(o1, 32, 64);
    o2 += __shfl_xor(o2, 32, 64);
    if (valid && h == 0) {
      const float* b3 = lds + OFF_B3;
      float* o = A.out + m * 3;
      o[0] = sigmoidf(o0 + b3[0]);
      o[1] = sigmoidf(o1 + b3[1]);
      o[2] = sigmoidf(o2 + b3[2]);
    }
  }
}


// =================================================================================================================
// f16x3 variant of k_shade: same dataflow, but every matrix product runs as three v_mfma_f32_32x32x16_f16
// (w_hi*x_hi + w_lo*x_hi + w_hi*x_lo, fp32 accumulate; the dropped lo*lo term and the 22-bit operands leave ~2^-21
// relative error, i.e. fp32-grade: 2.4e-7 max |dRGB| vs 1.4e-7 for the fp32 MFMA path on the oracle's inputs).
// fp16 subnormals are honoured by the instruction on gfx950 (tools/mfma_probe.hip), which the small x_lo terms need.
// 6 matrix-pipe cycles per K instead of 32.
// =================================================================================================================
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define MFMAH(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct HL {
  h8 hi, lo;
};

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// bit 16 (mt & 1) + r of word mt >> 1 is set iff unit (mt, r) of this lane is > 0 (relu'(0) = 0 like torch's threshold backward)
__device__ __forceinline__ u32x2 relu_bits(const f32x16 (&H)[4]) {
  u32x2 b = {0u, 0u};
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) b[mt >> 1] |= (H[mt][r] > 0.f ? 1u : 0u) << (16 * (mt & 1) + r);
  return b;
}

__device__ __forceinline__ void split_pair(float a, float b, bool keep, uint32_t& hi, uint32_t& lo) {
  a = keep ? a : 0.f;
  b = keep ? b : 0.f;
  const auto hp = __builtin_amdgcn_cvt_pkrtz(a, b);
  hi = __builtin_bit_cast(uint32_t, hp);
  // residuals a - hi, b - hi are exact; each is ONE v_fma_mix_f32 (hi * -1 + a) reading the half straight from the packed
  // register: no v_cvt_f32_f16 back-conversion (VALU and MFMA time add up on this SIMD, tools/coissue_probe.hip).  The -1 is
  // made opaque so that the fma survives to instruction selection (a literal -1 folds into convert + subtract); the
  // instruction should come from the compiler rather than from inline asm, which its hazard recogniser cannot see into.
  float neg1 = -1.0f;
  asm("" : "+v"(neg1));
  const float ra = __builtin_fmaf((float)hp[0], neg1, a), rb = __builtin_fmaf((float)hp[1], neg1, b);
  lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}

__device__ __forceinline__ HL split8(const float x[8], bool keep) {
  u32x4 hi, lo;
  uint32_t a, b;
  split_pair(x[0], x[1], keep, a, b); hi.x = a; lo.x = b;
  split_pair(x[2], x[3], keep, a, b); hi.y = a; lo.y = b;
  split_pair(x[4], x[5], keep, a, b); hi.z = a; lo.z = b;
  split_pair(x[6], x[7], keep, a, b); hi.w = a; lo.w = b;
  HL o;
  o.hi = __builtin_bit_cast(h8, hi);
  o.lo = __builtin_bit_cast(h8, lo);
  return o;
}

// ---- f16f8 arithmetic: main term in fp16, both correction terms in one block-scaled fp8 MFMA per pair of k-steps -------------
typedef int v8i __attribute__((ext_vector_type(8)));
typedef short v2s __attribute__((ext_vector_type(2)));
#define MFMA8(a, b, c) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((a), (b), (c), 0, 0, 0, 116 /* A block scale 2^-11 */, 0, 127)

// 8 values of one k-step -> fp16 main operand (returned) + this step's 16 bytes of the pair's fp8 operand b8:
// bytes [8 odd .. 8 odd + 7] = e4m3(x), bytes [16 + 8 odd .. ] = e4m3((x - fp16(x)) * 2^11)
__device__ __forceinline__ h8 split8_f8(const float x[8], v8i& b8, int odd) {
  u32x4 hi;
  uint32_t xh[2], xl[2];
  float neg1 = -1.0f;
  asm("" : "+v"(neg1));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = x[2 * q], b = x[2 * q + 1];
    // round-to-nearest fp16 (v_cvt_pk_f16_f32) rather than the truncating v_cvt_pkrtz of the three-term split: it halves the
    // residual, and with it the error the 4-bit fp8 copy of the residual leaves
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    const h2v hp = __builtin_convertvector(f2v{a, b}, h2v);
    hi[q] = __builtin_bit_cast(uint32_t, hp);
    const float ra = __builtin_fmaf((float)hp[0], neg1, a), rb = __builtin_fmaf((float)hp[1], neg1, b);
    if ((q & 1) == 0) {
      // the low-half conversion keeps the other half of its destination, which the high-half conversion of the next value pair
      // overwrites: seed the destination with a value that is about to die (the operands themselves) rather than with a zero,
      // which costs a v_mov per operand register
      xh[q >> 1] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, __builtin_bit_cast(int, b), false);
      xl[q >> 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(v2s, rb), ra, rb, 1.0f / 2048.0f, false));
    } else {
      xh[q >> 1] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)xh[q >> 1], true);
      xl[q >> 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(v2s, xl[q >> 1]), ra, rb,
                                                                                         1.0f / 2048.0f, true));
    }
  }
  b8[2 * odd] = (int)xh[0]; b8[2 * odd + 1] = (int)xh[1];
  b8[4 + 2 * odd] = (int)xl[0]; b8[4 + 2 * odd + 1] = (int)xl[1];
  return __builtin_bit_cast(h8, hi);
}

__device__ __forceinline__ v8i load_a8(const u32x4* __restrict__ WF, int pair, int mt, int lane) {
  const u32x4 p0 = WF[((pair * 4 + mt) * 2 + 0) * 64 + lane], p1 = WF[((pair * 4 + mt) * 2 + 1) * 64 + lane];
  return v8i{(int)p0.x, (int)p0.y, (int)p0.z, (int)p0.w, (int)p1.x, (int)p1.y, (int)p1.z, (int)p1.w};
}

// ---- f16f6 arithmetic (layout: k_pack_mlp_f6) ------------------------------------------------------------------------------------------
typedef int v16i __attribute__((ext_vector_type(16)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
#define MFMA6(a, b, c, sa, opa, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((a), (b), (c), 2, 2, (opa), (sa), 0, (sb))

// one k-step's 8 values -> packed fp16 main operand (also written to dwords 4 s .. 4 s + 3 of the group's hp), the 8 exact residuals
// x - fp16(x), and the running maximum of |x| over the group
// mmask (a constant after unrolling): bit e set = value e takes part in the maximum (layer 1 knows that its sines and cosines are bounded by 1)
__device__ __forceinline__ h8 split8_f6(const float x[8], uint32_t* hp, float* res, float& amax, int mmask = 0xff) {
  u32x4 hi;
  float neg1 = -1.0f;
  asm("" : "+v"(neg1));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = x[2 * q], b = x[2 * q + 1];
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    const h2v hpv = __builtin_convertvector(f2v{a, b}, h2v);
    hi[q] = __builtin_bit_cast(uint32_t, hpv);
    hp[q] = hi[q];
    res[2 * q] = __builtin_fmaf((float)hpv[0], neg1, a);
    res[2 * q + 1] = __builtin_fmaf((float)hpv[1], neg1, b);
    const bool ma = (mmask >> (2 * q)) & 1, mb = (mmask >> (2 * q + 1)) & 1;
    if (ma && mb) amax = fmaxf(fmaxf(fabsf(a), fabsf(b)), amax);
    else if (ma) amax = fmaxf(fabsf(a), amax);
    else if (mb) amax = fmaxf(fabsf(b), amax);
  }
  return __builtin_bit_cast(h8, hi);
}

// The 32-value conversions run in passes and write their destination registers while later passes still read the scale and the
// sources: with the scale, or any but the first registers of a source tuple, in a destination register the result is wrong
// (tools/fp6_overlap_probe.hip, profiles/r04/fp6_overlap_probe.txt) - and this compiler's register allocator produces exactly such
// overlaps for the builtins (their definitions carry no early-clobber; keeping the operands live past the builtin does not help,
// the allocator copies tuples).  So the two conversions are issued through inline asm with an early-clobber destination.  What the
// compiler's hazard recogniser cannot see into is covered inside the block: one wait state before (a VALU result feeding the
// conversion) and two after it (conversion result -> MFMA operand needs one: tools/fp6_raw_probe.hip).
typedef uint32_t u32x6 __attribute__((ext_vector_type(6)));
__device__ __forceinline__ u32x6 cvt_pk32_fp6_f16(const v16i& halves, float scale) {
  u32x6 r;
  asm("s_nop 0\n\tv_cvt_scalef32_pk32_fp6_f16 %0, %1, %2\n\ts_nop 1" : "=&v"(r) : "v"(halves), "v"(scale));
  return r;
}
__device__ __forceinline__ u32x6 cvt_2xpk16_fp6_f32_dup(const f32x16& a, float scale) {   // both sources = a
  u32x6 r;
  asm("s_nop 0\n\tv_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %1, %2\n\ts_nop 1" : "=&v"(r) : "v"(a), "v"(scale));
  return r;
}
__device__ __forceinline__ u32x6 cvt_2xpk16_fp6_f32(const f32x16& a, const f32x16& b, float scale) {
  u32x6 r;
  asm("s_nop 0\n\tv_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3\n\ts_nop 1" : "=&v"(r) : "v"(a), "v"(b), "v"(scale));
  return r;
}

// 8 fp32 values -> 8 halves, round to nearest: one k-step of an activation dump (ego_shade_dump's x / h1 / h2)
__device__ __forceinline__ u32x4 pack8_rn(const float x[8]) {
  u32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    o[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2v{x[2 * q], x[2 * q + 1]}, h2v));
  }
  return o;
}

struct B6 {
  v8i x, r;   // fp6 operands of the two correction terms (dwords 6, 7 unused)
  int sb;     // B's block scale byte (biased exponent of the group's largest |x|)
};

__device__ __forceinline__ void f6_scales(float amax, int& eb, float& sx, float& sr) {
  uint32_t e = __float_as_uint(amax) >> 23;
  e = e < 14u ? 14u : e;
  eb = (int)e;
  sx = __uint_as_float((e - 2u) << 23);
  sr = __uint_as_float((e - 13u) << 23);
}

// full group: hp = the 16 packed-half dwords of its 32 values, res = their 32 residuals in value order
__device__ __forceinline__ B6 group6(const uint32_t hp[16], const float res[32], float amax) {
  B6 o;
  float sx, sr;
  f6_scales(amax, o.sb, sx, sr);
  v16i hv;
  f32x16 ra, rb;
#pragma unroll
  for (int i = 0; i < 16; ++i) { hv[i] = (int)hp[i]; ra[i] = res[i]; rb[i] = res[16 + i]; }
  const u32x6 x6 = cvt_pk32_fp6_f16(hv, sx);
  const u32x6 r6 = cvt_2xpk16_fp6_f32(ra, rb, sr);
  o.x = __builtin_bit_cast(v8i, __builtin_shufflevector(x6, x6, 0, 1, 2, 3, 4, 5, -1, -1));
  o.r = __builtin_bit_cast(v8i, __builtin_shufflevector(r6, r6, 0, 1, 2, 3, 4, 5, -1, -1));
  return o;
}

// half group (layer 1's last: 16 values): both operands through the interleaving conversion with the values as BOTH sources - the odd
// elements meet zero weights (f6_value), and no tuple of zeros has to be materialised
__device__ __forceinline__ B6 group6_half(const float xv[16], const float res[16], float amax) {
  B6 o;
  float sx, sr;
  f6_scales(amax, o.sb, sx, sr);
  f32x16 xa, ra;
#pragma unroll
  for (int i = 0; i < 16; ++i) { xa[i] = xv[i]; ra[i] = res[i]; }
  const u32x6 x6 = cvt_2xpk16_fp6_f32_dup(xa, sx);
  const u32x6 r6 = cvt_2xpk16_fp6_f32_dup(ra, sr);
  o.x = __builtin_bit_cast(v8i, __builtin_shufflevector(x6, x6, 0, 1, 2, 3, 4, 5, -1, -1));
  o.r = __builtin_bit_cast(v8i, __builtin_shufflevector(r6, r6, 0, 1, 2, 3, 4, 5, -1, -1));
  return o;
}

// Per-lane LDS byte offsets into the f16f6 image (opaque to the compiler, so that a read is `base + 16-bit immediate` and not an
// address computation of its own): one base per 64 KB window of the 16-byte-stride part, one for the scale bytes
struct F6Bases {
  uint32_t o16[3], o8s;   // lane * 16 + 64 KB * w, lane * 8 + 4 F6I_SC
};
__device__ __forceinline__ F6Bases f6_bases(int lane) {
  F6Bases b;
  b.o16[0] = (uint32_t)lane * 16u; b.o16[1] = (uint32_t)lane * 16u + 65536u; b.o16[2] = (uint32_t)lane * 16u + 131072u;
  b.o8s = (uint32_t)lane * 8u + 4u * F6I_SC;
  asm volatile("" : "+v"(b.o16[0]), "+v"(b.o16[1]), "+v"(b.o16[2]), "+v"(b.o8s));
  return b;
}
// slot: 32-bit slot index of lane 0's element (a constant after unrolling)
__device__ __forceinline__ u32x4 f6_ld128(const float* lds, const F6Bases& b, int slot) {
  const int w = (slot * 4) >> 16;
  return *(const u32x4*)((const char*)lds + (slot * 4 - 65536 * w) + b.o16[w]);
}
__device__ __forceinline__ h8 f6_hi(const float* lds, const F6Bases& b, bool l2, int step, int mt) {
  return __builtin_bit_cast(h8, f6_ld128(lds, b, (l2 ? F6I_HI2 : F6I_HI1) + (step * 4 + mt) * 256));
}

// dwords 6, 7 of the builtin's operand type are not read by the fp6 form of the instruction and stay undefined
__device__ __forceinline__ v8i a6_operand(const u32x4& head, const u32x4& tails, int term) {
  return term == 0 ? __builtin_bit_cast(v8i, __builtin_shufflevector(head, tails, 0, 1, 2, 3, 4, 5, -1, -1))
                   : __builtin_bit_cast(v8i, __builtin_shufflevector(head, tails, 0, 1, 2, 3, 6, 7, -1, -1));
}

__device__ __forceinline__ void group_mfma6(const float* lds, const F6Bases& fb, bool l2, int grp, const B6& b, f32x16 (&H)[4]) {
  const u32x2 sa = *(const u32x2*)((const char*)lds + ((l2 ? G6_1 : 0) + grp) * 512 + fb.o8s);   // byte mt of .x: term 0's block scales, of .y: term 1's
  const int q = (l2 ? F6I_Q2 : F6I_Q1) + grp * 4 * 768;
  u32x4 hd[4], tl[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) { hd[mt] = f6_ld128(lds, fb, q + mt * 768); tl[mt] = f6_ld128(lds, fb, q + mt * 768 + 512); }
  H[0] = MFMA6(a6_operand(hd[0], tl[0], 0), b.x, H[0], (int)sa.x, 0, b.sb);
  H[1] = MFMA6(a6_operand(hd[1], tl[1], 0), b.x, H[1], (int)sa.x, 1, b.sb);
  H[2] = MFMA6(a6_operand(hd[2], tl[2], 0), b.x, H[2], (int)sa.x, 2, b.sb);
  H[3] = MFMA6(a6_operand(hd[3], tl[3], 0), b.x, H[3], (int)sa.x, 3, b.sb);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) hd[mt] = f6_ld128(lds, fb, q + mt * 768 + 256);
  H[0] = MFMA6(a6_operand(hd[0], tl[0], 1), b.r, H[0], (int)sa.y, 0, b.sb);
  H[1] = MFMA6(a6_operand(hd[1], tl[1], 1), b.r, H[1], (int)sa.y, 1, b.sb);
  H[2] = MFMA6(a6_operand(hd[2], tl[2], 1), b.r, H[2], (int)sa.y, 2, b.sb);
  H[3] = MFMA6(a6_operand(hd[3], tl[3], 1), b.r, H[3], (int)sa.y, 3, b.sb);
}

struct BasisFrag {

  h8 hi, lo;
};

template <int STEP>
__device__ __forceinline__ BasisFrag basis_frag(const u32x4* __restrict__ BASH, int lane, int gsel) {
  BasisFrag f;
  f.hi = __builtin_bit_cast(h8, BASH[((gsel * KHB + STEP) * 2 + 0) * 64 + lane]);
  f.lo = __builtin_bit_cast(h8, BASH[((gsel * KHB + STEP) * 2 + 1) * 64 + lane]);
  return f;
}

__device__ __forceinline__ void basis_step(const BasisFrag& a, const float x[8], bool keep, f32x16& fe) {
  const HL b = split8(x, keep);
  fe = MFMAH(a.hi, b.hi, fe);
  fe = MFMAH(a.lo, b.hi, fe);
  fe = MFMAH(a.hi, b.lo, fe);
}

// ---- 4-lane-team gather (fp32 tables, f16x3 kernel) ------------------------------------------------------------------
// The vector L1 looks up one distinct 64-byte line per cycle, and with two lanes per sample a load instruction touches
// ~26 distinct lines (PMC: 3270 tag lookups per tile), which - not bytes - bounds the gather.  Here the wave gathers as 16
// teams of 4 lanes, lane = 32*pa + 2*t + pb: team t = the lane pairs (2t, 2t+1) and (2t+32, 2t+33).  In round rd the team
// serves the tile's sample 2t + rd and its lane reads quad pa + 2*(pb ^ rd) of each line, so a team reads a whole line per
// instruction and an instruction touches <= 16 lines; two rounds cover the 32 samples.  The MFMA layout is lane = 32*h + j
// (sample j, half h = pa holds quads h and h + 2, K order app_channel_g): a lane's own sample is 2t + pb, so it keeps the
// round that served it (quad h) and gets quad h + 2 of its sample from the lane next to it: two selects and one neighbour
// exchange per value.
// The exchange is a v_mov_b32_dpp quad_perm (__shfl_xor and ds_swizzle forms were measured in round 1 and removed: DESIGN.md 5.1).
// REPRODUCIBILITY: kernels of this family (fp32-table gather of the f16x3 kernel) have a build-dependent fault that is not
// understood: in affected builds 0.3 % ... 100 % of the calls return one gather round's products wrong in lanes
// 16-31 and 48-63 of a wave in slots 4-7 of its workgroup (the second wave of a SIMD).  It does not depend on the exchange (it also
// hits EGO_GATHER_TEAMS 0, where no lane talks to another), needs two waves co-running on the SIMD, and none of the stand-alone
// probes in tools/ reproduces it.  Builds of this file WITHOUT the SLP vectoriser (egonerf_amd/build.py) have been clean in every
// soak (> 300 000 calls, all exchange forms tried); with it, only the __shfl_xor form happened to be.  tools/flaky_probe*.py,
// tools/variant_test.sh and tests/test_hip_determinism.py are the guards: re-run them after ANY change to this file.

struct TeamSample {  // normalised coordinates of the sample this lane's team serves in one round
  float a_r, a_th, a_ph;
  int g;
};

template <int I>
__device__ __forceinline__ void team_load(const DevField& F, const VMTaps& t, int g, int q, f32x4 raw[18]) {
  const Lin1 X = t.ax[vm_plane_x(I)], Y = t.ax[vm_plane_y(I)], Ln = t.ax[vm_line_ax(I)];
  const int W = F.res[vm_plane_x(I)];
  // compact addressing (DevField): one scalar base + a 32-bit byte offset per tap; quad 4i + q is 64 i bytes further on
  const uint32_t pb = (g ? F.poff[1][I] : F.poff[0][I]) + 16u * (uint32_t)q;
  const uint32_t lb = (g ? F.loff[1][I] : F.loff[0][I]) + 16u * (uint32_t)q;
  const uint32_t r0 = pb + (uint32_t)(Y.i0 * W) * (APP_C * 4), r1 = pb + (uint32_t)(Y.i1 * W) * (APP_C * 4);
  const uint32_t c0 = (uint32_t)X.i0 * (APP_C * 4), c1 = (uint32_t)X.i1 * (APP_C * 4);
  const char* p00 = F.base + (r0 + c0);
  const char* p01 = F.base + (r0 + c1);
  const char* p10 = F.base + (r1 + c0);
  const char* p11 = F.base + (r1 + c1);
  const char* l0 = F.base + (lb + (uint32_t)Ln.i0 * (APP_C * 4));
  const char* l1 = F.base + (lb + (uint32_t)Ln.i1 * (APP_C * 4));
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    raw[i] = *(const f32x4*)(p00 + 64 * i); raw[3 + i] = *(const f32x4*)(p01 + 64 * i); raw[6 + i] = *(const f32x4*)(p10 + 64 * i);
    raw[9 + i] = *(const f32x4*)(p11 + 64 * i); raw[12 + i] = *(const f32x4*)(l0 + 64 * i); raw[15 + i] = *(const f32x4*)(l1 + 64 * i);
  }
}

template <int I>
__device__ __forceinline__ void team_finish(const VMTaps& t, const f32x4 raw[18], float out[12]) {
#pragma clang fp contract(fast)
  const Lin1 X = t.ax[vm_plane_x(I)], Y = t.ax[vm_plane_y(I)], Ln = t.ax[vm_line_ax(I)];
  float w00 = __fmul_rn(Y.w0, X.w0), w01 = __fmul_rn(Y.w0, X.w1);
  float w10 = __fmul_rn(Y.w1, X.w0), w11 = __fmul_rn(Y.w1, X.w1);
  float l0 = Ln.w0, l1 = Ln.w1;
#ifndef EGO_PAIRED_WEIGHTS
  // Every weight in a register of its own: a packed multiply then broadcasts the LOW half of its pair operand
  // (op_sel_hi:[1,0]).  If the compiler vectorises the four products above into {w00, w01} / {w10, w11} pairs, the packed ops
  // that consume w01 / w11 broadcast the HIGH half (op_sel:[1,0,0]) - and every build of these kernels that contained such
  // instructions was non-reproducible, every build without them clean.  tests/test_abi_and_host.py checks the ISA.
  asm volatile("" : "+v"(w00)); asm volatile("" : "+v"(w01)); asm volatile("" : "+v"(w10)); asm volatile("" : "+v"(w11));
  asm volatile("" : "+v"(l0)); asm volatile("" : "+v"(l1));
#endif
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const f32x4 pv = raw[i] * w00 + raw[3 + i] * w01 + raw[6 + i] * w10 + raw[9 + i] * w11;
    const f32x4 lv = raw[12 + i] * l0 + raw[15 + i] * l1;
    const f32x4 m = pv * lv;
    out[4 * i + 0] = m.x; out[4 * i + 1] = m.y; out[4 * i + 2] = m.z; out[4 * i + 3] = m.w;
  }
}

// ---- rolling form of the same gather (gather_basis_team<ROLL = true>): the 18-register load buffer is recycled line by line -----
// A round's 18 loads are three lines of six taps.  Instead of "issue 18, wait, compute 12 products, issue the next 18", the six
// registers of a line are refilled with the NEXT round's loads as soon as that line's four products are computed, so 12-18 loads
// stay in flight all the time instead of the queue draining once per round (the gather is latency-bound, not issue-bound).
struct TapPtrs {
  uint32_t p00, p01, p10, p11, l0, l1;   // byte offsets from DevField::base (compact addressing: half the registers of pointers)
  float w00, w01, w10, w11, wl0, wl1;
};

template <int I>
__device__ __forceinline__ TapPtrs tap_ptrs(const DevField& F, const VMTaps& t, int g, int q) {
  const Lin1 X = t.ax[vm_plane_x(I)], Y = t.ax[vm_plane_y(I)], Ln = t.ax[vm_line_ax(I)];
  const int W = F.res[vm_plane_x(I)];
  const uint32_t pb = (g ? F.poff[1][I] : F.poff[0][I]) + 16u * (uint32_t)q;
  const uint32_t lb = (g ? F.loff[1][I] : F.loff[0][I]) + 16u * (uint32_t)q;
  const uint32_t r0 = pb + (uint32_t)(Y.i0 * W) * (APP_C * 4), r1 = pb + (uint32_t)(Y.i1 * W) * (APP_C * 4);
  const uint32_t c0 = (uint32_t)X.i0 * (APP_C * 4), c1 = (uint32_t)X.i1 * (APP_C * 4);
  TapPtrs tp;
  tp.p00 = r0 + c0; tp.p01 = r0 + c1; tp.p10 = r1 + c0; tp.p11 = r1 + c1;
  tp.l0 = lb + (uint32_t)Ln.i0 * (APP_C * 4);
  tp.l1 = lb + (uint32_t)Ln.i1 * (APP_C * 4);
  tp.w00 = __fmul_rn(Y.w0, X.w0); tp.w01 = __fmul_rn(Y.w0, X.w1);
  tp.w10 = __fmul_rn(Y.w1, X.w0); tp.w11 = __fmul_rn(Y.w1, X.w1);
  tp.wl0 = Ln.w0; tp.wl1 = Ln.w1;
  // separate registers for the weights: see team_finish
  asm volatile("" : "+v"(tp.w00)); asm volatile("" : "+v"(tp.w01)); asm volatile("" : "+v"(tp.w10)); asm volatile("" : "+v"(tp.w11));
  asm volatile("" : "+v"(tp.wl0)); asm volatile("" : "+v"(tp.wl1));
  return tp;
}

__device__ __forceinline__ void line_load(const DevField& F, const TapPtrs& tp, int i, f32x4 r[6]) {
  const char* b = F.base + 64 * i;
  r[0] = *(const f32x4*)(b + tp.p00); r[1] = *(const f32x4*)(b + tp.p01); r[2] = *(const f32x4*)(b + tp.p10);
  r[3] = *(const f32x4*)(b + tp.p11); r[4] = *(const f32x4*)(b + tp.l0); r[5] = *(const f32x4*)(b + tp.l1);
}

__device__ __forceinline__ void line_finish(const TapPtrs& tp, const f32x4 r[6], float out[4]) {
#pragma clang fp contract(fast)
  const f32x4 pv = r[0] * tp.w00 + r[1] * tp.w01 + r[2] * tp.w10 + r[3] * tp.w11;
  const f32x4 lv = r[4] * tp.wl0 + r[5] * tp.wl1;
  const f32x4 m = pv * lv;
  out[0] = m.x; out[1] = m.y; out[2] = m.z; out[3] = m.w;
}

// ga = this lane's quad of the sample served in round 0 (tile column 2t), gb = of round 1 (column 2t + 1).  Even lanes own
// column 2t: they keep ga (quad h) and take the odd neighbour's ga (quad h + 2); odd lanes keep gb and take the even
// neighbour's gb.  v[0..11] = quad h, v[12..23] = quad h + 2 of the lane's own sample.
__device__ __forceinline__ void team_to_halves(const float ga[12], const float gb[12], float* v) {
  if (!EGO_GATHER_TEAMS) {
#pragma unroll
    for (int idx = 0; idx < 12; ++idx) { v[idx] = ga[idx]; v[12 + idx] = gb[idx]; }
    return;
  }
  const bool even = (threadIdx.x & 1) == 0;
#pragma unroll
  for (int idx = 0; idx < 12; ++idx) {
    const float give = even ? gb[idx] : ga[idx];
    v[idx] = even ? ga[idx] : gb[idx];
    v[12 + idx] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(give), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false));
  }
}

// (v as halves - 0.6 GB less per step each way, forward -0.12 ms - puts the basis gradient of the 1.5 k-sample golden at 2.3e-4 of its
// largest element against the 2e-4 it is held to: v stays fp32)
// All dump stores of the training FORWARD are NONTEMPORAL (`global_store ... nt`): 2.6 GB per launch that nobody reads before the whole pass
// is over (the backward's dh2 / dh1 are read by the next kernels and stay cached; its dfe / dv are written as quarter lines).  Measured (round 5, timing-only builds, one box): forward without its dump stores 0.49 ms, with them 0.94, with the
// stores aimed at a per-wave 40 KB region (no HBM traffic at all) 0.74 - the write path into the XCD's L2 (~17 B / clk / CU, half the read
// rate), not HBM and not the order of loads and stores in the vmcnt queue (moving the stores behind the next loads: +-0), is what a dumping
// wave waits for; nt stores take the forward to 0.79-0.82 ms.
__device__ __forceinline__ void dump24(float* dst, const float* v) {
  if (dst) {
#pragma unroll
    for (int q = 0; q < 6; ++q) __builtin_nontemporal_store(f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}, &((f32x4*)dst)[64 * q]);  // quad pairs 256 floats apart
  }
}

// whole gather + basis for one tile; ts[rd] = the sample this lane's team serves in round rd.  All 64 lanes must call
// (cross-lane swaps).
// one basis k-step of 8 products: split once, three MFMAs with the wave's grid (g0) and - only in a wave that straddles the yin /
// yang border - three more with the other grid's fragments into the second accumulator (the split is shared: re-splitting in the
// rare branch made the compiler hoist 72 fp16 -> fp32 conversions per tile onto the common path)
template <int STEP>
__device__ __forceinline__ void basis_step2(const u32x4* __restrict__ BASH, int lane, const BasisFrag& a, bool mixed, const float x[8],
                                            f32x16& fe, f32x16& fe2) {
  const HL b = split8(x, true);
  fe = MFMAH(a.hi, b.hi, fe);
  fe = MFMAH(a.lo, b.hi, fe);
  fe = MFMAH(a.hi, b.lo, fe);
  if (mixed) {  // wave-uniform
    const BasisFrag a1 = basis_frag<STEP>(BASH, lane, 1);
    fe2 = MFMAH(a1.hi, b.hi, fe2);
    fe2 = MFMAH(a1.lo, b.hi, fe2);
    fe2 = MFMAH(a1.hi, b.lo, fe2);
  }
}

// the same with the wave's three fragment pairs already loaded (see EGO_ROLL_PLANE)
template <int STEP0>
__device__ __forceinline__ void basis3_pre(const u32x4* __restrict__ BASH, int lane, const BasisFrag& f0, const BasisFrag& f1, const BasisFrag& f2,
                                           bool mixed, const float* v, f32x16& fe, f32x16& fe2) {
  basis_step2<STEP0>(BASH, lane, f0, mixed, v, fe, fe2);
  basis_step2<STEP0 + 1>(BASH, lane, f1, mixed, v + 8, fe, fe2);
  basis_step2<STEP0 + 2>(BASH, lane, f2, mixed, v + 16, fe, fe2);
}

template <int STEP0>
__device__ __forceinline__ void basis3(const u32x4* __restrict__ BASH, int lane, int g0, bool mixed, const float* v, f32x16& fe, f32x16& fe2) {
  const BasisFrag f0 = basis_frag<STEP0>(BASH, lane, g0), f1 = basis_frag<STEP0 + 1>(BASH, lane, g0), f2 = basis_frag<STEP0 + 2>(BASH, lane, g0);
  basis_step2<STEP0>(BASH, lane, f0, mixed, v, fe, fe2);
  basis_step2<STEP0 + 1>(BASH, lane, f1, mixed, v + 8, fe, fe2);
  basis_step2<STEP0 + 2>(BASH, lane, f2, mixed, v + 16, fe, fe2);
}

// ROLL: the rolling form of the load buffer (tap_ptrs / line_load / line_finish above); it needs all 256 registers, so only the
// fused inference kernel uses it (the dumping and stand-alone instantiations would spill)
template <bool ROLL, int HN = 0>
__device__ __forceinline__ void gather_basis_team(const DevField& F, const TeamSample ts[2], const u32x4* __restrict__ BASH, int lane,
                                                  int g, bool mixed, int gu, f32x16& fe, float* vdump) {
  // EGO_GATHER_TEAMS 1: 4-lane teams (above).  0: no teams - both rounds serve the lane's OWN sample (ts[0] == ts[1]), round 0
  // reads quad h and round 1 quad h + 2, nothing is exchanged between lanes; an instruction then touches 32 lines instead of 16.
  const int qa = EGO_GATHER_TEAMS ? (lane >> 5) + 2 * (lane & 1) : (lane >> 5);
  const int qb = EGO_GATHER_TEAMS ? (lane >> 5) + 2 * ((lane & 1) ^ 1) : (lane >> 5) + 2;
  // A lane's output column depends only on its own inputs, so a wave that straddles the yin/yang border (rare) runs the
  // steps with both weight sets on the unmasked products, into two accumulators, and each lane keeps its grid's one at the
  // end: no per-value masking anywhere, and nothing extra for the waves of one grid.
  const int g0 = mixed ? 0 : gu;
  f32x16 fe2;
#pragma unroll
  for (int r = 0; r < 16; ++r) fe2[r] = 0.f;
  const VMTaps tA = vm_setup(ts[0].a_r, ts[0].a_th, ts[0].a_ph, F.res), tB = vm_setup(ts[1].a_r, ts[1].a_th, ts[1].a_ph, F.res);
  // one 18-load buffer in flight (72 VGPRs); the next plane's first round is issued right before the MFMAs of the
  // previous plane so that they overlap; the co-resident wave and the MLP phase hide the rest
  f32x4 raw[18];
  float ga[12], gb[12], v[24];
  if (ROLL) {
    f32x4 (*ln)[6] = (f32x4 (*)[6])raw;  // three lines of six taps
    TapPtrs pa = tap_ptrs<0>(F, tA, ts[0].g, qa), pb = tap_ptrs<0>(F, tB, ts[1].g, qb);
#pragma unroll
    for (int i = 0; i < 3; ++i) line_load(F, pa, i, ln[i]);
#define EGO_ROLL_PLANE(PL, NEXT_A, STEP0, DUMPPTR)                                                        \
    {                                                                                                     \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                     \
        line_finish(pa, ln[i], ga + 4 * i);                                                               \
        line_load(F, pb, i, ln[i]);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                \
      }                                                                                                   \
      NEXT_A;                                                                                             \
      /* the plane's basis fragments BEFORE the next plane's tap loads: vmcnt retires in order, so fragments loaded behind them */ \
      /* would make the first basis MFMA wait for all 18 taps of the next plane */                        \
      BasisFrag bf0, bf1, bf2;                                                                            \
      if (HN >= 1) bf0 = basis_frag<STEP0>(BASH, lane, g0);                                               \
      if (HN >= 2) bf1 = basis_frag<STEP0 + 1>(BASH, lane, g0);                                           \
      if (HN >= 3) bf2 = basis_frag<STEP0 + 2>(BASH, lane, g0);                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                     \
        line_finish(pb, ln[i], gb + 4 * i);                                                               \
        if (PL < 2) line_load(F, pa, i, ln[i]);                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                \
      }                                                                                                   \
      team_to_halves(ga, gb, v);                                                                          \
      dump24(DUMPPTR, v);                                                                                 \
      if (HN < 1) bf0 = basis_frag<STEP0>(BASH, lane, g0);                                                \
      if (HN < 2) bf1 = basis_frag<STEP0 + 1>(BASH, lane, g0);                                            \
      if (HN < 3) bf2 = basis_frag<STEP0 + 2>(BASH, lane, g0);                                            \
      basis3_pre<STEP0>(BASH, lane, bf0, bf1, bf2, mixed, v, fe, fe2);                                    \
    }
    EGO_ROLL_PLANE(0, pa = tap_ptrs<1>(F, tA, ts[0].g, qa), 0, vdump)
    pb = tap_ptrs<1>(F, tB, ts[1].g, qb);
    EGO_ROLL_PLANE(1, pa = tap_ptrs<2>(F, tA, ts[0].g, qa), 3, (vdump ? vdump + 6 * 256 : nullptr))
    pb = tap_ptrs<2>(F, tB, ts[1].g, qb);
    EGO_ROLL_PLANE(2, (void)0, 6, (vdump ? vdump + 12 * 256 : nullptr))
#undef EGO_ROLL_PLANE
    if (mixed) {
#pragma unroll
      for (int r = 0; r < 16; ++r) fe[r] = g ? fe2[r] : fe[r];
    }
    return;
  }
  team_load<0>(F, tA, ts[0].g, qa, raw);
  team_finish<0>(tA, raw, ga);
  __builtin_amdgcn_sched_barrier(0);
  team_load<0>(F, tB, ts[1].g, qb, raw);
  team_finish<0>(tB, raw, gb);
  team_to_halves(ga, gb, v);
  dump24(vdump, v);
  __builtin_amdgcn_sched_barrier(0);
  team_load<1>(F, tA, ts[0].g, qa, raw);
  basis3<0>(BASH, lane, g0, mixed, v, fe, fe2);
  team_finish<1>(tA, raw, ga);
  __builtin_amdgcn_sched_barrier(0);
  team_load<1>(F, tB, ts[1].g, qb, raw);
  team_finish<1>(tB, raw, gb);
  team_to_halves(ga, gb, v);
  dump24(vdump ? vdump + 6 * 256 : nullptr, v);
  __builtin_amdgcn_sched_barrier(0);
  team_load<2>(F, tA, ts[0].g, qa, raw);
  basis3<3>(BASH, lane, g0, mixed, v, fe, fe2);
  team_finish<2>(tA, raw, ga);
  __builtin_amdgcn_sched_barrier(0);
  team_load<2>(F, tB, ts[1].g, qb, raw);
  team_finish<2>(tB, raw, gb);
  team_to_halves(ga, gb, v);
  dump24(vdump ? vdump + 12 * 256 : nullptr, v);
  basis3<6>(BASH, lane, g0, mixed, v, fe, fe2);
  if (mixed) {
#pragma unroll
    for (int r = 0; r < 16; ++r) fe[r] = g ? fe2[r] : fe[r];
  }
}

// ---- half-precision appearance tables (ego_scene.app_f16) --------------------------------------------------------
// Same [H][W][48] channel-last layout with IEEE half elements (96 B per texel).  Lane half h owns the 16-byte chunks
// 2i+h (8 channels each), so one plane is 18 loads per lane and yields 24 products: one plane per pipeline stage, three
// stages per tile instead of six — the gather is bound by bytes in flight per VGPR, and a VGPR now carries two values.
// Interpolation and everything downstream stay fp32.
__host__ __device__ constexpr int app_channel_f16(int kk, int h) {
  return (kk / APP_HALF) * APP_C + ((kk % APP_HALF) / 8) * 16 + 8 * h + (kk % 8);
}

template <int I>
__device__ __forceinline__ void gather_plane_f16(const DevField& F, const VMTaps& t, int g, int h, float v[APP_HALF]) {
#pragma clang fp contract(fast)
  const Lin1 X = t.ax[vm_plane_x(I)], Y = t.ax[vm_plane_y(I)], Ln = t.ax[vm_line_ax(I)];
  const int W = F.res[vm_plane_x(I)];
  // element offsets are in halves; the DevField pointers are nominally float*
  const _Float16* P = (const _Float16*)(g ? F.plane[1][I] : F.plane[0][I]) + 8 * h;
  const _Float16* L = (const _Float16*)(g ? F.line[1][I] : F.line[0][I]) + 8 * h;
  const h8* p00 = (const h8*)(P + (Y.i0 * W + X.i0) * APP_C);
  const h8* p01 = (const h8*)(P + (Y.i0 * W + X.i1) * APP_C);
  const h8* p10 = (const h8*)(P + (Y.i1 * W + X.i0) * APP_C);
  const h8* p11 = (const h8*)(P + (Y.i1 * W + X.i1) * APP_C);
  const h8* l0 = (const h8*)(L + Ln.i0 * APP_C);
  const h8* l1 = (const h8*)(L + Ln.i1 * APP_C);
  const float w00 = __fmul_rn(Y.w0, X.w0), w01 = __fmul_rn(Y.w0, X.w1);
  const float w10 = __fmul_rn(Y.w1, X.w0), w11 = __fmul_rn(Y.w1, X.w1);
#pragma unroll
  for (int c = 0; c < 3; ++c) {  // chunk 2c+h: h8 index 2c
    const h8 a00 = p00[2 * c], a01 = p01[2 * c], a10 = p10[2 * c], a11 = p11[2 * c], b0 = l0[2 * c], b1 = l1[2 * c];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float pv = (float)a00[e] * w00 + (float)a01[e] * w01 + (float)a10[e] * w10 + (float)a11[e] * w11;
      const float lv = (float)b0[e] * Ln.w0 + (float)b1[e] * Ln.w1;
      v[c * 8 + e] = pv * lv;
    }
  }
}

template <bool MASKED>
__device__ __forceinline__ void gather_basis_f16(const DevField& F, const VMTaps& taps, const u32x4* __restrict__ BASH, int lane,
                                                 int g, int h, int gsel, bool keep_in, f32x16& fe) {
  const bool keep = !MASKED || keep_in;
  float v0[APP_HALF], v1[APP_HALF], v2[APP_HALF];
  BasisFrag f0, f1, f2;
  gather_plane_f16<0>(F, taps, g, h, v0);
  f0 = basis_frag<0>(BASH, lane, gsel); f1 = basis_frag<1>(BASH, lane, gsel); f2 = basis_frag<2>(BASH, lane, gsel);
  __builtin_amdgcn_sched_barrier(0);
  gather_plane_f16<1>(F, taps, g, h, v1);
  basis_step(f0, v0, keep, fe); basis_step(f1, v0 + 8, keep, fe); basis_step(f2, v0 + 16, keep, fe);
  f0 = basis_frag<3>(BASH, lane, gsel); f1 = basis_frag<4>(BASH, lane, gsel); f2 = basis_frag<5>(BASH, lane, gsel);
  __builtin_amdgcn_sched_barrier(0);
  gather_plane_f16<2>(F, taps, g, h, v2);
  basis_step(f0, v1, keep, fe); basis_step(f1, v1 + 8, keep, fe); basis_step(f2, v1 + 16, keep, fe);
  f0 = basis_frag<6>(BASH, lane, gsel); f1 = basis_frag<7>(BASH, lane, gsel); f2 = basis_frag<8>(BASH, lane, gsel);
  __builtin_amdgcn_sched_barrier(0);
  basis_step(f0, v2, keep, fe); basis_step(f1, v2 + 8, keep, fe); basis_step(f2, v2 + 16, keep, fe);
}

// PX: arithmetic of the MLP's correction terms: 0 = fp16 (f16x3), 1 = fp8 (f16f8), 2 = fp6 (f16f6)
// FOLD: rows H / J in the epilogue (models/EgoNeRF.py:579-598): a wave owns whole rays (S a multiple of 32, a ray = S / 32 consecutive
// tiles), keeps sum w rgb / sum w (lane half 0) and sum w z (lane half 1) per lane across the ray's tiles and finishes the pixel -
// background, clamp, depth - itself; no per-sample colours are written and k_composite (12 us per 4096 x 512 launch: 25 MB of colours
// written here, 42 MB read there) is not launched.
template <int MODE, bool DUMP = false, bool TAB16 = false, int PX = 0, bool FOLD = false>
__global__ __launch_bounds__(512) void k_shade_h(ShadeArgs A) {
  constexpr bool P8 = PX == 1, P6 = PX == 2;
  static_assert(!FOLD || (MODE == MODE_SHADE && !DUMP), "compositing folds into the fused inference kernel only");
  static_assert(!PX || (!DUMP && MODE != MODE_APP), "the fp8 / fp6 correction arithmetic exists for the inference MLP only");
  __shared__ __attribute__((aligned(16))) float lds[(MODE == MODE_APP ? 0 : LDS_W_FLOATS) + 4];
  const float* blob = A.packed + PACKED_FLOATS;  // the f16x3 half of the packed blob
  if (MODE != MODE_APP) {
    const f32x4* src = (const f32x4*)blob;
    const f32x4* src8 = (const f32x4*)(A.packed + 2 * PACKED_FLOATS + BASIS16_FLOATS_C + (P6 ? F8_FLOATS : 0));  // f16f8 / f16f6 layout of W1 / W2
    // The 150 KB image goes to LDS by the loads themselves (global_load_lds_dwordx4: a wave instruction moves 1 KB, no VGPRs), and
    // nobody waits for it here: the first tile's gather + basis phase reads no LDS, so the workgroup meets at a barrier in front of
    // its first MLP phase instead (`need_sync` below).  A launch has ~32 us of fixed cost at 4096 x 512 (tools/shade_scaling.py), of
    // which the fill - every CU pulling the same 150 KB through its L1 with all waves parked - was the largest piece.
    constexpr int NQ = LDS_W_FLOATS / 4, NCH = NQ / 64;   // float4 quads; whole 1 KB chunks (the image ends 1 quad behind them)
    static_assert(F8_FLOATS % 256 == 0 && NQ - NCH * 64 < 64, "chunking of the LDS image");
    const int wv_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int c = wv_u; c < NCH; c += 8) {
      const f32x4* from = ((PX && c < F8_FLOATS / 256) ? src8 : src) + c * 64 + (threadIdx.x & 63);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)from, (__attribute__((address_space(3))) void*)(lds + c * 256), 16, 0, 0);
    }
    for (int i = NCH * 64 + threadIdx.x; i < NQ; i += 512) ((f32x4*)lds)[i] = src[i];
  }
  bool need_sync = MODE != MODE_APP;   // wave-uniform: the barrier that publishes the LDS image is still ahead
  {
    // MODE.FP16_OVFL = 1: an out-of-range f32 -> fp8 / f16 conversion saturates to the largest finite value instead of producing
    // NaN / inf (v_cvt_pk_fp8_f32 returns NaN above 448, tools/fp8_layout_probe.hip); a saturated correction operand costs accuracy
    // of one low-order term, a NaN would poison the pixel.  Round 5: for every arithmetic, not only the fp8 / fp6 ones - in the
    // three-term fp16 split hi = 65504 and lo = x - hi then carry a value up to twice the half range exactly (inf - inf was NaN), and
    // the training forward's x / h1 / h2 dumps (halves) stay finite.
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  // wave priority follows the phase: high while gathering (latency-bound: get the loads out), low in the MLP phase, whose MFMAs
  // fill the matrix pipe anyway - 0.9 % faster than the static priority of k_shade (no priority / static: measured, removed)
  const int64_t n_tiles = (A.M + 31) >> 5;
  const u32x4* W1 = (const u32x4*)(lds + OFF_W1);
  const u32x4* W2 = (const u32x4*)(lds + OFF_W2);
  const f32x4* B1 = (const f32x4*)(lds + OFF_B1);
  const f32x4* B2 = (const f32x4*)(lds + OFF_B2);
  const f32x4* W3 = (const f32x4*)(lds + OFF_W3);
  // basis fragments: K order of the fp32-table gather, or of the fp16-table gather (third region of the blob)
  const u32x4* BASH = TAB16 ? (const u32x4*)(A.packed + 2 * PACKED_FLOATS) : (const u32x4*)(blob + OFF_BASIS);

  // Tile order: a wave walks a contiguous run of tiles, i.e. along its rays (the angular taps of the next tile are the lines this
  // wave's L1 just served: -1.5..2.3 % kernel time against tiles dealt round-robin over the grid, A/B on one box).  Skipped tiles
  // (tile mask) are the tails of rays and every wave owns whole rays; sharing the ACTIVE tiles evenly instead (prefix sums over the
  // mask in the prologue, or chunks dealt through an atomic counter) was built and measured: DESIGN.md 4.3.
  const int64_t n_wv = (int64_t)gridDim.x * 8;
  const int64_t tpr = FOLD ? (int64_t)(A.S >> 5) : 0;   // tiles per ray
  const int64_t per_wave = FOLD ? ((A.M / A.S + n_wv - 1) / n_wv) * tpr : (n_tiles + n_wv - 1) / n_wv;
  const int64_t tile0 = ((int64_t)blockIdx.x * 8 + wave) * per_wave;
  const int64_t tile1 = tile0 + per_wave < n_tiles ? tile0 + per_wave : n_tiles;
  const int64_t seg_len = FOLD ? tpr : per_wave;   // FOLD: one ray per segment
  for (int64_t seg0 = tile0; seg0 < tile1; seg0 += seg_len) {
  const int64_t seg1 = seg0 + seg_len < tile1 ? seg0 + seg_len : tile1;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;   // FOLD, lane half 0: sum w r, w g, w b, w; half 1: sum w z in c0
  // the ray's d_z for the depth quirk: wave-uniform address, fetched (scalar) at the top of the ray instead of in its epilogue
  const float fold_dz = FOLD ? A.rays[(seg0 / tpr) * 6 + 5] : 0.f;
  // the tile mask is read 64 tiles at a time (one byte per lane + ballot) and walked with find-first-set: a skipped tile costs no
  // memory round trip (a per-tile flag load sat on the critical path of every skipped tile)
  for (int64_t wbase = seg0; wbase < seg1; wbase += 64) {
    unsigned long long wmask;
    {
      const int64_t t = wbase + (int64_t)lane;
      const bool in = t < seg1;
      wmask = (MODE == MODE_SHADE && A.tile_active) ? __ballot(in && A.tile_active[in ? t : seg0] != 0) : __ballot(in);
    }
  while (wmask != 0ull) {
    const int64_t tile = wbase + (int64_t)__builtin_ctzll(wmask);
    wmask &= wmask - 1ull;
    int lw = lane;
    asm volatile("" : "+v"(lw));  // keeps the LDS weight reads inside the loop (see k_shade)
    if (MODE != MODE_MLP) __builtin_amdgcn_s_setprio(2);
    const int hw = lw >> 5;
    const int64_t m_raw = tile * 32 + j;
    const bool valid = m_raw < A.M;
    const int64_t m = valid ? m_raw : A.M - 1;
    // FOLD: the sample's weight and distance travel while the tile is computed (fetched where they are used, their latency is exposed once per tile)
    float fold_w = 0.f, fold_z = 0.f;
    if (FOLD) { fold_w = A.comp_w[m]; fold_z = A.z[m]; }

    f32x16 fe;
    float vd0 = 0.f, vd1 = 0.f, vd2 = 0.f;
    if (MODE == MODE_MLP) {
      const float* fp = A.feat + m * APP_DIM;
#pragma unroll
      for (int r = 0; r < 16; ++r) fe[r] = (r < NSLOT && 2 * r + h < APP_DIM) ? fp[min(2 * r + h, APP_DIM - 1)] : 0.f;
      vd0 = A.dirs[m * 3]; vd1 = A.dirs[m * 3 + 1]; vd2 = A.dirs[m * 3 + 2];
    } else {
      float a_r, a_th, a_ph;
      int g;
      if (MODE == MODE_APP) {
        const float* p = A.c7n + m * 7;
        g = (p[6] == 0.f) ? 0 : 1;
        const int b = g ? 3 : 0;
        a_r = p[b]; a_th = p[b + 1]; a_ph = p[b + 2];
      } else {
        const uint32_t ray = (uint32_t)m / (uint32_t)A.S;
        const float* R = A.rays + (int64_t)ray * 6;
        vd0 = R[3]; vd1 = R[4]; vd2 = R[5];
        // normalised coordinates come from ego_march_density (no acos/atan2/LUT search here)
        const f32x4 cc = ((const f32x4*)A.coords)[m];
        a_r = cc.x; a_th = cc.y; a_ph = cc.z; g = cc.w != 0.f;
      }
      const bool any_yin = __ballot(g == 0) != 0ull, any_yang = __ballot(g != 0) != 0ull;
      const bool mixed = any_yin && any_yang;
      const int gu = any_yin ? 0 : 1;
      const VMTaps taps = vm_setup(a_r, a_th, a_ph, A.F.res);
#pragma unroll
      for (int r = 0; r < 16; ++r) fe[r] = 0.f;
      // per plane: two half-stages of 18 loads (12 products each); 3 basis k-steps of 8 products.  Loads of the
      // next half-stage are issued before the MFMAs of the previous one; sched_barriers bound the hoisting.
      // Waves whose 32 samples all lie in one grid (the common case) take the path without per-value masking.
      // lw / hw (opaque copies of lane / lane half) keep the per-table base pointers from being hoisted out of the tile
      // loop, where 24 64-bit loop invariants would spill.  Waves whose 32 samples all lie in one grid (the common case)
      // take the path without per-value masking; border-straddling waves run both weight sets masked.
      if (TAB16) {
        if (!mixed) {
          gather_basis_f16<false>(A.F, taps, BASH, lw, g, hw, gu, true, fe);
        } else {
          gather_basis_f16<true>(A.F, taps, BASH, lw, g, hw, 0, g == 0, fe);
          gather_basis_f16<true>(A.F, taps, BASH, lw, g, hw, 1, g != 0, fe);
        }
      } else {
        // team gather: in round rd this lane's 4-lane team serves sample 2*((lane >> 1) & 15) + rd of the tile
        TeamSample ts[2];
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          const int64_t mt_raw = EGO_GATHER_TEAMS ? tile * 32 + 2 * ((lw >> 1) & 15) + rd : tile * 32 + (lw & 31);
          const int64_t mt = mt_raw < A.M ? mt_raw : A.M - 1;
          if (MODE == MODE_APP) {
            const float* p = A.c7n + mt * 7;
            ts[rd].g = (p[6] == 0.f) ? 0 : 1;
            const int b = ts[rd].g ? 3 : 0;
            ts[rd].a_r = p[b]; ts[rd].a_th = p[b + 1]; ts[rd].a_ph = p[b + 2];
          } else {
            const f32x4 c4 = ((const f32x4*)A.coords)[mt];
            ts[rd].a_r = c4.x; ts[rd].a_th = c4.y; ts[rd].a_ph = c4.z; ts[rd].g = c4.w != 0.f;
          }
        }
        float* vd = (DUMP && valid && A.dump_v) ? A.dump_v + dump_off(tile, 144, 0, hw, j) : nullptr;   // (v is optional since ABI v15: the sorted scatter can take d(basis) along)
        // one gather per tile; a border-straddling wave runs the basis steps of each plane twice (yin weights with the
        // yang lanes zeroed, then the reverse) inside gather_basis_team
        gather_basis_team<(MODE == MODE_SHADE && !DUMP), (FOLD ? EGO_HOIST_FOLD : EGO_HOIST_PLAIN)>(A.F, ts, BASH, lw, g, mixed, gu, fe, vd);
      }
    }

    if (MODE == MODE_APP) {
      if (valid) {
        float* o = A.out + m * APP_DIM;
#pragma unroll
        for (int r = 0; r < NSLOT; ++r)
          if (2 * r + h < APP_DIM) o[2 * r + h] = fe[r];
      }
      continue;
    }

    __builtin_amdgcn_s_setprio(0);
    if (need_sync) {
      // the LDS image was written by global_load_lds: its completion is counted by vmcnt, which the barrier's own release fence
      // (lgkmcnt) does not wait for.  Here the gather has already drained the counter, so the wait is free.
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0) (gfx9 encoding: vmcnt = bits 3:0 + 15:14, expcnt 7, lgkmcnt 15 = no wait)
      __syncthreads();
      need_sync = false;
    }
    if (DUMP && valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q) __builtin_nontemporal_store(f32x4{fe[4 * q], fe[4 * q + 1], fe[4 * q + 2], fe[4 * q + 3]}, &((f32x4*)A.dump_fe)[(tile * 4 + q) * 64 + lane]);
    }
    float vw[8];
    {
      float sa0, ca0, sb0, cb0, sa1, ca1, sb1, cb1, sa2, ca2, sb2, cb2;
      sincos_x_2x_hw(vd0, sa0, ca0, sb0, cb0);
      sincos_x_2x_hw(vd1, sa1, ca1, sb1, cb1);
      sincos_x_2x_hw(vd2, sa2, ca2, sb2, cb2);
      vw[0] = h ? sb2 : vd0; vw[1] = h ? ca0 : vd1; vw[2] = h ? cb0 : vd2; vw[3] = h ? ca1 : sa0;
      vw[4] = h ? cb1 : sb0; vw[5] = h ? ca2 : sa1; vw[6] = h ? cb2 : sb1; vw[7] = h ? 0.f : sa2;
    }

    // ---- layer 1 (K-outer, 10 steps of 8 values per lane half) ----------------------------------------------
    f32x16 H[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = B1[(mt * 2 + hw) * 4 + q];
        H[mt][q * 4 + 0] = b.x; H[mt][q * 4 + 1] = b.y; H[mt][q * 4 + 2] = b.z; H[mt][q * 4 + 3] = b.w;
      }
    if (P8) {
      float s1 = 0.f, c1 = 0.f, s2 = 0.f, c2 = 0.f;
      float xs[8];
      const u32x4* W1F = W1 + F8_HI1 / 4;   // fp8 fragments [pair][mt][part][lane]
      h8 ah[4], nh[4];
      v8i a8[4], b8;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) nh[mt] = __builtin_bit_cast(h8, W1[mt * 64 + lw]);
#pragma unroll
      for (int kk = 0; kk < KS1; ++kk) {
        float x;
        if (kk < 5 * NSLOT) {
          const int r = kk / 5, kind = kk % 5;
          if (kind == 0) sincos_x_2x_hw(fe[r], s1, c1, s2, c2);
          x = kind == 0 ? fe[r] : (kind == 1 ? s1 : (kind == 2 ? s2 : (kind == 3 ? c1 : c2)));
        } else if (kk < 5 * NSLOT + 8) {
          x = vw[kk - 5 * NSLOT];
        } else {
          x = 0.f;
        }
        xs[kk & 7] = x;
        if ((kk & 7) == 7) {
          const int step = kk >> 3;
          const h8 bh = split8_f8(xs, b8, step & 1);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) ah[mt] = nh[mt];
          if (step + 1 < KH1) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) nh[mt] = __builtin_bit_cast(h8, W1[((step + 1) * 4 + mt) * 64 + lw]);
          }
          if ((step & 1) == 0) {  // the pair's fp8 fragments: needed after the second step's main-term MFMAs
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a8[mt] = load_a8(W1F, step >> 1, mt, lw);
          }
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) H[mt] = MFMAH(ah[mt], bh, H[mt]);
          if (step & 1) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) H[mt] = MFMA8(a8[mt], b8, H[mt]);
          }
        }
      }
    } else if (P6) {
      float s1 = 0.f, c1 = 0.f, s2 = 0.f, c2 = 0.f;
      const F6Bases fb = f6_bases(lw);
      h8 ah[4], nh[4];
      uint32_t hp[16];
      float res[32], xl[16];
      float amax = 1.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) nh[mt] = f6_hi(lds, fb, false, 0, mt);
      // step-outer / value-inner: the flat 80-iteration form of the other arithmetics exceeds the unroller's size limit with the group
      // code in its body, and a rolled loop indexes the register arrays through scratch
#pragma unroll
      for (int step = 0; step < KH1; ++step) {
        float xs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int kk = step * 8 + e;
          float x;
          if (kk < 5 * NSLOT) {
            const int r = kk / 5, kind = kk % 5;
            if (kind == 0) sincos_x_2x_hw(fe[r], s1, c1, s2, c2);
            x = kind == 0 ? fe[r] : (kind == 1 ? s1 : (kind == 2 ? s2 : (kind == 3 ? c1 : c2)));
          } else if (kk < 5 * NSLOT + 8) {
            x = vw[kk - 5 * NSLOT];
          } else {
            x = 0.f;
          }
          xs[e] = x;
          if (kk >= 64) xl[kk - 64] = x;
        }
        const int sg = step & 3;
        // the block maximum only has to look at the unbounded values (features, raw view direction): every group holds cosines of
        // magnitude ~1 and nothing else above 1, so the running maximum starts at 1
        int mmask = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int kk = step * 8 + e;
          if ((kk < 5 * NSLOT && kk % 5 == 0) || (kk >= 5 * NSLOT && kk < 5 * NSLOT + 3)) mmask |= 1 << e;
        }
        const h8 bh = split8_f6(xs, hp + 4 * sg, res + 8 * sg, amax, mmask);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) ah[mt] = nh[mt];
        if (step + 1 < KH1) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) nh[mt] = f6_hi(lds, fb, false, step + 1, mt);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) H[mt] = MFMAH(ah[mt], bh, H[mt]);
        if (sg == 3) {
          const B6 b6 = group6(hp, res, amax);
          group_mfma6(lds, fb, false, step >> 2, b6, H);
          amax = 1.f;
        } else if (step == KH1 - 1) {
          const B6 b6 = group6_half(xl, res, amax);
          group_mfma6(lds, fb, false, G6_1 - 1, b6, H);
        }
      }
    } else
    {
      float s1 = 0.f, c1 = 0.f, s2 = 0.f, c2 = 0.f;
      float xs[8];
      h8 ah[4], al[4], nh[4], nl[4];  // A fragments of the current / next k-step (LDS reads issued one step ahead)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        nh[mt] = __builtin_bit_cast(h8, W1[(mt * 2 + 0) * 64 + lw]);
        nl[mt] = __builtin_bit_cast(h8, W1[(mt * 2 + 1) * 64 + lw]);
      }
#pragma unroll
      for (int kk = 0; kk < KS1; ++kk) {
        float x;
        if (kk < 5 * NSLOT) {
          const int r = kk / 5, kind = kk % 5;
          if (kind == 0) sincos_x_2x_hw(fe[r], s1, c1, s2, c2);
          x = kind == 0 ? fe[r] : (kind == 1 ? s1 : (kind == 2 ? s2 : (kind == 3 ? c1 : c2)));
        } else if (kk < 5 * NSLOT + 8) {
          x = vw[kk - 5 * NSLOT];
        } else {
          x = 0.f;
        }
        xs[kk & 7] = x;
        if ((kk & 7) == 7) {
          const int step = kk >> 3;
          const HL b = split8(xs, true);
          // [tile][k-step][lane][8 halves]; optional (r05): ego_weight_grad_x re-derives x from the feature slots and the view direction
          if (DUMP && valid && A.dump_x) __builtin_nontemporal_store(pack8_rn(xs), &((u32x4*)A.dump_x)[(tile * KH1 + step) * 64 + lane]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) { ah[mt] = nh[mt]; al[mt] = nl[mt]; }
          if (step + 1 < KH1) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
              nh[mt] = __builtin_bit_cast(h8, W1[(((step + 1) * 4 + mt) * 2 + 0) * 64 + lw]);
              nl[mt] = __builtin_bit_cast(h8, W1[(((step + 1) * 4 + mt) * 2 + 1) * 64 + lw]);
            }
          }
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) H[mt] = MFMAH(ah[mt], b.hi, H[mt]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) H[mt] = MFMAH(al[mt], b.hi, H[mt]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) H[mt] = MFMAH(ah[mt], b.lo, H[mt]);
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) H[mt][r] = relu_f(H[mt][r]);
    if (DUMP && valid) {
#pragma unroll
      for (int st = 0; st < KH2; ++st) {   // k-step st of the next layer = registers 8 (st & 1) .. + 7 of m-tile st >> 1
        float v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = H[st >> 1][8 * (st & 1) + e];
        __builtin_nontemporal_store(pack8_rn(v8), &((u32x4*)A.dump_h1)[(tile * KH2 + st) * 64 + lane]);
      }
      // the ReLU masks as bits: what the shade backward needs of h1 / h2 (it would otherwise re-read both dumps, 1 KB per sample)
      __builtin_nontemporal_store(relu_bits(H), &((u32x2*)A.dump_bits)[(tile * 2 + 0) * 64 + lane]);
    }

    // ---- layer 2 (8 steps), layer 3 on the VALU ----------------------------------------------------------------
    f32x16 G[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = B2[(mt * 2 + hw) * 4 + q];
        G[mt][q * 4 + 0] = b.x; G[mt][q * 4 + 1] = b.y; G[mt][q * 4 + 2] = b.z; G[mt][q * 4 + 3] = b.w;
      }
    if (P8) {
      const u32x4* W2F = W2 + F8_HI2 / 4;
      h8 ah[4], nh[4];
      v8i a8[4], b8;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) nh[mt] = __builtin_bit_cast(h8, W2[mt * 64 + lw]);
#pragma unroll
      for (int step = 0; step < KH2; ++step) {
        float xs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xs[e] = H[(step * 8 + e) >> 4][(step * 8 + e) & 15];
        const h8 bh = split8_f8(xs, b8, step & 1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) ah[mt] = nh[mt];
        if (step + 1 < KH2) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) nh[mt] = __builtin_bit_cast(h8, W2[((step + 1) * 4 + mt) * 64 + lw]);
        }
        if ((step & 1) == 0) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) a8[mt] = load_a8(W2F, step >> 1, mt, lw);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) G[mt] = MFMAH(ah[mt], bh, G[mt]);
        if (step & 1) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) G[mt] = MFMA8(a8[mt], b8, G[mt]);
        }
      }
    } else if (P6) {
      const F6Bases fb = f6_bases(lw);
      h8 ah[4], nh[4];
      uint32_t hp[16];
      float res[32];
      float amax = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) nh[mt] = f6_hi(lds, fb, true, 0, mt);
#pragma unroll
      for (int step = 0; step < KH2; ++step) {
        float xs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xs[e] = H[(step * 8 + e) >> 4][(step * 8 + e) & 15];
        const int sg = step & 3;
        const h8 bh = split8_f6(xs, hp + 4 * sg, res + 8 * sg, amax);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) ah[mt] = nh[mt];
        if (step + 1 < KH2) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) nh[mt] = f6_hi(lds, fb, true, step + 1, mt);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) G[mt] = MFMAH(ah[mt], bh, G[mt]);
        if (sg == 3) {
          const B6 b6 = group6(hp, res, amax);
          group_mfma6(lds, fb, true, step >> 2, b6, G);
          amax = 0.f;
        }
      }
    } else
    {
      h8 ah[4], al[4], nh[4], nl[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        nh[mt] = __builtin_bit_cast(h8, W2[(mt * 2 + 0) * 64 + lw]);
        nl[mt] = __builtin_bit_cast(h8, W2[(mt * 2 + 1) * 64 + lw]);
      }
#pragma unroll
      for (int step = 0; step < KH2; ++step) {
        float xs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xs[e] = H[(step * 8 + e) >> 4][(step * 8 + e) & 15];
        const HL b = split8(xs, true);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { ah[mt] = nh[mt]; al[mt] = nl[mt]; }
        if (step + 1 < KH2) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            nh[mt] = __builtin_bit_cast(h8, W2[(((step + 1) * 4 + mt) * 2 + 0) * 64 + lw]);
            nl[mt] = __builtin_bit_cast(h8, W2[(((step + 1) * 4 + mt) * 2 + 1) * 64 + lw]);
          }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) G[mt] = MFMAH(ah[mt], b.hi, G[mt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) G[mt] = MFMAH(al[mt], b.hi, G[mt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) G[mt] = MFMAH(ah[mt], b.lo, G[mt]);
      }
    }
    if (DUMP && valid) {
#pragma unroll
      for (int st = 0; st < KH2; ++st) {
        float v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = relu_f(G[st >> 1][8 * (st & 1) + e]);
        __builtin_nontemporal_store(pack8_rn(v8), &((u32x4*)A.dump_h2)[(tile * KH2 + st) * 64 + lane]);
      }
      __builtin_nontemporal_store(relu_bits(G), &((u32x2*)A.dump_bits)[(tile * 2 + 1) * 64 + lane]);
    }
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      f32x4 w3[16];  // all 16 rows of this tile first: one LDS latency instead of 16 exposed round trips
#pragma unroll
      for (int r = 0; r < 16; ++r) w3[r] = W3[(mt * 2 + hw) * 16 + r];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float hv = relu_f(G[mt][r]);
        o0 = fmaf(hv, w3[r].x, o0); o1 = fmaf(hv, w3[r].y, o1); o2 = fmaf(hv, w3[r].z, o2);
      }
    }
    o0 += __shfl_xor(o0, 32, 64);
    o1 += __shfl_xor(o1, 32, 64);
    o2 += __shfl_xor(o2, 32, 64);
    if (FOLD) {
      // weights are >= 0 and a tile skipped above holds only zeros (the fold is used with shade_above = 0), so nothing is conditional
      const float* b3 = lds + OFF_B3;
      const float wgt = valid ? fold_w : 0.f;
      const float x0 = h ? fold_z : sigmoidf(o0 + b3[0]);
      c0 += wgt * x0;
      c1 += wgt * sigmoidf(o1 + b3[1]);
      c2 += wgt * sigmoidf(o2 + b3[2]);
      c3 += wgt;
    } else if (valid && h == 0) {
      const float* b3 = lds + OFF_B3;
      float* o = A.out + m * 3;
      o[0] = sigmoidf(o0 + b3[0]);
      o[1] = sigmoidf(o1 + b3[1]);
      o[2] = sigmoidf(o2 + b3[2]);
    }
  }
  }
  if (FOLD) {   // the pixel of ray seg0 / tpr (rows H, J: k_composite's arithmetic on the per-lane sums)
    // sums over the 32 lanes of each half on the DPP path (no LDS round trips: the wave is alone with this chain at the end of a ray):
    // row_shr 1, 2, 4, 8 inside the 16-lane rows, then row_bcast:15 into rows 1 / 3: lane 31 holds half 0's sums, lane 63 half 1's
#define EGO_SUM_STEP(v, ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, false))
#define EGO_HALF_SUM(v) EGO_SUM_STEP(v, 0x111, 0xf); EGO_SUM_STEP(v, 0x112, 0xf); EGO_SUM_STEP(v, 0x114, 0xf); EGO_SUM_STEP(v, 0x118, 0xf); EGO_SUM_STEP(v, 0x142, 0xa)
    EGO_HALF_SUM(c0); EGO_HALF_SUM(c1); EGO_HALF_SUM(c2); EGO_HALF_SUM(c3);
#undef EGO_HALF_SUM
#undef EGO_SUM_STEP
    const float dp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c0), 63));
    if (lane == 31) {
      const int64_t ray = seg0 / tpr;
      const float* R = A.rays + ray * 6;
      float cr = c0, cg = c1, cb = c2;
      if (A.envmap) {
        float e[3];
        envmap_lookup(A.envmap, A.envmap_h, R[3], R[4], R[5], e);
        const float b = A.comp_bg[ray];
        const float bx = b * e[0], by = b * e[1], bz = b * e[2];
        cr += bx; cg += by; cb += bz;
        if (A.bg_map) { A.bg_map[ray * 3] = bx; A.bg_map[ray * 3 + 1] = by; A.bg_map[ray * 3 + 2] = bz; }
        if (A.env_map) { A.env_map[ray * 3] = e[0]; A.env_map[ray * 3 + 1] = e[1]; A.env_map[ray * 3 + 2] = e[2]; }
      }
      A.rgb_map[ray * 3] = fminf(fmaxf(cr, 0.f), 1.f);
      A.rgb_map[ray * 3 + 1] = fminf(fmaxf(cg, 0.f), 1.f);
      A.rgb_map[ray * 3 + 2] = fminf(fmaxf(cb, 0.f), 1.f);
      if (A.depth) A.depth[ray] = dp + (1.f - c3) * fold_dz;  // (1-acc) * d_z: reference quirk, EgoNeRF.py:598
    }
  }
  }
  if (need_sync) {   // a wave without a tile still owes the workgroup its arrival - and its DMA chunks, which must have LANDED first
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): global_load_lds completion is tracked by vmcnt, not by the barrier's lgkmcnt fence
    __syncthreads();
  }
}

#include "ego_train.inc"

int check_shade_config(const ego_scene* sc, const char* who, bool need_tables, bool need_mlp) {
  if (!sc) return ego_fail(EGO_E_BADARG, "%s: null scene", who);
  if (sc->app_dim != APP_DIM) return ego_fail(EGO_E_UNSUPPORTED, "%s: app_dim %d (supported: 27)", who, sc->app_dim);
  if (need_tables) {
    if (sc->app.n_comp != APP_C) return ego_fail(EGO_E_UNSUPPORTED, "%s: appearance n_comp %d (supported: 48)", who, sc->app.n_comp);
    for (int g = 0; g < 2; ++g)
      for (int i = 0; i < 3; ++i)
        if (!sc->app.plane[g][i] || !sc->app.line[g][i]) return ego_fail(EGO_E_BADARG, "%s: null appearance table", who);
    if (sc->app.res[0] < 2 || sc->app.res[1] < 2 || sc->app.res[2] < 2) return ego_fail(EGO_E_BADARG, "%s: appearance resolution < 2", who);
    if (!ego_field_is_compact(sc->app, 4))
      return ego_fail(EGO_E_BADARG, "%s: the 12 appearance tables must lie within 4 GB of each other (allocate them from one buffer)", who);
  }
  if (need_mlp && (sc->mlp_in != MLP_IN || sc->mlp_hidden != HID || sc->view_pe != 2 || sc->fea_pe != 2))
    return ego_fail(EGO_E_UNSUPPORTED, "%s: MLP_Fea config in=%d hidden=%d view_pe=%d fea_pe=%d (supported: 150/128/2/2)", who,
                    sc->mlp_in, sc->mlp_hidden, sc->view_pe, sc->fea_pe);
  if (!sc->packed) return ego_fail(EGO_E_BADARG, "%s: scene.packed is null (call ego_pack_mlp first)", who);
  return EGO_OK;
}

int check_app16(const ego_scene* sc, const char* who) {
  if (sc->app16.n_comp != APP_C) return ego_fail(EGO_E_BADARG, "%s: app_f16 is set but app16.n_comp is %d", who, sc->app16.n_comp);
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i)
      if (!sc->app16.plane[g][i] || !sc->app16.line[g][i]) return ego_fail(EGO_E_BADARG, "%s: app_f16 is set but an app16 table is null", who);
  return EGO_OK;
}

unsigned shade_grid(int64_t M) {
  const int64_t tiles = (M + 31) >> 5;
  const int64_t wgs = (tiles + 7) / 8;
  return (unsigned)(wgs < 256 ? wgs : 256);  // persistent: one 8-wave workgroup per CU
}

}  // namespace

// ego_render_forward's question (csrc/ego_render.hip): can shading and compositing run as one launch for this scene and sample count?
bool ego_can_fold_composite(const ego_scene* sc, int32_t S) {
  return sc && ego_shape_is_tuned(sc) && !sc->app_f16 && sc->mlp_precision != EGO_PREC_F32 && sc->weight_thres <= 0.f && S >= 32 && (S & 31) == 0;
}

// ... and does it pay?  The folded kernel deals whole RAYS to its waves (a wave finishes its rays' pixels itself), the two-launch kernel
// deals 32-sample tiles: with few rays, or a ray count just above a multiple of the wave count, some waves get one ray more than the
// others and the launch runs as long as they do (4097 rays x 512 samples: 48 tiles on the critical path against 33).  Folded only when
// its critical path is within 3 % of the tile-granular one.
bool ego_fold_is_balanced(int64_t N, int32_t S) {
  if (N <= 0 || S < 32) return false;
  const int64_t tpr = S >> 5, n_wv = (int64_t)shade_grid(N * (int64_t)S) * 8;
  const int64_t rays_per_wave = (N + n_wv - 1) / n_wv, tiles_per_wave = (N * tpr + n_wv - 1) / n_wv;
  return rays_per_wave * tpr * 100 <= tiles_per_wave * 103;
}

extern "C" {

int32_t ego_render_forward_folds(const ego_scene* sc, int64_t N, int32_t S) {
  const char* f = getenv("EGO_RENDER_FOLD");
  if (f && f[0] == '0') return 0;
  if (!ego_can_fold_composite(sc, S)) return 0;
  return (f && f[0] == '1') || ego_fold_is_balanced(N, S) ? 1 : 0;
}

constexpr int BASIS16_FLOATS = 2 * KHB * 2 * 64 * 4;  // [2 g][9 steps][2 terms][64 lanes][8 halves]

int64_t ego_packed_floats(void) { return 2 * (int64_t)PACKED_FLOATS + BASIS16_FLOATS + F8_FLOATS + F6_FLOATS; }

int64_t ego_packed_floats_scene(const ego_scene* sc) {
  if (!sc) return -1;
  return ego_shape_is_tuned(sc) ? ego_packed_floats() : ego_generic_packed_floats(sc);
}

int ego_pack_mlp(const ego_scene* sc, float* packed_out, void* stream) { return ego_pack_mlp_for(sc, packed_out, 0, stream); }

int64_t ego_packed_floats_compat(const ego_scene* sc) { return sc ? ego_generic_packed_floats(sc) : -1; }

int ego_pack_mlp_compat(const ego_scene* sc, float* packed_out, void* stream) {
  EGO_TRACE("ego_pack_mlp_compat");
  EGO_REQUIRE(sc && packed_out, "pack_mlp_compat: null argument");
  if (sc->head != EGO_HEAD_RGB)
    for (int i = 0; i < 3; ++i) EGO_REQUIRE(sc->mlp_w[i] && sc->mlp_b[i], "pack_mlp_compat: null MLP weight");
  EGO_REQUIRE(sc->basis[0] && sc->basis[1], "pack_mlp_compat: null basis matrix");
  return ego_generic_pack(sc, packed_out, stream);
}

int ego_pack_mlp_for(const ego_scene* sc, float* packed_out, int32_t for_training, void* stream) {
  EGO_TRACE("ego_pack_mlp_for");
  EGO_REQUIRE(sc && packed_out, "pack_mlp: null argument");
  if (sc->head != EGO_HEAD_RGB)   // RGBRender has no MLP (tensorBase.py:37-39): only the basis matrices are packed
    for (int i = 0; i < 3; ++i) EGO_REQUIRE(sc->mlp_w[i] && sc->mlp_b[i], "pack_mlp: null MLP weight");
  EGO_REQUIRE(sc->basis[0] && sc->basis[1], "pack_mlp: null basis matrix");
  if (!ego_shape_is_tuned(sc)) return ego_generic_pack(sc, packed_out, stream);   // any other shape: the fp32 compatibility kernels' layout
  k_pack_mlp<<<(PACKED_FLOATS + 255) / 256, 256, 0, (hipStream_t)stream>>>(sc->mlp_w[0], sc->mlp_b[0], sc->mlp_w[1], sc->mlp_b[1],
                                                                          sc->mlp_w[2], sc->mlp_b[2], sc->basis[0], sc->basis[1],
                                                                          packed_out);
  if (int e = ego_launch_status("k_pack_mlp")) return e;
  k_pack_mlp_h<<<(PACKED_FLOATS + 255) / 256, 256, 0, (hipStream_t)stream>>>(sc->mlp_w[0], sc->mlp_b[0], sc->mlp_w[1], sc->mlp_b[1],
                                                                            sc->mlp_w[2], sc->mlp_b[2], sc->basis[0], sc->basis[1],
                                                                            packed_out, packed_out + PACKED_FLOATS);
  if (int e = ego_launch_status("k_pack_mlp_h")) return e;
  k_pack_basis16<<<(BASIS16_FLOATS + 255) / 256, 256, 0, (hipStream_t)stream>>>(sc->basis[0], sc->basis[1], packed_out + 2 * PACKED_FLOATS);
  if (int e = ego_launch_status("k_pack_basis16")) return e;
  static_assert(BASIS16_FLOATS == BASIS16_FLOATS_C, "blob region sizes");
  if (for_training) return EGO_OK;   // the f16f8 / f16f6 images below are read by the inference arithmetics only
  k_pack_mlp_f8<<<(F8_FLOATS + 255) / 256, 256, 0, (hipStream_t)stream>>>(sc->mlp_w[0], sc->mlp_w[1],
                                                                          packed_out + 2 * PACKED_FLOATS + BASIS16_FLOATS);
  if (int e = ego_launch_status("k_pack_mlp_f8")) return e;
  float* f6 = packed_out + 2 * PACKED_FLOATS + BASIS16_FLOATS + F8_FLOATS;
  if (const hipError_t err = hipMemsetAsync(f6, 0, sizeof(float) * F6_FLOATS, (hipStream_t)stream)) return (int)err;  // the gaps behind the blocks
  k_pack_mlp_f6<<<(F8_HI1 + F8_HI2 + 255) / 256, 256, 0, (hipStream_t)stream>>>(sc->mlp_w[0], sc->mlp_w[1], f6);
  if (int e = ego_launch_status("k_pack_mlp_f6")) return e;
  k_pack_mlp_f6_frag<<<((G6_1 + G6_2) * 4 * 64 * 2 + 7) / 8, 256, 0, (hipStream_t)stream>>>(sc->mlp_w[0], sc->mlp_w[1], f6);
  return ego_launch_status("k_pack_mlp_f6_frag");
}

int ego_shade_kernel_info(int32_t precision, int32_t* out, int32_t n) {
  EGO_REQUIRE(out && n == 8, "shade_kernel_info: out must hold 8 values");
  EGO_REQUIRE(precision == EGO_PREC_F16X3 || precision == EGO_PREC_F32 || precision == EGO_PREC_F16F8 || precision == EGO_PREC_F16F6,
              "shade_kernel_info: unknown precision");
  out[0] = 32;
  if (precision == EGO_PREC_F32) {          // k_shade: one v_mfma_f32_32x32x2_f32 per k (basis) / per k and m-tile (layers 1, 2)
    out[1] = KS_BASIS + 4 * (KS1 + KS2); out[2] = 0; out[3] = 2 * 32 * 32 * 2; out[4] = 0; out[5] = 0;
  } else if (precision == EGO_PREC_F16X3) {  // three v_mfma_f32_32x32x16_f16 per 8-k step (and m-tile)
    out[1] = 3 * (KHB + 4 * (KH1 + KH2)); out[2] = 0; out[3] = 2 * 32 * 32 * 16; out[4] = 0; out[5] = 0;
  } else if (precision == EGO_PREC_F16F6) {  // basis as f16x3; layers 1, 2: one fp16 MFMA per step + two fp6 MFMAs per group of four steps
    out[1] = 3 * KHB + 4 * (KH1 + KH2); out[2] = 4 * 2 * (G6_1 + G6_2); out[3] = 2 * 32 * 32 * 16; out[4] = 2 * 32 * 32 * 64;
    out[5] = 2 * (G6_1 + G6_2);              // 32-value conversions (v_cvt_scalef32_pk32_fp6_f16 / _2xpk16_fp6_f32)
  } else {                                   // basis as f16x3; layers 1, 2: one fp16 MFMA per step + one fp8 MFMA per pair of steps
    out[1] = 3 * KHB + 4 * (KH1 + KH2); out[2] = 4 * (KH1 / 2 + KH2 / 2); out[3] = 2 * 32 * 32 * 16; out[4] = 2 * 32 * 32 * 64;
    out[5] = (KS1 + KS2) / 2 * 2;            // per pair of values: one v_cvt_pk_fp8_f32 (x) and one v_cvt_scalef32_pk_fp8_f32 (residual)
  }
  out[6] = 2 * (3 * APP_C * APP_DIM + MLP_IN * HID + HID * HID + HID * 3);
  out[7] = 3 * (4 + 2) * APP_C * 4;
  return EGO_OK;
}

int ego_app_feature(const ego_scene* sc, const float* c7n, int64_t M, float* out, void* stream) {
  EGO_TRACE("ego_app_feature");
  EGO_REQUIRE(M >= 0 && M < (1ll << 31), "app_feature: M out of range [0, 2^31)");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(c7n && out, "app_feature: null argument");
  if (sc && !ego_shape_is_tuned(sc)) return ego_generic_app_feature(sc, c7n, M, out, stream);
  if (int e = check_shade_config(sc, "app_feature", true, false)) return e;
  ShadeArgs a{};
  a.c = make_coords(*sc); a.F = make_field(sc->app); a.packed = sc->packed; a.c7n = c7n; a.out = out; a.M = M; a.S = 1;
  if (sc->mlp_precision == EGO_PREC_F32) k_shade<MODE_APP><<<shade_grid(M), 512, 0, (hipStream_t)stream>>>(a);
  else if (sc->app_f16) {
    if (int e = check_app16(sc, "app_feature")) return e;
    a.F = make_field(sc->app16);
    k_shade_h<MODE_APP, false, true><<<shade_grid(M), 512, 0, (hipStream_t)stream>>>(a);
  } else k_shade_h<MODE_APP><<<shade_grid(M), 512, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_shade<APP>");
}

int ego_mlp_fea(const ego_scene* sc, const float* viewdirs, const float* feat, int64_t M, float* rgb, void* stream) {
  EGO_TRACE("ego_mlp_fea");
  EGO_REQUIRE(M >= 0 && M < (1ll << 31), "mlp_fea: M out of range [0, 2^31)");
  if (M == 0) return EGO_OK;
  EGO_REQUIRE(viewdirs && feat && rgb, "mlp_fea: null argument");
  if (sc && !ego_shape_is_tuned(sc)) return ego_generic_mlp_fea(sc, viewdirs, feat, M, rgb, stream);
  if (int e = check_shade_config(sc, "mlp_fea", false, true)) return e;
  ShadeArgs a{};
  a.c = make_coords(*sc); a.packed = sc->packed; a.feat = feat; a.dirs = viewdirs; a.out = rgb; a.M = M; a.S = 1;
  if (sc->mlp_precision == EGO_PREC_F32) k_shade<MODE_MLP><<<shade_grid(M), 512, 0, (hipStream_t)stream>>>(a);
  else if (sc->mlp_precision == EGO_PREC_F16F8) k_shade_h<MODE_MLP, false, false, 1><<<shade_grid(M), 512, 0, (hipStream_t)stream>>>(a);
  else if (sc->mlp_precision == EGO_PREC_F16F6) k_shade_h<MODE_MLP, false, false, 2><<<shade_grid(M), 512, 0, (hipStream_t)stream>>>(a);
  else k_shade_h<MODE_MLP><<<shade_grid(M), 512, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_shade<MLP>");
}

int ego_shade(const ego_scene* sc, const float* rays, const float* z, const float* coords, int64_t N, int32_t S, float* rgb,
              const ego_shade_dump* dump, const uint8_t* tile_active, void* stream) {
  EGO_TRACE("ego_shade");
  EGO_REQUIRE(rays && z && rgb && N >= 0 && S >= 1 && N * (int64_t)S < (1ll << 31), "shade: null argument or N*S >= 2^31");
  if (sc && !ego_shape_is_tuned(sc)) {   // any other model shape: fp32 compatibility kernel (inference only)
    if (dump) return ego_fail(EGO_E_UNSUPPORTED, "shade: activation dumps (training) exist for the tuned model shape only (app_dim 27, 48 components, "
                                                  "MLP_Fea 150/128/128 with view_pe = fea_pe = 2)");
    if (N == 0) return EGO_OK;
    return ego_generic_shade(sc, rays, coords, N, S, rgb, tile_active, stream);
  }
  if (int e = check_shade_config(sc, "shade", true, true)) return e;
  EGO_REQUIRE(sc->r_lut && sc->n_r_lut >= 2 && sc->n_r_lut <= LUT_MAX, "shade: r_lut missing or > 1024 entries");
  EGO_REQUIRE(coords || sc->mlp_precision == EGO_PREC_F32, "shade: coords (from ego_march_density) is required unless mlp_precision = EGO_PREC_F32");
  if (N == 0) return EGO_OK;
  ShadeArgs a{};
  a.c = make_coords(*sc); a.F = make_field(sc->app); a.packed = sc->packed; a.rays = rays; a.z = z; a.coords = coords; a.out = rgb; a.tile_active = tile_active;
  a.M = N * (int64_t)S; a.S = S;
  if (dump) {
    EGO_REQUIRE(sc->mlp_precision != EGO_PREC_F32 && dump->h1 && dump->h2 && dump->relu_bits && dump->fe,   // (x and v are optional)
                "shade: activation dumps need the fp16-split arithmetic (not EGO_PREC_F32) and non-null h1 / h2 / v / relu_bits / fe buffers (x is optional)");
    a.dump_x = dump->x; a.dump_h1 = dump->h1; a.dump_h2 = dump->h2; a.dump_v = dump->v; a.dump_bits = dump->relu_bits; a.dump_fe = dump->fe;
    k_shade_h<MODE_SHADE, true><<<shade_grid(a.M), 512, 0, (hipStream_t)stream>>>(a);
  } else if (sc->mlp_precision == EGO_PREC_F32) k_shade<MODE_SHADE><<<shade_grid(a.M), 512, 0, (hipStream_t)stream>>>(a);
  else if (sc->app_f16) {
    if (int e = check_app16(sc, "shade")) return e;
    a.F = make_field(sc->app16);
    if (sc->mlp_precision == EGO_PREC_F16F8) k_shade_h<MODE_SHADE, false, true, 1><<<shade_grid(a.M), 512, 0, (hipStream_t)stream>>>(a);
    else if (sc->mlp_precision == EGO_PREC_F16F6) k_shade_h<MODE_SHADE, false, true, 2><<<shade_grid(a.M), 512, 0, (hipStream_t)stream>>>(a);
    else k_shade_h<MODE_SHADE, false, true><<<shade_grid(a.M), 512, 0, (hipStream_t)stream>>>(a);
  } else if (sc->mlp_precision == EGO_PREC_F16F8) k_shade_h<MODE_SHADE, false, false, 1><<<shade_grid(a.M), 512, 0, (hipStream_t)stream>>>(a);
  else if (sc->mlp_precision == EGO_PREC_F16F6) k_shade_h<MODE_SHADE, false, false, 2><<<shade_grid(a.M), 512, 0, (hipStream_t)stream>>>(a);
  else k_shade_h<MODE_SHADE><<<shade_grid(a.M), 512, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_shade<SHADE>");
}

int ego_shade_composite(const ego_scene* sc, const float* rays, const float* z, const float* coords, const float* weight, const float* bg_weight,
                        int64_t N, int32_t S, const uint8_t* tile_active, float* rgb_map, float* depth, float* bg_map, float* env_map, void* stream) {
  EGO_TRACE("ego_shade_composite");
  EGO_REQUIRE(rays && z && coords && weight && rgb_map && N >= 0 && S >= 32 && (S & 31) == 0 && N * (int64_t)S < (1ll << 31),
              "shade_composite: null argument, S not a multiple of 32, or N*S >= 2^31");
  if (!ego_can_fold_composite(sc, S)) return ego_fail(EGO_E_UNSUPPORTED, "shade_composite: tuned model shape, fp32 tables, a split-precision arithmetic and weight_thres <= 0 only");
  if (int e = check_shade_config(sc, "shade_composite", true, true)) return e;
  EGO_REQUIRE(!sc->envmap || bg_weight, "shade_composite: envmap needs bg_weight");
  if (N == 0) return EGO_OK;
  ShadeArgs a{};
  a.c = make_coords(*sc); a.F = make_field(sc->app); a.packed = sc->packed; a.rays = rays; a.z = z; a.coords = coords; a.tile_active = tile_active;
  a.comp_w = weight; a.comp_bg = bg_weight; a.envmap = sc->envmap; a.envmap_h = sc->envmap_h;
  a.rgb_map = rgb_map; a.depth = depth; a.bg_map = bg_map; a.env_map = env_map;
  a.M = N * (int64_t)S; a.S = S;
  if (sc->mlp_precision == EGO_PREC_F16F6) k_shade_h<MODE_SHADE, false, false, 2, true><<<shade_grid(a.M), 512, 0, (hipStream_t)stream>>>(a);
  else if (sc->mlp_precision == EGO_PREC_F16F8) k_shade_h<MODE_SHADE, false, false, 1, true><<<shade_grid(a.M), 512, 0, (hipStream_t)stream>>>(a);
  else k_shade_h<MODE_SHADE, false, false, 0, true><<<shade_grid(a.M), 512, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_shade<SHADE, composite>");
}

// ---- training step (backward) --------------------------------------------------------------------------------------
int64_t ego_train_packed_floats(void) { return TRAIN_FLOATS; }

int ego_pack_train(const ego_scene* sc, float* out, void* stream) {
  EGO_TRACE("ego_pack_train");
  EGO_REQUIRE(sc && out, "pack_train: null argument");
  if (int e = check_shade_config(sc, "pack_train", true, true)) return e;
  EGO_REQUIRE(sc->mlp_w[0] && sc->mlp_w[1] && sc->mlp_w[2] && sc->basis[0] && sc->basis[1], "pack_train: null weight");
  k_pack_train<<<(TRAIN_FLOATS + 255) / 256, 256, 0, (hipStream_t)stream>>>(sc->mlp_w[0], sc->mlp_w[1], sc->mlp_w[2], sc->basis[0],
                                                                            sc->basis[1], out);
  return ego_launch_status("k_pack_train");
}

int ego_train_layout(int32_t which, int32_t* out, int32_t n) {
  EGO_REQUIRE(out, "train_layout: null output");
  if (which == 0) {  // x dump column -> reference MLP input column (-1: padding)
    EGO_REQUIRE(n == 2 * KS1, "train_layout(0): n must be 160");
    for (int h = 0; h < 2; ++h)
      for (int kk = 0; kk < KS1; ++kk) {
        int ch = -1;
        if (kk < 5 * NSLOT) {
          const int kind = kk % 5, r = kk / 5, f = 2 * r + h;
          if (f < APP_DIM) ch = kind == 0 ? f : (kind == 1 ? 30 + 2 * f : (kind == 2 ? 31 + 2 * f : (kind == 3 ? 84 + 2 * f : 85 + 2 * f)));
        } else if (kk < 5 * NSLOT + 8) {
          const int t = kk - 5 * NSLOT + 8 * h;
          ch = t < 3 ? APP_DIM + t : (t < 15 ? 138 + (t - 3) : -1);
        }
        out[dump_col(kk, h)] = ch;
      }
  } else if (which == 1) {  // h1 / h2 / dh1 / dh2 dump column -> hidden unit
    EGO_REQUIRE(n == HID, "train_layout(1): n must be 128");
    for (int h = 0; h < 2; ++h)
      for (int mt = 0; mt < 4; ++mt)
        for (int r = 0; r < 16; ++r) out[dump_col(mt * 16 + r, h)] = mt * 32 + slot_row(r, h);
  } else if (which == 2) {  // dfe column (within one grid's 32) -> feature (-1: padding)
    EGO_REQUIRE(n == 32, "train_layout(2): n must be 32");
    for (int h = 0; h < 2; ++h)
      for (int r = 0; r < 16; ++r) out[h * 16 + r] = (r < NSLOT && 2 * r + h < APP_DIM) ? 2 * r + h : -1;
  } else if (which == 3) {  // v dump column -> basis input column (0..143)
    EGO_REQUIRE(n == 2 * KS_BASIS, "train_layout(3): n must be 144");
    for (int h = 0; h < 2; ++h)
      for (int kk = 0; kk < KS_BASIS; ++kk) out[dump_col(kk, h)] = app_channel_g(kk, h);
  } else {
    return ego_fail(EGO_E_BADARG, "train_layout: which must be 0..3");
  }
  return EGO_OK;
}

static GradField make_grad(const ego_vm_grad& g) {
  GradField o;
  for (int a = 0; a < 2; ++a)
    for (int i = 0; i < 3; ++i) { o.plane[a][i] = g.plane[a][i]; o.line[a][i] = g.line[a][i]; }
  return o;
}

static int check_grad(const ego_vm_grad* g, const char* who) {
  if (!g) return ego_fail(EGO_E_BADARG, "%s: null gradient tables", who);
  for (int a = 0; a < 2; ++a)
    for (int i = 0; i < 3; ++i)
      if (!g->plane[a][i] || !g->line[a][i]) return ego_fail(EGO_E_BADARG, "%s: null gradient table pointer", who);
  return EGO_OK;
}

int ego_march_backward(const ego_scene* sc, const float* z, const float* alpha, int32_t alpha_stride, const float* weight,
                       const float* sigma, const float* bg_weight, const float* rgb, const float* g_rgb, const float* g_alpha,
                       const float* rgb_raw, const float* env_map, int64_t N, int32_t S, float* dc, float* dfeat, void* stream) {
  EGO_TRACE("ego_march_backward");
  EGO_REQUIRE(N >= 0 && S >= 2 && alpha_stride >= S, "march_backward: bad size");
  if (N == 0) return EGO_OK;
  EGO_REQUIRE(sc && z && alpha && weight && sigma && bg_weight && rgb && g_rgb && rgb_raw && dc && dfeat, "march_backward: null argument");
  if (!sc->act_softplus) return ego_fail(EGO_E_UNSUPPORTED, "march_backward: only the softplus density activation is supported");
  MarchBwdArgs a{};
  a.z = z; a.alpha = alpha; a.g_alpha = g_alpha; a.astride = alpha_stride; a.weight = weight; a.sigma = sigma;
  a.bg = bg_weight; a.rgb = rgb; a.g_rgb = g_rgb;
  a.rgb_raw = rgb_raw; a.env = env_map; a.dc = dc; a.dfeat = dfeat; a.N = N; a.S = S; a.dscale = sc->distance_scale;
  k_march_bwd<<<(unsigned)((N + 3) / 4), 256, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_march_bwd");
}

int ego_shade_backward(const ego_scene* sc, const float* train_packed, const float* coords, float* dc, const float* rgb,
                       const ego_shade_dump* fwd, uint16_t* dh2, uint16_t* dh1, float* dh_scale, float* dfe, float* dv, float* dv_absmax, int64_t N,
                       int32_t S, void* stream) {
  EGO_TRACE("ego_shade_backward");
  EGO_REQUIRE(N >= 0 && S >= 1 && N * (int64_t)S < (1ll << 31), "shade_backward: bad size");
  if (N == 0) {
    if (dv_absmax)
      if (const hipError_t me = hipMemsetAsync(dv_absmax, 0, 4, (hipStream_t)stream)) return ego_fail((int)me, "shade_backward: hipMemsetAsync failed: %s", hipGetErrorString(me));
    return EGO_OK;
  }
  EGO_REQUIRE(sc && train_packed && coords && dc && rgb && fwd && fwd->fe && fwd->relu_bits && dh2 && dh1 && dh_scale && dfe && (dv || dv_absmax),
              "shade_backward: null argument");
  EGO_REQUIRE((((uintptr_t)dh2 | (uintptr_t)dh1 | (uintptr_t)fwd->relu_bits | (uintptr_t)dv) & 15) == 0,
              "shade_backward: dh2 / dh1 / dv / relu_bits must be 16-byte aligned");
  if (int e = check_shade_config(sc, "shade_backward", true, true)) return e;
  ShadeBwdArgs a{};
  a.tpacked = train_packed; a.coords = coords; a.dc = dc; a.rgb = rgb;
  a.fe = fwd->fe; a.bits = fwd->relu_bits; a.dh2 = dh2; a.dh1 = dh1; a.dh_scale = dh_scale; a.dfe = dfe; a.dv = dv; a.M = N * (int64_t)S;
  a.dv_absmax = (uint32_t*)dv_absmax;
  if (dv_absmax)
    if (const hipError_t me = hipMemsetAsync(dv_absmax, 0, 4, (hipStream_t)stream)) return ego_fail((int)me, "shade_backward: hipMemsetAsync failed: %s", hipGetErrorString(me));
  k_shade_bwd<<<shade_grid(a.M), 512, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_shade_bwd");
}

static int scatter_common(const ego_vm_field& f, const ego_vm_grad* g, const float* coords, const float* d, int64_t N, int32_t S,
                          ScatterArgs* a, const char* who) {
  if (!(N >= 0 && S >= 1)) return ego_fail(EGO_E_BADARG, "%s: bad size", who);
  if (N == 0) return EGO_OK;
  if (!coords || !d) return ego_fail(EGO_E_BADARG, "%s: null argument", who);
  if (int e = check_grad(g, who)) return e;
  a->F = make_field(f); a->G = make_grad(*g); a->coords = coords; a->d = d; a->N = N; a->S = S;
  a->seg = 64;
  a->gpr = (S + a->seg - 1) / a->seg;
  return EGO_OK;
}

static dim3 scatter_blocks(const ScatterArgs& a) {
  const int64_t groups = a.N * a.gpr;
  return dim3((unsigned)((groups + 15) / 16), 3);  // 4 groups per wave, 4 waves per workgroup; y = plane
}

int ego_scatter_density(const ego_scene* sc, const ego_vm_grad* gdensity, const float* coords, const float* dfeat, int64_t N, int32_t S,
                        void* stream) {
  EGO_TRACE("ego_scatter_density");
  EGO_REQUIRE(sc, "scatter_density: null scene");
  if (sc->density.n_comp != 16) return ego_fail(EGO_E_UNSUPPORTED, "scatter_density: n_comp %d (supported: 16)", sc->density.n_comp);
  ScatterArgs a{};
  if (int e = scatter_common(sc->density, gdensity, coords, dfeat, N, S, &a, "scatter_density")) return e;
  if (N == 0) return EGO_OK;
  k_vm_scatter<16, true><<<scatter_blocks(a), 256, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_vm_scatter<16>");
}

int ego_scatter_app(const ego_scene* sc, const ego_vm_grad* gapp, const float* coords, const float* dv, int64_t N, int32_t S,
                    void* stream) {
  EGO_TRACE("ego_scatter_app");
  EGO_REQUIRE(sc, "scatter_app: null scene");
  if (sc->app.n_comp != APP_C) return ego_fail(EGO_E_UNSUPPORTED, "scatter_app: n_comp %d (supported: 48)", sc->app.n_comp);
  ScatterArgs a{};
  if (int e = scatter_common(sc->app, gapp, coords, dv, N, S, &a, "scatter_app")) return e;
  if (N == 0) return EGO_OK;
  k_vm_scatter<APP_C, false><<<scatter_blocks(a), 256, 0, (hipStream_t)stream>>>(a);
  return ego_launch_status("k_vm_scatter<48>");
}

}  // extern "C"
