// gq_conv_f32.h -- 3x3 convolution (stride 1, zero padding 1, NHWC) on the fp32 matrix cores, for the narrow ends of the
// conv stack: the encoder's conv_out (512 -> 2 z_channels, pit/modules/unet.py:425-436 -- the layer that PRODUCES z, so
// its rounding decides tokens) and the decoder's conv_in (z_channels -> 512, unet.py:487-489).
//
// Why not the library: MIOpen's implicit-GEMM pick for the 512 -> 32 convolution splits K over workgroups and combines
// with floating-point atomics -- measured on MI355X (tools/determinism_trace.py): z differed by 2-4e-7 between two runs of
// the same input at every batch size, the only non-reproducible call of the encoder.  Here every output element is a
// fixed-order sum: v_mfma_f32_32x32x2_f32 accumulates K in program order (an exact fp32 FMA chain), and where K is split
// over the four waves of a block the four partial tiles are added in wave order through LDS.  Same bits on every run.
//
// Implicit GEMM, D[pixel][cout] += A[pixel][k] B[k][cout], k = (tap, cin).  A wave owns 32 pixels of one image row x 32
// output channels (one 32 x 32 accumulator tile).  Per tap and group of 8 input channels ("step") a lane reads ONE 16-byte
// vector of the activation patch from LDS (lane half h: channels 4h .. 4h + 3 of the group at pixel lane % 32, shifted by
// the tap) and ONE 16-byte vector of weights from L2 (host-side operand order: [cout tile][tap][group][lane][4]), and issues
// four MFMAs -- the K index of MFMA m is channel 4h + m, identically for both operands.  The patch (3 rows x 34 pixels x CK
// channels) is staged per channel chunk with a pixel stride of CK + 4 floats, so that the 16 lanes of a ds_read_b128
// phase cover all 64 banks; with FUSED GroupNorm + SiLU (conv_out: norm_out + swish, unet.py:432-435) the patch is
// normalised and activated on its way into LDS and the normalised tensor is never written.
#pragma once
#include "gq_common.h"
#include "gq_stats.h"
#include "gq_unet_aux.h"

namespace gqhip {

struct ConvF32Params {
  const float *x;         // [B][H][W][Cin] fp32 NHWC
  const float *gamma;     // GN: [Cin] (else null)
  const float *beta;
  const float *pre_bias;  // GN: per-channel bias still pending on x, or null
  const int64_t *stats;   // GN: [B][groups] statistics records of x (+ pre_bias) (gq_stats.h)
  const float *wk;        // weights in operand order [ceil(Cout / 32)][9][Cin / 8][64][4] (zero rows beyond Cout)
  const float *bias;      // [Cout] or null
  float *y;               // [B][H][W][Cout] fp32 NHWC
  int H, W, Cin, Cout, cpg;
  double eps;
};

constexpr int kConvF32PW = 34;   // patch width: 32 pixels + the two halo columns

// One step: 4 MFMAs on the four channels a lane half holds.
__device__ __forceinline__ void conv_f32_step(const f32x4 a, const f32x4 b, f32x16 &acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}

// ---- K split over the four waves of a block: few output channels, many input channels (conv_out) -------------------------
// grid = (B * H * W / 32, Cout / 32); chunk = 64 input channels, wave w multiplies channels 16 w .. 16 w + 15 of each chunk.
template <bool GN, int SILU>
__global__ __launch_bounds__(256, 2) void conv3x3_f32_ksplit_kernel(const ConvF32Params p) {
  constexpr int CK = 64, PS = CK + 4, PW = kConvF32PW;
  constexpr int ITEMS = 3 * PW * (CK / 4);                 // (pixel, channel quad) items of one chunk: 1632
  constexpr int ROUNDS = (ITEMS + 255) / 256;              // 7
  __shared__ __attribute__((aligned(16))) float sX[2][3 * PW * PS];
  __shared__ __attribute__((aligned(16))) float sAff[GN ? 2 : 1][GN ? 1024 : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int segs = (p.W + 31) / 32;               // the last segment of a row may be partly outside the image
  const int seg = blockIdx.x % segs;
  const int yrow = (blockIdx.x / segs) % p.H;
  const long b = blockIdx.x / ((long)segs * p.H);
  const int nt = blockIdx.y;
  const int x0 = seg * 32;
  const int C = p.Cin, ngrp = C / 8;
  const float *xb = p.x + b * (long)p.H * p.W * C;

  if constexpr (GN) {
    const int groups = C / p.cpg;
    const double n = (double)p.cpg * (double)p.H * (double)p.W;
    for (int ch = tid; ch < C; ch += 256) {
      const int g = ch / p.cpg;
      double st_s, st_ss;
      stat_load(p.stats + kStatWords * (b * groups + g), st_s, st_ss);
      const double mean = st_s / n;
      double var = st_ss / n - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double rstd = 1.0 / sqrt(var + p.eps);
      const double pbk = p.pre_bias ? (double)p.pre_bias[ch] : 0.0;
      sAff[0][ch] = (float)(rstd * (double)p.gamma[ch]);
      sAff[1][ch] = (float)((double)p.beta[ch] + (pbk - mean) * rstd * (double)p.gamma[ch]);
    }
    __syncthreads();
  }

  // this thread's items of a chunk: item = tid + 256 r -> pixel item / 16, channel quad item % 16 (the same quad every round)
  const int q = tid & 15;
  f32x4 stage[ROUNDS];
  auto stage_load = [&](int c0) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int item = tid + 256 * r;
      const int px = item >> 4;
      const int R = px / PW, X = px - R * PW;
      const int gy = yrow - 1 + R, gx = x0 - 1 + X;
      const int cy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy), cx = gx < 0 ? 0 : (gx >= p.W ? p.W - 1 : gx);
      if (item < ITEMS) stage[r] = *reinterpret_cast<const f32x4 *>(xb + ((long)cy * p.W + cx) * C + c0 + 4 * q);
    }
  };
  auto stage_store = [&](int c0, float *dst) {
    f32x4 a4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (GN) {
      a4 = *reinterpret_cast<const f32x4 *>(&sAff[0][c0 + 4 * q]);
      sh4 = *reinterpret_cast<const f32x4 *>(&sAff[GN ? 1 : 0][c0 + 4 * q]);
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int item = tid + 256 * r;
      const int px = item >> 4;
      const int R = px / PW, X = px - R * PW;
      const int gy = yrow - 1 + R, gx = x0 - 1 + X;
      const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      f32x4 v = stage[r];
      if constexpr (GN) v = gn_act<SILU>(v, a4, sh4);
      v = v * (in ? 1.f : 0.f);          // the zero padding is of the ACTIVATED tensor (unet.py:436)
      if (item < ITEMS) *reinterpret_cast<f32x4 *>(dst + px * PS + 4 * q) = v;
    }
  };

  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int li = lane & 31, lh = lane >> 5;
  const f32x4 *wv = reinterpret_cast<const f32x4 *>(p.wk) + (long)nt * 9 * ngrp * 64 + lane;

  stage_load(0);
  stage_store(0, sX[0]);
  __syncthreads();
  const int nchunks = C / CK;
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) stage_load((c + 1) * CK);
    // the 18 weight vectors of this wave's 16 channels of the chunk (all in flight before the first MFMA)
    f32x4 bw[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int g = 0; g < 2; ++g) bw[tap][g] = wv[((long)tap * ngrp + c * 8 + 2 * wave + g) * 64];
    const float *sx = sX[c & 1];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float *s = sx + ((tap / 3) * PW + li + tap % 3) * PS + 16 * wave + 4 * lh;
#pragma unroll
      for (int g = 0; g < 2; ++g) conv_f32_step(*reinterpret_cast<const f32x4 *>(s + 8 * g), bw[tap][g], acc);
    }
    if (c + 1 < nchunks) stage_store((c + 1) * CK, sX[(c + 1) & 1]);
    __syncthreads();
  }

  // the four partial tiles, added in wave order
  float *red = sX[0];                                   // 4 x 32 x 32 floats = 16 KiB (every wave is past its last read)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[r];
  __syncthreads();
  {
    const int i = tid >> 3, q4 = tid & 7;
    const f32x4 r0 = *reinterpret_cast<const f32x4 *>(red + (0 * 32 + i) * 32 + 4 * q4);
    const f32x4 r1 = *reinterpret_cast<const f32x4 *>(red + (1 * 32 + i) * 32 + 4 * q4);
    const f32x4 r2 = *reinterpret_cast<const f32x4 *>(red + (2 * 32 + i) * 32 + 4 * q4);
    const f32x4 r3 = *reinterpret_cast<const f32x4 *>(red + (3 * 32 + i) * 32 + 4 * q4);
    f32x4 v = ((r0 + r1) + r2) + r3;
    if (nt * 32 + 4 * q4 < p.Cout && x0 + i < p.W) {    // Cout % 4 == 0; the last tile may be partly padding
      if (p.bias) v = v + *reinterpret_cast<const f32x4 *>(p.bias + nt * 32 + 4 * q4);
      *reinterpret_cast<f32x4 *>(p.y + ((b * p.H + yrow) * (long)p.W + x0 + i) * p.Cout + nt * 32 + 4 * q4) = v;
    }
  }
}

// ---- output channels split over the waves: few input channels, many output channels (decoder conv_in) -------------------
// grid = B * H * W / 32; the block stages the 3 x 34 x Cin patch once (Cin <= 64), wave w computes the cout tiles
// w, w + 4, w + 8, ... -- each a complete K loop, so nothing is combined across waves.
template <int CIN>
__global__ __launch_bounds__(256, 2) void conv3x3_f32_nsplit_kernel(const ConvF32Params p) {
  constexpr int PS = CIN + 4, PW = kConvF32PW, NG = CIN / 8;
  constexpr int ITEMS = 3 * PW * (CIN / 4);
  __shared__ __attribute__((aligned(16))) float sX[3 * PW * PS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int segs = (p.W + 31) / 32;               // the last segment of a row may be partly outside the image
  const int seg = blockIdx.x % segs;
  const int yrow = (blockIdx.x / segs) % p.H;
  const long b = blockIdx.x / ((long)segs * p.H);
  const int x0 = seg * 32;
  const float *xb = p.x + b * (long)p.H * p.W * CIN;
  for (int item = tid; item < ITEMS; item += 256) {
    const int px = item / (CIN / 4), q = item % (CIN / 4);
    const int R = px / PW, X = px - R * PW;
    const int gy = yrow - 1 + R, gx = x0 - 1 + X;
    const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const int cy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy), cx = gx < 0 ? 0 : (gx >= p.W ? p.W - 1 : gx);
    const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + ((long)cy * p.W + cx) * CIN + 4 * q);
    *reinterpret_cast<f32x4 *>(sX + px * PS + 4 * q) = v * (in ? 1.f : 0.f);
  }
  __syncthreads();
  const int li = lane & 31, lh = lane >> 5;
  const int ntiles = (p.Cout + 31) / 32;
  for (int nt = wave; nt < ntiles; nt += 4) {
    const f32x4 *wv = reinterpret_cast<const f32x4 *>(p.wk) + (long)nt * 9 * NG * 64 + lane;
    f32x4 bw[9][NG];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int g = 0; g < NG; ++g) bw[tap][g] = wv[((long)tap * NG + g) * 64];
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float *s = sX + ((tap / 3) * PW + li + tap % 3) * PS + 4 * lh;
#pragma unroll
      for (int g = 0; g < NG; ++g) conv_f32_step(*reinterpret_cast<const f32x4 *>(s + 8 * g), bw[tap][g], acc);
    }
    // register r of lane (h, j): pixel (r & 3) + 8 (r >> 2) + 4 h, channel nt * 32 + j: a store is two 128-byte runs
    if (nt * 32 + li < p.Cout) {
      const float pb = p.bias ? p.bias[nt * 32 + li] : 0.f;
      float *yo = p.y + ((b * p.H + yrow) * (long)p.W + x0) * p.Cout + nt * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (x0 + i < p.W) yo[(long)i * p.Cout] = acc[r] + pb;
      }
    }
  }
}


// ---- conv3x3_cin_small_f32 (round 4): the encoder's conv_in, 3 -> 128 channels (pit/modules/unet.py:411-413) ----------------
// A 3x3 / stride 1 / pad 1 convolution of a channels_last image with CIN <= 4 input channels into 128 output channels: 9 CIN fp32
// FMAs per output in a FIXED order (tap-major, then input channel), + bias, + the statistics of the result for the GroupNorm
// that follows (the first ResnetBlock's norm1).  HBM-bound on its 512-byte-per-pixel output.  Why it exists: it was the last
// convolution of the bench shapes on MIOpen, whose immediate mode runs the first eight calls of a process on a 4 ms naive kernel
// (profiles/r03, r04: naive_conv_ab_nonpacked_fwd_nhwc_float_double_float x 8 = 32 ms per bench run, three of them inside a
// default run's timed region) before its CK solver (0.24 ms) takes over; here: one kernel, ~0.1 ms, no library, no statistics pass.
// Block = 8 x 32 output pixels x 128 channels; lane & 31 = a group of 4 output channels (= one GroupNorm(32) group, its 4 x 9 CIN
// weights in registers for the whole block), lane >> 5 and the wave pick a PAIR of horizontally adjacent pixels per pass
// (16 passes): the pair shares 6 of its 12 patch columns' LDS reads, and a wave's stores are 2 KiB contiguous.
struct ConvInParams {
  const float *x;        // [B, H, W, CIN]
  const float *wk;       // [9 CIN, 128]: wk[(tap * CIN + ci) * 128 + co] = weight[co][ci][tap / 3][tap % 3]
  const float *bias;     // [128] or NULL
  float *y;              // [B, H, W, 128]
  int64_t *stats;        // [B, 32, kStatWords] zeroed, or NULL
  int H, W;
};
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_cin_small_kernel(const ConvInParams p) {
  constexpr int TH = 8, TW = 32, COUT = 128, K = 9 * CIN;
  __shared__ float patch[TH + 2][(TW + 2) * CIN];
  __shared__ int64_t s_stat[32][kStatWords];
  const int tid = threadIdx.x;
  const int tiles_w = p.W / TW, tiles_h = p.H / TH;
  const int tx = blockIdx.x % tiles_w, ty = (blockIdx.x / tiles_w) % tiles_h, b = blockIdx.x / (tiles_w * tiles_h);
  const int y0 = ty * TH, x0 = tx * TW;
  for (int i = tid; i < (TH + 2) * (TW + 2) * CIN; i += 256) {
    const int r = i / ((TW + 2) * CIN), c = i % ((TW + 2) * CIN);
    const int yy = y0 + r - 1, xx = x0 + c / CIN - 1;
    float v = 0.0f;
    if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) v = p.x[(((long)b * p.H + yy) * p.W + xx) * CIN + c % CIN];
    patch[r][c] = v;
  }
  for (int i = tid; i < 32 * kStatWords; i += 256) (&s_stat[0][0])[i] = 0;
  const int g = tid & 31, pr = tid >> 5;            // channel group, pixel-pair slot (0 .. 7)
  float w[K][4];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(p.wk + k * COUT + 4 * g);
    w[k][0] = v.x; w[k][1] = v.y; w[k][2] = v.z; w[k][3] = v.w;
  }
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias4 = *reinterpret_cast<const f32x4 *>(p.bias + 4 * g);
  __syncthreads();
  float s = 0.0f, ss = 0.0f;
  for (int pass = 0; pass < (TH * TW) / 16; ++pass) {
    const int q = pass * 8 + pr;                    // pixel pair: pixels 2 q, 2 q + 1 of the tile (row-major)
    const int py = (2 * q) / TW, px = (2 * q) % TW;
    float xv[3][4 * CIN];                           // 3 rows x 4 columns x CIN of the patch
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int c = 0; c < 4 * CIN; ++c) xv[dy][c] = patch[py + dy][px * CIN + c];
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const int k = (dy * 3 + dx) * CIN + ci;
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            acc[0][o] = __builtin_fmaf(xv[dy][dx * CIN + ci], w[k][o], acc[0][o]);
            acc[1][o] = __builtin_fmaf(xv[dy][(dx + 1) * CIN + ci], w[k][o], acc[1][o]);
          }
        }
    float *dst = p.y + (((long)b * p.H + y0 + py) * p.W + x0 + px) * COUT + 4 * g;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const f32x4 o = {acc[e][0] + bias4.x, acc[e][1] + bias4.y, acc[e][2] + bias4.z, acc[e][3] + bias4.w};
      *reinterpret_cast<f32x4 *>(dst + e * COUT) = o;
      s += (o.x + o.y) + (o.z + o.w);
      ss += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
    }
  }
  if (p.stats) {
    stat_add_f32(&s_stat[g][0], s, ss);
    __syncthreads();
    const int64_t v = (&s_stat[0][0])[tid];         // 32 groups x 8 words = 256 words: one per thread
    stat_flush_word(p.stats + (long)b * 32 * kStatWords + tid, v);
  }
}

}  // namespace gqhip
