// gq_unet_aux.h -- the HBM-bound kernels of the conv stack around the matrix-core kernels: fused GroupNorm (+ SiLU),
// residual adds that leave the next GroupNorm's statistics behind, the Winograd F(2x2,3x3) / F(4x4,3x3) data transforms,
// the sub-pixel upsample glue and the operand splits of the attention GEMMs (pit/modules/unet.py:49-206).
#pragma once
#include "gq_common.h"
#include "gq_stats.h"

namespace gqhip {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// x * sigmoid(x) = x / (1 + e^-x), the quotient by one Newton step on v_rcp_f32 (q = x r; q += (x - d q) r: within
// half an ulp of the IEEE quotient up to a few units of 2^-24 of an ulp, at a third of the IEEE divide's instructions).
// EVERY SiLU of libgqhip goes through this function, so the GroupNorm applied as its own pass and the one fused into
// the Winograd input transforms produce identical bits.  d > 1e37 (x < -85.2; d = inf below -88.7): 1 / d is subnormal,
// the result (|.| < 1e-35) is returned as x * 0 = -0 -- the divide gives -0 for d = inf too -- and never a NaN.
__device__ __forceinline__ float silu_f32(float x) {
  const float d = 1.0f + __expf(-x);
  const float r = __builtin_amdgcn_rcpf(d);
  const float q = x * r;
  const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q, x), r, q);
  return d > 1e37f ? q : q1;
}

// ---- NHWC (channels_last) variants: x[b][hw][c], the layout MIOpen's fp32 igemm kernels want ----
// A block owns a slab of pixels of one image and ALL channels: thread -> channel quad q = tid % (C/4)
// (4 consecutive channels of ONE group since cpg % 4 == 0), pixel lane = tid / (C/4).
__global__ __launch_bounds__(256) void gn_stats_nhwc_kernel(const float *__restrict__ x,
                                                            const float *__restrict__ pre_bias,
                                                            int64_t *__restrict__ stats, int C, long HW, int cpg,
                                                            int slabs) {
  __shared__ int64_t red[kStatWords * 64];   // per-group statistics records (gq_stats.h), groups <= 64
  const int groups = C / cpg, quads = C / 4, lanes = 256 / quads;
  const long b = blockIdx.x / slabs;
  const int slab = blockIdx.x % slabs;
  const long per = (HW + slabs - 1) / slabs;
  const long lo = slab * per, hi = lo + per < HW ? lo + per : HW;
  const int q = threadIdx.x % quads, pl = threadIdx.x / quads;
  for (int w = threadIdx.x; w < kStatWords * groups; w += 256) red[w] = 0;
  __syncthreads();
  f32x4 pb = {0.f, 0.f, 0.f, 0.f};
  if (pre_bias) pb = *reinterpret_cast<const f32x4 *>(pre_bias + 4 * q);
  const float *base = x + (b * HW) * C + 4 * q;
  float s = 0.f, ss = 0.f;
  for (long p = lo + pl; p < hi; p += lanes) {
    f32x4 v = *reinterpret_cast<const f32x4 *>(base + p * C) + pb;
    s += (v.x + v.y) + (v.z + v.w);
    ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  const int g = (4 * q) / cpg;
  stat_add_f32(red + kStatWords * g, s, ss);
  __syncthreads();
  for (int w = threadIdx.x; w < kStatWords * groups; w += 256) stat_flush_word(stats + kStatWords * (b * groups) + w, red[w]);
}

template <int SILU>
__global__ __launch_bounds__(256) void gn_apply_nhwc_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float *__restrict__ y,
                                                            const int64_t *__restrict__ stats,
                                                            const float *__restrict__ pre_bias, int C, long HW,
                                                            int cpg, double eps, int slabs) {
  const int groups = C / cpg, quads = C / 4, lanes = 256 / quads;
  const long b = blockIdx.x / slabs;
  const int slab = blockIdx.x % slabs;
  const long per = (HW + slabs - 1) / slabs;
  const long lo = slab * per, hi = lo + per < HW ? lo + per : HW;
  const int q = threadIdx.x % quads, pl = threadIdx.x / quads;
  const int g = (4 * q) / cpg;
  const double n = (double)cpg * (double)HW;
  double st_s, st_ss;
  stat_load(stats + kStatWords * (b * groups + g), st_s, st_ss);
  const double mean = st_s / n;
  double var = st_ss / n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const double rstd = 1.0 / sqrt(var + eps);
  f32x4 a, sh;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * q + k;
    const double pbk = pre_bias ? (double)pre_bias[c] : 0.0;
    a[k] = (float)(rstd * (double)gamma[c]);
    sh[k] = (float)((double)beta[c] + (pbk - mean) * rstd * (double)gamma[c]);
  }
  const float *xi = x + (b * HW) * C + 4 * q;
  float *yo = y + (b * HW) * C + 4 * q;
  for (long p = lo + pl; p < hi; p += lanes) {
    f32x4 v = *reinterpret_cast<const f32x4 *>(xi + p * C) * a + sh;
    if (SILU) {
      v.x = silu_f32(v.x);
      v.y = silu_f32(v.y);
      v.z = silu_f32(v.z);
      v.w = silu_f32(v.w);
    }
    *reinterpret_cast<f32x4 *>(yo + p * C) = v;
  }
}

__global__ __launch_bounds__(256) void add_bias_nhwc_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                            const float *__restrict__ bias, float *__restrict__ y,
                                                            int C, long total4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4 *>(a)[i] + reinterpret_cast<const f32x4 *>(b)[i];
    if (bias) v = v + *reinterpret_cast<const f32x4 *>(bias + (int)((i * 4) % C));
    reinterpret_cast<f32x4 *>(y)[i] = v;
  }
}

// Residual add that also leaves the GroupNorm statistics of its OUTPUT behind (the next op of the UNet is a
// GroupNorm over exactly this tensor: unet.py:160 -> :140): same block/thread mapping as gn_stats_nhwc_kernel, so
// the consumer's statistics pass (one full read of the tensor) disappears.
__global__ __launch_bounds__(256) void add_bias_stats_nhwc_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                                  const float *__restrict__ bias, float *__restrict__ y,
                                                                  int64_t *__restrict__ stats, int C, long HW, int cpg,
                                                                  int slabs) {
  __shared__ int64_t red[kStatWords * 64];   // per-group statistics records (gq_stats.h), groups <= 64
  const int groups = C / cpg, quads = C / 4, lanes = 256 / quads;
  const long bi = blockIdx.x / slabs;
  const int slab = blockIdx.x % slabs;
  const long per = (HW + slabs - 1) / slabs;
  const long lo = slab * per, hi = lo + per < HW ? lo + per : HW;
  const int q = threadIdx.x % quads, pl = threadIdx.x / quads;
  for (int w = threadIdx.x; w < kStatWords * groups; w += 256) red[w] = 0;
  __syncthreads();
  f32x4 pb = {0.f, 0.f, 0.f, 0.f};
  if (bias) pb = *reinterpret_cast<const f32x4 *>(bias + 4 * q);
  const long off = (bi * HW) * C + 4 * q;
  float s = 0.f, ss = 0.f;
  for (long p = lo + pl; p < hi; p += lanes) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(a + off + p * C) + *reinterpret_cast<const f32x4 *>(b + off + p * C) + pb;
    *reinterpret_cast<f32x4 *>(y + off + p * C) = v;
    s += (v.x + v.y) + (v.z + v.w);
    ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  const int g = (4 * q) / cpg;
  stat_add_f32(red + kStatWords * g, s, ss);
  __syncthreads();
  for (int w = threadIdx.x; w < kStatWords * groups; w += 256) stat_flush_word(stats + kStatWords * (bi * groups) + w, red[w]);
}

// Nearest-neighbour x2 upsample, NHWC (unet.py:69-73 `interpolate(scale_factor=2, mode="nearest")`):
// one thread per (input pixel, channel quad); the 16-byte value is written to the 2x2 output pixels.
__global__ __launch_bounds__(256) void upsample2x_nhwc_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                              int H, int W, int C4, long total) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int q = (int)(t % C4);
    long p = t / C4;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const long b = p / H;
    const f32x4 v = reinterpret_cast<const f32x4 *>(x)[t];
    f32x4 *o = reinterpret_cast<f32x4 *>(y) + ((b * 2 * H + 2 * h) * (2L * W) + 2 * w) * C4 + q;
    o[0] = v;
    o[C4] = v;
    o[2L * W * C4] = v;
    o[2L * W * C4 + C4] = v;
  }
}

// ---- Winograd F(2x2, 3x3), NHWC, stride 1, padding 1 (used for the decoder's wide 3x3 convolutions) -------------
// Y = A^T [ (G g G^T) . (B^T d B) ] A per 4x4 input tile d (stride 2) and 2x2 output tile: 16 multiplies per 4 outputs
// instead of 36.  The element-wise products over the channels are 16 independent GEMMs [tiles, Cin] x [Cin, Cout]
// (torch.bmm -> hipBLASLt); these two kernels are the data transforms around them.
//   V[k][tile][c] = (B^T d B)[k]     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
//   y[2x2]        = A^T m A          A^T = [1 1 1 0; 0 1 -1 -1]
// One thread per (tile, channel quad); tiles = B * (H/2) * (W/2), H and W even.
// Where a transformed value goes.  F16X3 = false: V [k][tile][C] fp32.  F16X3 = true: the operand of ONE fp16 GEMM with
// fp32 accumulation whose K axis carries the three split products (v = h + l, two-term fp16 split of v * scale):
// V3 [k][tile][3C] fp16 = [ h | h | l ], to be multiplied by U3 [k][3C][Cout] = [ U_h ; U_l ; U_h ] (unet._wino_weights_f16):
// V3 U3 = h U_h + h U_l + l U_h, error ~3 * 2^-22 per product -- the level of hipBLASLt's own fp32 (split-bf16) GEMM, at
// 2-2.5x its speed (tools/bmm_bf16x3.py).  `scale` is a power of two chosen by the caller so that |v * scale| < 65504.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// F16X3 = 2 ("f16x2"): V2 [k][tile][2C] = [ h | l ] only -- the operand of wino_gemm_f16x2_kernel / wino_gemm_f16x2_w8_kernel (gq_wino_gemm.h), which form the
// three products itself (4 instead of 6 bytes per element).
template <int F16X3, typename VEC>
__device__ __forceinline__ void wino_store_v(void *V, int k, long tiles, long tile, int CV, int q, VEC v, float scale) {
  constexpr int VW = sizeof(VEC) / sizeof(float);          // channels per thread (4, or 2 in the two-channel kernels)
  typedef _Float16 hvec __attribute__((ext_vector_type(VW)));
  if constexpr (F16X3 != 0) {
    v = v * scale;
    hvec h, l;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      h[e] = (_Float16)v[e];
      l[e] = (_Float16)(v[e] - (float)h[e]);
    }
    if constexpr (F16X3 == 2) {
      hvec *row = reinterpret_cast<hvec *>(V) + ((long)k * tiles + tile) * (2 * CV);
      row[q] = h;
      row[CV + q] = l;
    } else {
      hvec *row = reinterpret_cast<hvec *>(V) + ((long)k * tiles + tile) * (3 * CV);
      row[q] = h;
      row[CV + q] = h;
      row[2 * CV + q] = l;
    }
  } else {
    reinterpret_cast<VEC *>(V)[((long)k * tiles + tile) * CV + q] = v;
  }
}

template <int F16X3>
__global__ __launch_bounds__(256) void wino_in_nhwc_kernel(const float *__restrict__ x, void *__restrict__ V, int H, int W,
                                                           int C4, long tiles, long total, float scale) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int q = (int)(t % C4);
    const long tile = t / C4;
    const int tw = (int)(tile % (W / 2));
    const long r = tile / (W / 2);
    const int th = (int)(r % (H / 2));
    const long b = r / (H / 2);
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sy = 2 * th - 1 + i, sx = 2 * tw - 1 + j;
        d[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sy >= 0 && sy < H && sx >= 0 && sx < W)
          d[i][j] = reinterpret_cast<const f32x4 *>(x)[((b * H + sy) * W + sx) * C4 + q];
      }
    f32x4 w[4][4];   // B^T d
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[0][j] = d[0][j] - d[2][j];
      w[1][j] = d[1][j] + d[2][j];
      w[2][j] = d[2][j] - d[1][j];
      w[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // (B^T d) B
      wino_store_v<F16X3>(V, 4 * i + 0, tiles, tile, C4, q, w[i][0] - w[i][2], scale);
      wino_store_v<F16X3>(V, 4 * i + 1, tiles, tile, C4, q, w[i][1] + w[i][2], scale);
      wino_store_v<F16X3>(V, 4 * i + 2, tiles, tile, C4, q, w[i][2] - w[i][1], scale);
      wino_store_v<F16X3>(V, 4 * i + 3, tiles, tile, C4, q, w[i][1] - w[i][3], scale);
    }
  }
}

// SiLU(GroupNorm(x)) of four channels: a, sh = the folded per-channel scale and shift.  The same arithmetic as
// gn_apply_nhwc_kernel, so the fused transforms write bit-for-bit the V of the two-pass route.
template <int SILU, typename VEC>
__device__ __forceinline__ VEC gn_act(VEC v, VEC a, VEC sh) {
  v = v * a + sh;
  if (SILU) {
#pragma unroll
    for (int e = 0; e < (int)(sizeof(VEC) / sizeof(float)); ++e) v[e] = silu_f32(v[e]);
  }
  return v;
}

// Input transform with the producer fused in: the conv input is GroupNorm(+SiLU) of x (unet.py:140-142, :146-149), so
// the normalisation is applied to the 16 loaded values on the fly (statistics from gn_stats / add_bias_stats) and the
// normalised tensor is never written: saves gn_apply's write and this kernel's read of it.  Zero padding applies to the
// ACTIVATED tensor, so out-of-bounds taps stay exactly 0.
template <int SILU, int F16X3>
__global__ __launch_bounds__(256) void wino_in_gn_nhwc_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta,
                                                              const float *__restrict__ pre_bias,
                                                              const int64_t *__restrict__ stats, void *__restrict__ V,
                                                              int H, int W, int C4, int cpg, double eps, long tiles,
                                                              long total, float scale) {
  const int groups = 4 * C4 / cpg;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int q = (int)(t % C4);
    const long tile = t / C4;
    const int tw = (int)(tile % (W / 2));
    const long r = tile / (W / 2);
    const int th = (int)(r % (H / 2));
    const long b = r / (H / 2);
    const int g = (4 * q) / cpg;
    const double n = (double)cpg * (double)H * (double)W;
    double st_s, st_ss;
    stat_load(stats + kStatWords * (b * groups + g), st_s, st_ss);
    const double mean = st_s / n;
    double var = st_ss / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + eps);
    f32x4 a, sh;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = 4 * q + k;
      const double pbk = pre_bias ? (double)pre_bias[c] : 0.0;
      a[k] = (float)(rstd * (double)gamma[c]);
      sh[k] = (float)((double)beta[c] + (pbk - mean) * rstd * (double)gamma[c]);
    }
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // no branch around the load: the 16 loads must all be in flight before the first activation (with the
        // activation inside an `if` hipcc keeps 16 load -> wait -> compute rounds: measured 965 vs 765 us); taps outside
        // the image read a clamped address and are zeroed afterwards
        const int sy = 2 * th - 1 + i, sx = 2 * tw - 1 + j;
        const int cy = sy < 0 ? 0 : (sy >= H ? H - 1 : sy), cx = sx < 0 ? 0 : (sx >= W ? W - 1 : sx);
        d[i][j] = reinterpret_cast<const f32x4 *>(x)[((b * H + cy) * W + cx) * C4 + q];
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sy = 2 * th - 1 + i, sx = 2 * tw - 1 + j;
        const float inb = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? 1.f : 0.f;
        d[i][j] = gn_act<SILU>(d[i][j], a, sh) * inb;
      }
    f32x4 w[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[0][j] = d[0][j] - d[2][j];
      w[1][j] = d[1][j] + d[2][j];
      w[2][j] = d[2][j] - d[1][j];
      w[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wino_store_v<F16X3>(V, 4 * i + 0, tiles, tile, C4, q, w[i][0] - w[i][2], scale);
      wino_store_v<F16X3>(V, 4 * i + 1, tiles, tile, C4, q, w[i][1] + w[i][2], scale);
      wino_store_v<F16X3>(V, 4 * i + 2, tiles, tile, C4, q, w[i][2] - w[i][1], scale);
      wino_store_v<F16X3>(V, 4 * i + 3, tiles, tile, C4, q, w[i][1] - w[i][3], scale);
    }
  }
}

// `mscale`: M came out of a GEMM on scaled operands (the f16x3 path): y = mscale * (A^T M A); 1 otherwise (a power of two).
__global__ __launch_bounds__(256) void wino_out_nhwc_kernel(const float *__restrict__ M, float *__restrict__ y, int H, int W,
                                                            int C4, long tiles, long total, float mscale) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int q = (int)(t % C4);
    const long tile = t / C4;
    const int tw = (int)(tile % (W / 2));
    const long r = tile / (W / 2);
    const int th = (int)(r % (H / 2));
    const long b = r / (H / 2);
    const f32x4 *mi = reinterpret_cast<const f32x4 *>(M) + tile * C4 + q;
    const long plane = tiles * C4;
    f32x4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) m[i][j] = mi[(4 * i + j) * plane];
    f32x4 u[2][4];   // A^T m
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u[0][j] = m[0][j] + m[1][j] + m[2][j];
      u[1][j] = m[1][j] - m[2][j] - m[3][j];
    }
    f32x4 *o = reinterpret_cast<f32x4 *>(y) + ((b * H + 2 * th) * W + 2 * tw) * C4 + q;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      o[(long)i * W * C4] = (u[i][0] + u[i][1] + u[i][2]) * mscale;
      o[(long)i * W * C4 + C4] = (u[i][1] - u[i][2] - u[i][3]) * mscale;
    }
  }
}

// ---- Winograd F(4x4, 3x3): 6x6 input tiles (stride 4), 36 GEMMs, 4x4 output tiles: 36 multiplies per 16 outputs (4x
// fewer than direct, 1.78x fewer than F(2x2,3x3)) and V / M are 2.25x the activation instead of 4x.  Larger transform
// constants (up to 8) cost ~10x the rounding error of F(2x2,3x3): used in the decoder only.
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
template <typename V>
__device__ __forceinline__ void wino4_bt(const V (&d)[6], V (&o)[6]) {
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
  o[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
  o[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
  o[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
template <typename V>
__device__ __forceinline__ void wino4_at(const V (&m)[6], V (&o)[4]) {
  const V s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = m[0] + s12 + s34;
  o[1] = d12 + 2.f * d34;
  o[2] = s12 + 4.f * s34;
  o[3] = d12 + 8.f * d34 + m[5];
}

template <int F16X3>
__global__ __launch_bounds__(256) void wino4_in_nhwc_kernel(const float *__restrict__ x, void *__restrict__ V, int H, int W,
                                                            int C4, long tiles, long total, float scale) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int q = (int)(t % C4);
    const long tile = t / C4;
    const int tw = (int)(tile % (W / 4));
    const long r = tile / (W / 4);
    const int th = (int)(r % (H / 4));
    const long b = r / (H / 4);
    f32x4 w[6][6];   // B^T d, column by column
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f32x4 col[6], o[6];
      const int sx = 4 * tw - 1 + j;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int sy = 4 * th - 1 + i;
        col[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sy >= 0 && sy < H && sx >= 0 && sx < W)
          col[i] = reinterpret_cast<const f32x4 *>(x)[((b * H + sy) * W + sx) * C4 + q];
      }
      wino4_bt(col, o);
#pragma unroll
      for (int i = 0; i < 6; ++i) w[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {   // (B^T d) B, row by row
      f32x4 o[6];
      wino4_bt(w[i], o);
#pragma unroll
      for (int j = 0; j < 6; ++j) wino_store_v<F16X3>(V, 6 * i + j, tiles, tile, C4, q, o[j], scale);
    }
  }
}

// wino4_in_nhwc_kernel with the producer fused in (see wino_in_gn_nhwc_kernel): every pixel is activated by the 2.25 tiles
// that overlap it.  VW = channels per thread: with 4 the 36 x 4 values of a tile plus the activation's temporaries take
// ~250 registers (two waves per SIMD, the activations' VALU time shows: 399 vs 304 us for the plain transform); with 2
// (whenever a 256-thread block still spans whole pixels) twice the waves hide it.
template <int SILU, int F16X3, int VW>
__global__ __launch_bounds__(256) void wino4_in_gn_nhwc_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               const float *__restrict__ pre_bias,
                                                               const int64_t *__restrict__ stats, void *__restrict__ V,
                                                               int H, int W, int CV, int cpg, double eps, long tiles,
                                                               long total, float scale) {
  typedef float vec __attribute__((ext_vector_type(VW)));
  const int groups = VW * CV / cpg;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int q = (int)(t % CV);
    const long tile = t / CV;
    const int tw = (int)(tile % (W / 4));
    const long r = tile / (W / 4);
    const int th = (int)(r % (H / 4));
    const long b = r / (H / 4);
    const int g = (VW * q) / cpg;
    const double n = (double)cpg * (double)H * (double)W;
    double st_s, st_ss;
    stat_load(stats + kStatWords * (b * groups + g), st_s, st_ss);
    const double mean = st_s / n;
    double var = st_ss / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + eps);
    vec a, sh;
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      const int c = VW * q + k;
      const double pbk = pre_bias ? (double)pre_bias[c] : 0.0;
      a[k] = (float)(rstd * (double)gamma[c]);
      sh[k] = (float)((double)beta[c] + (pbk - mean) * rstd * (double)gamma[c]);
    }
    vec w[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      vec col[6], o[6];
      const int sx = 4 * tw - 1 + j;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int sy = 4 * th - 1 + i;   // branch-free: see wino_in_gn_nhwc_kernel
        const int cy = sy < 0 ? 0 : (sy >= H ? H - 1 : sy), cx = sx < 0 ? 0 : (sx >= W ? W - 1 : sx);
        col[i] = reinterpret_cast<const vec *>(x)[((b * H + cy) * W + cx) * CV + q];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int sy = 4 * th - 1 + i;
        const float inb = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? 1.f : 0.f;
        col[i] = gn_act<SILU>(col[i], a, sh) * inb;
      }
      wino4_bt(col, o);
#pragma unroll
      for (int i = 0; i < 6; ++i) w[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      vec o[6];
      wino4_bt(w[i], o);
#pragma unroll
      for (int j = 0; j < 6; ++j) wino_store_v<F16X3>(V, 6 * i + j, tiles, tile, CV, q, o[j], scale);
    }
  }
}

__global__ __launch_bounds__(256) void wino4_out_nhwc_kernel(const float *__restrict__ M, float *__restrict__ y, int H, int W,
                                                             int C4, long tiles, long total, float mscale) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int q = (int)(t % C4);
    const long tile = t / C4;
    const int tw = (int)(tile % (W / 4));
    const long r = tile / (W / 4);
    const int th = (int)(r % (H / 4));
    const long b = r / (H / 4);
    const f32x4 *mi = reinterpret_cast<const f32x4 *>(M) + tile * C4 + q;
    const long plane = tiles * C4;
    f32x4 u[4][6];   // A^T m, column by column
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f32x4 col[6], o[4];
#pragma unroll
      for (int i = 0; i < 6; ++i) col[i] = mi[(6 * i + j) * plane];
      wino4_at(col, o);
#pragma unroll
      for (int i = 0; i < 4; ++i) u[i][j] = o[i];
    }
    f32x4 *out = reinterpret_cast<f32x4 *>(y) + ((b * H + 4 * th) * W + 4 * tw) * C4 + q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 o[4];
      wino4_at(u[i], o);
#pragma unroll
      for (int j = 0; j < 4; ++j) out[((long)i * W + j) * C4] = o[j] * mscale;
    }
  }
}

// Output transform with the ResnetBlock's tail fused in (unet.py:149-153): y = A^T M A + bias[c] (+ res), and the GroupNorm
// statistics of y for the block that follows (as add_bias_stats_nhwc_kernel).  A block owns a range of tiles of ONE
// image; thread -> channel quad q = tid % C4 (one GroupNorm group), tile lane = tid / C4.  T = 2: F(2x2,3x3), 4: F(4x4,3x3).
// VW = channels per thread: 4 (16-byte accesses), or 2 for F(4x4,3x3) -- with 4 the 36 loads of M, the 24 intermediate
// values and the 16 residual loads of a tile do not fit 256 registers, the residual is loaded late, one wave per SIMD
// waits for it 16 times (585 us at 16 x 256 x 256 x 128, 3.9 TB/s); with 2 everything is in flight at once at two waves
// per SIMD and an access of a wave is still whole 128-byte lines (64 lanes x 8 bytes = the 128 channels of a pixel).
template <int T, int VW>
__global__ __launch_bounds__(256) void wino_out_res_nhwc_kernel(const float *__restrict__ M, const float *__restrict__ res,
                                                                const float *__restrict__ bias, float *__restrict__ y,
                                                                int64_t *__restrict__ stats, int H, int W, int CV,
                                                                int cpg, long tiles, int slabs, float mscale) {
  typedef float vec __attribute__((ext_vector_type(VW)));
  constexpr int NI = T + 2;   // transform size (4 or 6)
  __shared__ int64_t red[kStatWords * 64];   // per-group statistics records (gq_stats.h)
  const int groups = VW * CV / cpg, lanes = 256 / CV;
  const long b = blockIdx.x / slabs;
  const int slab = blockIdx.x % slabs;
  const long tpi = (long)(H / T) * (W / T);           // tiles per image
  const long per = (tpi + slabs - 1) / slabs;
  const long lo = slab * per, hi = lo + per < tpi ? lo + per : tpi;
  const int q = threadIdx.x % CV, tl = threadIdx.x / CV;
  for (int w = threadIdx.x; w < kStatWords * groups; w += 256) red[w] = 0;
  __syncthreads();
  vec pb = (vec)(0.f);
  if (bias) pb = reinterpret_cast<const vec *>(bias)[q];
  const long plane = tiles * CV;
  float s = 0.f, ss = 0.f;
  for (long ti = lo + tl; ti < hi; ti += lanes) {
    const int tw = (int)(ti % (W / T)), th = (int)(ti / (W / T));
    const vec *mi = reinterpret_cast<const vec *>(M) + (b * tpi + ti) * CV + q;
    const long pix0 = ((b * H + (long)T * th) * W + (long)T * tw) * CV + q;
    vec r[T][T];
    if (res) {   // wave-uniform; issued ahead of M so that nothing waits for it at the end
#pragma unroll
      for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) r[i][j] = reinterpret_cast<const vec *>(res)[pix0 + ((long)i * W + j) * CV];
    } else {
#pragma unroll
      for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) r[i][j] = (vec)(0.f);
    }
    vec u[T][NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if constexpr (T == 4) {
        vec col[6], o[4];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = mi[(6 * i + j) * plane];
        wino4_at(col, o);
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i][j] = o[i];
      } else {
        const vec m0 = mi[(0 + j) * plane], m1 = mi[(4 + j) * plane], m2 = mi[(8 + j) * plane], m3 = mi[(12 + j) * plane];
        u[0][j] = m0 + m1 + m2;
        u[1][j] = m1 - m2 - m3;
      }
    }
#pragma unroll
    for (int i = 0; i < T; ++i) {
      vec o[T];
      if constexpr (T == 4) {
        wino4_at(u[i], o);
      } else {
        o[0] = u[i][0] + u[i][1] + u[i][2];
        o[1] = u[i][1] - u[i][2] - u[i][3];
      }
#pragma unroll
      for (int j = 0; j < T; ++j) {
        const long off = pix0 + ((long)i * W + j) * CV;
        vec v = o[j] * mscale + pb;
        if (res) v = v + r[i][j];
        reinterpret_cast<vec *>(y)[off] = v;
        if constexpr (VW == 4) {
          s += (v.x + v.y) + (v.z + v.w);
          ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        } else {
          s += v.x + v.y;
          ss += v.x * v.x + v.y * v.y;
        }
      }
    }
  }
  const int g = (VW * q) / cpg;
  stat_add_f32(red + kStatWords * g, s, ss);
  __syncthreads();
  for (int w = threadIdx.x; w < kStatWords * groups; w += 256) stat_flush_word(stats + kStatWords * (b * groups) + w, red[w]);
}

// scales[0] = v_scale = the largest power of two with amp * bound * v_scale <= 32768, bound = sqrt(max over (image,
// group) of the sum of squares) >= max|x| (rigorous: the L2 norm of a group bounds its largest element), from the
// GroupNorm statistics [2 * n_bg] (sum, sum of squares) the producer of x left behind; scales[1] = 1 / (v_scale * u_scale),
// the factor that takes the GEMM result back.  One wave.
__global__ __launch_bounds__(64) void f16_scales_from_stats_kernel(const int64_t *__restrict__ stats, int n_bg, float amp,
                                                                   float u_scale, float *__restrict__ scales) {
  double m = 0.0;
  for (int i = threadIdx.x; i < n_bg; i += 64) {
    double sum_unused, ss;
    stat_load(stats + kStatWords * i, sum_unused, ss);
    m = (ss != ss) ? __builtin_inf() : (ss > m ? ss : m);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double other = __shfl_xor(m, o);
    m = other > m ? other : m;
  }
  if (threadIdx.x == 0) {
    const double bound = sqrt(m) * (double)amp;
    int e = 14;                                        // never scale UP by more than 2^14
    if (bound > 0.0 && bound < 1e300) {
      int eb;
      (void)frexp(32768.0 / bound, &eb);               // 32768 / bound = f * 2^eb, f in [0.5, 1)
      e = eb - 1 < 14 ? eb - 1 : 14;
    }
    const float vs = (float)ldexp(1.0, e);
    scales[0] = vs;
    scales[1] = (float)(1.0 / ((double)vs * (double)u_scale));
  }
}

// y = a + b (+ bias[c]): the residual add of a ResnetBlock with the pending conv biases folded in.
__global__ __launch_bounds__(256) void add_bias_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                       const float *__restrict__ bias, float *__restrict__ y, int C,
                                                       long HW, long total4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4 *>(a)[i] + reinterpret_cast<const f32x4 *>(b)[i];
    if (bias) v = v + bias[(int)(((i * 4) / HW) % C)];
    reinterpret_cast<f32x4 *>(y)[i] = v;
  }
}

// ---- fused GroupNorm (+SiLU), NCHW fp32 --------------------------------------------
// stats: each block reduces a contiguous slice of one (b, g) chunk; fp32 per-thread partials,
// fp64 across threads/blocks (two atomics per block).
// pre_bias (nullable, [C]): a per-channel bias still pending on x (the producing conv ran without
// its bias); it is added on the fly so the separate bias pass disappears.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float *__restrict__ x, const float *__restrict__ pre_bias,
                                                       int64_t *__restrict__ stats, long chunk, int slices, long HW,
                                                       int cpg, int groups) {
  const long bg = blockIdx.x / slices;
  const int sl = blockIdx.x % slices;
  const long per = ((chunk / 4 + slices - 1) / slices) * 4;      // floats per slice (multiple of 4)
  const long lo = sl * per, hi = lo + per < chunk ? lo + per : chunk;
  const float *base = x + bg * chunk;
  const int c0 = (int)(bg % groups) * cpg;
  float s = 0.f, q = 0.f;
  for (long i = lo + threadIdx.x * 4; i + 3 < hi; i += 256 * 4) {
    f32x4 v = *reinterpret_cast<const f32x4 *>(base + i);
    if (pre_bias) v = v + pre_bias[c0 + (int)(i / HW)];
    s += (v.x + v.y) + (v.z + v.w);
    q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  double ds = (double)s, dq = (double)q;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ds += __shfl_xor(ds, o);
    dq += __shfl_xor(dq, o);
  }
  __shared__ double sh[8];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[wave] = ds; sh[4 + wave] = dq; }
  __syncthreads();
  if (threadIdx.x == 0)   // the block's sums in a fixed order, then order-independent across blocks (gq_stats.h)
    stat_add(stats + kStatWords * bg, (sh[0] + sh[1]) + (sh[2] + sh[3]), (sh[4] + sh[5]) + (sh[6] + sh[7]));
}

template <int SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float *__restrict__ y,
                                                       const int64_t *__restrict__ stats,
                                                       const float *__restrict__ pre_bias, int C, long HW, int cpg,
                                                       double eps, int segs) {
  const long row = blockIdx.x / segs;          // (b, c)
  const int seg = blockIdx.x % segs;
  const int c = (int)(row % C);
  const long b = row / C;
  const long bg = b * (C / cpg) + c / cpg;
  const double n = (double)cpg * (double)HW;
  double st_s, st_ss;
  stat_load(stats + kStatWords * bg, st_s, st_ss);
  const double mean = st_s / n;
  double var = st_ss / n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const double rstd = 1.0 / sqrt(var + eps);
  const float a = (float)(rstd * (double)gamma[c]);
  const double pb = pre_bias ? (double)pre_bias[c] : 0.0;
  const float sh = (float)((double)beta[c] + (pb - mean) * rstd * (double)gamma[c]);
  const long per = ((HW / 4 + segs - 1) / segs) * 4;
  const long lo = seg * per, hi = lo + per < HW ? lo + per : HW;
  const float *xi = x + row * HW;
  float *yo = y + row * HW;
  for (long i = lo + threadIdx.x * 4; i + 3 < hi; i += 256 * 4) {
    f32x4 v = *reinterpret_cast<const f32x4 *>(xi + i);
    v = v * a + sh;
    if (SILU) {
      v.x = silu_f32(v.x);
      v.y = silu_f32(v.y);
      v.z = silu_f32(v.z);
      v.w = silu_f32(v.w);
    }
    *reinterpret_cast<f32x4 *>(yo + i) = v;
  }
}

// ---- single-head attention of the deepest level (pit/modules/unet.py:185-206) on the fp16 matrix cores, fp32 results:
// softmax(q k^T c^-1/2) v as TWO library fp16 GEMMs with fp32 accumulation whose K axes carry the three products of
// two-term fp16 splits (the scheme of the Winograd GEMMs) instead of two fp32 GEMMs (a split-bf16 emulation on gfx950 at
// ~120 TFLOP/s).  attn_split_qkv_kernel prepares the operands of both GEMMs from the fused q|k|v projection; the softmax
// between them writes its result directly as the split operand of the second GEMM.
//   qkv [B][L][3C] fp32 (q | k | v per token)  ->  Q3 [B][L][3C] = [q_h | q_h | q_l] of q * sq
//                                                   K3 [B][L][3C] = [k_h | k_l | k_h] of k * sq     (S' = Q3 K3^T = sq^2 q k^T)
//                                                   V3 [B][3L][C] = [v_h ; v_l ; v_h] of v * sv     (rows stacked along K)
// thread = 4 channels of one token.
__global__ __launch_bounds__(256) void attn_split_qkv_kernel(const float *__restrict__ qkv, _Float16 *__restrict__ Q3,
                                                             _Float16 *__restrict__ K3, _Float16 *__restrict__ V3, long L,
                                                             int C4, float sq, float sv, long total) {
  const int C = 4 * C4;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int q4 = (int)(t % C4);
    const long tok = t / C4;            // b * L + l
    const long b = tok / L, l = tok % L;
    const float *src = qkv + tok * 3 * C + 4 * q4;
    f32x4 q = *reinterpret_cast<const f32x4 *>(src) * sq;
    f32x4 k = *reinterpret_cast<const f32x4 *>(src + C) * sq;
    f32x4 v = *reinterpret_cast<const f32x4 *>(src + 2 * C) * sv;
    f16x4 qh, ql, kh, kl, vh, vl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qh[e] = (_Float16)q[e]; ql[e] = (_Float16)(q[e] - (float)qh[e]);
      kh[e] = (_Float16)k[e]; kl[e] = (_Float16)(k[e] - (float)kh[e]);
      vh[e] = (_Float16)v[e]; vl[e] = (_Float16)(v[e] - (float)vh[e]);
    }
    f16x4 *qo = reinterpret_cast<f16x4 *>(Q3 + tok * 3 * C) + q4;
    qo[0] = qh; qo[C4] = qh; qo[2 * C4] = ql;
    f16x4 *ko = reinterpret_cast<f16x4 *>(K3 + tok * 3 * C) + q4;
    ko[0] = kh; ko[C4] = kl; ko[2 * C4] = kh;
    f16x4 *vo = reinterpret_cast<f16x4 *>(V3 + (b * 3 * L + l) * C) + q4;
    vo[0] = vh; vo[L * C4] = vl; vo[2 * L * C4] = vh;
  }
}

// Row softmax of S' * factor (factor = c^-1/2 / sq^2) written as the split operand of the second GEMM:
// P3 [rows][3L] = [p_h | p_h | p_l] of p * 2^14 (p <= 1).  One wave per row, the row in registers (L = 64 * NPL <= 4096).
template <int NPL>
__global__ __launch_bounds__(256) void attn_softmax_split_kernel(const float *__restrict__ S, _Float16 *__restrict__ P3,
                                                                 long rows, float factor) {
  constexpr int L = 64 * NPL;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *s = S + row * L;
  float v[NPL];
  float m = -__builtin_inff();
  if constexpr (NPL % 4 == 0) {      // lane owns runs of 4 consecutive elements: 16-byte loads, 8-byte stores
#pragma unroll
    for (int i = 0; i < NPL / 4; ++i) {
      const f32x4 x = *reinterpret_cast<const f32x4 *>(s + (i * 64 + lane) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[4 * i + e] = x[e] * factor; m = fmaxf(m, v[4 * i + e]); }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NPL; ++i) { v[i] = s[i * 64 + lane] * factor; m = fmaxf(m, v[i]); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) { v[i] = expf(v[i] - m); sum += v[i]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 16384.0f / sum;
  _Float16 *p = P3 + row * 3 * L;
  if constexpr (NPL % 4 == 0) {
#pragma unroll
    for (int i = 0; i < NPL / 4; ++i) {
      f16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = v[4 * i + e] * inv;
        // opaque: hipcc otherwise forms h as v_fma_mixlo_f16(v, inv) -- ONE rounding of the exact product -- for the residual
        // while the stored h is the conversion of the fp32 product: at double-rounding ties h + l was one fp16 ulp off
        asm volatile("" : "+v"(x));
        h[e] = (_Float16)x; l[e] = (_Float16)(x - (float)h[e]);
      }
      f16x4 *o = reinterpret_cast<f16x4 *>(p) + i * 64 + lane;
      o[0] = h; o[L / 4] = h; o[2 * (L / 4)] = l;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      float x = v[i] * inv;
      asm volatile("" : "+v"(x));
      const _Float16 h = (_Float16)x;
      p[i * 64 + lane] = h; p[L + i * 64 + lane] = h; p[2 * L + i * 64 + lane] = (_Float16)(x - (float)h);
    }
  }
}

// ---- content checksums of a module's parameters (the guard of the weight-derived caches, pit_hip/modules/unet.py) ----------
// The conv stack caches data derived from the weights (Winograd U matrices, fp16 splits in MFMA operand order, rigorous
// operand bounds), keyed on (data_ptr, _version).  A write through `param.data` changes neither, so the ONLY way to notice it
// is to look at the bytes: one launch per forward sums a position-salted hash of every 32-bit word of every parameter into one
// 64-bit word per tensor (integer atomics: order-independent); the module compares the sums with those its caches were built
// from and, on a difference, rebuilds the caches and runs the forward again.  Reads every parameter once: ~0.3 GB per step at
// config 2, ~0.2 % of the step.
//   table[t] = {pointer, 32-bit words}; grid = (tensors, kChecksumSlices); sums[t] zeroed by the caller.
struct ChecksumEntry {
  const unsigned *ptr;
  long words;
};
constexpr int kChecksumSlices = 32;
__global__ __launch_bounds__(256) void checksum_tensors_kernel(const ChecksumEntry *__restrict__ table,
                                                               unsigned long long *__restrict__ sums) {
  const ChecksumEntry e = table[blockIdx.x];
  unsigned long long acc = 0ull;
  const long n4 = e.words / 4;
  const u32x4_t *p4 = reinterpret_cast<const u32x4_t *>(e.ptr);
  for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < n4; i += (long)kChecksumSlices * 256) {
    const u32x4_t v = p4[i];
    const unsigned long long salt = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
    acc += ((unsigned long long)v.x ^ salt) * 0x85EBCA77C2B2AE63ull + ((unsigned long long)v.y ^ (salt >> 7)) * 0xC2B2AE3D27D4EB4Full;
    acc += ((unsigned long long)v.z ^ (salt >> 13)) * 0x165667B19E3779F9ull + ((unsigned long long)v.w ^ (salt >> 29)) * 0x27D4EB2F165667C5ull;
  }
  if (blockIdx.y == 0) {   // the tail (words % 4) and nothing else
    for (long i = 4 * n4 + threadIdx.x; i < e.words; i += 256)
      acc += ((unsigned long long)e.ptr[i] ^ ((unsigned long long)i * 0x9E3779B97F4A7C15ull)) * 0xFF51AFD7ED558CCDull;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0 && acc != 0ull) atomicAdd(&sums[blockIdx.x], acc);
}

}  // namespace gqhip
