// gq_gauss.h -- GQ2's Gaussian branch in eval (pit/quantization/gaussian.py:211-271): statistics of the per-row KL bits + the lambda state.
//
// The first launch of the fused call (gq_prep.h) leaves kl2row [rows]; ONE 256-thread block reduces it in a fixed order to mean / min /
// max and the re-weighted loss, and advances the adaptive lambda state exactly as the reference's Python does -- in fp64, on the device,
// so the forward needs no host read at all (the reference pays three bool(tensor) syncs per forward, gaussian.py:243-253).
//   scalars (64 B, 8-byte aligned): float[0..3] = { kl_loss, bits-mean, bits-min, bits-max };
//                                   double at byte 32: { lam, lam_min, lam_max } AFTER the update (what info["lam"...] reports).
// Where the block runs: as ONE EXTRA BLOCK of the re-rank launch -- block 0, so that it is dispatched first and runs beside the re-rank's
// blocks (as the LAST block it started when they were done and was the launch's tail: +10 us) -- at dims 8 / 16 / 32, where its input is
// two launches old by then and the call has no fourth launch (gq_rerank.h); or as its own one-block launch behind the other paths
// (dim 4's search, dims without a filter).  Either way the same function, the same order of additions: bit-identical results.
#pragma once
#include "gq_common.h"

namespace gqhip {

__device__ __forceinline__ void gauss_stats_block(const GaussStatsParams &p) {
#pragma clang fp contract(off)
  constexpr int NT = 256, VL = 4;                       // 256 threads act as 1024 virtual lanes: lane v takes rows v, v + 1024, ...
  const int tid = threadIdx.x;
  const double lam = p.lam_state[0], lam_min = p.lam_state[1], lam_max = p.lam_state[2];
  const float w_ge = (float)lam_max, w_le = (float)lam_min;
  double sum[VL], wsum[VL];
  float mn[VL], mx[VL];
  int nani = 0;
#pragma unroll
  for (int q = 0; q < VL; ++q) { sum[q] = 0.0; wsum[q] = 0.0; mn[q] = __builtin_inff(); mx[q] = -__builtin_inff(); }
  auto take = [&](int q, float k) {
    nani |= (k != k) ? 1 : 0;
    sum[q] += (double)k;
    mn[q] = __builtin_fminf(mn[q], k);
    mx[q] = __builtin_fmaxf(mx[q], k);
    // ge * kl2 + eq * kl2 + le * kl2 with the reference's fp32 products (gaussian.py:233-240)
    const float ge = (k > p.thr_hi ? 1.0f : 0.0f) * w_ge;
    const float eq = (k <= p.thr_hi ? 1.0f : 0.0f) * (k >= p.thr_lo ? 1.0f : 0.0f);
    const float le = (k < p.thr_lo ? 1.0f : 0.0f) * w_le;
    float e = ge * k;
    e = e + eq * k;
    e = e + le * k;
    wsum[q] += (double)e;
  };
  // rows in slabs of 1024 (one per virtual lane), four slabs' loads in flight at once: the block is one latency chain otherwise
  // (virtual lane v still adds its rows in ascending order: the result does not depend on the unrolling)
  constexpr int UN = 2;
  long base = 0;
  for (; base + (long)UN * NT * VL <= p.rows; base += (long)UN * NT * VL) {
    float k[UN][VL];
#pragma unroll
    for (int i = 0; i < UN; ++i)
#pragma unroll
      for (int q = 0; q < VL; ++q) k[i][q] = p.kl2row[base + (long)i * NT * VL + q * NT + tid];
#pragma unroll
    for (int i = 0; i < UN; ++i)
#pragma unroll
      for (int q = 0; q < VL; ++q) take(q, k[i][q]);
  }
  for (; base < p.rows; base += NT * VL)
#pragma unroll
    for (int q = 0; q < VL; ++q) {
      const long r = base + q * NT + tid;
      if (r < p.rows) take(q, p.kl2row[r]);
    }
  // 64-lane shuffle trees per (wave, virtual quarter), then the 16 results in the order (quarter, wave): a fixed tree
  __shared__ double s_a[16], s_b[16];
  __shared__ float s_mn[16], s_mx[16];
  __shared__ int s_nan[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nani |= __shfl_xor(nani, o);
#pragma unroll
  for (int q = 0; q < VL; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sum[q] += __shfl_xor(sum[q], o);
      wsum[q] += __shfl_xor(wsum[q], o);
      mn[q] = __builtin_fminf(mn[q], __shfl_xor(mn[q], o));
      mx[q] = __builtin_fmaxf(mx[q], __shfl_xor(mx[q], o));
    }
    if ((tid & 63) == 0) { const int w = q * 4 + (tid >> 6); s_a[w] = sum[q]; s_b[w] = wsum[q]; s_mn[w] = mn[q]; s_mx[w] = mx[q]; }
  }
  if ((tid & 63) == 0) s_nan[tid >> 6] = nani;
  __syncthreads();
  if (tid != 0) return;
  double a = s_a[0], b = s_b[0];
  float fmn = s_mn[0], fmx = s_mx[0];
  for (int w = 1; w < 16; ++w) {
    a += s_a[w];
    b += s_b[w];
    fmn = __builtin_fminf(fmn, s_mn[w]);
    fmx = __builtin_fmaxf(fmx, s_mx[w]);
  }
  const bool any_nan = (s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3]) != 0;
  const float qnan = __builtin_nanf("");
  const float mean = (float)(a / (double)p.rows);
  const float kmin = any_nan ? qnan : fmn, kmax = any_nan ? qnan : fmx;                 // torch.min / max propagate NaN
  const float wmean = (float)(b / (double)p.rows);
  const float kl_loss = wmean * (float)lam;                                             // torch.mean(kl_loss) * self.lam
  double l = lam, lmin = lam_min, lmax = lam_max;
  const double f = p.lam_factor;
  l = mean > p.log2n ? l * f : l / f;
  if (kmax > p.thr_hi) lmax = lmax * f;
  else if (p.lam_max_decreases) lmax = lmax / f;
  lmax = lmax < p.lam_hi ? lmax : p.lam_hi;       // max(min(lam_max, hi), 1.0)
  lmax = lmax > 1.0 ? lmax : 1.0;
  lmin = kmin < p.thr_lo ? lmin / f : lmin * f;
  lmin = lmin < 1.0 ? lmin : 1.0;                 // max(min(lam_min, 1.0), lo)
  lmin = lmin > p.lam_lo ? lmin : p.lam_lo;
  float *fo = static_cast<float *>(p.scalars);
  fo[0] = kl_loss; fo[1] = mean; fo[2] = kmin; fo[3] = kmax;
  double *d = reinterpret_cast<double *>(static_cast<char *>(p.scalars) + 32);
  d[0] = l; d[1] = lmin; d[2] = lmax;
  p.lam_state[0] = l; p.lam_state[1] = lmin; p.lam_state[2] = lmax;
}

// its own launch (one block of 256 threads) behind the paths that have no re-rank launch
__global__ __launch_bounds__(256) void gauss_stats_finalize_kernel(const GaussStatsParams p) { gauss_stats_block(p); }

}  // namespace gqhip
