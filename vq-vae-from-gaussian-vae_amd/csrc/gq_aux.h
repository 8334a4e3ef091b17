// gq_aux.h -- the small HBM-bound kernels around the filter: codebook abs-max,
// workspace header init, operand prep from the encoder output, dequant gather,
// the reference-compatible score-matrix op, LFQ bit pack/unpack, index
// histogram and the u16 wire format.
#pragma once
#include "gq_common.h"
#include "gq_rerank.h"
#include "gq_gauss.h"

namespace gqhip {

// ---- dequant: zhat <- cb[idx] in the module layout -------------------------
struct DequantParams {
  const int64_t *idx;
  const float *cb;
  float *zhat;
  long rows;
  int dim, n;
  OutMap omap;
};

__global__ __launch_bounds__(256) void dequant_kernel(const DequantParams p) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.rows * p.dim) return;
  long row;
  int g;
  if (p.omap.mode == 1) {
    // walk the output [B, c, L] linearly so stores coalesce
    const long l = t % p.omap.L;
    const int ch = (int)((t / p.omap.L) % p.omap.c);
    const long b = t / ((long)p.omap.L * p.omap.c);
    int k;
    if (p.omap.grouping == 0) { g = ch / p.omap.K; k = ch % p.omap.K; } else { k = ch / p.dim; g = ch % p.dim; }
    row = (b * p.omap.L + l) * p.omap.K + k;
  } else {
    row = t / p.dim;
    g = (int)(t % p.dim);
  }
  long j = p.idx[out_idx_offset(p.omap, row)];
  j = j < 0 ? 0 : (j >= p.n ? p.n - 1 : j);
  p.zhat[out_zhat_offset(p.omap, row, g, p.dim)] = p.cb[j * p.dim + g];
}

// ---- compat op: the reference's score matrix --------------------------------
// gq_cuda.cu:12-40.  One thread per code column, ROWS rows per block; the code
// row lives in registers, mu / sd are wave-uniform (scalar loads), stores are
// 1 KiB coalesced runs along n.  The running sum is a float; `iv*iv` is
// contracted into the subtraction (nvcc's default -fmad), and the beta term is
// accumulated through double exactly as the CUDA source spells it.
// Division: iv = (n - mu) / sd is an IEEE fp32 divide in the CUDA source.  A hardware divide costs
// ~10 VALU instructions per (row, code, dim); instead the correctly rounded reciprocal y = RN(1/sd)
// is formed once per (row, dim) and each quotient is q = a*y, r = fma(-sd, q, a), q' = fma(r, y, q)
// (Markstein: with y correctly rounded and no over/underflow, q' = RN(a/sd)).  Rows whose sd (or
// whose numerators) could leave the safe exponent range take the hardware divide instead.
// BETA1: beta == 1.0 -- (float)((double)acc + (double)co2) equals the fp32 add acc + co2 exactly
// (double rounding is innocuous for the sum of two floats), so the fp64 detour is dropped.
template <int DIM, int ROWS, bool BETA1>
__global__ __launch_bounds__(256) void gq_scores_kernel(const float *__restrict__ mu,
                                                        const float *__restrict__ sd,
                                                        const float *__restrict__ cb,
                                                        float *__restrict__ out, int rows, int n,
                                                        double beta) {
#pragma clang fp contract(off)
  // codes per thread.  2 was measured slower, and so was 2 held as one f32x2 (v_pk_add/mul/fma_f32): on gfx950 a scalar
  // fp32 VALU op issues every ~2.7-3.2 cycles per SIMD and a packed one every ~4.7-5 (tools/valu_pk_peak.hip), so packing
  // buys < 15 % before the splat moves it needs; the kernel is bound by its 6 VALU ops per (row, code, dim).
  constexpr int CPT = 1;
  __shared__ float s_mu[ROWS][DIM], s_sd[ROWS][DIM], s_y[ROWS][DIM];
  __shared__ int s_fast[ROWS];
  const int j0 = blockIdx.x * (256 * CPT) + threadIdx.x;
  const int r0 = blockIdx.y * ROWS;
  if (threadIdx.x < ROWS) s_fast[threadIdx.x] = 1;
  __syncthreads();
  for (int t = threadIdx.x; t < ROWS * DIM; t += 256) {
    const int r = t / DIM, i = t % DIM;
    const int rr = r0 + r < rows ? r0 + r : rows - 1;
    const float m = mu[(long)rr * DIM + i], s = sd[(long)rr * DIM + i];
    s_mu[r][i] = m;
    s_sd[r][i] = s;
    s_y[r][i] = __fdiv_rn(1.0f, s);
    const float as = fabsf(s), am = fabsf(m);
    if (!(as > 1e-18f && as < 1e18f && am < 1e18f)) s_fast[r] = 0;   // also catches NaN
  }
  float nv[CPT][DIM], n2[CPT][DIM];
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int j = j0 + 256 * k;
    const int jc = j < n ? j : n - 1;
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      nv[k][i] = cb[(long)jc * DIM + i];
      n2[k][i] = nv[k][i] * nv[k][i];
      amax = __builtin_fmaxf(amax, fabsf(nv[k][i]));
    }
  }
  const bool code_ok = amax < 1e18f;   // false for inf / NaN codes too
  __syncthreads();
  for (int r = 0; r < ROWS && r0 + r < rows; ++r) {
    float acc[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) acc[k] = 0.0f;
    if (s_fast[r] && __all(code_ok)) {
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
        const float m = s_mu[r][i], sg = s_sd[r][i], y = s_y[r][i];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
          const float a = nv[k][i] - m;
          const float q = a * y;
          const float rem = __builtin_fmaf(-sg, q, a);
          const float iv = __builtin_fmaf(rem, y, q);
          acc[k] = __builtin_fmaf(-iv, iv, acc[k]);
          if constexpr (BETA1) acc[k] = acc[k] + n2[k][i];
          else acc[k] = (float)((double)acc[k] + (double)n2[k][i] * beta);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
          const float iv = __fdiv_rn(nv[k][i] - s_mu[r][i], s_sd[r][i]);
          acc[k] = __builtin_fmaf(-iv, iv, acc[k]);
          acc[k] = (float)((double)acc[k] + (double)n2[k][i] * beta);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int j = j0 + 256 * k;
      if (j < n) out[(long)(r0 + r) * n + j] = acc[k];
    }
  }
}

__global__ __launch_bounds__(256) void gq_scores_generic_kernel(const float *__restrict__ mu,
                                                                const float *__restrict__ sd,
                                                                const float *__restrict__ cb,
                                                                float *__restrict__ out, int dim,
                                                                int rows, int n, double beta) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (j >= n || r >= rows) return;
  float acc = 0.0f;
  for (int i = 0; i < dim; ++i) {
    const float co = cb[(long)j * dim + i];
    const float iv = __fdiv_rn(co - mu[(long)r * dim + i], sd[(long)r * dim + i]);
    acc = __builtin_fmaf(-iv, iv, acc);
    acc = (float)((double)acc + (double)(co * co) * beta);
  }
  out[(long)r * n + j] = acc;
}

// ---- LFQ (lfq.py:147-158, 210-228) ------------------------------------------
__global__ __launch_bounds__(256) void lfq_pack_kernel(const float *__restrict__ x,
                                                       int64_t *__restrict__ idx,
                                                       float *__restrict__ q, long rows, int nbits) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  int64_t v = 0;
  for (int i = 0; i < nbits; ++i) {
    const bool bit = x[r * nbits + i] > 0.0f;
    v = v * 2 + (bit ? 1 : 0);
    if (q) q[r * nbits + i] = bit ? 1.0f : -1.0f;
  }
  idx[r] = v;
}

__global__ __launch_bounds__(256) void lfq_unpack_kernel(const int64_t *__restrict__ idx,
                                                         float *__restrict__ q, long rows, int nbits) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * nbits) return;
  const long r = t / nbits;
  const int i = (int)(t % nbits);
  const int64_t v = idx[r];
  q[t] = ((v >> (nbits - 1 - i)) & 1) ? 1.0f : -1.0f;
}

// ---- FSQ (fsq.py:29-89) ------------------------------------------------------
struct FsqLevels {
  int n;
  int lev[16];
};

__global__ __launch_bounds__(256) void fsq_quantize_kernel(const float *__restrict__ z, FsqLevels L,
                                                           float *__restrict__ zhat,
                                                           int32_t *__restrict__ idx, long rows) {
#pragma clang fp contract(off)
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  int packed = 0;
  for (int l = 0; l < L.n; ++l) {
    const int lv = L.lev[l];
    const float half_l = (float)(lv - 1) * (1.0f + 1e-3f) / 2.0f;          // (levels-1)*(1+eps)/2
    const float offset = (lv % 2 == 0) ? 0.5f : 0.0f;
    const float shift = (float)atanh((double)(offset / half_l));
    const float x = z[r * L.n + l] + shift;
    const float b = (float)tanh((double)x) * half_l - offset;
    const float q = rintf(b);                                              // torch.round: half to even
    const int hw = lv / 2;
    if (zhat) zhat[r * L.n + l] = q / (float)hw;
    packed = packed * lv + (int)(q + (float)hw);
  }
  idx[r] = packed;
}

__global__ __launch_bounds__(256) void fsq_dequant_kernel(const int32_t *__restrict__ idx, FsqLevels L,
                                                          float *__restrict__ zhat, long rows) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  int v = idx[r];
  for (int l = L.n - 1; l >= 0; --l) {
    const int lv = L.lev[l];
    const int d = v % lv;
    v /= lv;
    const int hw = lv / 2;
    zhat[r * L.n + l] = (float)(d - hw) / (float)hw;
  }
}

// ---- index histogram + u16 wire format (eval.py:127,137-141,152-154) --------
__global__ __launch_bounds__(256) void hist_kernel(const int64_t *__restrict__ idx, long count, int n,
                                                   int *__restrict__ hist) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
    const int64_t v = idx[i];
    if (v >= 0 && v < n) atomicAdd(&hist[v], 1);
  }
}
__global__ __launch_bounds__(256) void to_u16_kernel(const int64_t *__restrict__ idx,
                                                     uint16_t *__restrict__ out, long count) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < count) out[i] = (uint16_t)idx[i];
}
__global__ __launch_bounds__(256) void from_u16_kernel(const uint16_t *__restrict__ in,
                                                       int64_t *__restrict__ idx, long count) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < count) idx[i] = (int64_t)in[i];
}


// ---- one rank's per-step record in ONE launch (round 4): per-image PSNR + the uint16 index wire format -----------------------
// rec (int32 words) = [ B per-image PSNRs as fp32 bits | indices as uint16 pairs, low half first ] -- pit_hip/eval_dist.py:StepRecord
// with n_metrics = 1, i.e. what eval.py:165-169 (get_psnr(zero_mean=True), pit/evaluations/psnr.py:17-28) and the index gather
// of eval.py:152-154 publish per batch.  Blocks [0, B * chunks): one contiguous chunk of one image, the squared differences in
// the reference's fp32 op order ((x + 1) 127.5 - (x_rec + 1) 127.5, squared), summed in fp64; the LAST block of an image (a ticket
// per image, reset for the next call) adds the chunk sums in chunk order -- a fixed order: the value is bit-reproducible -- and
// writes 20 log10(255 / sqrt(mse)).  The blocks after those pack the indices.
struct StepRecordParams {
  const float *x, *x_rec;     // [B, per_image] in the SAME dense layout
  const int64_t *idx;         // [n_idx]
  int *rec;                   // [B + (n_idx + 1) / 2]
  double *partial;            // [B, chunks]      (workspace)
  int *ticket;                // [B], all zero between calls (workspace)
  long per_image, n_idx;
  int B, chunks, psnr_blocks;
};
constexpr int kPsnrChunk = 256 * 4 * 8;      // floats per block: eight 16-byte loads per thread

__global__ __launch_bounds__(256) void step_record_kernel(const StepRecordParams p) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= p.psnr_blocks) {
    const long w = (long)(blockIdx.x - p.psnr_blocks) * 256 + tid;        // one packed word per thread
    if (2 * w < p.n_idx) {
      const unsigned lo = (unsigned)p.idx[2 * w] & 0xffffu;
      const unsigned hi = 2 * w + 1 < p.n_idx ? (unsigned)p.idx[2 * w + 1] & 0xffffu : 0u;
      p.rec[p.B + w] = (int)(lo | (hi << 16));
    }
    return;
  }
  const int b = blockIdx.x / p.chunks, ch = blockIdx.x % p.chunks;
  const long e0 = (long)ch * kPsnrChunk, e1 = e0 + kPsnrChunk < p.per_image ? e0 + kPsnrChunk : p.per_image;
  const float *xa = p.x + (long)b * p.per_image, *xb = p.x_rec + (long)b * p.per_image;
  double acc = 0.0;
  auto term = [&](float u, float v) {
#pragma clang fp contract(off)
    const float a = (u + 1.0f) * 127.5f, c = (v + 1.0f) * 127.5f;
    const float d = a - c;
    acc += (double)(d * d);
  };
  if ((p.per_image & 3) == 0) {
    for (long e = e0 + 4 * tid; e < e1; e += 1024) {
      const f32x4 u = *reinterpret_cast<const f32x4 *>(xa + e), v = *reinterpret_cast<const f32x4 *>(xb + e);
      term(u.x, v.x); term(u.y, v.y); term(u.z, v.z); term(u.w, v.w);
    }
  } else {
    for (long e = e0 + tid; e < e1; e += 256) term(xa[e], xb[e]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  __shared__ double sh[4];
  __shared__ int sh_last;
  if ((tid & 63) == 0) sh[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    p.partial[(long)b * p.chunks + ch] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __threadfence();
    sh_last = atomicAdd(&p.ticket[b], 1) == p.chunks - 1;
  }
  __syncthreads();
  if (!sh_last || tid != 0) return;
  __threadfence();
  double sum = 0.0;
  for (int k = 0; k < p.chunks; ++k) sum += __hip_atomic_load(&p.partial[(long)b * p.chunks + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float mse = (float)(sum / (double)p.per_image);
  const float psnr = (float)(20.0 * log10(255.0 / sqrt((double)mse)));      // identical images: +inf, like the reference
  p.rec[b] = __float_as_int(psnr);
  p.ticket[b] = 0;                             // ready for the next call on this workspace
}

// ---- VQ's codebook loss in eval (pit/quantization/vq.py:78-86): mean((z_q - z)^2) over every element --------------------------
// After the arg-min has left its indices: element (row, g) contributes (emb[idx[row]][g] - z[row][g])^2 formed in fp32 like the
// reference, summed in fp64 -- per thread in a fixed assignment, per block by a fixed tree, and the LAST block (a ticket in the
// workspace header, reset by the call's first launch) adds the block sums in block order: bit-reproducible.
//   loss[0] = mean + beta * mean (legacy) | beta * mean + mean, in fp32 like the reference;  loss[1] = mean.
struct VqLossParams {
  const float *zrows;       // [rows, dim] the row operands gq_prep_kernel left in the workspace
  const int64_t *idx;       // module layout
  const float *emb;         // [n, dim]
  float *loss;              // [2]
  WsHeader *hdr;
  long rows;
  int dim, n;
  float beta;
  int legacy;
  OutMap omap;
};

__global__ __launch_bounds__(256) void vq_loss_kernel(const VqLossParams p) {
#pragma clang fp contract(off)
  const int tid = threadIdx.x;
  double acc = 0.0;
  // one row per thread and pass: ONE index load, then the row's codeword and operands as 16-byte loads (dim % 4 == 0) -- the first cut
  // walked elements (an index load and an integer division per element: 47 us at 65 536 rows, latency-bound)
  const bool vec = (p.dim & 3) == 0;
  for (long row = (long)blockIdx.x * 256 + tid; row < p.rows; row += (long)gridDim.x * 256) {
    const long j = p.idx[out_idx_offset(p.omap, row)];
    const bool ok = j >= 0 && j < p.n;
    const float *e = p.emb + (ok ? j : 0) * p.dim, *z = p.zrows + row * p.dim;
    if (vec) {
      for (int g = 0; g < p.dim; g += 4) {
        const f32x4 ev = *reinterpret_cast<const f32x4 *>(e + g), zv = *reinterpret_cast<const f32x4 *>(z + g);
        const float d0 = ev.x - zv.x, d1 = ev.y - zv.y, d2 = ev.z - zv.z, d3 = ev.w - zv.w;
        acc += (double)(d0 * d0);
        acc += (double)(d1 * d1);
        acc += (double)(d2 * d2);
        acc += (double)(d3 * d3);
      }
    } else {
      for (int g = 0; g < p.dim; ++g) {
        const float d = e[g] - z[g];
        acc += (double)(d * d);
      }
    }
    if (!ok) acc = __builtin_nan("");
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  __shared__ double sh[4];
  __shared__ int sh_last;
  if ((tid & 63) == 0) sh[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    p.hdr->loss_part[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __threadfence();
    sh_last = atomicAdd(&p.hdr->loss_ticket, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!sh_last || tid >= 64) return;
  __threadfence();
  // the last block: the block sums in block order per lane (lane l: l, l + 64, ...), then a fixed shuffle tree
  double sum = 0.0;
  for (int k = tid; k < (int)gridDim.x; k += 64) sum += __hip_atomic_load(&p.hdr->loss_part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (tid != 0) return;
  const float m = (float)(sum / (double)(p.rows * p.dim));
  p.loss[0] = p.legacy ? m + p.beta * m : p.beta * m + m;
  p.loss[1] = m;
}

}  // namespace gqhip
