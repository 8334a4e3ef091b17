// gq_prep.h -- the first launch of every fused arg-max call: everything that is derived from the inputs once.
//
// Row blocks (one 256-thread block per 256 / DIM consecutive rows):
//   * FROM_Z: z [mu | logvar] in the module layout -> mu, sd = float(exp(double(0.5 lv))), lsd = float(log(double(sd)))
//     rows (pit/quantization/gaussian.py:62-81,122-123 / :273-287; the group permutes are index arithmetic here), and
//     zhat_noquant = mu + noise * sd straight into the module layout (gaussian.py:121);
//     otherwise the caller's (mu, sd[, lsd]) rows are used as they are.
//   * the filter's row operands A = beta/2 - 1/(2 sd^2), B = mu / sd^2 (fp64, rounded once), split into two bf16
//     terms and written as the row image of gq_filter_bf16.h;
//   * the same A | B as fp32 rows (the re-rank's fp32 pre-filter evaluates the expansion with exactly these operands);
//   * four per-row sums the re-rank's rounding bound is made of (so that no kernel after this one divides in fp64):
//       S0 = sum 1/sd^2, S1 = sum |mu|/sd^2, S2 = sum mu^2/sd^2, S3 = sum |log sd|      (VQ: S1 = sum |z|).
// Code blocks (the 256 blocks after the row blocks): the bf16 tile image of the codebook [n^2 | n] (h / l parts, in the
// LDS tile order of the filter) and max |cb| as one partial per block.  Both are rebuilt on EVERY call from the
// codebook the caller passes: nothing derived from a codebook outlives the call, so a codebook that was edited in
// place (by whatever means) can never meet a stale image or a stale bound.
// Block 0 also resets the workspace header for the kernels that follow on the stream.
#pragma once
#include "gq_common.h"
#include "gq_filter_bf16.h"
#include "gq_rerank.h"

namespace gqhip {

constexpr int kPrepCodeBlocks = kAbsmaxParts;   // one max|cb| partial per code block

struct PrepParams {
  // FROM_Z
  const float *z;            // [B, 2c, L] (BCHW) or [B, L, 2c] (BLC)
  const float *noise;        // module layout of zhat_noquant, or NULL
  float *zhat_noquant;       // [B, c, L] / [B, L, c], or NULL
  float *sd_layout;          // FROM_Z, optional: sd in the layout of zhat (GQ2's info["std"], gaussian.py:263-264)
  float *kl2row;             // FROM_Z, optional: [rows] KL divergence of the row's Gaussian to N(0, 1) in bits (gaussian.py:225-229)
  int ste_kind;              // -> header: the straight-through mix applied where zhat is stored (gq_common.h:WsHeader)
  const float *ste;
  float *pure;
  GaussStatsParams gs;       // -> header: rows > 0 when the re-rank launch of this call carries the statistics block (gq_gauss.h)
  float lv_min, lv_max;
  // rows: outputs when FROM_Z, inputs otherwise (lsd may then be NULL: lsd_out receives float(log(double(sd))))
  float *mu, *sd, *lsd;      // [rows, dim]
  float *lsd_out;            // !FROM_Z only, may be NULL
  double *rowsum;            // [rows, 4]
  float *coef;               // [rows, 2, dim]: the fp32 filter coefficients A | B (the re-rank's pre-filter reads them)
  u32x4 *rowimg;             // [rows][NVEC][2] or NULL (fp32 filter: no images)
  float *rowscale;           // [rows] (fp16 images only): 2^e_r, the power of two the row's coefficients were divided by
  float *rowaux;             // [rows, 8] (fp16 main-product filter only): the sums of its data-dependent bound (gq_rerank.h:
                             // M_well, P, Q, Rb, |B|^2, max|A|, max(|A|, |B|), 0), each rounded UP to fp32
  const float *cb;           // [n, dim]
  u32x4 *cbimg;              // [tiles_total + CT][NVEC][2][32] or NULL
  WsHeader *hdr;
  unsigned long long *cache_sums;   // codebook cache: the kAbsmaxParts slice hashes it was built from, or NULL (no cache in this call)
  int *cache_stale;                 // ... its `stale` word: set when a slice's hash differs (the builder that follows rebuilds)
  long rows;
  int n, tiles_total;
  int row_blocks;            // blocks [0, row_blocks) prepare rows, the kPrepCodeBlocks after them the codebook
  float beta;
  OutMap omap;               // module layout (mode 1 BCHW, 2 BLC; 0: plain rows)
};

// fp8 (OCP e4m3) of two floats, packed into the low 16 bits (values in range by construction; RNE)
__device__ __forceinline__ unsigned fp8x2(float a, float b) {
  return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}

// Power-of-two normalisation of a row's filter coefficients for the MIXED (fp16 + fp8) images: returns s = 2^-e with
// max|c| * s in [2^13, 2^14), or NaN when the largest coefficient is 0, non-finite or outside [2^-100, 2^100] (such a row
// is poisoned through its bound sums and decided by the exhaustive stages).
__device__ __forceinline__ float mixed_row_scale(const float *coef, int n) {
  float amax = 0.0f;
  bool nan = false;
  for (int i = 0; i < n; ++i) {
    const float a = fabsf(coef[i]);
    nan = nan || (a != a);
    amax = __builtin_fmaxf(amax, a);
  }
  if (nan || !(amax >= 7.8886090522101181e-31f) || !(amax <= 1.2676506002282294e30f)) return __builtin_nanf("");
  int ex;
  (void)frexpf(amax, &ex);            // amax = f * 2^ex, f in [0.5, 1)
  return ldexpf(1.0f, 14 - ex);       // amax * s in [2^13, 2^14)
}

// fp64 -> fp32, never below the argument (bound inputs are rounded up)
__device__ __forceinline__ float f32_up(double v) { return (float)(v * 1.0000002384185791); }   // (1 + 2^-22): covers the RNE error

// one element of kl2 = 1.4426 * 0.5 * (mu^2 + var - 1 - logvar) in the reference's fp32 op order (gaussian.py:225), lv already
// clamped; var = float(exp(double(lv))); 1.4426 * 0.5 is formed in Python first (a double), then cast to the tensor's dtype
__device__ __forceinline__ float kl_bits_term(float m, float lv) {
#pragma clang fp contract(off)
  const float var = (float)exp((double)lv);
  float t = m * m;
  t = t + var;
  t = t - 1.0f;
  t = t - lv;
  return (float)0.7213 * t;
}

// FK (filter kind): 0 = the split-bf16 images; 1 = MIXED (DIM 16, GQ only): the operand images of the fp16 + fp8 filter;
// 2 = F16: the fp16 images of the main-product-only filter (gq_filter_bf16.h), every MFMA dim, GQ and VQ.
template <int MODE, int DIM, bool FROM_Z, int FK = 0>
__global__ __launch_bounds__(256) void gq_prep_kernel(const PrepParams p) {
  static_assert(DIM == 4 || DIM == 8 || DIM == 16 || DIM == 32, "MFMA filter dims");
  constexpr bool MIXED = FK == 1, F16 = FK == 2;
  static_assert(!MIXED || (DIM == 16 && MODE == kModeGQ), "fp16 + fp8 images: dim 16, Gaussian score");
  constexpr bool PACKED = DIM == 4;
  constexpr int NV = PACKED ? 1 : DIM / 8;     // 8-slot groups per operand half
  constexpr int NVEC = F16 ? NV : (PACKED ? 2 : 2 * NV);    // 16-byte vectors per (code | row, half) in an image
  constexpr int RB = 256 / DIM;                // rows per block
  const int tid = threadIdx.x;

  if (blockIdx.x == 0 && tid == 0) {           // header for the kernels that follow on the stream
    p.hdr->fb_count = 0;
    p.hdr->reranked = 0ull;
    p.hdr->grid_leaves = 0ull;
    p.hdr->grid_next = 0;
    p.hdr->loss_ticket = 0;
    p.hdr->ste_kind = p.ste_kind;
    p.hdr->ste = p.ste;
    p.hdr->pure = p.pure;
    p.hdr->gs = p.gs;
  }

#if defined(GQHIP_ABL) && (GQHIP_ABL & 1024)   // diagnostic build (tools/abl_prep.sh): the code blocks do nothing
  if ((int)blockIdx.x >= p.row_blocks) return;
#endif
#if defined(GQHIP_ABL) && (GQHIP_ABL & 2048)   // diagnostic build: the row blocks do nothing
  if ((int)blockIdx.x < p.row_blocks) return;
#endif
  if ((int)blockIdx.x >= p.row_blocks) {
    // ------------------------------------------------------------------ codebook image + max |cb|
    const int cbk = blockIdx.x - p.row_blocks;
    float amax = 0.0f, r2max = 0.0f;
    unsigned long long hsum = 0ull;
    if (p.cbimg) {
      const long items = (long)p.tiles_total * 64;   // (tile, half, code)
      // A CACHED image (F16 with a codebook cache, gqhip.h): pass 0 only reads -- max |cb|, the norm bound and the content hash of this
      // block's slice of the codebook --; if the hash is the one this slice of the image was built from, the slice is current and
      // nothing is written; otherwise pass 1 rebuilds it (from L2-hot data) and the block restamps its hash.  Slices are disjoint, so
      // every block validates and repairs its own: no flag, no other kernel, and an edit of the codebook by any route is seen here.
      const bool cached = F16 && p.cache_sums != nullptr && p.cache_stale == nullptr;
      __shared__ unsigned long long s_h2[4];
      __shared__ int s_slice_current;
      for (int pass = 0; pass < (cached ? 2 : 1); ++pass) {
      const bool do_write = !cached || pass == 1;
      if (cached && pass == 1) {
        unsigned long long hh = hsum;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hh += __shfl_xor(hh, o);
        if ((tid & 63) == 0) s_h2[tid >> 6] = hh;
        __syncthreads();
        if (tid == 0) {
          const unsigned long long h = s_h2[0] + s_h2[1] + s_h2[2] + s_h2[3];
          s_slice_current = p.cache_sums[cbk] == h ? 1 : 0;
          if (!s_slice_current) p.cache_sums[cbk] = h;          // (restamped by the block that rewrites the slice below)
        }
        __syncthreads();
        if (s_slice_current) break;
      }
      for (long t = (long)cbk * 256 + tid; t < items; t += (long)kPrepCodeBlocks * 256) {
        const int tile = (int)(t >> 6), c = (int)t & 31, h = (int)(t >> 5) & 1;
        const long code = (long)tile * 32 + c;
        u32x4 *dst = p.cbimg + (long)tile * (NVEC * 64) + h * 32 + c;
        if constexpr (F16) {
          // slots [ squares of the dims | values of the dims ] as fp16; vector m, half h = slots 16 m + 8 h .. + 7 (the operand
          // of MFMA m; DIM 4: its 8 slots in half 0, zeros in half 1).  Every lane reads the whole code row: max |n| and
          // the squared norm |n|^2 (the norm bound of the re-rank) come from it.
          float nv[DIM];
          float r2 = 0.0f;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            nv[k] = code < p.n ? p.cb[code * DIM + k] : 0.0f;
            if (pass == 0) {
              const float a = fabsf(nv[k]);
              amax = (a != a) ? __builtin_inff() : __builtin_fmaxf(amax, a);
              r2 = __builtin_fmaf(nv[k], nv[k], r2);
              if (h == 0 && code < p.n) hsum += cb_hash_term(nv[k], code * DIM + k);
            }
          }
          if (pass == 0) r2max = (r2 != r2) ? __builtin_inff() : __builtin_fmaxf(r2max, r2);
          if (!do_write) continue;
          typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#pragma unroll
          for (int m = 0; m < NV; ++m) {
            h8 v;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int g = PACKED ? k : 16 * m + 8 * h + k;
              const float x = nv[g < DIM ? g : g - DIM];
              float t = g < DIM ? x * x : x;
              asm volatile("" : "+v"(t));          // the fp32 square is the feature (no fused multiply-convert)
              v[k] = (PACKED && h == 1) ? (_Float16)0.0f : (_Float16)t;
            }
            dst[(long)m * 64] = __builtin_bit_cast(u32x4, v);
          }
          continue;
        }
        if constexpr (MIXED) {
          // slots: [0, 16) squares of the dims, [16, 32) their values.  Vector 0 / 1: fp16 h parts of slots 8h.. / 16 + 8h..
          // (the operands of the two K = 16 steps of the main product); vector 2: fp8 of the fp16 residuals * 2^11 of
          // slots 16h .. 16h + 15, vector 3: fp8 of the values of the same slots (K blocks 0 and 1 of the scaled MFMA)
          float nv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            nv[k] = code < p.n ? p.cb[code * 16 + k] : 0.0f;
            const float a = fabsf(nv[k]);
            amax = (a != a) ? __builtin_inff() : __builtin_fmaxf(amax, a);
          }
          typedef _Float16 h8 __attribute__((ext_vector_type(8)));
          h8 v0, v1;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float x = nv[8 * h + k];
            float sq = x * x;
            asm volatile("" : "+v"(sq));          // the fp32 square is the feature (no fused multiply-convert)
            v0[k] = (_Float16)sq;
            v1[k] = (_Float16)x;
          }
          u32x4 v2, v3;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            unsigned lo2 = 0, lo3 = 0, hi2 = 0, hi3 = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              float t[2], r[2];
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                const float x = nv[4 * w + 2 * e + q];
                float tt = h == 0 ? x * x : x;
                asm volatile("" : "+v"(tt));
                t[q] = tt;
                r[q] = (tt - (float)(_Float16)tt) * 2048.0f;
              }
              if (e == 0) { lo2 = fp8x2(r[0], r[1]); lo3 = fp8x2(t[0], t[1]); }
              else { hi2 = fp8x2(r[0], r[1]); hi3 = fp8x2(t[0], t[1]); }
            }
            v2[w] = lo2 | (hi2 << 16);
            v3[w] = lo3 | (hi3 << 16);
          }
          dst[0] = __builtin_bit_cast(u32x4, v0);
          dst[64] = __builtin_bit_cast(u32x4, v1);
          dst[128] = v2;
          dst[192] = v3;
          continue;
        }
#pragma unroll
        for (int m = 0; m < NV; ++m) {
          unsigned hi[8], lo[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int g = PACKED ? k : 16 * m + 8 * h + k;   // slot: [ squares of dims | values of dims ]
            float v = 0.0f;
            if (code < p.n) {
              v = p.cb[code * DIM + (g < DIM ? g : g - DIM)];
              const float a = fabsf(v);
              amax = (a != a) ? __builtin_inff() : __builtin_fmaxf(amax, a);
              if (g < DIM) v = v * v;
            }
            bf16_split(v, hi[k], lo[k]);
          }
          u32x4 vh, vl;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            vh[w] = hi[2 * w] | (hi[2 * w + 1] << 16);
            vl[w] = lo[2 * w] | (lo[2 * w + 1] << 16);
          }
          if constexpr (PACKED) {   // vector 0: (h, l) parts by half; vector 1: (h parts, zero padding)
            const u32x4 zero = {0u, 0u, 0u, 0u};
            dst[0] = h == 0 ? vh : vl;
            dst[64] = h == 0 ? vh : zero;
          } else {
            dst[(long)m * 64] = vh;
            dst[(long)(NV + m) * 64] = vl;
          }
        }
      }
      }   // pass
    } else {
      // no operand image in this call (fp32 filter; grid search, gq_grid.h): max |cb| and the content hash of this block's slice
      const long count = (long)p.n * DIM;
      for (long i = (long)cbk * 256 + tid; i < count; i += (long)kPrepCodeBlocks * 256) {
        const float x = p.cb[i];
        const float a = fabsf(x);
        amax = (a != a) ? __builtin_inff() : __builtin_fmaxf(amax, a);
        hsum += cb_hash_term(x, i);
      }
    }
    __shared__ float s_amax[4], s_r2[4];
    __shared__ unsigned long long s_hsum[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      amax = __builtin_fmaxf(amax, __shfl_xor(amax, o));
      r2max = __builtin_fmaxf(r2max, __shfl_xor(r2max, o));
      hsum += __shfl_xor(hsum, o);
    }
    if ((tid & 63) == 0) { s_amax[tid >> 6] = amax; s_r2[tid >> 6] = r2max; s_hsum[tid >> 6] = hsum; }
    __syncthreads();
    if (tid == 0) {
      p.hdr->absmax_part[cbk] = __builtin_fmaxf(__builtin_fmaxf(s_amax[0], s_amax[1]), __builtin_fmaxf(s_amax[2], s_amax[3]));
      p.hdr->r2_part[cbk] = __builtin_fmaxf(__builtin_fmaxf(s_r2[0], s_r2[1]), __builtin_fmaxf(s_r2[2], s_r2[3]));
      const unsigned long long h = s_hsum[0] + s_hsum[1] + s_hsum[2] + s_hsum[3];
      p.hdr->cbsum[cbk] = h;
      // grid index (gq_grid.h): built from other bytes -> rebuilt, by the one-block kernel that follows, before its use
      if (p.cache_sums && p.cache_stale && p.cache_sums[cbk] != h) *p.cache_stale = 1;
    }
    return;
  }

  // -------------------------------------------------------------------- rows
  __shared__ __attribute__((aligned(16))) float s_mu[256], s_sd[256], s_lsd[256];
  __shared__ __attribute__((aligned(16))) unsigned short s_hi[RB][2 * DIM], s_lo[RB][2 * DIM];
  __shared__ double s_sum[256][F16 ? 9 : 4];     // per-element terms of the row sums (F16: + the five sums of the data-dependent bound)
  __shared__ float s_kl[(FROM_Z && MODE == kModeGQ) ? 256 : 1];   // per-element KL bits (gq_quantize_z_gauss_f32)
  __shared__ float s_scale[RB];                  // MIXED / F16: the row's power-of-two normalisation (NaN: none usable)
  __shared__ __attribute__((aligned(16))) float s_coef[RB][2 * DIM];
  const long row0 = (long)blockIdx.x * RB;
  // element of the tile handled in phase 1: rows fastest for BCHW (consecutive l -> coalesced z reads), else dims fastest
  int lr, g;
  if (FROM_Z && p.omap.mode == 1) { lr = tid % RB; g = tid / RB; } else { lr = tid / DIM; g = tid % DIM; }
  const long row = row0 + lr;
  const bool live = row < p.rows;
  float m = 0.0f, s = 1.0f, ls = 0.0f;
  float klt = 0.0f;
  if (live) {
    if constexpr (FROM_Z) {
      // row = pos * K + k; channel of (k, g): strided g*K + k (GQ1), contiguous k*dim + g (GQ2)
      const OutMap &om = p.omap;
      const long pos = row / om.K;
      const int k = (int)(row % om.K);
      const long ch = om.grouping == 0 ? (long)g * om.K + k : (long)k * DIM + g;
      long zo_mu, zo_lv, oo;
      if (om.mode == 1) {
        const long b = pos / om.L, l = pos % om.L;
        zo_mu = (b * 2 * om.c + ch) * om.L + l;
        zo_lv = (b * 2 * om.c + om.c + ch) * om.L + l;
        oo = (b * om.c + ch) * om.L + l;
      } else {
        zo_mu = pos * 2 * om.c + ch;
        zo_lv = zo_mu + om.c;
        oo = pos * om.c + ch;
      }
      if constexpr (MODE == kModeVQ) {
        // VQ (pit/quantization/vq.py:39-53): z holds c channels, no logvar half -- the row operand is z itself
        m = p.z[oo];
      } else {
      m = p.z[zo_mu];
      float lv = p.z[zo_lv];
      // torch.clamp propagates NaN; min/max with explicit compares keeps that.
      lv = lv < p.lv_min ? p.lv_min : lv;
      lv = lv > p.lv_max ? p.lv_max : lv;
      const float half = 0.5f * lv;
#if defined(GQHIP_ABL) && (GQHIP_ABL & 4096)   // diagnostic build: fp32 exp / log (results differ; only the time is read)
      s = __expf(half);
      ls = __logf(s);
#else
      s = (float)exp((double)half);
      ls = (float)log((double)s);
#endif
      if (p.zhat_noquant) {
#pragma clang fp contract(off)
        const float e = p.noise[oo] * s;
        p.zhat_noquant[oo] = m + e;
      }
      if (p.sd_layout) p.sd_layout[oo] = s;
      if (p.kl2row) {
        // one element of kl2 = 1.4426 * 0.5 * (mu^2 + var - 1 - logvar) in the reference's fp32 op order (gaussian.py:225), var =
        // float(exp(double(logvar)))
        klt = kl_bits_term(m, lv);
      }
      }
    } else {
      m = p.mu[row * DIM + g];
      if constexpr (MODE == kModeGQ) {
        s = p.sd[row * DIM + g];
#if defined(GQHIP_ABL) && (GQHIP_ABL & 4096)
        ls = p.lsd ? p.lsd[row * DIM + g] : __logf(s);
#else
        ls = p.lsd ? p.lsd[row * DIM + g] : (float)log((double)s);
#endif
      }
    }
  }
  float cA, cB;
  double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
  if constexpr (MODE == kModeGQ) {
    const double sg = (double)s;
    double inv = 1.0 / (sg * sg);
    cA = (float)(0.5 * (double)p.beta - 0.5 * inv);
    cB = (float)((double)m * inv);
    if (!(sg > 0.0)) inv = __builtin_nan("");   // sd <= 0 or NaN: the row's bound is NaN -> undecided -> exhaustive semantics
    const double am = fabs((double)m);
    t0 = inv; t1 = am * inv; t2 = am * am * inv; t3 = fabs((double)ls);
  } else {
    cA = -1.0f;
    cB = 2.0f * m;
    t1 = fabs((double)m);
  }
  s_mu[lr * DIM + g] = m;
  s_sd[lr * DIM + g] = s;
  s_lsd[lr * DIM + g] = ls;
  s_coef[lr][g] = cA;
  s_coef[lr][DIM + g] = cB;
  unsigned ah, al, bh, bl;
  bf16_split(cA, ah, al);
  bf16_split(cB, bh, bl);
  s_hi[lr][g] = (unsigned short)ah;
  s_lo[lr][g] = (unsigned short)al;
  s_hi[lr][DIM + g] = (unsigned short)bh;
  s_lo[lr][DIM + g] = (unsigned short)bl;
  s_sum[lr * DIM + g][0] = t0;
  s_sum[lr * DIM + g][1] = t1;
  s_sum[lr * DIM + g][2] = t2;
  s_sum[lr * DIM + g][3] = t3;
  if constexpr (FROM_Z && MODE == kModeGQ) s_kl[lr * DIM + g] = klt;
  if constexpr (F16) {
    // This element's terms of the data-dependent bound (gq_rerank.h:f16_bound), from the fp32 coefficients the filter multiplies.
    // A coordinate is a "well" when A < 0 and the vertex mu' = B / (2 |A|) of its parabola lies within |mu'| <= 6 (any
    // classification is valid; this one keeps M_well small); everything else is charged at its worst case over |n| <= N1.
    // (Round 4: one fp64 division per THREAD here instead of DIM of them in a serial loop of 256 / DIM lanes.)
    const double A = (double)cA, B = (double)cB;
    const double a = fabs(A), b = fabs(B);
    const bool well = A < 0.0 && b <= 12.0 * a;
    s_sum[lr * DIM + g][4] = well ? B * B / (4.0 * a) : 0.0;      // M_well
    s_sum[lr * DIM + g][5] = well ? 0.0 : (A > 0.0 ? A : 0.0);     // P
    s_sum[lr * DIM + g][6] = well ? 0.0 : a;                       // Q
    s_sum[lr * DIM + g][7] = well ? 0.0 : b;                       // Rb
    s_sum[lr * DIM + g][8] = B * B;                                // |B|^2
  }
  __syncthreads();
  if constexpr (MIXED || F16) {
    if (tid < RB) s_scale[tid] = mixed_row_scale(&s_coef[tid][0], 2 * DIM);     // once per row
    __syncthreads();
  }

  // ---- phase 2: everything leaves the block as contiguous runs ----
  const long e_out = row0 * DIM + tid;               // the tile is contiguous in [rows, dim]
  if (e_out < p.rows * DIM) {
    if constexpr (FROM_Z) {
      p.mu[e_out] = s_mu[tid];
      if constexpr (MODE == kModeGQ) {               // (VQ has no sd / log sd rows: 8 bytes per element less to write)
        p.sd[e_out] = s_sd[tid];
        p.lsd[e_out] = s_lsd[tid];
      }
    } else if (MODE == kModeGQ && p.lsd_out) {
      p.lsd_out[e_out] = s_lsd[tid];
    }
  }
  {                                                  // A | B rows: 2 * DIM floats per row, contiguous for the tile
    const float *flat = &s_coef[0][0];
    const long base = row0 * 2 * DIM, lim = p.rows * 2 * DIM;
    if (base + tid < lim) p.coef[base + tid] = flat[tid];
    if (base + 256 + tid < lim) p.coef[base + 256 + tid] = flat[256 + tid];
  }
  if (tid < RB * 4 && row0 + tid / 4 < p.rows) {     // four sums per row, ascending dim order (deterministic)
    const int r = tid / 4, q = tid % 4;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) acc += s_sum[r * DIM + i][q];
    if constexpr (MIXED || F16) {
      const float sc = s_scale[r];
      if (sc != sc) acc = __builtin_nan("");     // no usable normalisation: the row's bound is NaN -> undecided
    }
    p.rowsum[(row0 + r) * 4 + q] = acc;
  }
  if constexpr (FROM_Z && MODE == kModeGQ) {
    if (p.kl2row && tid < RB && row0 + tid < p.rows) {     // the row's KL bits: ascending dim order in fp64, rounded once
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < DIM; ++i) acc += (double)s_kl[tid * DIM + i];
      p.kl2row[row0 + tid] = (float)acc;
    }
  }
  if constexpr (F16) {
    // The sums of the data-dependent bound, ascending dim order (deterministic), each rounded UP to fp32: thread (row, q) adds the
    // per-element terms phase 1 left in LDS; q = 5, 6: the two maxima max|A|, max(|A|, |B|).
    for (int w = tid; w < RB * 8; w += 256) {
      const int r = w >> 3, q = w & 7;
      if (row0 + r >= p.rows) continue;
      double acc = 0.0;
      if (q < 5) {
#pragma unroll
        for (int i = 0; i < DIM; ++i) acc += s_sum[r * DIM + i][4 + q];
      } else if (q < 7) {
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < DIM; ++i) {
          m = __builtin_fmaxf(m, fabsf(s_coef[r][i]));
          if (q == 6) m = __builtin_fmaxf(m, fabsf(s_coef[r][DIM + i]));
        }
        acc = (double)m;
      }
      p.rowaux[(row0 + r) * 8 + q] = q < 7 ? f32_up(acc) : 0.0f;
    }
    if (p.rowimg && tid < RB * NVEC * 2) {
      // row image: vector m, half h = fp16 of the normalised [A | B] slots 16 m + 8 h .. + 7 (DIM 4: half 0 holds all 8, half 1 zeros)
      const int r = tid / (NVEC * 2), v = (tid / 2) % NVEC, h = tid % 2;
      if (row0 + r < p.rows) {
        float sc = s_scale[r];
        if (v == 0 && h == 0) p.rowscale[row0 + r] = sc != sc ? sc : 1.0f / sc;     // 2^e_r (exact)
        if (sc != sc) sc = 0.0f;
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          o[k] = (PACKED && h == 1) ? (_Float16)0.0f : (_Float16)(s_coef[r][(PACKED ? 0 : 16 * v + 8 * h) + k] * sc);
        p.rowimg[(row0 + r) * (NVEC * 2) + v * 2 + h] = __builtin_bit_cast(u32x4, o);
      }
    }
    return;
  }
  if constexpr (MIXED) {
    if (p.rowimg && tid < RB * 8) {
      // row image: vector 0 / 1: fp16 h parts of the normalised A (dims 8h..) / B (dims 8h..); vector 2: fp8 of those h
      // parts * 2^-6 for slots 16h .. 16h + 15 (h = 0: A of all dims, h = 1: B), vector 3: fp8 of the fp16 residuals * 2^6
      const int r = tid / 8, v = (tid / 2) % 4, h = tid % 2;
      if (row0 + r < p.rows) {
        float sc = s_scale[r];
        if (v == 0 && h == 0) p.rowscale[row0 + r] = sc != sc ? sc : 1.0f / sc;     // 2^e_r (exact)
        if (sc != sc) sc = 0.0f;
        u32x4 out;
        if (v < 2) {
          typedef _Float16 h8 __attribute__((ext_vector_type(8)));
          h8 o;
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = (_Float16)(s_coef[r][16 * v + 8 * h + k] * sc);
          out = __builtin_bit_cast(u32x4, o);
        } else {
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            unsigned lo = 0, hi = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              float t[2];
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                const float x = s_coef[r][16 * h + 4 * w + 2 * e + q] * sc;
                const float xh = (float)(_Float16)x;
                t[q] = v == 2 ? xh * 0.015625f : (x - xh) * 64.0f;
              }
              if (e == 0) lo = fp8x2(t[0], t[1]); else hi = fp8x2(t[0], t[1]);
            }
            out[w] = lo | (hi << 16);
          }
        }
        p.rowimg[(row0 + r) * 8 + v * 2 + h] = out;
      }
    }
    return;
  }
  if (p.rowimg && tid < RB * NVEC * 2) {
    const int r = tid / (NVEC * 2), v = (tid / 2) % NVEC, h = tid % 2;
    if (row0 + r < p.rows) {
      u32x4 out;
      if constexpr (PACKED) {   // vector 0: h parts in both halves; vector 1: (l parts, zero padding)
        const u32x4 zero = {0u, 0u, 0u, 0u};
        out = v == 0 ? *reinterpret_cast<const u32x4 *>(&s_hi[r][0])
                     : (h == 0 ? *reinterpret_cast<const u32x4 *>(&s_lo[r][0]) : zero);
      } else {                  // vectors 0..NV-1: h parts of slots 16m+8h..+7; NV..2NV-1: l parts
        const int mm = v % NV;
        out = v < NV ? *reinterpret_cast<const u32x4 *>(&s_hi[r][16 * mm + 8 * h])
                     : *reinterpret_cast<const u32x4 *>(&s_lo[r][16 * mm + 8 * h]);
      }
      p.rowimg[(row0 + r) * (NVEC * 2) + v * 2 + h] = out;
    }
  }
}

// Legacy operand prep for shapes the MFMA filters do not cover (dim not in {4, 8, 16, 32}: exhaustive kernel):
// z [mu | logvar] -> mu, sd, lsd rows and zhat_noquant, one thread per element.
struct PrepPlainParams {
  const float *z, *noise;
  float *zhat_noquant;
  float *sd_layout, *kl2row;   // optional (see PrepParams)
  float *mu, *sd, *lsd;   // [rows, dim]
  long rows;
  int dim;
  int vq;                 // z holds c channels and no logvar half (pit/quantization/vq.py:39-53)
  WsHeader *hdr;          // already zeroed on the stream: thread 0 leaves the straight-through words
  int ste_kind;
  const float *ste;
  float *pure;
  float lv_min, lv_max;
  OutMap omap;
};


__global__ __launch_bounds__(256) void prep_plain_kernel(const PrepPlainParams p) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t == 0) { p.hdr->ste_kind = p.ste_kind; p.hdr->ste = p.ste; p.hdr->pure = p.pure; }
  if (t >= p.rows * p.dim) return;
  const OutMap &om = p.omap;
  // BCHW reads are coalesced along l when consecutive threads walk l: threads are (b, ch, l) for BCHW, (pos, ch) for BLC
  long b, l, pos;
  int ch;
  if (om.mode == 1) {
    l = t % om.L;
    ch = (int)((t / om.L) % om.c);
    b = t / ((long)om.L * om.c);
    pos = b * om.L + l;
  } else {
    ch = (int)(t % om.c);
    pos = t / om.c;
    b = pos / om.L;
    l = pos % om.L;
  }
  int g, k;
  if (om.grouping == 0) { g = ch / om.K; k = ch % om.K; } else { k = ch / p.dim; g = ch % p.dim; }
  const long row = pos * om.K + k;
  long zo, oo;
  if (om.mode == 1) { zo = (b * 2 * om.c + ch) * om.L + l; oo = (b * om.c + ch) * om.L + l; }
  else { zo = pos * 2 * om.c + ch; oo = pos * om.c + ch; }
  const long o = row * p.dim + g;
  if (p.vq) {
    p.mu[o] = p.z[oo];
    return;
  }
  const long lv_off = om.mode == 1 ? (long)om.c * om.L : (long)om.c;
  const float m = p.z[zo];
  float lv = p.z[zo + lv_off];
  lv = lv < p.lv_min ? p.lv_min : lv;
  lv = lv > p.lv_max ? p.lv_max : lv;
  const float half = 0.5f * lv;
  const float s = (float)exp((double)half);
  p.mu[o] = m;
  p.sd[o] = s;
  p.lsd[o] = (float)log((double)s);
  if (p.zhat_noquant) {
#pragma clang fp contract(off)
    const float e = p.noise[oo] * s;
    p.zhat_noquant[oo] = m + e;
  }
  if (p.sd_layout) p.sd_layout[oo] = s;
  if (p.kl2row && g == 0) {      // this (cold) path: the row's first element walks the row (ascending dim order, fp64 sum, rounded once)
    double acc = 0.0;
    for (int i = 0; i < p.dim; ++i) {
      const long chi = om.grouping == 0 ? (long)i * om.K + k : (long)k * p.dim + i;
      const long zi = om.mode == 1 ? (b * 2 * om.c + chi) * om.L + l : pos * 2 * om.c + chi;
      float lvi = p.z[zi + lv_off];
      lvi = lvi < p.lv_min ? p.lv_min : lvi;
      lvi = lvi > p.lv_max ? p.lv_max : lvi;
      acc += (double)kl_bits_term(p.z[zi], lvi);
    }
    p.kl2row[row] = (float)acc;
  }
}

}  // namespace gqhip
