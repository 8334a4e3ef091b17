// gq_common.h -- shared device helpers for libgqhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gqhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kTileCodes = 32;     // codes per MFMA 32x32 tile
constexpr int kMaxSplit = 64;      // code splits (one lane per split in the re-rank)
constexpr int kMaxDim = 64;

// float32(log(sqrt(2*pi))): torch casts the python scalar to the tensor dtype.
__device__ __host__ constexpr float half_log_2pi() { return 0.91893853320467274178f; }

// One candidate record per (row, record set), produced by the filter kernel: 16 bytes (round 4; 32 before -- the records are the
// largest stream between the filter and the re-rank: 2 x 8.4 MB less HBM traffic per call at config 2).
//   m1 >= m2 >= m3 >= m4 : the four largest half-group maxima of the filter score.  m1 is stored as it is; m2..m4 as the GAPS
//                          m1 - mk in fp16, rounded TOWARDS ZERO after the fp32 difference was scaled by (1 - 2^-22) (it may have
//                          been rounded up): the value the re-rank reconstructs, m1 - gap, is never BELOW the filter's mk, so a
//                          group is never missed -- at worst one more is looked at.  Gaps below 2^-24 read as 0, above 65504 as
//                          65504, "no group" (mk = -inf) as +inf.
//   id1, id2, id3        : half-group ids of m1, m2, m3 RELATIVE to the record set's split: ((tile - t_begin) / GT) * 2 + half; a
//                          half-group is the GT x 16 codes one lane half sees in GT consecutive tiles.  16 bits: the plan keeps
//                          2 * tiles_per_split / GT <= 65536 (gqhip.hip:make_plan).
// The fourth value only tells the re-rank whether a fourth group could matter (then the row is undecided).
struct __attribute__((aligned(16))) Rec {
  float m1;
  unsigned short d2, d3, d4;
  unsigned short id1, id2, id3;
};
static_assert(sizeof(Rec) == 16, "one 16-byte store / load per record");

__device__ __forceinline__ unsigned short rec_gap(float m1, float mk) {
  if (!(mk > -__builtin_inff())) return (unsigned short)0x7c00u;          // no group (or NaN): +inf
  const float g = (m1 - mk) * 0.99999976158142090f;                       // (1 - 2^-22)
  return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(g, 0.0f)) & 0xffffu);
}
// base2 = 2 * (t_begin / GT): the first half-group id of the record set's split
__device__ __forceinline__ Rec make_rec(float m1, float m2, float m3, float m4, int j1, int j2, int j3, int base2) {
  const float NEG_INF = -__builtin_inff();
  Rec r;
  r.m1 = m1;
  r.d2 = rec_gap(m1, m2); r.d3 = rec_gap(m1, m3); r.d4 = rec_gap(m1, m4);
  r.id1 = (unsigned short)(m1 > NEG_INF ? j1 - base2 : 0);
  r.id2 = (unsigned short)(m2 > NEG_INF ? j2 - base2 : 0);
  r.id3 = (unsigned short)(m3 > NEG_INF ? j3 - base2 : 0);
  return r;
}
__device__ __forceinline__ Rec empty_rec() {
  Rec r;
  r.m1 = -__builtin_inff();
  r.d2 = r.d3 = r.d4 = (unsigned short)0x7c00u;
  r.id1 = r.id2 = r.id3 = 0;
  return r;
}
// the reconstructed k-th value (k = 2, 3, 4): exact in fp64 whenever the gap is not below an fp32 ulp of m1 (then it reads m1)
__device__ __forceinline__ double rec_value(float m1, unsigned short gap) {
  return (double)m1 - (double)(float)__builtin_bit_cast(_Float16, gap);
}

// gq_quantize_z_gauss_f32's statistics block (gq_gauss.h); a copy travels in the workspace header when that block runs inside the
// re-rank launch.
struct GaussStatsParams {
  const float *kl2row;      // [rows] KL bits per row (gq_prep.h)
  double *lam_state;        // [3] in / out: lam, lam_min, lam_max
  void *scalars;            // 64 B out
  long rows;                // 0: no statistics in this call
  double lam_factor, lam_lo, lam_hi;
  float thr_hi, thr_lo, log2n;      // float(n + tol), float(n - tol), float(n): torch compares the fp32 tensor with the scalar cast to fp32
  int lam_max_decreases;    // 1: gaussian.py:109-112 (GQ1); 0: GQ2, whose decrease is a no-op expression (gaussian.py:251)
};
static_assert(sizeof(GaussStatsParams) == 72, "fits the header's pad");

// Workspace header (first 8 KiB of the caller's workspace).  Everything in it is (re)written by the kernels of ONE
// call: gq_prep_kernel resets the counters and writes the max|cb| partials, the re-rank reduces them per wave.
// Nothing here is read across calls.
constexpr int kAbsmaxParts = 256;
struct WsHeader {
  int fb_count;                       // rows the candidates could not decide (finished by the in-block scan; exhaustive kernel: its list)
  int pad0;
  unsigned long long reranked;        // half-pairs (32 codes each) evaluated exactly (grid search: codes given the reference's arithmetic)
  unsigned long long grid_leaves;     // grid search (gq_grid.h), debug statistics: leaves visited, summed over the rows
  int grid_next;                      // grid search: the next group of four rows a wave fetches
  int loss_ticket;                    // vq_loss_kernel (gq_aux.h): blocks that have left their partial sum (reset by the first launch)
  // straight-through mix where zhat is stored (gq_rerank.h:ste_mix), written by the call's first launch:
  int ste_kind;                       // 0: zhat = code;  1: (g - g) + code, g = ste[o] = zhat_noquant (GQ2, pit/quantization/gaussian.py:337-338);
                                      // 2: g + (code - g), g = ste[o] = z (VQ, pit/quantization/vq.py:89)
  int pad1a;
  const float *ste;                   // [the layout of zhat] or NULL
  float *pure;                        // optional second output in the layout of zhat: the codeword itself (GQ2's info["zhat_quant"])
  int pad1[18];
  float absmax_part[kAbsmaxParts];    // one partial per code block of gq_prep_kernel
  unsigned long long stamps[48];      // diagnostic builds only (GQHIP_CLOCK_STAMPS)
  GaussStatsParams gs;                // gs.rows > 0: the re-rank launch carries one extra block that runs gauss_stats_block (written by the first launch)
  int pad2[110];
  float r2_part[kAbsmaxParts];        // max squared code norm per code block (fp16 filter's norm bound, gq_rerank.h)
  int pad3[256];
  unsigned long long cbsum[kAbsmaxParts];   // content hash of the codebook slice each code block of gq_prep_kernel read in THIS call
                                            // (what the codebook cache is validated against and stamped with: gq_grid.h, gq_prep.h)
  double loss_part[256];              // vq_loss_kernel: one partial sum of (e - z)^2 per block
};
static_assert(sizeof(WsHeader) == 8192, "header is 8 KiB");

// Position-salted content hash of a run of fp32 words (codebook cache validation): sum of bits(x_i) + c times an odd multiplier
// derived from i, modulo 2^64 -- any single changed word and any swap of two different words changes the sum.
__device__ __forceinline__ unsigned long long cb_hash_term(float x, long i) {
  const unsigned long long b = (unsigned long long)__float_as_uint(x) + 0x9e3779b97f4a7c15ull;
  return b * (((unsigned long long)i * 0xd1342543de82ef95ull) | 1ull);
}

// Insert (t, id) into a descending top-4 (ids kept for the top 3 only).
__device__ __forceinline__ void top4_insert(float t, int id, float &m1, float &m2, float &m3, float &m4,
                                            int &i1, int &i2, int &i3) {
  const bool g1 = t > m1;
  const bool g2 = t > m2;
  const bool g3 = t > m3;
  m4 = __builtin_amdgcn_fmed3f(m3, m4, t);
  m3 = __builtin_amdgcn_fmed3f(m2, m3, t);
  m2 = __builtin_amdgcn_fmed3f(m1, m2, t);
  m1 = __builtin_fmaxf(m1, t);
  // plain selects (g1 implies g2 implies g3) -> v_cndmask, no control flow
  i3 = g3 ? id : i3;
  i3 = g2 ? i2 : i3;
  i2 = g2 ? id : i2;
  i2 = g1 ? i1 : i2;
  i1 = g1 ? id : i1;
}
// Top-3 variant (ids of the top 2): the third value is kept in `m4`'s role by the caller (see the packed dim-4
// filter, where the tracker's VALU work is on the critical path).
__device__ __forceinline__ void top3_insert(float t, int id, float &m1, float &m2, float &m3, int &i1, int &i2) {
  const bool g1 = t > m1;
  const bool g2 = t > m2;
  m3 = __builtin_amdgcn_fmed3f(m2, m3, t);
  m2 = __builtin_amdgcn_fmed3f(m1, m2, t);
  m1 = __builtin_fmaxf(m1, t);
  i2 = g2 ? id : i2;
  i2 = g1 ? i1 : i2;
  i1 = g1 ? id : i1;
}
__device__ __forceinline__ void top4_insert_value(float t, float m3, float &m4) {
  m4 = __builtin_amdgcn_fmed3f(m3, m4, t);
}

// torch.argmax order on (score, index): NaN is the maximum and the first NaN
// wins; otherwise the larger score, then the smaller index.
__device__ __forceinline__ bool ref_better(float sa, int ia, float sb, int ib) {
  const bool na = sa != sa, nb = sb != sb;
  if (na || nb) return na && (!nb || ia < ib);
  return sa > sb || (sa == sb && ia < ib);
}

// ---- the reference's fp32 arithmetic, op by op (no contraction) -----------
// pit/quantization/gaussian.py:51-52  Normal(0,1).log_prob(n)
__device__ __forceinline__ float ref_nlp(float n) {
#pragma clang fp contract(off)
  float d = n - 0.0f;
  float q = d * d;
  float t = (-q) / 2.0f;
  t = t - 0.0f;
  return t - half_log_2pi();
}

// pit/quantization/gaussian.py:142-146 one element of log_ratios
__device__ __forceinline__ float ref_term(float n, float mu, float var2, float lsd,
                                          float beta) {
#pragma clang fp contract(off)
  float d = n - mu;
  float q = d * d;
  float t = __fdiv_rn(-q, var2);
  t = t - lsd;
  t = t - half_log_2pi();
  float u = ref_nlp(n) * beta;
  return t - u;
}

// torch.sum(dim=2) order (8 strided accumulators, left-to-right combine) --
// see oracle/gq_oracle.c:gq_row_score.  All operands are in global memory:
// code row `n`, row operands mu / sd / lsd (lsd may be NULL -> fp64 log).
__device__ __forceinline__ float ref_lsd(const float *lsd, const float *sd, int i) {
  return lsd ? lsd[i] : (float)log((double)sd[i]);
}

__device__ inline float ref_score(const float *__restrict__ n, const float *__restrict__ mu,
                                  const float *__restrict__ sd, const float *__restrict__ lsd,
                                  int dim, float beta) {
#pragma clang fp contract(off)
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < dim) {
      float s = sd[k];
      float var2 = 2.0f * (s * s);
      acc[k] = ref_term(n[k], mu[k], var2, ref_lsd(lsd, sd, k), beta);
    } else {
      acc[k] = 0.0f;
    }
  }
  for (int i0 = 8; i0 < dim; i0 += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k;
      if (i < dim) {
        float s = sd[i];
        float var2 = 2.0f * (s * s);
        acc[k] = acc[k] + ref_term(n[i], mu[i], var2, ref_lsd(lsd, sd, i), beta);
      }
    }
  }
  float s = acc[0];
#pragma unroll
  for (int k = 1; k < 8; ++k)
    if (k < dim) s = s + acc[k];
  return s;
}

// VQ arbiter: -(|z|^2 + |e|^2 - 2 z.e) in fp64 (argmax of the negated distance).
__device__ inline double vq_neg_dist(const float *__restrict__ e, const float *__restrict__ z, int dim) {
  double zz = 0.0, ee = 0.0, ze = 0.0;
  for (int i = 0; i < dim; ++i) {
    const double a = z[i], b = e[i];
    zz += a * a;
    ee += b * b;
    ze += a * b;
  }
  double r = -(zz + ee - 2.0 * ze);
  // Opaque to the optimiser: with the negation visible, hipcc (ROCm 7.2, -O3) folds it into the fp64 compares of
  // the callers' (score, index) selection as source modifiers and the multi-candidate re-rank then keeps the lower
  // index instead of the larger score (reproduced on gfx950; tests/test_gpu_modules.py::test_vq_multi_candidate).
  asm volatile("" : "+v"(r));
  return r;
}

}  // namespace gqhip
