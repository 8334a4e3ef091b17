// gq_rerank.h -- exact re-rank of the filter's candidates, and the exhaustive
// fallback.  Both evaluate the reference's score in the reference's operation
// order (gq_common.h:ref_score), so the winning index is the one
// torch.argmax returns on the reference's CPU path
// (pit/quantization/gaussian.py:142-150).
//
// Why a small candidate set is enough (DESIGN.md "exactness argument"): for
// every code j,  |filter(r,j) + const(r) - ref_score(r,j)| <= E(r), where E is
// the rigorous rounding bound computed below from (mu, sd, max|cb|, dim).  The
// reference's arg-max j* therefore satisfies filter(r,j*) >= max_j filter(r,j)
// - 2E, i.e. it lies in a half-group whose maximum is within `margin` = 2.5 E of
// the row maximum.  The filter keeps, per (row, split), the best three such
// half-groups by id and the fourth by value; if the fourth is also within the
// margin (or the row has non-finite operands / bound) the row is undecided and
// goes to the next stage (fp32 filter level of the cascade, then the fp64 second
// stage) instead.  Either way no approximation reaches the output.
#pragma once
#include "gq_common.h"
#include "gq_filter.h"

namespace gqhip {

// Where results go: plain rows, or straight into the module's output layout
// (folds the inverse permutes of gaussian.py:153-158 / :318-327).
struct OutMap {
  int mode;      // 0: idx[row], zhat[row*dim + g];  1: BCHW;  2: BLC
  int K, L, c;   // sub-codebooks per position, positions per image, channels
  int grouping;  // 0 strided (GQ1), 1 contiguous (GQ2)
};

__device__ __forceinline__ long out_idx_offset(const OutMap &m, long row) {
  if (m.mode == 1) {
    const long pos = row / m.K, k = row % m.K;
    const long b = pos / m.L, l = pos % m.L;
    return (b * m.K + k) * m.L + l;
  }
  return row;  // plain rows and BLC ([B, L, K]) coincide
}
__device__ __forceinline__ long out_zhat_offset(const OutMap &m, long row, int g, int dim) {
  if (m.mode == 0) return row * dim + g;
  const long pos = row / m.K, k = row % m.K;
  const long ch = m.grouping == 0 ? (long)g * m.K + k : k * dim + g;
  if (m.mode == 1) {
    const long b = pos / m.L, l = pos % m.L;
    return (b * m.c + ch) * m.L + l;
  }
  return pos * m.c + ch;
}

// Scratch of the "spread" second stage (few listed rows, each spread over kSpreadSlices blocks).
constexpr int kCascadeMin = 64;     // list A longer than this goes through the fp32 second-level filter first
constexpr int kSpreadRows = 64;     // listed rows handled by the spread kernels (more: gq_fallback64_kernel)
constexpr int kSpreadSlices = 32;   // blocks per listed row
struct SpreadPartial {
  double s;
  int i;      // 0x7fffffff: empty
  int pad;
};
struct SpreadSlot {
  unsigned long long pad0;
  int done;                      // slices finished
  int pad;
  SpreadPartial part[kSpreadSlices];
};

struct RerankParams {
  const float *mu;    // [rows, dim] (VQ: z)
  const float *sd;    // [rows, dim]
  const float *lsd;   // [rows, dim] (NULL only on the exhaustive path: fp64 log of sd)
  const double *rowsum;  // [rows, 4] sums of gq_prep_kernel (bound of the re-rank)
  const float *coef;     // [rows, 2, dim] fp32 filter coefficients A | B of gq_prep_kernel (pre-filter of the re-rank)
  const float *cb;    // [n, dim]
  const Rec *rec;     // [nsplit, rows]
  int64_t *idx;
  float *zhat;        // may be NULL
  WsHeader *hdr;
  int *fb_list;       // [rows]  list A: rows the first filter + re-rank could not decide
  int *fb2_list;      // [rows]  list B: rows still undecided after the fp32 second level (cascade only)
  int cascade;        // 1: a second-level fp32 filter runs when list A has more than kCascadeMin rows
  int level;          // re-rank: 1 = all rows -> list A, 2 = list A -> list B (cascade only)
  SpreadSlot *spread; // [kSpreadRows]
  int rows, n, dim;
  float beta;
  int nsplit;
  int gt;             // tiles per candidate group -- must match the filter's GT
  float ef_coeff;     // filter error bound E_f = ef_coeff * 2^-24 * T  (fp32 filter: 2 dim + 4; split-bf16: 220 + 24 dim;
                      // fp16 + fp8: 2450)
  float n1_limit;     // > 0: the filter's operand formats assume n1_min <= max|cb| <= n1_limit (fp16 + fp8 images: 1 .. 16; below
                      // 1 the absolute errors of fp8-subnormal operands are not covered by the bound; fp16 images: 0 .. 255, the
                      // squares must stay below 65504); any other codebook makes every row undecided (cascade: fp32 filter,
                      // fp64 second stage)
  float n1_min;
  const float *rowaux;   // [rows, 8] sums of the data-dependent bound (gq_prep_kernel, F16) or NULL: the classic bound k u T
  int all_rows;       // exhaustive kernel: process every row (no filter ran)
  int stats;          // count re-ranked half-pairs (debug)
  int bar_spin_limit; // tail kernel: polls (~0.25 us each) a grid barrier waits before it gives up (gq_tail.h)
  OutMap omap;
};

// the list the fp64 second stage works on: list B when the cascade's second level ran, else list A
__device__ __forceinline__ void second_stage_list(const RerankParams &p, const int *&list, int &count) {
  list = p.fb_list;
  count = p.hdr->fb_count;
  if (p.cascade && count > kCascadeMin) {
    list = p.fb2_list;
    count = p.hdr->fb2_count;
  }
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Row operands shared by all lanes of the wave/block that scores one row:
// mu, var2 = 2*(sd*sd), lsd = log sd  (GQ)  |  z (VQ).
struct RowOps {
  float mu[kMaxDim], var2[kMaxDim], lsd[kMaxDim];
};

__device__ __forceinline__ void load_row_ops(const RerankParams &p, long row, int i, RowOps &ro) {
#pragma clang fp contract(off)
  const float m = p.mu[row * p.dim + i];
  ro.mu[i] = m;
  if (p.sd) {
    const float s = p.sd[row * p.dim + i];
    ro.var2[i] = 2.0f * (s * s);
    ro.lsd[i] = p.lsd ? p.lsd[row * p.dim + i] : (float)log((double)s);
  }
}

// torch.sum(dim=2) order: 8 strided accumulators, left-to-right combine
// (oracle/gq_oracle.c:gq_row_score).
__device__ inline float ref_score_ops(const float *__restrict__ n, const RowOps &ro, int dim, float beta) {
#pragma clang fp contract(off)
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    acc[k] = k < dim ? ref_term(n[k], ro.mu[k], ro.var2[k], ro.lsd[k], beta) : 0.0f;
  for (int i0 = 8; i0 < dim; i0 += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k;
      if (i < dim) acc[k] = acc[k] + ref_term(n[i], ro.mu[i], ro.var2[i], ro.lsd[i], beta);
    }
  }
  float s = acc[0];
#pragma unroll
  for (int k = 1; k < 8; ++k)
    if (k < dim) s = s + acc[k];
  return s;
}

template <int MODE>
__device__ __forceinline__ double exact_score(const RerankParams &p, const RowOps &ro, int code) {
  const float *n = p.cb + (long)code * p.dim;
  if constexpr (MODE == kModeGQ) {
    return (double)ref_score_ops(n, ro, p.dim, p.beta);
  } else {
    return vq_neg_dist(n, ro.mu, p.dim);
  }
}

// Out-of-line copy for the fp64 second stage: keeps the rarely taken exact evaluation out of the
// register budget of its hot fp64 loops.
template <int MODE>
__device__ __attribute__((noinline)) double exact_score_cold(const float *cb, const RowOps *ro, int code, int dim,
                                                             float beta) {
  const float *n = cb + (long)code * dim;
  if constexpr (MODE == kModeGQ) {
    return (double)ref_score_ops(n, *ro, dim, beta);
  } else {
    return vq_neg_dist(n, ro->mu, dim);
  }
}

// comparator on (double score, index) with torch.argmax semantics
__device__ __forceinline__ bool better_d(double sa, int ia, double sb, int ib) {
  const bool na = sa != sa, nb = sb != sb;
  if (na || nb) return na && (!nb || ia < ib);
  return sa > sb || (sa == sb && ia < ib);
}

__device__ __forceinline__ void write_result(const RerankParams &p, long row, int best, int lane) {
  if (lane == 0) p.idx[out_idx_offset(p.omap, row)] = (int64_t)best;
  if (p.zhat && lane < p.dim)
    p.zhat[out_zhat_offset(p.omap, row, lane, p.dim)] = p.cb[(long)best * p.dim + lane];
}

// max |cb| from the per-block partials gq_prep_kernel left in the header (one 1-KiB coalesced load per wave).
__device__ __forceinline__ float wave_absmax(const float *parts, int lane) {
  const f32x4 v = reinterpret_cast<const f32x4 *>(parts)[lane];
  float m = __builtin_fmaxf(__builtin_fmaxf(v.x, v.y), __builtin_fmaxf(v.z, v.w));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o));
  return m;
}
static_assert(kAbsmaxParts == 256, "wave_absmax reads 4 partials per lane");

// The rounding bound of one row from the four sums gq_prep_kernel left (S0 = sum 1/sd^2, S1 = sum |mu|/sd^2,
// S2 = sum mu^2/sd^2, S3 = sum |log sd|; VQ: S1 = sum |z|) and N1 = max|cb|:
//   T = sum_i (|beta|/2 + 1/(2 sd^2)) N1^2 + |mu| N1 / sd^2,
//   G = sum_i (N1 + |mu|)^2 / (2 sd^2) + |log sd| + c + |beta| (N1^2 / 2 + c)         (DESIGN.md section 3).
template <int MODE>
__device__ __forceinline__ void row_bound(const double *rs, double N1, int dim, float beta, double &T, double &G) {
  const double N2 = N1 * N1;
  if constexpr (MODE == kModeGQ) {
    const double b = fabs((double)beta), c = (double)half_log_2pi();
    T = (0.5 * b * dim + 0.5 * rs[0]) * N2 + rs[1] * N1;
    G = 0.5 * (N2 * rs[0] + 2.0 * N1 * rs[1] + rs[2]) + rs[3] + dim * (c + b * (0.5 * N2 + c));
  } else {
    T = dim * N2 + 2.0 * rs[1] * N1;
    G = 0.0;
  }
}

// The error bound of the fp16 main-product filter (gq_filter_bf16.h, F16), level 1 only.  u = 2^-24, k = ef_coeff.
// For every code j:  |f~(j) - f(j)| <= E(j) = k u T_j + E_abs,   T_j = sum_i |A_i| n_ji^2 + |B_i| |n_ji|,
// (two fp16 roundings per product, 2^-10 + 2^-22 relative; fp32 accumulation; E_abs: operands in fp16's subnormal range).
// Three bounds on T_j, all rigorous, the smallest wins:
//   * worst case over |n| <= N1:                       T_j <= T_old                                    (row_bound)
//   * Cauchy-Schwarz with R2 = max_j |n_j|^2:          T_j <= max|A| R2 + |B|_2 sqrt(R2) = T_norm
//   * through the code's own score.  Coordinates are classed by gq_prep_kernel: a "well" (A < 0, vertex mu' = B / 2|A|
//     with |mu'| <= 6) contributes  f_i = -a (n - mu')^2 + a mu'^2  and  |A| n^2 + |B||n| <= 3 a (n - mu')^2 + 5 a mu'^2
//     (d = |n - mu'|, m = |mu'|: a (d + m)^2 + 2 a m (d + m) = a d^2 + 4 a d m + 3 a m^2 <= 3 a d^2 + 5 a m^2); any other
//     coordinate contributes at most U_i = max(A, 0) N1^2 + |B| N1 to f and |A| N1^2 + |B| N1 to T.  Summing:
//     sum_well a d^2 <= M_well + U_wc - f(j), hence   T_j <= 8 M_well + 3 U_wc + T_wc - 3 f(j) = Cr - 3 f(j):
//     the better a code scores, the smaller its error.
// With j^ = arg max f~, F = f~(j^) and the reference's arg-max j* (f(j*) >= f(j^) - 2 E_r):
//     E(j^) <= Ea = min(E_unif, (k u (Cr - 3 F) + E_abs) / (1 - 3 k u)),        E_unif = k u min(T_old, T_norm) + E_abs,
//     E(j*) <= Eb = min(E_unif, k u (Cr - 3 F + 3 Ea + 6 E_r) + E_abs),
//     f~(j*) >= F - (Ea + Eb + 2 E_r).
// Returns Ea + Eb (the caller adds 2 E_r and its safety factor).  aux = (M_well, P, Q, Rb, |B|^2, max|A|, max(|A|,|B|), 0).
__device__ __forceinline__ double f16_bound(const float (&aux)[8], double T_old, double Er, double N1, double R2, double F,
                                            int dim, double ku) {
  const double N2 = N1 * N1;
  const double U_wc = (double)aux[1] * N2 + (double)aux[3] * N1;
  const double T_wc = (double)aux[2] * N2 + (double)aux[3] * N1;
  const double Cr = 8.0 * (double)aux[0] + 3.0 * U_wc + T_wc;
  const double T_norm = (double)aux[5] * R2 + sqrt((double)aux[4] * R2);
  const double NN = N2 > N1 ? N2 : N1;
  // subnormal operands (normalised units: the row's largest coefficient is in [2^13, 2^14), so 2^e_r <= cmax 2^-13):
  // a coefficient below 2^-14 is off by <= 2^-25 absolute, times |s| <= max(N1^2, N1); an s below 2^-14 is off by <= 2^-25,
  // times a coefficient < 2^14
  const double E_abs = 2.0 * dim * (2.98023223876953125e-08 * NN + 4.8828125e-04) * (double)aux[6] * 1.220703125e-04;
  const double Tu = T_old < T_norm ? T_old : T_norm;
  const double E_unif = ku * Tu + E_abs;
  double slack = Cr - 3.0 * F;
  slack = slack > 0.0 ? slack : 0.0;
  double Ea = (ku * slack + E_abs) / (1.0 - 3.0 * ku);
  Ea = Ea < E_unif ? Ea : E_unif;
  double Eb = ku * (slack + 3.0 * Ea + 6.0 * Er) + E_abs;
  Eb = Eb < E_unif ? Eb : E_unif;
  return Ea + Eb;       // NaN in, NaN out: the caller's `margin < 1e30` test sends the row to the next stage
}

// The reference score of a code row held in registers against row operands [mu | 2 sd^2 | log sd] in LDS (same
// operation order as ref_score_ops: 8 strided accumulators, left-to-right combine).
template <int DIM>
__device__ __forceinline__ float ref_score_lds(const float (&n)[DIM], const float *ops, float beta) {
#pragma clang fp contract(off)
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = k < DIM ? ref_term(n[k], ops[k], ops[DIM + k], ops[2 * DIM + k], beta) : 0.0f;
#pragma unroll
  for (int i0 = 8; i0 < DIM; i0 += 8)
#pragma unroll
    for (int k = 0; k < 8; ++k)
      acc[k] = acc[k] + ref_term(n[i0 + k], ops[i0 + k], ops[DIM + i0 + k], ops[2 * DIM + i0 + k], beta);
  float s = acc[0];
#pragma unroll
  for (int k = 1; k < 8; ++k)
    if (k < DIM) s = s + acc[k];
  return s;
}

// The exact re-rank.  16 lanes per row (4 rows per wave, 16 per block); a candidate = one "half-group" of the filter
// = the 16 codes of one lane half in each of `gt` consecutive tiles, so every lane owns `gt` codes per candidate.
// The kernel is a chain of dependent memory round trips (records -> code rows -> result), so everything a row needs
// besides the records is in flight at once: pass 1's operands A | B go to REGISTERS (16-byte broadcast loads), pass 2's
// (mu, 2 sd^2, log sd) to a padded LDS record, the bound comes from the four sums of gq_prep_kernel (no fp64 division
// here), and at level 1 the results of a block's 16 consecutive rows leave through LDS as contiguous
// runs in the module layout.
//
// Two passes over the candidates' codes.  Pass 1 evaluates the filter expansion f^(j) = sum_i A_i n_ji^2 + B_i n_ji as a
// plain fp32 FMA chain (2 dim FMAs, no division) and takes the group-wide maximum F.  Its error is that of the fp32 MFMA
// filter, |f^ - f| <= E32 = (2 dim + 4) u T, so by the same argument as for the filter the reference's arg-max j* --
// which IS among the candidates -- satisfies f^(j*) >= F - 2 (E32 + E_r).  Pass 2 therefore evaluates the reference's
// own score (ref_term: one IEEE division per dimension) only for the codes with f^ >= F - 2.5 (E32 + E_r): one or two
// per row instead of all 16 gt.  That makes coarse candidates (gt = 4: 64 codes) cheap here, and coarse candidates are
// what keeps the tracker of the split-bf16 filter off its critical path.
constexpr int kRerankLanes = 16;               // lanes per row
constexpr int kCandPad = 3 * kMaxSplit + 17;   // odd-ish stride: the row slots of a wave start in different LDS banks
template <int MODE, int DIM, int GT>
__device__ __forceinline__ void rerank_block(const RerankParams &p, const int vblock, const int nrows) {
  constexpr int GROUP = kRerankLanes;
  constexpr int RPW = 64 / GROUP;            // rows per wave
  constexpr int RPB = 4 * RPW;               // rows per block
  constexpr int NSI = kMaxSplit / GROUP;     // record passes per lane (code splits <= kMaxSplit)
  __shared__ int cand[RPB][kCandPad];
  __shared__ float s_ops[RPB][3 * DIM + 1];
  __shared__ float s_zhat[RPB][DIM + 1];
  __shared__ int s_best[RPB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % GROUP, grp = lane / GROUP;
  const int slot = wave * RPW + grp;
  const long pos_raw = (long)vblock * RPB + slot;
  const bool live = pos_raw < nrows;
  const long pos_c = live ? pos_raw : nrows - 1;                     // dead groups mirror the last row, write nothing
  const long row = p.level == 2 ? (long)p.fb_list[pos_c] : pos_c;
  const int gshift = grp * GROUP;
  const unsigned long long glow = (1ull << GROUP) - 1ull;
  auto group_bits = [&](bool c) { return (__ballot(c) >> gshift) & glow; };

  // ---- everything the row needs, issued together ---------------------------
  const float N1f = p.level == 2 ? p.hdr->absmax : wave_absmax(p.hdr->absmax_part, lane);
  const float R2f = (p.level == 1 && p.rowaux) ? wave_absmax(p.hdr->r2_part, lane) : 0.0f;   // max_j |cb_j|^2 (F16 bound)
  if (p.level == 1 && vblock == 0 && threadIdx.x == 0) p.hdr->absmax = N1f;   // for the tail kernel
  double rs[4];
  {
    const double *q = p.rowsum + row * 4;
    rs[0] = q[0]; rs[1] = q[1]; rs[2] = q[2]; rs[3] = q[3];
  }
  const float NEG_INF = -__builtin_inff();
  Rec r[NSI];
#pragma unroll
  for (int k = 0; k < NSI; ++k) {
    r[k].m1 = r[k].m2 = r[k].m3 = r[k].m4 = NEG_INF;
    r[k].id1 = r[k].id2 = r[k].id3 = 0;
    const int s = k * GROUP + sub;
    if (s < p.nsplit) r[k] = p.rec[(long)s * p.rows + row];
  }
  float cA[DIM], cB[DIM];
  auto load_row = [&](const float *src, float (&dst)[DIM]) {
    const f32x4 *q = reinterpret_cast<const f32x4 *>(src);
#pragma unroll
    for (int k = 0; k < DIM / 4; ++k) {
      const f32x4 v = q[k];
      dst[4 * k] = v.x; dst[4 * k + 1] = v.y; dst[4 * k + 2] = v.z; dst[4 * k + 3] = v.w;
    }
  };
  load_row(p.coef + row * 2 * DIM, cA);          // pass 1's operands: registers (16-byte broadcast loads)
  load_row(p.coef + row * 2 * DIM + DIM, cB);
  // pass 2's operands (mu | 2 sd^2 | log sd; used for the one or two codes that survive pass 1): LDS, one padded
  // record per row so that the four rows of a wave sit in different banks
  float *ops = s_ops[slot];
  for (int i = sub; i < DIM; i += GROUP) {
#pragma clang fp contract(off)
    ops[i] = p.mu[row * DIM + i];
    if constexpr (MODE == kModeGQ) {
      const float sg = p.sd[row * DIM + i];
      ops[DIM + i] = 2.0f * (sg * sg);
      ops[2 * DIM + i] = p.lsd[row * DIM + i];
    }
  }

  // ---- rounding bounds -> margins --------------------------------------------
  const double u = 5.9604644775390625e-08;  // 2^-24
  const double N1 = (double)N1f;
  double T, G;
  row_bound<MODE>(rs, N1, DIM, p.beta, T, G);
  const double Er = MODE == kModeGQ ? (DIM + 16.0) * u * G : 1e-12 * T;
  const float margin32 = (float)(2.5 * ((2.0 * DIM + 4.0) * u * T + Er) * 1.0000002 + 1e-30);   // around pass 1's F (rounded up)

  float fmax = NEG_INF;
#pragma unroll
  for (int k = 0; k < NSI; ++k) fmax = __builtin_fmaxf(fmax, r[k].m1);
#pragma unroll
  for (int o = GROUP / 2; o > 0; o >>= 1) fmax = __builtin_fmaxf(fmax, __shfl_xor(fmax, o));

  double margin = 2.5 * ((double)p.ef_coeff * u * T + Er) + 1e-30;      // around the filter's row maximum
  if (p.level == 1 && p.rowaux) {
    // fp16 main-product filter: the data-dependent bound (f16_bound above); 1.25 (Ea + Eb + 2 E_r) keeps the same 25 % slack
    float aux[8];
    const f32x4 *q = reinterpret_cast<const f32x4 *>(p.rowaux + row * 8);
    const f32x4 q0 = q[0], q1 = q[1];
    aux[0] = q0.x; aux[1] = q0.y; aux[2] = q0.z; aux[3] = q0.w; aux[4] = q1.x; aux[5] = q1.y; aux[6] = q1.z; aux[7] = q1.w;
    const double R2 = (double)R2f * (1.0 + 64.0 * u);       // the fp32 sum of squares of up to 64 dims, rounded up
    margin = 1.25 * (f16_bound(aux, T, Er, N1, R2, (double)fmax, DIM, (double)p.ef_coeff * u) + 2.0 * Er) + 1e-30;
  }
  bool bad = !(N1 == N1) || N1 > 1e18 || !(T < 1e30) || !(G < 1e30) || !(margin < 1e30);
  if (p.level == 1 && p.n1_limit > 0.f && !(N1f <= p.n1_limit && N1f >= p.n1_min)) bad = true;
  bad = bad || !(fmax == fmax) || !(fmax > NEG_INF) || !(fmax < __builtin_inff());

  const double thr = (double)fmax - margin;
  const unsigned long long lt = (1ull << sub) - 1ull;
  int total = 0;
  bool third = false;
#pragma unroll
  for (int k = 0; k < NSI; ++k) {
    const bool c1 = (double)r[k].m1 >= thr, c2 = (double)r[k].m2 >= thr, c3 = (double)r[k].m3 >= thr;
    const bool c4 = (double)r[k].m4 >= thr;
    const unsigned long long b1 = group_bits(c1), b2 = group_bits(c2), b3 = group_bits(c3);
    third = third || group_bits(c4) != 0ull;   // a fourth group of some split could matter: undecided
    const int n1 = __popcll(b1), n2 = __popcll(b2);
    if (c1) cand[slot][total + __popcll(b1 & lt)] = r[k].id1;
    if (c2) cand[slot][total + n1 + __popcll(b2 & lt)] = r[k].id2;
    if (c3) cand[slot][total + n1 + n2 + __popcll(b3 & lt)] = r[k].id3;
    total += n1 + n2 + __popcll(b3);
  }
  const bool undecided = bad || third;       // group-uniform
  if (undecided) {
    if (live && sub == 0) {
      const int pos = atomicAdd(p.level == 2 ? &p.hdr->fb2_count : &p.hdr->fb_count, 1);
      (p.level == 2 ? p.fb2_list : p.fb_list)[pos] = (int)row;
      if (pos < kSpreadRows) p.spread[pos].done = 0;
    }
    total = 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // this lane's code t (0 .. GT-1) of candidate id: lane half (id & 1) of tile (id >> 1) * GT + t
  const int code_in_tile = (sub & 3) + 8 * (sub >> 2);
  auto code_of = [&](int id, int t) { return ((id >> 1) * GT + t) * kTileCodes + code_in_tile + 4 * (id & 1); };
  auto expansion = [&](const float (&n)[DIM]) {   // fp32 FMA chain, the fp32 filter's operands
    float f = 0.0f;
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      f = __builtin_fmaf(cA[i], n[i] * n[i], f);
      f = __builtin_fmaf(cB[i], n[i], f);
    }
    return f;
  };
  const int wave_total = [&] {   // wave-uniform trip count
    int t = total;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t = max(t, __shfl_xor(t, o));
    return t;
  }();

  // ---- pass 1: the fp32 expansion of every code of every candidate.  Per lane: the best value with its code row
  // (kept in registers: it is almost always the only code of this lane that pass 2 wants) and the runner-up value;
  // per row: F = the maximum over the group's 16 lanes. ----
  constexpr int UNR = GT * DIM <= 32 ? GT : (DIM >= 32 ? 1 : 32 / DIM);     // code rows in flight per lane (<= 32 registers)
  float nb[DIM], fb = NEG_INF, fsecond = NEG_INF;
  int codeb = -1;
#pragma unroll
  for (int i = 0; i < DIM; ++i) nb[i] = 0.0f;
  for (int e = 0; e < wave_total; ++e) {
    if (e < total) {
      const int id = cand[slot][e];
#pragma unroll
      for (int t0 = 0; t0 < GT; t0 += UNR) {
        float n[UNR][DIM];
#pragma unroll
        for (int t = 0; t < UNR; ++t) {       // out-of-range codes read the last code row and are skipped below
          const int code = code_of(id, t0 + t);
          load_row(p.cb + (long)(code < p.n ? code : p.n - 1) * DIM, n[t]);
        }
#pragma unroll
        for (int t = 0; t < UNR; ++t) {
          const int code = code_of(id, t0 + t);
          float f = expansion(n[t]);
          if (f != f) f = __builtin_inff();   // a NaN value: "keep everything" below
          if (code < p.n) {
            const bool better = f > fb || codeb < 0;
            fsecond = __builtin_fmaxf(fsecond, better ? fb : f);
            if (better) {
              fb = f;
              codeb = code;
#pragma unroll
              for (int i = 0; i < DIM; ++i) nb[i] = n[t][i];
            }
          }
        }
      }
    }
  }
  float F = fb;
#pragma unroll
  for (int o = GROUP / 2; o > 0; o >>= 1) F = __builtin_fmaxf(F, __shfl_xor(F, o));
  const float thr32 = F - margin32;
  const bool keep_all = !(F < __builtin_inff()) || !(margin32 < 1e30f);

  // ---- pass 2: the reference's own score for the codes inside the window ----
  double best_s = 0.0;
  int best_i = 0x7fffffff;
  bool have = false;
  auto exact = [&](const float (&n)[DIM], int code) {
    double s;
    if constexpr (MODE == kModeGQ) s = (double)ref_score_lds<DIM>(n, ops, p.beta);
    else s = vq_neg_dist(n, ops, DIM);
    if (!have || better_d(s, code, best_s, best_i)) {
      best_s = s;
      best_i = code;
      have = true;
    }
  };
  if (codeb >= 0 && (keep_all || !(fb < thr32))) exact(nb, codeb);
  // a second code of the SAME lane inside the window (rare: a near-tie within one lane's few codes): go through
  // this lane's codes again
  if (__any(codeb >= 0 && (keep_all || !(fsecond < thr32)) && fsecond > NEG_INF)) {
    for (int e = 0; e < wave_total; ++e) {
      if (e < total && (keep_all || !(fsecond < thr32))) {
        const int id = cand[slot][e];
        for (int t = 0; t < GT; ++t) {
          const int code = code_of(id, t);
          if (code < p.n && code != codeb) {
            float n[DIM];
            load_row(p.cb + (long)code * DIM, n);
            float f = expansion(n);
            if (keep_all || !(f < thr32)) exact(n, code);   // a NaN value passes
          }
        }
      }
    }
  }
#pragma unroll
  for (int o = GROUP / 2; o > 0; o >>= 1) {
    const double os = __shfl_xor(best_s, o);
    const int oi = __shfl_xor(best_i, o);
    const bool oh = __shfl_xor((int)have, o) != 0;
    if (oh && (!have || better_d(os, oi, best_s, best_i))) {
      best_s = os;
      best_i = oi;
      have = true;
    }
  }
  const bool decided = live && !undecided;
  if (p.stats && decided && sub == 0) atomicAdd(&p.hdr->reranked, (unsigned long long)total);
  if (p.level == 2) {   // listed rows are scattered: write directly
    if (decided) {
      if (sub == 0) p.idx[out_idx_offset(p.omap, row)] = (int64_t)best_i;
      if (p.zhat)
        for (int i = sub; i < DIM; i += GROUP)
          p.zhat[out_zhat_offset(p.omap, row, i, DIM)] = p.cb[(long)best_i * DIM + i];
    }
    return;
  }
  // level 1: the block's RPB consecutive rows leave as contiguous runs (BCHW: along l per channel)
  if (sub == 0) s_best[slot] = decided ? best_i : -1;
  if (p.zhat && decided) {
    // the winner's code row is almost always still in a lane's registers (nb: the lane's best code of pass 1): no second
    // round trip to the codebook for zhat.  Otherwise (the winner came out of the rare second scan) it is loaded.
    const unsigned long long holders = group_bits(codeb == best_i && codeb >= 0);
    if (holders != 0ull) {
      if (sub == __builtin_ctzll(holders)) {
#pragma unroll
        for (int i = 0; i < DIM; ++i) s_zhat[slot][i] = nb[i];
      }
    } else {
      for (int i = sub; i < DIM; i += GROUP) s_zhat[slot][i] = p.cb[(long)best_i * DIM + i];
    }
  }
  __syncthreads();
  const long row0 = (long)vblock * RPB;
  if (threadIdx.x < RPB) {
    const int b = s_best[threadIdx.x];
    if (b >= 0) p.idx[out_idx_offset(p.omap, row0 + threadIdx.x)] = (int64_t)b;
  }
  if (p.zhat) {
    for (int j = threadIdx.x; j < RPB * DIM; j += 256) {
      int lr, g;
      if (p.omap.mode == 1) { lr = j % RPB; g = j / RPB; } else { lr = j / DIM; g = j % DIM; }
      if (s_best[lr] >= 0) p.zhat[out_zhat_offset(p.omap, row0 + lr, g, DIM)] = s_zhat[lr][g];
    }
  }
}

// (256, 4): at most 128 VGPRs, four blocks per CU -- the kernel lives on memory-level parallelism across waves
// (dim 32 needs 2 x 32 coefficient registers alone: two blocks per CU there)
template <int MODE, int DIM, int GT>
__global__ __launch_bounds__(256, DIM >= 32 ? 2 : 4) void gq_rerank_kernel(const RerankParams p) {
  rerank_block<MODE, DIM, GT>(p, (int)blockIdx.x, p.rows);
}

// Second-stage filter for the rows the fp32 filter could not decide (fallback list).
// When one sigma is tiny the expansion A n^2 + B n cancels catastrophically in fp32 and hundreds
// of codes fall inside the fp32 margin; in fp64 the same expansion is accurate to ~1e-16 * T, so
// the candidate window shrinks to the reference's OWN rounding noise around the maximum:
//   |s_ref(j) - s(j)| <= c_u * (sum_i |t_ji| + R),  sum_i |t_ji| = C_r + (beta/2)|n_j|^2 - f(j)
// (all t <= 0), i.e. for codes near the maximum it is small.  With g_j = fmax - f_j and
// Q = C_r + beta*dim*N2/2 - fmax, a code can win only if g_j <= 2 c_u (Q + R) / (1 - c_u).
// Pass 1: fp64 row maxima; pass 2: re-evaluate, exact reference-order score for the few codes
// inside the window.  A block = 8 rows x 32 code lanes: a thread keeps its row's fp64 coefficients
// in registers and walks every 32nd code, so the 8 rows of a block share each code row through L1.
// Rows with non-finite operands get an infinite window = the exhaustive semantics.
constexpr int kFallbackRows = 8;

template <int MODE, int DIM>
__device__ __forceinline__ void fallback64_long_list(const RerankParams &p, const int vblock, const int nvblocks) {
  constexpr int FR = kFallbackRows;
  __shared__ RowOps rops[FR];
  __shared__ double sh_d[8];
  __shared__ int sh_i[8];
  const int tid = threadIdx.x;
  const int *list;
  int count;
  second_stage_list(p, list, count);
  // Many listed rows: 8 rows per block (a half-wave each).  Few: the whole block on ONE row, so a
  // lone fallback row costs ~60 us instead of ~3 ms.
  const bool wide = count < 4 * nvblocks;              // block-uniform
  const int R = wide ? 1 : FR;
  const int r = wide ? 0 : tid >> 5;                   // row slot
  const int cl = wide ? tid : tid & 31;                // code lane
  const int cstride = wide ? 256 : 32;
  const double INF = __builtin_inf();
  for (int grp = vblock; grp * R < count; grp += nvblocks) {
    const int nrow = min(R, count - grp * R);
    const bool live = r < nrow;
    const long row = live ? list[grp * R + r] : 0;
    __syncthreads();
    if (live && (tid & 31) < DIM && (wide ? tid < 32 : true)) load_row_ops(p, row, tid & 31, rops[r]);
    if (live && (tid & 31) + 32 < DIM && (wide ? tid < 32 : true)) load_row_ops(p, row, (tid & 31) + 32, rops[r]);
    __syncthreads();
    // fp64 coefficients + the window constants of this thread's row (redundant per lane, cheap)
    double cA[DIM], cB[DIM];
    const double u = 5.9604644775390625e-08, N1 = (double)p.hdr->absmax, N2 = N1 * N1;
    const double bb = fabs((double)p.beta), c = (double)half_log_2pi();
    double Cr = 0.0, R0 = 0.0, T = 0.0;
    bool bad = !live || !(N1 == N1) || N1 > 1e18;
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      const double m = live ? (double)rops[r].mu[i] : 0.0;
      if constexpr (MODE == kModeGQ) {
        const double sg = live ? (double)p.sd[row * DIM + i] : 1.0;
        const double inv = 1.0 / (sg * sg);
        cA[i] = 0.5 * (double)p.beta - 0.5 * inv;
        cB[i] = m * inv;
        Cr += 0.5 * m * m * inv;
        R0 += fabs(live ? (double)rops[r].lsd[i] : 0.0) + c + bb * (0.5 * N2 + c);
        T += (0.5 * bb + 0.5 * inv) * N2 + fabs(m) * inv * N1;
        bad = bad || !(sg > 0.0) || !(inv < 1e300);
      } else {
        cA[i] = -1.0;
        cB[i] = 2.0 * m;
        T += N2 + 2.0 * fabs(m) * N1;
      }
    }
    auto f64_of = [&](int j) {
      const f32x4 *nj = reinterpret_cast<const f32x4 *>(p.cb + (long)j * DIM);
      double f = 0.0;
#pragma unroll
      for (int q = 0; q < DIM / 4; ++q) {
        const f32x4 v4 = nj[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double v = (double)v4[k];
          f = fma(cA[4 * q + k], v * v, f);
          f = fma(cB[4 * q + k], v, f);
        }
      }
      return f;
    };
    // ---- pass 1: fp64 maximum of the row --------------------------------------------------
    double fmax = -INF;
    if (live)
      for (int j = cl; j < p.n; j += cstride) {
        const double f = f64_of(j);
        fmax = f > fmax ? f : fmax;          // NaN never enters
      }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const double of = __shfl_xor(fmax, o);
      fmax = of > fmax ? of : fmax;
    }
    if (wide) {                               // combine the block's 8 half-waves
      if ((tid & 31) == 0) sh_d[tid >> 5] = fmax;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 8; ++k) fmax = sh_d[k] > fmax ? sh_d[k] : fmax;
      __syncthreads();
    }
    double marg;
    if constexpr (MODE == kModeGQ) {
      const double cu = (DIM + 16.0) * u;
      double Q = Cr + 0.5 * bb * DIM * N2 - fmax;
      Q = Q > 0.0 ? Q : 0.0;
      marg = 2.5 * cu * (Q + R0) / (1.0 - cu) + 1e-12 * T + 1e-30;
    } else {
      marg = 1e-11 * T + 1e-30;
    }
    bad = bad || !(fmax > -INF) || !(fmax < INF) || !(T < 1e300) || !(marg < 1e300);
    const double thr = bad ? -INF : fmax - marg;
    // ---- pass 2: exact reference-order scores inside the window ----------------------------
    double best_s = 0.0;
    int best_i = 0x7fffffff;
    bool have = false;
    if (live)
      for (int j = cl; j < p.n; j += cstride) {
        if (bad || f64_of(j) >= thr) {
          const double s = exact_score_cold<MODE>(p.cb, &rops[r], j, DIM, p.beta);
          if (!have || better_d(s, j, best_s, best_i)) { best_s = s; best_i = j; have = true; }
        }
      }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const double os = __shfl_xor(best_s, o);
      const int oi = __shfl_xor(best_i, o);
      const bool oh = __shfl_xor((int)have, o) != 0;
      if (oh && (!have || better_d(os, oi, best_s, best_i))) { best_s = os; best_i = oi; have = true; }
    }
    if (wide) {
      if ((tid & 31) == 0) { sh_d[tid >> 5] = best_s; sh_i[tid >> 5] = have ? best_i : 0x7fffffff; }
      __syncthreads();
      have = false;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double os = sh_d[k];
        const int oi = sh_i[k];
        if (oi != 0x7fffffff && (!have || better_d(os, oi, best_s, best_i))) { best_s = os; best_i = oi; have = true; }
      }
      __syncthreads();
    }
    const int ol = wide ? tid : tid & 31;     // output lane
    if (live && (wide ? tid < 64 : true)) {
      if (ol == 0) p.idx[out_idx_offset(p.omap, row)] = (int64_t)best_i;
      if (p.zhat && ol < DIM) p.zhat[out_zhat_offset(p.omap, row, ol, DIM)] = p.cb[(long)best_i * DIM + ol];
      if (!wide && p.zhat && ol + 32 < DIM)
        p.zhat[out_zhat_offset(p.omap, row, ol + 32, DIM)] = p.cb[(long)best_i * DIM + ol + 32];
    }
  }
}

// ---- short lists: every listed row is spread over kSpreadSlices blocks ------------------------------
// One launch.  A block owns a slice of the codes: fp64 values of the expansion for its codes (kept in
// registers), the slice maximum, then the exact reference-order score of every code inside the window taken
// around the SLICE maximum.  That window contains the window around the row maximum (slice max <= row max, and
// the margin grows as the maximum drops), so the union over the slices is a superset of the codes the row-wide
// rule would re-evaluate -- and the reference arg-max is the exact maximum over any set that contains it.
// Per-slice partial results go to the workspace; the last slice to finish combines them.  A lone undecided row
// costs a few microseconds instead of one block walking all 65 536 codes twice.
constexpr int kSpreadCodes = 16;    // codes per thread held in registers (slice <= 256 * kSpreadCodes codes)

template <int MODE, int DIM>
__device__ __forceinline__ void fallback64_spread(const RerankParams &p, const int vblock) {
  __shared__ RowOps rops;
  __shared__ double sh_d[4];
  __shared__ int sh_i[4];
  __shared__ int sh_last;
  const int *list;
  int count;
  second_stage_list(p, list, count);
  const int e = vblock / kSpreadSlices, sl = vblock % kSpreadSlices;
  if (e >= count) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row = list[e];
  if (tid < DIM) load_row_ops(p, row, tid, rops);
  __syncthreads();
  double cA[DIM], cB[DIM];
  const double u = 5.9604644775390625e-08, N1 = (double)p.hdr->absmax, N2 = N1 * N1;
  const double bb = fabs((double)p.beta), c = (double)half_log_2pi();
  const double INF = __builtin_inf();
  double Cr = 0.0, R0 = 0.0, T = 0.0;
  bool bad = !(N1 == N1) || N1 > 1e18;
#pragma unroll
  for (int i = 0; i < DIM; ++i) {
    const double m = (double)rops.mu[i];
    if constexpr (MODE == kModeGQ) {
      const double sg = (double)p.sd[row * DIM + i];
      const double inv = 1.0 / (sg * sg);
      cA[i] = 0.5 * (double)p.beta - 0.5 * inv;
      cB[i] = m * inv;
      Cr += 0.5 * m * m * inv;
      R0 += fabs((double)rops.lsd[i]) + c + bb * (0.5 * N2 + c);
      T += (0.5 * bb + 0.5 * inv) * N2 + fabs(m) * inv * N1;
      bad = bad || !(sg > 0.0) || !(inv < 1e300);
    } else {
      cA[i] = -1.0;
      cB[i] = 2.0 * m;
      T += N2 + 2.0 * fabs(m) * N1;
    }
  }
  auto f64_of = [&](int j) {
    const f32x4 *nj = reinterpret_cast<const f32x4 *>(p.cb + (long)j * DIM);
    double f = 0.0;
#pragma unroll
    for (int q = 0; q < DIM / 4; ++q) {
      const f32x4 v4 = nj[q];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double v = (double)v4[k];
        f = fma(cA[4 * q + k], v * v, f);
        f = fma(cB[4 * q + k], v, f);
      }
    }
    return f;
  };
  const int per = (p.n + kSpreadSlices - 1) / kSpreadSlices;
  const int j0 = sl * per, j1 = min(p.n, j0 + per);
  const bool in_regs = per <= 256 * kSpreadCodes;      // block-uniform; otherwise the values are recomputed
  double fv[kSpreadCodes];
  double fmax = -INF;
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < kSpreadCodes; ++k) {
      const int j = j0 + tid + 256 * k;
      fv[k] = j < j1 ? f64_of(j) : -INF;
      fmax = fv[k] > fmax ? fv[k] : fmax;              // NaN never enters
    }
  } else {
    for (int j = j0 + tid; j < j1; j += 256) {
      const double f = f64_of(j);
      fmax = f > fmax ? f : fmax;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double of = __shfl_xor(fmax, o);
    fmax = of > fmax ? of : fmax;
  }
  if (lane == 0) sh_d[wave] = fmax;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) fmax = sh_d[k] > fmax ? sh_d[k] : fmax;
  __syncthreads();
  double marg;
  if constexpr (MODE == kModeGQ) {
    const double cu = (DIM + 16.0) * u;
    double Q = Cr + 0.5 * bb * DIM * N2 - fmax;
    Q = Q > 0.0 ? Q : 0.0;
    marg = 2.5 * cu * (Q + R0) / (1.0 - cu) + 1e-12 * T + 1e-30;
  } else {
    marg = 1e-11 * T + 1e-30;
  }
  // a slice without a finite value contributes nothing unless the row itself is degenerate (then: exhaustive)
  bad = bad || !(fmax < INF) || !(T < 1e300) || !(marg < 1e300) || (fmax != fmax);
  const bool empty = !bad && !(fmax > -INF);
  const double thr = bad ? -INF : fmax - marg;
  double best_s = 0.0;
  int best_i = 0x7fffffff;
  bool have = false;
  if (!empty) {
    if (in_regs) {
#pragma unroll
      for (int k = 0; k < kSpreadCodes; ++k) {
        const int j = j0 + tid + 256 * k;
        if (j < j1 && (bad || fv[k] >= thr)) {
          const double s = exact_score_cold<MODE>(p.cb, &rops, j, DIM, p.beta);
          if (!have || better_d(s, j, best_s, best_i)) { best_s = s; best_i = j; have = true; }
        }
      }
    } else {
      for (int j = j0 + tid; j < j1; j += 256) {
        if (bad || f64_of(j) >= thr) {
          const double s = exact_score_cold<MODE>(p.cb, &rops, j, DIM, p.beta);
          if (!have || better_d(s, j, best_s, best_i)) { best_s = s; best_i = j; have = true; }
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double os = __shfl_xor(best_s, o);
    const int oi = __shfl_xor(best_i, o);
    const bool oh = __shfl_xor((int)have, o) != 0;
    if (oh && (!have || better_d(os, oi, best_s, best_i))) { best_s = os; best_i = oi; have = true; }
  }
  if (lane == 0) { sh_d[wave] = best_s; sh_i[wave] = have ? best_i : 0x7fffffff; }
  __syncthreads();
  SpreadSlot &slot = p.spread[e];
  if (tid == 0) {
    have = false;
    for (int k = 0; k < 4; ++k) {
      const double os = sh_d[k];
      const int oi = sh_i[k];
      if (oi != 0x7fffffff && (!have || better_d(os, oi, best_s, best_i))) { best_s = os; best_i = oi; have = true; }
    }
    slot.part[sl].s = best_s;
    slot.part[sl].i = have ? best_i : 0x7fffffff;
    __threadfence();
    sh_last = atomicAdd(&slot.done, 1) == kSpreadSlices - 1;
  }
  __syncthreads();
  if (!sh_last) return;
  __threadfence();
  // last slice of this row: combine the partial results (one wave) and write the answer
  if (wave == 0) {
    have = false;
    best_s = 0.0;
    best_i = 0x7fffffff;
    if (lane < kSpreadSlices) {
      const volatile SpreadPartial *pp = &slot.part[lane];
      best_s = pp->s;
      best_i = pp->i;
      have = best_i != 0x7fffffff;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double os = __shfl_xor(best_s, o);
      const int oi = __shfl_xor(best_i, o);
      const bool oh = __shfl_xor((int)have, o) != 0;
      if (oh && (!have || better_d(os, oi, best_s, best_i))) { best_s = os; best_i = oi; have = true; }
    }
    write_result(p, row, best_i, lane);
  }
}

// The fp64 second stage: the list length (known only on the device) picks the variant.  Any grid size works
// (both variants walk virtual blocks).
template <int MODE, int DIM>
__device__ __forceinline__ void second_stage(const RerankParams &p) {
  const int *list;
  int count;
  second_stage_list(p, list, count);
  if (count == 0) return;
  if (count <= kSpreadRows) {
    for (int vb = blockIdx.x; vb < count * kSpreadSlices; vb += gridDim.x) {
      fallback64_spread<MODE, DIM>(p, vb);
      __syncthreads();
    }
  } else {
    fallback64_long_list<MODE, DIM>(p, (int)blockIdx.x, (int)gridDim.x);
  }
}

// Grid barrier of the tail kernel.  The host sizes the grid so that every block is co-resident (occupancy API x CU
// count), but nothing in HIP guarantees that: a CU mask, another stream holding CUs or a profiler can leave blocks
// queued behind spinning ones.  So the barrier is allowed to FAIL: it returns false when this block's spin ran out or
// any block has reported that (hdr->bar_gen == kBarAbort).  From then on nobody waits at a barrier, and every block finishes
// list A through exhaustive_rows() below, which depends on no other block (gq_tail.h) -- a failed barrier costs time,
// never a wrong index.  A block only consumes other blocks' data behind a barrier that returned true, i.e. after
// all `nblocks` arrivals, each made after the arriving block's own phase was complete and released.
// Producer side: every wave drains its stores, the block meets, lane 0 releases at agent scope and arrives;
// consumer side: relaxed agent-scope poll, ONE acquire, block barrier (MI355X_MICROARCH.md, inter-workgroup
// visibility: per-XCD L2s are not coherent, a CU's L1 is never refreshed by other CUs' stores).
// The OUTCOME of a barrier is one atomic decision (round 4): the last arriver compare-and-swaps `bar_gen` from this
// barrier's generation to the next one, a block whose wait ran out compare-and-swaps it to kBarAbort; whichever lands first stands
// (an aborted `bar_gen` never changes again within the call), and EVERY block derives its return value from that one final word --
// never from its own view of the race.  So either all blocks pass a barrier or all of them leave it for the barrier-free finish.
constexpr unsigned kBarAbort = 0xffffffffu;
__device__ __forceinline__ bool barrier_aborted(WsHeader *hdr) {
  return __hip_atomic_load(&hdr->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kBarAbort;
}
__device__ __forceinline__ bool grid_barrier(WsHeader *hdr, unsigned nblocks, int spin_limit) {
  __shared__ int sh_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned gen = __hip_atomic_load(&hdr->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = false;
    if (gen != kBarAbort) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned arrived = __hip_atomic_fetch_add(&hdr->bar_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (arrived == nblocks - 1) {
        __hip_atomic_store(&hdr->bar_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned expect = gen;
        const unsigned next = gen + 1u == kBarAbort ? 0u : gen + 1u;
        (void)__hip_atomic_compare_exchange_strong(&hdr->bar_gen, &expect, next, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);      // fails only against an abort: that stands
      } else {
        // bounded (MI355X_MICROARCH.md: "bound every spin"): the default limit is ~0.5 s, orders of magnitude beyond
        // any real wait of a co-resident grid
        int spins = 0;
        while (__hip_atomic_load(&hdr->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > spin_limit) {
            unsigned expect = gen;
            if (__hip_atomic_compare_exchange_strong(&hdr->bar_gen, &expect, kBarAbort, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT))
              __hip_atomic_store(&hdr->bar_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (diagnostics mirror)
            __hip_atomic_fetch_add(&hdr->bar_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // the one final word: the next generation (every block had arrived first) or the abort (a wait had run out first)
      ok = __hip_atomic_load(&hdr->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kBarAbort;
    }
    sh_ok = ok;
  }
  __syncthreads();
  return sh_ok != 0;
}

// Exhaustive exact arg-max of rows list[first], list[first + stride], ... (list == NULL: the rows themselves): one
// block per row, every code scored in the reference's operation order.  Depends on nothing but the row operands and
// the codebook, so it is also what the tail kernel falls back to when a grid barrier fails.
template <int MODE>
__device__ __forceinline__ void exhaustive_rows(const RerankParams &p, const int *list, int count, int first, int stride) {
  __shared__ double sh_s[256];
  __shared__ int sh_i[256];
  __shared__ RowOps rops;
  const int tid = threadIdx.x;
  for (int e = first; e < count; e += stride) {
    const long row = list ? list[e] : e;
    if (tid < p.dim) load_row_ops(p, row, tid, rops);
    __syncthreads();
    double best_s = 0.0;
    int best_i = 0x7fffffff;
    bool have = false;
    for (int j = tid; j < p.n; j += 256) {
      const double s = exact_score<MODE>(p, rops, j);
      if (!have || better_d(s, j, best_s, best_i)) {
        best_s = s;
        best_i = j;
        have = true;
      }
    }
    sh_s[tid] = best_s;
    sh_i[tid] = have ? best_i : 0x7fffffff;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        const double os = sh_s[tid + o];
        const int oi = sh_i[tid + o];
        const bool mine = sh_i[tid] != 0x7fffffff;
        if (oi != 0x7fffffff && (!mine || better_d(os, oi, sh_s[tid], sh_i[tid]))) {
          sh_s[tid] = os;
          sh_i[tid] = oi;
        }
      }
      __syncthreads();
    }
    const int best = sh_i[0];
    __syncthreads();
    write_result(p, row, best, tid);
  }
}

// Exhaustive exact arg-max as a kernel: rows the filter could not decide, and shapes the MFMA
// filter does not cover (dim not in {4,8,16,32}).
template <int MODE>
__global__ __launch_bounds__(256) void gq_exhaustive_kernel(const RerankParams p) {
  const int count = p.all_rows ? p.rows : p.hdr->fb_count;
  exhaustive_rows<MODE>(p, p.all_rows ? nullptr : p.fb_list, count, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace gqhip
