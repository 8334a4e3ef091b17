// gq_rerank.h -- exact re-rank of the filter's candidates, the in-block finish of the rows the filter could not decide,
// and the exhaustive kernel.  All of them evaluate the reference's score in the reference's operation order
// (gq_common.h:ref_term), so the winning index is the one torch.argmax returns on the reference's CPU path
// (pit/quantization/gaussian.py:142-150).
//
// Why a small candidate set is enough: for every code j,
// |filter(r,j) + const(r) - ref_score(r,j)| <= E(r,j), a rigorous rounding bound computed below from the row's own sums,
// max|cb| and dim.  The reference's arg-max j* therefore lies in a half-group whose filter maximum is within `margin` of the
// row maximum.  The filter keeps, per (row, record set), the best three such half-groups by id and the fourth by value.
//   * No record set has its fourth value within the margin (almost every row): the candidates are complete, the row is
//     decided from them (two passes: fp32 expansion, then the reference's own arithmetic for the one or two codes that
//     survive it).
//   * Otherwise (round 4: this replaces the separate tail launch with its cascade, grid barriers and fp64 stage) the row is
//     finished INSIDE this kernel, by its own block, right after the block's decided rows have left: every record set that
//     holds any group within the margin is scanned completely -- fp32 expansion of every code of the set, the reference's
//     arithmetic for the codes above  F - margin - margin32  --, which is a superset of everything the reference's arg-max
//     can be.  A set is 2 048 ... 8 192 codes: a microsecond or two of one block's time per set.  Rows with non-finite
//     operands or bounds scan every set and keep every code (= the exhaustive semantics).
// Either way no approximation reaches the output, the call is three launches (prep, filter, re-rank), and no block ever
// waits for another one.
#pragma once
#include "gq_common.h"
#include "gq_filter.h"
#include "gq_gauss.h"

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
// The value stored at offset o of zhat for codeword element v: the straight-through mix of the module's eval forward, in the
// reference's fp32 op order (also stores v itself to `pure` when that is given).  The three words live in the workspace HEADER (written
// by the call's first launch), not in the kernel arguments: they are read here, at the very end of a row's life, and cost the hot
// loops no scalar registers (as kernel arguments they spilled 6-10 SGPRs in every re-rank / search kernel).
__device__ __forceinline__ float ste_mix(const WsHeader *h, long o, float v) {
#pragma clang fp contract(off)
  const int kind = h->ste_kind;
  float *pure = h->pure;
  if (pure) pure[o] = v;
  if (kind == 0) return v;
  const float g = h->ste[o];
  if (kind == 1) return (g - g) + v;     // finite g: + 0 + v; inf / NaN: NaN, as in the reference
  const float d = v - g;
  return g + d;
}

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
  int *fb_list;       // [rows]  exhaustive kernel only: its row list
  int rows, n, dim;
  float beta;
  int nsplit;         // record sets per row (<= kMaxSplit): the filter's code splits, times 2 when it leaves one record per lane half
  int rec_halves;     // 2: one record set per (split, lane half) -- a set holds codes (r & 3) + 8 (r >> 2) + 4 h of every tile of
                      // its split; 1: one per split (all 32 codes of every tile)
  int tiles_per_split, tiles_total;   // the filter's plan (set -> codes)
  int gt;             // tiles per candidate group -- must match the filter's GT
  float ef_coeff;     // filter error bound E_f = ef_coeff * 2^-24 * T  (fp32 filter: 2 dim + 4; split-bf16: 220 + 24 dim;
                      // fp16 + fp8: 2450; fp16 main product: 16700 against the data-dependent T of f16_bound)
  float n1_limit;     // > 0: the filter's operand formats assume n1_min <= max|cb| <= n1_limit (fp16 + fp8 images: 1 .. 16; below
                      // 1 the absolute errors of fp8-subnormal operands are not covered by the bound; fp16 images: 0 .. 255, the
                      // squares must stay below 65504); with any other codebook every row is finished by the in-block scan
  float n1_min;
  const float *rowaux;   // [rows, 8] sums of the data-dependent bound (gq_prep_kernel, F16) or NULL: the classic bound k u T
  void *dbg;          // diagnostic builds only (GQHIP_CLOCK_STAMPS: phase stamps of wave 0 of the first 1024 blocks)
  int all_rows;       // exhaustive kernel: process every row (no filter ran)
  int stats;          // count re-ranked half-pairs (debug)
  OutMap omap;
};

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

// comparator on (double score, index) with torch.argmax semantics
__device__ __forceinline__ bool better_d(double sa, int ia, double sb, int ib) {
  const bool na = sa != sa, nb = sb != sb;
  if (na || nb) return na && (!nb || ia < ib);
  return sa > sb || (sa == sb && ia < ib);
}

__device__ __forceinline__ void write_result(const RerankParams &p, long row, int best, int lane) {
  if (lane == 0) p.idx[out_idx_offset(p.omap, row)] = (int64_t)best;
  if (p.zhat && lane < p.dim) {
    const long o = out_zhat_offset(p.omap, row, lane, p.dim);
    p.zhat[o] = ste_mix(p.hdr, o, p.cb[(long)best * p.dim + lane]);
  }
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
// here), and the results of a block's 16 consecutive rows leave through LDS as contiguous runs in the module layout.
//
// Two passes over the candidates' codes.  Pass 1 evaluates the filter expansion f^(j) = sum_i A_i n_ji^2 + B_i n_ji as a
// plain fp32 FMA chain (2 dim FMAs, no division) and takes the group-wide maximum F.  Its error is that of the fp32 MFMA
// filter, |f^ - f| <= E32 = (2 dim + 4) u T, so by the same argument as for the filter the reference's arg-max j* --
// which IS among the candidates -- satisfies f^(j*) >= F - 2 (E32 + E_r).  Pass 2 therefore evaluates the reference's
// own score (ref_term: one IEEE division per dimension) only for the codes with f^ >= F - 2.5 (E32 + E_r): one or two
// per row instead of all 16 gt.  That makes coarse candidates (gt = 4: 64 codes) cheap here, and coarse candidates are
// what keeps the tracker of the filter off its critical path.
//
// NSI = record passes per lane: 16 NSI >= nsplit (the launcher picks 1 for up to 16 record sets -- every BASELINE shape --
// and 4 otherwise, so that the common case does not walk three empty passes).
constexpr int kRerankLanes = 16;               // lanes per row

// The in-block finish of ONE undecided row (block-wide; see the head of this file).  `mask`: record sets to scan; codes with
// f^ >= thr (or all of them: keep_all) get the reference's own arithmetic; the winner leaves in torch.argmax order.
// (Inlined at the block's very end: by then the hot path's registers are dead, and a call would cost the kernel a stack.)
template <int MODE, int DIM>
__device__ __forceinline__ void finish_row_by_scan(const RerankParams &p, long row, unsigned long long mask, float thr,
                                                             bool keep_all, const float *ops, double *sh_s, int *sh_i) {
  const int tid = threadIdx.x;
  float cA[DIM], cB[DIM];
  {
    const f32x4 *q = reinterpret_cast<const f32x4 *>(p.coef + row * 2 * DIM);
#pragma unroll
    for (int k = 0; k < DIM / 4; ++k) {
      const f32x4 a = q[k], b = q[DIM / 4 + k];
      cA[4 * k] = a.x; cA[4 * k + 1] = a.y; cA[4 * k + 2] = a.z; cA[4 * k + 3] = a.w;
      cB[4 * k] = b.x; cB[4 * k + 1] = b.y; cB[4 * k + 2] = b.z; cB[4 * k + 3] = b.w;
    }
  }
  double best_s = 0.0;
  int best_i = 0x7fffffff;
  bool have = false;
  const int cpt = p.rec_halves == 2 ? 16 : 32;          // codes of a set per tile
  for (int s = 0; s < p.nsplit; ++s) {
    if (!((mask >> s) & 1ull)) continue;                // block-uniform
    const int split = p.rec_halves == 2 ? s >> 1 : s, h = s & 1;
    const int t0 = split * p.tiles_per_split;
    const int t1 = min(t0 + p.tiles_per_split, p.tiles_total);
    const int items = (t1 - t0) * cpt;
    // U codes per thread and pass, their rows loaded BEFORE the first expansion: the scan is a chain of L2 latencies (one 4 DIM-byte
    // row per thread in flight and ~0.7 us each: 270 us per 65 536 codes at dim 16 with the plain loop), so the loads in flight are
    // what sets its speed
    constexpr int U = DIM <= 8 ? 8 : (DIM == 16 ? 3 : 4);
    for (int it0 = tid; it0 < items; it0 += 256 * U) {
      float n[U][DIM];
      int code[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it = it0 + 256 * u;
        const int itc = it < items ? it : it0;            // (clamped: a load that is not used)
        const int tile = t0 + itc / cpt, c = itc % cpt;
        code[u] = tile * kTileCodes + (p.rec_halves == 2 ? (c & 3) + 8 * (c >> 2) + 4 * h : c);
        ok[u] = it < items && code[u] < p.n;
        const f32x4 *q = reinterpret_cast<const f32x4 *>(p.cb + (long)(code[u] < p.n ? code[u] : p.n - 1) * DIM);
#pragma unroll
        for (int k = 0; k < DIM / 4; ++k) {
          const f32x4 v = q[k];
          n[u][4 * k] = v.x; n[u][4 * k + 1] = v.y; n[u][4 * k + 2] = v.z; n[u][4 * k + 3] = v.w;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f = 0.0f;
#pragma unroll
        for (int i = 0; i < DIM; ++i) f = __builtin_fmaf(__builtin_fmaf(cA[i], n[u][i], cB[i]), n[u][i], f);
        if (ok[u] && (keep_all || !(f < thr))) {          // a NaN value passes
          double sc;
          if constexpr (MODE == kModeGQ) sc = (double)ref_score_lds<DIM>(n[u], ops, p.beta);
          else sc = vq_neg_dist(n[u], ops, DIM);
          if (!have || better_d(sc, code[u], best_s, best_i)) { best_s = sc; best_i = code[u]; have = true; }
        }
      }
    }
  }
  __syncthreads();
  sh_s[tid] = best_s;
  sh_i[tid] = have ? best_i : 0x7fffffff;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      const double os = sh_s[tid + o];
      const int oi = sh_i[tid + o];
      const bool mine = sh_i[tid] != 0x7fffffff;
      if (oi != 0x7fffffff && (!mine || better_d(os, oi, sh_s[tid], sh_i[tid]))) { sh_s[tid] = os; sh_i[tid] = oi; }
    }
    __syncthreads();
  }
  const int best = sh_i[0];     // (always a code: the scanned sets contain the filter's own maximum, whose f^ passes thr)
  if (best != 0x7fffffff) write_result(p, row, best, tid);
  __syncthreads();
}

template <int MODE, int DIM, int GT, int NSI>
__device__ __forceinline__ void rerank_block(const RerankParams &p, const int vblock, const int nrows) {
  constexpr int GROUP = kRerankLanes;
  constexpr int RPW = 64 / GROUP;            // rows per wave
  constexpr int RPB = 4 * RPW;               // rows per block
  constexpr int CANDPAD = 3 * NSI * GROUP + 17;   // odd-ish stride: the row slots of a wave start in different LDS banks
  __shared__ int cand[RPB][CANDPAD];
  __shared__ float s_ops[RPB][3 * DIM + 1];
  __shared__ float s_zhat[RPB][DIM + 1];
  __shared__ int s_best[RPB];
  __shared__ unsigned long long s_scan_mask[RPB];   // record sets the in-block finish scans (0: the row was decided)
  __shared__ float s_scan_thr[RPB];
  __shared__ int s_scan_keep[RPB];
  __shared__ double sh_s[256];
  __shared__ int sh_i[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % GROUP, grp = lane / GROUP;
  const int slot = wave * RPW + grp;
  const long pos_raw = (long)vblock * RPB + slot;
  const bool live = pos_raw < nrows;
  const long row = live ? pos_raw : nrows - 1;                       // dead groups mirror the last row, write nothing
  const int gshift = grp * GROUP;
  const unsigned long long glow = (1ull << GROUP) - 1ull;
  auto group_bits = [&](bool c) { return (__ballot(c) >> gshift) & glow; };

#ifdef GQHIP_CLOCK_STAMPS
  unsigned long long st[8];
  int nst = 0;
#define GQ_RR_STAMP() do { if (nst < 8) st[nst++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GQ_RR_STAMP() do { } while (0)
#endif
  GQ_RR_STAMP();     // 0: start
  // ---- everything the row needs, issued together ---------------------------
  const float N1f = wave_absmax(p.hdr->absmax_part, lane);
  const float R2f = p.rowaux ? wave_absmax(p.hdr->r2_part, lane) : 0.0f;   // max_j |cb_j|^2 (F16 bound)
  double rs[4];
  {
    const double *q = p.rowsum + row * 4;
    rs[0] = q[0]; rs[1] = q[1]; rs[2] = q[2]; rs[3] = q[3];
  }
  const float NEG_INF = -__builtin_inff();
  Rec r[NSI];
#pragma unroll
  for (int k = 0; k < NSI; ++k) {
    r[k] = empty_rec();
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

  GQ_RR_STAMP();     // 1: records and sums have arrived (fmax reduced)
  double margin = 2.5 * ((double)p.ef_coeff * u * T + Er) + 1e-30;      // around the filter's row maximum
  if (p.rowaux) {
    // fp16 main-product filter: the data-dependent bound (f16_bound above); 1.25 (Ea + Eb + 2 E_r) keeps the same 25 % slack
    float aux[8];
    const f32x4 *q = reinterpret_cast<const f32x4 *>(p.rowaux + row * 8);
    const f32x4 q0 = q[0], q1 = q[1];
    aux[0] = q0.x; aux[1] = q0.y; aux[2] = q0.z; aux[3] = q0.w; aux[4] = q1.x; aux[5] = q1.y; aux[6] = q1.z; aux[7] = q1.w;
    const double R2 = (double)R2f * (1.0 + 64.0 * u);       // the fp32 sum of squares of up to 64 dims, rounded up
    margin = 1.25 * (f16_bound(aux, T, Er, N1, R2, (double)fmax, DIM, (double)p.ef_coeff * u) + 2.0 * Er) + 1e-30;
  }
  bool bad = !(N1 == N1) || N1 > 1e18 || !(T < 1e30) || !(G < 1e30) || !(margin < 1e30);
  if (p.n1_limit > 0.f && !(N1f <= p.n1_limit && N1f >= p.n1_min)) bad = true;
  bad = bad || !(fmax == fmax) || !(fmax > NEG_INF) || !(fmax < __builtin_inff());

  const double thr = (double)fmax - margin;
  const unsigned long long lt = (1ull << sub) - 1ull;
  int total = 0;
  bool fourth = false;
  unsigned long long sets_in = 0ull;          // record sets with any group within the margin
#pragma unroll
  for (int k = 0; k < NSI; ++k) {
    // m2..m4 as the record holds them: m1 - gap, never below the filter's value (gq_common.h:Rec)
    const bool c1 = (double)r[k].m1 >= thr, c2 = rec_value(r[k].m1, r[k].d2) >= thr, c3 = rec_value(r[k].m1, r[k].d3) >= thr;
    const bool c4 = rec_value(r[k].m1, r[k].d4) >= thr;
    const unsigned long long b1 = group_bits(c1), b2 = group_bits(c2), b3 = group_bits(c3);
    fourth = fourth || group_bits(c4) != 0ull;   // a fourth group of some set could matter: the candidates are incomplete
    sets_in |= b1 << (k * GROUP);
    const int n1 = __popcll(b1), n2 = __popcll(b2);
    // ids are relative to the set's split: + 2 * (first tile of the split / GT)
    const int sset = k * GROUP + sub;
    const int base2 = 2 * (((p.rec_halves == 2 ? sset >> 1 : sset) * p.tiles_per_split) / GT);
    if (c1) cand[slot][total + __popcll(b1 & lt)] = base2 + (int)r[k].id1;
    if (c2) cand[slot][total + n1 + __popcll(b2 & lt)] = base2 + (int)r[k].id2;
    if (c3) cand[slot][total + n1 + n2 + __popcll(b3 & lt)] = base2 + (int)r[k].id3;
    total += n1 + n2 + __popcll(b3);
  }
  const bool undecided = bad || fourth;       // group-uniform
  if (sub == 0) {
    const unsigned long long all_sets = p.nsplit >= 64 ? ~0ull : ((1ull << p.nsplit) - 1ull);
    s_scan_mask[slot] = (live && undecided) ? (bad ? all_sets : sets_in) : 0ull;
    // j* has f^(j*) >= F~ - (E_f | Ea) - 2 E_r - E32 (head of this file); margin >= (E_f | Ea) + 2 E_r and margin32 >= E32
    s_scan_thr[slot] = (float)((double)fmax - margin - (double)margin32) - 1.1920929e-07f * __builtin_fabsf(fmax);
    s_scan_keep[slot] = bad ? 1 : 0;
    if (live && undecided) atomicAdd(&p.hdr->fb_count, 1);
  }
  if (undecided) total = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  GQ_RR_STAMP();     // 2: bound, margins, candidate lists
  // this lane's code t (0 .. GT-1) of candidate id: lane half (id & 1) of tile (id >> 1) * GT + t
  const int code_in_tile = (sub & 3) + 8 * (sub >> 2);
  auto code_of = [&](int id, int t) { return ((id >> 1) * GT + t) * kTileCodes + code_in_tile + 4 * (id & 1); };
  // f^ = sum_i (A_i n_i + B_i) n_i: two FMAs per dimension (round 4: Horner form -- one rounding of A n + B, relative to
  // |A||n| + |B|, and one of the running sum per dimension: <= (dim + 1) u T, inside the E32 = (2 dim + 4) u T charged for it)
  auto expansion = [&](const float (&n)[DIM]) {
    float f = 0.0f;
#pragma unroll
    for (int i = 0; i < DIM; ++i) f = __builtin_fmaf(__builtin_fmaf(cA[i], n[i], cB[i]), n[i], f);
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
  GQ_RR_STAMP();     // 3: pass 1 (gathers + fp32 expansions)
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
  const bool in_window = codeb >= 0 && (keep_all || !(fb < thr32));
  const bool second_in_window = codeb >= 0 && (keep_all || !(fsecond < thr32)) && fsecond > NEG_INF;
  // ONE code of the row inside the window (19 rows of 20): every other candidate code's reference score is provably below this
  // code's (that is what the window means), so it is the arg-max whatever the two scores are -- the reference's arithmetic, a few
  // hundred instructions that the whole wave would sit through, is only needed to decide between several (round 5; the same
  // shortcut as gq_grid.h's)
  const bool only_one = !keep_all && __popcll(group_bits(in_window)) == 1 && group_bits(second_in_window) == 0ull;
  if (only_one) {
    if (in_window) { best_s = 0.0; best_i = codeb; have = true; }
  } else if (in_window) {
    exact(nb, codeb);
  }
  // a second code of the SAME lane inside the window (rare: a near-tie within one lane's few codes): go through
  // this lane's codes again
  if (__any(second_in_window)) {
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
  GQ_RR_STAMP();     // 4: pass 2 (the reference's arithmetic) + the group reduction
  const bool decided = live && !undecided;
  if (p.stats && decided && sub == 0) atomicAdd(&p.hdr->reranked, (unsigned long long)total);
  // the block's RPB consecutive rows leave as contiguous runs (BCHW: along l per channel)
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
  GQ_RR_STAMP();     // 5: block barrier
  const long row0 = (long)vblock * RPB;
  if (threadIdx.x < RPB) {
    const int b = s_best[threadIdx.x];
    if (b >= 0) p.idx[out_idx_offset(p.omap, row0 + threadIdx.x)] = (int64_t)b;
  }
  if (p.zhat) {
    for (int j = threadIdx.x; j < RPB * DIM; j += 256) {
      int lr, g;
      if (p.omap.mode == 1) { lr = j % RPB; g = j / RPB; } else { lr = j / DIM; g = j % DIM; }
      if (s_best[lr] >= 0) {
        const long o = out_zhat_offset(p.omap, row0 + lr, g, DIM);
        p.zhat[o] = ste_mix(p.hdr, o, s_zhat[lr][g]);
      }
    }
  }
#ifdef GQHIP_CLOCK_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GQ_RR_STAMP();     // 6: results stored
  if (threadIdx.x == 0 && vblock < 1024 && p.dbg) {
    unsigned long long *o = reinterpret_cast<unsigned long long *>(p.dbg) + 8 * vblock;
    for (int k = 0; k < 8; ++k) o[k] = k < nst ? st[k] : 0ull;
  }
#endif
  // ---- the rows the candidates could not decide: finished here, by the whole block, one after the other ----
  for (int sl = 0; sl < RPB; ++sl) {
    const unsigned long long m = s_scan_mask[sl];     // block-uniform (written before the barrier above)
    if (m != 0ull) finish_row_by_scan<MODE, DIM>(p, row0 + sl, m, s_scan_thr[sl], s_scan_keep[sl] != 0, s_ops[sl], sh_s, sh_i);
  }
}

// (256, 4): at most 128 VGPRs, four blocks per CU -- the kernel lives on memory-level parallelism across waves
// (dim 32 needs 2 x 32 coefficient registers alone: two blocks per CU there)
template <int MODE, int DIM, int GT, int NSI>
__global__ __launch_bounds__(256, DIM >= 32 ? 2 : 4) void gq_rerank_kernel(const RerankParams p) {
  // gq_quantize_z_gauss_f32 launches ONE block more than the rows need: block 0 then runs GQ2's statistics (gq_gauss.h) beside the
  // re-rank's blocks -- its input, the per-row KL bits, was left by the call's first launch -- so that call has no fourth launch
  int vblock = (int)blockIdx.x;
  if constexpr (MODE == kModeGQ) {
    const int extra = (int)gridDim.x - (p.rows + 15) / 16;          // 0 or 1
    if (extra > 0) {
      if (vblock == 0) {
        if (p.hdr->gs.rows > 0) gauss_stats_block(p.hdr->gs);
        return;
      }
      vblock -= extra;
    }
  }
  rerank_block<MODE, DIM, GT, NSI>(p, vblock, p.rows);
}

// Exhaustive exact arg-max of rows list[first], list[first + stride], ... (list == NULL: the rows themselves): one
// block per row, every code scored in the reference's operation order.  Depends on nothing but the row operands and
// the codebook (dims without an MFMA filter run on it).
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
