// gq_grid.h -- the fused arg-max at dim 4 (round 5; templated for dim 8): an exact, PRUNED search over a spatial index of the
// codebook instead of the dense rows x codes product.
//
// Why (profiles/r04/pmc_filter_gq_1.00_dim4_round3_kernels.txt): at K_eff = 2 dim = 8 .. 16 the dense MFMA filter is
// bound by the VALU fold of its own outputs (>= 8 v_max3 per 32-cycle MFMA), not by the matrix cores -- no schedule fixes that.
// But 65 536 codes are DENSE in 4 dimensions: the score  f(r, j) = sum_i A_ri n_ji^2 + B_ri n_ji  is a quadratic per coordinate,
// so its maximum over an axis-aligned BOX of codes is a closed form, and almost every box is far below the row's best code.  The
// codebook is therefore sorted once into a tree of boxes -- 16 -> 256 -> 1024 leaves -> 4096 sub-leaves of ~16 codes, TIGHT bounding
// boxes of their actual members (the quantile cells themselves are useless in the near-linear regime of sigma ~ 1:
// profiles/r05/grid_prune_study.txt).  Three levels of boxes and the leaves' code ranges sit in LDS (46 KiB at dim 4); the sub-leaf
// boxes and ranges are fetched from the cache (L2) for the leaves a row lists.  A row first spends ONE round trip on a good F
// (greedy descent to a leaf, or a beam to two leaves when its score is not concave), lists the leaves, then the sub-leaves, whose
// upper bound is not below  F - margin, and visits those six per round trip, F tightening.  The kernel is bound by VALU issue
// (~1500 instructions per four rows, three quarters of them bookkeeping of the 16-lane groups) with 5-7 dependent round trips of
// ~1.5 us on top; every variant that was built and timed on the way: profiles/r05/grid_search_variants.txt.
//
// Exactness is unchanged -- the reference's own arithmetic still decides (gq_common.h:ref_term / gq_rerank.h:ref_score_lds):
//   * f^(j) = the fp32 FMA expansion of f (pass 1 of the re-rank: |f^ - f| <= E32 = (2 dim + 4) u T);  E_r bounds the reference
//     score's own rounding (gq_rerank.h:row_bound);  margin32 = 2.5 (E32 + E_r).
//   * The reference's arg-max j* has f(j*) >= f(j^) - 2 E_r for every code j^, in particular for the one that set F:
//     f(j*) >= F - E32 - 2 E_r.  Every box that contains j* has a true upper bound >= f(j*); its computed bound U^ is below the true
//     one by at most (dim + 1) u T <= E32 (one FMA pair per axis evaluated at a point of the box, dim - 1 additions), so
//     U^ >= F - 2 E32 - 2 E_r >= F - margin32: the box is visited, at either level, whatever F was at the time (F only grows).
//   * Inside a sub-leaf, j* has f^(j*) >= F - 2 E32 - 2 E_r >= F - margin32 as well, so it receives the reference's arithmetic; the
//     winner among everything that did is taken in torch.argmax order (score, then lowest index; NaN first).
//   * Rows the search does not decide go to a list and are finished by gq_grid_finish_kernel (the launch after the search; it exits
//     at once when the list is empty), a block per row: a row whose leaf list does not fit even after one rebuild (a long tail of
//     loose boxes: nearly linear scores reach into the codebook's tails, where leaves are large) over its candidate leaves -- pass 1
//     the largest expansion, pass 2 the reference's arithmetic within the margin --; rows with a non-finite operand or bound, or
//     behind a non-finite codebook, over ALL codes, everything exact (the re-rank's exhaustive semantics).
//
// The index lives in a caller-owned, persistent buffer (the "codebook cache": gqhip_cb_cache_bytes), is validated on EVERY call
// against a content hash of the codebook the caller passes (the first launch hashes it anyway while it reduces max|cb|), and is
// rebuilt in-stream by a one-block kernel when the hash -- or the buffer's own stamp -- does not match: a codebook edited in place
// by any route is seen by the very next call, a fresh or clobbered buffer costs one rebuild.
#pragma once
#include <type_traits>

#include "gq_common.h"
#include "gq_rerank.h"

namespace gqhip {

constexpr unsigned kGridMagic = 0x47514733u;
constexpr int kGridLanes = 16;                     // lanes per row (one DPP row)
constexpr int kGridL1 = 16, kGridL2 = 256, kGridLeaves = 1024;    // 16 nodes x 16 nodes x 4 leaves
constexpr int kGridFan2 = kGridLeaves / kGridL1;                    // leaves under an L1 node
constexpr int kGridSubPerLeaf = 4, kGridSubs = kGridLeaves * kGridSubPerLeaf;   // sub-leaves (boxes in the cache, not in LDS): 4096
constexpr int kGridBuildThreads = 1024;
constexpr int kGridLeafCap = 128;                  // leaf-list capacity of the search (16-bit ids; 48 / 64 / 96 were measured: grid_search_variants.txt)
constexpr int kGridSubCap = 96;                    // sub-leaf list capacity (a longer list: its best part is visited, then ONE rebuild)

struct GridHdr {                                   // first 4 KiB of the codebook cache
  unsigned magic;
  int n, dim;
  int stale;                                       // set by the first launch when the codebook's hash differs; cleared by the builder
  int max_sub;                                     // the builder's census: codes in the fullest sub-leaf (> 255: the search hands every row that lists
                                                   // it to gq_grid_finish_kernel -- a degenerate book; hosts read it through gqhip_cb_cache_degenerate)
  int pad0[11];
  float thr[8][8];                                 // per axis: the (cells - 1) ascending thresholds of its coordinate, rest +inf
  unsigned long long blk_sum[kAbsmaxParts];        // hash of the codebook slice of prep's code block k when the index was built
  int pad1[432];
};
static_assert(sizeof(GridHdr) == 4096, "cache header is 4 KiB");

struct GridLayout {
  int64_t hdr, box1, box2, box3, start, sstart, sbox, scb, sidx, total;   // box1 | box2 | box3 | start are contiguous: one linear copy into LDS
};
__host__ __device__ inline int64_t grid_align(int64_t v) { return (v + 255) / 256 * 256; }
__host__ __device__ inline GridLayout grid_layout(int64_t n, int64_t dim) {
  GridLayout g{};
  int64_t off = 0;
  g.hdr = off;   off += (int64_t)sizeof(GridHdr);
  g.box1 = off;  off += kGridL1 * 2 * dim * 4;
  g.box2 = off;  off += kGridL2 * 2 * dim * 4;
  g.box3 = off;  off += (int64_t)kGridLeaves * 2 * dim * 4;
  g.start = off; off += grid_align((kGridLeaves + 16) * 4);
  g.sstart = off; off += grid_align((kGridSubs + 16) * 4);
  g.sbox = off;  off += (int64_t)kGridSubs * 2 * dim * 4;
  g.scb = off;   off += grid_align(n * dim * 4);
  g.sidx = off;  off += grid_align(n * 4);
  g.total = off;
  return g;
}

// cells per axis: dim 4: 8 x 8 x 8 x 8; dim 8: 4 x 4 x 4 x 4 x 2 x 2 x 2 x 2 -- 4096 sub-leaves either way, four to a leaf, and for a
// scrambled-Sobol codebook of 65 536 points mapped through the normal quantile (pit/quantization/gaussian.py:15-19) exactly 16
// codes in each (a (t, m, s)-net: every elementary interval holds its share).  Sub-leaf id = ((l1 * 16 + l2) * 4 + l3) * 4 + l4
// (boxes: 16 -> 256 -> 1024 leaves -> 4096 sub-leaves): l1 = the top bit of the coordinates of axes 0..3; dim 4: l2 = their second
// bit, l3 = the low bit of axes 0, 1, l4 = the low bit of axes 2, 3; dim 8: l2 = the low bit of axes 0, 1 and the bits of axes 4, 5,
// l3 = the bits of axes 6, 7, l4 = the low bit of axes 2, 3.
template <int DIM> __device__ __forceinline__ int grid_cells_of_axis(int i) { return DIM == 4 ? 8 : (i < 4 ? 4 : 2); }

template <int DIM>
__device__ __forceinline__ int grid_sub_of(const float (&x)[DIM], const float (*thr)[8]) {
  int c[DIM];
#pragma unroll
  for (int i = 0; i < DIM; ++i) {
    int k = 0;
#pragma unroll
    for (int t = 0; t < 7; ++t)
      if (t < grid_cells_of_axis<DIM>(i) - 1) k += x[i] >= thr[i][t] ? 1 : 0;       // NaN: 0
    c[i] = k;
  }
  int l1, l2, l3, l4;
  if constexpr (DIM == 4) {
    l1 = (c[0] >> 2) | ((c[1] >> 2) << 1) | ((c[2] >> 2) << 2) | ((c[3] >> 2) << 3);
    l2 = ((c[0] >> 1) & 1) | (((c[1] >> 1) & 1) << 1) | (((c[2] >> 1) & 1) << 2) | (((c[3] >> 1) & 1) << 3);
    l3 = (c[0] & 1) | ((c[1] & 1) << 1);
    l4 = (c[2] & 1) | ((c[3] & 1) << 1);
  } else {
    l1 = (c[0] >> 1) | ((c[1] >> 1) << 1) | ((c[2] >> 1) << 2) | ((c[3] >> 1) << 3);
    l2 = (c[0] & 1) | ((c[1] & 1) << 1) | (c[4] << 2) | (c[5] << 3);
    l3 = c[6] | (c[7] << 1);
    l4 = (c[2] & 1) | ((c[3] & 1) << 1);
  }
  return (((l1 * 16 + l2) * 4 + l3) * 4) + l4;
}

// ---------------------------------------------------------------------------------------------------- builder (one block)
struct GridBuildParams {
  const float *cb;
  char *cache;            // the codebook cache (grid_layout)
  const WsHeader *hdr;    // this call's workspace header: cbsum[] = the hashes the first launch just computed
  int n;
};

template <int DIM>
__global__ __launch_bounds__(kGridBuildThreads) void gq_grid_build_kernel(const GridBuildParams p) {
  GridHdr *gh = reinterpret_cast<GridHdr *>(p.cache);
  if (gh->magic == kGridMagic && gh->n == p.n && gh->dim == DIM && gh->stale == 0) return;   // (block-uniform) the index is current
  const GridLayout L = grid_layout(p.n, DIM);
  float *box1 = reinterpret_cast<float *>(p.cache + L.box1), *box2 = reinterpret_cast<float *>(p.cache + L.box2);
  float *box3 = reinterpret_cast<float *>(p.cache + L.box3);
  int *start = reinterpret_cast<int *>(p.cache + L.start);
  int *sstart = reinterpret_cast<int *>(p.cache + L.sstart);
  float *sbox = reinterpret_cast<float *>(p.cache + L.sbox);
  float *scb = reinterpret_cast<float *>(p.cache + L.scb);
  int *sidx = reinterpret_cast<int *>(p.cache + L.sidx);
  constexpr int NT = kGridBuildThreads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ double s_red[NT / 64][2 * DIM];
  __shared__ float s_thr[8][8];
  __shared__ int s_cnt[kGridSubs];
  __shared__ int s_wsum[NT / 64];

  // ---- 1. per-axis mean / deviation -> thresholds at the normal quantiles (any thresholds are valid; these balance a Gaussian book)
  double acc[2 * DIM];
#pragma unroll
  for (int i = 0; i < 2 * DIM; ++i) acc[i] = 0.0;
  for (int j = tid; j < p.n; j += NT) {
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      const double v = (double)p.cb[(long)j * DIM + i];
      acc[i] += v;
      acc[DIM + i] += v * v;
    }
  }
#pragma unroll
  for (int i = 0; i < 2 * DIM; ++i) {
    double v = acc[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) s_red[wave][i] = v;
  }
  __syncthreads();
  if (tid < 64) {
    const int i = tid >> 3, t = tid & 7;
    float th = __builtin_inff();
    if (i < DIM) {
      double s = 0.0, q = 0.0;
      for (int w = 0; w < NT / 64; ++w) { s += s_red[w][i]; q += s_red[w][DIM + i]; }
      const double mean = s / p.n;
      const double var = q / p.n - mean * mean;
      const double sd = sqrt(var > 0.0 ? var : 0.0);
      const int cells = grid_cells_of_axis<DIM>(i);
      // N(0,1) quantiles k / cells
      const double q8[7] = {-1.1503493803760079, -0.6744897501960817, -0.3186393639643752, 0.0, 0.3186393639643752,
                            0.6744897501960817, 1.1503493803760079};
      const double q4[3] = {-0.6744897501960817, 0.0, 0.6744897501960817};
      if (t < cells - 1) th = (float)(mean + sd * (cells == 8 ? q8[t] : (cells == 4 ? q4[t] : 0.0)));
    }
    s_thr[i][t] = th;
    gh->thr[i][t] = th;
  }
  for (int c = tid; c < kGridSubs; c += NT) s_cnt[c] = 0;
  __syncthreads();

  // ---- 2. sub-leaf histogram
  auto sub_of_code = [&](int j) {
    float x[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i) x[i] = p.cb[(long)j * DIM + i];
    return grid_sub_of<DIM>(x, s_thr);
  };
  for (int j = tid; j < p.n; j += NT) atomicAdd(&s_cnt[sub_of_code(j)], 1);
  __syncthreads();
  {   // census: the fullest sub-leaf (one wave-reduced atomicMax per wave into LDS)
    __shared__ int s_maxsub;
    if (tid == 0) s_maxsub = 0;
    __syncthreads();
    int mx = 0;
    for (int c = tid; c < kGridSubs; c += NT) mx = max(mx, s_cnt[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    if (lane == 0) atomicMax(&s_maxsub, mx);
    __syncthreads();
    if (tid == 0) gh->max_sub = s_maxsub;
  }

  // ---- 3. exclusive scan of the sub-leaf counters (one leaf = four of them per thread) -> sstart[], start[], cursors in s_cnt
  {
    constexpr int PER = kGridSubs / NT;
    static_assert(kGridSubs % NT == 0 && PER == kGridSubPerLeaf && NT == kGridLeaves, "a thread scans the sub-leaves of one leaf");
    int loc[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { loc[k] = s_cnt[tid * PER + k]; sum += loc[k]; }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(inc, o);
      if (lane >= o) inc += v;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
    int run = base + inc - sum;
    start[tid] = run;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      sstart[tid * PER + k] = run;
      s_cnt[tid * PER + k] = run;
      run += loc[k];
    }
    if (tid == NT - 1) { start[kGridLeaves] = run; sstart[kGridSubs] = run; }
  }
  __syncthreads();

  // ---- 4. scatter the codes into leaf order (order inside a leaf: arbitrary -- the search's result does not depend on it)
  for (int j = tid; j < p.n; j += NT) {
    float x[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i) x[i] = p.cb[(long)j * DIM + i];
    const int pos = atomicAdd(&s_cnt[grid_sub_of<DIM>(x, s_thr)], 1);
#pragma unroll
    for (int i = 0; i < DIM; ++i) scb[(long)pos * DIM + i] = x[i];
    sidx[pos] = j;
  }
  __threadfence();
  __syncthreads();

  // ---- 5. tight bounding boxes, bottom-up ([lo | hi]; an empty node: lo = +inf, hi = -inf)
  const float INF = __builtin_inff();
  for (int c = tid; c < kGridSubs; c += NT) {
    float lo[DIM], hi[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i) { lo[i] = INF; hi[i] = -INF; }
    const int s = sstart[c], e = sstart[c + 1];
    for (int j = s; j < e; ++j) {
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
        const float v = scb[(long)j * DIM + i];
        lo[i] = __builtin_fminf(lo[i], v);
        hi[i] = __builtin_fmaxf(hi[i], v);
      }
    }
#pragma unroll
    for (int i = 0; i < DIM; ++i) { sbox[(long)c * 2 * DIM + i] = lo[i]; sbox[(long)c * 2 * DIM + DIM + i] = hi[i]; }
  }
  __threadfence();
  __syncthreads();
  auto merge = [&](const float *child, float *parent, int node, int fan) {
    float lo[DIM], hi[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i) { lo[i] = INF; hi[i] = -INF; }
    for (int k = 0; k < fan; ++k) {
      const float *b = child + (long)(node * fan + k) * 2 * DIM;
#pragma unroll
      for (int i = 0; i < DIM; ++i) { lo[i] = __builtin_fminf(lo[i], b[i]); hi[i] = __builtin_fmaxf(hi[i], b[DIM + i]); }
    }
#pragma unroll
    for (int i = 0; i < DIM; ++i) { parent[(long)node * 2 * DIM + i] = lo[i]; parent[(long)node * 2 * DIM + DIM + i] = hi[i]; }
  };
  if (tid < kGridLeaves) merge(sbox, box3, tid, kGridSubPerLeaf);
  __threadfence();
  __syncthreads();
  if (tid < kGridL2) merge(box3, box2, tid, kGridLeaves / kGridL2);
  __threadfence();
  __syncthreads();
  if (tid < kGridL1) merge(box2, box1, tid, kGridL2 / kGridL1);
  if (tid < 15) { start[kGridLeaves + 1 + tid] = p.n; sstart[kGridSubs + 1 + tid] = p.n; }   // padding (start: copied to LDS in 16-byte pieces)
  // ---- 6. stamp: the hashes this call's first launch computed of the very codebook that was just indexed
  if (tid < kAbsmaxParts) gh->blk_sum[tid] = p.hdr->cbsum[tid];
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    gh->n = p.n;
    gh->dim = DIM;
    gh->stale = 0;
    gh->magic = kGridMagic;
  }
}

// ---------------------------------------------------------------------------------------------------- the search
struct GridParams {
  const float *mu, *sd, *lsd;      // [rows, dim] (VQ: z in mu)
  const double *rowsum;            // [rows, 4]
  const float *coef;               // [rows, 2, dim]  A | B
  const float *cb;                 // [n, dim] the caller's codebook (results, the block-wide scan)
  const char *cache;               // the codebook cache
  int64_t *idx;
  float *zhat;                     // may be NULL
  WsHeader *hdr;
  int rows, n;
  float beta;
  int leaf_cap;                    // leaves' worth of codes a row may visit before a truncated list sends it to gq_grid_finish_kernel
  int inwave_cap;                  // listed leaves a row's 16 lanes go through themselves (diagnostics: default = the list's capacity)
  int stats;
  int *und_row;                    // [rows] undecided rows, appended by the search ((row << 1) | keep_all), finished by gq_grid_finish_kernel
  float *und_thr, *und_margin;     // [rows] their threshold so far and margin32
  int abl;                         // diagnostic builds only (GQHIP_ABL): 1 = no search, 2 = greedy descent only (wrong results, timing)
  OutMap omap;
};

// max over the box [lo, hi] of sum_i A_i v_i^2 + B_i v_i: per axis at the clamped vertex (A < 0) or at the better endpoint.
// `concave` (wave-uniform): every axis of every row of the wave has A < 0 -- the trained operating point, sigma < 1 / sqrt(beta) --
// and the endpoint arithmetic is skipped (3 instead of 8 instructions per axis; the search is VALU-issue bound).
template <int DIM>
__device__ __forceinline__ void grid_load_box(const float *box, float (&lo)[DIM], float (&hi)[DIM]) {
  const f32x4 *q = reinterpret_cast<const f32x4 *>(box);
#pragma unroll
  for (int k = 0; k < DIM / 4; ++k) {
    const f32x4 a = q[k], b = q[DIM / 4 + k];
    lo[4 * k] = a.x; lo[4 * k + 1] = a.y; lo[4 * k + 2] = a.z; lo[4 * k + 3] = a.w;
    hi[4 * k] = b.x; hi[4 * k + 1] = b.y; hi[4 * k + 2] = b.z; hi[4 * k + 3] = b.w;
  }
}
template <int DIM>
__device__ __forceinline__ float grid_box_ub_v(const float (&A)[DIM], const float (&B)[DIM], const float (&M)[DIM], const float (&lo)[DIM],
                                               const float (&hi)[DIM], bool concave) {
  float u = 0.0f;
  if (concave) {
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      const float v = __builtin_amdgcn_fmed3f(M[i], lo[i], hi[i]);
      u = __builtin_fmaf(__builtin_fmaf(A[i], v, B[i]), v, u);
    }
  } else {
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      const float vin = __builtin_amdgcn_fmed3f(M[i], lo[i], hi[i]);
      const float e = __builtin_fmaf(A[i], hi[i] + lo[i], B[i]) > 0.0f ? hi[i] : lo[i];
      const float v = A[i] < 0.0f ? vin : e;
      u = __builtin_fmaf(__builtin_fmaf(A[i], v, B[i]), v, u);
    }
  }
  return lo[0] <= hi[0] ? u : -__builtin_inff();     // empty node (or a NaN box: never visited; such books are scanned)
}
template <int DIM>
__device__ __forceinline__ float grid_box_ub(const float (&A)[DIM], const float (&B)[DIM], const float (&M)[DIM], const float *box,
                                             bool concave) {
  float lo[DIM], hi[DIM];
  grid_load_box<DIM>(box, lo, hi);
  return grid_box_ub_v<DIM>(A, B, M, lo, hi, concave);
}

// Reductions over the 16 lanes of a DPP row (= one search group) without an LDS round trip: quad_perm [1,0,3,2], [2,3,0,1],
// row_half_mirror, row_mirror -- afterwards every lane of the row holds the result.
// (Written out: from the builtins the compiler makes v_mov_b32_dpp + a canonicalising v_max + the v_max itself per step, twelve
//  VALU instructions per reduction of a kernel that is VALU-issue bound; the fused form is four.  s_nop 1: the two wait states a
//  DPP read needs after a VALU write of its source.  Every lane of a row is active or none is, and no source lane is invalid for
//  these four patterns.)
__device__ __forceinline__ float row16_max(float v) {
  asm("s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
      : "+v"(v));
  return v;
}
// max of two values that are never NaN here (rows with non-finite operands or bounds do not get this far): one v_med3_f32 instead
// of the canonicalise-then-max pair that fmaxf becomes under IEEE mode
__device__ __forceinline__ float fmax_nn(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }

constexpr int kGridThreads = 512;      // 8 waves x 4 rows: 32 rows per block pass share one LDS copy of the tree; two blocks per CU

// 16 lanes per row, 4 rows per wave; a block's eight waves draw row sets from a counter in LDS (no block barrier inside the loop: a
// row's search is a chain of dependent steps of very uneven length; waiting for the slowest of 32 rows at every pass cost a third of
// the kernel, a static split of the row sets ended at twice the mean: profiles/r05/grid_search_variants.txt).  Three levels of
// boxes and the code ranges of the leaves sit in LDS.  A row:
//   (1) ONE round trip for a first F: concave rows -- greedy descent to one leaf, its 64 codes; otherwise a beam to two leaves;
//   (2) the leaves whose bound (and whose L1 / L2 nodes' bounds) is within the margin of F -> a list of up to 128 (LDS / VALU only);
//   (3) their sub-leaves' boxes and code ranges from the cache, twelve leaves per round trip; those within the margin -> the
//       sub-list (bound, packed range);
//   (4) six sub-leaves per round trip (one code per lane and sub-leaf) until the best remaining bound is below the threshold (F and
//       thr tighten with every batch): fp32 expansions; every lane keeps its own best code and its runner-up value (the re-rank's
//       pass-1 scheme); lists that did not fit: their best part is visited, then ONE rebuild under the tighter threshold;
//   (5) with the FINAL threshold: a row with one code within the margin is decided (no reference arithmetic needed to name the
//       arg-max); otherwise the lanes whose best code is within the margin give it the reference's arithmetic -- ONE instance of
//       that long, divergent code per row -- and a lane with a SECOND code within the margin (a near-tie) sends the row through
//       everything it visited again with the exact score for everything within the margin.
// Undecided rows (non-finite operands or bounds, a non-finite codebook, lists that do not fit after one rebuild, a sub-leaf of more
// than 255 codes) go to the call's list and are finished by gq_grid_finish_kernel, the next launch.

template <int MODE, int DIM>
__device__ __forceinline__ void grid_scan_accumulate(const GridParams &p, const float (&cA)[DIM], const float (&cB)[DIM], const float *ops,
                                                     float thr, bool keep_all, int first, int stride, double &best_s, int &best_i,
                                                     bool &have) {
  constexpr int U = 8;
  for (int j0 = first; j0 < p.n; j0 += stride * U) {
    float n[U][DIM];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + stride * u;
      const f32x4 *q = reinterpret_cast<const f32x4 *>(p.cb + (long)(j < p.n ? j : p.n - 1) * DIM);
#pragma unroll
      for (int k = 0; k < DIM / 4; ++k) {
        const f32x4 v = q[k];
        n[u][4 * k] = v.x; n[u][4 * k + 1] = v.y; n[u][4 * k + 2] = v.z; n[u][4 * k + 3] = v.w;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + stride * u;
      float f = 0.0f;
#pragma unroll
      for (int i = 0; i < DIM; ++i) f = __builtin_fmaf(__builtin_fmaf(cA[i], n[u][i], cB[i]), n[u][i], f);
      if (j < p.n && (keep_all || !(f < thr))) {           // a NaN value passes
        double sc;
        if constexpr (MODE == kModeGQ) sc = (double)ref_score_lds<DIM>(n[u], ops, p.beta);
        else sc = vq_neg_dist(n[u], ops, DIM);
        if (!have || better_d(sc, j, best_s, best_i)) { best_s = sc; best_i = j; have = true; }
      }
    }
  }
}

template <int DIM>
__device__ __forceinline__ void grid_load_coef(const GridParams &p, long row, float (&cA)[DIM], float (&cB)[DIM]) {
  const f32x4 *q = reinterpret_cast<const f32x4 *>(p.coef + row * 2 * DIM);
#pragma unroll
  for (int k = 0; k < DIM / 4; ++k) {
    const f32x4 a = q[k], b = q[DIM / 4 + k];
    cA[4 * k] = a.x; cA[4 * k + 1] = a.y; cA[4 * k + 2] = a.z; cA[4 * k + 3] = a.w;
    cB[4 * k] = b.x; cB[4 * k + 1] = b.y; cB[4 * k + 2] = b.z; cB[4 * k + 3] = b.w;
  }
}

__device__ __forceinline__ void grid_write_result(const GridParams &p, long row, int best, int who, int dim) {
  if (who == 0) p.idx[out_idx_offset(p.omap, row)] = (int64_t)best;
  if (p.zhat && who < dim) {
    const long o = out_zhat_offset(p.omap, row, who, dim);
    p.zhat[o] = ste_mix(p.hdr, o, p.cb[(long)best * dim + who]);
  }
}

template <int MODE, int DIM>
__global__ __launch_bounds__(kGridThreads, DIM >= 8 ? 2 : 4) void gq_grid_kernel(const GridParams p) {
  constexpr int GROUP = kGridLanes;
  constexpr int RPW = 64 / GROUP, RPB = (kGridThreads / 64) * RPW;
  constexpr int BOXF = 2 * DIM;                         // floats per box: [lo | hi]
  constexpr int TREE_F = (kGridL1 + kGridL2 + kGridLeaves) * BOXF;  // floats of box1 | box2 | box3
  constexpr int START_N = kGridLeaves + 16;
  constexpr int LPN = kGridLeaves / kGridL2;            // leaves per L2 node: 4
  constexpr int SPL = kGridSubPerLeaf;                  // sub-leaves per leaf: 4
  constexpr int CPL = 4;                                // codes per lane and LEAF in one round (64-code leaves: one round)
  constexpr int SUB_U = 6;                              // sub-leaves per round trip (16-code sub-leaves: one code per lane each)
  constexpr int FIRST_N = 2;                            // leaves of the first round trip of a non-concave row (a beam; concave rows: one)
  constexpr int CHUNKS = 3;                             // sub-box fetches in flight: 3 x (4 leaves x 4 sub-leaves = the 16 lanes)
  constexpr int EPL = kGridSubCap / GROUP;              // sub-list entries per lane: 6
  static_assert(SPL * (GROUP / SPL) == GROUP && kGridSubCap % GROUP == 0, "lane -> (leaf of the chunk, sub-leaf) mapping");
  __shared__ __attribute__((aligned(16))) float s_box[TREE_F];
  __shared__ __attribute__((aligned(16))) int s_start[START_N];
  __shared__ float s_ops[RPB][3 * DIM + 1];
  __shared__ unsigned short s_leaflist[RPB][kGridLeafCap];   // (leaf ids < 1024)
  __shared__ float s_subub[RPB][kGridSubCap];
  __shared__ unsigned s_subpk[RPB][kGridSubCap];        // start (24 bits: n <= 2^20) | length << 24 (longer than 255: the row is handed on)
  __shared__ unsigned s_sel[RPB][SUB_U];
  __shared__ int s_next;
  __shared__ unsigned s_stat_units, s_stat_exact;       // debug statistics of the block (one device atomic each at its end)
  const GridLayout L = grid_layout(p.n, DIM);
  const float *scb = reinterpret_cast<const float *>(p.cache + L.scb);
  const int *sidx = reinterpret_cast<const int *>(p.cache + L.sidx);
  const float *gsbox = reinterpret_cast<const float *>(p.cache + L.sbox);
  const int *gsstart = reinterpret_cast<const int *>(p.cache + L.sstart);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane % GROUP, grp = lane / GROUP, slot = wave * RPW + grp;
  const int gshift = grp * GROUP;
  auto group_bits = [&](bool c) { return (unsigned)((__builtin_amdgcn_ballot_w64(c) >> gshift) & 0xffffull); };
  auto group_max = [&](float v) { return row16_max(v); };
  auto wave_sync_lds = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  {   // three levels of boxes and the leaves' code ranges -> LDS: box1 | box2 | box3 | start are contiguous in the cache (grid_layout)
    const f32x4 *src = reinterpret_cast<const f32x4 *>(p.cache + L.box1);
    f32x4 *dst = reinterpret_cast<f32x4 *>(s_box);
    for (int k = tid; k < TREE_F / 4; k += kGridThreads) dst[k] = src[k];
    // (leaf ranges are clamped to [0, n] on their way into LDS: a cache body that was written behind the library's back -- the header's
    //  stamps cannot see that -- may then cost a wrong index, never an out-of-range read; gqhip.h, codebook cache)
    const i32x4 *src2 = reinterpret_cast<const i32x4 *>(p.cache + L.start);
    i32x4 *dst2 = reinterpret_cast<i32x4 *>(s_start);
    for (int k = tid; k < START_N / 4; k += kGridThreads) {
      i32x4 v = src2[k];
      v.x = min(max(v.x, 0), p.n); v.y = min(max(v.y, 0), p.n); v.z = min(max(v.z, 0), p.n); v.w = min(max(v.w, 0), p.n);
      dst2[k] = v;
    }
  }
  const float *s_box2 = s_box + kGridL1 * BOXF, *s_box3 = s_box + (kGridL1 + kGridL2) * BOXF;
  const float N1f = wave_absmax(p.hdr->absmax_part, lane);
  const float NEG_INF = -__builtin_inff();
  const int nquads = (p.rows + RPW - 1) / RPW;
  if (tid == 0) { s_next = 0; s_stat_units = 0u; s_stat_exact = 0u; }
  __syncthreads();

#ifdef GQHIP_CLOCK_STAMPS
  unsigned long long st_[8];
  int nst_ = 0;
  bool first_pass_ = true;
#define GQ_GRID_STAMP() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (nst_ < 8) st_[nst_++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GQ_GRID_STAMP() do { } while (0)
#endif
  // Work split: a block owns a contiguous share of the row sets (four rows each) and its eight waves draw from it through a counter
  // in LDS -- a row set takes 10 to 50 us depending on its rows (a flat score: long lists, a second round), and with a static
  // four sets per wave the slowest wave ended the kernel at twice the mean.  (A device-wide counter was tried first and serialised
  // the kernel on its atomics: 16 384 fetches of one address, 245 us instead of ~100; strided instead of contiguous shares: the
  // same within the lease-to-lease noise, and a few percent slower at the trained operating point.)
  const int per_block = (nquads + (int)gridDim.x - 1) / (int)gridDim.x;
  const int quad_lo = (int)blockIdx.x * per_block, quad_hi = min(nquads, quad_lo + per_block);
  for (;;) {
    int quad = 0;
    if (lane == 0) quad = quad_lo + atomicAdd(&s_next, 1);
    quad = __builtin_amdgcn_readfirstlane(quad);
    if (quad >= quad_hi) break;
    GQ_GRID_STAMP();   // 0: four rows start
    const long pos_raw = (long)quad * RPW + grp;
    const bool live = pos_raw < p.rows;
    const long row = live ? pos_raw : p.rows - 1;
    // ---- row operands
    float cA[DIM], cB[DIM], cM[DIM];
    grid_load_coef<DIM>(p, row, cA, cB);
    double rs[4];
    {
      const double *q = p.rowsum + row * 4;
      rs[0] = q[0]; rs[1] = q[1]; rs[2] = q[2]; rs[3] = q[3];
    }
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
    // the parabola's vertex, to within an ulp or two (v_rcp_f32): a bound evaluated a relative 1e-7 beside it is below the true one
    // by ~|A| mu'^2 1e-14, nothing against E32
    bool concave_row = true;
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      cM[i] = cA[i] < 0.0f ? -0.5f * cB[i] * __builtin_amdgcn_rcpf(cA[i]) : 0.0f;
      concave_row = concave_row && cA[i] < 0.0f;
    }
    const bool concave = __all(concave_row);
    const double u = 5.9604644775390625e-08;
    const double N1 = (double)N1f;
    double T, G;
    row_bound<MODE>(rs, N1, DIM, p.beta, T, G);
    const double Er = MODE == kModeGQ ? (DIM + 16.0) * u * G : 1e-12 * T;
    const float margin32 = (float)(2.5 * ((2.0 * DIM + 4.0) * u * T + Er) * 1.0000002 + 1e-30);
    const bool bad = !(N1 == N1) || N1 > 1e18 || !(T < 1e30) || !(G < 1e30) || !(margin32 < 1e30f);
    wave_sync_lds();

    auto expansion = [&](const float (&n)[DIM]) {
      float f = 0.0f;
#pragma unroll
      for (int i = 0; i < DIM; ++i) f = __builtin_fmaf(__builtin_fmaf(cA[i], n[i], cB[i]), n[i], f);
      return f;
    };
    float F = NEG_INF;
    // thr: what a box / a code must reach (rounded down: the subtraction and F's own last bit)
    auto thr_of = [&](float Fv) { return Fv > NEG_INF ? (Fv - margin32) - 2.4e-7f * __builtin_fabsf(Fv) : NEG_INF; };
    float thr = NEG_INF;
    float fb = NEG_INF, fsecond = NEG_INF;         // this lane's best code (value, sorted position) and its runner-up value
    int jb = -1;
    double best_s = 0.0;
    int best_i = 0x7fffffff, best_j = 0;           // the winner so far: original code id (ties: the lower one wins), position in the sorted codebook
    bool have = false;
    int units = 0, exact_n = 0;                    // units: visited sub-leaves (a whole leaf counts SPL)
    bool overflow = false;
    auto load_code = [&](int j, float (&n)[DIM]) {
      // (a 32-bit byte offset from a uniform base: one VALU instruction per address instead of a 64-bit multiply-add; n <= 2^20)
      const f32x4 *q = reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(scb) + (unsigned)j * (unsigned)(DIM * 4));
#pragma unroll
      for (int k = 0; k < DIM / 4; ++k) {
        const f32x4 v = q[k];
        n[4 * k] = v.x; n[4 * k + 1] = v.y; n[4 * k + 2] = v.z; n[4 * k + 3] = v.w;
      }
    };
    auto exact_code = [&](int j) {
      float nn[DIM];
      load_code(j, nn);
      const int code = (int)min((unsigned)sidx[j], (unsigned)(p.n - 1));
      double sc;
      if constexpr (MODE == kModeGQ) sc = (double)ref_score_lds<DIM>(nn, ops, p.beta);
      else sc = vq_neg_dist(nn, ops, DIM);
      if (!have || better_d(sc, code, best_s, best_i)) { best_s = sc; best_i = code; best_j = j; have = true; }
      ++exact_n;
    };
    // NR code ranges [s, e) of the sorted codebook at once: lane `sub` takes codes sub, sub + 16, ... (CP of them per round) of each
    // (a range's codes are contiguous; all CP x NR loads in flight together); longer ranges: further rounds.
    // EXACT = false: F and thr follow, the lane's best code and runner-up value are kept.  EXACT = true (the rare second pass of a
    // near-tie): every code with f^ >= thr receives the reference's arithmetic.
    auto visit = [&](auto nr_tag, auto cp_tag, auto exact_tag, const int (&s)[decltype(nr_tag)::value],
                     const int (&e)[decltype(nr_tag)::value]) {
      constexpr int NR = decltype(nr_tag)::value, CP = decltype(cp_tag)::value;
      constexpr bool EXACT = decltype(exact_tag)::value;
      int longest = 0;
#pragma unroll
      for (int t = 0; t < NR; ++t) longest = max(longest, e[t] - s[t]);
      for (int off = 0; off < longest; off += GROUP * CP) {          // (group-uniform trip count: one round for full ranges)
        float f[NR][CP];
        {
          float n[NR][CP][DIM];
#pragma unroll
          for (int t = 0; t < NR; ++t)
#pragma unroll
            for (int c = 0; c < CP; ++c) {
              const int j = s[t] + off + sub + GROUP * c;
              load_code(j < e[t] ? j : s[t], n[t][c]);
            }
#pragma unroll
          for (int t = 0; t < NR; ++t)
#pragma unroll
            for (int c = 0; c < CP; ++c) f[t][c] = s[t] + off + sub + GROUP * c < e[t] ? expansion(n[t][c]) : NEG_INF;
        }
        if constexpr (!EXACT) {
#pragma unroll
          for (int t = 0; t < NR; ++t)
#pragma unroll
            for (int c = 0; c < CP; ++c) {
              const bool better = f[t][c] > fb;
              fsecond = fmax_nn(fsecond, better ? fb : f[t][c]);
              jb = better ? s[t] + off + sub + GROUP * c : jb;
              fb = better ? f[t][c] : fb;
            }
          F = fmax_nn(F, group_max(fb));
          thr = thr_of(F);
        } else {
          unsigned pend = 0u;
#pragma unroll
          for (int t = 0; t < NR; ++t)
#pragma unroll
            for (int c = 0; c < CP; ++c)
              pend |= (f[t][c] > NEG_INF && !(f[t][c] < thr)) ? 1u << (t * CP + c) : 0u;
          while (__any(pend != 0u)) {
            if (pend != 0u) {
              const int q = __builtin_ctz(pend);
              pend &= pend - 1u;
              int st = s[0];
#pragma unroll
              for (int t = 1; t < NR; ++t) st = q / CP == t ? s[t] : st;
              exact_code(st + off + sub + GROUP * (q % CP));
            }
          }
        }
      }
    };
    using N1T = std::integral_constant<int, 1>;
    using NST = std::integral_constant<int, SUB_U>;
    using CLT = std::integral_constant<int, CPL>;
    using NFT = std::integral_constant<int, FIRST_N>;
    auto nonempty = [&](int leaf) { return s_start[leaf + 1] > s_start[leaf]; };
    GQ_GRID_STAMP();   // 1: operands, bounds, margins

#ifdef GQHIP_ABL
    const bool abl_no_search = (p.abl & 1) != 0, abl_greedy_only = (p.abl & 2) != 0;
#else
    constexpr bool abl_no_search = false, abl_greedy_only = false;
#endif
    if (!bad && !abl_no_search) {
      // ---- (1) a good F before anything is pruned (LDS only, then ONE round trip).  Concave rows (the trained operating point: the
      //      box that holds the parabola's vertex is the best one, and its bound is tight): greedy descent to one leaf, its 64 codes.
      //      Otherwise (sigma ~ 1: a nearly linear score reaches into the codebook's tails, where boxes are large and their bounds
      //      loose -- the greedy leaf holds the winner for a third of such rows): a beam -- the two best L1 nodes, the four best of
      //      their 32 L2 nodes, the FIRST_N best of those nodes' 16 leaves (lists under the resulting thr: 12 instead of 30 leaves).
      int first_leaf[FIRST_N];
#pragma unroll
      for (int t = 0; t < FIRST_N; ++t) first_leaf[t] = -1;
      const float ub1 = grid_box_ub<DIM>(cA, cB, cM, s_box + sub * BOXF, concave);
      if (concave) {
        const unsigned b1 = group_bits(ub1 == group_max(ub1) && ub1 > NEG_INF);
        if (b1 != 0u) {
          const int g1 = __builtin_ctz(b1);
          const float ub2 = grid_box_ub<DIM>(cA, cB, cM, s_box2 + (g1 * 16 + sub) * BOXF, concave);
          const unsigned b2 = group_bits(ub2 == group_max(ub2) && ub2 > NEG_INF);
          if (b2 != 0u) {
            const int node = g1 * 16 + __builtin_ctz(b2);
            const int lq = node * LPN + (sub & (LPN - 1));               // (every lane: leaf sub % 4 of the node)
            const float ub3 = nonempty(lq) ? grid_box_ub<DIM>(cA, cB, cM, s_box3 + lq * BOXF, concave) : NEG_INF;
            const unsigned b3 = group_bits(ub3 == group_max(ub3) && ub3 > NEG_INF) & ((1u << LPN) - 1u);
            if (b3 != 0u) {
              first_leaf[0] = node * LPN + __builtin_ctz(b3);
              const int s1[1] = {s_start[first_leaf[0]]}, e1[1] = {s_start[first_leaf[0] + 1]};
              visit(N1T{}, CLT{}, std::false_type{}, s1, e1);
              units = SPL;
            }
          }
        }
      } else {
        float u1m = ub1;
        const unsigned ba = group_bits(u1m == group_max(u1m) && u1m > NEG_INF);
        if (ba != 0u) {
          const int g1a = __builtin_ctz(ba);
          u1m = sub == g1a ? NEG_INF : u1m;
          const unsigned bb = group_bits(u1m == group_max(u1m) && u1m > NEG_INF);
          const int g1b = bb != 0u ? __builtin_ctz(bb) : g1a;
          const int id2a = g1a * 16 + sub, id2b = g1b * 16 + sub;
          float c2a = grid_box_ub<DIM>(cA, cB, cM, s_box2 + id2a * BOXF, false);
          float c2b = bb != 0u ? grid_box_ub<DIM>(cA, cB, cM, s_box2 + id2b * BOXF, false) : NEG_INF;
          int nd = 0;
          bool on = false;
#pragma unroll
          for (int t = 0; t < GROUP / LPN; ++t) {                        // the four best of the 32 L2 nodes; lane -> (the (sub / 4)-th, leaf sub % 4)
            const float em = fmax_nn(c2a, c2b);
            const float gm = group_max(em);
            const unsigned ob = group_bits(em == gm && em > NEG_INF);
            const int owner = ob != 0u ? __builtin_ctz(ob) : 0;
            const bool first = c2a == gm;
            const int node = __shfl(first ? id2a : id2b, gshift + owner);
            if (sub / LPN == t) { nd = node; on = ob != 0u; }
            if (ob != 0u && sub == owner) { c2a = first ? NEG_INF : c2a; c2b = first ? c2b : NEG_INF; }
          }
          const int lq = nd * LPN + (sub & (LPN - 1));
          float u3 = on && nonempty(lq) ? grid_box_ub<DIM>(cA, cB, cM, s_box3 + lq * BOXF, false) : NEG_INF;
          int sf[FIRST_N], ef[FIRST_N];
          int nfirst = 0;
#pragma unroll
          for (int t = 0; t < FIRST_N; ++t) {
            const unsigned ob = group_bits(u3 == group_max(u3) && u3 > NEG_INF);
            const int owner = ob != 0u ? __builtin_ctz(ob) : 0;
            const int lf = __shfl(lq, gshift + owner);
            if (ob != 0u) { first_leaf[t] = lf; nfirst = t + 1; }
            u3 = (ob != 0u && sub == owner) ? NEG_INF : u3;
            sf[t] = s_start[ob != 0u ? lf : 0];
            ef[t] = ob != 0u ? s_start[lf + 1] : sf[t];
          }
          if (nfirst > 0) {
            visit(NFT{}, CLT{}, std::false_type{}, sf, ef);
            units = SPL * nfirst;
          }
        }
      }
      GQ_GRID_STAMP();   // 2: greedy descent (one round trip)
      if (abl_greedy_only) thr = __builtin_inff();
      unsigned short *list = s_leaflist[slot];
      float *sub_ub = s_subub[slot];
      unsigned *sub_pk = s_subpk[slot];
      unsigned *sel = s_sel[slot];
      const unsigned lt = (1u << sub) - 1u;
      float eu[EPL];                                                      // this lane's entries of the sub-list: bound (-inf: visited / none)
      unsigned ep[EPL], vis = 0u;                                         // ... packed range; bit k of vis: entry k was visited
#pragma unroll
      for (int k = 0; k < EPL; ++k) { eu[k] = NEG_INF; ep[k] = 0u; }
      for (int round = 0;; ++round) {
        // ---- (2) the leaves whose bound -- and whose L1 and L2 nodes' bounds -- are within the margin of F -> a list (LDS / VALU
        //      only).  Per passing L1 node: its 16 L2 bounds (one per lane) -> the passing L2 nodes, as bytes; then their leaves four
        //      nodes at a time: lane -> (the (sub / 4)-th of them, leaf sub % 4).  (Going from each L1 node straight to its own L2
        //      nodes' leaves left three of four lanes idle in most passes: 7 passes instead of 4 on sigma ~ 1 rows.)
        int nleaf = 0, nl2 = 0;
        unsigned char *l2l = reinterpret_cast<unsigned char *>(sub_ub);     // (the sub-list's LDS is free until the fetches below)
        unsigned m1 = group_bits(!(ub1 < thr) && ub1 > NEG_INF);
        while (m1 != 0u) {                                                 // the passing L2 nodes of the passing L1 nodes -> a byte list
          const int q1 = __builtin_ctz(m1);
          m1 &= m1 - 1u;
          const float ub2 = grid_box_ub<DIM>(cA, cB, cM, s_box2 + (q1 * 16 + sub) * BOXF, concave);
          const bool c2 = !(ub2 < thr) && ub2 > NEG_INF;
          const unsigned pm2 = group_bits(c2);
          if (c2) l2l[nl2 + __builtin_popcount(pm2 & lt)] = (unsigned char)(q1 * 16 + sub);
          nl2 += __builtin_popcount(pm2);
        }
        wave_sync_lds();
        for (int i0 = 0; i0 < nl2; i0 += GROUP / LPN) {                    // their leaves, four nodes at a time whatever L1 node they hang under
          const int k = i0 + sub / LPN;
          const bool on = k < nl2;
          const int lq = (int)l2l[on ? k : 0] * LPN + (sub & (LPN - 1));
          const float ub3 = grid_box_ub<DIM>(cA, cB, cM, s_box3 + lq * BOXF, concave);
          bool c = on && nonempty(lq) && !(ub3 < thr);
#pragma unroll
          for (int t = 0; t < FIRST_N; ++t) c = c && lq != first_leaf[t];
          const unsigned pm = group_bits(c);
          const int pos = nleaf + __builtin_popcount(pm & lt);
          if (c && pos < kGridLeafCap) list[pos] = (unsigned short)lq;
          nleaf += __builtin_popcount(pm);
        }
        bool truncated = nleaf > kGridLeafCap;
        nleaf = min(nleaf, kGridLeafCap);
        if (nleaf > p.inwave_cap) { overflow = true; break; }
        wave_sync_lds();
        GQ_GRID_STAMP();   // 3: leaf list
        // ---- (2b) the sub-leaves of the listed leaves: their boxes and code ranges come from the cache (L2), 4 leaves x 4 sub-leaves
        //      = the group's 16 lanes per fetch, CHUNKS fetches in flight; the ones within the margin -> the sub-list (bound, range)
        int nsub = 0;
        bool big = false;
        // (wave-uniform trip counts below come from __any over the ACTIVE lanes: the registers of a group that has left -- a row
        //  with non-finite bounds, an overflowed row -- hold whatever was there)
        auto fetch = [&](auto nc_tag, int base) {
          constexpr int NC = decltype(nc_tag)::value;
          float lo[NC][DIM], hi[NC][DIM];
          int r0[NC], r1[NC];
          bool val[NC];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int k = base + c * (GROUP / SPL) + sub / SPL;
            val[c] = k < nleaf;
            const int sid = val[c] ? (int)list[k] * SPL + (sub & (SPL - 1)) : 0;   // (a group without that many leaves: any valid address)
            grid_load_box<DIM>(reinterpret_cast<const float *>(reinterpret_cast<const char *>(gsbox) + (unsigned)sid * (unsigned)(BOXF * 4)),
                               lo[c], hi[c]);
            const int *rp = reinterpret_cast<const int *>(reinterpret_cast<const char *>(gsstart) + (unsigned)sid * 4u);
#ifdef GQHIP_NO_BODY_CLAMP       // diagnostic build: what the bounds on the cache body cost
            r0[c] = rp[0];
            r1[c] = rp[1];
#else
            // (bounded by the codebook size -- see the copy of `start` above --; one unsigned minimum each: a negative word reads as n,
            //  and a range that ends before it starts is empty through the `len > 0` test below.  +1.2 us of 62 at the trained
            //  operating point with two instructions each, profiles/r06/grid_body_clamp_ab.txt)
            r0[c] = (int)min((unsigned)rp[0], (unsigned)p.n);
            r1[c] = (int)min((unsigned)rp[1], (unsigned)p.n);
#endif
          }
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const float ubs = grid_box_ub_v<DIM>(cA, cB, cM, lo[c], hi[c], concave);
            const int len = r1[c] - r0[c];
            const bool cnd = val[c] && len > 0 && !(ubs < thr);
            big = big || (cnd && len > 255);
            const unsigned pm = group_bits(cnd);
            const int pos = nsub + __builtin_popcount(pm & lt);
            if (cnd && pos < kGridSubCap) { sub_ub[pos] = ubs; sub_pk[pos] = (unsigned)r0[c] | ((unsigned)min(len, 255) << 24); }
            nsub += __builtin_popcount(pm);
          }
        };
        for (int base = 0; __any(base < nleaf); base += CHUNKS * (GROUP / SPL)) {
          if (!__any(nleaf - base > GROUP / SPL)) fetch(N1T{}, base);     // (the trained operating point: one to four listed leaves)
          else fetch(std::integral_constant<int, CHUNKS>{}, base);
        }
        // (a sub-leaf of more than 255 codes -- a degenerate codebook: most of it in one cell -- is a block's work, not 16 lanes')
        if (group_bits(big) != 0u) { overflow = true; break; }
        truncated = truncated || nsub > kGridSubCap;
        nsub = min(nsub, kGridSubCap);
        wave_sync_lds();
        vis = 0u;
        int kmax = 0;                                                     // (uniform) list slots per lane any group of the wave uses
#pragma unroll
        for (int k = 0; k < EPL; ++k) kmax += __any(nsub > GROUP * k) ? 1 : 0;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
          const int i = sub + GROUP * k;
          eu[k] = NEG_INF;
          ep[k] = 0u;
          if (k < kmax) {
            eu[k] = i < nsub ? sub_ub[i] : NEG_INF;
            ep[k] = i < nsub ? sub_pk[i] : 0u;
          }
        }
        GQ_GRID_STAMP();   // 4: sub-list
        // ---- (3) SUB_U sub-leaves per round trip until the best remaining bound is below the threshold.  The first batch of a long
        //      list: the entries in the upper half of [thr, best bound] (F is probably there); afterwards, and when everything fits
        //      one batch, whatever is still within the margin, in list order.
        int budget = truncated ? 2 * SUB_U : kGridSubCap;                 // a truncated list: its best part, then ONE rebuild with the tighter thr
        int remaining = nsub;
        bool first_batch = true;
        while (budget > 0) {
          float em = eu[0];
#pragma unroll
          for (int k = 1; k < EPL; ++k)
            if (k < kmax) em = fmax_nn(em, eu[k]);
          const float gm = group_max(em);
          if (!(gm > NEG_INF) || gm < thr) break;
          float cut = thr;
          if (first_batch && remaining > SUB_U) {
            cut = thr > NEG_INF ? __builtin_fmaxf(thr, 0.5f * gm + 0.5f * thr) : gm;
            cut = cut <= gm ? cut : gm;                                    // (rounding; NaN)
          }
          first_batch = false;
          wave_sync_lds();                                                // (the previous batch's reads of sel)
          int taken = 0;
#pragma unroll
          for (int k = 0; k < EPL; ++k) {
            if (k >= kmax) continue;
            const bool c = eu[k] > NEG_INF && !(eu[k] < cut);
            const unsigned pm = group_bits(c);
            const int r = taken + __builtin_popcount(pm & lt);
            if (c && r < SUB_U) { sel[r] = ep[k]; eu[k] = NEG_INF; vis |= 1u << k; }
            taken += __builtin_popcount(pm);
          }
          taken = min(taken, SUB_U);
          wave_sync_lds();
          int s[SUB_U], e[SUB_U];
#pragma unroll
          for (int t = 0; t < SUB_U; ++t) {
            const unsigned pk = sel[t < taken ? t : 0];
            s[t] = (int)(pk & 0xffffffu);
            e[t] = t < taken ? s[t] + (int)(pk >> 24) : s[t];
          }
          visit(NST{}, N1T{}, std::false_type{}, s, e);
          units += taken;
          budget -= taken;
          remaining -= taken;
        }
        if (!truncated) break;
        // still more than the lists hold: a row with a long tail of loose boxes (near-linear scores reach far into the codebook's
        // tails).  ONE rebuild under the tighter threshold; every code that can still matter is in a box that is listed again, so the
        // lanes' best codes start over (F and thr stay) and the first leaf is listed like any other.  A second truncation: the row
        // goes to gq_grid_finish_kernel (a whole block finishes it in a few microseconds; 16 lanes would hold up their wave).
        if (round >= 1 || units > p.leaf_cap * SPL) { overflow = true; break; }
        fb = NEG_INF; fsecond = NEG_INF; jb = -1;
#pragma unroll
        for (int t = 0; t < FIRST_N; ++t) first_leaf[t] = -1;
      }
      GQ_GRID_STAMP();   // 5: listed sub-leaves
      // ---- (4) the final threshold is known: the reference's arithmetic for every code within the margin
      if (!overflow) {
        const bool cand = jb >= 0 && !(fb < thr);
        const bool tie = group_bits(fsecond > NEG_INF && !(fsecond < thr)) != 0u;
        if (__builtin_popcount(group_bits(cand)) == 1 && !tie) {
          // ONE code of the row within the margin of its best expansion: every other code's reference score is provably below this
          // code's, whatever the two are -- it is the argmax, and the reference's arithmetic (a hundred-odd instructions that the whole
          // wave would sit through) is not needed to say so
          if (cand) { best_j = jb; best_s = 0.0; have = true; }        // (its original id: fetched with the result, below)
        } else {
        if (cand) exact_code(jb);
        if (tie) {
          // a lane holds a second code within the margin (a near-tie): everything visited again, every code within the margin exactly
#pragma unroll
          for (int t = 0; t < FIRST_N; ++t)
            if (first_leaf[t] >= 0) {                                      // (group-uniform)
              const int s1[1] = {s_start[first_leaf[t]]}, e1[1] = {s_start[first_leaf[t] + 1]};
              visit(N1T{}, CLT{}, std::true_type{}, s1, e1);
            }
#pragma unroll
          for (int k = 0; k < EPL; ++k) {
            unsigned bits = group_bits(((vis >> k) & 1u) != 0u);
            while (bits != 0u) {
              const int owner = __builtin_ctz(bits);
              bits &= bits - 1u;
              const unsigned pk = (unsigned)__shfl((int)ep[k], gshift + owner);
              const int s1[1] = {(int)(pk & 0xffffffu)}, e1[1] = {(int)(pk & 0xffffffu) + (int)(pk >> 24)};
              visit(N1T{}, N1T{}, std::true_type{}, s1, e1);
            }
          }
        }
        }
      }
    }
    GQ_GRID_STAMP();     // 6: exact pass
    // ---- the row's winner: the one lane that holds a code, or the best of several (score, then the lower code)
    const unsigned hb = group_bits(have);
    if (__builtin_popcount(hb) == 1) {
      best_j = __shfl(best_j, gshift + __builtin_ctz(hb));
      have = true;
    } else if (hb != 0u) {
#pragma unroll
      for (int o = GROUP / 2; o > 0; o >>= 1) {
        const double os = __shfl_xor(best_s, o);
        const int oi = __shfl_xor(best_i, o), oj = __shfl_xor(best_j, o);
        const bool oh = __shfl_xor((int)have, o) != 0;
        if (oh && (!have || better_d(os, oi, best_s, best_i))) { best_s = os; best_i = oi; best_j = oj; have = true; }
      }
    }
    const bool decided = live && !bad && !overflow && have;
    if (decided) {
      // the winner's original id and its code row come from the SORTED tables at the same position: two independent loads, one round
      // trip (through cb[sidx[j]] it was two; the sorted row is a bit copy of the caller's, and the cache was validated by this call)
      if (sub == 0) p.idx[out_idx_offset(p.omap, row)] = (int64_t)min((unsigned)sidx[best_j], (unsigned)(p.n - 1));
      if (p.zhat && sub < DIM) {
        const long o = out_zhat_offset(p.omap, row, sub, DIM);
        p.zhat[o] = ste_mix(p.hdr, o, scb[(long)best_j * DIM + sub]);
      }
    }
    if (p.stats && live) {
      int e = exact_n;
#pragma unroll
      for (int o = GROUP / 2; o > 0; o >>= 1) e += __shfl_xor(e, o);
      if (sub == 0) {
        atomicAdd(&s_stat_exact, (unsigned)e);
        atomicAdd(&s_stat_units, (unsigned)units);
      }
    }
    // ---- an undecided row: to the call's list; gq_grid_finish_kernel (the next launch) finishes it with a whole block
    if (live && !decided && sub == 0) {
      const bool keep_all = bad || (!overflow && !have) || !(thr > NEG_INF);      // (an overflowed row: its candidate leaves under thr)
      const int pos = atomicAdd(&p.hdr->fb_count, 1);
      p.und_row[pos] = (int)(row << 1) | (keep_all ? 1 : 0);
      p.und_thr[pos] = thr;
      p.und_margin[pos] = margin32;
    }
#ifdef GQHIP_CLOCK_STAMPS
    GQ_GRID_STAMP();     // 7: reduce, stores
    if (lane == 0 && first_pass_ && wave == 0 && blockIdx.x % 100 == 0 && blockIdx.x / 100 < 6) {   // six blocks' wave 0, first four rows
      for (int k = 0; k < 8; ++k) p.hdr->stamps[8 * (blockIdx.x / 100) + k] = k < nst_ ? st_[k] : 0ull;
    }
    first_pass_ = false;
    nst_ = 8;
#endif
  }
  if (p.stats) {                                         // (uniform)
    __syncthreads();
    if (tid == 0) {
      atomicAdd(&p.hdr->reranked, (unsigned long long)s_stat_exact);
      atomicAdd(&p.hdr->grid_leaves, (unsigned long long)s_stat_units);
    }
  }
}

// The rows the search left undecided (the 4th launch of a dim-4 call; exits at once when there are none).  A block per row, round
// robin over the call's list -- its own kernel so that clustered rows (flat image regions give runs of them) spread over the
// whole chip, and so that this code's registers are not the search's.
template <int MODE, int DIM>
__global__ __launch_bounds__(kGridThreads, 2) void gq_grid_finish_kernel(const GridParams p) {
  const int nund = p.hdr->fb_count;
  if (nund == 0) return;                                  // (uniform)
  constexpr int BOXF = 2 * DIM;
  constexpr int START_N = kGridLeaves + 16;
  __shared__ __attribute__((aligned(16))) float s_box3f[kGridLeaves * BOXF];
  __shared__ __attribute__((aligned(16))) int s_start[START_N];
  __shared__ int s_nblist;
  __shared__ float s_sops[3 * DIM + 1];
  __shared__ float s_red[kGridThreads / 64];
  __shared__ double sh_s[kGridThreads];
  __shared__ int sh_i[kGridThreads];
  const GridLayout L = grid_layout(p.n, DIM);
  const float *scb = reinterpret_cast<const float *>(p.cache + L.scb);
  const int *sidx = reinterpret_cast<const int *>(p.cache + L.sidx);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float NEG_INF = -__builtin_inff();
  {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(p.cache + L.box3);
    f32x4 *dst = reinterpret_cast<f32x4 *>(s_box3f);
    for (int k = tid; k < kGridLeaves * BOXF / 4; k += kGridThreads) dst[k] = src[k];
    // (leaf ranges are clamped to [0, n] on their way into LDS: a cache body that was written behind the library's back -- the header's
    //  stamps cannot see that -- may then cost a wrong index, never an out-of-range read; gqhip.h, codebook cache)
    const i32x4 *src2 = reinterpret_cast<const i32x4 *>(p.cache + L.start);
    i32x4 *dst2 = reinterpret_cast<i32x4 *>(s_start);
    for (int k = tid; k < START_N / 4; k += kGridThreads) {
      i32x4 v = src2[k];
      v.x = min(max(v.x, 0), p.n); v.y = min(max(v.y, 0), p.n); v.z = min(max(v.z, 0), p.n); v.w = min(max(v.w, 0), p.n);
      dst2[k] = v;
    }
  }
  const float *s_box3 = s_box3f;
  // ---- the block's undecided rows, one after the other, all 512 threads.  A row with a finite threshold (a list that did not fit:
  //      a long tail of loose boxes) is finished over its CANDIDATE LEAVES only: every thread bounds two of the 1024 leaves, the
  //      passing ones go to a list, a wave takes a leaf at a time (64 lanes = its 64 codes): pass 1 the largest expansion -> the
  //      final threshold, pass 2 the reference's arithmetic for the codes within the margin.  (A scan of all codes costs ~20 us of
  //      the block per row: 1.5 % of the rows of the bench's gq_1.00 z are such rows.)  Rows with non-finite operands / bounds, or
  //      behind a non-finite codebook: all codes, everything exact.
  __syncthreads();
  int *s_blist = reinterpret_cast<int *>(sh_s);          // (sh_s is free until a row's final reduction)
  static_assert(sizeof(double) * kGridThreads >= sizeof(int) * kGridLeaves, "the block's leaf list fits the reduction buffer");
  for (int i = (int)blockIdx.x; i < nund; i += (int)gridDim.x) {
    const long row = p.und_row[i] >> 1;
    const bool keep_all = (p.und_row[i] & 1) != 0;
    const float thr = p.und_thr[i];
    if (tid < DIM) {
#pragma clang fp contract(off)
      s_sops[tid] = p.mu[row * DIM + tid];
      if constexpr (MODE == kModeGQ) {
        const float sg = p.sd[row * DIM + tid];
        s_sops[DIM + tid] = 2.0f * (sg * sg);
        s_sops[2 * DIM + tid] = p.lsd[row * DIM + tid];
      }
    }
    if (tid == 0) s_nblist = 0;
    __syncthreads();
    float sA[DIM], sB[DIM];
    grid_load_coef<DIM>(p, row, sA, sB);
    double bs = 0.0;
    int bi = 0x7fffffff;
    bool hv = false;
    if (keep_all) {
      grid_scan_accumulate<MODE, DIM>(p, sA, sB, s_sops, thr, true, tid, kGridThreads, bs, bi, hv);
    } else {
      float sM[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) sM[k] = sA[k] < 0.0f ? -0.5f * sB[k] * __builtin_amdgcn_rcpf(sA[k]) : 0.0f;
      for (int lf = tid; lf < kGridLeaves; lf += kGridThreads) {
        const float ub = s_start[lf + 1] > s_start[lf] ? grid_box_ub<DIM>(sA, sB, sM, s_box3 + lf * BOXF, false) : NEG_INF;
        if (ub > NEG_INF && !(ub < thr)) s_blist[atomicAdd(&s_nblist, 1)] = lf;
      }
      __syncthreads();
      const int nl = s_nblist;
      auto leaf_codes = [&](auto &&body) {                // wave w: leaves w, w + 8, ...; lane: codes lane, lane + 64, ... of the leaf
        for (int k = wave; k < nl; k += kGridThreads / 64) {
          const int lf = s_blist[k];
          for (int j = s_start[lf] + lane; j < s_start[lf + 1]; j += 64) {
            float n[DIM];
            const f32x4 *q = reinterpret_cast<const f32x4 *>(scb + (long)j * DIM);
#pragma unroll
            for (int c = 0; c < DIM / 4; ++c) {
              const f32x4 v = q[c];
              n[4 * c] = v.x; n[4 * c + 1] = v.y; n[4 * c + 2] = v.z; n[4 * c + 3] = v.w;
            }
            float f = 0.0f;
#pragma unroll
            for (int d = 0; d < DIM; ++d) f = __builtin_fmaf(__builtin_fmaf(sA[d], n[d], sB[d]), n[d], f);
            body(j, n, f);
          }
        }
      };
      float fm = NEG_INF;
      leaf_codes([&](int, const float (&)[DIM], float f) { fm = __builtin_fmaxf(fm, f); });
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) fm = __builtin_fmaxf(fm, __shfl_xor(fm, o));
      if (lane == 0) s_red[wave] = fm;
      __syncthreads();
      fm = s_red[0];
#pragma unroll
      for (int w = 1; w < kGridThreads / 64; ++w) fm = __builtin_fmaxf(fm, s_red[w]);
      const float t2 = (fm - p.und_margin[i]) - 2.4e-7f * __builtin_fabsf(fm);
      const float thr2 = t2 > thr ? t2 : thr;             // (the row's own F was found among these leaves' codes or earlier: fm >= it)
      leaf_codes([&](int j, const float (&n)[DIM], float f) {
        if (!(f < thr2)) {
          const int code = (int)min((unsigned)sidx[j], (unsigned)(p.n - 1));
          double sc;
          if constexpr (MODE == kModeGQ) sc = (double)ref_score_lds<DIM>(n, s_sops, p.beta);
          else sc = vq_neg_dist(n, s_sops, DIM);
          if (!hv || better_d(sc, code, bs, bi)) { bs = sc; bi = code; hv = true; }
        }
      });
      __syncthreads();                                    // (the leaf list lives in sh_s)
    }
    sh_s[tid] = bs;
    sh_i[tid] = hv ? bi : 0x7fffffff;
    __syncthreads();
    for (int o = kGridThreads / 2; o > 0; o >>= 1) {
      if (tid < o) {
        const double os = sh_s[tid + o];
        const int oi = sh_i[tid + o];
        const bool mine = sh_i[tid] != 0x7fffffff;
        if (oi != 0x7fffffff && (!mine || better_d(os, oi, sh_s[tid], sh_i[tid]))) { sh_s[tid] = os; sh_i[tid] = oi; }
      }
      __syncthreads();
    }
    // (no code at all can only come out of a cache body that was overwritten behind an intact header -- every range empty --: the
    //  row still gets a defined, in-range result)
    const int best = sh_i[0] != 0x7fffffff ? sh_i[0] : 0;
    grid_write_result(p, row, best, tid, DIM);
    __syncthreads();
  }
}

}  // namespace gqhip