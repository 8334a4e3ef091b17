// gq_filter_bf16.h -- the filter on the bf16 matrix cores (split-precision variant of gq_filter.h).
//
// Same contract as gq_filter_kernel: for every row r and code j it evaluates
//     f(r, j) = sum_i  A[r,i] * c[j,i]^2  +  B[r,i] * c[j,i]
// and keeps, per (row, code split), the four largest half-group maxima + the ids of the best three (three / two
// in the packed dim-4 kernel); the re-rank kernel then decides exactly.  The products are formed on v_mfma_f32_32x32x16_bf16 (16x the fp32
// MFMA rate) from two-term bf16 splits of every fp32 operand q = q_h + q_l (+ residual <= 2^-18 |q|):
//     A * s  ~  A_h s_h  +  A_h s_l  +  A_l s_h          (each bf16 x bf16 product is exact in fp32)
// so a fp32 MAC costs three bf16 MACs and the filter value carries a relative error of ~3 * 2^-18 per
// product instead of ~2^-24 -- which only widens the re-rank margin (gq_rerank.h: `ef_coeff`): the indices
// stay bit-identical to the reference because the decision is still taken by the exact re-rank.
//
// K layout.  Per "type" (hh, lh, hl) there are 2*DIM slots [ squares of dims 0..DIM-1 | values of dims
// 0..DIM-1 ]; MFMA m of a type covers slots 16m .. 16m+15, lane half h supplies slots 16m+8h .. 16m+8h+7.
// NV = DIM/8 MFMAs per type (DIM = 4 packs its three 8-slot types into two MFMAs, see below).  Code side: vectors 0..NV-1 hold the h parts, NV..2NV-1 the l parts; row side
// likewise with the coefficients [A | B].  MFMA (type, m):  hh -> (code m, row m), lh -> (code NV+m, row m),
// hl -> (code m, row NV+m).  Both images are produced once per call by gq_prep_kernel (gq_prep.h), the
// codebook image already in the LDS tile order, so staging is a linear 16-byte copy.
//
// Tile image: [tile][code vector cv][half h][code c] x 16 bytes  ->  a wave's ds_read_b128 of one vector is
// 1 KiB contiguous (conflict-free).  D layout as in gq_filter.h (lane = one row, 16 codes).
#pragma once
#include <type_traits>
#include <utility>

#include "gq_filter.h"

namespace gqhip {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct FilterBfParams {
  const u32x4 *cbimg;   // [tiles_total + CT][2*NV][2][32]
  const u32x4 *rowimg;  // [rows][2*NV][2]
  Rec *rec;             // [nsplit, rows]
  int rows, n;
  int nsplit, tiles_total, tiles_per_split;
  WsHeader *hdr;
  void *dbg;            // diagnostic builds only
  const float *rowscale;   // MIXED: [rows] 2^e_r, the records leave the kernel multiplied by it (true units)
};

// round-to-nearest-even fp32 -> bf16 (as the upper 16 bits); finite inputs (non-finite rows / codebooks never
// reach a decision through the filter: their bound is not finite and the re-rank scans every code for them).
__device__ __forceinline__ unsigned bf16_rne(float f) {
  const unsigned x = __float_as_uint(f);
  return (x + 0x7fffu + ((x >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void bf16_split(float q, unsigned &hi, unsigned &lo) {
  hi = bf16_rne(q);
  const float r = q - __uint_as_float(hi << 16);   // exact in fp32
  lo = bf16_rne(r);
}

// Both operand images are built by gq_prep_kernel (gq_prep.h), once per call.
// DIM == 4 ("packed", NV = 0 in the filter): a type has only 8 slots, so two MFMAs carry the three types:
//   MFMA 0: half 0 = hh (code h parts x row h parts), half 1 = lh (code l parts x row h parts)
//   MFMA 1: half 0 = hl (code h parts x row l parts), half 1 = zero padding

// f(integral_constant<int, 0>{}, ..., integral_constant<int, N-1>{})
template <class F, int... I>
__device__ __forceinline__ void call_with_indices(F &&f, std::integer_sequence<int, I...>) {
  f(std::integral_constant<int, I>{}...);
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// Layout constants: NV = DIM / 8 MFMAs per type (NV = 0: the packed DIM = 4 layout, two MFMAs, two vectors).
template <int NV> struct BfLayout {
  static constexpr int NCV = NV ? 2 * NV : 2;   // code (and row) vectors per tile
  static constexpr int NM = NV ? 3 * NV : 2;    // MFMAs per tile and row tile
};

// MFMAs [S0, S1) of one tile, for the wave's RT row tiles (chains alternate in program order).
template <int NV, int RT, int S0, int S1>
__device__ __forceinline__ void tile_mfma_bf16(const u32x4 (&cv)[BfLayout<NV>::NCV],
                                               const u32x4 (&rv)[RT][BfLayout<NV>::NCV], f32x16 (&d)[RT]) {
  if constexpr (S0 == 0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      d[rt] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int s = S0; s < S1; ++s) {
    int ci = s, ri = s;                              // packed: MFMA s uses (code vector s, row vector s)
    if constexpr (NV > 0) {
      const int type = s / NV, m = s % NV;           // 0 hh, 1 lh, 2 hl
      ci = type == 1 ? NV + m : m;
      ri = type == 2 ? NV + m : m;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      d[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(cv[ci]), as_bf16x8(rv[rt][ri]), d[rt], 0, 0, 0);
  }
}

// ---- MIXED (round 2): the same filter value from ONE fp16 product and fp8 corrections.  Every operand q is split as
// q = q_h + q_l with q_h = fp16(q) (11 significant bits: |q_l| <= 2^-11 |q|):
//     A s  =  A_h s_h  +  A_h s_l  +  A_l s_h  (+ A_l s_l <= 2^-22 |A s|)
// The main product A_h s_h runs on v_mfma_f32_32x32x16_f16 (two K = 16 steps for the 32 slots of dim 16); both
// correction types together are ONE v_mfma_scale_f32_32x32x64_f8f6f4 with OCP e4m3 operands: K block 0 = (s_l 2^11) x
// (A_h 2^-6), K block 1 = s_h x (A_l 2^6), the E8M0 block scales 2^-11 2^6 and 2^-6 put them back (constants: the row's
// coefficients are normalised by a power of two 2^-e_r in gq_prep_kernel so that their largest is in [2^13, 2^14); the
// record values leave this kernel multiplied by 2^e_r).  A correction is at most 2^-11 of its product (fp16: 11 significant
// bits) and an e4m3 operand carries a relative rounding error <= 2^-4, so each correction type is accurate to
// (2^-3 + 2^-8) 2^-11 |A s| = 1057 u |A s| (u = 2^-24): representation error <= 2130 u per product against 196 u for the
// split-bf16 form -- a wider re-rank margin (`ef_coeff` 2450, incl. 4 u per accumulation step of the main product and 2 u
// per step of the corrections; derivation in DESIGN.md section 3), the same indices.  MFMA passes per tile
// and row tile: 2 x 8 + 16 = 32 instead of 6 x 8 = 48 (the block-scaled fp8 instruction runs K = 64 in 16 passes).
// Measured layout of the scaled instruction (tools/mfma_f8_layout.hip): byte j of lane (r, h) is k = 16 h + (j & 15) +
// 32 (j >> 4) -- the first 16 bytes belong to K block 0, the last 16 to block 1 --, and block b of row / column r takes
// its scale from lane r + 32 b.
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8m __attribute__((ext_vector_type(8)));

__device__ __forceinline__ i32x8 cat_u32x4(u32x4 a, u32x4 b) {
  return i32x8{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w};
}

// MFMAs [S0, S1) of one tile (3 per row tile: fp16 step 0, fp16 step 1, scaled fp8 corrections)
template <int RT, int S0, int S1>
__device__ __forceinline__ void tile_mfma_mixed(const u32x4 (&cv)[4], const u32x4 (&rv)[RT][4], f32x16 (&d)[RT], int sa,
                                                int sb) {
  if constexpr (S0 == 0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      d[rt] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int s = S0; s < S1; ++s) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (s < 2)
        d[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8m, cv[s]), __builtin_bit_cast(f16x8m, rv[rt][s]),
                                                       d[rt], 0, 0, 0);
      else
        d[rt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat_u32x4(cv[2], cv[3]), cat_u32x4(rv[rt][2], rv[rt][3]), d[rt],
                                                                0, 0, 0, sa, 0, sb);
    }
  }
}

// ---- F16 (round 3): the main product alone.  A s ~ A_h s_h on v_mfma_f32_32x32x16_f16, NO correction terms: the value is
// off by the two fp16 roundings, (2^-10 + 2^-22) |A s| per product -- 8x the fp16 + fp8 form's representation error -- and
// the re-rank pays for it with a bound that follows the DATA instead of the worst case (gq_rerank.h:f16_bound: the error of
// a high-scoring code is bounded through its own score, and through the codebook's largest norm), which more than makes up
// the difference: 1.24 candidates per row at config 2 against 1.40 behind the fp16 + fp8 filter (tools/bound_study.py).
// MFMA passes per tile and row tile: dim 16: 16 (32 / 48 before), dim 32: 32 (96), dims 8 and 4: 8 (24 / 16).  One operand
// vector per MFMA and lane: half the LDS reads and half the image of the fp16 + fp8 form.  Applies to VQ as well (A = -1,
// B = 2 z): nothing in it depends on the sign structure of the Gaussian score.
template <int NCV, int RT, int S0, int S1>
__device__ __forceinline__ void tile_mfma_f16(const u32x4 (&cv)[NCV], const u32x4 (&rv)[RT][NCV], f32x16 (&d)[RT]) {
  if constexpr (S0 == 0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      d[rt] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int s = S0; s < S1; ++s)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      d[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8m, cv[s]), __builtin_bit_cast(f16x8m, rv[rt][s]), d[rt],
                                                     0, 0, 0);
}

// WAVES = 8: one 512-thread block per CU, so the two waves that share a SIMD belong to the SAME block and meet at
// every chunk barrier.  With two independent 4-wave blocks per CU the SIMD's arbiter favours one of them: it
// finishes at ~70 % of the kernel time and the other runs the rest alone, without a partner to hide its LDS
// waits and epilogues behind (measured per block with GQHIP_CLOCK_STAMPS: 104 / 149 us).  One block per CU also
// halves the L2 -> LDS staging traffic (one chunk copy serves 8 waves).
// FK: 0 = split-bf16, 1 = fp16 + fp8 (MIXED), 2 = fp16 main product only (F16).
template <int NV, int RT, int CT, int GT, int WAVES, int FK = 0>
__global__ __launch_bounds__(64 * WAVES, WAVES == 8 ? 1 : 2) void gq_filter_bf16_kernel(const FilterBfParams p) {
  constexpr bool MIXED = FK == 1, F16 = FK == 2;
  static_assert(!MIXED || NV == 2, "fp16 + fp8 filter: dim 16");
  constexpr int NCV = F16 ? (NV ? NV : 1) : BfLayout<NV>::NCV;      // code (and row) vectors per tile
  constexpr int TILE_Q = NCV * 64;            // 16-byte slots per tile
  constexpr int CHUNK_Q = CT * TILE_Q;
  constexpr int NT = 64 * WAVES;              // threads per block
  constexpr int R4 = CHUNK_Q / NT;            // 16-byte loads per thread per chunk
  constexpr int NM = F16 ? NCV : (MIXED ? 3 : BfLayout<NV>::NM);   // MFMAs per tile and row tile
  // candidate tracker depth: top-4 (ids of three), except in the packed dim-4 kernel, which is bound by its
  // epilogue's VALU work (top-3 there: +18 % filter time otherwise, measured)
  constexpr bool TOP4 = NV > 0;
  static_assert(R4 >= 1 && CHUNK_Q % NT == 0 && CT % GT == 0, "chunk: whole tile groups, whole thread passes");
  __shared__ u32x4 lds[2][CHUNK_Q];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int split = blockIdx.x % p.nsplit;
  const int rowblk = blockIdx.x / p.nsplit;
  const int t_begin = split * p.tiles_per_split;          // multiple of GT
  const int t_end = min(t_begin + p.tiles_per_split, p.tiles_total);
  const int t_full_end = min(t_end, p.n / kTileCodes);    // complete tiles only

#ifdef GQHIP_CLOCK_STAMPS
  const unsigned long long st_b0 = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- row operands (B side of the MFMA), fixed for the whole kernel ----
  u32x4 rv[RT][NCV];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int row = rowblk * (32 * WAVES * RT) + (wave * RT + rt) * 32 + c;
    row = min(row, p.rows - 1);
#pragma unroll
    for (int v = 0; v < NCV; ++v) rv[rt][v] = p.rowimg[((long)row * NCV + v) * 2 + h];
  }

  // E8M0 block scales of the corrections' MFMA (MIXED): lanes < 32 hold the scale of K block 0, the others of block 1
  const int sa = lane < 32 ? 127 - 11 : 127, sb = lane < 32 ? 127 + 6 : 127 - 6;
  auto tile_mfma = [&](auto s0_tag, auto s1_tag, const u32x4 (&cvx)[NCV], f32x16 (&dx)[RT]) {
    constexpr int A0 = decltype(s0_tag)::value, A1 = decltype(s1_tag)::value;
    if constexpr (F16) tile_mfma_f16<NCV, RT, A0, A1>(cvx, rv, dx);
    else if constexpr (MIXED) tile_mfma_mixed<RT, A0, A1>(cvx, rv, dx, sa, sb);
    else tile_mfma_bf16<NV, RT, A0, A1>(cvx, rv, dx);
  };
  using IC0 = std::integral_constant<int, 0>;
  using ICNM = std::integral_constant<int, NM>;
  const float NEG_INF = -__builtin_inff();
  float m1[RT], m2[RT], m3[RT], m4[RT], tpend[RT];
  int i1[RT], i2[RT], i3[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    m1[rt] = m2[rt] = m3[rt] = m4[rt] = tpend[rt] = NEG_INF;
    i1[rt] = i2[rt] = i3[rt] = 0;
  }

  // ---- chunk staging: global image -> registers -> LDS (a linear copy) ----
  u32x4 stage[R4];
  auto load_chunk = [&](int tile0) {
    const u32x4 *src = p.cbimg + (long)tile0 * TILE_Q + tid;
#pragma unroll
    for (int r = 0; r < R4; ++r) stage[r] = src[NT * r];    // the image is padded by CT tiles
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int r = 0; r < R4; ++r) lds[buf][tid + NT * r] = stage[r];
  };
  auto read_ops = [&](const u32x4 *tile, u32x4 (&cv)[NCV]) {
#pragma unroll
    for (int v = 0; v < NCV; ++v) cv[v] = tile[v * 64];
  };
  auto close_group = [&](int tile) {
#if defined(GQHIP_ABL) && (GQHIP_ABL & 4)      // diagnostic build: no candidate tracker (results are garbage, only the time is read)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { m1[rt] = __builtin_fmaxf(m1[rt], tpend[rt]); tpend[rt] = NEG_INF; }
    return;
#endif
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if constexpr (TOP4)
        top4_insert(tpend[rt], tile / GT, m1[rt], m2[rt], m3[rt], m4[rt], i1[rt], i2[rt], i3[rt]);
      else
        top3_insert(tpend[rt], tile / GT, m1[rt], m2[rt], m3[rt], i1[rt], i2[rt]);
      tpend[rt] = NEG_INF;
    }
  };
  auto fold = [&](f32x16 (&d)[RT], int tile, bool closes) {
#if defined(GQHIP_ABL) && (GQHIP_ABL & 8)      // diagnostic build: no v_max3 fold
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) tpend[rt] = __builtin_fmaxf(tpend[rt], d[rt][0]);
    if (closes) close_group(tile);
    return;
#endif
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) tpend[rt] = F16 ? max_tree(tpend[rt], d[rt]) : max_chain(tpend[rt], d[rt]);
    if (closes) close_group(tile);
  };

  const int ntiles = t_full_end - t_begin;
  const int nchunks = ntiles > 0 ? (ntiles + CT - 1) / CT : 0;
  if (nchunks > 0) {
    load_chunk(t_begin);
    store_chunk(0);
  }
  // Touch the row operands here: the compiler then waits for their loads BEFORE the loop.  Otherwise its
  // wait-count bookkeeping carries "row operands may still be in flight" into the loop, and since loads retire
  // in order the first tile of every chunk would wait for that chunk's prefetch (vmcnt(0) in the hot loop).
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int v = 0; v < NCV; ++v) asm volatile("" ::"v"(rv[rt][v]));
  __syncthreads();

#ifdef GQHIP_CLOCK_STAMPS
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int S0 = NM >= 6 ? 2 : 1;   // MFMA steps issued before the previous tile's epilogue
  using ICS0 = std::integral_constant<int, S0>;
  // Staging schedule of a full chunk (round 2, after the ablation in tools/abl_filter.sh: with the copy of the next chunk
  // at the chunk's end -- 8 ds_write_b128 per thread, barrier, first operand reads -- the matrix pipes drained at every
  // chunk boundary: 137 us without the staging, 143 without the barriers, 155-164 with both):
  //   * the R4 register -> LDS writes of the NEXT chunk are spread over steps W0 .. CT-2 (its buffer is free for the whole
  //     chunk: the previous barrier came after every read of it; the loads were issued at the chunk's first step);
  //   * ONE barrier at the end of step CT-2: every wave's writes are visible, and every wave has finished READING the
  //     current buffer (the operands of tile CT-1 were read at the start of step CT-2);
  //   * step CT-1 then prefetches tile 0 of the next chunk under its own MFMAs, so no step ever starts with an LDS read.
  constexpr int W0 = CT / 2 - 1, WSTEPS = CT - 1 - W0;
  static_assert(CT >= 4 && WSTEPS >= 1, "chunk too short for the staggered staging");
  f32x16 dprev[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dprev[rt][r] = NEG_INF;
  int tprev = t_begin - 1;
  bool have_prev = false;
  u32x4 cv[NCV];
  if (nchunks > 0) read_ops(lds[0] + h * 32 + c, cv);

  for (int ch = 0; ch < nchunks; ++ch) {
    const int tile0 = t_begin + ch * CT;
#if !(defined(GQHIP_ABL) && (GQHIP_ABL & 2))    // diagnostic build: no chunk staging (every chunk multiplies the first one)
    if (ch + 1 < nchunks) load_chunk(tile0 + CT);
#endif
    const int nt = min(CT, t_full_end - tile0);
    const u32x4 *base = lds[ch & 1] + h * 32 + c;          // this lane's slot within a vector
    const u32x4 *nbase = lds[(ch + 1) & 1] + h * 32 + c;
    u32x4 *wdst = &lds[(ch + 1) & 1][tid];
    if (nt == CT) {
      auto step = [&](auto tt_tag) {
        constexpr int TT = decltype(tt_tag)::value;
        constexpr bool PREV_CLOSES = (TT % GT) == 0;
        u32x4 cvn[NCV];
        if constexpr (TT < CT - 1) read_ops(base + (TT + 1) * TILE_Q, cvn);
        else read_ops(nbase, cvn);                         // after the barrier of step CT-2: tile 0 of the next chunk
        // this step's share of the next chunk's copy (stale registers after the last chunk: written, never read)
#pragma unroll
        for (int r = 0; r < R4; ++r)
#if !(defined(GQHIP_ABL) && (GQHIP_ABL & 2))
          if constexpr (TT < CT - 1)
            if (W0 + (r * WSTEPS) / R4 == TT) wdst[NT * r] = stage[r];
#endif
        // hard fence: hipcc otherwise sinks these reads below the MFMAs of this step (seen in the ISA of the round-1 kernel:
        // every tile began with its own operand reads and lgkmcnt waits, the prefetch existed only in the source)
        __builtin_amdgcn_sched_barrier(0);
        f32x16 d[RT];
        tile_mfma(IC0{}, ICS0{}, cv, d);
        fold(dprev, tprev, PREV_CLOSES);
        tile_mfma(ICS0{}, ICNM{}, cv, d);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) dprev[rt] = d[rt];
        tprev = tile0 + TT;
#pragma unroll
        for (int v = 0; v < NCV; ++v) cv[v] = cvn[v];
        // pin the order: every MFMA of this tile followed by a slice of the PREVIOUS tile's epilogue (its accumulators
        // are long complete; bf16 MFMAs and VALU overlap)
        // (VALU budget slightly under the real count, none after the last MFMA: a slot left over would pull the
        // NEXT step's epilogue -- the accumulators just written -- forward and stall on the MFMA latency)
        constexpr int BUDGET = (PREV_CLOSES ? (TOP4 ? 20 : 16) : 8) * RT, K = NM * RT - 1;
        // the first SKIP MFMAs carry no VALU (the previous tile's last accumulators are not readable yet: hipcc pads with
        // s_nop otherwise), the next EXTRA carry PER+1 slots, the rest PER
        constexpr int SKIP = K > 4 ? 2 : 0, SLOTS = K - SKIP;
        constexpr int SLOTS1 = SLOTS > 0 ? SLOTS : 1;       // (one MFMA per step: no slot between MFMAs, nothing to divide)
        constexpr int PER = SLOTS > 0 ? BUDGET / SLOTS1 : 0, EXTRA = SLOTS > 0 ? BUDGET % SLOTS1 : 0;
#pragma unroll
        for (int k = 0; k < SKIP; ++k) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int k = 0; k < EXTRA; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, PER + 1, 0);
        }
#pragma unroll
        for (int k = EXTRA; k < SLOTS; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if constexpr (PER > 0) __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
#if !(defined(GQHIP_ABL) && (GQHIP_ABL & 1))    // diagnostic build: no chunk barriers
        if constexpr (TT == CT - 2) __syncthreads();
#endif
      };
      // one step per tile of the chunk, fully unrolled (the tags are compile-time constants)
      auto run_all = [&](auto... is) { (step(is), ...); };
      call_with_indices(run_all, std::make_integer_sequence<int, CT>{});
      have_prev = true;
    } else {
      // the one partial chunk (always the last): staged by the previous chunk (or the prologue), plain loop
      if (have_prev) fold(dprev, tprev, (tprev % GT) == GT - 1);
      have_prev = false;
      for (int tt = 0; tt < nt; ++tt) {
        u32x4 cw[NCV];
        read_ops(base + tt * TILE_Q, cw);
        f32x16 d[RT];
        tile_mfma(IC0{}, ICNM{}, cw, d);
        fold(d, tile0 + tt, ((tile0 + tt) % GT) == GT - 1);
      }
    }
  }
  if (have_prev) fold(dprev, tprev, (tprev % GT) == GT - 1);
#ifdef GQHIP_CLOCK_STAMPS
  const unsigned long long st_r2 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0 && blockIdx.x < 24) {   // diagnostic build only: shader clocks and 100 MHz ticks spent in the main loop
    p.hdr->stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c0;
    p.hdr->stamps[2 * blockIdx.x + 1] = st_r2 - st_r0;
  }
#endif

  // ---- leftovers: the one partial tile (n % 32 != 0; zero-padded in the image) / an open group ----
  const bool pending = ntiles > 0 && (ntiles % GT) != 0;
  if (t_end > t_full_end) {
    const int tile = t_full_end;
    u32x4 cv[NCV];
    read_ops(p.cbimg + (long)tile * TILE_Q + h * 32 + c, cv);
    f32x16 d[RT];
    tile_mfma(IC0{}, ICNM{}, cv, d);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cc = tile * kTileCodes + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (cc >= p.n) d[rt][r] = NEG_INF;
      }
      tpend[rt] = max_chain(tpend[rt], d[rt]);
    }
    close_group(tile);
  } else if (pending) {
    close_group(t_full_end - 1);
  }

  // ---- merge the two lane halves of each row, write one record ----
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float a1 = m1[rt], a2v = m2[rt], a3 = m3[rt], a4 = m4[rt];
    int j1 = i1[rt] * 2 + h, j2 = i2[rt] * 2 + h, j3 = i3[rt] * 2 + h;   // half-group ids
    if constexpr (MIXED || F16) {
      // One record per LANE HALF (the re-rank sees 2 nsplit "splits"): the wider margin of these filters makes "a fourth
      // group of one record within the margin" -- an undecided row: a block-wide scan of its record sets in the re-rank -- about
      // as likely as not per call with merged records (1 row in 16 384 at config 2); half the codes per record makes it
      // ~8x rarer, and the merge below is not needed.
      const int row = rowblk * (32 * WAVES * RT) + (wave * RT + rt) * 32 + c;
      if (row < p.rows) {
        const float rs = p.rowscale[row];      // back to true units: the row's coefficients were normalised by 2^-e_r
        p.rec[(long)(split * 2 + h) * p.rows + row] = make_rec(a1 * rs, a2v * rs, a3 * rs, a4 * rs, j1, j2, j3, 2 * (t_begin / GT));
      }
      continue;
    }
    if constexpr (TOP4) {
      const float b1 = __shfl_xor(a1, 32), b2 = __shfl_xor(a2v, 32), b3 = __shfl_xor(a3, 32), b4 = __shfl_xor(a4, 32);
      const int k1 = __shfl_xor(j1, 32), k2 = __shfl_xor(j2, 32), k3 = __shfl_xor(j3, 32);
      top4_insert(b1, k1, a1, a2v, a3, a4, j1, j2, j3);
      top4_insert(b2, k2, a1, a2v, a3, a4, j1, j2, j3);
      top4_insert(b3, k3, a1, a2v, a3, a4, j1, j2, j3);
      top4_insert_value(b4, a3, a4);
    } else {
      // top-3 tracker: the record's third slot stays empty and its fourth value is the third-largest maximum,
      // so the re-rank treats "third group within the margin" as undecided
      const float b1 = __shfl_xor(a1, 32), b2 = __shfl_xor(a2v, 32), b3 = __shfl_xor(a3, 32);
      const int k1 = __shfl_xor(j1, 32), k2 = __shfl_xor(j2, 32);
      top3_insert(b1, k1, a1, a2v, a3, j1, j2);
      top3_insert(b2, k2, a1, a2v, a3, j1, j2);
      a3 = __builtin_amdgcn_fmed3f(a2v, a3, b3);
      a4 = a3;
      a3 = -__builtin_inff();
      j3 = 0;
    }
    const int row = rowblk * (32 * WAVES * RT) + (wave * RT + rt) * 32 + c;
    if (h == 0 && row < p.rows) {
      p.rec[(long)split * p.rows + row] = make_rec(a1, a2v, a3, a4, j1, j2, j3, 2 * (t_begin / GT));
    }
  }
#ifdef GQHIP_CLOCK_STAMPS
  if (tid == 0 && blockIdx.x < 2048) {   // per-block timeline + placement, same record as gq_filter_kernel's
    unsigned long long *o = reinterpret_cast<unsigned long long *>(p.dbg) + 4 * blockIdx.x;
    o[0] = st_b0;
    o[1] = __builtin_amdgcn_s_memrealtime();
    o[2] = ((st_r0 - st_b0) << 32) | (st_r2 - st_b0);
    o[3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
           (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#endif
}

}  // namespace gqhip
