// gq_scores.h -- the compat op's score matrix on the fp32 matrix cores (reference: gq_cuda.cu:12-40, one thread per
// (row, code) pair: 16 fp32 divides each, VALU-bound; the restated per-pair kernel in gq_aux.h runs at ~1.5 TB/s).
//
//   out[r, j] = sum_i -((n_ji - mu_ri) / sd_ri)^2 + beta n_ji^2
//             = sum_i (beta - 1/sd^2) n_ji^2 + (2 mu / sd^2) n_ji   -   sum_i mu_ri^2 / sd_ri^2
// i.e. a [rows, 2 dim] x [2 dim, n] product plus a per-row constant: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains,
// 157 TFLOP/s) with the constant as the accumulators' initial value.  68.7 GFLOP at config 2 = 0.45 ms of matrix time
// against 0.78 ms to write the 4.29 GB matrix at ~5.5 TB/s: the kernel is bound by the HBM write stream, which is the
// roofline SURVEY.md 8(d) assigns to this op.
//
// Tiling: D[i = row][j = code] (A operand = row coefficients, B operand = codebook values / squares), so a lane's
// accumulator register holds one code column for 16 rows.  Block = 4 waves x RT row tiles of 32 rows; blockIdx % nsplit =
// code split (multiple of 8: each XCD's L2 streams 1/8 of the codebook), chunks of CT tiles staged through LDS exactly
// like gq_filter_kernel.  Tiles are processed in PAIRS so that a store instruction writes one 256-byte run (below).
// Where the time goes at 16 384 x 65 536 x dim 16 (4.29 GB; diagnostic builds and counters, profiles/r03/scores_ablation.txt,
// scores_pmc.txt, store_pattern.txt): the matrix work alone 0.55 ms (1.20 M shader cycles at 2.18 GHz, matrix pipes 88 % busy),
// the store stream alone 0.83-0.86 ms (2.02 M cycles at 2.35 GHz = 5.0-5.3 TB/s; a kernel that ONLY stores reaches 5.4-5.7 TB/s
// on this device whatever the pattern -- 64 rows x 256 B, 32 rows x 1 KiB in one dwordx4 instruction, whole rows -- and a
// torch fill 6.9), both together 1.05-1.09 ms = 1.91 M cycles: in CYCLES the two overlap completely (fewer cycles than the
// stores alone), but under fp32 MFMA load the chip holds 1.75 GHz instead of 2.35, and the store path moves ~2150 bytes per
// shader cycle either way.  Tried on top and measured within 4 %: dwordx4 stores through a wave-private LDS transpose, stores
// of the previous tile issued between the MFMAs of the next (ping-pong accumulators), one row tile per wave at 4 blocks per CU,
// waves of a block sharing rows, 16 ... 256 code splits.  What does help the store-bound dims (4 / 8: 61-63 % -> 66-67 %) is
// the rotation below: blocks that run together no longer write the same column phase of rows 256 KiB apart.
// Dims 16 and 32 run the same product as three fp16 products of two-term splits instead (gq_scores_f16.h: a fifth of the matrix
// cycles, 62-65 % of 8 TB/s at dim 16); this kernel serves dims 4 / 8 and GQHIP_SCORES=f32.
// The expansion differs from the per-pair formula by ~2^-24 * sum_i |terms| (cancellation between n^2/sd^2, mu n/sd^2 and
// mu^2/sd^2): relative to the score's own magnitude that is a few ulp, and the arg-max can differ from the per-pair
// formula's only at rounding ties -- the same caveat the CUDA kernel's own rounding carries (its bits cannot be pinned
// here: no nvcc).  GQHIP_SCORES=direct selects the per-pair kernel.
#pragma once
#include "gq_filter.h"

namespace gqhip {

struct ScoresParams {
  const float *mu, *sd, *cb;
  float *out;
  int rows, n;
  double beta;
  int nsplit, tiles_total, tiles_per_split;
  int rot;          // bit 0: chunk order rotated by the row block, bit 1: pair order rotated by the wave (see the kernel)
};

template <int DIM, int RT, int CT>
__global__ __launch_bounds__(256, 2) void gq_scores_mfma_kernel(const ScoresParams p) {
  constexpr int HD = DIM / 2;
  constexpr int TILE_F = kTileCodes * DIM;
  constexpr int CHUNK_F = CT * TILE_F;
  constexpr int R4 = CHUNK_F / 4 / 256;
  static_assert(R4 >= 1 && CHUNK_F % 1024 == 0, "chunk: multiple of 4 KiB");
  __shared__ __attribute__((aligned(16))) float lds[2][2][CHUNK_F];   // [buffer][0 = values, 1 = squares]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int split = blockIdx.x % p.nsplit;
  const int rowblk = blockIdx.x / p.nsplit;
  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.tiles_total);
  const long cb_last4 = (long)p.n * DIM - 4;

  // ---- row operands: A' = beta - 1/sd^2, B' = 2 mu / sd^2 for this lane's row (A operand of the MFMA: i = c, k = h) ----
  float coefA[RT][HD], coefB[RT][HD];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int row = rowblk * (128 * RT) + (wave * RT + rt) * 32 + c;
    row = min(row, p.rows - 1);
#pragma unroll
    for (int s = 0; s < HD; ++s) {
      const double sg = (double)p.sd[(long)row * DIM + h * HD + s];
      const double inv = 1.0 / (sg * sg);
      coefA[rt][s] = (float)(p.beta - inv);
      coefB[rt][s] = (float)(2.0 * (double)p.mu[(long)row * DIM + h * HD + s] * inv);
    }
  }
  // ---- per-row constant -sum mu^2/sd^2 for the 16 rows of this lane's accumulator registers ----
  // register r of lane (c, h) is row (r & 3) + 8 (r >> 2) + 4 h of the tile, column c
  f32x16 cinit[RT];
  __shared__ float s_const[4][RT][32];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    // lane (c, h) sums its half of the dims for row c; the two halves meet through a shuffle
    int row = rowblk * (128 * RT) + (wave * RT + rt) * 32 + c;
    row = min(row, p.rows - 1);
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < HD; ++s) {
      const double sg = (double)p.sd[(long)row * DIM + h * HD + s], m = (double)p.mu[(long)row * DIM + h * HD + s];
      acc += m * m / (sg * sg);
    }
    acc += __shfl_xor(acc, 32);
    if (h == 0) s_const[wave][rt][c] = (float)(-acc);
  }
  __syncthreads();
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[rt][r] = s_const[wave][rt][(r & 3) + 8 * (r >> 2) + 4 * h];

  // ---- chunk staging (as gq_filter_kernel) ----
  f32x4 stage[R4];
  auto load_chunk = [&](int tile0) {
    const long base = (long)tile0 * TILE_F;
#pragma unroll
    for (int r = 0; r < R4; ++r) {
      long off = base + (long)(tid + 256 * r) * 4;
      off = off < cb_last4 ? off : cb_last4;
      stage[r] = *reinterpret_cast<const f32x4 *>(p.cb + off);
    }
  };
  auto store_chunk = [&](int buf) {
    constexpr int SLOTS = DIM / 4;
#pragma unroll
    for (int r = 0; r < R4; ++r) {
      const int q = tid + 256 * r;
      const int code = q / SLOTS, slot = q % SLOTS;
      const int dst = (code * SLOTS + (slot ^ lds_swz<DIM>(code))) * 4;
      *reinterpret_cast<f32x4 *>(&lds[buf][0][dst]) = stage[r];
      *reinterpret_cast<f32x4 *>(&lds[buf][1][dst]) = stage[r] * stage[r];
    }
  };
  const int swz = lds_swz<DIM>(c);

  // Output addressing: register r of lane (c, h) is row (r & 3) + 8 (r >> 2) + 4 h of the wave's row tile, column
  // tile * 32 + c.  Everything wave-uniform (tile base, the register's row offset) stays in scalar registers; the lane
  // part is ONE 32-bit offset that advances by 32 columns per tile -- address VALU next to fp32 MFMAs is expensive.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const long tile_row0 = (long)rowblk * (128 * RT) + (long)wave_u * RT * 32;     // first row of the wave's RT tiles
  const bool all_rows = tile_row0 + 32 * RT <= p.rows;                          // wave-uniform
  const unsigned lane_off = (unsigned)(4 * h) * (unsigned)p.n + (unsigned)c;    // elements, < 32 * n

  const int ntiles = t_end - t_begin;
  const int nchunks = ntiles > 0 ? (ntiles + CT - 1) / CT : 0;
  // Rotation: every block of a split walks the same columns, and rows are n * 4 bytes apart (256 KiB at 65 536 codes) -- without
  // it all blocks that run together write the same few column phases at the same time, i.e. the same few memory channels.
  // Block rowblk starts at chunk rot0 of its split and wraps; inside a chunk wave w starts at pair w.
  const int rot0 = (p.rot & 1) && nchunks > 1 ? (int)((unsigned)rowblk * 5u % (unsigned)nchunks) : 0;
  auto chunk_of = [&](int ch) { const int k = ch + rot0; return k >= nchunks ? k - nchunks : k; };
  if (nchunks > 0) {
    load_chunk(t_begin + chunk_of(0) * CT);
    store_chunk(0);
  }
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int tile0 = t_begin + chunk_of(ch) * CT;
    if (ch + 1 < nchunks) load_chunk(t_begin + chunk_of(ch + 1) * CT);
    const int nt = min(CT, t_end - tile0);
    const float *val = lds[ch & 1][0] + c * DIM;
    const float *sq = lds[ch & 1][1] + c * DIM;
    // one tile's accumulators for the wave's RT row tiles
    auto compute = [&](int tt, f32x16 (&d)[RT]) {
      float a[HD], a2[HD];
      lds_read_half<DIM>(val + tt * TILE_F, h, swz, a);
      lds_read_half<DIM>(sq + tt * TILE_F, h, swz, a2);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) d[rt] = cinit[rt];
#if !(defined(GQHIP_ABL) && (GQHIP_ABL & 16))    // diagnostic build: no MFMAs (what the store stream alone takes)
#pragma unroll
      for (int s = 0; s < HD; ++s)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) d[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(coefA[rt][s], a2[s], d[rt], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < HD; ++s)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) d[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(coefB[rt][s], a[s], d[rt], 0, 0, 0);
#else
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) d[rt][0] += a[0] + a2[0];
#endif
    };
    int tt = 0;
    // Tile PAIRS: register r of tile A and of tile B hold the same two rows (ro, ro + 4: lane halves) for 32 codes each.
    // v_permlane32_swap exchanges A's upper half with B's lower half, so one register then holds 64 CONSECUTIVE codes of row
    // ro and the other those of row ro + 4: every store instruction writes one 256-byte run instead of two 128-byte runs in
    // two rows (measured with the matrix work compiled out: 5.0 -> ... TB/s; a plain fill of the buffer runs at 6.9).
    const int npairs = nt >> 1;
    const int prot = (p.rot & 2) ? wave_u % max(npairs, 1) : 0;
    for (int tp = 0; tp < npairs; ++tp) {
      tt = 2 * (tp + prot >= npairs ? tp + prot - npairs : tp + prot);
      f32x16 d0[RT], d1[RT];
      compute(tt, d0);
      compute(tt + 1, d1);
      const int code0 = (tile0 + tt) * kTileCodes;
#if defined(GQHIP_ABL) && (GQHIP_ABL & 32)       // diagnostic build: no stores (what the matrix work alone takes)
      if (d0[0][0] == 12345.678f && d1[RT - 1][15] == 0.5f)
#endif
      if (code0 + 2 * kTileCodes <= p.n) {   // wave-uniform: the swap below is a cross-lane operation
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          float *base = p.out + (tile_row0 + 32 * rt) * p.n + code0 + lane;     // row 0 of the tile, this lane's column
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0[rt][r]), __float_as_uint(d1[rt][r]), false, false);
            if (all_rows || tile_row0 + 32 * rt + ro < p.rows) base[(long)ro * p.n] = __uint_as_float(sw[0]);
            if (all_rows || tile_row0 + 32 * rt + ro + 4 < p.rows) base[(long)(ro + 4) * p.n] = __uint_as_float(sw[1]);
          }
        }
      } else if (code0 + c < p.n) {
        // the pair straddles the end of the codebook (n % 64 != 0): tile A alone, two 128-byte runs per register
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          float *base = p.out + (tile_row0 + 32 * rt) * p.n + code0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);
            if (all_rows || tile_row0 + 32 * rt + ro + 4 * h < p.rows) base[(long)ro * p.n + lane_off] = d0[rt][r];
          }
        }
        const int code1 = code0 + kTileCodes;
        if (code1 + c < p.n) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            float *base = p.out + (tile_row0 + 32 * rt) * p.n + code1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ro = (r & 3) + 8 * (r >> 2);
              if (all_rows || tile_row0 + 32 * rt + ro + 4 * h < p.rows) base[(long)ro * p.n + lane_off] = d1[rt][r];
            }
          }
        }
      }
    }
    for (tt = 2 * npairs; tt < nt; ++tt) {          // an odd tile at the end of the split
      f32x16 d[RT];
      compute(tt, d);
      const int code0 = (tile0 + tt) * kTileCodes;
#if defined(GQHIP_ABL) && (GQHIP_ABL & 32)
      if (d[0][0] == 12345.678f && d[RT - 1][15] == 0.5f)
#endif
      if (code0 + c < p.n) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          float *base = p.out + (tile_row0 + 32 * rt) * p.n + code0;            // wave-uniform
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);                               // + 4 h in lane_off
            if (all_rows || tile_row0 + 32 * rt + ro + 4 * h < p.rows)
              base[(long)ro * p.n + lane_off] = d[rt][r];
          }
        }
      }
    }
    if (ch + 1 < nchunks) store_chunk((ch + 1) & 1);
    __syncthreads();
  }
}

}  // namespace gqhip
