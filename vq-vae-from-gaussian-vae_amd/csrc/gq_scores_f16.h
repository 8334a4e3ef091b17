// gq_scores_f16.h -- the compat op's score matrix (gq_cuda.cu:12-40) for dims 16 and 32 on the fp16 matrix cores.
//
// gq_scores.h computes out = [A' | B'] x [n^2 | n] + C on v_mfma_f32_32x32x2_f32.  At dims 4 / 8 that kernel is bound by its
// store stream (61-65 % of 8 TB/s); at dims 16 / 32 its 32 / 64 fp32 MFMAs per tile cost more than the stores and, worse, pull
// the clock from 2.35 to 1.75 GHz while the store path moves a fixed number of bytes per cycle (profiles/r03/scores_pmc.txt).
// Here the same product runs as THREE fp16 products of two-term splits with fp32 accumulation (x = x_h + x_l, x_h = fp16(x),
// x_l = fp16(x - x_h): 22 significant bits per operand; x_h y_h + x_h y_l + x_l y_h, the dropped x_l y_l <= 2^-22 |x y|): 6 / 12
// v_mfma_f32_32x32x16_f16 of 8 passes per tile instead of 32 / 64 fp32 MFMAs of 16 -- a fifth of the matrix cycles.
//
// Ranges.  Row side: the 2 dim coefficients of a row are multiplied by 2^e_r (exact) so that the largest lies in [2^13, 2^14);
// the accumulator is multiplied by 2^-e_r (exact) when the row constant is added.  Code side: n and n^2 go in unscaled, which
// needs |n| <= 255 (n^2 finite in fp16); values whose split would reach fp16's subnormal range (|x| < 2^-3) keep an ABSOLUTE
// error <= 2^-25 instead of a relative one.  A chunk of codes that holds a value outside [-255, 255] (or a non-finite one) is
// computed by the per-pair formula instead (block-uniform branch; never taken for a prior-sample codebook, |n| <= 4.6).
// Error against the per-pair formula: ~3 2^-22 sum_i |terms| from the splits plus the fp32 accumulation, the same order as
// gq_scores.h's fp32 chain (its test tolerance is unchanged).  GQHIP_SCORES=f32 selects that kernel for these dims too.
#pragma once
#include "gq_scores.h"

namespace gqhip {

template <int DIM, int RT, int CT>
__global__ __launch_bounds__(256, 2) void gq_scores_f16x3_kernel(const ScoresParams p) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  static_assert(DIM == 16 || DIM == 32, "dims 16 and 32 (one or two k-steps per operand half)");
  constexpr int NS = (2 * DIM) / 16;          // MFMA k-steps per product: [n^2 | n] has 2 DIM elements
  constexpr int TILE_B = 2 * NS * 64 * 16;    // bytes of a tile's operand image: [plane h / l][k-step][lane half][code] x 16
  constexpr int CHUNK_B = CT * TILE_B;
  constexpr int TILE_F = kTileCodes * DIM;    // codebook floats per tile
  constexpr int CHUNK_F = CT * TILE_F;
  constexpr int R4 = CHUNK_F / 4 / 256;
  static_assert(R4 >= 1 && CHUNK_F % 1024 == 0, "chunk: multiple of 4 KiB of codebook");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][CHUNK_B];
  __shared__ __attribute__((aligned(16))) float s_const[4][RT][32], s_inv[4][RT][32];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int split = blockIdx.x % p.nsplit;
  const int rowblk = blockIdx.x / p.nsplit;
  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.tiles_total);
  const long cb_last4 = (long)p.n * DIM - 4;

  // ---- row operands: element e = 16 s + 8 h + t of [A' | B'], A' = beta - 1/sd^2, B' = 2 mu / sd^2, scaled and split ----
  f16x8 ah[RT][NS], al[RT][NS];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int row = rowblk * (128 * RT) + (wave * RT + rt) * 32 + c;
    row = min(row, p.rows - 1);
    double co[NS][8];
    double amax = 0.0, csum = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int e = 16 * s + 8 * h + t;               // compile-time but for h
        const int d = e < DIM ? e : e - DIM;            // e < DIM is a function of s alone (DIM is a multiple of 16)
        const double sg = (double)p.sd[(long)row * DIM + d], m = (double)p.mu[(long)row * DIM + d];
        const double inv = 1.0 / (sg * sg);
        co[s][t] = 16 * s < DIM ? p.beta - inv : 2.0 * m * inv;
        if (16 * s < DIM) csum += m * m * inv;          // every dim exactly once: the A' half
        const double a = co[s][t] < 0.0 ? -co[s][t] : co[s][t];
        amax = a > amax ? a : amax;                     // NaN never wins: handled by the finite test below
      }
    amax = fmax(amax, __shfl_xor(amax, 32));
    csum += __shfl_xor(csum, 32);
    int e_r = 0;
    if (amax > 0.0 && amax < 1.0e300) {
      e_r = 13 - ilogb(amax);
      e_r = e_r < -120 ? -120 : (e_r > 120 ? 120 : e_r);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const double v = ldexp(co[s][t], e_r);
        const _Float16 hi = (_Float16)(float)v;
        ah[rt][s][t] = hi;
        al[rt][s][t] = (_Float16)(float)(v - (double)(float)hi);
      }
    if (h == 0) {
      s_const[wave][rt][c] = (float)(-csum);
      s_inv[wave][rt][c] = (float)ldexp(1.0, -e_r);
    }
  }
  __syncthreads();
  // register r of lane (c, h) is row (r & 3) + 8 (r >> 2) + 4 h of the tile, column c.  The 32 per-row constants of a lane stay in
  // registers at dim 16; at dim 32 the row operands take twice the registers and the constants are re-read from LDS per tile
  // (four broadcast ds_read_b128 each).
  constexpr bool CREG = DIM == 16;
  f32x16 cinit[CREG ? RT : 1], cinv[CREG ? RT : 1];
  if constexpr (CREG) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        cinit[rt][r] = s_const[wave][rt][(r & 3) + 8 * (r >> 2) + 4 * h];
        cinv[rt][r] = s_inv[wave][rt][(r & 3) + 8 * (r >> 2) + 4 * h];
      }
  }

  // ---- chunk staging: fp32 codebook -> [n^2 | n] split into fp16 h / l, in MFMA operand order ----
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
  auto store_chunk = [&](int buf, int ntl) -> int {   // != 0: one of the chunk's ntl tiles holds a value the fp16 image cannot hold
    int bad = 0;
#pragma unroll
    for (int r = 0; r < R4; ++r) {
      const int f0 = (tid + 256 * r) * 4;                 // first float of this thread's quad inside the chunk
      const int code = f0 / DIM, d0 = f0 % DIM;           // 4 consecutive dims of one code
      const int tile = code >> 5, cc = code & 31;
      const f32x4 v = stage[r], sq = v * v;
#pragma unroll
      for (int k = 0; k < 4; ++k) bad |= tile < ntl && !(__builtin_fabsf(v[k]) <= 255.f);   // tiles past the split's end: loaded, never used
#pragma unroll
      for (int part = 0; part < 2; ++part) {              // 0: squares (elements d0 ..), 1: values (elements DIM + d0 ..)
        const f32x4 x = part ? v : sq;
        const int e = part * DIM + d0;
        const int s = e >> 4, hh = (e >> 3) & 1, t = e & 7;
        f16x4 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          hi[k] = (_Float16)x[k];
          lo[k] = (_Float16)(x[k] - (float)hi[k]);
        }
        unsigned char *dst = &lds[buf][tile * TILE_B + ((s * 2 + hh) * 32 + cc) * 16 + 2 * t];
        *reinterpret_cast<f16x4 *>(dst) = hi;
        *reinterpret_cast<f16x4 *>(dst + NS * 1024) = lo;     // plane l: NS k-steps x 2 halves x 32 codes x 16 bytes further
      }
    }
    return bad;
  };

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const long tile_row0 = (long)rowblk * (128 * RT) + (long)wave_u * RT * 32;
  const bool all_rows = tile_row0 + 32 * RT <= p.rows;
  const unsigned lane_off = (unsigned)(4 * h) * (unsigned)p.n + (unsigned)c;

  // one tile: three products, fp32 accumulation from zero; then * 2^-e_r + C_r
  auto compute = [&](const unsigned char *T, f32x16 (&d)[RT]) {
    f16x8 chh[NS], cll[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      chh[s] = *reinterpret_cast<const f16x8 *>(T + ((s * 2 + h) * 32 + c) * 16);
      cll[s] = *reinterpret_cast<const f16x8 *>(T + NS * 1024 + ((s * 2 + h) * 32 + c) * 16);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x16 a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#if !(defined(GQHIP_ABL) && (GQHIP_ABL & 16))    // diagnostic build: no MFMAs (what the store stream alone takes)
#pragma unroll
      for (int s = 0; s < NS; ++s) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rt][s], chh[s], a, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < NS; ++s) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rt][s], cll[s], a, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < NS; ++s) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[rt][s], chh[s], a, 0, 0, 0);
#else
      a[0] = (float)chh[0][0] + (float)cll[NS - 1][7] + (float)ah[rt][0][0] + (float)al[rt][NS - 1][7];
#endif
      if constexpr (CREG) {
#pragma unroll
        for (int r = 0; r < 16; ++r) d[rt][r] = __builtin_fmaf(a[r], cinv[rt][r], cinit[rt][r]);
      } else {
        int off = 4 * h;
        asm volatile("" : "+v"(off));     // opaque per tile: hipcc would otherwise hoist these 32 loop-invariant values into registers
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 ci = *reinterpret_cast<const f32x4 *>(&s_const[wave_u][rt][8 * q + off]);
          const f32x4 cv = *reinterpret_cast<const f32x4 *>(&s_inv[wave_u][rt][8 * q + off]);
#pragma unroll
          for (int k = 0; k < 4; ++k) d[rt][4 * q + k] = __builtin_fmaf(a[4 * q + k], cv[k], ci[k]);
        }
      }
    }
  };
  const int ntiles = t_end - t_begin;
  const int nchunks = ntiles > 0 ? (ntiles + CT - 1) / CT : 0;
  const int rot0 = (p.rot & 1) && nchunks > 1 ? (int)((unsigned)rowblk * 5u % (unsigned)nchunks) : 0;
  auto chunk_of = [&](int ch) { const int k = ch + rot0; return k >= nchunks ? k - nchunks : k; };
  int bad_cur = 0, any_bad = 0;
  if (nchunks > 0) {
    load_chunk(t_begin + chunk_of(0) * CT);
    bad_cur = store_chunk(0, min(CT, t_end - (t_begin + chunk_of(0) * CT)));
  }
  bad_cur = __syncthreads_or(bad_cur);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int tile0 = t_begin + chunk_of(ch) * CT;
    if (ch + 1 < nchunks) load_chunk(t_begin + chunk_of(ch + 1) * CT);
    const int nt = min(CT, t_end - tile0);
    const unsigned char *T0 = lds[ch & 1];
    int tt = 0;
    any_bad |= bad_cur;
    const int npairs = bad_cur ? 0 : (nt >> 1);      // a chunk the fp16 image cannot hold is left to the second pass below
    const int prot = (p.rot & 2) ? wave_u % max(npairs, 1) : 0;
    for (int tp = 0; tp < npairs; ++tp) {
      tt = 2 * (tp + prot >= npairs ? tp + prot - npairs : tp + prot);
      f32x16 d0[RT], d1[RT];
      compute(T0 + tt * TILE_B, d0);
      compute(T0 + (tt + 1) * TILE_B, d1);
      const int code0 = (tile0 + tt) * kTileCodes;
#if defined(GQHIP_ABL) && (GQHIP_ABL & 32)       // diagnostic build: no stores (what the matrix work alone takes)
      if (d0[0][0] == 12345.678f && d1[RT - 1][15] == 0.5f)
#endif
      if (code0 + 2 * kTileCodes <= p.n) {   // wave-uniform: the swap below is a cross-lane operation (see gq_scores.h)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          float *base = p.out + (tile_row0 + 32 * rt) * p.n + code0 + lane;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0[rt][r]), __float_as_uint(d1[rt][r]), false, false);
            if (all_rows || tile_row0 + 32 * rt + ro < p.rows) base[(long)ro * p.n] = __uint_as_float(sw[0]);
            if (all_rows || tile_row0 + 32 * rt + ro + 4 < p.rows) base[(long)(ro + 4) * p.n] = __uint_as_float(sw[1]);
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int cu = code0 + u * kTileCodes;
          if (cu + c < p.n) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
              float *base = p.out + (tile_row0 + 32 * rt) * p.n + cu;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                if (all_rows || tile_row0 + 32 * rt + ro + 4 * h < p.rows) base[(long)ro * p.n + lane_off] = u ? d1[rt][r] : d0[rt][r];
              }
            }
          }
        }
      }
    }
    for (tt = bad_cur ? nt : 2 * npairs; tt < nt; ++tt) {          // an odd tile at the end of the split
      f32x16 d[RT];
      compute(T0 + tt * TILE_B, d);
      const int code0 = (tile0 + tt) * kTileCodes;
      if (code0 + c < p.n) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          float *base = p.out + (tile_row0 + 32 * rt) * p.n + code0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);
            if (all_rows || tile_row0 + 32 * rt + ro + 4 * h < p.rows) base[(long)ro * p.n + lane_off] = d[rt][r];
          }
        }
      }
    }
    int bad_next = 0;
    if (ch + 1 < nchunks) bad_next = store_chunk((ch + 1) & 1, min(CT, t_end - (t_begin + chunk_of(ch + 1) * CT)));
    bad_cur = __syncthreads_or(bad_next);
  }
  // ---- second pass (block-uniform, never taken for a prior-sample codebook): chunks with a value outside fp16's range, by the
  //      per-pair formula of gq_cuda.cu:31-38; element (register r, lane) as in the MFMA layout, nothing unrolled ----
  if (any_bad) {
    for (int ch = 0; ch < nchunks; ++ch) {
      const int tile0 = t_begin + ch * CT;
      const int nt = min(CT, t_end - tile0);
      int bad = 0;
      for (long f = (long)tile0 * TILE_F + tid; f < (long)(tile0 + nt) * TILE_F && f < (long)p.n * DIM; f += 256)
        bad |= !(__builtin_fabsf(p.cb[f]) <= 255.f);
      if (!__syncthreads_or(bad)) continue;
#pragma unroll 1
      for (int tt = 0; tt < nt; ++tt) {
        const int code = (tile0 + tt) * kTileCodes + c;
        if (code >= p.n) continue;
#pragma unroll 1
        for (int k = 0; k < RT * 16; ++k) {
          const long row = tile_row0 + 32 * (k >> 4) + (k & 3) + 8 * ((k & 15) >> 2) + 4 * h;
          if (row >= p.rows) continue;
          float acc = 0.f;
#pragma unroll 1
          for (int i = 0; i < DIM; ++i) {
            const float co = p.cb[(long)code * DIM + i];
            const float iv = (co - p.mu[row * DIM + i]) / p.sd[row * DIM + i];
            acc -= iv * iv;
            acc = (float)((double)acc + (double)(co * co) * p.beta);
          }
          p.out[row * p.n + code] = acc;
        }
      }
    }
  }
}

}  // namespace gqhip
