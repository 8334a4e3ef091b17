// gq_filter.h -- the MFMA filter kernel (hot kernel of the path).
//
// For every row r and every code j it evaluates, on the fp32 matrix cores,
//     f(r, j) = sum_i  A[r,i] * c[j,i]^2  +  B[r,i] * c[j,i]
// which equals the reference score (pit/quantization/gaussian.py:142-147) up
// to a per-row constant when  A = beta/2 - 1/(2 sd^2),  B = mu / sd^2
// (and equals -|z - e_j|^2 + |z|^2 for VQ with A = -1, B = 2 z; vq.py:58-69).
// The score matrix never leaves registers: each lane keeps the three largest
// half-tile maxima it has seen (+ the ids of the best two); the re-rank kernel
// (gq_rerank.h) re-evaluates those few half-tiles in the reference's exact
// operation order, which is what makes the indices bit-identical.
//
// Tiling (wave64, v_mfma_f32_32x32x2_f32):
//   D[i = code in tile][j = row in tile] ; lane l: c = l & 31, h = l >> 5
//   A operand, k-step s : codebook value  cb[tile*32 + c][h*HD + s]   (squared for s < HD)
//   B operand, k-step s : row coefficient A|B[row c][h*HD + s]
//   D regs of lane (c,h): row c, codes (reg&3) + 8*(reg>>2) + 4*h  -> "half-tile" h
// so all 16 accumulator registers of a lane belong to ONE row and the running
// maximum is a v_max3 chain with no cross-lane traffic in the loop.
// A block = 4 waves x RT row tiles (128*RT rows); the code axis is split
// `nsplit` ways over blockIdx so that blockIdx % 8 (the XCD a block lands on)
// selects the code split: each XCD's L2 only ever holds 1/8 of the codebook.
// Codebook chunks of CT tiles are staged through LDS (double buffered,
// register-staged 16-byte loads) and shared by the block's 4 waves.
#pragma once
#include "gq_common.h"

namespace gqhip {

enum FilterMode { kModeGQ = 0, kModeVQ = 1 };

struct FilterParams {
  const float *mu;      // [rows, DIM]   (VQ: z)
  const float *sd;      // [rows, DIM]   (VQ: unused)
  const float *cb;      // [n, DIM]
  Rec *rec;             // [nsplit, rows]
  int rows, n;
  float beta;
  int nsplit, tiles_total, tiles_per_split;
};

template <int HD>
__device__ __forceinline__ void lds_read_half(const float *p, float (&a)[HD]) {
  if constexpr (HD >= 4) {
#pragma unroll
    for (int q = 0; q < HD / 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(p + 4 * q);
      a[4 * q + 0] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
  } else {
    const f32x2 v = *reinterpret_cast<const f32x2 *>(p);
    a[0] = v.x; a[1] = v.y;
  }
}

__device__ __forceinline__ float max16(const f32x16 &d) {
  float t = __builtin_fmaxf(__builtin_fmaxf(d[0], d[1]), d[2]);
  t = __builtin_fmaxf(__builtin_fmaxf(t, d[3]), d[4]);
  t = __builtin_fmaxf(__builtin_fmaxf(t, d[5]), d[6]);
  t = __builtin_fmaxf(__builtin_fmaxf(t, d[7]), d[8]);
  t = __builtin_fmaxf(__builtin_fmaxf(t, d[9]), d[10]);
  t = __builtin_fmaxf(__builtin_fmaxf(t, d[11]), d[12]);
  t = __builtin_fmaxf(__builtin_fmaxf(t, d[13]), d[14]);
  return __builtin_fmaxf(t, d[15]);
}

// One 32-code tile against the wave's RT row tiles: RT independent MFMA chains
// are issued first, the max/top-3 epilogues follow (chain rt+1 runs on the
// matrix pipe while the VALU reduces chain rt).  MASKED handles the single
// partial tile at the end of the codebook (codes >= n score -inf).
template <int DIM, int RT, bool MASKED>
__device__ __forceinline__ void tile_step(const float (&a)[DIM / 2], const float (&coefA)[RT][DIM / 2],
                                          const float (&coefB)[RT][DIM / 2], int tile, int n, int h,
                                          float (&m1)[RT], float (&m2)[RT], float (&m3)[RT],
                                          int (&i1)[RT], int (&i2)[RT]) {
  constexpr int HD = DIM / 2;
  float a2[HD];
#pragma unroll
  for (int s = 0; s < HD; ++s) a2[s] = a[s] * a[s];
  f32x16 d[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    d[rt] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < HD; ++s)
      d[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], coefA[rt][s], d[rt], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < HD; ++s)
      d[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], coefB[rt][s], d[rt], 0, 0, 0);
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    if constexpr (MASKED) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int code = tile * kTileCodes + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (code >= n) d[rt][r] = -__builtin_inff();
      }
    }
    const float t = max16(d[rt]);
    top3_insert(t, tile, m1[rt], m2[rt], m3[rt], i1[rt], i2[rt]);
  }
}

template <int DIM, int RT, int CT, int MODE>
__global__ __launch_bounds__(256, 2) void gq_filter_kernel(const FilterParams p) {
  constexpr int HD = DIM / 2;                 // dims per lane half
  constexpr int TILE_F = kTileCodes * DIM;    // floats per 32-code tile
  constexpr int CHUNK_F = CT * TILE_F;        // floats per LDS chunk
  constexpr int R4 = CHUNK_F / 4 / 256;       // 16-byte loads per thread per chunk
  static_assert(R4 >= 1 && CHUNK_F % 1024 == 0, "chunk must be a multiple of 4 KiB");
  __shared__ __attribute__((aligned(16))) float lds[2][CHUNK_F];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int split = blockIdx.x % p.nsplit;
  const int rowblk = blockIdx.x / p.nsplit;
  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.tiles_total);
  const int t_full_end = min(t_end, p.n / kTileCodes);   // complete tiles only
  const long cb_last4 = (long)p.n * DIM - 4;  // last valid 16-byte load offset (floats)

  // ---- row coefficients (B operands), fixed for the whole kernel ----------
  float coefA[RT][HD], coefB[RT][HD];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int row = rowblk * (128 * RT) + (wave * RT + rt) * 32 + c;
    row = min(row, p.rows - 1);
    const float *pm = p.mu + (long)row * DIM + h * HD;
#pragma unroll
    for (int s = 0; s < HD; ++s) {
      if constexpr (MODE == kModeGQ) {
        const double sg = (double)p.sd[(long)row * DIM + h * HD + s];
        const double inv = 1.0 / (sg * sg);
        coefA[rt][s] = (float)(0.5 * (double)p.beta - 0.5 * inv);
        coefB[rt][s] = (float)((double)pm[s] * inv);
      } else {
        coefA[rt][s] = -1.0f;
        coefB[rt][s] = 2.0f * pm[s];
      }
    }
  }

  const float NEG_INF = -__builtin_inff();
  float m1[RT], m2[RT], m3[RT];
  int i1[RT], i2[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    m1[rt] = m2[rt] = m3[rt] = NEG_INF;
    i1[rt] = i2[rt] = 0;
  }

  // ---- chunk staging -------------------------------------------------------
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
#pragma unroll
    for (int r = 0; r < R4; ++r)
      *reinterpret_cast<f32x4 *>(&lds[buf][(tid + 256 * r) * 4]) = stage[r];
  };

  const int ntiles = t_full_end - t_begin;
  const int nchunks = ntiles > 0 ? (ntiles + CT - 1) / CT : 0;
  if (nchunks > 0) {
    load_chunk(t_begin);
    store_chunk(0);
  }
  __syncthreads();

  for (int ch = 0; ch < nchunks; ++ch) {
    const int tile0 = t_begin + ch * CT;
    if (ch + 1 < nchunks) load_chunk(tile0 + CT);
    const int nt = min(CT, t_full_end - tile0);
    const float *buf = lds[ch & 1] + c * DIM + h * HD;
    for (int tt = 0; tt < nt; ++tt) {
      float a[HD];
      lds_read_half<HD>(buf + tt * TILE_F, a);
      tile_step<DIM, RT, false>(a, coefA, coefB, tile0 + tt, p.n, h, m1, m2, m3, i1, i2);
    }
    if (ch + 1 < nchunks) store_chunk((ch + 1) & 1);
    __syncthreads();
  }

  // ---- the one partial tile (n % 32 != 0), straight from global ------------
  if (t_end > t_full_end) {
    const int tile = t_full_end;
    const int code = min(tile * kTileCodes + c, p.n - 1);
    float a[HD];
#pragma unroll
    for (int s = 0; s < HD; ++s) a[s] = p.cb[(long)code * DIM + h * HD + s];
    tile_step<DIM, RT, true>(a, coefA, coefB, tile, p.n, h, m1, m2, m3, i1, i2);
  }

  // ---- merge the two lane halves of each row, write one record -------------
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float a1 = m1[rt], a2v = m2[rt], a3 = m3[rt];
    int j1 = i1[rt] * 2 + h, j2 = i2[rt] * 2 + h;
    const float b1 = __shfl_xor(a1, 32), b2 = __shfl_xor(a2v, 32), b3 = __shfl_xor(a3, 32);
    const int k1 = __shfl_xor(j1, 32), k2 = __shfl_xor(j2, 32);
    top3_insert(b1, k1, a1, a2v, a3, j1, j2);
    top3_insert(b2, k2, a1, a2v, a3, j1, j2);
    top3_insert_value(b3, a2v, a3);
    const int row = rowblk * (128 * RT) + (wave * RT + rt) * 32 + c;
    if (h == 0 && row < p.rows) {
      Rec r;
      r.m1 = a1; r.m2 = a2v; r.m3 = a3; r.id1 = j1; r.id2 = j2;
      r.pad[0] = r.pad[1] = r.pad[2] = 0;
      p.rec[(long)split * p.rows + row] = r;
    }
  }
}

}  // namespace gqhip
