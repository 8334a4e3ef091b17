// gq_filter.h -- the MFMA filter kernel (hot kernel of the path).
//
// For every row r and every code j it evaluates, on the fp32 matrix cores,
//     f(r, j) = sum_i  A[r,i] * c[j,i]^2  +  B[r,i] * c[j,i]
// which equals the reference score (pit/quantization/gaussian.py:142-147) up
// to a per-row constant when  A = beta/2 - 1/(2 sd^2),  B = mu / sd^2
// (and equals -|z - e_j|^2 + |z|^2 for VQ with A = -1, B = 2 z; vq.py:58-69).
// The score matrix never leaves registers: each lane keeps the four largest
// "half-group" maxima it has seen (+ the ids of the best three); the re-rank kernel
// (gq_rerank.h) re-evaluates those few half-groups in the reference's exact
// operation order, which is what makes the indices bit-identical.
//
// Tiling (wave64, v_mfma_f32_32x32x2_f32):
//   D[i = code in tile][j = row in tile] ; lane l: c = l & 31, h = l >> 5
//   A operand, k-step s : codebook value  cb[tile*32 + c][h*HD + s]   (its square for s < HD)
//   B operand, k-step s : row coefficient A|B[row c][h*HD + s]
//   D regs of lane (c,h): row c, codes (reg&3) + 8*(reg>>2) + 4*h of the tile
// so all 16 accumulator registers of a lane belong to ONE row and the running
// maximum is a v_max3 chain with no cross-lane traffic in the loop.  A half-group
// (group p of GT tiles, half h) = those 16 codes in tiles GT*p .. GT*p+GT-1 -> 16*GT codes
// (GT = 2, "half-pair", for dim >= 16; GT = 4 for dim <= 8 where a tile has few MFMAs).
// A block = 4 waves x RT row tiles (128*RT rows); the code axis is split
// `nsplit` ways over blockIdx so that blockIdx % 8 (the XCD a block lands on)
// selects the code split: each XCD's L2 only ever holds 1/8 of the codebook.
// Codebook chunks of CT tiles are staged through LDS (double buffered,
// register-staged 16-byte loads; the squares are formed once per chunk while
// staging) and shared by the block's 4 waves.
#pragma once
#include <type_traits>

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
  WsHeader *hdr;
  void *dbg;            // diagnostic builds only
};

// LDS image of a chunk: code rows of DIM floats = DIM/4 slots of 16 bytes.  Row-major rows of
// 32/64/128 bytes make the lanes of a ds_read_b128 group hit the same banks (4-way at DIM 16);
// slot s of code c is therefore stored at slot  s ^ swz(c)  (an involution, applied by the
// register-staged writer and by the readers), which spreads a lane group over all 64 banks.
template <int DIM>
__device__ __forceinline__ int lds_swz(int code) {
  if constexpr (DIM == 32) return (code >> 1) & 7;
  else if constexpr (DIM == 16) return (code >> 2) & 3;
  else if constexpr (DIM == 8) return (code >> 3) & 1;
  else return 0;
}

// the HD floats of lane half h of code row `c` (row base `row` = tile base + c*DIM floats)
template <int DIM>
__device__ __forceinline__ void lds_read_half(const float *row, int h, int x, float (&a)[DIM / 2]) {
  constexpr int HD = DIM / 2;
  if constexpr (HD >= 4) {
#pragma unroll
    for (int q = 0; q < HD / 4; ++q) {
      const int slot = (h * (HD / 4) + q) ^ x;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(row + 4 * slot);
      a[4 * q + 0] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
  } else {
    const f32x2 v = *reinterpret_cast<const f32x2 *>(row + h * HD);
    a[0] = v.x; a[1] = v.y;
  }
}

// Running maximum over the 16 accumulators of a lane, seeded with t0: 8 v_max3.
__device__ __forceinline__ float max_chain(float t0, const f32x16 &d) {
  float t = __builtin_fmaxf(__builtin_fmaxf(t0, d[0]), d[1]);
  t = __builtin_fmaxf(__builtin_fmaxf(t, d[2]), d[3]);
  t = __builtin_fmaxf(__builtin_fmaxf(t, d[4]), d[5]);
  t = __builtin_fmaxf(__builtin_fmaxf(t, d[6]), d[7]);
  t = __builtin_fmaxf(__builtin_fmaxf(t, d[8]), d[9]);
  t = __builtin_fmaxf(__builtin_fmaxf(t, d[10]), d[11]);
  t = __builtin_fmaxf(__builtin_fmaxf(t, d[12]), d[13]);
  return __builtin_fmaxf(__builtin_fmaxf(t, d[14]), d[15]);
}

// The same maximum as a tree: 8 v_max3 in 3 dependent levels instead of 8 (max is exact and associative, so the value
// is the same; a lane's 8-deep dependent chain was ~60 cycles of latency per tile once the MFMA work per tile shrank to
// one or two instructions).
__device__ __forceinline__ float max_tree(float t0, const f32x16 &d) {
  const float a = __builtin_fmaxf(__builtin_fmaxf(d[0], d[1]), d[2]);
  const float b = __builtin_fmaxf(__builtin_fmaxf(d[3], d[4]), d[5]);
  const float c = __builtin_fmaxf(__builtin_fmaxf(d[6], d[7]), d[8]);
  const float e = __builtin_fmaxf(__builtin_fmaxf(d[9], d[10]), d[11]);
  const float f = __builtin_fmaxf(__builtin_fmaxf(d[12], d[13]), d[14]);
  const float g = __builtin_fmaxf(__builtin_fmaxf(a, b), c);
  const float h = __builtin_fmaxf(__builtin_fmaxf(e, f), d[15]);
  return __builtin_fmaxf(__builtin_fmaxf(g, h), t0);
}

// MFMA k-steps [S0, S1) of one 32-code tile against the wave's RT row tiles
// (k-step s < HD multiplies the squares by A, s >= HD the values by B).  The RT
// chains alternate in program order (A1 B1 A2 B2 ...) so a chain's next MFMA is
// never issued back-to-back with its predecessor.
template <int DIM, int RT, int S0, int S1>
__device__ __forceinline__ void tile_mfma(const float (&a)[DIM / 2], const float (&a2)[DIM / 2],
                                          const float (&coefA)[RT][DIM / 2],
                                          const float (&coefB)[RT][DIM / 2], f32x16 (&d)[RT]) {
  constexpr int HD = DIM / 2;
  if constexpr (S0 == 0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      d[rt] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int s = S0; s < S1; ++s)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (s < HD)
        d[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], coefA[rt][s], d[rt], 0, 0, 0);
      else
        d[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s - HD], coefB[rt][s - HD], d[rt], 0, 0, 0);
    }
}

// fp32 MFMA and VALU share the SIMD's fp32 datapath on gfx950 (measured: every VALU
// instruction next to a v_mfma_f32_32x32x2_f32 stream costs ~4.4 matrix-pipe cycles,
// tools/mfma_peak.hip), so the epilogue is kept to the minimum instruction count:
// 8 v_max3 per 16-MFMA chain, and ONE top-4 update per GROUP of GT tiles (a candidate
// "half-group" = the GT x 16 codes a lane saw in tiles GT*p .. GT*p+GT-1).  The main loop is
// software pipelined so that nothing but those VALU instructions ever keeps the
// matrix pipe waiting: the LDS operands of tile t+1 are fetched before the MFMAs of
// tile t, and the epilogue of tile t-1 runs as ONE cluster right after the first
// MFMAs of tile t (no wait for the accumulators to drain; each MFMA<->VALU switch
// costs ~9 cycles, so the VALU work is clustered, not spread).
template <int DIM, int RT, int CT, int MODE, int GT>
__device__ __forceinline__ void filter_block(const FilterParams &p, const int vblock) {
  constexpr int HD = DIM / 2;                 // dims per lane half
  constexpr int TILE_F = kTileCodes * DIM;    // floats per 32-code tile
  constexpr int CHUNK_F = CT * TILE_F;        // floats per LDS chunk (per array)
  constexpr int R4 = CHUNK_F / 4 / 256;       // 16-byte loads per thread per chunk
  static_assert(R4 >= 1 && CHUNK_F % 1024 == 0 && CT % GT == 0, "chunk: multiple of 4 KiB, whole tile groups");
  // [buffer][0 = values, 1 = squares]
  __shared__ __attribute__((aligned(16))) float lds[2][2][CHUNK_F];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int split = vblock % p.nsplit;
  const int rowblk = vblock / p.nsplit;
  const int t_begin = split * p.tiles_per_split;          // multiple of GT (tiles_per_split is)
  const int t_end = min(t_begin + p.tiles_per_split, p.tiles_total);
  const int t_full_end = min(t_end, p.n / kTileCodes);    // complete tiles only
  const long cb_last4 = (long)p.n * DIM - 4;              // last valid 16-byte load offset (floats)

#ifdef GQHIP_CLOCK_STAMPS
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int nrows = p.rows;

  // ---- row coefficients (B operands), fixed for the whole kernel ----------
  float coefA[RT][HD], coefB[RT][HD];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int row = rowblk * (128 * RT) + (wave * RT + rt) * 32 + c;
    row = min(row, nrows - 1);
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
  float m1[RT], m2[RT], m3[RT], m4[RT], tpend[RT];
  int i1[RT], i2[RT], i3[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    m1[rt] = m2[rt] = m3[rt] = m4[rt] = tpend[rt] = NEG_INF;
    i1[rt] = i2[rt] = i3[rt] = 0;
  }

  // ---- chunk staging: global -> registers -> LDS (values and their squares) ----
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
    constexpr int SLOTS = DIM / 4;              // 16-byte slots per code row
#pragma unroll
    for (int r = 0; r < R4; ++r) {
      const int q = tid + 256 * r;              // linear slot index within the chunk
      const int code = q / SLOTS, slot = q % SLOTS;
      const int dst = (code * SLOTS + (slot ^ lds_swz<DIM>(code))) * 4;
      *reinterpret_cast<f32x4 *>(&lds[buf][0][dst]) = stage[r];
      *reinterpret_cast<f32x4 *>(&lds[buf][1][dst]) = stage[r] * stage[r];
    }
  };
  const int swz = lds_swz<DIM>(c);              // tiles start at multiples of 32 codes: swz(code) = swz(c)
  auto read_ops = [&](const float *val, const float *sq, float (&a)[HD], float (&a2)[HD]) {
    lds_read_half<DIM>(val, h, swz, a);
    lds_read_half<DIM>(sq, h, swz, a2);
  };
  auto close_pair = [&](int tile) {   // after the last tile of a group of GT tiles (or a lone last tile)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      top4_insert(tpend[rt], tile / GT, m1[rt], m2[rt], m3[rt], m4[rt], i1[rt], i2[rt], i3[rt]);
      tpend[rt] = NEG_INF;
    }
  };
  // epilogue of a finished tile: fold its accumulators into the pending group maximum
  auto fold = [&](f32x16 (&d)[RT], int tile, bool closes) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) tpend[rt] = max_chain(tpend[rt], d[rt]);
    if (closes) close_pair(tile);
  };

  const int ntiles = t_full_end - t_begin;
  const int nchunks = ntiles > 0 ? (ntiles + CT - 1) / CT : 0;
  if (nchunks > 0) {
    load_chunk(t_begin);
    store_chunk(0);
  }
  __syncthreads();

#ifdef GQHIP_CLOCK_STAMPS
  const unsigned long long st_r1 = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int S0 = 2;             // k-steps issued before the previous tile's epilogue
  f32x16 dprev[RT];                 // accumulators of the previous tile, epilogue pending
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dprev[rt][r] = NEG_INF;   // neutral element
  int tprev = t_begin - 1;          // its (no-op) epilogue closes an empty group: inserts -inf
  bool have_prev = false;           // dprev holds a real tile whose epilogue is still pending

  for (int ch = 0; ch < nchunks; ++ch) {
    const int tile0 = t_begin + ch * CT;
    if (ch + 1 < nchunks) load_chunk(tile0 + CT);
    const int nt = min(CT, t_full_end - tile0);
    const float *val = lds[ch & 1][0] + c * DIM;   // this lane's code row within a tile
    const float *sq = lds[ch & 1][1] + c * DIM;
    if (nt == CT) {
      float a[HD], a2[HD];
      read_ops(val, sq, a, a2);
      // PREV_CLOSES: the tile whose epilogue runs in this step is the last of its group of GT tiles
      // (tile0 is a multiple of GT, so that is static); LAST: no operand prefetch for a next tile.
      auto step = [&](int tt, auto prev_closes, auto last) {
        constexpr bool PREV_CLOSES = decltype(prev_closes)::value, LAST = decltype(last)::value;
        float an[HD], a2n[HD];
        if constexpr (!LAST) read_ops(val + (tt + 1) * TILE_F, sq + (tt + 1) * TILE_F, an, a2n);
        f32x16 d[RT];
        tile_mfma<DIM, RT, 0, S0>(a, a2, coefA, coefB, d);
        fold(dprev, tprev, PREV_CLOSES);
        tile_mfma<DIM, RT, S0, 2 * HD>(a, a2, coefA, coefB, d);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) dprev[rt] = d[rt];
        tprev = tile0 + tt;
        if constexpr (!LAST) {
#pragma unroll
          for (int s = 0; s < HD; ++s) { a[s] = an[s]; a2[s] = a2n[s]; }
          __builtin_amdgcn_sched_group_barrier(0x100, HD >= 4 ? HD / 2 : 2, 0);
        }
        // pin the order: LDS reads | first MFMAs | ONE VALU cluster | remaining MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, S0 * RT, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, RT * (PREV_CLOSES ? 22 : 9), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, (2 * HD - S0) * RT, 0);
      };
      auto run_all = [&](auto... is) {
        (step(decltype(is)::value, std::bool_constant<(decltype(is)::value % GT) == 0>{},
              std::bool_constant<decltype(is)::value == CT - 1>{}), ...);
      };
      using std::integral_constant;
      if constexpr (CT == 8)
        run_all(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 2>{},
                integral_constant<int, 3>{}, integral_constant<int, 4>{}, integral_constant<int, 5>{},
                integral_constant<int, 6>{}, integral_constant<int, 7>{});
      else
        run_all(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 2>{},
                integral_constant<int, 3>{});
      static_assert(CT == 8 || CT == 4, "unrolled tile loop is written for 4 or 8 tiles per chunk");
      have_prev = true;
    } else {
      if (have_prev) fold(dprev, tprev, (tprev % GT) == GT - 1);  // drain the pipeline, then plain tiles
      have_prev = false;
      for (int tt = 0; tt < nt; ++tt) {
        float a[HD], a2[HD];
        read_ops(val + tt * TILE_F, sq + tt * TILE_F, a, a2);
        f32x16 d[RT];
        tile_mfma<DIM, RT, 0, 2 * HD>(a, a2, coefA, coefB, d);
        fold(d, tile0 + tt, ((tile0 + tt) % GT) == GT - 1);
      }
    }
    if (ch + 1 < nchunks) store_chunk((ch + 1) & 1);
    __syncthreads();
  }
  if (have_prev) fold(dprev, tprev, (tprev % GT) == GT - 1);      // last pipelined tile
#ifdef GQHIP_CLOCK_STAMPS
  const unsigned long long st_r2 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- leftovers: a lone full tile and/or the one partial tile (n % 32 != 0) ----
  const bool pending = ntiles > 0 && (ntiles % GT) != 0;   // the last group is still open
  if (t_end > t_full_end) {
    const int tile = t_full_end;
    const int code = min(tile * kTileCodes + c, p.n - 1);
    float a[HD], a2[HD];
#pragma unroll
    for (int s = 0; s < HD; ++s) {
      a[s] = p.cb[(long)code * DIM + h * HD + s];
      a2[s] = a[s] * a[s];
    }
    f32x16 d[RT];
    tile_mfma<DIM, RT, 0, 2 * HD>(a, a2, coefA, coefB, d);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cc = tile * kTileCodes + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (cc >= p.n) d[rt][r] = NEG_INF;
      }
      tpend[rt] = max_chain(tpend[rt], d[rt]);
    }
    close_pair(tile);
  } else if (pending) {
    close_pair(t_full_end - 1);
  }

#ifdef GQHIP_CLOCK_STAMPS
  if (tid == 0 && vblock < 2048) {   // diagnostic build only: per-block timeline + placement
    unsigned long long *o = reinterpret_cast<unsigned long long *>(p.dbg) + 4 * vblock;
    o[0] = st_r0;
    o[1] = __builtin_amdgcn_s_memrealtime();
    o[2] = ((st_r1 - st_r0) << 32) | (st_r2 - st_r0);   // prologue end, loop end (100 MHz ticks from block start)
    o[3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
           (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#endif
  // ---- merge the two lane halves of each row, write one record -------------
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float a1 = m1[rt], a2v = m2[rt], a3 = m3[rt], a4 = m4[rt];
    int j1 = i1[rt] * 2 + h, j2 = i2[rt] * 2 + h, j3 = i3[rt] * 2 + h;   // half-group ids
    const float b1 = __shfl_xor(a1, 32), b2 = __shfl_xor(a2v, 32), b3 = __shfl_xor(a3, 32), b4 = __shfl_xor(a4, 32);
    const int k1 = __shfl_xor(j1, 32), k2 = __shfl_xor(j2, 32), k3 = __shfl_xor(j3, 32);
    top4_insert(b1, k1, a1, a2v, a3, a4, j1, j2, j3);
    top4_insert(b2, k2, a1, a2v, a3, a4, j1, j2, j3);
    top4_insert(b3, k3, a1, a2v, a3, a4, j1, j2, j3);
    top4_insert_value(b4, a3, a4);
    const int pos = rowblk * (128 * RT) + (wave * RT + rt) * 32 + c;
    if (h == 0 && pos < nrows) {
      const int row = pos;
      p.rec[(long)split * p.rows + row] = make_rec(a1, a2v, a3, a4, j1, j2, j3, 2 * (t_begin / GT));
    }
  }
}

template <int DIM, int RT, int CT, int MODE, int GT>
__global__ __launch_bounds__(256, 2) void gq_filter_kernel(const FilterParams p) {
  filter_block<DIM, RT, CT, MODE, GT>(p, (int)blockIdx.x);
}

}  // namespace gqhip
