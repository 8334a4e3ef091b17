// gq_wino_gemm.h -- the Winograd batched GEMMs of the 256- / 512-channel levels on the [h | l] operand (two-term fp16 split of
// V * scale, written by the input transforms in F16X2 mode: 4 instead of 6 bytes per element of V): the three products
// h U_h + h U_l + l U_h are formed in the kernel on v_mfma_f32_32x32x16_f16 (fp32 accumulation) -- the numerics of the
// K-concatenated library GEMM over [h | h | l] (same splits, same products).
#pragma once
#include "gq_common.h"

namespace gqhip {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- the Winograd batched GEMM of the wider levels (Cin, Cout in {256, 512}): M[p] = V[p] x U[p], [tiles, Cin] x [Cin, Cout].
// The library route (ONE hipBLASLt fp16 GEMM over K' = 3 Cin of V3 = [h | h | l]) runs these at 0.6-0.95 PFLOP/s executed
// and 3-4.4 TB/s: the 256-channel level is HBM-bound on V3 + M, the 512-channel ones sit between both limits
// (tools/gemm_shapes_step.py).  This kernel reads V2 = [h | l] (4 instead of 6 bytes per element, 9 instead of 13.5 out of
// the input transform) and forms the three products itself -- the machinery of conv1x1_f16x3_kernel (gq_conv3.h) without its
// conversions: block = 256 rows x 128 output columns, 4 waves (wave = 128 rows x 64 columns = eight 32 x 32 accumulators),
// two blocks per CU; a stage = 32 k of both planes (32 KiB: [k16 chunk][plane][row] x 32 bytes, the two 16-byte halves of a
// row swapped on odd groups of 8 rows: conflict-free ds_read_b128), double buffered; the weights come straight from L2 in
// MFMA operand order Wf [P][Cin/16][Cout/32][plane][lane][8] (a wave's load is 1 KiB contiguous), one k-step ahead.
// Workgroups go round-robin over the XCDs; the Cout/128 blocks that share a row tile sit next to each other on one XCD.
struct WinoGemm2Params {
  const _Float16 *V2;   // [P][tiles][2 cin]  (h[0..cin) | l[0..cin)) of V * v_scale
  const _Float16 *Wf;   // [P][cin/16][cout/32][2][64][8]  operand-order (U_h, U_l) of U * u_scale
  float *M;             // [P][tiles][cout]
  long tiles;           // a multiple of 256
  int cin, cout, nnb;   // nnb = cout / 128
  long mtiles;          // tiles / 256
  long ntile_total;     // P * mtiles
  long tiles_per_xcd;   // ceil(ntile_total / 8)
};

__global__ __launch_bounds__(256, 2) void wino_gemm_f16x2_kernel(const WinoGemm2Params p) {
  // the four (k16 chunk, plane) regions of a stage are 64 bytes apart modulo 256: the loader's ds_write_b128 of a wave (8 rows x
  // 8 pieces: 2 rows x 32 bytes in each of the four regions per 16 lanes) then covers all 64 banks instead of hitting 16 banks
  // four times (at an 8 KiB region stride the kernel was bound by exactly these writes: 0.66 PFLOP/s)
  constexpr int kPlane = 256 * 32 + 64, kChunk = 2 * kPlane, kStage = 2 * kChunk;   // bytes
  __shared__ __attribute__((aligned(16))) unsigned char sA[2 * kStage];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const long vb = blockIdx.x, kq = vb >> 3;
  const int nb = (int)(kq % p.nnb);
  const long tile = (vb & 7) * p.tiles_per_xcd + kq / p.nnb;
  if (tile >= p.ntile_total) return;
  const long pos = tile / p.mtiles, m0 = (tile % p.mtiles) * 256;
  const int cin = p.cin, nst = cin / 32;
  // loader: thread -> 16-byte piece w8 = tid & 7 of a row's stage (plane w8 >> 2, k16 chunk (w8 >> 1) & 1, half w8 & 1),
  // rows (tid >> 3) + 32 i
  const int w8 = tid & 7, r0 = tid >> 3, pl = w8 >> 2, jc = (w8 >> 1) & 1, jh = w8 & 1;
  const _Float16 *src = p.V2 + ((pos * p.tiles + m0 + r0) * 2L * cin) + pl * cin + 16 * jc + 8 * jh;
  const int loff = jc * kChunk + pl * kPlane + r0 * 32 + 16 * (jh ^ ((r0 >> 3) & 1));
  const long rstride = 64L * cin;   // 32 rows, halfs
  f16x8 st[8];
  auto issue = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = *reinterpret_cast<const f16x8 *>(src + i * rstride + stage * 32);
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f16x8 *>(sA + buf * kStage + loff + i * 1024) = st[i];
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      acc[rr][j] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int aoff = (4 * wm) * 1024 + c * 32 + 16 * (h ^ ((c >> 3) & 1));
  const long wstep = (long)p.nnb * (512 * 16);   // bytes per k-step
  const unsigned char *wbase = reinterpret_cast<const unsigned char *>(p.Wf) + pos * (cin / 16) * wstep +
                               (4 * nb + 2 * wn) * 128 * 16;
  const int wl = lane * 16;
  f16x8 b0[4], b1[4];
  auto load_b = [&](int ks, f16x8 (&dst)[4]) {
    const unsigned char *s = wbase + ks * wstep;
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = *reinterpret_cast<const f16x8 *>(s + k * 1024 + wl);
  };
  auto kstep = [&](const unsigned char *A, const f16x8 (&bq)[4]) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const f16x8 ah = *reinterpret_cast<const f16x8 *>(A + aoff + rr * 1024);
      const f16x8 al = *reinterpret_cast<const f16x8 *>(A + aoff + rr * 1024 + kPlane);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[rr][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[2 * j], acc[rr][j], 0, 0, 0);
        acc[rr][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[2 * j + 1], acc[rr][j], 0, 0, 0);
        acc[rr][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bq[2 * j], acc[rr][j], 0, 0, 0);
      }
    }
  };
  const int nks = 2 * nst;
  load_b(0, b0);
  issue(0);
  commit(0);
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    const bool more = s + 1 < nst;
    const unsigned char *A = sA + (s & 1) * kStage;
    load_b(2 * s + 1, b1);
    if (more) issue(s + 1);
    kstep(A, b0);
    __builtin_amdgcn_sched_barrier(0);
    load_b(2 * s + 2 < nks ? 2 * s + 2 : nks - 1, b0);
    kstep(A + kChunk, b1);
    if (more) commit((s + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  // ---- store: register r of lane (c, h) = row (4 wm + rr) * 32 + (r & 3) + 8 (r >> 2) + 4 h, column (2 wn + j) * 32 + c ----
  float *Mp = p.M + ((pos * p.tiles + m0 + 4 * wm * 32 + 4 * h) * (long)p.cout) + nb * 128 + c;
  const long cs = p.cout;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float *base = Mp + (rr * 32L) * cs + (2 * wn + j) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) base[(long)((r & 3) + 8 * (r >> 2)) * cs] = acc[rr][j][r];
    }
}

// ---- the same GEMM with 256 x 256 block tiles for the 512-channel levels: 8 waves (2 x 4: wave = 128 rows x 64 columns), ONE
// block per CU, a stage = 64 k of both planes (64 KiB, 96 MFMAs per wave between barriers), double buffered (2 x 64.25 KiB of
// LDS).  Against the 256 x 128 form: every row tile is read once per TWO column blocks of a 512-column output instead of
// once per four, the loads of the next stage have 96 instead of 48 MFMAs per wave to land, half the barriers.
__global__ __launch_bounds__(512, 1) void wino_gemm_f16x2_w8_kernel(const WinoGemm2Params p) {
  // eight (k16 chunk, plane) regions, 32 bytes apart modulo 256: the loader's ds_write_b128 of 16 lanes (one row: 16 pieces,
  // two per region) covers all 64 banks
  constexpr int kPlane = 256 * 32 + 32, kChunk = 2 * kPlane, kStage = 4 * kChunk;
  static_assert(kChunk % 256 == 64 && kPlane % 256 == 32, "region stagger");
  __shared__ __attribute__((aligned(16))) unsigned char sA[2 * kStage];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5, wm = wave >> 2, wn = wave & 3;
  const long vb = blockIdx.x, kq = vb >> 3;
  const int nb = (int)(kq % p.nnb);                 // nnb = cout / 256 here
  const long tile = (vb & 7) * p.tiles_per_xcd + kq / p.nnb;
  if (tile >= p.ntile_total) return;
  const long pos = tile / p.mtiles, m0 = (tile % p.mtiles) * 256;
  const int cin = p.cin, nst = cin / 64;
  // loader: thread -> 16-byte piece w16 = tid & 15 of a row's stage (plane w16 >> 3, k16 chunk (w16 >> 1) & 3, half w16 & 1),
  // rows (tid >> 4) + 32 i
  const int w16 = tid & 15, r0 = tid >> 4, pl = w16 >> 3, jc = (w16 >> 1) & 3, jh = w16 & 1;
  const _Float16 *src = p.V2 + ((pos * p.tiles + m0 + r0) * 2L * cin) + pl * cin + 16 * jc + 8 * jh;
  const int loff = jc * kChunk + pl * kPlane + r0 * 32 + 16 * (jh ^ ((r0 >> 3) & 1));
  const long rstride = 64L * cin;   // 32 rows, halfs
  f16x8 st[8];
  auto issue = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = *reinterpret_cast<const f16x8 *>(src + i * rstride + stage * 64);
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f16x8 *>(sA + buf * kStage + loff + i * 1024) = st[i];
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      acc[rr][j] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int aoff = (4 * wm) * 1024 + c * 32 + 16 * (h ^ ((c >> 3) & 1));
  const long wstep = (long)(p.cout / 32) * 2048;   // bytes per k-step: column tiles x 2 planes x 1 KiB
  const unsigned char *wbase = reinterpret_cast<const unsigned char *>(p.Wf) + pos * (cin / 16) * wstep +
                               (8 * nb + 2 * wn) * 2048L;
  const int wl = lane * 16;
  // Weights: three register sets, loads TWO k-steps ahead.  vmcnt retires in order, so the consumer of a weight load also waits
  // for every older load -- the 8 row loads of the next stage included: with one k-step of lead those had ~0.75 us to come back
  // from HBM (ablation: without the row loads 266 instead of 323 us, without the weight loads 267, without both 220).
  // A operands: the (h, l) pair of the next 32-row group is read from LDS before the six MFMAs of the current one (fenced:
  // hipcc otherwise sinks the reads to their use).
  const int nks = 4 * nst;
  f16x8 bA[4], bB[4], bC[4];
  auto load_b = [&](int ks, f16x8 (&dst)[4]) {
    const unsigned char *s = wbase + (long)(ks < nks ? ks : nks - 1) * wstep;
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = *reinterpret_cast<const f16x8 *>(s + k * 1024 + wl);
  };
  f16x8 ah, al, nh, nl;
  auto read_a = [&](const unsigned char *A, int u, f16x8 &oh, f16x8 &ol) {     // unit u = (k-step u >> 2, row group u & 3)
    const unsigned char *q = A + (u >> 2) * kChunk + aoff + (u & 3) * 1024;
    oh = *reinterpret_cast<const f16x8 *>(q);
    ol = *reinterpret_cast<const f16x8 *>(q + kPlane);
  };
  auto mfma6 = [&](int rr, const f16x8 (&bq)[4]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc[rr][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[2 * j], acc[rr][j], 0, 0, 0);
      acc[rr][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[2 * j + 1], acc[rr][j], 0, 0, 0);
      acc[rr][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bq[2 * j], acc[rr][j], 0, 0, 0);
    }
  };
  // one k-step: four row groups, the next unit's operands read ahead
  auto kstep = [&](const unsigned char *A, int ks4, const f16x8 (&bq)[4], bool last_of_stage) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int u = 4 * ks4 + rr;
      if (!(last_of_stage && rr == 3)) read_a(A, u + 1, nh, nl);
      __builtin_amdgcn_sched_barrier(0);
      mfma6(rr, bq);
      ah = nh;
      al = nl;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  load_b(0, bA);
  load_b(1, bB);
  issue(0);
  commit(0);
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    const bool more = s + 1 < nst;
    const unsigned char *A = sA + (s & 1) * kStage;
    read_a(A, 0, ah, al);
    load_b(4 * s + 2, bC);
    if (more) issue(s + 1);
    kstep(A, 0, bA, false);
    load_b(4 * s + 3, bA);
    kstep(A, 1, bB, false);
    load_b(4 * s + 4, bB);
    kstep(A, 2, bC, false);
    load_b(4 * s + 5, bC);
    kstep(A, 3, bA, true);
    if (more) commit((s + 1) & 1);
    // rotate: the next stage starts with k-step 4 s + 4 in bB and 4 s + 5 in bC
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bA[k] = bB[k];
      bB[k] = bC[k];
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  float *Mp = p.M + ((pos * p.tiles + m0 + 4 * wm * 32 + 4 * h) * (long)p.cout) + nb * 256 + c;
  const long cs = p.cout;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float *base = Mp + (rr * 32L) * cs + (2 * wn + j) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) base[(long)((r & 3) + 8 * (r >> 2)) * cs] = acc[rr][j][r];
    }
}

}  // namespace gqhip
