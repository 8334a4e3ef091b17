// gq_conv3.h -- direct 3x3 convolution (stride 1, zero padding 1), Cin -> Cout (a multiple of 128) channels, NHWC, as an
// implicit GEMM on v_mfma_f32_32x32x16_f16: the convolutions of the two widest levels (pit/modules/unet.py:142, :149).
//
// Why not Winograd here: at 128 channels the Winograd route is HBM-bound -- V and M (2.25x / 4x the activation each)
// cross HBM twice: 6.4 GB per F(4x4,3x3) convolution of a 16 x 256 x 256 x 128 tensor, 10.7 GB with F(2x2,3x3) -- and no
// fusion removes that: the transformed weights of all tile positions (2.4 MB with their fp16 split) do not fit a CU.
// The direct form reads the activation once and writes the result once (1.6 GB with the residual); its 2.25x / 4x more
// multiplies run on the matrix cores, which have the room (928 GFLOP of fp16 MFMA work per convolution).
//
// Numerics: the same fp16 x 3 scheme as the Winograd GEMMs (gq_wino_gemm.h): activation and weight are two-term fp16
// splits (h + l, 22 bits) of the scaled fp32 values, the three products h W_h + h W_l + l W_h accumulate in fp32 --
// without Winograd's transform amplification.
//
// conv3x3_gn_f16x3_kernel: block = 8 x 32 output pixels x 128 output channels, 4 waves (wave = 4 rows x 64 channels: eight
// 32 x 32 accumulator tiles).  Per 16-channel chunk the 10 x 34-pixel patch of the fp32 input is normalised (GroupNorm), activated
// (SiLU), scaled and split on its way into LDS (h and l planes, 21.8 KB, double buffered, 32 bytes per pixel and plane with the
// two 16-byte halves swapped on odd groups of 8 pixels: any 16 consecutive pixels then hit 64 different banks with ds_read_b128);
// each of the 9 taps is one MFMA k-step whose A operand is the patch shifted by (dy, dx) -- 8 LDS reads for 24 MFMAs per wave.
// The weights, laid out in operand order ([chunk][tap][column tile][plane][lane][8]), come straight from L2 into registers, two
// k-steps ahead (three register sets): a wave's load is 1 KB contiguous.  Epilogue: * mscale + bias (+ residual), store,
// GroupNorm statistics of the result.  (Round 2's two-kernel form -- a split pass + a convolution of the pre-split tensor --
// was never selected at a BASELINE shape and is gone: profiles/r03/route_table.txt.)
#pragma once
#include <type_traits>

#include "gq_common.h"
#include "gq_stats.h"
#include "gq_wino_gemm.h"

namespace gqhip {

constexpr int kC3TH = 8, kC3TW = 32;                  // output tile of a block
constexpr int kC3PH = kC3TH + 2, kC3PW = kC3TW + 2;   // staged patch: 10 x 34 pixels
constexpr int kC3Pix = kC3PH * kC3PW;                 // 340
constexpr int kC3RS = 48;                             // LDS row stride in pixels (a multiple of 16: see conv3_lds_off)
constexpr int kC3Plane = kC3PH * kC3RS * 32;          // bytes per plane: 16 channels x 2 bytes per pixel, 15 360
constexpr int kC3Buf = 2 * kC3Plane;                  // h plane + l plane of one chunk
constexpr int kC3Pieces = kC3Pix * 4;                 // 16-byte pieces per chunk (2 planes x 2 halves per pixel)
constexpr int kC3Loads = (kC3Pieces + 255) / 256;     // per thread and chunk: 6

// Byte offset of 16-byte half `half` of patch pixel (R, x) inside a plane: 32 bytes per pixel, the two halves swapped on
// odd groups of 8 pixels of the row -- any 16 consecutive pixels of a row then cover all 64 banks with ds_read_b128.  The
// row stride of 48 pixels keeps the swap a function of x alone, so a wave's read address for tap (dy, dx) is one of three
// per-lane registers (dx) plus a compile-time constant (row, plane, buffer).
__device__ __forceinline__ int conv3_lds_off(int R, int x, int half) {
  return (R * kC3RS + x) * 32 + 16 * (half ^ ((x >> 3) & 1));
}

struct Conv3Params {
  const _Float16 *Xs;   // [B][nch][H][W][2][16]
  const _Float16 *Wf;   // [nch][9][cout/32][2][64][8]
  const float *bias;    // [cout] or null
  const float *res;     // [B][H][W][cout] or null
  float *y;             // [B][H][W][cout]
  int64_t *stats;     // [B][groups] statistics records (gq_stats.h) or null
  int H, W, nch, cpg;   // cpg: channels per GroupNorm group of the output
  int cout, nnb;        // output channels (a multiple of 128), 128-channel blocks per tile (cout / 128)
  int tiles_x, tiles_y;
  long ntiles;          // B * tiles_y * tiles_x
  long tiles_per_xcd;   // ceil(ntiles / 8)
  float mscale;
};

// ---- pieces shared by the two convolution kernels ----
struct Conv3Tile {
  long b;
  int y0, x0, nb;   // nb: which 128 output channels
};
__device__ __forceinline__ bool conv3_tile(const Conv3Params &p, Conv3Tile &t, long vb = blockIdx.x) {
  // workgroups go round-robin over the 8 XCDs: consecutive tiles (shared halos, same weights) and the 128-channel blocks
  // of one tile (same input patch) land on the same XCD, next to each other in dispatch order
  const long k = vb >> 3;
  t.nb = (int)(k % p.nnb);
  const long tile = (long)(vb & 7) * p.tiles_per_xcd + k / p.nnb;
  if (tile >= p.ntiles) return false;
  const int tx = (int)(tile % p.tiles_x);
  const long t2 = tile / p.tiles_x;
  t.b = t2 / p.tiles_y;
  t.y0 = (int)(t2 % p.tiles_y) * kC3TH;
  t.x0 = tx * kC3TW;
  return true;
}

// One tap (= one MFMA k-step of 16 input channels) of a wave: A operands = the staged patch shifted by (dy, dx).
// `A` = buffer base + the lane's offset for this dx (conv3_lds_off(4 wm, c + dx, h)).
__device__ __forceinline__ void conv3_tap(const unsigned char *A, int dy, const f16x8 (&bq)[4], f32x16 (&acc)[4][2]) {
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
#if defined(GQHIP_ABL) && (GQHIP_ABL & 256)        // diagnostic build: one A operand per tap instead of eight
    const f16x8 ah = *reinterpret_cast<const f16x8 *>(A + dy * kC3RS * 32);
    const f16x8 al = ah;
#else
    const f16x8 ah = *reinterpret_cast<const f16x8 *>(A + (rr + dy) * kC3RS * 32);
    const f16x8 al = *reinterpret_cast<const f16x8 *>(A + (rr + dy) * kC3RS * 32 + kC3Plane);
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc[rr][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[2 * j], acc[rr][j], 0, 0, 0);
      acc[rr][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[2 * j + 1], acc[rr][j], 0, 0, 0);
      acc[rr][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bq[2 * j], acc[rr][j], 0, 0, 0);
    }
  }
}

// Epilogue: register r of lane (c, h) = pixel x0 + (r & 3) + 8 (r >> 2) + 4 h of row y0 + 4 wm + rr, channel
// (2 wn + j) * 32 + c: a store of one register is two 128-byte runs.
// COUT is a template parameter: with a run-time channel stride none of the 128 loads / 128 stores of a lane has a constant
// offset and hipcc materialises their addresses (~300 bytes of scratch per lane).
template <int COUT>
__device__ __forceinline__ void conv3_epilogue(const Conv3Params &p, const Conv3Tile &t, const f32x16 (&acc)[4][2], int64_t *red,
                                               int tid, int wm, int wn, int c, int h) {
  float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};
  const int W = p.W;
  constexpr int cout = COUT;
  const long pix0 = ((t.b * p.H + t.y0 + 4 * wm) * W + t.x0 + 4 * h) * cout + t.nb * 128;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = (2 * wn + j) * 32 + c;
    const float pb = p.bias ? p.bias[t.nb * 128 + n] : 0.f;
    float rv[4][16];   // the 64 residual values of this column tile in flight at once
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        rv[rr][r] = p.res ? p.res[pix0 + ((long)rr * W + (r & 3) + 8 * (r >> 2)) * cout + n] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[rr][j][r] * p.mscale + pb + rv[rr][r];
        p.y[pix0 + ((long)rr * W + (r & 3) + 8 * (r >> 2)) * cout + n] = v;
        s[j] += v;
        ss[j] += v * v;
      }
  }
  if (p.stats) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int g = ((2 * wn + j) * 32 + c) / p.cpg;      // group within this block's 128 channels
      stat_add_f32(red + kStatWords * g, s[j], ss[j]);
    }
    __syncthreads();
    const int gpb = 128 / p.cpg, groups = cout / p.cpg;    // groups per block, per image
    if (tid < kStatWords * gpb) stat_flush_word(p.stats + kStatWords * (t.b * groups + t.nb * gpb) + tid, red[tid]);
  }
}

#define GQ_C3_ZERO_ACC(acc)                                                                                         \
  _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[rr][j] =       \
      f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}

// ---- input = the fp32 tensor itself: SiLU(GroupNorm(x + pre_bias)) * scale and the fp16 split happen on the way into
//      LDS (no Xs: saves writing and re-reading 4 bytes per element and a launch).  A chunk is 16 channels = 64 bytes of
//      a pixel's fp32 row; thread -> 4 channels of up to 6 patch pixels (the same channel quad for all of them), the
//      folded scale / shift of the image's channels sit in LDS.  The conversions of chunk k + 1 are spread over taps 3..8
//      of chunk k (one piece per tap: in the MFMAs' shadow, and late enough for the loads issued at tap 0 to have landed).
struct Conv3GnParams {
  Conv3Params c;
  const float *x;          // [B][H][W][cin]
  const float *gamma, *beta, *pre_bias;
  const int64_t *stats_in;  // [B][groups_in] statistics records (gq_stats.h)
  int cin, cpg_in;
  double eps;
  float scale;
};

#ifdef GQHIP_CLOCK_STAMPS
__device__ unsigned long long g_c3_stamps[4 * 8192];   // diagnostic build: per-block timeline (tools/c3_timeline.py)
#endif
template <int SILU, int COUT>
__global__ __launch_bounds__(256, 2) void conv3x3_gn_f16x3_kernel(const Conv3GnParams pp) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const Conv3Params &p = pp.c;
  __shared__ __attribute__((aligned(16))) unsigned char sA[2 * kC3Buf];
  __shared__ int64_t red[kStatWords * 32];   // statistics records of the block's <= 32 groups (gq_stats.h)
  __shared__ __attribute__((aligned(16))) float sAff[2][512];   // folded scale, shift per input channel (cin <= 512)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
#ifdef GQHIP_CLOCK_STAMPS
  const unsigned long long st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  Conv3Tile t;
  if (!conv3_tile(p, t)) return;
  const int H = p.H, W = p.W, cin = pp.cin;
  red[tid] = 0;   // kStatWords * 32 = 256 words
  {
    const int groups = cin / pp.cpg_in;
    const double n = (double)pp.cpg_in * (double)H * (double)W;
    for (int ch = tid; ch < cin; ch += 256) {
      const int g = ch / pp.cpg_in;
      double st_s, st_ss;
      stat_load(pp.stats_in + kStatWords * (t.b * groups + g), st_s, st_ss);
      const double mean = st_s / n;
      double var = st_ss / n - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double rstd = 1.0 / sqrt(var + pp.eps);
      const double pbk = pp.pre_bias ? (double)pp.pre_bias[ch] : 0.0;
      sAff[0][ch] = (float)(rstd * (double)pp.gamma[ch]);
      sAff[1][ch] = (float)((double)pp.beta[ch] + (pbk - mean) * rstd * (double)pp.gamma[ch]);
    }
  }
  // loader bookkeeping: thread -> channel quad w = tid & 3 of patch pixels q = (tid >> 2) + 64 i (i < 6, q < 340)
  const int w = tid & 3, qs = tid >> 2;
  int goff[kC3Loads];
  unsigned inb = 0;
#pragma unroll
  for (int i = 0; i < kC3Loads; ++i) {
    int q = qs + 64 * i;
    q = q < kC3Pix ? q : kC3Pix - 1;
    const int R = q / kC3PW, px = q % kC3PW;
    const int gy = t.y0 - 1 + R, gx = t.x0 - 1 + px;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), cx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
    goff[i] = (cy * W + cx) * cin + 4 * w;                      // floats, within the image
    inb |= (unsigned)(in ? 1 : 0) << i;
  }
  const float *xb = pp.x + t.b * (long)H * W * cin;
  // three staging registers: piece i of the next chunk is loaded at tap i (i < 3) or tap i (3 <= i < 6, into the register
  // piece i - 3 just left) and converted three taps later
  f32x4 st[3];
  auto issue = [&](int i, int chunk) { st[i % 3] = *reinterpret_cast<const f32x4 *>(xb + goff[i] + chunk * 16); };
  auto convert = [&](int i, int chunk, int buf) {    // piece i of the chunk (in st[i % 3]) -> activated, scaled, split, into LDS
    const int q = qs + 64 * i;
    if (q >= kC3Pix) return;
    const f32x4 a4 = *reinterpret_cast<const f32x4 *>(&sAff[0][chunk * 16 + 4 * w]);
    const f32x4 sh4 = *reinterpret_cast<const f32x4 *>(&sAff[1][chunk * 16 + 4 * w]);
    const f32x4 v = gn_act<SILU>(st[i % 3], a4, sh4) * (((inb >> i) & 1) ? pp.scale : 0.f);
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = (_Float16)v[e];
      lo[e] = (_Float16)(v[e] - (float)hi[e]);
    }
    const int off = buf * kC3Buf + conv3_lds_off(q / kC3PW, q % kC3PW, w >> 1) + 8 * (w & 1);
    *reinterpret_cast<f16x4 *>(&sA[off]) = hi;
    *reinterpret_cast<f16x4 *>(&sA[off + kC3Plane]) = lo;
  };

  f32x16 acc[4][2];
  GQ_C3_ZERO_ACC(acc);
  int aoff[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) aoff[dx] = conv3_lds_off(4 * wm, c + dx, h);
  // weights of k-step ks = chunk * 9 + tap, column tiles 2 wn and 2 wn + 1, planes h / l: 4 operands of 1 KB per wave;
  // uniform base + lane offset.  THREE register sets, loads two taps ahead: vmcnt retires in order, so with one tap of
  // lead every wait for a weight operand (L2) also waited for the x loads issued just before it (HBM).  Now the wait at
  // tap T is for loads issued at tap T - 2, the x piece issued at the end of tap T - 3 is first covered by the wait at
  // tap T -- the tap that converts it.
  const unsigned char *wbase = reinterpret_cast<const unsigned char *>(p.Wf) + (4 * t.nb + 2 * wn) * 128 * 16;
  const int wl = lane * 16;
  const long wstep = (long)p.nnb * (512 * 16);   // bytes per k-step
  f16x8 bs[3][4];
  auto load_b = [&](int ks, f16x8 (&dst)[4]) {
    const unsigned char *s = wbase + ks * wstep;
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = *reinterpret_cast<const f16x8 *>(s + k * 1024 + wl);
  };
  const int nks = p.nch * 9;
  load_b(0, bs[0]);
  load_b(1, bs[1]);
  __syncthreads();          // sAff
  for (int i0 = 0; i0 < kC3Loads; i0 += 3) {   // chunk 0, three pieces at a time
#pragma unroll
    for (int i = 0; i < 3; ++i) issue(i0 + i, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i) convert(i0 + i, 0, 0);
  }
  __syncthreads();

#ifdef GQHIP_CLOCK_STAMPS
  const unsigned long long st_r1 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int chunk = 0; chunk < p.nch; ++chunk) {
    const bool more = chunk + 1 < p.nch;
    const unsigned char *A = sA + (chunk & 1) * kC3Buf;
    const int nbuf = (chunk + 1) & 1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ks = chunk * 9 + tap;
#if !(defined(GQHIP_ABL) && (GQHIP_ABL & 128))     // diagnostic build: weights loaded once
      load_b(ks + 2 < nks ? ks + 2 : nks - 1, bs[(tap + 2) % 3]);
#endif
      conv3_tap(A + aoff[tap % 3], tap / 3, bs[tap % 3], acc);
#if !(defined(GQHIP_ABL) && (GQHIP_ABL & 64))      // diagnostic build: no staging of the next chunk
      if (more) {
        if (tap >= 3) convert(tap - 3, chunk + 1, nbuf);
        if (tap < kC3Loads) issue(tap, chunk + 1);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);   // keeps hipcc from hoisting the LDS reads of later taps (spills otherwise)
    }
    __syncthreads();
  }
#ifdef GQHIP_CLOCK_STAMPS
  const unsigned long long st_r2 = __builtin_amdgcn_s_memrealtime();
#endif
#if defined(GQHIP_ABL) && (GQHIP_ABL & 512)         // diagnostic build: no epilogue
  if (acc[0][0][0] == 12345.678f)
#endif
  conv3_epilogue<COUT>(p, t, acc, red, tid, wm, wn, c, h);
#ifdef GQHIP_CLOCK_STAMPS
  if (tid == 0 && blockIdx.x < 8192) {
    unsigned long long *o = g_c3_stamps + 4 * blockIdx.x;
    o[0] = st_r0;
    o[1] = __builtin_amdgcn_s_memrealtime();
    o[2] = ((st_r1 - st_r0) << 32) | (st_r2 - st_r0);   // prologue end, loop end (100 MHz ticks from block start)
    o[3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#endif
}

// Where the time goes (round 3; profiles/r03/conv3_diagnosis.txt): per block (16 x 256 x 256 x 128 -> 128, 4096 blocks, two per
// CU) prologue 12 us, main loop 106 us, epilogue 21 us; the matrix pipes are busy 54 % of 1.65 M cycles at the 1.62 GHz the chip
// holds.  Diagnostic builds (make ablu ABL=...): without the weight loads 1.34 M cycles, without staging the next chunk 1.27 M,
// without the epilogue 1.37 M, one A operand per tap instead of eight 1.79 M (LDS reads cost nothing), all off 0.98 M cycles
// (pipes 90 % busy at 1.88 GHz).  Tried and measured within 2 %: delaying the second block of every CU by 10-80 us (the two
// blocks' phases are already spread: both are inside their main loop 50 % of the time, one 46 %), one block per CU (1124 vs
// 1019 us: a wave alone on its SIMD runs its main loop almost twice as fast), weight loads pinned to the top of the tap and the
// conversion's ~60 VALU instructions spread between the MFMAs with sched_group_barrier (hipcc keeps them in one block; 8 spills),
// a wave owning 8 rows x 32 channels (two weight operands per tap instead of four, none loaded twice per block, 16 A reads:
// bit-identical, no spills, 1045 vs 1031-1052 us), s_setprio 1 / 3 around the tap's MFMAs or around the conversion instead
// (1068-1080 us), __builtin_amdgcn_iglp_opt(0) per tap (-1 %, at the noise level).  The texture addresser is busy 26 % of the cycles, L2 read latency averages
// 317 cycles, 67 % of the L2 requests hit: no unit of the memory pipeline is saturated -- the waves' in-order waits are.

// Measured alternative (round 2, removed): a wave-specialised persistent variant -- one block of 8 waves per CU, waves 0-3
// only multiplying (every operand from LDS, next tap's operands read under the current tap's MFMAs), waves 4-7 only
// loading (x two chunks ahead, converted into the patch buffer; the weights three steps ahead into an LDS ring), one
// barrier per three taps.  Bit-identical results; 1003-1056 us against 854-873 us for the kernel above at 16 x 256 x 256 x
// 128.  Decomposition (loaders or multipliers reduced to their barriers): multipliers alone 808 us, of which 263 us is the
// epilogue that nothing overlaps in that design (the main loop itself ran at 1.7 PFLOP/s); loaders alone 737 us -- four
// loader waves per CU are as slow as the multipliers; both together 1032 us.  Two combined-role blocks per CU overlap one
// block's epilogue with the other's main loop for free, which is worth more here than the cleaner instruction streams.

// ---- 1x1 convolution Cin -> 128 / 256 channels (the ResnetBlocks' nin_shortcut, unet.py:151-152; the attention block's
// proj_out, unet.py:203) = a GEMM over the pixels, same fp16 x 3 scheme, the split of x done on the way into LDS.  MIOpen
// runs these on the fp32 matrix cores at ~100 TFLOP/s (725 us for 256 -> 128 at 16 x 256 x 256, HBM floor 320 us).
// The pixel rows are treated as an image of width 32 (tile = 256 consecutive pixels of ONE image: HW % 256 == 0), so tile
// decoding and the epilogue are those of the 3x3 kernel.  A stage = 32 channels (two MFMA k-steps; a pixel's 128 bytes =
// one line), double buffered: 2 x 32 KB.  x is NOT normalised here (the shortcut takes the raw residual stream): its
// scale comes from the host (a bound the caller knows) or from device memory (f16_scales_from_gn_stats: no sync).
struct Conv1Params {
  Conv3Params c;           // H = HW / 32, W = 32, tiles_x = 1, tiles_y = HW / 256
  const float *x;          // [B * HW][cin]
  const float *pre_bias;   // [cin] or null: added to x before the split (a bias still pending on x)
  const float *scales_dev; // {scale, 1 / (scale * u_scale)} or null (then c.mscale and `scale` below)
  float scale;
  int cin;
};

template <int COUT>
__global__ __launch_bounds__(256, 2) void conv1x1_f16x3_kernel(const Conv1Params pp) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  Conv3Params p = pp.c;
  constexpr int kPlane = 256 * 32, kChunk = 2 * kPlane, kStage = 2 * kChunk;   // bytes
  __shared__ __attribute__((aligned(16))) unsigned char sA[2 * kStage];
  __shared__ int64_t red[kStatWords * 32];   // statistics records of the block's <= 32 groups (gq_stats.h)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  Conv3Tile t;
  if (!conv3_tile(p, t)) return;
  red[tid] = 0;   // kStatWords * 32 = 256 words
  const float scale = pp.scales_dev ? pp.scales_dev[0] : pp.scale;
  if (pp.scales_dev) p.mscale = pp.scales_dev[1];
  const int cin = pp.cin, nst = cin / 32;
  // loader: thread -> channel quad w8 = tid & 7 of the stage's 32 channels, rows (tid >> 3) + 32 i
  const int w8 = tid & 7, r0 = tid >> 3;
  const float *xb = pp.x + ((t.b * p.H + t.y0) * 32L + r0) * cin + 4 * w8;
  const float *pbp = pp.pre_bias ? pp.pre_bias + 4 * w8 : nullptr;
  f32x4 pb4 = {0.f, 0.f, 0.f, 0.f};
  const int loff = (w8 >> 2) * kChunk + r0 * 32 + 16 * (((w8 >> 1) & 1) ^ ((r0 >> 3) & 1)) + 8 * (w8 & 1);
  f32x4 st[8];
  auto issue = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = *reinterpret_cast<const f32x4 *>(xb + (long)(32 * i) * cin + stage * 32);
  };
  auto convert = [&](int i, int buf) {     // rows r0 + 32 i: (r >> 3) & 1 = (r0 >> 3) & 1
    const f32x4 v = (st[i] + pb4) * scale;
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = (_Float16)v[e];
      lo[e] = (_Float16)(v[e] - (float)hi[e]);
    }
    unsigned char *d = sA + buf * kStage + loff + i * (32 * 32);
    *reinterpret_cast<f16x4 *>(d) = hi;
    *reinterpret_cast<f16x4 *>(d + kPlane) = lo;
  };
  f32x16 acc[4][2];
  GQ_C3_ZERO_ACC(acc);
  const int aoff = (4 * wm) * 1024 + c * 32 + 16 * (h ^ ((c >> 3) & 1));
  const unsigned char *wbase = reinterpret_cast<const unsigned char *>(p.Wf) + (4 * t.nb + 2 * wn) * 128 * 16;
  const int wl = lane * 16;
  const long wstep = (long)p.nnb * (512 * 16);
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
  if (pbp) pb4 = *reinterpret_cast<const f32x4 *>(pbp);
#pragma unroll
  for (int i = 0; i < 8; ++i) convert(i, 0);
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    const bool more = s + 1 < nst;
    const unsigned char *A = sA + (s & 1) * kStage;
    const int nbuf = (s + 1) & 1;
    load_b(2 * s + 1, b1);
    if (more) issue(s + 1);
    kstep(A, b0);
    __builtin_amdgcn_sched_barrier(0);
    load_b(2 * s + 2 < nks ? 2 * s + 2 : nks - 1, b0);
    kstep(A + kChunk, b1);
    if (more) {
      if (pbp) pb4 = *reinterpret_cast<const f32x4 *>(pbp + (s + 1) * 32);
#pragma unroll
      for (int i = 0; i < 8; ++i) convert(i, nbuf);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  conv3_epilogue<COUT>(p, t, acc, red, tid, wm, wn, c, h);
}

// ---- 3x3 convolution with stride 2 on an input padded by one zero row / column at the bottom / right (Downsample,
// unet.py:76-97): out(y, x) = sum_k w[ky][kx] in(2y + ky, 2x + kx).  MIOpen: fp32 implicit GEMM at ~115 TFLOP/s (0.64-0.70 ms
// per convolution of the encoder).  Here: the same fp16 x 3 machinery on the four PHASE images P_ab(y, x) = in(2y + a, 2x + b):
// the convolution is the sum of four stride-1 convolutions with 2x2, 2x1, 1x2 and 1x1 kernels (taps (dy, dx) with
// ky = 2 dy + a, kx = 2 dx + b <= 2) -- nine k-steps per 16 input channels, none wasted.  A staging unit = 16 channels of
// one phase over the tile's 9 x 33 patch (it fits the stride-1 kernel's LDS buffer and uses its operand addressing);
// phases are the OUTER loop so that the two 64-byte halves of an input line are fetched by consecutive units.
// x is not normalised here (Downsample takes the residual stream): scale from the host or from device memory.
// Wf: k-steps in the order (phase, chunk, tap of the phase) -- _lib.conv3s2_weights_f16.
struct Conv3S2Params {
  Conv3Params c;            // H, W: OUTPUT size; tiles over the output
  const float *x;           // [B][Hin][Win][cin]
  const float *scales_dev;  // {scale, 1 / (scale * u_scale)} or null
  float scale;
  int cin, Hin, Win;
};

template <int COUT>
__global__ __launch_bounds__(256, 2) void conv3x3s2_f16x3_kernel(const Conv3S2Params pp) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  Conv3Params p = pp.c;
  __shared__ __attribute__((aligned(16))) unsigned char sA[2 * kC3Buf];
  __shared__ int64_t red[kStatWords * 32];   // statistics records of the block's <= 32 groups (gq_stats.h)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  Conv3Tile t;
  if (!conv3_tile(p, t)) return;
  red[tid] = 0;   // kStatWords * 32 = 256 words
  const float scale = pp.scales_dev ? pp.scales_dev[0] : pp.scale;
  if (pp.scales_dev) p.mscale = pp.scales_dev[1];
  const int cin = pp.cin, nch = p.nch, Hin = pp.Hin, Win = pp.Win;
  constexpr int PR = kC3TH + 1, PC = kC3TW + 1, NP = 5;      // patch 9 x 33 pixels; pieces per thread (297 x 4 / 256)
  const int w = tid & 3, qs = tid >> 2;
  const float *xb = pp.x + t.b * (long)Hin * Win * cin + 4 * w;
  int goff[NP], loff[NP];
  unsigned inb = 0;
  auto setup_phase = [&](int a, int b) {
    inb = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int q = qs + 64 * i;
      const int qq = q < PR * PC ? q : PR * PC - 1;
      const int R = qq / PC, X = qq % PC;
      const int gy = 2 * (t.y0 + R) + a, gx = 2 * (t.x0 + X) + b;
      const bool in = gy < Hin && gx < Win;
      const int cy = gy < Hin ? gy : Hin - 1, cx = gx < Win ? gx : Win - 1;
      goff[i] = (cy * Win + cx) * cin;
      loff[i] = q < PR * PC ? conv3_lds_off(R, X, w >> 1) + 8 * (w & 1) : -1;
      inb |= (unsigned)(in ? 1 : 0) << i;
    }
  };
  f32x4 st[NP];
  auto issue = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NP; ++i) st[i] = *reinterpret_cast<const f32x4 *>(xb + goff[i] + chunk * 16);
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (loff[i] < 0) continue;
      const f32x4 v = st[i] * (((inb >> i) & 1) ? scale : 0.f);
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (_Float16)v[e];
        lo[e] = (_Float16)(v[e] - (float)hi[e]);
      }
      unsigned char *d = sA + buf * kC3Buf + loff[i];
      *reinterpret_cast<f16x4 *>(d) = hi;
      *reinterpret_cast<f16x4 *>(d + kC3Plane) = lo;
    }
  };
  f32x16 acc[4][2];
  GQ_C3_ZERO_ACC(acc);
  int aoff[2];
#pragma unroll
  for (int dx = 0; dx < 2; ++dx) aoff[dx] = conv3_lds_off(4 * wm, c + dx, h);
  const unsigned char *wbase = reinterpret_cast<const unsigned char *>(p.Wf) + (4 * t.nb + 2 * wn) * 128 * 16;
  const int wl = lane * 16;
  const long wstep = (long)p.nnb * (512 * 16);
  f16x8 bq[4], bn[4];
  auto load_b = [&](int ks, f16x8 (&dst)[4]) {
    const unsigned char *s = wbase + ks * wstep;
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = *reinterpret_cast<const f16x8 *>(s + k * 1024 + wl);
  };
  const int nks = 9 * nch;
  int ks = 0, unit = 0;
  load_b(0, bq);
  setup_phase(0, 0);
  issue(0);
  commit(0);
  __syncthreads();
  // one unit: NT taps of phase (a, b) on the staged patch, while the next unit (next chunk, or chunk 0 of the next phase) loads
  auto run_unit = [&](auto nt_tag, auto a_tag, auto b_tag, int chunk) {
    constexpr int NT = decltype(nt_tag)::value, PA = decltype(a_tag)::value, PB = decltype(b_tag)::value;
    const bool last_chunk = chunk + 1 >= nch;
    const bool more = !(last_chunk && PA == 1 && PB == 1);
    if (more) {
      if (last_chunk) setup_phase(PB == 1 ? 1 : PA, PB == 1 ? 0 : 1);      // (0,0) -> (0,1) -> (1,0) -> (1,1)
      issue(last_chunk ? 0 : chunk + 1);
    }
    const unsigned char *A = sA + (unit & 1) * kC3Buf;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      // phase (0,0): (dy, dx) = (ti / 2, ti % 2); (0,1): (ti, 0); (1,0): (0, ti); (1,1): (0, 0)
      const int ddy = (PA == 0 && PB == 0) ? ti / 2 : (PA == 0 && PB == 1) ? ti : 0;
      const int ddx = (PA == 0 && PB == 0) ? ti % 2 : (PA == 1 && PB == 0) ? ti : 0;
      load_b(ks + 1 < nks ? ks + 1 : ks, bn);
      conv3_tap(A + aoff[ddx], ddy, bq, acc);
#pragma unroll
      for (int k = 0; k < 4; ++k) bq[k] = bn[k];
      ++ks;
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) commit((unit + 1) & 1);
    ++unit;
    __syncthreads();
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
  for (int chunk = 0; chunk < nch; ++chunk) run_unit(I4{}, I0{}, I0{}, chunk);
  for (int chunk = 0; chunk < nch; ++chunk) run_unit(I2{}, I0{}, I1{}, chunk);
  for (int chunk = 0; chunk < nch; ++chunk) run_unit(I2{}, I1{}, I0{}, chunk);
  for (int chunk = 0; chunk < nch; ++chunk) run_unit(I1{}, I1{}, I1{}, chunk);
  conv3_epilogue<COUT>(p, t, acc, red, tid, wm, wn, c, h);
}

// ---- "nearest x2 upsample, then 3x3 convolution" (Upsample, unet.py:60-73) as its sub-pixel form, computed directly: output
// phase (a, b) = (oy & 1, ox & 1) is a 2x2 convolution of the LOW-resolution input,
//     y[2i + a][2j + b] = sum_{u, v in {0,1}} Wp[a][b][u][v] x[i - 1 + a + u][j - 1 + b + v]     (zero outside the image),
// with Wp = the sums of the 3x3 taps that land on the same source pixel (a = 0: rows {0}, {1,2}; a = 1: {0,1}, {2}).  The
// library route wrote the 2x2 patches as a [rows, 12 Cin] fp16 matrix (1.6 GB at 16 x 128^2 x 256), multiplied it by all four
// phases at once and interleaved the result with a pixel-shuffle pass (1 GB read + 1 GB written); here a block computes 8 x 32
// low-resolution positions x 128 output channels of ONE phase with the machinery of the stride-2 kernel above (9 x 33 patch,
// four taps per 16-channel chunk), reads x four times out of L2 / the Infinity Cache instead, writes every output pixel once
// and leaves the GroupNorm statistics of the result (the next ResnetBlock's norm1 needs them).  gridDim.y = 4 phases.
// Wf: [phase 2a + b][chunk][tap 2u + v][cout/32][2][64][8] -- _lib.upconv_weights_f16.
struct Upconv2Params {
  Conv3Params c;            // H, W: INPUT (low-resolution) size, tiles over it; y is [B][2H][2W][cout]
  const float *x;           // [B][H][W][cin]
  const float *scales_dev;  // {scale, 1 / (scale * u_scale)} or null
  float scale;
  int cin;
};

template <int COUT>
__global__ __launch_bounds__(256, 2) void upconv2x_f16x3_kernel(const Upconv2Params pp) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  Conv3Params p = pp.c;
  __shared__ __attribute__((aligned(16))) unsigned char sA[2 * kC3Buf];
  __shared__ int64_t red[kStatWords * 32];   // statistics records of the block's <= 32 groups (gq_stats.h)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  Conv3Tile t;
  if (!conv3_tile(p, t)) return;
  const int pa = (int)blockIdx.y >> 1, pb = (int)blockIdx.y & 1;
  red[tid] = 0;   // kStatWords * 32 = 256 words
  const float scale = pp.scales_dev ? pp.scales_dev[0] : pp.scale;
  if (pp.scales_dev) p.mscale = pp.scales_dev[1];
  const int cin = pp.cin, nch = p.nch, H = p.H, W = p.W;
  constexpr int PR = kC3TH + 1, PC = kC3TW + 1, NP = 5;      // patch 9 x 33 pixels; pieces per thread (297 x 4 / 256)
  const int w = tid & 3, qs = tid >> 2;
  const float *xb = pp.x + t.b * (long)H * W * cin + 4 * w;
  int goff[NP], loff[NP];
  unsigned inb = 0;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = qs + 64 * i;
    const int qq = q < PR * PC ? q : PR * PC - 1;
    const int R = qq / PC, X = qq % PC;
    const int gy = t.y0 + R + pa - 1, gx = t.x0 + X + pb - 1;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), cx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
    goff[i] = (cy * W + cx) * cin;
    loff[i] = q < PR * PC ? conv3_lds_off(R, X, w >> 1) + 8 * (w & 1) : -1;
    inb |= (unsigned)(in ? 1 : 0) << i;
  }
  f32x4 st[NP];
  auto issue = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NP; ++i) st[i] = *reinterpret_cast<const f32x4 *>(xb + goff[i] + chunk * 16);
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (loff[i] < 0) continue;
      const f32x4 v = st[i] * (((inb >> i) & 1) ? scale : 0.f);
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (_Float16)v[e];
        lo[e] = (_Float16)(v[e] - (float)hi[e]);
      }
      unsigned char *d = sA + buf * kC3Buf + loff[i];
      *reinterpret_cast<f16x4 *>(d) = hi;
      *reinterpret_cast<f16x4 *>(d + kC3Plane) = lo;
    }
  };
  f32x16 acc[4][2];
  GQ_C3_ZERO_ACC(acc);
  int aoff[2];
#pragma unroll
  for (int dx = 0; dx < 2; ++dx) aoff[dx] = conv3_lds_off(4 * wm, c + dx, h);
  const long wstep = (long)p.nnb * (512 * 16);   // bytes per k-step
  const int nks = 4 * nch;
  const unsigned char *wbase = reinterpret_cast<const unsigned char *>(p.Wf) + (long)blockIdx.y * nks * wstep +
                               (4 * t.nb + 2 * wn) * 128 * 16;
  const int wl = lane * 16;
  // weights two taps ahead (three register sets): vmcnt retires in order, so the consumer of a weight load also waits for the
  // x loads of the next chunk issued before it -- with one tap of lead those had one tap (~0.75 us) to come back
  f16x8 b0[4], b1[4], b2[4];
  auto load_b = [&](int ks, f16x8 (&dst)[4]) {
    const unsigned char *s = wbase + (long)(ks < nks ? ks : nks - 1) * wstep;
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = *reinterpret_cast<const f16x8 *>(s + k * 1024 + wl);
  };
  load_b(0, b0);
  load_b(1, b1);
  issue(0);
  commit(0);
  __syncthreads();
  int ks = 0;
  for (int chunk = 0; chunk < nch; ++chunk) {
    const bool more = chunk + 1 < nch;
    if (more) issue(chunk + 1);
    const unsigned char *A = sA + (chunk & 1) * kC3Buf;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {           // tap (u, v) = (ti >> 1, ti & 1)
      load_b(ks + 2, b2);
      conv3_tap(A + aoff[ti & 1], ti >> 1, b0, acc);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        b0[k] = b1[k];
        b1[k] = b2[k];
      }
      ++ks;
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) commit((chunk + 1) & 1);
    __syncthreads();
  }
  // ---- epilogue: register r of lane (c, h) = low-resolution position (y0 + 4 wm + rr, x0 + (r & 3) + 8 (r >> 2) + 4 h), i.e.
  // output pixel (2 y + pa, 2 x + pb); channel (2 wn + j) * 32 + c: every store instruction still writes 128-byte runs ----
  float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};
  constexpr int cout = COUT;
  const long W2 = 2L * W;
  const long pix0 = ((t.b * 2L * H + 2 * (t.y0 + 4 * wm) + pa) * W2 + 2 * (t.x0 + 4 * h) + pb) * cout + t.nb * 128;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = (2 * wn + j) * 32 + c;
    const float pbias = p.bias ? p.bias[t.nb * 128 + n] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[rr][j][r] * p.mscale + pbias;
        p.y[pix0 + ((long)rr * 2 * W2 + 2 * ((r & 3) + 8 * (r >> 2))) * cout + n] = v;
        s[j] += v;
        ss[j] += v * v;
      }
  }
  if (p.stats) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int g = ((2 * wn + j) * 32 + c) / p.cpg;      // group within this block's 128 channels
      stat_add_f32(red + kStatWords * g, s[j], ss[j]);
    }
    __syncthreads();
    const int gpb = 128 / p.cpg, groups = cout / p.cpg;    // groups per block, per image
    if (tid < kStatWords * gpb) stat_flush_word(p.stats + kStatWords * (t.b * groups + t.nb * gpb) + tid, red[tid]);
  }
}

// ---- 3x3 convolution (stride 1, zero padding 1) into a handful of channels (conv_out: 128 -> 3, unet.py:585-587) with the
// GroupNorm + SiLU of its input fused in: fp32 FMAs on the vector ALU.  With 3 output channels there is no GEMM to speak of
// (3456 FMAs per pixel, 7 GFLOP per 16 x 256 x 256 batch) and the job is to read the activation once: MIOpen's implicit GEMM
// takes 0.96 ms for it, after a 0.2 ms normalisation pass; this kernel is bound by its LDS reads near 0.25 ms.
// Block = 16 x 16 output pixels, thread = pixel.  Per 32-channel chunk the 18 x 18 patch is normalised, activated and staged
// in LDS (pixel stride 36 floats: 16 lanes reading 16 bytes each at that stride cover all 64 banks); the weights are uniform
// per instruction and come through scalar loads.  w: [COUT][3][3][Cin] (output channel, tap, input channel).
template <int SILU, int COUT>
__global__ __launch_bounds__(256) void conv3x3_gn_small_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               const float *__restrict__ pre_bias,
                                                               const int64_t *__restrict__ stats, const float *__restrict__ w,
                                                               const float *__restrict__ bias, float *__restrict__ y, int H,
                                                               int W, int C, int cpg, double eps) {
  constexpr int PW = 18, PS = 36;                       // patch width, floats per staged pixel
  __shared__ __attribute__((aligned(16))) float sX[PW * PW * PS];
  __shared__ __attribute__((aligned(16))) float sAff[2][512];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int tiles_x = W / 16, tiles_y = H / 16;
  const int txb = blockIdx.x % tiles_x, tyb = (blockIdx.x / tiles_x) % tiles_y;
  const long b = blockIdx.x / (tiles_x * tiles_y);
  const int y0 = tyb * 16, x0 = txb * 16;
  {
    const int groups = C / cpg;
    const double n = (double)cpg * (double)H * (double)W;
    for (int ch = tid; ch < C; ch += 256) {
      const int g = ch / cpg;
      double st_s, st_ss;
      stat_load(stats + kStatWords * (b * groups + g), st_s, st_ss);
      const double mean = st_s / n;
      double var = st_ss / n - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double rstd = 1.0 / sqrt(var + eps);
      const double pbk = pre_bias ? (double)pre_bias[ch] : 0.0;
      sAff[0][ch] = (float)(rstd * (double)gamma[ch]);
      sAff[1][ch] = (float)((double)beta[ch] + (pbk - mean) * rstd * (double)gamma[ch]);
    }
  }
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = bias ? bias[co] : 0.f;
  const float *xb = x + b * (long)H * W * C;
  __syncthreads();
  for (int c0 = 0; c0 < C; c0 += 32) {
    // stage: 324 pixels x 8 channel quads; borders read a clamped address and are zeroed (no branch around the load)
    for (int i = tid; i < PW * PW * 8; i += 256) {
      const int px = i >> 3, q = i & 7;
      const int R = px / PW, X = px % PW;
      const int gy = y0 - 1 + R, gx = x0 - 1 + X;
      const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), cx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
      const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + ((long)cy * W + cx) * C + c0 + 4 * q);
      const f32x4 a4 = *reinterpret_cast<const f32x4 *>(&sAff[0][c0 + 4 * q]);
      const f32x4 sh4 = *reinterpret_cast<const f32x4 *>(&sAff[1][c0 + 4 * q]);
      *reinterpret_cast<f32x4 *>(&sX[px * PS + 4 * q]) = gn_act<SILU>(v, a4, sh4) * (in ? 1.f : 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float *s = &sX[((ty + tap / 3) * PW + tx + tap % 3) * PS];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(s + 4 * q);
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          const float *wp = w + ((long)(co * 9 + tap) * C + c0 + 4 * q);    // uniform: scalar loads
          acc[co] = __builtin_fmaf(v.x, wp[0], acc[co]);
          acc[co] = __builtin_fmaf(v.y, wp[1], acc[co]);
          acc[co] = __builtin_fmaf(v.z, wp[2], acc[co]);
          acc[co] = __builtin_fmaf(v.w, wp[3], acc[co]);
        }
      }
    }
    __syncthreads();
  }
  float *yo = y + ((b * H + y0 + ty) * W + x0 + tx) * COUT;
#pragma unroll
  for (int co = 0; co < COUT; ++co) yo[co] = acc[co];
}

}  // namespace gqhip
