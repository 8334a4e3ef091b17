// Calibration 5: the split-bf16 filter's tile loop on the two bf16 MFMA shapes, same work per wave and tile
// (a 32-code x 64-row tile pair, K = 3 types x 32 slots):
//   SHAPE 0: 12 x v_mfma_f32_32x32x16_bf16 (what gq_filter_bf16_kernel issues at dim 16, RT = 2)
//   SHAPE 1: 24 x v_mfma_f32_16x16x32_bf16 (2 code blocks x 4 row blocks x 3 types)
// Both re-read their 4 code operand vectors (4 x ds_read_b128) from LDS every tile and run the v_max3 epilogue of
// the previous tile (16 v_max3 per tile).  One 512-thread block per CU (8 waves), random bf16 operands.
// MI355X_MICROARCH.md "DVFS give-back" item 7: the chip may hold a higher clock on the 16x16x32 shape.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_bf16_shapes tools/calibration/mfma_bf16_shapes.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void tile_loop(float *out, const u32x4 *in, int iters, unsigned long long *clk) {
  __shared__ u32x4 lds[2048];   // 32 KB: 8 tiles x 4 vectors x 64 lanes
  for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  bf16x8 b[8];   // row operands: 32x32: [rt][h|l parts x 2 k-halves]; 16x16: [row block][h|l]
  for (int s = 0; s < 8; ++s) b[s] = as_bf(in[(threadIdx.x * 9 + s) & 2047]);
  bf16x8 a[4];
  for (int s = 0; s < 4; ++s) a[s] = as_bf(in[(threadIdx.x * 5 + s) & 2047]);
  float t0 = -1e30f, t1 = -1e30f, t2 = -1e30f, t3 = -1e30f;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (SHAPE == 0) {
    f32x16 dp[2];
    for (int r = 0; r < 2; ++r) for (int k = 0; k < 16; ++k) dp[r][k] = -1e30f;
    for (int it = 0; it < iters; ++it) {
      const u32x4 *p = lds + (it & 7) * 256 + lane;
      a[0] = as_bf(p[0]); a[1] = as_bf(p[64]); a[2] = as_bf(p[128]); a[3] = as_bf(p[192]);
      f32x16 d[2];
      d[0] = d[1] = f32x16{0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f};
      constexpr int ca[6] = {0, 1, 2, 3, 0, 1}, cb[6] = {0, 1, 0, 1, 2, 3};
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        d[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ca[s]], b[cb[s]], d[0], 0, 0, 0);
        d[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ca[s]], b[4 + cb[s]], d[1], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
        t0 = __builtin_fmaxf(__builtin_fmaxf(t0, dp[0][k]), dp[0][k + 1]);
        t1 = __builtin_fmaxf(__builtin_fmaxf(t1, dp[1][k]), dp[1][k + 1]);
      }
      dp[0] = d[0]; dp[1] = d[1];
    }
    float s = 0.f;
    for (int r = 0; r < 2; ++r) for (int k = 0; k < 16; ++k) s += dp[r][k];
    t2 = s;
  } else {
    // code blocks cbk = 0,1 (16 codes each): operands a[2*cbk] (h parts), a[2*cbk+1] (l parts), K = 32 slots
    // row blocks rb = 0..3 (16 rows each): b[2*rb] (h parts), b[2*rb+1] (l parts)
    f32x4 dp[2][4];
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 4; ++r) dp[c][r] = f32x4{-1e30f, -1e30f, -1e30f, -1e30f};
    for (int it = 0; it < iters; ++it) {
      const u32x4 *p = lds + (it & 7) * 256 + lane;
      a[0] = as_bf(p[0]); a[1] = as_bf(p[64]); a[2] = as_bf(p[128]); a[3] = as_bf(p[192]);
      f32x4 d[2][4];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)   // hh, lh, hl
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            d[c][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * c + (ty == 1)], b[2 * r + (ty == 2)], d[c][r], 0, 0, 0);
      // epilogue on the previous tile: per row block one running maximum over 8 values (2 code blocks x 4)
      t0 = __builtin_fmaxf(__builtin_fmaxf(t0, dp[0][0][0]), dp[0][0][1]); t0 = __builtin_fmaxf(__builtin_fmaxf(t0, dp[0][0][2]), dp[0][0][3]);
      t0 = __builtin_fmaxf(__builtin_fmaxf(t0, dp[1][0][0]), dp[1][0][1]); t0 = __builtin_fmaxf(__builtin_fmaxf(t0, dp[1][0][2]), dp[1][0][3]);
      t1 = __builtin_fmaxf(__builtin_fmaxf(t1, dp[0][1][0]), dp[0][1][1]); t1 = __builtin_fmaxf(__builtin_fmaxf(t1, dp[0][1][2]), dp[0][1][3]);
      t1 = __builtin_fmaxf(__builtin_fmaxf(t1, dp[1][1][0]), dp[1][1][1]); t1 = __builtin_fmaxf(__builtin_fmaxf(t1, dp[1][1][2]), dp[1][1][3]);
      t2 = __builtin_fmaxf(__builtin_fmaxf(t2, dp[0][2][0]), dp[0][2][1]); t2 = __builtin_fmaxf(__builtin_fmaxf(t2, dp[0][2][2]), dp[0][2][3]);
      t2 = __builtin_fmaxf(__builtin_fmaxf(t2, dp[1][2][0]), dp[1][2][1]); t2 = __builtin_fmaxf(__builtin_fmaxf(t2, dp[1][2][2]), dp[1][2][3]);
      t3 = __builtin_fmaxf(__builtin_fmaxf(t3, dp[0][3][0]), dp[0][3][1]); t3 = __builtin_fmaxf(__builtin_fmaxf(t3, dp[0][3][2]), dp[0][3][3]);
      t3 = __builtin_fmaxf(__builtin_fmaxf(t3, dp[1][3][0]), dp[1][3][1]); t3 = __builtin_fmaxf(__builtin_fmaxf(t3, dp[1][3][2]), dp[1][3][3]);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) dp[c][r] = d[c][r];
    }
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 4; ++r) t3 += dp[c][r][0] + dp[c][r][1] + dp[c][r][2] + dp[c][r][3];
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 512 + threadIdx.x] = t0 + t1 + t2 + t3;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
void run(int blocks, float *out, const u32x4 *in, unsigned long long *clk, int rounds) {
  const int iters = 256;   // tiles per wave (config 2: 65536 codes / 8 splits / 32)
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 50; ++w) hipLaunchKernelGGL((tile_loop<SHAPE>), dim3(blocks), dim3(512), 0, 0, out, in, iters, clk);
  (void)hipEventRecord(e0);
  for (int w = 0; w < rounds; ++w) hipLaunchKernelGGL((tile_loop<SHAPE>), dim3(blocks), dim3(512), 0, 0, out, in, iters, clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 32 * 64 * 96 * (double)iters * blocks * 8;   // per launch: tile pair = 32 codes x 64 rows x K 96
  std::vector<unsigned long long> h(blocks * 2);
  (void)hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (int b = 0; b < blocks; ++b) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  const double us = ms / rounds * 1e3;
  printf("shape=%s blocks=%d: %.1f us/launch, %.3f PFLOP/s bf16 (%.1f%% of 2.5), in-kernel clock %.3f GHz, %.1f cycles per tile pair (nominal 384)\n",
         SHAPE == 0 ? "32x32x16" : "16x16x32", blocks, us, flop / (us * 1e-6) / 1e15, flop / (us * 1e-6) / 1e15 / 2.5 * 100,
         ghz[blocks / 2], us * 1e-6 * ghz[blocks / 2] * 1e9 / (iters * 2.0));
}

int main() {
  float *out; u32x4 *in; unsigned long long *clk;
  (void)hipMalloc(&out, 1024 * 512 * 4); (void)hipMalloc(&clk, 1024 * 16); (void)hipMalloc(&in, 2048 * 16);
  std::vector<unsigned> h(8192);
  unsigned s = 12345u;
  for (int i = 0; i < 8192; ++i) {   // two random bf16 values in (-2, 2) per word, random signs / mantissas
    s = s * 1664525u + 1013904223u; const unsigned lo = ((s >> 16) & 0x807fu) | 0x3f00u;
    s = s * 1664525u + 1013904223u; const unsigned hi = ((s >> 16) & 0x807fu) | 0x3f80u;
    h[i] = lo | (hi << 16);
  }
  (void)hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  // interleaved rounds in one process (cdna_hip_programming.md rule 24)
  for (int rep = 0; rep < 3; ++rep) {
    run<0>(256, out, in, clk, 2000);
    run<1>(256, out, in, clk, 2000);
  }
  return 0;
}
