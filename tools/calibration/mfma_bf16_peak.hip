// Calibration 4: a loop shaped like a split-bf16 filter tile: 12 v_mfma_f32_32x32x16_bf16 (6 per row tile, 2 row
// tiles) per 32-code tile, adding one feature at a time:
//   F&1: the 4 code operand vectors re-read from LDS every tile (4 x ds_read_b128)
//   F&2: 16 v_max3 epilogue per tile (on the previous tile's accumulators)
// Prints the bf16 MFMA rate (dense peak 2.5 PFLOP/s = 32 cycles per 32x32x16 per SIMD at 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int F>
__global__ __launch_bounds__(256, 2) void tile_loop(float *out, const u32x4 *in, int iters, unsigned long long *clk) {
  __shared__ u32x4 lds[2048];   // 32 KB
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  bf16x8 b[2][4];
  for (int r = 0; r < 2; ++r) for (int s = 0; s < 4; ++s) b[r][s] = as_bf(in[(threadIdx.x * 9 + r * 4 + s) & 2047]);
  bf16x8 a[4];
  for (int s = 0; s < 4; ++s) a[s] = as_bf(in[(threadIdx.x * 5 + s) & 2047]);
  f32x16 dp[2];
  for (int r = 0; r < 2; ++r) for (int k = 0; k < 16; ++k) dp[r][k] = -1e30f;
  float t0 = -1e30f, t1 = -1e30f;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (F & 1) {
      const u32x4 *p = lds + (it & 7) * 256 + lane;
      a[0] = as_bf(p[0]); a[1] = as_bf(p[64]); a[2] = as_bf(p[128]); a[3] = as_bf(p[192]);
    }
    f32x16 d[2];
    d[0] = d[1] = f32x16{0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f};
    constexpr int ca[6] = {0, 1, 2, 3, 0, 1}, cb[6] = {0, 1, 0, 1, 2, 3};
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      d[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ca[s]], b[0][cb[s]], d[0], 0, 0, 0);
      d[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ca[s]], b[1][cb[s]], d[1], 0, 0, 0);
    }
    if (F & 2) {
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
        t0 = __builtin_fmaxf(__builtin_fmaxf(t0, dp[0][k]), dp[0][k + 1]);
        t1 = __builtin_fmaxf(__builtin_fmaxf(t1, dp[1][k]), dp[1][k + 1]);
      }
    }
    dp[0] = d[0]; dp[1] = d[1];
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = t0 + t1;
  for (int r = 0; r < 2; ++r) for (int k = 0; k < 16; ++k) s += dp[r][k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int F>
void run(int blocks, float *out, const u32x4 *in, unsigned long long *clk) {
  const int iters = 256 * 512 / blocks;   // 256 tiles per wave at 512 blocks (config 2)
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((tile_loop<F>), dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
  (void)hipEventRecord(e0);
  const int L = 20;
  for (int w = 0; w < L; ++w) hipLaunchKernelGGL((tile_loop<F>), dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double nm = 12.0 * iters * blocks * 4;   // MFMAs per launch
  std::vector<unsigned long long> h(blocks * 2);
  (void)hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (int b = 0; b < blocks; ++b) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  printf("blocks=%d F=%d (lds=%d epi=%d): %.1f us/launch, %.2f PFLOP/s bf16 (%.1f%% of 2.5), clock %.3f GHz, %.1f cycles/MFMA/SIMD\n",
         blocks, F, F & 1, (F >> 1) & 1, ms / L * 1e3, nm * 32768 / (ms / L * 1e-3) / 1e15,
         nm * 32768 / (ms / L * 1e-3) / 1e15 / 2.5 * 100, ghz[blocks / 2],
         (ms / L * 1e-3) * ghz[blocks / 2] * 1e9 / (nm / 1024));
}

int main() {
  float *out; u32x4 *in; unsigned long long *clk;
  (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&clk, 1024 * 16); (void)hipMalloc(&in, 2048 * 16);
  std::vector<unsigned> h(8192);
  for (int i = 0; i < 8192; ++i) {   // two bf16 values in [-1, 1) per word
    const unsigned lo = 0x3f00u + ((i * 2654435761u) >> 25), hi = 0xbf00u + ((i * 40503u) & 0x7f);
    h[i] = lo | (hi << 16);
  }
  (void)hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  for (int blocks : {256, 512}) {
    run<0>(blocks, out, in, clk);
    run<1>(blocks, out, in, clk);
    run<2>(blocks, out, in, clk);
    run<3>(blocks, out, in, clk);
  }
  return 0;
}
