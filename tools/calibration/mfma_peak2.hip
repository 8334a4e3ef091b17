// Calibration 2: a loop shaped like the filter kernel's tile (32 MFMAs = 2 chains x 16 k-steps with
// distinct A/B operand registers), adding one feature at a time:
//   F&1: A operands re-read from LDS every tile (4 x ds_read_b128)
//   F&2: 16 v_max3 epilogue cluster per tile (on the previous tile's accumulators)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int F>
__global__ __launch_bounds__(256, 2) void tile_loop(float *out, const float *in, int iters, unsigned long long *clk) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float b[2][16];
  for (int r = 0; r < 2; ++r) for (int s = 0; s < 16; ++s) b[r][s] = in[(threadIdx.x * 33 + r * 16 + s) & 8191];
  float a[16];
  for (int s = 0; s < 16; ++s) a[s] = in[(threadIdx.x * 17 + s) & 8191];
  f32x16 dp[2];
  for (int r = 0; r < 2; ++r) for (int k = 0; k < 16; ++k) dp[r][k] = -1e30f;
  float t0 = -1e30f, t1 = -1e30f;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (F & 1) {
      const float *p = lds + ((it & 3) * 1024) + (lane & 31) * 16 + (lane >> 5) * 8;
      const f32x4 v0 = *(const f32x4 *)(p), v1 = *(const f32x4 *)(p + 4);
      const f32x4 v2 = *(const f32x4 *)(p + 4096), v3 = *(const f32x4 *)(p + 4100);
      a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
      a[8] = v2.x; a[9] = v2.y; a[10] = v2.z; a[11] = v2.w; a[12] = v3.x; a[13] = v3.y; a[14] = v3.z; a[15] = v3.w;
    }
    f32x16 d[2];
    d[0] = d[1] = f32x16{0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      d[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[0][s], d[0], 0, 0, 0);
      d[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[1][s], d[1], 0, 0, 0);
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
void run(int blocks, float *out, const float *in, unsigned long long *clk) {
  const int iters = 256 * 512 / blocks;   // 256 tiles per wave at 512 blocks (the filter's config-2 launch)
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((tile_loop<F>), dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
  (void)hipEventRecord(e0);
  const int L = 20;
  for (int w = 0; w < L; ++w) hipLaunchKernelGGL((tile_loop<F>), dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double nm = 32.0 * iters * blocks * 4;   // MFMAs per launch
  std::vector<unsigned long long> h(blocks * 2);
  (void)hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (int b = 0; b < blocks; ++b) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  printf("blocks=%d F=%d (lds=%d epi=%d): %.1f us/launch, %.1f TFLOP/s (%.1f%%), clock %.3f GHz, %.1f cycles/MFMA/SIMD\n",
         blocks, F, F & 1, (F >> 1) & 1, ms / L * 1e3, nm * 4096 / (ms / L * 1e-3) / 1e12,
         nm * 4096 / (ms / L * 1e-3) / 1e12 / 157.3 * 100, ghz[blocks / 2],
         (ms / L * 1e-3) * ghz[blocks / 2] * 1e9 / (nm / 1024));
}

int main() {
  float *out, *in; unsigned long long *clk;
  (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&clk, 1024 * 16); (void)hipMalloc(&in, 8192 * 4);
  std::vector<float> h(8192);
  for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  (void)hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  for (int blocks : {256, 512}) {
    run<0>(blocks, out, in, clk);
    run<1>(blocks, out, in, clk);
    run<2>(blocks, out, in, clk);
    run<3>(blocks, out, in, clk);
  }
  return 0;
}
