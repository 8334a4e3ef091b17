// Operand layout and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands, checked with exact
// small-integer data against two hypotheses for the K index of byte j (0..31) of lane (r = l & 31, h = l >> 5):
//   H1: k = 32 h + j          H2: k = 16 h + (j & 15) + 32 (j >> 4)
// and: does lane l's scale byte apply to the 32 K values that lane holds?  (opsel = byte index inside the scale VGPR)
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/mfma_f8_layout tools/calibration/mfma_f8_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void k(const unsigned char *a, const unsigned char *b, float *d, const int *sa, const int *sb, int opa, int opb) {
  const int l = threadIdx.x;
  i32x8 av, bv;
  for (int w = 0; w < 8; ++w) {
    av[w] = ((const int *)a)[l * 8 + w];
    bv[w] = ((const int *)b)[l * 8 + w];
  }
  f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 r;
  if (opa == 0 && opb == 0) r = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, sa[l], 0, sb[l]);
  else if (opa == 1 && opb == 0) r = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 1, sa[l], 0, sb[l]);
  else r = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 1, sa[l], 1, sb[l]);
  for (int i = 0; i < 16; ++i) d[l * 16 + i] = r[i];
}

static unsigned char enc(int v) {   // e4m3 encoding of small integers -4..4
  static const unsigned char t[5] = {0x00, 0x38, 0x40, 0x44, 0x48};   // 0, 1, 2, 3, 4
  return v < 0 ? (unsigned char)(0x80 | t[-v]) : t[v];
}

int main() {
  std::vector<int> A(32 * 64), B(64 * 32);
  srand(1);
  for (auto &x : A) x = rand() % 7 - 3;
  for (auto &x : B) x = rand() % 7 - 3;
  for (int hyp = 1; hyp <= 2; ++hyp) {
    std::vector<unsigned char> a(64 * 32), b(64 * 32);
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 32; ++j) {
        const int r = l & 31, h = l >> 5;
        const int kk = hyp == 1 ? 32 * h + j : 16 * h + (j & 15) + 32 * (j >> 4);
        a[l * 32 + j] = enc(A[r * 64 + kk]);     // A operand: row r
        b[l * 32 + j] = enc(B[kk * 32 + r]);     // B operand: column r
      }
    for (int test = 0; test < 4; ++test) {
      // test 0: all scales 2^0; 1: A scale 2^1 on lanes h = 0 (byte 0); 2: the same through byte 1 + opsel 1; 3: B scale 2^-2 on h = 1, byte 1
      std::vector<int> sa(64, 127), sb(64, 127);
      int opa = 0, opb = 0;
      double fa[2] = {1, 1}, fb[2] = {1, 1};
      if (test == 1) { for (int l = 0; l < 32; ++l) sa[l] = 128; fa[0] = 2; }
      if (test == 2) { for (int l = 0; l < 64; ++l) sa[l] = 127 | ((l < 32 ? 128 : 127) << 8) | (0x55 << 16); opa = 1; fa[0] = 2; }
      if (test == 3) { for (int l = 0; l < 64; ++l) { sa[l] = 127 | (127 << 8); sb[l] = 130 | ((l >= 32 ? 125 : 127) << 8); } opa = 1; opb = 1; fb[1] = 0.25; }
      unsigned char *da, *db; float *dd; int *dsa, *dsb;
      hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dd, 64 * 16 * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
      hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
      hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd, dsa, dsb, opa, opb);
      std::vector<float> d(64 * 16);
      hipMemcpy(d.data(), dd, 64 * 16 * 4, hipMemcpyDeviceToHost);
      // reference with block-of-32 scales: K block = kk / 32 under the hypothesis that lane half h holds block h (H1) --
      // for H2 a lane holds halves of both blocks, so "the lane's scale applies to its values" is evaluated per value
      int bad = 0;
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 16; ++i) {
          const int col = l & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
          double ref = 0;
          for (int kk = 0; kk < 64; ++kk) {
            int hl;   // which lane half holds this k
            if (hyp == 1) hl = kk / 32; else hl = (kk % 32) / 16;
            ref += A[row * 64 + kk] * B[kk * 32 + col] * fa[hl] * fb[hl];
          }
          if (ref != d[l * 16 + i]) ++bad;
        }
      // alternatives: (u) scales ignored, (b) scale by K block kk / 32 taken from lane (row | col) + 32 * block under both layouts
      int bad_u = 0, bad_b = 0;
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 16; ++i) {
          const int col = l & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
          double ru = 0, rb = 0;
          for (int kk = 0; kk < 64; ++kk) {
            const double pr = A[row * 64 + kk] * B[kk * 32 + col];
            ru += pr;
            rb += pr * fa[kk / 32] * fb[kk / 32];
          }
          if (ru != d[l * 16 + i]) ++bad_u;
          if (rb != d[l * 16 + i]) ++bad_b;
        }
      printf("hypothesis H%d, test %d: %d of 1024 results differ (scales ignored: %d, scale by K block kk/32: %d)  d[0..3] = %g %g %g %g\n", hyp, test, bad, bad_u, bad_b, d[0], d[1], d[2], d[3]);
    }
  }
  return 0;
}
