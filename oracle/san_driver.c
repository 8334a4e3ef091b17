/* san_driver.c -- exercises every entry point of gq_oracle.c under AddressSanitizer + UndefinedBehaviorSanitizer on
 * the CPU (`make -C oracle san`; run by tests/test_oracle.py).  Test infrastructure only.  Shapes are chosen to hit the
 * edges the kernels' parity tests use: dims that are not multiples of 8, a single row, a codebook of one code, rows of
 * non-finite operands (torch.argmax semantics: the first NaN wins), ties.  Exit code 0 and "ok" when nothing fired. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void gq_oracle_nlp(const float *cb, float *nlp, int64_t n, int64_t dim);
void gq_oracle_scores(const float *mu, const float *sd, const float *lsd, const float *cb, const float *nlp, float *out,
                      int64_t dim, int64_t rows, int64_t n, float beta);
void gq_oracle_argmax(const float *mu, const float *sd, const float *lsd, const float *cb, const float *nlp, int64_t *idx,
                      float *zhat, float *best, float *second, int64_t dim, int64_t rows, int64_t n, float beta, int nthreads);
void gq_oracle_cuda_scores(const float *mu, const float *sd, const float *cb, float *out, int64_t dim, int64_t rows,
                           int64_t n, double beta);
void vq_oracle_argmin(const float *z, const float *emb, int64_t *idx, double *best, double *second, int64_t dim,
                      int64_t rows, int64_t n, int nthreads);
void lfq_oracle_pack(const float *x, int64_t *idx, int64_t rows, int64_t nbits);
int gq_oracle_abi_version(void);

static unsigned long long s = 88172645463325252ull;
static float rnd(void) {
  s ^= s << 13; s ^= s >> 7; s ^= s << 17;
  return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

int main(void) {
  const int64_t dims[] = {1, 3, 4, 7, 8, 9, 16, 17, 32, 64};
  const int64_t rowsv[] = {1, 5, 33};
  const int64_t nv[] = {1, 2, 31, 257};
  long checks = 0;
  for (unsigned di = 0; di < sizeof(dims) / sizeof(*dims); ++di)
    for (unsigned ri = 0; ri < sizeof(rowsv) / sizeof(*rowsv); ++ri)
      for (unsigned ni = 0; ni < sizeof(nv) / sizeof(*nv); ++ni) {
        const int64_t dim = dims[di], rows = rowsv[ri], n = nv[ni];
        float *mu = malloc(sizeof(float) * rows * dim), *sd = malloc(sizeof(float) * rows * dim);
        float *lsd = malloc(sizeof(float) * rows * dim), *cb = malloc(sizeof(float) * n * dim);
        float *out = malloc(sizeof(float) * rows * n), *nlp = malloc(sizeof(float) * n * dim);
        float *best = malloc(sizeof(float) * rows), *second = malloc(sizeof(float) * rows);
        double *bd = malloc(sizeof(double) * rows), *sd2 = malloc(sizeof(double) * rows);
        int64_t *idx = malloc(sizeof(int64_t) * rows);
        for (int64_t i = 0; i < rows * dim; ++i) {
          mu[i] = 2.0f * rnd();
          sd[i] = expf(1.5f * rnd());
          lsd[i] = logf(sd[i]);
        }
        for (int64_t i = 0; i < n * dim; ++i) cb[i] = 3.0f * rnd();
        if (rows > 1) {              /* a row of non-finite operands, a row of ties (every code equal) */
          mu[0] = NAN;
          sd[dim] = 0.0f;
          lsd[dim] = -INFINITY;
        }
        if (n > 2)
          for (int64_t k = 0; k < dim; ++k) cb[2 * dim + k] = cb[k];   /* code 2 == code 0: first index must win */
        gq_oracle_nlp(cb, nlp, n, dim);
        float *zhat = malloc(sizeof(float) * rows * dim);
        gq_oracle_scores(mu, sd, lsd, cb, nlp, out, dim, rows, n, 1.0f);
        gq_oracle_argmax(mu, sd, lsd, cb, nlp, idx, zhat, best, second, dim, rows, n, 1.0f, 2);
        for (int64_t r = 0; r < rows; ++r)
          if (idx[r] < 0 || idx[r] >= n) { printf("FAIL argmax index out of range\n"); return 1; }
        gq_oracle_argmax(mu, sd, lsd, cb, nlp, idx, NULL, NULL, NULL, dim, rows, n, 0.25f, 1);
        free(zhat);
        gq_oracle_cuda_scores(mu, sd, cb, out, dim, rows, n, 1.0);
        vq_oracle_argmin(mu, cb, idx, bd, sd2, dim, rows, n, 2);
        for (int64_t r = 0; r < rows; ++r)
          if (idx[r] < 0 || idx[r] >= n) { printf("FAIL argmin index out of range\n"); return 1; }
        vq_oracle_argmin(mu, cb, idx, NULL, NULL, dim, rows, n, 1);
        if (dim <= 62) lfq_oracle_pack(mu, idx, rows, dim);
        free(mu); free(sd); free(lsd); free(cb); free(out); free(nlp); free(best); free(second); free(bd); free(sd2); free(idx);
        ++checks;
      }
  if (gq_oracle_abi_version() != 1) return 1;
  printf("ok: %ld shape combinations under ASan + UBSan\n", checks);
  return 0;
}
