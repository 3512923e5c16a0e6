/* ORACLE — TEST INFRASTRUCTURE ONLY. Self-test driver of oracle/ref_c.c for the sanitizer build
 * (gcc -fsanitize=address,undefined): runs every function on seeded inputs including the edge shapes the reference's
 * callers produce (one point, ragged N != M, K == M, duplicate points, an empty ball) with exactly-sized heap buffers,
 * checks the invariants that do not need a second implementation, and prints one checksum line per case that
 * tests/test_oracle_c_cpu.py compares with oracle/ref_numpy.py on the same inputs. Exit code 0 + "OK" = clean. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void refc_nn_bidir(const float*, int, const float*, int, double*, int64_t*, double*, int64_t*);
double refc_chamfer(const float*, int, const float*, int, double*, int64_t*, double*, int64_t*);
double refc_sgd_hausdorff(const float*, int, const float*, int, double*, int64_t*, double*, int64_t*);
double refc_bid_hausdorff(const float*, int, const float*, int, double*, int64_t*, double*, int64_t*);
void refc_knn(const float*, int, const float*, int, int, double*, int64_t*);
void refc_fps(const float*, int, int, int, int64_t*, float*);
void refc_ball_query(const float*, int, const float*, int, float, int, int64_t*);

static uint32_t lcg_state;
static float lcg_unit(void) { /* the generator tests/test_oracle_c_cpu.py mirrors: 24 random bits -> [0,1) */
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return (float)(lcg_state >> 8) / 16777216.0f;
}
static float* cloud(int n, uint32_t seed, int dup) {
  float* p = (float*)malloc(sizeof(float) * 3 * (size_t)n);
  lcg_state = seed;
  for (int i = 0; i < 3 * n; ++i) p[i] = lcg_unit() - 0.5f;
  if (dup && n > 3)
    for (int k = 0; k < 3; ++k) p[3 * (n - 1) + k] = p[k], p[3 * 2 + k] = p[3 * 1 + k]; /* exact duplicates -> ties */
  return p;
}
static int fail(const char* what) {
  printf("FAIL %s\n", what);
  return 1;
}

int main(void) {
  static const int shapes[][2] = {{1, 1}, {1, 7}, {7, 1}, {64, 64}, {100, 37}, {257, 512}, {1024, 1024}};
  for (size_t c = 0; c < sizeof(shapes) / sizeof(shapes[0]); ++c) {
    const int N = shapes[c][0], M = shapes[c][1];
    float* a = cloud(N, 11u + (uint32_t)c, 1);
    float* b = cloud(M, 101u + (uint32_t)c, 1);
    double* dA = (double*)malloc(sizeof(double) * (size_t)N);
    double* dB = (double*)malloc(sizeof(double) * (size_t)M);
    int64_t* iA = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
    int64_t* iB = (int64_t*)malloc(sizeof(int64_t) * (size_t)M);
    refc_nn_bidir(a, N, b, M, dA, iA, dB, iB);
    double s = 0.0;
    int64_t si = 0;
    for (int i = 0; i < N; ++i) {
      if (iA[i] < 0 || iA[i] >= M || !(dA[i] >= 0.0)) return fail("nn range A");
      s += dA[i], si += iA[i];
    }
    for (int j = 0; j < M; ++j) {
      if (iB[j] < 0 || iB[j] >= N || !(dB[j] >= 0.0)) return fail("nn range B");
      s += 2.0 * dB[j], si += 3 * iB[j];
    }
    const double ch = refc_chamfer(a, N, b, M, dA, iA, dB, iB);
    const double h1 = refc_sgd_hausdorff(a, N, b, M, dA, iA, dB, iB);
    const double h2 = refc_bid_hausdorff(a, N, b, M, dA, iA, dB, iB);
    if (h2 + 1e-15 < h1) return fail("hausdorff order");
    { /* self distance: every point is its own nearest neighbour at distance 0 */
      double* d2 = (double*)malloc(sizeof(double) * (size_t)N);
      int64_t* i2 = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
      refc_nn_bidir(a, N, a, N, dA, iA, d2, i2);
      for (int i = 0; i < N; ++i)
        if (dA[i] != 0.0 || d2[i] != 0.0) return fail("self distance");
      free(d2), free(i2);
    }
    printf("nn %d %d %.17g %lld %.17g %.17g %.17g\n", N, M, s, (long long)si, ch, h1, h2);
    /* kNN incl. K == M */
    const int Ks[3] = {1, M < 5 ? M : 5, M};
    for (int t = 0; t < 3; ++t) {
      const int K = Ks[t];
      double* d = (double*)malloc(sizeof(double) * (size_t)N * K);
      int64_t* id = (int64_t*)malloc(sizeof(int64_t) * (size_t)N * K);
      refc_knn(a, N, b, M, K, d, id);
      double sk = 0.0;
      int64_t sik = 0;
      for (int i = 0; i < N; ++i)
        for (int k = 0; k < K; ++k) {
          if (k && d[(size_t)i * K + k] < d[(size_t)i * K + k - 1]) return fail("knn order");
          if (id[(size_t)i * K + k] < 0 || id[(size_t)i * K + k] >= M) return fail("knn range");
          sk += d[(size_t)i * K + k] * (k + 1), sik += id[(size_t)i * K + k] * (k + 1);
        }
      printf("knn %d %d %d %.17g %lld\n", N, M, K, sk, (long long)sik);
      free(d), free(id);
    }
    /* FPS + ball query on a (the cloud samples itself) */
    const int S = N < 16 ? N : 16, ns = 8;
    int64_t* f = (int64_t*)malloc(sizeof(int64_t) * (size_t)S);
    float* ws = (float*)malloc(sizeof(float) * (size_t)N);
    refc_fps(a, N, S, 0, f, ws);
    int64_t sf = 0;
    for (int s2 = 0; s2 < S; ++s2) {
      if (f[s2] < 0 || f[s2] >= N) return fail("fps range");
      sf += f[s2] * (s2 + 1);
    }
    float* ctr = (float*)malloc(sizeof(float) * 3 * (size_t)(S + 1));
    for (int s2 = 0; s2 < S; ++s2)
      for (int k = 0; k < 3; ++k) ctr[3 * s2 + k] = a[3 * f[s2] + k];
    ctr[3 * S] = ctr[3 * S + 1] = ctr[3 * S + 2] = 100.f; /* a centre with an empty ball */
    int64_t* bq = (int64_t*)malloc(sizeof(int64_t) * (size_t)(S + 1) * ns);
    refc_ball_query(a, N, ctr, S + 1, 0.2f, ns, bq);
    int64_t sb = 0;
    for (int e = 0; e < (S + 1) * ns; ++e) {
      if (bq[e] < 0 || bq[e] > N) return fail("ball range");
      sb += bq[e] * (e % 7 + 1);
    }
    for (int k = 0; k < ns; ++k)
      if (bq[(size_t)S * ns + k] != N) return fail("empty ball marker");
    printf("fps %d %d %lld ball %lld\n", N, S, (long long)sf, (long long)sb);
    free(a), free(b), free(dA), free(dB), free(iA), free(iB), free(f), free(ws), free(ctr), free(bq);
  }
  printf("OK\n");
  return 0;
}
