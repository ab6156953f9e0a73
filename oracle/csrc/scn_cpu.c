/* TEST INFRASTRUCTURE — C/OpenMP kernels for the CPU oracle's hot loops ("CPU stand-in with the same
 * algorithm as upstream's CPU path: explicit rulebook, per-offset gather -> small GEMM -> scatter-add, OpenMP").
 * Used only when scn_oracle.FAST is switched on (bench.py's cpu_baseline leg); the default oracle path — the one the
 * golden fixtures and all parity tests anchor on — stays the torch-op restatement, and tests/test_oracle_fast.py holds
 * this path to it.  Restates: rule_conv (oracle/scn_oracle/__init__.py, per-offset x[in] @ W[k] added into y[out]) and
 * Grid.subm_rules (27 neighbour look-ups per site in the sorted key table).
 * Within one offset every output row (forward) / input row (data gradient) occurs at most once, so rule pairs can be
 * processed in parallel without atomics; the weight gradient reduces thread-local tiles in thread order. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

/* y[out[p]] += x[in[p]] (1 x cin) @ w (cin x cout) */
void scn_conv_fwd(const float *x, int cin, const float *w, int cout, const int64_t *in, const int64_t *out,
                  int64_t np, float *y) {
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < np; ++p) {
    const float *xr = x + in[p] * cin;
    float *yr = y + out[p] * cout;
    for (int c = 0; c < cin; ++c) {
      const float xv = xr[c];
      const float *wr = w + (int64_t)c * cout;
      for (int n = 0; n < cout; ++n) yr[n] += xv * wr[n];
    }
  }
}

/* dx[in[p]] += dy[out[p]] (1 x cout) @ w^T   (w is transposed once so that the inner loop is an axpy) */
void scn_conv_bwd_x(const float *dy, int cout, const float *w, int cin, const int64_t *in, const int64_t *out,
                    int64_t np, float *dx) {
  float *wt = (float *)malloc((size_t)cin * cout * sizeof(float));
  if (!wt) return;
  for (int c = 0; c < cin; ++c)
    for (int n = 0; n < cout; ++n) wt[(size_t)n * cin + c] = w[(size_t)c * cout + n];
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < np; ++p) {
    const float *dr = dy + out[p] * cout;
    float *xr = dx + in[p] * cin;
    for (int n = 0; n < cout; ++n) {
      const float dv = dr[n];
      const float *tr = wt + (size_t)n * cin;
      for (int c = 0; c < cin; ++c) xr[c] += dv * tr[c];
    }
  }
  free(wt);
}

/* dw (cin x cout) += sum_p x[in[p]]^T dy[out[p]] */
void scn_conv_bwd_w(const float *x, int cin, const float *dy, int cout, const int64_t *in, const int64_t *out,
                    int64_t np, float *dw) {
  const int nt = omp_get_max_threads();
  const size_t tile = (size_t)cin * cout;
  float *acc = (float *)calloc((size_t)nt * tile, sizeof(float));
  if (!acc) return;
#pragma omp parallel
  {
    float *mine = acc + (size_t)omp_get_thread_num() * tile;
#pragma omp for schedule(static)
    for (int64_t p = 0; p < np; ++p) {
      const float *xr = x + in[p] * cin;
      const float *dr = dy + out[p] * cout;
      for (int c = 0; c < cin; ++c) {
        const float xv = xr[c];
        float *mr = mine + (size_t)c * cout;
        for (int n = 0; n < cout; ++n) mr[n] += xv * dr[n];
      }
    }
  }
  for (int t = 0; t < nt; ++t)
    for (size_t e = 0; e < tile; ++e) dw[e] += acc[(size_t)t * tile + e];
  free(acc);
}

static inline int64_t find_key(const int64_t *keys, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  return (lo < n && keys[lo] == key) ? lo : -1;
}

/* nbr[k*n + j] = row of the site at coords[j] + d_k (k = (dz+1)*9 + (dy+1)*3 + (dx+1)), or -1.
 * coords (n,4) int64 z,y,x,b; sorted_keys / order: the grid's sorted key table (key = b<<48|z<<32|y<<16|x). */
void scn_subm_rules(const int64_t *coords, int64_t n, const int64_t *sorted_keys, const int64_t *order, int64_t *nbr) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n; ++j) {
    const int64_t z = coords[4 * j], y = coords[4 * j + 1], x = coords[4 * j + 2], b = coords[4 * j + 3];
    int k = 0;
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx, ++k) {
          const int64_t qz = z + dz, qy = y + dy, qx = x + dx;
          int64_t r = -1;
          if (qz >= 0 && qy >= 0 && qx >= 0 && qz <= 65535 && qy <= 65535 && qx <= 65535) {
            const int64_t pos = find_key(sorted_keys, n, (b << 48) | (qz << 32) | (qy << 16) | qx);
            if (pos >= 0) r = order[pos];
          }
          nbr[(int64_t)k * n + j] = r;
        }
  }
}

/* ---------------------------------------------------------------------------------------------------------------
 * Table form of the same convolution: tab[k*n_out + j] = input row feeding output row j through offset k (-1 = no rule)
 * — the rulebook as an offset-major neighbour table.  One parallel region per convolution (rows are independent), no
 * barrier between offsets; every output element is accumulated over (k, c) in the same order as the per-offset form
 * above, so the results are identical to it.  What bench.py's cpu_baseline leg times (27 fork/joins per
 * convolution made 128 threads slower than one).
 * ------------------------------------------------------------------------------------------------------------- */
/* y[j] = sum_k x[tab[k][j]] @ w[k]      (transpose_w: w[k] is (cout_layer = cin here ... ) see scn_conv_tab_t) */
void scn_conv_tab(const float *x, int cin, const float *w, int cout, int K, const int64_t *tab, int64_t n_out, float *y) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n_out; ++j) {
    float *yr = y + j * cout;
    for (int k = 0; k < K; ++k) {
      const int64_t i = tab[(int64_t)k * n_out + j];
      if (i < 0) continue;
      const float *xr = x + i * cin;
      const float *wk = w + (int64_t)k * cin * cout;
      for (int c = 0; c < cin; ++c) {
        const float xv = xr[c];
        const float *wr = wk + (int64_t)c * cout;
        for (int n = 0; n < cout; ++n) yr[n] += xv * wr[n];
      }
    }
  }
}

/* dx[i] = sum_k dy[tab[k][i]] @ w[kmap(k)]^T, w (K, cin, cout); flip: kmap(k) = K-1-k (3x3x3: the mirrored table) */
void scn_conv_tab_t(const float *dy, int cout, const float *w, int cin, int K, const int64_t *tab, int64_t n_in, int flip,
                    float *dx) {
  float *wt = (float *)malloc((size_t)K * cin * cout * sizeof(float));
  if (!wt) return;
  for (int k = 0; k < K; ++k)
    for (int c = 0; c < cin; ++c)
      for (int n = 0; n < cout; ++n) wt[((size_t)k * cout + n) * cin + c] = w[((size_t)k * cin + c) * cout + n];
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_in; ++i) {
    float *xr = dx + i * cin;
    for (int k = 0; k < K; ++k) {
      const int64_t j = tab[(int64_t)k * n_in + i];
      if (j < 0) continue;
      const float *dr = dy + j * cout;
      const float *tk = wt + (size_t)(flip ? (K - 1 - k) : k) * cout * cin;
      for (int n = 0; n < cout; ++n) {
        const float dv = dr[n];
        const float *tr = tk + (size_t)n * cin;
        for (int c = 0; c < cin; ++c) xr[c] += dv * tr[c];
      }
    }
  }
  free(wt);
}

/* dw[k] (cin x cout) = sum_j x[tab[k][j]]^T dy[j]: thread-local (K, cin, cout) tiles, reduced in thread order */
void scn_conv_tab_w(const float *x, int cin, const float *dy, int cout, int K, const int64_t *tab, int64_t n_out, float *dw) {
  const int nt = omp_get_max_threads();
  const size_t tile = (size_t)K * cin * cout;
  float *acc = (float *)calloc((size_t)nt * tile, sizeof(float));
  if (!acc) return;
#pragma omp parallel
  {
    float *mine = acc + (size_t)omp_get_thread_num() * tile;
#pragma omp for schedule(static)
    for (int64_t j = 0; j < n_out; ++j) {
      const float *dr = dy + j * cout;
      for (int k = 0; k < K; ++k) {
        const int64_t i = tab[(int64_t)k * n_out + j];
        if (i < 0) continue;
        const float *xr = x + i * cin;
        float *mk = mine + (size_t)k * cin * cout;
        for (int c = 0; c < cin; ++c) {
          const float xv = xr[c];
          float *mr = mk + (size_t)c * cout;
          for (int n = 0; n < cout; ++n) mr[n] += xv * dr[n];
        }
      }
    }
  }
  for (int t = 0; t < nt; ++t)
    for (size_t e = 0; e < tile; ++e) dw[e] += acc[(size_t)t * tile + e];
  free(acc);
}

int scn_cpu_threads(void) { return omp_get_max_threads(); }
