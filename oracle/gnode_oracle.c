/*
 * C restatement of the GN-ODE hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Same algorithm as oracle/gnode_oracle.py (which is pinned against the golden
 * vectors the reference produced), written in plain C + OpenMP so that parity
 * checks at BASELINE sizes finish in seconds and so that bench.py has a fast
 * host-CPU "port" baseline.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load liboracle.so.  tests/test_oracle_c.py checks this file
 * against the numpy oracle; the integrator (torchdiffeq, absent) stays
 * "parity unpinned" exactly as stated there.
 *
 * Citations are file:line into the reference tree.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
void oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
void oracle_set_threads(int n) { (void)n; }
#endif

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* Z[r] = sigmoid(W X[r] + b): ode_nn_ngraph_sim.py:62-63 */
static void node_mlp(const float* X, const float* W, const float* b, float* Z, long nrows, int H) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < nrows; ++r) {
        const float* x = X + r * H;
        for (int j = 0; j < H; ++j) {
            const float* w = W + (long)j * H;
            float acc = 0.0f;
            for (int k = 0; k < H; ++k) acc += x[k] * w[k];
            Z[r * H + j] = sigmoidf_(acc + b[j]);
        }
    }
}

/* dS, dI, dR of ode_nn_ngraph_sim.py:68-77 for rows = B*n (implicit block-diagonal A). */
static void sir_derivative(const int32_t* rowptr, const int32_t* col, int n, long rows, int H, const float* Z,
                           const float* beta, const float* gamma, long bg_stride, float* d) {
    const long slab = rows * H;
#pragma omp parallel for schedule(dynamic, 64)
    for (long r = 0; r < rows; ++r) {
        const long node = r % n, base = r - node;
        const float* zs = Z + r * H;
        const float* zi = Z + slab + r * H;
        const float nb = -beta[r * bg_stride], gm = gamma[r * bg_stride];
        for (int h = 0; h < H; ++h) {
            float ai = 0.0f;
            for (int e = rowptr[node]; e < rowptr[node + 1]; ++e) ai += Z[slab + (base + col[e]) * H + h];
            const float dS = nb * (ai * zs[h]);
            const float dR = gm * zi[h];
            d[r * H + h] = dS;
            d[slab + r * H + h] = -dS - dR;
            d[2 * slab + r * H + h] = dR;
        }
    }
}

/* ODEfunc.forward: x, dx [4*rows, H] */
void oracle_rhs(const int32_t* rowptr, const int32_t* col, int n, long rows, int H, const float* x, const float* W,
                const float* b, float* dx) {
    const long slab = rows * H;
    float* Z = (float*)malloc(sizeof(float) * 2 * slab);
    node_mlp(x, W, b, Z, 2 * rows, H);
    sir_derivative(rowptr, col, n, rows, H, Z, x + 3 * slab, x + 3 * slab + 1, H, dx);
    memset(dx + 3 * slab, 0, sizeof(float) * slab);
    free(Z);
}

static void readout_rows(const float* Y, long rows, int H, const float* w3, const float* b3, const float* w2,
                         const float* b2, float* S, float* I, float* R) {
    const long slab = rows * H;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        float q[3];
        for (int c = 0; c < 3; ++c) {
            const float* y = Y + c * slab + r * H;
            float acc2 = b2[0];
            for (int k = 0; k < 4; ++k) {
                float a = b3[k];
                for (int h = 0; h < H; ++h) a += w3[k * H + h] * y[h];
                acc2 += w2[k] * (a > 0.0f ? a : 0.0f);
            }
            q[c] = acc2;
        }
        const float m = fmaxf(q[0], fmaxf(q[1], q[2]));
        const float e0 = expf(q[0] - m), e1 = expf(q[1] - m), e2 = expf(q[2] - m);
        const float inv = 1.0f / (e0 + e1 + e2);
        S[r] = e0 * inv; I[r] = e1 * inv; R[r] = e2 * inv;
    }
}

/* ODEBlock.forward with on-grid Euler: ode_nn_ngraph_sim.py:148-188.
 * x [rows, 3+H]; S, I, R [(n_steps+1), rows].  Returns seconds spent is the caller's business. */
void oracle_forward_euler(const int32_t* rowptr, const int32_t* col, int n, long rows, int H, const float* x,
                          const float* W, const float* b, const float* w1, const float* b1, const float* w3,
                          const float* b3, const float* w2, const float* b2, const float* dt, int n_steps, float* S,
                          float* I, float* R) {
    const long slab = rows * H;
    float* Y = (float*)malloc(sizeof(float) * 3 * slab);
    float* Z = (float*)malloc(sizeof(float) * 2 * slab);
    float* D = (float*)malloc(sizeof(float) * 3 * slab);
    float* bg = (float*)malloc(sizeof(float) * 2 * rows);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        const float* xr = x + r * (3 + H);
        for (int c = 0; c < 3; ++c)
            for (int h = 0; h < H; ++h) {
                const float v = w1[h] * xr[c] + b1[h];
                Y[c * slab + r * H + h] = v > 0.0f ? v : 0.0f;
            }
        bg[r] = xr[3]; bg[rows + r] = xr[4];
    }
    readout_rows(Y, rows, H, w3, b3, w2, b2, S, I, R);
    for (int k = 0; k < n_steps; ++k) {
        node_mlp(Y, W, b, Z, 2 * rows, H);
        sir_derivative(rowptr, col, n, rows, H, Z, bg, bg + rows, 1, D);
        const float h = dt[k];
#pragma omp parallel for schedule(static)
        for (long i = 0; i < 3 * slab; ++i) Y[i] = Y[i] + h * D[i];
        readout_rows(Y, rows, H, w3, b3, w2, b2, S + (long)(k + 1) * rows, I + (long)(k + 1) * rows,
                     R + (long)(k + 1) * rows);
    }
    free(Y); free(Z); free(D); free(bg);
}

/* ---- Philox4x32-10 + production Monte-Carlo (spec: gnode_oracle.py sir_philox) ---- */
static inline void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* the coin of item `pos` (CSR position, kind 0; node id, kind 1): four consecutive items share one block (gnode_oracle.py philox_coin) */
static inline uint32_t philox_coin(uint32_t pos, uint32_t it, uint32_t sim, uint32_t kind, uint32_t k0, uint32_t k1) {
    uint32_t w[4];
    philox_block(pos >> 2, it, sim, kind, k0, k1, w);
    return w[pos & 3u];
}

static uint64_t coin_threshold(double p) {
    double t = floor(p * 4294967296.0);
    if (t < 0.0) t = 0.0;
    if (t > 4294967296.0) t = 4294967296.0;
    return (uint64_t)t;
}

/* counts uint32 [3,T,n] accumulated for t>=1; row 0 of S/I assigned (ode_nn.py:55-56). */
void oracle_sir_philox(const int32_t* rowptr, const int32_t* col, int n, const int32_t* seeds, int n_seeds, double beta,
                       double gamma, long sims, long sim_offset, int T, uint64_t rng_seed, uint32_t* counts) {
    const uint32_t k0 = (uint32_t)(rng_seed & 0xFFFFFFFFu), k1 = (uint32_t)(rng_seed >> 32);
    const uint64_t tb = coin_threshold(beta), tg = coin_threshold(gamma);
    const size_t plane = (size_t)T * n;
#pragma omp parallel
    {
        uint8_t* st = (uint8_t*)malloc(n);
        uint8_t* fl = (uint8_t*)malloc(n);
        uint32_t* loc = (uint32_t*)calloc(3 * plane, sizeof(uint32_t));
#pragma omp for schedule(dynamic, 4)
        for (long s = 0; s < sims; ++s) {
            const uint32_t sim = (uint32_t)(sim_offset + s);
            memset(st, 0, n); memset(fl, 0, n);
            for (int j = 0; j < n_seeds; ++j) st[seeds[j]] = 1;
            for (int it = 1; it < T; ++it) {
                for (int u = 0; u < n; ++u) {
                    if (st[u] != 1) continue;
                    for (int e = rowptr[u]; e < rowptr[u + 1]; ++e) {
                        const int v = col[e];
                        if (st[v] == 0 && (uint64_t)philox_coin((uint32_t)e, (uint32_t)it, sim, 0u, k0, k1) < tb) fl[v] = 1;
                    }
                    if ((uint64_t)philox_coin((uint32_t)u, (uint32_t)it, sim, 1u, k0, k1) < tg) fl[u] = 2;
                }
                for (int v = 0; v < n; ++v) {
                    if (fl[v] == 1) st[v] = 1; else if (fl[v] == 2) st[v] = 2;
                    fl[v] = 0;
                    loc[st[v] * plane + (size_t)it * n + v] += 1;
                }
            }
        }
#pragma omp critical
        for (size_t i = 0; i < 3 * plane; ++i) counts[i] += loc[i];
        free(st); free(fl); free(loc);
    }
    for (int v = 0; v < n; ++v) { counts[v] = 1; counts[plane + v] = 0; }
    for (int j = 0; j < n_seeds; ++j) { counts[seeds[j]] = 0; counts[plane + seeds[j]] = 1; }
}
