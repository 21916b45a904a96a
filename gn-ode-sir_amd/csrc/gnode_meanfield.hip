// Mean-field SIR baseline on MI355X (gfx950) -- SURVEY 8f rank 4.
//
// The reference integrates  dS = -beta (A I) * S,  dI = beta (A I) * S - gamma * I,  dR = gamma * I  (ode_nn.py:214-220,
// `sir`) with scipy's LSODA on the DENSE adjacency and samples it at t = 0, 1, .., maxTime-1 (`runge_kutta_order4`,
// :222-233).  Here: float64 state on the device, A I as a CSR sparse mat-vec, and an adaptive Dormand-Prince 5(4)
// pair whose steps are clipped to land exactly on the output times.  The step-size control runs on the host (one
// 8-byte error norm per step comes back); tolerances default far below LSODA's own (rtol = atol = 1.5e-8), so the
// two agree to ~1e-7 -- the test tolerance is 1e-6 absolute on probabilities.
#include "gnode_common.h"
#include <algorithm>
#include <cmath>
#include <cstring>

struct MfCoef { double c[7]; int n; };

// ytmp = y + h * sum_j c_j k_j        (3n doubles; k_j = K + j * 3n)
__global__ __launch_bounds__(256) void k_mf_comb(const double* __restrict__ y, const double* __restrict__ K, MfCoef cf, double h,
                                                long len, double* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    double s = 0.0;
    for (int j = 0; j < cf.n; ++j) s += cf.c[j] * K[(size_t)j * len + i];
    out[i] = y[i] + h * s;
}

// k = f(y)
__global__ __launch_bounds__(256) void k_mf_rhs(const int* __restrict__ rowptr, const int* __restrict__ col, int n, double beta,
                                               const double* __restrict__ gamma, const double* __restrict__ y,
                                               double* __restrict__ k) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    const double* I = y + n;
    double ai = 0.0;
    for (int e = rowptr[u]; e < rowptr[u + 1]; ++e) ai += I[col[e]];
    const double inf = beta * (ai * y[u]), rec = gamma[u] * I[u];
    k[u] = -inf; k[n + u] = inf - rec; k[2 * (size_t)n + u] = rec;
}

// err = max_i |h sum_j e_j k_j| / (atol + rtol max(|y_i|, |ynew_i|))   (non-negative doubles order like their bits)
__global__ __launch_bounds__(256) void k_mf_err(const double* __restrict__ y, const double* __restrict__ ynew,
                                               const double* __restrict__ K, MfCoef ef, double h, double rtol, double atol,
                                               long len, unsigned long long* __restrict__ err_bits) {
    __shared__ double red[256];
    double m = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < len; i += (long)gridDim.x * 256) {
        double s = 0.0;
        for (int j = 0; j < ef.n; ++j) s += ef.c[j] * K[(size_t)j * len + i];
        const double sc = atol + rtol * fmax(fabs(y[i]), fabs(ynew[i]));
        m = fmax(m, fabs(h * s) / sc);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(err_bits, (unsigned long long)__double_as_longlong(red[0]));
}

__global__ __launch_bounds__(256) void k_mf_init(const double* __restrict__ seed, int n, double* __restrict__ y) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    y[u] = 1.0 - seed[u]; y[n + u] = seed[u]; y[2 * (size_t)n + u] = 0.0;
}

// out rows (I, S, R order of the reference's return value, ode_nn.py:233): out[c][t][u]
__global__ __launch_bounds__(256) void k_mf_emit(const double* __restrict__ y, int n, int t, int T, double* __restrict__ outI,
                                                double* __restrict__ outS, double* __restrict__ outR) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    outS[(size_t)t * n + u] = y[u]; outI[(size_t)t * n + u] = y[n + u]; outR[(size_t)t * n + u] = y[2 * (size_t)n + u];
}

extern "C" size_t gnode_meanfield_workspace_bytes(gnode_graph_t g) {
    if (!g) return 0;
    const size_t v = gn_align((size_t)3 * g->n * sizeof(double));
    return 10 * v + gn_align((size_t)g->n * sizeof(double)) + 256;      // y, ynew, ytmp, K[7] | seed | err
}

extern "C" int gnode_meanfield_f64(gnode_graph_t g, const int32_t* seeds_host, int32_t n_seeds, double beta,
                                   const double* gamma, const double* t_out_host, int32_t n_out, double rtol, double atol,
                                   double* outI, double* outS,
                                   double* outR, int64_t* steps_host, void* workspace, size_t workspace_bytes, void* stream) {
    GN_CHECK_ARG(g && gamma && outI && outS && outR && workspace && (seeds_host || n_seeds == 0), "gnode_meanfield_f64: null pointer");
    GN_CHECK_ARG(t_out_host && n_out >= 1 && t_out_host[0] == 0.0, "gnode_meanfield_f64: need output times starting at 0");
    for (int i = 1; i < n_out; ++i)
        GN_CHECK_ARG(t_out_host[i] >= t_out_host[i - 1], "gnode_meanfield_f64: output times must be ascending");
    GN_CHECK_ARG(rtol > 0 && atol > 0, "gnode_meanfield_f64: tolerances must be positive");
    for (int i = 0; i < n_seeds; ++i)
        GN_CHECK_ARG(seeds_host[i] >= 0 && seeds_host[i] < g->n, "gnode_meanfield_f64: seed %d out of range", seeds_host[i]);
    if (workspace_bytes < gnode_meanfield_workspace_bytes(g)) {
        gnode_set_error("gnode_meanfield_f64: workspace %zu < %zu", workspace_bytes, gnode_meanfield_workspace_bytes(g));
        return GNODE_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int n = g->n;
    const long len = 3L * n;
    const size_t v = gn_align((size_t)len * sizeof(double));
    char* ws = (char*)workspace;
    double* y = (double*)ws; double* ynew = (double*)(ws + v); double* ytmp = (double*)(ws + 2 * v);
    double* K = (double*)(ws + 3 * v);                                // K[j] = K + j * len needs contiguity in `len` units
    double* seed = (double*)(ws + 10 * v);
    unsigned long long* err = (unsigned long long*)(ws + 10 * v + gn_align((size_t)n * sizeof(double)));
    // K is indexed K[j * len + i]: lay the 7 stages out back to back in elements (7 * len doubles fit in 7 aligned slots)
    GN_HIP(hipMemsetAsync(seed, 0, (size_t)n * sizeof(double), st));
    const double one = 1.0;
    for (int i = 0; i < n_seeds; ++i) GN_HIP(hipMemcpyAsync(seed + seeds_host[i], &one, sizeof(double), hipMemcpyHostToDevice, st));
    const unsigned ng = (unsigned)((n + 255) / 256), lg = (unsigned)((len + 255) / 256);
    hipLaunchKernelGGL(k_mf_init, dim3(ng), dim3(256), 0, st, seed, n, y);
    hipLaunchKernelGGL(k_mf_emit, dim3(ng), dim3(256), 0, st, y, n, 0, n_out, outI, outS, outR);
    GN_LAUNCH_CHECK();
    GN_HIP(hipStreamSynchronize(st));                                 // `one` / seeds_host are done with

    // Dormand-Prince 5(4)
    static const double A[7][6] = {
        {0, 0, 0, 0, 0, 0},
        {1.0 / 5, 0, 0, 0, 0, 0},
        {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
        {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
        {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
        {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
        {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
    static const double B4[7] = {5179.0 / 57600, 0, 7571.0 / 16695, 393.0 / 640, -92097.0 / 339200, 187.0 / 2100, 1.0 / 40};
    auto rhs = [&](const double* yy, double* kk) {
        hipLaunchKernelGGL(k_mf_rhs, dim3(ng), dim3(256), 0, st, g->rowptr, g->col, n, beta, gamma, yy, kk);
    };
    rhs(y, K);                                                        // k1 (FSAL: later steps reuse k7)
    double t = 0.0, h = 1e-3;
    long steps = 0;
    for (int to = 1; to < n_out; ++to) {
        const double t_end = t_out_host[to];
        while (t < t_end) {
            const double hh = std::min(h, t_end - t);
            for (int s = 1; s < 7; ++s) {
                MfCoef cf; cf.n = s;
                for (int j = 0; j < s; ++j) cf.c[j] = A[s][j];
                double* dst = (s == 6) ? ynew : ytmp;                 // stage 7's argument IS the 5th-order solution
                hipLaunchKernelGGL(k_mf_comb, dim3(lg), dim3(256), 0, st, y, K, cf, hh, len, dst);
                rhs(dst, K + (size_t)s * len);
            }
            MfCoef ef; ef.n = 7;
            for (int j = 0; j < 7; ++j) ef.c[j] = (j < 6 ? A[6][j] : 0.0) - B4[j];
            GN_HIP(hipMemsetAsync(err, 0, sizeof(unsigned long long), st));
            hipLaunchKernelGGL(k_mf_err, dim3((unsigned)std::min<long>(lg, 1024)), dim3(256), 0, st, y, ynew, K, ef, hh, rtol, atol,
                               len, err);
            GN_LAUNCH_CHECK();
            unsigned long long eb = 0;
            GN_HIP(hipMemcpyAsync(&eb, err, sizeof(eb), hipMemcpyDeviceToHost, st));
            GN_HIP(hipStreamSynchronize(st));
            double e;
            memcpy(&e, &eb, sizeof(e));
            GN_CHECK_ARG(std::isfinite(e), "gnode_meanfield_f64: non-finite state at t=%g", t);
            ++steps;
            GN_CHECK_ARG(steps < 2000000, "gnode_meanfield_f64: step count exploded (h=%g at t=%g)", hh, t);
            if (e <= 1.0) {                                           // accept
                t = (hh == t_end - t) ? t_end : t + hh;
                std::swap(y, ynew);
                GN_HIP(hipMemcpyAsync(K, K + (size_t)6 * len, (size_t)len * sizeof(double), hipMemcpyDeviceToDevice, st));   // FSAL
            }
            const double fac = (e == 0.0) ? 5.0 : std::min(5.0, std::max(0.2, 0.9 * std::pow(e, -0.2)));
            h = hh * fac;
        }
        hipLaunchKernelGGL(k_mf_emit, dim3(ng), dim3(256), 0, st, y, n, to, n_out, outI, outS, outR);
        GN_LAUNCH_CHECK();
    }
    if (steps_host) *steps_host = steps;
    return 0;
}
