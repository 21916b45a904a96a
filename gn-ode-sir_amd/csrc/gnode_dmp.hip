// Dynamic message passing baseline for SIR on MI355X (gfx950) -- SURVEY 8f rank 4.
//
// Restates DMP_SIR of the reference (dmp.py:74-170): messages live on the directed edges e = (src -> tar) of the
// weighted adjacency in COO row-major order (dmp.py:67-72 = the CSR positions of the graph handle);
//   theta_e   <- theta_e - w_e phi_e
//   P_k        = prod_{e: tar(e) = k} theta_e                 (torch_scatter scatter(reduce='mul'), :93-100)
//   Ps_e       = Ps0[src] * P[src] / theta_{rev(e)}           (the cavity product, :96-99)
//   phi_e     <- (1 - w_e)(1 - gamma_src) phi_e - (Ps_e - Ps_e^{prev})
//   marginals  Ps_k = Ps0_k P_k,  Pr_k += gamma_k Pi_k,  Pi_k = 1 - Ps_k - Pr_k       (:124-129, :141-145)
// Undirected graphs only (every shipped graph is `.to_undirected()`, ode_nn.py:402): the edges INTO k are then the
// reverses of row k's edges, so one CSR serves both directions and P_k multiplies theta_{rev(e)} over row k in
// ascending order -- the order the reference's CPU scatter uses.  fp32 like the reference's FloatTensors; two
// launches per time step (node pass, edge pass), theta ping-pongs because the edge pass reads its neighbours'.
#include "gnode_common.h"
#include <algorithm>

__global__ __launch_bounds__(256) void k_dmp_setup(const int* __restrict__ rowptr, const int* __restrict__ col, int n,
                                                  int* __restrict__ src, int* __restrict__ rev, int* __restrict__ bad) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    for (int e = rowptr[u]; e < rowptr[u + 1]; ++e) {
        const int v = col[e];
        src[e] = u;
        int lo = rowptr[v], hi = rowptr[v + 1] - 1, found = -1;       // position of u in row v (sorted columns)
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1, c = col[mid];
            if (c == u) { found = mid; break; }
            if (c < u) lo = mid + 1; else hi = mid - 1;
        }
        rev[e] = found;
        if (found < 0) atomicExch(bad, 1);
    }
}

// t = 1 (dmp.py:104-121): theta_1 = 1 - w phi_0 + 1e-10, Ps_e^{prev} = Ps0[src]
__global__ __launch_bounds__(256) void k_dmp_init(const int* __restrict__ src, const float* __restrict__ w,
                                                 const float* __restrict__ seed, long nnz, float* __restrict__ theta,
                                                 float* __restrict__ phi, float* __restrict__ ps) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const float ps0 = 1.0f - seed[src[e]];
    const float ph = 1.0f - ps0;
    theta[e] = (1.0f - w[e] * ph) + 1e-10f;
    phi[e] = ph;
    ps[e] = ps0;
}

// node pass: P_k and the marginals of time step t
__global__ __launch_bounds__(256) void k_dmp_node(const int* __restrict__ rowptr, const int* __restrict__ rev,
                                                 const float* __restrict__ theta, const float* __restrict__ seed,
                                                 const float* __restrict__ gamma, int n, int first, float* __restrict__ P,
                                                 float* __restrict__ Pr, float* __restrict__ Pi, float* __restrict__ out_t) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    float p = 1.0f;
    for (int e = rowptr[k]; e < rowptr[k + 1]; ++e) p = p * theta[rev[e]];
    P[k] = p;
    const float ps0 = 1.0f - seed[k];
    const float ps = ps0 * p;
    const float pi_prev = first ? seed[k] : Pi[k];
    const float pr = (first ? 0.0f : Pr[k]) + gamma[k] * pi_prev;
    const float pi = 1.0f - ps - pr;
    Pr[k] = pr; Pi[k] = pi;
    out_t[(size_t)k * 3 + 0] = ps; out_t[(size_t)k * 3 + 1] = pi; out_t[(size_t)k * 3 + 2] = pr;
}

// edge pass: Ps_e, phi_e of this step, theta of the NEXT step into the other buffer
__global__ __launch_bounds__(256) void k_dmp_edge(const int* __restrict__ src, const int* __restrict__ rev,
                                                 const float* __restrict__ w, const float* __restrict__ gamma,
                                                 const float* __restrict__ seed, const float* __restrict__ P,
                                                 const float* __restrict__ theta, long nnz, float* __restrict__ phi,
                                                 float* __restrict__ ps, float* __restrict__ theta_next) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const int u = src[e];
    const float mul = P[u] / theta[rev[e]];
    const float ps_new = (1.0f - seed[u]) * mul;
    const float ph = (1.0f - w[e]) * (1.0f - gamma[u]) * phi[e] - (ps_new - ps[e]);
    phi[e] = ph;
    ps[e] = ps_new;
    theta_next[e] = theta[e] - w[e] * ph;
}

__global__ __launch_bounds__(256) void k_dmp_out0(const float* __restrict__ seed, int n, float* __restrict__ out0) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    out0[(size_t)k * 3 + 0] = 1.0f - seed[k]; out0[(size_t)k * 3 + 1] = seed[k]; out0[(size_t)k * 3 + 2] = 0.0f;
}

extern "C" size_t gnode_dmp_workspace_bytes(gnode_graph_t g) {
    if (!g) return 0;
    const size_t eb = gn_align((size_t)std::max<int64_t>(g->nnz, 1) * 4), nb = gn_align((size_t)g->n * 4);
    return 6 * eb + 4 * nb + 256;          // src, rev, theta x2, phi, ps | seed, P, Pr, Pi | status
}

extern "C" int gnode_dmp_f32(gnode_graph_t g, const float* weights, const float* gamma, const int32_t* seeds_host,
                             int32_t n_seeds, int32_t maxTime, float* out, void* workspace, size_t workspace_bytes,
                             void* stream) {
    GN_CHECK_ARG(g && weights && gamma && out && workspace && (seeds_host || n_seeds == 0), "gnode_dmp_f32: null pointer");
    GN_CHECK_ARG(maxTime >= 2, "gnode_dmp_f32: maxTime must be >= 2 (got %d)", maxTime);
    for (int i = 0; i < n_seeds; ++i)
        GN_CHECK_ARG(seeds_host[i] >= 0 && seeds_host[i] < g->n, "gnode_dmp_f32: seed %d out of range", seeds_host[i]);
    if (workspace_bytes < gnode_dmp_workspace_bytes(g)) {
        gnode_set_error("gnode_dmp_f32: workspace %zu < %zu", workspace_bytes, gnode_dmp_workspace_bytes(g));
        return GNODE_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const long nnz = g->nnz;
    const int n = g->n;
    const size_t eb = gn_align((size_t)std::max<int64_t>(nnz, 1) * 4), nb = gn_align((size_t)n * 4);
    char* ws = (char*)workspace;
    int* src = (int*)ws; int* rev = (int*)(ws + eb);
    float* theta[2] = {(float*)(ws + 2 * eb), (float*)(ws + 3 * eb)};
    float* phi = (float*)(ws + 4 * eb); float* ps = (float*)(ws + 5 * eb);
    float* seed = (float*)(ws + 6 * eb); float* P = (float*)(ws + 6 * eb + nb);
    float* Pr = (float*)(ws + 6 * eb + 2 * nb); float* Pi = (float*)(ws + 6 * eb + 3 * nb);
    int* bad = (int*)(ws + 6 * eb + 4 * nb);
    GN_HIP(hipMemsetAsync(seed, 0, (size_t)n * 4, st));
    GN_HIP(hipMemsetAsync(bad, 0, 4, st));
    const float one = 1.0f;
    for (int i = 0; i < n_seeds; ++i) GN_HIP(hipMemcpyAsync(seed + seeds_host[i], &one, 4, hipMemcpyHostToDevice, st));
    const unsigned ng = (unsigned)((n + 255) / 256), eg = (unsigned)std::max<long>(1, (nnz + 255) / 256);
    hipLaunchKernelGGL(k_dmp_setup, dim3(ng), dim3(256), 0, st, g->rowptr, g->col, n, src, rev, bad);
    GN_LAUNCH_CHECK();
    int bad_h = 0;
    GN_HIP(hipMemcpyAsync(&bad_h, bad, 4, hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));          // also: `one` and seeds_host are done with
    GN_CHECK_ARG(!bad_h, "gnode_dmp_f32: the sparsity pattern is not symmetric (DMP here serves undirected graphs)");
    const size_t plane = (size_t)n * 3;
    hipLaunchKernelGGL(k_dmp_out0, dim3(ng), dim3(256), 0, st, seed, n, out);
    hipLaunchKernelGGL(k_dmp_init, dim3(eg), dim3(256), 0, st, src, weights, seed, nnz, theta[0], phi, ps);
    GN_LAUNCH_CHECK();
    for (int t = 1; t < maxTime; ++t) {
        const int cur = (t - 1) & 1;
        hipLaunchKernelGGL(k_dmp_node, dim3(ng), dim3(256), 0, st, g->rowptr, rev, theta[cur], seed, gamma, n, t == 1 ? 1 : 0, P, Pr,
                           Pi, out + (size_t)t * plane);
        hipLaunchKernelGGL(k_dmp_edge, dim3(eg), dim3(256), 0, st, src, rev, weights, gamma, seed, P, theta[cur], nnz, phi, ps,
                           theta[cur ^ 1]);
        GN_LAUNCH_CHECK();
    }
    return 0;
}
