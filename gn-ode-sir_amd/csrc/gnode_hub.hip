// Hub rows: degree-skew handling for the CSR pull-gather (gfx950).
//
// A row gather is a chain of dependent memory batches (4 neighbour rows in flight per 16-lane
// group, ~1-2 us each under load).  That is fine at degree ~13 (Erdos-Renyi) and terrible on the
// graphs the reference actually ships: wiki-vote has a row of degree 1 065, epinions ~3 000, and a
// single lane group walking such a row holds its whole workgroup for hundreds of microseconds
// (measured on a Chung-Lu graph, 75k nodes / 1M edges, max degree 12 777: 2 244 us per Euler step
// instead of 213).  So rows longer than HUB_T (96 edges: below that, the two extra launches per step cost
// more than the ~24 dependent batches they save) are taken out of the step kernels:
//
//   graph build  rows with degree > HUB_T become "hubs"; their neighbour lists are cut into
//                segments of <= 32 edges (uniform work items).
//   k_hub_seg    one lane group per (sample, segment): partial[b][s][:] = sum of the segment's
//                neighbour rows, ascending column order inside the segment.
//   consumers    the step kernels (H = 64 and generic H), the backward over kept activations and the H <= 32 interval
//                kernel add a hub's partials up THEMSELVES, in segment order, several partial rows in flight
//                (gn_hub_segments / gn_hub_segments2): a second launch per step cost 17-35 us on the 75k Chung-Lu
//                graph and as much as the whole step at mid size.  They gather the other rows as usual.
//   k_hub_reduce for the remaining consumers (RHS API, RK4 stages, the recomputing and five-launch backward forms):
//                one lane group per (sample, hub): AIhub[b][h][:] = partials of the hub summed in the same
//                segment order (gn_hub_gather); those kernels read AIhub for hub rows.
//
// Deterministic (no atomics): a hub's sum is blocked by segment but always in the same order.
// Up to two tables are reduced through the same index lists in one pass (the backward needs
// A Z_I and A q).  Segment partials and hub sums live in the CALLER's workspace
// (gn_hub_scratch_bytes is part of gnode_{rhs,forward,backward}_workspace_bytes): the launch path
// allocates nothing, synchronises nothing and keeps nothing in the handle, so it can be captured into a
// hipGraph on first use and one handle serves several streams.
#include "gnode_common.h"
#include "gnode_gather.h"
#include <algorithm>
#include <vector>

// HUB_SEG, GN_HUB_T: gnode_common.h (the persistent kernels cut the same segments)

__device__ __forceinline__ float4 hld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void hst4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// partial[t][b][s][H]: one lane group (LPR lanes, 4 features each) per (sample b, segment s)
template <int LPR>
__global__ __launch_bounds__(256) void k_hub_seg(const int* __restrict__ seg_lo, const int* __restrict__ seg_hi,
                                                 const int* __restrict__ col, int n, int n_seg, int H,
                                                 const float* __restrict__ T0, const float* __restrict__ T1,
                                                 float* __restrict__ P0, float* __restrict__ P1, long b0) {
    const int sub = threadIdx.x % LPR;
    const int s = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (s >= n_seg) return;
    const bool active = 4 * sub < H;
    const long b = b0 + blockIdx.y;                 // (grid.y is limited to 65 535: larger batches come as several launches)
    const float* t0 = T0 + (size_t)b * n * H;
    const float* t1 = T1 ? T1 + (size_t)b * n * H : nullptr;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    const int lo = seg_lo[s], hi = seg_hi[s];
    if (LPR <= 4) {
        // small lane groups (H = 8 / 16: the multi-graph launcher's hidden 8): fetching ids LPR at a time makes a 32-edge segment
        // 16 dependent round trips of 2 loads; every lane reads the ids itself instead and keeps 16 (8 per table) rows in flight
        // (gnode_gather.h) -- the same ascending sums.  Chung-Lu 75k x 4 samples at H = 8: 76 -> 5x us per Euler step.
        if (t1) gn_gather2<8>(col, lo, hi, t0, t1, H, sub, active, a0, a1);
        else a0 = gn_gather1<16>(col, lo, hi, t0, H, sub, active);
        if (!active) return;
        const size_t o = ((size_t)b * n_seg + s) * H + 4 * sub;
        hst4(P0 + o, a0);
        if (t1) hst4(P1 + o, a1);
        return;
    }
    for (int e0 = lo; e0 < hi; e0 += LPR) {
        const int cnt = min(LPR, hi - e0);
        const int mine = (sub < cnt) ? col[e0 + sub] : 0;
        // 16 neighbour rows (per table) in flight per lane group, summed in ascending column order
        for (int j = 0; j < cnt; j += 16) {
            int c[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) c[q] = __shfl(mine, min(j + q, LPR - 1), LPR);
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 u[16], v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                u[q] = z; v[q] = z;
                if (active && j + q < cnt) {
                    u[q] = hld4(t0 + (size_t)c[q] * H + 4 * sub);
                    if (t1) v[q] = hld4(t1 + (size_t)c[q] * H + 4 * sub);
                }
            }
#define HUB_ACC(A, V) A.x += V.x; A.y += V.y; A.z += V.z; A.w += V.w;
#pragma unroll
            for (int q = 0; q < 16; ++q) { HUB_ACC(a0, u[q]) }
#pragma unroll
            for (int q = 0; q < 16; ++q) { HUB_ACC(a1, v[q]) }
        }
    }
    if (!active) return;
    const size_t o = ((size_t)b * n_seg + s) * H + 4 * sub;
    hst4(P0 + o, a0);
    if (t1) hst4(P1 + o, a1);
}

// AIhub[t][b][h][H] = sum over the hub's segments, in segment order
template <int LPR>
__global__ __launch_bounds__(256) void k_hub_reduce(const int* __restrict__ hub_seg_ptr, int n_hub, int n_seg, int H,
                                                    const float* __restrict__ P0, const float* __restrict__ P1,
                                                    float* __restrict__ A0, float* __restrict__ A1) {
    const int sub = threadIdx.x % LPR;
    const int h = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (h >= n_hub || 4 * sub >= H) return;
    const long b = blockIdx.y;
    const int s0 = hub_seg_ptr[h], s1 = hub_seg_ptr[h + 1];
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    const float* p0 = P0 + ((size_t)b * n_seg) * H + 4 * sub;
    const float* p1 = P1 ? P1 + ((size_t)b * n_seg) * H + 4 * sub : nullptr;
    int s = s0;
    // the largest hub sets this launch's duration (one lane group walks its segments): 32 partial rows in flight,
    // summed in the same segment order as the 4-wide loop below (bit-identical results)
    for (; s + 32 <= s1; s += 32) {
        float4 u[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) u[q] = hld4(p0 + (size_t)(s + q) * H);
#pragma unroll
        for (int q = 0; q < 32; ++q) { HUB_ACC(a0, u[q]) }
        if (p1) {
            float4 v[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) v[q] = hld4(p1 + (size_t)(s + q) * H);
#pragma unroll
            for (int q = 0; q < 32; ++q) { HUB_ACC(a1, v[q]) }
        }
    }
    for (; s + 16 <= s1; s += 16) {
        float4 u[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) u[q] = hld4(p0 + (size_t)(s + q) * H);
#pragma unroll
        for (int q = 0; q < 16; ++q) { HUB_ACC(a0, u[q]) }
        if (p1) {
            float4 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = hld4(p1 + (size_t)(s + q) * H);
#pragma unroll
            for (int q = 0; q < 16; ++q) { HUB_ACC(a1, v[q]) }
        }
    }
    for (; s + 4 <= s1; s += 4) {
        const float4 u0 = hld4(p0 + (size_t)s * H), u1 = hld4(p0 + (size_t)(s + 1) * H);
        const float4 u2 = hld4(p0 + (size_t)(s + 2) * H), u3 = hld4(p0 + (size_t)(s + 3) * H);
        HUB_ACC(a0, u0) HUB_ACC(a0, u1) HUB_ACC(a0, u2) HUB_ACC(a0, u3)
        if (p1) {
            const float4 v0 = hld4(p1 + (size_t)s * H), v1 = hld4(p1 + (size_t)(s + 1) * H);
            const float4 v2 = hld4(p1 + (size_t)(s + 2) * H), v3 = hld4(p1 + (size_t)(s + 3) * H);
            HUB_ACC(a1, v0) HUB_ACC(a1, v1) HUB_ACC(a1, v2) HUB_ACC(a1, v3)
        }
    }
    for (; s < s1; ++s) {
        const float4 u = hld4(p0 + (size_t)s * H);
        HUB_ACC(a0, u)
        if (p1) { const float4 v = hld4(p1 + (size_t)s * H); HUB_ACC(a1, v) }
    }
#undef HUB_ACC
    const size_t o = ((size_t)b * n_hub + h) * H + 4 * sub;
    hst4(A0 + o, a0);
    if (p1) hst4(A1 + o, a1);
}

// --------------------------------------------------------------------------- host: build + launch
int gn_hub_build(gnode_graph_s* g, const int32_t* rowptr_host) {
    const int T = GN_HUB_T;
    std::vector<int32_t> hubidx((size_t)g->n, -1), seg_lo, seg_hi, hub_seg_ptr(1, 0);
    int n_hub = 0;
    for (int32_t r = 0; r < g->n; ++r) {
        const int32_t lo = rowptr_host[r], hi = rowptr_host[r + 1];
        if (hi - lo <= T) continue;
        hubidx[r] = n_hub++;
        for (int32_t e = lo; e < hi; e += HUB_SEG) {
            seg_lo.push_back(e);
            seg_hi.push_back(std::min(hi, e + HUB_SEG));
        }
        hub_seg_ptr.push_back((int32_t)seg_lo.size());
    }
    g->n_hub = n_hub;
    g->n_seg = (int32_t)seg_lo.size();
    g->hubidx = g->seg_lo = g->seg_hi = g->hub_seg_ptr = nullptr;
    if (n_hub == 0) return 0;
    auto up = [](int32_t** dst, const std::vector<int32_t>& v) -> hipError_t {
        hipError_t e = hipMalloc(dst, sizeof(int32_t) * v.size());
        if (e != hipSuccess) return e;
        return hipMemcpy(*dst, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice);
    };
    GN_HIP(up(&g->hubidx, hubidx));
    GN_HIP(up(&g->seg_lo, seg_lo));
    GN_HIP(up(&g->seg_hi, seg_hi));
    GN_HIP(up(&g->hub_seg_ptr, hub_seg_ptr));
    return 0;
}

void gn_hub_free(gnode_graph_s* g) {
    if (g->hubidx) (void)hipFree(g->hubidx);
    if (g->seg_lo) (void)hipFree(g->seg_lo);
    if (g->seg_hi) (void)hipFree(g->seg_hi);
    if (g->hub_seg_ptr) (void)hipFree(g->hub_seg_ptr);
}

static int hub_lpr(int H) {
    int need = H / 4, l = 1;
    while (l < need) l <<= 1;
    return l;
}

#define HUB_DISPATCH(lpr, ...)                                   \
    switch (lpr) {                                               \
        case 1: { constexpr int LPR = 1; __VA_ARGS__; } break;   \
        case 2: { constexpr int LPR = 2; __VA_ARGS__; } break;   \
        case 4: { constexpr int LPR = 4; __VA_ARGS__; } break;   \
        case 8: { constexpr int LPR = 8; __VA_ARGS__; } break;   \
        case 16: { constexpr int LPR = 16; __VA_ARGS__; } break; \
        case 32: { constexpr int LPR = 32; __VA_ARGS__; } break; \
        default: { constexpr int LPR = 64; __VA_ARGS__; } break; \
    }

size_t gn_hub_scratch_bytes(const gnode_graph_s* g, long B, int H, int ntables) {
    if (g->n_hub == 0) return 0;
    const size_t part_f = (size_t)B * g->n_seg * H, hub_f = (size_t)B * g->n_hub * H;
    return (size_t)ntables * (gn_align(sizeof(float) * part_f) + gn_align(sizeof(float) * hub_f));
}

// Segment partials only (one table): *P0 points at [B][n_seg][H] inside `scratch`; the consumer adds a hub's segments up in
// order itself (the H = 64 step kernel), which saves the reduction launch.  Same scratch layout as gn_hub_gather.
int gn_hub_segments(const gnode_graph_s* g, long B, int H, const float* T0, void* scratch, const float** P0out, hipStream_t st) {
    return gn_hub_segments2(g, B, H, T0, nullptr, scratch, P0out, nullptr, st);
}

// ... of one or two tables gathered through the same neighbour lists (T1 / P1out may be null)
int gn_hub_segments2(const gnode_graph_s* g, long B, int H, const float* T0, const float* T1, void* scratch, const float** P0out,
                     const float** P1out, hipStream_t st) {
    *P0out = nullptr;
    if (P1out) *P1out = nullptr;
    if (g->n_hub == 0) return 0;
    GN_CHECK_ARG(scratch, "hub rows present but no hub scratch was carved from the workspace");
    const size_t part_b = gn_align(sizeof(float) * (size_t)B * g->n_seg * H);
    float* P0 = (float*)scratch;
    float* P1 = T1 ? (float*)((char*)scratch + part_b) : nullptr;       // (inside the two-table scratch of gn_hub_scratch_bytes)
    const int lpr = hub_lpr(H), gpw = 256 / lpr;
    for (long b0 = 0; b0 < B; b0 += 65535) {
        const unsigned nb = (unsigned)std::min<long>(65535, B - b0);
        HUB_DISPATCH(lpr, hipLaunchKernelGGL(k_hub_seg<LPR>, dim3((unsigned)((g->n_seg + gpw - 1) / gpw), nb), dim3(256), 0, st,
                                             g->seg_lo, g->seg_hi, g->col, g->n, g->n_seg, H, T0, T1, P0, P1, b0));
    }
    GN_LAUNCH_CHECK();
    *P0out = P0;
    if (P1out) *P1out = P1;
    return 0;
}

// Hub sums of one or two tables [B*n][H] (T1 may be null).  On return *A0 / *A1 point at [B][n_hub][H]
// buffers inside `scratch` (>= gn_hub_scratch_bytes(g, B, H, T1 ? 2 : 1) bytes of the caller's workspace).
int gn_hub_gather(const gnode_graph_s* g, long B, int H, const float* T0, const float* T1, void* scratch, const float** A0,
                  const float** A1, hipStream_t st) {
    *A0 = nullptr;
    if (A1) *A1 = nullptr;
    if (g->n_hub == 0) return 0;
    GN_CHECK_ARG(scratch, "hub rows present but no hub scratch was carved from the workspace");
    const int nt = T1 ? 2 : 1;
    const size_t part_b = gn_align(sizeof(float) * (size_t)B * g->n_seg * H), hub_b = gn_align(sizeof(float) * (size_t)B * g->n_hub * H);
    char* base = (char*)scratch;
    float* P0 = (float*)base;
    float* a0 = (float*)(base + part_b);
    float* P1 = (float*)(base + part_b + hub_b);
    float* a1 = (float*)(base + 2 * part_b + hub_b);
    const int lpr = hub_lpr(H), gpw = 256 / lpr;
    GN_CHECK_ARG(B <= 65535, "hub sums: %ld samples per launch exceed the grid's y extent (split the batch)", B);
    HUB_DISPATCH(lpr, hipLaunchKernelGGL(k_hub_seg<LPR>, dim3((unsigned)((g->n_seg + gpw - 1) / gpw), (unsigned)B), dim3(256), 0, st,
                                         g->seg_lo, g->seg_hi, g->col, g->n, g->n_seg, H, T0, T1, P0, P1, 0L));
    GN_LAUNCH_CHECK();
    HUB_DISPATCH(lpr, hipLaunchKernelGGL(k_hub_reduce<LPR>, dim3((unsigned)((g->n_hub + gpw - 1) / gpw), (unsigned)B), dim3(256), 0,
                                         st, g->hub_seg_ptr, g->n_hub, g->n_seg, H, P0, nt == 2 ? P1 : nullptr, a0, a1));
    GN_LAUNCH_CHECK();
    *A0 = a0;
    if (A1) *A1 = nt == 2 ? a1 : nullptr;
    return 0;
}
