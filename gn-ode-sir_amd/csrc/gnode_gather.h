// Row gathers for the generic-H kernels (lane group of LPR = H/4 lanes per row, any LPR).
//
// Every lane reads the chunk's column ids itself (the lanes of a group hit the same addresses, so it is one
// request per group) and then has C neighbour rows in flight.  Fetching ids LPR at a time and passing them
// around by shuffle -- the natural scheme for 16-lane groups -- serialises small groups: at H = 8 (two lanes
// per row, the multi-graph launcher's hidden size, monitorer-ngraphs.py:20) a degree-28 row was 14 dependent
// id-load -> row-load round trips, 16 us per Euler step on a 22k-node batch.
// Accumulation order is ascending column position, the CPU scatter_add_ order of the reference
// (ode_nn_ngraph_sim.py:73).
#pragma once
#include <hip/hip_runtime.h>

template <int C>
__device__ __forceinline__ float4 gn_gather1(const int* __restrict__ col, int start, int end, const float* __restrict__ T,
                                             int H, int sub, bool active) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int c[C];
#pragma unroll
    for (int k = 0; k < C; ++k) c[k] = (start + k < end) ? col[start + k] : -1;
    for (int e0 = start; e0 < end; e0 += C) {
        int cn[C];                                   // ids of the NEXT chunk travel under this chunk's row loads
#pragma unroll
        for (int k = 0; k < C; ++k) cn[k] = (e0 + C + k < end) ? col[e0 + C + k] : -1;
        float4 v[C];
#pragma unroll
        for (int k = 0; k < C; ++k)
            v[k] = (active && c[k] >= 0) ? *reinterpret_cast<const float4*>(T + (size_t)c[k] * H + 4 * sub)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < C; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
#pragma unroll
        for (int k = 0; k < C; ++k) c[k] = cn[k];
    }
    return acc;
}

// two tables through one neighbour list
template <int C>
__device__ __forceinline__ void gn_gather2(const int* __restrict__ col, int start, int end, const float* __restrict__ T0,
                                           const float* __restrict__ T1, int H, int sub, bool active, float4& a0, float4& a1) {
    int c[C];
#pragma unroll
    for (int k = 0; k < C; ++k) c[k] = (start + k < end) ? col[start + k] : -1;
    for (int e0 = start; e0 < end; e0 += C) {
        int cn[C];
#pragma unroll
        for (int k = 0; k < C; ++k) cn[k] = (e0 + C + k < end) ? col[e0 + C + k] : -1;
        float4 u[C], v[C];
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const bool on = active && c[k] >= 0;
            const size_t o = (size_t)(on ? c[k] : 0) * H + 4 * sub;
            u[k] = on ? *reinterpret_cast<const float4*>(T0 + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[k] = on ? *reinterpret_cast<const float4*>(T1 + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < C; ++k) {
            a0.x += u[k].x; a0.y += u[k].y; a0.z += u[k].z; a0.w += u[k].w;
            a1.x += v[k].x; a1.y += v[k].y; a1.z += v[k].z; a1.w += v[k].w;
        }
#pragma unroll
        for (int k = 0; k < C; ++k) c[k] = cn[k];
    }
}
