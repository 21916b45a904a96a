// H = 128 node MLP on the fp32 matrix cores (gfx950):  Z = sigmoid(X W^T + b),  X [nrows, 128].
//
// The generic-H kernels do the node MLP as a lane-group mat-vec on the vector ALUs, fine up to H = 32 where the step
// stays bandwidth-bound, 12x off that bound at H = 128.  This kernel gives H = 128 the matrix cores for the
// contraction; the step then runs as two launches (this one, then gather + SIR update + read-out).
// 16-row tiles; W (128 x 128 fp32, 66 KB padded) stays in LDS for the whole launch, two 256-thread workgroups per
// CU; wave w owns output columns [32w, 32w + 32) as two independent 16x16 accumulators over K = 128
// (v_mfma_f32_16x16x4_f32, kappa = 32 (lane>>4) + 4m + c so that every fragment is one ds_read_b128).
#include "gnode_bwd.h"
#include <algorithm>

typedef float f32x4h __attribute__((ext_vector_type(4)));
static constexpr int TS128 = 132;      // LDS row stride (floats): 128 + 4 keeps 16-B alignment and staggers banks

__global__ __launch_bounds__(256) void k_mlp128(const float* __restrict__ X, const float* __restrict__ W,
                                                const float* __restrict__ bias, float* __restrict__ Z, long nrows) {
    extern __shared__ __attribute__((aligned(16))) float lds128[];
    float* Wl = lds128;                       // [128][TS128]
    float* T = Wl + 128 * TS128;              // [16][TS128]  X tile, then (after a barrier) the Z tile: 74 KB in all,
                                              // so TWO workgroups fit a CU and one's loads overlap the other's MFMAs
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
    for (int idx = threadIdx.x; idx < 128 * 32; idx += 256) {          // 128 rows x 32 float4
        const int r = idx >> 5, c4 = idx & 31;
        *reinterpret_cast<float4*>(Wl + r * TS128 + 4 * c4) = *reinterpret_cast<const float4*>(W + (size_t)r * 128 + 4 * c4);
    }
    const float b0 = bias[32 * w + i], b1 = bias[32 * w + 16 + i];
    const long ntiles = (nrows + 15) / 16;
    // this thread's two float4 of a tile: rows r0 = tid >> 5 and r0 + 8, column group c4 = tid & 31 (coalesced)
    const int r0 = threadIdx.x >> 5, c4 = threadIdx.x & 31;
    const float4 z4v = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 x0 = z4v, x1 = z4v;
    long t = blockIdx.x;
    if (t < ntiles) {
        const long ra = t * 16 + r0, rb = ra + 8;
        if (ra < nrows) x0 = *reinterpret_cast<const float4*>(X + (size_t)ra * 128 + 4 * c4);
        if (rb < nrows) x1 = *reinterpret_cast<const float4*>(X + (size_t)rb * 128 + 4 * c4);
    }
    for (; t < ntiles; t += gridDim.x) {
        __syncthreads();                                              // previous Z tile stored (and W staged)
        *reinterpret_cast<float4*>(T + r0 * TS128 + 4 * c4) = x0;
        *reinterpret_cast<float4*>(T + (r0 + 8) * TS128 + 4 * c4) = x1;
        const long tn = t + gridDim.x;                                // next tile's rows travel under the MFMAs
        x0 = z4v; x1 = z4v;
        if (tn < ntiles) {
            const long ra = tn * 16 + r0, rb = ra + 8;
            if (ra < nrows) x0 = *reinterpret_cast<const float4*>(X + (size_t)ra * 128 + 4 * c4);
            if (rb < nrows) x1 = *reinterpret_cast<const float4*>(X + (size_t)rb * 128 + 4 * c4);
        }
        __syncthreads();
        f32x4h acc0 = {b0, b0, b0, b0}, acc1 = {b1, b1, b1, b1};
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float4 a = *reinterpret_cast<const float4*>(T + i * TS128 + 32 * kq + 4 * m);
            const float4 w0 = *reinterpret_cast<const float4*>(Wl + (32 * w + i) * TS128 + 32 * kq + 4 * m);
            const float4 w1 = *reinterpret_cast<const float4*>(Wl + (32 * w + 16 + i) * TS128 + 32 * kq + 4 * m);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w1.w, acc1, 0, 0, 0);
        }
        __syncthreads();                                              // every wave is done reading the X tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                 // D: row = 4 kq + r, col = i
            T[(4 * kq + r) * TS128 + 32 * w + i] = __builtin_amdgcn_rcpf(1.0f + __expf(-acc0[r]));
            T[(4 * kq + r) * TS128 + 32 * w + 16 + i] = __builtin_amdgcn_rcpf(1.0f + __expf(-acc1[r]));
        }
        __syncthreads();
        const long ra = t * 16 + r0, rb = ra + 8;
        if (ra < nrows) *reinterpret_cast<float4*>(Z + (size_t)ra * 128 + 4 * c4) = *reinterpret_cast<const float4*>(T + r0 * TS128 + 4 * c4);
        if (rb < nrows) *reinterpret_cast<float4*>(Z + (size_t)rb * 128 + 4 * c4) = *reinterpret_cast<const float4*>(T + (r0 + 8) * TS128 + 4 * c4);
    }
}

static const size_t kMlp128Lds = (size_t)(128 + 16) * TS128 * sizeof(float);       // 76 032 B: two workgroups per CU
static const size_t kBwdMlp128Lds = (size_t)(128 + 64) * TS128 * sizeof(float);    // 101 376 B

int gn_h128_set_attributes();   // defined below both kernels

int gn_launch_mlp128(const gnode_graph_s* g, const float* X, const float* W, const float* b, float* Z, long nrows, hipStream_t st) {
    const size_t lds = kMlp128Lds;
    const long ntiles = (nrows + 15) / 16;
    hipLaunchKernelGGL(k_mlp128, dim3((unsigned)std::min<long>(ntiles, 2L * g->num_cu)), dim3(256), lds, st, X, W, b, Z, nrows);
    GN_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------- H = 128 backward: a += dt dpre W, gW, gb
// The generic k_bwd_mlp does both contractions on the vector ALUs (237 us per interval on a 7k-node graph, 75 % of an
// H = 128 training step).  Here, per 16-row tile and slab X in {S, I}, on the matrix cores with W resident in LDS:
//   gW[j][k] += sum_r dpre_X[r][j] y_X[r][k]   wave w owns the 16 rows j of block w, 8 accumulator tiles in registers
//                                              across ALL tiles of the workgroup;
//   g_Y = dpre_X W                              wave w owns output columns [16w, 16w+16); B fragments are read a row apart
//                                              from the same W[j][k] copy;  then a_X += dt g_Y, coalesced.
// 512 threads (8 waves), one workgroup per CU (101 KB of LDS).  Slot = blockIdx.x of the partial-gradient buffer.
__global__ __launch_bounds__(512) void k_bwd_mlp128(const float* __restrict__ dpre, const float* __restrict__ Ysol,
                                                    const float* __restrict__ W, float dt, float* __restrict__ a, long rows,
                                                    float* __restrict__ part_all) {
    extern __shared__ __attribute__((aligned(16))) float lds128[];
    float* Wl = lds128;                                   // [128][TS128]  W[j][k]
    float* D = Wl + 128 * TS128;                          // [2][16][TS128] dpre tile (S, I)
    float* Y = D + 2 * 16 * TS128;                        // [2][16][TS128] y tile, then g_Y
    const PartLayout L{128};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
    for (int idx = threadIdx.x; idx < 128 * 32; idx += 512) {
        const int r = idx >> 5, c4 = idx & 31;
        *reinterpret_cast<float4*>(Wl + r * TS128 + 4 * c4) = *reinterpret_cast<const float4*>(W + (size_t)r * 128 + 4 * c4);
    }
    const size_t slab = (size_t)rows * 128;
    const int tr = threadIdx.x >> 5, c4 = threadIdx.x & 31;   // this thread's row / column group of a tile
    f32x4h accW[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) accW[kt] = f32x4h{0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;
    const long ntiles = (rows + 15) / 16;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long row = t * 16 + tr;
        const bool ok = row < rows;
        const size_t off = (size_t)row * 128 + 4 * c4;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();                                  // previous tile consumed (and W staged)
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            *reinterpret_cast<float4*>(D + (X * 16 + tr) * TS128 + 4 * c4) = ok ? *reinterpret_cast<const float4*>(dpre + X * slab + off) : z;
            *reinterpret_cast<float4*>(Y + (X * 16 + tr) * TS128 + 4 * c4) = ok ? *reinterpret_cast<const float4*>(Ysol + X * slab + off) : z;
        }
        __syncthreads();
        // gW (rows j of block w) and gb
#pragma unroll
        for (int X = 0; X < 2; ++X) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int rr = 4 * s4 + kq;
                const float av = D[(X * 16 + rr) * TS128 + 16 * w + i];
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
                    accW[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Y[(X * 16 + rr) * TS128 + 16 * kt + i], accW[kt], 0, 0, 0);
            }
        }
        if (threadIdx.x < 128) {
            float sacc = 0.f;
            for (int rr = 0; rr < 32; ++rr) sacc += D[rr * TS128 + threadIdx.x];     // both slabs: rows 0..31 of D
            accb += sacc;
        }
        // g_Y = dpre W: columns [16w, 16w+16) of both slabs
        f32x4h g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float4 a0 = *reinterpret_cast<const float4*>(D + i * TS128 + 32 * kq + 4 * m);
            const float4 a1 = *reinterpret_cast<const float4*>(D + (16 + i) * TS128 + 32 * kq + 4 * m);
            const float* bp = Wl + (32 * kq + 4 * m) * TS128 + 16 * w + i;
            const float b0 = bp[0], b1 = bp[TS128], b2 = bp[2 * TS128], b3 = bp[3 * TS128];
            g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0, g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b0, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b1, g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b2, g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b2, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b3, g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b3, g1, 0, 0, 0);
        }
        __syncthreads();                                  // every wave is done with the y tile: g_Y goes over it
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Y[(4 * kq + r) * TS128 + 16 * w + i] = g0[r];
            Y[(16 + 4 * kq + r) * TS128 + 16 * w + i] = g1[r];
        }
        __syncthreads();
        if (ok) {
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                const float4 gy = *reinterpret_cast<const float4*>(Y + (X * 16 + tr) * TS128 + 4 * c4);
                float4 av = *reinterpret_cast<const float4*>(a + X * slab + off);
                av.x += dt * gy.x; av.y += dt * gy.y; av.z += dt * gy.z; av.w += dt * gy.w;
                *reinterpret_cast<float4*>(a + X * slab + off) = av;
            }
        }
    }
    float* part = part_all + (size_t)blockIdx.x * L.total();
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            part[L.oW() + (16 * w + 4 * kq + reg) * 128 + 16 * kt + i] += dt * accW[kt][reg];
    if (threadIdx.x < 128) part[L.ob() + threadIdx.x] += dt * accb;
}

int gn_h128_set_attributes() {      // once per device, from gnode_graph_create
    GN_HIP(hipFuncSetAttribute((const void*)k_mlp128, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMlp128Lds));
    GN_HIP(hipFuncSetAttribute((const void*)k_bwd_mlp128, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdMlp128Lds));
    return 0;
}

int gn_launch_bwd_mlp128(const gnode_graph_s* g, const float* dpre, const float* Ysol, const float* W, float dt, float* a, long rows,
                         float* part, int* slots_used, hipStream_t st) {
    const size_t lds = kBwdMlp128Lds;
    const long ntiles = (rows + 15) / 16;
    const int grid = (int)std::min<long>(std::min<long>(ntiles, g->num_cu), BWD_NWG);     // one slot of the partial buffer per workgroup
    *slots_used = std::max(*slots_used, grid);
    hipLaunchKernelGGL(k_bwd_mlp128, dim3(grid), dim3(512), lds, st, dpre, Ysol, W, dt, a, rows, part);
    GN_LAUNCH_CHECK();
    return 0;
}

