// H = 128 node MLP on the fp32 matrix cores (gfx950):  Z = sigmoid(X W^T + b),  X [nrows, 128].
//
// The generic-H kernels do the node MLP as a lane-group mat-vec on the vector ALUs, fine up to H = 32 where the step
// stays bandwidth-bound, 12x off that bound at H = 128.  This kernel gives H = 128 the matrix cores for the
// contraction; the step then runs as two launches (this one, then gather + SIR update + read-out).
// 16-row tiles; W (128 x 128 fp32, 66 KB padded) stays in LDS for the whole launch, two 256-thread workgroups per
// CU; wave w owns output columns [32w, 32w + 32) as two independent 16x16 accumulators over K = 128
// (v_mfma_f32_16x16x4_f32, kappa = 32 (lane>>4) + 4m + c so that every fragment is one ds_read_b128).
#include "gnode_common.h"
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

int gn_launch_mlp128(const float* X, const float* W, const float* b, float* Z, long nrows, hipStream_t st) {
    const size_t lds = (size_t)(128 + 16) * TS128 * sizeof(float);     // 76 032 B: two workgroups per CU
    static bool attr = false;                                          // once, never inside a stream capture
    if (!attr) {
        GN_HIP(hipFuncSetAttribute((const void*)k_mlp128, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const long ntiles = (nrows + 15) / 16;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipLaunchKernelGGL(k_mlp128, dim3((unsigned)std::min<long>(ntiles, 2L * cus)), dim3(256), lds, st, X, W, b, Z, nrows);
    GN_LAUNCH_CHECK();
    return 0;
}
