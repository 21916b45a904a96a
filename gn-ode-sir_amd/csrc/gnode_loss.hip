// L1 loss of the trajectories against the Monte-Carlo labels, with its gradient, in one pass (gfx950).
//
// The reference assembles  pred = cat(S, I, R)[rows, T, 3]  out of three transposed views and calls
// torch.nn.L1Loss()(pred[:, 1:, :], y[:, 1:, :])  (ode_nn_ngraph_sim.py:230-234; the epoch loops weight the batch means by
// their element counts, :248-249, :265-266, :290-294).  As torch ops that is a cat, a transpose copy, a dtype conversion, a
// subtraction, an abs, a reduction and -- backward -- a sign, a scale and three strided slice copies: ~1.4 ms of
// element-wise launches per training step on the 75k-node graph x 4, against 33 ms for everything else.
//
// Here: one workgroup takes a block of RB consecutive rows.  The label block y[row0 .. row0+RB, :, :] is ONE contiguous
// range, staged in LDS with coalesced loads (labels are [rows, T, 3]: a row's values sit together, the model's outputs
// are [T, rows]: a grid point's values sit together -- LDS is where the two layouts meet).  Then every (grid point, row)
// pair reads its three predictions coalesced over rows, adds |pred - y| to a float64 partial and writes the three
// signs.  Partials are reduced in a fixed order (no float atomics): bitwise reproducible.
//   sum    = sum over rows, t >= t0, c of |pred_c[t, row] - y[row, t, c]|      (float64)
//   sgn    = sign(pred - y) as float, 0 for t < t0: d(sum)/d(pred)
// The difference is taken in the labels' dtype (the reference's `pred.to(y.dtype) - y` for float64 labels).
#include "gnode_common.h"
#include <algorithm>

template <typename YT>
__global__ __launch_bounds__(256) void k_l1_loss(const float* __restrict__ S, const float* __restrict__ I,
                                                 const float* __restrict__ R, const YT* __restrict__ y, long rows, int T, int t0,
                                                 int RB, double* __restrict__ partial, float* __restrict__ sgn, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    YT* yl = reinterpret_cast<YT*>(lds_raw);                       // [RB][T*3]
    __shared__ double red[256];
    const int T3 = T * 3;
    double acc = 0.0;
    const long nblocks = (rows + RB - 1) / RB;
    for (long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long row0 = blk * RB;
        const int nr = (int)min((long)RB, rows - row0);
        __syncthreads();                                           // previous block's labels consumed
        const YT* src = y + (size_t)row0 * T3;
        for (int e = threadIdx.x; e < nr * T3; e += 256) yl[e] = src[e];
        __syncthreads();
        // (t, r) pairs, r fastest: predictions and signs are coalesced over rows
        for (int e = threadIdx.x; e < nr * T; e += 256) {
            const int t = e / nr, r = e - t * nr;
            const size_t o = (size_t)t * rows + row0 + r;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            if (t >= t0) {
                const YT* yr = yl + (size_t)r * T3 + 3 * t;
                const YT d0 = (YT)S[o] - yr[0], d1 = (YT)I[o] - yr[1], d2 = (YT)R[o] - yr[2];
                acc += (double)fabs(d0) + (double)fabs(d1) + (double)fabs(d2);
                s0 = d0 > 0 ? scale : (d0 < 0 ? -scale : 0.f);
                s1 = d1 > 0 ? scale : (d1 < 0 ? -scale : 0.f);
                s2 = d2 > 0 ? scale : (d2 < 0 ? -scale : 0.f);
            }
            if (sgn) {
                const size_t plane = (size_t)T * rows;
                sgn[o] = s0; sgn[plane + o] = s1; sgn[2 * plane + o] = s2;
            }
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void k_l1_reduce(const double* __restrict__ partial, int n, double* __restrict__ out) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += partial[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

static const int kLossGrid = 1024;

extern "C" size_t gnode_l1_loss_workspace_bytes(void) { return gn_align(kLossGrid * sizeof(double)); }

extern "C" int gnode_l1_loss_f32(const float* S, const float* I, const float* R, const void* y, int32_t y_is_f64, int64_t rows,
                                 int32_t T, int32_t t0, double* loss_sum, float* sgn, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    return gnode_l1_loss_scaled_f32(S, I, R, y, y_is_f64, rows, T, t0, loss_sum, sgn, 1.0f, workspace, workspace_bytes, stream);
}

extern "C" int gnode_l1_loss_scaled_f32(const float* S, const float* I, const float* R, const void* y, int32_t y_is_f64, int64_t rows,
                                        int32_t T, int32_t t0, double* loss_sum, float* sgn, float sign_scale, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    GN_CHECK_ARG(S && I && R && y && loss_sum && workspace, "gnode_l1_loss_f32: null pointer");
    GN_CHECK_ARG(rows > 0 && T > 0 && t0 >= 0 && t0 <= T, "gnode_l1_loss_f32: bad shape rows=%lld T=%d t0=%d", (long long)rows, T, t0);
    if (workspace_bytes < gnode_l1_loss_workspace_bytes()) {
        gnode_set_error("gnode_l1_loss_f32: workspace %zu < %zu", workspace_bytes, gnode_l1_loss_workspace_bytes());
        return GNODE_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t el = y_is_f64 ? 8 : 4;
    // rows per block: as many as fit 48 KB of LDS (no attribute needed), at most 64, at least 1
    const size_t per_row = (size_t)T * 3 * el;
    GN_CHECK_ARG(per_row <= 48 * 1024, "gnode_l1_loss_f32: T=%d does not fit a label row into LDS", T);
    // ... and few enough that a mid-size batch still spreads over the chip (1 893 rows in blocks of 64 are 30 workgroups staging
    // 46 KB each: 13 us; in blocks of 4, 474 workgroups)
    const size_t spread = std::max<size_t>(4, ((size_t)rows + kLossGrid - 1) / kLossGrid);
    const int RB = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(64, spread), (48 * 1024) / per_row));
    const long nblocks = (rows + RB - 1) / RB;
    const int grid = (int)std::min<long>(kLossGrid, nblocks);
    double* partial = (double*)workspace;
    if (y_is_f64)
        hipLaunchKernelGGL(k_l1_loss<double>, dim3(grid), dim3(256), RB * per_row, st, S, I, R, (const double*)y, (long)rows, T, t0,
                           RB, partial, sgn, sign_scale);
    else
        hipLaunchKernelGGL(k_l1_loss<float>, dim3(grid), dim3(256), RB * per_row, st, S, I, R, (const float*)y, (long)rows, T, t0, RB,
                           partial, sgn, sign_scale);
    GN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_l1_reduce, dim3(1), dim3(256), 0, st, partial, grid, loss_sum);
    GN_LAUNCH_CHECK();
    return 0;
}
