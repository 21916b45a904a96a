// VJP of the read-out head for rows held by LPR-lane groups, 4 features per lane (shared by the backward kernels).
#pragma once
#include "gnode_mfma64.h"

__device__ __forceinline__ float dot4t(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }

// per-lane-group accumulators of the read-out head's parameter gradients
struct HeadAcc {
    float4 dw3[4];
    float db3[4], dw2[4], db2;
};

// sum over the LPR lanes that hold one row: a DPP row reduction for 16-lane groups, xor-shuffles otherwise
template <int LPR>
__device__ __forceinline__ float head_rowsum(float v) {
    if (LPR == 16) return row_sum16(v);
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, LPR);
    return v;
}

// VJP of softmax(linearS2(relu(linear3(y_X)))) at one row (reference ode_nn_ngraph_sim.py:172-187): a_X += dL/dy_X
template <int LPR = 16>
__device__ __forceinline__ void head_vjp64(const float4 (&y)[3], const float (&gout)[3], const float4 (&w3v)[4],
                                           const float* __restrict__ b3, const float* __restrict__ w2,
                                           const float* __restrict__ b2, float4& aS, float4& aI, float4& aR, HeadAcc& acc) {
    float p3[3][4], q[3];
#pragma unroll
    for (int X = 0; X < 3; ++X) {
        q[X] = b2[0];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            p3[X][k] = head_rowsum<LPR>(dot4t(w3v[k], y[X])) + b3[k];
            q[X] = fmaf(w2[k], fmaxf(p3[X][k], 0.f), q[X]);
        }
    }
    const float m = fmaxf(q[0], fmaxf(q[1], q[2]));
    const float e0 = __expf(q[0] - m), e1 = __expf(q[1] - m), e2 = __expf(q[2] - m);
    const float inv = 1.0f / (e0 + e1 + e2);
    const float pr[3] = {e0 * inv, e1 * inv, e2 * inv};
    const float gp = gout[0] * pr[0] + gout[1] * pr[1] + gout[2] * pr[2];
    float4* av[3] = {&aS, &aI, &aR};
#pragma unroll
    for (int X = 0; X < 3; ++X) {
        const float dq = pr[X] * (gout[X] - gp);
        float4 dy = zero4();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dp3 = p3[X][k] > 0.f ? dq * w2[k] : 0.f;
            dy.x = fmaf(dp3, w3v[k].x, dy.x); dy.y = fmaf(dp3, w3v[k].y, dy.y);
            dy.z = fmaf(dp3, w3v[k].z, dy.z); dy.w = fmaf(dp3, w3v[k].w, dy.w);
            acc.dw3[k].x = fmaf(dp3, y[X].x, acc.dw3[k].x); acc.dw3[k].y = fmaf(dp3, y[X].y, acc.dw3[k].y);
            acc.dw3[k].z = fmaf(dp3, y[X].z, acc.dw3[k].z); acc.dw3[k].w = fmaf(dp3, y[X].w, acc.dw3[k].w);
            acc.db3[k] += dp3;
            acc.dw2[k] = fmaf(dq, fmaxf(p3[X][k], 0.f), acc.dw2[k]);
        }
        acc.db2 += dq;
        av[X]->x += dy.x; av[X]->y += dy.y; av[X]->z += dy.z; av[X]->w += dy.w;
    }
}

