// Lane-group helpers of the generic-H kernels (a row of H features lives 4 per lane in a group of LPR = H/4 lanes):
// sigmoid, the read-out head of one row, the node MLP as a lane-group mat-vec.  Shared by gnode_ode.hip (one launch per
// Euler step) and gnode_persg.hip (the whole integration in one launch).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float gn_sigmoid(float x) {
    // 1 / (1 + exp(-x)) with the hardware exp2/rcp (v_exp_f32, v_rcp_f32: 1 ulp each)
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, LPR);
    return v;
}

// Read-out head for one row held by an LPR-lane group (each lane 4 features):
// Linear(4,1)(relu(Linear(H,4)(y)))  ode_nn_ngraph_sim.py:172-182, for S, I, R, then
// the 3-way softmax :184-187.  Every lane of the group returns the same values.
template <int LPR>
__device__ __forceinline__ void readout_row(float4 yS, float4 yI, float4 yR, bool active, int sub, int H,
                                            const float* __restrict__ w3, const float* __restrict__ b3,
                                            const float* __restrict__ w2, const float* __restrict__ b2,
                                            float& pS, float& pI, float& pR) {
    float qS = b2[0], qI = b2[0], qR = b2[0];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float4 w = active ? ld4(w3 + (size_t)k * H + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
        float s = fmaf(w.x, yS.x, fmaf(w.y, yS.y, fmaf(w.z, yS.z, w.w * yS.w)));
        float i = fmaf(w.x, yI.x, fmaf(w.y, yI.y, fmaf(w.z, yI.z, w.w * yI.w)));
        float r = fmaf(w.x, yR.x, fmaf(w.y, yR.y, fmaf(w.z, yR.z, w.w * yR.w)));
        s = group_sum<LPR>(s) + b3[k];
        i = group_sum<LPR>(i) + b3[k];
        r = group_sum<LPR>(r) + b3[k];
        qS = fmaf(w2[k], fmaxf(s, 0.f), qS);
        qI = fmaf(w2[k], fmaxf(i, 0.f), qI);
        qR = fmaf(w2[k], fmaxf(r, 0.f), qR);
    }
    float m = fmaxf(qS, fmaxf(qI, qR));
    float eS = __expf(qS - m), eI = __expf(qI - m), eR = __expf(qR - m);
    float inv = __builtin_amdgcn_rcpf(eS + eI + eR);
    pS = eS * inv; pI = eI * inv; pR = eR * inv;
}

// The same read-out with the head's weights already in registers (w3[k]: the lane's 4 columns of row k): same operations in
// the same order as readout_row, so the same bits.
template <int LPR>
__device__ __forceinline__ void readout_row_regs(float4 yS, float4 yI, float4 yR, const float4 (&w3)[4], const float (&b3)[4],
                                                 const float (&w2)[4], float b2, float& pS, float& pI, float& pR) {
    float qS = b2, qI = b2, qR = b2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 w = w3[k];
        float s = fmaf(w.x, yS.x, fmaf(w.y, yS.y, fmaf(w.z, yS.z, w.w * yS.w)));
        float i = fmaf(w.x, yI.x, fmaf(w.y, yI.y, fmaf(w.z, yI.z, w.w * yI.w)));
        float r = fmaf(w.x, yR.x, fmaf(w.y, yR.y, fmaf(w.z, yR.z, w.w * yR.w)));
        s = group_sum<LPR>(s) + b3[k];
        i = group_sum<LPR>(i) + b3[k];
        r = group_sum<LPR>(r) + b3[k];
        qS = fmaf(w2[k], fmaxf(s, 0.f), qS);
        qI = fmaf(w2[k], fmaxf(i, 0.f), qI);
        qR = fmaf(w2[k], fmaxf(r, 0.f), qR);
    }
    float m = fmaxf(qS, fmaxf(qI, qR));
    float eS = __expf(qS - m), eI = __expf(qI - m), eR = __expf(qR - m);
    float inv = __builtin_amdgcn_rcpf(eS + eI + eR);
    pS = eS * inv; pI = eI * inv; pR = eR * inv;
}

// Node MLP of one row: sigmoid(W x + b).  x_k is broadcast inside the group by shuffle and multiplied with W^T (staged in LDS,
// Wt[k][j] = W[j][k]); every lane of the wave must call it (shuffles), `active` guards the LDS reads.
template <int LPR>
__device__ __forceinline__ float4 group_mlp(float4 x, const float* __restrict__ Wt, float4 bias4, int sub, bool active,
                                            int H) {
    float4 acc = bias4;
    const float xv[4] = {x.x, x.y, x.z, x.w};
    for (int kk = 0; 4 * kk < H; ++kk) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float xk = __shfl(xv[c], kk, LPR);
            if (active) {
                const float4 w = ld4(Wt + (size_t)(4 * kk + c) * H + 4 * sub);
                acc.x = fmaf(xk, w.x, acc.x); acc.y = fmaf(xk, w.y, acc.y);
                acc.z = fmaf(xk, w.z, acc.z); acc.w = fmaf(xk, w.w, acc.w);
            }
        }
    }
    return make_float4(gn_sigmoid(acc.x), gn_sigmoid(acc.y), gn_sigmoid(acc.z), gn_sigmoid(acc.w));
}

