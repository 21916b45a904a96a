// Device-side pieces shared by the H = 64 Euler-step kernels (gnode_h64.hip: one launch per step; gnode_pers64.hip: all
// steps in one persistent launch): the fused read-out head and the dual 16-row node MLP on the matrix cores.
#pragma once
#include "gnode_mfma64.h"

// PRJ: the R compartment arrives already projected (prj[k] = w3[k] . Y_R, see k_step64's PRJ mode)
template <bool PRJ>
__device__ __forceinline__ void readout64(float4 yS, float4 yI, float4 yR, const float (&prj)[4], int sub,
                                          const float* __restrict__ w3, const float* __restrict__ b3,
                                          const float* __restrict__ w2, const float* __restrict__ b2, float& pS,
                                          float& pI, float& pR) {
    float qS = b2[0], qI = qS, qR = qS;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 wv = ld4g(w3 + k * 64 + 4 * sub);
        float s = fmaf(wv.x, yS.x, fmaf(wv.y, yS.y, fmaf(wv.z, yS.z, wv.w * yS.w)));
        float i = fmaf(wv.x, yI.x, fmaf(wv.y, yI.y, fmaf(wv.z, yI.z, wv.w * yI.w)));
        float r;
        s = row_sum16(s) + b3[k];
        i = row_sum16(i) + b3[k];
        if (PRJ) r = prj[k] + b3[k];
        else r = row_sum16(fmaf(wv.x, yR.x, fmaf(wv.y, yR.y, fmaf(wv.z, yR.z, wv.w * yR.w)))) + b3[k];
        qS = fmaf(w2[k], fmaxf(s, 0.f), qS);
        qI = fmaf(w2[k], fmaxf(i, 0.f), qI);
        qR = fmaf(w2[k], fmaxf(r, 0.f), qR);
    }
    const float m = fmaxf(qS, fmaxf(qI, qR));
    const float eS = __expf(qS - m), eI = __expf(qI - m), eR = __expf(qR - m);
    const float inv = __builtin_amdgcn_rcpf(eS + eI + eR);
    pS = eS * inv; pI = eI * inv; pR = eR * inv;
}

// 16-row dual node MLP: OA = sigmoid(XA W^T + b), OB = sigmoid(XB W^T + b); per output the same two accumulation chains
// (even / odd k-quads) as mfma_tile16, so the results are bit-identical to two separate calls.  All LDS addresses are
// ONE per-lane offset plus compile-time constants (they fold into the DS instructions' offset fields).
//   fo = (i*TS + 16*kq) floats: fragment offset inside a tile or inside W's 16-row slab;  oo = (4*kq*TS + i): result offset
template <bool DO_A, bool DO_B>
__device__ __forceinline__ void mfma_dual16(const float* __restrict__ XA, const float* __restrict__ XB,
                                            const float* __restrict__ Wslab, float* __restrict__ OA,
                                            float* __restrict__ OB, float bias_l, int fo, int oo) {
    f32x4 a0 = {bias_l, bias_l, bias_l, bias_l}, a1 = {0.f, 0.f, 0.f, 0.f}, c0 = a0, c1 = a1;
    // one k-quad of fragments at a time (12 registers): chain 0 takes the even quads, chain 1 the odd ones, each in
    // the order mfma_tile16 feeds them; the A and B streams alternate, so a chain's next MFMA is two issue slots away
#define GN_MQ(M, ACA, ACB) {                                                                                   \
        const float4 wv = *reinterpret_cast<const float4*>(Wslab + fo + 4 * (M));                              \
        float4 xa, xb;                                                                                         \
        if (DO_A) xa = *reinterpret_cast<const float4*>(XA + fo + 4 * (M));                                    \
        if (DO_B) xb = *reinterpret_cast<const float4*>(XB + fo + 4 * (M));                                    \
        if (DO_A) ACA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, wv.x, ACA, 0, 0, 0);                        \
        if (DO_B) ACB = __builtin_amdgcn_mfma_f32_16x16x4f32(xb.x, wv.x, ACB, 0, 0, 0);                        \
        if (DO_A) ACA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, wv.y, ACA, 0, 0, 0);                        \
        if (DO_B) ACB = __builtin_amdgcn_mfma_f32_16x16x4f32(xb.y, wv.y, ACB, 0, 0, 0);                        \
        if (DO_A) ACA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, wv.z, ACA, 0, 0, 0);                        \
        if (DO_B) ACB = __builtin_amdgcn_mfma_f32_16x16x4f32(xb.z, wv.z, ACB, 0, 0, 0);                        \
        if (DO_A) ACA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, wv.w, ACA, 0, 0, 0);                        \
        if (DO_B) ACB = __builtin_amdgcn_mfma_f32_16x16x4f32(xb.w, wv.w, ACB, 0, 0, 0); }
    GN_MQ(0, a0, c0) GN_MQ(1, a1, c1) GN_MQ(2, a0, c0) GN_MQ(3, a1, c1)
#undef GN_MQ
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (DO_A) OA[oo + r * TS] = sigmoid_f(a0[r] + a1[r]);
        if (DO_B) OB[oo + r * TS] = sigmoid_f(c0[r] + c1[r]);
    }
}

