// Device-side building blocks of the H = 64 kernels (gnode_h64.hip, gnode_bwd.hip): LDS tile
// geometry, 16-lane DPP row helpers and the fp32 MFMA tile engine.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TS 68            // LDS row stride in floats (272 B): 16-B aligned rows, spreads banks
#define TILE_ROWS 32

// v_exp_f32 + v_rcp_f32 (1 ulp each).  __frcp_rn would expand to the IEEE-exact division sequence (~10 VALU
// instructions per value, 16 values per wave and tile) for nothing: the result is within 2e-7 either way.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4g(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// uniform base pointer + 32-bit BYTE offset: the compiler emits `global_load/store v, v_off, s[base:base+1]`, one
// 32-bit VALU op per address instead of a 64-bit shift-add pair (the step kernel is VALU-issue sensitive)
__device__ __forceinline__ float4 ld4o(const float* b, unsigned off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(b) + off);
}
__device__ __forceinline__ void st4o(float* b, unsigned off, float4 v) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(b) + off) = v;
}
// streaming (read-once / write-once) accesses: keep them out of the way of the gather table in L2
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld4s(const float* p) {
    if (NT) { const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return ld4g(p);
}
template <bool NT>
__device__ __forceinline__ void st4s(float* p, float4 v) {
    if (NT) { v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p)); }
    else st4g(p, v);
}

// the same through a uniform base + 32-bit byte offset
template <bool NT>
__device__ __forceinline__ float4 ld4so(const float* b, unsigned off) { return ld4s<NT>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(b) + off)); }
template <bool NT>
__device__ __forceinline__ void st4so(float* b, unsigned off, float4 v) { st4s<NT>(reinterpret_cast<float*>(reinterpret_cast<char*>(b) + off), v); }

// ---- DPP helpers: a 16-lane group is exactly one DPP row -----------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
// sum over the 16 lanes of a row, result in every lane: mirror, half-mirror, xor2, xor1
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_f<0x140>(v);   // row_mirror        i <-> 15-i
    v += dpp_f<0x141>(v);   // row_half_mirror   i <-> 7-i within each half
    v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
    return v;
}
// lane J of the row broadcast to the whole row
template <int J>
__device__ __forceinline__ int row_bcast(int v) { return dpp_i<0x150 + J>(v); }

// ---- MFMA tile engine: out[32][64] = sigmoid(X[32][64] W^T + b) -----------------------
// X in LDS tile Tin, W in LDS Wl ([64][TS]), result written to LDS tile Tout as [row][feature].
// Wave w owns output features [16w, 16w+16).  v_mfma_f32_16x16x4_f32 operand maps:
//   A[i][k'] / B[k'][j]: i = j = lane & 15, k' = lane >> 4;  D: col = lane & 15, row = 4*(lane>>4) + reg.
// k is visited as kappa = 16*(lane>>4) + 4m + c so every fragment fetch is one ds_read_b128.
// WT: contract with W instead of W^T (out = X W, the backward's g_Y = dpre W) from the SAME staged copy Wl[j][k]:
// the B fragment is then four ds_read_b32 a row apart instead of one ds_read_b128 -- LDS is far from the bound.
template <bool SIGMOID = true, bool WT = false>
__device__ __forceinline__ void mfma_tile(const float* __restrict__ Tin, const float* __restrict__ Wl,
                                          float* __restrict__ Tout, float bias_l, int w, int lane) {
    const int i = lane & 15, kq = lane >> 4;
    f32x4 acc0 = {bias_l, bias_l, bias_l, bias_l}, acc1 = acc0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        float4 b;
        if (WT) {
            const float* bp = Wl + (16 * kq + 4 * m) * TS + 16 * w + i;
            b = make_float4(bp[0], bp[TS], bp[2 * TS], bp[3 * TS]);
        } else {
            b = *reinterpret_cast<const float4*>(Wl + (16 * w + i) * TS + 16 * kq + 4 * m);
        }
        const float4 a0 = *reinterpret_cast<const float4*>(Tin + i * TS + 16 * kq + 4 * m);
        const float4 a1 = *reinterpret_cast<const float4*>(Tin + (16 + i) * TS + 16 * kq + 4 * m);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Tout[(4 * kq + r) * TS + 16 * w + i] = SIGMOID ? sigmoid_f(acc0[r]) : acc0[r];
        Tout[(16 + 4 * kq + r) * TS + 16 * w + i] = SIGMOID ? sigmoid_f(acc1[r]) : acc1[r];
    }
}

// 16-row variant: out[16][64] = sigmoid(X[16][64] W^T + b).  One 16x16 output tile per wave; K = 64 is split over
// two accumulators (k-steps of even / odd m) so the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 is hidden.
template <bool SIGMOID = true, bool WT = false>
__device__ __forceinline__ void mfma_tile16(const float* __restrict__ Tin, const float* __restrict__ Wl,
                                            float* __restrict__ Tout, float bias_l, int w, int lane) {
    const int i = lane & 15, kq = lane >> 4;
    f32x4 acc0 = {bias_l, bias_l, bias_l, bias_l}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; m += 2) {
        float4 b0, b1;
        if (WT) {
            const float* bp = Wl + (16 * kq + 4 * m) * TS + 16 * w + i;
            b0 = make_float4(bp[0], bp[TS], bp[2 * TS], bp[3 * TS]);
            b1 = make_float4(bp[4 * TS], bp[5 * TS], bp[6 * TS], bp[7 * TS]);
        } else {
            b0 = *reinterpret_cast<const float4*>(Wl + (16 * w + i) * TS + 16 * kq + 4 * m);
            b1 = *reinterpret_cast<const float4*>(Wl + (16 * w + i) * TS + 16 * kq + 4 * m + 4);
        }
        const float4 a0 = *reinterpret_cast<const float4*>(Tin + i * TS + 16 * kq + 4 * m);
        const float4 a1 = *reinterpret_cast<const float4*>(Tin + i * TS + 16 * kq + 4 * m + 4);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Tout[(4 * kq + r) * TS + 16 * w + i] = SIGMOID ? sigmoid_f(acc0[r] + acc1[r]) : acc0[r] + acc1[r];
}

// Wl[r][c] = W[r][c] (forward: out = X W^T), or TRANSPOSE: Wl[c][r] = W[r][c] (backward: out = X W)
template <bool TRANSPOSE = false>
__device__ __forceinline__ void load_W_to_lds(const float* __restrict__ W, float* __restrict__ Wl) {
    // 64x64 floats = 1024 float4: 4 per thread, coalesced
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = q * 256 + threadIdx.x;      // float4 index
        const int r = idx >> 4, c4 = idx & 15;
        const float4 v = ld4g(W + (size_t)idx * 4);
        if (TRANSPOSE) {
            Wl[(4 * c4 + 0) * TS + r] = v.x; Wl[(4 * c4 + 1) * TS + r] = v.y;
            Wl[(4 * c4 + 2) * TS + r] = v.z; Wl[(4 * c4 + 3) * TS + r] = v.w;
        } else {
            *reinterpret_cast<float4*>(Wl + r * TS + 4 * c4) = v;
        }
    }
}

