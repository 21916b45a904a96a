// Adjoint backward of the GN-ODE path for MI355X (gfx950).
//
// Semantics: torchdiffeq 0.2.2 `odeint_adjoint(..., method='euler')` as the reference
// calls it (ode_nn_ngraph_sim.py:16,168).  With sol saved by the
// forward and a = dL/dsol accumulated backwards,
//     for i = G-1 .. 1:   u = a
//         Z = sigmoid(W y_i + b) (S and I slabs), AI = A Z_I
//         v  = beta (u_I - u_S)
//         dZ_S = v * AI                  dZ_I = A^T (v * Z_S) + gamma (u_R - u_I)      (A symmetric)
//         dpre = dZ * Z (1 - Z)
//         a_{S,I} += dt_i * dpre W       gW += dt_i * dpre^T y_i      gb += dt_i * sum_rows dpre
//         a += dL/dsol[i-1]                                            (head backward at grid point i-1)
//     encoder backward on a_0.
// The Jacobians are evaluated at the RIGHT endpoint y_i (one Euler step of the augmented
// system from t_i to t_{i-1}), which is what the reference's training gradients are.
//
// Launch structure: H = 64 and H <= 32 take ONE launch per interval (k_bwd_fused64 / k_bwd_fused_generic: the
// interval's VJPs, the head's VJP at grid point i-1 and the NEXT interval's Z / q gather tables, double-buffered);
// graphs with n <= 64 at H = 64 take one launch for the whole sweep (gnode_bwd_tiny.hip); other H: five launches
// per interval (k_mlp_generic, k_bwd_q, k_bwd_gather, k_bwd_mlp, k_head_bwd).
//
// Parameter gradients are reduced deterministically: every workgroup owns one slot of a
// partial buffer [NWG][NP] that it updates with plain read-modify-writes across all
// launches (fixed grid, fixed row->workgroup map); one final kernel sums the slots in
// order.  No float atomics, bitwise reproducible run to run.
#include "gnode_bwd.h"
#include "gnode_h64.h"
#include "gnode_gather.h"
#include "gnode_mfma64.h"
#include "gnode_head64.h"
#include "gnode_pers64.h"
#include "gnode_persg.h"
#include <algorithm>

__device__ __forceinline__ float4 ld4b(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4b(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 z4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }

template <int LPR>
__device__ __forceinline__ float gsum(float v) {
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, LPR);
    return v;
}

// Deterministic in-workgroup reduction of per-group contributions staged in LDS:
// red[group][NE] -> part[e] += sum_groups (fixed order).
__device__ __forceinline__ void flush_groups(const float* red, int ngroups, int ne, float* __restrict__ part) {
    for (int e = threadIdx.x; e < ne; e += 256) {
        float s = 0.f;
        for (int gidx = 0; gidx < ngroups; ++gidx) s += red[(size_t)gidx * ne + e];
        part[e] += s;
    }
}

// --------------------------------------------------------------------------- head backward at one grid point
// a[3][rows][H] += d(readout+softmax)/dY ; partial grads of linear3 / linearS2.
template <int LPR>
__global__ __launch_bounds__(256) void k_head_bwd(const float* __restrict__ Ysol, long rows, int H,
                                                  const float* __restrict__ gS, const float* __restrict__ gI,
                                                  const float* __restrict__ gR, const float* __restrict__ w3,
                                                  const float* __restrict__ b3, const float* __restrict__ w2,
                                                  const float* __restrict__ b2, float* __restrict__ a,
                                                  float* __restrict__ part_all) {
    extern __shared__ float red[];                       // [G][4H + 9]
    constexpr int G = 256 / LPR;
    const PartLayout L{H};
    const int ne = 4 * H + 9;
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const bool active = 4 * sub < H;
    const size_t slab = (size_t)rows * H;
    float4 w3v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w3v[k] = active ? ld4b(w3 + (size_t)k * H + 4 * sub) : z4();
    float4 dw3[4] = {z4(), z4(), z4(), z4()};
    float db3[4] = {0.f, 0.f, 0.f, 0.f}, dw2[4] = {0.f, 0.f, 0.f, 0.f}, db2 = 0.f;
    for (long r = (long)blockIdx.x * G + grp; r < rows; r += (long)gridDim.x * G) {
        const size_t off = (size_t)r * H + 4 * sub;
        float4 y[3];
#pragma unroll
        for (int X = 0; X < 3; ++X) y[X] = active ? ld4b(Ysol + X * slab + off) : z4();
        float p3[3][4], q[3];
#pragma unroll
        for (int X = 0; X < 3; ++X) {
            q[X] = b2[0];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                p3[X][k] = gsum<LPR>(dot4(w3v[k], y[X])) + b3[k];
                q[X] = fmaf(w2[k], fmaxf(p3[X][k], 0.f), q[X]);
            }
        }
        const float m = fmaxf(q[0], fmaxf(q[1], q[2]));
        float e0 = __expf(q[0] - m), e1 = __expf(q[1] - m), e2 = __expf(q[2] - m);
        const float inv = 1.0f / (e0 + e1 + e2);
        const float p[3] = {e0 * inv, e1 * inv, e2 * inv};
        const float g[3] = {gS[r], gI[r], gR[r]};
        const float gp = g[0] * p[0] + g[1] * p[1] + g[2] * p[2];
#pragma unroll
        for (int X = 0; X < 3; ++X) {
            const float dq = p[X] * (g[X] - gp);
            float4 dy = z4();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dp3 = p3[X][k] > 0.f ? dq * w2[k] : 0.f;
                dy.x = fmaf(dp3, w3v[k].x, dy.x); dy.y = fmaf(dp3, w3v[k].y, dy.y);
                dy.z = fmaf(dp3, w3v[k].z, dy.z); dy.w = fmaf(dp3, w3v[k].w, dy.w);
                dw3[k].x = fmaf(dp3, y[X].x, dw3[k].x); dw3[k].y = fmaf(dp3, y[X].y, dw3[k].y);
                dw3[k].z = fmaf(dp3, y[X].z, dw3[k].z); dw3[k].w = fmaf(dp3, y[X].w, dw3[k].w);
                db3[k] += dp3;
                dw2[k] = fmaf(dq, fmaxf(p3[X][k], 0.f), dw2[k]);
            }
            db2 += dq;
            if (active) {
                float4 av = ld4b(a + X * slab + off);
                av.x += dy.x; av.y += dy.y; av.z += dy.z; av.w += dy.w;
                st4b(a + X * slab + off, av);
            }
        }
    }
    float* mine = red + (size_t)grp * ne;
    if (active)
#pragma unroll
        for (int k = 0; k < 4; ++k) st4b(mine + k * H + 4 * sub, dw3[k]);
    if (sub == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { mine[4 * H + k] = db3[k]; mine[4 * H + 4 + k] = dw2[k]; }
        mine[4 * H + 8] = db2;
    }
    __syncthreads();
    flush_groups(red, G, ne, part_all + (size_t)blockIdx.x * L.total() + L.ow3());
}

// --------------------------------------------------------------------------- q = beta (a_I - a_S) * Z_S
__global__ __launch_bounds__(256) void k_bwd_q(const float* __restrict__ a, const float* __restrict__ ZS,
                                               const float* __restrict__ beta, float* __restrict__ q, long rows, int H) {
    const size_t slab = (size_t)rows * H, n4 = slab / 4;
    const int h4 = H / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float bt = beta[i / h4];
        const float4 aS = ld4b(a + 4 * i), aI = ld4b(a + slab + 4 * i), z = ld4b(ZS + 4 * i);
        st4b(q + 4 * i, make_float4(bt * (aI.x - aS.x) * z.x, bt * (aI.y - aS.y) * z.y, bt * (aI.z - aS.z) * z.z,
                                    bt * (aI.w - aS.w) * z.w));
    }
}

// --------------------------------------------------------------------------- gathers + dpre
template <int LPR>
__global__ __launch_bounds__(256) void k_bwd_gather(const int* __restrict__ rowptr, const int* __restrict__ col, int n,
                                                    long rows, int H, const float* __restrict__ a,
                                                    const float* __restrict__ Z, const float* __restrict__ q,
                                                    const float* __restrict__ beta, const float* __restrict__ gamma,
                                                    float* __restrict__ dpre, const int* __restrict__ hubidx,
                                                    const float* __restrict__ AIhub, const float* __restrict__ GQhub, int n_hub) {
    const int sub = threadIdx.x % LPR;
    const int node = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (node >= n) return;
    const bool active = 4 * sub < H;
    const long base = (long)blockIdx.y * n, r = base + node;
    const size_t slab = (size_t)rows * H, off = (size_t)r * H + 4 * sub;
    const float* ZI = Z + slab;
    float4 ai = z4(), gq = z4();
    const int hub = hubidx ? hubidx[node] : -1;
    if (hub >= 0 && active) {
        ai = ld4b(AIhub + ((size_t)blockIdx.y * n_hub + hub) * H + 4 * sub);
        gq = ld4b(GQhub + ((size_t)blockIdx.y * n_hub + hub) * H + 4 * sub);
    }
    const int start = hub >= 0 ? 0 : rowptr[node], end = hub >= 0 ? 0 : rowptr[node + 1];
    gn_gather2<4>(col, start, end, ZI + (size_t)base * H, q + (size_t)base * H, H, sub, active, ai, gq);
    if (!active) return;
    const float bt = beta[r], gm = gamma[r];
    const float4 aS = ld4b(a + off), aI = ld4b(a + slab + off), aR = ld4b(a + 2 * slab + off);
    const float4 zs = ld4b(Z + off), zi = ld4b(ZI + off);
    float4 dS, dI;
#define GN_DPRE(c)                                                         \
    {                                                                      \
        const float v = bt * (aI.c - aS.c);                                \
        dS.c = (v * ai.c) * (zs.c * (1.0f - zs.c));                        \
        dI.c = (gq.c + gm * (aR.c - aI.c)) * (zi.c * (1.0f - zi.c));       \
    }
    GN_DPRE(x) GN_DPRE(y) GN_DPRE(z) GN_DPRE(w)
#undef GN_DPRE
    st4b(dpre + off, dS);
    st4b(dpre + slab + off, dI);
}

// --------------------------------------------------------------------------- a += dt dpre W ; partial gW, gb
template <int LPR>
__global__ __launch_bounds__(256) void k_bwd_mlp(const float* __restrict__ dpre, const float* __restrict__ Ysol,
                                                 const float* __restrict__ W, float dt, float* __restrict__ a, long rows,
                                                 int H, float* __restrict__ part_all) {
    extern __shared__ float lds[];
    constexpr int G = 256 / LPR;
    const PartLayout L{H};
    float* Wl = lds;                           // [H][H]   W[j][k]
    float* Dt = Wl + (size_t)H * H;            // [2][G][H] dpre tile (S, I)
    float* Yt = Dt + (size_t)2 * G * H;        // [2][G][H] y tile   (S, I)
    for (int idx = threadIdx.x; idx < H * H; idx += 256) Wl[idx] = W[idx];
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const bool lane_ok = 4 * sub < H;
    const size_t slab = (size_t)rows * H;
    const int nE = H * H;
    constexpr int MAXM = (LPR * LPR / 16) < 1 ? 1 : (LPR * LPR / 16);   // gW entries per thread: H*H/256 with H <= 4*LPR
    float accW[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) accW[m] = 0.f;
    float accb = 0.f;
    const int M = (nE + 255) / 256;
    for (long r0 = (long)blockIdx.x * G; r0 < rows; r0 += (long)gridDim.x * G) {
        const long r = r0 + grp;
        const bool ok = lane_ok && r < rows;
        const size_t off = (size_t)r * H + 4 * sub;
        __syncthreads();                       // previous tile fully consumed (also covers the W stage)
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            if (!lane_ok) continue;            // idle lanes of a group (H/4 not a power of two) would write into the next row
            st4b(Dt + ((size_t)X * G + grp) * H + 4 * sub, ok ? ld4b(dpre + X * slab + off) : z4());
            st4b(Yt + ((size_t)X * G + grp) * H + 4 * sub, ok ? ld4b(Ysol + X * slab + off) : z4());
        }
        __syncthreads();
        // g_Y = dpre W  (this lane: 4 columns of its own row, both slabs)
        float4 gS = z4(), gI = z4();
        const float* dS = Dt + (size_t)grp * H;
        const float* dI = Dt + ((size_t)G + grp) * H;
        for (int j = 0; j < H; ++j) {
            const float4 w = lane_ok ? ld4b(Wl + (size_t)j * H + 4 * sub) : z4();
            const float s = dS[j], i = dI[j];
            gS.x = fmaf(s, w.x, gS.x); gS.y = fmaf(s, w.y, gS.y); gS.z = fmaf(s, w.z, gS.z); gS.w = fmaf(s, w.w, gS.w);
            gI.x = fmaf(i, w.x, gI.x); gI.y = fmaf(i, w.y, gI.y); gI.z = fmaf(i, w.z, gI.z); gI.w = fmaf(i, w.w, gI.w);
        }
        if (ok) {
            float4 aS = ld4b(a + off), aI = ld4b(a + slab + off);
            aS.x += dt * gS.x; aS.y += dt * gS.y; aS.z += dt * gS.z; aS.w += dt * gS.w;
            aI.x += dt * gI.x; aI.y += dt * gI.y; aI.z += dt * gI.z; aI.w += dt * gI.w;
            st4b(a + off, aS); st4b(a + slab + off, aI);
        }
        // gW[j][k] += sum_rows dpre[r][j] * y[r][k]   (thread owns entries e = tid + 256 m)
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            if (m < M) {
                const int e = threadIdx.x + 256 * m;
                if (e < nE) {
                    const int j = e / H, k = e % H;
                    float s = 0.f;
                    for (int rr = 0; rr < 2 * G; ++rr) s = fmaf(Dt[(size_t)rr * H + j], Yt[(size_t)rr * H + k], s);
                    accW[m] += s;
                }
            }
        }
        if (threadIdx.x < H) {
            float s = 0.f;
            for (int rr = 0; rr < 2 * G; ++rr) s += Dt[(size_t)rr * H + threadIdx.x];
            accb += s;
        }
    }
    float* part = part_all + (size_t)blockIdx.x * L.total();
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        if (m < M) {
            const int e = threadIdx.x + 256 * m;
            if (e < nE) part[L.oW() + e] += dt * accW[m];
        }
    }
    if (threadIdx.x < H) part[L.ob() + threadIdx.x] += dt * accb;
}

// --------------------------------------------------------------------------- H = 64 fused backward step
// Per 32-row tile and slab X in {S, I}, on the fp32 matrix cores:
//   gW tile-accumulate  dW[j][k] += sum_r dpre_X[r][j] * y_X[r][k]   (wave w owns rows j in [16w,16w+16),
//                        4 accumulator tiles kept in registers across ALL tiles of the workgroup)
//   g_Y = dpre_X W       (mfma_tile with W staged transposed), written over the y tile, then
//   a_X += dt * g_Y      in the coalesced row layout.
// Z = sigmoid(y_i W^T + b) for the S and I slabs (same MFMA tile engine as the forward) with the
// S-slab epilogue also producing q = beta (a_I - a_S) * Z_S, the operand of the transposed gather.
__global__ __launch_bounds__(256) void k_mlp64_q(const float* __restrict__ X, const float* __restrict__ W,
                                                 const float* __restrict__ bias, float* __restrict__ Z,
                                                 const float* __restrict__ a, const float* __restrict__ beta,
                                                 float* __restrict__ q, long rows) {
    __shared__ __attribute__((aligned(16))) float Wl[64 * TS];
    __shared__ __attribute__((aligned(16))) float T[TILE_ROWS * TS];
    __shared__ __attribute__((aligned(16))) float T2[TILE_ROWS * TS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, sub = lane & 15;
    load_W_to_lds<false>(W, Wl);
    const float bias_l = bias[16 * w + (lane & 15)];
    const long nrows = 2 * rows, ntiles = (nrows + TILE_ROWS - 1) / TILE_ROWS;
    const size_t slab = (size_t)rows * 64;
    const int lr[2] = {w * 8 + g, w * 8 + 4 + g};
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        long r[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            r[p] = t * TILE_ROWS + lr[p];
            *reinterpret_cast<float4*>(T + lr[p] * TS + 4 * sub) = r[p] < nrows ? ld4g(X + (size_t)r[p] * 64 + 4 * sub) : zero4();
        }
        __syncthreads();
        mfma_tile<true>(T, Wl, T2, bias_l, w, lane);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (r[p] >= nrows) continue;
            const float4 z = *reinterpret_cast<const float4*>(T2 + lr[p] * TS + 4 * sub);
            const size_t off = (size_t)r[p] * 64 + 4 * sub;
            st4g(Z + off, z);
            if (r[p] < rows) {                                   // S slab: q rides along
                const float bt = beta[r[p]];
                const float4 aS = ld4g(a + off), aI = ld4g(a + slab + off);
                st4g(q + off, make_float4(bt * (aI.x - aS.x) * z.x, bt * (aI.y - aS.y) * z.y, bt * (aI.z - aS.z) * z.z,
                                          bt * (aI.w - aS.w) * z.w));
            }
        }
        __syncthreads();
    }
}

// two tables gathered through the same neighbour list (AI = A Z_I, Gq = A q), ascending column order.
// NB neighbours (2 NB row loads) are in flight per lane group.
template <int NB = 2>
__device__ __forceinline__ void gather2_row64(const int* __restrict__ rowptr, const int* __restrict__ col,
                                              const float* __restrict__ T0, const float* __restrict__ T1, int node,
                                              bool valid, int sub, float4& acc0, float4& acc1) {
    acc0 = zero4(); acc1 = zero4();
    int start = 0, end = 0;
    if (valid) { start = rowptr[node]; end = rowptr[node + 1]; }
    const unsigned lane_b = 16u * sub;
    for (int e0 = start; e0 < end; e0 += 16) {
        const int cnt = min(16, end - e0);
        const unsigned mine = (sub < cnt) ? (unsigned)col[e0 + sub] * 256u : 0u;     // byte offset of the neighbour row
#define GN_LD2(K, U, V)                                                                            \
            float4 U = zero4(), V = zero4();                                                       \
            if (K < cnt) { const unsigned o = (unsigned)row_bcast<(K) & 15>((int)mine) + lane_b; U = ld4o(T0, o); V = ld4o(T1, o); }
#define GN_AC2(U, V)                                                                               \
            acc0.x += U.x; acc0.y += U.y; acc0.z += U.z; acc0.w += U.w;                            \
            acc1.x += V.x; acc1.y += V.y; acc1.z += V.z; acc1.w += V.w;
#define GN_G2(J)                                                                                   \
        if (J < cnt) {                                                                             \
            GN_LD2(J, u0, v0) GN_LD2(J + 1, u1, v1)                                                \
            GN_AC2(u0, v0) GN_AC2(u1, v1)                                                          \
        }
#define GN_G4(J)                                                                                   \
        if (J < cnt) {                                                                             \
            GN_LD2(J, u0, v0) GN_LD2(J + 1, u1, v1) GN_LD2(J + 2, u2, v2) GN_LD2(J + 3, u3, v3)    \
            GN_AC2(u0, v0) GN_AC2(u1, v1) GN_AC2(u2, v2) GN_AC2(u3, v3)                            \
        }
        if (NB == 2) { GN_G2(0) GN_G2(2) GN_G2(4) GN_G2(6) GN_G2(8) GN_G2(10) GN_G2(12) GN_G2(14) }
        else { GN_G4(0) GN_G4(4) GN_G4(8) GN_G4(12) }
#undef GN_G4
#undef GN_AC2
#undef GN_LD2
#undef GN_G2
    }
}

// one table (Gq = A q) when the forward kept A Z_I(y_i): 2 NB row loads in flight per lane group, ascending column order
template <int NB = 4>
__device__ __forceinline__ float4 gather1_row64(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                const float* __restrict__ T0, int node, bool valid, int sub) {
    float4 acc = zero4();
    int start = 0, end = 0;
    if (valid) { start = rowptr[node]; end = rowptr[node + 1]; }
    const unsigned lane_b = 16u * sub;
    for (int e0 = start; e0 < end; e0 += 16) {
        const int cnt = min(16, end - e0);
        const unsigned mine = (sub < cnt) ? (unsigned)col[e0 + sub] * 256u : 0u;
#define GN_LD1(K, U) float4 U = zero4(); if (K < cnt) U = ld4o(T0, (unsigned)row_bcast<(K) & 15>((int)mine) + lane_b);
#define GN_AC1(U) acc.x += U.x; acc.y += U.y; acc.z += U.z; acc.w += U.w;
#define GN_G8(J)                                                                                   \
        if (J < cnt) {                                                                             \
            GN_LD1(J, u0) GN_LD1(J + 1, u1) GN_LD1(J + 2, u2) GN_LD1(J + 3, u3)                    \
            GN_LD1(J + 4, u4) GN_LD1(J + 5, u5) GN_LD1(J + 6, u6) GN_LD1(J + 7, u7)                \
            GN_AC1(u0) GN_AC1(u1) GN_AC1(u2) GN_AC1(u3) GN_AC1(u4) GN_AC1(u5) GN_AC1(u6) GN_AC1(u7) \
        }
        GN_G8(0) GN_G8(8)
#undef GN_G8
#undef GN_AC1
#undef GN_LD1
    }
    return acc;
}

#ifndef GN_BWD_RPG1_OCC
#define GN_BWD_RPG1_OCC 3
#endif
#ifndef GN_BWD_NB
#define GN_BWD_NB 4          // neighbours in flight per lane group in the fused kernel's two-table gather (2: 473 us per
                             // interval on the 75k graph x 4, 4: 456 us; mid-size train steps -5 %)
#endif
static_assert(GN_BWD_RPG1_OCC * 256 <= BWD_NWG, "fused backward grid exceeds the partial-gradient slots");
// GATHER_AI: gather A Z_I as well (the last grid point, and trajectories whose 4th slabs do not carry it); otherwise the
// row's A Z_I(y_i) is read back from `AIsaved` (the forward kept it) and only A q is gathered: one table instead of two.
template <int OCC, int RPG, bool GATHER_AI>
__global__ __launch_bounds__(256, OCC) void k_bwd_fused64(const int* __restrict__ rowptr, const int* __restrict__ col, int n,
                                                     long rows, int tiles_per_sample, long total_tiles,
                                                     const float* __restrict__ ZIc, const float* __restrict__ Qc,
                                                     float* __restrict__ ZIn /* next interval's Z_I table, or null when it
                                                     will read the kept A Z_I instead of gathering */,
                                                     float* __restrict__ Qn, const float* __restrict__ Ysol,
                                                     const float* __restrict__ Yprev, const float* __restrict__ W,
                                                     const float* __restrict__ bias, const float* __restrict__ beta,
                                                     const float* __restrict__ gamma, float dt, float* __restrict__ a,
                                                     float* __restrict__ part_all, const float* __restrict__ gS,
                                                     const float* __restrict__ gI, const float* __restrict__ gR,
                                                     const float* __restrict__ w3, const float* __restrict__ b3,
                                                     const float* __restrict__ w2, const float* __restrict__ b2,
                                                     const int* __restrict__ hubidx, const float* __restrict__ AIhub,
                                                     const float* __restrict__ GQhub, int n_hub, int do_next,
                                                     const float* __restrict__ AIsaved) {
    __shared__ __attribute__((aligned(16))) float Wl[64 * TS];
    constexpr int TR = 16 * RPG;                   // rows per tile (RPG rows per 16-lane group)
    __shared__ __attribute__((aligned(16))) float tiles[4][TR * TS];
    float (*Dt)[TR * TS] = &tiles[0];              // Dt[0..1]: dpre_S, dpre_I;  Yt[0..1]: y_S, y_I / g_Y
    float (*Yt)[TR * TS] = &tiles[2];
    const PartLayout L{64};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, sub = lane & 15;
    const int i = lane & 15, kq = lane >> 4;
    load_W_to_lds<false>(W, Wl);
    const float bias_l = bias[16 * w + i];
    const size_t slab = (size_t)rows * 64;
    int lr[RPG];
#pragma unroll
    for (int p = 0; p < RPG; ++p) lr[p] = w * 4 * RPG + 4 * p + g;
    const bool head = gS != nullptr;
    HeadAcc hacc;
#pragma unroll
    for (int k = 0; k < 4; ++k) { hacc.dw3[k] = zero4(); hacc.db3[k] = 0.f; hacc.dw2[k] = 0.f; }
    hacc.db2 = 0.f;
    f32x4 accW[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) accW[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;
    // Tile walk as in the forward's step kernel: the tile range is cut into 8 contiguous queues and workgroup i serves
    // queue i % 8 -- workgroups are dealt round-robin over the 8 XCDs, so one XCD's L2 sees the q table of ONE sample
    // (or half of one) instead of a slice of every sample's (L2 read hit rate of the interval kernel on the 75k graph x 4:
    // 8 % with the flat walk).  The partial-gradient slot stays the workgroup's own (blockIdx.x): placement only.
    const int xq = (gridDim.x % 8 == 0 && total_tiles >= 32) ? 8 : 1;
    const long q_lo = total_tiles * (blockIdx.x % xq) / xq, q_hi = total_tiles * (blockIdx.x % xq + 1) / xq;
    for (long t = q_lo + blockIdx.x / xq; t < q_hi; t += gridDim.x / xq) {
        const long b = t / tiles_per_sample;
        const int tile = (int)(t - b * tiles_per_sample);
        const long base = b * n;
        bool valid[RPG]; size_t off[RPG]; float4 aS[RPG], aI[RPG], aR[RPG], ai[RPG], gq[RPG]; float bt[RPG], gmv[RPG];
        __syncthreads();                                   // previous tile fully consumed (and W staged)
#pragma unroll
        for (int p = 0; p < RPG; ++p) {
            const int node = tile * TR + lr[p];
            valid[p] = node < n;
            off[p] = (size_t)(base + node) * 64 + 4 * sub;
            // own-row loads first: they travel under the gather's dependent id -> row round trips.  The row's own
            // Z_S(y_i), Z_I(y_i) are RECOMPUTED from the y_i rows staged for the gW contraction anyway (one more matrix
            // phase) instead of being written by the previous interval and read back: 4 of 17 slab transfers less
            float4 ysr = zero4(), yir = zero4();
            gmv[p] = 0.f;
            aS[p] = zero4(); aI[p] = zero4(); aR[p] = zero4(); bt[p] = 0.f;
            if (valid[p]) {
                bt[p] = beta[base + node];
                gmv[p] = gamma[base + node];
                aS[p] = ld4g(a + off[p]); aI[p] = ld4g(a + slab + off[p]); aR[p] = ld4g(a + 2 * slab + off[p]);
                ysr = ld4g(Ysol + off[p]); yir = ld4g(Ysol + slab + off[p]);
            }
            *reinterpret_cast<float4*>(&Yt[0][lr[p] * TS + 4 * sub]) = ysr;
            *reinterpret_cast<float4*>(&Yt[1][lr[p] * TS + 4 * sub]) = yir;
            const int hub = (hubidx && valid[p]) ? hubidx[node] : -1;
            if (GATHER_AI) {
                if (hub >= 0) {
                    ai[p] = ld4g(AIhub + ((size_t)b * n_hub + hub) * 64 + 4 * sub);
                    gq[p] = ld4g(GQhub + ((size_t)b * n_hub + hub) * 64 + 4 * sub);
                } else {
                    gather2_row64<GN_BWD_NB>(rowptr, col, ZIc + (size_t)base * 64, Qc + (size_t)base * 64, node, valid[p], sub, ai[p], gq[p]);
                }
            } else {
                ai[p] = valid[p] ? ld4g(AIsaved + off[p]) : zero4();
                if (hub >= 0) gq[p] = ld4g(GQhub + ((size_t)b * n_hub + hub) * 64 + 4 * sub);
                else gq[p] = gather1_row64<>(rowptr, col, Qc + (size_t)base * 64, node, valid[p], sub);
            }
        }
        __syncthreads();
        // Z_S(y_i), Z_I(y_i) of the tile's rows on the matrix cores (the engine and summation order of the forward)
        if (RPG == 2) { mfma_tile<true>(Yt[0], Wl, Dt[0], bias_l, w, lane); mfma_tile<true>(Yt[1], Wl, Dt[1], bias_l, w, lane); }
        else { mfma_tile16<true>(Yt[0], Wl, Dt[0], bias_l, w, lane); mfma_tile16<true>(Yt[1], Wl, Dt[1], bias_l, w, lane); }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < RPG; ++p) {
            float4 dS = zero4(), dI = zero4();
            if (valid[p]) {
                const float4 zs = *reinterpret_cast<const float4*>(&Dt[0][lr[p] * TS + 4 * sub]);
                const float4 zi = *reinterpret_cast<const float4*>(&Dt[1][lr[p] * TS + 4 * sub]);
                const float gm = gmv[p];
#define GN_DP(c)                                                               \
                {                                                              \
                    const float v = bt[p] * (aI[p].c - aS[p].c);               \
                    dS.c = (v * ai[p].c) * (zs.c * (1.0f - zs.c));             \
                    dI.c = (gq[p].c + gm * (aR[p].c - aI[p].c)) * (zi.c * (1.0f - zi.c)); \
                }
                GN_DP(x) GN_DP(y) GN_DP(z) GN_DP(w)
#undef GN_DP
            }
            *reinterpret_cast<float4*>(&Dt[0][lr[p] * TS + 4 * sub]) = dS;      // own row only: read above, rewritten here
            *reinterpret_cast<float4*>(&Dt[1][lr[p] * TS + 4 * sub]) = dI;
        }
        __syncthreads();
#pragma unroll
        for (int X = 0; X < 2; ++X) {
#pragma unroll
            for (int s8 = 0; s8 < 4 * RPG; ++s8) {
                const int rr = 4 * s8 + kq;
                const float av = Dt[X][rr * TS + 16 * w + i];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    accW[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Yt[X][rr * TS + 16 * kt + i], accW[kt], 0, 0, 0);
            }
        }
        if (threadIdx.x < 64) {
            float sacc = 0.f;
            for (int rr = 0; rr < TR; ++rr) sacc += Dt[0][rr * TS + threadIdx.x] + Dt[1][rr * TS + threadIdx.x];
            accb += sacc;
        }
        __syncthreads();
        if (RPG == 2) {                                                // g_Y = dpre W
            mfma_tile<false, true>(Dt[0], Wl, Yt[0], 0.f, w, lane); mfma_tile<false, true>(Dt[1], Wl, Yt[1], 0.f, w, lane);
        } else {
            mfma_tile16<false, true>(Dt[0], Wl, Yt[0], 0.f, w, lane); mfma_tile16<false, true>(Dt[1], Wl, Yt[1], 0.f, w, lane);
        }
        __syncthreads();
        // a += dt g_Y; then, row by row: fetch y_{i-1}, park its S/I rows in the (now free) own rows of Yt for the next
        // tables, and take dL/dsol[i-1] through the head.  Padding rows carry y = 0, gout = 0 and add nothing (the
        // head's row sums are group-wide, so lanes stay converged).
#pragma unroll
        for (int p = 0; p < RPG; ++p) {
            float4 y[3] = {zero4(), zero4(), zero4()};
            float gout[3] = {0.f, 0.f, 0.f};
            if (valid[p]) {
                const float4 uS = *reinterpret_cast<const float4*>(&Yt[0][lr[p] * TS + 4 * sub]);
                const float4 uI = *reinterpret_cast<const float4*>(&Yt[1][lr[p] * TS + 4 * sub]);
                aS[p].x += dt * uS.x; aS[p].y += dt * uS.y; aS[p].z += dt * uS.z; aS[p].w += dt * uS.w;
                aI[p].x += dt * uI.x; aI[p].y += dt * uI.y; aI[p].z += dt * uI.z; aI[p].w += dt * uI.w;
                if (head || do_next) y[0] = ld4g(Yprev + off[p]);
                if (head || (do_next && ZIn)) y[1] = ld4g(Yprev + slab + off[p]);
                if (head) {
                    y[2] = ld4g(Yprev + 2 * slab + off[p]);
                    const size_t o = (size_t)(base + tile * TR + lr[p]);
                    gout[0] = gS[o]; gout[1] = gI[o]; gout[2] = gR[o];
                }
            }
            if (do_next) {
                *reinterpret_cast<float4*>(&Yt[0][lr[p] * TS + 4 * sub]) = y[0];
                if (ZIn) *reinterpret_cast<float4*>(&Yt[1][lr[p] * TS + 4 * sub]) = y[1];
            }
            if (head) {
                float4 w3v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) w3v[k] = ld4g(w3 + k * 64 + 4 * sub);      // L1-resident
                head_vjp64(y, gout, w3v, b3, w2, b2, aS[p], aI[p], aR[p], hacc);
                if (valid[p]) st4g(a + 2 * slab + off[p], aR[p]);
            }
            if (valid[p]) { st4g(a + off[p], aS[p]); st4g(a + slab + off[p], aI[p]); }
        }
        if (do_next) {
            // q = beta (a_I - a_S) * Z_S(y_{i-1}): the table the NEXT interval gathers (its own-row Z_S, Z_I are recomputed there)
            __syncthreads();
            if (RPG == 2) mfma_tile<true>(Yt[0], Wl, Dt[0], bias_l, w, lane);
            else mfma_tile16<true>(Yt[0], Wl, Dt[0], bias_l, w, lane);
            if (ZIn) {                                                   // uniform: the next interval gathers A Z_I itself
                if (RPG == 2) mfma_tile<true>(Yt[1], Wl, Dt[1], bias_l, w, lane);
                else mfma_tile16<true>(Yt[1], Wl, Dt[1], bias_l, w, lane);
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < RPG; ++p) {
                if (!valid[p]) continue;
                if (ZIn) st4g(ZIn + off[p], *reinterpret_cast<const float4*>(&Dt[1][lr[p] * TS + 4 * sub]));
                const float4 zs = *reinterpret_cast<const float4*>(&Dt[0][lr[p] * TS + 4 * sub]);
                st4g(Qn + off[p], make_float4(bt[p] * (aI[p].x - aS[p].x) * zs.x, bt[p] * (aI[p].y - aS[p].y) * zs.y,
                                              bt[p] * (aI[p].z - aS[p].z) * zs.z, bt[p] * (aI[p].w - aS[p].w) * zs.w));
            }
        }
    }
    float* part = part_all + (size_t)blockIdx.x * L.total();
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            part[L.oW() + (16 * w + 4 * kq + reg) * 64 + 16 * kt + i] += dt * accW[kt][reg];
    if (threadIdx.x < 64) part[L.ob() + threadIdx.x] += dt * accb;
    if (head) {
        // lane-group partials of the head's parameter gradients -> this workgroup's slot, fixed order
        __syncthreads();
        constexpr int NE = 4 * 64 + 12;               // 4*64 + 9 used, rows kept 16-B aligned
        float* red = &tiles[0][0];                    // 16 groups x 268 floats = 17 152 B <= the four 16-row tiles (17 408 B)
        float* mine = red + (size_t)(threadIdx.x >> 4) * NE;
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(mine + k * 64 + 4 * sub) = hacc.dw3[k];
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { mine[256 + k] = hacc.db3[k]; mine[260 + k] = hacc.dw2[k]; }
            mine[264] = hacc.db2;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 265; e += 256) {
            float s = 0.f;
            for (int gi = 0; gi < 16; ++gi) s += red[(size_t)gi * NE + e];
            part[L.ow3() + e] += s;
        }
    }
}

// --------------------------------------------------------------------------- the interval kernel over KEPT activations
// k_bwd_fused64 is FP32-issue bound: on gfx950 the exact-fp32 MFMA and the VALU share the fp32 lanes, and it spends them on
// seven 64x64 products and 192 sigmoids per row.  Three of the products and every sigmoid only RECOMPUTE what the training
// forward had in registers: Z_S(y_i), Z_I(y_i) (for sigma') and Z_S(y_{i-1}) (for the next interval's q table).  When the
// forward kept them (gnode_forward_f32's `keep`, gn_keep_zs / gn_keep_zi) this kernel reads them back -- 2.5 slabs more
// traffic per interval for 43 % fewer matrix instructions and no transcendental at all -- and what is left needs two
// barriers per tile instead of seven: rows in, dpre -> LDS | gW += dpre^T y and g_Y = dpre W | rows out.
#ifndef GN_BWD_KEPT_HEAD_OCC
#define GN_BWD_KEPT_HEAD_OCC 2
#endif
// HUBS: the graph has rows longer than the hub threshold (compile-time, so graphs without them carry none of that code)
template <int OCC, bool HEAD, bool HUBS>
__global__ __launch_bounds__(256, OCC) void k_bwd_kept64(const int* __restrict__ rowhdr, const int* __restrict__ col, int n, long rows,
                                                    int tiles_per_sample, long total_tiles, const float* __restrict__ Qc,
                                                    float* __restrict__ Qn, const float* __restrict__ Ysol,
                                                    const float* __restrict__ Yprev, const float* __restrict__ PSk /* kept P_S(y_i) */,
                                                    const float* __restrict__ ZIk, const float* __restrict__ ZSp,
                                                    const float* __restrict__ W,
                                                    const float* __restrict__ beta, const float* __restrict__ gamma, float dt,
                                                    float* __restrict__ a, float* __restrict__ part_all,
                                                    const float* __restrict__ gS, const float* __restrict__ gI,
                                                    const float* __restrict__ gR, const float* __restrict__ w3,
                                                    const float* __restrict__ b3, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, const int* __restrict__ hubidx,
                                                    const float* __restrict__ HubP /* per-segment partial sums of A q at the hub rows, [B][n_seg][64] */,
                                                    const int* __restrict__ hub_seg_ptr, int n_seg, int do_next) {
    __shared__ __attribute__((aligned(16))) float Wl[64 * TS];
    __shared__ __attribute__((aligned(16))) float tiles[6][16 * TS];
    float (*Dt)[16 * TS] = &tiles[0];              // Dt[0..1]: dpre_S, dpre_I
    float (*Yt)[16 * TS] = &tiles[2];              // Yt[0..1]: y_S, y_I rows of grid point i
    float (*Gt)[16 * TS] = &tiles[4];              // Gt[0..1]: g_Y = dpre W
    const PartLayout L{64};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, sub = lane & 15;
    const int i = lane & 15, kq = lane >> 4;
    load_W_to_lds<false>(W, Wl);
    const size_t slab = (size_t)rows * 64;
    const float* aSp = a; const float* aIp = a + slab; const float* aRp = a + 2 * slab;
    const float* YIs = Ysol + slab;
    const float* YpI = Yprev + slab; const float* YpR = Yprev + 2 * slab;
    const int lr = w * 4 + g, ro = lr * TS + 4 * sub;
    const unsigned lane_b = 16u * sub;
    constexpr bool NT = true;                      // streamed rows are non-temporal: the L2 is for the q table the gather re-reads
    constexpr bool head = HEAD;                    // compile-time: the accumulators of the head's parameter gradients exist only here
    HeadAcc hacc;
#pragma unroll
    for (int k = 0; k < 4; ++k) { hacc.dw3[k] = zero4(); hacc.db3[k] = 0.f; hacc.dw2[k] = 0.f; }
    hacc.db2 = 0.f;
    f32x4 accW[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) accW[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 accb = zero4();                         // this lane group's share of gb (4 features of its rows)
    // XCD-affine tile queues, as k_bwd_fused64; everything about a row is 32-bit (the launcher bounds rows < 2^24)
    const int xq = (gridDim.x % 8 == 0 && total_tiles >= 32) ? 8 : 1;
    const int q_lo = (int)(total_tiles * (blockIdx.x % xq) / xq), q_hi = (int)(total_tiles * (blockIdx.x % xq + 1) / xq);
    const int t_stride = gridDim.x / xq;
    // The tile loop is software-pipelined one tile deep for everything that has a dependent round trip or feeds the
    // matrix phase: a row's header (extent + first 16 neighbour ids, one load) is fetched a tile ahead; its first 8
    // neighbour rows of the q table and its y_S, y_I rows are requested right before the PREVIOUS tile's matrix phase
    // and land under it; the rows only the epilogue needs (Z_S(y_{i-1}), y_{i-1}, the output cotangents) are requested
    // there too.  What is left at the top of a tile is one round trip: a, the kept Z_S, Z_I, A Z_I and neighbours 8..15.
    // Loads inside a tile are UNCONDITIONAL (no per-row branches: the register allocator loses the rolling gather
    // registers across them): padding rows read row 0 and are masked where it matters, absent neighbours read the q
    // table's ZERO ROW (row `rows`, kept zero by the host).
    struct Row { bool valid, hub; int start, end, cnt, hcnt; unsigned row, rowbase, mine, hoff; };
    const unsigned zoff = (unsigned)rows * 256u;
    auto head_of = [&](int t, Row& r) {
        const int bq = t / tiles_per_sample;
        const int node = (t - bq * tiles_per_sample) * 16 + lr;
        r.valid = t < q_hi && node < n;
        const int nodec = r.valid ? node : 0;
        r.rowbase = r.valid ? (unsigned)bq * (unsigned)n : 0u;      // first row of the tile's sample
        r.row = r.rowbase + (unsigned)nodec;
        const int* h = rowhdr + (size_t)nodec * 20;
        r.start = h[0]; r.end = r.valid ? h[1] : r.start;
        const int c0 = h[4 + sub];
        r.hub = false; r.hoff = 0u; r.hcnt = 0;
        if (HUBS) {
            const int hb = hubidx[nodec];
            r.hub = r.valid && hb >= 0;
            // a hub row's sum arrives as per-segment partials (k_hub_seg): rows [hoff, hoff + hcnt) of HubP
            const int s0 = r.hub ? hub_seg_ptr[hb] : 0, s1 = r.hub ? hub_seg_ptr[hb + 1] : 0;
            r.hoff = (unsigned)(bq * n_seg + s0) * 256u;
            r.hcnt = s1 - s0;
        }
        r.cnt = r.hub ? 0 : r.end - r.start;
        r.mine = (sub < r.cnt) ? (r.rowbase + (unsigned)c0) * 256u : zoff;
    };
    float4 v0, v1, v2, v3, v4, v5, v6, v7;
#define GN_PF(K, V) V = ld4o(Qc, (unsigned)row_bcast<(K) & 15>((int)m_) + lane_b);
#define GN_ISSUE8(M, K0) { const unsigned m_ = M; GN_PF(K0, v0) GN_PF(K0 + 1, v1) GN_PF(K0 + 2, v2) GN_PF(K0 + 3, v3) \
                                                  GN_PF(K0 + 4, v4) GN_PF(K0 + 5, v5) GN_PF(K0 + 6, v6) GN_PF(K0 + 7, v7) }
#define GN_AC1(U) gq.x += U.x; gq.y += U.y; gq.z += U.z; gq.w += U.w;
#define GN_ACC8 GN_AC1(v0) GN_AC1(v1) GN_AC1(v2) GN_AC1(v3) GN_AC1(v4) GN_AC1(v5) GN_AC1(v6) GN_AC1(v7)
    Row cur, nxt;
    int t = q_lo + blockIdx.x / xq;
    head_of(t, cur);
    {
        const float4 ys = ld4so<NT>(Ysol, cur.row * 256u + lane_b), yi = ld4so<NT>(YIs, cur.row * 256u + lane_b);
        GN_ISSUE8(cur.mine, 0)
        *reinterpret_cast<float4*>(&Yt[0][ro]) = ys; *reinterpret_cast<float4*>(&Yt[1][ro]) = yi;
    }
    for (; t < q_hi; t += t_stride) {
        head_of(t + t_stride, nxt);                    // next tile's header: requested before this tile's own rows
        const bool valid = cur.valid;
        const unsigned off = cur.row * 256u + lane_b;
        const float bt = valid ? beta[cur.row] : 0.f, gm = gamma[cur.row];
        float4 aS = ld4so<NT>(aSp, off), aI = ld4so<NT>(aIp, off), aR = ld4so<NT>(aRp, off);
        const float4 ps = ld4so<NT>(PSk, off), zi = ld4so<NT>(ZIk, off);
        float4 gq = zero4();
        if (HUBS) {                                    // hub rows (their gather reads zero rows)
            // segment partials added in segment order, 8 in flight: what a separate reduction launch used to do
            const int hc = cur.hcnt;
            if (__any(hc > 0)) {
                float4 hs = zero4();
                for (int sg = 0; __any(sg < hc); sg += 8) {
#define GN_HP(Q, U) float4 U = zero4(); if (sg + (Q) < hc) U = ld4o(HubP, cur.hoff + (unsigned)(sg + (Q)) * 256u + lane_b);
                    GN_HP(0, u0) GN_HP(1, u1) GN_HP(2, u2) GN_HP(3, u3) GN_HP(4, u4) GN_HP(5, u5) GN_HP(6, u6) GN_HP(7, u7)
#undef GN_HP
                    hs.x += u0.x; hs.y += u0.y; hs.z += u0.z; hs.w += u0.w;  hs.x += u1.x; hs.y += u1.y; hs.z += u1.z; hs.w += u1.w;
                    hs.x += u2.x; hs.y += u2.y; hs.z += u2.z; hs.w += u2.w;  hs.x += u3.x; hs.y += u3.y; hs.z += u3.z; hs.w += u3.w;
                    hs.x += u4.x; hs.y += u4.y; hs.z += u4.z; hs.w += u4.w;  hs.x += u5.x; hs.y += u5.y; hs.z += u5.z; hs.w += u5.w;
                    hs.x += u6.x; hs.y += u6.y; hs.z += u6.z; hs.w += u6.w;  hs.x += u7.x; hs.y += u7.y; hs.z += u7.z; hs.w += u7.w;
                }
                if (cur.hub) gq = hs;
            }
        }
        // finish the gather: neighbours 0..7 are in flight since the previous tile; ascending column order throughout
        GN_ACC8
        if (__any(cur.cnt > 8)) { GN_ISSUE8(cur.mine, 8) GN_ACC8 }
        if (__any(cur.cnt > 16)) {
            for (int e0 = cur.start + 16; __any(e0 < cur.end && !cur.hub); e0 += 16) {
                const int c2 = cur.hub ? 0 : cur.end - e0;
                const unsigned m2 = (sub < c2) ? (cur.rowbase + (unsigned)col[e0 + sub]) * 256u : zoff;
                GN_ISSUE8(m2, 0) GN_ACC8
                if (__any(c2 > 8)) { GN_ISSUE8(m2, 8) GN_ACC8 }
            }
        }
        {
            float4 dS, dI;
#define GN_DP(c)                                                               \
            {                                                                  \
                const float v = bt * (aI.c - aS.c);                            \
                dS.c = v * ps.c;                                               \
                dI.c = valid ? (gq.c + gm * (aR.c - aI.c)) * (zi.c * (1.0f - zi.c)) : 0.f;   \
            }
            GN_DP(x) GN_DP(y) GN_DP(z) GN_DP(w)
#undef GN_DP
            accb.x += dS.x + dI.x; accb.y += dS.y + dI.y; accb.z += dS.z + dI.z; accb.w += dS.w + dI.w;
            *reinterpret_cast<float4*>(&Dt[0][ro]) = dS; *reinterpret_cast<float4*>(&Dt[1][ro]) = dI;
        }
        __syncthreads();
        // requests that travel under the matrix phase: this tile's epilogue rows, then the next tile's y rows and first
        // 8 neighbour rows
        float4 zsp = zero4();
        float4 y[3] = {zero4(), zero4(), zero4()};
        float gout[3] = {0.f, 0.f, 0.f};
        if (do_next) zsp = ld4so<NT>(ZSp, off);
        if (head) {
            y[0] = ld4so<NT>(Yprev, off); y[1] = ld4so<NT>(YpI, off); y[2] = ld4so<NT>(YpR, off);
            gout[0] = valid ? gS[cur.row] : 0.f; gout[1] = valid ? gI[cur.row] : 0.f; gout[2] = valid ? gR[cur.row] : 0.f;
        }
        const float4 ysn = ld4so<NT>(Ysol, nxt.row * 256u + lane_b), yin = ld4so<NT>(YIs, nxt.row * 256u + lane_b);
        GN_ISSUE8(nxt.mine, 0)
        // gW += dpre^T y (contraction over the tile's 16 rows), then g_Y = dpre W: one matrix phase
#pragma unroll
        for (int X = 0; X < 2; ++X) {
#pragma unroll
            for (int s8 = 0; s8 < 4; ++s8) {
                const int rr = 4 * s8 + kq;
                const float av = Dt[X][rr * TS + 16 * w + i];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    accW[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Yt[X][rr * TS + 16 * kt + i], accW[kt], 0, 0, 0);
            }
        }
        mfma_tile16<false, true>(Dt[0], Wl, Gt[0], 0.f, w, lane);
        mfma_tile16<false, true>(Dt[1], Wl, Gt[1], 0.f, w, lane);
        __syncthreads();
        // a += dt g_Y; the head's VJP at grid point i-1; the next interval's q row; stage the next tile's y rows.  (No
        // barrier before the next tile's Dt writes: Dt / Yt were last read before the barrier above, Gt is rewritten only
        // after the next one.)
        {
            const float4 uS = *reinterpret_cast<const float4*>(&Gt[0][ro]);
            const float4 uI = *reinterpret_cast<const float4*>(&Gt[1][ro]);
            aS.x += dt * uS.x; aS.y += dt * uS.y; aS.z += dt * uS.z; aS.w += dt * uS.w;
            aI.x += dt * uI.x; aI.y += dt * uI.y; aI.z += dt * uI.z; aI.w += dt * uI.w;
        }
        if (head) {
            float4 w3v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w3v[k] = ld4g(w3 + k * 64 + 4 * sub);      // L1-resident
            head_vjp64(y, gout, w3v, b3, w2, b2, aS, aI, aR, hacc);                 // padding rows: gout = 0 adds nothing
            if (valid) st4so<NT>(a + 2 * slab, off, aR);
        }
        if (valid) {
            st4so<NT>(a, off, aS); st4so<NT>(a + slab, off, aI);
            if (do_next)
                st4so<NT>(Qn, off, make_float4(bt * (aI.x - aS.x) * zsp.x, bt * (aI.y - aS.y) * zsp.y,
                                          bt * (aI.z - aS.z) * zsp.z, bt * (aI.w - aS.w) * zsp.w));
        }
        *reinterpret_cast<float4*>(&Yt[0][ro]) = ysn; *reinterpret_cast<float4*>(&Yt[1][ro]) = yin;
        cur = nxt;
    }
#undef GN_ACC8
#undef GN_AC1
#undef GN_ISSUE8
#undef GN_PF
    float* part = part_all + (size_t)blockIdx.x * L.total();
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            part[L.oW() + (16 * w + 4 * kq + reg) * 64 + 16 * kt + i] += dt * accW[kt][reg];
    // lane-group partials (gb, and the head's parameter gradients) -> this workgroup's slot, fixed order
    __syncthreads();
    constexpr int NE = 5 * 64 + 12;                   // 4*64 + 9 head values, then the 64 gb values; rows kept 16-B aligned
    float* red = &tiles[0][0];                        // 16 groups x 332 floats = 21 248 B <= the six 16-row tiles (26 112 B)
    float* mine = red + (size_t)(threadIdx.x >> 4) * NE;
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(mine + k * 64 + 4 * sub) = hacc.dw3[k];
    if (sub == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { mine[256 + k] = hacc.db3[k]; mine[260 + k] = hacc.dw2[k]; }
        mine[264] = hacc.db2;
    }
    *reinterpret_cast<float4*>(mine + 268 + 4 * sub) = accb;
    __syncthreads();
    for (int e = threadIdx.x; e < 268 + 64; e += 256) {
        if (e >= 265 && e < 268) continue;
        if (e < 265 && !head) continue;
        float s = 0.f;
        for (int gi = 0; gi < 16; ++gi) s += red[(size_t)gi * NE + e];
        if (e < 265) part[L.ow3() + e] += s;
        else part[L.ob() + (e - 268)] += dt * s;
    }
}

// --------------------------------------------------------------------------- ONE launch per backward interval, H <= 32
// The generic-H twin of k_bwd_fused64 for the small hidden sizes the multi-graph launcher uses (H = 8,
// monitorer-ngraphs.py:20): a lane group of LPR = H/4 lanes owns a row; both mat-vecs (g_Y = dpre W, Z = sigmoid(W y + b))
// broadcast the row inside the group by shuffles against W / W^T staged in LDS, gW += dpre^T y goes through LDS row
// tiles as in k_bwd_mlp, and the head's VJP at grid point i-1 plus the next interval's Z / q tables are folded in.
template <int LPR>
__device__ __forceinline__ float4 group_lin(float4 x, const float* __restrict__ M, int sub, bool active, int H) {
    float4 acc = z4();
    const float xv[4] = {x.x, x.y, x.z, x.w};
    for (int kk = 0; 4 * kk < H; ++kk) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float xk = __shfl(xv[c], kk, LPR);
            if (active) {
                const float4 w = ld4b(M + (size_t)(4 * kk + c) * H + 4 * sub);
                acc.x = fmaf(xk, w.x, acc.x); acc.y = fmaf(xk, w.y, acc.y);
                acc.z = fmaf(xk, w.z, acc.z); acc.w = fmaf(xk, w.w, acc.w);
            }
        }
    }
    return acc;
}
__device__ __forceinline__ float sig_b(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <int LPR>
__global__ __launch_bounds__(256) void k_bwd_fused_generic(
    const int* __restrict__ rowptr, const int* __restrict__ col, int n, long rows, int H, float* __restrict__ ZS,
    const float* __restrict__ ZIc, const float* __restrict__ Qc, float* __restrict__ ZIn, float* __restrict__ Qn,
    const float* __restrict__ Ysol, const float* __restrict__ Yprev, const float* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ beta, const float* __restrict__ gamma, float dt,
    float* __restrict__ a, float* __restrict__ part_all, const float* __restrict__ gS, const float* __restrict__ gI,
    const float* __restrict__ gR, const float* __restrict__ w3, const float* __restrict__ b3, const float* __restrict__ w2,
    const float* __restrict__ b2, const int* __restrict__ hubidx,
    const float* __restrict__ HubP0 /* hub rows: per-segment partial sums of A Z_I, [B][n_seg][H] */,
    const float* __restrict__ HubP1 /* ... of A q */, const int* __restrict__ hub_seg_ptr, int n_seg, int do_next) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int G = 256 / LPR;
    const PartLayout L{H};
    float* Wl = lds;                           // W[j][k]
    float* Wt = Wl + (size_t)H * H;            // W^T
    float* Dt = Wt + (size_t)H * H;            // [2][G][H] dpre tile (S, I)
    float* Yt = Dt + (size_t)2 * G * H;        // [2][G][H] y tile   (S, I)
    for (int idx = threadIdx.x; idx < H * H; idx += 256) {
        const float v = W[idx];
        Wl[idx] = v;
        Wt[(size_t)(idx % H) * H + idx / H] = v;
    }
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const bool lane_ok = 4 * sub < H;
    const size_t slab = (size_t)rows * H;
    const int nE = H * H;
    constexpr int MAXM = (LPR * LPR / 16) < 1 ? 1 : (LPR * LPR / 16);
    float accW[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) accW[m] = 0.f;
    float accb = 0.f;
    const int M = (nE + 255) / 256;
    const bool head = gS != nullptr;
    const float4 bias4 = lane_ok ? ld4b(bias + 4 * sub) : z4();
    float4 w3v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w3v[k] = lane_ok ? ld4b(w3 + (size_t)k * H + 4 * sub) : z4();
    HeadAcc hacc;
#pragma unroll
    for (int k = 0; k < 4; ++k) { hacc.dw3[k] = z4(); hacc.db3[k] = 0.f; hacc.dw2[k] = 0.f; }
    hacc.db2 = 0.f;
    for (long r0 = (long)blockIdx.x * G; r0 < rows; r0 += (long)gridDim.x * G) {
        const long r = r0 + grp;
        const bool inrow = r < rows, ok = lane_ok && inrow;
        const long b = inrow ? r / n : 0;
        const int node = (int)(r - b * n);
        const size_t off = (size_t)r * H + 4 * sub;
        // 1. own-row loads first (they travel under the gathers' dependent round trips), then both gathers (no
        //    cross-lane traffic inside, so the tail rows may skip them)
        float4 aS = z4(), aI = z4(), aR = z4(), dS = z4(), dI = z4(), yS = z4(), yI = z4(), zs0 = z4(), zi0 = z4();
        float bt = 0.f, gm = 0.f;
        if (ok) {
            bt = beta[r]; gm = gamma[r];
            aS = ld4b(a + off); aI = ld4b(a + slab + off); aR = ld4b(a + 2 * slab + off);
            zs0 = ld4b(ZS + off); zi0 = ld4b(ZIc + off);
            yS = ld4b(Ysol + off); yI = ld4b(Ysol + slab + off);
        }
        float4 ai = z4(), gq = z4();
        if (inrow) {
            const int hub = hubidx ? hubidx[node] : -1;
            if (hub >= 0) {
                // the hub's segment partials, added in segment order (no separate reduction launch), 4 per table in flight
                const size_t pb = (size_t)b * n_seg * H + 4 * sub;
                const int s1 = hub_seg_ptr[hub + 1];
                for (int sg = hub_seg_ptr[hub]; sg < s1; sg += 4) {
                    float4 u[4], v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool on = lane_ok && sg + q < s1;
                        u[q] = on ? ld4b(HubP0 + pb + (size_t)(sg + q) * H) : z4();
                        v[q] = on ? ld4b(HubP1 + pb + (size_t)(sg + q) * H) : z4();
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        ai.x += u[q].x; ai.y += u[q].y; ai.z += u[q].z; ai.w += u[q].w;
                        gq.x += v[q].x; gq.y += v[q].y; gq.z += v[q].z; gq.w += v[q].w;
                    }
                }
            } else {
                gn_gather2<(LPR <= 4 ? 8 : 4)>(col, rowptr[node], rowptr[node + 1], ZIc + (size_t)b * n * H, Qc + (size_t)b * n * H, H, sub, lane_ok, ai, gq);
            }
        }
        if (ok) {
#define GN_DPRE(c)                                                         \
            {                                                              \
                const float v = bt * (aI.c - aS.c);                        \
                dS.c = (v * ai.c) * (zs0.c * (1.0f - zs0.c));              \
                dI.c = (gq.c + gm * (aR.c - aI.c)) * (zi0.c * (1.0f - zi0.c)); \
            }
            GN_DPRE(x) GN_DPRE(y) GN_DPRE(z) GN_DPRE(w)
#undef GN_DPRE
        }
        __syncthreads();                       // previous tile's gW pass is done with the tiles (also covers the W stage)
        if (lane_ok) {
            st4b(Dt + (size_t)grp * H + 4 * sub, dS);       st4b(Dt + ((size_t)G + grp) * H + 4 * sub, dI);
            st4b(Yt + (size_t)grp * H + 4 * sub, yS);       st4b(Yt + ((size_t)G + grp) * H + 4 * sub, yI);
        }
        // 2. g_Y = dpre W (row broadcast inside the lane group), a += dt g_Y
        const float4 uS = group_lin<LPR>(dS, Wl, sub, lane_ok, H), uI = group_lin<LPR>(dI, Wl, sub, lane_ok, H);
        aS.x += dt * uS.x; aS.y += dt * uS.y; aS.z += dt * uS.z; aS.w += dt * uS.w;
        aI.x += dt * uI.x; aI.y += dt * uI.y; aI.z += dt * uI.z; aI.w += dt * uI.w;
        // 3. dL/dsol[i-1] through the head (rows beyond the end carry y = 0, gout = 0)
        float4 y[3] = {z4(), z4(), z4()};
        float gout[3] = {0.f, 0.f, 0.f};
        if (ok && (head || do_next)) { y[0] = ld4b(Yprev + off); y[1] = ld4b(Yprev + slab + off); }
        if (head) {
            if (ok) y[2] = ld4b(Yprev + 2 * slab + off);
            if (inrow) { gout[0] = gS[r]; gout[1] = gI[r]; gout[2] = gR[r]; }
            head_vjp64<LPR>(y, gout, w3v, b3, w2, b2, aS, aI, aR, hacc);
            if (ok) st4b(a + 2 * slab + off, aR);
        }
        if (ok) { st4b(a + off, aS); st4b(a + slab + off, aI); }
        // 4. Z(y_{i-1}) and q for the next interval
        if (do_next) {
            float4 zs = group_lin<LPR>(y[0], Wt, sub, lane_ok, H), zi = group_lin<LPR>(y[1], Wt, sub, lane_ok, H);
            zs = make_float4(sig_b(zs.x + bias4.x), sig_b(zs.y + bias4.y), sig_b(zs.z + bias4.z), sig_b(zs.w + bias4.w));
            zi = make_float4(sig_b(zi.x + bias4.x), sig_b(zi.y + bias4.y), sig_b(zi.z + bias4.z), sig_b(zi.w + bias4.w));
            if (ok) {
                st4b(ZS + off, zs); st4b(ZIn + off, zi);
                st4b(Qn + off, make_float4(bt * (aI.x - aS.x) * zs.x, bt * (aI.y - aS.y) * zs.y, bt * (aI.z - aS.z) * zs.z,
                                           bt * (aI.w - aS.w) * zs.w));
            }
        }
        __syncthreads();                       // tiles complete
        // 5. gW[j][k] += sum_rows dpre[r][j] * y[r][k]   (thread owns entries e = tid + 256 m), gb += column sums
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            if (m < M) {
                const int e = threadIdx.x + 256 * m;
                if (e < nE) {
                    const int j = e / H, k = e % H;
                    float s = 0.f;
                    for (int rr = 0; rr < 2 * G; ++rr) s = fmaf(Dt[(size_t)rr * H + j], Yt[(size_t)rr * H + k], s);
                    accW[m] += s;
                }
            }
        }
        if (threadIdx.x < H) {
            float s = 0.f;
            for (int rr = 0; rr < 2 * G; ++rr) s += Dt[(size_t)rr * H + threadIdx.x];
            accb += s;
        }
    }
    float* part = part_all + (size_t)blockIdx.x * L.total();
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        if (m < M) {
            const int e = threadIdx.x + 256 * m;
            if (e < nE) part[L.oW() + e] += dt * accW[m];
        }
    }
    if (threadIdx.x < H) part[L.ob() + threadIdx.x] += dt * accb;
    if (head) {
        __syncthreads();
        const int ne = 4 * H + 12;                 // 4H + 9 used; rows stay 16-B aligned
        float* mine = lds + (size_t)grp * ne;
        if (lane_ok)
#pragma unroll
            for (int k = 0; k < 4; ++k) st4b(mine + k * H + 4 * sub, hacc.dw3[k]);
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { mine[4 * H + k] = hacc.db3[k]; mine[4 * H + 4 + k] = hacc.dw2[k]; }
            mine[4 * H + 8] = hacc.db2;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 4 * H + 9; e += 256) {
            float s = 0.f;
            for (int gi = 0; gi < G; ++gi) s += lds[(size_t)gi * ne + e];
            part[L.ow3() + e] += s;
        }
    }
}

// --------------------------------------------------------------------------- encoder backward
template <int LPR>
__global__ __launch_bounds__(256) void k_enc_bwd(const float* __restrict__ a, const float* __restrict__ sol0,
                                                 const float* __restrict__ x, long rows, int H,
                                                 float* __restrict__ part_all) {
    extern __shared__ float red[];                       // [G][2H]
    constexpr int G = 256 / LPR;
    const PartLayout L{H};
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const bool active = 4 * sub < H;
    const size_t slab = (size_t)rows * H;
    float4 dw = z4(), db = z4();
    for (long r = (long)blockIdx.x * G + grp; r < rows; r += (long)gridDim.x * G) {
        if (!active) continue;
        const size_t off = (size_t)r * H + 4 * sub;
#pragma unroll
        for (int X = 0; X < 3; ++X) {
            const float s = x[(size_t)r * (3 + H) + X];
            const float4 y = ld4b(sol0 + X * slab + off), av = ld4b(a + X * slab + off);
            const float4 mk = make_float4(y.x > 0.f ? av.x : 0.f, y.y > 0.f ? av.y : 0.f, y.z > 0.f ? av.z : 0.f,
                                          y.w > 0.f ? av.w : 0.f);
            dw.x = fmaf(mk.x, s, dw.x); dw.y = fmaf(mk.y, s, dw.y); dw.z = fmaf(mk.z, s, dw.z); dw.w = fmaf(mk.w, s, dw.w);
            db.x += mk.x; db.y += mk.y; db.z += mk.z; db.w += mk.w;
        }
    }
    float* mine = red + (size_t)grp * 2 * H;
    if (active) { st4b(mine + 4 * sub, dw); st4b(mine + H + 4 * sub, db); }
    __syncthreads();
    flush_groups(red, G, 2 * H, part_all + (size_t)blockIdx.x * L.total() + L.ow1());
}

// Sum the workgroup slots in order and write each parameter's gradient straight to its destination.
struct GradDst { float* dst[8]; int off[9]; };
// 16 elements x 16 slot slices per workgroup: slice s sums slots s, s+16, ... in order, then the slices are summed
// in order through LDS -- a fixed association, so still bitwise reproducible, without 768 serial loads per thread.
__global__ __launch_bounds__(256) void k_reduce_parts(const float* __restrict__ part_all, int nwg, int total, GradDst gd) {
    __shared__ float red[16][17];
    const int el = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    float s = 0.f;
    if (e < total)
        for (int w = slice; w < nwg; w += 16) s += part_all[(size_t)w * total + e];
    red[slice][el] = s;
    __syncthreads();
    if (slice != 0 || e >= total) return;
    s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += red[q][el];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (e >= gd.off[k] && e < gd.off[k + 1]) gd.dst[k][e - gd.off[k]] = s;
}

// --------------------------------------------------------------------------- host
int gn_launch_mlp_any(const gnode_graph_s* g, const float* X, const float* W, const float* b, float* Z, long nrows, int H,
                      hipStream_t st);

static int lpr_of(int H) {
    int need = H / 4, l = 1;
    while (l < need) l <<= 1;
    return l;
}

#define BWD_DISPATCH(lpr, ...)                                   \
    switch (lpr) {                                               \
        case 1: { constexpr int LPR = 1; __VA_ARGS__; } break;   \
        case 2: { constexpr int LPR = 2; __VA_ARGS__; } break;   \
        case 4: { constexpr int LPR = 4; __VA_ARGS__; } break;   \
        case 8: { constexpr int LPR = 8; __VA_ARGS__; } break;   \
        case 16: { constexpr int LPR = 16; __VA_ARGS__; } break; \
        default: { constexpr int LPR = 32; __VA_ARGS__; } break; \
    }

__global__ void k_extract_bg(const float* __restrict__ bgslab, long rows, int H, float* __restrict__ beta,
                             float* __restrict__ gamma) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    beta[r] = bgslab[(size_t)r * H];
    gamma[r] = bgslab[(size_t)r * H + 1];
}

// one workspace slab: rows x H floats plus ONE MORE ROW (the q tables' zero row, which k_bwd_kept64's gather reads for absent
// neighbours), rounded up
static size_t bwd_slab_bytes(int64_t rows, int32_t H) { return gn_align(((size_t)rows + 1) * H * sizeof(float)); }

static size_t backward_fixed_bytes(int64_t rows, int32_t H) {
    const PartLayout L{H};
    const size_t slab = bwd_slab_bytes(rows, H);
    // a[3], Z[2], q[1], dpre[2] slabs + beta, gamma + partial buffer + reduced gradient vector
    // (+ the control block of the persistent sweep, gnode_pers64_bwd.hip)
    return 8 * slab + 2 * gn_align((size_t)rows * sizeof(float)) +
           gn_align((size_t)BWD_NWG * L.total() * sizeof(float)) + gn_align((size_t)L.total() * sizeof(float)) +
           gn_pers64_ctl_bytes();
}

extern "C" size_t gnode_backward_workspace_bytes(gnode_graph_t g, int64_t rows, int32_t H) {
    if (!g || rows <= 0 || H <= 0) return 0;
    return backward_fixed_bytes(rows, H) + gn_hub_scratch_bytes(g, rows / g->n, H, 2);     // two tables per hub pass
}

// dynamic LDS above 64 KB (the five-launch generic path at H > 100) needs the attribute once per device
int gn_bwd_set_attributes() {
    GN_HIP(hipFuncSetAttribute((const void*)k_bwd_mlp<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

extern "C" int gnode_backward_f32(gnode_graph_t g, const float* x, const gnode_params* p, const float* dt_host,
                                  int32_t n_steps, const int32_t* out_rows_host, int32_t n_out, const float* sol,
                                  const float* keep, size_t keep_bytes, const float* gS, const float* gI, const float* gR,
                                  const gnode_params* grads, int64_t rows, int32_t H, void* workspace,
                                  size_t workspace_bytes, void* stream, int32_t flags, int32_t sol_info) {
    GN_CHECK_ARG(g && x && p && sol && gS && gI && gR && grads && workspace, "gnode_backward_f32: null pointer");
    GN_CHECK_ARG(n_steps >= 0 && (n_steps == 0 || dt_host), "gnode_backward_f32: bad n_steps/dt");
    GN_CHECK_ARG(H >= 4 && H <= 128 && H % 4 == 0, "gnode_backward_f32: need 4 <= H <= 128, H %% 4 == 0 (got %d)", H);
    GN_CHECK_ARG(rows > 0 && rows % g->n == 0, "gnode_backward_f32: rows=%lld is not a multiple of graph n=%d",
                 (long long)rows, g->n);
    GN_CHECK_ARG(grads->odefunc_linear_weight && grads->odefunc_linear_bias && grads->linearS1_weight &&
                     grads->linearS1_bias && grads->linear3_weight && grads->linear3_bias && grads->linearS2_weight &&
                     grads->linearS2_bias, "gnode_backward_f32: null gradient pointer");
    if (workspace_bytes < gnode_backward_workspace_bytes(g, rows, H)) {
        gnode_set_error("gnode_backward_f32: workspace %zu < %zu", workspace_bytes, gnode_backward_workspace_bytes(g, rows, H));
        return GNODE_ERR_WORKSPACE;
    }
    const int G = n_steps + 1;
    if (out_rows_host)
        for (int i = 0; i < n_out; ++i)
            GN_CHECK_ARG(out_rows_host[i] >= 0 && out_rows_host[i] < G && (i == 0 || out_rows_host[i] > out_rows_host[i - 1]),
                         "gnode_backward_f32: out_rows must be ascending grid indices in [0,%d)", G);
    hipStream_t st = (hipStream_t)stream;
    const PartLayout L{H};
    const size_t slab = (size_t)rows * H, slab_b = bwd_slab_bytes(rows, H);
    const size_t vec_b = gn_align((size_t)rows * sizeof(float));
    char* ws = (char*)workspace;
    float* a = (float*)ws;                               // 3 slabs (element-contiguous inside 3 aligned slabs)
    float* Z = (float*)(ws + 3 * slab_b);                // 2 slabs
    float* q = (float*)(ws + 5 * slab_b);                // 1 slab
    float* dpre = (float*)(ws + 6 * slab_b);             // 2 slabs
    float* beta = (float*)(ws + 8 * slab_b);
    float* gamma = (float*)(ws + 8 * slab_b + vec_b);
    float* part = (float*)(ws + 8 * slab_b + 2 * vec_b);
    float* red = (float*)(ws + 8 * slab_b + 2 * vec_b + gn_align((size_t)BWD_NWG * L.total() * sizeof(float)));
    void* hub_scratch = ws + backward_fixed_bytes(rows, H);
    int slots_used = 1;                                  // highest workgroup slot any launch wrote, for the final reduction
    // what the forward that produced `sol` / `keep` was (its sol_info says so; unchecked callers: the same question, same flags)
    const int n_emit = out_rows_host ? n_out : n_steps + 1;
    const bool fwd_tiny = sol_info >= 0 ? (sol_info & GNODE_SOL_TINY) != 0 : gn_forward_kind(g, rows, H, 0, n_steps, n_emit, true, flags, nullptr) == 1;
    const bool tiny = fwd_tiny && gn_tiny_bwd64_ok(g, rows, H, n_steps);
    if (tiny) {
        // graphs that fit one workgroup: the whole sweep is one launch writing slot b for sample b (gnode_bwd_tiny.hip)
        const size_t keep_need = gnode_forward_keep_bytes(g, rows, H, n_steps, out_rows_host ? n_out : n_steps + 1);
        if (keep && keep_need > 0 && keep_bytes < keep_need) {
            gnode_set_error("gnode_backward_f32: keep buffer %zu < %zu", keep_bytes, keep_need);
            return GNODE_ERR_WORKSPACE;
        }
        if (int e = gn_launch_tiny_bwd64(g, rows, x, p, dt_host, n_steps, out_rows_host, n_out, sol, gS, gI, gR, part,
                                         keep_need > 0 ? keep : nullptr, st))
            return e;
        slots_used = (int)(rows / g->n);
        {   // (gnode_backward_status reads a defined word after this path too)
            PersCtl* ctl = (PersCtl*)(ws + backward_fixed_bytes(rows, H) - gn_pers64_ctl_bytes());
            if (int e = gn_zero_async(ctl->error, sizeof(ctl->error), st)) return e;
        }
    } else {
    // every start-up zero-fill in ONE launch: the adjoint state, the gradient slots, (H = 64) the q tables' zero rows and the
    // control block of the persistent sweeps (they then skip their own zero-fill launch; a call that runs none leaves the
    // give-up word at 0, which is what gnode_backward_status reads)
    const bool has_ctl = H == 64 || H <= 32;
    char* ctl_ptr = ws + backward_fixed_bytes(rows, H) - gn_pers64_ctl_bytes();
    {
        GnZeroRegions zr;
        zr.n = 0;
        auto add = [&](void* ptr, size_t bytes) { zr.p[zr.n] = ptr; zr.bytes[zr.n] = bytes; ++zr.n; };
        add(a, 3 * slab * sizeof(float));
        add(part, (size_t)BWD_NWG * L.total() * sizeof(float));
        if (H == 64 && n_steps >= 1) {
            add(q + slab, (size_t)H * sizeof(float));
            add(dpre + slab_b / sizeof(float) + slab, (size_t)H * sizeof(float));
        }
        if (has_ctl) add(ctl_ptr, gn_pers64_ctl_bytes());
        if (int e = gn_zero_regions_async(zr, st)) return e;
    }
    // small hidden sizes, batches that fit one resident grid: the whole sweep INCLUDING its start-up is one persistent launch
    PersgPlan gp;
    const bool gpersist = H <= 32 && n_steps >= 1 && !(flags & GNODE_FWD_PER_STEP) && gn_persg_plan(g, rows, H, n_steps, &gp);
    if (!gpersist) {
    hipLaunchKernelGGL(k_extract_bg, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, sol + 3 * slab, (long)rows, H,
                       beta, gamma);
    GN_LAUNCH_CHECK();
    }

    const int lpr = lpr_of(H), rpw = 256 / lpr;
    auto slot_of = [&](int gi) -> int {
        if (!out_rows_host) return gi;
        for (int i = 0; i < n_out; ++i) if (out_rows_host[i] == gi) return i;
        return -1;
    };
    auto head = [&](int gi) -> int {
        const int s = slot_of(gi);
        if (s < 0) return 0;
        const size_t lds = (size_t)rpw * (4 * H + 9) * sizeof(float);
        const int hgrid = (int)std::min<long>(BWD_NWG, std::max<long>(1, (rows + rpw - 1) / rpw));
        slots_used = std::max(slots_used, hgrid);
        BWD_DISPATCH(lpr, hipLaunchKernelGGL(k_head_bwd<LPR>, dim3(hgrid), dim3(256), lds, st, sol + (size_t)gi * 4 * slab,
                                             (long)rows, H, gS + (size_t)s * rows, gI + (size_t)s * rows, gR + (size_t)s * rows,
                                             p->linear3_weight, p->linear3_bias, p->linearS2_weight, p->linearS2_bias, a, part));
        GN_LAUNCH_CHECK();
        return 0;
    };
    if (!gpersist)
        if (int e = head(G - 1)) return e;
    const size_t mlp_lds = ((size_t)H * H + (size_t)4 * rpw * H) * sizeof(float);
    if (H == 64 && n_steps >= 1) {
        // one launch per interval: a[3] | Z_S | Z_I(0) | q(0) | Z_I(1) | q(1)  (Z_S is row-local, the gather tables ping-pong)
        float* ZS = Z; float* ZIb[2] = {Z + slab, dpre};     // (k_mlp64_q fills Z_S | Z_I element-contiguously)
        float* Qb[2] = {q, dpre + slab_b / sizeof(float)};
        // (the q tables' zero rows, Qb[k] + slab, were zero-filled by the call's first launch)
        // 16-row tiles at 3 workgroups per CU (measured on the 75k graph, 4 samples: 32-row tiles at 3 / 2 per CU 481 / 548 us
        // per interval, the unfused three-launch form 563; 16-row tiles 451)
        // does this trajectory carry A Z_I(y_i) in its 4th slabs (gnode_forward_f32 wrote it: H = 64, not the one-launch path)?
        const bool ai_saved = !fwd_tiny;               // the 4th slabs (or the keep buffer's P_S) carry A Z_I unless the one-workgroup forward ran
        // ... and did the forward keep Z_S(y_k), Z_I(y_k) as well?  Then the intervals below the last read them back
        const size_t keep_need = gnode_forward_keep_bytes(g, rows, H, n_steps, out_rows_host ? n_out : G);
        if (sol_info >= 0) {
            // the forward said what it left in `sol` / `keep` (sol_info_host): a trajectory and a keep buffer belong together
            if (keep) GN_CHECK_ARG(sol_info & GNODE_SOL_KEEP, "gnode_backward_f32: a keep buffer was passed with a trajectory whose "
                                   "forward call filled none (sol_info %d)", sol_info);
            else if (ai_saved) GN_CHECK_ARG(sol_info & GNODE_SOL_AI, "gnode_backward_f32: this trajectory was produced WITH a keep buffer "
                                            "(its 4th slabs are unwritten): pass that buffer (sol_info %d)", sol_info);
        }
        if (!(ai_saved && keep_need > 0)) keep = nullptr;
        if (keep && keep_bytes < keep_need) {
            gnode_set_error("gnode_backward_f32: keep buffer %zu < %zu", keep_bytes, keep_need);
            return GNODE_ERR_WORKSPACE;
        }
        const int tps = (g->n + 15) / 16;
        const long total = (long)(rows / g->n) * tps;
        const int grid = (int)std::min<long>(std::min<long>((long)GN_BWD_RPG1_OCC * g->num_cu, BWD_NWG), total);
        slots_used = std::max(slots_used, grid);
        // mid-size graphs: intervals G-2 .. 1 over the kept activations in ONE persistent launch (gnode_pers64_bwd.hip)
        PersPlan pplan;
        const bool persist = keep && G >= 3 && !(flags & GNODE_FWD_PER_STEP) && gn_pers_bwd64_plan(g, rows / g->n, n_steps, &pplan);
        // The last grid point emits nothing in the reference's use (get_sir_t_nodes_torch keeps the integer times, the grid ends
        // half a step later): the adjoint is still ZERO through interval G-1 -- every product of that interval is zero -- and all
        // it leaves is the head's VJP at grid point G-2 and the first q table.  The persistent sweep does that at its start
        // (`fold`), and the two start-up launches (Z(y_{G-1}) + q, and the two-table interval kernel) do not run at all.
        const bool fold = persist && slot_of(G - 1) < 0;
        // the same on the one-launch-per-interval path over kept activations (large graphs): instead of the two-table interval kernel
        // on a zero adjoint (451 us at 75k x 4 against 282 for a kept interval) only the head's VJP at grid point G-2 and
        // q = beta (a_I - a_S) Z_S(y_{G-2}) from the kept table run
        const bool skip_last = !persist && keep && G >= 3 && slot_of(G - 1) < 0 && rows < (1L << 24) &&
                               (long)(rows / g->n) * g->n_seg < (1L << 24);
        if (skip_last) {
            if (int e = head(G - 2)) return e;
            hipLaunchKernelGGL(k_bwd_q, dim3(2048), dim3(256), 0, st, a, gn_keep_zs(keep, rows, G - 2), beta, Qb[1], (long)rows, 64);
            GN_LAUNCH_CHECK();
        }
        if (!fold && !skip_last) {
            const long mt = (2 * rows + TILE_ROWS - 1) / TILE_ROWS;
            hipLaunchKernelGGL(k_mlp64_q, dim3((unsigned)std::min<long>(mt, 1024)), dim3(256), 0, st, sol + (size_t)(G - 1) * 4 * slab,
                               p->odefunc_linear_weight, p->odefunc_linear_bias, Z, a, beta, q, (long)rows);
            GN_LAUNCH_CHECK();
        }
        for (int i = skip_last ? G - 2 : G - 1; i >= 1; --i) {
            if (persist && (i == G - 2 || fold)) {
                int slot_prev[128];
                for (int j = 1; j <= G - 1; ++j) slot_prev[j] = slot_of(j - 1);
                int pslots = 0;
                const bool sampled = gn_prof_begin(2, st);
                if (int e = gn_launch_pers_bwd64(g, pplan, rows, G, Qb[0], Qb[1], sol, keep, p->odefunc_linear_weight, beta, gamma, a, part,
                                                 gS, gI, gR, p, dt_host, slot_prev, ctl_ptr, true, fold, &pslots, st))
                    return e;
                if (sampled) gn_prof_end(2, st);
                slots_used = std::max(slots_used, pslots);
                break;
            }
            const int cur = (G - 1 - i) & 1;
            const bool two = !ai_saved || i == G - 1;          // A Z_I(y_{G-1}) was never needed by the forward
            const float *AIhub = nullptr, *GQhub = nullptr;
            const bool kept_launch = keep && !two && rows < (1L << 24) && (long)(rows / g->n) * g->n_seg < (1L << 24);   // 32-bit byte offsets of rows and hub partials
            const float* HubP = nullptr;               // kept kernel: segment partials only, it adds them up itself
            if (two) { if (int e = gn_hub_gather(g, rows / g->n, 64, ZIb[cur], Qb[cur], hub_scratch, &AIhub, &GQhub, st)) return e; }
            else if (kept_launch) { if (int e = gn_hub_segments(g, rows / g->n, 64, Qb[cur], hub_scratch, &HubP, st)) return e; }
            else if (int e = gn_hub_gather(g, rows / g->n, 64, Qb[cur], nullptr, hub_scratch, &GQhub, nullptr, st)) return e;
            const int s = slot_of(i - 1);
            const float* gSs = s >= 0 ? gS + (size_t)s * rows : nullptr;
            const bool sampled = gn_prof_begin(2, st);
            if (kept_launch) {
                // the head instance carries 25 more accumulators and the head's temporaries: two workgroups per CU there
                constexpr int HOCC = GN_BWD_KEPT_HEAD_OCC;
                const bool hubs = g->n_hub > 0;
                auto kept_kernel = gSs ? (hubs ? k_bwd_kept64<HOCC, true, true> : k_bwd_kept64<HOCC, true, false>)
                                       : (hubs ? k_bwd_kept64<GN_BWD_RPG1_OCC, false, true> : k_bwd_kept64<GN_BWD_RPG1_OCC, false, false>);
                const int kgrid = gSs ? (int)std::min<long>((long)HOCC * g->num_cu, grid) : grid;
                hipLaunchKernelGGL(kept_kernel, dim3(kgrid), dim3(256), 0, st, g->rowhdr, g->col, g->n, (long)rows,
                                   tps, total, Qb[cur], Qb[cur ^ 1], sol + (size_t)i * 4 * slab, sol + (size_t)(i - 1) * 4 * slab,
                                   gn_keep_ps(keep, rows, i), gn_keep_zi(keep, rows, i), gn_keep_zs(keep, rows, i - 1),
                                   p->odefunc_linear_weight, beta, gamma, dt_host[i - 1],
                                   a, part, gSs, s >= 0 ? gI + (size_t)s * rows : nullptr,
                                   s >= 0 ? gR + (size_t)s * rows : nullptr, p->linear3_weight, p->linear3_bias,
                                   p->linearS2_weight, p->linearS2_bias, g->hubidx, HubP, g->hub_seg_ptr, g->n_seg, i > 1 ? 1 : 0);
            } else {
            auto fused_kernel = two ? k_bwd_fused64<GN_BWD_RPG1_OCC, 1, true> : k_bwd_fused64<GN_BWD_RPG1_OCC, 1, false>;
            hipLaunchKernelGGL(fused_kernel, dim3(grid), dim3(256), 0, st, g->rowptr, g->col, g->n, (long)rows, tps, total,
                               ZIb[cur], Qb[cur], ai_saved ? nullptr : ZIb[cur ^ 1], Qb[cur ^ 1], sol + (size_t)i * 4 * slab,
                               sol + (size_t)(i - 1) * 4 * slab, p->odefunc_linear_weight, p->odefunc_linear_bias, beta, gamma,
                               dt_host[i - 1], a, part, gSs, s >= 0 ? gI + (size_t)s * rows : nullptr,
                               s >= 0 ? gR + (size_t)s * rows : nullptr, p->linear3_weight, p->linear3_bias,
                               p->linearS2_weight, p->linearS2_bias, g->hubidx, AIhub, GQhub, g->n_hub, i > 1 ? 1 : 0,
                               sol + (size_t)i * 4 * slab + 3 * slab);
            }
            if (sampled) gn_prof_end(2, st);
            GN_LAUNCH_CHECK();
        }
    } else if (H <= 32 && n_steps >= 1) {
        // small hidden sizes: the same one-launch-per-interval scheme on lane groups (k_bwd_fused_generic)
        float* ZS = Z; float* ZIb[2] = {Z + slab, dpre};
        float* Qb[2] = {q, dpre + slab};
        const float* yl = sol + (size_t)(G - 1) * 4 * slab;
        if (!gpersist) {
            if (int e = gn_launch_mlp_any(g, yl, p->odefunc_linear_weight, p->odefunc_linear_bias, Z, 2 * rows, H, st)) return e;
            hipLaunchKernelGGL(k_bwd_q, dim3(2048), dim3(256), 0, st, a, Z, beta, q, (long)rows, H);
            GN_LAUNCH_CHECK();
        }
        const size_t fl = std::max((size_t)2 * H * H + (size_t)4 * rpw * H, (size_t)rpw * (4 * H + 12));
        const int grid = (int)std::min<long>(BWD_NWG, std::max<long>(1, (rows + rpw - 1) / rpw));
        // batches that fit one resident grid: every interval in ONE persistent launch (gnode_persg.hip)
        if (gpersist) {
            int slot_prev[128];
            slot_prev[0] = slot_of(G - 1);
            for (int j = 1; j <= G - 1; ++j) slot_prev[j] = slot_of(j - 1);
            const bool sampled = gn_prof_begin(2, st);
            if (int e = gn_launch_persg_bwd(g, gp, rows, H, G, ZIb[0], ZIb[1], Qb[0], Qb[1], ZS, sol, beta, gamma, a, part, gS, gI, gR, p,
                                            dt_host, slot_prev, ctl_ptr, true, st))
                return e;
            if (sampled) gn_prof_end(2, st);
            slots_used = std::max(slots_used, gp.wgs);
        } else
        slots_used = std::max(slots_used, grid);
        for (int i = gpersist ? 0 : G - 1; i >= 1; --i) {
            const int cur = (G - 1 - i) & 1;
            const float *HubP0 = nullptr, *HubP1 = nullptr;        // segment partials; the interval kernel adds them up itself
            if (int e = gn_hub_segments2(g, rows / g->n, H, ZIb[cur], Qb[cur], hub_scratch, &HubP0, &HubP1, st)) return e;
            const int s = slot_of(i - 1);
            const bool sampled = gn_prof_begin(2, st);
            BWD_DISPATCH(lpr, hipLaunchKernelGGL(k_bwd_fused_generic<LPR>, dim3(grid), dim3(256), fl * sizeof(float), st, g->rowptr,
                                                 g->col, g->n, (long)rows, H, ZS, ZIb[cur], Qb[cur], ZIb[cur ^ 1], Qb[cur ^ 1],
                                                 sol + (size_t)i * 4 * slab, sol + (size_t)(i - 1) * 4 * slab,
                                                 p->odefunc_linear_weight, p->odefunc_linear_bias, beta, gamma, dt_host[i - 1], a,
                                                 part, s >= 0 ? gS + (size_t)s * rows : nullptr,
                                                 s >= 0 ? gI + (size_t)s * rows : nullptr, s >= 0 ? gR + (size_t)s * rows : nullptr,
                                                 p->linear3_weight, p->linear3_bias, p->linearS2_weight, p->linearS2_bias,
                                                 g->hubidx, HubP0, HubP1, g->hub_seg_ptr, g->n_seg, i > 1 ? 1 : 0));
            if (sampled) gn_prof_end(2, st);
            GN_LAUNCH_CHECK();
        }
    } else
    for (int i = G - 1; i >= 1; --i) {
        const float* yi = sol + (size_t)i * 4 * slab;
        const float dt = dt_host[i - 1];
        {
            if (int e = gn_launch_mlp_any(g, yi, p->odefunc_linear_weight, p->odefunc_linear_bias, Z, 2 * rows, H, st)) return e;
            hipLaunchKernelGGL(k_bwd_q, dim3(2048), dim3(256), 0, st, a, Z, beta, q, (long)rows, H);
            GN_LAUNCH_CHECK();
            dim3 ggrid((unsigned)((g->n + rpw - 1) / rpw), (unsigned)(rows / g->n));
            const float *AIhub = nullptr, *GQhub = nullptr;
            if (int e = gn_hub_gather(g, rows / g->n, H, Z + slab, q, hub_scratch, &AIhub, &GQhub, st)) return e;
            BWD_DISPATCH(lpr, hipLaunchKernelGGL(k_bwd_gather<LPR>, ggrid, dim3(256), 0, st, g->rowptr, g->col, g->n, (long)rows, H,
                                                 a, Z, q, beta, gamma, dpre, g->hubidx, AIhub, GQhub, g->n_hub));
            GN_LAUNCH_CHECK();
            if (H == 128) {
                if (int e = gn_launch_bwd_mlp128(g, dpre, yi, p->odefunc_linear_weight, dt, a, rows, part, &slots_used, st)) return e;
            } else
            BWD_DISPATCH(lpr, {
                slots_used = BWD_NWG;
                hipLaunchKernelGGL(k_bwd_mlp<LPR>, dim3(BWD_NWG), dim3(256), mlp_lds, st, dpre, yi, p->odefunc_linear_weight, dt,
                                   a, (long)rows, H, part);
            });
            GN_LAUNCH_CHECK();
        }
        if (int e = head(i - 1)) return e;
    }
    {
        const size_t lds = (size_t)rpw * 2 * H * sizeof(float);
        const int egrid = (int)std::min<long>(BWD_NWG, std::max<long>(1, (rows + rpw - 1) / rpw));
        slots_used = std::max(slots_used, egrid);
        BWD_DISPATCH(lpr, hipLaunchKernelGGL(k_enc_bwd<LPR>, dim3(egrid), dim3(256), lds, st, a, sol, x, (long)rows, H, part));
        GN_LAUNCH_CHECK();
    }
    }   // !tiny
    // slot layout order == PartLayout order: W, b, w3, b3, w2, b2, w1, b1
    GradDst gd;
    gd.dst[0] = (float*)grads->odefunc_linear_weight; gd.dst[1] = (float*)grads->odefunc_linear_bias;
    gd.dst[2] = (float*)grads->linear3_weight;        gd.dst[3] = (float*)grads->linear3_bias;
    gd.dst[4] = (float*)grads->linearS2_weight;       gd.dst[5] = (float*)grads->linearS2_bias;
    gd.dst[6] = (float*)grads->linearS1_weight;       gd.dst[7] = (float*)grads->linearS1_bias;
    const int offs[9] = {L.oW(), L.ob(), L.ow3(), L.ob3(), L.ow2(), L.ob2(), L.ow1(), L.ob1(), L.total()};
    for (int k = 0; k < 9; ++k) gd.off[k] = offs[k];
    hipLaunchKernelGGL(k_reduce_parts, dim3((L.total() + 15) / 16), dim3(256), 0, st, part, slots_used, L.total(), gd);
    GN_LAUNCH_CHECK();
    return 0;
}

// The persistent adjoint sweeps' give-up word (the control block at the end of the fixed part of the backward workspace).
extern "C" int gnode_backward_status(int64_t rows, int32_t H, const void* workspace, void* stream, int32_t* code_host) {
    GN_CHECK_ARG(workspace && code_host && rows > 0, "gnode_backward_status: null pointer");
    *code_host = 0;
    if (H != 64 && H > 32) return 0;
    unsigned err[2] = {0, 0};
    const PersCtl* ctl = (const PersCtl*)((const char*)workspace + backward_fixed_bytes(rows, H) - gn_pers64_ctl_bytes());
    GN_HIP(hipMemcpyAsync(err, ctl->error, sizeof(err), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GN_HIP(hipStreamSynchronize((hipStream_t)stream));
    *code_host = (int32_t)err[0];
    if (err[0]) gnode_set_error("persistent adjoint sweep: a workgroup gave up waiting for epoch %u of its group", err[1]);
    return 0;
}

// diagnostic build (GN_PERS_PROF): per-phase 100 MHz ticks of the last persistent adjoint sweep on this workspace
extern "C" int gnode_backward_phase_ticks(int64_t rows, int32_t H, const void* workspace, uint64_t* ticks8_host) {
    const PersCtl* ctl = (const PersCtl*)((const char*)workspace + backward_fixed_bytes(rows, H) - gn_pers64_ctl_bytes());
    GN_HIP(hipMemcpy(ticks8_host, ctl->prof, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}
