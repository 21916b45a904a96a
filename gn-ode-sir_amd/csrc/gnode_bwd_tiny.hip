// Adjoint backward of a batch of TINY graphs in one launch (gfx950).
//
// The reference's shipped experiment is karate (n = 34, batch size 1, monitorer-sim.py:10-20): with one launch
// per kernel the adjoint sweep of gnode_bwd.hip is ~110 dependent launches of a few microseconds of work each.
// Samples never interact (block-diagonal adjacency, ode_nn_ngraph_sim.py:68-71), so ONE workgroup owns one
// sample for the whole sweep i = G-1 .. 1:
//   * the adjoint rows a_S, a_I, a_R never leave the registers of the 16-lane group that owns the row;
//   * the two gather tables of an interval (Z_I and q = beta (a_I - a_S) Z_S) live in LDS, and the rows'
//     neighbour lists are cached in registers across intervals;
//   * Z = sigmoid(y_i W^T + b), gW += dpre^T y_i and g_Y = dpre W run on the fp32 matrix cores with W and W^T
//     both staged in LDS once;
//   * the head's VJP at every emitted grid point, the encoder's VJP at grid point 0 and all parameter-gradient
//     partials (kept in registers across intervals) are folded in; a fixed-order LDS reduction writes the
//     sample's partial slot, so the result is bitwise reproducible.
// Same recurrence as gnode_bwd.hip (torchdiffeq odeint_adjoint under method='euler'); only
// the association of the dt factor differs (dt is folded into dpre once instead of into gW / g_Y afterwards).
#include "gnode_bwd.h"
#include "gnode_mfma64.h"
#include "gnode_head64.h"
#include <cstdlib>

struct TinyBwdSched {
    float dt[128];
    short slot[129];      // output slot of grid point k, or -1
    int n_steps;
};

// two LDS tables gathered through one neighbour list, ascending column order (= gather2_row64 of gnode_bwd.hip)
__device__ __forceinline__ void gather2_row_lds(const int* __restrict__ col, const float* __restrict__ Zl,
                                                const float* __restrict__ Ql, int start, int end, int first16, int sub,
                                                float4& a0, float4& a1) {
    a0 = zero4(); a1 = zero4();
    for (int e0 = start; e0 < end; e0 += 16) {
        const int cnt = min(16, end - e0);
        const int mine = (e0 == start) ? first16 : ((sub < cnt) ? col[e0 + sub] : 0);
#define GN_L2(J)                                                                          \
        if (J < cnt) {                                                                    \
            const int o = row_bcast<J>(mine) * TS + 4 * sub;                              \
            const float4 u = *reinterpret_cast<const float4*>(Zl + o);                    \
            const float4 v = *reinterpret_cast<const float4*>(Ql + o);                    \
            a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;                           \
            a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;                           \
        }
        GN_L2(0) GN_L2(1) GN_L2(2) GN_L2(3) GN_L2(4) GN_L2(5) GN_L2(6) GN_L2(7)
        GN_L2(8) GN_L2(9) GN_L2(10) GN_L2(11) GN_L2(12) GN_L2(13) GN_L2(14) GN_L2(15)
#undef GN_L2
    }
}

// One workgroup = nt x 256 threads (nt = ceil(n/32) <= 2): row tile t is served by waves 4t .. 4t+3.
// KEEP: the forward's kept activations are read back (a compile-time switch: the recomputing instance stays as it was)
template <bool KEEP>
__global__ __launch_bounds__(512) void k_tiny_bwd64(const int* __restrict__ rowptr, const int* __restrict__ col, int n,
                                                    long rows, const float* __restrict__ sol, const float* __restrict__ x,
                                                    const float* __restrict__ gS, const float* __restrict__ gI,
                                                    const float* __restrict__ gR, const float* __restrict__ W,
                                                    const float* __restrict__ bias, const float* __restrict__ w3,
                                                    const float* __restrict__ b3, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, TinyBwdSched sched,
                                                    float* __restrict__ part_all, const float* __restrict__ keep) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const PartLayout L{64};
    const int nt = (n + TILE_ROWS - 1) / TILE_ROWS, tile_f = TILE_ROWS * TS;
    float* Wl = lds;                               // W   (Z = sigmoid(y W^T + b))
    float* WlT = Wl + 64 * TS;                     // W^T (g_Y = dpre W)
    float* ZIt = WlT + 64 * TS;                    // gather table Z_I  [nt*32][TS]
    float* Qt = ZIt + nt * tile_f;                 // gather table q    [nt*32][TS]
    float* tiles = Qt + nt * tile_f;               // per row tile: Dt0 | Dt1 | Yt0 | Yt1
    const int t = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int lane = tid & 63, w = tid >> 6, g = lane >> 4, sub = lane & 15;
    const int i16 = lane & 15, kq = lane >> 4;
    float* Dt0 = tiles + (size_t)(4 * t) * tile_f;
    float* Dt1 = Dt0 + tile_f;
    float* Yt0 = Dt1 + tile_f;
    float* Yt1 = Yt0 + tile_f;
    float* ZIm = ZIt + t * tile_f;                 // this tile's rows of the Z_I table
    float* Qm = Qt + t * tile_f;
    if (t == 0) { load_W_to_lds<false>(W, Wl); load_W_to_lds<true>(W, WlT); }     // threads 0..255
    const float bias_l = bias[16 * w + i16];
    const long base = (long)blockIdx.x * n;
    const size_t slab = (size_t)rows * 64;

    int lr[2], node[2]; bool valid[2]; size_t off[2];
    float bt[2] = {0.f, 0.f}, gm[2] = {0.f, 0.f};
    int e_lo[2] = {0, 0}, e_hi[2] = {0, 0}, first16[2] = {0, 0};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        lr[p] = w * 8 + 4 * p + g;
        node[p] = t * TILE_ROWS + lr[p];
        valid[p] = node[p] < n;
        off[p] = (size_t)(base + node[p]) * 64 + 4 * sub;
        if (valid[p]) {
            bt[p] = sol[3 * slab + (size_t)(base + node[p]) * 64];          // beta / gamma slab (ode_nn_ngraph_sim.py:60)
            gm[p] = sol[3 * slab + (size_t)(base + node[p]) * 64 + 1];
            e_lo[p] = rowptr[node[p]]; e_hi[p] = rowptr[node[p] + 1];
            first16[p] = (e_lo[p] + sub < e_hi[p]) ? col[e_lo[p] + sub] : 0;
        }
    }
    float4 w3v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w3v[k] = ld4g(w3 + k * 64 + 4 * sub);

    float4 aS[2] = {zero4(), zero4()}, aI[2] = {zero4(), zero4()}, aR[2] = {zero4(), zero4()};
    HeadAcc hacc;
#pragma unroll
    for (int k = 0; k < 4; ++k) { hacc.dw3[k] = zero4(); hacc.db3[k] = 0.f; hacc.dw2[k] = 0.f; }
    hacc.db2 = 0.f;
    f32x4 accW[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) accW[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;

    // Rows of sol at one grid point: y_i feeds the interval's MFMAs AND (one interval earlier in the sweep) the head's
    // VJP at that grid point, so each row is fetched once, one interval ahead of its first use.
    // With the forward's kept activations (gn_keep_zs / gn_keep_zi: grid points 0 .. n_steps-1) the rows' Z_S(y_i), Z_I(y_i)
    // ride along with the y rows instead of being recomputed: one matrix phase and one barrier less per interval -- the
    // sweep is a chain of matrix phases on ONE CU, so that is where its time goes.
    struct GridRows { float4 y[2][3]; float gout[2][3]; float4 zs[2], zi[2]; };
    auto fetch = [&](int gi, GridRows& r) {
        const int s = sched.slot[gi];
        const float* Yg = sol + (size_t)gi * 4 * slab;
        const bool kept = KEEP && gi >= 1 && gi < sched.n_steps;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            r.y[p][0] = r.y[p][1] = r.y[p][2] = zero4();
            r.gout[p][0] = r.gout[p][1] = r.gout[p][2] = 0.f;
            r.zs[p] = r.zi[p] = zero4();
            if (!valid[p]) continue;
            if (kept) { r.zs[p] = ld4g(gn_keep_zs(keep, rows, gi) + off[p]); r.zi[p] = ld4g(gn_keep_zi(keep, rows, gi) + off[p]); }
            r.y[p][0] = ld4g(Yg + off[p]); r.y[p][1] = ld4g(Yg + slab + off[p]);
            if (s >= 0) {
                r.y[p][2] = ld4g(Yg + 2 * slab + off[p]);
                const size_t o = (size_t)s * rows + base + node[p];
                r.gout[p][0] = gS[o]; r.gout[p][1] = gI[o]; r.gout[p][2] = gR[o];
            }
        }
    };
    // dL/dsol at grid point gi enters through the head (a += ...), gnode_bwd.hip `head`
    auto head_at = [&](int gi, const GridRows& r) {
        if (sched.slot[gi] < 0) return;
#pragma unroll
        for (int p = 0; p < 2; ++p)       // padding rows: gout = 0 -> adds nothing
            head_vjp64(r.y[p], r.gout[p], w3v, b3, w2, b2, aS[p], aI[p], aR[p], hacc);
    };

    // a tile whose rows 16..31 are all padding (karate: n = 34 -> tile 1 holds 2 rows) runs half the MFMAs, and the
    // gW contraction only visits the 4-row slices that hold real rows (padding rows of dpre are zero)
    const int rows_here = min(TILE_ROWS, n - t * TILE_ROWS);
    const bool blk2 = rows_here > 16;
    const int s8_end = (rows_here + 3) / 4;
    // rows 16..31 of the tile buffers are read by the row owners even when no MFMA writes them
    if (!blk2)
#pragma unroll
        for (int p = 0; p < 2; ++p)
            if (lr[p] >= 16) {
                *reinterpret_cast<float4*>(Dt0 + lr[p] * TS + 4 * sub) = zero4();
                *reinterpret_cast<float4*>(ZIm + lr[p] * TS + 4 * sub) = zero4();
            }
    GridRows cur, nxt;
    fetch(sched.n_steps, cur);
    head_at(sched.n_steps, cur);
    __syncthreads();                                                            // W, W^T staged
    for (int i = sched.n_steps; i >= 1; --i) {
        const float dt = sched.dt[i - 1];
        // 1. y_i rows of this tile (fetched one interval ago); start fetching grid point i-1
        const bool kept = KEEP && i < sched.n_steps;                // uniform: the forward evaluated the RHS at y_i and kept Z
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<float4*>(Yt0 + lr[p] * TS + 4 * sub) = cur.y[p][0];
            *reinterpret_cast<float4*>(Yt1 + lr[p] * TS + 4 * sub) = cur.y[p][1];
            // (own rows only: nobody else touches them between the previous interval's last barrier and the next one)
            if (kept) *reinterpret_cast<float4*>(ZIm + lr[p] * TS + 4 * sub) = cur.zi[p];
        }
        float4 zs[2], zi[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) { zs[p] = cur.zs[p]; zi[p] = cur.zi[p]; }
        fetch(i - 1, nxt);
        if (!kept) {
            __syncthreads();
            // 2. Z_S (into Dt0 for now) and Z_I (straight into the gather table)
            if (blk2) { mfma_tile<true>(Yt0, Wl, Dt0, bias_l, w, lane); mfma_tile<true>(Yt1, Wl, ZIm, bias_l, w, lane); }
            else { mfma_tile16<true>(Yt0, Wl, Dt0, bias_l, w, lane); mfma_tile16<true>(Yt1, Wl, ZIm, bias_l, w, lane); }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                zs[p] = *reinterpret_cast<const float4*>(Dt0 + lr[p] * TS + 4 * sub);
                zi[p] = *reinterpret_cast<const float4*>(ZIm + lr[p] * TS + 4 * sub);
            }
        }
        // 3. q = beta (a_I - a_S) Z_S
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<float4*>(Qm + lr[p] * TS + 4 * sub) =
                make_float4(bt[p] * (aI[p].x - aS[p].x) * zs[p].x, bt[p] * (aI[p].y - aS[p].y) * zs[p].y,
                            bt[p] * (aI[p].z - aS[p].z) * zs[p].z, bt[p] * (aI[p].w - aS[p].w) * zs[p].w);
        }
        __syncthreads();
        // 4. both gathers, dpre (already times dt)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float4 ai, gq;
            gather2_row_lds(col, ZIt, Qt, e_lo[p], e_hi[p], first16[p], sub, ai, gq);
            float4 dS = zero4(), dI = zero4();
            if (valid[p]) {
#define GN_DP(c)                                                                              \
                {                                                                             \
                    const float v = bt[p] * (aI[p].c - aS[p].c);                              \
                    dS.c = dt * ((v * ai.c) * (zs[p].c * (1.0f - zs[p].c)));                  \
                    dI.c = dt * ((gq.c + gm[p] * (aR[p].c - aI[p].c)) * (zi[p].c * (1.0f - zi[p].c))); \
                }
                GN_DP(x) GN_DP(y) GN_DP(z) GN_DP(w)
#undef GN_DP
            }
            *reinterpret_cast<float4*>(Dt0 + lr[p] * TS + 4 * sub) = dS;
            *reinterpret_cast<float4*>(Dt1 + lr[p] * TS + 4 * sub) = dI;
        }
        __syncthreads();
        // 5a. gW += dpre^T y_i (wave w owns rows [16w, 16w+16) of gW), gb += column sums of dpre
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const float* D = X ? Dt1 : Dt0;
            const float* Y = X ? Yt1 : Yt0;
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                if (s8 >= s8_end) break;
                const int rr = 4 * s8 + kq;
                const float av = D[rr * TS + 16 * w + i16];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    accW[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Y[rr * TS + 16 * kt + i16], accW[kt], 0, 0, 0);
            }
        }
        if (tid < 64) {
            float sacc = 0.f;
            for (int rr = 0; rr < TILE_ROWS; ++rr) sacc += Dt0[rr * TS + tid] + Dt1[rr * TS + tid];
            accb += sacc;
        }
        // 5b. g_Y = dpre W, written over this tile's rows of the two gather tables (free since the barrier after the
        //     gathers; no barrier needed against 5a, which only reads Dt / Yt), then a += g_Y
        if (blk2) { mfma_tile<false>(Dt0, WlT, Qm, 0.f, w, lane); mfma_tile<false>(Dt1, WlT, ZIm, 0.f, w, lane); }
        else { mfma_tile16<false>(Dt0, WlT, Qm, 0.f, w, lane); mfma_tile16<false>(Dt1, WlT, ZIm, 0.f, w, lane); }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (!valid[p]) continue;
            const float4 uS = *reinterpret_cast<const float4*>(Qm + lr[p] * TS + 4 * sub);
            const float4 uI = *reinterpret_cast<const float4*>(ZIm + lr[p] * TS + 4 * sub);
            aS[p].x += uS.x; aS[p].y += uS.y; aS[p].z += uS.z; aS[p].w += uS.w;
            aI[p].x += uI.x; aI[p].y += uI.y; aI[p].z += uI.z; aI[p].w += uI.w;
        }
        // 6. dL/dsol[i-1]
        head_at(i - 1, nxt);
        cur = nxt;
    }

    // encoder backward at grid point 0: y0_X = relu(s_X w1 + b1) (ode_nn_ngraph_sim.py:151-156)
    float4 dw1 = zero4(), db1 = zero4();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (!valid[p]) continue;
        const float4* av[3] = {&aS[p], &aI[p], &aR[p]};
#pragma unroll
        for (int X = 0; X < 3; ++X) {
            const float s = x[(size_t)(base + node[p]) * (3 + 64) + X];
            const float4 y = ld4g(sol + X * slab + off[p]);
            const float4 mk = make_float4(y.x > 0.f ? av[X]->x : 0.f, y.y > 0.f ? av[X]->y : 0.f, y.z > 0.f ? av[X]->z : 0.f,
                                          y.w > 0.f ? av[X]->w : 0.f);
            dw1.x = fmaf(mk.x, s, dw1.x); dw1.y = fmaf(mk.y, s, dw1.y); dw1.z = fmaf(mk.z, s, dw1.z); dw1.w = fmaf(mk.w, s, dw1.w);
            db1.x += mk.x; db1.y += mk.y; db1.z += mk.z; db1.w += mk.w;
        }
    }

    // ---- fixed-order reductions over row tiles / lane groups through LDS (the tile buffers are free now)
    float* part = part_all + (size_t)blockIdx.x * L.total();
    float* red = tiles;
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[t * 4096 + (16 * w + 4 * kq + reg) * 64 + 16 * kt + i16] = accW[kt][reg];
    if (tid < 64) red[nt * 4096 + t * 64 + tid] = accb;
    __syncthreads();
    for (int e = threadIdx.x; e < 4096 + 64; e += blockDim.x) {
        float s = 0.f;
        if (e < 4096) { for (int tt = 0; tt < nt; ++tt) s += red[tt * 4096 + e]; part[L.oW() + e] = s; }
        else          { for (int tt = 0; tt < nt; ++tt) s += red[nt * 4096 + tt * 64 + (e - 4096)]; part[L.ob() + (e - 4096)] = s; }
    }
    __syncthreads();
    // head partials [group][4*64 + 9 (+3 pad)], then encoder partials [group][2*64]
    {
        const int ngroups = nt * 16, grp = t * 16 + w * 4 + g;
        constexpr int NE = 4 * 64 + 12;              // padded to a multiple of 4 floats so rows stay 16-B aligned
        float* mine = red + (size_t)grp * NE;
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(mine + k * 64 + 4 * sub) = hacc.dw3[k];
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { mine[256 + k] = hacc.db3[k]; mine[260 + k] = hacc.dw2[k]; }
            mine[264] = hacc.db2;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 265; e += blockDim.x) {
            float s = 0.f;
            for (int gi = 0; gi < ngroups; ++gi) s += red[(size_t)gi * NE + e];
            part[L.ow3() + e] = s;
        }
        __syncthreads();
        float* mine2 = red + (size_t)grp * 128;
        *reinterpret_cast<float4*>(mine2 + 4 * sub) = dw1;
        *reinterpret_cast<float4*>(mine2 + 64 + 4 * sub) = db1;
        __syncthreads();
        for (int e = threadIdx.x; e < 128; e += blockDim.x) {
            float s = 0.f;
            for (int gi = 0; gi < ngroups; ++gi) s += red[(size_t)gi * 128 + e];
            part[L.ow1() + e] = s;
        }
    }
}

static size_t tiny_bwd_lds_bytes(int n) {
    const int nt = (n + TILE_ROWS - 1) / TILE_ROWS;
    return sizeof(float) * ((size_t)2 * 64 * TS + (size_t)6 * nt * TILE_ROWS * TS);
}

#ifndef GN_TINY_TRAIN
#define GN_TINY_TRAIN 1
#endif
bool gn_tiny_bwd64_ok(const gnode_graph_s* g, long rows, int H, int n_steps) {
    if (!GN_TINY_TRAIN) return false;
    return H == 64 && g->n <= 2 * TILE_ROWS && n_steps >= 0 && n_steps <= 128 && rows / g->n <= BWD_NWG &&
           tiny_bwd_lds_bytes(g->n) <= 160 * 1024;
}

int gn_bwd_tiny_set_attributes() {      // once per device, from gnode_graph_create
    GN_HIP(hipFuncSetAttribute((const void*)k_tiny_bwd64<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    GN_HIP(hipFuncSetAttribute((const void*)k_tiny_bwd64<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

int gn_launch_tiny_bwd64(const gnode_graph_s* g, long rows, const float* x, const gnode_params* p, const float* dt_host,
                         int n_steps, const int32_t* out_rows_host, int n_out, const float* sol, const float* gS,
                         const float* gI, const float* gR, float* part, const float* keep, hipStream_t st) {
    TinyBwdSched sched;
    sched.n_steps = n_steps;
    for (int k = 0; k < n_steps; ++k) sched.dt[k] = dt_host[k];
    for (int k = 0; k <= n_steps; ++k) sched.slot[k] = out_rows_host ? (short)-1 : (short)k;
    if (out_rows_host) {
        GN_CHECK_ARG(n_out < 32768, "gnode_backward_f32: too many output rows (%d)", n_out);
        for (int i = 0; i < n_out; ++i) sched.slot[out_rows_host[i]] = (short)i;
    }
    const unsigned B = (unsigned)(rows / g->n);
    const unsigned threads = 256u * (unsigned)((g->n + TILE_ROWS - 1) / TILE_ROWS);
    auto kern = keep ? k_tiny_bwd64<true> : k_tiny_bwd64<false>;
    hipLaunchKernelGGL(kern, dim3(B), dim3(threads), tiny_bwd_lds_bytes(g->n), st, g->rowptr, g->col, g->n, rows, sol, x,
                       gS, gI, gR, p->odefunc_linear_weight, p->odefunc_linear_bias, p->linear3_weight, p->linear3_bias,
                       p->linearS2_weight, p->linearS2_bias, sched, part, keep);
    GN_LAUNCH_CHECK();
    return 0;
}
