// H = 64 fast path of the GN-ODE step for MI355X (gfx950).
//
// One launch per Euler step (per chunk of samples):
//   k_step64<FUSE,PRJ,RPG>  persistent 256-thread workgroups walk 16-node tiles (RPG = 1, 5 workgroups per
//                           CU; RPG = 2: 32-node tiles, 4 per CU):
//     P1  stage the tile's Y_S rows to LDS (coalesced 16 B/lane, 4 rows per wave
//         instruction) and pull-gather AI = sum_{c in adj} Z_I[c] with one 16-lane
//         group per row (column indices broadcast inside the group by DPP
//         row_newbcast, neighbour rows as 256-B coalesced reads, 8 in flight);
//     P2  Z_S = sigmoid(Y_S W^T + b) on the fp32 matrix cores
//         (v_mfma_f32_16x16x4_f32, exact fp32), W^T resident in LDS for the
//         whole launch, each wave a 16-column slab of the tile's output;
//     P3  SIR derivative + Euler update of Y_S, Y_I, Y_R in place, optional
//         trajectory write, fused read-out head + 3-way softmax (16-lane DPP
//         reductions, no LDS traffic);
//     P4  (FUSE) Z_I of the NEXT step from the freshly updated Y_I rows, again on
//         the matrix cores, so the separate node-MLP launch disappears and
//         Y_S / Y_I are read once per step.
//     PRJ (inference): the R compartment is carried as w3 . Y_R (4 floats per row) -- it only
//         feeds the read-out, whose first layer is linear.
//     Rows longer than the hub threshold arrive pre-summed from gnode_hub.hip.
//   k_tiny64<PRJ>    graphs that fit one workgroup's LDS: ALL Euler steps in one launch.
//   k_mlp64          the same MFMA tile engine alone (RHS API, step 0, RK4 stages).
//
// Reference semantics: ode_nn_ngraph_sim.py:58-96 (RHS), :168 (euler), :172-187 (head).
#include "gnode_common.h"
#include "gnode_h64.h"
#include "gnode_mfma64.h"
#include <algorithm>
#include <cstdlib>

template <bool NT>
__device__ __forceinline__ float4 ld4so(const float* b, unsigned off) { return ld4s<NT>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(b) + off)); }
template <bool NT>
__device__ __forceinline__ void st4so(float* b, unsigned off, float4 v) { st4s<NT>(reinterpret_cast<float*>(reinterpret_cast<char*>(b) + off), v); }

// ---- pull-gather of one row by a 16-lane group ---------------------------------------
// ascending-column accumulation order = the CPU scatter_add_ order of the reference.
#ifndef GN_FWD_NB
#define GN_FWD_NB 8          // neighbour rows in flight per lane group (4 -> 8: 379 -> 375 us per launch on the 75k graph x 8,
                             // mid-size forwards -10 %; 16 would spill at 5 workgroups per CU)
#endif
__device__ __forceinline__ float4 gather_row64(const int* __restrict__ rowptr, const int* __restrict__ col,
                                               const float* __restrict__ ZI_base, int node, bool valid, int sub) {
    float4 acc = zero4();
    int start = 0, end = 0;
    if (valid) { start = rowptr[node]; end = rowptr[node + 1]; }
    const unsigned lane_b = 16u * sub;
    for (int e0 = start; e0 < end; e0 += 16) {
        const int cnt = min(16, end - e0);
        const unsigned mine = (sub < cnt) ? (unsigned)col[e0 + sub] * 256u : 0u;   // byte offset of the neighbour row
#define GN_LD(K, V) float4 V = zero4(); if (K < cnt) V = ld4o(ZI_base, (unsigned)row_bcast<(K) & 15>((int)mine) + lane_b);
#define GN_AC(V) acc.x += V.x; acc.y += V.y; acc.z += V.z; acc.w += V.w;
#define GN_G4(J)                                                                               \
        if (J < cnt) {                                                                         \
            GN_LD(J, v0) GN_LD(J + 1, v1) GN_LD(J + 2, v2) GN_LD(J + 3, v3)                    \
            GN_AC(v0) GN_AC(v1) GN_AC(v2) GN_AC(v3)                                            \
        }
#define GN_G8(J)                                                                               \
        if (J < cnt) {                                                                         \
            GN_LD(J, v0) GN_LD(J + 1, v1) GN_LD(J + 2, v2) GN_LD(J + 3, v3)                    \
            GN_LD(J + 4, v4) GN_LD(J + 5, v5) GN_LD(J + 6, v6) GN_LD(J + 7, v7)                \
            GN_AC(v0) GN_AC(v1) GN_AC(v2) GN_AC(v3) GN_AC(v4) GN_AC(v5) GN_AC(v6) GN_AC(v7)    \
        }
        if (GN_FWD_NB == 8) { GN_G8(0) GN_G8(8) } else { GN_G4(0) GN_G4(4) GN_G4(8) GN_G4(12) }
#undef GN_G8
#undef GN_G4
#undef GN_AC
#undef GN_LD
    }
    return acc;
}

// --------------------------------------------------------------------------- k_mlp64
__global__ __launch_bounds__(256) void k_mlp64(const float* __restrict__ X, const float* __restrict__ W,
                                               const float* __restrict__ bias, float* __restrict__ Z, long nrows) {
    __shared__ __attribute__((aligned(16))) float Wl[64 * TS];
    __shared__ __attribute__((aligned(16))) float T[TILE_ROWS * TS];
    __shared__ __attribute__((aligned(16))) float T2[TILE_ROWS * TS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, sub = lane & 15;
    load_W_to_lds<false>(W, Wl);
    const float bias_l = bias[16 * w + (lane & 15)];
    const long ntiles = (nrows + TILE_ROWS - 1) / TILE_ROWS;
    const int lr0 = w * 8 + g, lr1 = w * 8 + 4 + g;
    long t = blockIdx.x;
    float4 x0 = zero4(), x1 = zero4();
    if (t < ntiles) {
        const long r0 = t * TILE_ROWS + lr0, r1 = t * TILE_ROWS + lr1;
        if (r0 < nrows) x0 = ld4g(X + (size_t)r0 * 64 + 4 * sub);
        if (r1 < nrows) x1 = ld4g(X + (size_t)r1 * 64 + 4 * sub);
    }
    for (; t < ntiles; t += gridDim.x) {
        *reinterpret_cast<float4*>(T + lr0 * TS + 4 * sub) = x0;
        *reinterpret_cast<float4*>(T + lr1 * TS + 4 * sub) = x1;
        const long tn = t + gridDim.x;                 // prefetch the next tile under the MFMAs
        x0 = zero4(); x1 = zero4();
        if (tn < ntiles) {
            const long r0 = tn * TILE_ROWS + lr0, r1 = tn * TILE_ROWS + lr1;
            if (r0 < nrows) x0 = ld4g(X + (size_t)r0 * 64 + 4 * sub);
            if (r1 < nrows) x1 = ld4g(X + (size_t)r1 * 64 + 4 * sub);
        }
        __syncthreads();
        mfma_tile<true>(T, Wl, T2, bias_l, w, lane);
        __syncthreads();
        const long r0 = t * TILE_ROWS + lr0, r1 = t * TILE_ROWS + lr1;
        if (r0 < nrows) st4g(Z + (size_t)r0 * 64 + 4 * sub, *reinterpret_cast<const float4*>(T2 + lr0 * TS + 4 * sub));
        if (r1 < nrows) st4g(Z + (size_t)r1 * 64 + 4 * sub, *reinterpret_cast<const float4*>(T2 + lr1 * TS + 4 * sub));
    }
}

// --------------------------------------------------------------------------- k_step64
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

#ifndef GN_RPG1_OCC
#define GN_RPG1_OCC 5
#endif
// RPG = rows per 16-lane group and tile: 2 -> 32-row tiles, 4 workgroups per CU; 1 (default) -> 16-row tiles: half
// the LDS, the two row gathers of a lane group no longer run back to back, and no spills at 5 workgroups per CU.
// Measured per launch, 75k graph x 8 samples: RPG=2 406 us; RPG=1 at 3/4/5/6/8 workgroups per CU 457/403/380/389/417 us;
// one sample (latency-bound): 67 -> 59 us, fb-social-size graph (60 tiles): 12.6 -> 8.3 us per step.
template <bool FUSE, bool PRJ, int RPG>
__global__ __launch_bounds__(256, (RPG == 2 ? 4 : GN_RPG1_OCC)) void k_step64(const int* __restrict__ rowptr, const int* __restrict__ col, int n,
                                                long rows, int tiles_per_sample, long total_tiles,
                                                float* Y, const float* __restrict__ ZI,
                                                float* __restrict__ ZI_next, const float* __restrict__ W,
                                                const float* __restrict__ bias, const float* __restrict__ beta,
                                                const float* __restrict__ gamma, float dt,
                                                const float* __restrict__ w3, const float* __restrict__ b3,
                                                const float* __restrict__ w2, const float* __restrict__ b2,
                                                float* __restrict__ PR, Step64Out out,
                                                const int* __restrict__ hubidx, const float* __restrict__ AIhub, int n_hub) {
    // measured on the 75k-node benchmark and fixed: non-temporal streaming state accesses (+4.2 %: the gather
    // table keeps the L2) and 8 XCD-affine tile queues (+0.7 %)
    constexpr bool NT = true;
    constexpr int XQ = 8;
    __shared__ __attribute__((aligned(16))) float Wl[64 * TS];
    constexpr int TR = 16 * RPG;                  // rows per tile
    __shared__ __attribute__((aligned(16))) float T[TR * TS];
    __shared__ __attribute__((aligned(16))) float T2[TR * TS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, sub = lane & 15;
    load_W_to_lds<false>(W, Wl);
    const float bias_l = bias[16 * w + (lane & 15)];
    const size_t slab = (size_t)rows * 64;
    // state in / state out: in place (inference), or trajectory point k -> k+1 when the caller keeps `sol`
    // (then the trajectory IS the state and nothing is written twice)
    const float* YS = Y; const float* YI = Y + slab; const float* YR = Y + 2 * slab;
    float* Yo = out.sol ? out.sol : Y;
    float* YSo = Yo; float* YIo = Yo + slab; float* YRo = Yo + 2 * slab;

    int lr[RPG];
#pragma unroll
    for (int p = 0; p < RPG; ++p) lr[p] = w * 4 * RPG + 4 * p + g;

    // Tile walk.  XQ > 1: the tile range is cut into XQ contiguous queues (with 8 samples per launch a queue
    // is one sample) and workgroup i serves queue i % XQ.  Workgroups are dealt round-robin over the 8 XCDs,
    // so queue q is gathered through ONE XCD's L2: a sample's Z_I table competes for 4 MB instead of being
    // spread over all eight L2s.  Placement only affects speed: every tile is visited exactly once either way.
    const int xq = (XQ > 1 && gridDim.x % XQ == 0 && total_tiles >= 4 * XQ) ? XQ : 1;
    const int q = blockIdx.x % xq;
    const long q_lo = total_tiles * q / xq, q_hi = total_tiles * (q + 1) / xq;
    // (sample, tile) advance incrementally: no 64-bit division in the tile loop
    const long t_first = q_lo + blockIdx.x / xq;
    long b = t_first / tiles_per_sample;
    int tile = (int)(t_first - b * tiles_per_sample);
    const int t_stride = gridDim.x / xq;
    for (long t = t_first; t < q_hi; t += t_stride, tile += t_stride) {
        while (tile >= tiles_per_sample) { tile -= tiles_per_sample; ++b; }
        const long base = b * n;
        int node[RPG]; bool valid[RPG]; unsigned off[RPG];       // off: BYTE offset of this lane's 16 B inside a slab
        float4 ys[RPG], ai[RPG], yi[RPG], yr[RPG], zi[RPG];
#pragma unroll
        for (int p = 0; p < RPG; ++p) {
            node[p] = tile * TR + lr[p];
            valid[p] = node[p] < n;
            off[p] = (unsigned)(base + node[p]) * 256u + 16u * sub;
            ys[p] = valid[p] ? ld4so<NT>(YS, off[p]) : zero4();
        }
        // -------- P1: stage Y_S, issue own-row loads, gather
#pragma unroll
        for (int p = 0; p < RPG; ++p) {
            *reinterpret_cast<float4*>(T + lr[p] * TS + 4 * sub) = ys[p];
            yi[p] = valid[p] ? ld4so<NT>(YI, off[p]) : zero4();
            yr[p] = (!PRJ && valid[p]) ? ld4so<NT>(YR, off[p]) : zero4();
            zi[p] = valid[p] ? ld4o(ZI, off[p]) : zero4();
        }
        // (measured: gathering the two rows in lockstep with 8 loads in flight per lane is 25 % SLOWER --
        //  the memory system is already at its request-rate limit)
        // long rows ("hubs") were summed beforehand by the segment kernels of gnode_hub.hip
#pragma unroll
        for (int p = 0; p < RPG; ++p) {
            const int hub = (hubidx && valid[p]) ? hubidx[node[p]] : -1;
            if (hub >= 0) ai[p] = ld4o(AIhub, ((unsigned)b * (unsigned)n_hub + (unsigned)hub) * 256u + 16u * sub);
            else ai[p] = gather_row64(rowptr, col, ZI + (size_t)base * 64, node[p], valid[p], sub);
        }
        __syncthreads();
        // -------- P2: Z_S on the matrix cores
        if (RPG == 2) mfma_tile<true>(T, Wl, T2, bias_l, w, lane); else mfma_tile16<true>(T, Wl, T2, bias_l, w, lane);
        __syncthreads();
        // -------- P3: SIR derivative (ode_nn_ngraph_sim.py:75-77), Euler update, read-out
#pragma unroll
        for (int p = 0; p < RPG; ++p) {
            const float4 zs = *reinterpret_cast<const float4*>(T2 + lr[p] * TS + 4 * sub);
            float nb = 0.f, gm = 0.f;
            if (valid[p]) { nb = -beta[base + node[p]]; gm = gamma[base + node[p]]; }
            float4 dS, dI, dR;
            dS.x = nb * (ai[p].x * zs.x); dS.y = nb * (ai[p].y * zs.y); dS.z = nb * (ai[p].z * zs.z); dS.w = nb * (ai[p].w * zs.w);
            dR.x = gm * zi[p].x; dR.y = gm * zi[p].y; dR.z = gm * zi[p].z; dR.w = gm * zi[p].w;
            dI.x = -dS.x - dR.x; dI.y = -dS.y - dR.y; dI.z = -dS.z - dR.z; dI.w = -dS.w - dR.w;
            ys[p].x += dt * dS.x; ys[p].y += dt * dS.y; ys[p].z += dt * dS.z; ys[p].w += dt * dS.w;
            yi[p].x += dt * dI.x; yi[p].y += dt * dI.y; yi[p].z += dt * dI.z; yi[p].w += dt * dI.w;
            float prj[4] = {0.f, 0.f, 0.f, 0.f};
            if (PRJ) {
                // R only feeds the read-out, and its first layer is linear: carry w3 . Y_R (4 floats per row)
                // instead of Y_R (64):  w3 . (Y_R + dt*gamma*Z_I) = w3 . Y_R + dt*gamma*(w3 . Z_I)
                float4 pr = zero4();
                if (valid[p]) pr = ld4o(PR, (unsigned)(base + node[p]) * 16u);
                float4 w3r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) w3r[k] = ld4g(w3 + k * 64 + 4 * sub);   // L1-resident, shared with the read-out
                prj[0] = pr.x + dt * (gm * row_sum16(fmaf(w3r[0].x, zi[p].x, fmaf(w3r[0].y, zi[p].y, fmaf(w3r[0].z, zi[p].z, w3r[0].w * zi[p].w)))));
                prj[1] = pr.y + dt * (gm * row_sum16(fmaf(w3r[1].x, zi[p].x, fmaf(w3r[1].y, zi[p].y, fmaf(w3r[1].z, zi[p].z, w3r[1].w * zi[p].w)))));
                prj[2] = pr.z + dt * (gm * row_sum16(fmaf(w3r[2].x, zi[p].x, fmaf(w3r[2].y, zi[p].y, fmaf(w3r[2].z, zi[p].z, w3r[2].w * zi[p].w)))));
                prj[3] = pr.w + dt * (gm * row_sum16(fmaf(w3r[3].x, zi[p].x, fmaf(w3r[3].y, zi[p].y, fmaf(w3r[3].z, zi[p].z, w3r[3].w * zi[p].w)))));
                if (valid[p] && sub == 0) st4o(PR, (unsigned)(base + node[p]) * 16u, make_float4(prj[0], prj[1], prj[2], prj[3]));
            } else {
                yr[p].x += dt * dR.x; yr[p].y += dt * dR.y; yr[p].z += dt * dR.z; yr[p].w += dt * dR.w;
            }
            if (valid[p]) {
                st4so<NT>(YSo, off[p], ys[p]); st4so<NT>(YIo, off[p], yi[p]);
                if (!PRJ) st4so<NT>(YRo, off[p], yr[p]);
            }
            if (out.S) {
                float pS, pI, pR;
                readout64<PRJ>(ys[p], yi[p], yr[p], prj, sub, w3, b3, w2, b2, pS, pI, pR);
                if (valid[p] && sub == 0) {
                    out.S[base + node[p]] = pS; out.I[base + node[p]] = pI; out.R[base + node[p]] = pR;
                }
            }
            if (FUSE) *reinterpret_cast<float4*>(T + lr[p] * TS + 4 * sub) = yi[p];   // stage Y_I' for P4
        }
        if (FUSE) {
            // -------- P4: Z_I of the next step from the updated Y_I rows
            __syncthreads();
            if (RPG == 2) mfma_tile<true>(T, Wl, T2, bias_l, w, lane); else mfma_tile16<true>(T, Wl, T2, bias_l, w, lane);
            __syncthreads();
#pragma unroll
            for (int p = 0; p < RPG; ++p)
                if (valid[p]) st4o(ZI_next, off[p], *reinterpret_cast<const float4*>(T2 + lr[p] * TS + 4 * sub));
        }
        __syncthreads();   // T / T2 are rewritten by the next tile
    }
}

// --------------------------------------------------------------------------- k_tiny64: whole integration in ONE launch
// Graphs whose per-sample state fits a workgroup's LDS (n <= 96 nodes at H = 64: karate, dolphins -- the
// reference's shipped experiment is karate with batch size 1) are launch-latency bound with one launch per
// step (~7 us each).  Samples of a batch never interact (block-diagonal adjacency), so one workgroup owns one
// sample for ALL Euler steps: Y_S, Y_I, (Y_R | w3.Y_R) and both Z_I generations live in LDS, the gather reads
// LDS, W^T stays staged, and the only global traffic is the outputs (and the trajectory when training).
struct TinySched {
    float dt[128];
    short slot[128];      // output slot of grid point k+1, or -1
    int n_steps;
};

// (start, end, first 16 column ids) of a row are loop-invariant across the Euler steps: the caller keeps them in
// registers, so a step's gather touches global memory only for rows longer than 16 edges.
__device__ __forceinline__ float4 gather_row_lds(const int* __restrict__ col, const float* __restrict__ Zl, int start, int end,
                                                 int first16, int sub) {
    float4 acc = zero4();
    for (int e0 = start; e0 < end; e0 += 16) {
        const int cnt = min(16, end - e0);
        const int mine = (e0 == start) ? first16 : ((sub < cnt) ? col[e0 + sub] : 0);
#define GN_L1(J)                                                                                  \
        if (J < cnt) {                                                                            \
            const float4 v = *reinterpret_cast<const float4*>(Zl + row_bcast<J>(mine) * TS + 4 * sub); \
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;                               \
        }
        GN_L1(0) GN_L1(1) GN_L1(2) GN_L1(3) GN_L1(4) GN_L1(5) GN_L1(6) GN_L1(7)
        GN_L1(8) GN_L1(9) GN_L1(10) GN_L1(11) GN_L1(12) GN_L1(13) GN_L1(14) GN_L1(15)
#undef GN_L1
    }
    return acc;
}

// One workgroup = nt x 256 threads: tile t of the sample is served by waves 4t .. 4t+3, all tiles advance together.
template <bool PRJ>
__global__ __launch_bounds__(768) void k_tiny64(const int* __restrict__ rowptr, const int* __restrict__ col, int n,
                                                long rows, const float* __restrict__ Y0, const float* __restrict__ ZI0,
                                                const float* __restrict__ PR0, const float* __restrict__ W,
                                                const float* __restrict__ bias, const float* __restrict__ beta,
                                                const float* __restrict__ gamma, TinySched sched,
                                                const float* __restrict__ w3, const float* __restrict__ b3,
                                                const float* __restrict__ w2, const float* __restrict__ b2,
                                                float* __restrict__ So, float* __restrict__ Io, float* __restrict__ Ro,
                                                float* __restrict__ sol) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nt = (n + TILE_ROWS - 1) / TILE_ROWS, tile_f = TILE_ROWS * TS;
    float* Wl = lds;
    float* T2all = Wl + 64 * TS;                  // one scratch tile per row tile
    float* YS = T2all + nt * tile_f;
    float* YI = YS + nt * tile_f;
    float* ZA = YI + nt * tile_f;
    float* ZB = ZA + nt * tile_f;
    float* YR = ZB + nt * tile_f;                 // PRJ: [nt*32][4] projections, else [nt*32][TS] full rows
    const int t = threadIdx.x >> 8;               // this thread's row tile
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63, w = tid >> 6, g = lane >> 4, sub = lane & 15;
    float* T2 = T2all + t * tile_f;
    const long base = (long)blockIdx.x * n;
    const size_t slab = (size_t)rows * 64;
    if (t == 0) load_W_to_lds<false>(W, Wl);      // threads 0..255 (threadIdx.x == tid there)
    const float bias_l = bias[16 * w + (lane & 15)];
    int lrow[2]; bool valid[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        lrow[p] = t * TILE_ROWS + w * 8 + 4 * p + g;
        valid[p] = lrow[p] < n;
        const size_t off = (size_t)(base + lrow[p]) * 64 + 4 * sub;
        *reinterpret_cast<float4*>(YS + lrow[p] * TS + 4 * sub) = valid[p] ? ld4g(Y0 + off) : zero4();
        *reinterpret_cast<float4*>(YI + lrow[p] * TS + 4 * sub) = valid[p] ? ld4g(Y0 + slab + off) : zero4();
        *reinterpret_cast<float4*>(ZA + lrow[p] * TS + 4 * sub) = valid[p] ? ld4g(ZI0 + off) : zero4();
        if (PRJ) { if (sub == 0) *reinterpret_cast<float4*>(YR + lrow[p] * 4) = valid[p] ? ld4g(PR0 + (size_t)(base + lrow[p]) * 4) : zero4(); }
        else *reinterpret_cast<float4*>(YR + lrow[p] * TS + 4 * sub) = valid[p] ? ld4g(Y0 + 2 * slab + off) : zero4();
    }
    float nb[2] = {0.f, 0.f}, gm[2] = {0.f, 0.f};
    int e_lo[2] = {0, 0}, e_hi[2] = {0, 0}, first16[2] = {0, 0};
#pragma unroll
    for (int p = 0; p < 2; ++p)
        if (valid[p]) {
            nb[p] = -beta[base + lrow[p]]; gm[p] = gamma[base + lrow[p]];
            e_lo[p] = rowptr[lrow[p]]; e_hi[p] = rowptr[lrow[p] + 1];
            first16[p] = (e_lo[p] + sub < e_hi[p]) ? col[e_lo[p] + sub] : 0;
        }
    __syncthreads();
    float* Zc = ZA; float* Zn = ZB;
    // a tile whose rows 16..31 are all padding (karate: n = 34 -> tile 1 holds 2 rows) runs half the MFMAs
    const bool blk2 = n - t * TILE_ROWS > 16;
    if (!blk2)                                     // rows 16..31 of the scratch tile are read (by padding rows) but never written
#pragma unroll
        for (int p = 0; p < 2; ++p)
            if (lrow[p] - t * TILE_ROWS >= 16) *reinterpret_cast<float4*>(T2 + (lrow[p] - t * TILE_ROWS) * TS + 4 * sub) = zero4();
    // The rows' state lives in the owner lanes' REGISTERS across all steps; the LDS tiles YS / YI are only the MFMA
    // operands.  Per step: gather + SIR update + read-out (row-local), barrier, both node MLPs of the NEXT step
    // back to back on the matrix cores (Z_S' -> T2, Z_I' -> straight into the other gather table), barrier.
    float4 ys[2], yi[2], yr[2], pr[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        ys[p] = *reinterpret_cast<const float4*>(YS + lrow[p] * TS + 4 * sub);
        yi[p] = *reinterpret_cast<const float4*>(YI + lrow[p] * TS + 4 * sub);
        yr[p] = PRJ ? zero4() : *reinterpret_cast<const float4*>(YR + lrow[p] * TS + 4 * sub);
        pr[p] = PRJ ? *reinterpret_cast<const float4*>(YR + lrow[p] * 4) : zero4();
    }
    if (blk2) mfma_tile<true>(YS + t * tile_f, Wl, T2, bias_l, w, lane);      // Z_S of step 0
    else mfma_tile16<true>(YS + t * tile_f, Wl, T2, bias_l, w, lane);
    __syncthreads();
    for (int k = 0; k < sched.n_steps; ++k) {
        const float dt = sched.dt[k];
        const int slot = sched.slot[k];
        float* solk = sol ? sol + (size_t)(k + 1) * 4 * slab : nullptr;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float4 ai = gather_row_lds(col, Zc, e_lo[p], e_hi[p], first16[p], sub);
            const int lr = lrow[p] - t * TILE_ROWS;
            const float4 zs = *reinterpret_cast<const float4*>(T2 + lr * TS + 4 * sub);
            const float4 zi = *reinterpret_cast<const float4*>(Zc + lrow[p] * TS + 4 * sub);
            float4 dS, dI, dR;
            dS.x = nb[p] * (ai.x * zs.x); dS.y = nb[p] * (ai.y * zs.y); dS.z = nb[p] * (ai.z * zs.z); dS.w = nb[p] * (ai.w * zs.w);
            dR.x = gm[p] * zi.x; dR.y = gm[p] * zi.y; dR.z = gm[p] * zi.z; dR.w = gm[p] * zi.w;
            dI.x = -dS.x - dR.x; dI.y = -dS.y - dR.y; dI.z = -dS.z - dR.z; dI.w = -dS.w - dR.w;
            ys[p].x += dt * dS.x; ys[p].y += dt * dS.y; ys[p].z += dt * dS.z; ys[p].w += dt * dS.w;
            yi[p].x += dt * dI.x; yi[p].y += dt * dI.y; yi[p].z += dt * dI.z; yi[p].w += dt * dI.w;
            *reinterpret_cast<float4*>(YS + lrow[p] * TS + 4 * sub) = ys[p];
            *reinterpret_cast<float4*>(YI + lrow[p] * TS + 4 * sub) = yi[p];
            float prj[4] = {0.f, 0.f, 0.f, 0.f};
            if (PRJ) {
                float4 w3r[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) w3r[q] = ld4g(w3 + q * 64 + 4 * sub);
                prj[0] = pr[p].x + dt * (gm[p] * row_sum16(fmaf(w3r[0].x, zi.x, fmaf(w3r[0].y, zi.y, fmaf(w3r[0].z, zi.z, w3r[0].w * zi.w)))));
                prj[1] = pr[p].y + dt * (gm[p] * row_sum16(fmaf(w3r[1].x, zi.x, fmaf(w3r[1].y, zi.y, fmaf(w3r[1].z, zi.z, w3r[1].w * zi.w)))));
                prj[2] = pr[p].z + dt * (gm[p] * row_sum16(fmaf(w3r[2].x, zi.x, fmaf(w3r[2].y, zi.y, fmaf(w3r[2].z, zi.z, w3r[2].w * zi.w)))));
                prj[3] = pr[p].w + dt * (gm[p] * row_sum16(fmaf(w3r[3].x, zi.x, fmaf(w3r[3].y, zi.y, fmaf(w3r[3].z, zi.z, w3r[3].w * zi.w)))));
                pr[p] = make_float4(prj[0], prj[1], prj[2], prj[3]);
            } else {
                yr[p].x += dt * dR.x; yr[p].y += dt * dR.y; yr[p].z += dt * dR.z; yr[p].w += dt * dR.w;
            }
            if (solk && valid[p]) {
                const size_t off = (size_t)(base + lrow[p]) * 64 + 4 * sub;
                st4g(solk + off, ys[p]); st4g(solk + slab + off, yi[p]); st4g(solk + 2 * slab + off, yr[p]);
            }
            if (slot >= 0) {
                float pS, pI, pR;
                readout64<PRJ>(ys[p], yi[p], yr[p], prj, sub, w3, b3, w2, b2, pS, pI, pR);
                if (valid[p] && sub == 0) {
                    const size_t o = (size_t)slot * rows + base + lrow[p];
                    So[o] = pS; Io[o] = pI; Ro[o] = pR;
                }
            }
        }
        if (k + 1 == sched.n_steps) break;
        __syncthreads();                           // operand tiles complete; every read of Zc and T2 is done
        if (blk2) { mfma_tile<true>(YS + t * tile_f, Wl, T2, bias_l, w, lane); mfma_tile<true>(YI + t * tile_f, Wl, Zn + t * tile_f, bias_l, w, lane); }
        else { mfma_tile16<true>(YS + t * tile_f, Wl, T2, bias_l, w, lane); mfma_tile16<true>(YI + t * tile_f, Wl, Zn + t * tile_f, bias_l, w, lane); }
        __syncthreads();                           // every tile's Z_I' is in place before the next gather
        float* tmp = Zc; Zc = Zn; Zn = tmp;
    }
}

size_t gn_tiny64_lds_bytes(int n, bool prj) {
    const int nt = (n + TILE_ROWS - 1) / TILE_ROWS;
    const size_t tile_f = (size_t)TILE_ROWS * TS;
    return sizeof(float) * ((size_t)64 * TS + 5 * nt * tile_f + (prj ? (size_t)nt * TILE_ROWS * 4 : nt * tile_f));
}

// true when the whole integration of one sample fits a workgroup (and the schedule fits the kernel arguments)
bool gn_tiny64_ok(int n, int n_steps, int n_out, bool prj) {
    static const bool on = [] { const char* e = getenv("GNODE_TINY"); return !(e && e[0] == '0'); }();
    return on && n_steps >= 1 && n_steps <= 128 && n_out < 32768 && n <= 3 * TILE_ROWS &&
           gn_tiny64_lds_bytes(n, prj) <= 160 * 1024;
}

int gn_launch_tiny64(const gnode_graph_s* g, long rows, const float* Y0, const float* ZI0, const float* PR0, const float* W,
                     const float* bias, const float* beta, const float* gamma, const float* dt_host, const int* slot_host,
                     int n_steps, const gnode_params* p, float* S, float* I, float* R, float* sol, hipStream_t st) {
    TinySched sched;
    sched.n_steps = n_steps;
    for (int k = 0; k < n_steps; ++k) { sched.dt[k] = dt_host[k]; sched.slot[k] = (short)slot_host[k]; }
    const bool prj = PR0 != nullptr;
    const size_t lds = gn_tiny64_lds_bytes(g->n, prj);
    const unsigned B = (unsigned)(rows / g->n);
    const unsigned threads = 256u * (unsigned)((g->n + TILE_ROWS - 1) / TILE_ROWS);
    if (prj) {
        static bool attr = false;
        if (!attr) { GN_HIP(hipFuncSetAttribute((const void*)k_tiny64<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
        hipLaunchKernelGGL(k_tiny64<true>, dim3(B), dim3(threads), lds, st, g->rowptr, g->col, g->n, rows, Y0, ZI0, PR0, W, bias, beta,
                           gamma, sched, p->linear3_weight, p->linear3_bias, p->linearS2_weight, p->linearS2_bias, S, I, R, sol);
    } else {
        static bool attr = false;
        if (!attr) { GN_HIP(hipFuncSetAttribute((const void*)k_tiny64<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
        hipLaunchKernelGGL(k_tiny64<false>, dim3(B), dim3(threads), lds, st, g->rowptr, g->col, g->n, rows, Y0, ZI0, PR0, W, bias, beta,
                           gamma, sched, p->linear3_weight, p->linear3_bias, p->linearS2_weight, p->linearS2_bias, S, I, R, sol);
    }
    GN_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------- k_prologue64: everything before the first step
// encoder (ode_nn_ngraph_sim.py:151-156: relu(Linear(1,H)) on S0, I0, R0 -- a row select, the inputs are 0/1),
// beta / gamma extraction, trajectory point 0, read-out at grid point 0, projected R (PRJ) and Z_I(y_0) on the
// matrix cores: one launch instead of four (encoder, read-out, node MLP, R projection).
__global__ __launch_bounds__(256) void k_prologue64(const float* __restrict__ x, const float* __restrict__ w1,
                                                    const float* __restrict__ b1, const float* __restrict__ W,
                                                    const float* __restrict__ bias, const float* __restrict__ w3,
                                                    const float* __restrict__ b3, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, float* __restrict__ Y,
                                                    float* __restrict__ beta, float* __restrict__ gamma,
                                                    float* __restrict__ sol0, float* __restrict__ ZI, float* __restrict__ PR,
                                                    float* __restrict__ S0, float* __restrict__ I0, float* __restrict__ R0,
                                                    long rows) {
    __shared__ __attribute__((aligned(16))) float Wl[64 * TS];
    __shared__ __attribute__((aligned(16))) float T[TILE_ROWS * TS];
    __shared__ __attribute__((aligned(16))) float T2[TILE_ROWS * TS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, sub = lane & 15;
    load_W_to_lds<false>(W, Wl);
    const float bias_l = bias[16 * w + (lane & 15)];
    const float4 wv = ld4g(w1 + 4 * sub), bv = ld4g(b1 + 4 * sub);
    const size_t slab = (size_t)rows * 64;
    const long ntiles = (rows + TILE_ROWS - 1) / TILE_ROWS;
    const int lr[2] = {w * 8 + g, w * 8 + 4 + g};
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        long r[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            r[p] = t * TILE_ROWS + lr[p];
            const bool valid = r[p] < rows;
            float4 yS = zero4(), yI = zero4(), yR = zero4();
            if (valid) {
                const float* xr = x + r[p] * 67;
                const float s0 = xr[0], i0 = xr[1], r0 = xr[2];
                auto enc = [&](float v) {
                    return make_float4(fmaxf(fmaf(wv.x, v, bv.x), 0.f), fmaxf(fmaf(wv.y, v, bv.y), 0.f),
                                       fmaxf(fmaf(wv.z, v, bv.z), 0.f), fmaxf(fmaf(wv.w, v, bv.w), 0.f));
                };
                yS = enc(s0); yI = enc(i0); yR = enc(r0);
                const size_t off = (size_t)r[p] * 64 + 4 * sub;
                st4g(Y + off, yS); st4g(Y + slab + off, yI); st4g(Y + 2 * slab + off, yR);
                if (sub == 0) { beta[r[p]] = xr[3]; gamma[r[p]] = xr[4]; }
                if (sol0) {
                    st4g(sol0 + off, yS); st4g(sol0 + slab + off, yI); st4g(sol0 + 2 * slab + off, yR);
                    const float* bg = xr + 3 + 4 * sub;
                    st4g(sol0 + 3 * slab + off, make_float4(bg[0], bg[1], bg[2], bg[3]));
                }
            }
            *reinterpret_cast<float4*>(T + lr[p] * TS + 4 * sub) = yI;
            if (PR) {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 c = ld4g(w3 + k * 64 + 4 * sub);
                    v[k] = row_sum16(fmaf(c.x, yR.x, fmaf(c.y, yR.y, fmaf(c.z, yR.z, c.w * yR.w))));
                }
                if (valid && sub == 0) st4g(PR + (size_t)r[p] * 4, make_float4(v[0], v[1], v[2], v[3]));
            }
            if (S0) {
                float pS, pI, pR;
                const float none[4] = {0.f, 0.f, 0.f, 0.f};
                readout64<false>(yS, yI, yR, none, sub, w3, b3, w2, b2, pS, pI, pR);
                if (valid && sub == 0) { S0[r[p]] = pS; I0[r[p]] = pI; R0[r[p]] = pR; }
            }
        }
        __syncthreads();
        mfma_tile<true>(T, Wl, T2, bias_l, w, lane);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; ++p)
            if (r[p] < rows) st4g(ZI + (size_t)r[p] * 64 + 4 * sub, *reinterpret_cast<const float4*>(T2 + lr[p] * TS + 4 * sub));
        __syncthreads();
    }
}

int gn_launch_prologue64(const float* x, const gnode_params* p, float* Y, float* beta, float* gamma, float* sol0, float* ZI,
                         float* PR, float* S0, float* I0, float* R0, long rows, hipStream_t st) {
    const long ntiles = (rows + TILE_ROWS - 1) / TILE_ROWS;
    hipLaunchKernelGGL(k_prologue64, dim3((unsigned)std::min<long>(ntiles, 2048)), dim3(256), 0, st, x, p->linearS1_weight,
                       p->linearS1_bias, p->odefunc_linear_weight, p->odefunc_linear_bias, p->linear3_weight, p->linear3_bias,
                       p->linearS2_weight, p->linearS2_bias, Y, beta, gamma, sol0, ZI, PR, S0, I0, R0, rows);
    GN_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------- host launchers
// grid sizing knob for experiments: GNODE_WGS_PER_CU = k (persistent grid of k*CUs), 0 = one workgroup per tile
static int wgs_per_cu() {
    static const int v = [] { const char* e = getenv("GNODE_WGS_PER_CU"); return e ? atoi(e) : 4; }();
    return v;
}
static int g_num_cu = 0;
static int num_cus() {
    if (g_num_cu == 0) {
        int dev = 0, cu = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, dev) == hipSuccess) cu = p.multiProcessorCount;
        }
        g_num_cu = cu > 0 ? cu : 256;
    }
    return g_num_cu;
}

int gn_launch_mlp64(const float* X, const float* W, const float* b, float* Z, long nrows, hipStream_t st) {
    if (nrows <= 0) return 0;
    const long ntiles = (nrows + TILE_ROWS - 1) / TILE_ROWS;
    const int k = wgs_per_cu();
    const int grid = (int)(k > 0 ? std::min<long>(ntiles, (long)num_cus() * k) : ntiles);
    hipLaunchKernelGGL(k_mlp64, dim3(grid), dim3(256), 0, st, X, W, b, Z, nrows);
    GN_LAUNCH_CHECK();
    return 0;
}

int gn_launch_step64(gnode_graph_s* g, long rows, float* Y, const float* ZI, float* ZI_next, const float* W,
                     const float* bias, const float* beta, const float* gamma, float dt, const gnode_params* p,
                     float* PR, Step64Out out, bool fuse, hipStream_t st) {
    GN_CHECK_ARG(rows < (1L << 24), "H=64 step kernel addresses rows with 32-bit byte offsets: rows=%ld >= 2^24 per launch "
                 "(split the batch)", rows);
    static const int rpg = [] { const char* e = getenv("GNODE_RPG"); return (e && e[0] == '2') ? 2 : 1; }();
    const int tr = 16 * rpg;
    const int tps = (g->n + tr - 1) / tr;
    const long total = (long)(rows / g->n) * tps;
    const float* AIhub = nullptr;
    if (int e = gn_hub_gather(g, rows / g->n, 64, ZI, nullptr, &AIhub, nullptr, st)) return e;
    const int k = wgs_per_cu() > 0 ? (rpg == 1 ? wgs_per_cu() * GN_RPG1_OCC / 4 : wgs_per_cu()) : 0;
    // (measured: shrinking the grid so that every persistent workgroup gets the same number of tiles is 3 % SLOWER
    //  than filling all 4 x CUs slots and accepting a +-1 tile imbalance -- residency matters more)
    const int grid = (int)(k > 0 ? std::min<long>(total, (long)num_cus() * k) : total);
    const bool prj = PR != nullptr;
#define GN_STEP(F, P, Q)                                                                                                    \
    hipLaunchKernelGGL((k_step64<F, P, Q>), dim3(grid), dim3(256), 0, st, g->rowptr, g->col, g->n, rows, tps, total, Y, ZI, \
                       ZI_next, W, bias, beta, gamma, dt, p->linear3_weight, p->linear3_bias, p->linearS2_weight,           \
                       p->linearS2_bias, PR, out, g->hubidx, AIhub, g->n_hub)
#define GN_STEP_Q(F, P) do { if (rpg == 1) GN_STEP(F, P, 1); else GN_STEP(F, P, 2); } while (0)
    if (fuse) { if (prj) GN_STEP_Q(true, true); else GN_STEP_Q(true, false); }
    else { if (prj) GN_STEP_Q(false, true); else GN_STEP_Q(false, false); }
#undef GN_STEP_Q
#undef GN_STEP
    GN_LAUNCH_CHECK();
    return 0;
}
