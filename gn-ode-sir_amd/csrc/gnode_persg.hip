// Persistent one-launch integration and adjoint sweep for the SMALL hidden sizes (H = 8, 16, 32: lane groups of H/4 lanes per
// row; the multi-graph launcher trains with hidden 8 on a concatenated batch of graphs, monitorer-ngraphs.py:10,20,22).
//
// Where it sits: the per-step forms (k_step_generic / k_bwd_fused_generic, one launch per Euler step / adjoint interval) are
// launch-latency bound on such batches -- 22k rows x 32-byte rows is 0.7 MB of state, 14.9 / 27.8 us per step / interval
// measured -- exactly where H = 64 stood before gnode_pers64.hip.  Same cure, simpler shape:
//
//   * ONE launch, one workgroup per CU (dynamic LDS > half of the CU's 160 KB).  A workgroup is 1 .. 4 waves, a wave holds
//     64 / LPR rows of one sample (32 at H = 8); the plan takes the FEWEST rows per workgroup that keeps the whole batch resident
//     (gn_persg_plan: the busiest CU's gather is what a step waits for), the graph's row map deals the rows longest first in
//     snake order (gn_persg_build).  A row's state stays in REGISTERS for every step / interval;
//   * only the gather tables travel: Z_I (forward), Z_I and q (adjoint), double-buffered; the hand-off between steps is the
//     measured form of MI355X_MICROARCH.md's table for groups that span XCDs -- every storing wave `s_waitcnt vmcnt(0)`,
//     workgroup barrier, ONE lane publishes the workgroup's epoch flag; the consumers' wave 0 polls all flags, barrier, every
//     table load `sc1` (pers_wait / pers_ld / pers_st of gnode_pers64_dev.h);
//   * a row's neighbour ids never change: they are staged ONCE in LDS as table byte offsets, so a step's gather is LDS id ->
//     row load (one round trip), not row extent -> column ids -> rows;
//   * hub rows (longer than GN_HUB_T): their <= 32-edge segments are dealt to the workgroup's lane groups, partial sums go
//     through LDS and the owner adds them in segment order -- the same segments, the same ascending sums, the same
//     segment-order total as k_hub_seg + the per-step consumers, so the forward's bits agree with the per-step path;
//   * adjoint: the interval's Z(y_{i-1}), q, the head's VJP and a += dt g_Y as in k_bwd_fused_generic; gW = sum dpre^T y is
//     accumulated in REGISTERS (a lane owns 4 rows of the H x H matrix: 4 H accumulators) behind the flag, where it overlaps
//     the barrier's flight, and reduced once at the end of the sweep (the per-interval kernel walks its row tile with 64 of
//     256 threads every interval).  Summation order differs from the per-interval path: gradients agree to rounding, not bits.
//
// Bounded spins: a workgroup that waits 2 s for an epoch writes a give-up code and the whole grid drains (pers_wait).
#include "gnode_bwd.h"
#include "gnode_common.h"
#include "gnode_generic.h"
#include "gnode_head64.h"
#include "gnode_pers64_dev.h"
#include "gnode_persg.h"
#include <algorithm>
#include <vector>

#ifndef GN_PERSG_NB
#define GN_PERSG_NB 5          // batches of 8 neighbour rows in flight per lane, forward
#endif
#ifndef GN_PERSG_BNB
#define GN_PERSG_BNB 2         // ... per table, adjoint (two tables through one id list)
#endif
static_assert(BWD_NWG >= 256, "a persistent workgroup writes partial-gradient slot blockIdx.x");

struct PersgArgs {
    const int* rowptr; const int* col; const int* hubidx; const int* hub_seg_ptr; const int* seg_lo; const int* seg_hi;
    const int* rowmap;                   // [wps][G] lane-group slot -> node of the sample, -1 for padding (gnode_graph_s::pgmap)
    int n, wgs, wps, segcap, idcap; unsigned rows;
    const float* Y0;                     // forward: [3][rows][H] state at grid point 0
    float* T0; float* T1;                // gather tables (forward: Z_I ping-pong)
    float* Q0; float* Q1;                // adjoint: q ping-pong
    const float* W; const float* bias; const float* beta; const float* gamma;
    const float* w3; const float* b3; const float* w2; const float* b2;
    float* S; float* I; float* R;        // forward outputs / adjoint: cotangents gS, gI, gR
    float* sol;                          // trajectory [G][4][rows][H] (forward: written when non-null; adjoint: read)
    const float* ZS0;                    // adjoint: Z_S(y_{G-1}) rows
    float* a; float* part;               // adjoint state [3][rows][H], partial-gradient slots
    PersCtl* ctl;
    PersSched sched;                     // forward: dt[k], slot[k] of step k; adjoint: dt[i-1] at [i], slot of grid point i-1 at [i]
};

// LDS carve-up (floats): W^T [H*H] | W [H*H] | ids [idcap] | partials [2][segcap][H] | segment ids [segcap][32] |
// segment list [segcap][2] | meta [8]
struct PgLds {
    float* Wt; float* Wl; unsigned* IDS; float* HP0; float* HP1; unsigned* HI; int* HS; int* meta;
};
template <int H>
__device__ __forceinline__ PgLds pg_carve(float* lds, int idcap, int segcap) {
    PgLds L;
    L.Wt = lds; L.Wl = L.Wt + H * H; L.IDS = (unsigned*)(L.Wl + H * H);
    L.HP0 = (float*)(L.IDS + idcap); L.HP1 = L.HP0 + (size_t)segcap * H;
    L.HI = (unsigned*)(L.HP1 + (size_t)segcap * H); L.HS = (int*)(L.HI + (size_t)segcap * 32); L.meta = L.HS + 2 * segcap;
    return L;
}
static size_t pg_lds_bytes(int H, int idcap, int segcap) {
    const size_t need = sizeof(float) * ((size_t)2 * H * H + idcap + (size_t)2 * segcap * H + (size_t)segcap * 34 + 8);
    // >= 84 KB: one workgroup per CU; the adjoint's final reduction borrows 64 KB behind the two weight copies
    return std::max<size_t>(need, std::max<size_t>(84 * 1024, sizeof(float) * 2 * H * H + 64 * 1024 + 64));
}

// What a lane group knows about its row, staged once per launch.
struct PgRow { bool inrow; unsigned r, off_b; int cnt, hub_n, hub_base; unsigned estart; };

// Stage W (both orientations), the rows' neighbour ids and the hub segments' ids in LDS.  Returns the hub items of the workgroup.
template <int LPR>
__device__ __forceinline__ int pg_stage(const PersgArgs& a, const PgLds& L, PgRow& row) {
    constexpr int H = 4 * LPR;
    const int GPW = (int)blockDim.x / LPR;
    const int tid = threadIdx.x, sub = tid % LPR, grp = tid / LPR;
    for (int idx = tid; idx < H * H; idx += (int)blockDim.x) { const float v = a.W[idx]; L.Wl[idx] = v; L.Wt[(idx % H) * H + idx / H] = v; }
    if (tid < 8) L.meta[tid] = 0;
    __syncthreads();
    // workgroup w serves sample w / wps; its lane groups own the nodes the graph's row map deals to local workgroup w % wps
    const unsigned b = blockIdx.x / (unsigned)a.wps;
    const int node = a.rowmap[(size_t)(blockIdx.x - b * (unsigned)a.wps) * GPW + grp];
    row.inrow = node >= 0;
    row.r = b * (unsigned)a.n + (unsigned)(row.inrow ? node : 0);
    row.off_b = (row.r * (unsigned)H + 4u * sub) * 4u;
    int start = 0, s0 = 0;
    row.cnt = 0; row.hub_n = 0; row.hub_base = 0; row.estart = 0;
    if (row.inrow) {
        const int hub = a.hubidx ? a.hubidx[node] : -1;
        if (hub < 0) { start = a.rowptr[node]; row.cnt = a.rowptr[node + 1] - start; }
        else { s0 = a.hub_seg_ptr[hub]; row.hub_n = a.hub_seg_ptr[hub + 1] - s0; }
    }
    if (sub == 0) {
        if (row.cnt) row.estart = (unsigned)atomicAdd(&L.meta[0], row.cnt);       // placement in LDS only: no effect on any sum
        if (row.hub_n) row.hub_base = atomicAdd(&L.meta[1], row.hub_n);
    }
    row.estart = (unsigned)__shfl((int)row.estart, 0, LPR);
    row.hub_base = __shfl(row.hub_base, 0, LPR);
    for (int e = sub; e < row.cnt; e += LPR) L.IDS[row.estart + e] = (b * (unsigned)a.n + (unsigned)a.col[start + e]) * (unsigned)(H * 4);
    for (int j = sub; j < row.hub_n; j += LPR) { L.HS[2 * (row.hub_base + j)] = s0 + j; L.HS[2 * (row.hub_base + j) + 1] = (int)(b * (unsigned)a.n); }
    __syncthreads();
    const int items = L.meta[1];
    for (int it = grp; it < items; it += GPW) {
        const int seg = L.HS[2 * it], lo = a.seg_lo[seg], hi = a.seg_hi[seg];
        const unsigned bn = (unsigned)L.HS[2 * it + 1];
        for (int e = sub; e < 32; e += LPR) L.HI[it * 32 + e] = (lo + e < hi) ? (bn + (unsigned)a.col[lo + e]) * (unsigned)(H * 4) : PS_OOB;
    }
    __syncthreads();
    return items;
}

#define PG_ADD(A, V) A.x += V.x; A.y += V.y; A.z += V.z; A.w += V.w;

// Sum of the `cnt` table rows whose byte offsets sit at ids[0 .. cnt), ascending, up to 8 NB in flight.  Loads are issued in
// batches of 8 as far as the LONGEST row of the wave needs (cmax, wave-uniform): a load instruction costs the CU's texture
// unit 16 cycles whatever its lanes fetch, and with one wave per SIMD both the instruction count and the round trips are
// exposed.  Slots past a row's end carry PS_OOB (answered with 0 by the buffer's range check, no memory access; adding +0
// leaves the sum's bits alone).
template <int NB>
__device__ __forceinline__ void pg_sum1(float4& acc, rsrc_t tab, const unsigned* ids, int cnt, int cmax, unsigned lane_b) {
    for (int e0 = 0; e0 < cmax; e0 += 8 * NB) {
        float4 v[8 * NB];
        const int nb = (cmax - e0 + 7) >> 3;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < nb) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const int e = e0 + 8 * j + k; v[8 * j + k] = pers_ld<16>(tab, (e < cnt ? ids[e] : PS_OOB) + lane_b); }
            }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < nb) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { PG_ADD(acc, v[8 * j + k]) }
            }
    }
}
template <int NB>
__device__ __forceinline__ void pg_sum2(float4& a0, float4& a1, rsrc_t t0, rsrc_t t1, const unsigned* ids, int cnt, int cmax, unsigned lane_b) {
    for (int e0 = 0; e0 < cmax; e0 += 8 * NB) {
        float4 u[8 * NB], v[8 * NB];
        const int nb = (cmax - e0 + 7) >> 3;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < nb) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = e0 + 8 * j + k;
                    const unsigned o = (e < cnt ? ids[e] : PS_OOB) + lane_b;
                    u[8 * j + k] = pers_ld<16>(t0, o); v[8 * j + k] = pers_ld<16>(t1, o);
                }
            }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < nb) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { PG_ADD(a0, u[8 * j + k]) PG_ADD(a1, v[8 * j + k]) }
            }
    }
}
__device__ __forceinline__ int pg_wave_max(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

// wave 0 waits for `epoch`, everyone learns whether it gave up
__device__ __forceinline__ bool pg_barrier(PersCtl* ctl, int wgs, unsigned epoch, int* meta) {
    if (threadIdx.x < 64) {
        if (!pers_wait(ctl->flags, wgs, epoch, ctl->error, threadIdx.x) && threadIdx.x == 0) meta[2] = 1;
    }
    __syncthreads();
    return meta[2] == 0;
}
__device__ __forceinline__ void pg_publish(PersCtl* ctl, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains, then the barrier, then ONE flag
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(ctl->flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// --------------------------------------------------------------------------- forward: all Euler steps in one launch
template <int LPR>
__global__ __launch_bounds__(256) void k_persg(const PersgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int H = 4 * LPR;
    const int GPW = (int)blockDim.x / LPR;                  // 1 .. 4 waves per workgroup (the plan's choice)
    const PgLds L = pg_carve<H>(lds, a.idcap, a.segcap);
    PgRow row;
    const int items = pg_stage<LPR>(a, L, row);
    const int cmax = pg_wave_max(row.cnt);            // the wave's longest ordinary row
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const unsigned tbytes = a.rows * (unsigned)(H * 4);
    const rsrc_t tab[2] = {pers_rsrc(a.T0, tbytes), pers_rsrc(a.T1, tbytes)};
    const unsigned lane_b = 16u * sub;
    const size_t slab = (size_t)a.rows * H, off = (size_t)row.r * H + 4 * sub;
    const float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 yS = z0, yI = z0, yR = z0;
    float nb = 0.f, gm = 0.f;
    if (row.inrow) { yS = ld4(a.Y0 + off); yI = ld4(a.Y0 + slab + off); yR = ld4(a.Y0 + 2 * slab + off); nb = -a.beta[row.r]; gm = a.gamma[row.r]; }
    const float4 bias4 = ld4(a.bias + 4 * sub);
    // the read-out head's weights live in registers: with one wave per SIMD a global load inside the step is exposed latency
    float4 w3r[4];
    float b3r[4], w2r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { w3r[k] = ld4(a.w3 + (size_t)k * H + 4 * sub); b3r[k] = a.b3[k]; w2r[k] = a.w2[k]; }
    const float b2r = a.b2[0];
    float4 zmine = row.inrow ? ld4(a.T0 + off) : z0;      // the row's own Z_I(y_k): table 0 at step 0, carried in registers afterwards
    const int n_steps = a.sched.n_steps;
#ifdef GN_PERS_PROF
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_amdgcn_s_memrealtime();
#endif
    for (int k = 0; k < n_steps; ++k) {
        // Z_S(y_k) needs nothing from the other workgroups: it runs under the barrier's flight
        const float4 zs = group_mlp<LPR>(yS, L.Wt, bias4, sub, true, H);
        PS_STAMP(0)
        if (k > 0 && !pg_barrier(a.ctl, a.wgs, (unsigned)k, L.meta)) return;
        PS_STAMP(1)
        const rsrc_t t = tab[k & 1];
        float4 ai = z0;
        if (items > 0) {                          // workgroup-uniform
            for (int it = grp; it < items; it += GPW) {
                float4 s = z0;
                pg_sum1<4>(s, t, L.HI + it * 32, 32, 32, lane_b);
                st4(L.HP0 + (size_t)it * H + 4 * sub, s);
            }
            __syncthreads();
            for (int j = 0; j < row.hub_n; ++j) { const float4 u = ld4(L.HP0 + (size_t)(row.hub_base + j) * H + 4 * sub); PG_ADD(ai, u) }
        }
        PS_STAMP(2)
        pg_sum1<GN_PERSG_NB>(ai, t, L.IDS + row.estart, row.cnt, cmax, lane_b);
        PS_STAMP(3)
        const float dt = a.sched.dt[k];
        float4 dS, dI, dR;
        dS.x = nb * (ai.x * zs.x); dS.y = nb * (ai.y * zs.y); dS.z = nb * (ai.z * zs.z); dS.w = nb * (ai.w * zs.w);
        dR.x = gm * zmine.x; dR.y = gm * zmine.y; dR.z = gm * zmine.z; dR.w = gm * zmine.w;
        dI.x = -dS.x - dR.x; dI.y = -dS.y - dR.y; dI.z = -dS.z - dR.z; dI.w = -dS.w - dR.w;
        yS.x += dt * dS.x; yS.y += dt * dS.y; yS.z += dt * dS.z; yS.w += dt * dS.w;
        yI.x += dt * dI.x; yI.y += dt * dI.y; yI.z += dt * dI.z; yI.w += dt * dI.w;
        yR.x += dt * dR.x; yR.y += dt * dR.y; yR.z += dt * dR.z; yR.w += dt * dR.w;
        if (k + 1 < n_steps) {
            const float4 zn = group_mlp<LPR>(yI, L.Wt, bias4, sub, true, H);      // Z_I of the next step -> the other table
            if (row.inrow) pers_st<16>(tab[(k + 1) & 1], row.off_b, zn);
            PS_STAMP(4)
            pg_publish(a.ctl, (unsigned)k + 1u);
            PS_STAMP(5)
            zmine = zn;
        }
        // behind the flag: the trajectory point and the read-out
        if (a.sol && row.inrow) {
            float* Yo = a.sol + (size_t)(k + 1) * 4 * slab;
            st4(Yo + off, yS); st4(Yo + slab + off, yI); st4(Yo + 2 * slab + off, yR);
        }
        const int slot = a.sched.slot[k];
        if (slot >= 0) {
            float pS, pI, pR;
            readout_row_regs<LPR>(yS, yI, yR, w3r, b3r, w2r, b2r, pS, pI, pR);
            if (sub == 0 && row.inrow) { const size_t o = (size_t)slot * a.rows + row.r; a.S[o] = pS; a.I[o] = pI; a.R[o] = pR; }
        }
        PS_STAMP(6)
    }
#ifdef GN_PERS_PROF
    if (threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < 8; ++i) a.ctl->prof[i] = prof[i];
#endif
}

// --------------------------------------------------------------------------- adjoint: intervals G-1 .. 1 in one launch
template <int LPR>
__device__ __forceinline__ float4 pg_lin(float4 x, const float* __restrict__ M, int sub) {      // sum_k x_k M[k][4 sub ..]
    constexpr int H = 4 * LPR;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int kk = 0; kk < LPR; ++kk) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float xk = __shfl(xv[c], kk, LPR);
            const float4 w = ld4(M + (size_t)(4 * kk + c) * H + 4 * sub);
            acc.x = fmaf(xk, w.x, acc.x); acc.y = fmaf(xk, w.y, acc.y); acc.z = fmaf(xk, w.z, acc.z); acc.w = fmaf(xk, w.w, acc.w);
        }
    }
    return acc;
}
__device__ __forceinline__ float pg_sig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <int LPR>
__global__ __launch_bounds__(256) void k_persg_bwd(const PersgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int H = 4 * LPR;
    const int GPW = (int)blockDim.x / LPR;                  // 1 .. 4 waves per workgroup (the plan's choice)
    const PgLds L = pg_carve<H>(lds, a.idcap, a.segcap);
    PgRow row;
    const int items = pg_stage<LPR>(a, L, row);
    const int cmax = pg_wave_max(row.cnt);            // the wave's longest ordinary row
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const unsigned tbytes = a.rows * (unsigned)(H * 4);
    const rsrc_t zt[2] = {pers_rsrc(a.T0, tbytes), pers_rsrc(a.T1, tbytes)};
    const rsrc_t qt[2] = {pers_rsrc(a.Q0, tbytes), pers_rsrc(a.Q1, tbytes)};
    const unsigned lane_b = 16u * sub;
    const size_t slab = (size_t)a.rows * H, off = (size_t)row.r * H + 4 * sub;
    const float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int G = a.sched.n_steps + 1;
    float4 aS = z0, aI = z0, aR = z0, zs0 = z0, zi0 = z0, yS = z0, yI = z0;
    float bt = 0.f, gm = 0.f;
    const float4 bias4 = ld4(a.bias + 4 * sub);
    float4 w3v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w3v[k] = ld4(a.w3 + (size_t)k * H + 4 * sub);
    HeadAcc hacc;
#pragma unroll
    for (int k = 0; k < 4; ++k) { hacc.dw3[k] = z0; hacc.db3[k] = 0.f; hacc.dw2[k] = 0.f; }
    hacc.db2 = 0.f;
    {
        // The sweep's start-up, which used to be four launches (beta / gamma extraction, the head's VJP at the last grid point,
        // Z(y_{G-1}), q): the adjoint starts at dL/dsol[G-1], the first gather tables are this workgroup's rows of Z_I(y_{G-1})
        // and q = beta (a_I - a_S) Z_S(y_{G-1}), published under epoch 1 (every interval waits for its epoch, the first included).
        const int slot_last = a.sched.slot[0];
        float4 y[3] = {z0, z0, z0};
        float gout[3] = {0.f, 0.f, 0.f};
        if (row.inrow) {
            const float4 bg = ld4(a.sol + 3 * slab + (size_t)row.r * H);          // sol[0]'s 4th slab: beta, gamma, ... (ode_nn_ngraph_sim.py:149-168)
            bt = bg.x; gm = bg.y;
            const float* Yl = a.sol + (size_t)(G - 1) * 4 * slab;
            y[0] = ld4(Yl + off); y[1] = ld4(Yl + slab + off);
            if (slot_last >= 0) {
                y[2] = ld4(Yl + 2 * slab + off);
                const size_t o = (size_t)slot_last * a.rows + row.r;
                gout[0] = a.S[o]; gout[1] = a.I[o]; gout[2] = a.R[o];
            }
        }
        if (slot_last >= 0) head_vjp64<LPR>(y, gout, w3v, a.b3, a.w2, a.b2, aS, aI, aR, hacc);
        yS = y[0]; yI = y[1];
        zs0 = group_mlp<LPR>(yS, L.Wt, bias4, sub, true, H);                       // (the start-up launch's arithmetic: bias first)
        zi0 = group_mlp<LPR>(yI, L.Wt, bias4, sub, true, H);
        if (row.inrow) {
            pers_st<16>(zt[0], row.off_b, zi0);
            pers_st<16>(qt[0], row.off_b, make_float4(bt * (aI.x - aS.x) * zs0.x, bt * (aI.y - aS.y) * zs0.y,
                                                      bt * (aI.z - aS.z) * zs0.z, bt * (aI.w - aS.w) * zs0.w));
        }
        pg_publish(a.ctl, 1u);
    }
    float accW[4][H];                    // gW rows 4 sub .. 4 sub + 3 (dt folded in), all H columns
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < H; ++k) accW[j][k] = 0.f;
    float4 accb = z0;
#ifdef GN_PERS_PROF
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_amdgcn_s_memrealtime();
#endif
    for (int i = G - 1; i >= 1; --i) {
        const int cur = (G - 1 - i) & 1;
        const float dt = a.sched.dt[i];
        const int slot = a.sched.slot[i];
        // what the interval needs from the forward's trajectory travels under the barrier's flight
        float4 y[3] = {z0, z0, z0};
        float gout[3] = {0.f, 0.f, 0.f};
        if (row.inrow) {
            const float* Yp = a.sol + (size_t)(i - 1) * 4 * slab;
            y[0] = ld4(Yp + off); y[1] = ld4(Yp + slab + off);
            if (slot >= 0) {
                y[2] = ld4(Yp + 2 * slab + off);
                const size_t o = (size_t)slot * a.rows + row.r;
                gout[0] = a.S[o]; gout[1] = a.I[o]; gout[2] = a.R[o];
            }
        }
        PS_STAMP(0)
        if (!pg_barrier(a.ctl, a.wgs, (unsigned)(G - i), L.meta)) return;
        PS_STAMP(1)
        float4 ai = z0, gq = z0;
        if (items > 0) {
            for (int it = grp; it < items; it += GPW) {
                float4 s0 = z0, s1 = z0;
                pg_sum2<2>(s0, s1, zt[cur], qt[cur], L.HI + it * 32, 32, 32, lane_b);
                st4(L.HP0 + (size_t)it * H + 4 * sub, s0); st4(L.HP1 + (size_t)it * H + 4 * sub, s1);
            }
            __syncthreads();
            for (int j = 0; j < row.hub_n; ++j) {
                const float4 u = ld4(L.HP0 + (size_t)(row.hub_base + j) * H + 4 * sub), v = ld4(L.HP1 + (size_t)(row.hub_base + j) * H + 4 * sub);
                PG_ADD(ai, u) PG_ADD(gq, v)
            }
        }
        PS_STAMP(2)
        pg_sum2<GN_PERSG_BNB>(ai, gq, zt[cur], qt[cur], L.IDS + row.estart, row.cnt, cmax, lane_b);
        PS_STAMP(3)
        float4 dS, dI;
#define PG_DPRE(c)                                                         \
        {                                                                  \
            const float v = bt * (aI.c - aS.c);                            \
            dS.c = (v * ai.c) * (zs0.c * (1.0f - zs0.c));                  \
            dI.c = (gq.c + gm * (aR.c - aI.c)) * (zi0.c * (1.0f - zi0.c)); \
        }
        PG_DPRE(x) PG_DPRE(y) PG_DPRE(z) PG_DPRE(w)
#undef PG_DPRE
        // g_Y = dpre W, a += dt g_Y
        const float4 uS = pg_lin<LPR>(dS, L.Wl, sub), uI = pg_lin<LPR>(dI, L.Wl, sub);
        aS.x += dt * uS.x; aS.y += dt * uS.y; aS.z += dt * uS.z; aS.w += dt * uS.w;
        aI.x += dt * uI.x; aI.y += dt * uI.y; aI.z += dt * uI.z; aI.w += dt * uI.w;
        // dL/dsol[i-1] through the head
        if (slot >= 0) head_vjp64<LPR>(y, gout, w3v, a.b3, a.w2, a.b2, aS, aI, aR, hacc);
        // Z(y_{i-1}) and q: the tables the next interval gathers
        float4 zs = z0, zi = z0;
        if (i > 1) {
            zs = pg_lin<LPR>(y[0], L.Wt, sub); zi = pg_lin<LPR>(y[1], L.Wt, sub);
            zs = make_float4(pg_sig(zs.x + bias4.x), pg_sig(zs.y + bias4.y), pg_sig(zs.z + bias4.z), pg_sig(zs.w + bias4.w));
            zi = make_float4(pg_sig(zi.x + bias4.x), pg_sig(zi.y + bias4.y), pg_sig(zi.z + bias4.z), pg_sig(zi.w + bias4.w));
            if (row.inrow) {
                pers_st<16>(zt[cur ^ 1], row.off_b, zi);
                pers_st<16>(qt[cur ^ 1], row.off_b, make_float4(bt * (aI.x - aS.x) * zs.x, bt * (aI.y - aS.y) * zs.y,
                                                                bt * (aI.z - aS.z) * zs.z, bt * (aI.w - aS.w) * zs.w));
            }
            PS_STAMP(4)
            pg_publish(a.ctl, (unsigned)(G - i + 1));
            PS_STAMP(5)
        }
        // behind the flag: gW += dt dpre^T y (S and I parts), gb += dt dpre
        {
            const float dv[2][4] = {{dt * dS.x, dt * dS.y, dt * dS.z, dt * dS.w}, {dt * dI.x, dt * dI.y, dt * dI.z, dt * dI.w}};
            const float yv[2][4] = {{yS.x, yS.y, yS.z, yS.w}, {yI.x, yI.y, yI.z, yI.w}};
#pragma unroll
            for (int X = 0; X < 2; ++X)
#pragma unroll
                for (int kk = 0; kk < LPR; ++kk)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float yk = __shfl(yv[X][c], kk, LPR);
#pragma unroll
                        for (int j = 0; j < 4; ++j) accW[j][4 * kk + c] = fmaf(dv[X][j], yk, accW[j][4 * kk + c]);
                    }
            accb.x += dv[0][0] + dv[1][0]; accb.y += dv[0][1] + dv[1][1]; accb.z += dv[0][2] + dv[1][2]; accb.w += dv[0][3] + dv[1][3];
        }
        zs0 = zs; zi0 = zi; yS = y[0]; yI = y[1];
        PS_STAMP(6)
    }
#ifdef GN_PERS_PROF
    if (threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < 8; ++i) a.ctl->prof[i] = prof[i];
#endif
    if (row.inrow) { st4(a.a + off, aS); st4(a.a + slab + off, aI); st4(a.a + 2 * slab + off, aR); }
    // ---- one reduction per sweep: lane-group accumulators -> this workgroup's partial slot, fixed order
    const PartLayout PL{H};
    float* part = a.part + (size_t)blockIdx.x * PL.total();
    float* red = L.Wl + H * H;                                  // 64 KB behind the weight copies (ids / partials are done with)
    const int NR = min(16384 / (H * H), GPW);                   // lane groups per round
    for (int g0 = 0; g0 < GPW; g0 += NR) {
        __syncthreads();
        if (grp >= g0 && grp < g0 + NR) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < H; ++k) red[(size_t)(grp - g0) * H * H + (4 * sub + j) * H + k] = accW[j][k];
        }
        __syncthreads();
        const int ng = min(NR, GPW - g0);
        for (int e = threadIdx.x; e < H * H; e += (int)blockDim.x) {
            float s = 0.f;
            for (int gi = 0; gi < ng; ++gi) s += red[(size_t)gi * H * H + e];
            part[PL.oW() + e] += s;
        }
    }
    __syncthreads();
    {
        // gb, then the head's parameter gradients: [GPW][ne] rows, column sums in lane-group order
        constexpr int ne = 5 * H + 12;                          // 4H + 9 head values, then the H gb values; rows 16-B aligned
        float* mine = red + (size_t)grp * ne;
#pragma unroll
        for (int k = 0; k < 4; ++k) st4(mine + k * H + 4 * sub, hacc.dw3[k]);
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { mine[4 * H + k] = hacc.db3[k]; mine[4 * H + 4 + k] = hacc.dw2[k]; }
            mine[4 * H + 8] = hacc.db2;
        }
        st4(mine + 4 * H + 12 + 4 * sub, accb);
        __syncthreads();
        for (int e = threadIdx.x; e < 5 * H + 12; e += (int)blockDim.x) {
            if (e >= 4 * H + 9 && e < 4 * H + 12) continue;
            float s = 0.f;
            for (int gi = 0; gi < GPW; ++gi) s += red[(size_t)gi * ne + e];
            if (e < 4 * H + 9) part[PL.ow3() + e] += s;
            else part[PL.ob() + (e - 4 * H - 12)] += s;
        }
    }
}

// --------------------------------------------------------------------------- host: graph statistics, plan, launchers
static int persg_vi(int H) { return H == 8 ? 0 : H == 16 ? 1 : H == 32 ? 2 : -1; }

// Row maps for workgroups of 1 .. 4 waves (a wave holds 64 / LPR rows: 32 at H = 8): rows are dealt longest first, round-robin,
// to the sample's workgroups -- real node numberings put the big nodes next to each other, and one workgroup owning them all
// would need their segments' ids in its LDS and set every step's duration.  (Edge-balanced dealing -- each row to the workgroup
// with the fewest edges so far, what gn_pers64_build does -- measured worse here: heavy-tailed wiki-vote size 0.32 -> 0.37 ms;
// a hub's segments are spread over the workgroup's lane groups, an ordinary row is one lane group's serial chain, and the
// snake gives every workgroup the same number of rows from every length class.)  Per variant: the most neighbour ids of ordinary rows and the most hub segments one workgroup
// has to stage.  All maps of a graph live in ONE device allocation.
int gn_persg_build(gnode_graph_s* g, const int32_t* rowptr_host) {
    const int n = g->n;
    g->pgmap = nullptr;
    for (int vi = 0; vi < 3; ++vi) for (int nw = 0; nw < 4; ++nw) { g->pgoff[vi][nw] = -1; g->pgids[vi][nw] = g->pgsegs[vi][nw] = 0; }
    if ((long)n > 256L * 128) return 0;                      // never fits one resident grid
    // every row, longest first (hub rows lead), dealt in snake order: each workgroup gets its share of the long rows AND of the
    // neighbour ids -- what a step waits for is the busiest workgroup's gather
    std::vector<int> order(n);
    auto deg = [&](int i) { return rowptr_host[i + 1] - rowptr_host[i]; };
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return deg(x) > deg(y); });
    std::vector<int32_t> all;
    for (int vi = 0; vi < 3; ++vi)
    for (int nw = 1; nw <= 4; ++nw) {
        const int gpw = (32 >> vi) * nw, wps = (n + gpw - 1) / gpw;
        if (wps > 256) continue;
        std::vector<std::vector<int>> own(wps);
        for (size_t h = 0; h < order.size(); ++h) {            // snake order: 0 .. wps-1, wps-1 .. 0, ...
            const size_t lap = h / wps, pos = h % wps;
            own[(lap & 1) ? wps - 1 - pos : pos].push_back(order[h]);
        }
        bool ok = true;
        for (int w = 0; w < wps; ++w) if ((int)own[w].size() > gpw) ok = false;     // (cannot happen: wps * gpw >= n)
        if (!ok) continue;
        const size_t off = all.size();
        all.resize(off + (size_t)wps * gpw, -1);
        long best_i = 0, best_s = 0;
        for (int ww = 0; ww < wps; ++ww) {
            long ci = 0, cs = 0;
            for (size_t k = 0; k < own[ww].size(); ++k) {
                const int i = own[ww][k], d = deg(i);
                all[off + (size_t)ww * gpw + k] = i;
                if (g->n_hub > 0 && d > GN_HUB_T) cs += (d + HUB_SEG - 1) / HUB_SEG; else ci += d;
            }
            best_i = std::max(best_i, ci); best_s = std::max(best_s, cs);
        }
        g->pgoff[vi][nw - 1] = (int32_t)off;
        g->pgids[vi][nw - 1] = (int32_t)std::min<long>(best_i, 1L << 30);
        g->pgsegs[vi][nw - 1] = (int32_t)std::min<long>(best_s, 1L << 30);
    }
    if (all.empty()) return 0;
    GN_HIP(hipMalloc(&g->pgmap, all.size() * sizeof(int32_t)));
    GN_HIP(hipMemcpy(g->pgmap, all.data(), all.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return 0;
}
void gn_persg_free(gnode_graph_s* g) {
    if (g->pgmap) (void)hipFree(g->pgmap);
    g->pgmap = nullptr;
}

// The fewest rows per workgroup (most CUs) that still leaves every workgroup of the batch resident, one per CU: what a step
// waits for is its busiest CU's gather, and that is bound by the CU's rate of cache-line requests (a 32-byte row is a request
// of its own: ~1.3 ns each measured, 128 rows x 22 neighbours = 3.7 us) -- so spread the rows over as many CUs as there are.
bool gn_persg_plan(const gnode_graph_s* g, long rows, int H, int n_steps, PersgPlan* p) {
    const int vi = persg_vi(H);
    if (vi < 0 || n_steps < 1 || n_steps > 127 || !g->pgmap) return false;
    if ((long)rows * H * 4 >= (1L << 31) - (1L << 17)) return false;    // 32-bit table offsets below PS_OOB
    for (int nw = 1; nw <= 4; ++nw) {
        if (g->pgoff[vi][nw - 1] < 0) continue;
        const int gpw = (32 >> vi) * nw, wps = (g->n + gpw - 1) / gpw;
        const long wgs = (rows / g->n) * wps;
        if (wgs > std::min(g->num_cu, 256)) continue;              // one workgroup per CU, all resident; pers_wait sweeps 256 flags
        const int idcap = (g->pgids[vi][nw - 1] + 3) & ~3, segcap = std::max(4, (g->pgsegs[vi][nw - 1] + 3) & ~3);
        const size_t need = sizeof(float) * ((size_t)2 * H * H + idcap + (size_t)2 * segcap * H + (size_t)segcap * 34 + 8);
        if (need > 150 * 1024) continue;
        if (p) { p->wgs = (int)wgs; p->wps = wps; p->nw = nw; p->map_off = g->pgoff[vi][nw - 1]; p->idcap = idcap; p->segcap = segcap; p->lds = pg_lds_bytes(H, idcap, segcap); }
        return true;
    }
    return false;
}

int gn_persg_set_attributes() {
    const int mx = 160 * 1024;
    GN_HIP(hipFuncSetAttribute((const void*)k_persg<2>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    GN_HIP(hipFuncSetAttribute((const void*)k_persg<4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    GN_HIP(hipFuncSetAttribute((const void*)k_persg<8>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    GN_HIP(hipFuncSetAttribute((const void*)k_persg_bwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    GN_HIP(hipFuncSetAttribute((const void*)k_persg_bwd<4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    GN_HIP(hipFuncSetAttribute((const void*)k_persg_bwd<8>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    return 0;
}

static void persg_common(PersgArgs& a, const gnode_graph_s* g, const PersgPlan& pl, long rows, int H, const gnode_params* p, void* ctl) {
    a.rowptr = g->rowptr; a.col = g->col; a.hubidx = g->n_hub > 0 ? g->hubidx : nullptr; a.hub_seg_ptr = g->hub_seg_ptr;
    a.seg_lo = g->seg_lo; a.seg_hi = g->seg_hi;
    (void)H;
    a.rowmap = g->pgmap + pl.map_off;
    a.n = g->n; a.wgs = pl.wgs; a.wps = pl.wps; a.segcap = pl.segcap; a.idcap = pl.idcap; a.rows = (unsigned)rows;
    a.W = p->odefunc_linear_weight; a.bias = p->odefunc_linear_bias;
    a.w3 = p->linear3_weight; a.b3 = p->linear3_bias; a.w2 = p->linearS2_weight; a.b2 = p->linearS2_bias;
    a.ctl = (PersCtl*)ctl;
    a.Y0 = nullptr; a.T0 = a.T1 = a.Q0 = a.Q1 = nullptr; a.S = a.I = a.R = nullptr; a.sol = nullptr; a.ZS0 = nullptr; a.a = a.part = nullptr;
    a.beta = a.gamma = nullptr;
}

int gn_launch_persg(const gnode_graph_s* g, const PersgPlan& pl, long rows, int H, const float* Y0, float* Z0, float* Z1,
                    const float* beta, const float* gamma, const float* dt_host, const int* slot_host, int n_steps,
                    const gnode_params* p, float* S, float* I, float* R, float* sol, void* ctl, bool ctl_is_zero, hipStream_t st) {
    PersgArgs a;
    persg_common(a, g, pl, rows, H, p, ctl);
    a.Y0 = Y0; a.T0 = Z0; a.T1 = Z1; a.beta = beta; a.gamma = gamma; a.S = S; a.I = I; a.R = R; a.sol = sol;
    a.sched.n_steps = n_steps;
    for (int k = 0; k < n_steps; ++k) { a.sched.dt[k] = dt_host[k]; a.sched.slot[k] = (short)slot_host[k]; }
    if (!ctl_is_zero)                                              // flags and the give-up word: zeroed before EVERY launch (by the prologue here)
        if (int e = gn_pers64_zero_ctl(ctl, st)) return e;
    const dim3 grid((unsigned)pl.wgs);
    if (H == 8) hipLaunchKernelGGL(k_persg<2>, grid, dim3(64 * pl.nw), pl.lds, st, a);
    else if (H == 16) hipLaunchKernelGGL(k_persg<4>, grid, dim3(64 * pl.nw), pl.lds, st, a);
    else hipLaunchKernelGGL(k_persg<8>, grid, dim3(64 * pl.nw), pl.lds, st, a);
    GN_LAUNCH_CHECK();
    return 0;
}

int gn_launch_persg_bwd(const gnode_graph_s* g, const PersgPlan& pl, long rows, int H, int G, float* ZI0, float* ZI1, float* Q0, float* Q1,
                        const float* ZS0, const float* sol, const float* beta, const float* gamma, float* a_state, float* part,
                        const float* gS, const float* gI, const float* gR, const gnode_params* p, const float* dt_host,
                        const int* slot_of_prev, void* ctl, bool ctl_is_zero, hipStream_t st) {
    PersgArgs a;
    persg_common(a, g, pl, rows, H, p, ctl);
    a.T0 = ZI0; a.T1 = ZI1; a.Q0 = Q0; a.Q1 = Q1; a.ZS0 = ZS0; a.sol = const_cast<float*>(sol); a.beta = beta; a.gamma = gamma;
    a.a = a_state; a.part = part; a.S = const_cast<float*>(gS); a.I = const_cast<float*>(gI); a.R = const_cast<float*>(gR);
    a.sched.n_steps = G - 1;
    a.sched.dt[0] = 0.f; a.sched.slot[0] = (short)slot_of_prev[0];           // [0]: the output row of the LAST grid point (its head VJP starts the sweep)
    for (int i = 1; i <= G - 1; ++i) { a.sched.dt[i] = dt_host[i - 1]; a.sched.slot[i] = (short)slot_of_prev[i]; }
    if (!ctl_is_zero)
        if (int e = gn_pers64_zero_ctl(ctl, st)) return e;
    const dim3 grid((unsigned)pl.wgs);
    if (H == 8) hipLaunchKernelGGL(k_persg_bwd<2>, grid, dim3(64 * pl.nw), pl.lds, st, a);
    else if (H == 16) hipLaunchKernelGGL(k_persg_bwd<4>, grid, dim3(64 * pl.nw), pl.lds, st, a);
    else hipLaunchKernelGGL(k_persg_bwd<8>, grid, dim3(64 * pl.nw), pl.lds, st, a);
    GN_LAUNCH_CHECK();
    return 0;
}
