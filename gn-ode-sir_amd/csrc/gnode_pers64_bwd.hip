// Adjoint backward over kept activations for mid-size graphs in ONE persistent launch (gfx950): the backward twin of
// gnode_pers64.hip.  Intervals G-2 .. 1 of torchdiffeq's adjoint-Euler sweep (gnode_bwd.hip's header has
// the recurrences) -- the last interval, whose activations the forward never had, stays with the recomputing launch.
//
// Every workgroup owns its rows for the whole sweep (same plan, same row map, same placement by XCC_ID as the forward):
//   * the adjoint state a_S, a_I, a_R of a row, its share of gW (a 64x64 accumulator per 4-wave team), of gb and of the
//     read-out head's gradients stay in REGISTERS across all intervals; W stays in LDS;
//   * per interval the only traffic other workgroups see is the q table: q_{i-1} = beta (a_I - a_S) Z_S(y_{i-1}) rows are
//     stored, the group meets at the flag barrier (protocol: gnode_pers64.hip's header), A q_{i-1} is gathered;
//   * everything else an interval reads -- the forward's kept P_S(y_i), Z_I(y_i), Z_S(y_{i-1}), the trajectory rows y_i
//     (for gW) and y_{i-1} (for the head's VJP), the output cotangents -- depends on no other workgroup and is requested
//     BEFORE the barrier wait, so it lands under it;
//   * only g_Y = dpre W is on the way from the gather to the next q row; gW += dpre^T y runs behind the flag.
// Parameter gradients: each workgroup adds its sums into its own slot of the partial buffer (plain read-modify-write, as
// the per-interval kernels do) at the end; the final fixed-order reduction is unchanged -> bitwise reproducible run to run.
// Numerics: the same VJPs in the same per-row order as k_bwd_kept64; only the order in which rows enter the parameter
// sums differs (tests hold it to the per-interval path at 1e-5 and to the reference-class gradients at 2e-4).
#include "gnode_bwd.h"
#include "gnode_h64.h"
#include "gnode_mfma64.h"
#include "gnode_head64.h"
#include "gnode_pers64.h"
#include "gnode_pers64_dev.h"
#include <algorithm>

struct PersBwdArgs {
    const int* rowhdr; const int* col; const int* rowmap;
    const int* hubslot; const int* segptr; const int* segitem;
    int n, B, b0, lds_slots; unsigned rows;
    PersPlace pp;
    int G;                               // grid points; intervals i = G-2 .. 1 are run here
    float* Q[2];                         // q tables [rows + 1][64]; interval i gathers Q[(G-1-i) & 1]
    const float* sol; const float* keep;
    const float* W; const float* beta; const float* gamma;
    float* a;                            // adjoint state [3][rows][64], in and out
    float* part;                         // partial-gradient slots
    const float* gS; const float* gI; const float* gR;
    const float* w3; const float* b3; const float* w2; const float* b2;
    PersCtl* ctl;
    int fold;                            // 1: the sweep also covers interval G-1, whose adjoint is zero (last grid point not emitted)
    float dt[128];                       // dt[i]: step size of interval i (grid point i-1 -> i)
    short slot[128];                     // slot[i]: output row of grid point i-1, or -1
};

template <int NT, bool SC1ST, bool HUBS>
__global__ __launch_bounds__(256 * NT) void k_pers_bwd64(const PersBwdArgs a) {
    constexpr int STAUX = SC1ST ? 16 : 0;
    constexpr int NM = 6;                                  // neighbour-id registers (16 ids each): every row up to the hub threshold (96)
#ifndef GN_PERS_BWD_DEPTH
#define GN_PERS_BWD_DEPTH 2
#endif
    constexpr int DEPTH = GN_PERS_BWD_DEPTH;               // batches of 8 neighbour rows in flight (the state leaves room for 16)
    constexpr int O_W = 0, O_T = 64 * TS, TEAM_F = 6 * 16 * TS;
    extern __shared__ __attribute__((aligned(16))) float L[];
    __shared__ unsigned sh[4];
    const int team = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int lane = tid & 63, w = tid >> 6, g = lane >> 4, sub = lane & 15;
    const int i16 = lane & 15, kq = lane >> 4;
    if (team == 0) load_W_to_lds<false>(a.W, L + O_W);
    int gl, idx;
    if (!pers_place(a.pp, a.ctl, sh, gl, idx)) return;
    unsigned* const flags = a.ctl->flags + (size_t)gl * a.pp.fstride;
    unsigned* const err = a.ctl->error;
    const int b = a.b0 + gl;                               // this launch's samples: b0 .. b0 + concurrent - 1
    if (b >= a.B) return;

    float* const Dt0 = L + O_T + team * TEAM_F;            // dpre_S | dpre_I | y_S | y_I | g_YS | g_YI, 16 rows each
    float* const Dt1 = Dt0 + 16 * TS;
    float* const Yt0 = Dt1 + 16 * TS;
    float* const Yt1 = Yt0 + 16 * TS;
    float* const Gt0 = Yt1 + 16 * TS;
    float* const Gt1 = Gt0 + 16 * TS;
    const float* const Wl = L + O_W;
    float* const HP = L + O_T + NT * TEAM_F;               // HUBS: segment partials [S][64] | segment ids [S][32] | item lists (gnode_pers64_dev.h)
    const int lr = 4 * w + g, ro = lr * TS + 4 * sub;
    unsigned* const HI = reinterpret_cast<unsigned*>(HP + (size_t)a.lds_slots * 64);
    unsigned* const HLmine = HI + (size_t)a.lds_slots * 32 + (size_t)(team * 16 + lr) * PERS_MAX_ITEMS;
    const unsigned lane_b = 16u * sub;
    const unsigned rows = a.rows;
    const unsigned tbytes = (rows + 1u) * 256u;
    const size_t slab = (size_t)rows * 64;
    const int lgslot = idx * (16 * NT) + team * 16 + lr;
    const int node = a.rowmap[lgslot];
    const bool valid = node >= 0;
    int hs0 = -1, hcnt = 0, it0 = 0, itn = 0;               // the hub row this lane group owns; the segment sums it computes
    if (HUBS) { hs0 = a.hubslot[2 * lgslot]; hcnt = a.hubslot[2 * lgslot + 1]; it0 = a.segptr[2 * lgslot]; itn = a.segptr[2 * lgslot + 1]; }
    const unsigned base = (unsigned)b * (unsigned)a.n;
    const unsigned row = valid ? base + (unsigned)node : 0u;
    const unsigned off = row * 256u + lane_b;
    int start = 0, end = 0;
    unsigned m[NM];
#pragma unroll
    for (int j = 0; j < NM; ++j) m[j] = PS_OOB;
    if (valid) {
        const int* h = a.rowhdr + (size_t)node * 20;
        start = h[0]; end = h[1];
        if (HUBS && hs0 >= 0) end = start;                  // a hub row's sum arrives as segment partials
        const int d = end - start;
        if (sub < d) m[0] = (base + (unsigned)h[4 + sub]) * 256u;
#pragma unroll
        for (int j = 1; j < NM; ++j)
            if (16 * j + sub < d) m[j] = (base + (unsigned)a.col[start + 16 * j + sub]) * 256u;
    }
    const int nbt = pers_batches<NM>(end - start);
    if (HUBS) pers_hub_stage(a.col, a.segitem, it0, itn, base, HI, HLmine, sub);        // (published by the barrier behind the W staging)
    float4 aS = zero4(), aI = zero4(), aR = zero4();
    float bt = 0.f, gm = 0.f;
    if (valid) {
        aS = ld4o(a.a, off); aI = ld4o(a.a + slab, off); aR = ld4o(a.a + 2 * slab, off);
        bt = a.beta[row]; gm = a.gamma[row];
    }
    HeadAcc hacc;
#pragma unroll
    for (int k = 0; k < 4; ++k) { hacc.dw3[k] = zero4(); hacc.db3[k] = 0.f; hacc.dw2[k] = 0.f; }
    hacc.db2 = 0.f;
    f32x4 totW[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) totW[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 totb = zero4();
    const int G = a.G;
    __syncthreads();                                       // W staged

    // rows an interval needs from nobody else: requested ahead of the barrier wait
    struct Pre { float4 ps, zi, ys, yi, zsp, y[3]; float gout[3]; };
    auto prefetch = [&](int i, Pre& q) {
        const float* si = a.sol + (size_t)i * 4 * slab;
        q.ps = ld4so<true>(gn_keep_ps(a.keep, rows, i), off);
        q.zi = ld4so<true>(gn_keep_zi(a.keep, rows, i), off);
        q.ys = ld4so<true>(si, off); q.yi = ld4so<true>(si + slab, off);
        q.zsp = zero4();
        if (i > 1) q.zsp = ld4so<true>(gn_keep_zs(a.keep, rows, i - 1), off);
        const int s = a.slot[i];
        q.y[0] = q.y[1] = q.y[2] = zero4(); q.gout[0] = q.gout[1] = q.gout[2] = 0.f;
        if (s >= 0) {
            const float* sp = a.sol + (size_t)(i - 1) * 4 * slab;
            q.y[0] = ld4so<true>(sp, off); q.y[1] = ld4so<true>(sp + slab, off); q.y[2] = ld4so<true>(sp + 2 * slab, off);
            if (valid) { q.gout[0] = a.gS[(size_t)s * rows + row]; q.gout[1] = a.gI[(size_t)s * rows + row]; q.gout[2] = a.gR[(size_t)s * rows + row]; }
        }
    };
    Pre pre;
    prefetch(G - 2, pre);
    const int fold = a.fold;                               // 1: the sweep starts one interval early (see gnode_backward_f32), every epoch is one later
    if (fold) {
        // interval G-1 with a zero adjoint: only the head's VJP at grid point G-2 and the q row interval G-2 gathers remain
        const int s = a.slot[G - 1];
        if (s >= 0) {
            float4 y[3];
            float gout[3] = {0.f, 0.f, 0.f};
            const float* sp = a.sol + (size_t)(G - 2) * 4 * slab;
            y[0] = ld4so<true>(sp, off); y[1] = ld4so<true>(sp + slab, off); y[2] = ld4so<true>(sp + 2 * slab, off);
            if (valid) { gout[0] = a.gS[(size_t)s * rows + row]; gout[1] = a.gI[(size_t)s * rows + row]; gout[2] = a.gR[(size_t)s * rows + row]; }
            float4 w3v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w3v[q] = ld4g(a.w3 + q * 64 + 4 * sub);
            head_vjp64(y, gout, w3v, a.b3, a.w2, a.b2, aS, aI, aR, hacc);
        }
        const float4 zsp = ld4so<true>(gn_keep_zs(a.keep, rows, G - 2), off);
        if (valid) pers_st<STAUX>(pers_rsrc(a.Q[1], tbytes), off,
                                  make_float4(bt * (aI.x - aS.x) * zsp.x, bt * (aI.y - aS.y) * zsp.y,
                                              bt * (aI.z - aS.z) * zsp.z, bt * (aI.w - aS.w) * zsp.w));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flags + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = G - 2; i >= 1; --i) {
        const int k = G - 2 - i + fold;                    // epochs published so far
        const int cur = (G - 1 - i) & 1;
        const float dt = a.dt[i];
        if (k > 0) {
            bool ok = true;
            if (threadIdx.x < 64) ok = pers_wait(flags, a.pp.wgs, (unsigned)k, err, lane);
            if (threadIdx.x < 64 && !ok) sh[2] = 0u;
            __syncthreads();
            if (sh[2] == 0u) return;
        }
        // ---- A q: the transposed gather (A symmetric: same neighbour lists), ascending column order
        float4 gq = pers_gather<NM, DEPTH>(pers_rsrc(a.Q[cur], tbytes), m, nbt, lane_b, []() {});
        if (HUBS) {
            pers_hub_partials<DEPTH>(pers_rsrc(a.Q[cur], tbytes), itn, HI, HLmine, HP, sub, lane_b);
            __syncthreads();
            if (hs0 >= 0) gq = pers_hub_total(HP, hs0, hcnt, sub);
        }
        const Pre p = pre;
        {
            float4 dS, dI;
#define PB_DP(c)                                                                              \
            {                                                                                 \
                const float v = bt * (aI.c - aS.c);                                           \
                dS.c = v * p.ps.c;                                                            \
                dI.c = valid ? (gq.c + gm * (aR.c - aI.c)) * (p.zi.c * (1.0f - p.zi.c)) : 0.f; \
            }
            PB_DP(x) PB_DP(y) PB_DP(z) PB_DP(w)
#undef PB_DP
            totb.x += dt * (dS.x + dI.x); totb.y += dt * (dS.y + dI.y); totb.z += dt * (dS.z + dI.z); totb.w += dt * (dS.w + dI.w);
            *reinterpret_cast<float4*>(Dt0 + ro) = dS; *reinterpret_cast<float4*>(Dt1 + ro) = dI;
            *reinterpret_cast<float4*>(Yt0 + ro) = p.ys; *reinterpret_cast<float4*>(Yt1 + ro) = p.yi;
        }
        __syncthreads();
        mfma_tile16<false, true>(Dt0, Wl, Gt0, 0.f, w, lane);         // g_Y = dpre W: the only product on the way to the next q row
        mfma_tile16<false, true>(Dt1, Wl, Gt1, 0.f, w, lane);
        __syncthreads();
        {
            const float4 uS = *reinterpret_cast<const float4*>(Gt0 + ro);
            const float4 uI = *reinterpret_cast<const float4*>(Gt1 + ro);
            aS.x += dt * uS.x; aS.y += dt * uS.y; aS.z += dt * uS.z; aS.w += dt * uS.w;
            aI.x += dt * uI.x; aI.y += dt * uI.y; aI.z += dt * uI.z; aI.w += dt * uI.w;
        }
        if (a.slot[i] >= 0) {                                          // the head's VJP at grid point i-1 (uniform per interval)
            float4 w3v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w3v[q] = ld4g(a.w3 + q * 64 + 4 * sub);
            head_vjp64(p.y, p.gout, w3v, a.b3, a.w2, a.b2, aS, aI, aR, hacc);   // padding rows: gout = 0 adds nothing
        }
        if (i > 1) {
            if (valid) pers_st<STAUX>(pers_rsrc(a.Q[cur ^ 1], tbytes), off,
                                      make_float4(bt * (aI.x - aS.x) * p.zsp.x, bt * (aI.y - aS.y) * p.zsp.y,
                                                  bt * (aI.z - aS.z) * p.zsp.z, bt * (aI.w - aS.w) * p.zsp.w));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(flags + idx, (unsigned)k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prefetch(i - 1, pre);                                      // the next interval's own rows, under the barrier
        }
        // ---- behind the flag: gW += dt dpre^T y (contraction over the team's 16 rows)
        {
            f32x4 accW[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) accW[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                const float* D = X ? Dt1 : Dt0;
                const float* Y = X ? Yt1 : Yt0;
#pragma unroll
                for (int s8 = 0; s8 < 4; ++s8) {
                    const int rr = 4 * s8 + kq;
                    const float av = D[rr * TS + 16 * w + i16];
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        accW[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Y[rr * TS + 16 * kt + i16], accW[kt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) totW[kt][reg] += dt * accW[kt][reg];
        }
    }
    // ---- the adjoint state goes back for the encoder's backward; parameter sums into this workgroup's slot
    if (valid) { st4o(a.a, off, aS); st4o(a.a + slab, off, aI); st4o(a.a + 2 * slab, off, aR); }
    const PartLayout PL{64};
    float* part = a.part + (size_t)((b % (BWD_NWG / a.pp.wgs)) * a.pp.wgs + idx) * PL.total();
    for (int t = 0; t < NT; ++t) {                         // the teams' 64x64 accumulators, one team at a time
        __syncthreads();
        if (team == t)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    part[PL.oW() + (16 * w + 4 * kq + reg) * 64 + 16 * kt + i16] += totW[kt][reg];
    }
    __syncthreads();
    constexpr int NE = 5 * 64 + 12;                        // 4*64 + 9 head values, then the 64 gb values
    float* red = L + O_T;                                  // 16 NT lane groups x 332 floats <= the tiles (6 x 16 x 68 per team)
    float* mine = red + (size_t)(threadIdx.x >> 4) * NE;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(mine + q * 64 + 4 * sub) = hacc.dw3[q];
    if (sub == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { mine[256 + q] = hacc.db3[q]; mine[260 + q] = hacc.dw2[q]; }
        mine[264] = hacc.db2;
    }
    *reinterpret_cast<float4*>(mine + 268 + 4 * sub) = totb;
    __syncthreads();
    for (int e = threadIdx.x; e < 268 + 64; e += 256 * NT) {
        if (e >= 265 && e < 268) continue;
        float s = 0.f;
        for (int gi = 0; gi < 16 * NT; ++gi) s += red[(size_t)gi * NE + e];
        if (e < 265) part[PL.ow3() + e] += s;
        else part[PL.ob() + (e - 268)] += s;
    }
}

// --------------------------------------------------------------------------- host
static size_t pers_bwd_lds_bytes(int nt, int partial_slots = 128) {
    const size_t need = sizeof(float) * ((size_t)64 * TS + (size_t)nt * 6 * 16 * TS + (size_t)partial_slots * 96 + (size_t)16 * nt * PERS_MAX_ITEMS);
    return std::max<size_t>(need, 84 * 1024);              // > half of the CU's 160 KB: ONE workgroup per CU
}

int gn_pers_bwd64_set_attributes() {
#define PB_ATTR(N, S) GN_HIP(hipFuncSetAttribute((const void*)k_pers_bwd64<N, S, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pers_bwd_lds_bytes(N))); \
                      GN_HIP(hipFuncSetAttribute((const void*)k_pers_bwd64<N, S, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pers_bwd_lds_bytes(N)));
    PB_ATTR(1, false) PB_ATTR(1, true) PB_ATTR(2, false) PB_ATTR(2, true)
#undef PB_ATTR
    return 0;
}

// intervals G-2 .. 1 over the forward's kept activations; `a` holds the adjoint at grid point G-2 (the head's VJP there
// included), Q[(G-1-(G-2)) & 1] = Q[1] the q table the recomputing launch of interval G-1 left.  Returns the number of
// partial slots it wrote through *slots.
int gn_launch_pers_bwd64(const gnode_graph_s* g, const PersPlan& pl, long rows, int G, float* Q0, float* Q1, const float* sol,
                         const float* keep, const float* W, const float* beta, const float* gamma, float* a, float* part,
                         const float* gS, const float* gI, const float* gR, const gnode_params* p, const float* dt_host,
                         const int* slot_of_prev /* [G]: output row of grid point i-1 for interval i, or -1 */, void* ctl,
                         bool ctl_is_zero, bool fold, int* slots, hipStream_t st) {
    PersBwdArgs x;
    const int vi = pl.nt == 1 ? 0 : 1;
    const bool hubs = g->n_hub > 0;
    x.rowhdr = g->rowhdr; x.col = g->col; x.rowmap = g->persmap[vi];
    x.lds_slots = hubs ? g->perslds[vi] : 0;
    x.hubslot = hubs ? g->pershub[vi] : nullptr; x.segptr = hubs ? g->perssegptr[vi] : nullptr; x.segitem = hubs ? g->perssegitem[vi] : nullptr;
    x.n = g->n; x.B = (int)(rows / g->n); x.rows = (unsigned)rows; x.pp = pers_place_of(pl); x.G = G;
    x.Q[0] = Q0; x.Q[1] = Q1; x.sol = sol; x.keep = keep; x.W = W; x.beta = beta; x.gamma = gamma; x.a = a; x.part = part;
    x.gS = gS; x.gI = gI; x.gR = gR;
    x.w3 = p->linear3_weight; x.b3 = p->linear3_bias; x.w2 = p->linearS2_weight; x.b2 = p->linearS2_bias;
    x.ctl = (PersCtl*)ctl;
    for (int i = 0; i < 128; ++i) { x.dt[i] = 0.f; x.slot[i] = -1; }
    for (int i = 1; i <= G - 2; ++i) { x.dt[i] = dt_host[i - 1]; x.slot[i] = (short)slot_of_prev[i]; }
    x.fold = fold ? 1 : 0;
    if (fold && G - 1 < 128) x.slot[G - 1] = (short)slot_of_prev[G - 1];      // the head's VJP at grid point G-2 belongs to interval G-1
    const bool sc1 = pl.span > 1;
    const dim3 grid((unsigned)(pl.n_xcc * pl.slots));
    // samples are independent: batches beyond what one resident grid holds run as consecutive launches of `concurrent` samples
    for (int b0 = 0; b0 < x.B; b0 += pl.concurrent) {
        x.b0 = b0;
        if (!(ctl_is_zero && b0 == 0))
            if (int e = gn_pers64_zero_ctl(ctl, st)) return e;
#define PB_GO(N, S) { if (hubs) hipLaunchKernelGGL((k_pers_bwd64<N, S, true>), grid, dim3(256 * N), pers_bwd_lds_bytes(N, g->perslds[vi]), st, x); \
                      else hipLaunchKernelGGL((k_pers_bwd64<N, S, false>), grid, dim3(256 * N), pers_bwd_lds_bytes(N, 0), st, x); }
        if (pl.nt == 1) { if (sc1) PB_GO(1, true) else PB_GO(1, false) }
        else { if (sc1) PB_GO(2, true) else PB_GO(2, false) }
#undef PB_GO
        GN_LAUNCH_CHECK();
    }
    *slots = (int)std::min<long>((long)std::min<long>(x.B, BWD_NWG / pl.wgs) * pl.wgs, BWD_NWG);
    return 0;
}

// The sweep's plan: 1 or 2 tiles per workgroup only (its per-row state -- adjoint, gradient accumulators, the interval's own
// rows -- does not fit the 128 registers a 1024-thread workgroup leaves), up to 2 consecutive launches (measured: 4 launches at
// 600 nodes x 32 samples lose to one launch per interval)
bool gn_pers_bwd64_plan(const gnode_graph_s* g, long B, int n_steps, PersPlan* p) {
    if (n_steps < 2 || n_steps > 127 || B < 1) return false;
    for (long conc = B; conc >= 1; conc = (conc + 1) / 2) {
        PersPlan q;
        if (gn_pers64_plan(g, conc, n_steps, &q) && q.nt <= 2 && (B + q.concurrent - 1) / q.concurrent <= 2 && q.wgs <= BWD_NWG) { *p = q; return true; }
        if (conc == 1) break;
    }
    return false;
}
