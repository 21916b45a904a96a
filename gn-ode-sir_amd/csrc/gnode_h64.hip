// H = 64 fast path of the GN-ODE step for MI355X (gfx950).
//
// One launch per Euler step (per chunk of samples):
//   k_step64<PRJ>    persistent 256-thread workgroups (4 per CU) walk 16-node tiles, three tiles in flight per
//                    workgroup (software pipeline, see the kernel's header comment):
//       gather   AI = sum_{c in adj} Z_I[c], one 16-lane group per row, neighbour rows as 256-B coalesced reads,
//                eight in flight per group rolling through eight registers, column ids broadcast inside the group
//                by DPP row_newbcast fused into the address add, ascending-column sum (the CPU scatter_add_ order);
//       update   SIR derivative + Euler update of Y_S, Y_I, (Y_R) in place, optional trajectory write, fused read-out
//                head + 3-way softmax (16-lane DPP reductions);
//       MLPs     Z_I of the NEXT step from the freshly updated Y_I rows and Z_S of the NEXT tile from its Y_S rows in
//                ONE fp32 matrix-core phase (v_mfma_f32_16x16x4_f32, exact fp32), W^T resident in LDS for the whole
//                launch -- no separate node-MLP launch, Y_S / Y_I read once per step.
//       PRJ (inference): the R compartment is carried as w3 . Y_R (4 floats per row) -- it only feeds the read-out,
//                whose first layer is linear.
//       Rows longer than the hub threshold arrive pre-summed from gnode_hub.hip.
//   k_tiny64<PRJ>    graphs that fit one workgroup's LDS: ALL Euler steps in one launch.
//   k_mlp64          the MFMA tile engine alone (RHS API, RK4 stages, backward start-up).
//   k_prologue64     everything before the first step in one launch.
//
// Reference semantics: ode_nn_ngraph_sim.py:58-96 (RHS), :168 (euler), :172-187 (head).
#include "gnode_common.h"
#include "gnode_h64.h"
#include "gnode_mfma64.h"
#include "gnode_step64.h"
#include <algorithm>

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

// --------------------------------------------------------------------------- k_step64: software-pipelined Euler step
// Round 1's kernel walked a tile as a chain of five DEPENDENT memory round trips (Y_S row -> rowptr -> column ids ->
// neighbour rows 0..7 -> 8..15) behind five barriers with nothing of the next tile in flight (SQ_WAIT_ANY 66 % of wave
// cycles; 377 us per launch on the 75k graph x 8).  Here every workgroup keeps three tiles in flight (358 us):
//     tile t    gather being summed, SIR update, read-out                       (stages A, B)
//     tile t+1  Y_S row in registers -> LDS, first 8 neighbour rows + own rows requested   (stages B, C)
//     tile t+2  row header (extent + first 16 neighbour ids) / hub index / Y_S row / ids 16..31 requested   (top, D, F)
// and the two node MLPs that used to be separate barrier-fenced phases -- Z_I'(t) from the updated Y_I rows and
// Z_S(t+1) from the next tile's Y_S rows -- run in ONE matrix phase off one staged copy of W (each B fragment read
// once for both), between the only two barriers of the iteration.  The neighbour rows of tile t+1 travel under that
// matrix phase; the one exposed round trip per tile is the second half of the gather (8 loads in flight per lane group
// is the register budget), covered by the other three workgroups of the CU.
// Neighbour k of a row: lane (k & 15) of the 16-lane group holds its table byte offset in m (or the offset of the
// table's ZERO ROW when the row has fewer neighbours), so every load is unconditional: no compare, no exec-mask
// branch per neighbour, and hipcc counts the loads for its own s_waitcnt placement.
template <int K>
__device__ __forceinline__ float4 gat_ld(const float* __restrict__ ZI, unsigned m, unsigned lane_b) {
    return ld4o(ZI, (unsigned)row_bcast<K & 15>((int)m) + lane_b);
}
#define GP_ACC(V) acc.x += V.x; acc.y += V.y; acc.z += V.z; acc.w += V.w;
#define GP_SB __builtin_amdgcn_sched_barrier(0);

#ifndef GN_STEP_OCC
#define GN_STEP_OCC 4
#endif
// workgroups per CU: 4 for inference; the training instance (full Y_R rows, kept neighbour sums) needs ~140 VGPRs, and
// squeezed into the 128 of four waves per SIMD it spills INSIDE the loop -- a scratch reload is a vector-memory load,
// waiting for it drains the whole prefetch pipeline -- so it runs 3 per CU (the launch time does not depend on 3 vs 4)
template <bool PRJ> struct StepOcc { static constexpr int value = PRJ ? GN_STEP_OCC : (GN_STEP_OCC > 3 ? 3 : GN_STEP_OCC); };
// LAT ("latency mode"): launches in which every workgroup gets at most ONE tile (graphs up to ~16k rows per launch: the
// reference's fb-social / wiki-vote experiments at batch size 1..8).  Nothing overlaps there but a tile's own round trips,
// and the register file is nearly empty (one or two workgroups per CU), so the gather keeps 16 neighbour rows in flight
// per lane group instead of 8: one dependent round trip less per tile.  Same sums in the same order: bit-identical outputs.
// HUBS: the graph has rows longer than the hub threshold (compile-time, so that graphs without them -- the benchmark's -- carry
// neither the hub fields of the three in-flight stages nor their code: the 4-VGPR in-loop spill of round 2 is gone)
template <bool PRJ, bool LAT = false, bool HUBS = true>
__global__ __launch_bounds__(256, LAT ? 2 : StepOcc<PRJ>::value) void k_step64(const int* __restrict__ rowhdr, const int* __restrict__ col, int n,
                                                 long rows, int tiles_per_sample, long total_tiles,
                                                 float* Y, const float* __restrict__ ZI,
                                                 float* __restrict__ ZI_next, const float* __restrict__ W,
                                                 const float* __restrict__ bias, const float* __restrict__ beta,
                                                 const float* __restrict__ gamma, float dt,
                                                 const float* __restrict__ w3, const float* __restrict__ b3,
                                                 const float* __restrict__ w2, const float* __restrict__ b2,
                                                 float* __restrict__ PR, Step64Out out,
                                                 const int* __restrict__ hubidx, const float* __restrict__ HubP /* per-segment partial sums [B][n_seg][64] */,
                                                 const int* __restrict__ hub_seg_ptr, int n_seg) {
    // out.ai (training): the neighbour sums A Z_I(y_k) of this step are kept for the adjoint backward, which then reads a
    // row back instead of gathering the table a second time
    // measured on the 75k-node benchmark and fixed: non-temporal streaming state accesses (the gather table keeps the
    // L2) and 8 XCD-affine tile queues.  (Write-through `sc1` state stores, which drop the line from L2: 366 vs 360 us.)
    constexpr bool NT = true;
    constexpr int XQ = 8;
    // one LDS block, sub-arrays at compile-time offsets: every access below is (one per-lane offset) + constant
    constexpr int O_W = 0, O_TA = O_W + 64 * TS, O_TB = O_TA + 16 * TS, O_T2I = O_TB + 16 * TS, O_T2S = O_T2I + 16 * TS,
                  O_W3 = O_T2S + 16 * TS, O_END = O_W3 + 4 * 64;
    __shared__ __attribute__((aligned(16))) float L[O_END];
    //   TA  updated Y_I rows of tile t (MFMA operand)     TB  Y_S rows of tile t+1 (MFMA operand)
    //   T2I Z_I'(t)     T2S Z_S(t+1)     W3 read-out weight rows (re-read every tile: 16 loop-invariant VGPRs otherwise)
    // (Measured and dropped: RECOMPUTING the row's own Z_I from its Y_I row in the same matrix phase instead of
    //  reading it back -- 6 % fewer bytes through the L1-miss path, 44.5 KB of LDS, three workgroups per CU: 367 us
    //  against 360 us per launch on the 75k graph x 8.)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, sub = lane & 15;
    // training: ZI_next may be a fresh table of the kept-activation buffer (gn_keep_zi) -- give it its zero row
    if (!PRJ && blockIdx.x == 0 && threadIdx.x < 16) st4g(ZI_next + (size_t)rows * 64 + 4 * threadIdx.x, zero4());
    load_W_to_lds<false>(W, L + O_W);
    L[O_W3 + threadIdx.x] = w3[threadIdx.x];
    float bias_l = bias[16 * w + (lane & 15)];
    const size_t slab = (size_t)rows * 64;
    const float* YS = Y; const float* YI = Y + slab; const float* YR = Y + 2 * slab;
    float* Yo = out.sol ? out.sol : Y;
    float* YSo = Yo; float* YIo = Yo + slab; float* YRo = Yo + 2 * slab;
    const int lr = 4 * w + g;
    const unsigned lane_b = 16u * sub;
    int ro = lr * TS + 4 * sub;                            // this lane's 16 B of its row inside a tile
    int fo = (lane & 15) * TS + 16 * (lane >> 4);          // MFMA fragment offset (tile row i, k-quarter kq)
    int oo = 4 * (lane >> 4) * TS + 16 * w + (lane & 15);  // MFMA result offset (rows 4kq.., column 16w + i)

    // Tile walk: the tile range is cut into XQ contiguous queues (with 8 samples per launch a queue is one sample) and
    // workgroup i serves queue i % XQ.  Workgroups are dealt round-robin over the 8 XCDs, so queue q is gathered through
    // ONE XCD's L2: a sample's Z_I table competes for 4 MB instead of being spread over all eight L2s (speed only: every
    // tile is visited exactly once either way).  Everything here is uniform and
    // 32-bit (the launcher bounds rows < 2^24, so tiles < 2^20); readfirstlane pins the values to SGPRs -- left to
    // itself the compiler kept the 64-bit queue end in VGPRs, spilled it, and the reload's s_waitcnt vmcnt(0) at the
    // top of every iteration drained the whole prefetch pipeline.
    const int xq = (XQ > 1 && gridDim.x % XQ == 0 && total_tiles >= 4 * XQ) ? XQ : 1;
    const int q = blockIdx.x % xq;
    const int q_lo = __builtin_amdgcn_readfirstlane((int)(total_tiles * q / xq));
    const int q_hi = __builtin_amdgcn_readfirstlane((int)(total_tiles * (q + 1) / xq));
    const int t_stride = gridDim.x / xq;
    int t_it = q_lo + blockIdx.x / xq;                     // iterator: the tile the NEXT fetch_head() describes
    int b_it = __builtin_amdgcn_readfirstlane(t_it / tiles_per_sample);
    int tile_it = t_it - b_it * tiles_per_sample;

    // ---- per-stage state.  Uniform across the workgroup: *_ok (tile inside the queue).  Per lane group: the row.
    struct Stage { bool ok; bool valid; bool hub; unsigned row, base; int start, end; unsigned mine, mine2, hoff; int hcnt; };
    const unsigned zoff = (unsigned)rows * 256u;           // byte offset of the table's zero row (one past the last row)
    auto fetch_head = [&](Stage& s) {                      // top of the chain: row id, row header, hub index
        s.ok = t_it < q_hi;
        while (tile_it >= tiles_per_sample) { tile_it -= tiles_per_sample; ++b_it; }
        const int node = tile_it * 16 + lr;
        s.valid = s.ok && node < n;
        const int nodec = s.valid ? node : 0;
        s.row = s.valid ? (unsigned)b_it * (unsigned)n + (unsigned)node : 0u;
        s.start = 0; s.end = 0; s.hub = false; s.mine = zoff; s.hoff = 0u; s.hcnt = 0;
        int c0 = 0;
        if (s.ok) {                                        // uniform: past the end of the queue nothing is requested at all
            // the row's header: extent and first 16 neighbour ids in one round trip (inside a tile the loads are
            // unconditional: clamped row)
            const int* h = rowhdr + (size_t)nodec * 20;
            s.start = h[0]; s.end = h[1];
            c0 = h[4 + sub];
            if (HUBS) {                                    // compile-time: graphs without long rows carry none of this
                const int h = hubidx[nodec];
                s.hub = s.valid && h >= 0;
                // a hub row's neighbour sum arrives as per-segment partials (k_hub_seg): [hoff, hoff + hcnt rows) of HubP
                const int s0 = s.hub ? hub_seg_ptr[h] : 0, s1 = s.hub ? hub_seg_ptr[h + 1] : 0;
                s.hoff = ((unsigned)b_it * (unsigned)n_seg + (unsigned)s0) * 256u;
                s.hcnt = s1 - s0;
            }
        }
        if (!s.valid) s.end = s.start;
        s.base = (unsigned)b_it * (unsigned)n;             // first row of the tile's sample (uniform)
        if (sub < (s.hub ? 0 : s.end - s.start)) s.mine = (s.base + (unsigned)c0) * 256u;
        t_it += t_stride; tile_it += t_stride;
    };
    auto fetch_cols = [&](Stage& s) {                      // second link: column ids 16..31 (rows longer than the header's 16)
        s.mine2 = zoff;
        if (s.ok) {
            const int d = s.hub ? 0 : s.end - s.start;
            const int c1 = col[16 + sub < d ? s.start + 16 + sub : 0];
            s.mine2 = 16 + sub < d ? (s.base + (unsigned)c1) * 256u : zoff;
        }
        if (s.hub) s.end = s.start;                        // nothing to gather: the sum arrives from the hub kernels
    };

    Stage cur, n1, n2;
    cur.ok = false; cur.valid = false; cur.hub = false; cur.row = 0; cur.start = cur.end = 0; cur.mine = cur.mine2 = zoff; cur.hoff = 0; cur.hcnt = 0; cur.base = 0;
    fetch_head(n1);
    float4 ys_n1 = zero4();
    if (n1.ok) ys_n1 = ld4so<NT>(YS, n1.row * 256u + lane_b);
    fetch_cols(n1);
    float4 v0 = zero4(), v1 = zero4(), v2 = zero4(), v3 = zero4(), v4 = zero4(), v5 = zero4(), v6 = zero4(), v7 = zero4();
    float4 v8 = zero4(), v9 = zero4(), v10 = zero4(), v11 = zero4(), v12 = zero4(), v13 = zero4(), v14 = zero4(), v15 = zero4();   // LAT only
    float4 yi = zero4(), yr = zero4(), zi = zero4(), pr = zero4();
    float nb = 0.f, gm = 0.f;
    __syncthreads();                                       // W staged

    for (;;) {
        // keep the LDS offsets opaque per iteration: hoisted out of the loop, every (offset + constant) becomes its own
        // loop-invariant VGPR (a dozen of them) instead of an instruction offset field
        asm volatile("" : "+v"(ro), "+v"(fo), "+v"(oo), "+v"(bias_l));
        float* const tA = L + O_TA + ro;
        float* const tB = L + O_TB + ro;
        const float* const t2i = L + O_T2I + ro;
        const float* const t2s = L + O_T2S + ro;
        const float* const w3s = L + O_W3;
        // ---- top: head of tile t+2's chain
        fetch_head(n2);
        // ---- A: finish the gather of tile t.  v0..v7 hold neighbours 0..7 (requested one matrix phase ago); further
        // neighbours roll through the same eight registers four at a time, as far as the longest row of the WAVE needs
        // (wave-uniform branches), always summed in ascending column order (the CPU scatter_add_ order of the reference).
        float4 acc = zero4();
        {
            const int cnt = cur.hub ? 0 : cur.end - cur.start;
            const unsigned m = cur.mine, m2 = cur.mine2;
#define GP_R(V, K) GP_ACC(V) V = gat_ld<K>(ZI, (K) < 16 ? mm : mm2, lane_b); GP_SB
            // (GP_RQ: the offsets are re-opaqued per group of four -- otherwise a long case computes all of its broadcast
            //  addresses up front, and the longest one spilled a state register around its first loads)
#define GP_RQ asm volatile("" : "+v"(mm), "+v"(mm2));
#define GP_RA(K) GP_RQ GP_R(v0, K) GP_R(v1, (K) + 1) GP_R(v2, (K) + 2) GP_R(v3, (K) + 3)
#define GP_RB(K) GP_RQ GP_R(v4, K) GP_R(v5, (K) + 1) GP_R(v6, (K) + 2) GP_R(v7, (K) + 3)
#define GP_FA GP_ACC(v0) GP_ACC(v1) GP_ACC(v2) GP_ACC(v3) GP_ACC(v4) GP_ACC(v5) GP_ACC(v6) GP_ACC(v7)
#define GP_FB GP_ACC(v4) GP_ACC(v5) GP_ACC(v6) GP_ACC(v7) GP_ACC(v0) GP_ACC(v1) GP_ACC(v2) GP_ACC(v3)
            // one straight-line path per gather length (wave-uniform switch): only `acc` is live at the join, so the eight
            // row registers never meet in a phi and stay eight registers
            const int nb4 = LAT ? -1 : (__any(cnt > 8) ? 1 : 0) + (__any(cnt > 12) ? 1 : 0) + (__any(cnt > 16) ? 1 : 0) +
                                       (__any(cnt > 20) ? 1 : 0) + (__any(cnt > 24) ? 1 : 0) + (__any(cnt > 28) ? 1 : 0);
            // (GP_OPQ: a per-case opaque copy of the offsets -- otherwise the 24 broadcast addresses are hoisted above the
            //  switch as common subexpressions, 24 live VGPRs and a spill whose reload drains every load in flight)
#define GP_OPQ unsigned mm = m, mm2 = m2; asm volatile("" : "+v"(mm), "+v"(mm2));
            if (LAT) {
                // 16 rows were requested; rows 16..31 (through the second id chunk) in two unconditional batches of 8
                GP_ACC(v0) GP_ACC(v1) GP_ACC(v2) GP_ACC(v3) GP_ACC(v4) GP_ACC(v5) GP_ACC(v6) GP_ACC(v7)
                GP_ACC(v8) GP_ACC(v9) GP_ACC(v10) GP_ACC(v11) GP_ACC(v12) GP_ACC(v13) GP_ACC(v14) GP_ACC(v15)
                if (__any(cnt > 16)) {
                    v0 = gat_ld<0>(ZI, m2, lane_b); v1 = gat_ld<1>(ZI, m2, lane_b); v2 = gat_ld<2>(ZI, m2, lane_b); v3 = gat_ld<3>(ZI, m2, lane_b);
                    v4 = gat_ld<4>(ZI, m2, lane_b); v5 = gat_ld<5>(ZI, m2, lane_b); v6 = gat_ld<6>(ZI, m2, lane_b); v7 = gat_ld<7>(ZI, m2, lane_b);
                    if (__any(cnt > 24)) {
                        v8 = gat_ld<8>(ZI, m2, lane_b); v9 = gat_ld<9>(ZI, m2, lane_b); v10 = gat_ld<10>(ZI, m2, lane_b); v11 = gat_ld<11>(ZI, m2, lane_b);
                        v12 = gat_ld<12>(ZI, m2, lane_b); v13 = gat_ld<13>(ZI, m2, lane_b); v14 = gat_ld<14>(ZI, m2, lane_b); v15 = gat_ld<15>(ZI, m2, lane_b);
                        GP_ACC(v0) GP_ACC(v1) GP_ACC(v2) GP_ACC(v3) GP_ACC(v4) GP_ACC(v5) GP_ACC(v6) GP_ACC(v7)
                        GP_ACC(v8) GP_ACC(v9) GP_ACC(v10) GP_ACC(v11) GP_ACC(v12) GP_ACC(v13) GP_ACC(v14) GP_ACC(v15)
                    } else {
                        GP_ACC(v0) GP_ACC(v1) GP_ACC(v2) GP_ACC(v3) GP_ACC(v4) GP_ACC(v5) GP_ACC(v6) GP_ACC(v7)
                    }
                }
            } else
            switch (nb4) {
                case 0: { GP_FA } break;
                case 1: { GP_OPQ GP_RA(8) GP_FB } break;
                case 2: { GP_OPQ GP_RA(8) GP_RB(12) GP_FA } break;
                case 3: { GP_OPQ GP_RA(8) GP_RB(12) GP_RA(16) GP_FB } break;
                case 4: { GP_OPQ GP_RA(8) GP_RB(12) GP_RA(16) GP_RB(20) GP_FA } break;
                case 5: { GP_OPQ GP_RA(8) GP_RB(12) GP_RA(16) GP_RB(20) GP_RA(24) GP_FB } break;
                default: { GP_OPQ GP_RA(8) GP_RB(12) GP_RA(16) GP_RB(20) GP_RA(24) GP_RB(28) GP_FA } break;
            }
#undef GP_OPQ
#undef GP_FB
#undef GP_FA
#undef GP_RB
#undef GP_RA
#undef GP_RQ
#undef GP_R
            // rows of 33 .. hub-threshold edges: the plain chunked walk for the rest
            if (__any(cnt > 32)) {
                const unsigned base = cur.base;
                for (int e0 = cur.start + 32; e0 < cur.end; e0 += 16) {
                    const int c2 = cur.end - e0;
                    const unsigned mm = (sub < c2) ? (base + (unsigned)col[e0 + sub]) * 256u : zoff;
                    {
                        float4 u0 = gat_ld<0>(ZI, mm, lane_b), u1 = gat_ld<1>(ZI, mm, lane_b), u2 = gat_ld<2>(ZI, mm, lane_b), u3 = gat_ld<3>(ZI, mm, lane_b);
                        float4 u4 = gat_ld<4>(ZI, mm, lane_b), u5 = gat_ld<5>(ZI, mm, lane_b), u6 = gat_ld<6>(ZI, mm, lane_b), u7 = gat_ld<7>(ZI, mm, lane_b);
                        GP_ACC(u0) GP_ACC(u1) GP_ACC(u2) GP_ACC(u3) GP_ACC(u4) GP_ACC(u5) GP_ACC(u6) GP_ACC(u7)
                    }
                    if (c2 > 8) {
                        float4 u0 = gat_ld<8>(ZI, mm, lane_b), u1 = gat_ld<9>(ZI, mm, lane_b), u2 = gat_ld<10>(ZI, mm, lane_b), u3 = gat_ld<11>(ZI, mm, lane_b);
                        float4 u4 = gat_ld<12>(ZI, mm, lane_b), u5 = gat_ld<13>(ZI, mm, lane_b), u6 = gat_ld<14>(ZI, mm, lane_b), u7 = gat_ld<15>(ZI, mm, lane_b);
                        GP_ACC(u0) GP_ACC(u1) GP_ACC(u2) GP_ACC(u3) GP_ACC(u4) GP_ACC(u5) GP_ACC(u6) GP_ACC(u7)
                    }
                }
            }
        }
        // ---- B: P3 of tile t (ode_nn_ngraph_sim.py:75-77, Euler update, read-out), then stage the MFMA operands
        if (cur.ok) {
            const unsigned off = cur.row * 256u + lane_b;
            const float4 zs = *reinterpret_cast<const float4*>(t2s);
            float4 ys = *reinterpret_cast<const float4*>(tB);          // this row's Y_S, staged one iteration ago
            float4 dS, dI, dR;
            dS.x = nb * (acc.x * zs.x); dS.y = nb * (acc.y * zs.y); dS.z = nb * (acc.z * zs.z); dS.w = nb * (acc.w * zs.w);
            dR.x = gm * zi.x; dR.y = gm * zi.y; dR.z = gm * zi.z; dR.w = gm * zi.w;
            dI.x = -dS.x - dR.x; dI.y = -dS.y - dR.y; dI.z = -dS.z - dR.z; dI.w = -dS.w - dR.w;
            ys.x += dt * dS.x; ys.y += dt * dS.y; ys.z += dt * dS.z; ys.w += dt * dS.w;
            yi.x += dt * dI.x; yi.y += dt * dI.y; yi.z += dt * dI.z; yi.w += dt * dI.w;
            float prj[4] = {0.f, 0.f, 0.f, 0.f};
            if (PRJ) {
                float4 w3r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) w3r[k] = *reinterpret_cast<const float4*>(w3s + k * 64 + 4 * sub);
                prj[0] = pr.x + dt * (gm * row_sum16(fmaf(w3r[0].x, zi.x, fmaf(w3r[0].y, zi.y, fmaf(w3r[0].z, zi.z, w3r[0].w * zi.w)))));
                prj[1] = pr.y + dt * (gm * row_sum16(fmaf(w3r[1].x, zi.x, fmaf(w3r[1].y, zi.y, fmaf(w3r[1].z, zi.z, w3r[1].w * zi.w)))));
                prj[2] = pr.z + dt * (gm * row_sum16(fmaf(w3r[2].x, zi.x, fmaf(w3r[2].y, zi.y, fmaf(w3r[2].z, zi.z, w3r[2].w * zi.w)))));
                prj[3] = pr.w + dt * (gm * row_sum16(fmaf(w3r[3].x, zi.x, fmaf(w3r[3].y, zi.y, fmaf(w3r[3].z, zi.z, w3r[3].w * zi.w)))));
                if (cur.valid && sub == 0) st4o(PR, cur.row * 16u, make_float4(prj[0], prj[1], prj[2], prj[3]));
            } else {
                yr.x += dt * dR.x; yr.y += dt * dR.y; yr.z += dt * dR.z; yr.w += dt * dR.w;
            }
            if (cur.valid) {
                st4so<NT>(YSo, off, ys); st4so<NT>(YIo, off, yi);
                if (!PRJ) st4so<NT>(YRo, off, yr);
                if (!PRJ && out.ai) {
                    // without `keep`: A Z_I(y_k) itself; with it: P_S = A Z_I * Z_S (1 - Z_S), the only form the backward needs
                    if (out.zs) st4so<NT>(out.ai, off, make_float4(acc.x * (zs.x * (1.0f - zs.x)), acc.y * (zs.y * (1.0f - zs.y)),
                                                                  acc.z * (zs.z * (1.0f - zs.z)), acc.w * (zs.w * (1.0f - zs.w))));
                    else st4so<NT>(out.ai, off, acc);
                }
                if (!PRJ && out.zs) st4so<NT>(out.zs, off, zs);
            }
            if (out.S) {
                float pS, pI, pR;
                readout64<PRJ>(ys, yi, yr, prj, sub, w3s, b3, w2, b2, pS, pI, pR);
                if (cur.valid && sub == 0) { out.S[cur.row] = pS; out.I[cur.row] = pI; out.R[cur.row] = pR; }
            }
            *reinterpret_cast<float4*>(tA) = yi;                       // Y_I' -> operand of Z_I'(t)
        }
        if (n1.ok) *reinterpret_cast<float4*>(tB) = ys_n1;             // Y_S(t+1) -> operand of Z_S(t+1)
        // ---- C: request tile t+1's first 8 neighbour rows and its own rows (they travel under the matrix phase).
        // Unconditional on purpose (past the queue's end they read row 0 and the zero row): guarding this block and the
        // next load by the uniform `ok` flags cost 9 us per launch on the 75k graph x 8 (368 vs 359)
        {
            const unsigned m = n1.mine;
            v0 = gat_ld<0>(ZI, m, lane_b); v1 = gat_ld<1>(ZI, m, lane_b); v2 = gat_ld<2>(ZI, m, lane_b); v3 = gat_ld<3>(ZI, m, lane_b);
            v4 = gat_ld<4>(ZI, m, lane_b); v5 = gat_ld<5>(ZI, m, lane_b); v6 = gat_ld<6>(ZI, m, lane_b); v7 = gat_ld<7>(ZI, m, lane_b);
            if (LAT) {
                v8 = gat_ld<8>(ZI, m, lane_b); v9 = gat_ld<9>(ZI, m, lane_b); v10 = gat_ld<10>(ZI, m, lane_b); v11 = gat_ld<11>(ZI, m, lane_b);
                v12 = gat_ld<12>(ZI, m, lane_b); v13 = gat_ld<13>(ZI, m, lane_b); v14 = gat_ld<14>(ZI, m, lane_b); v15 = gat_ld<15>(ZI, m, lane_b);
            }
            if (HUBS) {                                                // compile-time: the graph has hub rows
                // a hub row's sum = its segment partials added in segment order (what a separate reduction launch used to
                // do: at mid size a launch costs as much as the step): 8 partial rows in flight, waves without a hub row skip
                const int hc = n1.hcnt;
                if (__any(hc > 0)) {
                    float4 hs = zero4();
                    for (int sg = 0; __any(sg < hc); sg += 8) {
#define GN_HP(Q, U) float4 U = zero4(); if (sg + (Q) < hc) U = ld4o(HubP, n1.hoff + (unsigned)(sg + (Q)) * 256u + lane_b);
                        GN_HP(0, u0) GN_HP(1, u1) GN_HP(2, u2) GN_HP(3, u3) GN_HP(4, u4) GN_HP(5, u5) GN_HP(6, u6) GN_HP(7, u7)
#undef GN_HP
                        hs.x += u0.x; hs.y += u0.y; hs.z += u0.z; hs.w += u0.w;  hs.x += u1.x; hs.y += u1.y; hs.z += u1.z; hs.w += u1.w;
                        hs.x += u2.x; hs.y += u2.y; hs.z += u2.z; hs.w += u2.w;  hs.x += u3.x; hs.y += u3.y; hs.z += u3.z; hs.w += u3.w;
                        hs.x += u4.x; hs.y += u4.y; hs.z += u4.z; hs.w += u4.w;  hs.x += u5.x; hs.y += u5.y; hs.z += u5.z; hs.w += u5.w;
                        hs.x += u6.x; hs.y += u6.y; hs.z += u6.z; hs.w += u6.w;  hs.x += u7.x; hs.y += u7.y; hs.z += u7.z; hs.w += u7.w;
                    }
                    if (n1.hub) v0 = hs;
                }
            }
            const unsigned off = n1.row * 256u + lane_b;
            yi = ld4so<NT>(YI, off);
            if (!PRJ) yr = ld4so<NT>(YR, off);
            zi = ld4o(ZI, off);
            if (PRJ) pr = ld4o(PR, n1.row * 16u);
            nb = -beta[n1.row]; gm = gamma[n1.row];
        }
        // ---- D: tile t+2's Y_S row
        const float4 ys_n2 = ld4so<NT>(YS, n2.row * 256u + lane_b);
        __syncthreads();
        // ---- E: Z_I'(t) and Z_S(t+1) in one matrix phase, each W fragment read once for both
#define GN_DUAL(A, B) mfma_dual16<A, B>(L + O_TA, L + O_TB, L + O_W + 16 * w * TS, L + O_T2I, L + O_T2S, bias_l, fo, oo)
        if (cur.ok && n1.ok) GN_DUAL(true, true);
        else if (cur.ok) GN_DUAL(true, false);
        else if (n1.ok) GN_DUAL(false, true);
#undef GN_DUAL
        __syncthreads();
        // ---- F: next step's gather table row of tile t; column ids of tile t+2
        if (cur.valid) st4so<NT>(ZI_next, cur.row * 256u + lane_b, *reinterpret_cast<const float4*>(t2i));   // (the next launch gathers it; this one never reads it)
        fetch_cols(n2);
        if (!n1.ok) break;
        cur = n1; n1 = n2; ys_n1 = ys_n2;
    }
}
#undef GP_SB
#undef GP_ACC

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
                                                float* __restrict__ sol, float* __restrict__ keep) {
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
                if (!PRJ && keep) {                // kept activations of grid point k (gn_keep_zs / gn_keep_zi): the adjoint sweep reads them back
                    st4g(gn_keep_zs(keep, rows, k) + off, zs); st4g(gn_keep_zi(keep, rows, k) + off, zi);
                }
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
#ifndef GN_TINY_TRAIN
#define GN_TINY_TRAIN 1      // 0: training forwards (and the adjoint sweep) of tiny graphs take the persistent launches instead
#endif
bool gn_tiny64_ok(int n, int n_steps, int n_out, bool prj) {
    if (!prj && !GN_TINY_TRAIN) return false;
    return n_steps >= 1 && n_steps <= 128 && n_out < 32768 && n <= 3 * TILE_ROWS &&
           gn_tiny64_lds_bytes(n, prj) <= 160 * 1024;
}

int gn_launch_tiny64(const gnode_graph_s* g, long rows, const float* Y0, const float* ZI0, const float* PR0, const float* W,
                     const float* bias, const float* beta, const float* gamma, const float* dt_host, const int* slot_host,
                     int n_steps, const gnode_params* p, float* S, float* I, float* R, float* sol, float* keep, hipStream_t st) {
    TinySched sched;
    sched.n_steps = n_steps;
    for (int k = 0; k < n_steps; ++k) { sched.dt[k] = dt_host[k]; sched.slot[k] = (short)slot_host[k]; }
    const bool prj = PR0 != nullptr;
    const size_t lds = gn_tiny64_lds_bytes(g->n, prj);
    const unsigned B = (unsigned)(rows / g->n);
    const unsigned threads = 256u * (unsigned)((g->n + TILE_ROWS - 1) / TILE_ROWS);
    if (prj) {
        hipLaunchKernelGGL(k_tiny64<true>, dim3(B), dim3(threads), lds, st, g->rowptr, g->col, g->n, rows, Y0, ZI0, PR0, W, bias, beta,
                           gamma, sched, p->linear3_weight, p->linear3_bias, p->linearS2_weight, p->linearS2_bias, S, I, R, sol, keep);
    } else {
        hipLaunchKernelGGL(k_tiny64<false>, dim3(B), dim3(threads), lds, st, g->rowptr, g->col, g->n, rows, Y0, ZI0, PR0, W, bias, beta,
                           gamma, sched, p->linear3_weight, p->linear3_bias, p->linearS2_weight, p->linearS2_bias, S, I, R, sol, keep);
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
                                                    float* __restrict__ sol0, float* __restrict__ ZI, float* __restrict__ ZI_alt,
                                                    float* __restrict__ PR,
                                                    float* __restrict__ S0, float* __restrict__ I0, float* __restrict__ R0,
                                                    long rows, uint32_t* __restrict__ zero_words, int n_zero_words) {
    if (zero_words && blockIdx.x == 0)          // the control block of the persistent launch that follows (gnode_pers64.hip)
        for (int i = threadIdx.x; i < n_zero_words; i += 256) zero_words[i] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < 32) {            // the zero rows behind the two gather tables (k_step64)
        float* z = (threadIdx.x < 16 ? ZI : ZI_alt) + (size_t)rows * 64 + 4 * (threadIdx.x & 15);
        st4g(z, zero4());
    }
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
                         float* ZI_alt, float* PR, float* S0, float* I0, float* R0, long rows, void* zero_ptr, size_t zero_bytes, hipStream_t st) {
    const long ntiles = (rows + TILE_ROWS - 1) / TILE_ROWS;
    hipLaunchKernelGGL(k_prologue64, dim3((unsigned)std::min<long>(ntiles, 2048)), dim3(256), 0, st, x, p->linearS1_weight,
                       p->linearS1_bias, p->odefunc_linear_weight, p->odefunc_linear_bias, p->linear3_weight, p->linear3_bias,
                       p->linearS2_weight, p->linearS2_bias, Y, beta, gamma, sol0, ZI, ZI_alt, PR, S0, I0, R0, rows, (uint32_t*)zero_ptr,
                       (int)(zero_ptr ? zero_bytes / 4 : 0));
    GN_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------- host launchers
int gn_h64_set_attributes() {
    // dynamic LDS above 64 KB needs the attribute once per device: done at graph creation, never on a launch path
    GN_HIP(hipFuncSetAttribute((const void*)k_tiny64<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    GN_HIP(hipFuncSetAttribute((const void*)k_tiny64<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

int gn_launch_mlp64(const gnode_graph_s* g, const float* X, const float* W, const float* b, float* Z, long nrows, hipStream_t st) {
    if (nrows <= 0) return 0;
    const long ntiles = (nrows + TILE_ROWS - 1) / TILE_ROWS;
    const int grid = (int)std::min<long>(ntiles, (long)g->num_cu * 4);
    hipLaunchKernelGGL(k_mlp64, dim3(grid), dim3(256), 0, st, X, W, b, Z, nrows);
    GN_LAUNCH_CHECK();
    return 0;
}

int gn_launch_step64(gnode_graph_s* g, long rows, float* Y, const float* ZI, float* ZI_next, const float* W,
                     const float* bias, const float* beta, const float* gamma, float dt, const gnode_params* p,
                     float* PR, Step64Out out, void* hub_scratch, hipStream_t st) {
    GN_CHECK_ARG(rows < (1L << 24), "H=64 step kernel addresses rows with 32-bit byte offsets: rows=%ld >= 2^24 per launch "
                 "(split the batch)", rows);
    // (the hub segment partials are addressed the same way: (sample * n_seg + segment) * 256 bytes)
    GN_CHECK_ARG((long)(rows / g->n) * g->n_seg < (1L << 24), "H=64 step kernel addresses hub segment partials with 32-bit byte offsets: "
                 "%ld samples x %d segments >= 2^24 per launch (split the batch)", (long)(rows / g->n), g->n_seg);
    const int tps = (g->n + 15) / 16;
    const long total = (long)(rows / g->n) * tps;
    const float* HubP = nullptr;               // per-segment partial sums of the hub rows; the step kernel adds them up itself
    if (int e = gn_hub_segments(g, rows / g->n, 64, ZI, hub_scratch, &HubP, st)) return e;
    // persistent grid: GN_STEP_OCC workgroups per CU (measured: 3 / 4 / 5 per CU -> 363 / 360 / 400 us per launch on the
    // 75k graph x 8; shrinking the grid so that every workgroup gets the same number of tiles is slower than filling
    // every slot and accepting a +-1 tile imbalance)
    const int grid = (int)std::min<long>(total, (long)g->num_cu * (PR ? StepOcc<true>::value : StepOcc<false>::value));
    const bool lat = total <= grid;            // one tile per workgroup: the latency-mode instantiation
#define GN_STEP1(P, L, HB) hipLaunchKernelGGL((k_step64<P, L, HB>), dim3(grid), dim3(256), 0, st, g->rowhdr, g->col, g->n, rows, tps, total, Y, ZI, \
                                         ZI_next, W, bias, beta, gamma, dt, p->linear3_weight, p->linear3_bias, p->linearS2_weight,     \
                                         p->linearS2_bias, PR, out, g->hubidx, HubP, g->hub_seg_ptr, g->n_seg)
#define GN_STEP(P, L) { if (g->n_hub > 0) GN_STEP1(P, L, true); else GN_STEP1(P, L, false); }
    if (PR) { if (lat) GN_STEP(true, true) else GN_STEP(true, false) }
    else { if (lat) GN_STEP(false, true) else GN_STEP(false, false) }
#undef GN_STEP
#undef GN_STEP1
    GN_LAUNCH_CHECK();
    return 0;
}
