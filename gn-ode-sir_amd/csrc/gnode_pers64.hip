// H = 64 Euler integration of mid-size graphs in ONE persistent launch (gfx950).
//
// Regime: the reference's actual experiments (monitorer-sim.py:10,17-22: batch_size 1, fb-social 1 893 / wiki-vote 7 066
// nodes) -- a sample's whole state is a few hundred KB.  With one launch per Euler step (gnode_h64.hip, latency mode) a
// step costs a kernel boundary plus a chain of dependent round trips (re-stage W, row header, neighbour rows, state rows):
// ~7 us at fb-social size for ~0.5 us of arithmetic.  Here every workgroup OWNS its rows for the whole integration:
//   * Y_S, Y_I, Y_R (or its 4-float projection), Z_S, Z_I, beta, gamma and the row's neighbour ids stay in REGISTERS across
//     all steps; W^T and the read-out weights stay in LDS.  Per step the only global traffic is the gather of the
//     neighbours' Z_I rows, the store of the row's own new Z_I, and the outputs.
//   * the workgroups of one sample (a "group") meet once per step at a flag barrier.  One workgroup per CU (LDS-forced),
//     groups placed by XCC_ID: every workgroup reads its XCD from the hardware register and draws a ticket there, so a
//     group that fits one XCD really sits on one XCD (its Z_I rows are produced and gathered through ONE L2), and larger
//     groups span 2 / 4 / 8 whole XCDs.  Nothing about correctness depends on the placement.
//   * hand-off protocol = MI355X_MICROARCH.md, "Valid forms", first row of the measured table: payload stores, every
//     storing wave `s_waitcnt vmcnt(0)`, workgroup barrier, ONE lane stores the workgroup's epoch flag `sc1`; the consumer's
//     wave 0 polls the group's flags with `sc1` loads, the other waves wait at the workgroup barrier that wave then joins;
//     EVERY load of a handed-off byte is a `buffer_load ... sc1` (bypasses the CU's L1, which another CU's stores never
//     refresh).  Payload stores are `sc1` (write-through) for groups that span XCDs; for single-XCD groups they are plain
//     stores -- producers and consumers share that XCD's L2, and tools/xcd_handoff_probe.hip checked every word of 200
//     steps under uneven load (0 stale in either form; the plain form is 0.4 us per step faster).
//   * every spin is bounded (2 s): a workgroup that gives up writes a code to the control block (gnode_persistent_status).
// Arithmetic, summation order and the MFMA chains are those of k_step64 / k_prologue64: outputs are bit-identical to the
// one-launch-per-step path (tests/test_gpu_persistent.py).
//
// Reference semantics: ode_nn_ngraph_sim.py:58-96 (RHS), :168 (euler), :172-187 (head).
#include "gnode_common.h"
#include "gnode_h64.h"
#include "gnode_mfma64.h"
#include "gnode_step64.h"
#include "gnode_pers64.h"
#include "gnode_pers64_dev.h"
#include <algorithm>
#include <queue>
#include <vector>

template <bool PRJ, int NT, bool SC1ST, bool HUBS>
__global__ __launch_bounds__(256 * NT) void k_pers64(const PersArgs a) {
    constexpr int STAUX = SC1ST ? 16 : 0;
    // registers: one workgroup per CU, so 256 / 512 / 1024 threads leave 512 / 256 / 128 VGPRs per lane
    constexpr int NM = 6;                                  // neighbour-id registers (16 ids each): every row up to the hub threshold (96)
    static_assert(16 * NM >= GN_HUB_T, "the id registers must cover every non-hub row");
#ifndef GN_PERS_DEPTH
#define GN_PERS_DEPTH 4
#endif
    constexpr int DEPTH = NT == 4 ? 1 : (PRJ ? GN_PERS_DEPTH : 3);   // batches of 8 neighbour rows in flight per lane group (training: more state)
    constexpr int O_W = 0, O_W3 = O_W + 64 * TS, O_HD = O_W3 + 256, O_T = O_HD + 16, TEAM_F = 4 * 16 * TS;   // O_HD: b3[4] | w2[4] | b2
    extern __shared__ __attribute__((aligned(16))) float L[];
    __shared__ unsigned sh[4];
    // (team and wave index are wave-uniform: pinned to SGPRs, so that the LDS tile addresses derived from them are scalars)
    const int team = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)), tid = threadIdx.x & 255;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, sub = lane & 15;
    if (team == 0) load_W_to_lds<false>(a.W, L + O_W);
    if (threadIdx.x < 256) L[O_W3 + threadIdx.x] = a.w3[threadIdx.x];
    if (threadIdx.x < 4) { L[O_HD + threadIdx.x] = a.b3[threadIdx.x]; L[O_HD + 4 + threadIdx.x] = a.w2[threadIdx.x]; }
    if (threadIdx.x == 4) L[O_HD + 8] = a.b2[0];
    int gl, idx;                                           // concurrent group, workgroup inside the group
    if (!pers_place(a.pp, a.ctl, sh, gl, idx)) return;
    unsigned* const flags = a.ctl->flags + (size_t)gl * a.pp.fstride;
    unsigned* const err = a.ctl->error;

    const float bias_l = a.bias[16 * w + (lane & 15)];
    float* const TA = L + O_T + team * TEAM_F;             // Y_I' rows (MFMA operand)
    float* const TB = TA + 16 * TS;                        // Y_S' rows
    float* const T2I = TB + 16 * TS;                       // Z_I'
    float* const T2S = T2I + 16 * TS;                      // Z_S'
    float* const HP = L + O_T + NT * TEAM_F;               // HUBS: segment partials [S][64] | segment ids [S][32] | item lists (gnode_pers64_dev.h)
    const float* const Wslab = L + O_W + 16 * w * TS;
    const float* const w3s = L + O_W3;
    const int lr = 4 * w + g;
    unsigned* const HI = reinterpret_cast<unsigned*>(HP + (size_t)a.lds_slots * 64);
    unsigned* const HLmine = HI + (size_t)a.lds_slots * 32 + (size_t)(team * 16 + lr) * PERS_MAX_ITEMS;
    const unsigned lane_b = 16u * sub;
    const int ro = lr * TS + 4 * sub;
    const int fo = (lane & 15) * TS + 16 * (lane >> 4);
    const int oo = 4 * (lane >> 4) * TS + 16 * w + (lane & 15);
    const unsigned rows = a.rows;
    const unsigned zoff = rows * 256u;
    const unsigned tbytes = (rows + 1u) * 256u;
    const size_t slab = (size_t)rows * 64;
    const int n_steps = a.sched.n_steps;
    // which node this lane group owns: the plan's row map deals degree-sorted quads of rows to the waves (a wave's four
    // rows have similar lengths -- its load count is its longest row's) and the quads round-robin to the workgroups
    const int lgslot = idx * (16 * NT) + team * 16 + lr;
    const int node = a.rowmap[lgslot];
    const bool valid = node >= 0;
    int hs0 = -1, hcnt = 0, it0 = 0, itn = 0;               // the hub row this lane group owns; the segment sums it computes
    if (HUBS) { hs0 = a.hubslot[2 * lgslot]; hcnt = a.hubslot[2 * lgslot + 1]; it0 = a.segptr[2 * lgslot]; itn = a.segptr[2 * lgslot + 1]; }

#ifdef GN_PERS_PROF
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_amdgcn_s_memrealtime();
#endif
    for (int round = 0; round < a.pp.rounds; ++round) {
        const int b = round * a.pp.concurrent + gl;
        if (b >= a.B) break;
        const unsigned ebase = (unsigned)round * (unsigned)n_steps;
        const unsigned base = (unsigned)b * (unsigned)a.n;
        const unsigned row = valid ? base + (unsigned)node : 0u;
        const unsigned off = row * 256u + lane_b;
        // ---- the row's loop-invariant data: extent, neighbour ids (as table byte offsets), beta, gamma
        int start = 0, end = 0;
        unsigned m[NM];
#pragma unroll
        for (int j = 0; j < NM; ++j) m[j] = PS_OOB;
        if (valid) {
            const int* h = a.rowhdr + (size_t)node * 20;
            start = h[0]; end = h[1];
            if (HUBS && hs0 >= 0) end = start;              // a hub row gathers nothing itself: its sum arrives as segment partials
            const int d = end - start;
            if (sub < d) m[0] = (base + (unsigned)h[4 + sub]) * 256u;
#pragma unroll
            for (int j = 1; j < NM; ++j)
                if (16 * j + sub < d) m[j] = (base + (unsigned)a.col[start + 16 * j + sub]) * 256u;
        }
        const int nbt = pers_batches<NM>(end - start);
        if (HUBS) pers_hub_stage(a.col, a.segitem, it0, itn, base, HI, HLmine, sub);    // (the barriers below publish it)
        float4 ys = zero4(), yi = zero4(), yr = zero4(), pr = zero4(), zi = zero4();
        float nb = 0.f, gm = 0.f;
        if (valid) {
            ys = ld4o(a.Y0, off); yi = ld4o(a.Y0 + slab, off);
            if (PRJ) pr = ld4o(a.PR0, row * 16u); else yr = ld4o(a.Y0 + 2 * slab, off);
            zi = ld4o(a.keep ? gn_keep_zi(a.keep, rows, 0) : a.Z0, off);        // Z_I(y_0): the prologue launch wrote it
            nb = -a.beta[row]; gm = a.gamma[row];
        }
        *reinterpret_cast<float4*>(TB + ro) = ys;
        __syncthreads();
        mfma_dual16<false, true>(TA, TB, Wslab, T2I, T2S, bias_l, fo, oo);       // Z_S(y_0)
        __syncthreads();
        float4 zs_k = zero4(), ai_k = zero4();             // step k-1's Z_S(y_{k-1}), A Z_I(y_{k-1}): stored one step late (training)

        // outputs of step j (trajectory point j+1, kept activations of grid point j, read-out at grid point j+1): nothing in
        // the launch consumes them, so they are issued UNDER the next step's gather (or after the last step)
        // (their base pointers are pinned to VGPRs: as wave-uniform values the compiler keeps them and every product of the
        //  grid index with the slab size in SGPRs, spills those, and a kernel with scratch cannot be captured on first use)
        const float* solv = a.sol; const float* keepv = a.keep;
        float* Sv = a.S; float* Iv = a.I; float* Rv = a.R;
        asm volatile("" : "+v"(solv), "+v"(keepv), "+v"(Sv), "+v"(Iv), "+v"(Rv));
        auto outputs = [&](int j) {
            if (!PRJ && valid) {
                if (solv) {
                    float* sn = const_cast<float*>(solv) + ((size_t)(j + 1) * 4 * slab + (size_t)row * 64 + 4 * sub);
                    st4s<true>(sn, ys); st4s<true>(sn + slab, yi); st4s<true>(sn + 2 * slab, yr);
                    if (j >= 1) {
                        if (keepv) st4s<true>(const_cast<float*>(keepv) + ((size_t)(3 * j + 2) * gn_keep_stride(rows) + (size_t)row * 64 + 4 * sub), ai_k);
                        else st4s<true>(const_cast<float*>(solv) + ((size_t)j * 4 * slab + 3 * slab + (size_t)row * 64 + 4 * sub), ai_k);
                    }
                }
                if (keepv) st4s<true>(const_cast<float*>(keepv) + ((size_t)(3 * j) * gn_keep_stride(rows) + (size_t)row * 64 + 4 * sub), zs_k);
            }
            const int slot = a.sched.slot[j];
            if (slot >= 0) {
                float pS, pI, pR;
                const float prj[4] = {pr.x, pr.y, pr.z, pr.w};
                readout64<PRJ>(ys, yi, yr, prj, sub, w3s, L + O_HD, L + O_HD + 4, L + O_HD + 8, pS, pI, pR);
                if (valid && sub == 0) {
                    const size_t o = (size_t)slot * rows + row;
                    Sv[o] = pS; Iv[o] = pI; Rv[o] = pR;
                }
            }
        };

        for (int k = 0; k < n_steps; ++k) {
            const float dt = a.sched.dt[k];
            float* const tab_cur = a.keep ? gn_keep_zi(a.keep, rows, k) : ((k & 1) ? a.Z1 : a.Z0);
            float* const tab_nxt = a.keep ? gn_keep_zi(a.keep, rows, k + 1) : ((k & 1) ? a.Z0 : a.Z1);
            if (k > 0) {
                // table k is complete once every workgroup of the group has published epoch k
                bool ok = true;
                if (threadIdx.x < 64) ok = pers_wait(flags, a.pp.wgs, ebase + (unsigned)k, err, lane);
                if (threadIdx.x < 64 && !ok) sh[2] = 0u;
                __syncthreads();
                if (sh[2] == 0u) return;
            }
            PS_STAMP(0)
            const float4 zs = *reinterpret_cast<const float4*>(T2S + ro);      // Z_S(y_k): the matrix phase behind the last flag
            // ---- gather (step k-1's outputs under its round trip) + SIR update (ode_nn_ngraph_sim.py:75-77) + Euler step
            // (1024-thread workgroups: the read-out's temporaries next to 32 registers of loads in flight do not fit 128
            //  VGPRs -- spills, and a kernel with scratch cannot be captured on first use -- so there it runs in front)
            if (NT == 4 && k > 0) outputs(k - 1);
            float4 acc = pers_gather<NM, DEPTH>(pers_rsrc(tab_cur, tbytes), m, nbt, lane_b, [&]() { if (NT < 4 && k > 0) outputs(k - 1); });
            if (HUBS) {
                pers_hub_partials<(NT == 4 ? 1 : 4)>(pers_rsrc(tab_cur, tbytes), itn, HI, HLmine, HP, sub, lane_b);
                __syncthreads();
                if (hs0 >= 0) acc = pers_hub_total(HP, hs0, hcnt, sub);
            }
            PS_STAMP(1)
            float4 dS, dI, dR;
            dS.x = nb * (acc.x * zs.x); dS.y = nb * (acc.y * zs.y); dS.z = nb * (acc.z * zs.z); dS.w = nb * (acc.w * zs.w);
            dR.x = gm * zi.x; dR.y = gm * zi.y; dR.z = gm * zi.z; dR.w = gm * zi.w;
            dI.x = -dS.x - dR.x; dI.y = -dS.y - dR.y; dI.z = -dS.z - dR.z; dI.w = -dS.w - dR.w;
            ys.x += dt * dS.x; ys.y += dt * dS.y; ys.z += dt * dS.z; ys.w += dt * dS.w;
            yi.x += dt * dI.x; yi.y += dt * dI.y; yi.z += dt * dI.z; yi.w += dt * dI.w;
            if (PRJ) {
                float4 w3r[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) w3r[q] = *reinterpret_cast<const float4*>(w3s + q * 64 + 4 * sub);
                float prj[4];
                prj[0] = pr.x + dt * (gm * row_sum16(fmaf(w3r[0].x, zi.x, fmaf(w3r[0].y, zi.y, fmaf(w3r[0].z, zi.z, w3r[0].w * zi.w)))));
                prj[1] = pr.y + dt * (gm * row_sum16(fmaf(w3r[1].x, zi.x, fmaf(w3r[1].y, zi.y, fmaf(w3r[1].z, zi.z, w3r[1].w * zi.w)))));
                prj[2] = pr.z + dt * (gm * row_sum16(fmaf(w3r[2].x, zi.x, fmaf(w3r[2].y, zi.y, fmaf(w3r[2].z, zi.z, w3r[2].w * zi.w)))));
                prj[3] = pr.w + dt * (gm * row_sum16(fmaf(w3r[3].x, zi.x, fmaf(w3r[3].y, zi.y, fmaf(w3r[3].z, zi.z, w3r[3].w * zi.w)))));
                pr = make_float4(prj[0], prj[1], prj[2], prj[3]);
            } else {
                yr.x += dt * dR.x; yr.y += dt * dR.y; yr.z += dt * dR.z; yr.w += dt * dR.w;
                // kept for the backward: Z_S(y_k) and A Z_I(y_k) [* Z_S (1 - Z_S) with a keep buffer]
                zs_k = zs;
                ai_k = acc;
                if (a.keep) ai_k = make_float4(acc.x * (zs.x * (1.0f - zs.x)), acc.y * (zs.y * (1.0f - zs.y)),
                                               acc.z * (zs.z * (1.0f - zs.z)), acc.w * (zs.w * (1.0f - zs.w)));
            }
            *reinterpret_cast<float4*>(TA + ro) = yi;
            *reinterpret_cast<float4*>(TB + ro) = ys;
            __syncthreads();
            PS_STAMP(2)
            // only Z_I(y_{k+1}) is on the way to the row store; Z_S(y_{k+1}) follows behind it (same chains either way)
            mfma_dual16<true, false>(TA, TB, Wslab, T2I, T2S, bias_l, fo, oo);
            __syncthreads();
            PS_STAMP(3)
            zi = *reinterpret_cast<const float4*>(T2I + ro);
            const bool last = k + 1 == n_steps;
            if (!last || a.keep) {
                const rsrc_t tn = pers_rsrc(tab_nxt, tbytes);
                if (valid) pers_st<STAUX>(tn, off, zi);
                if (a.keep && idx == 0 && team == 0 && tid < 16) pers_st<STAUX>(tn, zoff + 16u * tid, zero4());   // the fresh table's zero row
            }
            if (!last) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every storing wave drains, then the barrier, then ONE flag
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(flags + idx, ebase + (unsigned)k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                PS_STAMP(4)
                // Z_S(y_{k+1}) is read behind the next barrier: its matrix phase runs under the flag's flight (measured twice: in
                // FRONT of the flag -- also between the row stores and their vmcnt(0) wait -- it delays every workgroup of the group
                // by its full length: fb-social size B = 1 0.257 -> 0.322 ms, wiki-vote size 0.55 -> 0.59)
                mfma_dual16<false, true>(TA, TB, Wslab, T2I, T2S, bias_l, fo, oo);
                PS_STAMP(5)
            }
        }
        outputs(n_steps - 1);
        __syncthreads();                                    // the operand tiles are restaged by the next round
    }
#ifdef GN_PERS_PROF
    if (threadIdx.x == 0 && idx == 0 && gl == 0) for (int i = 0; i < 8; ++i) a.ctl->prof[i] = prof[i];
#endif
}
#undef PS_ACC

// --------------------------------------------------------------------------- host: row maps, plan, launch
// Slot s of the map for nt tiles per workgroup = lane group (s % (16 nt)) of workgroup s / (16 nt).  Rows are sorted by
// length (longest first, ties by id); hub rows are dealt singly, the others in quads, round-robin: the four rows of a wave have
// neighbouring lengths, and every workgroup gets rows from the whole length spectrum (equal bytes per CU).
static const int kPersMaxRows = 256 * 64;              // one resident grid: 256 workgroups x 64 rows
#define PERS_MAX_PARTIALS 128                          // partial-sum slots (256 B each) a workgroup may need for its hub rows
int gn_pers64_build(gnode_graph_s* g, const int32_t* rowptr_host) {
    for (int i = 0; i < 3; ++i) { g->persmap[i] = g->pershub[i] = g->perssegptr[i] = g->perssegitem[i] = nullptr; g->perslds[i] = 0; g->persitems[i] = 0; }
    if (g->n > kPersMaxRows) return 0;
    std::vector<int32_t> order((size_t)g->n);
    for (int32_t r = 0; r < g->n; ++r) order[r] = r;
    auto deg = [&](int32_t r) { return rowptr_host[r + 1] - rowptr_host[r]; };
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return deg(x) > deg(y); });
    auto up = [](int32_t** dst, const std::vector<int32_t>& v) -> hipError_t {
        hipError_t e = hipMalloc(dst, sizeof(int32_t) * std::max<size_t>(v.size(), 4));
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(*dst, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice);
    };
    for (int i = 0; i < 3; ++i) {
        const int nt = 1 << i, per_wg = 16 * nt, wgs = (g->n + per_wg - 1) / per_wg, quads_per_wg = 4 * nt;
        std::vector<int32_t> map((size_t)wgs * per_wg, -1);
        // Rows go to the workgroups longest first, each to the workgroup with the fewest EDGES so far that still has a slot
        // (longest-processing-time greedy): a step waits for its busiest workgroup, whose gather time follows the rows it has to
        // read -- hub rows included, their segments are summed by the owner's workgroup.  The two biggest hubs never share a
        // workgroup, the workgroup of a big hub gets short ordinary rows, and a workgroup's rows come out in descending
        // length, so the four rows of a wave are of neighbouring lengths (a wave issues loads as far as its longest row needs).
        std::vector<int> fill((size_t)wgs, 0);
        int32_t nh = 0;
        while (nh < g->n && deg(order[nh]) > GN_HUB_T) ++nh;
        if ((long)nh > (long)wgs * per_wg / 2) { continue; }   // (half the slots hubs: not a graph for this path)
        {
            typedef std::pair<long, int> WL;                   // (edges so far, workgroup)
            std::priority_queue<WL, std::vector<WL>, std::greater<WL>> heap;
            for (int wg = 0; wg < wgs; ++wg) heap.push(WL(0, wg));
            for (int32_t r = 0; r < g->n; ++r) {
                if (heap.empty()) return GNODE_ERR_ARG;        // cannot happen: wgs * per_wg >= n
                const WL w = heap.top();
                heap.pop();
                map[(size_t)w.second * per_wg + fill[w.second]++] = order[r];
                if (fill[w.second] < per_wg) heap.push(WL(w.first + std::max(1, deg(order[r])), w.second));
            }
        }
        (void)quads_per_wg;
        // hub rows: the segments of a workgroup's hubs are summed by that workgroup's own lane groups (partials through LDS),
        // dealt to the lane groups with the least gather work so far; a hub's partial slots are consecutive, in segment order
        std::vector<int32_t> hub((size_t)wgs * per_wg * 2, 0), segptr((size_t)wgs * per_wg * 2, 0), items;
        int max_slots = 0, max_items = 0;
        bool ok = true;
        for (int wg = 0; wg < wgs && ok; ++wg) {
            std::vector<long> load((size_t)per_wg, 0);
            std::vector<std::vector<int32_t>> mine((size_t)per_wg);
            int slots = 0;
            for (int s = 0; s < per_wg; ++s) {
                const int32_t r = map[(size_t)wg * per_wg + s];
                hub[((size_t)wg * per_wg + s) * 2] = -1;
                if (r >= 0 && deg(r) <= GN_HUB_T) load[s] = deg(r);
            }
            for (int s = 0; s < per_wg; ++s) {
                const int32_t r = map[(size_t)wg * per_wg + s];
                if (r < 0 || deg(r) <= GN_HUB_T) continue;
                const int32_t lo = rowptr_host[r], hi = rowptr_host[r + 1];
                hub[((size_t)wg * per_wg + s) * 2] = slots;
                hub[((size_t)wg * per_wg + s) * 2 + 1] = (hi - lo + HUB_SEG - 1) / HUB_SEG;
                for (int32_t e = lo; e < hi; e += HUB_SEG) {
                    int best = 0;
                    for (int t = 1; t < per_wg; ++t) if (load[t] < load[best]) best = t;
                    load[best] += HUB_SEG;
                    mine[best].push_back(e); mine[best].push_back(std::min(hi, e + HUB_SEG)); mine[best].push_back(slots++); mine[best].push_back(0);
                }
            }
            if (slots > PERS_MAX_PARTIALS) ok = false;
            for (int s = 0; s < per_wg; ++s) { if (mine[s].size() / 4 > PERS_MAX_ITEMS) ok = false; max_items = std::max(max_items, (int)(mine[s].size() / 4)); }
            max_slots = std::max(max_slots, slots);
            for (int s = 0; s < per_wg; ++s) {
                segptr[((size_t)wg * per_wg + s) * 2] = (int32_t)(items.size() / 4);
                segptr[((size_t)wg * per_wg + s) * 2 + 1] = (int32_t)(mine[s].size() / 4);
                items.insert(items.end(), mine[s].begin(), mine[s].end());
            }
        }
        if (!ok) continue;                                  // this tile count is not available for this graph (plan skips it)
        GN_HIP(up(&g->persmap[i], map));
        GN_HIP(up(&g->pershub[i], hub));
        GN_HIP(up(&g->perssegptr[i], segptr));
        GN_HIP(up(&g->perssegitem[i], items));
        g->perslds[i] = max_slots;
        g->persitems[i] = max_items;
    }
    return 0;
}
void gn_pers64_free(gnode_graph_s* g) {
    for (int i = 0; i < 3; ++i) {
        for (int32_t** q : {&g->persmap[i], &g->pershub[i], &g->perssegptr[i], &g->perssegitem[i]})
            if (*q) { (void)hipFree(*q); *q = nullptr; }
    }
}

bool gn_pers64_plan(const gnode_graph_s* g, long B, int n_steps, PersPlan* p) {
    if (n_steps < 1 || n_steps > 128 || B < 1) return false;
    const int n_xcc = 8;
    if (g->num_cu < 64 || g->num_cu % n_xcc) return false;
    const int slots = g->num_cu / n_xcc;
    if ((long)B * g->n >= (1L << 24)) return false;
    // the smallest tile count that holds the batch -- except that a graph whose biggest hub would give a lane group TWO segment
    // sums per step (a second ~2 us round of 32-row gathers every step) takes the next tile count when that halves the rounds
    // (fb-social size with a 738-edge row, B = 1: 8.7 -> 7.x us per step at 32 instead of 16 rows per workgroup)
    bool have = false;
    for (int nt = 1; nt <= 4; nt *= 2) {
        const int vi = nt == 1 ? 0 : nt == 2 ? 1 : 2;
        if (!g->persmap[vi]) continue;      // graph too large, or its hub rows need too many partial slots
        const int wgs = (g->n + 16 * nt - 1) / (16 * nt);
        PersPlan q;
        q.nt = nt; q.wgs = wgs; q.n_xcc = n_xcc; q.slots = slots;
        if (wgs <= slots) { q.span = 1; q.gpx = std::min(slots / wgs, PERS_FLAG_WORDS / 32 / n_xcc); q.per = wgs; q.concurrent = n_xcc * q.gpx; }   // (one 32-word flag line per group)
        else {
            int span = 2;
            while (span < n_xcc && wgs > span * slots) span *= 2;
            if (wgs > span * slots) continue;
            q.span = span; q.gpx = 1; q.per = (wgs + span - 1) / span; q.concurrent = n_xcc / span;
        }
        if (q.concurrent < B) continue;
        q.rounds = 1;
        q.fstride = (wgs + 31) / 32 * 32;
        if ((long)q.concurrent * q.fstride > PERS_FLAG_WORDS) continue;
        if (!have) { *p = q; have = true; if (g->persitems[vi] <= 1) return true; continue; }
        if (nt <= 2 && g->persitems[vi] < g->persitems[p->nt == 1 ? 0 : 1]) { *p = q; if (g->persitems[vi] <= 1) return true; }
    }
    return have;
}

size_t gn_pers64_ctl_bytes() { return gn_align(sizeof(PersCtl)); }

// Tickets, flags and the give-up word are zeroed in front of EVERY persistent launch -- by a kernel, not by hipMemsetAsync (see
// gn_zero_async: under graph replay the memset node left the tickets counting on, every workgroup found itself idle, and the
// trainer's replayed step either "ran" in 0.2 ms on stale outputs or waited out the 2 s give-up).
int gn_pers64_zero_ctl(void* ctl, hipStream_t st) { return gn_zero_async(ctl, sizeof(PersCtl), st); }

static size_t pers_lds_bytes(int nt, int partial_slots = PERS_MAX_PARTIALS) {
    const size_t need = sizeof(float) * ((size_t)64 * TS + 256 + 16 + (size_t)nt * 4 * 16 * TS + (size_t)partial_slots * 96 + (size_t)16 * nt * PERS_MAX_ITEMS);
    return std::max<size_t>(need, 84 * 1024);              // > half of the CU's 160 KB: ONE workgroup per CU
}

int gn_pers64_set_attributes() {
#define PS_ATTR(P, N, S) GN_HIP(hipFuncSetAttribute((const void*)k_pers64<P, N, S, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pers_lds_bytes(N))); \
                         GN_HIP(hipFuncSetAttribute((const void*)k_pers64<P, N, S, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pers_lds_bytes(N)));
    PS_ATTR(true, 1, false) PS_ATTR(true, 1, true) PS_ATTR(true, 2, false) PS_ATTR(true, 2, true) PS_ATTR(true, 4, false) PS_ATTR(true, 4, true)
    PS_ATTR(false, 1, false) PS_ATTR(false, 1, true) PS_ATTR(false, 2, false) PS_ATTR(false, 2, true) PS_ATTR(false, 4, false) PS_ATTR(false, 4, true)
#undef PS_ATTR
    return 0;
}

int gn_launch_pers64(const gnode_graph_s* g, const PersPlan& pl, long rows, const float* Y0, const float* PR0, float* Z0, float* Z1,
                     const float* W, const float* bias, const float* beta, const float* gamma, const float* dt_host,
                     const int* slot_host, int n_steps, const gnode_params* p, float* S, float* I, float* R, float* sol, float* keep,
                     void* ctl, bool ctl_is_zero, hipStream_t st) {
    PersArgs a;
    const int vi = pl.nt == 1 ? 0 : pl.nt == 2 ? 1 : 2;
    const bool hubs = g->n_hub > 0;
    a.rowhdr = g->rowhdr; a.col = g->col; a.rowmap = g->persmap[vi]; a.n = g->n;
    a.lds_slots = hubs ? g->perslds[vi] : 0;
    a.hubslot = hubs ? g->pershub[vi] : nullptr; a.segptr = hubs ? g->perssegptr[vi] : nullptr; a.segitem = hubs ? g->perssegitem[vi] : nullptr; a.B = (int)(rows / g->n); a.rows = (unsigned)rows;
    a.pp = pers_place_of(pl);
    a.Y0 = Y0; a.PR0 = PR0; a.beta = beta; a.gamma = gamma; a.Z0 = Z0; a.Z1 = Z1; a.keep = keep;
    a.W = W; a.bias = bias; a.w3 = p->linear3_weight; a.b3 = p->linear3_bias; a.w2 = p->linearS2_weight; a.b2 = p->linearS2_bias;
    a.S = S; a.I = I; a.R = R; a.sol = sol; a.ctl = (PersCtl*)ctl;
    a.sched.n_steps = n_steps;
    for (int k = 0; k < n_steps; ++k) { a.sched.dt[k] = dt_host[k]; a.sched.slot[k] = (short)slot_host[k]; }
    if (!ctl_is_zero)                                      // tickets, flags, give-up word: zeroed before EVERY launch (by the prologue launch where there is one)
        if (int e = gn_pers64_zero_ctl(ctl, st)) return e;
    const bool prj = PR0 != nullptr, sc1 = pl.span > 1;
    const dim3 grid((unsigned)(pl.n_xcc * pl.slots));
#define PS_GO(P, N, S) { if (hubs) hipLaunchKernelGGL((k_pers64<P, N, S, true>), grid, dim3(256 * N), pers_lds_bytes(N, g->perslds[vi]), st, a); \
                         else hipLaunchKernelGGL((k_pers64<P, N, S, false>), grid, dim3(256 * N), pers_lds_bytes(N, 0), st, a); }
#define PS_NT(P, S) { if (pl.nt == 1) PS_GO(P, 1, S) else if (pl.nt == 2) PS_GO(P, 2, S) else PS_GO(P, 4, S) }
    if (prj) { if (sc1) PS_NT(true, true) else PS_NT(true, false) }
    else { if (sc1) PS_NT(false, true) else PS_NT(false, false) }
#undef PS_NT
#undef PS_GO
    GN_LAUNCH_CHECK();
    return 0;
}
