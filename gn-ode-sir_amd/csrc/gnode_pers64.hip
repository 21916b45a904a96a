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
#include <algorithm>

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t pers_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
// AUX 16 = sc1 (agent-coherent: bypasses this CU's L1 / writes through), 0 = plain
template <int AUX> __device__ __forceinline__ float4 pers_ld(rsrc_t rs, unsigned off) {
    const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <int AUX> __device__ __forceinline__ void pers_st(rsrc_t rs, unsigned off, float4 v) {
    const v4f t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t), rs, off, 0, AUX);
}

#define PS_ACC(V) acc.x += V.x; acc.y += V.y; acc.z += V.z; acc.w += V.w;

// AI = sum of the row's neighbour rows of the table behind `tab`, ascending column order (the CPU scatter_add_ order of
// the reference, ode_nn_ngraph_sim.py:73): 16 rows in flight per lane group; neighbour k < 16 is lane k's `m` (a byte
// offset, or the table's zero row), 16 <= k < 32 lane (k-16)'s `m2`, beyond that the column list is walked.
__device__ __forceinline__ float4 pers_gather(rsrc_t tab, const int* __restrict__ col, unsigned base, unsigned zoff, unsigned m,
                                              unsigned m2, int start, int end, int sub, unsigned lane_b) {
    float4 acc = zero4();
    const int cnt = end - start;
#define PS_LD(K, M) pers_ld<16>(tab, (unsigned)row_bcast<(K) & 15>((int)(M)) + lane_b)
    {
        const float4 v0 = PS_LD(0, m), v1 = PS_LD(1, m), v2 = PS_LD(2, m), v3 = PS_LD(3, m), v4 = PS_LD(4, m), v5 = PS_LD(5, m),
                     v6 = PS_LD(6, m), v7 = PS_LD(7, m), v8 = PS_LD(8, m), v9 = PS_LD(9, m), v10 = PS_LD(10, m), v11 = PS_LD(11, m),
                     v12 = PS_LD(12, m), v13 = PS_LD(13, m), v14 = PS_LD(14, m), v15 = PS_LD(15, m);
        PS_ACC(v0) PS_ACC(v1) PS_ACC(v2) PS_ACC(v3) PS_ACC(v4) PS_ACC(v5) PS_ACC(v6) PS_ACC(v7)
        PS_ACC(v8) PS_ACC(v9) PS_ACC(v10) PS_ACC(v11) PS_ACC(v12) PS_ACC(v13) PS_ACC(v14) PS_ACC(v15)
    }
    if (__any(cnt > 16)) {
        const float4 v0 = PS_LD(0, m2), v1 = PS_LD(1, m2), v2 = PS_LD(2, m2), v3 = PS_LD(3, m2), v4 = PS_LD(4, m2), v5 = PS_LD(5, m2),
                     v6 = PS_LD(6, m2), v7 = PS_LD(7, m2);
        if (__any(cnt > 24)) {
            const float4 v8 = PS_LD(8, m2), v9 = PS_LD(9, m2), v10 = PS_LD(10, m2), v11 = PS_LD(11, m2), v12 = PS_LD(12, m2),
                         v13 = PS_LD(13, m2), v14 = PS_LD(14, m2), v15 = PS_LD(15, m2);
            PS_ACC(v0) PS_ACC(v1) PS_ACC(v2) PS_ACC(v3) PS_ACC(v4) PS_ACC(v5) PS_ACC(v6) PS_ACC(v7)
            PS_ACC(v8) PS_ACC(v9) PS_ACC(v10) PS_ACC(v11) PS_ACC(v12) PS_ACC(v13) PS_ACC(v14) PS_ACC(v15)
        } else {
            PS_ACC(v0) PS_ACC(v1) PS_ACC(v2) PS_ACC(v3) PS_ACC(v4) PS_ACC(v5) PS_ACC(v6) PS_ACC(v7)
        }
    }
    if (__any(cnt > 32)) {
        for (int e0 = start + 32; e0 < end; e0 += 16) {
            const int c2 = end - e0;
            const unsigned mm = (sub < c2) ? (base + (unsigned)col[e0 + sub]) * 256u : zoff;
            const float4 v0 = PS_LD(0, mm), v1 = PS_LD(1, mm), v2 = PS_LD(2, mm), v3 = PS_LD(3, mm), v4 = PS_LD(4, mm), v5 = PS_LD(5, mm),
                         v6 = PS_LD(6, mm), v7 = PS_LD(7, mm), v8 = PS_LD(8, mm), v9 = PS_LD(9, mm), v10 = PS_LD(10, mm), v11 = PS_LD(11, mm),
                         v12 = PS_LD(12, mm), v13 = PS_LD(13, mm), v14 = PS_LD(14, mm), v15 = PS_LD(15, mm);
            PS_ACC(v0) PS_ACC(v1) PS_ACC(v2) PS_ACC(v3) PS_ACC(v4) PS_ACC(v5) PS_ACC(v6) PS_ACC(v7)
            PS_ACC(v8) PS_ACC(v9) PS_ACC(v10) PS_ACC(v11) PS_ACC(v12) PS_ACC(v13) PS_ACC(v14) PS_ACC(v15)
        }
    }
#undef PS_LD
    return acc;
}

// Wave 0 of the workgroup waits until every workgroup of the group has published `epoch` (flags only grow inside a launch).
// Returns false on the give-up path (the caller leaves the kernel).  Callers follow it with __syncthreads().
__device__ __forceinline__ bool pers_wait(unsigned* flags, int wgs, unsigned epoch, unsigned* err, int lane) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        bool ok = true;
        for (int j = lane; j < wgs; j += 64)
            ok &= __hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {        // 2 s at 100 MHz
            if (lane == 0) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                             __hip_atomic_store(err + 1, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            return false;
        }
    }
}

template <bool PRJ, int NT, bool SC1ST>
__global__ __launch_bounds__(256 * NT) void k_pers64(const PersArgs a) {
    constexpr int STAUX = SC1ST ? 16 : 0;
    constexpr int O_W = 0, O_W3 = O_W + 64 * TS, O_T = O_W3 + 256, TEAM_F = 4 * 16 * TS;
    extern __shared__ __attribute__((aligned(16))) float L[];
    __shared__ unsigned sh[4];
    const int team = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int lane = tid & 63, w = tid >> 6, g = lane >> 4, sub = lane & 15;
    if (threadIdx.x == 0) {
        const unsigned xcc = ((unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11))) % (unsigned)a.n_xcc;   // HW_REG_XCC_ID
        sh[0] = xcc;
        sh[1] = atomicAdd(&a.ctl->ticket[xcc][0], 1u);
        sh[2] = 1u;
    }
    if (team == 0) load_W_to_lds<false>(a.W, L + O_W);
    if (threadIdx.x < 256) L[O_W3 + threadIdx.x] = a.w3[threadIdx.x];
    __syncthreads();
    const int xcc = (int)sh[0], tk = (int)sh[1];
    int gl, idx;                                           // concurrent group, workgroup inside the group
    bool active;
    if (a.span == 1) { const int gi = tk / a.wgs; idx = tk - gi * a.wgs; gl = gi * a.n_xcc + xcc; active = tk < a.slots && gi < a.gpx; }
    else { gl = xcc / a.span; idx = (xcc % a.span) * a.per + tk; active = tk < a.per && idx < a.wgs; }
    if (!active) return;
    unsigned* const flags = a.ctl->flags + (size_t)gl * a.fstride;
    unsigned* const err = a.ctl->error;

    const float bias_l = a.bias[16 * w + (lane & 15)];
    float* const TA = L + O_T + team * TEAM_F;             // Y_I' rows (MFMA operand)
    float* const TB = TA + 16 * TS;                        // Y_S' rows
    float* const T2I = TB + 16 * TS;                       // Z_I'
    float* const T2S = T2I + 16 * TS;                      // Z_S'
    const float* const Wslab = L + O_W + 16 * w * TS;
    const float* const w3s = L + O_W3;
    const int lr = 4 * w + g;
    const unsigned lane_b = 16u * sub;
    const int ro = lr * TS + 4 * sub;
    const int fo = (lane & 15) * TS + 16 * (lane >> 4);
    const int oo = 4 * (lane >> 4) * TS + 16 * w + (lane & 15);
    const unsigned rows = a.rows;
    const unsigned zoff = rows * 256u;
    const unsigned tbytes = (rows + 1u) * 256u;
    const size_t slab = (size_t)rows * 64;
    const int n_steps = a.sched.n_steps;
    const int node = idx * (16 * NT) + team * 16 + lr;
    const bool valid = node < a.n;

    for (int round = 0; round < a.rounds; ++round) {
        const int b = round * a.concurrent + gl;
        if (b >= a.B) break;
        const unsigned ebase = (unsigned)round * (unsigned)n_steps;
        const unsigned base = (unsigned)b * (unsigned)a.n;
        const unsigned row = valid ? base + (unsigned)node : 0u;
        const unsigned off = row * 256u + lane_b;
        // ---- the row's loop-invariant data: extent, neighbour ids (as table byte offsets), beta, gamma
        int start = 0, end = 0;
        unsigned m = zoff, m2 = zoff;
        if (valid) {
            const int* h = a.rowhdr + (size_t)node * 20;
            start = h[0]; end = h[1];
            const int d = end - start;
            if (sub < d) m = (base + (unsigned)h[4 + sub]) * 256u;
            if (16 + sub < d) m2 = (base + (unsigned)a.col[start + 16 + sub]) * 256u;
        }
        float4 ys = zero4(), yi = zero4(), yr = zero4(), pr = zero4(), zi = zero4(), zs;
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
        zs = *reinterpret_cast<const float4*>(T2S + ro);

        for (int k = 0; k < n_steps; ++k) {
            const float dt = a.sched.dt[k];
            const int slot = a.sched.slot[k];
            float* const tab_cur = a.keep ? gn_keep_zi(a.keep, rows, k) : ((k & 1) ? a.Z1 : a.Z0);
            float* const tab_nxt = a.keep ? gn_keep_zi(a.keep, rows, k + 1) : ((k & 1) ? a.Z0 : a.Z1);
            if (k > 0) {
                // table k is complete once every workgroup of the group has published epoch k
                bool ok = true;
                if (threadIdx.x < 64) ok = pers_wait(flags, a.wgs, ebase + (unsigned)k, err, lane);
                if (threadIdx.x < 64 && !ok) sh[2] = 0u;
                __syncthreads();
                if (sh[2] == 0u) return;
            }
            // ---- gather + SIR update (ode_nn_ngraph_sim.py:75-77) + Euler step
            const float4 acc = pers_gather(pers_rsrc(tab_cur, tbytes), a.col, base, zoff, m, m2, start, end, sub, lane_b);
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
                for (int q = 0; q < 4; ++q) w3r[q] = *reinterpret_cast<const float4*>(w3s + q * 64 + 4 * sub);
                prj[0] = pr.x + dt * (gm * row_sum16(fmaf(w3r[0].x, zi.x, fmaf(w3r[0].y, zi.y, fmaf(w3r[0].z, zi.z, w3r[0].w * zi.w)))));
                prj[1] = pr.y + dt * (gm * row_sum16(fmaf(w3r[1].x, zi.x, fmaf(w3r[1].y, zi.y, fmaf(w3r[1].z, zi.z, w3r[1].w * zi.w)))));
                prj[2] = pr.z + dt * (gm * row_sum16(fmaf(w3r[2].x, zi.x, fmaf(w3r[2].y, zi.y, fmaf(w3r[2].z, zi.z, w3r[2].w * zi.w)))));
                prj[3] = pr.w + dt * (gm * row_sum16(fmaf(w3r[3].x, zi.x, fmaf(w3r[3].y, zi.y, fmaf(w3r[3].z, zi.z, w3r[3].w * zi.w)))));
                pr = make_float4(prj[0], prj[1], prj[2], prj[3]);
            } else {
                yr.x += dt * dR.x; yr.y += dt * dR.y; yr.z += dt * dR.z; yr.w += dt * dR.w;
            }
            // kept for the backward (training): Z_S(y_k) and A Z_I(y_k) [* Z_S (1 - Z_S) with a keep buffer]; stored after the flag
            const float4 zs_k = zs;
            float4 ai_k = acc;
            if (!PRJ && a.keep) ai_k = make_float4(acc.x * (zs.x * (1.0f - zs.x)), acc.y * (zs.y * (1.0f - zs.y)),
                                                   acc.z * (zs.z * (1.0f - zs.z)), acc.w * (zs.w * (1.0f - zs.w)));
            *reinterpret_cast<float4*>(TA + ro) = yi;
            *reinterpret_cast<float4*>(TB + ro) = ys;
            __syncthreads();
            mfma_dual16<true, true>(TA, TB, Wslab, T2I, T2S, bias_l, fo, oo);    // Z_I(y_{k+1}), Z_S(y_{k+1})
            __syncthreads();
            zi = *reinterpret_cast<const float4*>(T2I + ro);
            zs = *reinterpret_cast<const float4*>(T2S + ro);
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
            }
            // ---- under the barrier's latency: trajectory point, kept activations, read-out head
            if (!PRJ && valid) {
                if (a.sol) {
                    float* sn = a.sol + (size_t)(k + 1) * 4 * slab;
                    st4so<true>(sn, off, ys); st4so<true>(sn + slab, off, yi); st4so<true>(sn + 2 * slab, off, yr);
                    if (k >= 1) {
                        if (a.keep) st4so<true>(gn_keep_ps(a.keep, rows, k), off, ai_k);
                        else st4so<true>(a.sol + (size_t)k * 4 * slab + 3 * slab, off, ai_k);
                    }
                }
                if (a.keep) st4so<true>(gn_keep_zs(a.keep, rows, k), off, zs_k);
            }
            if (slot >= 0) {
                float pS, pI, pR;
                readout64<PRJ>(ys, yi, yr, prj, sub, w3s, a.b3, a.w2, a.b2, pS, pI, pR);
                if (valid && sub == 0) {
                    const size_t o = (size_t)slot * rows + row;
                    a.S[o] = pS; a.I[o] = pI; a.R[o] = pR;
                }
            }
        }
        __syncthreads();                                    // the operand tiles are restaged by the next round
    }
}
#undef PS_ACC

// --------------------------------------------------------------------------- host: plan + launch
bool gn_pers64_plan(const gnode_graph_s* g, long B, int n_steps, PersPlan* p) {
    if (n_steps < 1 || n_steps > 128 || B < 1 || g->n_hub > 0) return false;
    const int n_xcc = 8;
    if (g->num_cu < 64 || g->num_cu % n_xcc) return false;
    const int slots = g->num_cu / n_xcc;
    if ((long)B * g->n >= (1L << 24)) return false;
    for (int nt = 1; nt <= 4; nt *= 2) {
        const int wgs = (g->n + 16 * nt - 1) / (16 * nt);
        PersPlan q;
        q.nt = nt; q.wgs = wgs; q.n_xcc = n_xcc; q.slots = slots;
        if (wgs <= slots) { q.span = 1; q.gpx = slots / wgs; q.per = wgs; q.concurrent = n_xcc * q.gpx; }
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
        *p = q;
        return true;
    }
    return false;
}

size_t gn_pers64_ctl_bytes() { return gn_align(sizeof(PersCtl)); }

static size_t pers_lds_bytes(int nt) {
    const size_t need = sizeof(float) * ((size_t)64 * TS + 256 + (size_t)nt * 4 * 16 * TS);
    return std::max<size_t>(need, 84 * 1024);              // > half of the CU's 160 KB: ONE workgroup per CU
}

int gn_pers64_set_attributes() {
#define PS_ATTR(P, N, S) GN_HIP(hipFuncSetAttribute((const void*)k_pers64<P, N, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pers_lds_bytes(N)));
    PS_ATTR(true, 1, false) PS_ATTR(true, 1, true) PS_ATTR(true, 2, false) PS_ATTR(true, 2, true) PS_ATTR(true, 4, false) PS_ATTR(true, 4, true)
    PS_ATTR(false, 1, false) PS_ATTR(false, 1, true) PS_ATTR(false, 2, false) PS_ATTR(false, 2, true) PS_ATTR(false, 4, false) PS_ATTR(false, 4, true)
#undef PS_ATTR
    return 0;
}

int gn_launch_pers64(const gnode_graph_s* g, const PersPlan& pl, long rows, const float* Y0, const float* PR0, float* Z0, float* Z1,
                     const float* W, const float* bias, const float* beta, const float* gamma, const float* dt_host,
                     const int* slot_host, int n_steps, const gnode_params* p, float* S, float* I, float* R, float* sol, float* keep,
                     void* ctl, hipStream_t st) {
    PersArgs a;
    a.rowhdr = g->rowhdr; a.col = g->col; a.n = g->n; a.B = (int)(rows / g->n); a.rows = (unsigned)rows;
    a.wgs = pl.wgs; a.span = pl.span; a.gpx = pl.gpx; a.per = pl.per; a.slots = pl.slots; a.n_xcc = pl.n_xcc;
    a.rounds = pl.rounds; a.concurrent = pl.concurrent; a.fstride = pl.fstride;
    a.Y0 = Y0; a.PR0 = PR0; a.beta = beta; a.gamma = gamma; a.Z0 = Z0; a.Z1 = Z1; a.keep = keep;
    a.W = W; a.bias = bias; a.w3 = p->linear3_weight; a.b3 = p->linear3_bias; a.w2 = p->linearS2_weight; a.b2 = p->linearS2_bias;
    a.S = S; a.I = I; a.R = R; a.sol = sol; a.ctl = (PersCtl*)ctl;
    a.sched.n_steps = n_steps;
    for (int k = 0; k < n_steps; ++k) { a.sched.dt[k] = dt_host[k]; a.sched.slot[k] = (short)slot_host[k]; }
    GN_HIP(hipMemsetAsync(ctl, 0, sizeof(PersCtl), st));  // tickets, flags, give-up word: zeroed before EVERY launch (a memset node under capture)
    const bool prj = PR0 != nullptr, sc1 = pl.span > 1;
    const dim3 grid((unsigned)(pl.n_xcc * pl.slots));
#define PS_GO(P, N, S) hipLaunchKernelGGL((k_pers64<P, N, S>), grid, dim3(256 * N), pers_lds_bytes(N), st, a)
#define PS_NT(P, S) { if (pl.nt == 1) PS_GO(P, 1, S); else if (pl.nt == 2) PS_GO(P, 2, S); else PS_GO(P, 4, S); }
    if (prj) { if (sc1) PS_NT(true, true) else PS_NT(true, false) }
    else { if (sc1) PS_NT(false, true) else PS_NT(false, false) }
#undef PS_NT
#undef PS_GO
    GN_LAUNCH_CHECK();
    return 0;
}
