// Device-side pieces of the persistent one-launch kernels (gnode_pers64.hip: forward, gnode_pers64_bwd.hip: adjoint):
// agent-coherent buffer accesses, the leveled row gather, the group barrier's poll, workgroup placement by XCC_ID.
#pragma once
#include "gnode_mfma64.h"
#include "gnode_pers64.h"

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

// Diagnostic build (-DGN_PERS_PROF, tools/bench_persist.py --prof): thread 0 of every group's workgroup 0 sums the 100 MHz
// clock between the phases of a step into ctl->prof (memory nothing else reads); in the product build no stamp executes.
#ifdef GN_PERS_PROF
#define PS_STAMP(I) { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); prof[I] += t_ - tp; tp = t_; } }
#else
#define PS_STAMP(I)
#endif

#define PS_ACC(V) acc.x += V.x; acc.y += V.y; acc.z += V.z; acc.w += V.w;

// A neighbour slot the row does not have carries PS_OOB: the buffer load's range check answers 0 WITHOUT a memory access
// (the per-step kernels point such slots at the table's zero row, which costs a real 256-B read each -- at 64 rows per CU
// those reads were half of the texture-unit time of a step).  Adding +0 leaves a sum's bits alone.
#define PS_OOB 0xFFFF0000u

// N loads of consecutive neighbour slots K, K+1, ...: slot k is lane (k & 15)'s m[k >> 4] (a table byte offset)
// 8 loads of consecutive neighbour slots K .. K+7: slot k is lane (k & 15)'s m[k >> 4] (a table byte offset)
template <int K, int N, int NM>
struct PsBatch {
    static __device__ __forceinline__ void load(float4* v, rsrc_t tab, const unsigned (&m)[NM], unsigned lane_b) {
        v[0] = pers_ld<16>(tab, (unsigned)row_bcast<K & 15>((int)m[K >> 4]) + lane_b);
        PsBatch<K + 1, N - 1, NM>::load(v + 1, tab, m, lane_b);
    }
};
template <int K, int NM>
struct PsBatch<K, 0, NM> { static __device__ __forceinline__ void load(float4*, rsrc_t, const unsigned (&)[NM], unsigned) {} };

// One WAVE of loads = up to DEPTH batches of 8 slots, starting at batch J0: batches are issued as far as the longest of the
// wave's four rows needs (wave-uniform count nb: a load instruction costs the CU's texture unit 16 cycles whatever its lanes
// fetch, and with one workgroup per CU that unit is what a step's gather waits for); then `under()` -- ONE copy of it --,
// then the sums in slot order.  Registers of batches that were not issued are never read (same nb on both sides).
template <int J0, int J, int NB, int NM>
struct PsIssue {
    static __device__ __forceinline__ void run(float4* v, int nb, rsrc_t tab, const unsigned (&m)[NM], unsigned lane_b) {
        if (J == 0 || nb > J) PsBatch<8 * (J0 + J), 8, NM>::load(v + 8 * J, tab, m, lane_b);
        PsIssue<J0, J + 1, NB, NM>::run(v, nb, tab, m, lane_b);
    }
};
template <int J0, int NB, int NM>
struct PsIssue<J0, NB, NB, NM> { static __device__ __forceinline__ void run(float4*, int, rsrc_t, const unsigned (&)[NM], unsigned) {} };

template <int J0, int DEPTH, int NM, class F>
__device__ __forceinline__ void ps_wave(float4& acc, int nbt, rsrc_t tab, const unsigned (&m)[NM], unsigned lane_b, F& under) {
    constexpr int NB = (2 * NM - J0) < DEPTH ? (2 * NM - J0) : DEPTH;      // batches this wave can hold ids for
    float4 v[8 * NB];
    const int nb = nbt - J0;                                               // batches of this wave the longest row needs (>= 1)
    PsIssue<J0, 0, NB, NM>::run(v, nb, tab, m, lane_b);
    __builtin_amdgcn_sched_barrier(0);
    under();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NB; ++j)
        if (j == 0 || nb > j) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { PS_ACC(v[8 * j + k]) }
        }
}

// the waves behind the first one, as far as the register-held ids go
template <int V, int NV, int DEPTH, int NM>
struct PsRest {
    static __device__ __forceinline__ void run(float4& acc, int nbt, rsrc_t tab, const unsigned (&m)[NM], unsigned lane_b) {
        if (nbt > DEPTH * V) {
            auto nothing = []() {};
            ps_wave<DEPTH * V, DEPTH, NM>(acc, nbt, tab, m, lane_b, nothing);
            PsRest<V + 1, NV, DEPTH, NM>::run(acc, nbt, tab, m, lane_b);
        }
    }
};
template <int NV, int DEPTH, int NM>
struct PsRest<NV, NV, DEPTH, NM> { static __device__ __forceinline__ void run(float4&, int, rsrc_t, const unsigned (&)[NM], unsigned) {} };

// batches of 8 neighbour slots the longest of the wave's four rows needs (wave-uniform, >= 1).  A row's length never changes:
// callers compute it ONCE per sample -- tested inside the step loop, each of the 2 NM thresholds becomes a loop-invariant
// 64-bit mask the compiler keeps live in SGPRs across the whole integration.
template <int NM>
__device__ __forceinline__ int pers_batches(int cnt) {
    int nbt = 1;
#pragma unroll
    for (int j = 1; j < 2 * NM; ++j) nbt += __any(cnt > 8 * j) ? 1 : 0;
    return __builtin_amdgcn_readfirstlane(nbt);
}

// AI = sum of the row's neighbour rows of the table behind `tab`, ascending column order (the CPU scatter_add_ order of
// the reference, ode_nn_ngraph_sim.py:73), 8 DEPTH rows in flight per lane group.  All of the row's neighbour ids live in
// registers (m: 16 NM >= the hub threshold; a hub row's own gather is empty, a segment has 32).  `under()` runs between
// the issue of the first wave of loads and its first use: independent work (the previous step's read-out and streamed
// stores) travels under the gather's round trip.  nbt: pers_batches() of the row lengths.
template <int NM, int DEPTH, class F>
__device__ __forceinline__ float4 pers_gather(rsrc_t tab, const unsigned (&m)[NM], int nbt, unsigned lane_b, F&& under) {
    float4 acc = zero4();
    ps_wave<0, DEPTH, NM>(acc, nbt, tab, m, lane_b, under);
    PsRest<1, (2 * NM + DEPTH - 1) / DEPTH, DEPTH, NM>::run(acc, nbt, tab, m, lane_b);
    return acc;
}

// Wave 0 of the workgroup waits until every workgroup of the group has published `epoch` (flags only grow inside a launch).
// Returns false on the give-up path (the caller leaves the kernel).  Callers follow it with __syncthreads().
__device__ __forceinline__ bool pers_wait(unsigned* flags, int wgs, unsigned epoch, unsigned* err, int lane) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // a poll is one fabric round trip; three sweeps stay in flight a third of a round trip apart, so the last flag is seen
    // one round trip after it lands instead of up to two
    // (a lane's words of one sweep are loaded before any is compared: the compares of sweep 0 come after sweep 2's loads)
    auto sweep = [&](unsigned (&f)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = (lane + 64 * q < wgs) ? __hip_atomic_load(flags + lane + 64 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
    };
    auto all_ge = [&](const unsigned (&f)[4]) { return min(min(f[0], f[1]), min(f[2], f[3])) >= epoch; };
    for (;;) {
#ifndef GN_PERS_POLL3
#define GN_PERS_POLL3 0
#endif
#ifndef GN_PERS_SLEEP
#define GN_PERS_SLEEP 1
#endif
        unsigned f0[4], f1[4], f2[4];
        sweep(f0);
        if (GN_PERS_POLL3) { __builtin_amdgcn_s_sleep(3); sweep(f1); __builtin_amdgcn_s_sleep(3); sweep(f2); }
        else __builtin_amdgcn_s_sleep(GN_PERS_SLEEP);
        const bool o0 = all_ge(f0), o1 = GN_PERS_POLL3 && all_ge(f1), o2 = GN_PERS_POLL3 && all_ge(f2);
        if (__all(o0) || __all(o1) || __all(o2)) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {        // 2 s at 100 MHz
            if (lane == 0) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                             __hip_atomic_store(err + 1, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            return false;
        }
    }
}


// Hub rows (longer than GN_HUB_T; gnode_hub.hip has the story): the row map deals a graph's hubs evenly to the workgroups, and
// a hub's <= 32-edge segments are summed by lane groups of ITS OWN workgroup (the plan's work lists), partials through LDS --
// the same segments, the same ascending sums and the same segment-order total as k_hub_seg + the consumers of the per-step
// kernels, so the bits agree, and no second group barrier is needed.  This lane group's items: [it0, it0 + itn).
#define PERS_MAX_ITEMS 8                               // segment sums one lane group may be given (plan-time bound)
// LDS behind the tiles (S = the plan's partial slots per workgroup): partials [S][64] floats | neighbour ids of every segment
// as table byte offsets [S][32] | per lane group its items [PERS_MAX_ITEMS] as (slot | edges << 16).  The ids and item lists
// are loop-invariant: staged once per sample, so a segment sum is ONE round trip per step (32 rows in flight where the
// registers allow), not descriptor -> column ids -> rows.
__device__ __forceinline__ size_t pers_hub_lds_floats(int S, int lane_groups) { return (size_t)S * 96 + (size_t)lane_groups * PERS_MAX_ITEMS; }
__device__ __forceinline__ void pers_hub_stage(const int* __restrict__ col, const int* __restrict__ items, int it0, int itn, unsigned base,
                                               unsigned* __restrict__ HI, unsigned* __restrict__ HLmine, int sub) {
    for (int t = 0; t < itn; ++t) {
        const int* it = items + 4 * (size_t)(it0 + t);
        const int lo = it[0], cnt = it[1] - lo, slot = it[2];
        HI[slot * 32 + sub] = (sub < cnt) ? (base + (unsigned)col[lo + sub]) * 256u : PS_OOB;
        HI[slot * 32 + 16 + sub] = (16 + sub < cnt) ? (base + (unsigned)col[lo + 16 + sub]) * 256u : PS_OOB;
        if (sub == 0) HLmine[t] = (unsigned)slot | ((unsigned)cnt << 16);
    }
}
template <int DEPTH>
__device__ __forceinline__ void pers_hub_partials(rsrc_t tab, int itn, const unsigned* __restrict__ HI,
                                                  const unsigned* __restrict__ HLmine, float* __restrict__ P, int sub, unsigned lane_b) {
    for (int t = 0; __any(t < itn); ++t) {
        int cnt = 0, slot = 0;
        if (t < itn) { const unsigned d = HLmine[t]; slot = (int)(d & 0xFFFFu); cnt = (int)(d >> 16); }
        unsigned ms[2] = {PS_OOB, PS_OOB};
        if (t < itn) { ms[0] = HI[slot * 32 + sub]; ms[1] = HI[slot * 32 + 16 + sub]; }
        const float4 part = pers_gather<2, DEPTH>(tab, ms, pers_batches<2>(cnt), lane_b, []() {});
        if (t < itn) *reinterpret_cast<float4*>(P + (size_t)slot * 64 + 4 * sub) = part;
    }
}
// the hub row's sum: its partials in segment order (what the per-step consumers do: hs = 0; hs += partial_s)
__device__ __forceinline__ float4 pers_hub_total(const float* __restrict__ P, int hs0, int hcnt, int sub) {
    float4 hs = zero4();
    for (int s = 0; s < hcnt; ++s) {
        const float4 u = *reinterpret_cast<const float4*>(P + (size_t)(hs0 + s) * 64 + 4 * sub);
        hs.x += u.x; hs.y += u.y; hs.z += u.z; hs.w += u.w;
    }
    return hs;
}

// Which sample ("group" gl) and which of its workgroups (idx) this workgroup is: read the XCD from the hardware register,
// draw a ticket there (PersPlan in gnode_pers64.h).  `sh`: 4 words of LDS; ends with a workgroup barrier.  false: idle.
__device__ __forceinline__ bool pers_place(const PersPlace& pp, PersCtl* ctl, unsigned* sh, int& gl, int& idx) {
    if (threadIdx.x == 0) {
        const unsigned xcc = ((unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11))) % (unsigned)pp.n_xcc;   // HW_REG_XCC_ID
        sh[0] = xcc;
        sh[1] = atomicAdd(&ctl->ticket[xcc][0], 1u);
        sh[2] = 1u;
    }
    __syncthreads();
    const int xcc = (int)sh[0], tk = (int)sh[1];
    if (pp.span == 1) { const int gi = tk / pp.wgs; idx = tk - gi * pp.wgs; gl = gi * pp.n_xcc + xcc; return tk < pp.slots && gi < pp.gpx; }
    gl = xcc / pp.span; idx = (xcc % pp.span) * pp.per + tk;
    return tk < pp.per && idx < pp.wgs;
}
