// Monte-Carlo SIR label generator for MI355X (gfx950).
//
// Restates sir_torch (reference ode_nn.py:30-88): `sims` independent discrete-time
// SIR trajectories; per step every (infected u -> susceptible v) directed edge
// fires with probability beta and every infected node recovers with probability
// gamma, both decided on the pre-step state (:60-78); per-(t, node) membership
// counts are accumulated (:80-82) with row 0 ASSIGNED (:55-56).
//
// The reference runs ~20 torch launches and >= 4 host syncs per (sim, step) and tests EVERY directed edge
// against the infected set each step (`isin` over all 2E rows, ode_nn.py:61).  Here one workgroup owns one
// trajectory and keeps its FRONTIER -- the list of currently infected nodes -- next to an "ever infected" bitmap in
// LDS; a step walks only the CSR rows of the listed nodes (one 16-lane group per node, very long rows by the whole
// workgroup), draws the recovery coin of each listed node, and builds the next list in the same pass: work per
// step is the frontier's out-degree, not nnz (k_sir_frontier).  Only the two EVENTS a node can have (infection
// step, recovery step) reach memory, as integer atomics; a final pass turns the event histograms into the S/I/R
// counts by a prefix sum over time.  Coins are Philox4x32-10 keyed (CSR position | node, step, trajectory, kind),
// so which edges are visited, in which order, by which lane, or on which GPU cannot change a count: all arithmetic
// is integer, results are bit-exact against the edge-scan statement of the same model (oracle: sir_philox; the
// edge-parallel kernel k_sir_philox below remains for graphs whose lists do not fit and as the in-library
// cross-check of the tests).
#include "gnode_common.h"
#include <algorithm>

enum : uint8_t { ST_S = 0, ST_I = 1, ST_R = 2 };

// --------------------------------------------------------------------------- Philox4x32-10
// FOUR consecutive items (CSR positions for infection coins, node ids for recovery coins) share one block: the coin of item
// pos is word (pos & 3) of philox(ctr = (pos >> 2, step, trajectory, kind), key).  Round 2 drew one block per coin and kept
// one word of four: 40 quarter-rate 32-bit multiplies per coin, the kernel's main cost (oracle: philox_coin).
__device__ __forceinline__ void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ uint32_t philox_coin(uint32_t pos, uint32_t it, uint32_t sim, uint32_t kind, uint32_t k0, uint32_t k1) {
    uint32_t w[4];
    philox_block(pos >> 2, it, sim, kind, k0, k1, w);
    const uint32_t j = pos & 3u;
    return j == 0 ? w[0] : j == 1 ? w[1] : j == 2 ? w[2] : w[3];
}

// --------------------------------------------------------------------------- production kernel
// hist: uint32 [2][T][n]: [0] = infection events (t = 0 for seeds), [1] = recovery events.
#ifndef GN_SIR_UNROLL
#define GN_SIR_UNROLL 8
#endif
template <bool STATE_IN_LDS>
__global__ __launch_bounds__(1024) void k_sir_philox(const int* __restrict__ src, const int* __restrict__ dst, long nnz,
                                                    int n, const int* __restrict__ seeds, int n_seeds,
                                                    unsigned long long thr_beta, unsigned long long thr_gamma,
                                                    long sims, long sim_offset, int T, uint32_t k0, uint32_t k1,
                                                    uint32_t* __restrict__ hist, uint8_t* __restrict__ gstate) {
    extern __shared__ uint8_t smem[];
    uint8_t* state = STATE_IN_LDS ? smem : gstate + (size_t)blockIdx.x * 2 * n;
    uint8_t* flag = state + n;
    uint32_t* hinf = hist;
    uint32_t* hrec = hist + (size_t)T * n;
    const int nthr = blockDim.x;                  // 256 for small graphs, 1024 when the LDS state allows one workgroup per CU
    for (long s = blockIdx.x; s < sims; s += gridDim.x) {
        const uint32_t sim = (uint32_t)(sim_offset + s);
        for (int v = threadIdx.x; v < n; v += nthr) { state[v] = ST_S; flag[v] = 0; }
        __syncthreads();
        for (int j = threadIdx.x; j < n_seeds; j += nthr) state[seeds[j]] = ST_I;   // duplicates: same value
        __syncthreads();
        for (int j = threadIdx.x; j < n_seeds; j += nthr) {
            // one infection event at t=0 per distinct seed node
            const int v = seeds[j];
            bool first = true;
            for (int q = 0; q < j; ++q) first = first && (seeds[q] != v);
            if (first) atomicAdd(&hinf[v], 1u);
        }
        for (int it = 1; it < T; ++it) {
            // GN_SIR_UNROLL source ids in flight per thread: the scan is a chain of (global id load -> LDS state read)
            // pairs, latency-bound when issued one at a time.  Wiki-vote-size graph, 10 000 sims x T = 20: 38.8 ms at 1,
            // 25.9 ms at 4, 23.6 ms at 8.  Tried on top, no gain: 16-bit ids (23.4 ms: not bound by the id bytes); four
            // consecutive edges per lane sharing one Philox block (32.5 ms: a wave then spans ~9 rows instead of ~2 and
            // runs mostly half-empty).
#define GN_EDGE(ON, E)                                                                                        \
                if (ON) {                                                                                     \
                    const int v = dst[E];                                                                     \
                    if (state[v] == ST_S &&                                                                   \
                        (unsigned long long)philox_coin((uint32_t)(E), (uint32_t)it, sim, 0u, k0, k1) < thr_beta) \
                        flag[v] = 1;                                                                          \
                }
            long e = threadIdx.x;
            for (; e + (long)(GN_SIR_UNROLL - 1) * nthr < nnz; e += (long)GN_SIR_UNROLL * nthr) {
                int u[GN_SIR_UNROLL]; bool inf[GN_SIR_UNROLL];
#pragma unroll
                for (int q = 0; q < GN_SIR_UNROLL; ++q) u[q] = src[e + (long)q * nthr];
#pragma unroll
                for (int q = 0; q < GN_SIR_UNROLL; ++q) inf[q] = state[u[q]] == ST_I;
#pragma unroll
                for (int q = 0; q < GN_SIR_UNROLL; ++q) GN_EDGE(inf[q], e + (long)q * nthr)
            }
            for (; e < nnz; e += nthr) {
                const bool i0 = state[src[e]] == ST_I;
                GN_EDGE(i0, e)
            }
#undef GN_EDGE
            for (int u = threadIdx.x; u < n; u += nthr)
                if (state[u] == ST_I &&
                    (unsigned long long)philox_coin((uint32_t)u, (uint32_t)it, sim, 1u, k0, k1) < thr_gamma)
                    flag[u] = 2;
            __syncthreads();
            int any = 0;
            for (int v = threadIdx.x; v < n; v += nthr) {
                const uint8_t f = flag[v];
                if (f == 1) { state[v] = ST_I; atomicAdd(&hinf[(size_t)it * n + v], 1u); }
                else if (f == 2) { state[v] = ST_R; atomicAdd(&hrec[(size_t)it * n + v], 1u); }
                flag[v] = 0;
                any |= (state[v] == ST_I);
            }
            if (!__syncthreads_or(any)) break;      // epidemic over: no further events in this trajectory
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------- frontier-driven kernel
// LDS: ever-infected bitmap (n bits) and, when they fit (uint16 ids: n up to ~25 000 -- every graph of the reference's
// multi-graph experiment but enron and epinions), the node lists: current / next frontier and the rows too long for
// one lane group.  Larger graphs keep the three lists in the caller's workspace (int32 ids, one set per workgroup).
// GN_SIR_BIGROW (gnode_common.h): rows longer than this are walked by the whole workgroup
// COUNT: the profiling instantiation (gnode_sir_mc_philox_counted) also tallies Philox blocks, coins and CSR entries read.
template <typename IdT, bool LISTS_IN_LDS, bool COUNT>
__global__ __launch_bounds__(1024) void k_sir_frontier(const int* __restrict__ rowptr, const int* __restrict__ col, int n,
                                                      const int* __restrict__ seeds, int n_seeds,
                                                      unsigned long long thr_beta, unsigned long long thr_gamma,
                                                      long sims, long sim_offset, int T, uint32_t k0, uint32_t k1,
                                                      uint32_t* __restrict__ hist, int32_t* __restrict__ glists,
                                                      unsigned long long* __restrict__ stats) {
    extern __shared__ uint32_t smem_w[];
    const int nwords = (n + 31) >> 5;
    const int nw4 = (nwords + 3) & ~3;
    uint32_t* bits = smem_w;                                   // ever infected (seeds included): susceptible <=> bit clear
    uint32_t* spent = smem_w + nw4;                            // node whose neighbours have ALL been infected: its row can never
                                                               // infect anybody again (infection is monotone) and is not walked any more
    uint32_t* recb = smem_w + 2 * nw4;                         // recovered
    // per-wave queue of (CSR position, target) pairs whose target was susceptible: their coins are drawn 64 at a time (below)
    int* const cq = reinterpret_cast<int*>(smem_w + 3 * nw4) + (threadIdx.x >> 6) * 256;       // 128 positions | 128 targets
    uint32_t* const after_q = smem_w + 3 * nw4 + (blockDim.x >> 6) * 256;
    IdT* lists = LISTS_IN_LDS ? reinterpret_cast<IdT*>(after_q)
                              : reinterpret_cast<IdT*>(glists + (size_t)blockIdx.x * 3 * n);
    IdT* cur = lists;
    IdT* nxt = lists + n;
    IdT* big = lists + 2 * (size_t)n;                          // [rows longer than GN_SIR_BIGROW in the graph] (LDS form), [n] (workspace form)
    __shared__ int cnt[3];                                     // [0] next-list length, [1] big-row list length, [2] ever infected
    uint32_t* hinf = hist;
    uint32_t* hrec = hist + (size_t)T * n;
    const int nthr = blockDim.x, tid = threadIdx.x;
    const int sub = tid & 15, gid = tid >> 4, ngroups = nthr >> 4, lane_in_wave = tid & 63;
    unsigned long long st_blocks = 0, st_ecoins = 0, st_rcoins = 0, st_entries = 0;
    for (long s = blockIdx.x; s < sims; s += gridDim.x) {
        const uint32_t sim = (uint32_t)(sim_offset + s);
        for (int w = tid; w < nwords; w += nthr) { bits[w] = 0u; spent[w] = 0u; recb[w] = 0u; }
        if (tid < 3) cnt[tid] = 0;
        __syncthreads();
        for (int j = tid; j < n_seeds; j += nthr) {            // distinct seeds: one infection event at t = 0 each
            const int v = seeds[j];
            const uint32_t m = 1u << (v & 31);
            if (!(atomicOr(&bits[v >> 5], m) & m)) { cur[atomicAdd(&cnt[0], 1)] = (IdT)v; atomicAdd(&hinf[v], 1u); }
        }
        __syncthreads();
        int n_inf = cnt[0];
        int n_ever = n_inf;                                    // once it reaches n nobody is left to infect: recovery coins only
        __syncthreads();
        if (tid == 0) cnt[2] = n_ever;
        for (int it = 1; it < T && n_inf > 0; ++it) {
            if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
            __syncthreads();
#ifndef GN_SIR_MODEB
#define GN_SIR_MODEB 1
#endif
            if (GN_SIR_MODEB && n_ever >= n) {
                // ---- everybody has been infected: only recovery coins remain, and the infected set is "not recovered".  No
                // lists: a thread takes four consecutive nodes = ONE Philox block
                int alive = 0;
                for (int q = tid; 4 * q < n; q += nthr) {
                    const int u0 = 4 * q;
                    uint32_t nib = (~recb[u0 >> 5] >> (u0 & 31)) & 0xFu;
                    if (u0 + 4 > n) nib &= (1u << (n - u0)) - 1u;
                    if (!nib) continue;
                    uint32_t w[4];
                    philox_block((uint32_t)q, (uint32_t)it, sim, 1u, k0, k1, w);
                    if (COUNT) { ++st_blocks; st_rcoins += __popc(nib); }
                    uint32_t gone = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((nib >> j) & 1u) {
                            if ((unsigned long long)w[j] < thr_gamma) { gone |= 1u << j; atomicAdd(&hrec[(size_t)it * n + u0 + j], 1u); }
                            else ++alive;
                        }
                    if (gone) atomicOr(&recb[u0 >> 5], gone << (u0 & 31));
                }
                if (alive) atomicAdd(&cnt[0], alive);
                __syncthreads();
                n_inf = cnt[0];
                __syncthreads();
                continue;
            }
            // infection attempts along the out-edges of the frontier: one 16-lane group per infected node, a lane takes
            // FOUR consecutive CSR positions (an aligned quad: one 16-byte read of the column list, one Philox block)
            auto infect = [&](int v) {
                const uint32_t m = 1u << (v & 31);
                if (!(atomicOr(&bits[v >> 5], m) & m)) {       // first edge to reach v this step (it was susceptible)
                    nxt[atomicAdd(&cnt[0], 1)] = (IdT)v;
                    atomicAdd(&cnt[2], 1);
                    atomicAdd(&hinf[(size_t)it * n + v], 1u);
                }
            };
            // Counted on the wiki-vote-size workload (beta 0.3): 2.6e9 CSR entries read, 0.24e9 of them with a susceptible target.
            // A coin drawn where it is found runs the ~60-instruction Philox sequence for the WHOLE wave whenever any of its 64
            // lanes needs one -- almost always, at 9 % of the lanes: the kernel issued 4.7e9 vector instructions, two per CSR
            // entry, ~40 % of the chip's issue rate, nearly all of it Philox on idle lanes.  So susceptible (position, target)
            // pairs go through a per-wave LDS queue and their coins are drawn 64 at a time.  A target may now be queued by
            // several rows before the first coin infects it; the extra coins change nothing (infection = OR over the coins of
            // the edges whose target was susceptible at the start of the step, exactly the reference's rule).
            int qn = 0;                                        // wave-uniform queue length
            auto drain = [&](int take) {                       // the last `take` (<= 64) queued pairs
                const int slot = qn - take + lane_in_wave;
                if (lane_in_wave < take) {
                    const int e = cq[slot], v = cq[128 + slot];
                    if (COUNT) { ++st_blocks; ++st_ecoins; }
                    if ((unsigned long long)philox_coin((uint32_t)e, (uint32_t)it, sim, 0u, k0, k1) < thr_beta) infect(v);
                }
                qn -= take;
            };
            auto push = [&](bool sus, int e, int v) {          // every lane of the wave calls it (sus = false: nothing to add)
                const unsigned long long mk = __ballot(sus);
                if (sus) { const int slot = qn + __popcll(mk & ((1ull << lane_in_wave) - 1ull)); cq[slot] = e; cq[128 + slot] = v; }
                qn += __popcll(mk);
                __builtin_amdgcn_wave_barrier();
                if (qn >= 64) drain(64);
            };
            auto try_edge = [&](int e, bool on) -> bool {      // one lane per CSR position; `on`: this lane has an entry
                const int v = on ? col[e] : 0;
                const bool sus = on && !(bits[v >> 5] & (1u << (v & 31)));
                if (COUNT && on) ++st_entries;
                push(sus, e, v);
                return sus;
            };
            // The walk of one frontier row is a chain of dependent loads (list entry -> row extent -> column ids -> bitmap), ~2 us
            // of L2 round trips for a few dozen integer instructions: counted, the wiki-vote-size workload reads 2.8e9 CSR
            // entries and draws 0.58e9 coins in 10 ms, an order of magnitude below both the integer and the L2 rate --
            // it was LATENCY bound, one row per lane group at a time.  So the rows are software-pipelined three deep per lane
            // group: while row i is tested, the column ids of row i+1 and the extent of row i+2 are in flight.
            struct Row { int u, lo, hi, ca, cb; };
            Row r1 = {-1, 0, 0, 0, 0}, r2 = {-1, 0, 0, 0, 0};
            for (int idx = gid; __any(idx < n_inf + 2 * ngroups); idx += ngroups) {
                // stage A: the next row's node and extent
                Row r0 = {-1, 0, 0, 0, 0};
                if (idx < n_inf) {
                    const int u = (int)cur[idx];
                    if (!(spent[u >> 5] & (1u << (u & 31)))) { r0.u = u; r0.lo = rowptr[u]; r0.hi = rowptr[u + 1]; }   // (spent: nothing left to do for u)
                }
                // stage B: column ids 0..31 of the row fetched one iteration ago (rows for the whole workgroup go on `big`)
                if (r1.u >= 0) {
                    if (r1.hi - r1.lo > GN_SIR_BIGROW) { if (sub == 0) big[atomicAdd(&cnt[1], 1)] = (IdT)r1.u; r1.u = -1; }
                    else {
                        if (r1.lo + sub < r1.hi) r1.ca = col[r1.lo + sub];
                        if (r1.lo + 16 + sub < r1.hi) r1.cb = col[r1.lo + 16 + sub];
                    }
                }
                // stage C: test the row whose column ids were requested one iteration ago (wave-uniform control flow: the
                // queue's ballots need every lane)
                {
                    bool any_sus = false;
                    auto test = [&](bool on, int e, int v) {
                        const bool sus = on && !(bits[v >> 5] & (1u << (v & 31)));
                        if (COUNT && on) ++st_entries;
                        push(sus, e, v);
                        any_sus |= sus;
                    };
                    const bool live = r2.u >= 0;
                    test(live && r2.lo + sub < r2.hi, r2.lo + sub, r2.ca);
                    if (__any(live && r2.hi - r2.lo > 16)) test(live && r2.lo + 16 + sub < r2.hi, r2.lo + 16 + sub, r2.cb);
                    for (int e0 = 32; __any(live && r2.lo + e0 < r2.hi); e0 += 16) {
                        const int e = r2.lo + e0 + sub;
                        const bool on = live && e < r2.hi;
                        test(on, e, on ? col[e] : 0);
                    }
                    // (a target infected during this very walk still counted as susceptible: the row is retired one step later)
                    const unsigned long long bal = __ballot(any_sus);
                    if (live && sub == 0 && !((bal >> (lane_in_wave & 48)) & 0xFFFFull)) atomicOr(&spent[r2.u >> 5], 1u << (r2.u & 31));
                }
                r2 = r1; r1 = r0;
            }
            if (qn > 0) drain(qn);
            // recovery coin of every node of the frontier (decided on the pre-step state: a node infected in this step is
            // not on `cur`); survivors go on the next list
            for (int idx = tid; idx < n_inf; idx += nthr) {
                const int u = (int)cur[idx];
                if (COUNT) { ++st_blocks; ++st_rcoins; }
                const bool gone = (unsigned long long)philox_coin((uint32_t)u, (uint32_t)it, sim, 1u, k0, k1) < thr_gamma;
                if (gone) {
                    atomicAdd(&hrec[(size_t)it * n + u], 1u);
                    atomicOr(&recb[u >> 5], 1u << (u & 31));
                }
                // survivors go on the next list: ONE LDS atomic per wave (64 lanes adding to the same word serialise)
                const unsigned long long keep = __ballot(!gone);
                int basep = 0;
                if (lane_in_wave == 0 && keep) basep = atomicAdd(&cnt[0], __popcll(keep));
                basep = __shfl(basep, 0, 64);
                if (!gone) nxt[basep + __popcll(keep & ((1ull << lane_in_wave) - 1ull))] = (IdT)u;
            }
            __syncthreads();
            const int n_big = cnt[1];
            for (int b = 0; b < n_big; ++b) {                   // hub rows: the whole workgroup strides one row
                const int u = (int)big[b];
                const int lo = rowptr[u], hi = rowptr[u + 1];
                for (int e0 = lo; e0 < hi; e0 += nthr) try_edge(e0 + tid, e0 + tid < hi);
                if (qn > 0) drain(qn);
            }
            __syncthreads();
            n_inf = cnt[0];
            n_ever = cnt[2];
            IdT* t = cur; cur = nxt; nxt = t;
            __syncthreads();
        }
    }
    if (COUNT) {
        atomicAdd(&stats[0], st_blocks); atomicAdd(&stats[1], st_ecoins); atomicAdd(&stats[2], st_rcoins); atomicAdd(&stats[3], st_entries);
    }
}

// counts[0..2][t][v] += (sims - cumInf, cumInf - cumRec, cumRec) for t >= 1; row 0 assigned.
__global__ __launch_bounds__(256) void k_sir_finalize(const uint32_t* __restrict__ hist, int n, int T, uint32_t sims,
                                                      uint32_t* __restrict__ counts) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const uint32_t* hinf = hist;
    const uint32_t* hrec = hist + (size_t)T * n;
    const size_t plane = (size_t)T * n;
    uint32_t ci = hinf[v], cr = 0;
    const uint32_t seeded = ci ? 1u : 0u;            // every trajectory starts from the same seed set
    counts[v] = 1u - seeded;                         // S row 0: assigned, ode_nn.py:56
    counts[plane + v] = seeded;                      // I row 0: assigned, ode_nn.py:55
    ci = seeded * sims;
    for (int t = 1; t < T; ++t) {
        ci += hinf[(size_t)t * n + v];
        cr += hrec[(size_t)t * n + v];
        counts[(size_t)t * n + v] += sims - ci;
        counts[plane + (size_t)t * n + v] += ci - cr;
        counts[2 * plane + (size_t)t * n + v] += cr;
    }
}

// src[e] for every CSR position (row expansion), once per call
__global__ void k_expand_rows(const int* __restrict__ rowptr, int n, int* __restrict__ src) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    for (int e = rowptr[u]; e < rowptr[u + 1]; ++e) src[e] = u;
}

// --------------------------------------------------------------------------- parity kernel (recorded coin stream)
__device__ __forceinline__ int block_rank(bool pred, int* wave_tot, int& total) {
    // stable exclusive rank of `pred` lanes over the 256-thread block
    const unsigned long long m = __ballot(pred);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) wave_tot[w] = __popcll(m);
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int c = wave_tot[q]; base += (q < w) ? c : 0; total += c; }
    return base + r;
}

__global__ __launch_bounds__(256) void k_sir_coins(const int* __restrict__ tsrc, const int* __restrict__ tdst, long nt,
                                                   int n, const int* __restrict__ seeds, int n_seeds, double beta,
                                                   double gamma, long sims, int T, const double* __restrict__ coins,
                                                   long n_coins, uint32_t* __restrict__ counts,
                                                   long long* __restrict__ used) {
    extern __shared__ uint8_t smem[];
    uint8_t* state = smem;
    uint8_t* flag = smem + n;
    __shared__ int wave_tot[4];
    const size_t plane = (size_t)T * n;
    long pos = 0;
    bool overrun = false;
    for (long s = 0; s < sims; ++s) {
        for (int v = threadIdx.x; v < n; v += 256) { state[v] = ST_S; flag[v] = 0; }
        __syncthreads();
        for (int j = threadIdx.x; j < n_seeds; j += 256) state[seeds[j]] = ST_I;
        __syncthreads();
        for (int v = threadIdx.x; v < n; v += 256) {          // row 0 assigned each sim (ode_nn.py:55-56)
            counts[v] = state[v] == ST_S;
            counts[plane + v] = state[v] == ST_I;
        }
        for (int it = 1; it < T; ++it) {
            // coins #1: one per active table row, in table order (ode_nn.py:61-67)
            for (long e0 = 0; e0 < nt; e0 += 256) {
                const long e = e0 + threadIdx.x;
                int v = 0;
                bool act = false;
                if (e < nt) { v = tdst[e]; act = state[tsrc[e]] == ST_I && state[v] == ST_S; }
                int total;
                const int rk = block_rank(act, wave_tot, total);
                if (act) {
                    const long ci = pos + rk;
                    if (ci < n_coins) { if (coins[ci] < beta) flag[v] = 1; } else overrun = true;
                }
                pos += total;
            }
            __syncthreads();
            // coins #2: one per infected node, ascending id (ode_nn.py:70-72)
            for (int u0 = 0; u0 < n; u0 += 256) {
                const int u = u0 + threadIdx.x;
                const bool inf = u < n && state[u] == ST_I;
                int total;
                const int rk = block_rank(inf, wave_tot, total);
                if (inf) {
                    const long ci = pos + rk;
                    if (ci < n_coins) { if (coins[ci] < gamma) flag[u] = 2; } else overrun = true;
                }
                pos += total;
            }
            __syncthreads();
            for (int v = threadIdx.x; v < n; v += 256) {
                const uint8_t f = flag[v];
                if (f == 1) state[v] = ST_I; else if (f == 2) state[v] = ST_R;
                flag[v] = 0;
                const uint8_t st = state[v];
                counts[(size_t)it * n + v] += st == ST_S;
                counts[plane + (size_t)it * n + v] += st == ST_I;
                counts[2 * plane + (size_t)it * n + v] += st == ST_R;
            }
            __syncthreads();
        }
    }
    if (__syncthreads_or(overrun)) pos = -1;
    if (threadIdx.x == 0) *used = pos;
}

// --------------------------------------------------------------------------- host side
static const size_t kLdsStateLimit = 150 * 1024;   // of the CU's 160 KiB

struct SeedArg { int32_t v[32]; };
__global__ void k_put_seeds(SeedArg sa, int n_seeds, int32_t* __restrict__ seeds) {
    if ((int)threadIdx.x < n_seeds) seeds[threadIdx.x] = sa.v[threadIdx.x];
}

// LDS of the frontier kernel: bitmap (+ three uint16 node lists -- current, next, long rows -- when they fit)
static size_t frontier_bitmap_bytes(int n) { return 3 * ((((size_t)n + 31) / 32 + 3) & ~(size_t)3) * 4; }   // ever-infected + spent + recovered
// list elements in LDS: current + next frontier [n] each, long rows [as many as the graph has, padded to 8]
static size_t frontier_list_bytes(int n, int n_big) { return 2 * (2 * (size_t)n + (((size_t)n_big + 7) & ~(size_t)7)); }
static bool frontier_lists_in_lds(int n, int n_big) { return n <= 65536 && frontier_bitmap_bytes(n) + frontier_list_bytes(n, n_big) <= 48 * 1024; }
// + 1 KB of coin queue per wave
static size_t frontier_lds_bytes(int n, int n_big, int threads) {
    return frontier_bitmap_bytes(n) + (size_t)(threads / 64) * 1024 + (frontier_lists_in_lds(n, n_big) ? frontier_list_bytes(n, n_big) : 0);
}
// Workgroup size and workgroups per CU (the LDS decides how many fit).  Measured, 10 000 x 20 (2 000 x 30 at epinions size),
// beta 0.3 / 0.05, ms:            16 waves per CU    24 waves      32 waves
//   fb-social size (256 threads)        --           2.90 / 4.12   3.26 / 4.14
//   wiki-vote size (512 threads)    9.8 / 28.0       8.2 / 22.5    9.3 / 25.1
//   epinions size  (512 threads)        --          29.5 / 36.8   35.7 / 41.9     (256 threads x 5: 37.1 / 46.8)
// -- past 24 waves the resident trajectories thrash each other's rows in the L2, below it the waits are exposed.  So: 256
// threads for small graphs, 512 otherwise (1 024 when the LDS leaves fewer than 16 waves), at most GN_SIR_WAVES waves per CU.
#ifndef GN_SIR_WAVES
#define GN_SIR_WAVES 24
#endif
#ifndef GN_SIR_THREADS
#define GN_SIR_THREADS 0
#endif
static int frontier_threads(int n, int n_big, int* per_cu_out) {
    for (int threads = GN_SIR_THREADS ? GN_SIR_THREADS : (n < 4096 ? 256 : 512); ; threads *= 2) {
        int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (frontier_lds_bytes(n, n_big, threads) + 64)));
        if (per_cu * (threads / 64) >= 16 || threads == 1024 || GN_SIR_THREADS) {
            per_cu = std::max(1, std::min(per_cu, GN_SIR_WAVES / (threads / 64)));
            *per_cu_out = per_cu;
            return threads;
        }
    }
}
static const int kFrontierGlobalGrid = 1024;       // workgroups that own a set of global lists (graphs past the LDS form)

int gn_sir_set_attributes() {       // once per device, from gnode_graph_create
    GN_HIP(hipFuncSetAttribute((const void*)k_sir_frontier<uint16_t, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsStateLimit));
    GN_HIP(hipFuncSetAttribute((const void*)k_sir_frontier<int32_t, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsStateLimit));
    GN_HIP(hipFuncSetAttribute((const void*)k_sir_frontier<uint16_t, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsStateLimit));
    GN_HIP(hipFuncSetAttribute((const void*)k_sir_frontier<int32_t, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsStateLimit));
    GN_HIP(hipFuncSetAttribute((const void*)k_sir_philox<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsStateLimit));
    GN_HIP(hipFuncSetAttribute((const void*)k_sir_coins, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsStateLimit));
    return 0;
}

extern "C" size_t gnode_sir_coins_workspace_bytes(void) { return gn_align(4096 * sizeof(int32_t)) + gn_align(64); }

extern "C" size_t gnode_sir_workspace_bytes(gnode_graph_t g, int32_t T) {
    if (!g) return 0;
    size_t b = gn_align((size_t)2 * T * g->n * sizeof(uint32_t)) + gn_align(4096 * sizeof(int32_t)) +
               gn_align((size_t)std::max<int64_t>(g->nnz, 1) * sizeof(int32_t));
    size_t tail = 0;                                       // one region, two users that never run together
    if ((size_t)2 * g->n > kLdsStateLimit) tail = (size_t)2048 * 2 * g->n;                                   // scan kernel, state in memory
    if (!frontier_lists_in_lds(g->n, g->n_bigrow)) tail = std::max(tail, (size_t)kFrontierGlobalGrid * 3 * g->n * sizeof(int32_t));   // frontier lists
    return b + gn_align(tail);
}

static int sir_mc_philox_impl(gnode_graph_t g, const int32_t* seeds_host, int32_t n_seeds, double beta,
                              double gamma, int64_t sims, int64_t sim_offset, int32_t T, uint64_t rng_seed,
                              uint32_t* counts, void* workspace, size_t workspace_bytes, void* stream, bool edge_scan,
                              unsigned long long* stats = nullptr /* device [4]: the counting instantiation, or null */) {
    GN_CHECK_ARG(g && counts && workspace, "gnode_sir_mc_philox: null pointer");
    GN_CHECK_ARG(n_seeds >= 0 && n_seeds <= 4096 && (seeds_host || n_seeds == 0), "gnode_sir_mc_philox: 0..4096 seeds");
    GN_CHECK_ARG(T >= 1 && sims >= 0 && sims <= 0xFFFFFFFFll && sim_offset >= 0 && sim_offset + sims <= 0xFFFFFFFFll,
                 "gnode_sir_mc_philox: bad T/sims/sim_offset");
    GN_CHECK_ARG(beta >= 0.0 && beta <= 1.0 && gamma >= 0.0 && gamma <= 1.0, "gnode_sir_mc_philox: beta, gamma in [0,1]");
    for (int i = 0; i < n_seeds; ++i)
        GN_CHECK_ARG(seeds_host[i] >= 0 && seeds_host[i] < g->n, "gnode_sir_mc_philox: seed %d out of range", seeds_host[i]);
    if (workspace_bytes < gnode_sir_workspace_bytes(g, T)) {
        gnode_set_error("gnode_sir_mc_philox: workspace %zu < %zu", workspace_bytes, gnode_sir_workspace_bytes(g, T));
        return GNODE_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const size_t hist_b = gn_align((size_t)2 * T * g->n * sizeof(uint32_t));
    uint32_t* hist = (uint32_t*)ws;
    int32_t* seeds = (int32_t*)(ws + hist_b);
    int32_t* src = (int32_t*)(ws + hist_b + gn_align(4096 * sizeof(int32_t)));
    uint8_t* gstate = (uint8_t*)(ws + hist_b + gn_align(4096 * sizeof(int32_t)) +
                                 gn_align((size_t)std::max<int64_t>(g->nnz, 1) * sizeof(int32_t)));
    if (int e = gn_zero_async(hist, hist_b, st)) return e;
    // seed ids: up to 32 travel as a kernel argument (no copy, no synchronisation -- the reference's experiments use 2);
    // longer lists are copied from the caller's host array, which may be a temporary, so the stream is synchronised
    // before returning control (documented in gnode.h)
    SeedArg sa;
    for (int i = 0; i < 32; ++i) sa.v[i] = i < n_seeds ? seeds_host[i] : 0;
    if (n_seeds <= 32) {
        hipLaunchKernelGGL(k_put_seeds, dim3(1), dim3(32), 0, st, sa, n_seeds, seeds);
        GN_LAUNCH_CHECK();
    } else {
        GN_HIP(hipMemcpyAsync(seeds, seeds_host, sizeof(int32_t) * n_seeds, hipMemcpyHostToDevice, st));
        GN_HIP(hipStreamSynchronize(st));
    }
    const unsigned long long tb = (unsigned long long)std::min(4294967296.0, std::max(0.0, std::floor(beta * 4294967296.0)));
    const unsigned long long tg = (unsigned long long)std::min(4294967296.0, std::max(0.0, std::floor(gamma * 4294967296.0)));
    const uint32_t k0 = (uint32_t)(rng_seed & 0xFFFFFFFFull), k1 = (uint32_t)(rng_seed >> 32);
    if (sims > 0) {
        const bool sampled = gn_prof_begin(3, st);
        int per_cu_f = 1;
        const int threads_f = frontier_threads(g->n, g->n_bigrow, &per_cu_f);
        const size_t fl = frontier_lds_bytes(g->n, g->n_bigrow, threads_f);
        const size_t lds = (size_t)2 * g->n;
        if (fl <= kLdsStateLimit && !edge_scan) {
            // frontier-driven walk.  Workgroups per CU by LDS, at least 16 waves per CU
            const int per_cu = per_cu_f, threads = threads_f;
            if (frontier_lists_in_lds(g->n, g->n_bigrow)) {
                const int grid = (int)std::min<int64_t>(sims, (int64_t)g->num_cu * per_cu);
                if (stats) hipLaunchKernelGGL((k_sir_frontier<uint16_t, true, true>), dim3(grid), dim3(threads), fl, st, g->rowptr, g->col, g->n, seeds,
                                              n_seeds, tb, tg, (long)sims, (long)sim_offset, T, k0, k1, hist, (int32_t*)nullptr, stats);
                else hipLaunchKernelGGL((k_sir_frontier<uint16_t, true, false>), dim3(grid), dim3(threads), fl, st, g->rowptr, g->col, g->n, seeds,
                                        n_seeds, tb, tg, (long)sims, (long)sim_offset, T, k0, k1, hist, (int32_t*)nullptr, stats);
            } else {
                const int grid = (int)std::min<int64_t>(std::min<int64_t>(sims, (int64_t)g->num_cu * per_cu), kFrontierGlobalGrid);
                if (stats) hipLaunchKernelGGL((k_sir_frontier<int32_t, false, true>), dim3(grid), dim3(threads), fl, st, g->rowptr, g->col, g->n, seeds,
                                              n_seeds, tb, tg, (long)sims, (long)sim_offset, T, k0, k1, hist, (int32_t*)gstate, stats);
                else hipLaunchKernelGGL((k_sir_frontier<int32_t, false, false>), dim3(grid), dim3(threads), fl, st, g->rowptr, g->col, g->n, seeds,
                                        n_seeds, tb, tg, (long)sims, (long)sim_offset, T, k0, k1, hist, (int32_t*)gstate, stats);
            }
        } else if (lds <= kLdsStateLimit) {
            // edge-parallel scan, node state in LDS: graphs whose frontier lists do not fit (n > ~25k with 32-bit ids)
            hipLaunchKernelGGL(k_expand_rows, dim3((g->n + 255) / 256), dim3(256), 0, st, g->rowptr, g->n, src);
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1)));
            const int threads = per_cu >= 4 ? 256 : (per_cu >= 2 ? 512 : 1024);
            const int grid = (int)std::min<int64_t>(sims, (int64_t)g->num_cu * per_cu);
            hipLaunchKernelGGL(k_sir_philox<true>, dim3(grid), dim3(threads), lds, st, src, g->col, (long)g->nnz, g->n, seeds,
                               n_seeds, tb, tg, (long)sims, (long)sim_offset, T, k0, k1, hist, (uint8_t*)nullptr);
        } else {
            hipLaunchKernelGGL(k_expand_rows, dim3((g->n + 255) / 256), dim3(256), 0, st, g->rowptr, g->n, src);
            const int grid = (int)std::min<int64_t>(sims, 2048);
            hipLaunchKernelGGL(k_sir_philox<false>, dim3(grid), dim3(256), 0, st, src, g->col, (long)g->nnz, g->n, seeds,
                               n_seeds, tb, tg, (long)sims, (long)sim_offset, T, k0, k1, hist, gstate);
        }
        if (sampled) gn_prof_end(3, st);
        GN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_sir_finalize, dim3((g->n + 255) / 256), dim3(256), 0, st, hist, g->n, T, (uint32_t)sims, counts);
    GN_LAUNCH_CHECK();
    return 0;
}

extern "C" int gnode_sir_mc_philox(gnode_graph_t g, const int32_t* seeds_host, int32_t n_seeds, double beta,
                                   double gamma, int64_t sims, int64_t sim_offset, int32_t T, uint64_t rng_seed,
                                   uint32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
    return sir_mc_philox_impl(g, seeds_host, n_seeds, beta, gamma, sims, sim_offset, T, rng_seed, counts, workspace,
                              workspace_bytes, stream, false);
}

// The production kernel's profiling instantiation: the same counts, plus what the launch did -- stats_host[0] Philox blocks
// computed, [1] infection coins drawn, [2] recovery coins drawn, [3] CSR entries read.  Synchronises `stream`.
extern "C" int gnode_sir_mc_philox_counted(gnode_graph_t g, const int32_t* seeds_host, int32_t n_seeds, double beta,
                                           double gamma, int64_t sims, int64_t sim_offset, int32_t T, uint64_t rng_seed,
                                           uint32_t* counts, void* workspace, size_t workspace_bytes, void* stream,
                                           uint64_t* stats_host) {
    GN_CHECK_ARG(stats_host && workspace, "gnode_sir_mc_philox_counted: null pointer");
    GN_CHECK_ARG(workspace_bytes >= gnode_sir_workspace_bytes(g, T), "gnode_sir_mc_philox_counted: workspace too small");
    // the tally lives in the (otherwise unused by the frontier walk) row-expansion region of the workspace
    char* ws = (char*)workspace;
    unsigned long long* stats = (unsigned long long*)(ws + gn_align((size_t)2 * T * g->n * sizeof(uint32_t)) + gn_align(4096 * sizeof(int32_t)));
    GN_CHECK_ARG(g->nnz >= 8, "gnode_sir_mc_philox_counted: graph too small");
    GN_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(unsigned long long), (hipStream_t)stream));
    if (int e = sir_mc_philox_impl(g, seeds_host, n_seeds, beta, gamma, sims, sim_offset, T, rng_seed, counts, workspace,
                                   workspace_bytes, stream, false, stats))
        return e;
    GN_HIP(hipMemcpyAsync(stats_host, stats, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GN_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

extern "C" int gnode_sir_mc_philox_scan(gnode_graph_t g, const int32_t* seeds_host, int32_t n_seeds, double beta,
                                        double gamma, int64_t sims, int64_t sim_offset, int32_t T, uint64_t rng_seed,
                                        uint32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
    return sir_mc_philox_impl(g, seeds_host, n_seeds, beta, gamma, sims, sim_offset, T, rng_seed, counts, workspace,
                              workspace_bytes, stream, true);
}

extern "C" int gnode_sir_mc_coins(const int32_t* table_src, const int32_t* table_dst, int64_t n_table, int32_t n,
                                  const int32_t* seeds_host, int32_t n_seeds, double beta, double gamma, int64_t sims,
                                  int32_t T, const double* coins, int64_t n_coins, uint32_t* counts,
                                  int64_t* coins_used_host, void* workspace, size_t workspace_bytes, void* stream) {
    GN_CHECK_ARG(table_src && table_dst && counts && workspace && coins_used_host && (coins || n_coins == 0),
                 "gnode_sir_mc_coins: null pointer");
    GN_CHECK_ARG(n > 0 && (size_t)2 * n <= kLdsStateLimit, "gnode_sir_mc_coins: parity mode supports n <= %zu (got %d)",
                 kLdsStateLimit / 2, n);
    GN_CHECK_ARG(n_seeds >= 0 && n_seeds <= 4096 && T >= 1 && sims >= 0 && n_table >= 0, "gnode_sir_mc_coins: bad sizes");
    for (int i = 0; i < n_seeds; ++i)
        GN_CHECK_ARG(seeds_host[i] >= 0 && seeds_host[i] < n, "gnode_sir_mc_coins: seed %d out of range", seeds_host[i]);
    if (workspace_bytes < gnode_sir_coins_workspace_bytes()) {
        gnode_set_error("gnode_sir_mc_coins: workspace too small");
        return GNODE_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    int32_t* seeds = (int32_t*)workspace;
    long long* used = (long long*)((char*)workspace + gn_align(4096 * sizeof(int32_t)));
    if (n_seeds) GN_HIP(hipMemcpyAsync(seeds, seeds_host, sizeof(int32_t) * n_seeds, hipMemcpyHostToDevice, st));
    {   // the parity kernel takes no graph handle: make sure this device's kernel attributes are set (once, under the lock)
        int dev = 0;
        GN_HIP(hipGetDevice(&dev));
        if (int e = gn_device_setup_once(dev)) return e;
    }
    hipLaunchKernelGGL(k_sir_coins, dim3(1), dim3(256), (size_t)2 * n, st, table_src, table_dst, (long)n_table, n, seeds,
                       n_seeds, beta, gamma, (long)sims, T, coins, (long)n_coins, counts, used);
    GN_LAUNCH_CHECK();
    long long h = 0;
    GN_HIP(hipMemcpyAsync(&h, used, sizeof(h), hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));
    *coins_used_host = h;
    if (h < 0) {
        gnode_set_error("gnode_sir_mc_coins: coin stream exhausted (n_coins=%lld)", (long long)n_coins);
        return GNODE_ERR_ARG;
    }
    return 0;
}
