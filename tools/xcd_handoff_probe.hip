// Stand-alone probe for the per-XCD persistent multi-step kernels: what does ONE step's
// producer -> group barrier -> consumer hand-off cost, and which store / load flavours are correct?
//
// Shape of the real kernel: 1 workgroup of 1024 threads per CU (LDS-forced), groups of workgroups own one sample's
// gather table ([rows][64] fp32, ping-pong).  Every step each 16-lane group rewrites ITS row of the table, the group
// of workgroups meets at a counter barrier, then every lane group gathers DEG random rows of the table and checks every
// word against what the step must have written.  Variants (argv[1]):
//   0  sc1 stores + sc1 loads, group = the workgroups of ONE XCD (XCC_ID-identified)      [guide's measured form, row 1]
//   1  plain stores + sc1 loads, same-XCD group          [hypothesis: stores and loads meet in the XCD's L2]
//   2  sc1 stores + sc1 loads, ONE group = all workgroups (cross-XCD)
//   3  plain stores + release fence / acquire fence + plain loads, same-XCD group         [Guideline 16 recipe form]
// Uneven load: every 3rd workgroup spins ~2 us extra before storing.  Reports wrong words, us per step (host events
// around the launch / steps) for several group sizes.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_handoff_probe tools/xcd_handoff_probe.hip && ./xcd_handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
template <int AUX> __device__ __forceinline__ v4f bload(rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX));
}
template <int AUX> __device__ __forceinline__ void bstore(rsrc_t rs, unsigned off, v4f v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rs, off, 0, AUX);
}

struct Ctl {                 // zeroed before every launch
    unsigned ticket[8][32];  // per-XCC ticket counters (one 128-B line each)
    unsigned bar[64][32];    // per-group barrier counters (one line each)
    unsigned bad[32];        // wrong words
    unsigned timeout[32];
};

__device__ __forceinline__ float expect(int step, int row, int j) { return (float)(step * 131 + row) + 0.001f * (float)j; }

template <int MODE>
__global__ __launch_bounds__(1024) void k_probe(Ctl* ctl, float* T0, float* T1, int rows_per_group, int wgs_per_group, int groups_per_xcd,
                                                int n_steps, int deg, unsigned long long* t_out) {
    extern __shared__ float lds[];
    __shared__ unsigned sh[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 15, lg = (threadIdx.x >> 4);   // 64 lane groups
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
        unsigned tk;
        if (MODE == 2) { tk = atomicAdd(&ctl->ticket[0][0], 1u); sh[0] = 0; sh[1] = tk; }
        else { tk = atomicAdd(&ctl->ticket[xcc][0], 1u); sh[0] = xcc; sh[1] = tk; }
    }
    __syncthreads();
    const unsigned xcc = sh[0], tk = sh[1];
    int group, idx;
    if (MODE == 2) { group = 0; idx = (int)tk; if (idx >= wgs_per_group) return; }
    else {
        const int gix = (int)tk / wgs_per_group;
        if (gix >= groups_per_xcd) return;
        group = (int)xcc * groups_per_xcd + gix; idx = (int)tk % wgs_per_group;
    }
    unsigned* bar = &ctl->bar[group][0];
    const int row = idx * 64 + lg;                       // this lane group's row inside the group's table
    const bool own = row < rows_per_group;
    const size_t gbase = (size_t)group * rows_per_group * 64;
    const unsigned tbytes = (unsigned)rows_per_group * 256u;
    const rsrc_t r0 = make_rsrc(T0 + gbase, tbytes), r1 = make_rsrc(T1 + gbase, tbytes);
    unsigned bad = 0;
    unsigned rng = 12345u + 977u * (unsigned)(group * 4096 + row);
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < n_steps; ++s) {
        const rsrc_t wr = (s & 1) ? r1 : r0;
        if ((idx % 3) == 1) { const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while (__builtin_amdgcn_s_memrealtime() - t0 < 200) {} }   // ~2 us
        if (own) {
            v4f v = {expect(s, row, 4 * sub), expect(s, row, 4 * sub + 1), expect(s, row, 4 * sub + 2), expect(s, row, 4 * sub + 3)};
            if (MODE == 1 || MODE == 3) bstore<0>(wr, (unsigned)row * 256u + 16u * sub, v);
            else bstore<16>(wr, (unsigned)row * 256u + 16u * sub, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (MODE == 3) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)wgs_per_group * (unsigned)(s + 1);
            unsigned spins = 0;
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000u) { atomicAdd(&ctl->timeout[0], 1u); break; }
            }
            if (MODE == 3) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
        __syncthreads();
        // gather: deg random rows of the table just completed, 16 in flight
        if (own) {
            for (int e0 = 0; e0 < deg; e0 += 16) {
                v4f v[16]; int rr[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    rng = rng * 1664525u + 1013904223u;
                    rr[k] = (int)((rng >> 8) % (unsigned)rows_per_group);
                    rr[k] = __builtin_amdgcn_readfirstlane(0) + __shfl(rr[k], lane & 48);        // the lane group's leader decides
                    if (MODE == 3) v[k] = bload<0>(wr, (unsigned)rr[k] * 256u + 16u * sub);
                    else v[k] = bload<16>(wr, (unsigned)rr[k] * 256u + 16u * sub);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    bad += v[k].x != expect(s, rr[k], 4 * sub); bad += v[k].y != expect(s, rr[k], 4 * sub + 1);
                    bad += v[k].z != expect(s, rr[k], 4 * sub + 2); bad += v[k].w != expect(s, rr[k], 4 * sub + 3);
                }
            }
        }
    }
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
    if (bad) atomicAdd(&ctl->bad[0], bad);
    if (threadIdx.x == 0 && idx == 0) t_out[group] = t_end - t_begin;
}

int main(int argc, char** argv) {
    int num_cu = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); num_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, num_cu);
    Ctl* ctl; CK(hipMalloc(&ctl, sizeof(Ctl)));
    const size_t tab = (size_t)64 * 8192 * 64;   // floats: up to 64 groups x 8192 rows
    float *T0, *T1; CK(hipMalloc(&T0, tab * 4)); CK(hipMalloc(&T1, tab * 4));
    unsigned long long* t_out; CK(hipMalloc(&t_out, 64 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n_steps = 200;
    struct Cfg { int mode, wgs_per_group, groups_per_xcd, rows, deg; };
    std::vector<Cfg> cfgs;
    for (int mode = 0; mode < 4; ++mode) {
        if (mode == 2) { cfgs.push_back({2, 30, 1, 1893, 16}); cfgs.push_back({2, 111, 1, 7066, 32}); cfgs.push_back({2, 256, 1, 16384, 16}); continue; }
        cfgs.push_back({mode, 30, 1, 1893, 16});     // fb-social: one sample per XCD, 1 row per lane group
        cfgs.push_back({mode, 15, 2, 960, 16});      // two groups per XCD
        cfgs.push_back({mode, 32, 1, 2048, 32});
        cfgs.push_back({mode, 8, 4, 512, 16});
    }
    for (const Cfg& c : cfgs) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), 0));
            CK(hipMemsetAsync(T0, 0xff, tab * 4, 0)); CK(hipMemsetAsync(T1, 0xff, tab * 4, 0));
            CK(hipEventRecord(e0, 0));
            const size_t lds = 96 * 1024;
#define LAUNCH(M) hipLaunchKernelGGL(k_probe<M>, dim3(num_cu), dim3(1024), lds, 0, ctl, T0, T1, c.rows, c.wgs_per_group, c.groups_per_xcd, n_steps, c.deg, t_out)
            CK(hipFuncSetAttribute((const void*)k_probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            CK(hipFuncSetAttribute((const void*)k_probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            CK(hipFuncSetAttribute((const void*)k_probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            CK(hipFuncSetAttribute((const void*)k_probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            if (c.mode == 0) LAUNCH(0); else if (c.mode == 1) LAUNCH(1); else if (c.mode == 2) LAUNCH(2); else LAUNCH(3);
            CK(hipGetLastError());
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            Ctl h; CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
            unsigned long long th[64]; CK(hipMemcpy(th, t_out, sizeof(th), hipMemcpyDeviceToHost));
            unsigned tickets = 0; for (int x = 0; x < 8; ++x) tickets += h.ticket[x][0];
            if (rep == 1) {
                printf("mode %d  wgs/group %3d  groups/xcd %d  rows %5d deg %2d : wrong words %u, timeouts %u, %.2f us/step (host, whole launch), "
                       "group 0 in-kernel %.2f us/step; tickets per XCC:", c.mode, c.wgs_per_group, c.groups_per_xcd, c.rows, c.deg,
                       h.bad[0], h.timeout[0], 1000.0 * ms / n_steps, (double)th[0] / 100.0 / n_steps);
                for (int x = 0; x < 8; ++x) printf(" %u", h.ticket[x][0]);
                printf("\n");
            }
        }
    }
    return 0;
}
