// Stand-alone probe: would DATA-TAGGED rows (cdna_hip_programming.md Guideline 16, R2: "the data IS the flag") let the persistent
// multi-step kernels drop their per-step group barrier?
//
// Shape of k_pers64 at fb-social / wiki-vote size: one workgroup of 256 threads per CU owns 16 rows of a [rows][64] fp32
// table (ping-pong); every step each 16-lane group rewrites ITS row, then gathers DEG neighbour rows of the table the step
// before completed.  Z_I = sigmoid(.) is positive, so the sign bit of every float is free: the producer stamps the step's
// tag there (one 16-byte sc1 store per lane = one granule, tag on all four floats), the consumer re-loads a granule until its
// four sign bits carry the expected tag and strips them -- no drain, no workgroup barrier, no flag, no poll of other words.
// Tag of step s = ((s >> 1) & 1) ^ 1: a table is rewritten every second step, so consecutive contents of a slot differ in tag,
// and zeroed memory (tag 0) is never mistaken for step 0's rows.  Double buffering is enough only because the neighbour
// relation is SYMMETRIC: a workgroup cannot finish step s+1's gather (and overwrite the slot a slow neighbour still reads for
// step s) before that neighbour has published its step-s row, which it does after its own step-s gather.
//   mode 0: sc1 row stores, every wave drains, workgroup barrier, one flag per workgroup, wave 0 polls ALL flags, barrier,
//           sc1 loads            (what k_pers64 does for groups that span XCDs)
//   mode 1: tagged rows as above, sc1 stores / sc1 loads, bounded re-load loop
// Uneven load: every 3rd workgroup spins ~1 us extra per step.  Every gathered word is checked.
//   hipcc --offload-arch=gfx950 -O3 -o tagged_rows_probe tools/tagged_rows_probe.hip && ./tagged_rows_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ v4u bload(rsrc_t rs, unsigned off) { return __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16)); }
__device__ __forceinline__ void bstore(rsrc_t rs, unsigned off, v4u v) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16); }

struct Ctl { unsigned flags[512]; unsigned bad[32]; unsigned timeout[32]; unsigned long long reloads; };

__device__ __forceinline__ float expect(int step, int row, int j) { return (float)((step * 131 + row) % 4099) + 0.001f * (float)j + 0.5f; }   // > 0

template <int MODE, int DEG>
__global__ __launch_bounds__(256) void k_probe(Ctl* ctl, float* T0, float* T1, int rows, int wgs, int n_steps, unsigned long long* t_out) {
    extern __shared__ float lds[];
    __shared__ int giveup;
    const int lane = threadIdx.x & 63, sub = lane & 15, lg = threadIdx.x >> 4;
    const int idx = blockIdx.x;
    if (idx >= wgs) return;
    if (threadIdx.x == 0) giveup = 0;
    __syncthreads();
    const int row = idx * 16 + lg;
    const bool own = row < rows;
    const unsigned tbytes = (unsigned)rows * 256u;
    const rsrc_t r[2] = {make_rsrc(T0, tbytes), make_rsrc(T1, tbytes)};
    // symmetric neighbour relation: row +- o_k (mod rows), offsets spread over the whole table (other workgroups, other XCDs)
    int nb[DEG];
#pragma unroll
    for (int k = 0; k < DEG; ++k) {
        const int o = 17 + (k >> 1) * (rows / (DEG / 2 + 1)) + 7 * (k >> 1);
        nb[k] = ((k & 1) ? row + rows - (o % rows) : row + o) % rows;
    }
    unsigned bad = 0;
    unsigned long long reloads = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < n_steps; ++s) {
        const rsrc_t wr = r[s & 1];
        const unsigned tag = ((((unsigned)s >> 1) & 1u) ^ 1u) << 31;
        if ((idx % 3) == 1) { const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while (__builtin_amdgcn_s_memrealtime() - t0 < 100) {} }   // ~1 us
        if (own) {
            const v4f v = {expect(s, row, 4 * sub), expect(s, row, 4 * sub + 1), expect(s, row, 4 * sub + 2), expect(s, row, 4 * sub + 3)};
            v4u u = __builtin_bit_cast(v4u, v);
            if (MODE == 1) { u.x |= tag; u.y |= tag; u.z |= tag; u.w |= tag; }
            bstore(wr, (unsigned)row * 256u + 16u * sub, u);
        }
        if (MODE == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(ctl->flags + idx, (unsigned)s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < 64) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    unsigned f[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) f[q] = (lane + 64 * q < wgs) ? __hip_atomic_load(ctl->flags + lane + 64 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
                    if (__all(min(min(f[0], f[1]), min(f[2], f[3])) >= (unsigned)s + 1u)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) { if (lane == 0) { giveup = 1; atomicAdd(&ctl->timeout[0], 1u); } break; }
                }
            }
            __syncthreads();
            if (giveup) return;
        }
        if (own) {
            v4u v[DEG];
#pragma unroll
            for (int k = 0; k < DEG; ++k) v[k] = bload(wr, (unsigned)nb[k] * 256u + 16u * sub);
            if (MODE == 1) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < DEG; ++k) {
                        const bool good = ((v[k].x & v[k].y & v[k].z & v[k].w) & 0x80000000u) == tag && ((v[k].x | v[k].y | v[k].z | v[k].w) & 0x80000000u) == tag;
                        if (!good) { v[k] = bload(wr, (unsigned)nb[k] * 256u + 16u * sub); ok = false; ++reloads; }
                    }
                    if (__all(ok)) break;                   // wave-uniform exit: the whole wave re-checks together
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) { if (lane == 0) atomicAdd(&ctl->timeout[0], 1u); break; }
                }
            }
#pragma unroll
            for (int k = 0; k < DEG; ++k) {
                const v4u u = {v[k].x & 0x7FFFFFFFu, v[k].y & 0x7FFFFFFFu, v[k].z & 0x7FFFFFFFu, v[k].w & 0x7FFFFFFFu};
                const v4f f = __builtin_bit_cast(v4f, u);
                bad += f.x != expect(s, nb[k], 4 * sub); bad += f.y != expect(s, nb[k], 4 * sub + 1);
                bad += f.z != expect(s, nb[k], 4 * sub + 2); bad += f.w != expect(s, nb[k], 4 * sub + 3);
            }
        }
    }
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
    if (bad) atomicAdd(&ctl->bad[0], bad);
    if (reloads) atomicAdd(&ctl->reloads, reloads);
    if (threadIdx.x == 0 && idx == 0) t_out[0] = t_end - t_begin;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    Ctl* ctl; CK(hipMalloc(&ctl, sizeof(Ctl)));
    const size_t tab = (size_t)8192 * 64;
    float *T0, *T1; CK(hipMalloc(&T0, tab * 4)); CK(hipMalloc(&T1, tab * 4));
    unsigned long long* t_out; CK(hipMalloc(&t_out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n_steps = 400;
    const size_t lds = 96 * 1024;                      // one workgroup per CU
    struct Cfg { int mode, rows, deg; };
    std::vector<Cfg> cfgs;
    for (int rows : {1893, 3500}) for (int deg : {16, 32}) for (int mode : {0, 1}) cfgs.push_back({mode, rows, deg});
#define ATTR(M, D) CK(hipFuncSetAttribute((const void*)k_probe<M, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
    ATTR(0, 16); ATTR(1, 16); ATTR(0, 32); ATTR(1, 32);
    for (const Cfg& c : cfgs) {
        const int wgs = (c.rows + 15) / 16;
        if (wgs > prop.multiProcessorCount) continue;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), 0));
            CK(hipMemsetAsync(T0, 0, tab * 4, 0)); CK(hipMemsetAsync(T1, 0, tab * 4, 0));
            CK(hipEventRecord(e0, 0));
#define GO(M, D) hipLaunchKernelGGL((k_probe<M, D>), dim3(wgs), dim3(256), lds, 0, ctl, T0, T1, c.rows, wgs, n_steps, t_out)
            if (c.deg == 16) { if (c.mode == 0) GO(0, 16); else GO(1, 16); } else { if (c.mode == 0) GO(0, 32); else GO(1, 32); }
            CK(hipGetLastError());
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            Ctl h; CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
            unsigned long long th; CK(hipMemcpy(&th, t_out, 8, hipMemcpyDeviceToHost));
            if (rep == 1)
                printf("mode %d (%s)  rows %4d (%3d workgroups) deg %2d : wrong words %u, timeouts %u, re-loaded granules per row-step %.3f, %.2f us/step (host), "
                       "workgroup 0 in-kernel %.2f us/step\n", c.mode, c.mode ? "tagged rows, no barrier" : "flag barrier", c.rows, wgs, c.deg, h.bad[0], h.timeout[0],
                       (double)h.reloads / ((double)c.rows * 16 * n_steps), 1000.0 * ms / n_steps, (double)th / 100.0 / n_steps);
        }
    }
    return 0;
}
