// Probe: does the fabric deliver random 256-B rows as fast as random 1-KB / 2-KB chunks?
// (decides whether a batch-innermost state layout could speed up the CSR gather)
//   hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// each 16-lane group reads `deg` rows of 256 B; CHUNK consecutive groups read consecutive rows
template <int CHUNK>
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ T, const int* __restrict__ idx, long n_items,
                                               int deg, long n_chunks, float* __restrict__ out) {
    const int sub = threadIdx.x & 15;
    const long grp = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (grp >= n_items) return;
    const long item = grp / CHUNK;
    const int within = (int)(grp % CHUNK);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < deg; j += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long c = idx[item * deg + j + u];                     // chunk id in [0, n_chunks)
            v[u] = *reinterpret_cast<const float4*>(T + ((size_t)c * CHUNK + within) * 64 + 4 * sub);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    *reinterpret_cast<float4*>(out + (size_t)grp * 64 + 4 * sub) = acc;
}

template <int CHUNK>
static void run(const float* T, long table_rows, long n_groups, int deg, float* out) {
    const long n_chunks = table_rows / CHUNK, n_items = n_groups / CHUNK;
    std::vector<int> h((size_t)n_items * deg);
    unsigned s = 12345u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (int)((s >> 4) % (unsigned)n_chunks); }
    int* idx;
    hipMalloc(&idx, h.size() * sizeof(int));
    hipMemcpy(idx, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const unsigned grid = (unsigned)((n_groups * 16 + 255) / 256);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_probe<CHUNK>, dim3(grid), dim3(256), 0, 0, T, idx, n_groups, deg, n_chunks, out);
    hipEventRecord(a);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k_probe<CHUNK>, dim3(grid), dim3(256), 0, 0, T, idx, n_groups, deg, n_chunks, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)n_groups * deg * 256.0 * 5;
    printf("chunk %4d B: %.2f TB/s gathered (%.1f us per launch)\n", CHUNK * 256, bytes / (ms * 1e-3) / 1e12, ms * 1e3 / 5);
    hipFree(idx);
}

int main(int argc, char** argv) {
    const long n_groups = 600000;              // one gathered row-sum per group
    const int deg = 12;
    float *T, *out;
    hipMalloc(&T, (size_t)600000 * 256);
    hipMalloc(&out, (size_t)n_groups * 256);
    hipMemset(T, 0, (size_t)600000 * 256);
    const long sizes[3] = {600000, 150000, 75000};   // 154 MB (8 samples' Z_I), 38 MB, 19 MB (one sample)
    for (long table_rows : sizes) {
        printf("table %.0f MB\n", table_rows * 256 / 1e6);
        run<1>(T, table_rows, n_groups, deg, out);
        run<8>(T, table_rows, n_groups, deg, out);
    }
    return 0;
}
