// Stand-alone C++ host driving libgnode_hip.so through the C ABI only (no Python, no torch):
// ring graph, B samples, one ODEBlock.forward; prints the outputs' checksum so that a test can
// compare it with the Python host path.
//   hipcc --offload-arch=gfx950 -I include examples/cabi_forward.cpp -L gn-ode-sir_amd/gnode -lgnode_hip -o cabi_forward
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gnode.h"

#define CK(x) do { if ((x) != 0) { fprintf(stderr, "%s failed: %s\n", #x, gnode_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static float lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000, B = argc > 2 ? atoi(argv[2]) : 2, H = 64, n_steps = 9;
    // ring + chords: every node i is linked to i+-1 and i+-7 (symmetric CSR, sorted columns)
    std::vector<int32_t> rowptr(n + 1, 0), col;
    for (int i = 0; i < n; ++i) {
        int nb[4] = {(i + n - 7) % n, (i + n - 1) % n, (i + 1) % n, (i + 7) % n};
        std::vector<int> v(nb, nb + 4);
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        for (int c : v) if (c != i) col.push_back(c);
        rowptr[i + 1] = (int32_t)col.size();
    }
    gnode_graph_t g = nullptr;
    CK(gnode_graph_create(rowptr.data(), col.data(), n, (int64_t)col.size(), &g));

    unsigned seed = 12345u;
    auto fill = [&](size_t cnt, float scale) { std::vector<float> v(cnt); for (auto& x : v) x = (lcg(seed) * 2.f - 1.f) * scale; return v; };
    std::vector<float> W = fill((size_t)H * H, 0.125f), b = fill(H, 0.125f), w1 = fill(H, 1.f), b1 = fill(H, 1.f);
    std::vector<float> w3 = fill(4 * H, 0.125f), b3 = fill(4, 0.125f), w2 = fill(4, 0.5f), b2 = fill(1, 0.5f);
    const long rows = (long)B * n;
    std::vector<float> x((size_t)rows * (3 + H), 0.f);
    for (long r = 0; r < rows; ++r) {
        float* xr = &x[(size_t)r * (3 + H)];
        const bool seeded = (r % n) == (r / n) * 3;                  // one seed node per sample
        xr[0] = seeded ? 0.f : 1.f; xr[1] = seeded ? 1.f : 0.f; xr[2] = 0.f;
        xr[3] = 0.2f + 0.05f * (float)(r / n); xr[4] = 0.1f;          // beta, gamma
    }
    auto dev = [&](const std::vector<float>& h, float** d) -> int {
        HK(hipMalloc(d, h.size() * sizeof(float)));
        HK(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        return 0;
    };
    float *dW, *db, *dw1, *db1, *dw3, *db3, *dw2, *db2, *dx, *dout;
    void* ws;
    if (dev(W, &dW) || dev(b, &db) || dev(w1, &dw1) || dev(b1, &db1) || dev(w3, &dw3) || dev(b3, &db3) || dev(w2, &dw2) ||
        dev(b2, &db2) || dev(x, &dx)) return 1;
    const int G = n_steps + 1;
    HK(hipMalloc(&dout, sizeof(float) * 3 * G * rows));
    const size_t ws_bytes = gnode_forward_workspace_bytes(g, rows, H, 0);
    HK(hipMalloc(&ws, ws_bytes));
    gnode_params p = {dW, db, dw1, db1, dw3, db3, dw2, db2};
    std::vector<float> dt(n_steps, 0.5f);
    hipStream_t st;
    HK(hipStreamCreate(&st));
    CK(gnode_forward_f32(g, dx, &p, dt.data(), n_steps, 0, nullptr, 0, dout, dout + (size_t)G * rows, dout + (size_t)2 * G * rows,
                         nullptr /* sol */, nullptr /* keep */, 0, rows, H, ws, ws_bytes, st, 0 /* flags */, nullptr /* sol_info */));
    HK(hipStreamSynchronize(st));
    std::vector<float> out((size_t)3 * G * rows);
    HK(hipMemcpy(out.data(), dout, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    double sumS = 0, sumI = 0, dev1 = 0;
    for (size_t i = 0; i < (size_t)G * rows; ++i) {
        sumS += out[i]; sumI += out[(size_t)G * rows + i];
        dev1 = fmax(dev1, fabs((double)out[i] + out[(size_t)G * rows + i] + out[(size_t)2 * G * rows + i] - 1.0));
    }
    printf("n=%d B=%d G=%d sumS=%.6f sumI=%.6f max|S+I+R-1|=%.2e\n", n, B, G, sumS, sumI, dev1);
    // bad argument -> error code + message, no crash
    const int rc = gnode_forward_f32(g, dx, &p, dt.data(), n_steps, 0, nullptr, 0, dout, dout, dout, nullptr, nullptr, 0, rows + 1, H, ws, ws_bytes, st, 0, nullptr);
    printf("bad rows -> rc=%d (%s)\n", rc, gnode_last_error());
    CK(gnode_graph_destroy(g));
    return dev1 < 1e-5 && rc == GNODE_ERR_ARG ? 0 : 2;
}
