// Shared pieces of the adjoint backward (gnode_bwd.hip, gnode_bwd_tiny.hip).
#pragma once
#include "gnode_common.h"

#define BWD_NWG 768     // 3 workgroups per CU (52 KB of LDS each) on 256 CUs

// partial-buffer layout per workgroup (floats)
struct PartLayout {
    int H;
    __host__ __device__ int oW() const { return 0; }
    __host__ __device__ int ob() const { return H * H; }
    __host__ __device__ int ow3() const { return H * H + H; }
    __host__ __device__ int ob3() const { return H * H + 5 * H; }
    __host__ __device__ int ow2() const { return H * H + 5 * H + 4; }
    __host__ __device__ int ob2() const { return H * H + 5 * H + 8; }
    __host__ __device__ int ow1() const { return H * H + 5 * H + 9; }
    __host__ __device__ int ob1() const { return H * H + 6 * H + 9; }
    __host__ __device__ int total() const { return H * H + 7 * H + 9; }
};

// Whole adjoint sweep of a batch of tiny graphs (n <= 64, H = 64) in ONE launch; writes partial slot b for sample b.
bool gn_tiny_bwd64_ok(const gnode_graph_s* g, long rows, int H, int n_steps);
int gn_launch_tiny_bwd64(const gnode_graph_s* g, long rows, const float* x, const gnode_params* p, const float* dt_host,
                         int n_steps, const int32_t* out_rows_host, int n_out, const float* sol, const float* gS,
                         const float* gI, const float* gR, float* part,
                         const float* keep /* the forward's kept activations, or null = recompute */, hipStream_t st);

// H = 128: a += dt dpre W, gW, gb on the matrix cores (gnode_h128.hip); raises *slots_used to its grid size
int gn_launch_bwd_mlp128(const gnode_graph_s* g, const float* dpre, const float* Ysol, const float* W, float dt, float* a, long rows,
                         float* part, int* slots_used, hipStream_t st);
