// H = 64 fast path launchers (gnode_h64.hip), used by the C-ABI host code in gnode_ode.hip.
#pragma once
#include "gnode_common.h"

struct Step64Out {
    float* S; float* I; float* R;   // this step's output rows [rows], or null
    float* sol;                     // sol[g+1] base ([4*rows, 64]), or null
    float* ai;                      // [rows, 64] receives this step's neighbour sums A Z_I(y_g) -- times Z_S (1 - Z_S) when `zs` is
                                    // given too (gn_keep_ps) -- kept for the backward, or null
    float* zs;                      // [rows, 64] receives Z_S(y_g) (kept for the backward, see gn_keep_zs), or null
};

// True when gnode_forward_f32 (H = 64, trajectory kept) stores A Z_I(y_k) in the 4th slab of sol[k], 1 <= k <= n_steps - 1,
// instead of the beta-gamma copy; gnode_backward_f32 asks the same question about the `sol` it is handed.
bool gn_sol_carries_ai(const gnode_graph_s* g, long rows, int H, int n_steps, int n_out, int flags);
// 2 = persistent launch, 1 = one workgroup per sample, 0 = one launch per step (gnode_ode.hip); plan may be null
struct PersPlan;
int gn_forward_kind(const gnode_graph_s* g, long rows, int H, int method, int n_steps, int n_out, bool with_sol, int flags, PersPlan* plan);

int gn_h64_set_attributes();    // once per device, from gnode_graph_create
int gn_launch_mlp64(const gnode_graph_s* g, const float* X, const float* W, const float* b, float* Z, long nrows, hipStream_t st);
// ZI / ZI_next: gather tables [rows + 1][64] whose last row is the zero row (gn_launch_prologue64 writes it)
int gn_launch_step64(gnode_graph_s* g, long rows, float* Y, const float* ZI, float* ZI_next, const float* W,
                     const float* bias, const float* beta, const float* gamma, float dt, const gnode_params* p,
                     float* PR /* [rows][4] projected R state, or null = carry Y_R */, Step64Out out,
                     void* hub_scratch /* gn_hub_scratch_bytes(g, B, 64, 1) bytes of the caller's workspace */, hipStream_t st);

// single-launch integration for graphs whose per-sample state fits one workgroup's LDS (gnode_h64.hip: k_tiny64)
bool gn_tiny64_ok(int n, int n_steps, int n_out, bool prj);
int gn_launch_tiny64(const gnode_graph_s* g, long rows, const float* Y0, const float* ZI0, const float* PR0, const float* W,
                     const float* bias, const float* beta, const float* gamma, const float* dt_host, const int* slot_host,
                     int n_steps, const gnode_params* p, float* S, float* I, float* R, float* sol,
                     float* keep /* kept activations (gn_keep_zs / gn_keep_zi of grid points 0 .. n_steps-1), or null */, hipStream_t st);

// encoder + beta/gamma + trajectory point 0 + read-out at grid point 0 + projected R + Z_I(y_0) in one launch
// ZI / ZI_alt: the two gather tables, each [rows + 1][64]: row `rows` is the table's ZERO ROW (written here)
int gn_launch_prologue64(const float* x, const gnode_params* p, float* Y, float* beta, float* gamma, float* sol0, float* ZI,
                         float* ZI_alt, float* PR, float* S0, float* I0, float* R0, long rows, void* zero_ptr /* optional: a region this launch zero-fills too */,
                         size_t zero_bytes, hipStream_t st);

// H = 128 node MLP on the matrix cores (gnode_h128.hip)
int gn_h128_set_attributes();   // once per device, from gnode_graph_create
int gn_launch_mlp128(const gnode_graph_s* g, const float* X, const float* W, const float* b, float* Z, long nrows, hipStream_t st);
