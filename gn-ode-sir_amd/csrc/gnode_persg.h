// Persistent one-launch integration / adjoint sweep for H = 8, 16, 32 (gnode_persg.hip): plan and launchers.
#pragma once
#include "gnode_common.h"

struct PersgPlan { int wgs, wps, nw, map_off, idcap, segcap; size_t lds; };   // nw: waves per workgroup (64 / LPR rows each)

// false: this (graph, rows, H, horizon) keeps the one-launch-per-step forms (more rows than one resident grid of one workgroup
// per CU holds, a window of rows whose neighbour ids / hub segments do not fit a workgroup's LDS, other hidden sizes)
bool gn_persg_plan(const gnode_graph_s* g, long rows, int H, int n_steps, PersgPlan* p);
int gn_persg_set_attributes();
// Y0 / beta / gamma / table Z0 = Z_I(y_0) as k_encode and the node-MLP launch left them; slot_host[k]: output row of grid point k+1 or -1
int gn_launch_persg(const gnode_graph_s* g, const PersgPlan& pl, long rows, int H, const float* Y0, float* Z0, float* Z1,
                    const float* beta, const float* gamma, const float* dt_host, const int* slot_host, int n_steps,
                    const gnode_params* p, float* S, float* I, float* R, float* sol, void* ctl, bool ctl_is_zero, hipStream_t st);
// intervals G-1 .. 1; ZS0 | ZI0 = Z(y_{G-1}), Q0 = its q table, a = the adjoint after the head's VJP at grid point G-1;
// slot_of_prev[i]: output row of grid point i-1 or -1
int gn_launch_persg_bwd(const gnode_graph_s* g, const PersgPlan& pl, long rows, int H, int G, float* ZI0, float* ZI1, float* Q0, float* Q1,
                        const float* ZS0, const float* sol, const float* beta, const float* gamma, float* a_state, float* part,
                        const float* gS, const float* gI, const float* gR, const gnode_params* p, const float* dt_host,
                        const int* slot_of_prev, void* ctl, bool ctl_is_zero, hipStream_t st);
