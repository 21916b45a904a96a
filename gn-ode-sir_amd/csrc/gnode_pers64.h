// Persistent multi-step H = 64 integration (gnode_pers64.hip): plan, control block, launcher.
#pragma once
#include "gnode_common.h"

#define PERS_FLAG_WORDS 2048

// How the workgroups of one launch are dealt to samples.  A "group" = the workgroups that own one sample's rows:
//   nt          16-row tiles per workgroup (blockDim = 256 * nt); wgs = ceil(n / (16 nt)) workgroups per group
//   span == 1   a group sits on ONE XCD, gpx groups side by side on each XCD (tickets [gi * wgs, (gi + 1) * wgs))
//   span  > 1   a group takes `span` whole XCDs, `per` tickets on each
//   concurrent  groups alive at once (>= B: one round)
struct PersPlan { int nt, wgs, span, gpx, per, slots, n_xcc, rounds, concurrent, fstride; };

struct PersCtl {                       // device memory, zeroed (k_pers_zero) in front of every launch
    unsigned ticket[8][32];            // per XCC (a 128-B line each): the next free slot on that XCD
    unsigned error[32];                // [0] != 0: a workgroup gave up waiting (code), [1]: the epoch it waited for
    unsigned flags[PERS_FLAG_WORDS];   // per group `fstride` words: workgroup idx's last published epoch
    unsigned long long prof[8];        // diagnostic build only (GN_PERS_PROF): 100 MHz ticks per phase, group 0 / workgroup 0
};

// the part of the plan a workgroup needs to find its place
struct PersPlace { int wgs, span, gpx, per, slots, n_xcc, rounds, concurrent, fstride; };
static inline PersPlace pers_place_of(const PersPlan& pl) {
    PersPlace q; q.wgs = pl.wgs; q.span = pl.span; q.gpx = pl.gpx; q.per = pl.per; q.slots = pl.slots; q.n_xcc = pl.n_xcc;
    q.rounds = pl.rounds; q.concurrent = pl.concurrent; q.fstride = pl.fstride; return q;
}

struct PersSched { float dt[128]; short slot[128]; int n_steps; };

struct PersArgs {
    const int* rowhdr; const int* col; const int* rowmap;   // rowmap: lane-group slot -> node (or -1), per tile count (gnode_graph_s::persmap)
    const int* hubslot; const int* segptr; const int* segitem;   // hub work lists (gnode_graph_s::pershub ...), null without hub rows
    int n, B, lds_slots; unsigned rows;
    PersPlace pp;
    const float* Y0; const float* PR0; const float* beta; const float* gamma;
    float* Z0; float* Z1; float* keep;
    const float* W; const float* bias; const float* w3; const float* b3; const float* w2; const float* b2;
    float* S; float* I; float* R; float* sol;
    PersCtl* ctl;
    PersSched sched;
};

// false: this (graph, batch, horizon) does not take the persistent path (hub rows, too many rows for one resident grid, ...)
bool gn_pers64_plan(const gnode_graph_s* g, long B, int n_steps, PersPlan* p);
size_t gn_pers64_ctl_bytes();
int gn_pers64_zero_ctl(void* ctl, hipStream_t st);      // in front of every persistent launch (a kernel: see gnode_pers64.hip)
int gn_pers64_set_attributes();
// Y0 / PR0 / beta / gamma / table 0 (Z0, or keep's first table) as gn_launch_prologue64 left them
int gn_launch_pers64(const gnode_graph_s* g, const PersPlan& pl, long rows, const float* Y0, const float* PR0, float* Z0, float* Z1,
                     const float* W, const float* bias, const float* beta, const float* gamma, const float* dt_host,
                     const int* slot_host, int n_steps, const gnode_params* p, float* S, float* I, float* R, float* sol, float* keep,
                     void* ctl, bool ctl_is_zero /* the caller's previous launch on this stream zero-filled it */, hipStream_t st);

// adjoint sweep, intervals G-2 .. 1, in one persistent launch (gnode_pers64_bwd.hip)
int gn_pers_bwd64_set_attributes();
bool gn_pers_bwd64_plan(const gnode_graph_s* g, long B, int n_steps, PersPlan* p);
int gn_launch_pers_bwd64(const gnode_graph_s* g, const PersPlan& pl, long rows, int G, float* Q0, float* Q1, const float* sol,
                         const float* keep, const float* W, const float* beta, const float* gamma, float* a, float* part,
                         const float* gS, const float* gI, const float* gR, const gnode_params* p, const float* dt_host,
                         const int* slot_of_prev, void* ctl, bool ctl_is_zero,
                         bool fold /* the adjoint is zero through interval G-1 (last grid point not emitted): start there */,
                         int* slots, hipStream_t st);
