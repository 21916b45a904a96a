// Shared host-side helpers of libgnode_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "gnode.h"

struct gnode_graph_s {
    int32_t n;
    int64_t nnz;
    int32_t max_degree;
    int32_t n_bigrow; // rows longer than GN_SIR_BIGROW (the Monte-Carlo kernel walks those with the whole workgroup; their list's capacity)
    int32_t device;   // the HIP device the CSR lives on (current device at gnode_graph_create)
    int32_t num_cu;   // its compute-unit count: persistent grids are sized from the handle, not from process globals
    int32_t* rowptr;  // device [n+1]
    int32_t* rowhdr;  // device [n][20]: {start, end, 0, 0, first 16 column ids (0-padded)} -- the H = 64 step kernel gets a
                      // row's extent AND its first 16 neighbour ids in ONE round trip instead of two dependent ones
    int32_t* col;     // device [nnz]
    // hub rows (gnode_hub.hip): rows longer than the hub threshold, cut into <= 32-edge segments
    int32_t n_hub, n_seg;
    int32_t* hubidx;        // device [n]: hub index of a row, -1 for ordinary rows (null when n_hub == 0)
    int32_t* seg_lo;        // device [n_seg]: first CSR position of a segment
    int32_t* seg_hi;        // device [n_seg]: one past its last
    int32_t* hub_seg_ptr;   // device [n_hub+1]: segments of hub h are [ptr[h], ptr[h+1])
    // persistent one-launch integration (gnode_pers64.hip): for 1 / 2 / 4 tiles per workgroup the node each lane-group slot
    // owns, -1 for padding slots; null when the graph never takes that path
    int32_t* persmap[3];
    // ... and its hub rows (graphs with rows longer than GN_HUB_T): per lane-group slot {first partial slot in the workgroup's
    // LDS, segments} of the hub row it owns (-1, 0 otherwise); per lane-group slot {first item, items} of the segment sums it
    // computes each step; the items {first CSR position, one past the last, partial slot, 0}; partial slots per workgroup
    int32_t* pershub[3];
    int32_t* perssegptr[3];
    int32_t* perssegitem[3];
    int32_t perslds[3];
    int32_t persitems[3];
    // small hidden sizes (gnode_persg.hip; H = 8 / 16 / 32 x workgroups of 1 .. 4 waves): lane-group slot -> node (-1 padding),
    // hub rows dealt round-robin, all variants in one allocation at pgoff[][] (-1: variant absent); per variant the most
    // neighbour ids of ordinary rows and the most hub segments one workgroup has to stage in LDS
    int32_t* pgmap;
    int32_t pgoff[3][4], pgids[3][4], pgsegs[3][4];
};

#ifndef GN_SIR_BIGROW
#define GN_SIR_BIGROW 512     // Monte-Carlo frontier kernel: rows longer than this are walked by the whole workgroup
#endif
#define HUB_SEG 32           // a hub row's neighbour list is cut into segments of this many edges
#ifndef GN_HUB_T
#define GN_HUB_T 96          // rows longer than this are hubs (measured break-even against the two extra launches per step)
#endif
int gn_hub_build(gnode_graph_s* g, const int32_t* rowptr_host);
int gn_pers64_build(gnode_graph_s* g, const int32_t* rowptr_host);     // the row maps above (gnode_pers64.hip)
void gn_pers64_free(gnode_graph_s* g);
int gn_persg_build(gnode_graph_s* g, const int32_t* rowptr_host);      // row maps above (gnode_persg.hip)
void gn_persg_free(gnode_graph_s* g);
void gn_hub_free(gnode_graph_s* g);
// Hub sums of `ntables` (1 or 2) tables for a batch of B samples need this much of the CALLER's workspace (0 for a graph
// without hub rows); gn_hub_gather carves its segment partials and hub sums from it: no allocation, no
// synchronisation, nothing retained in the handle.
size_t gn_hub_scratch_bytes(const gnode_graph_s* g, long B, int H, int ntables);
int gn_hub_gather(const gnode_graph_s* g, long B, int H, const float* T0, const float* T1, void* scratch, const float** A0,
                  const float** A1, hipStream_t st);
int gn_hub_segments(const gnode_graph_s* g, long B, int H, const float* T0, void* scratch, const float** P0, hipStream_t st);
int gn_hub_segments2(const gnode_graph_s* g, long B, int H, const float* T0, const float* T1, void* scratch, const float** P0,
                     const float** P1, hipStream_t st);

// per-device one-time setup (dynamic-LDS attributes of every kernel that may need more than 64 KB), run by
// gnode_graph_create for the current device; each translation unit contributes its kernels
int gn_device_setup_once(int dev);      // idempotent, locked; returns a gnode_status
int gn_ode_set_attributes();
int gn_bwd_set_attributes();
int gn_bwd_tiny_set_attributes();
int gn_sir_set_attributes();

// The training forward's KEPT ACTIVATIONS (H = 64): per grid point k three tables of rows + 1 rows of 64 floats --
//   Z_S(y_k); Z_I(y_k); P_S(y_k) = (A Z_I(y_k)) * Z_S(y_k) * (1 - Z_S(y_k)).
// Z_I(y_k) IS step k's gather table (the row behind it is the table's zero row), so keeping it costs the forward nothing;
// Z_S and P_S are one streamed slab each per step (P_S takes the place of the A Z_I row a forward without `keep` parks in the
// trajectory's 4th slab).  The adjoint backward reads them back instead of gathering a second table and recomputing three
// 64x64 products and 192 sigmoids per row and interval; P_S is the ONLY form in which it needs A Z_I, so it reads one slab
// row where A Z_I and Z_S(y_i) would be two.  The one-launch (tiny-graph) forms keep and use Z_S, Z_I only.
__host__ __device__ static inline size_t gn_keep_stride(long rows) { return ((size_t)rows + 1) * 64; }
static inline size_t gn_keep_floats(long rows, int n_steps) { return (size_t)3 * (n_steps + 1) * gn_keep_stride(rows); }
template <class T> __host__ __device__ static inline T* gn_keep_zs(T* keep, long rows, int k) { return keep + (size_t)(3 * k) * gn_keep_stride(rows); }
template <class T> __host__ __device__ static inline T* gn_keep_zi(T* keep, long rows, int k) { return keep + (size_t)(3 * k + 1) * gn_keep_stride(rows); }
template <class T> __host__ __device__ static inline T* gn_keep_ps(T* keep, long rows, int k) { return keep + (size_t)(3 * k + 2) * gn_keep_stride(rows); }

void gnode_set_error(const char* fmt, ...);

// Zero `bytes` (a multiple of 4) at `p` (4-byte aligned) on the stream, with a KERNEL: hipMemsetAsync becomes a memset node when a
// caller captures the call into a HIP graph, and on this ROCm such nodes were not re-executed reliably on the second and later
// replays (round 3: the trainer's replayed step summed stale gradient slots).  Kernel nodes replay in order.
int gn_zero_async(void* p, size_t bytes, hipStream_t st);
// several regions in ONE launch (each 4-byte aligned, a multiple of 4 bytes): the start-up zero-fills of a backward call are
// launch latency, not bytes, on the small graphs the reference trains on
struct GnZeroRegions { void* p[6]; size_t bytes[6]; int n; };
int gn_zero_regions_async(const GnZeroRegions& r, hipStream_t st);

// opt-in launch profiler (gnode_profile_enable): bracket a launch of `kind` with HIP events when it is sampled
bool gn_prof_begin(int kind, hipStream_t st);
void gn_prof_end(int kind, hipStream_t st);

#define GN_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            gnode_set_error(__VA_ARGS__);       \
            return GNODE_ERR_ARG;               \
        }                                       \
    } while (0)

#define GN_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            gnode_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return GNODE_ERR_HIP;                                                           \
        }                                                                                   \
    } while (0)

#define GN_LAUNCH_CHECK() GN_HIP(hipGetLastError())

static inline size_t gn_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
