// GN-ODE hot path for MI355X (gfx950): C-ABI host code (graph handle, RHS, forward) and the
// generic-H kernels (any H % 4 == 0): node MLP, CSR pull-gather, SIR derivative, Euler/RK4
// update and fused read-out, one sub-wave lane group of H/4 lanes per node row with 16-byte
// lane accesses.  H = 64 runs the fused MFMA kernels of gnode_h64.hip.
//
// Reference semantics restated (file:line into the reference tree):
//   ODEfunc.forward      ode_nn_ngraph_sim.py:58-96   (multi: ode_nn_ngraphs.py:54-83)
//   ODEBlock.forward     ode_nn_ngraph_sim.py:148-188 (multi: ode_nn_ngraphs.py:124-152)
//   odeint(euler)        torchdiffeq 0.2.2 fixed grid (call site :168)
//   get_sir_t_nodes_torch ode_nn.py:249-261 (fused as an output-row list)
//
// Compiled with -ffp-contract=off: the elementwise SIR update rounds exactly
// like the reference's separate torch ops; FMAs are written explicitly where
// they are wanted (the generic node-MLP inner product).
#include "gnode_common.h"
#include "gnode_gather.h"
#include "gnode_generic.h"
#include "gnode_h64.h"
#include <algorithm>

// --------------------------------------------------------------------------- error state
static thread_local char g_err[512] = "";
void gnode_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* gnode_last_error(void) { return g_err; }

__global__ __launch_bounds__(256) void k_zero_words(uint32_t* __restrict__ p, size_t nwords) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < nwords; i += stride) {
        if (i + 4 <= nwords && ((uintptr_t)(p + i) & 15u) == 0) *reinterpret_cast<uint4*>(p + i) = make_uint4(0u, 0u, 0u, 0u);
        else for (size_t j = i; j < nwords && j < i + 4; ++j) p[j] = 0u;
    }
}
int gn_zero_async(void* p, size_t bytes, hipStream_t st) {
    GN_CHECK_ARG(bytes % 4 == 0 && ((uintptr_t)p & 3u) == 0, "gn_zero_async: unaligned region");
    if (bytes == 0) return 0;
    const size_t nwords = bytes / 4;
    const unsigned grid = (unsigned)std::min<size_t>((nwords / 4 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(k_zero_words, dim3(grid), dim3(256), 0, st, (uint32_t*)p, nwords);
    GN_LAUNCH_CHECK();
    return 0;
}
__global__ __launch_bounds__(256) void k_zero_regions(GnZeroRegions r) {
    uint32_t* p = (uint32_t*)r.p[blockIdx.y];
    const size_t nwords = r.bytes[blockIdx.y] / 4;
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < nwords; i += stride) {
        if (i + 4 <= nwords && ((uintptr_t)(p + i) & 15u) == 0) *reinterpret_cast<uint4*>(p + i) = make_uint4(0u, 0u, 0u, 0u);
        else for (size_t j = i; j < nwords && j < i + 4; ++j) p[j] = 0u;
    }
}
int gn_zero_regions_async(const GnZeroRegions& r, hipStream_t st) {
    size_t mx = 0;
    for (int k = 0; k < r.n; ++k) {
        GN_CHECK_ARG(r.bytes[k] % 4 == 0 && ((uintptr_t)r.p[k] & 3u) == 0, "gn_zero_regions_async: unaligned region");
        mx = std::max(mx, r.bytes[k]);
    }
    if (r.n <= 0 || mx == 0) return 0;
    const unsigned gx = (unsigned)std::min<size_t>((mx / 16 + 255) / 256 + 1, 1024);
    hipLaunchKernelGGL(k_zero_regions, dim3(gx, (unsigned)r.n), dim3(256), 0, st, r);
    GN_LAUNCH_CHECK();
    return 0;
}
extern "C" int gnode_version(void) { return 221; }   // 221: gnode_backward_status, persistent launches for hidden 8 / 16 / 32 (gnode_forward_path kind 3); 220: forward takes flags + reports what sol / keep carry (sol_info), backward checks it; persistent one-launch path for mid-size graphs; 200: workspace sizes take the graph handle (hub scratch is carved from the caller's workspace); 210: forward / backward take the optional kept-activation buffer

// --------------------------------------------------------------------------- instrumentation
// HIP-event pairs around every launch of the two step kernels while enabled
// (bench.py's roofline leg: average launch duration measured on the launch stream).
// PROCESS-WIDE state, documented as such in gnode.h: a measuring harness switches it on around a single-threaded
// region; it is off by default and the launch paths then touch none of it.
#include <vector>
struct ProfKind { std::vector<hipEvent_t> ev; size_t used = 0; };
static bool g_prof_on = false;
static const int kProfKinds = 4;   // 0 = step (gather/update) kernel, 1 = node-MLP kernel, 2 = backward interval kernel, 3 = Monte-Carlo kernel
static ProfKind g_prof[kProfKinds];

static const int kProfEvery = 7;   // bracket one launch in 7 (odd: output and non-output intervals alternate): the events themselves cost GPU time
static long g_prof_seq[kProfKinds] = {0, 0, 0, 0};
static bool prof_begin(int kind, hipStream_t st);
static void prof_mark(int kind, hipStream_t st) {
    if (!g_prof_on) return;
    ProfKind& k = g_prof[kind];
    if (k.used == k.ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        k.ev.push_back(e);
    }
    (void)hipEventRecord(k.ev[k.used++], st);
}

// begin(): returns true when this launch is sampled; the caller then calls prof_mark again after the launch
static bool prof_begin(int kind, hipStream_t st) {
    if (!g_prof_on) return false;
    if ((g_prof_seq[kind]++ % kProfEvery) != 0) return false;
    prof_mark(kind, st);
    return true;
}
bool gn_prof_begin(int kind, hipStream_t st) { return prof_begin(kind, st); }      // for the other translation units
void gn_prof_end(int kind, hipStream_t st) { prof_mark(kind, st); }

extern "C" int gnode_profile_enable(int on) {
    g_prof_on = on != 0;
    for (int k = 0; k < kProfKinds; ++k) { g_prof_seq[k] = 0; g_prof[k].used = 0; }
    return 0;
}

extern "C" int gnode_profile_read_kind(int32_t kind, double* ms_out, int64_t* launches_out) {
    GN_CHECK_ARG(kind >= 0 && kind < kProfKinds, "gnode_profile_read_kind: kind %d out of range", kind);
    double ms = 0.0;
    int64_t cnt = 0;
    ProfKind& pk = g_prof[kind];
    for (size_t i = 0; i + 1 < pk.used; i += 2) {
        GN_HIP(hipEventSynchronize(pk.ev[i + 1]));
        float t = 0.f;
        GN_HIP(hipEventElapsedTime(&t, pk.ev[i], pk.ev[i + 1]));
        ms += t;
        cnt += 1;
    }
    if (ms_out) *ms_out = ms;
    if (launches_out) *launches_out = cnt;
    return 0;
}

extern "C" int gnode_profile_read(double* gather_ms, int64_t* gather_launches, double* mlp_ms, int64_t* mlp_launches) {
    double ms[2] = {0.0, 0.0};
    int64_t cnt[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {
        ProfKind& pk = g_prof[k];
        for (size_t i = 0; i + 1 < pk.used; i += 2) {
            GN_HIP(hipEventSynchronize(pk.ev[i + 1]));
            float t = 0.f;
            GN_HIP(hipEventElapsedTime(&t, pk.ev[i], pk.ev[i + 1]));
            ms[k] += t;
            cnt[k] += 1;
        }
    }
    if (gather_ms) *gather_ms = ms[0];
    if (gather_launches) *gather_launches = cnt[0];
    if (mlp_ms) *mlp_ms = ms[1];
    if (mlp_launches) *mlp_launches = cnt[1];
    return 0;
}

// --------------------------------------------------------------------------- K0: encoder
// y0 = cat(relu(Linear(1,H)(S0)), ..(I0), ..(R0), beta_gamma)   ode_nn_ngraph_sim.py:149-168
// x [rows, 3+H]; Y [3][rows][H]; beta/gamma [rows]; optional sol0 [4*rows, H].
template <int LPR>
__global__ __launch_bounds__(256) void k_encode(const float* __restrict__ x, const float* __restrict__ w1,
                                                const float* __restrict__ b1, float* __restrict__ Y,
                                                float* __restrict__ beta, float* __restrict__ gamma,
                                                float* __restrict__ sol0, long rows, int H) {
    const int sub = threadIdx.x % LPR;
    const long r = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (r >= rows || 4 * sub >= H) return;
    const float* xr = x + r * (3 + H);
    const float s0 = xr[0], i0 = xr[1], r0 = xr[2];
    const float4 w = ld4(w1 + 4 * sub), b = ld4(b1 + 4 * sub);
    auto enc = [&](float v) {
        return make_float4(fmaxf(fmaf(w.x, v, b.x), 0.f), fmaxf(fmaf(w.y, v, b.y), 0.f),
                           fmaxf(fmaf(w.z, v, b.z), 0.f), fmaxf(fmaf(w.w, v, b.w), 0.f));
    };
    const size_t slab = (size_t)rows * H, off = (size_t)r * H + 4 * sub;
    const float4 yS = enc(s0), yI = enc(i0), yR = enc(r0);
    st4(Y + off, yS); st4(Y + slab + off, yI); st4(Y + 2 * slab + off, yR);
    if (sub == 0) { beta[r] = xr[3]; gamma[r] = xr[4]; }
    if (sol0) {
        st4(sol0 + off, yS); st4(sol0 + slab + off, yI); st4(sol0 + 2 * slab + off, yR);
        const float* bg = xr + 3 + 4 * sub;
        st4(sol0 + 3 * slab + off, make_float4(bg[0], bg[1], bg[2], bg[3]));
    }
}

// Everything before the first Euler step of the generic-H fused path in ONE launch: encoder (k_encode), read-out at grid
// point 0 (k_readout) and Z_I(y_0) (k_mlp_generic's arithmetic in k_step_generic's lane-group form: same fma chain from the
// bias, same sigmoid -> the same bits), plus the zero-fill of the persistent launch's control block when one follows.
template <int LPR>
__global__ __launch_bounds__(256) void k_prologue_generic(const float* __restrict__ x, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, const float* __restrict__ W,
                                                          const float* __restrict__ bias, const float* __restrict__ w3,
                                                          const float* __restrict__ b3, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, float* __restrict__ Y,
                                                          float* __restrict__ beta, float* __restrict__ gamma,
                                                          float* __restrict__ sol0, float* __restrict__ ZI, float* __restrict__ S0,
                                                          float* __restrict__ I0, float* __restrict__ R0, long rows, int H,
                                                          uint32_t* __restrict__ zero_words, int n_zero_words) {
    extern __shared__ float Wt[];                 // [H][H] transposed: Wt[k][j] = W[j][k]
    for (int idx = threadIdx.x; idx < H * H; idx += 256) Wt[(size_t)(idx % H) * H + idx / H] = W[idx];
    if (zero_words && blockIdx.x == 0)
        for (int i = threadIdx.x; i < n_zero_words; i += 256) zero_words[i] = 0u;
    __syncthreads();
    const int sub = threadIdx.x % LPR;
    const long r = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    const bool inrow = r < rows, active = 4 * sub < H, ok = inrow && active;     // (every lane stays: the mat-vec and the read-out shuffle)
    const float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 yS = z0, yI = z0, yR = z0;
    const size_t slab = (size_t)rows * H, off = (size_t)(inrow ? r : 0) * H + 4 * sub;
    if (ok) {
        const float* xr = x + r * (3 + H);
        const float s0 = xr[0], i0 = xr[1], r0 = xr[2];
        const float4 w = ld4(w1 + 4 * sub), b = ld4(b1 + 4 * sub);
        auto enc = [&](float v) {
            return make_float4(fmaxf(fmaf(w.x, v, b.x), 0.f), fmaxf(fmaf(w.y, v, b.y), 0.f),
                               fmaxf(fmaf(w.z, v, b.z), 0.f), fmaxf(fmaf(w.w, v, b.w), 0.f));
        };
        yS = enc(s0); yI = enc(i0); yR = enc(r0);
        st4(Y + off, yS); st4(Y + slab + off, yI); st4(Y + 2 * slab + off, yR);
        if (sub == 0) { beta[r] = xr[3]; gamma[r] = xr[4]; }
        if (sol0) {
            st4(sol0 + off, yS); st4(sol0 + slab + off, yI); st4(sol0 + 2 * slab + off, yR);
            const float* bg = xr + 3 + 4 * sub;
            st4(sol0 + 3 * slab + off, make_float4(bg[0], bg[1], bg[2], bg[3]));
        }
    }
    if (S0) {
        float pS, pI, pR;
        readout_row<LPR>(yS, yI, yR, active, sub, H, w3, b3, w2, b2, pS, pI, pR);
        if (sub == 0 && inrow) { S0[r] = pS; I0[r] = pI; R0[r] = pR; }
    }
    const float4 bias4 = active ? ld4(bias + 4 * sub) : z0;
    const float4 zi = group_mlp<LPR>(yI, Wt, bias4, sub, active, H);
    if (ok) st4(ZI + off, zi);
}

// --------------------------------------------------------------------------- K1: node MLP  Z = sigmoid(Y W^T + b)
// Generic FMA path (any H % 4 == 0, H <= 128): W^T staged in LDS, one LPR-lane
// group per row, each lane 4 output features.
template <int LPR>
__global__ __launch_bounds__(256) void k_mlp_generic(const float* __restrict__ X, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ Z,
                                                     long nrows, int H) {
    extern __shared__ float lds[];
    float* Wt = lds;                      // [H][H] transposed: Wt[k][j] = W[j][k]
    float* Xs = lds + (size_t)H * H;      // [256/LPR][H]
    for (int idx = threadIdx.x; idx < H * H; idx += 256) {
        int j = idx / H, k = idx % H;
        Wt[(size_t)k * H + j] = W[idx];
    }
    const int sub = threadIdx.x % LPR, g = threadIdx.x / LPR;
    const long r = (long)blockIdx.x * (256 / LPR) + g;
    const bool active = (r < nrows) && (4 * sub < H);
    if (active) st4(Xs + (size_t)g * H + 4 * sub, ld4(X + (size_t)r * H + 4 * sub));
    __syncthreads();
    if (!active) return;
    float4 acc = ld4(bias + 4 * sub);
    const float* xr = Xs + (size_t)g * H;
    for (int k = 0; k < H; ++k) {
        const float xv = xr[k];
        const float4 w = ld4(Wt + (size_t)k * H + 4 * sub);
        acc.x = fmaf(xv, w.x, acc.x); acc.y = fmaf(xv, w.y, acc.y);
        acc.z = fmaf(xv, w.z, acc.z); acc.w = fmaf(xv, w.w, acc.w);
    }
    st4(Z + (size_t)r * H + 4 * sub,
        make_float4(gn_sigmoid(acc.x), gn_sigmoid(acc.y), gn_sigmoid(acc.z), gn_sigmoid(acc.w)));
}

// --------------------------------------------------------------------------- K2: gather + SIR derivative (+ Euler update + read-out)
// AI[r] = sum_{c in adj(node)} Z_I[base + c] in ascending column order (the CPU
// scatter_add_ order of ode_nn_ngraph_sim.py:73), then :75-77.  One LPR-lane
// group per row; the group fetches LPR column indices with one coalesced load and
// broadcasts them by lane shuffle, so neighbour-row loads are issued back to back.
template <int LPR>
__device__ __forceinline__ float4 gather_row(const int* __restrict__ rowptr, const int* __restrict__ col,
                                             const float* __restrict__ ZI_base, int node, int sub, bool active, int H) {
    // small lane groups have registers to spare: more rows in flight, fewer dependent round trips per row
    return gn_gather1<(LPR <= 4 ? 16 : 8)>(col, rowptr[node], rowptr[node + 1], ZI_base, H, sub, active);
}

struct StepOut {
    float* S; float* I; float* R;   // this step's output rows [rows] or null
    float* sol;                     // sol[g+1] base ([4*rows,H]) or null
};

// MODE 0: derivative only (dY <- f(Y)), MODE 1: Euler update in place (+ outputs)
template <int LPR, int MODE>
__global__ __launch_bounds__(256) void k_gather(const int* __restrict__ rowptr, const int* __restrict__ col, int n,
                                                long rows, int H, float* __restrict__ Y, const float* __restrict__ Z,
                                                const float* __restrict__ beta, const float* __restrict__ gamma,
                                                int bg_stride, float dt, float* __restrict__ dY,
                                                const float* __restrict__ w3, const float* __restrict__ b3,
                                                const float* __restrict__ w2, const float* __restrict__ b2, StepOut out,
                                                const int* __restrict__ hubidx, const float* __restrict__ AIhub, int n_hub) {
    const int sub = threadIdx.x % LPR;
    const int node = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (node >= n) return;                       // whole lane group leaves together
    const bool active = 4 * sub < H;
    const long base = (long)blockIdx.y * n;      // block-diagonal: sample b owns rows [b*n, (b+1)*n)
    const long r = base + node;
    const size_t slab = (size_t)rows * H, off = (size_t)r * H + 4 * sub;
    const float* ZS = Z;
    const float* ZI = Z + slab;

    const int hub = hubidx ? hubidx[node] : -1;  // long rows were summed by the hub kernels (gnode_hub.hip)
    float4 ai;
    if (hub >= 0) ai = active ? ld4(AIhub + ((size_t)blockIdx.y * n_hub + hub) * H + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    else ai = gather_row<LPR>(rowptr, col, ZI + (size_t)base * H, node, sub, active, H);
    float4 zs = make_float4(0.f, 0.f, 0.f, 0.f), zi = zs;
    if (active) { zs = ld4(ZS + off); zi = ld4(ZI + off); }
    const float nb = -beta[(size_t)r * bg_stride], gm = gamma[(size_t)r * bg_stride];
    float4 dS, dI, dR;
    dS.x = nb * (ai.x * zs.x); dS.y = nb * (ai.y * zs.y); dS.z = nb * (ai.z * zs.z); dS.w = nb * (ai.w * zs.w);
    dR.x = gm * zi.x; dR.y = gm * zi.y; dR.z = gm * zi.z; dR.w = gm * zi.w;
    dI.x = -dS.x - dR.x; dI.y = -dS.y - dR.y; dI.z = -dS.z - dR.z; dI.w = -dS.w - dR.w;
    if (MODE == 0) {
        if (active) { st4(dY + off, dS); st4(dY + slab + off, dI); st4(dY + 2 * slab + off, dR); }
        return;
    }
    float4 yS = make_float4(0.f, 0.f, 0.f, 0.f), yI = yS, yR = yS;
    if (active) {
        yS = ld4(Y + off); yI = ld4(Y + slab + off); yR = ld4(Y + 2 * slab + off);
        yS.x += dt * dS.x; yS.y += dt * dS.y; yS.z += dt * dS.z; yS.w += dt * dS.w;
        yI.x += dt * dI.x; yI.y += dt * dI.y; yI.z += dt * dI.z; yI.w += dt * dI.w;
        yR.x += dt * dR.x; yR.y += dt * dR.y; yR.z += dt * dR.z; yR.w += dt * dR.w;
        st4(Y + off, yS); st4(Y + slab + off, yI); st4(Y + 2 * slab + off, yR);
        if (out.sol) { st4(out.sol + off, yS); st4(out.sol + slab + off, yI); st4(out.sol + 2 * slab + off, yR); }
    }
    if (out.S) {
        float pS, pI, pR;
        readout_row<LPR>(yS, yI, yR, active, sub, H, w3, b3, w2, b2, pS, pI, pR);
        if (sub == 0) { out.S[r] = pS; out.I[r] = pI; out.R[r] = pR; }
    }
}

// Fused Euler step for generic H (the multi-graph launcher trains with H = 8): the same launch
// structure as k_step64 -- gather, Z_S from Y_S, update, read-out, next step's Z_I -- with the node
// MLP as a lane-group mat-vec: the row's H values live 4 per lane, x_k is broadcast inside the group
// by shuffle and multiplied with W^T (staged in LDS, [k][j]).
// HUBS (compile time, as k_step64): graphs without long rows carry none of that code and none of its registers -- the hub branch
// keeps 32 partial rows in flight at the small lane-group sizes (168 VGPRs against 120)
template <int LPR, bool HUBS>
__global__ __launch_bounds__(256) void k_step_generic(const int* __restrict__ rowptr, const int* __restrict__ col, int n,
                                                      long rows, int H, float* Y,
                                                      const float* __restrict__ ZI, float* __restrict__ ZI_next,
                                                      const float* __restrict__ W, const float* __restrict__ bias,
                                                      const float* __restrict__ beta, const float* __restrict__ gamma,
                                                      float dt, const float* __restrict__ w3, const float* __restrict__ b3,
                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                      StepOut out, const int* __restrict__ hubidx,
                                                      const float* __restrict__ HubP /* hub rows: per-segment partial sums [B][n_seg][H] */,
                                                      const int* __restrict__ hub_seg_ptr, int n_seg) {
    extern __shared__ float Wt[];                 // [H][H] transposed: Wt[k][j] = W[j][k]
    for (int idx = threadIdx.x; idx < H * H; idx += 256) Wt[(size_t)(idx % H) * H + idx / H] = W[idx];
    __syncthreads();
    const int sub = threadIdx.x % LPR;
    const int node = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (node >= n) return;                        // whole lane group leaves together (no barrier after this point)
    const bool active = 4 * sub < H;
    const long base = (long)blockIdx.y * n, r = base + node;
    const size_t slab = (size_t)rows * H, off = (size_t)r * H + 4 * sub;
    const float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bias4 = active ? ld4(bias + 4 * sub) : z0;

    // own-row loads first: they travel under the gather's dependent id -> row round trips instead of after them
    float4 yS = z0, yI = z0, yR = z0, zi = z0;
    if (active) { yS = ld4(Y + off); yI = ld4(Y + slab + off); yR = ld4(Y + 2 * slab + off); zi = ld4(ZI + off); }
    const float nb = -beta[r], gm = gamma[r];
    const int hub = HUBS ? hubidx[node] : -1;
    float4 ai;
    if (HUBS && hub >= 0) {
        // long rows: their 32-edge segments were summed by k_hub_seg; add the partials up in segment order (what a separate
        // reduction launch used to do -- at this size a launch costs as much as the step), 8 in flight
        ai = z0;
        const float* pp = HubP + (size_t)blockIdx.y * n_seg * H + 4 * sub;
        const int s1 = hub_seg_ptr[hub + 1];
        // (one lane group walks a hub's partials: a 12 777-edge row is 400 of them, and at 8 in flight that single chain was 30 us of
        //  a 76 us step at H = 8 -- 32 in flight for the small lane groups, which have the registers; same order, same bits)
#ifndef GN_HUB_PF
#define GN_HUB_PF 32
#endif
        constexpr int PF = LPR <= 4 ? GN_HUB_PF : 8;
        for (int sg = hub_seg_ptr[hub]; sg < s1; sg += PF) {
            float4 u[PF];
#pragma unroll
            for (int q = 0; q < PF; ++q) u[q] = (active && sg + q < s1) ? ld4(pp + (size_t)(sg + q) * H) : z0;
#pragma unroll
            for (int q = 0; q < PF; ++q) { ai.x += u[q].x; ai.y += u[q].y; ai.z += u[q].z; ai.w += u[q].w; }
        }
    } else ai = gather_row<LPR>(rowptr, col, ZI + (size_t)base * H, node, sub, active, H);
    const float4 zs = group_mlp<LPR>(yS, Wt, bias4, sub, active, H);
    float4 dS, dI, dR;
    dS.x = nb * (ai.x * zs.x); dS.y = nb * (ai.y * zs.y); dS.z = nb * (ai.z * zs.z); dS.w = nb * (ai.w * zs.w);
    dR.x = gm * zi.x; dR.y = gm * zi.y; dR.z = gm * zi.z; dR.w = gm * zi.w;
    dI.x = -dS.x - dR.x; dI.y = -dS.y - dR.y; dI.z = -dS.z - dR.z; dI.w = -dS.w - dR.w;
    yS.x += dt * dS.x; yS.y += dt * dS.y; yS.z += dt * dS.z; yS.w += dt * dS.w;
    yI.x += dt * dI.x; yI.y += dt * dI.y; yI.z += dt * dI.z; yI.w += dt * dI.w;
    yR.x += dt * dR.x; yR.y += dt * dR.y; yR.z += dt * dR.z; yR.w += dt * dR.w;
    if (active) {                                 // in place, or trajectory point k -> k+1 (the trajectory is the state)
        float* Yo = out.sol ? out.sol : Y;
        st4(Yo + off, yS); st4(Yo + slab + off, yI); st4(Yo + 2 * slab + off, yR);
    }
    if (out.S) {
        float pS, pI, pR;
        readout_row<LPR>(yS, yI, yR, active, sub, H, w3, b3, w2, b2, pS, pI, pR);
        if (sub == 0) { out.S[r] = pS; out.I[r] = pI; out.R[r] = pR; }
    }
    const float4 zn = group_mlp<LPR>(yI, Wt, bias4, sub, active, H);     // Z_I of the next step
    if (active) st4(ZI_next + off, zn);
}

// Stand-alone read-out of a state (grid point 0, and every point under RK4).
template <int LPR>
__global__ __launch_bounds__(256) void k_readout(const float* __restrict__ Y, long rows, int H,
                                                 const float* __restrict__ w3, const float* __restrict__ b3,
                                                 const float* __restrict__ w2, const float* __restrict__ b2,
                                                 float* __restrict__ S, float* __restrict__ I, float* __restrict__ R) {
    const int sub = threadIdx.x % LPR;
    const long r = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (r >= rows) return;
    const bool active = 4 * sub < H;
    const size_t slab = (size_t)rows * H, off = (size_t)r * H + 4 * sub;
    float4 yS = make_float4(0.f, 0.f, 0.f, 0.f), yI = yS, yR = yS;
    if (active) { yS = ld4(Y + off); yI = ld4(Y + slab + off); yR = ld4(Y + 2 * slab + off); }
    float pS, pI, pR;
    readout_row<LPR>(yS, yI, yR, active, sub, H, w3, b3, w2, b2, pS, pI, pR);
    if (sub == 0) { S[r] = pS; I[r] = pI; R[r] = pR; }
}

// out = y + a * (c1*k1 + c2*k2 + c3*k3 + c4*k4)   (RK4 3/8-rule stage combinations)
__global__ __launch_bounds__(256) void k_lincomb(float* __restrict__ out, const float* __restrict__ y, float a,
                                                 const float* __restrict__ k1, float c1, const float* __restrict__ k2,
                                                 float c2, const float* __restrict__ k3, float c3,
                                                 const float* __restrict__ k4, float c4, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = ld4(k1 + 4 * i);
        float4 s = make_float4(c1 * v.x, c1 * v.y, c1 * v.z, c1 * v.w);
        if (k2) { v = ld4(k2 + 4 * i); s.x += c2 * v.x; s.y += c2 * v.y; s.z += c2 * v.z; s.w += c2 * v.w; }
        if (k3) { v = ld4(k3 + 4 * i); s.x += c3 * v.x; s.y += c3 * v.y; s.z += c3 * v.z; s.w += c3 * v.w; }
        if (k4) { v = ld4(k4 + 4 * i); s.x += c4 * v.x; s.y += c4 * v.y; s.z += c4 * v.z; s.w += c4 * v.w; }
        v = ld4(y + 4 * i);
        st4(out + 4 * i, make_float4(v.x + a * s.x, v.y + a * s.y, v.z + a * s.z, v.w + a * s.w));
    }
}

// sol[g][3rd slab] = sol[0][3rd slab] for g = 1 .. G-1: the beta-gamma slab rides along unchanged
// (its derivative is 0, ode_nn_ngraph_sim.py:96); one launch instead of one copy per step.
__global__ __launch_bounds__(256) void k_fill_bg(float* __restrict__ sol, size_t slab4, int G) {
    const float* src = sol + 3 * slab4 * 4;                  // slab4 = floats per slab / 4
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < slab4; i += (size_t)gridDim.x * 256) {
        const float4 v = ld4(src + 4 * i);
        for (int g = 1; g < G; ++g) st4(sol + ((size_t)g * 4 + 3) * slab4 * 4 + 4 * i, v);
    }
}

// --------------------------------------------------------------------------- host side
static int lpr_for(int H) {
    int need = H / 4, l = 1;
    while (l < need) l <<= 1;
    return l;
}

static int check_H(int H) {
    if (H < 4 || H > 256 || (H % 4) != 0) {
        gnode_set_error("unsupported hidden size H=%d (need 4 <= H <= 256, H %% 4 == 0)", H);
        return GNODE_ERR_ARG;
    }
    return 0;
}

#define DISPATCH_LPR(lpr, ...)                                   \
    switch (lpr) {                                               \
        case 1: { constexpr int LPR = 1; __VA_ARGS__; } break;   \
        case 2: { constexpr int LPR = 2; __VA_ARGS__; } break;   \
        case 4: { constexpr int LPR = 4; __VA_ARGS__; } break;   \
        case 8: { constexpr int LPR = 8; __VA_ARGS__; } break;   \
        case 16: { constexpr int LPR = 16; __VA_ARGS__; } break; \
        case 32: { constexpr int LPR = 32; __VA_ARGS__; } break; \
        default: { constexpr int LPR = 64; __VA_ARGS__; } break; \
    }

static int launch_mlp(const gnode_graph_s* g, const float* X, const float* W, const float* b, float* Z, long nrows, int H, hipStream_t st) {
    if (nrows == 0) return 0;
    const bool sampled = prof_begin(1, st);
    if (H == 64) {
        if (int e = gn_launch_mlp64(g, X, W, b, Z, nrows, st)) return e;
    } else if (H == 128) {
        if (int e = gn_launch_mlp128(g, X, W, b, Z, nrows, st)) return e;    // matrix cores (gnode_h128.hip)
    } else {
        GN_CHECK_ARG(H <= 128, "generic node-MLP path supports H <= 128 (got %d)", H);
        const int lpr = lpr_for(H);
        const int rpw = 256 / lpr;
        const size_t lds = ((size_t)H * H + (size_t)rpw * H) * sizeof(float);
        DISPATCH_LPR(lpr, hipLaunchKernelGGL(k_mlp_generic<LPR>, dim3((unsigned)((nrows + rpw - 1) / rpw)), dim3(256), lds, st, X, W, b, Z, nrows, H));
    }
    if (sampled) prof_mark(1, st);
    GN_LAUNCH_CHECK();
    return 0;
}

int gn_launch_mlp_any(const gnode_graph_s* g, const float* X, const float* W, const float* b, float* Z, long nrows, int H,
                      hipStream_t st) {
    return launch_mlp(g, X, W, b, Z, nrows, H, st);
}

// dynamic LDS above 64 KB (H = 128 on the generic node-MLP: 68 KB) needs the attribute once per device
int gn_ode_set_attributes() {
    GN_HIP(hipFuncSetAttribute((const void*)k_mlp_generic<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    return 0;
}

// dY or in-place Euler update from (Y, Z)
static int launch_gather(gnode_graph_t g, int mode, long rows, int H, float* Y, const float* Z, const float* beta,
                         const float* gamma, int bg_stride, float dt, float* dY, const gnode_params* p, StepOut out,
                         void* hub_scratch, hipStream_t st) {
    const int lpr = lpr_for(H), rpw = 256 / lpr;
    const long B = rows / g->n;
    dim3 grid((unsigned)((g->n + rpw - 1) / rpw), (unsigned)B);
    const float *w3 = p ? p->linear3_weight : nullptr, *b3 = p ? p->linear3_bias : nullptr;
    const float *w2 = p ? p->linearS2_weight : nullptr, *b2 = p ? p->linearS2_bias : nullptr;
    const float* AIhub = nullptr;
    if (int e = gn_hub_gather(g, B, H, Z + (size_t)rows * H, nullptr, hub_scratch, &AIhub, nullptr, st)) return e;
    const bool sampled = mode == 1 && prof_begin(0, st);
    if (mode == 0) {
        DISPATCH_LPR(lpr, hipLaunchKernelGGL((k_gather<LPR, 0>), grid, dim3(256), 0, st, g->rowptr, g->col, g->n, rows, H,
                                             Y, Z, beta, gamma, bg_stride, dt, dY, w3, b3, w2, b2, out, g->hubidx, AIhub, g->n_hub));
    } else {
        DISPATCH_LPR(lpr, hipLaunchKernelGGL((k_gather<LPR, 1>), grid, dim3(256), 0, st, g->rowptr, g->col, g->n, rows, H,
                                             Y, Z, beta, gamma, bg_stride, dt, dY, w3, b3, w2, b2, out, g->hubidx, AIhub, g->n_hub));
    }
    if (sampled) prof_mark(0, st);
    GN_LAUNCH_CHECK();
    return 0;
}

static int launch_readout(const float* Y, long rows, int H, const gnode_params* p, float* S, float* I, float* R,
                          hipStream_t st) {
    const int lpr = lpr_for(H), rpw = 256 / lpr;
    DISPATCH_LPR(lpr, hipLaunchKernelGGL(k_readout<LPR>, dim3((unsigned)((rows + rpw - 1) / rpw)), dim3(256), 0, st, Y,
                                         rows, H, p->linear3_weight, p->linear3_bias, p->linearS2_weight,
                                         p->linearS2_bias, S, I, R));
    GN_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------- graph handle
// Per-device one-time setup, done when the first graph handle is created on a device (never on a launch path, so
// never inside a stream capture): dynamic-LDS attributes of every kernel that can ask for more than 64 KB, and the
// device's CU count.  The only process-wide state besides the opt-in profiler: write-once per device, under a lock.
#include <mutex>
#include "gnode_pers64.h"
#include "gnode_persg.h"
static std::mutex g_dev_mu;
static bool g_dev_done[64] = {};
static int g_dev_cu[64] = {};
int gn_device_setup_once(int dev) {
    GN_CHECK_ARG(dev >= 0 && dev < 64, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (g_dev_done[dev]) return 0;
    hipDeviceProp_t prop;
    GN_HIP(hipGetDeviceProperties(&prop, dev));
    g_dev_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (int e = gn_ode_set_attributes()) return e;
    if (int e = gn_h64_set_attributes()) return e;
    if (int e = gn_pers64_set_attributes()) return e;
    if (int e = gn_pers_bwd64_set_attributes()) return e;
    if (int e = gn_persg_set_attributes()) return e;
    if (int e = gn_h128_set_attributes()) return e;
    if (int e = gn_bwd_set_attributes()) return e;
    if (int e = gn_bwd_tiny_set_attributes()) return e;
    if (int e = gn_sir_set_attributes()) return e;
    g_dev_done[dev] = true;
    return 0;
}

extern "C" int gnode_graph_create(const int32_t* rowptr_host, const int32_t* col_host, int32_t n, int64_t nnz,
                                  gnode_graph_t* out) {
    GN_CHECK_ARG(rowptr_host && out && n > 0 && nnz >= 0, "gnode_graph_create: null pointer or empty graph");
    GN_CHECK_ARG(col_host || nnz == 0, "gnode_graph_create: col is null");
    GN_CHECK_ARG(rowptr_host[0] == 0 && rowptr_host[n] == nnz, "gnode_graph_create: rowptr[0] != 0 or rowptr[n] != nnz");
    int32_t maxdeg = 0, n_bigrow = 0;
    for (int32_t i = 0; i < n; ++i) {
        const int32_t d = rowptr_host[i + 1] - rowptr_host[i];
        GN_CHECK_ARG(d >= 0, "gnode_graph_create: rowptr not monotone at %d", i);
        maxdeg = d > maxdeg ? d : maxdeg;
        n_bigrow += d > GN_SIR_BIGROW;
    }
    for (int64_t e = 0; e < nnz; ++e)
        GN_CHECK_ARG(col_host[e] >= 0 && col_host[e] < n, "gnode_graph_create: col[%lld]=%d out of range",
                     (long long)e, col_host[e]);
    int dev = 0;
    GN_HIP(hipGetDevice(&dev));
    if (int e = gn_device_setup_once(dev)) return e;
    gnode_graph_s* g = new gnode_graph_s();
    g->n = n; g->nnz = nnz; g->max_degree = maxdeg; g->n_bigrow = n_bigrow; g->rowptr = nullptr; g->col = nullptr; g->rowhdr = nullptr;
    g->n_hub = g->n_seg = 0; g->hubidx = g->seg_lo = g->seg_hi = g->hub_seg_ptr = nullptr;
    g->device = dev;
    g->num_cu = g_dev_cu[dev];
    hipError_t e1 = hipMalloc(&g->rowptr, sizeof(int32_t) * (size_t)(n + 1));
    hipError_t e2 = hipMalloc(&g->col, sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
    if (e1 != hipSuccess || e2 != hipSuccess) {
        gnode_set_error("gnode_graph_create: hipMalloc failed");
        if (g->rowptr) (void)hipFree(g->rowptr);
        if (g->col) (void)hipFree(g->col);
        delete g;
        return GNODE_ERR_HIP;
    }
    hipError_t e3 = hipMemcpy(g->rowptr, rowptr_host, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyHostToDevice);
    hipError_t e4 = nnz ? hipMemcpy(g->col, col_host, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice) : hipSuccess;
    if (e3 != hipSuccess || e4 != hipSuccess) {
        gnode_set_error("gnode_graph_create: hipMemcpy failed: %s", hipGetErrorString(e3 != hipSuccess ? e3 : e4));
        (void)hipFree(g->rowptr);
        (void)hipFree(g->col);
        delete g;
        return GNODE_ERR_HIP;
    }
    {   // row headers (see gnode_common.h)
        std::vector<int32_t> hdr((size_t)n * 20, 0);
        for (int32_t r = 0; r < n; ++r) {
            const int32_t lo = rowptr_host[r], hi = rowptr_host[r + 1];
            hdr[(size_t)r * 20] = lo; hdr[(size_t)r * 20 + 1] = hi;
            for (int32_t k = 0; k < 16 && lo + k < hi; ++k) hdr[(size_t)r * 20 + 4 + k] = col_host[lo + k];
        }
        hipError_t e5 = hipMalloc(&g->rowhdr, sizeof(int32_t) * hdr.size());
        if (e5 == hipSuccess) e5 = hipMemcpy(g->rowhdr, hdr.data(), sizeof(int32_t) * hdr.size(), hipMemcpyHostToDevice);
        if (e5 != hipSuccess) {
            gnode_set_error("gnode_graph_create: row headers: %s", hipGetErrorString(e5));
            if (g->rowhdr) (void)hipFree(g->rowhdr);
            (void)hipFree(g->rowptr);
            (void)hipFree(g->col);
            delete g;
            return GNODE_ERR_HIP;
        }
    }
    for (int i = 0; i < 3; ++i) { g->persmap[i] = g->pershub[i] = g->perssegptr[i] = g->perssegitem[i] = nullptr; g->perslds[i] = 0; }
    int e_build = gn_hub_build(g, rowptr_host);
    if (!e_build) e_build = gn_pers64_build(g, rowptr_host);
    g->pgmap = nullptr;
    if (!e_build) e_build = gn_persg_build(g, rowptr_host);
    if (int e = e_build) {
        (void)hipFree(g->rowhdr);
        gn_pers64_free(g);
        gn_persg_free(g);
        gn_hub_free(g);
        (void)hipFree(g->rowptr);
        (void)hipFree(g->col);
        delete g;
        return e;
    }
    *out = g;
    return 0;
}

extern "C" int gnode_graph_destroy(gnode_graph_t g) {
    if (!g) return 0;
    gn_pers64_free(g);
    gn_persg_free(g);
    gn_hub_free(g);
    (void)hipFree(g->rowptr);
    (void)hipFree(g->col);
    if (g->rowhdr) (void)hipFree(g->rowhdr);
    delete g;
    return 0;
}

extern "C" int gnode_graph_info(gnode_graph_t g, int32_t* n, int64_t* nnz, int32_t* max_degree) {
    GN_CHECK_ARG(g, "gnode_graph_info: null graph");
    if (n) *n = g->n;
    if (nnz) *nnz = g->nnz;
    if (max_degree) *max_degree = g->max_degree;
    return 0;
}

// --------------------------------------------------------------------------- RHS
extern "C" size_t gnode_rhs_workspace_bytes(gnode_graph_t g, int64_t rows, int32_t H) {
    if (!g || rows <= 0 || H <= 0) return 0;
    return gn_align((size_t)2 * rows * H * sizeof(float)) + gn_hub_scratch_bytes(g, rows / g->n, H, 1);
}

extern "C" int gnode_rhs_f32(gnode_graph_t g, const float* x, const float* W, const float* b, float* dx, int64_t rows,
                             int32_t H, void* workspace, size_t workspace_bytes, void* stream) {
    GN_CHECK_ARG(g && x && W && b && dx && workspace, "gnode_rhs_f32: null pointer");
    if (int e = check_H(H)) return e;
    GN_CHECK_ARG(H >= 2, "gnode_rhs_f32: H >= 2 required (beta, gamma live in columns 0, 1)");
    GN_CHECK_ARG(rows > 0 && rows % g->n == 0, "gnode_rhs_f32: rows=%lld is not a multiple of graph n=%d",
                 (long long)rows, g->n);
    if (workspace_bytes < gnode_rhs_workspace_bytes(g, rows, H)) {
        gnode_set_error("gnode_rhs_f32: workspace %zu < %zu", workspace_bytes, gnode_rhs_workspace_bytes(g, rows, H));
        return GNODE_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* Z = (float*)workspace;
    void* hub_scratch = (char*)workspace + gn_align((size_t)2 * rows * H * sizeof(float));
    const size_t slab = (size_t)rows * H;
    if (int e = launch_mlp(g, x, W, b, Z, 2 * rows, H, st)) return e;   // R' is dead work in the reference: skipped
    StepOut none = {nullptr, nullptr, nullptr, nullptr};
    if (int e = launch_gather(g, 0, rows, H, const_cast<float*>(x), Z, x + 3 * slab, x + 3 * slab + 1, H, 0.f, dx,
                              nullptr, none, hub_scratch, st))
        return e;
    if (int e = gn_zero_async(dx + 3 * slab, slab * sizeof(float), st)) return e;   // 4th slab derivative = 0 (:96)
    return 0;
}

// --------------------------------------------------------------------------- forward
// Which form the H = 64 Euler forward takes for a batch: 2 = ONE persistent launch (gnode_pers64.hip; preferred wherever
// its plan fits -- down to karate: it measured faster than the one-workgroup kernels, 0.11 vs 0.17 ms for 39 steps, and
// its adjoint sweep 0.18 vs 0.35 ms), 1 = one workgroup per sample (tiny graphs in batches too large for one resident
// grid), 0 = one launch per step.  The backward asks the same question (same arguments) to know what `sol` / `keep` hold.
int gn_forward_kind(const gnode_graph_s* g, long rows, int H, int method, int n_steps, int n_out, bool with_sol, int flags, PersPlan* plan) {
    if (method == 0 && H <= 32 && n_steps >= 1 && !(flags & GNODE_FWD_PER_STEP) && gn_persg_plan(g, rows, H, n_steps, nullptr)) return 3;
    if (!(H == 64 && method == 0) || n_steps < 1) return 0;
    PersPlan pl;
    if (!(flags & GNODE_FWD_PER_STEP) && gn_pers64_plan(g, rows / g->n, n_steps, &pl)) { if (plan) *plan = pl; return 2; }
    if (gn_tiny64_ok(g->n, n_steps, n_out, !with_sol)) return 1;
    return 0;
}

static size_t forward_fixed_bytes(int64_t rows, int32_t H, int32_t method) {
    const size_t slab = gn_align((size_t)rows * H * sizeof(float));
    size_t nslab = 5;                       // Y[3], Z[2]
    if (method == 1) nslab += 3 * 5;        // k1..k4, ytmp (3 slabs each)
    // + one 256-B ZERO ROW behind each of the two gather tables (H = 64 step kernel: rows shorter than the gather
    //   width read it instead of branching per neighbour)
    // + the control block of the persistent one-launch path (gnode_pers64.hip): tickets, barrier flags, give-up word
    return nslab * slab + 512 + 2 * gn_align((size_t)rows * sizeof(float)) + gn_align((size_t)rows * 4 * sizeof(float)) +
           gn_pers64_ctl_bytes();
}
static char* forward_ctl_ptr(void* workspace, int64_t rows, int32_t H, int32_t method) {
    return (char*)workspace + forward_fixed_bytes(rows, H, method) - gn_pers64_ctl_bytes();
}

extern "C" size_t gnode_forward_workspace_bytes(gnode_graph_t g, int64_t rows, int32_t H, int32_t method) {
    if (!g || rows <= 0 || H <= 0) return 0;
    return forward_fixed_bytes(rows, H, method) + gn_hub_scratch_bytes(g, rows / g->n, H, 1);
}

extern "C" int gnode_forward_f32(gnode_graph_t g, const float* x, const gnode_params* p, const float* dt_host,
                                 int32_t n_steps, int32_t method, const int32_t* out_rows_host, int32_t n_out, float* S,
                                 float* I, float* R, float* sol, float* keep, size_t keep_bytes, int64_t rows, int32_t H,
                                 void* workspace, size_t workspace_bytes, void* stream, int32_t flags, int32_t* sol_info_host) {
    GN_CHECK_ARG(g && x && p && S && I && R && workspace, "gnode_forward_f32: null pointer");
    GN_CHECK_ARG(n_steps >= 0 && (n_steps == 0 || dt_host), "gnode_forward_f32: bad n_steps/dt");
    GN_CHECK_ARG(method == 0 || method == 1, "gnode_forward_f32: method must be 0 (euler) or 1 (rk4)");
    if (int e = check_H(H)) return e;
    GN_CHECK_ARG(H >= 2, "gnode_forward_f32: H >= 2 required (beta, gamma live in columns 3, 4 of x)");
    GN_CHECK_ARG(rows > 0 && rows % g->n == 0, "gnode_forward_f32: rows=%lld is not a multiple of graph n=%d",
                 (long long)rows, g->n);
    GN_CHECK_ARG(p->odefunc_linear_weight && p->odefunc_linear_bias && p->linearS1_weight && p->linearS1_bias &&
                     p->linear3_weight && p->linear3_bias && p->linearS2_weight && p->linearS2_bias,
                 "gnode_forward_f32: null parameter pointer");
    if (workspace_bytes < gnode_forward_workspace_bytes(g, rows, H, method)) {
        gnode_set_error("gnode_forward_f32: workspace %zu < %zu", workspace_bytes,
                        gnode_forward_workspace_bytes(g, rows, H, method));
        return GNODE_ERR_WORKSPACE;
    }
    const int G = n_steps + 1;
    if (out_rows_host) {
        GN_CHECK_ARG(n_out >= 0, "gnode_forward_f32: n_out < 0");
        for (int i = 0; i < n_out; ++i)
            GN_CHECK_ARG(out_rows_host[i] >= 0 && out_rows_host[i] < G && (i == 0 || out_rows_host[i] > out_rows_host[i - 1]),
                         "gnode_forward_f32: out_rows must be ascending grid indices in [0,%d)", G);
    }
    // kept activations: only the fused H = 64 training path fills them (elsewhere the backward recomputes)
    if (!(keep && sol && method == 0 && gnode_forward_keep_bytes(g, rows, H, n_steps, out_rows_host ? n_out : G) > 0)) keep = nullptr;
    if (keep && keep_bytes < gnode_forward_keep_bytes(g, rows, H, n_steps, out_rows_host ? n_out : G)) {
        gnode_set_error("gnode_forward_f32: keep buffer %zu < %zu", keep_bytes,
                        gnode_forward_keep_bytes(g, rows, H, n_steps, out_rows_host ? n_out : G));
        return GNODE_ERR_WORKSPACE;
    }
    if (sol_info_host) {                         // what this call leaves in `sol` / `keep`: gnode_backward_f32 checks it
        const int n_emit = out_rows_host ? n_out : G;
        const int kind = gn_forward_kind(g, rows, H, method, n_steps, n_emit, sol != nullptr, flags, nullptr);
        *sol_info_host = !sol ? 0 : ((keep ? GNODE_SOL_KEEP : (H == 64 && method == 0 && n_steps >= 1 && kind != 1 ? GNODE_SOL_AI : 0)) |
                                     (kind == 1 ? GNODE_SOL_TINY : 0));
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t slab = (size_t)rows * H, slab_b = gn_align(slab * sizeof(float));
    char* ws = (char*)workspace;
    float* Y = (float*)ws;                       // [3][rows][H] (contiguous: slab strides are elements, not aligned bytes)
    // keep slabs element-contiguous: Y uses 3*slab floats inside 3 aligned slabs
    float* Z = (float*)(ws + 3 * slab_b);
    char* after_z = ws + 5 * slab_b + 512;     // Z: [table 0][zero row][table 1][zero row] on the H = 64 path
    float* beta = (float*)after_z;
    float* gamma = (float*)(after_z + gn_align((size_t)rows * sizeof(float)));
    float* prbuf = (float*)(after_z + 2 * gn_align((size_t)rows * sizeof(float)));
    float* rk = (float*)(after_z + 2 * gn_align((size_t)rows * sizeof(float)) + gn_align((size_t)rows * 4 * sizeof(float)));
    void* hub_scratch = ws + forward_fixed_bytes(rows, H, method);      // segment partials + hub sums (graphs with hub rows)

    const int lpr = lpr_for(H), rpw = 256 / lpr;
    int next_out = 0;  // index into the output list
    auto out_slot = [&](int gidx) -> int {   // which output row (or -1) grid point gidx is written to
        if (!out_rows_host) return gidx;
        if (next_out < n_out && out_rows_host[next_out] == gidx) return next_out++;
        return -1;
    };
    int slot = out_slot(0);

    // H = 64: fused step kernels (gnode_h64.hip)
    const bool h64 = (H == 64 && method == 0);
    float* zi_cur = Z;
    float* zi_nxt = Z + slab + (h64 ? 64 : 0);           // H = 64: each table is followed by its zero row
    if (keep) { zi_cur = gn_keep_zi(keep, rows, 0); zi_nxt = gn_keep_zi(keep, rows, 1); }   // step k gathers table k, fills k+1
    // inference (no trajectory requested): R only feeds the read-out -> carry its 4-float projection
    float* PR = (h64 && !sol) ? prbuf : nullptr;
    bool pre_zeroed_ctl = false;
    if (h64 && n_steps > 0) {
        // encoder, beta/gamma, trajectory point 0, read-out at grid point 0, projected R and Z_I(y_0): one launch
        // (the persistent launch's control block is zeroed by the prologue's first workgroup: one launch less in front of it)
        pre_zeroed_ctl = gn_forward_kind(g, rows, H, method, n_steps, out_rows_host ? n_out : G, sol != nullptr, flags, nullptr) == 2;
        if (int e = gn_launch_prologue64(x, p, Y, beta, gamma, sol, zi_cur, zi_nxt, PR, slot >= 0 ? S + (size_t)slot * rows : nullptr,
                                         slot >= 0 ? I + (size_t)slot * rows : nullptr,
                                         slot >= 0 ? R + (size_t)slot * rows : nullptr, rows,
                                         pre_zeroed_ctl ? forward_ctl_ptr(workspace, rows, H, method) : nullptr, gn_pers64_ctl_bytes(), st))
            return e;
    } else if (method == 0 && H < 128 && n_steps > 0) {
        // generic-H fused path: encoder, read-out at grid point 0 and Z_I(y_0) in one launch (+ the persistent launch's control block)
        pre_zeroed_ctl = gn_forward_kind(g, rows, H, method, n_steps, out_rows_host ? n_out : G, sol != nullptr, flags, nullptr) == 3;
        DISPATCH_LPR(lpr, hipLaunchKernelGGL(k_prologue_generic<LPR>, dim3((unsigned)((rows + rpw - 1) / rpw)), dim3(256),
                                             (size_t)H * H * sizeof(float), st, x, p->linearS1_weight, p->linearS1_bias,
                                             p->odefunc_linear_weight, p->odefunc_linear_bias, p->linear3_weight, p->linear3_bias,
                                             p->linearS2_weight, p->linearS2_bias, Y, beta, gamma, sol, zi_cur,
                                             slot >= 0 ? S + (size_t)slot * rows : nullptr, slot >= 0 ? I + (size_t)slot * rows : nullptr,
                                             slot >= 0 ? R + (size_t)slot * rows : nullptr, (long)rows, H,
                                             pre_zeroed_ctl ? (uint32_t*)forward_ctl_ptr(workspace, rows, H, method) : nullptr,
                                             (int)(gn_pers64_ctl_bytes() / 4)));
        GN_LAUNCH_CHECK();
    } else {
        DISPATCH_LPR(lpr, hipLaunchKernelGGL(k_encode<LPR>, dim3((unsigned)((rows + rpw - 1) / rpw)), dim3(256), 0, st, x,
                                             p->linearS1_weight, p->linearS1_bias, Y, beta, gamma, sol, (long)rows, H));
        GN_LAUNCH_CHECK();
        if (slot >= 0)
            if (int e = launch_readout(Y, rows, H, p, S + (size_t)slot * rows, I + (size_t)slot * rows, R + (size_t)slot * rows, st))
                return e;
        if (method == 0 && H < 128 && n_steps > 0)
            if (int e = launch_mlp(g, Y + slab, p->odefunc_linear_weight, p->odefunc_linear_bias, zi_cur, rows, H, st)) return e;
    }
    PersPlan plan;
    const int fkind = gn_forward_kind(g, rows, H, method, n_steps, out_rows_host ? n_out : G, sol != nullptr, flags, &plan);
    if (fkind == 2) {
        // mid-size graphs: ONE persistent launch, every workgroup keeps its rows in registers for all steps (gnode_pers64.hip)
        int slots[128];
        for (int k = 0; k < n_steps; ++k) slots[k] = out_slot(k + 1);
        const bool sampled = prof_begin(0, st);
        if (int e = gn_launch_pers64(g, plan, rows, Y, PR, zi_cur, zi_nxt, p->odefunc_linear_weight, p->odefunc_linear_bias, beta,
                                     gamma, dt_host, slots, n_steps, p, S, I, R, sol, keep, forward_ctl_ptr(workspace, rows, H, method), pre_zeroed_ctl, st))
            return e;
        if (sampled) prof_mark(0, st);
        return 0;
    }

    if (fkind == 3) {
        // small hidden sizes, batches that fit one resident grid: ONE persistent launch (gnode_persg.hip)
        PersgPlan gp;
        gn_persg_plan(g, rows, H, n_steps, &gp);
        int slots[128];
        for (int k = 0; k < n_steps; ++k) slots[k] = out_slot(k + 1);
        const bool sampled = prof_begin(0, st);
        if (int e = gn_launch_persg(g, gp, rows, H, Y, zi_cur, zi_nxt, beta, gamma, dt_host, slots, n_steps, p, S, I, R, sol,
                                    forward_ctl_ptr(workspace, rows, H, method), pre_zeroed_ctl, st))
            return e;
        if (sampled) prof_mark(0, st);
        if (sol) {
            const size_t slab4 = slab / 4;
            hipLaunchKernelGGL(k_fill_bg, dim3((unsigned)std::min<size_t>((slab4 + 255) / 256, 2048)), dim3(256), 0, st, sol, slab4, G);
            GN_LAUNCH_CHECK();
        }
        return 0;
    }

    if (fkind == 1) {
        // tiny graphs: the whole integration in one launch (one workgroup per sample, state in LDS)
        int slots[128];
        for (int k = 0; k < n_steps; ++k) slots[k] = out_slot(k + 1);
        if (int e = gn_launch_tiny64(g, rows, Y, zi_cur, PR, p->odefunc_linear_weight, p->odefunc_linear_bias, beta, gamma,
                                     dt_host, slots, n_steps, p, S, I, R, sol, keep, st))
            return e;
        if (sol) {
            const size_t slab4 = slab / 4;
            hipLaunchKernelGGL(k_fill_bg, dim3((unsigned)std::min<size_t>((slab4 + 255) / 256, 2048)), dim3(256), 0, st, sol, slab4, G);
            GN_LAUNCH_CHECK();
        }
        return 0;
    }
    for (int k = 0; k < n_steps; ++k) {
        const float dt = dt_host[k];
        slot = out_slot(k + 1);
        float* sol_next = sol ? sol + (size_t)(k + 1) * 4 * slab : nullptr;
        // with a trajectory the fused kernels read point k and write point k+1 (no separate state copy)
        float* Ycur = (sol && (h64 || (method == 0 && H <= 128))) ? sol + (size_t)k * 4 * slab : Y;
        if (h64) {
            // training: the 4th slab of sol[k] (k >= 1; its odeint content is the constant beta-gamma slab of sol[0])
            // receives A Z_I(y_k), which the adjoint backward would otherwise gather again
            Step64Out out = {slot >= 0 ? S + (size_t)slot * rows : nullptr, slot >= 0 ? I + (size_t)slot * rows : nullptr,
                             slot >= 0 ? R + (size_t)slot * rows : nullptr, sol_next,
                             keep ? (k >= 1 ? gn_keep_ps(keep, rows, k) : nullptr)          // with `keep`: P_S(y_k) there instead
                                  : ((sol && k >= 1) ? sol + (size_t)k * 4 * slab + 3 * slab : nullptr),
                             keep ? gn_keep_zs(keep, rows, k) : nullptr};
            const bool sampled = prof_begin(0, st);
            if (int e = gn_launch_step64(g, rows, Ycur, zi_cur, zi_nxt, p->odefunc_linear_weight, p->odefunc_linear_bias, beta,
                                         gamma, dt, p, PR, out, hub_scratch, st))
                return e;
            if (sampled) prof_mark(0, st);
            if (keep) { zi_cur = zi_nxt; zi_nxt = gn_keep_zi(keep, rows, std::min(k + 2, n_steps)); }
            else std::swap(zi_cur, zi_nxt);
        } else if (method == 0 && H < 128) {
            // generic H: one fused launch per step (gather + both node MLPs as lane-group mat-vecs); H = 128 takes the
            // two-launch branch below, whose node MLP runs on the matrix cores (a VALU mat-vec is 12x off the bound there)
            StepOut out = {slot >= 0 ? S + (size_t)slot * rows : nullptr, slot >= 0 ? I + (size_t)slot * rows : nullptr,
                           slot >= 0 ? R + (size_t)slot * rows : nullptr, sol_next};
            const float* HubP = nullptr;          // segment partials of the hub rows; the step kernel adds them up itself
            if (int e = gn_hub_segments(g, rows / g->n, H, zi_cur, hub_scratch, &HubP, st)) return e;
            dim3 grid((unsigned)((g->n + rpw - 1) / rpw), (unsigned)(rows / g->n));
            const size_t lds = (size_t)H * H * sizeof(float);
            const bool sampled = prof_begin(0, st);
            // (H < 128 here: W^T is at most 61 KB of dynamic LDS, below the 64 KB that would need an attribute)
            const bool hubs = g->n_hub > 0;
            DISPATCH_LPR(lpr, hipLaunchKernelGGL((hubs ? k_step_generic<LPR, true> : k_step_generic<LPR, false>), grid, dim3(256), lds, st, g->rowptr, g->col, g->n, (long)rows, H,
                                                 Ycur, zi_cur, zi_nxt, p->odefunc_linear_weight, p->odefunc_linear_bias, beta, gamma,
                                                 dt, p->linear3_weight, p->linear3_bias, p->linearS2_weight, p->linearS2_bias, out,
                                                 g->hubidx, HubP, g->hub_seg_ptr, g->n_seg));
            if (sampled) prof_mark(0, st);
            GN_LAUNCH_CHECK();
            std::swap(zi_cur, zi_nxt);
        } else if (method == 0) {
            if (int e = launch_mlp(g, Y, p->odefunc_linear_weight, p->odefunc_linear_bias, Z, 2 * rows, H, st)) return e;
            StepOut out = {slot >= 0 ? S + (size_t)slot * rows : nullptr, slot >= 0 ? I + (size_t)slot * rows : nullptr,
                           slot >= 0 ? R + (size_t)slot * rows : nullptr, sol_next};
            if (int e = launch_gather(g, 1, rows, H, Y, Z, beta, gamma, 1, dt, nullptr, p, out, hub_scratch, st)) return e;
        } else {
            // torchdiffeq 'rk4' = 3/8 rule
            float* k1 = rk; float* k2 = rk + 3 * slab; float* k3 = rk + 6 * slab; float* k4 = rk + 9 * slab;
            float* yt = rk + 12 * slab;
            const size_t n4 = 3 * slab / 4;
            const int eg = (int)std::min<size_t>((n4 + 255) / 256, 2048);
            StepOut none = {nullptr, nullptr, nullptr, nullptr};
            auto f = [&](float* y, float* kout) -> int {
                if (int e = launch_mlp(g, y, p->odefunc_linear_weight, p->odefunc_linear_bias, Z, 2 * rows, H, st)) return e;
                return launch_gather(g, 0, rows, H, y, Z, beta, gamma, 1, 0.f, kout, nullptr, none, hub_scratch, st);
            };
            const float third = 1.0f / 3.0f;
            if (int e = f(Y, k1)) return e;
            hipLaunchKernelGGL(k_lincomb, dim3(eg), dim3(256), 0, st, yt, Y, dt, k1, third, nullptr, 0.f, nullptr, 0.f, nullptr, 0.f, n4);
            if (int e = f(yt, k2)) return e;
            hipLaunchKernelGGL(k_lincomb, dim3(eg), dim3(256), 0, st, yt, Y, dt, k1, -third, k2, 1.f, nullptr, 0.f, nullptr, 0.f, n4);
            if (int e = f(yt, k3)) return e;
            hipLaunchKernelGGL(k_lincomb, dim3(eg), dim3(256), 0, st, yt, Y, dt, k1, 1.f, k2, -1.f, k3, 1.f, nullptr, 0.f, n4);
            if (int e = f(yt, k4)) return e;
            hipLaunchKernelGGL(k_lincomb, dim3(eg), dim3(256), 0, st, Y, Y, dt, k1, 0.125f, k2, 0.375f, k3, 0.375f, k4, 0.125f, n4);
            GN_LAUNCH_CHECK();
            if (sol_next) GN_HIP(hipMemcpyAsync(sol_next, Y, 3 * slab * sizeof(float), hipMemcpyDeviceToDevice, st));
            if (slot >= 0)
                if (int e = launch_readout(Y, rows, H, p, S + (size_t)slot * rows, I + (size_t)slot * rows,
                                           R + (size_t)slot * rows, st))
                    return e;
        }
    }
    if (sol && n_steps > 0 && !h64) {        // (H = 64: those slabs hold A Z_I(y_k) instead, see above)
        const size_t slab4 = slab / 4;
        hipLaunchKernelGGL(k_fill_bg, dim3((unsigned)std::min<size_t>((slab4 + 255) / 256, 2048)), dim3(256), 0, st, sol, slab4, G);
        GN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int gnode_forward_path(gnode_graph_t g, int64_t rows, int32_t H, int32_t method, int32_t n_steps, int32_t n_out,
                                  int32_t with_sol, int32_t flags, int32_t* plan_host) {
    if (!g || rows <= 0 || rows % g->n) return -1;
    PersPlan pl;
    const int kind = gn_forward_kind(g, rows, H, method, n_steps, n_out, with_sol != 0, flags, &pl);
    if (kind == 2 && plan_host) { plan_host[0] = pl.nt; plan_host[1] = pl.wgs; plan_host[2] = pl.span; plan_host[3] = pl.gpx; plan_host[4] = pl.concurrent; }
    return kind;
}

// diagnostic build (GN_PERS_PROF): per-phase 100 MHz ticks of the last persistent launch on this workspace
extern "C" int gnode_forward_phase_ticks(int64_t rows, int32_t H, int32_t method, const void* workspace, uint64_t* ticks8_host) {
    const PersCtl* ctl = (const PersCtl*)forward_ctl_ptr(const_cast<void*>(workspace), rows, H, method);
    GN_HIP(hipMemcpy(ticks8_host, ctl->prof, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int gnode_forward_status(int64_t rows, int32_t H, int32_t method, const void* workspace, void* stream, int32_t* code_host) {
    GN_CHECK_ARG(workspace && code_host && rows > 0, "gnode_forward_status: null pointer");
    *code_host = 0;
    if (H != 64 && H > 32) return 0;
    unsigned err[2] = {0, 0};
    const PersCtl* ctl = (const PersCtl*)forward_ctl_ptr(const_cast<void*>(workspace), rows, H, method);
    GN_HIP(hipMemcpyAsync(err, ctl->error, sizeof(err), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GN_HIP(hipStreamSynchronize((hipStream_t)stream));
    *code_host = (int32_t)err[0];
    if (err[0]) gnode_set_error("persistent forward: a workgroup gave up waiting for epoch %u of its group", err[1]);
    return 0;
}

extern "C" size_t gnode_forward_keep_bytes(gnode_graph_t g, int64_t rows, int32_t H, int32_t n_steps, int32_t n_out) {
    (void)n_out;                                   // both H = 64 forms (tiled and one-launch) keep the same tables
    if (!g || rows <= 0 || H != 64 || n_steps < 1) return 0;
    return gn_keep_floats((long)rows, n_steps) * sizeof(float);
}

bool gn_sol_carries_ai(const gnode_graph_s* g, long rows, int H, int n_steps, int n_out, int flags) {
    return H == 64 && n_steps >= 1 && gn_forward_kind(g, rows, H, 0, n_steps, n_out, true, flags, nullptr) != 1;
}

extern "C" int gnode_sol_carries_neighbour_sums(gnode_graph_t g, int64_t rows, int32_t H, int32_t n_steps, int32_t n_out, int32_t flags) {
    if (!g || rows <= 0 || rows % g->n) return 0;
    return gn_sol_carries_ai(g, rows, H, n_steps, n_out, flags) ? 1 : 0;
}
