/*
 * gnode.h -- C ABI of libgnode_hip.so: the MI355X (gfx950) GN-ODE integration path.
 *
 * The reference (sissykosm/GN-ODE-SIR) is pure Python and has no FFI of its own;
 * each entry point below names the reference interface it replaces (file:line
 * into the reference tree).  INTEGRATION.md shows the ctypes stubs a maintainer
 * of the reference would add to bind them.
 *
 * Conventions
 *   - every function returns 0 on success, a negative gnode_status otherwise;
 *     gnode_last_error() returns a thread-local message for the last failure.
 *   - "device" pointers are caller-owned HBM allocations (the Python host hands
 *     over torch tensors' data_ptr()); the library never frees or retains them
 *     past the call.  "host" pointers are ordinary memory, read before return.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     All work is enqueued on it.  gnode_rhs_f32, gnode_forward_f32,
 *     gnode_backward_f32 and gnode_l1_loss_f32 allocate nothing, synchronise nothing and keep nothing in
 *     the graph handle: every byte of scratch (including the partial sums of long
 *     "hub" rows) is carved from the caller's workspace, so they can be captured
 *     into a hipGraph on first use and one handle may serve several streams (each
 *     with its own workspace).  Functions that DO synchronise `stream` say so below
 *     (gnode_graph_create, gnode_sir_mc_philox with more than 32 seeds,
 *     gnode_sir_mc_coins, gnode_dmp_f32, gnode_meanfield_f64).
 *   - process-wide state: (1) a per-device "set up once" table (compute-unit count,
 *     dynamic-LDS kernel attributes), written under a lock by the first
 *     gnode_graph_create on a device and read-only afterwards; (2) the opt-in
 *     launch profiler of gnode_profile_enable (off by default; while it is on, use
 *     the library from one thread); (3) the thread-local error string.  Nothing
 *     else: no environment variable is read, no kernel variant is selected at
 *     run time other than by the arguments.
 *   - all floating point is IEEE fp32 (the reference builds its model under
 *     torch.float32, ode_nn_ngraph_sim.py:433); indices are int32.
 *   - row layout is the reference's own: a batch of B samples on one graph of n
 *     nodes is `rows = B*n` rows, row r = b*n + node (ode_nn_ngraph_sim.py:149),
 *     the adjacency is applied block-diagonally (:68-69) WITHOUT ever building
 *     the block-diagonal index.  A multi-graph batch (ode_nn_ngraphs.py:179-196)
 *     is one graph handle holding the concatenated CSR and B = 1.
 */
#ifndef GNODE_H
#define GNODE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GNODE_OK = 0,
    GNODE_ERR_ARG = -1,      /* bad argument (null pointer, shape mismatch, unsupported H) */
    GNODE_ERR_HIP = -2,      /* a HIP runtime call failed */
    GNODE_ERR_WORKSPACE = -3 /* caller workspace too small */
} gnode_status;

typedef struct gnode_graph_s* gnode_graph_t;

/* Parameters of ODEBlock + ODEfunc, device pointers, names = reference state_dict
 * keys (ode_nn_ngraph_sim.py:48,123,126,131). */
typedef struct {
    const float* odefunc_linear_weight; /* [H,H] row-major (out,in) */
    const float* odefunc_linear_bias;   /* [H]   */
    const float* linearS1_weight;       /* [H,1] */
    const float* linearS1_bias;         /* [H]   */
    const float* linear3_weight;        /* [4,H] */
    const float* linear3_bias;          /* [4]   */
    const float* linearS2_weight;       /* [1,4] */
    const float* linearS2_bias;         /* [1]   */
} gnode_params;

const char* gnode_last_error(void);
int gnode_version(void);

/* ---- graph ---------------------------------------------------------------
 * Replaces the per-RHS `scipy.sparse.block_diag` + `torch.LongTensor(idx).to(device)`
 * of ode_nn_ngraph_sim.py:68-71 (ode_nn_ngraphs.py:65-71) and the adjacency
 * built by create_graph, ode_nn.py:413.  rowptr[n+1] / col[nnz] are HOST int32
 * CSR arrays (symmetric, sorted columns, values ignored); the handle owns its
 * device copy.  Synchronous (copies before returning). */
int gnode_graph_create(const int32_t* rowptr_host, const int32_t* col_host, int32_t n, int64_t nnz,
                       gnode_graph_t* out);
int gnode_graph_destroy(gnode_graph_t g);
int gnode_graph_info(gnode_graph_t g, int32_t* n, int64_t* nnz, int32_t* max_degree);

/* ---- RHS -----------------------------------------------------------------
 * ODEfunc.forward(t, x): ode_nn_ngraph_sim.py:58-96 (multi: ode_nn_ngraphs.py:54-83).
 * x, dx: device [4*rows, H], slabs S | I | R | beta-gamma (col 0 beta, col 1 gamma).
 * workspace: device, >= gnode_rhs_workspace_bytes(g, rows, H) (the size depends on the
 * graph: long rows are summed through scratch carved from the workspace). */
size_t gnode_rhs_workspace_bytes(gnode_graph_t g, int64_t rows, int32_t H);
int gnode_rhs_f32(gnode_graph_t g, const float* x, const float* W, const float* b, float* dx,
                  int64_t rows, int32_t H, void* workspace, size_t workspace_bytes, void* stream);

/* ---- forward -------------------------------------------------------------
 * ODEBlock.forward(x): ode_nn_ngraph_sim.py:148-188 (multi: ode_nn_ngraphs.py:124-152)
 * = encoder + odeint(method='euler' | 'rk4') over the grid + read-out + softmax,
 * optionally fused with get_sir_t_nodes_torch (ode_nn.py:249-261).
 *   x          device [rows, 3+H]
 *   dt_host    host  [n_steps] fp32 step sizes t[k+1]-t[k] (grid has n_steps+1 points)
 *   method     0 = euler, 1 = rk4 (torchdiffeq's 3/8 rule)
 *   out_rows_host  host [n_out] ascending grid indices to emit, or NULL = all
 *                  n_steps+1 points (n_out ignored)
 *   S, I, R    device [n_out, rows] each (the reference's [G, rows, 1])
 *   sol        NULL, or device [n_steps+1, 4*rows, H]: the trajectory odeint
 *              returns (needed by the adjoint backward, never by inference).
 *              Slabs S, I, R of every grid point are odeint's.  The 4th slab
 *              (beta, gamma; derivative 0, so odeint repeats sol[0]'s at every grid
 *              point) is odeint's at grid point 0 everywhere; at H = 64 (every form but the
 *              one-workgroup-per-sample launch, gnode_forward_path() == 1) grid points
 *              1 .. n_steps-1 carry A*Z_I(y_k) there instead -- the neighbour sums
 *              the adjoint backward would otherwise gather a second time -- when no
 *              `keep` buffer is given, and are left UNWRITTEN when one is (the sums then
 *              live in `keep`, in the form the backward uses); the last grid point's is
 *              always left unwritten; other H repeat beta, gamma.
 *   keep       NULL, or device buffer of keep_bytes >= gnode_forward_keep_bytes(...):
 *              the KEPT ACTIVATIONS sigmoid(W y_k + b) of the S and I compartments at
 *              every grid point, which the fused H = 64 path has in registers anyway
 *              and gnode_backward_f32 then reads back instead of recomputing (three
 *              of its seven 64x64 products per row and every sigmoid), plus, on the
 *              tiled path, the neighbour sums A*Z_I(y_k) already multiplied by the
 *              sigmoid's derivative.  Opaque layout; ignored when sol is NULL or
 *              gnode_forward_keep_bytes() is 0.  Outputs and the S, I, R slabs of sol
 *              do not depend on whether keep is given.  A trajectory and the keep buffer
 *              of the same call belong together: hand gnode_backward_f32 both, or a
 *              trajectory that was produced WITHOUT keep and NULL.
 *   workspace  device, >= gnode_forward_workspace_bytes(g, rows, H, method)
 *   flags      0, or GNODE_FWD_PER_STEP: never take the persistent one-launch path (below)
 *   sol_info_host  NULL, or host int32 that receives what this call leaves in `sol` / `keep`
 *              (GNODE_SOL_AI: the 4th slabs of sol[1 .. n_steps-1] carry A*Z_I; GNODE_SOL_KEEP: the
 *              keep buffer was filled and those slabs are unwritten; 0: neither): hand it to
 *              gnode_backward_f32, which then refuses a trajectory / keep pair that does not
 *              belong together instead of reading unwritten memory.
 * Mid-size graphs at H = 64 (a few hundred to ~16k rows per launch, no rows longer than the hub
 * threshold, <= 128 steps): the whole integration runs as ONE persistent launch in which every
 * workgroup keeps its rows in registers across all steps and the workgroups of a sample meet at an
 * in-launch barrier once per step.  Its spins are bounded: gnode_forward_status() reports a
 * workgroup that gave up (never observed; it exists so that a hang becomes an error).  Same
 * outputs, bit for bit, as the one-launch-per-step form. */
#define GNODE_FWD_PER_STEP 1
#define GNODE_SOL_AI 1
#define GNODE_SOL_KEEP 2
#define GNODE_SOL_TINY 4   /* produced by the one-workgroup-per-sample forward: no A*Z_I anywhere, keep holds the sigmoids only */
size_t gnode_forward_workspace_bytes(gnode_graph_t g, int64_t rows, int32_t H, int32_t method);
/* Which form gnode_forward_f32 runs for this shape: 0 = one launch per step, 1 = the one-workgroup-per-sample launch (tiny
 * graphs in batches too large for one resident grid), 2 = the persistent launch (H = 64: tiny and mid-size graphs) (then plan_host, if given, receives {16-row tiles per
 * workgroup, workgroups per sample, XCDs per sample, samples side by side per XCD, samples alive at once}), 3 = the persistent
 * launch of the small hidden sizes (H = 8, 16, 32; batches that fit one resident grid); -1 = bad arguments.  n_out: emitted grid points; with_sol: a trajectory is requested (training). */
int gnode_forward_path(gnode_graph_t g, int64_t rows, int32_t H, int32_t method, int32_t n_steps, int32_t n_out,
                       int32_t with_sol, int32_t flags, int32_t* plan_host);
/* Synchronises `stream` and writes 0 to *code_host, or the give-up code of the last gnode_forward_f32 call that ran the
 * persistent path on this workspace (then its outputs are invalid).  Meaningful only after a call for which
 * gnode_forward_path() says 2 or 3 (the other forms never touch the control block).  Not capturable. */
int gnode_forward_status(int64_t rows, int32_t H, int32_t method, const void* workspace, void* stream, int32_t* code_host);
/* The same for the last gnode_backward_f32 call on this workspace: the give-up code of its persistent adjoint sweep, 0 when
 * all went well or the call ran no persistent launch (every call leaves the word defined).  Not capturable. */
int gnode_backward_status(int64_t rows, int32_t H, const void* workspace, void* stream, int32_t* code_host);
/* 1 when gnode_forward_f32 (method 0, no `keep` buffer) on this graph stores A*Z_I(y_k) in the 4th slab of sol[k],
 * 1 <= k <= n_steps-1 (see `sol` below), 0 when the 4th slab repeats beta, gamma at every grid point.  n_out: number
 * of emitted grid points (n_steps+1 when out_rows_host is NULL). */
int gnode_sol_carries_neighbour_sums(gnode_graph_t g, int64_t rows, int32_t H, int32_t n_steps, int32_t n_out, int32_t flags);
/* Size of the optional `keep` buffer of gnode_forward_f32 / gnode_backward_f32 (method 0), or 0 when this H keeps
 * nothing (then pass NULL).  3 * (n_steps + 1) * (rows + 1) * H floats at H = 64 (the tiled and the one-launch form). */
size_t gnode_forward_keep_bytes(gnode_graph_t g, int64_t rows, int32_t H, int32_t n_steps, int32_t n_out);
int gnode_forward_f32(gnode_graph_t g, const float* x, const gnode_params* p, const float* dt_host,
                      int32_t n_steps, int32_t method, const int32_t* out_rows_host, int32_t n_out,
                      float* S, float* I, float* R, float* sol, float* keep, size_t keep_bytes,
                      int64_t rows, int32_t H, void* workspace, size_t workspace_bytes, void* stream,
                      int32_t flags, int32_t* sol_info_host);

/* ---- backward -------------------------------------------------------------
 * The gradient the reference trains with: torchdiffeq's odeint_adjoint under
 * method='euler' (imported at ode_nn_ngraph_sim.py:16, called at :168; semantics in
 * SURVEY Appendix A) followed by autograd through the head and the encoder.
 *   sol          device [n_steps+1, 4*rows, H] saved by gnode_forward_f32 on THIS graph with the
 *                same n_steps / out_rows (its 4th slabs are read as described there)
 *   keep         the buffer the SAME gnode_forward_f32 call filled (then every interval
 *                but the last reads the kept activations; gradients agree with the
 *                recomputing path to fp32 rounding of the summation order), or NULL for a
 *                trajectory that was produced without one
 *   flags        0, or GNODE_FWD_PER_STEP: one launch per interval even where the persistent sweep applies (mid-size
 *                graphs at H = 64 with a keep buffer: intervals n_steps-1 .. 1 run as ONE launch, see the forward)
 *   sol_info     what gnode_forward_f32 reported through sol_info_host for that call (checked against
 *                `keep`: GNODE_ERR_ARG on a mismatch), or -1 = unchecked (the caller vouches for the pairing)
 *   gS, gI, gR   device [n_out, rows] upstream gradients of the outputs
 *   grads        device pointers (same struct as the parameters) that RECEIVE
 *                dL/dparam (overwritten, not accumulated)
 * Euler only (the reference's method).  Deterministic (no float atomics).
 *   workspace  device, >= gnode_backward_workspace_bytes(g, rows, H) */
size_t gnode_backward_workspace_bytes(gnode_graph_t g, int64_t rows, int32_t H);
int gnode_backward_f32(gnode_graph_t g, const float* x, const gnode_params* p, const float* dt_host,
                       int32_t n_steps, const int32_t* out_rows_host, int32_t n_out, const float* sol,
                       const float* keep, size_t keep_bytes,
                       const float* gS, const float* gI, const float* gR, const gnode_params* grads,
                       int64_t rows, int32_t H, void* workspace, size_t workspace_bytes, void* stream,
                       int32_t flags, int32_t sol_info);

/* ---- Monte-Carlo SIR labels ------------------------------------------------
 * sir_torch(G, seed_set, beta, gamma, sims, T): ode_nn.py:30-88.
 *
 * gnode_sir_mc_philox: production mode.  One workgroup per trajectory walking the
 * out-edges of its current frontier (work per step = the frontier's out-degree, not
 * nnz); coins are counter-based Philox4x32-10: the coin of CSR position / node id p is
 * word (p & 3) of the block keyed (p >> 2, step, sim, kind) -- four consecutive items
 * share one block, which a lane computes once -- so that neither the visiting order nor any sharding of
 * [sim_offset, sim_offset+sims) over GPUs can change a count.  counts: device uint32 [3, T, n] (S, I, R), ACCUMULATED into (caller
 * zeroes it); rows t >= 1 add one per trajectory per node, row 0 of S and I is
 * written with the initial state once (reference quirk: assigned, ode_nn.py:55-56).
 * Up to 32 seeds travel as a kernel argument (nothing is synchronised); with more,
 * seeds_host is copied and `stream` is synchronised before the function returns.
 *
 * gnode_sir_mc_coins: parity mode.  Consumes a recorded coin stream exactly as
 * the reference consumes torch.rand (ode_nn.py:65,70): per step first one coin
 * per (infected src -> susceptible dst) row of the directed edge table in table
 * order, then one per infected node in ascending id.  table_src/table_dst:
 * device int32 [n_table] (ode_nn.py:32-38 order).  coins: device fp64.
 * Sequential over sims (one workgroup), writes counts as above and the number
 * of coins consumed to *coins_used_host after synchronising `stream`. */
size_t gnode_sir_workspace_bytes(gnode_graph_t g, int32_t T);   /* for gnode_sir_mc_philox */
size_t gnode_sir_coins_workspace_bytes(void);                    /* for gnode_sir_mc_coins  */
int gnode_sir_mc_philox(gnode_graph_t g, const int32_t* seeds_host, int32_t n_seeds, double beta, double gamma,
                        int64_t sims, int64_t sim_offset, int32_t T, uint64_t rng_seed, uint32_t* counts,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The same model, same coins, same counts by the edge-parallel statement: every workgroup tests EVERY directed edge
 * against the infected set each step (what the reference's `isin` scan does, ode_nn.py:61), O(nnz) per trajectory-step
 * whatever the frontier.  gnode_sir_mc_philox walks the frontier's rows instead and falls back to this scan only for
 * graphs whose frontier lists do not fit the LDS; exported as the cross-check of that kernel and as its measured
 * baseline. */
int gnode_sir_mc_philox_scan(gnode_graph_t g, const int32_t* seeds_host, int32_t n_seeds, double beta, double gamma,
                             int64_t sims, int64_t sim_offset, int32_t T, uint64_t rng_seed, uint32_t* counts,
                             void* workspace, size_t workspace_bytes, void* stream);
/* gnode_sir_mc_philox through the kernel's PROFILING instantiation (a template flag, not a different algorithm): the same
 * counts, plus what the launch did -- stats_host[0] Philox blocks computed, [1] infection coins drawn, [2] recovery coins
 * drawn, [3] CSR entries read (bench.py prices the kernel against the chip's integer rate with them).  Synchronises `stream`. */
int gnode_sir_mc_philox_counted(gnode_graph_t g, const int32_t* seeds_host, int32_t n_seeds, double beta, double gamma,
                                int64_t sims, int64_t sim_offset, int32_t T, uint64_t rng_seed, uint32_t* counts,
                                void* workspace, size_t workspace_bytes, void* stream, uint64_t* stats_host);
int gnode_sir_mc_coins(const int32_t* table_src, const int32_t* table_dst, int64_t n_table, int32_t n,
                       const int32_t* seeds_host, int32_t n_seeds, double beta, double gamma, int64_t sims,
                       int32_t T, const double* coins, int64_t n_coins, uint32_t* counts,
                       int64_t* coins_used_host, void* workspace, size_t workspace_bytes, void* stream);

/* ---- DMP baseline (reference dmp.py:74-170, `DMP_SIR.run`) ----
 * Dynamic message passing marginals of the SIR process on an UNDIRECTED graph
 * (symmetric sparsity pattern; GNODE_ERR_ARG otherwise).  Directed edges are the
 * CSR positions of the handle in row-major order (what `sp.coo_matrix(weight_adj)`
 * yields, dmp.py:67-72).
 *   weights   device fp32 [nnz]   transmission probability of each directed edge
 *                                 (the reference passes A*beta, dmp.py:349)
 *   gamma     device fp32 [n]     recovery probability of each node (dmp.py:349)
 *   out       device fp32 [maxTime, n, 3] = (Ps, Pi, Pr), row 0 = initial state
 *                                 (`DMP_SIR.output()`, dmp.py:159-162)
 * Synchronises `stream` (host arrays are staged through the workspace).
 * Not on the `model='ode_nn'` path: a comparison column of the paper. */
size_t gnode_dmp_workspace_bytes(gnode_graph_t g);
int gnode_dmp_f32(gnode_graph_t g, const float* weights, const float* gamma, const int32_t* seeds_host,
                  int32_t n_seeds, int32_t maxTime, float* out, void* workspace, size_t workspace_bytes,
                  void* stream);

/* ---- mean-field baseline (reference ode_nn.py:214-233) ----
 * `runge_kutta_order4(sir, A, ...)`: dS = -beta (A I) S, dI = beta (A I) S - gamma I,
 * dR = gamma I from S = 1 - seeds, I = seeds, R = 0, float64, sampled at the given
 * times (the reference samples scipy LSODA's solution at int(i/deltaT)*deltaT,
 * ode_nn.py:229-232,235-246).  A = the handle's unweighted adjacency (entries 1,
 * a self-loop counts once).  Adaptive Dormand-Prince 5(4) with steps clipped to the
 * output times; rtol/atol are per component.  t_out_host: host fp64 [n_out],
 * ascending, t_out[0] = 0.  gamma: device fp64 [n].  outI/outS/outR: device fp64
 * [n_out, n] (the reference returns I, S, R in that order).  *steps_host (may be
 * NULL) receives the number of attempted steps.  Synchronises `stream`.
 * Not on the `model='ode_nn'` path: a comparison column of the paper. */
size_t gnode_meanfield_workspace_bytes(gnode_graph_t g);
int gnode_meanfield_f64(gnode_graph_t g, const int32_t* seeds_host, int32_t n_seeds, double beta, const double* gamma,
                        const double* t_out_host, int32_t n_out, double rtol, double atol, double* outI, double* outS,
                        double* outR, int64_t* steps_host, void* workspace, size_t workspace_bytes, void* stream);

/* ---- loss ------------------------------------------------------------------
 * The training loss of ode_nn_ngraph_sim.py:230-234 (multi-graph: ode_nn_ngraphs.py:199-203) and its gradient in one
 * pass: pred = cat(S, I, R)[rows, T, 3] against the labels y, t = 0 excluded,
 *     *loss_sum = sum_{row, t >= t0, c} |pred_c[t, row] - y[row, t, c]|        (float64; L1Loss's mean = sum / count)
 *     sgn[c, t, row] = sign(pred_c[t, row] - y[row, t, c]), 0 for t < t0       (= d loss_sum / d pred; may be NULL)
 *   S, I, R   device [T, rows] fp32 (gnode_forward_f32's outputs)
 *   y         device [rows, T, 3], fp32 or fp64 (y_is_f64); the difference is taken in y's type
 *   loss_sum  device double;  sgn: device [3, T, rows] fp32 or NULL
 *   workspace device, >= gnode_l1_loss_workspace_bytes()
 * Deterministic (fixed-order reduction), asynchronous on `stream`. */
size_t gnode_l1_loss_workspace_bytes(void);
int gnode_l1_loss_f32(const float* S, const float* I, const float* R, const void* y, int32_t y_is_f64, int64_t rows,
                      int32_t T, int32_t t0, double* loss_sum, float* sgn, void* workspace, size_t workspace_bytes,
                      void* stream);
/* The same with the signs already multiplied by `sign_scale` (e.g. 1 / element count: L1Loss's mean): sgn is then the loss
 * gradient with respect to the outputs as it stands, and no scaling launch follows. */
int gnode_l1_loss_scaled_f32(const float* S, const float* I, const float* R, const void* y, int32_t y_is_f64, int64_t rows,
                             int32_t T, int32_t t0, double* loss_sum, float* sgn, float sign_scale, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---- instrumentation -------------------------------------------------------
 * While enabled, every launch of the two step kernels (0: gather + SIR update +
 * read-out, 1: node MLP) is bracketed by HIP events on the launch stream;
 * gnode_profile_read waits for them and returns summed milliseconds and launch
 * counts.  Used by bench.py's roofline leg; off by default (no overhead).
 * Process-wide and not thread-safe: switch it on around a single-threaded region. */
int gnode_profile_enable(int on);
int gnode_profile_read(double* gather_ms, int64_t* gather_launches, double* mlp_ms, int64_t* mlp_launches);
/* kind: 0 = Euler-step kernel, 1 = node-MLP kernel, 2 = backward interval kernel (H = 64), 3 = Monte-Carlo kernel;
 * one launch in 7 is sampled (the first of every 7 of its kind since gnode_profile_enable(1)). */
int gnode_profile_read_kind(int32_t kind, double* ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* GNODE_H */
