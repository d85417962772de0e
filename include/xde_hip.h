/*
 * xde_hip.h — C ABI of libxde_hip.so, the MI355X (gfx950) kernels behind
 * paddlexde_amd.functional.odeint / odeint_adjoint.
 *
 * The reference (DrownFish19/PaddleXDE) has no FFI: its hot path is eager Paddle ops in
 * Python.  Each entry point below replaces a group of those ops; the reference lines it
 * replaces are cited per function (paths relative to the reference root).
 *
 * Conventions
 *  - every function returns an int status: XDE_OK, XDE_EBADARG, XDE_EHIP
 *    (message of the last failure on the calling thread: xde_last_error());
 *  - data pointers are DEVICE pointers borrowed from the caller (torch tensors'
 *    data_ptr()); nothing is allocated, freed or retained by the library;
 *  - `stream` is a hipStream_t passed as void* (0 = the null stream); all work is
 *    enqueued asynchronously on it, the only blocking call is xde_ctrl_read();
 *  - state arrays hold `n` elements of `dtype` (XDE_F32 / XDE_F64), contiguous; 16-byte
 *    aligned pointers take the vector path, others a scalar path (same results);
 *  - a state may be a concatenation of up to XDE_MAX_SEG segments (tuple state of the
 *    adjoint pass); norms are taken per segment and combined with max, which is the
 *    reference's `_mixed_norm` / default adjoint norm (functional/odeint_adjoint.py:284-287);
 *  - `xde_ctrl_t` is the device-resident solver state (the reference's `_RungeKuttaState`
 *    scalars, solver/base_adaptive_solver_rk.py:22-24): kernels read dt / select flags from
 *    it, so the host never has to know the step size to enqueue the next step.
 *  - thread-safe for distinct streams + workspaces.
 */
#ifndef XDE_HIP_H
#define XDE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XDE_ABI_VERSION 6

#define XDE_OK 0
#define XDE_EBADARG 1
#define XDE_EHIP 2

#define XDE_F32 0
#define XDE_F64 1

#define XDE_MAX_K 14   /* operands of one combine: Dopri8 has 13 stages + f0 */
#define XDE_MAX_SEG 16 /* segments of a tuple state */
#define XDE_MAX_STAGE 13
#define XDE_MAX_PARTIALS 4096 /* upper bound on blocks of a norm launch */
#define XDE_MAX_PACK 64 /* members one xde_pack_segments launch writes */

/* combine modes (xde_stage_combine) */
#define XDE_COMBINE_RK 0    /* out = y0 + sum_j k_j * (coef_j * dt)            base_adaptive_solver_rk.py:166-168 */
#define XDE_COMBINE_FUSE 1  /* out = (sum_j coef_j * k_j) * dt + y0            xde/base_ode.py:58 (fuse) */
#define XDE_COMBINE_WFUSE 2 /* out = (sum_j coef_j * (k_j * dt + y0)) * scale  base_fixed_solver.py:192-197 */

/* norm kinds */
#define XDE_NORM_RMS 0  /* utils/ode_utils.py:8-9  */
#define XDE_NORM_LINF 1 /* utils/ode_utils.py:4-5  */

/* solver status (xde_ctrl_t.status) -> the reference's AssertionError messages */
#define XDE_STATUS_OK 0
#define XDE_STATUS_DT_UNDERFLOW 1 /* "underflow in dt {}"                base_adaptive_solver_rk.py:200 */
#define XDE_STATUS_NONFINITE 2    /* "non-finite values in state `y`"    base_adaptive_solver_rk.py:201-203 */
#define XDE_STATUS_MAX_STEPS 3    /* "max_num_steps exceeded ({}>={})"   base_adaptive_solver_rk.py:120-122 */

/*
 * Device-resident solver state.  Time-like values are stored as double; when the solver's
 * time dtype is float32 (the reference default, base_adaptive_solver_rk.py:47) every value
 * is float-representable and all controller arithmetic is done in float.
 */
typedef struct xde_ctrl {
  double t0;       /* rk_state.t0: start of the last attempted step */
  double t1;       /* rk_state.t1: current time */
  double dt;       /* signed step the next attempt will use (after step_t clipping) */
  double dt_last;  /* signed step used by the last attempt */
  double t_plan;   /* planned end time of the pending attempt (t1 + dt, or the clipped step_t point) */
  double ratio_prev; /* error ratio of the last ACCEPTED step (opt-in PI controller only) */
  double ratio;    /* error ratio of the last attempt */
  double ratio_seg[XDE_MAX_SEG]; /* per-segment norm of the last attempt */
  double nonfinite; /* count of non-finite elements of y0 seen by the last error-norm */
  int64_t n_steps;  /* attempted steps since xde_ctrl_init */
  int64_t n_accept;
  int64_t n_reject;
  int64_t steps_in_interval; /* attempts since the last emitted output (max_num_steps is per interval) */
  int32_t accept;    /* last attempt accepted; kernels of the NEXT step use it as operand select */
  int32_t sel_used;  /* operand select the last attempt ran with */
  int32_t status;    /* XDE_STATUS_* (sticky) */
  int32_t out_begin; /* rows [out_begin, out_end) of the solution are covered by the last accepted step */
  int32_t out_end;
  int32_t next_out;  /* next output row not yet produced */
  int32_t n_out;     /* number of output rows (len(t_span)) */
  int32_t done;      /* next_out == n_out */
  int32_t next_step_index; /* index into step_t (base_adaptive_solver_rk.py:109-111) */
  int32_t on_step_t; /* the pending dt was clipped to step_t[next_step_index] */
  int64_t seq;       /* controller launches so far (incl. no-op ones after `done`); orders the host mirror */
  uint64_t chk;      /* checksum of the block's other words (host mirror: a copy is accepted only if it holds; see xde_ctrl_wait) */
  int32_t reserved[2];
} xde_ctrl_t;

#define XDE_MIRROR_SLOTS 16 /* host mirror ring: slot[seq % XDE_MIRROR_SLOTS] (a replayed hipGraph may hold several controller launches) */
#define XDE_ETIMEOUT 3

/* Controller parameters (host struct, passed by pointer, copied at call time).
 * The first two words are the binding's statement of the layout it was written against: every entry point that takes
 * this struct returns XDE_EBADARG — before reading anything else, without launching — unless struct_size ==
 * sizeof(xde_ctrl_params_t) and abi_version == XDE_ABI_VERSION.  (A hand-written mirror that predates a field would
 * otherwise make the library read past its end: exactly how a stale `replay` pointer once aborted the process.) */
typedef struct xde_ctrl_params {
  uint32_t struct_size;      /* = sizeof(xde_ctrl_params_t) as the CALLER sees it */
  uint32_t abi_version;      /* = XDE_ABI_VERSION the caller was written against */
  double rtol, atol;         /* already rounded to the time dtype by the caller */
  double min_step, max_step; /* base_adaptive_solver_rk.py:36-37 */
  double safety, ifactor, dfactor; /* :41-43 */
  double order;              /* solver order (exponent = 1/order), ode_utils.py:92 */
  int64_t max_num_steps;     /* :44 */
  int32_t time_dtype;        /* XDE_F32 / XDE_F64: dtype of time-like scalars, :47 */
  int32_t state_dtype;       /* dtype the ratio / stage times are rounded to */
  int32_t direction;         /* +1 forward, -1 reverse time (D5: native signed dt instead of the t->-t flip) */
  int32_t norm_kind;         /* XDE_NORM_* */
  int32_t n_stage;           /* len(tableau.alpha) */
  int32_t n_seg;             /* segments that take part in the norm */
  int32_t n_step_t;          /* len(step_t) or 0 */
  int32_t pi_controller;     /* 0 = reference integral controller; 1 = opt-in PI (never used for parity) */
  double pi_beta;            /* PI only */
  double alpha[XDE_MAX_STAGE]; /* tableau.alpha in double */
  double seg_count[XDE_MAX_SEG]; /* GLOBAL element count of each segment (all ranks) */
  /* Prescribed step sequence ("replay"): NULL, or n_replay pairs {dt_i, accept_i} of doubles on the DEVICE.  Attempt i
   * then runs with the signed step dt_i and is accepted iff accept_i != 0, whatever its error ratio (which is still
   * computed and recorded); after the table is exhausted the controller decides again.  This is how a parity harness
   * holds two implementations of _adaptive_step (base_adaptive_solver_rk.py:183-284) to the same (dt, accept)
   * sequence, so that their per-step y1 / error ratio can be compared free of accept/reject divergence. */
  const double* replay;
  int64_t n_replay;
} xde_ctrl_params_t;

/* Description of one state operand list + segment layout, shared by the norm kernels. */
typedef struct xde_segments {
  uint32_t struct_size; /* = sizeof(xde_segments_t) as the caller sees it; anything else is XDE_EBADARG */
  int32_t n_seg;
  int64_t seg_start[XDE_MAX_SEG]; /* element offset of each segment in the flat state */
  int64_t seg_len[XDE_MAX_SEG];   /* LOCAL element count of each segment */
} xde_segments_t;

const char* xde_last_error(void);
int xde_abi_version(void);
int64_t xde_sizeof_ctrl(void);
/* sizeof(xde_ctrl_params_t): a binding that mirrors the struct by hand (ctypes, cgo) checks its layout against this */
int64_t xde_sizeof_ctrl_params(void);
int64_t xde_sizeof_segments(void);
/* bytes of scratch a norm launch needs (block partials + finalised per-segment sums) */
int64_t xde_workspace_bytes(void);

/*
 * Stage combine — replaces `yi = y0 + sum(k[..., :i+1] * (beta_i * dt), -1)` of
 * AdaptiveRKSolver._runge_kutta_step (solver/base_adaptive_solver_rk.py:166-168), BaseODE.fuse
 * (xde/base_ode.py:58) and the stage/final formulas of rk4_alt_step_func / rk4_step_func /
 * Euler.step / Midpoint.step (solver/base_fixed_solver.py:146-197, fixed_solver/{rk4,euler,midpoint}.py).
 *   out, y0, k[j]: n elements.  coef: nk doubles (rounded to dtype inside, as the reference casts
 *   its tableau, base_adaptive_solver_rk.py:73-79).  dt: read from ctrl->dt when ctrl != NULL,
 *   else dt_host.  If y0_alt/k0_alt != NULL and ctrl != NULL the kernel uses (y0_alt, k0_alt) in
 *   place of (y0, k[0]) when ctrl->accept != 0 (speculative enqueue: the host does not yet know
 *   whether the previous step was accepted).
 *   out2/coef2 (optional, RK mode; FUSE mode: see xde_stage_combine_pre_weighted): a second output
 *   out2 = sum_j k_j * (dt * coef2_j) formed from the SAME loaded operands.  The last stage of an FSAL pair loads exactly the operands the error estimate needs, so it emits the
 *   partial error sum there (`y1_error` minus its last term, base_adaptive_solver_rk.py:180) and the error-norm
 *   kernel reads 4 arrays instead of 8 (xde_error_norm_partial, e_pre).
 *   damping (FUSE/WFUSE): lambda of the delay-equation wrapper's fuse, `y = dy*dt + y0; (dy - lambda*y)*dt + y0`
 *   (xde/base_dde.py:55-58; the reference uses 0.001); 0 selects BaseODE.fuse.
 *   nt_mask: cache-policy hint — bit j set = this launch is the last reader of k[j] (bit 31: of y0), load it
 *   non-temporally so that it does not displace lines that will be re-read from the Infinity Cache.  Results never
 *   depend on it.  (Operands >= 64 MiB are always streamed, whatever the mask.)
 */
int xde_stage_combine(void* out, const void* y0, const void* y0_alt, const void* const* k,
                      const void* k0_alt, const double* coef, int nk, int mode, double scale,
                      double dt_host, const xde_ctrl_t* ctrl, int64_t n, int dtype, void* out2,
                      const double* coef2, double damping, uint32_t nt_mask, void* stream);

/*
 * A stage whose EARLIER operands were already summed by the previous stage's launch (ABI 5): that launch passes
 * `out2 = partial, coef2_j = beta_{i,j}` over the operands it holds anyway (xde_stage_combine's second output: `sum_j k_j (dt coef2_j)`,
 * left to right), and stage i becomes
 *     out = y0 + ((pre + k_0 (coef_0 dt)) + k_1 (coef_1 dt) ...)          k: the stage's 1..4 NEWEST derivatives only
 * — the same left-to-right association as `yi = y0 + sum(k[..., :i+1] * (beta_i * dt), -1)` (base_adaptive_solver_rk.py:166-168), so the
 * result is bit-identical, for 3 arrays read instead of (operands + 1): Dopri5's stage 5 reads y0, the partial sum and k_4 instead of
 * y0, k_0..k_4, and the step moves 30 N instead of 32 N elements through its stage combines.  y0_alt: the speculative pipeline's select
 * of y0 (the newest derivatives are never the select-able f0).
 */
int xde_stage_combine_pre(void* out, const void* y0, const void* y0_alt, const void* pre, const void* const* k, const double* coef,
                          int nk, double dt_host, const xde_ctrl_t* ctrl, int64_t n, int dtype, uint32_t nt_mask, void* stream);

/*
 * The same pre-summing for the FIXED-step solvers' final weighted sum (ABI 6): `y1 = (fuse(k1) w_1 + fuse(k2) w_2 + ... ) * scale`
 * (rk4_alt_step_func, solver/base_fixed_solver.py:190-197; rk4_step_func :160-164).  The launch that forms the LAST stage input holds
 * k_1..k_{s-1} anyway: with mode FUSE, `out2` / `coef2_j = w_j` it also emits `sum_j fuse(k_j, dt, y0) coef2_j` (left to right, no `dt`
 * on the weights); the final launch then reads y0, that partial sum and the newest derivative(s):
 *     out = ((pre + fuse(k_0, dt, y0) coef_0) + ...) * scale               — the full launch's association, bit for bit —
 * RK4: 3 arrays in instead of 5, the emitting launch writes one more: 18 N -> 17 N elements per step.
 */
int xde_stage_combine_pre_weighted(void* out, const void* y0, const void* pre, const void* const* k, const double* coef, int nk,
                                   double scale, double dt_host, const xde_ctrl_t* ctrl, int64_t n, int dtype, double damping,
                                   void* stream);

/*
 * Error-norm partials — replaces `y1_error = sum(k * (dt * c_error), -1)` (base_adaptive_solver_rk.py:180),
 * compute_error_ratio's element-wise chain and the reduction inside the norm (utils/ode_utils.py:80-82,
 * :4-9), and the `isfinite(y0).all()` pass (base_adaptive_solver_rk.py:201).
 * Writes per-block partial sums of (err/tol)^2 (or max |err/tol| for LINF) and the count of
 * non-finite y0 elements to `ws`; xde_norm_finalize / xde_rk_control reduce them in a fixed order.
 *   e_pre (optional): partial error sum written by the last stage's combine (out2); then nk must be 1 and the
 *   estimate is e_pre + k[0] * (dt * c_err[0]) — the same left-to-right association as the unfused sum.
 */
int xde_error_norm_partial(const void* const* k, const void* k0_alt, const double* c_err, int nk,
                           const void* y0, const void* y0_alt, const void* y1, double rtol,
                           double atol, double dt_host, const xde_ctrl_t* ctrl,
                           const xde_segments_t* segs, int norm_kind, int dtype, void* ws,
                           const void* e_pre, void* stream);

/*
 * Error norm AND step controller in ONE launch (single-GPU path with a native norm) — xde_error_norm_partial followed
 * by xde_rk_control, without the second launch: every workgroup publishes its partial write-through and takes a
 * ticket (agent-scope atomic add); the workgroup whose ticket is last acquires, reduces the partials in the same fixed
 * order and runs the controller.  No workgroup waits for another.  rtol/atol/norm_kind are taken from `params`; `ctrl`
 * supplies dt and the operand select and receives the updated state.  The ticket word lives in `ws` and must be zero
 * on entry (it is when `ws` was zero-initialised; the launch leaves it zero).
 */
int xde_error_norm_control(const void* const* k, const void* k0_alt, const double* c_err, int nk, const void* y0,
                           const void* y0_alt, const void* y1, const xde_segments_t* segs, int dtype, void* ws,
                           const void* e_pre, xde_ctrl_t* ctrl, const xde_ctrl_params_t* params,
                           const double* t_span_dev, const double* step_t_dev, void* t_stage_out,
                           xde_ctrl_t* host_mirror, void* stream);

/*
 * Element-wise error ratio for user-supplied norm callables — the tensor `error_estimate / error_tol` of
 * compute_error_ratio (utils/ode_utils.py:80-82) materialised once: out[i] = (sum_j k_j[i]*(dt*c_err_j)) /
 * (atol + rtol*max(|y0[i]|,|y1[i]|)).  The caller applies its own norm to `out` with framework ops and hands the
 * scalar to xde_rk_control through `sums` (norm_kind = XDE_NORM_LINF, n_seg = 1, sums[0] = value).
 *   y0_alt/k0_alt: the speculative pipeline's operand select, as in xde_stage_combine.
 *   nonfinite_out (optional): device double, zeroed by the caller; the number of non-finite y0 elements is ADDED to it
 *   (the `isfinite(y0).all()` pass, base_adaptive_solver_rk.py:201) — pass sums + XDE_MAX_SEG.
 */
int xde_error_ratio(void* out, const void* const* k, const void* k0_alt, const double* c_err, int nk, const void* y0,
                    const void* y0_alt, const void* y1, double rtol, double atol, double dt_host, const xde_ctrl_t* ctrl,
                    int64_t n, int dtype, double* nonfinite_out, void* stream);

/*
 * Scaled norms for the initial-step heuristic — replaces `scale = atol + abs(y0) * rtol`,
 * `norm(a / scale)` of AdaptiveSolver.select_initial_step (solver/base_adaptive_solver.py:50-53,64).
 *   b == NULL: partials of norm(a / scale);  b != NULL: partials of norm((a - b) / scale).
 * Two independent results can be produced by calling it twice with slot 0 and 1.
 */
int xde_scaled_norm_partial(const void* a, const void* b, const void* y0, double rtol, double atol,
                            const xde_segments_t* segs, int norm_kind, int dtype, void* ws, int slot,
                            void* stream);

/*
 * Reduce block partials of slot `slot` to per-segment sums: sums_out[2*XDE_MAX_SEG] doubles laid
 * out as [value(seg 0..), nonfinite(seg 0..)] on the device.  In a multi-GPU run the caller
 * all-reduces sums_out (SUM for RMS/nonfinite, MAX for LINF) before xde_rk_control /
 * xde_norm_result.  Replaces the tail of `_rms_norm` / `_linf_norm` (utils/ode_utils.py:4-9).
 */
int xde_norm_finalize(const void* ws, int slot, double* sums_out, void* stream);

/*
 * Turn finalised sums into the scalar norm max_seg f(sum_seg, count_seg) (sqrt(mean) or max) and
 * write it, rounded to `state_dtype`, to result_out[0] (device double).  Used by the initial-step
 * heuristic (the host reads three such scalars once per integrate()).
 */
int xde_norm_result(const double* sums, const double* seg_count, int n_seg, int norm_kind,
                    int state_dtype, double* result_out, void* stream);

/*
 * Step controller — replaces the accept/reject logic and state update of
 * AdaptiveRKSolver._adaptive_step (solver/base_adaptive_solver_rk.py:200-203,209-215,244-283),
 * optimal_step_size (utils/ode_utils.py:85-97), the stage-time computation of _runge_kutta_step
 * (:159-164) for the NEXT step, the `while next_t > t1` bookkeeping of step() (:116-127) and the
 * max_num_steps assertion.  One thread.
 *   sums: finalised (and, multi-GPU, all-reduced) sums from xde_norm_finalize, or NULL to reduce the
 *         partials in `ws` slot 0 inside this launch (single-GPU fast path);
 *   t_span_dev: n_out doubles (time dtype values) on the device;  step_t_dev: n_step_t doubles or NULL;
 *   t_stage_out: n_stage values of `state_dtype` on the device — times func() is called with in the
 *                next step;
 *   host_mirror: NULL, or a ring of XDE_MIRROR_SLOTS control blocks in pinned, device-mapped host memory
 *                (xde_host_alloc): the controller publishes the updated block to slot[seq % SLOTS] — all its
 *                words, `seq` and `chk` included, with one unordered store instruction — so the host learns
 *                accept/t/dt by polling (xde_ctrl_wait) — no copy command and no event packet on the stream.
 */
int xde_rk_control(xde_ctrl_t* ctrl, const xde_ctrl_params_t* params, const void* ws,
                   const double* sums, const double* t_span_dev, const double* step_t_dev,
                   void* t_stage_out, xde_ctrl_t* host_mirror, void* stream);

/*
 * Initialise the device control block before the first step (replaces the construction of
 * `_RungeKuttaState(y0, f0, t[0], t[0], first_step, ...)`, base_adaptive_solver_rk.py:89-92, and the
 * step_t bookkeeping :95-111).  Also writes the first step's stage times.
 */
int xde_ctrl_init(xde_ctrl_t* ctrl, const xde_ctrl_params_t* params, double t_start,
                  double first_step, int32_t n_out, const double* t_span_dev,
                  const double* step_t_dev, void* t_stage_out, int64_t seq0,
                  const double* first_step_dev, void* stream);

/*
 * Give a running solve a new list of output times — the device side of AdaptiveRKSolver.step(next_t)
 * (solver/base_adaptive_solver_rk.py:116-127), which may be called with any time at or after the start of the last accepted
 * step.  Rows of the new list that the last accepted step already covers are reported immediately in
 * ctrl->out_begin..out_end (the reference's `while next_t > rk_state.t1` does not step for them; xde_dense_eval on the
 * retained operands of that step produces them); the per-interval `max_num_steps` count restarts.  Counts as a controller
 * launch for the host mirror (seq + 1).
 */
int xde_ctrl_retarget(xde_ctrl_t* ctrl, const xde_ctrl_params_t* params, const double* t_span_dev, int32_t n_out,
                      xde_ctrl_t* host_mirror, void* stream);

/*
 * Scalar part of the initial-step heuristic on the device — AdaptiveSolver.select_initial_step
 * (solver/base_adaptive_solver.py:55-72) without reading the three norms back to the host.
 *   phase 0: res_dev = {d0, d1} (xde_norm_result outputs) -> h0 (hs_dev[2]); ctrl->dt = direction*h0, so that the Euler probe
 *            `fuse(f0, h0, y0)` is an xde_stage_combine that reads dt from ctrl; *t_probe_out = t_start + direction*h0 (dtype
 *            probe_dtype = promote(time dtype, state dtype)) is the time func is evaluated at.
 *   phase 1: res_dev = {norm((f1 - f0)/scale)} -> d2 = |./h0| -> h1 -> hs_dev[3] = min(100 h0, h1) in the time dtype;
 *            pass hs_dev + 3 to xde_ctrl_init as first_step_dev.  (first_step_dev != NULL overrides first_step; the
 *            direction's sign is applied there.)
 *   phase 2: phase 0 with the start time read from res_dev[2] instead of t_start — for a launch recorded in a hipGraph and replayed
 *            for one output interval after another (res_dev then holds three doubles).  xde_ctrl_init takes t_start = NaN as
 *            "the start time is t_span_dev[0]" and seq0 < 0 as "keep the block's controller-launch count" for the same use.
 */
int xde_initial_step(int phase, const double* res_dev, double* hs_dev, const xde_ctrl_params_t* params,
                     double t_start, void* t_probe_out, int probe_dtype, xde_ctrl_t* ctrl, void* stream);

/*
 * The same heuristic for SMALL states with its three norm passes inside (ABI 5): one workgroup per phase instead of
 * xde_scaled_norm_partial + xde_norm_finalize + xde_norm_result (x3) + xde_initial_step (x2) + xde_ctrl_init — 12 launches become 2
 * (+ the Euler probe's xde_stage_combine between them).  odeint_adjoint's backward pass starts one solve per output interval
 * (functional/odeint_adjoint.py:134-159), each with this heuristic in front (solver/base_adaptive_solver.py:33-72).
 *   phase 0: a = f0.  d0 = norm(y0/scale), d1 = norm(f0/scale) in ONE pass (scale = atol + |y0| rtol, :50-53), then phase 0 of
 *            xde_initial_step: hs_dev[0..2] = {d0, d1, h0}, ctrl->dt = direction*h0, *t_probe_out = t_start + direction*h0.
 *   phase 1: a = f1, b = f0.  norm((f1 - f0)/scale) (:64), phase 1 of xde_initial_step (hs_dev[3] = the first step, hs_dev[4] =
 *            that norm), and the control block's construction exactly as xde_ctrl_init(first_step_dev = hs_dev + 3) does it.
 *   rtol / atol / norm kind / segment counts come from params; segs as in xde_scaled_norm_partial; hs_dev: 5 doubles.
 *   States above 2^20 elements are refused (one workgroup): use the separate calls.
 *   Re-armable use (a launch recorded in a hipGraph and replayed for one output interval after another): t_start = NaN reads the
 *   start time from t_span_dev[0] (pass it in phase 0 too), and seq0 < 0 keeps the block's own controller-launch count
 *   (ctrl->seq, what the host mirror's slots are numbered with) instead of resetting it; xde_ctrl_init takes seq0 < 0 the same way.
 */
int xde_initial_step_fused(int phase, const void* a, const void* b, const void* y0, const xde_segments_t* segs, int dtype, double* hs_dev,
                           const xde_ctrl_params_t* params, double t_start, void* t_probe_out, int probe_dtype, xde_ctrl_t* ctrl,
                           int32_t n_out, const double* t_span_dev, const double* step_t_dev, void* t_stage_out, int64_t seq0,
                           void* stream);

/*
 * The same heuristic for states ABOVE the one-workgroup kernels' reach (added in round 6; ABI 6 still — nothing existing changed):
 * 12 launches become 4 (+ the Euler probe's xde_stage_combine).
 *   xde_scaled_norm2_partial: partials of d0 = norm(y0/scale) into slot 0 of `ws` and of d1 = norm(f0/scale) into slot 1 in ONE pass
 *            over (y0, f0) — they share scale = atol + |y0| rtol (solver/base_adaptive_solver.py:50-53); record for record what two
 *            xde_scaled_norm_partial launches (a = y0, slot 0; a = f0, slot 1) write.
 *   xde_initial_step_tail, one workgroup per phase, everything that followed a norm pass as launches of its own:
 *     phase 0: xde_norm_finalize + xde_norm_result of slots 0 and 1, then phase 0 of xde_initial_step (hs_dev[0..2] = {d0, d1, h0},
 *              ctrl->dt = direction*h0, *t_probe_out = t_start + direction*h0);
 *     phase 1: the same for slot 0 (filled by xde_scaled_norm_partial(a = f1, b = f0, slot 0)), phase 1 of xde_initial_step
 *              (hs_dev[3] = the first step, hs_dev[4] = that norm) and xde_ctrl_init(first_step_dev = hs_dev + 3).
 *   host_mirror (phase 1, seq0 >= 0; or NULL): the constructed block is also published to slot seq0 % XDE_MIRROR_SLOTS of the host mirror
 *   the way xde_rk_control publishes (checksummed protocol only), so that xde_ctrl_wait(host_mirror, seq0, ...) hands the host the
 *   block — where the first attempt lands, `t_plan` — without a copy command; the caller passes a seq0 no earlier launch published.
 *   Norm kind, segment counts, dtypes come from params (one GPU, a native norm of <= XDE_MAX_SEG segments: a sharded run exchanges the
 *   sums between finalize and result and keeps the separate calls).  t_start = NaN / seq0 < 0: as xde_initial_step_fused.  hs_dev: 5 doubles.
 */
int xde_scaled_norm2_partial(const void* f0, const void* y0, double rtol, double atol, const xde_segments_t* segs, int norm_kind,
                             int dtype, void* ws, void* stream);
int xde_initial_step_tail(int phase, const void* ws, double* hs_dev, const xde_ctrl_params_t* params, double t_start, void* t_probe_out,
                          int probe_dtype, xde_ctrl_t* ctrl, int32_t n_out, const double* t_span_dev, const double* step_t_dev,
                          void* t_stage_out, int64_t seq0, xde_ctrl_t* host_mirror, void* stream);

/* Blocking device->host copy of the control block (hipMemcpyAsync + stream synchronise). */
int xde_ctrl_read(const xde_ctrl_t* ctrl_dev, xde_ctrl_t* host_out, void* stream);

/*
 * Pinned, device-mapped, coherent host memory for the mirror ring (the one allocation the library makes
 * on request; the caller owns it and frees it with xde_host_free).
 */
int xde_host_alloc(int64_t bytes, void** ptr_out);
int xde_host_free(void* ptr);

/*
 * Host side of the mirror: spin (no HIP call) until the controller launch number `seq` has published
 * its block, then copy it to host_out.  The block's words reach the slot in no particular order (ABI 6): a copy
 * is accepted only if its `seq` is the expected one AND its checksum holds — `chk` = the sum, mod 2^64, over
 * every other 8-byte word w_i of the block of splitmix64's finaliser of (w_i + 0x9E3779B97F4A7C15 * (i + 1)) —
 * so a copy taken while words were still landing is never handed back (it is taken again).  XDE_ETIMEOUT after
 * timeout_ms; XDE_EBADARG if the slot was already overwritten by launch seq + XDE_MIRROR_SLOTS.
 * (XDE_CTRL_FLAGS without bit 8 selects rounds 2-4's protocol: seq invalidated, payload, seq — three ordered writes.)
 */
int xde_ctrl_wait(const xde_ctrl_t* host_mirror, int64_t seq, double timeout_ms, xde_ctrl_t* host_out);

/*
 * Dense output — replaces _interp_fit + interp_fit + interp_evaluate
 * (solver/base_adaptive_solver_rk.py:286-292; utils/ode_utils.py:28-49,52-77) for the output rows the
 * last accepted step covers (ctrl->out_begin..out_end); predicated on ctrl->accept.  The quartic's
 * coefficients are never materialised.  out_base is the [n_out, n] solution buffer.
 *   k/mid: operands with non-zero mid-point weight;  f0 = k[0] by construction (k0 candidates),
 *   f1: derivative at t1 (last stage);  y0 candidates selected by ctrl->sel_used.
 *   expect_step: if >= 0 the kernel runs only when ctrl->n_steps == expect_step, i.e. when the
 *   controller launch of this attempt really executed (a speculative attempt enqueued after the last
 *   output is skipped by the controller's `done` guard and must not re-emit rows).
 */
int xde_dense_eval(void* out_base, const void* const* k, const void* k0_alt, const double* mid,
                   int nk, const void* y0, const void* y0_alt, const void* y1, const void* f1,
                   const xde_ctrl_t* ctrl, const double* t_span_dev, int time_dtype, int64_t n,
                   int dtype, int64_t expect_step, void* stream);

/*
 * xde_dense_eval and xde_commit in ONE launch, for the hipGraph pipeline (one node less per replayed attempt; most replays
 * have no output time inside the step and only hand the state over): rows covered by the last accepted step are written as
 * xde_dense_eval writes them, then, if ctrl->accept, y0 <- y1 and f0 <- f1 IN PLACE (y0 and f0 = k[0] are both operands and
 * destinations; every lane reads an element before it overwrites it).
 */
int xde_dense_commit(void* out_base, const void* const* k, const double* mid, int nk, void* y0, const void* y1, void* f0,
                     const void* f1, const xde_ctrl_t* ctrl, const double* t_span_dev, int time_dtype, int64_t n, int dtype,
                     void* stream);

/*
 * Backward of xde_stage_combine for discretise-then-optimise training (the reference back-propagates through
 * its eager ops, example/ode_demo.py:51-53): every input gradient of a combine is a scalar multiple of the
 * output gradient, so one pass reads g once and writes outs[j] = g * factor_j * (dt_dev ? *dt_dev : 1).
 */
int xde_scale_fanout(void* const* outs, const void* g, const double* factors, int nout, const double* dt_dev,
                     int64_t n, int dtype, void* stream);

/*
 * Delay-equation history gather — replaces HistoryIndex.forward (xde/base_dde.py:82-121): the cubic-Hermite
 * spline of the history `his[outer, T, D]` sampled at `his_t[T]` is evaluated at the L learned lags, together with
 * its time derivative (needed by HistoryIndex.backward for d loss / d lags), in ONE pass:
 * CubicHermiteSpline.evaluate / .derivative (interpolation/interpolate_base.py:50-107,
 * interpolation/interpolate.py:100-204), including its conventions (index = clip(bucketize(t)-1, 0, T-1),
 * one-sided differences as node derivatives, the last interval repeated).  val_out, der_out: [outer, L, D].
 */
int xde_hermite_gather(void* val_out, void* der_out, const void* his, const void* his_t, const void* lags,
                       int64_t outer, int T, int D, int L, int dtype, void* stream);

/*
 * The same gather for every history spline HistoryIndex.forward offers (xde/base_dde.py:104-121; ABI 5): method XDE_HISTORY_CUBIC is
 * xde_hermite_gather; XDE_HISTORY_LINEAR — LinearInterpolation (interpolation/interpolate.py:6-99): rows i, i+1 with weights
 * [1-s, s] for the value and [-1, 1] for the derivative; XDE_HISTORY_BEZIER — BezierSpline (:207-298): rows i..i+3 (clamped at T-1)
 * with the Bernstein weights of s = (t - t_i) / (t_{i+3} - t_i).  Conventions as written in the reference's `_make_series`: every row is
 * divided by its own `scale{k+1}` (scale1 shifted right by k, first value repeated), the value is multiplied by scale1_i, the
 * derivative is not.  Linear needs T >= 2, Bezier T >= 4; at most 128 lags per launch.
 */
#define XDE_HISTORY_CUBIC 0
#define XDE_HISTORY_LINEAR 1
#define XDE_HISTORY_BEZIER 2
int xde_history_gather(void* val_out, void* der_out, const void* his, const void* his_t, const void* lags, int64_t outer, int T, int D,
                       int L, int dtype, int method, void* stream);

/*
 * HistoryIndex.backward (xde/base_dde.py:123-127): grad_lags_out[l] = sum over o, d of grad_y[o, l, d] * der[o, l, d] (`grad_y *
 * derivative_lags` summed over every axis but the lag axis; the history itself receives no gradient) — one launch, products formed
 * in `dtype`, accumulated in fp64 in a fixed order (bit-reproducible), result rounded to `dtype`.
 *   ws: xde_lag_grad_workspace_bytes(L) bytes of device memory, zero before the FIRST use (every launch leaves it re-armed).
 *       It holds the arrival counters and partials of the launch IN FLIGHT: one workspace per stream — two launches may share one
 *       only if they are ordered (same stream).  The Python binding keys its workspaces by (device, L, stream).
 */
int64_t xde_lag_grad_workspace_bytes(int L);
int xde_lag_grad(void* grad_lags_out, const void* grad_y, const void* der, int64_t outer, int D, int L, int dtype, void* ws, void* stream);

/*
 * Predicated commit for the hipGraph pipeline (operand addresses are baked into a captured graph, so the
 * host cannot swap pointers): if ctrl->accept then y0_dst <- y1_src and f0_dst <- f1_src — the state update
 * `(t, y, f) <- (t1, y1, f1)` of AdaptiveRKSolver._adaptive_step (solver/base_adaptive_solver_rk.py:258-277).
 * Costs 4*N*s extra bytes on accepted steps; meant for launch-bound (small-state) problems.
 */
int xde_commit(const xde_ctrl_t* ctrl, void* y0_dst, const void* y1_src, void* f0_dst, const void* f1_src,
               int64_t n, int dtype, void* stream);

/*
 * ROCm 7.2 work-around, for callers that capture this library's launches together with framework ops into a hipGraph:
 * MEMSET nodes do not hold their place in a graph there (PyTorch's multi-block reductions capture one; replayed among
 * ordinary stream work such a graph returns the previous replay's result in a large fraction of launches).  This call
 * replaces every memset node of a captured, not yet instantiated hipGraph_t by an equivalent fill-kernel node with the same
 * dependencies and dependents.  *n_replaced = how many there were.
 */
int xde_graph_replace_memsets(void* hip_graph, int* n_replaced);

/*
 * Multi-tensor pack: the members of a tuple state written into ONE flat buffer whose segments start 16-byte aligned, pads zero —
 * the layout every tuple-state entry point above integrates.  Replaces the per-member fill + copy launches of a framework-op
 * pack (the reference's tuple support — utils/misc.py:1-13 flat_to_shape, the commented-out concat of functional/odeint.py — was
 * removed; its intent is one concatenated buffer): odeint_adjoint's augmented dynamics (functional/odeint_adjoint.py:89-124)
 * returns a tuple on every evaluation, and packing it was 8 launches per evaluation for config 3's state.
 *   srcs[s]: lens[s] contiguous elements of `dtype`; starts[s]: element offset in flat_out, a multiple of the 16-byte vector width,
 *   ordered and disjoint, starts[0] == 0; total: elements of flat_out (a multiple of the vector width).  Elements of flat_out that no
 *   segment covers are written as zero.  scales (optional): member s is written as srcs[s] * scales[s] — the adjoint takes its vjp with
 *   the cotangent +adj_y and has the members that the reference computes from -adj_y (:108-114) negated here, exactly, instead of
 *   spending a launch on the negation.
 */
int xde_pack_segments(void* flat_out, const void* const* srcs, const int64_t* starts, const int64_t* lens, const double* scales,
                      int n_seg, int64_t total, int dtype, void* stream);

/*
 * One-shot peer-to-peer exchange of the norm sums between the GPUs of ONE node (new; the reference has no multi-GPU
 * integrator — what it replaces is the all-reduce the batch-sharded solve needs because `_rms_norm` reduces over the
 * whole batch, utils/ode_utils.py:8-9,80-82).  Every rank owns a small mailbox in uncached device memory
 * (xde_p2p_alloc), exports it (xde_p2p_export: a 64-byte IPC handle the host framework ships to the other ranks, e.g.
 * with all_gather_object) and maps every peer's (xde_p2p_import).  xde_p2p_exchange then replaces
 * `all_reduce(sums)` between xde_norm_finalize and xde_rk_control / xde_norm_result: ONE launch stores the caller's
 * 2*XDE_MAX_SEG doubles into every peer's mailbox over xGMI, waits (bounded by spin_limit polls; on expiry the sums are
 * made to stop the solve and xde_p2p_error reports the exchange number) until every peer's vector has arrived, and
 * overwrites sums_dev with their sum (XDE_NORM_LINF: max for the first XDE_MAX_SEG entries) taken in RANK ORDER —
 * bit-identical on every rank.  Ranks must call it the same number of times (lock-step, as the solver's ranks are).
 *   peer_mailboxes: host array of `world` device pointers, peer_mailboxes[rank] == local_mailbox.
 * Failure is group-wide: the rank whose wait ran out also marks every PEER's mailbox, so a peer still waiting stops inside
 * its wait and a peer that had already completed that exchange stops at its next one — no rank runs on alone, none posts again.
 *   xde_p2p_error: *error_out = exchange number of the first failed exchange on this rank (0 = none) — ONE word, as in ABI <= 3.
 *   xde_p2p_error_info (ABI 5; ABI 4 wrote these three words through xde_p2p_error, overrunning a caller built against ABI 3):
 *   info_out[n_out = 3] = {that number, the exchange number a peer reported (0 = none), that peer's rank + 1}.
 *
 * xde_p2p_rk_control (ABI 5) — the sharded attempt's finalize -> exchange -> controller as ONE launch: xde_norm_finalize's fixed-order
 * reduction of the error-norm launch's partial records in `ws` (slot 0), the mailbox exchange of the per-segment sums (as
 * xde_p2p_exchange: same protocol, same counter, same failure behaviour — the two calls may be mixed), and xde_rk_control on the
 * global sums, by one workgroup.  Bit-identical to the three launches; one dependent launch per attempt instead of three, and still
 * capturable into a hipGraph.  norm_kind comes from params.
 */
#define XDE_P2P_MAX_RANKS 16
#define XDE_P2P_HANDLE_BYTES 64
int64_t xde_p2p_mailbox_bytes(void);
int xde_p2p_alloc(void** ptr_out);
int xde_p2p_free(void* ptr);
int xde_p2p_export(void* ptr, void* handle_out);
int xde_p2p_import(const void* handle, void** ptr_out);
int xde_p2p_close(void* ptr);
int xde_p2p_exchange(double* sums_dev, void* local_mailbox, void* const* peer_mailboxes, int world, int rank,
                     int norm_kind, int64_t spin_limit, void* stream);
int xde_p2p_error(const void* local_mailbox, int64_t* error_out, void* stream);
int xde_p2p_error_info(const void* local_mailbox, int64_t* info_out, int n_out, void* stream);
int xde_p2p_rk_control(xde_ctrl_t* ctrl, const xde_ctrl_params_t* params, const void* ws, void* local_mailbox,
                       void* const* peer_mailboxes, int world, int rank, int64_t spin_limit, const double* t_span_dev,
                       const double* step_t_dev, void* t_stage_out, xde_ctrl_t* host_mirror, void* stream);

/*
 * Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the
 * roofline figure).  xde_prof_enable(period): 0 = off; p >= 1 = every p-th launch of each kernel id is
 * launched with a start/stop event pair stamped by the dispatch itself (hipExtLaunchKernelGGL), so the
 * measured time is the kernel's own duration.  xde_prof_collect() synchronises and returns, per kernel
 * id, the number of SAMPLED launches, their summed milliseconds and their summed algorithmic bytes.
 */
#define XDE_KID_COMBINE 0 /* xde_stage_combine, mode RK (adaptive stages); xde_scale_fanout */
#define XDE_KID_ERRNORM 1
#define XDE_KID_CONTROL 2
#define XDE_KID_DENSE 3
#define XDE_KID_SCALEDNORM 4
#define XDE_KID_FINALIZE 5
#define XDE_KID_COMMIT 6
#define XDE_KID_COMBINE_FUSE 7  /* xde_stage_combine, mode FUSE (fixed-step stage inputs) */
#define XDE_KID_COMBINE_WFUSE 8 /* xde_stage_combine, mode WFUSE (fixed-step final combine) */
#define XDE_KID_COUNT 9
int xde_prof_enable(int on);
int xde_prof_collect(int64_t* counts_out, double* ms_out, double* bytes_out);

#ifdef __cplusplus
}
#endif
#endif /* XDE_HIP_H */
