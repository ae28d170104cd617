/* libtriflow_hip -- C ABI of the MI355X-native triflow hot path.
 *
 * Drop-in boundary.  Plain pointers and sizes only; all
 * device memory is owned by the library, host pointers are borrowed for the
 * duration of a call.  Every function returns 0 on success, non-zero on error
 * (text via tf_last_error()); the Python host side raises RuntimeError from it,
 * which is what the reference's Simulation turns into status = 'failed'
 * (triflow/core/simulation.py:259-261).
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   seam #1  compiler plugin  compiler(model) -> (F_function, J_function)
 *            triflow/core/model.py:299-311, callables invoked as
 *            _ufunc(x, *dep_vars, *help_funcs, *pars, periodic)
 *            triflow/core/routines.py:37-45, 82-91
 *            -> tf_model_create, tf_solver_create, tf_set_*, tf_eval, tf_get_F, tf_get_J
 *   seam #2  scheme plugin    scheme(t, fields, dt, pars, hook) -> (t, fields)
 *            triflow/core/schemes.py:101-174 (ROW_general), 523-559 (Theta)
 *            -> tf_step_theta, tf_step_row, tf_step_bdf2 (BDF-2 is new)
 *   seam #3  linear-solver injection  Theta(model, solver=callable(A, b) -> x)
 *            triflow/core/schemes.py:518-521, 557
 *            -> tf_factor, tf_solve
 *
 * Threading: calls on one solver are not thread-safe; different solvers are
 * independent (one HIP stream each).
 */
#ifndef TRIFLOW_HIP_H
#define TRIFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tf_model tf_model;
typedef struct tf_solver tf_solver;

/* Constants of one compiled model (produced by triflow_amd.codegen). */
typedef struct tf_model_spec {
    int32_t nvar;         /* dependent variables                                   */
    int32_t nh;           /* help functions                                        */
    int32_t npar;         /* physical parameters (without 'periodic')              */
    int32_t mp;           /* ghost width: stencil window is 2*mp+1 nodes           */
    int32_t nnz;          /* structurally non-null Jacobian entries per node       */
    int32_t seg;          /* nodes per thread of the stencil sweep                 */
    int32_t sweep_block;  /* workgroup size of the stencil sweep                   */
    int32_t uses_x;       /* the expressions read the coordinate x                 */
    uint32_t parvec_mask; /* bit k set: parameter k is a per-node array            */
} tf_model_spec;

typedef struct tf_solver_opts {
    int32_t m1;           /* chunk length of the first solver level (0 = by problem
                             size: 4 ... 32 nodes)                                  */
    int32_t m_upper;      /* chunk length of the reduced levels (0 = default: 6 for
                             the chunk walks, 16 = the maximum for cyclic reduction) */
    int32_t nstate;       /* resident state slots                   (0 = default 3) */
    int32_t refine;       /* refinement sweeps per solve; 0 = none, -1 = automatic:
                             measure the backward error of the first solve after
                             each factorisation and polish only if it is > 1e-11   */
    int32_t device;       /* HIP device ordinal (-1 = current)                      */
    int32_t berr_every;   /* refine = -1: after the first factorisations (and whenever c
                             moves by > 10 %) the backward error is re-measured on every
                             berr_every-th factorisation (n > 0: that interval, 1 = always;
                             0 = default: 8, doubling up to 64 while the checks read
                             rounding level, back to 8 as soon as one does not); halfway
                             between two checks a Rosenbrock step measures it in passing  */
    int32_t reserved;     /* must be 0 */
} tf_solver_opts;

const char* tf_last_error(void);
int tf_runtime_info(int32_t* is_device_build, int32_t* device_count);
/* current HIP device of the calling thread: code objects are loaded, and solvers
 * created with device = -1 live, on it */
int tf_set_device(int32_t ordinal);

/* code object = gfx950 .hsaco image built from the generated per-model source */
int tf_model_create(const tf_model_spec* spec, const void* code_object, size_t code_size,
                    tf_model** out);
/* Second build of the same code object (lower optimisation level): the kernels whose bit
 * is set in kernel_mask (index = tf_kernel_name) are launched from it.  The compiler plugin
 * uses it for kernels that spill registers to scratch at -O3 (wide models). */
int tf_model_add_alternate(tf_model* model, const void* code_object, size_t code_size,
                           uint64_t kernel_mask);
void tf_model_destroy(tf_model* model);

/* nsys independent systems (ensemble members) of N nodes each share one solver */
int tf_solver_create(tf_model* model, int64_t N, int32_t nsys, int32_t periodic,
                     const tf_solver_opts* opts, tf_solver** out);
void tf_solver_destroy(tf_solver* solver);
int tf_solver_describe(tf_solver* solver, int32_t* nlevels, int32_t* chunks /*[nlevels]*/,
                       int32_t max_levels, int64_t* device_bytes);

/* ---- inputs.  Arrays are [nsys][N] in natural node order. ------------------- */
int tf_set_state(tf_solver*, int32_t slot, int32_t first_var, int32_t nvars, const double* host);
int tf_get_state(tf_solver*, int32_t slot, int32_t first_var, int32_t nvars, double* host);
int tf_set_state_flat(tf_solver*, int32_t slot, const double* uflat /*[nsys][N*nvar]*/);
int tf_get_state_flat(tf_solver*, int32_t slot, double* uflat);
int tf_copy_state(tf_solver*, int32_t src_slot, int32_t dst_slot);
int tf_set_helpers(tf_solver*, int32_t first, int32_t count, const double* host);
int tf_set_param_scalar(tf_solver*, int32_t k, const double* values /*[nsys]*/);
int tf_set_param_vector(tf_solver*, int32_t k, const double* host /*[nsys][N]*/);
int tf_set_dx(tf_solver*, const double* dx /*[nsys]*/);
int tf_set_x(tf_solver*, const double* x /*[nsys][N]*/);
/* declarative Dirichlet hook: U[var][node] = value (node < 0 counts from the end),
 * applied where the reference schemes call hook() */
int tf_set_dirichlet(tf_solver*, int32_t n, const int32_t* var, const int64_t* node,
                     const double* value);

/* time-dependent boundary values: `before` is applied where the reference calls
 * hook(t, ...) at the start of a step, `after` where it calls hook(t + dt, ...)
 * (either may be NULL = unchanged); same entries as the last tf_set_dirichlet */
int tf_set_dirichlet_values(tf_solver*, const double* before, const double* after);
/* n point writes state[var[i]][node[i]] = value[i] (negative nodes count from the end) into a
   resident slot, for every system: what a Python hook like the reference README's
   ``fields.U[0] = 1`` does, without moving the state over PCIe.                         */
int tf_poke(tf_solver*, int32_t slot, int32_t n, const int32_t* var, const int64_t* node,
            const double* value);
/* ... and the values of single nodes, out[i * nsys + system] */
int tf_peek(tf_solver*, int32_t slot, int32_t n, const int32_t* var, const int64_t* node,
            double* out);

/* ---- seam #1: F / J evaluation on the resident state ----------------------- */
int tf_eval(tf_solver*, int32_t slot, int32_t with_j);
/* `reps` back-to-back sweeps between two HIP events on the solver's stream */
int tf_eval_repeat(tf_solver*, int32_t slot, int32_t with_j, int32_t reps, double* total_ms);
/* F of the last tf_eval.  After a step function the F buffer is unspecified: the fused sweeps of
 * the theta and BDF-2 steps fold F into the right-hand side they write and leave the buffer alone,
 * a ROW step leaves dt*F there.  Call tf_eval before tf_get_F. */
int tf_get_F(tf_solver*, double* F /*[nsys][N*nvar], F[node*nvar+eq]*/);
int tf_get_J(tf_solver*, double* Jvals /*[nsys][N][nnz], reference pattern order*/);

/* the Jacobian values in the order of a caller's index list, uploaded once: out[t] = value-table
 * entry map[t] = node * nnz + k of system 0 -- the data array of the csc_matrix that the
 * reference's J function returns (compilers.py:303-331), gathered on the device */
int tf_set_csc_map(tf_solver*, const int32_t* map, int64_t n);
int tf_get_J_mapped(tf_solver*, double* out /*[n]*/);

/* Declares that no Jacobian entry of the model depends on the state or on the node (constant-
 * coefficient linear models such as the reference README's "k * dxxU - c * dxU"; codegen marks
 * the entries, spec["j_uniform"]).  The matrix I - c J of a step then only depends on c, the
 * scalar parameters and dx: the step functions keep the factorisation while those are unchanged
 * and only solve -- where the reference factorises in every step (schemes.py:148-149, 557).
 * Any tf_set_param_scalar / tf_set_param_vector / tf_set_dx invalidates it. */
int tf_set_constant_jacobian(tf_solver*, int32_t on);

/* ---- seam #3: (I - c J) x = b with the Jacobian of the last tf_eval -------- */
int tf_factor(tf_solver*, double c);
int tf_solve(tf_solver*, const double* rhs_flat /*[nsys][N*nvar]*/, double* x_flat);
int tf_matvec(tf_solver*, const double* v_flat, double* y_flat);     /* y = J @ v */

/* ---- seam #2: one time step, state slot src -> slot dst --------------------- */
int tf_step_theta(tf_solver*, int32_t src, int32_t dst, double dt, double theta);
/* alpha, gamma: [s][s] row major; b, b_pred: [s] (b_pred may be NULL);
 * err_out (may be NULL) receives ||U - U_pred||_inf, one scalar over all systems;
 * hook_after: apply the Dirichlet list to the result (fixed-step __call__) */
int tf_step_row(tf_solver*, int32_t src, int32_t dst, double dt, int32_t s,
                const double* alpha, const double* gamma, const double* b,
                const double* b_pred, int32_t hook_after, double* err_out);
/* linearly implicit BDF-2 (new, scheme protocol of schemes.py:523-559); the history U_{n-1}
 * is the solver's own: for a caller that owns the solver and steps one trajectory on it */
int tf_step_bdf2(tf_solver*, int32_t src, int32_t dst, double dt);
int tf_bdf2_reset(tf_solver*);
/* the same step for scheme objects that share a solver: `owner` (non-zero) names the history
 * buffer of one scheme instance, `continuing` != 0 says that slot `src` holds the state that
 * owner's previous step produced; otherwise the step restarts in backward-Euler form */
int tf_step_bdf2_owned(tf_solver*, int32_t src, int32_t dst, double dt, int64_t owner,
                       int32_t continuing);
int tf_bdf2_release(tf_solver*, int64_t owner);      /* frees that history buffer */
/* the same step with the history in a state slot of the caller's: `prev` holds U_{n-1}, or is -1
 * (first step, or the step size changed: backward-Euler form).  Nothing is copied.  With Dirichlet
 * values set, `src` is the next step's history and must hold the *hooked* state (the reference's
 * schemes keep the hooked copy): unless a step of this solver left it so, the boundary values are
 * written into slot `src` in place -- the one step function that modifies its input slot. */
int tf_step_bdf2_from(tf_solver*, int32_t src, int32_t dst, int32_t prev, double dt);
/* tf_step_row with the embedded error estimate left on the device, in reduction slot `err_slot`
 * (1, 2 or 3), until tf_read_err fetches it (blocking; the failure flag is looked at as well).  For a
 * driver of the adaptive schemes (triflow/core/schemes.py:176-238) that queues the first trial of the
 * next call before it reads the estimate of the step it is about to return: the host's decision
 * overlaps with the GPU's next step.  b_pred is required. */
int tf_step_row_queued(tf_solver*, int32_t src, int32_t dst, double dt, int32_t stages,
                       const double* alpha, const double* gamma, const double* b,
                       const double* b_pred, int32_t hook_after, int32_t err_slot);
int tf_read_err(tf_solver*, int32_t err_slot, double* err_out);
/* One trial of the reference's universal step-doubling controller (schemes.py:33-66; it wraps
 * every scheme a Simulation builds, simulation.py:190-197) without a host round trip per
 * sub-step: a coarse step m*dt (src -> coarse), `nfine` fine steps dt (src -> tmp -> dst ...,
 * nfine even: the last one lands in dst; the reference's loop bound is the literal 10) and the
 * norm of their difference, queued back to back; the call returns when the norms are on the
 * host: err_out[system] = max_var ||coarse - dst||_ord / (m*m - 1), ord = 2 or 0 (max).
 * The Dirichlet list of the solver is applied as in the single steps. */
enum { TF_SCHEME_THETA = 0, TF_SCHEME_ROW = 1 };
typedef struct tf_scheme {
    int32_t kind;            /* TF_SCHEME_THETA / TF_SCHEME_ROW */
    int32_t stages;          /* ROW: s */
    double theta;            /* Theta */
    const double* alpha;     /* ROW: [s][s] */
    const double* gamma;     /* ROW: [s][s] */
    const double* b;         /* ROW: [s] */
    int32_t hook_after;      /* ROW: the extra hook call of the fixed-step __call__ */
    int32_t reserved;
} tf_scheme;
int tf_step_doubling(tf_solver*, int32_t src, int32_t dst, int32_t tmp, int32_t coarse, double dt,
                     int32_t m, int32_t nfine, const tf_scheme* scheme, int32_t ord, double* err_out);
/* ||state[a] - state[b]||_ord of every dependent variable, out[nsys][nvar]; ord = 2
 * or 0 (max norm): the error estimate of the step-doubling wrapper
 * (schemes.py:41-44) without bringing the fields to the host */
int tf_diff_norm(tf_solver*, int32_t slot_a, int32_t slot_b, int32_t ord, double* out);

/* componentwise backward error max|b-Ax|/(|x|+|cJ||x|+|b|) measured on the first
 * solve after the last factorisation (refine = -1), and whether it triggered
 * refinement */
int tf_backward_error(tf_solver*, double* omega, int32_t* refined);

/* Between two explicit (synchronising) checks every step measures the backward error of its
 * factorisation's first solve at one node of every level-1 chunk, a different node in every step (no
 * synchronisation; a Rosenbrock step inside the launch of stage 1's right-hand side, a Theta / BDF-2
 * step in a small launch of its own; with refine = -2 there are no explicit checks).  The worst value
 * since the last synchronising call is looked at there (above 1e-11: the next factorisation is
 * checked and refined; above 1e-6: RuntimeError).  This reads it without resetting / raising. */
int tf_monitor_error(tf_solver*, double* worst);
/* counters since the solver was created: factorisations made (a step of a constant-matrix model
 * that finds its factorisation in memory -- one of two, keyed by c -- makes none), backward-error
 * checks that waited for the device, and factorisations that were redone on longer chunks because
 * the first attempt lost accuracy */
int tf_solver_counters(tf_solver*, int64_t* factorisations, int64_t* checks, int64_t* replans);
/* the workgroup size a kernel of the model's code object was built for (tfk_l1_factor*: 128 when
 * the factorisation walks are split over two wavefronts, tf_args.h TF_L1_SPLIT_MODEL) */
int tf_solver_kernel_block(tf_solver*, int32_t kernel, int32_t* block);
int tf_sync(tf_solver*);            /* waits for the stream, reports device-side failures */

/* ---- measurement: kernel begin/end timestamps (HIP events attached to the
 * launch) per kernel; mask bit k selects kernel k, -1 = all, 0 = off ---------- */
int tf_timing_enable(tf_solver*, int64_t mask);
int tf_timing_reset(tf_solver*);
int tf_timing_get(tf_solver*, int32_t kernel, double* total_ms, int64_t* launches);
/* diagnostic kernel builds (-DTF_STAMPS): shader-clock stamps of one workgroup per solver
 * level, out[level][64]; the first call only switches the recording on */
int tf_debug_stamps(tf_solver*, uint64_t* out, int32_t max_levels);
int tf_kernel_count(void);
const char* tf_kernel_name(int32_t kernel);

#ifdef __cplusplus
}
#endif
#endif
