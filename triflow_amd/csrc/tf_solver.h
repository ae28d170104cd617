// tf_solver: the state of one solver of the host runtime (libtriflow_hip, include/triflow_hip.h) and what
// its member functions share.  The runtime is split by concern:
//   tf_solver.h            this file: data, launching, HIP-graph replay of fixed steps, small inline helpers
//   tf_solver_sweeps.cpp   elementary steps: stencil sweeps, J @ v, vector algebra, hooks (tf_solver members)
//   tf_solver_linear.cpp   the banded solver: factor / solve / back-substitution chains, accuracy guard,
//                          rescue on longer chunks (tf_solver members)
//   tf_rt_plan.cpp         models, level plan and memory of a solver (tf_model_*, tf_solver_create ...)
//   tf_rt_io.cpp           inputs and outputs: states, parameters, hooks, F / J, tf_factor / tf_solve / tf_matvec
//   tf_rt_steps.cpp        the time-step drivers Theta / Rosenbrock-Wanner / BDF-2 / step doubling
//   tf_rt_diag.cpp         counters, monitors, stamps, timing, tf_sync
#pragma once
#include "../../include/triflow_hip.h"
#include "tf_args.h"
#include "tf_backend.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace tfrt {

extern thread_local std::string g_last_error;

#define TF_API_BEGIN try {
#define TF_API_END                                               \
    return 0;                                                    \
    }                                                            \
    catch (const std::exception& ex) { g_last_error = ex.what(); return 1; } \
    catch (...) { g_last_error = "unknown error"; return 1; }

inline void require(bool cond, const char* msg) {
    if (!cond) throw std::invalid_argument(msg);
}

inline TfLayout make_layout(int nsys, int N, int P, int periodic) {
    TfLayout L;
    L.nsys = nsys; L.N = N; L.P = P;
    L.mbase = N / P; L.rem = N % P;
    L.M = L.mbase + (L.rem > 0 ? 1 : 0);
    L.Ptot = nsys * P;
    L.periodic = periodic;
    L.plane = (int64_t)L.M * L.Ptot;
    return L;
}

struct DevBuf {
    double* p = nullptr;
    size_t n = 0;
    void alloc(size_t count, int64_t& total) {
        release();
        n = count;
        p = (double*)tfb::dev_alloc(std::max<size_t>(count, 1) * sizeof(double));
        total += (int64_t)(std::max<size_t>(count, 1) * sizeof(double));
    }
    void release() { if (p) tfb::dev_free(p); p = nullptr; n = 0; }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

struct Level {
    TfLayout L;
    int B = 0, MP = 0;
    bool cr = false;       // cyclic-reduction level (tfk_cr_*), else chunk walks
    DevBuf Ablk, rhs, x, Ut, Et, yt, tips_dn, tips_up, Dinv, Unup;
    DevBuf crf, zt;        // cyclic-reduction levels (records per node, see TfLevelArgs)
    unsigned* perm = nullptr;   // ... pivot orders of the last factorisation, [nodes + systems]
    ~Level() { if (perm) tfb::dev_free(perm); }
    Level() = default;
    Level(const Level&) = delete;
    Level& operator=(const Level&) = delete;
    void swap(Level& o) {
        std::swap(L, o.L); std::swap(B, o.B); std::swap(MP, o.MP); std::swap(cr, o.cr); std::swap(perm, o.perm);
        DevBuf* mine[] = {&Ablk, &rhs, &x, &Ut, &Et, &yt, &tips_dn, &tips_up, &Dinv, &Unup, &crf, &zt};
        DevBuf* theirs[] = {&o.Ablk, &o.rhs, &o.x, &o.Ut, &o.Et, &o.yt, &o.tips_dn, &o.tips_up, &o.Dinv, &o.Unup, &o.crf, &o.zt};
        for (int i = 0; i < 12; ++i) mine[i]->swap(*theirs[i]);
    }
    // buffers of one level (l = 0: level 1), as planned in L / B / MP / cr
    void alloc(size_t l, int nsys, bool l1_respike, int64_t& tot) {
        const int64_t pl = L.plane;
        if (cr) {
            // records per node in natural order (TfLevelArgs)
            const size_t nodes = (size_t)L.N * nsys;
            Ablk.alloc(nodes * 4 * B * B, tot);
            rhs.alloc(nodes * 2 * B, tot);
            x.alloc(nodes * B, tot);
            crf.alloc(nodes * 5 * B * B, tot);
            zt.alloc(nodes * B, tot);
            perm = (unsigned*)tfb::dev_alloc((nodes + nsys) * sizeof(unsigned));      // (zero-filled)
            tot += (int64_t)((nodes + nsys) * sizeof(unsigned));
            return;
        }
        // level 1 of a scalar model exchanges rows inside the band: U is 2*MP wide
        const int UW = (l == 0 && B == 1) ? 2 * MP : MP;
        Ut.alloc((size_t)UW * B * B * pl, tot);
        if (l == 0 && l1_respike) Et.alloc(1, tot);
        else Et.alloc((size_t)MP * B * B * pl, tot);
        yt.alloc((size_t)B * pl, tot);
        const size_t tipsz = (size_t)(MP * B + 2 * MP * MP * B * B) * L.Ptot;
        tips_dn.alloc(tipsz, tot);
        tips_up.alloc(tipsz, tot);
        if (l > 0) {
            Ablk.alloc((size_t)3 * B * B * pl, tot);
            Dinv.alloc((size_t)2 * B * B * pl, tot);
            Unup.alloc((size_t)B * B * pl, tot);
            rhs.alloc((size_t)B * pl, tot);
            x.alloc((size_t)B * pl, tot);
        }
    }
    void alloc_top(int b2, int nsys, int64_t& tot) {
        Ablk.alloc((size_t)4 * b2 * b2 * nsys, tot);
        rhs.alloc((size_t)2 * b2 * nsys, tot);
        x.alloc((size_t)b2 * nsys, tot);
    }
};

}  // namespace tfrt
using namespace tfrt;

// enum values of tf_kernels.h (kept in sync by tests/test_abi.py)
enum {
    TF_VEC_SUM = 0, TF_VEC_LIN2 = 1, TF_VEC_THETA_RHS = 2, TF_VEC_MAXABS = 3, TF_VEC_COPY = 4,
    TF_VEC_BDF2_RHS = 5, TF_VEC_ADD = 6, TF_VEC_RESID = 7, TF_VEC_MAXRATIO = 8, TF_VEC_SUM_ERR = 9
};

struct tf_model {
    tf_model_spec spec;
    tfb::Module* module = nullptr;
    ~tf_model() { tfb::module_unload(module); }
};

struct tf_solver {
    tf_model* model = nullptr;
    tf_model_spec spec;
    int64_t N = 0;
    int nsys = 1, periodic = 0, nstate = 3, refine = 0;
    TfLayout L1;
    bool use_cr = false;   // the back end has the cyclic-reduction kernels (tfk_cr_*) for this block size
    // storage of what a level hands to the next one: records per node (below a cyclic-
    // reduction level) or partition-interleaved planes
    unsigned cr_block() const { return top.B <= 2 ? 256u : 64u; }     // TF_CR_BLOCK of tf_entry_hip.h
    // wavefront for each of the 8 nodes of round 1; a level with more chunks than the GPU
    // holds at once (4 such workgroups per CU) takes 4 wavefronts per chunk, twice the chunks in flight
    unsigned cr_factor_block(int64_t chunks) const {
        if (top.B <= 2) return 256u;
        const char* v = getenv("TRIFLOW_CR_FACTOR_BLOCK");
        if (v) return atoi(v) >= 512 ? 512u : 256u;      // (the kernels are written for 4 or 8 wavefronts)
        return chunks > 1024 ? 256u : 512u;
    }
    // the last level is a cyclic-reduction level: it handles the top block itself
    bool fold_top() const { return levels.size() > 1 && levels.back()->cr; }
    bool level_cr(size_t l) const { return l < levels.size() && levels[l]->cr; }
    bool next_aos(size_t l) const { return l + 1 < levels.size() ? levels[l + 1]->cr : levels.back()->cr; }
    tfb::Stream* stream = nullptr;
    int64_t bytes = 0;

    std::vector<std::unique_ptr<DevBuf>> state;     // [nstate] x nvar planes
    DevBuf helpers, parvec, parsca, dx, xcoord;
    DevBuf F, Jv, Wstage, Wsum, Wjv, Wrhs, Wres, Wdel, K[TF_MAX_TERMS];
    DevBuf staging, normbuf;
    DevBuf red;            // reduction scalars
    int* status = nullptr;
    tfb::Mailbox* err_box[4] = {nullptr, nullptr, nullptr, nullptr};   // tf_step_row_queued / tf_read_err
    std::vector<std::unique_ptr<Level>> levels;     // chunk levels; the last one has P == 1
    Level top;             // single-node system per ensemble member
    DevBuf topAinv;
    double factor_c = 0.0;
    bool have_factor = false, have_jac = false;
    // Second factorisation in memory (constant matrices only, made on demand): the step-doubling
    // controller the reference wraps around every scheme (schemes.py:33-66, simulation.py:190-197)
    // alternates c = theta*m*dt and theta*dt, and with one set of factor buffers each change of c
    // would throw away a factorisation that the next-but-one step needs again.  The two sets trade
    // places (swap_slots); which one is current is part of the key of a captured step.
    std::vector<std::unique_ptr<Level>> levels_alt;
    Level top_alt;
    DevBuf topAinv_alt;
    bool alt_allocated = false;
    int slot_id = 0;
    struct SlotMeta {
        double factor_c = 0.0, cf_c = 0.0;
        bool have_factor = false, cf_valid = false, fact_checked = false, fact_needs_refine = false,
             check_now = true, delegated = false;
        uint64_t cf_ver = 0;
        int sweeps_needed = 0;
    } meta_alt;
    bool fact_checked = false, fact_needs_refine = false;   // refine == -1 (auto)
    double last_omega = 0.0, refine_trigger = 1e-11, monitor_omega = 0.0;
    // the backward-error check is a monitor: every factorisation while the matrix is new
    // (first 4, or c changed by > 10 %), then every berr_every-th one
    int berr_every = 8;
    // ... an interval that doubles after every check that reads rounding level (a hundredth of the
    // refinement trigger, no sweep needed), up to berr_max, and falls back to berr_every when a check
    // or the monitor reads more or c moves (a new verdict); fixed when the caller names an interval
    int berr_cur = 8, berr_max = 64;
    bool berr_adaptive = true;
    int64_t n_factor = 0, n_checks = 0, n_replans = 0;
    bool check_now = true;
    // verdicts of the checked factorisations by value of c (within 10 %): a controller that
    // alternates between two step sizes (step doubling: coarse m*dt, fine dt) does not trigger
    // a synchronising check at every switch
    struct Checked { double c; int sweeps; int64_t at; bool replan; };
    std::vector<Checked> checked;
    Checked* checked_like(double c) {
        for (auto& e : checked)
            if (std::fabs(c - e.c) <= 0.1 * std::fabs(e.c)) return &e;
        return nullptr;
    }

    // declarative Dirichlet hook
    int ndir = 0;
    int *dir_var = nullptr, *dir_node = nullptr;
    DevBuf dir_val, dir_val_post;   // values applied before the step (hook at t) / after (t+dt)

    DevBuf stamp_buf;              // diagnostic builds: 64 stamps per solver level (tf_debug_stamps)
    int* csc_map = nullptr;        // tf_set_csc_map: value-table index of every CSC data slot
    int64_t csc_n = 0;
    char* poke_buf = nullptr;      // scratch of tf_poke
    size_t poke_bytes = 0;

    // BDF-2 history U_{n-1}: one per scheme instance that steps on this solver ("owner";
    // owner 0 is the solver's own buffer Uprev, used by callers that own the solver)
    struct BdfHist { DevBuf Uprev; bool have_prev = false; double dt_prev = 0.0; };
    BdfHist bdf0;
    std::map<int64_t, std::unique_ptr<BdfHist>> bdf_owned;

    // A fixed step is a fixed string of launches: captured once per (scheme, slots, dt, ...)
    // into a HIP graph and replayed.  Worth it where a step is launch-bound (small grids: ~24
    // launches of a few microseconds, the host cannot issue them faster than they run), so on
    // by default up to 5e4 nodes (ROS2, N = 200 ... 2000: +11 ... 15 %, Theta: none); TRIFLOW_GRAPHS=0 / 1
    // forces it.  While a graph is captured
    // (TF_CAPTURE) the step function runs as usual, the launches are recorded instead of
    // executed; on a replay it runs "dry" (TF_DRY: the host-side bookkeeping -- factorisation
    // counters, flags -- without the launches) and the graph is launched.
    enum LaunchMode { TF_EAGER = 0, TF_CAPTURE, TF_DRY };
    LaunchMode mode = TF_EAGER;
    bool graphs_on = false;
    struct GraphEntry { tfb::Graph* graph; int64_t used; };
    std::map<std::string, GraphEntry> graphs;
    int64_t graph_clock = 0, graph_replays = 0;
    void drop_graphs() {
        if (!graphs.empty()) { try { tfb::stream_sync(stream); } catch (...) {} }   // (they may still be queued)
        for (auto& kv : graphs) tfb::graph_destroy(kv.second.graph);
        graphs.clear();
    }
    // A key is captured when it comes back, not when it is first seen: the adaptive Rosenbrock
    // schemes call tf_step_row with a new dt in every step, and a capture + instantiation per
    // step costs more than the ~24 eager launches it would replace.
    std::map<std::string, int64_t> seen_once;
    template <class Fn> void run_graphed(const std::string& key, bool graphable, Fn fn) {
        if (!graphs_on || !graphable || timing != 0) { fn(); return; }
        auto it = graphs.find(key);
        if (it == graphs.end()) {
            auto seen = seen_once.find(key);
            if (seen == seen_once.end()) {
                if (seen_once.size() >= 64) seen_once.clear();
                seen_once.emplace(key, ++graph_clock);
                fn();
                return;
            }
            seen_once.erase(seen);
            tfb::capture_begin(stream);
            mode = TF_CAPTURE;
            try { fn(); } catch (...) { mode = TF_EAGER; tfb::capture_abort(stream); throw; }
            mode = TF_EAGER;
            tfb::Graph* g = tfb::capture_end(stream);
            if (graphs.size() >= 8) {                               // least recently used goes
                auto victim = graphs.begin();
                for (auto jt = graphs.begin(); jt != graphs.end(); ++jt)
                    if (jt->second.used < victim->second.used) victim = jt;
                tfb::stream_sync(stream);                           // (it may still be queued)
                tfb::graph_destroy(victim->second.graph);
                graphs.erase(victim);
            }
            it = graphs.emplace(key, GraphEntry{g, 0}).first;
        } else {
            mode = TF_DRY;
            try { fn(); } catch (...) { mode = TF_EAGER; throw; }
            mode = TF_EAGER;
            ++graph_replays;
        }
        it->second.used = ++graph_clock;
        tfb::graph_launch(it->second.graph, stream);
    }
    // A step can be replayed when nothing in it waits for the host: no synchronising
    // backward-error check by the factorisation it makes, and -- constant matrix, the
    // factorisation in memory reused -- none left over by a tf_factor call without a solve
    // (polish() checks the first solve of an unchecked factorisation)
    bool step_graphable(double c) {
        if (const Checked* like = checked_like(c); like && like->replan) return false;   // (the child checks every solve)
        if (reuse_ok(c)) return refine != -1 || fact_checked;
        if (alt_ok(c)) return refine != -1 || meta_alt.fact_checked;
        return !check_due(c);
    }
    // before a step is captured or replayed: the second set of factor buffers, if this step is
    // the one that first needs it (an allocation cannot happen inside a capture)
    void prepare_step(double c) { if (wants_alt(c)) ensure_alt(); }
    // will factor(c) want the synchronising backward-error check?  (then the step is not captured)
    bool check_due(double c) {
        if (refine != -1) return false;
        const Checked* like = checked_like(c);
        return n_factor + 1 <= 4 || !like || n_factor + 1 - like->at >= berr_cur;
    }
    // refinement sweeps the solves of a factorisation with this c will run (part of the launch string)
    int sweeps_for(double c) { const Checked* like = checked_like(c); return like ? like->sweeps : -1; }
    void zero(void* p, size_t nbytes) { if (mode != TF_DRY) tfb::memset0(p, nbytes, stream); }
    void copy(void* dst, const void* src, size_t nbytes) { if (mode != TF_DRY) tfb::d2d(dst, src, nbytes, stream); }

    // timing
    uint64_t timing = 0;     // bit k: time launches of kernel k
    struct Stamp { int kernel; tfb::Event *a, *b; };
    std::vector<Stamp> stamps;
    std::vector<tfb::Event*> event_pool;
    double time_ms[TFK_COUNT] = {0};
    int64_t time_n[TFK_COUNT] = {0};

    ~tf_solver() {
        for (auto& st : stamps) { tfb::event_destroy(st.a); tfb::event_destroy(st.b); }
        for (auto* e : event_pool) tfb::event_destroy(e);
        drop_graphs();
        if (poke_buf) tfb::dev_free(poke_buf);
        if (csc_map) tfb::dev_free(csc_map);
        if (status) tfb::dev_free(status);
        if (sfuse_counter) tfb::dev_free(sfuse_counter);
        for (auto* m : err_box) tfb::mailbox_destroy(m);
        if (dir_var) tfb::dev_free(dir_var);
        if (dir_node) tfb::dev_free(dir_node);
        delete fallback;
        if (tiny_piv) tfb::dev_free(tiny_piv);
        if (owns_stream) tfb::stream_destroy(stream);
    }

    int64_t plane() const { return L1.plane; }
    int64_t vecn() const { return (int64_t)spec.nvar * L1.plane; }
    double* st(int slot) {
        if (slot < 0 || slot >= nstate) throw std::invalid_argument("state slot out of range");
        return state[slot]->p;
    }

    // ------------------------------------------------------------ launching
    tfb::Event* get_event() {
        if (!event_pool.empty()) { auto* e = event_pool.back(); event_pool.pop_back(); return e; }
        return tfb::event_create();
    }
    void launch(int kernel, unsigned gx, unsigned gy, unsigned block, const void* args, size_t sz,
                unsigned lds_bytes = 0) {
        if (mode == TF_DRY) return;
        if ((timing >> kernel) & 1ull) {
            Stamp stp{kernel, get_event(), get_event()};
            tfb::launch_timed(model->module, kernel, gx, gy, block, args, sz, stream, stp.a, stp.b, lds_bytes);
            stamps.push_back(stp);
        } else {
            tfb::launch(model->module, kernel, gx, gy, block, args, sz, stream, lds_bytes);
        }
    }
    void collect_timing() {
        if (stamps.empty()) return;
        tfb::stream_sync(stream);
        for (auto& stp : stamps) {
            time_ms[stp.kernel] += tfb::event_elapsed_ms(stp.a, stp.b);
            time_n[stp.kernel] += 1;
            event_pool.push_back(stp.a);
            event_pool.push_back(stp.b);
        }
        stamps.clear();
    }
    static unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }
    // grid.x of the stencil sweeps (grid.y = segments along the chunks).  Workgroups are dealt
    // round robin over the 8 XCDs in linear order (id = y * gx + x): with gx a multiple of 8
    // the segments above and below a workgroup -- whose first / last rows are its ghost rows --
    // run on the same XCD, and the re-read of those rows is served by that XCD's L2 instead of
    // crossing to the memory side (the extra workgroups find pg >= Ptot and leave).
    unsigned sweep_gx() const {
        const unsigned gx = cdiv(L1.Ptot, spec.sweep_block);
        return gx > 8 ? (gx + 7u) / 8u * 8u : gx;      // (a multiple of the 8 XCDs: DESIGN.md, sweep grid)
    }

    // ------------------------------------------------------ elementary steps
    void vec(int op, double* out, const double* base, int nterms, const double* const* xs, const double* cs, int64_t n = -1, int red_slot = 0, const double* cs2 = nullptr);

    void perm(int mode, const double* src, double* dst, int ncomp);
    void ensure_staging(size_t count);
    // host [ncomp][nsys][N] -> planes
    void upload_planes(const double* host, double* planes, int ncomp);
    void download_planes(const double* planes, double* host, int ncomp);
    void upload_aos(const double* host, double* planes, int ncomp);
    void download_aos(const double* planes, double* host, int ncomp);

    void apply_dirichlet(double* fields, bool post = false);

    void sweep(const double* fields, bool with_j, int nterms = 0, const double* const* kx = nullptr, const double* kc = nullptr, double fscale = 1.0, double* Fout = nullptr);
    // (the fused sweeps of the theta and BDF-2 steps do not store F next to the right-hand side it is
    // part of: tf_get_F after a step is unspecified, include/triflow_hip.h)
    // F, J, the BDF-2 right-hand side and the history update Uprev <- U in one pass
    void sweep_bdf2(const double* fields, bool two_step, double c0, double c1, double* rhs, const double* prev, double* prev_out);
    // F, J and rhs = dt*(F - theta*J@U) + U of the theta scheme in one pass
    void sweep_theta(const double* fields, double dt, double theta, double* rhs);
    void spmv(const double* v, double* y, double scale, bool absval = false);

    // Right-hand side of Rosenbrock stage i >= 1,  dt*F(U + sum_j alpha_ij k_j) + dt*(J @ sum_j gamma_ij k_j):
    // one pass (tfk_sweep_f_stage_rhs) that evaluates F from the window and multiplies J by the other
    // combination of the same k_j loads.  (fuse_stage off, TRIFLOW_FUSE_STAGE=0: the two-kernel form,
    // tfk_sweep_f_stage + tfk_spmv: same operations, same bits.)
    bool fuse_stage = true;
    // Constant matrix (tf_set_constant_jacobian: no Jacobian entry depends on the state or the node).
    // A factorisation made for c stays valid while c, the scalar parameters and dx are what they
    // were (par_ver counts their uploads); the step functions then only solve (factor_step).
    bool jconst = false, cf_valid = false, reused = false;
    double cf_c = 0.0;
    uint64_t par_ver = 0, cf_ver = 0;
    // (the same c up to a few ulp: a driver that lands on t + dt computes its step as target - t, which
    // is dt give or take the rounding of t + dt (schemes.py:58, 217; simulation.py:215-217) -- I - cJ then
    // differs from the factorised matrix by 1e-16 relative, the size of the factorisation's own rounding;
    // the right-hand side is formed with the caller's dt)
    static bool same_c(double a, double b) { return a == b || std::fabs(a - b) <= 1e-15 * std::fabs(b); }
    bool reuse_ok(double c) const { return jconst && cf_valid && have_jac && same_c(cf_c, c) && cf_ver == par_ver; }
    bool alt_ok(double c) const;
    // the factorisation in memory is valid for another c: the next one goes to the other set
    bool wants_alt(double c) const;
    bool two_slots = true;
    void ensure_alt();
    void swap_slots();
    // which set of factor buffers a step with this c will run on, and whether it reuses what is there
    // (both go into the key of a captured step)
    std::string slot_key(double c) const;
    bool l1_respike = false;       // level-1 spike response not stored (tf_args.h, TF_RESPIKE_*)
    int l1_twist = -1;             // -1: by the number of chunks; 0 / 1: TRIFLOW_L1_TWIST (tests, A/B runs)
    // level 1 below a cyclic-reduction level with b <= 6 (TF_FUSE_ASM_OK of tf_entry_hip.h): the walks
    // assemble the separator rows, tfk_l1_asm_mat / _rhs are not launched (TRIFLOW_L1_FUSE_ASM=0: A/B)
    bool l1_fuse_asm = true;
    bool fuse_asm_ok() const;
    // one or two wavefronts per 64 chunks and direction: what the code object was built for
    unsigned l1_factor_block_ = 0;
    unsigned l1_factor_block();
    // N < 2*mp + 1: dense factorisation, one thread per system (tfk_tiny_*)
    bool tiny = false;
    DevBuf tiny_lu;
    int* tiny_piv = nullptr;
    TfTinyArgs tiny_args(const double* rhs1, double* x1);
    bool l1_fuse_backsub = true;   // twisted form: tfk_l1_fwd2_backsub (TRIFLOW_L1_FUSE_BACKSUB=0: two launches)
    void stage_rhs(const double* Uin, int nterms, const double* const* ks, const double* ac, const double* gc, double dt, double* y,
                   const TfBerrArgs* probe = nullptr);     // probe (one term, fused form): rides in the same launch
    // y = cF*F + cA*(J @ sum_t vc_t vx_t)
    void spmv_stage(int nterms, const double* const* vx, const double* vc, const double* Fp, double cF, double cA, double* y);

    // State a step starts from: the reference copies the fields and applies the hook to
    // the copy (schemes.py:144-145, 548-549); without a hook the source slot is read in place.
    // A slot that a step of this solver left with the hook applied at t + dt, and that nothing
    // has written since, already holds what the copy would hold after the hook at the same t: it is
    // read in place as well (`slot_hook`: the Dirichlet values a slot's contents satisfy, compared
    // with the ones about to be applied).  Config 5: 160 MB less copied per step.
    std::vector<std::vector<double>> slot_hook;
    std::vector<double> dir_h, dir_post_h;         // host mirrors of dir_val / dir_val_post
    bool hook_in_place = true;                     // (TRIFLOW_HOOK_IN_PLACE=0: A/B runs, tests)
    void slot_written(int slot) { if (slot >= 0 && (size_t)slot < slot_hook.size()) slot_hook[slot].clear(); }
    void mark_hooked(int slot, bool post = true);
    bool input_is_hooked(int src) const;
    const double* stage_input(int src, double* U);

    // -------------------------------------------------------- banded solver
    // level-1 assemble kernels: a wavefront per separator node on the GPU (tf_entry_hip.h)
    unsigned asm_block() const { return tfb::is_device_build() ? 64u * (unsigned)spec.mp : 64u; }
    Level& next_of(size_t l) { return l + 1 < levels.size() ? *levels[l + 1] : top; }
    // The twisted level-1 kernels that keep a walk's y in LDS (tfk_l1_fwd2_backsub):
    // bytes of dynamic LDS per workgroup, 0 = not for this solver / plan.  Rows = the longer half of
    // the longest chunk (tf_twist_h of tf_kernels.h: chunks too short to split, and wide blocks, stay
    // one-sided); sets a.ylds_rows.
    unsigned l1_twist_lds(TfLevelArgs& a) const;
    TfLevelArgs level_args(size_t l, const double* rhs1, double* x1);
    TfTopArgs top_args();
    // Factorise I - c J.  With `rhs1` the first right-hand side is eliminated in the
    // same walks (level 1: the factor kernel carries it next to the spike columns;
    // reduced levels: one more column of the spike launch; the assemble kernels
    // already build the next level's rhs) and `x1` receives its solution: the
    // first solve of a time step costs only the back-substitutions.
    void factor(double c, const double* rhs1 = nullptr, double* x1 = nullptr);
    // The factorisation of a time step: made, or -- constant matrix, same c and parameters as the
    // one in memory -- reused, and the right-hand side solved like a later stage's
    // (schemes.py:148-149, 557: the reference factorises in every step)
    void factor_step(double c, const double* rhs1, double* x1);
    // The last solve of a time step may leave the new state instead of its solution (TfLevelArgs
    // upd_*: tfk_l1_fwd2_backsub adds base and the earlier stages while it back-substitutes -- no
    // vector kernel, no write and re-read of the last stage).  A step function asks for it right
    // before that solve; it happens when the launch in question is the one that can do it and nobody
    // needs the solution itself afterwards (a checked or refined solve does); otherwise the step
    // function runs the vector kernel as before.
    struct Update { double* out; const double* base; const double* k0; double c0, c1; int n; };
    Update upd_req{};
    bool upd_req_on = false, upd_done = false;
    bool upd_fuse = true;          // (TRIFLOW_FUSE_UPDATE=0: A/B runs, tests)
    void request_update(double* out, const double* base, const double* k0, double c0, double c1, int n);
    bool take_update_done() { const bool d = upd_done; upd_done = false; upd_req_on = false; return d; }
    bool update_allowed() const;
    // skip: that many of the last levels have been back-substituted already (1: the last level
    // inside its forward / factor kernel -- a cyclic-reduction level that folds the top block
    // in; 2: the two last levels by tfk_cr_tail)
    void backsub_chain(const double* rhs1, double* x1, int skip);
    // b = mp * nvar <= 2 with the plan [level 1 | 256-node cyclic-reduction chunks | one chunk]: a solve
    // is two launches (tfk_s_fwd / tfk_s_bwd, TfScalarArgs) instead of six
    bool s_fuse = true;            // (TRIFLOW_S_FUSE=0: A/B runs, tests)
    unsigned* sfuse_counter = nullptr;
    bool l1cr_fuse = true;         // (TRIFLOW_L1CR_FUSE=0: A/B runs, tests)
    bool l1cr_ok() const;
    bool scalar_fused_ok() const;
    TfScalarArgs scalar_args(const double* rhs1, double* x1);
    // The two last levels of a solve go in one launch (tfk_cr_tail) when both are cyclic-reduction
    // levels of 3 <= b <= 6 and the first of them has at most 8 chunks per system
    bool cr_tail = true;
    bool tail_ok() const;
    void solve_once(const double* rhs1, double* x1);
    void refine_sweep(const double* rhs1, double* x1);
    // componentwise (Oettli-Prager) backward error
    //   max_i |b - A x|_i / (|x| + |c J||x| + |b|)_i
    double backward_error(const double* rhs1, const double* x1);
    // The monitor of the steps (Theta, BDF-2, and the stage-0 solve of a Rosenbrock step -- until round 4
    // every 8th of those measured it inside a two-kernel form of stage 1's right-hand side, 69 us against
    // this launch's 4): between two synchronising checks every new factorisation has the same
    // backward error measured at ONE node of every level-1 chunk -- a different one in every step, so
    // that every chunk's elimination is probed in every step and every row once per chunk length (32
    // steps) -- with no host wait: a thread per chunk, ~45 loads each (config 5: 45 MB, ~1 % of a step;
    // the full pass is 576 MB).  The maximum goes to red[4] and is looked at by the next synchronising
    // call.
    // xbase: the state the step started from, when x1 is the new state of a step whose solve leaves
    // U + delta instead of delta (TfBerrArgs::xbase).
    unsigned mon_phase = 0;
    bool sampled_monitor_due() const;
    void monitor_sampled(const double* rhs1, const double* x1, const double* xbase);
    TfBerrArgs probe_args(const double* rhs1, const double* x1, const double* xbase);     // (marks the monitor as used)
    // x = (I - c J)^-1 rhs.  refine > 0: that many refinement sweeps; refine == -1
    // (default): the first solve after every factorisation measures the backward
    // error, and only a factorisation that lost accuracy (block elimination does
    // not pivot across blocks) is polished, this solve and the following ones.
    void solve(const double* rhs1, double* x1);
    // refine > 0: that many sweeps.  refine < 0 (default): on a *checked* solve the
    // backward error is measured; above the trigger, sweeps are added (at most 6) until
    // it is met, and later solves with the same factorisation repeat that number of
    // sweeps.  A factorisation that cannot be polished below 1e-6 is an error: the
    // elimination broke down (no pivoting across blocks), better loud than wrong.
    int sweeps_needed = 0;
    bool unstable = false;

    // ---- re-planning: a child solver of the same model on the same stream with 8 x longer level-1
    // chunks (and so on, down to one chunk per system), made when a factorisation of this plan
    // cannot be refined to 1e-6.  The matrix and the right-hand sides travel through the natural
    // node order (tfk_perm out of this layout, into the child's); the guard path only.
    tf_solver* fallback = nullptr;
    bool owns_stream = true;
    int m1_used = 0, mup_used = 0;
    bool delegated = false;        // the factorisation in memory lives in `fallback`
    bool replan_on = true;         // (TRIFLOW_REPLAN=0: tests of the refusal itself)
    bool can_replan() const { return replan_on && refine == -1 && !tiny && !levels.empty() && levels[0]->L.P > 1; }
    tf_solver* ensure_fallback();
    void transfer_to(tf_solver* dst, const double* src_planes, double* dst_planes, int ncomp);
    // what the child's factorisation in memory belongs to (the child holds one; both sets of factor
    // buffers of a constant-matrix solver may be delegated)
    double fb_c = 0.0;
    uint64_t fb_ver = 0;
    bool fb_valid = false;
    bool fb_touched = false;       // the child ran since this solver last looked at its status
    void delegate_factor(double c);
    void delegate_solve(const double* rhs1, double* x1);
    void polish(const double* rhs1, double* x1);

    // Between two explicit (synchronising) checks every step measures the backward error of its
    // factorisation at one node per chunk (monitor_sampled: no synchronisation); with refine = -2 there
    // are no explicit checks.  The worst value since the last look is read here, at the synchronising calls.
    bool monitored = false;
    // have_flag / have_worst: values that already came back with another download of this call
    void check_status(const int* have_flag = nullptr, const double* have_worst = nullptr);
};
