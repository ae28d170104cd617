// Host runtime of libtriflow_hip: the C ABI of include/triflow_hip.h.
//
// Owns the device memory, the HIP stream, the per-model code object and the
// orchestration of the kernels in tf_kernels.h:
//   * F / F+J stencil sweep on a resident state      (compilers.py:227-332)
//   * multi-level block-banded factor / solve         (SuperLU call sites schemes.py:149,557)
//   * the time-step drivers Theta / Rosenbrock-Wanner / BDF-2, written so that
//     the vector algebra follows the reference's expressions term by term
//     (schemes.py:142-174, 548-559)
// No compute happens on the host; the only host<->device traffic is what the
// caller asks for through tf_set_* / tf_get_*.
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

namespace {

thread_local std::string g_last_error;

#define TF_API_BEGIN try {
#define TF_API_END                                               \
    return 0;                                                    \
    }                                                            \
    catch (const std::exception& ex) { g_last_error = ex.what(); return 1; } \
    catch (...) { g_last_error = "unknown error"; return 1; }

void require(bool cond, const char* msg) {
    if (!cond) throw std::invalid_argument(msg);
}

TfLayout make_layout(int nsys, int N, int P, int periodic) {
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

}  // namespace

// enum values of tf_kernels.h (kept in sync by tests/test_abi.py)
enum {
    TF_VEC_SUM = 0, TF_VEC_LIN2 = 1, TF_VEC_THETA_RHS = 2, TF_VEC_MAXABS = 3, TF_VEC_COPY = 4,
    TF_VEC_BDF2_RHS = 5, TF_VEC_ADD = 6, TF_VEC_RESID = 7, TF_VEC_MAXRATIO = 8
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
             check_now = true, mon_this = false, delegated = false;
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
    void vec(int op, double* out, const double* base, int nterms, const double* const* xs,
             const double* cs, int64_t n = -1, int red_slot = 0) {
        TfVecArgs a;
        std::memset(&a, 0, sizeof(a));
        a.n = n < 0 ? vecn() : n;
        a.nterms = nterms; a.op = op; a.out = out; a.base = base; a.red = red.p + red_slot;
        for (int t = 0; t < nterms; ++t) { a.x[t] = xs[t]; a.c[t] = cs ? cs[t] : 1.0; }
        unsigned grid = std::min<unsigned>(cdiv(a.n, 256), 2048u);
        launch((op == TF_VEC_MAXABS || op == TF_VEC_MAXRATIO) ? TFK_VEC_MAXABS : TFK_VEC, std::max(grid, 1u), 1, 256, &a, sizeof(a));
    }

    void perm(int mode, const double* src, double* dst, int ncomp) {
        TfPermArgs a;
        a.L = L1; a.src = src; a.dst = dst; a.ncomp = ncomp; a.mode = mode;
        launch(TFK_PERM, cdiv((int64_t)nsys * N, 256), 1, 256, &a, sizeof(a));
    }
    void ensure_staging(size_t count) {
        if (staging.n < count) staging.alloc(count, bytes);
    }
    // host [ncomp][nsys][N] -> planes
    void upload_planes(const double* host, double* planes, int ncomp) {
        size_t cnt = (size_t)ncomp * nsys * N;
        ensure_staging(cnt);
        tfb::h2d(staging.p, host, cnt * sizeof(double), stream);
        perm(0 /*IN_SOA*/, staging.p, planes, ncomp);
    }
    void download_planes(const double* planes, double* host, int ncomp) {
        size_t cnt = (size_t)ncomp * nsys * N;
        ensure_staging(cnt);
        perm(1 /*OUT_SOA*/, planes, staging.p, ncomp);
        tfb::d2h(host, staging.p, cnt * sizeof(double), stream);
    }
    void upload_aos(const double* host, double* planes, int ncomp) {
        size_t cnt = (size_t)ncomp * nsys * N;
        ensure_staging(cnt);
        tfb::h2d(staging.p, host, cnt * sizeof(double), stream);
        perm(3 /*IN_AOS*/, staging.p, planes, ncomp);
    }
    void download_aos(const double* planes, double* host, int ncomp) {
        size_t cnt = (size_t)ncomp * nsys * N;
        ensure_staging(cnt);
        perm(2 /*OUT_AOS*/, planes, staging.p, ncomp);
        tfb::d2h(host, staging.p, cnt * sizeof(double), stream);
    }

    void apply_dirichlet(double* fields, bool post = false) {
        if (ndir == 0) return;
        TfDirichletArgs a;
        a.L = L1; a.fields = fields; a.n = ndir; a.var = dir_var; a.node = dir_node;
        a.value = post ? dir_val_post.p : dir_val.p;
        launch(TFK_DIRICHLET, cdiv((int64_t)ndir * nsys, 64), 1, 64, &a, sizeof(a));
    }

    void sweep(const double* fields, bool with_j, int nterms = 0, const double* const* kx = nullptr,
               const double* kc = nullptr, double fscale = 1.0, double* Fout = nullptr) {
        TfSweepArgs a;
        std::memset(&a, 0, sizeof(a));
        a.nterms = nterms; a.fscale = fscale;
        for (int t = 0; t < nterms; ++t) { a.kx[t] = kx[t]; a.kc[t] = kc[t]; }
        a.L = L1; a.fields = fields; a.helpers = helpers.p; a.parvec = parvec.p; a.parsca = parsca.p;
        a.dx = dx.p; a.xcoord = xcoord.p; a.F = Fout ? Fout : F.p; a.Jv = Jv.p; a.with_j = with_j ? 1 : 0;
        unsigned gx = sweep_gx(), gy = cdiv(L1.M, spec.seg);
        if (nterms > 0 && with_j) throw std::logic_error("stage sweep evaluates F only");
        launch(with_j ? TFK_SWEEP_FJ : (nterms > 0 ? TFK_SWEEP_F_STAGE : TFK_SWEEP_F), gx, gy,
               spec.sweep_block, &a, sizeof(a));
        if (with_j) { have_jac = true; have_factor = false; }
    }
    // (the fused sweeps of the theta and BDF-2 steps do not store F next to the right-hand side it is
    // part of: tf_get_F after a step is unspecified, include/triflow_hip.h)
    // F, J, the BDF-2 right-hand side and the history update Uprev <- U in one pass
    void sweep_bdf2(const double* fields, bool two_step, double c0, double c1, double* rhs,
                    const double* prev, double* prev_out) {
        TfSweepArgs a;
        std::memset(&a, 0, sizeof(a));
        a.fscale = 1.0;
        a.L = L1; a.fields = fields; a.helpers = helpers.p; a.parvec = parvec.p; a.parsca = parsca.p;
        a.dx = dx.p; a.xcoord = xcoord.p; a.F = nullptr; a.Jv = Jv.p; a.with_j = 1;
        a.bdf_rhs = rhs; a.bdf_prev = prev; a.bdf_prev_out = prev_out; a.bdf_c0 = c0; a.bdf_c1 = c1; a.bdf_two_step = two_step ? 1 : 0;
        unsigned gx = sweep_gx(), gy = cdiv(L1.M, spec.seg);
        launch(TFK_SWEEP_FJ_BDF2, gx, gy, spec.sweep_block, &a, sizeof(a));
        have_jac = true; have_factor = false;
    }
    // F, J and rhs = dt*(F - theta*J@U) + U of the theta scheme in one pass
    void sweep_theta(const double* fields, double dt, double theta, double* rhs) {
        TfSweepArgs a;
        std::memset(&a, 0, sizeof(a));
        a.fscale = 1.0;
        a.L = L1; a.fields = fields; a.helpers = helpers.p; a.parvec = parvec.p; a.parsca = parsca.p;
        a.dx = dx.p; a.xcoord = xcoord.p; a.F = nullptr; a.Jv = Jv.p; a.with_j = 1;
        a.theta_rhs = rhs; a.theta = theta; a.theta_dt = dt;
        unsigned gx = sweep_gx(), gy = cdiv(L1.M, spec.seg);
        launch(TFK_SWEEP_FJ_THETA, gx, gy, spec.sweep_block, &a, sizeof(a));
        have_jac = true; have_factor = false;
    }
    void spmv(const double* v, double* y, double scale, bool absval = false) {
        TfSpmvArgs a;
        std::memset(&a, 0, sizeof(a));
        a.L = L1; a.Jv = Jv.p; a.v = v; a.y = y; a.scale = scale; a.absval = absval ? 1 : 0;
        a.parsca = parsca.p; a.dx = dx.p;
        unsigned gx = sweep_gx(), gy = cdiv(L1.M, spec.seg);
        launch(TFK_SPMV, gx, gy, spec.sweep_block, &a, sizeof(a));
    }

    // y = cF*F + cA*(J @ sum_t vc_t vx_t): stage right-hand side of a ROW scheme in one pass
    // (every monitor_every-th factorisation: the magnitudes make the pass 40 % slower)
    bool mon_this = false;         // the factorisation in memory is one the monitor samples (set by factor())
    bool monitor_due(const double* monitor_rhs, int nterms, const double* vc) const {
        return monitor_rhs && nterms == 1 && vc[0] != 0.0 && refine < 0 && !reused && mon_this;
    }
    // will the next factorisation (with this c) be sampled by the monitor?  Halfway between two
    // explicit checks; refine = -2: every one
    bool will_monitor(double c) {
        if (refine == -2) return true;
        const Checked* like = checked_like(c);
        return refine == -1 && !check_due(c) && like && n_factor + 1 - like->at == berr_cur / 2;
    }
    // Right-hand side of Rosenbrock stage i >= 1,  dt*F(U + sum_j alpha_ij k_j) + dt*(J @ sum_j gamma_ij k_j):
    // one pass (tfk_sweep_f_stage_rhs) that evaluates F from the window and multiplies J by the other
    // combination of the same k_j loads.  When the monitor is due, the two-kernel form runs
    // (tfk_sweep_f_stage, tfk_spmv_mon): same operations, same bits.
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
    bool alt_ok(double c) const {
        return jconst && alt_allocated && meta_alt.cf_valid && have_jac && same_c(meta_alt.cf_c, c) && meta_alt.cf_ver == par_ver;
    }
    // the factorisation in memory is valid for another c: the next one goes to the other set
    bool wants_alt(double c) const {
        return jconst && two_slots && !reuse_ok(c) && !alt_ok(c) && cf_valid && have_jac && cf_ver == par_ver;
    }
    bool two_slots = true;
    void ensure_alt() {
        if (alt_allocated) return;
        for (size_t l = 0; l < levels.size(); ++l) {
            std::unique_ptr<Level> lv(new Level());
            lv->L = levels[l]->L; lv->B = levels[l]->B; lv->MP = levels[l]->MP; lv->cr = levels[l]->cr;
            lv->alloc(l, nsys, l1_respike, bytes);
            levels_alt.push_back(std::move(lv));
        }
        top_alt.L = top.L; top_alt.B = top.B; top_alt.MP = top.MP;
        top_alt.alloc_top(top.B, nsys, bytes);
        topAinv_alt.alloc((size_t)top.B * top.B * nsys, bytes);
        alt_allocated = true;
    }
    void swap_slots() {
        levels.swap(levels_alt);
        top.swap(top_alt);
        topAinv.swap(topAinv_alt);
        SlotMeta cur;
        cur.factor_c = factor_c; cur.cf_c = cf_c; cur.have_factor = have_factor; cur.cf_valid = cf_valid;
        cur.fact_checked = fact_checked; cur.fact_needs_refine = fact_needs_refine; cur.check_now = check_now;
        cur.mon_this = mon_this; cur.cf_ver = cf_ver; cur.sweeps_needed = sweeps_needed; cur.delegated = delegated;
        factor_c = meta_alt.factor_c; cf_c = meta_alt.cf_c; have_factor = meta_alt.have_factor; cf_valid = meta_alt.cf_valid;
        fact_checked = meta_alt.fact_checked; fact_needs_refine = meta_alt.fact_needs_refine; check_now = meta_alt.check_now;
        mon_this = meta_alt.mon_this; cf_ver = meta_alt.cf_ver; sweeps_needed = meta_alt.sweeps_needed; delegated = meta_alt.delegated;
        meta_alt = cur;
        slot_id ^= 1;
    }
    // which set of factor buffers a step with this c will run on, and whether it reuses what is there
    // (both go into the key of a captured step)
    std::string slot_key(double c) const {
        const bool reuse = reuse_ok(c) || alt_ok(c);
        const int slot = reuse_ok(c) ? slot_id : ((alt_ok(c) || wants_alt(c)) ? slot_id ^ 1 : slot_id);
        return std::string(reuse ? "|u" : "|f") + (slot ? "1" : "0");
    }
    bool l1_respike = false;       // level-1 spike response not stored (tf_args.h, TF_RESPIKE_*)
    int l1_twist = -1;             // -1: by the number of chunks; 0 / 1: TRIFLOW_L1_TWIST (tests, A/B runs)
    // level 1 below a cyclic-reduction level with b <= 6 (TF_FUSE_ASM_OK of tf_entry_hip.h): the walks
    // assemble the separator rows, tfk_l1_asm_mat / _rhs are not launched (TRIFLOW_L1_FUSE_ASM=0: A/B)
    bool l1_fuse_asm = true;
    bool fuse_asm_ok() const {
        const int b = spec.mp * spec.nvar;
        return l1_fuse_asm && levels.size() > 1 && levels[1]->cr && (2 * b * b + 1) * 64 * 8 <= 40 * 1024;
    }
    // one or two wavefronts per 64 chunks and direction: what the code object was built for
    unsigned l1_factor_block_ = 0;
    unsigned l1_factor_block() {
        if (!l1_factor_block_) l1_factor_block_ = tfb::kernel_block(model->module, TFK_L1_FACTOR) == 128 ? 128 : 64;
        return l1_factor_block_;
    }
    // N < 2*mp + 1: dense factorisation, one thread per system (tfk_tiny_*)
    bool tiny = false;
    DevBuf tiny_lu;
    int* tiny_piv = nullptr;
    TfTinyArgs tiny_args(const double* rhs1, double* x1) {
        TfTinyArgs t;
        std::memset(&t, 0, sizeof(t));
        t.L = L1; t.Jv = Jv.p; t.parsca = parsca.p; t.dx = dx.p; t.c = factor_c;
        t.lu = tiny_lu.p; t.piv = tiny_piv; t.rhs = rhs1; t.x = x1; t.status = status;
        return t;
    }
    bool l1_fuse_backsub = true;   // twisted form: tfk_l1_fwd2_backsub (TRIFLOW_L1_FUSE_BACKSUB=0: two launches)
    void stage_rhs(const double* Uin, int nterms, const double* const* ks, const double* ac,
                   const double* gc, double dt, double* y, const double* monitor_rhs) {
        if (!fuse_stage || monitor_due(monitor_rhs, nterms, gc)) {
            sweep(Uin, false, nterms, ks, ac, 1.0, Wstage.p);
            spmv_stage(nterms, ks, gc, Wstage.p, dt, dt, y, monitor_rhs);
            return;
        }
        TfSweepArgs a;
        std::memset(&a, 0, sizeof(a));
        a.nterms = nterms; a.fscale = 1.0;
        for (int t = 0; t < nterms; ++t) { a.kx[t] = ks[t]; a.kc[t] = ac[t]; a.gc[t] = gc[t]; }
        a.L = L1; a.fields = Uin; a.helpers = helpers.p; a.parvec = parvec.p; a.parsca = parsca.p;
        a.dx = dx.p; a.xcoord = xcoord.p; a.F = Wstage.p; a.Jv = Jv.p;
        a.stage_rhs = y; a.cF = dt; a.cA = dt;
        unsigned gx = sweep_gx(), gy = cdiv(L1.M, TF_STAGE_SEG);
        launch(TFK_SWEEP_F_STAGE_RHS, gx, gy, spec.sweep_block, &a, sizeof(a));
    }
    // monitor_rhs != NULL (first stage product of a Rosenbrock step, one term g*k0): the same pass
    // measures the backward error of the solve that produced k0 from monitor_rhs (red[4])
    void spmv_stage(int nterms, const double* const* vx, const double* vc, const double* Fp,
                    double cF, double cA, double* y, const double* monitor_rhs = nullptr) {
        TfSpmvArgs a;
        std::memset(&a, 0, sizeof(a));
        const bool mon = monitor_due(monitor_rhs, nterms, vc);
        if (mon) {
            a.mon_rhs = monitor_rhs; a.mon_c = factor_c; a.mon_inv_g = 1.0 / vc[0]; a.mon_red = red.p + 4;
            monitored = true;
        }
        a.L = L1; a.Jv = Jv.p; a.v = nullptr; a.y = y; a.scale = 1.0;
        a.parsca = parsca.p; a.dx = dx.p;
        a.nterms = nterms;
        for (int t = 0; t < nterms; ++t) { a.vx[t] = vx[t]; a.vc[t] = vc[t]; }
        a.addF = Fp; a.cF = cF; a.cA = cA;
        unsigned gx = sweep_gx(), gy = cdiv(L1.M, spec.seg);
        launch(mon ? TFK_SPMV_MON : TFK_SPMV, gx, gy, spec.sweep_block, &a, sizeof(a));
    }

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
    void mark_hooked(int slot, bool post = true) {
        if (slot < 0) return;
        if ((size_t)slot >= slot_hook.size()) slot_hook.resize((size_t)slot + 1);
        slot_hook[slot] = post ? dir_post_h : dir_h;
    }
    bool input_is_hooked(int src) const {
        return hook_in_place && ndir > 0 && src >= 0 && (size_t)src < slot_hook.size() &&
               !slot_hook[src].empty() && slot_hook[src] == dir_h;
    }
    const double* stage_input(int src, double* U) {
        if (ndir == 0 || input_is_hooked(src)) return st(src);
        copy(U, st(src), (size_t)vecn() * sizeof(double));
        apply_dirichlet(U);
        return U;
    }

    // -------------------------------------------------------- banded solver
    // level-1 assemble kernels: a wavefront per separator node on the GPU (tf_entry_hip.h)
    unsigned asm_block() const { return tfb::is_device_build() ? 64u * (unsigned)spec.mp : 64u; }
    Level& next_of(size_t l) { return l + 1 < levels.size() ? *levels[l + 1] : top; }
    // The twisted level-1 kernels that keep a walk's y in LDS (tfk_l1_fwd2_backsub):
    // bytes of dynamic LDS per workgroup, 0 = not for this solver / plan.  Rows = the longer half of
    // the longest chunk (tf_twist_h of tf_kernels.h: chunks too short to split, and wide blocks, stay
    // one-sided); sets a.ylds_rows.
    unsigned l1_twist_lds(TfLevelArgs& a) const {
        if (!(l1_respike && a.twist && l1_fuse_backsub && tfb::is_device_build())) return 0;
        auto half = [&](int mI) { return (spec.mp * spec.nvar <= 6 && mI >= 4 * spec.mp) ? (mI + 1) / 2 : mI; };
        const int mI_max = a.L.M - spec.mp;
        a.ylds_rows = std::max(half(mI_max), a.L.rem > 0 ? half(mI_max - 1) : 0);
        const size_t lds = (size_t)2 * a.ylds_rows * spec.nvar * 64 * sizeof(double);
        // (up to half of a CU's 160 KB: two workgroups = four wavefronts, one per SIMD; the stiff model's
        // 80 KB just fit -- config 5 687 -> 698 steps/s, profiles/r03_ab_runs.txt r3n)
        return lds <= 80u * 1024u ? (unsigned)lds : 0u;
    }
    TfLevelArgs level_args(size_t l, const double* rhs1, double* x1) {
        Level& lv = *levels[l];
        Level& nx = next_of(l);
        TfLevelArgs a;
        std::memset(&a, 0, sizeof(a));
        a.L = lv.L; a.Jv = Jv.p; a.parsca = parsca.p; a.dx = dx.p; a.c = factor_c; a.Ablk = lv.Ablk.p;
        a.rhs = l == 0 ? rhs1 : lv.rhs.p;
        a.x = l == 0 ? x1 : lv.x.p;
        a.Ut = lv.Ut.p; a.Et = lv.Et.p; a.yt = lv.yt.p; a.Dinv = lv.Dinv.p; a.Unup = lv.Unup.p;
        a.tips_dn = lv.tips_dn.p; a.tips_up = lv.tips_up.p;
        a.Lnext = nx.L; a.Anext = nx.Ablk.p; a.rhsnext = nx.rhs.p; a.xnext = nx.x.p;
        a.status = status;
        a.next_aos = next_aos(l) ? 1 : 0; a.crf = lv.crf.p; a.zt = lv.zt.p; a.perm = lv.perm;
        a.fold_top = fold_top() && l + 1 == levels.size() ? 1 : 0;
        a.respike = l == 0 && l1_respike ? 1 : 0;
        a.fuse_asm = l == 0 && fuse_asm_ok() ? 1 : 0;
        // twisted while one walk direction leaves SIMDs idle -- and beyond that wherever the two
        // launches become one with y in LDS (tfk_l1_fwd2_backsub: 8 members per GPU +2.6 %,
        // profiles/r03_ab_runs.txt; the stiff model's y block does not fit)
        a.twist = a.respike && (l1_twist < 0 ? lv.L.Ptot <= TF_TWIST_MAX_CHUNKS : l1_twist > 0) ? 1 : 0;
        if (a.respike && l1_twist < 0 && !a.twist) { a.twist = 1; if (!l1_twist_lds(a)) a.twist = 0; }
        a.topAinv = topAinv.p; a.topx = top.x.p;
        a.stamps = stamp_buf.n ? (unsigned long long*)stamp_buf.p + 64 * l : nullptr;
        return a;
    }
    TfTopArgs top_args() {
        TfTopArgs t;
        t.nsys = nsys; t.A = top.Ablk.p; t.rhs = top.rhs.p; t.Ainv = topAinv.p; t.x = top.x.p; t.status = status;
        t.aos = levels.back()->cr ? 1 : 0;
        return t;
    }
    // Factorise I - c J.  With `rhs1` the first right-hand side is eliminated in the
    // same walks (level 1: the factor kernel carries it next to the spike columns;
    // reduced levels: one more column of the spike launch; the assemble kernels
    // already build the next level's rhs) and `x1` receives its solution: the
    // first solve of a time step costs only the back-substitutions.
    void factor(double c, const double* rhs1 = nullptr, double* x1 = nullptr) {
        if (!have_jac) throw std::runtime_error("tf_factor: no Jacobian evaluated yet (call tf_eval with_j=1)");
        factor_c = c;
        delegated = false;
        if (const Checked* like0 = checked_like(c); like0 && like0->replan && can_replan()) {
            // this plan is known to break down for such a c: straight to the longer chunks
            ++n_factor;
            have_factor = true; cf_valid = jconst; cf_c = c; cf_ver = par_ver;
            check_now = false; mon_this = false; fact_checked = true; sweeps_needed = 0;
            delegate_factor(c);
            if (rhs1) delegate_solve(rhs1, x1);
            return;
        }
        const bool fused = rhs1 != nullptr && !tiny;
        if (tiny) {
            TfTinyArgs t = tiny_args(nullptr, nullptr);
            launch(TFK_TINY_FACTOR, cdiv(nsys, 64), 1, 64, &t, sizeof(t));
        }
        for (size_t l = 0; l < levels.size() && !tiny; ++l) {
            TfLevelArgs a = level_args(l, rhs1, x1);
            if (!fused) a.rhs = nullptr;
            unsigned gx = cdiv(a.L.Ptot, 64);
            if (l == 0) launch(fused ? TFK_L1_FACTOR_RHS : TFK_L1_FACTOR, gx, 2, l1_factor_block(), &a, sizeof(a));
            else if (levels[l]->cr) {
                // one wavefront per chunk; leaves the next level's rows (and rhs) behind
                a.cr_rhs = fused ? 1 : 0;
                launch(TFK_CR_FACTOR, (unsigned)a.L.Ptot, 1, cr_factor_block(a.L.Ptot), &a, sizeof(a));
                continue;
            } else {
                const int G = tfb::coop_group(levels[l]->B);
                const unsigned gc = cdiv((int64_t)a.L.Ptot * G, 64);
                const int ncols = levels[l]->B + (fused ? 1 : 0);
                a.lu_cols = G > 1 ? ncols : 0;        // the cooperative LU walks the columns itself
                launch(TFK_BT_LU, gc, 2, 64, &a, sizeof(a));
                if (G == 1) launch(TFK_BT_SPIKE, gc, 2 * (unsigned)ncols, 64, &a, sizeof(a));
            }
            if (l == 0) { if (!a.fuse_asm) launch(TFK_L1_ASM_MAT, gx, 1, asm_block(), &a, sizeof(a)); }
            else launch(TFK_BT_ASM_MAT, cdiv((int64_t)a.L.Ptot * tfb::coop_group(levels[l]->B), 64), 1, 64, &a, sizeof(a));
        }
        if (!fold_top() && !tiny) { TfTopArgs t = top_args(); launch(TFK_TOP_FACTOR, cdiv((int64_t)nsys * (tfb::coop_group(top.B) == 8 ? 8 : 1), 64), 1, 64, &t, sizeof(t)); }
        have_factor = true;
        ++n_factor;
        cf_valid = jconst; cf_c = c; cf_ver = par_ver;
        const Checked* like = checked_like(c);
        check_now = refine == -1 && (n_factor <= 4 || !like || n_factor - like->at >= berr_cur);
        mon_this = refine == -2 || (refine == -1 && !check_now && like && n_factor - like->at == berr_cur / 2);
        if (check_now) { fact_checked = false; fact_needs_refine = false; sweeps_needed = 0; }
        else if (like) { sweeps_needed = like->sweeps; fact_needs_refine = sweeps_needed > 0; }
        // (between checks the verdict of the last checked factorisation with such a c stands)
        if (rhs1 == nullptr) return;
        if (!fused) { solve(rhs1, x1); return; }
        if (!fold_top()) { TfTopArgs t = top_args(); launch(TFK_TOP_SOLVE, cdiv(nsys, 64), 1, 64, &t, sizeof(t)); }
        backsub_chain(rhs1, x1, fold_top() ? 1 : 0);
        polish(rhs1, x1);
    }
    // The factorisation of a time step: made, or -- constant matrix, same c and parameters as the
    // one in memory -- reused, and the right-hand side solved like a later stage's
    // (schemes.py:148-149, 557: the reference factorises in every step)
    void factor_step(double c, const double* rhs1, double* x1) {
        reused = reuse_ok(c);
        if (!reused && alt_ok(c)) { swap_slots(); reused = true; }
        else if (wants_alt(c)) { ensure_alt(); swap_slots(); }        // (allocated before any capture: prepare_step)
        if (!reused) { factor(c, rhs1, x1); return; }
        have_factor = true;                          // (the sweep of this step reset it)
        solve(rhs1, x1);
    }
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
    void request_update(double* out, const double* base, const double* k0, double c0, double c1, int n) {
        upd_req = Update{out, base, k0, c0, c1, n};
        upd_req_on = upd_fuse;
        upd_done = false;
    }
    bool take_update_done() { const bool d = upd_done; upd_done = false; upd_req_on = false; return d; }
    bool update_allowed() const {
        return !delegated && !tiny && refine <= 0 && (refine != -1 || (fact_checked && sweeps_needed == 0));
    }
    // skip: that many of the last levels have been back-substituted already (1: the last level
    // inside its forward / factor kernel -- a cyclic-reduction level that folds the top block
    // in; 2: the two last levels by tfk_cr_tail)
    void backsub_chain(const double* rhs1, double* x1, int skip) {
        if (skip == 1 && scalar_fused_ok()) {
            TfScalarArgs t = scalar_args(rhs1, x1);
            upd_req_on = false;
            launch(TFK_S_BWD, (unsigned)t.lv[1].L.Ptot, 1, 512, &t, sizeof(t));
            return;
        }
        for (size_t l = levels.size() - (size_t)skip; l-- > 0;) {
            TfLevelArgs a = level_args(l, rhs1, x1);
            if (l == 0) {
                // (twisted: grid.y = 2, the down and the up half of every chunk, tf_twist_h)
                const unsigned gy = a.twist ? 2u : 1u;
                const bool take = upd_req_on && update_allowed();
                upd_req_on = false;                                   // (one solve only: not its refinement sweeps)
                if (const unsigned lds = l1_twist_lds(a)) {
                    // both in one launch, y in LDS
                    if (take) {
                        a.upd_out = upd_req.out; a.upd_base = upd_req.base; a.upd_k0 = upd_req.k0;
                        a.upd_c0 = upd_req.c0; a.upd_c1 = upd_req.c1; a.upd_n = upd_req.n;
                        upd_done = true;
                    }
                    launch(TFK_L1_FWD2_BACKSUB, cdiv(a.L.Ptot, 64), 1, 128, &a, sizeof(a), lds);
                    continue;
                }
                if (l1_respike) launch(TFK_L1_FWD2, cdiv(a.L.Ptot, 64), gy, 64, &a, sizeof(a));
                launch(l1_respike ? TFK_L1_BACKSUB_U : TFK_L1_BACKSUB, cdiv(a.L.Ptot, 64), gy, 64, &a, sizeof(a));
            }
            else if (levels[l]->cr) launch(TFK_CR_BWD, (unsigned)a.L.Ptot, 1, cr_block(), &a, sizeof(a));
            else launch(TFK_BT_BACKSUB, cdiv((int64_t)a.L.Ptot * tfb::coop_group(levels[l]->B), 64), 1, 64, &a, sizeof(a));
        }
    }
    // b = mp * nvar <= 2 with the plan [level 1 | 256-node cyclic-reduction chunks | one chunk]: a solve
    // is two launches (tfk_s_fwd / tfk_s_bwd, TfScalarArgs) instead of six
    bool s_fuse = true;            // (TRIFLOW_S_FUSE=0: A/B runs, tests)
    unsigned* sfuse_counter = nullptr;
    bool scalar_fused_ok() const {
        return s_fuse && tfb::is_device_build() && !tiny && top.B <= 2 && levels.size() == 3 && levels[1]->cr &&
               levels[2]->cr && levels[2]->L.P == 1 && fold_top() && fuse_asm_ok() && !l1_respike;
    }
    TfScalarArgs scalar_args(const double* rhs1, double* x1) {
        TfScalarArgs t;
        for (size_t l = 0; l < 3; ++l) t.lv[l] = level_args(l, rhs1, x1);
        if (!sfuse_counter) sfuse_counter = (unsigned*)tfb::dev_alloc((size_t)nsys * sizeof(unsigned));
        t.counter = sfuse_counter;
        return t;
    }
    // The two last levels of a solve go in one launch (tfk_cr_tail) when both are cyclic-reduction
    // levels of 3 <= b <= 6 and the first of them has at most 8 chunks per system
    bool cr_tail = true;
    bool tail_ok() const {
        const size_t n = levels.size();
        return cr_tail && tfb::is_device_build() && n >= 3 && top.B >= 3 && top.B <= 6 && levels[n - 1]->cr && levels[n - 2]->cr &&
               levels[n - 1]->L.P == 1 && levels[n - 2]->L.P <= 8;      // TF_CR_TAIL_MAXB, TF_CR_TAIL_WAVES
    }
    void solve_once(const double* rhs1, double* x1) {
        if (tiny) {
            TfTinyArgs t = tiny_args(rhs1, x1);
            launch(TFK_TINY_SOLVE, cdiv(nsys, 64), 1, 64, &t, sizeof(t));
            return;
        }
        if (scalar_fused_ok()) {
            TfScalarArgs t = scalar_args(rhs1, x1);
            launch(TFK_S_FWD, (unsigned)t.lv[1].L.Ptot, 1, 512, &t, sizeof(t));
            backsub_chain(rhs1, x1, 1);
            return;
        }
        const bool tail = tail_ok();
        for (size_t l = 0; l < levels.size(); ++l) {
            if (tail && l + 2 == levels.size()) {
                TfTailArgs t;
                t.lv[0] = level_args(l, rhs1, x1);
                t.lv[1] = level_args(l + 1, rhs1, x1);
                launch(TFK_CR_TAIL, (unsigned)nsys, 1, 64u * 8u, &t, sizeof(t));
                backsub_chain(rhs1, x1, 2);
                return;
            }
            TfLevelArgs a = level_args(l, rhs1, x1);
            unsigned gx = cdiv(a.L.Ptot, 64);
            if (l == 0) launch(TFK_L1_SOLVE, gx, 2, 64, &a, sizeof(a));
            else if (levels[l]->cr) { launch(TFK_CR_FWD, (unsigned)a.L.Ptot, 1, cr_block(), &a, sizeof(a)); continue; }
            else launch(TFK_BT_RHS, cdiv((int64_t)a.L.Ptot * tfb::coop_group(levels[l]->B), 64), 2, 64, &a, sizeof(a));
            if (l == 0) { if (!a.fuse_asm) launch(TFK_L1_ASM_RHS, gx, 1, asm_block(), &a, sizeof(a)); }
            else launch(TFK_BT_ASM_RHS, cdiv((int64_t)a.L.Ptot * tfb::coop_group(levels[l]->B), 64), 1, 64, &a, sizeof(a));
        }
        if (!fold_top()) { TfTopArgs t = top_args(); launch(TFK_TOP_SOLVE, cdiv(nsys, 64), 1, 64, &t, sizeof(t)); }
        backsub_chain(rhs1, x1, fold_top() ? 1 : 0);
    }
    void refine_sweep(const double* rhs1, double* x1) {
        spmv(x1, Wjv.p, factor_c);                               // c J x
        const double* xs[3] = {rhs1, x1, Wjv.p};
        vec(TF_VEC_RESID, Wres.p, nullptr, 3, xs, nullptr);      // r = (b - x) + c J x
        solve_once(Wres.p, Wdel.p);
        const double* ys[2] = {x1, Wdel.p};
        vec(TF_VEC_ADD, x1, nullptr, 2, ys, nullptr);
    }
    // componentwise (Oettli-Prager) backward error
    //   max_i |b - A x|_i / (|x| + |c J||x| + |b|)_i
    double backward_error(const double* rhs1, const double* x1) {
        ++n_checks;
        tfb::memset0(red.p, sizeof(double), stream);
        TfBerrArgs a;
        std::memset(&a, 0, sizeof(a));
        a.L = L1; a.Jv = Jv.p; a.x = x1; a.rhs = rhs1; a.c = factor_c; a.red = red.p;
        a.parsca = parsca.p; a.dx = dx.p; a.one_node = -1;
        unsigned gx = sweep_gx(), gy = cdiv(L1.M, spec.seg);
        launch(TFK_BERR, gx, gy, spec.sweep_block, &a, sizeof(a));
        double h = 0;
        tfb::d2h(&h, red.p, sizeof(h), stream);
        return h;
    }
    // The monitor of the Theta and BDF-2 steps (the Rosenbrock steps have theirs inside the J @ v pass
    // of stage 1, tfk_spmv_mon): between two synchronising checks every new factorisation has the same
    // backward error measured at ONE node of every level-1 chunk -- a different one in every step, so
    // that every chunk's elimination is probed in every step and every row once per chunk length (32
    // steps) -- with no host wait: a thread per chunk, ~45 loads each (config 5: 45 MB, ~1 % of a step;
    // the full pass is 576 MB).  The maximum goes to red[4] and is looked at by the next synchronising
    // call, like the Rosenbrock monitor's.
    // xbase: the state the step started from, when x1 is the new state of a step whose solve leaves
    // U + delta instead of delta (TfBerrArgs::xbase).
    unsigned mon_phase = 0;
    bool sampled_monitor_due() const {
        return refine < 0 && !reused && !delegated && !tiny && have_factor && !(refine == -1 && check_now);
    }
    void monitor_sampled(const double* rhs1, const double* x1, const double* xbase) {
        TfBerrArgs a;
        std::memset(&a, 0, sizeof(a));
        a.L = L1; a.Jv = Jv.p; a.x = x1; a.rhs = rhs1; a.c = factor_c; a.red = red.p + 4;
        a.parsca = parsca.p; a.dx = dx.p; a.xbase = xbase;
        // (a stride coprime with every chunk length up to 64: consecutive steps look at nodes far apart)
        a.one_node = (int)((mon_phase++ * 13u) % 4096u);
        launch(TFK_BERR, sweep_gx(), 1, spec.sweep_block, &a, sizeof(a));
        monitored = true;
    }
    // x = (I - c J)^-1 rhs.  refine > 0: that many refinement sweeps; refine == -1
    // (default): the first solve after every factorisation measures the backward
    // error, and only a factorisation that lost accuracy (block elimination does
    // not pivot across blocks) is polished, this solve and the following ones.
    void solve(const double* rhs1, double* x1) {
        if (!have_factor) throw std::runtime_error("tf_solve: matrix not factorised");
        if (delegated) { delegate_solve(rhs1, x1); return; }
        solve_once(rhs1, x1);
        polish(rhs1, x1);
    }
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
    void transfer_to(tf_solver* dst, const double* src_planes, double* dst_planes, int ncomp) {
        ensure_staging((size_t)ncomp * nsys * N);
        perm(1 /*OUT_SOA*/, src_planes, staging.p, ncomp);
        dst->perm(0 /*IN_SOA*/, staging.p, dst_planes, ncomp);
    }
    // what the child's factorisation in memory belongs to (the child holds one; both sets of factor
    // buffers of a constant-matrix solver may be delegated)
    double fb_c = 0.0;
    uint64_t fb_ver = 0;
    bool fb_valid = false;
    bool fb_touched = false;       // the child ran since this solver last looked at its status
    void delegate_factor(double c) {
        tf_solver* fb = ensure_fallback();
        fb_touched = true;
        fb->mode = mode;
        copy(fb->parsca.p, parsca.p, parsca.n * sizeof(double));
        copy(fb->dx.p, dx.p, (size_t)nsys * sizeof(double));
        ++fb->par_ver;
        if (spec.parvec_mask) transfer_to(fb, parvec.p, fb->parvec.p, spec.npar);
        if (spec.nnz > 0) transfer_to(fb, Jv.p, fb->Jv.p, spec.nnz);
        fb->have_jac = true;
        fb->have_factor = false;
        fb->factor(c);
        fb_c = c; fb_ver = par_ver; fb_valid = true;
        delegated = true;
        fact_needs_refine = true;                    // (reported as "refined": the plan in use is not the one asked for)
    }
    void delegate_solve(const double* rhs1, double* x1) {
        if (!(fb_valid && fb_c == factor_c && fb_ver == par_ver)) delegate_factor(factor_c);
        tf_solver* fb = fallback;
        fb_touched = true;
        fb->mode = mode;
        transfer_to(fb, rhs1, fb->Wrhs.p, spec.nvar);
        fb->solve(fb->Wrhs.p, fb->Wstage.p);
        fb->transfer_to(this, fb->Wstage.p, x1, spec.nvar);
        last_omega = fb->last_omega;
        if (fb->unstable) { fb->unstable = false; unstable = true; }
    }
    void polish(const double* rhs1, double* x1) {
        if (refine > 0) {
            for (int it = 0; it < refine; ++it) refine_sweep(rhs1, x1);
        } else if (refine == -1) {
            if (!fact_checked) {
                last_omega = backward_error(rhs1, x1);
                fact_checked = true;
                // (recorded below, once the number of sweeps is known)
                sweeps_needed = 0;
                if (!(last_omega <= refine_trigger)) {
                    // polish down to 1e-13 (or until it stops improving), at most 6 sweeps
                    double prev = last_omega;
                    while (sweeps_needed < 6 && !(last_omega <= 1e-13)) {
                        refine_sweep(rhs1, x1);
                        ++sweeps_needed;
                        last_omega = backward_error(rhs1, x1);
                        if (!(last_omega < 0.5 * prev)) break;
                        prev = last_omega;
                    }
                }
                fact_needs_refine = sweeps_needed > 0;
                if (berr_adaptive)
                    berr_cur = (sweeps_needed == 0 && last_omega <= 0.01 * refine_trigger)
                        ? std::min(2 * berr_cur, std::max(berr_max, berr_every)) : berr_every;
                // A factorisation that refinement cannot bring below 1e-6 broke down (no pivoting across
                // separators).  SuperLU never refuses a non-singular system (schemes.py:149, 557), so
                // before giving up the library solves this matrix on a plan with 8 x longer level-1
                // chunks (fewer separators; one chunk of a scalar model is a pivoted band LU): replan
                // (the rescue is tried well before the refusal: where refinement was needed and leaves
                // more than 1e-12 -- healthy plans read 1e-16 ... 7e-13 without any -- the elimination is
                // losing digits that cond(A) multiplies in the solution)
                bool replan = false;
                if (sweeps_needed > 0 && !(last_omega <= 1e-12) && can_replan()) replan = true;
                else if (!(last_omega <= 1e-6)) unstable = true;
                if (Checked* e = checked_like(factor_c)) { e->c = factor_c; e->sweeps = sweeps_needed; e->at = n_factor; e->replan = replan; }
                else {
                    if (checked.size() >= 4) checked.erase(checked.begin());
                    checked.push_back(Checked{factor_c, sweeps_needed, n_factor, replan});
                }
                if (replan) {
                    ++n_replans;
                    tfb::memset0(status, sizeof(int), stream);      // (what the abandoned plan may have flagged)
                    delegate_factor(factor_c);
                    delegate_solve(rhs1, x1);
                }
            } else {
                for (int it = 0; it < sweeps_needed; ++it) refine_sweep(rhs1, x1);
            }
        }
    }

    // Between two explicit (synchronising) checks a Rosenbrock step measures the backward error
    // of its factorisation inside its first J @ v pass (tfk_spmv_mon: no launch of its own, no
    // synchronisation); with refine = -2 every step does.  The worst value since the last look is
    // read here, at the synchronising calls.
    bool monitored = false;
    // have_flag / have_worst: values that already came back with another download of this call
    void check_status(const int* have_flag = nullptr, const double* have_worst = nullptr) {
        int flag = 0;
        if (have_flag) flag = *have_flag;
        else tfb::d2h(&flag, status, sizeof(int), stream);
        if (monitored) {
            double worst = 0.0;
            if (have_worst) worst = *have_worst;
            else tfb::d2h(&worst, red.p + 4, sizeof(double), stream);
            tfb::memset0(red.p + 4, sizeof(double), stream);
            monitored = false;
            if (worst > refine_trigger || worst != worst) {
                // some factorisation since the last check lost accuracy that the checked ones had
                // not: forget the verdicts, the next factorisation is checked (and refined)
                checked.clear();
                berr_cur = berr_every;
                monitor_omega = worst;
                if (!(worst <= 1e-6)) { last_omega = worst; unstable = true; }
            }
        }
        if (flag != 0) {
            tfb::memset0(status, sizeof(int), stream);
            throw std::runtime_error("banded solver: singular or non-finite pivot block");
        }
        // (the child's flag costs a blocking read of its own: only when it ran since the last look)
        if (fallback && fb_touched) { fb_touched = false; fallback->check_status(); }
        if (unstable) {
            unstable = false;
            throw std::runtime_error("banded solver: the block elimination lost accuracy (backward error " +
                                     std::to_string(last_omega) + " after refinement); no pivoting across blocks");
        }
    }
};

extern "C" {

const char* tf_last_error(void) { return g_last_error.c_str(); }

int tf_runtime_info(int32_t* is_device_build, int32_t* device_count) {
    TF_API_BEGIN
    if (is_device_build) *is_device_build = tfb::is_device_build() ? 1 : 0;
    if (device_count) *device_count = tfb::device_count();
    TF_API_END
}

int tf_set_device(int32_t ordinal) {
    TF_API_BEGIN
    tfb::set_device(ordinal);
    TF_API_END
}

int tf_kernel_count(void) { return TFK_COUNT; }
const char* tf_kernel_name(int32_t kernel) {
    static const char* names[TFK_COUNT] = TF_KERNEL_NAMES;
    return (kernel >= 0 && kernel < TFK_COUNT) ? names[kernel] : "";
}

int tf_model_create(const tf_model_spec* spec, const void* code, size_t size, tf_model** out) {
    TF_API_BEGIN
    require(spec && out, "tf_model_create: null argument");
    require(spec->nvar >= 1 && spec->nvar + spec->nh <= TF_MAX_FIELDS, "tf_model_create: bad field count");
    require(spec->npar >= 0 && spec->npar <= TF_MAX_PARS, "tf_model_create: bad parameter count");
    require(spec->mp >= 1 && spec->seg >= 1 && spec->sweep_block >= 64, "tf_model_create: bad stencil constants");
    std::unique_ptr<tf_model> m(new tf_model());
    m->spec = *spec;
    m->module = tfb::module_load(code, size);
    *out = m.release();
    TF_API_END
}

int tf_model_add_alternate(tf_model* model, const void* code, size_t size, uint64_t kernel_mask) {
    TF_API_BEGIN
    require(model && code, "tf_model_add_alternate: null argument");
    tfb::module_add_alternate(model->module, code, size, kernel_mask);
    TF_API_END
}

void tf_model_destroy(tf_model* model) { delete model; }

}  // extern "C"
namespace {
// shared: the stream of the solver this one serves as the longer-chunk plan of (tf_solver::fallback)
tf_solver* make_solver(tf_model* model, int64_t N, int32_t nsys, int32_t periodic,
                       const tf_solver_opts* opts, tfb::Stream* shared) {
    require(model != nullptr, "tf_solver_create: null argument");
    const tf_model_spec& sp = model->spec;
    require(nsys >= 1, "tf_solver_create: nsys must be >= 1");
    // (shorter than one stencil window: the dense path, TfTinyArgs; periodic ghost cells need mp nodes
    // to copy from, compilers.py:257-260)
    require(N >= (periodic ? sp.mp : 1), "tf_solver_create: a periodic grid needs at least mp nodes");
    require(N * (int64_t)nsys < (int64_t)1 << 31, "tf_solver_create: too many nodes for 32-bit chunk indices");
    std::unique_ptr<tf_solver> s(new tf_solver());
    s->model = model; s->spec = sp; s->N = N; s->nsys = nsys; s->periodic = periodic ? 1 : 0;
    int m1 = opts && opts->m1 > 0 ? opts->m1 : 0;             // 0: chosen below from the problem size
    int mup = opts && opts->m_upper > 0 ? opts->m_upper : 6;
    s->nstate = opts && opts->nstate > 0 ? opts->nstate : 3;
    // 0 = never, n > 0 = fixed sweeps, -1 = auto (explicit checks + the in-pass monitor of the
    // Rosenbrock steps), -2 = the monitor only (no synchronising check at all)
    s->refine = opts ? opts->refine : -1;
    if (opts && opts->berr_every > 0) { s->berr_every = s->berr_cur = opts->berr_every; s->berr_adaptive = false; }
    if (opts && opts->device >= 0) tfb::set_device(opts->device);
    mup = std::max(mup, 2);
    if (shared) { s->stream = shared; s->owns_stream = false; }
    else s->stream = tfb::stream_create();
    if (const char* v = getenv("TRIFLOW_REPLAN")) s->replan_on = atoi(v) != 0;
    s->graphs_on = tfb::graphs_supported() && (int64_t)N * nsys <= 50000;
    if (const char* v = getenv("TRIFLOW_GRAPHS")) s->graphs_on = tfb::graphs_supported() && atoi(v) != 0;
    if (const char* v = getenv("TRIFLOW_FUSE_STAGE")) s->fuse_stage = atoi(v) != 0;      // A/B runs
    if (const char* v = getenv("TRIFLOW_S_FUSE")) s->s_fuse = atoi(v) != 0;
    if (const char* v = getenv("TRIFLOW_L1_FUSE_BACKSUB")) s->l1_fuse_backsub = atoi(v) != 0;
    if (const char* v = getenv("TRIFLOW_L1_FUSE_ASM")) s->l1_fuse_asm = atoi(v) != 0;
    if (const char* v = getenv("TRIFLOW_FUSE_UPDATE")) s->upd_fuse = atoi(v) != 0;
    if (const char* v = getenv("TRIFLOW_HOOK_IN_PLACE")) s->hook_in_place = atoi(v) != 0;
    s->l1_respike = TF_RESPIKE_MODEL(sp.mp, sp.nvar) && (int64_t)N * nsys >= TF_RESPIKE_MIN_NODES;
    if (const char* v = getenv("TRIFLOW_L1_TWIST")) s->l1_twist = atoi(v) != 0 ? 1 : 0;
    if (const char* v = getenv("TRIFLOW_L1_RESPIKE"))                                   // A/B runs, tests
        s->l1_respike = TF_RESPIKE_MODEL(sp.mp, sp.nvar) && atoi(v) != 0;

    // ---- level plan: chunk levels until a single chunk is left, then the top block.
    // Reduced levels: walks over chunks of m_upper nodes, or -- where the back end has
    // them (3 <= b <= 8 on the GPU) -- cyclic reduction inside chunks of up to 16 nodes.
    const int b2 = sp.mp * sp.nvar;
    s->use_cr = tfb::cyclic_reduction(b2);
    if (m1 == 0) {
        // Level-1 chunk length.  A walk costs ~4 us per node of a chunk whatever the grid
        // size, so a small problem (too few chunks to fill the GPU anyway) is latency-bound
        // by it: shorter chunks, more (cheap, cyclic-reduction) levels.  Large problems are
        // throughput-bound and want the smallest reduced system.  Scanned on MI355X with
        // tools/gpu_small_n_scan.py (N = 200 ... 4e5) and tools/gpu_plan_scan.sh (N = 1e6).
        const int64_t total = (int64_t)N * nsys;
        m1 = 32;
        // (not for a single equation with a 5-point stencil: dispersion-dominated ones -- KdV --
        // lose digits with every separator, tools/gpu_scalar_m1.py, so they keep long chunks)
        const bool dispersive_capable = sp.nvar == 1 && sp.mp >= 2;
        if (s->use_cr && !dispersive_capable)
            m1 = total <= 30000 ? 4 : (total <= 200000 ? 8 : (total <= 600000 ? 16 : 32));
        // b <= 2 (round 4): a solve is two launches when the plan is [level 1 | 256-node chunks | one
        // chunk] (tfk_s_fwd / tfk_s_bwd), i.e. while level 1 has at most 65 536 chunks per system; with
        // the reduced levels that cheap the shortest such chunks win (config 2: m1 = 16 against 32:
        // 19 400 against 18 400 steps/s, 16 000 against 14 600 factorising in every step,
        // profiles/r04_ab_runs.txt)
        if (s->use_cr && !dispersive_capable && b2 <= 2 && total > 600000) {
            m1 = 16;
            while (N / m1 > 65536) m1 *= 2;
        }
    }
    m1 = std::max(m1, 2 * sp.mp);
    s->m1_used = m1; s->mup_used = mup;
    s->tiny = N < 2 * sp.mp + 1;
    {
        // Reduced levels: cyclic reduction inside 16-node chunks wherever the back end has the
        // kernels for this block size (b <= 8).  Round 1 kept the chunk walks (tfk_bt_*) for levels
        // above 40 000 nodes, where its one-wavefront-per-chunk factorisation lost to them; with a
        // wavefront per node (tf_cr2_hip.h) cyclic reduction wins there too (config 5: 507 -> 524
        // steps/s, 8 members per GPU: 1897 -> 1983; profiles/r02_ab_runs.txt, r2v).  The walks
        // serve b > 8, the host emulation, and TRIFLOW_CR_MAX_NODES=<n> for comparisons.
        // (scalar models, b <= 2: one thread per node, chunks of 256)
        const int cr_cap = b2 <= 2 ? TF_CRS_MAXLEN : TF_CR_MAXLEN;
        const int cr_len = opts && opts->m_upper > 0 ? std::min(std::max(opts->m_upper, 2), cr_cap) : cr_cap;
        int64_t cr_max_nodes = (int64_t)1 << 40;
        if (const char* v = getenv("TRIFLOW_CR_MAX_NODES")) cr_max_nodes = atoll(v);
        int n = (int)N, B = sp.nvar, MP = sp.mp, m = m1;
        bool first = true;
        while (true) {
            const bool cr = !first && s->use_cr && (int64_t)n * nsys <= cr_max_nodes;
            int P = std::max(1, n / m);
            if (cr) P = (n + cr_len - 1) / cr_len;                 // chunk length <= cr_len
            std::unique_ptr<Level> lv(new Level());
            lv->L = make_layout(nsys, n, P, s->periodic);
            lv->B = B; lv->MP = MP; lv->cr = cr;
            s->levels.push_back(std::move(lv));
            if (P == 1) break;
            n = P; B = b2; MP = 1; m = mup; first = false;
        }
        s->top.L = make_layout(nsys, 1, 1, s->periodic);
        s->top.B = b2; s->top.MP = 1;
    }
    s->L1 = s->levels[0]->L;
    const int64_t plane = s->L1.plane;
    require(plane < ((int64_t)1 << 29), "tf_solver_create: more than 2^29 nodes per solver (a plane is addressed with 32-bit byte offsets)");

    // ---- memory
    int64_t& tot = s->bytes;
    for (int i = 0; i < s->nstate; ++i) {
        s->state.emplace_back(new DevBuf());
        s->state.back()->alloc((size_t)sp.nvar * plane, tot);
    }
    s->helpers.alloc((size_t)sp.nh * plane, tot);
    if (sp.parvec_mask) s->parvec.alloc((size_t)sp.npar * plane, tot); else s->parvec.alloc(1, tot);
    s->parsca.alloc((size_t)std::max(sp.npar, 1) * nsys, tot);
    s->dx.alloc(nsys, tot);
    if (sp.uses_x) s->xcoord.alloc(plane, tot); else s->xcoord.alloc(1, tot);
    s->F.alloc((size_t)sp.nvar * plane, tot);
    s->Jv.alloc((size_t)std::max(sp.nnz, 1) * plane, tot);
    DevBuf* work[] = {&s->Wstage, &s->Wsum, &s->Wjv, &s->Wrhs, &s->Wres, &s->Wdel};
    for (DevBuf* w : work) w->alloc((size_t)sp.nvar * plane, tot);
    for (int i = 0; i < TF_MAX_TERMS; ++i) s->K[i].alloc(i < 6 ? (size_t)sp.nvar * plane : 1, tot);
    s->red.alloc(8, tot);
    s->status = (int*)tfb::dev_alloc(sizeof(int));
    if (b2 <= 2) s->sfuse_counter = (unsigned*)tfb::dev_alloc((size_t)nsys * sizeof(unsigned));   // (TfScalarArgs)
    for (size_t l = 0; l < s->levels.size(); ++l) s->levels[l]->alloc(l, nsys, s->l1_respike, tot);
    s->top.alloc_top(b2, nsys, tot);
    s->topAinv.alloc((size_t)b2 * b2 * nsys, tot);
    if (s->tiny) {
        const size_t n = (size_t)N * sp.nvar;
        s->tiny_lu.alloc(n * n * nsys, tot);
        s->tiny_piv = (int*)tfb::dev_alloc(n * nsys * sizeof(int));
    }
    return s.release();
}
}  // namespace

tf_solver* tf_solver::ensure_fallback() {
    if (fallback) return fallback;
    tf_solver_opts o;
    std::memset(&o, 0, sizeof(o));
    o.m1 = (int32_t)std::min<int64_t>(N, std::max<int64_t>(64, 8 * (int64_t)m1_used));
    o.m_upper = 0; o.nstate = 1; o.refine = -1; o.device = -1;
    o.berr_every = 1;                                // every factorisation of the rescue plan is checked
    fallback = make_solver(model, N, nsys, periodic, &o, stream);
    fallback->refine_trigger = 1e-14;                // the rescue polishes whatever it can (the guard path: time is no object)
    bytes += fallback->bytes;
    return fallback;
}

extern "C" {
int tf_solver_create(tf_model* model, int64_t N, int32_t nsys, int32_t periodic,
                     const tf_solver_opts* opts, tf_solver** out) {
    TF_API_BEGIN
    require(model && out, "tf_solver_create: null argument");
    *out = make_solver(model, N, nsys, periodic, opts, nullptr);
    TF_API_END
}

void tf_solver_destroy(tf_solver* solver) {
    if (!solver) return;
    try { tfb::stream_sync(solver->stream); } catch (...) {}
    delete solver;
}

int tf_solver_describe(tf_solver* s, int32_t* nlevels, int32_t* chunks, int32_t max_levels,
                       int64_t* device_bytes) {
    TF_API_BEGIN
    require(s, "null solver");
    if (nlevels) *nlevels = (int32_t)s->levels.size();
    if (chunks)
        for (size_t l = 0; l < s->levels.size() && (int)l < max_levels; ++l) chunks[l] = s->levels[l]->L.P;
    if (device_bytes) *device_bytes = s->bytes;
    TF_API_END
}

// ------------------------------------------------------------------- inputs
int tf_set_state(tf_solver* s, int32_t slot, int32_t first, int32_t nv, const double* host) {
    TF_API_BEGIN
    require(s && host, "null argument");
    require(first >= 0 && nv >= 1 && first + nv <= s->spec.nvar, "tf_set_state: variable range");
    s->slot_written(slot);
    s->upload_planes(host, s->st(slot) + (int64_t)first * s->plane(), nv);
    TF_API_END
}
int tf_get_state(tf_solver* s, int32_t slot, int32_t first, int32_t nv, double* host) {
    TF_API_BEGIN
    require(s && host, "null argument");
    require(first >= 0 && nv >= 1 && first + nv <= s->spec.nvar, "tf_get_state: variable range");
    s->download_planes(s->st(slot) + (int64_t)first * s->plane(), host, nv);
    s->check_status();
    TF_API_END
}
int tf_set_state_flat(tf_solver* s, int32_t slot, const double* uflat) {
    TF_API_BEGIN
    require(s && uflat, "null argument");
    s->slot_written(slot);
    s->upload_aos(uflat, s->st(slot), s->spec.nvar);
    TF_API_END
}
int tf_get_state_flat(tf_solver* s, int32_t slot, double* uflat) {
    TF_API_BEGIN
    require(s && uflat, "null argument");
    s->download_aos(s->st(slot), uflat, s->spec.nvar);
    s->check_status();
    TF_API_END
}
int tf_copy_state(tf_solver* s, int32_t src, int32_t dst) {
    TF_API_BEGIN
    require(s, "null solver");
    if (src != dst) {
        tfb::d2d(s->st(dst), s->st(src), (size_t)s->vecn() * sizeof(double), s->stream);
        s->slot_written(dst);
    }
    TF_API_END
}
int tf_set_helpers(tf_solver* s, int32_t first, int32_t count, const double* host) {
    TF_API_BEGIN
    require(s && host, "null argument");
    require(first >= 0 && count >= 1 && first + count <= s->spec.nh, "tf_set_helpers: range");
    s->upload_planes(host, s->helpers.p + (int64_t)first * s->plane(), count);
    TF_API_END
}
int tf_set_param_scalar(tf_solver* s, int32_t k, const double* values) {
    TF_API_BEGIN
    require(s && values, "null argument");
    require(k >= 0 && k < s->spec.npar, "tf_set_param_scalar: index");
    require(!((s->spec.parvec_mask >> k) & 1u), "tf_set_param_scalar: parameter compiled as per-node array");
    tfb::h2d(s->parsca.p + (int64_t)k * s->nsys, values, sizeof(double) * s->nsys, s->stream);
    ++s->par_ver;
    TF_API_END
}
int tf_set_param_vector(tf_solver* s, int32_t k, const double* host) {
    TF_API_BEGIN
    require(s && host, "null argument");
    require(k >= 0 && k < s->spec.npar, "tf_set_param_vector: index");
    require((s->spec.parvec_mask >> k) & 1u, "tf_set_param_vector: parameter compiled as scalar");
    s->upload_planes(host, s->parvec.p + (int64_t)k * s->plane(), 1);
    ++s->par_ver;
    TF_API_END
}
int tf_set_dx(tf_solver* s, const double* dxv) {
    TF_API_BEGIN
    require(s && dxv, "null argument");
    tfb::h2d(s->dx.p, dxv, sizeof(double) * s->nsys, s->stream);
    ++s->par_ver;
    TF_API_END
}
int tf_set_constant_jacobian(tf_solver* s, int32_t on) {
    TF_API_BEGIN
    require(s, "null solver");
    s->jconst = on != 0;
    s->cf_valid = false;
    s->meta_alt.cf_valid = false;
    s->drop_graphs();
    TF_API_END
}
int tf_set_x(tf_solver* s, const double* x) {
    TF_API_BEGIN
    require(s && x, "null argument");
    if (s->spec.uses_x) s->upload_planes(x, s->xcoord.p, 1);
    TF_API_END
}
int tf_set_dirichlet(tf_solver* s, int32_t n, const int32_t* var, const int64_t* node, const double* value) {
    TF_API_BEGIN
    require(s, "null solver");
    require(n >= 0, "tf_set_dirichlet: n");
    s->drop_graphs();                       // the captured launches hold the old buffers
    if (s->dir_var) { tfb::dev_free(s->dir_var); s->dir_var = nullptr; }
    if (s->dir_node) { tfb::dev_free(s->dir_node); s->dir_node = nullptr; }
    s->ndir = 0;
    if (n > 0) {
        require(var && node && value, "tf_set_dirichlet: null arrays");
        std::vector<int> nodes(n);
        for (int i = 0; i < n; ++i) {
            require(var[i] >= 0 && var[i] < s->spec.nvar, "tf_set_dirichlet: variable index");
            require(node[i] >= -s->N && node[i] < s->N, "tf_set_dirichlet: node index");
            nodes[i] = (int)node[i];
        }
        s->dir_var = (int*)tfb::dev_alloc(sizeof(int) * n);
        s->dir_node = (int*)tfb::dev_alloc(sizeof(int) * n);
        int64_t dummy = 0;
        s->dir_val.alloc(n, dummy);
        s->dir_val_post.alloc(n, dummy);
        tfb::h2d(s->dir_var, var, sizeof(int) * n, s->stream);
        tfb::h2d(s->dir_node, nodes.data(), sizeof(int) * n, s->stream);
        tfb::h2d(s->dir_val.p, value, sizeof(double) * n, s->stream);
        tfb::h2d(s->dir_val_post.p, value, sizeof(double) * n, s->stream);
        s->ndir = n;
        s->dir_h.assign(value, value + n);
        s->dir_post_h = s->dir_h;
    }
    s->slot_hook.clear();
    TF_API_END
}

// Point writes into a resident state slot (what a Python hook such as the README's
// ``fields.U[0] = 1`` amounts to): n values, applied to every system of the solver.
int tf_poke(tf_solver* s, int32_t slot, int32_t n, const int32_t* var, const int64_t* node, const double* value) {
    TF_API_BEGIN
    require(s, "null solver");
    require(n >= 0, "tf_poke: n");
    if (n == 0) return 0;
    s->slot_written(slot);
    require(var && node && value, "tf_poke: null arrays");
    std::vector<int> nodes(n);
    for (int i = 0; i < n; ++i) {
        require(var[i] >= 0 && var[i] < s->spec.nvar, "tf_poke: variable index");
        require(node[i] >= -s->N && node[i] < s->N, "tf_poke: node index");
        nodes[i] = (int)node[i];
    }
    if (n <= TF_POKE_MAX) {
        // the usual case (a couple of boundary nodes): passed by value with the launch
        TfPokeArgs a;
        std::memset(&a, 0, sizeof(a));
        a.L = s->L1; a.fields = s->st(slot); a.n = n;
        for (int i = 0; i < n; ++i) { a.var[i] = var[i]; a.node[i] = nodes[i]; a.value[i] = value[i]; }
        s->launch(TFK_POKE, tf_solver::cdiv((int64_t)n * s->nsys, 64), 1, 64, &a, sizeof(a));
        return 0;
    }
    // one packed upload into a scratch buffer that stays with the solver: [values | vars | nodes]
    const size_t bytes = (size_t)n * (sizeof(double) + 2 * sizeof(int));
    if (s->poke_bytes < bytes) {
        if (s->poke_buf) tfb::dev_free(s->poke_buf);
        s->poke_bytes = std::max<size_t>(bytes, 1024);
        s->poke_buf = (char*)tfb::dev_alloc(s->poke_bytes);
    }
    std::vector<char> host(bytes);
    std::memcpy(host.data(), value, sizeof(double) * n);
    std::memcpy(host.data() + sizeof(double) * n, var, sizeof(int) * n);
    std::memcpy(host.data() + sizeof(double) * n + sizeof(int) * n, nodes.data(), sizeof(int) * n);
    tfb::h2d(s->poke_buf, host.data(), bytes, s->stream);       // returns when the copy has landed
    TfDirichletArgs a;
    a.L = s->L1; a.fields = s->st(slot); a.n = n;
    a.value = (const double*)s->poke_buf;
    a.var = (const int*)(s->poke_buf + sizeof(double) * n);
    a.node = (const int*)(s->poke_buf + sizeof(double) * n + sizeof(int) * n);
    s->launch(TFK_DIRICHLET, tf_solver::cdiv((int64_t)n * s->nsys, 64), 1, 64, &a, sizeof(a));
    TF_API_END
}

// Values of single nodes of a resident slot (a Python hook reading a neighbour, e.g. the
// zero-gradient condition ``fields.U[0] = fields.U[1]``): out[i * nsys + e].
int tf_peek(tf_solver* s, int32_t slot, int32_t n, const int32_t* var, const int64_t* node, double* out) {
    TF_API_BEGIN
    require(s, "null solver");
    require(n >= 0, "tf_peek: n");
    if (n == 0) return 0;
    require(var && node && out, "tf_peek: null arrays");
    const TfLayout& L = s->L1;
    const int big = L.rem * (L.mbase + 1);
    for (int i = 0; i < n; ++i) {
        require(var[i] >= 0 && var[i] < s->spec.nvar, "tf_peek: variable index");
        require(node[i] >= -s->N && node[i] < s->N, "tf_peek: node index");
        const int g = (int)(node[i] < 0 ? node[i] + s->N : node[i]);
        int p, li;                                         // node -> (chunk, row), as tf_locate
        if (g < big) { p = g / (L.mbase + 1); li = g - p * (L.mbase + 1); }
        else { const int h = g - big; p = L.rem + h / L.mbase; li = h - (h / L.mbase) * L.mbase; }
        for (int e = 0; e < s->nsys; ++e) {
            const int64_t off = (int64_t)var[i] * L.plane + (int64_t)li * L.Ptot + (e * L.P + p);
            tfb::d2h(out + (int64_t)i * s->nsys + e, s->st(slot) + off, sizeof(double), s->stream);
        }
    }
    TF_API_END
}

// --------------------------------------------------------------- seam #1
int tf_set_dirichlet_values(tf_solver* s, const double* before, const double* after) {
    TF_API_BEGIN
    require(s, "null solver");
    if (s->ndir > 0) {
        if (before) { tfb::h2d(s->dir_val.p, before, sizeof(double) * s->ndir, s->stream); s->dir_h.assign(before, before + s->ndir); }
        if (after) { tfb::h2d(s->dir_val_post.p, after, sizeof(double) * s->ndir, s->stream); s->dir_post_h.assign(after, after + s->ndir); }
    }
    TF_API_END
}

int tf_eval(tf_solver* s, int32_t slot, int32_t with_j) {
    TF_API_BEGIN
    require(s, "null solver");
    s->sweep(s->st(slot), with_j != 0);
    TF_API_END
}
int tf_eval_repeat(tf_solver* s, int32_t slot, int32_t with_j, int32_t reps, double* total_ms) {
    TF_API_BEGIN
    require(s && total_ms && reps >= 1, "tf_eval_repeat: arguments");
    const uint64_t saved = s->timing;
    s->timing = 0;
    tfb::Event* a = s->get_event();
    tfb::Event* b = s->get_event();
    tfb::event_record(a, s->stream);
    for (int i = 0; i < reps; ++i) s->sweep(s->st(slot), with_j != 0);
    tfb::event_record(b, s->stream);
    tfb::stream_sync(s->stream);
    *total_ms = tfb::event_elapsed_ms(a, b);
    s->event_pool.push_back(a);
    s->event_pool.push_back(b);
    s->timing = saved;
    TF_API_END
}
int tf_get_F(tf_solver* s, double* Fh) {
    TF_API_BEGIN
    require(s && Fh, "null argument");
    s->download_aos(s->F.p, Fh, s->spec.nvar);
    TF_API_END
}
int tf_get_J(tf_solver* s, double* Jh) {
    TF_API_BEGIN
    require(s && Jh, "null argument");
    require(s->have_jac, "tf_get_J: no Jacobian evaluated yet");
    if (s->spec.nnz > 0) s->download_aos(s->Jv.p, Jh, s->spec.nnz);
    TF_API_END
}

// The drop-in J function returns a scipy.sparse.csc_matrix (compilers.py:330-331): with the
// pattern fixed, its data array is a gather of the value table.  The caller uploads the index
// list once (entry t of the result = value-table entry map[t] = node * nnz + k, system 0) and
// then downloads Jacobians in that order -- no assembly on the host.
int tf_set_csc_map(tf_solver* s, const int32_t* map, int64_t n) {
    TF_API_BEGIN
    require(s && map && n >= 0, "tf_set_csc_map: arguments");
    if (s->csc_map) { tfb::dev_free(s->csc_map); s->csc_map = nullptr; }
    s->csc_map = (int*)tfb::dev_alloc(sizeof(int) * (size_t)std::max<int64_t>(n, 1));
    s->bytes += (int64_t)sizeof(int) * n;
    tfb::h2d(s->csc_map, map, sizeof(int) * (size_t)n, s->stream);
    s->csc_n = n;
    TF_API_END
}
int tf_get_J_mapped(tf_solver* s, double* out) {
    TF_API_BEGIN
    require(s && out, "null argument");
    require(s->have_jac, "tf_get_J_mapped: no Jacobian evaluated yet");
    require(s->csc_map != nullptr, "tf_get_J_mapped: no index list (tf_set_csc_map)");
    s->ensure_staging((size_t)s->csc_n);
    TfGatherArgs a;
    a.L = s->L1; a.Jv = s->Jv.p; a.map = s->csc_map; a.out = s->staging.p; a.n = s->csc_n; a.nnz = s->spec.nnz;
    s->launch(TFK_GATHER, tf_solver::cdiv(s->csc_n, 256), 1, 256, &a, sizeof(a));
    tfb::d2h(out, s->staging.p, (size_t)s->csc_n * sizeof(double), s->stream);
    TF_API_END
}

// --------------------------------------------------------------- seam #3
int tf_factor(tf_solver* s, double c) {
    TF_API_BEGIN
    require(s, "null solver");
    s->factor(c);
    TF_API_END
}
int tf_solve(tf_solver* s, const double* rhs_flat, double* x_flat) {
    TF_API_BEGIN
    require(s && rhs_flat && x_flat, "null argument");
    s->upload_aos(rhs_flat, s->Wrhs.p, s->spec.nvar);
    s->solve(s->Wrhs.p, s->Wstage.p);
    s->download_aos(s->Wstage.p, x_flat, s->spec.nvar);
    s->check_status();
    TF_API_END
}
int tf_matvec(tf_solver* s, const double* v_flat, double* y_flat) {
    TF_API_BEGIN
    require(s && v_flat && y_flat, "null argument");
    require(s->have_jac, "tf_matvec: no Jacobian evaluated yet");
    s->upload_aos(v_flat, s->Wsum.p, s->spec.nvar);
    s->spmv(s->Wsum.p, s->Wjv.p, 1.0);
    s->download_aos(s->Wjv.p, y_flat, s->spec.nvar);
    TF_API_END
}

// --------------------------------------------------------------- seam #2
}  // extern "C"
namespace {
// Theta scheme, reference schemes.py:548-559:
//   fields = copy; hook(t); F, J; B = dt*(F - theta*J@U) + U; A = I - theta*dt*J;
//   U+ = solve(A, B); hook(t+dt)
void step_theta(tf_solver* s, int32_t src, int32_t dst, double dt, double theta) {
    require(src != dst, "tf_step_theta: src and dst slots must differ");
    double* U = s->st(dst);
    s->slot_written(dst);
    const double* Uin = s->stage_input(src, U);                    // copy + hook only when there is a hook
    s->sweep_theta(Uin, dt, theta, s->Wrhs.p);                     // F, J, dt*(F - (theta*J)@U) + U
    s->factor_step(theta * dt, s->Wrhs.p, U);
    if (s->sampled_monitor_due()) s->monitor_sampled(s->Wrhs.p, U, nullptr);      // (I - theta dt J) U+ = B
    s->apply_dirichlet(U, true);
    s->mark_hooked(dst);
}

// Rosenbrock-Wanner fixed step, reference schemes.py:142-174.  With b_pred the maximum of
// |U - U_pred| is left in red[0] (the caller reads it).
void step_row(tf_solver* s, int32_t src, int32_t dst, double dt, int32_t ns, const double* alpha,
              const double* gamma, const double* b, const double* b_pred, bool hook_after, bool want_err) {
    require(ns >= 1 && ns <= 6, "tf_step_row: 1 <= s <= 6");
    require(src != dst, "tf_step_row: src and dst slots must differ");
    double* U = s->st(dst);
    s->slot_written(dst);
    const double* Uin = s->stage_input(src, U);
    s->sweep(Uin, true, 0, nullptr, nullptr, dt);   // J(U) and dt*F(U): right-hand side of stage 0
    const double* ks[TF_MAX_TERMS];
    double cs[TF_MAX_TERMS];
    for (int i = 0; i < ns; ++i) {
        if (i > 0) {
            // F(U + sum_j alpha_ij k_j): the stage state is formed inside the sweep.  It goes to a
            // buffer of its own: F keeps dt*F(U), the right-hand side of stage 0, for the monitor
            // ... plus dt*(J @ sum_j gamma_ij k_j), in the same pass; for i == 1 the pass (every 8th
            // factorisation) also measures the backward error of the stage-0 solve (k0 from dt*F(U))
            double gs[TF_MAX_TERMS];
            for (int j = 0; j < i; ++j) { ks[j] = s->K[j].p; cs[j] = alpha[i * ns + j]; gs[j] = gamma[i * ns + j]; }
            s->stage_rhs(Uin, i, ks, cs, gs, dt, s->Wrhs.p, i == 1 ? s->F.p : nullptr);
        }
        // the last stage of a fixed step of one or two stages: the new state leaves with the solve
        if (i == ns - 1 && ns <= 2 && !(b_pred && want_err))
            s->request_update(U, Uin, ns == 2 ? s->K[0].p : nullptr, b[0], ns == 2 ? b[1] : 0.0, ns);
        if (i == 0) s->factor_step(gamma[0] * dt, s->F.p, s->K[0].p); // factorise + first stage
        else s->solve(s->Wrhs.p, s->K[i].p);
    }
    for (int j = 0; j < ns; ++j) { ks[j] = s->K[j].p; cs[j] = b[j]; }
    if (!s->take_update_done()) s->vec(TF_VEC_SUM, U, Uin, ns, ks, cs);   // U + sum_i b_i k_i
    if (b_pred && want_err) {
        s->zero(s->red.p, sizeof(double));
        for (int j = 0; j < ns; ++j) cs[j] = b_pred[j];
        s->vec(TF_VEC_MAXABS, nullptr, U, ns, ks, cs);             // ||U - (U + sum b_pred k)||_inf
    }
    if (hook_after) { s->apply_dirichlet(U, true); s->mark_hooked(dst); }
}

// per system and variable ||state[a] - state[b]||_ord (ord 2 / 0 = max), out[nsys][nvar]
// with_status: the device-side failure flag and the monitor's worst value come back in the same
// download (one host wait for a whole step-doubling trial) and are looked at like tf_sync does
void diff_norm(tf_solver* s, int32_t slot_a, int32_t slot_b, int32_t ord, double* out, bool with_status = false) {
    require(ord == 0 || ord == 2, "tf_diff_norm: ord must be 2 or 0 (max norm)");
    // (enough workgroups to fill the GPU whatever the number of variables and members: 64 of them
    // took 32 us for the two 8 MB states of config 2; the host adds the partial sums in a fixed order)
    const int nvs = s->spec.nvar * s->nsys;
    const int nb = std::min(1024, std::max(64, 2048 / std::max(nvs, 1)));
    if (s->normbuf.n < (size_t)nb * nvs + 2) s->normbuf.alloc((size_t)nb * nvs + 2, s->bytes);
    TfNormArgs a;
    a.L = s->L1; a.a = s->st(slot_a); a.b = s->st(slot_b); a.partial = s->normbuf.p;
    a.nblocks = nb; a.ord = ord;
    a.status = with_status ? s->status : nullptr; a.mon = with_status ? s->red.p + 4 : nullptr;
    s->launch(TFK_DIFFNORM, nb, nvs, 256, &a, sizeof(a));
    std::vector<double> part((size_t)nb * nvs + 2);
    tfb::d2h(part.data(), s->normbuf.p, ((size_t)nb * nvs + (with_status ? 2 : 0)) * sizeof(double), s->stream);
    if (with_status) {
        int flag = 0;
        std::memcpy(&flag, &part[(size_t)nb * nvs], sizeof(int));
        s->check_status(&flag, &part[(size_t)nb * nvs + 1]);
    }
    for (int vs = 0; vs < nvs; ++vs) {                 // fixed order: deterministic
        double acc = 0.0;
        for (int b = 0; b < nb; ++b) {
            const double v = part[(size_t)vs * nb + b];
            acc = ord == 2 ? acc + v : (v > acc ? v : acc);
        }
        const int v = vs / s->nsys, e = vs % s->nsys;
        out[(size_t)e * s->spec.nvar + v] = ord == 2 ? std::sqrt(acc) : acc;
    }
}
}  // namespace
extern "C" {

namespace {
std::string bits_of(double v) { uint64_t b; std::memcpy(&b, &v, 8); char buf[20]; snprintf(buf, sizeof buf, "%llx", (unsigned long long)b); return buf; }
}
int tf_step_theta(tf_solver* s, int32_t src, int32_t dst, double dt, double theta) {
    TF_API_BEGIN
    require(s, "null solver");
    const std::string key = "T|" + std::to_string(src) + ">" + std::to_string(dst) + "|" + bits_of(dt) + "|" +
        bits_of(theta) + "|" + std::to_string(s->ndir) + (s->input_is_hooked(src) ? "h" : "c") + "|" + std::to_string(s->sweeps_for(theta * dt)) + "|" + std::to_string(s->refine) +
        s->slot_key(theta * dt);      // (a step that reuses a factorisation is another string of launches, on its buffers)
    s->prepare_step(theta * dt);
    s->run_graphed(key, s->step_graphable(theta * dt), [&] { step_theta(s, src, dst, dt, theta); });
    s->mark_hooked(dst);          // (a replayed graph does not run the host side of the step)
    TF_API_END
}

int tf_step_row(tf_solver* s, int32_t src, int32_t dst, double dt, int32_t ns,
                const double* alpha, const double* gamma, const double* b,
                const double* b_pred, int32_t hook_after, double* err_out) {
    TF_API_BEGIN
    require(s && alpha && gamma && b, "null argument");
    require(ns >= 1 && ns <= 6, "tf_step_row: 1 <= s <= 6");
    std::string key = "R|" + std::to_string(src) + ">" + std::to_string(dst) + "|" + bits_of(dt) + "|" +
        std::to_string(ns) + "|" + std::to_string(hook_after) + "|" + std::to_string(s->ndir) + (s->input_is_hooked(src) ? "h" : "c") + "|" +
        std::to_string(s->sweeps_for(gamma[0] * dt)) + "|" + std::to_string(s->refine) + "|" + (b_pred && err_out ? "e" : "-") +
        (s->will_monitor(gamma[0] * dt) ? "m" : "-") + s->slot_key(gamma[0] * dt);
    s->prepare_step(gamma[0] * dt);
    for (int i = 0; i < ns * ns; ++i) key += bits_of(alpha[i]) + bits_of(gamma[i]);
    for (int i = 0; i < ns; ++i) key += bits_of(b[i]) + (b_pred ? bits_of(b_pred[i]) : std::string("-"));
    s->run_graphed(key, s->step_graphable(gamma[0] * dt), [&] {
        step_row(s, src, dst, dt, ns, alpha, gamma, b, b_pred, hook_after != 0, err_out != nullptr); });
    if (hook_after) s->mark_hooked(dst); else s->slot_written(dst);     // (a replayed graph does not run the host side of the step)
    if (err_out) {
        *err_out = 0.0;
        if (b_pred) {
            uint64_t bits = 0;
            tfb::d2h(&bits, s->red.p, sizeof(bits), s->stream);
            std::memcpy(err_out, &bits, sizeof(double));
        }
        s->check_status();
    }
    TF_API_END
}

// One trial of the step-doubling controller that the reference wraps around every scheme
// (schemes.py:33-66, simulation.py:190-197): a coarse step m*dt against `nfine` fine steps
// dt from the same state, and the difference of the two results -- all queued back to back;
// the host waits once, for the norms.  src -> coarse (one step m*dt); src -> tmp -> dst ->
// tmp ... -> dst (nfine steps, nfine even); err_out[nsys] = max_var ||coarse - dst||_ord / (m^2 - 1).
int tf_step_doubling(tf_solver* s, int32_t src, int32_t dst, int32_t tmp, int32_t coarse, double dt,
                     int32_t m, int32_t nfine, const tf_scheme* sch, int32_t ord, double* err_out) {
    TF_API_BEGIN
    require(s && sch && err_out, "null argument");
    require(nfine >= 2 && nfine % 2 == 0, "tf_step_doubling: the fine steps ping-pong between two slots (nfine even)");
    require(m >= 2, "tf_step_doubling: m >= 2");
    const int32_t slots[4] = {src, dst, tmp, coarse};
    for (int i = 0; i < 4; ++i)
        for (int j = i + 1; j < 4; ++j) require(slots[i] != slots[j], "tf_step_doubling: the four slots must differ");
    require(sch->kind == TF_SCHEME_THETA || sch->kind == TF_SCHEME_ROW, "tf_step_doubling: scheme kind");
    auto one = [&](int32_t from, int32_t to, double h) {
        if (sch->kind == TF_SCHEME_THETA) step_theta(s, from, to, h, sch->theta);
        else {
            require(sch->alpha && sch->gamma && sch->b, "tf_step_doubling: tableau");
            step_row(s, from, to, h, sch->stages, sch->alpha, sch->gamma, sch->b, nullptr,
                     sch->hook_after != 0, false);
        }
    };
    one(src, coarse, m * dt);
    int32_t from = src;
    for (int i = 0; i < nfine; ++i) {
        const int32_t to = (i % 2 == 0) ? tmp : dst;
        one(from, to, dt);
        from = to;
    }
    std::vector<double> norms((size_t)s->nsys * s->spec.nvar);
    // the one synchronisation: the norms; the failure flag and the monitor's worst value come back
    // in the same download (written behind the partial sums by the norm kernel itself)
    diff_norm(s, coarse, dst, ord, norms.data(), true);
    for (int e = 0; e < s->nsys; ++e) {
        double worst = 0.0;
        for (int v = 0; v < s->spec.nvar; ++v) {
            const double n = norms[(size_t)e * s->spec.nvar + v];
            worst = (n > worst || n != n) ? n : worst;
        }
        err_out[e] = worst / ((double)m * m - 1.0);
    }
    TF_API_END
}

// Linearly implicit BDF-2 (not in the reference; DESIGN.md "BDF-2"):
//   (I - 2/3 dt J)(U+ - U) = 1/3 (U - Uprev) + 2/3 dt F     with history
//   (I -     dt J)(U+ - U) = dt F                           first step / dt changed
}  // extern "C"
namespace {
// prev_slot >= 0: U_{n-1} is in that state slot (the caller rotates three slots or more and says
// where; nothing is copied); -1: this step has no history (backward-Euler form); -2: the history
// buffer `h` of the solver / of a scheme object, updated by the sweep
void step_bdf2(tf_solver* s, int32_t src, int32_t dst, double dt, tf_solver::BdfHist* h, bool continuing,
               int32_t prev_slot = -2) {
    require(src != dst, "tf_step_bdf2: src and dst slots must differ");
    double* U = s->st(dst);
    s->slot_written(dst);
    const double* Uin;
    if (!h && s->ndir > 0 && !s->input_is_hooked(src)) {
        // The history of the next step is this step's *hooked* input (the oracle keeps the hooked
        // copy, oracle/numpy_path.py BDF2._prev), and with the history in a state slot that slot is
        // src itself: the boundary values go into src in place (no copy into dst) and the slot is
        // remembered as satisfying them.
        s->apply_dirichlet(s->st(src));
        s->mark_hooked(src, false);
        Uin = s->st(src);
    } else {
        Uin = s->stage_input(src, U);
    }
    bool two_step;
    const double* prev = nullptr;
    double* prev_out = nullptr;
    if (h) {
        if (h->Uprev.n == 0) h->Uprev.alloc((size_t)s->vecn(), s->bytes);   // history buffers are made on first use
        two_step = continuing && h->have_prev && std::fabs(h->dt_prev - dt) <= 1e-12 * std::fabs(dt);
        prev = prev_out = h->Uprev.p;
        h->have_prev = true;
        h->dt_prev = dt;
    } else {
        require(prev_slot != src && prev_slot != dst, "tf_step_bdf2_from: the history slot must differ from src and dst");
        two_step = prev_slot >= 0;
        prev = two_step ? s->st(prev_slot) : nullptr;
    }
    // rhs = 1/3 (U - Uprev) + 2/3 dt F (two-step) or dt F (first step), and Uprev <- U
    s->sweep_bdf2(Uin, two_step, 1.0 / 3.0, two_step ? (2.0 / 3.0) * dt : dt, s->Wrhs.p, prev, prev_out);
    s->request_update(U, Uin, nullptr, 1.0, 0.0, 1);               // (1.0 * x == x: the sum of TF_VEC_ADD)
    s->factor_step(two_step ? (2.0 / 3.0) * dt : dt, s->Wrhs.p, s->Wdel.p);
    const double* ys[2] = {Uin, s->Wdel.p};
    const bool in_walk = s->take_update_done();
    if (!in_walk) s->vec(TF_VEC_ADD, U, nullptr, 2, ys, nullptr);
    if (s->sampled_monitor_due()) {
        // (the solve left U + delta: measured on the state form unless the input was hooked into dst itself)
        if (!in_walk) s->monitor_sampled(s->Wrhs.p, s->Wdel.p, nullptr);
        else if (Uin != U) s->monitor_sampled(s->Wrhs.p, U, Uin);
    }
    s->apply_dirichlet(U, true);
    s->mark_hooked(dst);
}
}  // namespace
extern "C" {
int tf_step_bdf2(tf_solver* s, int32_t src, int32_t dst, double dt) {
    TF_API_BEGIN
    require(s, "null solver");
    step_bdf2(s, src, dst, dt, &s->bdf0, true);
    TF_API_END
}
// The same step for a scheme object that shares the solver with others: `owner` names its
// history (any non-zero id), `continuing` says that `src` is the state this owner's previous
// step produced -- otherwise the step restarts with the backward-Euler form.
int tf_step_bdf2_owned(tf_solver* s, int32_t src, int32_t dst, double dt, int64_t owner, int32_t continuing) {
    TF_API_BEGIN
    require(s, "null solver");
    require(owner != 0, "tf_step_bdf2_owned: owner id 0 is the solver's own history (tf_step_bdf2)");
    auto& slot = s->bdf_owned[owner];
    if (!slot) slot.reset(new tf_solver::BdfHist());
    step_bdf2(s, src, dst, dt, slot.get(), continuing != 0);
    TF_API_END
}
// The same step with the history in a state slot of the caller: `prev` holds U_{n-1} (the state
// the previous step of the same size started from), or is -1 for a first step / after a change of
// dt (backward-Euler form).  Nothing is copied: a caller that rotates three slots or more saves
// the pass over the history (config 5: 160 MB written per step).
int tf_step_bdf2_from(tf_solver* s, int32_t src, int32_t dst, int32_t prev, double dt) {
    TF_API_BEGIN
    require(s, "null solver");
    require(prev >= -1, "tf_step_bdf2_from: prev is a state slot or -1");
    step_bdf2(s, src, dst, dt, nullptr, true, prev);
    TF_API_END
}
int tf_bdf2_reset(tf_solver* s) {
    TF_API_BEGIN
    require(s, "null solver");
    s->bdf0.have_prev = false;
    TF_API_END
}
int tf_bdf2_release(tf_solver* s, int64_t owner) {
    TF_API_BEGIN
    require(s, "null solver");
    auto it = s->bdf_owned.find(owner);
    if (it != s->bdf_owned.end()) {
        tfb::stream_sync(s->stream);              // the buffer may still be read by a queued step
        s->bdf_owned.erase(it);
    }
    TF_API_END
}

int tf_diff_norm(tf_solver* s, int32_t slot_a, int32_t slot_b, int32_t ord, double* out) {
    TF_API_BEGIN
    require(s && out, "null argument");
    diff_norm(s, slot_a, slot_b, ord, out);
    TF_API_END
}

int tf_backward_error(tf_solver* s, double* omega, int32_t* refined) {
    TF_API_BEGIN
    require(s, "null solver");
    if (omega) *omega = s->last_omega;
    if (refined) *refined = s->fact_needs_refine ? 1 : 0;
    TF_API_END
}

// Diagnostic builds of the kernels (-DTF_STAMPS) record clock stamps per solver level:
// out[level][64] 64-bit counters; the first call switches the recording on.
int tf_debug_stamps(tf_solver* s, uint64_t* out, int32_t max_levels) {
    TF_API_BEGIN
    require(s && out, "null argument");
    // (regions beyond the solver's levels: per-workgroup begin / end times of the level-1 kernels,
    // TF_WGTRACE in tf_entry_hip.h)
    const size_t regions = std::max<size_t>(std::max<size_t>(s->levels.size(), 1), (size_t)std::max(max_levels, 0));
    const size_t need = 64 * regions;
    if (s->stamp_buf.n < need) { s->drop_graphs(); s->stamp_buf.alloc(need, s->bytes); return 0; }
    const size_t n = 64 * std::min<size_t>(s->stamp_buf.n / 64, (size_t)std::max(max_levels, 0));
    tfb::d2h(out, s->stamp_buf.p, n * sizeof(uint64_t), s->stream);
    TF_API_END
}

// Worst backward error the in-pass monitor of the Rosenbrock steps has seen since the last
// synchronising call (tf_sync / downloads reset it), without raising.
int tf_monitor_error(tf_solver* s, double* worst) {
    TF_API_BEGIN
    require(s && worst, "null argument");
    *worst = 0.0;
    if (s->monitored) tfb::d2h(worst, s->red.p + 4, sizeof(double), s->stream);
    TF_API_END
}

int tf_solver_counters(tf_solver* s, int64_t* factorisations, int64_t* checks, int64_t* replans) {
    TF_API_BEGIN
    require(s, "null solver");
    if (factorisations) *factorisations = s->n_factor;
    if (checks) *checks = s->n_checks;
    if (replans) *replans = s->n_replans;
    TF_API_END
}

int tf_solver_kernel_block(tf_solver* s, int32_t kernel, int32_t* block) {
    TF_API_BEGIN
    require(s && block, "null argument");
    require(kernel >= 0 && kernel < TFK_COUNT, "no such kernel");
    *block = (int32_t)tfb::kernel_block(s->model->module, kernel);
    TF_API_END
}

int tf_sync(tf_solver* s) {
    TF_API_BEGIN
    require(s, "null solver");
    tfb::stream_sync(s->stream);
    s->check_status();
    TF_API_END
}

// ------------------------------------------------------------- measurement
int tf_timing_enable(tf_solver* s, int64_t on) {
    TF_API_BEGIN
    require(s, "null solver");
    s->collect_timing();
    s->timing = on < 0 ? ~0ull : (uint64_t)on;
    TF_API_END
}
int tf_timing_reset(tf_solver* s) {
    TF_API_BEGIN
    require(s, "null solver");
    s->collect_timing();
    for (int k = 0; k < TFK_COUNT; ++k) { s->time_ms[k] = 0; s->time_n[k] = 0; }
    TF_API_END
}
int tf_timing_get(tf_solver* s, int32_t kernel, double* total_ms, int64_t* launches) {
    TF_API_BEGIN
    require(s, "null solver");
    require(kernel >= 0 && kernel < TFK_COUNT, "tf_timing_get: kernel index");
    s->collect_timing();
    if (total_ms) *total_ms = s->time_ms[kernel];
    if (launches) *launches = s->time_n[kernel];
    TF_API_END
}

}  // extern "C"
