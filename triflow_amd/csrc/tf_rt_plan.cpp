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
// (this file: the error string, models, the level plan and the memory of a solver)
#include "tf_solver.h"

namespace tfrt { thread_local std::string g_last_error; }

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
    if (const char* v = getenv("TRIFLOW_L1CR_FUSE")) s->l1cr_fuse = atoi(v) != 0;
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
        // (a level that is one chunk may be TF_CRS_TOPLEN long: b = 1 keeps m1 = 16 up to N = 2e6.  m1 = 8
        // with twice the workgroups: 20 500 against 21 000, the last workgroup arrives no earlier)
        if (s->use_cr && !dispersive_capable && b2 <= 2 && total > 600000) {
            m1 = 16;
            while (N / m1 > TF_CRS_MAXLEN * TF_CRS_TOPLEN(b2)) m1 *= 2;
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
            // (scalar models: what fits one chunk of the longer kind is the last level)
            if (cr && b2 <= 2 && cr_len == cr_cap && n <= TF_CRS_TOPLEN(b2)) P = 1;
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
}  // extern "C"
