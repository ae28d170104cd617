// tf_solver: elementary steps -- stencil sweeps, J @ v, vector algebra, declarative hooks (see tf_solver.h)
#include "tf_solver.h"

void tf_solver::vec(int op, double* out, const double* base, int nterms, const double* const* xs, const double* cs, int64_t n, int red_slot, const double* cs2) {
    TfVecArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = n < 0 ? vecn() : n;
    a.nterms = nterms; a.op = op; a.out = out; a.base = base; a.red = red.p + red_slot;
    for (int t = 0; t < nterms; ++t) { a.x[t] = xs[t]; a.c[t] = cs ? cs[t] : 1.0; a.c2[t] = cs2 ? cs2[t] : 0.0; }
    unsigned grid = std::min<unsigned>(cdiv(a.n, 256), 2048u);
    launch((op == TF_VEC_MAXABS || op == TF_VEC_MAXRATIO || op == TF_VEC_SUM_ERR) ? TFK_VEC_MAXABS : TFK_VEC, std::max(grid, 1u), 1, 256, &a, sizeof(a));
}

void tf_solver::perm(int mode, const double* src, double* dst, int ncomp) {
    TfPermArgs a;
    a.L = L1; a.src = src; a.dst = dst; a.ncomp = ncomp; a.mode = mode;
    launch(TFK_PERM, cdiv((int64_t)nsys * N, 256), 1, 256, &a, sizeof(a));
}

void tf_solver::ensure_staging(size_t count) {
    if (staging.n < count) staging.alloc(count, bytes);
}

void tf_solver::upload_planes(const double* host, double* planes, int ncomp) {
    size_t cnt = (size_t)ncomp * nsys * N;
    ensure_staging(cnt);
    tfb::h2d(staging.p, host, cnt * sizeof(double), stream);
    perm(0 /*IN_SOA*/, staging.p, planes, ncomp);
}

void tf_solver::download_planes(const double* planes, double* host, int ncomp) {
    size_t cnt = (size_t)ncomp * nsys * N;
    ensure_staging(cnt);
    perm(1 /*OUT_SOA*/, planes, staging.p, ncomp);
    tfb::d2h(host, staging.p, cnt * sizeof(double), stream);
}

void tf_solver::upload_aos(const double* host, double* planes, int ncomp) {
    size_t cnt = (size_t)ncomp * nsys * N;
    ensure_staging(cnt);
    tfb::h2d(staging.p, host, cnt * sizeof(double), stream);
    perm(3 /*IN_AOS*/, staging.p, planes, ncomp);
}

void tf_solver::download_aos(const double* planes, double* host, int ncomp) {
    size_t cnt = (size_t)ncomp * nsys * N;
    ensure_staging(cnt);
    perm(2 /*OUT_AOS*/, planes, staging.p, ncomp);
    tfb::d2h(host, staging.p, cnt * sizeof(double), stream);
}

void tf_solver::apply_dirichlet(double* fields, bool post) {
    if (ndir == 0) return;
    TfDirichletArgs a;
    a.L = L1; a.fields = fields; a.n = ndir; a.var = dir_var; a.node = dir_node;
    a.value = post ? dir_val_post.p : dir_val.p;
    launch(TFK_DIRICHLET, cdiv((int64_t)ndir * nsys, 64), 1, 64, &a, sizeof(a));
}

void tf_solver::sweep(const double* fields, bool with_j, int nterms, const double* const* kx, const double* kc, double fscale, double* Fout) {
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

void tf_solver::sweep_bdf2(const double* fields, bool two_step, double c0, double c1, double* rhs, const double* prev, double* prev_out) {
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

void tf_solver::sweep_theta(const double* fields, double dt, double theta, double* rhs) {
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

void tf_solver::spmv(const double* v, double* y, double scale, bool absval) {
    TfSpmvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.L = L1; a.Jv = Jv.p; a.v = v; a.y = y; a.scale = scale; a.absval = absval ? 1 : 0;
    a.parsca = parsca.p; a.dx = dx.p;
    unsigned gx = sweep_gx(), gy = cdiv(L1.M, spec.seg);
    launch(TFK_SPMV, gx, gy, spec.sweep_block, &a, sizeof(a));
}

bool tf_solver::alt_ok(double c) const {
    return jconst && alt_allocated && meta_alt.cf_valid && have_jac && same_c(meta_alt.cf_c, c) && meta_alt.cf_ver == par_ver;
}

bool tf_solver::wants_alt(double c) const {
    return jconst && two_slots && !reuse_ok(c) && !alt_ok(c) && cf_valid && have_jac && cf_ver == par_ver;
}

void tf_solver::ensure_alt() {
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

void tf_solver::swap_slots() {
    levels.swap(levels_alt);
    top.swap(top_alt);
    topAinv.swap(topAinv_alt);
    SlotMeta cur;
    cur.factor_c = factor_c; cur.cf_c = cf_c; cur.have_factor = have_factor; cur.cf_valid = cf_valid;
    cur.fact_checked = fact_checked; cur.fact_needs_refine = fact_needs_refine; cur.check_now = check_now;
    cur.cf_ver = cf_ver; cur.sweeps_needed = sweeps_needed; cur.delegated = delegated;
    factor_c = meta_alt.factor_c; cf_c = meta_alt.cf_c; have_factor = meta_alt.have_factor; cf_valid = meta_alt.cf_valid;
    fact_checked = meta_alt.fact_checked; fact_needs_refine = meta_alt.fact_needs_refine; check_now = meta_alt.check_now;
    cf_ver = meta_alt.cf_ver; sweeps_needed = meta_alt.sweeps_needed; delegated = meta_alt.delegated;
    meta_alt = cur;
    slot_id ^= 1;
}

std::string tf_solver::slot_key(double c) const {
    const bool reuse = reuse_ok(c) || alt_ok(c);
    const int slot = reuse_ok(c) ? slot_id : ((alt_ok(c) || wants_alt(c)) ? slot_id ^ 1 : slot_id);
    return std::string(reuse ? "|u" : "|f") + (slot ? "1" : "0");
}

bool tf_solver::fuse_asm_ok() const {
    const int b = spec.mp * spec.nvar;
    return l1_fuse_asm && levels.size() > 1 && levels[1]->cr && (2 * b * b + 1) * 64 * 8 <= 40 * 1024;
}

unsigned tf_solver::l1_factor_block() {
    if (!l1_factor_block_) l1_factor_block_ = tfb::kernel_block(model->module, TFK_L1_FACTOR) == 128 ? 128 : 64;
    return l1_factor_block_;
}

TfTinyArgs tf_solver::tiny_args(const double* rhs1, double* x1) {
    TfTinyArgs t;
    std::memset(&t, 0, sizeof(t));
    t.L = L1; t.Jv = Jv.p; t.parsca = parsca.p; t.dx = dx.p; t.c = factor_c;
    t.lu = tiny_lu.p; t.piv = tiny_piv; t.rhs = rhs1; t.x = x1; t.status = status;
    return t;
}

void tf_solver::stage_rhs(const double* Uin, int nterms, const double* const* ks, const double* ac, const double* gc, double dt, double* y, const TfBerrArgs* probe) {
    if (!fuse_stage) {
        if (probe) launch(TFK_BERR, sweep_gx(), 1, spec.sweep_block, probe, sizeof(*probe));
        sweep(Uin, false, nterms, ks, ac, 1.0, Wstage.p);
        spmv_stage(nterms, ks, gc, Wstage.p, dt, dt, y);
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
    if (probe && nterms == 1) {
        TfStageMonArgs m;
        m.s = a; m.b = *probe;
        launch(TFK_SWEEP_F_STAGE_RHS_MON, gx, gy + 1, spec.sweep_block, &m, sizeof(m));
        return;
    }
    if (probe) launch(TFK_BERR, gx, 1, spec.sweep_block, probe, sizeof(*probe));
    launch(nterms >= 2 && nterms <= 5 ? TFK_SWEEP_F_STAGE_RHS_N : TFK_SWEEP_F_STAGE_RHS, gx, gy, spec.sweep_block, &a, sizeof(a));
}

void tf_solver::spmv_stage(int nterms, const double* const* vx, const double* vc, const double* Fp, double cF, double cA, double* y) {
    TfSpmvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.L = L1; a.Jv = Jv.p; a.v = nullptr; a.y = y; a.scale = 1.0;
    a.parsca = parsca.p; a.dx = dx.p;
    a.nterms = nterms;
    for (int t = 0; t < nterms; ++t) { a.vx[t] = vx[t]; a.vc[t] = vc[t]; }
    a.addF = Fp; a.cF = cF; a.cA = cA;
    unsigned gx = sweep_gx(), gy = cdiv(L1.M, spec.seg);
    launch(TFK_SPMV, gx, gy, spec.sweep_block, &a, sizeof(a));
}

void tf_solver::mark_hooked(int slot, bool post) {
    if (slot < 0) return;
    if ((size_t)slot >= slot_hook.size()) slot_hook.resize((size_t)slot + 1);
    slot_hook[slot] = post ? dir_post_h : dir_h;
}

bool tf_solver::input_is_hooked(int src) const {
    return hook_in_place && ndir > 0 && src >= 0 && (size_t)src < slot_hook.size() &&
           !slot_hook[src].empty() && slot_hook[src] == dir_h;
}

const double* tf_solver::stage_input(int src, double* U) {
    if (ndir == 0 || input_is_hooked(src)) return st(src);
    copy(U, st(src), (size_t)vecn() * sizeof(double));
    apply_dirichlet(U);
    return U;
}
