// tf_solver: the banded solver -- factorisation, solves, accuracy guard, rescue on longer chunks (see tf_solver.h)
#include "tf_solver.h"

unsigned tf_solver::l1_twist_lds(TfLevelArgs& a) const {
    if (!(l1_respike && a.twist && l1_fuse_backsub && tfb::is_device_build())) return 0;
    auto half = [&](int mI) { return (spec.mp * spec.nvar <= 6 && mI >= 4 * spec.mp) ? (mI + 1) / 2 : mI; };
    const int mI_max = a.L.M - spec.mp;
    a.ylds_rows = std::max(half(mI_max), a.L.rem > 0 ? half(mI_max - 1) : 0);
    const size_t lds = (size_t)2 * a.ylds_rows * spec.nvar * 64 * sizeof(double);
    // (up to half of a CU's 160 KB: two workgroups = four wavefronts, one per SIMD; the stiff model's
    // 80 KB just fit -- config 5 687 -> 698 steps/s, profiles/r03_ab_runs.txt r3n)
    return lds <= 80u * 1024u ? (unsigned)lds : 0u;
}

TfLevelArgs tf_solver::level_args(size_t l, const double* rhs1, double* x1) {
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

TfTopArgs tf_solver::top_args() {
    TfTopArgs t;
    t.nsys = nsys; t.A = top.Ablk.p; t.rhs = top.rhs.p; t.Ainv = topAinv.p; t.x = top.x.p; t.status = status;
    t.aos = levels.back()->cr ? 1 : 0;
    return t;
}

void tf_solver::factor(double c, const double* rhs1, double* x1) {
    if (!have_jac) throw std::runtime_error("tf_factor: no Jacobian evaluated yet (call tf_eval with_j=1)");
    factor_c = c;
    delegated = false;
    if (const Checked* like0 = checked_like(c); like0 && like0->replan && can_replan()) {
        // this plan is known to break down for such a c: straight to the longer chunks
        ++n_factor;
        have_factor = true; cf_valid = jconst; cf_c = c; cf_ver = par_ver;
        check_now = false; fact_checked = true; sweeps_needed = 0;
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
    if (check_now) { fact_checked = false; fact_needs_refine = false; sweeps_needed = 0; }
    else if (like) { sweeps_needed = like->sweeps; fact_needs_refine = sweeps_needed > 0; }
    // (between checks the verdict of the last checked factorisation with such a c stands)
    if (rhs1 == nullptr) return;
    if (!fused) { solve(rhs1, x1); return; }
    if (!fold_top()) { TfTopArgs t = top_args(); launch(TFK_TOP_SOLVE, cdiv(nsys, 64), 1, 64, &t, sizeof(t)); }
    backsub_chain(rhs1, x1, fold_top() ? 1 : 0);
    polish(rhs1, x1);
}

void tf_solver::factor_step(double c, const double* rhs1, double* x1) {
    reused = reuse_ok(c);
    if (!reused && alt_ok(c)) { swap_slots(); reused = true; }
    else if (wants_alt(c)) { ensure_alt(); swap_slots(); }        // (allocated before any capture: prepare_step)
    if (!reused) { factor(c, rhs1, x1); return; }
    have_factor = true;                          // (the sweep of this step reset it)
    solve(rhs1, x1);
}

void tf_solver::request_update(double* out, const double* base, const double* k0, double c0, double c1, int n) {
    upd_req = Update{out, base, k0, c0, c1, n};
    upd_req_on = upd_fuse;
    upd_done = false;
}

bool tf_solver::update_allowed() const {
    return !delegated && !tiny && refine <= 0 && (refine != -1 || (fact_checked && sweeps_needed == 0));
}

void tf_solver::backsub_chain(const double* rhs1, double* x1, int skip) {
    if (skip == 1 && scalar_fused_ok()) {
        TfScalarArgs t = scalar_args(rhs1, x1);
        upd_req_on = false;
        launch(TFK_S_BWD, (unsigned)t.lv[1].L.Ptot, 1, 512, &t, sizeof(t));
        return;
    }
    // level 2 backwards inside the level-1 launch (tfk_l1_fwd2_backsub_cr) where that launch exists
    bool l2_in_l1 = false;
    if (l1cr_ok() && levels.size() - (size_t)skip >= 2) {
        TfLevelArgs a0 = level_args(0, rhs1, x1);
        // (its cyclic-reduction blocks are static LDS on top of y: where y alone takes a workgroup's half of
        // the CU -- config 5, 80 KB -- the launch would run one workgroup per CU: 810 -> 719 steps/s)
        const unsigned lds0 = l1_twist_lds(a0);
        l2_in_l1 = lds0 > 0 && lds0 <= 64u * 1024u;
    }
    for (size_t l = levels.size() - (size_t)skip; l-- > 0;) {
        if (l == 1 && l2_in_l1) continue;
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
                if (l2_in_l1) {
                    TfTailArgs t;
                    t.lv[0] = a;
                    t.lv[1] = level_args(1, rhs1, x1);
                    launch(TFK_L1_FWD2_BACKSUB_CR, (unsigned)nsys * cdiv(t.lv[1].L.P, 4), 1, 128, &t, sizeof(t), lds);
                    continue;
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

// 3 <= b <= 6 with at least two cyclic-reduction levels above level 2: level 1 and level 2 of a solve in one
// launch each way (tfk_l1_solve_cr, tfk_l1_fwd2_backsub_cr; the walks must assemble the separator rows)
bool tf_solver::l1cr_ok() const {
    return l1cr_fuse && tfb::is_device_build() && !tiny && top.B >= 3 && top.B <= 6 && levels.size() >= 4 &&
           levels[1]->cr && levels[2]->cr && fuse_asm_ok();
}

bool tf_solver::scalar_fused_ok() const {
    return s_fuse && tfb::is_device_build() && !tiny && top.B <= 2 && levels.size() == 3 && levels[1]->cr &&
           levels[2]->cr && levels[2]->L.P == 1 && fold_top() && fuse_asm_ok() && !l1_respike;
}

TfScalarArgs tf_solver::scalar_args(const double* rhs1, double* x1) {
    TfScalarArgs t;
    for (size_t l = 0; l < 3; ++l) t.lv[l] = level_args(l, rhs1, x1);
    if (!sfuse_counter) sfuse_counter = (unsigned*)tfb::dev_alloc((size_t)nsys * sizeof(unsigned));
    t.counter = sfuse_counter;
    return t;
}

bool tf_solver::tail_ok() const {
    const size_t n = levels.size();
    return cr_tail && tfb::is_device_build() && n >= 3 && top.B >= 3 && top.B <= 6 && levels[n - 1]->cr && levels[n - 2]->cr &&
           levels[n - 1]->L.P == 1 && levels[n - 2]->L.P <= 8;      // TF_CR_TAIL_MAXB, TF_CR_TAIL_WAVES
}

void tf_solver::solve_once(const double* rhs1, double* x1) {
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
        if (l == 0 && l1cr_ok()) {
            TfTailArgs t;
            t.lv[0] = level_args(0, rhs1, x1);
            t.lv[1] = level_args(1, rhs1, x1);
            launch(TFK_L1_SOLVE_CR, (unsigned)nsys * cdiv(t.lv[1].L.P, 4), 1, 128, &t, sizeof(t));
            l = 1;                                    // (level 2's forward elimination went with it)
            continue;
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

void tf_solver::refine_sweep(const double* rhs1, double* x1) {
    spmv(x1, Wjv.p, factor_c);                               // c J x
    const double* xs[3] = {rhs1, x1, Wjv.p};
    vec(TF_VEC_RESID, Wres.p, nullptr, 3, xs, nullptr);      // r = (b - x) + c J x
    solve_once(Wres.p, Wdel.p);
    const double* ys[2] = {x1, Wdel.p};
    vec(TF_VEC_ADD, x1, nullptr, 2, ys, nullptr);
}

double tf_solver::backward_error(const double* rhs1, const double* x1) {
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

bool tf_solver::sampled_monitor_due() const {
    return refine < 0 && !reused && !delegated && !tiny && have_factor && !(refine == -1 && check_now);
}

TfBerrArgs tf_solver::probe_args(const double* rhs1, const double* x1, const double* xbase) {
    TfBerrArgs a;
    std::memset(&a, 0, sizeof(a));
    a.L = L1; a.Jv = Jv.p; a.x = x1; a.rhs = rhs1; a.c = factor_c; a.red = red.p + 4;
    a.parsca = parsca.p; a.dx = dx.p; a.xbase = xbase;
    // (a stride coprime with every chunk length up to 64: consecutive steps look at nodes far apart.  Looking
    // at a quarter of the chunks per step on very many chunks was tried for config 5: nothing, its `tfk_berr`
    // time is the full passes of the synchronising checks, profiles/r04_ab_runs.txt)
    a.one_node = (int)((mon_phase++ * 13u) % 4096u);
    monitored = true;
    return a;
}

void tf_solver::monitor_sampled(const double* rhs1, const double* x1, const double* xbase) {
    TfBerrArgs a = probe_args(rhs1, x1, xbase);
    launch(TFK_BERR, sweep_gx(), 1, spec.sweep_block, &a, sizeof(a));
}

void tf_solver::solve(const double* rhs1, double* x1) {
    if (!have_factor) throw std::runtime_error("tf_solve: matrix not factorised");
    if (delegated) { delegate_solve(rhs1, x1); return; }
    solve_once(rhs1, x1);
    polish(rhs1, x1);
}

void tf_solver::transfer_to(tf_solver* dst, const double* src_planes, double* dst_planes, int ncomp) {
    ensure_staging((size_t)ncomp * nsys * N);
    perm(1 /*OUT_SOA*/, src_planes, staging.p, ncomp);
    dst->perm(0 /*IN_SOA*/, staging.p, dst_planes, ncomp);
}

void tf_solver::delegate_factor(double c) {
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

void tf_solver::delegate_solve(const double* rhs1, double* x1) {
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

void tf_solver::polish(const double* rhs1, double* x1) {
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

void tf_solver::check_status(const int* have_flag, const double* have_worst) {
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
