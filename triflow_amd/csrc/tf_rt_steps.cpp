// Host runtime of libtriflow_hip: the time-step drivers (seam #2 of include/triflow_hip.h)
#include "tf_solver.h"

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
              const double* gamma, const double* b, const double* b_pred, bool hook_after, bool want_err,
              int err_slot = 0) {
    require(ns >= 1 && ns <= 6, "tf_step_row: 1 <= s <= 6");
    require(src != dst, "tf_step_row: src and dst slots must differ");
    double* U = s->st(dst);
    s->slot_written(dst);
    const double* Uin = s->stage_input(src, U);
    s->sweep(Uin, true, 0, nullptr, nullptr, dt);   // J(U) and dt*F(U): right-hand side of stage 0
    const double* ks[TF_MAX_TERMS];
    double cs[TF_MAX_TERMS];
    bool probe_due = false;
    for (int i = 0; i < ns; ++i) {
        if (i > 0) {
            // F(U + sum_j alpha_ij k_j): the stage state is formed inside the sweep.  It goes to a
            // buffer of its own (F keeps dt*F(U), the right-hand side of stage 0)
            // ... plus dt*(J @ sum_j gamma_ij k_j), in the same pass
            double gs[TF_MAX_TERMS];
            for (int j = 0; j < i; ++j) { ks[j] = s->K[j].p; cs[j] = alpha[i * ns + j]; gs[j] = gamma[i * ns + j]; }
            // (i == 1: with the backward-error probe of the stage-0 solve, (I - gamma dt J) k0 = dt F(U), at one
            // node per chunk -- every step between the synchronising checks, nobody waits for it)
            if (i == 1 && probe_due) {
                const TfBerrArgs probe = s->probe_args(s->F.p, s->K[0].p, nullptr);
                s->stage_rhs(Uin, i, ks, cs, gs, dt, s->Wrhs.p, &probe);
            } else s->stage_rhs(Uin, i, ks, cs, gs, dt, s->Wrhs.p);
        }
        // the last stage of a fixed step of one or two stages: the new state leaves with the solve
        if (i == ns - 1 && ns <= 2 && !(b_pred && want_err))
            s->request_update(U, Uin, ns == 2 ? s->K[0].p : nullptr, b[0], ns == 2 ? b[1] : 0.0, ns);
        if (i == 0) {
            s->factor_step(gamma[0] * dt, s->F.p, s->K[0].p);         // factorise + first stage
            probe_due = ns > 1 && s->sampled_monitor_due();
        } else s->solve(s->Wrhs.p, s->K[i].p);
    }
    for (int j = 0; j < ns; ++j) { ks[j] = s->K[j].p; cs[j] = b[j]; }
    const bool updated = s->take_update_done();
    if (b_pred && want_err) {
        // the new state U + sum_i b_i k_i and ||U+ - (U+ + sum b_pred k)||_inf: one pass over the stages
        // (an adaptive step never has its update inside the back-substitution: request_update above)
        s->zero(s->red.p + err_slot, sizeof(double));
        if (!updated) s->vec(TF_VEC_SUM_ERR, U, Uin, ns, ks, cs, -1, err_slot, b_pred);
        else s->vec(TF_VEC_MAXABS, nullptr, U, ns, ks, b_pred, -1, err_slot);
    } else if (!updated) {
        s->vec(TF_VEC_SUM, U, Uin, ns, ks, cs);                    // U + sum_i b_i k_i
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
        s->slot_key(gamma[0] * dt);
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

// The same step with its embedded error estimate left on the device, in reduction slot `err_slot`
// (1 ... 3: slot 0 belongs to the blocking forms), and read later by tf_read_err: a driver that has the
// next step queued before it looks at this one's estimate keeps the GPU busy while it decides
// (schemes.ROW_general: the accepted trial of a call and the first trial of the next one).
int tf_step_row_queued(tf_solver* s, int32_t src, int32_t dst, double dt, int32_t ns,
                       const double* alpha, const double* gamma, const double* b,
                       const double* b_pred, int32_t hook_after, int32_t err_slot) {
    TF_API_BEGIN
    require(s && alpha && gamma && b && b_pred, "null argument");
    require(ns >= 1 && ns <= 6, "tf_step_row_queued: 1 <= s <= 6");
    require(err_slot >= 1 && err_slot <= 3, "tf_step_row_queued: err_slot is 1, 2 or 3");
    s->prepare_step(gamma[0] * dt);
    step_row(s, src, dst, dt, ns, alpha, gamma, b, b_pred, hook_after != 0, true, err_slot);
    if (hook_after) s->mark_hooked(dst); else s->slot_written(dst);
    // the estimate and the failure flag as they stand behind this step go to a page-locked mailbox:
    // tf_read_err waits for this spot of the stream, not for what the caller queues after it
    if (!s->err_box[err_slot]) s->err_box[err_slot] = tfb::mailbox_create();
    tfb::mailbox_post(s->err_box[err_slot], 0, s->red.p + err_slot, sizeof(double), s->stream);
    tfb::mailbox_post(s->err_box[err_slot], 8, s->status, sizeof(int), s->stream);
    tfb::mailbox_mark(s->err_box[err_slot], s->stream);
    TF_API_END
}
int tf_read_err(tf_solver* s, int32_t err_slot, double* err_out) {
    TF_API_BEGIN
    require(s && err_out, "null argument");
    require(err_slot >= 1 && err_slot <= 3 && s->err_box[err_slot], "tf_read_err: no step was queued with this err_slot");
    char buf[16];
    tfb::mailbox_wait(s->err_box[err_slot], buf, sizeof(buf));
    std::memcpy(err_out, buf, sizeof(double));
    int flag = 0;
    std::memcpy(&flag, buf + 8, sizeof(int));
    if (flag != 0) s->check_status();              // (reads the flag again behind everything queued, resets it, raises)
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

}  // extern "C"
