// Host runtime of libtriflow_hip: inputs and outputs of a solver (states, parameters, hooks, F / J) and the
// linear-solver entry points tf_factor / tf_solve / tf_matvec (seams #1 and #3 of include/triflow_hip.h)
#include "tf_solver.h"

extern "C" {

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

}  // extern "C"
