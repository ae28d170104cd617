// Host runtime of libtriflow_hip: counters, monitors, diagnostic stamps, per-kernel timing, tf_sync
#include "tf_solver.h"

extern "C" {

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
