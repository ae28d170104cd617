// TEST-ONLY emulation of tf_backend.h: executes the kernel bodies of
// triflow_amd/csrc/tf_kernels.h thread by thread on the host, so that the host
// runtime (level planning, launch order, scheme drivers) and the kernel
// arithmetic can be checked against the oracle in the CPU test suite.  Built
// per model by tests/emu/build_emu.py into tests/emu/_build/; never linked into
// libtriflow_hip.so and never imported by the triflow_amd package.
#include "tf_backend.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#define TF_DEVICE static inline
#define TF_DEVICE_M inline
#include "tf_math.h"
using std::sqrt; using std::exp; using std::log; using std::sin; using std::cos; using std::tan;
using std::tanh; using std::sinh; using std::cosh; using std::pow; using std::atan; using std::asin;
using std::acos; using std::log10; using std::log2; using std::cbrt; using std::expm1; using std::log1p;
using std::floor; using std::ceil;
#include TF_EMU_MODEL_HEADER
#include "tf_kernels.h"
#include "tf_crs.h"
#include <string>

namespace tfb {

struct Module { int dummy; };
struct Stream { int dummy; };
struct Event { int dummy; };

bool is_device_build() { return false; }
int coop_group(int) { return 1; }
// The cyclic-reduction levels run here with the one-thread-per-node kernels of tf_crs.h, one
// "thread" taking every node in turn (the wave-cooperative kernels for 3 <= b <= 8 are HIP only):
// record formats, two-part right-hand sides, folded top block, the walks' assembled separator rows.
// TRIFLOW_REDUCED=walk: chunk walks on the reduced levels, as on the GPU.
bool cyclic_reduction(int b) {
    const char* mode = std::getenv("TRIFLOW_REDUCED");
    if (mode && std::string(mode) == "walk") return false;
    return b >= 1 && b <= 8;
}
int device_count() { return 0; }
void set_device(int) {}
void* dev_alloc(size_t bytes) { void* p = std::calloc(bytes ? bytes : 8, 1); if (!p) throw std::bad_alloc(); return p; }
void dev_free(void* p) { std::free(p); }
void memset0(void* p, size_t bytes, Stream*) { std::memset(p, 0, bytes); }
void h2d(void* d, const void* s, size_t n, Stream*) { std::memcpy(d, s, n); }
void d2h(void* d, const void* s, size_t n, Stream*) { std::memcpy(d, s, n); }
void d2d(void* d, const void* s, size_t n, Stream*) { std::memmove(d, s, n); }
Module* module_load(const void*, size_t) { return new Module(); }
void module_add_alternate(Module*, const void*, size_t, uint64_t) {}
void module_unload(Module* m) { delete m; }
Stream* stream_create() { return new Stream(); }
void stream_destroy(Stream* s) { delete s; }
void stream_sync(Stream*) {}
struct Graph { int dummy; };
bool graphs_supported() { return false; }
void capture_begin(Stream*) {}
Graph* capture_end(Stream*) { return nullptr; }
void capture_abort(Stream*) {}
void graph_launch(Graph*, Stream*) {}
void graph_destroy(Graph*) {}
struct Mailbox { char host[64]; };
Mailbox* mailbox_create() { return new Mailbox(); }
void mailbox_destroy(Mailbox* m) { delete m; }
void mailbox_post(Mailbox* m, int at, const void* dev_src, size_t bytes, Stream*) { std::memcpy(m->host + at, dev_src, bytes); }
void mailbox_mark(Mailbox*, Stream*) {}
void mailbox_wait(Mailbox* m, void* dst, size_t bytes) { std::memcpy(dst, m->host, bytes); }
Event* event_create() { return new Event(); }
void event_destroy(Event* e) { delete e; }
void event_record(Event*, Stream*) {}
float event_elapsed_ms(Event*, Event*) { return 0.f; }

typedef TfRowsBT<TF_B2> TfRowsUp;

unsigned kernel_block(Module*, int) { return 64; }     // (the walks run in their one-wavefront form here)
void launch(Module*, int kernel, unsigned gx, unsigned gy, unsigned block,
            const void* args, size_t, Stream*, unsigned) {
    const int64_t nthreads = (int64_t)gx * block;
    switch (kernel) {
    case TFK_SWEEP_F: { const auto& a = *(const TfSweepArgs*)args;
        for (unsigned y = 0; y < gy; ++y) for (int64_t t = 0; t < nthreads; ++t) tfk_sweep_body<false>(a, (int)t, (int)y); } break;
    case TFK_SWEEP_F_STAGE: { const auto& a = *(const TfSweepArgs*)args;
        for (unsigned y = 0; y < gy; ++y) for (int64_t t = 0; t < nthreads; ++t) tfk_sweep_body<false, true>(a, (int)t, (int)y); } break;
    case TFK_SWEEP_F_STAGE_RHS_N:
    case TFK_SWEEP_F_STAGE_RHS: { const auto& a = *(const TfSweepArgs*)args;
        for (unsigned y = 0; y < gy; ++y) for (int64_t t = 0; t < nthreads; ++t) tfk_sweep_body<false, true, false, false, true, TF_STAGE_SEG>(a, (int)t, (int)y); } break;
    case TFK_SWEEP_FJ_BDF2: { const auto& a = *(const TfSweepArgs*)args;
        for (unsigned y = 0; y < gy; ++y) for (int64_t t = 0; t < nthreads; ++t) tfk_sweep_body<true, false, false, true>(a, (int)t, (int)y); } break;
    case TFK_SWEEP_FJ_THETA: { const auto& a = *(const TfSweepArgs*)args;
        for (unsigned y = 0; y < gy; ++y) for (int64_t t = 0; t < nthreads; ++t) tfk_sweep_body<true, false, true>(a, (int)t, (int)y); } break;
    case TFK_SWEEP_FJ: { const auto& a = *(const TfSweepArgs*)args;
        for (unsigned y = 0; y < gy; ++y) for (int64_t t = 0; t < nthreads; ++t) tfk_sweep_body<true>(a, (int)t, (int)y); } break;
    case TFK_SPMV: { const auto& a = *(const TfSpmvArgs*)args;
        for (unsigned y = 0; y < gy; ++y) for (int64_t t = 0; t < nthreads; ++t) tfk_spmv_body(a, (int)t, (int)y); } break;
    case TFK_VEC: { const auto& a = *(const TfVecArgs*)args;
        for (int64_t i = 0; i < a.n; ++i) tfk_vec_elem(a, i); } break;
    case TFK_VEC_MAXABS: { const auto& a = *(const TfVecArgs*)args;
        double m = *a.red;
        for (int64_t i = 0; i < a.n; ++i) { double v = a.op == TF_VEC_MAXRATIO ? tf_vec_ratio(a, i) : (a.op == TF_VEC_SUM_ERR ? tf_vec_sum_err(a, i) : tf_vec_err(a, i)); m = (v > m || v != v) ? v : m; }
        *a.red = m; } break;
    case TFK_SWEEP_F_STAGE_RHS_MON: { const auto& a = *(const TfStageMonArgs*)args;
        for (unsigned y = 0; y + 1 < gy; ++y) for (int64_t t = 0; t < nthreads; ++t) tfk_sweep_body<false, true, false, false, true, TF_STAGE_SEG>(a.s, (int)t, (int)y);
        double m = *a.b.red;
        for (int64_t t = 0; t < nthreads; ++t) { double v = tfk_berr_body(a.b, (int)t, 0); m = (v > m || v != v) ? v : m; }
        *a.b.red = m; } break;
    case TFK_BERR: { const auto& a = *(const TfBerrArgs*)args;
        double m = *a.red;
        for (unsigned y = 0; y < gy; ++y) for (int64_t t = 0; t < nthreads; ++t) {
            double v = tfk_berr_body(a, (int)t, (int)y); m = (v > m || v != v) ? v : m; }
        *a.red = m; } break;
    case TFK_DIFFNORM: { const auto& a = *(const TfNormArgs*)args;
        for (unsigned y = 0; y < gy; ++y) for (unsigned b = 0; b < gx; ++b)
            a.partial[(int64_t)y * a.nblocks + b] = tfk_diffnorm_partial(a, (int)y, (int)b, 0, 1);
        if (a.status) {
            double* tail = a.partial + (int64_t)gy * a.nblocks;
            unsigned long long bits = (unsigned)*a.status;
            std::memcpy(&tail[0], &bits, sizeof(double));
            tail[1] = *a.mon;
        } } break;
    case TFK_PERM: { const auto& a = *(const TfPermArgs*)args;
        for (int64_t t = 0; t < nthreads; ++t) tfk_perm_elem(a, t); } break;
    case TFK_GATHER: { const auto& a = *(const TfGatherArgs*)args;
        for (int64_t t = 0; t < a.n; ++t) tfk_gather_elem(a, t); } break;
    case TFK_POKE: { const auto& a = *(const TfPokeArgs*)args;
        for (int64_t t = 0; t < nthreads; ++t) tfk_poke_elem(a, (int)t); } break;
    case TFK_DIRICHLET: { const auto& a = *(const TfDirichletArgs*)args;
        for (int64_t t = 0; t < nthreads; ++t) tfk_dirichlet_elem(a, (int)t); } break;
// (a.fuse_asm: the walk leaves its half of a separator's record in `stage`, written out here)
#define TF_EMU_CHUNK(ID, ROWS, SP, SU, SY)                                              \
    case ID: { const auto& a = *(const TfLevelArgs*)args;                               \
        constexpr int HALF = 2 * TF_B2 * TF_B2;                                          \
        for (int dir = 0; dir < 2; ++dir)                                                \
            for (int64_t t = 0; t < nthreads; ++t) {                                     \
                double stage[HALF];                                                      \
                int rec = -1;                                                            \
                if (dir == 0) tfk_chunk_body<ROWS, +1, SP, SU, SY>(a, (int)t, nullptr, stage, &rec); \
                else tfk_chunk_body<ROWS, -1, SP, SU && TF_RESPIKE_MODEL(TF_MP, TF_NVAR), false>(a, (int)t, nullptr, stage, &rec); \
                if (SP && a.fuse_asm && rec >= 0)                                        \
                    for (int i = 0; i < HALF; ++i) a.Anext[(int64_t)rec * 2 * HALF + dir * HALF + i] = stage[i]; \
            } } break;
    TF_EMU_CHUNK(TFK_L1_FACTOR, TfRowsL1, true, true, false)
    TF_EMU_CHUNK(TFK_L1_SOLVE, TfRowsL1, false, false, true)
    TF_EMU_CHUNK(TFK_L1_FACTOR_RHS, TfRowsL1, true, true, true)
    case TFK_L1_FWD2: { const auto& a = *(const TfLevelArgs*)args;
        if constexpr (TF_RESPIKE_MODEL(TF_MP, TF_NVAR)) {
            for (int64_t t = 0; t < nthreads; ++t) tfk_chunk_body<TfRowsL1, +1, false, false, true, true>(a, (int)t);
            for (int64_t t = 0; t < nthreads; ++t) tfk_chunk_body<TfRowsL1, -1, false, false, true, true>(a, (int)t);
        } } break;
    case TFK_BT_LU: { const auto& a = *(const TfLevelArgs*)args;
        for (unsigned y = 0; y < gy; ++y) for (int64_t t = 0; t < nthreads; ++t)
            tfk_bt_lu_body<TF_B2>(a, (int)t, y == 0 ? +1 : -1); } break;
    case TFK_BT_SPIKE: { const auto& a = *(const TfLevelArgs*)args;
        for (unsigned y = 0; y < gy; ++y) for (int64_t t = 0; t < nthreads; ++t)
            tfk_bt_col_body<TF_B2>(a, (int)t, (y & 1) == 0 ? +1 : -1, (int)(y >> 1)); } break;
    case TFK_BT_RHS: { const auto& a = *(const TfLevelArgs*)args;
        for (unsigned y = 0; y < gy; ++y) for (int64_t t = 0; t < nthreads; ++t)
            tfk_bt_col_body<TF_B2>(a, (int)t, y == 0 ? +1 : -1, TF_B2); } break;
#define TF_EMU_LEVEL(ID, CALL)                                                          \
    case ID: { const auto& a = *(const TfLevelArgs*)args;                               \
        for (int64_t t = 0; t < nthreads; ++t) CALL(a, (int)t); } break;
    TF_EMU_LEVEL(TFK_L1_ASM_MAT, (tfk_asm_body<TfRowsL1, true>))
    TF_EMU_LEVEL(TFK_L1_ASM_RHS, (tfk_asm_body<TfRowsL1, false>))
    TF_EMU_LEVEL(TFK_L1_BACKSUB, (tfk_backsub_body<TfRowsL1, true>))
    case TFK_L1_BACKSUB_U: { const auto& a = *(const TfLevelArgs*)args;
        if constexpr (TF_RESPIKE_MODEL(TF_MP, TF_NVAR)) {
            for (int dir = 0; dir < 2; ++dir)
                for (int64_t t = 0; t < nthreads; ++t) tfk_backsub_twist_body<TfRowsL1>(a, (int)t, dir);
        } } break;
    TF_EMU_LEVEL(TFK_BT_ASM_MAT, (tfk_asm_body<TfRowsUp, true>))
    TF_EMU_LEVEL(TFK_BT_ASM_RHS, (tfk_asm_body<TfRowsUp, false>))
    TF_EMU_LEVEL(TFK_BT_BACKSUB, (tfk_backsub_body<TfRowsUp>))
    case TFK_TOP_FACTOR: { const auto& a = *(const TfTopArgs*)args;
        for (int64_t t = 0; t < nthreads; ++t) tfk_top_body<TF_B2, true>(a, (int)t); } break;
    case TFK_TOP_SOLVE: { const auto& a = *(const TfTopArgs*)args;
        for (int64_t t = 0; t < nthreads; ++t) tfk_top_body<TF_B2, false>(a, (int)t); } break;
    case TFK_CR_FACTOR: { const auto& a = *(const TfLevelArgs*)args;
        if constexpr (TF_B2 <= 8) for (unsigned c = 0; c < gx; ++c) tfk_crs_factor<TF_B2, 1>(a, (int)c, 0); } break;
    case TFK_CR_FWD: { const auto& a = *(const TfLevelArgs*)args;
        if constexpr (TF_B2 <= 8) for (unsigned c = 0; c < gx; ++c) tfk_crs_fwd<TF_B2, 1>(a, (int)c, 0); } break;
    case TFK_CR_BWD: { const auto& a = *(const TfLevelArgs*)args;
        if constexpr (TF_B2 <= 8) for (unsigned c = 0; c < gx; ++c) tfk_crs_bwd<TF_B2, 1>(a, (int)c, 0); } break;
    case TFK_TINY_FACTOR: { const auto& a = *(const TfTinyArgs*)args;
        for (int64_t t = 0; t < nthreads; ++t) tfk_tiny_factor_body(a, (int)t); } break;
    case TFK_TINY_SOLVE: { const auto& a = *(const TfTinyArgs*)args;
        for (int64_t t = 0; t < nthreads; ++t) tfk_tiny_solve_body(a, (int)t); } break;
    default: throw std::runtime_error("emu: unknown kernel");
    }
}

void launch_timed(Module* m, int kernel, unsigned gx, unsigned gy, unsigned block,
                  const void* args, size_t n, Stream* s, Event*, Event*, unsigned) {
    launch(m, kernel, gx, gy, block, args, n, s, 0);
}

}  // namespace tfb
