// Hand-written CDNA4 kernel skeletons of the triflow hot path.
//
// This header is included by a small per-model translation unit that the HIP
// compiler plugin (triflow_amd/compilers.py) generates at Model.compile time:
// the unit defines the model constants (TF_NVAR, TF_MP, the Jacobian sparsity
// pattern, ...) and the two per-node stencil bodies tf_eval_F / tf_eval_J that
// were lowered from the SymPy expressions; everything else -- data layout,
// neighbourhood handling, store pattern, the banded solver -- is fixed code
// below.  It replaces (reference file:line)
//   * compute_F_numpy / compute_J_numpy / init_computation_numpy
//       triflow/core/compilers.py:227-332            -> tfk_sweep
//   * CSC `J @ v`                 triflow/core/schemes.py:157,553 -> tfk_spmv
//   * scipy.sparse.linalg.spsolve / factorized (SuperLU)
//       triflow/core/schemes.py:149,557               -> tfk_chunk_* / tfk_asm_* / tfk_backsub / tfk_top_*
//   * the NumPy vector algebra of the schemes
//       triflow/core/schemes.py:152-174,553           -> tfk_vec
//
// Layout: see TfLayout in tf_args.h.  All loads and stores of the big arrays are
// 8 bytes per lane, 512 contiguous bytes per wavefront instruction; a thread
// walks along its chunk of consecutive nodes and keeps the (2*mp+1)-wide stencil
// neighbourhood of every field in registers, so each field value is read from
// HBM once per sweep and ghost cells only exist at chunk ends (served by the
// neighbouring lane's column, i.e. the same cache lines).
//
// The same source compiles for the host (TF_DEVICE empty) in the test-only
// emulation build under tests/emu/, which runs these very functions thread by
// thread on the CPU; the product library never contains that build.
#pragma once
#include "tf_args.h"

#ifndef TF_DEVICE
#error "define TF_DEVICE (e.g. __device__ __forceinline__) before including tf_kernels.h"
#endif

#define TF_W (2 * TF_MP + 1)
#define TF_NF (TF_NVAR + TF_NH)
#define TF_B2 (TF_MP * TF_NVAR)          // block size of the reduced (interface) systems

#include "tf_math.h"

// In-kernel stamps of a diagnostic build (-DTF_STAMPS, tools/gpu_stamps.py): lane 0 of the
// middle workgroup records the shader clock at phase boundaries.  No stamp executes in the
// product build, and nothing computed ever depends on one.
#if defined(TF_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define TF_STAMP(a, i) do { if ((a).stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) \
        (a).stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define TF_STAMP_T(a, i, tid) do { if ((a).stamps && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && threadIdx.x == (tid)) \
        (a).stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define TF_STAMP_REAL(a, i) do { if ((a).stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) \
        (a).stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
// ... by whichever workgroup satisfies `cond` (the last arriver of a hand-off)
#define TF_STAMP_IF(a, i, cond) do { if ((a).stamps && (cond) && threadIdx.x == 0) (a).stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define TF_STAMP_REAL_IF(a, i, cond) do { if ((a).stamps && (cond) && threadIdx.x == 0) (a).stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
// ... and event counters (every workgroup; one lane per wavefront counts)
#define TF_COUNT(a, i) do { if ((a).stamps && (threadIdx.x & 63) == 0) atomicAdd(&(a).stamps[i], 1ull); } while (0)
// ... and begin / end times (100 MHz) of every workgroup of a level-1 kernel: region k of 2048
// entries behind the 8 level regions (tools/gpu_wgtrace.py asks for that many)
// (level-1 walks: the down walk of the middle workgroup; slots 0.. re-elimination, 10.. factorisation,
// 20.. first solve walk)
#define TF_STAMP_L1(a, dir, i) do { if ((dir) > 0 && blockIdx.y == 0) TF_STAMP(a, i); } while (0)
#define TF_WGTRACE(a, k, which) do { if ((a).stamps && threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 1024) \
        (a).stamps[512 + 2048 * (k) + 2 * (blockIdx.y * gridDim.x + blockIdx.x) + (which)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TF_STAMP_L1(a, dir, i) do {} while (0)
#define TF_WGTRACE(a, k, which) do {} while (0)
#define TF_STAMP(a, i) do {} while (0)
#define TF_STAMP_T(a, i, tid) do {} while (0)
#define TF_STAMP_REAL(a, i) do {} while (0)
#define TF_STAMP_IF(a, i, cond) do {} while (0)
#define TF_STAMP_REAL_IF(a, i, cond) do {} while (0)
#define TF_COUNT(a, i) do {} while (0)
#endif


// workgroup barrier of the kernels whose wavefronts work together (the emulation runs the
// one-wavefront forms of those: nothing to wait for)
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define TF_WG_BARRIER() __syncthreads()
#else
#define TF_WG_BARRIER() do {} while (0)
#endif

// ------------------------------------------------------------------- layout
TF_DEVICE int tf_len(const TfLayout& L, int p) { return L.mbase + (p < L.rem ? 1 : 0); }
TF_DEVICE int tf_start(const TfLayout& L, int p) { return p * L.mbase + (p < L.rem ? p : L.rem); }
TF_DEVICE int64_t tf_idx(const TfLayout& L, int pg, int i) { return (int64_t)i * L.Ptot + pg; }
// Element of plane k at byte offset off8 = 8 * tf_idx(...): the plane's base stays in scalar
// registers and the lane offset is 32 bits wide (global_load/store with saddr: no 64-bit address
// arithmetic per access).  A plane is smaller than 4 GB (tf_solver_create checks).
TF_DEVICE unsigned tf_off8(const TfLayout& L, int pg, int i) { return (unsigned)tf_idx(L, pg, i) * 8u; }
TF_DEVICE double tf_ldp(const double* base, int64_t k, int64_t plane, unsigned off8) {
    return *(const double*)((const char*)(base + k * plane) + off8);
}
TF_DEVICE void tf_stp(double* base, int64_t k, int64_t plane, unsigned off8, double v) {
    *(double*)((char*)(base + k * plane) + off8) = v;
}

// element of the node `d` places after node i of chunk (e, p); wraps or clamps
// at the ends of the system exactly like the ghost cells of
// triflow/core/compilers.py:257-264.  Needs |d| <= mbase.
TF_DEVICE int64_t tf_nbr(const TfLayout& L, int e, int p, int len, int i, int d) {
    int ii = i + d;
    int pp = p;
    if (ii < 0) {
        if (p > 0) { pp = p - 1; ii += tf_len(L, pp); }
        else if (L.periodic) { pp = L.P - 1; ii += tf_len(L, pp); }
        else ii = 0;
    } else if (ii >= len) {
        if (p < L.P - 1) { pp = p + 1; ii -= len; }
        else if (L.periodic) { pp = 0; ii -= len; }
        else ii = len - 1;
    }
    return tf_idx(L, e * L.P + pp, ii);
}

// node index -> (chunk, row)
TF_DEVICE void tf_locate(const TfLayout& L, int g, int& p, int& i) {
    int big = L.rem * (L.mbase + 1);
    if (g < big) { p = g / (L.mbase + 1); i = g - p * (L.mbase + 1); }
    else { int h = g - big; p = L.rem + h / L.mbase; i = h - (h / L.mbase) * L.mbase; }
}

// element addresses in the NEXT solver level, which is stored either as
// partition-interleaved planes (s2 = tf_idx of the separator there) or, below
// cyclic-reduction levels, as one record per node in natural order (TfLevelArgs)
TF_DEVICE int64_t tf_next_A(const TfLevelArgs& a, int e, int p, int64_t s2, int blk, int rr, int cc, int bn) {
    return a.next_aos ? (((int64_t)(e * a.Lnext.N + p) * 4 + blk) * bn + rr) * bn + cc
                      : (int64_t)((blk * bn + rr) * bn + cc) * a.Lnext.plane + s2;
}
TF_DEVICE int64_t tf_next_rhs(const TfLevelArgs& a, int e, int p, int64_t s2, int k, int bn) {
    return a.next_aos ? (int64_t)(e * a.Lnext.N + p) * 2 * bn + k : (int64_t)k * a.Lnext.plane + s2;
}
TF_DEVICE int64_t tf_next_x(const TfLevelArgs& a, int e, int p, int64_t s2, int k, int bn) {
    return a.next_aos ? (int64_t)(e * a.Lnext.N + p) * bn + k : (int64_t)k * a.Lnext.plane + s2;
}

// Jacobian entries that are the same at every node of a system (tf_j_uniform: constant
// coefficients such as k/dx**2): the sweep stores them like every other entry of the
// reference's value table, but the kernels that read the table back -- the solver walks, J @ v,
// the backward-error monitor -- evaluate them once per thread from the scalar parameters,
// with the very expressions of tf_eval_J (same bits), and only load the others.  Film model:
// 10 of 19 entries are loaded, stiff model 11 of 24, scalar linear models none.
#define TF_JU(k) (tf_j_uniform[k])
// ... and entries that are an exact power-of-two multiple of another node-dependent entry
// (tf_j_alias / tf_j_alias_scale, codegen._proportional_entries: same bits as the stored value):
// one load serves both.  Film model: 7 planes are loaded instead of 10.
#define TF_JA(k) (!TF_JU(k) && tf_j_alias[k] >= 0)
struct TfJUniform {
    double v[TF_NNZ > 0 ? TF_NNZ : 1];
    TF_DEVICE_M void init(const double* parsca, const double* dxp, int nsys, int e) {
        double par[TF_NPAR > 0 ? TF_NPAR : 1];
#pragma unroll
        for (int k = 0; k < TF_NPAR; ++k) par[k] = tf_par_is_vec[k] ? 0.0 : parsca[k * nsys + e];
        tf_eval_J_uniform(par, dxp[e], v);
    }
    // the value table row of a node: ld(k) loads entry k; evaluated / scaled entries are not loaded
    template <class Ld>
    TF_DEVICE_M void row(Ld ld, double (&jr)[TF_NNZ > 0 ? TF_NNZ : 1]) const {
#pragma unroll
        for (int k = 0; k < TF_NNZ; ++k)
            jr[k] = TF_JU(k) ? v[k]
                  : (TF_JA(k) ? tf_j_alias_scale[k] * jr[tf_j_alias[k] >= 0 ? tf_j_alias[k] : 0] : ld(k));
    }
};

// ===========================================================================
// 1. F / F+J stencil sweep                       (compilers.py:227-332)
// ===========================================================================
// grid: x over chunks (all systems), y over segments of TF_SEG nodes.
// NTERMS > 0 (stage forms): the number of stage vectors k_j is a compile-time constant, so that the
// loads of a node's k_j are all issued before the first is used (with the run-time loop they went one
// after the other: the stage pass of RODASPR's later stages ran at 2.8 TB/s, profiles/r04_ab_runs.txt)
template <bool WITH_J, bool STAGE = false, bool THETA = false, bool BDF = false, bool STAGE_RHS = false,
          int SEG = TF_SEG, int NTERMS = 0>
TF_DEVICE void tfk_sweep_body(const TfSweepArgs& a, int pg, int seg) {
    const TfLayout& L = a.L;
    if (pg >= L.Ptot) return;
    const int e = pg / L.P, p = pg - e * L.P;
    const int len = tf_len(L, p);
    const int i0 = seg * SEG;
    if (i0 >= len) return;

    double par[TF_NPAR > 0 ? TF_NPAR : 1];
#pragma unroll
    for (int k = 0; k < TF_NPAR; ++k)
        if (!tf_par_is_vec[k]) par[k] = a.parsca[k * L.nsys + e];
    const double dx = a.dx[e];

    auto ld = [&](int f, int ii) -> double {
        const int64_t s = (ii >= 0 && ii < len) ? tf_idx(L, pg, ii) : tf_nbr(L, e, p, len, 0, ii);
        if (f >= TF_NVAR) return a.helpers[(int64_t)(f - TF_NVAR) * L.plane + s];
        const double u = a.fields[(int64_t)f * L.plane + s];
        if (!STAGE) return u;
        double acc = a.kc[0] * a.kx[0][(int64_t)f * L.plane + s];      // U + sum_j alpha_ij k_j
        for (int t = 1; t < a.nterms; ++t) acc = acc + a.kc[t] * a.kx[t][(int64_t)f * L.plane + s];
        return u + acc;
    };
    // STAGE_RHS: one load of every k_j serves the stage state and v = sum_j gamma_ij k_j
    auto ld2 = [&](int f, int ii, double& v) -> double {
        const int64_t q = (int64_t)f * L.plane +
            ((ii >= 0 && ii < len) ? tf_idx(L, pg, ii) : tf_nbr(L, e, p, len, 0, ii));
        const double u = a.fields[q];
        if constexpr (NTERMS > 0) {
            double kk[NTERMS];
#pragma unroll
            for (int t = 0; t < NTERMS; ++t) kk[t] = a.kx[t][q];
            double acc = a.kc[0] * kk[0], g = a.gc[0] * kk[0];
#pragma unroll
            for (int t = 1; t < NTERMS; ++t) { acc = acc + a.kc[t] * kk[t]; g = g + a.gc[t] * kk[t]; }
            v = g;
            return u + acc;
        }
        double k = a.kx[0][q];
        double acc = a.kc[0] * k, g = a.gc[0] * k;
        for (int t = 1; t < a.nterms; ++t) { k = a.kx[t][q]; acc = acc + a.kc[t] * k; g = g + a.gc[t] * k; }
        v = g;
        return u + acc;
    };

    double w[TF_NF][TF_W];
    double wv[STAGE_RHS ? TF_NVAR : 1][TF_W];
    TfJUniform ju;
    if (STAGE_RHS) ju.init(a.parsca, a.dx, L.nsys, e);
#pragma unroll
    for (int f = 0; f < TF_NF; ++f)
#pragma unroll
        for (int o = 1; o < TF_W; ++o) {
            if (STAGE_RHS && f < TF_NVAR) w[f][o] = ld2(f, i0 - TF_MP + o - 1, wv[STAGE_RHS ? f : 0][o]);
            else w[f][o] = ld(f, i0 - TF_MP + o - 1);
        }

#pragma unroll
    for (int j = 0; j < SEG; ++j) {
        const int i = i0 + j;
        if (i < len) {
#pragma unroll
            for (int f = 0; f < TF_NF; ++f) {
#pragma unroll
                for (int o = 0; o < TF_W - 1; ++o) w[f][o] = w[f][o + 1];
                if (STAGE_RHS && f < TF_NVAR) {
#pragma unroll
                    for (int o = 0; o < TF_W - 1; ++o) wv[STAGE_RHS ? f : 0][o] = wv[STAGE_RHS ? f : 0][o + 1];
                    w[f][TF_W - 1] = ld2(f, i + TF_MP, wv[STAGE_RHS ? f : 0][TF_W - 1]);
                } else {
                    w[f][TF_W - 1] = ld(f, i + TF_MP);
                }
            }
            const int64_t s = tf_idx(L, pg, i);
            const unsigned off = (unsigned)s * 8u;          // (tf_off8: scalar plane base + 32-bit lane offset)
#pragma unroll
            for (int k = 0; k < TF_NPAR; ++k)
                if (tf_par_is_vec[k]) par[k] = tf_ldp(a.parvec, k, L.plane, off);
            double xc = 0.0;
            if (TF_USES_X) xc = a.xcoord[s];
            double Fo[TF_NVAR];
            tf_eval_F(w, par, dx, xc, Fo);
            if (STAGE_RHS) {
                double acc[TF_NVAR];
#pragma unroll
                for (int v = 0; v < TF_NVAR; ++v) acc[v] = 0.0;
                double jr[TF_NNZ > 0 ? TF_NNZ : 1];
                ju.row([&](int k) { return tf_ldp(a.Jv, k, L.plane, off); }, jr);
#pragma unroll
                for (int k = 0; k < TF_NNZ; ++k) {       // tfk_spmv_body, scale = 1
                    const double jv = 1.0 * jr[k];
                    acc[tf_pat_eq[k]] = acc[tf_pat_eq[k]] +
                        jv * wv[STAGE_RHS ? tf_pat_var[k] : 0][tf_pat_off[k] + TF_MP];
                }
#pragma unroll
                for (int v = 0; v < TF_NVAR; ++v)
                    tf_stp(a.stage_rhs, v, L.plane, off, a.cF * (a.fscale * Fo[v]) + a.cA * acc[v]);
                continue;
            }
            // (the theta and BDF-2 sweeps leave F inside their right-hand side: nobody reads it alone)
            if (!(THETA || BDF) || a.F) {
#pragma unroll
                for (int v = 0; v < TF_NVAR; ++v)
                    TF_STORE_STREAM((double*)((char*)(a.F + (int64_t)v * L.plane) + off), a.fscale * Fo[v]);
            }
            if (WITH_J) {
                double Jo[TF_NNZ > 0 ? TF_NNZ : 1];
                tf_eval_J(w, par, dx, xc, Jo);
#pragma unroll
                for (int k = 0; k < TF_NNZ; ++k)
                    TF_STORE_STREAM((double*)((char*)(a.Jv + (int64_t)k * L.plane) + off), Jo[k]);
                if (BDF) {
                    // TF_VEC_BDF2_RHS and the copy of U into the history, per node
#pragma unroll
                    for (int v = 0; v < TF_NVAR; ++v) {
                        const double u = w[v][TF_MP];
                        const int64_t q = (int64_t)v * L.plane + s;
                        a.bdf_rhs[q] = a.bdf_two_step ? a.bdf_c0 * (u - a.bdf_prev[q]) + a.bdf_c1 * Fo[v]
                                                      : a.bdf_c1 * Fo[v];
                        if (a.bdf_prev_out) a.bdf_prev_out[q] = u;
                    }
                }
                if (THETA) {
                    // rhs of the theta scheme from the same window: the operations of
                    // tfk_spmv_body (scale = theta) followed by TF_VEC_THETA_RHS
                    double acc[TF_NVAR];
#pragma unroll
                    for (int v = 0; v < TF_NVAR; ++v) acc[v] = 0.0;
#pragma unroll
                    for (int k = 0; k < TF_NNZ; ++k) {
                        const double jv = a.theta * Jo[k];
                        acc[tf_pat_eq[k]] = acc[tf_pat_eq[k]] + jv * w[tf_pat_var[k]][tf_pat_off[k] + TF_MP];
                    }
#pragma unroll
                    for (int v = 0; v < TF_NVAR; ++v)
                        a.theta_rhs[(int64_t)v * L.plane + s] =
                            a.theta_dt * (a.fscale * Fo[v] - acc[v]) + w[v][TF_MP];
                }
            }
        }
    }
}

// ===========================================================================
// 2. y = scale * (J @ v)      (CSC product of schemes.py:157, 553; the column
//    of a stored value is the clamped / wrapped neighbour, compilers.py:303-328)
// ===========================================================================
TF_DEVICE void tfk_spmv_body(const TfSpmvArgs& a, int pg, int seg) {
    const TfLayout& L = a.L;
    if (pg >= L.Ptot) return;
    const int e = pg / L.P, p = pg - e * L.P;
    const int len = tf_len(L, p);
    const int i0 = seg * TF_SEG;
    if (i0 >= len) return;
    auto ld = [&](int v, int ii) -> double {
        const int64_t s = (ii >= 0 && ii < len) ? tf_idx(L, pg, ii) : tf_nbr(L, e, p, len, 0, ii);
        if (a.nterms == 0) return a.v[(int64_t)v * L.plane + s];
        double acc = a.vc[0] * a.vx[0][(int64_t)v * L.plane + s];       // sum_j gamma_ij k_j
        for (int t = 1; t < a.nterms; ++t) acc = acc + a.vc[t] * a.vx[t][(int64_t)v * L.plane + s];
        return acc;
    };
    TfJUniform ju;
    ju.init(a.parsca, a.dx, L.nsys, e);
    double w[TF_NVAR][TF_W];
#pragma unroll
    for (int v = 0; v < TF_NVAR; ++v)
#pragma unroll
        for (int o = 1; o < TF_W; ++o) w[v][o] = ld(v, i0 - TF_MP + o - 1);
#pragma unroll
    for (int j = 0; j < TF_SEG; ++j) {
        const int i = i0 + j;
        if (i < len) {
#pragma unroll
            for (int v = 0; v < TF_NVAR; ++v) {
#pragma unroll
                for (int o = 0; o < TF_W - 1; ++o) w[v][o] = w[v][o + 1];
                w[v][TF_W - 1] = ld(v, i + TF_MP);
            }
            const int64_t s = tf_idx(L, pg, i);
            double acc[TF_NVAR];
#pragma unroll
            for (int v = 0; v < TF_NVAR; ++v) acc[v] = 0.0;
            double jr[TF_NNZ > 0 ? TF_NNZ : 1];
            ju.row([&](int k) { return tf_ldp(a.Jv, k, L.plane, (unsigned)s * 8u); }, jr);
#pragma unroll
            for (int k = 0; k < TF_NNZ; ++k) {       // pattern order = ascending column
                double jv = a.scale * jr[k];
                double wv = w[tf_pat_var[k]][tf_pat_off[k] + TF_MP];
                if (a.absval) { jv = tf_abs(jv); wv = tf_abs(wv); }
                acc[tf_pat_eq[k]] = acc[tf_pat_eq[k]] + jv * wv;
            }
#pragma unroll
            for (int v = 0; v < TF_NVAR; ++v)
                a.y[(int64_t)v * L.plane + s] = a.addF
                    ? a.cF * a.addF[(int64_t)v * L.plane + s] + a.cA * acc[v] : acc[v];
        }
    }
}

// componentwise (Oettli-Prager) backward error of (I - cJ) x = b, one pass over J:
// returns this thread's  max |b - x + cJx| / (|x| + |cJ||x| + |b|)
TF_DEVICE double tfk_berr_body(const TfBerrArgs& a, int pg, int seg) {
    const TfLayout& L = a.L;
    if (pg >= L.Ptot) return 0.0;
    const int e = pg / L.P, p = pg - e * L.P;
    const int len = tf_len(L, p);
    // (sampled form: the one node of this chunk that is looked at, as segment + place in it)
    const int pick = a.one_node >= 0 ? a.one_node % len : -1;
    if (pick >= 0) seg = pick / TF_SEG;
    const int jlo = pick >= 0 ? pick - seg * TF_SEG : 0, jhi = pick >= 0 ? jlo + 1 : TF_SEG;
    const int i0 = seg * TF_SEG;
    if (i0 >= len) return 0.0;
    const bool based = a.xbase != nullptr;
    // value and magnitude of the unknown at (variable v, node ii of the chunk, ghost nodes included)
    auto ld = [&](int v, int ii, double& val, double& mag) {
        const int64_t at = (ii >= 0 && ii < len) ? tf_idx(L, pg, ii) : tf_nbr(L, e, p, len, 0, ii);
        const double xv = a.x[(int64_t)v * L.plane + at];
        const double xb = based ? a.xbase[(int64_t)v * L.plane + at] : 0.0;
        val = xv - xb;
        mag = based ? tf_abs(xv) + tf_abs(xb) : tf_abs(xv);
    };
    double w[TF_NVAR][TF_W], wm[TF_NVAR][TF_W];
#pragma unroll
    for (int v = 0; v < TF_NVAR; ++v)
#pragma unroll
        for (int o = 1; o < TF_W; ++o) ld(v, i0 + jlo - TF_MP + o - 1, w[v][o], wm[v][o]);
    TfJUniform ju;
    ju.init(a.parsca, a.dx, L.nsys, e);
    double worst = 0.0;
#pragma unroll
    for (int j = 0; j < TF_SEG; ++j) {
        const int i = i0 + j;
        if (i < len && j >= jlo && j < jhi) {
#pragma unroll
            for (int v = 0; v < TF_NVAR; ++v) {
#pragma unroll
                for (int o = 0; o < TF_W - 1; ++o) { w[v][o] = w[v][o + 1]; wm[v][o] = wm[v][o + 1]; }
                ld(v, i + TF_MP, w[v][TF_W - 1], wm[v][TF_W - 1]);
            }
            const int64_t s = tf_idx(L, pg, i);
            double acc[TF_NVAR], mag[TF_NVAR];
#pragma unroll
            for (int v = 0; v < TF_NVAR; ++v) { acc[v] = 0.0; mag[v] = 0.0; }
            double jr[TF_NNZ > 0 ? TF_NNZ : 1];
            ju.row([&](int k) { return a.Jv[(int64_t)k * L.plane + s]; }, jr);
#pragma unroll
            for (int k = 0; k < TF_NNZ; ++k) {
                const double jv = a.c * jr[k];
                acc[tf_pat_eq[k]] = acc[tf_pat_eq[k]] + jv * w[tf_pat_var[k]][tf_pat_off[k] + TF_MP];
                mag[tf_pat_eq[k]] = mag[tf_pat_eq[k]] + tf_abs(jv) * wm[tf_pat_var[k]][tf_pat_off[k] + TF_MP];
            }
#pragma unroll
            for (int v = 0; v < TF_NVAR; ++v) {
                const double b = a.rhs[(int64_t)v * L.plane + s];
                const double num = tf_abs((b - w[v][TF_MP]) + acc[v]);
                const double den = wm[v][TF_MP] + mag[v] + tf_abs(b);
                const double q = num == 0.0 ? 0.0 : num / den;
                worst = (q > worst || q != q) ? q : worst;
            }
        }
    }
    return worst;
}

// ===========================================================================
// 3. elementwise plane algebra of the schemes (schemes.py:152-174, 553)
//    written without contraction, in the association order NumPy uses
// ===========================================================================
enum TfVecOp {
    TF_VEC_SUM = 0,        // out = base + (((c0*x0) + c1*x1) + ...)       (base optional)
    TF_VEC_LIN2 = 1,       // out = c0*x0 + c1*x1
    TF_VEC_THETA_RHS = 2,  // out = c0*(x0 - x1) + x2       B = dt*(F - theta*J@U) + U
    TF_VEC_MAXABS = 3,     // red = max |(((c0*x0) + c1*x1) + ...)|          (error estimate)
    TF_VEC_COPY = 4,       // out = x0
    TF_VEC_BDF2_RHS = 5,   // out = c0*(x0 - x1) + c1*x2    1/3 (U - Uprev) + 2/3 dt F
    TF_VEC_ADD = 6,        // out = x0 + x1
    TF_VEC_RESID = 7,      // out = (x0 - x1) + x2          r = b - x + c J x
    TF_VEC_MAXRATIO = 8,   // red = max |x0| / (|x1| + |x2| + |x3|)   componentwise backward error
    TF_VEC_SUM_ERR = 9,    // TF_VEC_SUM and, on its result, TF_VEC_MAXABS with the coefficients c2: the new
                           // state and the embedded error estimate of an adaptive Rosenbrock step in one pass
};

TF_DEVICE double tf_vec_sum(const TfVecArgs& a, int64_t i) {
    double acc = a.c[0] * a.x[0][i];
    for (int t = 1; t < a.nterms; ++t) acc = acc + a.c[t] * a.x[t][i];
    return acc;
}
TF_DEVICE double tf_vec_ratio(const TfVecArgs& a, int64_t i) {
    const double num = tf_abs(a.x[0][i]);
    const double den = tf_abs(a.x[1][i]) + tf_abs(a.x[2][i]) + tf_abs(a.x[3][i]);
    return num == 0.0 ? 0.0 : num / den;          // x/0 = inf flags a broken row
}
// element of the ROW error estimate: |U - U_pred| with U_pred = U + sum_i b_pred_i k_i
// formed from the updated U (schemes.py:167-174); plain |sum| without a base
TF_DEVICE double tf_vec_err(const TfVecArgs& a, int64_t i) {
    const double acc = tf_vec_sum(a, i);
    return tf_abs(a.base ? a.base[i] - (a.base[i] + acc) : acc);
}

// TF_VEC_SUM_ERR: out = base + sum c x (tfk_vec_elem, TF_VEC_SUM), returns |out - (out + sum c2 x)| (tf_vec_err
// with the new state as base): the two kernels' operations in their order
TF_DEVICE double tf_vec_sum_err(const TfVecArgs& a, int64_t i) {
    double acc = a.c[0] * a.x[0][i], acc2 = a.c2[0] * a.x[0][i];
    for (int t = 1; t < a.nterms; ++t) { acc = acc + a.c[t] * a.x[t][i]; acc2 = acc2 + a.c2[t] * a.x[t][i]; }
    const double nu = a.base[i] + acc;
    a.out[i] = nu;
    return tf_abs(nu - (nu + acc2));
}

TF_DEVICE void tfk_vec_elem(const TfVecArgs& a, int64_t i) {
    switch (a.op) {
    case TF_VEC_SUM: {
        double acc = tf_vec_sum(a, i);
        a.out[i] = a.base ? a.base[i] + acc : acc;
    } break;
    case TF_VEC_LIN2: a.out[i] = a.c[0] * a.x[0][i] + a.c[1] * a.x[1][i]; break;
    case TF_VEC_THETA_RHS: a.out[i] = a.c[0] * (a.x[0][i] - a.x[1][i]) + a.x[2][i]; break;
    case TF_VEC_COPY: a.out[i] = a.x[0][i]; break;
    case TF_VEC_BDF2_RHS: a.out[i] = a.c[0] * (a.x[0][i] - a.x[1][i]) + a.c[1] * a.x[2][i]; break;
    case TF_VEC_ADD: a.out[i] = a.x[0][i] + a.x[1][i]; break;
    case TF_VEC_RESID: a.out[i] = (a.x[0][i] - a.x[1][i]) + a.x[2][i]; break;
    default: break;
    }
}

// ===========================================================================
// 4. natural order <-> partition-interleaved copies (host boundary only)
// ===========================================================================
enum TfPermMode {
    TF_PERM_IN_SOA = 0,      // src [ncomp][nsys][N]          -> dst planes
    TF_PERM_OUT_SOA = 1,     // src planes                    -> dst [ncomp][nsys][N]
    TF_PERM_OUT_AOS = 2,     // src planes                    -> dst [nsys][N][ncomp]  (uflat / J value table)
    TF_PERM_IN_AOS = 3,      // src [nsys][N][ncomp]          -> dst planes
};

TF_DEVICE void tfk_perm_elem(const TfPermArgs& a, int64_t t) {
    const TfLayout& L = a.L;
    const int64_t per = (int64_t)L.nsys * L.N;
    if (t >= per) return;
    const int e = (int)(t / L.N), g = (int)(t - (int64_t)e * L.N);
    int p, i;
    tf_locate(L, g, p, i);
    const int64_t s = tf_idx(L, e * L.P + p, i);
    for (int c = 0; c < a.ncomp; ++c) {
        switch (a.mode) {
        case TF_PERM_IN_SOA: a.dst[(int64_t)c * L.plane + s] = a.src[(int64_t)c * per + t]; break;
        case TF_PERM_OUT_SOA: a.dst[(int64_t)c * per + t] = a.src[(int64_t)c * L.plane + s]; break;
        case TF_PERM_OUT_AOS: a.dst[t * a.ncomp + c] = a.src[(int64_t)c * L.plane + s]; break;
        case TF_PERM_IN_AOS: a.dst[(int64_t)c * L.plane + s] = a.src[t * a.ncomp + c]; break;
        }
    }
}

// Jacobian values in the order of a caller's index list (the CSC data array of the drop-in
// J function: compilers.py:303-331 with the pattern computed once): entry map[t] = node * nnz + k
// of system 0's value table
TF_DEVICE void tfk_gather_elem(const TfGatherArgs& a, int64_t t) {
    if (t >= a.n) return;
    const int src = a.map[t];
    const int g = src / a.nnz, k = src - g * a.nnz;
    int p, i;
    tf_locate(a.L, g, p, i);
    a.out[t] = a.Jv[(int64_t)k * a.L.plane + tf_idx(a.L, p, i)];
}

// (a - b) of variable/system `vs` restricted to the slice of work item (blk, tid):
// this thread's partial sum of squares (ord 2) or maximum (ord 0)
TF_DEVICE double tfk_diffnorm_partial(const TfNormArgs& a, int vs, int blk, int tid, int nthreads) {
    const TfLayout& L = a.L;
    const int v = vs / L.nsys, e = vs - v * L.nsys;
    const int64_t total = (int64_t)L.M * L.P;
    const int64_t per = (total + a.nblocks - 1) / a.nblocks;
    const int64_t lo = (int64_t)blk * per, hi = lo + per < total ? lo + per : total;
    double acc = 0.0;
    for (int64_t q = lo + tid; q < hi; q += nthreads) {
        const int64_t i = q / L.P, p = q - i * L.P;
        const int64_t s = (int64_t)v * L.plane + i * L.Ptot + (int64_t)e * L.P + p;
        const double d = a.a[s] - a.b[s];
        if (a.ord == 2) acc = tf_fma(d, d, acc);
        else acc = tf_abs(d) > acc ? tf_abs(d) : acc;
    }
    return acc;
}

// declarative Dirichlet hook: U[var][node] = value in every system
TF_DEVICE void tfk_dirichlet_elem(const TfDirichletArgs& a, int t) {
    const TfLayout& L = a.L;
    if (t >= a.n * L.nsys) return;
    const int k = t % a.n, e = t / a.n;
    int g = a.node[k];
    if (g < 0) g += L.N;
    int p, i;
    tf_locate(L, g, p, i);
    a.fields[(int64_t)a.var[k] * L.plane + tf_idx(L, e * L.P + p, i)] = a.value[k];
}

TF_DEVICE void tfk_poke_elem(const TfPokeArgs& a, int t) {
    const TfLayout& L = a.L;
    if (t >= a.n * L.nsys) return;
    const int k = t % a.n, e = t / a.n;
    int g = 0, v = 0;
    double val = 0.0;
#pragma unroll
    for (int q = 0; q < TF_POKE_MAX; ++q)          // by-value arrays: no dynamic indexing
        if (q == k) { g = a.node[q]; v = a.var[q]; val = a.value[q]; }
    if (g < 0) g += L.N;
    int p, i;
    tf_locate(L, g, p, i);
    a.fields[(int64_t)v * L.plane + tf_idx(L, e * L.P + p, i)] = val;
}

// ===========================================================================
// 5. block-banded direct solver
// ===========================================================================
// A = I - c J is block banded: blocks of B = nvar unknowns per node, half
// bandwidth MP nodes, cyclic when the system is periodic.  Every chunk of the
// layout is split into its first len-MP nodes (interior) and its last MP nodes
// (separator).  The interiors are mutually independent once the separators
// are known, so
//   factor:  two threads per chunk eliminate the interior by block LU (partial
//            pivoting inside the B x B pivot block), one walking down, one
//            walking up, each carrying the coupling to the separator behind it
//            as extra right-hand sides ("spike").  Their last MP nodes give the
//            response of the chunk ends to the separator values (tips).
//   assemble: one thread per separator folds the tips of the chunks above and
//            below into a block-tridiagonal system of MP*B x MP*B blocks over
//            the separators -- the next level, solved the same way, until one
//            chunk is left whose separator system is a single block.
//   solve:   the same walk with one right-hand side (the elimination is
//            recomputed from J: 8*nnz bytes/node instead of reading stored L
//            factors), then back-substitution top level first, reading the
//            stored normalised U rows.

template <int B> struct TfBlk { double v[B][B]; };
template <int V> struct TfInt { static constexpr int value = V; };

template <int B>
TF_DEVICE void tf_blk_zero(double (&a)[B][B]) {
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
        for (int c = 0; c < B; ++c) a[r][c] = 0.0;
}
template <int B>
TF_DEVICE void tf_blk_copy(double (&d)[B][B], const double (&s)[B][B]) {
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
        for (int c = 0; c < B; ++c) d[r][c] = s[r][c];
}
// C -= A * Bm
template <int B>
TF_DEVICE void tf_mm_sub(double (&C)[B][B], const double (&A)[B][B], const double (&Bm)[B][B]) {
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
        for (int c = 0; c < B; ++c) {
            double acc = C[r][c];
#pragma unroll
            for (int k = 0; k < B; ++k) acc = tf_fma(-A[r][k], Bm[k][c], acc);
            C[r][c] = acc;
        }
}
// C = A * Bm
template <int B>
TF_DEVICE void tf_mm(double (&C)[B][B], const double (&A)[B][B], const double (&Bm)[B][B]) {
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
        for (int c = 0; c < B; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < B; ++k) acc = tf_fma(A[r][k], Bm[k][c], acc);
            C[r][c] = acc;
        }
}
template <int B>
TF_DEVICE void tf_mv_sub(double (&y)[B], const double (&A)[B][B], const double (&x)[B]) {
#pragma unroll
    for (int r = 0; r < B; ++r) {
        double acc = y[r];
#pragma unroll
        for (int k = 0; k < B; ++k) acc = tf_fma(-A[r][k], x[k], acc);
        y[r] = acc;
    }
}
template <int B>
TF_DEVICE void tf_mv(double (&y)[B], const double (&A)[B][B], const double (&x)[B]) {
#pragma unroll
    for (int r = 0; r < B; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < B; ++k) acc = tf_fma(A[r][k], x[k], acc);
        y[r] = acc;
    }
}

// Gauss-Jordan inverse with partial (row) pivoting, all in registers.
template <int B>
TF_DEVICE bool tf_blk_inverse(const double (&A)[B][B], double (&inv)[B][B]) {
    double a[B][B];
    tf_blk_copy<B>(a, A);
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
        for (int c = 0; c < B; ++c) inv[r][c] = (r == c) ? 1.0 : 0.0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < B; ++k) {
#pragma unroll
        for (int r = k + 1; r < B; ++r) {          // bubble the largest |a[r][k]| up to row k
            const bool sw = tf_abs(a[r][k]) > tf_abs(a[k][k]);
#pragma unroll
            for (int c = 0; c < B; ++c) {
                const double t0 = a[k][c], t1 = a[r][c];
                a[k][c] = sw ? t1 : t0;
                a[r][c] = sw ? t0 : t1;
                const double u0 = inv[k][c], u1 = inv[r][c];
                inv[k][c] = sw ? u1 : u0;
                inv[r][c] = sw ? u0 : u1;
            }
        }
        const double piv = a[k][k];
        ok = ok && (piv != 0.0) && tf_finite(piv);
        const double rp = 1.0 / piv;
#pragma unroll
        for (int c = 0; c < B; ++c) { a[k][c] *= rp; inv[k][c] *= rp; }
#pragma unroll
        for (int r = 0; r < B; ++r) {
            if (r == k) continue;
            const double f = a[r][k];
#pragma unroll
            for (int c = 0; c < B; ++c) {
                a[r][c] = tf_fma(-f, a[k][c], a[r][c]);
                inv[r][c] = tf_fma(-f, inv[k][c], inv[r][c]);
            }
        }
    }
    return ok;
}

// ---- spike tips of one chunk, struct-of-arrays over chunks -----------------
// y[k][B], V[k][t][B][B] (response to the separator ABOVE), W[k][t][B][B]
// (response to the separator BELOW); k = tip node, t = separator node, both in
// natural (top to bottom) order:  x_tip[k] = y[k] - sum_t V[k][t] s_above[t]
//                                                 - sum_t W[k][t] s_below[t]
template <int B, int MP> struct TfTips {
    static constexpr int NY = MP * B;
    static constexpr int NV = MP * MP * B * B;
    static constexpr int SIZE = NY + 2 * NV;
    TF_DEVICE_M static int y(int k, int r) { return k * B + r; }
    TF_DEVICE_M static int V(int k, int t, int r, int c) { return NY + ((k * MP + t) * B + r) * B + c; }
    TF_DEVICE_M static int W(int k, int t, int r, int c) { return NY + NV + ((k * MP + t) * B + r) * B + c; }
};

// ---- block-row providers ----------------------------------------------------
// load(i, row): row[MP + d] = A(node i, node i + d), d = -MP..MP, in the natural
// orientation, for a node i of the thread's own chunk.

// level 1: A = I - c J from the raw Jacobian value planes.  For a clamped
// system the ghost columns fold onto the boundary node (duplicates summed,
// compilers.py:330-331).
struct TfRowsL1 {
    static constexpr int B = TF_NVAR;
    static constexpr int MP = TF_MP;
    // scalar equations: partial pivoting over the band inside the chunk interior
    // (rows are exchanged, so U widens to 2*MP like LAPACK's gbtrf)
    static constexpr bool PIVOT = TF_NVAR == 1;
    const TfLevelArgs& a;
    int pg, e, p, len, start;
    TfJUniform ju;                 // the node-independent entries: not read back
    TF_DEVICE_M TfRowsL1(const TfLevelArgs& a_, int pg_) : a(a_), pg(pg_) {
        e = pg / a.L.P; p = pg - e * a.L.P;
        len = tf_len(a.L, p); start = tf_start(a.L, p);
        ju.init(a.parsca, a.dx, a.L.nsys, e);
    }
    // the values of one block row as they lie in memory; requested ahead of their use
    struct Raw { double jv[TF_NNZ > 0 ? TF_NNZ : 1]; };
    TF_DEVICE_M void request(int i, Raw& r) const {
        // (plane base in scalar registers + a 32-bit lane offset: planes are < 4 GB, tf_solver_create)
        // (proportional entries are filled in by decode(): scaling here would wait for the load
        // that this request only wants to put on its way)
        const unsigned off = tf_off8(a.L, pg, i);
#pragma unroll
        for (int k = 0; k < TF_NNZ; ++k)
            r.jv[k] = TF_JU(k) ? ju.v[k] : (TF_JA(k) ? 0.0 : tf_ldp(a.Jv, k, a.L.plane, off));
    }
    TF_DEVICE_M void decode(int i, const Raw& raw, double (&row)[2 * TF_MP + 1][TF_NVAR][TF_NVAR]) const {
#pragma unroll
        for (int d = 0; d < TF_W; ++d) tf_blk_zero<TF_NVAR>(row[d]);
#pragma unroll
        for (int k = 0; k < TF_NNZ; ++k) {
            const double jv = TF_JA(k) ? tf_j_alias_scale[k] * raw.jv[tf_j_alias[k] >= 0 ? tf_j_alias[k] : 0] : raw.jv[k];
            row[tf_pat_off[k] + TF_MP][tf_pat_eq[k]][tf_pat_var[k]] = -a.c * jv;
        }
        if (!a.L.periodic) {
            const int gl = start + i, gr = a.L.N - 1 - gl;
            if (gl < TF_MP) {
#pragma unroll
                for (int d = -TF_MP; d < 0; ++d)
                    if (d < -gl) {
#pragma unroll
                        for (int t = -TF_MP + 1; t <= 0; ++t)
                            if (t == -gl) {
#pragma unroll
                                for (int r = 0; r < TF_NVAR; ++r)
#pragma unroll
                                    for (int c = 0; c < TF_NVAR; ++c) {
                                        row[t + TF_MP][r][c] += row[d + TF_MP][r][c];
                                        row[d + TF_MP][r][c] = 0.0;
                                    }
                            }
                    }
            }
            if (gr < TF_MP) {
#pragma unroll
                for (int d = TF_MP; d > 0; --d)
                    if (d > gr) {
#pragma unroll
                        for (int t = TF_MP - 1; t >= 0; --t)
                            if (t == gr) {
#pragma unroll
                                for (int r = 0; r < TF_NVAR; ++r)
#pragma unroll
                                    for (int c = 0; c < TF_NVAR; ++c) {
                                        row[t + TF_MP][r][c] += row[d + TF_MP][r][c];
                                        row[d + TF_MP][r][c] = 0.0;
                                    }
                            }
                    }
            }
        }
#pragma unroll
        for (int r = 0; r < TF_NVAR; ++r) row[TF_MP][r][r] += 1.0;
    }
    TF_DEVICE_M void load(int i, double (&row)[2 * TF_MP + 1][TF_NVAR][TF_NVAR]) const {
        Raw raw;
        request(i, raw);
        decode(i, raw, row);
    }
};

// level >= 2: explicit block-tridiagonal rows [3][b][b]
template <int BB>
struct TfRowsBT {
    static constexpr int B = BB;
    static constexpr int MP = 1;
    static constexpr bool PIVOT = false;
    const TfLevelArgs& a;
    int pg, e, p, len, start;
    TF_DEVICE_M TfRowsBT(const TfLevelArgs& a_, int pg_) : a(a_), pg(pg_) {
        e = pg / a.L.P; p = pg - e * a.L.P;
        len = tf_len(a.L, p); start = tf_start(a.L, p);
    }
    struct Raw { double blk[3][BB][BB]; };
    TF_DEVICE_M void request(int i, Raw& r) const {
        const int64_t s = tf_idx(a.L, pg, i);
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int rr = 0; rr < BB; ++rr)
#pragma unroll
                for (int c = 0; c < BB; ++c)
                    r.blk[d][rr][c] = a.Ablk[(int64_t)((d * BB + rr) * BB + c) * a.L.plane + s];
    }
    TF_DEVICE_M void decode(int i, const Raw& raw, double (&row)[3][BB][BB]) const {
#pragma unroll
        for (int d = 0; d < 3; ++d) tf_blk_copy<BB>(row[d], raw.blk[d]);
        if (!a.L.periodic) {           // no neighbour beyond the ends
            const int g = start + i;
            if (g == 0) tf_blk_zero<BB>(row[0]);
            if (g == a.L.N - 1) tf_blk_zero<BB>(row[2]);
        }
    }
    TF_DEVICE_M void load(int i, double (&row)[3][BB][BB]) const {
        Raw raw;
        request(i, raw);
        decode(i, raw, row);
    }
};

// Rows requested ahead of their use in the level-1 walks: one (0) or two (1).  Two rows hide more
// latency but cost 2 x (nnz + nvar) registers in kernels that already overflow into AGPRs; with the
// walks bound by their instruction stream the moves cost more than the latency:
// one row ahead is 4 % faster on config 3 (tfk_l1_fwd2 63 -> 52 us), 2 % on config 5 / 8 members.
// The row requests of the walks are prefetches: they must be *issued* where they are written, one
// node ahead of their use.  TF_PIN keeps the compiler from sinking them towards the use (a
// memory clobber: loads and stores keep their side of it, the wait stays at the use).
// (config 3 +0.5 %, config 5 +3 %, 8 members +0.4 %: profiles/r03_ab_runs.txt)
#define TF_PIN_REQUESTS() asm volatile("" ::: "memory")
#ifndef TF_BACKSUB_DEPTH
#define TF_BACKSUB_DEPTH 3
#endif
// ... as long as the ring fits the registers: `nd` doubles of factors per node; wide blocks keep
// fewer nodes in flight instead of spilling (6 variables with 3-point stencils: 2, 5 with 5-point: 1)
// (`other`: doubles the walk holds besides the ring -- the middle system of the twisted form)
constexpr int tf_ring_depth(int nd, int other = 40) {
    return TF_BACKSUB_DEPTH * nd + other <= 240 ? TF_BACKSUB_DEPTH
         : (2 * nd + other <= 240 && TF_BACKSUB_DEPTH >= 2 ? 2 : 1);
}

// Twisted re-elimination (a.respike).  With the separators known, the interior of a chunk is a
// banded system with known values on both sides, so it need not be swept end to end: the down
// walk owns its first h nodes, the up walk the other mI - h, both eliminate towards the middle at
// the same time (tfk_l1_fwd2, two threads per chunk), the 2*MP nodes where they meet are solved
// as one small system, and the two halves are back-substituted outwards (tfk_l1_backsub_u, two
// threads per chunk).  Half the latency per walk and twice the wavefronts -- 31 250 chunks are
// only 489 wavefronts for 1024 SIMDs.  The factorisation stores the normalised pivot rows U of
// the down walk for the first h nodes and those of the up walk for the others: the same bytes.
// Chunks too short for two halves (and blocks too big for the middle system in registers) keep
// h = mI: the up half is empty and everything reduces to the one-sided form.  So do solvers whose
// chunks fill the GPU anyway (a.twist = 0: more than TF_TWIST_MAX_CHUNKS; there the second walk
// only adds the middle system: 8 members per GPU -3 %, config 5 -1.5 %).
template <int B, int MP>
TF_DEVICE int tf_twist_h(int mI, int enabled) {
    return (enabled && MP * B <= 6 && mI >= 4 * MP) ? (mI + 1) / 2 : mI;
}

// ---- the separator equations, assembled by the walks themselves ---------------
// When the next level keeps records per node (cyclic reduction: [sub, dia, sup, second part of
// dia] and a right-hand side in two parts, summed when the level loads them), the two halves
// of a separator's row can be made where the tips are: the DOWN walk of chunk p ends next to its
// own separator and owns the tips (V, W, y) of the nodes above it -- it writes sub, the first
// part of dia and of the right-hand side; the UP walk of chunk p+1 ends below that separator
// and writes sup and the second parts.  Same products as tfk_asm_body (whose sums take the two
// sides in one pass: the diagonal block differs in the last bit), no tips in memory, no
// tfk_l1_asm_mat / tfk_l1_asm_rhs launch (profiles/r03_ab_runs.txt).  `stage`: this lane's
// [2][b][b] part of the record (LDS, written out as whole records by the kernel); *rec = the
// record (node of the next level) it belongs to, -1 = none.
template <class Rows, int DIR, bool MATRIX, bool DO_V = true, bool DO_W = true, bool DO_G = true>
TF_DEVICE void tf_asm_side(const TfLevelArgs& a, const Rows& own, int pg,
                           const double (&yN)[Rows::MP][Rows::B],
                           const double (&VN)[MATRIX ? Rows::MP : 1][MATRIX ? Rows::MP : 1][Rows::B][Rows::B],
                           const double (&WN)[MATRIX ? Rows::MP : 1][MATRIX ? Rows::MP : 1][Rows::B][Rows::B],
                           double* stage, int* rec) {
    // DO_V / DO_W / DO_G (split walks, tfk_chunk_body ROLE): the block that takes the products
    // with V (down: sub, up: second part of dia), the one that takes those with W (down: dia,
    // up: sup) and the right-hand side are not all made by the same wavefront
    constexpr int B = Rows::B, MP = Rows::MP, W = 2 * MP + 1, BB = MP * B;
    constexpr int VBLK = DIR > 0 ? 0 : 1, WBLK = 1 - VBLK;
    const TfLayout& L = a.L;
    const int e = own.e, p = own.p;
    const bool has = DIR > 0 || L.periodic || p > 0;
    const int ps = DIR > 0 ? p : (p > 0 ? p - 1 : L.P - 1);      // the chunk that owns the separator
    const int r_next = has ? e * a.Lnext.N + ps : -1;             // (Lnext.N == L.P)
    if (rec) *rec = r_next;
    if (!has) return;
    const int pgs = e * L.P + ps;
    Rows rs(a, pgs);
    const int mIs = rs.len - MP;
#pragma unroll
    for (int t = 0; t < MP; ++t) {                 // separator node t = node mIs + t of chunk ps
        double row[W][B][B];
        rs.load(mIs + t, row);
        double AV[MP][B][B], AW[MP][B][B], g[B];
#pragma unroll
        for (int t2 = 0; t2 < MP; ++t2) { tf_blk_zero<B>(AV[t2]); tf_blk_zero<B>(AW[t2]); }
        const int64_t s = tf_idx(L, pgs, mIs + t);
#pragma unroll
        for (int r = 0; r < B; ++r) g[r] = (DO_G && DIR > 0 && a.rhs) ? a.rhs[(int64_t)r * L.plane + s] : 0.0;
#pragma unroll
        for (int d = -MP; d <= MP; ++d) {
            const int cn = t + d;                  // column relative to the separator start
            // down: the bottom tip node cn + MP of the own interior; up: the top tip node cn - MP
            // of the interior below
            const bool tip = DIR > 0 ? cn < 0 : cn >= MP;
            const int kt = DIR > 0 ? cn + MP : cn - MP;
            if (DIR > 0 && cn >= 0 && cn < MP) {
                if (MATRIX && DO_W) {
#pragma unroll
                    for (int r = 0; r < B; ++r)
#pragma unroll
                        for (int c = 0; c < B; ++c) AW[cn][r][c] += row[d + MP][r][c];
                }
            } else if (tip) {
                if (DO_G) tf_mv_sub<B>(g, row[d + MP], yN[kt]);
                if (MATRIX) {
#pragma unroll
                    for (int t2 = 0; t2 < MP; ++t2) {
                        if (DO_V) tf_mm_sub<B>(AV[t2], row[d + MP], VN[MATRIX ? kt : 0][MATRIX ? t2 : 0]);
                        if (DO_W) tf_mm_sub<B>(AW[t2], row[d + MP], WN[MATRIX ? kt : 0][MATRIX ? t2 : 0]);
                    }
                }
            }
        }
        if (DO_G && a.rhs) {
#pragma unroll
            for (int r = 0; r < B; ++r)
                a.rhsnext[(int64_t)r_next * 2 * BB + (DIR > 0 ? 0 : BB) + t * B + r] = g[r];
        }
        if (MATRIX) {
#pragma unroll
            for (int t2 = 0; t2 < MP; ++t2)
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int c = 0; c < B; ++c) {
                        const int rr = t * B + r, cc = t2 * B + c;
                        if (DO_V) stage[(VBLK * BB + rr) * BB + cc] = AV[t2][r][c];
                        if (DO_W) stage[(WBLK * BB + rr) * BB + cc] = AW[t2][r][c];
                    }
        }
    }
}

// ---- tips: back-substitute the last MP pivots (local k = 0..MP-1 <-> local node mI-MP+k);
// unknowns beyond the interior are the separator ahead.
//   x_k = yb_k - sum_t Vb[k][t] s_behind[t] - sum_t Wb[k][t] s_ahead[t]
// and leave them in natural orientation: in memory (tips_dn / tips_up) or, with a.fuse_asm, as
// this walk's half of the separator's row.  DO_V / DO_WY: the part made by this wavefront (both,
// or one each in a split walk).
template <class Rows, int DIR, bool SPIKE, bool DO_V, bool DO_W, bool DO_Y, int UW, int NE>
TF_DEVICE void tf_tips_out(const TfLevelArgs& a, const Rows& rows, int pg,
                           const double (&Uh)[Rows::MP][UW][Rows::B][Rows::B], const double (&yh)[Rows::MP][Rows::B],
                           const double (&Eh)[NE][NE][Rows::B][Rows::B],      // NE = SPIKE ? MP : 1
                           double* asm_stage, int* asm_rec) {
    constexpr int B = Rows::B, MP = Rows::MP;
    typedef TfTips<B, MP> Tip;
    const TfLayout& L = a.L;
    double yb[MP][B];
    double Vb[SPIKE ? MP : 1][SPIKE ? MP : 1][B][B], Wb[SPIKE ? MP : 1][SPIKE ? MP : 1][B][B];
#pragma unroll
    for (int k = MP - 1; k >= 0; --k) {
#pragma unroll
        for (int r = 0; r < B; ++r) yb[k][r] = DO_Y ? yh[k][r] : 0.0;
        if (SPIKE) {
#pragma unroll
            for (int t = 0; t < MP; ++t) {
                if (DO_V) tf_blk_copy<B>(Vb[SPIKE ? k : 0][SPIKE ? t : 0], Eh[SPIKE ? k : 0][SPIKE ? t : 0]);
                else tf_blk_zero<B>(Vb[SPIKE ? k : 0][SPIKE ? t : 0]);
                tf_blk_zero<B>(Wb[SPIKE ? k : 0][SPIKE ? t : 0]);
            }
        }
#pragma unroll
        for (int c = 1; c <= UW; ++c) {
            const int kk = k + c;
            if (kk >= 2 * MP) {
                // beyond the separator ahead: no interior row reaches there, the
                // (exchanged) pivot row holds an exact zero
            } else if (kk < MP) {
                if (DO_Y) tf_mv_sub<B>(yb[k], Uh[k][c - 1], yb[kk]);
                if (SPIKE) {
#pragma unroll
                    for (int t = 0; t < MP; ++t) {
                        if (DO_V) tf_mm_sub<B>(Vb[SPIKE ? k : 0][SPIKE ? t : 0], Uh[k][c - 1], Vb[SPIKE ? kk : 0][SPIKE ? t : 0]);
                        if (DO_W) tf_mm_sub<B>(Wb[SPIKE ? k : 0][SPIKE ? t : 0], Uh[k][c - 1], Wb[SPIKE ? kk : 0][SPIKE ? t : 0]);
                    }
                }
            } else if (SPIKE && DO_W) {
                const int t = kk - MP;                  // separator ahead, local position t
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int cc = 0; cc < B; ++cc) Wb[SPIKE ? k : 0][SPIKE ? t : 0][r][cc] += Uh[k][c - 1][r][cc];
            }
        }
    }

    if (a.fuse_asm) {
        // the tips in natural orientation stay in registers: this walk's half of the separator's row
        // (down: behind = above (V), ahead = below (W); up: the reverse)
        double yN[MP][B];
        double VN[SPIKE ? MP : 1][SPIKE ? MP : 1][B][B], WN[SPIKE ? MP : 1][SPIKE ? MP : 1][B][B];
#pragma unroll
        for (int k = 0; k < MP; ++k) {
            const int kn = DIR > 0 ? k : MP - 1 - k;
#pragma unroll
            for (int r = 0; r < B; ++r) yN[kn][r] = yb[k][r];
            if (SPIKE) {
#pragma unroll
                for (int t = 0; t < MP; ++t) {
                    const int tn = DIR > 0 ? t : MP - 1 - t;
                    tf_blk_copy<B>(VN[SPIKE ? kn : 0][SPIKE ? tn : 0], DIR > 0 ? Vb[SPIKE ? k : 0][SPIKE ? t : 0] : Wb[SPIKE ? k : 0][SPIKE ? t : 0]);
                    tf_blk_copy<B>(WN[SPIKE ? kn : 0][SPIKE ? tn : 0], DIR > 0 ? Wb[SPIKE ? k : 0][SPIKE ? t : 0] : Vb[SPIKE ? k : 0][SPIKE ? t : 0]);
                }
            }
        }
        // (in natural orientation the walk's V is the row's V going down and its W going up)
        constexpr bool DOWN = (DIR > 0), NAT_V = DOWN ? DO_V : DO_W, NAT_W = DOWN ? DO_W : DO_V;
        tf_asm_side<Rows, DIR, SPIKE, NAT_V, NAT_W, DO_Y>(a, rows, pg, yN, VN, WN, asm_stage, asm_rec);
        return;
    }
    // ---- write in natural orientation
    double* tips = DIR > 0 ? a.tips_dn : a.tips_up;
    auto put = [&](int slot, double v) { tips[(int64_t)slot * L.Ptot + pg] = v; };
#pragma unroll
    for (int k = 0; k < MP; ++k) {
        const int kn = DIR > 0 ? k : MP - 1 - k;        // natural tip index
        if (DO_Y) {
#pragma unroll
            for (int r = 0; r < B; ++r) put(Tip::y(kn, r), yb[k][r]);
        }
        if (SPIKE) {
#pragma unroll
            for (int t = 0; t < MP; ++t) {
                const int tn = DIR > 0 ? t : MP - 1 - t;
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int c = 0; c < B; ++c) {
                        // down: behind = above (V), ahead = below (W); up: the reverse
                        const double vb = Vb[SPIKE ? k : 0][SPIKE ? t : 0][r][c];
                        const double wb = Wb[SPIKE ? k : 0][SPIKE ? t : 0][r][c];
                        if (DO_V) put(DIR > 0 ? Tip::V(kn, tn, r, c) : Tip::W(kn, tn, r, c), vb);
                        if (DO_W) put(DIR > 0 ? Tip::W(kn, tn, r, c) : Tip::V(kn, tn, r, c), wb);
                    }
            }
        }
    }
}

// ---- interior elimination of one chunk in one direction --------------------
// DIR = +1 walks down (local j <-> node j), DIR = -1 walks up (local j <-> node
// mI-1-j, offsets mirrored).  SPIKE: also carry the coupling to the separator
// behind the walk as MP*B extra right-hand sides and emit the V/W tips.
// STORE_U: keep the normalised pivot rows Ut (and Et when SPIKE and the spike response is
// stored) for the back-substitution -- the down walk all of them, or, with a.respike, each walk
// those of its half (tf_twist_h); STORE_Y: the down solve walk stores yt.
// KNOWN (tfk_l1_fwd2): the separator behind the walk is solved; the walk eliminates its half
// of the chunk with those values on the right-hand side and stores yt.
// ylds (KNOWN only, optional): y goes to the workgroup's LDS instead of a.yt, element (j, r) of this
// lane's walk at ylds[(j * B + r) * 64] (tfk_l1_fwd2_backsub: the back-substitution follows in the
// same launch).
// (YLDS is a flag, not a null test of the pointer: testing an LDS pointer against NULL trips
// hipcc 7.2 on some models, "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base")
// asm_stage / asm_rec (a.fuse_asm): where tf_asm_side puts this walk's part of a separator's row.
// ROLE = 1 (tfk_l1_factor*, two wavefronts per 64 chunks and direction): this walk eliminates the
// band only and publishes, per pivot, the inverse of the pivot block and the blocks below it to
// `xch` (LDS, element k of slot j & 1 at xch[(slot * NX + k) * 64]); the wavefront next to it
// (tfk_rhs_follow_body) carries every right-hand side with them -- the spike columns and the
// first right-hand side of the step -- and makes their tips (V, y); this one makes W.  One
// barrier per pivot.  (STORE_Y is false here: nothing uses y, the compiler drops it.)
template <class Rows, int DIR, bool SPIKE, bool STORE_U, bool STORE_Y, bool KNOWN = false, bool YLDS = false,
          int ROLE = 0>
TF_DEVICE void tfk_chunk_body(const TfLevelArgs& a, int pg, double* ylds = nullptr,
                              double* asm_stage = nullptr, int* asm_rec = nullptr, double* xch = nullptr) {
    constexpr int B = Rows::B, MP = Rows::MP, W = 2 * MP + 1;
    static_assert(!KNOWN || (!SPIKE && !STORE_U), "the re-elimination takes one right-hand side");
    static_assert(ROLE == 0 || (SPIKE && !STORE_Y && !Rows::PIVOT), "the split walk carries spike columns and exchanges no rows");
    constexpr bool ES = SPIKE && ROLE == 0;       // the spike columns are eliminated by this walk
    constexpr int NX = (1 + MP) * B * B;          // doubles published per pivot (ROLE 1)
    constexpr bool PIV = Rows::PIVOT;             // row exchanges inside the window (B == 1)
    constexpr int UW = PIV ? 2 * MP : MP;         // blocks right of the pivot kept in U
    static_assert(!PIV || B == 1, "row exchanges are written for scalar blocks");
    typedef TfTips<B, MP> Tip;
    const TfLayout& L = a.L;
    if (pg >= L.Ptot) return;
    constexpr int SB = KNOWN ? 0 : (SPIKE ? 10 : 20);   // stamps (diagnostic builds)
    TF_STAMP_L1(a, DIR, SB + 0);
    Rows rows(a, pg);
    const int len = rows.len;
    const int mI = len - MP;                      // interior nodes
    auto node = [&](int j) { return DIR > 0 ? j : mI - 1 - j; };

    // working window: R[q][c] = A(local row j+q, local col j+c)
    double R[MP + 1][W][B][B];
    double y[MP + 1][B];
    double Es[ES ? MP + 1 : 1][ES ? MP : 1][B][B];   // columns: separator behind, local order
    // normalised rows of the last MP pivots (tips)
    double Uh[MP][UW][B][B], yh[MP][B];
    double Eh[SPIKE ? MP : 1][SPIKE ? MP : 1][B][B];
    bool ok = true;
    // KNOWN: the separator behind the walk is solved; its values move to the right-hand side of
    // the first MP rows, the only ones that couple to it (local order: the node next to the
    // interior is the last one)
    double sa[KNOWN ? MP : 1][B];
    const int hdn = tf_twist_h<B, MP>(mI, a.twist);  // nodes of the down half (a.respike)
    if (KNOWN) {
        const int e = pg / L.P, p = pg - e * L.P;
        const bool has_sep = DIR < 0 || L.periodic || p > 0;
        const int ps = DIR < 0 ? p : (p > 0 ? p - 1 : L.P - 1);
        int p2, i2;
        tf_locate(a.Lnext, ps, p2, i2);
        const int64_t s2a = tf_idx(a.Lnext, e * a.Lnext.P + p2, i2);
#pragma unroll
        for (int t = 0; t < MP; ++t) {
            const int tn = DIR > 0 ? t : MP - 1 - t;
#pragma unroll
            for (int r = 0; r < B; ++r)
                sa[KNOWN ? t : 0][r] = has_sep ? a.xnext[tf_next_x(a, e, ps, s2a, tn * B + r, MP * B)] : 0.0;
        }
    }

    // Rows enter the window in local order 0, 1, 2, ...; their values are requested
    // two rows ahead of their use so that the walk does not wait on memory per node.
    struct Pre { typename Rows::Raw raw; double y[B]; };
    auto request = [&](int jl, Pre& pre) {
        if (jl < mI) {
            const int i = node(jl);
            rows.request(i, pre.raw);
            const unsigned off = tf_off8(L, pg, i);
#pragma unroll
            for (int r = 0; r < B; ++r) pre.y[r] = a.rhs ? tf_ldp(a.rhs, r, L.plane, off) : 0.0;
        }
    };
    auto install = [&](int q, int jl, const Pre& pre) {   // local row jl into window slot q (pivot j = jl - q)
#pragma unroll
        for (int c = 0; c < W; ++c) tf_blk_zero<B>(R[q][c]);
#pragma unroll
        for (int r = 0; r < B; ++r) y[q][r] = 0.0;
        if (ES) {
#pragma unroll
            for (int t = 0; t < MP; ++t) tf_blk_zero<B>(Es[ES ? q : 0][ES ? t : 0]);
        }
        if (jl < mI) {
            double row[W][B][B];
            rows.decode(node(jl), pre.raw, row);
#pragma unroll
            for (int r = 0; r < B; ++r) y[q][r] = pre.y[r];
#pragma unroll
            for (int d = -MP; d <= MP; ++d) {
                const int c = q + d;               // local column relative to the pivot
                const int dd = DIR > 0 ? d : -d;   // natural offset
                if (c >= 0) {
                    if (c < W) tf_blk_copy<B>(R[q][c], row[dd + MP]);
                } else if (ES) {
                    // column jl + d < 0: separator behind, local position MP + (jl + d)
                    const int t = MP + jl + d;
                    if (t >= 0 && t < MP) tf_blk_copy<B>(Es[ES ? q : 0][ES ? t : 0], row[dd + MP]);
                } else if (KNOWN) {
                    const int t = MP + jl + d;
                    if (t >= 0 && t < MP) tf_mv_sub<B>(y[q], row[dd + MP], sa[KNOWN ? t : 0]);
                }
            }
        }
    };
    // columns c < 0 only occur for the first MP local rows (jl + d < 0); there the
    // window slot q equals jl (pivot 0), so `c = q + d < 0` is exactly that case.
    // (rows in flight ahead of the walk: one.  Two measured slower for blocks, and
    // four for scalar models bought nothing -- config 2 19 973 -> 19 693 steps/s, N = 2e5 9 836 -> 8 457,
    // profiles/r04_ab_runs.txt: their walk is the dependent chain of a node's division and FMAs, not
    // the latency of its load)
    constexpr int PD = 1;
    Pre pre[PD] = {};
#pragma unroll
    for (int d = 0; d < PD; ++d) request(d, pre[d]);
    auto fetch = [&](int q, int jl) {              // jl == next_row: rows are taken in order
        install(q, jl, pre[0]);
#pragma unroll
        for (int d = 0; d + 1 < PD; ++d) pre[d] = pre[d + 1];
        request(jl + PD, pre[PD - 1]);
        TF_PIN_REQUESTS();                           // (the loads leave here, whatever the scheduler would like)
    };

#pragma unroll
    for (int q = 0; q < MP; ++q) fetch(q, q);
    TF_STAMP_L1(a, DIR, SB + 1);

    // one pivot; HIST >= 0 also records the normalised row as tip history slot HIST
    // (only the last MP pivots are recorded, in a peeled epilogue, so that the
    // history is not carried through the main loop)
    auto pivot = [&](int j, auto hist_tag) {
        constexpr int HIST = decltype(hist_tag)::value;
        // normalise the pivot row first, fetch the incoming row afterwards: the
        // B x B temporaries of the inverse and the new row are never live together
        double Un[UW][B][B], yn[B];
        double En[SPIKE ? MP : 1][B][B];
        if (PIV) {
            // every row with an entry in column j is a candidate: bring in row
            // j+MP first, then move the largest |entry| to slot 0
            fetch(MP, j + MP);
#pragma unroll
            for (int q = 1; q <= MP; ++q) {
                const bool sw = tf_abs(R[q][0][0][0]) > tf_abs(R[0][0][0][0]);
#pragma unroll
                for (int c = 0; c < W; ++c) {
                    const double u = R[0][c][0][0], v = R[q][c][0][0];
                    R[0][c][0][0] = sw ? v : u; R[q][c][0][0] = sw ? u : v;
                }
                { const double u = y[0][0], v = y[q][0]; y[0][0] = sw ? v : u; y[q][0] = sw ? u : v; }
                if (ES) {
#pragma unroll
                    for (int t = 0; t < MP; ++t) {
                        const double u = Es[0][ES ? t : 0][0][0], v = Es[ES ? q : 0][ES ? t : 0][0][0];
                        Es[0][ES ? t : 0][0][0] = sw ? v : u; Es[ES ? q : 0][ES ? t : 0][0][0] = sw ? u : v;
                    }
                }
            }
        }
        {
            double Dinv[B][B];
            ok = tf_blk_inverse<B>(R[0][0], Dinv) && ok;
#pragma unroll
            for (int c = 1; c <= UW; ++c) tf_mm<B>(Un[c - 1], Dinv, R[0][c]);
            tf_mv<B>(yn, Dinv, y[0]);
            if (ES) {
#pragma unroll
                for (int t = 0; t < MP; ++t) tf_mm<B>(En[t], Dinv, Es[0][ES ? t : 0]);
            }
            if (ROLE == 1) {
                double* slot = xch + (j & 1) * (NX * 64);
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int k = 0; k < B; ++k) slot[(r * B + k) * 64] = Dinv[r][k];
            }
        }
        if (!PIV) fetch(MP, j + MP);
        if (ROLE == 1) {
            // (the blocks below the pivot: the row that has just entered the window is one of them)
            double* slot = xch + (j & 1) * (NX * 64);
#pragma unroll
            for (int q = 1; q <= MP; ++q)
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int k = 0; k < B; ++k) slot[((q * B + r) * B + k) * 64] = R[q][0][r][k];
            TF_WG_BARRIER();
        }
#pragma unroll
        for (int q = 1; q <= MP; ++q) {
#pragma unroll
            for (int c = 1; c <= UW; ++c) tf_mm_sub<B>(R[q][c], R[q][0], Un[c - 1]);
            tf_mv_sub<B>(y[q], R[q][0], yn);
            if (ES) {
#pragma unroll
                for (int t = 0; t < MP; ++t) tf_mm_sub<B>(Es[ES ? q : 0][ES ? t : 0], R[q][0], En[ES ? t : 0]);
            }
        }
        // (a.respike: E and y of the first elimination are not kept, and each walk keeps the U of
        // its half, see tfk_l1_fwd2)
        const bool keep = KNOWN || !a.respike;
        const bool keep_u = a.respike ? (DIR > 0 ? j < hdn : j < mI - hdn) : DIR > 0;
        if (STORE_U || STORE_Y) {
            const unsigned off = tf_off8(L, pg, node(j));
            if (STORE_U && keep_u) {
#pragma unroll
                for (int c = 0; c < UW; ++c)
#pragma unroll
                    for (int r = 0; r < B; ++r)
#pragma unroll
                        for (int k = 0; k < B; ++k) {
                            tf_stp(a.Ut, (c * B + r) * B + k, L.plane, off, Un[c][r][k]);
                            if (ES && keep && c < MP)
                                tf_stp(a.Et, (c * B + r) * B + k, L.plane, off, En[ES && c < MP ? c : 0][r][k]);
                        }
            }
            if (STORE_Y && keep) {
                if (KNOWN && YLDS) {
#pragma unroll
                    for (int r = 0; r < B; ++r) ylds[(j * B + r) * 64] = yn[r];
                } else {
#pragma unroll
                    for (int r = 0; r < B; ++r) tf_stp(a.yt, r, L.plane, off, yn[r]);
                }
            }
        }
        if (HIST >= 0) {
            constexpr int H = HIST >= 0 ? HIST : 0;
#pragma unroll
            for (int c = 0; c < UW; ++c) tf_blk_copy<B>(Uh[H][c], Un[c]);
#pragma unroll
            for (int r = 0; r < B; ++r) yh[H][r] = yn[r];
            if (ES) {
#pragma unroll
                for (int t = 0; t < MP; ++t) tf_blk_copy<B>(Eh[ES ? H : 0][ES ? t : 0], En[ES ? t : 0]);
            }
        }
        // slide the window
#pragma unroll
        for (int q = 0; q < MP; ++q) {
#pragma unroll
            for (int c = 0; c < W - 1; ++c) tf_blk_copy<B>(R[q][c], R[q + 1][c + 1]);
            tf_blk_zero<B>(R[q][W - 1]);
#pragma unroll
            for (int r = 0; r < B; ++r) y[q][r] = y[q + 1][r];
            if (ES) {
#pragma unroll
                for (int t = 0; t < MP; ++t) tf_blk_copy<B>(Es[ES ? q : 0][ES ? t : 0], Es[ES ? q + 1 : 0][ES ? t : 0]);
            }
        }
    };
    if (KNOWN) {
        const int np = DIR > 0 ? hdn : mI - hdn;     // this walk's half
        for (int j = 0; j < np; ++j) {
            if (j == 4) TF_STAMP_L1(a, DIR, SB + 4);
            if (j == 5) TF_STAMP_L1(a, DIR, SB + 5);
            pivot(j, TfInt<-1>());
        }
        TF_STAMP_L1(a, DIR, SB + 2);
        if (!ok) *a.status = 1;
        return;
    }
    for (int j = 0; j < mI - MP; ++j) {
        if (j == 4) TF_STAMP_L1(a, DIR, SB + 4);
        if (j == 5) TF_STAMP_L1(a, DIR, SB + 5);
        pivot(j, TfInt<-1>());
    }
    TF_STAMP_L1(a, DIR, SB + 2);
    pivot(mI - MP, TfInt<0>());
    if (MP > 1) pivot(mI - MP + 1, TfInt<(MP > 1 ? 1 : 0)>());
    static_assert(MP <= 2, "tip epilogue is written for MP <= 2");
    if (ROLE == 1) {
        // the follower makes the V tips (tf_tips_out): it needs the U blocks that couple tip to tip.
        // They go to the slot that is not the last pivot's (read before that pivot's barrier);
        // the second barrier keeps this walk's staging (same LDS) behind the follower's last read.
        double* hand = xch + (mI & 1) * (NX * 64);
#pragma unroll
        for (int k = 0; k + 1 < MP; ++k)
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int c = 0; c < B; ++c) hand[((k * B + r) * B + c) * 64] = Uh[k][0][r][c];
        TF_WG_BARRIER();
        TF_WG_BARRIER();
    }
    tf_tips_out<Rows, DIR, SPIKE, ES, true, ROLE == 0>(a, rows, pg, Uh, yh, Eh, asm_stage, asm_rec);
    TF_STAMP_L1(a, DIR, SB + 3);
    if (!ok) *a.status = 1;
}

// The right-hand sides of a split factorisation walk (ROLE 1 above publishes per pivot j: the
// inverse of the pivot block and the MP blocks below it): the MP*B spike columns and the first
// right-hand side of the step.  Same products in the same order as the one-wavefront form:
// En = Dinv Es[0], Es[q] -= R[q][0] En (yn, y alike), the window slides.  Rows beyond the first
// MP do not couple to the separator behind: only those are read from the Jacobian; the
// right-hand side is requested one row ahead.
template <class Rows, int DIR, bool STORE_U, bool STORE_Y>
TF_DEVICE void tfk_rhs_follow_body(const TfLevelArgs& a, int pg, double* xch, double* asm_stage) {
    constexpr int B = Rows::B, MP = Rows::MP, W = 2 * MP + 1;
    constexpr int NX = (1 + MP) * B * B;
    const TfLayout& L = a.L;
    if (pg >= L.Ptot) return;
    if (DIR > 0) TF_STAMP_T(a, 30, 64);
    Rows rows(a, pg);
    const int mI = rows.len - MP;
    auto node = [&](int j) { return DIR > 0 ? j : mI - 1 - j; };
    double Es[MP + 1][MP][B][B], Eh[MP][MP][B][B], y[MP + 1][B], yh[MP][B], ypre[B];
    auto request = [&](int jl) {
#pragma unroll
        for (int r = 0; r < B; ++r)
            ypre[r] = (a.rhs && jl < mI) ? tf_ldp(a.rhs, r, L.plane, tf_off8(L, pg, node(jl < mI ? jl : 0))) : 0.0;
    };
#pragma unroll
    for (int q = 0; q <= MP; ++q) {
#pragma unroll
        for (int t = 0; t < MP; ++t) tf_blk_zero<B>(Es[q][t]);
#pragma unroll
        for (int r = 0; r < B; ++r) y[q][r] = 0.0;
    }
    request(0);
#pragma unroll
    for (int jl = 0; jl < MP; ++jl) {
#pragma unroll
        for (int r = 0; r < B; ++r) y[jl][r] = ypre[r];
        request(jl + 1);
        if (jl < mI) {
            double row[W][B][B];
            rows.load(node(jl), row);
#pragma unroll
            for (int d = -MP; d <= MP; ++d) {
                const int t = MP + jl + d;                 // column jl + d < 0: separator behind
                if (jl + d < 0 && t >= 0 && t < MP) tf_blk_copy<B>(Es[jl][t], row[(DIR > 0 ? d : -d) + MP]);
            }
        }
    }
    const int hdn = tf_twist_h<B, MP>(mI, a.twist);
    auto step = [&](int j, auto hist_tag) {
        constexpr int HIST = decltype(hist_tag)::value;
        TF_WG_BARRIER();
        const double* slot = xch + (j & 1) * (NX * 64);
        double Dinv[B][B], En[MP][B][B], yn[B];
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int k = 0; k < B; ++k) Dinv[r][k] = slot[(r * B + k) * 64];
        tf_mv<B>(yn, Dinv, y[0]);
#pragma unroll
        for (int t = 0; t < MP; ++t) tf_mm<B>(En[t], Dinv, Es[0][t]);
        // row j + MP enters the window
#pragma unroll
        for (int r = 0; r < B; ++r) y[MP][r] = ypre[r];
        request(j + MP + 1);
        TF_PIN_REQUESTS();
#pragma unroll
        for (int q = 1; q <= MP; ++q) {
            double M[B][B];
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int k = 0; k < B; ++k) M[r][k] = slot[((q * B + r) * B + k) * 64];
            tf_mv_sub<B>(y[q], M, yn);
#pragma unroll
            for (int t = 0; t < MP; ++t) tf_mm_sub<B>(Es[q][t], M, En[t]);
        }
        // (a.respike: E and y of the first elimination are not kept)
        const bool keep_u = a.respike ? (DIR > 0 ? j < hdn : j < mI - hdn) : DIR > 0;
        if ((STORE_U || STORE_Y) && !a.respike) {
            const unsigned off = tf_off8(L, pg, node(j));
            if (STORE_U && keep_u) {
#pragma unroll
                for (int c = 0; c < MP; ++c)
#pragma unroll
                    for (int r = 0; r < B; ++r)
#pragma unroll
                        for (int k = 0; k < B; ++k) tf_stp(a.Et, (c * B + r) * B + k, L.plane, off, En[c][r][k]);
            }
            if (STORE_Y) {
#pragma unroll
                for (int r = 0; r < B; ++r) tf_stp(a.yt, r, L.plane, off, yn[r]);
            }
        }
        if (HIST >= 0) {
            constexpr int H = HIST >= 0 ? HIST : 0;
#pragma unroll
            for (int t = 0; t < MP; ++t) tf_blk_copy<B>(Eh[H][t], En[t]);
#pragma unroll
            for (int r = 0; r < B; ++r) yh[H][r] = yn[r];
        }
#pragma unroll
        for (int q = 0; q < MP; ++q) {
#pragma unroll
            for (int t = 0; t < MP; ++t) tf_blk_copy<B>(Es[q][t], Es[q + 1][t]);
#pragma unroll
            for (int r = 0; r < B; ++r) y[q][r] = y[q + 1][r];
        }
#pragma unroll
        for (int t = 0; t < MP; ++t) tf_blk_zero<B>(Es[MP][t]);
    };
    if (DIR > 0) TF_STAMP_T(a, 31, 64);
    for (int j = 0; j < mI - MP; ++j) step(j, TfInt<-1>());
    step(mI - MP, TfInt<0>());
    if (MP > 1) step(mI - MP + 1, TfInt<(MP > 1 ? 1 : 0)>());
    if (DIR > 0) TF_STAMP_T(a, 32, 64);
    // the V and y tips are made here (tf_tips_out): the U blocks that couple tip to tip come from
    // the band walk, in the slot that is not the last pivot's
    double Uh[MP][1][B][B];
#pragma unroll
    for (int k = 0; k < MP; ++k) tf_blk_zero<B>(Uh[k][0]);
    TF_WG_BARRIER();
    const double* hand = xch + (mI & 1) * (NX * 64);
#pragma unroll
    for (int k = 0; k + 1 < MP; ++k)
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int c = 0; c < B; ++c) Uh[k][0][r][c] = hand[((k * B + r) * B + c) * 64];
    TF_WG_BARRIER();
    if (DIR > 0) TF_STAMP_T(a, 33, 64);
    tf_tips_out<Rows, DIR, true, true, false, true>(a, rows, pg, Uh, yh, Eh, asm_stage, nullptr);
    if (DIR > 0) TF_STAMP_T(a, 34, 64);
}

// ---- levels >= 2: block-tridiagonal chunks with stored factors --------------
// The reduced systems are small (P1 nodes of b x b blocks) but their blocks are
// big (b = MP*nvar), so here the factors ARE stored and the work is split into
//   tfk_bt_lu   one thread per (chunk, direction): block LU of the interior,
//               keeps Dinv_j and the normalised ahead block Un_j = Dinv_j * A(j, j+1)
//   tfk_bt_col  one thread per (chunk, direction, column): forward elimination
//               of ONE right-hand side through the stored factors -- a column of
//               the spike (factor phase) or the actual rhs (solve phase)
// which keeps every thread at a few b-vectors of registers.
template <int BB>
TF_DEVICE void tfk_bt_lu_body(const TfLevelArgs& a, int pg, int dir) {
    typedef TfTips<BB, 1> Tip;
    const TfLayout& L = a.L;
    if (pg >= L.Ptot) return;
    TfRowsBT<BB> rows(a, pg);
    const int mI = rows.len - 1;
    const int di = dir > 0 ? 0 : 1;
    bool ok = true;
    double S[BB][BB], Un[BB][BB];
    for (int j = 0; j < mI; ++j) {
        const int i = dir > 0 ? j : mI - 1 - j;
        const int64_t s = tf_idx(L, pg, i);
        double row[3][BB][BB];
        rows.load(i, row);
        tf_blk_copy<BB>(S, row[1]);
        if (j > 0) tf_mm_sub<BB>(S, row[dir > 0 ? 0 : 2], Un);         // S = dia - behind * Un_prev
        double Dinv[BB][BB];
        ok = tf_blk_inverse<BB>(S, Dinv) && ok;
        tf_mm<BB>(Un, Dinv, row[dir > 0 ? 2 : 0]);
        double* Uout = dir > 0 ? a.Ut : a.Unup;
#pragma unroll
        for (int r = 0; r < BB; ++r)
#pragma unroll
            for (int c = 0; c < BB; ++c) {
                a.Dinv[(int64_t)((di * BB + r) * BB + c) * L.plane + s] = Dinv[r][c];
                Uout[(int64_t)(r * BB + c) * L.plane + s] = Un[r][c];
            }
    }
    // response of the last pivot to the separator ahead
    double* tips = dir > 0 ? a.tips_dn : a.tips_up;
#pragma unroll
    for (int r = 0; r < BB; ++r)
#pragma unroll
        for (int c = 0; c < BB; ++c)
            tips[(int64_t)(dir > 0 ? Tip::W(0, 0, r, c) : Tip::V(0, 0, r, c)) * L.Ptot + pg] = Un[r][c];
    if (!ok) *a.status = 1;
}

// col < BB: spike column `col` (factor phase); col == BB: the right-hand side
template <int BB>
TF_DEVICE void tfk_bt_col_body(const TfLevelArgs& a, int pg, int dir, int col) {
    typedef TfTips<BB, 1> Tip;
    const TfLayout& L = a.L;
    if (pg >= L.Ptot) return;
    const int e = pg / L.P, p = pg - e * L.P;
    const int len = tf_len(L, p), mI = len - 1, start = tf_start(L, p);
    const int di = dir > 0 ? 0 : 1;
    const bool is_rhs = col >= BB;
    const int behind = dir > 0 ? 0 : 2;              // block of a row that couples backwards
    double ev[BB], en[BB];
#pragma unroll
    for (int r = 0; r < BB; ++r) { ev[r] = 0.0; en[r] = 0.0; }
    // per-node operands, requested one node ahead of their use
    struct Node { double Lb[BB][BB]; double Di[BB][BB]; double v[BB]; };
    auto load = [&](int j, Node& n) {
        const int i = dir > 0 ? j : mI - 1 - j;
        const int64_t s = tf_idx(L, pg, i);
#pragma unroll
        for (int r = 0; r < BB; ++r)
#pragma unroll
            for (int k = 0; k < BB; ++k) {
                n.Di[r][k] = a.Dinv[(int64_t)((di * BB + r) * BB + k) * L.plane + s];
                if (j > 0) n.Lb[r][k] = a.Ablk[(int64_t)((behind * BB + r) * BB + k) * L.plane + s];
            }
        if (is_rhs) {
#pragma unroll
            for (int r = 0; r < BB; ++r) n.v[r] = a.rhs[(int64_t)r * L.plane + s];
        } else if (j == 0) {
            // the backward coupling of the first row (absent at a non-periodic system end)
            const int g = start + i;
            const bool has_behind = L.periodic || (dir > 0 ? g > 0 : g < L.N - 1);
#pragma unroll
            for (int r = 0; r < BB; ++r)
                n.v[r] = has_behind ? a.Ablk[(int64_t)((behind * BB + r) * BB + col) * L.plane + s] : 0.0;
        } else {
#pragma unroll
            for (int r = 0; r < BB; ++r) n.v[r] = 0.0;
        }
    };
    Node cur, nxt;
    load(0, cur);
    for (int j = 0; j < mI; ++j) {
        if (j + 1 < mI) load(j + 1, nxt);
        const int i = dir > 0 ? j : mI - 1 - j;
        const int64_t s = tf_idx(L, pg, i);
#pragma unroll
        for (int r = 0; r < BB; ++r) {
            double acc = cur.v[r];
            if (j > 0) {
#pragma unroll
                for (int k = 0; k < BB; ++k) acc = tf_fma(-cur.Lb[r][k], en[k], acc);
            }
            ev[r] = acc;
        }
#pragma unroll
        for (int r = 0; r < BB; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < BB; ++k) acc = tf_fma(cur.Di[r][k], ev[k], acc);
            en[r] = acc;
        }
        if (dir > 0) {
            if (is_rhs) {
#pragma unroll
                for (int r = 0; r < BB; ++r) a.yt[(int64_t)r * L.plane + s] = en[r];
            } else {
#pragma unroll
                for (int r = 0; r < BB; ++r) a.Et[(int64_t)(r * BB + col) * L.plane + s] = en[r];
            }
        }
        cur = nxt;
    }
    double* tips = dir > 0 ? a.tips_dn : a.tips_up;
#pragma unroll
    for (int r = 0; r < BB; ++r) {
        const int slot = is_rhs ? Tip::y(0, r)
                                : (dir > 0 ? Tip::V(0, 0, r, col) : Tip::W(0, 0, r, col));
        tips[(int64_t)slot * L.Ptot + pg] = en[r];
    }
}

// ---- interface (separator) equations -> next level --------------------------
// MATRIX: build the [3][b][b] block row (factor phase); always builds the rhs.
// `stage` (optional): the block row goes there as [3][b][b] instead of to a.Anext -- the HIP
// entry point collects the rows of a workgroup in LDS and writes whole records (a thread's own
// 8-byte stores into records of 4*b*b doubles are the slowest thing this kernel can do).
template <class Rows, bool MATRIX>
TF_DEVICE void tfk_asm_body(const TfLevelArgs& a, int pg, double* stage = nullptr, int tsel = -1) {
    constexpr int B = Rows::B, MP = Rows::MP, W = 2 * MP + 1, BB = MP * B;
    typedef TfTips<B, MP> Tip;
    const TfLayout& L = a.L;
    if (pg >= L.Ptot) return;
    Rows rows(a, pg);
    const int e = rows.e, p = rows.p, len = rows.len, mI = len - MP;
    const bool has_next = L.periodic || p < L.P - 1;
    const int pn = e * L.P + (p < L.P - 1 ? p + 1 : 0);
    auto tdn = [&](int slot) { return a.tips_dn[(int64_t)slot * L.Ptot + pg]; };
    auto tup = [&](int slot) { return a.tips_up[(int64_t)slot * L.Ptot + pn]; };

    // position of this separator in the next level
    int p2, i2;
    tf_locate(a.Lnext, p, p2, i2);
    const int64_t s2 = tf_idx(a.Lnext, e * a.Lnext.P + p2, i2);

#pragma unroll
    for (int t = 0; t < MP; ++t) {                 // separator node t = chunk node mI + t
        if (tsel >= 0 && t != tsel) continue;      // (a thread per separator node: tfk_l1_asm_*)
        double row[W][B][B];
        rows.load(mI + t, row);
        double sub[MP][B][B], dia[MP][B][B], sup[MP][B][B], g[B];
#pragma unroll
        for (int t2 = 0; t2 < MP; ++t2) { tf_blk_zero<B>(sub[t2]); tf_blk_zero<B>(dia[t2]); tf_blk_zero<B>(sup[t2]); }
        const int64_t s = tf_idx(L, pg, mI + t);
#pragma unroll
        for (int r = 0; r < B; ++r) g[r] = a.rhs ? a.rhs[(int64_t)r * L.plane + s] : 0.0;
#pragma unroll
        for (int d = -MP; d <= MP; ++d) {
            const int cn = t + d;                  // column relative to the separator start
            if (cn >= 0 && cn < MP) {
                if (MATRIX) {
#pragma unroll
                    for (int r = 0; r < B; ++r)
#pragma unroll
                        for (int c = 0; c < B; ++c) dia[cn][r][c] += row[d + MP][r][c];
                }
            } else if (cn < 0) {                   // bottom tip node kb of the own interior
                const int kb = cn + MP;
                double yk[B];
#pragma unroll
                for (int r = 0; r < B; ++r) yk[r] = tdn(Tip::y(kb, r));
                tf_mv_sub<B>(g, row[d + MP], yk);
                if (MATRIX) {
#pragma unroll
                    for (int t2 = 0; t2 < MP; ++t2) {
                        double Vk[B][B], Wk[B][B];
#pragma unroll
                        for (int r = 0; r < B; ++r)
#pragma unroll
                            for (int c = 0; c < B; ++c) { Vk[r][c] = tdn(Tip::V(kb, t2, r, c)); Wk[r][c] = tdn(Tip::W(kb, t2, r, c)); }
                        tf_mm_sub<B>(sub[t2], row[d + MP], Vk);
                        tf_mm_sub<B>(dia[t2], row[d + MP], Wk);
                    }
                }
            } else if (has_next) {                 // top tip node kt of the next interior
                const int kt = cn - MP;
                double yk[B];
#pragma unroll
                for (int r = 0; r < B; ++r) yk[r] = tup(Tip::y(kt, r));
                tf_mv_sub<B>(g, row[d + MP], yk);
                if (MATRIX) {
#pragma unroll
                    for (int t2 = 0; t2 < MP; ++t2) {
                        double Vk[B][B], Wk[B][B];
#pragma unroll
                        for (int r = 0; r < B; ++r)
#pragma unroll
                            for (int c = 0; c < B; ++c) { Vk[r][c] = tup(Tip::V(kt, t2, r, c)); Wk[r][c] = tup(Tip::W(kt, t2, r, c)); }
                        tf_mm_sub<B>(dia[t2], row[d + MP], Vk);
                        tf_mm_sub<B>(sup[t2], row[d + MP], Wk);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < B; ++r) a.rhsnext[tf_next_rhs(a, e, p, s2, t * B + r, BB)] = g[r];
        if (MATRIX) {
#pragma unroll
            for (int t2 = 0; t2 < MP; ++t2)
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int c = 0; c < B; ++c) {
                        const int rr = t * B + r, cc = t2 * B + c;
                        if (stage) {
                            stage[(0 * BB + rr) * BB + cc] = sub[t2][r][c];
                            stage[(1 * BB + rr) * BB + cc] = dia[t2][r][c];
                            stage[(2 * BB + rr) * BB + cc] = sup[t2][r][c];
                        } else {
                            a.Anext[tf_next_A(a, e, p, s2, 0, rr, cc, BB)] = sub[t2][r][c];
                            a.Anext[tf_next_A(a, e, p, s2, 1, rr, cc, BB)] = dia[t2][r][c];
                            a.Anext[tf_next_A(a, e, p, s2, 2, rr, cc, BB)] = sup[t2][r][c];
                        }
                    }
        }
    }
}

// ---- back-substitution of one chunk (separators known) ----------------------
template <class Rows, bool WITH_E = true>
TF_DEVICE void tfk_backsub_body(const TfLevelArgs& a, int pg) {
    constexpr int B = Rows::B, MP = Rows::MP;
    constexpr int UW = Rows::PIVOT ? 2 * MP : MP;  // see tfk_chunk_body
    const TfLayout& L = a.L;
    if (pg >= L.Ptot) return;
    const int e = pg / L.P, p = pg - e * L.P;
    const int len = tf_len(L, p), mI = len - MP;
    const bool has_above = L.periodic || p > 0;
    const int pa = p > 0 ? p - 1 : L.P - 1;
    double sa[MP][B], xn[UW][B];
#pragma unroll
    for (int t = MP; t < UW; ++t)                  // beyond the separator: coefficient is an exact zero
#pragma unroll
        for (int r = 0; r < B; ++r) xn[t][r] = 0.0;
    {
        int p2, i2;
        tf_locate(a.Lnext, p, p2, i2);
        const int64_t s2 = tf_idx(a.Lnext, e * a.Lnext.P + p2, i2);
#pragma unroll
        for (int t = 0; t < MP; ++t)
#pragma unroll
            for (int r = 0; r < B; ++r) xn[t][r] = a.xnext[tf_next_x(a, e, p, s2, t * B + r, MP * B)];
        tf_locate(a.Lnext, pa, p2, i2);
        const int64_t s2a = tf_idx(a.Lnext, e * a.Lnext.P + p2, i2);
#pragma unroll
        for (int t = 0; t < MP; ++t)
#pragma unroll
            for (int r = 0; r < B; ++r) sa[t][r] = has_above ? a.xnext[tf_next_x(a, e, pa, s2a, t * B + r, MP * B)] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < MP; ++t) {
        const int64_t s = tf_idx(L, pg, mI + t);
#pragma unroll
        for (int r = 0; r < B; ++r) a.x[(int64_t)r * L.plane + s] = xn[t][r];
    }
    // the factors of node j-1 are requested before node j is processed, so one
    // HBM latency is paid per chunk, not per node
    struct Node { double y[B]; double U[UW][B][B]; double E[WITH_E ? MP : 1][B][B]; };
    auto load = [&](int j, Node& n) {
        const unsigned off = tf_off8(L, pg, j);
#pragma unroll
        for (int r = 0; r < B; ++r) n.y[r] = tf_ldp(a.yt, r, L.plane, off);
#pragma unroll
        for (int c = 0; c < UW; ++c)
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    n.U[c][r][k] = tf_ldp(a.Ut, (c * B + r) * B + k, L.plane, off);
                    if (WITH_E && c < MP) n.E[WITH_E && c < MP ? c : 0][r][k] = tf_ldp(a.Et, (c * B + r) * B + k, L.plane, off);
                }
    };
    // TF_BACKSUB_DEPTH nodes of factors in flight per thread: the walk is a chain of loads
    // (few wavefronts: one lane per chunk), so its speed is the number of requests that are
    // outstanding, not the arithmetic.  The ring is unrolled: every index is static.
    constexpr int D = tf_ring_depth(B + UW * B * B + (WITH_E ? MP * B * B : 0));
    Node ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (mI - 1 - d >= 0) load(mI - 1 - d, ring[d]);
    for (int j0 = mI - 1; j0 >= 0; j0 -= D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int j = j0 - d;
            if (j < 0) break;
            const Node& cur = ring[d];
            const int64_t s = tf_idx(L, pg, j);
            double x[B];
#pragma unroll
            for (int r = 0; r < B; ++r) x[r] = cur.y[r];
#pragma unroll
            for (int c = 0; c < UW; ++c) tf_mv_sub<B>(x, cur.U[c], xn[c]);
            if (WITH_E) {
#pragma unroll
                for (int c = 0; c < MP; ++c) tf_mv_sub<B>(x, cur.E[WITH_E ? c : 0], sa[c]);
            }
#pragma unroll
            for (int c = UW - 1; c > 0; --c)
#pragma unroll
                for (int r = 0; r < B; ++r) xn[c][r] = xn[c - 1][r];
#pragma unroll
            for (int r = 0; r < B; ++r) { xn[0][r] = x[r]; a.x[(int64_t)r * L.plane + s] = x[r]; }
            if (j - D >= 0) load(j - D, ring[d]);
        }
    }
}

// n x n system with partial pivoting, everything in registers (n <= 6: the middle of a chunk)
template <int n>
TF_DEVICE bool tf_dense_solve(double (&S)[n][n], double (&g)[n]) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < n; ++k) {
#pragma unroll
        for (int r = k + 1; r < n; ++r) {            // largest |entry| of column k to row k
            const bool sw = tf_abs(S[r][k]) > tf_abs(S[k][k]);
#pragma unroll
            for (int c = k; c < n; ++c) { const double u = S[k][c], v = S[r][c]; S[k][c] = sw ? v : u; S[r][c] = sw ? u : v; }
            { const double u = g[k], v = g[r]; g[k] = sw ? v : u; g[r] = sw ? u : v; }
        }
        const double piv = S[k][k];
        ok = ok && (piv != 0.0) && (piv == piv);
        const double rp = 1.0 / piv;
#pragma unroll
        for (int c = k + 1; c < n; ++c) S[k][c] = S[k][c] * rp;
        g[k] = g[k] * rp;
#pragma unroll
        for (int r = k + 1; r < n; ++r) {
            const double f = S[r][k];
#pragma unroll
            for (int c = k + 1; c < n; ++c) S[r][c] = tf_fma(-f, S[k][c], S[r][c]);
            g[r] = tf_fma(-f, g[k], g[r]);
        }
    }
#pragma unroll
    for (int k = n - 2; k >= 0; --k)
#pragma unroll
        for (int c = k + 1; c < n; ++c) g[k] = tf_fma(-S[k][c], g[c], g[k]);
    return ok;
}

// Back-substitution of the re-elimination form (a.respike; tf_twist_h): dir 0 takes the down
// half of the chunk, dir 1 the up half.  Both first solve the 2*MP nodes where the halves meet
//   down rows r = h-MP..h-1:  x_r + sum_c U_r[c]  x_{r+1+c} = y_r
//   up rows   r = h..h+MP-1:  x_r + sum_c U'_r[c] x_{r-1-c} = y'_r
// (a = the down unknowns, b = the up unknowns: T1 a + C1 b = g1, C2 a + T2 b = g2 with unit
// triangular T1, T2; b from the Schur complement T2 - C2 T1^-1 C1, pivoted; the same arithmetic
// in both threads) and then stream their own half outwards like tfk_backsub_body.
// ylds_dn / ylds_up (optional): y of the two walks comes from the workgroup's LDS (see
// tfk_chunk_body) instead of a.yt.
template <class Rows, bool YLDS = false>
TF_DEVICE void tfk_backsub_twist_body(const TfLevelArgs& a, int pg, int dir,
                                      const double* ylds_dn = nullptr, const double* ylds_up = nullptr) {
    constexpr int B = Rows::B, MP = Rows::MP, NB = MP * B;
    static_assert(!Rows::PIVOT, "the re-elimination form is for block sizes that do not exchange rows");
    const TfLayout& L = a.L;
    if (pg >= L.Ptot) return;
    const int e = pg / L.P, p = pg - e * L.P;
    const int len = tf_len(L, p), mI = len - MP;
    const int h = tf_twist_h<B, MP>(mI, a.twist), hu = mI - h;
    if (dir == 1 && hu == 0) return;
    double xn[MP][B];                              // the MP solved nodes ahead of the walk, nearest first
    // local index (walking direction) of the first node of the streamed part: one-sided, ahead of
    // the last pivot lies the separator; twisted, the MP nodes next to the middle are solved there
    const int jstart = hu == 0 ? mI - 1 : (dir == 0 ? h - MP - 1 : hu - MP - 1);
    auto nat = [&](int j) { return dir == 0 ? j : mI - 1 - j; };
    auto ldU = [&](int node_nat, double (&U)[MP][B][B], double (&y)[B]) {
        const unsigned off = tf_off8(L, pg, node_nat);
        if (YLDS) {
            // the down walk numbers its nodes from the top of the chunk, the up walk from the bottom
            const double* src = node_nat < h ? ylds_dn + node_nat * B * 64 : ylds_up + (mI - 1 - node_nat) * B * 64;
#pragma unroll
            for (int r = 0; r < B; ++r) y[r] = src[r * 64];
        } else {
#pragma unroll
            for (int r = 0; r < B; ++r) y[r] = tf_ldp(a.yt, r, L.plane, off);
        }
#pragma unroll
        for (int c = 0; c < MP; ++c)
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int k = 0; k < B; ++k) U[c][r][k] = tf_ldp(a.Ut, (c * B + r) * B + k, L.plane, off);
    };
    // a.upd_n (the launch that holds the re-elimination, YLDS): the solve is the last one of a time
    // step and what leaves is the new state, base + c0 x (one term) or base + (c0 k0 + c1 x) -- the
    // operations of tfk_vec in their order -- instead of x.  The other operands of a node are
    // requested ahead of their use like its factors.
    struct Upd { double b[B], k[B]; };
    const bool upd = YLDS && a.upd_n > 0;
    auto upd_load = [&](int nd, Upd& u) {
        if (upd) {
            const unsigned off = tf_off8(L, pg, nd);
#pragma unroll
            for (int r = 0; r < B; ++r) {
                u.b[r] = tf_ldp(a.upd_base, r, L.plane, off);
                u.k[r] = a.upd_n > 1 ? tf_ldp(a.upd_k0, r, L.plane, off) : 0.0;
            }
        }
    };
    auto emit = [&](int nd, const double (&x)[B], const Upd& u) {     // the solution at node `nd` of the chunk
        const int64_t s = tf_idx(L, pg, nd);
        if (upd) {
#pragma unroll
            for (int r = 0; r < B; ++r) {
                double acc;
                if (a.upd_n > 1) { acc = a.upd_c0 * u.k[r]; acc = acc + a.upd_c1 * x[r]; }
                else acc = a.upd_c0 * x[r];
                a.upd_out[(int64_t)r * L.plane + s] = u.b[r] + acc;
            }
        } else {
#pragma unroll
            for (int r = 0; r < B; ++r) a.x[(int64_t)r * L.plane + s] = x[r];
        }
    };
    // the separator and the nodes next to the middle leave at the end (their other operands are
    // requested once the middle system is solved, which needs the registers)
    double xsep[MP][B], xmid[MP][B];
    const bool own_sep = hu == 0 || dir == 0;
    if (own_sep) {
        // the chunk's own separator: solved by the next level
        int p2, i2;
        tf_locate(a.Lnext, p, p2, i2);
        const int64_t s2 = tf_idx(a.Lnext, e * a.Lnext.P + p2, i2);
#pragma unroll
        for (int t = 0; t < MP; ++t) {
#pragma unroll
            for (int r = 0; r < B; ++r) {
                const double v = a.xnext[tf_next_x(a, e, p, s2, t * B + r, MP * B)];
                xn[t][r] = v;
                xsep[t][r] = v;
            }
        }
    }
    // the factors of the first streamed nodes are requested before the middle system is solved
    struct Node { double y[B]; double U[MP][B][B]; Upd u; };
    auto load = [&](int j, Node& n) { ldU(nat(j), n.U, n.y); upd_load(nat(j), n.u); };
    constexpr int D = tf_ring_depth(3 * B + MP * B * B, 2 * (MP * B) * (MP * B) + 2 * MP * B + MP * B * B + B);
    Node ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (jstart - d >= 0) load(jstart - d, ring[d]);
    if (hu != 0) {
        double Tc[NB][NB], Tg[NB];                 // T1^-1 [C1 | g1], rows = down unknowns
        double S[NB][NB], g2[NB];
        double Uk[MP][B][B], yk[B];
        // rows of the down half, last first, substituted into each other (T1 is unit upper triangular)
#pragma unroll
        for (int k = MP - 1; k >= 0; --k) {
            ldU(h - MP + k, Uk, yk);
#pragma unroll
            for (int i = 0; i < B; ++i) {
                Tg[k * B + i] = yk[i];
#pragma unroll
                for (int c = 0; c < NB; ++c) { Tc[k * B + i][c] = 0.0; }
            }
#pragma unroll
            for (int c = 0; c < MP; ++c) {
                const int u = k + 1 + c;           // unknown index over (a, b)
                if (u >= MP) {
#pragma unroll
                    for (int i = 0; i < B; ++i)
#pragma unroll
                        for (int jj = 0; jj < B; ++jj) Tc[k * B + i][(u - MP) * B + jj] += Uk[c][i][jj];
                } else {
                    // x_u of the down half: replace it by its own row, Tg_u - Tc_u b
#pragma unroll
                    for (int i = 0; i < B; ++i)
#pragma unroll
                        for (int jj = 0; jj < B; ++jj) {
                            const double f = Uk[c][i][jj];
                            Tg[k * B + i] = tf_fma(-f, Tg[u * B + jj], Tg[k * B + i]);
#pragma unroll
                            for (int cc = 0; cc < NB; ++cc)
                                Tc[k * B + i][cc] = tf_fma(-f, Tc[u * B + jj][cc], Tc[k * B + i][cc]);
                        }
                }
            }
        }
        // a = Tg - Tc b.  Rows of the up half: x_r + sum_c U'_r[c] x_{r-1-c} = y'_r
#pragma unroll
        for (int k = 0; k < MP; ++k) {
            ldU(h + k, Uk, yk);
#pragma unroll
            for (int i = 0; i < B; ++i) {
                g2[k * B + i] = yk[i];
#pragma unroll
                for (int c = 0; c < NB; ++c) S[k * B + i][c] = (c == k * B + i) ? 1.0 : 0.0;
            }
#pragma unroll
            for (int c = 0; c < MP; ++c) {
                const int u = MP + k - 1 - c;      // unknown index over (a, b)
                if (u >= MP) {
#pragma unroll
                    for (int i = 0; i < B; ++i)
#pragma unroll
                        for (int jj = 0; jj < B; ++jj) S[k * B + i][(u - MP) * B + jj] += Uk[c][i][jj];
                } else if (u >= 0) {
#pragma unroll
                    for (int i = 0; i < B; ++i)
#pragma unroll
                        for (int jj = 0; jj < B; ++jj) {
                            const double f = Uk[c][i][jj];
                            g2[k * B + i] = tf_fma(-f, Tg[u * B + jj], g2[k * B + i]);
#pragma unroll
                            for (int cc = 0; cc < NB; ++cc)
                                S[k * B + i][cc] = tf_fma(-f, Tc[u * B + jj][cc], S[k * B + i][cc]);
                        }
                }
            }
        }
        TF_STAMP_L1(a, 1 - 2 * dir, 6);
        if (!tf_dense_solve<NB>(S, g2)) *a.status = 1;
        TF_STAMP_L1(a, 1 - 2 * dir, 7);
        // g2 = b; a = Tg - Tc b
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int c = 0; c < NB; ++c) Tg[i] = tf_fma(-Tc[i][c], g2[c], Tg[i]);
        if (dir == 0) {
#pragma unroll
            for (int k = 0; k < MP; ++k)
#pragma unroll
                for (int r = 0; r < B; ++r) { xmid[k][r] = Tg[k * B + r]; xn[k][r] = Tg[k * B + r]; }
            // (ahead of node h-MP-1: h-MP (a_0), h-MP+1 (a_1), ...)
        } else {
#pragma unroll
            for (int k = 0; k < MP; ++k)
#pragma unroll
                for (int r = 0; r < B; ++r) { xmid[k][r] = g2[k * B + r]; xn[MP - 1 - k][r] = g2[k * B + r]; }
            // (ahead of node h+MP, walking up: h+MP-1 (b_{MP-1}), ...)
        }
    }
    Upd usep[MP], umid[MP];
#pragma unroll
    for (int t = 0; t < MP; ++t) {
        if (own_sep) upd_load(mI + t, usep[t]);
        if (hu != 0) upd_load(dir == 0 ? h - MP + t : h + t, umid[t]);
    }
    for (int j0 = jstart; j0 >= 0; j0 -= D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int j = j0 - d;
            if (j < 0) break;
            const Node& cur = ring[d];
            double x[B];
#pragma unroll
            for (int r = 0; r < B; ++r) x[r] = cur.y[r];
#pragma unroll
            for (int c = 0; c < MP; ++c) tf_mv_sub<B>(x, cur.U[c], xn[c]);
#pragma unroll
            for (int c = MP - 1; c > 0; --c)
#pragma unroll
                for (int r = 0; r < B; ++r) xn[c][r] = xn[c - 1][r];
#pragma unroll
            for (int r = 0; r < B; ++r) xn[0][r] = x[r];
            emit(nat(j), x, cur.u);
            if (j - D >= 0) load(j - D, ring[d]);
        }
    }
#pragma unroll
    for (int t = 0; t < MP; ++t) {
        if (own_sep) emit(mI + t, xsep[t], usep[t]);
        if (hu != 0) emit(dir == 0 ? h - MP + t : h + t, xmid[t], umid[t]);
    }
}

// ---- grids shorter than one stencil window: dense, one thread per system -----------
#define TF_TINY_MAXN (2 * TF_MP * TF_NVAR)        // N <= 2*mp nodes
TF_DEVICE void tfk_tiny_factor_body(const TfTinyArgs& a, int e) {
    const TfLayout& L = a.L;
    if (e >= L.nsys) return;
    const int N = L.N, n = N * TF_NVAR;
    double* A = a.lu + (int64_t)e * n * n;
    int* piv = a.piv + (int64_t)e * n;
    for (int i = 0; i < n * n; ++i) A[i] = 0.0;
    TfJUniform ju;
    ju.init(a.parsca, a.dx, L.nsys, e);
    for (int i = 0; i < N; ++i) {
        const int64_t s = tf_idx(L, e, i);             // (P == 1: chunk index = system index)
        double jr[TF_NNZ > 0 ? TF_NNZ : 1];
        ju.row([&](int k) { return a.Jv[(int64_t)k * L.plane + s]; }, jr);
        for (int k = 0; k < TF_NNZ; ++k) {
            int j = i + tf_pat_off[k];
            if (L.periodic) j = ((j % N) + N) % N;
            else j = j < 0 ? 0 : (j > N - 1 ? N - 1 : j);
            A[(i * TF_NVAR + tf_pat_eq[k]) * n + j * TF_NVAR + tf_pat_var[k]] += -a.c * jr[k];
        }
        for (int v = 0; v < TF_NVAR; ++v) A[(i * TF_NVAR + v) * (n + 1)] += 1.0;
    }
    bool ok = true;
    for (int k = 0; k < n; ++k) {                      // LU with partial pivoting, in place
        int p = k;
        for (int r = k + 1; r < n; ++r) if (tf_abs(A[r * n + k]) > tf_abs(A[p * n + k])) p = r;
        piv[k] = p;
        if (p != k)
            for (int c2 = 0; c2 < n; ++c2) { const double t = A[k * n + c2]; A[k * n + c2] = A[p * n + c2]; A[p * n + c2] = t; }
        const double d = A[k * n + k];
        ok = ok && d != 0.0 && tf_finite(d);
        const double rd = 1.0 / d;
        for (int r = k + 1; r < n; ++r) {
            const double f = A[r * n + k] * rd;
            A[r * n + k] = f;
            for (int c2 = k + 1; c2 < n; ++c2) A[r * n + c2] = tf_fma(-f, A[k * n + c2], A[r * n + c2]);
        }
    }
    if (!ok) *a.status = 1;
}
TF_DEVICE void tfk_tiny_solve_body(const TfTinyArgs& a, int e) {
    const TfLayout& L = a.L;
    if (e >= L.nsys) return;
    const int N = L.N, n = N * TF_NVAR;
    const double* A = a.lu + (int64_t)e * n * n;
    const int* piv = a.piv + (int64_t)e * n;
    double y[TF_TINY_MAXN];
    for (int i = 0; i < N; ++i)
        for (int v = 0; v < TF_NVAR; ++v) y[i * TF_NVAR + v] = a.rhs[(int64_t)v * L.plane + tf_idx(L, e, i)];
    for (int k = 0; k < n; ++k) {
        const int p = piv[k];
        if (p != k) { const double t = y[k]; y[k] = y[p]; y[p] = t; }
        for (int r = k + 1; r < n; ++r) y[r] = tf_fma(-A[r * n + k], y[k], y[r]);
    }
    for (int k = n - 1; k >= 0; --k) {
        double acc = y[k];
        for (int c2 = k + 1; c2 < n; ++c2) acc = tf_fma(-A[k * n + c2], y[c2], acc);
        y[k] = acc / A[k * n + k];
    }
    for (int i = 0; i < N; ++i)
        for (int v = 0; v < TF_NVAR; ++v) a.x[(int64_t)v * L.plane + tf_idx(L, e, i)] = y[i * TF_NVAR + v];
}

// ---- last level: one b x b block per system ---------------------------------
template <int BB, bool FACTOR>
TF_DEVICE void tfk_top_body(const TfTopArgs& a, int e) {
    if (e >= a.nsys) return;
    // aos: records per system written by a cyclic-reduction level (TfLevelArgs)
    auto A = [&](int blk, int r, int c) {
        return a.aos ? a.A[(((int64_t)e * 4 + blk) * BB + r) * BB + c]
                     : a.A[(int64_t)((blk * BB + r) * BB + c) * a.nsys + e];
    };
    if (FACTOR) {
        double D[BB][BB], Di[BB][BB];
#pragma unroll
        for (int r = 0; r < BB; ++r)
#pragma unroll
            for (int c = 0; c < BB; ++c) {
                D[r][c] = A(0, r, c) + A(1, r, c) + A(2, r, c);
                if (a.aos) D[r][c] += A(3, r, c);
            }
        if (!tf_blk_inverse<BB>(D, Di)) *a.status = 1;
#pragma unroll
        for (int r = 0; r < BB; ++r)
#pragma unroll
            for (int c = 0; c < BB; ++c) a.Ainv[(int64_t)(r * BB + c) * a.nsys + e] = Di[r][c];
    } else {
        double Di[BB][BB], g[BB], x[BB];
#pragma unroll
        for (int r = 0; r < BB; ++r) {
            g[r] = a.aos ? a.rhs[(int64_t)e * 2 * BB + r] + a.rhs[(int64_t)e * 2 * BB + BB + r]
                         : a.rhs[(int64_t)r * a.nsys + e];
#pragma unroll
            for (int c = 0; c < BB; ++c) Di[r][c] = a.Ainv[(int64_t)(r * BB + c) * a.nsys + e];
        }
        tf_mv<BB>(x, Di, g);
#pragma unroll
        for (int r = 0; r < BB; ++r) a.x[a.aos ? (int64_t)e * BB + r : (int64_t)r * a.nsys + e] = x[r];
    }
}
