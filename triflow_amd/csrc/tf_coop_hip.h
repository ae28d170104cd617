// Wave-cooperative kernels for the reduced (block-tridiagonal) solver levels.
//
// The reduced systems have few chunks but b x b blocks with b = mp*nvar (6 for
// the film model), and a chunk has to be walked node after node.  One thread
// per chunk (tfk_bt_*_body in tf_kernels.h) spends ~2000 dependent fp64
// instructions per node; here a group of G = 8 (b <= 8) or 16 (b <= 16) lanes
// shares a chunk instead: lane g owns row g of every block, the pieces that
// every lane needs (pivot row, previous Un, intermediate vectors) go through a
// few hundred bytes of LDS.  Workgroups are single wavefronts (64 threads), so
// the barriers between the LDS phases cost nothing.
//
// Same arithmetic as the one-thread-per-chunk bodies (Gauss-Jordan with
// partial pivoting inside the block, identical operation order per row), which
// remain the version the host emulation runs.  HIP only.
#pragma once
#include "tf_crs.h"

template <int BB> struct TfCoop {
    static constexpr int G = BB <= 2 ? 1 : (BB <= 8 ? 8 : (BB <= 16 ? 16 : 1));
    static constexpr int NGRP = G > 1 ? 64 / G : 64;       // chunks per 64-thread block
};

// ---- block LU of the chunk interiors ------------------------------------------
// Lane g starts with row g of S = dia - behind*Un_prev and of the identity.  Rows
// are never moved: the lane that holds the largest |S[.][k]| among the lanes not
// yet used becomes pivot k (the same row the one-thread version swaps into place),
// publishes its raw row through LDS, and every lane normalises it redundantly and
// eliminates column k from its own row.  Two barriers per pivot.
template <int BB>
__device__ __forceinline__ void tfk_bt_lu_coop(const TfLevelArgs& a, int dir) {
    typedef TfTips<BB, 1> Tip;
    constexpr int G = TfCoop<BB>::G, NGRP = TfCoop<BB>::NGRP;
    const TfLayout& L = a.L;
    const int grp = threadIdx.x / G, g = threadIdx.x % G;
    const int pg = blockIdx.x * NGRP + grp;
    const bool lane_on = pg < L.Ptot && g < BB;
    const int e = lane_on ? pg / L.P : 0, p = lane_on ? pg - e * L.P : 0;
    const int len = tf_len(L, p), mI = len - 1, start = tf_start(L, p);
    const int di = dir > 0 ? 0 : 1;
    const int behind = dir > 0 ? 0 : 2, ahead = dir > 0 ? 2 : 0;

    __shared__ double sUn[NGRP][BB][BB];
    __shared__ double sAh[NGRP][BB][BB];
    __shared__ double sAll[2][NGRP][G][2 * BB + 1];   // every lane's row [S | INV | |S[k]|], double buffered
    __shared__ double sE[NGRP][BB][BB + 1];          // spike / rhs columns: en of the previous node
    __shared__ double sV[NGRP][BB][BB + 1];          //                      ev of this node
    const int ncols = a.lu_cols;                     // 0, BB or BB + 1

    struct Row { double dia[BB], ah[BB], bh[BB], rhs; };
    auto load = [&](int j, Row& r) {
        const int i = dir > 0 ? j : mI - 1 - j;
        const int64_t s = tf_idx(L, pg, i);
        const int gn = start + i;
        const bool cut_b = !L.periodic && (dir > 0 ? gn == 0 : gn == L.N - 1);
        const bool cut_a = !L.periodic && (dir > 0 ? gn == L.N - 1 : gn == 0);
#pragma unroll
        for (int c = 0; c < BB; ++c) {
            r.dia[c] = a.Ablk[(int64_t)((1 * BB + g) * BB + c) * L.plane + s];
            r.ah[c] = cut_a ? 0.0 : a.Ablk[(int64_t)((ahead * BB + g) * BB + c) * L.plane + s];
            r.bh[c] = cut_b ? 0.0 : a.Ablk[(int64_t)((behind * BB + g) * BB + c) * L.plane + s];
        }
        r.rhs = ncols > BB ? a.rhs[(int64_t)g * L.plane + s] : 0.0;
    };
    Row cur, nxt;
    if (lane_on && mI > 0) load(0, cur);

    double S[BB], INV[BB], AHlast[BB], ENlast[BB + 1];
#pragma unroll
    for (int c = 0; c < BB; ++c) AHlast[c] = 0.0;
#pragma unroll
    for (int c = 0; c <= BB; ++c) ENlast[c] = 0.0;
    bool ok = true;
    int myk_last = 0;
    const int rounds = L.M - 1;                      // uniform over the block
    for (int j = 0; j < rounds; ++j) {
        const bool on = lane_on && j < mI;
        if (lane_on && j + 1 < mI) load(j + 1, nxt);   // next node in flight during the elimination
        if (on) {
#pragma unroll
            for (int c = 0; c < BB; ++c) {
                double acc = cur.dia[c];
                if (j > 0) {                         // S = dia - behind * Un_prev
#pragma unroll
                    for (int k = 0; k < BB; ++k) acc = tf_fma(-cur.bh[k], sUn[grp][k][c], acc);
                }
                S[c] = acc;
                INV[c] = c == g ? 1.0 : 0.0;
                sAh[grp][g][c] = cur.ah[c];
            }
        }
        int myk = -1;                                // pivot index this lane ended up serving
        // Before each pivot every lane publishes its current row and its candidate
        // magnitude; after ONE barrier everybody knows the pivot lane and has its row.
        auto publish = [&](int k) {
            double* dst = sAll[k & 1][grp][g];
#pragma unroll
            for (int c = 0; c < BB; ++c) { dst[c] = S[c]; dst[BB + c] = INV[c]; }
            dst[2 * BB] = (on && myk < 0 && g < BB) ? tf_abs(S[k]) : -1.0;
        };
        publish(0);
        __syncthreads();                             // also: everybody has read sUn
#pragma unroll
        for (int k = 0; k < BB; ++k) {
            int piv = 0;
            double best = -2.0;
#pragma unroll
            for (int r = 0; r < BB; ++r) {           // largest |S[.][k]| among unused lanes (first on ties)
                const double v = sAll[k & 1][grp][r][2 * BB];
                if (v > best) { best = v; piv = r; }
            }
            if (on) {
                const double* prow = sAll[k & 1][grp][piv];
                const double pv = prow[k];
                ok = ok && (pv != 0.0) && tf_finite(pv);
                const double rp = 1.0 / pv;
                if (g == piv) {
                    myk = k;
#pragma unroll
                    for (int c = 0; c < BB; ++c) { S[c] *= rp; INV[c] *= rp; }
                } else {
                    const double f = S[k];
#pragma unroll
                    for (int c = 0; c < BB; ++c) {
                        S[c] = tf_fma(-f, prow[c] * rp, S[c]);
                        INV[c] = tf_fma(-f, prow[BB + c] * rp, INV[c]);
                    }
                }
            }
            if (k + 1 < BB) publish(k + 1);
            __syncthreads();
        }
        if (on) {                                    // INV = row myk of S^-1;  Un = S^-1 * ahead
            const int64_t s = tf_idx(L, pg, dir > 0 ? j : mI - 1 - j);
            double* Uout = dir > 0 ? a.Ut : a.Unup;
#pragma unroll
            for (int c = 0; c < BB; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < BB; ++k) acc = tf_fma(INV[k], sAh[grp][k][c], acc);
                a.Dinv[(int64_t)((di * BB + myk) * BB + c) * L.plane + s] = INV[c];
                Uout[(int64_t)(myk * BB + c) * L.plane + s] = acc;
                sUn[grp][myk][c] = acc;
                AHlast[c] = acc;
            }
            myk_last = myk;
        }
        if (ncols > 0) {
            // ---- the spike columns (and the rhs) through this node, as tfk_bt_col_body does:
            //      ev = v - behind * en_prev,  en = Dinv * ev
            if (on) {
#pragma unroll
                for (int col = 0; col <= BB; ++col) {
                    if (col < ncols) {
                        double ev = col < BB ? (j == 0 ? cur.bh[col] : 0.0) : cur.rhs;
                        if (j > 0) {
#pragma unroll
                            for (int k = 0; k < BB; ++k) ev = tf_fma(-cur.bh[k], sE[grp][k][col], ev);
                        }
                        sV[grp][g][col] = ev;
                    }
                }
            }
            __syncthreads();
            if (on) {
                const int64_t s = tf_idx(L, pg, dir > 0 ? j : mI - 1 - j);
#pragma unroll
                for (int col = 0; col <= BB; ++col) {
                    if (col < ncols) {
                        double en = 0.0;
#pragma unroll
                        for (int k = 0; k < BB; ++k) en = tf_fma(INV[k], sV[grp][k][col], en);
                        sE[grp][myk][col] = en;
                        ENlast[col] = en;
                        if (dir > 0) {
                            if (col < BB) a.Et[(int64_t)(myk * BB + col) * L.plane + s] = en;
                            else a.yt[(int64_t)myk * L.plane + s] = en;
                        }
                    }
                }
            }
        }
        __syncthreads();
        cur = nxt;
    }
    if (lane_on) {                                   // response of the last pivot to the separator ahead
        // AHlast is row `myk_last` of the last Un
        double* tips = dir > 0 ? a.tips_dn : a.tips_up;
#pragma unroll
        for (int c = 0; c < BB; ++c)
            tips[(int64_t)(dir > 0 ? Tip::W(0, 0, myk_last, c) : Tip::V(0, 0, myk_last, c)) * L.Ptot + pg] = AHlast[c];
#pragma unroll
        for (int col = 0; col <= BB; ++col)
            if (col < ncols) {
                const int slot = col >= BB ? Tip::y(0, myk_last)
                    : (dir > 0 ? Tip::V(0, 0, myk_last, col) : Tip::W(0, 0, myk_last, col));
                tips[(int64_t)slot * L.Ptot + pg] = ENlast[col];
            }
    }
    if (!ok && lane_on) *a.status = 1;
}

// ---- one right-hand side / spike column through the stored factors ------------
template <int BB>
__device__ __forceinline__ void tfk_bt_col_coop(const TfLevelArgs& a, int dir, int col) {
    typedef TfTips<BB, 1> Tip;
    constexpr int G = TfCoop<BB>::G, NGRP = TfCoop<BB>::NGRP;
    const TfLayout& L = a.L;
    const int grp = threadIdx.x / G, g = threadIdx.x % G;
    const int pg = blockIdx.x * NGRP + grp;
    const bool lane_on = pg < L.Ptot && g < BB;
    const int e = lane_on ? pg / L.P : 0, p = lane_on ? pg - e * L.P : 0;
    const int len = tf_len(L, p), mI = len - 1, start = tf_start(L, p);
    const int di = dir > 0 ? 0 : 1;
    const bool is_rhs = col >= BB;
    const int behind = dir > 0 ? 0 : 2;

    __shared__ double sE[NGRP][BB];                  // en of the previous node
    __shared__ double sV[NGRP][BB];                  // ev of this node
    struct Node { double Di[BB]; double Lb[BB]; double v; };
    auto load = [&](int j, Node& n) {
        const int i = dir > 0 ? j : mI - 1 - j;
        const int64_t s = tf_idx(L, pg, i);
#pragma unroll
        for (int k = 0; k < BB; ++k) {
            n.Di[k] = a.Dinv[(int64_t)((di * BB + g) * BB + k) * L.plane + s];
            n.Lb[k] = j > 0 ? a.Ablk[(int64_t)((behind * BB + g) * BB + k) * L.plane + s] : 0.0;
        }
        if (is_rhs) n.v = a.rhs[(int64_t)g * L.plane + s];
        else if (j == 0) {
            const int gn = start + i;
            const bool has_behind = L.periodic || (dir > 0 ? gn > 0 : gn < L.N - 1);
            n.v = has_behind ? a.Ablk[(int64_t)((behind * BB + g) * BB + col) * L.plane + s] : 0.0;
        } else n.v = 0.0;
    };
    Node cur, nxt;
    if (lane_on && mI > 0) load(0, cur);
    double en = 0.0, en_last = 0.0;
    const int rounds = L.M - 1;
    for (int j = 0; j < rounds; ++j) {
        const bool on = lane_on && j < mI;
        if (lane_on && j + 1 < mI) load(j + 1, nxt);
        double ev = 0.0;
        if (on) {
            ev = cur.v;
            if (j > 0) {
#pragma unroll
                for (int k = 0; k < BB; ++k) ev = tf_fma(-cur.Lb[k], sE[grp][k], ev);
            }
            sV[grp][g] = ev;
        }
        __syncthreads();
        if (on) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < BB; ++k) acc = tf_fma(cur.Di[k], sV[grp][k], acc);
            en = acc;
            en_last = en;
            sE[grp][g] = en;
            if (dir > 0) {
                const int64_t s = tf_idx(L, pg, j);
                if (is_rhs) a.yt[(int64_t)g * L.plane + s] = en;
                else a.Et[(int64_t)(g * BB + col) * L.plane + s] = en;
            }
        }
        __syncthreads();
        cur = nxt;
    }
    if (lane_on) {
        double* tips = dir > 0 ? a.tips_dn : a.tips_up;
        const int slot = is_rhs ? Tip::y(0, g) : (dir > 0 ? Tip::V(0, 0, g, col) : Tip::W(0, 0, g, col));
        tips[(int64_t)slot * L.Ptot + pg] = en_last;
    }
}

// ---- back-substitution of a chunk (separators known) ---------------------------
template <int BB>
__device__ __forceinline__ void tfk_bt_backsub_coop(const TfLevelArgs& a) {
    constexpr int G = TfCoop<BB>::G, NGRP = TfCoop<BB>::NGRP;
    const TfLayout& L = a.L;
    const int grp = threadIdx.x / G, g = threadIdx.x % G;
    const int pg = blockIdx.x * NGRP + grp;
    const bool lane_on = pg < L.Ptot && g < BB;
    const int e = lane_on ? pg / L.P : 0, p = lane_on ? pg - e * L.P : 0;
    const int len = tf_len(L, p), mI = len - 1;
    const bool has_above = L.periodic || p > 0;
    const int pa = p > 0 ? p - 1 : L.P - 1;

    __shared__ double sX[NGRP][BB];                  // solution of the node ahead
    __shared__ double sA[NGRP][BB];                  // separator above
    if (lane_on) {
        int p2, i2;
        tf_locate(a.Lnext, p, p2, i2);
        const double xs = a.xnext[tf_next_x(a, e, p, tf_idx(a.Lnext, e * a.Lnext.P + p2, i2), g, BB)];
        tf_locate(a.Lnext, pa, p2, i2);
        sX[grp][g] = xs;
        sA[grp][g] = has_above
            ? a.xnext[tf_next_x(a, e, pa, tf_idx(a.Lnext, e * a.Lnext.P + p2, i2), g, BB)] : 0.0;
        a.x[(int64_t)g * L.plane + tf_idx(L, pg, mI)] = xs;
    }
    struct Node { double U[BB]; double E[BB]; double y; };
    auto load = [&](int j, Node& n) {
        const int64_t s = tf_idx(L, pg, j);
        n.y = a.yt[(int64_t)g * L.plane + s];
#pragma unroll
        for (int k = 0; k < BB; ++k) {
            n.U[k] = a.Ut[(int64_t)(g * BB + k) * L.plane + s];
            n.E[k] = a.Et[(int64_t)(g * BB + k) * L.plane + s];
        }
    };
    Node cur, nxt;
    if (lane_on && mI > 0) load(mI - 1, cur);
    __syncthreads();
    const int rounds = L.M - 1;
    for (int jj = 0; jj < rounds; ++jj) {
        const int j = mI - 1 - jj;
        const bool on = lane_on && j >= 0;
        if (lane_on && j - 1 >= 0) load(j - 1, nxt);
        double x = 0.0;
        if (on) {
            x = cur.y;
#pragma unroll
            for (int k = 0; k < BB; ++k) x = tf_fma(-cur.U[k], sX[grp][k], x);
#pragma unroll
            for (int k = 0; k < BB; ++k) x = tf_fma(-cur.E[k], sA[grp][k], x);
        }
        __syncthreads();
        if (on) {
            sX[grp][g] = x;
            a.x[(int64_t)g * L.plane + tf_idx(L, pg, j)] = x;
        }
        __syncthreads();
        cur = nxt;
    }
}

// ---- interface equations of a reduced level: one lane per (separator, block row) ----
// Row g of  [sub | dia | sup | rhs]  of the next level (tfk_asm_body for MP = 1), same
// accumulation order; no exchange between the lanes of a group is needed.
template <int BB, bool MATRIX>
__device__ __forceinline__ void tfk_bt_asm_coop(const TfLevelArgs& a) {
    typedef TfTips<BB, 1> Tip;
    constexpr int G = TfCoop<BB>::G;
    const TfLayout& L = a.L;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int pg = gid / G, g = gid % G;
    if (pg >= L.Ptot || g >= BB) return;
    const int e = pg / L.P, p = pg - e * L.P;
    const int len = tf_len(L, p), mI = len - 1, start = tf_start(L, p);
    const bool has_next = L.periodic || p < L.P - 1;
    const int pn = e * L.P + (p < L.P - 1 ? p + 1 : 0);
    const int gn = start + mI;                               // the separator node
    const bool cut_sub = !L.periodic && gn == 0, cut_sup = !L.periodic && gn == L.N - 1;
    const int64_t s = tf_idx(L, pg, mI);
    int p2, i2;
    tf_locate(a.Lnext, p, p2, i2);
    const int64_t s2 = tf_idx(a.Lnext, e * a.Lnext.P + p2, i2);
    auto tdn = [&](int slot) { return a.tips_dn[(int64_t)slot * L.Ptot + pg]; };
    auto tup = [&](int slot) { return a.tips_up[(int64_t)slot * L.Ptot + pn]; };

    double H[BB], D[BB], K[BB];
#pragma unroll
    for (int c = 0; c < BB; ++c) {
        H[c] = cut_sub ? 0.0 : a.Ablk[(int64_t)((0 * BB + g) * BB + c) * L.plane + s];
        D[c] = a.Ablk[(int64_t)((1 * BB + g) * BB + c) * L.plane + s];
        K[c] = cut_sup ? 0.0 : a.Ablk[(int64_t)((2 * BB + g) * BB + c) * L.plane + s];
    }
    // right-hand side:  g - H * yb - K * yt(next)
    double rg = a.rhs ? a.rhs[(int64_t)g * L.plane + s] : 0.0;
#pragma unroll
    for (int k = 0; k < BB; ++k) rg = tf_fma(-H[k], tdn(Tip::y(0, k)), rg);
    if (has_next) {
#pragma unroll
        for (int k = 0; k < BB; ++k) rg = tf_fma(-K[k], tup(Tip::y(0, k)), rg);
    }
    a.rhsnext[tf_next_rhs(a, e, p, s2, g, BB)] = rg;
    if (MATRIX) {
#pragma unroll
        for (int c = 0; c < BB; ++c) {
            double sub = 0.0, dia = 0.0, sup = 0.0;
#pragma unroll
            for (int k = 0; k < BB; ++k) {
                sub = tf_fma(-H[k], tdn(Tip::V(0, 0, k, c)), sub);
                dia = tf_fma(-H[k], tdn(Tip::W(0, 0, k, c)), dia);
            }
            dia += D[c];
            if (has_next) {
#pragma unroll
                for (int k = 0; k < BB; ++k) {
                    dia = tf_fma(-K[k], tup(Tip::V(0, 0, k, c)), dia);
                    sup = tf_fma(-K[k], tup(Tip::W(0, 0, k, c)), sup);
                }
            }
            a.Anext[tf_next_A(a, e, p, s2, 0, g, c, BB)] = sub;
            a.Anext[tf_next_A(a, e, p, s2, 1, g, c, BB)] = dia;
            a.Anext[tf_next_A(a, e, p, s2, 2, g, c, BB)] = sup;
        }
    }
}

// ===========================================================================
// Cyclic reduction inside the chunks of the reduced levels (3 <= b <= 8)
// ===========================================================================
// The walks above are latency-bound: a chunk of m nodes costs m-1 dependent
// block inversions.  Here ONE wavefront owns a chunk of up to 16 nodes and
// eliminates its interior in nested-dissection order: in round r every second
// remaining interior node goes at once (8 lanes per node, 8 nodes per round),
// so 15 interior nodes cost 4 dependent inversions instead of 15, and the
// Schur complement on the two separators that bound the chunk comes out of the
// last round -- no separate assemble kernel.  Levels shrink 16x per launch.
//
// Chain positions of chunk p: 0 = separator above (last node of chunk p-1),
// 1..mI = interior nodes, pe = mI+1 = own separator; position pos >= 1 is node
// start+pos-1.  Row pos keeps (L, D, U) = coupling to its current left
// neighbour, itself, its current right neighbour.  Round with stride s
// eliminates k = s*(odd) <= mI; its neighbours are k-s and (k+s <= mI ? k+s : pe):
//   E_k = D_k^-1 L_k,  F_k = D_k^-1 U_k,  z_k = D_k^-1 y_k
//   left  a:  D_a -= U_a E_k,  U_a' = -U_a F_k,  y_a -= U_a z_k
//   right b:  D_b -= L_b F_k,  L_b' = -L_b E_k,  y_b -= L_b z_k
// Stored per eliminated node (a.crf): D^-1, E, F, the U_a and L_b used above;
// a.zt holds z.  Buffers of these levels are records per node in natural order
// (TfLevelArgs).  What is left of rows 0 and pe is this chunk's share of the next
// level's rows: node p gets (L, D) from its own chunk and (U, second part of D)
// from chunk p+1.
template <int BB> struct TfCr {
    static constexpr int MAXLEN = TF_CR_MAXLEN;  // nodes per chunk
    static constexpr int NPOS = MAXLEN + 1;
    static constexpr int G = 8, NGRP = 8;
};

// Gauss-Jordan inverse of a b x b block shared by the 8 lanes of a group: lane g
// enters with row g of the block in S and leaves with row `myk` (returned) of the
// inverse in INV.  Rows are never moved: the lane with the largest |S[.][k]| among
// the lanes not used yet serves pivot k and publishes its row through `xch`
// ([2][G][2b+1] doubles of this group, double buffered: one LDS round per pivot).
template <int BB, int G>
__device__ __forceinline__ int tf_gj_coop(double (&S)[BB], double (&INV)[BB], bool on, int g,
                                          double* xch, bool& ok) {
    constexpr int RS = 2 * BB + 1;
    int myk = -1;
#pragma unroll
    for (int c = 0; c < BB; ++c) INV[c] = c == g ? 1.0 : 0.0;
    auto publish = [&](int kk) {
        double* dst = xch + ((kk & 1) * G + g) * RS;
#pragma unroll
        for (int c = 0; c < BB; ++c) { dst[c] = S[c]; dst[BB + c] = INV[c]; }
        dst[2 * BB] = (on && myk < 0) ? tf_abs(S[kk]) : -1.0;
    };
    publish(0);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BB; ++kk) {
        const double* buf = xch + (kk & 1) * G * RS;
        int piv = 0;
        double best = -2.0;
#pragma unroll
        for (int r = 0; r < BB; ++r) {
            const double v = buf[r * RS + 2 * BB];
            if (v > best) { best = v; piv = r; }
        }
        const double* prow = buf + piv * RS;
        const double pv = prow[kk];
        if (on) ok = ok && (pv != 0.0) && tf_finite(pv);
        const double rp = 1.0 / pv;
        const bool mine = g == piv;
        if (mine) myk = kk;
        // every other row: S -= (S[k]/pv) * pivot row; the pivot row itself is scaled
        const double f2 = mine ? 0.0 : -S[kk] * rp;
#pragma unroll
        for (int c = 0; c < BB; ++c) {
            S[c] = tf_fma(f2, prow[c], S[c]);
            INV[c] = tf_fma(f2, prow[BB + c], INV[c]);
        }
        if (mine) {
#pragma unroll
            for (int c = 0; c < BB; ++c) { S[c] *= rp; INV[c] *= rp; }
        }
        if (kk + 1 < BB) publish(kk + 1);
        __syncthreads();
    }
    return myk;
}

// The factorisation of these levels is tfk_cr_factor_v3 (tf_cr2_hip.h: a wavefront per node);
// the round-1 version that stood here gave a node 8 lanes and took 155 us per step against 94
// (profiles/README.md).  The solve kernels below read what the factorisation stores.

// One block row (BB doubles, contiguous) of a stored record.  Records start at multiples of
// 5*BB*BB doubles and a row at g*BB, so for even BB a row is 16-byte aligned and goes as
// BB/2 x global_load_dwordx4 instead of BB x dwordx2 (half the requests of the solve kernels).
template <int BB>
__device__ __forceinline__ void tf_load_row(const double* p, double (&dst)[BB]) {
    if constexpr (BB % 2 == 0) {
        const double2* q = reinterpret_cast<const double2*>(__builtin_assume_aligned(p, 16));
#pragma unroll
        for (int m = 0; m < BB / 2; ++m) {
            const double2 v = q[m];
            dst[2 * m] = v.x;
            dst[2 * m + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int m = 0; m < BB; ++m) dst[m] = p[m];
    }
}

// Rows of the stored reduction that a lane needs in round r are known up front
// (one task per group and phase), so the solve kernels request all of them
// before the first round: one memory latency per launch instead of one per round.
//
// One WAVEFRONT per chunk: the lanes exchange right-hand sides through the chunk's LDS block
// (TfCrSolveLds).  LDS executes the instructions of a wavefront in order, so the phases of a
// chunk need no s_barrier, only that the compiler keeps the order (tf_wave_sync) -- which lets
// several chunks, each with its own wavefront, share a workgroup (tfk_cr_tail below).
__device__ __forceinline__ void tf_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int BB> struct TfCrSolveLds {
    static constexpr int NPOS = TF_CR_MAXLEN + 1;
    double sYr[NPOS * 2 * BB], sY[NPOS][BB], sZ[NPOS][BB];
};

// Where a chunk's solve reads and writes (global memory in the one-chunk-per-workgroup launches;
// tfk_cr_tail hands the right-hand sides and solutions between its two levels through LDS)
template <int BB> struct TfCrIo {
    const double* ys;        // right-hand side records of the chunk's nodes [len][2][b] (two parts, summed)
    const double* crf0;      // stored reduction, record of the chunk's first node ([node][5][b][b])
    double* next_own;        // forward: part 0 of next-level node p   [b]   (not with fold_top)
    double* next_prev;       //          part 1 of next-level node p-1 [b]
    // (flags, not null tests: a test of an LDS pointer against NULL trips hipcc 7.2 for b = 7,
    // "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base")
    double* xmirror;         // fold_top: a second copy of the solution [node][b], if has_mirror
    bool has_mirror;
    const double* xn_own;    // backward: solution of next-level node p [b], of node p-1 if has_xprev
    const double* xn_prev;
    bool has_xprev;
    __device__ __forceinline__ TfCrIo(const TfLevelArgs& a, const TfCrChunk<BB>& ch) {
        ys = a.rhs + (ch.nbase + ch.start) * 2 * BB;
        crf0 = a.crf + (ch.nbase + ch.start) * 5 * BB * BB;
        next_own = a.rhsnext ? a.rhsnext + ((int64_t)ch.e * a.Lnext.N + ch.p) * 2 * BB : nullptr;
        next_prev = a.rhsnext ? a.rhsnext + ((int64_t)ch.e * a.Lnext.N + ch.pprev) * 2 * BB + BB : nullptr;
        xmirror = nullptr; has_mirror = false;
        xn_own = a.xnext ? a.xnext + ((int64_t)ch.e * a.Lnext.N + ch.p) * BB : nullptr;
        xn_prev = a.xnext ? a.xnext + ((int64_t)ch.e * a.Lnext.N + ch.pprev) * BB : nullptr;
        has_xprev = ch.has_prev;
    }
};

// The rows of the stored reduction a lane's forward tasks need (independent of the right-hand side)
template <int BB> struct TfCrFwdRows { double Di[4][BB], Lb[4][BB], Ua[4][BB]; };
template <int BB>
__device__ __forceinline__ void tfk_cr_fwd_load(const TfCrChunk<BB>& ch, int tid, const TfCrIo<BB>& io, TfCrFwdRows<BB>& rows) {
    constexpr int G = TfCr<BB>::G, B2 = BB * BB, MAXR = 4;
    const int grp = tid / G, g = tid % G;
    const bool row_on = g < BB;
    const int mI = ch.mI;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int s = 1 << r;
        const int nA = ((mI >> r) + 1) >> 1, nB = mI >> (r + 1), nq = mI >> r;
        const bool onA = s <= mI && grp < nA && row_on;
        const bool onB = s <= mI && grp <= nB && row_on;
        const bool ends = grp == nB;
        const int aa = 2 * s * (grp + 1);
        const bool vL = onB && (ends ? (nq & 1) != 0 : true);
        const bool vR = onB && (ends ? true : aa + s <= mI);
        const int k = s * (2 * grp + 1), kL = ends ? nq * s : aa - s, kR = ends ? s : aa + s;
        const double* rk = io.crf0 + ((onA ? k : 1) - 1) * 5 * B2 + g * BB;
        const double* rl = io.crf0 + ((vL ? kL : 1) - 1) * 5 * B2 + 4 * B2 + g * BB;
        const double* rr = io.crf0 + ((vR ? kR : 1) - 1) * 5 * B2 + 3 * B2 + g * BB;
        // (a lane without the task loads node 1's row and never uses it: every use sits behind the task's
        // own condition -- masking the 12 b values to zero was 24 b instructions per chunk)
        tf_load_row<BB>(rk, rows.Di[r]);
        tf_load_row<BB>(rl, rows.Lb[r]);
        tf_load_row<BB>(rr, rows.Ua[r]);
    }
}
// ... and the chunk's right-hand side records, TF_CR_NYS(b) values per lane
#define TF_CR_NYS(b) ((TF_CR_MAXLEN * 2 * (b) + 63) / 64)
template <int BB>
__device__ __forceinline__ void tfk_cr_fwd_load_ys(const TfCrChunk<BB>& ch, int tid, const TfCrIo<BB>& io, double (&ysv)[TF_CR_NYS(BB)]) {
#pragma unroll
    for (int q = 0; q < TF_CR_NYS(BB); ++q) ysv[q] = tid + 64 * q < ch.len * 2 * BB ? io.ys[tid + 64 * q] : 0.0;
}

// forward elimination of the right-hand side through chunk `ch`; tid = lane of its wavefront.
// zkeep[r] (optional): z of this lane's task of round r stays in a register for the
// back-substitution of the same launch.  pre / ysv (optional): rows and right-hand side records
// requested by the caller earlier (tfk_cr_fwd_load, tfk_cr_fwd_load_ys).
template <int BB, bool KEEPZ = false, bool PRE = false>
__device__ __forceinline__ void tfk_cr_fwd_chunk(const TfLevelArgs& a, const TfCrChunk<BB>& ch, int tid,
                                                 TfCrSolveLds<BB>& sh, const TfCrIo<BB>& io,
                                                 double* zkeep = nullptr, const TfCrFwdRows<BB>* pre = nullptr,
                                                 const double* ysv = nullptr) {
    typedef TfCr<BB> C;
    constexpr int G = C::G, B2 = BB * BB, MAXR = 4;
    static_assert(C::MAXLEN <= 16, "round count");
    const TfLayout& L = a.L;
    const int grp = tid / G, g = tid % G;
    const bool row_on = g < BB;
    const int mI = ch.mI, pe = ch.pe, len = ch.len;
    double (&sY)[TfCrSolveLds<BB>::NPOS][BB] = sh.sY;
    double (&sZ)[TfCrSolveLds<BB>::NPOS][BB] = sh.sZ;
    double* sYr = sh.sYr;

    double* const zt0 = a.zt + (ch.nbase + ch.start - 1) * BB + (row_on ? g : 0);     // z of position k: zt0[k * BB]
    tf_wave_sync();                                  // (the block may still be read by the previous chunk's phases)
    TfCrFwdRows<BB> own;
    if constexpr (PRE) {
#pragma unroll
        for (int q = 0; q < TF_CR_NYS(BB); ++q)
            if (tid + 64 * q < len * 2 * BB) sYr[2 * BB + tid + 64 * q] = ysv[q];
    } else {
        for (int i = tid; i < len * 2 * BB; i += 64) sYr[2 * BB + i] = io.ys[i];
        tfk_cr_fwd_load<BB>(ch, tid, io, own);
    }
    const TfCrFwdRows<BB>& rw = PRE ? *pre : own;
    const double (&Di)[4][BB] = rw.Di;
    const double (&Lb)[4][BB] = rw.Lb;
    const double (&Ua)[4][BB] = rw.Ua;
    tf_wave_sync();
    if (PRE && (ch.p & 3) == 0) TF_STAMP(a, 42);
    for (int i = tid; i < (len + 1) * BB; i += 64) {
        const int pos = i / BB, r = i - pos * BB;
        sY[pos][r] = pos > 0 ? sYr[pos * 2 * BB + r] + sYr[pos * 2 * BB + BB + r] : 0.0;
    }
    tf_wave_sync();
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int s = 1 << r;
        if (s <= mI) {
            const int nA = ((mI >> r) + 1) >> 1, nB = mI >> (r + 1), nq = mI >> r;
            if (grp < nA && row_on) {
                const int k = s * (2 * grp + 1);
                double z = 0.0;
#pragma unroll
                for (int m = 0; m < BB; ++m) z = tf_fma(Di[r][m], sY[k][m], z);
                sZ[k][g] = z;
                zt0[k * BB] = z;
                if (KEEPZ) zkeep[r] = z;
            }
            tf_wave_sync();
            if (grp <= nB && row_on) {
                const bool ends = grp == nB;
                const int aa = 2 * s * (grp + 1);
                const int aL = ends ? pe : aa, aU = ends ? 0 : aa;
                const bool vL = ends ? (nq & 1) != 0 : true;
                const bool vR = ends ? true : aa + s <= mI;
                const int kL = vL ? (ends ? nq * s : aa - s) : 1, kR = vR ? (ends ? s : aa + s) : 1;
                double yl = 0.0, yu = 0.0;
#pragma unroll
                for (int m = 0; m < BB; ++m) {
                    yl = tf_fma(-Lb[r][m], sZ[kL][m], yl);      // (used behind vL / vR only)
                    yu = tf_fma(-Ua[r][m], sZ[kR][m], yu);
                }
                if (vL) sY[aL][g] += yl;
                if (vR) sY[aU][g] += yu;
            }
            tf_wave_sync();
        }
    }
    if (PRE && (ch.p & 3) == 0) TF_STAMP(a, 43);
    if (a.fold_top) {
        // last level: apply the inverse of the remaining block (tfk_cr_factor_v3) and run
        // the back-substitution rounds of this level right away (tfk_cr_bwd_run)
        double x = 0.0;
        if (row_on && grp == 0) {
            const int nsys = L.Ptot;
#pragma unroll
            for (int c = 0; c < BB; ++c)
                x = tf_fma(a.topAinv[(int64_t)(g * BB + c) * nsys + ch.e], sY[pe][c] + sY[0][c], x);
        }
        tf_wave_sync();
        if (row_on && grp == 0) {
            a.topx[(int64_t)ch.e * BB + g] = x;
            a.x[(ch.nbase + ch.node(pe)) * BB + g] = x;
            if (io.has_mirror) io.xmirror[(pe - 1) * BB + g] = x;
            sY[pe][g] = x;
            sY[0][g] = ch.has_prev ? x : 0.0;
        }
        tf_wave_sync();
#pragma unroll
        for (int r = MAXR - 1; r >= 0; --r) {
            const int s = 1 << r;
            if (s <= mI) {
                const int nA = ((mI >> r) + 1) >> 1;
                if (grp < nA && row_on) {
                    const int k = s * (2 * grp + 1);
                    const int kl = k - s, kr = k + s <= mI ? k + s : pe;
                    const double* rec = io.crf0 + (k - 1) * 5 * B2 + g * BB;
                    double xk = sZ[k][g];
#pragma unroll
                    for (int m = 0; m < BB; ++m) {
                        xk = tf_fma(-rec[1 * B2 + m], sY[kl][m], xk);
                        xk = tf_fma(-rec[2 * B2 + m], sY[kr][m], xk);
                    }
                    sY[k][g] = xk;
                    a.x[(ch.nbase + ch.node(k)) * BB + g] = xk;
                    if (io.has_mirror) io.xmirror[(k - 1) * BB + g] = xk;
                }
                tf_wave_sync();
            }
        }
    } else if (row_on && grp < 2) {
        if (grp == 0) io.next_own[g] = sY[pe][g]; else io.next_prev[g] = sY[0][g];
    }
}

// back-substitution, in two parts: the rows of the stored reduction a lane needs (independent of
// the right-hand side: requested as early as the caller can), then the rounds with the
// separators that bound the chunk known
template <int BB> struct TfCrBwdRows { double Er[4][BB], Fr[4][BB], zk[4]; };

template <int BB>
__device__ __forceinline__ void tfk_cr_bwd_load(const TfLevelArgs& a, const TfCrChunk<BB>& ch, int tid,
                                                const TfCrIo<BB>& io, TfCrBwdRows<BB>& rows, bool load_z) {
    constexpr int G = TfCr<BB>::G, B2 = BB * BB, MAXR = 4;
    const int grp = tid / G, g = tid % G;
    const bool row_on = g < BB;
    const int mI = ch.mI;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int s = 1 << r;
        const int nA = ((mI >> r) + 1) >> 1;
        const bool on = s <= mI && grp < nA && row_on;
        const int k = on ? s * (2 * grp + 1) : 1;
        const double* rec = io.crf0 + (k - 1) * 5 * B2 + g * BB;
        tf_load_row<BB>(rec + 1 * B2, rows.Er[r]);
        tf_load_row<BB>(rec + 2 * B2, rows.Fr[r]);
        if (load_z) rows.zk[r] = a.zt[(ch.nbase + ch.node(k)) * BB + (row_on ? g : 0)];
    }
}

// ef (optional): the E and F blocks of the chunk's nodes staged in LDS, [node][2][b][b] from the
// chunk's first node -- then rows.Er / rows.Fr are not used (only rows.zk)
// PREX: xsep = the caller's earlier load of tfk_cr_bwd_sep (the separators that bound the chunk)
template <int BB>
__device__ __forceinline__ double tfk_cr_bwd_sep(int tid, const TfCrIo<BB>& io) {
    constexpr int G = TfCr<BB>::G;
    const int grp = tid / G, g = tid % G;
    if (g < BB && grp == 0) return io.xn_own[g];
    if (g < BB && grp == 1 && io.has_xprev) return io.xn_prev[g];
    return 0.0;
}
template <int BB, bool USE_EF = false, bool PREX = false>
__device__ __forceinline__ void tfk_cr_bwd_run(const TfLevelArgs& a, const TfCrChunk<BB>& ch, int tid,
                                               TfCrSolveLds<BB>& sh, const TfCrIo<BB>& io,
                                               const TfCrBwdRows<BB>& rows, const double* ef = nullptr, double xsep = 0.0) {
    constexpr int G = TfCr<BB>::G, MAXR = 4, B2 = BB * BB;
    const int grp = tid / G, g = tid % G;
    const bool row_on = g < BB;
    const int mI = ch.mI, pe = ch.pe;
    double (&sX)[TfCrSolveLds<BB>::NPOS][BB] = sh.sY;
    double* const x0 = a.x + (ch.nbase + ch.start - 1) * BB + (row_on ? g : 0);       // x of position k: x0[k * BB]
    tf_wave_sync();
    if (row_on && grp == 0) {
        const double xs = PREX ? xsep : io.xn_own[g];
        sX[pe][g] = xs;
        x0[pe * BB] = xs;
    }
    if (row_on && grp == 1) sX[0][g] = PREX ? xsep : (io.has_xprev ? io.xn_prev[g] : 0.0);
    tf_wave_sync();
#pragma unroll
    for (int r = MAXR - 1; r >= 0; --r) {
        const int s = 1 << r;
        if (s <= mI) {
            const int nA = ((mI >> r) + 1) >> 1;
            if (grp < nA && row_on) {
                const int k = s * (2 * grp + 1);
                const int kl = k - s, kr = k + s <= mI ? k + s : pe;
                double xk = rows.zk[r];
                if (USE_EF) {
                    const double* er = ef + (k - 1) * 2 * B2 + g * BB;
#pragma unroll
                    for (int m = 0; m < BB; ++m) {
                        xk = tf_fma(-er[m], sX[kl][m], xk);
                        xk = tf_fma(-er[B2 + m], sX[kr][m], xk);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < BB; ++m) {
                        xk = tf_fma(-rows.Er[r][m], sX[kl][m], xk);
                        xk = tf_fma(-rows.Fr[r][m], sX[kr][m], xk);
                    }
                }
                sX[k][g] = xk;
                x0[k * BB] = xk;
            }
            tf_wave_sync();
        }
    }
}

// one chunk per 64-thread workgroup (the launches of the big levels)
template <int BB>
__device__ __forceinline__ void tfk_cr_fwd_coop(const TfLevelArgs& a) {
    __shared__ TfCrSolveLds<BB> sh;
    const TfCrChunk<BB> ch(a.L, (int)blockIdx.x);
    tfk_cr_fwd_chunk<BB>(a, ch, (int)threadIdx.x, sh, TfCrIo<BB>(a, ch));
}
template <int BB>
__device__ __forceinline__ void tfk_cr_bwd_coop(const TfLevelArgs& a) {
    __shared__ TfCrSolveLds<BB> sh;
    const TfCrChunk<BB> ch(a.L, (int)blockIdx.x);
    const TfCrIo<BB> io(a, ch);
    TfCrBwdRows<BB> rows;
    tfk_cr_bwd_load<BB>(a, ch, (int)threadIdx.x, io, rows, true);
    tfk_cr_bwd_run<BB>(a, ch, (int)threadIdx.x, sh, io, rows);
}

// The two smallest levels of a solve in ONE launch: level T has at most a few chunks per
// system (8 for N = 1e6), level T+1 is the last one (one chunk, which also applies the inverse of
// the top block and back-substitutes itself).  As three launches (forward T, forward T+1,
// backward T) they cost 16 us: every launch is a round trip to memory for its rows and
// another one for what the previous launch left (profiles/r02_solver_levels_trace.txt; a first
// fused version that kept those round trips took 18 us, profiles/r03_ab_runs.txt).  Here a
// workgroup per system gives every chunk of level T a wavefront and pays the memory latency once:
// at the start the workgroup requests level T+1's whole stored reduction and the blocks of level
// T's back-substitution into LDS, and every wavefront the rows of its forward elimination; level T's share of level T+1's right-hand side, and level T+1's
// solution, pass through LDS.
#define TF_CR_TAIL_WAVES 8         // = the most chunks level T may have per system (one wavefront each)
#define TF_CR_TAIL_MAXB 6          // (b = 7 spills and trips a hipcc 7.2 code-generation error; b = 8 exceeds the LDS)
template <int BB>
__device__ __forceinline__ void tfk_cr_tail_coop(const TfTailArgs& t) {
    constexpr int B2 = BB * BB, NW = TF_CR_TAIL_WAVES, MAXLEN = TF_CR_MAXLEN;
    __shared__ TfCrSolveLds<BB> sh[NW];
    __shared__ double sEF[NW][MAXLEN * 2 * B2];      // level T: E and F blocks of every wavefront's chunk
    __shared__ double sCrfTop[NW * 5 * B2];          // level T+1: stored reduction of its one chunk (<= NW nodes)
    __shared__ double sRhsTop[NW * 2 * BB];          // ... its right-hand side records (two parts per node)
    __shared__ double sXTop[NW * BB];                // ... its solution
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63, e = (int)blockIdx.x;
    const TfLevelArgs& la = t.lv[0];
    const TfLevelArgs& lb = t.lv[1];
    const int P = la.L.P;                            // <= NW (host)
    const TfCrChunk<BB> chb(lb.L, e);
    const TfCrChunk<BB> ch(la.L, e * P + (w < P ? w : 0));
    const bool mine = w < P;
    TF_STAMP(la, 50);
    // level T+1's stored reduction: requested now, put into LDS once every other request of this
    // wavefront is on its way (the store waits for the data)
    constexpr int NTOP = (NW * 5 * B2 + 64 * NW - 1) / (64 * NW);
    double topv[NTOP];
    {
        const double* src = lb.crf + chb.nbase * 5 * B2;              // (one chunk: it starts at node 0 of the system)
#pragma unroll
        for (int q = 0; q < NTOP; ++q) {
            // (no branch around a request: beyond the level's records the last one is read again and not used)
            const int i = (int)threadIdx.x + 64 * NW * q, lim = P * 5 * B2 - 1;
            topv[q] = src[i < lim ? i : lim];
        }
    }
    TfCrIo<BB> io(la, ch);
    // E, F of my chunk's interior nodes (blocks 1 and 2 of a record are contiguous): needed last,
    // requested first, and parked in registers until the forward elimination is through (an LDS
    // store here would wait for them before the rows of the elimination are even requested)
    constexpr int NEF = (MAXLEN * 2 * B2 + 63) / 64;
    double ef[NEF];
#pragma unroll
    for (int q = 0; q < NEF; ++q) {
        const int i = lane + 64 * q;
        const int nd = i / (2 * B2), o = i - nd * 2 * B2;
        const int ndc = nd < ch.mI ? nd : (ch.mI > 0 ? ch.mI - 1 : 0);            // (clamped, not branched around)
        ef[q] = io.crf0[ndc * 5 * B2 + B2 + o];
    }
    TfCrBwdRows<BB> rows;
#pragma unroll
    for (int r = 0; r < 4; ++r) rows.zk[r] = 0.0;
    TF_STAMP(la, 51);
    if (mine) {
        io.next_own = sRhsTop + ch.p * 2 * BB;
        io.next_prev = sRhsTop + ch.pprev * 2 * BB + BB;
        tfk_cr_fwd_chunk<BB, true>(la, ch, lane, sh[w], io, rows.zk);
    }
    TF_STAMP(la, 52);
#pragma unroll
    for (int q = 0; q < NEF; ++q)
        if (lane + 64 * q < MAXLEN * 2 * B2) sEF[w][lane + 64 * q] = ef[q];
#pragma unroll
    for (int q = 0; q < NTOP; ++q) {
        const int i = (int)threadIdx.x + 64 * NW * q;
        if (i < NW * 5 * B2) sCrfTop[i] = topv[q];
    }
    __syncthreads();                                 // level T+1's right-hand side and stored reduction are in LDS
    TF_STAMP(la, 53);
    if (w == 0) {
        TfCrIo<BB> iob(lb, chb);
        iob.ys = sRhsTop;
        iob.crf0 = sCrfTop;
        iob.xmirror = sXTop; iob.has_mirror = true;
        tfk_cr_fwd_chunk<BB>(lb, chb, lane, sh[0], iob);
    }
    TF_STAMP(la, 54);
    __syncthreads();                                 // ... and its solution
    TF_STAMP(la, 55);
    if (mine) {
        io.xn_own = sXTop + ch.p * BB;
        io.xn_prev = sXTop + ch.pprev * BB;
        tfk_cr_bwd_run<BB, true>(la, ch, lane, sh[w], io, rows, sEF[w]);
    }
    TF_STAMP(la, 56);
}

// ---- top block: one group of 8 lanes inverts the b x b system of one ensemble member
template <int BB>
__device__ __forceinline__ void tfk_top_factor_coop(const TfTopArgs& a) {
    constexpr int G = 8, NGRP = 8;
    const int grp = threadIdx.x / G, g = threadIdx.x % G;
    const int e = blockIdx.x * NGRP + grp;
    const bool on = e < a.nsys && g < BB;
    const int es = e < a.nsys ? e : 0, gq = g < BB ? g : 0;
    __shared__ double sX[NGRP * 2 * G * (2 * BB + 1)];
    double S[BB], INV[BB];
#pragma unroll
    for (int c = 0; c < BB; ++c) {
        if (a.aos) {
            const double* rec = a.A + (int64_t)es * 4 * BB * BB + gq * BB + c;
            S[c] = rec[0] + rec[BB * BB] + rec[2 * BB * BB] + rec[3 * BB * BB];
        } else {
            S[c] = a.A[(int64_t)((0 * BB + gq) * BB + c) * a.nsys + es]
                 + a.A[(int64_t)((1 * BB + gq) * BB + c) * a.nsys + es]
                 + a.A[(int64_t)((2 * BB + gq) * BB + c) * a.nsys + es];
        }
    }
    bool ok = true;
    const int myk = tf_gj_coop<BB, G>(S, INV, on, g, sX + grp * 2 * G * (2 * BB + 1), ok);
    if (on) {
#pragma unroll
        for (int c = 0; c < BB; ++c) a.Ainv[(int64_t)(myk * BB + c) * a.nsys + e] = INV[c];
        if (!ok) *a.status = 1;
    }
}

