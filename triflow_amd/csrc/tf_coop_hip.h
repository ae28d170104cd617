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
        const double xs = a.xnext[(int64_t)g * a.Lnext.plane + tf_idx(a.Lnext, e * a.Lnext.P + p2, i2)];
        tf_locate(a.Lnext, pa, p2, i2);
        sX[grp][g] = xs;
        sA[grp][g] = has_above
            ? a.xnext[(int64_t)g * a.Lnext.plane + tf_idx(a.Lnext, e * a.Lnext.P + p2, i2)] : 0.0;
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
    a.rhsnext[(int64_t)g * a.Lnext.plane + s2] = rg;
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
            a.Anext[(int64_t)((0 * BB + g) * BB + c) * a.Lnext.plane + s2] = sub;
            a.Anext[(int64_t)((1 * BB + g) * BB + c) * a.Lnext.plane + s2] = dia;
            a.Anext[(int64_t)((2 * BB + g) * BB + c) * a.Lnext.plane + s2] = sup;
        }
    }
}
