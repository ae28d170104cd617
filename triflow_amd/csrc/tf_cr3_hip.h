// Cyclic-reduction factorisation of a reduced-level chunk, round 4 form (3 <= b <= 7): the
// elimination of a node AND the update of its two neighbours by the same wavefront, one barrier
// per round, columns of the augmented block in lanes.
//
// What the round-2/3 kernel (tf_cr2_hip.h, tfk_cr_factor_v3) spent per round on a small level
// (profiles/r03_solver_levels_trace.txt, cycles of one wavefront): phase A 480 until its rows are
// there + 2 190 block inversion + 1 000-1 260 stores (four divergent branches per entry, 64-bit
// addresses per lane) + barrier; phase B 680 prologue + 650-750 products + 830 stores + barrier:
// ~6 000 cycles for 6 pivots and 24 FMAs per lane, issued by a lone wavefront at one instruction per
// 8-10 cycles (SQ counters: SALU = 0.66 x VALU, any instruction active 0.30).  Here
//   * lane c of each half wavefront holds COLUMN c of the augmented block [L | D | U | y | I] of the
//     node that goes, its b rows in b registers.  A pivot step is: the pivot and the b-1 multipliers
//     by v_readlane (wave-uniform scalars), the reciprocal on them, one multiply and b-1 FMAs with
//     scalar operands -- no ds_bpermute, no DPP (nothing on the critical path leaves the register
//     file).  The remembered pivot order of tf_cr2_hip.h stays:
//     the rows are *loaded* in that order; growth above TF_GJ_GROWTH or anything non-finite sends the
//     node to the search (tf_gj_wave, the old lane layout, on the rows in natural order);
//   * the wavefront that eliminated node k has E = D^-1 L, F = D^-1 U, z = D^-1 y in registers and
//     goes straight on to the two neighbours a = k - s, b = k + s: half 0 multiplies U_a into its
//     columns (D_a -= U_a E, U_a' = -U_a F, y_a -= U_a z), half 1 L_b (L_b' = -L_b E, D_b -= L_b F,
//     y_b -= L_b z) -- 36 FMAs per lane with broadcast LDS operands.  The two contributions to a
//     diagonal block come from different wavefronts, so a row keeps two accumulators (DL, yL: what
//     the eliminations on its left added, DR, yR: on its right; D = DL + DR when the row itself
//     goes): no race, no second barrier, no round trip of E / F / z through LDS;
//   * record addresses: wave-uniform base + 32-bit lane offset (global_store with saddr), the lane
//     offsets worked out once per kernel; z leaves through LDS as one contiguous store per chunk.
// Stored quantities (a.crf, a.zt, a.Anext, a.rhsnext, a.perm) and their formats are those of
// tfk_cr_factor_v3: tfk_cr_fwd / tfk_cr_bwd / tfk_cr_tail read them unchanged.  b = 8 (33 augmented
// columns: more than a half wavefront) runs with one copy of the columns over the 64 lanes and the two
// neighbours one after the other.  HIP only.
#pragma once

template <int BB> struct TfCr3 {
    static constexpr int MAXLEN = TF_CR_MAXLEN, NPOS = MAXLEN + 1;
    // a block row of a chain position in LDS: [L | DL | U | yL | DR | yR | 0]
    static constexpr int oL = 0, oDL = BB, oU = 2 * BB, oYL = 3 * BB, oDR = 3 * BB + 1, oYR = 4 * BB + 1,
                         oZ = 4 * BB + 2;
    static constexpr int RS = (4 * BB + 3) | 1;          // odd: rows of a position start on different banks
    static constexpr int PS = BB * RS;
    static constexpr int NC = 4 * BB + 1;                // augmented columns [L | D | U | y | I]
    // b <= 7: the columns fit a half wavefront and both halves hold them (one half per neighbour);
    // b = 8 ... 15: one copy over the 64 lanes, the two neighbours one after the other
    static constexpr bool DUP = NC <= 32;
    static_assert(NC <= 64, "a wavefront holds the augmented columns");
};

// store to a wave-uniform base + 32-bit lane byte offset (saddr form: no 64-bit lane arithmetic)
__device__ __forceinline__ void tf_st_u(double* ubase, unsigned off8, double v) {
    *(double*)((char*)ubase + off8) = v;
}

// One pivot step of Gauss-Jordan in a fixed order, rows in registers, columns in lanes.
template <int BB, int K>
__device__ __forceinline__ void tf_gj3_step(double (&val)[BB], int c, bool& grow) {
    constexpr int PL = BB + K;                           // the lane (of half 0) that holds pivot column K
    const double pv = tf_readlane_f64(val[K], PL);
    const double rp = tf_rcp_newton(pv);
    if constexpr (K + 1 < BB) {
        // threshold pivoting on the remembered order: a multiplier above TF_GJ_GROWTH among the rows
        // not yet used (looked at in the pivot column's own lane, off the critical path)
        double mx = 0.0;
#pragma unroll
        for (int i = K + 1; i < BB; ++i) mx = __builtin_fmax(mx, tf_abs(val[i]));
        if (c == PL && !(mx <= TF_GJ_GROWTH * tf_abs(val[K]))) grow = true;
    }
    const double pr = val[K] * rp;
#pragma unroll
    for (int i = 0; i < BB; ++i) {
        if (i == K) continue;
        const double m = tf_readlane_f64(val[i], PL);
        val[i] = tf_fma(-m, pr, val[i]);
    }
    val[K] = pr;
    if constexpr (K + 1 < BB) tf_gj3_step<BB, K + 1>(val, c, grow);
}

template <int BB>
__device__ __forceinline__ void tfk_cr_factor_v4(const TfLevelArgs& a) {
    typedef TfCr3<BB> C;
    constexpr int NPOS = C::NPOS, B2 = BB * BB, REC = 4 * B2, NT_MIN = 256;
    constexpr int RS = C::RS, PS = C::PS, NC = C::NC;
    constexpr int oL = C::oL, oDL = C::oDL, oU = C::oU, oYL = C::oYL, oDR = C::oDR, oYR = C::oYR, oZ = C::oZ;
    constexpr int NO = 2 * BB + 1;                       // outputs of one side of the share
    constexpr int NQ = (2 * B2 + 63) / 64;               // instructions that copy the U_a, L_b blocks (2 b^2 entries)
    const int NT = blockDim.x, nw = NT >> 6;             // 8 wavefronts per chunk, or 4 (levels with many chunks)
    const TfLayout& L = a.L;
    const TfCrChunk<BB> ch(L, (int)blockIdx.x);
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    constexpr bool DUP = C::DUP;
    const int half = DUP ? lane >> 5 : 0, c = DUP ? lane & 31 : lane;
    const int mI = ch.mI, pe = ch.pe, len = ch.len;
    const bool with_rhs = a.cr_rhs != 0;

    __shared__ double sRow[NPOS * PS];
    __shared__ unsigned sPerm[NPOS + 1];                 // stored pivot order of the chunk's nodes
    auto row = [&](int pos, int r) { return sRow + pos * PS + r * RS; };
    unsigned* const perm = a.perm + ch.nbase + ch.start; // [node]; the top block's follows the nodes
    if (tid < len) sPerm[tid + 1] = perm[tid];
    unsigned* const perm_top = a.perm + (int64_t)L.nsys * L.N + ch.e;    // (fold_top: P == 1, one per system)
    if (a.fold_top && tid == 64) sPerm[0] = *perm_top;

    // ---- what a lane does, worked out once: column class, LDS slots, record offsets
    const bool cL = c < BB, cD = c >= BB && c < 2 * BB, cU = c >= 2 * BB && c < 3 * BB, cY = c == 3 * BB,
               cI = c > 3 * BB && c < NC;
    // the entry (or two, summed) of a block row that is this lane's column of the augmented block
    const int off1 = cL ? oL + c : (cD ? oDL + c - BB : (cU ? oU + c - 2 * BB : (cY ? oYL : oZ)));
    const int off2 = cD ? oDR + c - BB : (cY ? oYR : oZ);
    const int idc = cI ? c - 3 * BB - 1 : -1;            // identity column: 1 in the row that was row idc
    // results of node k kept in its (dead) row: E, F, z (fold_top's back-substitution; z's store)
    const bool keepE = half == 0 && (cL || cU), keepZ = half == 0 && cY;
    // record [Dinv | E | F | Ua | Lb][b][b] of node k: this lane's column of blocks 0 - 2
    const bool grec = half == 0 && (cL || cU || cI);
    const unsigned goff = (unsigned)((cI ? idc : (cL ? B2 + c : 2 * B2 + c - 2 * BB)) * 8);
    // neighbour update: side 0 works on a = k - s with U_a, side 1 on b = k + s with L_b (with both
    // halves holding the columns: half = side; else the lanes take the sides one after the other)
    const bool pvalid = cL || cU || cY;
    auto pacc_of = [&](int sd) { return sd == 0 ? (cL || cY) : (cU || cY); };
    auto pslot_of = [&](int sd) { return sd == 0 ? (cL ? oDR + c : (cU ? oU + c - 2 * BB : oYR))
                                                 : (cL ? oL + c : (cU ? oDL + c - 2 * BB : oYL)); };
    const int pslot0 = pslot_of(DUP ? half : 0), pslot1 = pslot_of(1);
    const bool pacc0 = pacc_of(DUP ? half : 0), pacc1 = pacc_of(1);
    const int opoff0 = (DUP ? half : 0) ? oL : oU;
    // copies of the blocks used in the elimination (record blocks 3, 4 = U_a, L_b before the update)
    int crel[NQ];
    bool cside[NQ], cok[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = lane + 64 * q, sd = e >= B2 ? 1 : 0, j = e - sd * B2;
        cok[q] = e < 2 * B2;
        cside[q] = sd != 0;
        crel[q] = cok[q] ? (j / BB) * RS + (sd ? oL : oU) + j % BB : 0;
    }

    TF_STAMP_REAL(a, 30);
    TF_STAMP(a, 0);
    // ---- load: records [node][L, D, U, second part of D][b][b] of a chunk are contiguous;
    //      every request is issued before the first value is used
    {
        const double* src = a.Ablk + (ch.nbase + ch.start) * REC;
        const int n3 = len * 3 * B2;
        constexpr int NIT = (C::MAXLEN * 3 * B2 + NT_MIN - 1) / NT_MIN;
        double v[NIT], v2[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * NT + tid;
            const int nd = i / (3 * B2), rem = i - nd * 3 * B2, blk = rem / B2, rc = rem - blk * B2;
            v[it] = i < n3 ? src[nd * REC + blk * B2 + rc] : 0.0;
            v2[it] = (i < n3 && blk == 1) ? src[nd * REC + 3 * B2 + rc] : 0.0;     // D = both parts
        }
        // position 0: the separator above; only its U block couples into this chunk
        const double* prev = a.Ablk + (ch.nbase + ch.gprev) * REC + 2 * B2;
        double p0 = 0.0, y0 = 0.0;
        if (tid < B2 && ch.has_prev) p0 = prev[tid];
        const double* ys = a.rhs + (ch.nbase + ch.start) * 2 * BB;
        if (with_rhs && tid < len * BB) y0 = ys[(tid / BB) * 2 * BB + tid % BB] + ys[(tid / BB) * 2 * BB + BB + tid % BB];
        // everything that is not loaded: position 0's L and DL, every DR, yR, yL and the zero slot
        for (int i = tid; i < (len + 1) * BB * (RS - 3 * BB); i += NT) {
            const int pr = i / (RS - 3 * BB), o = i - pr * (RS - 3 * BB);
            sRow[(pr / BB) * PS + (pr % BB) * RS + 3 * BB + o] = 0.0;
        }
        if (tid < 3 * B2) {
            const int blk = tid / B2, rc = tid - blk * B2, r = rc / BB, cc = rc - r * BB;
            row(0, r)[blk * BB + cc] = 0.0;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * NT + tid;
            const int nd = i / (3 * B2), rem = i - nd * 3 * B2, blk = rem / B2, rc = rem - blk * B2;
            const int r = rc / BB, cc = rc - r * BB;
            if (i < n3) row(nd + 1, r)[blk * BB + cc] = v[it] + v2[it];
        }
        __syncthreads();
        if (tid < B2) row(0, tid / BB)[oU + tid % BB] = p0;
        if (tid < len * BB) row(tid / BB + 1, tid % BB)[oYL] = y0;
        if (!L.periodic) {                           // no neighbour beyond the ends of a system
            if (ch.start == 0 && tid < B2) row(1, tid / BB)[oL + tid % BB] = 0.0;
            if (ch.start + len == L.N && tid >= 64 && tid < 64 + B2) row(pe, (tid - 64) / BB)[oU + (tid - 64) % BB] = 0.0;
        }
    }
    __syncthreads();

    bool ok = true;
    TF_STAMP(a, 1);
    int stamp_i = 2;
    for (int r = 0; (1 << r) <= mI; ++r) {
        const int s = 1 << r;
        const int nA = ((mI >> r) + 1) >> 1;         // nodes that go in this round (<= 8)
        for (int t = w; t < nA; t += nw) {
            const int k = s * (2 * t + 1);
            const int ia = k - s, ib = k + s <= mI ? k + s : pe;
            const double* rk = row(k, 0);
            double* ra = row(ia, 0);
            double* rb = row(ib, 0);
            double* nb = (DUP && half) ? rb : ra;    // the neighbour this half updates (one copy: the first of the two)
            if (r == 1) TF_STAMP(a, 44);
            // requested first, used last: what the neighbour rows hold now
            double old[BB], old1[DUP ? 1 : BB], cp[NQ];
#pragma unroll
            for (int i = 0; i < BB; ++i) old[i] = nb[i * RS + pslot0];
            if constexpr (!DUP) {
#pragma unroll
                for (int i = 0; i < BB; ++i) old1[i] = rb[i * RS + pslot1];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) cp[q] = (cside[q] ? rb : ra)[crel[q]];
            // ---- the block inversion of node k, rows in the remembered order
            unsigned code = (unsigned)__builtin_amdgcn_readfirstlane((int)sPerm[k]);
            double val[BB];
            bool done = false;
#pragma unroll 1
            for (int attempt = 0; attempt < 2 && !done; ++attempt) {
#pragma unroll
                for (int i = 0; i < BB; ++i) {
                    const int src = (int)(((code >> (3 * i)) & 7u) ^ (unsigned)i);       // wave-uniform
                    const double* rr = rk + src * RS;
                    const double v = rr[off1] + rr[off2];
                    val[i] = idc == src ? 1.0 : v;
                }
                bool grow = false;
                tf_gj3_step<BB, 0>(val, c, grow);
                double chk = 0.0;
#pragma unroll
                for (int i = 0; i < BB; ++i) chk += val[i] - val[i];                       // 0 for finite values, else NaN
                if (c < NC && !(chk == 0.0)) grow = true;
                done = __builtin_amdgcn_ballot_w64(grow) == 0ull;
                TF_COUNT(a, 41);
                if (!done) {
                    if (attempt == 1) { ok = false; break; }
                    // the order did not hold: partial pivoting on the rows in natural order (the lane layout
                    // of tf_cr2_hip.h) says which one to use, and it is kept for the next factorisation
                    TF_COUNT(a, 40);
                    constexpr int NJ = (NC + 7) / 8;
                    const int g = lane & 7, h = lane >> 3;
                    double sv[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int cc = h + 8 * j;
                        const double* rr = rk + (g < BB ? g : 0) * RS;
                        double v = 0.0;
                        if (g < BB) {
                            if (cc < BB) v = rr[oL + cc];
                            else if (cc < 2 * BB) v = rr[oDL + cc - BB] + rr[oDR + cc - BB];
                            else if (cc < 3 * BB) v = rr[oU + cc - 2 * BB];
                            else if (cc == 3 * BB) v = rr[oYL] + rr[oYR];
                            else if (cc < NC) v = cc - 3 * BB - 1 == g ? 1.0 : 0.0;
                        }
                        sv[j] = v;
                    }
                    const int myk = tf_gj_wave<BB, NC, BB>(sv, g, h, ok);
                    unsigned bits = 0;
                    if (g < BB && h == 0) {
                        if (myk >= 0) bits = ((unsigned)(g ^ myk)) << (3 * myk); else ok = false;
                    }
                    code = (unsigned)__builtin_amdgcn_readfirstlane((int)tf_group8_or(bits));
                    if (lane == 0) perm[k - 1] = code;
                }
            }
            if (r == 1) TF_STAMP(a, 45);
            // ---- lane c holds column c of [E | 1 | F | z | D^-1]: the record of node k, and E, F, z
            //      into its own (dead) row for the last level's back-substitution and z's store
            {
                double* rec = a.crf + (ch.nbase + ch.node(k)) * 5 * B2;                   // wave-uniform
                if (grec) {
#pragma unroll
                    for (int i = 0; i < BB; ++i) tf_st_u(rec, goff + (unsigned)(i * BB * 8), val[i]);
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if (cok[q]) tf_st_u(rec, (unsigned)((3 * B2 + lane + 64 * q) * 8), cp[q]);
                double* rkw = row(k, 0);
                if (keepZ || (keepE && a.fold_top)) {
#pragma unroll
                    for (int i = 0; i < BB; ++i) rkw[i * RS + off1] = val[i];
                }
            }
            if (r == 1) TF_STAMP(a, 46);
            // ---- the neighbours: out = -(U_a | L_b) * (my column of E, F, z)
            auto update = [&](double* nbp, int opoff, int pslot, bool pacc, const double* oldv) {
                const double* opb = nbp + opoff;
                double out[BB];
#pragma unroll
                for (int i = 0; i < BB; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int m = 0; m < BB; ++m) acc = tf_fma(-opb[i * RS + m], val[m], acc);
                    out[i] = acc;
                }
                if (pvalid) {
#pragma unroll
                    for (int i = 0; i < BB; ++i) nbp[i * RS + pslot] = (pacc ? oldv[i] : 0.0) + out[i];
                }
            };
            update(nb, opoff0, pslot0, pacc0, old);
            if (r == 1) TF_STAMP(a, 47);
            if constexpr (!DUP) update(rb, oL, pslot1, pacc1, old1);
            if (r == 1) TF_STAMP(a, 48);
        }
        TF_STAMP(a, stamp_i); ++stamp_i;
        __syncthreads();
        TF_STAMP(a, stamp_i); ++stamp_i;
    }

    // ---- z of the chunk's interior nodes: contiguous in a.zt
    if (with_rhs) {
        double* zt = a.zt + (ch.nbase + ch.start) * BB;
        for (int i = tid; i < mI * BB; i += NT) zt[i] = row(i / BB + 1, i % BB)[oYL];
    }
    // ---- this chunk's share of the next level's rows: node p gets (L, D, y) of position pe,
    //      node p-1 gets (U, second part of D, second part of y) of position 0
    for (int i = tid; i < 2 * BB * NO; i += NT) {
        const int side = i / (BB * NO), rem = i - side * BB * NO, r = rem / NO, o = rem - r * NO;
        const int nn = side == 0 ? ch.p : ch.pprev;
        double* rec = a.Anext + ((int64_t)ch.e * a.Lnext.N + nn) * REC;
        double* rr = a.rhsnext + ((int64_t)ch.e * a.Lnext.N + nn) * 2 * BB;
        const double* rs = row(side == 0 ? pe : 0, r);
        if (o < BB) rec[(side == 0 ? 0 : 2) * B2 + r * BB + o] = rs[(side == 0 ? oL : oU) + o];
        else if (o < 2 * BB) rec[(side == 0 ? 1 : 3) * B2 + r * BB + o - BB] = rs[oDL + o - BB] + rs[oDR + o - BB];
        else if (with_rhs) rr[(side == 0 ? 0 : BB) + r] = rs[oYL] + rs[oYR];
    }
    TF_STAMP(a, 20);
    if (a.fold_top) {
        // one chunk per system: what is left of rows 0 and pe couples the separator to
        // itself only (TfTopArgs): invert their sum here, and solve for the first rhs
        __syncthreads();
        constexpr int NCT = 2 * BB + 1, NJT = (NCT + 7) / 8;     // [S | y | I]
        const int g = lane & 7, h = lane >> 3;
        if (w == 0) {
            const double* r0 = row(0, 0);
            const double* rp = row(pe, 0);
            double val[NJT];
            unsigned code = sPerm[0];
            const unsigned code0 = code;
            const int myk = tf_gj_node<BB, NCT, 0>(val, g, h, code, ok, [&](int rr, int cc) {
                const int o = rr * RS;
                return rr < 0 ? 0.0 : (cc < BB ? rp[o + oL + cc] + (rp[o + oDL + cc] + rp[o + oDR + cc]) + r0[o + oU + cc]
                                                     + (r0[o + oDL + cc] + r0[o + oDR + cc])
                       : (cc == BB ? (rp[o + oYL] + rp[o + oYR]) + (r0[o + oYL] + r0[o + oYR])
                                   : ((cc < NCT && cc - BB - 1 == rr) ? 1.0 : 0.0))); });
            if (code != code0 && lane == 0) *perm_top = code;
            if (g < BB) {
                const int nsys = L.Ptot;             // P == 1
#pragma unroll
                for (int j = 0; j < NJT; ++j) {
                    const int cc = h + 8 * j;
                    if (cc > BB && cc < NCT) a.topAinv[(int64_t)(myk * BB + cc - BB - 1) * nsys + ch.e] = val[j];
                    if (cc == BB && with_rhs) {
                        // ... and the solution of the top block: the y slots take the solution
                        const double x = val[j];
                        a.topx[(int64_t)ch.e * BB + myk] = x;
                        a.x[(ch.nbase + ch.node(pe)) * BB + myk] = x;
                        row(pe, myk)[oYL] = x;
                        row(0, myk)[oYL] = ch.has_prev ? x : 0.0;
                    }
                }
            }
        }
        if (with_rhs) {
            // back-substitution of this level (tfk_cr_bwd_coop): E_k, F_k and z_k are in LDS
            __syncthreads();
            int r = 0;
            while ((2 << r) <= mI) ++r;
            for (; r >= 0; --r) {
                const int s = 1 << r;
                const int nA = ((mI >> r) + 1) >> 1;
                for (int t = w; t < nA; t += nw) {
                    if (!(h == 0 && g < BB)) continue;
                    const int k = s * (2 * t + 1);
                    const int kl = k - s, kr = k + s <= mI ? k + s : pe;
                    double* rk = row(k, g);
                    double xk = rk[oYL];
#pragma unroll
                    for (int m = 0; m < BB; ++m) {
                        xk = tf_fma(-rk[oL + m], row(kl, m)[oYL], xk);
                        xk = tf_fma(-rk[oU + m], row(kr, m)[oYL], xk);
                    }
                    rk[oYL] = xk;
                    a.x[(ch.nbase + ch.node(k)) * BB + g] = xk;
                }
                __syncthreads();
            }
        }
    }
    if (!ok && lane == 0) *a.status = 1;
    TF_STAMP(a, 21);
    TF_STAMP_REAL(a, 31);
}
