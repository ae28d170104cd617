// Cyclic reduction inside the chunks of a reduced level, one THREAD per node (blocks b x b, b small).
//
// This is the form the scalar models run on the GPU (b = mp <= 2: 256-node chunks, 256-thread
// workgroups, tf_entry_hip.h) and -- with one "thread" that takes every node in turn -- the form
// the test-only host emulation runs for every block size (tests/emu: so that the CPU suite covers
// the record formats, the two-part right-hand sides, the folded top block and the walks' assembled
// separator rows of the cyclic-reduction levels; the wave-cooperative kernels of tf_coop_hip.h /
// tf_cr2_hip.h for 3 <= b <= 8 compute the same quantities and are tested on the GPU).
// The algorithm, the chain positions and the stored quantities are described in tf_coop_hip.h.
#pragma once

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define TF_LDS __shared__
#define TF_BARRIER() __syncthreads()
// 8-byte agent-scope accesses (global_store / global_load ... sc1: written through to memory, read
// past the L1): the hand-off of a level's right-hand side share to another workgroup of the same
// launch (tfk_s_fwd) without writing back / invalidating whole caches
#define TF_ST_AGENT(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define TF_LD_AGENT(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// between two phases that one wavefront runs alone: LDS executes a wavefront's instructions in order,
// the compiler only has to keep them in order
#define TF_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// (loads stay where they are written: the compiler otherwise sinks one towards its only use)
#define TF_KEEP_ORDER() __asm__ volatile("" ::: "memory")
#else
#define TF_KEEP_ORDER() do {} while (0)
#define TF_WAVE_SYNC() do {} while (0)
#define TF_ST_AGENT(p, v) (*(p) = (v))
#define TF_LD_AGENT(p) (*(p))
#define TF_LDS static thread_local
#define TF_BARRIER() do {} while (0)
#endif

template <int BB>
struct TfCrChunk {
    int pg, e, p, len, start, mI, pe, gprev, pprev;
    bool has_prev;
    int64_t nbase;                                // first node record of system e
    TF_DEVICE_M TfCrChunk(const TfLayout& L, int chunk) {
        pg = chunk;
        e = pg / L.P; p = pg - e * L.P;
        len = tf_len(L, p); start = tf_start(L, p);
        mI = len - 1; pe = len;
        has_prev = L.periodic || p > 0;
        gprev = start > 0 ? start - 1 : L.N - 1;
        pprev = p > 0 ? p - 1 : L.P - 1;
        nbase = (int64_t)e * L.N;
    }
    TF_DEVICE_M int node(int pos) const { return start + pos - 1; }   // pos >= 1
};


// ===========================================================================
// Cyclic reduction for scalar models (b = mp <= 2): one thread per node
// ===========================================================================
// Same algorithm, level format and stored quantities as tfk_cr_* above; the blocks are
// 1 x 1 or 2 x 2, so a node is one thread's work and a chunk can be long: 256 nodes
// per 256-thread workgroup, 8 rounds, levels shrink 256x (N = 1e6: 31250 -> 123 -> 1
// chunks, where the chunk walks needed 6 levels of three kernels each).
template <int BB> struct TfCrs {
    static constexpr int MAXLEN = TF_CRS_TOPLEN(BB), NPOS = MAXLEN + 1;
    // LDS places of the solve kernels (tfk_crs_fwd / tfk_crs_bwd): a vector [pos][b]; the stored reduction by
    // FIELD, [5][pos][b][b] (Dinv, E, F, Ua, Lb) -- a node's five blocks then sit at one address plus constants
    // (a round is bound by the instructions one wavefront issues, profiles/r04_scalar_stamps.txt: padding
    // the arrays against the bank conflicts of the rounds' power-of-two strides changed nothing, fewer
    // address instructions do).
    static constexpr int VSIZE = NPOS * BB, FSIZE = NPOS * 5 * BB * BB;
    TF_DEVICE_M static int v(int pos, int q) { return pos * BB + q; }
    TF_DEVICE_M static int f(int pos, int fld = 0) { return (fld * NPOS + pos) * BB * BB; }
    // ... of entry i of a chunk's records as they are in memory ([node][5][b][b], from the chunk's first node)
    TF_DEVICE_M static int fmem(int i) {
        const int nd = i / (5 * BB * BB), rem = i - nd * 5 * BB * BB, fld = rem / (BB * BB);
        return f(nd + 1, fld) + rem - fld * BB * BB;
    }
};

// small-block helpers on rows stored as [r * BB + c] in LDS / global memory
template <int BB> TF_DEVICE void tf_ld_blk(double (&m)[BB][BB], const double* p) {
#pragma unroll
    for (int r = 0; r < BB; ++r)
#pragma unroll
        for (int c = 0; c < BB; ++c) m[r][c] = p[r * BB + c];
}
template <int BB> TF_DEVICE void tf_st_blk(double* p, const double (&m)[BB][BB]) {
#pragma unroll
    for (int r = 0; r < BB; ++r)
#pragma unroll
        for (int c = 0; c < BB; ++c) p[r * BB + c] = m[r][c];
}

template <int BB, int NT>
TF_DEVICE void tfk_crs_factor(const TfLevelArgs& a, int chunk, int tid) {
    typedef TfCrs<BB> C;
    constexpr int NPOS = C::NPOS, B2 = BB * BB, REC = 4 * B2;
    const TfLayout& L = a.L;
    const TfCrChunk<BB> ch(L, chunk);
    const int mI = ch.mI, pe = ch.pe, len = ch.len;
    const bool with_rhs = a.cr_rhs != 0;
    TF_LDS double sL[NPOS * B2], sD[NPOS * B2], sU[NPOS * B2], sY[NPOS * BB], sZ[NPOS * BB];

    // ---- load the chain (records of a chunk are contiguous)
    {
        const double* src = a.Ablk + (ch.nbase + ch.start) * REC;
        const bool cut_first = !L.periodic && ch.start == 0, cut_last = !L.periodic && ch.start + len == L.N;
        for (int i = tid; i < len * B2; i += NT) {
            const int nd = i / B2, rc = i - nd * B2;
            const double* rec = src + (int64_t)nd * REC;
            sL[(nd + 1) * B2 + rc] = (cut_first && nd == 0) ? 0.0 : rec[rc];
            sD[(nd + 1) * B2 + rc] = rec[B2 + rc] + rec[3 * B2 + rc];
            sU[(nd + 1) * B2 + rc] = (cut_last && nd == len - 1) ? 0.0 : rec[2 * B2 + rc];
        }
        const double* prev = a.Ablk + (ch.nbase + ch.gprev) * REC;
        for (int i = tid; i < B2; i += NT) {
            sL[i] = 0.0; sD[i] = 0.0;
            sU[i] = ch.has_prev ? prev[2 * B2 + i] : 0.0;
        }
        const double* ys = a.rhs + (ch.nbase + ch.start) * 2 * BB;
        for (int i = tid; i < (len + 1) * BB; i += NT) {
            const int pos = i / BB, r = i - pos * BB;
            sY[i] = (with_rhs && pos > 0) ? ys[(pos - 1) * 2 * BB + r] + ys[(pos - 1) * 2 * BB + BB + r] : 0.0;
        }
    }
    TF_BARRIER();

    bool ok = true;
    for (int s = 1; s <= mI; s <<= 1) {
        // ---- phase A: every second remaining interior node goes
        const int nA = (mI / s + 1) / 2;
        for (int j = tid; j < nA; j += NT) {
            const int k = s * (2 * j + 1);
            double D[BB][BB], Di[BB][BB], Lk[BB][BB], Uk[BB][BB], E[BB][BB], F[BB][BB], z[BB], y[BB];
            tf_ld_blk<BB>(D, sD + k * B2); tf_ld_blk<BB>(Lk, sL + k * B2); tf_ld_blk<BB>(Uk, sU + k * B2);
#pragma unroll
            for (int r = 0; r < BB; ++r) y[r] = sY[k * BB + r];
            ok = tf_blk_inverse<BB>(D, Di) && ok;
            tf_mm<BB>(E, Di, Lk); tf_mm<BB>(F, Di, Uk); tf_mv<BB>(z, Di, y);
            double* rec = a.crf + (ch.nbase + ch.node(k)) * 5 * B2;
            tf_st_blk<BB>(rec, Di); tf_st_blk<BB>(rec + B2, E); tf_st_blk<BB>(rec + 2 * B2, F);
            tf_st_blk<BB>(sL + k * B2, E); tf_st_blk<BB>(sU + k * B2, F);
#pragma unroll
            for (int r = 0; r < BB; ++r) {
                sZ[k * BB + r] = z[r];
                if (with_rhs) a.zt[(ch.nbase + ch.node(k)) * BB + r] = z[r];
            }
        }
        TF_BARRIER();
        // ---- phase B: the neighbours take the update (tasks as in tfk_cr_factor_v3)
        const int nB = mI / (2 * s);
        for (int t = tid; t <= nB; t += NT) {
            const bool ends = t == nB;
            const int aa = 2 * s * (t + 1), nq = mI / s;
            const int aL = ends ? pe : aa, aU = ends ? 0 : aa;
            const bool vL = ends ? (nq & 1) != 0 : true;
            const bool vR = ends ? true : aa + s <= mI;
            const int kL = ends ? nq * s : aa - s, kR = ends ? s : aa + s;
            if (vL) {
                double Lr[BB][BB], Ek[BB][BB], Fk[BB][BB], Dn[BB][BB], nl[BB][BB], zk[BB], yv[BB];
                tf_ld_blk<BB>(Lr, sL + aL * B2); tf_ld_blk<BB>(Ek, sL + kL * B2); tf_ld_blk<BB>(Fk, sU + kL * B2);
                tf_ld_blk<BB>(Dn, sD + aL * B2);
                tf_st_blk<BB>(a.crf + (ch.nbase + ch.node(kL)) * 5 * B2 + 4 * B2, Lr);
                tf_blk_zero<BB>(nl);
                tf_mm_sub<BB>(nl, Lr, Ek);            // -L E_kL
                tf_mm_sub<BB>(Dn, Lr, Fk);            // D - L F_kL
#pragma unroll
                for (int r = 0; r < BB; ++r) { zk[r] = sZ[kL * BB + r]; yv[r] = sY[aL * BB + r]; }
                tf_mv_sub<BB>(yv, Lr, zk);
                tf_st_blk<BB>(sL + aL * B2, nl); tf_st_blk<BB>(sD + aL * B2, Dn);
#pragma unroll
                for (int r = 0; r < BB; ++r) sY[aL * BB + r] = yv[r];
            }
            if (vR) {
                double Ur[BB][BB], Ek[BB][BB], Fk[BB][BB], Dn[BB][BB], nu[BB][BB], zk[BB], yv[BB];
                tf_ld_blk<BB>(Ur, sU + aU * B2); tf_ld_blk<BB>(Ek, sL + kR * B2); tf_ld_blk<BB>(Fk, sU + kR * B2);
                tf_ld_blk<BB>(Dn, sD + aU * B2);
                tf_st_blk<BB>(a.crf + (ch.nbase + ch.node(kR)) * 5 * B2 + 3 * B2, Ur);
                tf_blk_zero<BB>(nu);
                tf_mm_sub<BB>(nu, Ur, Fk);            // -U F_kR
                tf_mm_sub<BB>(Dn, Ur, Ek);            // D - U E_kR
#pragma unroll
                for (int r = 0; r < BB; ++r) { zk[r] = sZ[kR * BB + r]; yv[r] = sY[aU * BB + r]; }
                tf_mv_sub<BB>(yv, Ur, zk);
                tf_st_blk<BB>(sU + aU * B2, nu); tf_st_blk<BB>(sD + aU * B2, Dn);
#pragma unroll
                for (int r = 0; r < BB; ++r) sY[aU * BB + r] = yv[r];
            }
        }
        TF_BARRIER();
    }

    // ---- this chunk's share of the next level's rows
    for (int side = tid; side < 2; side += NT) {       // (loops over "threads": NT = 1 on the host)
        const int nn = side == 0 ? ch.p : ch.pprev;
        double* rec = a.Anext + ((int64_t)ch.e * a.Lnext.N + nn) * REC;
        double* rr = a.rhsnext + ((int64_t)ch.e * a.Lnext.N + nn) * 2 * BB;
        for (int i = 0; i < B2; ++i) {
            if (side == 0) { rec[i] = sL[pe * B2 + i]; rec[B2 + i] = sD[pe * B2 + i]; }
            else { rec[2 * B2 + i] = sU[i]; rec[3 * B2 + i] = sD[i]; }
        }
        if (with_rhs)
            for (int r = 0; r < BB; ++r) { if (side == 0) rr[r] = sY[pe * BB + r]; else rr[BB + r] = sY[r]; }
    }
    if (a.fold_top) {
        // one chunk per system: invert what is left, solve for the first rhs, back-substitute
        TF_BARRIER();
        if (tid == 0) {
            double S[BB][BB], Si[BB][BB];
#pragma unroll
            for (int r = 0; r < BB; ++r)
#pragma unroll
                for (int c = 0; c < BB; ++c) {
                    const int i = r * BB + c;
                    S[r][c] = sL[pe * B2 + i] + sD[pe * B2 + i] + sU[i] + sD[i];
                }
            ok = tf_blk_inverse<BB>(S, Si) && ok;
            const int nsys = L.Ptot;
#pragma unroll
            for (int r = 0; r < BB; ++r)
#pragma unroll
                for (int c = 0; c < BB; ++c) a.topAinv[(int64_t)(r * BB + c) * nsys + ch.e] = Si[r][c];
            if (with_rhs) {
                double yt[BB], x[BB];
#pragma unroll
                for (int r = 0; r < BB; ++r) yt[r] = sY[pe * BB + r] + sY[r];
                tf_mv<BB>(x, Si, yt);
#pragma unroll
                for (int r = 0; r < BB; ++r) {
                    a.topx[(int64_t)ch.e * BB + r] = x[r];
                    a.x[(ch.nbase + ch.node(pe)) * BB + r] = x[r];
                    sY[pe * BB + r] = x[r];
                    sY[r] = ch.has_prev ? x[r] : 0.0;
                }
            }
        }
        TF_BARRIER();
        if (with_rhs) {
            int s = 1;
            while (2 * s <= mI) s <<= 1;
            for (; s >= 1; s >>= 1) {
                const int nA = (mI / s + 1) / 2;
                for (int j = tid; j < nA; j += NT) {
                    const int k = s * (2 * j + 1);
                    const int kl = k - s, kr = k + s <= mI ? k + s : pe;
                    double E[BB][BB], F[BB][BB], xl[BB], xr[BB], xk[BB];
                    tf_ld_blk<BB>(E, sL + k * B2); tf_ld_blk<BB>(F, sU + k * B2);
#pragma unroll
                    for (int r = 0; r < BB; ++r) { xk[r] = sZ[k * BB + r]; xl[r] = sY[kl * BB + r]; xr[r] = sY[kr * BB + r]; }
                    tf_mv_sub<BB>(xk, E, xl); tf_mv_sub<BB>(xk, F, xr);
#pragma unroll
                    for (int r = 0; r < BB; ++r) { sY[k * BB + r] = xk[r]; a.x[(ch.nbase + ch.node(k)) * BB + r] = xk[r]; }
                }
                TF_BARRIER();
            }
        }
    }
    if (!ok) *a.status = 1;
}

// the stored reduction of a chunk's nodes into LDS (what tfk_crs_fwd / tfk_crs_bwd start with)
template <int BB, int NT>
TF_DEVICE void tfk_crs_stage(const TfLevelArgs& a, int chunk, int tid, double* sF) {
    const TfCrChunk<BB> ch(a.L, chunk);
    const double* src = a.crf + (ch.nbase + ch.start) * 5 * BB * BB;
    for (int i = tid; i < ch.len * 5 * BB * BB; i += NT) sF[TfCrs<BB>::fmem(i)] = src[i];
}
// ... in two parts, for a caller with work to do while the loads are in flight: request (into registers),
// then put into LDS
template <int BB, int NT, int LEN> struct TfCrsStaged { double v[(LEN * 5 * BB * BB + NT - 1) / NT]; };
template <int BB, int NT, int LEN>
TF_DEVICE void tfk_crs_stage_request(const TfLevelArgs& a, int chunk, int tid, TfCrsStaged<BB, NT, LEN>& regs) {
    const TfCrChunk<BB> ch(a.L, chunk);
    const double* src = a.crf + (ch.nbase + ch.start) * 5 * BB * BB;
    constexpr int NQ = (LEN * 5 * BB * BB + NT - 1) / NT;
#pragma unroll
    for (int q = 0; q < NQ; ++q) regs.v[q] = tid + q * NT < ch.len * 5 * BB * BB ? src[tid + q * NT] : 0.0;
}
template <int BB, int NT, int LEN>
TF_DEVICE void tfk_crs_stage_put(const TfLevelArgs& a, int chunk, int tid, const TfCrsStaged<BB, NT, LEN>& regs, double* sF) {
    const TfCrChunk<BB> ch(a.L, chunk);
    constexpr int NQ = (LEN * 5 * BB * BB + NT - 1) / NT;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = tid + q * NT;
        if (i < ch.len * 5 * BB * BB) sF[TfCrs<BB>::fmem(i)] = regs.v[q];
    }
}

// STAGED: the caller has put the stored reduction into `sF` already (tfk_crs_stage: a kernel that
// walks level 1 first requests it before the walks, tfk_s_fwd)
// AGENT_OUT / AGENT_IN: this chunk's share of the next level's right-hand side is stored / this level's
// right-hand side is loaded with agent-scope accesses (a producer / the consumer inside one launch)
// ys_own: the chunk's right-hand side records [len][2][b] where the caller's walks left them (LDS) instead
// of a.rhs; top_pre: the caller's earlier load of this lane's entry of the folded top block's inverse
// (thread 0 .. b*b-1: entry tid; with fold_top)
template <int BB, int NT, bool STAGED = false, bool AGENT_OUT = false, bool AGENT_IN = false, bool PRE = false>
TF_DEVICE void tfk_crs_fwd(const TfLevelArgs& a, int chunk, int tid, double* sF_staged = nullptr,
                           const double* ys_own = nullptr, const double* top_pre = nullptr) {
    typedef TfCrs<BB> C;
    constexpr int NPOS = C::NPOS, B2 = BB * BB;
    const TfLayout& L = a.L;
    const TfCrChunk<BB> ch(L, chunk);
    const int mI = ch.mI, pe = ch.pe, len = ch.len;
    TF_LDS double sF_own[STAGED ? 1 : C::FSIZE];             // stored reduction of the chunk's nodes
    TF_LDS double sY[C::VSIZE], sZ[C::VSIZE];
    double* const sF = STAGED ? sF_staged : sF_own;
    {
        if (!STAGED) tfk_crs_stage<BB, NT>(a, chunk, tid, sF);
        const double* ys = ys_own ? ys_own : a.rhs + (ch.nbase + ch.start) * 2 * BB;
        for (int i = tid; i < (len + 1) * BB; i += NT) {
            const int pos = i / BB, r = i - pos * BB;
            if (AGENT_IN) sY[C::v(pos, r)] = pos > 0 ? TF_LD_AGENT(ys + (pos - 1) * 2 * BB + r) + TF_LD_AGENT(ys + (pos - 1) * 2 * BB + BB + r) : 0.0;
            else sY[C::v(pos, r)] = pos > 0 ? ys[(pos - 1) * 2 * BB + r] + ys[(pos - 1) * 2 * BB + BB + r] : 0.0;
        }
    }
    TF_BARRIER();
    // One phase per round: the lane that updates node aa's right-hand side in round r is the one that goes
    // on to z_aa = Dinv_aa y_aa when aa leaves in round r + 1 (every second one does), so that the update
    // and the next round's first half need one set of LDS loads and one synchronisation -- a wavefront
    // alone on its SIMD issues an instruction every ~5 cycles and waits ~130 for LDS: a round of two
    // phases took 1150 cycles (profiles/r04_scalar_stamps.txt).
    // Rounds with at most 64 tasks (all but the first two of a 256-node chunk) are run by the first
    // wavefront ALONE in a loop of its own, with no workgroup barrier and none of their bookkeeping; one
    // barrier lets the other wavefronts see the result.
    double* const zt0 = a.zt + (ch.nbase + ch.start - 1) * BB;       // (wave-uniform: z / x of position k at [k * BB])
    double* const x0 = a.x + (ch.nbase + ch.start - 1) * BB;
    auto task = [&](int r, int t) {
        const int s = 1 << r, nq = mI >> r, nB = nq >> 1;
        const int nA1 = (2 << r) <= mI ? ((mI >> (r + 1)) + 1) >> 1 : 0;      // nodes that leave in round r + 1
        const bool ends = t == nB;
        const int aa = 2 * s * (t + 1);
        const int aL = ends ? pe : aa, aU = ends ? 0 : aa;
        const bool vL = ends ? (nq & 1) != 0 : true;
        const bool vR = ends ? true : aa + s <= mI;
        const int kL = vL ? (ends ? nq * s : aa - s) : 1, kR = vR ? (ends ? s : aa + s) : 1;
        const bool next = !ends && (t & 1) == 0 && (t >> 1) < nA1;    // aa = 2 s (2 j' + 1), j' = t / 2
        // every load of the task before the first use: one LDS latency
        double Lb[BB][BB], Ua[BB][BB], Di[BB][BB], zl[BB], zr[BB], yl[BB], yu[BB];
        tf_ld_blk<BB>(Lb, sF + C::f(kL, 4));
        tf_ld_blk<BB>(Ua, sF + C::f(kR, 3));
        tf_ld_blk<BB>(Di, sF + C::f(ends ? 1 : aa));
#pragma unroll
        for (int q = 0; q < BB; ++q) { zl[q] = sZ[C::v(kL, q)]; zr[q] = sZ[C::v(kR, q)]; yl[q] = sY[C::v(aL, q)]; yu[q] = sY[C::v(aU, q)]; }
        TF_KEEP_ORDER();
        if (vL) tf_mv_sub<BB>(yl, Lb, zl);
        if (!ends) {
            // (one node takes both updates, the lower neighbour's first)
            if (vR) tf_mv_sub<BB>(yl, Ua, zr);
#pragma unroll
            for (int q = 0; q < BB; ++q) sY[C::v(aL, q)] = yl[q];
            if (next) {
                double z[BB];
                tf_mv<BB>(z, Di, yl);
#pragma unroll
                for (int q = 0; q < BB; ++q) { sZ[C::v(aa, q)] = z[q]; zt0[aa * BB + q] = z[q]; }
            }
        } else {
            if (vR) tf_mv_sub<BB>(yu, Ua, zr);
#pragma unroll
            for (int q = 0; q < BB; ++q) { if (vL) sY[C::v(aL, q)] = yl[q]; if (vR) sY[C::v(aU, q)] = yu[q]; }
        }
    };
    if (mI >= 1) {
        const int nA = (mI + 1) >> 1;                        // round 0's first half: the odd nodes
        for (int j = tid; j < nA; j += NT) {
            const int k = 2 * j + 1;
            double Di[BB][BB], y[BB], z[BB];
            tf_ld_blk<BB>(Di, sF + C::f(k));
#pragma unroll
            for (int q = 0; q < BB; ++q) y[q] = sY[C::v(k, q)];
            tf_mv<BB>(z, Di, y);
#pragma unroll
            for (int q = 0; q < BB; ++q) { sZ[C::v(k, q)] = z[q]; zt0[k * BB + q] = z[q]; }
        }
        TF_BARRIER();
    }
    constexpr bool WAVES = NT > 64;                          // (the host emulation: one "thread", no wavefronts)
    int r = 0;
    for (; (1 << r) <= mI && (!WAVES || ((mI >> r) >> 1) + 1 > 64); ++r) {
        const int nB = (mI >> r) >> 1;
        for (int t = tid; t <= nB; t += NT) task(r, t);
        TF_BARRIER();
    }
    const int r_solo = r;                                    // the first round the first wavefront runs alone
    if (WAVES && (1 << r) <= mI) {
        if (tid < 64)
            for (; (1 << r) <= mI; ++r) {
                if (tid <= ((mI >> r) >> 1)) task(r, tid);
                TF_WAVE_SYNC();
            }
        if (!a.fold_top) TF_BARRIER();
    }
    if (a.fold_top) {
        // (everything up to the barrier below is the first wavefront's, in program order)
        if (tid == 0) {
            const int nsys = L.Ptot;
            double Si[BB][BB], yt[BB], x[BB];
#pragma unroll
            for (int q = 0; q < BB; ++q) {
                yt[q] = sY[C::v(pe, q)] + sY[C::v(0, q)];
#pragma unroll
                for (int c = 0; c < BB; ++c) Si[q][c] = PRE ? top_pre[q * BB + c] : a.topAinv[(int64_t)(q * BB + c) * nsys + ch.e];
            }
            tf_mv<BB>(x, Si, yt);
#pragma unroll
            for (int q = 0; q < BB; ++q) {
                a.topx[(int64_t)ch.e * BB + q] = x[q];
                x0[pe * BB + q] = x[q];
                sY[C::v(pe, q)] = x[q];
                sY[C::v(0, q)] = ch.has_prev ? x[q] : 0.0;
            }
        }
        auto back = [&](int rr, int j) {
            const int s = 1 << rr, k = s * (2 * j + 1);
            const int kl = k - s, kr = k + s <= mI ? k + s : pe;
            double E[BB][BB], F[BB][BB], xl[BB], xr[BB], xk[BB];
            tf_ld_blk<BB>(E, sF + C::f(k, 1)); tf_ld_blk<BB>(F, sF + C::f(k, 2));
#pragma unroll
            for (int q = 0; q < BB; ++q) { xk[q] = sZ[C::v(k, q)]; xl[q] = sY[C::v(kl, q)]; xr[q] = sY[C::v(kr, q)]; }
            tf_mv_sub<BB>(xk, E, xl); tf_mv_sub<BB>(xk, F, xr);
#pragma unroll
            for (int q = 0; q < BB; ++q) { sY[C::v(k, q)] = xk[q]; x0[k * BB + q] = xk[q]; }
        };
        int rb = 0;
        while ((2 << rb) <= mI) ++rb;                        // the last forward round is the first one backwards
        if (WAVES && rb >= r_solo && mI >= 1) {
            // the rounds the first wavefront ran alone forwards, alone backwards too
            if (tid < 64) {
                TF_WAVE_SYNC();
                for (; rb >= r_solo; --rb) {
                    if (tid < (((mI >> rb) + 1) >> 1)) back(rb, tid);
                    TF_WAVE_SYNC();
                }
            } else rb = r_solo - 1;
        }
        TF_BARRIER();
        for (; rb >= 0 && mI >= 1; --rb) {
            const int nA = ((mI >> rb) + 1) >> 1;
            for (int j = tid; j < nA; j += NT) back(rb, j);
            TF_BARRIER();
        }
    } else {
        // (the first wavefront's threads: what they read was written by it, or before a barrier)
        for (int side = tid; side < 2; side += NT) {
            const int nn = side == 0 ? ch.p : ch.pprev;
            double* rr = a.rhsnext + ((int64_t)ch.e * a.Lnext.N + nn) * 2 * BB;
            for (int q = 0; q < BB; ++q) {
                double* dst = side == 0 ? rr + q : rr + BB + q;
                const double v = side == 0 ? sY[C::v(pe, q)] : sY[C::v(0, q)];
                if (AGENT_OUT) TF_ST_AGENT(dst, v); else *dst = v;
            }
        }
    }
}

// store_prev: the solution of the separator above (a node of the previous chunk, known from the next
// level) is written into this level's x as well -- a caller that goes on to the level below inside
// the same workgroup (tfk_s_bwd) finds both separators of its first chunk without another workgroup
template <int BB, int NT>
TF_DEVICE void tfk_crs_bwd(const TfLevelArgs& a, int chunk, int tid, bool store_prev = false) {
    typedef TfCrs<BB> C;
    constexpr int NPOS = C::NPOS, B2 = BB * BB;
    const TfLayout& L = a.L;
    const TfCrChunk<BB> ch(L, chunk);
    const int mI = ch.mI, pe = ch.pe, len = ch.len;
    TF_LDS double sF[C::FSIZE];
    TF_LDS double sX[C::VSIZE], sZ[C::VSIZE];
    {
        const double* src = a.crf + (ch.nbase + ch.start) * 5 * B2;
        for (int i = tid; i < len * 5 * B2; i += NT) sF[C::fmem(i)] = src[i];
        const double* zs = a.zt + (ch.nbase + ch.start) * BB;
        for (int i = tid; i < len * BB; i += NT) sZ[BB + i] = zs[i];
    }
    for (int r = tid; r < BB; r += NT) {
        const double xs = a.xnext[((int64_t)ch.e * a.Lnext.N + ch.p) * BB + r];
        sX[C::v(pe, r)] = xs;
        a.x[(ch.nbase + ch.node(pe)) * BB + r] = xs;
        sX[C::v(0, r)] = ch.has_prev ? a.xnext[((int64_t)ch.e * a.Lnext.N + ch.pprev) * BB + r] : 0.0;
        if (store_prev && ch.has_prev) a.x[(ch.nbase + ch.gprev) * BB + r] = sX[C::v(0, r)];
    }
    TF_BARRIER();
    double* const x0 = a.x + (ch.nbase + ch.start - 1) * BB;
    auto back = [&](int rr, int j) {
        const int s = 1 << rr, k = s * (2 * j + 1);
        const int kl = k - s, kr = k + s <= mI ? k + s : pe;
        double E[BB][BB], F[BB][BB], xl[BB], xr[BB], xk[BB];
        tf_ld_blk<BB>(E, sF + C::f(k, 1)); tf_ld_blk<BB>(F, sF + C::f(k, 2));
#pragma unroll
        for (int q = 0; q < BB; ++q) { xk[q] = sZ[C::v(k, q)]; xl[q] = sX[C::v(kl, q)]; xr[q] = sX[C::v(kr, q)]; }
        tf_mv_sub<BB>(xk, E, xl); tf_mv_sub<BB>(xk, F, xr);
#pragma unroll
        for (int q = 0; q < BB; ++q) { sX[C::v(k, q)] = xk[q]; x0[k * BB + q] = xk[q]; }
    };
    // (rounds of at most 64 tasks by the first wavefront alone, in a loop of their own: see tfk_crs_fwd)
    constexpr bool WAVES = NT > 64;
    int rb = 0;
    while ((2 << rb) <= mI) ++rb;
    if (mI < 1) rb = -1;
    if (WAVES) {
        int r_many = -1;                                     // the highest round with more than 64 tasks
        for (int q = 0; q <= rb; ++q) if ((((mI >> q) + 1) >> 1) > 64) r_many = q;
        if (rb > r_many) {
            if (tid < 64)
                for (; rb > r_many; --rb) {
                    if (tid < (((mI >> rb) + 1) >> 1)) back(rb, tid);
                    TF_WAVE_SYNC();
                }
            else rb = r_many;
            TF_BARRIER();
        }
    }
    for (; rb >= 0; --rb) {
        const int nA = ((mI >> rb) + 1) >> 1;
        for (int j = tid; j < nA; j += NT) back(rb, j);
        TF_BARRIER();
    }
}
