// Cyclic-reduction factorisation of a reduced-level chunk with a WAVEFRONT per node (3 <= b <= 8).
//
// The algorithm, task lists and stored quantities (a.crf, a.zt: what tfk_cr_fwd / tfk_cr_bwd read)
// are described at the top of the cyclic-reduction section of tf_coop_hip.h.  The kernel is built
// around what these levels cost: the latency of a round of dependent block inversions.  With 8
// lanes per node (the round-1 version) every pivot step was ~160 instructions of ONE wavefront
// (measured: 1300 cycles per pivot, 11 000 per round, also with a DPP pivot search and a
// conflict-free LDS image: profiles/README.md).  Here:
//   * Gauss-Jordan on the augmented row [L | D | U | y | I] of the node that goes: the lanes end
//     with D^-1, E = D^-1 L, F = D^-1 U and z = D^-1 y, no product phase after the inversion;
//   * the pivot row of a step is found with three DPP max steps inside an 8-lane group (key =
//     float magnitude with the lane number in its low bits) and reaches the other rows through
//     ds_bpermute (lane crossbar: no LDS memory, no barrier);
//   * the chain lives in LDS as records per (position, block row) with a stride chosen against
//     bank conflicts, 16 KB per chunk instead of 36.
// HIP only.
#pragma once

// Row / position strides (in doubles) of the chain in LDS.  Every access is 8 bytes wide,
// i.e. LDS has 32 double-wide banks for the 32 lanes of a half wavefront; the strides are
// the ones for which the access patterns of all rounds (own row of the nodes that go, rows of
// the updated neighbours, broadcast reads of E / F / z) hit distinct banks: average conflict
// degree 1.01 - 1.04, worst 2 (searched over RS <= RW + 3, PS <= b*RS + 15; the natural
// [pos][4][b][b] image of the round-1 version is 4- to 8-way).
template <int BB> struct TfCr2Stride;
template <> struct TfCr2Stride<3> { static constexpr int RS = 11, PS = 38; };
template <> struct TfCr2Stride<4> { static constexpr int RS = 13, PS = 54; };
template <> struct TfCr2Stride<5> { static constexpr int RS = 17, PS = 93; };
template <> struct TfCr2Stride<6> { static constexpr int RS = 19, PS = 121; };
template <> struct TfCr2Stride<7> { static constexpr int RS = 23, PS = 166; };
template <> struct TfCr2Stride<8> { static constexpr int RS = 25, PS = 202; };

template <int BB> struct TfCr2 {
    static constexpr int MAXLEN = TF_CR_MAXLEN, NPOS = MAXLEN + 1;
    static constexpr int RW = 3 * BB + 1;                 // L | D | U | y of one block row
    static constexpr int RS = TfCr2Stride<BB>::RS;        // row stride
    static constexpr int PS = TfCr2Stride<BB>::PS;        // position stride
    static constexpr int oL = 0, oD = BB, oU = 2 * BB, oY = 3 * BB;
    static_assert(RS >= RW && PS >= BB * RS, "strides");
};

// max over the 8 lanes of a group (lanes 8q .. 8q+7 of the wavefront), result in every lane
__device__ __forceinline__ unsigned tf_group8_max(unsigned v) {
    unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);            // quad_perm [2,3,0,1]
    v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);           // row_half_mirror
    return o > v ? o : v;
}

// ===========================================================================
// One WAVEFRONT per node of the round, 8 (or 4) wavefronts per chunk
// ===========================================================================
// The 64 lanes of a wavefront share one node: lane (g, h) = (block row, column slice) holds
// the augmented entries M[g][h + 8j] of  [L | D | U | y | I].  Per pivot step a lane
// touches (4b+1)/8 ~ 4 entries; the pivot row index is wave-uniform (v_readlane), the pivot
// row reaches the other rows with ds_bpermute (lane crossbar, no LDS memory, no barrier).
// A chunk is a workgroup of 8 wavefronts = the 8 nodes that go in round 1 (later rounds
// leave wavefronts idle at the barriers; 4 wavefronts on levels with more chunks than the GPU
// holds at once).

// 1/x by v_rcp_f64 (about 26 bits) and two Newton steps: within an ulp or two of the IEEE
// quotient for normal x, a third of its instructions; 1/0 = inf like the division
__device__ __forceinline__ double tf_rcp_newton(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
__device__ __forceinline__ double tf_bperm_f64(int byte_addr, double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double tf_readlane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// Gauss-Jordan with partial pivoting on a b x NC augmented matrix held by one wavefront:
// lane (g = lane & 7, h = lane >> 3) has val[j] = M[g][h + 8j]; the block to invert is columns
// PC0 .. PC0+b-1.  Returns the row of [.. | M^-1 .. ] this lane's row ended as (rows are never
// moved: the unused row with the largest pivot candidate serves pivot k).  Every lane of the
// wavefront must be active.
template <int BB, int NC, int PC0>
__device__ __forceinline__ int tf_gj_wave(double (&val)[(NC + 7) / 8], int g, int h, bool& ok) {
    constexpr int NJ = (NC + 7) / 8;
    const int lane = (h << 3) | g;
    int myk = -1;
#pragma unroll
    for (int kk = 0; kk < BB; ++kk) {
        const int pc = PC0 + kk, ph = pc & 7, pj = pc >> 3;
        unsigned key = 0;
        if (h == ph && g < BB && myk < 0) {
            const double cand = val[pj];
            const float mag = (float)tf_abs(cand);
            unsigned bits = __float_as_uint(mag);
            if (bits == 0 && cand != 0.0) bits = 8;                // underflow is not a zero pivot
            if (mag != mag) bits = 0x7f800000u;                    // NaN: taken, reported through ok
            key = (bits & ~7u) | (unsigned)(7 - g);
        }
        // the reciprocal of every candidate is formed while the pivot is being found: the
        // division is off the critical path (only the pivot row's one is used)
        const double rcand = tf_rcp_newton(val[pj]);
        const double mult = tf_bperm_f64(((ph << 3) | g) << 2, val[pj]);   // my row's entry of the pivot column
        const unsigned best = (unsigned)__builtin_amdgcn_readlane((int)tf_group8_max(key), ph * 8);
        const int piv = 7 - (int)(best & 7u);                      // wave-uniform
        const int plane = (ph << 3) | piv;                         // lane that holds M[piv][pc]
        const double rp = tf_readlane_f64(rcand, plane);
        // the pivot row's entries of my column slice
        double prow[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) prow[j] = tf_bperm_f64(((h << 3) | piv) << 2, val[j]);
        if (g == piv) {
            myk = kk;
#pragma unroll
            for (int j = 0; j < NJ; ++j) val[j] *= rp;
        } else {
            const double f2 = -mult * rp;
#pragma unroll
            for (int j = 0; j < NJ; ++j) val[j] = tf_fma(f2, prow[j], val[j]);
        }
    }
    // a zero or non-finite pivot leaves infinities / NaNs behind: one test at the end
    double chk = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) chk += val[j] - val[j];           // 0 for finite values, else NaN
    if (g < BB && !(chk == 0.0)) ok = false;
    return myk;
}

// ---- Gauss-Jordan in a pivot order known beforehand -----------------------------------------
// What tf_gj_wave pays per pivot step is the search: key, three DPP steps, v_readlane, and only
// then the address of the pivot row -- ~630 cycles of one wavefront per step, 3 800 per block
// (profiles/r02_cr_factor_stamps.txt), for 4 FMAs of work.  The matrices of consecutive
// factorisations (time steps) differ little, so the order the search found last time is tried
// first: every node keeps it (TfLevelArgs::perm, 3 bits per block row, 0 = natural order), the
// lanes load their rows *in that order*, and the elimination runs with pivot k = lane row k:
// every lane index is a compile-time constant (pivot value: v_readlane, pivot row: two
// bank-masked DPP moves per dword inside the 8-lane group, no LDS crossbar round trip for it).
// Partial pivoting bounds the multipliers by 1; here a multiplier above TF_GJ_GROWTH among the
// rows not yet used (threshold pivoting, SuperLU's u = 1/8) -- or anything non-finite -- sends
// the wavefront to the search (tf_gj_wave on the rows in natural order), whose order is stored
// for the next time.
#ifndef TF_GJ_STATIC
#define TF_GJ_STATIC 1
#endif
#define TF_GJ_GROWTH 8.0

// value of lane k (compile-time) of each 8-lane group, in every lane of the group
template <int K>
__device__ __forceinline__ int tf_group8_bcast(int v) {
    constexpr int q = K & 3, qp = q | (q << 2) | (q << 4) | (q << 6);
    // the quad that holds lane k: every lane of it takes the value (banks = quads of a 16-lane row);
    // the other quad of the group mirrors it (row_half_mirror: lane i <- lane 7 - i)
    constexpr int own = K < 4 ? 0x5 : 0xA, other = K < 4 ? 0xA : 0x5;
    // (the first move writes every lane -- the quad without lane k takes a value the mirror
    // replaces -- so that it needs no previous value of its destination: one instruction, no zeroing)
    (void)own;
    int t = __builtin_amdgcn_update_dpp(0, v, qp, 0xF, 0xF, true);
    return __builtin_amdgcn_update_dpp(t, t, 0x141, 0xF, other, false);
}
template <int K>
__device__ __forceinline__ double tf_group8_bcast_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = tf_group8_bcast<K>((int)(unsigned)(b & 0xffffffffll));
    const int hi = tf_group8_bcast<K>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// OR over the 8 lanes of a group, result in every lane
__device__ __forceinline__ unsigned tf_group8_or(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    return v;
}

template <int BB, int NC, int PC0, int KK>
__device__ __forceinline__ void tf_gj_static_step(double (&val)[(NC + 7) / 8], int g, int h, bool& grow) {
    constexpr int NJ = (NC + 7) / 8;
    constexpr int pc = PC0 + KK, ph = pc & 7, pj = pc >> 3;
    const double pv = tf_readlane_f64(val[pj], (ph << 3) | KK);           // wave-uniform
    const double rp = tf_rcp_newton(pv);
    const double mult = tf_bperm_f64(((ph << 3) | g) << 2, val[pj]);      // my row's entry of the pivot column
    double prow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) prow[j] = tf_group8_bcast_f64<KK>(val[j]);
    const double f = mult * rp;
    if (g > KK && g < BB && !(tf_abs(f) <= TF_GJ_GROWTH)) grow = true;
    if (g == KK) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) val[j] *= rp;
    } else {
        const double f2 = -f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) val[j] = tf_fma(f2, prow[j], val[j]);
    }
    if constexpr (KK + 1 < BB) tf_gj_static_step<BB, NC, PC0, KK + 1>(val, g, h, grow);
}

// Returns false when the order did not hold (growth / non-finite values): the caller repeats
// the block with the search.  Lane row g ends as row g of the result.
template <int BB, int NC, int PC0>
__device__ __forceinline__ bool tf_gj_static(double (&val)[(NC + 7) / 8], int g, int h) {
    constexpr int NJ = (NC + 7) / 8;
    bool grow = false;
    tf_gj_static_step<BB, NC, PC0, 0>(val, g, h, grow);
    double chk = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) chk += val[j] - val[j];
    if (g < BB && !(chk == 0.0)) grow = true;
    return __builtin_amdgcn_ballot_w64(grow) == 0ull;
}

// The block inversion of a node with the stored order first (see above).  `load(r, c)` returns
// entry (r, c) of the augmented block (rows in natural order); returns the row of the result this
// lane's row ended as.  `code` (wave-uniform) is the node's stored order and is updated.
template <int BB, int NC, int PC0, class Load>
__device__ __forceinline__ int tf_gj_node(double (&val)[(NC + 7) / 8], int g, int h, unsigned& code,
                                          bool& ok, Load load, bool* searched = nullptr) {
    constexpr int NJ = (NC + 7) / 8;
#if TF_GJ_STATIC
    {
        // perm[g] = ((code >> 3g) & 7) ^ g: the row that serves pivot g
        const int src = g < BB ? (int)(((code >> (3 * g)) & 7u) ^ (unsigned)g) : 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) val[j] = load(g < BB ? src : -1, h + 8 * j);
        if (tf_gj_static<BB, NC, PC0>(val, g, h)) return g;
    }
#endif
    if (searched) *searched = true;
#pragma unroll
    for (int j = 0; j < NJ; ++j) val[j] = load(g < BB ? g : -1, h + 8 * j);
    const int myk = tf_gj_wave<BB, NC, PC0>(val, g, h, ok);
#if TF_GJ_STATIC
    // the order the search used: pivot myk was served by row g
    unsigned bits = 0;
    if (g < BB && h == 0) {
        if (myk >= 0) bits = ((unsigned)(g ^ myk)) << (3 * myk); else ok = false;
    }
    code = (unsigned)__builtin_amdgcn_readfirstlane((int)tf_group8_or(bits));
#endif
    return myk;
}

#ifndef TF_CR_SCALAR_W
#define TF_CR_SCALAR_W 1           // 0: the wavefront index as every lane computes it (A/B runs)
#endif
template <int BB>
__device__ __forceinline__ void tfk_cr_factor_v3(const TfLevelArgs& a) {
    typedef TfCr2<BB> C;
    constexpr int NPOS = C::NPOS, B2 = BB * BB, REC = 4 * B2, NT_MIN = 256;
    const int NT = blockDim.x, nw = NT >> 6;       // 8 wavefronts per chunk, or 4 (levels with many chunks)
    constexpr int RS = C::RS, PS = C::PS, RW = C::RW;
    constexpr int oL = C::oL, oD = C::oD, oU = C::oU, oY = C::oY;
    constexpr int NC = 4 * BB + 1, NJ = (NC + 7) / 8;       // augmented columns [L | D | U | y | I]
    constexpr int NO = 2 * BB + 1, NOJ = (NO + 7) / 8;      // outputs of one side of a phase-B task
    static_assert(oL == 0 && oD == BB && oU == 2 * BB && oY == 3 * BB, "LDS row order = augmented order");
    const TfLayout& L = a.L;
    const TfCrChunk<BB> ch(L, (int)blockIdx.x);
    // (the wavefront index as a scalar: the task loops of the rounds, the chain positions a task
    // touches and the record addresses are the same for every lane -- scalar registers and scalar
    // branches instead of per-lane arithmetic and exec-mask loops)
    const int tid = threadIdx.x, w = TF_CR_SCALAR_W ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6,
              lane = tid & 63, g = lane & 7, h = lane >> 3;
    const int gq = g < BB ? g : 0;
    const int mI = ch.mI, pe = ch.pe, len = ch.len;
    const bool with_rhs = a.cr_rhs != 0;

    __shared__ double sRow[NPOS * PS];
    __shared__ unsigned sPerm[NPOS + 1];           // stored pivot order of the chunk's nodes (tf_gj_node)
    auto row = [&](int pos, int r) { return sRow + pos * PS + r * RS; };
    unsigned* const perm = a.perm + ch.nbase + ch.start;      // [node]; the top block's follows the nodes
    if (tid < len) sPerm[tid + 1] = perm[tid];
    unsigned* const perm_top = a.perm + (int64_t)L.nsys * L.N + ch.e;    // (fold_top: P == 1, one per system)
    if (a.fold_top && tid == 64) sPerm[0] = *perm_top;

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
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * NT + tid;
            const int nd = i / (3 * B2), rem = i - nd * 3 * B2, blk = rem / B2, rc = rem - blk * B2;
            const int r = rc / BB, c = rc - r * BB;
            if (i < n3) row(nd + 1, r)[blk * BB + c] = v[it] + v2[it];
        }
        if (tid < 3 * B2) {
            const int blk = tid / B2, rc = tid - blk * B2, r = rc / BB, c = rc - r * BB;
            row(0, r)[blk * BB + c] = 0.0;
        }
        if (tid < (len + 1) * BB) row(tid / BB, tid % BB)[oY] = 0.0;
        __syncthreads();
        if (tid < B2) row(0, tid / BB)[oU + tid % BB] = p0;
        if (tid < len * BB) row(tid / BB + 1, tid % BB)[oY] = y0;
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
        // ---- phase A: wavefront w inverts node k = s * (2w + 1)
        const int nA = ((mI >> r) + 1) >> 1;         // <= 8
        for (int t = w; t < nA; t += nw) {
            const int k = s * (2 * t + 1);
            double val[NJ];
            unsigned code = sPerm[k];
            const unsigned code0 = code;
            bool searched = false;
            const double* rk0 = row(k, 0);
            if (r == 1) TF_STAMP(a, 44);
            const int myk = tf_gj_node<BB, NC, oD>(val, g, h, code, ok, [&](int rr, int c) {
                return rr < 0 ? 0.0 : (c < RW ? rk0[rr * RS + c] : ((c < NC && c - RW == rr) ? 1.0 : 0.0)); }, &searched);
            if (r == 1) TF_STAMP(a, 45);
            if (code != code0 && lane == 0) perm[k - 1] = code;
            TF_COUNT(a, 41);
            if (searched) TF_COUNT(a, 40);
            // my row is row myk of [E | . | F | z | D^-1]
            // (the entries' places in the stored record worked out once per lane, as predicated stores,
            // measured no faster and cost five registers: profiles/r03_ab_runs.txt)
            if (g < BB) {
                double* dst = row(k, myk);
                double* rec = a.crf + (ch.nbase + ch.node(k)) * 5 * B2 + myk * BB;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int c = h + 8 * j;
                    if (c < RW) dst[c] = val[j];                     // (the D slot is dead from here on)
                    if (c < BB) rec[1 * B2 + c] = val[j];
                    else if (c >= oU && c < oY) rec[2 * B2 + c - oU] = val[j];
                    else if (c == oY) { if (with_rhs) a.zt[(ch.nbase + ch.node(k)) * BB + myk] = val[j]; }
                    else if (c > oY && c < NC) rec[c - RW] = val[j];
                }
            }
            if (r == 1) TF_STAMP(a, 46);
        }
        __syncthreads();
        TF_STAMP(a, stamp_i); ++stamp_i;
        // ---- phase B: wavefront w updates the neighbours of task t.  Interior a = 2s(t+1):
        //      its L side lost kL = a-s, its U side loses kR = a+s (if there).  Last task: the
        //      L side of the own separator (pe) and the U side of position 0.
        const int nB = mI >> (r + 1);                // <= 7
        for (int t = w; t <= nB; t += nw) {
            const bool ends = t == nB;
            const int aa = 2 * s * (t + 1), nq = mI >> r;
            const int aL = ends ? pe : aa, aU = ends ? 0 : aa;
            const bool vL = ends ? (nq & 1) != 0 : true;
            const bool vR = ends ? true : aa + s <= mI;
            const int kL = ends ? nq * s : aa - s, kR = ends ? s : aa + s;
            // Row aL loses kL through its L block, row aU loses kR through its U block.  Lane
            // (g, h) owns column h of row g of every output block:
            //   j = 0: L' = -L E_kL   (h == b: y_aL -= L z_kL)     j = 2: D_aL -= L F_kL
            //   j = 1: U' = -U F_kR   (h == b: y_aU -= U z_kR)     j = 3: D_aU -= U E_kR
            // so the two updates of one D block (interior tasks: aL == aU) meet in one lane and
            // are applied in the order of the other versions.  (b == 8: the y updates are j = 4.)
            double* raL = row(aL, gq);
            double* raU = row(aU, gq);
            const int kLs = vL ? kL : 1, kRs = vR ? kR : 1;
            constexpr bool YSEP = BB >= 8;
            const int hc = h < BB ? h : 0;
            const bool isy = !YSEP && h == BB;
            const double oldDL = raL[oD + hc], oldDU = raU[oD + hc];
            const double oldYL = raL[oY], oldYU = raU[oY];
            // stored for the solves: the blocks used in this elimination (Lb of kL, Ua of kR)
            if (g < BB && h < BB) {
                if (vL) a.crf[(ch.nbase + ch.node(kL)) * 5 * B2 + 4 * B2 + g * BB + h] = raL[oL + h];
                if (vR) a.crf[(ch.nbase + ch.node(kR)) * 5 * B2 + 3 * B2 + g * BB + h] = raU[oU + h];
            }
            if (r == 1) TF_STAMP(a, 47);
            const int s0 = isy ? oY : oL + hc, s1 = isy ? oY : oU + hc;      // source columns of j = 0, 1
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, y0 = 0.0, y1 = 0.0;
            // (the row's coupling entries are read where they are used, not kept: registers)
#pragma unroll
            for (int m = 0; m < BB; ++m) {
                const double* eL = row(kLs, m);
                const double* eR = row(kRs, m);
                const double lm = -raL[oL + m], um = -raU[oU + m];
                a0 = tf_fma(lm, eL[s0], a0);
                a1 = tf_fma(um, eR[s1], a1);
                a2 = tf_fma(lm, eL[oU + hc], a2);
                a3 = tf_fma(um, eR[oL + hc], a3);
                if (YSEP) { y0 = tf_fma(lm, eL[oY], y0); y1 = tf_fma(um, eR[oY], y1); }
            }
            if (r == 1) TF_STAMP(a, 48);
            if (g < BB) {
                if (h < BB) {
                    if (vL) raL[oL + h] = a0;
                    if (vR) raU[oU + h] = a1;
                    if (aL == aU) {
                        double d = oldDL;
                        if (vL) d += a2;
                        if (vR) d += a3;
                        raL[oD + h] = d;
                    } else {
                        if (vL) raL[oD + h] = oldDL + a2;
                        if (vR) raU[oD + h] = oldDU + a3;
                    }
                }
                const bool ylane = YSEP ? h == 0 : h == BB;
                if (ylane) {
                    const double yl = YSEP ? y0 : a0, yu = YSEP ? y1 : a1;
                    if (aL == aU) {
                        double d = oldYL;
                        if (vL) d += yl;
                        if (vR) d += yu;
                        raL[oY] = d;
                    } else {
                        if (vL) raL[oY] = oldYL + yl;
                        if (vR) raU[oY] = oldYU + yu;
                    }
                }
            }
        }
        __syncthreads();
        TF_STAMP(a, stamp_i); ++stamp_i;
    }

    // ---- this chunk's share of the next level's rows: node p gets (L, D, y) of position pe,
    //      node p-1 gets (U, second part of D, second part of y) of position 0
    for (int i = tid; i < 2 * BB * NO; i += NT) {           // (b = 8: 272 entries, more than 4 wavefronts)
        const int side = i / (BB * NO), rem = i - side * BB * NO, r = rem / NO, o = rem - r * NO;
        const int nn = side == 0 ? ch.p : ch.pprev;
        double* rec = a.Anext + ((int64_t)ch.e * a.Lnext.N + nn) * REC;
        double* rr = a.rhsnext + ((int64_t)ch.e * a.Lnext.N + nn) * 2 * BB;
        const double* rs = row(side == 0 ? pe : 0, r);
        if (o < BB) rec[(side == 0 ? 0 : 2) * B2 + r * BB + o] = rs[(side == 0 ? oL : oU) + o];
        else if (o < 2 * BB) rec[(side == 0 ? 1 : 3) * B2 + r * BB + o - BB] = rs[oD + o - BB];
        else if (with_rhs) rr[(side == 0 ? 0 : BB) + r] = rs[oY];
    }
    TF_STAMP(a, 20);
    if (a.fold_top) {
        // one chunk per system: what is left of rows 0 and pe couples the separator to
        // itself only (TfTopArgs): invert their sum here, and solve for the first rhs
        __syncthreads();
        constexpr int NCT = 2 * BB + 1, NJT = (NCT + 7) / 8;     // [S | y | I]
        if (w == 0) {
            const double* r0 = row(0, 0);
            const double* rp = row(pe, 0);
            double val[NJT];
            unsigned code = sPerm[0];
            const unsigned code0 = code;
            const int myk = tf_gj_node<BB, NCT, 0>(val, g, h, code, ok, [&](int rr, int c) {
                const int o = rr * RS;
                return rr < 0 ? 0.0 : (c < BB ? rp[o + oL + c] + rp[o + oD + c] + r0[o + oU + c] + r0[o + oD + c]
                       : (c == BB ? rp[o + oY] + r0[o + oY] : ((c < NCT && c - BB - 1 == rr) ? 1.0 : 0.0))); });
            if (code != code0 && lane == 0) *perm_top = code;
            if (g < BB) {
                const int nsys = L.Ptot;             // P == 1
#pragma unroll
                for (int j = 0; j < NJT; ++j) {
                    const int c = h + 8 * j;
                    if (c > BB && c < NCT) a.topAinv[(int64_t)(myk * BB + c - BB - 1) * nsys + ch.e] = val[j];
                    if (c == BB && with_rhs) {
                        // ... and the solution of the top block: the y slots take the solution
                        const double x = val[j];
                        a.topx[(int64_t)ch.e * BB + myk] = x;
                        a.x[(ch.nbase + ch.node(pe)) * BB + myk] = x;
                        row(pe, myk)[oY] = x;
                        row(0, myk)[oY] = ch.has_prev ? x : 0.0;
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
                    double xk = rk[oY];
#pragma unroll
                    for (int m = 0; m < BB; ++m) {
                        xk = tf_fma(-rk[oL + m], row(kl, m)[oY], xk);
                        xk = tf_fma(-rk[oU + m], row(kr, m)[oY], xk);
                    }
                    rk[oY] = xk;
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
