// Block inversion by one wavefront with 8 lanes per block row (3 <= b <= 8): the pivot SEARCH of the
// cyclic-reduction factorisation and the inversion of the last level's top block.
//
// Round 2 / 3 ran the whole factorisation of a reduced-level chunk in this lane layout (lane (g, h) =
// (block row, column slice) of the augmented block [L | D | U | y | I] of the node that goes; the pivot
// row found with three DPP max steps inside an 8-lane group and handed to the other rows through
// ds_bpermute; from round 3 on the pivot order of the last factorisation tried first, tf_gj_node).  Since
// round 4 the factorisation itself is tf_cr3_hip.h (columns in lanes, node and neighbours by one
// wavefront); what is left here is what that kernel calls when a remembered pivot order does not hold
// (tf_gj_wave: partial pivoting on the rows in natural order) and for the single b x b block that is
// left at the top of the last level (tf_gj_node).  HIP only.
#pragma once

// max over the 8 lanes of a group (lanes 8q .. 8q+7 of the wavefront), result in every lane
__device__ __forceinline__ unsigned tf_group8_max(unsigned v) {
    unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);            // quad_perm [2,3,0,1]
    v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);           // row_half_mirror
    return o > v ? o : v;
}

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
    {
        // perm[g] = ((code >> 3g) & 7) ^ g: the row that serves pivot g
        const int src = g < BB ? (int)(((code >> (3 * g)) & 7u) ^ (unsigned)g) : 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) val[j] = load(g < BB ? src : -1, h + 8 * j);
        if (tf_gj_static<BB, NC, PC0>(val, g, h)) return g;
    }
    if (searched) *searched = true;
#pragma unroll
    for (int j = 0; j < NJ; ++j) val[j] = load(g < BB ? g : -1, h + 8 * j);
    const int myk = tf_gj_wave<BB, NC, PC0>(val, g, h, ok);
    // the order the search used: pivot myk was served by row g
    unsigned bits = 0;
    if (g < BB && h == 0) {
        if (myk >= 0) bits = ((unsigned)(g ^ myk)) << (3 * myk); else ok = false;
    }
    code = (unsigned)__builtin_amdgcn_readfirstlane((int)tf_group8_or(bits));
    return myk;
}
