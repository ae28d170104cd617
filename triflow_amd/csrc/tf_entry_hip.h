// __global__ entry points of a per-model code object (gfx950).  Included last
// by the generated translation unit; names are fixed so that the host runtime
// (tf_solver.h, kernel table TF_KERNEL_NAMES) finds them with
// hipModuleGetFunction.
#pragma once

#include "tf_coop_hip.h"
#include "tf_cr2_hip.h"
#include "tf_cr3_hip.h"

#define TF_GID ((int)(blockIdx.x * blockDim.x + threadIdx.x))

// a.fuse_asm: the walks also assemble their half of the separator rows (tf_asm_side); the 64 halves
// of a workgroup are collected in LDS and leave as whole record halves (2*b*b contiguous doubles).
// Offered where the stage of one wavefront leaves room for four workgroups per CU (b <= 6).
#define TF_ASM_HALF (2 * TF_B2 * TF_B2)
#define TF_ASM_SW (TF_ASM_HALF | 1)                  // odd stride: no bank conflicts
#define TF_FUSE_ASM_OK (TF_ASM_SW * 64 * 8 <= 40 * 1024)
#ifndef TF_L1_SPLIT
#define TF_L1_SPLIT 1              // 0: one wavefront per 64 chunks and direction (A/B runs)
#endif
#define TF_L1_SPLIT_ON (TF_L1_SPLIT && TF_L1_SPLIT_MODEL(TF_MP, TF_NVAR))
#define TF_L1_FACTOR_BLOCK (TF_L1_SPLIT_ON ? 128 : 64)
#define TF_L1_XCH (TF_L1_SPLIT_ON ? 2 * (1 + TF_MP) * TF_NVAR * TF_NVAR * 64 : 1)
#define TF_L1_STAGE (TF_FUSE_ASM_OK ? 64 * TF_ASM_SW : 1)
template <bool WITH_RHS>
__device__ __forceinline__ void tfk_l1_factor_any(const TfLevelArgs& a) {
    // (the exchange slots of a split walk and the stage of the separator rows share the block:
    // the last slot is read before the first row is staged)
    __shared__ double stage[TF_L1_STAGE > TF_L1_XCH ? TF_L1_STAGE : TF_L1_XCH];
    __shared__ int srec[64];
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6, pg = blockIdx.x * 64 + lane;
    const bool fuse = TF_FUSE_ASM_OK && a.fuse_asm;
    double* st = fuse ? stage + lane * TF_ASM_SW : nullptr;
    int rec = -1;
    TF_WGTRACE(a, 0, 0);
    constexpr bool UP_U = TF_RESPIKE_MODEL(TF_MP, TF_NVAR);
    if constexpr (TF_L1_SPLIT_ON) {
        if (role == 0) {
            if (blockIdx.y == 0) tfk_chunk_body<TfRowsL1, +1, true, true, false, false, false, 1>(a, pg, nullptr, st, &rec, stage + lane);
            else tfk_chunk_body<TfRowsL1, -1, true, UP_U, false, false, false, 1>(a, pg, nullptr, st, &rec, stage + lane);
        } else {
            if (blockIdx.y == 0) tfk_rhs_follow_body<TfRowsL1, +1, true, WITH_RHS>(a, pg, stage + lane, st);
            else tfk_rhs_follow_body<TfRowsL1, -1, UP_U, false>(a, pg, stage + lane, st);
        }
    } else {
        if (blockIdx.y == 0) tfk_chunk_body<TfRowsL1, +1, true, true, WITH_RHS>(a, pg, nullptr, st, &rec);
        else tfk_chunk_body<TfRowsL1, -1, true, UP_U, false>(a, pg, nullptr, st, &rec);
    }
    if (fuse) {
        if (role == 0) srec[lane] = rec;
        __syncthreads();
        TF_STAMP_T(a, 16, 0);
        const int side = blockIdx.y == 0 ? 0 : TF_ASM_HALF;       // [sub, dia | sup, second part of dia]
        // (independent iterations, unrolled: the LDS reads of several of them are in flight together --
        // this loop is the tail of every workgroup: 10 600 -> 8 400 cycles)
#pragma unroll 6
        for (int idx = threadIdx.x; idx < 64 * TF_ASM_HALF; idx += TF_L1_FACTOR_BLOCK) {
            const int t = idx / TF_ASM_HALF, off = idx - t * TF_ASM_HALF;
            const int r = srec[t];
            if (r >= 0) a.Anext[(int64_t)r * 2 * TF_ASM_HALF + side + off] = stage[t * TF_ASM_SW + off];
        }
    }
    TF_STAMP_T(a, 17, 0);
    TF_WGTRACE(a, 0, 1);
}

extern "C" {

// ---- stencil sweeps: block (64,1,1), grid (chunks/64, segments) ------------
#define TF_SWEEP_ATTR
__global__ void __launch_bounds__(TF_SWEEP_BLOCK) tfk_sweep_f(TfSweepArgs a) {
    tfk_sweep_body<false>(a, TF_GID, blockIdx.y);
}
// F at a Rosenbrock stage state U + sum_j alpha_ij k_j formed while loading
__global__ void __launch_bounds__(TF_SWEEP_BLOCK) tfk_sweep_f_stage(TfSweepArgs a) {
    tfk_sweep_body<false, true>(a, TF_GID, blockIdx.y);
}
__global__ void __launch_bounds__(TF_SWEEP_BLOCK) tfk_sweep_f_stage_rhs(TfSweepArgs a) {
    // (one stage vector -- stage 1 of every scheme -- without the run-time loop over them)
    if (a.nterms == 1) tfk_sweep_body<false, true, false, false, true, TF_STAGE_SEG, 1>(a, TF_GID, blockIdx.y);
    else tfk_sweep_body<false, true, false, false, true, TF_STAGE_SEG>(a, TF_GID, blockIdx.y);
}
// ... for the later stages of the 3-, 4- and 6-stage schemes: the number of stage vectors as a
// compile-time constant (tfk_sweep_body NTERMS).  A kernel of its own: the five-term path takes
// 200 registers, which the two-stage scheme's pass (one term, 144) should not pay for.
__global__ void __launch_bounds__(TF_SWEEP_BLOCK) tfk_sweep_f_stage_rhs_n(TfSweepArgs a) {
    switch (a.nterms) {
    case 2: tfk_sweep_body<false, true, false, false, true, TF_STAGE_SEG, 2>(a, TF_GID, blockIdx.y); break;
    case 3: tfk_sweep_body<false, true, false, false, true, TF_STAGE_SEG, 3>(a, TF_GID, blockIdx.y); break;
    case 4: tfk_sweep_body<false, true, false, false, true, TF_STAGE_SEG, 4>(a, TF_GID, blockIdx.y); break;
    case 5: tfk_sweep_body<false, true, false, false, true, TF_STAGE_SEG, 5>(a, TF_GID, blockIdx.y); break;
    default: tfk_sweep_body<false, true, false, false, true, TF_STAGE_SEG>(a, TF_GID, blockIdx.y);
    }
}
__global__ void TF_SWEEP_ATTR __launch_bounds__(TF_SWEEP_BLOCK) tfk_sweep_fj(TfSweepArgs a) {
    tfk_sweep_body<true>(a, TF_GID, blockIdx.y);
}
__global__ void TF_SWEEP_ATTR __launch_bounds__(TF_SWEEP_BLOCK) tfk_sweep_fj_theta(TfSweepArgs a) {
    tfk_sweep_body<true, false, true>(a, TF_GID, blockIdx.y);
}
__global__ void TF_SWEEP_ATTR __launch_bounds__(TF_SWEEP_BLOCK) tfk_sweep_fj_bdf2(TfSweepArgs a) {
    tfk_sweep_body<true, false, false, true>(a, TF_GID, blockIdx.y);
}
__global__ void __launch_bounds__(TF_SWEEP_BLOCK) tfk_spmv(TfSpmvArgs a) {
    tfk_spmv_body(a, TF_GID, blockIdx.y);
}
// running maximum kept as the bit pattern of a non-negative double.  The value only grows, so
// a wavefront whose candidate is not above what is already there skips the atomic (tens of
// thousands of same-address atomics otherwise: tfk_berr 72 -> 41 us, profiles/README.md).
__device__ __forceinline__ void tf_raise_max(unsigned long long* red, unsigned long long bits) {
    if (bits > __atomic_load_n(red, __ATOMIC_RELAXED)) atomicMax(red, bits);
}

// ---- plane algebra: grid-stride, 16 B per lane where the planes allow ------
__global__ void __launch_bounds__(256) tfk_vec(TfVecArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = TF_GID; i < a.n; i += stride) tfk_vec_elem(a, i);
}

// max |sum_t c_t x_t| : wavefront shuffle -> LDS -> one atomic per block.  The
// bit pattern of a non-negative double is monotone in its value.
__global__ void __launch_bounds__(256) tfk_vec_maxabs(TfVecArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double m = 0.0;
    for (int64_t i = TF_GID; i < a.n; i += stride) {
        const double v = a.op == TF_VEC_MAXRATIO ? tf_vec_ratio(a, i) : (a.op == TF_VEC_SUM_ERR ? tf_vec_sum_err(a, i) : tf_vec_err(a, i));
        m = (v > m || v != v) ? v : m;            // NaN wins, like np.linalg.norm(inf)
    }
    unsigned long long bits = (unsigned long long)__double_as_longlong(m);
    if (m != m) bits = 0x7ff8000000000000ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(bits, off, 64);
        bits = o > bits ? o : bits;
    }
    __shared__ unsigned long long part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) bits = part[w] > bits ? part[w] : bits;
        tf_raise_max((unsigned long long*)a.red, bits);
    }
}

__device__ __forceinline__ void tf_berr_reduce(const TfBerrArgs& a, double m) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(m);
    if (m != m) bits = 0x7ff8000000000000ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(bits, off, 64);
        bits = o > bits ? o : bits;
    }
    if ((threadIdx.x & 63) == 0) tf_raise_max((unsigned long long*)a.red, bits);
}
__global__ void __launch_bounds__(TF_SWEEP_BLOCK) tfk_berr(TfBerrArgs a) {
    tf_berr_reduce(a, tfk_berr_body(a, TF_GID, blockIdx.y));
}
// The right-hand side of Rosenbrock stage 1 and, as one more row of workgroups (the last blockIdx.y), the
// sampled backward-error probe of the stage-0 solve (tfk_berr's one-node form): both only read what the
// solve left, and the probe's own launch cost a step of config 3 ~10 us for 10 MB of loads.
__global__ void __launch_bounds__(TF_SWEEP_BLOCK) tfk_sweep_f_stage_rhs_mon(TfStageMonArgs a) {
    if (blockIdx.y + 1 == gridDim.y) { tf_berr_reduce(a.b, tfk_berr_body(a.b, TF_GID, 0)); return; }
    tfk_sweep_body<false, true, false, false, true, TF_STAGE_SEG, 1>(a.s, TF_GID, blockIdx.y);      // (the host launches it for one term)
}

// grid (nblocks, nvar*nsys): deterministic tree inside the block, one partial per block
__global__ void __launch_bounds__(256) tfk_diffnorm(TfNormArgs a) {
    double acc = tfk_diffnorm_partial(a, blockIdx.y, blockIdx.x, threadIdx.x, blockDim.x);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(acc, off, 64);
        acc = a.ord == 2 ? acc + o : (o > acc ? o : acc);
    }
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            acc = a.ord == 2 ? acc + part[w] : (part[w] > acc ? part[w] : acc);
        a.partial[(int64_t)blockIdx.y * a.nblocks + blockIdx.x] = acc;
        if (a.status && blockIdx.x == 0 && blockIdx.y == 0) {
            double* tail = a.partial + (int64_t)gridDim.y * a.nblocks;
            unsigned long long bits = (unsigned)*a.status;
            tail[0] = __longlong_as_double((long long)bits);
            tail[1] = *a.mon;
        }
    }
}

__global__ void __launch_bounds__(256) tfk_perm(TfPermArgs a) {
    tfk_perm_elem(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(256) tfk_gather(TfGatherArgs a) {
    tfk_gather_elem(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(64) tfk_dirichlet(TfDirichletArgs a) {
    tfk_dirichlet_elem(a, TF_GID);
}
__global__ void __launch_bounds__(64) tfk_poke(TfPokeArgs a) {
    tfk_poke_elem(a, TF_GID);
}

// ---- banded solver, level 1 (rows from the Jacobian planes) ----------------
// grid.y: 0 = walk down, 1 = walk up (wave-uniform)
#define TF_L1_ATTR
// the two wavefronts of a split factorisation walk share a SIMD: half the registers each
#if TF_L1_SPLIT_ON
#define TF_L1_FACTOR_ATTR __attribute__((amdgpu_waves_per_eu(2)))
#else
#define TF_L1_FACTOR_ATTR
#endif
__global__ void TF_L1_FACTOR_ATTR __launch_bounds__(TF_L1_FACTOR_BLOCK) tfk_l1_factor(TfLevelArgs a) { tfk_l1_factor_any<false>(a); }
// factorisation that also eliminates a first right-hand side (the first solve of a
// time step rides along: no second walk over J for it)
__global__ void TF_L1_FACTOR_ATTR __launch_bounds__(TF_L1_FACTOR_BLOCK) tfk_l1_factor_rhs(TfLevelArgs a) { tfk_l1_factor_any<true>(a); }
__global__ void __launch_bounds__(64) tfk_l1_solve(TfLevelArgs a) {
    TF_WGTRACE(a, 2, 0);
    if (blockIdx.y == 0) tfk_chunk_body<TfRowsL1, +1, false, false, true>(a, TF_GID);
    else tfk_chunk_body<TfRowsL1, -1, false, false, false>(a, TF_GID);
    TF_WGTRACE(a, 2, 1);
}
// Second elimination of a right-hand side, once the separators are solved: the values of the
// separator behind a walk go to the right-hand side of its first rows, the elimination is
// recomputed from J (as in tfk_l1_solve) and leaves y for the back-substitution with U alone.
// Replaces the stored spike response E (mp nvar^2 doubles per node written by the factorisation
// and read by every back-substitution) by the Jacobian planes read once more.  grid.y: 0 = down
// half, 1 = up half of every chunk (tf_twist_h).  Compiled for the models it can pay for
// (TF_RESPIKE_MODEL), chosen per solver by size (tf_args.h).
__global__ void __launch_bounds__(64) tfk_l1_fwd2(TfLevelArgs a) {
    if constexpr (TF_RESPIKE_MODEL(TF_MP, TF_NVAR)) {
        if (blockIdx.y == 0) tfk_chunk_body<TfRowsL1, +1, false, false, true, true>(a, TF_GID);
        else tfk_chunk_body<TfRowsL1, -1, false, false, true, true>(a, TF_GID);
    }
}
// tfk_l1_fwd2 and tfk_l1_backsub_u in one launch (twisted form only): wavefront 0 of a workgroup
// takes the down halves of 64 chunks, wavefront 1 the up halves of the same chunks.  The y of the
// second elimination never goes to memory -- every lane keeps its half in the workgroup's LDS
// ([direction][row][b][lane]: 49 KB for the film model), the barrier hands the rows next to the
// middle to the other direction, and the back-substitution starts from there: 8 bytes per node and
// variable less written, and read, per solve, and one launch less.
__global__ void TF_L1_ATTR __launch_bounds__(128) tfk_l1_fwd2_backsub(TfLevelArgs a) {
    if constexpr (TF_RESPIKE_MODEL(TF_MP, TF_NVAR)) {
        extern __shared__ double tf_dyn_lds[];
        TF_WGTRACE(a, 1, 0);
        const int lane = threadIdx.x & 63, dir = threadIdx.x >> 6, pg = blockIdx.x * 64 + lane;
        double* ydn = tf_dyn_lds + lane;
        double* yup = tf_dyn_lds + (size_t)a.ylds_rows * TF_NVAR * 64 + lane;
        if (dir == 0) tfk_chunk_body<TfRowsL1, +1, false, false, true, true, true>(a, pg, ydn);
        else tfk_chunk_body<TfRowsL1, -1, false, false, true, true, true>(a, pg, yup);
        __syncthreads();
        TF_STAMP(a, 3);
        tfk_backsub_twist_body<TfRowsL1, true>(a, pg, dir, ydn, yup);
        TF_STAMP(a, 8);
        TF_WGTRACE(a, 1, 1);
    }
}
// The next level's rows.  When that level keeps records per node (cyclic reduction), the 64
// rows of a workgroup are collected in LDS and leave as whole records: coalesced stores of
// 3*b*b contiguous doubles per separator instead of 8 bytes per lane and instruction.
// One wavefront per separator NODE (the host launches 64 * mp threads per workgroup: wavefront t
// of a workgroup takes node t of the same 64 separators): a thread's work is a chain of dependent
// loads of tips and rows, and 31 250 separators are only 489 wavefronts.
__global__ void __launch_bounds__(64 * TF_MP) tfk_l1_asm_mat(TfLevelArgs a) {
    constexpr int NREC = 3 * TF_B2 * TF_B2, SW = NREC | 1;           // odd stride: no bank conflicts
    const int lane = threadIdx.x & 63, pg = blockIdx.x * 64 + lane;
    const int tsel = blockDim.x > 64 ? (int)(threadIdx.x >> 6) : -1;
    if constexpr (SW * 64 * 8 <= 64 * 1024) {
        if (a.next_aos) {
            __shared__ double stage[64 * SW];
            tfk_asm_body<TfRowsL1, true>(a, pg, stage + lane * SW, tsel);
            __syncthreads();
            // node (e, p) of the next level is record e * Lnext.N + p = pg (Lnext.N == L.P)
            const int pg0 = blockIdx.x * 64;
            const int nrec = a.L.Ptot - pg0 < 64 ? a.L.Ptot - pg0 : 64;
            double* dst = a.Anext + (int64_t)pg0 * 4 * TF_B2 * TF_B2;
            for (int idx = threadIdx.x; idx < nrec * NREC; idx += blockDim.x) {
                const int t = idx / NREC, off = idx - t * NREC;
                dst[(int64_t)t * 4 * TF_B2 * TF_B2 + off] = stage[t * SW + off];
            }
            return;
        }
    }
    tfk_asm_body<TfRowsL1, true>(a, pg, nullptr, tsel);
}
__global__ void __launch_bounds__(64 * TF_MP) tfk_l1_asm_rhs(TfLevelArgs a) {
    tfk_asm_body<TfRowsL1, false>(a, blockIdx.x * 64 + (threadIdx.x & 63), nullptr,
                                  blockDim.x > 64 ? (int)(threadIdx.x >> 6) : -1);
}
__global__ void TF_L1_ATTR __launch_bounds__(64) tfk_l1_backsub(TfLevelArgs a) { tfk_backsub_body<TfRowsL1, true>(a, TF_GID); }
__global__ void TF_L1_ATTR __launch_bounds__(64) tfk_l1_backsub_u(TfLevelArgs a) {
    if constexpr (TF_RESPIKE_MODEL(TF_MP, TF_NVAR)) tfk_backsub_twist_body<TfRowsL1>(a, TF_GID, (int)blockIdx.y);
}

// ---- banded solver, levels >= 2 (explicit block-tridiagonal rows) -----------
typedef TfRowsBT<TF_B2> TfRowsUp;
// Wave-cooperative for b > 2 (tf_coop_hip.h: TfCoop<b>::G lanes per chunk, the host
// multiplies the thread count accordingly), one thread per chunk otherwise.
// grid.y: 0 = walk down, 1 = walk up; tfk_bt_spike: grid.y = 2 * b (direction, column)
__global__ void __launch_bounds__(64) tfk_bt_lu(TfLevelArgs a) {
    if constexpr (TfCoop<TF_B2>::G > 1) tfk_bt_lu_coop<TF_B2>(a, blockIdx.y == 0 ? +1 : -1);
    else tfk_bt_lu_body<TF_B2>(a, TF_GID, blockIdx.y == 0 ? +1 : -1);
}
__global__ void __launch_bounds__(64) tfk_bt_spike(TfLevelArgs a) {
    if constexpr (TfCoop<TF_B2>::G > 1)
        tfk_bt_col_coop<TF_B2>(a, (blockIdx.y & 1) == 0 ? +1 : -1, blockIdx.y >> 1);
    else tfk_bt_col_body<TF_B2>(a, TF_GID, (blockIdx.y & 1) == 0 ? +1 : -1, blockIdx.y >> 1);
}
__global__ void __launch_bounds__(64) tfk_bt_rhs(TfLevelArgs a) {
    if constexpr (TfCoop<TF_B2>::G > 1) tfk_bt_col_coop<TF_B2>(a, blockIdx.y == 0 ? +1 : -1, TF_B2);
    else tfk_bt_col_body<TF_B2>(a, TF_GID, blockIdx.y == 0 ? +1 : -1, TF_B2);
}
__global__ void __launch_bounds__(64) tfk_bt_asm_mat(TfLevelArgs a) {
    if constexpr (TfCoop<TF_B2>::G > 1) tfk_bt_asm_coop<TF_B2, true>(a);
    else tfk_asm_body<TfRowsUp, true>(a, TF_GID);
}
__global__ void __launch_bounds__(64) tfk_bt_asm_rhs(TfLevelArgs a) {
    if constexpr (TfCoop<TF_B2>::G > 1) tfk_bt_asm_coop<TF_B2, false>(a);
    else tfk_asm_body<TfRowsUp, false>(a, TF_GID);
}
__global__ void __launch_bounds__(64) tfk_bt_backsub(TfLevelArgs a) {
    if constexpr (TfCoop<TF_B2>::G > 1) tfk_bt_backsub_coop<TF_B2>(a);
    else tfk_backsub_body<TfRowsUp>(a, TF_GID);
}
__global__ void __launch_bounds__(64) tfk_top_factor(TfTopArgs a) {
    if constexpr (TfCoop<TF_B2>::G == 8) tfk_top_factor_coop<TF_B2>(a);       // 8 lanes per member
    else tfk_top_body<TF_B2, true>(a, TF_GID);
}
__global__ void __launch_bounds__(64) tfk_top_solve(TfTopArgs a) { tfk_top_body<TF_B2, false>(a, TF_GID); }
// grids shorter than one stencil window: a thread per system (TfTinyArgs)
__global__ void __launch_bounds__(64) tfk_tiny_factor(TfTinyArgs a) { tfk_tiny_factor_body(a, TF_GID); }
__global__ void __launch_bounds__(64) tfk_tiny_solve(TfTinyArgs a) { tfk_tiny_solve_body(a, TF_GID); }

// ---- cyclic-reduction levels (tf_coop_hip.h): 3 <= b <= 8 one wavefront per 16-node chunk
//      (8 lanes per node); b <= 2 one thread per node, 256-node chunks
#define TF_CR_BLOCK (TF_B2 <= 2 ? 256 : 64)
// the factorisation of 3 <= b <= 8: 4 or 8 wavefronts per chunk (tf_cr3_hip.h)
#define TF_CR_FACTOR_BLOCK (TF_B2 <= 2 ? 256 : 512)
#ifndef TF_CR_FACTOR_WAVES
#define TF_CR_FACTOR_WAVES (TF_B2 >= 8 ? 2 : 4)     // wavefronts per SIMD the register allocator makes room for
#endif
__global__ void __attribute__((amdgpu_waves_per_eu(TF_CR_FACTOR_WAVES))) __launch_bounds__(TF_CR_FACTOR_BLOCK)
tfk_cr_factor(TfLevelArgs a) {
    if constexpr (TF_B2 <= 2) tfk_crs_factor<TF_B2, 256>(a, (int)blockIdx.x, (int)threadIdx.x);
    else if constexpr (TF_B2 <= 8) tfk_cr_factor_v4<TF_B2>(a);
}
__global__ void __launch_bounds__(TF_CR_BLOCK) tfk_cr_fwd(TfLevelArgs a) {
    if constexpr (TF_B2 <= 2) tfk_crs_fwd<TF_B2, 256>(a, (int)blockIdx.x, (int)threadIdx.x);
    else if constexpr (TF_B2 <= 8) tfk_cr_fwd_coop<TF_B2>(a);
}
__global__ void __launch_bounds__(TF_CR_BLOCK) tfk_cr_bwd(TfLevelArgs a) {
    if constexpr (TF_B2 <= 2) tfk_crs_bwd<TF_B2, 256>(a, (int)blockIdx.x, (int)threadIdx.x);
    else if constexpr (TF_B2 <= 8) tfk_cr_bwd_coop<TF_B2>(a);
}
// the last two levels of a solve: one workgroup per system (3 <= b <= 8; the host only launches it there)
// ---- 3 <= b <= 6: level 1 and the first cyclic-reduction level of a solve in one launch each way
// (round 4; TfTailArgs: lv[0] = level 1, lv[1] = level 2).  The workgroup idea of tfk_s_fwd / tfk_s_bwd
// for the multi-variable models: workgroup q owns four 16-node chunks of level 2 and the <= 64 level-1
// chunks whose separators are their nodes.  Forward: wavefront 0 walks those chunks down, wavefront 1
// walks up the chunks one further on (the walks that end below the owned separators), both assemble
// their halves of the separators' right-hand sides (a.fuse_asm), and after a barrier each wavefront
// reduces two of the four chunks (tfk_cr_fwd_chunk: the body of tfk_cr_fwd) -- no tfk_cr_fwd launch for
// level 2, no round trip of its right-hand side.  Backward: each wavefront back-substitutes two of the
// chunks (tfk_cr_bwd_run), the separator above the first one is a level-3 node and is written by this
// workgroup too, then the launch goes on as tfk_l1_fwd2_backsub.
__device__ __forceinline__ void tf_own4(const TfLayout& L2, int blk, int& e, int& q0, int& nq, int& c0, int& cnt) {
    const int qps = (L2.P + 3) / 4;                 // workgroups per system
    e = blk / qps;
    q0 = (blk - e * qps) * 4;
    nq = L2.P - q0 < 4 ? L2.P - q0 : 4;
    c0 = tf_start(L2, q0);
    cnt = tf_start(L2, q0 + nq - 1) + tf_len(L2, q0 + nq - 1) - c0;     // <= 64 (chunks of <= 16 nodes)
}
__global__ void TF_L1_ATTR __launch_bounds__(128) tfk_l1_solve_cr(TfTailArgs t) {
    if constexpr (TF_B2 >= 3 && TF_B2 <= 6) {
        const TfLevelArgs& l1 = t.lv[0];
        const TfLevelArgs& l2 = t.lv[1];
        __shared__ TfCrSolveLds<TF_B2> sh[2];
        const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        int e, q0, nq, c0, cnt;
        tf_own4(l2.L, (int)blockIdx.x, e, q0, nq, c0, cnt);
        const int P1 = l1.L.P;
        // this wavefront's chunks of level 2: k = w and k = w + 2.  The rows of the first one's stored
        // reduction are requested before the walks (they do not depend on the right-hand side) ...
        const TfCrChunk<TF_B2> cha(l2.L, e * l2.L.P + q0 + (w < nq ? w : 0));
        const TfCrChunk<TF_B2> chb(l2.L, e * l2.L.P + q0 + (w + 2 < nq ? w + 2 : 0));
        const TfCrIo<TF_B2> ioa(l2, cha), iob(l2, chb);
        TfCrFwdRows<TF_B2> ra, rb;
        TF_STAMP(l2, 57);
        tfk_cr_fwd_load<TF_B2>(cha, lane, ioa, ra);
        if (lane < cnt) {
            if (w == 0) tfk_chunk_body<TfRowsL1, +1, false, false, true>(l1, e * P1 + c0 + lane);
            else {
                int c = c0 + lane + 1;
                if (c == P1) c = l1.L.periodic ? 0 : -1;
                if (c >= 0) tfk_chunk_body<TfRowsL1, -1, false, false, false>(l1, e * P1 + c);
            }
        }
        TF_STAMP(l2, 58);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        TF_STAMP(l2, 59);
        // ... the right-hand side records the walks left, of both chunks, and the second one's rows now:
        // one wait for memory, not one per chunk
        double ya[TF_CR_NYS(TF_B2)], yb[TF_CR_NYS(TF_B2)];
        tfk_cr_fwd_load_ys<TF_B2>(cha, lane, ioa, ya);
        tfk_cr_fwd_load_ys<TF_B2>(chb, lane, iob, yb);
        tfk_cr_fwd_load<TF_B2>(chb, lane, iob, rb);
        if (w < nq) tfk_cr_fwd_chunk<TF_B2, false, true>(l2, cha, lane, sh[w], ioa, nullptr, &ra, ya);
        TF_STAMP(l2, 60);
        if (w + 2 < nq) tfk_cr_fwd_chunk<TF_B2, false, true>(l2, chb, lane, sh[w], iob, nullptr, &rb, yb);
        TF_STAMP(l2, 61);
    }
}
__global__ void TF_L1_ATTR __launch_bounds__(128) tfk_l1_fwd2_backsub_cr(TfTailArgs t) {
    if constexpr (TF_RESPIKE_MODEL(TF_MP, TF_NVAR) && TF_B2 >= 3 && TF_B2 <= 6) {
        extern __shared__ double tf_dyn_lds[];
        const TfLevelArgs& a = t.lv[0];
        const TfLevelArgs& l2 = t.lv[1];
        __shared__ TfCrSolveLds<TF_B2> sh[2];
        const int lane = threadIdx.x & 63, dir = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        int e, q0, nq, c0, cnt;
        tf_own4(l2.L, (int)blockIdx.x, e, q0, nq, c0, cnt);
        // this wavefront's chunks of level 2 (k = dir, dir + 2): everything they read is requested at once
        const TfCrChunk<TF_B2> cha(l2.L, e * l2.L.P + q0 + (dir < nq ? dir : 0));
        const TfCrChunk<TF_B2> chb(l2.L, e * l2.L.P + q0 + (dir + 2 < nq ? dir + 2 : 0));
        const TfCrIo<TF_B2> ioa(l2, cha), iob(l2, chb);
        TfCrBwdRows<TF_B2> ra, rb;
        TF_STAMP(l2, 62);
        tfk_cr_bwd_load<TF_B2>(l2, cha, lane, ioa, ra, true);
        const double xa = tfk_cr_bwd_sep<TF_B2>(lane, ioa);
        tfk_cr_bwd_load<TF_B2>(l2, chb, lane, iob, rb, true);
        const double xb = tfk_cr_bwd_sep<TF_B2>(lane, iob);
        if (dir < nq) {
            // (the separator above the first owned chunk: known from level 3, wanted by the first walk)
            if (dir == 0 && cha.has_prev && lane >= TfCr<TF_B2>::G && lane < TfCr<TF_B2>::G + TF_B2)
                l2.x[(cha.nbase + cha.gprev) * TF_B2 + lane - TfCr<TF_B2>::G] = xa;
            tfk_cr_bwd_run<TF_B2, false, true>(l2, cha, lane, sh[dir], ioa, ra, nullptr, xa);
        }
        if (dir + 2 < nq) tfk_cr_bwd_run<TF_B2, false, true>(l2, chb, lane, sh[dir], iob, rb, nullptr, xb);
        TF_STAMP(l2, 63);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int pg = lane < cnt ? e * a.L.P + c0 + lane : a.L.Ptot;          // (beyond the owned chunks: nothing to do)
        double* ydn = tf_dyn_lds + lane;
        double* yup = tf_dyn_lds + (size_t)a.ylds_rows * TF_NVAR * 64 + lane;
        if (dir == 0) tfk_chunk_body<TfRowsL1, +1, false, false, true, true, true>(a, pg, ydn);
        else tfk_chunk_body<TfRowsL1, -1, false, false, true, true, true>(a, pg, yup);
        __syncthreads();
        tfk_backsub_twist_body<TfRowsL1, true>(a, pg, dir, ydn, yup);
    }
}

// ---- b <= 2: a solve in two launches (TfScalarArgs).  As six launches (walks, two forward levels, two
// backward levels, back-substitution) config 2 spent 41 us in them for 50 MB of traffic: every launch
// is a round trip to memory for what the previous one left and 4-14 us of one wavefront's latency.
// Here the workgroup that owns a level-2 chunk also runs the level-1 walks that feed it (threads
// 0-255: down walks of its <= 256 level-1 chunks, threads 256-511: the up walks that end below its
// separators, i.e. of the chunks one further on), so the level-2 right-hand side never waits for
// another workgroup; the last workgroup of a system to finish solves level 3 (no waiting: the
// others simply end).  tfk_s_bwd: level 2 backwards, then the level-1 back-substitution of the same
// chunks -- the separator above its first chunk is a level-3 node, known before the launch.
__global__ void __launch_bounds__(512) tfk_s_fwd(TfScalarArgs t) {
    if constexpr (TF_B2 <= 2) {
        const TfLevelArgs& l1 = t.lv[0];
        const TfLevelArgs& l2 = t.lv[1];
        const int tid = threadIdx.x;
        const TfCrChunk<TF_B2> ch(l2.L, (int)blockIdx.x);
        const int P1 = l1.L.P;
        // the stored reductions of this level-2 chunk and of level 3's one chunk: requested before the
        // walks (they do not depend on the right-hand side) and put into LDS after them, so that neither
        // the walks nor the levels wait for memory (put before the walks: 1.8 us of every workgroup)
        __shared__ double sF2[(TF_CRS_MAXLEN + 1) * 5 * TF_B2 * TF_B2], sF3[TfCrs<TF_B2>::FSIZE];
        TF_STAMP(l2, 57);
        TF_STAMP_REAL(l2, 30);
        TfCrsStaged<TF_B2, 512, TF_CRS_MAXLEN> rF2;
        TfCrsStaged<TF_B2, 512, TF_CRS_TOPLEN(TF_B2)> rF3;
        tfk_crs_stage_request<TF_B2, 512>(l2, (int)blockIdx.x, tid, rF2);
        tfk_crs_stage_request<TF_B2, 512>(t.lv[2], ch.e, tid, rF3);
        // (thread 0: the inverse of the folded top block, for whichever workgroup solves level 3)
        double topv[TF_B2 * TF_B2];
#pragma unroll
        for (int i = 0; i < TF_B2 * TF_B2; ++i) topv[i] = tid == 0 ? t.lv[2].topAinv[(int64_t)i * t.lv[2].L.Ptot + ch.e] : 0.0;
        TF_STAMP(l2, 58);
        // the walks leave their halves of the level-2 right-hand side in LDS, not in memory: nobody else
        // reads these records, and reading them back took a round trip to the L2 (the walks address them
        // through TfLevelArgs::rhsnext: a generic pointer placed so that this chunk's first node is sRhs[0])
        __shared__ double sRhs[TF_CRS_MAXLEN * 2 * TF_B2];
        TfLevelArgs l1w = l1;
        l1w.rhsnext = (double*)sRhs - ((int64_t)ch.e * l1.Lnext.N + ch.start) * 2 * TF_B2;
        if (tid < 256) {
            if (tid < ch.len) tfk_chunk_body<TfRowsL1, +1, false, false, true>(l1w, ch.e * P1 + ch.start + tid);
        } else if (tid - 256 < ch.len) {
            int c = ch.start + tid - 256 + 1;                       // the chunk whose up walk ends below node c - 1
            if (c == P1) c = l1.L.periodic ? 0 : -1;
            if (c >= 0) tfk_chunk_body<TfRowsL1, -1, false, false, false>(l1w, ch.e * P1 + c);
            else if (TF_B2 > 0) {
                // (clamped, the last chunk: no walk comes up to its separator -- its half is zero)
#pragma unroll
                for (int r = 0; r < TF_B2; ++r) sRhs[(tid - 256) * 2 * TF_B2 + TF_B2 + r] = 0.0;
            }
        }
        tfk_crs_stage_put<TF_B2, 512>(l2, (int)blockIdx.x, tid, rF2, sF2);
        tfk_crs_stage_put<TF_B2, 512>(t.lv[2], ch.e, tid, rF3, sF3);
        TF_STAMP(l2, 59);
        __syncthreads();
        TF_STAMP(l2, 60);
        tfk_crs_fwd<TF_B2, 512, true, true>(l2, (int)blockIdx.x, tid, sF2, sRhs);
        TF_STAMP(l2, 61);
        // ---- the workgroup of this system that arrives last solves level 3.  What changes hands between
        // workgroups is this chunk's share of level 3's right-hand side, 2 b doubles: stored with
        // agent-scope (write-through) stores by one wavefront, which waits for them and then counts
        // itself in; the workgroup whose count came last reads the shares with agent-scope loads, the
        // other wavefronts of it behind the barrier (MI355X_MICROARCH.md, hand-offs without the acquire:
        // one lane per storing workgroup adds to one counter, the last adder learns it from the value
        // returned) -- no write-back of the L2, no invalidate
        __shared__ int last;
        if (tid < 64) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) {
                const unsigned seen = __hip_atomic_fetch_add(t.counter + ch.e, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int is_last = seen == (unsigned)l2.L.P - 1u;
                if (is_last) __hip_atomic_store(t.counter + ch.e, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = is_last;
            }
        }
        __syncthreads();
        TF_STAMP(l2, 62);
        TF_STAMP_IF(l2, 55, last);
        TF_STAMP_REAL_IF(l2, 32, last);
        if (last) tfk_crs_fwd<TF_B2, 512, true, false, true, true>(t.lv[2], ch.e, tid, sF3, nullptr, topv);
        TF_STAMP_IF(l2, 56, last);
        TF_STAMP_REAL_IF(l2, 31, last);
    }
}
__global__ void __launch_bounds__(512) tfk_s_bwd(TfScalarArgs t) {
    if constexpr (TF_B2 <= 2) {
        const TfLevelArgs& l1 = t.lv[0];
        const TfLevelArgs& l2 = t.lv[1];
        const int tid = threadIdx.x;
        const TfCrChunk<TF_B2> ch(l2.L, (int)blockIdx.x);
        TF_STAMP(l2, 50);
        tfk_crs_bwd<TF_B2, 512>(l2, (int)blockIdx.x, tid, true);
        TF_STAMP(l2, 51);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        TF_STAMP(l2, 52);
        if (tid < ch.len) tfk_backsub_body<TfRowsL1, true>(l1, ch.e * l1.L.P + ch.start + tid);
        TF_STAMP(l2, 53);
    }
}
__global__ void __launch_bounds__(64 * TF_CR_TAIL_WAVES) tfk_cr_tail(TfTailArgs t) {
    if constexpr (TF_B2 >= 3 && TF_B2 <= TF_CR_TAIL_MAXB) tfk_cr_tail_coop<TF_B2>(t);
}

}  // extern "C"
