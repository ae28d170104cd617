// Plain-data launch argument blocks shared by the host runtime (tf_solver.h, tf_rt_*.cpp)
// and the per-model kernels (tf_kernels.h).  Every kernel takes exactly one of
// these structs by value.
#pragma once
#include <stdint.h>

#define TF_MAX_FIELDS 16   // dependent variables + help functions
#define TF_MAX_PARS 16
#define TF_MAX_TERMS 8

// Partition-interleaved layout of one solver level.
//
// A system of N nodes is cut into P chunks of consecutive nodes; the first
// `rem` chunks own mbase+1 nodes, the others mbase.  Node i of chunk p of
// system e is stored at element  i * Ptot + (e * P + p)  of a plane, so that
// the 64 lanes of a wavefront, which work on 64 neighbouring chunks, always
// touch 512 contiguous bytes, whatever i is.  One thread then walks *along* a
// chunk with its stencil neighbourhood in registers.
struct TfLayout {
    int nsys;      // independent systems (ensemble members)
    int N;         // nodes per system
    int P;         // chunks per system
    int mbase;     // base chunk length
    int rem;       // chunks [0, rem) are one node longer
    int M;         // rows of a plane = mbase + (rem > 0)
    int Ptot;      // nsys * P = row stride of a plane
    int periodic;
    int64_t plane; // M * Ptot elements
};

struct TfSweepArgs {               // F / F+J stencil sweep, J @ v, A-row build
    TfLayout L;
    const double* fields;          // [nvar] planes: dependent variables
    const double* helpers;         // [nh] planes: help functions
    const double* parvec;          // [npar] planes (only those flagged vector are read)
    const double* parsca;          // [npar][nsys] scalar parameter values
    const double* dx;              // [nsys]
    const double* xcoord;          // 1 plane (only read when the model uses x)
    double* F;                     // [nvar] planes
    double* Jv;                    // [nnz] planes (raw values, reference pattern order)
    int with_j;
    // Rosenbrock stage state evaluated on the fly: fields + (((kc0*kx0) + kc1*kx1) + ...)
    int nterms;
    const double* kx[TF_MAX_TERMS];
    double kc[TF_MAX_TERMS];
    double fscale;                 // F is stored as fscale * F (dt * F: the first stage's right-hand side)
    // tfk_sweep_fj_theta: also the right-hand side of the theta scheme in the same pass,
    //   rhs = dt * (F - (theta * J) @ U) + U        (schemes.py:553-556)
    double* theta_rhs;
    double theta, theta_dt;
    // tfk_sweep_fj_bdf2: right-hand side of the linearly implicit BDF-2 step and the history
    // update in the same pass:  rhs = c0*(U - Uprev) + c1*F (two_step) or c1*F;  Uprev <- U
    // (bdf_prev_out NULL: the history is not copied -- the caller keeps U_n in a state slot of its
    // own, tf_step_bdf2_from)
    double* bdf_rhs;
    const double* bdf_prev;
    double* bdf_prev_out;
    double bdf_c0, bdf_c1;
    int bdf_two_step;
    // tfk_sweep_f_stage_rhs: the right-hand side of Rosenbrock stage i in the pass that evaluates
    // its F:  rhs = cF*(fscale*F(U + sum_j kc_j k_j)) + cA*(J @ (sum_j gc_j k_j)), the operations of
    // tfk_sweep_f_stage followed by tfk_spmv's stage form, in their order (schemes.py:152-160).
    // F itself is not stored; Jv is read (the entries that are not node-independent).
    double* stage_rhs;
    double gc[TF_MAX_TERMS];
    double cF, cA;
};

struct TfSpmvArgs {                // y = scale * J @ v  (clamped/wrapped columns)
    TfLayout L;
    const double* parsca;          // scalar parameters / dx: the node-independent Jacobian entries
    const double* dx;              // are evaluated in the kernel, not read back (TfJUniform)
    const double* Jv;
    const double* v;               // [nvar] planes
    double* y;                     // [nvar] planes
    double scale;
    int absval;                    // 1: y = |scale J| @ |v|  (componentwise backward error)
    // Rosenbrock stage right-hand side in one pass: v = ((vc0*vx0) + vc1*vx1) + ...,
    // y = cF*addF + cA*(J @ v)        (schemes.py:156-160)
    int nterms;
    const double* vx[TF_MAX_TERMS];
    double vc[TF_MAX_TERMS];
    const double* addF;
    double cF, cA;
};

struct TfNormArgs {               // per-variable, per-system norm of (a - b): partial sums
    TfLayout L;
    const double* a;               // [nvar] planes
    const double* b;
    double* partial;               // [nvar*nsys][nblocks]
    int nblocks;
    int ord;                       // 2: sum of squares, 0: max |.|
    // non-NULL: the failure flag and the monitor's worst value ride along behind the partials
    // (partial[nvar*nsys*nblocks], [.. + 1]): one download, one host wait per trial
    const int* status;
    const double* mon;
};

struct TfBerrArgs {               // componentwise backward error of (I - cJ) x = b
    TfLayout L;
    const double* parsca;
    const double* dx;
    const double* Jv;
    const double* x;               // [nvar] planes
    const double* rhs;             // [nvar] planes
    double c;
    double* red;                   // max_i |b - Ax|_i / (|x| + |cJ||x| + |b|)_i
    // Sampled form (the monitor of the Theta / BDF-2 steps), one_node >= 0: a thread looks at ONE node of
    // its chunk, node one_node modulo the chunk's length (grid.y = 1); -1: every node of segment grid.y
    int one_node;
    int reserved_;
    // non-NULL: x holds a new state U+ = xbase + delta and the system is (I - cJ) delta = b, i.e.
    // (I - cJ) U+ = b + (I - cJ) xbase: the error is measured on that form (magnitudes |U+| + |xbase|) --
    // delta itself is not in memory when the back-substitution forms the new state, and U+ - xbase
    // carries the rounding of the sum
    const double* xbase;
};

// tfk_sweep_f_stage_rhs_mon: the stage pass and, as one more row of workgroups, the sampled backward-error
// probe of the solve before it (TfBerrArgs::one_node >= 0)
struct TfStageMonArgs { TfSweepArgs s; TfBerrArgs b; };

struct TfVecArgs {                 // elementwise plane algebra
    int64_t n;                     // elements (nvar * plane)
    int nterms;
    int op;
    double* out;
    const double* base;
    const double* x[TF_MAX_TERMS];
    double c[TF_MAX_TERMS];
    double* red;                   // reduction target (max-norm)
    double c2[TF_MAX_TERMS];       // TF_VEC_SUM_ERR: the coefficients of the error estimate
};

struct TfPermArgs {                // natural order <-> partition-interleaved
    TfLayout L;
    const double* src;
    double* dst;
    int ncomp;                     // components handled (planes / interleave width)
    int mode;
};

struct TfGatherArgs {              // out[t] = J value table entry map[t] = node * nnz + k  (CSC assembly)
    TfLayout L;
    const double* Jv;
    const int* map;
    double* out;
    int64_t n;
    int nnz;
};

struct TfDirichletArgs {
    TfLayout L;
    double* fields;
    int n;
    const int* var;
    const int* node;               // node index inside a system (applied to all systems)
    const double* value;
};

// A few point writes passed by value (tf_poke: node assignments of a Python hook); no
// device arrays, so the launch needs no upload and no synchronisation.
#define TF_POKE_MAX 8
struct TfPokeArgs {
    TfLayout L;
    double* fields;
    int n;
    int var[TF_POKE_MAX];
    int node[TF_POKE_MAX];
    double value[TF_POKE_MAX];
};

// One level of the banded solver.  Level 1 reads its block rows from the
// Jacobian planes (A = I - c J, built on the fly); levels >= 2 read the
// explicit block-tridiagonal reduced system produced by the level below.
struct TfLevelArgs {
    TfLayout L;
    // level 1 matrix source: the Jacobian value planes; entries that do not depend on the node
    // are evaluated from the scalar parameters instead of being read (TfJUniform)
    const double* Jv;
    const double* parsca;
    const double* dx;
    double c;
    // level >= 2 matrix source: [3][b][b] planes (sub, diag, super)
    const double* Ablk;
    // right-hand side in / solution out: [B] planes
    const double* rhs;
    double* x;
    // factor storage (down direction): Ut [MP][B][B], Et [MP][B][B] planes; yt [B] planes
    double* Ut;
    double* Et;
    double* yt;
    // levels >= 2 only: stored pivot inverses [2 directions][b][b] planes and the
    // normalised ahead blocks of the upward walk [b][b] planes (downward: Ut)
    double* Dinv;
    double* Unup;
    // spike tips, SoA over chunks: see tf_kernels.h (TipLayout)
    double* tips_dn;
    double* tips_up;
    // next (coarser) level: matrix / rhs written by the assemble kernels,
    // solution read by the back-substitution
    TfLayout Lnext;
    double* Anext;
    double* rhsnext;
    const double* xnext;
    int* status;                   // != 0 when a pivot block was singular / non-finite
    // cooperative reduced-level LU: also eliminate this many columns (b spike columns,
    // +1 for a right-hand side) in the same walk; 0 = the separate column kernel does it
    int lu_cols;
    // cyclic-reduction levels (tf_coop_hip.h, tfk_cr_*): this level's and the next
    // level's buffers are in natural node order, one record per node:
    //   Ablk [node][4][b][b]  (sub, diag, super, second part of diag)
    //   rhs  [node][2][b]     (two parts, summed on load)      x [node][b]
    // next_aos: the NEXT level is stored that way (level 1 writes / reads it)
    int next_aos;
    int cr_rhs;                    // tfk_cr_factor: also eliminate the right-hand side
    double* crf;                   // [node][5][b][b]: Dinv, E, F, Ua, Lb of the eliminated node
    double* zt;                    // [node][b]: Dinv * y of the eliminated node
    // [node] + [system]: pivot order the block inversion of a node (and of the top block) used last
    // time, 3 bits per block row, 0 = natural order; tried first by the next factorisation (tf_gj_node)
    unsigned* perm;
    // last cyclic-reduction level (one chunk per system): it also inverts / applies the
    // single block that is left (what tfk_top_* do otherwise)
    int fold_top;
    // level 1 only: the spike response E of the down walk is not stored; once the separators are
    // solved, tfk_l1_fwd2 eliminates the right-hand side again with the separator above known, and
    // the back-substitution uses U alone (tfk_l1_backsub_u).  Chosen per solver (TF_RESPIKE_*).
    int respike;
    int twist;                     // ... with the down and the up walk sharing every chunk (tf_twist_h)
    // tfk_l1_fwd2_backsub (both of the above in one launch): rows of a walk's y block in the
    // workgroup's dynamic LDS, [direction][row][b][64 lanes]
    int ylds_rows;
    // level 1 below a cyclic-reduction level: the walks assemble the separator rows themselves
    // (tf_asm_side: no tips in memory, no tfk_l1_asm_* launch)
    int fuse_asm;
    // tfk_l1_fwd2_backsub of the last solve of a time step: the new state leaves instead of x,
    // upd_out = upd_base + upd_c0 x (upd_n == 1) or upd_base + (upd_c0 upd_k0 + upd_c1 x) (upd_n == 2)
    double* upd_out;
    const double* upd_base;
    const double* upd_k0;
    double upd_c0, upd_c1;
    int upd_n;
    double* topAinv;               // [b][b] planes over systems (TfTopArgs::Ainv)
    double* topx;                  // [sys][b]
    // diagnostic builds (-DTF_STAMPS): one workgroup writes s_memtime stamps here (else NULL)
    unsigned long long* stamps;
};

// Grids shorter than one stencil window (N < 2*mp + 1): the ghost cells of compilers.py:257-264 wrap
// or clamp onto nodes that are already in the window, the matrix I - cJ is small and dense.  One
// thread per system assembles it (duplicate columns summed, like csc_matrix() does), factorises it
// with partial pivoting and solves (tfk_tiny_factor / tfk_tiny_solve).
struct TfTinyArgs {
    TfLayout L;                    // one chunk per system (P == 1)
    const double* Jv;
    const double* parsca;
    const double* dx;
    double c;
    double* lu;                    // [nsys][n][n], n = N * nvar
    int* piv;                      // [nsys][n]
    const double* rhs;             // [nvar] planes
    double* x;                     // [nvar] planes
    int* status;
};

struct TfTailArgs {                // tfk_cr_tail: the last two cyclic-reduction levels of a solve in one launch
    TfLevelArgs lv[2];
};

// tfk_s_fwd / tfk_s_bwd: a whole solve of a model with b = mp * nvar <= 2 in two launches.  The plan is
// [level 1 | level 2: cyclic reduction in 256-node chunks | level 3: one chunk per system]; workgroup q
// owns chunk q of level 2, i.e. the level-1 chunks whose separators are that chunk's nodes: it walks
// them, reduces its chunk and hands its share of level 3 on -- the workgroup of a system that arrives
// last (`counter`, one per system, left at zero) solves level 3.  The second launch is the way back.
struct TfScalarArgs {
    TfLevelArgs lv[3];
    unsigned* counter;             // [nsys]
};

struct TfTopArgs {                 // final 1-node system per ensemble member
    int nsys;
    const double* A;               // [3][b][b] planes with Ptot = nsys
    const double* rhs;             // [b] planes
    double* Ainv;                  // [b][b] planes
    double* x;                     // [b] planes
    int* status;
    int aos;                       // 1: A [sys][4][b][b], rhs [sys][2][b], x [sys][b] (cyclic-reduction levels below)
};

// Level-1 spike response stored (E: mp*nvar^2 doubles per node, written by the factorisation and
// read by every back-substitution) or replaced by a second elimination of the right-hand side:
// the second form moves fewer bytes but costs a walk per solve.  It wins where E is big and the
// problem fills the GPU (config 3 x 8 members +9 %, config 5 +10 %, config 3 +1.3 %) and loses
// on scalar models (config 2 -12 %) and small grids (-5 % at 2e5 nodes): profiles/r02_ab_runs.txt.
#define TF_RESPIKE_MODEL(mp, nvar) ((mp) * (nvar) * (nvar) >= 8)
#define TF_RESPIKE_MIN_NODES 750000
// ... in twisted form while one walk direction leaves SIMDs idle (1024 SIMDs x 64 lanes)
#define TF_TWIST_MAX_CHUNKS 65536

// Level-1 factorisation walks by two wavefronts per 64 chunks and direction (tf_kernels.h, ROLE):
// one eliminates the band, the other carries the right-hand sides (the spike columns and the first
// right-hand side of the step) with the pivot blocks the first one publishes in LDS (two slots of
// (1 + mp) nvar^2 doubles per lane).  Each wavefront then fits 256 registers (the one-wavefront
// form of the film model needs 374 and keeps 100 values in AGPRs) and the two share a SIMD.
// Config 3: tfk_l1_factor_rhs 81 -> 75 us, 8 members per GPU 641 -> 600 us; the stiff model's
// slots (51 KB) leave room for three workgroups per CU only and the exchange costs more than the
// registers did: 488 -> 558 us, so blocks above 40 KB of slots keep the one-wavefront form
// (profiles/r03_ab_runs.txt, r3o).  Not for scalar models (rows are exchanged there).  The code
// object says which form it holds: the launch bound of tfk_l1_factor is 128 or 64
// (tfb::kernel_block).
#ifndef TF_L1_SPLIT_LDS
#define TF_L1_SPLIT_LDS (40 * 1024)
#endif
#define TF_L1_SPLIT_MODEL(mp, nvar) ((nvar) >= 2 && 2 * (1 + (mp)) * (nvar) * (nvar) * 64 * 8 <= TF_L1_SPLIT_LDS)

// nodes per thread of tfk_sweep_f_stage_rhs (the other sweeps: TF_SEG of the code object, 4 or 8).
// Two register windows per variable make its ghost rows twice as expensive: 8 nodes per thread
// read 152 MB where 4 read 176 MB (config 3; 43 against 51 us per launch, profiles/r02_ab_runs.txt)
#define TF_STAGE_SEG 8

// nodes per chunk of a cyclic-reduction level (one wavefront: 8 nodes x 8 lanes per round)
#define TF_CR_MAXLEN 16
// ... and of the scalar variant (b <= 2: one thread per node, 256-thread workgroups)
#define TF_CRS_MAXLEN 256
// ... and of a level that is ONE chunk per system (a 1 x 1 block's records fit the LDS twice as long)
#define TF_CRS_TOPLEN(b) ((b) == 1 ? 512 : 256)

// Kernel table: index = launch id used by the runtime, name = entry point in
// the per-model code object (tf_entry_hip.h).
enum TfKernel {
    TFK_SWEEP_F = 0, TFK_SWEEP_FJ, TFK_SPMV, TFK_VEC, TFK_VEC_MAXABS, TFK_PERM, TFK_DIRICHLET,
    TFK_L1_FACTOR, TFK_L1_SOLVE, TFK_L1_ASM_MAT, TFK_L1_ASM_RHS, TFK_L1_BACKSUB,
    TFK_BT_LU, TFK_BT_SPIKE, TFK_BT_RHS, TFK_BT_ASM_MAT, TFK_BT_ASM_RHS, TFK_BT_BACKSUB,
    TFK_TOP_FACTOR, TFK_TOP_SOLVE, TFK_BERR, TFK_DIFFNORM, TFK_L1_FACTOR_RHS, TFK_SWEEP_F_STAGE,
    TFK_CR_FACTOR, TFK_CR_FWD, TFK_CR_BWD, TFK_POKE, TFK_SWEEP_FJ_THETA, TFK_SWEEP_FJ_BDF2, TFK_GATHER,
    TFK_SWEEP_F_STAGE_RHS, TFK_L1_FWD2, TFK_L1_BACKSUB_U, TFK_CR_TAIL, TFK_L1_FWD2_BACKSUB, TFK_TINY_FACTOR, TFK_TINY_SOLVE,
    TFK_S_FWD, TFK_S_BWD, TFK_SWEEP_F_STAGE_RHS_N, TFK_L1_SOLVE_CR, TFK_L1_FWD2_BACKSUB_CR, TFK_SWEEP_F_STAGE_RHS_MON, TFK_COUNT
};
#define TF_KERNEL_NAMES { \
    "tfk_sweep_f", "tfk_sweep_fj", "tfk_spmv", "tfk_vec", "tfk_vec_maxabs", "tfk_perm", "tfk_dirichlet", \
    "tfk_l1_factor", "tfk_l1_solve", "tfk_l1_asm_mat", "tfk_l1_asm_rhs", "tfk_l1_backsub", \
    "tfk_bt_lu", "tfk_bt_spike", "tfk_bt_rhs", "tfk_bt_asm_mat", "tfk_bt_asm_rhs", "tfk_bt_backsub", \
    "tfk_top_factor", "tfk_top_solve", "tfk_berr", "tfk_diffnorm", "tfk_l1_factor_rhs", "tfk_sweep_f_stage", \
    "tfk_cr_factor", "tfk_cr_fwd", "tfk_cr_bwd", "tfk_poke", "tfk_sweep_fj_theta", "tfk_sweep_fj_bdf2", \
    "tfk_gather", "tfk_sweep_f_stage_rhs", "tfk_l1_fwd2", "tfk_l1_backsub_u", "tfk_cr_tail", \
    "tfk_l1_fwd2_backsub", "tfk_tiny_factor", "tfk_tiny_solve", "tfk_s_fwd", "tfk_s_bwd", "tfk_sweep_f_stage_rhs_n", \
    "tfk_l1_solve_cr", "tfk_l1_fwd2_backsub_cr", "tfk_sweep_f_stage_rhs_mon" }
