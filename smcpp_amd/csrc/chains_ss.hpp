// Chains on the SEMISEPARABLE structure of the SMC++ transition matrix: O(M) per position instead of O(M^2) per row.
//
// The reference builds T in HJTransition (src/transition.cpp:176-254): below the diagonal T(i,j) depends on the column only,
// above it it is  p_float(i) * exp(R_i) * exp(-R_{j-1}) (1 - exp(-inc_j)), i.e. rank one, and the final mixing
// Phi*(1-beta) + beta/(M+1) adds one constant.  So with generators  d (diagonal), g (lower), c0 (the constant) and the
// recurrence  Phi'(i,j+1) = a_j Phi'(i,j),  Phi'(i,i+1) = b_i  for the upper part,
//   (T^T x)_j = d_j x_j + g_j sum_{i>j} x_i + c0 sum_{i<j} x_i + Z_j,            Z_{j+1} = a_j Z_j + b_j x_j,
//   (T w)_i   = d_i w_i + sum_{j<i} g_j w_j + c0 sum_{j>i} w_j + b_i V_i,        V_i = w_{i+1} + a_{i+1} V_{i+1},
// which are prefix sums over the state index.  One WAVEFRONT owns one chunk, lane l owns NPL consecutive states (M <= 64 NPL),
// the prefix sums are Kogge-Stone scans over the lanes with DPP moves (row_shr 1/2/4/8, row_bcast 15/31), the weighted
// ones with per-lane level multipliers - no LDS exchange, no barrier, no operand matrix at all.  A row of span s is s such
// steps of its key's operator diag(e) T^T (forward) or T diag(e) (backward): exactly the operator whose eigensystem the
// reference takes (transition_bundle.cpp:15-25), so NO eigensystem is needed by the chains; the engine extracts the
// generators from T on every E-step, verifies the reconstruction entry by entry and falls back to the dense kernels
// when it fails (smcpp_set_raw with an arbitrary matrix).
//
// Semantics (hmm.cpp:57-149) as in chains2.hpp: stored alpha is float(normalised vector) floored at 1e-10f, stored beta is the
// vector in a running scale (every consumer is scale free), only the normaliser c of a row is stored.  The chain itself runs
// in fp64; what the reference feeds back into it row by row - the float rounding and the 1e-10 floor of the stored vector,
// and for a span-1 row the float rounding of the operator's diagonal float(e_j T_jj) - is reproduced through the diagonal of
// the (linear) operator; the off-diagonal remainder is 1e-2 x 6e-8 (tests: xisum / gamma sums <= 1e-6 against the compiled
// reference and the C restatement, where the dense kernels reach 3e-6).
// The backward lanes hold the states in REVERSED order so that its sums over j > i are prefix scans as well.
// Chunk-parallel fixed point, skip test, merge exit and certificate: as chains2.hpp, per wavefront.
#pragma once

namespace smcpp_dev {

struct SsArgs {
    int M, Mp, nchunks, pass, K, nlds;   // nlds: emission vectors of key slots < nlds live in LDS, the rest comes from L2
    const Chunk *chunks;        // chunks of the forward chain
    // The backward chain costs more per position (three scans against two) and forgets more slowly, so it gets MORE, SHORTER
    // chunks than the forward chain: own chunk list; `tasks` assigns (direction << 30 | chunk) to every wavefront of the launch,
    // -1 = none
    const Chunk *chunks_b;
    int nchunks_b;
    const int *tasks;
    const int2 *rowdesc;        // [rows, padded] {key slot, span}
    const double *E;            // [K][MS] emission vectors by key slot, state order, zero padded (MS = 64 NPL)
    const float *pi_f;          // [Mp]
    // generators by POSITION (MS doubles each); forward position p = state p, backward position p = state MS-1-p
    const double *f_dc, *f_g, *f_cg, *f_b, *f_a, *f_d;
    const double *b_dc, *b_g, *b_b, *b_a;
    double c0;
    float *alpha;
    double *beta, *cnorm;
    float *ends_f, *used_f;
    double *ends_b, *used_b;
    int *changed_f, *changed_b;
    // [pass]: some chunk of the direction REWROTE its end vector in the pass (ran to its end without merging, or a pass that stores
    // everything).  A pass in which neither direction did leaves every chunk's input exactly what it last ran from, so the pass after
    // it would skip every chunk: convergence is certified without launching that pass (round 5; engine_plans.hpp: run_chains_ss)
    int *endchg_f, *endchg_b;
    float eps_f;
    double eps_b;
    int full_f, full_b;         // 1: every chunk of the direction runs whole from the previous pass's end vectors (no skip
                                // test, no merge exit, the contig's first / last chunk included): the fp64 pass after light passes
    int mode_f, mode_b;         // per direction: 0 = first pass (from pi / uniform, stores), 1 = re-run pass, 2 = light pass (float,
                                // store-free: only the chunk's end vector is produced - history for the passes that follow)
    int halo = 0;               // first pass only: every chunk is entered through its halo (Chunk::h0 / h1) instead of from pi / uniform
    long long *dbg;             // optional [8] (SMCPP_DEBUG_CYCLES): shader-clock / 100 MHz ticks / positions of chunk 1, pass 0
    // HYBRID rows (un-binned data, M <= 64 and the tables fit LDS): a row whose span exceeds hyb_th is ONE eigen-power step
    // x <- P (d~^s o (P^-1 x))  (forward; hmm.cpp:72-78) /  b <- P^-T (d~^s o (P^T b))  (backward; hmm.cpp:104-112) on the
    // eigensystem of its key - two M-long mat-vecs per lane against LDS tables - instead of `span` scan steps; shorter rows
    // keep the scans.  rowdesc.x carries the eigen key in its upper 16 bits.  hyb_th = INT_MAX: no such rows.
    int mixed = 0;              // M <= 64, no save_gamma: every scan of the full / re-run passes in float (ss_fwd_step<1, true>; the suffix sums native)
    int dirsplit = 0;           // hybrid rows, M > 32: a workgroup runs ONE direction and stages only that direction's two tables per eigen key
    int hyb_th = 0x7fffffff, Ke = 0, hot_ek = 0;   // hot_ek: the eigen key with the most rows (its table rows stay in registers)
    // eigen keys whose tables live in LDS (the most frequent ones, as many as fit): nk_lds of them, key_of_slot[slot] = eigen key,
    // slot_of_key[key] = LDS slot or -1 - a COLD key's table rows come from L2 (M > 32 with three or four eigen keys: round 5)
    int nk_lds = 0, key_of_slot[4] = {0, 1, 2, 3}, slot_of_key[4] = {0, 1, 2, 3};
    const double *Pinvrm = nullptr, *Prm = nullptr, *PinvT = nullptr, *PT = nullptr;   // [Ke][Mp][Mp]
    const double *dsc = nullptr;               // [Ke][Mp] scaled eigenvalues d / scale
};

constexpr int SS_KE_MAX = 4;
// rowdesc.x: bits 0-15 key slot, 16-23 eigen key (hybrid rows), bit 30: the row continues the caller's row of the row before it (a long
// row of binned data cut into pieces, engine_manager.hpp: build) - the alpha stored at its START is an interior point of a row of the
// reference's, which neither floors nor stores it: it is stored un-floored (the statistics of the next piece read it)
constexpr int SS_ROW_CONT = 1 << 30;
// (rows are only cut when M > 64, i.e. with two or more states per lane: the one-state-per-lane instantiations keep the constant)
template <int NPL>
__device__ __forceinline__ float ss_floor_of(int descx) { return (NPL >= 2 && (descx & SS_ROW_CONT)) ? 0.f : 1e-10f; }
template <int NPL>
__device__ __forceinline__ float ss_floor_at(const SsArgs &a, long long row) { return NPL >= 2 ? ss_floor_of<NPL>(a.rowdesc[row].x) : 1e-10f; }

template <int CTRL>
__device__ __forceinline__ double dpp0(double v) {      // shifted copy, 0.0 where the source lane does not exist
    const long long x = __builtin_bit_cast(long long, v);
    int lo = (int)(x & 0xffffffffll), hi = (int)(x >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

constexpr int DPP_SHR1 = 0x111, DPP_SHR2 = 0x112, DPP_SHR4 = 0x114, DPP_SHR8 = 0x118, DPP_BC15 = 0x142, DPP_BC31 = 0x143,
              DPP_WSHR1 = 0x138;

// inclusive prefix sum over the 64 lanes; c15 / c31 = 1.0 on the lanes the two broadcast levels feed (rows 1,3 / rows 2,3)
template <typename T>
__device__ __forceinline__ T ss_scan(T v, T c15, T c31) {
    v += dpp0<DPP_SHR1>(v);
    v += dpp0<DPP_SHR2>(v);
    v += dpp0<DPP_SHR4>(v);
    v += dpp0<DPP_SHR8>(v);
    v = __builtin_fma(c15, dpp0<DPP_BC15>(v), v);
    v = __builtin_fma(c31, dpp0<DPP_BC31>(v), v);
    return v;
}
__device__ __forceinline__ float ss_scan_f(float v, float c15, float c31) {
    v += dpp0<DPP_SHR1>(v);
    v += dpp0<DPP_SHR2>(v);
    v += dpp0<DPP_SHR4>(v);
    v += dpp0<DPP_SHR8>(v);
    v = __builtin_fmaf(c15, dpp0<DPP_BC15>(v), v);
    v = __builtin_fmaf(c31, dpp0<DPP_BC31>(v), v);
    return v;
}
// inclusive scan of the recurrence  z_l = A_l z_{l-1} + y_l  with the level multipliers lv[] of ss_levels()
__device__ __forceinline__ double ss_scan_w(double v, const double (&lv)[6]) {
    v = __builtin_fma(lv[0], dpp0<DPP_SHR1>(v), v);
    v = __builtin_fma(lv[1], dpp0<DPP_SHR2>(v), v);
    v = __builtin_fma(lv[2], dpp0<DPP_SHR4>(v), v);
    v = __builtin_fma(lv[3], dpp0<DPP_SHR8>(v), v);
    v = __builtin_fma(lv[4], dpp0<DPP_BC15>(v), v);
    v = __builtin_fma(lv[5], dpp0<DPP_BC31>(v), v);
    return v;
}
// level multipliers of the lane recurrence with per-lane factor A: before level D, m_l = prod of A over the window the lane
// has absorbed so far (lanes l-D+1 .. l, cut at the start of its 16-lane row; after the row levels, the whole row / pair)
__device__ __forceinline__ void ss_levels(double A, int lane, double (&lv)[6]) {
    const int r = lane & 15, row = lane >> 4;
    double m = A, t;
    lv[0] = m; t = dpp0<DPP_SHR1>(m); m = (r >= 1) ? m * t : m;
    lv[1] = m; t = dpp0<DPP_SHR2>(m); m = (r >= 2) ? m * t : m;
    lv[2] = m; t = dpp0<DPP_SHR4>(m); m = (r >= 4) ? m * t : m;
    lv[3] = m; t = dpp0<DPP_SHR8>(m); m = (r >= 8) ? m * t : m;
    lv[4] = (row & 1) ? m : 0.0; t = dpp0<DPP_BC15>(m); m = (row & 1) ? m * t : m;
    lv[5] = (row >= 2) ? m : 0.0;
}

template <int NPL>
struct SsFwdC {
    double dc[NPL], g[NPL], cg[NPL], b[NPL], a[NPL], d[NPL], cumA[NPL], lv[6], c15, c31;
    float lvf[6];
    // all-float scans (ss_fwd_step<1, 2>): diagonal d - g in fp64, the off-diagonal coefficients in float, row masks of the suffix scan
    double adg;
    float bgf, bff, c0f, m1, m2, m3;
};
template <int NPL>
struct SsBwdC {
    double dc[NPL], g[NPL], b[NPL], a[NPL], cumA[NPL], lv[6], c15, c31, c0;
    float c15f, c31f, lvf[6];
    double adg;                                    // (d - c0) - g: the diagonal of the all-float-scan step (ss_bwd_step<1, 2>)
    float gf, bff, c0f, m1, m2, m3;
};

template <int NPL>
__device__ __forceinline__ void ss_load_fwd(const SsArgs &a, int lane, SsFwdC<NPL> &c) {
    double cum = 1.0;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int p = lane * NPL + k;
        c.dc[k] = a.f_dc[p]; c.g[k] = a.f_g[p]; c.cg[k] = a.f_cg[p]; c.b[k] = a.f_b[p]; c.a[k] = a.f_a[p]; c.d[k] = a.f_d[p];
        cum *= c.a[k];
        c.cumA[k] = cum;
    }
    ss_levels(cum, lane, c.lv);
#pragma unroll
    for (int q = 0; q < 6; ++q) c.lvf[q] = (float)c.lv[q];
    const int row = lane >> 4;
    c.c15 = (row & 1) ? 1.0 : 0.0;
    c.c31 = (row >= 2) ? 1.0 : 0.0;
    c.adg = c.d[0] - c.g[0];
    c.bgf = (float)(c.g[0] - a.c0); c.bff = (float)c.b[0]; c.c0f = (float)a.c0;
    c.m1 = row < 1 ? 1.f : 0.f; c.m2 = row < 2 ? 1.f : 0.f; c.m3 = row < 3 ? 1.f : 0.f;
}
template <int NPL>
__device__ __forceinline__ void ss_load_bwd(const SsArgs &a, int lane, SsBwdC<NPL> &c) {
    double cum = 1.0;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int p = lane * NPL + k;
        c.dc[k] = a.b_dc[p]; c.g[k] = a.b_g[p]; c.b[k] = a.b_b[p]; c.a[k] = a.b_a[p];
        cum *= c.a[k];
        c.cumA[k] = cum;
    }
    ss_levels(cum, lane, c.lv);
    const int row = lane >> 4;
    c.c15 = (row & 1) ? 1.0 : 0.0;
    c.c31 = (row >= 2) ? 1.0 : 0.0;
    c.c15f = (float)c.c15; c.c31f = (float)c.c31;
#pragma unroll
    for (int q = 0; q < 6; ++q) c.lvf[q] = (float)c.lv[q];
    c.c0 = a.c0;
    c.adg = c.dc[0] - c.g[0];
    c.gf = (float)c.g[0]; c.bff = (float)c.b[0]; c.c0f = (float)a.c0;
    c.m1 = row < 1 ? 1.f : 0.f; c.m2 = row < 2 ? 1.f : 0.f; c.m3 = row < 3 ? 1.f : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// ALL scans of a stored position in float (MIX; one state per lane; round 5).  What made a scan need fp64 was the DIFFERENCE taken of
// it: the sum over the states ABOVE (forward: g_j sum_{i>j} x_i; backward, reversed lanes: sum_{j<i} g_j w_j) was formed as
// total - inclusive prefix, which cancels.  Formed directly - an inclusive SUFFIX scan over the lanes: row_shl 1/2/4/8 inside the
// 16-lane rows, then the totals of the rows above through three v_readlane and three masked v_fmac - it is a sum of positive
// terms like the weighted scans, good to a float ulp, and a level is ONE fused DPP instruction instead of two DPP moves and an
// fp64 add.  These sums are the OFF-DIAGONAL part of the operator (<= 1e-2 of the row's mass): their float rounding enters a
// position at 1e-2 x 6e-8; the diagonal term and the vector itself stay in fp64.
// The normaliser becomes a float-accurate total.  That is harmless by construction: the row is divided by exactly the number
// that is stored as its normaliser (c~ = sum (1 + delta), delta ~ 1e-7), the stored vector is then scaled by 1 / (1 + delta), the
// next row's total carries the same factor, and every statistic takes the vector and its normaliser together (hmm.cpp:113-138
// are invariant to a per-row scale); the log-likelihood sum log c~ telescopes to the same product (the backward chain has
// always run in such a running scale).
// The blocks are hand scheduled: a DPP read needs two wait states after the VALU write of its source, v_readlane one, a VALU
// read of an SGPR written by v_readlane two (gfx940 family); the compiler counts an asm statement as no wait state at all.
// ---------------------------------------------------------------------------------------------------------------
#define SS_D_ " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// forward: z = weighted prefix scan (level multipliers lv), s = inclusive suffix sum.  On return zp = bg * s + c0 * total + (z of
// the lane below), tot = s of lane 0 (the total).
// H32 (M <= 32: the live states fill two 16-lane rows - lanes 0..31 forward, 32..63 backward): one cross-row level instead of two
// and one row total instead of three.
template <bool H32>
__device__ __forceinline__ void ss_x_scan_fwd(float &z, float &s, float &zp, float &tot, const float (&lv)[6], float m1, float m2,
                                              float m3, float bg, float c0) {
    float t1, t2, t3;
    if (H32) {
        asm volatile("s_nop 1\n"
                     "v_add_f32_dpp %1, %1, %1 row_shl:1" SS_D_ "v_fmac_f32_dpp %0, %0, %5 row_shr:1" SS_D_ "s_nop 0\n"
                     "v_add_f32_dpp %1, %1, %1 row_shl:2" SS_D_ "v_fmac_f32_dpp %0, %0, %6 row_shr:2" SS_D_ "s_nop 0\n"
                     "v_add_f32_dpp %1, %1, %1 row_shl:4" SS_D_ "v_fmac_f32_dpp %0, %0, %7 row_shr:4" SS_D_ "s_nop 0\n"
                     "v_add_f32_dpp %1, %1, %1 row_shl:8" SS_D_ "v_fmac_f32_dpp %0, %0, %8 row_shr:8" SS_D_
                     "v_readlane_b32 %4, %1, 16\n"
                     "s_nop 0\n"
                     "v_fmac_f32_dpp %0, %0, %9 row_bcast:15" SS_D_
                     "v_fmac_f32_e32 %1, %4, %10\n"
                     "s_nop 0\n"
                     "v_readlane_b32 %3, %1, 0\n"
                     "v_mov_b32_dpp %2, %0 wave_shr:1" SS_D_
                     "v_fmac_f32_e32 %2, %11, %1\n"
                     "v_fmac_f32_e32 %2, %3, %12\n"
                     : "+v"(z), "+v"(s), "=&v"(zp), "=&s"(tot), "=&s"(t1)
                     : "v"(lv[0]), "v"(lv[1]), "v"(lv[2]), "v"(lv[3]), "v"(lv[4]), "v"(m1), "v"(bg), "v"(c0));
        return;
    }
    asm volatile("s_nop 1\n"
                 "v_add_f32_dpp %1, %1, %1 row_shl:1" SS_D_ "v_fmac_f32_dpp %0, %0, %7 row_shr:1" SS_D_ "s_nop 0\n"
                 "v_add_f32_dpp %1, %1, %1 row_shl:2" SS_D_ "v_fmac_f32_dpp %0, %0, %8 row_shr:2" SS_D_ "s_nop 0\n"
                 "v_add_f32_dpp %1, %1, %1 row_shl:4" SS_D_ "v_fmac_f32_dpp %0, %0, %9 row_shr:4" SS_D_ "s_nop 0\n"
                 "v_add_f32_dpp %1, %1, %1 row_shl:8" SS_D_ "v_fmac_f32_dpp %0, %0, %10 row_shr:8" SS_D_
                 "v_readlane_b32 %4, %1, 16\n"
                 "v_readlane_b32 %5, %1, 32\n"
                 "v_fmac_f32_dpp %0, %0, %11 row_bcast:15" SS_D_
                 "v_readlane_b32 %6, %1, 48\n"
                 "v_fmac_f32_e32 %1, %4, %13\n"
                 "v_fmac_f32_dpp %0, %0, %12 row_bcast:31" SS_D_
                 "v_fmac_f32_e32 %1, %5, %14\n"
                 "v_fmac_f32_e32 %1, %6, %15\n"
                 "v_mov_b32_dpp %2, %0 wave_shr:1" SS_D_
                 "v_readlane_b32 %3, %1, 0\n"
                 "v_fmac_f32_e32 %2, %16, %1\n"
                 "s_nop 0\n"
                 "v_fmac_f32_e32 %2, %3, %17\n"
                 : "+v"(z), "+v"(s), "=&v"(zp), "=&s"(tot), "=&s"(t1), "=&s"(t2), "=&s"(t3)
                 : "v"(lv[0]), "v"(lv[1]), "v"(lv[2]), "v"(lv[3]), "v"(lv[4]), "v"(lv[5]), "v"(m1), "v"(m2), "v"(m3), "v"(bg), "v"(c0));
}
// backward (reversed lanes): v = weighted prefix scan, f = plain prefix sum, gs = inclusive suffix sum of g o w.  On return
// gs += c0 * f + b * (v of the lane below), tot = f of lane 63 (the total).
template <bool H32>
__device__ __forceinline__ void ss_x_scan_bwd(float &v, float &f, float &gs, float &tot, const float (&lv)[6], float c15, float c31,
                                              float m1, float m2, float m3, float b, float c0) {
    float t1, t2, t3, vp;
    // (f is an OUTPUT: its first level reads v before v's own first level rewrites it - no copy of the input)
    if (H32) {
        asm volatile("s_nop 1\n"
                     "v_add_f32_dpp %1, %0, %0 row_shr:1" SS_D_ "v_add_f32_dpp %2, %2, %2 row_shl:1" SS_D_ "v_fmac_f32_dpp %0, %0, %6 row_shr:1" SS_D_
                     "v_add_f32_dpp %1, %1, %1 row_shr:2" SS_D_ "v_add_f32_dpp %2, %2, %2 row_shl:2" SS_D_ "v_fmac_f32_dpp %0, %0, %7 row_shr:2" SS_D_
                     "v_add_f32_dpp %1, %1, %1 row_shr:4" SS_D_ "v_add_f32_dpp %2, %2, %2 row_shl:4" SS_D_ "v_fmac_f32_dpp %0, %0, %8 row_shr:4" SS_D_
                     "v_add_f32_dpp %1, %1, %1 row_shr:8" SS_D_ "v_add_f32_dpp %2, %2, %2 row_shl:8" SS_D_ "v_fmac_f32_dpp %0, %0, %9 row_shr:8" SS_D_
                     "v_readlane_b32 %5, %2, 48\n"
                     "v_fmac_f32_dpp %1, %1, %11 row_bcast:15" SS_D_
                     "v_fmac_f32_dpp %0, %0, %10 row_bcast:15" SS_D_
                     "v_fmac_f32_e32 %2, %5, %12\n"
                     "v_readlane_b32 %3, %1, 63\n"
                     "v_mov_b32_dpp %4, %0 wave_shr:1" SS_D_
                     "v_fmac_f32_e32 %2, %14, %1\n"
                     "v_fmac_f32_e32 %2, %13, %4\n"
                     : "+v"(v), "=&v"(f), "+v"(gs), "=&s"(tot), "=&v"(vp), "=&s"(t3)
                     : "v"(lv[0]), "v"(lv[1]), "v"(lv[2]), "v"(lv[3]), "v"(lv[4]), "v"(c15), "v"(m3), "v"(b), "v"(c0));
        return;
    }
    asm volatile("s_nop 1\n"
                 "v_add_f32_dpp %1, %0, %0 row_shr:1" SS_D_ "v_add_f32_dpp %2, %2, %2 row_shl:1" SS_D_ "v_fmac_f32_dpp %0, %0, %8 row_shr:1" SS_D_
                 "v_add_f32_dpp %1, %1, %1 row_shr:2" SS_D_ "v_add_f32_dpp %2, %2, %2 row_shl:2" SS_D_ "v_fmac_f32_dpp %0, %0, %9 row_shr:2" SS_D_
                 "v_add_f32_dpp %1, %1, %1 row_shr:4" SS_D_ "v_add_f32_dpp %2, %2, %2 row_shl:4" SS_D_ "v_fmac_f32_dpp %0, %0, %10 row_shr:4" SS_D_
                 "v_add_f32_dpp %1, %1, %1 row_shr:8" SS_D_ "v_add_f32_dpp %2, %2, %2 row_shl:8" SS_D_ "v_fmac_f32_dpp %0, %0, %11 row_shr:8" SS_D_
                 "v_readlane_b32 %5, %2, 16\n"
                 "v_fmac_f32_dpp %1, %1, %14 row_bcast:15" SS_D_
                 "v_fmac_f32_dpp %0, %0, %12 row_bcast:15" SS_D_
                 "v_readlane_b32 %6, %2, 32\n"
                 "v_readlane_b32 %7, %2, 48\n"
                 "v_fmac_f32_dpp %1, %1, %15 row_bcast:31" SS_D_
                 "v_fmac_f32_dpp %0, %0, %13 row_bcast:31" SS_D_
                 "v_fmac_f32_e32 %2, %5, %16\n"
                 "v_fmac_f32_e32 %2, %6, %17\n"
                 "v_fmac_f32_e32 %2, %7, %18\n"
                 "v_mov_b32_dpp %4, %0 wave_shr:1" SS_D_
                 "v_readlane_b32 %3, %1, 63\n"
                 "v_fmac_f32_e32 %2, %20, %1\n"
                 "v_fmac_f32_e32 %2, %19, %4\n"
                 : "+v"(v), "=&v"(f), "+v"(gs), "=&s"(tot), "=&v"(vp), "=&s"(t1), "=&s"(t2), "=&s"(t3)
                 : "v"(lv[0]), "v"(lv[1]), "v"(lv[2]), "v"(lv[3]), "v"(lv[4]), "v"(lv[5]), "v"(c15), "v"(c31), "v"(m1), "v"(m2), "v"(m3),
                   "v"(b), "v"(c0));
}
#undef SS_D_

// The scans of one position are written level by level across the independent chains: a DPP move may only read a register two
// instructions after it was written, so one chain alone pays a wait state per level, two or three interleaved pay none.
// one position of the forward chain:  out = e o (T^T x);  S = sum x
// MIX (one state per lane; the full and re-run passes of an E-step without save_gamma): every scan in float, the sums over the
// states above as native suffix scans (ss_x_scan_fwd / ss_x_scan_bwd above); the vector and the diagonal term stay in fp64.
template <int NPL, bool MIX = false, bool H32 = false>
__device__ __forceinline__ void ss_fwd_step(const SsFwdC<NPL> &c, const double (&x)[NPL], const double (&e)[NPL],
                                            double (&out)[NPL], double &S) {
    if (MIX && NPL == 1) {
        // (T^T x)_j = (d_j - g_j) x_j + (g_j - c0) sum_{i >= j} x_i + c0 S + Z_j
        float s = (float)x[0];
        float z = c.bff * s, zp, tot;
        ss_x_scan_fwd<H32>(z, s, zp, tot, c.lvf, c.m1, c.m2, c.m3, c.bgf, c.c0f);
        S = (double)tot;
        out[0] = e[0] * __builtin_fma(c.adg, x[0], (double)zp);
        return;
    }
    double lp[NPL], w[NPL];
    lp[0] = x[0];
    w[0] = c.b[0] * x[0];
#pragma unroll
    for (int k = 1; k < NPL; ++k) {
        lp[k] = lp[k - 1] + x[k];
        w[k] = __builtin_fma(c.a[k], w[k - 1], c.b[k] * x[k]);
    }
    double p_ = lp[NPL - 1], z_ = w[NPL - 1];
    {
        double tp, tz;
        tp = dpp0<DPP_SHR1>(p_); tz = dpp0<DPP_SHR1>(z_); p_ += tp; z_ = __builtin_fma(c.lv[0], tz, z_);
        tp = dpp0<DPP_SHR2>(p_); tz = dpp0<DPP_SHR2>(z_); p_ += tp; z_ = __builtin_fma(c.lv[1], tz, z_);
        tp = dpp0<DPP_SHR4>(p_); tz = dpp0<DPP_SHR4>(z_); p_ += tp; z_ = __builtin_fma(c.lv[2], tz, z_);
        tp = dpp0<DPP_SHR8>(p_); tz = dpp0<DPP_SHR8>(z_); p_ += tp; z_ = __builtin_fma(c.lv[3], tz, z_);
        tp = dpp0<DPP_BC15>(p_); tz = dpp0<DPP_BC15>(z_); p_ = __builtin_fma(c.c15, tp, p_); z_ = __builtin_fma(c.lv[4], tz, z_);
        tp = dpp0<DPP_BC31>(p_); tz = dpp0<DPP_BC31>(z_); p_ = __builtin_fma(c.c31, tp, p_); z_ = __builtin_fma(c.lv[5], tz, z_);
    }
    const double li = p_, LI = z_;
    S = lane_get(li, 63);
    const double LIp = dpp0<DPP_WSHR1>(LI);
    if (NPL == 1) {
        out[0] = e[0] * __builtin_fma(c.dc[0], x[0], __builtin_fma(c.g[0], S, __builtin_fma(c.cg[0], li, LIp)));
        return;
    }
    const double lex = li - lp[NPL - 1];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const double incl = lex + lp[k];
        const double Z = (k == 0) ? LIp : __builtin_fma(c.cumA[k - 1 < 0 ? 0 : k - 1], LIp, w[k - 1 < 0 ? 0 : k - 1]);
        const double t = __builtin_fma(c.dc[k], x[k], __builtin_fma(c.g[k], S, __builtin_fma(c.cg[k], incl, Z)));
        out[k] = e[k] * t;
    }
}

// one position of the backward chain (position p = state MS-1-p):  out = T (e o b);  Sw = sum (e o b) (float accuracy)
template <int NPL, bool MIX = false, bool H32 = false>
__device__ __forceinline__ void ss_bwd_step(const SsBwdC<NPL> &c, const double (&bv)[NPL], const double (&e)[NPL],
                                            double (&out)[NPL], float &Sw) {
    if (MIX && NPL == 1) {
        // (T w)_i = ((d_i - c0) - g_i) w_i + sum_{j <= i} g_j w_j + c0 inclW_i + b_i V_i   (lanes hold the states reversed)
        const double w0 = e[0] * bv[0];
        float v = (float)w0;
        float gs = c.gf * v, f;
        ss_x_scan_bwd<H32>(v, f, gs, Sw, c.lvf, c.c15f, c.c31f, c.m1, c.m2, c.m3, c.bff, c.c0f);
        out[0] = __builtin_fma(c.adg, w0, (double)gs);
        return;
    }
    double w[NPL], lg[NPL], u[NPL];
    float lf[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) w[k] = e[k] * bv[k];
    lg[0] = c.g[0] * w[0];
    u[0] = w[0];
    lf[0] = (float)w[0];
#pragma unroll
    for (int k = 1; k < NPL; ++k) {
        lg[k] = __builtin_fma(c.g[k], w[k], lg[k - 1]);
        u[k] = __builtin_fma(c.a[k], u[k - 1], w[k]);
        lf[k] = lf[k - 1] + (float)w[k];
    }
    double p_ = lg[NPL - 1], z_ = u[NPL - 1];
    float f_ = lf[NPL - 1];
    {
        double tp, tz;
        float tf;
        tp = dpp0<DPP_SHR1>(p_); tz = dpp0<DPP_SHR1>(z_); tf = dpp0<DPP_SHR1>(f_); p_ += tp; z_ = __builtin_fma(c.lv[0], tz, z_); f_ += tf;
        tp = dpp0<DPP_SHR2>(p_); tz = dpp0<DPP_SHR2>(z_); tf = dpp0<DPP_SHR2>(f_); p_ += tp; z_ = __builtin_fma(c.lv[1], tz, z_); f_ += tf;
        tp = dpp0<DPP_SHR4>(p_); tz = dpp0<DPP_SHR4>(z_); tf = dpp0<DPP_SHR4>(f_); p_ += tp; z_ = __builtin_fma(c.lv[2], tz, z_); f_ += tf;
        tp = dpp0<DPP_SHR8>(p_); tz = dpp0<DPP_SHR8>(z_); tf = dpp0<DPP_SHR8>(f_); p_ += tp; z_ = __builtin_fma(c.lv[3], tz, z_); f_ += tf;
        tp = dpp0<DPP_BC15>(p_); tz = dpp0<DPP_BC15>(z_); tf = dpp0<DPP_BC15>(f_);
        p_ = __builtin_fma(c.c15, tp, p_); z_ = __builtin_fma(c.lv[4], tz, z_); f_ = __builtin_fmaf(c.c15f, tf, f_);
        tp = dpp0<DPP_BC31>(p_); tz = dpp0<DPP_BC31>(z_); tf = dpp0<DPP_BC31>(f_);
        p_ = __builtin_fma(c.c31, tp, p_); z_ = __builtin_fma(c.lv[5], tz, z_); f_ = __builtin_fmaf(c.c31f, tf, f_);
    }
    const double lig = p_, LI = z_;
    const float lif = f_;
    const double Gtot = lane_get(lig, 63);
    Sw = lane_get(lif, 63);
    const double LIp = dpp0<DPP_WSHR1>(LI);
    if (NPL == 1) {
        out[0] = __builtin_fma(c.dc[0], w[0], (Gtot - lig) + __builtin_fma(c.c0, (double)lif, c.b[0] * LIp));
        return;
    }
    const double lexg = lig - lg[NPL - 1];
    const float lexf = lif - lf[NPL - 1];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const double inclG = lexg + lg[k];
        const double inclW = (double)(lexf + lf[k]);
        const double V = (k == 0) ? LIp : __builtin_fma(c.cumA[k - 1 < 0 ? 0 : k - 1], LIp, u[k - 1 < 0 ? 0 : k - 1]);
        out[k] = __builtin_fma(c.dc[k], w[k], (Gtot - inclG) + __builtin_fma(c.c0, inclW, c.b[k] * V));
    }
}

// A block of 64 row descriptors has just been requested (one per lane): make the wavefront wait for it HERE, once per 64 rows.
// Left pending, the load is carried around the row loop in registers, and the compiler - which cannot count how many stores the
// rows in between have issued on the path it came by - puts an s_waitcnt vmcnt(0) at the head of EVERY row: with one wavefront
// per SIMD that parks the chain until the alpha / beta row it has just stored is acknowledged by L2 (rocprofv3, round 4:
// SQ_WAIT_ANY = 34 % of the wave cycles of this kernel).
__device__ __forceinline__ void ss_desc_settle(int2 &d) { asm volatile("" : "+v"(d.x), "+v"(d.y)); }
// ... and for the same reason every load issued BEFORE the row loop (generators, start vector, first descriptors) is waited for at
// the loop's door: a constant first used inside the loop, behind a store, costs an s_waitcnt vmcnt(0) per row otherwise.
// (gfx9 encoding of s_waitcnt: vmcnt(0), expcnt and lgkmcnt left at their maxima)
__device__ __forceinline__ void ss_vm_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// A wavefront owns ONE chunk: its descriptor is the same in every lane, but it arrives through per-lane loads (the chunk index comes
// from the thread index), so the compiler treats the row loop's trip count as divergent and wraps every iteration in exec-mask
// bookkeeping (s_and_saveexec / s_andn2 exec / v_cmp against a vector register: a dozen scalar instructions per row).  Moved to
// scalar registers the loop is a plain uniform loop.
__device__ __forceinline__ int ss_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ Chunk ss_uniform_chunk(const Chunk &c) {
    Chunk u;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(c.base & 0xffffffffll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(c.base >> 32));
    u.base = (long long)(((unsigned long long)hi << 32) | lo);
    u.r0 = ss_uni(c.r0); u.r1 = ss_uni(c.r1); u.contig = ss_uni(c.contig); u.first = ss_uni(c.first); u.last = ss_uni(c.last);
    u.pad = ss_uni(c.pad); u.h0 = ss_uni(c.h0); u.h1 = ss_uni(c.h1);
    return u;
}

// emission vector of key slot `slot` at this lane's positions (forward: states lane NPL + k; backward: MS-1-(lane NPL + k))
// ALLLDS: every key slot lives in LDS (K <= nlds) - no global path, and with it no vector-memory destination register the
// compiler would have to guard with waits
template <int NPL, bool BWD, bool ALLLDS = false>
__device__ __forceinline__ void ss_emission(const SsArgs &a, const double *sE, int slot, int lane, double (&e)[NPL]) {
    constexpr int MS = 64 * NPL;
    // `slot` is wavefront-uniform: two separate address spaces, not a flat pointer
    if (ALLLDS || slot < a.nlds) {
        const double *src = sE + (size_t)slot * MS;
#pragma unroll
        for (int k = 0; k < NPL; ++k) e[k] = src[BWD ? MS - 1 - (lane * NPL + k) : lane * NPL + k];
    } else {
        const double *src = a.E + (size_t)slot * MS;
#pragma unroll
        for (int k = 0; k < NPL; ++k) e[k] = src[BWD ? MS - 1 - (lane * NPL + k) : lane * NPL + k];
    }
}

// sum over the two 32-lane halves, result in both (v_permlane32_swap: no LDS round trip)
__device__ __forceinline__ double ss_sum_halves(double v) {
    const long long x = __builtin_bit_cast(long long, v);
    const int lo = (int)(x & 0xffffffffll), hi = (int)(x >> 32);
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double a = __builtin_bit_cast(double, ((long long)(unsigned)r1[0] << 32) | (unsigned)r0[0]);
    const double b = __builtin_bit_cast(double, ((long long)(unsigned)r1[1] << 32) | (unsigned)r0[1]);
    return a + b;
}
// ---- hybrid rows (NPL = 1): eigen-power step of the vector held one state per lane ----
// "State coordinate" of a lane: lp = lane (forward) or 63 - lane (backward).  With Mp <= 32 the 64 lanes are G = 2
// groups of W = 64 / G: lane (g, r) sums the columns [g nb, (g + 1) nb) of table row r, nb = Mp / G, against the vector staged
// in a per-wavefront LDS scratch (one broadcast read per column), and the groups' partial sums are added with two xor-shuffles:
// every lane ends with the full product of row lp mod W.
// exp(x) for x <= 0 (powers of eigenvalues scaled to |d| <= 1): 2^n e^r with n = rint(x log2 e), |r| <= ln 2 / 2, e^r by its
// Taylor polynomial of degree 13 (remainder < 4e-18) - twenty-odd instructions against ~100 of the general routine; 0 below -745
__device__ __forceinline__ double ss_exp_neg(double x) {
    if (!(x > -745.0)) return 0.0;
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(p, (int)n);
}
struct SsEigC { double ld[SS_KE_MAX]; bool neg[SS_KE_MAX]; int slot[SS_KE_MAX]; int r, g, nb, G, tpk; };      // tpk: tables per eigen key in LDS (4, or 2 with `dirsplit`)
__device__ __forceinline__ void ss_load_eig(const SsArgs &a, int lp, SsEigC &c) {
    c.G = a.Mp <= 32 ? 2 : 1;
    c.tpk = a.dirsplit ? 2 : 4;
    const int W = 64 / c.G;
    c.r = lp & (W - 1); c.g = lp / W; c.nb = a.Mp / c.G;
#pragma unroll
    for (int e = 0; e < SS_KE_MAX; ++e) {
        double d = 0.0;
        if (e < a.Ke && c.r < a.M) d = a.dsc[(size_t)e * a.Mp + c.r];
        c.ld[e] = log(fabs(d));                     // -inf for a zero (padded) eigenvalue: its power is 0
        c.neg[e] = d < 0.0;
        c.slot[e] = a.slot_of_key[e];
    }
}
__device__ __forceinline__ double ss_eig_pow(const SsEigC &c, int ek, int span) {
    double l = c.ld[0];
    bool ng = c.neg[0];
#pragma unroll
    for (int e = 1; e < SS_KE_MAX; ++e) if (ek == e) { l = c.ld[e]; ng = c.neg[e]; }
    const double p = ss_exp_neg((double)span * l);
    return (ng && (span & 1)) ? -p : p;
}
// LDS tables: [Ke][4][Mp][Mp + 1]: 0 = Pinv, 1 = P (forward), 2 = P^T, 3 = Pinv^T (backward), row-major, padded rows; behind
// them one [64] scratch vector per wavefront.  `dirsplit` (M > 32: four tables of 33 KB per key do not fit): [Ke][2][..], the pair of
// the ONE direction the workgroup runs - table `which` sits at index which & 1
__device__ __forceinline__ double ss_eig_matvec(const SsEigC &c, const double *tab, int Mp, int slot, int which, double *sx, int lp, double x) {
    sx[lp] = x;
    wave_lds_fence();
    const double *row = tab + ((size_t)(slot * c.tpk + (which & (c.tpk - 1))) * Mp + min(c.r, Mp - 1)) * (Mp + 1) + c.g * c.nb;
    const double *xv = sx + c.g * c.nb;
    double a0 = 0.0, a1 = 0.0;
    for (int b = 0; b < c.nb; b += 4) {
        const double r0 = row[b], r1 = row[b + 1], r2 = row[b + 2], r3 = row[b + 3];
        const double v0 = xv[b], v1 = xv[b + 1], v2 = xv[b + 2], v3 = xv[b + 3];
        a0 = __builtin_fma(r0, v0, a0); a1 = __builtin_fma(r1, v1, a1);
        a0 = __builtin_fma(r2, v2, a0); a1 = __builtin_fma(r3, v3, a1);
    }
    double acc = a0 + a1;
    if (c.G >= 2) acc = ss_sum_halves(acc);
    wave_lds_fence();
    return acc;
}
// ... of a COLD eigen key (no LDS slot): this lane's table row straight from the row-major matrix in global memory (L2), eight
// loads in flight; the vector through the same LDS scratch.  `mat` = the key's Mp x Mp matrix of the product (P^-1, P, P^T, P^-T).
__device__ __forceinline__ double ss_eig_matvec_cold(const SsEigC &c, const double *mat, int Mp, double *sx, int lp, double x) {
    sx[lp] = x;
    wave_lds_fence();
    const double *row = mat + (size_t)min(c.r, Mp - 1) * Mp + c.g * c.nb;
    const double *xv = sx + c.g * c.nb;
    double a0 = 0.0, a1 = 0.0;
    for (int b = 0; b < c.nb; b += 8) {
        double rr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) rr[q] = row[b + q];
#pragma unroll
        for (int q = 0; q < 8; q += 2) { a0 = __builtin_fma(rr[q], xv[b + q], a0); a1 = __builtin_fma(rr[q + 1], xv[b + q + 1], a1); }
    }
    double acc = a0 + a1;
    if (c.G >= 2) acc = ss_sum_halves(acc);
    wave_lds_fence();
    return acc;
}

// The whole eigen-power step  out = M1 (pw o (M0 x))  for Mp <= 32 (two lane groups, NB = Mp / 2 columns each), arranged for
// LATENCY - one wavefront per SIMD has nothing to hide an LDS round trip behind: the table rows of BOTH products are requested
// first, the eigenvalue power (a software exp) is computed while they are on their way, and the only dependent round trips left
// are the two stagings of the vector.
template <int NB>
struct SsHotRows { double a[NB], b[NB]; };          // this lane's pieces of the two table rows of the HOT eigen key (registers)
template <int NB>
__device__ __forceinline__ void ss_eig_rows(const SsEigC &c, const double *tab, int Mp, int ek, int w0, double (&ra)[NB], double (&rb)[NB]) {
    const size_t rowoff = (size_t)min(c.r, Mp - 1) * (Mp + 1) + c.g * NB;
    const double *rowA = tab + (size_t)(ek * c.tpk + (w0 & (c.tpk - 1))) * Mp * (Mp + 1) + rowoff;
    const double *rowB = tab + (size_t)(ek * c.tpk + (w0 & (c.tpk - 1)) + 1) * Mp * (Mp + 1) + rowoff;
#pragma unroll
    for (int b = 0; b < NB; ++b) { ra[b] = rowA[b]; rb[b] = rowB[b]; }
}
template <int NB>
__device__ __forceinline__ double ss_eig_step(const SsEigC &c, const double *tab, int Mp, int ek, int w0, double *sx, int lp,
                                              double xin, int span, const SsHotRows<NB> &hot, int hot_ek) {
    double ra[NB], rb[NB];
    if (ek == hot_ek) {                                   // wave-uniform
#pragma unroll
        for (int b = 0; b < NB; ++b) { ra[b] = hot.a[b]; rb[b] = hot.b[b]; }
    } else ss_eig_rows<NB>(c, tab, Mp, ek, w0, ra, rb);
    sx[lp] = xin;
    const double pw = ss_eig_pow(c, ek, span);
    wave_lds_fence();
    const double *xv = sx + c.g * NB;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int b = 0; b < NB; b += 2) { a0 = __builtin_fma(ra[b], xv[b], a0); a1 = __builtin_fma(ra[b + 1], xv[b + 1], a1); }
    const double u = ss_sum_halves(a0 + a1) * pw;
    wave_lds_fence();
    sx[c.r] = u;                     // (both groups hold the full sum of row r and write the same value)
    wave_lds_fence();
    a0 = 0.0; a1 = 0.0;
#pragma unroll
    for (int b = 0; b < NB; b += 2) { a0 = __builtin_fma(rb[b], xv[b], a0); a1 = __builtin_fma(rb[b + 1], xv[b + 1], a1); }
    const double o = ss_sum_halves(a0 + a1);
    wave_lds_fence();
    return o;
}
// dispatcher: Mp = 32 / 16 take the latency-arranged form (every eigen key has its tables in LDS there), larger Mp the generic
// products - from LDS for the keys that have a slot, from L2 for a cold key
__device__ __forceinline__ double ss_eig_apply(const SsArgs &a, const SsEigC &c, const double *tab, int Mp, int ek, int w0, double *sx, int lp,
                                               double xin, int span, const SsHotRows<16> &hot, int hot_ek) {
    if (Mp == 32) return ss_eig_step<16>(c, tab, Mp, ek, w0, sx, lp, xin, span, hot, hot_ek);
    if (Mp == 16) {
        SsHotRows<8> none;                                // (M <= 16: the rows come from LDS)
        return ss_eig_step<8>(c, tab, Mp, ek, w0, sx, lp, xin, span, none, -1);
    }
    int slot = c.slot[0];
#pragma unroll
    for (int e = 1; e < SS_KE_MAX; ++e) if (ek == e) slot = c.slot[e];
    if (slot >= 0) {                                      // wave-uniform
        double u = ss_eig_matvec(c, tab, Mp, slot, w0, sx, lp, xin);
        u *= ss_eig_pow(c, ek, span);
        return ss_eig_matvec(c, tab, Mp, slot, w0 + 1, sx, lp, lp < 64 / c.G ? u : 0.0);
    }
    const size_t MM = (size_t)Mp * Mp;
    const double *m0 = (w0 == 0 ? a.Pinvrm : a.PT) + (size_t)ek * MM, *m1 = (w0 == 0 ? a.Prm : a.PinvT) + (size_t)ek * MM;
    double u = ss_eig_matvec_cold(c, m0, Mp, sx, lp, xin);
    u *= ss_eig_pow(c, ek, span);
    return ss_eig_matvec_cold(c, m1, Mp, sx, lp, lp < 64 / c.G ? u : 0.0);
}

template <int NPL, bool ALLLDS>
__device__ __forceinline__ void ss_fwd_light_rows(const SsArgs &a, const double *sE, long long base, int first, int last, int lane, float (&x)[NPL]);
template <int NPL, bool ALLLDS>
__device__ __forceinline__ void ss_bwd_light_rows(const SsArgs &a, const double *sE, long long base, int rhi, int rlo, int lane, float (&b)[NPL]);

template <int NPL, bool RERUN, bool HYB, bool ALLLDS, bool MIX = false, bool H32 = false>
__device__ __forceinline__ void ss_forward_wave(const SsArgs &a, const double *sE, int c, int lane) {
    constexpr int MS = 64 * NPL;
    const int M = a.M, Mp = a.Mp, pass = a.pass;
    const Chunk ch = ss_uniform_chunk(a.chunks[c]);
    float *end_cur = a.ends_f + ((size_t)(pass & 1) * a.nchunks + c) * Mp;
    const float *end_prev = a.ends_f + ((size_t)((pass + 1) & 1) * a.nchunks + c) * Mp;
    int st[NPL];
    bool live[NPL], stor[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) { st[k] = lane * NPL + k; live[k] = st[k] < M; stor[k] = st[k] < Mp; }
    if (RERUN && ch.first && !a.full_f) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) if (stor[k]) end_cur[st[k]] = end_prev[st[k]];
        return;
    }
    double x[NPL];
    {
        const float *src = (ch.first || !RERUN) ? a.pi_f : a.ends_f + ((size_t)((pass + 1) & 1) * a.nchunks + (c - 1)) * Mp;
#pragma unroll
        for (int k = 0; k < NPL; ++k) x[k] = live[k] ? (double)src[st[k]] : 0.0;
    }
    // Halo (first pass): the rows h0+1 .. r0 of the neighbour are walked first - h0+1 .. h1 in float, h1+1 .. r0 by the very loop
    // below with its stores off - so that the vector at r0 is the neighbour's end vector to the certificate's tolerance
    const bool halo = !RERUN && !HYB && a.halo && ch.h0 < ch.r0;
    const int rbeg = halo ? ch.h1 : ch.r0;
    const int jst = ch.r0 - rbeg;                  // the iteration that finishes row r0 (0 without a halo)
    if (halo && ch.h0 < ch.h1) {
        float xf[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) xf[k] = (float)x[k];
        ss_fwd_light_rows<NPL, ALLLDS>(a, sE, ch.base, ch.h0, ch.h1, lane, xf);
#pragma unroll
        for (int k = 0; k < NPL; ++k) x[k] = live[k] ? (double)xf[k] : 0.0;
        if (jst == 0) {
            // (no fp64 part: the float halo ends on row r0 itself - its normalised, floored vector is what this chunk starts from)
            double part = 0.0;
#pragma unroll
            for (int k = 0; k < NPL; ++k) part += x[k];
            const double iv = 1.0 / wave_sum_dpp(part);
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const float an = live[k] ? fmaxf((float)(x[k] * iv), ss_floor_at<NPL>(a, ch.base + ch.r0 + 1)) : 0.f;
                x[k] = (double)an;
                if (stor[k]) a.used_f[(size_t)c * Mp + st[k]] = an;
            }
        }
    }
    if (RERUN && !a.full_f) {
        bool diff = false;
#pragma unroll
        for (int k = 0; k < NPL; ++k)
            if (live[k]) {
                const float u = a.used_f[(size_t)c * Mp + st[k]];
                if (!(fabsf((float)x[k] - u) <= a.eps_f * fabsf(u))) diff = true;
            }
        if (!__any(diff)) {
#pragma unroll
            for (int k = 0; k < NPL; ++k) if (stor[k]) end_cur[st[k]] = end_prev[st[k]];
            return;
        }
    }
    if (!halo) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) if (stor[k]) a.used_f[(size_t)c * Mp + st[k]] = (float)x[k];
    }
    if (lane == 0) a.changed_f[pass] = 1;
    if (ch.first) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) if (stor[k]) a.alpha[(size_t)ch.base * Mp + st[k]] = (float)x[k];
        if (lane == 0) a.cnorm[ch.base] = 1.0;
    }
    SsFwdC<NPL> cst;
    ss_load_fwd<NPL>(a, lane, cst);
    const int2 *rd = a.rowdesc + ch.base + rbeg + 1;           // descriptor of iteration j (row ell = rbeg + 1 + j)
    const int nrows = ch.r1 - rbeg;
    int2 dcur = rd[lane], dnxt = rd[64 + lane];
    ss_desc_settle(dcur); ss_desc_settle(dnxt);
    float *arow = a.alpha + (size_t)(ch.base + rbeg) * Mp;     // row ell - 1 of iteration j is arow + j Mp
    double *crow = a.cnorm + ch.base + rbeg;
    // MIX: the row's address advances by Mp floats per iteration (a 64-bit multiply per row otherwise), and the normalisers - float
    // totals there - are collected one per lane and stored 64 rows at a time instead of through a one-lane store per row
    float *ap[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) ap[k] = arow + st[k];
    int cacc = 0;                                  // bit patterns of the normalisers of rows (j & ~63) + lane
    auto cflush = [&](int jhi) {                   // store the collected normalisers of the 64-row block that holds row jhi, rows <= jhi
        const int jj = (jhi & ~63) + lane;
        if (jj <= jhi && jj > jst && jj > 0) crow[jj] = (double)__builtin_bit_cast(float, cacc);
    };
    double e[NPL];
    ss_emission<NPL, false, ALLLDS>(a, sE, __builtin_amdgcn_readlane(dcur.x, 0) & 0xFFFF, lane, e);
    constexpr bool hyb = HYB && NPL == 1;
    const double *tab = sE + (size_t)a.nlds * MS;
    SsEigC ec;
    SsHotRows<16> hot;
    int hot_ek = -1;
    if (hyb) {
        ss_load_eig(a, lane, ec);
        if (Mp == 32) { hot_ek = a.hot_ek; ss_eig_rows<16>(ec, tab, Mp, hot_ek, 0, hot.a, hot.b); }
    }
    double *sxw = const_cast<double *>(tab) + (size_t)a.nk_lds * (a.dirsplit ? 2 : 4) * Mp * (Mp + 1) + (threadIdx.x >> 6) * 64;
    bool merged = false;
    const long long t0c = __builtin_readcyclecounter(), t0r = __builtin_amdgcn_s_memrealtime();
    long long npos = 0;
    ss_vm_drain();
    for (int j = 0; j < nrows; ++j) {
        const int jl = j & 63;
        const int span = __builtin_amdgcn_readlane(dcur.y, jl);
        const int ekr = (__builtin_amdgcn_readlane(dcur.x, jl) >> 16) & 0xFF;
        const float flo = NPL >= 2 ? ss_floor_of<NPL>(__builtin_amdgcn_readlane(dcur.x, jl)) : 1e-10f;   // floor of the vector stored where this row begins
        npos += span;
        // descriptor / emission vector of the next row
        if (jl == 63) { dcur = dnxt; dnxt = rd[j + 65 + lane]; ss_desc_settle(dnxt); }
        if (MIX && jl == 0 && j > 0) cflush(j - 1);
        const int slot_n = __builtin_amdgcn_readlane(dcur.x, (j + 1) & 63) & 0xFFFF;
        double en[NPL];
        ss_emission<NPL, false, ALLLDS>(a, sE, slot_n, lane, en);
        if (hyb && span > a.hyb_th) {
            // ---- hybrid row: finish the previous row exactly as below, then ONE eigen-power step from the STORED vector ----
            const double S = wave_sum_dpp(x[0]);
            const double inv = rcp_f64(S);
            double xin = x[0] * inv;
            if (j > 0) {
                const float an = live[0] ? fmaxf((float)xin, flo) : 0.f;
                if (RERUN && !a.full_f && (j & 15) == 0 && j >= 16) {
                    bool bad = false;
                    if (live[0]) {
                        const float old = arow[(size_t)j * Mp + st[0]];
                        if (!(fabsf(an - old) <= a.eps_f * fabsf(old))) bad = true;
                    }
                    if (!__any(bad)) { merged = true; break; }
                }
                if (stor[0]) arow[(size_t)j * Mp + st[0]] = an;
                if (lane == 0) crow[j] = S;
                xin = (double)an;
            }
            const double xo = ss_eig_apply(a, ec, tab, Mp, ekr, 0, sxw, lane, xin, span, hot, hot_ek);
            x[0] = live[0] ? xo : 0.0;
#pragma unroll
            for (int k = 0; k < NPL; ++k) e[k] = en[k];
            continue;
        }
        // first position of the row: the sum of the incoming vector finishes the PREVIOUS row
        double y[NPL], S;
        ss_fwd_step<NPL, MIX, MIX && H32>(cst, x, e, y, S);
        double inv;
        if (MIX) {
            // S is a float total: its float reciprocal and ONE Newton step give 1 / S to 1e-14 (the stored normaliser is S itself)
            const double r0 = (double)__builtin_amdgcn_rcpf((float)S);
            inv = __builtin_fma(__builtin_fma(-S, r0, 1.0), r0, r0);
        } else inv = rcp_f64(S);
        // The reference continues from the STORED vector: normalised, rounded to float, floored at 1e-10 (hmm.cpp:80-94).  That
        // feedback is not noise for small entries (a state whose alpha falls under the floor after a heterozygous row re-enters
        // the next row with 1e-10 instead: 1e-5 on the statistics of the most recent states), so it is reproduced: the operator
        // is linear, and of the perturbation  fb = stored - exact  only the diagonal part e d fb matters (the rest is
        // fb times the off-diagonal mass of T, <= 1e-2 x 6e-8).
        double fb[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) fb[k] = 0.0;
        if (j > 0) {
            float an[NPL];
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const double xs = x[k] * inv;
                an[k] = live[k] ? fmaxf((float)xs, flo) : 0.f;
                fb[k] = (double)an[k] - xs;
            }
            if (RERUN && !a.full_f && (j & 15) == 0 && j >= 16) {
                bool bad = false;
#pragma unroll
                for (int k = 0; k < NPL; ++k)
                    if (live[k]) {
                        const float old = MIX ? *ap[k] : arow[(size_t)j * Mp + st[k]];
                        if (!(fabsf(an[k] - old) <= a.eps_f * fabsf(old))) bad = true;
                    }
                if (!__any(bad)) { merged = true; if (MIX) cflush(j - 1); break; }
            }
            if (j > jst) {
                if (MIX) {
#pragma unroll
                    for (int k = 0; k < NPL; ++k) if (stor[k]) *ap[k] = an[k];
                    cacc = lane == jl ? __builtin_bit_cast(int, (float)S) : cacc;
                } else {
#pragma unroll
                    for (int k = 0; k < NPL; ++k) if (stor[k]) arow[(size_t)j * Mp + st[k]] = an[k];
                    if (lane == 0) crow[j] = S;
                }
            } else if (j == jst) {
                // (halo only: row r0 belongs to the neighbour; what this chunk starts from is what the certificate compares)
#pragma unroll
                for (int k = 0; k < NPL; ++k) if (stor[k]) a.used_f[(size_t)c * Mp + st[k]] = an[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            double ed = e[k] * cst.d[k];
            // hmm.cpp:85-86 multiplies a span-1 row by float(e_j T_ij): the rounding of the diagonal entry is systematic per state
            if (span == 1) { const double edf = (double)(float)ed; y[k] = __builtin_fma(edf - ed, x[k], y[k]); ed = edf; }
            x[k] = __builtin_fma(ed, fb[k], y[k] * inv);
        }
        {
            // two positions per trip (the DPP moves are convergent operations: the compiler will not unroll this loop itself)
            double S2;
            int t = 1;
            for (; t + 1 < span; t += 2) {
                ss_fwd_step<NPL, MIX, MIX && H32>(cst, x, e, y, S2);
#pragma unroll
                for (int k = 0; k < NPL; ++k) x[k] = y[k];
                ss_fwd_step<NPL, MIX, MIX && H32>(cst, x, e, y, S2);
#pragma unroll
                for (int k = 0; k < NPL; ++k) x[k] = y[k];
            }
            if (t < span) {
                ss_fwd_step<NPL, MIX, MIX && H32>(cst, x, e, y, S2);
#pragma unroll
                for (int k = 0; k < NPL; ++k) x[k] = y[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k) e[k] = en[k];
        if (MIX) {
#pragma unroll
            for (int k = 0; k < NPL; ++k) ap[k] += Mp;
        }
    }
    if (MIX && !merged && nrows > 0) cflush(nrows - 1);
    if (a.dbg && !RERUN && c == 1 && lane == 0) {
        a.dbg[0] = __builtin_readcyclecounter() - t0c; a.dbg[1] = __builtin_amdgcn_s_memrealtime() - t0r;
        a.dbg[2] = npos; a.dbg[3] = nrows;
    }
    if (merged) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) if (stor[k]) end_cur[st[k]] = end_prev[st[k]];
        return;
    }
    {
        double part = 0.0;
#pragma unroll
        for (int k = 0; k < NPL; ++k) part += x[k];
        const double S = wave_sum_dpp(part);
        const double inv = 1.0 / S;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const float an = live[k] ? fmaxf((float)(x[k] * inv), ss_floor_at<NPL>(a, ch.base + ch.r1 + 1)) : 0.f;
            if (stor[k]) { a.alpha[(size_t)(ch.base + ch.r1) * Mp + st[k]] = an; end_cur[st[k]] = an; }
        }
        if (lane == 0) { a.cnorm[ch.base + ch.r1] = S; a.endchg_f[pass] = 1; }
    }
}

template <int NPL, bool RERUN, bool HYB, bool ALLLDS, bool MIX = false, bool H32 = false>
__device__ __forceinline__ void ss_backward_wave(const SsArgs &a, const double *sE, int c, int lane) {
    constexpr int MS = 64 * NPL;
    const int M = a.M, Mp = a.Mp, pass = a.pass;
    const Chunk ch = ss_uniform_chunk(a.chunks_b[c]);
    double *end_cur = a.ends_b + ((size_t)(pass & 1) * a.nchunks_b + c) * Mp;
    const double *end_prev = a.ends_b + ((size_t)((pass + 1) & 1) * a.nchunks_b + c) * Mp;
    int st[NPL];
    bool live[NPL], stor[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) { st[k] = MS - 1 - (lane * NPL + k); live[k] = st[k] < M; stor[k] = st[k] < Mp; }
    if (RERUN && ch.last && !a.full_b) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) if (stor[k]) end_cur[st[k]] = end_prev[st[k]];
        return;
    }
    double b[NPL];
    {
        const bool fresh = ch.last || !RERUN;
        const double *src = a.ends_b + ((size_t)((pass + 1) & 1) * a.nchunks_b + (fresh ? c : c + 1)) * Mp;
#pragma unroll
        for (int k = 0; k < NPL; ++k) b[k] = live[k] ? (fresh ? 1.0 / (double)M : src[st[k]]) : 0.0;
    }
    // Halo (first pass): rows h0 .. r1+1 of the neighbour first - h0 .. h1+1 in float, h1 .. r1+1 by the loop below with its
    // stores off; at row r1 the vector is normalised (as a chunk's end vector is) and becomes what the certificate compares
    const bool halo = !RERUN && !HYB && a.halo && ch.h0 > ch.r1;
    const int rend = halo ? ch.h1 : ch.r1;
    const int jst = rend - ch.r1;
    if (halo && ch.h0 > ch.h1) {
        float bf_[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) bf_[k] = (float)b[k];
        ss_bwd_light_rows<NPL, ALLLDS>(a, sE, ch.base, ch.h0, ch.h1, lane, bf_);
        float pt = 0.f;
#pragma unroll
        for (int k = 0; k < NPL; ++k) pt += live[k] ? bf_[k] : 0.f;
        const float iv = 1.f / wave_sum_dpp(pt);
#pragma unroll
        for (int k = 0; k < NPL; ++k) b[k] = live[k] ? (double)(bf_[k] * iv) : 0.0;
    }
    if (RERUN && !a.full_b) {
        bool diff = false;
#pragma unroll
        for (int k = 0; k < NPL; ++k)
            if (live[k]) {
                const double u = a.used_b[(size_t)c * Mp + st[k]];
                if (!(fabs(b[k] - u) <= a.eps_b * fabs(u))) diff = true;
            }
        if (!__any(diff)) {
#pragma unroll
            for (int k = 0; k < NPL; ++k) if (stor[k]) end_cur[st[k]] = end_prev[st[k]];
            return;
        }
    }
    if (!halo) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) if (stor[k]) a.used_b[(size_t)c * Mp + st[k]] = b[k];
    }
    if (lane == 0) a.changed_b[pass] = 1;
    SsBwdC<NPL> cst;
    ss_load_bwd<NPL>(a, lane, cst);
    const int2 *rd = a.rowdesc + ch.base + rend;                // descriptor of iteration j (row ell = rend - j) is rd[-j]
    const int nrows = rend - ch.r0;
    int2 dcur = rd[-lane], dnxt = rd[-64 - lane];
    ss_desc_settle(dcur); ss_desc_settle(dnxt);
    double *brow = a.beta + (size_t)(ch.base + rend) * Mp;      // row ell of iteration j is brow - j Mp
    double *bp[NPL];                                            // MIX: the same address, stepped by -Mp per iteration
#pragma unroll
    for (int k = 0; k < NPL; ++k) bp[k] = brow + st[k];
    double e[NPL];
    ss_emission<NPL, true, ALLLDS>(a, sE, __builtin_amdgcn_readlane(dcur.x, 0) & 0xFFFF, lane, e);
    constexpr bool hyb = HYB && NPL == 1;
    const double *tab = sE + (size_t)a.nlds * MS;
    SsEigC ec;
    SsHotRows<16> hot;
    int hot_ek = -1;
    if (hyb) {
        ss_load_eig(a, 63 - lane, ec);
        if (Mp == 32) { hot_ek = a.hot_ek; ss_eig_rows<16>(ec, tab, Mp, hot_ek, 2, hot.a, hot.b); }
    }
    double *sxw = const_cast<double *>(tab) + (size_t)a.nk_lds * (a.dirsplit ? 2 : 4) * Mp * (Mp + 1) + (threadIdx.x >> 6) * 64;
    bool merged = false;
    const long long t0c = __builtin_readcyclecounter(), t0r = __builtin_amdgcn_s_memrealtime();
    long long npos = 0;
    ss_vm_drain();
    for (int j = 0; j < nrows; ++j) {
        const int jl = j & 63;
        const int span = __builtin_amdgcn_readlane(dcur.y, jl);
        const int ekr = (__builtin_amdgcn_readlane(dcur.x, jl) >> 16) & 0xFF;
        npos += span;
        if (jl == 63) { dcur = dnxt; dnxt = rd[-(j + 65) - lane]; ss_desc_settle(dnxt); }
        const int slot_n = __builtin_amdgcn_readlane(dcur.x, (j + 1) & 63) & 0xFFFF;
        double en[NPL];
        ss_emission<NPL, true, ALLLDS>(a, sE, slot_n, lane, en);
        // beta[ell] in the running scale (hmm.cpp:142 renormalises; every consumer is invariant to a per-row scale)
        if (RERUN && !a.full_b && (j & 15) == 0 && j >= 16) {
            bool bad = false;
#pragma unroll
            for (int k = 0; k < NPL; ++k)
                if (live[k]) {
                    const double old = MIX ? *bp[k] : brow[-(ptrdiff_t)j * Mp + st[k]];
                    if (!(fabs(b[k] - old) <= a.eps_b * fabs(old))) bad = true;
                }
            if (!__any(bad)) { merged = true; break; }
        }
        if (halo && j == jst) {
            // the halo has reached row r1: normalised like a chunk's end vector - the start vector of this chunk's own rows
            double part = 0.0;
#pragma unroll
            for (int k = 0; k < NPL; ++k) part += live[k] ? b[k] : 0.0;
            const double iv = 1.0 / wave_sum_dpp(part);
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                b[k] = live[k] ? b[k] * iv : 0.0;
                if (stor[k]) a.used_b[(size_t)c * Mp + st[k]] = b[k];
            }
        }
        if (j >= jst) {
#pragma unroll
            for (int k = 0; k < NPL; ++k) if (stor[k]) { if (MIX) *bp[k] = b[k]; else brow[-(ptrdiff_t)j * Mp + st[k]] = b[k]; }
        }
        if (hyb && span > a.hyb_th) {
            // ---- hybrid row: b <- P^-T (d~^s o (P^T b)), renormalised (every consumer of beta is scale free) ----
            const double bin = live[0] ? b[0] : 0.0;
            double bo = ss_eig_apply(a, ec, tab, Mp, ekr, 2, sxw, 63 - lane, bin, span, hot, hot_ek);
            bo = live[0] ? bo : 0.0;
            b[0] = bo * rcp_f64(wave_sum_dpp(bo));
#pragma unroll
            for (int k = 0; k < NPL; ++k) e[k] = en[k];
            continue;
        }
        double y[NPL];
        float Sw;
        ss_bwd_step<NPL, MIX, MIX && H32>(cst, b, e, y, Sw);
        const double inv = (double)__builtin_amdgcn_rcpf(Sw);
#pragma unroll
        for (int k = 0; k < NPL; ++k) b[k] = y[k] * inv;
        {
            // two positions per trip (the DPP moves are convergent operations: the compiler will not unroll this loop itself)
            float S2;
            int t = 1;
            for (; t + 1 < span; t += 2) {
                ss_bwd_step<NPL, MIX, MIX && H32>(cst, b, e, y, S2);
#pragma unroll
                for (int k = 0; k < NPL; ++k) b[k] = y[k];
                ss_bwd_step<NPL, MIX, MIX && H32>(cst, b, e, y, S2);
#pragma unroll
                for (int k = 0; k < NPL; ++k) b[k] = y[k];
            }
            if (t < span) {
                ss_bwd_step<NPL, MIX, MIX && H32>(cst, b, e, y, S2);
#pragma unroll
                for (int k = 0; k < NPL; ++k) b[k] = y[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k) e[k] = en[k];
        if (MIX) {
#pragma unroll
            for (int k = 0; k < NPL; ++k) bp[k] -= Mp;
        }
    }
    if (a.dbg && !RERUN && c == 1 && lane == 0) {
        a.dbg[4] = __builtin_readcyclecounter() - t0c; a.dbg[5] = __builtin_amdgcn_s_memrealtime() - t0r;
        a.dbg[6] = npos; a.dbg[7] = nrows;
    }
    if (merged) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) if (stor[k]) end_cur[st[k]] = end_prev[st[k]];
        return;
    }
    {
        // (padded positions hold the lower-triangle total, not zero: their emission entry is zero, so they never feed a scan)
        double part = 0.0;
#pragma unroll
        for (int k = 0; k < NPL; ++k) part += live[k] ? b[k] : 0.0;
        const double S = wave_sum_dpp(part);
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const double bf = live[k] ? b[k] / S : 0.0;          // beta /= beta.sum()  (seeds gamma[:,0], hmm.cpp:150)
            if (stor[k]) { end_cur[st[k]] = bf; if (ch.first) a.beta[(size_t)ch.base * Mp + st[k]] = bf; }
        }
        if (lane == 0) a.endchg_b[pass] = 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Light passes.  A chunk needs ~13 e-folds of history before its rows are exact (the chains forget with an e-fold of
// 240 / 340 positions), i.e. on one 100 Mbp contig cut into 512 chunks MORE positions of history than positions of its own.
// Whatever a pass computes from a start vector that is still off is recomputed later, so those passes neither store rows
// nor need fp64: they run the same scans in float (fused v_add_f32_dpp: one instruction per level instead of three) and
// hand on the chunk's end vector only.  The pass that follows them is a full fp64 pass from their end vectors, then the
// usual re-run passes with skip test, merge exit and certificate - so a light pass can only change HOW FAST the fixed
// point is reached, never the result.
// ---------------------------------------------------------------------------------------------------------------
template <int NPL>
struct SsLightC { float dc[NPL], g[NPL], cg[NPL], b[NPL], a[NPL], cumA[NPL], lv[6], c15, c31, c0; };

template <int NPL, bool BWD>
__device__ __forceinline__ void ss_load_light(const SsArgs &a, int lane, SsLightC<NPL> &c) {
    double cum = 1.0;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int p = lane * NPL + k;
        c.dc[k] = (float)(BWD ? a.b_dc[p] : a.f_dc[p]);
        c.g[k] = (float)(BWD ? a.b_g[p] : a.f_g[p]);
        c.cg[k] = BWD ? 0.f : (float)a.f_cg[p];
        c.b[k] = (float)(BWD ? a.b_b[p] : a.f_b[p]);
        const double av = BWD ? a.b_a[p] : a.f_a[p];
        c.a[k] = (float)av;
        cum *= av;
        c.cumA[k] = (float)cum;
    }
    double lv[6];
    ss_levels(cum, lane, lv);
#pragma unroll
    for (int q = 0; q < 6; ++q) c.lv[q] = (float)lv[q];
    const int row = lane >> 4;
    c.c15 = (row & 1) ? 1.f : 0.f;
    c.c31 = (row >= 2) ? 1.f : 0.f;
    c.c0 = (float)a.c0;
}

// The lane-level float scans of the light passes as ONE instruction per level and chain: v_add_f32 / v_fmac_f32 take the DPP
// move as an operand modifier (the compiler fuses the add but not the multiply-add).  A DPP read of a register needs two wait
// states after the VALU write: three interleaved chains provide them, two chains take one s_nop per level.
// tools/dpp_lab.hip: every one of these instructions issues in 5.3 clocks, as a plain v_add_f32 does.
#define SS_DPP_ "row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
template <bool H32>
__device__ __forceinline__ void ss_light_scan2(float &p, float &z, const float (&lv)[6], float c15, float c31) {
    if (H32) {        // M <= 32: the live states fill two rows - no row_bcast:31 level
        asm volatile("s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %0 row_shr:1 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %2 row_shr:1 " SS_DPP_ "s_nop 0\n"
                     "v_add_f32_dpp %0, %0, %0 row_shr:2 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %3 row_shr:2 " SS_DPP_ "s_nop 0\n"
                     "v_add_f32_dpp %0, %0, %0 row_shr:4 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %4 row_shr:4 " SS_DPP_ "s_nop 0\n"
                     "v_add_f32_dpp %0, %0, %0 row_shr:8 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %5 row_shr:8 " SS_DPP_ "s_nop 0\n"
                     "v_fmac_f32_dpp %0, %0, %7 row_bcast:15 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %6 row_bcast:15 " SS_DPP_ "s_nop 1\n"
                     : "+v"(p), "+v"(z)
                     : "v"(lv[0]), "v"(lv[1]), "v"(lv[2]), "v"(lv[3]), "v"(lv[4]), "v"(c15));
        return;
    }
    asm volatile("s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 row_shr:1 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %2 row_shr:1 " SS_DPP_ "s_nop 0\n"
                 "v_add_f32_dpp %0, %0, %0 row_shr:2 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %3 row_shr:2 " SS_DPP_ "s_nop 0\n"
                 "v_add_f32_dpp %0, %0, %0 row_shr:4 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %4 row_shr:4 " SS_DPP_ "s_nop 0\n"
                 "v_add_f32_dpp %0, %0, %0 row_shr:8 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %5 row_shr:8 " SS_DPP_ "s_nop 0\n"
                 "v_fmac_f32_dpp %0, %0, %8 row_bcast:15 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %6 row_bcast:15 " SS_DPP_ "s_nop 0\n"
                 "v_fmac_f32_dpp %0, %0, %9 row_bcast:31 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %7 row_bcast:31 " SS_DPP_ "s_nop 1\n"
                 : "+v"(p), "+v"(z)
                 : "v"(lv[0]), "v"(lv[1]), "v"(lv[2]), "v"(lv[3]), "v"(lv[4]), "v"(lv[5]), "v"(c15), "v"(c31));
}
template <bool H32>
__device__ __forceinline__ void ss_light_scan3(float &p, float &z, float &f, const float (&lv)[6], float c15, float c31) {
    if (H32) {
        asm volatile("s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %0 row_shr:1 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %3 row_shr:1 " SS_DPP_ "v_add_f32_dpp %2, %2, %2 row_shr:1 " SS_DPP_
                     "v_add_f32_dpp %0, %0, %0 row_shr:2 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %4 row_shr:2 " SS_DPP_ "v_add_f32_dpp %2, %2, %2 row_shr:2 " SS_DPP_
                     "v_add_f32_dpp %0, %0, %0 row_shr:4 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %5 row_shr:4 " SS_DPP_ "v_add_f32_dpp %2, %2, %2 row_shr:4 " SS_DPP_
                     "v_add_f32_dpp %0, %0, %0 row_shr:8 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %6 row_shr:8 " SS_DPP_ "v_add_f32_dpp %2, %2, %2 row_shr:8 " SS_DPP_
                     "v_fmac_f32_dpp %0, %0, %8 row_bcast:15 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %7 row_bcast:15 " SS_DPP_ "v_fmac_f32_dpp %2, %2, %8 row_bcast:15 " SS_DPP_ "s_nop 1\n"
                     : "+v"(p), "+v"(z), "+v"(f)
                     : "v"(lv[0]), "v"(lv[1]), "v"(lv[2]), "v"(lv[3]), "v"(lv[4]), "v"(c15));
        return;
    }
    asm volatile("s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 row_shr:1 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %3 row_shr:1 " SS_DPP_ "v_add_f32_dpp %2, %2, %2 row_shr:1 " SS_DPP_
                 "v_add_f32_dpp %0, %0, %0 row_shr:2 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %4 row_shr:2 " SS_DPP_ "v_add_f32_dpp %2, %2, %2 row_shr:2 " SS_DPP_
                 "v_add_f32_dpp %0, %0, %0 row_shr:4 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %5 row_shr:4 " SS_DPP_ "v_add_f32_dpp %2, %2, %2 row_shr:4 " SS_DPP_
                 "v_add_f32_dpp %0, %0, %0 row_shr:8 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %6 row_shr:8 " SS_DPP_ "v_add_f32_dpp %2, %2, %2 row_shr:8 " SS_DPP_
                 "v_fmac_f32_dpp %0, %0, %9 row_bcast:15 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %7 row_bcast:15 " SS_DPP_ "v_fmac_f32_dpp %2, %2, %9 row_bcast:15 " SS_DPP_
                 "v_fmac_f32_dpp %0, %0, %10 row_bcast:31 " SS_DPP_ "v_fmac_f32_dpp %1, %1, %8 row_bcast:31 " SS_DPP_ "v_fmac_f32_dpp %2, %2, %10 row_bcast:31 " SS_DPP_ "s_nop 1\n"
                 : "+v"(p), "+v"(z), "+v"(f)
                 : "v"(lv[0]), "v"(lv[1]), "v"(lv[2]), "v"(lv[3]), "v"(lv[4]), "v"(lv[5]), "v"(c15), "v"(c31));
}
#undef SS_DPP_

template <int NPL, bool H32 = false>
__device__ __forceinline__ void ss_fwd_step_f(const SsLightC<NPL> &c, const float (&x)[NPL], const float (&e)[NPL],
                                              float (&out)[NPL], float &S) {
    float lp[NPL], w[NPL];
    lp[0] = x[0];
    w[0] = c.b[0] * x[0];
#pragma unroll
    for (int k = 1; k < NPL; ++k) {
        lp[k] = lp[k - 1] + x[k];
        w[k] = __builtin_fmaf(c.a[k], w[k - 1], c.b[k] * x[k]);
    }
    float p_ = lp[NPL - 1], z_ = w[NPL - 1];
    ss_light_scan2<H32>(p_, z_, c.lv, c.c15, c.c31);
    S = lane_get(p_, H32 ? 31 : 63);              // (H32: the live states end at lane 31, the rows above never receive their sum)
    const float LIp = dpp0<DPP_WSHR1>(z_);
    if (NPL == 1) {
        out[0] = e[0] * __builtin_fmaf(c.dc[0], x[0], __builtin_fmaf(c.g[0], S, __builtin_fmaf(c.cg[0], p_, LIp)));
        return;
    }
    const float lex = p_ - lp[NPL - 1];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const float incl = lex + lp[k];
        const float Z = (k == 0) ? LIp : __builtin_fmaf(c.cumA[k - 1 < 0 ? 0 : k - 1], LIp, w[k - 1 < 0 ? 0 : k - 1]);
        out[k] = e[k] * __builtin_fmaf(c.dc[k], x[k], __builtin_fmaf(c.g[k], S, __builtin_fmaf(c.cg[k], incl, Z)));
    }
}

template <int NPL, bool H32 = false>
__device__ __forceinline__ void ss_bwd_step_f(const SsLightC<NPL> &c, const float (&bv)[NPL], const float (&e)[NPL],
                                              float (&out)[NPL], float &Sw) {
    float w[NPL], lg[NPL], u[NPL], lf[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) w[k] = e[k] * bv[k];
    lg[0] = c.g[0] * w[0];
    u[0] = w[0];
    lf[0] = w[0];
#pragma unroll
    for (int k = 1; k < NPL; ++k) {
        lg[k] = __builtin_fmaf(c.g[k], w[k], lg[k - 1]);
        u[k] = __builtin_fmaf(c.a[k], u[k - 1], w[k]);
        lf[k] = lf[k - 1] + w[k];
    }
    float p_ = lg[NPL - 1], z_ = u[NPL - 1], f_ = lf[NPL - 1];
    ss_light_scan3<H32>(p_, z_, f_, c.lv, c.c15, c.c31);
    const float Gtot = lane_get(p_, 63);
    Sw = lane_get(f_, 63);
    const float LIp = dpp0<DPP_WSHR1>(z_);
    if (NPL == 1) {
        out[0] = __builtin_fmaf(c.dc[0], w[0], (Gtot - p_) + __builtin_fmaf(c.c0, f_, c.b[0] * LIp));
        return;
    }
    const float lexg = p_ - lg[NPL - 1], lexf = f_ - lf[NPL - 1];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const float inclG = lexg + lg[k], inclW = lexf + lf[k];
        const float V = (k == 0) ? LIp : __builtin_fmaf(c.cumA[k - 1 < 0 ? 0 : k - 1], LIp, u[k - 1 < 0 ? 0 : k - 1]);
        out[k] = __builtin_fmaf(c.dc[k], w[k], (Gtot - inclG) + __builtin_fmaf(c.c0, inclW, c.b[k] * V));
    }
}

template <int NPL, bool BWD, bool ALLLDS>
__device__ __forceinline__ void ss_emission_f(const SsArgs &a, const double *sE, int slot, int lane, float (&e)[NPL]) {
    double ed[NPL];
    ss_emission<NPL, BWD, ALLLDS>(a, sE, slot, lane, ed);
#pragma unroll
    for (int k = 0; k < NPL; ++k) e[k] = (float)ed[k];
}

// float, store-free steps of the forward chain over rows first+1 .. last of the contig at `base`: x on entry = the vector at row
// `first`, on return the (un-normalised) vector at row `last`.  The halo of a first pass (ss_forward_wave).
template <int NPL, bool ALLLDS>
__device__ __forceinline__ void ss_fwd_light_rows(const SsArgs &a, const double *sE, long long base, int first, int last, int lane,
                                                  float (&x)[NPL]) {
    SsLightC<NPL> cst;
    ss_load_light<NPL, false>(a, lane, cst);
    const int2 *rd = a.rowdesc + base + first + 1;
    const int nrows = last - first;
    int2 dcur = rd[lane], dnxt = rd[64 + lane];
    ss_desc_settle(dcur); ss_desc_settle(dnxt);
    float e[NPL];
    ss_emission_f<NPL, false, ALLLDS>(a, sE, __builtin_amdgcn_readlane(dcur.x, 0) & 0xFFFF, lane, e);
    ss_vm_drain();
    for (int j = 0; j < nrows; ++j) {
        const int jl = j & 63;
        const int span = __builtin_amdgcn_readlane(dcur.y, jl);
        if (jl == 63) { dcur = dnxt; dnxt = rd[j + 65 + lane]; ss_desc_settle(dnxt); }
        const int slot_n = __builtin_amdgcn_readlane(dcur.x, (j + 1) & 63) & 0xFFFF;
        float en[NPL], y[NPL], S;
        ss_emission_f<NPL, false, ALLLDS>(a, sE, slot_n, lane, en);
        ss_fwd_step_f<NPL>(cst, x, e, y, S);
        const float inv = __builtin_amdgcn_rcpf(S);
#pragma unroll
        for (int k = 0; k < NPL; ++k) x[k] = y[k] * inv;
        float S2;
        int t = 1;
        for (; t + 1 < span; t += 2) {
            ss_fwd_step_f<NPL>(cst, x, e, y, S2);
#pragma unroll
            for (int k = 0; k < NPL; ++k) x[k] = y[k];
            ss_fwd_step_f<NPL>(cst, x, e, y, S2);
#pragma unroll
            for (int k = 0; k < NPL; ++k) x[k] = y[k];
        }
        if (t < span) {
            ss_fwd_step_f<NPL>(cst, x, e, y, S2);
#pragma unroll
            for (int k = 0; k < NPL; ++k) x[k] = y[k];
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k) e[k] = en[k];
    }
}
// ... and of the backward chain over rows rhi .. rlo+1 (b on entry = the vector at row rhi, on return at row rlo)
template <int NPL, bool ALLLDS>
__device__ __forceinline__ void ss_bwd_light_rows(const SsArgs &a, const double *sE, long long base, int rhi, int rlo, int lane,
                                                  float (&b)[NPL]) {
    SsLightC<NPL> cst;
    ss_load_light<NPL, true>(a, lane, cst);
    const int2 *rd = a.rowdesc + base + rhi;
    const int nrows = rhi - rlo;
    int2 dcur = rd[-lane], dnxt = rd[-64 - lane];
    ss_desc_settle(dcur); ss_desc_settle(dnxt);
    float e[NPL];
    ss_emission_f<NPL, true, ALLLDS>(a, sE, __builtin_amdgcn_readlane(dcur.x, 0) & 0xFFFF, lane, e);
    ss_vm_drain();
    for (int j = 0; j < nrows; ++j) {
        const int jl = j & 63;
        const int span = __builtin_amdgcn_readlane(dcur.y, jl);
        if (jl == 63) { dcur = dnxt; dnxt = rd[-(j + 65) - lane]; ss_desc_settle(dnxt); }
        const int slot_n = __builtin_amdgcn_readlane(dcur.x, (j + 1) & 63) & 0xFFFF;
        float en[NPL], y[NPL], Sw;
        ss_emission_f<NPL, true, ALLLDS>(a, sE, slot_n, lane, en);
        ss_bwd_step_f<NPL>(cst, b, e, y, Sw);
        const float inv = __builtin_amdgcn_rcpf(Sw);
#pragma unroll
        for (int k = 0; k < NPL; ++k) b[k] = y[k] * inv;
        float S2;
        int t = 1;
        for (; t + 1 < span; t += 2) {
            ss_bwd_step_f<NPL>(cst, b, e, y, S2);
#pragma unroll
            for (int k = 0; k < NPL; ++k) b[k] = y[k];
            ss_bwd_step_f<NPL>(cst, b, e, y, S2);
#pragma unroll
            for (int k = 0; k < NPL; ++k) b[k] = y[k];
        }
        if (t < span) {
            ss_bwd_step_f<NPL>(cst, b, e, y, S2);
#pragma unroll
            for (int k = 0; k < NPL; ++k) b[k] = y[k];
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k) e[k] = en[k];
    }
}

template <int NPL, bool ALLLDS, bool H32 = false>
__device__ __forceinline__ void ss_forward_light(const SsArgs &a, const double *sE, int c, int lane) {
    const int M = a.M, Mp = a.Mp, pass = a.pass;
    const Chunk ch = ss_uniform_chunk(a.chunks[c]);
    float *end_cur = a.ends_f + ((size_t)(pass & 1) * a.nchunks + c) * Mp;
    int st[NPL];
    bool live[NPL], stor[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) { st[k] = lane * NPL + k; live[k] = st[k] < M; stor[k] = st[k] < Mp; }
    float x[NPL];
    {
        const float *src = (ch.first || pass == 0) ? a.pi_f : a.ends_f + ((size_t)((pass + 1) & 1) * a.nchunks + (c - 1)) * Mp;
#pragma unroll
        for (int k = 0; k < NPL; ++k) x[k] = live[k] ? src[st[k]] : 0.f;
    }
    if (lane == 0) { a.changed_f[pass] = 1; a.endchg_f[pass] = 1; }
    SsLightC<NPL> cst;
    ss_load_light<NPL, false>(a, lane, cst);
    const int2 *rd = a.rowdesc + ch.base + ch.r0 + 1;
    const int nrows = ch.r1 - ch.r0;
    int2 dcur = rd[lane], dnxt = rd[64 + lane];
    ss_desc_settle(dcur); ss_desc_settle(dnxt);
    float e[NPL];
    ss_emission_f<NPL, false, ALLLDS>(a, sE, __builtin_amdgcn_readlane(dcur.x, 0) & 0xFFFF, lane, e);
    ss_vm_drain();
    for (int j = 0; j < nrows; ++j) {
        const int jl = j & 63;
        const int span = __builtin_amdgcn_readlane(dcur.y, jl);
        if (jl == 63) { dcur = dnxt; dnxt = rd[j + 65 + lane]; ss_desc_settle(dnxt); }
        const int slot_n = __builtin_amdgcn_readlane(dcur.x, (j + 1) & 63) & 0xFFFF;
        float en[NPL], y[NPL], S;
        ss_emission_f<NPL, false, ALLLDS>(a, sE, slot_n, lane, en);
        ss_fwd_step_f<NPL, H32>(cst, x, e, y, S);
        const float inv = __builtin_amdgcn_rcpf(S);
#pragma unroll
        for (int k = 0; k < NPL; ++k) x[k] = y[k] * inv;
        {
            // two positions per trip (the DPP moves are convergent operations: the compiler will not unroll this loop itself)
            float S2;
            int t = 1;
            for (; t + 1 < span; t += 2) {
                ss_fwd_step_f<NPL, H32>(cst, x, e, y, S2);
#pragma unroll
                for (int k = 0; k < NPL; ++k) x[k] = y[k];
                ss_fwd_step_f<NPL, H32>(cst, x, e, y, S2);
#pragma unroll
                for (int k = 0; k < NPL; ++k) x[k] = y[k];
            }
            if (t < span) {
                ss_fwd_step_f<NPL, H32>(cst, x, e, y, S2);
#pragma unroll
                for (int k = 0; k < NPL; ++k) x[k] = y[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k) e[k] = en[k];
    }
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) part += x[k];
    const float inv = 1.f / wave_sum_dpp(part);
#pragma unroll
    for (int k = 0; k < NPL; ++k) if (stor[k]) end_cur[st[k]] = live[k] ? fmaxf(x[k] * inv, ss_floor_at<NPL>(a, ch.base + ch.r1 + 1)) : 0.f;
}

template <int NPL, bool ALLLDS, bool H32 = false>
__device__ __forceinline__ void ss_backward_light(const SsArgs &a, const double *sE, int c, int lane) {
    constexpr int MS = 64 * NPL;
    const int M = a.M, Mp = a.Mp, pass = a.pass;
    const Chunk ch = ss_uniform_chunk(a.chunks_b[c]);
    double *end_cur = a.ends_b + ((size_t)(pass & 1) * a.nchunks_b + c) * Mp;
    int st[NPL];
    bool live[NPL], stor[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) { st[k] = MS - 1 - (lane * NPL + k); live[k] = st[k] < M; stor[k] = st[k] < Mp; }
    float b[NPL];
    {
        const bool fresh = ch.last || pass == 0;
        const double *src = a.ends_b + ((size_t)((pass + 1) & 1) * a.nchunks_b + (fresh ? c : c + 1)) * Mp;
#pragma unroll
        for (int k = 0; k < NPL; ++k) b[k] = live[k] ? (fresh ? 1.f / (float)M : (float)src[st[k]]) : 0.f;
    }
    if (lane == 0) { a.changed_b[pass] = 1; a.endchg_b[pass] = 1; }
    SsLightC<NPL> cst;
    ss_load_light<NPL, true>(a, lane, cst);
    const int2 *rd = a.rowdesc + ch.base + ch.r1;
    const int nrows = ch.r1 - ch.r0;
    int2 dcur = rd[-lane], dnxt = rd[-64 - lane];
    ss_desc_settle(dcur); ss_desc_settle(dnxt);
    float e[NPL];
    ss_emission_f<NPL, true, ALLLDS>(a, sE, __builtin_amdgcn_readlane(dcur.x, 0) & 0xFFFF, lane, e);
    ss_vm_drain();
    for (int j = 0; j < nrows; ++j) {
        const int jl = j & 63;
        const int span = __builtin_amdgcn_readlane(dcur.y, jl);
        if (jl == 63) { dcur = dnxt; dnxt = rd[-(j + 65) - lane]; ss_desc_settle(dnxt); }
        const int slot_n = __builtin_amdgcn_readlane(dcur.x, (j + 1) & 63) & 0xFFFF;
        float en[NPL], y[NPL], Sw;
        ss_emission_f<NPL, true, ALLLDS>(a, sE, slot_n, lane, en);
        ss_bwd_step_f<NPL, H32>(cst, b, e, y, Sw);
        const float inv = __builtin_amdgcn_rcpf(Sw);
#pragma unroll
        for (int k = 0; k < NPL; ++k) b[k] = y[k] * inv;
        {
            // two positions per trip (the DPP moves are convergent operations: the compiler will not unroll this loop itself)
            float S2;
            int t = 1;
            for (; t + 1 < span; t += 2) {
                ss_bwd_step_f<NPL, H32>(cst, b, e, y, S2);
#pragma unroll
                for (int k = 0; k < NPL; ++k) b[k] = y[k];
                ss_bwd_step_f<NPL, H32>(cst, b, e, y, S2);
#pragma unroll
                for (int k = 0; k < NPL; ++k) b[k] = y[k];
            }
            if (t < span) {
                ss_bwd_step_f<NPL, H32>(cst, b, e, y, S2);
#pragma unroll
                for (int k = 0; k < NPL; ++k) b[k] = y[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k) e[k] = en[k];
    }
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) part += live[k] ? b[k] : 0.f;
    const float inv = 1.f / wave_sum_dpp(part);
#pragma unroll
    for (int k = 0; k < NPL; ++k) if (stor[k]) end_cur[st[k]] = live[k] ? (double)(b[k] * inv) : 0.0;
}

// One workgroup = 4 wavefronts = chunks 2 blk, 2 blk + 1 forward (wavefronts 0, 1) and backward (wavefronts 2, 3); they share
// one LDS copy of the emission vectors of the `nlds` most frequent keys.
// (the hybrid instantiation may run 8 wavefronts per workgroup - two per SIMD behind ONE copy of the eigenvector tables)
// H32: M <= 32, one state per lane - the scans of the light passes and of the all-float stored passes skip their widest level
template <int NPL, bool HYB, bool ALLLDS, bool H32 = false>
__global__ __launch_bounds__(HYB ? 512 : 256) void k_chain_ss(SsArgs a) {
    constexpr int MS = 64 * NPL;
    extern __shared__ __attribute__((aligned(16))) double ss_lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // (a FULL pass never idles: it is the first pass that stores rows, and with a warm start it may be the first pass launched at
    // all - the flag of the pass before it was then never written)
    const bool idle_f = (a.mode_f == 1 && !a.full_f && a.changed_f[a.pass - 1] == 0);
    const bool idle_b = (a.mode_b == 1 && !a.full_b && a.changed_b[a.pass - 1] == 0);
    if (idle_f && idle_b) return;
    const int nthr = HYB ? (int)blockDim.x : 256;
    for (int idx = tid; idx < a.nlds * MS; idx += nthr) ss_lds[idx] = a.E[idx];
    if (HYB && NPL == 1) {
        // eigenvector tables of the hybrid rows behind the emission vectors: [Ke][4][Mp][Mp + 1]
        double *tab = ss_lds + (size_t)a.nlds * MS;
        const int Mp = a.Mp, MM = Mp * Mp;
        if (a.dirsplit) {
            // [Ke][2][Mp][Mp + 1]: the task table gives every wavefront of this workgroup the same direction (the host builds it so)
            const int nw = nthr >> 6;
            int t0 = -1;
            for (int i = 0; i < nw && t0 < 0; ++i) t0 = a.tasks[nw * blockIdx.x + i];
            const bool wg_bwd = t0 >= 0 && (t0 >> 30);
            for (int idx = tid; idx < a.nk_lds * 2 * MM; idx += nthr) {
                const int mat = idx / MM, rc = idx % MM, r = rc / Mp, cc = rc % Mp;
                const int e = a.key_of_slot[mat >> 1], which = (mat & 1) + (wg_bwd ? 2 : 0);
                const double *src = which == 0 ? a.Pinvrm : which == 1 ? a.Prm : which == 2 ? a.PT : a.PinvT;
                tab[((size_t)mat * Mp + r) * (Mp + 1) + cc] = src[(size_t)e * MM + rc];
            }
        } else
        for (int idx = tid; idx < a.nk_lds * 4 * MM; idx += nthr) {
            const int mat = idx / MM, rc = idx % MM, r = rc / Mp, cc = rc % Mp;
            const int e = a.key_of_slot[mat >> 2], which = mat & 3;
            const double *src = which == 0 ? a.Pinvrm : which == 1 ? a.Prm : which == 2 ? a.PT : a.PinvT;
            tab[((size_t)mat * Mp + r) * (Mp + 1) + cc] = src[(size_t)e * MM + rc];
        }
    }
    __syncthreads();
    const int task = ss_uni(a.tasks[(nthr >> 6) * blockIdx.x + w]);
    if (task < 0) return;
    const bool fwd = !(task >> 30);
    const int c = task & 0x3FFFFFFF;
    if (fwd) {
        if (idle_f) return;
        if (NPL == 1 && !HYB && a.mixed) {
            if (a.mode_f == 0) ss_forward_wave<NPL, false, HYB, ALLLDS, true, H32>(a, ss_lds, c, lane);
            else if (a.mode_f == 1) ss_forward_wave<NPL, true, HYB, ALLLDS, true, H32>(a, ss_lds, c, lane);
            else ss_forward_light<NPL, ALLLDS, H32>(a, ss_lds, c, lane);
        } else
        if (a.mode_f == 0) ss_forward_wave<NPL, false, HYB, ALLLDS>(a, ss_lds, c, lane);
        else if (a.mode_f == 1) ss_forward_wave<NPL, true, HYB, ALLLDS>(a, ss_lds, c, lane);
        else ss_forward_light<NPL, ALLLDS, H32>(a, ss_lds, c, lane);
    } else {
        if (idle_b) return;
        if (NPL == 1 && !HYB && a.mixed) {
            if (a.mode_b == 0) ss_backward_wave<NPL, false, HYB, ALLLDS, true, H32>(a, ss_lds, c, lane);
            else if (a.mode_b == 1) ss_backward_wave<NPL, true, HYB, ALLLDS, true, H32>(a, ss_lds, c, lane);
            else ss_backward_light<NPL, ALLLDS, H32>(a, ss_lds, c, lane);
        } else
        if (a.mode_b == 0) ss_backward_wave<NPL, false, HYB, ALLLDS>(a, ss_lds, c, lane);
        else if (a.mode_b == 1) ss_backward_wave<NPL, true, HYB, ALLLDS>(a, ss_lds, c, lane);
        else ss_backward_light<NPL, ALLLDS, H32>(a, ss_lds, c, lane);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Span fold on the scans (round 4).  The eigen-free span statistics (kernels.hpp: k_span_big) need, per (contig, key),
//     F_t = Acc_{t+1} + F_{t+1} A,   H_t = F_t + A H_{t+1}   (t = s_max - 1 .. 0),   W = H_0,  g = diag(A W),   A = diag(e) T^T.
// k_span_big forms the products on the matrix cores: 2 M^3 flop per step, and the steps are serial.  But a ROW of F times A is
// the backward operator applied to that row,  (f A)_j = sum_k f_k e_k T[j][k] = (T (e o f))_j,  and A times a COLUMN of H is the
// forward operator,  (A h)_i = e_i (T^T h)_i  - the very O(M) scan steps of the chains above.  Rows of F do not mix, columns of H
// do not mix: ONE WAVEFRONT per row (phase 0) / column (phase 1) walks all s_max steps on its own - no barrier, no LDS, no
// matrix product: 2 s_max M applications of an O(M) operator instead of 2 s_max products of M x M matrices.
// Phase 0 writes F_t TRANSPOSED (scatter) so that phase 1 reads its column as a contiguous row.
// grid = ceil(n_ce * M / 4) workgroups of 4 independent wavefronts.
// ---------------------------------------------------------------------------------------------------------------
template <int NPL, int PHASE>
__global__ __launch_bounds__(256) void k_span_scan(SsArgs sa, FinArgs a, int smax, double *__restrict__ Fall) {
    constexpr int MS = 64 * NPL;
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int M = a.M, Mp = a.Mp;
    const int ce = ss_uni(gw / M), rc = ss_uni(gw % M);                // (contig, key) and the row (phase 0) / column (phase 1)
    if (ce >= a.n_contigs * a.Ke) return;
    const int b0 = ss_uni(a.ce_bucket_off[ce]), b1 = ss_uni(a.ce_bucket_off[ce + 1]);
    if (b0 == b1) return;
    const int e = ce % a.Ke;
    const double *ek = a.E + (size_t)a.e_kid[e] * Mp;
    double *Fce = Fall + (size_t)ce * smax * Mp * Mp;
    // states of this lane: forward positions p = state, backward positions p = MS - 1 - state (chains above)
    int st[NPL];
    bool live[NPL];
    double ev[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int p = lane * NPL + k;
        st[k] = PHASE == 0 ? MS - 1 - p : p;
        live[k] = st[k] < M;
        ev[k] = live[k] ? ek[st[k]] : 0.0;
    }
    double x[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) x[k] = 0.0;
    if (PHASE == 0) {
        // bucket of every span: lane L learns the bucket that holds span L + 1 (s_max <= 64), -1 = none
        int mybk = -1;
        for (int b = b0; b < b1; ++b) {
            const int sp = a.g_span[a.eb_gid[b]];
            if (sp - 1 == lane) mybk = b;
        }
        SsBwdC<NPL> c;
        ss_load_bwd<NPL>(sa, lane, c);
        auto fetch = [&](int t, double (&v)[NPL]) {                      // row rc of the bucket of span t + 1 (raw; masked when used)
            const int bk = __builtin_amdgcn_readlane(mybk, max(t, 0));
            const double *src = a.red_e + (size_t)max(bk, b0) * Mp * Mp + (size_t)rc * Mp;
#pragma unroll
            for (int k = 0; k < NPL; ++k) v[k] = src[min(st[k], Mp - 1)];
            return bk;
        };
        double nx[NPL];
        int bkn = fetch(smax - 1, nx);
        for (int t = smax - 1; t >= 0; --t) {
            double cur[NPL], out[NPL];
            const bool has = bkn >= 0;
#pragma unroll
            for (int k = 0; k < NPL; ++k) cur[k] = (has && live[k]) ? nx[k] : 0.0;
            bkn = fetch(t - 1, nx);                                      // next step's bucket row is on its way during this step
            float Sw;
            ss_bwd_step<NPL>(c, x, ev, out, Sw);
            double *Ft = Fce + (size_t)t * Mp * Mp;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                x[k] = out[k] + cur[k];
                if (live[k]) Ft[(size_t)st[k] * Mp + rc] = x[k];         // F_t[rc][state] stored as FT_t[state][rc]
            }
        }
        return;
    }
    SsFwdC<NPL> c;
    ss_load_fwd<NPL>(sa, lane, c);
    auto fetchF = [&](int t, double (&v)[NPL]) {                         // column rc of F_t = row rc of the transposed copy
        const double *src = Fce + (size_t)max(t, 0) * Mp * Mp + (size_t)rc * Mp;
#pragma unroll
        for (int k = 0; k < NPL; ++k) v[k] = src[min(st[k], Mp - 1)];
    };
    double nx[NPL];
    fetchF(smax - 1, nx);
    for (int t = smax - 1; t >= 0; --t) {
        double cur[NPL], out[NPL], S;
#pragma unroll
        for (int k = 0; k < NPL; ++k) cur[k] = live[k] ? nx[k] : 0.0;
        fetchF(t - 1, nx);
        ss_fwd_step<NPL>(c, x, ev, out, S);
#pragma unroll
        for (int k = 0; k < NPL; ++k) x[k] = out[k] + cur[k];
    }
    // W[:, rc] = H_0 (row-major W in the eigen path's Y buffer) and g[rc] = (A W)[rc][rc] (first Mp entries of its Z buffer)
    double *Wout = a.Y + (size_t)ce * Mp * Mp;
    double *gout = a.Z + (size_t)ce * Mp * Mp;
    double out[NPL], S;
    ss_fwd_step<NPL>(c, x, ev, out, S);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        if (live[k]) Wout[(size_t)st[k] * Mp + rc] = x[k];
        if (st[k] == rc) gout[rc] = out[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Per-row posteriors of LONG rows at 64 < M <= 256 (un-binned data, the input of `smc++ posterior`; round 6).  The dense chains take
// such a row in ONE eigen-power step (hmm.cpp:72-78,104-112) and the reference then forms its gamma from the eigensystem
// (hmm.cpp:113-121: 2 M^3 flop per row - a scalar kernel here beyond 64 states, 3.4 s per E-step on 10^5 rows at M = 256).  The same
// vector is the sum over the row's positions of their posteriors (k_gamma_rows_scan below), and that sum is cut into PIECES of at most
// 64 positions that run in parallel: the forward vector at a piece's start and the backward vector at its end are eigen-power
// INTERPOLATIONS of the row's two stored vectors,
//     f_o = P (d~^o o (P^-1 alpha_{l-1})),     h_o = P^-T (d~^(span - o) o (P^T beta_l)),
// - two M x M products per piece on the matrix cores (k_piece_vectors; P^-1 alpha and P^T beta are the statistics' own k_eig_uw
// products) -, the positions inside a piece are walked by the O(M) scan steps, and k_gamma_merge_pieces adds a row's pieces up.
// No span-Q table, no per-row M^3 product; the chains and the statistics stay un-cut (one eigen step per row).
// ---------------------------------------------------------------------------------------------------------------
struct GPiece {
    long long row;            // global row of the piece's data row
    int q;                    // position of that row in the sorted eigen-row permutation (Xs / Ys)
    int o0, len, span;        // the piece covers positions o0 + 1 .. o0 + len of the row's `span`
    int kid, es;              // emission key, eigen key
};
struct GTile { int es, cnt; int pid[16]; };       // sixteen pieces of ONE eigen key that need an interpolated vector

struct PieceArgs {
    int M, Mp, npieces, ntiles;
    const GPiece *pieces;
    const GTile *tiles;
    const double *Xs, *Ys;    // [eigen rows][Mp]  omega P^-1 alpha_{l-1},  P^T beta_l   (k_eig_uw)
    const double *dsc;        // [Ke][Mp] scaled eigenvalues
    const double *PT, *Pinvrm;        // [Ke][Mp][Mp]  PT[a][i] = P[i][a],  Pinvrm[a][i] = Pinv[a][i]
    const double *cs;         // [Ke][2][Mp] row sums of PT / Pinvrm (k_piece_rowsums): the sum of an interpolated vector without forming it
    const double *l2d;        // [Ke][Mp] log2 |d~| (k_piece_rowsums): d~^k = exp2(k log2 |d~|) - a fifth of pow()'s instructions
    float *pvf;               // [npieces][Mp] forward vector at the piece's start (o0 > 0)
    double *pvb;              // [npieces][Mp] backward vector at the piece's end (o0 + len < span)
    double *pgam;             // [npieces][Mp] the piece's share of its row's gamma (k_gamma_rows_scan<., true>)
};

__global__ __launch_bounds__(256) void k_piece_rowsums(int Ke, int Mp, const double *__restrict__ PT, const double *__restrict__ Pinvrm,
                                                       double *__restrict__ cs, const double *__restrict__ dsc, double *__restrict__ l2d) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < Ke * Mp) l2d[idx] = log2(fabs(dsc[idx]));            // (-inf for a zero eigenvalue: exp2 gives 0)
    if (idx >= Ke * 2 * Mp) return;
    const int a = idx % Mp, dir = (idx / Mp) & 1, e = idx / (2 * Mp);
    const double *src = (dir ? Pinvrm : PT) + (size_t)e * Mp * Mp + (size_t)a * Mp;
    double acc = 0.0;
    for (int i = 0; i < Mp; ++i) acc += src[i];
    cs[idx] = acc;
}

// One wavefront per tile of sixteen pieces and direction: x_piece = d~^k o (omega u | w) through LDS as the A operand
// (A[m = piece][k = state a]), the key's matrix as the B operand (B[k = a][n = i]: sixteen consecutive doubles of row a per lane
// group - whole 128-byte lines), D[piece][i] scaled to unit sum and stored as the piece's start (float, as alpha is) / end vector.
__global__ __launch_bounds__(256) void k_piece_vectors(PieceArgs a) {
    extern __shared__ __attribute__((aligned(16))) double pv_sm[];
    const int Mp = a.Mp, M = a.M, LDX = Mp + 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    double *sX = pv_sm + (size_t)wv * 16 * LDX;
    // A workgroup takes FOUR consecutive tiles in ONE direction and its wavefronts walk the output tiles in step (a barrier per tile):
    // the 128-byte lines of the key's matrix one of them pulls from L2 are still in the CU's vector L1 when the other three ask for them.
    const int nunits = 2 * ((a.ntiles + 3) / 4);
    for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        const int dir = unit & 1;
        const int tile = 4 * (unit >> 1) + wv;
        const bool have = tile < a.ntiles;
        const GTile &tl = a.tiles[min(tile, a.ntiles - 1)];
        const int es = ss_uni(tl.es), cnt = have ? ss_uni(tl.cnt) : 0;
        const int pid = tl.pid[max(0, min(n, cnt - 1))];
        const GPiece pc = a.pieces[pid];
        const int kpow = dir ? pc.span - pc.o0 - pc.len : pc.o0;
        const bool need = n < cnt && kpow > 0;
        const double *X = (dir ? a.Ys : a.Xs) + (size_t)pc.q * Mp;
        const double *dsc = a.dsc + (size_t)es * Mp;
        const double *l2d = a.l2d + (size_t)es * Mp;
        const double *cs = a.cs + ((size_t)es * 2 + dir) * Mp;
        const double kd = (double)kpow;
        double part = 0.0;
        for (int kk = 0; kk < Mp / 4; ++kk) {
            const int st = 4 * kk + kq;
            double v = 0.0;
            if (need && st < M) {
                v = exp2(kd * l2d[st]) * X[st];
                if ((kpow & 1) && dsc[st] < 0.0) v = -v;               // (an eigenvalue rounding made negative: (-|d|)^k)
            }
            part = fma(cs[st], v, part);
            sX[n * LDX + st] = v;
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const double inv = (need && part != 0.0) ? 1.0 / part : 0.0;
        int pid_r[4];
        double inv_r[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pid_r[r] = __shfl(pid, kq + 4 * r);
            inv_r[r] = __shfl(inv, kq + 4 * r);
        }
        wave_lds_fence();
        const double *Mat = (dir ? a.Pinvrm : a.PT) + (size_t)es * Mp * Mp;
        for (int it = 0; it < Mp / 16; ++it) {
            __syncthreads();
            f64x4 D = (f64x4){0, 0, 0, 0};
            const double *Bp = Mat + (size_t)kq * Mp + it * 16 + n;
            for (int kk = 0; kk < Mp / 4; kk += 4) {              // (Mp is a multiple of 16)
                double av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { av[u] = sX[n * LDX + 4 * (kk + u) + kq]; bv[u] = Bp[(size_t)4 * (kk + u) * Mp]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) D = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], D, 0, 0, 0);
            }
            const int i = it * 16 + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (inv_r[r] == 0.0) continue;
                const double v = D[r] * inv_r[r];
                if (dir) a.pvb[(size_t)pid_r[r] * Mp + i] = v;
                else a.pvf[(size_t)pid_r[r] * Mp + i] = (float)v;
            }
        }
        wave_lds_fence();
    }
}

// gamma of eigen row q = the sum of its pieces' shares (pieces pfirst[q] .. pfirst[q + 1] - 1, in position order: deterministic)
__global__ __launch_bounds__(256) void k_gamma_merge_pieces(int Mp, int nrows, const int *__restrict__ pfirst, const GPiece *__restrict__ pieces,
                                                            const double *__restrict__ pgam, double *__restrict__ gamma_rows) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long q = idx / Mp;
    const int i = (int)(idx % Mp);
    if (q >= nrows) return;
    const int p0 = pfirst[q], p1 = pfirst[q + 1];
    double acc = 0.0;
    for (int p = p0; p < p1; ++p) acc += pgam[(size_t)p * Mp + i];
    gamma_rows[(size_t)pieces[p0].row * Mp + i] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// Per-row posterior of the span > 1 rows WITHOUT an eigensystem (round 6; save_gamma on the scan chains).
// hmm.cpp:113-121 forms the row's gamma from the eigensystem of its key - diag(P d (Q_r o span_Q) P^-1), 2 M^3 flop per row -
// and normalises it to the row's span.  That vector is the sum over the `span` positions of the row of their posteriors:
//     v = sum_{t=1..span} f_t o h_t / (f_t . h_t),   f_t = (B T^T)^t alpha_{ell-1},   h_t = (T B)^{span-t} beta_ell
// (f_t . h_t is the same number for every t; each term sums to one, so v sums to the span as the reference's does).  With the scan
// form of the operator (ss_fwd_step / ss_bwd_step: O(M) per position) a row costs 2 span - 1 steps instead of 2 M^3 flop: 1e5
// against 2.7e8 operations at M = 512, and no eigensystem, no span-Q table - save_gamma keeps the eigen-free statistics, and is no
// longer limited to 256 states.  One wavefront per row (persistent wavefronts, NPL states per lane): the forward vectors f_1 .. f_s are
// parked as floats (alpha itself is a float) in the wavefront's own scratch piece, then the backward walk multiplies them in.  Both
// walks rescale by the running sum; every term is normalised by its own dot product, so the scales cancel.
// The generators of one direction are loaded inside that direction's walk (sixteen states per lane: both sets at once do not fit).
// ---------------------------------------------------------------------------------------------------------------
// PIECES: the work items are the pieces of k_piece_vectors' table instead of whole rows - start / end vectors from the interpolated
// buffers (or the row's own alpha / beta where the piece starts / ends the row), the share goes to the piece's own line of `pgam`.
template <int NPL, bool PIECES = false>
__global__ __launch_bounds__(256) void k_gamma_rows_scan(SsArgs sa, GammaRowArgs a, const RowInfo *__restrict__ rowinfo,
                                                         const double *__restrict__ Ek, float *__restrict__ scratch, int smax, int nwaves,
                                                         PieceArgs pa = PieceArgs()) {
    constexpr int MS = 64 * NPL;
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= nwaves) return;
    const int M = a.M, Mp = a.Mp;
    float *park = scratch + (size_t)gw * smax * MS;
    const int nitems = PIECES ? pa.npieces : a.nrows;
    for (int q = gw; q < nitems; q += nwaves) {
        const int qs = ss_uni(q);
        int span;
        size_t row;
        const double *ek;
        const float *ap;
        const double *bp;
        double *gout;
        if (PIECES) {
            const GPiece pc = pa.pieces[qs];
            span = ss_uni(pc.len);
            row = (size_t)pc.row;
            ek = Ek + (size_t)ss_uni(pc.kid) * Mp;
            ap = ss_uni(pc.o0) == 0 ? a.alpha + (row - 1) * Mp : pa.pvf + (size_t)qs * Mp;
            bp = ss_uni(pc.o0 + pc.len - pc.span) == 0 ? a.beta + row * Mp : pa.pvb + (size_t)qs * Mp;
            gout = pa.pgam + (size_t)qs * Mp;
        } else {
            const Slab sl = a.slabs[a.row_slab[qs]];
            span = ss_uni(a.g_span[sl.aux]);
            row = (size_t)(sl.base + a.perm[qs]);
            ek = Ek + (size_t)ss_uni(rowinfo[row].kid) * Mp;
            ap = a.alpha + (row - 1) * Mp;
            bp = a.beta + row * Mp;
            gout = a.gamma_rows + row * Mp;
        }
        {
            // forward: positions p = state
            SsFwdC<NPL> c;
            ss_load_fwd<NPL>(sa, lane, c);
            double x[NPL], ev[NPL];
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int st = lane * NPL + k;
                const bool live = st < M;
                x[k] = live ? (double)ap[live ? st : 0] : 0.0;
                ev[k] = live ? ek[live ? st : 0] : 0.0;
            }
            double inv = 1.0;
            for (int t = 0; t < span; ++t) {
                double out[NPL], S;
                ss_fwd_step<NPL>(c, x, ev, out, S);
                // (the sum of the vector the step started from: bounded, and it CANCELS below - every term is normalised by its own dot
                // product -, so the rescaling factor need not be exact: the float reciprocal, three instructions instead of the ~30 of an
                // fp64 division, in each of the 2 span - 1 steps)
                inv = (double)__builtin_amdgcn_rcpf((float)S);
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    x[k] = out[k] * inv;
                    park[(size_t)t * MS + lane * NPL + k] = (float)x[k];
                }
            }
        }
        // the parked vectors are read back by OTHER lanes of this wavefront (reversed state order): the stores have to be acknowledged
        // first (one CU, one vector L1: a workgroup-scope fence is a wait, no cache maintenance)
        __threadfence_block();
        {
            // backward: positions p = state MS - 1 - p
            SsBwdC<NPL> c;
            ss_load_bwd<NPL>(sa, lane, c);
            double h[NPL], ev[NPL], g[NPL];
            int st[NPL];
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                st[k] = MS - 1 - (lane * NPL + k);
                const bool live = st[k] < M;
                h[k] = live ? bp[live ? st[k] : 0] : 0.0;
                ev[k] = live ? ek[live ? st[k] : 0] : 0.0;
                g[k] = 0.0;
            }
            for (int t = span - 1; t >= 0; --t) {
                double pr[NPL], part = 0.0;
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    pr[k] = (double)park[(size_t)t * MS + st[k]] * h[k];
                    part += pr[k];
                }
                const double idot = 1.0 / wave_sum_dpp(part);
#pragma unroll
                for (int k = 0; k < NPL; ++k) g[k] = __builtin_fma(pr[k], idot, g[k]);
                if (t > 0) {
                    double out[NPL];
                    float Sw;
                    ss_bwd_step<NPL>(c, h, ev, out, Sw);
                    const double is = (double)__builtin_amdgcn_rcpf(Sw);          // (a rescaling only, as above)
#pragma unroll
                    for (int k = 0; k < NPL; ++k) h[k] = out[k] * is;
                }
            }
#pragma unroll
            for (int k = 0; k < NPL; ++k)
                if (st[k] < Mp) gout[st[k]] = st[k] < M ? g[k] : 0.0;
        }
    }
}

// Unit-test entry (tests/test_gpu_ss.py through smcpp_debug_ss_apply): out_f = e o (T^T x), out_b = T (e o x) by the scans,
// one wavefront per vector.  MIX / H32: the all-float scans of the stored passes (one state per lane) and their M <= 32 form.
template <int NPL, bool MIX = false, bool H32 = false>
__global__ __launch_bounds__(64) void k_ss_apply(SsArgs a, const double *__restrict__ x, const double *__restrict__ e,
                                                 double *__restrict__ out_f, double *__restrict__ out_b, int nvec) {
    constexpr int MS = 64 * NPL;
    const int lane = threadIdx.x, v = blockIdx.x;
    if (v >= nvec) return;
    SsFwdC<NPL> cf;
    SsBwdC<NPL> cb;
    ss_load_fwd<NPL>(a, lane, cf);
    ss_load_bwd<NPL>(a, lane, cb);
    double xf[NPL], ef[NPL], xb[NPL], eb[NPL], yf[NPL], yb[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int p = lane * NPL + k, q = MS - 1 - p;
        xf[k] = x[(size_t)v * MS + p]; ef[k] = e[(size_t)v * MS + p];
        xb[k] = x[(size_t)v * MS + q]; eb[k] = e[(size_t)v * MS + q];
    }
    double S; float Sw;
    ss_fwd_step<NPL, MIX, H32>(cf, xf, ef, yf, S);
    ss_bwd_step<NPL, MIX, H32>(cb, xb, eb, yb, Sw);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int p = lane * NPL + k, q = MS - 1 - p;
        out_f[(size_t)v * MS + p] = yf[k];
        out_b[(size_t)v * MS + q] = yb[k];
    }
}

}  // namespace smcpp_dev
