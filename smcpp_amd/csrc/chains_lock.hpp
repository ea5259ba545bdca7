// Lock-step chains on the matrix cores (M <= 64, MT = padded width; inputs with long chunks even at 16 chunks per CU: a whole
// genome on one GPU).  SIXTEEN chunks advance together in one workgroup: the state is the MT x 16 matrix X (one column per chunk) and a
// row-step is  Y = T^T X  (chunks on a span-1 row),  U = Pinv X, V = P (d^s o U)  (chunks on an eigen row) as
// v_mfma_f64_16x16x4_f64 tiles - wavefront w owns state tile w (MT / 16 wavefronts), the A fragments (MT / 4 k-steps each) of T and
// of the hot eigen key stay in registers, X goes through LDS once per product.  All products are computed for all 16 columns
// (neighbouring chunks sit on rows of different type) and the result is selected per column.  Emission and eigenvalue-power
// vectors of the NEXT row are fetched from L2 one step ahead (the descriptors are known), so no table lives in LDS and a
// forward and a backward workgroup share a CU.
//
// Semantics are those of hmm.cpp:57-149 exactly as k_fwd_coop2 / k_bwd_coop2 (chains2.hpp) implement them: float alpha with
// the 1e-10 floor applied relative to the running sum, double beta, chunk-parallel fixed point with per-chunk skip test,
// merge exit and certificate (here per COLUMN; a workgroup stops when all of its columns have).  The forward span-1 product is a
// float product as in the reference (v_mfma_f32_16x16x4, other summation order than Eigen's: the float noise the reference
// itself carries).
// tools/chain_lab.hip `m` measured the structure first: 2.35 us per lock-step row with three products = 147 ns per chain-row
// against 630 ns of the cooperative kernels - but 16 x more chunks, hence 16 x shorter ones, and every chunk pays ~900 rows of
// re-run history: it only wins above ~3.7 M rows per GPU.
#pragma once

namespace smcpp_dev {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int LOCK_NC = 16;      // chunks (columns) per workgroup
constexpr int LOCK_W = 64;       // descriptor window (rows per refill)

template <int MT>
struct LockShared {
    double Xs[2][MT][LOCK_NC];   // exchanged state, double-buffered
    double Us[MT][LOCK_NC];      // d^s o (Pinv x) of the eigen columns
    int2 desc[LOCK_NC][LOCK_W];
    long long base[LOCK_NC];     // global row index of (contig row r0 + 1) of every column
    int nrows[LOCK_NC];
    int bad[LOCK_NC];            // merge test: some state of the column still differs
    int act[LOCK_NC];            // skip test: the column's start vector changed
    int any_active;
};

// one operator applied to X: acc = A X with A fragments `af` (16 k-steps) and B fragments `b`
template <int KS>
__device__ __forceinline__ f64x4 lock_prod(const double (&af)[KS], const double (&b)[KS]) {
    f64x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < KS; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t], b[t], acc, 0, 0, 0);
    return acc;
}

// A fragments of a row-major [MT][MT] matrix restricted to output tile w: A[m = lane & 15][k = 4 t + (lane >> 4)] = Mx[(16 w + m) MT + k]
template <int MT>
__device__ __forceinline__ void lock_frags_rm(const double *__restrict__ Mx, int w, int lane, double (&af)[MT / 4]) {
    const int m = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int t = 0; t < MT / 4; ++t) af[t] = Mx[(size_t)(16 * w + m) * MT + 4 * t + kk];
}
// ... of the TRANSPOSE of a row-major matrix: A[m][k] = Mx[k MT + 16 w + m]
template <int MT>
__device__ __forceinline__ void lock_frags_tr(const double *__restrict__ Mx, int w, int lane, double (&af)[MT / 4]) {
    const int m = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int t = 0; t < MT / 4; ++t) af[t] = Mx[(size_t)(4 * t + kk) * MT + 16 * w + m];
}

template <int MT, bool RERUN>
__global__ __launch_bounds__(MT * 4) void k_fwd_lock(ChainArgs a) {
    static_assert(MT % 16 == 0 && MT >= 16 && MT <= 64, "lock-step chains: one wavefront per tile of 16 padded states");
    constexpr int Mp = MT, KS = MT / 4, NTH = MT * 4;
    __shared__ __attribute__((aligned(16))) LockShared<MT> sh;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 15, kk = lane >> 4;                 // column (chunk), k within a 4-step / row group of the D tile
    const int M = a.M, pass = a.pass;
    if (RERUN && a.changed[pass - 1] == 0) return;
    const int c = blockIdx.x * LOCK_NC + n;                  // this lane's chunk
    const bool exists = c < a.nchunks;
    Chunk ch = a.chunks[exists ? c : a.nchunks - 1];
    const int nrows = exists ? ch.r1 - ch.r0 : 0;
    float *end_cur = a.ends_f + ((size_t)(pass & 1) * a.nchunks + c) * Mp;
    const float *end_prev = a.ends_f + ((size_t)((pass + 1) & 1) * a.nchunks + c) * Mp;
    // ---- start vectors: states i_r = 16 w + kk + 4 r of column n ----
    float al[4];
    {
        const float *src = ch.first ? a.pi_f
                           : !RERUN ? (a.warm_f ? a.warm_f + (size_t)(c - 1) * Mp : a.pi_f)
                                    : a.ends_f + ((size_t)((pass + 1) & 1) * a.nchunks + (c - 1)) * Mp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * w + kk + 4 * r;
            al[r] = (exists && i < M) ? src[i] : 0.f;
        }
    }
    if (tid < LOCK_NC) { sh.act[tid] = 0; sh.bad[tid] = 0; }
    if (tid == 0) sh.any_active = 0;
    __syncthreads();
    bool active = exists;
    if (RERUN) {
        if (exists && ch.first) active = false;              // the first chunk of a contig never changes
        if (active) {
            bool diff = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * w + kk + 4 * r;
                if (i < M) {
                    const float u = a.used_f[(size_t)c * Mp + i];
                    if (!(fabsf(al[r] - u) <= a.eps_f * fabsf(u))) diff = true;
                }
            }
            if (diff) sh.act[n] = 1;
        }
        __syncthreads();
        active = active && sh.act[n] != 0;
        if (exists && !active) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int i = 16 * w + kk + 4 * r; end_cur[i] = end_prev[i]; }
        }
    }
    if (active) sh.any_active = 1;
    __syncthreads();
    if (sh.any_active == 0) return;
    if (tid == 0) a.changed[pass] = 1;
    if (active) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * w + kk + 4 * r;
            a.used_f[(size_t)c * Mp + i] = al[r];
            if (ch.first) a.alpha[(size_t)ch.base * Mp + i] = al[r];
        }
        if (ch.first && w == 0 && kk == 0) a.cnorm[ch.base] = 1.0;
    }
    // ---- operators: T^T and the hot eigen key in registers ----
    // T^T in FLOAT (the reference's span-1 product is a float product, hmm.cpp:83): v_mfma_f32_16x16x4 is half the cycles of
    // the f64 form.  Its D layout is row = 4 (lane >> 4) + reg where the f64 form has (lane >> 4) + 4 reg, so the rows of the
    // A operand are permuted (tile row m stands for state (m >> 2) + 4 (m & 3)): both products then hold the same four states
    // in the same four registers of a lane
    float at[KS];
    double ap[KS], aq[KS];
    {
        const int m = lane & 15, sig = (m >> 2) + 4 * (m & 3);
#pragma unroll
        for (int t = 0; t < KS; ++t) at[t] = (float)a.TdT[(size_t)(16 * w + sig) * MT + 4 * t + kk];       // (T^T)[i][k] = TdT[i][k]
    }
    const size_t ho = (size_t)(a.hot < 0 ? 0 : a.hot) * Mp * Mp;
    lock_frags_rm<MT>(a.Pinvrm + ho, w, lane, ap);                            // Pinv[i][k]
    lock_frags_rm<MT>(a.Prm + ho, w, lane, aq);                               // P[i][k]
#pragma unroll
    for (int t = 0; t < KS; ++t) { pin_reg(at[t]); pin_reg(ap[t]); pin_reg(aq[t]); }
    // ---- per-column bookkeeping in LDS, first descriptor window, start state ----
    if (w == 0 && kk == 0) {
        sh.base[n] = ch.base + ch.r0 + 1;
        sh.nrows[n] = active ? nrows : 0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sh.Xs[0][16 * w + kk + 4 * r][n] = (double)al[r];
    __syncthreads();
    int maxrows = 0;
    for (int q = 0; q < LOCK_NC; ++q) maxrows = max(maxrows, sh.nrows[q]);
    auto fill_window = [&](int j0) {
        for (int x = tid; x < LOCK_NC * LOCK_W; x += NTH) {
            const int q = x / LOCK_W, o = x % LOCK_W;
            const int nr = sh.nrows[q];
            sh.desc[q][o] = a.rowdesc[sh.base[q] + (nr > 0 ? min(j0 + o, nr - 1) : 0)];
        }
    };
    fill_window(0);
    __syncthreads();
    // descriptor and per-row operands (emission vector, eigenvalue powers) of the row the next step processes: fetched from L2
    // right after the hand-over of the previous step, in flight while the products run
    int2 d_cur = sh.desc[n][0];
    double e_cur[4], dp_cur[4];
    auto load_row_operands = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * w + kk + 4 * r;
            e_cur[r] = a.E[(size_t)d_cur.x * Mp + i];
            dp_cur[r] = d_cur.y >= 0 ? a.dpow[(size_t)SMCPP_GID(d_cur.y) * Mp + i] : 0.0;
        }
    };
    load_row_operands();
    float vprev[4] = {al[0], al[1], al[2], al[3]};
    float *arow = a.alpha + (size_t)(ch.base + ch.r0) * Mp;              // row r0 + j at iteration j
    double *crow = a.cnorm + ch.base + ch.r0;
    bool running = active;               // still computing / storing
    bool merged = false;
    for (int j = 0; j <= maxrows; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        // ---- B fragments of X, column sum (= normaliser of the previous row), relative floor ----
        double b[KS];
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < KS; ++t) { b[t] = sh.Xs[cur][4 * t + kk][n]; s += b[t]; }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float sprev = (j == 0) ? 1.0f : (float)s;
        const float inv = __builtin_amdgcn_rcpf(sprev);
        const double thr = (double)(1e-10f * sprev);
        // ---- the previous row can be finished: alpha = max(v / s, 1e-10)   (hmm.cpp:89-94) ----
        if (RERUN && j > 16 && (j & 15) == 1 && running && sh.bad[n] == 0) { running = false; merged = true; }
        if (j > 0 && running && j <= nrows) {
            float an[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * w + kk + 4 * r;
                an[r] = (i < M) ? fmaxf(vprev[r] * inv, 1e-10f) : 0.f;
            }
            if (RERUN && (j & 15) == 0 && j < nrows) {
                bool bad = false;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * w + kk + 4 * r;
                    const float old = arow[(size_t)j * Mp + i];
                    if (i < M && !(fabsf(an[r] - old) <= a.eps_f * fabsf(old))) bad = true;
                }
                if (bad) sh.bad[n] = 1;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) arow[(size_t)j * Mp + 16 * w + kk + 4 * r] = an[r];
            if (w == 0 && kk == 0) crow[j] = (double)sprev;
            if (j == nrows) {
#pragma unroll
                for (int r = 0; r < 4; ++r) end_cur[16 * w + kk + 4 * r] = an[r];
                running = false;
            }
        }
        if (RERUN && (j & 15) == 15 && tid < LOCK_NC) sh.bad[tid] = 0;       // cleared one step before the next test row
        // every wavefront sees the status of all 16 columns (lanes n, any kk): uniform exit
        if (!__any(running)) break;
        const bool compute = running && j < nrows;
        // ---- products ----
#pragma unroll
        for (int t = 0; t < KS; ++t) b[t] = fmax(b[t], thr);
        const bool eig = compute && d_cur.y >= 0;
        const int es = eig ? SMCPP_ES(d_cur.y) : -1;
        f32x4v Y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KS; ++t) Y = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t], (float)b[t], Y, 0, 0, 0);
        f64x4 U = {0, 0, 0, 0};
        const bool any_eig = __any(eig);
        bool other = eig && es != a.hot;
        if (any_eig) {
            if (__any(eig && es == a.hot)) U = lock_prod<KS>(ap, b);
            // eigen keys that are not register-resident: one at a time, fragments from L2
            unsigned long long rest = __ballot(other);
            while (rest) {
                const int src = __ffsll((long long)rest) - 1;
                const int e2 = __shfl(es, src, 64);
                double af[KS];
                lock_frags_rm<MT>(a.Pinvrm + (size_t)e2 * Mp * Mp, w, lane, af);
                const f64x4 U2 = lock_prod<KS>(af, b);
                if (es == e2) U = U2;
                rest &= ~__ballot(es == e2);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * w + kk + 4 * r;
                sh.Us[i][n] = eig ? U[r] * dp_cur[r] * (double)inv : 0.0;
            }
        }
        lds_barrier();                       // Us complete (every wave takes the same branches: it sees all 16 descriptors)
        f64x4 V = {0, 0, 0, 0};
        if (any_eig) {
            double ub[KS];
#pragma unroll
            for (int t = 0; t < KS; ++t) ub[t] = sh.Us[4 * t + kk][n];
            if (__any(eig && es == a.hot)) V = lock_prod<KS>(aq, ub);
            unsigned long long rest = __ballot(other);
            while (rest) {
                const int src = __ffsll((long long)rest) - 1;
                const int e2 = __shfl(es, src, 64);
                double af[KS];
                lock_frags_rm<MT>(a.Prm + (size_t)e2 * Mp * Mp, w, lane, af);
                const f64x4 V2 = lock_prod<KS>(af, ub);
                if (es == e2) V = V2;
                rest &= ~__ballot(es == e2);
            }
        }
        // ---- select, round to float as alpha_hat is, hand over ----
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * w + kk + 4 * r;
            const float y = Y[r] * inv;
            float v = eig ? (float)V[r] : (float)((double)y * e_cur[r]);
            v = (i < M && compute) ? v : 0.f;
            vprev[r] = v;
            sh.Xs[nxt][i][n] = (double)v;
        }
        lds_barrier();                       // Xs[nxt] complete
        // ---- next row: descriptor (window refilled every LOCK_W steps), operands ----
        if ((j + 1) % LOCK_W == 0) {
            fill_window(j + 1);
            __syncthreads();
        }
        d_cur = sh.desc[n][(j + 1) % LOCK_W];
        load_row_operands();
    }
    if (merged) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int i = 16 * w + kk + 4 * r; end_cur[i] = end_prev[i]; }
    }
}

// Backward chain, same organisation.  Iteration j handles row ell = r1 - j of every column: the exchanged vector z is beta
// itself before an eigen row and e o beta before a span-1 row (the emission factor is applied by the PRODUCER of z, on its four
// D rows, with the emission vector of the next row fetched one step ahead); beta is stored in the running scale - every
// consumer of beta is invariant to a per-row scale - and the chunk's end vector is normalised exactly.
template <int MT, bool RERUN>
__global__ __launch_bounds__(MT * 4) void k_bwd_lock(ChainArgs a) {
    static_assert(MT % 16 == 0 && MT >= 16 && MT <= 64, "lock-step chains: one wavefront per tile of 16 padded states");
    constexpr int Mp = MT, KS = MT / 4, NTH = MT * 4;
    __shared__ __attribute__((aligned(16))) LockShared<MT> sh;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 15, kk = lane >> 4;
    const int M = a.M, pass = a.pass;
    if (RERUN && a.changed[pass - 1] == 0) return;
    const int c = blockIdx.x * LOCK_NC + n;
    const bool exists = c < a.nchunks;
    Chunk ch = a.chunks[exists ? c : a.nchunks - 1];
    const int nrows = exists ? ch.r1 - ch.r0 : 0;
    double *end_cur = a.ends_b + ((size_t)(pass & 1) * a.nchunks + c) * Mp;
    const double *end_prev = a.ends_b + ((size_t)((pass + 1) & 1) * a.nchunks + c) * Mp;
    double bs[4];
    {
        const bool fresh = ch.last || (!RERUN && a.warm_b == nullptr);
        const double *src = !RERUN ? a.warm_b + (size_t)(c + 1) * Mp
                                   : a.ends_b + ((size_t)((pass + 1) & 1) * a.nchunks + (c + 1)) * Mp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * w + kk + 4 * r;
            bs[r] = (exists && i < M) ? (fresh ? 1.0 / (double)M : src[i]) : 0.0;
        }
    }
    if (tid < LOCK_NC) { sh.act[tid] = 0; sh.bad[tid] = 0; }
    if (tid == 0) sh.any_active = 0;
    __syncthreads();
    bool active = exists;
    if (RERUN) {
        if (exists && ch.last) active = false;
        if (active) {
            bool diff = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * w + kk + 4 * r;
                if (i < M) {
                    const double u = a.used_b[(size_t)c * Mp + i];
                    if (!(fabs(bs[r] - u) <= a.eps_b * fabs(u))) diff = true;
                }
            }
            if (diff) sh.act[n] = 1;
        }
        __syncthreads();
        active = active && sh.act[n] != 0;
        if (exists && !active) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int i = 16 * w + kk + 4 * r; end_cur[i] = end_prev[i]; }
        }
    }
    if (active) sh.any_active = 1;
    __syncthreads();
    if (sh.any_active == 0) return;
    if (tid == 0) a.changed[pass] = 1;
    if (active) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.used_b[(size_t)c * Mp + 16 * w + kk + 4 * r] = bs[r];
    }
    // ---- operators: T, P^T, Pinv^T of the hot key (A[m][k] of the transposes of the row-major / stored forms) ----
    double at[KS], ap[KS], aq[KS];
    lock_frags_tr<MT>(a.TdT, w, lane, at);                                    // T[i][k] = TdT[k][i]
    const size_t ho = (size_t)(a.hot < 0 ? 0 : a.hot) * Mp * Mp;
    lock_frags_tr<MT>(a.Prm + ho, w, lane, ap);                               // (P^T)[i][k] = P[k][i]
    lock_frags_tr<MT>(a.Pinvrm + ho, w, lane, aq);                            // (Pinv^T)[i][k] = Pinv[k][i]
#pragma unroll
    for (int t = 0; t < KS; ++t) { pin_reg(at[t]); pin_reg(ap[t]); pin_reg(aq[t]); }
    if (w == 0 && kk == 0) {
        sh.base[n] = ch.base + ch.r1;              // iteration j reads the descriptor of row r1 - j
        sh.nrows[n] = active ? nrows : 0;
    }
    __syncthreads();
    int maxrows = 0;
    for (int q = 0; q < LOCK_NC; ++q) maxrows = max(maxrows, sh.nrows[q]);
    auto fill_window = [&](int j0) {
        for (int x = tid; x < LOCK_NC * LOCK_W; x += NTH) {
            const int q = x / LOCK_W, o = x % LOCK_W;
            const int nr = sh.nrows[q];
            sh.desc[q][o] = a.rowdesc[sh.base[q] - (nr > 0 ? min(j0 + o, nr - 1) : 0)];
        }
    };
    fill_window(0);
    __syncthreads();
    int2 d_cur = sh.desc[n][0];
    double dp_cur[4];
    auto load_dp = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            dp_cur[r] = d_cur.y >= 0 ? a.dpow[(size_t)SMCPP_GID(d_cur.y) * Mp + 16 * w + kk + 4 * r] : 0.0;
    };
    load_dp();
    double b_raw[4] = {bs[0], bs[1], bs[2], bs[3]};
    {
        // z of the first row: e o beta if it is a span-1 row
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * w + kk + 4 * r;
            const double e0 = (d_cur.y < 0 && nrows > 0) ? a.E[(size_t)d_cur.x * Mp + i] : 1.0;
            sh.Xs[0][i][n] = bs[r] * e0;
        }
    }
    __syncthreads();
    double *brow = a.beta + (size_t)(ch.base + ch.r1) * Mp;             // row r1 - j at iteration j
    bool running = active, merged = false;
    for (int j = 0; j <= maxrows; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        double b[KS];
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < KS; ++t) { b[t] = sh.Xs[cur][4 * t + kk][n]; s += b[t]; }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        // descriptor of the NEXT row and, if it is a span-1 row, its emission vector on this lane's D rows: in flight during
        // the products (at a window boundary they are fetched after the refill at the end of the step)
        const bool boundary = (j + 1) % LOCK_W == 0;
        int2 d_nxt = make_int2(0, 0);
        double e_nxt[4] = {1.0, 1.0, 1.0, 1.0};
        auto load_next = [&]() {
            d_nxt = sh.desc[n][(j + 1) % LOCK_W];
            if (d_nxt.y < 0 && j + 1 < nrows) {
#pragma unroll
                for (int r = 0; r < 4; ++r) e_nxt[r] = a.E[(size_t)d_nxt.x * Mp + 16 * w + kk + 4 * r];
            }
        };
        if (!boundary) load_next();
        if (RERUN && j > 16 && (j & 15) == 1 && running && sh.bad[n] == 0) { running = false; merged = true; }
        if (running && j == nrows) {
            // end of the chunk: z is plain beta here (no emission factor was applied): exact normalisation (hmm.cpp:142)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * w + kk + 4 * r;
                const double bf = (i < M) ? b_raw[r] / s : 0.0;
                end_cur[i] = bf;
                if (ch.first) a.beta[(size_t)ch.base * Mp + i] = bf;
            }
            running = false;
        }
        if (running) {
            if (RERUN && (j & 15) == 0 && j > 0) {
                bool bad = false;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * w + kk + 4 * r;
                    const double old = brow[-(ptrdiff_t)j * Mp + i];
                    if (i < M && !(fabs(b_raw[r] - old) <= a.eps_b * fabs(old))) bad = true;
                }
                if (bad) sh.bad[n] = 1;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) brow[-(ptrdiff_t)j * Mp + 16 * w + kk + 4 * r] = b_raw[r];
        }
        if (RERUN && (j & 15) == 15 && tid < LOCK_NC) sh.bad[tid] = 0;
        if (!__any(running)) break;
        const bool compute = running && j < nrows;
        const double inv = rcp_f64(s);
        const bool eig = compute && d_cur.y >= 0;
        const int es = eig ? SMCPP_ES(d_cur.y) : -1;
        const f64x4 Z = lock_prod<KS>(at, b);
        f64x4 Wv = {0, 0, 0, 0};
        const bool any_eig = __any(eig);
        const bool other = eig && es != a.hot;
        if (any_eig) {
            if (__any(eig && es == a.hot)) Wv = lock_prod<KS>(ap, b);
            unsigned long long rest = __ballot(other);
            while (rest) {
                const int src = __ffsll((long long)rest) - 1;
                const int e2 = __shfl(es, src, 64);
                double af[KS];
                lock_frags_tr<MT>(a.Prm + (size_t)e2 * Mp * Mp, w, lane, af);
                const f64x4 W2 = lock_prod<KS>(af, b);
                if (es == e2) Wv = W2;
                rest &= ~__ballot(es == e2);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) sh.Us[16 * w + kk + 4 * r][n] = eig ? Wv[r] * dp_cur[r] * inv : 0.0;
        }
        lds_barrier();
        f64x4 O = {0, 0, 0, 0};
        if (any_eig) {
            double ub[KS];
#pragma unroll
            for (int t = 0; t < KS; ++t) ub[t] = sh.Us[4 * t + kk][n];
            if (__any(eig && es == a.hot)) O = lock_prod<KS>(aq, ub);
            unsigned long long rest = __ballot(other);
            while (rest) {
                const int src = __ffsll((long long)rest) - 1;
                const int e2 = __shfl(es, src, 64);
                double af[KS];
                lock_frags_tr<MT>(a.Pinvrm + (size_t)e2 * Mp * Mp, w, lane, af);
                const f64x4 O2 = lock_prod<KS>(af, ub);
                if (es == e2) O = O2;
                rest &= ~__ballot(es == e2);
            }
        }
        // ---- hand over: the producer applies the next row's emission vector when that row is a span-1 row ----
        if (boundary) {                      // (all waves are past this step's window reads: the barrier above)
            fill_window(j + 1);
            __syncthreads();
            load_next();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * w + kk + 4 * r;
            double bn = eig ? O[r] : Z[r] * inv;
            bn = (i < M && compute) ? bn : 0.0;
            b_raw[r] = bn;
            sh.Xs[nxt][i][n] = bn * e_nxt[r];
        }
        d_cur = d_nxt;
        load_dp();
        lds_barrier();                       // Xs[nxt] complete
    }
    if (merged) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int i = 16 * w + kk + 4 * r; end_cur[i] = end_prev[i]; }
    }
}

}  // namespace smcpp_dev
