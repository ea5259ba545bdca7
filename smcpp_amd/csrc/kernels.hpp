// HIP kernels of the SMC++ E-step engine for gfx950 (MI355X, CDNA4; 64-wide wavefronts).
//
// Reference semantics reproduced here: HMM::Estep, src/hmm.cpp:45-153.
// The restructuring (why these kernels do not look like hmm.cpp) is described in DESIGN.md:
//   * K1 `k_fwd_pass` / K2 `k_bwd_pass`: the two dependent chains, chunk-parallel.  One wavefront owns one chunk
//     of consecutive rows; chunk-boundary vectors are iterated to a fixed point over passes (the chains forget
//     their start vector geometrically), so after convergence every chunk has been run from the vector its left
//     (right) neighbour ended on.  Lane i owns hidden state i (+64q for M > 64).
//   * K3 `k_s1_scalars`, K4 `k_rank_acc`, K5 `k_eig_uw`: sufficient statistics, embarrassingly parallel over rows,
//     fp64 MFMA (v_mfma_f64_16x16x4_f64) rank-k updates into per-bucket M x M accumulators.
//   * K6 finalisation kernels: span-Q Hadamard, P * Z * Pinv * B, gamma diagonals, xisum o Td with the 1e-20 floor.
//   * K7 `k_loglik_partial` / `k_loglik_final`: log-normaliser reduction (deterministic two-stage tree).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smcpp_dev {

typedef double f64x4 __attribute__((ext_vector_type(4)));

struct Chunk {
    long long base;   // index of the contig's row 0 in the global (all contigs) row arrays
    int r0, r1;       // the chunk owns rows ell = r0+1 .. r1 (contig relative, 1-based like hmm.cpp)
    int contig;
    int first;        // r0 == 0
    int last;         // r1 == L
    int pad;
    // HALO of the scan chains' first pass (chains_ss.hpp): instead of starting at its own first row from pi / the uniform vector
    // the chunk's wavefront walks into it from rows of its NEIGHBOUR - forward: rows h0+1 .. h1 in float without stores, rows
    // h1+1 .. r0 in fp64 without stores (h0 <= h1 <= r0); backward: rows h0 .. h1+1 in float, h1 .. r1+1 in fp64 (h0 >= h1 >= r1).
    // h0 == h1 == r0 (forward) / r1 (backward): no halo.
    int h0 = 0, h1 = 0;
};

struct RowInfo {      // per global row index (entry for ell = 0 of each contig is unused)
    int kid;          // key id
    int gid;          // (span,key) group id for span > 1 rows, -1 for span == 1
};
// the chain kernels read a packed descriptor: gid | (eigen index << 20), or -1 for span-1 rows
#define SMCPP_GID(ge) ((ge) & 0xFFFFF)
#define SMCPP_ES(ge) ((ge) >> 20)

// Orders this wavefront's LDS traffic without touching the vector-memory counters (a __syncthreads() would also
// drain the outstanding alpha/beta stores and prefetches, which is what made the first version latency-bound).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    const int x = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const long long x = __builtin_bit_cast(long long, v);
    int lo = (int)(x & 0xffffffffll), hi = (int)(x >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float lane_get(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double lane_get(double v, int l) {
    const long long x = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(x & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(x >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// butterfly inside each 16-lane DPP row (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), then the
// four row totals are combined through SGPRs; every lane receives the same bits
template <typename T>
__device__ __forceinline__ T wave_sum_dpp(T v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    const T r0 = lane_get(v, 0), r1 = lane_get(v, 16), r2 = lane_get(v, 32), r3 = lane_get(v, 48);
    return (r0 + r1) + (r2 + r3);
}

// ---------------------------------------------------------------------------------------------------------------
// wavefront reductions (all 64 lanes receive the result)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// sum over the 16 lanes of a DPP row (lanes with equal lane>>4); every lane of the row gets the result
__device__ __forceinline__ double row16_sum(double v) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// K1 / K2: the forward (hmm.cpp:57-96) and backward (hmm.cpp:97-149) chains
// ---------------------------------------------------------------------------------------------------------------
struct ChainArgs {
    int M, Mp, nchunks, pass, hot;   // hot = eigen index whose matrices the *_hot kernels keep in registers (-1: none)
    int hot2;                        // second register-resident eigen key of the generation-2 cooperative chains (-1: none)
    int variant;                     // host-side dispatch: 0 = by pass number, 1 = eigen-free pre-pass (pass 0 on group powers),
                                     // 2 = full pass from the previous pass's end vectors (no skip test, no merge exit)
    const Chunk *chunks;
    const int2 *rowdesc;    // [rows] {kid, gid | es << 20} (-1 for span-1 rows)
    const double *E;        // [K][Mp] emission vectors
    const double *dpow;     // [G][Mp] (d_r/scale)^span per group
    // forward operands
    const float *pi_f;      // [Mp] float(pi)
    const float *Tf;        // [Mp][Mp] Tf[k][i] = float(T[k][i])
    const double *PinvT;    // [Ke][Mp][Mp] PinvT[j][i] = Pinv_r[i][j]
    const double *PT;       // [Ke][Mp][Mp] PT[j][i]    = P_r[i][j]
    // backward operands
    const double *TdT;      // [Mp][Mp] TdT[j][i] = T[i][j]
    const double *Prm;      // [Ke][Mp][Mp] P_r row-major
    const double *Pinvrm;   // [Ke][Mp][Mp] Pinv_r row-major
    // state
    float *alpha;           // [rows][Mp]
    double *beta;           // [rows][Mp]
    double *cnorm;          // [rows] forward normaliser c (log_c = log c + span*log scale is taken later)
    float *ends_f;          // [2][nchunks][Mp]
    float *used_f;          // [nchunks][Mp]
    double *ends_b;         // [2][nchunks][Mp]
    double *used_b;         // [nchunks][Mp]
    int *changed;           // [max passes]
    float eps_f;
    double eps_b;
    long long *dbg;         // optional [8]: cycle counters of workgroup 0 (SMCPP_DEBUG_CYCLES)
    // optional warm start (cooperative kernels): the converged chunk-boundary vectors of the previous E-step of this
    // manager, used instead of pi / the uniform vector as pass-0 start vectors; nullptr = cold start
    // eigen-free pre-pass only: binary powers A^2, A^4, A^8, A^16 of A = diag(e) T^T per eigen key, float row-major for
    // the forward chain, double transposed for the backward chain; span of every group
    const float *Bf;        // [Ke][4][Mp][Mp]
    const double *Bb;       // [Ke][4][Mp][Mp]
    const int *g_span;      // [G]
    int nbits, npow;        // eigen-free pre-pass: bits of the longest span, powers stored per key (= max(4, nbits - 1))
    int prio;               // wave priority (s_setprio) of the backward cooperative kernels, 0..3
    const float *warm_f;    // [nchunks][Mp] end vectors of the forward chunks
    const double *warm_b;   // [nchunks][Mp] end vectors of the backward chunks
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct f32x4p { f32x2 lo, hi; };   // 16 bytes: (x, y), (z, w)

__device__ __forceinline__ void lds_stage(void *dst, const void *src, int nbytes, int tid, int nthreads) {
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    for (int i = tid; i < nbytes / 16; i += nthreads) d[i] = s[i];
}

// ---------------------------------------------------------------------------------------------------------------
// K1'' / K2'': CU-cooperative chains for M <= 64 (Mp == MT in {16,32,48,64}).
// One workgroup of NW = MT/16 wavefronts advances ONE chunk.  Lane = 4*il + kq: wavefront w owns the 16 outputs
// i = 16w + il, and the four lanes of a quad split the inner dimension into quarters k in [kq*KQ, (kq+1)*KQ),
// KQ = MT/4, keeping their KQ elements of every operand matrix in registers (80 VGPRs at MT = 64, no LDS matrix
// traffic).  A mat-vec is KQ FMAs + a 2-step DPP quad reduction; vectors travel between wavefronts through a
// double-buffered LDS array and ONE s_barrier per mat-vec.  The chain state is carried unnormalised: a producer
// writes v (and its wavefront's partial sum), the consumer scales its *result* by 1/sum and clamps its inputs at
// 1e-10*sum, which is the reference's normalise-then-clamp (hmm.cpp:87-94) up to where the float rounding lands.
// Because a row costs a few hundred cycles instead of a few thousand, chunks can be 4x longer (one per CU instead
// of one per SIMD) and the boundary fixed point needs 2-3 passes instead of 5-8.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double quad_sum_d(double v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    return v;
}
__device__ __forceinline__ float quad_sum_f(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    return v;
}

struct CoopArgs {
    int K, G;
    int power_off;   // byte offset of the scratch vectors of the eigen-free pre-pass in the dynamic LDS
};

// Makes the compiler wait for a loaded value HERE (an empty asm that reads and writes the register).  Without it the
// one-time operand loads issued before the row loop are waited for lazily inside the loop with small vmcnt counts,
// and since stores share that counter on gfx9 every row would then also wait for the previous row's alpha store.
template <typename T>
__device__ __forceinline__ void pin_reg(T &x) {
    asm volatile("" : "+v"(x));
}

// Workgroup barrier that waits for this wavefront's LDS traffic only.  __syncthreads() also drains vmcnt, i.e. the
// alpha/beta stores of the row just finished (a full HBM write latency per row on the critical path).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ double rcp_f64(double s) {
    // v_rcp_f64 + two Newton steps: 1/s to ~1 ulp without the div_scale / div_fmas / div_fixup sequence
    double r = __builtin_amdgcn_rcp(s);
    r = fma(fma(-s, r, 1.0), r, r);
    r = fma(fma(-s, r, 1.0), r, r);
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// Cooperative chains for 64 < M <= 256: same workgroup-per-chunk scheme as k_fwd_coop / k_bwd_coop (lane 4*il + kq owns
// quarter kq of the inner index of state i, state exchanged through LDS, consumer-side normaliser), but the operand
// quarters no longer fit the register file (1024 threads leave 128 VGPRs each), so every matrix is STREAMED from L2
// once per row in a quarter-interleaved layout  Q[t][i][kq] = Mt[(kq*KQ + t)*Mp + i]  in which the 64 lanes of a
// wavefront read 64 consecutive elements (one fully used 256-/512-byte access per k-step).  One chunk per CU instead
// of the four independent wavefronts of the generic kernels: a quarter of the row-steps re-read the matrices, and
// chunks four times as long need 2-3 passes instead of 6-7.
// ---------------------------------------------------------------------------------------------------------------
struct BigArgs {
    const float *qTf;        // [KQ][Mp][4]            forward span-1 operand
    const double *qPinvT;    // [Ke][KQ][Mp][4]        forward eigen operands
    const double *qPT;
    const double *qTdT;      // [KQ][Mp][4]            backward span-1 operand
    const double *qPrm;      // [Ke][KQ][Mp][4]        backward eigen operands
    const double *qPinvrm;
    // eigen-free pre-pass (PRE instantiations): float binary powers A^(2^b), b = 0..nbits-1 (rescaled from A^32 on), of A = diag(e) T^T per eigen key,
    // qBf: Q[t][i][kq] = A^p[i][kq*KQ + t] (forward), qBb: Q[t][i][kq] = A^p[kq*KQ + t][i] (backward)   (k_pow_layout)
    const float *qBf;        // [Ke][nbits][KQ][Mp][4]
    const float *qBb;        // [Ke][nbits][KQ][Mp][4]
};

// One streamed quarter product: acc = sum_t q[t*QS] * x(t), with B independent loads in flight per batch (the whole
// row of the operand has to come from L2 every step; the loop is latency-bound unless many loads are outstanding).
template <int KQ, int B, typename TM, typename F>
__device__ __forceinline__ auto stream_dot(const TM *__restrict__ q, size_t QS, int rot, F &&xval) {
    using TA = decltype(q[0] * xval(0));
    TA acc[4] = {TA(0), TA(0), TA(0), TA(0)};
    static_assert(KQ % 4 == 0, "quarter length must be a multiple of 4");
    constexpr int BB = (KQ % B == 0) ? B : 4;
    // `rot` (a multiple of BB, different per workgroup) rotates the order of the k-steps: all workgroups stream the
    // same matrix at the same pace, and without it they all ask the same one or two L2 channels at the same time
#pragma unroll 1
    for (int n = 0; n < KQ; n += BB) {
        int t0 = n + rot;
        if (t0 >= KQ) t0 -= KQ;
        TM m[BB];
#pragma unroll
        for (int u = 0; u < BB; ++u) m[u] = q[(size_t)(t0 + u) * QS];
#pragma unroll
        for (int u = 0; u < BB; ++u) acc[u & 3] += m[u] * xval(t0 + u);
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// PRE = eigen-free pre-pass (engine_plans.hpp: stage_static_and_prepass): eigen rows apply the float binary powers of
// A = diag(e) T^T, one per set bit of the span, nothing is stored but the chunk's end vector - it only has to hand pass 1
// (a.variant == 2: a full pass from those end vectors, no skip test, no merge exit) a start vector while the host is
// still solving the eigenproblems.
template <int MT, bool PRE = false>
__global__ __launch_bounds__(MT * 4) void k_fwd_big(ChainArgs a, BigArgs qa) {
    constexpr int NW = MT / 16, KQ = MT / 4, Mp = MT, UP = KQ + 2, QS = 4 * MT;   // QS = stride of one k-step in a Q layout
    constexpr int FB = 16, DB = (MT > 128) ? 8 : 16;   // loads in flight per lane (1024 threads leave 128 VGPRs each)
    __shared__ __attribute__((aligned(16))) double ub[4 * UP];
    __shared__ __attribute__((aligned(16))) float xf[2 * MT];
    __shared__ __attribute__((aligned(16))) float tbf[PRE ? 2 * MT : 4];
    __shared__ int2 sdesc[128];
    __shared__ int sflag;
    __shared__ int mflag[NW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int il = lane >> 2, kq = lane & 3, i = 16 * w + il;
    const bool owner = kq == 0;
    const int M = a.M, pass = a.pass, c = blockIdx.x;
    const bool full = a.variant == 2;
    if (pass > 0 && a.changed[pass - 1] == 0) return;
    const Chunk ch = a.chunks[c];
    float *end_cur = a.ends_f + ((size_t)(pass & 1) * a.nchunks + c) * Mp;
    const float *end_prev = a.ends_f + ((size_t)((pass + 1) & 1) * a.nchunks + c) * Mp;
    if (pass > 0 && !full && ch.first) {
        if (owner) end_cur[i] = end_prev[i];
        return;
    }
    float al = 0.f;
    {
        const float *src = (ch.first || pass == 0) ? a.pi_f
                                                   : a.ends_f + ((size_t)((pass + 1) & 1) * a.nchunks + (c - 1)) * Mp;
        if (i < M) al = src[i];
    }
    if (tid == 0) sflag = 0;
    if (lane == 0) mflag[w] = 1;
    __syncthreads();
    if (pass > 0 && !full) {
        bool diff = false;
        if (owner && i < M) {
            const float u = a.used_f[(size_t)c * Mp + i];
            if (!(fabsf(al - u) <= a.eps_f * fabsf(u))) diff = true;
        }
        if (__any(diff) && lane == 0) sflag = 1;
        __syncthreads();
        if (sflag == 0) {
            if (owner) end_cur[i] = end_prev[i];
            return;
        }
    }
    if (owner) a.used_f[(size_t)c * Mp + i] = al;
    if (tid == 0) a.changed[pass] = 1;
    if (ch.first && !PRE) {
        if (owner) a.alpha[(size_t)ch.base * Mp + i] = al;
        if (tid == 0) a.cnorm[ch.base] = 1.0;
    }
    const int2 *rd = a.rowdesc + ch.base;
    const int nrows = ch.r1 - ch.r0;
    if (w == 0) {
        sdesc[lane] = (lane < nrows) ? rd[ch.r0 + 1 + lane] : make_int2(0, -1);
        sdesc[64 + lane] = (lane + 64 < nrows) ? rd[ch.r0 + 1 + 64 + lane] : make_int2(0, -1);
    }
    if (owner) xf[i] = al;
    __syncthreads();
    const size_t qoff = (size_t)4 * i + kq;                 // this lane's element of every k-step of a Q layout
    constexpr int FBB = (KQ % FB == 0) ? FB : 4, DBB = (KQ % DB == 0) ? DB : 4;
    const int rotf = (int)((blockIdx.x * 7u) % (unsigned)(KQ / FBB)) * FBB, rotd = (int)((blockIdx.x * 7u) % (unsigned)(KQ / DBB)) * DBB;
    float v_prev = al;
    const bool rerun = pass > 0 && !full;
    bool merged = false;
    for (int j = 0; j < nrows; ++j) {
        const int ell = ch.r0 + 1 + j;
        const int jb = j & 63, bsel = (j >> 6) & 1, cur = j & 1, nxt = cur ^ 1;
        if (rerun && j > 16 && (j & 15) == 1) {
            int nm = mflag[0];
#pragma unroll
            for (int q = 1; q < NW; ++q) nm |= mflag[q];
            if (nm == 0) { merged = true; break; }
        }
        if (w == 0 && jb == 32 && j >= 64) {
            const int2 dn = (j + 32 + lane < nrows) ? rd[ch.r0 + 1 + j + 32 + lane] : make_int2(0, -1);
            sdesc[(bsel ^ 1) * 64 + lane] = dn;
        }
        const int2 d0 = sdesc[bsel * 64 + jb];
        const int kid = __builtin_amdgcn_readfirstlane(d0.x);
        const int ge = __builtin_amdgcn_readfirstlane(d0.y);
        // ---- normaliser of the previous row from this lane's quarter of the state ----
        const float *xin = xf + cur * MT + kq * KQ;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int t = 0; t < KQ; t += 4) {
            const f32x4p x = *reinterpret_cast<const f32x4p *>(xin + t);
            s0 += x.lo.x + x.lo.y; s1 += x.hi.x + x.hi.y;
        }
        float sprev = quad_sum_f(s0 + s1);
        if (j == 0) sprev = 1.0f;
        const float inv = __builtin_amdgcn_rcpf(sprev);
        const float thr = 1e-10f * sprev;
        if (j > 0) {
            float an = v_prev * inv;
            an = (i < M) ? fmaxf(an, 1e-10f) : 0.f;
            if (rerun && (j & 15) == 0) {
                const float old_pref = owner ? a.alpha[(size_t)(ch.base + ell - 1) * Mp + i] : 0.f;
                const bool bad = owner && i < M && !(fabsf(an - old_pref) <= a.eps_f * fabsf(old_pref));
                const bool anyb = __any(bad);
                if (lane == 0) mflag[w] = anyb ? 1 : 0;
            }
            if (owner && !PRE) a.alpha[(size_t)(ch.base + ell - 1) * Mp + i] = an;
            if (tid == 0 && !PRE) a.cnorm[ch.base + ell - 1] = (double)sprev;
        }
        float vout;
        if (ge < 0) {
            const double e_cur = a.E[(size_t)kid * Mp + i];
            const float *q = qa.qTf + qoff;
            const float dot = stream_dot<KQ, FB>(q, QS, rotf, [&](int t) { return fmaxf(xin[t], thr); });
            const float y = quad_sum_f(dot) * inv;
            vout = (i < M) ? (float)((double)y * e_cur) : 0.f;
        } else if (PRE) {
            const int es = SMCPP_ES(ge);
            const int sp = a.g_span[SMCPP_GID(ge)];
            float outv = 0.f;
            int napp = 0;
            for (int b = 0; b < a.nbits; ++b) {
                if (!((sp >> b) & 1)) continue;
                const float *q = qa.qBf + ((size_t)es * a.nbits + b) * Mp * Mp + qoff;
                float dot;
                if (napp == 0) dot = stream_dot<KQ, FB>(q, QS, rotf, [&](int t) { return fmaxf(xin[t], thr); });
                else {
                    float *tb = tbf + (napp & 1) * MT;
                    if (owner) tb[i] = outv;
                    lds_barrier();
                    const float *tin = tb + kq * KQ;
                    dot = stream_dot<KQ, FB>(q, QS, rotf, [&](int t) { return tin[t]; });
                }
                outv = quad_sum_f(dot);
                ++napp;
            }
            vout = (i < M) ? outv * inv : 0.f;
        } else {
            const int es = SMCPP_ES(ge);
            const double dp_cur = a.dpow[(size_t)SMCPP_GID(ge) * Mp + i];
            const double *q1 = qa.qPinvT + (size_t)es * Mp * Mp + qoff;
            double u = quad_sum_d(stream_dot<KQ, DB>(q1, QS, rotd, [&](int t) { return (double)fmaxf(xin[t], thr); }));
            u = u * dp_cur * (double)inv;
            if (owner) ub[(i / KQ) * UP + (i % KQ)] = (i < M) ? u : 0.0;
            lds_barrier();
            const double *uin = ub + kq * UP;
            const double *q2 = qa.qPT + (size_t)es * Mp * Mp + qoff;
            const double av = quad_sum_d(stream_dot<KQ, DB>(q2, QS, rotd, [&](int t) { return uin[t]; }));
            vout = (i < M) ? (float)av : 0.f;
        }
        if (owner) xf[nxt * MT + i] = vout;
        v_prev = vout;
        lds_barrier();
    }
    if (merged) {
        if (owner) end_cur[i] = end_prev[i];
        return;
    }
    {
        const float *xin = xf + (nrows & 1) * MT + kq * KQ;
        float sl = 0.f;
#pragma unroll
        for (int t = 0; t < KQ; ++t) sl += xin[t];
        const float sprev = quad_sum_f(sl);
        const float inv = __builtin_amdgcn_rcpf(sprev);
        if (owner) {
            float an = v_prev * inv;
            an = (i < M) ? fmaxf(an, 1e-10f) : 0.f;
            if (!PRE) a.alpha[(size_t)(ch.base + ch.r1) * Mp + i] = an;
            end_cur[i] = an;
        }
        if (tid == 0 && !PRE) a.cnorm[ch.base + ch.r1] = (double)sprev;
    }
}

template <int MT, bool PRE = false>
__global__ __launch_bounds__(MT * 4) void k_bwd_big(ChainArgs a, BigArgs qa) {
    constexpr int NW = MT / 16, KQ = MT / 4, Mp = MT, UP = KQ + 2, QS = 4 * MT;
    constexpr int DB = (MT > 128) ? 8 : 16;
    __shared__ __attribute__((aligned(16))) double ub[4 * UP];
    __shared__ __attribute__((aligned(16))) double xb[2 * 4 * UP];
    __shared__ __attribute__((aligned(16))) double tbd[PRE ? 2 * 4 * UP : 2];
    __shared__ int2 sdesc[128];
    __shared__ int sflag;
    __shared__ int mflag[NW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int il = lane >> 2, kq = lane & 3, i = 16 * w + il;
    const bool owner = kq == 0;
    const int M = a.M, pass = a.pass, c = blockIdx.x;
    const bool full = a.variant == 2;
    if (pass > 0 && a.changed[pass - 1] == 0) return;
    const Chunk ch = a.chunks[c];
    double *end_cur = a.ends_b + ((size_t)(pass & 1) * a.nchunks + c) * Mp;
    const double *end_prev = a.ends_b + ((size_t)((pass + 1) & 1) * a.nchunks + c) * Mp;
    if (pass > 0 && !full && ch.last) {
        if (owner) end_cur[i] = end_prev[i];
        return;
    }
    double b = 0.0;
    {
        const bool fresh = ch.last || pass == 0;
        const double *src = a.ends_b + ((size_t)((pass + 1) & 1) * a.nchunks + (fresh ? c : c + 1)) * Mp;
        if (i < M) b = fresh ? 1.0 / (double)M : src[i];
    }
    if (tid == 0) sflag = 0;
    if (lane == 0) mflag[w] = 1;
    __syncthreads();
    if (pass > 0 && !full) {
        bool diff = false;
        if (owner && i < M) {
            const double u = a.used_b[(size_t)c * Mp + i];
            if (!(fabs(b - u) <= a.eps_b * fabs(u))) diff = true;
        }
        if (__any(diff) && lane == 0) sflag = 1;
        __syncthreads();
        if (sflag == 0) {
            if (owner) end_cur[i] = end_prev[i];
            return;
        }
    }
    if (owner) a.used_b[(size_t)c * Mp + i] = b;
    if (tid == 0) a.changed[pass] = 1;
    const int2 *rd = a.rowdesc + ch.base;
    const int nrows = ch.r1 - ch.r0;
    if (w == 0) {
        sdesc[lane] = (lane < nrows) ? rd[ch.r1 - lane] : make_int2(0, -1);
        sdesc[64 + lane] = (lane + 64 < nrows) ? rd[ch.r1 - lane - 64] : make_int2(0, -1);
    }
    if (owner) xb[(i / KQ) * UP + (i % KQ)] = b;
    __syncthreads();
    const size_t qoff = (size_t)4 * i + kq;
    constexpr int DBB = (KQ % DB == 0) ? DB : 4;
    const int rotd = (int)((blockIdx.x * 7u) % (unsigned)(KQ / DBB)) * DBB;
    double b_raw = b;
    const bool rerun = pass > 0 && !full;
    bool merged = false;
    for (int j = 0; j < nrows; ++j) {
        const int ell = ch.r1 - j;
        const int jb = j & 63, bsel = (j >> 6) & 1, cur = j & 1, nxt = cur ^ 1;
        if (rerun && j > 16 && (j & 15) == 1) {
            int nm = mflag[0];
#pragma unroll
            for (int q = 1; q < NW; ++q) nm |= mflag[q];
            if (nm == 0) { merged = true; break; }
        }
        if (w == 0 && jb == 32 && j >= 64) {
            const int2 dn = (j + 32 + lane < nrows) ? rd[ch.r1 - (j + 32 + lane)] : make_int2(0, -1);
            sdesc[(bsel ^ 1) * 64 + lane] = dn;
        }
        const int2 d0 = sdesc[bsel * 64 + jb];
        const int kid = __builtin_amdgcn_readfirstlane(d0.x);
        const int ge = __builtin_amdgcn_readfirstlane(d0.y);
        const double *xin = xb + cur * 4 * UP + kq * UP;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int t = 0; t < KQ; t += 2) {
            const double2 v = *reinterpret_cast<const double2 *>(xin + t);
            s0 += v.x; s1 += v.y;
        }
        double sprev = quad_sum_d(s0 + s1);
        if (j == 0) sprev = 1.0;
        const double inv = rcp_f64(sprev);
        {
            const double bnrm = (i < M) ? b_raw * inv : 0.0;
            if (rerun && (j & 15) == 0 && j > 0) {
                const double old_pref = owner ? a.beta[(size_t)(ch.base + ell) * Mp + i] : 0.0;
                const bool bad = owner && i < M && !(fabs(bnrm - old_pref) <= a.eps_b * fabs(old_pref));
                const bool anyb = __any(bad);
                if (lane == 0) mflag[w] = anyb ? 1 : 0;
            }
            if (owner && !PRE) a.beta[(size_t)(ch.base + ell) * Mp + i] = bnrm;
        }
        double bn;
        if (ge < 0) {
            // beta <- T (e o beta): this lane needs e on its quarter of the inner index
            const double *eq = a.E + (size_t)kid * Mp + kq * KQ;
            const double *q = qa.qTdT + qoff;
            bn = quad_sum_d(stream_dot<KQ, DB>(q, QS, rotd, [&](int t) { return eq[t] * xin[t]; })) * inv;
        } else if (PRE) {
            const int es = SMCPP_ES(ge);
            const int sp = a.g_span[SMCPP_GID(ge)];
            double outv = 0.0;
            int napp = 0;
            for (int b = 0; b < a.nbits; ++b) {
                if (!((sp >> b) & 1)) continue;
                const float *q = qa.qBb + ((size_t)es * a.nbits + b) * Mp * Mp + qoff;
                double dot;
                if (napp == 0) dot = stream_dot<KQ, DB>(q, QS, rotd, [&](int t) { return xin[t]; });
                else {
                    double *tb = tbd + (napp & 1) * 4 * UP;
                    if (owner) tb[(i / KQ) * UP + (i % KQ)] = outv;
                    lds_barrier();
                    const double *tin = tb + kq * UP;
                    dot = stream_dot<KQ, DB>(q, QS, rotd, [&](int t) { return tin[t]; });
                }
                outv = quad_sum_d(dot);
                ++napp;
            }
            bn = outv * inv;
        } else {
            const int es = SMCPP_ES(ge);
            const double dp_cur = a.dpow[(size_t)SMCPP_GID(ge) * Mp + i];
            const double *q1 = qa.qPrm + (size_t)es * Mp * Mp + qoff;
            double wv = quad_sum_d(stream_dot<KQ, DB>(q1, QS, rotd, [&](int t) { return xin[t]; }));
            wv = wv * dp_cur * inv;
            if (owner) ub[(i / KQ) * UP + (i % KQ)] = (i < M) ? wv : 0.0;
            lds_barrier();
            const double *uin = ub + kq * UP;
            const double *q2 = qa.qPinvrm + (size_t)es * Mp * Mp + qoff;
            bn = quad_sum_d(stream_dot<KQ, DB>(q2, QS, rotd, [&](int t) { return uin[t]; }));
        }
        if (!(i < M)) bn = 0.0;
        if (owner) xb[nxt * 4 * UP + (i / KQ) * UP + (i % KQ)] = bn;
        b_raw = bn;
        lds_barrier();
    }
    if (merged) {
        if (owner) end_cur[i] = end_prev[i];
        return;
    }
    {
        const double *xin = xb + (nrows & 1) * 4 * UP + kq * UP;
        double sl = 0.0;
#pragma unroll
        for (int t = 0; t < KQ; ++t) sl += xin[t];
        const double sprev = quad_sum_d(sl);
        if (owner) {
            const double bf = (i < M) ? b_raw / sprev : 0.0;
            end_cur[i] = bf;
            if (ch.first && !PRE) a.beta[(size_t)ch.base * Mp + i] = bf;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Operands of the eigen-free pre-pass for 64 < M <= 256, built on the device from the row-major transition matrix and
// the emission table right after they are uploaded (the host is busy with the eigenproblems): the two streaming layouts
// of T, A_e = diag(e) T^T per eigen key, its squares A^2 .. A^16 (fp64 MFMA, one 16 x 16 tile per wavefront) and the
// float streaming layouts of all five powers.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_big_tq(int Mp, const double *__restrict__ Td, float *__restrict__ qTf,
                                                 double *__restrict__ qTdT) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= Mp * Mp) return;
    const int KQ = Mp / 4, q = d & 3, i = (d >> 2) % Mp, t = (d >> 2) / Mp, k = q * KQ + t;
    qTf[d] = (float)Td[(size_t)k * Mp + i];          // Tf[k][i]
    qTdT[d] = Td[(size_t)i * Mp + k];                 // TdT[k][i] = T[i][k]
}

__global__ __launch_bounds__(256) void k_pow_init(int M, int Mp, int npw, const int *__restrict__ e_kid,
                                                   const double *__restrict__ E, const double *__restrict__ Td,
                                                   double *__restrict__ W) {
    const int idx = blockIdx.x * 256 + threadIdx.x, e = blockIdx.y;
    if (idx >= Mp * Mp) return;
    const int i = idx / Mp, k = idx % Mp;
    const double ev = E[(size_t)e_kid[e] * Mp + i];
    W[(size_t)e * npw * Mp * Mp + idx] = (i < M && k < M) ? ev * Td[(size_t)k * Mp + i] : 0.0;     // A[i][k] = e_i T[k][i]
}

// Powers from A^32 on (spans of 32 and more) are rescaled to max |entry| = 1 before they are squared again: with |lambda| < 1
// they would leave the float range of the streaming layouts, and the pre-pass only needs directions.  One workgroup per matrix.
__global__ __launch_bounds__(256) void k_pow_rescale(int Mp, double *__restrict__ Wb, size_t stride) {
    __shared__ double smax[4];
    double *P = Wb + blockIdx.x * stride;
    const int tid = threadIdx.x;
    double mx = 0.0;
    for (int idx = tid; idx < Mp * Mp; idx += 256) mx = fmax(mx, fabs(P[idx]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) smax[tid >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    const double sc = mx > 0.0 ? 1.0 / mx : 1.0;
    for (int idx = tid; idx < Mp * Mp; idx += 256) P[idx] *= sc;
}

// dst = src * src, row-major [Mp][Mp]; grid (Mp/16, Mp/16, Ke), one wavefront per 16 x 16 tile; `stride` = doubles between
// the matrices of consecutive eigen keys
__global__ __launch_bounds__(64) void k_sq_f64(int Mp, const double *__restrict__ src, double *__restrict__ dst, size_t stride) {
    const int lane = threadIdx.x, m = lane & 15, qd = lane >> 4;
    const double *S = src + blockIdx.z * stride;
    double *D = dst + blockIdx.z * stride;
    const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
    f64x4 acc = {0, 0, 0, 0};
    for (int kk = 0; kk < Mp / 4; ++kk) {
        const double av = S[(size_t)(r0 + m) * Mp + 4 * kk + qd];      // A[m][k]
        const double bv = S[(size_t)(4 * kk + qd) * Mp + c0 + m];      // B[k][n]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) D[(size_t)(r0 + qd + 4 * r) * Mp + c0 + m] = acc[r];     // D[row = qd + 4r][col = m]
}

__global__ __launch_bounds__(256) void k_pow_layout(int Mp, const double *__restrict__ W, float *__restrict__ qBf,
                                                     float *__restrict__ qBb) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= Mp * Mp) return;
    const size_t mo = (size_t)blockIdx.y * Mp * Mp;          // blockIdx.y = e * (powers per key) + b
    const int KQ = Mp / 4, q = d & 3, i = (d >> 2) % Mp, t = (d >> 2) / Mp, k = q * KQ + t;
    qBf[mo + d] = (float)W[mo + (size_t)i * Mp + k];
    qBb[mo + d] = (float)W[mo + (size_t)k * Mp + i];
}

// Eigenvalue powers of every (span, eigen key) group: dpow[g][i] = (d_r[i] / scale)^span   (transition_bundle.h:9-30 keeps
// d_scaled; hmm.cpp:72,112 raise it to the span of the row)
__global__ __launch_bounds__(256) void k_group_dpow(int G, int M, int Mp, const int *__restrict__ g_span,
                                                     const int *__restrict__ g_eig, const double *__restrict__ dsc,
                                                     double *__restrict__ dpow) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)G * Mp) return;
    const int g = (int)(idx / Mp), i = (int)(idx % Mp);
    dpow[idx] = (i < M) ? pow(dsc[(size_t)g_eig[g] * Mp + i], (double)g_span[g]) : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------
// K7: log-likelihood  ll = sum_ell log c_ell + span_ell log scale   (hmm.cpp:79,88,95)
// ---------------------------------------------------------------------------------------------------------------
struct LoglikArgs {
    const double *cnorm;
    const RowInfo *rowinfo;
    const double *g_logscale;     // [G] span * log(scale)
    const long long *contig_base; // [n_contigs]
    const int *contig_L;          // [n_contigs]
    double *partial;              // [n_contigs][nblk]
    double *loglik;               // [n_contigs]
    double *loglik_host;          // optional: device view of a pinned host array that receives the same values (no copy afterwards)
    double *logc;                 // optional [rows]: log_c per row (needed by the span-1 weights)
    int nblk;
};

__global__ __launch_bounds__(256) void k_loglik_partial(LoglikArgs a) {
    __shared__ double red[256];
    const int ct = blockIdx.y;
    const long long base = a.contig_base[ct];
    const int L = a.contig_L[ct];
    const int per = (L + a.nblk - 1) / a.nblk;
    const int lo = 1 + blockIdx.x * per;
    const int hi = min(L, lo + per - 1);
    double s = 0.0;
    for (int ell = lo + threadIdx.x; ell <= hi; ell += 256) {
        const RowInfo ri = a.rowinfo[base + ell];
        double lc = log(a.cnorm[base + ell]);
        if (ri.gid >= 0) lc += a.g_logscale[ri.gid];
        a.logc[base + ell] = lc;
        s += lc;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.partial[(size_t)ct * a.nblk + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void k_loglik_final(LoglikArgs a) {
    __shared__ double red[256];
    const int ct = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < a.nblk; i += 256) s += a.partial[(size_t)ct * a.nblk + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.loglik[ct] = red[0];
        if (a.loglik_host) a.loglik_host[ct] = red[0];
    }
}

// Completion signal into pinned host memory: the host polls the word instead of blocking in hipStreamSynchronize (the stream is
// in-order, so everything queued before this launch has finished and released its writes when the value arrives)
__global__ void k_signal(int *flag, int value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... and one scalar with it (the all-reduced sum of the log-likelihoods, first entry of the packed statistics)
__global__ void k_publish_scalar(const double *src, double *host_val, int *flag, int value) {
    *host_val = *src;
    __threadfence_system();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------------------------
// K3: span-1 rows — per-row scalars and gamma sums  (hmm.cpp:134-138,146-148)
//   v = alpha_ell o beta_ell / p,  p = sum(alpha_ell o beta_ell);  w1 = 1 / (c_ell p)
// One wavefront walks a slab of rows of one (contig, key) segment and keeps the running sum of v in registers.
// ---------------------------------------------------------------------------------------------------------------
struct Slab {
    int start, end;    // range in the sorted row permutation
    int bucket;        // accumulator the slab contributes to
    int aux;           // span-1: kid; eigen: gid
    long long base;    // contig base row
};

struct S1Args {
    int M, Mp, nslabs;
    const Slab *slabs;
    const int *perm;          // sorted span-1 rows (contig-relative ell)
    const float *alpha;
    const double *beta;
    const double *cnorm;      // [rows] forward normaliser c_ell of every row (span-1 rows carry no eigenvalue scale)
    double *w1;               // [rows] per-row weight
    double *gpart;            // [nslabs][Mp] partial gamma sums
    double *gamma_rows;       // optional [rows][Mp] (save_gamma)
    int only_w1;              // 1: weights only (the eigen-free statistics of span > 1 rows take their gamma from the span fold)
};

template <int NPL>
__global__ __launch_bounds__(256) void k_s1_scalars(S1Args a) {
    __shared__ double comb[4][NPL * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Slab sl = a.slabs[blockIdx.x];
    const int M = a.M, Mp = a.Mp;
    double gs[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) gs[q] = 0.0;
    // wavefront w takes rows start + 4*(4*it + u) + w: four independent rows in flight per iteration.  All loads are
    // unconditional (row index clamped into the slab, padded states read the zeros the chains left there) and the row
    // indices of the NEXT iteration are fetched while this one is reduced: one memory round trip per iteration instead
    // of three (index -> operands -> log_c), and no exec-mask branch around a load (see k_rank_acc).
    const int last = sl.end - 1;
    int elln[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) elln[u] = a.perm[min(sl.start + wave + 4 * u, last)];
    for (int r0 = sl.start + wave; r0 < sl.end; r0 += 16) {
        double v[4][NPL], lc[4];
        size_t row[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ok[u] = r0 + 4 * u < sl.end;
            row[u] = (size_t)(sl.base + elln[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) elln[u] = a.perm[min(r0 + 16 + 4 * u, last)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lc[u] = a.cnorm[row[u]];
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int i = min(lane + 64 * q, Mp - 1);
                v[u][q] = (double)a.alpha[row[u] * Mp + i] * a.beta[row[u] * Mp + i];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double part = 0.0;
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                if (lane + 64 * q >= M) v[u][q] = 0.0;
                part += v[u][q];
            }
            const double p = wave_sum_dpp(part);
            const double ip = 1.0 / p;
            if (!ok[u]) continue;                                   // wave-uniform
            if (a.gamma_rows) {                                     // (save_gamma: the stored posterior keeps the division)
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const double g = v[u][q] / p;
                    gs[q] += g;
                    const int i = lane + 64 * q;
                    if (i < Mp) a.gamma_rows[row[u] * Mp + i] = g;
                }
            } else if (!a.only_w1) {                                // (weights only: no gamma sums - the divisions were most of this kernel's instructions)
#pragma unroll
                for (int q = 0; q < NPL; ++q) gs[q] = fma(v[u][q], ip, gs[q]);
            }
            if (lane == 0) a.w1[row[u]] = ip / lc[u];
        }
    }
    if (a.only_w1) return;
#pragma unroll
    for (int q = 0; q < NPL; ++q) comb[wave][lane + 64 * q] = gs[q];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int i = lane + 64 * q;
            if (i < Mp) a.gpart[(size_t)blockIdx.x * Mp + i] = ((comb[0][i] + comb[1][i]) + comb[2][i]) + comb[3][i];
        }
    }
}

// Deterministic reduction of per-slab partials: out[(b*ZS + z)][len] = sum over the z-th share of bucket b's slabs.
// One thread per output element, four independent partial sums over the slabs of its share (fixed order, no atomics):
// every wavefront reads 512 contiguous bytes of each slab.  grid = (ceil(len/256), buckets, ZS).
__global__ __launch_bounds__(256) void k_sum_parts(const double *__restrict__ part, const int *__restrict__ off,
                                                   double *__restrict__ out, int len, int ZS) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y, z = blockIdx.z;
    const int s0 = off[b], s1 = off[b + 1];
    const int per = (s1 - s0 + ZS - 1) / ZS;
    const int lo = s0 + z * per, hi = min(s1, lo + per);
    if (idx >= len) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int s = lo;
    for (; s + 3 < hi; s += 4) {
        a0 += part[(size_t)s * len + idx];
        a1 += part[(size_t)(s + 1) * len + idx];
        a2 += part[(size_t)(s + 2) * len + idx];
        a3 += part[(size_t)(s + 3) * len + idx];
    }
    for (; s < hi; ++s) a0 += part[(size_t)s * len + idx];
    out[((size_t)b * ZS + z) * len + idx] = (a0 + a1) + (a2 + a3);
}

// ---------------------------------------------------------------------------------------------------------------
// K5: eigen rows — U = Pinv alpha_{ell-1}, W = P^T beta_ell, omega = 1 / (scale * sum_j d~_j^span U_j W_j)
// (the exact normaliser of hmm.cpp:116-122) with fp64 MFMA; writes omega*U and W, 16 rows per
// wavefront iteration.  MFMA f64 16x16x4 operand map: A[m = l&15][k = l>>4], B[k = l>>4][n = l&15],
// D[row = (l>>4) + 4*reg][col = l&15].
// ---------------------------------------------------------------------------------------------------------------
struct UWArgs {
    int M, Mp, nslabs;
    const Slab *slabs;
    const int *perm;          // sorted eigen rows (contig-relative ell)
    const float *alpha;
    const double *beta;
    const int *g_eig;
    const double *g_scale;    // [G] eigen scale of the group's key
    const double *dpow;       // [G][Mp]
    const double *PinvT;      // [Ke][Mp][Mp]
    const double *Prm;        // [Ke][Mp][Mp]
    double *Xs;               // [n eigen rows][Mp]  omega * U   (indexed by position in perm)
    double *Ys;               // [n eigen rows][Mp]  W
    const int *pos_gid;       // [n eigen rows] group of every sorted position (k_eig_fused2: slabs that mix groups)
    const int *g_span;        // [G]
};

template <int NT>   // NT = Mp / 16 state tiles
__global__ __launch_bounds__(64) void k_eig_uw(UWArgs a) {
    const int lane = threadIdx.x;
    const int m = lane & 15, qd = lane >> 4;
    const Slab sl = a.slabs[blockIdx.x];
    const int Mp = a.Mp;
    const int es = a.g_eig[sl.aux];
    const double *PinvT = a.PinvT + (size_t)es * Mp * Mp;
    const double *Prm = a.Prm + (size_t)es * Mp * Mp;
    const double *dp = a.dpow + (size_t)sl.aux * Mp;
    const double scale = a.g_scale[sl.aux];
    for (int r0 = sl.start; r0 < sl.end; r0 += 16) {
        // A operands: data rows.  lane (m, qd) feeds row r0+m, state 4*kk+qd.
        const int ra = r0 + m;
        const bool va = ra < sl.end;
        const int ell_a = va ? a.perm[ra] : 1;
        const float *arow = a.alpha + (size_t)(sl.base + ell_a - 1) * Mp;   // alpha_{ell-1}
        const double *brow = a.beta + (size_t)(sl.base + ell_a) * Mp;       // beta_ell
        f64x4 U[NT], W[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { U[t] = (f64x4){0, 0, 0, 0}; W[t] = (f64x4){0, 0, 0, 0}; }
        for (int kk = 0; kk < Mp / 4; ++kk) {
            const int st = 4 * kk + qd;
            const double av = va ? (double)arow[st] : 0.0;
            const double bv = va ? brow[st] : 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double pinv = PinvT[(size_t)st * Mp + 16 * t + m];   // B[k=st][n=16t+m] = Pinv[16t+m][st]
                const double pp = Prm[(size_t)st * Mp + 16 * t + m];       // B[k=st][n=16t+m] = P[st][16t+m]
                U[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, pinv, U[t], 0, 0, 0);
                W[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, pp, W[t], 0, 0, 0);
            }
        }
        // D layout: lane (m, qd), reg r holds [data row r0 + qd + 4r][state 16t + m]
        double om[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double part = 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) part += dp[16 * t + m] * U[t][r] * W[t][r];
            const double s = row16_sum(part);
            om[r] = 1.0 / (scale * s);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = r0 + qd + 4 * r;
            if (rr < sl.end) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    a.Xs[(size_t)rr * Mp + 16 * t + m] = om[r] * U[t][r];
                    a.Ys[(size_t)rr * Mp + 16 * t + m] = W[t][r];
                }
            }
        }
    }
}


// K5+K4 fused, generation 2 (round 3): slabs that MIX span groups.  The span-Q weighting that follows the accumulation,
// Z = sum_g S_g o Acc_g with S_g[j][k] = (p_j - p_k) / (d_j - d_k), p = (d / scale)^span of the group, is linear in the group's
// powers, so it moves inside the sum over rows:
//     Z[j][k] = 1 / (d_j - d_k) * sum_rows [ (p_j omega u_j) w_k - (omega u_j) (p_k w_k) ]        (j != k)
//     Z[j][j] = 1 / d_j * sum_rows span p_j omega u_j w_j
// - two rank updates into ONE accumulator and one vector, whatever the spans of the rows are.  Un-binned data (posterior
// decoding) have 10^5 groups of a handful of rows each: generation 1 wrote and re-read one M x M partial per GROUP (800 MB
// each way on the posterior workload, and a [groups][8][M][M] reduction buffer), this one writes one per 128-row slab.
// aux of a slab = its eigen key; the host pads the slab list so that a workgroup never mixes keys.
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_eig_fused2(UWArgs a, double *part) {
    constexpr int KQ = 4 * NT;                 // states per lane of the k dimension: lane (m, qd) owns KQ*qd .. +KQ-1
    constexpr int MT = 16 * NT;
    constexpr int LD = MT + 1;
    extern __shared__ double eig_lds[];        // [2][MT][LD]: Pinv^T and P of the eigen key of this block's first slab
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = lane & 15, qd = lane >> 4;
    const int Mp = a.Mp;
    const int slab0 = blockIdx.x * 4;
    const int es0 = a.slabs[slab0].aux;
    {
        const double *g0 = a.PinvT + (size_t)es0 * Mp * Mp, *g1 = a.Prm + (size_t)es0 * Mp * Mp;
        for (int idx = threadIdx.x; idx < MT * MT; idx += 256) {
            const int r = idx / MT, c = idx % MT;
            eig_lds[r * LD + c] = g0[(size_t)r * Mp + c];
            eig_lds[MT * LD + r * LD + c] = g1[(size_t)r * Mp + c];
        }
    }
    __syncthreads();
    const int slab = slab0 + wv;
    if (slab >= a.nslabs) return;
    const Slab sl = a.slabs[slab];
    const double *sPinvT = eig_lds, *sPrm = eig_lds + MT * LD;
    f64x4 acc[NT][NT];
    double dgacc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        dgacc[i] = 0.0;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f64x4){0, 0, 0, 0};
    }
    double scale = 1.0;
    if (sl.start < sl.end) scale = a.g_scale[a.pos_gid[sl.start]];          // the eigen scale belongs to the key
    for (int r0 = sl.start; r0 < sl.end; r0 += 16) {
        const int ra = min(r0 + m, sl.end - 1);
        const int ell_a = a.perm[ra];
        // D layout below: lane (m, qd), register r holds data row r0 + qd + 4 r: its group, span and powers
        int gr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gr[r] = a.pos_gid[min(r0 + qd + 4 * r, sl.end - 1)];
        const float4 *arow = reinterpret_cast<const float4 *>(a.alpha + (size_t)(sl.base + ell_a - 1) * Mp + KQ * qd);
        const double2 *brow = reinterpret_cast<const double2 *>(a.beta + (size_t)(sl.base + ell_a) * Mp + KQ * qd);
        double dp[4][NT], spn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            spn[r] = (double)a.g_span[gr[r]];
#pragma unroll
            for (int t = 0; t < NT; ++t) dp[r][t] = a.dpow[(size_t)gr[r] * Mp + 16 * t + m];
        }
        f64x4 U[NT], W[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { U[t] = (f64x4){0, 0, 0, 0}; W[t] = (f64x4){0, 0, 0, 0}; }
        float4 a4 = arow[0];
        double2 b01 = brow[0], b23 = brow[1];
#pragma unroll 1
        for (int t4 = 0; t4 < NT; ++t4) {
            const float avv[4] = {a4.x, a4.y, a4.z, a4.w};
            const double bvv[4] = {b01.x, b01.y, b23.x, b23.y};
            {
                const int tn = min(t4 + 1, NT - 1);
                a4 = arow[tn];
                b01 = brow[2 * tn];
                b23 = brow[2 * tn + 1];
            }
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int st = KQ * qd + 4 * t4 + k4;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const double pinv = sPinvT[st * LD + 16 * t + m];
                    const double pp = sPrm[st * LD + 16 * t + m];
                    U[t] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)avv[k4], pinv, U[t], 0, 0, 0);
                    W[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(bvv[k4], pp, W[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double pr = 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) pr += dp[r][t] * U[t][r] * W[t][r];
            const double sm = row16_sum(pr);
            const bool vr = r0 + qd + 4 * r < sl.end;        // padded rows must not contribute
            const double om = vr ? 1.0 / (scale * sm) : 0.0;
            double xa[NT], xp[NT], yp[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                xa[t] = om * U[t][r];
                xp[t] = xa[t] * dp[r][t];
                yp[t] = W[t][r] * dp[r][t];
                dgacc[t] = fma(spn[r] * xp[t], W[t][r], dgacc[t]);
                xa[t] = -xa[t];
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xp[i], W[j][r], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[i], yp[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    // partial of the slab: [Mp][Mp] accumulator followed by the [Mp] diagonal sums (ONE reduction pass takes both)
    double *out = part + (size_t)slab * ((size_t)Mp * Mp + Mp);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                out[(size_t)(16 * i + qd + 4 * rg) * Mp + 16 * j + m] = acc[i][j][rg];
    // the diagonal sums: over the four lane groups (data rows qd + 4 r of every tile)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        double v = dgacc[t];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (qd == 0) out[(size_t)Mp * Mp + 16 * t + m] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K4: rank-k accumulation  C_slab[j][k] = sum_rows X_row[j] * Y_row[k]   (fp64 MFMA, 64x64 output block per wave)
//   MODE 0 (span-1 rows, hmm.cpp:137-138):  X = w1 * alpha_{ell-1},  Y = beta_ell o e_key
//   MODE 1 (eigen rows):                    X = Xs[pos] (= omega U), Y = Ys[pos] (= W)
//   MODE 2 (span > 1 rows, eigen-free):     X = w1 * alpha_{ell-1},  Y = beta_ell   (k_span_fold expands the span afterwards)
//   MODE 3 (span-1 rows, M <= 64, slabs of ONE key in key-sorted order): MODE 0 plus the key's gamma sums
//           sum_rows alpha_ell o beta_ell / p_ell (hmm.cpp:134-136,146-148) from the operands the weight is formed from anyway -
//           k_s1_scalars (a second pass over alpha and beta of every span-1 row) does not run
// grid = (nslabs, NB*NB) with NB = ceil(Mp/64); blockIdx.y selects the 64x64 block of the M x M output.
// ---------------------------------------------------------------------------------------------------------------
struct AccArgs {
    int M, Mp, nslabs, NB;
    const Slab *slabs;
    const int *perm;
    const int2 *permk;        // MODE 0: {ell, key id} per sorted span-1 row
    const RowInfo *rowinfo;
    const float *alpha;
    const double *beta;
    const double *w1;
    const double *cnorm;      // MODE 2, M <= 64: the weight is formed in the kernel
    const double *E;
    const double *Xs, *Ys;
    double *part;             // [nslabs][Mp][Mp]   (TEAM: [nteams][Mp][Mp])
    double *gpart;            // MODE 3: [nslabs][Mp] gamma sums of the slab's key
    const int2 *teams;        // TEAM: {first slab, slabs (1..4)} of a workgroup - slabs of ONE reduction range
};

// TEAM (round 5): a workgroup of four wavefronts takes up to four consecutive slabs of one reduction range, adds the four accumulators
// through LDS (fixed order: (w0 + w2) + (w1 + w3)) and writes ONE partial - a quarter of the partial bytes written here and read back by
// k_sum_parts (headline: 60 + 64 MB of 390 per E-step).  grid.x = teams.
template <int MODE, bool TEAM = false>
__global__ __launch_bounds__(TEAM ? 256 : 64) __attribute__((amdgpu_waves_per_eu(2))) void k_rank_acc(AccArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = lane & 15, qd = lane >> 4;
    int slab_id = blockIdx.x;
    bool have = true;
    if (TEAM) {
        const int2 tm = a.teams[blockIdx.x];
        const int wu = __builtin_amdgcn_readfirstlane(wv);          // (scalar: the slab and everything derived from it stay uniform)
        have = wu < tm.y;
        slab_id = tm.x + (have ? wu : 0);
    }
    const Slab sl = a.slabs[slab_id];
    const int Mp = a.Mp;
    const int jb = (blockIdx.y / a.NB) * 64, kb = (blockIdx.y % a.NB) * 64;
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0, 0, 0, 0};
    if (TEAM && !have) {
        // a wavefront without a slab only takes part in the reduction (with zeros)
    } else if (MODE == 0 || MODE == 2 || MODE == 3) {
        // Software pipeline over groups of 4 rows (the MFMA k dimension): the {ell, key} pair is fetched two groups
        // ahead and the operands one group ahead, so the dependent chain  index -> row -> operands  (three memory
        // round trips, measured 4.8 us per group against 0.43 us of MFMA) no longer serialises every group.
        // every load is unconditional (indices clamped into the slab / the matrix, results masked afterwards):
        // a load under an exec-mask branch makes the compiler fall back to s_waitcnt vmcnt(0), which would drain the
        // prefetched group as well
        const int last = sl.end - 1;
        auto fetch_pk = [&](int r0) {
            const int r = r0 + qd;
            int2 pk;
            if (MODE == 0) pk = a.permk[min(r, last)];
            else { pk.x = a.perm[min(r, last)]; pk.y = MODE == 3 ? sl.aux : 0; }
            pk.x = (r <= last) ? pk.x : -1;
            return pk;
        };
        struct Ops { double w; float ap[4], an[4]; double bp[4], ep[4]; bool valid; };
        // (round 5) COLUMN MAP of these modes: tile t of lane m is column 4 m + t of the block (not 16 t + m): a lane's four columns are
        // consecutive, so a row's operands are ONE 16-byte load of alpha and two of beta per lane (four and four before: the rank
        // updates were issuing 13 - 21 loads per group of four rows), and a row of the block is read as one contiguous 256 / 512 bytes.
        // Which products meet in which accumulator entry is unchanged - only where the entry is written (the store below maps back).
        int jc[4], kc[4];
        bool jv[4], kv[4];
        const int j0 = jb + 4 * m, k0 = kb + 4 * m;
        const int j0c = j0 < Mp ? j0 : 0, k0c = k0 < Mp ? k0 : 0;     // (Mp is a multiple of 16: a lane's four columns are in or out together)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            jv[t] = j0 < Mp; kv[t] = k0 < Mp;
            jc[t] = j0c + t; kc[t] = k0c + t;
        }
        double ekc[4] = {1.0, 1.0, 1.0, 1.0}, gsum[4] = {0.0, 0.0, 0.0, 0.0};
        if (MODE == 3) {
#pragma unroll
            for (int t = 0; t < 4; ++t) ekc[t] = a.E[(size_t)sl.aux * Mp + kc[t]];
        }
        auto fetch_ops = [&](const int2 pk) {      // raw loads only; masking happens where the values are consumed
            Ops o;
            o.valid = pk.x >= 0;
            const size_t row = (size_t)(sl.base + (o.valid ? pk.x : 1));
            // the whole state vector in this block (M <= 64): the weight 1 / (c_ell sum alpha_ell beta_ell) is formed here from the
            // row's own alpha (4 M bytes more per row; no pass over alpha and beta has to run first)
            const bool inl = a.NB == 1;
            o.w = inl ? a.cnorm[row] : a.w1[row];
            const float *ap = a.alpha + (row - 1) * Mp;
            const double *bp = a.beta + row * Mp;
            const double *ep = a.E + (size_t)pk.y * Mp;
            const float4 a4 = *reinterpret_cast<const float4 *>(ap + j0c);
            const double4 b4 = *reinterpret_cast<const double4 *>(bp + k0c);
            o.ap[0] = a4.x; o.ap[1] = a4.y; o.ap[2] = a4.z; o.ap[3] = a4.w;
            o.bp[0] = b4.x; o.bp[1] = b4.y; o.bp[2] = b4.z; o.bp[3] = b4.w;
            if (MODE == 0) {
                const double4 e4 = *reinterpret_cast<const double4 *>(ep + k0c);
                o.ep[0] = e4.x; o.ep[1] = e4.y; o.ep[2] = e4.z; o.ep[3] = e4.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) o.ep[t] = MODE == 3 ? ekc[t] : 1.0;
            }
            if (inl) {
                const float4 n4 = *reinterpret_cast<const float4 *>(ap + Mp + k0c);
                o.an[0] = n4.x; o.an[1] = n4.y; o.an[2] = n4.z; o.an[3] = n4.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) o.an[t] = 0.f;
            }
            return o;
        };
        auto consume = [&](const Ops &cur) {
            double xa[4], yb[4];
            double wgt = cur.w;
            if (a.NB == 1) {
                double pp = 0.0;
#pragma unroll
                for (int t = 0; t < 4; ++t) pp += kv[t] ? (double)cur.an[t] * cur.bp[t] : 0.0;
                const double psum = row16_sum(pp);
                wgt = 1.0 / (cur.w * psum);
                if (MODE == 3) {
                    const double ip = 1.0 / psum;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        gsum[t] += (cur.valid && kv[t]) ? (double)cur.an[t] * cur.bp[t] * ip : 0.0;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xa[t] = (cur.valid && jv[t]) ? wgt * (double)cur.ap[t] : 0.0;
                yb[t] = kv[t] ? cur.bp[t] * cur.ep[t] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[i], yb[j], acc[i][j], 0, 0, 0);
        };
        // (round 3: TWO groups of operands in flight, descriptors three groups ahead - with one, a wavefront had 4 KB on its way
        // during a ~2 us round trip: 2.4 TB/s on the whole genome)
        int2 pk1 = fetch_pk(sl.start);
        Ops cur = fetch_ops(pk1);
        pk1 = fetch_pk(sl.start + 4);
        Ops nx1 = fetch_ops(pk1);
        pk1 = fetch_pk(sl.start + 8);
        for (int r0 = sl.start; r0 < sl.end; r0 += 4) {
            const int2 pk2 = fetch_pk(r0 + 12);
            const Ops nxt = fetch_ops(pk1);          // operands of group r0 + 8 (all-zero past the end of the slab)
            consume(cur);
            cur = nx1;
            nx1 = nxt;
            pk1 = pk2;
        }
        if (MODE == 3) {
            // lane (m, qd) summed the rows qd, qd + 4, ... of the slab at states kb + 16 t + m: fold the four lane groups
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                double v = gsum[t];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (qd == 0 && k0 < Mp) a.gpart[(size_t)slab_id * Mp + k0 + t] = v;
            }
        }
    } else {
        for (int r0 = sl.start; r0 < sl.end; r0 += 4) {
            const int r = r0 + qd;                 // this lane's data row (the MFMA k index)
            const bool valid = r < sl.end;
            double xa[4], yb[4];
            const size_t row = (size_t)(valid ? r : sl.start);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = jb + 16 * t + m, k = kb + 16 * t + m;
                xa[t] = (valid && j < Mp) ? a.Xs[row * Mp + j] : 0.0;
                yb[t] = (valid && k < Mp) ? a.Ys[row * Mp + k] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[i], yb[j], acc[i][j], 0, 0, 0);
        }
    }
    if (TEAM) {
        // lane-major slots: [register 0..63][lane] - every ds access is 64 consecutive doubles
        __shared__ double red[2][64 * 64];
        if (wv >= 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) red[wv - 2][((i * 4 + j) * 4 + rg) * 64 + lane] = acc[i][j][rg];
        }
        __syncthreads();
        if (wv < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) acc[i][j][rg] += red[wv][((i * 4 + j) * 4 + rg) * 64 + lane];
        }
        __syncthreads();
        if (wv == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) red[0][((i * 4 + j) * 4 + rg) * 64 + lane] = acc[i][j][rg];
        }
        __syncthreads();
        if (wv != 0) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) acc[i][j][rg] += red[0][((i * 4 + j) * 4 + rg) * 64 + lane];
    }
    double *out = a.part + (size_t)blockIdx.x * Mp * Mp;
    if (MODE != 1) {
        // the column map of modes 0 / 2 / 3: accumulator (tile i, D row r) is block row 4 r + i, (tile j, D column m) is block column
        // 4 m + j - a lane's four tiles j are four consecutive columns: one 32-byte store per (i, register)
        const int col = kb + 4 * m;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row = jb + 4 * (qd + 4 * rg) + i;
                if (row < Mp && col < Mp)
                    *reinterpret_cast<double4 *>(out + (size_t)row * Mp + col) = make_double4(acc[i][0][rg], acc[i][1][rg], acc[i][2][rg], acc[i][3][rg]);
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row = jb + 16 * i + qd + 4 * rg;   // D row
                const int col = kb + 16 * j + m;             // D col
                if (row < Mp && col < Mp) out[(size_t)row * Mp + col] = acc[i][j][rg];
            }
}


// ---------------------------------------------------------------------------------------------------------------
// K4w (round 6): the rank update of MODES 0 / 2 for Mp > 128 with its operand rows staged through LDS.
// k_rank_acc gives every wavefront a 64 x 64 block of the M x M output and lets it read its operands from global memory: at
// M = 256 sixteen blocks walk the same rows, alpha and beta of a row are each read four times (12 KB per row instead of 3), two
// wavefronts per SIMD wait on operand round trips, and the two rank updates of config C5 run at 45 % of the fp64 matrix peak.
// Here a workgroup of four wavefronts (one per SIMD, 512 registers each) owns a 256 (X = omega alpha_{ell-1}) x 128 (Y = beta_ell
// [o e_key]) block: 32 accumulator tiles per wavefront (its 64 X columns x the 128 Y columns).  The rows of the workgroup's slabs are
// staged 16 at a time: every thread converts / weights its share of the row (alpha: one float4, beta [and e]: one double2 per row, four
// rows per thread) into a padded LDS tile while the matrix cores work on the previous stage (register-staged double buffer, ONE
// barrier per stage); the MFMA operands are 12 ds_read_b64 per 32 v_mfma_f64_16x16x4 (2048 matrix-pipe cycles).  Per row and block
// 1 KB of alpha + 1 KB of beta: 4 KB per row at M = 256 instead of 12.
// Rows: the slabs of a team are consecutive slabs of ONE reduction range, i.e. one contiguous piece [first.start, last.end) of the
// sorted permutation on one contig - walked in order by the one workgroup (no cross-wavefront reduction; the partial is the team's).
// grid = (teams or slabs, ceil(Mp / 256) * ceil(Mp / 128)); dynamic LDS RW_LDS bytes.
// ---------------------------------------------------------------------------------------------------------------
constexpr int RW_R = 16;               // rows per stage (four MFMA k-steps)
constexpr int RW_XS = 256 + 16;        // row stride of the X tile in doubles (the pad moves consecutive rows by 32 banks)
constexpr int RW_YS = 128 + 16;
constexpr size_t RW_LDS = 2 * (size_t)RW_R * (RW_XS + RW_YS) * sizeof(double);

// NWV wavefronts per workgroup: 4 (one per SIMD, 64 X columns = 32 accumulator tiles each) or 8 (two per SIMD, 32 X columns = 16 tiles
// each: while one wavefront of a SIMD converts and stores its share of the next stage, waits at the barrier or for its LDS operands, the
// other one feeds the matrix pipe).
template <int MODE, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_rank_acc_wide(AccArgs a) {
    constexpr int XT = 16 / NWV;          // X tiles (16 columns) per wavefront
    constexpr int RPT = RW_R / NWV;       // rows of a stage per thread
    static_assert(MODE == 0 || MODE == 2, "span-1 rows (0) or the eigen-free span > 1 rows (2)");
    extern __shared__ double rw_lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int m = lane & 15, qd = lane >> 4;
    const int Mp = a.Mp;
    const int nby = (Mp + 127) / 128;
    const int jb = ((int)blockIdx.y / nby) * 256, kb = ((int)blockIdx.y % nby) * 128;
    int first = blockIdx.x, cnt = 1;
    if (a.teams) { const int2 tm = a.teams[blockIdx.x]; first = tm.x; cnt = tm.y; }
    const Slab s_first = a.slabs[first], s_last = a.slabs[first + cnt - 1];
    const int R0 = s_first.start, R1 = s_last.end;
    const long long base = s_first.base;
    auto sXb = [&](int buf) { return rw_lds + buf * (RW_R * (RW_XS + RW_YS)); };
    const bool wave_on = jb + 16 * XT * wv < Mp;
    f64x4 acc[XT][8];
#pragma unroll
    for (int i = 0; i < XT; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f64x4){0, 0, 0, 0};
    // this thread's share of a row: X columns jb + 4 lane .. + 3 (one float4 of alpha), Y columns kb + 2 lane, + 1 (one double2 of beta);
    // rows wv, wv + NWV, ... of the stage.  Columns past Mp (a multiple of 16) are loaded from column 0 and stored as zeros.
    const bool xin = jb + 4 * lane < Mp, yin = kb + 2 * lane < Mp;
    const int xcol = xin ? jb + 4 * lane : 0, ycol = yin ? kb + 2 * lane : 0;
    struct Stage { float4 al[RPT]; double2 be[RPT]; double2 em[RPT]; double w[RPT]; };
    // every load is unconditional (indices clamped, results masked where they are consumed): see k_rank_acc
    auto fetch_idx = [&](int r0, int2 (&pk)[RPT]) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = r0 + wv + NWV * i;
            const int rc = min(r, R1 - 1);
            if (MODE == 0) pk[i] = a.permk[rc];
            else { pk[i].x = a.perm[rc]; pk[i].y = 0; }
            if (r >= R1) pk[i].x = -1;
        }
    };
    auto fetch_ops = [&](const int2 (&pk)[RPT], Stage &o) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const bool valid = pk[i].x >= 0;
            const size_t row = (size_t)(base + (valid ? pk[i].x : 1));
            const double w = a.w1[row];
            o.w[i] = valid ? w : 0.0;
            o.al[i] = *reinterpret_cast<const float4 *>(a.alpha + (row - 1) * Mp + xcol);
            o.be[i] = *reinterpret_cast<const double2 *>(a.beta + row * Mp + ycol);
            if (MODE == 0) o.em[i] = *reinterpret_cast<const double2 *>(a.E + (size_t)pk[i].y * Mp + ycol);
        }
    };
    auto stash = [&](const Stage &o, int buf) {
        double *sX = sXb(buf), *sY = sXb(buf) + RW_R * RW_XS;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int rr = wv + NWV * i;
            const double w = xin ? o.w[i] : 0.0;
            double *px = sX + rr * RW_XS + 4 * lane;
            *reinterpret_cast<double2 *>(px) = make_double2(w * (double)o.al[i].x, w * (double)o.al[i].y);
            *reinterpret_cast<double2 *>(px + 2) = make_double2(w * (double)o.al[i].z, w * (double)o.al[i].w);
            double2 y = o.be[i];
            if (MODE == 0) { y.x *= o.em[i].x; y.y *= o.em[i].y; }
            if (!yin) y = make_double2(0.0, 0.0);
            *reinterpret_cast<double2 *>(sY + rr * RW_YS + 2 * lane) = y;
        }
    };
    auto compute = [&](int buf) {
        const double *sX = sXb(buf) + 16 * XT * wv + m, *sY = sXb(buf) + RW_R * RW_XS + m;
#pragma unroll
        for (int kk = 0; kk < RW_R / 4; ++kk) {
            const double *px = sX + (4 * kk + qd) * RW_XS, *py = sY + (4 * kk + qd) * RW_YS;
            double xa[XT], yb[8];
#pragma unroll
            for (int i = 0; i < XT; ++i) xa[i] = px[16 * i];
#pragma unroll
            for (int j = 0; j < 8; ++j) yb[j] = py[16 * j];
#pragma unroll
            for (int i = 0; i < XT; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[i], yb[j], acc[i][j], 0, 0, 0);
        }
    };
    if (R1 > R0) {
        const int nst = (R1 - R0 + RW_R - 1) / RW_R;
        int2 pk[RPT];
        Stage st;
        fetch_idx(R0, pk);
        fetch_ops(pk, st);
        fetch_idx(R0 + RW_R, pk);
        stash(st, 0);
        __syncthreads();
        // (measured and dropped: the two wavefronts of a SIMD in anti-phase - w + 4 storing its share of the next stage in FRONT of its
        // products, w behind them, operands loaded one iteration earlier: 336 -> 441 us (mode 0), 309 -> 319 (mode 2); the matrix pipe is
        // 69 % busy in either form - gpurun_out/r06_w3, r06_w4)
        for (int s = 0; s < nst; ++s) {
            fetch_ops(pk, st);                          // stage s + 1 (past the end: weight zero)
            fetch_idx(R0 + RW_R * (s + 2), pk);         // descriptors of stage s + 2
            if (wave_on) compute(s & 1);
            stash(st, (s + 1) & 1);
            __syncthreads();
        }
    }
    if (!wave_on) return;
    double *out = a.part + (size_t)blockIdx.x * Mp * Mp;
#pragma unroll
    for (int i = 0; i < XT; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row = jb + 16 * XT * wv + 16 * i + qd + 4 * rg;   // D row = X column
                const int col = kb + 16 * j + m;                       // D column = Y column
                if (row < Mp && col < Mp) out[(size_t)row * Mp + col] = acc[i][j][rg];
            }
}


// Packed per-rank statistics for the single all-reduce of a multi-GPU E-step, written straight into the caller's device
// buffer:  out = [ sum loglik | gamma0 (M) | xisum (M*M) | gamma-sums by GLOBAL key index (Kg*M) ],
// each summed over this rank's contigs in contig order (deterministic).  One thread per output element.
struct PackArgs {
    int M, Mp, K, Kg, n_contigs;
    const double *loglik;         // [n_contigs]
    const double *gamma0;         // [n_contigs][Mp]
    const double *xisum;          // [n_contigs][Mp][Mp]
    const double *gsum;           // [n_contigs][K][Mp]
    const unsigned char *present; // [n_contigs][K]
    const int *g2l;               // [Kg] local key id of a global key, -1 if this rank never sees it
    double *out;
};

__global__ __launch_bounds__(256) void k_pack_stats(PackArgs a) {
    const long n = 1 + a.M + (long)a.M * a.M + (long)a.Kg * a.M;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    double acc = 0.0;
    if (idx == 0) {
        for (int c = 0; c < a.n_contigs; ++c) acc += a.loglik[c];
    } else if (idx < 1 + a.M) {
        const int i = (int)idx - 1;
        for (int c = 0; c < a.n_contigs; ++c) acc += a.gamma0[(size_t)c * a.Mp + i];
    } else if (idx < 1 + a.M + (long)a.M * a.M) {
        const long e = idx - 1 - a.M;
        const int i = (int)(e / a.M), j = (int)(e % a.M);
        for (int c = 0; c < a.n_contigs; ++c) acc += a.xisum[((size_t)c * a.Mp + i) * a.Mp + j];
    } else {
        const long e = idx - 1 - a.M - (long)a.M * a.M;
        const int kg = (int)(e / a.M), i = (int)(e % a.M);
        const int k = a.g2l[kg];
        if (k >= 0)
            for (int c = 0; c < a.n_contigs; ++c)
                if (a.present[(size_t)c * a.K + k]) acc += a.gsum[((size_t)c * a.K + k) * a.Mp + i];
    }
    a.out[idx] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// K6: finalisation
// ---------------------------------------------------------------------------------------------------------------
struct FinArgs {
    int M, Mp, K, G, Ke, n_contigs;
    // bucket bookkeeping
    const int *eb_slab_off;   // [n_ebuckets+1] slabs of eigen bucket b are eb_slab_off[b] .. eb_slab_off[b+1]
    const int *eb_gid;        // [n_ebuckets]
    const int *ce_bucket_off; // [n_contigs*Ke + 1] eigen buckets of (contig, eigen key)
    const int *s1_slab_off;   // [n_contigs + 1] span-1 rank slabs per contig (for X1)
    const int *gk_slab_off;   // [n_contigs*K + 1] span-1 scalar slabs per (contig, key) (for gamma sums)
    const int *g_span;        // [G]
    const int *e_kid;         // [Ke] key id of eigen system
    const double *dsc;        // [Ke][Mp] scaled eigenvalues
    const double *dun;        // [Ke][Mp] unscaled eigenvalues
    const double *Prm, *Pinvrm;   // [Ke][Mp][Mp]
    const double *E;          // [K][Mp]
    const double *Td;         // [Mp][Mp] row-major
    int ZS;                   // shares per bucket of the reduced partials
    const double *red_e;      // [n eigen buckets][ZS][Mp][Mp]   (k_sum_parts of the eigen rank partials)
    const double *red_1;      // [n_contigs][ZS][Mp][Mp]         (span-1 rank partials)
    const double *red_g;      // [n_contigs*K][ZG][Mp]           (span-1 gamma partials; ZG shares, 1 except in the one-pass span-1 form)
    int ZG;
    const float *alpha;
    const double *beta;
    const long long *contig_base;
    double *Z;                // [n_contigs*Ke][Mp][Mp]
    double *Zpart;            // [slices][n_contigs*Ke][Mp][Mp] partial sums of k_fin_Z over slices of the groups
    double *Y;                // [n_contigs*Ke][Mp][Mp]
    double *xisum;            // [n_contigs][Mp][Mp]
    double *gsum;             // [n_contigs][K][Mp]
    double *gamma0;           // [n_contigs][Mp]
    int eigfree;              // 1: Y holds W of k_span_fold, the first Mp entries of Z its diag(A W) (no eigensystem anywhere)
    const double *dpow;       // [G][Mp] (d_r / scale)^span per group
    const double *part_e;     // [eigen slabs][Mp][Mp] rank partials of the eigen rows (k_fin_Z sums a bucket's slabs itself)
};

// span_Qs entry (transition_bundle.cpp:29-59) evaluated on the fly
__device__ __forceinline__ double span_q_elem(const double *dsc, int a, int b, int span) {
    double d1 = dsc[a];
    if (a == b) return pow(d1, span - 1) * (double)span;
    double d2 = dsc[b];
    if (fabs(d1) < fabs(d2)) { const double t = d1; d1 = d2; d2 = t; }
    if (d1 == d2) return pow(d1, span - 1) * (double)span;   // limit; the reference formula is 0/0 here
    const double q = exp((double)span * log(d1) + log1p(-pow(d2 / d1, span)));
    return q / (d1 - d2);
}

// The same entry from the group's eigenvalue powers pw[i] = dsc[i]^span (k_group_dpow): Q(a,b) = (d_a^s - d_b^s) / (d_a - d_b),
// Q(a,a) = s d_a^(s-1) - one division instead of pow + exp + log + log1p per entry (un-binned data have 10^4 - 10^5 groups of
// M^2 entries each: the transcendental form was 0.9 ms per posterior E-step).  Same conditioning as the reference's expression
// (both take the difference of two nearly equal numbers when d_a is close to d_b); equal eigenvalues give the limit.
__device__ __forceinline__ double span_q_pow(const double *dsc, const double *pw, int a, int b, int span) {
    const double d1 = dsc[a], p1 = pw[a];
    const double dg = d1 != 0.0 ? (double)span * p1 / d1 : 0.0;
    if (a == b) return dg;
    const double d2 = dsc[b];
    if (d1 == d2) return dg;
    return (p1 - pw[b]) / (d1 - d2);
}

// Z[(contig,e)][j][k] = sum over groups g of key e:  S_g[j][k] * Acc[contig][g][j][k]
// blockIdx.z = slice of the (contig, key)'s groups: binned data have a few dozen groups (one slice), un-binned data
// (posterior decoding) tens of thousands - one thread per matrix element looping over all of them serially left the
// chip idle for 60 ms.  Slices are contiguous ranges summed in order (k_fin_Zsum), so the result is deterministic.
__global__ __launch_bounds__(256) void k_fin_Z(FinArgs a) {
    const int ce = blockIdx.y;                    // contig * Ke + e
    const int e = ce % a.Ke;
    const int Mp = a.Mp, M = a.M;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Mp * Mp) return;
    const int j = idx / Mp, k = idx % Mp;
    const int nsl = gridDim.z, sl = blockIdx.z;
    const size_t MM = (size_t)Mp * Mp;
    double z = 0.0;
    if (j < M && k < M) {
        const double *dsc = a.dsc + (size_t)e * Mp;
        const double d1 = dsc[j], d2 = dsc[k];
        const double id1 = d1 != 0.0 ? 1.0 / d1 : 0.0;
        const bool same = j == k || d1 == d2;
        const double idd = same ? 0.0 : 1.0 / (d1 - d2);
        const int b0 = a.ce_bucket_off[ce], b1 = a.ce_bucket_off[ce + 1];
        const int per = (b1 - b0 + nsl - 1) / nsl;
        const int lo = b0 + sl * per, hi = min(b1, lo + per);
        // The slab partials of a bucket are summed HERE (un-binned data: one slab per group - a separate reduction pass only copied
        // 245 MB), four buckets in flight: the chain  bucket -> group -> (span, powers, partial)  is two round trips per batch
        // instead of two per bucket.
        for (int b = lo; b < hi; b += 4) {
            int gid[4], s0[4], s1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int bu = min(b + u, hi - 1);
                gid[u] = a.eb_gid[bu]; s0[u] = a.eb_slab_off[bu]; s1[u] = a.eb_slab_off[bu + 1];
            }
            double pa[4], pb[4], ac[4];
            int spn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                spn[u] = a.g_span[gid[u]];
                pa[u] = a.dpow[(size_t)gid[u] * Mp + j];
                pb[u] = a.dpow[(size_t)gid[u] * Mp + k];
                ac[u] = s1[u] > s0[u] ? a.part_e[(size_t)s0[u] * MM + idx] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                for (int q = s0[u] + 1; q < s1[u]; ++q) ac[u] += a.part_e[(size_t)q * MM + idx];
                // span-Q entry from the powers (span_q_pow): (p_a - p_b) / (d_a - d_b), span d^(span-1) on the diagonal
                const double qv = same ? (double)spn[u] * pa[u] * id1 : (pa[u] - pb[u]) * idd;
                if (b + u < hi) z += qv * ac[u];
            }
        }
    }
    double *out = nsl == 1 ? a.Z : a.Zpart;
    out[((size_t)sl * gridDim.y + ce) * MM + idx] = z;
}
// Z of the key from the reduced generation-2 partials:  red [ce][nsh][Mp*Mp + Mp] (k_sum_parts in nsh shares)
__global__ __launch_bounds__(256) void k_fin_Z2(FinArgs a, const double *red, int nsh) {
    const int ce = blockIdx.y;
    const int e = ce % a.Ke;
    const int Mp = a.Mp, M = a.M;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Mp * Mp) return;
    const int j = idx / Mp, k = idx % Mp;
    const size_t MM = (size_t)Mp * Mp, LEN = MM + Mp;
    double z = 0.0;
    if (j < M && k < M) {
        const double *dsc = a.dsc + (size_t)e * Mp;
        const double d1 = dsc[j], d2 = dsc[k];
        const bool diag = j == k;
        if ((diag && d1 != 0.0) || (!diag && d1 != d2)) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            const double *rp = red + (size_t)ce * nsh * LEN + (diag ? MM + j : (size_t)idx);
            int zz = 0;
            for (; zz + 3 < nsh; zz += 4) {
                s0 += rp[(size_t)zz * LEN]; s1 += rp[(size_t)(zz + 1) * LEN];
                s2 += rp[(size_t)(zz + 2) * LEN]; s3 += rp[(size_t)(zz + 3) * LEN];
            }
            for (; zz < nsh; ++zz) s0 += rp[(size_t)zz * LEN];
            z = ((s0 + s1) + (s2 + s3)) / (diag ? d1 : d1 - d2);
        }
    }
    a.Z[(size_t)ce * MM + idx] = z;
}

__global__ __launch_bounds__(256) void k_fin_Zsum(FinArgs a, int nsl, int nce) {
    const int ce = blockIdx.y;
    const int MM = a.Mp * a.Mp;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= MM) return;
    double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
    int sl = 0;
    for (; sl + 3 < nsl; sl += 4) {
        z0 += a.Zpart[((size_t)sl * nce + ce) * MM + idx];
        z1 += a.Zpart[((size_t)(sl + 1) * nce + ce) * MM + idx];
        z2 += a.Zpart[((size_t)(sl + 2) * nce + ce) * MM + idx];
        z3 += a.Zpart[((size_t)(sl + 3) * nce + ce) * MM + idx];
    }
    for (; sl < nsl; ++sl) z0 += a.Zpart[((size_t)sl * nce + ce) * MM + idx];
    a.Z[(size_t)ce * MM + idx] = (z0 + z1) + (z2 + z3);
}

// Y = Z * Pinv
__global__ __launch_bounds__(256) void k_fin_Y(FinArgs a) {
    const int ce = blockIdx.y;
    const int e = ce % a.Ke;
    const int Mp = a.Mp;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Mp * Mp) return;
    const int j = idx / Mp, i = idx % Mp;
    const double *Z = a.Z + (size_t)ce * Mp * Mp + (size_t)j * Mp;
    const double *Pinv = a.Pinvrm + (size_t)e * Mp * Mp;
    // four independent partial sums: the loop is bound by the latency of its loads, not by the M multiply-adds
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
#pragma unroll 2
    for (; k + 3 < a.M; k += 4) {
        s0 = fma(Z[k], Pinv[(size_t)k * Mp + i], s0);
        s1 = fma(Z[k + 1], Pinv[(size_t)(k + 1) * Mp + i], s1);
        s2 = fma(Z[k + 2], Pinv[(size_t)(k + 2) * Mp + i], s2);
        s3 = fma(Z[k + 3], Pinv[(size_t)(k + 3) * Mp + i], s3);
    }
    for (; k < a.M; ++k) s0 = fma(Z[k], Pinv[(size_t)k * Mp + i], s0);
    a.Y[(size_t)ce * Mp * Mp + idx] = (s0 + s1) + (s2 + s3);
}

// ---------------------------------------------------------------------------------------------------------------
// Eigen-free statistics of the span > 1 rows (binned data: spans of a few dozen positions at most).
// A row of span s applies A = diag(e) T^T s times; the reference evaluates the sum over the s positions of the row through
// the eigensystem of A and the span-Q matrix (hmm.cpp:113-122, transition_bundle.cpp:29-59: Q(a,b) = sum_t d_a^t d_b^(s-1-t)).
// Written without the eigensystem the same quantity is
//     xis_row = [ sum_{t=0}^{s-1} A^t (alpha_{ell-1} beta_ell^T) A^(s-1-t) ] diag(e),   v_row = diag( A [ ... ] ),
// both linear in the rank-one matrix, so with Acc_s = sum over the rows of span s of omega alpha beta^T (k_rank_acc<2>)
//     W = sum_s sum_{t<s} A^t Acc_s A^(s-1-t)  =  H_0,    F_t = Acc_{t+1} + F_{t+1} A,   H_t = F_t + A H_{t+1}
// (two M x M products per step, smax steps) and  xisum += W diag(e),  gamma_sums[key] += diag(A W).
// Rows of F do not mix and columns of H do not mix, so both recurrences split into independent 16-wide strips, one wavefront
// each, and written as LEFT products (F^T_t = Acc^T + A^T F^T_{t+1}) the D fragments of one v_mfma_f64_16x16x4 step are the B
// fragments of the next (row 4 kk + qd of the operand sits on lane group qd, register kk % 4): no LDS, no barrier.
// ---------------------------------------------------------------------------------------------------------------
// One workgroup per (contig, key), 2 NT wavefronts: wavefronts 0 .. NT-1 run the F strips one step AHEAD of wavefronts
// NT .. 2 NT - 1, which run the H strips; F_t goes from one group to the other through a double-buffered LDS copy, one barrier
// per step.

// The same fold with one workgroup per 16-wide STRIP (NTP = padded tile count: 1 .. 4 for M <= 64, 8 / 12 / 16 up to 256): a
// workgroup of NTP wavefronts, wavefront tt owns output tile tt of the strip and keeps ITS A fragments (MT/4 doubles per lane) in
// registers for all steps; the strip (MT x 16) is exchanged through a double-buffered LDS copy, one barrier per step.
// PHASE 0: F^T strips (rows of F), F_t written to scratch;  PHASE 1: H strips (columns of H), W = H_0 and diag(A W) written at
// the end.  A one-workgroup-per-(contig, key) fold (round 3, removed) kept a (contig, key) on ONE CU, whose four matrix pipes then bound a step (2 x 16 tiles x 16 k-steps
// x 64 cycles / 4 = 3.4 us); here a step is 16 MFMAs per wavefront.  What a step adds to the product (the Acc bucket of its span,
// or F_t from the scratch) is fetched FOUR STEPS AHEAD with unconditional, index-clamped loads - on the serial path the three
// dependent round trips bucket -> span -> matrix cost more than the product itself; the bucket of every step comes from a
// small LDS table built once.
template <int NTP, int PHASE>
__device__ __forceinline__ void span_big_body(const FinArgs &a, int smax, double *__restrict__ Fall, int blk, double (*sX)[16 * NTP * 17],
                                              int *sbk) {
    constexpr int MT = 16 * NTP, LDX = 17;
    __builtin_amdgcn_s_setprio(3);                                    // a serial chain of small products beside chip-filling kernels
    const int ns = (a.Mp + 15) / 16;                                  // strips that exist
    const int ce = blk / ns, sp = blk % ns, e = ce % a.Ke;
    const int b0 = a.ce_bucket_off[ce], b1 = a.ce_bucket_off[ce + 1];
    if (b0 == b1) return;
    const int tid = threadIdx.x, lane = tid & 63, tt = tid >> 6;
    const int m = lane & 15, qd = lane >> 4;
    const int Mp = a.Mp, M = a.M;
    const double *ek = a.E + (size_t)a.e_kid[e] * Mp;
    if (PHASE == 0) {
        for (int idx = tid; idx < 64; idx += 64 * NTP) sbk[idx] = -1;
        __syncthreads();
        for (int b = b0 + tid; b < b1; b += 64 * NTP) {
            const int sp_ = a.g_span[a.eb_gid[b]];
            if (sp_ >= 1 && sp_ <= 64) sbk[sp_ - 1] = b;
        }
    }
    // A operand of this wavefront's output tile, k = 4 kk + qd:  PHASE 0: A^T[16 tt + m][k] = e_k T[16 tt + m][k];  PHASE 1: A[16 tt + m][k] = e_i T[k][i]
    double af[MT / 4];
#pragma unroll
    for (int kk = 0; kk < MT / 4; ++kk) {
        const int k = 4 * kk + qd, i = 16 * tt + m;
        const int kc = min(k, Mp - 1), ic = min(i, Mp - 1);
        const double v = PHASE == 0 ? ek[kc] * a.Td[(size_t)ic * Mp + kc] : ek[ic] * a.Td[(size_t)kc * Mp + ic];
        af[kk] = (k < M && i < M) ? v : 0.0;
    }
    // the strip starts at zero
    for (int idx = tid; idx < MT * LDX; idx += 64 * NTP) { sX[0][idx] = 0.0; sX[1][idx] = 0.0; }
    __syncthreads();
    double *Fce = Fall + (size_t)ce * smax * Mp * Mp;
    // element (row, column) this lane adds in register r:  PHASE 0: Acc[row of F = 16 sp + m][column of F = 16 tt + qd + 4 r];
    // PHASE 1: F_t[row 16 tt + qd + 4 r][column 16 sp + m]
    int eoff[4];
    bool eok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = PHASE == 0 ? 16 * sp + m : 16 * tt + qd + 4 * r;
        const int col = PHASE == 0 ? 16 * tt + qd + 4 * r : 16 * sp + m;
        eok[r] = row < Mp && col < Mp;
        eoff[r] = min(row, Mp - 1) * Mp + min(col, Mp - 1);
    }
    auto fetch = [&](int t, double (&v)[4]) {                          // raw loads; masked where they are consumed
        const int tc = max(t, 0);
        const double *src;
        if (PHASE == 0) src = a.red_e + (size_t)max(sbk[tc], b0) * Mp * Mp;
        else src = Fce + (size_t)tc * Mp * Mp;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = src[eoff[r]];
    };
    // ring of PF steps in flight: beside the chip-filling rank updates a round trip to L2 / HBM takes several steps' worth of time
    constexpr int PF = 4;
    double q[PF][4];
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(smax - 1 - d, q[d]);
    f64x4 X = {0, 0, 0, 0};
    for (int t0 = smax - 1; t0 >= 0; t0 -= PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int t = t0 - d;
            if (t < 0) break;
            const int cur = (smax - 1 - t) & 1;
            const double *sr = sX[cur];
            double *sw = sX[cur ^ 1];
            const bool has = PHASE == 1 || sbk[t] >= 0;
            double av[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) av[r] = (has && eok[r]) ? q[d][r] : 0.0;
            fetch(t - PF, q[d]);
            f64x4 Xn = {0, 0, 0, 0};
#pragma unroll 8
            for (int kk = 0; kk < MT / 4; ++kk)
                Xn = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], sr[(4 * kk + qd) * LDX + m], Xn, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                X[r] = Xn[r] + av[r];
                sw[(16 * tt + qd + 4 * r) * LDX + m] = X[r];
            }
            if (PHASE == 0) {
                double *Ft = Fce + (size_t)t * Mp * Mp;
#pragma unroll
                for (int r = 0; r < 4; ++r)                           // F_t[row 16 sp + m][column 16 tt + qd + 4 r]
                    if (eok[r]) Ft[eoff[r]] = X[r];
            }
            __syncthreads();
        }
    }
    if (PHASE == 0) return;
    // W = H_0 strip; diag(A W): tile sp of A W, computed by wavefront sp from the final strip
    double *Wout = a.Y + (size_t)ce * Mp * Mp;
    double *gout = a.Z + (size_t)ce * Mp * Mp;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (eok[r]) Wout[eoff[r]] = X[r];
    if (tt == sp) {
        const double *sr = sX[smax & 1];
        f64x4 G = {0, 0, 0, 0};
#pragma unroll 8
        for (int kk = 0; kk < MT / 4; ++kk)
            G = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], sr[(4 * kk + qd) * LDX + m], G, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (qd + 4 * r == m && 16 * sp + m < Mp) gout[16 * sp + m] = G[r];
    }
}

template <int NTP, int PHASE>
__global__ __launch_bounds__(64 * NTP) void k_span_big(FinArgs a, int smax, double *__restrict__ Fall) {
    __shared__ double sX[2][16 * NTP * 17];
    __shared__ int sbk[64];                                           // bucket holding span t + 1, or -1 (smax <= 64)
    span_big_body<NTP, PHASE>(a, smax, Fall, (int)blockIdx.x, sX, sbk);
}

// xisum[contig] = max( (X1 + sum_e P_e Y_e diag(b_e)) o Td , 1e-20 )   (hmm.cpp:122,141,151-152)
__device__ __forceinline__ void fin_xisum_body(const FinArgs &a, int bx, int ct) {
    const int Mp = a.Mp, M = a.M;
    const int idx = bx * 256 + threadIdx.x;
    if (idx >= Mp * Mp) return;
    const int i = idx / Mp, k = idx % Mp;
    double x = 0.0;
    if (i < M && k < M) {
        for (int z = 0; z < a.ZS; ++z) x += a.red_1[((size_t)ct * a.ZS + z) * Mp * Mp + idx];
        for (int e = 0; e < a.Ke; ++e) {
            const int ce = ct * a.Ke + e;
            if (a.ce_bucket_off[ce] == a.ce_bucket_off[ce + 1]) continue;
            const double *P = a.Prm + (size_t)e * Mp * Mp + (size_t)i * Mp;
            const double *Y = a.Y + (size_t)ce * Mp * Mp;
            if (a.eigfree) { x += Y[(size_t)i * Mp + k] * a.E[(size_t)a.e_kid[e] * Mp + k]; continue; }
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int j = 0;
#pragma unroll 2
            for (; j + 3 < M; j += 4) {
                s0 = fma(P[j], Y[(size_t)j * Mp + k], s0);
                s1 = fma(P[j + 1], Y[(size_t)(j + 1) * Mp + k], s1);
                s2 = fma(P[j + 2], Y[(size_t)(j + 2) * Mp + k], s2);
                s3 = fma(P[j + 3], Y[(size_t)(j + 3) * Mp + k], s3);
            }
            for (; j < M; ++j) s0 = fma(P[j], Y[(size_t)j * Mp + k], s0);
            x += ((s0 + s1) + (s2 + s3)) * a.E[(size_t)a.e_kid[e] * Mp + k];
        }
        x *= a.Td[(size_t)i * Mp + k];
        if (x < 1e-20) x = 1e-20;
    }
    a.xisum[(size_t)ct * Mp * Mp + idx] = x;
}

// gamma_sums[contig][key] and gamma0[contig]   (hmm.cpp:116-121,146,150)
__device__ __forceinline__ void fin_gamma_body(const FinArgs &a, int bx, int ct) {
    const int Mp = a.Mp, M = a.M;
    const int idx = bx * 256 + threadIdx.x;
    if (idx >= (a.K + 1) * Mp) return;
    const int k = idx / Mp, i = idx % Mp;
    if (k == a.K) {                               // gamma.col(0) = alpha_0 o beta_0 (not normalised)
        const size_t row = (size_t)a.contig_base[ct];
        a.gamma0[(size_t)ct * Mp + i] = (i < M) ? (double)a.alpha[row * Mp + i] * a.beta[row * Mp + i] : 0.0;
        return;
    }
    double g = 0.0;
    if (i < M) {
        const int ck = ct * a.K + k;
        for (int z = 0; z < a.ZG; ++z) g += a.red_g[((size_t)ck * a.ZG + z) * Mp + i];
        for (int e = 0; e < a.Ke; ++e) {
            if (a.e_kid[e] != k) continue;
            const int ce = ct * a.Ke + e;
            if (a.ce_bucket_off[ce] == a.ce_bucket_off[ce + 1]) continue;
            if (a.eigfree) { g += a.Z[(size_t)ce * Mp * Mp + i]; continue; }
            const double *P = a.Prm + (size_t)e * Mp * Mp + (size_t)i * Mp;
            const double *Y = a.Y + (size_t)ce * Mp * Mp;
            const double *d = a.dun + (size_t)e * Mp;
            double s = 0.0;
            for (int j = 0; j < M; ++j) s = fma(P[j] * d[j], Y[(size_t)j * Mp + i], s);
            g += s;
        }
    }
    a.gsum[((size_t)ct * a.K + k) * Mp + i] = g;
}
// both finalisations in one launch (they read disjoint inputs of the same phase): blocks [0, nbx) xisum, [nbx, nbx + nbg) gamma sums
// `done` != nullptr: this launch ends the E-step's queue - the block that finishes LAST (a counter that only ever grows: `target` =
// blocks of every such launch so far) raises the host-visible completion word itself instead of a one-thread kernel behind it
__global__ __launch_bounds__(256) void k_fin_both(FinArgs a, int nbx, unsigned *ctr, unsigned target, int *done, int epoch) {
    if ((int)blockIdx.x < nbx) fin_xisum_body(a, (int)blockIdx.x, (int)blockIdx.y);
    else fin_gamma_body(a, (int)blockIdx.x - nbx, (int)blockIdx.y);
    if (!done) return;
    __syncthreads();                                   // every store of this block has been issued ...
    if (threadIdx.x == 0) {
        __threadfence();                               // ... and is visible device-wide before the block counts itself
        const unsigned prev = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == target) {
            __threadfence_system();
            __hip_atomic_store(done, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K8: per-row gamma of eigen rows for save_gamma  (hmm.cpp:113-121,147-148)
//   g_i = sum_j P_ij d_j u_j (sum_k S_jk w_k Pinv_ki);  gamma_row = span * |g| / sum |g|
// One workgroup (256 threads) per row; S is evaluated once per group into a table by k_span_q.
// ---------------------------------------------------------------------------------------------------------------
struct GammaRowArgs {
    int M, Mp, nrows;
    const int *perm;          // eigen rows (position p -> contig-relative ell), sorted by bucket
    const int *row_slab;      // [nrows] slab index of each position (to find base / gid)
    const Slab *slabs;
    const int *g_eig;
    const int *g_span;
    const double *dun;        // [Ke][Mp]
    const double *dsc;        // [Ke][Mp] scaled eigenvalues (what the span-Q entries are built from)
    const double *dpow;       // [G][Mp] dsc^span per group
    const double *Prm, *Pinvrm, *PinvT;
    const double *Sq;         // [G][Mp][Mp] span-Q tables (k_gamma_rows_eig only)
    const float *alpha;
    const double *beta;
    double *gamma_rows;       // [rows][Mp]
    const int4 *erow_desc = nullptr;   // [nrows] {row (low, high word), group, span} of every sorted position: ONE load where perm -> row_slab -> slabs -> g_span are three dependent ones (k_gamma_rows_b)
};

__global__ __launch_bounds__(256) void k_span_q(int M, int Mp, int G, const int *g_span, const int *g_eig,
                                                const double *dsc, const double *dpow, double *Sq) {
    const int g = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Mp * Mp) return;
    const int j = idx / Mp, k = idx % Mp;
    double v = 0.0;
    if (j < M && k < M) v = span_q_pow(dsc + (size_t)g_eig[g] * Mp, dpow + (size_t)g * Mp, j, k, g_span[g]);
    Sq[(size_t)g * Mp * Mp + idx] = v;
}

__global__ __launch_bounds__(256) void k_gamma_rows_eig(GammaRowArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int Mp = a.Mp, M = a.M;
    double *u = sm, *w = sm + Mp, *g = sm + 2 * Mp, *red = sm + 3 * Mp;   // red[256]
    const int p = blockIdx.x;
    const Slab sl = a.slabs[a.row_slab[p]];
    const int gid = sl.aux;
    const int es = a.g_eig[gid];
    const int span = a.g_span[gid];
    const int ell = a.perm[p];
    const size_t row = (size_t)(sl.base + ell);
    const double *Prm = a.Prm + (size_t)es * Mp * Mp;
    const double *Pinvrm = a.Pinvrm + (size_t)es * Mp * Mp;
    const double *PinvT = a.PinvT + (size_t)es * Mp * Mp;
    const double *S = a.Sq + (size_t)gid * Mp * Mp;      // symmetric
    const double *d = a.dun + (size_t)es * Mp;
    const int tid = threadIdx.x;
    // u_j <- d_j * (Pinv alpha_{ell-1})_j ; w = P^T beta_ell
    for (int i = tid; i < Mp; i += 256) {
        double su = 0.0, sw = 0.0;
        if (i < M) {
            const float *ap = a.alpha + (row - 1) * Mp;
            const double *bp = a.beta + row * Mp;
            for (int j = 0; j < M; ++j) {
                su = fma(PinvT[(size_t)j * Mp + i], (double)ap[j], su);
                sw = fma(Prm[(size_t)j * Mp + i], bp[j], sw);
            }
            su *= d[i];
        }
        u[i] = su; w[i] = sw;
    }
    __syncthreads();
    // g_i = sum_k w_k Pinv_ki H_ik,  H_ik = sum_j (P_ij d_j u_j) S_jk.  Threads = TI states x KP slices of k.
    int TI = 1;
    while (TI < Mp && TI < 256) TI <<= 1;
    const int KP = 256 / TI;
    const int il = tid % TI, kp = tid / TI;
    double tot = 0.0;
    for (int i0 = 0; i0 < Mp; i0 += TI) {
        const int i = i0 + il;
        double gi = 0.0;
        if (i < M) {
            const double *Pi = Prm + (size_t)i * Mp;
            for (int k = kp; k < M; k += KP) {
                const double *Sk = S + (size_t)k * Mp;
                double h = 0.0;
                for (int j = 0; j < M; ++j) h = fma(Pi[j] * u[j], Sk[j], h);
                gi = fma(w[k] * Pinvrm[(size_t)k * Mp + i], h, gi);
            }
        }
        red[tid] = gi;
        __syncthreads();
        if (kp == 0 && i < Mp) {
            double sacc = 0.0;
            for (int t = 0; t < KP; ++t) sacc += red[t * TI + il];
            g[i] = (i < M) ? fabs(sacc) : 0.0;
        }
        __syncthreads();
    }
    for (int i = tid; i < M; i += 256) tot += g[i];
    red[tid] = tot;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    const double sum = red[0];
    for (int i = tid; i < Mp; i += 256)
        a.gamma_rows[row * Mp + i] = (i < M) ? (double)span * g[i] / sum : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------
// gamma rows of eigen rows on the matrix cores (M <= 64).  Per row the reference evaluates (hmm.cpp:113-121)
//     g = diag( P D (u w^T o S) P^-1 ),   gamma_row = span |g| / sum |g|,     u = P^-1 alpha_{l-1}, w = P^T beta_l,
// i.e. 2 M^3 flops: G = P Z with Z[a][b] = (d_a u_a) S_ab w_b, then g_i = sum_b G_ib Pinv_bi.  k_gamma_rows_eig did this
// with one workgroup of scalar FMAs per row (0.35 TFLOP/s); here one WAVEFRONT owns a row and the M x M x M product runs
// as v_mfma_f64_16x16x4_f64 tiles: A = P from an LDS copy shared by the 4 wavefronts of the workgroup (one launch per
// (contig, eigen key), so a workgroup never mixes keys), B = Z built on the fly from the group's span-Q table (read
// once per row, coalesced 128-byte pieces) and the row's u, w; the D tiles are folded into g against Pinv (LDS) and
// reduced over the 16 lanes of a DPP row.  A wavefront walks over ROWS rows so the LDS staging is amortised.
// MFMA operand map (guide §3): A[m = l&15][k = l>>4], B[k = l>>4][n = l&15], D[row = (l>>4) + 4 reg][col = l&15].
// ---------------------------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------------------------
// k_gamma_rows_b (round 3; generation 1, k_gamma_rows_mfma, was removed in round 5).  The mathematics and the operand map are those
// described above; against a one-row-per-wavefront kernel with scalar dot products and a span-Q table in memory:
//   * u = d o (Pinv alpha), w = P^T beta of SIXTEEN rows at a time as two MFMA products (the rows of a launch share the key, so
//     Pinv / P are common; the scalar dot products of generation 1 cost as many cycles as the M^3 product itself);
//   * the span-Q entries are formed on the fly from the group's eigenvalue powers, S_ab = (p_a - p_b) * 1/(d_a - d_b) with the
//     reciprocal differences in an LDS table per key: no [G][M][M] table in memory (245 MB on the posterior workload), no 8 KB
//     of it through L2 per row, no k_span_q launch;
//   * the fold's reduction over the 16 columns goes through a padded LDS tile (one write per value, 16 reads per state)
//     instead of a DPP tree per value.
// A wavefront walks `nbatch` batches of 16 consecutive rows (sorted by group: the powers are re-read only when the group changes).
// ---------------------------------------------------------------------------------------------------------------
// (round 4) M > 32: the kernel is bound by the matrix pipe of the SIMD its wavefront sits on (256 MFMAs per row at NT = 4), and
// with the 33 KB table of reciprocal differences in LDS only TWO wavefronts fitted a CU - half its matrix pipes idle.  Now that
// table lives in registers (NT KS doubles per lane; one wavefront per SIMD has 512 of them) and the fold tile is half as wide
// (a DPP pair-add first): FOUR wavefronts per workgroup at every NT.
template <int NT>
__global__ __launch_bounds__(256) void k_gamma_rows_b(GammaRowArgs a, int p0, int p1, int es, int nbatch) {
    constexpr int MT = 16 * NT, LD = MT + 1, NW = 4, KS = MT / 4;
    constexpr bool REG = NT <= 2;            // M <= 32: the lane's fragments of P, Pinv and the reciprocal differences live in registers
    constexpr bool RID = NT > 2;             // M > 32: the reciprocal differences only (no LDS copy of them)
    constexpr int GW = RID ? 9 : 17;         // row pitch of the fold tile (RID: column pairs summed in registers first)
    constexpr int NR = REG ? NT * KS : 1, NID = (REG || RID) ? NT * KS : 1, NF = REG ? NT * NT * 4 : 1;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *sP = sm;                         // [MT][LD]  P row-major
    double *sPinv = sP + MT * LD;            // [MT][LD]  Pinv row-major
    double *sInvD = sPinv + MT * LD;         // [MT][LD]  1 / (d_a - d_b), 0 where the eigenvalues are equal   (!RID)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *sU = sPinv + (RID ? 1 : 2) * MT * LD + (size_t)wv * (2 * 16 * LD + MT * GW + MT);   // per wavefront: x = d o u [16 rows][LD], w [16][LD], fold tile [MT][GW], the group's powers [MT]
    double *sW = sU + 16 * LD, *sG = sW + 16 * LD, *sPW = sG + MT * GW;
    double *sIda = sPinv + (RID ? 1 : 2) * MT * LD + (size_t)NW * (2 * 16 * LD + MT * GW + MT);   // [MT] 1 / d~_a (RID)
    const int Mp = a.Mp, M = a.M;
    const double *dsc = a.dsc + (size_t)es * Mp, *dun = a.dun + (size_t)es * Mp;
    {
        const double *Prm = a.Prm + (size_t)es * Mp * Mp, *Pinvrm = a.Pinvrm + (size_t)es * Mp * Mp;
        for (int idx = tid; idx < MT * MT; idx += 64 * NW) {
            const int r = idx / MT, c = idx % MT;
            sP[r * LD + c] = Prm[(size_t)r * Mp + c];
            sPinv[r * LD + c] = Pinvrm[(size_t)r * Mp + c];
            if (!RID) {
                const double dd = dsc[r] - dsc[c];
                sInvD[r * LD + c] = (dd != 0.0 && r < M && c < M) ? 1.0 / dd : 0.0;
            }
        }
        if (RID && tid < MT) {
            const double d = dsc[min(tid, Mp - 1)];
            sIda[tid] = (tid < M && d != 0.0) ? 1.0 / d : 0.0;
        }
    }
    __syncthreads();
    const int kq = lane >> 4, n = lane & 15;
    // 1 / d_aa (the diagonal of the span-Q matrix) and the unscaled eigenvalues: per-lane registers for M <= 32; for larger M they are
    // re-read (L1 / L2 hits, once per batch of 16 rows / per change of group) - the registers are needed for the fragments
    constexpr int NK = RID ? 1 : KS, ND = RID ? 1 : NT;
    double invda[NK], dux[ND][4];
    if (!RID) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int aa = 4 * kk + kq;
            const double d = dsc[aa];
            invda[kk] = (aa < M && d != 0.0) ? 1.0 / d : 0.0;
        }
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = it * 16 + kq + 4 * r;
                dux[it][r] = i < M ? dun[i] : 0.0;
            }
    }
    double rP[NR], rID[NID], rF[NF];         // A fragments of P [it][kk], 1/(d_aa - d_bb) [bt][kk], Pinv[bb][i] of the fold [bt][it][r]
    if (RID) {
#pragma unroll
        for (int bt = 0; bt < NT; ++bt)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int aa = 4 * kk + kq, bb = bt * 16 + n;
                const double dd = dsc[aa] - dsc[bb];
                rID[bt * KS + kk] = (dd != 0.0 && aa < M && bb < M) ? 1.0 / dd : 0.0;
            }
    }
    if (REG) {
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                rP[it * KS + kk] = sP[(it * 16 + n) * LD + 4 * kk + kq];
                rID[it * KS + kk] = sInvD[(4 * kk + kq) * LD + it * 16 + n];
            }
#pragma unroll
        for (int bt = 0; bt < NT; ++bt)
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) rF[(bt * NT + it) * 4 + r] = sPinv[(bt * 16 + n) * LD + it * 16 + kq + 4 * r];
    }
    const int pw0 = p0 + (blockIdx.x * NW + wv) * nbatch * 16;
    const int pw1 = min(p1, pw0 + nbatch * 16);
    int gid_prev = -1, span = 1;
    // (round 6) un-binned data change the (span, key) group on nearly every row: the group's M powers used to be fetched - twenty
    // dependent global loads per lane, and sixteen divisions 1 / d~ at M > 32 - at the head of the row, a full memory round trip with
    // nothing else resident on the SIMD.  Now lane a fetches power a of the NEXT row's group while this row's products run (one
    // double per lane), the head of the row spreads it through LDS, and 1 / d~ comes from a table formed once per workgroup.
    int gid_pf = -1;
    double pw_pf = 0.0;
    double pa[KS], sd[KS], pb[NT];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) { pa[kk] = 0.0; sd[kk] = 0.0; }
#pragma unroll
    for (int bt = 0; bt < NT; ++bt) pb[bt] = 0.0;
    // (round 6) the descriptor of the NEXT batch's rows is fetched while this batch runs (one 16-byte load per lane)
    int4 desc_nx = pw0 < pw1 ? a.erow_desc[min(pw0 + n, pw1 - 1)] : make_int4(0, 0, 0, 1);
    for (int pbat = pw0; pbat < pw1; pbat += 16) {
        const int nb = min(16, pw1 - pbat);
        // lane n describes row pbat + n: the row loop below reads group, span and row index with v_readlane (no dependent
        // global loads on the per-row path)
        const int4 desc = desc_nx;
        if (pbat + 16 < pw1) desc_nx = a.erow_desc[min(pbat + 16 + n, pw1 - 1)];
        const int gid_n = desc.z;
        const long long rown = ((long long)desc.y << 32) | (unsigned)desc.x;
        const int span_n = desc.w;
        // ---- x = d o (Pinv alpha_{l-1}),  w = P^T beta_l for the 16 rows of the batch (column n of the products = row pbat + n) ----
        {
            const float *ap = a.alpha + (size_t)(rown - 1) * Mp;
            const double *bp = a.beta + (size_t)rown * Mp;
            f64x4 DU[NT], DW[NT];
#pragma unroll
            for (int it = 0; it < NT; ++it) { DU[it] = (f64x4){0, 0, 0, 0}; DW[it] = (f64x4){0, 0, 0, 0}; }
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int st = 4 * kk + kq;
                const double av = (double)ap[st], bv = bp[st];
#pragma unroll
                for (int it = 0; it < NT; ++it) {
                    DU[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(sPinv[(it * 16 + n) * LD + st], av, DU[it], 0, 0, 0);
                    DW[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(sP[st * LD + it * 16 + n], bv, DW[it], 0, 0, 0);
                }
            }
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = it * 16 + kq + 4 * r;
                    sU[n * LD + i] = DU[it][r] * (RID ? (i < M ? dun[min(i, Mp - 1)] : 0.0) : dux[RID ? 0 : it][r]);
                    sW[n * LD + i] = DW[it][r];
                }
        }
        wave_lds_fence();
        for (int q = 0; q < nb; ++q) {
            const int gid = __builtin_amdgcn_readlane(gid_n, q);
            const size_t row = ((size_t)(unsigned)__builtin_amdgcn_readlane((int)(rown >> 32), q) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)(rown & 0xffffffffll), q);
            if (gid != gid_prev) {                                   // wave-uniform
                gid_prev = gid;
                span = __builtin_amdgcn_readlane(span_n, q);
                if (RID) {
                    double mine = pw_pf;
                    if (gid != gid_pf) mine = a.dpow[(size_t)gid * Mp + min(lane, Mp - 1)];      // (first row of a batch, or a row whose group was not the announced one)
                    if (lane < MT) sPW[lane] = mine;
                    wave_lds_fence();
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) {
                        pa[kk] = sPW[4 * kk + kq];
                        sd[kk] = (double)span * pa[kk] * sIda[4 * kk + kq];            // span d^(span-1)
                    }
#pragma unroll
                    for (int bt = 0; bt < NT; ++bt) pb[bt] = sPW[16 * bt + n];
                } else {
                    // (M <= 32: six to ten loads per lane, and the detour through LDS costs more than it hides - 0.83 -> 0.99 ms measured)
                    const double *pw = a.dpow + (size_t)gid * Mp;
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) {
                        pa[kk] = pw[4 * kk + kq];
                        sd[kk] = (double)span * pa[kk] * invda[RID ? 0 : kk];
                    }
#pragma unroll
                    for (int bt = 0; bt < NT; ++bt) pb[bt] = pw[16 * bt + n];
                }
            }
            if (RID && q + 1 < nb) {
                const int gnx = __builtin_amdgcn_readlane(gid_n, q + 1);
                if (gnx != gid) {                                    // wave-uniform
                    pw_pf = a.dpow[(size_t)gnx * Mp + min(lane, Mp - 1)];
                    gid_pf = gnx;
                }
            }
            double xs[NK];
            if (!RID) {
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) xs[kk] = sU[q * LD + 4 * kk + kq];
            }
            double gacc[NT][4];
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) gacc[it][r] = 0.0;
#pragma unroll
            for (int bt = 0; bt < NT; ++bt) {
                const int bb = bt * 16 + n;
                const double wb = sW[q * LD + bb];
                f64x4 D[NT];
#pragma unroll
                for (int it = 0; it < NT; ++it) D[it] = (f64x4){0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    const int aa = 4 * kk + kq;
                    const double idf = (REG || RID) ? rID[bt * KS + kk] : sInvD[aa * LD + bb];
                    const double sq = (aa == bb) ? sd[kk] : (pa[kk] - pb[bt]) * idf;
                    const double bf = (RID ? sU[q * LD + aa] : xs[RID ? 0 : kk]) * wb * sq;     // B[k = aa][n = bb] = (d u)_aa S_ab w_bb
#pragma unroll
                    for (int it = 0; it < NT; ++it)
                        D[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(REG ? rP[it * KS + kk] : sP[(it * 16 + n) * LD + aa], bf, D[it], 0, 0, 0);
                }
#pragma unroll
                for (int it = 0; it < NT; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        gacc[it][r] = fma(D[it][r], REG ? rF[(bt * NT + it) * 4 + r] : sPinv[bb * LD + it * 16 + kq + 4 * r], gacc[it][r]);
            }
            // g_i = sum over the columns: tile [i][n] through LDS, lane i sums its row
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (RID) {
                        const double v = gacc[it][r] + dpp_mov<0xB1>(gacc[it][r]);      // columns n and n ^ 1 (quad_perm [1,0,3,2])
                        if (!(n & 1)) sG[(it * 16 + kq + 4 * r) * GW + (n >> 1)] = v;
                    } else sG[(it * 16 + kq + 4 * r) * GW + n] = gacc[it][r];
                }
            wave_lds_fence();
            double mine = 0.0;
            if (lane < MT) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int c = 0; c < (RID ? 8 : 16); c += 2) { s0 += sG[lane * GW + c]; s1 += sG[lane * GW + c + 1]; }
                mine = lane < M ? fabs(s0 + s1) : 0.0;
            }
            const double tot = wave_sum(mine);
            if (lane < Mp) a.gamma_rows[row * Mp + lane] = (double)span * mine / tot;
            wave_lds_fence();
        }
    }
}

// argmax over states per row (posterior decoding indices)
__global__ __launch_bounds__(256) void k_gamma_argmax(int M, int Mp, long long nrows, const double *gamma_rows, int *out) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const double *g = gamma_rows + (size_t)r * Mp;
    int best = 0;
    double bv = g[0];
    for (int i = 1; i < M; ++i)
        if (g[i] > bv) { bv = g[i]; best = i; }
    out[r] = best;
}

// rows cut into pieces (engine_manager.hpp: build): the posterior of a caller's row is the sum of its pieces' posteriors.
// first[u] .. first[u + 1] - 1 are the pieces (contig-relative rows) of the caller's row u (u = 1 .. Lu); row 0 is not touched.
__global__ __launch_bounds__(256) void k_gamma_merge(int Mp, int Lu, const int *__restrict__ first, const double *__restrict__ gamma_rows,
                                                     double *__restrict__ out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long u = idx / Mp + 1;
    const int i = (int)(idx % Mp);
    if (u > Lu) return;
    double acc = 0.0;
    for (int l = first[u]; l < first[u + 1]; ++l) acc += gamma_rows[(size_t)l * Mp + i];
    out[(size_t)u * Mp + i] = acc;
}

}  // namespace smcpp_dev
