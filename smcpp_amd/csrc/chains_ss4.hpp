// Scan chains, four chains per wavefront (M <= 64): the fp64 passes of chains_ss.hpp in the layout that makes a position cheap.
//
// In chains_ss.hpp a chain owns the 64 lanes, one state per lane: a position is six DPP levels per scan and every level of an
// fp64 scan is three instructions (two 32-bit DPP moves and the add), ~55 instructions per position.  Here a chain owns ONE
// DPP row (16 lanes) and a lane owns SPL = M/16 consecutive states: the scans are SPL - 1 serial steps inside the lane plus
// FOUR row levels, and one instruction stream advances the four chains of the wavefront: ~80 instructions per step of four
// chains.  The four chains are four different chunks, so their rows end at different steps: what a row boundary needs
// (normaliser, float rounding / floor of the stored vector and its feedback, stores, next descriptor and emission vector)
// runs under a per-chain predicate on the steps where some chain is at a boundary.
// Four times as many chunks as wavefronts: the light passes of chains_ss.hpp (which pay the history) keep the coarse chunks
// and hand over the boundary vectors of the fine ones.
#pragma once

namespace smcpp_dev {

constexpr int DPP_RBC15 = 0x15F;     // row_newbcast:15 (gfx90a+): lane 15 of every row to the 16 lanes of the row

template <int SPL>
struct S4FwdC { double dc[SPL], g[SPL], cg[SPL], b[SPL], a[SPL], d[SPL], cumA[SPL], lv[6]; };
template <int SPL>
struct S4BwdC { double dc[SPL], g[SPL], b[SPL], a[SPL], cumA[SPL], lv[6], c0; };

template <int SPL>
__device__ __forceinline__ void ss4_load_fwd(const SsArgs &a, int q, int lane, S4FwdC<SPL> &c) {
    double cum = 1.0;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int p = q * SPL + k;
        c.dc[k] = a.f_dc[p]; c.g[k] = a.f_g[p]; c.cg[k] = a.f_cg[p]; c.b[k] = a.f_b[p]; c.a[k] = a.f_a[p]; c.d[k] = a.f_d[p];
        cum *= c.a[k];
        c.cumA[k] = cum;
    }
    ss_levels(cum, lane, c.lv);          // only the four row levels are used
}
template <int SPL>
__device__ __forceinline__ void ss4_load_bwd(const SsArgs &a, int q, int lane, S4BwdC<SPL> &c) {
    double cum = 1.0;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int p = q * SPL + k;
        c.dc[k] = a.b_dc[p]; c.g[k] = a.b_g[p]; c.b[k] = a.b_b[p]; c.a[k] = a.b_a[p];
        cum *= c.a[k];
        c.cumA[k] = cum;
    }
    ss_levels(cum, lane, c.lv);
    c.c0 = a.c0;
}

// one position of four forward chains:  out = e o (T^T x);  S = sum x of the lane's chain
template <int SPL>
__device__ __forceinline__ void ss4_fwd_step(const S4FwdC<SPL> &c, const double (&x)[SPL], const double (&e)[SPL],
                                             double (&out)[SPL], double &S) {
    double lp[SPL], w[SPL];
    lp[0] = x[0];
    w[0] = c.b[0] * x[0];
#pragma unroll
    for (int k = 1; k < SPL; ++k) {
        lp[k] = lp[k - 1] + x[k];
        w[k] = __builtin_fma(c.a[k], w[k - 1], c.b[k] * x[k]);
    }
    double p_ = lp[SPL - 1], z_ = w[SPL - 1];
    {
        double tp, tz;
        tp = dpp0<DPP_SHR1>(p_); tz = dpp0<DPP_SHR1>(z_); p_ += tp; z_ = __builtin_fma(c.lv[0], tz, z_);
        tp = dpp0<DPP_SHR2>(p_); tz = dpp0<DPP_SHR2>(z_); p_ += tp; z_ = __builtin_fma(c.lv[1], tz, z_);
        tp = dpp0<DPP_SHR4>(p_); tz = dpp0<DPP_SHR4>(z_); p_ += tp; z_ = __builtin_fma(c.lv[2], tz, z_);
        tp = dpp0<DPP_SHR8>(p_); tz = dpp0<DPP_SHR8>(z_); p_ += tp; z_ = __builtin_fma(c.lv[3], tz, z_);
    }
    S = dpp0<DPP_RBC15>(p_);
    const double LIp = dpp0<DPP_SHR1>(z_);
    const double lex = p_ - lp[SPL - 1];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const double incl = lex + lp[k];
        const double Z = (k == 0) ? LIp : __builtin_fma(c.cumA[k - 1 < 0 ? 0 : k - 1], LIp, w[k - 1 < 0 ? 0 : k - 1]);
        out[k] = e[k] * __builtin_fma(c.dc[k], x[k], __builtin_fma(c.g[k], S, __builtin_fma(c.cg[k], incl, Z)));
    }
}

// one position of four backward chains (position p = state 16 SPL - 1 - p):  out = T (e o b);  Sw = sum (e o b) (float accuracy)
template <int SPL>
__device__ __forceinline__ void ss4_bwd_step(const S4BwdC<SPL> &c, const double (&bv)[SPL], const double (&e)[SPL],
                                             double (&out)[SPL], float &Sw) {
    double w[SPL], lg[SPL], u[SPL];
    float lf[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) w[k] = e[k] * bv[k];
    lg[0] = c.g[0] * w[0];
    u[0] = w[0];
    lf[0] = (float)w[0];
#pragma unroll
    for (int k = 1; k < SPL; ++k) {
        lg[k] = __builtin_fma(c.g[k], w[k], lg[k - 1]);
        u[k] = __builtin_fma(c.a[k], u[k - 1], w[k]);
        lf[k] = lf[k - 1] + (float)w[k];
    }
    double p_ = lg[SPL - 1], z_ = u[SPL - 1];
    float f_ = lf[SPL - 1];
    {
        double tp, tz;
        float tf;
        tp = dpp0<DPP_SHR1>(p_); tz = dpp0<DPP_SHR1>(z_); tf = dpp0<DPP_SHR1>(f_); p_ += tp; z_ = __builtin_fma(c.lv[0], tz, z_); f_ += tf;
        tp = dpp0<DPP_SHR2>(p_); tz = dpp0<DPP_SHR2>(z_); tf = dpp0<DPP_SHR2>(f_); p_ += tp; z_ = __builtin_fma(c.lv[1], tz, z_); f_ += tf;
        tp = dpp0<DPP_SHR4>(p_); tz = dpp0<DPP_SHR4>(z_); tf = dpp0<DPP_SHR4>(f_); p_ += tp; z_ = __builtin_fma(c.lv[2], tz, z_); f_ += tf;
        tp = dpp0<DPP_SHR8>(p_); tz = dpp0<DPP_SHR8>(z_); tf = dpp0<DPP_SHR8>(f_); p_ += tp; z_ = __builtin_fma(c.lv[3], tz, z_); f_ += tf;
    }
    const double Gtot = dpp0<DPP_RBC15>(p_);
    Sw = dpp0<DPP_RBC15>(f_);
    const double LIp = dpp0<DPP_SHR1>(z_);
    const double lexg = p_ - lg[SPL - 1];
    const float lexf = f_ - lf[SPL - 1];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const double inclG = lexg + lg[k];
        const double inclW = (double)(lexf + lf[k]);
        const double V = (k == 0) ? LIp : __builtin_fma(c.cumA[k - 1 < 0 ? 0 : k - 1], LIp, u[k - 1 < 0 ? 0 : k - 1]);
        out[k] = __builtin_fma(c.dc[k], w[k], (Gtot - inclG) + __builtin_fma(c.c0, inclW, c.b[k] * V));
    }
}

// sum over the 16 lanes of a DPP row, every lane of the row receives it
__device__ __forceinline__ double row_total(double v) {
    v += dpp0<DPP_SHR1>(v);
    v += dpp0<DPP_SHR2>(v);
    v += dpp0<DPP_SHR4>(v);
    v += dpp0<DPP_SHR8>(v);
    return dpp0<DPP_RBC15>(v);
}
// does any lane of this lane's DPP row raise the flag?
__device__ __forceinline__ bool row_any(bool f, int lane) {
    const unsigned long long bal = __ballot(f);
    return ((bal >> (lane & 48)) & 0xFFFFull) != 0;
}
__device__ __forceinline__ int row_pick(int v, int lane, int idx) {       // v of lane idx (0..15) of this lane's row
    return __builtin_amdgcn_ds_bpermute(((lane & 48) + idx) << 2, v);
}

template <int SPL, bool RERUN>
__device__ __forceinline__ void ss4_forward_wave(const SsArgs &a, const double *sE, int cbase, int lane) {
    constexpr int MS4 = 16 * SPL;
    const int M = a.M, Mp = a.Mp, pass = a.pass;
    const int r = lane >> 4, q = lane & 15;
    const int c = cbase + r;
    const bool exists = c < a.nchunks;
    const Chunk ch = a.chunks[exists ? c : a.nchunks - 1];
    float *end_cur = a.ends_f + ((size_t)(pass & 1) * a.nchunks + (exists ? c : 0)) * Mp;
    const float *end_prev = a.ends_f + ((size_t)((pass + 1) & 1) * a.nchunks + (exists ? c : 0)) * Mp;
    int st[SPL];
    bool live[SPL], stor[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) { st[k] = q * SPL + k; live[k] = st[k] < M; stor[k] = st[k] < Mp; }
    bool active = exists;
    bool copy_end = false;
    if (RERUN && ch.first && !a.full_f) { copy_end = active; active = false; }
    double x[SPL];
    {
        const float *src = (ch.first || !RERUN) ? a.pi_f : a.ends_f + ((size_t)((pass + 1) & 1) * a.nchunks + (c - 1)) * Mp;
#pragma unroll
        for (int k = 0; k < SPL; ++k) x[k] = (exists && live[k]) ? (double)src[st[k]] : 0.0;
    }
    if (RERUN && !a.full_f) {
        bool diff = false;
        if (active) {
#pragma unroll
            for (int k = 0; k < SPL; ++k)
                if (live[k]) {
                    const float u = a.used_f[(size_t)c * Mp + st[k]];
                    if (!(fabsf((float)x[k] - u) <= a.eps_f * fabsf(u))) diff = true;
                }
        }
        const bool rd_ = row_any(diff, lane);
        if (active && !rd_) { copy_end = true; active = false; }
    }
    if (copy_end) {
#pragma unroll
        for (int k = 0; k < SPL; ++k) if (stor[k]) end_cur[st[k]] = end_prev[st[k]];
    }
    if (!__any(active)) return;
    if (active) {
#pragma unroll
        for (int k = 0; k < SPL; ++k) if (stor[k]) a.used_f[(size_t)c * Mp + st[k]] = (float)x[k];
        if (q == 0) a.changed_f[pass] = 1;
        if (ch.first) {
#pragma unroll
            for (int k = 0; k < SPL; ++k) if (stor[k]) a.alpha[(size_t)ch.base * Mp + st[k]] = (float)x[k];
            if (q == 0) a.cnorm[ch.base] = 1.0;
        }
    }
    S4FwdC<SPL> cst;
    ss4_load_fwd<SPL>(a, q, lane, cst);
    const int2 *rd = a.rowdesc + ch.base + ch.r0 + 1;          // descriptor of this chain's iteration j (row ell = r0 + 1 + j)
    const int nrows = active ? ch.r1 - ch.r0 : 0;
    int2 dwin = rd[q];                                          // window of 16 descriptors: iterations wbase .. wbase + 15
    int wbase = 0, j = 0;
    int slot = row_pick(dwin.x, lane, 0), rem = row_pick(dwin.y, lane, 0);
    bool span1 = rem == 1;
    double e[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) e[k] = sE[(size_t)slot * MS4 + st[k]];
    float *arow = a.alpha + (size_t)(ch.base + ch.r0) * Mp;    // row ell - 1 of iteration j is arow + j Mp
    double *crow = a.cnorm + ch.base + ch.r0;
    bool f0 = true, done = nrows == 0, merged = false;
    double xfin[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) xfin[k] = x[k];
    while (__any(!done)) {
        double y[SPL], S;
        ss4_fwd_step<SPL>(cst, x, e, y, S);
        const bool fr = f0 && !done;
        if (__any(fr)) {
            // first position of a row: the sum of the incoming vector finishes the chain's PREVIOUS row (chains_ss.hpp)
            const double inv = rcp_f64(S);
            double fb[SPL];
            float an[SPL];
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                const double xs = x[k] * inv;
                an[k] = live[k] ? fmaxf((float)xs, 1e-10f) : 0.f;
                fb[k] = (j > 0) ? (double)an[k] - xs : 0.0;
            }
            const bool st_ = fr && j > 0;
            if (RERUN && !a.full_f) {
                const bool chk = st_ && (j & 15) == 0 && j >= 16;
                if (__any(chk)) {
                    bool bad = false;
                    if (chk) {
#pragma unroll
                        for (int k = 0; k < SPL; ++k)
                            if (live[k]) {
                                const float old = arow[(size_t)j * Mp + st[k]];
                                if (!(fabsf(an[k] - old) <= a.eps_f * fabsf(old))) bad = true;
                            }
                    }
                    const bool rb = row_any(bad, lane);
                    if (chk && !rb) { merged = true; done = true; }
                }
            }
            if (st_ && !done) {
#pragma unroll
                for (int k = 0; k < SPL; ++k) if (stor[k]) arow[(size_t)j * Mp + st[k]] = an[k];
                if (q == 0) crow[j] = S;
            }
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                double ed = e[k] * cst.d[k];
                double yk = y[k];
                if (span1) { const double edf = (double)(float)ed; yk = __builtin_fma(edf - ed, x[k], yk); ed = edf; }
                const double xn = __builtin_fma(ed, fb[k], yk * inv);
                y[k] = fr ? xn : y[k];
            }
        }
#pragma unroll
        for (int k = 0; k < SPL; ++k) x[k] = y[k];
        f0 = false;
        rem -= 1;
        const bool adv = rem == 0 && !done;
        if (__any(adv)) {
            if (adv) ++j;
            const bool fin = adv && j == nrows;
            if (fin) {
                done = true;
#pragma unroll
                for (int k = 0; k < SPL; ++k) xfin[k] = x[k];
            }
            const bool more = adv && !fin;
            if (__any(more && j - wbase == 16)) {
                if (more && j - wbase == 16) { wbase = j; dwin = rd[j + q]; }
            }
            const int ns = row_pick(dwin.x, lane, (j - wbase) & 15), nsp = row_pick(dwin.y, lane, (j - wbase) & 15);
            if (more) {
                slot = ns; rem = nsp; span1 = nsp == 1; f0 = true;
#pragma unroll
                for (int k = 0; k < SPL; ++k) e[k] = sE[(size_t)slot * MS4 + st[k]];
            }
        }
    }
    if (active) {
        if (merged) {
#pragma unroll
            for (int k = 0; k < SPL; ++k) if (stor[k]) end_cur[st[k]] = end_prev[st[k]];
        }
    }
    {
        double part = 0.0;
#pragma unroll
        for (int k = 0; k < SPL; ++k) part += xfin[k];
        const double S = row_total(part);
        if (active && !merged) {
            const double inv = 1.0 / S;
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                const float an = live[k] ? fmaxf((float)(xfin[k] * inv), 1e-10f) : 0.f;
                if (stor[k]) { a.alpha[(size_t)(ch.base + ch.r1) * Mp + st[k]] = an; end_cur[st[k]] = an; }
            }
            if (q == 0) a.cnorm[ch.base + ch.r1] = S;
        }
    }
}

template <int SPL, bool RERUN>
__device__ __forceinline__ void ss4_backward_wave(const SsArgs &a, const double *sE, int cbase, int lane) {
    constexpr int MS4 = 16 * SPL;
    const int M = a.M, Mp = a.Mp, pass = a.pass;
    const int r = lane >> 4, q = lane & 15;
    const int c = cbase + r;
    const bool exists = c < a.nchunks_b;
    const Chunk ch = a.chunks_b[exists ? c : a.nchunks_b - 1];
    double *end_cur = a.ends_b + ((size_t)(pass & 1) * a.nchunks_b + (exists ? c : 0)) * Mp;
    const double *end_prev = a.ends_b + ((size_t)((pass + 1) & 1) * a.nchunks_b + (exists ? c : 0)) * Mp;
    int st[SPL];
    bool live[SPL], stor[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) { st[k] = MS4 - 1 - (q * SPL + k); live[k] = st[k] < M; stor[k] = st[k] < Mp; }
    bool active = exists;
    bool copy_end = false;
    if (RERUN && ch.last && !a.full_b) { copy_end = active; active = false; }
    double b[SPL];
    {
        const bool fresh = ch.last || !RERUN;
        const double *src = a.ends_b + ((size_t)((pass + 1) & 1) * a.nchunks_b + ((fresh || !exists) ? (exists ? c : 0) : c + 1)) * Mp;
#pragma unroll
        for (int k = 0; k < SPL; ++k) b[k] = (exists && live[k]) ? (fresh ? 1.0 / (double)M : src[st[k]]) : 0.0;
    }
    if (RERUN && !a.full_b) {
        bool diff = false;
        if (active) {
#pragma unroll
            for (int k = 0; k < SPL; ++k)
                if (live[k]) {
                    const double u = a.used_b[(size_t)c * Mp + st[k]];
                    if (!(fabs(b[k] - u) <= a.eps_b * fabs(u))) diff = true;
                }
        }
        const bool rd_ = row_any(diff, lane);
        if (active && !rd_) { copy_end = true; active = false; }
    }
    if (copy_end) {
#pragma unroll
        for (int k = 0; k < SPL; ++k) if (stor[k]) end_cur[st[k]] = end_prev[st[k]];
    }
    if (!__any(active)) return;
    if (active) {
#pragma unroll
        for (int k = 0; k < SPL; ++k) if (stor[k]) a.used_b[(size_t)c * Mp + st[k]] = b[k];
        if (q == 0) a.changed_b[pass] = 1;
    }
    S4BwdC<SPL> cst;
    ss4_load_bwd<SPL>(a, q, lane, cst);
    const int2 *rd = a.rowdesc + ch.base + ch.r1;               // descriptor of this chain's iteration j (row ell = r1 - j) is rd[-j]
    const int nrows = active ? ch.r1 - ch.r0 : 0;
    int2 dwin = rd[-q];
    int wbase = 0, j = 0;
    int slot = row_pick(dwin.x, lane, 0), rem = row_pick(dwin.y, lane, 0);
    double e[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) e[k] = sE[(size_t)slot * MS4 + st[k]];
    double *brow = a.beta + (size_t)(ch.base + ch.r1) * Mp;     // row ell of iteration j is brow - j Mp
    bool f0 = true, done = nrows == 0, merged = false;
    double bfin[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) bfin[k] = b[k];
    while (__any(!done)) {
        const bool fr = f0 && !done;
        if (__any(fr)) {
            // beta[ell] in the running scale is the vector that ENTERS the row
            if (RERUN && !a.full_b) {
                const bool chk = fr && (j & 15) == 0 && j >= 16;
                if (__any(chk)) {
                    bool bad = false;
                    if (chk) {
#pragma unroll
                        for (int k = 0; k < SPL; ++k)
                            if (live[k]) {
                                const double old = brow[-(ptrdiff_t)j * Mp + st[k]];
                                if (!(fabs(b[k] - old) <= a.eps_b * fabs(old))) bad = true;
                            }
                    }
                    const bool rb = row_any(bad, lane);
                    if (chk && !rb) { merged = true; done = true; }
                }
            }
            if (fr && !done) {
#pragma unroll
                for (int k = 0; k < SPL; ++k) if (stor[k]) brow[-(ptrdiff_t)j * Mp + st[k]] = b[k];
            }
        }
        double y[SPL];
        float Sw;
        ss4_bwd_step<SPL>(cst, b, e, y, Sw);
        {
            // per-row running scale (first position of a row only): reciprocal of the float sum of e o beta
            const float invf = __builtin_amdgcn_rcpf(Sw);
            const double inv = fr ? (double)invf : 1.0;
#pragma unroll
            for (int k = 0; k < SPL; ++k) b[k] = y[k] * inv;
        }
        f0 = false;
        rem -= 1;
        const bool adv = rem == 0 && !done;
        if (__any(adv)) {
            if (adv) ++j;
            const bool fin = adv && j == nrows;
            if (fin) {
                done = true;
#pragma unroll
                for (int k = 0; k < SPL; ++k) bfin[k] = b[k];
            }
            const bool more = adv && !fin;
            if (__any(more && j - wbase == 16)) {
                if (more && j - wbase == 16) { wbase = j; dwin = rd[-j - q]; }
            }
            const int ns = row_pick(dwin.x, lane, (j - wbase) & 15), nsp = row_pick(dwin.y, lane, (j - wbase) & 15);
            if (more) {
                slot = ns; rem = nsp; f0 = true;
#pragma unroll
                for (int k = 0; k < SPL; ++k) e[k] = sE[(size_t)slot * MS4 + st[k]];
            }
        }
    }
    if (active && merged) {
#pragma unroll
        for (int k = 0; k < SPL; ++k) if (stor[k]) end_cur[st[k]] = end_prev[st[k]];
    }
    {
        double part = 0.0;
#pragma unroll
        for (int k = 0; k < SPL; ++k) part += live[k] ? bfin[k] : 0.0;
        const double S = row_total(part);
        if (active && !merged) {
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                const double bf = live[k] ? bfin[k] / S : 0.0;       // beta /= beta.sum()  (seeds gamma[:,0], hmm.cpp:150)
                if (stor[k]) { end_cur[st[k]] = bf; if (ch.first) a.beta[(size_t)ch.base * Mp + st[k]] = bf; }
            }
        }
    }
}

// One workgroup = 4 wavefronts: wavefronts 0, 1 run forward chunks 8 blk .. 8 blk + 7 (four each), wavefronts 2, 3 the same
// chunks backward; one LDS copy of the emission table ([K][16 SPL], every key: M <= 64).
template <int SPL>
__global__ __launch_bounds__(256) void k_chain_ss4(SsArgs a) {
    constexpr int MS4 = 16 * SPL;
    extern __shared__ __attribute__((aligned(16))) double ss4_lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool fwd = w < 2;
    const bool idle_f = a.mode_f == 3 || (a.mode_f == 1 && a.changed_f[a.pass - 1] == 0);
    const bool idle_b = a.mode_b == 3 || (a.mode_b == 1 && a.changed_b[a.pass - 1] == 0);
    if (idle_f && idle_b) return;
    for (int idx = tid; idx < a.K * MS4; idx += 256) ss4_lds[idx] = a.E[idx];
    __syncthreads();
    const int cbase = 8 * blockIdx.x + 4 * (w & 1);
    if (cbase >= a.nchunks) return;
    if (fwd) {
        if (idle_f) return;
        if (a.mode_f == 0) ss4_forward_wave<SPL, false>(a, ss4_lds, cbase, lane);
        else ss4_forward_wave<SPL, true>(a, ss4_lds, cbase, lane);
    } else {
        if (idle_b) return;
        if (a.mode_b == 0) ss4_backward_wave<SPL, false>(a, ss4_lds, cbase, lane);
        else ss4_backward_wave<SPL, true>(a, ss4_lds, cbase, lane);
    }
}

// Unit-test entry: one position of both layouts-of-four on nvec vectors (vector v runs on row v & 3 of wavefront v >> 2)
template <int SPL>
__global__ __launch_bounds__(64) void k_ss4_apply(SsArgs a, const double *__restrict__ x, const double *__restrict__ e,
                                                  double *__restrict__ out_f, double *__restrict__ out_b, int nvec) {
    constexpr int MS4 = 16 * SPL;
    const int lane = threadIdx.x, q = lane & 15;
    const int v = min(4 * blockIdx.x + (lane >> 4), nvec - 1);
    S4FwdC<SPL> cf;
    S4BwdC<SPL> cb;
    ss4_load_fwd<SPL>(a, q, lane, cf);
    ss4_load_bwd<SPL>(a, q, lane, cb);
    double xf[SPL], ef[SPL], xb[SPL], eb[SPL], yf[SPL], yb[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int p = q * SPL + k, s = MS4 - 1 - p;
        xf[k] = x[(size_t)v * MS4 + p]; ef[k] = e[(size_t)v * MS4 + p];
        xb[k] = x[(size_t)v * MS4 + s]; eb[k] = e[(size_t)v * MS4 + s];
    }
    double S; float Sw;
    ss4_fwd_step<SPL>(cf, xf, ef, yf, S);
    ss4_bwd_step<SPL>(cb, xb, eb, yb, Sw);
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int p = q * SPL + k, s = MS4 - 1 - p;
        out_f[(size_t)v * MS4 + p] = yf[k];
        out_b[(size_t)v * MS4 + s] = yb[k];
    }
}

}  // namespace smcpp_dev
