// Cold parameter preparation ON THE DEVICE (SURVEY.md §8(a) rows A6 + A10): per hidden state the conditioned SFS
// (src/conditioned_sfs.cpp:13-148 with the integrals of src/piecewise_constant_rate_function.cpp:214-334), incorporate_theta
// (src/inference_manager.cpp:389-407 / conditioned_sfs.h) and the emission table of every block key
// (src/inference_manager.cpp:409-482), with forward-mode derivatives: the role of the reference's `adouble`
// (include/common.h:22-25).  prep.hpp holds the host implementation of the same mathematics (O(pieces n^2) factored form); this
// file evaluates it with one workgroup per (hidden state, derivative direction):
//
//   k_prep_tables   the prefix / suffix sums over the model pieces (CsfsPieceTables of prep.hpp): every (rate, piece) term by its
//                   own thread, then one thread per rate runs the linear recurrence over the pieces
//   k_prep_csfs     workgroup (h, d): hoisted exponentials of the state's pieces -> one thread per (lambda, rate) pair
//                   accumulates the double integrals -> compensated contractions with X0 / X2 -> Moran back-transformation ->
//                   the "below" part -> incorporate_theta -> emission vectors of all keys for state h, written straight into
//                   the layouts the chains and the statistics read
//
// A number with derivatives is a (value, ONE directional derivative) pair: direction d lives in workgroup (h, d), so no
// register arrays and any number of directions; the values are recomputed by every direction's workgroup (the chip is
// otherwise idle while the parameters are prepared).  Arithmetic mirrors prep.hpp operation by operation (FMA contraction off),
// so the two agree to the last bits of the transcendental functions; the phases are __host__ __device__ so that the CPU test
// suite runs the same code through emulate_*() without a GPU.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#pragma clang fp contract(off)

#define SMCPP_HD __host__ __device__ inline

namespace smcpp_dev {

// ---- value + N directional derivatives (mirrors smcpp_host::Dual, prep.hpp, operation by operation) ------------------------------
// N directions ride in one thread: the value part (where the exponentials are) is evaluated once for all of them.
template <int N>
struct DN {
    double v, d[N];
    SMCPP_HD DN() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
    SMCPP_HD DN(double x) : v(x) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
};
#define SMCPP_DN_LOOP _Pragma("unroll") for (int i_ = 0; i_ < N; ++i_)
template <int N> SMCPP_HD DN<N> operator+(const DN<N> &a, const DN<N> &b) { DN<N> r; r.v = a.v + b.v; SMCPP_DN_LOOP r.d[i_] = a.d[i_] + b.d[i_]; return r; }
template <int N> SMCPP_HD DN<N> operator-(const DN<N> &a, const DN<N> &b) { DN<N> r; r.v = a.v - b.v; SMCPP_DN_LOOP r.d[i_] = a.d[i_] - b.d[i_]; return r; }
template <int N> SMCPP_HD DN<N> operator*(const DN<N> &a, const DN<N> &b) { DN<N> r; r.v = a.v * b.v; SMCPP_DN_LOOP r.d[i_] = a.d[i_] * b.v + a.v * b.d[i_]; return r; }
template <int N> SMCPP_HD DN<N> operator/(const DN<N> &a, const DN<N> &b) { DN<N> r; const double ib = 1 / b.v; r.v = a.v * ib; SMCPP_DN_LOOP r.d[i_] = (a.d[i_] - r.v * b.d[i_]) * ib; return r; }
template <int N> SMCPP_HD DN<N> operator-(const DN<N> &a) { DN<N> r; r.v = -a.v; SMCPP_DN_LOOP r.d[i_] = -a.d[i_]; return r; }
template <int N> SMCPP_HD DN<N> operator+(const DN<N> &a, double b) { DN<N> r(a); r.v += b; return r; }
template <int N> SMCPP_HD DN<N> operator+(double b, const DN<N> &a) { DN<N> r(a); r.v += b; return r; }
template <int N> SMCPP_HD DN<N> operator-(const DN<N> &a, double b) { DN<N> r(a); r.v -= b; return r; }
template <int N> SMCPP_HD DN<N> operator-(double b, const DN<N> &a) { DN<N> r = -a; r.v += b; return r; }
template <int N> SMCPP_HD DN<N> operator*(const DN<N> &a, double b) { DN<N> r; r.v = a.v * b; SMCPP_DN_LOOP r.d[i_] = a.d[i_] * b; return r; }
template <int N> SMCPP_HD DN<N> operator*(double b, const DN<N> &a) { return a * b; }
template <int N> SMCPP_HD DN<N> operator/(const DN<N> &a, double b) { return a * (1.0 / b); }
template <int N> SMCPP_HD DN<N> operator/(double b, const DN<N> &a) { DN<N> r; r.v = b / a.v; SMCPP_DN_LOOP r.d[i_] = -r.v * a.d[i_] / a.v; return r; }
template <int N> SMCPP_HD DN<N> &operator+=(DN<N> &a, const DN<N> &b) { a.v += b.v; SMCPP_DN_LOOP a.d[i_] += b.d[i_]; return a; }
template <int N> SMCPP_HD DN<N> &operator*=(DN<N> &a, const DN<N> &b) { a = a * b; return a; }
template <int N> SMCPP_HD DN<N> &operator/=(DN<N> &a, const DN<N> &b) { a = a / b; return a; }
SMCPP_HD double m_exp(double x) { return ::exp(x); }
SMCPP_HD double m_expm1(double x) { return ::expm1(x); }
SMCPP_HD double m_log(double x) { return ::log(x); }
template <int N> SMCPP_HD DN<N> m_exp(const DN<N> &a) { DN<N> r; r.v = ::exp(a.v); SMCPP_DN_LOOP r.d[i_] = a.d[i_] * r.v; return r; }
template <int N> SMCPP_HD DN<N> m_expm1(const DN<N> &a) { DN<N> r; r.v = ::expm1(a.v); const double e = ::exp(a.v); SMCPP_DN_LOOP r.d[i_] = a.d[i_] * e; return r; }
template <int N> SMCPP_HD DN<N> m_log(const DN<N> &a) { DN<N> r; r.v = ::log(a.v); SMCPP_DN_LOOP r.d[i_] = a.d[i_] / a.v; return r; }
SMCPP_HD double sval(double x) { return x; }
template <int N> SMCPP_HD double sval(const DN<N> &x) { return x.v; }
// number of directions a scalar type carries, and its q-th derivative
template <typename S> struct NDir { static constexpr int value = 0; };
template <int N> struct NDir<DN<N>> { static constexpr int value = N; };
SMCPP_HD double sder(double, int) { return 0.0; }
template <int N> SMCPP_HD double sder(const DN<N> &x, int q) { return x.d[q]; }
// load entry i of an array with derivative planes [nder][stride]: the directions of group `dir` are dir N .. dir N + N - 1
SMCPP_HD void ldp(double &o, const double *v, const double *, int, int, int, int i) { o = v[i]; }
template <int N> SMCPP_HD void ldp(DN<N> &o, const double *v, const double *d, int stride, int dir, int nder, int i) {
    o.v = v[i];
    SMCPP_DN_LOOP { const int q = dir * N + i_; o.d[i_] = (d && q < nder) ? d[(size_t)q * stride + i] : 0.0; }
}
// cascaded TwoSum accumulation of prep.hpp's accurate_sum: value compensated, derivatives plain sums
template <typename S> struct AccT { double hi = 0.0, lo = 0.0; };
template <int N> struct AccT<DN<N>> { double hi = 0.0, lo = 0.0, d[N]; SMCPP_HD AccT() { for (int i = 0; i < N; ++i) d[i] = 0.0; } };
typedef AccT<double> AccD;
SMCPP_HD void acc_add(AccD &a, double x) { const double t = a.hi + x, z = t - a.hi; a.lo += (a.hi - (t - z)) + (x - z); a.hi = t; }
template <int N> SMCPP_HD void acc_add(AccT<DN<N>> &a, const DN<N> &x) {
    const double t = a.hi + x.v, z = t - a.hi; a.lo += (a.hi - (t - z)) + (x.v - z); a.hi = t;
    SMCPP_DN_LOOP a.d[i_] += x.d[i_];
}
SMCPP_HD void acc_get(const AccD &a, double &o) { o = a.hi + a.lo; }
template <int N> SMCPP_HD void acc_get(const AccT<DN<N>> &a, DN<N> &o) { o.v = a.hi + a.lo; SMCPP_DN_LOOP o.d[i_] = a.d[i_]; }

SMCPP_HD long nC2(long n) { return n * (n - 1) / 2; }

// ---- inputs -----------------------------------------------------------------------------------------------------------------
// The rate function after the hidden states were inserted as break points (RateFunctionT's constructor, prep.hpp; host, O(K)):
// K pieces, ts [K+1] (ts[K] = inf), ada = 1 / a and the cumulative hazard Rrng [K+1] with their derivative planes
// [nder][K] / [nder][K+1], hs_indices [M+1], the average coalescence time per state [M] (+ planes).
struct PrepModel {
    int K = 0, n = 0, M = 0, nder = 0;
    const double *ts = nullptr, *ada_v = nullptr, *ada_d = nullptr, *R_v = nullptr, *R_d = nullptr, *act_v = nullptr, *act_d = nullptr;
    const int *hsi = nullptr;
    double theta = 0.0, alpha = 1.0;
};
// n-only tables (CsfsTables of prep.hpp, row-major) and the keys' marginalisation bins in CSR form (OnePopPrep::bins_for)
struct PrepStatic {
    const double *X0 = nullptr, *X2 = nullptr, *M0 = nullptr, *M1 = nullptr, *U0 = nullptr, *U2 = nullptr;
    int Kk = 0;                      // keys the table is prepared for (the global list of a sharded manager)
    const int *kind = nullptr;       // [Kk] 0 = bins, 1 = missing (all ones), 2 / 3 = reduced key, even / odd a
    const int *boff = nullptr, *bidx = nullptr;
    const double *bw = nullptr;
    const int *local = nullptr;      // [Kk] local key id (row of the E the statistics read) or -1
    const int *slot = nullptr;       // [Kk] key slot of the scan chains' emission table or -1
    const int *maxspan = nullptr;    // [Kk] longest span of a row of the key that the scan chains expand step by step (or 1)
};
// where the results go.  Planes: direction d of an array X of `sz` doubles lives at X_d + d * sz.
struct PrepOut {
    double *sfs_v = nullptr, *sfs_d = nullptr;       // [M][C], C = 3 (n+1): InferenceManager::emission
    double *Eg_v = nullptr, *Eg_d = nullptr;         // [Kk][M] every prepared key (getters, Q on reduced statistics)
    double *El_v = nullptr;                          // [K_local][Mp] what the statistics kernels read (may be null)
    double *Es_v = nullptr;                          // [K_local][MS] by key slot: the scan chains' table (may be null)
    int Mp = 0, MS = 0;
    // one word per condition (plain stores of 1, no read-modify-write): [0] an emission entry left (0, 1], [1] a conditioned
    // SFS is not a probability distribution, [2] an emission entry is so small that `span` scan steps would underflow
    int *flags = nullptr;
};
// Tables of k_prep_tables, per direction GROUP g (a group = the directions one scalar carries; one group for plain doubles), piece-
// major so that the threads of the consuming kernel read neighbouring words:   Ssuf [g][K][n]      Ppre [g][K+1][n+1]
template <typename S> struct Tables {
    S *Ssuf = nullptr, *Ppre = nullptr;
    SMCPP_HD static size_t per_group(int n, int K) { return (size_t)n * K + (size_t)(n + 1) * (K + 1); }
    SMCPP_HD void carve(S *base, int n, int K, int ng) {
        Ssuf = base; base += (size_t)ng * n * K;
        Ppre = base;
    }
};

template <typename S> SMCPP_HD S ld_ada(const PrepModel &pm, int dir, int m) { S r; ldp(r, pm.ada_v, pm.ada_d, pm.K, dir, pm.nder, m); return r; }
template <typename S> SMCPP_HD S ld_R(const PrepModel &pm, int dir, int m) { S r; ldp(r, pm.R_v, pm.R_d, pm.K + 1, dir, pm.nder, m); return r; }

// ---- k_prep_tables: rate r (0 .. n-1: the "above" rates C(j,2); n .. 2n: the "below" rates C(j,2) - 1) --------------------------
// phase 1: the term of piece m (g, and for the above rates the factor f of the recurrence Ssuf[m-1] = g_m + f_m Ssuf[m])
template <typename S>
SMCPP_HD void tables_term(const PrepModel &pm, int dir, int r, int m, S &g, S &f) {
    const int n = pm.n;
    f = S(0.0);
    if (r < n) {
        if (m == 0) { g = S(0.0); return; }                     // (slot m holds the term of piece m; piece 0 has none)
        const double rate = (double)nC2(r + 2);
        const S ad = ld_ada<S>(pm, dir, m);
        if (pm.ts[m + 1] < INFINITY) {
            const S em = m_expm1(-rate * ad * (pm.ts[m + 1] - pm.ts[m]));
            g = -em / (ad * rate);
            f = 1.0 + em;
        } else g = 1.0 / (ad * rate);
        return;
    }
    const long ratel = nC2(r - n + 2) - 1;
    if (ratel == 0) { g = S(pm.ts[m + 1]); return; }
    const double rate = (double)ratel;
    const S ad = ld_ada<S>(pm, dir, m);
    g = m_exp(-rate * ld_R<S>(pm, dir, m));
    if (pm.ts[m + 1] < INFINITY) g *= -m_expm1(-rate * ad * (pm.ts[m + 1] - pm.ts[m]));
    g /= ad * rate;
}
// phase 2 (one thread per rate): the recurrence over the K terms G / F (the workgroup's scratch), results to the tables
template <typename S>
SMCPP_HD void tables_scan(const PrepModel &pm, int dir, const Tables<S> &tb, int r, const S *G, const S *F) {
    const int K = pm.K, n = pm.n;
    if (r < n) {
        S *O = tb.Ssuf + (size_t)dir * K * n + r;
        S acc(0.0);                              // Ssuf[K-1] = 0
        for (int m = K - 1; m >= 1; --m) {
            O[(size_t)m * n] = acc;
            if (pm.ts[m + 1] < INFINITY) acc = G[m] + F[m] * acc; else acc = G[m];
        }
        O[0] = acc;
        return;
    }
    const int t = r - n;
    S *O = tb.Ppre + (size_t)dir * (K + 1) * (n + 1) + t;
    O[0] = S(0.0);
    if (nC2(t + 2) - 1 == 0) {                   // (rate 0: the prefix sums are the break points themselves)
        for (int m = 0; m < K; ++m) O[(size_t)(m + 1) * (n + 1)] = G[m];
        return;
    }
    S acc(0.0);
    for (int m = 0; m < K; ++m) { acc = acc + G[m]; O[(size_t)(m + 1) * (n + 1)] = acc; }
}

// ---- k_prep_csfs: the work of one workgroup (hidden state h, direction group dir) ----------------------------------------------
// `sh` = the workgroup's scratch (LDS on the device): Ca [(n+1) n], A, A1, B, El [n+1], ert, e1 [n], tmp0, tmp2, below [n+1],
// out [3 (n+1)], e2 [2], ldn [1]
template <typename S> struct CsfsScratch {
    S *Ca, *A, *A1, *B, *El, *ert, *e1, *tmp0, *tmp2, *below, *out, *e2, *ldn;
    SMCPP_HD static size_t count(int n) { return (size_t)(n + 1) * n + 4 * (n + 1) + 2 * n + 3 * (n + 1) + 3 * (n + 1) + 3; }
    SMCPP_HD void carve(S *base, int n) {
        Ca = base; base += (size_t)(n + 1) * n;
        A = base; base += n + 1; A1 = base; base += n + 1; B = base; base += n + 1; El = base; base += n + 1;
        ert = base; base += n; e1 = base; base += n;
        tmp0 = base; base += n + 1; tmp2 = base; base += n + 1; below = base; base += n + 1;
        out = base; base += 3 * (n + 1); e2 = base; base += 2; ldn = base;
    }
};

template <typename S> struct CsfsCtx {
    PrepModel pm;
    PrepStatic ps;
    PrepOut po;
    Tables<S> tb;
    int h, dir;
    CsfsScratch<S> sh;
    SMCPP_HD S ada(int m) const { return ld_ada<S>(pm, dir, m); }
    SMCPP_HD S R(int m) const { return ld_R<S>(pm, dir, m); }
    SMCPP_HD S log_denom_compute() const {
        const S Rh = R(pm.hsi[h]), Rh1 = R(pm.hsi[h + 1]);
        S ldn = -Rh;
        if (sval(Rh1) != INFINITY) ldn = ldn + m_log(-m_expm1(-(Rh1 - Rh)));
        return ldn;
    }
    SMCPP_HD S log_denom() const { return sh.ldn[0]; }
};

template <typename S>
SMCPP_HD S below_helper(long rate, double tsm, double tsm1, const S &ad, const S &Rr, const S &log_denom) {
    const long l1r = 1 + rate;
    const double l1rinv = 1.0 / (double)l1r;
    const S adadiff = ad * (tsm1 - tsm);
    if (rate == 0) {
        if (tsm1 == INFINITY) return m_exp(-Rr - log_denom) / ad;
        return m_exp(-Rr - log_denom) * (1.0 - m_exp(-adadiff) * (1.0 + adadiff)) / ad;
    }
    if (tsm1 == INFINITY) return m_exp(-(double)l1r * Rr - log_denom) * (1.0 - l1rinv) / ((double)rate * ad);
    return m_exp(-(double)l1r * Rr - log_denom) * (m_expm1(-(double)l1r * adadiff) * l1rinv - m_expm1(-adadiff)) / ((double)rate * ad);
}

// phase 0: clear the accumulators; one thread forms the state's log normaliser
template <typename S>
SMCPP_HD void csfs_clear(const CsfsCtx<S> &c, int t, int nt) {
    const int n = c.pm.n;
    for (int p = t; p < (n + 1) * n; p += nt) c.sh.Ca[p] = S(0.0);
    for (int p = t; p < n + 1; p += nt) c.sh.below[p] = S(0.0);
    for (int p = t; p < 3 * (n + 1); p += nt) c.sh.out[p] = S(0.0);
    if (t == nt - 1) c.sh.ldn[0] = c.log_denom_compute();
}
// phase 1 of piece m: the exponentials that depend on one index only.  Seven tables (A, A1, B, El over lambda; ert, e1 over the
// rate; the "below" integrals of the piece): table q is formed by the threads of wavefront q (mod the wavefronts there are), entry
// i by lane i, so that no wavefront runs more than one kind of exponential and the seven kinds run side by side.
template <typename S>
SMCPP_HD void csfs_piece_table(const CsfsCtx<S> &c, int m, int q, int i) {
    const PrepModel &pm = c.pm;
    const int n = pm.n, K = pm.K;
    const bool fin = pm.ts[m + 1] < INFINITY;
    if (q < 4) {
        if (i > n) return;
        const double l1 = (double)nC2(i + 2);
        if (q == 0) c.sh.A[i] = m_exp(-l1 * c.R(m) + -c.log_denom());
        else if (q == 1) { if (m + 1 < K) c.sh.A1[i] = m_exp(-l1 * c.R(m + 1) + -c.log_denom()); }
        else if (fin) {
            const S adadiff = c.ada(m) * (pm.ts[m + 1] - pm.ts[m]);
            if (q == 2) c.sh.B[i] = m_expm1(-l1 * adadiff); else c.sh.El[i] = m_exp(-l1 * adadiff);
        }
    } else if (q < 6) {
        if (i >= n || !fin) return;
        if (q == 4) c.sh.ert[i] = m_exp(-(double)nC2(i + 2) * (c.ada(m) * (pm.ts[m + 1] - pm.ts[m])));
        else c.sh.e1[i] = m_exp(-(double)nC2(i + 2) * (c.R(m + 1) - c.R(m)));
    } else {
        if (i > n) return;
        const int j = i + 2;                                   // j = 2 .. n+2
        const S Rm = c.R(m), Rm1 = c.R(m + 1);
        const S log_denom = c.log_denom();
        const S cc = -Rm - log_denom;
        S fac(1.0);
        if (m < K - 1) fac = -m_expm1(-(Rm1 - Rm));
        const S ec = m > 0 ? m_exp(cc) : S(0.0);
        const long rate = nC2(j) - 1;
        S val = below_helper<S>(rate, pm.ts[m], pm.ts[m + 1], c.ada(m), Rm, log_denom);
        if (m > 0) val = val + fac * (ec * c.tb.Ppre[((size_t)c.dir * (K + 1) + m) * (n + 1) + (j - 2)]);
        c.sh.below[j - 2] = c.sh.below[j - 2] + val;
    }
}
template <typename S>
SMCPP_HD void csfs_piece_tables(const CsfsCtx<S> &c, int m, int t, int nt) {
    const int nw = nt >= 64 ? nt / 64 : 1, w = t / 64, lane = t % 64;
    if (w >= nw) return;
    for (int q = w; q < 7; q += nw)
        for (int i = lane; i <= c.pm.n; i += 64) csfs_piece_table(c, m, q, i);
}
// phase 2 of piece m: one (lambda, rate) pair
template <typename S>
SMCPP_HD void csfs_pair(const CsfsCtx<S> &c, int m, int p) {
    const PrepModel &pm = c.pm;
    const int n = pm.n, K = pm.K;
    const int jl = p / n, jr = p % n;
    const bool fin = pm.ts[m + 1] < INFINITY;
    const S ad = c.ada(m);
    const S adadiff = ad * (pm.ts[m + 1] - pm.ts[m]);
    const S Rm = c.R(m), Rm1 = c.R(m + 1);
    const S dR = Rm1 - Rm;
    const long l1l = nC2(jl + 2), ratel = nC2(jr + 2);
    const double l1 = (double)l1l, rt = (double)ratel;
    const S A = c.sh.A[jl];
    S tgt = c.sh.Ca[p];
    if (l1l == ratel) {
        if (!fin) tgt += A / rt / rt / ad;
        else tgt += A * (1.0 - c.sh.ert[jr] * (1.0 + rt * adadiff)) / rt / rt / ad;
    } else if (!fin) tgt += A / l1 / rt / ad;
    else if (ratel < l1l)
        tgt += -A * (c.sh.B[jl] / l1 + (c.sh.ert[jr] * -m_expm1(-(l1 - rt) * adadiff) / (l1 - rt))) / rt / ad;
    else
        tgt += -A * (c.sh.B[jl] / l1 + (c.sh.El[jl] * m_expm1(-(rt - l1) * adadiff) / (l1 - rt))) / rt / ad;
    if (m + 1 < K) {
        S coef(0.0), fac(0.0);
        const long rp = l1l - ratel;
        const double rpd = (double)rp;
        if (rp == 0) { fac = dR; coef = c.sh.A1[jl]; }
        else if (rp < 0) {
            if (-rpd * sval(dR) > 20) { coef = c.sh.A1[jl]; fac = S(-1.0 / rpd); }
            else { coef = A * c.sh.e1[jr]; fac = -m_expm1(-rpd * dR) / rpd; }
        } else {
            if (-rpd * sval(Rm - Rm1) > 20) { coef = A * c.sh.e1[jr]; fac = S(1.0 / rpd); }
            else { coef = c.sh.A1[jl]; fac = m_expm1(-rpd * (Rm - Rm1)) / rpd; }
        }
        tgt += coef * c.tb.Ssuf[((size_t)c.dir * K + m) * n + jr] * fac;
    }
    c.sh.Ca[p] = tgt;
}
// phase 3: compensated contractions with X0 / X2 (threads 0..n: tmp0, n+1..2n+1: tmp2)
template <typename S>
SMCPP_HD void csfs_contract(const CsfsCtx<S> &c, int t) {
    const int n = c.pm.n;
    if (t <= n) {
        const int j = t;
        AccT<S> a;
        for (int i = 0; i < n; ++i) acc_add(a, c.sh.Ca[(size_t)j * n + i] * c.ps.X0[(size_t)i * (n + 1) + j]);
        acc_get(a, c.sh.tmp0[j]);
    } else if (t <= 2 * n + 1) {
        const int j = t - (n + 1);
        AccT<S> a;
        for (int i = 0; i < n; ++i) acc_add(a, c.sh.Ca[(size_t)(n - j) * n + i] * c.ps.X2[(size_t)i * (n + 1) + j]);
        acc_get(a, c.sh.tmp2[j]);
    }
}
// phase 4: Moran back-transformation of the "above" part, M0 / M1 contractions of the "below" part
template <typename S>
SMCPP_HD void csfs_backtransform(const CsfsCtx<S> &c, int t) {
    const int n = c.pm.n;
    if (t < n) {
        const int b = t;
        S s0(0.0), s2(0.0);
        for (int j = 0; j < n + 1; ++j) {
            s0 += c.sh.tmp0[j] * c.ps.U0[(size_t)j * n + b];
            s2 += c.sh.tmp2[j] * c.ps.U2[(size_t)j * n + b];
        }
        S o0 = c.sh.out[1 + b];
        o0 += s0;
        c.sh.out[2 * (n + 1) + b] += s2;
        S s(0.0);
        for (int j = 0; j < n + 1; ++j) s += c.sh.below[j] * c.ps.M0[(size_t)j * n + b];
        o0 += s;
        c.sh.out[1 + b] = o0;
    } else if (t < 2 * n + 1) {
        const int b = t - n;
        S s(0.0);
        for (int j = 0; j < n + 1; ++j) s += c.sh.below[j] * c.ps.M1[(size_t)j * (n + 1) + b];
        c.sh.out[(n + 1) + b] += s;
    }
}
// phase 5: incorporate_theta in three steps - (a) one thread: the total and the factor f (kept in e2[0]); (b) all threads: x *= f;
// (c) one thread: the new total, x[0] = 1 - total, the reduced-key factors; (d) all threads: the 1e-10 floor and the range check.
// The two totals are sequential sums in the host's order; the element-wise steps have no order.
template <typename S>
SMCPP_HD void csfs_theta_a(const CsfsCtx<S> &c) {
    const int C = 3 * (c.pm.n + 1);
    const S *x = c.sh.out;
    S tauh(0.0);
    for (int i = 0; i < C; ++i) tauh += x[i];
    c.sh.e2[0] = -m_expm1(-c.pm.theta * tauh) / tauh;
}
template <typename S>
SMCPP_HD void csfs_theta_b(const CsfsCtx<S> &c, int t, int nt) {
    const int C = 3 * (c.pm.n + 1);
    const S f = c.sh.e2[0];
    for (int i = t; i < C; i += nt) c.sh.out[i] *= f;
}
template <typename S>
SMCPP_HD void csfs_theta_c(const CsfsCtx<S> &c) {
    const PrepModel &pm = c.pm;
    const int C = 3 * (pm.n + 1);
    S *x = c.sh.out;
    S tot(0.0);
    for (int i = 0; i < C; ++i) tot += x[i];
    x[0] = 1.0 - tot;
    S act;
    ldp(act, pm.act_v, pm.act_d, pm.M, c.dir, pm.nder, c.h);
    if (sval(act) != sval(act)) { c.sh.e2[0] = S(1e-20); c.sh.e2[1] = S(1e-20); }
    else {
        const S le = -2.0 * pm.alpha * pm.theta * act;
        c.sh.e2[0] = m_exp(le);
        c.sh.e2[1] = -m_expm1(le);
    }
}
template <typename S>
SMCPP_HD void csfs_theta_d(const CsfsCtx<S> &c, int t, int nt) {
    const int C = 3 * (c.pm.n + 1);
    S *x = c.sh.out;
    for (int i = t; i < C; i += nt) {
        if (sval(x[i]) < 1e-10) x[i] = S(1e-10);
        const double v = sval(x[i]);
        if ((v < 0 || v > 1 || v != v) && c.po.flags) c.po.flags[1] = 1;
    }
}
// phase 6: emission vectors of key k at state h (OnePopPrep::emission_probs), and the state's conditioned SFS
template <typename S>
SMCPP_HD void csfs_emit(const CsfsCtx<S> &c, int t, int nt) {
    const PrepModel &pm = c.pm;
    constexpr int ND = NDir<S>::value;
    const int n = pm.n, C = 3 * (n + 1), M = pm.M, h = c.h;
    for (int i = t; i < C; i += nt) {
        if (c.dir == 0 && c.po.sfs_v) c.po.sfs_v[(size_t)h * C + i] = sval(c.sh.out[i]);
        if (c.po.sfs_d)
            for (int q = 0; q < ND; ++q)
                if (c.dir * ND + q < pm.nder) c.po.sfs_d[((size_t)(c.dir * ND + q) * M + h) * C + i] = sder(c.sh.out[i], q);
    }
    for (int k = t; k < c.ps.Kk; k += nt) {
        const int kind = c.ps.kind[k];
        S e(0.0);
        if (kind == 1) e = S(1.0);
        else if (kind >= 2) e = c.sh.e2[kind - 2];
        else for (int b = c.ps.boff[k]; b < c.ps.boff[k + 1]; ++b) e += c.ps.bw[b] * c.sh.out[c.ps.bidx[b]];
        const double v = sval(e);
        if (c.po.flags) {
            if (!(v > 0.0) || v > 1.0) c.po.flags[0] = 1;
            else if (c.ps.maxspan && (double)c.ps.maxspan[k] * ::log(v) < -450.0) c.po.flags[2] = 1;
        }
        if (c.dir == 0) {
            c.po.Eg_v[(size_t)k * M + h] = v;
            const int kl = c.ps.local ? c.ps.local[k] : k;
            if (kl >= 0) {
                if (c.po.El_v) c.po.El_v[(size_t)kl * c.po.Mp + h] = v;
                if (c.po.Es_v) c.po.Es_v[(size_t)(c.ps.slot ? c.ps.slot[k] : kl) * c.po.MS + h] = v;
            }
        }
        for (int q = 0; q < ND; ++q)
            if (c.dir * ND + q < pm.nder) c.po.Eg_d[((size_t)(c.dir * ND + q) * c.ps.Kk + k) * M + h] = sder(e, q);
    }
}

// ---- the same phases run serially on the host (CPU tests; never part of the product path) -----------------------------------
template <typename S> SMCPP_HD int n_groups(int nder) { return NDir<S>::value ? (nder + NDir<S>::value - 1) / NDir<S>::value : 1; }
template <typename S>
inline void emulate_tables(const PrepModel &pm, const Tables<S> &tb) {
    std::vector<S> G(pm.K), F(pm.K);
    for (int dir = 0; dir < n_groups<S>(pm.nder); ++dir)
        for (int r = 0; r < 2 * pm.n + 1; ++r) {
            for (int m = 0; m < pm.K; ++m) tables_term<S>(pm, dir, r, m, G[m], F[m]);
            tables_scan<S>(pm, dir, tb, r, G.data(), F.data());
        }
}
template <typename S>
inline void emulate_csfs(const PrepModel &pm, const PrepStatic &ps, const PrepOut &po, const Tables<S> &tb) {
    const int n = pm.n;
    std::vector<S> scratch(CsfsScratch<S>::count(n));
    const int nt = 448;
    for (int h = 0; h < pm.M; ++h)
        for (int dir = 0; dir < n_groups<S>(pm.nder); ++dir) {
            CsfsCtx<S> c;
            c.pm = pm; c.ps = ps; c.po = po; c.tb = tb; c.h = h; c.dir = dir;
            c.sh.carve(scratch.data(), n);
            for (int t = 0; t < nt; ++t) csfs_clear(c, t, nt);
            for (int m = pm.hsi[h]; m < pm.hsi[h + 1]; ++m) {
                for (int t = 0; t < nt; ++t) csfs_piece_tables(c, m, t, nt);
                for (int p = 0; p < (n + 1) * n; ++p) csfs_pair(c, m, p);
            }
            for (int t = 0; t < nt; ++t) csfs_contract(c, t);
            for (int t = 0; t < nt; ++t) csfs_backtransform(c, t);
            csfs_theta_a(c);
            for (int t = 0; t < nt; ++t) csfs_theta_b(c, t, nt);
            csfs_theta_c(c);
            for (int t = 0; t < nt; ++t) csfs_theta_d(c, t, nt);
            for (int t = 0; t < nt; ++t) csfs_emit(c, t, nt);
        }
}

// ---- Q (HMM::Q, src/hmm.cpp:155-193) and its gradient on the device ---------------------------------------------------------------
// Q's four terms are sums of w log x over (gamma0, pi), (gamma sums, emission table: keys without / with undistinguished
// lineages) and (xisum, T); their forward-mode derivatives are sums of (w / x) dx.  Block 0 forms the values, block 1 + d the
// derivatives along direction d.  The transition matrix is never materialised: T(i, c) and dT(i, c) come from the O(M) generators
// (prep.hpp: TransitionGenJac; below the diagonal ed[c], above it pf[i] W[c], the diagonal closes the row, 1e-20 floor, uniform mix).
struct QArgs {
    int M = 0, Kq = 0, nder = 0;
    const double *g0 = nullptr, *xi = nullptr, *gs = nullptr;       // statistics summed over contigs: [M], [M][M], [Kq][M]
    const int *key_nb = nullptr;                                    // [Kq] the key has undistinguished lineages (term 2) or not (term 1)
    const double *pi_v = nullptr, *pi_d = nullptr;                  // [M], planes [nder][M]
    const double *E_v = nullptr, *E_d = nullptr;                    // [Kq][M], planes [nder][Kq][M]
    const double *ed_v = nullptr, *ed_d = nullptr;                  // [M] (M - 1 used), planes [nder][M]
    const double *pf_v = nullptr, *pf_d = nullptr, *W_v = nullptr, *W_d = nullptr;   // [M], planes [nder][M]
    double mix_p2 = 0.0;                                            // 1e-5 / (M + 1)
    int nslice = 1;                                                 // the items of a block row are split over `nslice` workgroups
    double *out = nullptr;                                          // [(1 + nder)][nslice][4]: the caller adds the slices in order
};
// the generators of the transition matrix (values and ONE direction's derivative plane) as the block reads them: global memory
// in the host emulation, a staged LDS copy in the kernel
struct QGen { const double *ed, *W, *pf, *ded, *dW, *dpf; };
SMCPP_HD QGen q_gen_global(const QArgs &a, int dir) {
    QGen g;
    g.ed = a.ed_v; g.W = a.W_v; g.pf = a.pf_v;
    g.ded = dir >= 0 ? a.ed_d + (size_t)dir * a.M : nullptr;
    g.dW = dir >= 0 ? a.W_d + (size_t)dir * a.M : nullptr;
    g.dpf = dir >= 0 ? a.pf_d + (size_t)dir * a.M : nullptr;
    return g;
}
// unfloored row sum of the off-diagonal entries of row i, accumulated in column order as transition_expand does, and (dir >= 0)
// the derivative of that sum
SMCPP_HD void q_rowsum(const QGen &g, int M, int dir, int i, double &sm, double &dsm) {
    sm = 0.0; dsm = 0.0;
    const double pf = g.pf[i];
    const double dpf = dir >= 0 ? g.dpf[i] : 0.0;
    for (int c = 0; c < M; ++c) {
        if (c == i) continue;
        sm += c < i ? g.ed[c] : pf * g.W[c];
        if (dir >= 0) dsm += c < i ? g.ded[c] : dpf * g.W[c] + pf * g.dW[c];
    }
}
// contribution of item idx (0 .. M + Kq M + M M) of block b: term index and value; diag / ddiag = the rows' diagonals (1 - sm, -dsm)
SMCPP_HD double q_item(const QArgs &a, const QGen &g, int b, long idx, const double *diag, const double *ddiag, int &term) {
    const int M = a.M, dir = b - 1;
    double w, x, dx = 0.0;
    if (idx < M) {
        term = 0;
        w = a.g0[idx]; x = a.pi_v[idx];
        if (dir >= 0) dx = a.pi_d[(size_t)dir * M + idx];
    } else if (idx < (long)M + (long)a.Kq * M) {
        const long e = idx - M;
        term = a.key_nb[e / M] ? 2 : 1;
        w = a.gs[e]; x = a.E_v[e];
        if (dir >= 0) dx = a.E_d[(size_t)dir * a.Kq * M + e];
    } else {
        const long e = idx - M - (long)a.Kq * M;
        const int i = (int)(e / M), c = (int)(e % M);
        term = 3;
        w = a.xi[e];
        double t, dt = 0.0;
        if (c == i) { t = diag[i]; dt = dir >= 0 ? ddiag[i] : 0.0; }
        else if (c < i) { t = g.ed[c]; if (dir >= 0) dt = g.ded[c]; }
        else {
            t = g.pf[i] * g.W[c];
            if (dir >= 0) dt = g.dpf[i] * g.W[c] + g.pf[i] * g.dW[c];
        }
        if (t < 1e-20) { t = 1e-20; dt = 0.0; }
        x = t * (1 - 1e-5) + a.mix_p2;
        dx = dt * (1 - 1e-5);
    }
    if (w == 0.0) return 0.0;                  // (a key no contig holds, an empty row: no contribution, as hmm.cpp:166-181 skips them)
    return dir < 0 ? w * ::log(x) : (w / x) * dx;
}
inline void emulate_q(const QArgs &a) {
    const int M = a.M;
    std::vector<double> diag(M), ddiag(M);
    const long items = (long)M + (long)a.Kq * M + (long)M * M;
    for (int b = 0; b <= a.nder; ++b) {
        const QGen g = q_gen_global(a, b - 1);
        for (int i = 0; i < M; ++i) { double sm, dsm; q_rowsum(g, M, b - 1, i, sm, dsm); diag[i] = 1.0 - sm; ddiag[i] = -dsm; }
        AccD acc[4];
        for (long idx = 0; idx < items; ++idx) { int term; const double v = q_item(a, g, b, idx, diag.data(), ddiag.data(), term); acc_add(acc[term], v); }
        for (int t = 0; t < 4; ++t) acc_get(acc[t], a.out[(size_t)b * 4 + t]);
    }
}

#ifdef __HIPCC__
// ---- kernels ------------------------------------------------------------------------------------------------------------------
// statistics of all contigs summed in contig order into the compact layout Q reads: [gamma0 M | xisum M M | gamma sums K M]
__global__ void k_q_stats(int n_contigs, int M, int Mp, int K, const double *gamma0, const double *xisum, const double *gsum, double *out) {
    const long n = (long)M + (long)M * M + (long)K * M;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        if (idx < M) for (int c = 0; c < n_contigs; ++c) s += gamma0[(size_t)c * Mp + idx];
        else if (idx < (long)M + (long)M * M) {
            const long e = idx - M;
            const int i = (int)(e / M), j = (int)(e % M);
            for (int c = 0; c < n_contigs; ++c) s += xisum[((size_t)c * Mp + i) * Mp + j];
        } else {
            const long e = idx - M - (long)M * M;
            const int k = (int)(e / M), i = (int)(e % M);
            for (int c = 0; c < n_contigs; ++c) s += gsum[((size_t)c * K + k) * Mp + i];
        }
        out[idx] = s;
    }
}

__global__ __launch_bounds__(1024) void k_q_reduce(QArgs a) {
    extern __shared__ double q_lds[];
    const int M = a.M, b = blockIdx.x, sl = blockIdx.y, t = threadIdx.x, nt = blockDim.x;
    double *diag = q_lds, *ddiag = q_lds + M, *gen = q_lds + 2 * M, *red = q_lds + 8 * M;        // gen [6][M], red [4][nt / 64][2]
    {
        // the generators (and this block's derivative plane) staged once: every row sum and every item reads them from LDS
        const QGen gg = q_gen_global(a, b - 1);
        for (int i = t; i < M; i += nt) {
            gen[i] = gg.ed[i]; gen[M + i] = gg.W[i]; gen[2 * M + i] = gg.pf[i];
            gen[3 * M + i] = b > 0 ? gg.ded[i] : 0.0; gen[4 * M + i] = b > 0 ? gg.dW[i] : 0.0; gen[5 * M + i] = b > 0 ? gg.dpf[i] : 0.0;
        }
    }
    __syncthreads();
    QGen g;
    g.ed = gen; g.W = gen + M; g.pf = gen + 2 * M; g.ded = gen + 3 * M; g.dW = gen + 4 * M; g.dpf = gen + 5 * M;
    for (int i = t; i < M; i += nt) { double sm, dsm; q_rowsum(g, M, b - 1, i, sm, dsm); diag[i] = 1.0 - sm; ddiag[i] = -dsm; }
    __syncthreads();
    AccD acc[4];
    const long items = (long)M + (long)a.Kq * M + (long)M * M;
    for (long idx = (long)sl * nt + t; idx < items; idx += (long)nt * a.nslice) {
        int term;
        const double v = q_item(a, g, b, idx, diag, ddiag, term);
        // (one accumulator per term; the term of an item is uniform over long runs of idx, so the selects are cheap)
        for (int q = 0; q < 4; ++q) if (q == term) acc_add(acc[q], v);
    }
    // wavefront tree on (hi, lo) pairs, then one thread adds the wavefronts' partials in order: deterministic
    const int lane = t & 63, w = t >> 6, nw = nt >> 6;
    for (int q = 0; q < 4; ++q) {
        double hi = acc[q].hi, lo = acc[q].lo;
        for (int off = 32; off >= 1; off >>= 1) {
            const double oh = __shfl_down(hi, off), ol = __shfl_down(lo, off);
            const double s = hi + oh, z = s - hi;
            lo += ol + ((hi - (s - z)) + (oh - z));
            hi = s;
        }
        if (lane == 0) { red[(q * nw + w) * 2] = hi; red[(q * nw + w) * 2 + 1] = lo; }
    }
    __syncthreads();
    if (t < 4) {
        AccD r;
        for (int ww = 0; ww < nw; ++ww) { acc_add(r, red[(t * nw + ww) * 2]); r.lo += red[(t * nw + ww) * 2 + 1]; }
        a.out[((size_t)b * a.nslice + sl) * 4 + t] = r.hi + r.lo;
    }
}

// grid (direction groups, 2 n + 1 rates): the workgroup forms the K terms of ITS rate (one per thread) in LDS, then one thread runs
// the rate's recurrence over them
template <typename S>
__global__ __launch_bounds__(256) void k_prep_tables(PrepModel pm, Tables<S> tb) {
    extern __shared__ double tables_lds[];
    S *G = reinterpret_cast<S *>(tables_lds), *F = G + pm.K;
    const int dir = blockIdx.x, r = blockIdx.y;
    for (int m = threadIdx.x; m < pm.K; m += blockDim.x) tables_term<S>(pm, dir, r, m, G[m], F[m]);
    __syncthreads();
    if (threadIdx.x == 0) tables_scan<S>(pm, dir, tb, r, G, F);
}

// (at most 512 threads: with the default bound of 1024 the compiler caps the kernel at 128 registers and the four-direction
// instantiation spills 120 of them)
template <typename S>
__global__ __launch_bounds__(512) void k_prep_csfs(PrepModel pm, PrepStatic ps, PrepOut po, Tables<S> tb) {
    extern __shared__ double prep_lds[];
    CsfsCtx<S> c;
    c.pm = pm; c.ps = ps; c.po = po; c.tb = tb; c.h = blockIdx.x; c.dir = blockIdx.y;
    c.sh.carve(reinterpret_cast<S *>(prep_lds), pm.n);
    const int t = threadIdx.x, nt = blockDim.x, n = pm.n;
    csfs_clear(c, t, nt);
    __syncthreads();
    for (int m = pm.hsi[c.h]; m < pm.hsi[c.h + 1]; ++m) {
        csfs_piece_tables(c, m, t, nt);
        __syncthreads();
        for (int p = t; p < (n + 1) * n; p += nt) csfs_pair(c, m, p);
        __syncthreads();
    }
    csfs_contract(c, t);
    __syncthreads();
    csfs_backtransform(c, t);
    __syncthreads();
    if (t == 0) csfs_theta_a(c);
    __syncthreads();
    csfs_theta_b(c, t, nt);
    __syncthreads();
    if (t == 0) csfs_theta_c(c);
    __syncthreads();
    csfs_theta_d(c, t, nt);
    __syncthreads();
    csfs_emit(c, t, nt);
}

// The conditioned SFS alone - phases 0 .. 4 of k_prep_csfs, values only, no incorporate_theta, no emission assembly: raw [M][3 (n+1)].
// The batched intervals of the two-population preparation (jcsfs.hpp: everything below the split under the truncated model, everything
// above it under the shifted one) run through this while the host forms the pieces that do not depend on the hidden state.
__global__ __launch_bounds__(512) void k_prep_csfs_raw(PrepModel pm, PrepStatic ps, Tables<double> tb, double *raw) {
    extern __shared__ double prep_lds[];
    CsfsCtx<double> c;
    c.pm = pm; c.ps = ps; c.tb = tb; c.h = blockIdx.x; c.dir = 0;
    c.sh.carve(prep_lds, pm.n);
    const int t = threadIdx.x, nt = blockDim.x, n = pm.n, C = 3 * (n + 1);
    csfs_clear(c, t, nt);
    __syncthreads();
    for (int m = pm.hsi[c.h]; m < pm.hsi[c.h + 1]; ++m) {
        csfs_piece_tables(c, m, t, nt);
        __syncthreads();
        for (int p = t; p < (n + 1) * n; p += nt) csfs_pair(c, m, p);
        __syncthreads();
    }
    csfs_contract(c, t);
    __syncthreads();
    csfs_backtransform(c, t);
    __syncthreads();
    for (int i = t; i < C; i += nt) raw[(size_t)c.h * C + i] = c.sh.out[i];
}
#endif

}  // namespace smcpp_dev

#pragma clang fp contract(fast)
