// Host-side cold parameter preparation: model pieces (a, s) + hidden states ->
// pi [M], T [M x M], emission table [K x M].  Values only (no derivative seeds; those ride on the M-step path).
//
// Own restatement of the mathematics of
//   A9  PiecewiseConstantRateFunction            src/piecewise_constant_rate_function.cpp:31-84,157-211,214-334,372-420
//   A8  HJTransition / compute_transition         src/transition.cpp:113-262 (3x3 chain in long double instead of MPFR)
//   A10 OnePopConditionedSFS, MatrixCache, Moran  src/conditioned_sfs.cpp:13-148, src/matrix_cache.cpp:115-282,
//                                                 src/moran_eigensystem.cpp:31-96 (exact rationals on GMP's C API)
//   A7  recompute_initial_distribution            src/inference_manager.cpp:56-69
//   A6  construct_bins / recompute_emission_probs src/inference_manager.cpp:329-482, bin_key.h:36-64,
//                                                 marginalize_key.h:21-51, tensorslice.h:31-42
// Pinned by tests/test_prep.py against the compiled reference (oracle/_ref: ref_prep) and the golden parameter files.
#pragma once
#include <gmp.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <vector>

namespace smcpp_host {

struct ModelParams {
    std::vector<double> a, s;   // piece sizes and lengths (ParameterVector, _smcpp.pyx:66-83)
};

// ---------------------------------------------------------------------------------------------------------------
// exact rationals
// ---------------------------------------------------------------------------------------------------------------
class Q {
public:
    Q() { mpq_init(v); }
    Q(long num) { mpq_init(v); mpq_set_si(v, num, 1); }
    Q(long num, long den) {
        mpq_init(v);
        if (den < 0) { num = -num; den = -den; }
        mpq_set_si(v, num, (unsigned long)den);
        mpq_canonicalize(v);
    }
    Q(const Q &o) { mpq_init(v); mpq_set(v, o.v); }
    Q(Q &&o) noexcept { mpq_init(v); mpq_swap(v, o.v); }
    Q &operator=(const Q &o) { if (this != &o) mpq_set(v, o.v); return *this; }
    Q &operator=(Q &&o) noexcept { mpq_swap(v, o.v); return *this; }
    ~Q() { mpq_clear(v); }
    static Q binom(unsigned long n, unsigned long k) {
        Q r;
        mpz_bin_uiui(mpq_numref(r.v), n, k);
        return r;
    }
    Q operator+(const Q &o) const { Q r; mpq_add(r.v, v, o.v); return r; }
    Q operator-(const Q &o) const { Q r; mpq_sub(r.v, v, o.v); return r; }
    Q operator*(const Q &o) const { Q r; mpq_mul(r.v, v, o.v); return r; }
    Q operator/(const Q &o) const { Q r; mpq_div(r.v, v, o.v); return r; }
    Q operator-() const { Q r; mpq_neg(r.v, v); return r; }
    Q &operator+=(const Q &o) { mpq_add(v, v, o.v); return *this; }
    Q &operator-=(const Q &o) { mpq_sub(v, v, o.v); return *this; }
    bool is_zero() const { return mpq_sgn(v) == 0; }
    double to_double() const { return mpq_get_d(v); }
private:
    mpq_t v;
};

struct QMat {
    int r = 0, c = 0;
    std::vector<Q> d;
    QMat() {}
    QMat(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_) {}
    Q &operator()(int i, int j) { return d[(size_t)i * c + j]; }
    const Q &operator()(int i, int j) const { return d[(size_t)i * c + j]; }
};

struct DMat {
    int r = 0, c = 0;
    std::vector<double> d;
    DMat() {}
    DMat(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return d[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
};

inline DMat to_double(const QMat &m) {
    DMat o(m.r, m.c);
    for (size_t i = 0; i < m.d.size(); ++i) o.d[i] = m.d[i].to_double();
    return o;
}

// C = A * diag(w) * B  (w may be empty = identity), OpenMP over rows
inline QMat qmul(const QMat &A, const std::vector<Q> &w, const QMat &B) {
    QMat C(A.r, B.c);
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < A.r; ++i)
        for (int k = 0; k < A.c; ++k) {
            if (A(i, k).is_zero()) continue;
            const Q aik = w.empty() ? A(i, k) : A(i, k) * w[k];
            for (int j = 0; j < B.c; ++j)
                if (!B(k, j).is_zero()) C(i, j) += aik * B(k, j);
        }
    return C;
}

// ---------------------------------------------------------------------------------------------------------------
// the n-only constant tables of the conditioned SFS (matrix_cache.cpp:212-282)
// ---------------------------------------------------------------------------------------------------------------
struct CsfsTables {
    int n = 0;
    DMat X0, X2, M0, M1, Uinv_mp0, Uinv_mp2;
};

namespace detail {

// modified_moran_rate_matrix(N, a = 0, na = 2) is tridiagonal (moran_eigensystem.cpp:31-52)
struct Tri {
    int n;                       // size
    std::vector<Q> lo, di, up;   // lo[i] = M(i, i-1), di[i] = M(i, i), up[i] = M(i, i+1)
};

inline Tri moran_matrix(int N) {
    Tri t;
    t.n = N + 1;
    t.lo.assign(N + 1, Q(0)); t.di.assign(N + 1, Q(0)); t.up.assign(N + 1, Q(0));
    const int a = 0, na = 2;
    for (int i = 0; i <= N; ++i) {
        Q sm(0);
        if (i > 0) {
            Q b = Q((long)(na - a) * i) + Q((long)i * (N - i), 2);
            t.lo[i] = b;
            sm += b;
        }
        if (i < N) {
            Q b = Q((long)a * (N - i)) + Q((long)i * (N - i), 2);
            t.up[i] = b;
            sm += b;
        }
        t.di[i] = -sm;
    }
    return t;
}

// back-substitution `solve` (moran_eigensystem.cpp:54-64) on the tridiagonal A = Tri - rate*I restricted to
// rows/cols [off, off+m): x[m-1] = 1; x[i] = (A.row(i+1) . x) / -A(i+1, i)
inline std::vector<Q> solve_tri(const std::vector<Q> &lo, const std::vector<Q> &di, const std::vector<Q> &up,
                                const Q &rate, int off, int m) {
    std::vector<Q> x(m, Q(0));
    x[m - 1] = Q(1);
    for (int i = m - 2; i >= 0; --i) {
        const int r = off + i + 1;                 // global row i+1
        Q acc = (di[r] - rate) * x[i + 1];
        if (i + 2 < m) acc += up[r] * x[i + 2];
        // A(i+1, i) = lo[r]
        x[i] = acc / (-lo[r]);
    }
    return x;
}

struct Moran {
    QMat U, Uinv;
    std::vector<Q> D;
};

inline Moran moran_eigensystem(int n) {
    Moran me;
    const int N1 = n + 1;
    me.U = QMat(N1, N1);
    me.Uinv = QMat(N1, N1);
    me.D.assign(N1, Q(0));
    Tri M = moran_matrix(n);
    // transpose: Mt(i, i-1) = M(i-1, i) = up[i-1]; Mt(i, i+1) = M(i+1, i) = lo[i+1]
    std::vector<Q> tlo(N1, Q(0)), tup(N1, Q(0));
    for (int i = 0; i < N1; ++i) {
        if (i > 0) tlo[i] = M.up[i - 1];
        if (i < N1 - 1) tup[i] = M.lo[i + 1];
    }
    me.Uinv(0, 0) = Q(2);                           // `mpq_1` is defined as 2/1 (moran_eigensystem.cpp:5)
    for (int k = 2; k < n + 3; ++k) {
        const Q rate(-((long)k * (k - 1) / 2 - 1));
        me.D[k - 2] = rate;
        std::vector<Q> col = solve_tri(M.lo, M.di, M.up, rate, 0, N1);
        for (int i = 0; i < N1; ++i) me.U(i, k - 2) = col[i];
        if (k > 2) {
            if (n >= 1) {
                std::vector<Q> row = solve_tri(tlo, M.di, tup, rate, 1, n);   // bottom-right n x n block of Mt - rate I
                for (int j = 0; j < n; ++j) me.Uinv(k - 2, 1 + j) = row[j];
            }
            // Uinv(k-2, 0) = -Uinv(k-2, 1) * A(0,1) / A(0,0) with A = Mt - rate I
            const Q a01 = tup[0], a00 = M.di[0] - rate;
            if (N1 > 1) me.Uinv(k - 2, 0) = -(me.Uinv(k - 2, 1) * a01) / a00;
        }
    }
    // U <- U * diag(1 / diag(Uinv U))
    for (int j = 0; j < N1; ++j) {
        Q dsum(0);
        for (int k = 0; k < N1; ++k) dsum += me.Uinv(j, k) * me.U(k, j);
        for (int i = 0; i < N1; ++i) me.U(i, j) = me.U(i, j) / dsum;
    }
    return me;
}

inline Q wnbj(int n, int b, int j, std::map<std::pair<int, int>, Q> &memo) {
    if (j == 2) return Q(6, n + 1);
    if (j == 3) {
        if (n == 2 * b) return Q(0);
        return Q(30L * (n - 2 * b), (long)(n + 1) * (n + 2));
    }
    auto key = std::make_pair(b, j);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    const long jj = j - 2;
    const Q c1(-(1 + jj) * (3 + 2 * jj) * (n - jj), jj * (2 * jj - 1) * (n + jj + 1));
    const Q c2((3 + 2 * jj) * (n - 2L * b), jj * (n + jj + 1));
    Q ret = wnbj(n, b, (int)jj, memo) * c1 + wnbj(n, b, (int)jj + 1, memo) * c2;
    memo.emplace(key, ret);
    return ret;
}

inline QMat below_coeffs(int n) {
    QMat mlast;
    for (int nn = 2; nn < n + 3; ++nn) {
        QMat mnew(n + 1, nn - 1);
        mnew(nn - 2, nn - 2) = Q(1);
        for (int k = nn - 1; k > 1; --k) {
            const long denom = (long)(nn + 1) * (nn - 2) - (long)(k + 1) * (k - 2);
            const Q c1((long)(nn + 1) * (nn - 2), denom);
            for (int i = 0; i < n + 1; ++i) mnew(i, k - 2) = mlast(i, k - 2) * c1;
        }
        for (int k = nn - 1; k > 1; --k) {
            const long denom = (long)(nn + 1) * (nn - 2) - (long)(k + 1) * (k - 2);
            const Q c2((long)(k + 2) * (k - 1), denom);
            for (int i = 0; i < n + 1; ++i) mnew(i, k - 2) -= mnew(i, k - 1) * c2;
        }
        mlast = mnew;
    }
    return mlast;
}

}  // namespace detail

inline std::shared_ptr<const CsfsTables> csfs_tables(int n) {
    static std::mutex mu;
    static std::map<int, std::shared_ptr<const CsfsTables>> memo;
    std::lock_guard<std::mutex> lk(mu);
    auto it = memo.find(n);
    if (it != memo.end()) return it->second;
    auto t = std::make_shared<CsfsTables>();
    t->n = n;
    const detail::Moran me = detail::moran_eigensystem(n);
    const int N1 = n + 1;
    // Uinv_mp0 = Uinv.rightCols(n); Uinv_mp2 = Uinv.reverse().leftCols(n)   (conditioned_sfs.cpp:8-9)
    t->Uinv_mp0 = DMat(N1, n);
    t->Uinv_mp2 = DMat(N1, n);
    for (int i = 0; i < N1; ++i)
        for (int j = 0; j < n; ++j) {
            t->Uinv_mp0(i, j) = me.Uinv(i, 1 + j).to_double();
            t->Uinv_mp2(i, j) = me.Uinv(n - i, n - j).to_double();
        }
    std::vector<Q> Dab(n), oneDab(n), Dbe(N1), oneDbe(N1), lsp(N1);
    for (int i = 0; i < n; ++i) { Dab[i] = Q(i + 1, n + 1); oneDab[i] = Q(1) - Dab[i]; }
    for (int i = 0; i < N1; ++i) { Dbe[i] = Q(2, i + 2); oneDbe[i] = Q(1) - Dbe[i]; lsp[i] = Q(i + 2); }
    QMat WnbjT(n, n);                               // Wnbj^T: (j-2, b-1)
    {
        std::map<std::pair<int, int>, Q> memoW;
        for (int b = 1; b < n + 1; ++b)
            for (int j = 2; j < n + 2; ++j) WnbjT(j - 2, b - 1) = detail::wnbj(n + 1, b, j, memoW);
    }
    QMat P_dist(N1, N1), P_undist(N1, n);
    for (int k = 0; k < N1; ++k)
        for (int b = 1; b < n - k + 2; ++b)
            P_dist(k, b - 1) = Q((long)b) * Q::binom(n + 2 - b, k + 1) / Q::binom(n + 3, k + 3);
    for (int k = 1; k < N1; ++k)
        for (int b = 1; b < n - k + 2; ++b)
            P_undist(k, b - 1) = Q::binom(n + 3 - b, k + 2) / Q::binom(n + 3, k + 3);
    QMat Ubot(n, N1), Urevtop(n, N1);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < N1; ++j) {
            Ubot(i, j) = me.U(1 + i, j);            // U.bottomRows(n)
            Urevtop(i, j) = me.U(n - i, n - j);     // U.reverse().topRows(n)
        }
    const QMat bc = detail::below_coeffs(n);
    std::vector<Q> w0(N1), w1(N1);
    for (int i = 0; i < N1; ++i) { w0[i] = lsp[i] * oneDbe[i]; w1[i] = lsp[i] * Dbe[i]; }
    t->X0 = to_double(qmul(WnbjT, oneDab, Ubot));
    t->X2 = to_double(qmul(WnbjT, Dab, Urevtop));
    t->M0 = to_double(qmul(bc, w0, P_undist));
    t->M1 = to_double(qmul(bc, w1, P_dist));
    memo.emplace(n, t);
    return t;
}

// ---------------------------------------------------------------------------------------------------------------
// A9: piecewise-constant rate function
// ---------------------------------------------------------------------------------------------------------------
class RateFunction {
public:
    RateFunction(const ModelParams &p, const std::vector<double> &hs) : hidden_states(hs) {
        if (p.a.size() != p.s.size() || p.a.empty()) throw std::runtime_error("all params must have same size");
        K = (int)p.a.size();
        ada.resize(K);
        ts.assign(K + 1, 0.0);
        for (int k = 0; k < K; ++k) {
            ada[k] = 1.0 / p.a[k];
            ts[k + 1] = ts[k] + p.s[k];
        }
        ts[K] = INFINITY;                            // the final piece always extends to infinity
        for (double h : hidden_states) {
            if (std::isinf(h)) { hs_indices.push_back((int)ts.size() - 1); continue; }
            auto ti = std::upper_bound(ts.begin(), ts.end(), h) - 1;
            const int ip = (int)(ti - ts.begin());
            if (std::fabs(*ti - h) < 1e-8) hs_indices.push_back(ip);
            else if (ti + 1 < ts.end() && std::fabs(*(ti + 1) - h) < 1e-8) hs_indices.push_back(ip + 1);
            else {
                ts.insert(ti + 1, h);
                ada.insert(ada.begin() + ip + 1, ada[ip]);
                hs_indices.push_back(ip + 1);
            }
        }
        K = (int)ada.size();
        Rrng.assign(K + 1, 0.0);
        for (int k = 0; k < K; ++k) Rrng[k + 1] = Rrng[k] + ada[k] * (ts[k + 1] - ts[k]);
    }

    double R(double t) const {
        auto ti = std::upper_bound(ts.begin(), ts.end(), t) - 1;
        const int ip = (int)(ti - ts.begin());
        return Rrng[ip] + ada[ip] * (t - *ti);
    }

    // int_a^b exp(-(R(t) + log_denom)) dt
    double R_integral(double a, double b, double log_denom) const {
        const int ip_a = (int)(std::upper_bound(ts.begin(), ts.end(), a) - 1 - ts.begin());
        int ip_b = (int)(std::upper_bound(ts.begin(), ts.end(), b) - 1 - ts.begin());
        if (std::isinf(b)) ip_b = (int)ts.size() - 2;
        double ret = 0.0;
        for (int i = ip_a; i < ip_b + 1; ++i) {
            const double left = std::max(a, ts[i]), right = std::min(b, ts[i + 1]);
            const double diff = right - left;
            double r = std::exp(-(R(left) + log_denom));
            if (ada[i] > 0.0) {
                if (!std::isinf(diff)) r *= -std::expm1(-diff * ada[i]);
                r /= ada[i];
            } else r *= diff;
            ret += r;
        }
        return ret;
    }

    std::vector<double> average_coal_times() const {
        std::vector<double> ret;
        for (size_t i = 1; i < hidden_states.size(); ++i) {
            const double R0 = Rrng[hs_indices[i - 1]], R1 = Rrng[hs_indices[i]];
            if (R0 == R1) { ret.push_back(std::numeric_limits<double>::quiet_NaN()); continue; }
            double log_denom = -R0;
            const bool inf = std::isinf(ts[hs_indices[i]]);
            if (!inf) log_denom += std::log(-std::expm1(-(R1 - R0)));
            double x = hidden_states[i - 1] * std::exp(-(R0 + log_denom)) +
                       R_integral(ts[hs_indices[i - 1]], ts[hs_indices[i]], log_denom);
            if (!inf) x -= hidden_states[i] * std::exp(-(R1 + log_denom));
            ret.push_back(x);
            if (x > hidden_states[i] || x < hidden_states[i - 1])
                throw std::runtime_error("erroneous average coalescence time");
        }
        return ret;
    }

    // ---- integrals used by the conditioned SFS (piecewise_constant_rate_function.cpp:87-138,198-334) ----
    static double below_helper(long rate, double tsm, double tsm1, double ad, double Rr, double log_denom) {
        if (ad == 0) return 0.0;
        const long l1r = 1 + rate;
        const double l1rinv = 1.0 / (double)l1r;
        const double adadiff = ad * (tsm1 - tsm);
        if (rate == 0) {
            if (tsm1 == INFINITY) return std::exp(-Rr - log_denom) / ad;
            return std::exp(-Rr - log_denom) * (1.0 - std::exp(-adadiff) * (1.0 + adadiff)) / ad;
        }
        if (tsm1 == INFINITY) return std::exp(-l1r * Rr - log_denom) * (1.0 - l1rinv) / (rate * ad);
        return std::exp(-l1r * Rr - log_denom) * (std::expm1(-l1r * adadiff) * l1rinv - std::expm1(-adadiff)) / (rate * ad);
    }
    static double above_helper(long rate, long lam, double tsm, double tsm1, double ad, double Rr, double log_coef) {
        if (ad == 0) return 0.0;
        const double adadiff = ad * (tsm1 - tsm);
        const long l1 = lam + 1;
        if (rate == 0)
            return std::exp(-l1 * Rr + log_coef) * (std::expm1(-l1 * adadiff) + l1 * adadiff) / l1 / l1 / ad;
        if (l1 == rate) {
            if (tsm1 == INFINITY) return std::exp(-rate * Rr + log_coef) / rate / rate / ad;
            return std::exp(-rate * Rr + log_coef) * (1 - std::exp(-rate * adadiff) * (1 + rate * adadiff)) / rate / rate / ad;
        }
        if (tsm1 == INFINITY) return std::exp(-l1 * Rr + log_coef) / l1 / rate / ad;
        if (rate < l1)
            return -std::exp(-l1 * Rr + log_coef) *
                   (std::expm1(-l1 * adadiff) / l1 +
                    (std::exp(-rate * adadiff) * -std::expm1(-(double)(l1 - rate) * adadiff) / (double)(l1 - rate))) /
                   rate / ad;
        return -std::exp(-l1 * Rr + log_coef) *
               (std::expm1(-l1 * adadiff) / l1 +
                (std::exp(-l1 * adadiff) * std::expm1(-(double)(rate - l1) * adadiff) / (double)(l1 - rate))) /
               rate / ad;
    }
    static double single_integral(long rate, double tsm, double tsm1, double ad, double Rr, double log_coef) {
        if (rate == 0) return std::exp(log_coef) * (tsm1 - tsm);
        double ret = std::exp(-rate * Rr + log_coef);
        if (tsm1 < INFINITY) ret *= -std::expm1(-rate * ad * (tsm1 - tsm));
        ret /= ad * rate;
        return ret;
    }
    static long nC2(long n) { return n * (n - 1) / 2; }

    // row jj-2 of C[h] ((n+1) x n each), h over hidden states
    void tjj_double_integral_above(int n, long jj, std::vector<DMat> &C) const {
        const long lam = nC2(jj) - 1;
        for (size_t h = 0; h + 1 < hs_indices.size(); ++h) {
            for (int j = 0; j < n; ++j) C[h]((int)jj - 2, j) = 0.0;
            const double Rh = Rrng[hs_indices[h]], Rh1 = Rrng[hs_indices[h + 1]];
            double log_denom = -Rh;
            if (Rh1 != INFINITY) log_denom += std::log(-std::expm1(-(Rh1 - Rh)));
            for (int m = hs_indices[h]; m < hs_indices[h + 1]; ++m)
                for (int j = 2; j < n + 2; ++j) {
                    const long rate = nC2(j);
                    double &tgt = C[h]((int)jj - 2, j - 2);
                    tgt += above_helper(rate, lam, ts[m], ts[m + 1], ada[m], Rrng[m], -log_denom);
                    double log_coef = -log_denom, fac;
                    const long rp = lam + 1 - rate;
                    const double Rm1 = Rrng[m + 1], Rm = Rrng[m];
                    if (rp == 0) fac = Rm1 - Rm;
                    else if (rp < 0) {
                        if (-rp * (Rm1 - Rm) > 20) { log_coef += -rp * Rm1; fac = -1.0 / rp; }
                        else { log_coef += -rp * Rm; fac = -std::expm1(-rp * (Rm1 - Rm)) / rp; }
                    } else {
                        if (-rp * (Rm - Rm1) > 20) { log_coef += -rp * Rm; fac = 1.0 / rp; }
                        else { log_coef += -rp * Rm1; fac = std::expm1(-rp * (Rm - Rm1)) / rp; }
                    }
                    for (int k = m + 1; k < K; ++k)
                        tgt += single_integral(rate, ts[k], ts[k + 1], ada[k], Rrng[k], log_coef) * fac;
                }
        }
    }

    // row h of tgt (M x (n+1))
    void tjj_double_integral_below(int n, int h, DMat &tgt) const {
        const double Rh = Rrng[hs_indices[h]], Rh1 = Rrng[hs_indices[h + 1]];
        double log_denom = -Rh;
        if (Rh1 != INFINITY) log_denom += std::log(-std::expm1(-(Rh1 - Rh)));
        for (int m = hs_indices[h]; m < hs_indices[h + 1]; ++m) {
            const double Rm = Rrng[m], Rm1 = Rrng[m + 1];
            const double log_coef = -Rm;
            double fac = 1.0;
            if (m < K - 1) fac = -std::expm1(-(Rm1 - Rm));
            for (int j = 2; j < n + 3; ++j) {
                const long rate = nC2(j) - 1;
                double v = below_helper(rate, ts[m], ts[m + 1], ada[m], Rrng[m], log_denom);
                for (int k = 0; k < m; ++k)
                    v += fac * single_integral(rate, ts[k], ts[k + 1], ada[k], Rrng[k], log_coef - log_denom);
                tgt(h, j - 2) += v;
            }
        }
    }

    std::vector<double> hidden_states, ts, ada, Rrng;
    std::vector<int> hs_indices;
    int K = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// A8: transition matrix (transition.cpp:113-262)
// ---------------------------------------------------------------------------------------------------------------
namespace detail {
typedef long double ld;
struct M3 { ld m[3][3]; };
inline M3 m3_identity() { M3 r{}; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = i == j; return r; }
inline M3 m3_mul(const M3 &a, const M3 &b) {
    M3 r{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { ld s = 0; for (int k = 0; k < 3; ++k) s += a.m[i][k] * b.m[k][j]; r.m[i][j] = s; }
    return r;
}
// closed-form exponential of c_rho * A_rho + c_eta * A_eta (transition.cpp:113-130); the factors e * cosh and
// e * sinh are combined so nothing overflows (the reference relies on 256-bit MPFR here)
inline M3 matrix_exp(ld c_rho, ld c_eta) {
    const ld sq = sqrtl(4 * c_eta * c_eta + c_rho * c_rho);
    const ld y = c_eta + c_rho / 2, x = sq / 2;
    ld ec, es;   // e*cosh(x), e*sinh(x)/sq
    if (sq == 0) { ec = 1; es = 0.5L; }
    else {
        const ld ep = expl(x - y), em = expl(-x - y);
        ec = 0.5L * (ep + em);
        es = (x < 0.5L) ? expl(-y) * sinhl(x) / sq : 0.5L * (ep - em) / sq;
    }
    M3 Qm{};
    Qm.m[0][0] = ec + (2 * c_eta - c_rho) * es;
    Qm.m[0][1] = 2 * c_rho * es;
    Qm.m[0][2] = 1 - Qm.m[0][0] - Qm.m[0][1];
    Qm.m[1][0] = 2 * c_eta * es;
    Qm.m[1][1] = ec - (2 * c_eta - c_rho) * es;
    Qm.m[1][2] = 1 - Qm.m[1][0] - Qm.m[1][1];
    Qm.m[2][0] = 0; Qm.m[2][1] = 0; Qm.m[2][2] = 1;
    return Qm;
}
}  // namespace detail

inline std::vector<double> compute_transition(const RateFunction &eta, double rho) {
    using namespace detail;
    const std::vector<double> &ts = eta.ts, &ada = eta.ada;
    const std::vector<int> &hsi = eta.hs_indices;
    const int Mh = (int)eta.hidden_states.size();   // the reference's `this->M` (breakpoints)
    const int M = Mh - 1;
    const std::vector<double> avg = eta.average_coal_times();
    const int nts = (int)ts.size();
    std::vector<M3> expms(nts, m3_identity()), prods(nts, m3_identity());
    for (int i = hsi[0] + 1; i < nts; ++i) {
        if (!std::isinf(ts[i])) {
            const double delta = ts[i] - ts[i - 1];
            expms[i] = matrix_exp((ld)delta * (ld)rho, (ld)ada[i - 1] * (ld)delta);
        }   // infinite end: the reference push_back()s instead of assigning, so expm_U[i] stays the identity
        prods[i] = m3_mul(prods[i - 1], expms[i]);
    }
    std::vector<int> avc_ip(M);
    for (int j = 0; j < M; ++j)
        avc_ip[j] = (int)(std::upper_bound(ts.begin(), ts.end(), avg[j]) - ts.begin()) - 1;
    std::vector<double> expm_diff(std::max(0, M - 1));
    for (int k = 1; k < M; ++k)
        expm_diff[k - 1] = (double)prods[hsi[k]].m[0][2] - (double)prods[hsi[k - 1]].m[0][2];
    std::vector<double> Phi((size_t)M * M, 0.0);
    for (int j = 1; j < Mh; ++j) {
        double *row = &Phi[(size_t)(j - 1) * M];
        for (int k = 0; k < j - 1; ++k) row[k] = expm_diff[k];
        const double rct = avg[j - 1];
        const int rct_ip = avc_ip[j - 1];
        M3 A = m3_identity();
        for (int ell = hsi[j - 1]; ell < rct_ip; ++ell) {
            M3 e = expms[ell];
            for (auto &r : e.m) for (auto &v : r) v = (ld)(double)v;   // `expms` are stored as double
            A = m3_mul(A, e);
        }
        const double delta = rct - ts[rct_ip];
        const double c_eta = ada[rct_ip] * delta;
        const double c_rho = delta * rho;
        {
            M3 e = matrix_exp((ld)c_rho, (ld)c_eta);
            A = m3_mul(A, e);
        }
        M3 Pj = prods[hsi[j - 1]];
        for (auto &r : Pj.m) for (auto &v : r) v = (ld)(double)v;
        const M3 B = m3_mul(Pj, A);
        double Rj = c_eta;
        Rj += ada[rct_ip] * (ts[rct_ip + 1] - rct);
        for (int jj = rct_ip + 2; jj < hsi[j]; ++jj) Rj += ada[jj] * (ts[jj + 1] - ts[jj]);
        const double p_float = (double)B.m[0][1] * std::exp(-Rj);
        double Rjk1 = 0.0;
        for (int k = j + 1; k < Mh; ++k) {
            double inc = 0.0;
            for (int jj = hsi[k - 1]; jj < hsi[k]; ++jj) inc += ada[jj] * (ts[jj + 1] - ts[jj]);
            double p_coal = std::exp(-Rjk1);
            Rjk1 += inc;
            if (!std::isinf(inc)) p_coal *= -std::expm1(-inc);
            row[k - 1] += p_float * p_coal;
        }
        row[j - 1] = 0.0;
        double s = 0.0;
        for (int k = 0; k < M; ++k) s += row[k];
        row[j - 1] = 1.0 - s;
    }
    const double beta = 1e-5, p2 = beta / Mh;        // uniform mix with denominator M+1 (quirk 13)
    for (auto &x : Phi) {
        if (x < 1e-20) x = 1e-20;
        x = x * (1 - beta) + p2;
    }
    return Phi;
}

// ---------------------------------------------------------------------------------------------------------------
// A10: conditioned SFS (conditioned_sfs.cpp:13-148)
// ---------------------------------------------------------------------------------------------------------------
inline double dcs_sorted(std::vector<double> &v) {
    std::sort(v.begin(), v.end(), [](double x, double y) { return std::fabs(x) > std::fabs(y); });
    if (v.empty()) return 0.0;
    double s = v[0], c = 0.0;
    for (size_t i = 1; i < v.size(); ++i) {
        const double y = c + v[i];
        const double u = v[i] - (y - c);
        const double t = y + s;
        const double w = y - (t - s);
        const double z = u + w;
        s = t + z;
        c = z - (s - t);
    }
    return s;
}

// raw CSFS per hidden state: out[m] is 3 x (n+1) row-major
inline std::vector<DMat> conditioned_sfs(const RateFunction &eta, const CsfsTables &tb) {
    const int n = tb.n;
    const int M = (int)eta.hidden_states.size() - 1;
    std::vector<DMat> csfs(M, DMat(3, n + 1));
    // ---- above ----
    std::vector<DMat> C_above(M, DMat(n + 1, std::max(n, 0)));
    if (n >= 1) {
#pragma omp parallel for schedule(dynamic)
        for (int j = 2; j < n + 3; ++j) eta.tjj_double_integral_above(n, j, C_above);
#pragma omp parallel for schedule(dynamic)
        for (int m = 0; m < M; ++m) {
            const DMat &Ca = C_above[m];
            std::vector<double> tmp0(n + 1), tmp2(n + 1), v(n);
            for (int j = 0; j < n + 1; ++j) {
                for (int i = 0; i < n; ++i) v[i] = tb.X0(i, j) * Ca(j, i);              // C0(i,j) = C(j,i)
                std::vector<double> w(v);
                tmp0[j] = dcs_sorted(w);
                for (int i = 0; i < n; ++i) v[i] = tb.X2(i, j) * Ca(n - j, i);          // C2(i,j) = C(n-j,i)
                w = v;
                tmp2[j] = dcs_sorted(w);
            }
            for (int b = 0; b < n; ++b) {
                double s0 = 0.0, s2 = 0.0;
                for (int j = 0; j < n + 1; ++j) { s0 += tmp0[j] * tb.Uinv_mp0(j, b); s2 += tmp2[j] * tb.Uinv_mp2(j, b); }
                csfs[m](0, 1 + b) += s0;
                csfs[m](2, b) += s2;
            }
        }
    }
    // ---- below ----
    DMat tjj_below(M, n + 1);
#pragma omp parallel for schedule(dynamic)
    for (int m = 0; m < M; ++m) eta.tjj_double_integral_below(n, m, tjj_below);
    for (int m = 0; m < M; ++m) {
        for (int b = 0; b < n; ++b) {
            double s = 0.0;
            for (int j = 0; j < n + 1; ++j) s += tjj_below(m, j) * tb.M0(j, b);
            csfs[m](0, 1 + b) += s;
        }
        for (int b = 0; b < n + 1; ++b) {
            double s = 0.0;
            for (int j = 0; j < n + 1; ++j) s += tjj_below(m, j) * tb.M1(j, b);
            csfs[m](1, b) += s;
        }
    }
    return csfs;
}

inline void incorporate_theta(std::vector<DMat> &csfs, double theta) {
    if (theta <= 0) throw std::runtime_error("mutation rate theta <= 0");
    for (auto &c : csfs) {
        double tauh = 0.0;
        for (double x : c.d) tauh += x;
        const double f = -std::expm1(-theta * tauh) / tauh;
        for (double &x : c.d) x *= f;
        double tot = 0.0;
        for (double x : c.d) tot += x;
        c(0, 0) = 1.0 - tot;
        for (double &x : c.d) if (x < 1e-10) x = 1e-10;
        for (double x : c.d)
            if (x < 0 || x > 1 || std::isnan(x)) throw std::runtime_error("csfs is not a probability distribution");
    }
}

// ---------------------------------------------------------------------------------------------------------------
// A6 + A7: one-population preparation
// ---------------------------------------------------------------------------------------------------------------
class OnePopPrep {
public:
    OnePopPrep(int n, const std::vector<double> &hs, double polarization_error)
        : n_(n), hs_(hs), pol_(polarization_error), tables_(csfs_tables(n)) {}

    // keys: [K][3] (a, b, nb); outputs pi [M], T [M*M] row-major, E [K*M]
    void compute(const ModelParams &mp, double theta, double rho, double alpha, const std::vector<int> &keys, int K,
                 std::vector<double> &pi, std::vector<double> &T, std::vector<double> &E) {
        RateFunction eta(mp, hs_);
        const int M = (int)hs_.size() - 1;
        // pi (inference_manager.cpp:56-69)
        pi.assign(M, 0.0);
        for (int m = 0; m < M - 1; ++m) pi[m] = std::exp(-eta.R(hs_[m])) - std::exp(-eta.R(hs_[m + 1]));
        pi[M - 1] = std::exp(-eta.R(hs_[M - 1]));
        double ps = 0.0;
        for (double &x : pi) { if (x < 1e-20) x = 1e-20; ps += x; }
        for (double &x : pi) x /= ps;
        T = compute_transition(eta, rho);
        std::vector<DMat> sfs = conditioned_sfs(eta, *tables_);
        incorporate_theta(sfs, theta);
        const std::vector<double> avg_ct = eta.average_coal_times();
        emission_probs(sfs, avg_ct, theta, alpha, keys, K, E);
    }

    // restated marginalisation machinery -----------------------------------------------------------------------
    typedef std::array<int, 3> Key;

    static double hypergeom_pdf(unsigned k, unsigned n1, unsigned n2, unsigned t) {
        if (t > n1 + n2) t = n1 + n2;
        if (k > n1 || k > t) return 0.0;
        if (t > n2 && k + n2 < t) return 0.0;
        auto lnchoose = [](unsigned nn, unsigned mm) {
            return std::lgamma(nn + 1.0) - std::lgamma(mm + 1.0) - std::lgamma(nn - mm + 1.0);
        };
        return std::exp(lnchoose(n1, k) + lnchoose(n2, t - k) - lnchoose(n1 + n2, t));
    }

    std::map<std::pair<int, int>, double> bins_for(const Key &bk) const {
        const int na = 2;
        auto is_mono = [&](const Key &k) { return k[0] == na && k[1] == k[2]; };
        std::set<Key> bins;
        std::vector<Key> todo;
        if (bk[0] == -1) for (int aa = 0; aa <= na; ++aa) todo.push_back(Key{aa, bk[1], bk[2]});
        else todo.push_back(bk);
        for (const Key &k : todo) bins.insert(k);   // bin_key<1>::run with cutoff 1.0: b/nb > 1 never holds
        std::map<Key, double> m;
        for (const Key &k : bins) {
            // marginalize_key<1>: lift nb -> n
            std::map<Key, double> probs;
            for (int n1 = k[1]; n1 <= n_ + k[1] - k[2]; ++n1) {
                const int n2 = n_ - n1;
                probs[Key{k[0], n1, n_}] += hypergeom_pdf(k[1], n1, n2, k[2]);
            }
            for (const auto &p : probs) {
                Key mbk = p.first;
                if (is_mono(mbk)) mbk = Key{0, 0, mbk[2]};
                m[mbk] += (1.0 - pol_) * p.second;
                const Key fk{na - mbk[0], mbk[2] - mbk[1], mbk[2]};
                m[fk] += pol_ * p.second;
            }
        }
        double s = 0.0;
        std::map<Key, double> m2;
        for (const auto &p : m) {
            if (p.second <= 0 || is_mono(p.first)) continue;
            m2[p.first] = p.second;
            s += p.second;
        }
        if (s <= 0) throw std::runtime_error("s<=0");
        std::map<std::pair<int, int>, double> out;
        for (const auto &p : m2) out[{p.first[0], p.first[1]}] += p.second / s;
        return out;
    }

    void emission_probs(const std::vector<DMat> &sfs, const std::vector<double> &avg_ct, double theta, double alpha,
                        const std::vector<int> &keys, int K, std::vector<double> &E) const {
        const int M = (int)sfs.size();
        std::vector<double> e2((size_t)M * 2);
        for (int m = 0; m < M; ++m) {
            if (std::isnan(avg_ct[m])) { e2[2 * m] = 1e-20; e2[2 * m + 1] = 1e-20; }
            else {
                const double le = -2.0 * alpha * theta * avg_ct[m];
                e2[2 * m] = std::exp(le);
                e2[2 * m + 1] = -std::expm1(le);
            }
        }
        E.assign((size_t)K * M, 0.0);
        for (int k = 0; k < K; ++k) {
            const Key bk{keys[3 * k], keys[3 * k + 1], keys[3 * k + 2]};
            const bool reduced = bk[2] == 0, miss = bk[0] == -1;
            double *e = &E[(size_t)k * M];
            if (reduced && (miss || bk[0] >= 0)) {
                for (int m = 0; m < M; ++m) e[m] = miss ? 1.0 : e2[2 * m + (bk[0] % 2)];
            } else {
                const auto bins = bins_for(bk);
                for (const auto &p : bins)
                    for (int m = 0; m < M; ++m) e[m] += p.second * sfs[m](p.first.first, p.first.second);
            }
            double mx = e[0], mn = e[0];
            for (int m = 1; m < M; ++m) { mx = std::max(mx, e[m]); mn = std::min(mn, e[m]); }
            if (mx > 1.0 || mn <= 0.0) throw std::runtime_error("probability vector not in [0, 1]");
        }
    }

private:
    int n_;
    std::vector<double> hs_;
    double pol_;
    std::shared_ptr<const CsfsTables> tables_;
};

}  // namespace smcpp_host
