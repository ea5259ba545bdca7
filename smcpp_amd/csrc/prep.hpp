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
#include <functional>
#include <exception>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <random>
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

// On-disk store of the n-only tables (the role of MatrixCache's cereal archive, src/matrix_cache.cpp:46-110): one
// little-endian file `<prefix>.n<N>` per sample size: magic, n, then the six matrices as (rows, cols, doubles).  Written
// to a temporary name and renamed, so concurrent processes never read a partial file.  Empty prefix = no store.
inline std::string &csfs_cache_prefix() { static std::string p; return p; }

namespace detail {
inline bool load_tables(const std::string &fn, int n, CsfsTables &t) {
    FILE *f = fopen(fn.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    char magic[8];
    int nn = -1;
    if (fread(magic, 1, 8, f) == 8 && !memcmp(magic, "SMCPPT01", 8) && fread(&nn, sizeof(int), 1, f) == 1 && nn == n) {
        ok = true;
        for (DMat *m : {&t.X0, &t.X2, &t.M0, &t.M1, &t.Uinv_mp0, &t.Uinv_mp2}) {
            int rc[2];
            if (fread(rc, sizeof(int), 2, f) != 2 || rc[0] < 0 || rc[1] < 0 || (long long)rc[0] * rc[1] > (1ll << 26)) { ok = false; break; }
            *m = DMat(rc[0], rc[1]);
            if (fread(m->d.data(), sizeof(double), m->d.size(), f) != m->d.size()) { ok = false; break; }
        }
        // every table at the shape csfs_tables() builds (a truncated or foreign file must not be indexed out of bounds)
        ok = ok && t.X0.r == n && t.X0.c == n + 1 && t.X2.r == t.X0.r && t.X2.c == t.X0.c &&
             t.M1.r == n + 1 && t.M1.c == n + 1 && t.M0.r == n + 1 && t.M0.c == n &&
             t.Uinv_mp0.r == n + 1 && t.Uinv_mp0.c == n && t.Uinv_mp2.r == n + 1 && t.Uinv_mp2.c == n;
    }
    fclose(f);
    t.n = n;
    return ok;
}
inline void store_tables(const std::string &fn, const CsfsTables &t) {
    const std::string tmp = fn + ".tmp" + std::to_string((long long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;                                   // an unwritable cache directory is not an error (as in the reference)
    bool ok = fwrite("SMCPPT01", 1, 8, f) == 8 && fwrite(&t.n, sizeof(int), 1, f) == 1;
    for (const DMat *m : {&t.X0, &t.X2, &t.M0, &t.M1, &t.Uinv_mp0, &t.Uinv_mp2}) {
        const int rc[2] = {m->r, m->c};
        ok = ok && fwrite(rc, sizeof(int), 2, f) == 2 && fwrite(m->d.data(), sizeof(double), m->d.size(), f) == m->d.size();
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), fn.c_str()) != 0) remove(tmp.c_str());
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
    const std::string cache_fn = csfs_cache_prefix().empty() ? std::string() : csfs_cache_prefix() + ".n" + std::to_string(n);
    if (!cache_fn.empty() && detail::load_tables(cache_fn, n, *t)) {
        memo.emplace(n, t);
        return t;
    }
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
    if (!cache_fn.empty()) detail::store_tables(cache_fn, *t);
    memo.emplace(n, t);
    return t;
}

// ---------------------------------------------------------------------------------------------------------------
// forward-mode dual numbers (the role of Eigen::AutoDiffScalar `adouble`, include/common.h:22-25): value + up to MAXD
// directional derivatives.  The active number of directions is a thread-local set by DualScope.
// ---------------------------------------------------------------------------------------------------------------
constexpr int MAXD = 64;
inline int &dual_nder() { static thread_local int n = 0; return n; }
struct DualScope {
    int prev;
    explicit DualScope(int n) : prev(dual_nder()) {
        if (n > MAXD) throw std::runtime_error("too many derivative directions (max 64)");
        dual_nder() = n;
    }
    ~DualScope() { dual_nder() = prev; }
};

template <typename F>
struct Dual {
    F v;
    F d[MAXD];
    Dual() : v(0) { for (int i = 0; i < dual_nder(); ++i) d[i] = 0; }
    Dual(F x) : v(x) { for (int i = 0; i < dual_nder(); ++i) d[i] = 0; }
    template <typename G>
    explicit Dual(const Dual<G> &o) : v((F)o.v) { for (int i = 0; i < dual_nder(); ++i) d[i] = (F)o.d[i]; }
    // copies move the ACTIVE directions only (the implicit ones copy all MAXD slots: 520 bytes per number, a kilobyte in long
    // double - most of the time of every dual routine with a handful of directions)
    Dual(const Dual &o) : v(o.v) { for (int i = 0, n = dual_nder(); i < n; ++i) d[i] = o.d[i]; }
    Dual &operator=(const Dual &o) { v = o.v; for (int i = 0, n = dual_nder(); i < n; ++i) d[i] = o.d[i]; return *this; }
};
#define SMCPP_DUAL_LOOP for (int i_ = 0, n_ = dual_nder(); i_ < n_; ++i_)
template <typename F> inline Dual<F> operator+(const Dual<F> &a, const Dual<F> &b) { Dual<F> r; r.v = a.v + b.v; SMCPP_DUAL_LOOP r.d[i_] = a.d[i_] + b.d[i_]; return r; }
template <typename F> inline Dual<F> operator-(const Dual<F> &a, const Dual<F> &b) { Dual<F> r; r.v = a.v - b.v; SMCPP_DUAL_LOOP r.d[i_] = a.d[i_] - b.d[i_]; return r; }
template <typename F> inline Dual<F> operator*(const Dual<F> &a, const Dual<F> &b) { Dual<F> r; r.v = a.v * b.v; SMCPP_DUAL_LOOP r.d[i_] = a.d[i_] * b.v + a.v * b.d[i_]; return r; }
template <typename F> inline Dual<F> operator/(const Dual<F> &a, const Dual<F> &b) { Dual<F> r; const F ib = 1 / b.v; r.v = a.v * ib; SMCPP_DUAL_LOOP r.d[i_] = (a.d[i_] - r.v * b.d[i_]) * ib; return r; }
template <typename F> inline Dual<F> operator-(const Dual<F> &a) { Dual<F> r; r.v = -a.v; SMCPP_DUAL_LOOP r.d[i_] = -a.d[i_]; return r; }
template <typename F, typename G> inline Dual<F> operator+(const Dual<F> &a, G b) { Dual<F> r(a); r.v += (F)b; return r; }
template <typename F, typename G> inline Dual<F> operator+(G b, const Dual<F> &a) { return a + b; }
template <typename F, typename G> inline Dual<F> operator-(const Dual<F> &a, G b) { Dual<F> r(a); r.v -= (F)b; return r; }
template <typename F, typename G> inline Dual<F> operator-(G b, const Dual<F> &a) { Dual<F> r = -a; r.v += (F)b; return r; }
template <typename F, typename G> inline Dual<F> operator*(const Dual<F> &a, G b) { Dual<F> r; r.v = a.v * (F)b; SMCPP_DUAL_LOOP r.d[i_] = a.d[i_] * (F)b; return r; }
template <typename F, typename G> inline Dual<F> operator*(G b, const Dual<F> &a) { return a * b; }
template <typename F, typename G> inline Dual<F> operator/(const Dual<F> &a, G b) { return a * ((F)1 / (F)b); }
template <typename F, typename G> inline Dual<F> operator/(G b, const Dual<F> &a) { Dual<F> r; r.v = (F)b / a.v; SMCPP_DUAL_LOOP r.d[i_] = -r.v * a.d[i_] / a.v; return r; }
template <typename F> inline Dual<F> &operator+=(Dual<F> &a, const Dual<F> &b) { a.v += b.v; SMCPP_DUAL_LOOP a.d[i_] += b.d[i_]; return a; }
template <typename F> inline Dual<F> &operator-=(Dual<F> &a, const Dual<F> &b) { a.v -= b.v; SMCPP_DUAL_LOOP a.d[i_] -= b.d[i_]; return a; }
template <typename F, typename G> inline Dual<F> &operator+=(Dual<F> &a, G b) { a.v += (F)b; return a; }
template <typename F, typename G> inline Dual<F> &operator*=(Dual<F> &a, G b) { a.v *= (F)b; SMCPP_DUAL_LOOP a.d[i_] *= (F)b; return a; }
template <typename F> inline Dual<F> &operator*=(Dual<F> &a, const Dual<F> &b) { a = a * b; return a; }
template <typename F> inline Dual<F> &operator/=(Dual<F> &a, const Dual<F> &b) { a = a / b; return a; }
template <typename F, typename G> inline Dual<F> &operator/=(Dual<F> &a, G b) { a = a / b; return a; }
inline double m_exp(double x) { return std::exp(x); }
inline long double m_exp(long double x) { return expl(x); }
inline double m_expm1(double x) { return std::expm1(x); }
inline long double m_expm1(long double x) { return expm1l(x); }
inline double m_log(double x) { return std::log(x); }
inline long double m_log(long double x) { return logl(x); }
inline double m_log1p(double x) { return std::log1p(x); }
inline long double m_log1p(long double x) { return log1pl(x); }
inline double m_sqrt(double x) { return std::sqrt(x); }
inline long double m_sqrt(long double x) { return sqrtl(x); }
inline double m_sinh(double x) { return std::sinh(x); }
inline long double m_sinh(long double x) { return sinhl(x); }
inline double m_cosh(double x) { return std::cosh(x); }
inline long double m_cosh(long double x) { return coshl(x); }
template <typename F> inline Dual<F> m_exp(const Dual<F> &a) { Dual<F> r; r.v = m_exp(a.v); SMCPP_DUAL_LOOP r.d[i_] = a.d[i_] * r.v; return r; }
template <typename F> inline Dual<F> m_expm1(const Dual<F> &a) { Dual<F> r; r.v = m_expm1(a.v); const F e = m_exp(a.v); SMCPP_DUAL_LOOP r.d[i_] = a.d[i_] * e; return r; }
template <typename F> inline Dual<F> m_log(const Dual<F> &a) { Dual<F> r; r.v = m_log(a.v); SMCPP_DUAL_LOOP r.d[i_] = a.d[i_] / a.v; return r; }
template <typename F> inline Dual<F> m_log1p(const Dual<F> &a) { Dual<F> r; r.v = m_log1p(a.v); SMCPP_DUAL_LOOP r.d[i_] = a.d[i_] / (1 + a.v); return r; }
template <typename F> inline Dual<F> m_sqrt(const Dual<F> &a) { Dual<F> r; r.v = m_sqrt(a.v); SMCPP_DUAL_LOOP r.d[i_] = a.d[i_] / (2 * r.v); return r; }
template <typename F> inline Dual<F> m_sinh(const Dual<F> &a) { Dual<F> r; r.v = m_sinh(a.v); const F c = m_cosh(a.v); SMCPP_DUAL_LOOP r.d[i_] = a.d[i_] * c; return r; }
inline double sval(double x) { return x; }
inline long double sval(long double x) { return x; }
template <typename F> inline F sval(const Dual<F> &x) { return x.v; }
typedef Dual<double> dual;

template <typename S>
struct ModelParamsT {
    std::vector<S> a;           // piece sizes (with derivative seeds when S is dual)
    std::vector<double> s;      // piece lengths carry no derivatives (_smcpp.pyx:78-81)
};

// ---------------------------------------------------------------------------------------------------------------
// A9: piecewise-constant rate function
// ---------------------------------------------------------------------------------------------------------------
template <typename S>
class RateFunctionT {
public:
    RateFunctionT() {}                               // (filled member by member: the value part of a dual rate function)
    RateFunctionT(const ModelParamsT<S> &p, const std::vector<double> &hs) : hidden_states(hs) {
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
        Rrng.assign(K + 1, S(0.0));
        for (int k = 0; k < K; ++k) Rrng[k + 1] = Rrng[k] + ada[k] * (ts[k + 1] - ts[k]);
    }

    S R(double t) const {
        auto ti = std::upper_bound(ts.begin(), ts.end(), t) - 1;
        const int ip = (int)(ti - ts.begin());
        return Rrng[ip] + ada[ip] * (t - *ti);
    }

    // R at a time that itself carries derivatives (jcsfs.cpp:127: R of a random coalescence time)
    S R_at(const S &t) const {
        auto ti = std::upper_bound(ts.begin(), ts.end(), (double)sval(t)) - 1;
        const int ip = (int)(ti - ts.begin());
        return Rrng[ip] + ada[ip] * (t - *ti);
    }

    // inverse of the cumulative hazard (piecewise_constant_rate_function.cpp:415-420)
    S Rinv(const S &y) const {
        int ip = 0;
        while (ip + 1 < (int)Rrng.size() && !(sval(y) < sval(Rrng[ip + 1]))) ++ip;
        if (ip >= K) ip = K - 1;
        return (y - Rrng[ip]) / ada[ip] + ts[ip];
    }

    // coalescence time conditioned on [a, b): inverse-cdf draw from Exp(1) truncated to [R(a), R(b))
    // (piecewise_constant_rate_function.cpp:337-368); one std::mt19937 per call, seeded by the caller
    S random_time(double a, double b, unsigned long long seed) const {
        std::mt19937 gen(seed);
        return random_time(a, b, gen);
    }
    S random_time(double a, double b, std::mt19937 &gen) const {
        const double unif = std::uniform_real_distribution<double>{0.0, 1.0}(gen);
        const S Ra = R(a);
        if (std::isinf(b)) return Rinv(Ra - std::log1p(-unif));
        const S Rb = R(b);
        return Rinv(Ra - m_log1p(m_expm1(-(Rb - Ra)) * unif));
    }

    // int_a^b exp(-(R(t) + log_denom)) dt
    S R_integral(double a, double b, const S &log_denom) const {
        const int ip_a = (int)(std::upper_bound(ts.begin(), ts.end(), a) - 1 - ts.begin());
        int ip_b = (int)(std::upper_bound(ts.begin(), ts.end(), b) - 1 - ts.begin());
        if (std::isinf(b)) ip_b = (int)ts.size() - 2;
        S ret(0.0);
        for (int i = ip_a; i < ip_b + 1; ++i) {
            const double left = std::max(a, ts[i]), right = std::min(b, ts[i + 1]);
            const double diff = right - left;
            S r = m_exp(-(R(left) + log_denom));
            if (sval(ada[i]) > 0.0) {
                if (!std::isinf(diff)) r *= -m_expm1(-diff * ada[i]);
                r /= ada[i];
            } else r *= diff;
            ret += r;
        }
        return ret;
    }

    // NaN value marks "no coalescence possible in this interval" (piecewise_constant_rate_function.cpp:380-388)
    std::vector<S> average_coal_times() const {
        std::vector<S> ret;
        for (size_t i = 1; i < hidden_states.size(); ++i) {
            const S R0 = Rrng[hs_indices[i - 1]], R1 = Rrng[hs_indices[i]];
            if (sval(R0) == sval(R1)) { ret.push_back(S(std::numeric_limits<double>::quiet_NaN())); continue; }
            S log_denom = -R0;
            const bool inf = std::isinf(ts[hs_indices[i]]);
            if (!inf) log_denom += m_log(-m_expm1(-(R1 - R0)));
            S x = hidden_states[i - 1] * m_exp(-(R0 + log_denom)) +
                  R_integral(ts[hs_indices[i - 1]], ts[hs_indices[i]], log_denom);
            if (!inf) x -= hidden_states[i] * m_exp(-(R1 + log_denom));
            ret.push_back(x);
            if (sval(x) > hidden_states[i] || sval(x) < hidden_states[i - 1])
                throw std::runtime_error("erroneous average coalescence time");
        }
        return ret;
    }

    // ---- integrals used by the conditioned SFS (piecewise_constant_rate_function.cpp:87-138,198-334) ----
    static S below_helper(long rate, double tsm, double tsm1, const S &ad, const S &Rr, const S &log_denom) {
        if (sval(ad) == 0) return S(0.0);
        const long l1r = 1 + rate;
        const double l1rinv = 1.0 / (double)l1r;
        const S adadiff = ad * (tsm1 - tsm);
        if (rate == 0) {
            if (tsm1 == INFINITY) return m_exp(-Rr - log_denom) / ad;
            return m_exp(-Rr - log_denom) * (1.0 - m_exp(-adadiff) * (1.0 + adadiff)) / ad;
        }
        if (tsm1 == INFINITY) return m_exp(-(double)l1r * Rr - log_denom) * (1.0 - l1rinv) / ((double)rate * ad);
        return m_exp(-(double)l1r * Rr - log_denom) * (m_expm1(-(double)l1r * adadiff) * l1rinv - m_expm1(-adadiff)) /
               ((double)rate * ad);
    }
    static S above_helper(long rate, long lam, double tsm, double tsm1, const S &ad, const S &Rr, const S &log_coef) {
        if (sval(ad) == 0) return S(0.0);
        const S adadiff = ad * (tsm1 - tsm);
        const double l1 = (double)(lam + 1), rt = (double)rate;
        if (rate == 0)
            return m_exp(-l1 * Rr + log_coef) * (m_expm1(-l1 * adadiff) + l1 * adadiff) / l1 / l1 / ad;
        if (lam + 1 == rate) {
            if (tsm1 == INFINITY) return m_exp(-rt * Rr + log_coef) / rt / rt / ad;
            return m_exp(-rt * Rr + log_coef) * (1.0 - m_exp(-rt * adadiff) * (1.0 + rt * adadiff)) / rt / rt / ad;
        }
        if (tsm1 == INFINITY) return m_exp(-l1 * Rr + log_coef) / l1 / rt / ad;
        if (rate < lam + 1)
            return -m_exp(-l1 * Rr + log_coef) *
                   (m_expm1(-l1 * adadiff) / l1 + (m_exp(-rt * adadiff) * -m_expm1(-(l1 - rt) * adadiff) / (l1 - rt))) / rt / ad;
        return -m_exp(-l1 * Rr + log_coef) *
               (m_expm1(-l1 * adadiff) / l1 + (m_exp(-l1 * adadiff) * m_expm1(-(rt - l1) * adadiff) / (l1 - rt))) / rt / ad;
    }
    static S single_integral(long rate, double tsm, double tsm1, const S &ad, const S &Rr, const S &log_coef) {
        if (rate == 0) return m_exp(log_coef) * (tsm1 - tsm);
        S ret = m_exp(-(double)rate * Rr + log_coef);
        if (tsm1 < INFINITY) ret *= -m_expm1(-(double)rate * ad * (tsm1 - tsm));
        ret /= ad * (double)rate;
        return ret;
    }
    static long nC2(long n) { return n * (n - 1) / 2; }

    // row jj-2 of C[h] ((n+1) x n each, row-major vectors), h over hidden states
    void tjj_double_integral_above(int n, long jj, std::vector<std::vector<S>> &C) const {
        const long lam = nC2(jj) - 1;
        for (size_t h = 0; h + 1 < hs_indices.size(); ++h) {
            for (int j = 0; j < n; ++j) C[h][(size_t)(jj - 2) * n + j] = S(0.0);
            const S Rh = Rrng[hs_indices[h]], Rh1 = Rrng[hs_indices[h + 1]];
            S log_denom = -Rh;
            if (sval(Rh1) != INFINITY) log_denom += m_log(-m_expm1(-(Rh1 - Rh)));
            for (int m = hs_indices[h]; m < hs_indices[h + 1]; ++m)
                for (int j = 2; j < n + 2; ++j) {
                    const long rate = nC2(j);
                    S &tgt = C[h][(size_t)(jj - 2) * n + (j - 2)];
                    tgt += above_helper(rate, lam, ts[m], ts[m + 1], ada[m], Rrng[m], -log_denom);
                    S log_coef = -log_denom, fac(0.0);
                    const long rp = lam + 1 - rate;
                    const double rpd = (double)rp;
                    const S Rm1 = Rrng[m + 1], Rm = Rrng[m];
                    if (rp == 0) fac = Rm1 - Rm;
                    else if (rp < 0) {
                        if (-rpd * sval(Rm1 - Rm) > 20) { log_coef += -rpd * Rm1; fac = S(-1.0 / rpd); }
                        else { log_coef += -rpd * Rm; fac = -m_expm1(-rpd * (Rm1 - Rm)) / rpd; }
                    } else {
                        if (-rpd * sval(Rm - Rm1) > 20) { log_coef += -rpd * Rm; fac = S(1.0 / rpd); }
                        else { log_coef += -rpd * Rm1; fac = m_expm1(-rpd * (Rm - Rm1)) / rpd; }
                    }
                    for (int k = m + 1; k < K; ++k)
                        tgt += single_integral(rate, ts[k], ts[k + 1], ada[k], Rrng[k], log_coef) * fac;
                }
        }
    }

    // row h of tgt (M x (n+1), row-major vector)
    void tjj_double_integral_below(int n, int h, std::vector<S> &tgt) const {
        const S Rh = Rrng[hs_indices[h]], Rh1 = Rrng[hs_indices[h + 1]];
        S log_denom = -Rh;
        if (sval(Rh1) != INFINITY) log_denom += m_log(-m_expm1(-(Rh1 - Rh)));
        for (int m = hs_indices[h]; m < hs_indices[h + 1]; ++m) {
            const S Rm = Rrng[m], Rm1 = Rrng[m + 1];
            const S log_coef = -Rm;
            S fac(1.0);
            if (m < K - 1) fac = -m_expm1(-(Rm1 - Rm));
            for (int j = 2; j < n + 3; ++j) {
                const long rate = nC2(j) - 1;
                S v = below_helper(rate, ts[m], ts[m + 1], ada[m], Rrng[m], log_denom);
                for (int k = 0; k < m; ++k)
                    v += fac * single_integral(rate, ts[k], ts[k + 1], ada[k], Rrng[k], log_coef - log_denom);
                tgt[(size_t)h * (n + 1) + (j - 2)] += v;
            }
        }
    }

    std::vector<double> hidden_states, ts;
    std::vector<S> ada, Rrng;
    std::vector<int> hs_indices;
    int K = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// A8: transition matrix (transition.cpp:113-262)
// ---------------------------------------------------------------------------------------------------------------
namespace detail {
typedef long double ld;
template <typename L> struct M3T { L m[3][3]; };
template <typename L> inline M3T<L> m3_identity() { M3T<L> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = L((ld)(i == j)); return r; }
template <typename L> inline M3T<L> m3_mul(const M3T<L> &a, const M3T<L> &b) {
    M3T<L> r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { L s((ld)0); for (int k = 0; k < 3; ++k) s += a.m[i][k] * b.m[k][j]; r.m[i][j] = s; }
    return r;
}
// closed-form exponential of c_rho * A_rho + c_eta * A_eta (transition.cpp:113-130); the factors e * cosh and
// e * sinh are combined so nothing overflows (the reference relies on 256-bit MPFR here)
template <typename L> inline M3T<L> matrix_exp(const L &c_rho, const L &c_eta) {
    const L sq = m_sqrt(4 * c_eta * c_eta + c_rho * c_rho);
    const L y = c_eta + c_rho / (ld)2, x = sq / (ld)2;
    L ec((ld)1), es((ld)0.5L);   // e*cosh(x), e*sinh(x)/sq
    if (sval(sq) != 0) {
        if (sval(x) < 40.0L) {
            // two transcendentals instead of four: e^-y and e^x - 1; e^-x = 1 / e^x, sinh x = (t + t e^-x) / 2 with t = e^x - 1
            // (no cancellation for small x, and e^x stays far from overflow)
            const L ey = m_exp(-y), t = m_expm1(x);
            const L exm = (ld)1 / ((ld)1 + t);
            ec = (ld)0.5L * (ey * ((ld)1 + t) + ey * exm);
            es = ey * ((ld)0.5L * (t + t * exm)) / sq;
        } else {
            const L ep = m_exp(x - y), em = m_exp(-x - y);
            ec = (ld)0.5L * (ep + em);
            es = (ld)0.5L * (ep - em) / sq;
        }
    }
    M3T<L> Qm;
    Qm.m[0][0] = ec + (2 * c_eta - c_rho) * es;
    Qm.m[0][1] = 2 * c_rho * es;
    Qm.m[0][2] = (ld)1 - Qm.m[0][0] - Qm.m[0][1];
    Qm.m[1][0] = 2 * c_eta * es;
    Qm.m[1][1] = ec - (2 * c_eta - c_rho) * es;
    Qm.m[1][2] = (ld)1 - Qm.m[1][0] - Qm.m[1][1];
    Qm.m[2][0] = L((ld)0); Qm.m[2][1] = L((ld)0); Qm.m[2][2] = L((ld)1);
    return Qm;
}
// The same exponential with its two partial derivatives in closed form (no second pass through the formulas with dual numbers):
// with S = sq es = e^-y sinh x,  d ec = x' S - y' ec,  d S = x' ec - y' S,  d es = (d S - es d sq) / sq.
inline void matrix_exp_partials(ld c_rho, ld c_eta, M3T<ld> &Q, M3T<ld> &Qr, M3T<ld> &Qe, bool partials = true) {
    // (values: the operations of matrix_exp<ld>, in its order)
    const ld sq = sqrtl(4 * c_eta * c_eta + c_rho * c_rho);
    const ld y = c_eta + c_rho / (ld)2, x = sq / (ld)2;
    ld ec = 1.0L, es = 0.5L;
    if (sq != 0) {
        if (x < 40.0L) {
            const ld ey = expl(-y), t = expm1l(x);
            const ld exm = (ld)1 / ((ld)1 + t);
            ec = (ld)0.5L * (ey * ((ld)1 + t) + ey * exm);
            es = ey * ((ld)0.5L * (t + t * exm)) / sq;
        } else {
            const ld ep = expl(x - y), em = expl(-x - y);
            ec = (ld)0.5L * (ep + em);
            es = (ld)0.5L * (ep - em) / sq;
        }
    }
    Q.m[0][0] = ec + (2 * c_eta - c_rho) * es;
    Q.m[0][1] = 2 * c_rho * es;
    Q.m[0][2] = (ld)1 - Q.m[0][0] - Q.m[0][1];
    Q.m[1][0] = 2 * c_eta * es;
    Q.m[1][1] = ec - (2 * c_eta - c_rho) * es;
    Q.m[1][2] = (ld)1 - Q.m[1][0] - Q.m[1][1];
    Q.m[2][0] = 0.0L; Q.m[2][1] = 0.0L; Q.m[2][2] = 1.0L;
    if (!partials) return;                            // (values only: a caller without derivative directions)
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { Qr.m[a][b] = 0.0L; Qe.m[a][b] = 0.0L; }
    if (sq == 0) return;                              // (no time, no change: never reached with positive piece lengths)
    const ld Sv = sq * es;
    const ld w = 2 * c_eta - c_rho;
    for (int v = 0; v < 2; ++v) {                     // v = 0: d / d c_rho, v = 1: d / d c_eta
        const ld sqv = v ? 4 * c_eta / sq : c_rho / sq, xv = sqv / 2.0L, yv = v ? 1.0L : 0.5L;
        const ld ecv = xv * Sv - yv * ec, Svv = xv * ec - yv * Sv, esv = (Svv - es * sqv) / sq;
        const ld wv = v ? 2.0L : -1.0L, crv = v ? 0.0L : 1.0L, cev = v ? 1.0L : 0.0L;
        M3T<ld> &D = v ? Qe : Qr;
        D.m[0][0] = ecv + wv * es + w * esv;
        D.m[0][1] = 2 * (crv * es + c_rho * esv);
        D.m[0][2] = -D.m[0][0] - D.m[0][1];
        D.m[1][0] = 2 * (cev * es + c_eta * esv);
        D.m[1][1] = ecv - (wv * es + w * esv);
        D.m[1][2] = -D.m[1][0] - D.m[1][1];
    }
}
template <typename S> struct WideOf;
template <> struct WideOf<double> {
    typedef ld type;
    static ld up(double x) { return (ld)x; }
    static double down(ld x) { return (double)x; }
};
template <> struct WideOf<dual> {
    typedef Dual<ld> type;
    static Dual<ld> up(const dual &x) { return Dual<ld>(x); }
    static dual down(const Dual<ld> &x) { return dual(x); }
};
}  // namespace detail

// Stage 1: everything that is O(pieces + states): the wide 3 x 3 chain, one partial exponential per state, the cumulative
// hazards.  Stage 2 (transition_expand) fills the M x M matrix from these generators: below the diagonal T(i, c) = expm_diff[c]
// (the column only), above it p_float[i] / Ek[i] * (Ek[c] * qk[c + 1]) (rank one), the diagonal closes the row; then the 1e-20
// floor and the uniform mix.  (The scan chains of the engine work on exactly this structure, chains_ss.hpp.)
template <typename S>
struct TransitionGenerators {
    int Mh = 0;                                        // breakpoints (the reference's `this->M`), M = Mh - 1 states
    std::vector<S> expm_diff, p_float, Ek, qk, inc_k;  // [M - 1], [Mh] (index j = 1 .. M), [Mh], [Mh], [Mh]
};

template <typename S>
inline TransitionGenerators<S> transition_generators(const RateFunctionT<S> &eta, double rho, const std::vector<S> &avg) {
    using namespace detail;
    typedef typename WideOf<S>::type L;
    typedef WideOf<S> W;
    const std::vector<double> &ts = eta.ts;
    const std::vector<S> &ada = eta.ada;
    const std::vector<int> &hsi = eta.hs_indices;
    const int Mh = (int)eta.hidden_states.size();   // the reference's `this->M` (breakpoints)
    const int M = Mh - 1;
    const int nts = (int)ts.size();
    TransitionGenerators<S> g;
    g.Mh = Mh;
    std::vector<M3T<L>> expms(nts, m3_identity<L>()), prods(nts, m3_identity<L>());
    for (int i = hsi[0] + 1; i < nts; ++i) {
        if (!std::isinf(ts[i])) {
            const double delta = ts[i] - ts[i - 1];
            expms[i] = matrix_exp<L>(L((ld)delta * (ld)rho), W::up(ada[i - 1]) * (ld)delta);
        }   // infinite end: the reference push_back()s instead of assigning, so expm_U[i] stays the identity
        prods[i] = m3_mul(prods[i - 1], expms[i]);
    }
    // the reference keeps expms / expm_prods in working (double) precision after the wide computation
    auto narrow = [](const M3T<L> &m) { M3T<S> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = W::down(m.m[i][j]); return r; };
    std::vector<int> avc_ip(M);
    for (int j = 0; j < M; ++j)
        avc_ip[j] = (int)(std::upper_bound(ts.begin(), ts.end(), (double)sval(avg[j])) - ts.begin()) - 1;
    g.expm_diff.assign(std::max(0, M - 1), S(0.0));
    for (int k = 1; k < M; ++k)
        g.expm_diff[k - 1] = W::down(prods[hsi[k]].m[0][2]) - W::down(prods[hsi[k - 1]].m[0][2]);
    // Upper part: the reference evaluates, for every pair j < k, exp(-sum of the increments of the states between them) and
    // -expm1(-inc_k) (transition.cpp:217-233: M^2 / 2 pairs, two transcendental calls each - 0.1 ms at M = 64, serial).  Both
    // factor: with C_t = inc_1 + ... + inc_t the first is exp(-C_{k-1}) / exp(-C_j), the second depends on k only: O(M)
    // calls, one division per pair (differences of ~3e-16 relative against the pairwise form; the ratio is not used where
    // exp(-C_j) has left the normal range).
    g.inc_k.assign(Mh, S(0.0)); g.Ek.assign(Mh, S(1.0)); g.qk.assign(Mh, S(1.0)); g.p_float.assign(Mh, S(0.0));
    std::vector<S> Ccum(Mh, S(0.0));
    for (int k = 1; k < Mh; ++k) {
        S inc(0.0);
        for (int jj = hsi[k - 1]; jj < hsi[k]; ++jj) inc += ada[jj] * (ts[jj + 1] - ts[jj]);
        g.inc_k[k] = inc;
        Ccum[k] = Ccum[k - 1] + inc;
        g.Ek[k] = m_exp(-Ccum[k]);
        g.qk[k] = std::isinf((double)sval(inc)) ? S(1.0) : S(-m_expm1(-inc));
    }
    for (int j = 1; j < Mh; ++j) {
        const S rct = avg[j - 1];
        const int rct_ip = avc_ip[j - 1];
        M3T<S> A = m3_identity<S>();
        for (int ell = hsi[j - 1]; ell < rct_ip; ++ell) A = m3_mul(A, narrow(expms[ell]));
        const S delta = rct - ts[rct_ip];
        const S c_eta = ada[rct_ip] * delta;
        const S c_rho = delta * rho;
        A = m3_mul(A, narrow(matrix_exp<L>(W::up(c_rho), W::up(c_eta))));
        const M3T<S> B = m3_mul(narrow(prods[hsi[j - 1]]), A);
        S Rj = c_eta;
        Rj += ada[rct_ip] * (ts[rct_ip + 1] - rct);
        for (int jj = rct_ip + 2; jj < hsi[j]; ++jj) Rj += ada[jj] * (ts[jj + 1] - ts[jj]);
        g.p_float[j] = B.m[0][1] * m_exp(-Rj);
    }
    return g;
}

template <typename S>
inline std::vector<S> transition_expand(const TransitionGenerators<S> &g) {
    const int Mh = g.Mh, M = Mh - 1;
    std::vector<S> Phi((size_t)M * M, S(0.0));
    for (int j = 1; j < Mh; ++j) {
        S *row = &Phi[(size_t)(j - 1) * M];
        for (int k = 0; k < j - 1; ++k) row[k] = g.expm_diff[k];
        const S &p_float = g.p_float[j];
        if (sval(g.Ek[j]) > 1e-250) {
            const S pf = p_float / g.Ek[j];
            for (int k = j + 1; k < Mh; ++k) row[k - 1] += pf * (g.Ek[k - 1] * g.qk[k]);
        } else {
            S Rjk1(0.0);
            for (int k = j + 1; k < Mh; ++k) {
                S p_coal = m_exp(-Rjk1);
                Rjk1 += g.inc_k[k];
                row[k - 1] += p_float * (p_coal * g.qk[k]);
            }
        }
        row[j - 1] = S(0.0);
        S sm(0.0);
        for (int k = 0; k < M; ++k) sm += row[k];
        row[j - 1] = 1.0 - sm;
    }
    const double beta = 1e-5, p2 = beta / Mh;        // uniform mix with denominator M+1 (quirk 13)
    for (auto &x : Phi) {
        if (sval(x) < 1e-20) x = S(1e-20);
        x = x * (1 - beta) + p2;
    }
    return Phi;
}

template <typename S>
inline std::vector<S> compute_transition(const RateFunctionT<S> &eta, double rho) {
    return transition_expand<S>(transition_generators<S>(eta, rho, eta.average_coal_times()));
}

// The generators in the form the engine's kernels consume, with forward-mode derivative PLANES [x][nder] (directions
// contiguous): ed [M - 1] = expm_diff, pf [M] = p_float / Ek per row (0 for the last row, which has no entry above the
// diagonal), W [M] = Ek[c] qk[c + 1] per column c >= 1.  `ok` is false when a row needs the pairwise fallback of
// transition_expand (cumulative hazard beyond 575: exp(-C) has left the normal range); callers then take the generic duals.
struct TransitionGenJac {
    int M = 0, nder = 0;
    bool ok = true;
    std::vector<double> ed, pf, W, ded, dpf, dW;
};

// Values as transition_generators<double> forms them (the same operations; every exponential is evaluated ONCE, together with its
// partial derivatives in closed form); derivatives by the chain rule over plain arrays instead of carrying `nder` directions
// through every operation of the wide 3 x 3 chain.  Every factor exp(c_rho A_rho + c_eta A_eta) depends on the parameters
// through ONE scalar (c_eta = delta / a_piece), so the directions enter as scalar multiples.  Only row 0 of the running product
// is used, and of it only u = (P00, P01): rows of these matrices sum to one identically, so d P02 = -(d P00 + d P01), and u -
// the probability of not having coalesced - decays geometrically, which keeps its derivative RELATIVELY accurate in double
// where the accumulated d P02 of the three-column form would cancel.
// dada [K x nder] = derivative planes of 1 / a per piece (pieces AFTER the hidden states were inserted), davg [M x nder] those of
// the average coalescence times.  Also returns the generators themselves (for transition_expand) through `gout`.
inline TransitionGenJac transition_generators_jac(const RateFunctionT<double> &eta, double rho, const std::vector<double> &avg,
                                                  const double *dada, const double *davg, int nder,
                                                  TransitionGenerators<double> *gout = nullptr) {
    using namespace detail;
    const std::vector<double> &ts = eta.ts, &ada = eta.ada;
    const std::vector<int> &hsi = eta.hs_indices;
    const int Mh = (int)eta.hidden_states.size(), M = Mh - 1, nts = (int)ts.size();
    TransitionGenerators<double> g;
    g.Mh = Mh;
    TransitionGenJac out;
    out.M = M; out.nder = nder;
    // ---- the wide chain: factors E_i with their c_eta-derivatives G_i, row 0 of the running product, planes of u = (P00, P01) ----
    std::vector<M3T<ld>> E(nts, m3_identity<ld>()), G(nts);
    for (auto &m : G) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) m.m[a][b] = 0.0L;
    std::vector<ld> p((size_t)nts * 3, 0.0L);
    std::vector<double> du((size_t)nts * 2 * std::max(1, nder), 0.0);
    for (int i = 0; i <= hsi[0]; ++i) p[(size_t)i * 3 + 0] = 1.0L;
    for (int i = hsi[0] + 1; i < nts; ++i) {
        const ld *pp = &p[(size_t)(i - 1) * 3];
        ld *pc = &p[(size_t)i * 3];
        const double *dup = &du[(size_t)(i - 1) * 2 * nder];
        double *duc = &du[(size_t)i * 2 * nder];
        if (std::isinf(ts[i])) {
            // (the reference push_back()s instead of assigning: the factor stays the identity; m3_mul with it, as the generic routine)
            for (int c = 0; c < 3; ++c) { ld sacc = 0.0L; for (int k = 0; k < 3; ++k) sacc += pp[k] * (ld)(k == c); pc[c] = sacc; }
            for (int x = 0; x < 2 * nder; ++x) duc[x] = dup[x];
            continue;
        }
        const double delta = ts[i] - ts[i - 1];
        M3T<ld> Gr;
        matrix_exp_partials((ld)delta * (ld)rho, (ld)ada[i - 1] * (ld)delta, E[i], Gr, G[i], nder > 0);
        double gv[2];
        for (int c = 0; c < 3; ++c) {
            ld sacc = 0.0L;
            for (int k = 0; k < 3; ++k) sacc += pp[k] * E[i].m[k][c];
            pc[c] = sacc;
        }
        if (nder) {
            for (int c = 0; c < 2; ++c) gv[c] = (double)(pp[0] * G[i].m[0][c] + pp[1] * G[i].m[1][c]) * delta;
            const double *da = dada + (size_t)(i - 1) * nder;
            for (int c = 0; c < 2; ++c) {
                const double e0 = (double)E[i].m[0][c], e1 = (double)E[i].m[1][c], gd = gv[c];
                double *o = duc + (size_t)c * nder;
                const double *d0 = dup, *d1 = dup + nder;
                for (int d = 0; d < nder; ++d) o[d] = d0[d] * e0 + d1[d] * e1 + gd * da[d];
            }
        }
    }
    g.expm_diff.assign(std::max(0, M - 1), 0.0);
    for (int k = 1; k < M; ++k) g.expm_diff[k - 1] = (double)p[(size_t)hsi[k] * 3 + 2] - (double)p[(size_t)hsi[k - 1] * 3 + 2];
    out.ed = g.expm_diff;
    out.ded.assign((size_t)std::max(0, M - 1) * nder, 0.0);
    for (int k = 1; k < M; ++k) {
        const double *a = &du[(size_t)hsi[k - 1] * 2 * nder], *b = &du[(size_t)hsi[k] * 2 * nder];
        for (int d = 0; d < nder; ++d) out.ded[(size_t)(k - 1) * nder + d] = (a[d] + a[nder + d]) - (b[d] + b[nder + d]);
    }
    // ---- cumulative hazards ----
    g.inc_k.assign(Mh, 0.0); g.Ek.assign(Mh, 1.0); g.qk.assign(Mh, 1.0); g.p_float.assign(Mh, 0.0);
    std::vector<double> Ccum(Mh, 0.0);
    std::vector<double> dC((size_t)Mh * nder, 0.0), dEk((size_t)Mh * nder, 0.0), dqk((size_t)Mh * nder, 0.0), dinc(std::max(1, nder));
    for (int k = 1; k < Mh; ++k) {
        double inc = 0.0;
        for (int jj = hsi[k - 1]; jj < hsi[k]; ++jj) inc += ada[jj] * (ts[jj + 1] - ts[jj]);
        g.inc_k[k] = inc;
        Ccum[k] = Ccum[k - 1] + inc;
        g.Ek[k] = std::exp(-Ccum[k]);
        const bool inf = std::isinf(inc);
        g.qk[k] = inf ? 1.0 : -std::expm1(-inc);
        if (!nder) continue;
        for (int d = 0; d < nder; ++d) dinc[d] = 0.0;
        if (!inf)
            for (int jj = hsi[k - 1]; jj < hsi[k]; ++jj) {
                const double w = ts[jj + 1] - ts[jj];
                for (int d = 0; d < nder; ++d) dinc[d] += dada[(size_t)jj * nder + d] * w;
            }
        const double em = inf ? 0.0 : 1.0 - g.qk[k];           // exp(-inc)
        for (int d = 0; d < nder; ++d) {
            dC[(size_t)k * nder + d] = dC[(size_t)(k - 1) * nder + d] + dinc[d];
            dEk[(size_t)k * nder + d] = inf ? 0.0 : -dC[(size_t)k * nder + d] * g.Ek[k];
            dqk[(size_t)k * nder + d] = inf ? 0.0 : dinc[d] * em;
        }
    }
    out.pf.assign(M, 0.0); out.W.assign(M, 0.0);
    out.dpf.assign((size_t)M * nder, 0.0); out.dW.assign((size_t)M * nder, 0.0);
    for (int c = 1; c < M; ++c) {
        out.W[c] = g.Ek[c] * g.qk[c + 1];
        for (int d = 0; d < nder; ++d)
            out.dW[(size_t)c * nder + d] = dEk[(size_t)c * nder + d] * g.qk[c + 1] + g.Ek[c] * dqk[(size_t)(c + 1) * nder + d];
    }
    // ---- one partial exponential per row ----
    std::vector<double> dA((size_t)9 * std::max(1, nder)), dAn((size_t)9 * std::max(1, nder)), dce(std::max(1, nder)), dcr(std::max(1, nder)),
        dRj(std::max(1, nder));
    auto narrow = [](const M3T<ld> &m) { M3T<double> r; for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) r.m[a][b] = (double)m.m[a][b]; return r; };
    for (int j = 1; j < Mh; ++j) {
        const double rct = avg[j - 1];
        const int rct_ip = (int)(std::upper_bound(ts.begin(), ts.end(), rct) - ts.begin()) - 1;
        const double *dav = nder ? davg + (size_t)(j - 1) * nder : nullptr;
        // A_pre = prod narrow(E_ell), ell = hsi[j-1] .. rct_ip - 1, with its planes
        M3T<double> A = m3_identity<double>();
        const bool pre = hsi[j - 1] < rct_ip;            // (usually not: the average time lies in the state's first piece)
        if (pre) std::fill(dA.begin(), dA.end(), 0.0);
        for (int ell = hsi[j - 1]; ell < rct_ip; ++ell) {
            const M3T<double> En = narrow(E[ell]), Gn = narrow(G[ell]);
            const double dl = ell >= 1 ? ts[ell] - ts[ell - 1] : 0.0;
            const double *da = (ell >= 1 && nder) ? dada + (size_t)(ell - 1) * nder : nullptr;
            const M3T<double> AG = m3_mul(A, Gn);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double *o = &dAn[(size_t)(a * 3 + b) * nder];
                    for (int d = 0; d < nder; ++d) {
                        double sacc = 0.0;
                        for (int k = 0; k < 3; ++k) sacc += dA[(size_t)(a * 3 + k) * nder + d] * En.m[k][b];
                        o[d] = sacc + (da ? AG.m[a][b] * dl * da[d] : 0.0);
                    }
                }
            dA.swap(dAn);
            A = m3_mul(A, En);
        }
        const double delta = rct - ts[rct_ip];
        const double c_eta = ada[rct_ip] * delta, c_rho = delta * rho;
        M3T<double> X, Xr, Xe;
        {
            M3T<ld> Xl, Xrl, Xel;
            matrix_exp_partials((ld)c_rho, (ld)c_eta, Xl, Xrl, Xel, nder > 0);
            X = narrow(Xl);
            if (nder) { Xr = narrow(Xrl); Xe = narrow(Xel); }
        }
        A = m3_mul(A, X);                                  // (A_pre X: the generic routine's A)
        M3T<double> Pn;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) Pn.m[a][b] = a == 0 ? (double)p[(size_t)hsi[j - 1] * 3 + b] : (a == b ? 1.0 : 0.0);
        // (rows 1 and 2 of the narrowed running product do not enter B(0, 1); row 0 is formed as m3_mul does)
        double B01 = 0.0;
        for (int k = 0; k < 3; ++k) B01 += Pn.m[0][k] * A.m[k][1];
        double Rj = c_eta;
        Rj += ada[rct_ip] * (ts[rct_ip + 1] - rct);
        for (int jj = rct_ip + 2; jj < hsi[j]; ++jj) Rj += ada[jj] * (ts[jj + 1] - ts[jj]);
        const double eR = std::exp(-Rj);
        g.p_float[j] = B01 * eR;
        if (j >= M) continue;                               // the last row has no entry above the diagonal
        if (!(g.Ek[j] > 1e-250)) { out.ok = false; continue; }
        out.pf[j - 1] = g.p_float[j] / g.Ek[j];
        if (!nder) continue;
        for (int d = 0; d < nder; ++d) {
            dce[d] = dada[(size_t)rct_ip * nder + d] * delta + ada[rct_ip] * dav[d];
            dcr[d] = dav[d] * rho;
            dRj[d] = dce[d] + dada[(size_t)rct_ip * nder + d] * (ts[rct_ip + 1] - rct) - ada[rct_ip] * dav[d];
        }
        for (int jj = rct_ip + 2; jj < hsi[j]; ++jj) {
            const double w = ts[jj + 1] - ts[jj];
            for (int d = 0; d < nder; ++d) dRj[d] += dada[(size_t)jj * nder + d] * w;
        }
        // column 1 of A_pre X_r and A_pre X_e (A currently holds A_pre X; A_pre = A X^-1 is not needed: recompute from the factors)
        M3T<double> Ap = m3_identity<double>();
        if (pre) for (int ell = hsi[j - 1]; ell < rct_ip; ++ell) Ap = m3_mul(Ap, narrow(E[ell]));
        double AXr1[3], AXe1[3];
        for (int k = 0; k < 3; ++k) {
            AXr1[k] = AXe1[k] = 0.0;
            for (int m = 0; m < 3; ++m) { AXr1[k] += Ap.m[k][m] * Xr.m[m][1]; AXe1[k] += Ap.m[k][m] * Xe.m[m][1]; }
        }
        const double *dul = &du[(size_t)hsi[j - 1] * 2 * nder];
        const double pfv = out.pf[j - 1], Ekj = g.Ek[j];
        for (int d = 0; d < nder; ++d) {
            double dB = 0.0;
            for (int k = 0; k < 3; ++k) {
                double dAX = AXr1[k] * dcr[d] + AXe1[k] * dce[d];
                if (pre) for (int m = 0; m < 3; ++m) dAX += dA[(size_t)(k * 3 + m) * nder + d] * X.m[m][1];
                dB += Pn.m[0][k] * dAX;
                if (k < 2) dB += dul[(size_t)k * nder + d] * A.m[k][1];      // (A(2, 1) = 0: the third column's plane never enters)
            }
            const double dpfl = dB * eR - B01 * eR * dRj[d];
            out.dpf[(size_t)(j - 1) * nder + d] = (dpfl - pfv * dEk[(size_t)j * nder + d]) / Ekj;
        }
    }
    if (gout) *gout = g;
    return out;
}

// dT [M*M x nder] from the generator planes: the derivative of transition_expand entry by entry (floored entries carry no
// derivative; the diagonal closes the row over the UNfloored entries).  O(M^2 nder): only the Jacobian getter pays it.
inline void transition_expand_jac(const TransitionGenJac &g, std::vector<double> &dT) {
    const int M = g.M, nder = g.nder;
    const double beta = 1e-5;
    dT.assign((size_t)M * M * nder, 0.0);
    std::vector<double> ds(nder);
    for (int i = 0; i < M; ++i) {
        std::fill(ds.begin(), ds.end(), 0.0);
        double sm = 0.0;
        for (int c = 0; c < M; ++c) {
            if (c == i) continue;
            const double x = c < i ? g.ed[c] : g.pf[i] * g.W[c];
            sm += x;
            double *o = &dT[((size_t)i * M + c) * nder];
            for (int d = 0; d < nder; ++d) {
                const double dx = c < i ? g.ded[(size_t)c * nder + d] : g.dpf[(size_t)i * nder + d] * g.W[c] + g.pf[i] * g.dW[(size_t)c * nder + d];
                ds[d] += dx;
                o[d] = x < 1e-20 ? 0.0 : dx * (1 - beta);
            }
        }
        const double diag = 1.0 - sm;
        double *o = &dT[((size_t)i * M + i) * nder];
        for (int d = 0; d < nder; ++d) o[d] = diag < 1e-20 ? 0.0 : -ds[d] * (1 - beta);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// A10: conditioned SFS (conditioned_sfs.cpp:13-148)
// ---------------------------------------------------------------------------------------------------------------
template <typename S>
inline S dcs_sorted(std::vector<S> &v) {
    std::sort(v.begin(), v.end(), [](const S &x, const S &y) { return std::fabs((double)sval(x)) > std::fabs((double)sval(y)); });
    if (v.empty()) return S(0.0);
    S s = v[0], c(0.0);
    for (size_t i = 1; i < v.size(); ++i) {
        const S y = c + v[i];
        const S u = v[i] - (y - c);
        const S t = y + s;
        const S w = y - (t - s);
        const S z = u + w;
        s = t + z;
        c = z - (s - t);
    }
    return s;
}

// raw CSFS per hidden state: out[m] is 3 x (n+1) row-major.  Literal evaluation, term by term as the reference
// writes it (O(pieces^2 n^2) transcendental calls); kept as the in-tree cross-check of conditioned_sfs() below
// (tests/test_prep.py) and selected at run time by SMCPP_CSFS_DIRECT=1 / smcpp_host_set_csfs_direct(1).
template <typename S>
inline std::vector<std::vector<S>> conditioned_sfs_direct(const RateFunctionT<S> &eta, const CsfsTables &tb, bool below_only = false) {
    const int n = tb.n;
    const int M = (int)eta.hidden_states.size() - 1;
    const int nd = dual_nder();
    std::vector<std::vector<S>> csfs(M, std::vector<S>((size_t)3 * (n + 1), S(0.0)));
    // ---- above ----
    if (n >= 1 && !below_only) {
        std::vector<std::vector<S>> C_above(M, std::vector<S>((size_t)(n + 1) * n, S(0.0)));
#pragma omp parallel for schedule(dynamic)
        for (int j = 2; j < n + 3; ++j) { DualScope sc(nd); eta.tjj_double_integral_above(n, j, C_above); }
#pragma omp parallel for schedule(dynamic)
        for (int m = 0; m < M; ++m) {
            DualScope sc(nd);
            const std::vector<S> &Ca = C_above[m];
            std::vector<S> tmp0(n + 1, S(0.0)), tmp2(n + 1, S(0.0)), v(n, S(0.0));
            for (int j = 0; j < n + 1; ++j) {
                for (int i = 0; i < n; ++i) v[i] = Ca[(size_t)j * n + i] * tb.X0(i, j);            // C0(i,j) = C(j,i)
                std::vector<S> w(v);
                tmp0[j] = dcs_sorted(w);
                for (int i = 0; i < n; ++i) v[i] = Ca[(size_t)(n - j) * n + i] * tb.X2(i, j);      // C2(i,j) = C(n-j,i)
                w = v;
                tmp2[j] = dcs_sorted(w);
            }
            for (int b = 0; b < n; ++b) {
                S s0(0.0), s2(0.0);
                for (int j = 0; j < n + 1; ++j) { s0 += tmp0[j] * tb.Uinv_mp0(j, b); s2 += tmp2[j] * tb.Uinv_mp2(j, b); }
                csfs[m][0 * (n + 1) + 1 + b] += s0;
                csfs[m][2 * (n + 1) + b] += s2;
            }
        }
    }
    // ---- below ----
    std::vector<S> tjj_below((size_t)M * (n + 1), S(0.0));
#pragma omp parallel for schedule(dynamic)
    for (int m = 0; m < M; ++m) { DualScope sc(nd); eta.tjj_double_integral_below(n, m, tjj_below); }
    for (int m = 0; m < M; ++m) {
        for (int b = 0; b < n; ++b) {
            S s(0.0);
            for (int j = 0; j < n + 1; ++j) s += tjj_below[(size_t)m * (n + 1) + j] * tb.M0(j, b);
            csfs[m][0 * (n + 1) + 1 + b] += s;
        }
        for (int b = 0; b < n + 1; ++b) {
            S s(0.0);
            for (int j = 0; j < n + 1; ++j) s += tjj_below[(size_t)m * (n + 1) + j] * tb.M1(j, b);
            csfs[m][1 * (n + 1) + b] += s;
        }
    }
    return csfs;
}

// The same quantity in O(pieces n^2) transcendental calls.  The reference's inner sums over the pieces after (above)
// or before (below) piece m,
//     sum_k single_integral(rate, piece k, log_coef) = sum_k exp(-rate R_k + log_coef) g(rate, k),
//     g(rate, k) = -expm1(-rate ada_k (t_{k+1} - t_k)) / (ada_k rate)        (no expm1 factor on the infinite last piece)
// depend on (m, lam) only through the factor exp(log_coef), so they are
//     above:  exp(-rate R_{m+1} + log_coef) * Ssuf[rate][m],   Ssuf[rate][m-1] = g(rate, m) + exp(-rate (R_{m+1} - R_m)) Ssuf[rate][m]
//     below:  exp(log_coef)                 * Ppre[rate][m],   Ppre[rate][m+1] = Ppre[rate][m] + exp(-rate R_m) g(rate, m)
// (all terms are non-negative: no cancellation; the leading factor is formed with the reference's own operation order
// for its first term k = m+1).  Everything that depends on one index only (exp(-l1 R_m + log_coef), expm1(-l1 adadiff),
// exp(-rate adadiff)) is hoisted out of the (lam, rate) double loop with its arithmetic unchanged.  One OpenMP task per
// hidden state does the integrals, the compensated contractions with X0 / X2 and the Moran back-transformation of
// that state.
// 0 = factored evaluation (default), 1 = literal evaluation; initialised from SMCPP_CSFS_DIRECT, set over the C ABI by
// smcpp_host_set_csfs_direct (test hook)
inline int &csfs_direct_flag() {
    static int flag = opt().has(smcpp_opt::O_CSFS_DIRECT) ? 1 : 0;
    return flag;
}

template <typename S>
struct CsfsPieceTables {
    int K = 0, n = 0;
    std::vector<S> Ssuf;     // [n][K]       rate = C(j,2),   j = 2..n+1 : suffix sums of the "above" integrals
    std::vector<S> Ppre;     // [n+1][K+1]   rate = C(j,2)-1, j = 2..n+2 : prefix sums of the "below" integrals
};

// Fills t (sized by the caller) with orphaned work-sharing loops: call from every thread of a parallel region (each with
// its DualScope set); returns after the barrier that ends the second loop.
template <typename S>
inline void csfs_piece_tables_ws(const RateFunctionT<S> &eta, int n, bool above, CsfsPieceTables<S> &t) {
    const int K = eta.K;
    const std::vector<double> &ts = eta.ts;
    if (above) {
#pragma omp for schedule(static) nowait
        for (int jr = 0; jr < n; ++jr) {
            const double rate = (double)RateFunctionT<S>::nC2(jr + 2);
            S *Sj = &t.Ssuf[(size_t)jr * K];
            Sj[K - 1] = S(0.0);
            for (int m = K - 1; m >= 1; --m) {
                const S &ad = eta.ada[m];
                if (ts[m + 1] < INFINITY) {
                    const S em = m_expm1(-rate * ad * (ts[m + 1] - ts[m]));
                    Sj[m - 1] = -em / (ad * rate) + (1.0 + em) * Sj[m];
                } else Sj[m - 1] = 1.0 / (ad * rate);
            }
        }
    }
#pragma omp for schedule(static)
    for (int jr = 0; jr < n + 1; ++jr) {
        const long ratel = RateFunctionT<S>::nC2(jr + 2) - 1;
        const double rate = (double)ratel;
        S *Pj = &t.Ppre[(size_t)jr * (K + 1)];
        Pj[0] = S(0.0);
        for (int m = 0; m < K; ++m) {
            if (ratel == 0) { Pj[m + 1] = S(ts[m + 1]); continue; }      // sum_k (t_{k+1} - t_k) = t_{m+1}
            const S &ad = eta.ada[m];
            S g = m_exp(-rate * eta.Rrng[m]);
            if (ts[m + 1] < INFINITY) g *= -m_expm1(-rate * ad * (ts[m + 1] - ts[m]));
            g /= ad * rate;
            Pj[m + 1] = Pj[m] + g;
        }
    }
}

// Accurate sum of v[0..cnt).  The reference sorts the terms by decreasing magnitude and applies doubly-compensated
// summation (common.h:27-46, conditioned_sfs.cpp:63-67), whose result is within 2 ulp of the exact sum; a cascaded
// TwoSum accumulation (error eps |sum| + cnt eps^2 sum |v|) gives that quality without the sort, which was 85 % of the
// time of this function once the integrals were factored.  Derivative parts are plain sums (they are linear in the terms).
inline double accurate_sum(const double *v, int cnt) {
    double hi = 0.0, lo = 0.0;
    for (int i = 0; i < cnt; ++i) {
        const double x = v[i], t = hi + x, z = t - hi;
        lo += (hi - (t - z)) + (x - z);
        hi = t;
    }
    return hi + lo;
}
template <typename F>
inline Dual<F> accurate_sum(const Dual<F> *v, int cnt) {
    Dual<F> r;
    F hi = 0, lo = 0;
    for (int i = 0; i < cnt; ++i) {
        const F x = v[i].v, t = hi + x, z = t - hi;
        lo += (hi - (t - z)) + (x - z);
        hi = t;
        SMCPP_DUAL_LOOP r.d[i_] += v[i].d[i_];
    }
    r.v = hi + lo;
    return r;
}

// One batch of hidden states of one rate function: the shared state of conditioned_sfs_team().
template <typename S>
struct CsfsJob {
    const RateFunctionT<S> *eta = nullptr;
    const CsfsTables *tb = nullptr;
    bool below_only = false;
    CsfsPieceTables<S> pt;
    std::vector<std::vector<S>> csfs;
    double tmark[64][4] = {};
    std::chrono::steady_clock::time_point tbase;
    std::exception_ptr side_err;
    bool factored() const {
        if (csfs_direct_flag() != 0) return false;
        for (const S &x : eta->ada) if (sval(x) == 0) return false;               // ada == 0: the factored sums divide by it
        return true;
    }
    void init(const RateFunctionT<S> &e, const CsfsTables &t, bool below) {
        eta = &e; tb = &t; below_only = below;
        const int n = t.n, M = (int)e.hidden_states.size() - 1, K = e.K;
        const bool above = n >= 1 && !below;
        pt.K = K; pt.n = n;
        if (above) pt.Ssuf.assign((size_t)n * K, S(0.0));
        pt.Ppre.assign((size_t)(n + 1) * (K + 1), S(0.0));
        csfs.assign(M, std::vector<S>((size_t)3 * (n + 1), S(0.0)));
        tbase = std::chrono::steady_clock::now();
    }
};

// The work of conditioned_sfs() on a job prepared with CsfsJob::init(): called by EVERY thread of a team (each with its DualScope
// set) - the piece tables and the hidden states are shared out by orphaned work-sharing loops, the call returns after the barrier
// that ends the last of them.  (The joint CSFS runs its batches through this inside its own parallel region: waking libomp's
// sleeping workers for a region of its own costs more than a batch.)
template <typename S>
inline void conditioned_sfs_team(CsfsJob<S> &job, const std::function<void()> *side = nullptr) {
    typedef RateFunctionT<S> RF;
    const RateFunctionT<S> &eta = *job.eta;
    const CsfsTables &tb = *job.tb;
    const int n = tb.n;
    const int M = (int)eta.hidden_states.size() - 1;
    const int K = eta.K;
    const std::vector<double> &ts = eta.ts;
    const std::vector<S> &ada = eta.ada, &Rrng = eta.Rrng;
    const std::vector<int> &hsi = eta.hs_indices;
    const bool above = n >= 1 && !job.below_only;
    CsfsPieceTables<S> &pt = job.pt;
    std::vector<std::vector<S>> &csfs = job.csfs;
    std::exception_ptr &side_err = job.side_err;
    const bool tm = opt().has(smcpp_opt::O_HOST_TIMING);
    double (&tmark)[64][4] = job.tmark;
    const auto tbase = job.tbase;
    auto now_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tbase).count(); };
        const int tid_ = omp_get_thread_num();
        if (tm && tid_ < 64) tmark[tid_][0] = now_us();
        csfs_piece_tables_ws<S>(eta, n, above, pt);       // orphaned work-sharing loops, barrier at the end
        if (tm && tid_ < 64) tmark[tid_][1] = now_us();
#pragma omp single nowait
        {
            if (side) {
                try { (*side)(); } catch (...) { side_err = std::current_exception(); }
            }
        }
        if (tm && tid_ < 64) tmark[tid_][2] = now_us();
        std::vector<S> Ca(above ? (size_t)(n + 1) * n : 0), A(n + 1), A1(n + 1), B(n + 1), El(n + 1), ert(n), e1(n), tmp0(n + 1), tmp2(n + 1), v(n),
            below(n + 1);
#pragma omp for schedule(dynamic)
        for (int h = 0; h < M; ++h) {
            const S Rh = Rrng[hsi[h]], Rh1 = Rrng[hsi[h + 1]];
            S log_denom = -Rh;
            if (sval(Rh1) != INFINITY) log_denom += m_log(-m_expm1(-(Rh1 - Rh)));
            S *out = csfs[h].data();
            // ---- above (tjj_double_integral_above for every jj, piecewise_constant_rate_function.cpp:214-299) ----
            if (above) {
                for (S &x : Ca) x = S(0.0);
                const S log_coef0 = -log_denom;
                for (int m = hsi[h]; m < hsi[h + 1]; ++m) {
                    const S &ad = ada[m];
                    const bool fin = ts[m + 1] < INFINITY;
                    const S adadiff = ad * (ts[m + 1] - ts[m]);
                    const S Rm = Rrng[m], Rm1 = Rrng[m + 1];
                    const S dR = Rm1 - Rm;
                    for (int jl = 0; jl <= n; ++jl) {
                        const double l1 = (double)(RF::nC2(jl + 2));               // lam + 1
                        A[jl] = m_exp(-l1 * Rm + log_coef0);
                        if (m + 1 < K) A1[jl] = m_exp(-l1 * Rm1 + log_coef0);
                        if (fin) { B[jl] = m_expm1(-l1 * adadiff); El[jl] = m_exp(-l1 * adadiff); }
                    }
                    if (fin) for (int jr = 0; jr < n; ++jr) {
                        ert[jr] = m_exp(-(double)RF::nC2(jr + 2) * adadiff);
                        e1[jr] = m_exp(-(double)RF::nC2(jr + 2) * dR);
                    }
                    for (int jl = 0; jl <= n; ++jl) {
                        const long l1l = RF::nC2(jl + 2);
                        const double l1 = (double)l1l;
                        for (int jr = 0; jr < n; ++jr) {
                            const long ratel = RF::nC2(jr + 2);
                            const double rt = (double)ratel;
                            S &tgt = Ca[(size_t)jl * n + jr];
                            // _double_integral_above_helper on the piece itself (rate >= 1 here)
                            if (l1l == ratel) {
                                if (!fin) tgt += A[jl] / rt / rt / ad;
                                else tgt += A[jl] * (1.0 - ert[jr] * (1.0 + rt * adadiff)) / rt / rt / ad;
                            } else if (!fin) tgt += A[jl] / l1 / rt / ad;
                            else if (ratel < l1l)
                                tgt += -A[jl] * (B[jl] / l1 + (ert[jr] * -m_expm1(-(l1 - rt) * adadiff) / (l1 - rt))) / rt / ad;
                            else
                                tgt += -A[jl] * (B[jl] / l1 + (El[jl] * m_expm1(-(rt - l1) * adadiff) / (l1 - rt))) / rt / ad;
                            // the later pieces k > m
                            if (m + 1 >= K) continue;
                            // exp(-rate R_{m+1} + log_coef) with log_coef = -log_denom - rp R_{m+1} or - rp R_m (rp = l1 - rate) is
                            // exp(-log_denom - l1 R_{m+1}) = A1[jl]  or  exp(-log_denom - l1 R_m) exp(-rate (R_{m+1} - R_m)) =
                            // A[jl] e1[jr]: two hoisted tables instead of one exp per (lam, rate) pair (and without the
                            // cancellation between -rate R and -rp R inside the exponent)
                            S coef(0.0), fac(0.0);
                            const long rp = l1l - ratel;
                            const double rpd = (double)rp;
                            if (rp == 0) { fac = dR; coef = A1[jl]; }
                            else if (rp < 0) {
                                if (-rpd * sval(dR) > 20) { coef = A1[jl]; fac = S(-1.0 / rpd); }
                                else { coef = A[jl] * e1[jr]; fac = -m_expm1(-rpd * dR) / rpd; }
                            } else {
                                if (-rpd * sval(Rm - Rm1) > 20) { coef = A[jl] * e1[jr]; fac = S(1.0 / rpd); }
                                else { coef = A1[jl]; fac = m_expm1(-rpd * (Rm - Rm1)) / rpd; }
                            }
                            tgt += coef * pt.Ssuf[(size_t)jr * K + m] * fac;
                        }
                    }
                }
                // ---- contractions (conditioned_sfs.cpp:42-83) ----
                for (int j = 0; j < n + 1; ++j) {
                    for (int i = 0; i < n; ++i) v[i] = Ca[(size_t)j * n + i] * tb.X0(i, j);              // C0(i,j) = C(j,i)
                    tmp0[j] = accurate_sum(v.data(), n);
                    for (int i = 0; i < n; ++i) v[i] = Ca[(size_t)(n - j) * n + i] * tb.X2(i, j);        // C2(i,j) = C(n-j,i)
                    tmp2[j] = accurate_sum(v.data(), n);
                }
                for (int b = 0; b < n; ++b) {
                    S s0(0.0), s2(0.0);
                    for (int j = 0; j < n + 1; ++j) { s0 += tmp0[j] * tb.Uinv_mp0(j, b); s2 += tmp2[j] * tb.Uinv_mp2(j, b); }
                    out[0 * (n + 1) + 1 + b] += s0;
                    out[2 * (n + 1) + b] += s2;
                }
            }
            // ---- below (tjj_double_integral_below, piecewise_constant_rate_function.cpp:302-334) ----
            for (S &x : below) x = S(0.0);
            for (int m = hsi[h]; m < hsi[h + 1]; ++m) {
                const S Rm = Rrng[m], Rm1 = Rrng[m + 1];
                const S c = -Rm - log_denom;
                S fac(1.0);
                if (m < K - 1) fac = -m_expm1(-(Rm1 - Rm));
                const S ec = m > 0 ? m_exp(c) : S(0.0);
                for (int j = 2; j < n + 3; ++j) {
                    const long rate = RF::nC2(j) - 1;
                    S val = RF::below_helper(rate, ts[m], ts[m + 1], ada[m], Rm, log_denom);
                    if (m > 0) val += fac * (ec * pt.Ppre[(size_t)(j - 2) * (K + 1) + m]);
                    below[j - 2] += val;
                }
            }
            for (int b = 0; b < n; ++b) {
                S s(0.0);
                for (int j = 0; j < n + 1; ++j) s += below[j] * tb.M0(j, b);
                out[0 * (n + 1) + 1 + b] += s;
            }
            for (int b = 0; b < n + 1; ++b) {
                S s(0.0);
                for (int j = 0; j < n + 1; ++j) s += below[j] * tb.M1(j, b);
                out[1 * (n + 1) + b] += s;
            }
            if (tm && tid_ < 64) tmark[tid_][3] = now_us();      // nowait loop below: last state finished by this thread
        }
}

// `side` (optional): an independent serial job of the caller (the transition matrix, 0.06 ms) that one thread of this
// function's parallel region runs while the others already work on hidden states.
template <typename S>
inline std::vector<std::vector<S>> conditioned_sfs(const RateFunctionT<S> &eta, const CsfsTables &tb, bool below_only = false,
                                                   const std::function<void()> *side = nullptr) {
    CsfsJob<S> job;
    job.eta = &eta;
    if (!job.factored()) {
        if (side) (*side)();
        return conditioned_sfs_direct<S>(eta, tb, below_only);
    }
    job.init(eta, tb, below_only);
    const int nd = dual_nder();
    const bool tm = opt().has(smcpp_opt::O_HOST_TIMING);
    // one parallel region for everything: the piece tables (a few microseconds, static), then - without a barrier in
    // between - the caller's side job on whichever thread gets there first and the hidden states on all of them
    // (called per hidden state from inside another parallel loop: no nested team there)
#pragma omp parallel if (!omp_in_parallel())
    {
        DualScope sc(nd);
        conditioned_sfs_team<S>(job, side);
    }
    if (tm && !omp_in_parallel()) {
        const double tend = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - job.tbase).count();
        fprintf(stderr, "[csfs] region %.1f us; per thread (enter, tables done, side done, last state):", tend);
        for (int t = 0; t < std::min(64, omp_get_max_threads()); ++t)
            fprintf(stderr, " [%.0f %.0f %.0f %.0f]", job.tmark[t][0], job.tmark[t][1], job.tmark[t][2], job.tmark[t][3]);
        fprintf(stderr, "\n");
    }
    if (job.side_err) std::rethrow_exception(job.side_err);
    return std::move(job.csfs);
}

template <typename S>
inline void incorporate_theta(std::vector<std::vector<S>> &csfs, double theta) {
    if (theta <= 0) throw std::runtime_error("mutation rate theta <= 0");
    for (auto &c : csfs) {
        S tauh(0.0);
        for (const S &x : c) tauh += x;
        const S f = -m_expm1(-theta * tauh) / tauh;
        for (S &x : c) x *= f;
        S tot(0.0);
        for (const S &x : c) tot += x;
        c[0] = 1.0 - tot;
        for (S &x : c) if (sval(x) < 1e-10) x = S(1e-10);
        for (const S &x : c)
            if (sval(x) < 0 || sval(x) > 1 || std::isnan((double)sval(x))) throw std::runtime_error("csfs is not a probability distribution");
    }
}

// ---------------------------------------------------------------------------------------------------------------
// A6 + A7: one-population preparation
// ---------------------------------------------------------------------------------------------------------------
// A7: pi (inference_manager.cpp:56-69)
template <typename S>
inline void initial_distribution(const RateFunctionT<S> &eta, std::vector<S> &pi) {
    const std::vector<double> &hs = eta.hidden_states;
    const int M = (int)hs.size() - 1;
    pi.assign(M, S(0.0));
    for (int m = 0; m < M - 1; ++m) pi[m] = m_exp(-eta.R(hs[m])) - m_exp(-eta.R(hs[m + 1]));
    pi[M - 1] = m_exp(-eta.R(hs[M - 1]));
    S ps(0.0);
    for (S &x : pi) { if (sval(x) < 1e-20) x = S(1e-20); ps += x; }
    for (S &x : pi) x /= ps;
}

class OnePopPrep {
public:
    OnePopPrep(int n, const std::vector<double> &hs, double polarization_error)
        : n_(n), hs_(hs), pol_(polarization_error), tables_(csfs_tables(n)) {}
    const CsfsTables &tables() const { return *tables_; }
    int n() const { return n_; }

    // keys: [K][3] (a, b, nb); outputs pi [M], T [M*M] row-major, E [K*M]
    template <typename S>
    void compute_t(const ModelParamsT<S> &mp, double theta, double rho, double alpha, const std::vector<int> &keys,
                   int K, std::vector<S> &pi, std::vector<S> &T, std::vector<S> &E, std::vector<S> *emission_out = nullptr) {
        const bool tm = opt().has(smcpp_opt::O_HOST_TIMING);
        auto clk = [] { return std::chrono::steady_clock::now(); };
        auto t0 = clk();
        RateFunctionT<S> eta(mp, hs_);
        const int M = (int)hs_.size() - 1;
        initial_distribution<S>(eta, pi);
        auto t1 = clk();
        auto t2 = t1;
        // the transition matrix (serial, long double) rides along in the conditioned SFS's parallel region
        const std::function<void()> side = [&] { T = compute_transition<S>(eta, rho); };
        std::vector<std::vector<S>> sfs = conditioned_sfs<S>(eta, *tables_, false, &side);
        auto t3 = clk();
        incorporate_theta<S>(sfs, theta);
        if (emission_out) {                      // InferenceManager::emission: the table per state, flattened row-major
            emission_out->clear();
            for (const auto &c : sfs) emission_out->insert(emission_out->end(), c.begin(), c.end());
        }
        const std::vector<S> avg_ct = eta.average_coal_times();
        emission_probs<S>(sfs, avg_ct, theta, alpha, keys, K, E);
        if (tm) {
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "[prep] rate+pi %.3f ms, (transition inside csfs: %.3f) csfs+transition %.3f ms, theta+emission %.3f ms\n", ms(t0, t1),
                    ms(t1, t2), ms(t2, t3), ms(t3, clk()));
        }
    }

    void compute(const ModelParams &mp, double theta, double rho, double alpha, const std::vector<int> &keys, int K,
                 std::vector<double> &pi, std::vector<double> &T, std::vector<double> &E,
                 std::vector<double> *emission = nullptr) {
        ModelParamsT<double> p;
        p.a = mp.a; p.s = mp.s;
        compute_t<double>(p, theta, rho, alpha, keys, K, pi, T, E, emission);
    }

    // values + Jacobians: da [Kp x nder] seeds of the piece sizes; outputs d* are [size x nder] row-major
    void compute_with_jacobian(const ModelParams &mp, const std::vector<double> &da, int nder, double theta, double rho,
                               double alpha, const std::vector<int> &keys, int K, std::vector<double> &pi,
                               std::vector<double> &T, std::vector<double> &E, std::vector<double> &dpi,
                               std::vector<double> &dT, std::vector<double> &dE, std::vector<double> *emission = nullptr,
                               std::vector<double> *demission = nullptr) {
        DualScope sc(nder);
        ModelParamsT<dual> p;
        p.s = mp.s;
        p.a.resize(mp.a.size());
        for (size_t k = 0; k < mp.a.size(); ++k) {
            p.a[k] = dual(mp.a[k]);
            for (int d = 0; d < nder; ++d) p.a[k].d[d] = da[k * nder + d];
        }
        std::vector<dual> pd, Td, Ed, emd;
        compute_t<dual>(p, theta, rho, alpha, keys, K, pd, Td, Ed, emission ? &emd : nullptr);
        auto split = [nder](const std::vector<dual> &x, std::vector<double> &v, std::vector<double> &j) {
            v.resize(x.size()); j.resize(x.size() * (size_t)nder);
            for (size_t i = 0; i < x.size(); ++i) { v[i] = x[i].v; for (int d = 0; d < nder; ++d) j[i * nder + d] = x[i].d[d]; }
        };
        split(pd, pi, dpi); split(Td, T, dT); split(Ed, E, dE);
        if (emission) { std::vector<double> tmp; split(emd, *emission, demission ? *demission : tmp); }
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

    // how a key's emission vector is formed: 1 = missing reduced key (all ones), 2 / 3 = reduced key with a even / odd
    // (exp / -expm1 of -2 alpha theta E[T]), 0 = marginalisation bins over the conditioned SFS
    static int key_kind(const Key &bk) {
        const bool reduced = bk[2] == 0, miss = bk[0] == -1;
        if (reduced && (miss || bk[0] >= 0)) return miss ? 1 : 2 + (bk[0] % 2);
        return 0;
    }
    // the bins of a key as (flattened CSFS index, weight) in bins_for's order (hypergeometric weights: lgamma calls, a
    // std::map): they depend on the manager only and are built once
    const std::vector<std::pair<int, double>> &bins_of(const Key &bk) const {
        auto it = bins_cache_.find(bk);
        if (it == bins_cache_.end()) {
            std::vector<std::pair<int, double>> flat;
            for (const auto &p : bins_for(bk)) flat.emplace_back(p.first.first * (n_ + 1) + p.first.second, p.second);
            it = bins_cache_.emplace(bk, std::move(flat)).first;
        }
        return it->second;
    }

    template <typename S>
    void emission_probs(const std::vector<std::vector<S>> &sfs, const std::vector<S> &avg_ct, double theta, double alpha,
                        const std::vector<int> &keys, int K, std::vector<S> &E) const {
        const int M = (int)sfs.size();
        std::vector<S> e2((size_t)M * 2, S(0.0));
        for (int m = 0; m < M; ++m) {
            if (std::isnan((double)sval(avg_ct[m]))) { e2[2 * m] = S(1e-20); e2[2 * m + 1] = S(1e-20); }
            else {
                const S le = -2.0 * alpha * theta * avg_ct[m];
                e2[2 * m] = m_exp(le);
                e2[2 * m + 1] = -m_expm1(le);
            }
        }
        E.assign((size_t)K * M, S(0.0));
        for (int k = 0; k < K; ++k) {
            const Key bk{keys[3 * k], keys[3 * k + 1], keys[3 * k + 2]};
            const int kind = key_kind(bk);
            S *e = &E[(size_t)k * M];
            if (kind != 0) {
                for (int m = 0; m < M; ++m) e[m] = kind == 1 ? S(1.0) : e2[2 * m + (kind - 2)];
            } else {
                for (const auto &p : bins_of(bk))
                    for (int m = 0; m < M; ++m) e[m] += p.second * sfs[m][(size_t)p.first];
            }
            double mx = sval(e[0]), mn = sval(e[0]);
            for (int m = 1; m < M; ++m) { mx = std::max(mx, (double)sval(e[m])); mn = std::min(mn, (double)sval(e[m])); }
            if (mx > 1.0 || mn <= 0.0) throw std::runtime_error("probability vector not in [0, 1]");
        }
    }

private:
    int n_;
    std::vector<double> hs_;
    double pol_;
    std::shared_ptr<const CsfsTables> tables_;
    mutable std::map<Key, std::vector<std::pair<int, double>>> bins_cache_;   // key -> (flattened CSFS index, weight), in bins_for's order
};

}  // namespace smcpp_host
