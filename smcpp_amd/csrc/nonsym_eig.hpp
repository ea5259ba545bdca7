// Host-side dense helpers for the cold preparation step A3:
//   eigendecomposition of the real non-symmetric matrix diag(b_k) * Td^T
//   (reference: Eigen::EigenSolver at src/transition_bundle.cpp:22, consumed by `eigensystem`,
//    include/transition_bundle.h:9-30: P_r = Re P, Pinv_r = Re P^-1, d_r = Re d, scale = max |d|).
//
// Own implementation of the classical EISPACK pipeline (Householder reduction to Hessenberg form `orthes`,
// Francis double-shift QR with back-substitution `hqr2`), followed by unit-norm eigenvectors as EigenSolver
// returns them and an LU inverse in complex arithmetic.  Matrices are row-major std::vector<double>.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <stdexcept>
#include <vector>

namespace smcpp_host {

struct EigenSystem {
    int n = 0;
    std::vector<double> P, Pinv;   // real parts, row-major n x n
    std::vector<double> d;         // real parts of the eigenvalues
    double scale = 0.0;            // max |d| (complex modulus)
    double max_imag = 0.0;
};

namespace detail {

inline void cdiv(double xr, double xi, double yr, double yi, double &cr, double &ci) {
    double r, dd;
    if (std::fabs(yr) > std::fabs(yi)) {
        r = yi / yr; dd = yr + r * yi;
        cr = (xr + r * xi) / dd; ci = (xi - r * xr) / dd;
    } else {
        r = yr / yi; dd = yi + r * yr;
        cr = (r * xr + xi) / dd; ci = (r * xi - xr) / dd;
    }
}

// Householder reduction of H (n x n, row-major, overwritten) to upper Hessenberg; V accumulates the transforms.
inline void orthes(int n, std::vector<double> &H, std::vector<double> &V) {
    auto h = [&](int i, int j) -> double & { return H[(size_t)i * n + j]; };
    auto v = [&](int i, int j) -> double & { return V[(size_t)i * n + j]; };
    std::vector<double> ort(n, 0.0), fcol(n, 0.0);
    const int low = 0, high = n - 1;
    for (int m = low + 1; m <= high - 1; ++m) {
        double sc = 0.0;
        for (int i = m; i <= high; ++i) sc += std::fabs(h(i, m - 1));
        if (sc != 0.0) {
            double hh = 0.0;
            for (int i = high; i >= m; --i) { ort[i] = h(i, m - 1) / sc; hh += ort[i] * ort[i]; }
            double g = std::sqrt(hh);
            if (ort[m] > 0) g = -g;
            hh -= ort[m] * g;
            ort[m] -= g;
            // H <- (I - u u^T / hh) H on columns m..n-1.  Row sweeps with a row of partial sums: every f[j] is still
            // accumulated over i = high .. m in that order (bit-identical to the column-at-a-time form), but the matrix
            // is walked along its rows
            std::fill(fcol.begin() + m, fcol.end(), 0.0);
            for (int i = high; i >= m; --i) {
                const double oi = ort[i];
                const double *hr = &H[(size_t)i * n];
                for (int j = m; j < n; ++j) fcol[j] += oi * hr[j];
            }
            for (int j = m; j < n; ++j) fcol[j] /= hh;
            for (int i = m; i <= high; ++i) {
                const double oi = ort[i];
                double *hr = &H[(size_t)i * n];
                for (int j = m; j < n; ++j) hr[j] -= fcol[j] * oi;
            }
            for (int i = 0; i <= high; ++i) {
                double f = 0.0;
                for (int j = high; j >= m; --j) f += ort[j] * h(i, j);
                f /= hh;
                for (int j = m; j <= high; ++j) h(i, j) -= f * ort[j];
            }
            ort[m] = sc * ort[m];
            h(m, m - 1) = sc * g;
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) v(i, j) = (i == j) ? 1.0 : 0.0;
    for (int m = high - 1; m >= low + 1; --m) {
        if (h(m, m - 1) != 0.0) {
            for (int i = m + 1; i <= high; ++i) ort[i] = h(i, m - 1);
            std::fill(fcol.begin() + m, fcol.begin() + high + 1, 0.0);
            for (int i = m; i <= high; ++i) {                     // g[j] accumulated over i = m .. high as before
                const double oi = ort[i];
                const double *vr = &V[(size_t)i * n];
                for (int j = m; j <= high; ++j) fcol[j] += oi * vr[j];
            }
            const double om = ort[m], hm = h(m, m - 1);
            for (int j = m; j <= high; ++j) fcol[j] = (fcol[j] / om) / hm;
            for (int i = m; i <= high; ++i) {
                const double oi = ort[i];
                double *vr = &V[(size_t)i * n];
                for (int j = m; j <= high; ++j) vr[j] += fcol[j] * oi;
            }
        }
    }
}

// Real Schur form + eigenvectors of the Hessenberg matrix H; on exit wr/wi hold the eigenvalues and the columns
// of V the (real-packed) eigenvectors of the original matrix.
inline void hqr2(int nn, std::vector<double> &H, std::vector<double> &V, std::vector<double> &wr,
                 std::vector<double> &wi) {
    auto h = [&](int i, int j) -> double & { return H[(size_t)i * nn + j]; };
    // the accumulated transform is kept TRANSPOSED while this routine runs: every reflection of a sweep updates three
    // columns of V for all rows - with Vt those are three contiguous, vectorisable rows instead of a stride-n walk
    std::vector<double> Vt((size_t)nn * nn);
    for (int i = 0; i < nn; ++i)
        for (int j = 0; j < nn; ++j) Vt[(size_t)j * nn + i] = V[(size_t)i * nn + j];
    auto v = [&](int i, int j) -> double & { return Vt[(size_t)j * nn + i]; };
    int n = nn - 1;
    const int low = 0, high = nn - 1;
    const double eps = std::pow(2.0, -52.0);
    double exshift = 0.0, p = 0, q = 0, r = 0, s = 0, z = 0, t, w, x, y;
    double norm = 0.0;
    for (int i = 0; i < nn; ++i)
        for (int j = std::max(i - 1, 0); j < nn; ++j) norm += std::fabs(h(i, j));
    int iter = 0, total_iter = 0;
    while (n >= low) {
        int l = n;
        while (l > low) {
            s = std::fabs(h(l - 1, l - 1)) + std::fabs(h(l, l));
            if (s == 0.0) s = norm;
            if (std::fabs(h(l, l - 1)) < eps * s) break;
            --l;
        }
        if (l == n) {                       // one root
            h(n, n) += exshift;
            wr[n] = h(n, n); wi[n] = 0.0;
            --n; iter = 0;
        } else if (l == n - 1) {            // two roots
            w = h(n, n - 1) * h(n - 1, n);
            p = (h(n - 1, n - 1) - h(n, n)) / 2.0;
            q = p * p + w;
            z = std::sqrt(std::fabs(q));
            h(n, n) += exshift;
            h(n - 1, n - 1) += exshift;
            x = h(n, n);
            if (q >= 0) {                   // real pair
                z = (p >= 0) ? p + z : p - z;
                wr[n - 1] = x + z;
                wr[n] = wr[n - 1];
                if (z != 0.0) wr[n] = x - w / z;
                wi[n - 1] = 0.0; wi[n] = 0.0;
                x = h(n, n - 1);
                s = std::fabs(x) + std::fabs(z);
                p = x / s; q = z / s;
                r = std::sqrt(p * p + q * q);
                p /= r; q /= r;
                for (int j = n - 1; j < nn; ++j) {
                    z = h(n - 1, j);
                    h(n - 1, j) = q * z + p * h(n, j);
                    h(n, j) = q * h(n, j) - p * z;
                }
                for (int i = 0; i <= n; ++i) {
                    z = h(i, n - 1);
                    h(i, n - 1) = q * z + p * h(i, n);
                    h(i, n) = q * h(i, n) - p * z;
                }
                for (int i = low; i <= high; ++i) {
                    z = v(i, n - 1);
                    v(i, n - 1) = q * z + p * v(i, n);
                    v(i, n) = q * v(i, n) - p * z;
                }
            } else {                        // complex pair
                wr[n - 1] = x + p; wr[n] = x + p;
                wi[n - 1] = z; wi[n] = -z;
            }
            n -= 2; iter = 0;
        } else {                            // no convergence yet: form shift
            x = h(n, n); y = 0.0; w = 0.0;
            if (l < n) { y = h(n - 1, n - 1); w = h(n, n - 1) * h(n - 1, n); }
            if (iter == 10) {               // Wilkinson's original ad hoc shift
                exshift += x;
                for (int i = low; i <= n; ++i) h(i, i) -= x;
                s = std::fabs(h(n, n - 1)) + std::fabs(h(n - 1, n - 2));
                x = y = 0.75 * s;
                w = -0.4375 * s * s;
            }
            if (iter == 30) {               // MATLAB's new ad hoc shift
                s = (y - x) / 2.0;
                s = s * s + w;
                if (s > 0) {
                    s = std::sqrt(s);
                    if (y < x) s = -s;
                    s = x - w / ((y - x) / 2.0 + s);
                    for (int i = low; i <= n; ++i) h(i, i) -= s;
                    exshift += s;
                    x = y = w = 0.964;
                }
            }
            ++iter; ++total_iter;
            if (total_iter > 60 * nn) throw std::runtime_error("eigensolver did not converge");
            int m = n - 2;
            while (m >= l) {               // look for two consecutive small sub-diagonal elements
                z = h(m, m);
                r = x - z; s = y - z;
                p = (r * s - w) / h(m + 1, m) + h(m, m + 1);
                q = h(m + 1, m + 1) - z - r - s;
                r = h(m + 2, m + 1);
                s = std::fabs(p) + std::fabs(q) + std::fabs(r);
                p /= s; q /= s; r /= s;
                if (m == l) break;
                if (std::fabs(h(m, m - 1)) * (std::fabs(q) + std::fabs(r)) <
                    eps * (std::fabs(p) * (std::fabs(h(m - 1, m - 1)) + std::fabs(z) + std::fabs(h(m + 1, m + 1)))))
                    break;
                --m;
            }
            for (int i = m + 2; i <= n; ++i) {
                h(i, i - 2) = 0.0;
                if (i > m + 2) h(i, i - 3) = 0.0;
            }
            for (int k = m; k <= n - 1; ++k) {      // double QR step on rows l..n, columns m..n
                const bool notlast = (k != n - 1);
                if (k != m) {
                    p = h(k, k - 1); q = h(k + 1, k - 1);
                    r = notlast ? h(k + 2, k - 1) : 0.0;
                    x = std::fabs(p) + std::fabs(q) + std::fabs(r);
                    if (x == 0.0) continue;
                    p /= x; q /= x; r /= x;
                }
                s = std::sqrt(p * p + q * q + r * r);
                if (p < 0) s = -s;
                if (s != 0) {
                    if (k != m) h(k, k - 1) = -s * x;
                    else if (l != m) h(k, k - 1) = -h(k, k - 1);
                    p += s;
                    x = p / s; y = q / s; z = r / s;
                    q /= p; r /= p;
                    for (int j = k; j < nn; ++j) {
                        p = h(k, j) + q * h(k + 1, j);
                        if (notlast) { p += r * h(k + 2, j); h(k + 2, j) -= p * z; }
                        h(k, j) -= p * x;
                        h(k + 1, j) -= p * y;
                    }
                    for (int i = 0; i <= std::min(n, k + 3); ++i) {
                        p = x * h(i, k) + y * h(i, k + 1);
                        if (notlast) { p += z * h(i, k + 2); h(i, k + 2) -= p * r; }
                        h(i, k) -= p;
                        h(i, k + 1) -= p * q;
                    }
                    {
                        double *v0 = &Vt[(size_t)k * nn], *v1 = &Vt[(size_t)(k + 1) * nn];
                        if (notlast) {
                            double *v2 = &Vt[(size_t)(k + 2) * nn];
                            for (int i = low; i <= high; ++i) {
                                double pp = x * v0[i] + y * v1[i];
                                pp += z * v2[i];
                                v2[i] -= pp * r;
                                v0[i] -= pp;
                                v1[i] -= pp * q;
                            }
                        } else {
                            for (int i = low; i <= high; ++i) {
                                const double pp = x * v0[i] + y * v1[i];
                                v0[i] -= pp;
                                v1[i] -= pp * q;
                            }
                        }
                    }
                }
            }
        }
    }
    auto put_back = [&]() {
        for (int i = 0; i < nn; ++i)
            for (int j = 0; j < nn; ++j) V[(size_t)i * nn + j] = Vt[(size_t)j * nn + i];
    };
    if (norm == 0.0) { put_back(); return; }
    // back-substitute to find the vectors of the upper (quasi-)triangular form
    std::vector<double> xc(nn, 0.0);
    for (n = nn - 1; n >= 0; --n) {
        p = wr[n]; q = wi[n];
        if (q == 0) {                       // real vector
            // column n of H is mirrored in the contiguous vector xc so that the dot products below walk two
            // contiguous arrays (same operands, same order)
            int l = n;
            h(n, n) = 1.0;
            xc[n] = 1.0;
            for (int i = n - 1; i >= 0; --i) {
                w = h(i, i) - p;
                r = 0.0;
                {
                    const double *hr = &H[(size_t)i * nn];
                    for (int j = l; j <= n; ++j) r += hr[j] * xc[j];
                }
                if (wi[i] < 0.0) { z = w; s = r; }
                else {
                    l = i;
                    if (wi[i] == 0.0) {
                        if (w != 0.0) h(i, n) = -r / w;
                        else h(i, n) = -r / (eps * norm);
                        xc[i] = h(i, n);
                    } else {                // solve the 2x2 real block
                        x = h(i, i + 1); y = h(i + 1, i);
                        q = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i];
                        t = (x * s - z * r) / q;
                        h(i, n) = t;
                        if (std::fabs(x) > std::fabs(z)) h(i + 1, n) = (-r - w * t) / x;
                        else h(i + 1, n) = (-s - y * t) / z;
                        xc[i] = h(i, n); xc[i + 1] = h(i + 1, n);
                    }
                    t = std::fabs(h(i, n));            // overflow control
                    if ((eps * t) * t > 1)
                        for (int j = i; j <= n; ++j) { h(j, n) /= t; xc[j] = h(j, n); }
                }
            }
        } else if (q < 0) {                 // complex vector, stored in columns n-1 (re) and n (im)
            int l = n - 1;
            if (std::fabs(h(n, n - 1)) > std::fabs(h(n - 1, n))) {
                h(n - 1, n - 1) = q / h(n, n - 1);
                h(n - 1, n) = -(h(n, n) - p) / h(n, n - 1);
            } else {
                double cr, ci;
                cdiv(0.0, -h(n - 1, n), h(n - 1, n - 1) - p, q, cr, ci);
                h(n - 1, n - 1) = cr; h(n - 1, n) = ci;
            }
            h(n, n - 1) = 0.0;
            h(n, n) = 1.0;
            for (int i = n - 2; i >= 0; --i) {
                double ra = 0.0, sa = 0.0, vr, vi, cr, ci;
                for (int j = l; j <= n; ++j) { ra += h(i, j) * h(j, n - 1); sa += h(i, j) * h(j, n); }
                w = h(i, i) - p;
                if (wi[i] < 0.0) { z = w; r = ra; s = sa; }
                else {
                    l = i;
                    if (wi[i] == 0) {
                        cdiv(-ra, -sa, w, q, cr, ci);
                        h(i, n - 1) = cr; h(i, n) = ci;
                    } else {                // solve complex 2x2 block
                        x = h(i, i + 1); y = h(i + 1, i);
                        vr = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i] - q * q;
                        vi = (wr[i] - p) * 2.0 * q;
                        if (vr == 0.0 && vi == 0.0)
                            vr = eps * norm * (std::fabs(w) + std::fabs(q) + std::fabs(x) + std::fabs(y) + std::fabs(z));
                        cdiv(x * r - z * ra + q * sa, x * s - z * sa - q * ra, vr, vi, cr, ci);
                        h(i, n - 1) = cr; h(i, n) = ci;
                        if (std::fabs(x) > (std::fabs(z) + std::fabs(q))) {
                            h(i + 1, n - 1) = (-ra - w * h(i, n - 1) + q * h(i, n)) / x;
                            h(i + 1, n) = (-sa - w * h(i, n) - q * h(i, n - 1)) / x;
                        } else {
                            cdiv(-r - y * h(i, n - 1), -s - y * h(i, n), z, q, cr, ci);
                            h(i + 1, n - 1) = cr; h(i + 1, n) = ci;
                        }
                    }
                    t = std::max(std::fabs(h(i, n - 1)), std::fabs(h(i, n)));   // overflow control
                    if ((eps * t) * t > 1)
                        for (int j = i; j <= n; ++j) { h(j, n - 1) /= t; h(j, n) /= t; }
                }
            }
        }
    }
    // multiply by the accumulated orthogonal transform
    // V <- V * (upper triangle of H), i.e. column j of V becomes sum_{k <= j} h(k,j) * (column k of V), k ascending as
    // in the scalar form; columns of V are rows of Vt, done from the last column down so that it can run in place
    {
        std::vector<double> out(nn);
        for (int j = nn - 1; j >= low; --j) {
            std::fill(out.begin(), out.end(), 0.0);
            for (int k = low; k <= std::min(j, high); ++k) {
                const double hk = h(k, j);
                const double *vk = &Vt[(size_t)k * nn];
                for (int i = low; i <= high; ++i) out[i] += vk[i] * hk;
            }
            std::copy(out.begin() + low, out.begin() + high + 1, &Vt[(size_t)j * nn + low]);
        }
    }
    put_back();
}

// In-place inverse by LU with partial pivoting (row-major n x n).  T is double or std::complex<double>.
template <typename T>
inline void invert(int n, std::vector<T> &A) {
    std::vector<T> inv((size_t)n * n, T(0));
    for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = T(1);
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = std::abs(A[(size_t)c * n + c]);
        for (int r = c + 1; r < n; ++r) {
            double a = std::abs(A[(size_t)r * n + c]);
            if (a > best) { best = a; piv = r; }
        }
        if (best == 0.0) throw std::runtime_error("singular eigenvector matrix");
        if (piv != c)
            for (int j = 0; j < n; ++j) {
                std::swap(A[(size_t)piv * n + j], A[(size_t)c * n + j]);
                std::swap(inv[(size_t)piv * n + j], inv[(size_t)c * n + j]);
            }
        const T ip = T(1) / A[(size_t)c * n + c];
        for (int r = c + 1; r < n; ++r) {
            const T f = A[(size_t)r * n + c] * ip;
            if (f == T(0)) continue;
            for (int j = c; j < n; ++j) A[(size_t)r * n + j] -= f * A[(size_t)c * n + j];
            for (int j = 0; j < n; ++j) inv[(size_t)r * n + j] -= f * inv[(size_t)c * n + j];
        }
    }
    for (int c = n - 1; c >= 0; --c) {     // back substitution, all right-hand sides at once
        const T ip = T(1) / A[(size_t)c * n + c];
        for (int j = 0; j < n; ++j) inv[(size_t)c * n + j] *= ip;
        for (int r = 0; r < c; ++r) {
            const T f = A[(size_t)r * n + c];
            if (f == T(0)) continue;
            for (int j = 0; j < n; ++j) inv[(size_t)r * n + j] -= f * inv[(size_t)c * n + j];
        }
    }
    A.swap(inv);
}

}  // namespace detail

// eigensystem(EigenSolver(A)) with A row-major n x n.
inline EigenSystem eigensystem(int n, const std::vector<double> &A) {
    EigenSystem es;
    es.n = n;
    std::vector<double> H(A), V((size_t)n * n), wr(n), wi(n);
    if (n == 1) {
        es.P = {1.0}; es.Pinv = {1.0}; es.d = {A[0]}; es.scale = std::fabs(A[0]); es.max_imag = 0.0;
        return es;
    }
    {
        // the zero matrix (e.g. the Moran rate matrix of a single lineage) has no direction for the back-substitution
        // to normalise: every vector is an eigenvector, take the identity
        double amax = 0.0;
        for (double x : A) amax = std::max(amax, std::fabs(x));
        if (amax == 0.0) {
            es.P.assign((size_t)n * n, 0.0);
            for (int i = 0; i < n; ++i) es.P[(size_t)i * n + i] = 1.0;
            es.Pinv = es.P; es.d.assign(n, 0.0); es.scale = 0.0; es.max_imag = 0.0;
            return es;
        }
    }
    detail::orthes(n, H, V);
    detail::hqr2(n, H, V, wr, wi);
    es.d = wr;
    bool cplx = false;
    for (int i = 0; i < n; ++i) {
        es.scale = std::max(es.scale, std::hypot(wr[i], wi[i]));
        es.max_imag = std::max(es.max_imag, std::fabs(wi[i]));
        if (wi[i] != 0.0) cplx = true;
    }
    es.P.assign((size_t)n * n, 0.0);
    es.Pinv.assign((size_t)n * n, 0.0);
    if (!cplx) {
        for (int j = 0; j < n; ++j) {       // unit 2-norm columns, as EigenSolver::eigenvectors() returns them
            double nr = 0.0;
            for (int i = 0; i < n; ++i) nr += V[(size_t)i * n + j] * V[(size_t)i * n + j];
            nr = std::sqrt(nr);
            for (int i = 0; i < n; ++i) es.P[(size_t)i * n + j] = V[(size_t)i * n + j] / nr;
        }
        std::vector<double> inv(es.P);
        detail::invert<double>(n, inv);
        es.Pinv = inv;
        return es;
    }
    using cd = std::complex<double>;
    std::vector<cd> Pc((size_t)n * n);
    for (int j = 0; j < n; ++j) {
        if (wi[j] == 0.0) {
            double nr = 0.0;
            for (int i = 0; i < n; ++i) nr += V[(size_t)i * n + j] * V[(size_t)i * n + j];
            nr = std::sqrt(nr);
            for (int i = 0; i < n; ++i) Pc[(size_t)i * n + j] = cd(V[(size_t)i * n + j] / nr, 0.0);
        } else if (wi[j] > 0.0 && j + 1 < n) {   // columns j (re), j+1 (im): v = re + i im, conj for j+1
            double nr = 0.0;
            for (int i = 0; i < n; ++i)
                nr += V[(size_t)i * n + j] * V[(size_t)i * n + j] + V[(size_t)i * n + j + 1] * V[(size_t)i * n + j + 1];
            nr = std::sqrt(nr);
            for (int i = 0; i < n; ++i) {
                const double re = V[(size_t)i * n + j] / nr, im = V[(size_t)i * n + j + 1] / nr;
                Pc[(size_t)i * n + j] = cd(re, im);
                Pc[(size_t)i * n + j + 1] = cd(re, -im);
            }
            ++j;
        }
    }
    std::vector<cd> Pi(Pc);
    detail::invert<cd>(n, Pi);
    for (size_t i = 0; i < (size_t)n * n; ++i) { es.P[i] = Pc[i].real(); es.Pinv[i] = Pi[i].real(); }
    return es;
}

}  // namespace smcpp_host
