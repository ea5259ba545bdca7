// Team-parallel variant of nonsym_eig.hpp for the large eigenproblems of the cold preparation (n >= 128: at n = 256 one
// serial eigensystem takes 35 ms of host time per E-step).  Same EISPACK pipeline, same arithmetic in
// the same order on every matrix element - the result is BIT-IDENTICAL to smcpp_host::eigensystem() - with the work
// split over a small team of threads that all call eigensystem_team() (SPMD) and meet at spin barriers:
//   * orthes: each Householder step updates column blocks, then row blocks (two barriers per step); the accumulation of
//     the transforms is column-independent and needs no barrier at all;
//   * hqr2: the double-shift QR sweeps themselves are sequential (rank 0), but they never READ the accumulated
//     transform V - every reflection is appended to a list that the other ranks apply to their slices of V^T while rank 0
//     is still iterating;
//   * back-substitution (one eigenvector per task), V * X, the column norms, the LU factorisation (row blocks, two
//     barriers per pivot) and the triangular solves for the inverse (column panels, no barrier) are split the same way.
// Matrices with complex eigenvalues or any failure fall back to the serial routine on rank 0.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "nonsym_eig.hpp"

#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define SMCPP_CPU_RELAX() _mm_pause()
#else
#define SMCPP_CPU_RELAX() ((void)0)
#endif
#include <sched.h>

namespace smcpp_host {

struct QrStep {           // one similarity step of hqr2 as far as V is concerned
    int kind;             // 0 = three-row reflection, 1 = two-row reflection (last of a sweep), 2 = rotation of a real pair
    int k;                // first of the rows of V^T it touches
    double x, y, z, q, r;
};

// 64-byte aligned storage: the column / row blocks the ranks own start on cache-line boundaries (with std::vector's
// 16-byte alignment every line of a row was shared by two ranks - measured: the parallel reduction ran no faster than
// the serial one)
struct AlignedBuf {
    double *p = nullptr;
    size_t cap = 0;
    AlignedBuf() = default;
    AlignedBuf(const AlignedBuf &) = delete;
    AlignedBuf &operator=(const AlignedBuf &) = delete;
    ~AlignedBuf() { std::free(p); }
    void ensure(size_t n) {
        if (n <= cap) return;
        std::free(p);
        void *q = nullptr;
        if (posix_memalign(&q, 64, ((n * sizeof(double) + 63) / 64) * 64) != 0) throw std::bad_alloc();
        p = static_cast<double *>(q);
        cap = n;
    }
};

struct EigTeam {
    int size = 1;
    alignas(64) std::atomic<int> arrived{0};
    alignas(64) std::atomic<int> generation{0};
    // list of V-steps: written by rank 0 only, read by the others up to `published` (own cache lines: the readers poll them)
    alignas(64) std::atomic<long> published{0};
    alignas(64) std::atomic<int> qr_done{0};
    alignas(64) std::atomic<int> failed{0};
    // shared state of one eigenproblem
    alignas(64) int n = 0;
    AlignedBuf H, V, Vt, X, W, lu, LmT, UT, inv;
    std::vector<double> ortm, wr, wi;
    std::vector<int> perm;
    QrStep *steps = nullptr;
    size_t steps_cap = 0, steps_n = 0;
    std::string error;
    double norm = 0.0;
    bool overflow = false;

    explicit EigTeam(int team_size) : size(team_size) {}
    ~EigTeam() { std::free(steps); }
    EigTeam(const EigTeam &) = delete;
    EigTeam &operator=(const EigTeam &) = delete;

    void barrier(int &gen) {
        const int g = gen + 1;
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == size) {
            arrived.store(0, std::memory_order_relaxed);
            generation.store(g, std::memory_order_release);
        } else {
            int spins = 0;
            while (generation.load(std::memory_order_acquire) < g) {
                SMCPP_CPU_RELAX();
                if (++spins > 20000) { sched_yield(); spins = 0; }
            }
        }
        gen = g;
    }
};

// ---- optional placement: one team = one L3 domain ------------------------------------------------------------------------
// The team algorithm hands every matrix element from one owner to another twice per Householder step; between cores that share
// an L3 slice that is a 40 ns transfer, between CCDs or sockets it is 150-300 ns and the parallel reduction runs slower than
// the serial one (measured on a two-socket EPYC 9575F, unpinned: 28 ms vs 5.5 ms).  cpu_l3_groups() lists the allowed CPUs by
// L3 domain (sysfs); ScopedAffinity confines the calling thread to one of them and restores its mask on destruction.
inline std::vector<std::vector<int>> cpu_l3_groups() {
    std::vector<std::vector<int>> groups;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return groups;
    std::vector<char> seen(CPU_SETSIZE, 0);
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (!CPU_ISSET(c, &allowed) || seen[c]) continue;
        char path[128];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", c);
        FILE *f = fopen(path, "r");
        if (!f) return std::vector<std::vector<int>>();
        char buf[4096];
        const bool got = fgets(buf, sizeof(buf), f) != nullptr;
        fclose(f);
        if (!got) return std::vector<std::vector<int>>();
        std::vector<int> g;
        for (char *t = buf; *t && *t != '\n';) {              // "0-7,128-135"
            char *e = nullptr;
            const long a = strtol(t, &e, 10);
            long b = a;
            if (e == t) break;
            if (*e == '-') { t = e + 1; b = strtol(t, &e, 10); }
            for (long x = a; x <= b && x < CPU_SETSIZE; ++x)
                if (CPU_ISSET((int)x, &allowed) && !seen[x]) { seen[x] = 1; g.push_back((int)x); }
            t = (*e == ',') ? e + 1 : e;
            if (*e != ',' ) break;
        }
        if (!g.empty()) groups.push_back(g);
    }
    return groups;
}

struct ScopedAffinity {
    cpu_set_t saved;
    bool active = false;
    explicit ScopedAffinity(const std::vector<int> *cpus) {
        if (!cpus || cpus->empty()) return;
        if (sched_getaffinity(0, sizeof(saved), &saved) != 0) return;
        cpu_set_t want;
        CPU_ZERO(&want);
        for (int c : *cpus) CPU_SET(c, &want);
        active = sched_setaffinity(0, sizeof(want), &want) == 0;
    }
    ~ScopedAffinity() { if (active) (void)sched_setaffinity(0, sizeof(saved), &saved); }
};

namespace detail {

// block-cyclic ownership: blocks of BS consecutive indices dealt round-robin (contiguous inner loops, balanced triangles)
template <typename F>
inline void for_blocks(int lo, int hi, int rank, int size, int BS, F &&f) {
    for (int b = lo / BS; b * BS < hi; ++b) {
        if (b % size != rank) continue;
        const int s = std::max(lo, b * BS), e = std::min(hi, (b + 1) * BS);
        if (s < e) f(s, e);
    }
}

inline void apply_step(const QrStep &st, double *Vt, int nn, int i0, int i1) {
    double *v0 = Vt + (size_t)st.k * nn, *v1 = v0 + nn;
    if (st.kind == 0) {
        double *v2 = v1 + nn;
        const double x = st.x, y = st.y, z = st.z, q = st.q, r = st.r;
        for (int i = i0; i < i1; ++i) {
            double pp = x * v0[i] + y * v1[i];
            pp += z * v2[i];
            v2[i] -= pp * r;
            v0[i] -= pp;
            v1[i] -= pp * q;
        }
    } else if (st.kind == 1) {
        const double x = st.x, y = st.y, q = st.q;
        for (int i = i0; i < i1; ++i) {
            const double pp = x * v0[i] + y * v1[i];
            v0[i] -= pp;
            v1[i] -= pp * q;
        }
    } else {
        const double p = st.x, q = st.q;
        for (int i = i0; i < i1; ++i) {
            const double z = v0[i];
            v0[i] = q * z + p * v1[i];
            v1[i] = q * v1[i] - p * z;
        }
    }
}

// The QR iteration of hqr2 on H alone (identical arithmetic); every transformation of V goes to `tm.steps` and is
// published in batches.  Throws like hqr2.
inline void hqr_iterate_record(EigTeam &tm) {
    const int nn = tm.n;
    double *H = tm.H.p;
    std::vector<double> &wr = tm.wr, &wi = tm.wi;
    auto h = [&](int i, int j) -> double & { return H[(size_t)i * nn + j]; };
    QrStep *steps = tm.steps;
    const size_t cap = tm.steps_cap;
    size_t count = 0, unpublished = 0;
    auto push = [&](const QrStep &st) {
        if (count == cap) { tm.overflow = true; return; }          // readers hold the pointer: never reallocate
        steps[count++] = st;
        if (++unpublished >= 64) { tm.published.store((long)count, std::memory_order_release); unpublished = 0; }
    };
    int n = nn - 1;
    const int low = 0;
    const double eps = std::pow(2.0, -52.0);
    double exshift = 0.0, p = 0, q = 0, r = 0, s = 0, z = 0, w, x, y;
    double norm = 0.0;
    for (int i = 0; i < nn; ++i)
        for (int j = std::max(i - 1, 0); j < nn; ++j) norm += std::fabs(h(i, j));
    tm.norm = norm;
    int iter = 0, total_iter = 0;
    while (n >= low) {
        int l = n;
        while (l > low) {
            s = std::fabs(h(l - 1, l - 1)) + std::fabs(h(l, l));
            if (s == 0.0) s = norm;
            if (std::fabs(h(l, l - 1)) < eps * s) break;
            --l;
        }
        if (l == n) {
            h(n, n) += exshift;
            wr[n] = h(n, n); wi[n] = 0.0;
            --n; iter = 0;
        } else if (l == n - 1) {
            w = h(n, n - 1) * h(n - 1, n);
            p = (h(n - 1, n - 1) - h(n, n)) / 2.0;
            q = p * p + w;
            z = std::sqrt(std::fabs(q));
            h(n, n) += exshift;
            h(n - 1, n - 1) += exshift;
            x = h(n, n);
            if (q >= 0) {
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
                push(QrStep{2, n - 1, p, 0.0, 0.0, q, 0.0});
            } else {
                wr[n - 1] = x + p; wr[n] = x + p;
                wi[n - 1] = z; wi[n] = -z;
            }
            n -= 2; iter = 0;
        } else {
            x = h(n, n); y = 0.0; w = 0.0;
            if (l < n) { y = h(n - 1, n - 1); w = h(n, n - 1) * h(n - 1, n); }
            if (iter == 10) {
                exshift += x;
                for (int i = low; i <= n; ++i) h(i, i) -= x;
                s = std::fabs(h(n, n - 1)) + std::fabs(h(n - 1, n - 2));
                x = y = 0.75 * s;
                w = -0.4375 * s * s;
            }
            if (iter == 30) {
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
            while (m >= l) {
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
            for (int k = m; k <= n - 1; ++k) {
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
                    {
                        double *h0 = &H[(size_t)k * nn], *h1 = h0 + nn;
                        if (notlast) {
                            double *h2 = h1 + nn;
                            for (int j = k; j < nn; ++j) {
                                double pp = h0[j] + q * h1[j];
                                pp += r * h2[j];
                                h2[j] -= pp * z;
                                h0[j] -= pp * x;
                                h1[j] -= pp * y;
                            }
                        } else {
                            for (int j = k; j < nn; ++j) {
                                const double pp = h0[j] + q * h1[j];
                                h0[j] -= pp * x;
                                h1[j] -= pp * y;
                            }
                        }
                    }
                    for (int i = 0; i <= std::min(n, k + 3); ++i) {
                        double *hr = &H[(size_t)i * nn + k];
                        p = x * hr[0] + y * hr[1];
                        if (notlast) { p += z * hr[2]; hr[2] -= p * r; }
                        hr[0] -= p;
                        hr[1] -= p * q;
                    }
                    push(QrStep{notlast ? 0 : 1, k, x, y, z, q, r});
                }
            }
        }
    }
    tm.steps_n = count;
    tm.published.store((long)count, std::memory_order_release);
}

}  // namespace detail

// SPMD: every thread of the team calls this with its rank; `es` is written by the team (valid on return on every rank
// after the final barrier).  `gen` is the caller's barrier generation counter (start at the team's current generation).
inline void eigensystem_team(int n, const std::vector<double> &A, EigenSystem &es, EigTeam &tm, int rank, int &gen) {
    using namespace detail;
    const int T = tm.size;
    constexpr int BS = 16;      // doubles per ownership block: two cache lines of the 64-byte aligned rows
    const bool tmg = opt().has(smcpp_opt::O_HOST_TIMING);
    const auto tc0 = std::chrono::steady_clock::now();
    double marks[8] = {0};
    auto mark = [&](int k) { if (tmg && rank == 0) marks[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count(); };
    auto fail_serial = [&]() {
        // complex spectrum / anything unusual: rank 0 runs the serial routine, the others wait
        if (rank == 0) {
            try { es = eigensystem(n, A); } catch (const std::exception &ex) { tm.error = ex.what(); tm.failed.store(2); }
        }
        tm.barrier(gen);
    };
    const size_t NN = (size_t)n * n;
    if (rank == 0) {
        tm.n = n;
        tm.H.ensure(NN); tm.V.ensure(NN); tm.Vt.ensure(NN); tm.X.ensure(NN); tm.W.ensure(NN); tm.lu.ensure(NN);
        tm.LmT.ensure(NN); tm.UT.ensure(NN); tm.inv.ensure(NN);
        std::memcpy(tm.H.p, A.data(), NN * sizeof(double));
        std::memset(tm.X.p, 0, NN * sizeof(double));
        tm.ortm.assign(n, 0.0);
        tm.wr.assign(n, 0.0); tm.wi.assign(n, 0.0);
        const size_t want = (size_t)n * 400;
        if (tm.steps_cap < want) {
            std::free(tm.steps);
            tm.steps = static_cast<QrStep *>(std::malloc(want * sizeof(QrStep)));
            if (!tm.steps) throw std::bad_alloc();
            tm.steps_cap = want;
        }
        tm.steps_n = 0;
        tm.published.store(0); tm.qr_done.store(0); tm.failed.store(0); tm.overflow = false;
        tm.error.clear();
        es.n = n; es.scale = 0.0; es.max_imag = 0.0;
        es.P.assign(NN, 0.0); es.Pinv.assign(NN, 0.0); es.d.assign(n, 0.0);
    }
    tm.barrier(gen);
    double *H = tm.H.p, *V = tm.V.p, *Vt = tm.Vt.p;
    const int high = n - 1;
    // ---------------- orthes: reduction ----------------
    {
        std::vector<double> ort(n, 0.0), fcol(n, 0.0);
        for (int m = 1; m <= high - 1; ++m) {
            double sc = 0.0;
            for (int i = m; i <= high; ++i) sc += std::fabs(H[(size_t)i * n + m - 1]);
            if (sc != 0.0) {
                double hh = 0.0;
                for (int i = high; i >= m; --i) { ort[i] = H[(size_t)i * n + m - 1] / sc; hh += ort[i] * ort[i]; }
                double g = std::sqrt(hh);
                if (ort[m] > 0) g = -g;
                hh -= ort[m] * g;
                ort[m] -= g;
                for_blocks(m, n, rank, T, BS, [&](int j0, int j1) {
                    for (int j = j0; j < j1; ++j) fcol[j] = 0.0;
                    for (int i = high; i >= m; --i) {
                        const double oi = ort[i];
                        const double *hr = H + (size_t)i * n;
                        for (int j = j0; j < j1; ++j) fcol[j] += oi * hr[j];
                    }
                    for (int j = j0; j < j1; ++j) fcol[j] /= hh;
                    for (int i = m; i <= high; ++i) {
                        const double oi = ort[i];
                        double *hr = H + (size_t)i * n;
                        for (int j = j0; j < j1; ++j) hr[j] -= fcol[j] * oi;
                    }
                });
                tm.barrier(gen);
                for_blocks(0, high + 1, rank, T, BS, [&](int i0, int i1) {
                    for (int i = i0; i < i1; ++i) {
                        double *hr = H + (size_t)i * n;
                        double f = 0.0;
                        for (int j = high; j >= m; --j) f += ort[j] * hr[j];
                        f /= hh;
                        for (int j = m; j <= high; ++j) hr[j] -= f * ort[j];
                    }
                });
                if (rank == 0) tm.ortm[m] = sc * ort[m];
                tm.barrier(gen);
                if (rank == 0) H[(size_t)m * n + m - 1] = sc * g;       // column m-1 is not read again by the reduction
            }
        }
        tm.barrier(gen);
        mark(0);
        // ---------------- orthes: accumulation of the transforms (column blocks, no barriers) ----------------
        for_blocks(0, n, rank, T, BS, [&](int j0, int j1) {
            for (int i = 0; i < n; ++i)
                for (int j = j0; j < j1; ++j) V[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
            for (int m = high - 1; m >= 1; --m) {
                const double hm = H[(size_t)m * n + m - 1];
                if (hm == 0.0) continue;
                const int a0 = std::max(j0, m), a1 = std::min(j1, high + 1);
                if (a0 >= a1) continue;
                for (int i = m + 1; i <= high; ++i) ort[i] = H[(size_t)i * n + m - 1];
                ort[m] = tm.ortm[m];
                for (int j = a0; j < a1; ++j) fcol[j] = 0.0;
                for (int i = m; i <= high; ++i) {
                    const double oi = ort[i];
                    const double *vr = V + (size_t)i * n;
                    for (int j = a0; j < a1; ++j) fcol[j] += oi * vr[j];
                }
                const double om = ort[m];
                for (int j = a0; j < a1; ++j) fcol[j] = (fcol[j] / om) / hm;
                for (int i = m; i <= high; ++i) {
                    const double oi = ort[i];
                    double *vr = V + (size_t)i * n;
                    for (int j = a0; j < a1; ++j) vr[j] += fcol[j] * oi;
                }
            }
            // transposed copy for the QR phase: Vt[j][i] = V[i][j]
            for (int j = j0; j < j1; ++j)
                for (int i = 0; i < n; ++i) Vt[(size_t)j * n + i] = V[(size_t)i * n + j];
        });
    }
    tm.barrier(gen);
    mark(1);
    // ---------------- hqr2: rank 0 iterates on H, the others apply the published steps to their slices of V^T ----------------
    if (rank == 0) {
        try { hqr_iterate_record(tm); } catch (const std::exception &ex) { tm.error = ex.what(); tm.failed.store(1); }
        tm.qr_done.store(1, std::memory_order_release);
        mark(2);
    }
    {
        // slices of the row index i of V^T: ranks 1..T-1 work while rank 0 iterates (a team of one does everything itself)
        const int workers = std::max(1, T - 1), wr_ = T > 1 ? rank - 1 : 0;
        const int per = (n + workers - 1) / workers;
        const int i0 = (T > 1 && rank == 0) ? n : std::min(n, wr_ * per), i1 = std::min(n, i0 + per);
        long done = 0;
        while (true) {
            const int fin = tm.qr_done.load(std::memory_order_acquire);
            const long avail = tm.published.load(std::memory_order_acquire);
            if (avail > done) {
                const QrStep *st = tm.steps;            // sized before the phase: no reallocation while readers are active
                for (long t = done; t < avail; ++t) apply_step(st[t], Vt, n, i0, i1);
                done = avail;
            } else if (fin) {
                if (tm.published.load(std::memory_order_acquire) == done) break;
            } else SMCPP_CPU_RELAX();
        }
    }
    tm.barrier(gen);
    mark(3);
    if (tm.failed.load()) { if (rank == 0) { /* error text already set */ } return; }
    bool cplx = false;
    for (int i = 0; i < n; ++i) cplx = cplx || tm.wi[i] != 0.0;
    if (cplx || tm.norm == 0.0 || tm.overflow) {
        if (tmg && rank == 0) fprintf(stderr, "[eig-team] serial fallback: complex %d, zero norm %d, step list overflow %d (%zu steps)\n",
                                      (int)cplx, (int)(tm.norm == 0.0), (int)tm.overflow, tm.steps_n);
        fail_serial();
        return;
    }
    // ---------------- back-substitution: one (real) eigenvector of the triangular form per task ----------------
    {
        const double eps = std::pow(2.0, -52.0), norm = tm.norm;
        double *X = tm.X.p;
        std::vector<double> xc(n, 0.0);
        for (int nv = n - 1 - rank; nv >= 0; nv -= T) {
            const double p = tm.wr[nv];
            xc[nv] = 1.0;
            for (int i = nv - 1; i >= 0; --i) {
                const double *hr = H + (size_t)i * n;
                const double w = hr[i] - p;
                double r = 0.0;
                for (int j = i + 1; j <= nv; ++j) r += hr[j] * xc[j];
                if (w != 0.0) xc[i] = -r / w;
                else xc[i] = -r / (eps * norm);
                const double t = std::fabs(xc[i]);
                if ((eps * t) * t > 1)
                    for (int j = i; j <= nv; ++j) xc[j] /= t;
            }
            for (int i = 0; i <= nv; ++i) X[(size_t)i * n + nv] = xc[i];
        }
    }
    tm.barrier(gen);
    mark(4);
    // ---------------- V * X (row j of W = final column j of V), column norms, P ----------------
    {
        const double *X = tm.X.p;
        double *W = tm.W.p;
        for (int j = n - 1 - rank; j >= 0; j -= T) {
            double *out = W + (size_t)j * n;
            for (int i = 0; i < n; ++i) out[i] = 0.0;
            for (int k = 0; k <= j; ++k) {
                const double hk = X[(size_t)k * n + j];
                const double *vk = Vt + (size_t)k * n;
                for (int i = 0; i < n; ++i) out[i] += vk[i] * hk;
            }
            double nr = 0.0;
            for (int i = 0; i < n; ++i) nr += out[i] * out[i];
            nr = std::sqrt(nr);
            for (int i = 0; i < n; ++i) es.P[(size_t)i * n + j] = out[i] / nr;
        }
    }
    tm.barrier(gen);
    mark(5);
    // ---------------- inverse of P: LU with partial pivoting on row blocks, then column panels of the inverse ----------------
    {
        double *L = tm.lu.p;
        double *LmT = tm.LmT.p;         // multipliers, TRANSPOSED (LmT[c][r] = l_rc): the solves read them along r.  Kept OUT of
                                        // `lu`: a rank that is already eliminating must not change column c under a rank still
                                        // searching its pivot
        double *UT = tm.UT.p;           // UT[c][r] = u_rc for r <= c, filled after the factorisation
        if (rank == 0) { tm.perm.resize(n); for (int i = 0; i < n; ++i) tm.perm[i] = i; }
        for_blocks(0, n, rank, T, 8, [&](int r0, int r1) {
            for (int r = r0; r < r1; ++r) std::memcpy(L + (size_t)r * n, &es.P[(size_t)r * n], n * sizeof(double));
        });
        tm.barrier(gen);
        for (int c = 0; c < n; ++c) {
            int piv = c;
            double best = std::fabs(L[(size_t)c * n + c]);
            for (int r = c + 1; r < n; ++r) {
                const double a = std::fabs(L[(size_t)r * n + c]);
                if (a > best) { best = a; piv = r; }
            }
            if (best == 0.0) {          // every rank sees the same column: consistent exit
                if (rank == 0) { tm.error = "singular eigenvector matrix"; tm.failed.store(3); }
                tm.barrier(gen);
                return;
            }
            if (piv != c) {
                tm.barrier(gen);        // everybody has finished reading column c
                if (rank == 0) {
                    for (int j = 0; j < n; ++j) std::swap(L[(size_t)piv * n + j], L[(size_t)c * n + j]);
                    for (int j = 0; j < c; ++j) std::swap(LmT[(size_t)j * n + piv], LmT[(size_t)j * n + c]);
                    std::swap(tm.perm[piv], tm.perm[c]);
                }
                tm.barrier(gen);
            }
            const double ip = 1.0 / L[(size_t)c * n + c];
            const double *lc = L + (size_t)c * n;
            double *mc = LmT + (size_t)c * n;
            for_blocks(c + 1, n, rank, T, 8, [&](int r0, int r1) {
                for (int r = r0; r < r1; ++r) {
                    double *lr = L + (size_t)r * n;
                    const double f = lr[c] * ip;
                    mc[r] = f;                                  // multiplier kept for the forward substitutions
                    if (f == 0.0) continue;
                    for (int j = c + 1; j < n; ++j) lr[j] -= f * lc[j];
                }
            });
            tm.barrier(gen);
        }
        mark(6);
        for_blocks(0, n, rank, T, 8, [&](int c0, int c1) {
            for (int c = c0; c < c1; ++c)
                for (int r = 0; r <= c; ++r) UT[(size_t)c * n + r] = L[(size_t)r * n + c];
        });
        tm.barrier(gen);
        // inverse, a panel of PW columns at a time: column j of inv(P) solves L U x = Perm e_j
        constexpr int PW = 32;
        double *inv = tm.inv.p;
        for_blocks(0, n, rank, T, PW, [&](int j0, int j1) {
            const int wdt = j1 - j0;
            for (int r = 0; r < n; ++r)
                for (int j = 0; j < wdt; ++j) inv[(size_t)r * n + j0 + j] = (tm.perm[r] == j0 + j) ? 1.0 : 0.0;
            for (int c = 0; c < n; ++c) {                         // forward: rows below c
                const double *xc = inv + (size_t)c * n + j0;
                const double *mc = LmT + (size_t)c * n;
                for (int r = c + 1; r < n; ++r) {
                    const double f = mc[r];
                    if (f == 0.0) continue;
                    double *xr = inv + (size_t)r * n + j0;
                    for (int j = 0; j < wdt; ++j) xr[j] -= f * xc[j];
                }
            }
            for (int c = n - 1; c >= 0; --c) {                    // backward
                const double *uc = UT + (size_t)c * n;
                const double ip = 1.0 / uc[c];
                double *xc = inv + (size_t)c * n + j0;
                for (int j = 0; j < wdt; ++j) xc[j] *= ip;
                for (int r = 0; r < c; ++r) {
                    const double f = uc[r];
                    if (f == 0.0) continue;
                    double *xr = inv + (size_t)r * n + j0;
                    for (int j = 0; j < wdt; ++j) xr[j] -= f * xc[j];
                }
            }
            for (int r = 0; r < n; ++r) std::memcpy(&es.Pinv[(size_t)r * n + j0], inv + (size_t)r * n + j0, wdt * sizeof(double));
        });
    }
    if (rank == 0) {
        es.d = tm.wr;
        for (int i = 0; i < n; ++i) es.scale = std::max(es.scale, std::fabs(tm.wr[i]));
        es.max_imag = 0.0;
    }
    tm.barrier(gen);
    mark(7);
    if (tmg && rank == 0)
        fprintf(stderr, "[eig-team] n=%d T=%d steps=%zu: reduce %.2f accumulate %.2f qr(rank0) %.2f +V tail %.2f backsub %.2f multiply %.2f "
                "lu %.2f solves %.2f ms\n", n, T, tm.steps_n, marks[0], marks[1] - marks[0], marks[2] - marks[1],
                marks[3] - marks[2], marks[4] - marks[3], marks[5] - marks[4], marks[6] - marks[5], marks[7] - marks[6]);
}

}  // namespace smcpp_host
