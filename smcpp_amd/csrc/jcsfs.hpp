// Host-side preparation for two-population managers: the joint conditioned SFS and the two-population emission table.
//
// What it computes follows the reference's JointCSFS (src/jcsfs.cpp:83-420, include/jcsfs.h; the better documented
// Python original is smcpp/jcsfs.py) and the generic-P emission assembly of NPopInferenceManager
// (src/inference_manager.cpp:263-482, include/bin_key.h, include/marginalize_key.h, include/tensorslice.h).
//
// PARITY NOTE: the reference translation unit jcsfs.cpp includes GSL headers and GSL is not part of this image, so it
// cannot be compiled here and the numbers of this file are NOT pinned against a reference build.  They are pinned
// (tests/test_jcsfs.py) by the invariants the reference's own tests assert (test/unit/test_jcsfs.py: pop-1 and pop-2
// marginals against raw_sfs), by exact identities (total branch length, symmetry of the two helper paths at the
// split) and, for every ingredient that is shared with the one-population path (rate function, conditioned SFS,
// shift/truncate), by the goldens of that path.
//
// Everything is templated on the scalar so the same code yields values (double) and forward-mode Jacobians (dual).
#pragma once
#include <atomic>
#include <string>

#include "nonsym_eig.hpp"
#include "prep.hpp"

namespace smcpp_host {

// model seen from `shift` units back in time (common.cpp:63-78)
template <typename S>
inline ModelParamsT<S> shift_params(const ModelParamsT<S> &p, double shift) {
    const int K = (int)p.s.size();
    std::vector<double> cs(K + 1, 0.0);
    for (int k = 0; k < K; ++k) cs[k + 1] = cs[k] + p.s[k];
    cs[K] = INFINITY;
    const int ip = (int)(std::upper_bound(cs.begin(), cs.end(), shift) - cs.begin()) - 1;
    ModelParamsT<S> r;
    r.s.assign(p.s.begin() + ip, p.s.end());
    r.a.assign(p.a.begin() + ip, p.a.end());
    r.s[0] = cs[ip + 1] - shift;
    r.s.back() = 1.0;
    return r;
}

// model cut at `tt` and crashed to size 1e-8 above it, so that nothing coalesces later (common.cpp:80-97)
template <typename S>
inline ModelParamsT<S> truncate_params(const ModelParamsT<S> &p, double tt) {
    const int K = (int)p.s.size();
    std::vector<double> cs(K + 1, 0.0);
    for (int k = 0; k < K; ++k) cs[k + 1] = cs[k] + p.s[k];
    cs[K] = INFINITY;
    const int ip = (int)(std::upper_bound(cs.begin(), cs.end(), tt) - cs.begin()) - 1;
    ModelParamsT<S> r;
    r.s.assign(p.s.begin(), p.s.begin() + ip + 1);
    r.a.assign(p.a.begin(), p.a.begin() + ip + 1);
    r.s[ip] = tt - cs[ip];
    r.s.push_back(1.0);
    r.a.push_back(S(1e-8));
    return r;
}

// exp(t * Q) of a (modified) Moran rate matrix through its eigensystem (jcsfs.h:38-58)
struct MoranExp {
    int dim = 0;
    EigenSystem es;
    MoranExp() {}
    // plain Moran model on N lineages (moran_eigensystem.cpp:8-29)
    static MoranExp plain(int N) {
        std::vector<double> Q((size_t)(N + 1) * (N + 1), 0.0);
        for (int i = 0; i <= N; ++i) {
            double sm = 0.0;
            const double b = 0.5 * i * (N - i);
            if (i > 0) { Q[(size_t)i * (N + 1) + i - 1] = b; sm += b; }
            if (i < N) { Q[(size_t)i * (N + 1) + i + 1] = b; sm += b; }
            Q[(size_t)i * (N + 1) + i] = -sm;
        }
        return MoranExp(N + 1, Q);
    }
    // Moran model on N lineages next to `na` distinguished ones of which `a` are derived (moran_eigensystem.cpp:31-52)
    static MoranExp modified(int N, int a, int na) {
        std::vector<double> Q((size_t)(N + 1) * (N + 1), 0.0);
        for (int i = 0; i <= N; ++i) {
            double sm = 0.0;
            if (i > 0) { const double b = (double)(na - a) * i + 0.5 * i * (N - i); Q[(size_t)i * (N + 1) + i - 1] = b; sm += b; }
            if (i < N) { const double b = (double)a * (N - i) + 0.5 * i * (N - i); Q[(size_t)i * (N + 1) + i + 1] = b; sm += b; }
            Q[(size_t)i * (N + 1) + i] = -sm;
        }
        return MoranExp(N + 1, Q);
    }
    template <typename S>
    std::vector<S> expM(const S &t) const {
        std::vector<S> eD(dim);
        for (int j = 0; j < dim; ++j) eD[j] = m_exp(t * es.d[j]);
        std::vector<S> out((size_t)dim * dim, S(0.0));
        for (int i = 0; i < dim; ++i)
            for (int j = 0; j < dim; ++j) {
                const S pe = eD[j] * es.P[(size_t)i * dim + j];
                for (int k = 0; k < dim; ++k) out[(size_t)i * dim + k] += pe * es.Pinv[(size_t)j * dim + k];
            }
        return out;
    }

private:
    MoranExp(int d, const std::vector<double> &Q) : dim(d), es(eigensystem(d, Q)) {}
};

// fold a 3 x (n+1) conditioned SFS over the distinguished pair: entry k-1 = total weight of k derived among n+2
// (jcsfs.cpp:57-69)
template <typename S>
inline std::vector<S> undistinguished_sfs(const std::vector<S> &csfs, int n) {
    std::vector<S> ret(n + 1, S(0.0));
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < n + 1; ++b)
            if (1 <= a + b && a + b < n + 2) ret[a + b - 1] += csfs[(size_t)a * (n + 1) + b];
    return ret;
}

// Optional accelerator for the two BATCHED conditioned-SFS problems of together() (values only): the engine implements it on the
// device (k_prep_tables + k_prep_csfs_raw of prep_dev.hpp).  launch() enqueues and returns (false: not taken - the caller computes the
// batch on the host team), collect() waits and hands the states' tables back.  `which`: 0 = below the split, 1 = above it.
struct CsfsBatchDevice {
    virtual ~CsfsBatchDevice() {}
    virtual bool launch(int which, const RateFunctionT<double> &eta, int n) = 0;
    virtual void collect(int which, std::vector<std::vector<double>> &out) = 0;
};
template <typename S> struct CsfsBatchHook {
    static bool launch(CsfsBatchDevice *, int, const RateFunctionT<S> &, int) { return false; }
    static void collect(CsfsBatchDevice *, int, std::vector<std::vector<S>> &) {}
};
template <> struct CsfsBatchHook<double> {
    static bool launch(CsfsBatchDevice *d, int which, const RateFunctionT<double> &eta, int n) { return d && d->launch(which, eta, n); }
    static void collect(CsfsBatchDevice *d, int which, std::vector<std::vector<double>> &out) { d->collect(which, out); }
};

template <typename S>
class JointCsfsT {
public:
    CsfsBatchDevice *batch_dev = nullptr;       // (set by TwoPopPrep::jcsfs; used by the double instantiation only)
    JointCsfsT(int n1, int n2, int a1, int a2, const std::vector<double> &hidden_states, int K = 10)
        : n1(n1), n2(n2), a1(a1), a2(a2), hs(hidden_states), M((int)hidden_states.size() - 1), K(K) {
        if (!((a1 == 2 && a2 == 0) || (a1 == 1 && a2 == 1))) throw std::runtime_error("unsupported jcsfs configuration");
        d2 = n2 + 1; d1 = (a2 + 1) * d2; d0 = (n1 + 1) * d1;
        if (a1 == 2) {
            Mn1p1 = MoranExp::plain(n1 + 1);
            Mn2 = MoranExp::plain(n2);
            Mn10 = MoranExp::modified(n1, 0, 2);
            Mn11 = MoranExp::modified(n1, 1, 2);
            Mn12 = MoranExp::modified(n1, 2, 2);
            // tau_below_split averages  A_k diag(s0) B_k  and  A_k diag(s2) C_k  over K random times, A / B / C being matrix
            // exponentials of Moran models: in their eigenbases only the K rank-one products of eigenvalue powers depend on the
            // time, the sandwich  W0 = P1^-1 diag(s0) P0  (W2 likewise) depends on n1 alone - formed here, once
            const int r = n1 + 2, c = n1 + 1;
            W0_.assign((size_t)r * c, 0.0); W2_.assign((size_t)r * c, 0.0);
            for (int x = 0; x < r; ++x)
                for (int y = 0; y < c; ++y) {
                    double w0 = 0.0, w2 = 0.0;
                    for (int q = 0; q < c; ++q) {
                        w0 += Mn1p1.es.Pinv[(size_t)x * r + q] * (1.0 - (double)q / (double)(n1 + 1)) * Mn10.es.P[(size_t)q * c + y];
                        w2 += Mn1p1.es.Pinv[(size_t)x * r + q + 1] * ((double)(q + 1) / (double)(n1 + 1)) * Mn12.es.P[(size_t)q * c + y];
                    }
                    W0_[(size_t)x * c + y] = w0; W2_[(size_t)x * c + y] = w2;
                }
        } else {
            Mn10 = MoranExp::modified(n1, 0, 1);
            Mn11 = MoranExp::modified(n1, 1, 1);
            Mn20 = MoranExp::modified(n2, 0, 1);
            Mn21 = MoranExp::modified(n2, 1, 1);
        }
        // hypergeometric tables (jcsfs.cpp:19-55); scipy.stats.hypergeom.pmf(k, Mtot, n, N) in GSL argument order
        hyp1.assign((size_t)(n1 + 1) * (n1 + n2 + 1), 0.0);
        for (int nseg = 0; nseg <= n1 + n2; ++nseg)
            for (int np1 = std::max(nseg - n2, 0); np1 <= std::min(nseg, n1); ++np1)
                hyp1[(size_t)np1 * (n1 + n2 + 1) + nseg] = OnePopPrep::hypergeom_pdf(np1, nseg, n1 + n2 - nseg, n1);
        hyp2.assign((size_t)(n1 + 2) * (n1 + n2), 0.0);
        for (int nseg = 1; nseg <= n1 + n2; ++nseg)
            for (int np1 = std::max(nseg - n2, 0); np1 <= std::min(nseg, n1 + 1); ++np1)
                hyp2[(size_t)np1 * (n1 + n2) + nseg - 1] = OnePopPrep::hypergeom_pdf(np1, nseg, n1 + n2 + 1 - nseg, n1 + 1);
    }

    int tensor_size() const { return (a1 + 1) * d0; }

    // per hidden state the tensor [(a1+1), (n1+1), (a2+1), (n2+1)] flattened row-major (jcsfs.h:83-87)
    std::vector<std::vector<S>> compute(const ModelParamsT<S> &p1, const ModelParamsT<S> &p2, double split_) {
        params1 = p1; params2 = p2; split = split_;
        J.assign(M, std::vector<S>((size_t)tensor_size(), S(0.0)));
        if (a1 == 1) apart();
        else together();
        for (int m = 0; m < M; ++m) {
            for (S &x : J[m])
                if (!(sval(x) > 1e-20)) x = S(1e-20);                 // floor; the derivative of a floored entry is zero
            at(m, 0, 0, 0, 0) = S(0.0);                                // non-segregating configurations carry no mass
            at(m, a1, n1, a2, n2) = S(0.0);
        }
        return std::move(J);                                           // (J is rebuilt by the next compute())
    }

private:
    S &at(int m, int i, int j, int k, int l) { return J[m][(size_t)i * d0 + (size_t)j * d1 + (size_t)k * d2 + l]; }
    std::vector<std::vector<S>> csfs_of(int n, const RateFunctionT<S> &eta, bool below_only = false) const {
        return conditioned_sfs<S>(eta, *csfs_tables(n), below_only);
    }
    double h1(int np1, int nseg) const { return hyp1[(size_t)np1 * (n1 + n2 + 1) + nseg]; }
    double h2(int np1, int nseg) const { return hyp2[(size_t)np1 * (n1 + n2) + nseg - 1]; }

    // ---- both distinguished lineages in population 1 (jcsfs.cpp:371-420) ----
    void together() {
        const bool tm_ = opt().has(smcpp_opt::O_HOST_TIMING);
        const double tw0 = tm_ ? omp_get_wtime() : 0.0;
        double tw_region = 0.0, tw_single = 0.0, tw_collected = 0.0, tw_sec[4] = {0, 0, 0, 0};
        eta1.reset(new RateFunctionT<S>(params1, std::vector<double>{split - 1e-6, split + 1e-6}));
        const RateFunctionT<S> eta2(params2, std::vector<double>());
        Rts1 = eta1->R(split);
        Rts2 = eta2.R(split);
        eMn1[0] = Mn10.expM(Rts1);
        eMn1[1] = Mn11.expM(Rts1);
        eMn1[2].assign(eMn1[0].rbegin(), eMn1[0].rend());              // rows and columns reversed
        eMn2 = Mn2.expM(Rts2);
        // ONE parallel region for everything below (libomp's workers sleep between regions - every region of its own costs a
        // wake-up that is longer than the work of a batch here):
        //  (A) the conditioned SFS of EVERY interval below the split (truncated model of population 1) and above it (shifted model,
        //      n1 + n2 lineages) as two batches whose hidden states are shared out over the team - the reference, and rounds 1-4 here,
        //      built a rate function and the piece tables of the factored CSFS per hidden state (jcsfs.cpp:91-93,170-172); the
        //      intervals are consecutive, so they are the hidden states of ONE rate function whose tables are built once;
        //  (B) the pieces that do not depend on the hidden state (the reference recomputes them per state), one per thread: the folded
        //      truncated SFS of population 2, the SFS above the split with its contraction Cb, the contraction Da, the below-part at
        //      the split.  The two quadruple loops of the reference (jcsfs.cpp:141-160 and 181-200) sum, per hidden state, products
        //      in which only ONE factor depends on the state; the state-independent contractions
        //        Cb[np1][b2]         = sum_nseg sfs_above[nseg-1] h2(np1, nseg) eMn2[nseg-np1][b2]
        //        Da[i][nseg][b1][b2] = sum_np1  eMn1[i][np1][b1] eMn2[nseg-np1][b2] h1(np1, nseg)
        //      are formed once (all terms are non-negative: only the order of the sums changes);
        //  (C) the hidden states, one task each.
        const int c1 = n1 + 1, c2 = n2 + 1;
        bool any_below = false, any_above = false;
        for (int m = 0; m < M; ++m) { any_below = any_below || hs[m] < split; any_above = any_above || hs[m + 1] > split; }
        below_of.assign(M, -1); above_of.assign(M, -1);
        std::vector<double> hb, ha;
        for (int m = 0; m < M; ++m) {
            const double t1 = hs[m], t2 = hs[m + 1];
            if (t1 < split) { below_of[m] = (int)hb.size(); hb.push_back(t1); }
            if (t2 > split) { above_of[m] = (int)ha.size(); ha.push_back(std::max(t1, split) - split); }
        }
        std::unique_ptr<RateFunctionT<S>> eta_trunc_all, eta_shift_all;
        CsfsJob<S> job_b, job_a;
        bool team_b = false, team_a = false, dev_b = false, dev_a = false;
        if (!hb.empty()) {
            hb.push_back(std::min(split, hs[M]));
            eta_trunc_all.reset(new RateFunctionT<S>(truncate_params(params1, split), hb));
            job_b.eta = eta_trunc_all.get();
            team_b = job_b.factored();
            dev_b = team_b && CsfsBatchHook<S>::launch(batch_dev, 0, *eta_trunc_all, n1);      // (enqueued: runs beside the host work below)
            if (dev_b) team_b = false;
            if (team_b) job_b.init(*eta_trunc_all, *csfs_tables(n1), false);
        }
        if (!ha.empty()) {
            ha.push_back(INFINITY);
            eta_shift_all.reset(new RateFunctionT<S>(shift_params(params1, split), ha));
            job_a.eta = eta_shift_all.get();
            team_a = job_a.factored();
            dev_a = team_a && CsfsBatchHook<S>::launch(batch_dev, 1, *eta_shift_all, n1 + n2);
            if (dev_a) team_a = false;
            if (team_a) job_a.init(*eta_shift_all, *csfs_tables(n1 + n2), false);
        }
        eta_plain.reset(new RateFunctionT<S>(params1, std::vector<double>()));
        std::vector<S> r2;
        Cb_.assign((size_t)(n1 + 2) * c2, S(0.0));
        Da_.assign((size_t)3 * (n1 + n2 + 1) * c1 * c2, S(0.0));
        trunc_all.clear(); rsfs_all.clear();
        const int nd = dual_nder();
        std::string err;
        std::atomic<bool> failed{false};
        auto guarded = [&](auto &&fn) {
            try { fn(); } catch (const std::exception &ex) {
                failed = true;
#pragma omp critical
                err = ex.what();
            }
        };
        if (tm_) tw_region = omp_get_wtime();
#pragma omp parallel
        {
            DualScope sc(nd);
            // ---- (A) ----
            if (team_b) conditioned_sfs_team<S>(job_b);
            if (team_a) conditioned_sfs_team<S>(job_a);
            // ---- (B) ----
#pragma omp sections
            {
#pragma omp section
                guarded([&] {
                    const double ts_ = tm_ ? omp_get_wtime() : 0.0;
                    if (n2 > 1) {
                        const RateFunctionT<S> eta2_trunc(truncate_params(params2, split), std::vector<double>{0.0, INFINITY});
                        r2 = undistinguished_sfs(csfs_of(n2 - 2, eta2_trunc)[0], n2 - 2);
                    }
                    if (tm_) tw_sec[0] = omp_get_wtime() - ts_;
                });
#pragma omp section
                guarded([&] {
                    const double ts_ = tm_ ? omp_get_wtime() : 0.0;
                    if (!any_below) return;
                    const RateFunctionT<S> eta1_shift(shift_params(params1, split), std::vector<double>{0.0, INFINITY});
                    sfs_above_split = undistinguished_sfs(csfs_of(n1 + n2 - 1, eta1_shift)[0], n1 + n2 - 1);
                    for (int nseg = 1; nseg <= n1 + n2; ++nseg)
                        for (int np1 = std::max(nseg - n2, 0); np1 <= std::min(nseg, n1 + 1); ++np1) {
                            const S f = sfs_above_split[nseg - 1] * h2(np1, nseg);
                            for (int b2 = 0; b2 <= n2; ++b2) Cb_[(size_t)np1 * c2 + b2] += f * eMn2[(size_t)(nseg - np1) * c2 + b2];
                        }
                    if (tm_) tw_sec[1] = omp_get_wtime() - ts_;
                });
#pragma omp section
                guarded([&] {
                    const double ts_ = tm_ ? omp_get_wtime() : 0.0;
                    if (!any_above) return;
                    for (int i = 0; i < 3; ++i)
                        for (int nseg = 0; nseg <= n1 + n2; ++nseg)
                            for (int np1 = std::max(nseg - n2, 0); np1 <= std::min(nseg, n1); ++np1) {
                                const double h = h1(np1, nseg);
                                S *dst = &Da_[((size_t)i * (n1 + n2 + 1) + nseg) * c1 * c2];
                                for (int b1 = 0; b1 <= n1; ++b1) {
                                    const S f = eMn1[i][(size_t)np1 * c1 + b1] * h;
                                    for (int b2 = 0; b2 <= n2; ++b2) dst[(size_t)b1 * c2 + b2] += f * eMn2[(size_t)(nseg - np1) * c2 + b2];
                                }
                            }
                    if (tm_) tw_sec[2] = omp_get_wtime() - ts_;
                });
#pragma omp section
                guarded([&] {
                    const double ts_ = tm_ ? omp_get_wtime() : 0.0;
                    if (any_above) below_at_split = csfs_of(n1, *eta1, true)[0];          // (the same for every state above the split)
                    // (a batch whose model the factored evaluation cannot take - a zero rate - goes through the generic routine)
                    if (!hb.empty() && !team_b && !dev_b) trunc_all = csfs_of(n1, *eta_trunc_all);
                    if (!ha.empty() && !team_a && !dev_a) rsfs_all = csfs_of(n1 + n2, *eta_shift_all);
                    if (tm_) tw_sec[3] = omp_get_wtime() - ts_;
                });
            }
#pragma omp single
            {
                if (tm_) tw_single = omp_get_wtime();
                if (team_b) trunc_all.swap(job_b.csfs);
                if (team_a) rsfs_all.swap(job_a.csfs);
                if (dev_b) guarded([&] { CsfsBatchHook<S>::collect(batch_dev, 0, trunc_all); });
                if (dev_a) guarded([&] { CsfsBatchHook<S>::collect(batch_dev, 1, rsfs_all); });
                if (tm_) tw_collected = omp_get_wtime();
            }
            // ---- (C) ----
#pragma omp for schedule(dynamic)
            for (int m = 0; m < M; ++m) {
                if (failed) continue;
                try {
                    const double t1 = hs[m], t2 = hs[m + 1];
                    if (t1 < t2 && t2 <= split) tau_below_split(m, t1, t2, S(1.0));
                    else if (split <= t1 && t1 < t2) tau_above_split(m, t1, t2, S(1.0));
                    else {
                        const S e1 = m_exp(-eta1->R(t1));
                        const S e2 = std::isinf(t2) ? S(0.0) : m_exp(-eta1->R(t2));
                        const S w = (m_exp(-Rts1) - e2) / (e1 - e2);
                        tau_below_split(m, t1, split, 1.0 - w);
                        tau_above_split(m, split, t2, w);
                    }
                    // population 2 below the split: no distinguished lineage there, so its private branches are the folded
                    // truncated SFS
                    if (n2 == 1) at(m, 0, 0, 0, 1) += split;
                    if (n2 > 1) {
                        S remain(0.0);
                        for (int i = 0; i < n2 - 1; ++i) {
                            at(m, 0, 0, 0, i + 1) += r2[i];
                            remain += r2[i] * ((double)(i + 1) / (double)n2);
                        }
                        remain -= S(split);
                        at(m, 0, 0, 0, n2) -= remain;
                    }
                } catch (const std::exception &ex) {
                    failed = true;
#pragma omp critical
                    err = ex.what();
                }
            }
        }
        if (tm_) {
            const double t1 = omp_get_wtime();
            fprintf(stderr, "[jcsfs] serial head %.1f us (rate functions, expM, device batches enqueued), batches + sections %.1f us, collect %.1f us, "
                    "states %.1f us; sections: r2 %.1f, sfs above + Cb %.1f, Da %.1f, below at split / fallbacks %.1f us\n", 1e6 * (tw_region - tw0), 1e6 * (tw_single - tw_region), 1e6 * (tw_collected - tw_single), 1e6 * (t1 - tw_collected), 1e6 * tw_sec[0], 1e6 * tw_sec[1], 1e6 * tw_sec[2], 1e6 * tw_sec[3]);
        }
        if (job_b.side_err) std::rethrow_exception(job_b.side_err);
        if (job_a.side_err) std::rethrow_exception(job_a.side_err);
        if (!err.empty()) throw std::runtime_error(err);
    }

    // distinguished pair coalesces in [t1, t2) with t2 <= split (jcsfs.cpp:83-163)
    void tau_below_split(int m, double t1, double t2, const S &weight) {
        const RateFunctionT<S> &eta = *eta_plain;
        const std::vector<S> &trunc = trunc_all[below_of[m]];          // CSFS of [t1, t2) under the model truncated at the split
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j <= n1; ++j)
                if (sval(trunc[(size_t)i * (n1 + 1) + j]) > 0) at(m, i, j, 0, 0) = weight * trunc[(size_t)i * (n1 + 1) + j];
        const std::vector<S> tsfs = undistinguished_sfs(trunc, n1);
        S Et(0.0);
        for (int k = 0; k <= n1; ++k) Et += tsfs[k] * ((double)(k + 1) / (double)(n1 + 2));
        at(m, 2, n1, 0, 0) = (split - Et) * weight;
        // above the split: SFS of the n1 + n2 + 1 lineages there (together()), carried down through the Moran models:
        //   avg0 = 1/K sum_k (A_k diag(s0)).leftCols(c) B_k,  avg2 = 1/K sum_k (A_k diag(s2)).rightCols(c) C_k,
        // A_k = expM_{n1+1}(Rts1 - R(t_k)), B_k / C_k = expM of the modified models at R(t_k) (jcsfs.cpp:120-138).  With
        // A = P1 e^{.d1} P1^-1 etc. the sum over k only touches the eigenvalue powers:
        //   avg0 = P1 [ (1/K sum_k u_k v_k^T) o W0 ] P0^-1,   u_k = e^{(Rts1 - R_k) d1},  v_k = e^{R_k d0}      (W0: constructor)
        const int r = n1 + 2, c = n1 + 1;
        std::vector<S> G0((size_t)r * c, S(0.0)), G2((size_t)r * c, S(0.0)), u(r), v(c), w(c);
        std::mt19937 gen;                                               // default seed, re-created per call (quirk 14)
        for (int k = 0; k < K; ++k) {
            const S t = eta.random_time(t1, t2, gen);
            const S Rt = eta.R_at(t);
            const S dR = Rts1 - Rt;
            for (int x = 0; x < r; ++x) u[x] = m_exp(dR * Mn1p1.es.d[x]);
            for (int y = 0; y < c; ++y) { v[y] = m_exp(Rt * Mn10.es.d[y]); w[y] = m_exp(Rt * Mn12.es.d[y]); }
            for (int x = 0; x < r; ++x)
                for (int y = 0; y < c; ++y) { G0[(size_t)x * c + y] += u[x] * v[y]; G2[(size_t)x * c + y] += u[x] * w[y]; }
        }
        for (size_t i = 0; i < G0.size(); ++i) { G0[i] *= W0_[i] / (double)K; G2[i] *= W2_[i] / (double)K; }
        // H = G Pinv (r x c), then avg = P1 H (r x c)
        std::vector<S> H0((size_t)r * c, S(0.0)), H2((size_t)r * c, S(0.0));
        for (int x = 0; x < r; ++x)
            for (int y = 0; y < c; ++y) {
                const S g0 = G0[(size_t)x * c + y], g2 = G2[(size_t)x * c + y];
                for (int j = 0; j < c; ++j) {
                    H0[(size_t)x * c + j] += g0 * Mn10.es.Pinv[(size_t)y * c + j];
                    H2[(size_t)x * c + j] += g2 * Mn12.es.Pinv[(size_t)y * c + j];
                }
            }
        std::vector<S> avg0((size_t)r * c, S(0.0)), avg2((size_t)r * c, S(0.0));
        for (int i = 0; i < r; ++i)
            for (int x = 0; x < r; ++x) {
                const double p = Mn1p1.es.P[(size_t)i * r + x];
                for (int j = 0; j < c; ++j) {
                    avg0[(size_t)i * c + j] += p * H0[(size_t)x * c + j];
                    avg2[(size_t)i * c + j] += p * H2[(size_t)x * c + j];
                }
            }
        for (int b1 = 0; b1 <= n1; ++b1)
            for (int b2 = 0; b2 <= n2; ++b2) {
                S s0(0.0), s2(0.0);
                for (int np1 = 0; np1 <= n1 + 1; ++np1) {
                    const S &cb = Cb_[(size_t)np1 * (n2 + 1) + b2];
                    s0 += cb * avg0[(size_t)np1 * c + b1];
                    s2 += cb * avg2[(size_t)np1 * c + b1];
                }
                at(m, 0, b1, 0, b2) += weight * s0;
                at(m, 2, b1, 0, b2) += weight * s2;
            }
    }

    // distinguished pair coalesces in [t1, t2) with split <= t1 (jcsfs.cpp:166-216)
    void tau_above_split(int m, double t1, double t2, const S &weight) {
        (void)t1; (void)t2;
        const std::vector<S> &rsfs = rsfs_all[above_of[m]];             // 3 x (n1+n2+1): CSFS of [t1, t2) - split under the shifted model
        const int w = n1 + n2 + 1;
        {
            const int c1 = n1 + 1, c2 = n2 + 1;
            for (int i = 0; i < 3; ++i)
                for (int nseg = 0; nseg <= n1 + n2; ++nseg) {
                    const S f = rsfs[(size_t)i * w + nseg] * weight;
                    const S *src = &Da_[((size_t)i * w + nseg) * c1 * c2];
                    for (int b1 = 0; b1 <= n1; ++b1)
                        for (int b2 = 0; b2 <= n2; ++b2) at(m, i, b1, 0, b2) += f * src[(size_t)b1 * c2 + b2];
                }
        }
        // population 1 below the split: the below-part of a CSFS conditioned on coalescence right at the split
        const std::vector<S> &below = below_at_split;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j <= n1; ++j)
                if (sval(below[(size_t)i * (n1 + 1) + j]) > 0) at(m, i, j, 0, 0) += weight * below[(size_t)i * (n1 + 1) + j];
    }

    // ---- one distinguished lineage in each population (jcsfs.cpp:256-366) ----
    void apart() {
        std::vector<double> times{0.0};
        for (int m = 1; m < M; ++m)
            if (hs[m] > split) times.push_back(hs[m] - split);
        times.push_back(INFINITY);
        const RateFunctionT<S> shifted(shift_params(params1, split), times);
        const std::vector<std::vector<S>> at_split = csfs_of(n1 + n2, shifted);
        const S R1 = RateFunctionT<S>(params1, std::vector<double>()).R(split);
        const S R2 = RateFunctionT<S>(params2, std::vector<double>()).R(split);
        const std::vector<S> T10 = Mn10.expM(R1), T11 = Mn11.expM(R1), T20 = Mn20.expM(R2), T21 = Mn21.expM(R2);
        const int w = n1 + n2 + 1;
        int i = 0;
        for (int m = 0; m < M; ++m) {
            if (hs[m + 1] <= split) continue;       // the two lineages cannot meet below the split
            const std::vector<S> &cs = at_split[i++];
            for (int b1 = 0; b1 <= n1; ++b1)
                for (int b2 = 0; b2 <= n2; ++b2)
                    for (int nseg = 0; nseg <= n1 + n2; ++nseg)
                        for (int np1 = std::max(nseg - n2, 0); np1 <= std::min(nseg, n1); ++np1) {
                            const int np2 = nseg - np1;
                            const double h = h1(np1, nseg);
                            const S &t10 = T10[(size_t)np1 * (n1 + 1) + b1], &t11 = T11[(size_t)np1 * (n1 + 1) + b1];
                            const S &t20 = T20[(size_t)np2 * (n2 + 1) + b2], &t21 = T21[(size_t)np2 * (n2 + 1) + b2];
                            at(m, 1, b1, 1, b2) += h * cs[(size_t)2 * w + nseg] * t11 * t21;
                            at(m, 1, b1, 0, b2) += 0.5 * h * cs[(size_t)1 * w + nseg] * t11 * t20;
                            at(m, 0, b1, 1, b2) += 0.5 * h * cs[(size_t)1 * w + nseg] * t10 * t21;
                            at(m, 0, b1, 0, b2) += h * cs[(size_t)0 * w + nseg] * t10 * t20;
                        }
        }
        if (split == 0.0) return;
        // private branches of each population below the split, the same for every hidden state
        for (int pop = 0; pop < 2; ++pop) {
            const ModelParamsT<S> &pp = pop == 0 ? params1 : params2;
            const int ni = pop == 0 ? n1 : n2;
            const RateFunctionT<S> eta_trunc(truncate_params(pp, split), std::vector<double>{0.0, INFINITY});
            std::vector<S> r;
            if (ni > 0) r = undistinguished_sfs(csfs_of(ni - 1, eta_trunc)[0], ni - 1);
            for (int k = 1; k <= ni; ++k) {
                const double fac = (double)k / (double)(ni + 1);
                const S x1 = (1.0 - fac) * r[k - 1], x2 = fac * r[k - 1];
                for (int m = 0; m < M; ++m) {
                    if (pop == 0) { at(m, 0, k, 0, 0) += x1; at(m, 1, k - 1, 0, 0) += x2; }
                    else { at(m, 0, 0, 0, k) += x1; at(m, 0, 0, 1, k - 1) += x2; }
                }
            }
            S remain(0.0);
            for (int k = 1; k <= ni; ++k) remain += r[k - 1] * (double)k;
            remain /= (double)(ni + 1);
            remain -= S(split);
            for (int m = 0; m < M; ++m) {
                if (pop == 0) at(m, 1, ni, 0, 0) -= remain;
                else at(m, 0, 0, 1, ni) -= remain;
            }
        }
    }

    const int n1, n2, a1, a2;
    const std::vector<double> hs;
    const int M, K;
    int d0, d1, d2;
    MoranExp Mn1p1, Mn2, Mn10, Mn11, Mn12, Mn20, Mn21;
    std::vector<double> hyp1, hyp2;
    // state of one compute()
    ModelParamsT<S> params1, params2;
    double split = 0.0;
    std::unique_ptr<RateFunctionT<S>> eta1;
    S Rts1, Rts2;
    std::array<std::vector<S>, 3> eMn1;
    std::vector<S> eMn2, sfs_above_split;
    std::vector<S> Cb_;      // [(n1+2)][(n2+1)]  state-independent contraction of tau_below_split (see together())
    std::vector<S> Da_;      // [3][(n1+n2+1)][(n1+1)][(n2+1)]  ... of tau_above_split
    std::vector<std::vector<S>> J;
    std::vector<double> W0_, W2_;                       // [(n1+2)][(n1+1)] static sandwiches of tau_below_split (constructor)
    std::vector<int> below_of, above_of;                // hidden state -> its interval in trunc_all / rsfs_all (-1: none)
    std::vector<std::vector<S>> trunc_all, rsfs_all;    // batched conditioned SFS below / above the split (together())
    std::vector<S> below_at_split;
    std::unique_ptr<RateFunctionT<S>> eta_plain;
};

// ---------------------------------------------------------------------------------------------------------------
// two-population preparation: pi / transition from the distinguished model, emission table from the joint CSFS
// (TwoPopInferenceManager::setParams inference_manager.cpp:542-550 + the generic-P do_dirty_work)
// ---------------------------------------------------------------------------------------------------------------
class TwoPopPrep {
public:
    TwoPopPrep(int n1, int n2, int a1, int a2, const std::vector<double> &hs, double polarization_error, int K = 10)
        : hs_(hs), pol_(polarization_error), K_(K) {
        if (a1 + a2 != 2) throw std::runtime_error("configuration not supported");
        if (a1 == 0 && a2 == 2) throw std::runtime_error("(0,2) not supported");
        n_[0] = n1; n_[1] = n2; na_[0] = a1; na_[1] = a2;
    }

    typedef std::array<int, 6> Key;

    // raw joint CSFS per hidden state (what `joint_csfs` of smcpp/_smcpp.pyx:416-437 returns)
    template <typename S>
    std::vector<std::vector<S>> jcsfs(const ModelParamsT<S> &p1, const ModelParamsT<S> &p2, double split) const {
        // (the object is kept: its constructor takes five Moran eigensystems and the static sandwiches W0 / W2 - 0.1 ms that depend
        // on the sample sizes alone; compute() resets everything that depends on the parameters)
        JointCsfsT<S> &j = joint<S>();
        j.batch_dev = batch_dev;
        return j.compute(p1, p2, split);
    }
    template <typename S> JointCsfsT<S> &joint() const {
        if constexpr (std::is_same<S, double>::value) {
            if (!joint_d_) joint_d_.reset(new JointCsfsT<double>(n_[0], n_[1], na_[0], na_[1], hs_, K_));
            return *joint_d_;
        } else {
            if (!joint_x_) joint_x_.reset(new JointCsfsT<S>(n_[0], n_[1], na_[0], na_[1], hs_, K_));
            return *joint_x_;
        }
    }
    mutable std::unique_ptr<JointCsfsT<double>> joint_d_;
    mutable std::unique_ptr<JointCsfsT<dual>> joint_x_;
    mutable CsfsBatchDevice *batch_dev = nullptr;   // the engine's device route of the batched conditioned SFS (values only), not owned

    // keys [K][6]; outputs pi [M], T [M*M], E [K*M]
    template <typename S>
    void compute_t(const ModelParamsT<S> &dist, const ModelParamsT<S> &p1, const ModelParamsT<S> &p2, double split,
                   double theta, double rho, double alpha, const std::vector<int> &keys, int K, std::vector<S> &pi,
                   std::vector<S> &T, std::vector<S> &E, std::vector<S> *emission_out = nullptr) const {
        RateFunctionT<S> eta(dist, hs_);
        const int M = (int)hs_.size() - 1;
        pi.assign(M, S(0.0));
        for (int m = 0; m < M - 1; ++m) pi[m] = m_exp(-eta.R(hs_[m])) - m_exp(-eta.R(hs_[m + 1]));
        pi[M - 1] = m_exp(-eta.R(hs_[M - 1]));
        S ps(0.0);
        for (S &x : pi) { if (sval(x) < 1e-20) x = S(1e-20); ps += x; }
        for (S &x : pi) x /= ps;
        const bool tm = opt().has(smcpp_opt::O_HOST_TIMING);
        const auto tc0 = std::chrono::steady_clock::now();
        T = compute_transition<S>(eta, rho);
        const auto tc1 = std::chrono::steady_clock::now();
        std::vector<std::vector<S>> sfs = jcsfs<S>(p1, p2, split);
        const auto tc2 = std::chrono::steady_clock::now();
        incorporate_theta<S>(sfs, theta);
        if (emission_out) {
            emission_out->clear();
            for (const auto &c : sfs) emission_out->insert(emission_out->end(), c.begin(), c.end());
        }
        const std::vector<S> avg_ct = eta.average_coal_times();
        std::vector<S> e2((size_t)M * 2, S(0.0));
        for (int m = 0; m < M; ++m) {
            if (std::isnan((double)sval(avg_ct[m]))) { e2[2 * m] = S(1e-20); e2[2 * m + 1] = S(1e-20); }
            else {
                const S le = -2.0 * alpha * theta * avg_ct[m];
                e2[2 * m] = m_exp(le);
                e2[2 * m + 1] = -m_expm1(le);
            }
        }
        const int d2 = n_[1] + 1, d1 = (na_[1] + 1) * d2, d0 = (n_[0] + 1) * d1;
        E.assign((size_t)K * M, S(0.0));
        // the bins of a key (tensor index, weight) depend on the key only: built once per key dictionary, not per E-step
        if (bins_keys_.size() != (size_t)6 * K || !std::equal(bins_keys_.begin(), bins_keys_.end(), keys.begin())) {
            bins_keys_.assign(keys.begin(), keys.begin() + (size_t)6 * K);
            bins_cache_.assign(K, {});
            for (int k = 0; k < K; ++k) {
                Key bk;
                for (int q = 0; q < 6; ++q) bk[q] = keys[(size_t)6 * k + q];
                for (const auto &p : bins_for(bk))
                    bins_cache_[k].emplace_back((size_t)p.first[0] * d0 + (size_t)p.first[1] * d1 + (size_t)p.first[2] * d2 + p.first[3],
                                                p.second);
            }
        }
        // keys are independent (192 of them on config C4, a few hundred tensor bins each): one task per key; the table is read
        // bin by bin for all states, so it is transposed once ([bin][state]: the inner loop runs over contiguous memory)
        const size_t nbin_ = sfs.empty() ? 0 : sfs[0].size();
        std::vector<S> sfsT(nbin_ * (size_t)M);
        for (int m = 0; m < M; ++m)
            for (size_t b = 0; b < nbin_; ++b) sfsT[b * M + m] = sfs[m][b];
        const int nd_ = dual_nder();
        bool bad_ = false;
#pragma omp parallel for schedule(static)
        for (int k = 0; k < K; ++k) {
            DualScope sc_(nd_);
            Key bk;
            for (int q = 0; q < 6; ++q) bk[q] = keys[(size_t)6 * k + q];
            bool reduced = true, miss = true;
            int amin = 1 << 30, asum = 0;
            for (int p = 0; p < 2; ++p) {
                reduced &= bk[3 * p + 2] == 0;
                if (na_[p] > 0) miss &= bk[3 * p] == -1;
                amin = std::min(amin, bk[3 * p]);
                asum += bk[3 * p];
            }
            S *e = &E[(size_t)k * M];
            if (reduced && (miss || amin >= 0)) {
                for (int m = 0; m < M; ++m) e[m] = miss ? S(1.0) : e2[2 * m + (asum % 2)];
            } else {
                for (const auto &p : bins_cache_[k]) {
                    const S *col = &sfsT[p.first * (size_t)M];
                    for (int m = 0; m < M; ++m) e[m] += p.second * col[m];
                }
            }
            double mx = sval(e[0]), mn = sval(e[0]);
            for (int m = 1; m < M; ++m) { mx = std::max(mx, (double)sval(e[m])); mn = std::min(mn, (double)sval(e[m])); }
            if (mx > 1.0 || mn <= 0.0) {
#pragma omp atomic write
                bad_ = true;
            }
        }
        if (bad_) throw std::runtime_error("probability vector not in [0, 1]");
        if (tm) {
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "[prep2] transition %.3f ms, joint csfs %.3f ms, theta + emission assembly (%d keys) %.3f ms\n", ms(tc0, tc1),
                    ms(tc1, tc2), K, ms(tc2, std::chrono::steady_clock::now()));
        }
    }

    // construct_bins for one observed key (inference_manager.cpp:329-386 with P = 2): weights over (a1, b1, a2, b2)
    mutable std::vector<int> bins_keys_;                                         // key dictionary the cache below belongs to
    mutable std::vector<std::vector<std::pair<size_t, double>>> bins_cache_;     // per key: (flattened tensor index, weight)

    std::map<std::array<int, 4>, double> bins_for(const Key &bk) const {
        auto is_mono = [&](const Key &k) {
            for (int p = 0; p < 2; ++p)
                if (k[3 * p] != na_[p] || k[3 * p + 1] != k[3 * p + 2]) return false;
            return true;
        };
        // bin_key<2> with cutoff 1.0: a == -1 expands over 0..na(p); b/nb > 1 never holds
        std::vector<std::array<int, 3>> side[2];
        for (int p = 0; p < 2; ++p) {
            const int a = bk[3 * p], b = bk[3 * p + 1], nb = bk[3 * p + 2];
            if (a == -1) for (int aa = 0; aa <= na_[p]; ++aa) side[p].push_back({aa, b, nb});
            else side[p].push_back({a, b, nb});
        }
        std::map<Key, double> m;
        for (const auto &k0 : side[0])
            for (const auto &k1 : side[1]) {
                // marginalize_key<2>: lift each population from nb to n observed, independent hypergeometric weights
                std::vector<std::pair<std::array<int, 3>, double>> lift[2];
                const std::array<int, 3> kk[2] = {k0, k1};
                for (int p = 0; p < 2; ++p) {
                    std::map<std::array<int, 3>, double> acc;
                    for (int x = kk[p][1]; x <= n_[p] + kk[p][1] - kk[p][2]; ++x)
                        acc[{kk[p][0], x, n_[p]}] += OnePopPrep::hypergeom_pdf(kk[p][1], x, n_[p] - x, kk[p][2]);
                    lift[p].assign(acc.begin(), acc.end());
                }
                for (const auto &l0 : lift[0])
                    for (const auto &l1 : lift[1]) {
                        Key mbk{l0.first[0], l0.first[1], l0.first[2], l1.first[0], l1.first[1], l1.first[2]};
                        const double pr = l0.second * l1.second;
                        if (is_mono(mbk)) mbk = Key{0, 0, mbk[2], 0, 0, mbk[5]};
                        m[mbk] += (1.0 - pol_) * pr;
                        Key fk;
                        for (int p = 0; p < 2; ++p) {
                            fk[3 * p] = na_[p] - mbk[3 * p];
                            fk[3 * p + 1] = mbk[3 * p + 2] - mbk[3 * p + 1];
                            fk[3 * p + 2] = mbk[3 * p + 2];
                        }
                        m[fk] += pol_ * pr;
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
        std::map<std::array<int, 4>, double> out;
        for (const auto &p : m2) out[{p.first[0], p.first[1], p.first[3], p.first[4]}] += p.second / s;
        return out;
    }

private:
    int n_[2], na_[2];
    std::vector<double> hs_;
    double pol_;
    int K_;
};

}  // namespace smcpp_host
