// TEST INFRASTRUCTURE ONLY — never linked into the product (smcpp_amd/).
//
// C-ABI harness around the REAL reference sources, compiled where they lie
// under /root/reference (see oracle/Makefile; outputs go to oracle/_ref/).
// It drives the reference's own
//   HMM::HMM / HMM::Estep / HMM::Q           (src/hmm.cpp:8-193)
//   TransitionBundle::update                 (src/transition_bundle.cpp:3-61)
//   compute_transition                       (src/transition.cpp:256-262)
//   PiecewiseConstantRateFunction::R / average_coal_times
//                                            (src/piecewise_constant_rate_function.cpp:372-420)
//   OnePopConditionedSFS::compute, incorporate_theta
//                                            (src/conditioned_sfs.cpp:86-148)
// directly, bypassing InferenceManager (whose translation unit needs GSL, which
// this image does not have — so inference_manager.cpp / jcsfs.cpp are NOT built
// and nothing is stubbed).  The only thing this file adds is glue: building
// the InferenceBundle the reference's InferenceManager ctor would build
// (inference_manager.cpp:21-54,232-254) and copying private HMM state out.
//
// `#define private public` is confined to the include of hmm.h in this TU so the harness can read
// HMM::alpha_hat / log_c / xisum / gamma / gamma_sums (hmm.h:29-37).

#include <vector>
#include <map>
#include <set>
#include <cstring>
#include <string>
#include <stdexcept>

#include "common.h"
#include "block_key.h"
#include "transition_bundle.h"
#include "inference_bundle.h"
// every header hmm.h pulls in is already included above, so the macro only touches class HMM
#define private public
#include "hmm.h"
#undef private
#include "piecewise_constant_rate_function.h"
#include "transition.h"
#include "conditioned_sfs.h"

static thread_local std::string g_err;

static block_key make_key(const int *p, int keylen)
{
    Vector<int> v(keylen);
    for (int i = 0; i < keylen; ++i) v(i) = p[i];
    return block_key(v);
}

extern "C" const char *ref_last_error() { return g_err.c_str(); }

// One contig, raw parameters in, every E-step product out.
//   keys  [K x keylen] int, E [K x M] double (emission vector per key)
//   pi [M], T [M x M] row-major
//   obs [L x (1+keylen)] int32 row-major (reference layout, inference_manager.cpp:180-188)
// Outputs (any pointer may be NULL):
//   loglik[1]; xisum [M x M] row-major; gamma [M x (L+1)] row-major if save_gamma else [M];
//   gs_nkeys[1], gs_keys [<=K x keylen], gs_vals [<=K x M]   (std::map order)
//   alpha_hat [(L+1) x M] float (column ell of the reference matrix = row ell here)
//   log_c [L+1]; q [4] (values of HMM::Q, hmm.cpp:155-193)
//   eig_* : eigensystem of key index eig_key (if >= 0): P_r, Pinv_r [M x M row-major], d_r [M], scale[1], max |imag|[1]
extern "C" int ref_estep(int M, int K, int keylen, const int *keys, const double *E,
                         const double *pi_in, const double *T_in,
                         int L, const int *obs_in, int save_gamma,
                         double *loglik, double *xisum, double *gamma,
                         int *gs_nkeys, int *gs_keys, double *gs_vals,
                         float *alpha_hat, double *log_c, double *q,
                         int eig_key, double *eig_P, double *eig_Pinv, double *eig_d,
                         double *eig_scale, double *eig_maximag)
{
    try
    {
        Vector<adouble> pi(M);
        for (int m = 0; m < M; ++m) pi(m) = adouble(pi_in[m]);
        Matrix<adouble> T(M, M);
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < M; ++j)
                T(i, j) = adouble(T_in[i * M + j]);
        std::map<block_key, Vector<adouble> > emission_probs;
        for (int k = 0; k < K; ++k)
        {
            Vector<adouble> e(M);
            for (int m = 0; m < M; ++m) e(m) = adouble(E[k * M + m]);
            emission_probs.emplace(make_key(keys + k * keylen, keylen), e);
        }
        const int ncol = 1 + keylen;
        // obs is declared non-const in the reference's Map type; it is never written.
        Eigen::Map<Eigen::Matrix<int, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> > obs(
            const_cast<int *>(obs_in), L, ncol);
        // InferenceManager::fill_targets (inference_manager.cpp:232-254)
        spp::sparse_hash_set<std::pair<int, block_key> > targets;
        for (int i = 0; i < L; ++i)
        {
            if (obs(i, 0) <= 0) throw std::runtime_error("data are malformed: span <= 0");
            if (obs(i, 0) > 1) targets.insert({obs(i, 0), make_key(obs_in + i * ncol + 1, keylen)});
        }
        TransitionBundle tb(targets, &emission_probs);
        bool sg = save_gamma != 0;
        InferenceBundle ib{&pi, &tb, &emission_probs, &sg};
        HMM hmm(0, obs, &ib);
        tb.update(T, true);   // InferenceManager::Estep (inference_manager.cpp:108-114)
        hmm.Estep(false);
        if (loglik) *loglik = hmm.loglik();
        if (xisum)
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j) xisum[i * M + j] = hmm.xisum(i, j);
        if (gamma)
        {
            const int nc = hmm.gamma.cols();
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < nc; ++j) gamma[(size_t)i * nc + j] = hmm.gamma(i, j);
        }
        if (gs_nkeys)
        {
            int n = 0;
            for (auto &p : hmm.gamma_sums)
            {
                if (gs_keys) for (int i = 0; i < keylen; ++i) gs_keys[n * keylen + i] = p.first(i);
                if (gs_vals) for (int m = 0; m < M; ++m) gs_vals[n * M + m] = p.second(m);
                ++n;
            }
            *gs_nkeys = n;
        }
        if (alpha_hat)
            for (int l = 0; l <= L; ++l)
                for (int m = 0; m < M; ++m) alpha_hat[(size_t)l * M + m] = hmm.alpha_hat(m, l);
        if (log_c)
            for (int l = 0; l <= L; ++l) log_c[l] = hmm.log_c(l);
        if (q)
        {
            Vector<adouble> qq = hmm.Q();
            for (int i = 0; i < 4; ++i) q[i] = qq(i).value();
        }
        if (eig_key >= 0)
        {
            block_key bk = make_key(keys + eig_key * keylen, keylen);
            auto it = tb.eigensystems.find(bk);
            if (it == tb.eigensystems.end()) throw std::runtime_error("no eigensystem for requested key");
            const eigensystem &es = it->second;
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j)
                {
                    if (eig_P) eig_P[i * M + j] = es.P_r(i, j);
                    if (eig_Pinv) eig_Pinv[i * M + j] = es.Pinv_r(i, j);
                }
            if (eig_d) for (int i = 0; i < M; ++i) eig_d[i] = es.d_r(i);
            if (eig_scale) *eig_scale = es.scale;
            if (eig_maximag) *eig_maximag = es.d.imag().cwiseAbs().maxCoeff();
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        g_err = e.what();
        return 1;
    }
}

static ParameterVector make_params(int K, const double *a, const double *s)
{
    // mirrors _smcpp.pyx:66-83 make_params with an empty derivative list
    std::vector<adouble> av, sv;
    for (int k = 0; k < K; ++k) { av.push_back(adouble(a[k])); sv.push_back(adouble(s[k])); }
    return {av, sv};
}

// Parameter preparation with the reference's own code.
//   a[Kp], s[Kp]: piecewise-constant size history; hs[M+1] hidden-state boundaries.
// Outputs: pi [M] (inference_manager.cpp:56-69 restated around the reference's eta->R),
//          T [M x M] row-major (transition.cpp:256), avg_ct [M],
//          csfs [M x 3 x (n+1)] after incorporate_theta (conditioned_sfs.cpp:100-148) when n >= 0.
extern "C" int ref_prep(int Kp, const double *a, const double *s, int M, const double *hs,
                        double rho, double theta, int n,
                        double *pi_out, double *T_out, double *avg_ct_out, double *csfs_out,
                        double *raw_csfs_out)
{
    try
    {
        ParameterVector params = make_params(Kp, a, s);
        std::vector<double> hidden_states(hs, hs + M + 1);
        PiecewiseConstantRateFunction<adouble> eta(params, hidden_states);
        if (pi_out)
        {
            Vector<adouble> pi(M);
            for (int m = 0; m < M - 1; ++m)
                pi(m) = exp(-(eta.R(hidden_states.at(m)))) - exp(-(eta.R(hidden_states.at(m + 1))));
            pi(M - 1) = exp(-(eta.R(hidden_states.at(M - 1))));
            adouble small = eta.zero() + 1e-20;
            pi = pi.unaryExpr([small](const adouble &x) { if (x < 1e-20) return small; return x; });
            pi /= pi.sum();
            for (int m = 0; m < M; ++m) pi_out[m] = pi(m).value();
        }
        if (T_out)
        {
            Matrix<adouble> T = compute_transition(eta, rho);
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j) T_out[i * M + j] = T(i, j).value();
        }
        if (avg_ct_out)
        {
            std::vector<adouble> v = eta.average_coal_times();
            for (int m = 0; m < M; ++m) avg_ct_out[m] = v.at(m).value();
        }
        if ((csfs_out || raw_csfs_out) && n >= 0)
        {
            OnePopConditionedSFS<adouble> csfs(n);
            std::vector<Matrix<adouble> > raw = csfs.compute(eta);
            if (raw_csfs_out)
                for (int m = 0; m < M; ++m)
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j <= n; ++j)
                            raw_csfs_out[(m * 3 + i) * (n + 1) + j] = raw.at(m)(i, j).value();
            if (csfs_out)
            {
                std::vector<Matrix<adouble> > v = incorporate_theta(raw, theta);
                for (int m = 0; m < M; ++m)
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j <= n; ++j)
                            csfs_out[(m * 3 + i) * (n + 1) + j] = v.at(m)(i, j).value();
            }
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        g_err = e.what();
        return 1;
    }
}

// Same as ref_prep but with forward-mode derivative seeds on the piece sizes (make_params of _smcpp.pyx:66-83 with a
// non-empty dlist): da [Kp x nder].  Jacobian outputs are [size x nder] row-major.
extern "C" int ref_prep_jac(int Kp, const double *a, const double *da, int nder, const double *s, int M,
                            const double *hs, double rho, double theta, int n,
                            double *pi_out, double *dpi_out, double *T_out, double *dT_out,
                            double *avg_ct_out, double *davg_ct_out, double *csfs_out, double *dcsfs_out)
{
    try
    {
        std::vector<adouble> av, sv;
        for (int k = 0; k < Kp; ++k)
        {
            Eigen::VectorXd d(nder);
            for (int j = 0; j < nder; ++j) d(j) = da[k * nder + j];
            av.push_back(adouble(a[k], d));
            sv.push_back(adouble(s[k]));
        }
        ParameterVector params{av, sv};
        std::vector<double> hidden_states(hs, hs + M + 1);
        PiecewiseConstantRateFunction<adouble> eta(params, hidden_states);
        auto put = [nder](const adouble &x, double *v, double *j, size_t idx) {
            if (v) v[idx] = x.value();
            if (j) for (int d = 0; d < nder; ++d) j[idx * nder + d] = x.derivatives().size() ? x.derivatives()(d) : 0.0;
        };
        {
            Vector<adouble> pi(M);
            for (int m = 0; m < M - 1; ++m)
                pi(m) = exp(-(eta.R(hidden_states.at(m)))) - exp(-(eta.R(hidden_states.at(m + 1))));
            pi(M - 1) = exp(-(eta.R(hidden_states.at(M - 1))));
            adouble small = eta.zero() + 1e-20;
            pi = pi.unaryExpr([small](const adouble &x) { if (x < 1e-20) return small; return x; });
            pi /= pi.sum();
            for (int m = 0; m < M; ++m) put(pi(m), pi_out, dpi_out, m);
        }
        {
            Matrix<adouble> T = compute_transition(eta, rho);
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j) put(T(i, j), T_out, dT_out, (size_t)i * M + j);
        }
        {
            std::vector<adouble> v = eta.average_coal_times();
            for (int m = 0; m < M; ++m) put(v.at(m), avg_ct_out, davg_ct_out, m);
        }
        if (n >= 0 && (csfs_out || dcsfs_out))
        {
            OnePopConditionedSFS<adouble> csfs(n);
            std::vector<Matrix<adouble> > v = incorporate_theta(csfs.compute(eta), theta);
            for (int m = 0; m < M; ++m)
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j <= n; ++j) put(v.at(m)(i, j), csfs_out, dcsfs_out, (size_t)(m * 3 + i) * (n + 1) + j);
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        g_err = e.what();
        return 1;
    }
}

// PyRateFunction.R / random_coal_times and raw_sfs(below_only) of the real reference (piecewise_constant_rate_function.cpp:
// 157-161,337-368; conditioned_sfs.cpp:13-39): R at t[0..nt), one random_time per seed (with R of it), and the
// compute_below part of the conditioned SFS of the single state [t1, t2).
extern "C" int ref_rate(int Kp, const double *a, const double *s, int nt, const double *t, double *R_out,
                        double t1, double t2, int nseeds, const long long *seeds, double *rt_out, double *rR_out,
                        int n, double *below_out)
{
    try
    {
        ParameterVector params = make_params(Kp, a, s);
        {
            PiecewiseConstantRateFunction<adouble> eta(params, std::vector<double>());
            for (int i = 0; i < nt; ++i) R_out[i] = eta.R(adouble(t[i])).value();
            for (int i = 0; i < nseeds; ++i)
            {
                adouble x = eta.random_time(t1, t2, seeds[i]);
                rt_out[i] = x.value();
                rR_out[i] = eta.R(x).value();
            }
        }
        if (n >= 0 && below_out)
        {
            std::vector<double> hs{t1, t2};
            PiecewiseConstantRateFunction<adouble> eta(params, hs);
            OnePopConditionedSFS<adouble> csfs(n);
            Matrix<adouble> m = csfs.compute_below(eta).at(0);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j <= n; ++j) below_out[i * (n + 1) + j] = m(i, j).value();
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        g_err = e.what();
        return 1;
    }
}

// K coalescence times drawn the way JointCSFS draws them (jcsfs.cpp:120-127): ONE default-constructed std::mt19937
// shared by the K calls of the reference's public random_time(fac, a, b, gen); returns t and R(t).
extern "C" int ref_random_times_shared(int Kp, const double *a, const double *s, double t1, double t2, int K,
                                       double *t_out, double *R_out)
{
    try
    {
        ParameterVector params = make_params(Kp, a, s);
        const PiecewiseConstantRateFunction<adouble> eta(params, std::vector<double>());
        std::mt19937 gen;
        for (int k = 0; k < K; ++k)
        {
            const adouble t = eta.random_time(1., t1, t2, gen);
            t_out[k] = t.value();
            R_out[k] = eta.R(t).value();
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        g_err = e.what();
        return 1;
    }
}

// Q with gradients by the reference's own forward-mode AD (Eigen::AutoDiffScalar, common.h:22-25), one-population:
// pi, the transition matrix, the conditioned SFS (incorporate_theta) and the average coalescence times come from the
// reference functions on adouble inputs seeded with da [Kp x nder]; the emission vector of key k is the linear
// combination the reference's recompute_emission_probs forms (inference_manager.cpp:413-465, a translation unit that
// cannot be built here): kind[k] = 1: all ones; 2 / 3: e2 column 0 / 1 (exp(-2 alpha theta E[T]) and its complement);
// 0: sum of w * csfs[m](idx / (n+1), idx % (n+1)) over bin_idx/bin_w[bin_off[k] .. bin_off[k+1]) — the weights are
// plain doubles computed by oracle/prep_oracle.py.  Then the real HMM::Estep and HMM::Q run on those adoubles.
// Outputs: q [4], jac [4 x nder], loglik [1].
extern "C" int ref_q_jac(int M, int K, const int *keys, int L, const int *obs_in, int n,
                         int Kp, const double *a, const double *da, int nder, const double *s, const double *hs,
                         double rho, double theta, double alpha,
                         const int *kind, const int *bin_off, const int *bin_idx, const double *bin_w,
                         double *q_out, double *jac_out, double *loglik)
{
    try
    {
        const int keylen = 3;
        std::vector<adouble> av, sv;
        for (int k = 0; k < Kp; ++k)
        {
            Eigen::VectorXd d(nder);
            for (int j = 0; j < nder; ++j) d(j) = da[k * nder + j];
            av.push_back(adouble(a[k], d));
            sv.push_back(adouble(s[k]));
        }
        ParameterVector params{av, sv};
        std::vector<double> hidden_states(hs, hs + M + 1);
        PiecewiseConstantRateFunction<adouble> eta(params, hidden_states);
        Vector<adouble> pi(M);
        for (int m = 0; m < M - 1; ++m)
            pi(m) = exp(-(eta.R(hidden_states.at(m)))) - exp(-(eta.R(hidden_states.at(m + 1))));
        pi(M - 1) = exp(-(eta.R(hidden_states.at(M - 1))));
        adouble small = eta.zero() + 1e-20;
        pi = pi.unaryExpr([small](const adouble &x) { if (x < 1e-20) return small; return x; });
        pi /= pi.sum();
        Matrix<adouble> T = compute_transition(eta, rho);
        OnePopConditionedSFS<adouble> csfs(n);
        std::vector<Matrix<adouble> > sfs = incorporate_theta(csfs.compute(eta), theta);
        std::vector<adouble> avg_ct = eta.average_coal_times();
        Matrix<adouble> e2(M, 2);
        for (int m = 0; m < M; ++m)
        {
            adouble log_e2m = -2. * alpha * theta * avg_ct.at(m);
            e2(m, 0) = exp(log_e2m);
            e2(m, 1) = -expm1(log_e2m);
        }
        std::map<block_key, Vector<adouble> > emission_probs;
        for (int k = 0; k < K; ++k)
        {
            Vector<adouble> e(M);
            e.fill(eta.zero());
            if (kind[k] == 1) e.fill(eta.zero() + 1.);
            else if (kind[k] == 2) e = e2.col(0);
            else if (kind[k] == 3) e = e2.col(1);
            else
                for (int b = bin_off[k]; b < bin_off[k + 1]; ++b)
                    for (int m = 0; m < M; ++m)
                        e(m) += bin_w[b] * sfs.at(m)(bin_idx[b] / (n + 1), bin_idx[b] % (n + 1));
            emission_probs.emplace(make_key(keys + k * keylen, keylen), e);
        }
        const int ncol = 1 + keylen;
        Eigen::Map<Eigen::Matrix<int, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> > obs(
            const_cast<int *>(obs_in), L, ncol);
        spp::sparse_hash_set<std::pair<int, block_key> > targets;
        for (int i = 0; i < L; ++i)
            if (obs(i, 0) > 1) targets.insert({obs(i, 0), make_key(obs_in + i * ncol + 1, keylen)});
        TransitionBundle tb(targets, &emission_probs);
        bool sg = false;
        InferenceBundle ib{&pi, &tb, &emission_probs, &sg};
        HMM hmm(0, obs, &ib);
        tb.update(T, true);
        hmm.Estep(false);
        if (loglik) *loglik = hmm.loglik();
        Vector<adouble> qq = hmm.Q();
        for (int i = 0; i < 4; ++i)
        {
            q_out[i] = qq(i).value();
            for (int d = 0; d < nder; ++d)
                jac_out[i * nder + d] = qq(i).derivatives().size() ? qq(i).derivatives()(d) : 0.0;
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        g_err = e.what();
        return 1;
    }
}

// ---- pieces of JointCSFS<T>::pre_compute_apart (src/jcsfs.cpp:258-367) that ARE compiled here --------------------------
// jcsfs.cpp itself needs GSL and is not built; these three entry points hand the compiled building blocks it calls to
// oracle/jcsfs_apart_oracle.py, which restates only the assembly loops around them.

#include "moran_eigensystem.h"

// shiftParams / truncateParams (src/common.cpp:63-98): which = 0 shift, 1 truncate.  Returns the new piece count.
extern "C" int ref_shift_or_truncate(int which, int Kp, const double *a, const double *s, double t, int *Kout,
                                     double *a_out, double *s_out)
{
    try
    {
        ParameterVector params = make_params(Kp, a, s);
        ParameterVector r = which == 0 ? shiftParams(params, t) : truncateParams(params, t);
        *Kout = (int)r[0].size();
        for (int k = 0; k < *Kout; ++k) { a_out[k] = r[0][k].value(); s_out[k] = r[1][k].value(); }
        return 0;
    }
    catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// modified_moran_rate_matrix(N, a, na) (src/moran_eigensystem.cpp:31-52), dense row-major [(N+1) x (N+1)]
extern "C" int ref_modified_moran(int N, int a, int na, double *out)
{
    try
    {
        Eigen::SparseMatrix<mpq_class, Eigen::RowMajor> Mq = modified_moran_rate_matrix(N, a, na);
        for (int i = 0; i <= N; ++i)
            for (int j = 0; j <= N; ++j) out[i * (N + 1) + j] = 0.0;
        for (int k = 0; k < Mq.outerSize(); ++k)
            for (Eigen::SparseMatrix<mpq_class, Eigen::RowMajor>::InnerIterator it(Mq, k); it; ++it)
                out[it.row() * (N + 1) + it.col()] = it.value().get_d();
        return 0;
    }
    catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// OnePopConditionedSFS<adouble>(n).compute(eta) only (src/conditioned_sfs.cpp:86-97): raw CSFS [M x 3 x (n+1)] of the
// model (a, s) on the hidden states hs[M+1], and R(t) at nt points
extern "C" int ref_raw_csfs(int Kp, const double *a, const double *s, int M, const double *hs, int n, double *out,
                            int nt, const double *t, double *R_out)
{
    try
    {
        ParameterVector params = make_params(Kp, a, s);
        std::vector<double> hidden_states(hs, hs + M + 1);
        PiecewiseConstantRateFunction<adouble> eta(params, hidden_states);
        if (out)
        {
            OnePopConditionedSFS<adouble> csfs(n);
            std::vector<Matrix<adouble> > raw = csfs.compute(eta);
            for (int m = 0; m < M; ++m)
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j <= n; ++j) out[(m * 3 + i) * (n + 1) + j] = raw.at(m)(i, j).value();
        }
        for (int i = 0; i < nt; ++i) R_out[i] = eta.R(t[i]).value();
        return 0;
    }
    catch (const std::exception &e) { g_err = e.what(); return 1; }
}
