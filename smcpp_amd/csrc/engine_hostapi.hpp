// engine_hostapi.hpp - part of the ONE translation unit engine.hip (included there, in order; not a standalone header):
// host-only and debug exports used by the test-suite.
extern "C" {
// ---- host-only helpers exported for the CPU test-suite (no device needed) --------------------------------------

// eigensystem(EigenSolver(A)) as used by TransitionBundle::update: P_r, Pinv_r [n x n], d_r [n], scale, max|imag|
// the same through the team-parallel routine (nonsym_eig_team.hpp) with `threads` cooperating threads
int smcpp_host_eigensystem_team(int n, const double *A, int threads, double *P, double *Pinv, double *d, double *scale,
                                double *max_imag) {
    API_BEGIN
    if (n < 1 || threads < 1 || threads > 64) throw std::runtime_error("bad arguments");
    std::vector<double> a(A, A + (size_t)n * n);
    smcpp_host::EigTeam tm(threads);
    smcpp_host::EigenSystem es;
    bool ok = true;
    if (n == 1) es = smcpp_host::eigensystem(n, a);
    else {
#pragma omp parallel num_threads(threads)
        {
            if (omp_get_num_threads() != threads) {
#pragma omp single
                ok = false;
            } else {
                int gen = 0;
                smcpp_host::eigensystem_team(n, a, es, tm, omp_get_thread_num(), gen);
            }
        }
    }
    if (!ok) throw std::runtime_error("the OpenMP runtime did not provide the requested team");
    if (tm.failed.load()) throw std::runtime_error(tm.error.empty() ? "eigensolver failed" : tm.error);
    std::copy(es.P.begin(), es.P.end(), P);
    std::copy(es.Pinv.begin(), es.Pinv.end(), Pinv);
    std::copy(es.d.begin(), es.d.end(), d);
    *scale = es.scale; *max_imag = es.max_imag;
    API_END
}

int smcpp_host_eigensystem(int n, const double *A, double *P, double *Pinv, double *d, double *scale, double *max_imag) {
    API_BEGIN
    std::vector<double> a(A, A + (size_t)n * n);
    smcpp_host::EigenSystem es = smcpp_host::eigensystem(n, a);
    std::memcpy(P, es.P.data(), sizeof(double) * n * n);
    std::memcpy(Pinv, es.Pinv.data(), sizeof(double) * n * n);
    std::memcpy(d, es.d.data(), sizeof(double) * n);
    *scale = es.scale;
    *max_imag = es.max_imag;
    API_END
}

// one-population parameter preparation on the host without an engine instance
int smcpp_host_prep_onepop(int n, int n_hs, const double *hs, double polarization_error, int Kp, const double *a,
                           const double *s, double theta, double rho, double alpha, int K, const int *keys,
                           double *pi, double *T, double *E) {
    API_BEGIN
    std::vector<double> hsv(hs, hs + n_hs);
    smcpp_host::OnePopPrep prep(n, hsv, polarization_error);
    smcpp_host::ModelParams mp;
    mp.a.assign(a, a + Kp);
    mp.s.assign(s, s + Kp);
    std::vector<int> kv(keys, keys + (size_t)K * 3);
    std::vector<double> piv, Tv, Ev;
    prep.compute(mp, theta, rho, alpha, kv, K, piv, Tv, Ev);
    const int M = n_hs - 1;
    if (pi) std::memcpy(pi, piv.data(), sizeof(double) * M);
    if (T) std::memcpy(T, Tv.data(), sizeof(double) * M * M);
    if (E) std::memcpy(E, Ev.data(), sizeof(double) * (size_t)K * M);
    API_END
}


int smcpp_num_derivatives(smcpp_im *im) { return im->have_raw ? 0 : im->nder; }

// values and Jacobians of the one-population preparation: da [Kp x nder]; dpi [M x nder], dT [M*M x nder], dE [K*M x nder]
int smcpp_host_prep_onepop_jac(int n, int n_hs, const double *hs, double polarization_error, int Kp, const double *a,
                               const double *da, int nder, const double *s, double theta, double rho, double alpha,
                               int K, const int *keys, double *pi, double *T, double *E, double *dpi, double *dT,
                               double *dE) {
    API_BEGIN
    std::vector<double> hsv(hs, hs + n_hs);
    smcpp_host::OnePopPrep prep(n, hsv, polarization_error);
    smcpp_host::ModelParams mp;
    mp.a.assign(a, a + Kp);
    mp.s.assign(s, s + Kp);
    std::vector<double> dav(da, da + (size_t)Kp * nder);
    std::vector<int> kv(keys, keys + (size_t)K * 3);
    std::vector<double> piv, Tv, Ev, dpiv, dTv, dEv;
    prep.compute_with_jacobian(mp, dav, nder, theta, rho, alpha, kv, K, piv, Tv, Ev, dpiv, dTv, dEv);
    const int M = n_hs - 1;
    std::memcpy(pi, piv.data(), sizeof(double) * M);
    std::memcpy(T, Tv.data(), sizeof(double) * M * M);
    std::memcpy(E, Ev.data(), sizeof(double) * (size_t)K * M);
    std::memcpy(dpi, dpiv.data(), sizeof(double) * (size_t)M * nder);
    std::memcpy(dT, dTv.data(), sizeof(double) * (size_t)M * M * nder);
    std::memcpy(dE, dEv.data(), sizeof(double) * (size_t)K * M * nder);
    API_END
}


// PyRateFunction.R / average_coal_times (_smcpp.pyx:370-389) without an engine instance: R at nt time points and,
// if n_hs >= 2, the average coalescence time inside each of the n_hs-1 hidden-state intervals
int smcpp_host_rate_function(int Kp, const double *a, const double *s, int n_hs, const double *hs, int nt,
                             const double *t, double *R_out, double *avg_ct_out) {
    API_BEGIN
    smcpp_host::ModelParamsT<double> mp;
    mp.a.assign(a, a + Kp);
    mp.s.assign(s, s + Kp);
    std::vector<double> hsv(hs, hs + std::max(0, n_hs));
    smcpp_host::RateFunctionT<double> eta(mp, hsv);
    for (int i = 0; i < nt; ++i) R_out[i] = eta.R(t[i]);
    if (n_hs >= 2 && avg_ct_out) {
        const std::vector<double> v = eta.average_coal_times();
        std::memcpy(avg_ct_out, v.data(), sizeof(double) * v.size());
    }
    API_END
}

// seeds a dual model from (a, da); nder == 0 leaves the derivative parts empty
static smcpp_host::ModelParamsT<smcpp_host::dual> dual_model(int Kp, const double *a, const double *da, int nder,
                                                              const double *s) {
    smcpp_host::ModelParamsT<smcpp_host::dual> mp;
    mp.s.assign(s, s + Kp);
    mp.a.resize(Kp);
    for (int k = 0; k < Kp; ++k) {
        mp.a[k] = smcpp_host::dual(a[k]);
        for (int d = 0; d < nder; ++d) mp.a[k].d[d] = da[(size_t)k * nder + d];
    }
    return mp;
}

int smcpp_host_rate_function_jac(int Kp, const double *a, const double *da, int nder, const double *s, int n_hs,
                                 const double *hs, int nt, const double *t, double *R_out, double *dR_out,
                                 double *avg_ct_out, double *davg_ct_out) {
    API_BEGIN
    if (nder < 0 || nder > smcpp_host::MAXD) throw std::runtime_error("too many derivatives");
    smcpp_host::DualScope sc(nder);
    std::vector<double> hsv(hs, hs + std::max(0, n_hs));
    smcpp_host::RateFunctionT<smcpp_host::dual> eta(dual_model(Kp, a, da, nder, s), hsv);
    for (int i = 0; i < nt; ++i) {
        const smcpp_host::dual r = eta.R(t[i]);
        R_out[i] = r.v;
        for (int d = 0; d < nder; ++d) dR_out[(size_t)i * nder + d] = r.d[d];
    }
    if (n_hs >= 2 && avg_ct_out) {
        const std::vector<smcpp_host::dual> v = eta.average_coal_times();
        for (size_t i = 0; i < v.size(); ++i) {
            avg_ct_out[i] = v[i].v;
            if (davg_ct_out) for (int d = 0; d < nder; ++d) davg_ct_out[i * nder + d] = v[i].d[d];
        }
    }
    API_END
}

// Test hook: the one-population cold preparation with the conditioned SFS / emission table evaluated by the device kernels of
// prep_dev.hpp (mode 0) or by the same phases run serially on the host (mode 1: CPU tests); pi and the transition matrix come
// from the host routines either way.  Outputs as smcpp_host_prep_onepop_jac, plus the conditioned SFS after incorporate_theta
// sfs [M x 3 (n+1)] and its Jacobian (both may be NULL).
int smcpp_dev_prep_onepop(int n, int n_hs, const double *hs, double polarization_error, int Kp, const double *a,
                          const double *da, int nder, const double *s, double theta, double rho, double alpha, int K,
                          const int *keys, int mode, double *pi, double *T, double *E, double *dpi, double *dT, double *dE,
                          double *sfs, double *dsfs) {
    API_BEGIN
    if (!DevPrep::supported(n)) throw std::runtime_error("device preparation does not support this sample size");
    const std::vector<double> hsv(hs, hs + n_hs);
    const int M = n_hs - 1;
    smcpp_host::OnePopPrep hp(n, hsv, polarization_error);
    DevPrep dp;
    dp.emulate = mode != 0;
    if (!dp.emulate) { int dev = 0; HIPCHK(hipGetDevice(&dev)); }
    dp.set_static(hp.tables());
    const std::vector<int> kv(keys, keys + (size_t)3 * K);
    dp.set_keys(hp, kv, K, {}, {}, {}, K, M, (M + 15) / 16 * 16, 0);
    std::vector<double> Ev, dEv, sf, dsf;
    if (da && nder > 0) {
        smcpp_host::DualScope sc(nder);
        const auto p = dual_model(Kp, a, da, nder, s);
        smcpp_host::RateFunctionT<smcpp_host::dual> eta(p, hsv);
        std::vector<smcpp_host::dual> pd;
        smcpp_host::initial_distribution(eta, pd);
        const std::vector<smcpp_host::dual> act = eta.average_coal_times();
        dp.run(eta, act, theta, alpha, nder, nullptr);
        for (int i = 0; i < M; ++i) { pi[i] = pd[i].v; for (int d = 0; d < nder; ++d) dpi[(size_t)i * nder + d] = pd[i].d[d]; }
        // the transition matrix as the engine forms it: values from the double routine, derivative planes of the O(M) generators by
        // the chain rule (transition_generators_jac), expanded to dT
        std::vector<double> Tv, dTv;
        smcpp_host::TransitionGenJac tj;
        if (!host_transition_with_planes(eta, act, rho, nder, Tv, tj))
            split_duals(smcpp_host::compute_transition<smcpp_host::dual>(eta, rho), nder, Tv, dTv);
        else smcpp_host::transition_expand_jac(tj, dTv);
        std::memcpy(T, Tv.data(), sizeof(double) * Tv.size());
        std::memcpy(dT, dTv.data(), sizeof(double) * dTv.size());
    } else {
        nder = 0;
        smcpp_host::ModelParamsT<double> p;
        p.a.assign(a, a + Kp); p.s.assign(s, s + Kp);
        smcpp_host::RateFunctionT<double> eta(p, hsv);
        std::vector<double> pv;
        smcpp_host::initial_distribution(eta, pv);
        dp.run(eta, eta.average_coal_times(), theta, alpha, 0, nullptr);
        const std::vector<double> Tv = smcpp_host::compute_transition<double>(eta, rho);
        std::memcpy(pi, pv.data(), sizeof(double) * M);
        std::memcpy(T, Tv.data(), sizeof(double) * (size_t)M * M);
    }
    if (!dp.emulate) HIPCHK(hipDeviceSynchronize());
    dp.fetch(Ev, dEv, sf, dsf);
    dp.check_flags();
    std::memcpy(E, Ev.data(), sizeof(double) * Ev.size());
    if (nder && dE) std::memcpy(dE, dEv.data(), sizeof(double) * dEv.size());
    if (sfs) std::memcpy(sfs, sf.data(), sizeof(double) * sf.size());
    if (nder && dsfs) std::memcpy(dsfs, dsf.data(), sizeof(double) * dsf.size());
    API_END
}

// Test hook: Q's four terms and their gradient [4 x nder] for given summed statistics g0 [M], xi [M x M], gs [K x M], evaluated
// by the phases of the device kernel k_q_reduce run serially on the host (prep_dev.hpp: emulate_q) from the emulated device
// preparation and the generator planes of the transition matrix - the data path smcpp_q takes on the GPU.
int smcpp_dev_q_emulate(int n, int n_hs, const double *hs, double polarization_error, int Kp, const double *a, const double *da,
                        int nder, const double *s, double theta, double rho, double alpha, int K, const int *keys,
                        const double *g0, const double *xi, const double *gs, double *val, double *jac) {
    API_BEGIN
    if (!DevPrep::supported(n)) throw std::runtime_error("device preparation does not support this sample size");
    const std::vector<double> hsv(hs, hs + n_hs);
    const int M = n_hs - 1;
    smcpp_host::OnePopPrep hp(n, hsv, polarization_error);
    DevPrep dp;
    dp.emulate = true;
    dp.set_static(hp.tables());
    const std::vector<int> kv(keys, keys + (size_t)3 * K);
    dp.set_keys(hp, kv, K, {}, {}, {}, K, M, (M + 15) / 16 * 16, 0);
    smcpp_host::DualScope sc(nder);
    const auto p = dual_model(Kp, a, da, nder, s);
    smcpp_host::RateFunctionT<smcpp_host::dual> eta(p, hsv);
    std::vector<smcpp_host::dual> pd;
    smcpp_host::initial_distribution(eta, pd);
    const std::vector<smcpp_host::dual> act = eta.average_coal_times();
    dp.run(eta, act, theta, alpha, nder, nullptr);
    dp.check_flags();
    std::vector<double> Tv;
    smcpp_host::TransitionGenJac tj;
    if (!host_transition_with_planes(eta, act, rho, nder, Tv, tj)) throw std::runtime_error("transition generators need the pairwise fallback");
    std::vector<double> blk((size_t)4 * M * (1 + nder), 0.0), out((size_t)4 * (1 + nder), 0.0);
    for (int i = 0; i < M; ++i) { blk[i] = pd[i].v; blk[M + i] = i < M - 1 ? tj.ed[i] : 0.0; blk[2 * M + i] = tj.pf[i]; blk[3 * M + i] = tj.W[i]; }
    double *pl = blk.data() + (size_t)4 * M;
    const size_t ps = (size_t)nder * M;
    for (int d = 0; d < nder; ++d)
        for (int i = 0; i < M; ++i) {
            pl[(size_t)d * M + i] = pd[i].d[d];
            pl[ps + (size_t)d * M + i] = i < M - 1 ? tj.ded[(size_t)i * nder + d] : 0.0;
            pl[2 * ps + (size_t)d * M + i] = tj.dpf[(size_t)i * nder + d];
            pl[3 * ps + (size_t)d * M + i] = tj.dW[(size_t)i * nder + d];
        }
    std::vector<int> knb(K);
    for (int k = 0; k < K; ++k) knb[k] = keys[3 * k + 2] > 0;
    smcpp_dev::QArgs q;
    q.M = M; q.Kq = K; q.nder = nder;
    q.g0 = g0; q.xi = xi; q.gs = gs; q.key_nb = knb.data();
    q.pi_v = blk.data(); q.ed_v = blk.data() + M; q.pf_v = blk.data() + 2 * M; q.W_v = blk.data() + 3 * M;
    q.pi_d = pl; q.ed_d = pl + ps; q.pf_d = pl + 2 * ps; q.W_d = pl + 3 * ps;
    q.mix_p2 = 1e-5 / (double)(M + 1);
    q.E_v = dp.e_Eg_v.data(); q.E_d = dp.e_Eg_d.data();
    q.out = out.data();
    smcpp_dev::emulate_q(q);
    for (int t = 0; t < 4; ++t) { val[t] = out[t]; for (int d = 0; d < nder; ++d) jac[(size_t)t * nder + d] = out[(size_t)4 * (1 + d) + t]; }
    API_END
}

int smcpp_host_random_coal_times(int Kp, const double *a, const double *s, double t1, double t2, int K,
                                 const unsigned long long *seeds, double *t_out, double *R_out) {
    API_BEGIN
    smcpp_host::ModelParamsT<double> mp;
    mp.a.assign(a, a + Kp);
    mp.s.assign(s, s + Kp);
    smcpp_host::RateFunctionT<double> eta(mp, std::vector<double>());
    for (int i = 0; i < K; ++i) {
        t_out[i] = eta.random_time(t1, t2, seeds[i]);
        R_out[i] = eta.R(t_out[i]);
    }
    API_END
}

int smcpp_host_raw_sfs(int n, int Kp, const double *a, const double *da, int nder, const double *s, double t1,
                       double t2, int below_only, double *sfs, double *dsfs) {
    API_BEGIN
    if (n < 0) throw std::runtime_error("n must be >= 0");
    if (nder < 0 || nder > smcpp_host::MAXD) throw std::runtime_error("too many derivatives");
    smcpp_host::DualScope sc(nder);
    const std::vector<double> hsv{t1, t2};
    smcpp_host::RateFunctionT<smcpp_host::dual> eta(dual_model(Kp, a, da, nder, s), hsv);
    const auto tb = smcpp_host::csfs_tables(n);
    const auto v = smcpp_host::conditioned_sfs<smcpp_host::dual>(eta, *tb, below_only != 0);
    for (size_t i = 0; i < v[0].size(); ++i) {
        sfs[i] = v[0][i].v;
        if (dsfs) for (int d = 0; d < nder; ++d) dsfs[i * nder + d] = v[0][i].d[d];
    }
    API_END
}

int smcpp_host_joint_csfs(int n1, int n2, int a1, int a2, int n_hs, const double *hs, int K1, const double *pa1,
                          const double *ps1, const double *da1, int K2, const double *pa2, const double *ps2,
                          const double *da2, int nder, double split, int Kmc, double *out, double *dout) {
    API_BEGIN
    if (nder < 0 || nder > smcpp_host::MAXD) throw std::runtime_error("too many derivative directions (max 64)");
    std::vector<double> hsv(hs, hs + n_hs);
    smcpp_host::ModelParams m1, m2;
    m1.a.assign(pa1, pa1 + K1); m1.s.assign(ps1, ps1 + K1);
    m2.a.assign(pa2, pa2 + K2); m2.s.assign(ps2, ps2 + K2);
    if (nder == 0) {
        smcpp_host::ModelParamsT<double> p1, p2;
        p1.a = m1.a; p1.s = m1.s; p2.a = m2.a; p2.s = m2.s;
        smcpp_host::JointCsfsT<double> j(n1, n2, a1, a2, hsv, Kmc);
        const auto J = j.compute(p1, p2, split);
        size_t o = 0;
        for (const auto &m : J) { std::memcpy(out + o, m.data(), sizeof(double) * m.size()); o += m.size(); }
    } else {
        smcpp_host::DualScope sc(nder);
        std::vector<double> d1, d2;
        if (da1) d1.assign(da1, da1 + (size_t)K1 * nder);
        if (da2) d2.assign(da2, da2 + (size_t)K2 * nder);
        smcpp_host::JointCsfsT<smcpp_host::dual> j(n1, n2, a1, a2, hsv, Kmc);
        const auto J = j.compute(make_dual_model(m1, d1, nder), make_dual_model(m2, d2, nder), split);
        size_t o = 0;
        for (const auto &m : J)
            for (const auto &x : m) {
                out[o] = x.v;
                if (dout) for (int d = 0; d < nder; ++d) dout[o * nder + d] = x.d[d];
                ++o;
            }
    }
    API_END
}

int smcpp_host_prep_twopop(int n1, int n2, int a1, int a2, int n_hs, const double *hs, double polarization_error,
                           int Kd, const double *ad, const double *sd, int K1, const double *pa1, const double *ps1,
                           int K2, const double *pa2, const double *ps2, double split, double theta, double rho,
                           double alpha, int K, const int *keys, double *pi, double *T, double *E) {
    API_BEGIN
    std::vector<double> hsv(hs, hs + n_hs);
    smcpp_host::TwoPopPrep prep(n1, n2, a1, a2, hsv, polarization_error);
    smcpp_host::ModelParamsT<double> d, p1, p2;
    d.a.assign(ad, ad + Kd); d.s.assign(sd, sd + Kd);
    p1.a.assign(pa1, pa1 + K1); p1.s.assign(ps1, ps1 + K1);
    p2.a.assign(pa2, pa2 + K2); p2.s.assign(ps2, ps2 + K2);
    std::vector<int> kv(keys, keys + (size_t)K * 6);
    std::vector<double> piv, Tv, Ev;
    prep.compute_t<double>(d, p1, p2, split, theta, rho, alpha, kv, K, piv, Tv, Ev);
    const int M = n_hs - 1;
    std::memcpy(pi, piv.data(), sizeof(double) * M);
    std::memcpy(T, Tv.data(), sizeof(double) * M * M);
    std::memcpy(E, Ev.data(), sizeof(double) * (size_t)K * M);
    API_END
}

}  // extern "C"
