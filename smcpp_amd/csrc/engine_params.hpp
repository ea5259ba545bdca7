// engine_params.hpp - part of the ONE translation unit engine.hip (included there, in order; not a standalone header):
// parameters: cold preparation (host and device routes), device Q / gradient, uploads.
// ---------------------------------------------------------------------------------------------------------------
// parameters
// ---------------------------------------------------------------------------------------------------------------
static smcpp_host::ModelParamsT<smcpp_host::dual> make_dual_model(const smcpp_host::ModelParams &mp,
                                                                   const std::vector<double> &da, int nder) {
    smcpp_host::ModelParamsT<smcpp_host::dual> r;
    r.s = mp.s;
    r.a.resize(mp.a.size());
    for (size_t k = 0; k < mp.a.size(); ++k) {
        r.a[k] = smcpp_host::dual(mp.a[k]);
        if (!da.empty()) for (int d = 0; d < nder; ++d) r.a[k].d[d] = da[k * nder + d];
    }
    return r;
}

static void split_duals(const std::vector<smcpp_host::dual> &x, int nder, std::vector<double> &v, std::vector<double> &j) {
    v.resize(x.size());
    j.resize(x.size() * (size_t)nder);
    for (size_t i = 0; i < x.size(); ++i) {
        v[i] = x[i].v;
        for (int d = 0; d < nder; ++d) j[i * nder + d] = x[i].d[d];
    }
}

void smcpp_im::prepare_params() {
    // do_dirty_work (inference_manager.cpp:213-229) for the model-parameter path; the raw path already has pi/T/E.
    if (have_raw || params_fresh) return;
    if (!have_model) throw std::runtime_error("no model parameters: call set_params or set_raw before E_step");
    if (std::isnan(theta) || std::isnan(rho)) throw std::runtime_error("theta / rho / alpha must be set");
    if (npop == 2) {
        // TwoPopInferenceManager::setParams (inference_manager.cpp:542-550): pi / T from the distinguished model,
        // emissions from the joint CSFS of (population 1, population 2, split)
        if (model_p1.a.empty() || model_p2.a.empty())
            throw std::runtime_error("two-population manager: call set_params_twopop (or set_raw) before E_step");
        if (!twopop_prep) {
            twopop_prep.reset(new smcpp_host::TwoPopPrep(n[0], n[1], na[0], na[1], hs, polarization_error));
            // The two-population preparation is the one host phase that still runs on a team of threads, in two parallel regions per
            // eval; with libomp's workers asleep in between (block time 0, above) each region pays their wake-up - more than its
            // work.  Spinning that spans the GPU phase of an eval: config C4 561 -> 676 evals/s with one millisecond (15 threads; measured
            // profiles/r05_*).  One millisecond is bistable, though - an eval of C4 takes 1.2 - 1.3 ms, and once one eval is late the workers
            // are asleep at every following region (runs at 680 - 720 evals/s next to runs at 800 - 830): TWO milliseconds (6 runs of 6
            // at 800 - 825).  SMCPP_OMP_BLOCKTIME overrides.
            if (kmp_set_blocktime && !opt().has(smcpp_opt::O_OMP_BLOCKTIME)) kmp_set_blocktime(2);
        }
        smcpp_host::TwoPopPrep &prep = *twopop_prep;
        {
            // the batched conditioned SFS on the device (values; SMCPP_PREP=host / smcpp_set_prep_mode(1): everything on the host)
            const bool host_only2 = opt().is(smcpp_opt::O_PREP, "host");
            if (!twopop_dev) { twopop_dev.reset(new TwoPopDevCsfs()); twopop_dev->device = device; twopop_dev->stream = stream; }
            prep.batch_dev = (host_only2 || force_host_prep || nder > 0) ? nullptr : twopop_dev.get();
        }
        E_on_dev = false;
        tgen_valid = false; dT_valid = true; T_lazy = false;
        // with a global key dictionary (multi-GPU) the table is prepared for EVERY global key - Q on the all-reduced statistics
        // also covers keys only other ranks' contigs hold; the local table is the sub-list of this rank's keys
        const std::vector<int> &pk2 = have_global ? gkeys : keys;
        const int K2 = (int)(pk2.size() / keylen);
        std::vector<double> Ep, dEp;
        if (nder > 0) {
            smcpp_host::DualScope sc(nder);
            std::vector<smcpp_host::dual> pd, Td, Ed, emd;
            prep.compute_t<smcpp_host::dual>(make_dual_model(model, model_da, nder), make_dual_model(model_p1, model_da1, nder),
                                             make_dual_model(model_p2, model_da2, nder), split, theta, rho, alpha, pk2, K2,
                                             pd, Td, Ed, &emd);
            split_duals(pd, nder, pi, dpi); split_duals(Td, nder, T, dT); split_duals(Ed, nder, Ep, dEp);
            split_duals(emd, nder, emission, demission);
        } else {
            smcpp_host::ModelParamsT<double> d, p1, p2;
            d.a = model.a; d.s = model.s; p1.a = model_p1.a; p1.s = model_p1.s; p2.a = model_p2.a; p2.s = model_p2.s;
            prep.compute_t<double>(d, p1, p2, split, theta, rho, alpha, pk2, K2, pi, T, Ep, &emission);
            demission.clear();
        }
        if (!have_global) { E.swap(Ep); dE.swap(dEp); Eg.clear(); dEg.clear(); }
        else {
            E.assign((size_t)K * M, 0.0);
            dE.assign(nder > 0 ? (size_t)K * M * nder : 0, 0.0);
            for (int k = 0; k < K; ++k) {
                const int kg = local_to_global[k];
                std::memcpy(&E[(size_t)k * M], &Ep[(size_t)kg * M], sizeof(double) * M);
                if (nder > 0) std::memcpy(&dE[(size_t)k * M * nder], &dEp[(size_t)kg * M * nder], sizeof(double) * M * nder);
            }
            Eg.swap(Ep); dEg.swap(dEp);
        }
        params_fresh = true;
        return;
    }
    // (kept across E-steps: it caches the keys' marginalisation bins; rebuilt when the hidden states change)
    if (!prep1 || prep1_hs != hs) {
        prep1.reset(new smcpp_host::OnePopPrep(n[0], hs, polarization_error)); prep1_hs = hs;
        if (dprep) dprep->keys_ready = false;
    }
    smcpp_host::OnePopPrep &prep = *prep1;
    {
        // conditioned SFS + emission table on the device (SMCPP_PREP=host: the host routines, as in rounds 1-3)
        const bool host_only = opt().is(smcpp_opt::O_PREP, "host");
        if (!host_only && !force_host_prep && DevPrep::supported(n[0], (int)model.a.size() + (int)hs.size()) && !smcpp_host::csfs_direct_flag()) { dev_prepare(); return; }
    }
    E_on_dev = false;
    tgen_valid = false; dT_valid = true; T_lazy = false;
    // with a global key dictionary (multi-GPU) the emission table is prepared for every global key; the local table
    // is the sub-list of the keys this rank's contigs hold
    const std::vector<int> &pk = have_global ? gkeys : keys;
    const int Kp_ = (int)(pk.size() / keylen);
    std::vector<double> Ep, dEp;
    if (nder > 0) prep.compute_with_jacobian(model, model_da, nder, theta, rho, alpha, pk, Kp_, pi, T, Ep, dpi, dT, dEp,
                                             &emission, &demission);
    else { prep.compute(model, theta, rho, alpha, pk, Kp_, pi, T, Ep, &emission); demission.clear(); }
    if (!have_global) { E.swap(Ep); dE.swap(dEp); }
    else {
        E.assign((size_t)K * M, 0.0);
        dE.assign(nder > 0 ? (size_t)K * M * nder : 0, 0.0);
        for (int k = 0; k < K; ++k) {
            const int kg = local_to_global[k];
            std::memcpy(&E[(size_t)k * M], &Ep[(size_t)kg * M], sizeof(double) * M);
            if (nder > 0) std::memcpy(&dE[(size_t)k * M * nder], &dEp[(size_t)kg * M * nder], sizeof(double) * M * nder);
        }
        Eg.swap(Ep); dEg.swap(dEp);
    }
    params_fresh = true;
}

// Transition matrix of a model with derivative seeds: values by the double routines on the VALUES of the dual rate function,
// derivative planes of the O(M) generators by the chain rule over plain arrays (prep.hpp: transition_generators_jac).  Returns
// false when a row needs the pairwise fallback (the caller then takes the generic duals through the whole matrix).
static bool host_transition_with_planes(const smcpp_host::RateFunctionT<smcpp_host::dual> &eta, const std::vector<smcpp_host::dual> &act,
                                        double rho, int nder, std::vector<double> &T, smcpp_host::TransitionGenJac &tj) {
    smcpp_host::RateFunctionT<double> ev;
    ev.hidden_states = eta.hidden_states; ev.ts = eta.ts; ev.hs_indices = eta.hs_indices; ev.K = eta.K;
    ev.ada.resize(eta.ada.size()); ev.Rrng.resize(eta.Rrng.size());
    const int K = eta.K, M = (int)eta.hidden_states.size() - 1;
    std::vector<double> dada((size_t)K * nder), avg(M), davg((size_t)M * nder);
    for (int k = 0; k < K; ++k) { ev.ada[k] = eta.ada[k].v; for (int d = 0; d < nder; ++d) dada[(size_t)k * nder + d] = eta.ada[k].d[d]; }
    for (size_t k = 0; k < eta.Rrng.size(); ++k) ev.Rrng[k] = eta.Rrng[k].v;
    for (int m = 0; m < M; ++m) { avg[m] = act[m].v; for (int d = 0; d < nder; ++d) davg[(size_t)m * nder + d] = act[m].d[d]; }
    smcpp_host::TransitionGenerators<double> g;
    tj = smcpp_host::transition_generators_jac(ev, rho, avg, dada.data(), davg.data(), nder, &g);
    if (!tj.ok) return false;
    T = smcpp_host::transition_expand<double>(g);
    return true;
}

// One-population do_dirty_work with the O(states x n^2 x directions) part on the device: the host builds the rate function
// (O(pieces)), pi, the average coalescence times and - while the kernels already run - the transition matrix.
void smcpp_im::dev_prepare() {
    HIPCHK(hipSetDevice(device));
    if (!dprep) { dprep.reset(new DevPrep()); dprep->set_static(prep1->tables()); }
    const std::vector<int> &pk = have_global ? gkeys : keys;
    const int Kp_ = (int)(pk.size() / keylen);
    if (!dprep->keys_ready) {
        // per prepared key: its row of the statistics' table, its slot of the scan chains' table, the longest span the scan
        // chains expand position by position (ss_extract_generators' underflow bound, checked by the kernel)
        std::vector<int> ms_local(K, 1), local(Kp_, -1), slot(Kp_, -1), maxspan(Kp_, 1);
        for (const Group &gr : groups)
            if (!(ss_hybrid && gr.span > ss_hyb_th)) ms_local[gr.kid] = std::max(ms_local[gr.kid], gr.span);
        for (int k = 0; k < K; ++k) {
            const int kg = have_global ? local_to_global[k] : k;
            local[kg] = k;
            slot[kg] = (ss_static && (int)ss_slot_of_key.size() == K) ? ss_slot_of_key[k] : k;
            maxspan[kg] = ms_local[k];
        }
        dprep->set_keys(*prep1, pk, Kp_, local, slot, maxspan, K, M, Mp, ss_static ? 64 * NPL : 0);
    }
    T_lazy = false;
    if (nder > 0) {
        HostTrace tr;
        smcpp_host::DualScope sc(nder);
        const smcpp_host::RateFunctionT<smcpp_host::dual> eta(make_dual_model(model, model_da, nder), hs);
        tr.mark("prep(d): rate function");
        const std::vector<smcpp_host::dual> act = eta.average_coal_times();
        tr.mark("prep(d): average coal times");
        dprep->run(eta, act, theta, alpha, nder, stream);
        tr.mark("prep(d): pack + 2 launches");
        std::vector<smcpp_host::dual> pd;
        smcpp_host::initial_distribution(eta, pd);
        split_duals(pd, nder, pi, dpi);
        tr.mark("prep(d): pi");
        // transition matrix: values + the derivative planes of its O(M) generators; the M x M x nder Jacobian is expanded
        // only when its getter asks (ensure_dT), Q's gradient reads the planes on the device
        tgen_valid = host_transition_with_planes(eta, act, rho, nder, T, tgen);
        tr.mark("prep(d): T + generator planes");
        dT.clear();
        dT_valid = false;
        if (!tgen_valid) { split_duals(smcpp_host::compute_transition<smcpp_host::dual>(eta, rho), nder, T, dT); dT_valid = true; }
    } else {
        smcpp_host::ModelParamsT<double> p;
        p.a = model.a; p.s = model.s;
        HostTrace tr;
        const smcpp_host::RateFunctionT<double> eta(p, hs);
        tr.mark("prep: rate function");
        const std::vector<double> act = eta.average_coal_times();
        tr.mark("prep: average coal times");
        dprep->run(eta, act, theta, alpha, 0, stream);
        tr.mark("prep: pack + 2 launches");
        smcpp_host::initial_distribution(eta, pi);
        smcpp_host::TransitionGenerators<double> g;
        tgen = smcpp_host::transition_generators_jac(eta, rho, act, nullptr, nullptr, 0, &g);
        tr.mark("prep: pi + T generators");
        tgen_valid = tgen.ok;
        // (the expansion waits until somebody reads the matrix - ensure_T; without usable generators, or with SMCPP_T_LAZY=0: here)
        tgen_g = std::move(g);
        T_lazy = tgen_valid && !opt().off(smcpp_opt::O_T_LAZY);
        if (!T_lazy) T = smcpp_host::transition_expand<double>(tgen_g);
        tr.mark("prep: T expand");
        dpi.clear(); dT.clear();
        dT_valid = true;
    }
    E_on_dev = true;
    Eg.clear(); dEg.clear();
    params_fresh = true;
}

// The emission table (and its Jacobian, and InferenceManager::emission) of a device preparation, to the host vectors
void smcpp_im::sync_host_E() {
    if (!E_on_dev) return;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamSynchronize(stream));
    std::vector<double> Ep, dEp;
    dprep->fetch(Ep, dEp, emission, demission);
    dprep->check_flags();
    if (!have_global) { E.swap(Ep); dE.swap(dEp); }
    else {
        E.assign((size_t)K * M, 0.0);
        dE.assign(nder > 0 ? (size_t)K * M * nder : 0, 0.0);
        for (int k = 0; k < K; ++k) {
            const int kg = local_to_global[k];
            std::memcpy(&E[(size_t)k * M], &Ep[(size_t)kg * M], sizeof(double) * M);
            if (nder > 0) std::memcpy(&dE[(size_t)k * M * nder], &dEp[(size_t)kg * M * nder], sizeof(double) * M * nder);
        }
        Eg.swap(Ep); dEg.swap(dEp);
    }
    E_on_dev = false;
}

void smcpp_im::ensure_dT() {
    ensure_T();
    if (dT_valid) return;
    smcpp_host::transition_expand_jac(tgen, dT);
    dT_valid = true;
}

// HMM::Q (src/hmm.cpp:155-193) summed over contigs (inference_manager.cpp:116-126) and its forward-mode gradient, evaluated on
// the device from the statistics that already live there, the device-prepared emission table (+ planes) and the generators of
// the transition matrix.  Returns false when this call has to take the host route (no device preparation, a local / global
// key-list mismatch, the pairwise fallback of the transition matrix).
bool smcpp_im::q_device(double val[4], double *jac) {
    const bool off = opt().is(smcpp_opt::O_Q, "host");
    if (off || !E_on_dev || !tgen_valid || have_raw || (have_global && !have_reduced)) return false;
    if (have_reduced && (int)g_stats.size() != 1 + M + M * M + dprep->Kk * M) return false;
    HIPCHK(hipSetDevice(device));
    if (!qdev) qdev.reset(new QDev());
    QDev &q = *qdev;
    const int Kq = dprep->Kk, nd = nder;
    const size_t nstat = (size_t)M + (size_t)M * M + (size_t)Kq * M;
    if (!q.stats_ready || q.Kq != Kq) {
        q.d_stats.alloc(nstat);
        std::vector<int> knb(Kq);
        const std::vector<int> &pk = have_global ? gkeys : keys;
        for (int k = 0; k < Kq; ++k) { int nb = 0; for (int p = 0; p < npop; ++p) nb += pk[(size_t)k * keylen + 3 * p + 2]; knb[k] = nb > 0; }
        q.d_keynb.alloc(Kq);
        HIPCHK(hipMemcpyAsync(q.d_keynb.p, knb.data(), sizeof(int) * Kq, hipMemcpyHostToDevice, stream));
        if (have_reduced) HIPCHK(hipMemcpyAsync(q.d_stats.p, g_stats.data() + 1, sizeof(double) * nstat, hipMemcpyHostToDevice, stream));
        else if (!estep_done) {
            fetch_stats();                                   // the statistics of a freshly constructed HMM (host)
            std::vector<double> st(nstat, 0.0);
            for (int c = 0; c < n_contigs; ++c) {
                for (int i = 0; i < M; ++i) st[i] += h_gamma0[(size_t)c * M + i];
                for (size_t e = 0; e < (size_t)M * M; ++e) st[M + e] += h_xisum[(size_t)c * M * M + e];
                for (size_t e = 0; e < (size_t)K * M; ++e) st[M + (size_t)M * M + e] += h_gsum[(size_t)c * K * M + e];
            }
            HIPCHK(hipMemcpyAsync(q.d_stats.p, st.data(), sizeof(double) * nstat, hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));            // (st is pageable and local)
        } else
            hipLaunchKernelGGL(smcpp_dev::k_q_stats, dim3(ceil_div((long long)nstat, 256)), dim3(256), 0, stream, n_contigs, M, Mp, K,
                               (const double *)d_gamma0.p, (const double *)d_xisum.p, (const double *)d_gsum.p, q.d_stats.p);
        HIPCHK(hipStreamSynchronize(stream));                // (knb is local)
        q.stats_ready = true;
        q.Kq = Kq;
    }
    // ---- per call: pi and the generators with their planes, one pinned block: values [4][M], planes [4][nder][M] ----
    const size_t ndbl = (size_t)4 * M * (1 + nd);
    q.stage.reset(ndbl * sizeof(double) + 256);
    if (ndbl * sizeof(double) > q.in_cap) {
        if (q.d_in) (void)hipFree(q.d_in);
        q.in_cap = ndbl * sizeof(double) * 2;
        HIPCHK(hipMalloc((void **)&q.d_in, q.in_cap));
        smcpp_opt::poison(q.d_in, q.in_cap, __LINE__, __FILE__);
    }
    double *hb = reinterpret_cast<double *>(q.stage.base);
    for (int i = 0; i < M; ++i) {
        hb[i] = pi[i]; hb[M + i] = i < M - 1 ? tgen.ed[i] : 0.0; hb[2 * M + i] = tgen.pf[i]; hb[3 * M + i] = tgen.W[i];
    }
    double *pl = hb + (size_t)4 * M;
    const size_t ps = (size_t)nd * M;                    // one array's planes
    for (int d = 0; d < nd; ++d)
        for (int i = 0; i < M; ++i) {
            pl[(size_t)d * M + i] = dpi[(size_t)i * nd + d];
            pl[ps + (size_t)d * M + i] = i < M - 1 ? tgen.ded[(size_t)i * nd + d] : 0.0;
            pl[2 * ps + (size_t)d * M + i] = tgen.dpf[(size_t)i * nd + d];
            pl[3 * ps + (size_t)d * M + i] = tgen.dW[(size_t)i * nd + d];
        }
    HostTrace trq;
    HIPCHK(hipMemcpyAsync(q.d_in, hb, ndbl * sizeof(double), hipMemcpyHostToDevice, stream));
    const int nslice = 4;
    const size_t nout = (size_t)4 * (1 + nd) * nslice;
    q.d_out.alloc(nout);
    if (nout > q.h_out_cap) {
        if (q.h_out) (void)hipHostFree(q.h_out);
        q.h_out_cap = nout * 2;
        HIPCHK(hipHostMalloc((void **)&q.h_out, q.h_out_cap * sizeof(double), hipHostMallocDefault));
    }
    const double *bd = reinterpret_cast<const double *>(q.d_in);
    const double *bp = bd + (size_t)4 * M;
    smcpp_dev::QArgs a;
    a.M = M; a.Kq = Kq; a.nder = nd;
    a.g0 = q.d_stats.p; a.xi = q.d_stats.p + M; a.gs = q.d_stats.p + M + (size_t)M * M;
    a.key_nb = q.d_keynb.p;
    a.pi_v = bd; a.ed_v = bd + M; a.pf_v = bd + 2 * M; a.W_v = bd + 3 * M;
    a.pi_d = bp; a.ed_d = bp + ps; a.pf_d = bp + 2 * ps; a.W_d = bp + 3 * ps;
    a.mix_p2 = 1e-5 / (double)(M + 1);
    a.E_v = dprep->d_Eg_v.p; a.E_d = dprep->d_Eg_d.p;
    a.out = q.d_out.p;
    a.nslice = nslice;
    const int nt = 1024;
    const size_t lds = (size_t)(8 * M + 4 * (nt / 64) * 2) * sizeof(double);
    hipLaunchKernelGGL(smcpp_dev::k_q_reduce, dim3(1 + nd, nslice), dim3(nt), lds, stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(q.h_out, q.d_out.p, nout * sizeof(double), hipMemcpyDeviceToHost, stream));
    trq.mark("q: enqueue");
    HIPCHK(hipStreamSynchronize(stream));
    trq.mark("q: wait (prep kernels + q + copies)");
    dprep->check_flags();
    auto slices = [&](int b, int t) { double r = 0.0; for (int sl = 0; sl < nslice; ++sl) r += q.h_out[((size_t)b * nslice + sl) * 4 + t]; return r; };
    for (int t = 0; t < 4; ++t) val[t] = slices(0, t);
    if (jac) for (int t = 0; t < 4; ++t) for (int d = 0; d < nd; ++d) jac[(size_t)t * nd + d] = slices(1 + d, t);
    return true;
}

// Emission vectors of the global keys for the reduced Q when the parameters did not come from prepare_params
void smcpp_im::global_emissions() {
    sync_host_E();
    const int Kg = (int)(gkeys.size() / keylen);
    if (!have_raw && (int)Eg.size() == Kg * M) return;        // prepare_params filled them
    Eg.assign((size_t)Kg * M, NAN);
    dEg.clear();
    std::map<std::vector<int>, int> gm;
    for (int k = 0; k < Kg; ++k) gm[std::vector<int>(gkeys.begin() + (size_t)k * keylen, gkeys.begin() + (size_t)(k + 1) * keylen)] = k;
    if (have_raw) {
        const int Kr = (int)(raw_keys.size() / keylen);
        for (int k = 0; k < Kr; ++k) {
            auto it = gm.find(std::vector<int>(raw_keys.begin() + (size_t)k * keylen, raw_keys.begin() + (size_t)(k + 1) * keylen));
            if (it != gm.end()) std::memcpy(&Eg[(size_t)it->second * M], &raw_E[(size_t)k * M], sizeof(double) * M);
        }
    } else {
        // two-population path: the joint-CSFS preparation works on the local key list only
        for (int k = 0; k < K; ++k) std::memcpy(&Eg[(size_t)local_to_global[k] * M], &E[(size_t)k * M], sizeof(double) * M);
        if (nder > 0) {
            dEg.assign((size_t)Kg * M * nder, NAN);
            for (int k = 0; k < K; ++k)
                std::memcpy(&dEg[(size_t)local_to_global[k] * M * nder], &dE[(size_t)k * M * nder], sizeof(double) * M * nder);
        }
    }
}

void smcpp_im::host_prep_and_upload() {
    ensure_T();            // (a lazily kept transition matrix is expanded HERE: behind the scan chains' launches, in front of its upload)
    hipStream_t s = stream;
    const bool tm = opt().has(smcpp_opt::O_HOST_TIMING);
    auto tp0 = std::chrono::steady_clock::now();
    const size_t MM = (size_t)Mp * Mp;
    const size_t em = std::max<size_t>(1, (size_t)Ke) * MM;
    // staging vectors live in the manager: allocated and zeroed once (only entries of real states are ever written, so
    // the padding stays zero), not ~0.5 MB of fresh zero-filled storage per E-step
    auto ensure = [](auto &v, size_t n, auto init) { if (v.size() != n) v.assign(n, init); };
    ensure(hs_PinvT, em, 0.0); ensure(hs_PT, em, 0.0); ensure(hs_Prm, em, 0.0); ensure(hs_Pinvrm, em, 0.0);
    ensure(hs_dsc, std::max<size_t>(1, (size_t)Ke) * Mp, 0.0); ensure(hs_dun, std::max<size_t>(1, (size_t)Ke) * Mp, 0.0);
    ensure(hs_gsc, (size_t)std::max(1, G), 1.0);
    ensure(hs_gls, (size_t)std::max(1, G), 0.0);
    ensure(hs_pi_f, (size_t)Mp, 0.f); ensure(hs_Tf, MM, 0.f);
    ensure(hs_TdT, MM, 0.0); ensure(hs_Td, MM, 0.0); ensure(hs_Ep, (size_t)K * Mp, 0.0);
    std::vector<double> &PinvT = hs_PinvT, &PT = hs_PT, &Prm = hs_Prm, &Pinvrm = hs_Pinvrm, &dsc = hs_dsc, &dun = hs_dun,
                        &gsc = hs_gsc, &gls = hs_gls, &TdT = hs_TdT, &Td = hs_Td, &Ep = hs_Ep;
    std::vector<float> &pi_f = hs_pi_f, &Tf = hs_Tf;
    // groups of each eigen key (so that one task finishes everything that depends on one eigensystem)
    std::vector<std::vector<int>> groups_of(Ke);
    for (int g = 0; g < G; ++g) groups_of[groups[g].eig].push_back(g);
    // ---- TransitionBundle::update: eigensystems of diag(b_k) Td^T per eigen key (transition_bundle.cpp:15-25), the
    // transposed / row-major copies the kernels read and the eigenvalue powers of every (span, key) group, ONE
    // parallel region (task Ke packs the key-independent arrays)
    std::string err;
    // scan chains + eigen-free statistics: nothing on the device reads the float / transposed copies of T or any eigenvector
    // matrix - they are neither packed nor staged nor copied (M = 256: 7.5 MB through the pinned arena, 1 ms of host time)
    const bool lean = eigfree && ss_active;
    auto pack_static = [&]() {
        if (static_packed) return;
        for (int i = 0; i < M; ++i) {
            pi_f[i] = (float)pi[i];
            if (lean) {
                for (int j = 0; j < M; ++j) Td[(size_t)i * Mp + j] = T[(size_t)i * M + j];
                continue;
            }
            for (int j = 0; j < M; ++j) {
                Tf[(size_t)i * Mp + j] = (float)T[(size_t)i * M + j];
                Td[(size_t)i * Mp + j] = T[(size_t)i * M + j];
                TdT[(size_t)j * Mp + i] = T[(size_t)i * M + j];
            }
        }
        if (E_on_dev) return;                  // (the device preparation wrote the table where the statistics read it)
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < M; ++i) Ep[(size_t)k * Mp + i] = E[(size_t)k * M + i];
    };
    auto make_A = [&](int e, std::vector<double> &A) {
        const double *b = &E[(size_t)eig_kid[e] * M];
        A.resize((size_t)M * M);
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < M; ++j) A[(size_t)i * M + j] = b[i] * T[(size_t)j * M + i];
    };
    // rows i = r0, r0 + step, ... of the device layouts of eigen key e; the eigenvalue powers of its groups with r0 == 0
    auto unpack = [&](int e, const smcpp_host::EigenSystem &s_, int r0, int step) {
        for (int i = r0; i < M; i += step) {
            dun[(size_t)e * Mp + i] = s_.d[i];
            dsc[(size_t)e * Mp + i] = s_.d[i] / s_.scale;
            for (int j = 0; j < M; ++j) {
                const double p = s_.P[(size_t)i * M + j], pi_ = s_.Pinv[(size_t)i * M + j];
                Prm[e * MM + (size_t)i * Mp + j] = p;
                PT[e * MM + (size_t)j * Mp + i] = p;
                Pinvrm[e * MM + (size_t)i * Mp + j] = pi_;
                PinvT[e * MM + (size_t)j * Mp + i] = pi_;
            }
        }
        if (r0 != 0) return;
        const double ls = std::log(s_.scale);
        for (int g : groups_of[e]) {
            const int sp = groups[g].span;
            gsc[g] = s_.scale;
            // (the eigenvalue powers (d_r / scale)^span of the group: k_group_dpow, on the device)
            // the scan steps apply the operator itself: their normalisers carry no eigenvalue scale (hybrid rows do: d / scale)
            gls[g] = (ss_active && !(ss_hybrid && sp > ss_hyb_th)) ? 0.0 : sp * ls;
        }
    };
    // M >= 128: a team of threads per eigen key (nonsym_eig_team.hpp: bit-identical to the serial routine); the size follows
    // the thread count the caller allows (smcpp_set_num_threads), SMCPP_EIG_TEAM overrides it (1 = serial routine)
    // Every team is confined to one L3 domain for the duration of the region (see nonsym_eig_team.hpp: unpinned on a
    // two-socket host the element hand-overs make it slower than the serial routine); without sysfs topology, or with
    // SMCPP_EIG_TEAM=1, the serial routine runs.
    static std::vector<std::vector<int>> l3;
    static std::once_flag l3_once;
    int team = (M >= 128 && Ke >= 1) ? std::min(8, omp_get_max_threads() / Ke) : 1;
    if (opt().has(smcpp_opt::O_EIG_TEAM)) team = std::max(1, std::min(16, opt().i(smcpp_opt::O_EIG_TEAM, 1)));
    if (M < 32) team = 1;
    if (team >= 2) {
        std::call_once(l3_once, [] { l3 = smcpp_host::cpu_l3_groups(); });     // a few hundred sysfs reads, once per process
        if ((int)l3.size() < Ke) team = 1;
    }
    bool team_done = false;
    if (eigfree) {
        // no eigensystem is needed anywhere in this E-step: only the key-independent arrays are packed
        pack_static();
        for (int g = 0; g < G; ++g) { gsc[g] = 1.0; gls[g] = 0.0; }
        team = 1;
        team_done = true;
    }
    if (team >= 2) {
        pack_static();
        if ((int)eig_teams.size() != Ke || eig_teams[0]->size != team) {
            eig_teams.clear();
            for (int e = 0; e < Ke; ++e) eig_teams.emplace_back(new smcpp_host::EigTeam(team));
        }
        std::vector<std::vector<double>> As(Ke);
        std::vector<smcpp_host::EigenSystem> ess(Ke);
        for (int e = 0; e < Ke; ++e) make_A(e, As[e]);
        bool ok = true;
        // L3 domains next to the one the calling thread runs in (same socket first: sysfs lists them in CPU order)
        int g0 = 0;
        {
            const int here = sched_getcpu();
            for (size_t g = 0; g < l3.size(); ++g)
                for (int c : l3[g]) if (c == here) g0 = (int)g;
        }
        const bool pin = !opt().off(smcpp_opt::O_EIG_PIN);
#pragma omp parallel num_threads(Ke * team)
        {
            if (omp_get_num_threads() != Ke * team) {
#pragma omp single
                ok = false;
            } else {
                const int tid = omp_get_thread_num(), e = tid / team, rank = tid % team;
                smcpp_host::ScopedAffinity aff(pin ? &l3[(size_t)(g0 + e) % l3.size()] : nullptr);
                smcpp_host::EigTeam &tm = *eig_teams[e];
                int gen = tm.generation.load(std::memory_order_acquire);
                smcpp_host::eigensystem_team(M, As[e], ess[e], tm, rank, gen);
                if (!tm.failed.load()) unpack(e, ess[e], rank, team);
            }
        }
        if (ok) {
            for (int e = 0; e < Ke; ++e)
                if (eig_teams[e]->failed.load()) err = eig_teams[e]->error.empty() ? "eigensolver failed" : eig_teams[e]->error;
            team_done = true;
        }
    }
    if (!team_done) {
#pragma omp parallel for schedule(dynamic) num_threads(std::max(1, std::min(Ke + 1, omp_get_max_threads())))
        for (int e = 0; e <= Ke; ++e) {
            if (e == Ke) { pack_static(); continue; }
            try {
                std::vector<double> A;
                make_A(e, A);
                const smcpp_host::EigenSystem s_ = smcpp_host::eigensystem(M, A);
                unpack(e, s_, 0, 1);
            } catch (const std::exception &ex) {
#pragma omp critical
                err = ex.what();
            }
        }
    }
    if (!err.empty()) throw std::runtime_error(err);
    auto tp1 = std::chrono::steady_clock::now();
    std::vector<float> qTf;
    std::vector<double> qTdT, qPinvT, qPT, qPrm, qPinvrm;
    if (Mp > 64 && chain_mode == 3 && !ss_active) {          // (the scan chains stream no operand)
        // quarter-interleaved streaming layouts  Q[t][i][kq] = Mt[(kq*KQ + t)*Mp + i]  (k_fwd_big / k_bwd_big)
        const int KQ = Mp / 4;
        qTf.assign(MM, 0.f); qTdT.assign(MM, 0.0);
        qPinvT.assign(em, 0.0); qPT.assign(em, 0.0); qPrm.assign(em, 0.0); qPinvrm.assign(em, 0.0);
#pragma omp parallel for schedule(static) num_threads(std::max(1, std::min(8, omp_get_max_threads())))
        for (int t = 0; t < KQ; ++t)
            for (int i = 0; i < Mp; ++i)
                for (int q = 0; q < 4; ++q) {
                    const size_t dst = ((size_t)t * Mp + i) * 4 + q, src = (size_t)(q * KQ + t) * Mp + i;
                    qTf[dst] = Tf[src];
                    qTdT[dst] = TdT[src];
                    for (int e = 0; e < Ke; ++e) {
                        qPinvT[e * MM + dst] = PinvT[e * MM + src];
                        qPT[e * MM + dst] = PT[e * MM + src];
                        qPrm[e * MM + dst] = Prm[e * MM + src];
                        qPinvrm[e * MM + dst] = Pinvrm[e * MM + src];
                    }
                }
    }
    // ---- one contiguous parameter arena on the device, mirrored in pinned host memory: ONE copy per E-step ----
    static const std::vector<double> none_d;
    static const std::vector<float> none_f;
    const std::vector<float> &uTf = lean ? none_f : Tf;
    const std::vector<double> &uTdT = lean ? none_d : TdT, &uPinvT = lean ? none_d : PinvT, &uPT = lean ? none_d : PT,
                              &uPrm = lean ? none_d : Prm, &uPinvrm = lean ? none_d : Pinvrm;
    size_t need = 32 * 256;
    need += qTf.size() * 4 + (qTdT.size() + qPinvT.size() + qPT.size() + qPrm.size() + qPinvrm.size()) * 8;
    need += (pi_f.size() + uTf.size()) * 4;
    need += (uTdT.size() + Td.size() + (E_on_dev ? 0 : Ep.size()) + uPinvT.size() + uPT.size() + uPrm.size() + uPinvrm.size() + dsc.size() +
             dun.size() + gsc.size() + gls.size()) * 8;
    stage.reset(need);
    if (need > param_cap) {
        if (d_param) (void)hipFree(d_param);
        param_cap = need + need / 4;
        HIPCHK(hipMalloc((void **)&d_param, param_cap));
        smcpp_opt::poison(d_param, param_cap, __LINE__, __FILE__);
    }
    size_t off = 0;
    char *hb = stage.base;
    d_pi_f.place(pi_f, d_param, hb, off); d_Tf.place(uTf, d_param, hb, off); d_TdT.place(uTdT, d_param, hb, off);
    d_Td.place(Td, d_param, hb, off);
    if (E_on_dev) {
        if (d_E.p && !d_E.borrowed) (void)hipFree(d_E.p);
        d_E.p = dprep->d_El.p; d_E.n = (size_t)K * Mp; d_E.borrowed = true;
    } else d_E.place(Ep, d_param, hb, off);
    d_PinvT.place(uPinvT, d_param, hb, off); d_PT.place(uPT, d_param, hb, off); d_Prm.place(uPrm, d_param, hb, off);
    d_Pinvrm.place(uPinvrm, d_param, hb, off);
    d_dsc.place(dsc, d_param, hb, off); d_dun.place(dun, d_param, hb, off);
    d_g_scale.place(gsc, d_param, hb, off); d_g_logscale.place(gls, d_param, hb, off);
    if (!qTf.empty()) {
        d_qTf.place(qTf, d_param, hb, off); d_qTdT.place(qTdT, d_param, hb, off);
        d_qPinvT.place(qPinvT, d_param, hb, off); d_qPT.place(qPT, d_param, hb, off);
        d_qPrm.place(qPrm, d_param, hb, off); d_qPinvrm.place(qPinvrm, d_param, hb, off);
    }
    if (off > need) throw std::runtime_error("internal: parameter arena overflow");
    auto tp2 = std::chrono::steady_clock::now();
    arena_side = lean && stream2 != nullptr && dual_stream;
    if (arena_side) {
        // nothing the chains read lives in this arena: the copy runs beside them, the statistics wait for it
        HIPCHK(hipMemcpyAsync(d_param, hb, off, hipMemcpyHostToDevice, stream2));
        HIPCHK(hipEventRecord(ev[20], stream2));
    } else
    HIPCHK(hipMemcpyAsync(d_param, hb, off, hipMemcpyHostToDevice, s));
    // (d_r / scale)^span for every (span, eigen key) group: G x M calls of pow() - 2 ms of host time on data with a few
    // thousand distinct spans, microseconds here
    if (G > 0 && !eigfree)
        hipLaunchKernelGGL(k_group_dpow, dim3(ceil_div((long long)G * Mp, 256)), dim3(256), 0, s, G, M, Mp, (const int *)d_g_span.p,
                           (const int *)d_g_eig.p, (const double *)d_dsc.p, d_dpow.p);
    {
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        host_timing[1] = ms(tp0, tp1);
        host_timing[2] = ms(tp1, std::chrono::steady_clock::now());
    }
    if (tm) {
        auto tp3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[host] eigensystems+packing %.3f ms, layouts+staging %.3f ms, copy enqueue %.3f ms (%zu bytes)\n",
                ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3), off);
    }
    // no synchronisation: the copies read the pinned arena, which lives until the next E-step resets it
}

