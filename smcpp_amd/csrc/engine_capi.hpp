// engine_capi.hpp - part of the ONE translation unit engine.hip (included there, in order; not a standalone header):
// the C ABI of include/smcpp_engine.h.
// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
#define API_BEGIN try {
#define API_END                                                                    \
    return 0;                                                                      \
    }                                                                              \
    catch (const std::exception &e) { g_err = e.what(); return 1; }               \
    catch (...) { g_err = "unknown error"; return 1; }

extern "C" {

const char *smcpp_last_error(void) { return g_err.c_str(); }

int smcpp_create_onepop(int n, int n_contigs, const int *Ls, const int *const *obs, int n_hs, const double *hs,
                        double polarization_error, int device, smcpp_im **out) {
    API_BEGIN
    std::unique_ptr<smcpp_im> im(new smcpp_im());
    const int nn[1] = {n}, nna[1] = {2};
    im->build(1, nn, nna, n_contigs, Ls, obs, n_hs, hs, polarization_error, device);
    *out = im.release();
    API_END
}

int smcpp_create_twopop(int n1, int n2, int a1, int a2, int n_contigs, const int *Ls, const int *const *obs,
                        int n_hs, const double *hs, double polarization_error, int device, smcpp_im **out) {
    API_BEGIN
    if (a1 == 0 && a2 == 2) throw std::runtime_error("(0,2) not supported");
    if (a1 + a2 != 2) throw std::runtime_error("configuration not supported");
    std::unique_ptr<smcpp_im> im(new smcpp_im());
    const int nn[2] = {n1, n2}, nna[2] = {a1, a2};
    im->build(2, nn, nna, n_contigs, Ls, obs, n_hs, hs, polarization_error, device);
    *out = im.release();
    API_END
}

int smcpp_rccl_destroy(smcpp_im *im);
void smcpp_destroy(smcpp_im *im) { if (im && im->rccl) (void)smcpp_rccl_destroy(im); delete im; }

int smcpp_set_theta(smcpp_im *im, double v) { API_BEGIN im->params_fresh = false; im->theta = v; im->dirty = true; if (im->have_model) im->have_raw = false; API_END }
int smcpp_set_rho(smcpp_im *im, double v) { API_BEGIN im->params_fresh = false; im->rho = v; im->dirty = true; if (im->have_model) im->have_raw = false; API_END }
int smcpp_set_alpha(smcpp_im *im, double v) { API_BEGIN im->params_fresh = false; im->alpha = v; im->dirty = true; if (im->have_model) im->have_raw = false; API_END }

int smcpp_set_params(smcpp_im *im, int K, const double *a, const double *da, int nder, const double *s) {
    API_BEGIN
    if (K <= 0) throw std::runtime_error("empty parameter vector");
    for (int k = 0; k < K; ++k)
        if (!(a[k] > 0)) throw std::runtime_error("model pieces must be positive");
    if (nder > smcpp_host::MAXD) throw std::runtime_error("too many derivative directions (max 64)");
    im->model.a.assign(a, a + K);
    im->model.s.assign(s, s + K);
    im->nder = (da && nder > 0) ? nder : 0;
    im->model_da.clear();
    if (im->nder) im->model_da.assign(da, da + (size_t)K * nder);
    im->params_fresh = false;
    im->have_model = true;
    im->have_raw = false;
    im->dirty = true;
    API_END
}

int smcpp_set_params_twopop(smcpp_im *im, int Kd, const double *ad, const double *sd, const double *dad, int K1,
                            const double *a1, const double *s1, const double *da1, int K2, const double *a2,
                            const double *s2, const double *da2, double split, int nder) {
    API_BEGIN
    if (im->npop != 2) throw std::runtime_error("set_params_twopop on a one-population manager");
    if (Kd <= 0 || K1 <= 0 || K2 <= 0) throw std::runtime_error("empty parameter vector");
    if (!(split >= 0)) throw std::runtime_error("split time must be >= 0");
    if (nder > smcpp_host::MAXD) throw std::runtime_error("too many derivative directions (max 64)");
    auto chk = [](int K, const double *a) {
        for (int k = 0; k < K; ++k)
            if (!(a[k] > 0)) throw std::runtime_error("model pieces must be positive");
    };
    chk(Kd, ad); chk(K1, a1); chk(K2, a2);
    im->model.a.assign(ad, ad + Kd); im->model.s.assign(sd, sd + Kd);
    im->model_p1.a.assign(a1, a1 + K1); im->model_p1.s.assign(s1, s1 + K1);
    im->model_p2.a.assign(a2, a2 + K2); im->model_p2.s.assign(s2, s2 + K2);
    im->split = split;
    im->nder = nder > 0 ? nder : 0;
    im->model_da.clear(); im->model_da1.clear(); im->model_da2.clear();
    if (im->nder) {
        if (dad) im->model_da.assign(dad, dad + (size_t)Kd * nder);
        if (da1) im->model_da1.assign(da1, da1 + (size_t)K1 * nder);
        if (da2) im->model_da2.assign(da2, da2 + (size_t)K2 * nder);
    }
    im->params_fresh = false;
    im->have_model = true;
    im->have_raw = false;
    im->dirty = true;
    API_END
}

int smcpp_set_prep_mode(smcpp_im *im, int host) {
    API_BEGIN
    im->force_host_prep = host != 0;
    im->params_fresh = false;
    im->dirty = true;
    API_END
}

int smcpp_set_warm_start(smcpp_im *im, int on) {
    API_BEGIN
    im->warm_start = on != 0;
    if (!on) im->warm_valid = false;
    API_END
}

int smcpp_set_raw(smcpp_im *im, const double *pi, const double *T, int K, const int *keys, const double *E) {
    API_BEGIN
    const int M = im->M, kl = im->keylen;
    std::map<std::vector<int>, int> given;
    for (int k = 0; k < K; ++k) given[std::vector<int>(keys + (size_t)k * kl, keys + (size_t)(k + 1) * kl)] = k;
    std::vector<double> Enew((size_t)im->K * M);
    for (int k = 0; k < im->K; ++k) {
        std::vector<int> key(im->keys.begin() + (size_t)k * kl, im->keys.begin() + (size_t)(k + 1) * kl);
        auto it = given.find(key);
        if (it == given.end()) throw std::runtime_error("set_raw: an observed key has no emission vector");
        std::memcpy(&Enew[(size_t)k * M], E + (size_t)it->second * M, sizeof(double) * M);
    }
    im->pi.assign(pi, pi + M);
    im->T.assign(T, T + (size_t)M * M);
    im->E.swap(Enew);
    im->raw_keys.assign(keys, keys + (size_t)K * kl);
    im->raw_E.assign(E, E + (size_t)K * M);
    im->have_raw = true;
    im->E_on_dev = false;
    im->tgen_valid = false; im->dT_valid = true; im->T_lazy = false;
    im->dirty = true;
    im->nder = 0;
    API_END
}

int smcpp_estep(smcpp_im *im, int fb_only) {
    API_BEGIN
    (void)fb_only;   // accepted and ignored, as in the reference (hmm.cpp:45)
    im->estep();
    API_END
}

int smcpp_loglik(smcpp_im *im, double *out) {
    API_BEGIN
    std::memcpy(out, im->loglik.data(), sizeof(double) * im->n_contigs);    // 0 before the first E-step (hmm.cpp:11: ll(0.))
    API_END
}

static double dcs(const std::vector<double> &x) {   // doubly_compensated_summation, common.h:27-46
    if (x.empty()) return 0.0;
    double s = x[0], c = 0.0;
    for (size_t i = 1; i < x.size(); ++i) {
        const double y = c + x[i];
        const double u = x[i] - (y - c);
        const double t = y + s;
        const double v = y - (t - s);
        const double z = u + v;
        s = t + z;
        c = z - (s - t);
    }
    return s;
}

int smcpp_q(smcpp_im *im, double val[4], double *jac) {
    API_BEGIN
    const int M = im->M, K = im->K;
    if (!im->have_raw) im->prepare_params();   // Q() does do_dirty_work() first (inference_manager.cpp:119)
    if (im->q_device(val, jac)) return 0;
    im->sync_host_E();
    im->ensure_dT();
    if ((int)im->pi.size() != M) throw std::runtime_error("parameters are not set");
    const int nder = im->have_raw ? 0 : im->nder;
    if (jac) for (int i = 0; i < 4 * nder; ++i) jac[i] = 0.0;
    for (int i = 0; i < 4; ++i) val[i] = 0.0;
    im->ensure_T();
    std::vector<double> logpi(M), logT((size_t)M * M), logE((size_t)K * M);
    for (int i = 0; i < M; ++i) logpi[i] = std::log(im->pi[i]);
    for (size_t i = 0; i < logT.size(); ++i) logT[i] = std::log(im->T[i]);
    std::vector<unsigned char> bad(K, 0);
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < M; ++i) {
            if (im->E[(size_t)k * M + i] <= 0.0) bad[k] = 1;
            logE[(size_t)k * M + i] = std::log(im->E[(size_t)k * M + i]);
        }
    // d/d(seed) of sum w log x = sum (w / x) dx  (forward-mode derivatives of hmm.cpp:161-185)
    auto add_jac = [&](int term, const double *w, const double *x, const double *dx, size_t cnt) {
        if (!jac || nder == 0) return;
        for (size_t i = 0; i < cnt; ++i) {
            const double f = w[i] / x[i];
            for (int d = 0; d < nder; ++d) jac[term * nder + d] += f * dx[i * nder + d];
        }
    };
    if (im->have_reduced) {
        // statistics already summed over every rank's contigs: every global key contributes, also those no contig of
        // this rank holds (their emission vectors come from the same preparation, see prepare_params)
        const double *g0 = &im->g_stats[1], *xs = g0 + M, *gs = xs + (size_t)M * M;
        const int Kg = (int)(im->gkeys.size() / im->keylen), kl = im->keylen;
        im->global_emissions();
        for (int i = 0; i < M; ++i) val[0] += logpi[i] * g0[i];
        add_jac(0, g0, im->pi.data(), im->dpi.data(), M);
        std::vector<double> b0, b1;
        bool inf0 = false, inf1 = false;
        for (int kg = 0; kg < Kg; ++kg) {
            const double *e = &im->Eg[(size_t)kg * M], *g = gs + (size_t)kg * M;
            int nb = 0;
            for (int p = 0; p < im->npop; ++p) nb += im->gkeys[(size_t)kg * kl + 3 * p + 2];
            bool any = false, nan = false, nonpos = false;
            for (int i = 0; i < M; ++i) { any = any || g[i] != 0.0; nan = nan || std::isnan(e[i]); nonpos = nonpos || e[i] <= 0.0; }
            if (!any) continue;                          // no contig anywhere holds the key (hmm.cpp:166-181 skips it too)
            if (nan) throw std::runtime_error("Q on all-reduced statistics: no emission vector for a key that another "
                                              "rank's contigs hold (set_raw must supply every global key)");
            if (nonpos) { (nb > 0 ? inf1 : inf0) = true; continue; }
            auto &b = nb > 0 ? b1 : b0;
            for (int i = 0; i < M; ++i) b.push_back(std::log(e[i]) * g[i]);
            if (nder) {
                if (im->dEg.empty()) throw std::runtime_error("Q gradient on all-reduced statistics needs model parameters (set_params)");
                add_jac(nb > 0 ? 2 : 1, g, e, &im->dEg[(size_t)kg * M * nder], M);
            }
        }
        val[1] = inf0 ? -INFINITY : dcs(b0);
        val[2] = inf1 ? -INFINITY : dcs(b1);
        std::vector<double> es((size_t)M * M);
        for (int j = 0; j < M; ++j)
            for (int i = 0; i < M; ++i) es[(size_t)j * M + i] = logT[(size_t)i * M + j] * xs[(size_t)i * M + j];
        val[3] = dcs(es);
        add_jac(3, xs, im->T.data(), im->dT.data(), (size_t)M * M);
        return 0;
    }
    im->fetch_stats();
    for (int c = 0; c < im->n_contigs; ++c) {
        double q0 = 0.0;
        for (int i = 0; i < M; ++i) q0 += logpi[i] * im->h_gamma0[(size_t)c * M + i];
        val[0] += q0;
        add_jac(0, &im->h_gamma0[(size_t)c * M], im->pi.data(), im->dpi.data(), M);
        std::vector<double> b0, b1;
        bool inf0 = false, inf1 = false;
        for (int k = 0; k < K; ++k) {
            if (!im->present[(size_t)c * K + k]) continue;
            if (bad[k]) { (im->key_nbpos[k] ? inf1 : inf0) = true; continue; }
            auto &b = im->key_nbpos[k] ? b1 : b0;
            for (int i = 0; i < M; ++i)
                b.push_back(logE[(size_t)k * M + i] * im->h_gsum[((size_t)c * K + k) * M + i]);
            add_jac(im->key_nbpos[k] ? 2 : 1, &im->h_gsum[((size_t)c * K + k) * M], &im->E[(size_t)k * M],
                    nder ? &im->dE[(size_t)k * M * nder] : nullptr, M);
        }
        val[1] += inf0 ? -INFINITY : dcs(b0);
        val[2] += inf1 ? -INFINITY : dcs(b1);
        std::vector<double> es((size_t)M * M);
        const double *xs = &im->h_xisum[(size_t)c * M * M];
        for (int j = 0; j < M; ++j)
            for (int i = 0; i < M; ++i) es[(size_t)j * M + i] = logT[(size_t)i * M + j] * xs[(size_t)i * M + j];
        val[3] += dcs(es);
        add_jac(3, xs, im->T.data(), im->dT.data(), (size_t)M * M);
    }
    API_END
}

int smcpp_set_save_gamma(smcpp_im *im, int on) { API_BEGIN im->save_gamma = on != 0; API_END }
int smcpp_get_save_gamma(smcpp_im *im) { return im->save_gamma ? 1 : 0; }
int smcpp_num_states(smcpp_im *im) { return im->M; }
int smcpp_num_contigs(smcpp_im *im) { return im->n_contigs; }
int smcpp_num_keys(smcpp_im *im) { return im->K; }
int smcpp_key_len(smcpp_im *im) { return im->keylen; }

int smcpp_get_hidden_states(smcpp_im *im, double *hs) {
    API_BEGIN std::memcpy(hs, im->hs.data(), sizeof(double) * im->hs.size()); API_END
}
int smcpp_set_hidden_states(smcpp_im *im, int n_hs, const double *hs) {
    API_BEGIN
    if (n_hs != (int)im->hs.size()) throw std::runtime_error("hidden states must be same size");
    im->hs.assign(hs, hs + n_hs);
    im->update_pi_default();
    im->twopop_prep.reset();
    if (!im->estep_done) im->stats_on_host = false;
    if (im->qdev) im->qdev->stats_ready = false;      // the pre-E-step statistics are span_sum * pi_default: restage them
    im->dirty = true;
    im->params_fresh = false;
    if (im->have_model) im->have_raw = false;
    API_END
}
int smcpp_get_keys(smcpp_im *im, int *keys) {
    API_BEGIN std::memcpy(keys, im->keys.data(), sizeof(int) * im->keys.size()); API_END
}

int smcpp_get_xisum(smcpp_im *im, int c, double *out) {
    API_BEGIN
    if (c < 0 || c >= im->n_contigs) throw std::runtime_error("contig index out of range");
    im->fetch_stats();
    std::memcpy(out, &im->h_xisum[(size_t)c * im->M * im->M], sizeof(double) * im->M * im->M);
    API_END
}

// The caller's rows of contig c with the pieces of every cut row added up, on the device ([Lu + 1][Mp]; row 0 is left to the caller).
const double *smcpp_im::merged_gamma(int c) {
    const int Lu = user_Ls[c];
    if (d_piece_first.size() != (size_t)n_contigs) d_piece_first.resize(n_contigs);
    if (!d_piece_first[c].p) {
        std::vector<int> first((size_t)Lu + 2, 0);
        const std::vector<int> &pr = piece_row[c];                 // piece -> caller's row (non-decreasing)
        for (int l = (int)pr.size() - 1; l >= 1; --l) first[pr[l]] = l;
        first[Lu + 1] = Ls[c] + 1;
        d_piece_first[c].upload(first, stream);
        HIPCHK(hipStreamSynchronize(stream));          // (`first` is a local: the copy has to be done before it goes)
    }
    d_gamma_user.alloc((size_t)(Lu + 1) * Mp);
    hipLaunchKernelGGL(k_gamma_merge, dim3((unsigned)ceil_div((long long)Lu * Mp, 256)), dim3(256), 0, stream, Mp, Lu,
                       (const int *)d_piece_first[c].p, (const double *)(d_gamma_rows.p + (size_t)contig_base[c] * Mp), d_gamma_user.p);
    return d_gamma_user.p;
}

int smcpp_get_gamma(smcpp_im *im, int c, double *out) {
    API_BEGIN
    if (c < 0 || c >= im->n_contigs) throw std::runtime_error("contig index out of range");
    const int M = im->M, Mp = im->Mp;
    im->fetch_stats();
    if (!im->gamma_valid) {
        std::memcpy(out, &im->h_gamma0[(size_t)c * M], sizeof(double) * M);   // gamma is M x 1 (hmm.cpp:12-14)
        return 0;
    }
    HIPCHK(hipSetDevice(im->device));
    if (im->split_spans) {
        // the pieces of a long row add up to the row's posterior (engine_manager.hpp: build): added on the device, then as below
        const int Lu = im->user_Ls[c];
        const double *src = im->merged_gamma(c);
        std::vector<double> rows((size_t)(Lu + 1) * Mp);
        HIPCHK(hipMemcpyAsync(rows.data(), src, rows.size() * sizeof(double), hipMemcpyDeviceToHost, im->stream));
        HIPCHK(hipStreamSynchronize(im->stream));
        for (int i = 0; i < M; ++i) {
            out[(size_t)i * (Lu + 1)] = im->h_gamma0[(size_t)c * M + i];
            for (int l = 1; l <= Lu; ++l) out[(size_t)i * (Lu + 1) + l] = rows[(size_t)l * Mp + i];
        }
        return 0;
    }
    const int L = im->Ls[c];
    std::vector<double> rows((size_t)(L + 1) * Mp);
    HIPCHK(hipMemcpy(rows.data(), im->d_gamma_rows.p + (size_t)im->contig_base[c] * Mp, rows.size() * sizeof(double),
                     hipMemcpyDeviceToHost));
    for (int i = 0; i < M; ++i) {
        out[(size_t)i * (L + 1)] = im->h_gamma0[(size_t)c * M + i];
        for (int l = 1; l <= L; ++l) out[(size_t)i * (L + 1) + l] = rows[(size_t)l * Mp + i];
    }
    API_END
}

int smcpp_gamma_cols(smcpp_im *im, int c) {
    if (c < 0 || c >= im->n_contigs) return -1;
    return im->gamma_valid ? im->user_Ls[c] + 1 : 1;
}

int smcpp_get_gamma_argmax(smcpp_im *im, int c, int *out) {
    API_BEGIN
    if (c < 0 || c >= im->n_contigs) throw std::runtime_error("contig index out of range");
    if (!im->gamma_valid) throw std::runtime_error("save_gamma was not set for the last E-step");
    HIPCHK(hipSetDevice(im->device));
    const int L = im->Ls[c];
    im->fetch_stats();
    // (rows cut into pieces: the pieces' posteriors are added up on the device first; L = the caller's row count then)
    const double *src = im->split_spans ? im->merged_gamma(c) : (const double *)(im->d_gamma_rows.p + (size_t)im->contig_base[c] * im->Mp);
    const int Lc = im->split_spans ? im->user_Ls[c] : L;
    im->d_argmax.alloc((size_t)im->total_rows);
    hipLaunchKernelGGL(k_gamma_argmax, dim3(ceil_div(Lc + 1, 256)), dim3(256), 0, im->stream, im->M, im->Mp,
                       (long long)(Lc + 1), src, im->d_argmax.p);
    HIPCHK(hipMemcpyAsync(out, im->d_argmax.p, sizeof(int) * (Lc + 1), hipMemcpyDeviceToHost, im->stream));
    HIPCHK(hipStreamSynchronize(im->stream));
    // column 0 is alpha_0 o beta_0 (hmm.cpp:150), which lives in gamma0
    int best = 0;
    for (int i = 1; i < im->M; ++i)
        if (im->h_gamma0[(size_t)c * im->M + i] > im->h_gamma0[(size_t)c * im->M + best]) best = i;
    out[0] = best;
    API_END
}

int smcpp_get_gamma_sums(smcpp_im *im, int c, double *vals, unsigned char *present) {
    API_BEGIN
    if (c < 0 || c >= im->n_contigs) throw std::runtime_error("contig index out of range");
    im->fetch_stats();
    std::memcpy(vals, &im->h_gsum[(size_t)c * im->K * im->M], sizeof(double) * im->K * im->M);
    std::memcpy(present, &im->present[(size_t)c * im->K], im->K);
    API_END
}

int smcpp_get_pi(smcpp_im *im, double *out) {
    API_BEGIN
    if (im->pi.empty()) throw std::runtime_error("parameters are not set");
    std::memcpy(out, im->pi.data(), sizeof(double) * im->M);
    API_END
}
int smcpp_get_transition(smcpp_im *im, double *out) {
    API_BEGIN
    im->ensure_T();
    if (im->T.empty()) throw std::runtime_error("parameters are not set");
    std::memcpy(out, im->T.data(), sizeof(double) * im->M * im->M);
    API_END
}
int smcpp_get_emission_probs(smcpp_im *im, double *out) {
    API_BEGIN
    im->sync_host_E();
    if (im->E.empty()) throw std::runtime_error("parameters are not set");
    std::memcpy(out, im->E.data(), sizeof(double) * im->K * im->M);
    API_END
}

// ---- derivative-carrying getters (what the binding wraps into ad numbers, _smcpp.pyx:103-120,215-275) ----
static void need_model_params(smcpp_im *im) {
    if (im->have_raw) throw std::runtime_error("parameters were set with set_raw: no model, no derivatives");
    im->prepare_params();
}
int smcpp_get_pi_jac(smcpp_im *im, double *out) {
    API_BEGIN
    need_model_params(im);
    if (im->nder > 0) std::memcpy(out, im->dpi.data(), sizeof(double) * im->dpi.size());
    API_END
}
int smcpp_get_transition_jac(smcpp_im *im, double *out) {
    API_BEGIN
    need_model_params(im);
    im->ensure_dT();
    if (im->nder > 0) std::memcpy(out, im->dT.data(), sizeof(double) * im->dT.size());
    API_END
}
int smcpp_get_emission_probs_jac(smcpp_im *im, double *out) {
    API_BEGIN
    need_model_params(im);
    im->sync_host_E();
    if (im->nder > 0) std::memcpy(out, im->dE.data(), sizeof(double) * im->dE.size());
    API_END
}
int smcpp_num_emission_cols(smcpp_im *im) {
    int cols = 1;
    for (int p = 0; p < im->npop; ++p) cols *= (im->na[p] + 1) * (im->n[p] + 1);
    return cols;
}
int smcpp_get_emission(smcpp_im *im, double *out, double *jac) {
    API_BEGIN
    need_model_params(im);
    im->sync_host_E();
    if (im->emission.size() != (size_t)im->M * smcpp_num_emission_cols(im)) throw std::runtime_error("emission matrix is not available");
    std::memcpy(out, im->emission.data(), sizeof(double) * im->emission.size());
    if (jac && im->nder > 0) std::memcpy(jac, im->demission.data(), sizeof(double) * im->demission.size());
    API_END
}

void smcpp_init_logger_cb(void (*cb)(const char *, const char *, const char *)) { g_logger_cb = cb; }

int smcpp_init_cache(const char *path) {
    API_BEGIN
    smcpp_host::csfs_cache_prefix() = path ? path : "";
    API_END
}

int smcpp_set_global_keys(smcpp_im *im, int Kg, const int *gkeys) {
    API_BEGIN
    const int kl = im->keylen;
    std::map<std::vector<int>, int> gm;
    for (int k = 0; k < Kg; ++k) gm[std::vector<int>(gkeys + (size_t)k * kl, gkeys + (size_t)(k + 1) * kl)] = k;
    im->local_to_global.assign(im->K, -1);
    for (int k = 0; k < im->K; ++k) {
        auto it = gm.find(std::vector<int>(im->keys.begin() + (size_t)k * kl, im->keys.begin() + (size_t)(k + 1) * kl));
        if (it == gm.end()) throw std::runtime_error("global key list misses a local key");
        im->local_to_global[k] = it->second;
    }
    im->gkeys.assign(gkeys, gkeys + (size_t)Kg * kl);
    im->have_global = true;
    im->pack_tables_ready = false;
    if (im->dprep) im->dprep->keys_ready = false;
    if (im->qdev) im->qdev->stats_ready = false;
    im->E_on_dev = false;
    im->params_fresh = false;              // the emission table is now prepared over the global key list
    im->Eg.clear(); im->dEg.clear();
    API_END
}

int smcpp_pack_stats(smcpp_im *im, double *buf, long *n_out, int dev) {
    API_BEGIN
    const int M = im->M, K = im->K;
    const int Kg = im->have_global ? (int)(im->gkeys.size() / im->keylen) : K;
    const long n = 1 + M + (long)M * M + (long)Kg * M;
    if (n_out) *n_out = n;
    if (!buf) return 0;
    if (dev && !im->estep_done) throw std::runtime_error("no E-step has been run on this manager yet");
    if (dev) {
        // device path: one kernel writes the packed layout into the caller's device buffer (e.g. the tensor that is
        // all-reduced over RCCL) - no host round trip
        HIPCHK(hipSetDevice(im->device));
        if (!im->pack_tables_ready) {
            std::vector<int> g2l(Kg, -1);
            for (int k = 0; k < K; ++k) g2l[im->have_global ? im->local_to_global[k] : k] = k;
            im->d_g2l.upload(g2l, im->stream);
            im->d_present.upload(im->present, im->stream);
            HIPCHK(hipStreamSynchronize(im->stream));
            im->pack_tables_ready = true;
        }
        PackArgs pa;
        pa.M = M; pa.Mp = im->Mp; pa.K = K; pa.Kg = Kg; pa.n_contigs = im->n_contigs;
        pa.loglik = im->d_loglik.p; pa.gamma0 = im->d_gamma0.p; pa.xisum = im->d_xisum.p; pa.gsum = im->d_gsum.p;
        pa.present = im->d_present.p; pa.g2l = im->d_g2l.p; pa.out = buf;
        hipLaunchKernelGGL(k_pack_stats, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, im->stream, pa);
        HIPCHK(hipGetLastError());
        // dev == 2: stream-ordered hand-over - the caller consumes `buf` on the engine's stream (smcpp_stream), e.g. an RCCL
        // all-reduce enqueued behind the pack kernel, so there is no host wait between the E-step and the collective
        if (dev != 2) HIPCHK(hipStreamSynchronize(im->stream));
        return 0;
    }
    im->fetch_stats();
    std::vector<double> h(n, 0.0);
    for (int c = 0; c < im->n_contigs; ++c) {
        h[0] += im->loglik[c];
        for (int i = 0; i < M; ++i) h[1 + i] += im->h_gamma0[(size_t)c * M + i];
        for (size_t i = 0; i < (size_t)M * M; ++i) h[1 + M + i] += im->h_xisum[(size_t)c * M * M + i];
        for (int k = 0; k < K; ++k) {
            if (!im->present[(size_t)c * K + k]) continue;
            const int kg = im->have_global ? im->local_to_global[k] : k;
            for (int i = 0; i < M; ++i) h[1 + M + (size_t)M * M + (size_t)kg * M + i] += im->h_gsum[((size_t)c * K + k) * M + i];
        }
    }
    std::memcpy(buf, h.data(), sizeof(double) * n);
    API_END
}

int smcpp_unpack_stats(smcpp_im *im, const double *buf, long n, int dev) {
    API_BEGIN
    const int M = im->M;
    const int Kg = im->have_global ? (int)(im->gkeys.size() / im->keylen) : im->K;
    if (n != 1 + M + (long)M * M + (long)Kg * M) throw std::runtime_error("unpack_stats: wrong buffer length");
    if (!im->have_global) {
        im->local_to_global.resize(im->K);
        for (int k = 0; k < im->K; ++k) im->local_to_global[k] = k;
        im->gkeys = im->keys;
    }
    im->g_stats.resize(n);
    if (dev) {
        HIPCHK(hipSetDevice(im->device));
        HIPCHK(hipMemcpy(im->g_stats.data(), buf, sizeof(double) * n, hipMemcpyDeviceToHost));
    } else std::memcpy(im->g_stats.data(), buf, sizeof(double) * n);
    im->have_reduced = true;
    if (im->qdev) im->qdev->stats_ready = false;
    API_END
}

}  // extern "C"
