// engine_plans.hpp - part of the ONE translation unit engine.hip (included there, in order; not a standalone header):
// launch plans: chain families, the scan chains' fixed point, statistics, the E-step.
// ---------------------------------------------------------------------------------------------------------------
// kernel launches
// ---------------------------------------------------------------------------------------------------------------
// the cooperative chains (chains2.hpp): pass 0 and the re-run passes are separate instantiations
template <int MT_, bool TAB_, bool RERUN_, bool HOT2_>
static void launch_chain_coop2_tt(bool fwd, const ChainArgs &a, const CoopArgs &ca, size_t shm, hipStream_t s) {
    if (fwd) {
        static bool once = false;
        if (!once) { HIPCHK(hipFuncSetAttribute((const void *)k_fwd_coop2<MT_, TAB_, RERUN_, HOT2_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; }
        hipLaunchKernelGGL((k_fwd_coop2<MT_, TAB_, RERUN_, HOT2_>), dim3(a.nchunks), dim3(MT_ * 4), shm, s, a, ca);
    } else {
        static bool once = false;
        if (!once) { HIPCHK(hipFuncSetAttribute((const void *)k_bwd_coop2<MT_, TAB_, RERUN_, HOT2_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; }
        hipLaunchKernelGGL((k_bwd_coop2<MT_, TAB_, RERUN_, HOT2_>), dim3(a.nchunks), dim3(MT_ * 4), shm, s, a, ca);
    }
}
template <int MT_, bool TAB_>
static void launch_chain_power_t(bool fwd, const ChainArgs &a, const CoopArgs &ca, size_t shm, hipStream_t s) {
    if (fwd) {
        static bool once = false;
        if (!once) { HIPCHK(hipFuncSetAttribute((const void *)k_fwd_coop2<MT_, TAB_, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; }
        hipLaunchKernelGGL((k_fwd_coop2<MT_, TAB_, false, false, true>), dim3(a.nchunks), dim3(MT_ * 4), shm, s, a, ca);
    } else {
        static bool once = false;
        if (!once) { HIPCHK(hipFuncSetAttribute((const void *)k_bwd_coop2<MT_, TAB_, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; }
        hipLaunchKernelGGL((k_bwd_coop2<MT_, TAB_, false, false, true>), dim3(a.nchunks), dim3(MT_ * 4), shm, s, a, ca);
    }
}
template <int MT_, bool TAB_, bool RERUN_>
static void launch_chain_coop2_t(bool fwd, const ChainArgs &a, const CoopArgs &ca, size_t shm, hipStream_t s) {
    // without a second eigen key the 64 VGPRs of its operands are not allocated (measured on the whole genome: keeping
    // it beats the extra occupancy, 33.2 vs 36.6 ms - the L2 path of a non-resident key costs more than a lost wavefront)
    if (a.hot2 >= 0) launch_chain_coop2_tt<MT_, TAB_, RERUN_, true>(fwd, a, ca, shm, s);
    else launch_chain_coop2_tt<MT_, TAB_, RERUN_, false>(fwd, a, ca, shm, s);
}
template <int MT_, bool TAB_>
static void launch_chain_coop_t(bool fwd, const ChainArgs &a, const CoopArgs &ca, size_t shm, hipStream_t s) {
    if (a.variant == 1) launch_chain_power_t<MT_, TAB_>(fwd, a, ca, shm, s);
    else if (a.pass > 0 && a.variant != 2) launch_chain_coop2_t<MT_, TAB_, true>(fwd, a, ca, shm, s);
    else launch_chain_coop2_t<MT_, TAB_, false>(fwd, a, ca, shm, s);
}
static bool launch_chain_coop(bool fwd, int Mp, const ChainArgs &a, const CoopArgs &ca, int tab, size_t shm, hipStream_t s) {
    switch (Mp) {
#define C_(x) case x: if (tab) launch_chain_coop_t<x, true>(fwd, a, ca, shm, s); else launch_chain_coop_t<x, false>(fwd, a, ca, shm, s); return true;
        C_(16) C_(32) C_(48) C_(64)
#undef C_
        default: return false;
    }
}

template <int MT_>
static void launch_chain_big_t(bool fwd, const ChainArgs &a, const BigArgs &qa, hipStream_t s) {
    if (a.variant == 1) {          // eigen-free pre-pass
        if (fwd) hipLaunchKernelGGL((k_fwd_big<MT_, true>), dim3(a.nchunks), dim3(MT_ * 4), 0, s, a, qa);
        else hipLaunchKernelGGL((k_bwd_big<MT_, true>), dim3(a.nchunks), dim3(MT_ * 4), 0, s, a, qa);
        return;
    }
    if (fwd) hipLaunchKernelGGL((k_fwd_big<MT_>), dim3(a.nchunks), dim3(MT_ * 4), 0, s, a, qa);
    else hipLaunchKernelGGL((k_bwd_big<MT_>), dim3(a.nchunks), dim3(MT_ * 4), 0, s, a, qa);
}
template <int MT_>
static void launch_chain_lock_t(bool fwd, const ChainArgs &a, hipStream_t s) {
    const dim3 grid((unsigned)((a.nchunks + LOCK_NC - 1) / LOCK_NC)), block(MT_ * 4);
    if (fwd) {
        if (a.pass > 0) hipLaunchKernelGGL((k_fwd_lock<MT_, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_fwd_lock<MT_, false>), grid, block, 0, s, a);
    } else {
        if (a.pass > 0) hipLaunchKernelGGL((k_bwd_lock<MT_, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_bwd_lock<MT_, false>), grid, block, 0, s, a);
    }
}
static bool launch_chain_lock(bool fwd, int Mp, const ChainArgs &a, hipStream_t s) {
    switch (Mp) {
        case 16: launch_chain_lock_t<16>(fwd, a, s); return true;
        case 32: launch_chain_lock_t<32>(fwd, a, s); return true;
        case 48: launch_chain_lock_t<48>(fwd, a, s); return true;
        case 64: launch_chain_lock_t<64>(fwd, a, s); return true;
        default: return false;
    }
}
static bool launch_chain_big(bool fwd, int Mp, const ChainArgs &a, const BigArgs &qa, hipStream_t s) {
    switch (Mp) {
#define B_(x) case x: launch_chain_big_t<x>(fwd, a, qa, s); return true;
        B_(80) B_(96) B_(112) B_(128) B_(144) B_(160) B_(176) B_(192) B_(208) B_(224) B_(240) B_(256)
#undef B_
        default: return false;
    }
}

template <int NT_>
static void launch_uw_t(const UWArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_eig_uw<NT_>, dim3(a.nslabs), dim3(64), 0, s, a);
}
static void launch_uw(int nt, const UWArgs &a, hipStream_t s) {
    switch (nt) {
#define C_(x) case x: launch_uw_t<x>(a, s); break;
        C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
#undef C_
        default: throw std::runtime_error("unsupported number of hidden states");
    }
}
template <int NPL_>
static void launch_s1_t(const S1Args &a, hipStream_t s) {
    hipLaunchKernelGGL(k_s1_scalars<NPL_>, dim3(a.nslabs), dim3(256), 0, s, a);
}
static void launch_s1(int npl, const S1Args &a, hipStream_t s) {
    switch (npl) {
        case 1: launch_s1_t<1>(a, s); break;
        case 2: launch_s1_t<2>(a, s); break;
        case 3: launch_s1_t<3>(a, s); break;
        case 4: launch_s1_t<4>(a, s); break;
        case 8: launch_s1_t<8>(a, s); break;
        case 16: launch_s1_t<16>(a, s); break;
        default: throw std::runtime_error("unsupported number of hidden states");
    }
}

ChainArgs smcpp_im::chain_args() {
    ChainArgs a;
    a.M = M; a.Mp = Mp; a.nchunks = (int)chunks.size(); a.pass = 0;
    a.hot = hot_eig; a.hot2 = hot_eig2; a.variant = 0;
    a.chunks = d_chunks.p; a.rowdesc = d_rowdesc.p + ROWDESC_PAD; a.E = d_E.p; a.dpow = d_dpow.p;
    a.pi_f = d_pi_f.p; a.Tf = d_Tf.p; a.PinvT = d_PinvT.p; a.PT = d_PT.p;
    a.TdT = d_TdT.p; a.Prm = d_Prm.p; a.Pinvrm = d_Pinvrm.p;
    a.alpha = d_alpha.p; a.beta = d_beta.p; a.cnorm = d_cnorm.p;
    a.ends_f = d_ends_f.p; a.used_f = d_used_f.p; a.ends_b = d_ends_b.p; a.used_b = d_used_b.p;
    a.eps_f = eps_f; a.eps_b = eps_b;
    a.dbg = nullptr;
    a.warm_f = nullptr; a.warm_b = nullptr;
    a.Bf = d_Bf.p; a.Bb = d_Bb.p; a.g_span = d_g_span.p; a.nbits = pw_nbits; a.npow = pw_npow;
    a.prio = std::max(0, std::min(3, opt().i(smcpp_opt::O_BWD_PRIO, 1)));
    a.changed = nullptr;
    return a;
}

// LDS budget of the cooperative kernels: exchange buffers + descriptors (+ the emission / eigenvalue-power tables when
// they fit: TAB)
static void coop_lds(int Mp, int K, int G, int &tab_c, size_t &shm_c) {
    const int KQ = Mp / 4, UP = KQ + 2;
    const size_t base_c = (size_t)(4 * UP + 8 * UP) * 8 + 2 * Mp * 4 + 1024 + 64;
    const size_t tabs = ((size_t)K * 4 * UP + (size_t)G * Mp) * 8;      // backward layout of generation 1 is the larger one
    tab_c = (base_c + tabs <= 64 * 1024) ? 1 : 0;
    if (opt().has(smcpp_opt::O_COOP_TAB)) tab_c = tab_c && opt().i(smcpp_opt::O_COOP_TAB, 1) != 0;    // test hook: force the global-table path
    shm_c = base_c + (tab_c ? tabs : 0);
}

// Eigen-free pre-pass: upload pi / T / emission table, build the group powers on the device and launch pass 0 of both
// chains on them; the host then solves the eigenproblems while the GPU runs (estep()).
void smcpp_im::stage_static_and_prepass() {
    ensure_T();
    prepass_launched = false;
    static_packed = false;
    if (!power_ok || (warm_start && warm_valid)) return;
    hipStream_t s = stream, sb = dual_stream ? stream2 : stream;
    const size_t MM = (size_t)Mp * Mp;
    auto ensure = [](auto &v, size_t n, auto init) { if (v.size() != n) v.assign(n, init); };
    ensure(hs_pi_f, (size_t)Mp, 0.f); ensure(hs_Tf, MM, 0.f);
    ensure(hs_TdT, MM, 0.0); ensure(hs_Td, MM, 0.0); ensure(hs_Ep, (size_t)K * Mp, 0.0);
    for (int i = 0; i < M; ++i) {
        hs_pi_f[i] = (float)pi[i];
        for (int j = 0; j < M; ++j) {
            hs_Tf[(size_t)i * Mp + j] = (float)T[(size_t)i * M + j];
            hs_Td[(size_t)i * Mp + j] = T[(size_t)i * M + j];
            hs_TdT[(size_t)j * Mp + i] = T[(size_t)i * M + j];
        }
    }
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < M; ++i) hs_Ep[(size_t)k * Mp + i] = E[(size_t)k * M + i];
    static_packed = true;
    // own small arena (the main one is filled and copied after the eigensolve)
    const size_t need = 8 * 256 + (hs_pi_f.size() + hs_Tf.size()) * 4 + (hs_TdT.size() + hs_Td.size() + hs_Ep.size()) * 8;
    pre_stage.reset(need);
    if (need > pre_cap) {
        if (d_pre) (void)hipFree(d_pre);
        pre_cap = need + need / 4;
        HIPCHK(hipMalloc((void **)&d_pre, pre_cap));
        smcpp_opt::poison(d_pre, pre_cap, __LINE__, __FILE__);
    }
    size_t off = 0;
    auto put = [&](const void *src, size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        std::memcpy(pre_stage.base + off, src, bytes);
        char *dp = d_pre + off;
        off += bytes;
        return dp;
    };
    ChainArgs a = chain_args();
    if (chain_mode == 3) {
        // streamed-operand chains: pi, T (row-major) and the emission table go up, everything else is built on the device
        a.pi_f = reinterpret_cast<const float *>(put(hs_pi_f.data(), hs_pi_f.size() * 4));
        const double *pre_Td = reinterpret_cast<const double *>(put(hs_Td.data(), hs_Td.size() * 8));
        a.E = reinterpret_cast<const double *>(put(hs_Ep.data(), hs_Ep.size() * 8));
        HIPCHK(hipMemcpyAsync(d_pre, pre_stage.base, off, hipMemcpyHostToDevice, s));
        d_changed_f.zero(s);
        d_changed_b.zero(s);
        const int nb = ceil_div((long long)MM, 256);
        hipLaunchKernelGGL(k_big_tq, dim3(nb), dim3(256), 0, s, Mp, pre_Td, d_pre_qTf.p, d_pre_qTdT.p);
        hipLaunchKernelGGL(k_pow_init, dim3(nb, Ke), dim3(256), 0, s, M, Mp, pw_nbits, (const int *)d_e_kid.p, a.E, pre_Td, d_W.p);
        for (int b = 0; b + 1 < pw_nbits; ++b) {
            hipLaunchKernelGGL(k_sq_f64, dim3(Mp / 16, Mp / 16, Ke), dim3(64), 0, s, Mp, (const double *)(d_W.p + (size_t)b * MM),
                               d_W.p + (size_t)(b + 1) * MM, (size_t)pw_nbits * MM);
            if (b + 1 >= 5)
                hipLaunchKernelGGL(k_pow_rescale, dim3(Ke), dim3(256), 0, s, Mp, d_W.p + (size_t)(b + 1) * MM, (size_t)pw_nbits * MM);
        }
        hipLaunchKernelGGL(k_pow_layout, dim3(nb, Ke * pw_nbits), dim3(256), 0, s, Mp, (const double *)d_W.p, d_qBf.p, d_qBb.p);
        pre_bargs = BigArgs();
        pre_bargs.qTf = d_pre_qTf.p; pre_bargs.qTdT = d_pre_qTdT.p;
        pre_bargs.qPinvT = pre_bargs.qPT = pre_bargs.qPrm = pre_bargs.qPinvrm = nullptr;
        pre_bargs.qBf = d_qBf.p; pre_bargs.qBb = d_qBb.p;
        a.variant = 1; a.pass = 0;
        if (sb != s) {
            HIPCHK(hipEventRecord(ev[6], s));
            HIPCHK(hipStreamWaitEvent(sb, ev[6], 0));
        }
        HIPCHK(hipEventRecord(ev[10], s));
        a.changed = d_changed_f.p;
        launch_chain_big(true, Mp, a, pre_bargs, s);
        HIPCHK(hipEventRecord(ev[11], s));
        HIPCHK(hipEventRecord(ev[12], sb));
        a.changed = d_changed_b.p;
        launch_chain_big(false, Mp, a, pre_bargs, sb);
        HIPCHK(hipEventRecord(ev[13], sb));
        HIPCHK(hipGetLastError());
        prepass_launched = true;
        return;
    }
    a.pi_f = reinterpret_cast<const float *>(put(hs_pi_f.data(), hs_pi_f.size() * 4));
    a.Tf = reinterpret_cast<const float *>(put(hs_Tf.data(), hs_Tf.size() * 4));
    a.TdT = reinterpret_cast<const double *>(put(hs_TdT.data(), hs_TdT.size() * 8));
    const double *pre_Td = reinterpret_cast<const double *>(put(hs_Td.data(), hs_Td.size() * 8));
    a.E = reinterpret_cast<const double *>(put(hs_Ep.data(), hs_Ep.size() * 8));
    HIPCHK(hipMemcpyAsync(d_pre, pre_stage.base, off, hipMemcpyHostToDevice, s));
    d_changed_f.zero(s);
    d_changed_b.zero(s);
    {
        const size_t shm = (size_t)(2 * Mp * (Mp + 1) + 8) * sizeof(double);
        switch (Mp) {
#define P_(x) case x: { static bool once = false; if (!once) { HIPCHK(hipFuncSetAttribute((const void *)k_binary_powers<x>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; } \
                    hipLaunchKernelGGL(k_binary_powers<x>, dim3(Ke), dim3(256), shm, s, M, pw_npow, (const int *)d_e_kid.p, a.E, pre_Td, d_Bf.p, d_Bb.p); } break;
            P_(16) P_(32) P_(48) P_(64)
#undef P_
            default: throw std::runtime_error("internal: power pre-pass with an unsupported state count");
        }
    }
    const int dbg_level = opt().i(smcpp_opt::O_POWER_DEBUG, 0);
    if (dbg_level == 1) { HIPCHK(hipStreamSynchronize(s)); fprintf(stderr, "[power] powers ok\n"); static_packed = false; return; }
    int tab_c; size_t shm_c;
    coop_lds(Mp, K, G, tab_c, shm_c);
    CoopArgs cargs;
    cargs.K = K; cargs.G = G;
    cargs.power_off = (int)shm_c;          // two scratch vectors behind the regular carve-up
    shm_c += 2048;
    a.variant = 1; a.pass = 0;
    if (!(opt().i(smcpp_opt::O_BWD_PRIO_MASK, 7) & 1)) a.prio = 0;
    if (sb != s) {
        HIPCHK(hipEventRecord(ev[6], s));
        HIPCHK(hipStreamWaitEvent(sb, ev[6], 0));
    }
    HIPCHK(hipEventRecord(ev[10], s));
    a.changed = d_changed_f.p;
    launch_chain_coop(true, Mp, a, cargs, tab_c, shm_c, s);
    HIPCHK(hipEventRecord(ev[11], s));
    HIPCHK(hipEventRecord(ev[12], sb));
    a.changed = d_changed_b.p;
    launch_chain_coop(false, Mp, a, cargs, tab_c, shm_c, sb);
    HIPCHK(hipEventRecord(ev[13], sb));
    HIPCHK(hipGetLastError());
    if (dbg_level == 2) { HIPCHK(hipStreamSynchronize(s)); HIPCHK(hipStreamSynchronize(sb)); fprintf(stderr, "[power] pre-pass ok\n"); }
    prepass_launched = true;
}

void smcpp_im::run_chains() {
    hipStream_t s = stream;
    ChainArgs a = chain_args();
    CoopArgs cargs;
    cargs.K = K; cargs.G = G; cargs.power_off = 0;
    BigArgs bargs;
    bargs.qTf = d_qTf.p; bargs.qPinvT = d_qPinvT.p; bargs.qPT = d_qPT.p; bargs.qTdT = d_qTdT.p;
    bargs.qPrm = d_qPrm.p; bargs.qPinvrm = d_qPinvrm.p; bargs.qBf = nullptr; bargs.qBb = nullptr;
    size_t shm_c = 0;
    int tab_c = 0;
    coop_lds(Mp, K, G, tab_c, shm_c);
    const bool warm = warm_start && warm_valid && chain_mode == 2 && Mp <= 64 &&
                      d_warm_f.n == chunks.size() * (size_t)Mp && d_warm_b.n == chunks.size() * (size_t)Mp;
    a.warm_f = warm ? d_warm_f.p : nullptr;
    a.warm_b = warm ? d_warm_b.p : nullptr;
    if (opt().has(smcpp_opt::O_DEBUG_CYCLES)) { d_dbg.alloc(16); d_dbg.zero(s); a.dbg = d_dbg.p; }
    const bool pre = prepass_launched;      // pass 0 of both chains already runs (eigen-free pre-pass, flags zeroed there)
    if (!pre) {
        d_changed_f.zero(s);
        d_changed_b.zero(s);
    }
    if (h_flags_cap < 2 * (max_pass + 1)) {
        if (h_flags) (void)hipHostFree(h_flags);
        h_flags_cap = 2 * (max_pass + 1);
        HIPCHK(hipHostMalloc((void **)&h_flags, sizeof(int) * h_flags_cap, hipHostMallocCoherent | hipHostMallocMapped));
        d_flags_view = nullptr;
    }
    int *chf = h_flags, *chb = h_flags + (max_pass + 1);
    auto first_quiet = [](const int *ch, int upto) {
        for (int j = 0; j < upto; ++j)
            if (ch[j] == 0) return j;
        return -1;
    };
    int launched_f = pre ? 1 : 0, launched_b = pre ? 1 : 0;
    int want_f = std::min(max_pass, last_fwd_passes > 0 ? last_fwd_passes + 1 : std::min(max_pass, 8));
    int want_b = std::min(max_pass, last_bwd_passes > 0 ? last_bwd_passes + 1 : std::min(max_pass, 8));
    const size_t nel_ends = chunks.size() * (size_t)Mp;
    // after a pre-pass, pass 1 is a FULL pass from the pre-pass's end vectors (no skip test, no merge exit): every stored
    // row then comes from the exact kernels
    // issue priority of the backward wavefronts per pass (SMCPP_BWD_PRIO_MASK: bit 0 pre-pass, bit 1 the full pass after
    // it, bit 2 every other pass)
    const int prio_mask = opt().i(smcpp_opt::O_BWD_PRIO_MASK, 7);
    const int prio0 = a.prio;
    auto set_variant = [&](int pass) {
        a.pass = pass;
        if (pre && pass == 1) { a.variant = 2; a.warm_f = d_ends_f.p; a.warm_b = d_ends_b.p; (void)nel_ends; a.prio = (prio_mask & 2) ? prio0 : 0; }
        else { a.variant = 0; a.warm_f = warm ? d_warm_f.p : nullptr; a.warm_b = warm ? d_warm_b.p : nullptr; a.prio = (prio_mask & 4) ? prio0 : 0; }
    };
    // The two chains are independent (beta does not depend on alpha).  The cooperative kernels leave most of a CU's
    // LDS and issue slots idle, so the backward passes run on a second stream and share the CUs with the forward ones.
    const bool dual = dual_stream && ((chain_mode == 2 && Mp <= 64) || chain_mode == 4 || Mp > 64);
    hipStream_t sb = dual ? stream2 : s;
    if (dual) {
        HIPCHK(hipEventRecord(ev[6], s));              // parameters / zeroed flags are ready on the main stream
        HIPCHK(hipStreamWaitEvent(sb, ev[6], 0));
    }
    HIPCHK(hipEventRecord(ev[1], s));
    bool fdone = false, bdone = false;
    int fq = -1, bq = -1;
    bool first_round = true;
    while (true) {
        if (!fdone) {
            a.changed = d_changed_f.p;
            for (; launched_f < want_f; ++launched_f) {
                set_variant(launched_f);
                if (!(chain_mode == 4 && launch_chain_lock(true, Mp, a, s)) &&
                    !(chain_mode == 3 && launch_chain_big(true, Mp, a, bargs, s)) &&
                    !(chain_mode == 2 && launch_chain_coop(true, Mp, a, cargs, tab_c, shm_c, s)))
                    throw std::runtime_error("internal: no dense chain kernel for this number of hidden states");
            }
        }
        if (first_round) HIPCHK(hipEventRecord(ev[2], dual ? sb : s));
        if (!bdone) {
            a.changed = d_changed_b.p;
            for (; launched_b < want_b; ++launched_b) {
                set_variant(launched_b);
                if (!(chain_mode == 4 && launch_chain_lock(false, Mp, a, sb)) &&
                    !(chain_mode == 3 && launch_chain_big(false, Mp, a, bargs, sb)) &&
                    !(chain_mode == 2 && launch_chain_coop(false, Mp, a, cargs, tab_c, shm_c, sb)))
                    throw std::runtime_error("internal: no dense chain kernel for this number of hidden states");
            }
        }
        HIPCHK(hipGetLastError());
        if (first_round && dual) HIPCHK(hipEventRecord(ev[7], s));      // end of the first batch of forward passes
        HIPCHK(hipMemcpyAsync(chf, d_changed_f.p, sizeof(int) * (max_pass + 1), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(chb, d_changed_b.p, sizeof(int) * (max_pass + 1), hipMemcpyDeviceToHost, sb));
        if (dual) {
            HIPCHK(hipEventRecord(ev[6], sb));
            HIPCHK(hipStreamWaitEvent(s, ev[6], 0));   // the statistics (main stream) need both chains
        }
        HIPCHK(hipEventRecord(ev[3], s));
        // Optimistic: the first batch normally contains the quiet pass (it is sized from the previous E-step), so the
        // statistics are queued behind it BEFORE the host waits for the flags - the read-back round trip and their launch
        // latency disappear behind GPU work.  If the flags say otherwise the statistics are simply queued again later.
        if (first_round && !save_gamma) enqueue_stats();
        else stats_enqueued = false;
        HIPCHK(hipStreamSynchronize(s));
        if (dual) HIPCHK(hipStreamSynchronize(sb));
        first_round = false;
        fq = first_quiet(chf, launched_f);
        bq = first_quiet(chb, launched_b);
        fdone = fq >= 0 || launched_f >= max_pass;
        bdone = bq >= 0 || launched_b >= max_pass;
        if (fdone && bdone) break;
        stats_enqueued = false;             // more passes follow: whatever was queued is stale
        if (!fdone) want_f = std::min(max_pass, launched_f + 4);
        if (!bdone) want_b = std::min(max_pass, launched_b + 4);
    }
    chains_dual = dual;
    if (a.dbg) {
        long long h[16];
        HIPCHK(hipMemcpy(h, d_dbg.p, sizeof(h), hipMemcpyDeviceToHost));
        for (int w = 0; w < 4; ++w)
            fprintf(stderr, "[cycles] fwd wg1 wave%d: loop %lld, end-barrier %lld, mid-barrier %lld, rows %lld\n", w, h[4 * w], h[4 * w + 1], h[4 * w + 2], h[4 * w + 3]);
    }
    if (fq < 0 || bq < 0) { stats_enqueued = false; throw std::runtime_error("chunk-boundary iteration did not converge"); }
    last_fwd_passes = fq;
    last_bwd_passes = bq;
    if (warm_start && chain_mode == 2 && Mp <= 64) {
        // every pass after the first quiet one only copies the boundary vectors forward: the buffer of the last
        // launched pass holds the converged ones
        const size_t nel = chunks.size() * (size_t)Mp;
        d_warm_f.alloc(nel); d_warm_b.alloc(nel);
        HIPCHK(hipMemcpyAsync(d_warm_f.p, d_ends_f.p + (size_t)((launched_f - 1) & 1) * nel, nel * sizeof(float),
                              hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(d_warm_b.p, d_ends_b.p + (size_t)((launched_b - 1) & 1) * nel, nel * sizeof(double),
                              hipMemcpyDeviceToDevice, s));
        warm_valid = true;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// chains on the semiseparable structure of T (chains_ss.hpp)
// ---------------------------------------------------------------------------------------------------------------
// Generators of T = diag(d) + [below: g_j] + [above: c0 + Phi'(i,j)], Phi'(i,i+1) = b_i, Phi'(i,j+1) = a_j Phi'(i,j)
// (transition.cpp:176-254; c0 = 1e-5 / (M + 1) is the mixing constant of lines 249-254).  The dense T is what the reference's
// getters hand out and what the statistics use, so the generators are taken FROM it and the reconstruction is checked entry by
// entry: a T without this structure (smcpp_set_raw with an arbitrary matrix) sends the E-step to the dense kernels.
static bool ss_generators(int M, int MS, const double *Tm, std::vector<double> &gen, double &c0_out) {
    const double c0 = 1e-5 / (double)(M + 1);
    const double tol = 1e-11;
    std::vector<double> d(M), g(M, 0.0), a(M, 0.0), b(M, 0.0);
    for (int j = 0; j < M; ++j) d[j] = Tm[(size_t)j * M + j];
    for (int j = 0; j + 1 < M; ++j) {
        g[j] = Tm[(size_t)(M - 1) * M + j];
        b[j] = Tm[(size_t)j * M + j + 1] - c0;
    }
    // (both sweeps below walk T row by row: at M = 256 the matrix is 512 KB and a column walk misses the cache on every entry)
    for (int i = 1; i < M; ++i) {
        const double *row = Tm + (size_t)i * M;
        for (int j = 0; j < i && j + 1 < M; ++j)
            if (!(std::fabs(row[j] - g[j]) <= tol * std::fabs(g[j]))) return false;
    }
    {
        // a_j from the row with the LARGEST entry in column j above the diagonal (best conditioned quotient)
        std::vector<int> ib(M, 0);
        std::vector<double> best(M, -1.0);
        for (int i = 0; i + 2 < M; ++i) {
            const double *row = Tm + (size_t)i * M;
            for (int j = std::max(1, i + 1); j + 1 < M; ++j)
                if (row[j] > best[j]) { best[j] = row[j]; ib[j] = i; }
        }
        for (int j = 1; j + 1 < M; ++j) {
            const double den = Tm[(size_t)ib[j] * M + j] - c0;
            a[j] = den > 0.0 ? (Tm[(size_t)ib[j] * M + j + 1] - c0) / den : 0.0;
        }
    }
    for (int i = 0; i + 1 < M; ++i) {
        double v = b[i];
        for (int j = i + 1; j < M; ++j) {
            const double t = Tm[(size_t)i * M + j];
            if (!(std::fabs(c0 + v - t) <= tol * std::fabs(t)) || !(t > 0.0)) return false;
            v *= a[j];
        }
    }
    for (int j = 0; j < M; ++j)
        if (!(d[j] > 0.0) || !std::isfinite(a[j]) || !std::isfinite(b[j])) return false;
    c0_out = c0;
    gen.assign((size_t)10 * MS, 0.0);
    double *f_dc = &gen[0], *f_g = f_dc + MS, *f_cg = f_g + MS, *f_b = f_cg + MS, *f_a = f_b + MS, *f_d = f_a + MS,
           *b_dc = f_d + MS, *b_g = b_dc + MS, *b_b = b_g + MS, *b_a = b_b + MS;
    for (int j = 0; j < M; ++j) {
        f_dc[j] = d[j] - c0; f_g[j] = g[j]; f_cg[j] = c0 - g[j]; f_b[j] = b[j]; f_a[j] = a[j]; f_d[j] = d[j];
        const int p = MS - 1 - j;
        b_dc[p] = d[j] - c0; b_g[p] = g[j]; b_b[p] = b[j]; b_a[p] = a[j];
    }
    return true;
}

// The same generators straight from the O(M) quantities the device-prepared model path computed (prep.hpp: TransitionGenJac - below
// the diagonal T(i, c) = ed[c], above it pf[i] W[c], the diagonal closes the row; transition_expand then floors at 1e-20 and mixes with
// c0 = 1e-5 / (M + 1)): no M x M matrix is formed or checked.  The diagonal's row sum is taken from a prefix sum of ed and a suffix sum
// of W instead of entry by entry: 1e-16 of what the expanded matrix holds.
static bool ss_generators_from_tgen(int M, int MS, const smcpp_host::TransitionGenJac &tj, std::vector<double> &gen, double &c0_out) {
    if (!tj.ok || tj.M != M || (int)tj.ed.size() != std::max(0, M - 1) || (int)tj.pf.size() != M || (int)tj.W.size() != M) return false;
    const double beta = 1e-5, c0 = beta / (double)(M + 1), om = 1.0 - beta;
    auto mix = [&](double raw) { return std::max(raw, 1e-20) * om; };          // (without the + c0)
    std::vector<double> d(M), g(M, 0.0), a(M, 0.0), b(M, 0.0), below(M, 0.0), above(M + 1, 0.0);
    for (int j = 1; j < M; ++j) below[j] = below[j - 1] + tj.ed[j - 1];
    for (int c = M - 1; c >= 1; --c) above[c] = above[c + 1] + tj.W[c];            // above[c] = sum_{k >= c} W[k]
    for (int j = 0; j + 1 < M; ++j) {
        g[j] = mix(tj.ed[j]) + c0;
        b[j] = mix(tj.pf[j] * tj.W[j + 1]);
    }
    for (int j = 1; j + 1 < M; ++j) a[j] = tj.W[j] > 0.0 ? tj.W[j + 1] / tj.W[j] : 0.0;
    for (int j = 0; j < M; ++j) d[j] = mix(1.0 - (below[j] + tj.pf[j] * above[j + 1])) + c0;
    for (int j = 0; j < M; ++j)
        if (!(d[j] > 0.0) || !std::isfinite(a[j]) || !std::isfinite(b[j]) || !(b[j] >= 0.0) || (j + 1 < M && !(g[j] > 0.0))) return false;
    c0_out = c0;
    gen.assign((size_t)10 * MS, 0.0);
    double *f_dc = &gen[0], *f_g = f_dc + MS, *f_cg = f_g + MS, *f_b = f_cg + MS, *f_a = f_b + MS, *f_d = f_a + MS,
           *b_dc = f_d + MS, *b_g = b_dc + MS, *b_b = b_g + MS, *b_a = b_b + MS;
    for (int j = 0; j < M; ++j) {
        f_dc[j] = d[j] - c0; f_g[j] = g[j]; f_cg[j] = c0 - g[j]; f_b[j] = b[j]; f_a[j] = a[j]; f_d[j] = d[j];
        const int p = MS - 1 - j;
        b_dc[p] = d[j] - c0; b_g[p] = g[j]; b_b[p] = b[j]; b_a[p] = a[j];
    }
    return true;
}

bool smcpp_im::ss_generators_only() {
    if (T_lazy && tgen_valid && ss_generators_from_tgen(M, 64 * NPL, tgen, ss_gen, ss_c0)) {
        // (no expanded matrix yet, and none needed by the chains)
    } else {
        ensure_T();
        if (!ss_generators(M, 64 * NPL, T.data(), ss_gen, ss_c0)) return false;
    }
    return true;
}

bool smcpp_im::ss_extract_generators() {
    if (!ss_generators_only()) return false;
    // a row of span s applies its operator s times without rescaling: keep clear of underflow
    // (a device-prepared table is checked by the kernel that forms it: DevPrep flag 2, looked at when the E-step has drained)
    if (!E_on_dev) for (const Group &gr : groups) {
        if (ss_hybrid && gr.span > ss_hyb_th) continue;          // an eigen-power step, not `span` scan steps
        double mn = 1.0;
        for (int i = 0; i < M; ++i) mn = std::min(mn, E[(size_t)gr.kid * M + i]);
        if (!(mn > 0.0) || (double)gr.span * std::log(mn) < -450.0) return false;
    }
    return true;
}

template <int NPL_, bool HYB_, bool ALL_, bool H32_ = false>
static void launch_chain_ss_tt(const SsArgs &a, int ntasks, size_t shm, hipStream_t s, int wgw) {
    static bool once = false;
    if (!once) {
        HIPCHK(hipFuncSetAttribute((const void *)k_chain_ss<NPL_, HYB_, ALL_, H32_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once = true;
    }
    hipLaunchKernelGGL((k_chain_ss<NPL_, HYB_, ALL_, H32_>), dim3(ntasks / wgw), dim3(64 * wgw), shm, s, a);
}
template <int NPL_, bool HYB_>
static void launch_chain_ss_t(const SsArgs &a, int ntasks, size_t shm, hipStream_t s, int wgw) {
    // every key slot in LDS (the usual case): the instantiation without the global path of the emission vectors
    // (M <= 32 with one state per lane: the instantiation whose scans skip the level that would only move zeros)
    const bool h32_off = opt().off(smcpp_opt::O_SS_H32);
    if (NPL_ == 1 && !HYB_ && a.K <= a.nlds && a.Mp <= 32 && !h32_off) launch_chain_ss_tt<NPL_, HYB_, true, NPL_ == 1 && !HYB_>(a, ntasks, shm, s, wgw);
    else if (a.K <= a.nlds) launch_chain_ss_tt<NPL_, HYB_, true>(a, ntasks, shm, s, wgw);
    else launch_chain_ss_tt<NPL_, HYB_, false>(a, ntasks, shm, s, wgw);
}
static void launch_chain_ss(int npl, const SsArgs &a, int ntasks, size_t shm, hipStream_t s, int wgw = 4) {
    switch (npl) {
        case 1: if (a.hyb_th != 0x7fffffff) launch_chain_ss_t<1, true>(a, ntasks, shm, s, wgw); else launch_chain_ss_t<1, false>(a, ntasks, shm, s, 4); break;
        case 2: launch_chain_ss_t<2, false>(a, ntasks, shm, s, 4); break;
        case 3: launch_chain_ss_t<3, false>(a, ntasks, shm, s, 4); break;
        case 4: launch_chain_ss_t<4, false>(a, ntasks, shm, s, 4); break;
        case 8: launch_chain_ss_t<8, false>(a, ntasks, shm, s, 4); break;
        case 16: launch_chain_ss_t<16, false>(a, ntasks, shm, s, 4); break;
        default: throw std::runtime_error("unsupported number of hidden states");
    }
}


void smcpp_im::ss_launch_passes(int upto) {
    const size_t shm = (size_t)ss_nlds * 64 * NPL * sizeof(double) + ss_tab_bytes();
    for (; ss_launched < upto; ++ss_launched) {
        // per direction: `light` store-free float passes (history), then one full pass from their end vectors, then re-run
        // passes; without light passes the first pass is the full one (from pi / the uniform vector)
        const int p = ss_launched;
        const bool lf = p < ss_light_f, lb = p < ss_light_b;
        ss_args.pass = p;
        ss_args.mode_f = lf ? 2 : p == 0 ? 0 : 1;
        ss_args.mode_b = lb ? 2 : p == 0 ? 0 : 1;
        ss_args.full_f = (p > 0 && p == ss_light_f) ? 1 : 0;
        ss_args.full_b = (p > 0 && p == ss_light_b) ? 1 : 0;
        launch_chain_ss(NPL, ss_args, (int)ss_tasks.size(), shm, stream, ss_wg_waves);
    }
    HIPCHK(hipGetLastError());
}

// Upload pi, the generators and the emission vectors (by key slot) and start the passes: nothing here needs an eigensystem,
// so the host solves the eigenproblems of the statistics while the chains run.
void smcpp_im::ss_launch_initial() {
    hipStream_t s = stream;
    const int MS = 64 * NPL;
    std::vector<float> &pi_f = hs_pi_f;
    if (pi_f.size() != (size_t)Mp) pi_f.assign(Mp, 0.f);
    for (int i = 0; i < M; ++i) pi_f[i] = (float)pi[i];
    const size_t need = 12 * 256 + pi_f.size() * 4 + ss_gen.size() * 8 + (size_t)K * MS * 8;
    pre_stage.reset(need);
    if (need > pre_cap) {
        if (d_pre) (void)hipFree(d_pre);
        pre_cap = need + need / 4;
        HIPCHK(hipMalloc((void **)&d_pre, pre_cap));
        smcpp_opt::poison(d_pre, pre_cap, __LINE__, __FILE__);
    }
    size_t off = 0;
    auto put = [&](const void *src, size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        if (src) std::memcpy(pre_stage.base + off, src, bytes);
        char *dp = d_pre + off;
        off += bytes;
        return dp;
    };
    SsArgs &a = ss_args;
    a = SsArgs();
    a.M = M; a.Mp = Mp; a.nchunks = (int)chunks.size(); a.pass = 0; a.K = K; a.nlds = ss_nlds;
    a.chunks = d_chunks.p; a.rowdesc = d_rowdesc_ss.p + ROWDESC_PAD;
    a.chunks_b = d_chunks_b.p; a.nchunks_b = (int)chunks_b.size(); a.tasks = d_tasks.p;
    a.pi_f = reinterpret_cast<const float *>(put(pi_f.data(), pi_f.size() * 4));
    const double *gd = reinterpret_cast<const double *>(put(ss_gen.data(), ss_gen.size() * 8));
    a.f_dc = gd; a.f_g = gd + MS; a.f_cg = gd + 2 * MS; a.f_b = gd + 3 * MS; a.f_a = gd + 4 * MS; a.f_d = gd + 5 * MS;
    a.b_dc = gd + 6 * MS; a.b_g = gd + 7 * MS; a.b_b = gd + 8 * MS; a.b_a = gd + 9 * MS;
    a.c0 = ss_c0;
    if (E_on_dev) a.E = dprep->d_Es.p;       // written by the device preparation, by key slot
    else {
        const size_t eoff = (off + 255) & ~(size_t)255;
        double *he = reinterpret_cast<double *>(pre_stage.base + eoff);
        std::memset(he, 0, (size_t)K * MS * 8);
        for (int k = 0; k < K; ++k)
            std::memcpy(he + (size_t)ss_slot_of_key[k] * MS, &E[(size_t)k * M], sizeof(double) * M);
        a.E = reinterpret_cast<const double *>(put(nullptr, (size_t)K * MS * 8));
    }
    a.alpha = d_alpha.p; a.beta = d_beta.p; a.cnorm = d_cnorm.p;
    a.ends_f = d_ends_f.p; a.used_f = d_used_f.p; a.ends_b = d_ends_b.p; a.used_b = d_used_b.p;
    {
        // the per-pass flags live in pinned host memory: written by the kernels through its device view, cleared and read by the host
        // ([0, max_pass]: "a chunk of the forward chain re-ran", then the same of the backward chain, then the two "an end vector
        // was rewritten" arrays of the scan chains)
        if (h_flags_cap < 4 * (max_pass + 1)) {
            if (h_flags) (void)hipHostFree(h_flags);
            h_flags_cap = 4 * (max_pass + 1);
            HIPCHK(hipHostMalloc((void **)&h_flags, sizeof(int) * h_flags_cap, hipHostMallocCoherent | hipHostMallocMapped));
            d_flags_view = nullptr;
        }
        if (!d_flags_view) HIPCHK(hipHostGetDevicePointer((void **)&d_flags_view, h_flags, 0));
        if (!h_done) {
            HIPCHK(hipHostMalloc((void **)&h_done, 64, hipHostMallocCoherent | hipHostMallocMapped));
            *h_done = 0;
            HIPCHK(hipHostGetDevicePointer((void **)&d_done_view, h_done, 0));
        }
        std::memset(h_flags, 0, sizeof(int) * h_flags_cap);
    }
    a.changed_f = d_flags_view; a.changed_b = d_flags_view + (max_pass + 1);
    a.endchg_f = d_flags_view + 2 * (max_pass + 1); a.endchg_b = d_flags_view + 3 * (max_pass + 1);
    a.eps_f = eps_f; a.eps_b = eps_b; a.full_f = a.full_b = 0;
    {
        // All scans of the stored passes in float (chains_ss.hpp: ss_x_scan_fwd / ss_x_scan_bwd), the default since round 5 for one
        // state per lane; SMCPP_SS_MIXED=0 keeps the fp64 scans (tests and bench.py's `value_ref_width` compare the two).
        // Round 6: with save_gamma as well - the decoded index of EVERY column of the two full-size binned contigs (471 106 columns,
        // goldens G19) equals the compiled reference's under the float scans (tests/test_gpu_argmax.py), so the path that is timed and
        // the path whose indices are checked are the same arithmetic.
        const bool mixed_on = !opt().off(smcpp_opt::O_SS_MIXED);
        a.mixed = (mixed_on && NPL == 1 && !ss_hybrid) ? 1 : 0;
    }
    if (ss_hybrid) {
        a.hyb_th = ss_hyb_th; a.Ke = Ke; a.hot_ek = std::max(0, hot_eig); a.dirsplit = ss_dirsplit ? 1 : 0;
        a.nk_lds = ss_nk_lds;
        for (int e = 0; e < 4; ++e) { a.key_of_slot[e] = ss_ekey_of_slot[e]; a.slot_of_key[e] = ss_eslot_of_key[e]; }
        a.Pinvrm = d_Pinvrm.p; a.Prm = d_Prm.p; a.PinvT = d_PinvT.p; a.PT = d_PT.p; a.dsc = d_dsc.p;
    }
    {
        // light passes: enough of them that the full pass starts ~11 e-folds of history in (the chains forget with an e-fold of
        // ~240 positions forward, ~340 backward on the benchmark model); none when the chunks are long against that
        long long pos = 0;
        const int ef = opt().i(smcpp_opt::O_SS_LIGHT_F, -1), eb = opt().i(smcpp_opt::O_SS_LIGHT_B, -1);
        pos = ss_positions / std::max<size_t>(1, chunks.size());
        const long long pos_b = ss_positions / std::max<size_t>(1, chunks_b.size());
        auto pick = [&](double hist, long long p_) { return p_ <= 0 || (double)p_ > 1.5 * hist ? 0 : std::min(4, (int)std::ceil(hist / (double)p_)); };
        ss_light_f = ef >= 0 ? ef : pick(2800.0, pos);
        ss_light_b = eb >= 0 ? eb : pick(3900.0, pos_b);
        if (chunks.size() <= (size_t)n_contigs) ss_light_f = 0;      // one chunk per contig: nothing to iterate
        if (chunks_b.size() <= (size_t)n_contigs) ss_light_b = 0;
    }
    if (ss_hybrid) ss_light_f = ss_light_b = 0;      // (the light passes have no eigen-power step; un-binned inputs have long chunks)
    // halo pass: the first pass enters every chunk through its halo and stores rows that are already exact; no light passes
    const bool use_halo = ss_halo && !(warm_start && ss_warm_valid) && chunks.size() > (size_t)n_contigs && chunks_b.size() > (size_t)n_contigs;
    if (use_halo) ss_light_f = ss_light_b = 0;
    a.halo = use_halo ? 1 : 0;
    ss_pass0 = 0;
    if (warm_start && ss_warm_valid && chunks.size() > (size_t)n_contigs && chunks_b.size() > (size_t)n_contigs) {
        // the boundary vectors of the previous E-step are exact for ITS parameters, i.e. off by the parameter step instead of by
        // O(1): they replace one light pass; every stored row still comes from the full fp64 pass on the new parameters
        ss_pass0 = ss_warm_parity == 0 ? 1 : 2;
        ss_light_f = ss_pass0 + std::max(0, ss_light_f - 1);
        ss_light_b = ss_pass0 + std::max(0, ss_light_b - 1);
    }
    ss_warm_valid = false;                     // (set again when this E-step's chains have converged)
    a.dbg = nullptr;
    if (opt().has(smcpp_opt::O_DEBUG_CYCLES)) { d_dbg.alloc(16); d_dbg.zero(s); a.dbg = d_dbg.p; }
    HIPCHK(hipMemcpyAsync(d_pre, pre_stage.base, off, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(ev[10], s));
    ss_launched = ss_pass0;
    // (round 5) the passes launched up front end with the pass that is expected to rewrite no end vector - the all-skip pass behind
    // it (0.01 ms of kernel + its place in the queue) is not launched: run_chains_ss certifies from the end-vector flags
    const bool cert_launch = opt().on(smcpp_opt::O_SS_CERT_PASS);
    // (never fewer than one re-run pass behind the pass that stores everything - that one always rewrites its end vectors: with one
    // chunk per contig, or boundaries that were already exact, the quiet pass IS that re-run pass)
    const int want = std::min(max_pass, std::max(ss_pass0 + (last_ss_passes > 0 ? last_ss_passes + (cert_launch || ss_need_cert_pass ? 1 : 0) : 6),
                                                 std::max(ss_light_f, ss_light_b) + 2));
    ss_launch_passes(want);
    // (no event behind the passes here: run_chains_ss records ev[3] at this very position, and every record costs the queue ~3 us
    // in front of the statistics' critical branch - tools/sync_lab.hip)
}

bool smcpp_im::wait_done(int epoch) {
    // poll the pinned word the last kernel of the queue writes; a generous deadline, then the ordinary blocking wait
    const auto t0 = std::chrono::steady_clock::now();
    unsigned n = 0;
    while (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) != epoch) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
        if ((++n & 0x3fff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
            HIPCHK(hipStreamSynchronize(stream));
            return __atomic_load_n(h_done, __ATOMIC_ACQUIRE) == epoch;
        }
    }
    return true;
}

void smcpp_im::run_chains_ss() {
    hipStream_t s = stream;
    const int *chf = h_flags, *chb = h_flags + (max_pass + 1);
    const int p0 = ss_pass0;
    auto first_quiet = [p0](const int *cf, const int *cb, int upto) {
        for (int j = p0; j < upto; ++j)
            if (cf[j] == 0 && cb[j] == 0) return j;
        return -1;
    };
    ss_warm_valid = false;
    bool first_round = true;
    int q = -1, first_launched = -1;      // (first_launched: passes launched when the first round failed to certify, -1: it did)
    const bool poll = !opt().off(smcpp_opt::O_POLL);
    while (true) {
        HIPCHK(hipEventRecord(ev[3], s));
        // optimistic, as run_chains(): the statistics are queued right behind the passes; the host only looks at the flags (pinned
        // memory the kernels wrote) when the queue has drained; in the rare round that needs more passes the statistics are redone
        const bool spec_gamma = !opt().off(smcpp_opt::O_SPEC_GAMMA);
        done_folded = false;
        if (first_round && (!save_gamma || spec_gamma)) {              // (save_gamma too: the passes launched up front almost always suffice)
            fold_done_epoch = poll ? done_epoch + 1 : 0;
            enqueue_stats();
            fold_done_epoch = 0;
        } else stats_enqueued = false;
        done_covers_stats = stats_enqueued;
        if (poll) {
            ++done_epoch;
            if (!done_folded) hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s, d_done_view, done_epoch);
            if (!wait_done(done_epoch)) throw std::runtime_error("the device did not signal completion");
        } else HIPCHK(hipStreamSynchronize(s));
        first_round = false;
        q = first_quiet(chf, chb, ss_launched);
        if (q < 0 && ss_launched > p0) {
            // no launched pass was quiet - but if the LAST one rewrote no end vector (every chunk that re-ran merged with its stored
            // trajectory; a pass that stores everything always sets its flag), every chunk's input is what it last ran from and the
            // next pass would skip them all: that pass is the quiet one, without having been launched
            const int L = ss_launched - 1;
            const int *ecf = h_flags + 2 * (max_pass + 1), *ecb = h_flags + 3 * (max_pass + 1);
            if (ecf[L] == 0 && ecb[L] == 0) q = ss_launched;
        }
        if (q >= 0 || ss_launched >= max_pass) break;
        stats_enqueued = false;
        done_covers_stats = false;
        if (first_launched < 0) first_launched = ss_launched;
        ss_launch_passes(std::min(max_pass, ss_launched + 3));
    }
    chains_dual = false;
    if (ss_args.dbg) {
        long long h[8];
        HIPCHK(hipMemcpy(h, d_dbg.p, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "[cycles] ss pass 0, chunk 1: forward %lld shader clocks, %lld x 10 ns, %lld positions, %lld rows; backward %lld "
                "clocks, %lld x 10 ns, %lld positions, %lld rows\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
    if (q < 0) { stats_enqueued = false; throw std::runtime_error("chunk-boundary iteration did not converge"); }
    // the flags of the first round did not certify, yet the very next pass was quiet: on this input the last working pass rewrites
    // end vectors WITHIN the tolerance - from now on the all-skip pass is launched up front again (one round instead of two)
    if (first_launched >= 0 && q == first_launched) ss_need_cert_pass = true;
    last_ss_passes = q - p0;
    last_fwd_passes = last_bwd_passes = q - p0;
    // every launched pass carried the end vectors forward (a skipped chunk copies them): they sit at the last pass's parity
    ss_warm_parity = (ss_launched - 1) & 1;
    ss_warm_valid = true;
}

void smcpp_im::run_stats() {
    if (!stats_enqueued) enqueue_stats();
    finish_stats();
}

void smcpp_im::finish_stats() {
    if (!(ss_active && done_covers_stats)) HIPCHK(hipStreamSynchronize(stream));      // (else: run_chains_ss saw the queue drain)
    done_covers_stats = false;
    std::memcpy(loglik.data(), h_ll, sizeof(double) * n_contigs);
    stats_enqueued = false;
}

void smcpp_im::enqueue_stats() {
    hipStream_t s = stream;
    if (h_ll_cap < n_contigs) {
        if (h_ll) (void)hipHostFree(h_ll);
        h_ll_cap = n_contigs;
        HIPCHK(hipHostMalloc((void **)&h_ll, sizeof(double) * h_ll_cap, hipHostMallocCoherent | hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void **)&d_ll_view, h_ll, 0));
    }
    // The parameter arena of a lean E-step (Td, the emission table of a set_raw manager, the groups' log-scales) is copied on the
    // SECOND stream beside the chains (engine_params.hpp: arena_side, event 20).  The main stream waits for it here; (round 6) so
    // does EVERY side stream at the point where it forks off the chains' end (`arena_wait` below) - until round 6 only the main
    // stream waited, behind the fork event, and a side-stream kernel (k_loglik_partial reads the log-scales, the rank updates of a
    // set_raw manager the emission table) could run before the copy had landed: an order-dependent wrong log-likelihood on recycled
    // device memory, NaN under SMCPP_DEBUG_POISON at M = 768 on 70 rows (tools/poison_probe.py found it).  A wait in front of the
    // fork event itself was measured at 12 us per eval (every branch then starts behind the barrier packet); on the side streams,
    // which have slack, it costs nothing measurable.
    if (arena_side) HIPCHK(hipStreamWaitEvent(s, ev[20], 0));
    auto arena_wait = [&](hipStream_t x) { if (arena_side && x != s) HIPCHK(hipStreamWaitEvent(x, ev[20], 0)); };
    // log-likelihood (also materialises log_c per row)
    LoglikArgs la;
    la.cnorm = d_cnorm.p; la.rowinfo = d_rowinfo.p; la.g_logscale = d_g_logscale.p;
    la.contig_base = d_contig_base.p; la.contig_L = d_contig_L.p; la.partial = d_llpart.p; la.loglik = d_loglik.p;
    la.logc = d_logc.p; la.nblk = llblk;
    la.loglik_host = d_ll_view;          // the final kernel writes the per-contig values into the pinned array as well: no copy
    // save_gamma: the per-row gammas of the span > 1 rows (2 M^3 flop each, the matrix pipe's business for ~1 ms on a million rows)
    // need alpha, beta and the eigensystems only - not a single statistic: they run on their own stream BESIDE the (memory- and
    // latency-bound) statistics instead of behind them
    // (round 6) long rows at 64 < M <= 256 on a transition matrix with the reference's structure: eigen-power pieces + scan steps instead
    // of the scalar eigensystem kernel (they read the statistics' own U / W products, so they stay behind them on the main stream)
    const bool gamma_pieces = save_gamma && n_e_rows > 0 && !eigfree && NT > 4 && Mp <= 256 && !opt().off(smcpp_opt::O_GAMMA_PIECES) &&
                              (double)gamma_piece_count() * Mp * 20.0 < 96e9 && (ss_active || ss_generators_only());
    const bool gamma_side = save_gamma && n_e_rows > 0 && dual_stream && stream_hi != nullptr && !gamma_pieces &&
                            !opt().off(smcpp_opt::O_GAMMA_SIDE);
    if (save_gamma) {
        d_gamma_rows.alloc((size_t)total_rows * Mp);
        // Every row 1 .. L of a contig is written whole by the statistics (span-1 rows: k_s1_scalars; span > 1 rows: the per-row gamma
        // kernels); only row 0 (column 0 of the caller's matrix comes from gamma0) is nobody's: it alone is cleared.  Until round 6 the
        // WHOLE buffer was cleared here, on the main stream - behind the scan chains' fork event (ev[3], recorded by run_chains_ss), so
        // the span-1 branch on its side stream could write rows the memset then wiped: zero columns in the decoded path on a timing-
        // dependent 2 - 7 % of the headline contig (tests/test_gpu_argmax.py, seen once the suite's order shifted the timing).
        for (int c = 0; c < n_contigs; ++c)
            HIPCHK(hipMemsetAsync(d_gamma_rows.p + (size_t)contig_base[c] * Mp, 0, sizeof(double) * Mp, s));
        if (gamma_side) { HIPCHK(hipEventRecord(ev[22], s)); HIPCHK(hipStreamWaitEvent(stream_hi, ev[22], 0)); }
    }
    // The eigen-row branch (U/W products, rank update, span-Q Hadamard, Y) does not depend on the span-1 branch
    // (log_c, omega_1, rank update); with two streams the short launches of one fill the gaps of the other.
    const bool split_streams = dual_stream && stream2 != nullptr && !slabs_eg.empty();
    const int stats_variant = opt().i(smcpp_opt::O_STATS_VARIANT, 0);
    // Eigen-free statistics of small inputs: the branch rank update of the span > 1 rows -> reduction -> span fold (2 x s_max serial
    // steps) is the critical path of the phase, and a hop between two streams costs ~10 us on this runtime (tools/sync_lab.hip:
    // event record -> wait on another queue; 2 us between two kernels of one queue).  So THAT branch stays on the main stream,
    // directly behind the last pass of the chains and in front of the finalisation, and the two span-1 branches (which have slack)
    // fork to the side streams.  (Rounds 2-3 had it the other way round: 40 us between the chains' end and the first kernel of the
    // critical branch.)  SMCPP_STATS_VARIANT & 4 restores the old arrangement.
    const bool crit_main = eigfree && dual_stream && stream2 != nullptr && !slabs_eg.empty() && !(stats_variant & 6) && n_e_rows < 1000000 &&
                           Mp <= 64;      // (M > 64: chip-filling rank updates, the hops do not matter and the old order is 3 % faster)
    // (round 4, later) with the span fold on the scans (k_span_scan: 23 us instead of 72) and shares in the span-1 reductions the two
    // branches are ~100 and ~77 us: the span > 1 branch is still the longer one and keeps the main stream (922 against 910 evals/s);
    // SMCPP_STATS_VARIANT & 8 gives the main stream to the span-1 branch instead
    const bool span_scan_off = opt().off(smcpp_opt::O_SPAN_SCAN);
    const bool scan_fold = eigfree && ss_active && !span_scan_off;
    const bool swap_main = crit_main && scan_fold && (stats_variant & 8);
    hipStream_t se = crit_main ? (swap_main ? stream2 : s) : split_streams ? ((eigfree && (stats_variant & 1)) ? stream_hi : stream2) : s;
    hipStream_t sp1 = crit_main ? (swap_main ? s : stream2) : s;          // the span-1 branch
    // (scan chains: run_chains_ss has just recorded ev[3] behind the last pass - the fork event, without a second record)
    hipEvent_t ev_fork = ss_active ? ev[3] : ev[8];
    if (split_streams) {
        if (!ss_active) HIPCHK(hipEventRecord(ev[8], s));
        if (se != s) { HIPCHK(hipStreamWaitEvent(se, ev_fork, 0)); arena_wait(se); }
        if (sp1 != s) { HIPCHK(hipStreamWaitEvent(sp1, ev_fork, 0)); arena_wait(sp1); }
    }
    // nothing in the statistics reads log_c any more (the span-1 weights take c itself): the two log-likelihood kernels
    // ride on the eigen stream instead of heading the critical path of the main one
    // ... and on a third stream when there is one: on un-binned data (a million rows per contig) they take 0.1 ms
    // (which form the span-1 statistics take decides which streams are free: details where they are launched, below)
    // M <= 64, from half a million span-1 rows on: ONE pass over the span-1 rows in key-sorted order, single-key slabs - the rank
    // update and the key's gamma sums from the same operands (k_rank_acc<3>); k_s1_scalars and its second read of alpha / beta do
    // not run.  Measured: whole genome (3.6 M span-1 rows, bandwidth-bound) 3.77 -> 3.15 ms of statistics; one 100 Mbp contig
    // (129 k rows, one wavefront per SIMD, latency-bound) 0.208 -> 0.225 ms - there the gamma sums stay a third concurrent
    // branch.  SMCPP_S1_FUSE=0 / 1 forces either form.
    const bool kfuse = (Mp + 63) / 64 == 1 && !save_gamma && !slabs_fk.empty() &&
                       (opt().has(smcpp_opt::O_S1_FUSE) ? opt().i(smcpp_opt::O_S1_FUSE, 0) != 0 : (n_1_rows >= 500000 || crit_main));
    // (round 4: with the span > 1 branch on the main stream the span-1 statistics are ONE side branch in the one-pass form instead
    // of two - 887 against 873 headline evals per second, and 140 MB less traffic per E-step)
    // (eigen-free with the span > 1 branch on the main stream: free at its head, which waits there; round 5: the other eigen-free cases -
    // M > 64, or a million span > 1 rows - no longer make the main stream wait, so the two kernels would delay its weights pass by their
    // 60 - 130 us: third stream there too)
    const bool ll_own = split_streams && stream3 != nullptr && (!eigfree || !crit_main);
    // crit_main with the one-pass span-1 form: the third stream has nothing else to do - the log-likelihood kernels run there,
    // beside both branches instead of at the head of the span-1 branch (joined in front of the finalisation)
    const bool ll3 = crit_main && kfuse && stream3 != nullptr;
    hipStream_t sl = (ll_own || ll3) ? stream3 : (crit_main ? sp1 : eigfree ? s : se);   // (the eigen-free branch is the longer one)
    if (ll_own || ll3) { HIPCHK(hipStreamWaitEvent(sl, ev_fork, 0)); arena_wait(sl); }
    hipLaunchKernelGGL(k_loglik_partial, dim3(llblk, n_contigs), dim3(256), 0, sl, la);
    hipLaunchKernelGGL(k_loglik_final, dim3(n_contigs), dim3(256), 0, sl, la);
    if (ll_own || ll3) HIPCHK(hipEventRecord(ev[19], sl));
    FinArgs fa;
    fa.M = M; fa.Mp = Mp; fa.K = K; fa.G = G; fa.Ke = Ke; fa.n_contigs = n_contigs;
    fa.eb_slab_off = d_eb_slab_off.p; fa.eb_gid = d_eb_gid.p; fa.ce_bucket_off = d_ce_bucket_off.p;
    fa.s1_slab_off = d_s1_slab_off.p; fa.gk_slab_off = d_gk_slab_off.p; fa.g_span = d_g_span.p;
    fa.e_kid = d_e_kid.p; fa.dsc = d_dsc.p; fa.dun = d_dun.p; fa.Prm = d_Prm.p; fa.Pinvrm = d_Pinvrm.p;
    fa.E = d_E.p; fa.Td = d_Td.p; fa.ZS = ZS; fa.red_e = nullptr; fa.red_1 = d_red_1.p; fa.red_g = d_red_g.p; fa.ZG = kfuse ? ZG : 1;
    fa.alpha = d_alpha.p; fa.beta = d_beta.p; fa.contig_base = d_contig_base.p;
    fa.Z = d_Z.p; fa.Y = d_Y.p; fa.xisum = d_xisum.p; fa.gsum = d_gsum.p; fa.gamma0 = d_gamma0.p;
    fa.dpow = d_dpow.p;
    fa.part_e = nullptr;
    const int MMi = Mp * Mp;
    const int nb2 = ceil_div((long long)Mp * Mp, 256);
    AccArgs aa;
    aa.M = M; aa.Mp = Mp; aa.NB = (Mp + 63) / 64; aa.rowinfo = d_rowinfo.p; aa.alpha = d_alpha.p; aa.beta = d_beta.p;
    aa.w1 = d_w1.p; aa.cnorm = d_cnorm.p; aa.E = d_E.p; aa.Xs = d_Xs.p; aa.Ys = d_Ys.p;
    aa.gpart = nullptr; aa.teams = nullptr;
    // (round 5) teams of four slabs reduce their accumulators through LDS: a quarter of the partial bytes (SMCPP_STATS_TEAM=0: one
    // partial per slab, as in rounds 2-4)
    const bool team_env = stats_team_on();
    const bool team2 = team_env && eigfree && !teams_eg.empty();
    const bool team3 = team_env && !teams_fk.empty();
    const bool team0 = team_env && !teams_rk.empty();
    // (round 6) Mp > 128: the rank updates of modes 0 / 2 stage their operand rows through LDS (kernels.hpp: k_rank_acc_wide - one
    // workgroup per team and 256 x 128 block of the output instead of one wavefront per slab and 64 x 64 block); SMCPP_RANK_WIDE=0:
    // the per-wavefront form
    const bool rank_wide = Mp > 128 && !opt().off(smcpp_opt::O_RANK_WIDE);
    auto launch_wide = [&](int mode, const AccArgs &ar, unsigned gx, hipStream_t st) {
        static bool once = false;
        if (!once) {
            HIPCHK(hipFuncSetAttribute((const void *)k_rank_acc_wide<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHK(hipFuncSetAttribute((const void *)k_rank_acc_wide<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHK(hipFuncSetAttribute((const void *)k_rank_acc_wide<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHK(hipFuncSetAttribute((const void *)k_rank_acc_wide<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            once = true;
        }
        const dim3 grid(gx, (unsigned)(((Mp + 255) / 256) * ((Mp + 127) / 128)));
        const bool four = opt().i(smcpp_opt::O_RANK_WIDE, 8) == 4;      // (SMCPP_RANK_WIDE=4: one wavefront per SIMD)
        if (four) {
            if (mode == 0) hipLaunchKernelGGL((k_rank_acc_wide<0, 4>), grid, dim3(256), RW_LDS, st, ar);
            else hipLaunchKernelGGL((k_rank_acc_wide<2, 4>), grid, dim3(256), RW_LDS, st, ar);
        } else {
            if (mode == 0) hipLaunchKernelGGL((k_rank_acc_wide<0, 8>), grid, dim3(512), RW_LDS, st, ar);
            else hipLaunchKernelGGL((k_rank_acc_wide<2, 8>), grid, dim3(512), RW_LDS, st, ar);
        }
    };
    // Eigen-free statistics: the span fold (tens of serial steps on a few CUs) ends the longest dependency chain of the
    // phase, so what it waits for - the rank accumulation of the span > 1 rows - goes FIRST and alone; the span-1 branches start
    // behind it and run while the fold does
    // (small inputs only: from ~10^6 span > 1 rows on, the rank updates are bound by memory parallelism and the two of them
    // running side by side finish sooner than one after the other - whole genome: 3.76 -> 3.36 ms of statistics)
    const bool rank2_early = eigfree && split_streams && !slabs_eg.empty() && !(stats_variant & 2) && n_e_rows < 1000000;
    bool wait17 = false;
    const bool eig_gen2 = !eigfree && NT <= 4 && !slabs_eg.empty();      // (M > 64: the two-kernel form below)
    if (!slabs_eg.empty() && !eig_gen2) {
        d_part_e.alloc(std::max<size_t>(1, team2 ? teams_eg.size() : slabs_eg.size()) * Mp * Mp);
        fa.part_e = d_part_e.p;
        if (eigfree) { d_red_e.alloc(std::max<size_t>(1, eb_gid.size()) * Mp * Mp); fa.red_e = d_red_e.p; }
    }
    if (rank2_early) {
        if (aa.NB != 1) {
            S1Args se_a;
            se_a.M = M; se_a.Mp = Mp; se_a.nslabs = (int)slabs_eg.size(); se_a.slabs = d_slabs_eg.p; se_a.perm = d_perme.p;
            se_a.alpha = d_alpha.p; se_a.beta = d_beta.p; se_a.cnorm = d_cnorm.p; se_a.w1 = d_w1.p; se_a.gpart = d_gpart.p;
            se_a.gamma_rows = nullptr; se_a.only_w1 = 1;
            launch_s1(NPL, se_a, se);
        }
        AccArgs ae = aa;
        ae.nslabs = (int)slabs_eg.size(); ae.slabs = d_slabs_eg.p; ae.perm = d_perme.p; ae.part = d_part_e.p;
        if (team2) ae.teams = d_teams_eg.p;
        if (rank_wide) launch_wide(2, ae, team2 ? (unsigned)teams_eg.size() : (unsigned)ae.nslabs, se);
        else if (team2)
            hipLaunchKernelGGL((k_rank_acc<2, true>), dim3((unsigned)teams_eg.size(), ae.NB * ae.NB), dim3(256), 0, se, ae);
        else
        hipLaunchKernelGGL(k_rank_acc<2>, dim3(ae.nslabs, ae.NB * ae.NB), dim3(64), 0, se, ae);
        if (!crit_main) {
            // (round 5) the span-1 branch waits for this rank update only where its OWN rank update starts: its weights pass
            // (k_s1_scalars, bound by memory round trips) runs beside the weights pass and the head of the rank update of this branch
            HIPCHK(hipEventRecord(ev[17], se));
            wait17 = true;
        }
    }
    // ---- span-1 branch (main stream) ----
    // M <= 64: k_rank_acc forms the weights itself, so the per-key gamma sums (k_s1_scalars + their reduction) are a third
    // independent branch: own stream, joined before the finalisation
    const bool s1_own = !kfuse && dual_stream && stream3 != nullptr && (Mp + 63) / 64 == 1 && !save_gamma && !slabs_sc.empty();
    hipStream_t s1s = s1_own ? stream3 : sp1;
    if (s1_own) {
        if (crit_main) { HIPCHK(hipStreamWaitEvent(s1s, ev_fork, 0)); arena_wait(s1s); }     // (forks where the span-1 branch does: at the chains' end)
        else {
            HIPCHK(hipEventRecord(ev[15], s));
            HIPCHK(hipStreamWaitEvent(s1s, ev[15], 0));
        }
    }
    if (!slabs_sc.empty() && !kfuse) {
        S1Args sa;
        sa.M = M; sa.Mp = Mp; sa.nslabs = (int)slabs_sc.size(); sa.slabs = d_slabs_sc.p; sa.perm = d_perm1.p;
        sa.alpha = d_alpha.p; sa.beta = d_beta.p; sa.cnorm = d_cnorm.p; sa.w1 = d_w1.p; sa.gpart = d_gpart.p;
        sa.gamma_rows = save_gamma ? d_gamma_rows.p : nullptr;
        sa.only_w1 = 0;
        launch_s1(NPL, sa, s1s);
        if (s1_own) {
            hipLaunchKernelGGL(k_sum_parts, dim3(ceil_div(Mp, 256), n_contigs * K, 1), dim3(256), 0, s1s,
                               (const double *)d_gpart.p, (const int *)d_gk_slab_off.p, d_red_g.p, Mp, 1);
            HIPCHK(hipEventRecord(ev[16], s1s));
        } else if (split_streams) HIPCHK(hipEventRecord(ev[14], sp1));
    }
    if (wait17) HIPCHK(hipStreamWaitEvent(sp1, ev[17], 0));
    if (kfuse) {
        d_part_1.alloc(std::max<size_t>(1, team3 ? teams_fk.size() : slabs_fk.size()) * Mp * Mp);
        d_gpart_fk.alloc(slabs_fk.size() * Mp);
        aa.nslabs = (int)slabs_fk.size(); aa.slabs = d_slabs_fk.p; aa.perm = d_perm1.p; aa.permk = nullptr; aa.part = d_part_1.p;
        aa.gpart = d_gpart_fk.p;
        if (team3) {
            aa.teams = d_teams_fk.p;
            hipLaunchKernelGGL((k_rank_acc<3, true>), dim3((unsigned)teams_fk.size(), 1), dim3(256), 0, sp1, aa);
        } else
        hipLaunchKernelGGL(k_rank_acc<3>, dim3(aa.nslabs, 1), dim3(64), 0, sp1, aa);
        hipLaunchKernelGGL(k_sum_parts, dim3(ceil_div(Mp, 256), n_contigs * K, ZG), dim3(256), 0, sp1,
                           (const double *)d_gpart_fk.p, (const int *)d_fk_gk_off.p, d_red_g.p, Mp, ZG);
        hipLaunchKernelGGL(k_sum_parts, dim3(ceil_div(MMi, 256), n_contigs, ZS), dim3(256), 0, sp1,
                           (const double *)d_part_1.p, (const int *)(team3 ? d_fk_c_team_off.p : d_fk_c_off.p), d_red_1.p, MMi, ZS);
    } else
    if (!slabs_rk.empty()) {
        aa.nslabs = (int)slabs_rk.size(); aa.slabs = d_slabs_rk.p; aa.perm = d_perm1.p; aa.permk = d_perm1k.p; aa.part = d_part_1.p;
        if (team0) aa.teams = d_teams_rk.p;
        if (rank_wide) launch_wide(0, aa, team0 ? (unsigned)teams_rk.size() : (unsigned)aa.nslabs, sp1);
        else if (team0)
            hipLaunchKernelGGL((k_rank_acc<0, true>), dim3((unsigned)teams_rk.size(), aa.NB * aa.NB), dim3(256), 0, sp1, aa);
        else
        hipLaunchKernelGGL(k_rank_acc<0>, dim3(aa.nslabs, aa.NB * aa.NB), dim3(64), 0, sp1, aa);
    }
    // the per-key gamma sums only need the span-1 scalars: with two streams their reduction runs at the tail of the eigen
    // stream (which finishes earlier) instead of between the two rank-update kernels of the main one
    const bool gsum_on_se = split_streams && !slabs_sc.empty() && !s1_own && !kfuse;
    if (!gsum_on_se && !s1_own && !kfuse)
        hipLaunchKernelGGL(k_sum_parts, dim3(ceil_div(Mp, 256), n_contigs * K, 1), dim3(256), 0, sp1,
                           (const double *)d_gpart.p, (const int *)d_gk_slab_off.p, d_red_g.p, Mp, 1);
    if (!kfuse)
    hipLaunchKernelGGL(k_sum_parts, dim3(ceil_div(MMi, 256), n_contigs, ZS), dim3(256), 0, sp1,
                       (const double *)d_part_1.p, (const int *)(team0 ? d_s1_team_off.p : d_s1_slab_off.p), d_red_1.p, MMi, ZS);
    HIPCHK(hipEventRecord(ev[4], sp1));
    // ---- eigen branch (second stream when available) ----
    fa.eigfree = eigfree ? 1 : 0;
    if (!slabs_eg.empty() && eigfree) {
        // weights of the span > 1 rows (as those of the span-1 rows), rank accumulation per (span, key) group, deterministic
        // reduction of the slab partials, then the span fold per (contig, key)
        S1Args se_a;
        se_a.M = M; se_a.Mp = Mp; se_a.nslabs = (int)slabs_eg.size(); se_a.slabs = d_slabs_eg.p; se_a.perm = d_perme.p;
        se_a.alpha = d_alpha.p; se_a.beta = d_beta.p; se_a.cnorm = d_cnorm.p; se_a.w1 = d_w1.p; se_a.gpart = d_gpart.p;
        se_a.gamma_rows = nullptr; se_a.only_w1 = 1;
        if (!rank2_early) {
            if (aa.NB != 1) launch_s1(NPL, se_a, se);          // M <= 64: k_rank_acc<2> forms the weights itself
            AccArgs ae = aa;
            ae.nslabs = (int)slabs_eg.size(); ae.slabs = d_slabs_eg.p; ae.perm = d_perme.p; ae.part = d_part_e.p;
            if (team2) ae.teams = d_teams_eg.p;
            if (rank_wide) launch_wide(2, ae, team2 ? (unsigned)teams_eg.size() : (unsigned)ae.nslabs, se);
            else if (team2)
                hipLaunchKernelGGL((k_rank_acc<2, true>), dim3((unsigned)teams_eg.size(), ae.NB * ae.NB), dim3(256), 0, se, ae);
            else
            hipLaunchKernelGGL(k_rank_acc<2>, dim3(ae.nslabs, ae.NB * ae.NB), dim3(64), 0, se, ae);
        }
        if (!eb_gid.empty())                                     // ONE share per bucket: k_span_F reads it on its serial path
            hipLaunchKernelGGL(k_sum_parts, dim3(ceil_div(MMi, 256), (unsigned)eb_gid.size(), 1), dim3(256), 0, se,
                               (const double *)d_part_e.p, (const int *)(team2 ? d_eb_team_off.p : d_eb_slab_off.p), d_red_e.p, MMi, 1);
        const size_t shm = (size_t)2 * Mp * (Mp + 1) * sizeof(double);
        // SMCPP_SPAN_FH=1: the one-workgroup-per-(contig, key) fold (M <= 64) instead of the strip kernels
        // (round 4) the fold on the SCANS: a row of F times A / A times a column of H is one O(M) step of the backward / forward chain
        // operator, so one wavefront per row / column walks all s_max steps on its own (chains_ss.hpp: k_span_scan) - no matrix
        // product, no barrier.  SMCPP_SPAN_SCAN=0: the matrix-core strips of round 3.
        if (scan_fold) {
            d_Fall.alloc((size_t)n_contigs * Ke * ss_max_span * Mp * Mp);
            const int nwav = n_contigs * Ke * M;
            const dim3 grid(ceil_div(nwav, 4)), block(256);
            switch (NPL) {
#define SC_(x) case x: hipLaunchKernelGGL((k_span_scan<x, 0>), grid, block, 0, se, ss_args, fa, ss_max_span, d_Fall.p); \
                       hipLaunchKernelGGL((k_span_scan<x, 1>), grid, block, 0, se, ss_args, fa, ss_max_span, d_Fall.p); break;
                SC_(1) SC_(2) SC_(3) SC_(4) SC_(8)
                default: SC_(16)
#undef SC_
            }
        } else {
            // the fold on the matrix cores (round 3; SMCPP_SPAN_SCAN=0 or no scan chains): strips of 16 rows (F) / columns (H), one
            // workgroup each, F_t through scratch
            d_Fall.alloc((size_t)n_contigs * Ke * ss_max_span * Mp * Mp);
            const int nstrip = NT, nwg = n_contigs * Ke * nstrip;
#define B_(x) { hipLaunchKernelGGL((k_span_big<x, 0>), dim3(nwg), dim3(64 * x), 0, se, fa, ss_max_span, d_Fall.p); \
                hipLaunchKernelGGL((k_span_big<x, 1>), dim3(nwg), dim3(64 * x), 0, se, fa, ss_max_span, d_Fall.p); }
            if (NT == 1) B_(1) else if (NT == 2) B_(2) else if (NT == 3) B_(3) else if (NT == 4) B_(4)
            else if (NT <= 8) B_(8) else if (NT <= 12) B_(12) else B_(16)
#undef B_
        }
    }
    if (eig_gen2) {
        // generation 2 (M <= 64): slabs that mix span groups, the span-Q weighting inside the accumulation (k_eig_fused2)
        UWArgs ua;
        ua.M = M; ua.Mp = Mp; ua.nslabs = (int)slabs_ek.size(); ua.slabs = d_slabs_ek.p; ua.perm = d_perme.p;
        ua.alpha = d_alpha.p; ua.beta = d_beta.p; ua.g_eig = d_g_eig.p; ua.g_scale = d_g_scale.p;
        ua.dpow = d_dpow.p; ua.PinvT = d_PinvT.p; ua.Prm = d_Prm.p; ua.Xs = nullptr; ua.Ys = nullptr;
        ua.pos_gid = d_epos_gid.p; ua.g_span = d_g_span.p;
        const int nce = n_contigs * Ke;
        const int LEN = MMi + Mp;                       // per slab: the M x M accumulator and the M diagonal sums
        d_part_ek.alloc(std::max<size_t>(1, slabs_ek.size()) * LEN);
        // shares of the cross-slab reduction: ~32 slabs each (un-binned data: thousands of slabs on a handful of (contig, key) pairs)
        int max_sl = 1;
        for (int ce = 0; ce < nce; ++ce) max_sl = std::max(max_sl, ek_slab_off[ce + 1] - ek_slab_off[ce]);
        const int nsh = std::max(1, std::min(128, (max_sl + 31) / 32));
        d_red_ek.alloc((size_t)nce * nsh * LEN);
        const int nblk = ceil_div(ua.nslabs, 4);
        const size_t shm = (size_t)2 * (16 * NT) * (16 * NT + 1) * sizeof(double);
        switch (NT) {
#define F_(x) case x: { static bool once = false; if (!once) { HIPCHK(hipFuncSetAttribute((const void *)k_eig_fused2<x>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; } \
                    hipLaunchKernelGGL(k_eig_fused2<x>, dim3(nblk), dim3(256), shm, se, ua, d_part_ek.p); } break;
            F_(1) F_(2) F_(3)
            default: F_(4)
#undef F_
        }
        hipLaunchKernelGGL(k_sum_parts, dim3(ceil_div(LEN, 256), (unsigned)nce, nsh), dim3(256), 0, se,
                           (const double *)d_part_ek.p, (const int *)d_ek_slab_off.p, d_red_ek.p, LEN, nsh);
        hipLaunchKernelGGL(k_fin_Z2, dim3(nb2, nce), dim3(256), 0, se, fa, (const double *)d_red_ek.p, nsh);
        hipLaunchKernelGGL(k_fin_Y, dim3(nb2, nce), dim3(256), 0, se, fa);
    }
    if (!slabs_eg.empty() && !eigfree && !eig_gen2) {
        UWArgs ua;
        ua.M = M; ua.Mp = Mp; ua.nslabs = (int)slabs_eg.size(); ua.slabs = d_slabs_eg.p; ua.perm = d_perme.p;
        ua.alpha = d_alpha.p; ua.beta = d_beta.p; ua.g_eig = d_g_eig.p; ua.g_scale = d_g_scale.p;
        ua.dpow = d_dpow.p; ua.PinvT = d_PinvT.p; ua.Prm = d_Prm.p; ua.Xs = d_Xs.p; ua.Ys = d_Ys.p; ua.pos_gid = nullptr; ua.g_span = nullptr;
        {
            launch_uw(NT, ua, se);
            AccArgs ae = aa;
            ae.nslabs = (int)slabs_eg.size(); ae.slabs = d_slabs_eg.p; ae.perm = d_perme.p; ae.part = d_part_e.p;
            hipLaunchKernelGGL(k_rank_acc<1>, dim3(ae.nslabs, ae.NB * ae.NB), dim3(64), 0, se, ae);
        }
        // (no reduction pass over the slab partials: k_fin_Z sums the slabs of a bucket itself)
    }
    if (Ke > 0 && !eigfree && !eig_gen2) {
        // slices of the groups of one (contig, key): enough blocks to fill the chip when there are many groups
        int max_b = 0;
        for (size_t ce = 0; ce + 1 < ce_bucket_off.size(); ++ce) max_b = std::max(max_b, ce_bucket_off[ce + 1] - ce_bucket_off[ce]);
        const int nsl = std::max(1, std::min(std::min(256, max_b), 2048 / std::max(1, nb2 * n_contigs * Ke)));
        if (nsl > 1) { d_Zpart.alloc((size_t)nsl * n_contigs * Ke * Mp * Mp); fa.Zpart = d_Zpart.p; }
        else fa.Zpart = nullptr;
        hipLaunchKernelGGL(k_fin_Z, dim3(nb2, n_contigs * Ke, nsl), dim3(256), 0, se, fa);
        if (nsl > 1) hipLaunchKernelGGL(k_fin_Zsum, dim3(nb2, n_contigs * Ke), dim3(256), 0, se, fa, nsl, n_contigs * Ke);
        hipLaunchKernelGGL(k_fin_Y, dim3(nb2, n_contigs * Ke), dim3(256), 0, se, fa);
    }
    if (gsum_on_se) {
        HIPCHK(hipStreamWaitEvent(se, ev[14], 0));
        hipLaunchKernelGGL(k_sum_parts, dim3(ceil_div(Mp, 256), n_contigs * K, 1), dim3(256), 0, se,
                           (const double *)d_gpart.p, (const int *)d_gk_slab_off.p, d_red_g.p, Mp, 1);
    }
    if (crit_main) {
        // (the third stream joins the side stream, and the main stream waits for ONE event: every wait is a barrier packet of a few
        // microseconds on the queue it is put on, signalled or not)
        hipStream_t side = swap_main ? se : sp1;
        if (ll3) HIPCHK(hipStreamWaitEvent(side, ev[19], 0));
        HIPCHK(hipEventRecord(ev[9], side));
        HIPCHK(hipStreamWaitEvent(s, ev[9], 0));
    } else if (split_streams) {
        HIPCHK(hipEventRecord(ev[9], se));
        HIPCHK(hipStreamWaitEvent(s, ev[9], 0));
    }
    if (s1_own) HIPCHK(hipStreamWaitEvent(s, ev[16], 0));
    {
        const int nbf = nb2 + ceil_div((long long)(K + 1) * Mp, 256);
        // nothing follows the finalisation on this stream when gamma rows are not asked for: its last block signals the host
        // (only while the launch is small: every block ends with a device-scope fence and an atomic on ONE counter - 40 500 blocks of
        // 1 500 contigs took 3.2 ms for it, 80 ns each, tools/many_contigs_probe.py; beyond 128 blocks the one-thread kernel signals: 100 contigs 0.37 -> 0.20 ms of finalisation, 22 contigs 0.049 -> 0.03)
        done_folded = fold_done_epoch != 0 && !save_gamma && !ll_own && (long long)nbf * n_contigs <= 128;
        if (done_folded) {
            if (!d_fin_ctr.p) { d_fin_ctr.alloc(1); HIPCHK(hipMemsetAsync(d_fin_ctr.p, 0, sizeof(unsigned), s)); fin_target = 0; }
            fin_target += (unsigned)nbf * (unsigned)n_contigs;
        }
        hipLaunchKernelGGL(k_fin_both, dim3(nbf, n_contigs), dim3(256), 0, s, fa, nb2, d_fin_ctr.p, fin_target,
                           done_folded ? d_done_view : (int *)nullptr, fold_done_epoch);
    }
    if (save_gamma && n_e_rows > 0) {
        hipStream_t sg = gamma_side ? stream_hi : s;
        GammaRowArgs ga;
        ga.M = M; ga.Mp = Mp; ga.nrows = (int)n_e_rows; ga.perm = d_perme.p; ga.row_slab = d_erow_slab.p;
        ga.slabs = d_slabs_eg.p; ga.g_eig = d_g_eig.p; ga.g_span = d_g_span.p; ga.dun = d_dun.p; ga.dsc = d_dsc.p; ga.dpow = d_dpow.p;
        ga.Prm = d_Prm.p; ga.Pinvrm = d_Pinvrm.p; ga.PinvT = d_PinvT.p; ga.Sq = nullptr;
        ga.alpha = d_alpha.p; ga.beta = d_beta.p; ga.gamma_rows = d_gamma_rows.p; ga.erow_desc = d_erow_desc.p;
        const bool mfma_rows = NT <= 4;          // (M > 64: the scalar kernel on a span-Q table in memory)
        gamma_pieces_last = false;
        if (gamma_pieces) {
            // (round 6) long rows at 64 < M <= 256: eigen-power pieces + scan steps (chains_ss.hpp: k_piece_vectors)
            build_gamma_pieces();
            const int MS = 64 * NPL;
            d_gp_gen.upload(ss_gen, sg);
            d_gp_cs.alloc((size_t)Ke * 2 * Mp);
            d_gp_l2d.alloc((size_t)Ke * Mp);
            const size_t npc = gp_pieces.size();
            d_gp_pvf.alloc(npc * Mp); d_gp_pvb.alloc(npc * Mp); d_gp_pgam.alloc(npc * Mp);
            PieceArgs pa;
            pa.M = M; pa.Mp = Mp; pa.npieces = (int)npc; pa.ntiles = (int)gp_tiles.size();
            pa.pieces = d_gp_pieces.p; pa.tiles = d_gp_tiles.p; pa.Xs = d_Xs.p; pa.Ys = d_Ys.p; pa.dsc = d_dsc.p;
            pa.PT = d_PT.p; pa.Pinvrm = d_Pinvrm.p; pa.cs = d_gp_cs.p; pa.l2d = d_gp_l2d.p; pa.pvf = d_gp_pvf.p; pa.pvb = d_gp_pvb.p; pa.pgam = d_gp_pgam.p;
            hipLaunchKernelGGL(k_piece_rowsums, dim3(ceil_div(Ke * 2 * Mp, 256)), dim3(256), 0, sg, Ke, Mp, (const double *)d_PT.p,
                               (const double *)d_Pinvrm.p, d_gp_cs.p, (const double *)d_dsc.p, d_gp_l2d.p);
            if (pa.ntiles > 0) {
                static bool once = false;
                if (!once) { HIPCHK(hipFuncSetAttribute((const void *)k_piece_vectors, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; }
                const size_t shm = (size_t)4 * 16 * (Mp + 4) * sizeof(double);
                const int nblk = std::min(2 * ceil_div(pa.ntiles, 4), 2048);
                hipLaunchKernelGGL(k_piece_vectors, dim3(nblk), dim3(256), shm, sg, pa);
            }
            SsArgs sa = SsArgs();
            sa.M = M; sa.Mp = Mp;
            const double *gd = d_gp_gen.p;
            sa.f_dc = gd; sa.f_g = gd + MS; sa.f_cg = gd + 2 * MS; sa.f_b = gd + 3 * MS; sa.f_a = gd + 4 * MS; sa.f_d = gd + 5 * MS;
            sa.b_dc = gd + 6 * MS; sa.b_g = gd + 7 * MS; sa.b_b = gd + 8 * MS; sa.b_a = gd + 9 * MS;
            sa.c0 = ss_c0;
            const int nw = (int)std::min<size_t>(npc, 4096);
            d_gpark.alloc((size_t)nw * 64 * MS);
            const dim3 grid(ceil_div(nw, 4)), block(256);
            switch (NPL) {
#define GP_(x) case x: hipLaunchKernelGGL((k_gamma_rows_scan<x, true>), grid, block, 0, sg, sa, ga, (const RowInfo *)d_rowinfo.p, \
                                          (const double *)d_E.p, d_gpark.p, 64, nw, pa); break;
                GP_(2) GP_(3)
                default: GP_(4)
#undef GP_
            }
            hipLaunchKernelGGL(k_gamma_merge_pieces, dim3((unsigned)ceil_div((long long)n_e_rows * Mp, 256)), dim3(256), 0, sg, Mp, (int)n_e_rows,
                               (const int *)d_gp_pfirst.p, (const GPiece *)d_gp_pieces.p, (const double *)d_gp_pgam.p, d_gamma_rows.p);
            gamma_pieces_last = true;
        } else
        if (eigfree) {
            // no eigensystem on this path: 2 span - 1 scan steps per row, one (persistent) wavefront per row, the forward vectors parked
            // as floats in the wavefront's piece of a scratch buffer
            const int nw = (int)std::min<long long>((long long)n_e_rows, NPL >= 8 ? 1024 : 4096);
            d_gpark.alloc((size_t)nw * ss_max_span * 64 * NPL);
            const dim3 grid(ceil_div(nw, 4)), block(256);
            switch (NPL) {
#define GS_(x) case x: hipLaunchKernelGGL((k_gamma_rows_scan<x>), grid, block, 0, sg, ss_args, ga, (const RowInfo *)d_rowinfo.p, \
                                          (const double *)d_E.p, d_gpark.p, ss_max_span, nw); break;
                GS_(1) GS_(2) GS_(3) GS_(4) GS_(8)
                default: GS_(16)
#undef GS_
            }
        } else
        if (!mfma_rows) {
            d_Sq.alloc((size_t)G * Mp * Mp);
            hipLaunchKernelGGL(k_span_q, dim3(nb2, G), dim3(256), 0, sg, M, Mp, G, (const int *)d_g_span.p,
                               (const int *)d_g_eig.p, (const double *)d_dsc.p, (const double *)d_dpow.p, d_Sq.p);
            ga.Sq = d_Sq.p;
        }
        if (eigfree || gamma_pieces) {
        } else if (mfma_rows) {
            // one launch per (contig, eigen key): a workgroup shares one LDS copy of P, Pinv and the reciprocal eigenvalue differences
            // (NT > 2: the reciprocal differences live in registers and the fold tile is half as wide - four wavefronts fit as well)
            const int NW = 4;
            const size_t shm2 = (size_t)((NT <= 2 ? 3 : 2) * Mp * (Mp + 1) + NW * (2 * 16 * (Mp + 1) + Mp * (NT <= 2 ? 17 : 9) + Mp) + Mp) * sizeof(double);
            for (int ce = 0; ce < n_contigs * Ke; ++ce) {
                const int q0 = ce_row_off[ce], q1 = ce_row_off[ce + 1];
                if (q1 <= q0) continue;
                const int nbatch = std::max(1, std::min(4, (q1 - q0 + 16 * NW * 2048 - 1) / (16 * NW * 2048)));   // batches of 16 rows per wavefront
                const int nblk = ceil_div(q1 - q0, 16 * NW * nbatch);
                switch (NT) {
#define G_(x) case x: { static bool once = false; if (!once) { HIPCHK(hipFuncSetAttribute((const void *)k_gamma_rows_b<x>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; } \
                        hipLaunchKernelGGL(k_gamma_rows_b<x>, dim3(nblk), dim3(256), shm2, sg, ga, q0, q1, ce % Ke, nbatch); } break;
                    G_(1) G_(2) G_(3)
                    default: G_(4)
#undef G_
                }
            }
        } else {
            const size_t shm = (size_t)(3 * Mp + 256) * sizeof(double);
            hipLaunchKernelGGL(k_gamma_rows_eig, dim3((unsigned)n_e_rows), dim3(256), shm, sg, ga);
        }
            if (gamma_side) { HIPCHK(hipEventRecord(ev[23], sg)); HIPCHK(hipStreamWaitEvent(s, ev[23], 0)); }
    }
    HIPCHK(hipGetLastError());
    if (ll_own) HIPCHK(hipStreamWaitEvent(s, ev[19], 0));
    HIPCHK(hipEventRecord(ev[5], s));
    stats_enqueued = true;
}

// The pieces of the eigen rows (chains_ss.hpp: k_piece_vectors): at most 64 positions each, in the order of the sorted eigen-row
// permutation (so the pieces of one (contig, eigen key) are contiguous and a row's pieces are in position order); tiles of sixteen
// pieces of ONE eigen key that need an interpolated vector (every piece of a row that has more than one).
long long smcpp_im::gamma_piece_count() {
    if (gp_count < 0) {
        gp_count = 0;
        for (size_t q = 0; q < perme.size(); ++q) gp_count += (groups[slabs_eg[erow_slab[q]].aux].span + 63) / 64;
    }
    return gp_count;
}

void smcpp_im::build_gamma_pieces() {
    if (gp_built) return;
    constexpr int PL = 64;
    gp_pieces.clear(); gp_tiles.clear(); gp_pfirst.clear();
    gp_pfirst.reserve(perme.size() + 1);
    GTile cur; cur.es = -1; cur.cnt = 0;
    auto flush = [&]() { if (cur.cnt > 0) { for (int k = cur.cnt; k < 16; ++k) cur.pid[k] = cur.pid[cur.cnt - 1]; gp_tiles.push_back(cur); } cur.cnt = 0; };
    for (size_t q = 0; q < perme.size(); ++q) {
        const Slab &sl = slabs_eg[erow_slab[q]];
        const Group &gr = groups[sl.aux];
        const long long row = sl.base + perme[q];
        gp_pfirst.push_back((int)gp_pieces.size());
        const int np = (gr.span + PL - 1) / PL;
        for (int j = 0; j < np; ++j) {
            GPiece pc;
            pc.row = row; pc.q = (int)q; pc.o0 = j * PL; pc.len = std::min(PL, gr.span - j * PL); pc.span = gr.span; pc.kid = gr.kid; pc.es = gr.eig;
            if (np > 1) {
                if (cur.es != gr.eig || cur.cnt == 16) flush();
                cur.es = gr.eig;
                cur.pid[cur.cnt++] = (int)gp_pieces.size();
            }
            gp_pieces.push_back(pc);
        }
    }
    flush();
    gp_pfirst.push_back((int)gp_pieces.size());
    d_gp_pieces.upload(gp_pieces, stream);
    d_gp_tiles.upload(gp_tiles, stream);
    d_gp_pfirst.upload(gp_pfirst, stream);
    gp_built = true;
}

// Event intervals of the last E-step -> timing[] (lazily: see estep)
void smcpp_im::resolve_timing() {
    if (!timing_pending) return;
    timing_pending = false;
    HIPCHK(hipSetDevice(device));
    (void)hipEventSynchronize(ev[5]);
    float f_ms = 0, b_ms = 0, s_ms = 0, fin_ms = 0;
    if (ss_active) {
        // one launch per pass for both directions, timed below
    } else if (chains_dual) {
        (void)hipEventElapsedTime(&f_ms, ev[1], ev[7]);   // forward passes (main stream)
        (void)hipEventElapsedTime(&b_ms, ev[2], ev[3]);   // backward passes (second stream), overlapping the forward ones
    } else {
        (void)hipEventElapsedTime(&f_ms, ev[1], ev[2]);
        (void)hipEventElapsedTime(&b_ms, ev[2], ev[3]);
    }
    float chains_ms = 0;
    if (ss_active) {
        // every pass of both directions between two events: ev[10] in front of the first launch, ev[3] behind the last one (a rare
        // round that needs more passes than were launched up front includes the host's look at the flags)
        (void)hipEventElapsedTime(&chains_ms, ev[10], ev[3]);
        f_ms = b_ms = chains_ms;
    } else (void)hipEventElapsedTime(&chains_ms, ev[1], ev[3]);
    if (prepass_launched) {
        // pass 0 ran before ev[1] (concurrently with the host eigensolve): add its kernel intervals
        (void)hipEventElapsedTime(&pre_f_ms, ev[10], ev[11]);
        (void)hipEventElapsedTime(&pre_b_ms, ev[12], ev[13]);
        f_ms += pre_f_ms; b_ms += pre_b_ms;
        chains_ms += std::max(pre_f_ms, pre_b_ms);
    }
    (void)hipEventElapsedTime(&s_ms, ev[3], ev[4]);
    (void)hipEventElapsedTime(&fin_ms, ev[4], ev[5]);
    (void)hipGetLastError();      // an interval over an event this E-step never recorded must not surface in the next launch check
    timing[0] = t_host01;
    timing[1] = chains_ms;   // wall time of both chains (they overlap in dual-stream mode)
    timing[2] = f_ms; timing[3] = b_ms; timing[4] = s_ms; timing[5] = fin_ms;
    timing[6] = t_host12;
    timing[7] = last_fwd_passes; timing[8] = last_bwd_passes;
}

void smcpp_im::estep() {
    if (std::isnan(theta) || std::isnan(rho) || std::isnan(alpha))
        throw std::runtime_error("theta / rho / alpha must be set");
    HIPCHK(hipSetDevice(device));
    timing_pending = false;          // (intervals nobody asked for: the events are about to be recorded again)
    auto t0 = std::chrono::steady_clock::now();
    HostTrace tr;
    prepare_params();
    tr.mark("estep: prepare_params");
    host_timing[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if ((int)pi.size() != M || (!T_lazy && (int)T.size() != M * M) || (!E_on_dev && (int)E.size() != K * M))
        throw std::runtime_error("parameters are not set");
    HIPCHK(hipEventRecord(ev[0], stream));
    // span > 1 rows without an eigensystem (kernels.hpp: k_span_fold): the span is expanded by smax steps of two M x M products
    const bool eigfree_off = opt().off(smcpp_opt::O_EIGFREE);
    // (round 6) save_gamma keeps the eigen-free path: the per-row posteriors of the span > 1 rows come from scan steps as well
    // (chains_ss.hpp: k_gamma_rows_scan); SMCPP_GAMMA_SCAN=0: eigensystems, as in rounds 1-5 (M <= 256)
    const bool gamma_scan = !opt().off(smcpp_opt::O_GAMMA_SCAN);
    const bool eigfree_static = !eigfree_off && Mp <= 1024 && ss_max_span <= 64 && (!save_gamma || gamma_scan);
    if (Mp > 256 && !(ss_static && eigfree_static))
        throw std::runtime_error("more than 256 hidden states: only the scan chains with eigen-free statistics are built (rows longer than "
                                 "64 positions are cut into pieces at construction unless SMCPP_SPLIT_SPANS=0 / SMCPP_EIGFREE=0 forbid it or "
                                 "the pieces would not fit the device)");
    // only the lean path (scan chains + eigen-free statistics) reads a device-prepared emission table from HBM alone (its
    // underflow bound is checked by the kernel that forms the table); eigensystems, operand layouts, the dense chains and the
    // bound for longer spans need the host copy
    if (E_on_dev && !(ss_static && eigfree_static)) sync_host_E();
    ss_active = ss_static && ss_extract_generators();
    if (Mp > 256 && !ss_active)
        throw std::runtime_error("more than 256 hidden states: the transition matrix must have the structure of the reference's "
                                 "HJTransition (the dense fallback kernels stop at 256)");
    tr.mark("estep: extract generators");
    if (!ss_active) ss_warm_valid = false;
    eigfree = ss_active && eigfree_static;
    if (E_on_dev && !(ss_active && eigfree)) sync_host_E();      // (a transition matrix without the structure)
    if (ss_active && !ss_hybrid) { prepass_launched = false; static_packed = false; ss_launch_initial(); }   // the chains need no eigensystem: they start now
    else if (ss_active) { prepass_launched = false; static_packed = false; }
    else stage_static_and_prepass();   // (when eligible) pass 0 of both chains starts now, on eigen-free operands
    tr.mark("estep: first launches");
    host_prep_and_upload();   // the reference rebuilds the eigensystems on every E-step (inference_manager.cpp:112)
    tr.mark("estep: host_prep_and_upload");
    if (ss_active && ss_hybrid) ss_launch_initial();       // hybrid rows read the eigensystems: the chains start behind them
    auto t1 = std::chrono::steady_clock::now();
    if (ss_active) run_chains_ss(); else run_chains();
    tr.mark("estep: chains (host view)");
    run_stats();
    tr.mark("estep: statistics enqueued");
    if (E_on_dev) {
        dprep->check_flags();
        if (ss_active && dprep->flags()[2]) {
            // an emission entry so small that `span` scan steps underflow (ss_extract_generators' bound, evaluated by the kernel
            // that formed the table): this E-step is redone on the dense kernels from the host copy of the same parameters
            sync_host_E();
            estep();
            return;
        }
    }
    auto t2 = std::chrono::steady_clock::now();
    // the event intervals are read when somebody asks for them (smcpp_last_timing, the debug log): ev[5] sits BEHIND the completion
    // word the host has just seen, so querying it here would mean waiting for it
    t_host01 = std::chrono::duration<double, std::milli>(t1 - t0).count();
    t_host12 = std::chrono::duration<double, std::milli>(t2 - t1).count();
    host_timing[3] = t_host01;
    timing_pending = true;
    if (g_logger_cb) {
        resolve_timing();
        log_msg("DEBUG", "E-step: %d contig(s), %lld rows, M = %d, K = %d keys; host %.3f ms, chains %.3f ms (%d forward / %d "
                "backward passes), statistics %.3f ms; loglik[0] = %.10g", n_contigs, total_rows - n_contigs, M, K, timing[0],
                timing[1], last_fwd_passes, last_bwd_passes, timing[4] + timing[5], loglik.empty() ? 0.0 : loglik[0]);
    }
    stats_on_host = false;
    have_reduced = false;
    if (qdev) qdev->stats_ready = false;
    gamma_valid = save_gamma;
    estep_done = true;
    dirty = false;
}

void smcpp_im::fetch_stats() {
    if (stats_on_host) return;
    if (!estep_done) {
        // the statistics of a freshly constructed HMM (hmm.cpp:8-29): xisum = 0, gamma = 0 and per key the positions it
        // covers weighted by the default model's initial distribution - what Q() sees before the first E-step (the
        // reference derives its regularisation weight from exactly that value, smcpp/analysis/analysis.py:120-125)
        h_xisum.assign((size_t)n_contigs * M * M, 0.0);
        h_gamma0.assign((size_t)n_contigs * M, 0.0);
        h_gsum.assign((size_t)n_contigs * K * M, 0.0);
        for (int c = 0; c < n_contigs; ++c)
            for (int k = 0; k < K; ++k)
                for (int i = 0; i < M; ++i)
                    h_gsum[((size_t)c * K + k) * M + i] = span_sum[(size_t)c * K + k] * pi_default[i];
        stats_on_host = true;
        return;
    }
    HIPCHK(hipSetDevice(device));
    std::vector<double> x((size_t)n_contigs * Mp * Mp), g((size_t)n_contigs * K * Mp), g0((size_t)n_contigs * Mp);
    HIPCHK(hipMemcpy(x.data(), d_xisum.p, x.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(g.data(), d_gsum.p, g.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(g0.data(), d_gamma0.p, g0.size() * sizeof(double), hipMemcpyDeviceToHost));
    h_xisum.assign((size_t)n_contigs * M * M, 0.0);
    h_gsum.assign((size_t)n_contigs * K * M, 0.0);
    h_gamma0.assign((size_t)n_contigs * M, 0.0);
    for (int c = 0; c < n_contigs; ++c) {
        for (int i = 0; i < M; ++i) {
            h_gamma0[(size_t)c * M + i] = g0[(size_t)c * Mp + i];
            for (int j = 0; j < M; ++j)
                h_xisum[((size_t)c * M + i) * M + j] = x[((size_t)c * Mp + i) * Mp + j];
        }
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < M; ++i)
                h_gsum[((size_t)c * K + k) * M + i] = g[((size_t)c * K + k) * Mp + i];
    }
    stats_on_host = true;
}

