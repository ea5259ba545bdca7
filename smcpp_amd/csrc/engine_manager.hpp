// engine_manager.hpp - part of the ONE translation unit engine.hip (included there, in order; not a standalone header):
// the manager (struct smcpp_im): observation layout, chunks, slabs, device allocation.
struct smcpp_im {
    // ---- static problem description -------------------------------------------------------------------------
    int npop = 1, keylen = 3, M = 0, Mp = 0, NPL = 1, NT = 1, n_contigs = 0, K = 0, G = 0, Ke = 0;
    int n[2] = {0, 0}, na[2] = {2, 0};
    double polarization_error = 0.5;
    std::vector<double> hs;
    std::unique_ptr<smcpp_host::TwoPopPrep> twopop_prep;     // two-population preparation (key -> tensor-bin tables cached inside)
    std::unique_ptr<TwoPopDevCsfs> twopop_dev;               // ... its two batched conditioned-SFS problems on the device (values only)
    std::vector<int> keys;                 // [K][keylen], lexicographic
    std::vector<int> Ls;
    std::vector<long long> contig_base;    // row index of ell = 0 of each contig
    long long total_rows = 0;              // sum (L+1)
    std::vector<RowInfo> rowinfo;          // host copy
    std::vector<Group> groups;             // sorted by (eig/kid, span)
    std::vector<int> eig_kid;              // [Ke]
    std::vector<int> eig_of_key;           // [K]
    std::vector<unsigned char> present;    // [n_contigs][K] key occurs in contig
    std::vector<double> span_sum;          // [n_contigs][K] positions covered by the key in the contig
    std::vector<double> pi_default;        // [M] initial distribution of the constant-size default model (defaultEta)
    std::vector<unsigned char> key_nbpos;  // [K] key.nb() > 0
    std::vector<Chunk> chunks;
    int max_chunks_per_contig = 1;
    int user_rows_per_chunk = 0;
    // sorted permutations and slabs
    std::vector<int> perm1, perme;
    std::vector<int2> perm1k;
    std::vector<Slab> slabs_sc, slabs_rk, slabs_eg;   // span-1 scalar slabs, span-1 rank slabs, eigen slabs
    std::vector<int> gk_slab_off, s1_slab_off, eb_slab_off, eb_gid, ce_bucket_off, erow_slab;
    // fused span-1 statistics (M <= 64): single-key slabs over the key-sorted span-1 rows (perm1), with their ranges per contig
    // (rank partials) and per (contig, key) (gamma partials)
    std::vector<Slab> slabs_fk;
    std::vector<int> fk_c_off, fk_gk_off;
    DevBuf<Slab> d_slabs_fk;
    DevBuf<int> d_fk_c_off, d_fk_gk_off;
    DevBuf<double> d_gpart_fk;
    // teams (round 5): up to four consecutive slabs of ONE reduction range share a workgroup of k_rank_acc<., true> and one partial
    std::vector<int2> teams_fk, teams_eg, teams_rk;
    std::vector<int> fk_c_team_off, eb_team_off, s1_team_off;
    DevBuf<int2> d_teams_fk, d_teams_eg, d_teams_rk;
    DevBuf<int> d_fk_c_team_off, d_eb_team_off, d_s1_team_off;
    // generation-2 eigen statistics (M <= 64): slabs over the sorted eigen rows of a (contig, eigen key) that MIX span groups
    std::vector<Slab> slabs_ek;
    std::vector<int> ek_slab_off, epos_gid;
    std::vector<int4> erow_desc;           // per sorted eigen-row position: {row low / high word, group, span} (k_gamma_rows_b)
    DevBuf<int4> d_erow_desc;
    DevBuf<Slab> d_slabs_ek;
    DevBuf<int> d_ek_slab_off, d_epos_gid;
    DevBuf<double> d_part_ek, d_red_ek;
    std::vector<int> ce_row_off;           // [n_contigs*Ke + 1] first position in perme of every (contig, eigen key)
    long long n_e_rows = 0, n_1_rows = 0;
    // ---- parameters -------------------------------------------------------------------------------------------
    double theta = NAN, rho = NAN, alpha = 1.0;
    bool have_raw = false, dirty = true, params_fresh = false;
    std::vector<double> pi, T, E;          // [M], [M*M], [K*M]
    smcpp_host::ModelParams model;         // a, s (for set_params); the distinguished model of a two-population manager
    smcpp_host::ModelParams model_p1, model_p2;          // two populations: per-population pieces (set_params_twopop)
    std::vector<double> model_da1, model_da2;            // their derivative seeds [K x nder]
    double split = 0.0;
    std::vector<double> model_da;          // [Kp x nder] derivative seeds of a
    int nder = 0;
    std::vector<double> dpi, dT, dE;       // Jacobians [size x nder] of pi, T, E w.r.t. the seeds
    std::vector<double> emission, demission;   // InferenceManager::emission [M x cols] (+ Jacobian), model path only
    bool have_model = false;
    bool save_gamma = false, gamma_valid = false, estep_done = false;
    // ---- device -----------------------------------------------------------------------------------------------
    int device = 0;
    hipStream_t stream3 = nullptr;         // third branch of the statistics (per-key gamma sums)
    hipStream_t stream = nullptr, stream2 = nullptr;   // stream2: backward chain when it may overlap the forward one
    hipEvent_t ev[24];                                  // 10..13: forward / backward interval of the eigen-free pre-pass; 14: span-1 scalars done
    int dual_stream = 1;
    hipStream_t stream_hi = nullptr;
    bool chains_dual = false;
    DevBuf<RowInfo> d_rowinfo;
    DevBuf<int2> d_rowdesc;
    DevBuf<long long> d_dbg;
    DevBuf<float> d_qTf;
    DevBuf<double> d_qTdT, d_qPinvT, d_qPT, d_qPrm, d_qPinvrm;   // quarter-interleaved operands of the big-M chains
    int chain_mode = 2;   // dense fallback family: 2 CU-cooperative (one workgroup per chunk, chains2.hpp),
                          // 3 CU-cooperative with streamed operands (64 < M <= 256), 4 lock-step on the matrix cores (16 chunks per workgroup)
    int coop_bpc = 1;     // cooperative workgroups resident per CU the automatic chunking aims at
    // ---- chains on the semiseparable structure of T (chains_ss.hpp; chain_mode 5) ------------------------------------------
    std::unique_ptr<smcpp_host::OnePopPrep> prep1;   // one-population cold preparation (caches per-key tables)
    std::vector<double> prep1_hs;
    // device cold preparation (prep_dev.hpp): the emission table of the current parameters lives on the device only and the host
    // vectors E / dE / emission / Eg are stale until sync_host_E() fetches them (getters, non-lean E-steps)
    std::unique_ptr<DevPrep> dprep;
    bool E_on_dev = false, force_host_prep = false;
    void dev_prepare();
    void sync_host_E();
    // Q and its gradient on the device (prep_dev.hpp: k_q_reduce): the O(M) generators of the transition matrix with their
    // derivative planes (host, prep.hpp: transition_generators_jac; dT is expanded on the host only when its getter asks)
    smcpp_host::TransitionGenJac tgen;
    bool tgen_valid = false, dT_valid = true;
    // (round 6) the device-prepared model path keeps the O(M) generators of T and expands the M x M matrix only when somebody reads it
    // (ensure_T): the scan chains take their operator from the generators (ss_generators_from_tgen), so the expansion - 51 us at
    // M = 256 - and the entry-by-entry structure check of the expanded matrix - 58 us - leave the critical path in front of the chains;
    // host_prep_and_upload expands it for the statistics while the chains run
    bool T_lazy = false;
    smcpp_host::TransitionGenerators<double> tgen_g;
    void ensure_T() { if (T_lazy) { T = smcpp_host::transition_expand<double>(tgen_g); T_lazy = false; } }
    struct QDev {
        DevBuf<double> d_stats, d_out;
        DevBuf<int> d_keynb;
        PinnedArena stage;
        char *d_in = nullptr;
        size_t in_cap = 0;
        double *h_out = nullptr;
        size_t h_out_cap = 0;
        bool stats_ready = false;
        int Kq = 0;
        ~QDev() { if (d_in) (void)hipFree(d_in); if (h_out) (void)hipHostFree(h_out); }
    };
    std::unique_ptr<QDev> qdev;
    bool q_device(double val[4], double *jac);
    void ensure_dT();
    bool ss_static = false;                // the input qualifies (short spans); whether T does is decided on every E-step
    // hybrid scan chains (un-binned data): rows whose span exceeds ss_hyb_th take ONE eigen-power step inside the scan kernel
    // (chains_ss.hpp); they cost about SS_HYB_COST scan positions each, which is what the chunk list is balanced on
    bool ss_hybrid = false;
    bool ss_halo = false;                  // the first pass of the scan chains walks into every chunk from a halo (make_chunks)
    int ss_hyb_th = 0x7fffffff;
    static constexpr int SS_HYB_COST = 8;
    long long ss_row_cost(int span) const { return (ss_hybrid && span > ss_hyb_th) ? SS_HYB_COST : span; }
    bool ss_dirsplit = false;              // hybrid rows at M > 32: single-direction workgroups with two tables per eigen key (chains_ss.hpp)
    // eigen keys whose tables the hybrid rows keep in LDS (all of them unless they do not fit: then the most frequent ones, the
    // rest - COLD keys - read their table rows from L2; M > 32 with three or four eigen keys)
    int ss_nk_lds = 0, ss_ekey_of_slot[4] = {0, 1, 2, 3}, ss_eslot_of_key[4] = {0, 1, 2, 3};
    size_t ss_tab_bytes() const { return ss_hybrid ? ((size_t)ss_nk_lds * (ss_dirsplit ? 2 : 4) * Mp * (Mp + 1) + 8 * 64) * sizeof(double) : 0; }   // + one scratch vector per wavefront
    bool ss_active = false;                // this E-step's chains run on the scan kernels
    bool eigfree = false;                  // ... and its statistics need no eigensystem either (k_span_fold): no eigensolve at all
    int ss_max_span = 0;
    int ss_nlds = 0;                       // key slots whose emission vectors live in LDS
    int ss_wpc = 1;                        // scan chains: wavefronts per SIMD (workgroups per CU) the chunk list is cut for
    bool ss_mid = false;                   // one state per lane, 1.35 - 12 million positions: float halo instead of light passes, from 1.95 million on with two wavefronts per SIMD (make_chunks)
    int ss_wg_waves = 4;                   // wavefronts per workgroup of k_chain_ss (hybrid with two per SIMD: 8, one table copy)
    int ss_launched = 0, last_ss_passes = 0;
    bool ss_need_cert_pass = false;     // this input's last working pass rewrites end vectors within tolerance: launch the all-skip pass up front
    long long ss_positions = 0;            // sum of spans
    int ss_light_f = 0, ss_light_b = 0;    // light (float, store-free) passes per direction before the full fp64 pass
    // opt-in warm start of the scan chains (smcpp_set_warm_start): the first pass of an E-step starts every chunk from the boundary
    // vector the PREVIOUS converged E-step left (parity ss_warm_parity of the end-vector arrays) instead of pi / the uniform
    // vector, and one light pass fewer runs; pass indices then start at ss_pass0 (1 or 2: the parity the first pass reads)
    bool ss_warm_valid = false;
    // lean E-steps copy the (small) parameter arena on stream2 while the chains run; the statistics wait for ev[20]
    bool arena_side = false;
    int ss_warm_parity = 0, ss_pass0 = 0;
    std::vector<int> ss_slot_of_key;       // frequency rank of every key (slot 0 = most rows)
    DevBuf<int2> d_rowdesc_ss;             // [rows, padded] {key slot, span}
    // rows of binned data longer than 64 positions, cut into pieces at construction (build()): the caller's row counts, the pieces'
    // rows (what every kernel sees) and piece -> caller's row, per contig (1-based rows; entry 0 = row 0)
    bool split_spans = false;
    std::vector<int> user_Ls;
    std::vector<std::vector<int>> split_store, piece_row;
    std::vector<const int *> split_ptr;
    std::vector<DevBuf<int>> d_piece_first;       // per contig: first piece of every caller's row ([Lu + 2]; k_gamma_merge)
    DevBuf<double> d_gamma_user;                  // the caller's rows of ONE contig, pieces added up: [Lu + 1][Mp]
    const double *merged_gamma(int c);            // (device pointer; row 0 unset)
    // per-row posteriors of long rows at 64 < M <= 256 from eigen-power pieces (chains_ss.hpp: k_piece_vectors; engine_plans.hpp)
    std::vector<GPiece> gp_pieces;
    std::vector<GTile> gp_tiles;
    std::vector<int> gp_pfirst;
    bool gp_built = false, gamma_pieces_last = false;
    DevBuf<GPiece> d_gp_pieces;
    DevBuf<GTile> d_gp_tiles;
    DevBuf<int> d_gp_pfirst;
    DevBuf<float> d_gp_pvf;
    DevBuf<double> d_gp_pvb, d_gp_pgam, d_gp_cs, d_gp_l2d, d_gp_gen;
    long long gp_count = -1;
    long long gamma_piece_count();         // pieces of at most 64 positions the eigen rows fall into
    void build_gamma_pieces();
    bool ss_generators_only();             // the generators of T into ss_gen / ss_c0 (no underflow bound: the walks rescale every step)
    DevBuf<float> d_gpark;                 // k_gamma_rows_scan: [wavefronts][max span][64 NPL] parked forward vectors
    DevBuf<double> d_Fall;                 // [n_contigs Ke][smax][Mp][Mp] scratch of the span fold for M > 64 (k_span_big)
    SsArgs ss_args;
    std::vector<Chunk> chunks_b;           // backward chunks of the scan chains (more and shorter than the forward ones)
    std::vector<int> ss_tasks;             // (direction << 30 | chunk) per wavefront of a k_chain_ss launch
    DevBuf<Chunk> d_chunks_b;
    DevBuf<int> d_tasks;
    void update_pi_default();
    bool debug = false;                    // InferenceManager::debug (_smcpp.pxd:53): declared by the reference, read by nothing
    void upload_chunk_state();
    bool ss_extract_generators();          // generators of T (verified entry by entry) into ss_gen; false: T has no such structure
    std::vector<double> ss_gen;            // [10][MS]: f_dc f_g f_cg f_b f_a f_d b_dc b_g b_b b_a
    double ss_c0 = 0.0;
    void ss_launch_initial();
    void ss_launch_passes(int upto);
    void run_chains_ss();
    int hot_eig = -1, hot_eig2 = -1;
    // eigen-free pre-pass (chains2.hpp: k_group_powers, POWER instantiations): pass 0 runs on group powers while the host
    // solves the eigenproblems; only for short, few spans (binned data) and chunks short enough that pass 1 re-runs them whole
    bool power_ok = false, prepass_launched = false;
    int max_span_pw = 0, pw_nbits = 5, pw_npow = 4;
    PinnedArena pre_stage;                 // static operands of the pre-pass (pi, T, emission table): own pinned mirror
    char *d_pre = nullptr;
    size_t pre_cap = 0;
    bool static_packed = false;
    float pre_f_ms = 0.f, pre_b_ms = 0.f;
    DevBuf<float> d_Bf;                    // [Ke][4][Mp][Mp] binary powers A^2..A^16 per eigen key (forward operand)
    DevBuf<double> d_Bb;                   // [Ke][4][Mp][Mp] their transposes (backward operand)
    // pre-pass of the streamed-operand chains (64 < M <= 256): device-built layouts of T and of the powers A .. A^16
    DevBuf<double> d_W, d_pre_qTdT;        // [Ke][nbits][Mp][Mp] row-major powers (fp64) / [KQ][Mp][4]
    DevBuf<float> d_qBf, d_qBb, d_pre_qTf; // [Ke][nbits][KQ][Mp][4] float streaming layouts / [KQ][Mp][4]
    BigArgs pre_bargs;
    std::vector<std::unique_ptr<smcpp_host::EigTeam>> eig_teams;    // team-parallel eigensolver (M >= 128), one team per eigen key
    DevBuf<Chunk> d_chunks;
    DevBuf<Slab> d_slabs_sc, d_slabs_rk, d_slabs_eg;
    DevBuf<int2> d_perm1k;                 // span-1 rows sorted by key: {ell, key id} (one load resolves both)
    DevBuf<int> d_perm1, d_perme, d_gk_slab_off, d_s1_slab_off, d_eb_slab_off, d_eb_gid, d_ce_bucket_off,
        d_erow_slab, d_g_span, d_g_eig, d_e_kid, d_contig_L, d_changed_f, d_changed_b, d_argmax;
    DevBuf<long long> d_contig_base;
    DevBuf<float> d_pi_f, d_Tf, d_alpha, d_ends_f, d_used_f;
    DevBuf<double> d_E, d_dpow, d_PinvT, d_PT, d_TdT, d_Td, d_Prm, d_Pinvrm, d_dsc, d_dun, d_g_scale,
        d_g_logscale, d_beta, d_cnorm, d_logc, d_ends_b, d_used_b, d_llpart, d_loglik, d_w1, d_gpart, d_Xs, d_Ys,
        d_part_e, d_part_1, d_red_e, d_red_1, d_red_g, d_Z, d_Zpart, d_Y, d_xisum, d_gsum, d_gamma0, d_gamma_rows, d_Sq;
    // opt-in warm start: chunk-boundary vectors of the previous converged E-step (see smcpp_set_warm_start)
    bool warm_start = false, warm_valid = false;
    DevBuf<float> d_warm_f;
    DevBuf<double> d_warm_b;
    DevBuf<unsigned char> d_present;       // device copies used by k_pack_stats
    DevBuf<int> d_g2l;
    bool pack_tables_ready = false;
    std::vector<double> hs_PinvT, hs_PT, hs_Prm, hs_Pinvrm, hs_dsc, hs_dun, hs_dpow, hs_gsc, hs_gls, hs_TdT, hs_Td, hs_Ep;
    std::vector<float> hs_pi_f, hs_Tf;      // host staging of the per-E-step parameter arrays (see host_prep_and_upload)
    PinnedArena stage;
    char *d_param = nullptr;      // device side of the per-E-step parameter arena
    int *h_flags = nullptr;       // pinned: per-pass "something re-ran" flags of both chains, read back every round
    // scan-chain E-steps: the chain kernels write their flags and the log-likelihood kernel its result STRAIGHT into pinned host
    // memory (device views below) and a one-thread kernel at the end of the queue raises h_done; the host polls that word - no
    // small copies or fills on the stream, no blocking wait (together ~30 us of a 1.4 ms eval)
    int *d_flags_view = nullptr;  // device address of h_flags
    double *d_ll_view = nullptr;  // device address of h_ll
    int *h_done = nullptr, *d_done_view = nullptr;
    struct RcclDirect *rccl = nullptr;       // the E-step's exchange issued from here, on `stream` (smcpp_rccl_* below); owned
    int done_epoch = 0;
    DevBuf<unsigned> d_fin_ctr;          // blocks of the finalisation launches that raise h_done themselves (k_fin_both)
    unsigned fin_target = 0;
    int fold_done_epoch = 0;             // != 0: the statistics being enqueued end the queue and signal this epoch
    bool done_folded = false;
    bool timing_pending = false;         // the event intervals of the last E-step are read when somebody asks (resolve_timing)
    double t_host01 = 0, t_host12 = 0;
    void resolve_timing();
    bool done_covers_stats = false;
    bool wait_done(int epoch);
    double *h_ll = nullptr;       // pinned: per-contig log-likelihoods
    int h_flags_cap = 0, h_ll_cap = 0;
    size_t param_cap = 0;
    int llblk = 64;
    int ZS = 8;
    int ZG = 1;                          // shares of the per-key gamma-sum reduction of the one-pass span-1 form (a hot key holds most slabs)
    int max_pass = 0;
    int last_fwd_passes = 0, last_bwd_passes = 0;
    float eps_f = 2e-6f;
    double eps_b = 1e-6;   // relative; beta only enters products with the float alpha (noise floor 2e-6), see DESIGN.md §3
    // ---- results (host) ---------------------------------------------------------------------------------------
    std::vector<double> loglik, h_xisum, h_gsum, h_gamma0;
    bool stats_on_host = false;
    double timing[9] = {0};
    double host_timing[4] = {0};   // [cold preparation A6-A10, eigensystems, layouts + staging, whole host phase] of the last E-step, ms
    // multi-GPU
    std::vector<int> gkeys;                // global key list [Kg][keylen]
    std::vector<int> local_to_global;
    bool have_global = false;
    std::vector<double> g_stats;           // reduced [1 + M + M*M + Kg*M]
    bool have_reduced = false;
    // emission vectors (and Jacobians) of EVERY global key, so that Q on the all-reduced statistics also covers keys
    // that only other ranks' contigs hold; rows of keys nobody supplied (set_raw) are NaN
    std::vector<double> Eg, dEg;
    std::vector<int> raw_keys;             // what the last set_raw handed over: [Kr][keylen], raw_E [Kr][M]
    std::vector<double> raw_E;
    void global_emissions();

    ~smcpp_im() {
        if (stream) {
            for (auto &e : ev) (void)hipEventDestroy(e);
            (void)hipStreamDestroy(stream);
            if (stream2) (void)hipStreamDestroy(stream2);
            if (stream3) (void)hipStreamDestroy(stream3);
            if (stream_hi) (void)hipStreamDestroy(stream_hi);
        }
        if (d_param) (void)hipFree(d_param);
        if (d_pre) (void)hipFree(d_pre);
        if (h_flags) (void)hipHostFree(h_flags);
        if (h_ll) (void)hipHostFree(h_ll);
        if (h_done) (void)hipHostFree(h_done);
    }

    void build(int npop_, const int *nn, const int *nna, int n_contigs_, const int *Ls_, const int *const *obs,
               int n_hs, const double *hs_, double pol, int dev);
    void make_chunks();
    void make_slabs();
    void alloc_device();
    void host_prep_and_upload();
    void stage_static_and_prepass();
    void setup_power();
    ChainArgs chain_args();
    void run_chains();
    void run_stats();            // = enqueue_stats() unless run_chains() already queued them, + finish_stats()
    void enqueue_stats();
    void finish_stats();
    bool stats_enqueued = false;
    void estep();
    void fetch_stats();
    void prepare_params();
};

// ---------------------------------------------------------------------------------------------------------------
// construction
// ---------------------------------------------------------------------------------------------------------------
void smcpp_im::build(int npop_, const int *nn, const int *nna, int n_contigs_, const int *Ls_,
                     const int *const *obs, int n_hs, const double *hs_, double pol, int dev) {
    npop = npop_;
    keylen = 3 * npop;
    for (int p = 0; p < npop; ++p) { n[p] = nn[p]; na[p] = nna[p]; }
    polarization_error = pol;
    if (n_contigs_ <= 0) throw std::runtime_error("Observations list is empty");
    if (n_hs < 2) throw std::runtime_error("need at least two hidden state boundaries");
    hs.assign(hs_, hs_ + n_hs);
    for (int i = 1; i < n_hs; ++i)
        if (!(hs[i] >= hs[i - 1])) throw std::runtime_error("Hidden states must be in ascending order");
    M = n_hs - 1;
    Mp = (M + 15) / 16 * 16;
    // states per lane of the one-wavefront-per-chunk kernels: 1 .. 4 up to M = 256; 256 < M <= 512 (round 5): eight - the scan chains
    // and the eigen-free statistics only (binned data, a transition matrix with the reference's structure, no save_gamma: what
    // `smc++ estimate` runs); the dense fallback kernels and the eigensystem statistics stop at 256
    // (round 6) 512 < M <= 1024: sixteen states per lane, the same restriction
    NPL = M > 512 ? 16 : M > 256 ? 8 : (M + 63) / 64;
    NT = Mp / 16;
    if (M > 1024) throw std::runtime_error("M > 1024 hidden states is not supported by this build");
    n_contigs = n_contigs_;
    Ls.assign(Ls_, Ls_ + n_contigs);
    user_Ls = Ls;
    // (round 6) Long rows of BINNED data cut into pieces of at most 64 positions.  A row of span s with key k is s identical positions;
    // the rows (s1, k), (s2, k) with s1 + s2 = s are the same positions: the same likelihood, the same xi sums and gamma sums, and
    // the row's posterior (hmm.cpp:113-121: the sum over its positions, normalised to s) is the sum of the pieces' posteriors.  The
    // eigen-free statistics walk a span in at most 64 steps (k_span_scan) and are all that exists beyond 256 states; up to 256 the
    // alternative for such rows is an eigensolve of every key's M x M operator on the host per E-step.  So for M > 64 the pieces are
    // what every kernel sees; the getters that report per row (gamma, its argmax, the column count) add the pieces up again.
    // Only where the pieces stay few: un-binned data (spans of 10^4 - 10^5 base pairs) keep their rows and the eigen-power steps.
    {
        const int ncol_ = 1 + keylen;
        long long rows = 0, pieces = 0;
        int maxspan = 0;
        for (int c = 0; c < n_contigs; ++c)
            for (int i = 0; i < Ls[c]; ++i) {
                const int sp = obs[c][(size_t)i * ncol_];
                if (sp <= 0) throw std::runtime_error("data are malformed: span <= 0");
                rows++; pieces += (sp + 63) / 64; maxspan = std::max(maxspan, sp);
            }
        const bool want = M > 64 && !opt().off(smcpp_opt::O_SPLIT_SPANS);      // (M <= 64: the one-state-per-lane chains have no un-floored store)
        // (beyond 256 states there is no other path: un-binned rows are cut as well - the chains then walk every position, which is
        // what a row costs there anyway - as long as the pieces' alpha / beta rows fit a third of a 288 GB device)
        const bool few = pieces <= 2 * rows + 1024;
        const bool must = (M > 256 || opt().i(smcpp_opt::O_SPLIT_SPANS, 1) == 2) && (double)pieces * (double)Mp * 12.0 < 96e9;   // (=2: test switch)
        split_spans = want && maxspan > 64 && (few || must);
        if (split_spans) {
            split_store.assign(n_contigs, std::vector<int>());
            split_ptr.assign(n_contigs, nullptr);
            piece_row.assign(n_contigs, std::vector<int>());
            for (int c = 0; c < n_contigs; ++c) {
                std::vector<int> &so = split_store[c];
                std::vector<int> &pr = piece_row[c];
                pr.push_back(0);                              // (internal row 0 = the caller's row 0: the initial distribution)
                for (int i = 0; i < user_Ls[c]; ++i) {
                    const int *r = obs[c] + (size_t)i * ncol_;
                    for (int left = r[0]; left > 0; left -= 64) {
                        so.push_back(std::min(left, 64));
                        so.insert(so.end(), r + 1, r + ncol_);
                        pr.push_back(i + 1);
                    }
                }
                Ls[c] = (int)(so.size() / ncol_);
                split_ptr[c] = so.data();
            }
            obs = split_ptr.data();
        }
    }
    contig_base.resize(n_contigs);
    total_rows = 0;
    for (int c = 0; c < n_contigs; ++c) {
        if (Ls[c] <= 0) throw std::runtime_error("empty contig");
        contig_base[c] = total_rows;
        total_rows += (long long)Ls[c] + 1;
    }
    const int ncol = 1 + keylen;
    // key dictionary (populate_emission_probs, inference_manager.cpp:190-211): distinct keys, lexicographic
    std::map<std::vector<int>, int> kmap;
    for (int c = 0; c < n_contigs; ++c) {
        const int *ob = obs[c];
        std::vector<int> prev;
        for (int i = 0; i < Ls[c]; ++i) {
            const int *r = ob + (size_t)i * ncol;
            if (r[0] <= 0) throw std::runtime_error("data are malformed: span <= 0");
            if (!prev.empty() && std::equal(prev.begin(), prev.end(), r + 1)) continue;
            prev.assign(r + 1, r + ncol);
            kmap.emplace(prev, 0);
        }
    }
    K = (int)kmap.size();
    keys.clear();
    {
        int id = 0;
        for (auto &kv : kmap) { kv.second = id++; keys.insert(keys.end(), kv.first.begin(), kv.first.end()); }
    }
    key_nbpos.assign(K, 0);
    for (int k = 0; k < K; ++k) {
        int nb = 0;
        for (int p = 0; p < npop; ++p) nb += keys[(size_t)k * keylen + 3 * p + 2];
        key_nbpos[k] = nb > 0;
    }
    // rows -> (kid, span); fill_targets (inference_manager.cpp:232-254): distinct (span > 1, key) pairs
    rowinfo.assign((size_t)total_rows, RowInfo{0, -1});
    std::vector<int> span_of((size_t)total_rows, 1);
    present.assign((size_t)n_contigs * K, 0);
    span_sum.assign((size_t)n_contigs * K, 0.0);
    std::map<std::pair<int, int>, int> gmap;   // (kid, span) -> gid
#pragma omp parallel for schedule(dynamic)
    for (int c = 0; c < n_contigs; ++c) {
        const int *ob = obs[c];
        std::vector<int> prev;
        int prev_id = -1;
        for (int i = 0; i < Ls[c]; ++i) {
            const int *r = ob + (size_t)i * ncol;
            int id;
            if (!prev.empty() && std::equal(prev.begin(), prev.end(), r + 1)) id = prev_id;
            else {
                prev.assign(r + 1, r + ncol);
                id = kmap.find(prev)->second;
                prev_id = id;
            }
            const size_t g = (size_t)contig_base[c] + i + 1;
            rowinfo[g].kid = id;
            span_of[g] = r[0];
            present[(size_t)c * K + id] = 1;
            span_sum[(size_t)c * K + id] += (double)r[0];
        }
    }
    for (int c = 0; c < n_contigs; ++c)
        for (int i = 1; i <= Ls[c]; ++i) {
            const size_t g = (size_t)contig_base[c] + i;
            if (span_of[g] > 1) gmap.emplace(std::make_pair(rowinfo[g].kid, span_of[g]), 0);
        }
    G = (int)gmap.size();
    if (G >= (1 << 20)) throw std::runtime_error("too many distinct (span, key) pairs");
    groups.clear();
    eig_kid.clear();
    eig_of_key.assign(K, -1);
    {
        int id = 0;
        for (auto &kv : gmap) {
            kv.second = id++;
            const int kid = kv.first.first;
            if (eig_of_key[kid] < 0) { eig_of_key[kid] = (int)eig_kid.size(); eig_kid.push_back(kid); }
            groups.push_back(Group{kv.first.second, kid, eig_of_key[kid]});
        }
    }
    Ke = (int)eig_kid.size();
    for (int c = 0; c < n_contigs; ++c)
        for (int i = 1; i <= Ls[c]; ++i) {
            const size_t g = (size_t)contig_base[c] + i;
            if (span_of[g] > 1) rowinfo[g].gid = gmap[std::make_pair(rowinfo[g].kid, span_of[g])];
        }
    // device
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw std::runtime_error("no HIP device available: the SMC++ MI355X engine has no CPU fallback");
    if (dev >= 0) HIPCHK(hipSetDevice(dev));
    HIPCHK(hipGetDevice(&device));
    HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&stream3, hipStreamNonBlocking));
    {
        // the eigen-free statistics end in a serial fold on a few CUs: its branch gets a stream of the highest priority so that its
        // workgroups are placed ahead of the chip-filling rank updates they run beside
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(&stream_hi, hipStreamNonBlocking, greatest));
    }
    if (opt().has(smcpp_opt::O_DUAL_STREAM)) dual_stream = opt().i(smcpp_opt::O_DUAL_STREAM, 1);
    {
        // The events order streams of this device and bracket intervals; nothing the host reads depends on a system-scope fence at
        // an event (results reach it through stream-ordered copies and the pinned completion word of k_signal / k_fin_both).  A
        // default event makes every record a cache writeback + invalidation (hip_runtime_api.h: hipEventDisableSystemFence); an E-step
        // records about ten: headline 1 068 -> 1 082 evals/s, M = 256 2.79 -> 2.74 ms without it.  SMCPP_EVENT_FLAGS=0: default events.
        const unsigned evf = (unsigned)opt().ll(smcpp_opt::O_EVENT_FLAGS, (long long)hipEventDisableSystemFence);
        for (auto &e : ev) HIPCHK(hipEventCreateWithFlags(&e, evf));
    }
    make_chunks();
    make_slabs();
    alloc_device();
    update_pi_default();
    // defaults after construction (_smcpp.pyx:318-320)
    alpha = 1.0; theta = 1e-4; rho = 1e-4;
    loglik.assign(n_contigs, 0.0);
}

// rows per (CU x 16) from which the lock-step chains win (tools/lock_crossover.py on the whole-genome generator: M = 64 and 48
// from ~400, M = 32 from ~700; at M = 16 the cooperative kernels are never slower)
static long long lock_min_rows(int Mp) {
    const long long v = opt().ll(smcpp_opt::O_LOCK_MIN_ROWS, -1);
    if (v >= 0) return v;
    return Mp >= 48 ? 450 : Mp >= 32 ? 800 : (1ll << 40);
}

// Chunks per contig of the scan chains for `nslots` wavefront slots (cost = positions, or cost units with hybrid rows): start from
// the rounded-down share of every contig and hand the remaining slots, one at a time, to the contig whose chunks are currently
// the longest - never more chunks than slots (unless there are more contigs than slots), and the longest chunk is as short as the
// slot count allows.  Host-only; exported as smcpp_host_chunk_counts for the CPU tests.
static std::vector<int> ss_chunk_counts(const std::vector<long long> &cpos, const std::vector<int> &rows, long long nslots,
                                        long long floor_cost) {
    const int n = (int)cpos.size();
    long long total = 0;
    for (long long c : cpos) total += c;
    const long long bpc = std::max<long long>(std::max<long long>(1, floor_cost), (total + nslots - 1) / std::max<long long>(1, nslots));
    std::vector<int> ncs(n, 1);
    long long used = 0;
    for (int c = 0; c < n; ++c) {
        ncs[c] = (int)std::max<long long>(1, std::min<long long>(rows[c], cpos[c] / bpc));
        used += ncs[c];
    }
    const long long want = std::max<long long>(n, std::min<long long>(nslots, (total + bpc - 1) / bpc));
    while (used < want) {
        int best = -1;
        double bl = 0.0;
        for (int c = 0; c < n; ++c) {
            if (ncs[c] >= rows[c]) continue;
            if (cpos[c] < (long long)(ncs[c] + 1) * std::max<long long>(1, floor_cost)) continue;     // no chunk below the floor
            const double len = (double)cpos[c] / ncs[c];
            if (best < 0 || len > bl) { best = c; bl = len; }
        }
        if (best < 0) break;
        ++ncs[best];
        ++used;
    }
    return ncs;
}

void smcpp_im::make_chunks() {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    {
        // SMCPP_CHAIN = lock: the lock-step kernels forced (M <= 64); = dense: the cooperative kernels; ss (or unset): see below
        const char *m = opt().has(smcpp_opt::O_CHAIN) ? opt().str(smcpp_opt::O_CHAIN).c_str() : nullptr;
        if (m) chain_mode = (!strcmp(m, "lock") && Mp <= 64) ? 4 : 2;
        // lock-step chains on the matrix cores (chains_lock.hpp): 16 chunks per workgroup, so 16 x more and 16 x shorter
        // chunks - they pay off when those are still long against the ~900 rows of history every chunk re-runs
        if (!m && Mp <= 64 && (total_rows - n_contigs) / ((long long)prop.multiProcessorCount * LOCK_NC) >= lock_min_rows(Mp)) {
            // ... and when one eigen key dominates the span > 1 rows (binned data: the monomorphic key): only its operators are
            // register-resident there, every other key present in a step costs two L2 round trips for the whole workgroup
            std::vector<long long> cnt(std::max(1, Ke), 0);
            long long ne = 0;
            for (const RowInfo &ri : rowinfo)
                if (ri.gid >= 0) { ++cnt[groups[ri.gid].eig]; ++ne; }
            const long long top = *std::max_element(cnt.begin(), cnt.end());
            if (ne == 0 || 10 * top >= 9 * ne) chain_mode = 4;
        }
        // 64 < M <= 256: the streaming cooperative kernels (k_fwd_big / k_bwd_big)
        if (Mp > 64) chain_mode = 3;
        // Chains on the semiseparable structure of T (chains_ss.hpp): one position per step, so the input qualifies when
        // its spans are short (binned data; un-binned posterior data with spans of 10^4 .. 10^5 keep the eigen kernels).
        // chain_mode then names the DENSE kernels an E-step falls back to when its T has no such structure.
        ss_max_span = 1;
        for (const Group &g : groups) ss_max_span = std::max(ss_max_span, g.span);
        {
            const bool ss_ok = !opt().off(smcpp_opt::O_SS) && (!m || !strcmp(m, "ss")) && Mp <= 1024;
            ss_static = ss_ok && ss_max_span <= 512;
            ss_hybrid = false; ss_hyb_th = 0x7fffffff;
            if (ss_ok && !ss_static) {
                // longer spans: the hybrid form, when one state per lane holds the vector and the eigenvector tables of every eigen
                // key fit LDS beside the emission vectors (SMCPP_HYBRID=0: the dense kernels)
                const bool hy_off = opt().off(smcpp_opt::O_HYBRID);
                const int hyb_th_opt = std::max(1, opt().i(smcpp_opt::O_HYB_TH, 6));
                const size_t tab = (size_t)Ke * 4 * Mp * (Mp + 1) * sizeof(double);
                ss_dirsplit = false;
                if (!hy_off && Mp <= 64 && Ke >= 1 && Ke <= 4 && tab <= 120 * 1024) {
                    ss_static = ss_hybrid = true;
                    ss_hyb_th = hyb_th_opt;
                } else if (!hy_off && Mp <= 64 && Ke >= 1 && Ke <= 4 && tab / 2 <= 136 * 1024) {
                    // (round 4) M > 32: four 33 KB tables per eigen key do not fit, the two a DIRECTION needs do - every workgroup
                    // runs one direction (the task table keeps them apart) and stages that direction's pair
                    ss_static = ss_hybrid = ss_dirsplit = true;
                    ss_hyb_th = hyb_th_opt;
                } else if (!hy_off && Mp > 32 && Mp <= 64 && Ke >= 1 && Ke <= 4 && tab / 2 / Ke <= 136 * 1024) {
                    // (round 5) ... and with three or four eigen keys at M > 32 not even those: the most frequent keys keep their pair
                    // in LDS, a COLD key's table rows are read from L2 on the rows that need them (chains_ss.hpp: ss_eig_matvec_cold)
                    ss_static = ss_hybrid = ss_dirsplit = true;
                    ss_hyb_th = hyb_th_opt;
                }
                ss_nk_lds = Ke;
                for (int e = 0; e < 4; ++e) ss_ekey_of_slot[e] = ss_eslot_of_key[e] = e;
                if (ss_hybrid && ss_dirsplit && tab / 2 > 136 * 1024) {
                    // slots by frequency of the keys' hybrid rows
                    std::vector<long long> cnt(Ke, 0);
                    for (const RowInfo &ri : rowinfo)
                        if (ri.gid >= 0 && groups[ri.gid].span > ss_hyb_th) ++cnt[groups[ri.gid].eig];
                    std::vector<int> order(Ke);
                    for (int e = 0; e < Ke; ++e) order[e] = e;
                    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cnt[x] > cnt[y]; });
                    ss_nk_lds = (int)std::max<size_t>(1, std::min<size_t>((size_t)Ke, (size_t)(136 * 1024) / (tab / 2 / Ke)));
                    for (int e = 0; e < 4; ++e) ss_eslot_of_key[e] = -1;
                    for (int sl = 0; sl < ss_nk_lds; ++sl) { ss_ekey_of_slot[sl] = order[sl]; ss_eslot_of_key[order[sl]] = sl; }
                }
            }
            if (ss_static) chain_mode = Mp > 64 ? 3 : 2;
        }
        if (opt().i(smcpp_opt::O_COOP_BPC, 0) > 0) coop_bpc = opt().i(smcpp_opt::O_COOP_BPC, 0);
        else {
            // More workgroups per CU hide the per-row latency of the cooperative kernels (measured on 6.8 M rows:
            // throughput x1.27 / x1.36 / x1.42 for 2 / 3 / 4 per CU) but shorten the chunks, and every chunk pays
            // ~1100 rows of re-run history; the break-even points below follow from those two numbers.
            const long long per_cu = (total_rows - n_contigs) / std::max(1, prop.multiProcessorCount);
            coop_bpc = per_cu < 3000 ? 1 : per_cu < 9000 ? 2 : per_cu < 17000 ? 3 : 4;
        }
    }
    // chunks in flight: one per SIMD for the per-wavefront kernels, coop_bpc per CU for the cooperative ones
    long long slots = (long long)prop.multiProcessorCount *
                      (chain_mode == 4 ? LOCK_NC : chain_mode == 3 ? 1 : coop_bpc);
    if (ss_static && user_rows_per_chunk <= 0 && !opt().has(smcpp_opt::O_ROWS_PER_CHUNK)) {
        // scan chains: one wavefront per chunk and direction, a workgroup = 2 forward + 2 backward chunks = one wavefront per
        // SIMD; chunks are cut by POSITIONS (sum of spans), the unit of work of these kernels.  Every chunk pays the same
        // ~3000 positions of re-run history however short it is (the chains forget with an e-fold of ~240 positions).
        std::vector<long long> cum;
        long long total_bins = 0;
        for (int c = 0; c < n_contigs; ++c)
            for (int i = 1; i <= Ls[c]; ++i) {
                const RowInfo &ri = rowinfo[(size_t)contig_base[c] + i];
                total_bins += ri.gid < 0 ? 1 : ss_row_cost(groups[ri.gid].span);
            }
        // Wavefronts per SIMD: one wavefront leaves a quarter of the issue slots empty (an instruction occupies the SIMD for 4 of
        // the ~5.3 clocks between two issues of one wavefront), a second and third fill them - but every chunk pays ~3 000 positions
        // of re-run history, so only inputs whose chunks stay long (>= 9 000 positions) take them.  Whole genome (28.7 M
        // positions): 9.3 / 8.2 / 7.9 ms of chains with 1 / 2 / 3; a 3.4 M-position shard: 1.83 / 1.95 ms with 1 / 2.
        const long long simds = (long long)prop.multiProcessorCount * 4;
        // (round 6, last session) one state per lane, re-measured on single contigs of 150 ... 1 700 Mbp and a rank's shard of the genome
        // (gpurun_out/r06_gp12 ... gp17): between 1.95 and 12 million positions TWO wavefronts per SIMD that enter their chunks through a
        // float halo (no light passes, no fp64 part of the halo) beat one wavefront with light passes by 10 - 19 % (250 Mbp 1.66 -> 1.44 ms,
        // 700 Mbp 2.94 -> 2.46, 1 100 Mbp 4.15 -> 3.37, the 8-GPU run's shard 1.75 -> 1.57), from 12 million on three wavefronts without a
        // halo win (1 700 Mbp 6.01 -> 4.59; whole genome unchanged); below 1.35 million (the headline: 1 million) nothing changes.
        ss_mid = false;
        int wpc_auto = (int)std::max<long long>(1, std::min<long long>(3, total_bins / (simds * 9000)));
        if (NPL == 1 && !ss_hybrid) {
            const long long per_simd = total_bins / std::max<long long>(1, simds);
            // (... and between 1.35 and 1.95 million the halo with ONE wavefront: 135 / 150 / 175 Mbp 1.19 / 1.34 / 1.52 -> 1.12 / 1.18 / 1.27 ms;
            // at 200 Mbp the light passes hit a sweet spot - 1 + 1 of them, three launches, 1.26 ms - that the halo forms miss by 3 %:
            // the rule stays monotone, gpurun_out/r06_gp17)
            wpc_auto = per_simd >= 11700 ? 3 : per_simd >= 1900 ? 2 : 1;
            ss_mid = (wpc_auto == 2 || per_simd >= 1300) && wpc_auto < 3 && !opt().has(smcpp_opt::O_SS_WPC);
        }
        ss_wpc = opt().has(smcpp_opt::O_SS_WPC) ? std::max(1, std::min(4, opt().i(smcpp_opt::O_SS_WPC, 1))) : wpc_auto;
        // hybrid rows are bound by instruction and LDS LATENCY (a dependent chain of ~200 instructions per row): a second wavefront
        // per SIMD fills the gaps from ~2 000 cost units per chunk on (posterior workload: 1.79 -> 1.35 ms of chains; a third one
        // needs an extra pass: 1.80); the eight wavefronts form ONE workgroup so that the CU holds one copy of the tables
        if (ss_hybrid && !opt().has(smcpp_opt::O_SS_WPC)) ss_wpc = (int)std::max<long long>(1, std::min<long long>(2, total_bins / (simds * 2000)));
        if (ss_hybrid) ss_wpc = std::min(ss_wpc, 2);
        ss_wg_waves = (ss_hybrid && ss_wpc == 2) ? 8 : 4;
        const long long waves = simds * ss_wpc;
        max_chunks_per_contig = 1;
        // positions per contig
        std::vector<long long> cpos(n_contigs, 0);
        for (int c = 0; c < n_contigs; ++c)
            for (int i = 1; i <= Ls[c]; ++i) {
                const RowInfo &ri = rowinfo[(size_t)contig_base[c] + i];
                cpos[c] += ri.gid < 0 ? 1 : ss_row_cost(groups[ri.gid].span);
            }
        auto cut = [&](long long nslots, long long floor_bins, std::vector<Chunk> &out, bool bwd, long long halo_l, long long halo_d) {
            // Chunks per contig: NEVER more chunks than wavefront slots in total (a launch of 1046 wavefronts on 1024 SIMDs puts two
            // on some of them, and the kernel then lasts as long as those take: whole genome, 22 contigs each rounded up, +27 %).
            // Start from the rounded-down share of every contig and hand the remaining slots, one at a time, to the contig whose
            // chunks are currently the longest (minimises the longest chunk).
            const std::vector<int> ncs = ss_chunk_counts(cpos, Ls, nslots, floor_bins);
            out.clear();
            for (int c = 0; c < n_contigs; ++c) {
                const int L = Ls[c];
                cum.assign((size_t)L + 1, 0);
                for (int i = 1; i <= L; ++i) {
                    const RowInfo &ri = rowinfo[(size_t)contig_base[c] + i];
                    cum[i] = cum[i - 1] + (ri.gid < 0 ? 1 : ss_row_cost(groups[ri.gid].span));
                }
                const int nc = ncs[c];
                max_chunks_per_contig = std::max(max_chunks_per_contig, nc);
                int prev = 0;
                for (int j = 0; j < nc; ++j) {
                    int r1;
                    if (j == nc - 1) r1 = L;
                    else {
                        const long long target = cum[L] * (j + 1) / nc;
                        r1 = (int)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
                        r1 = std::max(prev + 1, std::min(r1, L - (nc - 1 - j)));
                    }
                    Chunk ch;
                    ch.base = contig_base[c];
                    ch.r0 = prev; ch.r1 = r1; ch.contig = c;
                    ch.first = (j == 0); ch.last = (j == nc - 1); ch.pad = 0;
                    // halo rows (positions counted on this contig's cumulative costs): forward chunks look back, backward ones ahead
                    if (bwd) {
                        const long long e1 = cum[r1] + halo_d, e0 = e1 + halo_l;
                        ch.h1 = (int)std::min<long long>(L, std::lower_bound(cum.begin(), cum.end(), e1) - cum.begin());
                        ch.h0 = (int)std::min<long long>(L, std::lower_bound(cum.begin(), cum.end(), e0) - cum.begin());
                        if (halo_l + halo_d == 0 || ch.last) ch.h0 = ch.h1 = r1;
                    } else {
                        const long long e1 = cum[prev] - halo_d, e0 = e1 - halo_l;
                        ch.h1 = e1 <= 0 ? 0 : (int)(std::upper_bound(cum.begin(), cum.end(), e1) - cum.begin()) - 1;
                        ch.h0 = e0 <= 0 ? 0 : (int)(std::upper_bound(cum.begin(), cum.end(), e0) - cum.begin()) - 1;
                        ch.h1 = std::min(ch.h1, prev); ch.h0 = std::min(ch.h0, ch.h1);
                        if (halo_l + halo_d == 0 || ch.first) ch.h0 = ch.h1 = prev;
                    }
                    out.push_back(ch);
                    prev = r1;
                }
            }
        };
        // Halo pass (chains_ss.hpp): with several chunks per contig every wavefront first walks into its chunk from its neighbour's
        // rows - `light` positions in float, then `dbl` in fp64, neither stored - so that ONE launch leaves rows that are already
        // exact to the certificate's tolerance (the chains forget with an e-fold of ~240 positions forward, ~340 backward: 11.5 +
        // 3.3 e-folds), instead of two store-free light passes over the WHOLE chunk, a full pass and a merge re-run.
        // Measured (profiles/r04_e_halo_probe.log): on one 100 Mbp contig at M <= 64 the halo is as long as two chunks - the same
        // history the two light passes walk - so it only trades the merge re-run against chunks of unequal length: 0.92 ms against
        // 0.87; with several states per lane (M > 64), where a light position costs relatively more, it wins (c5: 1.45 against 1.64 ms).
        // Default: M > 64 only; SMCPP_SS_HALO = 1 / 0 forces it.
        // (the mid-size regime of one state per lane, above: halo on, float part only)
        ss_halo = !ss_hybrid && (opt().has(smcpp_opt::O_SS_HALO) ? opt().i(smcpp_opt::O_SS_HALO, 0) != 0 : (NPL >= 2 || ss_mid));
        const bool mid_halo = ss_halo && ss_mid && NPL == 1;
        const long long hlf = ss_halo ? opt().ll(smcpp_opt::O_HALO_LF, 2800) : 0, hdf = ss_halo ? opt().ll(smcpp_opt::O_HALO_DF, mid_halo ? 0 : 800) : 0,
                        hlb = ss_halo ? opt().ll(smcpp_opt::O_HALO_LB, 3900) : 0, hdb = ss_halo ? opt().ll(smcpp_opt::O_HALO_DB, mid_halo ? 0 : 1100) : 0;
        {
            // the forward chain gets SMCPP_SS_FWD_SHARE of the wavefronts.  Default one half: the backward chain's light position
            // costs 38 instructions against 25, but the fp64 passes of the two directions take the same time (forward: stores, the
            // reciprocal and the feedback of the stored vector per row), and measured on the headline 0.45 / 0.42 / 0.38 / 0.34 lose
            // 6 / 12 / 28 / 37 % of chain time against 0.5
            // (halo pass: the backward wavefronts carry the longer halo and the dearer position, so they get more, shorter chunks:
            // per wavefront halo_f + 53 P / c_f = halo_b + 59 P / c_b instructions with c_f + c_b = waves)
            double dflt_share = 0.5;
            if (ss_halo && total_bins > 0) {
                const double Hf = 25.0 * hlf + 53.0 * hdf, Hb = 29.0 * hlb + 59.0 * hdb, P = (double)total_bins, W = (double)waves;
                double lo = 0.05, hi = 0.95;
                for (int it = 0; it < 40; ++it) {
                    const double m = 0.5 * (lo + hi);
                    const double f = Hf + 53.0 * P / (m * W), b = Hb + 59.0 * P / ((1.0 - m) * W);
                    if (f > b) lo = m; else hi = m;
                }
                dflt_share = std::min(0.5, std::max(0.25, 0.5 * (lo + hi)));
                // (one state per lane, float halo only: the stored passes of the two directions cost the same since round 5 and the task
                // table pairs them on the SIMDs one to one - 0.30 ... 0.58 measured on 175 ... 1 100 Mbp, gpurun_out/r06_gp22 / gp23: one half
                // is the optimum, 0 - 3 % ahead of the balance above, whose per-position costs are those of several states per lane)
                if (mid_halo) dflt_share = 0.5;
            }
            const double share = opt().d(smcpp_opt::O_SS_FWD_SHARE, dflt_share);
            const long long nf = std::max<long long>(1, (long long)(share * (double)waves + 0.5));
            cut(nf, 1024, chunks, false, hlf, hdf);
            cut(std::max<long long>(1, waves - nf), 1024, chunks_b, true, hlb, hdb);
        }
        max_pass = max_chunks_per_contig + 3 + 4 + 2;  // (+4: light passes, +2: a warm start numbers its passes from 1 or 2)
        return;
    }
    long long rows = total_rows - n_contigs;
    int lc = user_rows_per_chunk;
    if (lc <= 0) {
        lc = opt().i(smcpp_opt::O_ROWS_PER_CHUNK, lc);
    }
    // every chunk pays ~1000 rows of re-run history however short it is, so small inputs get few, long chunks rather
    // than one sliver per CU (a 1 500-row contig: 3 chunks and 4 passes instead of 24 chunks and 15 passes)
    if (lc <= 0) lc = (int)std::max<long long>(512, (rows + slots - 1) / slots);
    chunks.clear();
    max_chunks_per_contig = 1;
    for (int c = 0; c < n_contigs; ++c) {
        const int L = Ls[c];
        const int nc = std::max(1, ceil_div(L, lc));
        max_chunks_per_contig = std::max(max_chunks_per_contig, nc);
        for (int j = 0; j < nc; ++j) {
            Chunk ch;
            ch.base = contig_base[c];
            ch.r0 = (int)((long long)L * j / nc);
            ch.r1 = (int)((long long)L * (j + 1) / nc);
            ch.contig = c;
            ch.first = (j == 0);
            ch.last = (j == nc - 1);
            ch.pad = 0;
            ch.h0 = ch.h1 = ch.r0;               // (forward list; the backward copy below is given r1: no halo on this path)
            chunks.push_back(ch);
        }
    }
    max_pass = max_chunks_per_contig + 3;   // (+1: the full pass that follows an eigen-free pre-pass)
    if (ss_static) max_pass += 4 + 2;       // light passes of the scan chains; a warm start numbers its passes from 1 or 2
    chunks_b = chunks;
    for (Chunk &cb : chunks_b) cb.h0 = cb.h1 = cb.r1;
    ss_halo = false;
}

void smcpp_im::upload_chunk_state() {
    ss_warm_valid = false;
    const size_t nch = std::max(chunks.size(), chunks_b.size());
    d_chunks.upload(chunks, stream);
    d_chunks_b.upload(chunks_b, stream);
    {
        // wavefront -> (direction, chunk) of the one-chain-per-wavefront launches: the two directions interleaved in proportion, so
        // that every workgroup (4 wavefronts = the 4 SIMDs of a CU) holds its share of both
        const size_t nf = chunks.size(), nb = chunks_b.size();
        ss_tasks.clear();
        size_t i = 0, j = 0;
        if (ss_hybrid && ss_dirsplit) {
            // single-direction workgroups, the two kinds interleaved in proportion
            const size_t W = (size_t)ss_wg_waves;
            while (i < nf || j < nb) {
                const bool take_f = j >= nb || (i < nf && (double)i * (double)nb <= (double)j * (double)nf);
                for (size_t q = 0; q < W; ++q) {
                    if (take_f) ss_tasks.push_back(i < nf ? (int)i++ : -1);
                    else ss_tasks.push_back(j < nb ? ((1 << 30) | (int)j++) : -1);
                }
            }
        } else
        while (i < nf || j < nb) {
            // next task: the direction that is behind its proportional share
            const bool take_f = j >= nb || (i < nf && (double)i * (double)nb <= (double)j * (double)nf);
            if (take_f) ss_tasks.push_back((int)i++);
            else ss_tasks.push_back((1 << 30) | (int)j++);
        }
        while (ss_tasks.size() % ss_wg_waves) ss_tasks.push_back(-1);
        d_tasks.upload(ss_tasks, stream);
    }
    d_ends_f.alloc(2 * nch * Mp); d_used_f.alloc(nch * Mp);
    d_ends_b.alloc(2 * nch * Mp); d_used_b.alloc(nch * Mp);
    d_changed_f.alloc(max_pass + 1); d_changed_b.alloc(max_pass + 1);
    HIPCHK(hipStreamSynchronize(stream));
}

// pi of defaultEta (a = s = {1}: R(t) = t), inference_manager.cpp:12-19,43,56-69: what a fresh HMM's statistics hold; follows
// the hidden states (smcpp_set_hidden_states before the first E-step)
void smcpp_im::update_pi_default() {
    pi_default.assign(M, 0.0);
    double sm = 0.0;
    for (int m = 0; m < M; ++m) {
        double v = std::exp(-hs[m]) - ((m + 1 < M) ? std::exp(-hs[m + 1]) : 0.0);
        if (v < 1e-20) v = 1e-20;
        pi_default[m] = v;
        sm += v;
    }
    for (double &v : pi_default) v /= sm;
}

static bool stats_team_on() { return !opt().off(smcpp_opt::O_STATS_TEAM); }

void smcpp_im::make_slabs() {
    // counting sorts of rows per contig
    perm1.clear(); perme.clear(); perm1k.clear();
    slabs_sc.clear(); slabs_rk.clear(); slabs_eg.clear();
    gk_slab_off.assign((size_t)n_contigs * K + 1, 0);
    s1_slab_off.assign(n_contigs + 1, 0);
    ce_bucket_off.assign((size_t)n_contigs * Ke + 1, 0);
    ce_row_off.assign((size_t)n_contigs * Ke + 1, 0);
    eb_slab_off.clear(); eb_gid.clear(); erow_slab.clear();
    int last_eig_key = -1;
    long long n1 = 0, ne = 0;
    // (rows with ell = 0 have kid = 0, gid = -1 and are skipped below)
    for (int c = 0; c < n_contigs; ++c)
        for (int i = 1; i <= Ls[c]; ++i) (rowinfo[(size_t)contig_base[c] + i].gid < 0 ? n1 : ne)++;
    n_1_rows = n1; n_e_rows = ne;
    const long long part_bytes = (long long)Mp * Mp * 8;
    // slabs = independent single-wavefront work items; several thousand keep the 2048 resident wavefronts of the
    // chip balanced on large inputs (each slab owns an Mp x Mp partial: at most 256 MB of them)
    // (round 5: teams of four slabs share one partial - k_rank_acc<., true> -, so four times as many slabs fit the same 256 MB)
    const long long target = std::max<long long>(256, std::min<long long>(8192, ((stats_team_on() ? 1024ll : 256ll) << 20) / part_bytes));
    // (at least SMCPP_SLAB_ROWS rows per slab, default 128: every slab costs an Mp x Mp partial written and read back - 128 MB of
    // traffic per headline E-step with 64-row slabs -, but a slab is walked by ONE wavefront, and below ~1000 slabs the rank
    // kernels leave SIMDs idle: 64 .. 192 rows measured: 633 / 641 / 666 / 665 headline evals per second)
    const int slab_rows = opt().has(smcpp_opt::O_SLAB_ROWS) ? std::max(16, opt().i(smcpp_opt::O_SLAB_ROWS, 128)) : 128;
    int S_RK = (int)std::max<long long>(slab_rows, (n1 + target - 1) / target);
    S_RK = (S_RK + 3) / 4 * 4;
    int S_EG = (int)std::max<long long>(slab_rows, (ne + target - 1) / target);
    S_EG = (S_EG + 15) / 16 * 16;
    const int S_SC = 256;
    for (int c = 0; c < n_contigs; ++c) {
        const long long base = contig_base[c];
        // ---- span-1 rows sorted by key ----
        std::vector<std::vector<int>> by_key(K);
        std::vector<std::vector<int>> by_grp(G);
        for (int i = 1; i <= Ls[c]; ++i) {
            const RowInfo &ri = rowinfo[(size_t)base + i];
            if (ri.gid < 0) by_key[ri.kid].push_back(i);
            else by_grp[ri.gid].push_back(i);
        }
        const int seg_start = (int)perm1.size();
        for (int k = 0; k < K; ++k) {
            gk_slab_off[(size_t)c * K + k] = (int)slabs_sc.size();
            const int s0 = (int)perm1.size();
            perm1.insert(perm1.end(), by_key[k].begin(), by_key[k].end());
            for (int ell : by_key[k]) perm1k.push_back(make_int2(ell, k));
            const int s1 = (int)perm1.size();
            for (int s = s0; s < s1; s += S_SC)
                slabs_sc.push_back(Slab{s, std::min(s + S_SC, s1), c * K + k, k, base});
        }
        const int seg_end = (int)perm1.size();
        // the rank update does not need key-homogeneous slabs (the key only selects an L2-resident emission vector):
        // its copy of the permutation runs in natural row order, so every slab streams through alpha / beta
        std::sort(perm1k.begin() + seg_start, perm1k.begin() + seg_end,
                  [](const int2 &x, const int2 &y) { return x.x < y.x; });
        s1_slab_off[c] = (int)slabs_rk.size();
        for (int s = seg_start; s < seg_end; s += S_RK)
            slabs_rk.push_back(Slab{s, std::min(s + S_RK, seg_end), c, -1, base});
        // ---- eigen rows sorted by (eigen key, group) ----
        for (int e = 0; e < Ke; ++e) {
            ce_bucket_off[(size_t)c * Ke + e] = (int)eb_gid.size();
            ce_row_off[(size_t)c * Ke + e] = (int)perme.size();
            for (int g = 0; g < G; ++g) {
                if (groups[g].eig != e || by_grp[g].empty()) continue;
                // the fused eigen kernel shares one LDS copy of (Pinv, P) among the 4 slabs of a workgroup: pad with
                // empty slabs (they add zero partials to the previous bucket) so that no workgroup mixes eigen keys
                if (!slabs_eg.empty() && last_eig_key != e) {
                    while (slabs_eg.size() % 4 != 0) {
                        Slab pad = slabs_eg.back();
                        pad.start = pad.end;
                        slabs_eg.push_back(pad);
                    }
                }
                last_eig_key = e;
                eb_slab_off.push_back((int)slabs_eg.size());
                eb_gid.push_back(g);
                const int s0 = (int)perme.size();
                perme.insert(perme.end(), by_grp[g].begin(), by_grp[g].end());
                const int s1 = (int)perme.size();
                for (int s = s0; s < s1; s += S_EG) {
                    const int se = std::min(s + S_EG, s1);
                    for (int r = s; r < se; ++r) erow_slab.push_back((int)slabs_eg.size());
                    slabs_eg.push_back(Slab{s, se, (int)eb_gid.size() - 1, g, base});
                }
            }
        }
    }
    gk_slab_off[(size_t)n_contigs * K] = (int)slabs_sc.size();
    s1_slab_off[n_contigs] = (int)slabs_rk.size();
    ce_bucket_off[(size_t)n_contigs * Ke] = (int)eb_gid.size();
    ce_row_off[(size_t)n_contigs * Ke] = (int)perme.size();
    eb_slab_off.push_back((int)slabs_eg.size());
    // single-key span-1 slabs in key-sorted order (k_rank_acc<3>)
    slabs_fk.clear();
    fk_c_off.assign(n_contigs + 1, 0);
    fk_gk_off.assign((size_t)n_contigs * K + 1, 0);
    for (int c = 0; c < n_contigs; ++c) {
        fk_c_off[c] = (int)slabs_fk.size();
        for (int k = 0; k < K; ++k) {
            fk_gk_off[(size_t)c * K + k] = (int)slabs_fk.size();
            const int g0 = gk_slab_off[(size_t)c * K + k], g1 = gk_slab_off[(size_t)c * K + k + 1];
            if (g1 <= g0) continue;
            const int q0 = slabs_sc[g0].start, q1 = slabs_sc[g1 - 1].end;       // the (contig, key) segment of perm1
            for (int q = q0; q < q1; q += S_RK) slabs_fk.push_back(Slab{q, std::min(q + S_RK, q1), c, k, contig_base[c]});
        }
    }
    fk_c_off[n_contigs] = (int)slabs_fk.size();
    fk_gk_off[(size_t)n_contigs * K] = (int)slabs_fk.size();
    {
        auto cut = [](const std::vector<int> &range_off, std::vector<int2> &teams, std::vector<int> &team_off) {
            teams.clear();
            team_off.assign(range_off.size(), 0);
            for (size_t r = 0; r + 1 < range_off.size(); ++r) {
                team_off[r] = (int)teams.size();
                for (int q = range_off[r]; q < range_off[r + 1]; q += 4) teams.push_back(make_int2(q, std::min(4, range_off[r + 1] - q)));
            }
            if (!range_off.empty()) team_off.back() = (int)teams.size();
        };
        cut(fk_c_off, teams_fk, fk_c_team_off);
        cut(eb_slab_off, teams_eg, eb_team_off);
        cut(s1_slab_off, teams_rk, s1_team_off);
    }
    // generation-2 eigen slabs: the sorted eigen rows of every (contig, eigen key) cut into S_EG-row pieces regardless of the span
    // groups; padded like slabs_eg so that a workgroup of four never mixes keys
    slabs_ek.clear(); epos_gid.clear();
    ek_slab_off.assign((size_t)n_contigs * Ke + 1, 0);
    epos_gid.reserve(perme.size());
    for (size_t q = 0; q < erow_slab.size(); ++q) epos_gid.push_back(slabs_eg[erow_slab[q]].aux);
    erow_desc.clear();
    erow_desc.reserve(perme.size());
    for (size_t q = 0; q < erow_slab.size(); ++q) {
        const Slab &sl = slabs_eg[erow_slab[q]];
        const long long row = sl.base + perme[q];
        erow_desc.push_back(make_int4((int)(row & 0xffffffffll), (int)(row >> 32), sl.aux, groups[sl.aux].span));
    }
    for (int c = 0; c < n_contigs; ++c)
        for (int e = 0; e < Ke; ++e) {
            const size_t ce = (size_t)c * Ke + e;
            const int q0 = ce_row_off[ce], q1 = ce_row_off[ce + 1];
            if (q1 > q0) while (slabs_ek.size() % 4 != 0) { Slab pad = slabs_ek.back(); pad.start = pad.end; slabs_ek.push_back(pad); }
            ek_slab_off[ce] = (int)slabs_ek.size();
            for (int q = q0; q < q1; q += S_EG) slabs_ek.push_back(Slab{q, std::min(q + S_EG, q1), (int)ce, e, contig_base[c]});
        }
    // (a padding slab sits in front of the first slab of the next key: it belongs to the PREVIOUS (contig, key)'s range only if
    // that range is recorded after it, so ranges are closed here, over the padded list)
    ek_slab_off[(size_t)n_contigs * Ke] = (int)slabs_ek.size();
}

void smcpp_im::setup_power() {
    int mx = 0;
    for (int g = 0; g < G; ++g) mx = std::max(mx, groups[g].span);
    int longest = 0;
    for (const Chunk &ch : chunks) longest = std::max(longest, ch.r1 - ch.r0);
    const char *pe = opt().has(smcpp_opt::O_POWER_PREPASS) ? opt().str(smcpp_opt::O_POWER_PREPASS).c_str() : nullptr;
    // spans below 32 (binned data: four squarings give every power); chunks short enough that pass 1 re-runs them whole
    // anyway (the rows of the pre-pass are all overwritten: it runs in float and its normalisers carry no eigenvalue scale)
    const bool coop_pre = chain_mode == 2 && Mp <= 64;
    const bool big_pre = chain_mode == 3 && Mp > 64 && Mp <= 256;
    // spans up to twelve bits (4095 positions); the cooperative chains read the powers beyond A^16 from L2 on the few rows
    // that need them, the streamed-operand ones stream every power anyway
    power_ok = (coop_pre || big_pre) && mx <= 4095 && Ke >= 1 && G >= 1 && longest <= 2000 && !(pe && atoi(pe) == 0) && !ss_static;
    max_span_pw = mx;
    pw_nbits = 5;
    while ((1 << pw_nbits) <= mx) ++pw_nbits;
    pw_npow = pw_nbits - 1;
    if (!power_ok) return;
    if (big_pre) {
        const size_t MM = (size_t)Mp * Mp;
        d_W.alloc((size_t)Ke * pw_nbits * MM);
        d_qBf.alloc((size_t)Ke * pw_nbits * MM);
        d_qBb.alloc((size_t)Ke * pw_nbits * MM);
        d_pre_qTf.alloc(MM);
        d_pre_qTdT.alloc(MM);
        return;
    }
    d_Bf.alloc((size_t)Ke * pw_npow * Mp * Mp);
    d_Bb.alloc((size_t)Ke * pw_npow * Mp * Mp);
}

void smcpp_im::alloc_device() {
    hipStream_t s = stream;
    d_rowinfo.upload(rowinfo, s);
    {
        // packed descriptors of the chain kernels and the "hot" eigen key (most span>1 rows) they keep in registers
        // ROWDESC_PAD span-1 descriptors of key 0 on both sides: the chain kernels prefetch descriptors up to 192 rows
        // past either end of a chunk without bounds tests (chains2.hpp)
        std::vector<int2> rd((size_t)total_rows + 2 * ROWDESC_PAD, make_int2(0, -1));
        std::vector<long long> cnt(std::max(1, Ke), 0);
        for (size_t r = 0; r < (size_t)total_rows; ++r) {
            const RowInfo &ri = rowinfo[r];
            rd[ROWDESC_PAD + r].x = ri.kid;
            rd[ROWDESC_PAD + r].y = ri.gid < 0 ? -1 : (ri.gid | (groups[ri.gid].eig << 20));
            if (ri.gid >= 0) cnt[groups[ri.gid].eig]++;
        }
        hot_eig = hot_eig2 = -1;
        for (int e = 0; e < Ke; ++e)
            if (hot_eig < 0 || cnt[e] > cnt[hot_eig]) hot_eig = e;
        for (int e = 0; e < Ke; ++e)
            if (e != hot_eig && (hot_eig2 < 0 || cnt[e] > cnt[hot_eig2])) hot_eig2 = e;
        d_rowdesc.upload(rd, s);
        HIPCHK(hipStreamSynchronize(s));
    }
    {
        // scan chains: descriptors {key slot, span}; slot = frequency rank of the key (the emission vectors of the first
        // ss_nlds slots live in LDS); same padding as above with span-1 rows of slot 0
        std::vector<long long> kc(K, 0);
        for (int c = 0; c < n_contigs; ++c)
            for (int i = 1; i <= Ls[c]; ++i) kc[rowinfo[(size_t)contig_base[c] + i].kid]++;
        std::vector<int> order(K);
        for (int k = 0; k < K; ++k) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return kc[x] > kc[y]; });
        ss_slot_of_key.assign(K, 0);
        for (int r = 0; r < K; ++r) ss_slot_of_key[order[r]] = r;
        const int MS = 64 * NPL;
        ss_nlds = (int)std::min<long long>(K, ((150 / std::max(1, ss_wpc)) * 1024) / ((long long)MS * 8));   // ss_wpc workgroups share a CU's 160 KB
        // (until round 5 the table was capped at 64 KB: the 192 six-int keys of config C4 - 96 KB at M = 48 - left 64 slots to the L2 path
        // and with them the kernel to its instantiation with vector-memory waits on every row: chains 0.86 -> see DESIGN.md section 6)
        if (ss_hybrid) ss_nlds = (int)std::max<long long>(1, std::min<long long>(K, (long long)((ss_dirsplit ? 158 : 150) * 1024 - ss_tab_bytes()) / ((long long)MS * 8)));   // one workgroup per CU
        ss_positions = 0;
        for (int c = 0; c < n_contigs; ++c)
            for (int i = 1; i <= Ls[c]; ++i) {
                const RowInfo &ri = rowinfo[(size_t)contig_base[c] + i];
                ss_positions += ri.gid < 0 ? 1 : ss_row_cost(groups[ri.gid].span);
            }
        if (ss_static) {
            std::vector<int2> rd((size_t)total_rows + 2 * ROWDESC_PAD, make_int2(0, 1));
            for (size_t r = 0; r < (size_t)total_rows; ++r) {
                const RowInfo &ri = rowinfo[r];
                // (upper 16 bits of x: the eigen key of a span > 1 row, read by the hybrid rows only)
                rd[ROWDESC_PAD + r] = make_int2(ss_slot_of_key[ri.kid] | ((ri.gid < 0 ? 0 : groups[ri.gid].eig) << 16), ri.gid < 0 ? 1 : groups[ri.gid].span);
            }
            // (round 6) a row that CONTINUES the caller's row of the row before it (long rows cut into pieces, build()): the vector
            // stored where it begins is not one the reference stores - no 1e-10 floor there (SS_ROW_CONT; chains_ss.hpp)
            if (split_spans)
                for (int c = 0; c < n_contigs; ++c)
                    for (int ell = 2; ell <= Ls[c]; ++ell)
                        if (piece_row[c][ell] == piece_row[c][ell - 1]) rd[ROWDESC_PAD + (size_t)contig_base[c] + ell].x |= SS_ROW_CONT;
            d_rowdesc_ss.upload(rd, s);
            HIPCHK(hipStreamSynchronize(s));
        }
    }
    upload_chunk_state();
    d_slabs_sc.upload(slabs_sc, s);
    d_slabs_rk.upload(slabs_rk, s);
    d_slabs_eg.upload(slabs_eg, s);
    d_perm1.upload(perm1, s);
    d_perm1k.upload(perm1k, s);
    d_perme.upload(perme, s);
    d_gk_slab_off.upload(gk_slab_off, s);
    d_s1_slab_off.upload(s1_slab_off, s);
    d_eb_slab_off.upload(eb_slab_off, s);
    d_eb_gid.upload(eb_gid, s);
    d_ce_bucket_off.upload(ce_bucket_off, s);
    d_erow_slab.upload(erow_slab, s);
    d_slabs_ek.upload(slabs_ek, s);
    d_slabs_fk.upload(slabs_fk, s);
    d_fk_c_off.upload(fk_c_off, s);
    d_fk_gk_off.upload(fk_gk_off, s);
    d_teams_fk.upload(teams_fk, s);
    d_teams_eg.upload(teams_eg, s);
    d_fk_c_team_off.upload(fk_c_team_off, s);
    d_eb_team_off.upload(eb_team_off, s);
    d_teams_rk.upload(teams_rk, s);
    d_s1_team_off.upload(s1_team_off, s);
    d_ek_slab_off.upload(ek_slab_off, s);
    d_epos_gid.upload(epos_gid, s);
    d_erow_desc.upload(erow_desc, s);
    d_contig_base.upload(contig_base, s);
    d_contig_L.upload(Ls, s);
    std::vector<int> gs(G), ge(G);
    for (int g = 0; g < G; ++g) { gs[g] = groups[g].span; ge[g] = groups[g].eig; }
    d_g_span.upload(gs, s);
    d_g_eig.upload(ge, s);
    d_e_kid.upload(eig_kid, s);
    setup_power();
    d_alpha.alloc((size_t)total_rows * Mp);
    d_beta.alloc((size_t)total_rows * Mp);
    d_cnorm.alloc((size_t)total_rows);
    d_logc.alloc((size_t)total_rows);
    d_w1.alloc((size_t)total_rows);
    {
        // blocks per contig of the log-likelihood reduction: ~2 000 rows each (64 blocks took 0.13 ms on a contig of a million rows)
        int maxL = 0;
        for (int c = 0; c < n_contigs; ++c) maxL = std::max(maxL, Ls[c]);
        llblk = std::max(64, std::min(1024, (maxL + 2047) / 2048));
    }
    d_llpart.alloc((size_t)n_contigs * llblk);
    d_loglik.alloc(n_contigs);
    d_gpart.alloc(std::max<size_t>(1, slabs_sc.size()) * Mp);
    // omega*U and W of the eigen rows only go through memory when the fused kernel cannot be used (M > 64)
    d_Xs.alloc(NT <= 4 ? 1 : std::max<size_t>(1, (size_t)n_e_rows) * Mp);
    d_Ys.alloc(NT <= 4 ? 1 : std::max<size_t>(1, (size_t)n_e_rows) * Mp);
    // (d_part_e / d_red_e - one M x M partial per span GROUP slab / bucket - are allocated where they are used: un-binned data have
    // 10^5 groups and never take those paths when M <= 64)
    d_part_1.alloc(std::max<size_t>(1, stats_team_on() ? teams_rk.size() : slabs_rk.size()) * Mp * Mp);      // (one partial per team)
    // shares of the cross-slab reduction of the span-1 rank partials: few contigs, small M -> more, shorter shares (one contig at M = 64:
    // 8 shares of 126 slabs took 38 us of dependent loads)
    ZS = (int)std::max<long long>(8, std::min<long long>(16, 2048 / std::max<long long>(1, (long long)n_contigs * ceil_div((long long)Mp * Mp, 256))));
    d_red_1.alloc((size_t)n_contigs * ZS * Mp * Mp);
    // (the monomorphic key alone holds half the span-1 slabs of a contig: one block walking them took 42 us on the headline)
    ZG = (int)std::max<long long>(1, std::min<long long>(16, 1024 / std::max<long long>(1, (long long)n_contigs * K)));
    d_red_g.alloc((size_t)n_contigs * K * Mp * ZG);
    d_Z.alloc(std::max<size_t>(1, (size_t)n_contigs * Ke) * Mp * Mp);
    d_Y.alloc(std::max<size_t>(1, (size_t)n_contigs * Ke) * Mp * Mp);
    d_xisum.alloc((size_t)n_contigs * Mp * Mp);
    d_gsum.alloc((size_t)n_contigs * K * Mp);
    d_gamma0.alloc((size_t)n_contigs * Mp);
    d_E.alloc((size_t)K * Mp);
    d_dpow.alloc(std::max<size_t>(1, (size_t)G) * Mp);
    d_g_scale.alloc(std::max(1, G));
    d_g_logscale.alloc(std::max(1, G));
    d_pi_f.alloc(Mp);
    d_Tf.alloc((size_t)Mp * Mp);
    d_TdT.alloc((size_t)Mp * Mp);
    d_Td.alloc((size_t)Mp * Mp);
    const size_t em = std::max<size_t>(1, (size_t)Ke) * Mp * Mp;
    d_PinvT.alloc(em); d_PT.alloc(em); d_Prm.alloc(em); d_Pinvrm.alloc(em);
    d_dsc.alloc(std::max<size_t>(1, (size_t)Ke) * Mp);
    d_dun.alloc(std::max<size_t>(1, (size_t)Ke) * Mp);
    // zero the row state once so padded lanes / unused rows hold finite values
    d_alpha.zero(s); d_beta.zero(s); d_cnorm.zero(s); d_logc.zero(s); d_w1.zero(s);
    HIPCHK(hipStreamSynchronize(s));
}

