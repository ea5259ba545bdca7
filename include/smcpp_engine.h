/* smcpp_engine.h — C ABI of the MI355X-native SMC++ E-step engine (libsmcpp_engine.so).
 *
 * This is the drop-in boundary for the reference's Cython binding (smcpp/_smcpp.pyx + _smcpp.pxd): every entry
 * point below replaces one member of the C++ classes that `_smcpp.pxd:41-70` declares to Cython.  Opaque handle,
 * plain pointers and sizes, int status (0 = ok, nonzero = error with smcpp_last_error() holding the what()-string
 * the reference would have thrown as std::runtime_error -> Python RuntimeError, `_smcpp.pxd:43-52`).
 *
 * Matrix arguments are C-contiguous row-major doubles unless stated otherwise (the layout `store_matrix`,
 * src/common.cpp:8-11, hands to numpy).  Observation arrays are int32 [L x (1+3P)] rows (span, (a,b,nb) x P),
 * exactly what `InferenceManager::map_obs` (src/inference_manager.cpp:180-188) maps; they are copied to the
 * device at construction and need not outlive the call.
 *
 * Thread safety: an instance is used from one host thread at a time (as in the reference);
 * different instances are independent.  Every call may be made with the Python GIL released.
 */
#ifndef SMCPP_ENGINE_H
#define SMCPP_ENGINE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smcpp_im smcpp_im;

/* Thread-local message of the last failing call on this thread. */
const char *smcpp_last_error(void);

/* ---- construction / destruction -------------------------------------------------------------------------- */

/* OnePopInferenceManager(n, obs_lengths, observations, hidden_states, polarization_error)
 * include/inference_manager.h:130-139, src/inference_manager.cpp:506-516; bound at _smcpp.pyx:312-320.
 * hs has n_hs = M+1 ascending entries (last may be +inf).  device < 0 selects the current HIP device. */
int smcpp_create_onepop(int n, int n_contigs, const int *Ls, const int *const *obs,
                        int n_hs, const double *hs, double polarization_error, int device, smcpp_im **out);

/* TwoPopInferenceManager(n1, n2, a1, a2, ...)  include/inference_manager.h:141-157,
 * src/inference_manager.cpp:518-540; bound at _smcpp.pyx:338-351.  Requires a1 + a2 == 2 and (a1,a2) != (0,2). */
int smcpp_create_twopop(int n1, int n2, int a1, int a2, int n_contigs, const int *Ls, const int *const *obs,
                        int n_hs, const double *hs, double polarization_error, int device, smcpp_im **out);

/* delete im  (_smcpp.pyx:154-155) */
void smcpp_destroy(smcpp_im *im);

/* ---- parameters ----------------------------------------------------------------------------------------- */

/* InferenceManager::setTheta / setRho / setAlpha  (src/inference_manager.cpp:71-87): mark dirty only. */
int smcpp_set_theta(smcpp_im *im, double theta);
int smcpp_set_rho(smcpp_im *im, double rho);
int smcpp_set_alpha(smcpp_im *im, double alpha);

/* InferenceManager::setParams(ParameterVector)  (src/inference_manager.cpp:256-260, _smcpp.pyx:66-83,327-332).
 * a[K] piece sizes, s[K] piece lengths; da[K x nder] forward-mode derivative seeds of a (may be NULL, nder = 0). */
int smcpp_set_params(smcpp_im *im, int K, const double *a, const double *da, int nder, const double *s);

/* Raw-parameter entry (extension, SURVEY.md §7 design note): hand the engine the products of the cold host
 * preparation directly — pi[M], T[M x M], and one emission vector per key (keys [K x 3P] int32, E [K x M]).
 * Every key that occurs in the observations must be present.  Replaces, for one E-step, what
 * do_dirty_work() (src/inference_manager.cpp:213-229) would have recomputed. */
int smcpp_set_raw(smcpp_im *im, const double *pi, const double *T, int K, const int *keys, const double *E);

/* ---- the hot path --------------------------------------------------------------------------------------- */

/* InferenceManager::Estep(bool)  (src/inference_manager.cpp:108-114 -> HMM::Estep, src/hmm.cpp:45-153). */
int smcpp_estep(smcpp_im *im, int forward_backward_only);

/* InferenceManager::loglik()  (src/inference_manager.cpp:174-177): one value per contig. */
int smcpp_loglik(smcpp_im *im, double *out);

/* InferenceManager::Q()  (src/inference_manager.cpp:116-126 -> HMM::Q, src/hmm.cpp:155-193): the four terms
 * summed over contigs; jac [4 x nder] may be NULL. */
int smcpp_q(smcpp_im *im, double val[4], double *jac);

/* Number of derivative directions the current parameters carry (nder of the last smcpp_set_params; 0 after set_raw). */
int smcpp_num_derivatives(smcpp_im *im);

/* ---- state / getters (all copy out) -------------------------------------------------------------------------- */

int smcpp_set_save_gamma(smcpp_im *im, int on);          /* InferenceManager::saveGamma, _smcpp.pyx:201-205 */
int smcpp_get_save_gamma(smcpp_im *im);
int smcpp_num_states(smcpp_im *im);                      /* M = n_hs - 1 */
int smcpp_num_contigs(smcpp_im *im);
int smcpp_num_keys(smcpp_im *im);                        /* distinct block_keys over all contigs */
int smcpp_key_len(smcpp_im *im);                         /* 3P */
int smcpp_get_hidden_states(smcpp_im *im, double *hs);   /* _smcpp.pyx:207-213 */
int smcpp_set_hidden_states(smcpp_im *im, int n_hs, const double *hs);
int smcpp_get_keys(smcpp_im *im, int *keys);             /* [K x 3P], lexicographic (block_key.h:47-56) */

int smcpp_get_xisum(smcpp_im *im, int contig, double *out);        /* getXisums(): [M x M]   */
/* getGammas(): [M x (L+1)] row-major if save_gamma was set for the last E-step, else [M x 1] */
int smcpp_get_gamma(smcpp_im *im, int contig, double *out);
/* Number of columns smcpp_get_gamma writes for this contig: L+1 if the LAST E-step ran with save_gamma, else 1
 * (the flag may have been toggled since); -1 for a bad contig index.  Size the buffer from this. */
int smcpp_gamma_cols(smcpp_im *im, int contig);
/* getGammaSums(): vals [K x M]; present[K] = 1 where the reference's std::map would hold the key */
int smcpp_get_gamma_sums(smcpp_im *im, int contig, double *vals, unsigned char *present);
int smcpp_get_pi(smcpp_im *im, double *out);                       /* getPi(): [M]           */
int smcpp_get_transition(smcpp_im *im, double *out);               /* getTransition(): [M x M] */
int smcpp_get_emission_probs(smcpp_im *im, double *out);           /* getEmissionProbs(): [K x M] */
/* Jacobians of the three getters above with respect to the derivative seeds of the last smcpp_set_params
 * (`store_matrix(const Matrix<adouble>&, double*, double*)`, src/common.cpp; _smcpp.pyx:215-275): row-major
 * [M x nder], [M*M x nder], [K*M x nder].  Nothing is written when nder == 0; an error after smcpp_set_raw. */
int smcpp_get_pi_jac(smcpp_im *im, double *out);
int smcpp_get_transition_jac(smcpp_im *im, double *out);
int smcpp_get_emission_probs_jac(smcpp_im *im, double *out);
/* getEmission() (include/inference_manager.h:72,166; src/inference_manager.cpp:392-407): the conditioned SFS after
 * incorporate_theta per hidden state, flattened row-major: [M x cols], cols = prod_p (a_p + 1)(n_p + 1);
 * jac [M*cols x nder] may be NULL. */
int smcpp_num_emission_cols(smcpp_im *im);
int smcpp_get_emission(smcpp_im *im, double *out, double *jac);
/* posterior decoding indices: argmax_m gamma[m, ell] for ell = 0..L (needs save_gamma) */
int smcpp_get_gamma_argmax(smcpp_im *im, int contig, int *out);

/* ---- multi-GPU ----------------------------------------------------------------------- */

/* Key dictionary shared by every rank (union of the ranks' smcpp_get_keys lists, lexicographic): fixes the layout
 * of the gamma_sums block of the packed buffer. */
int smcpp_set_global_keys(smcpp_im *im, int Kg, const int *gkeys);

/* Packed sufficient statistics of this rank's contigs, ready for one all-reduce(sum, fp64):
 * [ sum loglik | gamma0 (M) | xisum (M*M) | gamma_sums dense (K*M) ].  Returns the length via n_out when
 * buf == NULL.  dev != 0: buf is a device pointer (filled by one kernel on the engine's stream) else a host pointer.
 * dev == 1 returns after that kernel has finished; dev == 2 returns after the ENQUEUE: the caller consumes buf on the
 * engine's stream (smcpp_stream), e.g. an RCCL all-reduce ordered behind the pack kernel with no host wait in between. */
int smcpp_pack_stats(smcpp_im *im, double *buf, long *n_out, int dev);
/* Hand the all-reduced buffer back; Q() then evaluates on the global statistics. */
int smcpp_unpack_stats(smcpp_im *im, const double *buf, long n, int dev);

/* The same exchange issued by the engine itself through RCCL's C API on its own stream (no counterpart in the reference, which
 * sums over the contigs of one process: src/inference_manager.cpp:116-126, smcpp/_smcpp.pyx:303-308):
 *   smcpp_rccl_unique_id   rank 0 obtains the 128-byte ncclUniqueId (distribute it to the other ranks by any means);
 *   smcpp_rccl_init        every rank, after smcpp_set_global_keys: ncclCommInitRank on the manager's device;
 *   smcpp_rccl_exchange    after smcpp_estep: k_pack_stats -> ncclAllReduce(sum, f64, in place) -> the reduced sum of the
 *                          log-likelihoods into pinned host memory, all on the engine's stream; returns that sum;
 *   smcpp_rccl_unpack      hands the reduced statistics (still in the engine's device buffer) to Q();
 *   smcpp_rccl_fetch       (tests) a host copy of that buffer;  smcpp_rccl_destroy: ncclCommDestroy (also done by smcpp_destroy).
 * libpath: the RCCL library the process already uses (e.g. torch's lib/librccl.so), NULL or "" for "librccl.so.1"; it is resolved
 * with dlopen - the engine has no link-time dependency on RCCL. */
int smcpp_rccl_unique_id(const char *libpath, char *out128);
int smcpp_rccl_init(smcpp_im *im, const char *libpath, const char *id128, int rank, int world);
int smcpp_rccl_exchange(smcpp_im *im, double *loglik_sum);
int smcpp_rccl_unpack(smcpp_im *im);
int smcpp_rccl_fetch(smcpp_im *im, double *out, long n);
int smcpp_rccl_destroy(smcpp_im *im);

/* InferenceManager::debug (_smcpp.pxd:53): a public flag the reference declares to Cython; nothing in its C++ reads it. */
int smcpp_set_debug(smcpp_im *im, int on);
int smcpp_get_debug(smcpp_im *im);

/* ---- engine controls (no reference counterpart) ---------------------------------------------------------- */

/* Rows per chunk of the chunk-parallel chains (0 = automatic) and the chunk-boundary convergence tolerances. */
int smcpp_set_chunking(smcpp_im *im, int rows_per_chunk, double eps_alpha, double eps_beta);

/* Where the cold preparation of a one-population manager runs: 0 (default) = conditioned SFS, incorporate_theta and the
 * emission table on the device (smcpp_amd/csrc/prep_dev.hpp; the host only builds the O(pieces) rate function, pi and the
 * transition matrix), 1 = everything on the host (smcpp_amd/csrc/prep.hpp, rounds 1-3; also selected by SMCPP_PREP=host).
 * Results agree to the last bits of the exponentials; kept for the tests and as the baseline of bench.py --workload qgrad. */
int smcpp_set_prep_mode(smcpp_im *im, int host);

/* Extension (off by default): start the chunk-parallel chains of the next E-step from the converged chunk-boundary
 * vectors of the previous E-step of this manager instead of pi / the uniform vector.  In an EM or optimiser loop the
 * parameters move little between calls, so the re-run passes merge after a fraction of a chunk; the fixed-point
 * iteration and its convergence certificate are unchanged, results agree with a cold start within eps. */
int smcpp_set_warm_start(smcpp_im *im, int on);
/* Kernel-time breakdown of the last E-step in milliseconds:
 * [host_prep, chains_wall, forward, backward, stats, finalize, total_device, fwd_passes, bwd_passes]
 * (forward and backward overlap when the two chains run on separate streams; chains_wall is their union) */
int smcpp_last_timing(smcpp_im *im, double out[9]);
/* diagnostics: the chain kernel family in use (0 generic, 1 LDS-resident, 2 cooperative, 3 cooperative with streamed
 * operands, 4 lock-step on the matrix cores, 5 scans over the semiseparable structure of the transition matrix -
 * src/transition.cpp:176-254 - one position per step, no eigensystem; an E-step whose T lacks that structure runs the
 * dense kernels instead; 6 = family 5 with HYBRID rows: un-binned data, a row whose span exceeds a few positions is one
 * eigen-power step P d^s P^-1 inside the scan kernel - hmm.cpp:72-78,104-112 - instead of `span` scan steps) */
/* Host-only (no device needed): chunks per contig of the scan chains for `nslots` wavefront slots, cost[c] = positions (or cost
 * units) of contig c, rows[c] its row count; never more chunks than slots in total unless there are more contigs than slots, and
 * the longest chunk as short as the slot count allows (the reference parallelises over contigs only, inference_manager.cpp:89-94;
 * this is the engine's own decomposition, exported for the CPU tests). */
int smcpp_host_chunk_counts(int n_contigs, const long long *cost, const int *rows, long long nslots, long long floor_cost, int *out);
int smcpp_chain_mode(smcpp_im *im);
/* Every SMCPP_* environment switch of the engine is parsed ONCE per process (smcpp_amd/csrc/engine_options.hpp holds the one
 * table of them); smcpp_reload_options re-reads the environment (tests; never while an E-step runs).  smcpp_describe writes one
 * JSON object - the switches that are set and the plan `im` resolved (chain family, chunk counts, history passes, whether the
 * stored passes of the last E-step ran their scans in float) - and returns the length it needs; `im` may be NULL. */
void smcpp_reload_options(void);
int smcpp_describe(smcpp_im *im, char *buf, int cap);
/* Test hook of family 5: one position of both scan chains on nvec vectors: out_f = e o (T^T x), out_b = T (e o x); T is
 * [M][M] row-major, x / e / out_* are [nvec][M].  Returns 2 when T has no semiseparable structure (nothing is written). */
int smcpp_debug_ss_apply(int M, const double *T, int nvec, const double *x, const double *e, double *out_f, double *out_b);
/* ... the same position by the step of the stored passes with every scan in float (M <= 64; the M <= 32 form when M <= 32). */
int smcpp_debug_ss_apply_float_scans(int M, const double *T, int nvec, const double *x, const double *e, double *out_f, double *out_b);
/* Host phase of the last E-step in milliseconds: [cold preparation A6-A10 (0 when the parameters were still fresh or
 * came from smcpp_set_raw), eigensystems, layouts + staging + copy enqueue, whole host phase] */
int smcpp_last_host_timing(smcpp_im *im, double out[4]);
/* The HIP stream the engine launches on (a hipStream_t), for event timing by the caller. */
void *smcpp_stream(smcpp_im *im);
/* The HIP device the manager lives on (buffers handed to smcpp_pack_stats / smcpp_unpack_stats must live there). */
int smcpp_device(smcpp_im *im);

/* init_logger_cb (include/common.h, _smcpp.pxd:26, _smcpp.pyx:32-55): the engine's messages (level "DEBUG", "INFO",
 * "WARNING", ...) are handed to the binding, which forwards them to Python's logging; NULL = silent (the default). */
void smcpp_init_logger_cb(void (*cb)(const char *name, const char *level, const char *message));
/* init_cache (include/matrix_cache.h, _smcpp.pxd:87-88, src/matrix_cache.cpp:46-110): prefix of the on-disk store of the
 * n-only conditioned-SFS tables (exact-rational work: 0.35 s at n = 10, 0.9 s at n = 50 when computed); one file
 * `<path>.n<N>` per sample size, written atomically.  Empty / NULL = no store (tables live in the process only). */
int smcpp_init_cache(const char *path);
/* openmp.omp_set_num_threads (_smcpp.pyx:61-64): threads of the host-side preparation. */
void smcpp_set_num_threads(int k);

/* ---- pre-HMM data shaping on the device ------------------------------------------------ */

/* What `smc++ estimate` does to a contig before an inference manager sees it (smcpp/data_filter.py:166-203), as device kernels
 * over rows int32 [L][1 + 3 P] handed over by the caller (host memory; copied to HBM once):
 *   mode 0  thin_data(rows, thinning = p0, offset = p1)          smcpp/_estimation_tools.pyx:8-84
 *   mode 1  bin_observations(rows, w = p0), na[P] distinguished lineages per population   _estimation_tools.pyx:113-173
 *   mode 2  compress_repeated_obs(rows)                           smcpp/estimation_tools.py:51-60
 *   mode 3  Thin(p0) -> Bin(p1, na) -> Compress without leaving HBM (the pipeline of data_filter.py)
 * Bit-exact with the reference's code (goldens G23 / G11).  The result stays on the device (per calling thread): *rows_out = its
 * row count, *kernel_ms (optional) = the device time with the input resident in HBM; smcpp_dev_shape_fetch copies the rows out
 * (int32 [rows_out][ncol]).  smcpp_amd/data.py holds the host implementation of the same functions. */
int smcpp_dev_shape(int mode, long long L, int ncol, const int *rows, long long p0, long long p1, const long long *na,
                    long long *rows_out, double *kernel_ms);
int smcpp_dev_shape_fetch(int *out);

/* ---- host-only helpers (no device needed; used by the CPU test-suite) ------------------------------------ */

/* Test hook: 1 = evaluate the conditioned SFS term by term exactly as src/piecewise_constant_rate_function.cpp:214-334
 * and src/conditioned_sfs.cpp:42-83 write it (O(pieces^2 n^2), reproduces the compiled reference to 1e-15 relative);
 * 0 (default) = the factored O(pieces n^2) evaluation (same integrals through prefix / suffix sums over the pieces,
 * within 5e-16 absolute of the literal one on the emission table).  Returns the previous setting. */
int smcpp_host_set_csfs_direct(int on);

/* eigensystem(EigenSolver(A)) as TransitionBundle::update uses it (src/transition_bundle.cpp:22,
 * include/transition_bundle.h:9-30): P_r, Pinv_r [n x n], d_r [n], scale = max |d|, max |imag d|. */
int smcpp_host_eigensystem(int n, const double *A, double *P, double *Pinv, double *d, double *scale,
                           double *max_imag);
/* the same, computed by `threads` cooperating threads (smcpp_amd/csrc/nonsym_eig_team.hpp: bit-identical results); what the
 * engine uses for M >= 128 */
int smcpp_host_eigensystem_team(int n, const double *A, int threads, double *P, double *Pinv, double *d, double *scale,
                                double *max_imag);

/* One-population cold preparation without an engine instance:
 * model pieces (a, s)[Kp] + hidden states -> pi [M], T [M x M], E [K x M] for the given keys [K x 3]. */
int smcpp_host_prep_onepop(int n, int n_hs, const double *hs, double polarization_error, int Kp, const double *a,
                           const double *s, double theta, double rho, double alpha, int K, const int *keys,
                           double *pi, double *T, double *E);

/* The same with forward-mode derivative seeds da [Kp x nder] on the piece sizes: additionally returns the Jacobians
 * dpi [M x nder], dT [M*M x nder], dE [K*M x nder] (what the reference carries in its adouble type, common.h:22-25). */
int smcpp_host_prep_onepop_jac(int n, int n_hs, const double *hs, double polarization_error, int Kp, const double *a,
                               const double *da, int nder, const double *s, double theta, double rho, double alpha,
                               int K, const int *keys, double *pi, double *T, double *E, double *dpi, double *dT,
                               double *dE);

/* The same preparation with the conditioned SFS (A10), incorporate_theta and the emission table (A6) evaluated by the HIP
 * kernels of smcpp_amd/csrc/prep_dev.hpp - what smcpp_estep / smcpp_q run for one-population managers - instead of the host
 * routines (src/conditioned_sfs.cpp:13-148, src/inference_manager.cpp:389-482; pi and the transition matrix come from the
 * host either way).  mode 0: on the current HIP device; mode 1: the same kernel phases run serially on the host (no device
 * needed: CPU test-suite).  da / nder may be NULL / 0 (then dpi, dT, dE, dsfs are not written); sfs [M x 3(n+1)] and dsfs
 * [M*3(n+1) x nder] (the conditioned SFS per hidden state after incorporate_theta) may be NULL. */
int smcpp_dev_prep_onepop(int n, int n_hs, const double *hs, double polarization_error, int Kp, const double *a,
                          const double *da, int nder, const double *s, double theta, double rho, double alpha, int K,
                          const int *keys, int mode, double *pi, double *T, double *E, double *dpi, double *dT, double *dE,
                          double *sfs, double *dsfs);

/* Test hook (no device needed): Q's four terms (src/hmm.cpp:155-193) and their gradient jac [4 x nder] for statistics summed
 * over contigs - g0 [M], xi [M x M], gs [K x M] - computed by the phases of the device kernel (prep_dev.hpp: k_q_reduce) run
 * serially on the host, from the emulated device preparation and the O(M) generator planes of the transition matrix: the data
 * path smcpp_q takes on the GPU for one-population managers. */
int smcpp_dev_q_emulate(int n, int n_hs, const double *hs, double polarization_error, int Kp, const double *a, const double *da,
                        int nder, const double *s, double theta, double rho, double alpha, int K, const int *keys,
                        const double *g0, const double *xi, const double *gs, double *val, double *jac);

/* PyRateFunction.R / average_coal_times (smcpp/_smcpp.pyx:370-389): cumulative hazard R at t[0..nt) for the model
 * pieces (a, s) and, when n_hs >= 2, E[T | hs_i <= T < hs_{i+1}] for the n_hs-1 intervals. */
int smcpp_host_rate_function(int Kp, const double *a, const double *s, int n_hs, const double *hs, int nt,
                             const double *t, double *R_out, double *avg_ct_out);

/* The same with derivative seeds (PyRateFunction on a model with differentiable pieces): dR_out [nt x nder],
 * davg_ct_out [(n_hs-1) x nder]. */
int smcpp_host_rate_function_jac(int Kp, const double *a, const double *da, int nder, const double *s, int n_hs,
                                 const double *hs, int nt, const double *t, double *R_out, double *dR_out,
                                 double *avg_ct_out, double *davg_ct_out);

/* PyRateFunction.random_coal_times (smcpp/_smcpp.pyx:391-399, piecewise_constant_rate_function.cpp:337-368): K
 * coalescence times conditioned on [t1, t2), one std::mt19937 per draw seeded with seeds[i]; returns t and R(t). */
int smcpp_host_random_coal_times(int Kp, const double *a, const double *s, double t1, double t2, int K,
                                 const unsigned long long *seeds, double *t_out, double *R_out);

/* raw_sfs (smcpp/_smcpp.pyx:401-412, sfs_cython inference_manager.cpp:492-504): the 3 x (n+1) conditioned SFS of the
 * single hidden state [t1, t2) before incorporate_theta; dsfs [3*(n+1) x nder] may be NULL when nder == 0. */
int smcpp_host_raw_sfs(int n, int Kp, const double *a, const double *da, int nder, const double *s, double t1,
                       double t2, int below_only, double *sfs, double *dsfs);

/* TwoPopInferenceManager::setParams (src/inference_manager.cpp:542-550, smcpp/_smcpp.pyx:353-368): the distinguished
 * model (pi, transition, average coalescence times), the two per-population models and the split time for the joint
 * CSFS.  d* are derivative seeds [K x nder] (NULL = zero) sharing one set of nder directions (`model.dlist`). */
int smcpp_set_params_twopop(smcpp_im *im, int Kd, const double *ad, const double *sd, const double *dad, int K1,
                            const double *a1, const double *s1, const double *da1, int K2, const double *a2,
                            const double *s2, const double *da2, double split, int nder);

/* joint_csfs (smcpp/_smcpp.pyx:416-437; JointCSFS src/jcsfs.cpp:219-420): per hidden state the tensor
 * [(a1+1) x (n1+1) x (a2+1) x (n2+1)], out [(n_hs-1) x that], Kmc Monte-Carlo draws for the averaged Moran
 * transition below the split (the reference's default is 10).  With nder > 0, da1 / da2 [K x nder] seed the piece
 * sizes (NULL = zero) and dout [size x nder] receives the Jacobian. */
int smcpp_host_joint_csfs(int n1, int n2, int a1, int a2, int n_hs, const double *hs, int K1, const double *pa1,
                          const double *ps1, const double *da1, int K2, const double *pa2, const double *ps2,
                          const double *da2, int nder, double split, int Kmc, double *out, double *dout);

/* Whole two-population cold preparation without an engine instance: pi [M], T [M x M], E [K x M] for keys [K x 6]. */
int smcpp_host_prep_twopop(int n1, int n2, int a1, int a2, int n_hs, const double *hs, double polarization_error,
                           int Kd, const double *ad, const double *sd, int K1, const double *pa1, const double *ps1,
                           int K2, const double *pa2, const double *ps2, double split, double theta, double rho,
                           double alpha, int K, const int *keys, double *pi, double *T, double *E);

#ifdef __cplusplus
}
#endif
#endif
