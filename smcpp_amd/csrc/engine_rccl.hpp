// engine_rccl.hpp - part of the ONE translation unit engine.hip (included there, in order; not a standalone header):
// the engine-issued exchange through RCCL's C API (opt-in).
extern "C" {
// ---------------------------------------------------------------------------------------------------------------
// The exchange of an E-step issued by the ENGINE on its own stream through RCCL's C API (the reference has no
// counterpart: it sums over contigs in one process, inference_manager.cpp:116-126):
//     k_pack_stats -> ncclAllReduce(sum, f64, in place) -> k_publish_scalar (sum of the log-likelihoods into pinned host memory)
// all stream-ordered, the host polls one word - no event hop to a communication stream, no host wait before the collective, no
// copy engine for the scalar.  The library is the one the process already holds (path handed over by the caller: torch's RCCL when
// torch.distributed is in use), resolved with dlopen / dlsym so that the engine has no link-time dependency on it.
// ---------------------------------------------------------------------------------------------------------------
struct ncclUniqueIdBlob { char b[128]; };       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
static void check_rc(int rc) { if (rc) throw std::runtime_error(g_err); }
struct RcclDirect {
    void *lib = nullptr;
    void *comm = nullptr;
    int world = 1, rank = 0;
    int (*get_uid)(void *) = nullptr;
    int (*init_rank)(void **, int, ncclUniqueIdBlob, int) = nullptr;
    int (*all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*destroy)(void *) = nullptr;
    const char *(*err_string)(int) = nullptr;
    DevBuf<double> buf;
    long n = 0;
    double *h_val = nullptr, *d_val_view = nullptr;
    int *h_flag = nullptr, *d_flag_view = nullptr;
    int epoch = 0;
    bool reduced_in_buf = false;
};
// ONE dlopen handle per process and library path (never closed: the library is the one torch.distributed holds anyway, and a
// communicator may outlive the manager that asked for it first); the enum values and the by-value unique id hard-coded below are
// those of rccl.h of the NCCL 2.x ABI, so the version the library reports is checked once per handle.
static void *rccl_handle(const char *libpath) {
    static std::mutex mu;
    static std::map<std::string, void *> handles;
    const std::string key = libpath && *libpath ? libpath : "librccl.so.1";
    std::lock_guard<std::mutex> lk(mu);
    auto it = handles.find(key);
    if (it != handles.end()) return it->second;
    void *h = dlopen(key.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) throw std::runtime_error(std::string("RCCL library not loadable: ") + dlerror());
    auto get_version = reinterpret_cast<int (*)(int *)>(dlsym(h, "ncclGetVersion"));
    int ver = 0;
    if (!get_version || get_version(&ver) != 0 || ver < 20000 || ver >= 30000) {
        dlclose(h);
        throw std::runtime_error("RCCL library " + key + ": ncclGetVersion reports " + std::to_string(ver) +
                                 " - the direct exchange is written against the NCCL 2.x ABI (ncclDouble = 8, ncclSum = 0, 128-byte unique id)");
    }
    handles[key] = h;
    return h;
}
static void rccl_resolve(RcclDirect &r, const char *libpath) {
    r.lib = rccl_handle(libpath);
    auto sym = [&](const char *nm) {
        void *p = dlsym(r.lib, nm);
        if (!p) throw std::runtime_error(std::string("RCCL symbol missing: ") + nm);
        return p;
    };
    r.get_uid = reinterpret_cast<int (*)(void *)>(sym("ncclGetUniqueId"));
    r.init_rank = reinterpret_cast<int (*)(void **, int, ncclUniqueIdBlob, int)>(sym("ncclCommInitRank"));
    r.all_reduce = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)>(sym("ncclAllReduce"));
    r.destroy = reinterpret_cast<int (*)(void *)>(sym("ncclCommDestroy"));
    r.err_string = reinterpret_cast<const char *(*)(int)>(sym("ncclGetErrorString"));
}
static void rccl_check(const RcclDirect &r, int rc, const char *what) {
    if (rc != 0) throw std::runtime_error(std::string("RCCL: ") + what + ": " + (r.err_string ? r.err_string(rc) : "error"));
}
int smcpp_rccl_unique_id(const char *libpath, char *out128) {
    API_BEGIN
    RcclDirect r;
    rccl_resolve(r, libpath);
    ncclUniqueIdBlob id;
    rccl_check(r, r.get_uid(&id), "ncclGetUniqueId");
    std::memcpy(out128, id.b, 128);
    API_END
}
int smcpp_rccl_init(smcpp_im *im, const char *libpath, const char *id128, int rank, int world) {
    API_BEGIN
    if (im->rccl) throw std::runtime_error("smcpp_rccl_init: already initialised");
    HIPCHK(hipSetDevice(im->device));
    std::unique_ptr<RcclDirect> r(new RcclDirect());
    rccl_resolve(*r, libpath);
    ncclUniqueIdBlob id;
    std::memcpy(id.b, id128, 128);
    r->world = world; r->rank = rank;
    rccl_check(*r, r->init_rank(&r->comm, world, id, rank), "ncclCommInitRank");
    HIPCHK(hipHostMalloc((void **)&r->h_val, 64, hipHostMallocCoherent | hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void **)&r->d_val_view, r->h_val, 0));
    HIPCHK(hipHostMalloc((void **)&r->h_flag, 64, hipHostMallocCoherent | hipHostMallocMapped));
    *r->h_flag = 0;
    HIPCHK(hipHostGetDevicePointer((void **)&r->d_flag_view, r->h_flag, 0));
    im->rccl = r.release();
    API_END
}
int smcpp_rccl_destroy(smcpp_im *im) {
    API_BEGIN
    if (im->rccl) {
        HIPCHK(hipSetDevice(im->device));
        (void)hipStreamSynchronize(im->stream);
        if (im->rccl->comm && im->rccl->destroy) (void)im->rccl->destroy(im->rccl->comm);
        if (im->rccl->h_val) (void)hipHostFree(im->rccl->h_val);
        if (im->rccl->h_flag) (void)hipHostFree(im->rccl->h_flag);
        delete im->rccl;
        im->rccl = nullptr;
    }
    API_END
}
// After smcpp_estep: pack -> all-reduce -> publish, returns the all-reduced sum of the log-likelihoods.  The reduced statistics
// stay in the engine's device buffer until smcpp_rccl_unpack hands them to Q.
int smcpp_rccl_exchange(smcpp_im *im, double *loglik_sum) {
    API_BEGIN
    RcclDirect *r = im->rccl;
    if (!r) throw std::runtime_error("smcpp_rccl_exchange: smcpp_rccl_init has not been called");
    long n = 0;
    check_rc(smcpp_pack_stats(im, nullptr, &n, 0));
    if (r->n != n) { r->buf.alloc((size_t)n); r->n = n; }
    check_rc(smcpp_pack_stats(im, r->buf.p, nullptr, 2));                  // enqueue only
    rccl_check(*r, r->all_reduce(r->buf.p, r->buf.p, (size_t)n, /* ncclDouble */ 8, /* ncclSum */ 0, r->comm, im->stream), "ncclAllReduce");
    const int ep = ++r->epoch;
    hipLaunchKernelGGL(k_publish_scalar, dim3(1), dim3(1), 0, im->stream, (const double *)r->buf.p, r->d_val_view, r->d_flag_view, ep);
    HIPCHK(hipGetLastError());
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(r->h_flag, __ATOMIC_ACQUIRE) != ep) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
        if ((++spins & 0x3fff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) {
            HIPCHK(hipStreamSynchronize(im->stream));
            break;
        }
    }
    *loglik_sum = *r->h_val;
    r->reduced_in_buf = true;
    API_END
}
int smcpp_rccl_unpack(smcpp_im *im) {
    API_BEGIN
    RcclDirect *r = im->rccl;
    if (!r || !r->reduced_in_buf) throw std::runtime_error("smcpp_rccl_unpack: no reduced statistics to hand over");
    check_rc(smcpp_unpack_stats(im, r->buf.p, r->n, 1));
    r->reduced_in_buf = false;
    API_END
}
// (test hook) a copy of the engine's reduce buffer
int smcpp_rccl_fetch(smcpp_im *im, double *out, long n) {
    API_BEGIN
    RcclDirect *r = im->rccl;
    if (!r || n != r->n) throw std::runtime_error("smcpp_rccl_fetch: wrong length");
    HIPCHK(hipSetDevice(im->device));
    HIPCHK(hipStreamSynchronize(im->stream));
    HIPCHK(hipMemcpy(out, r->buf.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    API_END
}

int smcpp_set_chunking(smcpp_im *im, int rows_per_chunk, double eps_alpha, double eps_beta) {
    API_BEGIN
    HIPCHK(hipSetDevice(im->device));
    if (eps_alpha > 0) im->eps_f = (float)eps_alpha;
    if (eps_beta > 0) im->eps_b = eps_beta;
    if (rows_per_chunk != im->user_rows_per_chunk) {
        im->user_rows_per_chunk = rows_per_chunk;
        im->warm_valid = false;
        im->make_chunks();
        im->upload_chunk_state();
        im->setup_power();
        im->last_fwd_passes = im->last_bwd_passes = 0;
    }
    API_END
}

int smcpp_last_timing(smcpp_im *im, double out[9]) {
    API_BEGIN im->resolve_timing(); std::memcpy(out, im->timing, sizeof(double) * 9); API_END
}

int smcpp_last_host_timing(smcpp_im *im, double out[4]) {
    API_BEGIN std::memcpy(out, im->host_timing, sizeof(double) * 4); API_END
}

void *smcpp_stream(smcpp_im *im) { return (void *)im->stream; }

// which chain kernels this manager runs: 2 cooperative, 3 cooperative with streamed operands,
// 4 lock-step on the matrix cores (chosen at construction / smcpp_set_chunking from the state count and the input size)
// 5 = scans over the semiseparable structure of T (chains_ss.hpp; the dense kernels named by the other values remain the
// fallback of an E-step whose T has no such structure)
int smcpp_host_chunk_counts(int n_contigs, const long long *cost, const int *rows, long long nslots, long long floor_cost, int *out) {
    API_BEGIN
    if (n_contigs <= 0 || nslots <= 0) throw std::runtime_error("smcpp_host_chunk_counts: empty input");
    const std::vector<long long> c(cost, cost + n_contigs);
    const std::vector<int> r(rows, rows + n_contigs);
    const std::vector<int> ncs = ss_chunk_counts(c, r, nslots, floor_cost);
    std::copy(ncs.begin(), ncs.end(), out);
    API_END
}

int smcpp_chain_mode(smcpp_im *im) { return im ? (im->ss_static ? (im->ss_hybrid ? 6 : 5) : im->chain_mode) : -1; }

// Every SMCPP_* switch is parsed once per process (engine_options.hpp); this re-reads the environment.
void smcpp_reload_options(void) { smcpp_opt::reload(); }

// One JSON object: the switches found in the environment and the plan this manager resolved from them and from its input (chain
// family, states per lane, chunk counts, history passes, arithmetic of the stored passes of the LAST E-step).  Returns the length
// the text needs (without the terminating zero); writes at most cap - 1 characters.
int smcpp_describe(smcpp_im *im, char *buf, int cap) {
    std::string s = "{";
    s += smcpp_opt::describe_options();
    if (im) {
        char t[1024];
        const int fam = im->ss_static ? (im->ss_hybrid ? 6 : 5) : im->chain_mode;
        snprintf(t, sizeof t,
                 ", \"plan\": {\"chain_family\": %d, \"scan_chains\": %s, \"hybrid_rows\": %s, \"states\": %d, \"states_padded\": %d, "
                 "\"states_per_lane\": %d, \"keys\": %d, \"eigen_keys\": %d, \"rows\": %lld, \"positions\": %lld, \"max_span\": %d, "
                 "\"chunks_forward\": %zu, \"chunks_backward\": %zu, \"wavefronts_per_simd\": %d, \"halo_pass\": %s, "
                 "\"light_passes_forward\": %d, \"light_passes_backward\": %d, \"float_scans_in_stored_passes\": %s, "
                 "\"passes_to_certificate\": %d, \"passes_launched\": %d, \"certificate_pass_launched_up_front\": %s, "
                 "\"save_gamma\": %s, \"eigen_free_statistics\": %s, \"per_row_gamma\": \"%s\", \"long_rows_cut\": %s, \"warm_start\": %s, \"host_threads\": %d}",
                 fam, im->ss_static ? "true" : "false", im->ss_hybrid ? "true" : "false", im->M, im->Mp, im->NPL, im->K, im->Ke,
                 (long long)(im->total_rows - im->n_contigs), (long long)im->ss_positions, im->ss_max_span, im->chunks.size(),
                 im->chunks_b.size(), im->ss_wpc, (im->ss_static && im->ss_args.halo) ? "true" : "false", im->ss_light_f, im->ss_light_b,
                 (im->ss_static && im->ss_args.mixed) ? "true" : "false", im->last_ss_passes, im->ss_launched,
                 (opt().on(smcpp_opt::O_SS_CERT_PASS) || im->ss_need_cert_pass) ? "true" : "false", im->save_gamma ? "true" : "false",
                 im->eigfree ? "true" : "false", !im->save_gamma ? "none" : im->eigfree ? "scan steps" : im->gamma_pieces_last ? "eigen-power pieces + scan steps" : "eigensystem", im->split_spans ? "true" : "false",
                 im->warm_start ? "true" : "false", omp_get_max_threads());
        s += t;
    }
    s += "}";
    if (buf && cap > 0) {
        const size_t n = std::min(s.size(), (size_t)cap - 1);
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (int)s.size();
}

// Test hook (tests/test_gpu_ss.py): one position of both scan chains on nvec vectors, out_f = e o (T^T x), out_b = T (e o x);
// x, e and the outputs are [nvec][M].  Returns 2 when T has no semiseparable structure.  float_scans != 0 (M <= 64): the step of
// the stored passes with every scan in float (chains_ss.hpp: ss_x_scan_fwd / ss_x_scan_bwd; the M <= 32 form when M <= 32).
static int ss_debug_apply(int M, const double *T, int nvec, const double *x, const double *e, double *out_f, double *out_b, int float_scans);
int smcpp_debug_ss_apply(int M, const double *T, int nvec, const double *x, const double *e, double *out_f, double *out_b) {
    API_BEGIN
    return ss_debug_apply(M, T, nvec, x, e, out_f, out_b, 0);
    API_END
}
int smcpp_debug_ss_apply_float_scans(int M, const double *T, int nvec, const double *x, const double *e, double *out_f, double *out_b) {
    API_BEGIN
    if (M > 64) throw std::runtime_error("the all-float scans hold one state per lane: M <= 64");
    return ss_debug_apply(M, T, nvec, x, e, out_f, out_b, 1);
    API_END
}
static int ss_debug_apply(int M, const double *T, int nvec, const double *x, const double *e, double *out_f, double *out_b, int float_scans) {
    {
    const int NPL = M > 512 ? 16 : M > 256 ? 8 : (M + 63) / 64, MS = 64 * NPL;
    if (M > 1024) throw std::runtime_error("unsupported number of hidden states");
    std::vector<double> gen;
    double c0 = 0.0;
    if (!ss_generators(M, MS, T, gen, c0)) return 2;
    std::vector<double> hx((size_t)nvec * MS, 0.0), he((size_t)nvec * MS, 0.0);
    for (int v = 0; v < nvec; ++v) {
        std::memcpy(&hx[(size_t)v * MS], x + (size_t)v * M, sizeof(double) * M);
        std::memcpy(&he[(size_t)v * MS], e + (size_t)v * M, sizeof(double) * M);
    }
    DevBuf<double> dg, dx, de, df, db;
    hipStream_t s = nullptr;
    dg.upload(gen, s); dx.upload(hx, s); de.upload(he, s);
    df.alloc(hx.size()); db.alloc(hx.size());
    SsArgs a = SsArgs();
    a.M = M;
    const double *gd = dg.p;
    a.f_dc = gd; a.f_g = gd + MS; a.f_cg = gd + 2 * MS; a.f_b = gd + 3 * MS; a.f_a = gd + 4 * MS; a.f_d = gd + 5 * MS;
    a.b_dc = gd + 6 * MS; a.b_g = gd + 7 * MS; a.b_b = gd + 8 * MS; a.b_a = gd + 9 * MS;
    a.c0 = c0;
    if (float_scans && M <= 32) hipLaunchKernelGGL((k_ss_apply<1, true, true>), dim3(nvec), dim3(64), 0, s, a, (const double *)dx.p, (const double *)de.p, df.p, db.p, nvec);
    else if (float_scans) hipLaunchKernelGGL((k_ss_apply<1, true, false>), dim3(nvec), dim3(64), 0, s, a, (const double *)dx.p, (const double *)de.p, df.p, db.p, nvec);
    else switch (NPL) {
        case 1: hipLaunchKernelGGL(k_ss_apply<1>, dim3(nvec), dim3(64), 0, s, a, (const double *)dx.p, (const double *)de.p, df.p, db.p, nvec); break;
        case 2: hipLaunchKernelGGL(k_ss_apply<2>, dim3(nvec), dim3(64), 0, s, a, (const double *)dx.p, (const double *)de.p, df.p, db.p, nvec); break;
        case 3: hipLaunchKernelGGL(k_ss_apply<3>, dim3(nvec), dim3(64), 0, s, a, (const double *)dx.p, (const double *)de.p, df.p, db.p, nvec); break;
        case 4: hipLaunchKernelGGL(k_ss_apply<4>, dim3(nvec), dim3(64), 0, s, a, (const double *)dx.p, (const double *)de.p, df.p, db.p, nvec); break;
        case 8: hipLaunchKernelGGL(k_ss_apply<8>, dim3(nvec), dim3(64), 0, s, a, (const double *)dx.p, (const double *)de.p, df.p, db.p, nvec); break;
        default: hipLaunchKernelGGL(k_ss_apply<16>, dim3(nvec), dim3(64), 0, s, a, (const double *)dx.p, (const double *)de.p, df.p, db.p, nvec); break;
    }
    HIPCHK(hipGetLastError());
    std::vector<double> hf(hx.size()), hb(hx.size());
    HIPCHK(hipMemcpy(hf.data(), df.p, hf.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hb.data(), db.p, hb.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int v = 0; v < nvec; ++v) {
        std::memcpy(out_f + (size_t)v * M, &hf[(size_t)v * MS], sizeof(double) * M);
        std::memcpy(out_b + (size_t)v * M, &hb[(size_t)v * MS], sizeof(double) * M);
    }
    }
    return 0;
}


int smcpp_device(smcpp_im *im) { return im ? im->device : -1; }
int smcpp_set_debug(smcpp_im *im, int on) { API_BEGIN im->debug = on != 0; API_END }
int smcpp_get_debug(smcpp_im *im) { return im && im->debug ? 1 : 0; }

void smcpp_set_num_threads(int k) { if (k > 0) omp_set_num_threads(k); }

int smcpp_host_set_csfs_direct(int on) {
    const int prev = smcpp_host::csfs_direct_flag();
    smcpp_host::csfs_direct_flag() = on != 0;
    return prev;
}

}  // extern "C"
