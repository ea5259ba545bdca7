// engine_base.hpp - part of the ONE translation unit engine.hip (included there, in order; not a standalone header):
// logging, pinned arena, device buffers, the device route of the cold preparation (DevPrep, TwoPopDevCsfs).
using namespace smcpp_dev;

static thread_local std::string g_err;

// Logger::logger_cb (src/common.cpp:35-40, _smcpp.pxd:26): messages of the engine go to the binding's callback
typedef void (*smcpp_logger_cb_t)(const char *name, const char *level, const char *message);
static smcpp_logger_cb_t g_logger_cb = nullptr;
static void log_msg(const char *level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static void log_msg(const char *level, const char *fmt, ...) {
    if (!g_logger_cb) return;
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_logger_cb("engine", level, buf);
}

// SMCPP_HOST_TRACE=1: microsecond stamps of the host phase of an E-step on stderr (diagnostics; no effect on the results)
struct HostTrace {
    bool on;
    std::chrono::steady_clock::time_point t;
    HostTrace() { on = opt().i(smcpp_opt::O_HOST_TRACE, 0) > 0; if (on) t = std::chrono::steady_clock::now(); }
    void mark(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[host-trace] %-28s %7.1f us\n", what, std::chrono::duration<double, std::micro>(n - t).count());
        t = n;
    }
};

// libomp keeps its workers spinning for 200 ms after a parallel region by default; that steals the cores the HIP
// runtime's own threads need between the short host-side parallel loops of an E-step.
extern "C" void kmp_set_blocktime(int) __attribute__((weak));
namespace {
struct OmpInit {
    OmpInit() {
        if (kmp_set_blocktime) kmp_set_blocktime(opt().i(smcpp_opt::O_OMP_BLOCKTIME, 0));
    }
} g_omp_init;
}

#define HIPCHK(x)                                                                                              \
    do {                                                                                                       \
        hipError_t e_ = (x);                                                                                   \
        if (e_ != hipSuccess)                                                                                  \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " + __FILE__ +  \
                                     ":" + std::to_string(__LINE__));                                          \
    } while (0)

namespace {

// Pinned host staging for the per-E-step parameter upload: pageable hipMemcpyAsync is staged synchronously by the
// runtime (~10 us per call, ~20 calls per E-step); from pinned memory the copies are plain DMA enqueues and the host
// does not have to wait for them before launching the chains.  Reset at the start of every upload; the previous
// E-step has synchronised its stream by then.
struct PinnedArena {
    char *base = nullptr;
    size_t cap = 0, off = 0;
    void reset(size_t need) {
        off = 0;
        if (need <= cap) return;
        if (base) (void)hipHostFree(base);
        cap = need + need / 4 + 4096;
        HIPCHK(hipHostMalloc((void **)&base, cap, hipHostMallocDefault));
    }
    void *take(size_t bytes) {
        const size_t o = (off + 255) & ~(size_t)255;
        if (o + bytes > cap) throw std::runtime_error("internal: pinned staging arena too small");
        off = o + bytes;
        return base + o;
    }
    ~PinnedArena() { if (base) (void)hipHostFree(base); }
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool borrowed = false;      // p points into the parameter arena (see ParamArena): never freed here
    void alloc(size_t count, int line = __builtin_LINE(), const char *file = __builtin_FILE()) {
        if (count <= n && p && !borrowed) return;
        free();
        n = count;
        if (count) {
            HIPCHK(hipMalloc((void **)&p, count * sizeof(T)));
            smcpp_opt::poison(p, count * sizeof(T), line, file);
        }
    }
    void free() {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        n = 0;
        borrowed = false;
    }
    // place this buffer at byte offset `off` of the parameter arena and stage its contents at the same offset of the
    // pinned mirror; the caller issues ONE copy for the whole arena afterwards
    void place(const std::vector<T> &h, char *dev_base, char *host_base, size_t &off) {
        if (p && !borrowed) (void)hipFree(p);
        off = (off + 255) & ~(size_t)255;
        p = reinterpret_cast<T *>(dev_base + off);
        n = h.size();
        borrowed = true;
        if (!h.empty()) std::memcpy(host_base + off, h.data(), h.size() * sizeof(T));
        off += h.size() * sizeof(T);
    }
    void upload(const std::vector<T> &h, hipStream_t s, int line = __builtin_LINE(), const char *file = __builtin_FILE()) {
        alloc(h.size(), line, file);
        if (!h.empty()) HIPCHK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void upload_staged(const std::vector<T> &h, PinnedArena &ar, hipStream_t s, int line = __builtin_LINE(), const char *file = __builtin_FILE()) {
        alloc(h.size(), line, file);
        if (h.empty()) return;
        void *q = ar.take(h.size() * sizeof(T));
        std::memcpy(q, h.data(), h.size() * sizeof(T));
        HIPCHK(hipMemcpyAsync(p, q, h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void zero(hipStream_t s) {
        if (n) HIPCHK(hipMemsetAsync(p, 0, n * sizeof(T), s));
    }
    ~DevBuf() { free(); }
};

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

struct Group { int span, kid, eig; };
constexpr int ROWDESC_PAD = 256;

}  // namespace


// ---------------------------------------------------------------------------------------------------------------
// Cold preparation on the device (prep_dev.hpp): the host part of one call is O(pieces): the rate function with the
// hidden states inserted (RateFunctionT), packed with its derivative planes into one pinned block; everything that is
// O(states x n^2 x directions) - conditioned SFS, incorporate_theta, emission table - runs in two kernels.
// `emulate`: the same phases run serially on host vectors (CPU tests), nothing touches a device.
// ---------------------------------------------------------------------------------------------------------------
struct DevPrep {
    int n = 0, M = 0, Kk = 0, Klocal = 0, Mp = 0, MS = 0;
    bool emulate = false, keys_ready = false, static_ready = false;
    // static: n-only tables | bin weights, key tables
    std::vector<double> h_sd;
    std::vector<int> h_si;
    DevBuf<double> d_sd;
    DevBuf<int> d_si;
    size_t off_bw = 0, off_kind = 0, off_boff = 0, off_bidx = 0, off_local = 0, off_slot = 0, off_maxspan = 0;
    // per call
    PinnedArena stage;
    std::vector<char> h_in;            // emulate: the packed block
    char *d_in = nullptr;
    size_t in_cap = 0;
    DevBuf<double> d_tab, d_sfs_v, d_sfs_d, d_Eg_v, d_Eg_d, d_El, d_Es;
    std::vector<double> e_tab, e_sfs_v, e_sfs_d, e_Eg_v, e_Eg_d;     // emulate
    int *h_flags = nullptr, *d_flags_view = nullptr;
    int e_flags[4] = {0, 0, 0, 0};
    int last_nder = 0;
    hipEvent_t ev_done = nullptr;        // recorded behind the last preparation's kernels: the staging block and the flag words
    bool in_flight = false;              // are rewritten only after it has completed
    ~DevPrep() {
        if (d_in) (void)hipFree(d_in);
        if (h_flags) (void)hipHostFree(h_flags);
        if (ev_done) (void)hipEventDestroy(ev_done);
    }
    typedef smcpp_dev::DN<4> SD;             // scalar of the derivative kernels: value + four directions per thread
    // n: the CSFS scratch of one hidden state must fit LDS; K (pieces after the hidden states were inserted): so must the
    // 2 K scan terms of k_prep_tables (80 B per piece with four directions per scalar)
    static bool supported(int n, int K = 0) {
        return n >= 1 && smcpp_dev::CsfsScratch<SD>::count(n) * sizeof(SD) <= 150 * 1024 && (size_t)2 * K * sizeof(SD) <= 150 * 1024;
    }

    void set_static(const smcpp_host::CsfsTables &t) {
        n = t.n;
        h_sd.clear();
        for (const smcpp_host::DMat *m : {&t.X0, &t.X2, &t.M0, &t.M1, &t.Uinv_mp0, &t.Uinv_mp2}) h_sd.insert(h_sd.end(), m->d.begin(), m->d.end());
        off_bw = h_sd.size();
        static_ready = true;
        keys_ready = false;
    }
    // keys [Kk][3]; local[k] / slot[k] / maxspan[k] may be empty (identity / none)
    void set_keys(const smcpp_host::OnePopPrep &hp, const std::vector<int> &keys, int Kk_, const std::vector<int> &local,
                  const std::vector<int> &slot, const std::vector<int> &maxspan, int Klocal_, int M_, int Mp_, int MS_) {
        Kk = Kk_; Klocal = Klocal_; M = M_; Mp = Mp_; MS = MS_;
        h_sd.resize(off_bw);
        std::vector<int> kind(Kk), boff(Kk + 1, 0), bidx;
        for (int k = 0; k < Kk; ++k) {
            const smcpp_host::OnePopPrep::Key bk{keys[3 * k], keys[3 * k + 1], keys[3 * k + 2]};
            kind[k] = smcpp_host::OnePopPrep::key_kind(bk);
            if (kind[k] == 0)
                for (const auto &pr : hp.bins_of(bk)) { bidx.push_back(pr.first); h_sd.push_back(pr.second); }
            boff[k + 1] = (int)bidx.size();
        }
        h_si.clear();
        auto put = [&](const std::vector<int> &v, size_t &off) { off = h_si.size(); h_si.insert(h_si.end(), v.begin(), v.end()); };
        std::vector<int> loc(local), sl(slot), ms(maxspan);
        if (loc.empty()) { loc.resize(Kk); for (int k = 0; k < Kk; ++k) loc[k] = k; }
        if (sl.empty()) sl = loc;
        if (ms.empty()) ms.assign(Kk, 1);
        put(kind, off_kind); put(boff, off_boff); put(bidx, off_bidx); put(loc, off_local); put(sl, off_slot); put(ms, off_maxspan);
        if (!emulate) {
            d_sd.alloc(h_sd.size()); d_si.alloc(std::max<size_t>(1, h_si.size()));
            HIPCHK(hipMemcpy(d_sd.p, h_sd.data(), h_sd.size() * sizeof(double), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(d_si.p, h_si.data(), h_si.size() * sizeof(int), hipMemcpyHostToDevice));
            // the tables the kernels write rows of: allocated and cleared once (padding stays zero)
            d_Eg_v.alloc((size_t)Kk * M);
            d_El.alloc((size_t)std::max(1, Klocal) * Mp); d_Es.alloc((size_t)std::max(1, Klocal) * std::max(1, MS));
            HIPCHK(hipMemset(d_El.p, 0, d_El.n * sizeof(double)));
            HIPCHK(hipMemset(d_Es.p, 0, d_Es.n * sizeof(double)));
            d_sfs_v.alloc((size_t)M * 3 * (n + 1));
            if (!h_flags) {
                HIPCHK(hipHostMalloc((void **)&h_flags, 64, hipHostMallocCoherent | hipHostMallocMapped));
                HIPCHK(hipHostGetDevicePointer((void **)&d_flags_view, h_flags, 0));
            }
        } else {
            e_Eg_v.assign((size_t)Kk * M, 0.0);
            e_sfs_v.assign((size_t)M * 3 * (n + 1), 0.0);
        }
        keys_ready = true;
    }
    smcpp_dev::PrepStatic ps_view() const {
        const double *sd = emulate ? h_sd.data() : d_sd.p;
        const int *si = emulate ? h_si.data() : d_si.p;
        smcpp_dev::PrepStatic ps;
        const size_t a = (size_t)n * (n + 1), b = (size_t)(n + 1) * n, c = (size_t)(n + 1) * (n + 1);
        ps.X0 = sd; ps.X2 = sd + a; ps.M0 = sd + 2 * a; ps.M1 = sd + 2 * a + b; ps.U0 = sd + 2 * a + b + c; ps.U2 = sd + 2 * a + 2 * b + c;
        ps.bw = sd + off_bw;
        ps.Kk = Kk;
        ps.kind = si + off_kind; ps.boff = si + off_boff; ps.bidx = si + off_bidx; ps.local = si + off_local; ps.slot = si + off_slot;
        ps.maxspan = si + off_maxspan;
        return ps;
    }

    template <typename S> static double dpart(const S &x, int d);

    // Pack the rate function and launch.  HS = double or smcpp_host::dual (nder directions).  Returns after the ENQUEUE.
    template <typename HS>
    void run(const smcpp_host::RateFunctionT<HS> &eta, const std::vector<HS> &act, double theta, double alpha, int nder,
             hipStream_t s) {
        if (!static_ready || !keys_ready) throw std::runtime_error("internal: device preparation without its tables");
        const int K = eta.K;
        last_nder = nder;
        // ---- pack: doubles ts [K+1] | ada_v [K] | R_v [K+1] | act_v [M] | ada_d [nder][K] | R_d [nder][K+1] | act_d [nder][M]; ints hsi [M+1]
        const size_t ndbl = (size_t)(K + 1) + K + (K + 1) + M + (size_t)nder * (K + (K + 1) + M);
        const size_t bytes = ndbl * sizeof(double) + (size_t)(M + 1) * sizeof(int) + 64;
        char *hb;
        if (emulate) { h_in.resize(bytes); hb = h_in.data(); }
        else {
            // an earlier preparation may still be reading the staging block / raising flags (the Jacobian getters return after the
            // enqueue): wait for it before either is rewritten
            if (in_flight) { HIPCHK(hipEventSynchronize(ev_done)); in_flight = false; }
            stage.reset(bytes);
            hb = stage.base;
            if (bytes > in_cap) {
                if (d_in) (void)hipFree(d_in);
                in_cap = bytes + bytes / 2;
                HIPCHK(hipMalloc((void **)&d_in, in_cap));
                smcpp_opt::poison(d_in, in_cap, __LINE__, __FILE__);
            }
        }
        double *hd = reinterpret_cast<double *>(hb);
        size_t o = 0;
        const size_t o_ts = o; for (int i = 0; i <= K; ++i) hd[o++] = eta.ts[i];
        const size_t o_ada = o; for (int i = 0; i < K; ++i) hd[o++] = smcpp_host::sval(eta.ada[i]);
        const size_t o_R = o; for (int i = 0; i <= K; ++i) hd[o++] = smcpp_host::sval(eta.Rrng[i]);
        const size_t o_act = o; for (int i = 0; i < M; ++i) hd[o++] = smcpp_host::sval(act[i]);
        const size_t o_adad = o; for (int d = 0; d < nder; ++d) for (int i = 0; i < K; ++i) hd[o++] = dpart(eta.ada[i], d);
        const size_t o_Rd = o; for (int d = 0; d < nder; ++d) for (int i = 0; i <= K; ++i) hd[o++] = dpart(eta.Rrng[i], d);
        const size_t o_actd = o; for (int d = 0; d < nder; ++d) for (int i = 0; i < M; ++i) hd[o++] = dpart(act[i], d);
        int *hi = reinterpret_cast<int *>(hd + o);
        for (int i = 0; i <= M; ++i) hi[i] = eta.hs_indices[i];
        const char *base = emulate ? hb : d_in;
        const double *bd = reinterpret_cast<const double *>(base);
        smcpp_dev::PrepModel pm;
        pm.K = K; pm.n = n; pm.M = M; pm.nder = nder; pm.theta = theta; pm.alpha = alpha;
        pm.ts = bd + o_ts; pm.ada_v = bd + o_ada; pm.R_v = bd + o_R; pm.act_v = bd + o_act;
        pm.ada_d = bd + o_adad; pm.R_d = bd + o_Rd; pm.act_d = bd + o_actd;
        pm.hsi = reinterpret_cast<const int *>(bd + o);
        const smcpp_dev::PrepStatic ps = ps_view();
        const int C = 3 * (n + 1);
        const int ng = nder > 0 ? (nder + 3) / 4 : 1;                               // direction groups (four directions per scalar)
        const size_t per = smcpp_dev::Tables<double>::per_group(n, K);             // table entries per group
        const size_t ssz = nder > 0 ? sizeof(SD) / sizeof(double) : 1;             // doubles per scalar
        smcpp_dev::PrepOut po;
        po.Mp = Mp; po.MS = MS;
        if (emulate) {
            e_tab.assign(per * ng * ssz, 0.0);
            if (nder) { e_sfs_d.assign((size_t)nder * M * C, 0.0); e_Eg_d.assign((size_t)nder * Kk * M, 0.0); }
            po.sfs_v = e_sfs_v.data(); po.sfs_d = nder ? e_sfs_d.data() : nullptr; po.Eg_v = e_Eg_v.data(); po.Eg_d = nder ? e_Eg_d.data() : nullptr;
            e_flags[0] = e_flags[1] = e_flags[2] = 0;
            po.flags = e_flags;
            if (nder) {
                smcpp_dev::Tables<SD> tb;
                tb.carve(reinterpret_cast<SD *>(e_tab.data()), n, K, ng);
                smcpp_dev::emulate_tables(pm, tb);
                smcpp_dev::emulate_csfs(pm, ps, po, tb);
            } else {
                smcpp_dev::Tables<double> tb;
                tb.carve(e_tab.data(), n, K, 1);
                smcpp_dev::emulate_tables(pm, tb);
                smcpp_dev::emulate_csfs(pm, ps, po, tb);
            }
            return;
        }
        d_tab.alloc(per * ng * ssz);
        if (nder) { d_sfs_d.alloc((size_t)nder * M * C); d_Eg_d.alloc((size_t)nder * Kk * M); }
        po.sfs_v = d_sfs_v.p; po.sfs_d = nder ? d_sfs_d.p : nullptr; po.Eg_v = d_Eg_v.p; po.Eg_d = nder ? d_Eg_d.p : nullptr;
        po.El_v = d_El.p; po.Es_v = MS > 0 ? d_Es.p : nullptr;
        h_flags[0] = h_flags[1] = h_flags[2] = 0;
        po.flags = d_flags_view;
        HIPCHK(hipMemcpyAsync(d_in, hb, bytes, hipMemcpyHostToDevice, s));
        const int pairs = (n + 1) * n;
        const int nt = std::min(512, std::max(64 * ceil_div(3 * n + 2, 64), 64 * ceil_div(pairs, 64)));
        const int ntt = std::min(256, 64 * ceil_div(K, 64));
        if (nder) {
            typedef SD S;
            smcpp_dev::Tables<S> tb;
            tb.carve(reinterpret_cast<S *>(d_tab.p), n, K, ng);
            const size_t lds = smcpp_dev::CsfsScratch<S>::count(n) * sizeof(S);
            static bool once = false;
            if (!once) {
                HIPCHK(hipFuncSetAttribute((const void *)smcpp_dev::k_prep_csfs<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIPCHK(hipFuncSetAttribute((const void *)smcpp_dev::k_prep_tables<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                once = true;
            }
            hipLaunchKernelGGL(smcpp_dev::k_prep_tables<S>, dim3(ng, 2 * n + 1), dim3(ntt), (size_t)2 * K * sizeof(S), s, pm, tb);
            hipLaunchKernelGGL(smcpp_dev::k_prep_csfs<S>, dim3(M, ng), dim3(nt), lds, s, pm, ps, po, tb);
        } else {
            typedef double S;
            smcpp_dev::Tables<S> tb;
            tb.carve(d_tab.p, n, K, 1);
            const size_t lds = smcpp_dev::CsfsScratch<S>::count(n) * sizeof(S);
            static bool once = false;
            if (!once) {
                HIPCHK(hipFuncSetAttribute((const void *)smcpp_dev::k_prep_csfs<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIPCHK(hipFuncSetAttribute((const void *)smcpp_dev::k_prep_tables<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                once = true;
            }
            hipLaunchKernelGGL(smcpp_dev::k_prep_tables<S>, dim3(1, 2 * n + 1), dim3(ntt), (size_t)2 * K * sizeof(S), s, pm, tb);
            hipLaunchKernelGGL(smcpp_dev::k_prep_csfs<S>, dim3(M, 1), dim3(nt), lds, s, pm, ps, po, tb);
        }
        HIPCHK(hipGetLastError());
        // (only ever waited for by the host before it rewrites the staging block: no data visibility hangs on it - no system-scope fence)
        if (!ev_done) HIPCHK(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming | hipEventDisableSystemFence));
        HIPCHK(hipEventRecord(ev_done, s));
        in_flight = true;
    }
    const int *flags() const { return emulate ? e_flags : h_flags; }
    // Results to the host (after the stream has drained): E [Kk][M], dE [Kk*M][nder], sfs [M][C], dsfs [M*C][nder]
    void fetch(std::vector<double> &Ev, std::vector<double> &dEv, std::vector<double> &sfs, std::vector<double> &dsfs) {
        const int C = 3 * (n + 1), nder = last_nder;
        std::vector<double> pl;
        auto get = [&](const DevBuf<double> &d, const std::vector<double> &e, size_t cnt, std::vector<double> &out) {
            out.resize(cnt);
            if (emulate) std::memcpy(out.data(), e.data(), cnt * sizeof(double));
            else HIPCHK(hipMemcpy(out.data(), d.p, cnt * sizeof(double), hipMemcpyDeviceToHost));
        };
        get(d_Eg_v, e_Eg_v, (size_t)Kk * M, Ev);
        get(d_sfs_v, e_sfs_v, (size_t)M * C, sfs);
        dEv.clear(); dsfs.clear();
        if (nder) {
            get(d_Eg_d, e_Eg_d, (size_t)nder * Kk * M, pl);
            dEv.resize(pl.size());
            const size_t sz = (size_t)Kk * M;
            for (int d = 0; d < nder; ++d) for (size_t i = 0; i < sz; ++i) dEv[i * nder + d] = pl[(size_t)d * sz + i];
            get(d_sfs_d, e_sfs_d, (size_t)nder * M * C, pl);
            dsfs.resize(pl.size());
            const size_t s2 = (size_t)M * C;
            for (int d = 0; d < nder; ++d) for (size_t i = 0; i < s2; ++i) dsfs[i * nder + d] = pl[(size_t)d * s2 + i];
        }
    }
    void check_flags() const {
        const int *f = flags();
        if (f[1]) throw std::runtime_error("csfs is not a probability distribution");
        if (f[0]) throw std::runtime_error("probability vector not in [0, 1]");
    }
};
template <> inline double DevPrep::dpart<double>(const double &, int) { return 0.0; }
template <> inline double DevPrep::dpart<smcpp_host::dual>(const smcpp_host::dual &x, int d) { return x.d[d]; }

// The two batched conditioned-SFS problems of the two-population preparation on the device (round 5; jcsfs.hpp: CsfsBatchDevice):
// every interval below the split under the truncated model (n1 lineages) and every interval above it under the shifted model
// (n1 + n2) - k_prep_tables + k_prep_csfs_raw on the manager's stream, the states' tables copied back to pinned memory - while
// the host forms the state-independent pieces of the joint CSFS.  Values only; the Jacobian route stays on the host.
struct TwoPopDevCsfs : smcpp_host::CsfsBatchDevice {
    struct Inst {
        int n = -1, M = 0, C = 0;
        DevBuf<double> d_sd, d_tab, d_raw;
        PinnedArena stage, res;
        char *d_in = nullptr;
        size_t in_cap = 0;
        double *h_raw = nullptr;
        hipEvent_t ev = nullptr;
        bool in_flight = false;
        ~Inst() { if (d_in) (void)hipFree(d_in); if (ev) (void)hipEventDestroy(ev); }
    } inst[2];
    int device = 0;
    hipStream_t stream = nullptr;
    static bool fits(int n, int K) {
        return n >= 1 && smcpp_dev::CsfsScratch<double>::count(n) * sizeof(double) <= 150 * 1024 && (size_t)2 * K * sizeof(double) <= 150 * 1024;
    }
    bool launch(int which, const smcpp_host::RateFunctionT<double> &eta, int n) override {
        const int K = eta.K, M = (int)eta.hidden_states.size() - 1;
        if (M <= 0 || !fits(n, K)) return false;
        HIPCHK(hipSetDevice(device));
        Inst &I = inst[which];
        if (I.in_flight) { HIPCHK(hipEventSynchronize(I.ev)); I.in_flight = false; }
        if (I.n != n) {
            const smcpp_host::CsfsTables &t = *smcpp_host::csfs_tables(n);
            std::vector<double> sd;
            for (const smcpp_host::DMat *m : {&t.X0, &t.X2, &t.M0, &t.M1, &t.Uinv_mp0, &t.Uinv_mp2}) sd.insert(sd.end(), m->d.begin(), m->d.end());
            I.d_sd.alloc(sd.size());
            HIPCHK(hipMemcpy(I.d_sd.p, sd.data(), sd.size() * sizeof(double), hipMemcpyHostToDevice));
            I.n = n;
        }
        I.M = M; I.C = 3 * (n + 1);
        // pack: doubles ts [K+1] | ada [K] | R [K+1]; ints hsi [M+1]
        const size_t ndbl = (size_t)(K + 1) + K + (K + 1);
        const size_t bytes = ndbl * sizeof(double) + (size_t)(M + 1) * sizeof(int) + 64;
        I.stage.reset(bytes);
        char *hb = I.stage.base;
        if (bytes > I.in_cap) {
            if (I.d_in) (void)hipFree(I.d_in);
            I.in_cap = bytes + bytes / 2;
            HIPCHK(hipMalloc((void **)&I.d_in, I.in_cap));
            smcpp_opt::poison(I.d_in, I.in_cap, __LINE__, __FILE__);
        }
        double *hd = reinterpret_cast<double *>(hb);
        size_t o = 0;
        const size_t o_ts = o; for (int i = 0; i <= K; ++i) hd[o++] = eta.ts[i];
        const size_t o_ada = o; for (int i = 0; i < K; ++i) hd[o++] = eta.ada[i];
        const size_t o_R = o; for (int i = 0; i <= K; ++i) hd[o++] = eta.Rrng[i];
        int *hi = reinterpret_cast<int *>(hd + o);
        for (int i = 0; i <= M; ++i) hi[i] = eta.hs_indices[i];
        const double *bd = reinterpret_cast<const double *>(I.d_in);
        smcpp_dev::PrepModel pm;
        pm.K = K; pm.n = n; pm.M = M; pm.nder = 0;
        pm.ts = bd + o_ts; pm.ada_v = bd + o_ada; pm.R_v = bd + o_R;
        pm.hsi = reinterpret_cast<const int *>(bd + o);
        smcpp_dev::PrepStatic ps;
        {
            const size_t a = (size_t)n * (n + 1), b = (size_t)(n + 1) * n, c = (size_t)(n + 1) * (n + 1);
            const double *sd = I.d_sd.p;
            ps.X0 = sd; ps.X2 = sd + a; ps.M0 = sd + 2 * a; ps.M1 = sd + 2 * a + b; ps.U0 = sd + 2 * a + b + c; ps.U2 = sd + 2 * a + 2 * b + c;
        }
        I.d_tab.alloc(smcpp_dev::Tables<double>::per_group(n, K));
        smcpp_dev::Tables<double> tb;
        tb.carve(I.d_tab.p, n, K, 1);
        I.d_raw.alloc((size_t)M * I.C);
        I.res.reset((size_t)M * I.C * sizeof(double));
        I.h_raw = reinterpret_cast<double *>(I.res.base);
        HIPCHK(hipMemcpyAsync(I.d_in, hb, bytes, hipMemcpyHostToDevice, stream));
        const int pairs = (n + 1) * n;
        const int nt = std::min(512, std::max(64 * ceil_div(3 * n + 2, 64), 64 * ceil_div(pairs, 64)));
        const int ntt = std::min(256, 64 * ceil_div(K, 64));
        static bool once = false;
        if (!once) {
            HIPCHK(hipFuncSetAttribute((const void *)smcpp_dev::k_prep_csfs_raw, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHK(hipFuncSetAttribute((const void *)smcpp_dev::k_prep_tables<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            once = true;
        }
        hipLaunchKernelGGL(smcpp_dev::k_prep_tables<double>, dim3(1, 2 * n + 1), dim3(ntt), (size_t)2 * K * sizeof(double), stream, pm, tb);
        hipLaunchKernelGGL(smcpp_dev::k_prep_csfs_raw, dim3(M), dim3(nt), smcpp_dev::CsfsScratch<double>::count(n) * sizeof(double), stream, pm, ps, tb, I.d_raw.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(I.h_raw, I.d_raw.p, (size_t)M * I.C * sizeof(double), hipMemcpyDeviceToHost, stream));
        if (!I.ev) HIPCHK(hipEventCreateWithFlags(&I.ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(I.ev, stream));
        I.in_flight = true;
        return true;
    }
    void collect(int which, std::vector<std::vector<double>> &out) override {
        Inst &I = inst[which];
        if (!I.in_flight) throw std::runtime_error("internal: collect without a launched batch");
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipEventSynchronize(I.ev));
        I.in_flight = false;
        out.assign(I.M, std::vector<double>());
        for (int m = 0; m < I.M; ++m) out[m].assign(I.h_raw + (size_t)m * I.C, I.h_raw + (size_t)(m + 1) * I.C);
    }
};

