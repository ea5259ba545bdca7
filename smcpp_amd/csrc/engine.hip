// Host orchestration + C ABI of the MI355X-native SMC++ E-step engine.
//
// Mirrors the reference's InferenceManager (include/inference_manager.h, src/inference_manager.cpp):
//   ctor / map_obs / fill_targets / populate_emission_probs  -> Engine::Engine   (21-54, 180-211, 232-254)
//   Estep                                                    -> Engine::estep    (108-114)
//   TransitionBundle::update(T, true)                        -> Engine::host_prep (src/transition_bundle.cpp:3-61)
//   loglik / Q / getters                                     -> smcpp_loglik / smcpp_q / smcpp_get_*  (116-177)
// The hot path (HMM::Estep, src/hmm.cpp:45-153) runs in the HIP kernels of kernels.hpp.  There is no CPU fallback:
// every compute entry point fails with an error if no HIP device is usable.
#include <hip/hip_runtime.h>
#include <omp.h>
#include <dlfcn.h>
#include <mutex>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/smcpp_engine.h"
#include "kernels.hpp"
#include "chains2.hpp"
#include "chains_lock.hpp"
#include "chains_ss.hpp"
#include "nonsym_eig.hpp"
#include "nonsym_eig_team.hpp"
#include "prep.hpp"
#include "prep_dev.hpp"
#include "jcsfs.hpp"

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
    HostTrace() { static const bool e = getenv("SMCPP_HOST_TRACE") && atoi(getenv("SMCPP_HOST_TRACE")) > 0; on = e; if (on) t = std::chrono::steady_clock::now(); }
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
        const char *e = getenv("SMCPP_OMP_BLOCKTIME");
        if (kmp_set_blocktime) kmp_set_blocktime(e ? atoi(e) : 0);
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
    void alloc(size_t count) {
        if (count <= n && p && !borrowed) return;
        free();
        n = count;
        if (count) HIPCHK(hipMalloc((void **)&p, count * sizeof(T)));
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
    void upload(const std::vector<T> &h, hipStream_t s) {
        alloc(h.size());
        if (!h.empty()) HIPCHK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void upload_staged(const std::vector<T> &h, PinnedArena &ar, hipStream_t s) {
        alloc(h.size());
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
        if (!ev_done) HIPCHK(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
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
    // generation-2 eigen statistics (M <= 64): slabs over the sorted eigen rows of a (contig, eigen key) that MIX span groups
    std::vector<Slab> slabs_ek;
    std::vector<int> ek_slab_off, epos_gid;
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
    int ss_wg_waves = 4;                   // wavefronts per workgroup of k_chain_ss (hybrid with two per SIMD: 8, one table copy)
    int ss_launched = 0, last_ss_passes = 0;
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
    NPL = M > 256 ? 8 : (M + 63) / 64;
    NT = Mp / 16;
    if (M > 512) throw std::runtime_error("M > 512 hidden states is not supported by this build");
    n_contigs = n_contigs_;
    Ls.assign(Ls_, Ls_ + n_contigs);
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
    if (const char *d = getenv("SMCPP_DUAL_STREAM")) dual_stream = atoi(d);
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
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
    static const long long v = getenv("SMCPP_LOCK_MIN_ROWS") ? atoll(getenv("SMCPP_LOCK_MIN_ROWS")) : -1;
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
        const char *m = getenv("SMCPP_CHAIN");
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
            const char *se = getenv("SMCPP_SS");
            const bool ss_ok = !(se && atoi(se) == 0) && (!m || !strcmp(m, "ss")) && Mp <= 512;
            ss_static = ss_ok && ss_max_span <= 512;
            ss_hybrid = false; ss_hyb_th = 0x7fffffff;
            if (ss_ok && !ss_static) {
                // longer spans: the hybrid form, when one state per lane holds the vector and the eigenvector tables of every eigen
                // key fit LDS beside the emission vectors (SMCPP_HYBRID=0: the dense kernels)
                const char *hy = getenv("SMCPP_HYBRID");
                const size_t tab = (size_t)Ke * 4 * Mp * (Mp + 1) * sizeof(double);
                ss_dirsplit = false;
                if (!(hy && atoi(hy) == 0) && Mp <= 64 && Ke >= 1 && Ke <= 4 && tab <= 120 * 1024) {
                    ss_static = ss_hybrid = true;
                    ss_hyb_th = getenv("SMCPP_HYB_TH") ? std::max(1, atoi(getenv("SMCPP_HYB_TH"))) : 6;
                } else if (!(hy && atoi(hy) == 0) && Mp <= 64 && Ke >= 1 && Ke <= 4 && tab / 2 <= 136 * 1024) {
                    // (round 4) M > 32: four 33 KB tables per eigen key do not fit, the two a DIRECTION needs do - every workgroup
                    // runs one direction (the task table keeps them apart) and stages that direction's pair
                    ss_static = ss_hybrid = ss_dirsplit = true;
                    ss_hyb_th = getenv("SMCPP_HYB_TH") ? std::max(1, atoi(getenv("SMCPP_HYB_TH"))) : 6;
                } else if (!(hy && atoi(hy) == 0) && Mp > 32 && Mp <= 64 && Ke >= 1 && Ke <= 4 && tab / 2 / Ke <= 136 * 1024) {
                    // (round 5) ... and with three or four eigen keys at M > 32 not even those: the most frequent keys keep their pair
                    // in LDS, a COLD key's table rows are read from L2 on the rows that need them (chains_ss.hpp: ss_eig_matvec_cold)
                    ss_static = ss_hybrid = ss_dirsplit = true;
                    ss_hyb_th = getenv("SMCPP_HYB_TH") ? std::max(1, atoi(getenv("SMCPP_HYB_TH"))) : 6;
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
        const char *b = getenv("SMCPP_COOP_BPC");
        if (b && atoi(b) > 0) coop_bpc = atoi(b);
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
    if (ss_static && user_rows_per_chunk <= 0 && !getenv("SMCPP_ROWS_PER_CHUNK")) {
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
        ss_wpc = getenv("SMCPP_SS_WPC") ? std::max(1, std::min(4, atoi(getenv("SMCPP_SS_WPC"))))
                                        : (int)std::max<long long>(1, std::min<long long>(3, total_bins / (simds * 9000)));
        // hybrid rows are bound by instruction and LDS LATENCY (a dependent chain of ~200 instructions per row): a second wavefront
        // per SIMD fills the gaps from ~2 000 cost units per chunk on (posterior workload: 1.79 -> 1.35 ms of chains; a third one
        // needs an extra pass: 1.80); the eight wavefronts form ONE workgroup so that the CU holds one copy of the tables
        if (ss_hybrid && !getenv("SMCPP_SS_WPC")) ss_wpc = (int)std::max<long long>(1, std::min<long long>(2, total_bins / (simds * 2000)));
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
        ss_halo = !ss_hybrid && (getenv("SMCPP_SS_HALO") ? atoi(getenv("SMCPP_SS_HALO")) != 0 : NPL >= 2);
        auto env_ll = [](const char *nm, long long dflt) { const char *e = getenv(nm); return e ? atoll(e) : dflt; };
        const long long hlf = ss_halo ? env_ll("SMCPP_HALO_LF", 2800) : 0, hdf = ss_halo ? env_ll("SMCPP_HALO_DF", 800) : 0,
                        hlb = ss_halo ? env_ll("SMCPP_HALO_LB", 3900) : 0, hdb = ss_halo ? env_ll("SMCPP_HALO_DB", 1100) : 0;
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
            }
            const double share = getenv("SMCPP_SS_FWD_SHARE") ? atof(getenv("SMCPP_SS_FWD_SHARE")) : dflt_share;
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
        const char *envv = getenv("SMCPP_ROWS_PER_CHUNK");
        if (envv) lc = atoi(envv);
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
    const long long target = std::max<long long>(256, std::min<long long>(8192, (256ll << 20) / part_bytes));
    // (at least SMCPP_SLAB_ROWS rows per slab, default 128: every slab costs an Mp x Mp partial written and read back - 128 MB of
    // traffic per headline E-step with 64-row slabs -, but a slab is walked by ONE wavefront, and below ~1000 slabs the rank
    // kernels leave SIMDs idle: 64 .. 192 rows measured: 633 / 641 / 666 / 665 headline evals per second)
    static const int slab_rows = getenv("SMCPP_SLAB_ROWS") ? std::max(16, atoi(getenv("SMCPP_SLAB_ROWS"))) : 128;
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
    // generation-2 eigen slabs: the sorted eigen rows of every (contig, eigen key) cut into S_EG-row pieces regardless of the span
    // groups; padded like slabs_eg so that a workgroup of four never mixes keys
    slabs_ek.clear(); epos_gid.clear();
    ek_slab_off.assign((size_t)n_contigs * Ke + 1, 0);
    epos_gid.reserve(perme.size());
    for (size_t q = 0; q < erow_slab.size(); ++q) epos_gid.push_back(slabs_eg[erow_slab[q]].aux);
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
    const char *pe = getenv("SMCPP_POWER_PREPASS");
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
    d_ek_slab_off.upload(ek_slab_off, s);
    d_epos_gid.upload(epos_gid, s);
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
    d_part_1.alloc(std::max<size_t>(1, slabs_rk.size()) * Mp * Mp);
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
            // work.  One millisecond of spinning spans the GPU phase of an eval: config C4 561 -> 676 evals/s (15 threads; measured
            // profiles/r05_*).  SMCPP_OMP_BLOCKTIME overrides.
            if (kmp_set_blocktime && !getenv("SMCPP_OMP_BLOCKTIME")) kmp_set_blocktime(1);
        }
        smcpp_host::TwoPopPrep &prep = *twopop_prep;
        {
            // the batched conditioned SFS on the device (values; SMCPP_PREP=host / smcpp_set_prep_mode(1): everything on the host)
            static const bool host_only2 = getenv("SMCPP_PREP") && !strcmp(getenv("SMCPP_PREP"), "host");
            if (!twopop_dev) { twopop_dev.reset(new TwoPopDevCsfs()); twopop_dev->device = device; twopop_dev->stream = stream; }
            prep.batch_dev = (host_only2 || force_host_prep || nder > 0) ? nullptr : twopop_dev.get();
        }
        E_on_dev = false;
        tgen_valid = false; dT_valid = true;
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
        static const bool host_only = getenv("SMCPP_PREP") && !strcmp(getenv("SMCPP_PREP"), "host");
        if (!host_only && !force_host_prep && DevPrep::supported(n[0], (int)model.a.size() + (int)hs.size()) && !smcpp_host::csfs_direct_flag()) { dev_prepare(); return; }
    }
    E_on_dev = false;
    tgen_valid = false; dT_valid = true;
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
        T = smcpp_host::transition_expand<double>(g);
        tr.mark("prep: T expand");
        tgen_valid = tgen.ok;
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
    if (dT_valid) return;
    smcpp_host::transition_expand_jac(tgen, dT);
    dT_valid = true;
}

// HMM::Q (src/hmm.cpp:155-193) summed over contigs (inference_manager.cpp:116-126) and its forward-mode gradient, evaluated on
// the device from the statistics that already live there, the device-prepared emission table (+ planes) and the generators of
// the transition matrix.  Returns false when this call has to take the host route (no device preparation, a local / global
// key-list mismatch, the pairwise fallback of the transition matrix).
bool smcpp_im::q_device(double val[4], double *jac) {
    static const bool off = getenv("SMCPP_Q") && !strcmp(getenv("SMCPP_Q"), "host");
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
    hipStream_t s = stream;
    const bool tm = getenv("SMCPP_HOST_TIMING") != nullptr;
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
    if (const char *te = getenv("SMCPP_EIG_TEAM")) team = std::max(1, std::min(16, atoi(te)));
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
        static const bool pin = !(getenv("SMCPP_EIG_PIN") && atoi(getenv("SMCPP_EIG_PIN")) == 0);
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
    { static const int pr = getenv("SMCPP_BWD_PRIO") ? std::max(0, std::min(3, atoi(getenv("SMCPP_BWD_PRIO")))) : 1; a.prio = pr; }
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
    if (const char *tv = getenv("SMCPP_COOP_TAB")) tab_c = tab_c && atoi(tv) != 0;    // test hook: force the global-table path
    shm_c = base_c + (tab_c ? tabs : 0);
}

// Eigen-free pre-pass: upload pi / T / emission table, build the group powers on the device and launch pass 0 of both
// chains on them; the host then solves the eigenproblems while the GPU runs (estep()).
void smcpp_im::stage_static_and_prepass() {
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
    static const int dbg_level = getenv("SMCPP_POWER_DEBUG") ? atoi(getenv("SMCPP_POWER_DEBUG")) : 0;
    if (dbg_level == 1) { HIPCHK(hipStreamSynchronize(s)); fprintf(stderr, "[power] powers ok\n"); static_packed = false; return; }
    int tab_c; size_t shm_c;
    coop_lds(Mp, K, G, tab_c, shm_c);
    CoopArgs cargs;
    cargs.K = K; cargs.G = G;
    cargs.power_off = (int)shm_c;          // two scratch vectors behind the regular carve-up
    shm_c += 2048;
    a.variant = 1; a.pass = 0;
    { static const int pm = getenv("SMCPP_BWD_PRIO_MASK") ? atoi(getenv("SMCPP_BWD_PRIO_MASK")) : 7; if (!(pm & 1)) a.prio = 0; }
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
    if (getenv("SMCPP_DEBUG_CYCLES")) { d_dbg.alloc(16); d_dbg.zero(s); a.dbg = d_dbg.p; }
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
    static const int prio_mask = getenv("SMCPP_BWD_PRIO_MASK") ? atoi(getenv("SMCPP_BWD_PRIO_MASK")) : 7;
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

bool smcpp_im::ss_extract_generators() {
    if (!ss_generators(M, 64 * NPL, T.data(), ss_gen, ss_c0)) return false;
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
    static const bool h32_off = getenv("SMCPP_SS_H32") && atoi(getenv("SMCPP_SS_H32")) == 0;
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
        if (h_flags_cap < 2 * (max_pass + 1)) {
            if (h_flags) (void)hipHostFree(h_flags);
            h_flags_cap = 2 * (max_pass + 1);
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
    a.eps_f = eps_f; a.eps_b = eps_b; a.full_f = a.full_b = 0;
    {
        // All scans of the stored passes in float (chains_ss.hpp: ss_x_scan_fwd / ss_x_scan_bwd), the default since round 5 for one
        // state per lane; SMCPP_SS_MIXED=0 keeps the fp64 scans (read on every E-step: tests compare the two).  Never with
        // save_gamma - the posterior's argmax is compared index by index against the reference's.
        const char *mx = getenv("SMCPP_SS_MIXED");
        const bool mixed_on = !(mx && atoi(mx) == 0);
        a.mixed = (mixed_on && !save_gamma && NPL == 1 && !ss_hybrid) ? 1 : 0;
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
        const int ef = getenv("SMCPP_SS_LIGHT_F") ? atoi(getenv("SMCPP_SS_LIGHT_F")) : -1;
        const int eb = getenv("SMCPP_SS_LIGHT_B") ? atoi(getenv("SMCPP_SS_LIGHT_B")) : -1;
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
    if (getenv("SMCPP_DEBUG_CYCLES")) { d_dbg.alloc(16); d_dbg.zero(s); a.dbg = d_dbg.p; }
    HIPCHK(hipMemcpyAsync(d_pre, pre_stage.base, off, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(ev[10], s));
    ss_launched = ss_pass0;
    const int want = std::min(max_pass, ss_pass0 + (last_ss_passes > 0 ? last_ss_passes + 1 : 6));
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
    int q = -1;
    static const bool poll = !(getenv("SMCPP_POLL") && atoi(getenv("SMCPP_POLL")) == 0);
    while (true) {
        HIPCHK(hipEventRecord(ev[3], s));
        // optimistic, as run_chains(): the statistics are queued right behind the passes; the host only looks at the flags (pinned
        // memory the kernels wrote) when the queue has drained; in the rare round that needs more passes the statistics are redone
        static const bool spec_gamma = !(getenv("SMCPP_SPEC_GAMMA") && atoi(getenv("SMCPP_SPEC_GAMMA")) == 0);
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
        if (q >= 0 || ss_launched >= max_pass) break;
        stats_enqueued = false;
        done_covers_stats = false;
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
    if (arena_side) HIPCHK(hipStreamWaitEvent(s, ev[20], 0));
    // log-likelihood (also materialises log_c per row)
    LoglikArgs la;
    la.cnorm = d_cnorm.p; la.rowinfo = d_rowinfo.p; la.g_logscale = d_g_logscale.p;
    la.contig_base = d_contig_base.p; la.contig_L = d_contig_L.p; la.partial = d_llpart.p; la.loglik = d_loglik.p;
    la.logc = d_logc.p; la.nblk = llblk;
    la.loglik_host = d_ll_view;          // the final kernel writes the per-contig values into the pinned array as well: no copy
    // save_gamma: the per-row gammas of the span > 1 rows (2 M^3 flop each, the matrix pipe's business for ~1 ms on a million rows)
    // need alpha, beta and the eigensystems only - not a single statistic: they run on their own stream BESIDE the (memory- and
    // latency-bound) statistics instead of behind them
    const bool gamma_side = save_gamma && n_e_rows > 0 && dual_stream && stream_hi != nullptr &&
                            !(getenv("SMCPP_GAMMA_SIDE") && atoi(getenv("SMCPP_GAMMA_SIDE")) == 0);
    if (save_gamma) {
        d_gamma_rows.alloc((size_t)total_rows * Mp);
        d_gamma_rows.zero(s);
        if (gamma_side) { HIPCHK(hipEventRecord(ev[22], s)); HIPCHK(hipStreamWaitEvent(stream_hi, ev[22], 0)); }
    }
    // The eigen-row branch (U/W products, rank update, span-Q Hadamard, Y) does not depend on the span-1 branch
    // (log_c, omega_1, rank update); with two streams the short launches of one fill the gaps of the other.
    const bool split_streams = dual_stream && stream2 != nullptr && !slabs_eg.empty();
    const int stats_variant = getenv("SMCPP_STATS_VARIANT") ? atoi(getenv("SMCPP_STATS_VARIANT")) : 0;
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
    static const bool span_scan_off = getenv("SMCPP_SPAN_SCAN") && atoi(getenv("SMCPP_SPAN_SCAN")) == 0;
    const bool scan_fold = eigfree && ss_active && !span_scan_off;
    const bool swap_main = crit_main && scan_fold && (stats_variant & 8);
    hipStream_t se = crit_main ? (swap_main ? stream2 : s) : split_streams ? ((eigfree && (stats_variant & 1)) ? stream_hi : stream2) : s;
    hipStream_t sp1 = crit_main ? (swap_main ? s : stream2) : s;          // the span-1 branch
    // (scan chains: run_chains_ss has just recorded ev[3] behind the last pass - the fork event, without a second record)
    hipEvent_t ev_fork = ss_active ? ev[3] : ev[8];
    if (split_streams) {
        if (!ss_active) HIPCHK(hipEventRecord(ev[8], s));
        if (se != s) HIPCHK(hipStreamWaitEvent(se, ev_fork, 0));
        if (sp1 != s) HIPCHK(hipStreamWaitEvent(sp1, ev_fork, 0));
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
    const char *kf_env = getenv("SMCPP_S1_FUSE");
    const bool kfuse = (Mp + 63) / 64 == 1 && !save_gamma && !slabs_fk.empty() &&
                       (kf_env ? atoi(kf_env) != 0 : (n_1_rows >= 500000 || crit_main));
    // (round 4: with the span > 1 branch on the main stream the span-1 statistics are ONE side branch in the one-pass form instead
    // of two - 887 against 873 headline evals per second, and 140 MB less traffic per E-step)
    const bool ll_own = split_streams && stream3 != nullptr && !eigfree;     // (eigen-free: free at the head of the main stream, which waits there)
    // crit_main with the one-pass span-1 form: the third stream has nothing else to do - the log-likelihood kernels run there,
    // beside both branches instead of at the head of the span-1 branch (joined in front of the finalisation)
    const bool ll3 = crit_main && kfuse && stream3 != nullptr;
    hipStream_t sl = (ll_own || ll3) ? stream3 : (crit_main ? sp1 : eigfree ? s : se);   // (the eigen-free branch is the longer one)
    if (ll_own || ll3) HIPCHK(hipStreamWaitEvent(sl, ev_fork, 0));
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
    aa.gpart = nullptr;
    // Eigen-free statistics: the span fold (tens of serial steps on a few CUs) ends the longest dependency chain of the
    // phase, so what it waits for - the rank accumulation of the span > 1 rows - goes FIRST and alone; the span-1 branches start
    // behind it and run while the fold does
    // (small inputs only: from ~10^6 span > 1 rows on, the rank updates are bound by memory parallelism and the two of them
    // running side by side finish sooner than one after the other - whole genome: 3.76 -> 3.36 ms of statistics)
    const bool rank2_early = eigfree && split_streams && !slabs_eg.empty() && !(stats_variant & 2) && n_e_rows < 1000000;
    const bool eig_gen2 = !eigfree && NT <= 4 && !slabs_eg.empty();      // (M > 64: the two-kernel form below)
    if (!slabs_eg.empty() && !eig_gen2) {
        d_part_e.alloc(std::max<size_t>(1, slabs_eg.size()) * Mp * Mp);
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
        hipLaunchKernelGGL(k_rank_acc<2>, dim3(ae.nslabs, ae.NB * ae.NB), dim3(64), 0, se, ae);
        if (!crit_main) {
            HIPCHK(hipEventRecord(ev[17], se));
            HIPCHK(hipStreamWaitEvent(s, ev[17], 0));
        }
    }
    // ---- span-1 branch (main stream) ----
    // M <= 64: k_rank_acc forms the weights itself, so the per-key gamma sums (k_s1_scalars + their reduction) are a third
    // independent branch: own stream, joined before the finalisation
    const bool s1_own = !kfuse && dual_stream && stream3 != nullptr && (Mp + 63) / 64 == 1 && !save_gamma && !slabs_sc.empty();
    hipStream_t s1s = s1_own ? stream3 : sp1;
    if (s1_own) {
        if (crit_main) HIPCHK(hipStreamWaitEvent(s1s, ev_fork, 0));     // (forks where the span-1 branch does: at the chains' end)
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
    if (kfuse) {
        d_part_1.alloc(std::max<size_t>(1, slabs_fk.size()) * Mp * Mp);
        d_gpart_fk.alloc(slabs_fk.size() * Mp);
        aa.nslabs = (int)slabs_fk.size(); aa.slabs = d_slabs_fk.p; aa.perm = d_perm1.p; aa.permk = nullptr; aa.part = d_part_1.p;
        aa.gpart = d_gpart_fk.p;
        hipLaunchKernelGGL(k_rank_acc<3>, dim3(aa.nslabs, 1), dim3(64), 0, sp1, aa);
        hipLaunchKernelGGL(k_sum_parts, dim3(ceil_div(Mp, 256), n_contigs * K, ZG), dim3(256), 0, sp1,
                           (const double *)d_gpart_fk.p, (const int *)d_fk_gk_off.p, d_red_g.p, Mp, ZG);
        hipLaunchKernelGGL(k_sum_parts, dim3(ceil_div(MMi, 256), n_contigs, ZS), dim3(256), 0, sp1,
                           (const double *)d_part_1.p, (const int *)d_fk_c_off.p, d_red_1.p, MMi, ZS);
    } else
    if (!slabs_rk.empty()) {
        aa.nslabs = (int)slabs_rk.size(); aa.slabs = d_slabs_rk.p; aa.perm = d_perm1.p; aa.permk = d_perm1k.p; aa.part = d_part_1.p;
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
                       (const double *)d_part_1.p, (const int *)d_s1_slab_off.p, d_red_1.p, MMi, ZS);
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
            hipLaunchKernelGGL(k_rank_acc<2>, dim3(ae.nslabs, ae.NB * ae.NB), dim3(64), 0, se, ae);
        }
        if (!eb_gid.empty())                                     // ONE share per bucket: k_span_F reads it on its serial path
            hipLaunchKernelGGL(k_sum_parts, dim3(ceil_div(MMi, 256), (unsigned)eb_gid.size(), 1), dim3(256), 0, se,
                               (const double *)d_part_e.p, (const int *)d_eb_slab_off.p, d_red_e.p, MMi, 1);
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
                SC_(1) SC_(2) SC_(3) SC_(4)
                default: SC_(8)
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
        done_folded = fold_done_epoch != 0 && !save_gamma && !ll_own;
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
        ga.alpha = d_alpha.p; ga.beta = d_beta.p; ga.gamma_rows = d_gamma_rows.p;
        const bool mfma_rows = NT <= 4;          // (M > 64: the scalar kernel on a span-Q table in memory)
        if (!mfma_rows) {
            d_Sq.alloc((size_t)G * Mp * Mp);
            hipLaunchKernelGGL(k_span_q, dim3(nb2, G), dim3(256), 0, sg, M, Mp, G, (const int *)d_g_span.p,
                               (const int *)d_g_eig.p, (const double *)d_dsc.p, (const double *)d_dpow.p, d_Sq.p);
            ga.Sq = d_Sq.p;
        }
        if (mfma_rows) {
            // one launch per (contig, eigen key): a workgroup shares one LDS copy of P, Pinv and the reciprocal eigenvalue differences
            // (NT > 2: the reciprocal differences live in registers and the fold tile is half as wide - four wavefronts fit as well)
            const int NW = 4;
            const size_t shm2 = (size_t)((NT <= 2 ? 3 : 2) * Mp * (Mp + 1) + NW * (2 * 16 * (Mp + 1) + Mp * (NT <= 2 ? 17 : 9))) * sizeof(double);
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
    if ((int)pi.size() != M || (int)T.size() != M * M || (!E_on_dev && (int)E.size() != K * M))
        throw std::runtime_error("parameters are not set");
    HIPCHK(hipEventRecord(ev[0], stream));
    // span > 1 rows without an eigensystem (kernels.hpp: k_span_fold): the span is expanded by smax steps of two M x M products
    static const bool eigfree_off = getenv("SMCPP_EIGFREE") && atoi(getenv("SMCPP_EIGFREE")) == 0;
    const bool eigfree_static = !eigfree_off && Mp <= 512 && ss_max_span <= 64 && !save_gamma;
    if (Mp > 256 && !(ss_static && eigfree_static))
        throw std::runtime_error("more than 256 hidden states: only the scan chains with eigen-free statistics are built (binned data "
                                 "with spans <= 64, no save_gamma)");
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
    im->tgen_valid = false; im->dT_valid = true;
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
    return im->gamma_valid ? im->Ls[c] + 1 : 1;
}

int smcpp_get_gamma_argmax(smcpp_im *im, int c, int *out) {
    API_BEGIN
    if (c < 0 || c >= im->n_contigs) throw std::runtime_error("contig index out of range");
    if (!im->gamma_valid) throw std::runtime_error("save_gamma was not set for the last E-step");
    HIPCHK(hipSetDevice(im->device));
    const int L = im->Ls[c];
    im->fetch_stats();
    im->d_argmax.alloc((size_t)im->total_rows);
    hipLaunchKernelGGL(k_gamma_argmax, dim3(ceil_div(L + 1, 256)), dim3(256), 0, im->stream, im->M, im->Mp,
                       (long long)(L + 1), (const double *)(im->d_gamma_rows.p + (size_t)im->contig_base[c] * im->Mp),
                       im->d_argmax.p);
    HIPCHK(hipMemcpyAsync(out, im->d_argmax.p, sizeof(int) * (L + 1), hipMemcpyDeviceToHost, im->stream));
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
static void rccl_resolve(RcclDirect &r, const char *libpath) {
    r.lib = dlopen(libpath && *libpath ? libpath : "librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) throw std::runtime_error(std::string("RCCL library not loadable: ") + dlerror());
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
    const int NPL = M > 256 ? 8 : (M + 63) / 64, MS = 64 * NPL;
    if (M > 512) throw std::runtime_error("unsupported number of hidden states");
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
        default: hipLaunchKernelGGL(k_ss_apply<8>, dim3(nvec), dim3(64), 0, s, a, (const double *)dx.p, (const double *)de.p, df.p, db.p, nvec); break;
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
