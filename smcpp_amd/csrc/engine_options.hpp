// engine_options.hpp - part of the ONE translation unit engine.hip (included first): every SMCPP_* environment switch of the
// engine in one table, parsed ONCE per process (first use) and again only by smcpp_reload_options().  No other file of the engine
// calls getenv.  smcpp_describe() prints the table together with the plan a manager resolved from it.
// (The reference has no counterpart: its only run-time controls are set_num_threads and the logger, _smcpp.pyx:32-64.)
#pragma once

namespace smcpp_opt {

enum OptId {
    O_HOST_TRACE, O_HOST_TIMING, O_OMP_BLOCKTIME, O_DUAL_STREAM, O_EVENT_FLAGS, O_LOCK_MIN_ROWS, O_CHAIN, O_SS, O_HYBRID, O_HYB_TH,
    O_COOP_BPC, O_ROWS_PER_CHUNK, O_SS_WPC, O_SS_HALO, O_HALO_LF, O_HALO_DF, O_HALO_LB, O_HALO_DB, O_SS_FWD_SHARE, O_STATS_TEAM,
    O_SLAB_ROWS, O_POWER_PREPASS, O_PREP, O_Q, O_EIG_TEAM, O_EIG_PIN, O_BWD_PRIO, O_COOP_TAB, O_POWER_DEBUG, O_BWD_PRIO_MASK,
    O_DEBUG_CYCLES, O_SS_H32, O_SS_MIXED, O_SS_LIGHT_F, O_SS_LIGHT_B, O_SS_CERT_PASS, O_POLL, O_SPEC_GAMMA, O_GAMMA_SIDE,
    O_STATS_VARIANT, O_SPAN_SCAN, O_S1_FUSE, O_EIGFREE, O_CSFS_DIRECT, O_RANK_WIDE, O_GAMMA_SCAN, O_GAMMA_PIECES, O_SPLIT_SPANS, O_T_LAZY, O_DEBUG_POISON, O_DEBUG_POISON_ONLY, O_DEBUG_POISON_LOG, O_COUNT
};

struct OptDef { const char *name, *help; };

// clang-format off
static const OptDef OPT_DEFS[O_COUNT] = {
    {"SMCPP_HOST_TRACE",     "> 0: print a timestamped trace of the host phase of every E-step to stderr"},
    {"SMCPP_HOST_TIMING",    "set: print the split of the cold preparation / eigensolver / joint CSFS to stderr"},
    {"SMCPP_OMP_BLOCKTIME",  "libomp block time in ms (default 0; two-population managers 2)"},
    {"SMCPP_DUAL_STREAM",    "0: statistics branches on one stream"},
    {"SMCPP_EVENT_FLAGS",    "hipEventCreateWithFlags flags (default hipEventDisableSystemFence; 0 = default events)"},
    {"SMCPP_LOCK_MIN_ROWS",  "rows per (CU x 16) from which the lock-step MFMA chains are selected"},
    {"SMCPP_CHAIN",          "lock | dense | ss: force a chain family"},
    {"SMCPP_SS",             "0: no scan chains (the dense families)"},
    {"SMCPP_HYBRID",         "0: no hybrid rows on un-binned data (dense chains)"},
    {"SMCPP_HYB_TH",         "span above which a row is one eigen-power step (default 6)"},
    {"SMCPP_COOP_BPC",       "workgroups per CU of the cooperative chains"},
    {"SMCPP_ROWS_PER_CHUNK", "rows per chunk (0 = automatic)"},
    {"SMCPP_SS_WPC",         "wavefronts per SIMD of the scan chains (1..4; default by input size: M <= 64 two from 1.95, three from 12 million positions)"},
    {"SMCPP_SS_HALO",        "1 / 0: force / forbid the halo pass of the scan chains (default: M > 64, and M <= 64 with 1.35 - 12 million positions)"},
    {"SMCPP_HALO_LF",        "float halo positions, forward (2800)"},
    {"SMCPP_HALO_DF",        "exact halo positions, forward (800)"},
    {"SMCPP_HALO_LB",        "float halo positions, backward (3900)"},
    {"SMCPP_HALO_DB",        "exact halo positions, backward (1100)"},
    {"SMCPP_SS_FWD_SHARE",   "share of the wavefronts given to the forward chain"},
    {"SMCPP_STATS_TEAM",     "0: one rank partial per slab instead of per team of four"},
    {"SMCPP_SLAB_ROWS",      "minimum rows per statistics slab (128)"},
    {"SMCPP_POWER_PREPASS",  "0 / 1: eigen-free pre-pass of the dense chains off / forced"},
    {"SMCPP_PREP",           "host: the whole cold preparation on host threads"},
    {"SMCPP_Q",              "host: Q and its gradient on the host"},
    {"SMCPP_EIG_TEAM",       "threads per eigen key of the host eigensolver"},
    {"SMCPP_EIG_PIN",        "0: do not pin eigensolver teams to one L3 domain"},
    {"SMCPP_BWD_PRIO",       "s_setprio of the backward cooperative chain (0..3, default 1)"},
    {"SMCPP_COOP_TAB",       "0: test hook, tables of the cooperative chains from global memory"},
    {"SMCPP_POWER_DEBUG",    "debug level of the pre-pass"},
    {"SMCPP_BWD_PRIO_MASK",  "bit mask of the passes that raise the backward chain's priority (7)"},
    {"SMCPP_DEBUG_CYCLES",   "set: shader clocks / positions of chunk 1, pass 0 to stderr"},
    {"SMCPP_SS_H32",         "0: no two-row form of the scans at M <= 32"},
    {"SMCPP_SS_MIXED",       "0: fp64 scans in the stored passes; 1: float scans (default, one state per lane)"},
    {"SMCPP_SS_LIGHT_F",     "number of float history passes, forward"},
    {"SMCPP_SS_LIGHT_B",     "number of float history passes, backward"},
    {"SMCPP_SS_CERT_PASS",   "1: launch the all-skip certificate pass up front"},
    {"SMCPP_POLL",           "0: hipStreamSynchronize instead of polling the pinned completion word"},
    {"SMCPP_SPEC_GAMMA",     "0: save_gamma E-steps wait for the certificate before the statistics"},
    {"SMCPP_GAMMA_SIDE",     "0: per-row gammas on the main stream"},
    {"SMCPP_STATS_VARIANT",  "bit mask of stream arrangements of the statistics phase"},
    {"SMCPP_SPAN_SCAN",      "0: span fold on the matrix cores (k_span_big) instead of the scans"},
    {"SMCPP_S1_FUSE",        "0 / 1: two-kernel / one-pass span-1 statistics"},
    {"SMCPP_EIGFREE",        "0: eigensystem statistics even where the eigen-free form applies"},
    {"SMCPP_CSFS_DIRECT",    "set: literal O(pieces^2 n^2) conditioned SFS (test hook)"},
    {"SMCPP_RANK_WIDE",      "0: rank updates at M > 128 without the LDS-staged operand rows (k_rank_acc instead of k_rank_acc_wide); 4: four wavefronts per workgroup instead of eight"},
    {"SMCPP_GAMMA_SCAN",     "0: save_gamma E-steps take the eigensystem statistics and per-row gammas (M <= 256) instead of the eigen-free scans"},
    {"SMCPP_GAMMA_PIECES",   "0: per-row gammas of long rows at 64 < M <= 256 from the eigensystem (k_gamma_rows_eig) instead of eigen-power pieces + scan steps"},
    {"SMCPP_SPLIT_SPANS",    "0: rows of binned data longer than 64 positions are not cut into pieces (M > 64); 2: un-binned rows are cut as well at M <= 256 (test switch)"},
    {"SMCPP_T_LAZY",         "0: the model path expands the M x M transition matrix before the chains start and takes their generators from it (rounds 3-5)"},
    {"SMCPP_DEBUG_POISON",   "byte value every fresh device allocation is filled with (255: NaN / -1): a kernel that reads memory it was "
                             "never given shows up as NaN in a fresh process instead of depending on what the allocator recycles"},
    {"SMCPP_DEBUG_POISON_ONLY", "poison only the allocation with this index (counted from the last smcpp_reload_options): tools/poison_probe.py"},
    {"SMCPP_DEBUG_POISON_LOG",  "set: print index, source line and size of every device allocation to stderr"},
};
// clang-format on

struct EngineOptions {
    bool set_[O_COUNT];
    std::string val_[O_COUNT];
    long long ival_[O_COUNT];
    double dval_[O_COUNT];
    unsigned generation = 0;
    void parse() {
        for (int k = 0; k < O_COUNT; ++k) {
            const char *e = getenv(OPT_DEFS[k].name);
            set_[k] = e != nullptr;
            val_[k] = e ? e : "";
            ival_[k] = e ? strtoll(e, nullptr, 0) : 0;
            dval_[k] = e ? atof(e) : 0.0;
        }
        ++generation;
    }
    bool has(OptId k) const { return set_[k]; }
    long long ll(OptId k, long long dflt) const { return set_[k] ? ival_[k] : dflt; }
    int i(OptId k, int dflt) const { return set_[k] ? (int)ival_[k] : dflt; }
    double d(OptId k, double dflt) const { return set_[k] ? dval_[k] : dflt; }
    bool is(OptId k, const char *s) const { return set_[k] && val_[k] == s; }
    bool off(OptId k) const { return set_[k] && ival_[k] == 0; }          // "NAME=0" (a switch that defaults to on)
    bool on(OptId k) const { return set_[k] && ival_[k] != 0; }           // "NAME=<non-zero>" (a switch that defaults to off)
    const std::string &str(OptId k) const { return val_[k]; }
};

inline EngineOptions &options_mut() {
    static EngineOptions o;
    static std::once_flag once;
    std::call_once(once, [] { o.parse(); });
    return o;
}
inline const EngineOptions &opt() { return options_mut(); }
// Re-read the environment (tests; a process that changes a switch between managers).  Not to be called while an E-step runs.
inline int &alloc_counter() { static int c = 0; return c; }
inline void reload() { options_mut().parse(); alloc_counter() = 0; }
// Debug aid (SMCPP_DEBUG_POISON): fill a FRESH device allocation with a byte pattern.  hipMalloc hands out zeroed pages in a fresh
// process and recycled ones later; a kernel that reads memory nobody wrote is correct in the first case only by accident.
inline void poison_fill(void *p, size_t bytes, int value) { (void)hipMemset(p, value, bytes); (void)hipDeviceSynchronize(); }
inline void poison(void *p, size_t bytes, int line, const char *file) {
    const EngineOptions &o = opt();
    const int idx = alloc_counter()++;
    if (o.has(O_DEBUG_POISON_LOG)) fprintf(stderr, "[alloc %d] %s:%d %zu bytes\n", idx, file, line, bytes);
    if (!o.has(O_DEBUG_POISON) || !p || !bytes) return;
    if (o.has(O_DEBUG_POISON_ONLY) && o.i(O_DEBUG_POISON_ONLY, -1) != idx) return;
    poison_fill(p, bytes, o.i(O_DEBUG_POISON, 255) & 0xff);
}

inline std::string describe_options() {
    const EngineOptions &o = opt();
    std::string s = "\"options\": {";
    bool first = true;
    for (int k = 0; k < O_COUNT; ++k) {
        if (!o.set_[k]) continue;
        s += first ? "" : ", ";
        first = false;
        s += std::string("\"") + OPT_DEFS[k].name + "\": \"" + o.val_[k] + "\"";
    }
    s += "}, \"options_known\": " + std::to_string((int)O_COUNT) + ", \"options_generation\": " + std::to_string(o.generation);
    return s;
}

}  // namespace smcpp_opt
using smcpp_opt::opt;
