// shaping.hpp - part of the ONE translation unit engine.hip: the pre-HMM data shaping of SURVEY.md 8 f-2 on the device.
//
// What `smc++ estimate` does to every contig before an inference manager sees it (smcpp/data_filter.py:166-203):
//     Thin      estimation_tools.thin_data        (smcpp/_estimation_tools.pyx:8-84)
//     Bin       estimation_tools.bin_observations (smcpp/_estimation_tools.pyx:113-173)
//     Compress  compress_repeated_obs             (smcpp/estimation_tools.py:51-60)
// The reference walks the rows with carried state (the counter `i` of the thinning window, `seen` of the bin).  Both counters are
// functions of the POSITION a row starts at - the exclusive prefix sum of the spans - so every output row can be produced on its
// own: integer, HBM-bound work - prefix scans over the spans (own three-kernel scan: block sums, one block over the partials,
// downsweep), binary searches in the scanned array, coalesced row copies.  No matrix unit is involved and none is reshaped into it.
//   thin:      a row emits 1 .. 2k+2 rows (k = full positions inside it); counts -> scan -> one thread per OUTPUT row finds its
//              source row by binary search (a 10^8-position row of un-binned data emits 5 10^5 rows: a thread per input row would
//              serialise them)
//   bin:       one thread per bin: binary search for the first row that overlaps it, then the reference's selection rule over the
//              rows of the bin in order
//   compress:  run heads by comparison with the previous row, scan of the head flags, a head's span = difference of the scanned
//              spans at the next head and at itself
// Results are bit-exact with the reference's own code (golden G23 / G11, tests/test_gpu_shaping.py).
#pragma once

namespace smcpp_dev {

constexpr int SCAN_BLOCK = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

// ---- exclusive prefix sum of long long values: out[i] = sum_{q < i} in[q], out[n] = total ----
template <typename F>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_partials(long long n, F value, long long *__restrict__ partial) {
    __shared__ long long red[SCAN_BLOCK];
    const long long base = (long long)blockIdx.x * SCAN_TILE;
    long long s = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long i = base + (long long)k * SCAN_BLOCK + threadIdx.x;      // (coalesced: consecutive threads, consecutive items)
        if (i < n) s += value(i);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = SCAN_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// one block: exclusive scan of the nb block sums in place, total to partial[nb]
__global__ __launch_bounds__(1024) void k_scan_of_partials(long long nb, long long *__restrict__ partial) {
    __shared__ long long part[1024];
    const long long per = (nb + 1023) / 1024;
    const long long lo = (long long)threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    long long s = 0;
    for (long long i = lo; i < hi; ++i) s += partial[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                 // Hillis-Steele inclusive scan of the 1024 thread sums
        long long v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    long long run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (long long i = lo; i < hi; ++i) { const long long v = partial[i]; partial[i] = run; run += v; }
    if (threadIdx.x == 1023) partial[nb] = part[1023];
}
template <typename F>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_down(long long n, F value, const long long *__restrict__ partial, long long *__restrict__ out) {
    __shared__ long long tsum[SCAN_BLOCK];
    const long long base = (long long)blockIdx.x * SCAN_TILE;
    // thread t owns the SCAN_ITEMS consecutive items base + t * SCAN_ITEMS ..: sequential inside a thread, scanned across threads
    long long v[SCAN_ITEMS], s = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long i = base + (long long)threadIdx.x * SCAN_ITEMS + k;
        v[k] = i < n ? value(i) : 0;
        s += v[k];
    }
    tsum[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        long long u = (int)threadIdx.x >= off ? tsum[threadIdx.x - off] : 0;
        __syncthreads();
        tsum[threadIdx.x] += u;
        __syncthreads();
    }
    long long run = partial[blockIdx.x] + (threadIdx.x ? tsum[threadIdx.x - 1] : 0);
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long i = base + (long long)threadIdx.x * SCAN_ITEMS + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLOCK - 1) out[n] = partial[gridDim.x];
}

// first index q in [0, n) with arr[q + 1] > x, arr non-decreasing with arr[0] = 0 (arr = exclusive prefix sums, arr[n] = total):
// the item that holds position / output slot x
__device__ __forceinline__ long long holder_of(const long long *__restrict__ arr, long long n, long long x) {
    long long lo = 0, hi = n - 1;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (arr[mid + 1] > x) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// ... for a BLOCK of consecutive slots: the holders of slots x0 .. x0 + 255 lie between the holders of the first and the last slot, so
// two threads run the full search (22 dependent loads on 3 10^6 rows), everybody else searches the window between them (a block of 256
// bins of thinned data spans ~180 rows: 8 loads)
__device__ __forceinline__ long long holder_in_block(const long long *__restrict__ arr, long long n, long long x, long long x_first, long long x_last) {
    __shared__ long long win[2];
    if (threadIdx.x == 0) win[0] = holder_of(arr, n, x_first);
    if (threadIdx.x == blockDim.x - 1) win[1] = holder_of(arr, n, x_last);
    __syncthreads();
    long long lo = win[0], hi = win[1];
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (arr[mid + 1] > x) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// ---- thin_data (_estimation_tools.pyx:8-84) ----
// Phase i0 of a row = (offset + position it starts at) mod thinning while offset < thinning (the counter is reset at every kept
// position); a row of span s emits, in order: [thinning - i0 - 1 thinned positions] (only if > 0), [1 kept position], then for
// every further kept position inside it [thinning - 1 thinned] (only if thinning > 1) [1 kept], then [the rest thinned] (if > 0);
// or just [s thinned] when it holds no kept position.  offset >= thinning: the reference's test `i < thinning` never holds and
// every row is emitted thinned.
struct ThinGeom { long long first, kept, rem; int head; };      // first: positions up to and including the first kept one; head: 1 if first > 1
__device__ __forceinline__ ThinGeom thin_geom(long long start, long long span, long long thinning, long long offset) {
    ThinGeom g;
    if (offset >= thinning) { g.first = 0; g.kept = 0; g.rem = span; g.head = 0; return g; }
    const long long i0 = (offset + start) % thinning;
    g.first = thinning - i0;
    if (span < g.first) { g.kept = 0; g.rem = span; g.head = 0; return g; }
    g.kept = 1 + (span - g.first) / thinning;
    g.rem = (span - g.first) - (g.kept - 1) * thinning;
    g.head = g.first > 1 ? 1 : 0;
    return g;
}
__device__ __forceinline__ long long thin_count(const ThinGeom &g, long long thinning) {
    if (g.kept == 0) return 1;
    return g.head + 1 + (g.kept - 1) * ((thinning > 1 ? 1 : 0) + 1) + (g.rem > 0 ? 1 : 0);
}
struct SpanOf {
    const int *rows; int ncol;
    __device__ long long operator()(long long i) const { return (long long)rows[i * ncol]; }
};
struct ThinCountOf {
    const int *rows; int ncol; const long long *cum; long long thinning, offset;
    __device__ long long operator()(long long i) const {
        return thin_count(thin_geom(cum[i], (long long)rows[i * ncol], thinning, offset), thinning);
    }
};
__global__ __launch_bounds__(256) void k_thin_emit(long long L, int ncol, const int *__restrict__ rows, const long long *__restrict__ cum,
                                                   const long long *__restrict__ ocum, long long thinning, long long offset,
                                                   int *__restrict__ out) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nout = ocum[L];
    const long long o_first = (long long)blockIdx.x * 256, o_last = min(o_first + 255, nout - 1);
    const long long j = holder_in_block(ocum, L, min(o, nout - 1), o_first, o_last);      // (every thread takes part in the barrier)
    if (o >= nout) return;
    const long long local = o - ocum[j];
    const int *src = rows + j * ncol;
    const long long span = (long long)src[0];
    const ThinGeom g = thin_geom(cum[j], span, thinning, offset);
    const int npop = (ncol - 1) / 3;
    int sa = 0;
    for (int n = 0; n < npop; ++n) sa += src[1 + 3 * n];
    // which piece of the row is output `local`?
    long long piece_span;
    bool kept = false;
    if (g.kept == 0) piece_span = span;
    else {
        const int per = (thinning > 1 ? 1 : 0) + 1;               // outputs per further kept position
        if (local < g.head) piece_span = g.first - 1;
        else if (local == g.head) { piece_span = 1; kept = true; }
        else {
            const long long r = local - g.head - 1;
            const long long full = (g.kept - 1) * per;
            if (r < full) {
                if (per == 2 && (r & 1) == 0) piece_span = thinning - 1;
                else { piece_span = 1; kept = true; }
            } else piece_span = g.rem;
        }
    }
    int *dst = out + o * ncol;
    dst[0] = (int)piece_span;
    if (kept) {
        // sa == 2: the reference writes its `nonseg` scratch row, whose b / nb views were never filled: all zeros
        for (int c = 1; c < ncol; ++c) dst[c] = sa == 2 ? 0 : src[c];
    } else {
        for (int n = 0; n < npop; ++n) { dst[1 + 3 * n] = sa == 2 ? 0 : src[1 + 3 * n]; dst[2 + 3 * n] = 0; dst[3 + 3 * n] = 0; }
    }
}

// ---- bin_observations (_estimation_tools.pyx:113-173) ----
// bin k = positions [k w, (k + 1) w); its rows = the rows with a positive overlap, in order; the representative is the row with the
// largest observed sample size  sum_pops nb + na (a >= 0)  (first such), except that once the largest size so far is 2 a row with
// exactly one derived distinguished allele takes over (process_bin, 113-143); emitted with span 1.
__global__ __launch_bounds__(256) void k_bin_emit(long long L, int ncol, const int *__restrict__ rows, const long long *__restrict__ cum, long long w,
                                                  const long long *__restrict__ na, long long nbins, int *__restrict__ out) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long k_first = (long long)blockIdx.x * 256, k_last = min(k_first + 255, nbins - 1);
    long long q = holder_in_block(cum, L, min(k, nbins - 1) * w, k_first * w, k_last * w);
    if (k >= nbins) return;
    const long long p0 = k * w, p1 = p0 + w;
    const int K = (ncol - 1) / 3;
    long long mq = q;
    int max_ss = -2;
    for (; q < L && cum[q] < p1; ++q) {
        const int *r = rows + q * ncol;
        if (r[0] <= 0) continue;                                   // (a row without positions never has an overlap; mirrors `data[q, 0] == 0`)
        int ss = 0, seg = 0;
        for (int aa = 0; aa < K; ++aa) {
            ss += r[3 * aa + 3] + (int)na[aa] * (r[3 * aa + 1] >= 0 ? 1 : 0);
            seg += r[3 * aa + 1] > 0 ? r[3 * aa + 1] : 0;
        }
        if (ss > max_ss) { mq = q; max_ss = ss; }
        if (max_ss == 2 && seg == 1) mq = q;
    }
    int *dst = out + k * ncol;
    const int *r = rows + mq * ncol;
    dst[0] = 1;
    for (int c = 1; c < ncol; ++c) dst[c] = r[c];
}

// ---- compress_repeated_obs (estimation_tools.py:51-60) ----
struct HeadOf {
    const int *rows; int ncol;
    __device__ long long operator()(long long i) const {
        if (i == 0) return 1;
        const int *a = rows + i * ncol, *b = a - ncol;
        for (int c = 1; c < ncol; ++c) if (a[c] != b[c]) return 1;
        return 0;
    }
};
__global__ __launch_bounds__(256) void k_compress_emit(long long L, int ncol, const int *__restrict__ rows, const long long *__restrict__ cum,
                                                       const long long *__restrict__ hcum, int *__restrict__ out) {
    // one thread per OUTPUT row (run): its head is the holder of slot o in the scanned head flags, the run ends in front of the head
    // of slot o + 1 - the neighbour thread's head
    __shared__ long long heads[257];
    const long long nout = hcum[L];
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long o_first = (long long)blockIdx.x * 256, o_last = min(o_first + 255, nout - 1);
    const long long i = holder_in_block(hcum, L, min(o, nout - 1), o_first, o_last);
    if (o < nout) heads[threadIdx.x] = i;                          // (threads beyond the last run only took part in the window search)
    if (o < nout && (threadIdx.x == 255 || o == nout - 1)) heads[threadIdx.x + 1] = o + 1 < nout ? holder_of(hcum, L, o + 1) : L;
    __syncthreads();
    if (o >= nout) return;
    const long long nxt = heads[threadIdx.x + 1];
    int *dst = out + o * ncol;
    dst[0] = (int)(cum[nxt] - cum[i]);                             // (numpy: int64 differences stored into the int32 array)
    const int *src = rows + i * ncol;
    for (int c = 1; c < ncol; ++c) dst[c] = src[c];
}

}  // namespace smcpp_dev

// ---------------------------------------------------------------------------------------------------------------
// host side: a per-thread work area on the current device; the C ABI entry points below
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct ShapeArea {
    DevBuf<int> in, out, out2;
    DevBuf<long long> cum, ocum, partial, na;
    long long rows_out = 0;
    int ncol = 0;
    int *result = nullptr;                                         // device pointer of the last result (out or out2)
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    void init() {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            throw std::runtime_error("no HIP device available: the data-shaping kernels have no CPU fallback (smcpp_amd.data is the host implementation)");
        if (!s) { HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); }
    }
};
static thread_local ShapeArea g_shape;

template <typename F>
static void shape_scan(long long n, F value, DevBuf<long long> &out, ShapeArea &A) {
    using namespace smcpp_dev;
    out.alloc((size_t)n + 1);
    const long long nb = std::max<long long>(1, (n + SCAN_TILE - 1) / SCAN_TILE);
    A.partial.alloc((size_t)nb + 1);
    hipLaunchKernelGGL((k_scan_partials<F>), dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, A.s, n, value, A.partial.p);
    hipLaunchKernelGGL(k_scan_of_partials, dim3(1), dim3(1024), 0, A.s, nb, A.partial.p);
    hipLaunchKernelGGL((k_scan_down<F>), dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, A.s, n, value, (const long long *)A.partial.p, out.p);
}
static long long shape_total(const DevBuf<long long> &scanned, long long n, ShapeArea &A) {
    long long t = 0;
    HIPCHK(hipMemcpyAsync(&t, scanned.p + n, sizeof t, hipMemcpyDeviceToHost, A.s));
    HIPCHK(hipStreamSynchronize(A.s));
    return t;
}
// the three steps on rows that already live on the device: src -> dst, returns the number of rows written
static long long shape_thin(const int *src, long long L, int ncol, long long thinning, long long offset, DevBuf<int> &dst, ShapeArea &A) {
    using namespace smcpp_dev;
    shape_scan(L, SpanOf{src, ncol}, A.cum, A);
    shape_scan(L, ThinCountOf{src, ncol, A.cum.p, thinning, offset}, A.ocum, A);
    const long long nout = shape_total(A.ocum, L, A);
    dst.alloc((size_t)std::max<long long>(1, nout) * ncol);
    if (nout > 0)
        hipLaunchKernelGGL(k_thin_emit, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, A.s, L, ncol, src, (const long long *)A.cum.p,
                           (const long long *)A.ocum.p, thinning, offset, dst.p);
    return nout;
}
static long long shape_bin(const int *src, long long L, int ncol, long long w, const long long *na_dev, DevBuf<int> &dst, ShapeArea &A) {
    using namespace smcpp_dev;
    shape_scan(L, SpanOf{src, ncol}, A.cum, A);
    const long long P = shape_total(A.cum, L, A);
    const long long nbins = (P + w - 1) / w;
    dst.alloc((size_t)std::max<long long>(1, nbins) * ncol);
    if (nbins > 0)
        hipLaunchKernelGGL(k_bin_emit, dim3((unsigned)((nbins + 255) / 256)), dim3(256), 0, A.s, L, ncol, src, (const long long *)A.cum.p, w, na_dev,
                           nbins, dst.p);
    return nbins;
}
static long long shape_compress(const int *src, long long L, int ncol, DevBuf<int> &dst, ShapeArea &A) {
    using namespace smcpp_dev;
    shape_scan(L, SpanOf{src, ncol}, A.cum, A);
    shape_scan(L, HeadOf{src, ncol}, A.ocum, A);
    const long long nout = shape_total(A.ocum, L, A);
    dst.alloc((size_t)std::max<long long>(1, nout) * ncol);
    if (nout > 0)
        hipLaunchKernelGGL(k_compress_emit, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, A.s, L, ncol, src, (const long long *)A.cum.p,
                           (const long long *)A.ocum.p, dst.p);
    return nout;
}
static void shape_upload(ShapeArea &A, long long L, int ncol, const int *rows) {
    if (L <= 0 || ncol < 4 || (ncol - 1) % 3) throw std::runtime_error("data shaping: rows must be [L][1 + 3 P] int32 with L > 0");
    A.init();
    A.in.alloc((size_t)L * ncol);
    HIPCHK(hipMemcpyAsync(A.in.p, rows, sizeof(int) * (size_t)L * ncol, hipMemcpyHostToDevice, A.s));
    A.ncol = ncol;
}
}  // namespace

extern "C" {

// `mode`: 0 = thin_data(rows, p0 = thinning, p1 = offset); 1 = bin_observations(rows, p0 = w, na); 2 = compress_repeated_obs(rows);
// 3 = the pipeline of data_filter.py:166-203 without leaving HBM: Thin(p0 = thinning) -> Bin(p1 = w, na) -> Compress.
// The result stays on the device (per calling thread); *rows_out receives its row count, *kernel_ms (optional) the time of the device
// work with the input already resident in HBM (HIP events on the stream the kernels run on); smcpp_dev_shape_fetch copies it out.
int smcpp_dev_shape(int mode, long long L, int ncol, const int *rows, long long p0, long long p1, const long long *na, long long *rows_out,
                    double *kernel_ms) {
    API_BEGIN
    ShapeArea &A = g_shape;
    shape_upload(A, L, ncol, rows);
    const int npop = (ncol - 1) / 3;
    if (mode == 1 || mode == 3) {
        if (!na) throw std::runtime_error("bin_observations needs the distinguished lineages per population");
        std::vector<long long> h(na, na + npop);
        A.na.alloc(npop);
        HIPCHK(hipMemcpyAsync(A.na.p, h.data(), sizeof(long long) * npop, hipMemcpyHostToDevice, A.s));
        HIPCHK(hipStreamSynchronize(A.s));
    }
    if ((mode == 0 || mode == 3) && p0 <= 0) throw std::runtime_error("thinning must be positive");
    if ((mode == 1 && p0 <= 0) || (mode == 3 && p1 <= 0)) throw std::runtime_error("the bin width must be positive");
    HIPCHK(hipEventRecord(A.e0, A.s));
    long long n = 0;
    if (mode == 0) { n = shape_thin(A.in.p, L, ncol, p0, p1, A.out, A); A.result = A.out.p; }
    else if (mode == 1) { n = shape_bin(A.in.p, L, ncol, p0, A.na.p, A.out, A); A.result = A.out.p; }
    else if (mode == 2) { n = shape_compress(A.in.p, L, ncol, A.out, A); A.result = A.out.p; }
    else if (mode == 3) {
        const long long n1 = shape_thin(A.in.p, L, ncol, p0, 0, A.out, A);
        const long long n2 = shape_bin(A.out.p, n1, ncol, p1, A.na.p, A.out2, A);
        n = shape_compress(A.out2.p, n2, ncol, A.out, A);
        A.result = A.out.p;
    } else throw std::runtime_error("smcpp_dev_shape: unknown mode");
    HIPCHK(hipEventRecord(A.e1, A.s));
    HIPCHK(hipStreamSynchronize(A.s));
    HIPCHK(hipGetLastError());
    A.rows_out = n;
    if (rows_out) *rows_out = n;
    if (kernel_ms) { float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, A.e0, A.e1)); *kernel_ms = (double)ms; }
    API_END
}
int smcpp_dev_shape_fetch(int *out) {
    API_BEGIN
    ShapeArea &A = g_shape;
    if (!A.result) throw std::runtime_error("smcpp_dev_shape_fetch: no result on this thread");
    if (A.rows_out > 0) HIPCHK(hipMemcpy(out, A.result, sizeof(int) * (size_t)A.rows_out * A.ncol, hipMemcpyDeviceToHost));
    API_END
}

}  // extern "C"
