// Developer probe (not product, not test): what does ONE row of the cooperative forward chain cost on gfx950?
// One workgroup of 4 wavefronts per "chunk" as in k_fwd_coop<64> (lane 4*i + kq owns quarter kq of state i, float T
// quarter in registers, state exchanged through a double-buffered LDS vector, one barrier per row); variants strip or
// restructure parts of the row to see where the ~1800 cycles of the production loop go.
//   hipcc -O3 --offload-arch=gfx950 tools/chain_lab.hip -o /tmp/chain_lab && /tmp/chain_lab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct f32x4p { f32x2 lo, hi; };

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum_f(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    return v;
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float imax(float a, float b) {   // max of two non-negative floats without canonicalisation
    return __builtin_bit_cast(float, max(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b)));
}

constexpr int MT = 64, KQ = 16;

// VAR 0: production structure (sum of the quarter, rcp, clamp with fmaxf, fma, quad sum, e-multiply in double, store)
// VAR 1: as 0 with integer max (no v_max canonicalisation)
// VAR 2: as 1 with a tree-shaped quarter sum
// VAR 3: no clamp at all (sum only for the normaliser)
// VAR 4: bare mat-vec: read, fma, quad sum, write, barrier
// VAR 5: as 4 without the barrier (wrong results; timing only)
// VAR 6: as 2, normaliser taken from the row before the previous one (sum off the critical path), clamp kept via imax
// VAR 7: as 2 without the global alpha store
template <int VAR>
__global__ __launch_bounds__(256) void k_lab(const float *__restrict__ Tf, const double *__restrict__ E, float *alpha,
                                              double *cnorm, int nrows, long long *cycles) {
    __shared__ __attribute__((aligned(16))) float xf[2 * MT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int il = lane >> 2, kq = lane & 3, i = 16 * w + il;
    const bool owner = kq == 0;
    float tf[KQ];
#pragma unroll
    for (int t = 0; t < KQ; ++t) tf[t] = Tf[(size_t)(kq * KQ + t) * MT + i];
#pragma unroll
    for (int t = 0; t < KQ; ++t) asm volatile("" : "+v"(tf[t]));
    const double e = E[i];
    if (owner) xf[i] = 1.0f / MT;
    __syncthreads();
    float v_prev = 1.0f / MT, s_old = 1.0f;
    float *arow = alpha + (size_t)blockIdx.x * nrows * MT;
    const long long t0 = __builtin_readcyclecounter();
    for (int j = 0; j < nrows; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        const float *xin = xf + cur * MT + kq * KQ;
        f32x2 xl[4], xh[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4p x = *reinterpret_cast<const f32x4p *>(xin + 4 * t);
            xl[t] = x.lo; xh[t] = x.hi;
        }
        float sprev = 1.f, inv = 1.f, thr = 0.f;
        if (VAR <= 3 || VAR == 6 || VAR == 7 || VAR == 9) {
            float sq;
            if (VAR == 0 || VAR == 1 || VAR == 3) {
                f32x2 s01 = {0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) { s01 += xl[t]; s01 += xh[t]; }
                sq = s01.x + s01.y;
            } else {
                const f32x2 a = (xl[0] + xh[0]) + (xl[1] + xh[1]), b = (xl[2] + xh[2]) + (xl[3] + xh[3]);
                const f32x2 c = a + b;
                sq = c.x + c.y;
            }
            sprev = quad_sum_f(sq);
            if (VAR == 6) { const float s_now = sprev; sprev = s_old; s_old = s_now; }
            inv = __builtin_amdgcn_rcpf(sprev);
            thr = 1e-10f * sprev;
            if (j > 0 && VAR == 9) {
                // rotating store: wavefront (j & 3) writes the WHOLE previous row (256 contiguous bytes) from the exchanged
                // vector in LDS, the other three skip the store and everything that only feeds it
                if (w == (j & 3)) {
                    const float xv = xf[cur * MT + lane];
                    arow[(size_t)(j - 1) * MT + lane] = imax(xv * inv, 1e-10f);
                    if (lane == 0) cnorm[(size_t)blockIdx.x * nrows + j - 1] = (double)sprev;
                }
            } else if (j > 0) {
                float an = v_prev * inv;
                an = VAR == 0 ? fmaxf(an, 1e-10f) : imax(an, 1e-10f);
                if (VAR != 7 && owner) arow[(size_t)(j - 1) * MT + i] = an;
                if (VAR != 7 && tid == 0) cnorm[(size_t)blockIdx.x * nrows + j - 1] = (double)sprev;
            }
        }
        f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x2 l = xl[t], h = xh[t];
            if (VAR == 0) { l.x = fmaxf(l.x, thr); l.y = fmaxf(l.y, thr); h.x = fmaxf(h.x, thr); h.y = fmaxf(h.y, thr); }
            if (VAR == 1 || VAR == 2 || VAR == 6 || VAR == 7 || VAR == 9) { l.x = imax(l.x, thr); l.y = imax(l.y, thr); h.x = imax(h.x, thr); h.y = imax(h.y, thr); }
            const f32x2 m01 = {tf[4 * t], tf[4 * t + 1]}, m23 = {tf[4 * t + 2], tf[4 * t + 3]};
            acc01 = __builtin_elementwise_fma(m01, l, acc01);
            acc23 = __builtin_elementwise_fma(m23, h, acc23);
        }
        const float y = quad_sum_f((acc01.x + acc01.y) + (acc23.x + acc23.y)) * inv;
        float vout = (VAR == 4 || VAR == 5) ? y * 1.7f : (float)((double)y * e);
        if (owner) xf[nxt * MT + i] = vout;
        v_prev = vout;
        if (VAR == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else lds_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    if (owner) arow[(size_t)(nrows - 1) * MT + i] = v_prev;
}

// single wavefront per chunk, lane = state, whole float T row in registers, x broadcast from LDS (no barrier)
template <int VAR>
__global__ __launch_bounds__(64) void k_lab1(const float *__restrict__ Tf, const double *__restrict__ E, float *alpha,
                                              double *cnorm, int nrows, long long *cycles) {
    __shared__ __attribute__((aligned(16))) float xf[2 * MT];
    const int i = threadIdx.x;
    float tf[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) tf[t] = Tf[(size_t)t * MT + i];
#pragma unroll
    for (int t = 0; t < MT; ++t) asm volatile("" : "+v"(tf[t]));
    const double e = E[i];
    xf[i] = 1.0f / MT;
    float *arow = alpha + (size_t)blockIdx.x * nrows * MT;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float v = 1.0f / MT;
    const long long t0 = __builtin_readcyclecounter();
    for (int j = 0; j < nrows; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        // wave sum of v (normaliser of the incoming row) by DPP
        float s = v;
        s += dpp_mov<0xB1>(s); s += dpp_mov<0x4E>(s);
        s += dpp_mov<0x114>(s); s += dpp_mov<0x118>(s);                 // row_shr:4, row_shr:8 (partial; timing probe)
        s += dpp_mov<0x142>(s); s += dpp_mov<0x143>(s);                 // row_bcast15, row_bcast31
        s = __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 63) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 63)) : 1.f;
        const float inv = __builtin_amdgcn_rcpf(s), thr = 1e-10f * s;
        if (j > 0) arow[(size_t)(j - 1) * MT + i] = imax(v * inv, 1e-10f);
        const float *xin = xf + cur * MT;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int t = 0; t < MT; t += 4) {
            const f32x4p x = *reinterpret_cast<const f32x4p *>(xin + t);
            a0 = fmaf(tf[t], imax(x.lo.x, thr), a0);
            a1 = fmaf(tf[t + 1], imax(x.lo.y, thr), a1);
            a2 = fmaf(tf[t + 2], imax(x.hi.x, thr), a2);
            a3 = fmaf(tf[t + 3], imax(x.hi.y, thr), a3);
        }
        const float y = ((a0 + a1) + (a2 + a3)) * inv;
        v = (float)((double)y * e);
        xf[nxt * MT + i] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    if (i == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    arow[(size_t)(nrows - 1) * MT + i] = v;
}


// ---------------------------------------------------------------------------------------------------------------
// Mixed loop: span-1 rows (float mat-vec) and eigen rows (two fp64 mat-vecs through the eigenbasis of ONE key held in
// registers), row types from a descriptor array with the production mix (45 % eigen rows in runs), emission and
// eigenvalue-power tables in LDS, descriptors staged 64 rows at a time.  Candidate structure for k_fwd_coop pass 0:
// no re-run logic, no j == 0 special cases inside the loop, integer-max clamps, descriptors padded.
// MODE 0: straightforward; MODE 1: eigen rows skip the second barrier by letting every lane compute u for its quarter
// redundantly?  (not possible: u needs the full P^-1 row)  -> MODE 1 = stores dropped (timing probe)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double quad_sum_d(double v) {
    v += __builtin_bit_cast(double, (long long)0) * 0.0;   // keep the signature simple; DPP on the two halves below
    int lo = __builtin_bit_cast(int2, v).x, hi = __builtin_bit_cast(int2, v).y;
    int lo1 = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true), hi1 = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
    v += __builtin_bit_cast(double, make_int2(lo1, hi1));
    lo = __builtin_bit_cast(int2, v).x; hi = __builtin_bit_cast(int2, v).y;
    lo1 = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi1 = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true);
    v += __builtin_bit_cast(double, make_int2(lo1, hi1));
    return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_mix(const float *__restrict__ Tf, const double *__restrict__ Pinv,
                                              const double *__restrict__ Pt, const double *__restrict__ Etab,
                                              const double *__restrict__ Dtab, const int2 *__restrict__ desc, int K, int G,
                                              float *alpha, double *cnorm, int nrows, long long *cycles) {
    constexpr int UP = KQ + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *sE = reinterpret_cast<double *>(smem);
    double *sD = sE + K * MT;
    double *ub = sD + G * MT;                                   // [4][UP]
    float *xf = reinterpret_cast<float *>(ub + 4 * UP);         // [2][MT]
    int2 *sdesc = reinterpret_cast<int2 *>(xf + 2 * MT);        // [2][64]
    float *ring = reinterpret_cast<float *>(sdesc + 128);       // [8][MT] alpha rows waiting for a coalesced store
    double *cring = reinterpret_cast<double *>(ring + 8 * MT);  // [8]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int il = lane >> 2, kq = lane & 3, i = 16 * w + il;
    const bool owner = kq == 0;
    for (int x = tid; x < K * MT; x += 256) sE[x] = Etab[x];
    for (int x = tid; x < G * MT; x += 256) sD[x] = Dtab[x];
    float tf[KQ]; double pinv[KQ], pt[KQ];
#pragma unroll
    for (int t = 0; t < KQ; ++t) {
        tf[t] = Tf[(size_t)(kq * KQ + t) * MT + i];
        pinv[t] = Pinv[(size_t)(kq * KQ + t) * MT + i];
        pt[t] = Pt[(size_t)(kq * KQ + t) * MT + i];
    }
#pragma unroll
    for (int t = 0; t < KQ; ++t) { asm volatile("" : "+v"(tf[t])); asm volatile("" : "+v"(pinv[t])); asm volatile("" : "+v"(pt[t])); }
    const int2 *rd = desc + (size_t)blockIdx.x * (nrows + 192);        // padded: rows beyond nrows are span-1 rows of key 0
    if (w == 0) { sdesc[lane] = rd[lane]; sdesc[64 + lane] = rd[64 + lane]; }
    if (owner) xf[i] = 1.0f / MT;
    __syncthreads();
    float *arow = alpha + (size_t)blockIdx.x * nrows * MT;
    double *crow = cnorm + (size_t)blockIdx.x * nrows;
    int2 d0 = sdesc[0];
    int ge = __builtin_amdgcn_readfirstlane(d0.y), kid = __builtin_amdgcn_readfirstlane(d0.x);
    double e_cur = sE[kid * MT + i];
    double dp_cur = ge >= 0 ? sD[ge * MT + i] : 0.0;
    int2 d1 = sdesc[1];
    float v_prev = 1.0f / MT;
    const long long t0 = __builtin_readcyclecounter();
    for (int j = 0; j < nrows; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        if (w == 0 && (j & 63) == 32 && j >= 64) {
            const int2 dn = rd[j + 32 + lane];
            sdesc[(((j >> 6) & 1) ^ 1) * 64 + lane] = dn;
        }
        const int kid_n = __builtin_amdgcn_readfirstlane(d1.x), ge_n = __builtin_amdgcn_readfirstlane(d1.y);
        const double e_nxt = sE[kid_n * MT + i];
        const double dp_nxt = sD[(ge_n < 0 ? 0 : ge_n) * MT + i];
        const int2 d2 = sdesc[(((j + 2) >> 6) & 1) * 64 + ((j + 2) & 63)];
        const float *xin = xf + cur * MT + kq * KQ;
        f32x2 xl[4], xh[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4p x = *reinterpret_cast<const f32x4p *>(xin + 4 * t);
            xl[t] = x.lo; xh[t] = x.hi;
        }
        const f32x2 sa = (xl[0] + xh[0]) + (xl[1] + xh[1]), sb = (xl[2] + xh[2]) + (xl[3] + xh[3]);
        const f32x2 sc = sa + sb;
        const float sprev = quad_sum_f(sc.x + sc.y);
        const float inv = __builtin_amdgcn_rcpf(sprev), thr = 1e-10f * sprev;
        {
            const float an = imax(v_prev * inv, 1e-10f);
            if (MODE == 0) {
                if (owner) arow[(size_t)j * MT + i] = an;
                if (tid == 0) crow[j] = (double)sprev;
            }
            if (MODE == 2) {
                // rows go to an LDS ring; every 4th row ONE wavefront writes the four finished rows (1 KB) with a single
                // 16-byte store per lane (the barrier of the previous row ordered the ring writes)
                if (owner) ring[(j & 7) * MT + i] = an;
                if (tid == 0) cring[j & 7] = (double)sprev;
                if ((j & 3) == 0 && j >= 4 && w == ((j >> 2) & 3)) {
                    const int r0 = j - 4;                                   // rows r0 .. r0+3 are complete
                    const float4 v = *reinterpret_cast<const float4 *>(ring + (r0 & 7) * MT + lane * 4);
                    *reinterpret_cast<float4 *>(alpha + ((size_t)blockIdx.x * nrows + r0) * MT + lane * 4) = v;
                    if (lane < 4) crow[r0 + lane] = cring[(r0 & 7) + lane];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xl[t].x = imax(xl[t].x, thr); xl[t].y = imax(xl[t].y, thr); xh[t].x = imax(xh[t].x, thr); xh[t].y = imax(xh[t].y, thr);
        }
        float vout;
        if (ge < 0) {
            f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x2 m01 = {tf[4 * t], tf[4 * t + 1]}, m23 = {tf[4 * t + 2], tf[4 * t + 3]};
                acc01 = __builtin_elementwise_fma(m01, xl[t], acc01);
                acc23 = __builtin_elementwise_fma(m23, xh[t], acc23);
            }
            const float y = quad_sum_f((acc01.x + acc01.y) + (acc23.x + acc23.y)) * inv;
            vout = (float)((double)y * e_cur);
        } else {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a0 = fma(pinv[4 * t], (double)xl[t].x, a0);
                a1 = fma(pinv[4 * t + 1], (double)xl[t].y, a1);
                a0 = fma(pinv[4 * t + 2], (double)xh[t].x, a0);
                a1 = fma(pinv[4 * t + 3], (double)xh[t].y, a1);
            }
            double u = quad_sum_d(a0 + a1) * dp_cur * (double)inv;
            if (owner) ub[(i / KQ) * UP + (i % KQ)] = u;
            lds_barrier();
            const double *uin = ub + kq * UP;
            a0 = 0.0; a1 = 0.0;
#pragma unroll
            for (int t = 0; t < KQ; t += 2) {
                const double2 x = *reinterpret_cast<const double2 *>(uin + t);
                a0 = fma(pt[t], x.x, a0);
                a1 = fma(pt[t + 1], x.y, a1);
            }
            vout = (float)quad_sum_d(a0 + a1);
        }
        if (owner) xf[nxt * MT + i] = vout;
        v_prev = vout;
        ge = ge_n; kid = kid_n; e_cur = e_nxt; dp_cur = dp_nxt; d1 = d2;
        lds_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    if (owner) arow[(size_t)(nrows - 1) * MT + i] = v_prev;
}

template <typename K>
static void run(const char *name, K kernel, int threads, const float *dT, const double *dE, float *dA, double *dC,
                long long *dcyc, int nrows, int blocks) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, dT, dE, dA, dC, nrows, dcyc);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, dT, dE, dA, dC, nrows, dcyc);
    CHK(hipEventRecord(e1, 0));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    long long cyc = 0;
    CHK(hipMemcpy(&cyc, dcyc, sizeof(cyc), hipMemcpyDeviceToHost));
    printf("%-58s %8.1f ns/row  %8.1f counter ticks/row\n", name, 1e6 * ms / 5 / nrows, (double)cyc / nrows);
}

int lab_mfma();
int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 'm') return lab_mfma();
    const int nrows = 4000, blocks = 512;
    std::vector<float> T(MT * MT);
    std::vector<double> E(MT);
    for (int k = 0; k < MT; ++k) for (int i = 0; i < MT; ++i) T[k * MT + i] = (k == i ? 0.9f : 0.1f / 63);
    for (int i = 0; i < MT; ++i) E[i] = 0.5 + 0.005 * i;
    float *dT, *dA; double *dE, *dC; long long *dcyc;
    CHK(hipMalloc(&dT, T.size() * 4)); CHK(hipMalloc(&dE, E.size() * 8));
    CHK(hipMalloc(&dA, (size_t)blocks * nrows * MT * 4)); CHK(hipMalloc(&dC, (size_t)blocks * nrows * 8));
    CHK(hipMalloc(&dcyc, 8));
    CHK(hipMemcpy(dT, T.data(), T.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dE, E.data(), E.size() * 8, hipMemcpyHostToDevice));
    for (int b : {256, 512}) {
        printf("---- %d workgroups ----\n", b);
        run("0 production structure (fmaxf clamp)", k_lab<0>, 256, dT, dE, dA, dC, dcyc, nrows, b);
        run("1 integer-max clamp", k_lab<1>, 256, dT, dE, dA, dC, dcyc, nrows, b);
        run("2 integer-max clamp + tree sum", k_lab<2>, 256, dT, dE, dA, dC, dcyc, nrows, b);
        run("3 no clamp", k_lab<3>, 256, dT, dE, dA, dC, dcyc, nrows, b);
        run("4 bare mat-vec + barrier", k_lab<4>, 256, dT, dE, dA, dC, dcyc, nrows, b);
        run("5 bare mat-vec, no barrier", k_lab<5>, 256, dT, dE, dA, dC, dcyc, nrows, b);
        run("6 as 2, normaliser one row late", k_lab<6>, 256, dT, dE, dA, dC, dcyc, nrows, b);
        run("7 as 2, no global stores", k_lab<7>, 256, dT, dE, dA, dC, dcyc, nrows, b);
        run("8 one wavefront per chunk, lane = state", k_lab1<0>, 64, dT, dE, dA, dC, dcyc, nrows, b);
        run("9 as 2, whole-row store by wavefront (j & 3)", k_lab<9>, 256, dT, dE, dA, dC, dcyc, nrows, b);
    }
    // ---- mixed loop ----
    {
        const int K = 43, G = 54, nr = 920;
        std::vector<double> Pinv(MT * MT), Pt(MT * MT), Et((size_t)K * MT), Dt((size_t)G * MT);
        for (int k = 0; k < MT; ++k) for (int i = 0; i < MT; ++i) { Pinv[k * MT + i] = (k == i ? 1.0 : 0.001); Pt[k * MT + i] = (k == i ? 1.0 : -0.001); }
        for (auto &x : Et) x = 0.7; for (auto &x : Dt) x = 0.9;
        std::vector<int2> desc((size_t)blocks * (nr + 192));
        unsigned rs = 12345;
        for (size_t r = 0; r < desc.size(); ++r) {
            rs = rs * 1664525u + 1013904223u;
            const bool eig = (rs >> 8) % 100 < 45;
            desc[r] = make_int2((int)((rs >> 16) % K), eig ? (int)((rs >> 20) % G) : -1);
        }
        double *dPi, *dPt, *dEt, *dDt; int2 *dD;
        CHK(hipMalloc(&dPi, Pinv.size() * 8)); CHK(hipMalloc(&dPt, Pt.size() * 8)); CHK(hipMalloc(&dEt, Et.size() * 8));
        CHK(hipMalloc(&dDt, Dt.size() * 8)); CHK(hipMalloc(&dD, desc.size() * 8));
        CHK(hipMemcpy(dPi, Pinv.data(), Pinv.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dPt, Pt.data(), Pt.size() * 8, hipMemcpyHostToDevice));
        CHK(hipMemcpy(dEt, Et.data(), Et.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dDt, Dt.data(), Dt.size() * 8, hipMemcpyHostToDevice));
        CHK(hipMemcpy(dD, desc.data(), desc.size() * 8, hipMemcpyHostToDevice));
        const size_t shm = (size_t)(K + G) * MT * 8 + 4 * (KQ + 2) * 8 + 2 * MT * 4 + 128 * 8 + 8 * MT * 4 + 64 + 64;
        CHK(hipFuncSetAttribute((const void *)k_mix<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CHK(hipFuncSetAttribute((const void *)k_mix<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CHK(hipFuncSetAttribute((const void *)k_mix<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1;
            CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k_mix<0>, dim3(256), dim3(256), shm, 0, dT, dPi, dPt, dEt, dDt, dD, K, G, dA, dC, nr, dcyc);
                else if (mode == 1) hipLaunchKernelGGL(k_mix<1>, dim3(256), dim3(256), shm, 0, dT, dPi, dPt, dEt, dDt, dD, K, G, dA, dC, nr, dcyc);
                else hipLaunchKernelGGL(k_mix<2>, dim3(256), dim3(256), shm, 0, dT, dPi, dPt, dEt, dDt, dD, K, G, dA, dC, nr, dcyc);
            };
            launch(); CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, 0));
            for (int r = 0; r < 10; ++r) launch();
            CHK(hipEventRecord(e1, 0)); CHK(hipDeviceSynchronize());
            float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
            long long cyc = 0; CHK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
            printf("mixed loop (45%% eigen rows), %s: %8.1f ns/row, %.1f ticks/row, %.3f ms per 920-row pass\n",
                   mode == 0 ? "with stores" : mode == 1 ? "no stores  " : "LDS-staged stores", 1e6 * ms / 10 / nr, (double)cyc / nr, ms / 10);
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// MFMA lock-step probe (north_star's kernel form): 16 chains per workgroup advance together, the state is a 64 x 16
// matrix X (one column per chain) and a row-step is  Y = T^T X  (span-1 chains),  U = Pinv X, V = P (d^s o U)  (eigen
// chains) as v_mfma_f64_16x16x4_f64 tiles: wavefront w owns state tile w (the D layout of its 16 x 16 tile is also the B
// layout the next step needs, but the other three tiles live in other wavefronts, so X goes through LDS once per
// product).  All three products are computed for all 16 chains (the row types of neighbouring chains differ) and the
// result is selected per chain.  Operand quarters (A fragments of T^T, Pinv, P: 16 k-steps each) stay in registers.
// ---------------------------------------------------------------------------------------------------------------
typedef double f64x4_t __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: all three products; 1: T product only (lower bound for an all-span-1 input)
__global__ __launch_bounds__(256) void k_mfma_lock(const double *__restrict__ Tt, const double *__restrict__ Pinv,
                                                    const double *__restrict__ Pt, const double *__restrict__ Etab,
                                                    const double *__restrict__ Dtab, const int2 *__restrict__ desc, int K, int G,
                                                    float *alpha, double *cnorm, int nrows, long long *cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *sE = reinterpret_cast<double *>(smem);          // [K][64]
    double *sD = sE + K * MT;                                // [G][64]
    double *Xs = sD + G * MT;                                // [2][64][16]
    double *Us = Xs + 2 * MT * 16;                           // [64][16]
    int2 *sdesc = reinterpret_cast<int2 *>(Us + MT * 16);    // [16 chains][64 rows]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 15, kk = lane >> 4;                 // chain (column), k within a 4-step / row group of the D tile
    for (int x = tid; x < K * MT; x += 256) sE[x] = Etab[x];
    for (int x = tid; x < G * MT; x += 256) sD[x] = Dtab[x];
    // A fragments: A[m = lane & 15][k = lane >> 4] of the three operators restricted to output tile w
    double at[16], ap[16], aq[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int k = 4 * t + kk, i = 16 * w + n;
        at[t] = Tt[(size_t)k * MT + i];                      // (T^T)[i][k] = T[k][i]
        ap[t] = Pinv[(size_t)i * MT + k];                    // Pinv[i][k]
        aq[t] = Pt[(size_t)k * MT + i];                      // P[i][k] stored transposed
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) { asm volatile("" : "+v"(at[t])); asm volatile("" : "+v"(ap[t])); asm volatile("" : "+v"(aq[t])); }
    const int chain0 = blockIdx.x * 16;
    // descriptors of 64 rows of each of the 16 chains
    for (int x = tid; x < 16 * 64; x += 256) sdesc[x] = desc[(size_t)(chain0 + x / 64) * (nrows + 192) + (x % 64)];
    for (int x = tid; x < MT * 16; x += 256) Xs[x] = 1.0 / MT;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int j = 0; j < nrows; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        if ((j & 63) == 0 && j > 0) {                                    // refill the descriptor window (synchronously)
            __syncthreads();
            for (int x = tid; x < 16 * 64; x += 256) sdesc[x] = desc[(size_t)(chain0 + x / 64) * (nrows + 192) + j + (x % 64)];
            __syncthreads();
        }
        const int2 d = sdesc[n * 64 + (j & 63)];                         // this lane's chain
        const bool eig = d.y >= 0;
        // ---- B fragments of X (all 64 rows of column n over the 16 k-steps), column sum, clamp ----
        const double *xin = Xs + cur * MT * 16;
        double b[16];
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < 16; ++t) { b[t] = xin[(4 * t + kk) * 16 + n]; s += b[t]; }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const double inv = 1.0 / s, thr = 1e-10 * s;
#pragma unroll
        for (int t = 0; t < 16; ++t) b[t] = fmax(b[t], thr);
        f64x4_t Y = {0, 0, 0, 0}, U = {0, 0, 0, 0}, V = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            Y = __builtin_amdgcn_mfma_f64_16x16x4f64(at[t], b[t], Y, 0, 0, 0);
            if (MODE == 0) U = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[t], b[t], U, 0, 0, 0);
        }
        // D layout: row = kk + 4 r of tile w, column n
        double out[4];
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * w + kk + 4 * r;
                Us[i * 16 + n] = U[r] * sD[(eig ? d.y : 0) * MT + i] * inv;
            }
            lds_barrier();
#pragma unroll
            for (int t = 0; t < 16; ++t) V = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[t], Us[(4 * t + kk) * 16 + n], V, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * w + kk + 4 * r;
            const double y1 = Y[r] * inv * sE[d.x * MT + i];
            out[r] = (MODE == 0 && eig) ? V[r] : y1;
            out[r] = (double)(float)out[r];
            Xs[nxt * MT * 16 + i * 16 + n] = out[r];
            alpha[((size_t)(chain0 + n) * nrows + j) * MT + i] = (float)(out[r] * inv);
        }
        if (w == 0 && kk == 0) cnorm[(size_t)(chain0 + n) * nrows + j] = s;
        lds_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

int lab_mfma() {
    const int K = 43, G = 54, nr = 1650, blocks = 256;            // 4096 chains of 1650 rows = the whole-genome workload
    std::vector<double> Tt(MT * MT), Pinv(MT * MT), Pt(MT * MT), Et((size_t)K * MT), Dt((size_t)G * MT);
    for (int k = 0; k < MT; ++k) for (int i = 0; i < MT; ++i) { Tt[k * MT + i] = (k == i ? 0.9 : 0.1 / 63); Pinv[k * MT + i] = (k == i ? 1.0 : 0.001); Pt[k * MT + i] = (k == i ? 1.0 : -0.001); }
    for (auto &x : Et) x = 0.7; for (auto &x : Dt) x = 0.9;
    std::vector<int2> desc((size_t)blocks * 16 * (nr + 192));
    unsigned rs = 777;
    for (size_t r = 0; r < desc.size(); ++r) {
        rs = rs * 1664525u + 1013904223u;
        const bool eig = (rs >> 8) % 100 < 45;
        desc[r] = make_int2((int)((rs >> 16) % K), eig ? (int)((rs >> 20) % G) : -1);
    }
    double *dT, *dPi, *dPt, *dEt, *dDt, *dC; int2 *dD; float *dA; long long *dcyc;
    CHK(hipMalloc(&dT, Tt.size() * 8)); CHK(hipMalloc(&dPi, Pinv.size() * 8)); CHK(hipMalloc(&dPt, Pt.size() * 8));
    CHK(hipMalloc(&dEt, Et.size() * 8)); CHK(hipMalloc(&dDt, Dt.size() * 8)); CHK(hipMalloc(&dD, desc.size() * 8));
    CHK(hipMalloc(&dA, (size_t)blocks * 16 * nr * MT * 4)); CHK(hipMalloc(&dC, (size_t)blocks * 16 * nr * 8)); CHK(hipMalloc(&dcyc, 8));
    CHK(hipMemcpy(dT, Tt.data(), Tt.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dPi, Pinv.data(), Pinv.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dPt, Pt.data(), Pt.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dEt, Et.data(), Et.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dDt, Dt.data(), Dt.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dD, desc.data(), desc.size() * 8, hipMemcpyHostToDevice));
    const size_t shm = (size_t)(K + G) * MT * 8 + 3 * MT * 16 * 8 + 16 * 64 * 8 + 64;
    CHK(hipFuncSetAttribute((const void *)k_mfma_lock<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHK(hipFuncSetAttribute((const void *)k_mfma_lock<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k_mfma_lock<0>, dim3(blocks), dim3(256), shm, 0, dT, dPi, dPt, dEt, dDt, dD, K, G, dA, dC, nr, dcyc);
            else hipLaunchKernelGGL(k_mfma_lock<1>, dim3(blocks), dim3(256), shm, 0, dT, dPi, dPt, dEt, dDt, dD, K, G, dA, dC, nr, dcyc);
        };
        launch(); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0));
        for (int r = 0; r < 3; ++r) launch();
        CHK(hipEventRecord(e1, 0)); CHK(hipDeviceSynchronize());
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        long long cyc = 0; CHK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
        printf("MFMA lock-step, 16 chains / workgroup, %s: %8.1f ns per step = %6.1f ns per chain-row, %.1f ticks/step; one pass over "
               "%d chains x %d rows: %.3f ms\n", mode == 0 ? "T + Pinv + P products" : "T product only     ", 1e6 * ms / 3 / nr,
               1e6 * ms / 3 / nr / 16, (double)cyc / nr, blocks * 16, nr, ms / 3);
    }
    return 0;
}
