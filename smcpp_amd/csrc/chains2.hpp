// Second generation of the cooperative chain kernels (M <= 64): same mapping as k_fwd_coop / k_bwd_coop in kernels.hpp
// (one workgroup of MT/16 wavefronts per chunk, lane 4*il + kq owns quarter kq of the inner index of state i = 16 w + il,
// operand quarters of T and of the hot eigen key in registers, state exchanged through a double-buffered LDS vector,
// one s_barrier per mat-vec), with the row loop rebuilt around what tools/chain_lab.hip measured on gfx950:
//   * pass 0 and the re-run passes are separate instantiations (RERUN): the first pass carries no merge logic at all;
//   * the row-descriptor array is padded on both sides (engine_manager.hpp: alloc_device), so the three-stage descriptor
//     pipeline and the 64-row staging loads need no bounds tests;
//   * clamps are integer maxima (the operands are non-negative floats or tiny negative rounding residues, for which
//     the signed-integer order gives the same result): fmaxf costs a v_max canonicalisation per operand in IEEE mode;
//   * quarter sums are trees, the fp64 dot products run on four accumulators;
//   * the two most frequent eigen keys are register-resident (HOT2 instantiations exist only when there is a second key);
//   * backward chain: the emission factor of a span-1 row is applied by the PRODUCER of the exchanged vector (one
//     multiply on 16 lanes instead of 16 multiplies and 8 LDS reads on every lane), and the running scale is the sum
//     of the vector exchanged ONE ROW EARLIER - beta enters every statistic only through scale-free ratios
//     (DESIGN.md §3), so the reciprocal leaves the critical path; the chunk's end vector is still normalised exactly
//     (it seeds gamma[:,0], hmm.cpp:150).
// Semantics are those of hmm.cpp:57-149 exactly as documented at k_fwd_coop / k_bwd_coop.
#pragma once

namespace smcpp_dev {

__device__ __forceinline__ float imax_f(float a, float b) {
    return __builtin_bit_cast(float, max(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b)));
}

template <int MT, bool TAB, bool RERUN, bool HOT2, bool POWER = false>
__global__ __launch_bounds__(MT * 4) void k_fwd_coop2(ChainArgs a, CoopArgs ca) {
    constexpr int NW = MT / 16, KQ = MT / 4, Mp = MT, UP = KQ + 2, Q4 = KQ / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int il = lane >> 2, kq = lane & 3, i = 16 * w + il;
    const bool owner = kq == 0;
    const int M = a.M, pass = a.pass, c = blockIdx.x;
    if (RERUN && a.changed[pass - 1] == 0) return;
    double *sE = reinterpret_cast<double *>(smem);
    double *sD = sE + (TAB ? ca.K * MT : 0);
    double *ub = sD + (TAB ? ca.G * MT : 0);                 // [4][UP]   u exchange of eigen rows
    float *xf = reinterpret_cast<float *>(ub + 4 * UP);       // [2][MT]   unnormalised chain state (float)
    int2 *sdesc = reinterpret_cast<int2 *>(xf + 2 * MT);      // [2][64]   row descriptors
    int *sflag = reinterpret_cast<int *>(sdesc + 128);        // [1]
    int *mflag = sflag + 4;                                    // [4] per-wavefront "not merged yet" flags
    float *ptmp = reinterpret_cast<float *>(smem + ca.power_off);   // [2][MT] POWER: vector between two power applications
    if (TAB) {
        lds_stage(sE, a.E, ca.K * MT * 8, tid, NW * 64);
        if (!POWER && ca.G > 0) lds_stage(sD, a.dpow, ca.G * MT * 8, tid, NW * 64);
    }
    const Chunk ch = a.chunks[c];
    float *end_cur = a.ends_f + ((size_t)(pass & 1) * a.nchunks + c) * Mp;
    const float *end_prev = a.ends_f + ((size_t)((pass + 1) & 1) * a.nchunks + c) * Mp;
    if (RERUN && ch.first) {
        if (owner) end_cur[i] = end_prev[i];
        return;
    }
    float al = 0.f;
    {
        const float *src = ch.first ? a.pi_f
                           : !RERUN ? (a.warm_f ? a.warm_f + (size_t)(c - 1) * Mp : a.pi_f)
                                    : a.ends_f + ((size_t)((pass + 1) & 1) * a.nchunks + (c - 1)) * Mp;
        if (i < M) al = src[i];
    }
    if (tid == 0) *sflag = 0;
    if (lane == 0) mflag[w] = 1;
    __syncthreads();
    if (RERUN) {
        bool diff = false;
        if (owner && i < M) {
            const float u = a.used_f[(size_t)c * Mp + i];
            if (!(fabsf(al - u) <= a.eps_f * fabsf(u))) diff = true;
        }
        if (__any(diff) && lane == 0) *sflag = 1;
        __syncthreads();
        if (*sflag == 0) {
            if (owner) end_cur[i] = end_prev[i];
            return;
        }
    }
    if (owner) a.used_f[(size_t)c * Mp + i] = al;
    if (tid == 0) a.changed[pass] = 1;
    // POWER = pre-pass: pass 1 (a full pass) rewrites every row, only the chunk's end vector is kept
    if (ch.first && !POWER) {
        if (owner) a.alpha[(size_t)ch.base * Mp + i] = al;
        if (tid == 0) a.cnorm[ch.base] = 1.0;
    }
    // operand quarters in registers: float T and the eigenvector matrices of the two eigen keys with the most span > 1
    // rows (binned data: the monomorphic and the heterozygous reduced key); other eigen keys read theirs from L2
    float tf[KQ];
    double pinv[KQ], pt[KQ], pinv2[HOT2 ? KQ : 1], pt2[HOT2 ? KQ : 1];
    {
        const size_t ho = (size_t)(a.hot < 0 ? 0 : a.hot) * Mp * Mp, ho2 = (size_t)(a.hot2 < 0 ? 0 : a.hot2) * Mp * Mp;
#pragma unroll
        for (int t = 0; t < KQ; ++t) {
            const int k = kq * KQ + t;
            tf[t] = a.Tf[(size_t)k * Mp + i];
            pinv[t] = (!POWER && a.hot >= 0) ? a.PinvT[ho + (size_t)k * Mp + i] : 0.0;
            pt[t] = (!POWER && a.hot >= 0) ? a.PT[ho + (size_t)k * Mp + i] : 0.0;
            if (HOT2 && !POWER) {
                pinv2[HOT2 ? t : 0] = (a.hot2 >= 0) ? a.PinvT[ho2 + (size_t)k * Mp + i] : 0.0;
                pt2[HOT2 ? t : 0] = (a.hot2 >= 0) ? a.PT[ho2 + (size_t)k * Mp + i] : 0.0;
            }
        }
#pragma unroll
        for (int t = 0; t < KQ; ++t) { pin_reg(tf[t]); pin_reg(pinv[t]); pin_reg(pt[t]); if (HOT2) { pin_reg(pinv2[HOT2 ? t : 0]); pin_reg(pt2[HOT2 ? t : 0]); } }
    }
    // eigen-free pre-pass: quarters of A^2, A^4, A^8, A^16 of the hot eigen key, A = diag(e) T^T (k_binary_powers)
    float pw[POWER ? 4 : 1][POWER ? KQ : 1];
    if (POWER) {
        const float *B0 = a.Bf + (size_t)(a.hot < 0 ? 0 : a.hot) * a.npow * MT * MT + (size_t)i * MT + kq * KQ;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int t = 0; t < KQ; ++t) pw[POWER ? b : 0][POWER ? t : 0] = B0[(size_t)b * MT * MT + t];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int t = 0; t < KQ; ++t) pin_reg(pw[POWER ? b : 0][POWER ? t : 0]);
    }
    // descriptors of the chunk's rows, staged 64 at a time (batch b in sdesc[b & 1]); the array is padded, reads past the
    // chunk return descriptors of rows this workgroup never processes
    const int2 *rd = a.rowdesc + ch.base + ch.r0 + 1;
    const int nrows = ch.r1 - ch.r0;
    if (w == 0) {
        sdesc[lane] = rd[lane];
        sdesc[64 + lane] = rd[64 + lane];
    }
    if (owner) xf[i] = al;
    __syncthreads();
    int2 d0 = sdesc[0];
    int ge = __builtin_amdgcn_readfirstlane(d0.y);
    const int kid0 = __builtin_amdgcn_readfirstlane(d0.x);
    double e_cur = TAB ? sE[kid0 * MT + i] : a.E[(size_t)kid0 * Mp + i];
    double dp_cur = (!POWER && ge >= 0) ? (TAB ? sD[SMCPP_GID(ge) * MT + i] : a.dpow[(size_t)SMCPP_GID(ge) * Mp + i]) : 0.0;
    int2 d1 = sdesc[1];
    float v_prev = al;
    float *arow = a.alpha + (size_t)(ch.base + ch.r0) * Mp + i;        // row ell-1 of iteration j is arow + j * Mp
    double *crow = a.cnorm + ch.base + ch.r0;
    bool merged = false;
    for (int j = 0; j < nrows; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        if (RERUN && j > 16 && (j & 15) == 1) {
            int nm = mflag[0];
#pragma unroll
            for (int q = 1; q < NW; ++q) nm |= mflag[q];
            if (nm == 0) { merged = true; break; }
        }
        if (w == 0 && (j & 63) == 32 && j >= 64) {
            const int2 dn = rd[j + 32 + lane];
            sdesc[(((j >> 6) & 1) ^ 1) * 64 + lane] = dn;
        }
        // descriptor pipeline: (kid, ge, e, dp) of row j+1 from d1, raw descriptor of row j+2
        const int kid_n = __builtin_amdgcn_readfirstlane(d1.x);
        const int ge_n = __builtin_amdgcn_readfirstlane(d1.y);
        const int gid_n = ge_n < 0 ? 0 : SMCPP_GID(ge_n);
        const double e_nxt = TAB ? sE[kid_n * MT + i] : a.E[(size_t)kid_n * Mp + i];
        const double dp_nxt = POWER ? 0.0 : (TAB ? sD[gid_n * MT + i] : ((ge_n >= 0) ? a.dpow[(size_t)gid_n * Mp + i] : 0.0));
        const int2 d2 = sdesc[(((j + 2) >> 6) & 1) * 64 + ((j + 2) & 63)];
        // ---- incoming state: quarter of x, its sum (= normaliser of the previous row), clamp threshold ----
        const float *xin = xf + cur * MT + kq * KQ;
        f32x2 xl[Q4], xh[Q4];
#pragma unroll
        for (int t = 0; t < Q4; ++t) {
            const f32x4p x = *reinterpret_cast<const f32x4p *>(xin + 4 * t);
            xl[t] = x.lo; xh[t] = x.hi;
        }
        f32x2 sp[Q4];
#pragma unroll
        for (int t = 0; t < Q4; ++t) sp[t] = xl[t] + xh[t];
#pragma unroll
        for (int st = 1; st < Q4; st *= 2)
#pragma unroll
            for (int t = 0; t + st < Q4; t += 2 * st) sp[t] += sp[t + st];
        float sprev = quad_sum_f(sp[0].x + sp[0].y);
        sprev = (j == 0) ? 1.0f : sprev;
        const float inv = __builtin_amdgcn_rcpf(sprev);
        const float thr = 1e-10f * sprev;
        // the previous row can be finished now that its normaliser is known: alpha = clamp(v / s)   (hmm.cpp:89-94)
        if (j > 0) {
            float an = imax_f(v_prev * inv, 1e-10f);
            an = (i < M) ? an : 0.f;
            if (RERUN && (j & 15) == 0) {
                const float old_pref = owner ? arow[(size_t)j * Mp] : 0.f;
                const bool bad = owner && i < M && !(fabsf(an - old_pref) <= a.eps_f * fabsf(old_pref));
                const bool anyb = __any(bad);
                if (lane == 0) mflag[w] = anyb ? 1 : 0;
            }
            if (owner && !POWER) arow[(size_t)j * Mp] = an;
            if (tid == 0 && !POWER) crow[j] = (double)sprev;
        }
        // (the pre-pass only has to deliver a start vector: the 1e-10 floor moves its end vector by less than the tolerance
        // of the skip test, and sixteen integer maxima per lane and row are 8 % of a span-1 row)
        if (!POWER) {
#pragma unroll
            for (int t = 0; t < Q4; ++t) {
                xl[t].x = imax_f(xl[t].x, thr); xl[t].y = imax_f(xl[t].y, thr);
                xh[t].x = imax_f(xh[t].x, thr); xh[t].y = imax_f(xh[t].y, thr);
            }
        }
        float vout;
        if (ge < 0) {
            // span == 1: y = Tf^T max(x, thr) / s ; v = float(y e)
            f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < Q4; ++t) {
                const f32x2 m01 = {tf[4 * t], tf[4 * t + 1]}, m23 = {tf[4 * t + 2], tf[4 * t + 3]};
                acc01 = __builtin_elementwise_fma(m01, xl[t], acc01);
                acc23 = __builtin_elementwise_fma(m23, xh[t], acc23);
            }
            const float y = quad_sum_f((acc01.x + acc01.y) + (acc23.x + acc23.y)) * inv;
            vout = (float)((double)y * e_cur);
        } else if (POWER) {
            // eigen-free pre-pass: alpha <- A^span alpha, one application of A^(2^b) per set bit b of the span (float: this
            // pass only produces chunk-boundary vectors that the exact passes correct), the vector going through LDS
            // between two applications
            const int sp = a.g_span[SMCPP_GID(ge)];
            const bool hotk = SMCPP_ES(ge) == a.hot;
            const float *Bq = a.Bf + (size_t)SMCPP_ES(ge) * a.npow * MT * MT + (size_t)i * MT + kq * KQ;
            float vq[KQ];
#pragma unroll
            for (int t = 0; t < Q4; ++t) { vq[4 * t] = xl[t].x; vq[4 * t + 1] = xl[t].y; vq[4 * t + 2] = xh[t].x; vq[4 * t + 3] = xh[t].y; }
            float outv = 0.f;
            int napp = 0;
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                if (!((sp >> b) & 1)) continue;
                if (napp > 0) {
                    float *tb = ptmp + (napp & 1) * MT;
                    if (owner) tb[i] = outv;
                    lds_barrier();
#pragma unroll
                    for (int t = 0; t < KQ; ++t) vq[t] = tb[kq * KQ + t];
                }
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                if (b == 0) {
#pragma unroll
                    for (int t = 0; t < KQ; t += 4) {
                        a0 = fmaf(tf[t], vq[t], a0); a1 = fmaf(tf[t + 1], vq[t + 1], a1);
                        a2 = fmaf(tf[t + 2], vq[t + 2], a2); a3 = fmaf(tf[t + 3], vq[t + 3], a3);
                    }
                } else if (hotk) {
#pragma unroll
                    for (int t = 0; t < KQ; t += 4) {
                        a0 = fmaf(pw[POWER ? b - 1 : 0][POWER ? t : 0], vq[t], a0);
                        a1 = fmaf(pw[POWER ? b - 1 : 0][POWER ? t + 1 : 0], vq[t + 1], a1);
                        a2 = fmaf(pw[POWER ? b - 1 : 0][POWER ? t + 2 : 0], vq[t + 2], a2);
                        a3 = fmaf(pw[POWER ? b - 1 : 0][POWER ? t + 3 : 0], vq[t + 3], a3);
                    }
                } else {
                    const float *Bb_ = Bq + (size_t)(b - 1) * MT * MT;      // another eigen key: its powers come from L2
#pragma unroll
                    for (int t = 0; t < KQ; t += 4) {
                        const float4 m = *reinterpret_cast<const float4 *>(Bb_ + t);
                        a0 = fmaf(m.x, vq[t], a0); a1 = fmaf(m.y, vq[t + 1], a1);
                        a2 = fmaf(m.z, vq[t + 2], a2); a3 = fmaf(m.w, vq[t + 3], a3);
                    }
                }
                outv = quad_sum_f((a0 + a1) + (a2 + a3));
                if (b == 0) outv *= (float)e_cur;
                ++napp;
            }
            // spans of 32 and more (un-thinned or sparsely polymorphic data): the higher powers of every key come from L2
            for (int b = 5; b < a.nbits; ++b) {
                if (!((sp >> b) & 1)) continue;
                if (napp > 0) {
                    float *tb = ptmp + (napp & 1) * MT;
                    if (owner) tb[i] = outv;
                    lds_barrier();
#pragma unroll
                    for (int t = 0; t < KQ; ++t) vq[t] = tb[kq * KQ + t];
                }
                const float *Bb_ = Bq + (size_t)(b - 1) * MT * MT;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int t = 0; t < KQ; t += 4) {
                    const float4 m = *reinterpret_cast<const float4 *>(Bb_ + t);
                    a0 = fmaf(m.x, vq[t], a0); a1 = fmaf(m.y, vq[t + 1], a1);
                    a2 = fmaf(m.z, vq[t + 2], a2); a3 = fmaf(m.w, vq[t + 3], a3);
                }
                outv = quad_sum_f((a0 + a1) + (a2 + a3));
                ++napp;
            }
            vout = outv * inv;
        } else {
            const int es = SMCPP_ES(ge);
            double u;
            if (es == a.hot) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int t = 0; t < Q4; ++t) {
                    a0 = fma(pinv[4 * t], (double)xl[t].x, a0);
                    a1 = fma(pinv[4 * t + 1], (double)xl[t].y, a1);
                    a2 = fma(pinv[4 * t + 2], (double)xh[t].x, a2);
                    a3 = fma(pinv[4 * t + 3], (double)xh[t].y, a3);
                }
                u = quad_sum_d((a0 + a1) + (a2 + a3));
            } else if (HOT2 && es == a.hot2) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int t = 0; t < Q4; ++t) {
                    a0 = fma(pinv2[HOT2 ? (4 * t) : 0], (double)xl[t].x, a0);
                    a1 = fma(pinv2[HOT2 ? (4 * t + 1) : 0], (double)xl[t].y, a1);
                    a2 = fma(pinv2[HOT2 ? (4 * t + 2) : 0], (double)xh[t].x, a2);
                    a3 = fma(pinv2[HOT2 ? (4 * t + 3) : 0], (double)xh[t].y, a3);
                }
                u = quad_sum_d((a0 + a1) + (a2 + a3));
            } else {
                const double *Pm = a.PinvT + (size_t)es * Mp * Mp;
                double a0 = 0.0;
                for (int t = 0; t < KQ; ++t) a0 = fma(Pm[(size_t)(kq * KQ + t) * Mp + i], (double)imax_f(xin[t], thr), a0);
                u = quad_sum_d(a0);
            }
            u = u * dp_cur * (double)inv;
            if (owner) ub[(i / KQ) * UP + (i % KQ)] = u;
            lds_barrier();
            const double *uin = ub + kq * UP;
            double av;
            if (es == a.hot) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int t = 0; t < KQ; t += 4) {
                    const double2 x0 = *reinterpret_cast<const double2 *>(uin + t);
                    const double2 x1 = *reinterpret_cast<const double2 *>(uin + t + 2);
                    a0 = fma(pt[t], x0.x, a0);
                    a1 = fma(pt[t + 1], x0.y, a1);
                    a2 = fma(pt[t + 2], x1.x, a2);
                    a3 = fma(pt[t + 3], x1.y, a3);
                }
                av = quad_sum_d((a0 + a1) + (a2 + a3));
            } else if (HOT2 && es == a.hot2) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int t = 0; t < KQ; t += 4) {
                    const double2 x0 = *reinterpret_cast<const double2 *>(uin + t);
                    const double2 x1 = *reinterpret_cast<const double2 *>(uin + t + 2);
                    a0 = fma(pt2[HOT2 ? t : 0], x0.x, a0);
                    a1 = fma(pt2[HOT2 ? (t + 1) : 0], x0.y, a1);
                    a2 = fma(pt2[HOT2 ? (t + 2) : 0], x1.x, a2);
                    a3 = fma(pt2[HOT2 ? (t + 3) : 0], x1.y, a3);
                }
                av = quad_sum_d((a0 + a1) + (a2 + a3));
            } else {
                const double *Pm = a.PT + (size_t)es * Mp * Mp;
                double a0 = 0.0;
                for (int t = 0; t < KQ; ++t) a0 = fma(Pm[(size_t)(kq * KQ + t) * Mp + i], uin[t], a0);
                av = quad_sum_d(a0);
            }
            vout = (float)av;                      // the state is rounded to float as alpha_hat is (hmm.cpp:80)
        }
        vout = (i < M) ? vout : 0.f;
        if (owner) xf[nxt * MT + i] = vout;
        v_prev = vout;
        ge = ge_n; e_cur = e_nxt; dp_cur = dp_nxt; d1 = d2;
        lds_barrier();
    }
    if (merged) {
        if (owner) end_cur[i] = end_prev[i];      // the stored tail of the chunk and its end vector are still valid
        return;
    }
    // ---- last row: normalise, clamp, store, publish the end vector ----
    {
        const float *xin = xf + (nrows & 1) * MT + kq * KQ;
        float sl = 0.f;
#pragma unroll
        for (int t = 0; t < KQ; ++t) sl += xin[t];
        const float sprev = quad_sum_f(sl);
        const float inv = __builtin_amdgcn_rcpf(sprev);
        if (owner) {
            float an = v_prev * inv;
            an = (i < M) ? fmaxf(an, 1e-10f) : 0.f;
            if (!POWER) a.alpha[(size_t)(ch.base + ch.r1) * Mp + i] = an;
            end_cur[i] = an;
        }
        if (tid == 0 && !POWER) a.cnorm[ch.base + ch.r1] = (double)sprev;
    }
}

template <int MT, bool TAB, bool RERUN, bool HOT2, bool POWER = false>
__global__ __launch_bounds__(MT * 4) void k_bwd_coop2(ChainArgs a, CoopArgs ca) {
    constexpr int NW = MT / 16, KQ = MT / 4, Mp = MT, UP = KQ + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int il = lane >> 2, kq = lane & 3, i = 16 * w + il;
    const bool owner = kq == 0;
    const int M = a.M, pass = a.pass, c = blockIdx.x;
    if (RERUN && a.changed[pass - 1] == 0) return;
    // the backward chain (fp64 throughout) is the longer of the two that share a CU: its wavefronts win the issue arbitration
    // against the forward kernel's (the engine passes SMCPP_BWD_PRIO through ChainArgs::prio; 0 = hardware default)
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    double *sE = reinterpret_cast<double *>(smem);            // [K][MT]
    double *sD = sE + (TAB ? ca.K * MT : 0);                   // [G][MT]
    double *ub = sD + (TAB ? ca.G * MT : 0);                   // [4][UP]     w exchange of eigen rows
    double *xb = ub + 4 * UP;                                  // [2][4][UP]  exchanged vector (e o beta before a span-1 row)
    int2 *sdesc = reinterpret_cast<int2 *>(xb + 8 * UP);      // [2][64]
    int *sflag = reinterpret_cast<int *>(sdesc + 128);
    int *mflag = sflag + 4;
    double *ptmp = reinterpret_cast<double *>(smem + ca.power_off);   // [2][4][UP] POWER: vector between two power applications
    if (TAB) {
        lds_stage(sE, a.E, ca.K * MT * 8, tid, NW * 64);
        if (!POWER && ca.G > 0) lds_stage(sD, a.dpow, ca.G * MT * 8, tid, NW * 64);
    }
    const Chunk ch = a.chunks[c];
    double *end_cur = a.ends_b + ((size_t)(pass & 1) * a.nchunks + c) * Mp;
    const double *end_prev = a.ends_b + ((size_t)((pass + 1) & 1) * a.nchunks + c) * Mp;
    if (RERUN && ch.last) {
        if (owner) end_cur[i] = end_prev[i];
        return;
    }
    double b = 0.0;
    {
        const double *src = !RERUN ? a.warm_b + (size_t)(c + 1) * Mp
                                   : a.ends_b + ((size_t)((pass + 1) & 1) * a.nchunks + (c + 1)) * Mp;
        const bool fresh = ch.last || (!RERUN && a.warm_b == nullptr);
        if (i < M) b = fresh ? 1.0 / (double)M : src[i];
    }
    if (tid == 0) *sflag = 0;
    if (lane == 0) mflag[w] = 1;
    __syncthreads();
    if (RERUN) {
        bool diff = false;
        if (owner && i < M) {
            const double u = a.used_b[(size_t)c * Mp + i];
            if (!(fabs(b - u) <= a.eps_b * fabs(u))) diff = true;
        }
        if (__any(diff) && lane == 0) *sflag = 1;
        __syncthreads();
        if (*sflag == 0) {
            if (owner) end_cur[i] = end_prev[i];
            return;
        }
    }
    if (owner) a.used_b[(size_t)c * Mp + i] = b;
    if (tid == 0) a.changed[pass] = 1;
    double tdt[KQ], prm[KQ], pinvrm[KQ], prm2[HOT2 ? KQ : 1], pinvrm2[HOT2 ? KQ : 1];
    {
        const size_t ho = (size_t)(a.hot < 0 ? 0 : a.hot) * Mp * Mp, ho2 = (size_t)(a.hot2 < 0 ? 0 : a.hot2) * Mp * Mp;
#pragma unroll
        for (int t = 0; t < KQ; ++t) {
            const int k = kq * KQ + t;
            tdt[t] = a.TdT[(size_t)k * Mp + i];
            prm[t] = (!POWER && a.hot >= 0) ? a.Prm[ho + (size_t)k * Mp + i] : 0.0;
            pinvrm[t] = (!POWER && a.hot >= 0) ? a.Pinvrm[ho + (size_t)k * Mp + i] : 0.0;
            if (HOT2 && !POWER) {
                prm2[HOT2 ? t : 0] = (a.hot2 >= 0) ? a.Prm[ho2 + (size_t)k * Mp + i] : 0.0;
                pinvrm2[HOT2 ? t : 0] = (a.hot2 >= 0) ? a.Pinvrm[ho2 + (size_t)k * Mp + i] : 0.0;
            }
        }
#pragma unroll
        for (int t = 0; t < KQ; ++t) { pin_reg(tdt[t]); pin_reg(prm[t]); pin_reg(pinvrm[t]); if (HOT2) { pin_reg(prm2[HOT2 ? t : 0]); pin_reg(pinvrm2[HOT2 ? t : 0]); } }
    }
    // eigen-free pre-pass: quarters of the TRANSPOSED powers (A^2)^T .. (A^16)^T of the hot eigen key
    double pw[POWER ? 4 : 1][POWER ? KQ : 1];
    if (POWER) {
        const double *B0 = a.Bb + (size_t)(a.hot < 0 ? 0 : a.hot) * a.npow * MT * MT + (size_t)i * MT + kq * KQ;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int t = 0; t < KQ; ++t) pw[POWER ? b : 0][POWER ? t : 0] = B0[(size_t)b * MT * MT + t];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int t = 0; t < KQ; ++t) pin_reg(pw[POWER ? b : 0][POWER ? t : 0]);
    }
    // descriptors in processing order: iteration j handles row ell = r1 - j; the array is padded in front as well
    const int2 *rd = a.rowdesc + ch.base + ch.r1;
    const int nrows = ch.r1 - ch.r0;
    if (w == 0) {
        sdesc[lane] = rd[-lane];
        sdesc[64 + lane] = rd[-lane - 64];
    }
    __syncthreads();
    int2 d0 = sdesc[0];
    int ge = __builtin_amdgcn_readfirstlane(d0.y);
    const int kid0 = __builtin_amdgcn_readfirstlane(d0.x);
    int kid_cur = kid0;
    double dp_cur = (!POWER && ge >= 0) ? (TAB ? sD[SMCPP_GID(ge) * MT + i] : a.dpow[(size_t)SMCPP_GID(ge) * Mp + i]) : 0.0;
    int2 d1 = sdesc[1];
    // the exchanged vector z: beta itself before an eigen row, e o beta before a span-1 row (hmm.cpp:139)
    {
        const double e0 = TAB ? sE[kid0 * MT + i] : a.E[(size_t)kid0 * Mp + i];
        if (owner) xb[(i / KQ) * UP + (i % KQ)] = (ge < 0) ? b * e0 : b;
    }
    __syncthreads();
    double b_raw = b;          // owner: beta of the row being processed, in the running scale
    double inv_cur = 1.0;      // running scale: reciprocal of the sum of the vector exchanged one row earlier
    double *brow = a.beta + (size_t)(ch.base + ch.r1) * Mp + i;         // row ell of iteration j is brow - j * Mp
    bool merged = false;
    for (int j = 0; j < nrows; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        if (RERUN && j > 16 && (j & 15) == 1) {
            int nm = mflag[0];
#pragma unroll
            for (int q = 1; q < NW; ++q) nm |= mflag[q];
            if (nm == 0) { merged = true; break; }
        }
        if (w == 0 && (j & 63) == 32 && j >= 64) {
            const int2 dn = rd[-(j + 32 + lane)];
            sdesc[(((j >> 6) & 1) ^ 1) * 64 + lane] = dn;
        }
        const int kid_n = __builtin_amdgcn_readfirstlane(d1.x);
        const int ge_n = __builtin_amdgcn_readfirstlane(d1.y);
        const int gid_n = ge_n < 0 ? 0 : SMCPP_GID(ge_n);
        const double e_nxt = TAB ? sE[kid_n * MT + i] : a.E[(size_t)kid_n * Mp + i];
        const double dp_nxt = POWER ? 0.0 : (TAB ? sD[gid_n * MT + i] : ((ge_n >= 0) ? a.dpow[(size_t)gid_n * Mp + i] : 0.0));
        const int2 d2 = sdesc[(((j + 2) >> 6) & 1) * 64 + ((j + 2) & 63)];
        // ---- incoming exchanged vector: this lane's quarter ----
        const double *xin = xb + cur * 4 * UP + kq * UP;
        double x[KQ];
#pragma unroll
        for (int t = 0; t < KQ; t += 2) {
            const double2 v = *reinterpret_cast<const double2 *>(xin + t);
            x[t] = v.x; x[t + 1] = v.y;
        }
        // beta[ell] in the running scale (hmm.cpp:142 stores the vector renormalised to sum 1; every consumer of beta is
        // invariant to a per-row scale)
        {
            const double bnrm = b_raw;
            if (RERUN && (j & 15) == 0 && j > 0) {
                const double old_pref = owner ? brow[-(ptrdiff_t)j * Mp] : 0.0;
                const bool bad = owner && i < M && !(fabs(bnrm - old_pref) <= a.eps_b * fabs(old_pref));
                const bool anyb = __any(bad);
                if (lane == 0) mflag[w] = anyb ? 1 : 0;
            }
            if (owner && !POWER) brow[-(ptrdiff_t)j * Mp] = bnrm;
        }
        // running scale for the NEXT row from the sum of the vector this row consumes (off the critical path)
        double inv_next;
        {
            double s0 = (x[0] + x[1]) + (x[2] + x[3]);
#pragma unroll
            for (int t = 4; t < KQ; t += 4) s0 += (x[t] + x[t + 1]) + (x[t + 2] + x[t + 3]);
            inv_next = rcp_f64(quad_sum_d(s0));
        }
        double bn;
        if (ge < 0) {
            // beta <- T (e o beta): e was applied by the producer of x
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int t = 0; t < KQ; t += 4) {
                a0 = fma(tdt[t], x[t], a0);
                a1 = fma(tdt[t + 1], x[t + 1], a1);
                a2 = fma(tdt[t + 2], x[t + 2], a2);
                a3 = fma(tdt[t + 3], x[t + 3], a3);
            }
            bn = quad_sum_d((a0 + a1) + (a2 + a3)) * inv_cur;
        } else if (POWER) {
            // eigen-free pre-pass: beta <- (A^T)^span beta, A^T = T diag(e); one application per set bit of the span
            const int sp = a.g_span[SMCPP_GID(ge)];
            const bool hotk = SMCPP_ES(ge) == a.hot;
            const double *Bq = a.Bb + (size_t)SMCPP_ES(ge) * a.npow * MT * MT + (size_t)i * MT + kq * KQ;
            double outv = 0.0;
            int napp = 0;
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                if (!((sp >> b) & 1)) continue;
                if (napp > 0) {
                    double *tb = ptmp + (napp & 1) * 4 * UP;
                    if (owner) tb[(i / KQ) * UP + (i % KQ)] = outv;
                    lds_barrier();
#pragma unroll
                    for (int t = 0; t < KQ; ++t) x[t] = tb[kq * UP + t];
                }
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                if (b == 0) {
                    const double *eq = TAB ? sE + (size_t)kid_cur * MT + kq * KQ : a.E + (size_t)kid_cur * Mp + kq * KQ;
#pragma unroll
                    for (int t = 0; t < KQ; t += 4) {
                        a0 = fma(tdt[t] * eq[t], x[t], a0); a1 = fma(tdt[t + 1] * eq[t + 1], x[t + 1], a1);
                        a2 = fma(tdt[t + 2] * eq[t + 2], x[t + 2], a2); a3 = fma(tdt[t + 3] * eq[t + 3], x[t + 3], a3);
                    }
                } else if (hotk) {
#pragma unroll
                    for (int t = 0; t < KQ; t += 4) {
                        a0 = fma(pw[POWER ? b - 1 : 0][POWER ? t : 0], x[t], a0);
                        a1 = fma(pw[POWER ? b - 1 : 0][POWER ? t + 1 : 0], x[t + 1], a1);
                        a2 = fma(pw[POWER ? b - 1 : 0][POWER ? t + 2 : 0], x[t + 2], a2);
                        a3 = fma(pw[POWER ? b - 1 : 0][POWER ? t + 3 : 0], x[t + 3], a3);
                    }
                } else {
                    const double *Bb_ = Bq + (size_t)(b - 1) * MT * MT;
#pragma unroll
                    for (int t = 0; t < KQ; t += 4) {
                        const double2 m01 = *reinterpret_cast<const double2 *>(Bb_ + t);
                        const double2 m23 = *reinterpret_cast<const double2 *>(Bb_ + t + 2);
                        a0 = fma(m01.x, x[t], a0); a1 = fma(m01.y, x[t + 1], a1);
                        a2 = fma(m23.x, x[t + 2], a2); a3 = fma(m23.y, x[t + 3], a3);
                    }
                }
                outv = quad_sum_d((a0 + a1) + (a2 + a3));
                ++napp;
            }
            for (int b = 5; b < a.nbits; ++b) {          // long spans: higher powers from L2 (see k_fwd_coop2)
                if (!((sp >> b) & 1)) continue;
                if (napp > 0) {
                    double *tb = ptmp + (napp & 1) * 4 * UP;
                    if (owner) tb[(i / KQ) * UP + (i % KQ)] = outv;
                    lds_barrier();
#pragma unroll
                    for (int t = 0; t < KQ; ++t) x[t] = tb[kq * UP + t];
                }
                const double *Bb_ = Bq + (size_t)(b - 1) * MT * MT;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int t = 0; t < KQ; t += 4) {
                    const double2 m01 = *reinterpret_cast<const double2 *>(Bb_ + t);
                    const double2 m23 = *reinterpret_cast<const double2 *>(Bb_ + t + 2);
                    a0 = fma(m01.x, x[t], a0); a1 = fma(m01.y, x[t + 1], a1);
                    a2 = fma(m23.x, x[t + 2], a2); a3 = fma(m23.y, x[t + 3], a3);
                }
                outv = quad_sum_d((a0 + a1) + (a2 + a3));
                ++napp;
            }
            bn = outv * inv_cur;
        } else {
            const int es = SMCPP_ES(ge);
            double wv;
            if (es == a.hot) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int t = 0; t < KQ; t += 4) {
                    a0 = fma(prm[t], x[t], a0);
                    a1 = fma(prm[t + 1], x[t + 1], a1);
                    a2 = fma(prm[t + 2], x[t + 2], a2);
                    a3 = fma(prm[t + 3], x[t + 3], a3);
                }
                wv = quad_sum_d((a0 + a1) + (a2 + a3));
            } else if (HOT2 && es == a.hot2) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int t = 0; t < KQ; t += 4) {
                    a0 = fma(prm2[HOT2 ? t : 0], x[t], a0);
                    a1 = fma(prm2[HOT2 ? (t + 1) : 0], x[t + 1], a1);
                    a2 = fma(prm2[HOT2 ? (t + 2) : 0], x[t + 2], a2);
                    a3 = fma(prm2[HOT2 ? (t + 3) : 0], x[t + 3], a3);
                }
                wv = quad_sum_d((a0 + a1) + (a2 + a3));
            } else {
                const double *Pm = a.Prm + (size_t)es * Mp * Mp;
                double a0 = 0.0;
                for (int t = 0; t < KQ; ++t) a0 = fma(Pm[(size_t)(kq * KQ + t) * Mp + i], x[t], a0);
                wv = quad_sum_d(a0);
            }
            wv = wv * dp_cur * inv_cur;
            if (owner) ub[(i / KQ) * UP + (i % KQ)] = wv;
            lds_barrier();
            const double *uin = ub + kq * UP;
            if (es == a.hot) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int t = 0; t < KQ; t += 4) {
                    const double2 v0 = *reinterpret_cast<const double2 *>(uin + t);
                    const double2 v1 = *reinterpret_cast<const double2 *>(uin + t + 2);
                    a0 = fma(pinvrm[t], v0.x, a0);
                    a1 = fma(pinvrm[t + 1], v0.y, a1);
                    a2 = fma(pinvrm[t + 2], v1.x, a2);
                    a3 = fma(pinvrm[t + 3], v1.y, a3);
                }
                bn = quad_sum_d((a0 + a1) + (a2 + a3));
            } else if (HOT2 && es == a.hot2) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int t = 0; t < KQ; t += 4) {
                    const double2 v0 = *reinterpret_cast<const double2 *>(uin + t);
                    const double2 v1 = *reinterpret_cast<const double2 *>(uin + t + 2);
                    a0 = fma(pinvrm2[HOT2 ? t : 0], v0.x, a0);
                    a1 = fma(pinvrm2[HOT2 ? (t + 1) : 0], v0.y, a1);
                    a2 = fma(pinvrm2[HOT2 ? (t + 2) : 0], v1.x, a2);
                    a3 = fma(pinvrm2[HOT2 ? (t + 3) : 0], v1.y, a3);
                }
                bn = quad_sum_d((a0 + a1) + (a2 + a3));
            } else {
                const double *Pm = a.Pinvrm + (size_t)es * Mp * Mp;
                double a0 = 0.0;
                for (int t = 0; t < KQ; ++t) a0 = fma(Pm[(size_t)(kq * KQ + t) * Mp + i], uin[t], a0);
                bn = quad_sum_d(a0);
            }
        }
        bn = (i < M) ? bn : 0.0;
        inv_cur = inv_next;
        // exchanged vector for the next row; the last row of the chunk hands over plain beta (exact normalisation below)
        const bool e_next = ge_n < 0 && j + 1 < nrows;
        if (owner) xb[nxt * 4 * UP + (i / KQ) * UP + (i % KQ)] = e_next ? bn * e_nxt : bn;
        b_raw = bn;
        kid_cur = kid_n;
        ge = ge_n; dp_cur = dp_nxt; d1 = d2;
        lds_barrier();
    }
    if (merged) {
        if (owner) end_cur[i] = end_prev[i];
        return;
    }
    {
        const double *xin = xb + (nrows & 1) * 4 * UP + kq * UP;
        double sl = 0.0;
#pragma unroll
        for (int t = 0; t < KQ; ++t) sl += xin[t];
        const double sprev = quad_sum_d(sl);
        if (owner) {
            const double bf = (i < M) ? b_raw / sprev : 0.0;       // beta /= beta.sum()
            end_cur[i] = bf;
            if (ch.first && !POWER) a.beta[(size_t)ch.base * Mp + i] = bf;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Operands of the eigen-free pre-pass.  A_e = diag(e_key) T^T is the one-position forward operator of eigen key e
// (the matrix whose eigensystem TransitionBundle::update takes, transition_bundle.cpp:15-25); a span-s row applies
// A_e^s (forward) or its transpose (backward) as the product of the binary powers A^(2^b) of the set bits of s.
// One workgroup per eigen key squares A nsq times (A^2 .. A^(2^nsq); nsq = 4 covers spans up to 31, 11 spans up to 4095)
// in LDS on the matrix cores and stores them as
// float row-major (forward operand Bf[e][b-1][i][k]) and double transposed (backward operand Bb[e][b-1][i][k] =
// A^(2^b)[k][i]).  Tens of microseconds, against 0.6 ms of host eigensolves taken off the critical path (engine_plans.hpp: estep).
// ---------------------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void k_binary_powers(int M, int nsq, const int *__restrict__ e_kid,
                                                        const double *__restrict__ E, const double *__restrict__ Td,
                                                        float *__restrict__ Bf, double *__restrict__ Bb) {
    constexpr int LD = MT + 1, NT = MT / 16, NE = (MT * MT + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) double smp[];
    double *sP = smp;                       // [2][MT][LD] running power, rows padded
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m = lane & 15, qd = lane >> 4;
    const double *em = E + (size_t)e_kid[e] * MT;
    {
        // A[i][k] = e_i T[k][i]: coalesced reads of T's rows, every load issued before the first LDS store
        double v[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int idx = tid + 256 * u, k = idx / MT, i = idx % MT;
            v[u] = (idx < MT * MT && i < M && k < M) ? em[i] * Td[(size_t)k * MT + i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int idx = tid + 256 * u, k = idx / MT, i = idx % MT;
            if (idx < MT * MT) sP[i * LD + k] = v[u];
        }
    }
    double *smax = sP + 2 * MT * LD;        // [4] per-wavefront maxima (behind the two matrices, see the launch)
    __syncthreads();
    for (int b = 0; b < nsq; ++b) {
        const double *Pc = sP + (b & 1) * MT * LD;
        double *Pn = sP + ((b & 1) ^ 1) * MT * LD;
        // square on the matrix cores: wavefront w owns rows 16w .. 16w+15 (v_mfma_f64_16x16x4: A[m][k = qd], B[k = qd][n = m],
        // D[row = qd + 4r][col = m])
        if (w < NT) {
            f64x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = (f64x4){0, 0, 0, 0};
#pragma unroll 4
            for (int kk = 0; kk < MT / 4; ++kk) {
                const double av = Pc[(16 * w + m) * LD + 4 * kk + qd];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const double bv = Pc[(4 * kk + qd) * LD + 16 * t + m];
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) Pn[(16 * w + qd + 4 * r) * LD + 16 * t + m] = acc[t][r];
        }
        __syncthreads();
        // Powers beyond A^16 (spans of 32 and more) would leave the float range (|lambda| < 1): every power is rescaled to
        // max |entry| = 1.  The pre-pass only needs directions - both chains renormalise their vector on every row.
        if (b >= 4) {
            double mx = 0.0;
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                const int idx = tid + 256 * u, r = idx / MT, c = idx % MT;
                if (idx < MT * MT) mx = fmax(mx, fabs(Pn[r * LD + c]));
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
            if (lane == 0) smax[w] = mx;
            __syncthreads();
            mx = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
            const double sc = mx > 0.0 ? 1.0 / mx : 1.0;
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                const int idx = tid + 256 * u, r = idx / MT, c = idx % MT;
                if (idx < MT * MT) Pn[r * LD + c] *= sc;
            }
            __syncthreads();
        }
        // copy the new power out, both layouts coalesced (row-major float: lanes along k; transposed double: lanes along i)
        float *of = Bf + ((size_t)e * nsq + b) * MT * MT;
        double *ob = Bb + ((size_t)e * nsq + b) * MT * MT;
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int idx = tid + 256 * u, r = idx / MT, c = idx % MT;
            if (idx < MT * MT) {
                of[idx] = (float)Pn[r * LD + c];
                ob[idx] = Pn[c * LD + r];
            }
        }
    }
}

}  // namespace smcpp_dev
