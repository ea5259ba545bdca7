// Micro-benchmarks behind chains_ss.hpp: issue cost (shader clocks per instruction, one wavefront per SIMD) of the cross-lane
// moves a wavefront scan is made of.   hipcc --offload-arch=gfx950 -O3 tools/dpp_lab.hip -o tools/dpp_lab && tools/dpp_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int V>
__global__ __launch_bounds__(64) void k(long long *out, float *sink, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (V == 0) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %4, %4, %5\n v_add_f32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (V == 1) { REP16(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (V == 2) { REP16(asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (V == 3) { REP16(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (V == 4) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %0\n v_fma_f64 %3, %3, %0, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (V == 5) { REP16(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (V == 6) { REP16(asm volatile("v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 row_bcast:31 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %7 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (V == 7) { REP16(asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (V == 8) { REP16(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (V == 9) { REP16(asm volatile("v_fmac_f32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %2, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %4, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %6, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (V == 10) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (V == 11) { REP16(asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:15 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %2, %3 row_newbcast:15 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[V] = t1 - t0;
    sink[threadIdx.x + 64 * blockIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3);
}

int main() {
    long long *d;
    float *sink;
    hipMalloc(&d, 16 * sizeof(long long));
    hipMemset(d, 0, 16 * sizeof(long long));
    hipMalloc(&sink, 64 * 1024 * sizeof(float));
    const int iters = 2000, blocks = 1024;
    const char *names[] = {"v_add_f32", "v_mov_b32_dpp row_shr:1", "v_add_f32_dpp row_shr:1", "v_add_f64", "v_fma_f64", "v_mov_b32_dpp quad_perm",
                           "v_mov_b32_dpp bcast15/31/wave_shr/shr8", "v_mul_f64", "v_permlane32/16_swap", "v_fmac_f32_dpp", "v_pk_add_f32", "v_mov_b64_dpp row_newbcast"};
#define RUN(V) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, sink, iters);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
    hipDeviceSynchronize();
    long long h[16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int v = 0; v < 12; ++v) printf("%-45s %6.2f clocks / instruction\n", names[v], (double)h[v] / (iters * 64.0));
    // aggregate VALU issue rate of the chip with W single-wavefront workgroups per SIMD (wall clock): the roofline's peak
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto rate = [&](int v, int W) {
        const int nb = 1024 * W, it = 4000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(64), 0, 0, d, sink, it);
            if (v == 4) hipLaunchKernelGGL(k<4>, dim3(nb), dim3(64), 0, 0, d, sink, it);
            if (v == 9) hipLaunchKernelGGL(k<9>, dim3(nb), dim3(64), 0, 0, d, sink, it);
            if (v == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(64), 0, 0, d, sink, it);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        return (double)nb * it * 64.0 / (ms * 1e-3) / 1e9;
    };
    hipFree(sink);
    hipMalloc(&sink, (size_t)64 * 1024 * 8 * sizeof(float));
    const int vs[] = {0, 9, 1, 4};
    for (int v : vs) {
        printf("%-28s G wave-instructions / s, chip, with 1 / 2 / 4 / 8 wavefronts per SIMD:", names[v]);
        for (int W = 1; W <= 8; W *= 2) printf(" %7.1f", rate(v, W));
        printf("\n");
    }
    return 0;
}
