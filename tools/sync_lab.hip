// Cost of the stream-ordering primitives the statistics phase is built from, on this box (diagnostics; not part of the product).
//   hipcc --offload-arch=gfx950 -O2 -o tools/sync_lab tools/sync_lab.hip && tools/sync_lab
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_spin(long long cycles, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 1000) *sink = 1;
}
__global__ void k_wait_flag(const int *flag, int target, long long cycles) {
    if (threadIdx.x == 0) while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
}
__global__ void k_spin_set(long long cycles, int *flag, int value) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    CK(hipSetDevice(0));
    hipStream_t s1, s2, s3;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    hipEvent_t et[4], en[4];
    for (int i = 0; i < 4; ++i) { CK(hipEventCreate(&et[i])); CK(hipEventCreateWithFlags(&en[i], hipEventDisableTiming)); }
    int *flag;
    CK(hipMalloc((void **)&flag, 64));
    CK(hipMemset(flag, 0, 64));
    const long long C = 500;          // wall_clock64 ticks at 100 MHz: 5 us
    const int N = 200;
    auto run = [&](const char *name, auto body, int kernels_per_iter) {
        for (int w = 0; w < 20; ++w) body(w);
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < N; ++i) body(20 + i);
        CK(hipDeviceSynchronize());
        const double per = (now() - t0) / N;
        printf("%-64s %7.2f us per iteration  (%d kernels of 5 us: overhead %6.2f us)\n", name, per, kernels_per_iter, per - 5.0 * kernels_per_iter);
    };
    run("T1 two kernels, one stream", [&](int) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, C, nullptr); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, C, nullptr); }, 2);
    run("T2 kernel, timing event record, kernel (one stream)", [&](int) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, C, nullptr); CK(hipEventRecord(et[0], s1)); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, C, nullptr); }, 2);
    run("T3 kernel, no-timing event record, kernel (one stream)", [&](int) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, C, nullptr); CK(hipEventRecord(en[0], s1)); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, C, nullptr); }, 2);
    run("T4 s1 kernel -> timing event -> s2 kernel -> timing event -> s1", [&](int) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, C, nullptr); CK(hipEventRecord(et[0], s1)); CK(hipStreamWaitEvent(s2, et[0], 0));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, C, nullptr); CK(hipEventRecord(et[1], s2)); CK(hipStreamWaitEvent(s1, et[1], 0)); }, 2);
    run("T5 the same with no-timing events", [&](int) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, C, nullptr); CK(hipEventRecord(en[0], s1)); CK(hipStreamWaitEvent(s2, en[0], 0));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, C, nullptr); CK(hipEventRecord(en[1], s2)); CK(hipStreamWaitEvent(s1, en[1], 0)); }, 2);
    run("T6 fork to two streams and join (no-timing events)", [&](int) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, C, nullptr); CK(hipEventRecord(en[0], s1));
        CK(hipStreamWaitEvent(s2, en[0], 0)); CK(hipStreamWaitEvent(s3, en[0], 0));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, C, nullptr); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s3, C, nullptr);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, C, nullptr);
        CK(hipEventRecord(en[1], s2)); CK(hipEventRecord(en[2], s3)); CK(hipStreamWaitEvent(s1, en[1], 0)); CK(hipStreamWaitEvent(s1, en[2], 0));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, C, nullptr); }, 3);
    {
        int epoch = 0;
        run("T7 s1 kernel sets a device flag, s2 kernel (resident) polls it, and back", [&](int) {
            const int a = ++epoch, b = ++epoch;
            hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(64), 0, s1, flag, a - 1, C);      // waits for the previous iteration's s2 kernel
            hipLaunchKernelGGL(k_spin_set, dim3(1), dim3(64), 0, s1, 0LL, flag, a);
            hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(64), 0, s2, flag, a, C);
            hipLaunchKernelGGL(k_spin_set, dim3(1), dim3(64), 0, s2, 0LL, flag, b); (void)b; }, 2);
    }
    // host-visible completion: pinned flag written by a kernel vs hipStreamSynchronize
    {
        int *h; CK(hipHostMalloc((void **)&h, 64, hipHostMallocCoherent | hipHostMallocMapped)); *h = 0;
        int *dv; CK(hipHostGetDevicePointer((void **)&dv, h, 0));
        int ep = 0;
        double acc = 0;
        for (int i = 0; i < N + 20; ++i) {
            const double t0 = now();
            hipLaunchKernelGGL(k_spin_set, dim3(1), dim3(64), 0, s1, C, dv, ++ep);
            while (__atomic_load_n(h, __ATOMIC_ACQUIRE) != ep) {}
            if (i >= 20) acc += now() - t0;
        }
        printf("%-64s %7.2f us per iteration  (1 kernel of 5 us: overhead %6.2f us)\n", "T8 launch + poll a pinned flag the kernel writes", acc / N, acc / N - 5.0);
        acc = 0;
        for (int i = 0; i < N + 20; ++i) {
            const double t0 = now();
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, C, nullptr);
            CK(hipStreamSynchronize(s1));
            if (i >= 20) acc += now() - t0;
        }
        printf("%-64s %7.2f us per iteration  (1 kernel of 5 us: overhead %6.2f us)\n", "T9 launch + hipStreamSynchronize", acc / N, acc / N - 5.0);
    }
    return 0;
}
