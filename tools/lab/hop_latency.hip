// Development probe: what one cross-stream hand-over costs on the GPU timeline (kernel on stream A -> event ->
// hipStreamWaitEvent -> kernel on stream B -> event -> back to A), for the stream / event flavours the multi-rank CG loop
// could use; against the same kernels launched back to back on ONE stream.
// hipcc --offload-arch=gfx950 -O2 tools/lab/hop_latency.hip -o tools/lab/hop_latency && tools/lab/hop_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_touch(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }
static double run(hipStream_t A, hipStream_t B, unsigned evflags, int iters, double *d, bool pingpong)
{
    hipEvent_t e[2];
    hipEventCreateWithFlags(&e[0], evflags); hipEventCreateWithFlags(&e[1], evflags);
    auto body = [&] {
        for (int i = 0; i < iters; ++i) {
            hipLaunchKernelGGL(k_touch, dim3(1), dim3(256), 0, A, d, 256);
            if (pingpong) { hipEventRecord(e[0], A); hipStreamWaitEvent(B, e[0], 0); }
            hipLaunchKernelGGL(k_touch, dim3(1), dim3(256), 0, pingpong ? B : A, d, 256);
            if (pingpong) { hipEventRecord(e[1], B); hipStreamWaitEvent(A, e[1], 0); }
        }
    };
    body(); hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    body(); hipDeviceSynchronize();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    hipEventDestroy(e[0]); hipEventDestroy(e[1]);
    return us / iters;
}
int main()
{
    double *d; hipMalloc(&d, 256 * sizeof(double)); hipMemset(d, 0, 256 * sizeof(double));
    int least = 0, greatest = 0; hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStream_t A, Bn, Bh, Bd;
    hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&Bn, hipStreamNonBlocking);
    hipStreamCreateWithPriority(&Bh, hipStreamNonBlocking, greatest);
    hipStreamCreate(&Bd);
    const int iters = 2000;
    std::printf("{\"one_stream_two_kernels_us\": %.2f", run(A, A, hipEventDisableTiming, iters, d, false));
    std::printf(", \"hop_pair_nonblocking_disable_timing_us\": %.2f", run(A, Bn, hipEventDisableTiming, iters, d, true));
    std::printf(", \"hop_pair_nonblocking_default_events_us\": %.2f", run(A, Bn, hipEventDefault, iters, d, true));
    std::printf(", \"hop_pair_high_priority_disable_timing_us\": %.2f", run(A, Bh, hipEventDisableTiming, iters, d, true));
    std::printf(", \"hop_pair_blocking_stream_disable_timing_us\": %.2f", run(A, Bd, hipEventDisableTiming, iters, d, true));
    std::printf(", \"priority_range\": [%d, %d]}\n", least, greatest);
    return 0;
}
