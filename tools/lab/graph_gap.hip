// Development probe: GPU-side gap between dependent kernels, stream launches vs a captured hipGraph.
// hipcc --offload-arch=gfx950 -O2 tools/lab/graph_gap.hip -o /tmp/graph_gap && /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_touch(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }
int main()
{
    const int n = 256, iters = 2000, per = 4;
    double *d; hipMalloc(&d, n * sizeof(double)); hipMemset(d, 0, n * sizeof(double));
    hipStream_t st; hipStreamCreate(&st);
    auto run_stream = [&] { for (int i = 0; i < iters * per; ++i) hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, st, d, n); };
    run_stream(); hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now(); run_stream(); hipStreamSynchronize(st);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("stream: %.2f us per kernel\n", ms * 1e3 / (iters * per));
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 32 * per; ++i) hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, st, d, n);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 8; ++i) hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters / 32; ++i) hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("graph : %.2f us per kernel\n", ms * 1e3 / ((iters / 32) * 32 * per));
    // single kernel duration via events
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st); for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, st, d, n); hipEventRecord(e1, st);
    hipStreamSynchronize(st); float f; hipEventElapsedTime(&f, e0, e1); printf("events: %.2f us per kernel (back to back)\n", f * 10.0);
    return 0;
}
