// Gap between dependent tiny kernels on one stream: plain launches against a captured hipGraph (gfx950, ROCm 7).
// Is a graph worth it for the ~40-kernel tail of a short row's commit+open?   hipcc --offload-arch=gfx950 -O2 graph_gap.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_tiny(unsigned* p, int spin) {
    unsigned v = p[threadIdx.x];
    for (int i = 0; i < spin; i++) v = v * 1664525u + 1013904223u;
    p[threadIdx.x] = v;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 40;
    for (int spin : {1, 2000}) {
        auto run_plain = [&]() { for (int i = 0; i < N; i++) k_tiny<<<64, 64, 0, s>>>(d, spin); };
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        run_plain();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int mode = 0; mode < 2; mode++) {
            for (int w = 0; w < 200; w++) { if (mode) hipGraphLaunch(ge, s); else run_plain(); hipStreamSynchronize(s); }
            const int reps = 300;
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; r++) { if (mode) hipGraphLaunch(ge, s); else run_plain(); hipStreamSynchronize(s); }
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            printf("spin %4d  %s: %7.1f us per %d dependent kernels = %.2f us each\n", spin, mode ? "graph " : "stream", us, N, us / N);
        }
    }
    // GPU-side gap when the host is AHEAD (the 40 launches are queued behind a 300-us kernel, as the tail of an MSM is
    // queued behind its accumulate): events around the 40 tiny kernels
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < N; i++) k_tiny<<<64, 64, 0, s>>>(d, 1);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int mode = 0; mode < 2; mode++) {
            float tot = 0;
            const int reps = 100;
            for (int r = 0; r < reps + 20; r++) {
                k_tiny<<<64, 64, 0, s>>>(d, 15000);
                hipEventRecord(a, s);
                if (mode) hipGraphLaunch(ge, s); else for (int i = 0; i < N; i++) k_tiny<<<64, 64, 0, s>>>(d, 1);
                hipEventRecord(b, s);
                hipStreamSynchronize(s);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (r >= 20) tot += ms;
            }
            printf("queued behind a long kernel, %s: %.1f us GPU time for %d tiny dependent kernels = %.2f us each\n",
                   mode ? "graph " : "stream", tot / reps * 1e3, N, tot / reps * 1e3 / N);
        }
    }
    return 0;
}
