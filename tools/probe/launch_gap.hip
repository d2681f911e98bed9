// probe: per-launch cost of dependent small kernels in one stream, plain launches against a captured graph
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/launch_gap tools/probe/launch_gap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void ka(float *p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
__global__ void kb(float *p) { if (threadIdx.x == 0) p[blockIdx.x] += 2.f; }
int main() {
    float *p; hipMalloc(&p, 4096 * 4); hipMemset(p, 0, 4096 * 4);
    hipStream_t st; hipStreamCreate(&st);
    const int PAIRS = 200, REPS = 20;
    for (int w = 0; w < 50; ++w) { hipLaunchKernelGGL(ka, dim3(100), dim3(256), 0, st, p); hipLaunchKernelGGL(kb, dim3(18), dim3(1024), 0, st, p); }
    hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < REPS; ++r)
        for (int i = 0; i < PAIRS; ++i) { hipLaunchKernelGGL(ka, dim3(100), dim3(256), 0, st, p); hipLaunchKernelGGL(kb, dim3(18), dim3(1024), 0, st, p); }
    hipStreamSynchronize(st);
    auto t1 = std::chrono::high_resolution_clock::now();
    printf("plain launches: %.2f us per kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / (2.0 * PAIRS * REPS));
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < PAIRS; ++i) { hipLaunchKernelGGL(ka, dim3(100), dim3(256), 0, st, p); hipLaunchKernelGGL(kb, dim3(18), dim3(1024), 0, st, p); }
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < REPS; ++r) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    t1 = std::chrono::high_resolution_clock::now();
    printf("graph of %d pairs: %.2f us per kernel\n", PAIRS, std::chrono::duration<double, std::micro>(t1 - t0).count() / (2.0 * PAIRS * REPS));
    for (int np : {1, 9, 18}) {   // small graphs launched often (one minibatch / one epoch / two epochs)
        hipGraph_t g2; hipGraphExec_t ge2;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < np; ++i) { hipLaunchKernelGGL(ka, dim3(100), dim3(256), 0, st, p); hipLaunchKernelGGL(kb, dim3(18), dim3(1024), 0, st, p); }
        hipStreamEndCapture(st, &g2);
        hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
        hipGraphLaunch(ge2, st); hipStreamSynchronize(st);
        const int L = 4000 / np;
        t0 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < L; ++r) hipGraphLaunch(ge2, st);
        hipStreamSynchronize(st);
        t1 = std::chrono::high_resolution_clock::now();
        printf("graph of %d pairs, %d launches: %.2f us per kernel\n", np, L, std::chrono::duration<double, std::micro>(t1 - t0).count() / (2.0 * np * L));
    }
    return 0;
}
