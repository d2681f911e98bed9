// tools/ablate_mh.hip -- timing ablation of the MH step (developer tool, not part of the library).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I nnest_amd/csrc tools/ablate_mh.hip -o /tmp/ablate_mh && /tmp/ablate_mh
// Each variant removes one component of the step while keeping its outputs live (cdna guide rule 17).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "flow_tile.h"
using namespace nnest;

#ifndef LTV
#define LTV 1
#endif
template <int NT, int NH, bool RNG, bool FLOW, bool LIKE, bool F64LIKE>
__global__ void __launch_bounds__(256) step_kernel(const float *gimg, FlowShape s, float *z_io, int C, int S, float *sink) {
    extern __shared__ __attribute__((aligned(16))) float lds_img[];
    for (int i = threadIdx.x; i < s.image_floats / 4; i += blockDim.x)
        reinterpret_cast<f32x4 *>(lds_img)[i] = reinterpret_cast<const f32x4 *>(gimg)[i];
    __syncthreads();
    const float *img = lds_img;
    const int lane = threadIdx.x & 63, tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile * 16 >= C) return;
    const int w = lane & 15, g = lane >> 4, row = tile * 16 + w;
    const bool ok = row < C;
    const int D = s.D;
    f32x4 z[2][NT], x[2][NT];
    load_tile<NT>(z_io, row, ok, D, lane, z);
    for (int c = 0; c < 2; ++c) for (int t = 0; t < NT; ++t) x[c][t] = z[c][t];
    float ld = 0.f;
    double logl = -1e30;
    Xoshiro128 rn = xoshiro_seed(1, row, g, 0), ru = xoshiro_seed(1, row, 99, 1);
    int nacc = 0;
    for (int it = 0; it < S; ++it) {
        f32x4 zp[2][NT], xp[2][NT];
        float u = 0.5f;
        if (RNG) {
            for (int t = 0; t < NT; ++t) {
                float n[8];
                xoshiro_normal8(rn, n);
                zp[0][t].x = z[0][t].x + n[0] * 0.01f; zp[1][t].x = z[1][t].x + n[1] * 0.01f;
                zp[0][t].y = z[0][t].y + n[2] * 0.01f; zp[1][t].y = z[1][t].y + n[3] * 0.01f;
                zp[0][t].z = z[0][t].z + n[4] * 0.01f; zp[1][t].z = z[1][t].z + n[5] * 0.01f;
                zp[0][t].w = z[0][t].w + n[6] * 0.01f; zp[1][t].w = z[1][t].w + n[7] * 0.01f;
            }
            u = xoshiro_uniform(ru);
        } else {
            for (int c = 0; c < 2; ++c) for (int t = 0; t < NT; ++t) zp[c][t] = z[c][t] * 1.0001f;
        }
        for (int c = 0; c < 2; ++c) for (int t = 0; t < NT; ++t) xp[c][t] = zp[c][t];
        float ldp = 0.f;
        if (FLOW) ldp = group_sum(flow_inverse_tile<NT, NH, LTV>(img, s.net_floats, s.B, s.L, lane, xp));
        const int inb = inbox_tile<NT>(xp, lane);
        float ratio = fminf(__expf(inb ? ldp - ld : -INFINITY), 1.f);
        bool pre = ok && u < ratio;
        double lp = 0;
        if (LIKE) {
            if (F64LIKE) { LikeSpec lk; lk.id = 0; lk.scale = 5.0f; lp = loglike_tile<NT>(lk, D, lane, xp); }
            else {
                float ss = 0.f;
                for (int c = 0; c < 2; ++c) for (int t = 0; t < NT; ++t) ss += xp[c][t].x * xp[c][t].y + xp[c][t].z * xp[c][t].w;
                lp = (double)group_sum(ss);
            }
        }
        bool acc = pre && lp > -1e29;
        nacc += acc;
        for (int c = 0; c < 2; ++c) for (int t = 0; t < NT; ++t) {
            z[c][t].x = acc ? zp[c][t].x : z[c][t].x; z[c][t].y = acc ? zp[c][t].y : z[c][t].y;
            z[c][t].z = acc ? zp[c][t].z : z[c][t].z; z[c][t].w = acc ? zp[c][t].w : z[c][t].w;
            x[c][t].x = acc ? xp[c][t].x : x[c][t].x; x[c][t].y = acc ? xp[c][t].y : x[c][t].y;
            x[c][t].z = acc ? xp[c][t].z : x[c][t].z; x[c][t].w = acc ? xp[c][t].w : x[c][t].w;
        }
        ld = acc ? ldp : ld;
        logl = acc ? lp : logl;
    }
    store_tile<NT>(z_io, row, ok, D, lane, z);
    if (ok && g == 0) sink[row] = (float)logl + ld + nacc + x[0][0].x;
}

template <bool RNG, bool FLOW, bool LIKE, bool F64>
float run(const float *img, FlowShape s, float *z, int C, int S, float *sink, int wpb) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int ntiles = (C + 15) / 16, grid = (ntiles + wpb - 1) / wpb;
    size_t lds = (size_t)s.image_floats * 4;
    auto k = step_kernel<2, 1, RNG, FLOW, LIKE, F64>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(64 * wpb), lds, 0, img, s, z, C, S, sink);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(64 * wpb), lds, 0, img, s, z, C, S, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main(int argc, char **argv) {
    int C = argc > 1 ? atoi(argv[1]) : 1000, S = argc > 2 ? atoi(argv[2]) : 250, wpb = argc > 3 ? atoi(argv[3]) : 1;
    FlowShape s; s.D = 50; s.H = 16; s.B = 3; s.L = 1; s.NT = 2; s.NH = 1;
    s.net_floats = frag_net_floats(2, 1, 1); s.image_floats = 6 * s.net_floats; s.net_params = 0;
    std::vector<float> h(s.image_floats);
    for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    float *img, *z, *sink;
    hipMalloc(&img, h.size() * 4); hipMemcpy(img, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hz((size_t)C * 50);
    for (auto &v : hz) v = (rand() / (float)RAND_MAX - 0.5f);
    hipMalloc(&z, hz.size() * 4); hipMemcpy(z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&sink, C * 4);
    double us = 1e3 / S;
    printf("C=%d S=%d wpb=%d  (us per step per wave)\n", C, S, wpb);
    printf("full            %.3f\n", run<true, true, true, true>(img, s, z, C, S, sink, wpb) * us);
    printf("no rng          %.3f\n", run<false, true, true, true>(img, s, z, C, S, sink, wpb) * us);
    printf("no like         %.3f\n", run<true, true, false, true>(img, s, z, C, S, sink, wpb) * us);
    printf("f32 like        %.3f\n", run<true, true, true, false>(img, s, z, C, S, sink, wpb) * us);
    printf("no flow         %.3f\n", run<true, false, true, true>(img, s, z, C, S, sink, wpb) * us);
    printf("flow only       %.3f\n", run<false, true, false, true>(img, s, z, C, S, sink, wpb) * us);
    printf("rng only        %.3f\n", run<true, false, false, true>(img, s, z, C, S, sink, wpb) * us);
    return 0;
}
