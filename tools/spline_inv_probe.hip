// tools/spline_inv_probe.hip -- the spline flow's inverse alone, team form (16 walkers per workgroup) against the halves form
// (8 walkers per workgroup, held in both halves of the columns), four waves per tile: cycles per inverse without the proposal
// loop around it (developer tool, not part of the library).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I nnest_amd/csrc tools/spline_inv_probe.hip -o tools/bin/spline_inv_probe && tools/bin/spline_inv_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "spline_train_tile.h"
using namespace nnest;

static SplineShape make_shape(int D, int H, int B) {
    SplineShape s;
    memset(&s, 0, sizeof(s));
    s.D = D; s.H = H; s.B = B; s.K = 8; s.tail = 3.f;
    s.nu = D / 2; s.nl = D - s.nu;
    s.NTh = (s.nl + 15) / 16;
    s.NH = H / 16;
    s.SL = (s.nl + 3) / 4; s.SU = (s.nu + 3) / 4;
    s.aff_floats = (2 * s.NTh) * (2 * s.NTh) * 256 + 2 * s.NTh * 16;
    s.f1_floats = spl_cond_floats(s.NTh, s.NH, s.SU);
    s.f2_floats = spl_cond_floats(s.NTh, s.NH, s.SL);
    s.blk_floats = 2 * s.aff_floats + s.f1_floats + s.f2_floats + 4;
    s.image_floats = B * s.blk_floats;
    return s;
}

// FORM 0: team (spline_inverse_tile<.., 4>), 16 walkers; FORM 1: halves, 8 walkers
template <int NT, int NH, int FORM>
__global__ void __launch_bounds__(256) probe(const float *img, SplineShape sp, const float *x_in, float *out, int C, int S, long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds_buf[];
#ifdef PROBE_UNIFORM_WV
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tile = blockIdx.x;
#else
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, tile = blockIdx.x;
#endif
    f32x4 *xch = reinterpret_cast<f32x4 *>(lds_buf);
    float *ldred = reinterpret_cast<float *>(xch + 2 * 4 * NT * 64);   // (two exchange buffers: spline_inverse_tile_halves alternates)
    int xsel = 0;
    const int GW = FORM == 0 ? 16 : 8;
    const int w = lane & 15, row = tile * GW + (w & (GW - 1));
    const bool ok = row < C;
    f32x4 t[2][NT];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const float v = ok ? x_in[(size_t)row * 64 + (c * NT + k) * 4 + (lane >> 4)] : 0.f;
            t[c][k] = (f32x4){v, 0.5f * v, -v, 0.25f * v};
        }
    float acc = 0.f;
    const long long t0 = wall_clock64();
    for (int it = 0; it < S; ++it) {
        float ld;
        if (FORM == 0) ld = group_sum(spline_inverse_tile<NT, NH, 4>(img, sp, lane, t, wv, xch));
        else ld = group_sum(spline_inverse_tile_halves<NT, NH>(img, sp, lane, t, wv, xch, xsel));
        if (lane < 16) ldred[wv * 16 + lane] = ld;
        spl_team_barrier();
        ld = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) ld += ldred[k * 16 + w];
        spl_team_barrier();
        acc += 0.25f * ld;
#pragma unroll
        for (int c = 0; c < 2; ++c)   // keep the state bounded and dependent on the step
#pragma unroll
            for (int k = 0; k < NT; ++k) t[c][k] = t[c][k] * 0.5f + (f32x4){0.1f, -0.2f, 0.3f, 0.05f};
    }
    const long long t1 = wall_clock64();
    if (ok && (lane >> 4) == 0 && (GW == 16 || w < 8) && wv == 0) out[row] = acc + t[0][0].x + t[1][NT - 1].w;
    if (threadIdx.x == 0 && tile == 0) cyc[0] = t1 - t0;
}

template <int FORM>
static void run(const float *img, const SplineShape &sp, const float *x, float *out, int C, int S, long long *cyc, std::vector<float> &host) {
    const int GW = FORM == 0 ? 16 : 8, grid = (C + GW - 1) / GW;
    const size_t lds = (size_t)(2 * 4 * 2 * 64 * 4 + 64) * sizeof(float);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<2, 1, FORM>), dim3(grid), dim3(256), lds, 0, img, sp, x, out, C, S, cyc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    long long c;
    hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost);
    hipMemcpy(host.data(), out, (size_t)C * sizeof(float), hipMemcpyDeviceToHost);
    printf("%s: %d workgroups, %d inverses: %.3f ms = %.2f us per inverse (wall-clock ticks per inverse %.0f)\n", FORM == 0 ? "team  " : "halves", grid,
           S, best, best * 1e3 / S, (double)c / S);
}

int main(int argc, char **argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 1000, S = argc > 2 ? atoi(argv[2]) : 251;
    const SplineShape sp = make_shape(50, 16, 3);
    std::vector<float> himg(sp.image_floats);
    srand(1);
    for (auto &v : himg) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
    std::vector<float> hx((size_t)C * 64);
    for (auto &v : hx) v = ((float)rand() / RAND_MAX - 0.5f) * 2.f;
    float *img, *x, *out;
    long long *cyc;
    hipMalloc(&img, himg.size() * 4); hipMalloc(&x, hx.size() * 4); hipMalloc(&out, (size_t)C * 4); hipMalloc(&cyc, 64);
    hipMemcpy(img, himg.data(), himg.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> o0(C), o1(C);
    run<0>(img, sp, x, out, C, S, cyc, o0);
    run<1>(img, sp, x, out, C, S, cyc, o1);
    int diff = 0;
    for (int i = 0; i < C; ++i) diff += memcmp(&o0[i], &o1[i], 4) != 0;
    printf("walkers whose result differs between the forms: %d of %d\n", diff, C);
    return 0;
}
