// tools/spline_mh_probe.hip -- the spline proposal kernel's two small-population forms alone (team: 16 walkers per workgroup;
// pair: 8 walkers held in both halves of the columns), a few shapes only so that it compiles in half a minute instead of the
// library's minutes: fixed step, in-kernel noise, random weights (developer tool, not part of the library).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-mllvm -disable-machine-licm] -I nnest_amd/csrc tools/spline_mh_probe.hip -o tools/bin/spline_mh_probe
// Without the flag the pair form runs 10.2 ms per 1000 x 250 launch, with it 6.6 (team form 7.9 / 8.0): nnest_spline_mh.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "nnest_spline_mh.hip"   // the two forms as the library compiles them (and mh_body.h inside namespace nnest)
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

int main(int argc, char **argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 1000, S = argc > 2 ? atoi(argv[2]) : 250, D = 50;
    const SplineShape sp = make_shape(D, 16, 3);
    std::vector<float> himg(sp.image_floats);
    srand(1);
    for (auto &v : himg) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
    std::vector<float> hz((size_t)C * D);
    for (auto &v : hz) v = ((float)rand() / RAND_MAX - 0.5f) * 0.5f;
    std::vector<double> hl(C, -1e30);
    float *img, *z, *x;
    double *logl;
    int *nacc, *ncall;
    hipMalloc(&img, himg.size() * 4); hipMalloc(&z, hz.size() * 4); hipMalloc(&x, hz.size() * 4); hipMalloc(&logl, C * 8);
    hipMalloc(&nacc, C * 4); hipMalloc(&ncall, C * 4);
    hipMemcpy(img, himg.data(), himg.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> res[2];
    for (int form = 0; form < 2; ++form) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemcpy(z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(logl, hl.data(), C * 8, hipMemcpyHostToDevice);
            MhArgs a{};
            a.s.D = D;
            a.z = z; a.x = x; a.logl = logl; a.loglstar = -1e300; a.step_size = 0.05f; a.steps = S; a.C = C; a.flags = 0;
            a.like.id = 0; a.like.scale = 5.0f;
            a.seed = 7; a.n_accept = nacc; a.n_call = ncall;
            SplArgs q = {img, sp};
            const size_t ldsb = (size_t)(((4 * 16 * (D + 1) + 3) & ~3) + 2 * 4 * 2 * 64 * 4 + 2 * 4 * 16 + sp.B * 2 * spl_cond_hidden_floats(2, 1)) * sizeof(float);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (form == 0) hipLaunchKernelGGL((spline_mh_kernel_team<2, 1, 4, false>), dim3((C + 15) / 16), dim3(256), ldsb, 0, a, q);
            else hipLaunchKernelGGL((spline_mh_kernel_pair<2, 1, false>), dim3((C + 7) / 8), dim3(256), ldsb, 0, a, q);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        res[form].resize(hz.size());
        hipMemcpy(res[form].data(), x, hz.size() * 4, hipMemcpyDeviceToHost);
        std::vector<int> na(C);
        hipMemcpy(na.data(), nacc, C * 4, hipMemcpyDeviceToHost);
        long tot = 0;
        for (int v : na) tot += v & 0xFFFFF;
        printf("%s: %d walkers x %d steps: %.3f ms per launch = %.2f us per step (accepted %.3f)\n", form == 0 ? "team" : "pair", C, S, best,
               best * 1e3 / S, (double)tot / ((double)C * S));
    }
    size_t diff = 0;
    for (size_t i = 0; i < res[0].size(); ++i) diff += memcmp(&res[0][i], &res[1][i], 4) != 0;
    printf("coordinates of the final x that differ between the forms: %zu of %zu\n", diff, res[0].size());
    return 0;
}
