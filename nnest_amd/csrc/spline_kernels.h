// spline_kernels.h -- kernels of the neural-spline flow; part of the nnest_kernels.hip translation unit (included at
// its end, inside namespace nnest) so that the proposal loop (mh_body), the noise streams, the likelihood / prior tile
// code and the row-major tile I/O are the SAME code the RealNVP path runs.
//
//   spline_pass_kernel   forward / inverse / log_probs / inverse + box + likelihood   (networks.py:24-42, :71-76)
//   spline_mh_kernel     Sampler._mcmc_sample's constrained Metropolis loop with the spline inverse (sampler.py:229-463);
//                        _team: four waves per 16-walker tile (small populations); the 8-walker PAIR form, which takes over at
//                        x_dim > 32 while its tiles fit one per CU, is compiled in nnest_spline_mh.hip
//
// One wave per 16 walkers.  The state lives in the parity-class tiles of flow_tile.h (what mh_body, loglike_tile,
// inbox_tile and load/store_tile work on); around each flow evaluation it is re-laid into the contiguous-halves tiles
// of spline_tile.h through a 16 x (D+1) float LDS buffer private to the wave.  The weight image (258 KB at x_dim 50) is
// read from global memory / L2: every fragment load is one coalesced 256-byte line per wave.
#pragma once
#include "spline_train_tile.h"   // (already in at file scope: the fragment-prefetch helpers and the paired coupling of the 8-walker form)

// (struct SplArgs: nnest_internal.h)

template <int NT, int NH>
__global__ void __launch_bounds__(256) spline_pass_kernel(PassArgs a, SplArgs q) {
    extern __shared__ __attribute__((aligned(16))) float lds_buf[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    float *buf = lds_buf + (size_t)wave * 16 * (q.sp.D + 1);
    const int ntiles = (a.N + 15) >> 4;
    const int w = lane & 15, g = lane >> 4;
    const int D = q.sp.D;
    for (int tile = blockIdx.x * wpb + wave; tile < ntiles; tile += gridDim.x * wpb) {
        const int row = tile * 16 + w;
        const bool ok = row < a.N;
        f32x4 xs[2][NT], sp[2][NT];
        load_tile<NT>(a.in, row, ok, D, lane, xs);
        spl_from_parity<NT>(buf, D, q.sp.nl, lane, xs, sp);
        float ld;
        if (a.mode == PASS_FORWARD || a.mode == PASS_LOGPROB) ld = spline_forward_tile<NT, NH>(q.img, q.sp, lane, sp);
        else ld = spline_inverse_tile<NT, NH>(q.img, q.sp, lane, sp);
        ld = group_sum(ld);
        spl_to_parity<NT>(buf, D, q.sp.nl, lane, sp, xs);
        if (a.mode == PASS_LOGPROB) {
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int tau = 0; tau < NT; ++tau) ss += base_E4(xs[c][tau], q.sp.base_beta);
            ss = group_sum(ss);
            if (ok && g == 0) a.out[row] = -ss + q.sp.base_const * (float)D + ld;
            continue;
        }
        if (a.out) store_tile<NT>(a.out, row, ok, D, lane, xs);
        if (a.logdet && ok && g == 0) a.logdet[row] = ld;
        if (a.mode == PASS_INVERSE_LOGLIKE) {
            int inb = inbox_tile<NT>(xs, lane);
            double ll = loglike_tile<NT>(a.like, D, lane, xs);
            if (ok && g == 0) {
                a.logl[row] = ll;
                if (a.inbox) a.inbox[row] = inb;
            }
        }
    }
}

template <int NT, int NH>
struct SplineInverse {
    const float *img;
    SplineShape sp;
    float *buf;
    int lane;
#ifdef NNEST_STAMP
    unsigned long long t_mlp = 0, t_xch = 0, t_upd = 0;
#endif
    __device__ __forceinline__ float operator()(f32x4 (&xs)[2][NT]) const {
        f32x4 t[2][NT];
        spl_from_parity<NT>(buf, sp.D, sp.nl, lane, xs, t);
        const float ld = spline_inverse_tile<NT, NH>(img, sp, lane, t);
        spl_to_parity<NT>(buf, sp.D, sp.nl, lane, t, xs);
        return ld;
    }
};

template <int NT, int NH, bool DBG>
__global__ void __launch_bounds__(256) spline_mh_kernel(MhArgs a, SplArgs q) {
    extern __shared__ __attribute__((aligned(16))) float lds_buf[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int tile = blockIdx.x * wpb + wave;
    if (tile >= ((a.C + 15) >> 4)) return;
    SplineInverse<NT, NH> inv = {q.img, q.sp, lds_buf + (size_t)wave * 16 * (q.sp.D + 1), lane};
    XoshiroNoise<NT> noise;
    noise.init(a.seed, a.walker_offset + (uint64_t)(tile * 16 + (lane & 15)), lane >> 4, q.sp.D);
    mh_body<NT, DBG>(a, tile, lane, inv, noise, true);
}

// (the team form and the pair form of the proposal kernel: nnest_spline_mh.hip)

bool spline_shape_supported(const SplineShape &s) {
    if (s.K != SPL_K) return false;
    if (s.NH == 1) return s.NTh >= 1 && s.NTh <= 4;
    if (s.NH == 2) return s.NTh >= 1 && s.NTh <= 2;
    return false;
}

#define DISPATCH_SPLINE(FN, sp, ...)                                               \
    do {                                                                           \
        const int key__ = (sp).NTh * 10 + (sp).NH;                                 \
        switch (key__) {                                                           \
            case 11: return FN<1, 1>(__VA_ARGS__);                                 \
            case 21: return FN<2, 1>(__VA_ARGS__);                                 \
            case 31: return FN<3, 1>(__VA_ARGS__);                                 \
            case 41: return FN<4, 1>(__VA_ARGS__);                                 \
            case 12: return FN<1, 2>(__VA_ARGS__);                                 \
            case 22: return FN<2, 2>(__VA_ARGS__);                                 \
            default: return hipErrorInvalidConfiguration;                          \
        }                                                                          \
    } while (0)

template <int NT, int NH>
static hipError_t launch_spline_pass_t(const PassArgs &a, const SplArgs &q, int num_cu, hipStream_t st) {
    const int ntiles = (a.N + 15) / 16;
    int block, grid;
    pick_geometry(ntiles, num_cu, 4, &block, &grid);
    if (grid > 8 * num_cu) grid = 8 * num_cu;
    const size_t lds = (size_t)(block / 64) * 16 * (q.sp.D + 1) * sizeof(float);
    hipLaunchKernelGGL((spline_pass_kernel<NT, NH>), dim3(grid), dim3(block), lds, st, a, q);
    return hipGetLastError();
}

// Which form of the proposal kernel runs for C walkers under `flags` (nnest_spline_mh_form_for, include/nnest_hip.h):
//   NNEST_SPLINE_MH_PAIR  8-walker tiles, each walker in both halves of the columns (nnest_spline_mh.hip): x_dim > 32 (two or more
//                         16-slot tiles per half -- below that a wave of the team form has one super-tile per coupling already),
//                         a fixed step or the batch-wide rule (the per-16-walker rule's group is the team form's tile), tiles that
//                         fit one per CU; NNEST_SPLINE_MH_FORM=team in the environment keeps the team form (tests)
//   NNEST_SPLINE_MH_TEAM  four waves per 16-walker tile while the tiles fit two per CU
//   NNEST_SPLINE_MH_WAVE  one wave per 16-walker tile
// -1: the batch-wide rule on a grid that would not be resident
int spline_mh_form(const SplineShape &sp, int C, int flags, int num_cu) {
    const int ntiles = (C + 15) / 16, ntiles8 = (C + 7) / 8;
    const bool batch = (flags & NNEST_MH_DYNAMIC_BATCH) != 0;
    if (batch && ntiles > num_cu) return -1;
    if (sp.NTh >= 2) {
        static const bool team_only = [] { const char *e = getenv("NNEST_SPLINE_MH_FORM"); return e && !strcmp(e, "team"); }();
        const bool group_rule = (flags & NNEST_MH_DYNAMIC_STEP) && !batch;
        if (!team_only && !group_rule && ntiles8 <= num_cu) return NNEST_SPLINE_MH_PAIR;
    }
    if (ntiles <= 2 * num_cu) return NNEST_SPLINE_MH_TEAM;
    int block, grid;
    pick_geometry(ntiles, num_cu, 4, &block, &grid);
    if (batch && grid > num_cu) return -1;
    return NNEST_SPLINE_MH_WAVE;
}

template <int NT, int NH>
static hipError_t launch_spline_mh_t(const MhArgs &a, const SplArgs &q, int num_cu, hipStream_t st) {
    const int ntiles = (a.C + 15) / 16;
    const bool dbg = a.noise_dz || a.hist_x || a.hist_logl;
    // (eight waves per tile measured slower than four: 12.0 vs 8.0 ms at x_dim 50 -- the 512-thread workgroup halves the
    // register budget and the redundant trunk / affine work grows)
    const int form = spline_mh_form(q.sp, a.C, a.flags, num_cu);
    if (form < 0) return hipErrorInvalidConfiguration;  // batch rule: resident grid only
    if (form == NNEST_SPLINE_MH_PAIR) return launch_spline_mh_pair(a, q, dbg, st);   // nnest_spline_mh.hip
    if (form == NNEST_SPLINE_MH_TEAM) return launch_spline_mh_team(a, q, dbg, st);   // nnest_spline_mh.hip
    int block, grid;
    pick_geometry(ntiles, num_cu, 4, &block, &grid);
    const size_t lds = (size_t)(block / 64) * 16 * (q.sp.D + 1) * sizeof(float);
    if (dbg)
        hipLaunchKernelGGL((spline_mh_kernel<NT, NH, true>), dim3(grid), dim3(block), lds, st, a, q);
    else
        hipLaunchKernelGGL((spline_mh_kernel<NT, NH, false>), dim3(grid), dim3(block), lds, st, a, q);
    return hipGetLastError();
}

hipError_t launch_spline_pass(const float *img, const SplineShape &sp, int mode, const float *in, float *out, float *logdet,
                              double *logl, int *inbox, int N, const LikeSpec &like, int num_cu, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    PassArgs a{};
    a.mode = mode; a.in = in; a.out = out; a.logdet = logdet; a.logl = logl; a.inbox = inbox; a.N = N; a.like = like;
    a.s.D = sp.D;
    SplArgs q = {img, sp};
    DISPATCH_SPLINE(launch_spline_pass_t, sp, a, q, num_cu, st);
}

hipError_t launch_spline_mh(const float *img, const SplineShape &sp, const LikeSpec &like, float *z, float *x, double *logl,
                            double loglstar, float step_size, int steps, int C, int flags, const float *noise_dz,
                            const float *noise_u, uint64_t seed, uint64_t walker_offset, float *hist_x, double *hist_logl,
                            int *n_accept, int *n_call, float *scale_out, unsigned long long *sync, int num_cu, hipStream_t st) {
    if (C <= 0) return hipSuccess;
    if ((flags & NNEST_MH_DYNAMIC_BATCH) && !sync) return hipErrorInvalidValue;
    MhArgs a{};
    a.sync = sync;
    a.sync_err = sync ? reinterpret_cast<int *>(sync + mh_sync_words(steps)) : nullptr;
    a.s.D = sp.D;
    a.z = z; a.x = x; a.logl = logl; a.loglstar = loglstar; a.step_size = step_size; a.steps = steps; a.C = C; a.flags = flags;
    a.like = like; a.noise_dz = noise_dz; a.noise_u = noise_u; a.seed = seed; a.walker_offset = walker_offset;
    a.hist_x = hist_x; a.hist_logl = hist_logl; a.n_accept = n_accept; a.n_call = n_call; a.scale_out = scale_out;
    SplArgs q = {img, sp};
    if (a.x && a.n_accept && !(a.x0 = mh_first_x_buffer((size_t)C * sp.D, st))) return hipErrorOutOfMemory;
    hipError_t e = [&]() -> hipError_t { DISPATCH_SPLINE(launch_spline_mh_t, sp, a, q, num_cu, st); }();
    if (e == hipSuccess) e = launch_mh_zero_other_sync(a, st);
    return e != hipSuccess ? e : launch_mh_all_moved(a, st);
}
