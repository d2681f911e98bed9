// nnest_kernels.hip -- inference-side kernels of the nnest hot path for MI355X (gfx950):
//   repack_fragments_kernel   packed state_dict weights -> MFMA A-fragment image
//   flow_pass_kernel          K1 forward / K2 inverse / log_probs / K3 fused inverse + box prior + loglike
//   loglike_kernel            K6 batched analytic likelihoods
//   mh_kernel                 K4 persistent multi-step constrained Metropolis (Sampler._mcmc_sample)
//   fill_noise_kernel         the in-kernel proposal noise as arrays (tests)
// Launch wrappers (C++, used by nnest_abi.hip) are at the bottom.  See flow_tile.h for the data layout.
#include "flow_tile.h"
#include "nnest_internal.h"

namespace nnest {

// ------------------------------------------------------------------------------------------------
// repack: one thread per float of the fragment image
// ------------------------------------------------------------------------------------------------
__global__ void repack_fragments_kernel(const float *__restrict__ packed, float *__restrict__ img, FlowShape s) {
    const int NT = s.NT, NH = s.NH, L = s.L, D = s.D, H = s.H;
    const int total = s.image_floats;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int bn = idx / s.net_floats, o = idx - bn * s.net_floats;
        int b = bn >> 1, net = bn & 1;
        const float *p = packed + ((size_t)b * 2 + net) * s.net_params;
        const int pc = (b + 1) & 1, pt = b & 1;  // conditioning / transformed parity class of block b
        const int pW0 = 0, pb0 = H * D, phid = H * D + H, pWo = H * D + H + L * (H * H + H), pbo = pWo + D * H;
        float v = 0.f;
        if (o < frag_off_L2(NT, NH)) {  // L1 [ht][tau][r][lane]
            int lane = o & 63, q = o >> 6, r = q & 3, tau = (q >> 2) % NT, ht = (q >> 2) / NT;
            int g = lane >> 4, i = lane & 15;
            int j = 16 * ht + i, d = 2 * (16 * tau + 4 * g + r) + pc;
            if (d < D) v = p[pW0 + j * D + d];
        } else if (o < frag_off_L3(NT, NH, L)) {  // L2 [l][hto][hti][r][lane]
            int oo = o - frag_off_L2(NT, NH);
            int lane = oo & 63, q = oo >> 6, r = q & 3, hti = (q >> 2) % NH, hto = ((q >> 2) / NH) % NH,
                l = (q >> 2) / (NH * NH);
            int g = lane >> 4, i = lane & 15;
            v = p[phid + l * (H * H + H) + (16 * hto + i) * H + 16 * hti + 4 * g + r];
        } else if (o < frag_off_b1(NT, NH, L)) {  // L3 [tau][ht][r][lane]
            int oo = o - frag_off_L3(NT, NH, L);
            int lane = oo & 63, q = oo >> 6, r = q & 3, ht = (q >> 2) % NH, tau = (q >> 2) / NH;
            int g = lane >> 4, i = lane & 15;
            int d = 2 * (16 * tau + i) + pt;
            if (d < D) v = p[pWo + d * H + 16 * ht + 4 * g + r];
        } else if (o < frag_off_b2(NT, NH, L)) {
            v = p[pb0 + (o - frag_off_b1(NT, NH, L))];
        } else if (o < frag_off_b3(NT, NH, L)) {
            int oo = o - frag_off_b2(NT, NH, L), l = oo / (16 * NH), j = oo % (16 * NH);
            v = p[phid + l * (H * H + H) + H * H + j];
        } else {
            int sl = o - frag_off_b3(NT, NH, L), d = 2 * sl + pt;
            if (d < D) v = p[pbo + d];
        }
        img[idx] = v;
    }
}

// cooperative copy of the fragment image into LDS (float4, coalesced)
__device__ __forceinline__ void stage_image(float *lds, const float *__restrict__ img, int nfloats) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(img);
    f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
    for (int i = threadIdx.x; i < nfloats / 4; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// single-pass kernels
// ------------------------------------------------------------------------------------------------
struct PassArgs {
    const float *img;
    FlowShape s;
    int mode;  // PASS_*
    const float *in;
    float *out;
    float *logdet;
    double *logl;
    int *inbox;
    int N;
    int like_id;
    float like_scale;
};

template <int NT, int NH, bool WLDS>
__global__ void __launch_bounds__(256) flow_pass_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_img[];
    const float *img = a.img;
    if (WLDS) {
        stage_image(lds_img, a.img, a.s.image_floats);
        img = lds_img;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int ntiles = (a.N + 15) >> 4;
    const int w = lane & 15, g = lane >> 4;
    for (int tile = blockIdx.x * wpb + wave; tile < ntiles; tile += gridDim.x * wpb) {
        const int row = tile * 16 + w;
        const bool ok = row < a.N;
        f32x4 xs[2][NT];
        load_tile<NT>(a.in, row, ok, a.s.D, lane, xs);
        float ld;
        if (a.mode == PASS_FORWARD || a.mode == PASS_LOGPROB)
            ld = flow_forward_tile<NT, NH>(img, a.s.net_floats, a.s.B, a.s.L, lane, xs);
        else
            ld = flow_inverse_tile<NT, NH>(img, a.s.net_floats, a.s.B, a.s.L, lane, xs);
        ld = group_sum(ld);
        if (a.mode == PASS_LOGPROB) {
            // MVN(0,I).log_prob(u) + logdet  (networks.py:51-57, :71-76)
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int tau = 0; tau < NT; ++tau) {
                    f32x4 v = xs[c][tau];
                    ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
            ss = group_sum(ss);
            if (ok && g == 0) a.out[row] = -0.5f * ss - 0.91893853320467274f * (float)a.s.D + ld;
            continue;
        }
        if (a.out) store_tile<NT>(a.out, row, ok, a.s.D, lane, xs);
        if (a.logdet && ok && g == 0) a.logdet[row] = ld;
        if (a.mode == PASS_INVERSE_LOGLIKE) {
            int inb = inbox_tile<NT>(xs);
            double ll = loglike_tile<NT>(a.like_id, a.like_scale, a.s.D, lane, xs);
            if (ok && g == 0) {
                a.logl[row] = ll;
                if (a.inbox) a.inbox[row] = inb;
            }
        }
    }
}

// K6: likelihood only (no flow): reuses the tile layout so the arithmetic is the same code as the fused path
template <int NT>
__global__ void __launch_bounds__(256) loglike_kernel(const float *__restrict__ x, double *__restrict__ logl, int N, int D,
                                                      int like_id, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int ntiles = (N + 15) >> 4;
    for (int tile = blockIdx.x * wpb + wave; tile < ntiles; tile += gridDim.x * wpb) {
        const int row = tile * 16 + (lane & 15);
        const bool ok = row < N;
        f32x4 xs[2][NT];
        load_tile<NT>(x, row, ok, D, lane, xs);
        double ll = loglike_tile<NT>(like_id, scale, D, lane, xs);
        if (ok && (lane >> 4) == 0) logl[row] = ll;
    }
}

// ------------------------------------------------------------------------------------------------
// K4: persistent constrained Metropolis (Sampler._mcmc_sample hard-constraint branch, sampler.py:229-463)
// One wave = 16 walkers = one step-size adaptation group.  State (z, x, logdet, logl) stays in registers
// for all `steps`; the only global traffic is the start/end state (plus optional recorded noise / history).
// ------------------------------------------------------------------------------------------------
struct MhArgs {
    const float *img;
    FlowShape s;
    float *z;
    float *x;
    double *logl;
    double loglstar;
    float step_size;
    int steps;
    int C;
    int flags;
    int like_id;
    float like_scale;
    const float *noise_dz;
    const float *noise_u;
    uint64_t seed;
    uint64_t walker_offset;
    float *hist_x;
    double *hist_logl;
    int *n_accept;
    int *n_call;
    float *scale_out;
};

template <int NT, int NH, bool WLDS>
__global__ void __launch_bounds__(256) mh_kernel(MhArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_img[];
    const float *img = a.img;
    if (WLDS) {
        stage_image(lds_img, a.img, a.s.image_floats);
        img = lds_img;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int tile = blockIdx.x * wpb + wave;
    const int ntiles = (a.C + 15) >> 4;
    if (tile >= ntiles) return;
    const int w = lane & 15, g = lane >> 4;
    const int row = tile * 16 + w;
    const bool ok = row < a.C;
    const int D = a.s.D, S = a.steps;
    const uint64_t walker = a.walker_offset + (uint64_t)row;
    const int nvalid = min(16, a.C - tile * 16);  // walkers in this adaptation group

    f32x4 z[2][NT], x[2][NT];
    load_tile<NT>(a.z, row, ok, D, lane, z);
    // x = f^-1(z), log_det_J  (sampler.py:266, :295; the per-step re-inversion of the current z is
    // value-identical and therefore carried instead)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t) x[c][t] = z[c][t];
    float ld = group_sum(flow_inverse_tile<NT, NH>(img, a.s.net_floats, a.s.B, a.s.L, lane, x));
    double logl = ok ? a.logl[row] : 0.0;
    double scale = (double)a.step_size;  // python float in the reference (sampler.py:255, :428-431)
    int accept = 0, reject = 0, n_acc = 0, n_call = 0;

    if (a.hist_x) store_tile<NT>(a.hist_x, (long)row * (S + 1), ok, D, lane, x);
    if (a.hist_logl && ok && g == 0) a.hist_logl[(size_t)row * (S + 1)] = logl;

    for (int it = 1; it <= S; ++it) {
        // proposal z' = z + randn * scale  (sampler.py:310, :316); float32 like torch
        const float fs = (float)scale;
        f32x4 zp[2][NT], xp[2][NT];
        float u;
        if (a.noise_dz) {
            f32x4 dz[2][NT];
            load_tile<NT>(a.noise_dz + (size_t)(it - 1) * a.C * D, row, ok, D, lane, dz);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < NT; ++t) zp[c][t] = z[c][t] + dz[c][t] * fs;
            u = ok ? a.noise_u[(size_t)(it - 1) * a.C + row] : 1.f;
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                // dims 32t+8g+[0,4) and +[4,8): (c0r0,c1r0,c0r1,c1r1) and (c0r2,c1r2,c0r3,c1r3)
                f32x4 n0 = noise_normal4(a.seed, walker, (uint32_t)it, (uint32_t)(8 * t + 2 * g), NOISE_STREAM_DZ);
                f32x4 n1 = noise_normal4(a.seed, walker, (uint32_t)it, (uint32_t)(8 * t + 2 * g + 1), NOISE_STREAM_DZ);
                zp[0][t].x = z[0][t].x + n0.x * fs; zp[1][t].x = z[1][t].x + n0.y * fs;
                zp[0][t].y = z[0][t].y + n0.z * fs; zp[1][t].y = z[1][t].y + n0.w * fs;
                zp[0][t].z = z[0][t].z + n1.x * fs; zp[1][t].z = z[1][t].z + n1.y * fs;
                zp[0][t].w = z[0][t].w + n1.z * fs; zp[1][t].w = z[1][t].w + n1.w * fs;
            }
            // padded dims must stay exactly 0 (their weight fragments are 0, but 0*inf would poison)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int d0 = 32 * t + 8 * g;
                if (d0 + 0 >= D) zp[0][t].x = 0.f; if (d0 + 1 >= D) zp[1][t].x = 0.f;
                if (d0 + 2 >= D) zp[0][t].y = 0.f; if (d0 + 3 >= D) zp[1][t].y = 0.f;
                if (d0 + 4 >= D) zp[0][t].z = 0.f; if (d0 + 5 >= D) zp[1][t].z = 0.f;
                if (d0 + 6 >= D) zp[0][t].w = 0.f; if (d0 + 7 >= D) zp[1][t].w = 0.f;
            }
            u = noise_uniform(a.seed, walker, (uint32_t)it);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < NT; ++t) xp[c][t] = zp[c][t];
        float ldp = group_sum(flow_inverse_tile<NT, NH>(img, a.s.net_floats, a.s.B, a.s.L, lane, xp));  // :321

        // log_ratio = log_det_J' - log_det_J, -inf outside the prior box  (sampler.py:326-331)
        const int inb = inbox_tile<NT>(xp);
        float log_ratio = inb ? (ldp - ld) : -INFINITY;
        float ratio = fminf(__expf(log_ratio), 1.0f);  // exp().clamp(max=1)  :335
        if (log_ratio != log_ratio) ratio = log_ratio;  // NaN stays NaN (u < NaN is false, as in torch)
        const bool pre = ok && (u < ratio);             // :336

        // likelihood of the proposal (the reference evaluates it only for `pre` rows, :358-360; here it is
        // evaluated for every row -- the lanes run in lock step anyway -- and only counted for `pre` rows)
        double lp = loglike_tile<NT>(a.like_id, a.like_scale, D, lane, xp);
        const bool acc = pre && (lp > a.loglstar);  // finite is guaranteed by the -1e100 clamp  :361
        n_call += pre ? 1 : 0;
        n_acc += acc ? 1 : 0;
        if (acc) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < NT; ++t) { z[c][t] = zp[c][t]; x[c][t] = xp[c][t]; }
            ld = ldp;
            logl = lp;
        }
        if (a.flags & NNEST_MH_DYNAMIC_STEP) {  // sampler.py:422-431, per adaptation group
            unsigned long long bal = __ballot(acc && g == 0);
            int num_accepted = __popcll(bal);
            if (2 * num_accepted > nvalid) accept += 1; else reject += 1;
            if (accept > reject) scale *= exp(1.0 / (1 + accept));
            if (accept < reject) scale /= exp(1.0 / (1 + reject));
        }
        if (a.hist_x) store_tile<NT>(a.hist_x, (long)row * (S + 1) + it, ok, D, lane, x);
        if (a.hist_logl && ok && g == 0) a.hist_logl[(size_t)row * (S + 1) + it] = logl;
    }
    store_tile<NT>(a.z, row, ok, D, lane, z);
    if (a.x) store_tile<NT>(a.x, row, ok, D, lane, x);
    if (ok && g == 0) {
        a.logl[row] = logl;
        if (a.n_accept) a.n_accept[row] = n_acc;
        if (a.n_call) a.n_call[row] = n_call;
    }
    if (a.scale_out && lane == 0) a.scale_out[tile] = (float)scale;
}

__global__ void fill_noise_kernel(float *__restrict__ dz, float *__restrict__ u, int steps, int C, int D, uint64_t seed,
                                  uint64_t walker_offset) {
    const int nq = (D + 3) / 4;
    const long total = (long)steps * C * nq;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int q = (int)(i % nq);
        long sc = i / nq;
        int c = (int)(sc % C), s = (int)(sc / C);
        f32x4 n = noise_normal4(seed, walker_offset + (uint64_t)c, (uint32_t)(s + 1), (uint32_t)q, NOISE_STREAM_DZ);
        float *o = dz + ((size_t)s * C + c) * D + 4 * q;
        if (4 * q + 0 < D) o[0] = n.x;
        if (4 * q + 1 < D) o[1] = n.y;
        if (4 * q + 2 < D) o[2] = n.z;
        if (4 * q + 3 < D) o[3] = n.w;
        if (q == 0 && u) u[(size_t)s * C + c] = noise_uniform(seed, walker_offset + (uint64_t)c, (uint32_t)(s + 1));
    }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static const int LDS_IMAGE_LIMIT = 150 * 1024;  // leave headroom under the 160 KiB/CU LDS

template <typename K>
static hipError_t allow_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

// pick workgroup width: few tiles -> one wave per workgroup so the tiles spread over CUs
static void pick_geometry(int ntiles, int num_cu, int *block, int *grid) {
    int wpb = 1;
    if (ntiles > 2 * num_cu) wpb = 2;
    if (ntiles > 8 * num_cu) wpb = 4;
    *block = 64 * wpb;
    int g = (ntiles + wpb - 1) / wpb;
    *grid = g;
}

template <int NT, int NH>
static hipError_t launch_pass_t(const PassArgs &a, int num_cu, hipStream_t st) {
    const int ntiles = (a.N + 15) / 16;
    int block, grid;
    pick_geometry(ntiles, num_cu, &block, &grid);
    const size_t img_bytes = (size_t)a.s.image_floats * 4;
    // grid-stride over tiles once there are more than ~8 workgroups per CU (amortises the LDS staging)
    if (grid > 8 * num_cu) grid = 8 * num_cu;
    if (img_bytes <= (size_t)LDS_IMAGE_LIMIT) {
        hipError_t e = allow_lds(flow_pass_kernel<NT, NH, true>, img_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((flow_pass_kernel<NT, NH, true>), dim3(grid), dim3(block), img_bytes, st, a);
    } else {
        hipLaunchKernelGGL((flow_pass_kernel<NT, NH, false>), dim3(grid), dim3(block), 0, st, a);
    }
    return hipGetLastError();
}

template <int NT, int NH>
static hipError_t launch_mh_t(const MhArgs &a, int num_cu, hipStream_t st) {
    const int ntiles = (a.C + 15) / 16;
    int block, grid;
    pick_geometry(ntiles, num_cu, &block, &grid);
    const size_t img_bytes = (size_t)a.s.image_floats * 4;
    if (img_bytes <= (size_t)LDS_IMAGE_LIMIT) {
        hipError_t e = allow_lds(mh_kernel<NT, NH, true>, img_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((mh_kernel<NT, NH, true>), dim3(grid), dim3(block), img_bytes, st, a);
    } else {
        hipLaunchKernelGGL((mh_kernel<NT, NH, false>), dim3(grid), dim3(block), 0, st, a);
    }
    return hipGetLastError();
}

#define DISPATCH_SHAPE(FN, s, ...)                                   \
    do {                                                             \
        if ((s).NH == 1) {                                           \
            switch ((s).NT) {                                        \
                case 1: return FN<1, 1>(__VA_ARGS__);                \
                case 2: return FN<2, 1>(__VA_ARGS__);                \
                case 3: return FN<3, 1>(__VA_ARGS__);                \
                case 4: return FN<4, 1>(__VA_ARGS__);                \
            }                                                        \
        } else if ((s).NH == 2) {                                    \
            switch ((s).NT) {                                        \
                case 1: return FN<1, 2>(__VA_ARGS__);                \
                case 2: return FN<2, 2>(__VA_ARGS__);                \
            }                                                        \
        } else if ((s).NH == 4) {                                    \
            switch ((s).NT) {                                        \
                case 1: return FN<1, 4>(__VA_ARGS__);                \
            }                                                        \
        }                                                            \
        return hipErrorInvalidConfiguration;                         \
    } while (0)

bool shape_supported(const FlowShape &s) {
    if (s.NH == 1) return s.NT >= 1 && s.NT <= 4;
    if (s.NH == 2) return s.NT >= 1 && s.NT <= 2;
    if (s.NH == 4) return s.NT == 1;
    return false;
}

hipError_t launch_repack(const float *packed, float *img, const FlowShape &s, hipStream_t st) {
    int block = 256, grid = (s.image_floats + block - 1) / block;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(repack_fragments_kernel, dim3(grid), dim3(block), 0, st, packed, img, s);
    return hipGetLastError();
}

hipError_t launch_pass(const float *img, const FlowShape &s, int mode, const float *in, float *out, float *logdet,
                       double *logl, int *inbox, int N, int like_id, float like_scale, int num_cu, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    PassArgs a;
    a.img = img; a.s = s; a.mode = mode; a.in = in; a.out = out; a.logdet = logdet; a.logl = logl; a.inbox = inbox;
    a.N = N; a.like_id = like_id; a.like_scale = like_scale;
    DISPATCH_SHAPE(launch_pass_t, s, a, num_cu, st);
}

hipError_t launch_mh(const float *img, const FlowShape &s, int like_id, float like_scale, float *z, float *x, double *logl,
                     double loglstar, float step_size, int steps, int C, int flags, const float *noise_dz,
                     const float *noise_u, uint64_t seed, uint64_t walker_offset, float *hist_x, double *hist_logl,
                     int *n_accept, int *n_call, float *scale_out, int num_cu, hipStream_t st) {
    if (C <= 0) return hipSuccess;
    MhArgs a;
    a.img = img; a.s = s; a.z = z; a.x = x; a.logl = logl; a.loglstar = loglstar; a.step_size = step_size;
    a.steps = steps; a.C = C; a.flags = flags; a.like_id = like_id; a.like_scale = like_scale;
    a.noise_dz = noise_dz; a.noise_u = noise_u; a.seed = seed; a.walker_offset = walker_offset;
    a.hist_x = hist_x; a.hist_logl = hist_logl; a.n_accept = n_accept; a.n_call = n_call; a.scale_out = scale_out;
    DISPATCH_SHAPE(launch_mh_t, s, a, num_cu, st);
}

hipError_t launch_loglike(int like_id, const float *x, float scale, double *logl, int N, int D, int num_cu, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    const int NT = ((D + 1) / 2 + 15) / 16;
    const int ntiles = (N + 15) / 16;
    int block = 256, grid = (ntiles + 3) / 4;
    if (grid > 8 * num_cu) grid = 8 * num_cu;
    switch (NT) {
        case 1: hipLaunchKernelGGL((loglike_kernel<1>), dim3(grid), dim3(block), 0, st, x, logl, N, D, like_id, scale); break;
        case 2: hipLaunchKernelGGL((loglike_kernel<2>), dim3(grid), dim3(block), 0, st, x, logl, N, D, like_id, scale); break;
        case 3: hipLaunchKernelGGL((loglike_kernel<3>), dim3(grid), dim3(block), 0, st, x, logl, N, D, like_id, scale); break;
        case 4: hipLaunchKernelGGL((loglike_kernel<4>), dim3(grid), dim3(block), 0, st, x, logl, N, D, like_id, scale); break;
        default: return hipErrorInvalidConfiguration;
    }
    return hipGetLastError();
}

hipError_t launch_fill_noise(float *dz, float *u, int steps, int C, int D, uint64_t seed, uint64_t walker_offset,
                             hipStream_t st) {
    long total = (long)steps * C * ((D + 3) / 4);
    if (total <= 0) return hipSuccess;
    int block = 256;
    long grid = (total + block - 1) / block;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(fill_noise_kernel, dim3((int)grid), dim3(block), 0, st, dz, u, steps, C, D, seed, walker_offset);
    return hipGetLastError();
}

int mh_num_groups(int C) { return (C + 15) / 16; }

}  // namespace nnest
