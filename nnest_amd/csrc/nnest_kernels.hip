// nnest_kernels.hip -- inference-side kernels of the nnest hot path for MI355X (gfx950):
//   repack_fragments_kernel   packed state_dict weights -> MFMA A-fragment image
//   flow_pass_kernel          K1 forward / K2 inverse / log_probs / K3 fused inverse + box prior + loglike
//   loglike_kernel            K6 batched analytic likelihoods
//   mh_kernel                 K4 persistent multi-step constrained Metropolis (Sampler._mcmc_sample)
//   fill_noise_kernel         the in-kernel proposal noise as arrays (tests)
// Launch wrappers (C++, used by nnest_abi.hip) are at the bottom.  See flow_tile.h for the data layout.
#include <map>
#include <mutex>
#include <utility>
#include <string.h>

#include "flow_tile.h"
#include "maf_tile.h"
#include "mh_common.h"
#include "nnest_internal.h"
#include "spline_train_tile.h"   // (spline_kernels.h, included inside the namespace at the end, uses its fragment-prefetch helpers)

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
        if (o < frag_off_L2(NT, NH)) {  // L1 [ht][tau][lane][r]
            int r = o & 3, lane = (o >> 2) & 63, q = (o >> 8) << 2, tau = (q >> 2) % NT, ht = (q >> 2) / NT;
            int g = lane >> 4, i = lane & 15;
            int j = 16 * ht + i, d = 2 * (16 * tau + 4 * g + r) + pc;
            if (d < D) v = p[pW0 + j * D + d];
        } else if (o < frag_off_L3(NT, NH, L)) {  // L2 [l][hto][hti][lane][r]
            int oo = o - frag_off_L2(NT, NH);
            int r = oo & 3, lane = (oo >> 2) & 63, q = (oo >> 8) << 2, hti = (q >> 2) % NH, hto = ((q >> 2) / NH) % NH,
                l = (q >> 2) / (NH * NH);
            int g = lane >> 4, i = lane & 15;
            v = p[phid + l * (H * H + H) + (16 * hto + i) * H + 16 * hti + 4 * g + r];
        } else if (o < frag_off_b1(NT, NH, L)) {  // L3 [tau][ht][lane][r]
            int oo = o - frag_off_L3(NT, NH, L);
            int r = oo & 3, lane = (oo >> 2) & 63, q = (oo >> 8) << 2, ht = (q >> 2) % NH, tau = (q >> 2) / NH;
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
    // ScaleLayer scalars (scale='constant') ride behind the fragments
    if (s.scale_mode == 2 && blockIdx.x == 0 && threadIdx.x < s.B) img[total + threadIdx.x] = packed[s.nets_params() + threadIdx.x];
}

// scale='translate' / 'constant': the scale_net slots of the packed vector are unused and must read as log_s = 0
__global__ void zero_scale_nets_kernel(float *__restrict__ packed, FlowShape s) {
    const int n = s.nets_params();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (((i / s.net_params) & 1) == 0) packed[i] = 0.f;
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
    int waves_active;  // waves per workgroup that own tiles; the remaining waves only help stage the image
    LikeSpec like;
};

template <int NT, int NH, int LT, bool WLDS>
__global__ void __launch_bounds__(256) flow_pass_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_img[];
    const float *img = a.img;
    if (WLDS) {
        stage_image(lds_img, a.img, a.s.image_floats);
        img = lds_img;
    }
    const float *blk_scale = a.s.scale_mode == 2 ? a.img + a.s.image_floats : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = a.waves_active;
    const int ntiles = (a.N + 15) >> 4;
    const int w = lane & 15, g = lane >> 4;
    if (wave >= wpb) return;  // staging helper only (a lone wave needs ~10 us to pull a 32 KB image into LDS)
    for (int tile = blockIdx.x * wpb + wave; tile < ntiles; tile += gridDim.x * wpb) {
        const int row = tile * 16 + w;
        const bool ok = row < a.N;
        f32x4 xs[2][NT];
        load_tile<NT>(a.in, row, ok, a.s.D, lane, xs);
        float ld;
        if (a.mode == PASS_FORWARD || a.mode == PASS_LOGPROB)
            ld = flow_forward_tile<NT, NH, LT>(img, a.s.net_floats, a.s.B, a.s.L, lane, xs, blk_scale);
        else
            ld = flow_inverse_tile<NT, NH, LT>(img, a.s.net_floats, a.s.B, a.s.L, lane, xs, blk_scale);
        ld = group_sum(ld);
        if (a.mode == PASS_LOGPROB) {
            // MVN(0,I).log_prob(u) + logdet  (networks.py:51-57, :71-76)
            float ss = 0.f;  // padded slots hold 0 and E(0) = 0
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int tau = 0; tau < NT; ++tau) ss += base_E4(xs[c][tau], a.s.base_beta);
            ss = group_sum(ss);
            if (ok && g == 0) a.out[row] = -ss + a.s.base_const * (float)a.s.D + ld;
            continue;
        }
        if (a.out) store_tile<NT>(a.out, row, ok, a.s.D, lane, xs);
        if (a.logdet && ok && g == 0) a.logdet[row] = ld;
        if (a.mode == PASS_INVERSE_LOGLIKE) {
            int inb = inbox_tile<NT>(xs, lane);
            double ll = loglike_tile<NT>(a.like, a.s.D, lane, xs);
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
                                                      LikeSpec like) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int ntiles = (N + 15) >> 4;
    for (int tile = blockIdx.x * wpb + wave; tile < ntiles; tile += gridDim.x * wpb) {
        const int row = tile * 16 + (lane & 15);
        const bool ok = row < N;
        f32x4 xs[2][NT];
        load_tile<NT>(x, row, ok, D, lane, xs);
        double ll = loglike_tile<NT>(like, D, lane, xs);
        if (ok && (lane >> 4) == 0) logl[row] = ll;
    }
}

#include "mh_body.h"

// Form 1 (any shape): weight fragments read from the image (LDS copy, or global when it does not fit) at each use.
template <int NT, int NH, int LT>
struct ImageInverse {
    const float *img;
    int net_floats, B, L, lane;
    const float *blk_scale;
#ifdef NNEST_STAMP
    unsigned long long t_mlp = 0, t_xch = 0, t_upd = 0;
#endif
    __device__ __forceinline__ float operator()(f32x4 (&xs)[2][NT]) const {
        return flow_inverse_tile<NT, NH, LT>(img, net_floats, B, L, lane, xs, blk_scale);
    }
};

// OCC = waves per SIMD the build is compiled for (its register budget): 3 -- up to eight waves per workgroup, <= 168 registers,
// the chains of one wave hide behind the others (chip-filling populations at <= 2 tiles per class; 61-118 spilled registers
// there, and still 5 % faster at 131 072 walkers than the spill-free OCC 2 build); 2 -- <= 256 registers, no spills at <= 2 tiles
// per class; 1 -- at most two waves per workgroup and the whole register file: 3-4 tiles per class (x_dim > 64), where the
// proposal state alone is ~160 registers (BASELINE config 5 on ONE GPU: 500 tiles on 256 CUs; 0 spills against 396).
// Measured (profiles/r03/k4_image_occupancy_d{50,100}.txt): x_dim 100 -- OCC 1 9.1 us per step at 8000 walkers (OCC 3: 13.2),
// OCC 2 best from 32 768 walkers on (1.98e9 evals/s at 131 072 against 1.70e9 / 1.31e9); x_dim 50 -- OCC 2 and 3 within 5 %.
template <int NT, int NH, int LT, bool WLDS, bool DBG, int OCC = 3>
__global__ void __launch_bounds__(OCC == 1 ? 128 : 512, OCC) mh_kernel(MhArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_img[];
    const float *img = a.img;
    if (WLDS) {
        stage_image(lds_img, a.img, a.s.image_floats);
        img = lds_img;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int tile = blockIdx.x * wpb + wave;
    if (tile >= ((a.C + 15) >> 4)) return;
    ImageInverse<NT, NH, LT> inv = {img, a.s.net_floats, a.s.B, a.s.L, lane, a.s.scale_mode == 2 ? a.img + a.s.image_floats : nullptr};
    XoshiroNoise<NT> noise;
    noise.init(a.seed, a.walker_offset + (uint64_t)(tile * 16 + (lane & 15)), lane >> 4, a.s.D);
    mh_body<NT, DBG>(a, tile, lane, inv, noise, true);
}

// Form 2 (the reference's default flow: num_blocks = B, num_layers = L fixed at compile time): every weight
// fragment of the stack is loaded ONCE into this lane's registers (one float per MFMA, 40 per block at
// x_dim 50) and stays there for all steps; only the biases sit in LDS.  The image-backed form measured ~75
// cycles per MFMA because each fragment read exposes LDS latency in front of its MFMA (tools/ablate_mh.hip);
// here the MFMA A operands are plain registers.  One wave per workgroup.
template <int NT, int NH, int L, int B>
struct RegInverse {
    typedef FragCount<NT, NH, L> FC;
    RegFrags<FC::N> w[B][2];
    const float *bias;  // LDS: [b][net][16*NH*(1+L) + 16*NT]
    int lane;
#ifdef NNEST_STAMP
    unsigned long long t_mlp = 0, t_xch = 0, t_upd = 0;
#endif
    static constexpr int NBIAS = 16 * NH * (1 + L) + 16 * NT;
    __device__ __forceinline__ float operator()(f32x4 (&xs)[2][NT]) const {
        float ld = 0.f;
#pragma unroll
        for (int b = B - 1; b >= 0; --b) {
            const float *bs = bias + (b * 2) * NBIAS, *bt = bs + NBIAS;
            if (b & 1) ld += coupling_core<NT, NH, L, true>(w[b][0], w[b][1], bs, bt, lane, xs[0], xs[1]);
            else       ld += coupling_core<NT, NH, L, true>(w[b][0], w[b][1], bs, bt, lane, xs[1], xs[0]);
        }
        return ld;
    }
};

template <int NT, int NH, int L, int B, bool DBG>
__global__ void __launch_bounds__(64) mh_kernel_reg(MhArgs a) {
    typedef RegInverse<NT, NH, L, B> RI;
    __shared__ __attribute__((aligned(16))) float bias_lds[B * 2 * RI::NBIAS];
    const int lane = threadIdx.x;
    const int tile = blockIdx.x;
    const int net_floats = a.s.net_floats;
    RI inv;
    inv.bias = bias_lds;
    inv.lane = lane;
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float *src = a.img + (size_t)(b * 2 + n) * net_floats;
#pragma unroll
            for (int i = 0; i < RI::FC::N; ++i) inv.w[b][n].v[i] = src[frag_elem(i, lane)];
            for (int i = lane; i < RI::NBIAS; i += 64) bias_lds[(b * 2 + n) * RI::NBIAS + i] = src[frag_off_b1(NT, NH, L) + i];
        }
    __syncthreads();
    XoshiroNoise<NT> noise;
    noise.init(a.seed, a.walker_offset + (uint64_t)(tile * 16 + (lane & 15)), lane >> 4, a.s.D);
    mh_body<NT, DBG>(a, tile, lane, inv, noise, true);
}

// Form 3 (small populations, default flow): a TEAM of three waves on one CU per 16-walker tile.
//   wave 0  scale net       wave 1  translate net       wave 2  proposal noise for the NEXT step
// A lone wave spends a step in a strictly serial chain (noise -> 3 x [Linear, act, Linear, act, Linear, affine] ->
// likelihood -> accept, ~10k cycles with both nets' MFMAs in one issue stream); with fewer tiles than CUs the
// other three SIMDs of the CU idle.  Here the two nets of a block run concurrently on two SIMDs and meet once per
// block through LDS (each wave publishes its ls / t tiles, one barrier, each reads the other's and applies the
// same affine update, so both hold bit-identical state), and the noise leaves the critical path.  Waves 0 and 1
// evaluate prior / likelihood / accept redundantly on identical values; wave 0 writes the results.
// WREG: this wave's fragments live in its registers (x_dim <= 64); otherwise (3-4 tiles per class: the proposal state alone
// takes ~160 VGPRs) they are read from a copy of the image in LDS -- the split over three waves stays.
template <int NT, int L, int B, bool WREG = (NT <= 2)>
struct TeamInverse {
    typedef FragCount<NT, 1, L> FC;
    static constexpr int NBIAS = 16 * (1 + L) + 16 * NT;
    RegFrags<WREG ? FC::N : 1> w[B];  // this wave's net only
    const float *wl;       // !WREG: LDS image + 4 * lane
    int net_floats;
    const float *bias;     // LDS [b][net][NBIAS]
    f32x4 *xch;            // LDS [parity][net][NT][64 lanes]
    int lane, role;
#ifdef NNEST_STAMP
    mutable unsigned long long t_mlp, t_xch, t_upd;
#endif
    __device__ __forceinline__ float operator()(f32x4 (&xs)[2][NT]) const {
        float ld = 0.f;
        NetBias<NT, 1, L> nb;
        nb.load(bias + ((B - 1) * 2 + role) * NBIAS, lane);
#pragma unroll
        for (int b = B - 1; b >= 0; --b) {
            f32x4 mine[NT], other[NT];
            unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            (void)s0; (void)s1; (void)s2; (void)s3;
            STAMP(s0);
            if constexpr (WREG) {
                if (b & 1) {
                    if (role == 0) mlp_core<NT, 1, L, 0>(w[b], nb, xs[0], mine);
                    else           mlp_core<NT, 1, L, 1>(w[b], nb, xs[0], mine);
                } else {
                    if (role == 0) mlp_core<NT, 1, L, 0>(w[b], nb, xs[1], mine);
                    else           mlp_core<NT, 1, L, 1>(w[b], nb, xs[1], mine);
                }
            } else {
                const ImageFrags fr = {wl + (size_t)(b * 2 + role) * net_floats};
                if (b & 1) {
                    if (role == 0) mlp_core<NT, 1, L, 0>(fr, nb, xs[0], mine);
                    else           mlp_core<NT, 1, L, 1>(fr, nb, xs[0], mine);
                } else {
                    if (role == 0) mlp_core<NT, 1, L, 0>(fr, nb, xs[1], mine);
                    else           mlp_core<NT, 1, L, 1>(fr, nb, xs[1], mine);
                }
            }
            f32x4 *slot = xch + (size_t)((b & 1) * 2) * NT * 64;  // double-buffered by block parity
            STAMP(s1);
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) slot[(role * NT + tau) * 64 + lane] = mine[tau];
            if (b > 0) nb.load(bias + ((b - 1) * 2 + role) * NBIAS, lane);  // next block's biases ride under the barrier
            __syncthreads();
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) other[tau] = slot[((1 - role) * NT + tau) * 64 + lane];
            STAMP(s2);
            if (b & 1) ld += role == 0 ? affine_update<NT, true>(mine, other, xs[1]) : affine_update<NT, true>(other, mine, xs[1]);
            else       ld += role == 0 ? affine_update<NT, true>(mine, other, xs[0]) : affine_update<NT, true>(other, mine, xs[0]);
            STAMP(s3);
#ifdef NNEST_STAMP
            t_mlp += s1 - s0; t_xch += s2 - s1; t_upd += s3 - s2;
#endif
        }
        return ld;
    }
};

template <int NT>
struct LdsNoise {
    const float *nbuf;  // [2][NT*8][64]
    const float *ubuf;  // [2][64]
    int lane, k;
    // batch-wide step rule at lag >= 2 (mh_common.h): the noise wave does the global-memory work -- it posts the tile's accepted
    // count it finds in acc[] and fetches the batch total the net waves will apply -- and hands the total over here, at the
    // per-step barrier; a step of the net waves then contains no global-memory operation
    int *acc;          // [2]: accepted count of step s at s & 1 (written by the writer wave)
    const int *res;    // [2]: batch total to apply in step k at k & 1 (written by the noise wave)
    bool relay;
    int total;
    __device__ __forceinline__ bool relays() const { return relay; }
    __device__ __forceinline__ int relayed_total() const { return total; }
    __device__ __forceinline__ void relay_count(int it, int accepted) const { acc[it & 1] = accepted; }
    __device__ __forceinline__ void next(float (&nz)[NT][8], float &u) {
        __syncthreads();  // the noise wave has published buffer k&1
        const float *p = nbuf + (size_t)(k & 1) * NT * 8 * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) nz[t][j] = p[(t * 8 + j) * 64];
        u = ubuf[(k & 1) * 64 + lane];
        if (relay) total = res[k & 1];
        ++k;
    }
};

template <int NT, int L, int B, bool DBG>
// min 2 waves/SIMD, i.e. <= 256 VGPRs: MFMA results stay in VGPRs.  The build with the whole register file (-DNNEST_TEAM_FULL_FROM=<tiles
// per class>; 0 spilled registers against 41 at 2 tiles per class and 105 at 4) measured SLOWER, round 3: 4.34 against 4.12 us per
// step at x_dim 50 / 4000 walkers, 6.76 against 6.50-6.68 at x_dim 100 / 2000-4000 (profiles/r03/k4_team_register_budget.txt).
#ifndef NNEST_TEAM_FULL_FROM
#define NNEST_TEAM_FULL_FROM 99
#endif
__global__ void __launch_bounds__(192, NT >= NNEST_TEAM_FULL_FROM ? 1 : 2) mh_kernel_team(MhArgs a) {
    typedef TeamInverse<NT, L, B> TI;
    constexpr bool WREG = NT <= 2;
    extern __shared__ __attribute__((aligned(16))) float team_img[];  // !WREG: the fragment image
    __shared__ __attribute__((aligned(16))) float bias_lds[B * 2 * TI::NBIAS];
    __shared__ __attribute__((aligned(16))) f32x4 xch[2 * 2 * NT * 64];
    __shared__ float nbuf[2 * NT * 8 * 64];
    __shared__ float ubuf[2 * 64];
    __shared__ int acc_lds[2], res_lds[2];
    const int lane = threadIdx.x & 63, role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform
    const int tile = blockIdx.x;
    const int net_floats = a.s.net_floats, S = a.steps;
    const int ntiles = (a.C + 15) >> 4;
    const bool batch_rule = (a.flags & NNEST_MH_DYNAMIC_BATCH) != 0;
    const int lag_rule = mh_flag_lag(a.flags);
    const bool relay = batch_rule && lag_rule >= 2;
    if (tile >= ntiles) {   // the workgroup behind the tiles publishes the batch-wide accept counts (mh_common.h)
        if (relay && role == 0) mh_sync_publisher(a.sync, S, S - lag_rule, ntiles, a.C, lane, a.sync_err);
        return;
    }
    for (int i = threadIdx.x; i < B * 2 * TI::NBIAS; i += blockDim.x) {
        int bn = i / TI::NBIAS, o = i - bn * TI::NBIAS;
        bias_lds[i] = a.img[(size_t)bn * net_floats + frag_off_b1(NT, 1, L) + o];
    }
    if (!WREG) stage_image(team_img, a.img, a.s.image_floats);
    __syncthreads();
    if (role == 2) {  // noise producer: S+1 buffers, each published by the barrier the consumers wait on
        XoshiroNoise<NT> gen;
        gen.init(a.seed, a.walker_offset + (uint64_t)(tile * 16 + (lane & 15)), lane >> 4, a.s.D);
        for (int k = 0; k <= S; ++k) {
            // relay: between the barriers k - 1 and k the net waves run step k - 1; the count of step k - 2 is in LDS, and the
            // total they apply at the end of step k (that of step k - lag) has to be in LDS by barrier k
            unsigned long long early = 0;
            const int want = k - lag_rule;
            if (relay) {
                if (k >= 2 && lane == 0) mh_sync_post(a.sync, k - 2, tile, acc_lds[k & 1]);
                if (want >= 1) early = mh_result_load(a.sync, S, want, tile);
            }
            float nz[NT][8], u;
            gen.next(nz, u);
            float *p = nbuf + (size_t)(k & 1) * NT * 8 * 64 + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) p[(t * 8 + j) * 64] = nz[t][j];
            ubuf[(k & 1) * 64 + lane] = u;
            if (relay && want >= 1) {
                const int total = mh_result_wait(a.sync, S, want, tile, early, a.sync_err);
                if (lane == 0) res_lds[k & 1] = total;
            }
            if (k == 0)
                for (int b = 0; b < B; ++b) __syncthreads();  // the consumers' initial inverse
            __syncthreads();                                   // publish buffer k
            if (k >= 1)
                for (int b = 0; b < B; ++b) __syncthreads();  // step k's inverse
        }
        return;
    }
    TI inv;
    inv.bias = bias_lds;
    inv.xch = xch;
    inv.lane = lane;
    inv.role = role;
#ifdef NNEST_STAMP
    inv.t_mlp = inv.t_xch = inv.t_upd = 0;
#endif
    inv.wl = team_img + 4 * lane;
    inv.net_floats = net_floats;
    if constexpr (WREG) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const float *src = a.img + (size_t)(b * 2 + role) * net_floats;
#pragma unroll
            for (int i = 0; i < TI::FC::N; ++i) inv.w[b].v[i] = src[frag_elem(i, lane)];
        }
    }
    LdsNoise<NT> noise = {nbuf, ubuf, lane, 0, acc_lds, res_lds, relay, 0};
    mh_body<NT, DBG>(a, tile, lane, inv, noise, role == 0);
}

// The proposal noise of mh_kernel as arrays: one thread per (walker, lane group) replays the same streams.
__global__ void fill_noise_kernel(float *__restrict__ dz, float *__restrict__ u, int steps, int C, int D, uint64_t seed,
                                  uint64_t walker_offset) {
    const int NT = ((D + 1) / 2 + 15) / 16;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * 4) return;
    const int c = i >> 2, g = i & 3;
    const uint64_t walker = walker_offset + (uint64_t)c;
    Xoshiro128 rng_n = xoshiro_seed(seed, walker, (uint32_t)g, NOISE_STREAM_DZ);
    Xoshiro128 rng_u = xoshiro_seed(seed, walker, 0xffffffffu, NOISE_STREAM_U);
    for (int s = 0; s < steps; ++s) {
        for (int t = 0; t < NT; ++t) {
            float n[8];
            xoshiro_normal8(rng_n, n);
            for (int j = 0; j < 8; ++j) {
                int d = 32 * t + 8 * g + j;
                if (d < D) dz[((size_t)s * C + c) * D + d] = n[j];
            }
        }
        float uu = xoshiro_uniform(rng_u);
        if (g == 0 && u) u[(size_t)s * C + c] = uu;
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
static void pick_geometry(int ntiles, int num_cu, int max_wpb, int *block, int *grid) {
    int wpb = 1;
    if (ntiles > 2 * num_cu) wpb = 2;
    if (ntiles > 4 * num_cu) wpb = 4;
    if (ntiles > 8 * num_cu) wpb = 8;  // eight waves share one LDS copy of the weight image
    if (wpb > max_wpb) wpb = max_wpb;
    *block = 64 * wpb;
    int g = (ntiles + wpb - 1) / wpb;
    *grid = g;
}

template <int NT, int NH, int LT>
static hipError_t launch_pass_t(const PassArgs &a_in, int num_cu, hipStream_t st) {
    const int ntiles = (a_in.N + 15) / 16;
    int block, grid;
    pick_geometry(ntiles, num_cu, 4, &block, &grid);  // flow_pass_kernel: __launch_bounds__(256)
    PassArgs a = a_in;
    a.waves_active = block / 64;
    block = 256;  // always four waves to stage the image; waves_active of them own tiles
    const size_t img_bytes = (size_t)a.s.image_floats * 4;
    // grid-stride over tiles once there are more than ~8 workgroups per CU (amortises the LDS staging)
    if (grid > 8 * num_cu) grid = 8 * num_cu;
    if (img_bytes <= (size_t)LDS_IMAGE_LIMIT) {
        hipError_t e = allow_lds(flow_pass_kernel<NT, NH, LT, true>, img_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((flow_pass_kernel<NT, NH, LT, true>), dim3(grid), dim3(block), img_bytes, st, a);
    } else {
        hipLaunchKernelGGL((flow_pass_kernel<NT, NH, LT, false>), dim3(grid), dim3(block), 0, st, a);
    }
    return hipGetLastError();
}

// The image form's launch shape, decided in ONE place (the launcher uses it; pick_mh_form asks it whether the batch-wide rule's
// grid is resident).  occ: the build (waves per SIMD it is compiled for, see mh_kernel); capacity: workgroups of this shape the
// chip holds at once -- by wave slots (4 SIMDs x occ per CU) and by LDS (one copy of the fragment image per workgroup).  Under
// the batch-wide rule every workgroup waits on every other one, so the grid must not exceed the capacity: the workgroups are made
// as large as that takes (the fewest waves per workgroup that fit: small populations still spread over the CUs).
struct ImageGeom { int occ, block, grid, capacity; };
template <int NT>
static ImageGeom image_geometry(int ntiles, int num_cu, size_t img_bytes, bool batch, bool dbg) {
    static const int occ_env = [] { const char *e = getenv("NNEST_MH_OCC"); return e ? atoi(e) : 0; }();   // diagnostic: pin the build
    const bool in_lds = img_bytes <= (size_t)LDS_IMAGE_LIMIT;
    ImageGeom g;
    pick_geometry(ntiles, num_cu, 8, &g.block, &g.grid);
    g.occ = 3;
    if (in_lds && !dbg) {   // (the builds that record history / replay noise exist at 3 waves per SIMD only)
        if (NT >= 3 && ((occ_env == 0 && ntiles <= 4 * num_cu) || occ_env == 1)) g.occ = 1;        // <= one tile per SIMD
        else if (occ_env == 2 || (occ_env == 0 && (NT >= 3 || ntiles <= 16 * num_cu))) g.occ = 2;
    }
    const int max_wpb = g.occ == 1 ? 2 : 8;
    int wpb = g.block / 64 < max_wpb ? g.block / 64 : max_wpb;
    auto capacity = [&](int w) {
        int per_cu = (4 * g.occ) / w;
        if (in_lds) {
            const int by_lds = (int)(((size_t)160 * 1024 - 4096) / (img_bytes > 0 ? img_bytes : 1));
            per_cu = per_cu < by_lds ? per_cu : by_lds;
        }
        return num_cu * (per_cu > 0 ? per_cu : 0);
    };
    if (batch)
        while (wpb < max_wpb && (ntiles + wpb - 1) / wpb > capacity(wpb)) wpb *= 2;
    g.block = 64 * wpb;
    g.grid = (ntiles + wpb - 1) / wpb;
    g.capacity = capacity(wpb);
    return g;
}

// Which form runs (DESIGN.md "K4"): quad (4 walkers per tile, nnest_quad.hip) while its tiles fit the CUs, then team, register,
// image by population; flags bits 16..19 pin a form (a caller that shards one batch over ranks pins the form the whole
// batch would get, so a shard reproduces the slice of the unsharded run bit for bit).  The batch-wide step rule needs every
// workgroup resident: it is refused where the grid could exceed the chip.
// The form a launch runs, decided in ONE place (also behind nnest_mh_form_for): the pinned form if it applies to this shape /
// population / rule, the first eligible one otherwise; -1 = none (pinned form not applicable, or the batch-wide rule on a grid
// that may not be resident).
template <int NT, int NH, int LT>
static int pick_mh_form(const MhArgs &a, int num_cu) {
    const int ntiles = (a.C + 15) / 16;
    const int form = mh_flag_form(a.flags);
    const bool batch = (a.flags & NNEST_MH_DYNAMIC_BATCH) != 0;
    if ((form == MH_FORM_AUTO || form == MH_FORM_SOLO) && solo_form_eligible(a, num_cu)) return MH_FORM_SOLO;
    if (form == MH_FORM_SOLO) return -1;
    if (batch && mh_flag_lag(a.flags) > 0 && mh_flag_warm(a.flags) > 0) return -1;   // exact warm-up steps in front of a lagged rule: the solo form only
    if ((form == MH_FORM_AUTO || form == MH_FORM_QUAD || form == MH_FORM_QUAD1) && quad_form_eligible(a, num_cu))
        return form == MH_FORM_QUAD1 ? MH_FORM_QUAD1 : MH_FORM_QUAD;
    if (form == MH_FORM_QUAD || form == MH_FORM_QUAD1) return -1;
    if constexpr (LT == 1 && NH == 1) {  // fewer tiles than CUs: three waves per tile (team form)
        if ((form == MH_FORM_AUTO || form == MH_FORM_TEAM) && a.s.B == 3 && ntiles <= num_cu && !a.noise_dz && a.s.scale_mode != 2) return MH_FORM_TEAM;
    }
    if (form == MH_FORM_TEAM) return -1;
    if constexpr (LT == 1 && NH == 1 && NT <= 2) {  // register form: NT >= 3 would spill.  One wave per SIMD available; SingleSpeedNVP defaults: hidden_dim 16, num_blocks 3, num_layers 1 (nnest/sampler.py:37-43)
        if ((form == MH_FORM_AUTO || form == MH_FORM_REG) && a.s.B == 3 && ntiles <= 4 * num_cu && a.s.scale_mode != 2) return MH_FORM_REG;
    }
    if (form == MH_FORM_REG) return -1;
    const ImageGeom g = image_geometry<NT>(ntiles, num_cu, (size_t)a.s.image_floats * 4, batch, a.noise_dz || a.hist_x || a.hist_logl);
    if (batch && g.grid > g.capacity) return -1;  // the batch-wide rule needs every workgroup resident
    return MH_FORM_IMAGE;
}

// ---- the usable-chain test of the 16-walker-tile forms (nested.py:432), behind their launch --------------------------------------
// one wave per walker: every coordinate of the last x differs from the first -> NNEST_MH_ALL_MOVED into the accept count's word
__global__ void __launch_bounds__(256) mh_all_moved_kernel(const float *__restrict__ x0, const float *__restrict__ x, int *__restrict__ n_accept, int C, int D) {
    const int lane = threadIdx.x & 63;
    for (int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); c < C; c += gridDim.x * (blockDim.x >> 6)) {
        bool all = true;
        for (int d = lane; d < D; d += 64) all = all && x[(size_t)c * D + d] != x0[(size_t)c * D + d];
        if (__ballot(all) == ~0ull && lane == 0) n_accept[c] |= NNEST_MH_ALL_MOVED;
    }
}
// the side buffer the tile forms park the first x in (for the usable-chain test behind the launch): one per (device, stream), grown on
// demand under a lock and reused by that stream's launches -- the MH kernel and mh_all_moved_kernel that reads the buffer are
// ordered by the stream, and launches on OTHER streams (another flow, another host thread: ctypes releases the interpreter lock)
// have a buffer of their own (round 6, ADVICE r05: it was one unlocked buffer per device).  Growing frees the old buffer, which
// waits for the device.  NULL = allocation failed: the caller fails the launch (the flag's meaning must not change silently).
float *mh_first_x_buffer(size_t floats, hipStream_t st) {
    struct Buf { float *p = nullptr; size_t cap = 0; };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Buf> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    Buf &b = bufs[std::make_pair(dev, st)];
    if (b.cap < floats) {
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr; b.cap = 0;
        const size_t want = floats + floats / 2 + 4096;
        if (hipMalloc((void **)&b.p, want * sizeof(float)) != hipSuccess) { b.p = nullptr; return nullptr; }
        b.cap = want;
    }
    return b.p;
}
// NNEST_MH_SYNC_ZERO_NEXT / _PREV for the forms that do not zero the other half of the caller's double buffer themselves
hipError_t launch_mh_zero_other_sync(const MhArgs &a, hipStream_t st) {
    if (!a.sync || !(a.flags & (NNEST_MH_SYNC_ZERO_NEXT | NNEST_MH_SYNC_ZERO_PREV))) return hipSuccess;
    const size_t W = mh_sync_words(a.steps) + 1;
    unsigned long long *other = (a.flags & NNEST_MH_SYNC_ZERO_NEXT) ? a.sync + W : a.sync - W;
    return hipMemsetAsync(other, 0, W * sizeof(unsigned long long), st);
}
hipError_t launch_mh_all_moved(const MhArgs &a, hipStream_t st) {
    if (!a.x0 || !a.x || !a.n_accept) return hipSuccess;
    const int grid = min((a.C + 3) / 4, 2048);
    hipLaunchKernelGGL(mh_all_moved_kernel, dim3(grid), dim3(256), 0, st, a.x0, a.x, a.n_accept, a.C, a.s.D);
    return hipGetLastError();
}

// Which form runs (DESIGN.md "K4"): solo (1 walker per wave, nnest_solo.hip) / quad (4 walkers per tile, nnest_quad.hip) while their
// tiles fit the CUs, then team, register, image by population; flags bits 16..19 pin a form (a caller that shards one batch over ranks
// pins the form the whole batch would get, so a shard reproduces the slice of the unsharded run bit for bit).  The batch-wide
// step rule needs every workgroup resident: it is refused where the grid could exceed the chip.
template <int NT, int NH, int LT>
static hipError_t launch_mh_tiles_t(const MhArgs &a, int form, int num_cu, hipStream_t st);
template <int NT, int NH, int LT>
static hipError_t launch_mh_t(const MhArgs &a_in, int num_cu, hipStream_t st) {
    const bool batch = (a_in.flags & NNEST_MH_DYNAMIC_BATCH) != 0;
    if (batch && !a_in.sync) return hipErrorInvalidValue;
    const int form = pick_mh_form<NT, NH, LT>(a_in, num_cu);
    if (form < 0) return hipErrorInvalidConfiguration;
    if (form == MH_FORM_SOLO) return launch_mh_solo(a_in, num_cu, st);   // (solo and quad test the chain in-kernel)
    if (form == MH_FORM_QUAD || form == MH_FORM_QUAD1) {
        hipError_t e = launch_mh_quad(a_in, num_cu, st);
        return e != hipSuccess ? e : launch_mh_zero_other_sync(a_in, st);
    }
    MhArgs a = a_in;
    if (a.x && a.n_accept && !(a.x0 = mh_first_x_buffer((size_t)a.C * a.s.D, st))) return hipErrorOutOfMemory;
    hipError_t e = launch_mh_tiles_t<NT, NH, LT>(a, form, num_cu, st);
    if (e != hipSuccess) return e;
    e = launch_mh_zero_other_sync(a, st);
    if (e != hipSuccess) return e;
    return launch_mh_all_moved(a, st);
}
template <int NT, int NH, int LT>
static hipError_t launch_mh_tiles_t(const MhArgs &a, int form, int num_cu, hipStream_t st) {
    const int ntiles = (a.C + 15) / 16;
    const bool batch = (a.flags & NNEST_MH_DYNAMIC_BATCH) != 0;
    if constexpr (LT == 1 && NH == 1) {
        if (form == MH_FORM_TEAM) {
            const size_t timg = NT <= 2 ? 0 : (size_t)a.s.image_floats * 4;  // 3-4 tiles per class: fragments from an LDS image
            const int grid = ntiles + ((batch && mh_flag_lag(a.flags) >= 2) ? 1 : 0);  // + the workgroup that publishes the batch totals
            if (a.hist_x || a.hist_logl) {
                hipError_t e = allow_lds(mh_kernel_team<NT, 1, 3, true>, timg + 48 * 1024);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL((mh_kernel_team<NT, 1, 3, true>), dim3(grid), dim3(192), timg, st, a);
            } else {
                hipError_t e = allow_lds(mh_kernel_team<NT, 1, 3, false>, timg + 48 * 1024);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL((mh_kernel_team<NT, 1, 3, false>), dim3(grid), dim3(192), timg, st, a);
            }
            return hipGetLastError();
        }
    }
    if constexpr (LT == 1 && NH == 1 && NT <= 2) {
        if (form == MH_FORM_REG) {
            if (a.noise_dz || a.hist_x || a.hist_logl)
                hipLaunchKernelGGL((mh_kernel_reg<NT, 1, 1, 3, true>), dim3(ntiles), dim3(64), 0, st, a);
            else
                hipLaunchKernelGGL((mh_kernel_reg<NT, 1, 1, 3, false>), dim3(ntiles), dim3(64), 0, st, a);
            return hipGetLastError();
        }
    }
    const size_t img_bytes = (size_t)a.s.image_floats * 4;
    const bool dbg = a.noise_dz || a.hist_x || a.hist_logl;
    const ImageGeom geo = image_geometry<NT>(ntiles, num_cu, img_bytes, batch, dbg);
    const int block = geo.block, grid = geo.grid;
    if constexpr (NT >= 3) {
        if (geo.occ == 1) {   // one wave per SIMD, the whole register file
            hipError_t e = allow_lds(mh_kernel<NT, NH, LT, true, false, 1>, img_bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((mh_kernel<NT, NH, LT, true, false, 1>), dim3(grid), dim3(block), img_bytes, st, a);
            return hipGetLastError();
        }
    }
    if (geo.occ == 2) {
        hipError_t e = allow_lds(mh_kernel<NT, NH, LT, true, false, 2>, img_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((mh_kernel<NT, NH, LT, true, false, 2>), dim3(grid), dim3(block), img_bytes, st, a);
        return hipGetLastError();
    }
    if (img_bytes <= (size_t)LDS_IMAGE_LIMIT) {
        if (dbg) {
            hipError_t e = allow_lds(mh_kernel<NT, NH, LT, true, true>, img_bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((mh_kernel<NT, NH, LT, true, true>), dim3(grid), dim3(block), img_bytes, st, a);
        } else {
            hipError_t e = allow_lds(mh_kernel<NT, NH, LT, true, false>, img_bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((mh_kernel<NT, NH, LT, true, false>), dim3(grid), dim3(block), img_bytes, st, a);
        }
    } else {
        if (dbg) hipLaunchKernelGGL((mh_kernel<NT, NH, LT, false, true>), dim3(grid), dim3(block), 0, st, a);
        else hipLaunchKernelGGL((mh_kernel<NT, NH, LT, false, false>), dim3(grid), dim3(block), 0, st, a);
    }
    return hipGetLastError();
}

template <int NT, int NH, int LT>
static int mh_form_t(const MhArgs &a, int num_cu) { return pick_mh_form<NT, NH, LT>(a, num_cu); }

// L = 1 (the reference default, nnest/sampler.py:43) gets the compile-time interleaved form; other depths run
// the generic runtime-L form (LT = -1)
#define DISPATCH_NTNH(FN, s, LT, ...)                                \
    do {                                                             \
        if ((s).NH == 1) {                                           \
            switch ((s).NT) {                                        \
                case 1: return FN<1, 1, LT>(__VA_ARGS__);            \
                case 2: return FN<2, 1, LT>(__VA_ARGS__);            \
                case 3: return FN<3, 1, LT>(__VA_ARGS__);            \
                case 4: return FN<4, 1, LT>(__VA_ARGS__);            \
            }                                                        \
        } else if ((s).NH == 2) {                                    \
            switch ((s).NT) {                                        \
                case 1: return FN<1, 2, LT>(__VA_ARGS__);            \
                case 2: return FN<2, 2, LT>(__VA_ARGS__);            \
            }                                                        \
        } else if ((s).NH == 4) {                                    \
            switch ((s).NT) {                                        \
                case 1: return FN<1, 4, LT>(__VA_ARGS__);            \
            }                                                        \
        }                                                            \
        return hipErrorInvalidConfiguration;                         \
    } while (0)

#define DISPATCH_SHAPE(FN, s, ...)                                   \
    do {                                                             \
        if ((s).L == 1) DISPATCH_NTNH(FN, s, 1, __VA_ARGS__);        \
        DISPATCH_NTNH(FN, s, -1, __VA_ARGS__);                       \
    } while (0)

bool shape_supported(const FlowShape &s) {
    if (s.NH == 1) return s.NT >= 1 && s.NT <= 4;
    if (s.NH == 2) return s.NT >= 1 && s.NT <= 2;
    if (s.NH == 4) return s.NT == 1;
    return false;
}

hipError_t launch_zero_scale_nets(float *packed, const FlowShape &s, hipStream_t st) {
    hipLaunchKernelGGL(zero_scale_nets_kernel, dim3(64), dim3(256), 0, st, packed, s);
    return hipGetLastError();
}

hipError_t launch_repack(const float *packed, float *img, const FlowShape &s, hipStream_t st) {
    int block = 256, grid = (s.image_floats + block - 1) / block;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(repack_fragments_kernel, dim3(grid), dim3(block), 0, st, packed, img, s);
    return hipGetLastError();
}

static hipError_t launch_maf_pass(const PassArgs &a, int num_cu, hipStream_t st);   // maf_kernels.h
static hipError_t launch_maf_mh(const MhArgs &a, int num_cu, hipStream_t st);

hipError_t launch_pass(const float *img, const FlowShape &s, int mode, const float *in, float *out, float *logdet,
                       double *logl, int *inbox, int N, const LikeSpec &like, int num_cu, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    PassArgs a;
    a.img = img; a.s = s; a.mode = mode; a.in = in; a.out = out; a.logdet = logdet; a.logl = logl; a.inbox = inbox;
    a.N = N; a.like = like;
    if (s.kind == FLOW_KIND_MAF) return launch_maf_pass(a, num_cu, st);
    DISPATCH_SHAPE(launch_pass_t, s, a, num_cu, st);
}

hipError_t launch_mh(const float *img, const FlowShape &s, const LikeSpec &like, float *z, float *x, double *logl,
                     double loglstar, float step_size, int steps, int C, int flags, const float *noise_dz,
                     const float *noise_u, uint64_t seed, uint64_t walker_offset, float *hist_x, double *hist_logl,
                     int *n_accept, int *n_call, float *scale_out, const float *packed, unsigned long long *sync, int num_cu,
                     hipStream_t st) {
    if (C <= 0) return hipSuccess;
    MhArgs a;
    a.packed = packed;
    a.sync = sync;
    a.sync_err = sync ? reinterpret_cast<int *>(sync + mh_sync_words(steps)) : nullptr;  // the word behind the counters
    a.img = img; a.s = s; a.z = z; a.x = x; a.logl = logl; a.loglstar = loglstar; a.step_size = step_size;
    a.steps = steps; a.C = C; a.flags = flags; a.like = like;
    a.noise_dz = noise_dz; a.noise_u = noise_u; a.seed = seed; a.walker_offset = walker_offset;
    a.hist_x = hist_x; a.hist_logl = hist_logl; a.n_accept = n_accept; a.n_call = n_call; a.scale_out = scale_out;
    a.x0 = nullptr;
    if (s.kind == FLOW_KIND_MAF) {
        if (a.x && a.n_accept && !(a.x0 = mh_first_x_buffer((size_t)C * s.D, st))) return hipErrorOutOfMemory;
        hipError_t e = launch_maf_mh(a, num_cu, st);
        if (e == hipSuccess) e = launch_mh_zero_other_sync(a, st);
        return e != hipSuccess ? e : launch_mh_all_moved(a, st);
    }
    DISPATCH_SHAPE(launch_mh_t, s, a, num_cu, st);
}

hipError_t launch_loglike(const LikeSpec &like, const float *x, double *logl, int N, int D, int num_cu, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    const int NT = ((D + 1) / 2 + 15) / 16;
    const int ntiles = (N + 15) / 16;
    int block = 256, grid = (ntiles + 3) / 4;
    if (grid > 8 * num_cu) grid = 8 * num_cu;
    switch (NT) {
        case 1: hipLaunchKernelGGL((loglike_kernel<1>), dim3(grid), dim3(block), 0, st, x, logl, N, D, like); break;
        case 2: hipLaunchKernelGGL((loglike_kernel<2>), dim3(grid), dim3(block), 0, st, x, logl, N, D, like); break;
        case 3: hipLaunchKernelGGL((loglike_kernel<3>), dim3(grid), dim3(block), 0, st, x, logl, N, D, like); break;
        case 4: hipLaunchKernelGGL((loglike_kernel<4>), dim3(grid), dim3(block), 0, st, x, logl, N, D, like); break;
        default: return hipErrorInvalidConfiguration;
    }
    return hipGetLastError();
}

hipError_t launch_fill_noise(float *dz, float *u, int steps, int C, int D, uint64_t seed, uint64_t walker_offset,
                             hipStream_t st) {
    if (steps <= 0 || C <= 0) return hipSuccess;
    int block = 256;
    int grid = (C * 4 + block - 1) / block;
    hipLaunchKernelGGL(fill_noise_kernel, dim3(grid), dim3(block), 0, st, dz, u, steps, C, D, seed, walker_offset);
    return hipGetLastError();
}

int mh_num_groups(int C) { return (C + 15) / 16; }

// the form nnest_mh_constrained_steps would run for C walkers with these flags (in-kernel noise, no history): MH_FORM_* or -1
int mh_form_for(const FlowShape &s, int C, int flags, int num_cu) {
    MhArgs a = MhArgs();
    a.s = s; a.C = C; a.flags = flags;
    if (s.kind == FLOW_KIND_MAF) {   // the image form only (maf_kernels.h)
        const int form = mh_flag_form(flags);
        if (form != MH_FORM_AUTO && form != MH_FORM_IMAGE) return -1;
        if ((flags & NNEST_MH_DYNAMIC_BATCH) && mh_flag_lag(flags) > 0 && mh_flag_warm(flags) > 0) return -1;
        int block, grid;
        pick_geometry((C + 15) / 16, num_cu, 4, &block, &grid);
        return ((flags & NNEST_MH_DYNAMIC_BATCH) && grid > num_cu) ? -1 : MH_FORM_IMAGE;
    }
    DISPATCH_SHAPE(mh_form_t, s, a, num_cu);
}

#include "maf_kernels.h"
#include "spline_kernels.h"

}  // namespace nnest
