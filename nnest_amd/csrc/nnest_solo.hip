// nnest_solo.hip -- K4, "solo" form (round 3): the persistent constrained-Metropolis kernel (Sampler._mcmc_sample,
// sampler.py:229-463) with ONE walker per wave, for populations of at most four walkers per compute unit -- BASELINE config 2
// (1000 walkers) is 1000 net waves on the chip's 1024 SIMDs.
//
// Why.  The quad form (nnest_quad.hip) spends a step in a chain of nine small layers, each [3 DPP moves -> 4 dependent
// v_mfma_f32_4x4x1 -> reduce-scatter over the four 16-lane rows (3 permlane swaps + adds + hazard nops) -> activation]:
// 232 cycles per layer of which the matrix pipe is 54, plus three LDS exchanges + barriers per step between the scale-net
// and the translate-net wave (730 cycles).  On gfx950 the f32 MFMA has no peak advantage over the f32 VALU (both
// 64 FLOP/clk/SIMD, the vector figure with packed FMAs; MI355X_MICROARCH.md), so at this population -- latency-bound, one wave per SIMD -- the matrix pipe buys
// nothing and its operand layouts cost the cross-lane traffic.  Here a layer is a chain of v_fmac_f32 with a DPP row
// rotation on the activation operand: 16 lanes of a row hold the 16 features of a layer, lane p accumulates
// out[p] = sum_t W[p][(p - t) & 15] * in[(p - t) & 15] with the weights W[p][(p - t) & 15] in its own registers -- no
// operand movement at all, output feature p lands in lane p, which is the next layer's input layout.
//
// Lane layout.  lane = 32 n + 16 h + p:  n = net (0 scale_net / tanh, 1 translate_net / relu), h = K-half, p = position.
// Every lane holds the walker's state at position p: xs[c][u] = dim 2U p + 2u + c (U = FlowShape::NT; parity class c), i.e.
// 2U consecutive floats of the row; the four (n, h) rows hold bit-identical copies.  Rows h = 0 / 1 accumulate the rotations
// t = 0..7 / 8..15 of every dot product (the input of the h = 1 rows is pre-rotated by 8 with a row-masked DPP move, so all
// rows issue the same row_ror:0..7), one v_permlane16_swap + add joins the halves, one v_permlane32_swap hands log_s to the
// translate half and t to the scale half: both nets of a coupling block run concurrently in ONE wave, with no LDS exchange
// and no workgroup barrier inside the flow.  Per block and U = 2: 40 v_fmac (against 12 MFMA + 9 DPP + 9 permlane + ...),
// 4 half-joins, 2 net swaps; 44 weight registers per lane and block.
//
// One workgroup = WPG net waves (walkers WPG tile .. WPG tile + WPG - 1; WPG = 4, or 8 / 12 beyond one walker per SIMD: round 4)
// + one noise wave that draws the next step's proposal noise (the xoshiro streams of every other form: nnest_mh_fill_noise
// replays them) + one relay wave for the batch-wide step rule (mh_common.h), so that a step of the net waves holds no
// global-memory operation; ONE workgroup barrier per step.
// Summation order differs from the other forms (K split in two halves, left to right), so results agree with them (and with a
// CPU restatement) to rounding, not bitwise; decisions on the same noise are the same except at rounding-borderline ratios.
#include <string.h>
#include "mh_common.h"
#include "nnest_internal.h"
#include "solo_tile.h"

namespace nnest {


// ---- likelihoods on a solo wave (the per-term arithmetic of loglike_tile, flow_tile.h; sums over the 16 positions) ----
#pragma clang fp contract(off)
template <int U, int LK>   // LK >= 0: the likelihood id is known at compile time (the other branches are not instantiated)
static __device__ __forceinline__ double solo_loglike(const LikeSpec &lk_in, int D, int lane, const float (&xs)[2][U]) {
    struct { int id; float scale; const float *p; } lk = {LK >= 0 ? LK : lk_in.id, lk_in.scale, lk_in.p};
    const int m = lane & 15;
    const float scale = lk.scale;
    float th[2 * U + 1];
#pragma unroll
    for (int u = 0; u < U; ++u) { th[2 * u] = scale * xs[0][u]; th[2 * u + 1] = scale * xs[1][u]; }
    double acc;
    if (lk.id == 0) {
        // Rosenbrock (likelihoods.py:51): -sum_i 100 (x[i+1] - x[i]^2)^2 + (1 - x[i])^2, i = 0..D-2
        th[2 * U] = solo_ror<15>(th[0]);  // first dim of position m + 1
        float facc = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * U; ++k) {
            const int i = 2 * U * m + k;
            float a = th[k] * th[k];
            float b = th[k + 1] - a;
            float c = b * b;
            float e = 100.0f * c;
            float f = 1.0f - th[k];
            float q = f * f;
            float term = e + q;
            facc = facc + ((i + 1 < D) ? term : 0.f);
        }
        acc = -(double)solo_row_sum(facc);
    } else if (lk.id == 1) {
        // GaussianMix (likelihoods.py:165-189): logsumexp_k[ log w_k - |theta - mu_k|^2/2 - (D/2) log 2pi ]
        float facc = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * U; ++k) {
            const int d = 2 * U * m + k;
            float sq = th[k] * th[k];
            facc = facc + ((d >= 2 && d < D) ? sq : 0.f);
        }
        const double base = (double)solo_row_sum(facc);
        const float t0 = solo_lane0(th[0]), t1 = solo_lane0(th[1]);  // theta[0], theta[1]: position 0
        const float mu0[4] = {0.f, 0.f, 4.f, -4.f}, mu1[4] = {4.f, -4.f, 0.f, 0.f};
        const double lw[4] = {-0.916290731874155, -1.203972804325936, -1.6094379124341003, -2.302585092994046};
        double l[4], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = t0 - mu0[k], b = t1 - mu1[k];
            double s = base + (double)(a * a) + (D > 1 ? (double)(b * b) : 0.0);
            l[k] = -(s * 0.5) - 0.9189385332046727 * (double)D + lw[k];
            mx = l[k] > mx ? l[k] : mx;
        }
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) se += __expf((float)(l[k] - mx));
        acc = mx + (double)__logf(se);
    } else if (lk.id == 2) {
        // Himmelblau (likelihoods.py:70) summed over consecutive pairs (x[2i], x[2i+1])
        float facc = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int d1 = 2 * U * m + 2 * u + 1;
            float x0 = th[2 * u], x1 = th[2 * u + 1];
            float a = x0 * x0 + x1 - 11.f;
            float b = x0 + x1 * x1 - 7.f;
            float v = -(a * a) - b * b;
            facc = facc + ((d1 < D) ? v : 0.f);
        }
        acc = (double)solo_row_sum(facc);
    } else if (lk.id == 4) {
        // Eggbox (likelihoods.py:104-106), x_dim = 2
        const float t0 = solo_lane0(th[0]), t1 = solo_lane0(th[1]);
        float chi = cosf(t0 / 2.f) * cosf(t1 / 2.f);
        float b = 2.f + chi;
        float b2 = b * b;
        acc = (double)(b2 * b2 * b);
    } else {
        // float64 moments of theta (Gaussian, GaussianShell, DoubleGaussianShell: loglike_tile, flow_tile.h)
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 2 * U; ++k) {
            const bool valid = 2 * U * m + k < D;
            const double t = valid ? (double)th[k] : 0.0;
            s1 += t;
            s2 += t * t;
        }
#pragma unroll
        for (int o = 1; o <= 8; o <<= 1) {  // totals over the 16 positions, identical in every lane of a row
            s1 = s1 + __shfl_xor(s1, o);
            s2 = s2 + __shfl_xor(s2, o);
        }
        const double Dd = (double)D;
        if (lk.id == 3) {
            const double c = (double)lk.p[0];
            const double quad = (s2 - c * s1 * s1 / (1.0 + (Dd - 1.0) * c)) / (1.0 - c);
            const double logdet = (Dd - 1.0) * log(1.0 - c) + log(1.0 + (Dd - 1.0) * c);
            acc = -0.5 * quad - 0.5 * logdet - 0.9189385332046727 * Dd;
        } else {
            double sh[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double sig = (double)lk.p[3 * k], rs = (double)lk.p[3 * k + 1], cen = (double)lk.p[3 * k + 2];
                double r2 = s2 - 2.0 * cen * s1 + Dd * cen * cen;
                double rad = sqrt(r2 > 0.0 ? r2 : 0.0);
                sh[k] = -((rad - rs) * (rad - rs)) / (2.0 * sig * sig);
            }
            if (lk.id == 5) acc = sh[0];
            else {
                const double mx = sh[0] > sh[1] ? sh[0] : sh[1], mn = sh[0] > sh[1] ? sh[1] : sh[0];
                acc = mx + log1p(exp(mn - mx));
            }
        }
    }
    if (!(fabs(acc) <= 1.79769313486231570e308)) acc = -1e100;  // logl[~isfinite] = -1e100   sampler.py:128
    return acc;
}
#pragma clang fp contract(fast)

static constexpr int SOLO_ETAB = 1024;   // steps + 2 <= SOLO_ETAB: exp(1 / (1 + k)) from a table

// WPG = walkers (= net waves) per workgroup: 4 (one per SIMD: populations up to 4 x CUs), 8 or 12 (two / three per SIMD; round 4).
// Beyond one wave per SIMD the register file no longer holds a lane's 132 weights beside the state (9 waves -> 168 registers, 13
// waves -> 128), so they are read from the workgroup's field-major LDS copy as at x_dim > 64.
template <int U, int WPG> constexpr bool solo_lds_weights() { return U >= 3 || WPG > 4; }
template <int U, int WPG> constexpr int solo_reg_blocks() { return !solo_lds_weights<U, WPG>() ? 3 : (WPG == 4 ? 1 : (U <= 2 && WPG == 8 ? 1 : 0)); }

enum { MH_SOLO_LAG_INTERNAL = 8 };   // the relay's schedule under lag 0 (no lagged step exists then: only its bookkeeping uses it)
template <int U, bool DBG, int LK, int WPG>
__global__ void __launch_bounds__(64 * (WPG + 2)) mh_kernel_solo(MhArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wlds[];  // the packed weights
    __shared__ __attribute__((aligned(16))) float nbuf[2][WPG][16][2 * U];
    __shared__ float ubuf[2][WPG];
    __shared__ __attribute__((aligned(16))) uint32_t rawbuf[4 * WPG * 8 * U];   // the noise wave's raw draws of a step, stream-major
    __shared__ int acc_lds[2][WPG];   // batch rule: the walkers' accepts of a step, posted by the noise wave
    __shared__ float fsbuf[2];      // the proposal scale that goes with a noise buffer
    __shared__ float fs_exact;      // warm-up steps of the batch rule: the scale of the next step, known only after this step's votes
    __shared__ float fs_cand[2];    // ... and the two values it can take: the batch votes up / down
    __shared__ int vote_sel;        // which of the two it was
    __shared__ double etab[SOLO_ETAB];

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;
    const int D = a.s.D, S = a.steps, C = a.C;
    const bool recorded = DBG && a.noise_dz;
    const bool dynamic = (a.flags & NNEST_MH_DYNAMIC_BATCH) != 0;  // the per-16-walker rule belongs to the 16-walker forms
    const bool use_tab = S + 2 <= SOLO_ETAB;
    // lag 0 = the reference's rule itself (sampler.py:422-431: the vote of step s sets the scale of step s + 1): EVERY step is an
    // exact step -- the two-candidate evaluation below, which hides the vote's round trip -- and the last step's vote is applied
    // too, for scale_out (round 5; round 4 sent lag 0 to the quad form, which waits out a grid-wide round trip per step).
    const bool exact_all = dynamic && mh_flag_lag(a.flags) == 0;
    const bool direct_votes = (a.flags & NNEST_MH_WINDOW_VOTES_ONLY) == 0;
    const int lag = exact_all ? MH_SOLO_LAG_INTERNAL : mh_flag_lag(a.flags);   // otherwise >= 3 (solo_form_eligible): the rule is always relayed
    // the first `warm` steps apply the rule exactly (a grid-wide wait per step, hidden behind two evaluations) -- where its gain
    // 1 / (1 + n) is large and the scale still far from where it settles -- the rest `lag` steps behind (mh_common.h); at most S - 1
    const int warm = dynamic ? min(exact_all ? S : mh_flag_warm(a.flags), S - 1) : 0;
    constexpr bool LDSW = solo_lds_weights<U, WPG>();
    constexpr int RB = solo_reg_blocks<U, WPG>();
    const int ntiles = (C + WPG - 1) / WPG;
    if (tile >= ntiles) {
        // batch-wide step rule (mh_common.h): the one workgroup behind the tiles sums every step's counters as soon as they
        // are complete and publishes the total; its other waves leave at once
        if (dynamic && wave == 0) mh_sync_publisher(a.sync, S, exact_all ? S : max(S - lag, warm), ntiles, C, lane, a.sync_err);
        else if (dynamic && (a.flags & (NNEST_MH_SYNC_ZERO_NEXT | NNEST_MH_SYNC_ZERO_PREV))) {
            // this workgroup's other waves: the OTHER half of the caller's double buffer, zeroed for the next launch (nothing of this
            // launch touches it; the kernel boundary orders it in front of the next one)
            const size_t W = (size_t)mh_sync_words(S) + 1;   // (+ the error word)
            unsigned long long *other = (a.flags & NNEST_MH_SYNC_ZERO_NEXT) ? a.sync + W : a.sync - W;
            const int nthr = (int)blockDim.x - 64;
            for (size_t i = threadIdx.x - 64; i < W; i += nthr) other[i] = 0ull;
        }
        return;
    }
    {
        if constexpr (!LDSW) {
            const int n = a.s.nets_params();
            for (int i = threadIdx.x; i < n; i += blockDim.x) wlds[i] = a.packed[i];
        } else {   // net wave b (of waves 0..2) gathers block b's lane shares from the packed vector into the field-major LDS copy
            if (wave < 3) {
                SoloNet<U> nb;
                solo_gather<U>(nb, a.packed + (size_t)(wave * 2 + (lane >= 32 ? 1 : 0)) * a.s.net_params, D, (wave + 1) & 1, wave & 1, lane);
                solo4_store<U>(wlds, wave, nb, lane);
            }
        }
        if (dynamic && use_tab)
            for (int k = threadIdx.x; k < S + 2; k += blockDim.x) etab[k] = exp(1.0 / (double)(1 + k));
    }
    __syncthreads();

    if (wave == WPG) {
        // proposal noise: the streams of the 16-walker forms -- per (walker, lane group g) 8 normals for dims
        // 32 t + 8 g + [0, 8), t < U -- generated by 4 WPG lanes (walkers x 4 groups) and written where the net waves read them.
        // Buffer k carries the noise of step k + 1; its scale (fsbuf) comes from the relay wave under the batch rule.
        // Round 4: the draws of a stream are sequential, so only 4 WPG lanes can advance the generators -- but the Box-Muller
        // transform of a draw pair (v_log / v_sqrt / v_sin / v_cos, half of the wave's cycles when 16 lanes did 8 pairs each) is
        // not: the raw 32-bit draws go through an LDS staging row and all 64 lanes transform WPG U / 8 quads (4 draws -> 4
        // normals) each, with the arithmetic of box_muller() -- the normals are bit for bit those of XoshiroNoise::next.
        const int g = lane & 3, j = min(lane >> 2, WPG - 1);
        const bool gen = lane < 4 * WPG;
        Xoshiro128 rn = xoshiro_seed(a.seed, a.walker_offset + (uint64_t)(tile * WPG + j), (uint32_t)g, NOISE_STREAM_DZ);
        Xoshiro128 ru = xoshiro_seed(a.seed, a.walker_offset + (uint64_t)(tile * WPG + j), 0xffffffffu, NOISE_STREAM_U);
        constexpr int NQ = 8 * WPG * U;   // quads per step: (4 WPG streams) x (2 U quads each)
        unsigned long long n0 = 0, n1 = 0, n2 = 0, n_gen = 0, n_all = 0;
        (void)n0; (void)n1; (void)n2; (void)n_gen; (void)n_all;
        // Buffer k (the noise of step k + 1) is generated ONE barrier ahead of the barrier that publishes it, so that under the
        // warm-up's order of barriers (P0, P1, then A_k, P_{k+1}, B_k per exact step: the net waves' loop) no barrier waits for a draw.
        auto draw = [&](int k) {
            if (k > S) return;
            STAMP(n0);
            if (!recorded) {
                if (gen) {
#pragma unroll
                    for (int t = 0; t < 2 * U; ++t) {   // stream order: tile t / 2, draws 4 (t & 1) .. + 3 of its eight
                        u32x4 r;
                        r.x = xoshiro_next(rn); r.y = xoshiro_next(rn); r.z = xoshiro_next(rn); r.w = xoshiro_next(rn);
                        *reinterpret_cast<u32x4 *>(&rawbuf[lane * 8 * U + 4 * t]) = r;
                    }
                    const float u = xoshiro_uniform(ru);
                    if (g == 0) ubuf[k & 1][j] = u;
                }
                float *nb = &nbuf[k & 1][0][0][0];
#pragma unroll
                for (int i = 0; i < (NQ + 63) / 64; ++i) {
                    const int q = lane + 64 * i;
                    if (NQ % 64 == 0 || q < NQ) {
                        const u32x4 r = *reinterpret_cast<const u32x4 *>(&rawbuf[4 * q]);
                        const int sl = q / (2 * U), qq = q % (2 * U);          // stream (walker sl >> 2, lane group sl & 3), quad of the stream
                        const int d0 = 32 * (qq >> 1) + 8 * (sl & 3) + 4 * (qq & 1);  // first of the quad's four dims (padded dims, d >= D, carry 0)
                        float o0, o1, o2, o3;
                        box_muller(r.x, r.y, o0, o1);
                        box_muller(r.z, r.w, o2, o3);
                        const f32x4 o = {d0 < D ? o0 : 0.f, d0 + 1 < D ? o1 : 0.f, d0 + 2 < D ? o2 : 0.f, d0 + 3 < D ? o3 : 0.f};
                        *reinterpret_cast<f32x4 *>(nb + (sl >> 2) * 32 * U + d0) = o;
                    }
                }
            }
            if (!dynamic && lane == 0) fsbuf[k & 1] = a.step_size;
            STAMP(n1);
#ifdef NNEST_STAMP
            n_gen += n1 - n0;
#endif
        };
        STAMP(n2);
        draw(0);
        solo_barrier();                    // P0
        if (S >= 1) {
            draw(1);
            solo_barrier();                // P1
            draw(2);
            for (int k = 1; k <= warm; ++k) {
                solo_barrier();            // A_k
                solo_barrier();            // P_{k+1}
                draw(k + 2);
                solo_barrier();            // B_k
            }
            for (int k = warm > 0 ? warm + 2 : 2; k <= S; ++k) {
                solo_barrier();            // P_k
                draw(k + 1);
            }
        }
#ifdef NNEST_STAMP
        { unsigned long long n3 = 0; STAMP(n3); n_all = n3 - n2; }
#endif
#ifdef NNEST_STAMP
        if (a.scale_out && lane == 0 && tile == 0) { a.scale_out[4] = (float)n_gen; a.scale_out[5] = (float)n_all; }
#endif
#ifndef NNEST_STAMP
        if (!dynamic && a.scale_out && lane == 0) {   // scale_out has one entry per 16 walkers (nnest_mh_num_groups): the tile that holds a group's first walker reports
            const int g0 = (tile * WPG + 15) >> 4;
            if (16 * g0 < (tile + 1) * WPG && 16 * g0 < C) a.scale_out[g0] = a.step_size;
        }
#endif
        return;
    }
    if (wave == WPG + 1) {
        // The batch-wide step rule (sampler.py:422-431, `lag` steps behind; mh_common.h) lives in this wave entirely: it posts the
        // tile's accepted count, learns the batch's votes, keeps (accept, reject, scale) in the reference's float64 arithmetic and
        // hands the net waves the float32 scale of their next proposal -- a step of the net waves holds neither a global-memory
        // operation nor the rule's arithmetic.  (Round 3 had the noise wave do this too: its draws and the rule's round trips
        // then added up to more than a step of the net waves, who waited ~210 cycles per step at the barrier; as two waves they overlap.)
        if (!dynamic) return;   // a wave that has ended no longer counts at s_barrier
        double scale = (double)a.step_size;   // python float in the reference (sampler.py:255, :428-431)
        int accept = 0, reject = 0;
        const int last = S - lag;  // last step whose vote is ever applied
        // The votes arrive as "window" words (every decision published so far in one 64-bit word, mh_common.h), requested TWO
        // iterations before they are needed: a round trip to the memory side (~1.5 us) is longer than an iteration (~1.2 us).
        // hipcc waits for a load right where it is issued as soon as its destination is a loop-carried variable, so the requests
        // are inline asm (sc1: served by the memory side, as the relaxed agent-scope loads elsewhere) and the wait is explicit.
        // Every iteration issues exactly ONE atomic and ONE request, in that order and after its own consumption, so
        // "vmcnt(2)" -- everything but the two youngest operations has completed -- is exactly "the request made two
        // iterations ago has landed" (vector-memory operations complete in order, MI355X_MICROARCH.md).  The loop runs two
        // iterations per trip: each request has a register pair of its own (qa / qb) and is never moved.
        const unsigned long long *wbase = a.sync + 2 * mh_sync_counter_words(S) + (size_t)(tile & (MH_SYNC_SHARDS - 1)) * MH_SYNC_STRIDE;
#define SOLO_RELAY_ITERATION(k_, q_)                                                                                        \
        {                                                                                                                   \
            const int k = (k_), want = k - lag;                                                                             \
            /* the wait is UNCONDITIONAL: it is also what tells the compiler that q_ is read in every iteration.  Where the */  \
            /* compiler can see that the first iterations never consume (want <= warm) it treats the request's destination  */  \
            /* registers as free, hands them to another value, and the load lands on top of it (round 4: the accept count   */  \
            /* posted for step warm + 2 came out as a window word in ~4 % of the launches -- a batch total that never       */  \
            /* completes, 126 ms until the bounded wait gives up).                                                          */  \
            asm volatile("s_waitcnt vmcnt(2)" : "+v"(q_) : : "memory");                                                     \
            if (want > warm) {                                                                                              \
                const bool up = mh_window_vote(a.sync, S, want, tile, q_, a.sync_err);                                      \
                if (up) accept += 1; else reject += 1;                                                                      \
                if (accept > reject) scale *= use_tab ? etab[accept] : exp(1.0 / (1 + accept));                             \
                if (accept < reject) scale /= use_tab ? etab[reject] : exp(1.0 / (1 + reject));                             \
            }                                                                                                               \
            /* between the barriers k - 1 and k the net waves run step k - 1: the accepts of step k - 2 are in LDS */       \
            /* (no step of its own to post: a zero to the unused slot of step 0 -- every iteration issues the same operations) */ \
            if (lane == 0) {                                                                                                \
                const int *ac = acc_lds[k & 1];                                                                             \
                const bool mine = k - 2 > warm; /* (the warm-up posted its steps itself) */                                \
                int acs = 0;                                                                                                \
                _Pragma("unroll") for (int w_ = 0; w_ < WPG; ++w_) acs += ac[w_];                                           \
                mh_sync_post(a.sync, mine ? k - 2 : 0, tile, mine ? acs : 0);                                               \
            }                                                                                                               \
            const int ask = min(max(want + 2, 1), max(last, 1));   /* the window of the step needed two iterations on */      \
            const unsigned long long *wp = wbase + (size_t)((ask - 1) >> 5) * MH_SYNC_SHARDS * MH_SYNC_STRIDE;              \
            asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(q_) : "v"(wp) : "memory");                            \
            /* buffer k carries the noise of step k + 1 and its scale: the scale after the votes of the steps <= k - lag */  \
            if (lane == 0) fsbuf[k & 1] = (float)scale;                                                                     \
            solo_barrier(); /* publish buffer k */                                                                          \
        }
        unsigned long long qa = 0, qb = 0;
        int k0 = 0;
        if (warm > 0) {
            // Warm-up: behind the net waves' step k (barrier A) post that step's accepts and wait for the batch's vote while the
            // net waves evaluate step k + 1 for BOTH scales it can lead to (fs_cand, written one vote ahead); at barrier B name
            // the one it was (vote_sel) and the scale itself (fs_exact).  Barriers in the net waves' order: P0, P1, then
            // A_k, P_{k+1}, B_k per exact step.
            auto E = [&](int n) { return use_tab ? etab[n] : exp(1.0 / (1 + n)); };
            auto next_scale = [&](int acc, int rej, double sc) {   // sampler.py:428-431
                if (acc > rej) sc *= E(acc);
                if (acc < rej) sc /= E(rej);
                return sc;
            };
            if (lane == 0) {
                fsbuf[0] = (float)scale;
                fs_cand[0] = (float)next_scale(accept + 1, reject, scale);
                fs_cand[1] = (float)next_scale(accept, reject + 1, scale);
            }
            solo_barrier();                   // P0
            if (lane == 0) fsbuf[1] = (float)scale;
            solo_barrier();                   // P1
            for (int k = 1; k <= warm; ++k) {
                solo_barrier();               // A_k: the net waves have decided step k
                if (lane == 0) {
                    const int *ac = acc_lds[k & 1];
                    int acs = 0;
#pragma unroll
                    for (int w_ = 0; w_ < WPG; ++w_) acs += ac[w_];
                    mh_sync_post(a.sync, k, tile, acs);
                    fsbuf[(k + 1) & 1] = (float)scale;   // (buffer k + 1's scale is not used: step k + 2 takes fs_exact or a candidate)
                }
                solo_barrier();               // P_{k+1}
                // (an exact step's vote is read from the tiles' counters themselves -- one memory-side hop less than the window word
                // the publisher workgroup writes for the lagged steps: counters -> publisher -> window; NNEST_SOLO_VOTE=window keeps the latter)
                const bool up = direct_votes ? 2 * mh_sync_total(a.sync, k, ntiles, 0ull, a.sync_err) > C
                                             : mh_window_vote(a.sync, S, k, tile, 0ull, a.sync_err);
                if (up) accept += 1; else reject += 1;
                scale = next_scale(accept, reject, scale);
                if (lane == 0) {
                    vote_sel = up ? 0 : 1;
                    fs_exact = (float)scale;
                    fs_cand[0] = (float)next_scale(accept + 1, reject, scale);
                    fs_cand[1] = (float)next_scale(accept, reject + 1, scale);
                }
                solo_barrier();               // B_k
            }
            k0 = warm + 2;
        }
        for (int k2 = k0; k2 <= S; k2 += 2) {
            SOLO_RELAY_ITERATION(k2, qa)
            if (k2 + 1 <= S) SOLO_RELAY_ITERATION(k2 + 1, qb)
        }
#undef SOLO_RELAY_ITERATION
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(qa), "+v"(qb) : : "memory");   // no request outlives its registers
        if (exact_all) {   // the vote of the LAST step: no step follows it, but the reference updates the scale all the same (scale_out)
            solo_barrier();               // the net waves have decided step S
            if (lane == 0) {
                const int *ac = acc_lds[S & 1];
                int acs = 0;
#pragma unroll
                for (int w_ = 0; w_ < WPG; ++w_) acs += ac[w_];
                mh_sync_post(a.sync, S, tile, acs);
            }
            const bool up = direct_votes ? 2 * mh_sync_total(a.sync, S, ntiles, 0ull, a.sync_err) > C : mh_window_vote(a.sync, S, S, tile, 0ull, a.sync_err);
            if (up) accept += 1; else reject += 1;
            if (accept > reject) scale *= use_tab ? etab[accept] : exp(1.0 / (1 + accept));
            if (accept < reject) scale /= use_tab ? etab[reject] : exp(1.0 / (1 + reject));
        }
#ifndef NNEST_STAMP
        // scale_out has one entry per 16 walkers (nnest_mh_num_groups): the tile that holds a group's first walker reports
        if (a.scale_out && lane == 0) {
            const int g0 = (tile * WPG + 15) >> 4;
            if (16 * g0 < (tile + 1) * WPG && 16 * g0 < C) a.scale_out[g0] = (float)scale;
        }
#endif
        return;
    }

    // ---- one walker per wave ----
    const int j = wave;
    const int pos = lane & 15;
    const bool translate_half = lane >= 32;
    const bool writer_lane = lane < 16;  // row (n = 0, h = 0) does the stores
    const int row = tile * WPG + j;
    const bool ok = row < C;
    const LikeSpec like = a.like;
    const double loglstar = a.loglstar;
    const bool free_mode = (a.flags & NNEST_MH_UNCONSTRAINED) != 0;

    SoloNet<U> net[RB > 0 ? RB : 1];   // (weights in LDS: blocks 0 and 2 stay there, block 1 in registers where they fit, solo_coupling_inverse4)
    if constexpr (LDSW) {
        if constexpr (RB == 1)
            solo_gather<U>(net[0], a.packed + (size_t)(1 * 2 + (translate_half ? 1 : 0)) * a.s.net_params, D, (1 + 1) & 1, 1 & 1, lane);
    } else {
#pragma unroll
        for (int b = 0; b < 3; ++b)
            solo_gather<U>(net[b], wlds + (size_t)(b * 2 + (translate_half ? 1 : 0)) * a.s.net_params, D, (b + 1) & 1, b & 1, lane);
    }
    // NormalizingFlow.inverse (networks.py:34-42), num_blocks = 3: blocks 2, 1, 0; block b conditions on class (b+1)&1 and
    // transforms class b&1
    const unsigned sel = translate_half ? 0xffffffffu : 0u;
    const bool h1 = (lane & 16) != 0;
    auto inverse = [&](float (&xs)[2][U]) {
        if constexpr (LDSW) {
            float ld = solo_coupling_inverse4<U>(Solo4Lds{wlds + (size_t)2 * SOLO4_NF * 64, lane}, sel, h1, xs[1], xs[0]);
            if constexpr (RB == 1) ld += solo_coupling_inverse4<U>(Solo4Reg<U>{net[0]}, sel, h1, xs[0], xs[1]);
            else ld += solo_coupling_inverse4<U>(Solo4Lds{wlds + (size_t)1 * SOLO4_NF * 64, lane}, sel, h1, xs[0], xs[1]);
            ld += solo_coupling_inverse4<U>(Solo4Lds{wlds, lane}, sel, h1, xs[1], xs[0]);
            return ld;
        } else {
            float ld = solo_coupling_inverse<U>(net[2], sel, h1, xs[1], xs[0]);
            ld += solo_coupling_inverse<U>(net[1], sel, h1, xs[0], xs[1]);
            ld += solo_coupling_inverse<U>(net[0], sel, h1, xs[1], xs[0]);
            return ld;
        }
    };

    float z[2][U], x[2][U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int d = 2 * U * pos + 2 * u + c;
            z[c][u] = (ok && d < D) ? a.z[(size_t)row * D + d] : 0.f;
            x[c][u] = z[c][u];
        }
    float ld = solo_logdet_total(inverse(x));  // x = f^-1(z), log_det_J  (sampler.py:266, :295)
    double logl = ok ? a.logl[row] : 0.0;
    int n_acc = 0, n_call = 0;

    auto store_row = [&](float *base, size_t r, const float (&v)[2][U]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int d = 2 * U * pos + 2 * u + c;
                if (d < D) base[r * D + d] = v[c][u];
            }
    };
    // the chain's first x stays in the x output buffer for the launch: the reference counts a chain only if EVERY coordinate of its
    // last x differs from its first (nested.py:432), tested at the end (no register is held for it)
    if (writer_lane && ok && a.x) store_row(a.x, (size_t)row, x);
    if (DBG && writer_lane && ok) {
        if (a.hist_x) store_row(a.hist_x, (size_t)row * (S + 1), x);
        if (a.hist_logl && pos == 0) a.hist_logl[(size_t)row * (S + 1)] = logl;
    }

    float nz[2 * U], u_next = 0.f, fs_next = a.step_size;
    int kbuf = 0;
    auto fetch_noise = [&]() {
        solo_barrier();  // the noise wave has published buffer kbuf: noise, accept uniform and scale of the next step
        if (!recorded) {
#pragma unroll
            for (int k = 0; k < 2 * U; ++k) nz[k] = nbuf[kbuf & 1][j][pos][k];
            u_next = ubuf[kbuf & 1][j];
        }
        fs_next = fsbuf[kbuf & 1];
        ++kbuf;
    };
    fetch_noise();

    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, a_prop = 0, a_inv = 0, a_post = 0, a_tot = 0;
    (void)st0; (void)st1; (void)st2; (void)st3; (void)a_prop; (void)a_inv; (void)a_post; (void)a_tot;
    // One Metropolis step in three parts, so that the warm-up below can evaluate a step for TWO candidate scales:
    //   propose   z' = z + randn * scale  (sampler.py:310, :316), float32 like torch -- from the noise held in registers;
    //   finish    x' = f^-1(z'), box prior, Jacobian ratio, likelihood, the decision (sampler.py:321-361 / :396-410);
    //   apply     counters, z / x / log-det / logL of the accepted proposal (sampler.py:342-365), the accept flag for the rule.
    struct Prop { float zp[2][U], xp[2][U]; float ldp; double lp; bool pre, acc; };
    auto propose = [&](int it, float fs, float (&zp)[2][U], float &u) {
        if (recorded) {
#pragma unroll
            for (int uu = 0; uu < U; ++uu)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int d = 2 * U * pos + 2 * uu + c;
                    const float dz = (ok && d < D) ? a.noise_dz[((size_t)(it - 1) * C + row) * D + d] : 0.f;
                    zp[c][uu] = __builtin_fmaf(dz, fs, z[c][uu]);
                }
            u = ok ? a.noise_u[(size_t)(it - 1) * C + row] : 1.f;
        } else {
#pragma unroll
            for (int uu = 0; uu < U; ++uu) {
                zp[0][uu] = __builtin_fmaf(nz[2 * uu], fs, z[0][uu]);       // (one fused multiply-add, spelled out: the step is compiled
                zp[1][uu] = __builtin_fmaf(nz[2 * uu + 1], fs, z[1][uu]);   // in three places and all of them must round alike)
            }
            u = u_next;
        }
    };
    auto finish = [&](Prop &r, float u) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int uu = 0; uu < U; ++uu) r.xp[c][uu] = r.zp[c][uu];
        STAMP(st1);
        r.ldp = solo_logdet_total(inverse(r.xp));  // sampler.py:321
        STAMP(st2);
        // log_ratio = log_det_J' - log_det_J, -inf outside the prior box  (sampler.py:326-331); UniformPrior(D,-1,1)
        // (priors.py:39-43): NaN compares false, i.e. counts as inside; padded dims hold 0
        int okl = 1;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int uu = 0; uu < U; ++uu) okl &= !(r.xp[c][uu] < -1.f || r.xp[c][uu] > 1.f);
        const bool inb = __ballot(okl != 0) == ~0ull;
        float log_ratio = inb ? (r.ldp - ld) : -INFINITY;
        float ratio = fminf(__expf(log_ratio), 1.0f);  // exp().clamp(max=1)  :335
        if (log_ratio != log_ratio) ratio = log_ratio;  // NaN stays NaN (u < NaN is false, as in torch)
        r.pre = ok && (u < ratio);                      // :336
        r.lp = solo_loglike<U, LK>(like, D, lane, r.xp);
        r.acc = r.pre && (r.lp > loglstar);  // :361
        if (free_mode) {  // sampler.py:396-410
            const double lr = inb ? (double)(r.ldp - ld) + (r.lp - logl) : -INFINITY;
            const double rt = fmin(exp(lr), 1.0);
            r.acc = ok && ((double)u < rt);
        }
    };
    auto apply = [&](int it, const Prop &r) {
        n_call += (free_mode ? ok : r.pre) ? 1 : 0;
        n_acc += r.acc ? 1 : 0;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int uu = 0; uu < U; ++uu) {
                z[c][uu] = r.acc ? r.zp[c][uu] : z[c][uu];
                x[c][uu] = r.acc ? r.xp[c][uu] : x[c][uu];
            }
        ld = r.acc ? r.ldp : ld;
        logl = r.acc ? r.lp : logl;
        if (dynamic && lane == 0) acc_lds[it & 1][j] = r.acc ? 1 : 0;  // sampler.py:422-431: counted by the relay wave after the next barrier
        if (DBG && writer_lane && ok) {
            if (a.hist_x) store_row(a.hist_x, (size_t)row * (S + 1) + it, x);
            if (a.hist_logl && pos == 0) a.hist_logl[(size_t)row * (S + 1) + it] = logl;
        }
    };
    Prop cur;
    bool have = false;   // `cur` already holds this step's evaluated proposal (warm-up: chosen among the two candidates)
    for (int it = 1; it <= S; ++it) {
        STAMP(st0);
        if (!have) {
            float u;
            propose(it, fs_next, cur.zp, u);
            fetch_noise();
            finish(cur, u);
        }
        apply(it, cur);
        have = false;
        if (it <= warm) {
            // Warm-up (round 4).  The scale of step it + 1 exists only after the whole batch's vote on step it -- a grid-wide
            // round trip (~2.5 us) that round 3 spent waiting.  There are only TWO scales it can be (the vote is up or down;
            // the relay wave works both out beforehand, in the rule's own float64 arithmetic), so step it + 1 is evaluated for
            // both while the vote travels, and the one the vote names is kept: the chain is, bit for bit, the one that waits.
            solo_barrier();                    // A: the relay posts this step's accepts and starts waiting for the vote
            const float f0 = fs_cand[0], f1 = fs_cand[1];
            Prop c1;
            float u;
            propose(it + 1, f0, cur.zp, u);
            propose(it + 1, f1, c1.zp, u);
            fetch_noise();                     // (the noise of step it + 2)
            finish(cur, u);
            finish(c1, u);
            solo_barrier();                    // B: the vote is in
            if (vote_sel != 0) cur = c1;
            have = true;
            fs_next = fs_exact;                // (step warm + 2 runs at the scale the exact steps end with: no lagged vote is due yet)
        }
        STAMP(st3);
#ifdef NNEST_STAMP
        a_prop += st1 - st0; a_inv += st2 - st1; a_post += st3 - st2; a_tot += st3 - st0;
#endif
    }
    if (exact_all) solo_barrier();   // (the relay wave reads the last step's accepts behind it)
#ifdef NNEST_STAMP
    if (a.scale_out && lane == 0 && tile == 0) {
        float *o = a.scale_out;
        if (j == 0) { o[0] = (float)a_tot; o[1] = (float)a_prop; o[2] = (float)a_inv; o[3] = (float)a_post; }
        if (j < 4) { o[8 + 2 * j] = (float)a_inv; o[9 + 2 * j] = (float)a_post; }   // the four net waves: which one shares its SIMD with the noise wave
    }
#endif
    bool all_moved = n_acc > 0;   // (no x buffer: the accept count stands in)
    if (a.x) {
        bool mine = true;
        if (writer_lane) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int d = 2 * U * pos + 2 * u + c;
                    const float x0 = (ok && d < D) ? a.x[(size_t)row * D + d] : 0.f;
                    mine = mine && (d >= D || x[c][u] != x0);
                }
        }
        all_moved = (__ballot(mine) & 0xffffull) == 0xffffull;   // the 16 positions (the writer row holds the walker once)
    }
    if (writer_lane && ok) {
        store_row(a.z, (size_t)row, z);
        if (a.x) store_row(a.x, (size_t)row, x);
        if (pos == 0) {
            a.logl[row] = logl;
            if (a.n_accept) a.n_accept[row] = n_acc | (all_moved ? NNEST_MH_ALL_MOVED : 0);
            if (a.n_call) a.n_call[row] = n_call;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// SLICE proposal in latent space (BASELINE north_star: "the slice/MH proposal step in latent space").
// ABSENT FROM THE REFERENCE -- nnest/sampler.py:310-316 proposes Gaussian random-walk Metropolis moves only -- so the step is
// build-defined and its parity unpinned: univariate slice sampling (Neal 2003: stepping out + shrinkage) along a
// random direction of latent space, of the SAME target the reference's constrained Metropolis step leaves invariant
// (sampler.py:326-361): pi(z) ~ |det dx/dz| (the uniform prior's density seen from latent space) on {x(z) in the prior box,
// logL(x(z)) > L*}.  Per step and walker:
//   eps ~ N(0, I_D) (noise_normal4, stream DZ);  candidates z(t) = z + t * width * eps,  t in R
//   log y = log|det|(z) + log u_1;   inside(t) := x(z(t)) in the box  and  log|det|(z(t)) > log y  and  logL(x(z(t))) > L*
//   bracket [t_l, t_r] = [-u_0, 1 - u_0];  stepping out: while inside(t_l) and fewer than `max_out` steps: t_l -= 1 (the same to the right)
//   shrinkage: t = t_l + (t_r - t_l) u_k (k = 2, 3, ...); inside(t) -> the walker moves there; else the bracket's end on t's side
//   becomes t; after `max_shrink` draws the walker stays.
// Every evaluation is one "eval" of the hot path (coupling-stack inverse + log-det + box + likelihood); n_call counts, as the
// Metropolis kernel does, the candidates whose likelihood decided (those that passed the box and the slice level).
// One walker per wave (the solo layout): the data-dependent loops of a walker are uniform over its wave, walkers do not wait for
// each other, and no step crosses a workgroup -- any population, no resident-grid requirement.  u_k = noise_uniform(seed, walker,
// 64 step + k): exact in float32, so the CPU checker of the tests restates them word for word; the normals are exported for
// it (nnest_slice_fill_noise), like nnest_mh_fill_noise exports the Metropolis kernel's.
struct SliceArgs {
    FlowShape s;
    const float *packed;
    float *z, *x;
    double *logl;
    double loglstar;
    float width;
    int steps, C, max_out, max_shrink;
    LikeSpec like;
    uint64_t seed, walker_offset;
    const float *noise_dz;   // recorded directions [steps][C][D] (tests) or NULL
    float *hist_x;           // [C][steps + 1][D] or NULL
    int *n_call, *n_move, *n_eval;
};

template <int U>   // this lane's 2U normals of (walker, step): dims 2U pos + 2u + c, the quads of noise_normal4
static __device__ __forceinline__ void slice_normals(uint64_t seed, uint64_t walker, uint32_t step, int pos, int D, float (&e)[2][U]) {
    const int d0 = 2 * U * pos;
    f32x4 q[(2 * U + 3) / 4 + 1];
    const int q0 = d0 >> 2;
#pragma unroll
    for (int k = 0; k < (2 * U + 3) / 4 + 1; ++k) q[k] = noise_normal4(seed, walker, step, (uint32_t)(q0 + k), NOISE_STREAM_DZ);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int d = d0 + 2 * u + c, k = (d >> 2) - q0, r = d & 3;
            float v = 0.f;
#pragma unroll
            for (int kk = 0; kk < (2 * U + 3) / 4 + 1; ++kk)
                if (kk == k) v = r == 0 ? q[kk].x : (r == 1 ? q[kk].y : (r == 2 ? q[kk].z : q[kk].w));
            e[c][u] = d < D ? v : 0.f;
        }
}

template <int U, int LK>
__global__ void __launch_bounds__(256) slice_kernel_solo(SliceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = a.s.D, S = a.steps, C = a.C;
    constexpr bool LDSW = solo_lds_weights<U, 4>();
    {
        if constexpr (!LDSW) {
            const int n = a.s.nets_params();
            for (int i = threadIdx.x; i < n; i += blockDim.x) wlds[i] = a.packed[i];
        } else if (wave < 3) {
            SoloNet<U> nb;
            solo_gather<U>(nb, a.packed + (size_t)(wave * 2 + (lane >= 32 ? 1 : 0)) * a.s.net_params, D, (wave + 1) & 1, wave & 1, lane);
            solo4_store<U>(wlds, wave, nb, lane);
        }
    }
    __syncthreads();
    const int pos = lane & 15;
    const bool translate_half = lane >= 32, writer_lane = lane < 16;
    const int row = blockIdx.x * 4 + wave;
    const bool ok = row < C;
    if (!ok) return;   // (no barrier behind this point)
    const LikeSpec like = a.like;
    const double loglstar = a.loglstar;
    SoloNet<U> net[LDSW ? 1 : 3];
    if constexpr (!LDSW) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
            solo_gather<U>(net[b], wlds + (size_t)(b * 2 + (translate_half ? 1 : 0)) * a.s.net_params, D, (b + 1) & 1, b & 1, lane);
    }
    const unsigned sel = translate_half ? 0xffffffffu : 0u;
    const bool h1 = (lane & 16) != 0;
    auto inverse = [&](float (&xs)[2][U]) {   // NormalizingFlow.inverse (networks.py:34-42), blocks 2, 1, 0
        if constexpr (LDSW) {
            float ld = solo_coupling_inverse4<U>(Solo4Lds{wlds + (size_t)2 * SOLO4_NF * 64, lane}, sel, h1, xs[1], xs[0]);
            ld += solo_coupling_inverse4<U>(Solo4Lds{wlds + (size_t)1 * SOLO4_NF * 64, lane}, sel, h1, xs[0], xs[1]);
            ld += solo_coupling_inverse4<U>(Solo4Lds{wlds, lane}, sel, h1, xs[1], xs[0]);
            return ld;
        } else {
            float ld = solo_coupling_inverse<U>(net[2], sel, h1, xs[1], xs[0]);
            ld += solo_coupling_inverse<U>(net[1], sel, h1, xs[0], xs[1]);
            ld += solo_coupling_inverse<U>(net[0], sel, h1, xs[1], xs[0]);
            return ld;
        }
    };
    auto store_row = [&](float *base, size_t r, const float (&v)[2][U]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int d = 2 * U * pos + 2 * u + c;
                if (d < D) base[r * D + d] = v[c][u];
            }
    };
    float z[2][U], x[2][U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int d = 2 * U * pos + 2 * u + c;
            z[c][u] = d < D ? a.z[(size_t)row * D + d] : 0.f;
            x[c][u] = z[c][u];
        }
    float ld = solo_logdet_total(inverse(x));
    double logl = a.logl[row];
    if (writer_lane && a.x) store_row(a.x, (size_t)row, x);   // the chain's first x (usable-chain test at the end, nested.py:432)
    if (writer_lane && a.hist_x) store_row(a.hist_x, (size_t)row * (S + 1), x);
    const uint64_t walker = a.walker_offset + (uint64_t)row;
    int n_call = 0, n_move = 0, n_eval = 0;
    float zc[2][U], xc[2][U], ldc = 0.f;
    double lc = 0.0;
    for (int it = 1; it <= S; ++it) {
        float e[2][U];
        if (a.noise_dz) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int d = 2 * U * pos + 2 * u + c;
                    e[c][u] = d < D ? a.noise_dz[((size_t)(it - 1) * C + row) * D + d] : 0.f;
                }
        } else {
            slice_normals<U>(a.seed, walker, (uint32_t)it, pos, D, e);
        }
        const float u0 = noise_uniform(a.seed, walker, 64u * (uint32_t)it + 0u), u1 = noise_uniform(a.seed, walker, 64u * (uint32_t)it + 1u);
        const float logy = ld + __logf(u1);   // (u1 = 0: -inf, the whole feasible line is the slice)
        // one evaluation: candidate t -> (zc, xc, ldc, lc); true if it lies in the slice
        auto inside = [&](float t) -> bool {
            const float tw = t * a.width;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) { zc[c][u] = __builtin_fmaf(e[c][u], tw, z[c][u]); xc[c][u] = zc[c][u]; }
            ldc = solo_logdet_total(inverse(xc));
            int okl = 1;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int u = 0; u < U; ++u) okl &= !(xc[c][u] < -1.f || xc[c][u] > 1.f);   // UniformPrior(D, -1, 1): priors.py:39-43
            const bool inb = __ballot(okl != 0) == ~0ull;
            const bool pre = inb && (ldc > logy);
            lc = solo_loglike<U, LK>(like, D, lane, xc);
            n_eval += 1;
            n_call += pre ? 1 : 0;
            return pre && (lc > loglstar);
        };
        float tl = -u0, tr = 1.0f - u0;
        for (int j = 0; j < a.max_out; ++j) { if (!inside(tl)) break; tl -= 1.0f; }
        for (int j = 0; j < a.max_out; ++j) { if (!inside(tr)) break; tr += 1.0f; }
        bool moved = false;
        for (int k = 0; k < a.max_shrink; ++k) {
            const float uk = noise_uniform(a.seed, walker, 64u * (uint32_t)it + 2u + (uint32_t)k);
            const float t = __builtin_fmaf(tr - tl, uk, tl);
            if (inside(t)) { moved = true; break; }
            if (t < 0.f) tl = t; else tr = t;
        }
        if (moved) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int u = 0; u < U; ++u) { z[c][u] = zc[c][u]; x[c][u] = xc[c][u]; }
            ld = ldc; logl = lc; n_move += 1;
        }
        if (writer_lane && a.hist_x) store_row(a.hist_x, (size_t)row * (S + 1) + it, x);
    }
    bool all_moved = n_move > 0;
    if (a.x) {
        bool mine = true;
        if (writer_lane) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int d = 2 * U * pos + 2 * u + c;
                    const float x0 = d < D ? a.x[(size_t)row * D + d] : 0.f;
                    mine = mine && (d >= D || x[c][u] != x0);
                }
        }
        all_moved = (__ballot(mine) & 0xffffull) == 0xffffull;
    }
    if (writer_lane) {
        store_row(a.z, (size_t)row, z);
        if (a.x) store_row(a.x, (size_t)row, x);
        if (pos == 0) {
            a.logl[row] = logl;
            if (a.n_call) a.n_call[row] = n_call;
            if (a.n_move) a.n_move[row] = n_move | (all_moved ? NNEST_MH_ALL_MOVED : 0);
            if (a.n_eval) a.n_eval[row] = n_eval;
        }
    }
}

// the directions the slice kernel draws (noise_normal4, stream DZ), exported for the checker: dz [steps][C][D]
__global__ void slice_fill_noise_kernel(float *__restrict__ dz, int steps, int C, int D, uint64_t seed, uint64_t walker_offset) {
    const long nq = (long)steps * C * ((D + 3) / 4);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (long)gridDim.x * blockDim.x) {
        const int Q = (D + 3) / 4, q = (int)(i % Q), c = (int)((i / Q) % C), it = (int)(i / ((long)Q * C)) + 1;
        const f32x4 n = noise_normal4(seed, walker_offset + (uint64_t)c, (uint32_t)it, (uint32_t)q, NOISE_STREAM_DZ);
        float *o = dz + ((size_t)(it - 1) * C + c) * D + 4 * q;
        if (4 * q + 0 < D) o[0] = n.x;
        if (4 * q + 1 < D) o[1] = n.y;
        if (4 * q + 2 < D) o[2] = n.z;
        if (4 * q + 3 < D) o[3] = n.w;
    }
}

template <int U, int LK>
static hipError_t launch_slice_k(const SliceArgs &a, hipStream_t st) {
    const size_t lds = solo_lds_weights<U, 4>() ? (size_t)3 * SOLO4_NF * 64 * sizeof(float) : (size_t)a.s.nets_params() * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(slice_kernel_solo<U, LK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((slice_kernel_solo<U, LK>), dim3((a.C + 3) / 4), dim3(256), lds, st, a);
    return hipGetLastError();
}

bool slice_form_eligible(const FlowShape &s) {
    return s.kind == FLOW_KIND_NVP && s.H == 16 && s.B == 3 && s.L == 1 && s.scale_mode == 0 && s.NT >= 1 && s.NT <= 4;
}

hipError_t launch_slice_solo(const FlowShape &s, const float *packed, const LikeSpec &like, float *z, float *x, double *logl, double loglstar,
                             float width, int steps, int C, int max_out, int max_shrink, uint64_t seed, uint64_t walker_offset,
                             const float *noise_dz, float *hist_x, int *n_call, int *n_move, int *n_eval, hipStream_t st) {
    if (C <= 0) return hipSuccess;
    if (!slice_form_eligible(s)) return hipErrorInvalidConfiguration;
    SliceArgs a;
    memset(&a, 0, sizeof(a));
    a.s = s; a.packed = packed; a.z = z; a.x = x; a.logl = logl; a.loglstar = loglstar; a.width = width; a.steps = steps; a.C = C;
    a.max_out = max_out; a.max_shrink = max_shrink; a.like = like; a.seed = seed; a.walker_offset = walker_offset;
    a.noise_dz = noise_dz; a.hist_x = hist_x; a.n_call = n_call; a.n_move = n_move; a.n_eval = n_eval;
    const bool rosen = like.id == NNEST_LIKE_ROSENBROCK;
    switch (s.NT) {
        case 1: return rosen ? launch_slice_k<1, NNEST_LIKE_ROSENBROCK>(a, st) : launch_slice_k<1, -1>(a, st);
        case 2: return rosen ? launch_slice_k<2, NNEST_LIKE_ROSENBROCK>(a, st) : launch_slice_k<2, -1>(a, st);
        case 3: return rosen ? launch_slice_k<3, NNEST_LIKE_ROSENBROCK>(a, st) : launch_slice_k<3, -1>(a, st);
        case 4: return rosen ? launch_slice_k<4, NNEST_LIKE_ROSENBROCK>(a, st) : launch_slice_k<4, -1>(a, st);
    }
    return hipErrorInvalidConfiguration;
}

hipError_t launch_slice_fill_noise(float *dz, int steps, int C, int D, uint64_t seed, uint64_t walker_offset, hipStream_t st) {
    hipLaunchKernelGGL(slice_fill_noise_kernel, dim3(256), dim3(256), 0, st, dz, steps, C, D, seed, walker_offset);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// walkers (net waves) per workgroup for a population: 4 while one walker per SIMD covers it, then 8 and 12 (two / three walkers
// per SIMD: the SIMD issues one vector instruction per 2 cycles, a lone wave one per 4 -- MI355X_MICROARCH.md).  0 = the
// population does not fit the form (every workgroup must be resident: the batch rule's waits span the grid).
// NNEST_SOLO_WPG in the environment pins the value (diagnostic: tools/time_k4.py).
int solo_walkers_per_group(int C, int num_cu) {
    static const int pinned = [] { const char *e = getenv("NNEST_SOLO_WPG"); return e ? atoi(e) : 0; }();
    for (int wpg = 4; wpg <= 12; wpg += 4) {
        if (pinned && wpg != pinned) continue;
        if ((C + wpg - 1) / wpg + 1 <= num_cu) return wpg;   // + the workgroup that publishes the batch totals
    }
    return 0;
}

bool solo_form_eligible(const MhArgs &a, int num_cu) {
    const FlowShape &s = a.s;
    // 132 weight registers at NT = 2; NT = 3, 4 (x_dim 65..128) and more than one walker per SIMD keep the weights in LDS
    if (s.H != 16 || s.B != 3 || s.L != 1 || s.scale_mode != 0 || s.NT < 1 || s.NT > 4) return false;
    if (a.flags & NNEST_MH_DYNAMIC_STEP) return false;  // the per-16-walker rule belongs to the 16-walker forms
    if ((a.flags & NNEST_MH_DYNAMIC_BATCH) && (mh_flag_lag(a.flags) == 1 || mh_flag_lag(a.flags) == 2)) return false;  // the relay posts step k - 2 in iteration k and asks for step k - lag before that: lag 1, 2 run the quad form (lag 0: every step exact)
    const int wpg = solo_walkers_per_group(a.C, num_cu);
    if (wpg != 4 && (a.noise_dz || a.hist_x || a.hist_logl)) return false;   // history / recorded noise: the one-walker-per-SIMD build only
    return wpg != 0;
}

template <int U, bool DBG, int LK, int WPG>
static hipError_t launch_solo_k(const MhArgs &a, hipStream_t st) {
    const int batch = (a.flags & NNEST_MH_DYNAMIC_BATCH) ? 1 : 0;
    const int grid = (a.C + WPG - 1) / WPG + batch;  // + the workgroup that publishes the batch-wide counts
    const size_t lds = solo_lds_weights<U, WPG>() ? (size_t)3 * SOLO4_NF * 64 * sizeof(float) : (size_t)a.s.nets_params() * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mh_kernel_solo<U, DBG, LK, WPG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((mh_kernel_solo<U, DBG, LK, WPG>), dim3(grid), dim3(64 * (WPG + 2)), lds, st, a);
    return hipGetLastError();
}
template <int U, bool DBG, int LK>
static hipError_t launch_solo_w(const MhArgs &a, int wpg, hipStream_t st) {
    if (wpg == 4) return launch_solo_k<U, DBG, LK, 4>(a, st);
    if constexpr (!DBG) {   // (history / recorded noise: the one-walker-per-SIMD build only)
        if (wpg == 8) return launch_solo_k<U, DBG, LK, 8>(a, st);
        if (wpg == 12) return launch_solo_k<U, DBG, LK, 12>(a, st);
    }
    return hipErrorInvalidConfiguration;
}

hipError_t launch_mh_solo(const MhArgs &a_in, int num_cu, hipStream_t st) {
    MhArgs a = a_in;
    static const bool window_votes = [] { const char *e = getenv("NNEST_SOLO_VOTE"); return e && !strcmp(e, "window"); }();
    if (window_votes) a.flags |= NNEST_MH_WINDOW_VOTES_ONLY;
    const bool dbg = a.noise_dz || a.hist_x || a.hist_logl;
    const int wpg = solo_walkers_per_group(a.C, num_cu);
    // the production kernel of the BASELINE likelihood (Rosenbrock) is compiled with the likelihood fixed: the step loop then
    // carries no other likelihood's code (the float64 paths cost 14 spilled registers and a chain of scalar branches per step)
    if (!dbg && a.like.id == NNEST_LIKE_ROSENBROCK) {
        if (a.s.NT == 1) return launch_solo_w<1, false, NNEST_LIKE_ROSENBROCK>(a, wpg, st);
        if (a.s.NT == 2) return launch_solo_w<2, false, NNEST_LIKE_ROSENBROCK>(a, wpg, st);
        if (a.s.NT == 3) return launch_solo_w<3, false, NNEST_LIKE_ROSENBROCK>(a, wpg, st);
        if (a.s.NT == 4) return launch_solo_w<4, false, NNEST_LIKE_ROSENBROCK>(a, wpg, st);
    }
    switch (a.s.NT) {
        case 1: return dbg ? launch_solo_w<1, true, -1>(a, wpg, st) : launch_solo_w<1, false, -1>(a, wpg, st);
        case 2: return dbg ? launch_solo_w<2, true, -1>(a, wpg, st) : launch_solo_w<2, false, -1>(a, wpg, st);
        case 3: return dbg ? launch_solo_w<3, true, -1>(a, wpg, st) : launch_solo_w<3, false, -1>(a, wpg, st);
        case 4: return dbg ? launch_solo_w<4, true, -1>(a, wpg, st) : launch_solo_w<4, false, -1>(a, wpg, st);
    }
    return hipErrorInvalidConfiguration;
}

}  // namespace nnest
