// nnest_quad.hip -- K4, "quad" form: the persistent constrained-Metropolis kernel (Sampler._mcmc_sample, sampler.py:229-463)
// for populations of at most 4 walkers per compute unit -- BASELINE config 2 (1000 walkers) is 250 tiles on 256 CUs.
//
// Why.  At 1000 walkers the 16-walker tile of v_mfma_f32_16x16x4_f32 gives 63 tiles: 193 of 256 CUs idle, and each
// busy wave drags 16 registers of state per lane through every elementwise stage of a strictly serial step.  Here a
// wave owns FOUR walkers; the layers run on v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per
// instruction, 8 cycles): the 16 blocks are 4 output-feature groups x 4 K-slices, so a 16-wide layer takes K/4
// instructions, and every elementwise stage (activation, affine update, likelihood, select) works on a quarter of the
// registers.  Four times the workgroups, a step about half as long (tools/mfma4x4_probe.hip has the primitive costs).
//
// Lane layout.  lane = 16 ks + 4 fg + j:  j = walker of the tile (0..3), fg = output-feature group, ks = K-slice.
// "Position" m = 4 fg + ks (0..15).  A walker's vector lives at its 16 positions: position m holds dims
// [2U m, 2U m + 2U) -- for parity class c and u < U, register xs[c][u] = dim 2U m + 2u + c (U = 16-slot groups per class
// = FlowShape::NT) -- i.e. each lane owns 2U consecutive floats of its walker's row.
// MFMA: block b = lane >> 2 = (ks, fg); A operand of lane (ks, fg, i) = W[4 fg + i][k], B operand of lane (ks, fg, j) =
// In[k][walker j], D register i of lane (ks, fg, j) = partial of output row 4 fg + i over this block's k.  At rotation
// t (DPP row_ror:4t: lane (ks, fg, j) receives from (ks, fg - t, j)) block (ks, fg) consumes the input held at position
// 4 ((fg - t) & 3) + ks, so over t = 0..3 and ks = 0..3 every input is consumed once per feature group.  The four
// K-slices of an output row are then summed across the four 16-lane rows (two v_permlane32_swap + one
// v_permlane16_swap: a reduce-scatter that leaves row ks with register ks), which puts output unit 4 fg + ks = m at
// position m: a layer's output is laid out as the next layer's input, one value per lane.
//
// One workgroup = two net waves (scale net / translate net of every coupling block; they exchange their outputs through LDS
// at one barrier per block, apply the same affine update and take the same decisions; wave 0 writes) + a noise wave
// (proposal noise for the next step, written to LDS; the same xoshiro streams as the other forms, so nnest_mh_fill_noise
// replays them); one more workgroup behind the tiles publishes the batch-wide accept counts.  QUAD1 keeps both nets on one
// wave (same bits; A/B diagnostic).  Weights are gathered once per launch from the packed (state_dict-order) vector
// through an LDS copy.
// Summation order differs from the 16-walker forms (K split in four), so results agree with them to rounding, not
// bitwise; the form is chosen by population, and callers that need shard-invariance pin it (flags bits 16..19).
#include "mh_common.h"
#include "nnest_internal.h"

namespace nnest {

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() also drains every outstanding global-memory
// operation of the wave (s_waitcnt vmcnt(0)): in the step loop that would expose, at every barrier, the full memory-side
// latency of the batch-rule counters (a load requested a step ahead, an atomic posted a step behind) -- 0.3 us per step, measured.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ f32x4 mfma1(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

template <int N4>  // lane (ks, fg, j) <- lane (ks, fg - N4/4, j)
__device__ __forceinline__ float row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N4, 0xf, 0xf, true));
}

// reduce-scatter of a 4-register accumulator over the four 16-lane rows: row k ends with register k summed over rows
__device__ __forceinline__ float reduce_rows(f32x4 p) {
    auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p.x), __float_as_uint(p.z), false, false);
    auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p.y), __float_as_uint(p.w), false, false);
    const float a = __uint_as_float(s0[0]) + __uint_as_float(s0[1]);  // rows 0,1: reg 0 over (r, r+2); rows 2,3: reg 2
    const float b = __uint_as_float(s1[0]) + __uint_as_float(s1[1]);  // rows 0,1: reg 1;               rows 2,3: reg 3
    auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

// sum of v over the 16 positions of a walker, bit-identical in all 16 lanes: (fg ^ 2) first -- an involution, so both
// partners add the same pair -- then fg - 1 (the two values it can meet are already pairwise equal), then the rows.
// (v_permlane*_swap with both operands the same value: hipcc 7.2 folds the two results, hence the asm, DESIGN.md.)
__device__ __forceinline__ float walker_sum(float v) {
    v = v + row_ror<8>(v);
    v = v + row_ror<4>(v);
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = a + b;
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

template <int U>
struct QuadNet {  // one (block, net), num_layers = 1, hidden 16
    float w1[U][4], w2[4], w3[U][4];
    float b1, b2, b3[U];
};

// gather this lane's fragments of one (block, net) from its packed region (LDS copy), state_dict layout:
// W0[H][D] b0[H] W1[H][H] b1[H] Wo[D][H] bo[D]   (nnest/networks.py:271-282)
template <int U>
__device__ __forceinline__ void quad_gather(QuadNet<U> &n, const float *p, int D, int cc, int ct, int lane) {
    const int H = 16;
    const int i = lane & 3, fg = (lane >> 2) & 3, ks = lane >> 4, m = 4 * fg + ks;
    const int pb0 = H * D, pW1 = pb0 + H, pb1 = pW1 + H * H, pWo = pb1 + H, pbo = pWo + D * H;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int mp = 4 * ((fg - t) & 3) + ks;  // position whose value this block consumes at rotation t
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int d = 2 * U * mp + 2 * u + cc;
            n.w1[u][t] = d < D ? p[(4 * fg + i) * D + d] : 0.f;
            const int dO = 2 * U * (4 * fg + i) + 2 * u + ct;  // dim of transformed slot (u, position 4 fg + i)
            n.w3[u][t] = dO < D ? p[pWo + dO * H + mp] : 0.f;
        }
        n.w2[t] = p[pW1 + (4 * fg + i) * H + mp];
    }
    n.b1 = p[pb0 + m];
    n.b2 = p[pb1 + m];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int dm = 2 * U * m + 2 * u + ct;
        n.b3[u] = dm < D ? p[pbo + dm] : 0.f;
    }
}

// CouplingLayer.inverse (networks.py:300-309) on a quad tile, both nets interleaved; returns the lane's log-det partial
template <int U>
__device__ __forceinline__ float quad_coupling_inverse(const QuadNet<U> &ns, const QuadNet<U> &nt, const float (&cond)[U],
                                                       float (&trans)[U]) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ps = zero4, pt = zero4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float c0 = cond[u], c1 = row_ror<4>(cond[u]), c2 = row_ror<8>(cond[u]), c3 = row_ror<12>(cond[u]);
        ps = mfma1(ns.w1[u][0], c0, ps); pt = mfma1(nt.w1[u][0], c0, pt);
        ps = mfma1(ns.w1[u][1], c1, ps); pt = mfma1(nt.w1[u][1], c1, pt);
        ps = mfma1(ns.w1[u][2], c2, ps); pt = mfma1(nt.w1[u][2], c2, pt);
        ps = mfma1(ns.w1[u][3], c3, ps); pt = mfma1(nt.w1[u][3], c3, pt);
    }
    float hs = fast_tanh(reduce_rows(ps) + ns.b1);
    float ht = fmaxf(reduce_rows(pt) + nt.b1, 0.f);
    {
        ps = zero4; pt = zero4;
        const float s1 = row_ror<4>(hs), s2 = row_ror<8>(hs), s3 = row_ror<12>(hs);
        const float t1 = row_ror<4>(ht), t2 = row_ror<8>(ht), t3 = row_ror<12>(ht);
        ps = mfma1(ns.w2[0], hs, ps); pt = mfma1(nt.w2[0], ht, pt);
        ps = mfma1(ns.w2[1], s1, ps); pt = mfma1(nt.w2[1], t1, pt);
        ps = mfma1(ns.w2[2], s2, ps); pt = mfma1(nt.w2[2], t2, pt);
        ps = mfma1(ns.w2[3], s3, ps); pt = mfma1(nt.w2[3], t3, pt);
        hs = fast_tanh(reduce_rows(ps) + ns.b2);
        ht = fmaxf(reduce_rows(pt) + nt.b2, 0.f);
    }
    const float s1 = row_ror<4>(hs), s2 = row_ror<8>(hs), s3 = row_ror<12>(hs);
    const float t1 = row_ror<4>(ht), t2 = row_ror<8>(ht), t3 = row_ror<12>(ht);
    float ld = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        f32x4 qs = zero4, qt = zero4;
        qs = mfma1(ns.w3[u][0], hs, qs); qt = mfma1(nt.w3[u][0], ht, qt);
        qs = mfma1(ns.w3[u][1], s1, qs); qt = mfma1(nt.w3[u][1], t1, qt);
        qs = mfma1(ns.w3[u][2], s2, qs); qt = mfma1(nt.w3[u][2], t2, qt);
        qs = mfma1(ns.w3[u][3], s3, qs); qt = mfma1(nt.w3[u][3], t3, qt);
        const float ls = reduce_rows(qs) + ns.b3[u];
        const float tt = reduce_rows(qt) + nt.b3[u];
        trans[u] = (trans[u] - tt) * __expf(-ls);  // (inputs - t) * exp(-log_s)   networks.py:307-309
        ld -= ls;
    }
    return ld;
}

// ONE net of a coupling block (the team form: scale and translate nets on different waves): the same chains of the same
// operations as the interleaved form above, so both forms produce the same bits
template <int U, int ACT>
__device__ __forceinline__ void quad_net(const QuadNet<U> &n, const float (&cond)[U], float (&out)[U]) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 p = zero4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float c0 = cond[u], c1 = row_ror<4>(cond[u]), c2 = row_ror<8>(cond[u]), c3 = row_ror<12>(cond[u]);
        p = mfma1(n.w1[u][0], c0, p);
        p = mfma1(n.w1[u][1], c1, p);
        p = mfma1(n.w1[u][2], c2, p);
        p = mfma1(n.w1[u][3], c3, p);
    }
    float h = reduce_rows(p) + n.b1;
    h = ACT == 0 ? fast_tanh(h) : fmaxf(h, 0.f);
    {
        p = zero4;
        const float h1 = row_ror<4>(h), h2 = row_ror<8>(h), h3 = row_ror<12>(h);
        p = mfma1(n.w2[0], h, p);
        p = mfma1(n.w2[1], h1, p);
        p = mfma1(n.w2[2], h2, p);
        p = mfma1(n.w2[3], h3, p);
        h = reduce_rows(p) + n.b2;
        h = ACT == 0 ? fast_tanh(h) : fmaxf(h, 0.f);
    }
    const float h1 = row_ror<4>(h), h2 = row_ror<8>(h), h3 = row_ror<12>(h);
    f32x4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = mfma1(n.w3[u][0], h, zero4);
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = mfma1(n.w3[u][1], h1, q[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = mfma1(n.w3[u][2], h2, q[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = mfma1(n.w3[u][3], h3, q[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) out[u] = reduce_rows(q[u]) + n.b3[u];
}

// NormalizingFlow.inverse (networks.py:34-42), num_blocks = 3: blocks 2, 1, 0; block b conditions on class (b+1)&1 and
// transforms class b&1.  TEAM: this wave holds ONE net (role 0 scale, 1 translate) of every block; the two waves of a
// tile publish their net's outputs in LDS, meet at one barrier per block and apply the same affine update, so both hold
// bit-identical state.
template <int U, bool TEAM>
struct QuadFlow {
    QuadNet<U> n[3][TEAM ? 1 : 2];
    float *xch;  // TEAM: LDS [3 blocks][2 nets][U][64]
    int role, lane;

    template <int BLK>
    __device__ __forceinline__ float team_block(const float (&cond)[U], float (&trans)[U]) const {
        float mine[U], other[U];
        if (role == 0) quad_net<U, 0>(n[BLK][0], cond, mine);
        else           quad_net<U, 1>(n[BLK][0], cond, mine);
        float *slot = xch + BLK * 2 * U * 64;  // one slot per block: reused only three barriers later
#pragma unroll
        for (int u = 0; u < U; ++u) slot[(role * U + u) * 64 + lane] = mine[u];
        lds_barrier();
#pragma unroll
        for (int u = 0; u < U; ++u) other[u] = slot[((1 - role) * U + u) * 64 + lane];
        float ld = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float ls = role == 0 ? mine[u] : other[u], tt = role == 0 ? other[u] : mine[u];
            trans[u] = (trans[u] - tt) * __expf(-ls);
            ld -= ls;
        }
        return ld;
    }

    __device__ __forceinline__ float inverse(float (&xs)[2][U]) const {
        if constexpr (TEAM) {
            float ld = team_block<2>(xs[1], xs[0]);
            ld += team_block<1>(xs[0], xs[1]);
            ld += team_block<0>(xs[1], xs[0]);
            return ld;
        } else {
            float ld = quad_coupling_inverse<U>(n[2][0], n[2][1], xs[1], xs[0]);
            ld += quad_coupling_inverse<U>(n[1][0], n[1][1], xs[0], xs[1]);
            ld += quad_coupling_inverse<U>(n[0][0], n[0][1], xs[1], xs[0]);
            return ld;
        }
    }
};

// ---- likelihoods on a quad tile (same arithmetic per term as loglike_tile, flow_tile.h; sums over the 16 positions) ----
#pragma clang fp contract(off)
template <int U>
__device__ __forceinline__ double quad_loglike(const LikeSpec &lk, int D, int lane, int nxt_lane, const float (&xs)[2][U]) {
    const int fg = (lane >> 2) & 3, ks = lane >> 4, m = 4 * fg + ks, j = lane & 3;
    const float scale = lk.scale;
    float th[2 * U + 1];
#pragma unroll
    for (int u = 0; u < U; ++u) { th[2 * u] = scale * xs[0][u]; th[2 * u + 1] = scale * xs[1][u]; }
    double acc;
    if (lk.id == 0) {
        // Rosenbrock (likelihoods.py:51): -sum_i 100 (x[i+1] - x[i]^2)^2 + (1 - x[i])^2, i = 0..D-2
        th[2 * U] = __shfl(th[0], nxt_lane);  // first dim of position m + 1
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
        acc = -(double)walker_sum(facc);
    } else if (lk.id == 1) {
        // GaussianMix (likelihoods.py:165-189): logsumexp_k[ log w_k - |theta - mu_k|^2/2 - (D/2) log 2pi ]
        float facc = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * U; ++k) {
            const int d = 2 * U * m + k;
            float sq = th[k] * th[k];
            facc = facc + ((d >= 2 && d < D) ? sq : 0.f);
        }
        const double base = (double)walker_sum(facc);
        const float t0 = __shfl(th[0], j), t1 = __shfl(th[1], j);  // theta[0], theta[1]: position 0 = lanes 0..3
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
        acc = (double)walker_sum(facc);
    } else if (lk.id == 4) {
        // Eggbox (likelihoods.py:104-106), x_dim = 2
        const float t0 = __shfl(th[0], j), t1 = __shfl(th[1], j);
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
        // walker totals, identical in the 16 lanes: xor butterflies over the lane bits 2..5
#pragma unroll
        for (int o = 4; o <= 32; o <<= 1) {
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

// UniformPrior(D,-1,1) (priors.py:39-43) for the 4 walkers of the tile: 1 if all 16 positions are inside
template <int U>
__device__ __forceinline__ int quad_inbox(const float (&xs)[2][U], unsigned long long walker_lanes) {
    int ok = 1;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int u = 0; u < U; ++u) ok &= !(xs[c][u] < -1.f || xs[c][u] > 1.f);
    const unsigned long long out = ~__ballot(ok != 0);
    return (out & walker_lanes) == 0ull;
}

static constexpr int QUAD_ETAB = 1024;  // steps + 2 <= QUAD_ETAB: exp(1 / (1 + k)) from a table

template <int U, bool DBG, bool TEAM>
__global__ void __launch_bounds__(TEAM ? 192 : 128, 1) mh_kernel_quad(MhArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wlds[];  // the packed weights
    __shared__ __attribute__((aligned(16))) float nbuf[2][64][2 * U];
    __shared__ float ubuf[2][4];
    __shared__ int acc_lds[2], res_lds[2];  // batch rule relayed by the noise wave: this tile's accepted count / the batch total
    __shared__ float xch[TEAM ? 3 * 2 * U * 64 : 1];
    __shared__ double etab[QUAD_ETAB];
    constexpr int NOISE_WAVE = TEAM ? 2 : 1, NBAR = TEAM ? 3 : 0;  // NBAR: workgroup barriers inside one flow inverse

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;
    const int D = a.s.D, S = a.steps, C = a.C;
    const bool recorded = DBG && a.noise_dz;
    const bool dynamic = (a.flags & (NNEST_MH_DYNAMIC_STEP | NNEST_MH_DYNAMIC_BATCH)) != 0;
    const bool use_tab = S + 2 <= QUAD_ETAB;
    const bool batch_rule = dynamic && (a.flags & NNEST_MH_DYNAMIC_BATCH) != 0;
    const int lag_rule = mh_flag_lag(a.flags);
    // lag >= 2: the noise wave -- which has slack -- does all the global-memory work of the batch rule and hands the totals to
    // the net waves through LDS at the per-step barrier: a step of the net waves then contains no global-memory operation.
    // (Posting from a net wave cost ~1000 cycles per step: the compiler has to drain the atomic before it may reuse its
    // operand registers, and an atomic takes 600-3000 cycles to retire.)  lag < 2 leaves no time for the relay: direct path.
    const bool relay = batch_rule && lag_rule >= 2;
    const int ntiles = (C + 3) >> 2;
    if (tile >= ntiles) {
        // batch-wide step rule (mh_common.h): the one workgroup behind the tiles sums every step's counters as soon as they
        // are complete and publishes the total; its other waves leave at once
        if (batch_rule && wave == 0) mh_sync_publisher(a.sync, S, S - lag_rule, ntiles, C, lane, a.sync_err);
        return;
    }
    {
        const int n = a.s.nets_params();
        for (int i = threadIdx.x; i < n; i += blockDim.x) wlds[i] = a.packed[i];
        if (dynamic && use_tab)
            for (int k = threadIdx.x; k < S + 2; k += blockDim.x) etab[k] = exp(1.0 / (double)(1 + k));
    }
    __syncthreads();

    if (wave == NOISE_WAVE) {
        // proposal noise: the streams of the 16-walker forms -- per (walker, lane group g) 8 normals for dims
        // 32 t + 8 g + [0, 8), t < U -- generated by 16 lanes (4 walkers x 4 groups) and written where the net waves read them.
        // TEAM: this wave joins the NBAR barriers of every flow inverse the net waves run (the initial one, then one per step).
        const int j = lane & 3, g = (lane >> 2) & 3;
        const bool gen = lane < 16;
        const int row = tile * 4 + j;
        XoshiroNoise<U> rng;
        rng.init(a.seed, a.walker_offset + (uint64_t)row, g, D);
        for (int k = 0; k <= S; ++k) {
            // between the barriers k - 1 and k the net waves run step k - 1: the count of step k - 2 is in LDS, and the total
            // they will apply at the end of step k (that of step k - lag) has to be in LDS by barrier k
            unsigned long long early = 0;
            const int want = k - lag_rule;
            if (relay) {
                if (k >= 2 && lane == 0) mh_sync_post(a.sync, k - 2, tile, acc_lds[k & 1]);
                if (want >= 1) early = mh_result_load(a.sync, S, want, tile);
            }
            if (gen && !recorded) {
                float nz[U][8], u;
                rng.next(nz, u);
#pragma unroll
                for (int t = 0; t < U; ++t)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int d = 32 * t + 8 * g + q;        // padded dims (d >= D) carry 0
                        const int m = d / (2 * U), kk = d % (2 * U);
                        const int dst = 16 * (m & 3) + 4 * (m >> 2) + j;
                        nbuf[k & 1][dst][kk] = nz[t][q];
                    }
                if (g == 0) ubuf[k & 1][j] = u;
            }
            if (relay && want >= 1) {
                const int total = mh_result_wait(a.sync, S, want, tile, early, a.sync_err);
                if (lane == 0) res_lds[k & 1] = total;
            }
            if (k == 0)
                for (int b = 0; b < NBAR; ++b) lds_barrier();  // the net waves' initial inverse
            lds_barrier();                                     // publish buffer k
            if (k >= 1)
                for (int b = 0; b < NBAR; ++b) lds_barrier();  // step k's inverse
        }
        return;
    }

    // ---- the walkers: wave 0 (TEAM: waves 0 and 1, one net each, the same state and the same decisions; wave 0 writes) ----
    const bool writer = wave == 0;
    const int j = lane & 3, fg = (lane >> 2) & 3, ks = lane >> 4, m = 4 * fg + ks;
    const int row = tile * 4 + j;
    const bool ok = row < C;
    const int nvalid = min(4, C - tile * 4);
    (void)ntiles;
    const LikeSpec like = a.like;
    const double loglstar = a.loglstar;
    const bool free_mode = (a.flags & NNEST_MH_UNCONSTRAINED) != 0;
    const bool batch = batch_rule;
    const int lag = lag_rule;
    const unsigned long long walker_lanes = 0x1111111111111111ull << j;
    const int mn = m + 1;  // position holding the next dims (Rosenbrock couples dim 2U m + 2U - 1 with 2U (m + 1))
    const int nxt_lane = mn < 16 ? 16 * (mn & 3) + 4 * (mn >> 2) + j : lane;

    QuadFlow<U, TEAM> flow;
    flow.xch = xch;
    flow.role = wave;
    flow.lane = lane;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        if constexpr (TEAM) {
            quad_gather<U>(flow.n[b][0], wlds + (size_t)(b * 2 + wave) * a.s.net_params, D, (b + 1) & 1, b & 1, lane);
        } else {
#pragma unroll
            for (int n = 0; n < 2; ++n)
                quad_gather<U>(flow.n[b][n], wlds + (size_t)(b * 2 + n) * a.s.net_params, D, (b + 1) & 1, b & 1, lane);
        }
    }

    float z[2][U], x[2][U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int d = 2 * U * m + 2 * u + c;
            z[c][u] = (ok && d < D) ? a.z[(size_t)row * D + d] : 0.f;
            x[c][u] = z[c][u];
        }
    float ld = walker_sum(flow.inverse(x));  // x = f^-1(z), log_det_J  (sampler.py:266, :295)
    double logl = ok ? a.logl[row] : 0.0;
    double scale = (double)a.step_size;
    int accept = 0, reject = 0, n_acc = 0, n_call = 0;

    auto store_row = [&](float *base, size_t r, const float (&v)[2][U]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int d = 2 * U * m + 2 * u + c;
                if (d < D) base[r * D + d] = v[c][u];
            }
    };
    // the chain's first x stays in the x output buffer for the launch: the reference counts a chain only if EVERY coordinate of its
    // last x differs from its first (nested.py:432), tested at the end (no register is held for it)
    if (writer && ok && a.x) store_row(a.x, (size_t)row, x);
    if (DBG && writer) {
        if (a.hist_x && ok) store_row(a.hist_x, (size_t)row * (S + 1), x);
        if (a.hist_logl && ok && m == 0) a.hist_logl[(size_t)row * (S + 1)] = logl;
    }

    float nz[2 * U], u_next = 0.f;
    int kbuf = 0, relayed_total = 0;
    auto fetch_noise = [&]() {
        lds_barrier();  // the noise wave has published buffer kbuf (and, relayed, the batch total to apply in step kbuf)
        if (!recorded) {
#pragma unroll
            for (int k = 0; k < 2 * U; ++k) nz[k] = nbuf[kbuf & 1][lane][k];
            u_next = ubuf[kbuf & 1][j];
        }
        if (relay) relayed_total = res_lds[kbuf & 1];
        ++kbuf;
    };
    fetch_noise();

    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, a_prop = 0, a_inv = 0, a_post = 0, a_tot = 0;
    unsigned long long sp0 = 0, sp1 = 0, sp2 = 0, a_p0 = 0, a_p1 = 0, a_p2 = 0, a_p3 = 0;
    (void)st0; (void)st1; (void)st2; (void)st3; (void)a_prop; (void)a_inv; (void)a_post; (void)a_tot;
    (void)sp0; (void)sp1; (void)sp2; (void)a_p0; (void)a_p1; (void)a_p2; (void)a_p3;
    for (int it = 1; it <= S; ++it) {
        STAMP(st0);
        // batch-wide rule: request the counts of step it - lag now, consume them at the end of the step
        unsigned long long early = 0;
        const bool have_total = batch && dynamic && it - lag >= 1;
        if (have_total && lag > 0 && !relay) early = mh_result_load(a.sync, S, it - lag, tile);
        // proposal z' = z + randn * scale  (sampler.py:310, :316); float32 like torch
        const float fs = (float)scale;
        float zp[2][U], xp[2][U];
        float u;
        if (recorded) {
#pragma unroll
            for (int uu = 0; uu < U; ++uu)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int d = 2 * U * m + 2 * uu + c;
                    const float dz = (ok && d < D) ? a.noise_dz[((size_t)(it - 1) * C + row) * D + d] : 0.f;
                    zp[c][uu] = z[c][uu] + dz * fs;
                }
            u = ok ? a.noise_u[(size_t)(it - 1) * C + row] : 1.f;
            fetch_noise();
        } else {
#pragma unroll
            for (int uu = 0; uu < U; ++uu) {
                zp[0][uu] = z[0][uu] + nz[2 * uu] * fs;
                zp[1][uu] = z[1][uu] + nz[2 * uu + 1] * fs;
            }
            u = u_next;
            fetch_noise();
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int uu = 0; uu < U; ++uu) xp[c][uu] = zp[c][uu];
        STAMP(st1);
        const float ldp = walker_sum(flow.inverse(xp));  // sampler.py:321
        STAMP(st2);

        // log_ratio = log_det_J' - log_det_J, -inf outside the prior box  (sampler.py:326-331)
        const int inb = quad_inbox<U>(xp, walker_lanes);
        float log_ratio = inb ? (ldp - ld) : -INFINITY;
        float ratio = fminf(__expf(log_ratio), 1.0f);  // exp().clamp(max=1)  :335
        if (log_ratio != log_ratio) ratio = log_ratio;  // NaN stays NaN (u < NaN is false, as in torch)
        const bool pre = ok && (u < ratio);             // :336
        const double lp = quad_loglike<U>(like, D, lane, nxt_lane, xp);
        bool acc = pre && (lp > loglstar);  // :361
        if (free_mode) {  // sampler.py:396-410
            const double lr = inb ? (double)(ldp - ld) + (lp - logl) : -INFINITY;
            const double rt = fmin(exp(lr), 1.0);
            acc = ok && ((double)u < rt);
        }
        n_call += (free_mode ? ok : pre) ? 1 : 0;
        n_acc += acc ? 1 : 0;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int uu = 0; uu < U; ++uu) {
                z[c][uu] = acc ? zp[c][uu] : z[c][uu];
                x[c][uu] = acc ? xp[c][uu] : x[c][uu];
            }
        ld = acc ? ldp : ld;
        logl = acc ? lp : logl;
        STAMP(sp0);
        if (dynamic) {  // sampler.py:422-431
            const int tile_accepted = __popcll(__ballot(acc && m == 0));
            int num_accepted = tile_accepted, num_total = nvalid;
            bool apply = true;
            if (batch) {
                apply = have_total;
                if (relay) {
                    num_accepted = relayed_total;                                 // fetched at this step's barrier
                    if (writer && lane == 0) acc_lds[it & 1] = tile_accepted;     // posted by the noise wave after the next one
                } else {
                    // lag >= 1: consume the total requested at the top of the step BEFORE posting this step's count (memory
                    // operations retire in order)
                    if (apply && lag > 0) num_accepted = mh_result_wait(a.sync, S, it - lag, tile, early, a.sync_err);
                    STAMP(sp1);
                    if (writer && lane == 0) mh_sync_post(a.sync, it, tile, tile_accepted);
                    STAMP(sp2);
                    if (apply && lag == 0) num_accepted = mh_result_wait(a.sync, S, it, tile, mh_result_load(a.sync, S, it, tile), a.sync_err);
                }
                num_total = C;
            }
            if (apply) {
                if (2 * num_accepted > num_total) accept += 1; else reject += 1;
                if (accept > reject) scale *= use_tab ? etab[accept] : exp(1.0 / (1 + accept));
                if (accept < reject) scale /= use_tab ? etab[reject] : exp(1.0 / (1 + reject));
            }
        }
        STAMP(st3);
#ifdef NNEST_STAMP
        a_prop += st1 - st0; a_inv += st2 - st1; a_post += st3 - st2; a_tot += st3 - st0;
        a_p0 += sp0 - st2; if (sp1 > sp0) { a_p1 += sp1 - sp0; a_p2 += sp2 - sp1; a_p3 += st3 - sp2; }
#endif
        if (DBG && writer) {
            if (a.hist_x && ok) store_row(a.hist_x, (size_t)row * (S + 1) + it, x);
            if (a.hist_logl && ok && m == 0) a.hist_logl[(size_t)row * (S + 1) + it] = logl;
        }
    }
    if (!writer) return;
#ifdef NNEST_STAMP
    if (a.scale_out && lane == 0 && tile == 0) {
        float *o = a.scale_out;
        o[0] = (float)a_tot; o[1] = (float)a_prop; o[2] = (float)a_inv; o[3] = (float)a_post;
        o[4] = (float)a_p0; o[5] = (float)a_p1; o[6] = (float)a_p2; o[7] = (float)a_p3;
    }
#endif
    bool all_moved = n_acc > 0;   // (no x buffer: the accept count stands in)
    if (a.x) {
        bool mine = true;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int d = 2 * U * m + 2 * u + c;
                const float x0 = (ok && d < D) ? a.x[(size_t)row * D + d] : 0.f;
                mine = mine && (d >= D || x[c][u] != x0);
            }
        all_moved = (__ballot(mine) & walker_lanes) == walker_lanes;   // every position of walker j
    }
    if (ok) {
        store_row(a.z, (size_t)row, z);
        if (a.x) store_row(a.x, (size_t)row, x);
        if (m == 0) {
            a.logl[row] = logl;
            if (a.n_accept) a.n_accept[row] = n_acc | (all_moved ? NNEST_MH_ALL_MOVED : 0);
            if (a.n_call) a.n_call[row] = n_call;
        }
    }
#ifndef NNEST_STAMP
    // scale_out has one entry per 16 walkers (nnest_mh_num_groups): the first tile of each group reports
    if (a.scale_out && lane == 0 && (tile & 3) == 0) a.scale_out[tile >> 2] = (float)scale;
#endif
}

// ------------------------------------------------------------------------------------------------
bool quad_form_eligible(const MhArgs &a, int num_cu) {
    const FlowShape &s = a.s;
    if (s.H != 16 || s.B != 3 || s.L != 1 || s.scale_mode != 0 || s.NT < 1 || s.NT > 4) return false;
    if ((a.flags & NNEST_MH_DYNAMIC_STEP) && !(a.flags & NNEST_MH_DYNAMIC_BATCH)) return false;  // the per-16-walker rule belongs to the 16-walker forms
    const int ntiles = (a.C + 3) / 4;
    // one tile per CU; two where two workgroups fit a CU's LDS (packed weights + ~14 KB each) -- both resident, which the
    // batch-wide step rule needs (244 VGPRs: two waves per SIMD)
    const size_t lds = (size_t)s.nets_params() * sizeof(float) + 16 * 1024;
    const int per_cu = 2 * lds <= 160 * 1024 ? 2 : 1;
    return ntiles + 1 <= per_cu * num_cu;
}

template <int U, bool DBG, bool TEAM>
static hipError_t launch_quad_k(const MhArgs &a, hipStream_t st) {
    const int batch = (a.flags & NNEST_MH_DYNAMIC_BATCH) ? 1 : 0;
    const int ntiles = (a.C + 3) / 4 + batch;  // + the workgroup that publishes the batch-wide counts
    const size_t lds = (size_t)a.s.nets_params() * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mh_kernel_quad<U, DBG, TEAM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((mh_kernel_quad<U, DBG, TEAM>), dim3(ntiles), dim3(TEAM ? 192 : 128), lds, st, a);
    return hipGetLastError();
}

template <int U>
static hipError_t launch_quad_u(const MhArgs &a, int num_cu, hipStream_t st) {
    const bool dbg = a.noise_dz || a.hist_x || a.hist_logl;
    // one tile per CU: scale / translate nets on two waves (+ the noise wave).  Two tiles per CU: both nets on one wave, so that
    // the four waves of a CU's two tiles have a SIMD each (measured at 2000 walkers: 0.60 ms against 0.68 with three waves
    // per tile and 0.76 for the 16-walker team form).  The two schedules produce the same bits.
    const bool team = mh_flag_form(a.flags) != MH_FORM_QUAD1 && (a.C + 3) / 4 <= num_cu;
    if (team) return dbg ? launch_quad_k<U, true, true>(a, st) : launch_quad_k<U, false, true>(a, st);
    return dbg ? launch_quad_k<U, true, false>(a, st) : launch_quad_k<U, false, false>(a, st);
}

hipError_t launch_mh_quad(const MhArgs &a, int num_cu, hipStream_t st) {
    switch (a.s.NT) {
        case 1: return launch_quad_u<1>(a, num_cu, st);
        case 2: return launch_quad_u<2>(a, num_cu, st);
        case 3: return launch_quad_u<3>(a, num_cu, st);
        case 4: return launch_quad_u<4>(a, num_cu, st);
    }
    return hipErrorInvalidConfiguration;
}

}  // namespace nnest
