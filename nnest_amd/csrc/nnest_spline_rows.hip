// nnest_spline_rows.hip -- the "rows" form of the spline flow's training step (round 6): ONE ROW OF THE MINIBATCH PER WORKGROUP.
//
// Reference: Trainer._train on SingleSpeedSpline (nnest/trainer.py:384-403; networks.py:393-715): loss = -mean(log_probs(X[perm] +
// jitter * randn)), Adam with coupled weight decay.  The tile form (nnest_spline_train.hip: spl_grad_kernel + spl_update_kernel)
// gives every lane one whole rational-quadratic-spline evaluation per coupling -- ~1100 vector instructions per wave and coupling
// pass, 46 of them quarter-rate transcendentals, on 13 workgroups; a minibatch is a chain of twelve such passes.  Here the
// evaluation itself is spread: item (row, transformed dimension) sits on EIGHT lanes, lane k holding bin k (its width / height
// logits and the derivative logit of its right knot); softmax sums, the knot positions (a prefix sum) and the bin search are
// DPP operations inside the 8-lane group.  A row of x_dim 50 is 25 items = 200 lanes = four waves, a minibatch of 100 rows is
// 100 workgroups on 100 CUs, and a coupling pass is ~350 vector instructions per wave.  Layout of one coupling pass:
//   trunk (3 hidden layers of 16): every wave computes it redundantly on the matrix cores (y = W x with the vector in all sixteen
//       columns of the B operand; SplrTrunkF has the layout), the weights read straight from the PACKED state_dict-order vector (no
//       image) -- a lane's four A operands of a layer are 16 contiguous bytes of the weight row;
//   last layer (16 -> 23 n_out): the same product per sixteen output rows, the tiles dealt over the waves; the raw parameters go
//       through LDS to the
//   evaluation in the item layout above; the transformed values go back to the row vector in LDS.
// The backward pass mirrors it (the forward pass's per-lane intermediates are kept in LDS: 16 floats per lane and coupling).
// Parameter gradients are NOT accumulated here: each row stages the operands (activations, pre-activation gradients, raw-parameter
// gradients, the 1x1 conv's input / output gradient) and splr_update_kernel contracts them over the rows of the minibatch on the
// matrix cores, one 16x16 output tile per wave (K = rows), takes the Adam step in registers and writes the packed vector --
// fixed summation order, no atomics: bitwise repeatable.  The 1x1 conv's LU parameters are stepped by one workgroup per block as in
// the tile form (spl_update_kernel's head), from the contraction instead of 13 tiles' partial sums.
//
// Shapes: hidden_dim 16, num_bins 8, x_dim <= 64, minibatch <= 128 rows (the reference's defaults and BASELINE configs 1-3 with
// flow='spline'); everything else keeps the tile form.  NNEST_SPL_ROWS=0 pins the tile form (A/B, tests of both).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "spline_host.h"
#include "spline_train_tile.h"

using namespace nnest;

namespace nnest {

enum { SPLR_TROW = 160, SPLR_VROW = 64 };  // trunk staging row: h0 h1 h2 d0 d1 d2 (16 each) + u (64); block staging rows

// staging buffer (floats): T[c] = off_T + (c rows_cap + r) 160; G[c] = off_G + (c rows_cap + r) grow; V[b][which] = off_V + ((3b + which)
// rows_cap + r) 64 with which = 0: a (ActNorm output = conv input), 1: g_c (gradient at the conv output), 2: g_a (at the conv input)
struct SplRowsLayout {
    int rows_cap, grow;
    size_t off_T, off_G, off_V, total;
};

struct SplRowsState {
    SplRowsLayout lay;
    float *stg;
    float *wmatT;   // [B][D][D]
    float *ldc;     // [B]: sum(s) + sum(log|S|) of a block (networks.py:650, :676)
    float *rowlp;   // [rows_cap + valid_cap]
    int valid_cap;
};

struct SplRowsArgs {
    const float *w, *wmat, *wmatT, *ldc;
    SplTrainShape ts;
    const float *x;
    const int *perm;
    int M, mtot;
    const float *noise;
    uint64_t seed;
    long noise_row0;
    int epoch;
    float jitter;
    const float *xv;
    int Mv;
    float *rowlp;
    float *stg;
    SplRowsLayout lay;
    const int *stop;
};

// ---- lane primitives ------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float splr_dpp(float v) {   // invalid source lanes read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// over the 8 lanes of an item (lanes 8j .. 8j+7 of a wave): every lane ends with the same bits (each stage adds a pair both partners see)
__device__ __forceinline__ float splr_sum8(float v) {
    v += splr_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
    v += splr_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
    v += splr_dpp<0x141>(v);   // row_half_mirror
    return v;
}
__device__ __forceinline__ float splr_max8(float v) {
    v = fmaxf(v, splr_dpp<0xB1>(v));
    v = fmaxf(v, splr_dpp<0x4E>(v));
    v = fmaxf(v, splr_dpp<0x141>(v));
    return v;
}
// inclusive prefix sum over the 8 lanes of an item (k = lane & 7)
__device__ __forceinline__ float splr_scan8(float v, int k) {
    float t = splr_dpp<0x111>(v);  // row_shr:1
    v += k >= 1 ? t : 0.f;
    t = splr_dpp<0x112>(v);
    v += k >= 2 ? t : 0.f;
    t = splr_dpp<0x114>(v);
    v += k >= 4 ? t : 0.f;
    return v;
}
__device__ __forceinline__ float splr_rl(float v, int lane) {   // lane: wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ void splr_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float splr_lrelu(float v) { return v > 0.f ? v : 0.2f * v; }
__device__ __forceinline__ float splr_slope(float post) { return post > 0.f ? 1.f : 0.2f; }   // (the activation keeps the sign)
__device__ __forceinline__ float splr_comp(const f32x4 &v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// what the backward pass needs again of a lane's share of an evaluation (LDS, 16 floats per lane and coupling)
struct SplrKeep {
    float a_w, p_w, a_h, p_h;      // the two softmax stages of the width / height logits (spl_knots)
    float dd, left, width, chl;    // d(right knot derivative)/d(logit); this bin's left edge, width, bottom edge
    float ht, d0, d1, x;           // height, knot derivatives, the input
    float cnt, use, ribw, theta;   // selected bin (as a float), 1 if this lane IS the selected bin of an input inside the interval; 1 / width, (x - left) / width
};

// forward evaluation, lane k of an item: logits rw, rh (bins) and rd (right knot of bin k; unused for k = 7).  Returns y (all 8
// lanes), adds the log-derivative to `ld` on the selected lane.  networks.py:425-556 through NSF_CL's :583-587, as spl_rqs.
__device__ __forceinline__ float splr_eval(float rw, float rh, float rd, float x, float tail, int k, bool active, float &ld, SplrKeep &kp) {
    const float T2 = 2.f * tail, cmin = 1.f - 1e-3f * SPL_K;
    const bool inside = x >= -tail && x <= tail;
    // widths
    float e = spl_exp(rw - splr_max8(rw));
    const float a_w = e * spl_rcp(splr_sum8(e));
    e = spl_exp(T2 * a_w - T2);                       // (the second softmax: its arguments lie in [0, 2 tail])
    const float p_w = e * spl_rcp(splr_sum8(e));
    const float wk = 1e-3f + cmin * p_w;
    const float cw_in = splr_scan8(wk, k);
    float t = splr_dpp<0x111>(cw_in);
    const float left = k == 0 ? -tail : T2 * t + (-tail);
    const float right = k == SPL_K - 1 ? tail : T2 * cw_in + (-tail);
    const float width = right - left;
    // heights
    e = spl_exp(rh - splr_max8(rh));
    const float a_h = e * spl_rcp(splr_sum8(e));
    e = spl_exp(T2 * a_h - T2);
    const float p_h = e * spl_rcp(splr_sum8(e));
    const float hk = 1e-3f + cmin * p_h;
    const float ch_in = splr_scan8(hk, k);
    t = splr_dpp<0x111>(ch_in);
    const float chl = k == 0 ? -tail : T2 * t + (-tail);
    const float chr = k == SPL_K - 1 ? tail : T2 * ch_in + (-tail);
    const float ht = chr - chl;
    // knot derivatives: the right knot of bin k is inner knot k + 1 (k < 7), the end knots are 1
    // softplus(softplus(v)) = log(1 + e^{log(1 + e^v)}) = log(2 + e^v), and its derivative sigmoid(softplus(v)) sigmoid(v) = e^v / (2 + e^v):
    // one exponential, one logarithm, one reciprocal for what is written as two softplus and two sigmoids (v > 20: both softplus
    // are the identity there, networks.py / F.softplus threshold, and the derivative is 1 to 2e-9)
    const float ev = spl_exp(rd < 20.f ? rd : 20.f);
    const float dr = 1e-3f + (rd > 20.f ? rd : spl_log(2.f + ev));
    const float d1 = k == SPL_K - 1 ? 1.0f : dr;
    t = splr_dpp<0x111>(d1);
    const float d0 = k == 0 ? 1.0f : t;
    const float dd = k == SPL_K - 1 ? 0.f : (rd > 20.f ? 1.f : ev * spl_rcp(2.f + ev));
    // searchsorted (networks.py:417-422): the inner edges <= x (edge 0 = -tail always is, the last one + 1e-6 never)
    const float cnt = splr_sum8((k >= 1 && x >= left) ? 1.f : 0.f);
    const bool sel = (float)k == cnt;
    // the rational-quadratic map on this lane's bin (networks.py:541-556); only the selected lane's result is used
    const float ribw = spl_rcp(width);
    const float delta = ht * ribw;
    const float theta = (x - left) * ribw;
    const float tomt = theta * (1.f - theta);
    const float Nn = ht * (delta * theta * theta + d0 * tomt);
    const float Dn = delta + (d0 + d1 - 2.f * delta) * tomt;
    const float Q = d1 * theta * theta + 2.f * delta * tomt + d0 * (1.f - theta) * (1.f - theta);
    const bool use = sel && inside;
    const float rDn = spl_rcp(Dn);
    const float y_own = use ? chl + Nn * rDn : 0.f;
    const float y = inside ? splr_sum8(y_own) : x;
    if (use && active) ld += spl_log((delta * rDn) * (delta * rDn) * Q);   // log(delta^2 Q) - 2 log(Dn)
    kp.a_w = a_w; kp.p_w = p_w; kp.a_h = a_h; kp.p_h = p_h;
    kp.dd = dd; kp.left = left; kp.width = width; kp.chl = chl;
    kp.ht = ht; kp.d0 = d0; kp.d1 = d1; kp.x = x;
    kp.cnt = cnt; kp.use = use ? 1.f : 0.f; kp.ribw = ribw; kp.theta = theta;
    return y;
}

// reverse mode of the same (spl_rqs_fwd_bwd restated on the 8-lane layout): gy = dLoss/dy of the item, gl = dLoss/d(log-derivative).
// Returns dLoss/dx (all 8 lanes) and this lane's dLoss/d(rw, rh, rd).
__device__ __forceinline__ float splr_eval_bwd(const SplrKeep &kp, float tail, int k, float gy, float gl, float &g_rw, float &g_rh, float &g_rd) {
    const float T2 = 2.f * tail, cmin = 1.f - 1e-3f * SPL_K;
    const bool inside = kp.x >= -tail && kp.x <= tail;
    const bool use = kp.use != 0.f;
    const float ih = kp.ht, d0 = kp.d0, d1 = kp.d1;
    const float ribw = kp.ribw, theta = kp.theta;
    const float delta = ih * ribw;
    const float tomt = theta * (1.f - theta);
    const float sdd = d0 + d1 - 2.f * delta;
    const float Nn = ih * (delta * theta * theta + d0 * tomt);
    const float Dn = delta + sdd * tomt;
    const float rDn = spl_rcp(Dn);
    const float Q = d1 * theta * theta + 2.f * delta * tomt + d0 * (1.f - theta) * (1.f - theta);
    const float dn = delta * delta * Q;
    float g_ich = gy, g_N = gy * rDn, g_Dn = -gy * Nn * rDn * rDn - 2.f * gl * rDn;
    const float g_dn = gl * spl_rcp(dn);
    float g_delta = g_dn * (2.f * delta * Q + delta * delta * 2.f * tomt);
    const float g_Q = g_dn * delta * delta;
    float g_d1 = g_Q * theta * theta, g_d0 = g_Q * (1.f - theta) * (1.f - theta);
    float g_theta = g_Q * (2.f * d1 * theta - 2.f * d0 * (1.f - theta));
    float g_t = g_Q * 2.f * delta;
    g_delta += g_Dn * (1.f - 2.f * tomt);
    g_d0 += g_Dn * tomt; g_d1 += g_Dn * tomt;
    g_t += g_Dn * sdd;
    float g_ih = g_N * (delta * theta * theta + d0 * tomt);
    g_delta += g_N * ih * theta * theta;
    g_theta += g_N * ih * delta * 2.f * theta;
    g_d0 += g_N * ih * tomt;
    g_t += g_N * ih * d0;
    g_theta += g_t * (1.f - 2.f * theta);
    g_ih += g_delta * ribw;
    float g_ibw = -g_delta * ih * ribw * ribw;
    const float gx_own = g_theta * ribw;
    const float g_icw = -g_theta * ribw;
    g_ibw += -g_theta * theta * ribw;
    // the selected lane's results, in all 8 lanes
    const float G_icw = splr_sum8(use ? g_icw : 0.f), G_ibw = splr_sum8(use ? g_ibw : 0.f);
    const float G_ich = splr_sum8(use ? g_ich : 0.f), G_ih = splr_sum8(use ? g_ih : 0.f);
    const float gx = inside ? splr_sum8(use ? gx_own : 0.f) : gy;
    // knot construction, reverse (spl_knots_bwd): edge_k = -B + 2B sum_{i<k} w_i, size_k = 2B w_k
    const bool below = (float)k < kp.cnt, at = (float)k == kp.cnt;
    {
        const float gp = cmin * (T2 * ((below ? G_icw : 0.f) + (at ? G_ibw : 0.f)));
        const float dot = splr_sum8(kp.p_w * gp);
        const float ga = T2 * (kp.p_w * (gp - dot));
        const float dot2 = splr_sum8(kp.a_w * ga);
        g_rw = inside ? kp.a_w * (ga - dot2) : 0.f;
    }
    {
        const float gp = cmin * (T2 * ((below ? G_ich : 0.f) + (at ? G_ih : 0.f)));
        const float dot = splr_sum8(kp.p_h * gp);
        const float ga = T2 * (kp.p_h * (gp - dot));
        const float dot2 = splr_sum8(kp.a_h * ga);
        g_rh = inside ? kp.a_h * (ga - dot2) : 0.f;
    }
    // this lane's logit is the right knot of its bin: d1 of bin k, d0 of bin k + 1
    const float from_next = splr_dpp<0x101>(use ? g_d0 : 0.f);   // row_shl:1: lane k <- lane k + 1
    g_rd = ((use ? g_d1 : 0.f) + (k < SPL_K - 1 ? from_next : 0.f)) * kp.dd;
    return gx;
}

// ---- the gradient kernel -----------------------------------------------------------------------------------------------------
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes at a dword-aligned address: the packed vector's offsets are odd
__device__ __forceinline__ f32x4 splr_ld4(const float *p) { const f32x4u v = *reinterpret_cast<const f32x4u *>(p); return (f32x4){v.x, v.y, v.z, v.w}; }

// The conditioner's trunk and last layer on the matrix cores, one row per workgroup: y = W x as D = A B with A the weights and B the
// vector in ALL sixteen columns -- fifteen of them are redundant, but a 16 -> 16 layer is four v_mfma_f32_16x16x4 instead of 16
// v_readlane + 16 v_fmac on a lone wave's dependent chain, and a lane holds 4 weights of the layer instead of 16.
// Layout of a 16-vector y: the lanes of k-group lk = lane >> 4 hold y[4 lk + r] in register r (the same in all 16 columns).  The
// contraction index of step s, group lk is 4 lk + s -- so a lane's four A operands of a layer are 16 contiguous bytes of the weight row
// (one load), its B operand of step s is its own register s of the layer's input, and a product leaves row 4 lk + r of D in register r:
// the layers chain without any exchange between lanes.  (A first version had the contraction index 4 s + lk: four 4-byte loads per
// lane and layer, 79 loads in flight per wave -- more than the 63 the counter holds: the prefetch serialised, 4.4 us per last layer.)
__device__ __forceinline__ f32x4 splr_lrelu4(f32x4 v) { return (f32x4){splr_lrelu(v.x), splr_lrelu(v.y), splr_lrelu(v.z), splr_lrelu(v.w)}; }
__device__ __forceinline__ f32x4 splr_slope4(f32x4 g, f32x4 post) {
    return (f32x4){g.x * splr_slope(post.x), g.y * splr_slope(post.y), g.z * splr_slope(post.z), g.w * splr_slope(post.w)};
}
// D += W[:, 4 lk .. 4 lk + 3] x[4 lk .. 4 lk + 3] over the four k-groups: four products on two accumulators
__device__ __forceinline__ f32x4 splr_mv16(const f32x4 &w, const f32x4 &x, f32x4 acc) {
    f32x4 a2 = mfma4(w.y, x.y, (f32x4){0.f, 0.f, 0.f, 0.f});
    acc = mfma4(w.x, x.x, acc);
    a2 = mfma4(w.w, x.w, a2);
    acc = mfma4(w.z, x.z, acc);
    return acc + a2;
}

// Every set is loaded right behind the LAST USE of the set before it (the trunk of coupling c + 1 behind the trunk of coupling c, ...):
// the registers are the same, the loads have the rest of the running coupling to arrive, and no pass opens with a round trip to L2
// (stamps of the first version: 1.8 us per coupling waited).
template <int NC>
struct SplrTrunkF {   // forward trunk: this lane's quarter of row c16 of the three layers, the biases of its four units
    f32x4 t0[NC], t1, t2, b0, b1, b2;
    __device__ __forceinline__ void load(const float *pn, int nin, int c16, int lk) {
        const float *W0 = pn, *pb0 = pn + 16 * nin, *W1 = pb0 + 16, *pb1 = W1 + 256, *W2 = pb1 + 16, *pb2 = W2 + 256;
#pragma unroll
        for (int t = 0; t < NC; ++t) t0[t] = splr_ld4(W0 + c16 * nin + 16 * t + 4 * lk);   // (past the row's end: inside the net, finite; the B operand there is 0)
        t1 = splr_ld4(W1 + c16 * 16 + 4 * lk); t2 = splr_ld4(W2 + c16 * 16 + 4 * lk);
        b0 = splr_ld4(pb0 + 4 * lk); b1 = splr_ld4(pb1 + 4 * lk); b2 = splr_ld4(pb2 + 4 * lk);
    }
};
template <int NW, int NS>
struct SplrLastW {    // the last layer, tile st of this wave = output rows 16 (wv + NW st) .. + 15: this lane's quarter 4 lk .. of row c16 of
    f32x4 l3[NS];     // the tile -- forward as A operands, on the way back as the weights of sum_o g[o] W3[o][:]; e0..2: the three biases
    float e0, e1, e2; // of the outputs this lane reads back as (item, bin), added there
    __device__ __forceinline__ void load(const float *pn, int nin, int nout, int wv, int c16, int lk, int item, int k, bool bias) {
        const float *W3 = pn + 16 * nin + 16 + 2 * (256 + 16), *pb3 = W3 + (size_t)SPL_P * nout * 16;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            // (rows past the outputs -- the biases, the next net, or w_dev's zeroed slack: finite values, never stored forward, times 0 on
            // the way back; no clamp: the tiles' addresses stay one base + a constant stride)
            l3[st] = splr_ld4(W3 + (size_t)(16 * (wv + NW * st) + c16) * 16 + 4 * lk);
        }
        if (bias) {
            const int jb = SPL_P * (item < nout ? item : 0);
            e0 = pb3[jb + k]; e1 = pb3[jb + 8 + k]; e2 = pb3[jb + 16 + (k < 7 ? k : 6)];
        }
    }
};
// The last layer forward on v_mfma_f32_4x4x1 (sixteen independent 4x4 outer products an instruction, 8 cycles): block b = lane >> 2,
// row i = lane & 3 of the block -- lane L supplies output row 64 g + L of group g, the vector element of step k is the B operand of
// every lane, and register i of the block's lanes is output row 64 g + 4 b + i.  64 output rows per 16 instructions of 8 cycles against
// 16 rows per 4 instructions of 32 cycles for the 16x16x4 shape (whose sixteen columns all carry the same vector): four times the
// rows per matrix-core cycle.  A lane's operands are its whole weight row: 64 contiguous bytes.
template <int NW>
struct SplrLastF {
    enum { NG = 3 };   // groups of 64 rows per wave: 23 * 32 rows are 12 groups, NW <= 4 waves take three each (8 n_out <= 64 NW lanes)
    f32x4 w[NG][4];
    float e0, e1, e2;   // the three biases of the outputs this lane reads back as (item, bin), added there
    __device__ __forceinline__ void load(const float *pn, int nin, int nout, int wv, int lane, int item, int k) {
        const float *W3 = pn + 16 * nin + 16 + 2 * (256 + 16), *pb3 = W3 + (size_t)SPL_P * nout * 16;
#pragma unroll
        for (int sg = 0; sg < NG; ++sg) {
            const float *row = W3 + (size_t)(64 * (wv + NW * sg) + lane) * 16;   // (rows past the outputs: finite -- w_dev's slack -- and not stored)
#pragma unroll
            for (int q = 0; q < 4; ++q) w[sg][q] = splr_ld4(row + 4 * q);
        }
        const int jb = SPL_P * (item < nout ? item : 0);
        e0 = pb3[jb + k]; e1 = pb3[jb + 8 + k]; e2 = pb3[jb + 16 + (k < 7 ? k : 6)];
    }
};
__device__ __forceinline__ f32x4 mfma1(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

template <int NC>
struct SplrTrunkB {   // the trunk's transposed layers: A[i = c16][k = 4 lk + s] = W[4 lk + s][i]
    float c2[4], c1[4], c0[NC][4];
    __device__ __forceinline__ void load(const float *pn, int nin, int c16, int lk) {
        const float *W0 = pn, *W1 = pn + 16 * nin + 16, *W2 = W1 + 256 + 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c2[q] = W2[(4 * lk + q) * 16 + c16];
            c1[q] = W1[(4 * lk + q) * 16 + c16];
#pragma unroll
            for (int t = 0; t < NC; ++t) c0[t][q] = W0[(4 * lk + q) * nin + 16 * t + c16];   // (columns past the inputs: the next rows' weights, finite; those outputs are not written back)
        }
    }
};

template <int NW, int NS>   // waves per row; last-layer tiles per wave
__global__ void __launch_bounds__(64 * NW) splr_grad_kernel(SplRowsArgs a) {
    constexpr int NC = NW > 2 ? 2 : 1;   // 16-lane chunks of a conditioner's inputs (x_dim <= 16 NW)
    constexpr int NT = 64 * NW;
    constexpr int PM = 64 / NW;          // conv: input indices per wave (x_dim <= 64)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // The early-stopping flag of the running training call is REQUESTED here and looked at behind the first batch of loads: as the
    // kernel's first statement it was a cold round trip of its own.  A plain (volatile) load: the compiler's own wait sits at the use.
    // (An inline-asm load whose result is waited for later can be copied by the register allocator BEFORE the data is there.)
    int stop_flag = a.stop ? *reinterpret_cast<const volatile int *>(a.stop) : 0;
    const SplTrainShape &ts = a.ts;
    const SplineShape &s = ts.s;
    const int D = s.D, B = s.B, nl = s.nl, nu = s.nu;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), o16 = lane & 15;
    const int item = tid >> 3, k = tid & 7;
    float *xrow = lds;                       // [64] the row
    float *grow = xrow + 64;                 // [64] its gradient
    float *red = grow + 64;                  // [NW][64]
    float *hred = red + NW * 64;             // [4 NW][16]
    float *trk = hred + 4 * NW * 16;         // [2B][3][16] trunk activations
    float *rawbuf = trk + 2 * B * 48;        // [grow]
    f32x4 *keep = reinterpret_cast<f32x4 *>(rawbuf + a.lay.grow);   // [2B][NT][4]
    const bool vrow = (int)blockIdx.x >= a.M;
    const int row = vrow ? (int)blockIdx.x - a.M : (int)blockIdx.x;
    const int jl = lane < D ? lane : D - 1;
    // the row vector and its gradient in LDS: the lower half (the first coupling's conditioning dims) at 0, the upper half at 32 -- both
    // halves start on 16 bytes whatever x_dim (the trunk reads its input as whole 16-byte quarters); the slots past a half stay 0
    const int xo = jl < nl ? jl : 32 + jl - nl;
    long src = row;
    if (!vrow && a.perm) src = a.perm[row];
    // (No warm-up of this XCD's L2 as the tile form has one: every weight is read once per workgroup, a coupling ahead of its use, and
    // touching all 380 KB up front made the first barrier wait for all of it: 0.498 -> 0.483 ms per epoch without.)
    // data = X[perm] + jitter * randn (trainer.py:392); the noise of dim d is component d & 3 of quad d >> 2 (spl_grad_kernel's numbering)
    if (wv == 0) {
        float xv = (vrow ? a.xv : a.x)[(size_t)src * D + jl];
        if (!vrow && a.jitter != 0.f) {
            float nz;
            if (a.noise) nz = a.noise[(size_t)row * D + jl];
            else {
                const f32x4 n4 = noise_normal4(a.seed, (uint64_t)(a.noise_row0 + row), (uint32_t)a.epoch, (uint32_t)(jl >> 2), NOISE_STREAM_JITTER);
                nz = splr_comp(n4, jl & 3);
            }
            xv += nz * a.jitter;
        }
        xrow[lane] = 0.f; grow[lane] = 0.f;   // (64 slots; lane < 64 = this wave)
        __builtin_amdgcn_wave_barrier();
        if (lane < D) xrow[xo] = xv;
    }
    // the first block's conv column, ActNorm vectors and the first coupling's weights: requested here, used behind the first barrier
    const int ci0 = wv * PM;
    auto load_col = [&](const float *Mt, float (&wcol)[PM]) {
#pragma unroll
        for (int t = 0; t < PM; ++t) wcol[t] = Mt[(size_t)(ci0 + t) * D + jl];   // (rows past x_dim: the next block's, or the zeroed slack; times 0)
    };
    float wcol[PM], an_s, an_t;
    SplrTrunkF<NC> tw;
    SplrLastF<NW> lf;      // forward
    SplrLastW<NW, NS> lw;  // the way back
    SplrTrunkB<NC> bw;
    const int lk = lane >> 4;
    load_col(a.wmat, wcol);
    an_s = a.w[ts.p_s + jl]; an_t = a.w[ts.p_t + jl];
    float ldc_sum = 0.f;
    for (int b = 0; b < B; ++b) ldc_sum += a.ldc[b];
    tw.load(a.w + ts.p_f[0], nl, o16, lk);
    lf.load(a.w + ts.p_f[0], nl, nu, wv, lane, item, k);
    if (stop_flag) return;   // (uniform over the workgroup, in front of its first barrier)
    splr_barrier();
    float *const T0 = a.stg + a.lay.off_T, *const G0 = a.stg + a.lay.off_G, *const V0 = a.stg + a.lay.off_V;
    const size_t rc = (size_t)a.lay.rows_cap;
    float ld = 0.f;
#ifdef NNEST_STAMP
    long long st_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_a = wall_clock64();
    const long long st_0 = st_a;
#define R_STAMP(i) { const long long st_n = wall_clock64(); st_t[i] += st_n - st_a; st_a = st_n; }
#else
#define R_STAMP(i)
#endif

    // c = v M of the row vector held one element per lane (M's rows ci0 .. of this wave in `wcol`); result in every wave's lanes j < D
    auto matvec = [&](const float (&wc)[PM], float v) -> float {
        // (v is 0 in the lanes past x_dim, so rows past x_dim of M -- finite -- add nothing; two accumulators: the chain is half as long)
        float acc = 0.f, acc1 = 0.f;
#pragma unroll
        for (int t = 0; t < PM; t += 2) {
            acc = fmaf(splr_rl(v, ci0 + t), wc[t], acc);
            if (t + 1 < PM) acc1 = fmaf(splr_rl(v, ci0 + t + 1), wc[t + 1], acc1);
        }
        acc += acc1;
        red[wv * 64 + lane] = acc;
        splr_barrier();
        float c = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) c += red[w * 64 + lane];
        return c;
    };
    auto net_of = [&](int ci) -> const float * { return a.w + (size_t)(ci >> 1) * s.blk_params + ts.p_f[ci & 1]; };

    // ---- forward (networks.py:24-32) ----------------------------------------------------------------------------------------
    for (int b = 0; b < B; ++b) {
        // ActNorm (networks.py:672-677) and the 1x1 conv z = x W (:649)
        {
            const float xd = xrow[xo];
            const float av = lane < D ? xd * spl_exp(an_s) + an_t : 0.f;
            if (wv == 0 && !vrow && lane < D) V0[((size_t)(3 * b + 0) * rc + row) * SPLR_VROW + lane] = av;
            const float c = matvec(wcol, av);
            if (wv == 0 && lane < D) xrow[xo] = c;
            if (b + 1 < B) {   // the next block's: two couplings to arrive
                const float *pbn = a.w + (size_t)(b + 1) * s.blk_params;
                load_col(a.wmat + (size_t)(b + 1) * D * D, wcol);
                an_s = pbn[ts.p_s + jl]; an_t = pbn[ts.p_t + jl];
            }
            splr_barrier();
        }
        R_STAMP(0)
#pragma unroll 1
        for (int c = 0; c < 2; ++c) {
            const int ci = 2 * b + c;
            const int nin = c ? nu : nl, nout = c ? nl : nu, idoff = c ? 32 : 0, troff = c ? 0 : 32;
            // trunk (networks.py:393-409): Linear LReLU x3 on the matrix cores, every wave the same
            f32x4 ub[NC];
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                const int i = 16 * t + 4 * lk;
                ub[t] = *reinterpret_cast<const f32x4 *>(xrow + idoff + i);   // (one 16-byte read; 0 past the half's dims)
            }
            const float uL = lane < nin ? xrow[idoff + lane] : 0.f;   // (for the staging only)
            R_STAMP(1)
            f32x4 acc = tw.b0;
#pragma unroll
            for (int t = 0; t < NC; ++t) acc = splr_mv16(tw.t0[t], ub[t], acc);
            const f32x4 h0 = splr_lrelu4(acc);
            const f32x4 h1 = splr_lrelu4(splr_mv16(tw.t1, h0, tw.b1));
            const f32x4 h2 = splr_lrelu4(splr_mv16(tw.t2, h1, tw.b2));
            // the next coupling's trunk into the same registers
            if (ci + 1 < 2 * B) tw.load(net_of(ci + 1), c ? nl : nu, o16, lk);
            if (wv == 0 && !vrow) {   // staged for the contractions, kept for the way back: group lk's four units, 16 bytes
                float *Tr = T0 + ((size_t)ci * rc + row) * SPLR_TROW;
                if (o16 == 0) {
                    *reinterpret_cast<f32x4 *>(Tr + 4 * lk) = h0; *reinterpret_cast<f32x4 *>(Tr + 16 + 4 * lk) = h1; *reinterpret_cast<f32x4 *>(Tr + 32 + 4 * lk) = h2;
                    *reinterpret_cast<f32x4 *>(trk + ci * 48 + 4 * lk) = h0;
                    *reinterpret_cast<f32x4 *>(trk + ci * 48 + 16 + 4 * lk) = h1;
                    *reinterpret_cast<f32x4 *>(trk + ci * 48 + 32 + 4 * lk) = h2;
                }
                Tr[96 + lane] = uL;
            }
            R_STAMP(2)
            // last layer: group sg of this wave, 64 output rows per sixteen 4x4x1 products (SplrLastF); the vector's sixteen elements first,
            // each in every lane
            float hb[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hb[4 * q + 0] = splr_rl(h2.x, 16 * q); hb[4 * q + 1] = splr_rl(h2.y, 16 * q);
                hb[4 * q + 2] = splr_rl(h2.z, 16 * q); hb[4 * q + 3] = splr_rl(h2.w, 16 * q);
            }
            f32x4 rl3[SplrLastF<NW>::NG];
#pragma unroll
            for (int sg = 0; sg < SplrLastF<NW>::NG; ++sg) rl3[sg] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int sg = 0; sg < SplrLastF<NW>::NG; ++sg) rl3[sg] = mfma1(splr_comp(lf.w[sg][q], e), hb[4 * q + e], rl3[sg]);
            if ((lane & 3) == 0) {
#pragma unroll
                for (int sg = 0; sg < SplrLastF<NW>::NG; ++sg) {
                    const int o = 64 * (wv + NW * sg) + lane;   // rows o .. o + 3 of the row: register i is row 4 b + i of the group
                    if (o + 3 < a.lay.grow) *reinterpret_cast<f32x4 *>(rawbuf + o) = rl3[sg];
                }
            }
            const float e0 = lf.e0, e1 = lf.e1, e2 = lf.e2;
            if (ci + 1 < 2 * B) lf.load(net_of(ci + 1), c ? nl : nu, c ? nu : nl, wv, lane, item, k);
            splr_barrier();
            R_STAMP(3)
            // the spline, eight lanes per item (networks.py:583-587, :425-556)
            {
                const bool active = item < nout;
                const int jb = SPL_P * (active ? item : 0);
                const float rw = rawbuf[jb + k] + e0, rh = rawbuf[jb + 8 + k] + e1, rd = rawbuf[jb + 16 + (k < 7 ? k : 6)] + e2;
                const float x = xrow[troff + (active ? item : 0)];
                SplrKeep kp;
                const float y = splr_eval(rw, rh, rd, x, s.tail, k, active, ld, kp);
                if (!vrow) {
                    f32x4 *kq = keep + ((size_t)ci * NT + tid) * 4;
                    kq[0] = (f32x4){kp.a_w, kp.p_w, kp.a_h, kp.p_h}; kq[1] = (f32x4){kp.dd, kp.left, kp.width, kp.chl};
                    kq[2] = (f32x4){kp.ht, kp.d0, kp.d1, kp.x}; kq[3] = (f32x4){kp.cnt, kp.use, kp.ribw, kp.theta};
                }
                if (active && k == 0) xrow[troff + item] = y;
            }
            splr_barrier();
            R_STAMP(4)
        }
    }
    // log_probs (networks.py:71-76): base density of z + the log-determinants
    {
        if (!vrow) {   // the way back opens with the last block: its W^T column and ActNorm scale, its second coupling's transposed weights
            load_col(a.wmatT + (size_t)(B - 1) * D * D, wcol);
            an_s = a.w[(size_t)(B - 1) * s.blk_params + ts.p_s + jl];
            bw.load(net_of(2 * B - 1), nu, o16, lk);
            lw.load(net_of(2 * B - 1), nu, nl, wv, o16, lk, item, k, false);
        }
        // one reduction: the lanes' log-derivatives, minus (wave 0) the base density's terms; the blocks' log-det constants were
        // requested in the prologue (written by the update kernel a launch ago: a cold load here was 1 us of the row's chain)
        float t = ld - ((wv == 0 && lane < D) ? base_E(xrow[xo], s.base_beta) : 0.f);
        t += splr_dpp<0x128>(t); t += splr_dpp<0x124>(t); t += splr_dpp<0x122>(t); t += splr_dpp<0x121>(t);   // the four rows of the wave
        t = (splr_rl(t, 0) + splr_rl(t, 16)) + (splr_rl(t, 32) + splr_rl(t, 48));
        if (lane == 0) red[wv] = t;
        splr_barrier();
        if (tid == 0) {
            float lt = ldc_sum;
            for (int w = 0; w < NW; ++w) lt += red[w];
            a.rowlp[vrow ? a.M + row : row] = s.base_const * (float)D + lt;
        }
    }
    if (vrow) return;
    R_STAMP(5)

    // ---- backward: loss = -mean(log_probs) (trainer.py:394) ------------------------------------------------------------------
    const float invM = 1.0f / (float)a.mtot, gld = -invM;
    if (wv == 0 && lane < D) grow[xo] = base_dE(xrow[xo], s.base_beta) * invM;
    splr_barrier();
    for (int b = B - 1; b >= 0; --b) {
#pragma unroll 1
        for (int c = 1; c >= 0; --c) {
            const int ci = 2 * b + c;
            const int nout = c ? nl : nu, idoff = c ? 32 : 0, troff = c ? 0 : 32;
            const int nrows = SPL_P * nout;
            // evaluation, reverse
            {
                const bool active = item < nout;
                const f32x4 *kq = keep + ((size_t)ci * NT + tid) * 4;
                const f32x4 k0 = kq[0], k1 = kq[1], k2 = kq[2], k3 = kq[3];
                SplrKeep kp;
                kp.a_w = k0.x; kp.p_w = k0.y; kp.a_h = k0.z; kp.p_h = k0.w; kp.dd = k1.x; kp.left = k1.y; kp.width = k1.z; kp.chl = k1.w;
                kp.ht = k2.x; kp.d0 = k2.y; kp.d1 = k2.z; kp.x = k2.w; kp.cnt = k3.x; kp.use = k3.y; kp.ribw = k3.z; kp.theta = k3.w;
                const float gy = grow[troff + (active ? item : 0)];
                float g_rw, g_rh, g_rd;
                const float gx = splr_eval_bwd(kp, s.tail, k, gy, gld, g_rw, g_rh, g_rd);
                if (active) {
                    const int jb = SPL_P * item;
                    rawbuf[jb + k] = g_rw; rawbuf[jb + 8 + k] = g_rh;
                    if (k < 7) rawbuf[jb + 16 + k] = g_rd;
                    if (k == 0) grow[troff + item] = gx;
                }
            }
            splr_barrier();
            R_STAMP(6)
            // dLoss/d(raw) of the row: staged for the contraction over rows; dLoss/dh2 = W3^T g: this lane's row of each tile times its
            // quarter of W3, then the sum over the tile's rows (the 16 columns of the lane's k-group) -- group lk ends with units 4 lk ..
            {
                float pt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < NS; ++st) {
                    const int o = 16 * (wv + NW * st) + o16;
                    const float g = o < nrows ? rawbuf[o < nrows ? o : 0] : 0.f;
                    pt[0] = fmaf(g, lw.l3[st].x, pt[0]); pt[1] = fmaf(g, lw.l3[st].y, pt[1]); pt[2] = fmaf(g, lw.l3[st].z, pt[2]); pt[3] = fmaf(g, lw.l3[st].w, pt[3]);
                }
                if (ci > 0) lw.load(net_of(ci - 1), c ? nl : nu, c ? nu : nl, wv, o16, lk, item, k, false);   // the coupling before, into the same registers
                float *Gr = G0 + ((size_t)ci * rc + row) * a.lay.grow;
                for (int o = tid; o < nrows; o += NT) Gr[o] = rawbuf[o];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pt[e] += splr_dpp<0x128>(pt[e]);   // row_ror 8, 4, 2, 1: every lane of the row ends with the row's sum
                    pt[e] += splr_dpp<0x124>(pt[e]);
                    pt[e] += splr_dpp<0x122>(pt[e]);
                    pt[e] += splr_dpp<0x121>(pt[e]);
                }
                if (o16 == 0) *reinterpret_cast<f32x4 *>(hred + wv * 16 + 4 * lk) = (f32x4){pt[0], pt[1], pt[2], pt[3]};
            }
            splr_barrier();
            R_STAMP(7)
            // trunk, reverse (every wave the same): d = W^T d on the matrix cores
            {
                f32x4 dh = *reinterpret_cast<const f32x4 *>(hred + 4 * lk);
#pragma unroll
                for (int w = 1; w < NW; ++w) dh = dh + *reinterpret_cast<const f32x4 *>(hred + w * 16 + 4 * lk);
                const f32x4 h0 = *reinterpret_cast<const f32x4 *>(trk + ci * 48 + 4 * lk), h1 = *reinterpret_cast<const f32x4 *>(trk + ci * 48 + 16 + 4 * lk),
                            h2 = *reinterpret_cast<const f32x4 *>(trk + ci * 48 + 32 + 4 * lk);
                const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 d2 = splr_slope4(dh, h2);
                const f32x4 d1 = splr_slope4(splr_mv16((f32x4){bw.c2[0], bw.c2[1], bw.c2[2], bw.c2[3]}, d2, z4), h1);
                const f32x4 d0 = splr_slope4(splr_mv16((f32x4){bw.c1[0], bw.c1[1], bw.c1[2], bw.c1[3]}, d1, z4), h0);
                f32x4 du[NC];
#pragma unroll
                for (int t = 0; t < NC; ++t) du[t] = splr_mv16((f32x4){bw.c0[t][0], bw.c0[t][1], bw.c0[t][2], bw.c0[t][3]}, d0, z4);   // inputs 16 t + 4 lk + r
                if (ci > 0) bw.load(net_of(ci - 1), c ? nl : nu, o16, lk);
                if (wv == NW - 1 && o16 == 0) {   // (the staging stores on another wave than the gradient row's update: both tails are on the way to the barrier)
                    float *Tr = T0 + ((size_t)ci * rc + row) * SPLR_TROW;
                    *reinterpret_cast<f32x4 *>(Tr + 48 + 4 * lk) = d0; *reinterpret_cast<f32x4 *>(Tr + 64 + 4 * lk) = d1; *reinterpret_cast<f32x4 *>(Tr + 80 + 4 * lk) = d2;
                }
                if (wv == 0 && o16 == 0) {
                    // (all reads of the update first: as four guarded read-modify-writes per tile they were eight dependent LDS round
                    // trips on the one wave the other three wait for)
                    f32x4 gv[NC];
#pragma unroll
                    for (int t = 0; t < NC; ++t) gv[t] = *reinterpret_cast<const f32x4 *>(grow + idoff + 16 * t + 4 * lk);
#pragma unroll
                    for (int t = 0; t < NC; ++t) {
                        const int i = 16 * t + 4 * lk;
                        // (slots past the half's dims take the products of the next rows' weights: finite, never read)
                        if (i < 32) *reinterpret_cast<f32x4 *>(grow + idoff + i) = gv[t] + du[t];
                    }
                }
            }
            splr_barrier();
            R_STAMP(8)
        }
        // 1x1 conv c = a W: g_a = g_c W^T (dLoss/dW = sum_rows a^T g_c: the update kernel's contraction); ActNorm a = x e^s + t
        {
            const float gc = lane < D ? grow[xo] : 0.f;
            if (wv == 0 && lane < D) V0[((size_t)(3 * b + 1) * rc + row) * SPLR_VROW + lane] = gc;
            const float es = spl_exp(an_s);
            const float ga = matvec(wcol, gc);
            if (wv == 0 && lane < D) grow[xo] = ga * es;   // (ActNorm's own gradients: from dLoss/dW in the update kernel)
            if (b > 0) {
                load_col(a.wmatT + (size_t)(b - 1) * D * D, wcol);
                an_s = a.w[(size_t)(b - 1) * s.blk_params + ts.p_s + jl];
            }
            splr_barrier();
        }
        R_STAMP(9)
    }
#ifdef NNEST_STAMP
    if (blockIdx.x == 0 && tid == 0)
        printf("splr_update head (the launch before): loads issued %.0f | old head to LDS %.0f | contraction + wait %.0f | LU grads + Adam %.0f | W, W^T %.0f | logdet %.0f; job 97: %.0f (x10 ns)\n",
               a.ldc[4], a.ldc[5], a.ldc[6], a.ldc[7], a.ldc[8], a.ldc[9], a.ldc[10]);
    if (blockIdx.x == 0 && tid == 0) printf("   (LU products up to their barrier: %.0f of the LU + Adam phase)\n", a.ldc[11]);
    if (blockIdx.x == 0 && lane == 0)
        printf("splr_grad wave %d: total %lld | fwd: conv %lld, loads waited %lld, trunk %lld, last layer %lld, eval %lld | logp %lld | bwd: eval %lld, W3^T g %lld, trunk %lld, conv %lld (x10 ns)\n",
               wv, wall_clock64() - st_0, st_t[0], st_t[1], st_t[2], st_t[3], st_t[4], st_t[5], st_t[6], st_t[7], st_t[8], st_t[9]);
#endif
}

// ---- the parameter update of a minibatch: contractions over its rows + Adam, one launch ----------------------------------------
struct SplRowsUpdArgs {
    float *w, *m, *v;
    const float *stg;
    SplRowsLayout lay;
    const float *rowlp;
    const int *pi, *pi_inv;
    float *wmat, *wmatT, *ldc;
    SplTrainShape ts;
    int M;
    float step_size, inv_bc2s, wd, ldw;
    float *loss_out;
    float loss_scale;
    const int *stop;
    float *grad_out, *gwsum_out;   // both non-NULL: gradient only (packed conditioner / ActNorm gradients, dLoss/dW of the convs), no step
    int n_jobs;
};

// one Adam step of one parameter (torch/optim/adam.py _single_tensor_adam, coupled weight decay), as spl_adam_one
__device__ __forceinline__ float splr_adam_one(float w, float g, float &m, float &v, float step_size, float inv_bc2s, float wd) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const float gi = g + wd * w;
    m = m + (gi - m) * (1.0f - b1);
    v = v * b2 + (1.0f - b2) * gi * gi;
    // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the correctly rounded sequences (~25 instructions per element): the block heads
    // are one compute unit's instruction throughput (16 waves x 10 elements), and 3e-7 of an update of 1e-3 w is nothing
    return w - step_size * (m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) * inv_bc2s + eps));
}
__device__ __forceinline__ int splr_div_small(int x, int n) {   // x / n for 0 <= x < 64, 1 <= n <= 4 (tile counts): no integer division
    return n == 4 ? x >> 2 : (n == 2 ? x >> 1 : (n == 1 ? x : (x * 43) >> 7));
}

__host__ __device__ inline int splr_jobs_of(const SplineShape &s, int c) {   // conditioner c of a block: W3 tiles + W2 + W1 + W0 tiles
    const int nin = c ? s.nu : s.nl, nout = c ? s.nl : s.nu;
    return (SPL_P * nout + 15) / 16 + 2 + (nin + 15) / 16;
}

// One wave, one 16x16 tile of one weight matrix of one conditioner: dW[o][i] = sum_rows A[row][o] B[row][i] with A the staged
// gradient at the layer's output (g_raw, d2, d1, d0) and B its staged input (h2, h1, h0, u); the bias gradient is A's column sum.
template <int STEPS>
__device__ __forceinline__ void splr_job(const SplRowsUpdArgs &a, int id, int lane, int &stop_flag) {
#ifdef NNEST_STAMP
    const long long j_0 = wall_clock64();
#endif
    const SplTrainShape &ts = a.ts;
    const SplineShape &s = ts.s;
    const int j0 = splr_jobs_of(s, 0), j1 = splr_jobs_of(s, 1), jb = j0 + j1;
    const int b = id / jb;
    int r = id - b * jb;
    const int c = r >= j0 ? 1 : 0;
    r -= c * j0;
    const int ci = 2 * b + c, nin = c ? s.nu : s.nl, nout = c ? s.nl : s.nu, n3 = (SPL_P * nout + 15) / 16;
    const int pn = b * s.blk_params + ts.p_f[c];
    const int pW0 = pn, pb0 = pW0 + 16 * nin, pW1 = pb0 + 16, pb1 = pW1 + 256, pW2 = pb1 + 16, pb2 = pW2 + 256, pW3 = pb2 + 16, pb3 = pW3 + SPL_P * nout * 16;
    const size_t rc = (size_t)a.lay.rows_cap;
    const float *T = a.stg + a.lay.off_T + (size_t)ci * rc * SPLR_TROW, *G = a.stg + a.lay.off_G + (size_t)ci * rc * a.lay.grow;
    const float *Ap, *Bp;
    int as, obase = 0, ibase = 0, wbase, instride, bbase, nov, niv;
    if (r < n3) { Ap = G + 16 * r; as = a.lay.grow; Bp = T + 32; obase = 16 * r; wbase = pW3; instride = 16; bbase = pb3; nov = SPL_P * nout; niv = 16; }
    else if (r == n3) { Ap = T + 80; as = SPLR_TROW; Bp = T + 16; wbase = pW2; instride = 16; bbase = pb2; nov = 16; niv = 16; }
    else if (r == n3 + 1) { Ap = T + 64; as = SPLR_TROW; Bp = T + 0; wbase = pW1; instride = 16; bbase = pb1; nov = 16; niv = 16; }
    else { const int t = r - n3 - 2; Ap = T + 48; as = SPLR_TROW; Bp = T + 96 + 16 * t; ibase = 16 * t; wbase = pW0; instride = nin; bbase = t == 0 ? pb0 : -1; nov = 16; niv = nin; }
    const int p = lane & 15, kq = lane >> 4, M = a.M;
    // (rows past the minibatch are read as they lie -- the buffers hold 128 rows, finite values or zeros -- and masked: clamping the row
    // index made every load's address a value of its own, 2 x 52 of them live at once, spilled)
    float av[STEPS], bv[STEPS];
    {
        const float *pa = Ap + (size_t)kq * as + p, *pb = Bp + (size_t)kq * SPLR_TROW + p;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) { av[st] = *pa; bv[st] = *pb; pa += 4 * as; pb += 4 * SPLR_TROW; }
    }
    // the Adam state of the tile's elements (requested with the operands: one round trip)
    int idx[5];
    float wi[5], mi[5], vi[5];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int o = obase + 4 * kq + e, i = ibase + p;
        idx[e] = (o < nov && i < niv) ? wbase + o * instride + i : -1;
    }
    idx[4] = (kq == 0 && bbase >= 0 && obase + p < nov) ? bbase + obase + p : -1;
    const bool stepping = a.grad_out == nullptr;
#pragma unroll
    for (int e = 0; e < 5; ++e) {
        const int q = idx[e] >= 0 ? idx[e] : 0;
        wi[e] = a.w[q];
        mi[e] = stepping ? a.m[q] : 0.f;
        vi[e] = stepping ? a.v[q] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    float db = 0.f;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const bool live = 4 * st + kq < M;
        const float x = live ? av[st] : 0.f;
        acc = mfma4(x, live ? bv[st] : 0.f, acc);
        db += x;
    }
    db += __shfl_xor(db, 16);
    db += __shfl_xor(db, 32);
    const float g[5] = {acc.x, acc.y, acc.z, acc.w, db};
    if (stop_flag) return;
#pragma unroll
    for (int e = 0; e < 5; ++e) {
        if (idx[e] < 0) continue;
        if (!stepping) { a.grad_out[idx[e]] = g[e]; continue; }
        const float wn = splr_adam_one(wi[e], g[e], mi[e], vi[e], a.step_size, a.inv_bc2s, a.wd);
        a.m[idx[e]] = mi[e]; a.v[idx[e]] = vi[e]; a.w[idx[e]] = wn;
    }
#ifdef NNEST_STAMP
    if (id == 97 && lane == 0) a.ldc[10] = (float)(wall_clock64() - j_0);
#endif
}

// Workgroup b < B: the head of block b (ActNorm s, t; the 1x1 conv's L, S, U) -- dLoss/dW = a^T g_c contracted over the rows on the
//   matrix cores, then as spl_update_kernel's head: dLoss/dW -> dLoss/d(L, S, U), the step, W and W^T from the new values, the
//   log-det constant.
// Workgroups B .. : sixteen waves = sixteen splr_job tiles each.  The last workgroup's wave 0 also sums the rows' log_probs.
template <int STEPS>   // k-steps of the contractions over rows: 4 STEPS >= the minibatch
__global__ void __launch_bounds__(1024) splr_update_kernel(SplRowsUpdArgs a) {
    // The early-stopping flag of the running training call is REQUESTED here and looked at behind the first batch of loads: as the
    // kernel's first statement it was a cold round trip of its own.  A plain (volatile) load: the compiler's own wait sits at the use.
    // (An inline-asm load whose result is waited for later can be copied by the register allocator BEFORE the data is there.)
    int stop_flag = a.stop ? *reinterpret_cast<const volatile int *>(a.stop) : 0;
    extern __shared__ float ulds[];
    const SplTrainShape &ts = a.ts;
    const SplineShape &s = ts.s;
    const int D = s.D, B = s.B, nhead = ts.p_f[0], tid = threadIdx.x, M = a.M;
    const int lane = tid & 63, wave = tid >> 6;
    const bool stepping = a.grad_out == nullptr;
    if ((int)blockIdx.x >= B) {
        const int id = ((int)blockIdx.x - B) * 16 + wave;
        if (id < a.n_jobs) splr_job<STEPS>(a, id, lane, stop_flag);
        else if (id == a.n_jobs && a.loss_out) {   // loss = -mean(log_probs) of the minibatch, rows added in a fixed order
            float v = (lane < M ? a.rowlp[lane] : 0.f) + (lane + 64 < M ? a.rowlp[lane + 64] : 0.f);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            if (!stop_flag && lane == 0) *a.loss_out = v * a.loss_scale;
        }
        return;
    }
    const int b = blockIdx.x, base = b * s.blk_params;
#ifdef NNEST_STAMP
    long long u_t[8], u_f[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u_t[0] = wall_clock64();
#define F_STAMP(i) u_f[i] = wall_clock64();
#define U_STAMP(i) u_t[i] = wall_clock64();
#else
#define U_STAMP(i)
#define F_STAMP(i)
#endif
    constexpr int DP = 65, MAT = 64 * DP;
    // LDS: five 64 x 64 operand matrices at stride 65 (zero-padded; the stride keeps a column walk off one bank): G = the row-permuted
    // dLoss/dW, Um = triu(U,1) + diag(S) and Lm = tril(L,-1) + I of the old values; UmN / LmN first take the two products' results
    // (dLoss/dUm, dLoss/dL), then the new values.
    float *G = ulds, *Um = G + MAT, *Lm = Um + MAT, *UmN = Lm + MAT, *LmN = UmN + MAT, *snew = LmN + MAT, *sact = snew + 64, *tval = sact + 64;
    int *spi = reinterpret_cast<int *>(tval + 64);
    float *anp = reinterpret_cast<float *>(spi + 64);   // [2][4][64]: ActNorm partial sums, one per column tile of the contraction
    (void)nhead;
    const int nwv = (int)(blockDim.x >> 6), nt = (D + 15) >> 4, li = lane & 15, lk = lane >> 4;
    const size_t rc = (size_t)a.lay.rows_cap;
    const float *Va = a.stg + a.lay.off_V + (size_t)(3 * b + 0) * rc * SPLR_VROW, *Vgc = Va + rc * SPLR_VROW;
    // ---- every global load of the head, one batch: the contraction's operands (what other XCDs' gradient workgroups wrote: the long
    // round trip), this lane's elements of the old W, and the head's parameters with their Adam state in rows of 64 columns -- thread
    // (row = tid >> 6 + 16 u, col = lane) owns L[row][col] and U[row][col]; the thread on the diagonal owns S[col] too; thread tid < 2 D
    // owns ActNorm's s / t.  Coalesced, no division, and the Adam step below runs in the same layout.
    float av[STEPS], bv[STEPS];
    const bool ctile = wave < nt * nt;
    const int cti = splr_div_small(wave, nt), ctj = wave - cti * nt;
    {
        const float *pa = Va + (size_t)lk * SPLR_VROW + 16 * (ctile ? cti : 0) + li, *pb = Vgc + (size_t)lk * SPLR_VROW + 16 * (ctile ? ctj : 0) + li;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) { av[st] = pa[st * 4 * SPLR_VROW]; bv[st] = pb[st * 4 * SPLR_VROW]; }   // (rows past the minibatch: masked below)
    }
    // ActNorm's gradients come out of the same contraction: sum_rows g_a a = sum_j W[d][j] dW[d][j] and sum_rows g_a = sum_j W[d][j]
    // colsum_j(g_c), because g_a = g_c W^T -- this lane's four elements of the (old) W instead of two more operand streams
    float Wt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int d = 16 * cti + 4 * lk + r, j = 16 * ctj + li;
        Wt[r] = a.wmat[(size_t)b * D * D + ((ctile && d < D && j < D) ? d * D + j : 0)];
    }
    int prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int i = 16 * cti + 4 * lk + r; prow[r] = (ctile && i < D) ? a.pi[b * D + i] : -1; }
    for (int i = tid; i < D; i += blockDim.x) spi[i] = a.pi[b * D + i];
    const int hrow = tid >> 6, hcol = lane;
    float wL[4], wU[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = hrow + 16 * u, o = (row < D && hcol < D) ? row * D + hcol : 0;
        wL[u] = a.w[base + ts.p_L + o]; wU[u] = a.w[base + ts.p_U + o];
    }
    const bool diag = (lane & 15) == hrow && hcol < D;   // exactly one u has row == col = lane: u = lane >> 4
    const int sidx = base + ts.p_S + (diag ? hcol : 0);
    float wS = a.w[sidx];
    const bool an = tid < ts.p_L;
    const int aidx = base + (an ? tid : 0);
    float wA = a.w[aidx];
    __builtin_amdgcn_sched_barrier(0);
    U_STAMP(1)
    for (int idx = tid; idx < MAT; idx += blockDim.x) {  // the padding of the five matrices (no operand yet: the loads are in flight)
        const int row = idx / DP, col = idx - row * DP;
        if (row >= D || col >= D) { G[idx] = 0.f; Um[idx] = 0.f; Lm[idx] = 0.f; UmN[idx] = 0.f; LmN[idx] = 0.f; }
    }
    // dLoss/dW[i][j] = sum_rows a[row][i] g_c[row][j]
    f32x4 cacc = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const bool live = 4 * st + lk < M;
        const float gcv = live ? bv[st] : 0.f;
        cacc = mfma4(live ? av[st] : 0.f, gcv, cacc);
        csum += gcv;
    }
    csum += __shfl_xor(csum, 16);
    csum += __shfl_xor(csum, 32);   // column 16 ctj + li of g_c, summed over the rows
    if (ctile) {
        const float cv[4] = {cacc.x, cacc.y, cacc.z, cacc.w};
        const int j = 16 * ctj + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 16 * cti + 4 * lk + r;
            const bool in = d < D && j < D;
            float p0 = in ? Wt[r] * csum : 0.f, p1 = in ? Wt[r] * cv[r] : 0.f;
            p0 += splr_dpp<0x128>(p0); p1 += splr_dpp<0x128>(p1);   // over the tile's 16 columns (row_ror 8, 4, 2, 1)
            p0 += splr_dpp<0x124>(p0); p1 += splr_dpp<0x124>(p1);
            p0 += splr_dpp<0x122>(p0); p1 += splr_dpp<0x122>(p1);
            p0 += splr_dpp<0x121>(p0); p1 += splr_dpp<0x121>(p1);
            if (li == 0 && d < D) { anp[ctj * 64 + d] = p0; anp[256 + ctj * 64 + d] = p1; }
            // G = dLoss/dW with its rows permuted as P does: G[pi(i)][j]
            if (prow[r] >= 0 && j < D) {
                G[prow[r] * DP + j] = cv[r];
                if (!stepping) a.gwsum_out[(size_t)b * D * D + (size_t)d * D + j] = cv[r];
            }
        }
    }
    U_STAMP(2)
    // the Adam moments (the same layout; 82 KB through this CU's load path): requested now, behind the contraction -- they are used behind
    // two barriers and the products, and in the first batch they only stood in front of the operands the contraction waits for
    float mL[4], mU[4], vL[4], vU[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = hrow + 16 * u, o = (row < D && hcol < D) ? row * D + hcol : 0;
        mL[u] = stepping ? a.m[base + ts.p_L + o] : 0.f; mU[u] = stepping ? a.m[base + ts.p_U + o] : 0.f;
        vL[u] = stepping ? a.v[base + ts.p_L + o] : 0.f; vU[u] = stepping ? a.v[base + ts.p_U + o] : 0.f;
    }
    float mS = stepping ? a.m[sidx] : 0.f, vS = stepping ? a.v[sidx] : 0.f, mA = stepping ? a.m[aidx] : 0.f, vA = stepping ? a.v[aidx] : 0.f;
    // the old head as dense operands: Lm = tril(L,-1) + I, Um = triu(U,1) + diag(S)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = hrow + 16 * u;
        if (row < D && hcol < D) {
            Lm[row * DP + hcol] = hcol < row ? wL[u] : (hcol == row ? 1.f : 0.f);
            if (row != hcol) Um[row * DP + hcol] = row < hcol ? wU[u] : 0.f;
        }
    }
    if (diag) Um[hcol * (DP + 1)] = wS;
    if (an && tid >= ts.p_t) tval[tid - ts.p_t] = wA;
    if (stop_flag) return;   // (uniform over the workgroup)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    U_STAMP(3)
    // ActNorm a = x e^s + t: g_s = sum_rows g_a x e^s = sum g_a a - t sum g_a (+ the log-det term), g_t = sum_rows g_a
    float an_g = 0.f;
    if (an) {
        const int d = tid < ts.p_t ? tid : tid - ts.p_t;
        float s0 = 0.f, s1 = 0.f;
        for (int tj = 0; tj < nt; ++tj) { s0 += anp[tj * 64 + d]; s1 += anp[256 + tj * 64 + d]; }
        an_g = tid < ts.p_t ? s1 - tval[d] * s0 + a.ldw : s0;
        if (!stepping) a.grad_out[base + tid] = an_g;
    }
    if (!stepping) return;
    // dLoss/dL = tril(G Um^T, -1) and dLoss/d(Um) = triu(Lm^T G): two D^3 products on the matrix cores, one 16x16 output tile per wave
    // and round, left where the new values will go (LmN / UmN)
    // Only the tiles that hold a kept element are computed (dL: on or below the diagonal, dUm: on or above it -- nt (nt + 1) of the
    // 2 nt^2), and only the k-steps where the triangular operand is not structurally zero (Um[c][k] = 0 for k < c, Lm[k][c] = 0 for
    // k < c): the heads are one compute unit's matrix throughput (sixteen waves on four matrix cores), half of these products is zeros.
    // Live tile q of a triangle, rows first: (ti, tj) with tj <= ti for dL, mirrored for dUm; dealt to the waves longest first.
    const int KP = 16 * nt, ntri = nt * (nt + 1) / 2;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int tile = wave + u * nwv;
        if (tile >= 2 * ntri) continue;
        const bool isU = tile >= ntri;
        int q = isU ? tile - ntri : tile, ta = 0;
        while (q > ta) { q -= ta + 1; ++ta; }   // q-th pair (ta, q) with q <= ta
        const int ti = isU ? q : ta, tj = isU ? ta : q;
        const int i = 16 * ti + li, j = 16 * tj + li;
        // C[r][c] = sum_k G[r][k] Um[c][k] (k >= 16 tj)   |   C[c][j] = sum_k Lm[k][c] G[k][j] (k >= 16 ti)
        const float *pa = isU ? Lm + lk * DP + i : G + i * DP + lk, *pb = isU ? G + lk * DP + j : Um + j * DP + lk;
        const int sa = isU ? 4 * DP : 4, sb = isU ? 4 * DP : 4, q0 = 4 * (isU ? ti : tj);
        float fa[16], fb[16];   // (all LDS reads of the product up front: x_dim <= 64 is at most 16 k-steps)
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) { const int qc = (qq >= q0 && 4 * qq < KP) ? qq : q0; fa[qq] = pa[qc * sa]; fb[qq] = pb[qc * sb]; }
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) if (qq >= q0 && 4 * qq < KP) acc = mfma4(fa[qq], fb[qq], acc);
        float *dst = isU ? UmN : LmN;
        const int col = 16 * tj + li;
        dst[(16 * ti + 4 * lk + 0) * DP + col] = acc.x; dst[(16 * ti + 4 * lk + 1) * DP + col] = acc.y;
        dst[(16 * ti + 4 * lk + 2) * DP + col] = acc.z; dst[(16 * ti + 4 * lk + 3) * DP + col] = acc.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    F_STAMP(0)
    // the Adam step, element by element in the layout the parameters were loaded in; the new dense operands replace the gradients
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = hrow + 16 * u;
        if (row < D && hcol < D) {
            const int o = row * D + hcol;
            const float gL = hcol < row ? LmN[row * DP + hcol] : 0.f;
            const float gUraw = UmN[row * DP + hcol], gU = row < hcol ? gUraw : 0.f;
            const float nL = splr_adam_one(wL[u], gL, mL[u], vL[u], a.step_size, a.inv_bc2s, a.wd);
            const float nU = splr_adam_one(wU[u], gU, mU[u], vU[u], a.step_size, a.inv_bc2s, a.wd);
            a.m[base + ts.p_L + o] = mL[u]; a.v[base + ts.p_L + o] = vL[u]; a.w[base + ts.p_L + o] = nL;
            a.m[base + ts.p_U + o] = mU[u]; a.v[base + ts.p_U + o] = vU[u]; a.w[base + ts.p_U + o] = nU;
            LmN[row * DP + hcol] = hcol < row ? nL : (hcol == row ? 1.f : 0.f);
            float nUm = row < hcol ? nU : 0.f;
            if (row == hcol) {   // S[col]: its gradient is the diagonal of the second product + the conv's log-det term
                const float nS = splr_adam_one(wS, gUraw + a.ldw * __builtin_amdgcn_rcpf(wS), mS, vS, a.step_size, a.inv_bc2s, a.wd);
                a.m[sidx] = mS; a.v[sidx] = vS; a.w[sidx] = nS;
                snew[hcol] = nS;
                nUm = nS;
            }
            UmN[row * DP + hcol] = nUm;
        }
    }
    if (an) {
        const float wn = splr_adam_one(wA, an_g, mA, vA, a.step_size, a.inv_bc2s, a.wd);
        a.m[aidx] = mA; a.v[aidx] = vA; a.w[aidx] = wn;
        if (tid < ts.p_t) sact[tid] = wn;  // (the new s, for the log-det constant)
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    U_STAMP(4)
    // W = (P Lm) Um from the new values, and its transpose (the gradient kernel reads both row-major)
    float *Wm = a.wmat + (size_t)b * D * D, *WmT = a.wmatT + (size_t)b * D * D;
    if (wave < nt * nt) {
        const int ti = splr_div_small(wave, nt), tj = wave - ti * nt, i = 16 * ti + li, j = 16 * tj + li;
        const int pr = i < D ? spi[i] : 63;
        const float *pa = LmN + pr * DP + lk, *pb = UmN + lk * DP + j;
        const int q1 = 4 * (tj + 1);   // Um[k][j] = 0 for k > j: the k-steps past the column tile's last row are zeros
        float fa[16], fb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int qq = q < q1 ? q : 0; fa[q] = pa[4 * qq]; fb[q] = pb[4 * qq * DP]; }
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q) if (q < q1) acc = mfma4(fa[q], fb[q], acc);
        const float cv[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ti + 4 * lk + r, col = 16 * tj + li;
            if (row < D && col < D) { Wm[row * D + col] = cv[r]; WmT[col * D + row] = cv[r]; }
        }
    }
    U_STAMP(5)
    // log|det| of ActNorm + conv (networks.py:650, :676): the terms in parallel, summed in d order
    if (wave == nwv - 1) {   // (the last wave: its product tile, if it has one, is done)
        float t = lane < D ? sact[lane] + logf(fabsf(snew[lane])) : 0.f;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc += splr_rl(t, d);
        if (lane == 0) a.ldc[b] = acc;
    }
#ifdef NNEST_STAMP
    U_STAMP(6)
    // (device printf from this kernel's 1024-thread workgroups does not come out on this image: the gradient kernel of the NEXT launch
    // prints what is left here, behind the log-det constants)
    if (tid == 0 && b == 0) {
        for (int i = 0; i < 6; ++i) a.ldc[4 + i] = (float)(u_t[i + 1] - u_t[i]);
        a.ldc[11] = (float)(u_f[0] - u_t[3]);
    }
#endif
}

// W^T and the log-det constants from the current packed weights / W (start of a training call, and the gradient-only entry)
__global__ void splr_init_kernel(const float *__restrict__ w, const float *__restrict__ wmat, float *__restrict__ wmatT, float *__restrict__ ldc,
                                 SplTrainShape ts, const int *__restrict__ stop) {
    if (stop && *stop) return;
    const int D = ts.s.D, n = ts.s.B * D * D;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const int b = idx / (D * D), i = (idx / D) % D, j = idx % D;
        wmatT[(size_t)b * D * D + (size_t)j * D + i] = wmat[idx];
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ts.s.B) {
        const float *pb = w + (size_t)threadIdx.x * ts.s.blk_params;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc += pb[ts.p_s + d] + logf(fabsf(pb[ts.p_S + d]));
        ldc[threadIdx.x] = acc;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
bool spline_rows_eligible(const SplineShape &s, int batch) {
    static const bool off = [] { const char *e = getenv("NNEST_SPL_ROWS"); return e && !strcmp(e, "0"); }();
    return !off && s.H == 16 && s.K == SPL_K && s.D >= 2 && s.D <= 64 && batch >= 1 && batch <= 128 && s.B >= 1 && s.B <= 4;
}

static int rows_waves(const SplineShape &s) { const int n = 8 * (s.nl > s.nu ? s.nl : s.nu); return n <= 64 ? 1 : (n <= 128 ? 2 : 4); }

static size_t rows_grad_lds(const SplineShape &s, const SplRowsLayout &lay) {
    const int NW = rows_waves(s);
    return (size_t)(64 + 64 + NW * 64 + 4 * NW * 16 + 2 * s.B * 48 + lay.grow + 2 * s.B * 64 * NW * 16) * sizeof(float);
}

void spline_rows_free(nnest_spline *h) {
    SplRowsState *r = (SplRowsState *)h->rows;
    if (!r) return;
    (void)hipFree(r->stg); (void)hipFree(r->wmatT); (void)hipFree(r->ldc); (void)hipFree(r->rowlp);
    delete r;
    h->rows = nullptr;
}

// buffers for minibatches of up to `max_rows` rows (+ `valid_rows` forward-only rows in the same launch); W^T and the log-det constants
// from the current w_dev / wmat (queued on `st` behind whatever wrote them)
int spline_rows_prepare(nnest_spline *h, const SplTrainShape &ts, int max_rows, int valid_rows, hipStream_t st, const int *stop) {
    const SplineShape &s = h->s;
    SplRowsState *r = (SplRowsState *)h->rows;
    if (!r) {
        r = new SplRowsState();
        memset(r, 0, sizeof(*r));
        h->rows = r;
        SHIP_TRY(hipMalloc((void **)&r->wmatT, (size_t)s.B * s.D * s.D * sizeof(float) + SPL_W_SLACK_BYTES));   // (zeroed slack, as wmat's)
        SHIP_TRY(hipMemsetAsync(reinterpret_cast<char *>(r->wmatT) + (size_t)s.B * s.D * s.D * sizeof(float), 0, SPL_W_SLACK_BYTES, st));
        SHIP_TRY(hipMalloc((void **)&r->ldc, 32 * sizeof(float)));
    }
    if (max_rows < 128) max_rows = 128;   // (the contractions read 4 x 26 or 4 x 32 rows whatever the minibatch)
    if (max_rows > r->lay.rows_cap) {
        if (r->stg) (void)hipFree(r->stg);
        r->stg = nullptr;
        SplRowsLayout &l = r->lay;
        l.rows_cap = max_rows;
        const int nmax = s.nl > s.nu ? s.nl : s.nu;
        l.grow = ((SPL_P * nmax + 15) / 16) * 16;
        l.off_T = 0;
        l.off_G = l.off_T + (size_t)2 * s.B * l.rows_cap * SPLR_TROW;
        l.off_V = l.off_G + (size_t)2 * s.B * l.rows_cap * l.grow;
        l.total = l.off_V + (size_t)3 * s.B * l.rows_cap * SPLR_VROW;
        SHIP_TRY(hipMalloc((void **)&r->stg, l.total * sizeof(float)));
        SHIP_TRY(hipMemsetAsync(r->stg, 0, l.total * sizeof(float), st));   // (columns past a conditioner's outputs are read by the contraction tiles and discarded)
        if (r->rowlp) (void)hipFree(r->rowlp);
        r->rowlp = nullptr; r->valid_cap = 0;
    }
    if (!r->rowlp || valid_rows > r->valid_cap) {
        if (r->rowlp) (void)hipFree(r->rowlp);
        r->rowlp = nullptr;
        SHIP_TRY(hipMalloc((void **)&r->rowlp, (size_t)(r->lay.rows_cap + valid_rows + 4) * sizeof(float)));
        r->valid_cap = valid_rows;
    }
    if (rows_grad_lds(s, r->lay) > (size_t)160 * 1024) return spline_fail(NNEST_E_UNSUPPORTED, "rows form: LDS");
    hipLaunchKernelGGL(splr_init_kernel, dim3(32), dim3(256), 0, st, h->w_dev, h->wmat, r->wmatT, r->ldc, ts, stop);
    SHIP_TRY(hipGetLastError());
    return NNEST_OK;
}

float *spline_rows_rowlp(nnest_spline *h) { return ((SplRowsState *)h->rows)->rowlp; }

hipError_t spline_rows_grad(nnest_spline *h, const SplTrainShape &ts, const SplRowsBatch &bt, hipStream_t st) {
    SplRowsState *r = (SplRowsState *)h->rows;
    SplRowsArgs a;
    memset(&a, 0, sizeof(a));
    a.w = h->w_dev; a.wmat = h->wmat; a.wmatT = r->wmatT; a.ldc = r->ldc; a.ts = ts;
    a.x = bt.x; a.perm = bt.perm; a.M = bt.M; a.mtot = bt.mtot; a.noise = bt.noise; a.seed = bt.seed; a.noise_row0 = bt.noise_row0;
    a.epoch = bt.epoch; a.jitter = bt.jitter; a.xv = bt.xv; a.Mv = bt.Mv; a.rowlp = r->rowlp; a.stg = r->stg; a.lay = r->lay; a.stop = bt.stop;

    const size_t ldsb = rows_grad_lds(h->s, r->lay);
    const int NW = rows_waves(h->s), grid = bt.M + bt.Mv;
    hipError_t e = hipSuccess;
    const int nmax = h->s.nl > h->s.nu ? h->s.nl : h->s.nu, tiles = (SPL_P * nmax + 15) / 16, ns = (tiles + NW - 1) / NW;
#define SPLR_LAUNCH(NWv, NSv)                                                                                                              \
    {                                                                                                                                      \
        static bool attr[64] = {false};   /* (per device: the attribute belongs to the device's copy of the function) */                  \
        const int dv = h->device & 63;                                                                                                     \
        if (!attr[dv]) { e = hipFuncSetAttribute((const void *)splr_grad_kernel<NWv, NSv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[dv] = true; } \
        if (e == hipSuccess) hipLaunchKernelGGL((splr_grad_kernel<NWv, NSv>), dim3(grid), dim3(64 * NWv), ldsb, st, a);                    \
    }
    if (NW == 1) SPLR_LAUNCH(1, 12)
    else if (NW == 2 && ns <= 8) SPLR_LAUNCH(2, 8)
    else if (NW == 2) SPLR_LAUNCH(2, 12)
    else if (ns <= 9) SPLR_LAUNCH(4, 9)
    else SPLR_LAUNCH(4, 12)
#undef SPLR_LAUNCH
    return e != hipSuccess ? e : hipGetLastError();
}

hipError_t spline_rows_update(nnest_spline *h, const SplTrainShape &ts, const SplRowsStep &u, hipStream_t st) {
    SplRowsState *r = (SplRowsState *)h->rows;
    const SplineShape &s = h->s;
    SplRowsUpdArgs a;
    memset(&a, 0, sizeof(a));
    a.w = h->w_dev; a.m = h->adam_m; a.v = h->adam_v; a.stg = r->stg; a.lay = r->lay; a.rowlp = r->rowlp;
    a.pi = h->pi_dev; a.pi_inv = h->pi_dev + s.B * s.D; a.wmat = h->wmat; a.wmatT = r->wmatT; a.ldc = r->ldc; a.ts = ts; a.M = u.M;
    a.step_size = u.step_size; a.inv_bc2s = u.inv_bc2s; a.wd = u.wd; a.ldw = u.ldw; a.loss_out = u.loss_out; a.loss_scale = u.loss_scale;
    a.stop = u.stop; a.grad_out = u.grad_out; a.gwsum_out = u.gwsum_out;
    a.n_jobs = s.B * (splr_jobs_of(s, 0) + splr_jobs_of(s, 1));
    const size_t ldsb = ((size_t)5 * 64 * 65 + 3 * 64 + 512) * sizeof(float) + 64 * sizeof(int);
    static bool attr[64] = {false};   // (per device)
    hipError_t e = hipSuccess;
    if (!attr[h->device & 63]) {
        e = hipFuncSetAttribute((const void *)splr_update_kernel<26>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)splr_update_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr[h->device & 63] = true;
    }
    if (e != hipSuccess) return e;
    const dim3 grid(s.B + (a.n_jobs + 1 + 15) / 16);
    static const bool s32 = getenv("NNEST_SPLR_STEPS32") != nullptr;   // (diagnostic)
    if (u.M <= 104 && !s32) hipLaunchKernelGGL(splr_update_kernel<26>, grid, dim3(1024), ldsb, st, a);
    else hipLaunchKernelGGL(splr_update_kernel<32>, grid, dim3(1024), ldsb, st, a);
    return hipGetLastError();
}

}  // namespace nnest
