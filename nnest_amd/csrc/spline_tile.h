// spline_tile.h -- device tile code of the neural-spline flow (reference nnest/networks.py:393-715,
// SingleSpeedSpline = [ActNorm, Invertible1x1Conv, NSF_CL] x num_blocks) for gfx950, one wave64 per 16 walkers.
//
// Same transposed-MFMA formulation as flow_tile.h (weights = A operand, walkers = B columns, v_mfma_f32_16x16x4_f32),
// with a layout chosen for the spline:
//   * the vector is held as its two CONTIGUOUS halves (NSF_CL splits lower = x[:, :nl], upper = x[:, nl:],
//     networks.py:578-581), each padded to NTh tiles of 16 slots;  slot (t, g, r) of a half  <->  dimension
//     j = 16 t + 4 r + g of that half, held by lane (g, w) register r of tile t for walker w.  With this numbering
//     k-step r of tile t feeds dims {16t + 4r + g : g = 0..3} to the MFMA, i.e. a tile is at once an accumulator
//     (C/D layout) and the B operand of the next product;
//   * ActNorm and the 1x1 convolution are folded (at image-build time, host side) into ONE affine map per block and
//     direction:  forward  z = x A_f + b_f  with A_f = diag(e^s) W, b_f = t W;   inverse  x = z A_b + b_b  with
//     A_b = W^-1 diag(e^-s), b_b = -t e^-s;  their log-determinant is a per-block constant;
//   * the last conditioner layer has (3K-1) = 23 outputs per transformed dimension.  Its rows are ordered so that one
//     "super-tile" of 6 MFMA tiles delivers to lane (g, w) all 23 spline parameters (+1 pad) of dimension 4s + g of
//     walker w in 24 registers: the rational-quadratic spline is then evaluated entirely in registers, with no
//     cross-lane traffic, for 4 dimensions x 16 walkers per wave at a time.
#pragma once
#include "flow_tile.h"

namespace nnest {

enum { SPL_K = 8, SPL_P = 3 * SPL_K - 1, SPL_QT = 6 };  // bins, conditioner outputs per dim, MFMA tiles per super-tile

struct SplineShape {
    int D, H, B, K;
    float tail;
    int nl, nu;  // dims in the lower / upper half (nl = nu + (D odd))
    int NTh;     // 16-slot tiles per half
    int NH;      // H / 16
    int SL, SU;  // super-tiles (4 dims each) of the lower / upper half
    // image (floats), per block: [aff_f | aff_b | cond f1 | cond f2 | ldconst(4)]
    int aff_floats;      // (2 NTh)^2 * 256 weights + 2 NTh * 16 bias
    int f1_floats, f2_floats;
    int blk_floats, image_floats;
    int blk_params, num_params;  // packed (state_dict order) parameter counts
    float base_beta, base_const;  // base distribution, as FlowShape
};

__host__ __device__ inline int spl_cond_hidden_floats(int NTh, int NH) {  // L1 + L2 + L3 + b1 b2 b3
    return NH * NTh * 256 + 2 * NH * NH * 256 + 3 * 16 * NH;
}
__host__ __device__ inline int spl_cond_floats(int NTh, int NH, int S) {
    return spl_cond_hidden_floats(NTh, NH) + S * SPL_QT * NH * 256 + S * SPL_QT * 16;
}

__device__ __forceinline__ f32x4 lrelu4(f32x4 v) {  // nn.LeakyReLU(0.2), networks.py:400
    f32x4 o;
    o.x = v.x > 0.f ? v.x : 0.2f * v.x; o.y = v.y > 0.f ? v.y : 0.2f * v.y;
    o.z = v.z > 0.f ? v.z : 0.2f * v.z; o.w = v.w > 0.f ? v.w : 0.2f * v.w;
    return o;
}

__device__ __forceinline__ float reg_of(const f32x4 &v, int r) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); }
__device__ __forceinline__ void set_reg(f32x4 &v, int r, float x) {
    if (r == 0) v.x = x; else if (r == 1) v.y = x; else if (r == 2) v.z = x; else v.w = x;
}

// ---- the folded ActNorm + 1x1 conv: out = in A + b over both halves ----------------------------------------------
// image: weights [to (2 NTh)][ti (2 NTh)][r][64] then bias [to][g][r]
template <int NTh>
__device__ __forceinline__ void spl_affine(const float *__restrict__ aff, int lane, const f32x4 (&in)[2][NTh], f32x4 (&out)[2][NTh]) {
    constexpr int T2 = 2 * NTh;
    const int g = lane >> 4;
    const float *bias = aff + T2 * T2 * 256;
#pragma unroll
    for (int to = 0; to < T2; ++to) {
        f32x4 acc0 = *reinterpret_cast<const f32x4 *>(bias + (to * 4 + g) * 4);
        f32x4 acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < T2; ++ti) {
            const f32x4 v = in[ti / NTh][ti % NTh];
            const float *a = aff + (size_t)((to * T2 + ti) * 4) * 64 + lane;
            acc0 = mfma4(a[0], v.x, acc0);
            acc1 = mfma4(a[64], v.y, acc1);
            acc0 = mfma4(a[128], v.z, acc0);
            acc1 = mfma4(a[192], v.w, acc1);
        }
        out[to / NTh][to % NTh] = acc0 + acc1;
    }
}

// ---- conditioner trunk (networks.py:393-409): Linear(n,H) LReLU Linear(H,H) LReLU Linear(H,H) LReLU -------------
// image: L1 [ht][t][r][64] | L2 [hto][hti][r][64] | L3 | b1[H] b2[H] b3[H]
template <int NTh, int NH>
__device__ __forceinline__ void spl_hidden(const float *__restrict__ net, int lane, const f32x4 (&in)[NTh], f32x4 (&h)[NH]) {
    const int g = lane >> 4;
    const float *L1 = net, *L2 = net + NH * NTh * 256, *L3 = L2 + NH * NH * 256, *b = L3 + NH * NH * 256;
    f32x4 acc[NH];
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) {
        f32x4 a0 = *reinterpret_cast<const f32x4 *>(b + 16 * ht + 4 * g), a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NTh; ++t) {
            const float *a = L1 + (size_t)((ht * NTh + t) * 4) * 64 + lane;
            a0 = mfma4(a[0], in[t].x, a0);
            a1 = mfma4(a[64], in[t].y, a1);
            a0 = mfma4(a[128], in[t].z, a0);
            a1 = mfma4(a[192], in[t].w, a1);
        }
        acc[ht] = lrelu4(a0 + a1);
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const float *Lw = l == 0 ? L2 : L3;
        const float *bl = b + 16 * NH * (l + 1);
        f32x4 nxt[NH];
#pragma unroll
        for (int hto = 0; hto < NH; ++hto) {
            f32x4 a0 = *reinterpret_cast<const f32x4 *>(bl + 16 * hto + 4 * g), a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hti = 0; hti < NH; ++hti) {
                const float *a = Lw + (size_t)((hto * NH + hti) * 4) * 64 + lane;
                a0 = mfma4(a[0], acc[hti].x, a0);
                a1 = mfma4(a[64], acc[hti].y, a1);
                a0 = mfma4(a[128], acc[hti].z, a0);
                a1 = mfma4(a[192], acc[hti].w, a1);
            }
            nxt[hto] = lrelu4(a0 + a1);
        }
#pragma unroll
        for (int hto = 0; hto < NH; ++hto) acc[hto] = nxt[hto];
    }
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) h[ht] = acc[ht];
}

// ---- rational-quadratic spline of one scalar (networks.py:425-556, through NSF_CL's :583-587) --------------------
// raw[24]: widths logits [0,8), heights logits [8,16), inner-derivative logits [16,23), pad.  Identity outside
// [-tail, tail].  The reference applies softmax twice to widths/heights (NSF_CL, scaled by 2B; then RQS) and
// softplus twice to the inner derivatives: restated as is.  Returns y; *ld += log|dy/dx|.
// v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sqrt_f32 (1 ulp each; quarter rate) instead of the correctly-rounded library
// sequences: the spline evaluation is VALU-bound (about 40 exponentials per dimension and direction), and with the
// library calls it cost ~6x more than the matrix work around it.  Measured in float64 arithmetic the passes stay at
// the 1e-6 level of the reference's own float32 (tests/diag_spline_err.py).
__device__ __forceinline__ float spl_exp(float v) { return __builtin_amdgcn_exp2f(v * 1.4426950408889634f); }
__device__ __forceinline__ float spl_log(float v) { return __builtin_amdgcn_logf(v) * 0.6931471805599453f; }
__device__ __forceinline__ float spl_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float spl_softplus(float v) { return v > 20.f ? v : spl_log(1.f + spl_exp(v)); }

__device__ __forceinline__ void spl_softmax8(const float (&in)[SPL_K], float (&out)[SPL_K]) {
    float mx = in[0];
#pragma unroll
    for (int k = 1; k < SPL_K; ++k) mx = fmaxf(mx, in[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) { out[k] = spl_exp(in[k] - mx); s += out[k]; }
    const float rs = spl_rcp(s);
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) out[k] = out[k] * rs;
}

// knots of one axis: unnormalised -> softmax -> min bin + cumsum -> [-tail, tail]; returns the K+1 edges and K sizes
__device__ __forceinline__ void spl_knots(const float (&logits)[SPL_K], float tail, float (&edge)[SPL_K + 1], float (&size)[SPL_K]) {
    float a[SPL_K], u[SPL_K], p[SPL_K];
    spl_softmax8(logits, a);
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) u[k] = 2.f * tail * a[k];
    spl_softmax8(u, p);
    float c = 0.f;
    edge[0] = -tail;
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) {
        float wk = 1e-3f + (1.f - 1e-3f * SPL_K) * p[k];
        c += wk;
        edge[k + 1] = (2.f * tail) * c + (-tail);
    }
    edge[SPL_K] = tail;
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) size[k] = edge[k + 1] - edge[k];
}

// the same for the two axes of a spline at once, {width, height} in the two halves of a float pair: the two constructions are the same
// instruction stream on different numbers, and on packed operands (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) half of it -- 120 of
// the ~350 instructions of an evaluation.  Operation for operation what spl_knots does (the exponentials, the reciprocals and the
// maxima stay scalar): the same values.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void spl_softmax8x2(const f32x2 (&in)[SPL_K], f32x2 (&out)[SPL_K]) {
    f32x2 mx = in[0];
#pragma unroll
    for (int k = 1; k < SPL_K; ++k) { mx.x = fmaxf(mx.x, in[k].x); mx.y = fmaxf(mx.y, in[k].y); }
    f32x2 s = (f32x2){0.f, 0.f};
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) {
        const f32x2 d = (in[k] - mx) * 1.4426950408889634f;
        out[k] = (f32x2){__builtin_amdgcn_exp2f(d.x), __builtin_amdgcn_exp2f(d.y)};
        s += out[k];
    }
    const f32x2 rs = (f32x2){spl_rcp(s.x), spl_rcp(s.y)};
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) out[k] = out[k] * rs;
}
// (a, p: the two softmax outputs, which the reverse mode of the training kernel needs again -- spl_knots_bwd2)
__device__ __forceinline__ void spl_knots2(const f32x2 (&logits)[SPL_K], float tail, f32x2 (&edge)[SPL_K + 1], f32x2 (&size)[SPL_K],
                                           f32x2 (&a)[SPL_K], f32x2 (&p)[SPL_K]) {
    f32x2 u[SPL_K];
    spl_softmax8x2(logits, a);
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) u[k] = (2.f * tail) * a[k];
    spl_softmax8x2(u, p);
    f32x2 c = (f32x2){0.f, 0.f};
    edge[0] = (f32x2){-tail, -tail};
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) {
        const f32x2 wk = 1e-3f + (1.f - 1e-3f * SPL_K) * p[k];
        c += wk;
        edge[k + 1] = (2.f * tail) * c + (-tail);
    }
    edge[SPL_K] = (f32x2){tail, tail};
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) size[k] = edge[k + 1] - edge[k];
}

__device__ __forceinline__ void spl_knots2(const f32x2 (&logits)[SPL_K], float tail, f32x2 (&edge)[SPL_K + 1], f32x2 (&size)[SPL_K]) {
    f32x2 a[SPL_K], p[SPL_K];
    spl_knots2(logits, tail, edge, size, a, p);
}

// inner-knot derivative k (1..K-1) from its logit: min_derivative + softplus(softplus(v)); the end knots are
// min_derivative + softplus(log(e^{1 - min_derivative} - 1)) = 1 (networks.py:436-439, :486)
__device__ __forceinline__ float spl_knot_deriv(const float (&ldv)[SPL_K - 1], int k) {
    float v = ldv[0];
#pragma unroll
    for (int i = 1; i < SPL_K - 1; ++i) v = (k == i + 1) ? ldv[i] : v;
    const float d = 1e-3f + spl_softplus(spl_softplus(v));
    return (k <= 0 || k >= SPL_K) ? 1.0f : d;
}

template <bool INV>
__device__ __forceinline__ float spl_rqs(const f32x4 (&raw)[SPL_QT], float tail, float x, float &ld) {
    float lw[SPL_K] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w, raw[1].x, raw[1].y, raw[1].z, raw[1].w};
    float lh[SPL_K] = {raw[2].x, raw[2].y, raw[2].z, raw[2].w, raw[3].x, raw[3].y, raw[3].z, raw[3].w};
    float ldv[SPL_K - 1] = {raw[4].x, raw[4].y, raw[4].z, raw[4].w, raw[5].x, raw[5].y, raw[5].z};
    const bool inside = x >= -tail && x <= tail;
    float cw[SPL_K + 1], wd[SPL_K], ch[SPL_K + 1], ht[SPL_K];
#ifdef SPL_KNOTS_SCALAR   // (the two axes one after the other: tools/spline_mh_probe.hip -DSPL_KNOTS_SCALAR, for the A/B)
    spl_knots(lw, tail, cw, wd);
    spl_knots(lh, tail, ch, ht);
#else
    {
        f32x2 l2[SPL_K], e2[SPL_K + 1], s2[SPL_K];
#pragma unroll
        for (int k = 0; k < SPL_K; ++k) l2[k] = (f32x2){lw[k], lh[k]};
        spl_knots2(l2, tail, e2, s2);
#pragma unroll
        for (int k = 0; k <= SPL_K; ++k) { cw[k] = e2[k].x; ch[k] = e2[k].y; }
#pragma unroll
        for (int k = 0; k < SPL_K; ++k) { wd[k] = s2[k].x; ht[k] = s2[k].y; }
    }
#endif
    // searchsorted (networks.py:417-422): edges <= x, last edge + 1e-6
    int bin = -1;
#pragma unroll
    for (int k = 0; k <= SPL_K; ++k) {
        float e = INV ? ch[k] : cw[k];
        if (k == SPL_K) e += 1e-6f;
        bin += (x >= e) ? 1 : 0;
    }
    bin = bin < 0 ? 0 : (bin > SPL_K - 1 ? SPL_K - 1 : bin);
    float icw = cw[0], ibw = wd[0], ich = ch[0], ih = ht[0];
#pragma unroll
    for (int k = 1; k < SPL_K; ++k) {
        const bool s = bin == k;
        icw = s ? cw[k] : icw; ibw = s ? wd[k] : ibw; ich = s ? ch[k] : ich; ih = s ? ht[k] : ih;
    }
    const float d0 = spl_knot_deriv(ldv, bin), d1 = spl_knot_deriv(ldv, bin + 1);
    const float ribw = spl_rcp(ibw);
    const float delta = ih * ribw;
    float out, lad;
    if (INV) {  // networks.py:515-539
        const float dx = x - ich, sdd = d0 + d1 - 2.f * delta;
        const float a = dx * sdd + ih * (delta - d0);
        const float b = ih * d0 - dx * sdd;
        const float c = -delta * dx;
        const float disc = b * b - 4.f * a * c;
        const float root = (2.f * c) * spl_rcp(-b - __builtin_amdgcn_sqrtf(disc));
        out = root * ibw + icw;
        const float tomt = root * (1.f - root);
        const float den = delta + sdd * tomt;
        const float num = delta * delta * (d1 * root * root + 2.f * delta * tomt + d0 * (1.f - root) * (1.f - root));
        lad = -(spl_log(num) - 2.f * spl_log(den));
    } else {  // networks.py:541-556
        const float theta = (x - icw) * ribw;
        const float tomt = theta * (1.f - theta);
        const float numer = ih * (delta * theta * theta + d0 * tomt);
        const float den = delta + (d0 + d1 - 2.f * delta) * tomt;
        out = ich + numer * spl_rcp(den);
        const float num = delta * delta * (d1 * theta * theta + 2.f * delta * tomt + d0 * (1.f - theta) * (1.f - theta));
        lad = spl_log(num) - 2.f * spl_log(den);
    }
    ld += inside ? lad : 0.f;
    return inside ? out : x;
}

// ---- one RQ-spline coupling: the `n_out` dims of `tr` are transformed, conditioned on `cond` ----------------------
// net image: hidden part (spl_hidden) | L4 [s][q][hti][r][64] | b4 [s][q][g][r]
// TEAM = 4 or 8: the waves of a workgroup hold the same tile; wave `wv` evaluates the super-tiles s = wv (mod TEAM) and the
// results are merged through `xch` ([TEAM][NTh][64] f32x4 of LDS).  The trunk is
// recomputed by every wave (16 MFMAs against 24 per super-tile).
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the wave's outstanding global loads and stores
// (a round trip to memory per team merge when stash / gradient stores are in flight)
__device__ __forceinline__ void spl_team_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NTh, int NH, bool INV, int TEAM = 1>
__device__ __forceinline__ float spl_coupling(const float *__restrict__ net, int S, int n_out, float tail, int lane,
                                              const f32x4 (&cond)[NTh], f32x4 (&tr)[NTh], int wv = 0, f32x4 *xch = nullptr) {
    const int g = lane >> 4;
    f32x4 h[NH];
    spl_hidden<NTh, NH>(net, lane, cond, h);
    const float *L4 = net + spl_cond_hidden_floats(NTh, NH);
    const float *b4 = L4 + (size_t)S * SPL_QT * NH * 256;
    float ld = 0.f;
#pragma unroll
    for (int s = 0; s < 4 * NTh; ++s) {
        if (s < S && (TEAM == 1 || (s & (TEAM - 1)) == wv)) {  // uniform over the wave
            f32x4 raw[SPL_QT];
#pragma unroll
            for (int q = 0; q < SPL_QT; ++q) {
                f32x4 acc = *reinterpret_cast<const f32x4 *>(b4 + ((s * SPL_QT + q) * 4 + g) * 4);
#pragma unroll
                for (int hti = 0; hti < NH; ++hti) {
                    const float *a = L4 + (size_t)(((s * SPL_QT + q) * NH + hti) * 4) * 64 + lane;
                    acc = mfma4(a[0], h[hti].x, acc);
                    acc = mfma4(a[64], h[hti].y, acc);
                    acc = mfma4(a[128], h[hti].z, acc);
                    acc = mfma4(a[192], h[hti].w, acc);
                }
                raw[q] = acc;
            }
            // lane (g, w): dimension 4 s + g of the transformed half = slot (t = s >> 2, g, r = s & 3)
            const float x = reg_of(tr[s >> 2], s & 3);
            float l = 0.f;
#ifdef PROBE_NOEVAL   // (tools/spline_inv_probe.hip -DPROBE_NOEVAL: the inverse without its spline arithmetic -- never defined in the library)
            const float y = x + raw[0].x + raw[1].y + raw[2].z + raw[3].w + raw[4].x + raw[5].y;
#else
            const float y = spl_rqs<INV>(raw, tail, x, l);
#endif
            const bool valid = 4 * s + g < n_out;
            set_reg(tr[s >> 2], s & 3, valid ? y : 0.f);
            ld += valid ? l : 0.f;
        }
    }
    if (TEAM > 1) {  // super-tile s = 4t + r (register r of tile t) comes from wave s mod TEAM
#pragma unroll
        for (int t = 0; t < NTh; ++t) xch[(wv * NTh + t) * 64 + lane] = tr[t];
        spl_team_barrier();
#pragma unroll
        for (int t = 0; t < NTh; ++t)
            tr[t] = (f32x4){xch[(((4 * t + 0) & (TEAM - 1)) * NTh + t) * 64 + lane].x, xch[(((4 * t + 1) & (TEAM - 1)) * NTh + t) * 64 + lane].y,
                            xch[(((4 * t + 2) & (TEAM - 1)) * NTh + t) * 64 + lane].z, xch[(((4 * t + 3) & (TEAM - 1)) * NTh + t) * 64 + lane].w};
        spl_team_barrier();
    }
    return ld;
}

// ---- the two halves of the 16 columns (8-row tiles: lanes w and w ^ 8 carry the same row / walker) --------------------------
// value of the partner lane w ^ 8 (DPP row_ror:8 inside the 16-lane row)
__device__ __forceinline__ float half_swap(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}
__device__ __forceinline__ f32x4 half_swap4(f32x4 v) { return (f32x4){half_swap(v.x), half_swap(v.y), half_swap(v.z), half_swap(v.w)}; }
__device__ __forceinline__ f32x4 sel4(bool c, f32x4 a, f32x4 b) { return (f32x4){c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w}; }

// Values made opaque to the optimiser at the top of every block iteration of the TRAINING kernel.  The unrolled coupling
// bodies derive dozens of masks and offsets from the lane index and the half sizes; all of them are invariant over the
// blocks, so they were hoisted out of the loops, spilled (250-500 SGPRs, VGPR copies in scratch) and re-read one by one on
// every iteration: over the three blocks of a gradient pass that cost more than recomputing them in place (1.3x on the
// epoch).  The MH kernels run the same body 750 times per launch and keep the hoisting (made opaque there: 8.0 -> 11.6 ms).
__device__ __forceinline__ int spl_opaque_s(int v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ int spl_opaque_v(int v) { asm volatile("" : "+v"(v)); return v; }

// ---- the stack -----------------------------------------------------------------------------------------------
// xs[0] = lower half tiles, xs[1] = upper half tiles.  Returns this lane's log-det partial (sum over the 4 lanes of a
// walker = the row's log-det); the per-block constants are added on lane group 0.
template <int NTh, int NH, int TEAM = 1>
__device__ __forceinline__ float spline_forward_tile(const float *__restrict__ img, const SplineShape &s, int lane, f32x4 (&xs)[2][NTh],
                                                     int wv = 0, f32x4 *xch = nullptr) {
    float ld = 0.f;
    for (int b = 0; b < s.B; ++b) {
        const int nu = s.nu, nl = s.nl, SU = s.SU, SL = s.SL;
        const float *blk = img + (size_t)b * s.blk_floats;
        f32x4 y[2][NTh];
        spl_affine<NTh>(blk, lane, xs, y);
        const float *f1 = blk + 2 * s.aff_floats, *f2 = f1 + s.f1_floats;
        ld += spl_coupling<NTh, NH, false, TEAM>(f1, SU, nu, s.tail, lane, y[0], y[1], wv, xch);   // upper | lower  (networks.py:582-588)
        ld += spl_coupling<NTh, NH, false, TEAM>(f2, SL, nl, s.tail, lane, y[1], y[0], wv, xch);   // lower | new upper (:589-598)
        if (lane < 16 && wv == 0) ld += (f2 + s.f2_floats)[0];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < NTh; ++t) xs[c][t] = y[c][t];
    }
    return ld;
}

template <int NTh, int NH, int TEAM = 1>
__device__ __forceinline__ float spline_inverse_tile(const float *__restrict__ img, const SplineShape &s, int lane, f32x4 (&xs)[2][NTh],
                                                     int wv = 0, f32x4 *xch = nullptr) {
    float ld = 0.f;
    for (int b = s.B - 1; b >= 0; --b) {
        const int nu = s.nu, nl = s.nl, SU = s.SU, SL = s.SL;
        const float *blk = img + (size_t)b * s.blk_floats;
        const float *f1 = blk + 2 * s.aff_floats, *f2 = f1 + s.f1_floats;
        ld += spl_coupling<NTh, NH, true, TEAM>(f2, SL, nl, s.tail, lane, xs[1], xs[0], wv, xch);  // networks.py:605-614
        ld += spl_coupling<NTh, NH, true, TEAM>(f1, SU, nu, s.tail, lane, xs[0], xs[1], wv, xch);  // :615-621
        f32x4 y[2][NTh];
        spl_affine<NTh>(blk + s.aff_floats, lane, xs, y);
        if (lane < 16 && wv == 0) ld -= (f2 + s.f2_floats)[0];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < NTh; ++t) xs[c][t] = y[c][t];
    }
    return ld;
}

// ---- layout exchange with the row-major / parity-class tiles of flow_tile.h (one wave, through LDS) ----------------
// buf: 16 rows x (D + 1) floats owned by this wave
template <int NT>
__device__ __forceinline__ void spl_from_parity(float *buf, int D, int nl, int lane, const f32x4 (&xs)[2][NT], f32x4 (&sp)[2][NT]) {
    const int w = lane & 15, g = lane >> 4;
    float *row = buf + w * (D + 1);
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        float v[8] = {xs[0][tau].x, xs[1][tau].x, xs[0][tau].y, xs[1][tau].y, xs[0][tau].z, xs[1][tau].z, xs[0][tau].w, xs[1][tau].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = 32 * tau + 8 * g + j;
            if (d < D) row[d] = v[j];
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes are visible to its own lanes
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * t + 4 * r + g, n = hf ? D - nl : nl;
                v[r] = j < n ? row[(hf ? nl : 0) + j] : 0.f;
            }
            sp[hf][t] = (f32x4){v[0], v[1], v[2], v[3]};
        }
    __builtin_amdgcn_wave_barrier();
}

template <int NT>
__device__ __forceinline__ void spl_to_parity(float *buf, int D, int nl, int lane, const f32x4 (&sp)[2][NT], f32x4 (&xs)[2][NT]) {
    const int w = lane & 15, g = lane >> 4;
    float *row = buf + w * (D + 1);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float v[4] = {sp[hf][t].x, sp[hf][t].y, sp[hf][t].z, sp[hf][t].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * t + 4 * r + g, n = hf ? D - nl : nl;
                if (j < n) row[(hf ? nl : 0) + j] = v[r];
            }
        }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = 32 * tau + 8 * g + j;
            v[j] = d < D ? row[d] : 0.f;
        }
        xs[0][tau] = (f32x4){v[0], v[2], v[4], v[6]};
        xs[1][tau] = (f32x4){v[1], v[3], v[5], v[7]};
    }
    __builtin_amdgcn_wave_barrier();
}

}  // namespace nnest
