// flow_tile.h -- device-side building blocks of the RealNVP coupling stack on gfx950 (CDNA4).
//
// Data layout (DESIGN.md "Data layout").  One wave64 owns a tile of 16 walkers.  The D coordinates
// of a walker are split by parity class (class 0 = even dims, 1 = odd dims: the two sides of the
// reference's alternating mask, nnest/networks.py:333-334, :346) into "slots": dim d = 2*slot + class.
// A class is padded to 16*NT slots.  A [16 slots x 16 walkers] sub-tile lives in the accumulator
// (C/D) layout of v_mfma_f32_16x16x4_f32:
//       lane = 16*g + w  (g = 0..3, w = walker 0..15),  register r = 0..3   <->   slot 16*tau + 4*g + r
// so lane (g,w) holds dims [32*tau + 8*g, 32*tau + 8*g + 8) of walker w: 8 consecutive floats of its row.
//
// Every layer is computed transposed, Out^T[feature][walker] = W[feature][k] * In^T[k][walker], with the
// weights as the MFMA A operand and the activations as the B operand.  The C/D layout of one layer's
// output is then directly the B operand of the next layer's k-steps (k-step r takes register r; lane
// group g supplies k = 4g + r), with the weight fragments permuted to match at repack time.  No LDS
// traffic, no cross-lane movement between layers.
//
// Weight fragment image (built by repack_fragments_kernel, one image per flow), per (block, net):
//   L1   [ht][tau][r][64 lanes]          A-fragments of Linear(D,H) restricted to the conditioning class
//   L2   [l][hto][hti][r][64 lanes]      A-fragments of the L hidden Linear(H,H)
//   L3   [tau][ht][r][64 lanes]          A-fragments of Linear(H,D) restricted to the transformed class
//   b1 [16*NH]  b2 [l][16*NH]  b3 [16*NT]   biases (b3 in slot order, zero on padded slots)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nnest {

typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { FLOW_KIND_NVP = 0, FLOW_KIND_MAF = 1 };

struct FlowShape {
    int D, H, B, L;
    int NT;  // 16-slot tiles per parity class: ceil(ceil(D/2)/16)
    int NH;  // H / 16
    int net_floats;    // floats per (block, net) in the fragment image
    int image_floats;  // B * 2 * net_floats
    int net_params;    // packed parameter count per net (state_dict layout)
    // SingleSpeedNVP scale variant (networks.py:328-347): 0 '' (affine), 1 'translate' (translate-only couplings),
    // 2 'constant' (translate-only couplings + one ScaleLayer scalar after each).  In modes 1 and 2 the scale_net
    // slots of the packed vector stay zero (log_s = 0 exactly) and receive no gradient; in mode 2 the B scalars
    // follow the B blocks in the packed vector and the image_floats of the fragment image.
    int scale_mode;
    // base distribution of the flow (NormalizingFlowModel.prior): base_beta = 0 is N(0, I) (networks.py:51-57); > 0 the
    // reference's GeneralisedNormal(0, 1, beta) (nnest/distributions/generalised_normal.py:66-71).  base_const = the
    // per-dimension constant of its log density.
    float base_beta, base_const;
    // which flow the handle is: FLOW_KIND_NVP the reference's SingleSpeedNVP; FLOW_KIND_MAF the build-defined masked autoregressive
    // flow (maf_tile.h: same parameter layout; NT stays the tiles per parity class, the fragment image is laid out for 2 NT
    // tiles; G = groups of the sequential inverse)
    int kind, G;
    __host__ __device__ int nets_params() const { return B * 2 * net_params; }
    __host__ __device__ int num_params() const { return B * 2 * net_params + (scale_mode == 2 ? B : 0); }
    __host__ __device__ int image_total() const { return image_floats + (scale_mode == 2 ? B : 0); }
};

__host__ __device__ inline int frag_off_L1() { return 0; }
__host__ __device__ inline int frag_off_L2(int NT, int NH) { return NH * NT * 4 * 64; }
__host__ __device__ inline int frag_off_L3(int NT, int NH, int L) { return frag_off_L2(NT, NH) + L * NH * NH * 4 * 64; }
__host__ __device__ inline int frag_off_b1(int NT, int NH, int L) { return frag_off_L3(NT, NH, L) + NT * NH * 4 * 64; }
__host__ __device__ inline int frag_off_b2(int NT, int NH, int L) { return frag_off_b1(NT, NH, L) + 16 * NH; }
__host__ __device__ inline int frag_off_b3(int NT, int NH, int L) { return frag_off_b2(NT, NH, L) + L * 16 * NH; }
__host__ __device__ inline int frag_net_floats(int NT, int NH, int L) { return frag_off_b3(NT, NH, L) + 16 * NT; }

// ---- elementwise helpers ------------------------------------------------------------------------
// tanh(x) = 1 - 2/(e^{2x}+1): v_exp_f32 + v_rcp_f32; abs error ~1e-7 (inputs to a linear layer, so
// absolute accuracy is what matters).  Saturates correctly for |x| large; NaN propagates.
__device__ __forceinline__ float fast_tanh(float x) {
    float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // e^{2x} = 2^{x * 2 log2(e)}: one multiply + v_exp_f32
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

template <int ACT>  // 0 = tanh (scale_net, networks.py:271-276), 1 = relu (translate_net, networks.py:278-282)
__device__ __forceinline__ f32x4 activate(f32x4 v) {
    f32x4 o;
    if (ACT == 0) {
        o.x = fast_tanh(v.x); o.y = fast_tanh(v.y); o.z = fast_tanh(v.z); o.w = fast_tanh(v.w);
    } else {
        o.x = fmaxf(v.x, 0.f); o.y = fmaxf(v.y, 0.f); o.z = fmaxf(v.z, 0.f); o.w = fmaxf(v.w, 0.f);
    }
    return o;
}

// ---- base density ------------------------------------------------------------------------------------
// log p(u) = -sum_d E(u_d) + D * base_const with E(u) = u^2 / 2 (beta = 0) or |u|^beta; base_dE = dE/du
__device__ __forceinline__ float base_pow(float a, float beta) { return __builtin_amdgcn_exp2f(beta * __builtin_amdgcn_logf(a)); }
__device__ __forceinline__ float base_E(float u, float beta) { return beta == 0.f ? 0.5f * u * u : base_pow(fabsf(u), beta); }
__device__ __forceinline__ float base_dE(float u, float beta) {
    if (beta == 0.f) return u;
    return u == 0.f ? 0.f : beta * base_pow(fabsf(u), beta) / u;  // beta |u|^(beta-1) sign(u)
}
__device__ __forceinline__ float base_E4(f32x4 v, float beta) { return (base_E(v.x, beta) + base_E(v.y, beta)) + (base_E(v.z, beta) + base_E(v.w, beta)); }
__device__ __forceinline__ f32x4 base_dE4(f32x4 v, float beta) { return (f32x4){base_dE(v.x, beta), base_dE(v.y, beta), base_dE(v.z, beta), base_dE(v.w, beta)}; }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Fragment image addressing.  The four K-step fragments (r = 0..3) of one (output tile, input tile) pair -- fragment numbers
// 4q .. 4q+3 -- sit together per lane: element (fragment idx, lane) is at ((idx >> 2) * 64 + lane) * 4 + (idx & 3) of its
// (block, net) region, so one 16-byte read (ds_read_b128 from the LDS copy) feeds four MFMAs instead of four 4-byte reads.
__host__ __device__ inline int frag_elem(int idx, int lane) { return (((idx >> 2) * 64 + lane) << 2) + (idx & 3); }
__device__ __forceinline__ f32x4 frag_quad(const float *__restrict__ region_lane /* region + 4 * lane */, int q) {
    return *reinterpret_cast<const f32x4 *>(region_lane + q * 256);
}

// One MLP of a coupling layer (networks.py:271-282) on a 16-walker tile.
//   wn   : base of this (block, net) in the fragment image (LDS or global)
//   in   : conditioning-class tiles; out: transformed-class tiles (ls or t), both in C/D layout
template <int NT, int NH, int ACT>
__device__ __forceinline__ void mlp_tile(const float *__restrict__ wn, int L, int lane, const f32x4 (&in)[NT],
                                         f32x4 (&out)[NT]) {
    const int g4 = (lane >> 4) * 4;
    const float *fL1 = wn + frag_off_L1() + 4 * lane;
    const float *fL2 = wn + frag_off_L2(NT, NH) + 4 * lane;
    const float *fL3 = wn + frag_off_L3(NT, NH, L) + 4 * lane;
    const float *b1 = wn + frag_off_b1(NT, NH, L) + g4;
    const float *b2 = wn + frag_off_b2(NT, NH, L) + g4;
    const float *b3 = wn + frag_off_b3(NT, NH, L) + g4;
    f32x4 h[NH];
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(b1 + 16 * ht);
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            const f32x4 f = frag_quad(fL1, ht * NT + tau);
            acc = mfma4(f.x, in[tau].x, acc);
            acc = mfma4(f.y, in[tau].y, acc);
            acc = mfma4(f.z, in[tau].z, acc);
            acc = mfma4(f.w, in[tau].w, acc);
        }
        h[ht] = activate<ACT>(acc);
    }
    for (int l = 0; l < L; ++l) {
        f32x4 h2[NH];
#pragma unroll
        for (int hto = 0; hto < NH; ++hto) {
            f32x4 acc = *reinterpret_cast<const f32x4 *>(b2 + (l * NH + hto) * 16);
#pragma unroll
            for (int hti = 0; hti < NH; ++hti) {
                const f32x4 f = frag_quad(fL2, (l * NH + hto) * NH + hti);
                acc = mfma4(f.x, h[hti].x, acc);
                acc = mfma4(f.y, h[hti].y, acc);
                acc = mfma4(f.z, h[hti].z, acc);
                acc = mfma4(f.w, h[hti].w, acc);
            }
            h2[hto] = activate<ACT>(acc);
        }
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) h[ht] = h2[ht];
    }
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(b3 + 16 * tau);
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
            const f32x4 f = frag_quad(fL3, tau * NH + ht);
            acc = mfma4(f.x, h[ht].x, acc);
            acc = mfma4(f.y, h[ht].y, acc);
            acc = mfma4(f.z, h[ht].z, acc);
            acc = mfma4(f.w, h[ht].w, acc);
        }
        out[tau] = acc;
    }
}

// CouplingLayer.forward / inverse on a tile (networks.py:289-309).  `cond` passes through untouched
// (bit-exact, which is what makes the inverse exact); `trans` is updated in place.  Returns this
// lane's partial of the block's log-det (sum over its 4*NT slots; padded slots contribute exactly 0).
template <int NT, int NH, bool INVERSE>
__device__ __forceinline__ float coupling_tile(const float *__restrict__ wblk, int net_floats, int L, int lane,
                                               const f32x4 (&cond)[NT], f32x4 (&trans)[NT]) {
    f32x4 ls[NT], t[NT];
    mlp_tile<NT, NH, 0>(wblk, L, lane, cond, ls);
    mlp_tile<NT, NH, 1>(wblk + net_floats, L, lane, cond, t);
    float ld = 0.f;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        if (!INVERSE) {  // inputs * exp(log_s) + t ; +sum(log_s)      networks.py:296-298
            trans[tau].x = trans[tau].x * __expf(ls[tau].x) + t[tau].x;
            trans[tau].y = trans[tau].y * __expf(ls[tau].y) + t[tau].y;
            trans[tau].z = trans[tau].z * __expf(ls[tau].z) + t[tau].z;
            trans[tau].w = trans[tau].w * __expf(ls[tau].w) + t[tau].w;
            ld += (ls[tau].x + ls[tau].y) + (ls[tau].z + ls[tau].w);
        } else {  // (inputs - t) * exp(-log_s) ; -sum(log_s)             networks.py:307-309
            trans[tau].x = (trans[tau].x - t[tau].x) * __expf(-ls[tau].x);
            trans[tau].y = (trans[tau].y - t[tau].y) * __expf(-ls[tau].y);
            trans[tau].z = (trans[tau].z - t[tau].z) * __expf(-ls[tau].z);
            trans[tau].w = (trans[tau].w - t[tau].w) * __expf(-ls[tau].w);
            ld -= (ls[tau].x + ls[tau].y) + (ls[tau].z + ls[tau].w);
        }
    }
    return ld;
}

// The same coupling block with the layer structure fixed at compile time (L = number of hidden HxH layers) and
// written for instruction-level parallelism: a lone wave issues a dependent v_mfma_f32_16x16x4_f32 only every
// ~40 cycles (MI355X_MICROARCH.md), and the straightforward net-after-net order above measured 75 cycles per
// MFMA (tools/ablate_mh.hip).  Here the scale and translate nets advance layer by layer together and every
// K-accumulation is split over two accumulators, so four independent chains are in flight per layer.
// Fragment accessors: fragment number idx within one (block, net) -- L1 [o][tau][r], then L2 [l][o][i][r],
// then L3 [tau][i][r] -- read either from the image (LDS or global; pointer already offset by the lane) or from
// a per-lane register array filled once per kernel (image element of fragment idx for a lane: frag_elem).
struct ImageFrags {
    const float *p;  // (block, net) region + 4 * lane, 16-byte aligned
    __device__ __forceinline__ float operator()(int idx) const {
        return static_cast<const float *>(__builtin_assume_aligned(p, 16))[(idx >> 2) * 256 + (idx & 3)];
    }
};
template <int N> struct RegFrags {
    float v[N];
    __device__ __forceinline__ float operator()(int idx) const { return v[idx]; }
};
template <int NT, int NH, int L> struct FragCount {
    static constexpr int F1 = NH * NT * 4, F2 = L * NH * NH * 4, F3 = NT * NH * 4, N = F1 + F2 + F3;
};

template <int NT, int NH, int L, bool INVERSE, class FS, class FT>
__device__ __forceinline__ float coupling_core(const FS &fs, const FT &ft, const float *__restrict__ bias_s,
                                               const float *__restrict__ bias_t, int lane, const f32x4 (&cond)[NT],
                                               f32x4 (&trans)[NT]) {
    // bias_s / bias_t -> [b1: 16*NH][b2: L*16*NH][b3: 16*NT] of the scale / translate net
    typedef FragCount<NT, NH, L> FC;
    const int g4 = (lane >> 4) * 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 hs[NH], ht[NH];
    {  // Linear(D,H) + activation, both nets
        const float *bs = bias_s + g4, *bt = bias_t + g4;
#pragma unroll
        for (int o = 0; o < NH; ++o) {
            f32x4 s0 = *reinterpret_cast<const f32x4 *>(bs + 16 * o), s1 = zero4;
            f32x4 t0 = *reinterpret_cast<const f32x4 *>(bt + 16 * o), t1 = zero4;
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) {
                const int f = (o * NT + tau) * 4;
                s0 = mfma4(fs(f), cond[tau].x, s0);
                t0 = mfma4(ft(f), cond[tau].x, t0);
                s1 = mfma4(fs(f + 1), cond[tau].y, s1);
                t1 = mfma4(ft(f + 1), cond[tau].y, t1);
                s0 = mfma4(fs(f + 2), cond[tau].z, s0);
                t0 = mfma4(ft(f + 2), cond[tau].z, t0);
                s1 = mfma4(fs(f + 3), cond[tau].w, s1);
                t1 = mfma4(ft(f + 3), cond[tau].w, t1);
            }
            hs[o] = activate<0>(s0 + s1);
            ht[o] = activate<1>(t0 + t1);
        }
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {  // hidden Linear(H,H) + activation, both nets
        const float *bs = bias_s + 16 * NH + g4, *bt = bias_t + 16 * NH + g4;
        f32x4 hs2[NH], ht2[NH];
#pragma unroll
        for (int o = 0; o < NH; ++o) {
            f32x4 s0 = *reinterpret_cast<const f32x4 *>(bs + (l * NH + o) * 16), s1 = zero4;
            f32x4 t0 = *reinterpret_cast<const f32x4 *>(bt + (l * NH + o) * 16), t1 = zero4;
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                const int f = FC::F1 + ((l * NH + o) * NH + i) * 4;
                s0 = mfma4(fs(f), hs[i].x, s0);
                t0 = mfma4(ft(f), ht[i].x, t0);
                s1 = mfma4(fs(f + 1), hs[i].y, s1);
                t1 = mfma4(ft(f + 1), ht[i].y, t1);
                s0 = mfma4(fs(f + 2), hs[i].z, s0);
                t0 = mfma4(ft(f + 2), ht[i].z, t0);
                s1 = mfma4(fs(f + 3), hs[i].w, s1);
                t1 = mfma4(ft(f + 3), ht[i].w, t1);
            }
            hs2[o] = activate<0>(s0 + s1);
            ht2[o] = activate<1>(t0 + t1);
        }
#pragma unroll
        for (int o = 0; o < NH; ++o) { hs[o] = hs2[o]; ht[o] = ht2[o]; }
    }
    float ld = 0.f;
    {  // Linear(H,D) restricted to the transformed class, both nets, then the affine update
        const float *bs = bias_s + 16 * NH * (1 + L) + g4, *bt = bias_t + 16 * NH * (1 + L) + g4;
        f32x4 ls[NT], t[NT];
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            ls[tau] = *reinterpret_cast<const f32x4 *>(bs + 16 * tau);
            t[tau] = *reinterpret_cast<const f32x4 *>(bt + 16 * tau);
        }
#pragma unroll
        for (int i = 0; i < NH; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float bsv = r == 0 ? hs[i].x : r == 1 ? hs[i].y : r == 2 ? hs[i].z : hs[i].w;
                const float btv = r == 0 ? ht[i].x : r == 1 ? ht[i].y : r == 2 ? ht[i].z : ht[i].w;
#pragma unroll
                for (int tau = 0; tau < NT; ++tau) {  // 2*NT independent chains, round robin
                    const int f = FC::F1 + FC::F2 + (tau * NH + i) * 4 + r;
                    ls[tau] = mfma4(fs(f), bsv, ls[tau]);
                    t[tau] = mfma4(ft(f), btv, t[tau]);
                }
            }
        }
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            if (!INVERSE) {  // inputs * exp(log_s) + t ; +sum(log_s)      networks.py:296-298
                trans[tau].x = trans[tau].x * __expf(ls[tau].x) + t[tau].x;
                trans[tau].y = trans[tau].y * __expf(ls[tau].y) + t[tau].y;
                trans[tau].z = trans[tau].z * __expf(ls[tau].z) + t[tau].z;
                trans[tau].w = trans[tau].w * __expf(ls[tau].w) + t[tau].w;
                ld += (ls[tau].x + ls[tau].y) + (ls[tau].z + ls[tau].w);
            } else {  // (inputs - t) * exp(-log_s) ; -sum(log_s)             networks.py:307-309
                trans[tau].x = (trans[tau].x - t[tau].x) * __expf(-ls[tau].x);
                trans[tau].y = (trans[tau].y - t[tau].y) * __expf(-ls[tau].y);
                trans[tau].z = (trans[tau].z - t[tau].z) * __expf(-ls[tau].z);
                trans[tau].w = (trans[tau].w - t[tau].w) * __expf(-ls[tau].w);
                ld -= (ls[tau].x + ls[tau].y) + (ls[tau].z + ls[tau].w);
            }
        }
    }
    return ld;
}

// ONE net of a coupling block for the multi-wave "team" kernel (scale and translate nets on different waves),
// written for the shortest dependent path of a lone wave: the biases arrive in registers (prefetched by the
// caller while the previous block's exchange is in flight) and every K-accumulation is spread over enough
// accumulators that no v_mfma waits on the previous one (dependent issue 40 cycles vs 32 back to back), the
// partial accumulators being summed on the VALU afterwards.
template <int NT, int NH, int L> struct NetBias {
    f32x4 b1[NH];
    f32x4 b2[L > 0 ? L : 1][NH];
    f32x4 b3[NT];
    __device__ __forceinline__ void load(const float *__restrict__ bias, int lane) {
        const int g4 = (lane >> 4) * 4;
#pragma unroll
        for (int o = 0; o < NH; ++o) b1[o] = *reinterpret_cast<const f32x4 *>(bias + g4 + 16 * o);
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int o = 0; o < NH; ++o) b2[l][o] = *reinterpret_cast<const f32x4 *>(bias + 16 * NH + g4 + (l * NH + o) * 16);
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) b3[tau] = *reinterpret_cast<const f32x4 *>(bias + 16 * NH * (1 + L) + g4 + 16 * tau);
    }
};

template <int NT, int NH, int L, int ACT, class F>
__device__ __forceinline__ void mlp_core(const F &fr, const NetBias<NT, NH, L> &nb, const f32x4 (&cond)[NT], f32x4 (&out)[NT]) {
    typedef FragCount<NT, NH, L> FC;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 h[NH];
#pragma unroll
    for (int o = 0; o < NH; ++o) {  // four accumulators (one per k-step r), NT deep
        f32x4 a0 = nb.b1[o], a1 = zero4, a2 = zero4, a3 = zero4;
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            const int f = (o * NT + tau) * 4;
            a0 = mfma4(fr(f), cond[tau].x, a0);
            a1 = mfma4(fr(f + 1), cond[tau].y, a1);
            a2 = mfma4(fr(f + 2), cond[tau].z, a2);
            a3 = mfma4(fr(f + 3), cond[tau].w, a3);
        }
        h[o] = activate<ACT>((a0 + a1) + (a2 + a3));
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        f32x4 h2[NH];
#pragma unroll
        for (int o = 0; o < NH; ++o) {
            f32x4 a0 = nb.b2[l][o], a1 = zero4, a2 = zero4, a3 = zero4;
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                const int f = FC::F1 + ((l * NH + o) * NH + i) * 4;
                a0 = mfma4(fr(f), h[i].x, a0);
                a1 = mfma4(fr(f + 1), h[i].y, a1);
                a2 = mfma4(fr(f + 2), h[i].z, a2);
                a3 = mfma4(fr(f + 3), h[i].w, a3);
            }
            h2[o] = activate<ACT>((a0 + a1) + (a2 + a3));
        }
#pragma unroll
        for (int o = 0; o < NH; ++o) h[o] = h2[o];
    }
    f32x4 p0[NT], p1[NT];  // two accumulators per output tile
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) { p0[tau] = nb.b3[tau]; p1[tau] = zero4; }
#pragma unroll
    for (int i = 0; i < NH; ++i) {
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            const int f = FC::F1 + FC::F2 + (tau * NH + i) * 4;
            p0[tau] = mfma4(fr(f), h[i].x, p0[tau]);
            p1[tau] = mfma4(fr(f + 1), h[i].y, p1[tau]);
        }
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            const int f = FC::F1 + FC::F2 + (tau * NH + i) * 4;
            p0[tau] = mfma4(fr(f + 2), h[i].z, p0[tau]);
            p1[tau] = mfma4(fr(f + 3), h[i].w, p1[tau]);
        }
    }
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) out[tau] = p0[tau] + p1[tau];
}

// the affine update of a coupling block given both nets' outputs; returns the lane's log-det partial
template <int NT, bool INVERSE>
__device__ __forceinline__ float affine_update(const f32x4 (&ls)[NT], const f32x4 (&t)[NT], f32x4 (&trans)[NT]) {
    float ld = 0.f;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        if (!INVERSE) {
            trans[tau].x = trans[tau].x * __expf(ls[tau].x) + t[tau].x;
            trans[tau].y = trans[tau].y * __expf(ls[tau].y) + t[tau].y;
            trans[tau].z = trans[tau].z * __expf(ls[tau].z) + t[tau].z;
            trans[tau].w = trans[tau].w * __expf(ls[tau].w) + t[tau].w;
            ld += (ls[tau].x + ls[tau].y) + (ls[tau].z + ls[tau].w);
        } else {
            trans[tau].x = (trans[tau].x - t[tau].x) * __expf(-ls[tau].x);
            trans[tau].y = (trans[tau].y - t[tau].y) * __expf(-ls[tau].y);
            trans[tau].z = (trans[tau].z - t[tau].z) * __expf(-ls[tau].z);
            trans[tau].w = (trans[tau].w - t[tau].w) * __expf(-ls[tau].w);
            ld -= (ls[tau].x + ls[tau].y) + (ls[tau].z + ls[tau].w);
        }
    }
    return ld;
}

// image-backed form (fragments read from LDS/global at each use)
template <int NT, int NH, int L, bool INVERSE>
__device__ __forceinline__ float coupling_tile_il(const float *__restrict__ wblk, int net_floats, int lane,
                                                  const f32x4 (&cond)[NT], f32x4 (&trans)[NT]) {
    ImageFrags fs = {wblk + 4 * lane}, ft = {wblk + net_floats + 4 * lane};
    const int ob = frag_off_b1(NT, NH, L);
    return coupling_core<NT, NH, L, INVERSE>(fs, ft, wblk + ob, wblk + net_floats + ob, lane, cond, trans);
}

// Sum a per-lane partial over the 4 lane groups of a walker (lanes w, w+16, w+32, w+48); every lane ends with
// the walker total.  ONE round trip through the LDS crossbar (three independent ds_bpermute) instead of a
// two-round butterfly: measured, the serialized cross-lane rounds were ~45 % of the post-inverse part of an MH
// step.  The association (own pair) + (other pair) is the same set of additions in all four lanes (addition
// commutes), so the four lanes of a walker hold bit-identical totals -- they must, they take the same
// accept/reject decision.
__device__ __forceinline__ float group_sum(float v) {
    float t1 = __shfl_xor(v, 16), t2 = __shfl_xor(v, 32), t3 = __shfl_xor(v, 48);
    return (v + t1) + (t2 + t3);
}
// float partials, float64 total
__device__ __forceinline__ double group_sum_wide(float v) {
    float t1 = __shfl_xor(v, 16), t2 = __shfl_xor(v, 32), t3 = __shfl_xor(v, 48);
    return ((double)v + (double)t1) + ((double)t2 + (double)t3);
}
// all four lanes of a walker true?  No LDS: one ballot, scalar shifts.
__device__ __forceinline__ int group_all(bool ok_lane, int lane) {
    unsigned long long m = __ballot(ok_lane);
    m &= m >> 32;
    m &= m >> 16;
    return (int)((m >> (lane & 15)) & 1ull);
}

// one coupling block in either direction; LT >= 0 selects the compile-time-L interleaved form, LT = -1 the
// generic runtime-L form
template <int NT, int NH, int LT, bool INVERSE>
__device__ __forceinline__ float coupling_any(const float *__restrict__ wblk, int net_floats, int L, int lane,
                                              const f32x4 (&cond)[NT], f32x4 (&trans)[NT]) {
    if constexpr (LT >= 0) return coupling_tile_il<NT, NH, LT, INVERSE>(wblk, net_floats, lane, cond, trans);
    else return coupling_tile<NT, NH, INVERSE>(wblk, net_floats, L, lane, cond, trans);
}

// NormalizingFlow.forward (networks.py:24-32): blocks 0..B-1.  xs[c][tau] = class c tiles.
// Block b conditions on class (b+1)&1 and transforms class b&1 (mask = arange(D)%2 flipped per block).
// ScaleLayer (networks.py:312-325; SingleSpeedNVP scale='constant', networks.py:343-344): y = x e^s on every
// dimension, logdet += s (the scalar itself).  `ld` is a lane partial that is later summed over the four lane
// groups of a walker, so s is added on lane group 0 only.
template <int NT>
__device__ __forceinline__ float scale_layer_tile(float s, bool inverse, int lane, f32x4 (&xs)[2][NT]) {
    const float e = __expf(inverse ? -s : s);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t) xs[c][t] = xs[c][t] * e;
    return lane < 16 ? (inverse ? -s : s) : 0.f;
}

// blk_scale: the B ScaleLayer scalars (scale='constant'), or nullptr
template <int NT, int NH, int LT = -1>
__device__ __forceinline__ float flow_forward_tile(const float *__restrict__ img, int net_floats, int B, int L,
                                                   int lane, f32x4 (&xs)[2][NT], const float *blk_scale = nullptr) {
    float ld = 0.f;
    for (int b = 0; b < B; ++b) {
        const float *wblk = img + (size_t)b * 2 * net_floats;
        if (b & 1) ld += coupling_any<NT, NH, LT, false>(wblk, net_floats, L, lane, xs[0], xs[1]);
        else       ld += coupling_any<NT, NH, LT, false>(wblk, net_floats, L, lane, xs[1], xs[0]);
        if (blk_scale) ld += scale_layer_tile<NT>(blk_scale[b], false, lane, xs);
    }
    return ld;
}
template <int NT, int NH, int LT = -1>
__device__ __forceinline__ float flow_inverse_tile(const float *__restrict__ img, int net_floats, int B, int L,
                                                   int lane, f32x4 (&xs)[2][NT], const float *blk_scale = nullptr) {
    float ld = 0.f;
    for (int b = B - 1; b >= 0; --b) {
        const float *wblk = img + (size_t)b * 2 * net_floats;
        if (blk_scale) ld += scale_layer_tile<NT>(blk_scale[b], true, lane, xs);
        if (b & 1) ld += coupling_any<NT, NH, LT, true>(wblk, net_floats, L, lane, xs[0], xs[1]);
        else       ld += coupling_any<NT, NH, LT, true>(wblk, net_floats, L, lane, xs[1], xs[0]);
    }
    return ld;
}

// ---- global <-> tile ------------------------------------------------------------------------------
// lane (g,w) owns dims [32*tau + 8*g, +8) of row (row0 + w): xs[c][tau][r] = dim 32*tau + 8*g + 2*r + c.
template <int NT>
__device__ __forceinline__ void load_tile(const float *__restrict__ rows, long row, bool row_ok, int D, int lane,
                                          f32x4 (&xs)[2][NT]) {
    const int g = lane >> 4;
    const float *p = rows + (size_t)row * D;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int d = 32 * tau + 8 * g + j;
            v[j] = (row_ok && d < D) ? p[d] : 0.f;
        }
        xs[0][tau].x = v[0]; xs[1][tau].x = v[1];
        xs[0][tau].y = v[2]; xs[1][tau].y = v[3];
        xs[0][tau].z = v[4]; xs[1][tau].z = v[5];
        xs[0][tau].w = v[6]; xs[1][tau].w = v[7];
    }
}

// every coordinate d < D of the tile's walkers differs between a and b (the walker's dims are spread over its four lane groups): the
// same answer in the four lanes of a walker
template <int NT>
__device__ __forceinline__ int mh_all_coordinates_differ(const f32x4 (&a)[2][NT], const f32x4 (&b)[2][NT], int D, int lane) {
    const int g = lane >> 4, w = lane & 15;
    bool all = true;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int d0 = 32 * tau + 8 * g + c;   // load_tile: component r of class c is dim 32 tau + 8 g + 2 r + c
            all = all && (d0 >= D || a[c][tau].x != b[c][tau].x) && (d0 + 2 >= D || a[c][tau].y != b[c][tau].y) &&
                  (d0 + 4 >= D || a[c][tau].z != b[c][tau].z) && (d0 + 6 >= D || a[c][tau].w != b[c][tau].w);
        }
    const unsigned long long m = __ballot(all);
    return (int)((m >> w) & (m >> (w + 16)) & (m >> (w + 32)) & (m >> (w + 48)) & 1ull);
}

// the same against a tile still in memory (`rows` in load_tile's layout), a few coordinates at a time: no second tile in registers
template <int NT>
__device__ __forceinline__ int mh_all_coordinates_differ_from(const f32x4 (&a)[2][NT], const float *__restrict__ rows, long row, bool row_ok, int D, int lane) {
    const int g = lane >> 4, w = lane & 15;
    const float *p = rows + (size_t)row * D;
    bool all = true;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        const int d0 = 32 * tau + 8 * g;
        const float av[8] = {a[0][tau].x, a[1][tau].x, a[0][tau].y, a[1][tau].y, a[0][tau].z, a[1][tau].z, a[0][tau].w, a[1][tau].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool valid = row_ok && d0 + j < D;
            const float b = valid ? p[d0 + j] : 0.f;
            all = all && (!valid || av[j] != b);
        }
    }
    const unsigned long long m = __ballot(all);
    return (int)((m >> w) & (m >> (w + 16)) & (m >> (w + 32)) & (m >> (w + 48)) & 1ull);
}

template <int NT>
__device__ __forceinline__ void store_tile(float *__restrict__ rows, long row, bool row_ok, int D, int lane,
                                           const f32x4 (&xs)[2][NT]) {
    if (!row_ok) return;
    const int g = lane >> 4;
    float *p = rows + (size_t)row * D;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        float v[8] = {xs[0][tau].x, xs[1][tau].x, xs[0][tau].y, xs[1][tau].y,
                      xs[0][tau].z, xs[1][tau].z, xs[0][tau].w, xs[1][tau].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int d = 32 * tau + 8 * g + j;
            if (d < D) p[d] = v[j];
        }
    }
}

// ---- box prior ------------------------------------------------------------------------------------
// UniformPrior(D,-1,1).__call__ (nnest/priors.py:39-43): out of box iff any(x < -1) or any(x > 1);
// NaN compares false, i.e. counts as inside, exactly like the reference.  Padded dims hold 0.
template <int NT>
__device__ __forceinline__ int inbox_tile(const f32x4 (&xs)[2][NT], int lane) {
    int ok = 1;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            f32x4 v = xs[c][tau];
            ok &= !(v.x < -1.f || v.x > 1.f) & !(v.y < -1.f || v.y > 1.f) & !(v.z < -1.f || v.z > 1.f) &
                  !(v.w < -1.f || v.w > 1.f);
        }
    return group_all(ok != 0, lane);
}

// ---- likelihoods (nnest/likelihoods.py) through safe_loglike (nnest/sampler.py:110-133) --------------
// theta = like_scale * x in float32 (transform = lambda x: s*x on a float32 array, examples/nested/run.py:25-42);
// per-term arithmetic in float32 with the reference's operation order and no FMA contraction; each lane sums
// its own <= 8*NT terms in float32 (branch-free), the four lane partials of a walker are combined in float64
// (the reference sums all terms sequentially in float32; DESIGN.md "Precision").
// which likelihood, on scale * x, with its parameters (mirror of nnest_like_t, include/nnest_hip.h)
struct LikeSpec {
    int id = 0;
    float scale = 1.f;
    float p[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
};

// float64 sum of a per-lane float64 partial over a walker's 4 lane groups (one round trip, consistent in all 4 lanes)
__device__ __forceinline__ double group_sum(double v) {
    double t1 = __shfl_xor(v, 16), t2 = __shfl_xor(v, 32), t3 = __shfl_xor(v, 48);
    return (v + t1) + (t2 + t3);
}

#pragma clang fp contract(off)
template <int NT>
__device__ __forceinline__ double loglike_tile(const LikeSpec &lk, int D, int lane, const f32x4 (&xs)[2][NT]) {
    const int like_id = lk.id;
    const float scale = lk.scale;
    const int g = lane >> 4;
    double acc = 0.0;
    float facc = 0.f;
    if (like_id == 0) {
        // Rosenbrock (likelihoods.py:51): -sum_i 100*(x[i+1]-x[i]^2)^2 + (1-x[i])^2, i = 0..D-2
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            float th[9];
            th[0] = scale * xs[0][tau].x; th[1] = scale * xs[1][tau].x;
            th[2] = scale * xs[0][tau].y; th[3] = scale * xs[1][tau].y;
            th[4] = scale * xs[0][tau].z; th[5] = scale * xs[1][tau].z;
            th[6] = scale * xs[0][tau].w; th[7] = scale * xs[1][tau].w;
            // first dim of the next 8-block: lane group g+1 of this tile, or group 0 of the next tile
            float nxt_same = __shfl(th[0], (lane + 16) & 63);
            float nxt_tile = 0.f;
            if (tau + 1 < NT) nxt_tile = __shfl(scale * xs[0][tau + 1 < NT ? tau + 1 : tau].x, (lane + 16) & 63);
            th[8] = (g < 3) ? nxt_same : nxt_tile;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int i = 32 * tau + 8 * g + j;
                float a = th[j] * th[j];
                float b = th[j + 1] - a;
                float c = b * b;
                float e = 100.0f * c;
                float f = 1.0f - th[j];
                float q = f * f;
                float term = e + q;
                facc = facc + ((i + 1 < D) ? term : 0.f);
            }
        }
        acc = -group_sum_wide(facc);
    } else if (like_id == 1) {
        // GaussianMix (likelihoods.py:165-189): logsumexp_k[ log w_k - |theta - mu_k|^2/2 - (D/2) log 2pi ]
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            float th[8];
            th[0] = scale * xs[0][tau].x; th[1] = scale * xs[1][tau].x;
            th[2] = scale * xs[0][tau].y; th[3] = scale * xs[1][tau].y;
            th[4] = scale * xs[0][tau].z; th[5] = scale * xs[1][tau].z;
            th[6] = scale * xs[0][tau].w; th[7] = scale * xs[1][tau].w;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int d = 32 * tau + 8 * g + j;
                float sq = th[j] * th[j];
                facc = facc + ((d >= 2 && d < D) ? sq : 0.f);
            }
        }
        const double base = group_sum_wide(facc);
        const int w = lane & 15;
        float t0 = __shfl(scale * xs[0][0].x, w);  // theta[0], theta[1] live in lane group 0
        float t1 = __shfl(scale * xs[1][0].x, w);
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
    } else if (like_id == 2) {
        // Himmelblau (likelihoods.py:70) summed over consecutive pairs (x[2i], x[2i+1]) (= reference at D=2)
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            float e[4] = {xs[0][tau].x, xs[0][tau].y, xs[0][tau].z, xs[0][tau].w};
            float o[4] = {xs[1][tau].x, xs[1][tau].y, xs[1][tau].z, xs[1][tau].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int d1 = 2 * (16 * tau + 4 * g + r) + 1;
                float x0 = scale * e[r], x1 = scale * o[r];
                float a = x0 * x0 + x1 - 11.f;
                float b = x0 + x1 * x1 - 7.f;
                float v = -(a * a) - b * b;
                facc = facc + ((d1 < D) ? v : 0.f);
            }
        }
        acc = group_sum_wide(facc);
    } else if (like_id == 4) {
        // Eggbox (likelihoods.py:104-106), x_dim = 2: (2 + cos(x0/2) cos(x1/2))^5 in float32 like the reference
        const int w = lane & 15;
        float t0 = __shfl(scale * xs[0][0].x, w), t1 = __shfl(scale * xs[1][0].x, w);
        float chi = cosf(t0 / 2.f) * cosf(t1 / 2.f);
        float b = 2.f + chi;
        float b2 = b * b;
        acc = (double)(b2 * b2 * b);
    } else {
        // likelihoods whose reference arithmetic is float64 on the float32-rounded inputs (scipy logpdf / int64
        // centre arrays promote): moments of theta in float64
        double s1 = 0.0, s2 = 0.0;  // sum theta, sum theta^2
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            float th[8];
            th[0] = scale * xs[0][tau].x; th[1] = scale * xs[1][tau].x;
            th[2] = scale * xs[0][tau].y; th[3] = scale * xs[1][tau].y;
            th[4] = scale * xs[0][tau].z; th[5] = scale * xs[1][tau].z;
            th[6] = scale * xs[0][tau].w; th[7] = scale * xs[1][tau].w;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool valid = 32 * tau + 8 * g + j < D;
                const double t = valid ? (double)th[j] : 0.0;
                s1 += t;
                s2 += t * t;
            }
        }
        s1 = group_sum(s1);
        s2 = group_sum(s2);
        const double Dd = (double)D;
        if (like_id == 3) {
            // Gaussian (likelihoods.py:77-94): N(0, Sigma), Sigma = (1-c) I + c 11^T  (equicorrelated):
            // x^T Sigma^-1 x = (s2 - c s1^2 / (1 + (D-1) c)) / (1 - c),  log det = (D-1) log(1-c) + log(1 + (D-1) c)
            const double c = (double)lk.p[0];
            const double quad = (s2 - c * s1 * s1 / (1.0 + (Dd - 1.0) * c)) / (1.0 - c);
            const double logdet = (Dd - 1.0) * log(1.0 - c) + log(1.0 + (Dd - 1.0) * c);
            acc = -0.5 * quad - 0.5 * logdet - 0.9189385332046727 * Dd;
        } else {
            // GaussianShell (likelihoods.py:113-132): -(|theta - centre| - r)^2 / (2 sigma^2); |theta - c 1|^2 = s2 - 2 c s1 + D c^2
            // DoubleGaussianShell (likelihoods.py:135-150): logaddexp of two shells (weights 1, 1)
            double sh[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double sig = (double)lk.p[3 * k], rs = (double)lk.p[3 * k + 1], cen = (double)lk.p[3 * k + 2];
                double r2 = s2 - 2.0 * cen * s1 + Dd * cen * cen;
                double rad = sqrt(r2 > 0.0 ? r2 : 0.0);
                sh[k] = -((rad - rs) * (rad - rs)) / (2.0 * sig * sig);
            }
            if (like_id == 5) acc = sh[0];
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

// ---- counter-based noise: Philox4x32-10 + Box-Muller (build-defined; DESIGN.md "Proposal noise") ----
struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

enum { NOISE_STREAM_DZ = 0, NOISE_STREAM_U = 1, NOISE_STREAM_JITTER = 2 };

// four N(0,1) draws for dims 4q..4q+3 of (walker, step)
__device__ __forceinline__ f32x4 noise_normal4(uint64_t seed, uint64_t walker, uint32_t step, uint32_t q, uint32_t stream) {
    u32x4 c;
    c.x = q;
    c.y = (uint32_t)walker;
    c.z = step;
    c.w = ((uint32_t)(walker >> 32) & 0x0fffffffu) | (stream << 28);
    u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float two_m32 = 2.3283064365386963e-10f, two_m33 = 1.1641532182693481e-10f;
    float u1 = fmaf((float)r.x, two_m32, two_m33), a1 = (float)r.y * two_m32;
    float u2 = fmaf((float)r.z, two_m32, two_m33), a2 = (float)r.w * two_m32;
    float rad1 = __builtin_sqrtf(-2.0f * __logf(u1)), rad2 = __builtin_sqrtf(-2.0f * __logf(u2));
    f32x4 o;  // v_sin_f32 / v_cos_f32 take revolutions
    o.x = rad1 * __builtin_amdgcn_cosf(a1);
    o.y = rad1 * __builtin_amdgcn_sinf(a1);
    o.z = rad2 * __builtin_amdgcn_cosf(a2);
    o.w = rad2 * __builtin_amdgcn_sinf(a2);
    return o;
}

// one U[0,1) draw (24-bit, like torch.rand on float32) for (walker, step)
__device__ __forceinline__ float noise_uniform(uint64_t seed, uint64_t walker, uint32_t step) {
    u32x4 c;
    c.x = 0;
    c.y = (uint32_t)walker;
    c.z = step;
    c.w = ((uint32_t)(walker >> 32) & 0x0fffffffu) | ((uint32_t)NOISE_STREAM_U << 28);
    u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (float)(r.x >> 8) * 5.9604644775390625e-08f;
}

// ---- proposal stream of the persistent MH kernel -------------------------------------------------------
// Philox costs two 32x32->64 multiplies per round (quarter rate on CDNA) -- measured as the largest single
// VALU cost of the MH step when used per draw -- so inside the step loop each lane advances a
// xoshiro128++ state (Blackman & Vigna 2019: add / xor / shift / rotate only, full-rate VALU) that is SEEDED
// by one Philox4x32-10 block keyed by (seed; walker, lane group, stream).  A walker's stream still depends
// only on (seed, global walker index), so sharded runs reproduce the unsharded run bit for bit.
struct Xoshiro128 { uint32_t a, b, c, d; };

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int k) { return __builtin_amdgcn_alignbit(x, x, 32 - k); }

__device__ __forceinline__ uint32_t xoshiro_next(Xoshiro128 &s) {  // xoshiro128++
    uint32_t r = rotl32(s.a + s.d, 7) + s.a;
    uint32_t t = s.b << 9;
    s.c ^= s.a;
    s.d ^= s.b;
    s.b ^= s.c;
    s.a ^= s.d;
    s.c ^= t;
    s.d = rotl32(s.d, 11);
    return r;
}

__device__ __forceinline__ Xoshiro128 xoshiro_seed(uint64_t seed, uint64_t walker, uint32_t sub, uint32_t stream) {
    u32x4 c;
    c.x = sub;
    c.y = (uint32_t)walker;
    c.z = 0x58f1c3a5u;  // distinguishes the seeding blocks from the per-draw blocks above
    c.w = ((uint32_t)(walker >> 32) & 0x0fffffffu) | (stream << 28);
    u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    Xoshiro128 s = {r.x, r.y, r.z, r.w | 1u};  // never the all-zero state
    return s;
}

// Box-Muller pair from two 32-bit draws (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32 in revolutions)
__device__ __forceinline__ void box_muller(uint32_t r0, uint32_t r1, float &n0, float &n1) {
    const float two_m32 = 2.3283064365386963e-10f, two_m33 = 1.1641532182693481e-10f;
    float u1 = fmaf((float)r0, two_m32, two_m33), ang = (float)r1 * two_m32;
    // v_log_f32 is log2: -2 ln(u) = (-2 ln 2) log2(u); raw v_sqrt_f32 (1 ulp) -- no refinement sequence needed for noise
    float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    n0 = rad * __builtin_amdgcn_cosf(ang);
    n1 = rad * __builtin_amdgcn_sinf(ang);
}

// eight N(0,1) draws = the 8 consecutive dims a lane owns in one tile
__device__ __forceinline__ void xoshiro_normal8(Xoshiro128 &s, float (&n)[8]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        uint32_t r0 = xoshiro_next(s), r1 = xoshiro_next(s);
        box_muller(r0, r1, n[2 * p], n[2 * p + 1]);
    }
}

__device__ __forceinline__ float xoshiro_uniform(Xoshiro128 &s) { return (float)(xoshiro_next(s) >> 8) * 5.9604644775390625e-08f; }

}  // namespace nnest
