// maf_tile.h -- masked autoregressive flow (MAF) on the tile code of the coupling stack.
// [Build-defined: the reference has no MAF (nnest/trainer.py:83-100 accepts 'choleksy' / 'nvp' / 'spline').]
//
// Definition.  B blocks; block b holds two MADE nets with the shapes of the reference's coupling nets (networks.py:271-282):
// scale net Linear(D,H) Tanh [Linear(H,H) Tanh]xL Linear(H,D), translate net the same with ReLU -- the packed parameter vector
// has the RealNVP layout and size.  Degrees: dimension d has degree d + 1 in even blocks, D - d in odd blocks (order reversed
// between blocks); hidden unit k has degree 1 + floor(k (D - 1) / H).  Masks: first layer W[k,d] lives iff deg(k) >= deg(d);
// hidden W[k',k] iff deg(k') >= deg(k); last layer W[d,k] iff deg(d) > deg(k).
// Orientation (MAF proper): the DENSITY direction is the single pass
//     forward  x -> z:  z_d = x_d exp(s_d(x)) + t_d(x),    logdet = +sum s      (training, log_probs)
//     inverse  z -> x:  x_d = (z_d - t_d(x)) exp(-s_d(x)),  logdet = -sum s      (sampling, the Metropolis proposals)
// and the inverse is sequential in the degrees -- but with H hidden units there are at most H distinct hidden degrees, so the
// dimensions fall into G <= H + 1 groups (group of d = number of distinct hidden degrees below deg(d)) whose members depend on
// earlier groups only: the inverse is G passes of the nets, not D (DESIGN.md 3c has the cost and why this orientation).
//
// On the GPU a block is the coupling tile code run "dense": the walker's whole vector -- both parity classes of flow_tile.h's
// layout, 2 NT tiles -- is at once the conditioning input and the transformed output of mlp_tile<2 NT, NH, ACT>, the masks are
// zeros in the weight fragment image.  Image = per (block, net) the fragment layout of flow_tile.h with NT2 = 2 NT, then per
// block the group of every slot (16 NT2 floats holding small integers).  Tile tp of the flattened vector = class tp / NT,
// tile tp % NT; slot i of it is dimension 2 (16 (tp % NT) + i) + tp / NT.
#pragma once
#include "flow_tile.h"

namespace nnest {

__host__ __device__ inline int maf_deg_in(int D, int b, int d) { return (b & 1) ? D - d : d + 1; }
__host__ __device__ inline int maf_deg_hid(int D, int H, int k) { return D < 2 ? 1 : 1 + (int)(((long long)k * (D - 1)) / H); }
__host__ __device__ inline int maf_group_of(int D, int H, int b, int d) {
    const int din = maf_deg_in(D, b, d);
    int n = 0, prev = 0;
    for (int k = 0; k < H; ++k) {
        const int dk = maf_deg_hid(D, H, k);
        if (dk != prev) { if (dk < din) ++n; prev = dk; }
    }
    return n;
}
__host__ __device__ inline int maf_num_groups(int D, int H) {
    int n = 0, prev = 0;
    for (int k = 0; k < H; ++k) { const int dk = maf_deg_hid(D, H, k); if (dk != prev) { ++n; prev = dk; } }
    return n + 1;
}
__host__ __device__ inline int maf_dim(int NT, int tp, int i) { return 2 * (16 * (tp % NT) + i) + tp / NT; }

// element idx of the forward image -> packed parameter index (-1: structural zero = padding or masked); behind the fragments
// the group table, signalled as -2 - group
__host__ __device__ inline int maf_fwd_src(const FlowShape &s, int idx) {
    const int NT = s.NT, NT2 = 2 * s.NT, NH = s.NH, L = s.L, D = s.D, H = s.H;
    const int nets = s.B * 2 * s.net_floats;
    if (idx >= nets) {
        const int o = idx - nets, b = o / (16 * NT2), sl = o % (16 * NT2), d = maf_dim(NT, sl / 16, sl % 16);
        return -2 - (d < D ? maf_group_of(D, H, b, d) : 0x3fffff);
    }
    const int bn = idx / s.net_floats, o = idx - bn * s.net_floats, b = bn >> 1;
    const int base = bn * s.net_params;
    const int pb0 = H * D, phid = H * D + H, pWo = H * D + H + L * (H * H + H), pbo = pWo + D * H;
    if (o < frag_off_L2(NT2, NH)) {   // L1 [ht][tp][lane][r]: A-fragment of W0: lane (g, i) = W0[16 ht + i][dim(tp, 4 g + r)]
        const int r = o & 3, lane = (o >> 2) & 63, q = o >> 8, tp = q % NT2, ht = q / NT2;
        const int g = lane >> 4, i = lane & 15, k = 16 * ht + i, d = maf_dim(NT, tp, 4 * g + r);
        return (d < D && maf_deg_hid(D, H, k) >= maf_deg_in(D, b, d)) ? base + k * D + d : -1;
    } else if (o < frag_off_L3(NT2, NH, L)) {   // L2 [l][hto][hti][lane][r]: lane (g, i) = W_l[16 hto + i][16 hti + 4 g + r]
        const int oo = o - frag_off_L2(NT2, NH);
        const int r = oo & 3, lane = (oo >> 2) & 63, q = oo >> 8, hti = q % NH, hto = (q / NH) % NH, l = q / (NH * NH);
        const int g = lane >> 4, i = lane & 15, ko = 16 * hto + i, ki = 16 * hti + 4 * g + r;
        return maf_deg_hid(D, H, ko) >= maf_deg_hid(D, H, ki) ? base + phid + l * (H * H + H) + ko * H + ki : -1;
    } else if (o < frag_off_b1(NT2, NH, L)) {   // L3 [tp][ht][lane][r]: lane (g, i) = Wout[dim(tp, i)][16 ht + 4 g + r]
        const int oo = o - frag_off_L3(NT2, NH, L);
        const int r = oo & 3, lane = (oo >> 2) & 63, q = oo >> 8, ht = q % NH, tp = q / NH;
        const int g = lane >> 4, i = lane & 15, d = maf_dim(NT, tp, i), k = 16 * ht + 4 * g + r;
        return (d < D && maf_deg_in(D, b, d) > maf_deg_hid(D, H, k)) ? base + pWo + d * H + k : -1;
    } else if (o < frag_off_b2(NT2, NH, L)) {
        return base + pb0 + (o - frag_off_b1(NT2, NH, L));
    } else if (o < frag_off_b3(NT2, NH, L)) {
        const int oo = o - frag_off_b2(NT2, NH, L), l = oo / (16 * NH), j = oo % (16 * NH);
        return base + phid + l * (H * H + H) + H * H + j;
    } else {
        const int sl = o - frag_off_b3(NT2, NH, L), d = maf_dim(NT, sl / 16, sl % 16);
        return d < D ? base + pbo + d : -1;
    }
}

// backward image (transposed A-fragments for the delta-propagation of training; regions as nnest_train.hip's bwd_image_src)
//   L1-sized  B3 [ht][tp][lane][r]      g_h[ht]  += Wout^T : lane (g, i) = Wout[dim(tp, 4 g + r)][16 ht + i]
//   L2-sized  B2 [l][hti][hto][lane][r] g_h[hti] += W_l^T  : lane (g, i) = W_l[16 hto + 4 g + r][16 hti + i]
//   L3-sized  B1 [tp][ht][lane][r]      g_x[tp]  += W0^T   : lane (g, i) = W0[16 ht + 4 g + r][dim(tp, i)]
__host__ __device__ inline int maf_bwd_src(const FlowShape &s, int idx) {
    const int NT = s.NT, NT2 = 2 * s.NT, NH = s.NH, L = s.L, D = s.D, H = s.H;
    if (idx >= s.B * 2 * s.net_floats) return -1;
    const int bn = idx / s.net_floats, o = idx - bn * s.net_floats, b = bn >> 1;
    const int base = bn * s.net_params;
    const int phid = H * D + H, pWo = H * D + H + L * (H * H + H);
    if (o < frag_off_L2(NT2, NH)) {
        const int r = o & 3, lane = (o >> 2) & 63, q = o >> 8, tp = q % NT2, ht = q / NT2;
        const int g = lane >> 4, i = lane & 15, d = maf_dim(NT, tp, 4 * g + r), k = 16 * ht + i;
        return (d < D && maf_deg_in(D, b, d) > maf_deg_hid(D, H, k)) ? base + pWo + d * H + k : -1;
    } else if (o < frag_off_L3(NT2, NH, L)) {
        const int oo = o - frag_off_L2(NT2, NH);
        const int r = oo & 3, lane = (oo >> 2) & 63, q = oo >> 8, hto = q % NH, hti = (q / NH) % NH, l = q / (NH * NH);
        const int g = lane >> 4, i = lane & 15, ko = 16 * hto + 4 * g + r, ki = 16 * hti + i;
        return maf_deg_hid(D, H, ko) >= maf_deg_hid(D, H, ki) ? base + phid + l * (H * H + H) + ko * H + ki : -1;
    } else if (o < frag_off_b1(NT2, NH, L)) {
        const int oo = o - frag_off_L3(NT2, NH, L);
        const int r = oo & 3, lane = (oo >> 2) & 63, q = oo >> 8, ht = q % NH, tp = q / NH;
        const int g = lane >> 4, i = lane & 15, d = maf_dim(NT, tp, i), k = 16 * ht + 4 * g + r;
        return (d < D && maf_deg_hid(D, H, k) >= maf_deg_in(D, b, d)) ? base + k * D + d : -1;
    }
    return -1;
}

// ---- tile code -----------------------------------------------------------------------------------------------------
// density direction, all blocks: z = f(x); returns the lane's log-det partial
template <int NT, int NH>
__device__ __forceinline__ float maf_forward_tile(const float *__restrict__ img, const FlowShape &s, int lane, f32x4 (&xs)[2][NT]) {
    constexpr int NT2 = 2 * NT;
    f32x4 (&v)[NT2] = reinterpret_cast<f32x4 (&)[NT2]>(xs);
    float ld = 0.f;
    for (int b = 0; b < s.B; ++b) {
        const float *wblk = img + (size_t)b * 2 * s.net_floats;
        f32x4 ls[NT2], t[NT2];
        mlp_tile<NT2, NH, 0>(wblk, s.L, lane, v, ls);
        mlp_tile<NT2, NH, 1>(wblk + s.net_floats, s.L, lane, v, t);
        ld += affine_update<NT2, false>(ls, t, v);
    }
    return ld;
}

// sampling direction, all blocks (reversed), each group by group
template <int NT, int NH>
__device__ __forceinline__ float maf_inverse_tile(const float *__restrict__ img, const FlowShape &s, int lane, f32x4 (&xs)[2][NT]) {
    constexpr int NT2 = 2 * NT;
    f32x4 (&v)[NT2] = reinterpret_cast<f32x4 (&)[NT2]>(xs);
    float ld = 0.f;
    const int G = s.G;
    for (int b = s.B - 1; b >= 0; --b) {
        const float *wblk = img + (size_t)b * 2 * s.net_floats;
        const float *grp = img + (size_t)s.B * 2 * s.net_floats + (size_t)b * 16 * NT2 + (lane >> 4) * 4;
        for (int g = 0; g < G; ++g) {
            f32x4 ls[NT2], t[NT2];
            mlp_tile<NT2, NH, 0>(wblk, s.L, lane, v, ls);
            mlp_tile<NT2, NH, 1>(wblk + s.net_floats, s.L, lane, v, t);
            const float gf = (float)g;
#pragma unroll
            for (int tp = 0; tp < NT2; ++tp) {
                const f32x4 gid = *reinterpret_cast<const f32x4 *>(grp + 16 * tp);
                if (gid.x == gf) { v[tp].x = (v[tp].x - t[tp].x) * __expf(-ls[tp].x); ld -= ls[tp].x; }
                if (gid.y == gf) { v[tp].y = (v[tp].y - t[tp].y) * __expf(-ls[tp].y); ld -= ls[tp].y; }
                if (gid.z == gf) { v[tp].z = (v[tp].z - t[tp].z) * __expf(-ls[tp].z); ld -= ls[tp].z; }
                if (gid.w == gf) { v[tp].w = (v[tp].w - t[tp].w) * __expf(-ls[tp].w); ld -= ls[tp].w; }
            }
        }
    }
    return ld;
}

}  // namespace nnest
