// spline_train_tile.h -- device code of the spline flow's training step (reference nnest/trainer.py:384-403 on
// SingleSpeedSpline, networks.py:393-715): forward with the block inputs stashed, then a hand-written backward that
// recomputes one block at a time (ActNorm -> 1x1 conv -> two RQ-spline couplings), one wave64 per 16 rows.
//
// Tiles and slot numbering are those of spline_tile.h.  Parameter gradients are contractions over the 16 rows of the
// tile: both operands are transposed through a 16x17 LDS tile and multiplied on the matrix cores
// (out[m][n] = sum_row G[m][row] A[n][row]); every wave writes its partial sums to its own slice of the gradient
// workspace, which the optimiser kernel adds up in a fixed order (no atomics: training is bitwise reproducible).
#pragma once
#include "spline_tile.h"

namespace nnest {

// training image of one block (rebuilt on the device from the packed weights before every minibatch):
//   [ conv fwd: W as A-fragments of c = a W | conv bwd: fragments of g_a = g_c W^T | f1 fwd | f2 fwd | f1 bwd | f2 bwd ]
// conditioner fwd = the inference layout (spline_tile.h); conditioner bwd = [B1 | B2 | B3 | B4]:
//   B1 [t][ht][r][64]      g_in[tile t]  += W0^T g_pre1      lane(g,i) = W0[16ht+4g+r][dim of (tile t, row i)]
//   B2 [hti][hto][r][64]   g_h1[hti]     += W1^T g_pre2      lane(g,i) = W1[16hto+4g+r][16hti+i]
//   B3 likewise for W2;
//   B4 [s][q][hto][r][64]  g_h3[hto]     += W3^T g_raw       lane(g,i) = W3[(4s+g)*23 + 4q+r][16hto+i]
struct SplTrainShape {
    SplineShape s;
    int conv_floats;         // (2 NTh)^2 * 256, one direction
    int cf[2], cb[2];        // conditioner fwd / bwd fragment floats (f1, f2)
    int tblk_floats, timage_floats;
    int p_s, p_t, p_L, p_S, p_U, p_f[2];  // offsets inside a packed block
    int SM;                  // max(SU, SL): super-tile slots per coupling in the raw-gradient buffer (spl_w3_*)
    int gw_floats;           // workspace per wave: packed gradient slice + B * D * D (dLoss/dW of the convs) + 4 (loss)
};

__host__ __device__ inline int spl_cond_bwd_floats(int NTh, int NH, int S) { return NTh * NH * 256 + 2 * NH * NH * 256 + S * SPL_QT * NH * 256; }

// 16x16 tile transpose through LDS: in (C/D layout) lane (g,w) reg r = V[4g+r][w]  ->  out[kk] of lane (gq,i) = V[i][4kk+gq],
// i.e. the A (or B) operand of a product contracted over the 16 rows/walkers of the tile
__device__ __forceinline__ void tile_transpose(float *lds17, int lane, f32x4 v, float (&out)[4]) {
    const int w = lane & 15, g = lane >> 4;
    lds17[(4 * g + 0) * 17 + w] = v.x; lds17[(4 * g + 1) * 17 + w] = v.y;
    lds17[(4 * g + 2) * 17 + w] = v.z; lds17[(4 * g + 3) * 17 + w] = v.w;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) out[kk] = lds17[w * 17 + 4 * kk + g];
    __builtin_amdgcn_wave_barrier();
}

// N tiles in one LDS round trip (scratch: N x 16 x 17 floats, N <= SPL_TBATCH)
enum { SPL_TBATCH = 8 };
template <int N>
__device__ __forceinline__ void tile_transpose_batch(float *lds17, int lane, const f32x4 (&v)[N], float (&out)[N][4]) {
    static_assert(N <= SPL_TBATCH, "scratch holds SPL_TBATCH tiles");
    const int w = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float *t = lds17 + i * (16 * 17);
        t[(4 * g + 0) * 17 + w] = v[i].x; t[(4 * g + 1) * 17 + w] = v[i].y;
        t[(4 * g + 2) * 17 + w] = v[i].z; t[(4 * g + 3) * 17 + w] = v[i].w;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) out[i][kk] = lds17[i * (16 * 17) + w * 17 + 4 * kk + g];
    __builtin_amdgcn_wave_barrier();
}

// out[m][n] = sum_row G[m][row] A[n][row]; operands already transposed; result lane (g,j) reg r = out[4g+r][j]
__device__ __forceinline__ f32x4 contract16(const float (&gt)[4], const float (&at)[4]) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = mfma4(gt[kk], at[kk], acc);
    return acc;
}

__device__ __forceinline__ f32x4 lrelu_grad4(f32x4 g, f32x4 post) {  // post-activation sign = pre-activation sign
    f32x4 o;
    o.x = post.x > 0.f ? g.x : 0.2f * g.x; o.y = post.y > 0.f ? g.y : 0.2f * g.y;
    o.z = post.z > 0.f ? g.z : 0.2f * g.z; o.w = post.w > 0.f ? g.w : 0.2f * g.w;
    return o;
}

// N consecutive 64-float fragment rows into registers, ALL loads issued before the first use (the scheduler is held to it):
// written fragment by fragment next to its matrix instruction, the compiler waits for every group of four loads in turn
// -- one exposed L2 round trip per group, ~850 s_waitcnt in the gradient kernel, a quarter of its cycles.
template <int N>
__device__ __forceinline__ void load_frags(const float *__restrict__ base, int lane, float (&f)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) f[i] = base[(size_t)i * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
}
// N bias quads of this lane group (f32x4 at stride `stride` floats), same discipline
template <int N>
__device__ __forceinline__ void load_bias4(const float *__restrict__ base, int stride, f32x4 (&b)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) b[i] = *reinterpret_cast<const f32x4 *>(base + (size_t)i * stride);
    __builtin_amdgcn_sched_barrier(0);
}

// conditioner trunk keeping the three hidden activations
template <int NTh, int NH>
__device__ __forceinline__ void spl_hidden_keep(const float *__restrict__ net, int lane, const f32x4 (&in)[NTh], f32x4 (&h)[3][NH]) {
    const int g = lane >> 4;
    const float *L1 = net, *L2 = net + NH * NTh * 256, *b = L2 + 2 * NH * NH * 256;
    float w1[NH * NTh * 4], w23[2 * NH * NH * 4];
    f32x4 bq[3 * NH];  // b1 | b2 | b3, 16 NH floats each: this lane group's quad of every hidden tile
    load_bias4<3 * NH>(b + 4 * g, 16, bq);
    load_frags<NH * NTh * 4>(L1, lane, w1);
    load_frags<2 * NH * NH * 4>(L2, lane, w23);
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) {
        f32x4 a0 = bq[ht], a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NTh; ++t) {
            const float *a = w1 + (ht * NTh + t) * 4;
            a0 = mfma4(a[0], in[t].x, a0);
            a1 = mfma4(a[1], in[t].y, a1);
            a0 = mfma4(a[2], in[t].z, a0);
            a1 = mfma4(a[3], in[t].w, a1);
        }
        h[0][ht] = lrelu4(a0 + a1);
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma unroll
        for (int hto = 0; hto < NH; ++hto) {
            f32x4 a0 = bq[(l + 1) * NH + hto], a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hti = 0; hti < NH; ++hti) {
                const float *a = w23 + ((l * NH + hto) * NH + hti) * 4;
                a0 = mfma4(a[0], h[l][hti].x, a0);
                a1 = mfma4(a[1], h[l][hti].y, a1);
                a0 = mfma4(a[2], h[l][hti].z, a0);
                a1 = mfma4(a[3], h[l][hti].w, a1);
            }
            h[l + 1][hto] = lrelu4(a0 + a1);
        }
    }
}

// the trunk's fragments and biases in registers (spl_hidden_keep's load phase on its own: the training kernel's forward pass
// requests them one coupling ahead)
template <int NTh, int NH>
struct SplTrunkFrags {
    float w1[NH * NTh * 4], w23[2 * NH * NH * 4];
    f32x4 bq[3 * NH];
};
template <int NTh, int NH>
__device__ __forceinline__ void spl_trunk_load(const float *__restrict__ net, int lane, SplTrunkFrags<NTh, NH> &f) {
    const float *L1 = net, *L2 = net + NH * NTh * 256, *b = L2 + 2 * NH * NH * 256;
    load_bias4<3 * NH>(b + 4 * (lane >> 4), 16, f.bq);
    load_frags<NH * NTh * 4>(L1, lane, f.w1);
    load_frags<2 * NH * NH * 4>(L2, lane, f.w23);
}
template <int NTh, int NH>
__device__ __forceinline__ void spl_hidden_keep_pre(const SplTrunkFrags<NTh, NH> &f, const f32x4 (&in)[NTh], f32x4 (&h)[3][NH]) {
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) {
        f32x4 a0 = f.bq[ht], a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NTh; ++t) {
            const float *a = f.w1 + (ht * NTh + t) * 4;
            a0 = mfma4(a[0], in[t].x, a0);
            a1 = mfma4(a[1], in[t].y, a1);
            a0 = mfma4(a[2], in[t].z, a0);
            a1 = mfma4(a[3], in[t].w, a1);
        }
        h[0][ht] = lrelu4(a0 + a1);
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma unroll
        for (int hto = 0; hto < NH; ++hto) {
            f32x4 a0 = f.bq[(l + 1) * NH + hto], a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hti = 0; hti < NH; ++hti) {
                const float *a = f.w23 + ((l * NH + hto) * NH + hti) * 4;
                a0 = mfma4(a[0], h[l][hti].x, a0);
                a1 = mfma4(a[1], h[l][hti].y, a1);
                a0 = mfma4(a[2], h[l][hti].z, a0);
                a1 = mfma4(a[3], h[l][hti].w, a1);
            }
            h[l + 1][hto] = lrelu4(a0 + a1);
        }
    }
}

// raw spline parameters of super-tile s from the last hidden activation
template <int NH>
__device__ __forceinline__ void spl_raw_load(const float *__restrict__ L4, const float *__restrict__ b4, int s, int lane,
                                             float (&w4)[SPL_QT * NH * 4], f32x4 (&bq)[SPL_QT]) {
    load_bias4<SPL_QT>(b4 + (s * SPL_QT * 4 + (lane >> 4)) * 4, 16, bq);
    load_frags<SPL_QT * NH * 4>(L4 + (size_t)s * SPL_QT * NH * 256, lane, w4);
}
template <int NH>
__device__ __forceinline__ void spl_raw_mma(const float (&w4)[SPL_QT * NH * 4], const f32x4 (&bq)[SPL_QT], const f32x4 (&h3)[NH], f32x4 (&raw)[SPL_QT]) {
#pragma unroll
    for (int q = 0; q < SPL_QT; ++q) {
        f32x4 acc = bq[q];
#pragma unroll
        for (int hti = 0; hti < NH; ++hti) {
            const float *a = w4 + (q * NH + hti) * 4;
            acc = mfma4(a[0], h3[hti].x, acc);
            acc = mfma4(a[1], h3[hti].y, acc);
            acc = mfma4(a[2], h3[hti].z, acc);
            acc = mfma4(a[3], h3[hti].w, acc);
        }
        raw[q] = acc;
    }
}
template <int NH>
__device__ __forceinline__ void spl_raw(const float *__restrict__ L4, const float *__restrict__ b4, int s, int lane, const f32x4 (&h3)[NH],
                                        f32x4 (&raw)[SPL_QT]) {
    float w4[SPL_QT * NH * 4];
    f32x4 bq[SPL_QT];
    spl_raw_load<NH>(L4, b4, s, lane, w4, bq);
    spl_raw_mma<NH>(w4, bq, h3, raw);
}
// the last-layer fragments of a wave's first pair of super-tiles (s = wv and wv + 4) in registers, requested a coupling ahead
template <int NH>
struct SplRawFrags {
    float wA[SPL_QT * NH * 4], wB[SPL_QT * NH * 4];
    f32x4 bA[SPL_QT], bB[SPL_QT];
};
template <int NTh, int NH>
__device__ __forceinline__ void spl_rawfrags_load(const float *__restrict__ net, int S, int wv, int lane, SplRawFrags<NH> &f) {
    const float *L4 = net + spl_cond_hidden_floats(NTh, NH);
    const float *b4 = L4 + (size_t)S * SPL_QT * NH * 256;
    const int sA = wv < S ? wv : 0, sB = (NTh >= 2 && wv + 4 < S) ? wv + 4 : sA;
    spl_raw_load<NH>(L4, b4, sA, lane, f.wA, f.bA);
    spl_raw_load<NH>(L4, b4, sB, lane, f.wB, f.bB);
}

// (half_swap / half_swap4 / sel4 -- the two halves of the 16 columns of an 8-row tile -- live in spline_tile.h: the proposal kernel's
// 8-walker form uses them too)

// f32x4 per lane that one wave keeps per coupling (spl_coupling_pair): 3 NH activations + 6 per pair of super-tiles
__host__ __device__ inline int spl_keep_floats4(int NTh, int NH) { return 3 * NH + ((NTh + 1) / 2) * SPL_QT; }

// Forward coupling of the training kernel.  Wave wv owns the super-tiles s = wv + 4k (register wv of tile k); it takes them two
// at a time: (k, k + 1) go to the low / high half of the columns, so ONE spline evaluation per lane serves two super-tiles
// (the matrix work per super-tile is unchanged, the spline arithmetic -- the bulk of the instruction stream -- halves).
// Both halves leave with the full transformed half again (results swapped across).  Returns this lane's share of log|det|.
// TAILB = false: no barrier behind the exchange's reads -- the caller alternates between two exchange buffers, so the next
// exchange writes the other one and the one after it lies behind that exchange's barrier (round 6: one barrier per exchange)
template <int NTh, int NH, int TEAM, bool TAILB = true>
__device__ __forceinline__ float spl_coupling_pair(const float *__restrict__ net, int S, int n_out, float tail, int lane,
                                                   const f32x4 (&cond)[NTh], f32x4 (&tr)[NTh], int wv, f32x4 *xch,
                                                   f32x4 *__restrict__ keep, SplTrunkFrags<NTh, NH> &tf, SplRawFrags<NH> &rf,
                                                   const float *__restrict__ next_net, int next_S
#ifdef NNEST_STAMP
                                                   , long long *fst = nullptr
#endif
                                                   ) {
    static_assert(TEAM == 4, "one wave per register of a tile");
#ifdef NNEST_STAMP   // (diagnostic build: where a forward coupling's time goes -- tools/stamp_spline_train.py)
    long long f_a = wall_clock64();
#define CF_STAMP(i, v) { if (fst) { asm volatile("" :: "v"(v)); const long long f_n = wall_clock64(); fst[i] += f_n - f_a; f_a = f_n; } }
#else
#define CF_STAMP(i, v)
#endif
    const int g = lane >> 4;
    const bool lo = (lane & 15) < 8;
    // `keep` (this wave's slice, spl_keep_floats4): the three hidden activations and the spline parameters of the wave's pairs
    // stay for the backward pass, which then neither repeats the trunk nor the last layer
    // `tf`: this coupling's trunk fragments, requested a coupling ago; the next coupling's (`next_net`) are requested as soon as
    // this trunk has consumed them, and arrive behind the spline arithmetic
    f32x4 hk[3][NH];
    spl_hidden_keep_pre<NTh, NH>(tf, cond, hk);
    CF_STAMP(0, hk[2][0].x)
    if (next_net) spl_trunk_load<NTh, NH>(next_net, lane, tf);
    f32x4 h[NH];
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) h[ht] = hk[2][ht];
    if (keep) {
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int ht = 0; ht < NH; ++ht) keep[(l * NH + ht) * 64 + lane] = hk[l][ht];
    }
    const float *L4 = net + spl_cond_hidden_floats(NTh, NH);
    const float *b4 = L4 + (size_t)S * SPL_QT * NH * 256;
    float ld = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r != wv) continue;  // uniform over the wave
#pragma unroll
        for (int k = 0; k < NTh; k += 2) {
            const int sA = 4 * k + r, sB = 4 * (k + 1) + r;
            if (sA >= S) continue;
            const bool hasB = (k + 1 < NTh) && sB < S;
            f32x4 raw[SPL_QT];
            if (k == 0) spl_raw_mma<NH>(rf.wA, rf.bA, h, raw);  // (the first pair's fragments came a coupling ahead)
            else spl_raw<NH>(L4, b4, sA, lane, h, raw);
            float x = reg_of(tr[k], r);
            if (k + 1 < NTh) {
                if (hasB) {
                    f32x4 rawB[SPL_QT];
                    if (k == 0) spl_raw_mma<NH>(rf.wB, rf.bB, h, rawB);
                    else spl_raw<NH>(L4, b4, sB, lane, h, rawB);
#pragma unroll
                    for (int q = 0; q < SPL_QT; ++q) raw[q] = sel4(lo, raw[q], rawB[q]);
                }
                x = lo ? x : reg_of(tr[(k + 1 < NTh) ? k + 1 : k], r);
            }
            CF_STAMP(1, raw[0].x + raw[5].z)
            if (keep) {
#pragma unroll
                for (int q = 0; q < SPL_QT; ++q) keep[(3 * NH + (k >> 1) * SPL_QT + q) * 64 + lane] = raw[q];
            }
            const bool valid = lo ? (4 * sA + g < n_out) : (hasB && 4 * sB + g < n_out);
            float l = 0.f;
            const float y = spl_rqs<false>(raw, tail, x, l);
            const float yo = valid ? y : 0.f, yp = half_swap(yo);
            ld += valid ? l : 0.f;
            set_reg(tr[k], r, lo ? yo : yp);
            if (k + 1 < NTh && hasB) set_reg(tr[(k + 1 < NTh) ? k + 1 : k], r, lo ? yp : yo);
            CF_STAMP(2, yo + ld)
        }
    }
    if (next_net) spl_rawfrags_load<NTh, NH>(next_net, next_S, wv, lane, rf);
    // super-tile s = 4t + r (register r of tile t) comes from wave r
#pragma unroll
    for (int t = 0; t < NTh; ++t) xch[(wv * NTh + t) * 64 + lane] = tr[t];
    spl_team_barrier();
#pragma unroll
    for (int t = 0; t < NTh; ++t)
        tr[t] = (f32x4){xch[(0 * NTh + t) * 64 + lane].x, xch[(1 * NTh + t) * 64 + lane].y, xch[(2 * NTh + t) * 64 + lane].z, xch[(3 * NTh + t) * 64 + lane].w};
    if constexpr (TAILB) spl_team_barrier();
    CF_STAMP(3, tr[0].x)
    return ld;
}

// ---- the proposal kernel's 8-walker form (spline_kernels.h: spline_mh_kernel_pair) ---------------------------------------------
// spl_coupling<NTh, NH, INV, 4> on an 8-walker tile whose walkers sit in BOTH halves of the 16 columns: wave wv owns the super-tiles
// s = wv + 4t (register wv of tile t) and takes them two at a time -- tile t in the low half of the columns, tile t + 1 in the high
// half -- so ONE spline evaluation per lane serves two super-tiles (the matrix work per super-tile is unchanged; the spline
// arithmetic, the bulk of the instruction stream, halves).  Both halves leave with the full transformed half (results swapped
// across) and with the same log-det partial, summed in the order spl_coupling sums it: the walkers' values are those of the
// 16-walker form up to the multiply-add contractions hipcc picks per kernel.
// (The last layer's fragments are loaded where they are used.  Requesting them a coupling ahead, as the training kernel's forward pass
// does, was measured here twice and is WORSE -- 38.4 us per inverse against 23.1 in tools/spline_inv_probe.hip, 6.4 ms per launch
// against 5.9 in tools/spline_mh_probe.hip: hipcc parks the early fragments in accumulation registers and waits for them on the way.
// Sending them to a per-wave LDS slot by LDS-DMA instead (global_load_lds, 12 x 1 KB per coupling, issued in front of the spline
// arithmetic, read back with ds_read_b32) gave the same chains and 6.05 ms: no gain either -- the step is a chain of dependent
// vector instructions on a lone wave, not a queue of exposed loads.)
// `trunk`: the conditioner's hidden part (spl_hidden's image: the first spl_cond_hidden_floats of `net`) -- the workgroup's copy in
// LDS where the kernel keeps one (all four waves read the same 4 KB per coupling: from L2 that is a 0.7 us round trip in front of
// every trunk; the last layer's 9 KB per wave and coupling do not fit and stay in L2)
template <int NTh, int NH, bool INV, bool TAILB = true>   // TAILB = false: the caller alternates exchange buffers (spl_coupling_pair)
__device__ __forceinline__ float spl_coupling_halves(const float *__restrict__ net, const float *trunk, int S, int n_out, float tail, int lane,
                                                     const f32x4 (&cond)[NTh], f32x4 (&tr)[NTh], int wv, f32x4 *xch) {
    const int g = lane >> 4;
    const bool lo = (lane & 15) < 8;
    f32x4 h[NH];
    spl_hidden<NTh, NH>(trunk, lane, cond, h);
    const float *L4 = net + spl_cond_hidden_floats(NTh, NH);
    const float *b4 = L4 + (size_t)S * SPL_QT * NH * 256;
    float ld = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r != wv) continue;  // uniform over the wave
#pragma unroll
        for (int k = 0; k < NTh; k += 2) {
            const int sA = 4 * k + r, sB = 4 * (k + 1) + r;
            if (sA >= S) continue;
            const bool hasB = (k + 1 < NTh) && sB < S;
            f32x4 raw[SPL_QT];
            spl_raw<NH>(L4, b4, sA, lane, h, raw);
            float x = reg_of(tr[k], r);
            if (k + 1 < NTh) {
                if (hasB) {
                    f32x4 rawB[SPL_QT];
                    spl_raw<NH>(L4, b4, sB, lane, h, rawB);
#pragma unroll
                    for (int q = 0; q < SPL_QT; ++q) raw[q] = sel4(lo, raw[q], rawB[q]);
                }
                x = lo ? x : reg_of(tr[(k + 1 < NTh) ? k + 1 : k], r);
            }
            const bool valid = lo ? (4 * sA + g < n_out) : (hasB && 4 * sB + g < n_out);
            float l = 0.f;
#ifdef PROBE_NOEVAL   // (tools/spline_inv_probe.hip -DPROBE_NOEVAL: the inverse without its spline arithmetic -- never defined in the library)
            const float y = x + raw[0].x + raw[1].y + raw[2].z + raw[3].w + raw[4].x + raw[5].y;
#else
            const float y = spl_rqs<INV>(raw, tail, x, l);
#endif
            const float yo = valid ? y : 0.f, yp = half_swap(yo);
            const float lv = valid ? l : 0.f, lp = half_swap(lv);
            ld += lo ? lv : lp;   // super-tile sA's share, then sB's: spl_coupling's order, in both halves
            ld += lo ? lp : lv;
            set_reg(tr[k], r, lo ? yo : yp);
            if (k + 1 < NTh && hasB) set_reg(tr[(k + 1 < NTh) ? k + 1 : k], r, lo ? yp : yo);
        }
    }
    // super-tile s = 4t + r (register r of tile t) comes from wave r
#pragma unroll
    for (int t = 0; t < NTh; ++t) xch[(wv * NTh + t) * 64 + lane] = tr[t];
    spl_team_barrier();
#pragma unroll
    for (int t = 0; t < NTh; ++t)
        tr[t] = (f32x4){xch[(0 * NTh + t) * 64 + lane].x, xch[(1 * NTh + t) * 64 + lane].y, xch[(2 * NTh + t) * 64 + lane].z, xch[(3 * NTh + t) * 64 + lane].w};
    if constexpr (TAILB) spl_team_barrier();
    return ld;
}

// the workgroup's LDS copy of the conditioners' hidden parts: [block][f1 | f2][spl_cond_hidden_floats] (B x 2 x 4.3 KB at x_dim 50)
template <int NTh, int NH>
__device__ __forceinline__ void spline_stage_trunks(const float *__restrict__ img, const SplineShape &s, float *trunks, int tid, int nthreads) {
    constexpr int TF = NH * NTh * 256 + 2 * NH * NH * 256 + 3 * 16 * NH;
    for (int i = tid; i < s.B * 2 * TF; i += nthreads) {
        const int c = i / TF, j = i - c * TF;
        const float *net = img + (size_t)(c >> 1) * s.blk_floats + 2 * s.aff_floats + ((c & 1) ? s.f1_floats : 0);
        trunks[i] = net[j];
    }
}

// spl_affine with its output tiles dealt out over the four waves of the team (tile `to` to wave `to & 3`) and exchanged through LDS
// (`xch`: at least 2 NTh x 64 f32x4): every wave of the team form repeats all (2 NTh)^2 x 4 matrix instructions -- 64 of them, 0.85 us,
// per block at x_dim 50.  Same accumulation order as spl_affine: the same values.
template <int NTh, bool TAILB = true>
__device__ __forceinline__ void spl_affine_team(const float *__restrict__ aff, int lane, int wv, f32x4 *xch, const f32x4 (&in)[2][NTh],
                                                f32x4 (&out)[2][NTh]) {
    constexpr int T2 = 2 * NTh;
    const int g = lane >> 4;
    const float *bias = aff + T2 * T2 * 256;
#pragma unroll
    for (int to = 0; to < T2; ++to) {
        if ((to & 3) != wv) continue;  // uniform over the wave
        float wf[T2 * 4];
        load_frags<T2 * 4>(aff + (size_t)to * T2 * 256, lane, wf);
        f32x4 acc0 = *reinterpret_cast<const f32x4 *>(bias + (to * 4 + g) * 4);
        f32x4 acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < T2; ++ti) {
            const f32x4 v = in[ti / NTh][ti % NTh];
            acc0 = mfma4(wf[ti * 4 + 0], v.x, acc0);
            acc1 = mfma4(wf[ti * 4 + 1], v.y, acc1);
            acc0 = mfma4(wf[ti * 4 + 2], v.z, acc0);
            acc1 = mfma4(wf[ti * 4 + 3], v.w, acc1);
        }
        xch[to * 64 + lane] = acc0 + acc1;
    }
    spl_team_barrier();
#pragma unroll
    for (int to = 0; to < T2; ++to) out[to / NTh][to % NTh] = xch[to * 64 + lane];
    if constexpr (TAILB) spl_team_barrier();
}

// the inverse on an 8-walker tile held in both halves of the columns (spl_coupling_halves), four waves per tile
// `trunks`: NULL, or the workgroup's LDS copy of every conditioner's hidden part, [block][f1 | f2][spl_cond_hidden_floats]
// (spline_stage_trunks)
// (round 6) `xch`: TWO exchange buffers of 4 NTh x 64 f32x4 back to back, `xsel` which of them the next exchange takes: the nine
// exchanges of a pass alternate between the two and keep ONE barrier each (between an exchange's writes and its reads) -- the
// barrier that kept the next exchange's writes off this one's reads is the next exchange's own
template <int NTh, int NH>
__device__ __forceinline__ float spline_inverse_tile_halves(const float *__restrict__ img, const SplineShape &s, int lane, f32x4 (&xs)[2][NTh],
                                                            int wv, f32x4 *xch, int &xsel, const float *trunks = nullptr) {
    auto next_xch = [&]() -> f32x4 * { f32x4 *p = xch + ((xsel & 1) ? 4 * NTh * 64 : 0); xsel ^= 1; return p; };
    float ld = 0.f;
    constexpr int TF = NH * NTh * 256 + 2 * NH * NH * 256 + 3 * 16 * NH;   // spl_cond_hidden_floats(NTh, NH)
    for (int b = s.B - 1; b >= 0; --b) {
        const int nu = s.nu, nl = s.nl, SU = s.SU, SL = s.SL;
        const float *blk = img + (size_t)b * s.blk_floats;
        const float *f1 = blk + 2 * s.aff_floats, *f2 = f1 + s.f1_floats;
        const float *t1 = trunks ? trunks + (size_t)(2 * b) * TF : f1, *t2 = trunks ? trunks + (size_t)(2 * b + 1) * TF : f2;
        ld += spl_coupling_halves<NTh, NH, true, false>(f2, t2, SL, nl, s.tail, lane, xs[1], xs[0], wv, next_xch());  // networks.py:605-614
        ld += spl_coupling_halves<NTh, NH, true, false>(f1, t1, SU, nu, s.tail, lane, xs[0], xs[1], wv, next_xch());  // :615-621
        f32x4 y[2][NTh];
        spl_affine_team<NTh, false>(blk + s.aff_floats, lane, wv, next_xch(), xs, y);
        if (lane < 16 && wv == 0) ld -= (f2 + s.f2_floats)[0];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < NTh; ++t) xs[c][t] = y[c][t];
    }
    return ld;
}

// ---- reverse mode through one axis' knot construction (spl_knots): gradient wrt the 8 logits -----------------------
// edge_k = -B + 2B sum_{i<k} w_i, size_k = 2B w_k, w = m + (1 - mK) softmax(2B softmax(logits))
// g_edge = dLoss/d(edge of the selected bin), g_size = dLoss/d(size of the selected bin)
__device__ __forceinline__ void spl_knots_bwd(const float (&logits)[SPL_K], float tail, int bin, float g_edge, float g_size, float (&g_logits)[SPL_K]) {
    float a[SPL_K], u[SPL_K], p[SPL_K];
    spl_softmax8(logits, a);
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) u[k] = 2.f * tail * a[k];
    spl_softmax8(u, p);
    float gp[SPL_K], dot = 0.f;
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) {
        const float gw = 2.f * tail * ((k < bin ? g_edge : 0.f) + (k == bin ? g_size : 0.f));
        gp[k] = (1.f - 1e-3f * SPL_K) * gw;
        dot += p[k] * gp[k];
    }
    float ga[SPL_K], dot2 = 0.f;
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) {
        const float gu = p[k] * (gp[k] - dot);
        ga[k] = 2.f * tail * gu;
        dot2 += a[k] * ga[k];
    }
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) g_logits[k] = a[k] * (ga[k] - dot2);
}

// the same for the two axes at once on packed operands, from the softmax outputs the forward construction left (spl_knots2's a, p)
// instead of two more softmaxes per axis: 32 exponentials and ~200 other instructions less per reverse-mode evaluation
__device__ __forceinline__ void spl_knots_bwd2(const f32x2 (&a)[SPL_K], const f32x2 (&p)[SPL_K], float tail, int bin, f32x2 g_edge, f32x2 g_size,
                                               f32x2 (&g_logits)[SPL_K]) {
    const f32x2 zero = (f32x2){0.f, 0.f};
    f32x2 gp[SPL_K], dot = zero;
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) {
        const f32x2 gw = (2.f * tail) * ((k < bin ? g_edge : zero) + (k == bin ? g_size : zero));
        gp[k] = (1.f - 1e-3f * SPL_K) * gw;
        dot += p[k] * gp[k];
    }
    f32x2 ga[SPL_K], dot2 = zero;
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) {
        const f32x2 gu = p[k] * (gp[k] - dot);
        ga[k] = (2.f * tail) * gu;
        dot2 += a[k] * ga[k];
    }
#pragma unroll
    for (int k = 0; k < SPL_K; ++k) g_logits[k] = a[k] * (ga[k] - dot2);
}

__device__ __forceinline__ float spl_sigmoid(float v) { return spl_rcp(1.f + spl_exp(-v)); }

// Forward RQ spline of one scalar with its reverse mode: given gy = dLoss/dy and gl = dLoss/d(log|dy/dx|), returns
// dLoss/dx and dLoss/d(raw[24]) (0 outside the interval, where the map is the identity).  y and the log-derivative are
// returned as in spl_rqs<false>.
__device__ __forceinline__ float spl_rqs_fwd_bwd(const f32x4 (&raw)[SPL_QT], float tail, float x, float gy, float gl, float &y, float &lad,
                                                 f32x4 (&graw)[SPL_QT]) {
    float lw[SPL_K] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w, raw[1].x, raw[1].y, raw[1].z, raw[1].w};
    float lh[SPL_K] = {raw[2].x, raw[2].y, raw[2].z, raw[2].w, raw[3].x, raw[3].y, raw[3].z, raw[3].w};
    float ldv[SPL_K - 1] = {raw[4].x, raw[4].y, raw[4].z, raw[4].w, raw[5].x, raw[5].y, raw[5].z};
    const bool inside = x >= -tail && x <= tail;
    float cw[SPL_K + 1], wd[SPL_K], ch[SPL_K + 1], ht[SPL_K];
    f32x2 sm_a[SPL_K], sm_p[SPL_K];   // the knot constructions' softmax outputs, {width, height}: kept for the reverse mode
    {
        f32x2 l2[SPL_K], e2[SPL_K + 1], s2[SPL_K];
#pragma unroll
        for (int k = 0; k < SPL_K; ++k) l2[k] = (f32x2){lw[k], lh[k]};
        spl_knots2(l2, tail, e2, s2, sm_a, sm_p);
#pragma unroll
        for (int k = 0; k <= SPL_K; ++k) { cw[k] = e2[k].x; ch[k] = e2[k].y; }
#pragma unroll
        for (int k = 0; k < SPL_K; ++k) { wd[k] = s2[k].x; ht[k] = s2[k].y; }
    }
    int bin = -1;
#pragma unroll
    for (int k = 0; k <= SPL_K; ++k) {
        float e = cw[k];
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
    // the two knot derivatives of the selected bin and d(knot)/d(logit) = sigmoid(softplus(v)) sigmoid(v)
    float v0 = ldv[0], v1 = ldv[0];
#pragma unroll
    for (int i = 1; i < SPL_K - 1; ++i) { v0 = (bin == i + 1) ? ldv[i] : v0; v1 = (bin + 1 == i + 1) ? ldv[i] : v1; }
    const bool in0 = bin >= 1, in1 = bin + 1 <= SPL_K - 1;  // inner knots carry a parameter, the end knots are 1
    const float s0 = spl_softplus(v0), s1 = spl_softplus(v1);
    const float d0 = in0 ? 1e-3f + spl_softplus(s0) : 1.0f, d1 = in1 ? 1e-3f + spl_softplus(s1) : 1.0f;
    const float dd0 = in0 ? spl_sigmoid(s0) * spl_sigmoid(v0) : 0.f, dd1 = in1 ? spl_sigmoid(s1) * spl_sigmoid(v1) : 0.f;
    // forward (networks.py:541-556)
    const float ribw = spl_rcp(ibw);
    const float delta = ih * ribw;
    const float theta = (x - icw) * ribw;
    const float tomt = theta * (1.f - theta);
    const float sdd = d0 + d1 - 2.f * delta;
    const float Nn = ih * (delta * theta * theta + d0 * tomt);
    const float Dn = delta + sdd * tomt;
    const float rDn = spl_rcp(Dn);
    const float Q = d1 * theta * theta + 2.f * delta * tomt + d0 * (1.f - theta) * (1.f - theta);
    const float dn = delta * delta * Q;
    y = inside ? ich + Nn * rDn : x;
    lad = inside ? spl_log(dn) - 2.f * spl_log(Dn) : 0.f;
    // reverse
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
    const float gx = g_theta * ribw;
    const float g_icw = -g_theta * ribw;
    g_ibw += -g_theta * theta * ribw;
    float glw[SPL_K], glh[SPL_K];
    {
        f32x2 gl2[SPL_K];
        spl_knots_bwd2(sm_a, sm_p, tail, bin, (f32x2){g_icw, g_ich}, (f32x2){g_ibw, g_ih}, gl2);
#pragma unroll
        for (int k = 0; k < SPL_K; ++k) { glw[k] = gl2[k].x; glh[k] = gl2[k].y; }
    }
    float gld[SPL_K];  // index k-1 for inner knot k
#pragma unroll
    for (int k = 1; k < SPL_K; ++k) gld[k - 1] = ((bin == k) ? g_d0 * dd0 : 0.f) + ((bin + 1 == k) ? g_d1 * dd1 : 0.f);
    gld[SPL_K - 1] = 0.f;
    const float m = inside ? 1.f : 0.f;
    graw[0] = (f32x4){glw[0], glw[1], glw[2], glw[3]} * m; graw[1] = (f32x4){glw[4], glw[5], glw[6], glw[7]} * m;
    graw[2] = (f32x4){glh[0], glh[1], glh[2], glh[3]} * m; graw[3] = (f32x4){glh[4], glh[5], glh[6], glh[7]} * m;
    graw[4] = (f32x4){gld[0], gld[1], gld[2], gld[3]} * m; graw[5] = (f32x4){gld[4], gld[5], gld[6], 0.f} * m;
    return inside ? gx : gy;
}

}  // namespace nnest
