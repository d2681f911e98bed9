// solo_tile.h -- the one-walker-per-wave ("solo") layer primitives: a 16 -> 16 layer as a chain of v_fmac_f32 with a DPP row
// rotation on the activation operand, both nets of a coupling block in one wave (nnest_solo.hip has the layout and why).
// Shared by the proposal kernel (nnest_solo.hip: K4) and, since round 4, the row-per-wave training kernel (nnest_train_rows.h: K5).
#pragma once
#include "flow_tile.h"

namespace nnest {

static __device__ __forceinline__ void solo_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N>  // lane p of every 16-lane row <- lane (p - N) & 15 of the same row
static __device__ __forceinline__ float solo_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true));
}
// sum of the two K-halves: rows (n, 0) and (n, 1) both end with row(n,0) + row(n,1) -- the same addition in both, so the
// copies stay bit-identical.  (v_permlane16_swap with both operands the same value: hipcc 7.2 folds the two results of the
// builtin, hence the asm; s_nop 1: the swap reads VGPRs a VALU instruction may just have written.)
static __device__ __forceinline__ float solo_join(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// The same join for a hidden layer's pre-activation, with the NEXT layer's view of it built in: the h = 1 rows consume their
// layer input rotated by 8 (solo_rot8_h1), and the copy the swap needs anyway can be that rotation -- rows (n, 1) then end with
// ror8(row(n,0)) + ror8(row(n,1)), bit for bit ror8 of what rows (n, 0) hold.  Saves the row-masked move and its two wait states
// per hidden layer (activations are elementwise, so they commute with the rotation).
static __device__ __forceinline__ float solo_join_rot(float v) {
    float a = v, b;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "=&v"(b));
    return a + b;
}
// (value held by the scale half, value held by the translate half), in every lane
static __device__ __forceinline__ void solo_nets(float v, float &from_scale, float &from_translate) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    from_scale = a;
    from_translate = b;
}
// sum over the 16 positions of a row, bit-identical in its 16 lanes (each stage adds a pair that both partners see)
static __device__ __forceinline__ float solo_row_sum(float v) {
    v = v + solo_ror<8>(v);
    v = v + solo_ror<4>(v);
    v = v + solo_ror<2>(v);
    v = v + solo_ror<1>(v);
    return v;
}

static __device__ __forceinline__ float solo_lane0(float v) {  // position 0's value (every row holds the same copy)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
}

template <int U>
struct SoloNet {  // this lane's share of one coupling block (its net n, its K-half h), num_layers = 1, hidden 16
    float w1[U][8], w2[8], w3[U][8];
    float b1, b2, b3[U];  // biases ride in the h = 0 half; 0 in the h = 1 half
};

// gather from the packed (block, net) region (LDS copy), state_dict layout W0[H][D] b0[H] W1[H][H] b1[H] Wo[D][H] bo[D]
// (nnest/networks.py:271-282).  cc / ct: conditioning / transformed parity class of the block.
// The scale net's hidden layers (the lanes below 32) are stored times 2 log2(e): their sums then are the argument of v_exp_f32
// itself -- tanh(a) = 1 - 2 / (2^{a 2 log2 e} + 1), solo_activate -- one multiply less per activation on the step's serial chain
// (the translate half's relu lanes keep their weights as they are).
constexpr float SOLO_TANH_PRESCALE = 2.8853900817779268f;
template <int U>
static __device__ __forceinline__ void solo_gather(SoloNet<U> &n, const float *p, int D, int cc, int ct, int lane) {
    const int H = 16;
    const int pos = lane & 15, h = (lane >> 4) & 1;
    const float cs = lane < 32 ? SOLO_TANH_PRESCALE : 1.0f;
    const int pb0 = H * D, pW1 = pb0 + H, pb1 = pW1 + H * H, pWo = pb1 + H, pbo = pWo + D * H;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int q = (pos - t - 8 * h) & 15;  // position whose value this lane consumes at rotation t
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int d = 2 * U * q + 2 * u + cc;
            n.w1[u][t] = d < D ? cs * p[pos * D + d] : 0.f;
            const int dO = 2 * U * pos + 2 * u + ct;  // dim of this lane's transformed slot u
            n.w3[u][t] = dO < D ? p[pWo + dO * H + q] : 0.f;
        }
        n.w2[t] = cs * p[pW1 + pos * H + q];
    }
    n.b1 = h == 0 ? cs * p[pb0 + pos] : 0.f;
    n.b2 = h == 0 ? cs * p[pb1 + pos] : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int dO = 2 * U * pos + 2 * u + ct;
        n.b3[u] = (h == 0 && dO < D) ? p[pbo + dO] : 0.f;
    }
}

// ---- a layer's chain of multiply-adds as ONE asm statement ----------------------------------------------------------------
// hipcc 7.2 does not fold update_dpp into a multiply-add (it emits v_mov_b32_dpp + v_pk_fma_f32), so the v_fmac_f32_dpp are
// inline asm; written statement by statement, hipcc puts an s_nop between every two of them -- it has to assume that an
// inline asm reading a VGPR the previous instruction wrote might be a DPP read of it (two wait states) -- which at the one
// instruction per 4 cycles a lone wave issues is a fifth of the layer.  Inside one statement the hazards are ours: the only DPP
// source is the layer's input, written before the statement; the accumulators are plain VOP2 operands, which need no wait states.
#define SOLO_D(acc, x, w, t) "v_fmac_f32_dpp %" #acc ", %" #x ", %" #w " row_ror:" #t " row_mask:0xf bank_mask:0xf\n\t"
#define SOLO_F(acc, x, w) "v_fmac_f32 %" #acc ", %" #x ", %" #w "\n\t"

#define SOLO_M(acc, x, w, t) "v_mul_f32_dpp %" #acc ", %" #x ", %" #w " row_ror:" #t " row_mask:0xf bank_mask:0xf\n\t"
// Hazards inside a statement are ours: a DPP read needs two wait states behind the VALU write of its source (the layer input,
// written right in front of the statement).  The two-accumulator chains open with two plain (non-DPP) multiply-adds, which ARE
// those wait states; the one-input chains open with one, so they carry an s_nop 0.
// The chains START their accumulators -- a0 = bias + w[0] x (v_fma_f32), a1 = the first product (v_mul_f32) -- instead of
// adding to registers the caller initialised: a layer's bias is loop-invariant, so a "+v" accumulator cost a v_mov_b32 per
// accumulator and layer (18 per step) on a wave that issues one instruction per 4 cycles.  The *_acc forms add to what is there.
// a0 = b0 + sum_t w0[t] x0[(p - t) & 15],  a1 = sum_t w1[t] x1[(p - t) & 15]   (t = 0..7)
static __device__ __forceinline__ void solo_chain_2in(float &a0, float &a1, float b0, float x0, float x1, const float (&w0)[8], const float (&w1)[8]) {
    asm(
        "v_fma_f32 %0, %2, %4, %20\n\t" "v_mul_f32 %1, %3, %12\n\t"
        SOLO_D(0, 2, 5, 1) SOLO_D(1, 3, 13, 1) SOLO_D(0, 2, 6, 2) SOLO_D(1, 3, 14, 2) SOLO_D(0, 2, 7, 3) SOLO_D(1, 3, 15, 3)
        SOLO_D(0, 2, 8, 4) SOLO_D(1, 3, 16, 4) SOLO_D(0, 2, 9, 5) SOLO_D(1, 3, 17, 5) SOLO_D(0, 2, 10, 6) SOLO_D(1, 3, 18, 6)
        SOLO_D(0, 2, 11, 7) SOLO_D(1, 3, 19, 7)
        : "=&v"(a0), "=&v"(a1)
        : "v"(x0), "v"(x1), "v"(w0[0]), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]), "v"(w0[4]), "v"(w0[5]), "v"(w0[6]), "v"(w0[7]),
          "v"(w1[0]), "v"(w1[1]), "v"(w1[2]), "v"(w1[3]), "v"(w1[4]), "v"(w1[5]), "v"(w1[6]), "v"(w1[7]), "v"(b0));
}
static __device__ __forceinline__ void solo_chain_2in_acc(float &a0, float &a1, float x0, float x1, const float (&w0)[8], const float (&w1)[8]) {
    asm(
        SOLO_F(0, 2, 4) SOLO_F(1, 3, 12)
        SOLO_D(0, 2, 5, 1) SOLO_D(1, 3, 13, 1) SOLO_D(0, 2, 6, 2) SOLO_D(1, 3, 14, 2) SOLO_D(0, 2, 7, 3) SOLO_D(1, 3, 15, 3)
        SOLO_D(0, 2, 8, 4) SOLO_D(1, 3, 16, 4) SOLO_D(0, 2, 9, 5) SOLO_D(1, 3, 17, 5) SOLO_D(0, 2, 10, 6) SOLO_D(1, 3, 18, 6)
        SOLO_D(0, 2, 11, 7) SOLO_D(1, 3, 19, 7)
        : "+v"(a0), "+v"(a1)
        : "v"(x0), "v"(x1), "v"(w0[0]), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]), "v"(w0[4]), "v"(w0[5]), "v"(w0[6]), "v"(w0[7]),
          "v"(w1[0]), "v"(w1[1]), "v"(w1[2]), "v"(w1[3]), "v"(w1[4]), "v"(w1[5]), "v"(w1[6]), "v"(w1[7]));
}
// a0 = b0 + sum_t w0[t] x[(p - t) & 15],  a1 = b1 + sum_t w1[t] x[(p - t) & 15]   (one input, two outputs)
static __device__ __forceinline__ void solo_chain_2out(float &a0, float &a1, float b0, float b1, float x, const float (&w0)[8], const float (&w1)[8]) {
    asm(
        "v_fma_f32 %0, %2, %3, %19\n\t" "v_fma_f32 %1, %2, %11, %20\n\t"
        SOLO_D(0, 2, 4, 1) SOLO_D(1, 2, 12, 1) SOLO_D(0, 2, 5, 2) SOLO_D(1, 2, 13, 2) SOLO_D(0, 2, 6, 3) SOLO_D(1, 2, 14, 3)
        SOLO_D(0, 2, 7, 4) SOLO_D(1, 2, 15, 4) SOLO_D(0, 2, 8, 5) SOLO_D(1, 2, 16, 5) SOLO_D(0, 2, 9, 6) SOLO_D(1, 2, 17, 6)
        SOLO_D(0, 2, 10, 7) SOLO_D(1, 2, 18, 7)
        : "=&v"(a0), "=&v"(a1)
        : "v"(x), "v"(w0[0]), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]), "v"(w0[4]), "v"(w0[5]), "v"(w0[6]), "v"(w0[7]),
          "v"(w1[0]), "v"(w1[1]), "v"(w1[2]), "v"(w1[3]), "v"(w1[4]), "v"(w1[5]), "v"(w1[6]), "v"(w1[7]), "v"(b0), "v"(b1));
}
// a0 = b0 + sum_{t even} w[t] x[(p - t) & 15],  a1 = sum_{t odd} ...   (one input, one output over two accumulators)
static __device__ __forceinline__ void solo_chain_1(float &a0, float &a1, float b0, float x, const float (&w)[8]) {
    asm("s_nop 0\n\t"
        "v_fma_f32 %0, %2, %3, %11\n\t" SOLO_M(1, 2, 4, 1) SOLO_D(0, 2, 5, 2) SOLO_D(1, 2, 6, 3) SOLO_D(0, 2, 7, 4) SOLO_D(1, 2, 8, 5) SOLO_D(0, 2, 9, 6)
        SOLO_D(1, 2, 10, 7)
        : "=&v"(a0), "=&v"(a1)
        : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(b0));
}
static __device__ __forceinline__ void solo_chain_1_acc(float &a0, float &a1, float x, const float (&w)[8]) {
    asm("s_nop 0\n\t"
        SOLO_F(0, 2, 3) SOLO_D(1, 2, 4, 1) SOLO_D(0, 2, 5, 2) SOLO_D(1, 2, 6, 3) SOLO_D(0, 2, 7, 4) SOLO_D(1, 2, 8, 5) SOLO_D(0, 2, 9, 6)
        SOLO_D(1, 2, 10, 7)
        : "+v"(a0), "+v"(a1)
        : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
}
// the h = 1 rows' view of a layer input: rotated by 8 in rows 1 and 3, unchanged in rows 0 and 2 (the s_nop 1 at the head of
// the chain that consumes it is the DPP read's wait)
static __device__ __forceinline__ float solo_rot8_h1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x128, 0xa, 0xf, false));
}
// tanh in the scale half, relu in the translate half, without a branch (hipcc turns the plain conditional into a divergent
// branch: both sides then run one after the other under exec masks, plus the mask bookkeeping): `sel` is all ones in the
// translate half; v_bfi_b32 picks the bits
static __device__ __forceinline__ float solo_activate(float v, unsigned sel) {
    // (scale half: v arrives times 2 log2(e), solo_gather)
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v) + 1.0f), rl = fmaxf(v, 0.f);
    unsigned out;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(out) : "v"(sel), "v"(__float_as_uint(rl)), "v"(__float_as_uint(th)));
    return __uint_as_float(out);
}
// v_permlane16_swap of two DIFFERENT registers: rows (n, 1) of `a` trade places with rows (n, 0) of `b`
static __device__ __forceinline__ void solo_swap16(float &a, float &b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
}

// CouplingLayer.inverse (networks.py:300-309), both nets at once (the wave's two halves); returns the lane's log-det partial.
// U = 2: the last layer's two outputs are reduce-SCATTERED over the K-halves (one swap: row h ends with output u = h), so the
// affine update runs once per lane on the slot its row owns and one more swap hands both results to both rows; the log-det
// partial is then per row (the caller sums rows h = 0 and 1, solo_logdet_total).
template <int U>
static __device__ __forceinline__ float solo_coupling_inverse(const SoloNet<U> &w, unsigned sel, bool h1, const float (&cond)[U],
                                                              float (&trans)[U]) {
    float a0, a1;
    if constexpr (U == 2) {
        solo_chain_2in(a0, a1, w.b1, solo_rot8_h1(cond[0]), solo_rot8_h1(cond[1]), w.w1[0], w.w1[1]);
    } else {
        solo_chain_1(a0, a1, w.b1, solo_rot8_h1(cond[0]), w.w1[0]);
    }
    float hid = solo_activate(solo_join_rot(a0 + a1), sel);   // (h = 1 rows: rotated by 8, as the next chain reads it)
    solo_chain_1(a0, a1, w.b2, hid, w.w2);
    const float hin = solo_activate(solo_join_rot(a0 + a1), sel);
    if constexpr (U == 2) {
        float o0, o1;
        solo_chain_2out(o0, o1, w.b3[0], w.b3[1], hin, w.w3[0], w.w3[1]);
        solo_swap16(o0, o1);               // rows h = 0: both halves of output 0; rows h = 1: both halves of output 1
        float ls, tt;
        solo_nets(o0 + o1, ls, tt);
        const float cur = h1 ? trans[1] : trans[0];
        float nw = (cur - tt) * __expf(-ls);  // (inputs - t) * exp(-log_s)   networks.py:307-309
        float nb = nw;
        solo_swap16(nw, nb);               // every row: nw = slot 0's new value, nb = slot 1's
        trans[0] = nw;
        trans[1] = nb;
        return -ls;
    } else {
        float o0, o1;
        solo_chain_1(o0, o1, w.b3[0], hin, w.w3[0]);
        float ls, tt;
        solo_nets(solo_join(o0 + o1), ls, tt);
        trans[0] = (trans[0] - tt) * __expf(-ls);
        return h1 ? 0.f : -ls;             // the same value in both rows: counted once
    }
}
// log-det of the walker from the lanes' partials: over the 16 positions of a row, then over the two K-halves
static __device__ __forceinline__ float solo_logdet_total(float ld) { return solo_join(solo_row_sum(ld)); }

// ---- x_dim 97..128 (U = 4): a lane's share of a block is 78 weights, 234 for the three blocks -- more than the register file
// leaves beside the state.  They live in LDS instead, field-major ([block][field / 4][lane][4]: one conflict-free ds_read_b128
// per four fields; the four net waves of a workgroup hold identical copies, so ONE copy per workgroup, 60 KB), and a layer's
// weights are read into registers right before the layer's chain.  Fields: w1[u][t] at 8 u + t, w2[t] at 32 + t, w3[u][t] at
// 40 + 8 u + t, b1 72, b2 73, b3[u] 74 + u.
constexpr int SOLO4_NF = 80;
template <int U>   // the fields of absent slots stay unwritten (never read)
static __device__ __forceinline__ void solo4_store(float *base, int b, const SoloNet<U> &n, int lane) {
    float *q = base + (size_t)b * SOLO4_NF * 64;
    auto put = [&](int f, float v) { q[((size_t)(f >> 2) * 64 + lane) * 4 + (f & 3)] = v; };
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < 8; ++t) { put(8 * u + t, n.w1[u][t]); put(40 + 8 * u + t, n.w3[u][t]); }
#pragma unroll
    for (int t = 0; t < 8; ++t) put(32 + t, n.w2[t]);
    put(72, n.b1); put(73, n.b2);
#pragma unroll
    for (int u = 0; u < 4; ++u) put(74 + u, u < U ? n.b3[u < U ? u : 0] : 0.f);
    put(78, 0.f); put(79, 0.f);
}
static __device__ __forceinline__ void solo4_load8(float (&w)[8], const float *blk, int f0, int lane) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(blk + ((size_t)(f0 >> 2) * 64 + lane) * 4);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(blk + ((size_t)((f0 >> 2) + 1) * 64 + lane) * 4);
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
// CouplingLayer.inverse as solo_coupling_inverse<2>, four slots per class: the first layer takes its four inputs as two pairs
// on the same two accumulators, the last layer's four outputs are two reduce-scattered pairs (row h ends with outputs h and 2 + h)
struct Solo4Lds {   // a block's weights where they live: the workgroup's LDS copy ...
    const float *blk; int lane;
    __device__ __forceinline__ void load8(float (&w)[8], int f0) const { solo4_load8(w, blk, f0, lane); }
    __device__ __forceinline__ f32x4 bias(int c) const { return *reinterpret_cast<const f32x4 *>(blk + ((size_t)(18 + c) * 64 + lane) * 4); }
};
template <int U>
struct Solo4Reg {   // ... or this lane's registers (ONE of the three blocks: a third of the LDS traffic of a step)
    const SoloNet<U> &n;
    __device__ __forceinline__ void load8(float (&w)[8], int f0) const {
#pragma unroll
        for (int t = 0; t < 8; ++t) w[t] = f0 < 32 ? n.w1[(f0 >> 3) < U ? (f0 >> 3) : 0][t] : (f0 < 40 ? n.w2[t] : n.w3[((f0 - 40) >> 3) < U ? ((f0 - 40) >> 3) : 0][t]);
    }
    __device__ __forceinline__ f32x4 bias(int c) const {
        return c == 0 ? (f32x4){n.b1, n.b2, n.b3[0], U > 1 ? n.b3[U > 1 ? 1 : 0] : 0.f}
                      : (f32x4){U > 2 ? n.b3[U > 2 ? 2 : 0] : 0.f, U > 3 ? n.b3[U > 3 ? 3 : 0] : 0.f, 0.f, 0.f};
    }
};
// The coupling with its weights fetched through a source W (Solo4Lds / Solo4Reg), U = 1..4 -- the arithmetic, operation for
// operation, of solo_coupling_inverse<U> (U <= 2), so a launch whose weights live in LDS reproduces the register-resident one bit
// for bit.  U = 4: the first layer takes its four inputs as two pairs on the same two accumulators, the last layer's four outputs
// are two reduce-scattered pairs (row h ends with outputs h and 2 + h).  Odd U (x_dim 65..96, x_dim <= 32): the last slot goes
// through the one-slot chains -- its output whole in both rows, its log-det counted in the h = 0 rows.
template <int U, class W>
static __device__ __forceinline__ float solo_coupling_inverse4(const W &wsrc, unsigned sel, bool h1, const float (&cond)[U], float (&trans)[U]) {
    const f32x4 bA = wsrc.bias(0);   // b1 b2 b3[0] b3[1]
    const f32x4 bB = wsrc.bias(1);   // b3[2] b3[3] - -
    float wa[8], wb[8];
    float a0, a1;
    if constexpr (U >= 2) {
        wsrc.load8(wa, 0); wsrc.load8(wb, 8);
        solo_chain_2in(a0, a1, bA.x, solo_rot8_h1(cond[0]), solo_rot8_h1(cond[U >= 2 ? 1 : 0]), wa, wb);
    }
    if constexpr (U == 4) {
        wsrc.load8(wa, 16); wsrc.load8(wb, 24);
        solo_chain_2in_acc(a0, a1, solo_rot8_h1(cond[U == 4 ? 2 : 0]), solo_rot8_h1(cond[U == 4 ? 3 : 0]), wa, wb);
    }
    if constexpr (U == 1) {
        wsrc.load8(wa, 0);
        solo_chain_1(a0, a1, bA.x, solo_rot8_h1(cond[0]), wa);
    }
    if constexpr (U == 3) {
        wsrc.load8(wa, 16);
        solo_chain_1_acc(a0, a1, solo_rot8_h1(cond[U - 1]), wa);
    }
    float hid = solo_activate(solo_join_rot(a0 + a1), sel);   // (h = 1 rows: rotated by 8, as the next chain reads it)
    wsrc.load8(wa, 32);
    solo_chain_1(a0, a1, bA.y, hid, wa);
    const float hin = solo_activate(solo_join_rot(a0 + a1), sel);
    float ld = 0.f;
#pragma unroll
    for (int k = 0; k < U / 2; ++k) {
        float o0, o1;
        wsrc.load8(wa, 40 + 16 * k); wsrc.load8(wb, 48 + 16 * k);
        solo_chain_2out(o0, o1, k == 0 ? bA.z : bB.x, k == 0 ? bA.w : bB.y, hin, wa, wb);
        solo_swap16(o0, o1);               // rows h = 0: both halves of output 2 k; rows h = 1: both halves of output 2 k + 1
        float ls, tt;
        solo_nets(o0 + o1, ls, tt);
        const float cur = h1 ? trans[2 * k + 1] : trans[2 * k];
        float nw = (cur - tt) * __expf(-ls);  // (inputs - t) * exp(-log_s)   networks.py:307-309
        float nb = nw;
        solo_swap16(nw, nb);
        trans[2 * k] = nw;
        trans[2 * k + 1] = nb;
        ld -= ls;
    }
    if constexpr (U & 1) {
        float o0, o1;
        wsrc.load8(wa, 40 + 8 * (U - 1));
        solo_chain_1(o0, o1, U == 1 ? bA.z : bB.x, hin, wa);
        float ls, tt;
        solo_nets(solo_join(o0 + o1), ls, tt);
        trans[U - 1] = (trans[U - 1] - tt) * __expf(-ls);
        ld -= h1 ? 0.f : ls;               // the same value in both rows: counted once
    }
    return ld;
}

}  // namespace nnest
