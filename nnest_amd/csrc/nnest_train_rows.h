// nnest_train_rows.h -- K5 with ONE ROW OF THE MINIBATCH PER WAVE (round 4; included behind nnest_train_grid.h, inside namespace nnest).
//
// train_kernel_grid gives every 16-row tile of a minibatch a workgroup and runs its forward + backward pass on TWO of that
// workgroup's eight waves (scale net / translate net, MFMA 16x16x4 tiles, an LDS exchange per block and direction): 18.4 k of a
// config-2 minibatch's 35.6 k cycles on 14 waves of the chip (profiles/r03/k5_stamps.txt).  The chain is latency-bound exactly as
// K4's was, and the cure is the same (nnest_solo.hip): one ROW per wave, a layer as a chain of v_fmac_f32 with a DPP row rotation,
// both nets of a block in the wave's two halves, no LDS exchange and no workgroup barrier inside the pass.  A minibatch of 100
// rows is then 100 waves on 25 compute units (4 row waves per workgroup, one per SIMD; the workgroup's other 4 waves carry
// weight-gradient jobs and the row prefetch), and forward + backward is ~2 x K4's inverse + the transposed products.
//
// Everything around the pass is train_kernel_grid's, unchanged in what it computes:
//   F+B   wave (wg, w < 4): row 4 wg + w forward (activations kept in REGISTERS: one float per layer and net) and backward
//         (trainer.py:384-403; reverse mode through the affine couplings, networks.py:289-298); the per-row gradients G and
//         activations go to the global staging area [ct][row][16] -- the layout contract_rows_grid reads;
//   grid barrier;
//   W+A   the weight-gradient jobs, one per wave of the grid in train_kernel_grid's static order (job J on wave J / G of workgroup
//         J % G), the owner applies Adam in registers and publishes its new tile into the backward fragment image (+ biases);
//   R     every workgroup re-lays the published image into its two LDS "solo" images (forward and transposed: lane (net, K-half,
//         position) x field, solo_tile.h) through a destination map built once per launch.  The tiles are published as
//         {weight, tag} granules, so this phase polls the DATA (no second grid barrier in front of it: one cross-CU round trip less).
// Summation inside a layer is the solo order (K in two halves, left to right), not the MFMA order of train_kernel, so the two
// agree to rounding (tests/test_gpu_train.py::test_grid_training_vs_single_workgroup: 1e-4 of the largest element after 12 epochs);
// every reduction has a fixed order and every gradient element one producer: the kernel reproduces itself bit for bit, which
// is what replicated training across ranks relies on.
//
// Shapes: hidden_dim 16, num_layers 1, num_blocks 3 (the reference's defaults, nnest/sampler.py:37-43), x_dim <= 128, batch <= 128;
// everything else runs train_kernel_grid / train_kernel.

enum { ROWS_PER_WG = 4, ROWS_B = 3 };

template <int U> struct RowsKeep {   // what the backward pass needs of a block's forward pass, per lane
    float h1, h2;                    // hidden activations (rows h = 1: rotated by 8, as the chains read them)
    float ls[(U + 1) / 2], tt[(U + 1) / 2];   // log s, t of the slot this lane's row owns in pair k (odd U: the last slot, whole in both rows)
};

// float offset of field f of lane `lane` in block b of a solo image [B][SOLO4_NF / 4][64][4]
__device__ __forceinline__ int rows_img_off(int b, int f, int lane) { return b * SOLO4_NF * 64 + ((f >> 2) * 64 + lane) * 4 + (f & 3); }

// Where parameter `o` (offset inside its net's state_dict region W0[H][D] b0[H] W1[H][H] b1[H] Wo[D][H] bo[D], networks.py:271-282)
// of block b / net n sits in the forward solo image (df) and in the transposed one (db); -1 = nowhere (masked-out entries,
// biases in the transposed image).  scaled: the forward copy is stored times SOLO_TANH_PRESCALE (solo_gather).
// Forward fields as solo_gather / solo4_store; transposed image: fields 8 u + t = the product Wo^T (first layer of the backward
// pass: U inputs), 32 + t = W1^T, 40 + 8 u + t = W0^T (its last layer: U outputs).
template <int U>
__device__ inline void rows_param_dest(int D, int b, int n, int o, int &df, int &db, bool &scaled) {
    const int cc = (b + 1) & 1, ct = b & 1;
    const int pb0 = 16 * D, pW1 = pb0 + 16, pb1 = pW1 + 256, pWo = pb1 + 16, pbo = pWo + 16 * D;
    df = db = -1;
    scaled = false;
    auto lane_of = [&](int pos, int src) { const int dl = (pos - src) & 15; return 32 * n + 16 * (dl >> 3) + pos; };
    auto rot_of = [&](int pos, int src) { return ((pos - src) & 15) & 7; };
    if (o < pb0) {                       // W0[j][d]
        const int j = o / D, d = o % D;
        if ((d & 1) != cc) return;
        const int s = d >> 1, q = s / U, u = s % U;
        df = rows_img_off(b, 8 * u + rot_of(j, q), lane_of(j, q));
        db = rows_img_off(b, 40 + 8 * u + rot_of(q, j), lane_of(q, j));
        scaled = n == 0;
    } else if (o < pW1) {                // b0[j]
        df = rows_img_off(b, 72, 32 * n + (o - pb0));
        scaled = n == 0;
    } else if (o < pb1) {                // W1[j][k]
        const int j = (o - pW1) >> 4, k = (o - pW1) & 15;
        df = rows_img_off(b, 32 + rot_of(j, k), lane_of(j, k));
        db = rows_img_off(b, 32 + rot_of(k, j), lane_of(k, j));
        scaled = n == 0;
    } else if (o < pWo) {                // b1[j]
        df = rows_img_off(b, 73, 32 * n + (o - pb1));
        scaled = n == 0;
    } else if (o < pbo) {                // Wo[d][k]
        const int d = (o - pWo) >> 4, k = (o - pWo) & 15;
        if ((d & 1) != ct) return;
        const int s = d >> 1, q = s / U, u = s % U;
        df = rows_img_off(b, 40 + 8 * u + rot_of(q, k), lane_of(q, k));
        db = rows_img_off(b, 8 * u + rot_of(k, q), lane_of(k, q));
    } else {                             // bo[d]
        const int d = o - pbo;
        if ((d & 1) != ct) return;
        const int s = d >> 1, q = s / U, u = s % U;
        df = rows_img_off(b, 74 + u, 32 * n + q);
    }
}

// the refresh maps: for element i of the published backward fragment image (weights as bwd_image_src lays them out, biases where
// the forward image keeps them) its two destinations in a workgroup's LDS, packed in one word: low half = forward solo image
// offset | (1 << 14 if stored times the prescale), high half = IMG + transposed solo image offset; 0xffff = none
template <int U>
__global__ void rows_maps_kernel(int *__restrict__ maps, FlowShape s) {
    constexpr int IMG = ROWS_B * SOLO4_NF * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < s.image_floats; i += gridDim.x * blockDim.x) {
        int mf = -1, mb = -1;
        const int o = i % s.net_floats;
        const int p = o >= frag_off_b1(U, 1, 1) ? fwd_image_src(s, i) : bwd_image_src(s, i);
        if (p >= 0) {
            const int bn = p / s.net_params;
            int df, db; bool sc;
            rows_param_dest<U>(s.D, bn >> 1, bn & 1, p - bn * s.net_params, df, db, sc);
            if (df >= 0) mf = df | (sc ? 1 << 14 : 0);
            if (db >= 0) mb = IMG + db;
        }
        maps[i] = (mf & 0xffff) | (mb << 16);   // (offsets < 2 IMG = 30 720; none = 0xffff in its half)
    }
}

// ---- CouplingLayer.forward (networks.py:289-298) of one row, activations kept ----
template <int U>
static __device__ __forceinline__ float rows_block_forward(const Solo4Lds &w, unsigned sel, bool h1, const float (&cond)[U], float (&trans)[U], RowsKeep<U> &kp) {
    const f32x4 bA = w.bias(0);   // b1 b2 b3[0] b3[1]
    const f32x4 bB = w.bias(1);   // b3[2] b3[3] - -
    float wa[8], wb[8], a0, a1;
    if constexpr (U >= 2) {
        w.load8(wa, 0); w.load8(wb, 8);
        solo_chain_2in(a0, a1, bA.x, solo_rot8_h1(cond[0]), solo_rot8_h1(cond[U >= 2 ? 1 : 0]), wa, wb);
    }
    if constexpr (U == 4) {
        w.load8(wa, 16); w.load8(wb, 24);
        solo_chain_2in_acc(a0, a1, solo_rot8_h1(cond[U == 4 ? 2 : 0]), solo_rot8_h1(cond[U == 4 ? 3 : 0]), wa, wb);
    }
    if constexpr (U == 1) {
        w.load8(wa, 0);
        solo_chain_1(a0, a1, bA.x, solo_rot8_h1(cond[0]), wa);
    }
    if constexpr (U == 3) {
        w.load8(wa, 16);
        solo_chain_1_acc(a0, a1, solo_rot8_h1(cond[U - 1]), wa);
    }
    // (the next layer's weights are requested before this layer's activation: their LDS latency sits behind the join / exp / rcp)
    float w2[8], w3a[8], w3b[8];
    w.load8(w2, 32);
    kp.h1 = solo_activate(solo_join_rot(a0 + a1), sel);
    if constexpr (U >= 2) { w.load8(w3a, 40); w.load8(w3b, 48); }
    solo_chain_1(a0, a1, bA.y, kp.h1, w2);
    kp.h2 = solo_activate(solo_join_rot(a0 + a1), sel);
    float ld = 0.f;
#pragma unroll
    for (int k = 0; k < U / 2; ++k) {
        float o0, o1;
        if (k > 0) { w.load8(w3a, 40 + 16 * k); w.load8(w3b, 48 + 16 * k); }
        solo_chain_2out(o0, o1, k == 0 ? bA.z : bB.x, k == 0 ? bA.w : bB.y, kp.h2, w3a, w3b);
        solo_swap16(o0, o1);               // rows h = 0: both halves of output 2 k; rows h = 1: both halves of output 2 k + 1
        float ls, tt;
        solo_nets(o0 + o1, ls, tt);
        const float cur = h1 ? trans[2 * k + 1] : trans[2 * k];
        float nw = cur * __expf(ls) + tt;  // inputs * exp(log_s) + t   networks.py:295-297
        float nb = nw;
        solo_swap16(nw, nb);
        trans[2 * k] = nw;
        trans[2 * k + 1] = nb;
        ld += ls;
        kp.ls[k] = ls; kp.tt[k] = tt;
    }
    if constexpr (U & 1) {
        float o0, o1;
        w.load8(wa, 40 + 8 * (U - 1));
        solo_chain_1(o0, o1, U == 1 ? bA.z : bB.x, kp.h2, wa);
        float ls, tt;
        solo_nets(solo_join(o0 + o1), ls, tt);
        trans[U - 1] = trans[U - 1] * __expf(ls) + tt;
        ld += h1 ? 0.f : ls;               // the same value in both rows: counted once
        kp.ls[U / 2] = ls; kp.tt[U / 2] = tt;
    }
    return ld;
}

// g * act'(pre) from the post-activation a (act_grad, nnest_train.hip): tanh' = 1 - a^2 in the scale half, relu' = [a > 0] in the
// translate half, without a branch
static __device__ __forceinline__ float rows_act_grad(float g, float a, unsigned sel) {
    const float th = g * (1.f - a * a), rl = a > 0.f ? g : 0.f;
    unsigned out;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(out) : "v"(sel), "v"(__float_as_uint(rl)), "v"(__float_as_uint(th)));
    return __uint_as_float(out);
}

// one 16-float row of a staged tensor: lanes of the h = 0 rows (one per net) write position `col` of row `row` of column tile ct
static __device__ __forceinline__ void rows_stage(float *stg_net, int ct, int row, int col, float v, bool stager) {
    if (stager) st_sc1(stg_net + ((size_t)ct * TRAIN_MAX_ROWS + row) * 16 + col, v);
}
// a value per slot: the lane at position pos holds slots U pos .. U pos + U - 1 (column tile ct0 + slot / 16, column slot % 16).
// U = 4: the lane's four slots are one aligned 16-byte store.  (U = 1, 2 collecting the neighbours' values by DPP into 16-byte stores
// was measured SLOWER, 11.8 against 11.3 us per minibatch at x_dim 50: the drain in front of the barrier is not what the barrier waits for.)
template <int U>
static __device__ __forceinline__ void rows_stage_slots(float *stg_net, int ct0, int row, int pos, const float (&v)[U], bool stager) {
    if constexpr (U == 4) {
        const int s = 4 * pos;
        if (stager) st_sc1_f32x4(stg_net + ((size_t)(ct0 + (s >> 4)) * TRAIN_MAX_ROWS + row) * 16 + (s & 15), (f32x4){v[0], v[1], v[2], v[3]});
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = U * pos + u;
            rows_stage(stg_net, ct0 + (s >> 4), row, s & 15, v[u], stager);
        }
    }
}

// TAGGED staging (train_kernel_pipe): every staged value travels as an 8-byte {value, tag} granule: a reader that finds the
// minibatch's tag has the minibatch's value, whatever the order its loads were served in (MI355X_MICROARCH.md, "R2"), so the writer
// never waits for its stores.  Inside a column tile (128 rows x 16 columns) the granules are laid out for the READER, the
// weight-gradient job's v_mfma_f32_16x16x4: lane (gq, j) of MFMA k of row group `it` takes row 16 it + 4 k + gq, column j -- so the
// two rows k = 2 m, 2 m + 1 of a lane sit side by side, one 16-byte load for two MFMAs (56 loads per job; as 112 8-byte loads they
// ran over the 63 a wave can have in flight, measured: two latencies instead of one):
//     byte offset of (row r, column j) = 2048 (r >> 4) + 256 (4 m + gq) + 16 j + 8 p,   gq = r & 3, k = (r >> 2) & 3, m = k >> 1, p = k & 1.
// A store takes the block's base + the column tile in SCALAR registers and the lane's (net, row, column) as one 32-bit offset that
// is the same for every store of the launch (voff_pos; voff_slot for the per-slot tensors): no 64-bit address per store for hipcc to
// hoist and spill.
struct RowsTagged {
    float *blk;        // the block's staging region (uniform)
    int voff_pos;      // byte offset of (net, row, column = pos)
    int voff_slot;     // byte offset of (net, row, the lane's first slot: tile (U pos) >> 4, column (U pos) & 15)
    float tg;
};
constexpr int ROWS_CTB = TRAIN_MAX_ROWS * 16 * 8;   // bytes of one column tile of granules
static __device__ __forceinline__ int rows_tagged_row_off(int r) {   // the row's part of a granule's offset
    const int gq = r & 3, k = (r >> 2) & 3;
    return 2048 * (r >> 4) + 256 * (4 * (k >> 1) + gq) + 8 * (k & 1);
}
__device__ __forceinline__ void st_sc1_x2(float *p, float v, float tg) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ d = {v, tg};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(d) : "memory");
}
static __device__ __forceinline__ float pipe_opaque_f(float v) { asm volatile("" : "+v"(v)); return v; }
// one granule at byte offset voff (+ imm) behind the scalar base
template <int IMM>
static __device__ __forceinline__ void rows_granule(const float *sb, int voff, float v, float tg, bool stager) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x2_ d;   // (built from two scalars behind an empty asm: as a vector literal of a kept activation hipcc widened the KEPT value into memory)
    d.x = pipe_opaque_f(v);
    d.y = tg;
    if (stager) asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3 sc1" : : "v"(voff), "v"(d), "s"(sb), "n"(IMM) : "memory");
}
static __device__ __forceinline__ void rows_stage_t(const RowsTagged &t, int ct, float v, bool stager) {
    rows_granule<0>(t.blk + (size_t)ct * (ROWS_CTB / 4), t.voff_pos, v, t.tg, stager);
}
template <int U>
static __device__ __forceinline__ void rows_stage_slots_t(const RowsTagged &t, int ct0, const float (&v)[U], bool stager) {
    const float *sb = t.blk + (size_t)ct0 * (ROWS_CTB / 4);
    if constexpr (U == 3) {   // slots 3 pos .. 3 pos + 2 may straddle a column tile
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int col0 = (t.voff_slot >> 4) & 15;   // the first slot's column
            const int wrap = col0 + u > 15 ? ROWS_CTB - 256 : 0;
            typedef float f32x2_ __attribute__((ext_vector_type(2)));
            f32x2_ d;
            d.x = pipe_opaque_f(v[u]);
            d.y = t.tg;
            if (stager) asm volatile("global_store_dwordx2 %0, %1, %2 sc1" : : "v"(t.voff_slot + 16 * u + wrap), "v"(d), "s"(sb) : "memory");
        }
    } else {   // U = 1, 2, 4: the lane's slots are columns of one tile, 16 bytes apart
        rows_granule<0>(sb, t.voff_slot, v[0], t.tg, stager);
        if constexpr (U >= 2) rows_granule<16>(sb, t.voff_slot, v[U >= 2 ? 1 : 0], t.tg, stager);
        if constexpr (U == 4) {
            rows_granule<32>(sb, t.voff_slot, v[U == 4 ? 2 : 0], t.tg, stager);
            rows_granule<48>(sb, t.voff_slot, v[U == 4 ? 3 : 0], t.tg, stager);
        }
    }
}

// ---- reverse mode through one coupling block (block_backward_grid's arithmetic on one row) ----
// in: ytrans = the block's OUTPUT on the transformed side, gtrans / gcond = d loss / d (block outputs);
// out: ytrans = the block's input, gtrans = d loss / d (that input), gcond += the two nets' contributions
// STAGE_FWD = false: the activations and the conditioning input were staged by the forward pass (rows_stage_forward, nnest_train_pipe.h)
// TAGGED: the staged values as {value, tag} granules (rows_stage_t) with tag `tg`
template <int U, bool STAGE_FWD = true, bool TAGGED = false>
static __device__ __forceinline__ void rows_block_backward(const Solo4Lds &wb_, unsigned sel, bool h1, bool translate_half, int pos, int D, int ct,
                                                           bool row_ok, float gld, const float (&cond)[U], float (&ytrans)[U], float (&gcond)[U],
                                                           float (&gtrans)[U], const RowsKeep<U> &kp, float *stg_net, int row, bool stager, const RowsTagged &tgd = RowsTagged{}) {
    typedef StageMap<U, 1, 1> SM;
    float go[U];
#pragma unroll
    for (int k = 0; k < U / 2; ++k) {
        const int u = 2 * k + (h1 ? 1 : 0);
        const bool valid = row_ok && 2 * U * pos + 2 * u + ct < D;
        const float yv = h1 ? ytrans[2 * k + 1] : ytrans[2 * k], gv = h1 ? gtrans[2 * k + 1] : gtrans[2 * k];
        const float ymt = yv - kp.tt[k];
        const float g_ls = valid ? gv * ymt + gld : 0.f, g_t = valid ? gv : 0.f;
        float x0 = ymt * __expf(-kp.ls[k]), gx0 = gv * __expf(kp.ls[k]), gs0 = translate_half ? g_t : g_ls;
        float x1 = x0, gx1 = gx0, gs1 = gs0;
        solo_swap16(x0, x1);
        solo_swap16(gx0, gx1);
        solo_swap16(gs0, gs1);
        ytrans[2 * k] = x0; ytrans[2 * k + 1] = x1;
        gtrans[2 * k] = gx0; gtrans[2 * k + 1] = gx1;
        go[2 * k] = gs0; go[2 * k + 1] = gs1;
    }
    if constexpr (U & 1) {
        const bool valid = row_ok && 2 * U * pos + 2 * (U - 1) + ct < D;
        const float gv = gtrans[U - 1], ymt = ytrans[U - 1] - kp.tt[U / 2];
        const float g_ls = valid ? gv * ymt + gld : 0.f, g_t = valid ? gv : 0.f;
        ytrans[U - 1] = ymt * __expf(-kp.ls[U / 2]);
        gtrans[U - 1] = gv * __expf(kp.ls[U / 2]);
        go[U - 1] = translate_half ? g_t : g_ls;
    }
    if constexpr (TAGGED) rows_stage_slots_t<U>(tgd, SM::gout(0), go, stager);
    else rows_stage_slots<U>(stg_net, SM::gout(0), row, pos, go, stager);
    // g_h2 = Wo^T g_out   (the transposed image's first U field groups)
    float wa[8], wb[8], a0, a1;
    if constexpr (U >= 2) {
        wb_.load8(wa, 0); wb_.load8(wb, 8);
        solo_chain_2in(a0, a1, 0.f, solo_rot8_h1(go[0]), solo_rot8_h1(go[U >= 2 ? 1 : 0]), wa, wb);
    }
    if constexpr (U == 4) {
        wb_.load8(wa, 16); wb_.load8(wb, 24);
        solo_chain_2in_acc(a0, a1, solo_rot8_h1(go[U == 4 ? 2 : 0]), solo_rot8_h1(go[U == 4 ? 3 : 0]), wa, wb);
    }
    if constexpr (U == 1) {
        wb_.load8(wa, 0);
        solo_chain_1(a0, a1, 0.f, solo_rot8_h1(go[0]), wa);
    }
    if constexpr (U == 3) {
        wb_.load8(wa, 16);
        solo_chain_1_acc(a0, a1, solo_rot8_h1(go[U - 1]), wa);
    }
    float w2[8], w3a[8], w3b[8];
    wb_.load8(w2, 32);   // (requested ahead of the activation gradient, as in the forward pass)
    const float g_a2 = rows_act_grad(solo_join_rot(a0 + a1), kp.h2, sel);   // (h = 1 rows: rotated by 8, like kp.h2)
    if constexpr (TAGGED) rows_stage_t(tgd, SM::gpre(1, 0), g_a2, stager);
    else rows_stage(stg_net, SM::gpre(1, 0), row, pos, g_a2, stager);
    if constexpr (STAGE_FWD) rows_stage(stg_net, SM::act(1, 0), row, pos, kp.h2, stager);
    // g_h1 = W1^T g_a2
    if constexpr (U >= 2) { wb_.load8(w3a, 40); wb_.load8(w3b, 48); }
    solo_chain_1(a0, a1, 0.f, g_a2, w2);
    const float g_a1 = rows_act_grad(solo_join_rot(a0 + a1), kp.h1, sel);
    if constexpr (TAGGED) rows_stage_t(tgd, SM::gpre(0, 0), g_a1, stager);
    else rows_stage(stg_net, SM::gpre(0, 0), row, pos, g_a1, stager);
    if constexpr (STAGE_FWD) rows_stage(stg_net, SM::act(0, 0), row, pos, kp.h1, stager);
    if constexpr (STAGE_FWD) {
        float cm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cm[u] = row_ok ? cond[u] : 0.f;
        rows_stage_slots<U>(stg_net, SM::m(0), row, pos, cm, stager);
    }
    // d loss / d (conditioning input) += W0^T g_a1 of both nets
#pragma unroll
    for (int k = 0; k < U / 2; ++k) {
        float o0, o1;
        if (k > 0) { wb_.load8(w3a, 40 + 16 * k); wb_.load8(w3b, 48 + 16 * k); }
        solo_chain_2out(o0, o1, 0.f, 0.f, g_a1, w3a, w3b);
        solo_swap16(o0, o1);               // rows h = 0: both halves of slot 2 k; rows h = 1: both halves of slot 2 k + 1
        float fs, ft;
        solo_nets(o0 + o1, fs, ft);
        float t0 = fs + ft, t1 = t0;
        solo_swap16(t0, t1);
        gcond[2 * k] += t0;
        gcond[2 * k + 1] += t1;
    }
    if constexpr (U & 1) {
        float o0, o1;
        wb_.load8(wa, 40 + 8 * (U - 1));
        solo_chain_1(o0, o1, 0.f, g_a1, wa);
        float fs, ft;
        solo_nets(solo_join(o0 + o1), fs, ft);
        gcond[U - 1] += fs + ft;
    }
}

// data = X[perm] + jitter * randn (trainer.py:392) for the ROWS_PER_WG rows of workgroup wg in minibatch (epoch, mb), as flat rows
// [row][32 U] in LDS: thread idx handles four consecutive dims of one row = one Philox block of grid_rows_piece (same keys: the
// jitter draws are those of train_kernel / train_kernel_grid).  Run by the workgroup's waves 4.. while the row waves are in the pass.
template <int U>
__device__ __forceinline__ void rows_prepare(const TrainArgs &a, int epoch, int mb, int wg, float *buf) {
    const int idx = (int)threadIdx.x - 64 * ROWS_PER_WG;   // the waves behind the row waves (idle during the pass)
    if (idx < 0 || idx >= ROWS_PER_WG * 8 * U) return;
    const int r = idx / (8 * U), q = idx % (8 * U), D = a.s.D;
    const int M = min(a.batch, a.n_train - mb * a.batch);
    const int row = wg * ROWS_PER_WG + r;
    const bool row_ok = row < M;
    long src = 0;
    if (row_ok) src = a.perm[(size_t)epoch * a.n_train + mb * a.batch + row];
    const int d0 = 4 * q;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (row_ok && d0 + j < D) ? a.xtrain[(size_t)src * D + d0 + j] : 0.f;
    if (a.jitter != 0.f && row_ok) {
        const long p = (long)mb * a.batch + row;
        float n[4];
        if (a.noise) {
#pragma unroll
            for (int j = 0; j < 4; ++j) n[j] = d0 + j < D ? a.noise[((size_t)epoch * a.n_train + p) * D + d0 + j] : 0.f;
        } else {
            const f32x4 nn = noise_normal4(a.seed, (uint64_t)p, (uint32_t)(a.epoch_offset + epoch), (uint32_t)q, NOISE_STREAM_JITTER);
            n[0] = nn.x; n[1] = nn.y; n[2] = nn.z; n[3] = nn.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (d0 + j < D) v[j] += n[j] * a.jitter;
    }
    *reinterpret_cast<f32x4 *>(buf + r * 32 * U + d0) = (f32x4){v[0], v[1], v[2], v[3]};
}

template <int U>
__global__ void __launch_bounds__(TRAIN_THREADS) train_kernel_rows(TrainArgs a) {
    typedef StageMap<U, 1, 1> SM;
    constexpr int B = ROWS_B, IMG = ROWS_B * SOLO4_NF * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *imgf = smem;          // forward solo image  [B][SOLO4_NF / 4][64][4]
    float *imgb = smem + IMG;    // transposed solo image
    __shared__ __attribute__((aligned(16))) float xpre[2][ROWS_PER_WG * 32 * U];   // the minibatch's rows, flat, a phase ahead
    __shared__ int ctl[4];      // [0] stop flag, [1] counter, [2] best epoch
    __shared__ float ctlf[2];   // [0] best validation loss
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pos = lane & 15;
    const bool h1 = (lane & 16) != 0, translate_half = lane >= 32, stager = (lane & 16) == 0;
    const int sl_r = (int)a.gld_in & 255, sl_b = ((int)a.gld_in >> 8) & 255;   // hold-backs of the two polls, units of 256 cycles (launch_train_rows_t)
    const unsigned sel = translate_half ? 0xffffffffu : 0u;
    const int wg = blockIdx.x, G = gridDim.x;
    const int D = a.s.D;
    constexpr int NJOBS = 2 * U + 1;
    const int NJ = B * 2 * NJOBS;
    int phase = 0;

    // ---- the two solo images from the packed weights; the refresh maps of this thread ----
    for (int i = threadIdx.x; i < 2 * IMG; i += blockDim.x) smem[i] = 0.f;
    __syncthreads();
    for (int p = threadIdx.x; p < a.s.nets_params(); p += blockDim.x) {
        const int bn = p / a.s.net_params, o = p - bn * a.s.net_params;
        int df, db; bool sc;
        rows_param_dest<U>(D, bn >> 1, bn & 1, o, df, db, sc);
        const float v = a.w[p];
        if (df >= 0) imgf[df] = sc ? SOLO_TANH_PRESCALE * v : v;
        if (db >= 0) imgb[db] = v;
    }
    // The owners publish their new weights as tiles of the backward FRAGMENT image (+ the biases in its forward-image bias area):
    // element i of that image is parameter src(i), whose places in the two solo images rows_maps_kernel has packed into one word.
    // Thread tid re-lays elements 4 (tid + 512 u) .. + 3 every minibatch.
    constexpr int IMGF = ROWS_B * 2 * (2 * U * 256 + 256 + 16 + 16 + 16 * U);   // image_floats of the shape (flow_tile.h frag_net_floats)
    constexpr int RU = (IMGF / 4 + TRAIN_THREADS - 1) / TRAIN_THREADS;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 *gmap = reinterpret_cast<const i32x4 *>(a.gown);
    static_assert(2 * IMG < 0xffff && IMG < (1 << 14), "refresh map packing");

    // ---- who owns what (train_kernel_grid): wave (wg, wave) runs job Jmine of every minibatch and keeps that tile's parameters
    const int Jmine = wave * G + wg;
    const bool owner = Jmine < NJ;
    int off_f = 0, off_b = 0, off_bias = -1;
    if (owner) grid_job_image_offsets<U, 1, 1>(a.s, Jmine, &off_f, &off_b, &off_bias);
    OwnState os;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        os.wt[r] = os.bt[r] = -1;
        os.tw[r] = os.tm[r] = os.tv[r] = os.bw[r] = os.bm[r] = os.bv[r] = 0.f;
    }
    if (owner) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            grid_job_targets<U, 1, 1>(a.s, Jmine, lane, r, &os.wt[r], &os.bt[r]);
            if (os.wt[r] >= 0) { os.tw[r] = a.w[os.wt[r]]; os.tm[r] = a.m[os.wt[r]]; os.tv[r] = a.v[os.wt[r]]; }
            if (os.bt[r] >= 0) { os.bw[r] = a.w[os.bt[r]]; os.bm[r] = a.m[os.bt[r]]; os.bv[r] = a.v[os.bt[r]]; }
        }
    }
    // the parameters no job reaches, shared out over all waves of the grid in whole 256-byte rows of a compact private array
    const int n_waves = G * TRAIN_WAVES, ndead = *a.gndead;
    const int ndead_pad = (ndead + 63) & ~63;
    const int dead_per = ((ndead + n_waves - 1) / n_waves + 63) & ~63;
    const int dead0 = min(ndead, Jmine * dead_per), dead1 = min(ndead, dead0 + dead_per);
    float *dw = a.gdst, *dm = a.gdst + ndead_pad, *dv = a.gdst + 2 * ndead_pad;
    for (int k = dead0 + lane; k < dead1; k += 64) {
        const int pidx = a.gdead[k];
        dw[k] = a.w[pidx]; dm[k] = a.m[pidx]; dv[k] = a.v[pidx];
    }
    const bool resume = (a.flags & NNEST_TRAIN_RESUME) != 0;
    if (threadIdx.x == 0) {
        ctl[0] = 0;
        ctl[1] = resume ? a.result->counter : 0;
        ctl[2] = resume ? a.result->best_epoch : 0;
        ctlf[0] = resume ? a.result->best_validation_loss : INFINITY;
    }
    if (!resume) {   // best_model = deepcopy(netG)  (trainer.py:194): every wave snapshots what it owns (one writer per entry)
        if (owner) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (os.wt[r] >= 0) st_sc1(a.best_w + os.wt[r], os.tw[r]);
                if (os.bt[r] >= 0) st_sc1(a.best_w + os.bt[r], os.bw[r]);
            }
        }
        for (int k = dead0 + lane; k < dead1; k += 64) st_sc1(a.best_w + a.gdead[k], dw[k]);
    }
    const int n_mb = (a.n_train + a.batch - 1) / a.batch;
    if (a.max_epochs > 0) rows_prepare<U>(a, 0, 0, wg, xpre[0]);
    __syncthreads();

    float *pub = a.gimgf;         // the published weights: {weight, tag} pairs, 2 x image_floats floats (the two image areas of train_kernel_grid)
    float *part_base = a.gtile;   // [2][128] log p of a minibatch's rows, [2][128] validation sums of the row waves (sc1 words, one writer each)
    // the sum of the 4 G row waves' words in a fixed order: lane l takes words l and l + 64, then a butterfly over the lanes
    auto sum_rows = [&](const float *words) {
        float v = (lane < ROWS_PER_WG * G ? ld_sc1(words + lane) : 0.f) + (lane + 64 < ROWS_PER_WG * G ? ld_sc1(words + lane + 64) : 0.f);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
        return v;
    };
    int adam_t = a.adam_step ? *a.adam_step : 0;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0;
    (void)ph; (void)q0; (void)q1; (void)q2; (void)q3; (void)q4; (void)q5; (void)q6;
    int epochs_run = 0, mbcount = 0;
    float last_train_loss = 0.f;
    bool alive = true;
    unsigned long long dbg_arr = 0, dbg_st = 0;
    (void)dbg_arr; (void)dbg_st;
    const int row = wg * ROWS_PER_WG + wave;          // (waves 0..3)
    float *stg_net0 = a.gstage + (size_t)(translate_half ? 1 : 0) * SM::count * TRAIN_MAX_ROWS * 16;   // + 2 b regions per block

    // NormalizingFlow.forward (networks.py:24-32) of this wave's row; returns the lane's log-det partial
    auto forward = [&](float (&xs)[2][U], RowsKeep<U> (&kp)[B]) {
        float ld = rows_block_forward<U>(Solo4Lds{imgf, lane}, sel, h1, xs[1], xs[0], kp[0]);
        ld += rows_block_forward<U>(Solo4Lds{imgf + SOLO4_NF * 64, lane}, sel, h1, xs[0], xs[1], kp[1]);
        ld += rows_block_forward<U>(Solo4Lds{imgf + 2 * SOLO4_NF * 64, lane}, sel, h1, xs[1], xs[0], kp[2]);
        return ld;
    };
    // log p(row) = -sum E(z) + D base_const + log|det|   (networks.py:71-76), the same value in every lane
    auto log_prob = [&](const float (&xs)[2][U], float ld_lane) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int u = 0; u < U; ++u) ss += base_E(xs[c][u], a.s.base_beta);
        return -solo_row_sum(ss) + a.s.base_const * (float)D + solo_logdet_total(ld_lane);
    };

    for (int epoch = 0; epoch < a.max_epochs && alive; ++epoch) {
        float epoch_loss = 0.f;
        for (int mb = 0; mb < n_mb && alive; ++mb, ++mbcount) {
            const int M = min(a.batch, a.n_train - mb * a.batch);
            const int rows_pad = ((M + 15) >> 4) * 16;
            const bool row_ok = wave < ROWS_PER_WG && row < M;
            float *part = part_base + (mbcount & 1) * TRAIN_MAX_ROWS;
            AdamStep ad;
            {
                adam_t += 1;
                const double bc1 = 1.0 - pow(0.9, (double)adam_t), bc2 = 1.0 - pow(0.999, (double)adam_t);
                ad.step_size = (float)((double)a.lr / bc1);
                ad.inv_bc2s = (float)(1.0 / sqrt(bc2));
            }
            TSTAMP(q0);
#ifdef NNEST_STAMP
            const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
#endif
            {   // the NEXT minibatch's rows, beside this one's pass
                int e2 = epoch, m2 = mb + 1;
                if (m2 == n_mb) { m2 = 0; e2 = epoch + 1; }
                if (e2 < a.max_epochs) rows_prepare<U>(a, e2, m2, wg, xpre[(mbcount + 1) & 1]);
            }
            if (wave < ROWS_PER_WG) {
                float lp = 0.f;
                if (row < rows_pad) {   // (rows M .. rows_pad - 1 run on zeros with row_ok = false: their staged gradients must read 0)
                    float xs[2][U], gs[2][U];
                    const float *xr = xpre[mbcount & 1] + wave * 32 * U + 2 * U * pos;
#pragma unroll
                    for (int u = 0; u < U; ++u) { xs[0][u] = xr[2 * u]; xs[1][u] = xr[2 * u + 1]; }
                    RowsKeep<U> kp[B];
                    const float ld_lane = forward(xs, kp);
                    lp = row_ok ? log_prob(xs, ld_lane) : 0.f;
                    TSTAMP(q1);
                    // d(loss)/du = dE/du / M ; d(loss)/d(logdet) = -1/M
                    const float invM = 1.0f / (float)M, gld = -invM;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int u = 0; u < U; ++u) gs[c][u] = row_ok ? base_dE(xs[c][u], a.s.base_beta) * invM : 0.f;
                    constexpr size_t RS = (size_t)2 * SM::count * TRAIN_MAX_ROWS * 16;   // a block's two staging regions
                    rows_block_backward<U>(Solo4Lds{imgb + 2 * SOLO4_NF * 64, lane}, sel, h1, translate_half, pos, D, 0, row_ok, gld, xs[1], xs[0], gs[1], gs[0], kp[2], stg_net0 + 2 * RS, row, stager);
                    rows_block_backward<U>(Solo4Lds{imgb + SOLO4_NF * 64, lane}, sel, h1, translate_half, pos, D, 1, row_ok, gld, xs[0], xs[1], gs[0], gs[1], kp[1], stg_net0 + RS, row, stager);
                    rows_block_backward<U>(Solo4Lds{imgb, lane}, sel, h1, translate_half, pos, D, 0, row_ok, gld, xs[1], xs[0], gs[1], gs[0], kp[0], stg_net0, row, stager);
                }
                if (lane == 0) st_sc1(part + row, lp);   // (with the staging stores: drained by the barrier's own wait)
            }
            TSTAMP(q2);
#ifdef NNEST_STAMP
            if (mbcount == 20) { dbg_arr = __builtin_amdgcn_s_memrealtime(); dbg_st = rt_start; }   // 100 MHz constant clock: comparable across CUs
#endif
#ifdef NNEST_STAMP
            {   // diagnostic build: the first grid barrier taken apart (thread 0 of workgroup 0): ph[6] = drain + workgroup barrier, ph[7] = polls
                unsigned long long s0 = 0, s1 = 0;
                TSTAMP(s0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                TSTAMP(s1);
                phase += 1;
                __shared__ int okd;
                if (threadIdx.x == 0) {
                    __hip_atomic_fetch_add(a.gsync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned int want = (unsigned int)G * (unsigned int)phase;
                    int polls = 0;
                    __builtin_amdgcn_s_sleep(16);
                    while (__hip_atomic_load(a.gsync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) { ++polls; __builtin_amdgcn_s_sleep(1); }
                    okd = 1;
                    TACC(ph[6], s1, s0);
                    ph[7] += polls;
                }
                __syncthreads();
                alive = okd != 0;
            }
#else
            alive = grid_barrier(a.gsync, phase, G, a.gerr, 256 + sl_b);   // (the workgroups leave the pass ~1 k cycles apart)
#endif
            if (!alive) break;
            TSTAMP(q3);
            // ---- W + A: one weight-gradient job per wave of the grid; Adam on the tile's parameters in this wave's registers ----
            if (owner) {
                const int J = Jmine;
                const int bn = J / NJOBS;
                int q = J % NJOBS;
                const float *stg = a.gstage + (size_t)bn * SM::count * TRAIN_MAX_ROWS * 16;
                f32x4 bt = {0.f, 0.f, 0.f, 0.f}, t;
                constexpr int J_W3 = U, J_W2 = 1;
                if (q < J_W3) {
                    t = contract_rows_grid<true>(stg, rows_pad, SM::gout(q), SM::act(1, 0), lane, bt);
                } else if (q - J_W3 < J_W2) {
                    t = contract_rows_grid<true>(stg, rows_pad, SM::gpre(1, 0), SM::act(0, 0), lane, bt);
                } else {
                    q -= J_W3 + J_W2;
                    t = q == 0 ? contract_rows_grid<true>(stg, rows_pad, SM::gpre(0, 0), SM::m(q), lane, bt)
                               : contract_rows_grid<false>(stg, rows_pad, SM::gpre(0, 0), SM::m(q), lane, bt);
                }
                const float gt[4] = {t.x, t.y, t.z, t.w}, gb[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (os.wt[r] >= 0) adam_reg(a, ad, os.tw[r], gt[r], os.tm[r], os.tv[r]);
                    if (os.bt[r] >= 0) adam_reg(a, ad, os.bw[r], gb[r], os.bm[r], os.bv[r]);
                }
                // the NEW weights as they stand, every one with the minibatch's TAG beside it ({w, tag} granules of 8 bytes inside
                // 16-byte stores: element i of the backward fragment image at pub[2 i], its tag at pub[2 i + 1]).  A reader that finds
                // the tag of this minibatch has this minibatch's weight, whatever the order its loads were served in -- which is what
                // lets the refresh below poll the data itself instead of waiting at a second grid barrier and THEN loading.
                const float tg = __int_as_float(mbcount + 1);
                st_sc1_f32x4(pub + 2 * (off_b + lane * 4), (f32x4){os.tw[0], tg, os.tw[1], tg});
                st_sc1_f32x4(pub + 2 * (off_b + lane * 4) + 4, (f32x4){os.tw[2], tg, os.tw[3], tg});
                if (off_bias >= 0 && (lane & 15) == 0) {
                    st_sc1_f32x4(pub + 2 * (off_bias + (lane >> 4) * 4), (f32x4){os.bw[0], tg, os.bw[1], tg});
                    st_sc1_f32x4(pub + 2 * (off_bias + (lane >> 4) * 4) + 4, (f32x4){os.bw[2], tg, os.bw[3], tg});
                }
            }
            for (int k = dead0 + lane; k < dead1; k += 64) {   // the parameters no job reaches: zero gradient, weight decay only
                float w_ = dw[k], m_ = dm[k], v_ = dv[k];
                adam_reg(a, ad, w_, 0.f, m_, v_);
                dw[k] = w_; dm[k] = m_; dv[k] = v_;
            }
            TSTAMP(q4);
            // (no second grid barrier: what it ordered -- every job has read the staging area, every tile is published -- is exactly
            // "all tiles carry this minibatch's tag", which the refresh establishes by itself; a workgroup enters the next pass only
            // behind that, and the owners publish the next tiles only behind the next FIRST barrier, i.e. behind every refresh)
            __syncthreads();
            TSTAMP(q5);
            // loss = -mean(log_probs)  (trainer.py:394) over the rows' words (written before the first barrier); only the thread that
            // reports the epoch losses needs it
            float lsum = 0.f;
            if (wave == 0) lsum = sum_rows(part);
            // ---- R: the published tiles into this workgroup's two solo images.  All loads of a thread are requested at once; a
            // granule whose tag is not this minibatch's yet sends the thread round again (bounded).
            {
                constexpr int CH = RU > 4 ? 4 : RU;   // loads in flight per thread and round (x_dim > 64: two rounds, the second finds its data there; 8 in flight was measured slower, 1.22 against 1.02 ms per epoch at x_dim 100, and 2 the same as 4)
                const int want = mbcount + 1;
                for (int i = 0; i < sl_r; ++i) __builtin_amdgcn_s_sleep(4);   // (the owners' stores are ~1.5 k cycles from being visible: a poll issued at once would miss and cost a whole round trip)
#pragma unroll
                for (int c0 = 0; c0 < RU; c0 += CH) {
                    f32x4 va[CH], vc[CH];
                    i32x4 mp[CH];
#pragma unroll
                    for (int u = 0; u < CH; ++u) mp[u] = gmap[min((int)threadIdx.x + (c0 + u) * TRAIN_THREADS, IMGF / 4 - 1)];
                    bool fresh = false;
                    for (int polls = 0; !fresh; ++polls) {
                        if (polls > GRID_MAX_POLLS / 64) { *a.gerr = 1; break; }
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const int i = min((int)threadIdx.x + (c0 + u) * TRAIN_THREADS, IMGF / 4 - 1);
                            va[u] = ld_sc1_x4_issue(pub + 8 * (size_t)i);
                            vc[u] = ld_sc1_x4_issue(pub + 8 * (size_t)i + 4);
                        }
#pragma unroll
                        for (int u = 0; u < CH; ++u) asm volatile("s_waitcnt vmcnt(0)" : "+v"(va[u]), "+v"(vc[u]) : : "memory");
                        fresh = true;
#pragma unroll
                        for (int u = 0; u < CH; ++u)
                            fresh = fresh && __float_as_int(va[u].y) == want && __float_as_int(va[u].w) == want && __float_as_int(vc[u].y) == want &&
                                    __float_as_int(vc[u].w) == want;
                    }
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        if (c0 + u < RU && (int)threadIdx.x + (c0 + u) * TRAIN_THREADS < IMGF / 4) {
                            const float e4[4] = {va[u].x, va[u].z, vc[u].x, vc[u].z};
                            const int m4[4] = {mp[u].x, mp[u].y, mp[u].z, mp[u].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int f = m4[e] & 0xffff, k = (m4[e] >> 16) & 0xffff;
                                if (f != 0xffff) smem[f & 0x3fff] = (f >> 14) ? SOLO_TANH_PRESCALE * e4[e] : e4[e];
                                if (k != 0xffff) smem[k] = e4[e];
                            }
                        }
                    }
                }
            }
            epoch_loss += -lsum / (float)M;
            alive = __syncthreads_and(*a.gerr == 0 ? 1 : 0) != 0;   // (a refresh that ran out ends the launch: result.stopped = 2)
            if (!alive) break;
            TSTAMP(q6);
            if (wave == 0 && q1 > q0) { TACC(ph[0], q1, q0); TACC(ph[1], q2, q1); }
            TACC(ph[2], q3, q2); TACC(ph[3], q4, q3); TACC(ph[4], q5, q4); TACC(ph[5], q6, q5);
        }
        if (!alive) break;
        // ---- Trainer._validate (trainer.py:405-418): validation row r on wave r % 4 of workgroup (r / 4) % G ----
        {
            float vsum = 0.f;
            if (wave < ROWS_PER_WG) {
                for (int r = row; r < a.n_valid; r += ROWS_PER_WG * G) {
                    float xs[2][U];
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const int d = 2 * U * pos + 2 * u + c;
                            xs[c][u] = d < D ? a.xvalid[(size_t)r * D + d] : 0.f;
                        }
                    RowsKeep<U> kp[B];
                    const float ld_lane = forward(xs, kp);
                    vsum += log_prob(xs, ld_lane);
                }
                if (lane == 0) st_sc1(part_base + (2 + (epoch & 1)) * TRAIN_MAX_ROWS + row, vsum);
            }
            alive = grid_barrier(a.gsync, phase, G, a.gerr);
            if (!alive) break;
            const float vtot = sum_rows(part_base + (2 + (epoch & 1)) * TRAIN_MAX_ROWS);
            const float valid_loss = (-vtot / (float)a.n_valid) / (float)a.n_valid;  // mean, then / len(dataset)  :418
            const float train_loss = epoch_loss / (float)a.n_train;                   // trainer.py:403
            last_train_loss = train_loss;
            epochs_run = epoch + 1;
#ifndef NNEST_STAMP
            if (a.losses && wg == 0 && threadIdx.x == 0) {
                a.losses[2 * epoch] = train_loss;
                a.losses[2 * epoch + 1] = valid_loss;
            }
#endif
            // early stopping bookkeeping (trainer.py:205-209, :223-232); every thread of the grid evaluates the same values
            const bool improved = valid_loss < ctlf[0];
            __syncthreads();
            if (improved) {   // best_model = deepcopy(netG)  (trainer.py:208): every wave snapshots what it owns
                if (owner) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (os.wt[r] >= 0) st_sc1(a.best_w + os.wt[r], os.tw[r]);
                        if (os.bt[r] >= 0) st_sc1(a.best_w + os.bt[r], os.bw[r]);
                    }
                }
                for (int k = dead0 + lane; k < dead1; k += 64) st_sc1(a.best_w + a.gdead[k], dw[k]);
                if (threadIdx.x == 0) { ctlf[0] = valid_loss; ctl[2] = a.epoch_offset + epoch + 1; ctl[1] = 0; }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                ctl[1] += 1;
                if (ctl[1] > a.patience) ctl[0] = 1;
            }
            __syncthreads();
            if (ctl[0]) break;
        }
    }
#ifdef NNEST_STAMP
    if (threadIdx.x == 0 && a.losses && wg < 28) {   // (the diagnostic run has 40 epochs: 80 floats)
        a.losses[20 + wg] = (float)(dbg_arr & 0xffffff);
        a.losses[50 + wg] = (float)(dbg_st & 0xffffff);
    }
#endif
    // every wave writes what it owns back -- write-through stores, one writer per entry (launch_repack rebuilds the inference
    // kernels' forward image from a.w behind this launch)
    __syncthreads();
    const bool stopped = ctl[0] != 0;
    const bool restore = stopped || (a.flags & NNEST_TRAIN_FINALIZE);   // netG.load_state_dict(best_model)  (trainer.py:241)
    if (owner) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (os.wt[r] >= 0) {
                st_sc1(a.w + os.wt[r], restore ? ld_sc1(a.best_w + os.wt[r]) : os.tw[r]);
                st_sc1(a.m + os.wt[r], os.tm[r]); st_sc1(a.v + os.wt[r], os.tv[r]);
            }
            if (os.bt[r] >= 0) {
                st_sc1(a.w + os.bt[r], restore ? ld_sc1(a.best_w + os.bt[r]) : os.bw[r]);
                st_sc1(a.m + os.bt[r], os.bm[r]); st_sc1(a.v + os.bt[r], os.bv[r]);
            }
        }
    }
    for (int k = dead0 + lane; k < dead1; k += 64) {
        const int pidx = a.gdead[k];
        st_sc1(a.w + pidx, restore ? ld_sc1(a.best_w + pidx) : dw[k]);
        st_sc1(a.m + pidx, dm[k]); st_sc1(a.v + pidx, dv[k]);
    }
    if (wg != 0) return;
#ifdef NNEST_STAMP
    if (threadIdx.x == 0 && a.losses)   // diagnostic build: cycles per phase, summed over the minibatches (workgroup 0, wave 0)
        for (int i = 0; i < 8; ++i) a.losses[i] = (float)ph[i];
#endif
    if (threadIdx.x == 0) {
        if (a.adam_step) *a.adam_step = adam_t;
        a.result->epochs_run = a.epoch_offset + epochs_run;
        a.result->best_epoch = ctl[2];
        a.result->best_validation_loss = ctlf[0];
        a.result->last_train_loss = last_train_loss;
        a.result->counter = ctl[1];
        a.result->stopped = *a.gerr ? 2 : (stopped ? 1 : 0);   // 2: a grid barrier ran out (include/nnest_hip.h)
    }
}

// rows form: the reference's default coupling shape; NNEST_TRAIN_FORM=grid in the environment keeps train_kernel_grid (diagnostic)
static bool rows_eligible(const TrainArgs &a) {
    static const bool off = [] { const char *e = getenv("NNEST_TRAIN_FORM"); return e && !strcmp(e, "grid"); }();
    return !off && a.s.H == 16 && a.s.NH == 1 && a.s.L == 1 && a.s.B == ROWS_B && a.s.NT >= 1 && a.s.NT <= 4 && a.batch >= 1 &&
           a.batch <= TRAIN_MAX_ROWS;
}

template <int U>
static hipError_t launch_train_rows_t(TrainArgs a, float *gridws, hipStream_t st) {
    typedef GridSizes<U, 1, 1> GS;
    // (the workspace layout of launch_train_grid_t)
    a.gstage = gridws;
    a.gtile = a.gstage + GS::stage(a.s);
    a.gpos = reinterpret_cast<int *>(a.gtile + GS::tiles(a.s));
    a.gpart = reinterpret_cast<float *>(a.gpos + a.s.num_params());
    a.gsync = reinterpret_cast<unsigned int *>(a.gpart + 64);
    a.gerr = reinterpret_cast<int *>(a.gsync + 8);
    a.gndead = reinterpret_cast<int *>(a.gsync + 12);
    a.gdead = reinterpret_cast<int *>(a.gpart + 64 + 16);
    a.gown = reinterpret_cast<float *>(a.gdead + a.s.num_params());
    a.gown += (64 - ((size_t)(a.gown - gridws) & 63)) & 63;
    a.gdst = a.gown + (size_t)GRID_WG * TRAIN_WAVES * 64 * 32;
    a.gimgf = a.gdst + (size_t)3 * ((a.s.num_params() + 63) & ~63) + 64;
    a.gimgf += (64 - ((size_t)(a.gimgf - gridws) & 63)) & 63;
    a.gimgb = a.gimgf + ((a.s.image_floats + 63) & ~63);
    hipError_t e = hipMemsetAsync(a.gstage, 0, GS::stage(a.s) * sizeof(float), st);   // rows beyond the batch are never written: they must read 0
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.gpos, 0xFF, (size_t)a.s.num_params() * sizeof(int), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.gimgf, 0, (size_t)2 * ((a.s.image_floats + 63) & ~63) * sizeof(float), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.gpart, 0, (64 + 16) * sizeof(float), st);  // the barrier counter, the error word, the dead count
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((grid_gpos_kernel<U, 1, 1>), dim3(32), dim3(256), 0, st, a.gpos, a.s);
    hipLaunchKernelGGL(grid_dead_kernel, dim3(32), dim3(256), 0, st, a.gpos, a.s.num_params(), a.gdead, a.gndead);
    hipLaunchKernelGGL((rows_maps_kernel<U>), dim3(32), dim3(256), 0, st, reinterpret_cast<int *>(a.gown), a.s);   // (the owners' record area of train_kernel_grid: unused here)
    {   // how long the first poll of the grid barrier / of the refresh is held back (units of 256 cycles; NNEST_K5_SLEEP=r,b overrides)
        int r = 4, b = 4;
        if (const char *e = getenv("NNEST_K5_SLEEP")) sscanf(e, "%d,%d", &r, &b);
        a.gld_in = (float)((r & 255) + 256 * (b & 255));   // (the VJP's scalar: unused by the training loop)
    }
    const int NJ = ROWS_B * 2 * (2 * U + 1);
    const int G = max((a.batch + ROWS_PER_WG - 1) / ROWS_PER_WG, (NJ + TRAIN_WAVES - 1) / TRAIN_WAVES);
    const size_t lds = (size_t)2 * ROWS_B * SOLO4_NF * 64 * sizeof(float);
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(train_kernel_rows<U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((train_kernel_rows<U>), dim3(G), dim3(TRAIN_THREADS), lds, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_repack(a.w, a.img_fwd, a.s, st);
}

static hipError_t dispatch_train_rows(const TrainArgs &a, float *gridws, hipStream_t st) {
    switch (a.s.NT) {
        case 1: return launch_train_rows_t<1>(a, gridws, st);
        case 2: return launch_train_rows_t<2>(a, gridws, st);
        case 3: return launch_train_rows_t<3>(a, gridws, st);
        case 4: return launch_train_rows_t<4>(a, gridws, st);
    }
    return hipErrorInvalidConfiguration;
}
