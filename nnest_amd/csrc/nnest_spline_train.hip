// nnest_spline_train.hip -- Trainer.train for the neural-spline flow (reference nnest/trainer.py:134-245, :384-418 on
// SingleSpeedSpline, networks.py:393-715).
//
// The gradient of a batch (nnest_spline_loss_grad / _vjp and, for x_dim > 64, the training loop), all on the device:
//   1 spl_assemble_kernel   W = P (tril(L,-1)+I) (triu(U,1)+diag(S)) per block (networks.py:640-645)
//   2 spl_timage_kernel     MFMA fragment images of W, W^T and of both conditioners (forward and transposed)
//   3 spl_grad_kernel       four waves per 16 rows: forward (block inputs stashed), loss, hand-written backward one block
//                           at a time (spline_train_tile.h); per-tile partial gradients, no atomics; the rows' dLoss/d(raw
//                           spline parameters) and last hidden activations go to memory
//   3b spl_w3_kernel        the conditioners' last-layer weight gradients as one contraction over ALL rows
//   4 spl_reduce_kernel     fixed-order sum of the partials (+ the constant log-det terms of ActNorm / conv)
//   5 spl_lu_grad_kernel    dLoss/dW -> dLoss/d(L, S, U)
//   6 spl_adam_kernel       torch.optim.Adam with coupled weight decay (trainer.py:121-122)
// The training loop at x_dim <= 64 is two launches per minibatch: (3) and spl_update_kernel = 3b + 4 + 5 + 6 + the images kept
// current through position maps; the validation pass of an epoch (forward-only tiles of (3)) rides along with the first
// gradient launch of the next epoch, and the early-stopping bookkeeping (trainer.py:198-241) is a one-workgroup kernel after
// it, so the host queues epochs without draining the stream.  ActNorm's data-dependent initialisation (networks.py:698-705)
// runs as spl_init_kernel on the first batch pushed forward through a fresh flow.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "spline_host.h"
#include "spline_train_tile.h"

using namespace nnest;

namespace nnest {

static SplTrainShape make_train_shape(const SplineShape &s) {
    SplTrainShape t;
    memset(&t, 0, sizeof(t));
    t.s = s;
    const int T2 = 2 * s.NTh;
    t.conv_floats = T2 * T2 * 256;
    t.cf[0] = s.f1_floats; t.cf[1] = s.f2_floats;
    t.cb[0] = spl_cond_bwd_floats(s.NTh, s.NH, s.SU);
    t.cb[1] = spl_cond_bwd_floats(s.NTh, s.NH, s.SL);
    t.tblk_floats = 2 * t.conv_floats + t.cf[0] + t.cf[1] + t.cb[0] + t.cb[1] + 4;
    t.timage_floats = s.B * t.tblk_floats;
    const int D = s.D;
    t.p_s = 0; t.p_t = D; t.p_L = 2 * D; t.p_S = t.p_L + D * D; t.p_U = t.p_S + D;
    t.p_f[0] = t.p_U + D * D;
    t.p_f[1] = t.p_f[0] + spline_mlp_params(s.nl, SPL_P * s.nu, s.H);
    t.gw_floats = s.num_params + s.B * D * D + 4;
    t.SM = s.SU > s.SL ? s.SU : s.SL;
    return t;
}

// slot (half, tile, k-step r, lane group g) -> dimension or -1
__host__ __device__ inline int tslot_dim(const SplineShape &s, int hf, int t, int r, int g) {
    const int j = 16 * t + 4 * r + g, n = hf ? s.nu : s.nl;
    return j < n ? (hf ? s.nl : 0) + j : -1;
}
// row i of an MFMA tile (accumulator row 4g+r <-> i) of half hf, tile t -> dimension or -1
__host__ __device__ inline int trow_dim(const SplineShape &s, int hf, int t, int i) { return tslot_dim(s, hf, t, i & 3, i >> 2); }

// ---- 1: W per block ---------------------------------------------------------------------------------------------
// W[i][j] = sum_k (P Lm)[i][k] Um[k][j] with (P Lm)[i][k] = Lm[r][k], r = the column of the 1 in row i of P
// (the loads are unconditional and the loop is unrolled so that eight steps' loads are in flight together: one load round trip
// per step made the tiny assemble kernel 11 us)
__device__ __forceinline__ float spl_w_entry(const float *__restrict__ Lp, const float *__restrict__ Sp, const float *__restrict__ Up,
                                             int r, int j, int D) {
    const int kmax = r < j ? r : j;
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k <= kmax; ++k) {
        const float lv = Lp[r * D + k], uv = Up[k * D + j], sv = Sp[k];
        const float l = k < r ? lv : 1.f;
        const float u = k < j ? uv : sv;
        acc += l * u;
    }
    return acc;
}

// `stop` (every training kernel): the early-stopping flag of the running nnest_spline_train call, or NULL.  Once it is set,
// the launches that are already queued leave the state as it is.
__global__ void spl_assemble_kernel(const float *__restrict__ w, const int *__restrict__ pi, float *__restrict__ wmat, SplTrainShape ts,
                                    const int *__restrict__ stop) {
    if (stop && *stop) return;
    const int D = ts.s.D, n = ts.s.B * D * D;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const int b = idx / (D * D), i = (idx / D) % D, j = idx % D;
        const float *pb = w + (size_t)b * ts.s.blk_params;
        wmat[idx] = spl_w_entry(pb + ts.p_L, pb + ts.p_S, pb + ts.p_U, pi[b * D + i], j, D);
    }
}

// ---- 2: training image -------------------------------------------------------------------------------------------
// (index of the conditioner's packed parameter that sits at element o of its forward image, -1 = structural zero)
__device__ inline int cond_fwd_index(const SplineShape &s, int nin, int nout, int o) {
    const int H = s.H, NH = s.NH, NTh = s.NTh, S = (nout + 3) / 4, P = SPL_P;
    const int W0 = 0, b0 = W0 + H * nin, W1 = b0 + H, b1 = W1 + H * H, W2 = b1 + H, b2 = W2 + H * H, W3 = b2 + H,
              b3 = W3 + P * nout * H;
    const int oL2 = NH * NTh * 256, oL3 = oL2 + NH * NH * 256, ob = oL3 + NH * NH * 256, oL4 = ob + 3 * 16 * NH, ob4 = oL4 + S * SPL_QT * NH * 256;
    if (o < oL2) {
        const int lane = o & 63, q = o >> 6, r = q & 3, t = (q >> 2) % NTh, ht = (q >> 2) / NTh;
        const int g = lane >> 4, i = lane & 15, j = 16 * t + 4 * r + g;
        return j < nin ? W0 + (16 * ht + i) * nin + j : -1;
    } else if (o < ob) {
        const int oo = o < oL3 ? o - oL2 : o - oL3;
        const int W = o < oL3 ? W1 : W2;
        const int lane = oo & 63, q = oo >> 6, r = q & 3, hti = (q >> 2) % NH, hto = (q >> 2) / NH;
        const int g = lane >> 4, i = lane & 15;
        return W + (16 * hto + i) * H + 16 * hti + 4 * g + r;
    } else if (o < oL4) {
        const int oo = o - ob, l = oo / H, j = oo % H;
        return (l == 0 ? b0 : (l == 1 ? b1 : b2)) + j;
    } else if (o < ob4) {
        const int oo = o - oL4;
        const int lane = oo & 63, q4 = oo >> 6, r = q4 & 3, hti = (q4 >> 2) % NH, sq = (q4 >> 2) / NH, q = sq % SPL_QT, sidx = sq / SPL_QT;
        const int g = lane >> 4, i = lane & 15;
        const int jo = 4 * sidx + (i >> 2), pp = 4 * q + (i & 3);
        return (jo < nout && pp < P) ? W3 + (jo * P + pp) * H + 16 * hti + 4 * g + r : -1;
    } else {
        const int oo = o - ob4, r = oo & 3, g = (oo >> 2) & 3, sq = oo >> 4, q = sq % SPL_QT, sidx = sq / SPL_QT;
        const int jo = 4 * sidx + g, pp = 4 * q + r;
        return (jo < nout && pp < P) ? b3 + jo * P + pp : -1;
    }
}
__device__ inline float cond_fwd_value(const SplineShape &s, const float *p, int nin, int nout, int o) {
    const int i = cond_fwd_index(s, nin, nout, o);
    return i >= 0 ? p[i] : 0.f;
}

__device__ inline int cond_bwd_index(const SplineShape &s, int nin, int nout, int o) {
    const int H = s.H, NH = s.NH, NTh = s.NTh, P = SPL_P;
    const int W0 = 0, W1 = W0 + H * nin + H, W2 = W1 + H * H + H, W3 = W2 + H * H + H;
    const int oB2 = NTh * NH * 256, oB3 = oB2 + NH * NH * 256, oB4 = oB3 + NH * NH * 256;
    if (o < oB2) {  // B1 [t][ht][r][64]: W0[16ht+4g+r][dim of (tile t, row i)]
        const int lane = o & 63, q = o >> 6, r = q & 3, ht = (q >> 2) % NH, t = (q >> 2) / NH;
        const int g = lane >> 4, i = lane & 15, j = 16 * t + 4 * (i & 3) + (i >> 2);
        return j < nin ? W0 + (16 * ht + 4 * g + r) * nin + j : -1;
    } else if (o < oB4) {  // B2 / B3 [hti][hto][r][64]: W[16hto+4g+r][16hti+i]
        const int oo = o < oB3 ? o - oB2 : o - oB3;
        const int W = o < oB3 ? W1 : W2;
        const int lane = oo & 63, q = oo >> 6, r = q & 3, hto = (q >> 2) % NH, hti = (q >> 2) / NH;
        const int g = lane >> 4, i = lane & 15;
        return W + (16 * hto + 4 * g + r) * H + 16 * hti + i;
    } else {  // B4 [s][q][hto][r][64]: W3[(4s+g)*23 + 4q+r][16hto+i]
        const int oo = o - oB4;
        const int lane = oo & 63, q4 = oo >> 6, r = q4 & 3, hto = (q4 >> 2) % NH, sq = (q4 >> 2) / NH, q = sq % SPL_QT, sidx = sq / SPL_QT;
        const int g = lane >> 4, i = lane & 15;
        const int jo = 4 * sidx + g, pp = 4 * q + r;
        return (jo < nout && pp < P) ? W3 + (jo * P + pp) * H + 16 * hto + i : -1;
    }
}
__device__ inline float cond_bwd_value(const SplineShape &s, const float *p, int nin, int nout, int o) {
    const int i = cond_bwd_index(s, nin, nout, o);
    return i >= 0 ? p[i] : 0.f;
}

// Position maps packed conditioner parameter -> its element of the forward / transposed image (every such parameter occurs
// at most once in each; -1 = not in that image, e.g. the biases in the transposed one), built once per flow: with them the
// update kernel of the training loop keeps the image current (spl_update_kernel) instead of a rebuild per minibatch.
__global__ void spl_build_pos_kernel(int *__restrict__ pos_f, int *__restrict__ pos_b, int *__restrict__ conv_src, SplTrainShape ts) {
    const SplineShape &s = ts.s;
    const long total = (long)ts.timage_floats;
    // element of W (row-major index) behind element o of the two conv fragment images of a block, -1 = padding (as spl_timage_kernel)
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < 2 * ts.conv_floats; o += gridDim.x * blockDim.x) {
        const int NTh = s.NTh, T2 = 2 * NTh, D = s.D;
        const bool bwd = o >= ts.conv_floats;
        const int oo = bwd ? o - ts.conv_floats : o;
        const int lane = oo & 63, q = oo >> 6, r = q & 3, ti = (q >> 2) % T2, to = (q >> 2) / T2;
        const int g = lane >> 4, i = lane & 15;
        const int dk = tslot_dim(s, ti / NTh, ti % NTh, r, g), dm = trow_dim(s, to / NTh, to % NTh, i);
        conv_src[o] = (dk >= 0 && dm >= 0) ? (bwd ? dm * D + dk : dk * D + dm) : -1;
    }
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / ts.tblk_floats);
        int o = (int)(idx - (long)b * ts.tblk_floats) - 2 * ts.conv_floats;
        if (o < 0) continue;
        const int base = b * s.blk_params;
        int src;
        if (o < ts.cf[0]) { if ((src = cond_fwd_index(s, s.nl, s.nu, o)) >= 0) pos_f[base + ts.p_f[0] + src] = (int)idx; }
        else if ((o -= ts.cf[0]) < ts.cf[1]) { if ((src = cond_fwd_index(s, s.nu, s.nl, o)) >= 0) pos_f[base + ts.p_f[1] + src] = (int)idx; }
        else if ((o -= ts.cf[1]) < ts.cb[0]) { if ((src = cond_bwd_index(s, s.nl, s.nu, o)) >= 0) pos_b[base + ts.p_f[0] + src] = (int)idx; }
        else if ((o -= ts.cb[0]) < ts.cb[1]) { if ((src = cond_bwd_index(s, s.nu, s.nl, o)) >= 0) pos_b[base + ts.p_f[1] + src] = (int)idx; }
    }
}

__global__ void spl_timage_kernel(const float *__restrict__ w, const float *__restrict__ wmat, float *__restrict__ timg, SplTrainShape ts,
                                  const int *__restrict__ stop) {
    if (stop && *stop) return;
    const SplineShape &s = ts.s;
    const int D = s.D, NTh = s.NTh, T2 = 2 * NTh;
    const long total = (long)ts.timage_floats;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / ts.tblk_floats);
        int o = (int)(idx - (long)b * ts.tblk_floats);
        const float *pb = w + (size_t)b * s.blk_params;
        const float *Wm = wmat + (size_t)b * D * D;
        float v = 0.f;
        if (o < 2 * ts.conv_floats) {
            const bool bwd = o >= ts.conv_floats;
            if (bwd) o -= ts.conv_floats;
            const int lane = o & 63, q = o >> 6, r = q & 3, ti = (q >> 2) % T2, to = (q >> 2) / T2;
            const int g = lane >> 4, i = lane & 15;
            const int dk = tslot_dim(s, ti / NTh, ti % NTh, r, g);   // contracted index
            const int dm = trow_dim(s, to / NTh, to % NTh, i);       // output row
            if (dk >= 0 && dm >= 0) v = bwd ? Wm[(size_t)dm * D + dk] : Wm[(size_t)dk * D + dm];  // c = a W ; g_a = g_c W^T
        } else {
            o -= 2 * ts.conv_floats;
            if (o < ts.cf[0]) v = cond_fwd_value(s, pb + ts.p_f[0], s.nl, s.nu, o);
            else if ((o -= ts.cf[0]) < ts.cf[1]) v = cond_fwd_value(s, pb + ts.p_f[1], s.nu, s.nl, o);
            else if ((o -= ts.cf[1]) < ts.cb[0]) v = cond_bwd_value(s, pb + ts.p_f[0], s.nl, s.nu, o);
            else if ((o -= ts.cb[0]) < ts.cb[1]) v = cond_bwd_value(s, pb + ts.p_f[1], s.nu, s.nl, o);
            else if ((o -= ts.cb[1]) == 0) {  // log|det| of ActNorm + conv (networks.py:650, :676)
                float acc = 0.f;
                for (int d = 0; d < D; ++d) acc += pb[ts.p_s + d] + logf(fabsf(pb[ts.p_S + d]));
                v = acc;
            }
        }
        timg[idx] = v;
    }
}

// ---- 3: forward / backward ----------------------------------------------------------------------------------------
// VJP: the flow as one stage of a composite model: upstream gradient gz and dL/d(logdet) in, dL/dw and gx = dL/dx out
enum { SPL_MODE_GRAD = 0, SPL_MODE_LOSS = 1, SPL_MODE_VJP = 2 };

struct SplGradArgs {
    const float *timg;
    const float *w;
    SplTrainShape ts;
    const float *x;     // [n_rows, D] row-major
    const int *perm;    // row index of batch row i (NULL: identity)
    int M;              // rows in this batch
    int mtot;           // the mean's denominator
    const float *noise; // recorded jitter noise rows (batch order) or NULL
    uint64_t seed;
    long noise_row0;    // index of batch row 0 in the epoch (in-kernel stream position)
    int epoch;
    float jitter;
    float *partial;     // [tiles][gw_floats]
    float *stash;       // [tiles][B][2 NTh][64] f32x4
    int mode;
    const float *gz;    // VJP: upstream gradient [M, D]
    float *gx;          // VJP: gradient wrt the input rows [M, D]
    float gld_in;       // VJP: dL/d(logdet)
    const int *stop;    // early-stopping flag (see spl_assemble_kernel) or NULL
    float *gbuf;        // [B][2][SM][SPL_QT][tiles][64] f32x4: dLoss/d(raw spline parameters) of every row, per coupling
    float *keep;        // [tiles][TEAM][B][2][spl_keep_floats4][64] f32x4: activations and spline parameters kept for the backward pass
    float *hbuf;        // [B][2][tiles][NH][64] f32x4: the conditioners' last hidden activations (spl_w3_item contracts the two)
    int val_tiles;      // training loop: workgroups past the batch's own tiles run the forward-only pass over the validation rows
    const float *xv;    //   (the previous epoch's validation loss, trainer.py:405-418, in the shadow of this minibatch's gradient pass)
    int Mv;
    int rows_per_tile;  // 16, 8 or 4: a minibatch is only 100 rows, so the tiles are made shallower to spread them over
                        // more waves / CUs (the matrix-core columns of the unused walkers idle; the launch is latency-bound)
    int lds_heads;      // round 6: the blocks' conv fragments (each wave: the output tile it is dealt) and ActNorm vectors are staged in
                        // LDS by the prologue, beside the rows' loads -- every block's conv and ActNorm stage used to open with a round
                        // trip to L2 of its own (stamps: 3.6 us per block in the forward pass, of which the arithmetic is a tenth)
};

template <int NTh>
__device__ __forceinline__ void spl_actnorm_vecs(const SplTrainShape &ts, const float *pb, int lane, f32x4 (&es)[2][NTh], f32x4 (&tv)[2][NTh]) {
    const int g = lane >> 4;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int t = 0; t < NTh; ++t) {
            float e[4], tt[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = tslot_dim(ts.s, hf, t, r, g), dc = d >= 0 ? d : 0;  // unconditional loads: they issue together
                const float sv = pb[ts.p_s + dc], tl = pb[ts.p_t + dc];
                e[r] = d >= 0 ? expf(sv) : 0.f;
                tt[r] = d >= 0 ? tl : 0.f;
            }
            es[hf][t] = (f32x4){e[0], e[1], e[2], e[3]};
            tv[hf][t] = (f32x4){tt[0], tt[1], tt[2], tt[3]};
        }
}

// c = a W (no bias): the conv fragments have the layout of spl_affine's weights
template <int NTh>
__device__ __forceinline__ void spl_matmul(const float *__restrict__ frag, int lane, const f32x4 (&in)[2][NTh], f32x4 (&out)[2][NTh]) {
    constexpr int T2 = 2 * NTh;
    float wf[T2 * T2 * 4];
    load_frags<T2 * T2 * 4>(frag, lane, wf);
#pragma unroll
    for (int to = 0; to < T2; ++to) {
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < T2; ++ti) {
            const f32x4 v = in[ti / NTh][ti % NTh];
            const float *a = wf + (to * T2 + ti) * 4;
            acc0 = mfma4(a[0], v.x, acc0);
            acc1 = mfma4(a[1], v.y, acc1);
            acc0 = mfma4(a[2], v.z, acc0);
            acc1 = mfma4(a[3], v.w, acc1);
        }
        out[to / NTh][to % NTh] = acc0 + acc1;
    }
}

// the same product with its output tiles dealt out over the team's waves (to mod TEAM) and exchanged through LDS: every wave
// repeated all (2 NTh)^2 x 4 matrix instructions before, 0.85 us per product at x_dim 50
template <int NTh, int TEAM, bool TAILB = true>
__device__ __forceinline__ void spl_matmul_team(const float *__restrict__ frag, int lane, int wv, f32x4 *xch, const f32x4 (&in)[2][NTh],
                                                f32x4 (&out)[2][NTh]) {
    constexpr int T2 = 2 * NTh;
#pragma unroll
    for (int to = 0; to < T2; ++to) {
        if ((to & (TEAM - 1)) != wv) continue;  // uniform over the wave
        float wf[T2 * 4];
        load_frags<T2 * 4>(frag + (size_t)to * T2 * 256, lane, wf);
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
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

// the same, the wave's output tile `wv` (TEAM >= 2 NTh: one tile per wave at most) with its fragments read from the wave's LDS slot
template <int NTh, int TEAM, bool TAILB = true>
__device__ __forceinline__ void spl_matmul_team_lds(const float *slot /* LDS: [2 NTh * 4][64] */, int lane, int wv, f32x4 *xch, const f32x4 (&in)[2][NTh],
                                                    f32x4 (&out)[2][NTh]) {
    constexpr int T2 = 2 * NTh;
    static_assert(T2 <= TEAM, "one output tile per wave");
    if (wv < T2) {   // uniform over the wave
        float wf[T2 * 4];
#pragma unroll
        for (int i = 0; i < T2 * 4; ++i) wf[i] = slot[i * 64 + lane];
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < T2; ++ti) {
            const f32x4 v = in[ti / NTh][ti % NTh];
            acc0 = mfma4(wf[ti * 4 + 0], v.x, acc0);
            acc1 = mfma4(wf[ti * 4 + 1], v.y, acc1);
            acc0 = mfma4(wf[ti * 4 + 2], v.z, acc0);
            acc1 = mfma4(wf[ti * 4 + 3], v.w, acc1);
        }
        xch[wv * 64 + lane] = acc0 + acc1;
    }
    spl_team_barrier();
#pragma unroll
    for (int to = 0; to < T2; ++to) out[to / NTh][to % NTh] = xch[to * 64 + lane];
    if constexpr (TAILB) spl_team_barrier();
}

// sum over the 16 rows (lanes w) of a tile; valid in every lane
// (the 16 lanes of a lane group are one DPP row: quad swaps, then the half-row and row mirrors -- four VALU adds per value
// instead of four ds_bpermute round trips)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}
__device__ __forceinline__ f32x4 rows_sum(f32x4 v) {
    return (f32x4){row16_sum(v.x), row16_sum(v.y), row16_sum(v.z), row16_sum(v.w)};
}

// Backward of one coupling: `tr` (n_out dims, S super-tiles) was transformed conditioned on `cond` by the conditioner
// at packed offset pnet / images cf (forward) and cbw (transposed).  On entry g_tr = dLoss/d(outputs of the transformed
// half), x_tr = the half BEFORE the transform; on exit g_tr = dLoss/d(x_tr) and g_cond has the conditioner path added.
// Team of four waves per tile (TEAM = 4): wave wv owns the super-tiles s = wv (mod 4) -- their spline reverse mode, last-layer
// weight gradients and share of g_h3 -- and the hidden layers' weight gradients are dealt out one layer per wave; the
// delta propagation through the trunk is repeated by every wave so that all four leave with the same g_tr / g_cond.
template <int NTh, int NH, int TEAM, bool DUP, bool TAILB = true>
__device__ __forceinline__ void spl_coupling_bwd(const SplTrainShape &ts, const float *__restrict__ cf, const float *__restrict__ cbw, int pnet,
                                                 int nin, int nout, int S, int lane, bool row_ok, bool cmask, float gld, float *lds17, float *gp,
                                                 const f32x4 (&cond)[NTh], const f32x4 (&x_tr)[NTh], f32x4 (&g_tr)[NTh], f32x4 (&g_cond)[NTh],
                                                 int wv, f32x4 *xch, f32x4 *__restrict__ gq, f32x4 *__restrict__ hq, int item_stride, const f32x4 *__restrict__ keep,
                                                 int w0_wave   // the wave that has the FIRST layer's weight-gradient job (round 6: it was the last wave's too, which then came
                                                               // ~0.6 us per coupling late to the block's closing barrier: the stamps' "matmulT" of the other three)
#ifdef NNEST_STAMP
                                                 , long long *cst
#endif
                                                 ) {
#ifdef NNEST_STAMP
    long long c_a = wall_clock64();
#define CB_STAMP(i) { const long long c_n = wall_clock64(); cst[i] += c_n - c_a; c_a = c_n; }
#else
#define CB_STAMP(i)
#endif
    const int g = lane >> 4, w = lane & 15, H = ts.s.H;
    const float tail = ts.s.tail;
    f32x4 h[3][NH];
    if constexpr (DUP) {  // kept by the forward pass (spl_coupling_pair)
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int ht = 0; ht < NH; ++ht) h[l][ht] = keep[(l * NH + ht) * 64 + lane];
    } else {
        spl_hidden_keep<NTh, NH>(cf, lane, cond, h);
    }
    const float *L4 = cf + spl_cond_hidden_floats(NTh, NH);
    const float *b4 = L4 + (size_t)S * SPL_QT * NH * 256;
    const float *B1 = cbw, *B2 = cbw + NTh * NH * 256, *B3 = B2 + NH * NH * 256, *B4 = B3 + NH * NH * 256;
    // packed offsets of this conditioner's parameters
    const int pW0 = pnet, pb0 = pW0 + H * nin, pW1 = pb0 + H, pb1 = pW1 + H * H, pW2 = pb1 + H, pb2 = pW2 + H * H;  // (W3, b3: spl_w3_item)
    float hT[2][NH][4];  // activations transposed for the weight-gradient contractions: the wave that has those jobs only
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) hT[l][ht][0] = hT[l][ht][1] = hT[l][ht][2] = hT[l][ht][3] = 0.f;
    if (TEAM == 1 || wv == TEAM - 1) {
#pragma unroll
        for (int l = 0; l < 2; ++l) tile_transpose_batch<NH>(lds17, lane, h[l], hT[l]);
    }
    // The last layer's weight gradients dW3 = G^T h3 are NOT contracted here, tile by tile on the critical path (they were half of
    // this function's time): the rows' G (below) and h3 go to memory and spl_w3_item contracts them over ALL rows of the batch.
    if (TEAM == 1 || wv == TEAM - 1) {
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) hq[ht * 64 + lane] = row_ok ? h[2][ht] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 g_h[NH];
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) g_h[ht] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the transposed trunk fragments (B1 | B2 | B3 are contiguous): requested now, used after the spline stage
    float wb1[NTh * NH * 4], wb23[2 * NH * NH * 4];
    load_frags<NTh * NH * 4>(B1, lane, wb1);
    load_frags<2 * NH * NH * 4>(B2, lane, wb23);
    CB_STAMP(0)
    if constexpr (DUP) {
        // the wave's super-tiles s = wv + 4k two at a time, (k, k + 1) in the low / high half of the columns (spl_coupling_pair)
        const bool lo = w < 8;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r != wv) continue;  // uniform over the wave
#pragma unroll
            for (int k = 0; k < NTh; k += 2) {
                const int sA = 4 * k + r, sB = 4 * (k + 1) + r;
                if (sA >= S) continue;
                const bool hasB = (k + 1 < NTh) && sB < S;
                const int kb = (k + 1 < NTh) ? k + 1 : k;  // (in range for odd NTh)
                f32x4 raw[SPL_QT], graw[SPL_QT];
#pragma unroll
                for (int q = 0; q < SPL_QT; ++q) raw[q] = keep[(3 * NH + (k >> 1) * SPL_QT + q) * 64 + lane];  // (each half its own parameters)
                float x = reg_of(x_tr[k], r), gyr = reg_of(g_tr[k], r);
                if (k + 1 < NTh) {
                    x = lo ? x : reg_of(x_tr[kb], r);
                    gyr = lo ? gyr : reg_of(g_tr[kb], r);
                }
                // the transposed last-layer fragments of both super-tiles: requested before the spline arithmetic, used after it
                float wA[SPL_QT * NH * 4], wB[SPL_QT * NH * 4];
                load_frags<SPL_QT * NH * 4>(B4 + (size_t)sA * SPL_QT * NH * 256, lane, wA);
                load_frags<SPL_QT * NH * 4>(B4 + (size_t)(hasB ? sB : sA) * SPL_QT * NH * 256, lane, wB);
                const bool valid = row_ok && (lo ? (4 * sA + g < nout) : (hasB && 4 * sB + g < nout));
                const float gy = valid ? gyr : 0.f;
                float y, lad;
#ifdef NNEST_STAMP
                long long c_b = wall_clock64();
                { asm volatile("" :: "v"(raw[0].x + raw[5].z)); const long long c_n = wall_clock64(); cst[4] += c_n - c_b; c_b = c_n; }
#endif
                const float gx = spl_rqs_fwd_bwd(raw, tail, x, gy, valid ? gld : 0.f, y, lad, graw);
#ifdef NNEST_STAMP
                { asm volatile("" :: "v"(gx + graw[0].x + graw[5].z)); const long long c_n = wall_clock64(); cst[5] += c_n - c_b; c_b = c_n; }
#endif
                const float gxo = valid ? gx : 0.f, gxp = half_swap(gxo);
                set_reg(g_tr[k], r, lo ? gxo : gxp);
                if (k + 1 < NTh && hasB) set_reg(g_tr[kb], r, lo ? gxp : gxo);
                const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < SPL_QT; ++q) {
                    if (!valid) graw[q] = zero4;
                    // last layer: G to memory (dW3, db3: spl_w3_item; each row once: A from the low lanes, B from the high ones),
                    // g_h3 += W3^T G
                    const f32x4 gA = lo ? graw[q] : zero4, gB = lo ? zero4 : graw[q];
                    gq[(size_t)(sA * SPL_QT + q) * item_stride + lane] = gA;
#pragma unroll
                    for (int hto = 0; hto < NH; ++hto) {
                        const float *a = wA + (q * NH + hto) * 4;
                        g_h[hto] = mfma4(a[0], gA.x, g_h[hto]);
                        g_h[hto] = mfma4(a[1], gA.y, g_h[hto]);
                        g_h[hto] = mfma4(a[2], gA.z, g_h[hto]);
                        g_h[hto] = mfma4(a[3], gA.w, g_h[hto]);
                    }
                    if (hasB) {
                        gq[(size_t)(sB * SPL_QT + q) * item_stride + lane] = gB;
#pragma unroll
                        for (int hto = 0; hto < NH; ++hto) {
                            const float *a = wB + (q * NH + hto) * 4;
                            g_h[hto] = mfma4(a[0], gB.x, g_h[hto]);
                            g_h[hto] = mfma4(a[1], gB.y, g_h[hto]);
                            g_h[hto] = mfma4(a[2], gB.z, g_h[hto]);
                            g_h[hto] = mfma4(a[3], gB.w, g_h[hto]);
                        }
                    }
                }
#ifdef NNEST_STAMP
                { asm volatile("" :: "v"(g_h[0].x)); const long long c_n = wall_clock64(); cst[6] += c_n - c_b; }
#endif
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < 4 * NTh; ++s) {
            if (s < S && (TEAM == 1 || (s & (TEAM - 1)) == wv)) {
                f32x4 raw[SPL_QT], graw[SPL_QT];
                spl_raw<NH>(L4, b4, s, lane, h[2], raw);
                const bool valid = row_ok && (4 * s + g < nout);
                const float x = reg_of(x_tr[s >> 2], s & 3);
                const float gy = valid ? reg_of(g_tr[s >> 2], s & 3) : 0.f;
                float y, lad;
                const float gx = spl_rqs_fwd_bwd(raw, tail, x, gy, valid ? gld : 0.f, y, lad, graw);
                set_reg(g_tr[s >> 2], s & 3, valid ? gx : 0.f);
#pragma unroll
                for (int q = 0; q < SPL_QT; ++q) {
                    if (!valid) graw[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    // last layer: G to memory (dW3, db3: spl_w3_item), g_h3 += W3^T G
                    gq[(size_t)(s * SPL_QT + q) * item_stride + lane] = cmask ? graw[q] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int hto = 0; hto < NH; ++hto) {
                        const float *a = B4 + (size_t)(((s * SPL_QT + q) * NH + hto) * 4) * 64 + lane;
                        g_h[hto] = mfma4(a[0], graw[q].x, g_h[hto]);
                        g_h[hto] = mfma4(a[64], graw[q].y, g_h[hto]);
                        g_h[hto] = mfma4(a[128], graw[q].z, g_h[hto]);
                        g_h[hto] = mfma4(a[192], graw[q].w, g_h[hto]);
                    }
                }
            }
        }
    }
    CB_STAMP(1)
    if (TEAM > 1) {  // merge the transformed-half gradients (register r of tile t from wave (4t + r) mod TEAM) and sum g_h3
#pragma unroll
        for (int t = 0; t < NTh; ++t) xch[(wv * NTh + t) * 64 + lane] = g_tr[t];
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) xch[(TEAM * NTh + wv * NH + ht) * 64 + lane] = g_h[ht];
        spl_team_barrier();
#pragma unroll
        for (int t = 0; t < NTh; ++t)
            g_tr[t] = (f32x4){xch[(((4 * t + 0) & (TEAM - 1)) * NTh + t) * 64 + lane].x, xch[(((4 * t + 1) & (TEAM - 1)) * NTh + t) * 64 + lane].y,
                              xch[(((4 * t + 2) & (TEAM - 1)) * NTh + t) * 64 + lane].z, xch[(((4 * t + 3) & (TEAM - 1)) * NTh + t) * 64 + lane].w};
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
            f32x4 acc = xch[(TEAM * NTh + 0 * NH + ht) * 64 + lane];
#pragma unroll
            for (int k = 1; k < TEAM; ++k) acc = acc + xch[(TEAM * NTh + k * NH + ht) * 64 + lane];
            g_h[ht] = DUP ? acc + half_swap4(acc) : acc;  // (DUP: column w has the A share, column w ^ 8 the B share of the row)
        }
        if constexpr (TAILB) spl_team_barrier();
    }
    CB_STAMP(2)
    // hidden layers 3 and 2 (W2 over h[1], W1 over h[0])
#pragma unroll
    for (int l = 2; l >= 1; --l) {
        const bool mine = TEAM == 1 || wv == TEAM - 1;  // the trunk's weight gradients: the last wave's (it has the short share of the super-tiles)
        const int pW = l == 2 ? pW2 : pW1, pb = l == 2 ? pb2 : pb1;
        f32x4 g_pre[NH], g_prev[NH];
        float gT[NH][4];
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
            g_pre[ht] = lrelu_grad4(g_h[ht], h[l][ht]);
            if (mine) {
                const f32x4 gc = cmask ? g_pre[ht] : (f32x4){0.f, 0.f, 0.f, 0.f};  // (each row once in the sums over rows)
                tile_transpose(lds17, lane, gc, gT[ht]);
                const f32x4 db = rows_sum(gc);
                if (w == 0) {
                    gp[pb + 16 * ht + 4 * g + 0] = db.x; gp[pb + 16 * ht + 4 * g + 1] = db.y;
                    gp[pb + 16 * ht + 4 * g + 2] = db.z; gp[pb + 16 * ht + 4 * g + 3] = db.w;
                }
            }
        }
        if (mine) {
#pragma unroll
            for (int hto = 0; hto < NH; ++hto)
#pragma unroll
                for (int hti = 0; hti < NH; ++hti) {
                    const f32x4 dW = contract16(gT[hto], hT[l - 1][hti]);  // [out 16hto+4g+r][in 16hti+w]
                    gp[pW + (16 * hto + 4 * g + 0) * H + 16 * hti + w] = dW.x; gp[pW + (16 * hto + 4 * g + 1) * H + 16 * hti + w] = dW.y;
                    gp[pW + (16 * hto + 4 * g + 2) * H + 16 * hti + w] = dW.z; gp[pW + (16 * hto + 4 * g + 3) * H + 16 * hti + w] = dW.w;
                }
        }
#pragma unroll
        for (int hti = 0; hti < NH; ++hti) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hto = 0; hto < NH; ++hto) {
                const float *a = wb23 + ((l - 1) * NH * NH + hti * NH + hto) * 4;
                acc = mfma4(a[0], g_pre[hto].x, acc);
                acc = mfma4(a[1], g_pre[hto].y, acc);
                acc = mfma4(a[2], g_pre[hto].z, acc);
                acc = mfma4(a[3], g_pre[hto].w, acc);
            }
            g_prev[hti] = acc;
        }
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) g_h[ht] = g_prev[ht];
    }
    // first layer: W0 over the conditioning half
    {
        const bool mine = TEAM == 1 || wv == w0_wave;
        f32x4 g_pre[NH];
        float gT[NH][4];
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
            g_pre[ht] = lrelu_grad4(g_h[ht], h[0][ht]);
            if (mine) {
                const f32x4 gc = cmask ? g_pre[ht] : (f32x4){0.f, 0.f, 0.f, 0.f};
                tile_transpose(lds17, lane, gc, gT[ht]);
                const f32x4 db = rows_sum(gc);
                if (w == 0) {
                    gp[pb0 + 16 * ht + 4 * g + 0] = db.x; gp[pb0 + 16 * ht + 4 * g + 1] = db.y;
                    gp[pb0 + 16 * ht + 4 * g + 2] = db.z; gp[pb0 + 16 * ht + 4 * g + 3] = db.w;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NTh; ++t) {
            if (mine) {
                f32x4 cin = cond[t];
                if (!row_ok) cin = (f32x4){0.f, 0.f, 0.f, 0.f};  // (the delta side carries the each-row-once mask)
                float cT[4];
                tile_transpose(lds17, lane, cin, cT);
                const int j = 16 * t + 4 * (w & 3) + (w >> 2);  // input dim of tile row w
#pragma unroll
                for (int ht = 0; ht < NH; ++ht) {
                    const f32x4 dW = contract16(gT[ht], cT);  // [hidden 16ht+4g+r][tile row w]
                    if (j < nin) {
                        gp[pW0 + (16 * ht + 4 * g + 0) * nin + j] = dW.x; gp[pW0 + (16 * ht + 4 * g + 1) * nin + j] = dW.y;
                        gp[pW0 + (16 * ht + 4 * g + 2) * nin + j] = dW.z; gp[pW0 + (16 * ht + 4 * g + 3) * nin + j] = dW.w;
                    }
                }
            }
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ht = 0; ht < NH; ++ht) {
                const float *a = wb1 + (t * NH + ht) * 4;
                acc = mfma4(a[0], g_pre[ht].x, acc);
                acc = mfma4(a[1], g_pre[ht].y, acc);
                acc = mfma4(a[2], g_pre[ht].z, acc);
                acc = mfma4(a[3], g_pre[ht].w, acc);
            }
            g_cond[t] = g_cond[t] + acc;
        }
    }
    CB_STAMP(3)
}

// One workgroup of SPL_TEAM waves per 16-row tile.  Every wave carries the tile's rows; the spline work (the bulk of the
// instruction stream) and the weight-gradient contractions are divided among them (spl_coupling / spl_coupling_bwd).
#ifndef SPL_TEAM_N
#define SPL_TEAM_N 4
#endif
enum { SPL_TEAM = SPL_TEAM_N };
enum { SPL_STASH = 7 };  // tiles of NTh f32x4 per lane the forward pass leaves per block for the backward pass

template <int NTh, int NH>
__global__ void __launch_bounds__(64 * SPL_TEAM) spl_grad_kernel(SplGradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TEAM = SPL_TEAM;
    // The early-stopping flag is REQUESTED here and looked at behind the prologue's loads (the first thing with a side effect
    // outside the workgroup comes later): as the kernel's first statement it was a cold round trip to memory of its own in front
    // of every launch (round 6).  The flag was written by an earlier launch: every wave reads the same value.
    // (a plain volatile load: the compiler's own wait sits at the first use.  As an inline-asm load waited for later, the register
    // allocator was free to copy the result BEFORE the data was there -- seen in nnest_spline_rows.hip)
    int stop_flag = a.stop ? *reinterpret_cast<const volatile int *>(a.stop) : 0;
    const SplTrainShape &ts = a.ts;
    const SplineShape &s = ts.s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = lane & 15, g = lane >> 4, tile = blockIdx.x, lane_k = lane;
    const int D = s.D, B = s.B;
    const int per_wave = ((16 * (D + 1) + SPL_TBATCH * 16 * 17) + 3) & ~3;
    float *buf = lds + (size_t)wv * per_wave;   // 16 x (D+1): layout exchange (per wave)
    float *lds17 = buf + 16 * (D + 1);           // SPL_TBATCH x 16 x 17: tile transposes (per wave)
    // the team's exchange buffer, twice: exchanges alternate between the two, which lets each do with ONE barrier (between its
    // writes and its reads) -- the barrier that kept the next exchange's writes off this one's reads is the next exchange's own
    // (round 6; 21 exchanges per minibatch)
    f32x4 *xch0 = reinterpret_cast<f32x4 *>(lds + (size_t)TEAM * per_wave);  // 2 x [TEAM][NTh + NH][64]
    constexpr int XCH_N = TEAM * (NTh + NH) * 64;
    int xsel = 0;
    auto next_xch = [&]() -> f32x4 * { f32x4 *p = xch0 + ((xsel & 1) ? XCH_N : 0); xsel ^= 1; return p; };
    float *ldred = reinterpret_cast<float *>(xch0 + 2 * XCH_N);  // [TEAM][16]
    // (lds_heads) per block: the conv fragments of output tile `wv` -- forward first, the transposed ones take their place once the
    // forward product has used them -- [B][2 NTh][2 NTh * 4][64] floats, and ActNorm's e^s | t as this lane holds them [B][4 NTh][64] f32x4
    constexpr int T2K = 2 * NTh;
    constexpr bool LDS_OK = T2K <= TEAM;
    const bool lds_heads = LDS_OK && a.lds_heads != 0;
    float *lconv = ldred + TEAM * 16;
    f32x4 *lact = reinterpret_cast<f32x4 *>(lconv + (size_t)s.B * T2K * T2K * 256);
    const int ntl = (int)gridDim.x - a.val_tiles;  // the batch's own tiles
    const bool vtile = tile >= ntl;
    const int mode = vtile ? (int)SPL_MODE_LOSS : a.mode;
    // rows_per_tile 8: the tile's 8 rows sit in BOTH halves of the 16 matrix-core columns (lanes w and w + 8 carry row w & 7).
    // The halves run the spline stage on different dimensions (spl_coupling_pair*), everything else is computed twice and the
    // sums over rows take the low half only (`okc`).
    constexpr bool DUP = NTh >= 2;  // (one tile per half: every wave has at most one super-tile per coupling, nothing to pair)
    const bool dup = DUP;
    const int row = (vtile ? tile - ntl : tile) * a.rows_per_tile + (dup ? (w & 7) : w);
    const bool ok = (dup || w < a.rows_per_tile) && row < (vtile ? a.Mv : a.M);
    const bool cmask = !dup || w < 8, okc = ok && cmask;
    float *gp = a.partial + (size_t)tile * ts.gw_floats;
    f32x4 *stash = reinterpret_cast<f32x4 *>(a.stash) + ((size_t)tile * TEAM + wv) * B * SPL_STASH * NTh * 64;  // per block: input halves, upper', ActNorm output, conv output

    // The image was written by another kernel, i.e. into other XCDs' L2s: from here every fragment load of the pass would be
    // a cold miss (1.5-2 us each, and the layers consume them in small dependent batches -- by the stamps that was most of the
    // kernel).  One dword per 128-byte line, all in flight at once, brings the image into this XCD's L2 first.
    // (the row index of this lane first: the rows' loads then queue behind the warm-up instead of behind a second round trip;
    // the warm-up is waited for only after the rows and their jitter are done -- the noise draws need no memory)
    long src = 0;
    if (ok) src = (a.perm && !vtile) ? a.perm[row] : row;
    float sink = 0.f;
    {
        const char *base = reinterpret_cast<const char *>(a.timg);
        const int n_lines = (int)(((size_t)B * ts.tblk_floats * sizeof(float) + 127) >> 7);
        for (int i = threadIdx.x; i < n_lines; i += 64 * TEAM) {
            const char *p = base + ((size_t)i << 7);
            asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p) : "memory");
        }
        // (and the ActNorm vectors of every block, the only packed parameters the pass reads: 2 D floats at the head of a block)
        const int lines_an = (2 * D * (int)sizeof(float) + 127 + 127) >> 7;
        for (int i = threadIdx.x; i < B * lines_an; i += 64 * TEAM) {
            const char *p = reinterpret_cast<const char *>(a.w + (size_t)(i / lines_an) * s.blk_params) + ((size_t)(i % lines_an) << 7);
            asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p) : "memory");
        }
    }

    // data = X[perm] + jitter * randn  (trainer.py:392)
    f32x4 xp[2][NTh], xs[2][NTh];
    load_tile<NTh>(vtile ? a.xv : a.x, src, ok, D, lane, xp);
    if (mode == SPL_MODE_GRAD && a.jitter != 0.f) {
        if (a.noise) {
            f32x4 nz[2][NTh];
            load_tile<NTh>(a.noise, row, ok, D, lane, nz);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < NTh; ++t) xp[c][t] = xp[c][t] + nz[c][t] * a.jitter;
        } else {
            const long p = a.noise_row0 + row;
#pragma unroll
            for (int t = 0; t < NTh; ++t) {
                f32x4 n0 = noise_normal4(a.seed, (uint64_t)p, (uint32_t)a.epoch, (uint32_t)(8 * t + 2 * g), NOISE_STREAM_JITTER);
                f32x4 n1 = noise_normal4(a.seed, (uint64_t)p, (uint32_t)a.epoch, (uint32_t)(8 * t + 2 * g + 1), NOISE_STREAM_JITTER);
                const int d0 = 32 * t + 8 * g;
                if (ok) {
                    if (d0 + 0 < D) xp[0][t].x += n0.x * a.jitter; if (d0 + 1 < D) xp[1][t].x += n0.y * a.jitter;
                    if (d0 + 2 < D) xp[0][t].y += n0.z * a.jitter; if (d0 + 3 < D) xp[1][t].y += n0.w * a.jitter;
                    if (d0 + 4 < D) xp[0][t].z += n1.x * a.jitter; if (d0 + 5 < D) xp[1][t].z += n1.y * a.jitter;
                    if (d0 + 6 < D) xp[0][t].w += n1.z * a.jitter; if (d0 + 7 < D) xp[1][t].w += n1.w * a.jitter;
                }
            }
        }
    }
    if constexpr (LDS_OK) {
        if (lds_heads) {
            // the wave's conv fragments of every block (three blocks' loads in flight at a time) and the ActNorm vectors, block b by
            // wave b mod TEAM (they depend on the lane only: every wave would compute the same)
            for (int b0 = 0; b0 < B; b0 += 3) {
                float cf[3][T2K * 4];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int b = b0 + u < B ? b0 + u : B - 1;
                    const float *src = a.timg + (size_t)b * ts.tblk_floats + (size_t)(wv < T2K ? wv : 0) * T2K * 256;
#pragma unroll
                    for (int i = 0; i < T2K * 4; ++i) cf[u][i] = src[i * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    if (b0 + u < B && wv < T2K) {
                        float *dst = lconv + ((size_t)(b0 + u) * T2K + wv) * T2K * 256;
#pragma unroll
                        for (int i = 0; i < T2K * 4; ++i) dst[i * 64 + lane] = cf[u][i];
                    }
                }
            }
            for (int b = wv; b < B; b += TEAM) {
                f32x4 es[2][NTh], tv[2][NTh];
                spl_actnorm_vecs<NTh>(ts, a.w + (size_t)b * s.blk_params, lane, es, tv);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int t = 0; t < NTh; ++t) {
                        lact[((size_t)b * 4 * NTh + hf * NTh + t) * 64 + lane] = es[hf][t];
                        lact[((size_t)b * 4 * NTh + 2 * NTh + hf * NTh + t) * 64 + lane] = tv[hf][t];
                    }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");  // the warm-up's target register is free again only now
    if (stop_flag) return;   // (uniform over the workgroup, in front of its first barrier)
    if constexpr (LDS_OK) {
        if (lds_heads) spl_team_barrier();   // (the ActNorm vectors are read by every wave)
    }
    spl_from_parity<NTh>(buf, D, s.nl, lane, xp, xs);
#ifdef NNEST_STAMP
    long long st_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_0 = wall_clock64(), st_a, cb_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cf_t[4] = {0, 0, 0, 0}, cv_t = 0, cv_a = st_0;
#define SPL_STAMP(i) { const long long st_n = wall_clock64(); st_t[i] += st_n - st_a; st_a = st_n; }
    st_a = st_0;
#else
#define SPL_STAMP(i)
#endif

    // ---- forward (networks.py:24-32), block inputs stashed -----------------------------------------------------------
    float ld = 0.f;
    SplTrunkFrags<NTh, NH> tfr;  // the trunk fragments of the coupling ahead (spl_coupling_pair)
    SplRawFrags<NH> rfr;         // and the last-layer fragments of its first pair of super-tiles
    if (DUP) {
        spl_trunk_load<NTh, NH>(a.timg + 2 * ts.conv_floats, lane, tfr);
        spl_rawfrags_load<NTh, NH>(a.timg + 2 * ts.conv_floats, s.SU, wv, lane, rfr);
    }
    for (int b = 0; b < B; ++b) {
        // (lane- and shape-derived values kept opaque per block: see the backward loop)
        const int lane_o = spl_opaque_v(lane_k), nu_o = spl_opaque_s(s.nu), nl_o = spl_opaque_s(s.nl), SL_o = spl_opaque_s(s.SL), SU_o = spl_opaque_s(s.SU);
        const int lane = lane_o;
        const float *blk = a.timg + (size_t)b * ts.tblk_floats;
        const float *pb = a.w + (size_t)b * s.blk_params;
        if (mode != SPL_MODE_LOSS) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < NTh; ++t) stash[((size_t)b * SPL_STASH * NTh + c * NTh + t) * 64 + lane] = xs[c][t];
        }
        f32x4 es[2][NTh], tv[2][NTh], av[2][NTh], c[2][NTh];
        bool staged = false;
        if constexpr (LDS_OK) {
            if (lds_heads) {
                staged = true;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int t = 0; t < NTh; ++t) {
                        es[hf][t] = lact[((size_t)b * 4 * NTh + hf * NTh + t) * 64 + lane];
                        tv[hf][t] = lact[((size_t)b * 4 * NTh + 2 * NTh + hf * NTh + t) * 64 + lane];
                    }
            }
        }
        if (!staged) spl_actnorm_vecs<NTh>(ts, pb, lane, es, tv);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int t = 0; t < NTh; ++t) av[hf][t] = xs[hf][t] * es[hf][t] + tv[hf][t];
        float ctf[LDS_OK ? T2K * 4 : 1];   // (lds_heads) the transposed conv fragments of this block: requested now, in LDS behind coupling 1
        if constexpr (LDS_OK) {
            if (staged) {
                spl_matmul_team_lds<NTh, TEAM, false>(lconv + ((size_t)b * T2K + (wv < T2K ? wv : 0)) * T2K * 256, lane, wv, next_xch(), av, c);
                if (mode != SPL_MODE_LOSS) load_frags<T2K * 4>(blk + ts.conv_floats + (size_t)(wv < T2K ? wv : 0) * T2K * 256, lane, ctf);
            }
        }
        if (!staged) spl_matmul_team<NTh, TEAM, false>(blk, lane, wv, next_xch(), av, c);
        if (mode != SPL_MODE_LOSS) {  // ActNorm and conv outputs: the backward pass does not repeat them
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int t = 0; t < NTh; ++t) {
                    stash[((size_t)b * SPL_STASH * NTh + (3 + hf) * NTh + t) * 64 + lane] = av[hf][t];
                    stash[((size_t)b * SPL_STASH * NTh + (5 + hf) * NTh + t) * 64 + lane] = c[hf][t];
                }
        }
        const float *f1 = blk + 2 * ts.conv_floats, *f2 = f1 + ts.cf[0];
        f32x4 *kpf = mode != SPL_MODE_LOSS ? reinterpret_cast<f32x4 *>(a.keep) + (((size_t)tile * TEAM + wv) * B + b) * 2 * spl_keep_floats4(NTh, NH) * 64 : nullptr;
#ifdef NNEST_STAMP
        { asm volatile("" :: "v"(c[0][0].x)); const long long n_ = wall_clock64(); cv_t += n_ - cv_a; }   // (stash + ActNorm + conv of this block)
#define SPL_FST , cf_t
#else
#define SPL_FST
#endif
        if constexpr (DUP) ld += spl_coupling_pair<NTh, NH, TEAM, false>(f1, SU_o, nu_o, s.tail, lane, c[0], c[1], wv, next_xch(), kpf, tfr, rfr, f2, SL_o SPL_FST);
        else ld += spl_coupling<NTh, NH, false, TEAM>(f1, SU_o, nu_o, s.tail, lane, c[0], c[1], wv, next_xch());
        if (mode != SPL_MODE_LOSS) {  // upper' conditions the second coupling: kept for the backward pass
#pragma unroll
            for (int t = 0; t < NTh; ++t) stash[((size_t)b * SPL_STASH * NTh + 2 * NTh + t) * 64 + lane] = c[1][t];
        }
        if constexpr (LDS_OK) {
            if (staged && mode != SPL_MODE_LOSS && wv < T2K) {   // (the wave's own slot: nobody else reads it)
                float *dst = lconv + ((size_t)b * T2K + wv) * T2K * 256;
#pragma unroll
                for (int i = 0; i < T2K * 4; ++i) dst[i * 64 + lane] = ctf[i];
            }
        }
        if constexpr (DUP) ld += spl_coupling_pair<NTh, NH, TEAM, false>(f2, SL_o, nl_o, s.tail, lane, c[1], c[0], wv, next_xch(), kpf ? kpf + spl_keep_floats4(NTh, NH) * 64 : nullptr, tfr, rfr,
                                                                  b + 1 < B ? blk + ts.tblk_floats + 2 * ts.conv_floats : nullptr, SU_o SPL_FST);
        else ld += spl_coupling<NTh, NH, false, TEAM>(f2, SL_o, nl_o, s.tail, lane, c[1], c[0], wv, next_xch());
#ifdef NNEST_STAMP
        cv_a = wall_clock64();
#endif
        if (lane < (DUP ? 8 : 16) && wv == 0) ld += blk[ts.tblk_floats - 4];  // (once per row: DUP adds the halves up below)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int t = 0; t < NTh; ++t) xs[hf][t] = c[hf][t];
    }
    ld = group_sum(ld);
    if (TEAM > 1) {  // the waves hold the log-det of their own super-tiles: add them up
        if (lane < 16) ldred[wv * 16 + lane] = ld;
        spl_team_barrier();
        ld = 0.f;
#pragma unroll
        for (int k = 0; k < TEAM; ++k) ld += ldred[k * 16 + w];
        spl_team_barrier();
    }
    if (DUP) ld += half_swap(ld);  // a row's log|det|: the shares of its two halves
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < NTh; ++t) ss += base_E4(xs[c][t], s.base_beta);
    ss = group_sum(ss);
    float lp = (okc && g == 0) ? (-ss + s.base_const * (float)D + ld) : 0.f;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) lp += __shfl_xor(lp, o);
    if (lane == 0 && wv == 0) gp[ts.gw_floats - 4] = lp;  // sum of log_probs over this tile's rows
    if (mode == SPL_MODE_LOSS) return;
    SPL_STAMP(0)

    // ---- backward: loss = -mean(log_probs)  (trainer.py:394) ---------------------------------------------------------
    const float invM = 1.0f / (float)a.mtot, gld = mode == SPL_MODE_VJP ? a.gld_in : -invM;
    f32x4 gs[2][NTh];
    if (mode == SPL_MODE_VJP) {
        f32x4 gp4[2][NTh];
        load_tile<NTh>(a.gz, row, ok, D, lane, gp4);
        spl_from_parity<NTh>(buf, D, s.nl, lane, gp4, gs);
    } else {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < NTh; ++t) gs[c][t] = ok ? base_dE4(xs[c][t], s.base_beta) * invM : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int b = B - 1; b >= 0; --b) {
        // Everything the unrolled body derives from the lane index and the half sizes is invariant over the blocks; hoisted
        // out of this loop it came to ~500 spilled SGPRs (and as many VGPR copies) that each iteration re-read one by one.
        // Kept opaque per iteration, the masks and offsets are recomputed where they are used.
        const int lane_o = spl_opaque_v(lane_k), nu_o = spl_opaque_s(s.nu), nl_o = spl_opaque_s(s.nl), SL_o = spl_opaque_s(s.SL), SU_o = spl_opaque_s(s.SU);
        const int lane = lane_o, w = lane & 15, g = lane >> 4;
        const float *blk = a.timg + (size_t)b * ts.tblk_floats;
        const float *pb = a.w + (size_t)b * s.blk_params;
        const float *f1 = blk + 2 * ts.conv_floats, *f2 = f1 + ts.cf[0], *f1b = f2 + ts.cf[1], *f2b = f1b + ts.cb[0];
        const int pblk = b * s.blk_params;
        f32x4 xin[2][NTh], es[2][NTh], tv[2][NTh], av[2][NTh], c[2][NTh];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int t = 0; t < NTh; ++t) xin[cc][t] = stash[((size_t)b * SPL_STASH * NTh + cc * NTh + t) * 64 + lane];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int t = 0; t < NTh; ++t) {
                av[hf][t] = stash[((size_t)b * SPL_STASH * NTh + (3 + hf) * NTh + t) * 64 + lane];
                c[hf][t] = stash[((size_t)b * SPL_STASH * NTh + (5 + hf) * NTh + t) * 64 + lane];
            }
        bool staged = false;
        if constexpr (LDS_OK) {
            if (lds_heads) {
                staged = true;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int t = 0; t < NTh; ++t) es[hf][t] = lact[((size_t)b * 4 * NTh + hf * NTh + t) * 64 + lane];
            }
        }
        if (!staged) spl_actnorm_vecs<NTh>(ts, pb, lane, es, tv);  // (e^s for the ActNorm gradients at the end of the block)
        SPL_STAMP(5)
        // upper' = RQS(upper; f1(lower)) is the conditioning input of the second coupling
        f32x4 up2[NTh];
#pragma unroll
        for (int t = 0; t < NTh; ++t) up2[t] = stash[((size_t)b * SPL_STASH * NTh + 2 * NTh + t) * 64 + lane];
        // second coupling: lower' = RQS(lower; f2(upper'))   (networks.py:589-598)
        SPL_STAMP(1)
        const int item_stride = ntl * 64;
        f32x4 *gq2 = reinterpret_cast<f32x4 *>(a.gbuf) + ((size_t)(2 * b + 1) * ts.SM * SPL_QT * ntl + tile) * 64;
        f32x4 *gq1 = reinterpret_cast<f32x4 *>(a.gbuf) + ((size_t)(2 * b + 0) * ts.SM * SPL_QT * ntl + tile) * 64;
        f32x4 *hq2 = reinterpret_cast<f32x4 *>(a.hbuf) + ((size_t)(2 * b + 1) * ntl + tile) * NH * 64;
        f32x4 *hq1 = reinterpret_cast<f32x4 *>(a.hbuf) + ((size_t)(2 * b + 0) * ntl + tile) * NH * 64;
        const f32x4 *kp = reinterpret_cast<const f32x4 *>(a.keep) + (((size_t)tile * TEAM + wv) * B + b) * 2 * spl_keep_floats4(NTh, NH) * 64;
        spl_coupling_bwd<NTh, NH, TEAM, DUP, false>(ts, f2, f2b, pblk + ts.p_f[1], nu_o, nl_o, SL_o, lane, ok, cmask, gld, lds17, gp, up2, c[0], gs[0], gs[1], wv, next_xch(), gq2, hq2, item_stride, kp + spl_keep_floats4(NTh, NH) * 64, TEAM > 1 ? (2 * b + 1) % (TEAM > 1 ? TEAM - 1 : 1) : 0
#ifdef NNEST_STAMP
            , cb_t
#endif
            );
        SPL_STAMP(2)
        // first coupling: upper' = RQS(upper; f1(lower))      (networks.py:582-588)
        spl_coupling_bwd<NTh, NH, TEAM, DUP, false>(ts, f1, f1b, pblk + ts.p_f[0], nl_o, nu_o, SU_o, lane, ok, cmask, gld, lds17, gp, c[0], c[1], gs[1], gs[0], wv, next_xch(), gq1, hq1, item_stride, kp, TEAM > 1 ? (2 * b) % (TEAM > 1 ? TEAM - 1 : 1) : 0
#ifdef NNEST_STAMP
            , cb_t
#endif
            );
        SPL_STAMP(3)
        // 1x1 conv c = a W: dLoss/dW[i][o] = sum_rows a[i] g_c[o];  g_a = g_c W^T
        {
            constexpr int T2 = 2 * NTh;
            float aT[T2][4], gT[T2][4];
            f32x4 av0[T2], gs0[T2];
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                av0[t] = okc ? av[t / NTh][t % NTh] : (f32x4){0.f, 0.f, 0.f, 0.f};
                gs0[t] = gs[t / NTh][t % NTh];
            }
            tile_transpose_batch<T2>(lds17, lane, av0, aT);
            tile_transpose_batch<T2>(lds17, lane, gs0, gT);
            SPL_STAMP(6)
            float *gW = gp + s.num_params + (size_t)b * D * D;
#pragma unroll
            for (int ti = 0; ti < T2; ++ti)
#pragma unroll
                for (int to = 0; to < T2; ++to) {
                    if (TEAM > 1 && (to & (TEAM - 1)) != wv) continue;  // output tiles dealt out over the team
                    const f32x4 dW = contract16(aT[ti], gT[to]);  // [a row 4g+r of tile ti][g_c row w of tile to]
                    const int dout = trow_dim(s, to / NTh, to % NTh, w);
                    const float dv[4] = {dW.x, dW.y, dW.z, dW.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int din = trow_dim(s, ti / NTh, ti % NTh, 4 * g + r);
                        if (din >= 0 && dout >= 0) gW[(size_t)din * D + dout] = dv[r];
                    }
                }
        }
#ifdef NNEST_STAMP
        __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
        SPL_STAMP(7)
        f32x4 ga[2][NTh];
        if constexpr (LDS_OK) {
            if (staged) spl_matmul_team_lds<NTh, TEAM, false>(lconv + ((size_t)b * T2K + (wv < T2K ? wv : 0)) * T2K * 256, lane, wv, next_xch(), gs, ga);
        }
        if (!staged) spl_matmul_team<NTh, TEAM, false>(blk + ts.conv_floats, lane, wv, next_xch(), gs, ga);
        SPL_STAMP(8)
        // ActNorm a = x e^s + t: g_s = sum_rows g_a x e^s, g_t = sum_rows g_a, g_x = g_a e^s  (the -1 of log|det| is added by the reducer)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int t = 0; t < NTh; ++t) {
                const f32x4 gx = ga[hf][t] * es[hf][t];
                const f32x4 gac = cmask ? ga[hf][t] : (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 dsum = rows_sum(gac * es[hf][t] * xin[hf][t]);
                const f32x4 tsum = rows_sum(gac);
                // every lane of every wave holds the sums: lane w < 4 of wave (hf NTh + t) mod TEAM writes register r = w
                if (w < 4 && (TEAM == 1 || wv == ((hf * NTh + t) & (TEAM - 1)))) {
                    const int d = tslot_dim(s, hf, t, w, g);
                    if (d >= 0) { gp[pblk + ts.p_s + d] = reg_of(dsum, w); gp[pblk + ts.p_t + d] = reg_of(tsum, w); }
                }
                gs[hf][t] = gx;
            }
        SPL_STAMP(4)
    }
#ifdef NNEST_STAMP
    if (tile == 0 && lane == 0)
        printf("spl_grad wave %d: fwd %lld | stash+actnorm %lld matmul %lld | c2_bwd %lld c1_bwd %lld | transposes %lld dW %lld matmulT %lld actnorm %lld (x10 ns)\n", wv,
               st_t[0], st_t[5], st_t[1], st_t[2], st_t[3], st_t[6], st_t[7], st_t[8], st_t[4]);
    if (tile == 0 && lane == 0)
        printf("   forward (3 blocks) wave %d: stash+actnorm+conv %lld | six couplings: trunk %lld | last layer %lld | keep stores + spline evaluation %lld | exchange %lld (x10 ns)\n", wv,
               cv_t, cf_t[0], cf_t[1], cf_t[2], cf_t[3]);
    if (tile == 0 && lane == 0)
        printf("   couplings (6) wave %d: trunk+transposes %lld | super-tiles %lld [kept parameters arrived %lld | evaluation fwd+bwd %lld | G stores + W3^T G %lld] | merge %lld | trunk backward %lld (x10 ns)\n", wv, cb_t[0], cb_t[1], cb_t[4], cb_t[5], cb_t[6], cb_t[2], cb_t[3]);
#endif
    if (mode == SPL_MODE_VJP) {
        f32x4 gp4[2][NTh];
        spl_to_parity<NTh>(buf, D, s.nl, lane, gs, gp4);
        if (wv == 0) store_tile<NTh>(a.gx, row, okc, D, lane, gp4);
    }
}

// ---- 3b: last-layer weight gradients of the conditioners, one GEMM over ALL rows of the batch ---------------------------------
// Item (coupling, super-tile s, parameter tile q): dW3[(dim 4s+g, param 4q+r)][hidden j] = sum over rows G[row][(g, r)] h3[row][j]
// and db3 = sum over rows G.  A = G^T and B = h3 are read straight in operand layout from the tiles' stores (lane (i, kk) of
// k-step m of tile T: row 4m + kk); one wave per item, 4 NH matrix instructions per 16 rows.
// Result: dW[hto] lane (g, j) reg r <-> packed index pW3 + ((4s+g) 23 + 4q+r) H + 16 hto + j; db (every lane) feature lane & 15.
template <int NH>
__device__ __forceinline__ void spl_w3_item(const float *__restrict__ gq_item /* [tiles][64][4] */, const float *__restrict__ hq /* [tiles][NH][64][4] */,
                                            int tiles, int lane, f32x4 (&dW)[NH], float &db) {
    const int i = lane & 15, kk = lane >> 4;
    const int offA = ((i >> 2) * 16 + kk) * 4 + (i & 3);  // + 16 m + 256 T
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) dW[ht] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dsum = 0.f;
    for (int T0 = 0; T0 < tiles; T0 += 4) {  // 16 + 16 NH loads in flight per round
        float av[4][4], bv[4][NH][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int T = T0 + u < tiles ? T0 + u : tiles - 1;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                av[u][m] = gq_item[(size_t)T * 256 + offA + 16 * m];
#pragma unroll
                for (int ht = 0; ht < NH; ++ht) bv[u][ht][m] = hq[((size_t)T * NH + ht) * 256 + offA + 16 * m];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool live = T0 + u < tiles;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float a = live ? av[u][m] : 0.f;
                dsum += a;
#pragma unroll
                for (int ht = 0; ht < NH; ++ht) dW[ht] = mfma4(a, bv[u][ht][m], dW[ht]);
            }
        }
    }
    dsum += __shfl_xor(dsum, 16);
    dsum += __shfl_xor(dsum, 32);
    db = dsum;
}

// (coupling index, super-tile, parameter tile) of item `it`, or false
__device__ __forceinline__ bool spl_w3_decode(const SplTrainShape &ts, int it, int &cidx, int &sidx, int &q) {
    const int per = ts.SM * SPL_QT;
    cidx = it / per;
    const int r = it - cidx * per;
    sidx = r / SPL_QT; q = r - sidx * SPL_QT;
    return cidx < 2 * ts.s.B && sidx < ((cidx & 1) ? ts.s.SL : ts.s.SU);
}

// the gradient-returning entry points (nnest_spline_loss_grad / _vjp): dW3, db3 into `grad`
template <int NTh, int NH>
__global__ void __launch_bounds__(256) spl_w3_kernel(SplGradArgs a, int tiles, float *__restrict__ grad) {
    const SplTrainShape &ts = a.ts;
    const int lane = threadIdx.x & 63, it = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a.stop && *a.stop) return;
    int cidx, sidx, q;
    if (!spl_w3_decode(ts, it, cidx, sidx, q)) return;
    const int b = cidx >> 1, c = cidx & 1, H = ts.s.H;
    const int nin = c ? ts.s.nu : ts.s.nl, nout = c ? ts.s.nl : ts.s.nu;
    const int pW3 = b * ts.s.blk_params + ts.p_f[c] + H * nin + H + 2 * (H * H + H), pb3 = pW3 + SPL_P * nout * H;
    f32x4 dW[NH];
    float db;
    spl_w3_item<NH>(a.gbuf + ((size_t)(cidx * ts.SM + sidx) * SPL_QT + q) * tiles * 256, a.hbuf + (size_t)cidx * tiles * NH * 256, tiles, lane, dW, db);
    const int g = lane >> 4, j = lane & 15;
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) {
        const float dv[4] = {dW[ht].x, dW[ht].y, dW[ht].z, dW[ht].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jo = 4 * sidx + g, pp = 4 * q + r;
            if (jo < nout && pp < SPL_P) grad[pW3 + (jo * SPL_P + pp) * H + 16 * ht + j] = dv[r];
        }
    }
    if (lane < 16) {
        const int jo = 4 * sidx + (lane >> 2), pp = 4 * q + (lane & 3);
        if (jo < nout && pp < SPL_P) grad[pb3 + jo * SPL_P + pp] = db;
    }
}

// packed offset o inside a block: one of the conditioners' last-layer parameters (the tail of each conditioner)?
__host__ __device__ inline bool spl_is_w3(const SplTrainShape &ts, int o) {
    const int H = ts.s.H;
    const int t0 = ts.p_f[0] + H * ts.s.nl + H + 2 * (H * H + H), t1 = ts.p_f[1] + H * ts.s.nu + H + 2 * (H * H + H);
    return (o >= t0 && o < ts.p_f[1]) || o >= t1;
}

// ---- 4: reduce ------------------------------------------------------------------------------------------------------
__global__ void spl_reduce_kernel(const float *__restrict__ partial, int tiles, SplTrainShape ts, const float *__restrict__ w,
                                  float *__restrict__ grad, float *__restrict__ gwsum, float *__restrict__ loss_out, float loss_scale,
                                  float ldw /* sum over rows of dL/d(logdet): -1 for loss = -mean(log_probs) */, const int *__restrict__ stop) {
    if (stop && *stop) return;
    const int np = ts.s.num_params, D = ts.s.D, n = ts.gw_floats;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float acc = 0.f;
        if (i < np) {
            const int o = i % ts.s.blk_params;
            if (o >= ts.p_L && o < ts.p_f[0]) continue;  // L, S, U: from dLoss/dW (spl_lu_grad_kernel)
            if (spl_is_w3(ts, o)) continue;              // W3, b3 of the conditioners: spl_w3_item
            for (int t = 0; t < tiles; ++t) acc += partial[(size_t)t * n + i];
            if (o < ts.p_t) acc += ldw;  // logdet of ActNorm = sum(s) on every row
            grad[i] = acc;
        } else if (i < np + ts.s.B * D * D) {
            for (int t = 0; t < tiles; ++t) acc += partial[(size_t)t * n + i];
            gwsum[i - np] = acc;
        } else if (i == n - 4) {
            for (int t = 0; t < tiles; ++t) acc += partial[(size_t)t * n + i];
            if (loss_out) *loss_out = acc * loss_scale;
        }
    }
}

// ---- 5: W = (P Lm) Um  ->  L, S, U -------------------------------------------------------------------------------------
__global__ void spl_lu_grad_kernel(const float *__restrict__ w, const int *__restrict__ pi_inv, const int *__restrict__ pi,
                                   const float *__restrict__ gwsum, float *__restrict__ grad, SplTrainShape ts, float ldw,
                                   const int *__restrict__ stop) {
    if (stop && *stop) return;  // (uniform: before the barrier)
    const int D = ts.s.D, per = 2 * D * D, n = ts.s.B * per;
    extern __shared__ int lu_perm[];  // [pi | pi_inv], B x D each: the permutation look-ups leave the dependent-load chains
    int *spi = lu_perm, *spi_inv = lu_perm + ts.s.B * D;
    for (int i = threadIdx.x; i < ts.s.B * D; i += blockDim.x) { spi[i] = pi[i]; spi_inv[i] = pi_inv[i]; }
    __syncthreads();
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const int b = idx / per, o = idx % per;
        const float *pb = w + (size_t)b * ts.s.blk_params;
        const float *Lp = pb + ts.p_L, *Sp = pb + ts.p_S, *Up = pb + ts.p_U;
        const float *gW = gwsum + (size_t)b * D * D;
        float *gb = grad + (size_t)b * ts.s.blk_params;
        if (o < D * D) {  // dLoss/dL[r][k] = sum_j gW[pi^-1(r)][j] Um[k][j]   (k < r; other entries are unused: 0)
            const int r = o / D, k = o % D;
            float acc = 0.f;
            if (k < r) {
                const int i = spi_inv[b * D + r];
                const float sk = Sp[k];
#pragma unroll 8
                for (int j = k; j < D; ++j) {
                    const float gv = gW[(size_t)i * D + j], uv = Up[k * D + j];
                    acc += gv * (j > k ? uv : sk);
                }
            }
            gb[ts.p_L + o] = acc;
        } else {  // dLoss/dUm[k][j] = sum_i gW[i][j] Lm[pi(i)][k]   (k <= j)
            const int k = (o - D * D) / D, j = (o - D * D) % D;
            float acc = 0.f;
            if (k <= j) {
#pragma unroll 16
                for (int i = 0; i < D; ++i) {
                    const int r = spi[b * D + i];
                    const float lv = Lp[r * D + k], gv = gW[(size_t)i * D + j];
                    const float l = k < r ? lv : (k == r ? 1.f : 0.f);
                    acc += gv * l;
                }
            }
            if (k < j) gb[ts.p_U + k * D + j] = acc;
            else if (k == j) { gb[ts.p_S + k] = acc + ldw / Sp[k]; gb[ts.p_U + k * D + j] = 0.f; }  // logdet of the conv = sum log|S| on every row
            else gb[ts.p_U + k * D + j] = 0.f;
        }
    }
}

// ---- 6: Adam (torch/optim/adam.py _single_tensor_adam, coupled weight decay) -----------------------------------------------
__global__ void spl_adam_kernel(float *__restrict__ w, const float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v, int n,
                                float step_size, float inv_bc2s, float wd, const int *__restrict__ stop) {
    if (stop && *stop) return;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float gi = grad[i] + wd * w[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w[i] = w[i] - step_size * (mi / (sqrtf(vi) * inv_bc2s + eps));
    }
}

// one Adam step of one parameter (torch/optim/adam.py _single_tensor_adam, coupled weight decay)
__device__ __forceinline__ float spl_adam_one(float w, float g, float &m, float &v, float step_size, float inv_bc2s, float wd) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const float gi = g + wd * w;
    m = m + (gi - m) * (1.0f - b1);
    v = v * b2 + (1.0f - b2) * gi * gi;
    return w - step_size * (m / (sqrtf(v) * inv_bc2s + eps));
}

// ---- 6c: the training loop's whole parameter update in ONE launch per minibatch ------------------------------------------------
// (was: a reduce, an LU-gradient and an Adam/image kernel -- three launches and their gaps, ~30 us of a 170 us step)
// Workgroup b < B: the head of block b -- sums the tiles' ActNorm and dLoss/dW partials, takes dLoss/dW to dLoss/d(L, S, U)
//   (old L, S, U staged in LDS), steps the head, assembles W from the new values and writes the conv images and the log-det
//   constant.
// The next `n_w3` workgroups: one wave per (coupling, super-tile, parameter tile) item of the conditioners' last layers: dW3,
//   db3 as one contraction over all rows of the batch (spl_w3_item), stepped in registers and scattered to the images.
// The rest: the conditioners' trunks -- sum of the tiles' partials, step, scatter.
struct SplUpdateArgs {
    float *w, *m, *v;
    const float *partial;
    int tiles;
    const float *gbuf, *hbuf;
    const int *pos_f, *pos_b, *conv_src, *pi, *pi_inv;
    float *wmat, *timg;
    SplTrainShape ts;
    float step_size, inv_bc2s, wd, ldw;
    float *loss_out;
    float loss_scale;
    const int *stop;
    int n_w3;
};

// sum over the tiles' slices of one workspace element, in tile order; eight loads in flight (a load round trip per tile made
// the sums the longest part of the update)
__device__ __forceinline__ float spl_sum_tiles(const float *__restrict__ p, size_t stride, int tiles) {
    float acc = 0.f;
    for (int t0 = 0; t0 < tiles; t0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(t0 + u < tiles ? t0 + u : tiles - 1) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t0 + u < tiles ? v[u] : 0.f;
    }
    return acc;
}

__device__ __forceinline__ int spi_of(const int *__restrict__ pi, int b, int D, int i) { return pi[b * D + i]; }

struct SplAdamScatter {
    float *w, *m, *v, *timg;
    const int *pos_f, *pos_b;
    float step_size, inv_bc2s, wd;
    // N parameters of one lane (index < 0: none): every load issued before the first store
    template <int N>
    __device__ __forceinline__ void many(const int (&p)[N], const float (&g)[N]) const {
        float wi[N], mi[N], vi[N];
        int pf[N], pb[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int q = p[k] >= 0 ? p[k] : 0;
            wi[k] = w[q]; mi[k] = m[q]; vi[k] = v[q]; pf[k] = pos_f[q]; pb[k] = pos_b[q];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (p[k] < 0) continue;
            const float wn = spl_adam_one(wi[k], g[k], mi[k], vi[k], step_size, inv_bc2s, wd);
            m[p[k]] = mi[k]; v[p[k]] = vi[k]; w[p[k]] = wn;
            if (pf[k] >= 0) timg[pf[k]] = wn;
            if (pb[k] >= 0) timg[pb[k]] = wn;
        }
    }
    __device__ __forceinline__ void operator()(int p, float g) const {
        float mi = m[p], vi = v[p];
        const float wn = spl_adam_one(w[p], g, mi, vi, step_size, inv_bc2s, wd);
        m[p] = mi; v[p] = vi; w[p] = wn;
        const int pf = pos_f[p], pb = pos_b[p];
        if (pf >= 0) timg[pf] = wn;
        if (pb >= 0) timg[pb] = wn;
    }
};

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's outstanding global stores, a
// round trip to memory at every phase boundary of the head's update
__device__ __forceinline__ void spl_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NTh, int NH>
__global__ void __launch_bounds__(1024) spl_update_kernel(SplUpdateArgs a) {
    // (the early-stopping flag: requested now, looked at behind each role's first batch of loads and in front of its first store --
    // see spl_grad_kernel)
    // (a plain volatile load: the compiler's own wait sits at the first use.  As an inline-asm load waited for later, the register
    // allocator was free to copy the result BEFORE the data was there -- seen in nnest_spline_rows.hip)
    int stop_flag = a.stop ? *reinterpret_cast<const volatile int *>(a.stop) : 0;
#define SPL_STOP_CHECK() { if (stop_flag) return; }
    extern __shared__ float ulds[];
    const SplTrainShape &ts = a.ts;
    const SplineShape &s = ts.s;
    const int D = s.D, B = s.B, np = s.num_params, nhead = ts.p_f[0], n = ts.gw_floats, tid = threadIdx.x;
    if ((int)blockIdx.x < B) {
        // LDS: the old head, and five 64 x 64 operand matrices at stride 65 (zero-padded; stride 65 keeps a column walk off one
        // bank): G = the row-permuted dLoss/dW, Um = triu(U,1) + diag(S) and Lm = tril(L,-1) + I of the old values, and of the new
        // ones (UmN, LmN).  Dense operands make the products' inner loops two unconditional LDS reads per matrix instruction
        // (fetched through per-lane triangular conditions a 13-step product took 2-3 us).
        const int b = blockIdx.x, base = b * s.blk_params;
        constexpr int DP = 65, MAT = 64 * DP;
        float *hold = ulds, *G = ulds + nhead, *Um = G + MAT, *Lm = Um + MAT, *UmN = Lm + MAT, *LmN = UmN + MAT, *snew = LmN + MAT;
        int *spi = reinterpret_cast<int *>(snew + 64);
#ifdef NNEST_STAMP
        long long u_t[6]; u_t[0] = wall_clock64();
#define U_STAMP(i) u_t[i] = wall_clock64();
#else
#define U_STAMP(i)
#endif
        // Every global load of the head's update is issued up front, in one batch (old parameters, the tiles' partial sums, the Adam
        // state of the elements this lane will step): taken phase by phase they were three dependent round trips to memory.
        const int lane = tid & 63, wave = tid >> 6, nwv = (int)(blockDim.x >> 6), nt = (D + 15) >> 4, li = lane & 15, lk = lane >> 4;
        // (a) the elements of the two products' output tiles this lane holds: [tile u][4 accumulator registers + S on the diagonal]
        int hi[2][5];
        float hm[2][5], hv[2][5];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tile = wave + u * nwv;
            const bool live = tile < 2 * nt * nt, isU = tile >= nt * nt;
            const int tt = isU ? tile - nt * nt : tile, ti = tt / nt, tj = tt % nt, col = 16 * tj + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + 4 * lk + r;
                hi[u][r] = (live && row < D && col < D) ? (isU ? ts.p_U : ts.p_L) + row * D + col : -1;
            }
            const int rd = col - 16 * ti - 4 * lk;  // the diagonal element of this lane's column, if this lane holds it
            hi[u][4] = (live && isU && col < D && rd >= 0 && rd < 4) ? ts.p_S + col : -1;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int ii = hi[u][r] >= 0 ? hi[u][r] : 0;
                hm[u][r] = a.m[base + ii]; hv[u][r] = a.v[base + ii];
            }
        }
        // (b) ActNorm s, t: one element per thread; (c) the old head; (d) the tiles' partial sums of dLoss/dW, rows permuted as P
        // does: G[pi(i)][j] = gW[i][j].  Up to 16 tiles: every load of the phase is in flight before the first one is consumed (the
        // scheduler is held to it -- as first written the sums came out as one drained batch of 8 loads after another, a dozen
        // dependent round trips to memory).
        const bool an = tid < ts.p_L;
        float an_m = a.m[base + (an ? tid : 0)], an_v = a.v[base + (an ? tid : 0)];
        float an_g = 0.f;
        float hw[9];  // x_dim <= 64: the head is at most 8384 floats
        float gs[4];
        int gd[4];
        {
#pragma unroll
            for (int u = 0; u < 9; ++u) { const int i = tid + u * 1024; hw[u] = a.w[base + (i < nhead ? i : 0)]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = tid + u * 1024, ic = idx < D * D ? idx : 0;
                gd[u] = idx < D * D ? a.pi[b * D + ic / D] * DP + ic % D : -1;
            }
            const float *pan = a.partial + base + (an ? tid : 0);
            if (a.tiles <= 13 && D * D <= 3 * 1024) {
                // round 6: ONE round -- up to 13 tiles (a minibatch of 100 rows in 8-row tiles) x (3 elements of dLoss/dW + 1 ActNorm
                // element) = 52 loads in flight: at x_dim <= 55 the fourth element of every thread lay past D * D (a quarter of the
                // loads fetched element 0 again), and the second round was a second cold round trip to memory (~3.5 us)
#pragma unroll
                for (int u = 0; u < 4; ++u) gs[u] = 0.f;
                float pv[4][13];
#pragma unroll
                for (int t = 0; t < 13; ++t) {
                    const size_t off = (size_t)(t < a.tiles ? t : a.tiles - 1) * n;
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int idx = tid + u * 1024, ic = idx < D * D ? idx : 0;
                        pv[u][t] = a.partial[off + np + b * D * D + ic];
                    }
                    pv[3][t] = pan[off];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 13; ++t) {   // (summed in tile order, as the two-round form sums them)
                    const bool live = t < a.tiles;
#pragma unroll
                    for (int u = 0; u < 3; ++u) gs[u] += live ? pv[u][t] : 0.f;
                    an_g += live ? pv[3][t] : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
            } else if (a.tiles <= 16) {  // two rounds of 8 tiles x 5 elements in flight (the 1024-thread workgroup has 128 registers per lane)
#pragma unroll
                for (int u = 0; u < 4; ++u) gs[u] = 0.f;
#pragma unroll
                for (int t0 = 0; t0 < 16; t0 += 8) {
                    float pv[5][8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const size_t off = (size_t)(t0 + t < a.tiles ? t0 + t : a.tiles - 1) * n;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int idx = tid + u * 1024, ic = idx < D * D ? idx : 0;
                            pv[u][t] = a.partial[off + np + b * D * D + ic];
                        }
                        pv[4][t] = pan[off];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const bool live = t0 + t < a.tiles;
#pragma unroll
                        for (int u = 0; u < 4; ++u) gs[u] += live ? pv[u][t] : 0.f;
                        an_g += live ? pv[4][t] : 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = tid + u * 1024, ic = idx < D * D ? idx : 0;
                    gs[u] = spl_sum_tiles(a.partial + np + b * D * D + ic, n, a.tiles);
                }
                an_g = spl_sum_tiles(pan, n, a.tiles);
            }
            for (int idx = tid; idx < MAT; idx += blockDim.x) {  // the padding of the five matrices
                const int row = idx / DP, col = idx - row * DP;
                if (row >= D || col >= D) { G[idx] = 0.f; Um[idx] = 0.f; Lm[idx] = 0.f; UmN[idx] = 0.f; LmN[idx] = 0.f; }
            }
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const int i = tid + u * 1024;
                if (i >= nhead) continue;
                hold[i] = hw[u];
                if (i >= ts.p_U) {         // U[k][j]: the strict upper triangle counts (the diagonal of Um is S's)
                    const int o = i - ts.p_U, k = o / D, j = o - k * D;
                    if (k != j) Um[k * DP + j] = k < j ? hw[u] : 0.f;
                } else if (i >= ts.p_S) {  // S[k]
                    Um[(i - ts.p_S) * (DP + 1)] = hw[u];
                } else if (i >= ts.p_L) {  // L[r][k]: the strict lower triangle, 1 on the diagonal
                    const int o = i - ts.p_L, r = o / D, k = o - r * D;
                    Lm[r * DP + k] = k < r ? hw[u] : (k == r ? 1.f : 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (gd[u] >= 0) G[gd[u]] = gs[u];
            for (int i = tid; i < D; i += blockDim.x) spi[i] = a.pi[b * D + i];
        }
        SPL_STOP_CHECK()   // (every load of the phase has been consumed by now: the wait is free; uniform over the workgroup)
        spl_lds_barrier();
        U_STAMP(1)
        // dLoss/dL = tril(G Um^T, -1) and dLoss/d(Um) = triu(Lm^T G): two D^3 products on the matrix cores, one 16x16 output tile
        // per wave and round; every output element is one head parameter and takes its Adam step where it lands, the new value
        // going straight into the new dense operands.
        const float *Sp = hold + ts.p_S;
        const int KP = 16 * nt;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tile = wave + u * nwv;
            if (tile >= 2 * nt * nt) continue;
            const bool isU = tile >= nt * nt;
            const int tt = isU ? tile - nt * nt : tile, ti = tt / nt, tj = tt % nt;
            const int i = 16 * ti + li, j = 16 * tj + li;  // A row / B column of this lane
            // C[r][c] = sum_k G[r][k] Um[c][k]   |   C[c][j] = sum_k Lm[k][c] G[k][j]
            const float *pa = isU ? Lm + lk * DP + i : G + i * DP + lk, *pb = isU ? G + lk * DP + j : Um + j * DP + lk;
            const int sa = isU ? 4 * DP : 4, sb = isU ? 4 * DP : 4;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < KP; k0 += 4) acc = mfma4(pa[(k0 >> 2) * sa], pb[(k0 >> 2) * sb], acc);
            const float cv[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int e = hi[u][r];
                if (e < 0) continue;
                const int col = 16 * tj + li, row = r < 4 ? 16 * ti + 4 * lk + r : col;
                float g;
                if (r < 4) g = (isU ? row < col : col < row) ? cv[r] : 0.f;
                else {
                    const int rd = col - 16 * ti - 4 * lk;
                    g = (rd == 0 ? cv[0] : rd == 1 ? cv[1] : rd == 2 ? cv[2] : cv[3]) + a.ldw / Sp[col];  // + the conv's log-det term
                }
                const float wn = spl_adam_one(hold[e], g, hm[u][r], hv[u][r], a.step_size, a.inv_bc2s, a.wd);
                a.m[base + e] = hm[u][r]; a.v[base + e] = hv[u][r]; a.w[base + e] = wn;
                if (!isU) LmN[row * DP + col] = col < row ? wn : (col == row ? 1.f : 0.f);
                else if (r == 4) { UmN[col * (DP + 1)] = wn; snew[col] = wn; }
                else if (row != col) UmN[row * DP + col] = row < col ? wn : 0.f;
            }
        }
        if (an) {  // ActNorm s, t (+ the log-det term of s on every row)
            if (tid < ts.p_t) an_g += a.ldw;
            const float wn = spl_adam_one(hold[tid], an_g, an_m, an_v, a.step_size, a.inv_bc2s, a.wd);
            a.m[base + tid] = an_m; a.v[base + tid] = an_v; a.w[base + tid] = wn;
            if (tid < ts.p_t) hold[tid] = wn;  // (the new s, for the log-det constant: nobody reads the old one any more)
        }
        spl_lds_barrier();
        U_STAMP(2)
        // W = (P Lm) Um from the new values: a third product of the same form (x_dim <= 64: at most 16 output tiles, one per wave);
        // G's LDS takes W (row-major, stride D: the conv images' source table indexes it)
        float *Wm = a.wmat + (size_t)b * D * D, *Wl = G;
        if (wave < nt * nt) {
            const int ti = wave / nt, tj = wave % nt, i = 16 * ti + li, j = 16 * tj + li;
            const int pr = i < D ? spi[i] : 63;
            const float *pa = LmN + pr * DP + lk, *pb = UmN + lk * DP + j;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < KP; k0 += 4) acc = mfma4(pa[k0], pb[k0 * DP], acc);
            const float cv[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + 4 * lk + r, col = 16 * tj + li;
                if (row < D && col < D) { Wm[row * D + col] = cv[r]; Wl[row * D + col] = cv[r]; }
            }
        }
        spl_lds_barrier();
        U_STAMP(3)
        float *blk = a.timg + (size_t)b * ts.tblk_floats;
        for (int o = tid; o < 2 * ts.conv_floats; o += blockDim.x) {
            const int src = a.conv_src[o];
            blk[o] = src >= 0 ? Wl[src] : 0.f;
        }
        U_STAMP(4)
        // log|det| of ActNorm + conv (networks.py:650, :676): the terms in parallel, summed in d order
        for (int d = tid; d < D; d += blockDim.x) Um[d] = hold[ts.p_s + d] + logf(fabsf(snew[d]));
        spl_lds_barrier();
        if (tid == 0) {
            float acc = 0.f;
            for (int d = 0; d < D; ++d) acc += Um[d];
            blk[ts.tblk_floats - 4] = acc;
        }
        U_STAMP(5)
#ifdef NNEST_STAMP
        if (tid == 0 && b == 0) printf("spl_update head: loads+sums %lld | LU grads + Adam %lld | W %lld | conv images %lld | logdet %lld (x10 ns)\n",
                                       u_t[1] - u_t[0], u_t[2] - u_t[1], u_t[3] - u_t[2], u_t[4] - u_t[3], u_t[5] - u_t[4]);
#endif
        return;
    }
    const SplAdamScatter step = {a.w, a.m, a.v, a.timg, a.pos_f, a.pos_b, a.step_size, a.inv_bc2s, a.wd};
    const int H = s.H;
    if ((int)blockIdx.x < B + a.n_w3) {
        const int lane = tid & 63, it = ((int)blockIdx.x - B) * (int)(blockDim.x >> 6) + (tid >> 6);
        int cidx, sidx, q;
        if (!spl_w3_decode(ts, it, cidx, sidx, q)) return;
        const int b = cidx >> 1, c = cidx & 1;
        const int nin = c ? s.nu : s.nl, nout = c ? s.nl : s.nu;
        const int pW3 = b * s.blk_params + ts.p_f[c] + H * nin + H + 2 * (H * H + H), pb3 = pW3 + SPL_P * nout * H;
        f32x4 dW[NH];
        float db;
        spl_w3_item<NH>(a.gbuf + ((size_t)(cidx * ts.SM + sidx) * SPL_QT + q) * a.tiles * 256, a.hbuf + (size_t)cidx * a.tiles * NH * 256, a.tiles, lane, dW, db);
        const int g = lane >> 4, j = lane & 15;
        int pi_[4 * NH + 1];
        float gi_[4 * NH + 1];
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
            const float dv[4] = {dW[ht].x, dW[ht].y, dW[ht].z, dW[ht].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jo = 4 * sidx + g, pp = 4 * q + r;
                pi_[4 * ht + r] = (jo < nout && pp < SPL_P) ? pW3 + (jo * SPL_P + pp) * H + 16 * ht + j : -1;
                gi_[4 * ht + r] = dv[r];
            }
        }
        {
            const int jo = 4 * sidx + (lane >> 2), pp = 4 * q + (lane & 3);
            pi_[4 * NH] = (lane < 16 && jo < nout && pp < SPL_P) ? pb3 + jo * SPL_P + pp : -1;
            gi_[4 * NH] = db;
        }
        SPL_STOP_CHECK()
        step.template many<4 * NH + 1>(pi_, gi_);
        return;
    }
    // the trunks: W0, b0, W1, b1, W2, b2 of both conditioners of every block
    const int tr0 = H * s.nl + H + 2 * (H * H + H), tr1 = H * s.nu + H + 2 * (H * H + H), per = tr0 + tr1;
    const int nw = (int)gridDim.x - B - a.n_w3, first = (int)blockIdx.x - B - a.n_w3;
    bool stop_checked = false;
    for (int idx = first * (int)blockDim.x + tid; idx < B * per; idx += nw * (int)blockDim.x) {
        const int b = idx / per, o = idx - b * per;
        const int p = b * s.blk_params + (o < tr0 ? ts.p_f[0] + o : ts.p_f[1] + (o - tr0));
        const float gsum = spl_sum_tiles(a.partial + p, n, a.tiles);
        if (!stop_checked) { SPL_STOP_CHECK() stop_checked = true; }   // (behind the first sums' loads, in front of the first store)
        step(p, gsum);
    }
    if (first == 0 && tid == 0 && a.loss_out) {
        float acc = 0.f;
        for (int t = 0; t < a.tiles; ++t) acc += a.partial[(size_t)t * n + n - 4];
        *a.loss_out = acc * a.loss_scale;
    }
}

// ---- 7: end of an epoch (trainer.py:198-232), on the device so that the host does not have to drain the stream per epoch ----
struct SplTrainCtl {
    float best, last_train;
    int best_epoch, counter, epochs_run, stopped, improved;
};

// One workgroup: thread 0 keeps the books, then all threads copy the weights if the epoch improved the validation loss
// (best_model = deepcopy(netG), trainer.py:205-209).  `vpartial` != NULL: the validation loss comes straight from the `vtiles`
// per-tile sums of the forward-only tiles (their slots of the gradient workspace, stride `gw` floats) instead of losses[n_mb].
enum { SPL_EPOCH_END_THREADS = 512 };
__global__ void __launch_bounds__(1024) spl_epoch_end_kernel(SplTrainCtl *__restrict__ c, const float *__restrict__ losses, int n_mb, int n_train,
                                                             int n_valid, int epoch, int patience, float *__restrict__ epoch_losses,
                                                             const float *__restrict__ vpartial, int gw, int vtiles,
                                                             const float *__restrict__ w, float *__restrict__ best_w, int np) {
    __shared__ int flag[2];
    __shared__ float vsum;
    // (the validation rows' sum by the first wave, its loads in flight together: as a loop of thread 0 over up to 128 per-row values it
    // was 2 us of every epoch; the order of the additions is fixed: lane, then the shuffle tree)
    if (threadIdx.x < 64 && vpartial) {
        float acc = 0.f;
        for (int t = threadIdx.x; t < vtiles; t += 64) acc += vpartial[(size_t)t * gw + gw - 4];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        if (threadIdx.x == 0) vsum = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        flag[0] = c->stopped;
        flag[1] = 0;
        if (!flag[0]) {
            float tl = 0.f;
            for (int mb = 0; mb < n_mb; ++mb) tl += losses[mb];
            const float train_loss = tl / (float)n_train;            // trainer.py:403
            float vmean = losses[n_mb];
            if (vpartial) vmean = vsum * (-1.0f / (float)n_valid);
            const float valid_loss = vmean / (float)n_valid;         // trainer.py:418
            epoch_losses[2 * epoch] = train_loss;
            epoch_losses[2 * epoch + 1] = valid_loss;
            c->last_train = train_loss;
            c->epochs_run = epoch + 1;
            const int improved = valid_loss < c->best;               // trainer.py:205-209
            if (improved) { c->best = valid_loss; c->best_epoch = epoch + 1; c->counter = 0; }
            c->improved = improved;
            c->counter += 1;                                         // trainer.py:223-232
            if (c->counter > patience) c->stopped = 1;
            flag[1] = improved;
        }
    }
    __syncthreads();
    if (flag[0] || !flag[1]) return;
    for (int i = threadIdx.x; i < np; i += blockDim.x) best_w[i] = w[i];
}

// ---- ActNorm data-dependent initialisation (networks.py:698-705): one workgroup, rows in tiles of 16 -------------------------
struct SplInitArgs {
    const float *timg;
    float *w;
    SplTrainShape ts;
    const float *x;  // [N, D] rows already jittered / gathered
    int N;
    float *scratch;  // [tiles][2 NTh][64] f32x4
};

template <int NTh, int NH>
__global__ void __launch_bounds__(512) spl_init_kernel(SplInitArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const SplTrainShape &ts = a.ts;
    const SplineShape &s = ts.s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, w = lane & 15, g = lane >> 4;
    const int D = s.D, ntiles = (a.N + 15) >> 4;
    constexpr int T2 = 2 * NTh;
    float *buf = lds + (size_t)wave * 16 * (D + 1);
    float *red = lds + (size_t)nw * 16 * (D + 1);  // [nw][T2*16] partial sums, then [T2*16] results
    float *res = red + nw * T2 * 16;
    f32x4 *scr = reinterpret_cast<f32x4 *>(a.scratch);
    for (int tile = wave; tile < ntiles; tile += nw) {
        const int row = tile * 16 + w;
        f32x4 xp[2][NTh], xs[2][NTh];
        load_tile<NTh>(a.x, row, row < a.N, D, lane, xp);
        spl_from_parity<NTh>(buf, D, s.nl, lane, xp, xs);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < NTh; ++t) scr[((size_t)tile * T2 + c * NTh + t) * 64 + lane] = xs[c][t];
    }
    __syncthreads();
    for (int b = 0; b < s.B; ++b) {
        float *pb = a.w + (size_t)b * s.blk_params;
        // pass 0: mean; pass 1: unbiased variance
        for (int pass = 0; pass < 2; ++pass) {
            f32x4 acc[2][NTh];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < NTh; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int tile = wave; tile < ntiles; tile += nw) {
                const bool ok = tile * 16 + w < a.N;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int t = 0; t < NTh; ++t) {
                        f32x4 v = scr[((size_t)tile * T2 + c * NTh + t) * 64 + lane];
                        if (pass == 1) {
                            const f32x4 mu = *reinterpret_cast<const f32x4 *>(res + ((c * NTh + t) * 4 + g) * 4);
                            v = (v - mu) * (v - mu);
                        }
                        if (ok) acc[c][t] = acc[c][t] + v;
                    }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < NTh; ++t) {
                    const f32x4 sm = rows_sum(acc[c][t]);
                    if (w == 0) *reinterpret_cast<f32x4 *>(red + (size_t)wave * T2 * 16 + ((c * NTh + t) * 4 + g) * 4) = sm;
                }
            __syncthreads();
            for (int i = threadIdx.x; i < T2 * 16; i += blockDim.x) {
                float tot = 0.f;
                for (int k = 0; k < nw; ++k) tot += red[(size_t)k * T2 * 16 + i];
                const int tt = i >> 4, gg = (i >> 2) & 3, r = i & 3;
                const int d = tslot_dim(s, tt / NTh, tt % NTh, r, gg);
                if (pass == 0) res[i] = tot / (float)a.N;
                else if (d >= 0) {
                    const float mean = res[i];
                    const float sv = -logf(sqrtf(tot / (float)(a.N - 1)));  // s = -log std (unbiased)
                    pb[ts.p_s + d] = sv;
                    pb[ts.p_t + d] = -(mean * expf(sv));                   // t = -mean(x e^s)
                }
            }
            __syncthreads();
        }
        __threadfence();
        __syncthreads();
        // push the rows through block b
        const float *blk = a.timg + (size_t)b * ts.tblk_floats;
        const float *f1 = blk + 2 * ts.conv_floats, *f2 = f1 + ts.cf[0];
        for (int tile = wave; tile < ntiles; tile += nw) {
            f32x4 xs[2][NTh], es[2][NTh], tv[2][NTh], av[2][NTh], c[2][NTh];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int t = 0; t < NTh; ++t) xs[cc][t] = scr[((size_t)tile * T2 + cc * NTh + t) * 64 + lane];
            spl_actnorm_vecs<NTh>(ts, pb, lane, es, tv);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int t = 0; t < NTh; ++t) av[hf][t] = xs[hf][t] * es[hf][t] + tv[hf][t];
            spl_matmul<NTh>(blk, lane, av, c);
            spl_coupling<NTh, NH, false>(f1, s.SU, s.nu, s.tail, lane, c[0], c[1]);
            spl_coupling<NTh, NH, false>(f2, s.SL, s.nl, s.tail, lane, c[1], c[0]);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int t = 0; t < NTh; ++t) scr[((size_t)tile * T2 + cc * NTh + t) * 64 + lane] = c[cc][t];
        }
        __syncthreads();
    }
}

// (above 64 KiB of dynamic LDS a kernel has to be told so once)
#define SPLT_LAUNCH(K, grid, block, ldsb, st, ...)                                                                    \
    do {                                                                                                              \
        if ((size_t)(ldsb) > 64 * 1024) {                                                                             \
            hipError_t e__ = hipFuncSetAttribute(reinterpret_cast<const void *>(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ldsb)); \
            if (e__ != hipSuccess) return e__;                                                                        \
        }                                                                                                             \
        hipLaunchKernelGGL((K), dim3(grid), dim3(block), ldsb, st, __VA_ARGS__);                                              \
    } while (0)

#define DISPATCH_SPLT(KERNEL, sp, grid, block, ldsb, st, ...)                                                         \
    do {                                                                                                              \
        const int key__ = (sp).NTh * 10 + (sp).NH;                                                                    \
        switch (key__) {                                                                                              \
            case 11: SPLT_LAUNCH((KERNEL<1, 1>), grid, block, ldsb, st, __VA_ARGS__); break;               \
            case 21: SPLT_LAUNCH((KERNEL<2, 1>), grid, block, ldsb, st, __VA_ARGS__); break;               \
            case 31: SPLT_LAUNCH((KERNEL<3, 1>), grid, block, ldsb, st, __VA_ARGS__); break;               \
            case 41: SPLT_LAUNCH((KERNEL<4, 1>), grid, block, ldsb, st, __VA_ARGS__); break;               \
            case 12: SPLT_LAUNCH((KERNEL<1, 2>), grid, block, ldsb, st, __VA_ARGS__); break;               \
            case 22: SPLT_LAUNCH((KERNEL<2, 2>), grid, block, ldsb, st, __VA_ARGS__); break;               \
            default: return hipErrorInvalidConfiguration;                                                             \
        }                                                                                                             \
    } while (0)

static int grad_tiles(const SplGradArgs &a) { return (a.M + a.rows_per_tile - 1) / a.rows_per_tile; }

static hipError_t launch_grad(const SplGradArgs &a_in, hipStream_t st) {
    SplGradArgs a = a_in;
    const int tiles = grad_tiles(a);
    const int per_wave = ((16 * (a.ts.s.D + 1) + SPL_TBATCH * 16 * 17) + 3) & ~3;
    size_t ldsb = (size_t)(SPL_TEAM * per_wave + 2 * SPL_TEAM * (a.ts.s.NTh + a.ts.s.NH) * 64 * 4 + SPL_TEAM * 16) * sizeof(float);
    // the blocks' conv fragments and ActNorm vectors in LDS where they fit (x_dim 50, three blocks: 48 + 12 KB on top of 61 KB)
    static const bool heads_off = [] { const char *e = getenv("NNEST_SPL_LDS_HEADS"); return e && !strcmp(e, "0"); }();   // (diagnostic)
    const int T2 = 2 * a.ts.s.NTh;
    const size_t heads_b = (size_t)a.ts.s.B * (T2 * T2 * 256 + 4 * a.ts.s.NTh * 64 * 4) * sizeof(float);
    a.lds_heads = (!heads_off && T2 <= SPL_TEAM && ldsb + heads_b <= (size_t)156 * 1024) ? 1 : 0;
    if (a.lds_heads) ldsb += heads_b;
    DISPATCH_SPLT(spl_grad_kernel, a.ts.s, tiles + a.val_tiles, 64 * SPL_TEAM, ldsb, st, a);
    return hipGetLastError();
}

// dW3, db3 of every conditioner from the rows' stores of the gradient launch
static hipError_t launch_w3(const SplGradArgs &a, float *grad, hipStream_t st) {
    const int items = 2 * a.ts.s.B * a.ts.SM * SPL_QT, tiles = grad_tiles(a);
    DISPATCH_SPLT(spl_w3_kernel, a.ts.s, (items + 3) / 4, 256, 0, st, a, tiles, grad);
    return hipGetLastError();
}

static hipError_t launch_init(const SplInitArgs &a, hipStream_t st) {
    const int nw = 8, T2 = 2 * a.ts.s.NTh;
    const size_t ldsb = (size_t)(nw * 16 * (a.ts.s.D + 1) + nw * T2 * 16 + T2 * 16) * sizeof(float);
    DISPATCH_SPLT(spl_init_kernel, a.ts.s, 1, 64 * nw, ldsb, st, a);
    return hipGetLastError();
}

}  // namespace nnest

// ---- host orchestration ----------------------------------------------------------------------------------------------------
static int ensure_train_state(nnest_spline *h, int max_rows, hipStream_t st) {
    const SplTrainShape ts = make_train_shape(h->s);
    const int D = h->s.D, B = h->s.B;
    const size_t nb = (size_t)h->s.num_params * sizeof(float);
    if (!h->w_dev) {
        // (w_dev with zeroed slack behind it: the rows form's last-layer tiles read whole 16-row tiles, the last coupling's last tile past
        // the end of the packed vector -- times 0 or never stored, but read)
        SHIP_TRY(hipMalloc((void **)&h->w_dev, nb + SPL_W_SLACK_BYTES));
        SHIP_TRY(hipMemsetAsync(reinterpret_cast<char *>(h->w_dev) + nb, 0, SPL_W_SLACK_BYTES, st));
        SHIP_TRY(hipMalloc((void **)&h->adam_m, nb));
        SHIP_TRY(hipMalloc((void **)&h->adam_v, nb));
        SHIP_TRY(hipMalloc((void **)&h->best_w, nb));
        SHIP_TRY(hipMalloc((void **)&h->grad, nb));
        SHIP_TRY(hipMalloc((void **)&h->pi_dev, (size_t)2 * B * D * sizeof(int)));
        SHIP_TRY(hipMalloc((void **)&h->wmat, (size_t)B * D * D * sizeof(float) + SPL_W_SLACK_BYTES));   // (zeroed slack: the rows form reads whole 16-row groups)
        SHIP_TRY(hipMemsetAsync(reinterpret_cast<char *>(h->wmat) + (size_t)B * D * D * sizeof(float), 0, SPL_W_SLACK_BYTES, st));
        SHIP_TRY(hipMalloc((void **)&h->gwsum, (size_t)B * D * D * sizeof(float)));
        SHIP_TRY(hipMalloc((void **)&h->timg, (size_t)ts.timage_floats * sizeof(float)));
        SHIP_TRY(hipMalloc((void **)&h->losses_dev, 1024 * sizeof(float)));
        SHIP_TRY(hipMalloc((void **)&h->pos_dev, (2 * (size_t)h->s.num_params + 2 * (size_t)ts.conv_floats) * sizeof(int)));
        SHIP_TRY(hipMemsetAsync(h->pos_dev, 0xFF, 2 * (size_t)h->s.num_params * sizeof(int), st));
        hipLaunchKernelGGL(spl_build_pos_kernel, dim3(256), dim3(256), 0, st, h->pos_dev, h->pos_dev + h->s.num_params, h->pos_dev + 2 * (size_t)h->s.num_params, ts);
        SHIP_TRY(hipGetLastError());
        SHIP_TRY(hipMemsetAsync(h->adam_m, 0, nb, st));
        SHIP_TRY(hipMemsetAsync(h->adam_v, 0, nb, st));
        h->adam_step = 0;
        h->w_dev_current = false;
    }
    const int tiles = (max_rows + 7) / 8 > 40 ? (max_rows + 7) / 8 : 40;
    if (tiles > h->partial_tiles) {
        if (h->partial) { (void)hipFree(h->partial); (void)hipFree(h->stash); (void)hipFree(h->gbuf); (void)hipFree(h->hbuf); (void)hipFree(h->keep); }
        SHIP_TRY(hipMalloc((void **)&h->partial, (size_t)tiles * ts.gw_floats * sizeof(float)));
        SHIP_TRY(hipMalloc((void **)&h->stash, (size_t)tiles * SPL_TEAM * B * SPL_STASH * h->s.NTh * 64 * 4 * sizeof(float)));
        SHIP_TRY(hipMalloc((void **)&h->gbuf, (size_t)2 * B * ts.SM * SPL_QT * tiles * 256 * sizeof(float)));
        SHIP_TRY(hipMalloc((void **)&h->hbuf, (size_t)2 * B * tiles * h->s.NH * 256 * sizeof(float)));
        SHIP_TRY(hipMalloc((void **)&h->keep, (size_t)tiles * SPL_TEAM * B * 2 * spl_keep_floats4(h->s.NTh, h->s.NH) * 256 * sizeof(float)));
        h->partial_tiles = tiles;
    }
    if (!h->w_dev_current) {
        SHIP_TRY(hipMemcpyAsync(h->w_dev, h->w.data(), nb, hipMemcpyHostToDevice, st));
        std::vector<int> pi((size_t)2 * B * D, 0);  // [pi | pi^-1]
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < D; ++i)
                for (int k = 0; k < D; ++k)
                    if (h->perm[((size_t)b * D + i) * D + k] != 0.f) { pi[b * D + i] = k; pi[(size_t)B * D + b * D + k] = i; }
        SHIP_TRY(hipMemcpyAsync(h->pi_dev, pi.data(), pi.size() * sizeof(int), hipMemcpyHostToDevice, st));
        SHIP_TRY(hipStreamSynchronize(st));
        h->w_dev_current = true;
    }
    return NNEST_OK;
}

// x_dim > 32: 8-row tiles, the rows held in both halves of the 16 columns (spl_grad_kernel): a minibatch of 100 rows is 13
// workgroups; up to x_dim 32 (one tile per half) the plain 16-row tiles
static int rows_per_tile(const nnest::SplineShape &s) { return s.NTh >= 2 ? 8 : 16; }

static int build_timage(nnest_spline *h, const SplTrainShape &ts, hipStream_t st, const int *stop = nullptr) {
    hipLaunchKernelGGL(spl_assemble_kernel, dim3(64), dim3(256), 0, st, h->w_dev, h->pi_dev, h->wmat, ts, stop);
    hipLaunchKernelGGL(spl_timage_kernel, dim3(512), dim3(256), 0, st, h->w_dev, h->wmat, h->timg, ts, stop);
    SHIP_TRY(hipGetLastError());
    return NNEST_OK;
}

// after training (or an init): bring the packed weights back to the host master copy and rebuild the inference image
static int sync_to_host(nnest_spline *h, hipStream_t st) {
    SHIP_TRY(hipMemcpyAsync(h->w.data(), h->w_dev, (size_t)h->s.num_params * sizeof(float), hipMemcpyDeviceToHost, st));
    SHIP_TRY(hipStreamSynchronize(st));
    int rc = spline_build_image(h);
    if (rc) return rc;
    SHIP_TRY(hipMemcpyAsync(h->img, h->img_host.data(), h->img_host.size() * sizeof(float), hipMemcpyHostToDevice, st));
    SHIP_TRY(hipStreamSynchronize(st));
    return NNEST_OK;
}

extern "C" {

int nnest_spline_actnorm_init(nnest_spline_t *h, const float *x_dev, int N, void *stream) {
    if (!h || !x_dev) return spline_fail(NNEST_E_ARG, "NULL argument");
    if (N < 2) return spline_fail(NNEST_E_ARG, "ActNorm's data-dependent initialisation needs >= 2 rows (unbiased std), got %d", N);
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_train_state(h, 128, st);
    if (rc) return rc;
    const SplTrainShape ts = make_train_shape(h->s);
    if ((rc = build_timage(h, ts, st))) return rc;
    float *scratch = nullptr;
    const int tiles = (N + 15) / 16;
    SHIP_TRY(hipMalloc((void **)&scratch, (size_t)tiles * 2 * h->s.NTh * 64 * 4 * sizeof(float)));
    SplInitArgs a;
    a.timg = h->timg; a.w = h->w_dev; a.ts = ts; a.x = x_dev; a.N = N; a.scratch = scratch;
    hipError_t e = launch_init(a, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(scratch);
    if (e != hipSuccess) return spline_fail(NNEST_E_HIP, "spl_init_kernel: %s", hipGetErrorString(e));
    return sync_to_host(h, st);
}

int nnest_spline_loss_grad(nnest_spline_t *h, const float *x_dev, int M, float *grad_dev, float *loss_dev, void *stream) {
    if (!h || !x_dev || !grad_dev || !loss_dev) return spline_fail(NNEST_E_ARG, "NULL argument");
    if (M < 1) return spline_fail(NNEST_E_ARG, "M=%d", M);
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_train_state(h, M > 128 ? M : 128, st);
    if (rc) return rc;
    const SplTrainShape ts = make_train_shape(h->s);
    if (spline_rows_eligible(h->s, M)) {   // one row per workgroup (nnest_spline_rows.hip); the gradient instead of the step
        hipLaunchKernelGGL(spl_assemble_kernel, dim3(64), dim3(256), 0, st, h->w_dev, h->pi_dev, h->wmat, ts, (const int *)nullptr);
        if ((rc = spline_rows_prepare(h, ts, 128, 0, st, nullptr))) return rc;
        SplRowsBatch bt;
        memset(&bt, 0, sizeof(bt));
        bt.x = x_dev; bt.M = M; bt.mtot = M;
        SHIP_TRY(spline_rows_grad(h, ts, bt, st));
        SplRowsStep u;
        memset(&u, 0, sizeof(u));
        u.M = M; u.ldw = -1.0f; u.loss_out = loss_dev; u.loss_scale = -1.0f / (float)M; u.grad_out = grad_dev; u.gwsum_out = h->gwsum;
        SHIP_TRY(spline_rows_update(h, ts, u, st));
        hipLaunchKernelGGL(spl_lu_grad_kernel, dim3(64), dim3(256), 2 * (size_t)h->s.B * h->s.D * sizeof(int), st, h->w_dev, h->pi_dev + h->s.B * h->s.D, h->pi_dev, h->gwsum, grad_dev, ts, -1.0f, (const int *)nullptr);
        SHIP_TRY(hipGetLastError());
        return NNEST_OK;
    }
    if ((rc = build_timage(h, ts, st))) return rc;
    SplGradArgs a;
    memset(&a, 0, sizeof(a));
    a.timg = h->timg; a.w = h->w_dev; a.ts = ts; a.x = x_dev; a.M = M; a.mtot = M; a.partial = h->partial; a.stash = h->stash;
    a.mode = SPL_MODE_GRAD; a.gbuf = h->gbuf; a.hbuf = h->hbuf; a.keep = h->keep;
    a.rows_per_tile = rows_per_tile(h->s);
    SHIP_TRY(launch_grad(a, st));
    SHIP_TRY(launch_w3(a, grad_dev, st));
    const int tiles = grad_tiles(a);
    hipLaunchKernelGGL(spl_reduce_kernel, dim3(256), dim3(256), 0, st, h->partial, tiles, ts, h->w_dev, grad_dev, h->gwsum, loss_dev,
                       -1.0f / (float)M, -1.0f, (const int *)nullptr);
    hipLaunchKernelGGL(spl_lu_grad_kernel, dim3(64), dim3(256), 2 * (size_t)h->s.B * h->s.D * sizeof(int), st, h->w_dev, h->pi_dev + h->s.B * h->s.D, h->pi_dev, h->gwsum, grad_dev, ts, -1.0f, (const int *)nullptr);
    SHIP_TRY(hipGetLastError());
    return NNEST_OK;
}

int nnest_spline_vjp(nnest_spline_t *h, const float *x_dev, const float *gz_dev, float gld, int M, float *grad_dev, float *gx_dev,
                     void *stream) {
    if (!h || !x_dev || !gz_dev || !grad_dev || !gx_dev) return spline_fail(NNEST_E_ARG, "NULL argument");
    if (M < 1) return spline_fail(NNEST_E_ARG, "M=%d", M);
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_train_state(h, M > 128 ? M : 128, st);
    if (rc) return rc;
    const SplTrainShape ts = make_train_shape(h->s);
    if ((rc = build_timage(h, ts, st))) return rc;
    SplGradArgs a;
    memset(&a, 0, sizeof(a));
    a.timg = h->timg; a.w = h->w_dev; a.ts = ts; a.x = x_dev; a.M = M; a.mtot = M; a.partial = h->partial; a.stash = h->stash;
    a.mode = SPL_MODE_VJP; a.gz = gz_dev; a.gx = gx_dev; a.gld_in = gld; a.gbuf = h->gbuf; a.hbuf = h->hbuf; a.keep = h->keep;
    a.rows_per_tile = rows_per_tile(h->s);
    SHIP_TRY(launch_grad(a, st));
    SHIP_TRY(launch_w3(a, grad_dev, st));
    hipLaunchKernelGGL(spl_reduce_kernel, dim3(256), dim3(256), 0, st, h->partial, grad_tiles(a), ts, h->w_dev, grad_dev, h->gwsum,
                       (float *)nullptr, 0.f, (float)M * gld, (const int *)nullptr);
    hipLaunchKernelGGL(spl_lu_grad_kernel, dim3(64), dim3(256), 2 * (size_t)h->s.B * h->s.D * sizeof(int), st, h->w_dev, h->pi_dev + h->s.B * h->s.D, h->pi_dev, h->gwsum, grad_dev, ts,
                       (float)M * gld, (const int *)nullptr);
    SHIP_TRY(hipGetLastError());
    return NNEST_OK;
}

int nnest_spline_adam_step(nnest_spline_t *h, const float *grad_dev, float lr, float weight_decay, void *stream) {
    if (!h || !grad_dev) return spline_fail(NNEST_E_ARG, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_train_state(h, 128, st);
    if (rc) return rc;
    h->adam_step += 1;
    const double bc1 = 1.0 - pow(0.9, (double)h->adam_step), bc2 = 1.0 - pow(0.999, (double)h->adam_step);
    hipLaunchKernelGGL(spl_adam_kernel, dim3(256), dim3(256), 0, st, h->w_dev, grad_dev, h->adam_m, h->adam_v, h->s.num_params,
                       (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), weight_decay, (const int *)nullptr);
    SHIP_TRY(hipGetLastError());
    return sync_to_host(h, st);
}

int nnest_spline_train_form(const nnest_spline_t *h, int batch) { return (h && spline_rows_eligible(h->s, batch)) ? 1 : 0; }

int nnest_spline_train(nnest_spline_t *h, const float *xtrain_dev, int n_train, const float *xvalid_dev, int n_valid,
                       const int *perm_dev, const float *noise_dev, uint64_t seed, float jitter, int batch, int max_epochs,
                       int patience, float lr, float weight_decay, float *losses_host, nnest_train_result_t *result_host,
                       void *stream) {
    if (!h || !xtrain_dev || !xvalid_dev || !perm_dev || !result_host) return spline_fail(NNEST_E_ARG, "NULL argument");
    if (n_train < 1 || n_valid < 1 || batch < 1 || max_epochs < 0)
        return spline_fail(NNEST_E_ARG, "bad sizes n_train=%d n_valid=%d batch=%d max_epochs=%d", n_train, n_valid, batch, max_epochs);
    if (batch > 128) return spline_fail(NNEST_E_UNSUPPORTED, "batch_size=%d > 128", batch);
    hipStream_t st = (hipStream_t)stream;
    const int max_rows = n_valid + 128;  // a minibatch's tiles and the validation tiles share one launch
    int rc = ensure_train_state(h, max_rows, st);
    if (rc) return rc;
    const SplTrainShape ts = make_train_shape(h->s);
    const int np = h->s.num_params, D = h->s.D, B = h->s.B;
    const int n_mb = (n_train + batch - 1) / batch;
    if (n_mb + 1 > 1024) return spline_fail(NNEST_E_UNSUPPORTED, "more than 1023 minibatches per epoch");
    const size_t nb = (size_t)np * sizeof(float);
    SHIP_TRY(hipMemcpyAsync(h->best_w, h->w_dev, nb, hipMemcpyDeviceToDevice, st));  // best_model = deepcopy(netG)  trainer.py:194
    // The epoch bookkeeping (losses, best model, patience) runs on the device (spl_epoch_end_kernel): the host queues epochs
    // without draining the stream and looks at the state once per chunk of epochs, one chunk behind the queue; launches queued
    // past the stop do nothing.
    if (!h->ctl_dev) {
        SHIP_TRY(hipMalloc(&h->ctl_dev, sizeof(SplTrainCtl)));
        SHIP_TRY(hipHostMalloc(&h->ctl_host, 2 * sizeof(SplTrainCtl), hipHostMallocDefault));
    }
    if (h->epoch_losses_cap < max_epochs) {
        if (h->epoch_losses_dev) (void)hipFree(h->epoch_losses_dev);
        h->epoch_losses_dev = nullptr; h->epoch_losses_cap = 0;
        SHIP_TRY(hipMalloc((void **)&h->epoch_losses_dev, 2 * (size_t)(max_epochs > 0 ? max_epochs : 1) * sizeof(float)));
        h->epoch_losses_cap = max_epochs;
    }
    SplTrainCtl *ctl = (SplTrainCtl *)h->ctl_dev, *snap = (SplTrainCtl *)h->ctl_host;
    const int *stop = &ctl->stopped;
    SplTrainCtl c0;
    memset(&c0, 0, sizeof(c0));
    c0.best = INFINITY;
    snap[0] = c0; snap[1] = c0;
    SHIP_TRY(hipMemcpyAsync(ctl, &snap[0], sizeof(SplTrainCtl), hipMemcpyHostToDevice, st));
    const int adam_step0 = h->adam_step;
    // Shapes with at most two tiles per half (x_dim <= 64): per minibatch TWO launches -- the gradient pass and spl_update_kernel,
    // which keeps the training image current -- and the validation pass of an epoch rides along with the first gradient pass of
    // the next one (forward-only tiles on other CUs; the weights do not change in between), followed by the epoch's bookkeeping.
    // The larger shapes rebuild the image per minibatch (their conv images are too much work for one workgroup per block).
    const bool fused = h->s.NTh <= 2;
    // the rows form (nnest_spline_rows.hip): one row per workgroup, the parameter gradients contracted over the rows in its update kernel
    const bool rows = spline_rows_eligible(h->s, batch);
    if (rows) {
        hipLaunchKernelGGL(spl_assemble_kernel, dim3(64), dim3(256), 0, st, h->w_dev, h->pi_dev, h->wmat, ts, (const int *)nullptr);
        if ((rc = spline_rows_prepare(h, ts, 128, n_valid, st, nullptr))) return rc;
    }
    // How far the host may queue without looking: the books of an epoch stop the run when `counter` (epochs since the last improvement,
    // this one included) exceeds `patience`, so from a state read after epoch e0 with counter c0 the earliest epoch that can stop is
    // e0 + patience + 1 - c0.  Up to there the launches are queued blind (the validation pass of an epoch riding with the next epoch's
    // first gradient launch); THAT epoch ends with its own validation launch and its books, the stream is drained and the state read:
    // stopped, or a later check epoch.  (Was: a snapshot every 8 epochs, looked at one chunk later -- up to 16 epochs of launches
    // queued past the stop, each a no-op but a launch: 1.4 ms of a 40 ms call.)
    int check_epoch = patience;   // e0 = -1, c0 = 0
    bool stopped_seen = false, pending_validation = false;
    const int vtiles = (n_valid + rows_per_tile(h->s) - 1) / rows_per_tile(h->s);
    auto read_state = [&](int epoch) -> int {   // after the books of `epoch` have been queued
        SHIP_TRY(hipMemcpyAsync(&snap[0], ctl, sizeof(SplTrainCtl), hipMemcpyDeviceToHost, st));
        SHIP_TRY(hipStreamSynchronize(st));
        stopped_seen = snap[0].stopped != 0;
        check_epoch = epoch + patience + 1 - snap[0].counter;
        if (check_epoch <= epoch) check_epoch = epoch + 1;
        return NNEST_OK;
    };
    auto valid_args = [&]() {
        SplGradArgs a;
        memset(&a, 0, sizeof(a));
        a.timg = h->timg; a.w = h->w_dev; a.ts = ts; a.x = xvalid_dev; a.M = n_valid; a.mtot = n_valid;
        a.partial = h->partial; a.stash = h->stash; a.mode = SPL_MODE_LOSS; a.stop = stop;
        a.rows_per_tile = rows_per_tile(h->s);
        return a;
    };
    for (int epoch = 0; epoch < max_epochs && !stopped_seen; ++epoch) {
        for (int mb = 0; mb < n_mb; ++mb) {
            const int M = batch < n_train - mb * batch ? batch : n_train - mb * batch;
            SplGradArgs a;
            memset(&a, 0, sizeof(a));
            a.timg = h->timg; a.w = h->w_dev; a.ts = ts; a.x = xtrain_dev; a.perm = perm_dev + (size_t)epoch * n_train + (size_t)mb * batch;
            a.M = M; a.mtot = M; a.noise = noise_dev ? noise_dev + ((size_t)epoch * n_train + (size_t)mb * batch) * D : nullptr;
            a.seed = seed; a.noise_row0 = (long)mb * batch; a.epoch = epoch; a.jitter = jitter;
            a.partial = h->partial; a.stash = h->stash; a.mode = SPL_MODE_GRAD; a.stop = stop; a.gbuf = h->gbuf; a.hbuf = h->hbuf; a.keep = h->keep;
            a.rows_per_tile = rows_per_tile(h->s);
            const int tiles = grad_tiles(a);
            h->adam_step += 1;
            const double bc1 = 1.0 - pow(0.9, (double)h->adam_step), bc2 = 1.0 - pow(0.999, (double)h->adam_step);
            if (rows) {
                const bool ride = mb == 0 && pending_validation;  // the validation pass of the epoch before: forward-only rows behind the minibatch's
                SplRowsBatch bt;
                memset(&bt, 0, sizeof(bt));
                bt.x = a.x; bt.perm = a.perm; bt.M = M; bt.mtot = M; bt.noise = a.noise; bt.seed = seed; bt.noise_row0 = a.noise_row0; bt.epoch = epoch;
                bt.jitter = jitter; bt.stop = stop;
                if (ride) { bt.xv = xvalid_dev; bt.Mv = n_valid; }
                SHIP_TRY(spline_rows_grad(h, ts, bt, st));
                if (ride) {
                    hipLaunchKernelGGL(spl_epoch_end_kernel, dim3(1), dim3(SPL_EPOCH_END_THREADS), 0, st, ctl, h->losses_dev, n_mb, n_train, n_valid, epoch - 1, patience,
                                       h->epoch_losses_dev, spline_rows_rowlp(h) + M + 3, 1, n_valid, h->w_dev, h->best_w, np);
                    pending_validation = false;
                }
                SplRowsStep u;
                memset(&u, 0, sizeof(u));
                u.M = M; u.step_size = (float)((double)lr / bc1); u.inv_bc2s = (float)(1.0 / sqrt(bc2)); u.wd = weight_decay; u.ldw = -1.0f;
                u.loss_out = h->losses_dev + mb; u.loss_scale = -1.0f / (float)M; u.stop = stop;
                SHIP_TRY(spline_rows_update(h, ts, u, st));
            } else if (fused) {
                if (mb == 0 && epoch == 0 && (rc = build_timage(h, ts, st, stop))) return rc;
                const bool ride = mb == 0 && pending_validation;  // the validation pass of the epoch before
                if (ride) { a.val_tiles = vtiles; a.xv = xvalid_dev; a.Mv = n_valid; }
                SHIP_TRY(launch_grad(a, st));
                if (ride) {
                    hipLaunchKernelGGL(spl_epoch_end_kernel, dim3(1), dim3(SPL_EPOCH_END_THREADS), 0, st, ctl, h->losses_dev, n_mb, n_train, n_valid, epoch - 1, patience,
                                       h->epoch_losses_dev, h->partial + (size_t)tiles * ts.gw_floats, ts.gw_floats, vtiles, h->w_dev, h->best_w, np);
                    pending_validation = false;
                }
                SplUpdateArgs u;
                memset(&u, 0, sizeof(u));
                u.w = h->w_dev; u.m = h->adam_m; u.v = h->adam_v; u.partial = h->partial; u.tiles = tiles; u.gbuf = h->gbuf; u.hbuf = h->hbuf;
                u.pos_f = h->pos_dev; u.pos_b = h->pos_dev + np; u.conv_src = h->pos_dev + 2 * (size_t)np; u.pi = h->pi_dev; u.pi_inv = h->pi_dev + B * D;
                u.wmat = h->wmat; u.timg = h->timg; u.ts = ts;
                u.step_size = (float)((double)lr / bc1); u.inv_bc2s = (float)(1.0 / sqrt(bc2)); u.wd = weight_decay; u.ldw = -1.0f;
                u.loss_out = h->losses_dev + mb; u.loss_scale = -1.0f / (float)M; u.stop = stop;
                const int H = h->s.H, items = 2 * B * ts.SM * SPL_QT;
                const int trunk = B * (H * h->s.nl + H + H * h->s.nu + H + 4 * (H * H + H));
                u.n_w3 = (items + 15) / 16;
                const size_t ldsb = ((size_t)ts.p_f[0] + 5 * 64 * 65 + 64) * sizeof(float) + (size_t)D * sizeof(int);
                DISPATCH_SPLT(spl_update_kernel, h->s, B + u.n_w3 + (trunk + 1023) / 1024, 1024, ldsb, st, u);
            } else {
                if ((mb > 0 || epoch == 0) && (rc = build_timage(h, ts, st, stop))) return rc;
                SHIP_TRY(launch_grad(a, st));
                SHIP_TRY(launch_w3(a, h->grad, st));
                hipLaunchKernelGGL(spl_reduce_kernel, dim3(256), dim3(256), 0, st, h->partial, tiles, ts, h->w_dev, h->grad, h->gwsum,
                                   h->losses_dev + mb, -1.0f / (float)M, -1.0f, stop);
                hipLaunchKernelGGL(spl_lu_grad_kernel, dim3(64), dim3(256), 2 * (size_t)h->s.B * h->s.D * sizeof(int), st, h->w_dev, h->pi_dev + B * D, h->pi_dev, h->gwsum, h->grad, ts, -1.0f, stop);
                hipLaunchKernelGGL(spl_adam_kernel, dim3(256), dim3(256), 0, st, h->w_dev, h->grad, h->adam_m, h->adam_v, np,
                                   (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), weight_decay, stop);
            }
        }
        SHIP_TRY(hipGetLastError());
        if ((fused || rows) && epoch + 1 < max_epochs && epoch < check_epoch) { pending_validation = true; continue; }  // rides with the next epoch's first gradient pass
        // Trainer._validate (trainer.py:405-418): one full batch; mean, then / len(dataset)
        if (rows) {
            SplRowsBatch bt;
            memset(&bt, 0, sizeof(bt));
            bt.xv = xvalid_dev; bt.Mv = n_valid; bt.stop = stop;
            SHIP_TRY(spline_rows_grad(h, ts, bt, st));
            hipLaunchKernelGGL(spl_epoch_end_kernel, dim3(1), dim3(SPL_EPOCH_END_THREADS), 0, st, ctl, h->losses_dev, n_mb, n_train, n_valid, epoch, patience,
                               h->epoch_losses_dev, spline_rows_rowlp(h) + 3, 1, n_valid, h->w_dev, h->best_w, np);
            SHIP_TRY(hipGetLastError());
            if (epoch + 1 < max_epochs && epoch >= check_epoch && (rc = read_state(epoch))) return rc;
            continue;
        }
        if (!fused && (rc = build_timage(h, ts, st, stop))) return rc;
        {
            const SplGradArgs a = valid_args();
            SHIP_TRY(launch_grad(a, st));
        }
        hipLaunchKernelGGL(spl_epoch_end_kernel, dim3(1), dim3(SPL_EPOCH_END_THREADS), 0, st, ctl, h->losses_dev, n_mb, n_train, n_valid, epoch, patience,
                           h->epoch_losses_dev, h->partial, ts.gw_floats, vtiles, h->w_dev, h->best_w, np);
        SHIP_TRY(hipGetLastError());
        if (epoch + 1 < max_epochs && epoch >= check_epoch && (rc = read_state(epoch))) return rc;
    }
    SHIP_TRY(hipMemcpyAsync(h->w_dev, h->best_w, nb, hipMemcpyDeviceToDevice, st));  // netG.load_state_dict(best_model)  trainer.py:241
    SHIP_TRY(hipMemcpyAsync(&snap[0], ctl, sizeof(SplTrainCtl), hipMemcpyDeviceToHost, st));
    SHIP_TRY(hipStreamSynchronize(st));
    const SplTrainCtl fin = snap[0];
    if (losses_host && fin.epochs_run > 0)
        SHIP_TRY(hipMemcpy(losses_host, h->epoch_losses_dev, 2 * (size_t)fin.epochs_run * sizeof(float), hipMemcpyDeviceToHost));
    h->adam_step = adam_step0 + fin.epochs_run * n_mb;  // the steps queued past the stop did not happen
    result_host->epochs_run = fin.epochs_run; result_host->best_epoch = fin.best_epoch; result_host->best_validation_loss = fin.best;
    result_host->last_train_loss = fin.last_train; result_host->counter = fin.counter; result_host->stopped = fin.stopped;
    return sync_to_host(h, st);
}

}  // extern "C"
