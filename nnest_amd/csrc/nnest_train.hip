// nnest_train.hip -- K5: Trainer.train's epoch loop (reference nnest/trainer.py:134-245, :384-418) as ONE
// persistent single-workgroup kernel on gfx950: forward, hand-written backward, Adam, validation, early
// stopping and best-model restore all on device; the host only launches and reads the result struct.
//
// Why one workgroup: a minibatch is 100 rows x ~12.7 kFLOP x 3 (fwd + recompute + bwd) = 4 MFLOP, and the
// Adam steps are strictly sequential (9 per epoch at 1000 live points), so the loop is latency-bound; a
// grid barrier (>= 4 us, MI355X_MICROARCH.md "barrier-xcd") per phase would cost more than the phase.
//
// Per minibatch (M <= 128 rows = up to 8 row tiles of 16, one wave each):
//   A  row-tile waves: forward (same MFMA tile code as inference) -> per-row log_prob, loss;
//      backward block by block WITHOUT stored activations: a coupling block is invertible, so its input
//      is recovered from its output ((y - t) e^{-ls}) after recomputing the two MLPs from the untouched
//      conditioning half; delta-propagation uses transposed weight fragments (the "backward image").
//      For each (block, net) the per-row gradients G and activations are staged in LDS as [row][16].
//   B  tile-owner waves: dW[out,in] = sum_rows G[row,out] * Act[row,in] as MFMA 16x16x4 over all rows
//      (A and B operands are both plain [row][16] LDS reads; biases use B = 1), scattered into the
//      packed gradient vector.  No atomics: every gradient element has exactly one producer, so training
//      is bitwise reproducible (which is what lets every rank of a multi-GPU run train an identical
//      replica with no weight broadcast).
//   C  all threads: Adam with coupled weight decay over the packed vector (torch.optim.Adam,
//      trainer.py:121-122), then rebuild the forward + backward fragment images.
#include <string.h>
#include "nnest_internal.h"
#include "solo_tile.h"

namespace nnest {

// NNEST_STAMP: diagnostic build only (tools/time_train.py --stamps): cycles per phase of a minibatch
#ifdef NNEST_STAMP
#define TSTAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TACC(acc, t1, t0) acc += (t1) - (t0)
#else
#define TSTAMP(v) do { } while (0)
#define TACC(acc, t1, t0) do { } while (0)
#endif

// GRAD_ONLY: loss and dLoss/dw of one batch.  VJP: the flow as one stage of a composite model (fast/slow hierarchy): given the
// batch x, an upstream gradient gz = dL/d(output) and gld = dL/d(logdet) per row, it returns dL/dw and gx = dL/dx.
enum { TRAIN_MODE_EPOCHS = 0, TRAIN_MODE_GRAD_ONLY = 1, TRAIN_MODE_VJP = 2 };
static const int TRAIN_THREADS = 512;  // 8 waves
static const int TRAIN_WAVES = 8;
static const int TRAIN_MAX_ROWS = 128;

struct TrainArgs {
    float *w, *m, *v, *best_w, *img_fwd, *img_bwd, *grad;
    float *stash;  // hidden activations of the forward pass, [TRAIN_WAVES][B][2][L+1][NH][64] f32x4 (block_backward reads them back)
    const int *fwd_pos, *bwd_pos;
    int *adam_step;
    FlowShape s;
    const float *xtrain;
    int n_train;
    const float *xvalid;
    int n_valid;
    const int *perm;
    const float *noise;
    uint64_t seed;
    float jitter;
    int batch, max_epochs, patience;
    float lr, wd;
    float *losses;
    nnest_train_result_t *result;
    float *loss_out;
    int mode;
    int epoch_offset, flags;
    // train_kernel_grid (nnest_train_grid.h): global staging [B*2][CT][128][16], job results, parameter -> job slot, partial sums,
    // barrier counter, error word
    float *gstage, *gtile;
    int *gpos;
    float *gpart;
    float *gimgf, *gimgb;          // grid kernel: the published forward / backward fragment images (the owners' new weights)
    float *gdst;                   // grid kernel: (w, exp_avg, exp_avg_sq) of the parameters no job reaches, compact [3][ndead_pad]
    float *gown;                   // grid kernel, > 2 tiles per class: the job owners' parameter records [64 waves][64 lanes][32]
    int *gdead, *gndead;           // grid kernel: the parameters no job reaches (+ their count)
    unsigned int *gsync;
    int *gerr;
    const float *gz;  // VJP: upstream gradient [M, D]
    float *gx;        // VJP: gradient wrt the input rows [M, D]
    float gld_in;     // VJP: dL/d(logdet), the same for every row
};

// ---- fragment images -----------------------------------------------------------------------------------
// Which packed parameter (index into the state_dict-order vector) sits at element idx of the forward image
// (same layout as repack_fragments_kernel in nnest_kernels.hip); -1 = structural zero (padding / pruned).
__host__ __device__ inline int fwd_image_src(const FlowShape &s, int idx) {
    const int NT = s.NT, NH = s.NH, L = s.L, D = s.D, H = s.H;
    int bn = idx / s.net_floats, o = idx - bn * s.net_floats;
    int b = bn >> 1;
    const int base = bn * s.net_params;
    const int pc = (b + 1) & 1, pt = b & 1;
    const int pb0 = H * D, phid = H * D + H, pWo = H * D + H + L * (H * H + H), pbo = pWo + D * H;
    if (o < frag_off_L2(NT, NH)) {
        int r = o & 3, lane = (o >> 2) & 63, q = (o >> 8) << 2, tau = (q >> 2) % NT, ht = (q >> 2) / NT;
        int g = lane >> 4, i = lane & 15, d = 2 * (16 * tau + 4 * g + r) + pc;
        return d < D ? base + (16 * ht + i) * D + d : -1;
    } else if (o < frag_off_L3(NT, NH, L)) {
        int oo = o - frag_off_L2(NT, NH);
        int r = oo & 3, lane = (oo >> 2) & 63, q = (oo >> 8) << 2, hti = (q >> 2) % NH, hto = ((q >> 2) / NH) % NH, l = (q >> 2) / (NH * NH);
        int g = lane >> 4, i = lane & 15;
        return base + phid + l * (H * H + H) + (16 * hto + i) * H + 16 * hti + 4 * g + r;
    } else if (o < frag_off_b1(NT, NH, L)) {
        int oo = o - frag_off_L3(NT, NH, L);
        int r = oo & 3, lane = (oo >> 2) & 63, q = (oo >> 8) << 2, ht = (q >> 2) % NH, tau = (q >> 2) / NH;
        int g = lane >> 4, i = lane & 15, d = 2 * (16 * tau + i) + pt;
        return d < D ? base + pWo + d * H + 16 * ht + 4 * g + r : -1;
    } else if (o < frag_off_b2(NT, NH, L)) {
        return base + pb0 + (o - frag_off_b1(NT, NH, L));
    } else if (o < frag_off_b3(NT, NH, L)) {
        int oo = o - frag_off_b2(NT, NH, L), l = oo / (16 * NH), j = oo % (16 * NH);
        return base + phid + l * (H * H + H) + H * H + j;
    } else {
        int sl = o - frag_off_b3(NT, NH, L), d = 2 * sl + pt;
        return d < D ? base + pbo + d : -1;
    }
}

// backward image: transposed A-fragments, same region sizes as the forward image (no biases used)
//   region L1-sized  B3 [ht][tau][r][lane]   g_h[ht]   += Wout^T : lane(g,i) = Wout[dim(16tau+4g+r)][16ht+i]
//   region L2-sized  B2 [l][hti][hto][r][lane] g_h[hti] += W_l^T : lane(g,i) = W_l[16hto+4g+r][16hti+i]
//   region L3-sized  B1 [tau][ht][r][lane]   g_m[tau]  += W0^T   : lane(g,i) = W0[16ht+4g+r][dim(16tau+i)]
__host__ __device__ inline int bwd_image_src(const FlowShape &s, int idx) {
    const int NT = s.NT, NH = s.NH, L = s.L, D = s.D, H = s.H;
    int bn = idx / s.net_floats, o = idx - bn * s.net_floats;
    int b = bn >> 1;
    const int base = bn * s.net_params;
    const int pc = (b + 1) & 1, pt = b & 1;
    const int phid = H * D + H, pWo = H * D + H + L * (H * H + H);
    if (o < frag_off_L2(NT, NH)) {
        int r = o & 3, lane = (o >> 2) & 63, q = (o >> 8) << 2, tau = (q >> 2) % NT, ht = (q >> 2) / NT;
        int g = lane >> 4, i = lane & 15, d = 2 * (16 * tau + 4 * g + r) + pt;
        return d < D ? base + pWo + d * H + 16 * ht + i : -1;
    } else if (o < frag_off_L3(NT, NH, L)) {
        int oo = o - frag_off_L2(NT, NH);
        int r = oo & 3, lane = (oo >> 2) & 63, q = (oo >> 8) << 2, hto = (q >> 2) % NH, hti = ((q >> 2) / NH) % NH, l = (q >> 2) / (NH * NH);
        int g = lane >> 4, i = lane & 15;
        return base + phid + l * (H * H + H) + (16 * hto + 4 * g + r) * H + 16 * hti + i;
    } else if (o < frag_off_b1(NT, NH, L)) {
        int oo = o - frag_off_L3(NT, NH, L);
        int r = oo & 3, lane = (oo >> 2) & 63, q = (oo >> 8) << 2, ht = (q >> 2) % NH, tau = (q >> 2) / NH;
        int g = lane >> 4, i = lane & 15, d = 2 * (16 * tau + i) + pc;
        return d < D ? base + (16 * ht + 4 * g + r) * D + d : -1;
    }
    return -1;
}

// inverse maps packed parameter -> image element (each parameter occurs at most once per image; -1 = absent:
// exactly the parameters the alternating mask never reaches, whose gradient is identically zero)
__global__ void build_pos_kernel(int *__restrict__ fwd_pos, int *__restrict__ bwd_pos, FlowShape s) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < s.image_floats; i += gridDim.x * blockDim.x) {
        int f = fwd_image_src(s, i), b = bwd_image_src(s, i);
        if (f >= 0) fwd_pos[f] = i;
        if (b >= 0) bwd_pos[b] = i;
    }
}

hipError_t launch_build_pos(int *fwd_pos, int *bwd_pos, const FlowShape &s, hipStream_t st) {
    const size_t nb = (size_t)s.num_params() * sizeof(int);  // ScaleLayer scalars (scale='constant') have no image element
    hipError_t e = hipMemsetAsync(fwd_pos, 0xFF, nb, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(bwd_pos, 0xFF, nb, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(build_pos_kernel, dim3(64), dim3(256), 0, st, fwd_pos, bwd_pos, s);
    return hipGetLastError();
}

// (re)build images from the packed weights: into global memory, and into the LDS copies when given
__device__ __forceinline__ void rebuild_images_to(const TrainArgs &a, float *imgf, float *imgb) {
    for (int i = threadIdx.x; i < a.s.image_floats; i += blockDim.x) {
        int f = fwd_image_src(a.s, i), b = bwd_image_src(a.s, i);
        imgf[i] = f >= 0 ? a.w[f] : 0.f;
        imgb[i] = b >= 0 ? a.w[b] : 0.f;
    }
}

__device__ __forceinline__ void rebuild_images(const TrainArgs &a) {
    rebuild_images_to(a, a.img_fwd, a.img_bwd);
    if (a.s.scale_mode == 2 && threadIdx.x < a.s.B)  // the inference kernels read the ScaleLayer scalars behind the fragments
        a.img_fwd[a.s.image_floats + threadIdx.x] = a.w[a.s.nets_params() + threadIdx.x];
}

// ---- MLP forward keeping the hidden activations (for the backward pass) ---------------------------------
template <int NT, int NH, int L, int ACT>
__device__ __forceinline__ void mlp_fwd_keep(const float *__restrict__ wn, int lane, const f32x4 (&in)[NT],
                                             f32x4 (&acts)[L + 1][NH], f32x4 (&out)[NT]) {
    const int g4 = (lane >> 4) * 4;
    const float *fL1 = wn + frag_off_L1() + 4 * lane;
    const float *fL2 = wn + frag_off_L2(NT, NH) + 4 * lane;
    const float *fL3 = wn + frag_off_L3(NT, NH, L) + 4 * lane;
    const float *b1 = wn + frag_off_b1(NT, NH, L) + g4;
    const float *b2 = wn + frag_off_b2(NT, NH, L) + g4;
    const float *b3 = wn + frag_off_b3(NT, NH, L) + g4;
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(b1 + 16 * ht);
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            const f32x4 f = frag_quad(fL1, (ht * NT + tau));
            acc = mfma4(f.x, in[tau].x, acc);
            acc = mfma4(f.y, in[tau].y, acc);
            acc = mfma4(f.z, in[tau].z, acc);
            acc = mfma4(f.w, in[tau].w, acc);
        }
        acts[0][ht] = activate<ACT>(acc);
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
#pragma unroll
        for (int hto = 0; hto < NH; ++hto) {
            f32x4 acc = *reinterpret_cast<const f32x4 *>(b2 + (l * NH + hto) * 16);
#pragma unroll
            for (int hti = 0; hti < NH; ++hti) {
                const f32x4 f = frag_quad(fL2, ((l * NH + hto) * NH + hti));
                acc = mfma4(f.x, acts[l][hti].x, acc);
                acc = mfma4(f.y, acts[l][hti].y, acc);
                acc = mfma4(f.z, acts[l][hti].z, acc);
                acc = mfma4(f.w, acts[l][hti].w, acc);
            }
            acts[l + 1][hto] = activate<ACT>(acc);
        }
    }
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(b3 + 16 * tau);
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
            const f32x4 f = frag_quad(fL3, (tau * NH + ht));
            acc = mfma4(f.x, acts[L][ht].x, acc);
            acc = mfma4(f.y, acts[L][ht].y, acc);
            acc = mfma4(f.z, acts[L][ht].z, acc);
            acc = mfma4(f.w, acts[L][ht].w, acc);
        }
        out[tau] = acc;
    }
}

template <int ACT>
__device__ __forceinline__ f32x4 act_grad(f32x4 g, f32x4 a) {  // g * act'(pre), written with the post-activation a
    f32x4 o;
    if (ACT == 0) {  // tanh' = 1 - a^2
        o.x = g.x * (1.f - a.x * a.x); o.y = g.y * (1.f - a.y * a.y);
        o.z = g.z * (1.f - a.z * a.z); o.w = g.w * (1.f - a.w * a.w);
    } else {  // relu' = [a > 0]
        o.x = a.x > 0.f ? g.x : 0.f; o.y = a.y > 0.f ? g.y : 0.f;
        o.z = a.z > 0.f ? g.z : 0.f; o.w = a.w > 0.f ? g.w : 0.f;
    }
    return o;
}

// staging: column tile ct of the (block, net) being processed = [rows][16] floats
__device__ __forceinline__ void stage_tile(float *stg, int rows_pad, int ct, int row, int lane, f32x4 v) {
    // lane (g,w) holds features 4g..4g+3 of row w -> 16 contiguous bytes of that row
    *reinterpret_cast<f32x4 *>(stg + ((size_t)ct * rows_pad + row) * 16 + (lane >> 4) * 4) = v;
}

// column-tile map of the staging area for one net
template <int NT, int NH, int L> struct StageMap {
    static constexpr int gout(int tau) { return tau; }
    static constexpr int gpre(int l, int ht) { return NT + l * NH + ht; }
    static constexpr int act(int l, int ht) { return NT + (L + 1) * NH + l * NH + ht; }
    static constexpr int m(int tau) { return NT + 2 * (L + 1) * NH + tau; }
    static constexpr int count = 2 * NT + 2 * (L + 1) * NH;
};

// the last Linear of mlp_fwd_keep alone, from kept activations (same accumulation order)
template <int NT, int NH, int L>
__device__ __forceinline__ void mlp_out_layer(const float *__restrict__ wn, int lane, const f32x4 (&hl)[NH], f32x4 (&out)[NT]) {
    const int g4 = (lane >> 4) * 4;
    const float *fL3 = wn + frag_off_L3(NT, NH, L) + 4 * lane;
    const float *b3 = wn + frag_off_b3(NT, NH, L) + g4;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(b3 + 16 * tau);
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
            const f32x4 f = frag_quad(fL3, (tau * NH + ht));
            acc = mfma4(f.x, hl[ht].x, acc);
            acc = mfma4(f.y, hl[ht].y, acc);
            acc = mfma4(f.z, hl[ht].z, acc);
            acc = mfma4(f.w, hl[ht].w, acc);
        }
        out[tau] = acc;
    }
}

// One block of the training forward pass (CouplingLayer.forward, networks.py:289-298) that leaves the hidden activations of
// both nets in `stash` (this wave's slice for block b: [net][l][ht][64] f32x4, global memory / L2): the backward pass used
// to recompute both MLPs from the block input, two of whose three layers this saves.
template <int NT, int NH, int L>
__device__ __forceinline__ float block_forward_keep(const float *__restrict__ wf, int net_floats, bool affine, int lane,
                                                    const f32x4 (&cond)[NT], f32x4 (&trans)[NT], f32x4 *__restrict__ stash) {
    f32x4 as[L + 1][NH], at[L + 1][NH], ls[NT], t[NT];
    if (affine) mlp_fwd_keep<NT, NH, L, 0>(wf, lane, cond, as, ls);
    mlp_fwd_keep<NT, NH, L, 1>(wf + net_floats, lane, cond, at, t);
#pragma unroll
    for (int l = 0; l <= L; ++l)
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
            if (affine) stash[((0 * (L + 1) + l) * NH + ht) * 64 + lane] = as[l][ht];
            stash[((1 * (L + 1) + l) * NH + ht) * 64 + lane] = at[l][ht];
        }
    float ld = 0.f;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        if (affine) {  // inputs * exp(log_s) + t ; +sum(log_s)
            trans[tau].x = trans[tau].x * __expf(ls[tau].x) + t[tau].x;
            trans[tau].y = trans[tau].y * __expf(ls[tau].y) + t[tau].y;
            trans[tau].z = trans[tau].z * __expf(ls[tau].z) + t[tau].z;
            trans[tau].w = trans[tau].w * __expf(ls[tau].w) + t[tau].w;
            ld += (ls[tau].x + ls[tau].y) + (ls[tau].z + ls[tau].w);
        } else {       // translate-only coupling: log_s = 0
            trans[tau] = trans[tau] + t[tau];
        }
    }
    return ld;
}

// backward through one MLP; stages G tiles and activations; returns g_m (gradient wrt the conditioning inputs)
template <int NT, int NH, int L, int ACT>
__device__ __forceinline__ void mlp_bwd(const float *__restrict__ bn, int lane, float *stg, int rows_pad, int row,
                                        const f32x4 (&g_out)[NT], const f32x4 (&acts)[L + 1][NH], f32x4 (&g_m)[NT]) {
    typedef StageMap<NT, NH, L> SM;
    const float *B3 = bn + frag_off_L1() + 4 * lane;
    const float *B2 = bn + frag_off_L2(NT, NH) + 4 * lane;
    const float *B1 = bn + frag_off_L3(NT, NH, L) + 4 * lane;
    f32x4 gh[NH];
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) stage_tile(stg, rows_pad, SM::gout(tau), row, lane, g_out[tau]);
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            const f32x4 f = frag_quad(B3, (ht * NT + tau));
            acc = mfma4(f.x, g_out[tau].x, acc);
            acc = mfma4(f.y, g_out[tau].y, acc);
            acc = mfma4(f.z, g_out[tau].z, acc);
            acc = mfma4(f.w, g_out[tau].w, acc);
        }
        gh[ht] = acc;
    }
#pragma unroll
    for (int l = L; l >= 1; --l) {
        f32x4 gpre[NH];
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
            gpre[ht] = act_grad<ACT>(gh[ht], acts[l][ht]);
            stage_tile(stg, rows_pad, SM::gpre(l, ht), row, lane, gpre[ht]);
            stage_tile(stg, rows_pad, SM::act(l, ht), row, lane, acts[l][ht]);
        }
#pragma unroll
        for (int hti = 0; hti < NH; ++hti) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hto = 0; hto < NH; ++hto) {
                const f32x4 f = frag_quad(B2, (((l - 1) * NH + hti) * NH + hto));
                acc = mfma4(f.x, gpre[hto].x, acc);
                acc = mfma4(f.y, gpre[hto].y, acc);
                acc = mfma4(f.z, gpre[hto].z, acc);
                acc = mfma4(f.w, gpre[hto].w, acc);
            }
            gh[hti] = acc;
        }
    }
    f32x4 gpre0[NH];
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) {
        gpre0[ht] = act_grad<ACT>(gh[ht], acts[0][ht]);
        stage_tile(stg, rows_pad, SM::gpre(0, ht), row, lane, gpre0[ht]);
        stage_tile(stg, rows_pad, SM::act(0, ht), row, lane, acts[0][ht]);
    }
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
            const f32x4 f = frag_quad(B1, (tau * NH + ht));
            acc = mfma4(f.x, gpre0[ht].x, acc);
            acc = mfma4(f.y, gpre0[ht].y, acc);
            acc = mfma4(f.z, gpre0[ht].z, acc);
            acc = mfma4(f.w, gpre0[ht].w, acc);
        }
        g_m[tau] = acc;
    }
}

// ---- phase B: one dW tile = sum over rows of G[row][16] (x) Act[row][16] ----------------------------------
// WITH_BIAS: the bias gradient (column sums of G = the same product with Act = 1) rides on the G operands this job loads
// anyway, instead of being a job of its own: 5 jobs for 8 waves at the default shape rather than 9 (one wave used to run
// two jobs back to back while seven waited at the barrier).
template <bool WITH_BIAS>
__device__ __forceinline__ f32x4 contract_rows(const float *stg, int rows_pad, int ct_g, int ct_a, int lane, f32x4 &bias) {
    // rows_pad is a multiple of 16: four independent accumulators, one per 4-row k-step of a 16-row tile, so the
    // MFMAs issue back to back (a single chain waits 40 cycles per dependent v_mfma_f32_16x16x4_f32) and the LDS
    // reads of a tile are all in flight together
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 a0 = zero4, a1 = zero4, a2 = zero4, a3 = zero4;
    float bs = 0.f;  // WITH_BIAS: this lane's rows of column (lane & 15) of G, on the VALU (as MFMAs against ones the bias sums
                     // doubled the matrix-core time of the jobs that carry them, and those jobs set the length of the phase)
    const float *G = stg + (size_t)ct_g * rows_pad * 16 + (lane >> 4) * 16 + (lane & 15);
    const float *A = stg + (size_t)ct_a * rows_pad * 16 + (lane >> 4) * 16 + (lane & 15);
    for (int r = 0; r < rows_pad; r += 16) {
        const float g0 = G[(r + 0) * 16], g1 = G[(r + 4) * 16], g2 = G[(r + 8) * 16], g3 = G[(r + 12) * 16];
        a0 = mfma4(g0, A[(r + 0) * 16], a0);
        a1 = mfma4(g1, A[(r + 4) * 16], a1);
        a2 = mfma4(g2, A[(r + 8) * 16], a2);
        a3 = mfma4(g3, A[(r + 12) * 16], a3);
        if (WITH_BIAS) bs += (g0 + g1) + (g2 + g3);
    }
    if (WITH_BIAS) {
        bs += __shfl_xor(bs, 16);
        bs += __shfl_xor(bs, 32);  // every lane: the sum over all rows of column (lane & 15)
        const int q4 = (lane >> 4) * 4;  // lane (gq, j) reg r  <->  out feature 4*gq + r, as the product against ones gave it
        bias = (f32x4){__shfl(bs, q4 + 0), __shfl(bs, q4 + 1), __shfl(bs, q4 + 2), __shfl(bs, q4 + 3)};
    }
    return (a0 + a1) + (a2 + a3);  // lane (gq, j) reg r  <->  (out feature 4*gq + r, in feature j)
}

// per-minibatch Adam scalars (torch/optim/adam.py _single_tensor_adam)
struct AdamStep {
    float step_size, inv_bc2s;
};

// torch.optim.Adam with coupled weight decay (trainer.py:121-122) over the whole packed vector, float4-coalesced,
// every load of a thread's slice issued before its arithmetic.  Parameters the mask never reaches have an exactly
// zero gradient (the gradient buffer is zeroed once and never written there) but still take their step, as in the
// reference.  Each updated weight is scattered to its element of the forward / backward fragment image (position
// maps built once per flow), which replaces rebuilding both images from the packed vector every minibatch.
__device__ __forceinline__ void adam_one(const TrainArgs &a, const AdamStep &ad, float &w, float g, float &m, float &v, int fp,
                                         int bp, float *imgf, float *imgb) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    float gi = g + a.wd * w;
    m = m + (gi - m) * (1.0f - b1);
    v = v * b2 + (1.0f - b2) * gi * gi;
    // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the correctly-rounded sequences: the update is lr * m / denom with
    // lr = 1e-3, so an ulp of the ratio is ~1e-10 on the weight
    float denom = __builtin_amdgcn_sqrtf(v) * ad.inv_bc2s + eps;
    w = w - ad.step_size * (m * __builtin_amdgcn_rcpf(denom));
    if (fp >= 0) imgf[fp] = w;
    if (bp >= 0) imgb[bp] = w;
}

__device__ __forceinline__ void adam_sweep(const TrainArgs &a, const AdamStep &ad, int np, float *imgf, float *imgb) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    constexpr int U = 4;  // float4 groups in flight per thread: the global-load latency is paid once per U groups
    const int n4 = np >> 2;
    for (int i0 = threadIdx.x; i0 < n4; i0 += U * blockDim.x) {
        f32x4 w4[U], g4[U], m4[U], v4[U];
        i32x4 fp[U], bp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < n4) {
                w4[u] = reinterpret_cast<const f32x4 *>(a.w)[i];
                g4[u] = reinterpret_cast<const f32x4 *>(a.grad)[i];
                m4[u] = reinterpret_cast<const f32x4 *>(a.m)[i];
                v4[u] = reinterpret_cast<const f32x4 *>(a.v)[i];
                fp[u] = reinterpret_cast<const i32x4 *>(a.fwd_pos)[i];
                bp[u] = reinterpret_cast<const i32x4 *>(a.bwd_pos)[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < n4) {
                float w[4] = {w4[u].x, w4[u].y, w4[u].z, w4[u].w}, g[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w};
                float m[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w}, v[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
                const int f[4] = {fp[u].x, fp[u].y, fp[u].z, fp[u].w}, bq[4] = {bp[u].x, bp[u].y, bp[u].z, bp[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) adam_one(a, ad, w[k], g[k], m[k], v[k], f[k], bq[k], imgf, imgb);
                reinterpret_cast<f32x4 *>(a.w)[i] = (f32x4){w[0], w[1], w[2], w[3]};
                reinterpret_cast<f32x4 *>(a.m)[i] = (f32x4){m[0], m[1], m[2], m[3]};
                reinterpret_cast<f32x4 *>(a.v)[i] = (f32x4){v[0], v[1], v[2], v[3]};
            }
        }
    }
    for (int p = 4 * n4 + threadIdx.x; p < np; p += blockDim.x) {  // tail (np not a multiple of 4)
        float w = a.w[p], m = a.m[p], v = a.v[p];
        adam_one(a, ad, w, a.grad[p], m, v, a.fwd_pos[p], a.bwd_pos[p], imgf, imgb);
        a.w[p] = w; a.m[p] = m; a.v[p] = v;
    }
}

template <int NT, int NH, int L>
__device__ __forceinline__ void weight_grad_jobs(const TrainArgs &a, const float *stg, int rows_pad, int b, int net,
                                                 int wave, int lane, const AdamStep &ad, float *imgf, float *imgb) {
    typedef StageMap<NT, NH, L> SM;
    const int D = a.s.D, H = a.s.H;
    const int pc = (b + 1) & 1, pt = b & 1;
    const int pb0 = H * D, phid = H * D + H, pWo = H * D + H + L * (H * H + H), pbo = pWo + D * H;
    const int pbase = (b * 2 + net) * a.s.net_params;
    // every gradient element has exactly one producer (no atomics): scattered into the packed gradient vector
    (void)ad; (void)imgf; (void)imgb;
    auto emit = [&](int idx, float g) { a.grad[pbase + idx] = g; };
    const int gq = lane >> 4, j = lane & 15;
    constexpr int J_W3 = NT * NH, J_W2 = L * NH * NH, J_W1 = NH * NT;
    constexpr int NJOBS = J_W3 + J_W2 + J_W1;
    f32x4 bt = {0.f, 0.f, 0.f, 0.f};
    for (int job = wave; job < NJOBS; job += TRAIN_WAVES) {
        int q = job;
        if (q < J_W3) {  // dWout[slot][hidden]  (+ dbout with the first hidden tile)
            int tau = q / NH, ht = q % NH;
            f32x4 t = ht == 0 ? contract_rows<true>(stg, rows_pad, SM::gout(tau), SM::act(L, ht), lane, bt)
                              : contract_rows<false>(stg, rows_pad, SM::gout(tau), SM::act(L, ht), lane, bt);
            float v[4] = {t.x, t.y, t.z, t.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int d = 2 * (16 * tau + 4 * gq + r) + pt;
                if (d < D) {
                    emit(pWo + d * H + 16 * ht + j, v[r]);
                    if (ht == 0 && j == 0) emit(pbo + d, bv[r]);
                }
            }
            continue;
        }
        q -= J_W3;
        if (q < J_W2) {  // hidden layer l = 1..L : dW_l[out][in]  (+ db_l with the first input tile)
            int l = q / (NH * NH) + 1, hto = (q / NH) % NH, hti = q % NH;
            f32x4 t = hti == 0 ? contract_rows<true>(stg, rows_pad, SM::gpre(l, hto), SM::act(l - 1, hti), lane, bt)
                               : contract_rows<false>(stg, rows_pad, SM::gpre(l, hto), SM::act(l - 1, hti), lane, bt);
            float v[4] = {t.x, t.y, t.z, t.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                emit(phid + (l - 1) * (H * H + H) + (16 * hto + 4 * gq + r) * H + 16 * hti + j, v[r]);
                if (hti == 0 && j == 0) emit(phid + (l - 1) * (H * H + H) + H * H + 16 * hto + 4 * gq + r, bv[r]);
            }
            continue;
        }
        q -= J_W2;
        {  // dW0[hidden][dim]  (+ db0 with the first slot tile)
            int ht = q / NT, tau = q % NT;
            f32x4 t = tau == 0 ? contract_rows<true>(stg, rows_pad, SM::gpre(0, ht), SM::m(tau), lane, bt)
                               : contract_rows<false>(stg, rows_pad, SM::gpre(0, ht), SM::m(tau), lane, bt);
            float v[4] = {t.x, t.y, t.z, t.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
            int d = 2 * (16 * tau + j) + pc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (d < D) emit((16 * ht + 4 * gq + r) * D + d, v[r]);
                if (tau == 0 && j == 0) emit(pb0 + 16 * ht + 4 * gq + r, bv[r]);
            }
        }
    }
}

// ---- one coupling block of the backward pass (on a row tile) -----------------------------------------------
// On entry: cond / ytrans = the block's OUTPUT halves, gcond / gtrans = dLoss/d(output halves), gld = dLoss/dlogdet.
// On exit : ytrans = the block's input (recovered through the inverse), gcond / gtrans = dLoss/d(input halves).
// Staging + weight-gradient jobs for both nets happen inside (4 workgroup barriers).
template <int NT, int NH, int L>
__device__ __forceinline__ void block_backward(const TrainArgs &a, float *stg, int rows_pad, int b, int wave, int lane,
                                               bool tile_active, int row, bool row_ok, const f32x4 (&cond)[NT],
                                               f32x4 (&ytrans)[NT], f32x4 (&gcond)[NT], f32x4 (&gtrans)[NT], float gld,
                                               const AdamStep &ad, float *imgf, float *imgb, unsigned long long (&ph)[8],
                                               const f32x4 *__restrict__ stash) {
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0;
    (void)q0; (void)q1; (void)q2; (void)q3; (void)q4; (void)ph;
    TSTAMP(q0);
    typedef StageMap<NT, NH, L> SM;
    const int pt = b & 1;
    const float *wf = imgf + (size_t)b * 2 * a.s.net_floats;
    const float *wb = imgb + (size_t)b * 2 * a.s.net_floats;
    f32x4 as[L + 1][NH], at[L + 1][NH], ls[NT], t[NT], g_ls[NT], g_t[NT], gm_s[NT], gm_t[NT];
    const int g = lane >> 4;
    // translate-only couplings (scale='translate' / 'constant', networks.py:293-294): no scale net, log_s = 0
    const bool affine = a.s.scale_mode == 0;
    if (tile_active) {
        // the hidden activations come back from the forward pass's stash; only the output layers are evaluated again
#pragma unroll
        for (int l = 0; l <= L; ++l)
#pragma unroll
            for (int ht = 0; ht < NH; ++ht) {
                if (affine) as[l][ht] = stash[((0 * (L + 1) + l) * NH + ht) * 64 + lane];
                at[l][ht] = stash[((1 * (L + 1) + l) * NH + ht) * 64 + lane];
            }
        if (affine) mlp_out_layer<NT, NH, L>(wf, lane, as[L], ls);
        else {
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) ls[tau] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        mlp_out_layer<NT, NH, L>(wf + a.s.net_floats, lane, at[L], t);
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            float lsv[4] = {ls[tau].x, ls[tau].y, ls[tau].z, ls[tau].w};
            float tv[4] = {t[tau].x, t[tau].y, t[tau].z, t[tau].w};
            float yv[4] = {ytrans[tau].x, ytrans[tau].y, ytrans[tau].z, ytrans[tau].w};
            float gv[4] = {gtrans[tau].x, gtrans[tau].y, gtrans[tau].z, gtrans[tau].w};
            float o_gls[4], o_gt[4], o_x[4], o_gx[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 2 * (16 * tau + 4 * g + r) + pt;
                const bool valid = row_ok && d < a.s.D;
                float ymt = yv[r] - tv[r];                 // = x_in * e^{ls}
                o_gls[r] = valid ? gv[r] * ymt + gld : 0.f;  // y = x e^{ls} + t ; logdet += ls
                o_gt[r] = valid ? gv[r] : 0.f;
                o_x[r] = ymt * __expf(-lsv[r]);            // block input (networks.py:307-309)
                o_gx[r] = gv[r] * __expf(lsv[r]);          // direct path dy/dx
            }
            g_ls[tau] = (f32x4){o_gls[0], o_gls[1], o_gls[2], o_gls[3]};
            g_t[tau] = (f32x4){o_gt[0], o_gt[1], o_gt[2], o_gt[3]};
            ytrans[tau] = (f32x4){o_x[0], o_x[1], o_x[2], o_x[3]};
            gtrans[tau] = (f32x4){o_gx[0], o_gx[1], o_gx[2], o_gx[3]};
        }
        // scale net: stage G, activations and the shared conditioning input m
        if (affine) mlp_bwd<NT, NH, L, 0>(wb, lane, stg, rows_pad, row, g_ls, as, gm_s);
        else {
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) gm_s[tau] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            f32x4 mv = cond[tau];
            if (!row_ok) mv = (f32x4){0.f, 0.f, 0.f, 0.f};
            stage_tile(stg, rows_pad, SM::m(tau), row, lane, mv);
        }
    }
    if (affine) {
        __syncthreads();
        TSTAMP(q1);
        weight_grad_jobs<NT, NH, L>(a, stg, rows_pad, b, 0, wave, lane, ad, imgf, imgb);
        __syncthreads();
    }
    TSTAMP(q2);
    if (tile_active) mlp_bwd<NT, NH, L, 1>(wb + a.s.net_floats, lane, stg, rows_pad, row, g_t, at, gm_t);
    __syncthreads();
    TSTAMP(q3);
    weight_grad_jobs<NT, NH, L>(a, stg, rows_pad, b, 1, wave, lane, ad, imgf, imgb);
    __syncthreads();
    TSTAMP(q4);
    TACC(ph[1], q1, q0); TACC(ph[2], q2, q1); TACC(ph[3], q3, q2); TACC(ph[4], q4, q3);
    if (tile_active) {
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) gcond[tau] = gcond[tau] + gm_s[tau] + gm_t[tau];  // masked_inputs = inputs * mask
    }
}

// ---- the kernel ------------------------------------------------------------------------------------------
// IMGLDS: the forward and backward fragment images live in LDS next to the staging area (config 2: 2 x 32 KB +
// 64 KB); otherwise they are read from / updated in global memory (L2).
// (IMGLDS = 1: only the forward image fits beside the staging area -- 3-4 tiles per class; the transposed image stays in L2)
template <int NT, int NH, int L, int IMGLDS>
__global__ void __launch_bounds__(TRAIN_THREADS) train_kernel(TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *imgf = IMGLDS >= 1 ? smem : a.img_fwd;
    float *imgb = IMGLDS == 2 ? smem + a.s.image_floats : a.img_bwd;
    float *stg = smem + IMGLDS * a.s.image_floats;
    __shared__ float red[TRAIN_WAVES];
    __shared__ int ctl[4];      // [0] stop flag, [1] counter, [2] best epoch
    __shared__ float ctlf[2];   // [0] best validation loss
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = lane & 15, g = lane >> 4;
    const int D = a.s.D, B = a.s.B;
    const int np = a.s.num_params();
    // ScaleLayer scalars (scale='constant'): read from the packed vector, which Adam updates in place
    const float *blk_scale = a.s.scale_mode == 2 ? a.w + a.s.nets_params() : nullptr;
    f32x4 *stash_w = reinterpret_cast<f32x4 *>(a.stash) + (size_t)wave * B * 2 * (L + 1) * NH * 64;
    __shared__ float sred[8 * TRAIN_WAVES];  // per (block, wave) partials of dLoss/ds_b; B <= 8 checked by the launcher

    rebuild_images_to(a, imgf, imgb);
    if (a.mode != TRAIN_MODE_EPOCHS)
        for (int i = threadIdx.x; i < np; i += blockDim.x) a.grad[i] = 0.f;
    const bool resume = a.mode == TRAIN_MODE_EPOCHS && (a.flags & NNEST_TRAIN_RESUME);
    if (threadIdx.x == 0) {
        ctl[0] = 0;
        ctl[1] = resume ? a.result->counter : 0;
        ctl[2] = resume ? a.result->best_epoch : 0;
        ctlf[0] = resume ? a.result->best_validation_loss : INFINITY;
    }
    if (a.mode == TRAIN_MODE_EPOCHS)
        for (int i = threadIdx.x; i < np; i += blockDim.x) {
            if (!resume) a.best_w[i] = a.w[i];  // best_model = deepcopy(netG)  trainer.py:194
            a.grad[i] = 0.f;
        }
    __syncthreads();

    const int n_mb = a.mode != TRAIN_MODE_EPOCHS ? 1 : (a.n_train + a.batch - 1) / a.batch;
    const int n_epochs = a.mode != TRAIN_MODE_EPOCHS ? 1 : a.max_epochs;
    int adam_t = a.adam_step ? *a.adam_step : 0;
    int epochs_run = 0;
    float last_train_loss = 0.f;

    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    (void)p0; (void)p1; (void)p2; (void)p3;
    for (int epoch = 0; epoch < n_epochs; ++epoch) {
        float epoch_loss = 0.f;  // sum of minibatch means (trainer.py:398)
        for (int mb = 0; mb < n_mb; ++mb) {
            const int M = a.mode != TRAIN_MODE_EPOCHS ? a.n_train : min(a.batch, a.n_train - mb * a.batch);
            const int ntile = (M + 15) >> 4;
            const int rows_pad = ntile * 16;
            const bool tile_active = wave < ntile;
            const int row = wave * 16 + w;  // row inside the minibatch
            const bool row_ok = tile_active && row < M;
            f32x4 xs[2][NT], gs[2][NT];
            float ld = 0.f;
            TSTAMP(p0);
            if (tile_active) {
                // data = X[perm] + jitter * randn  (trainer.py:392)
                long src = 0;
                if (row_ok) src = a.mode != TRAIN_MODE_EPOCHS ? row : a.perm[(size_t)epoch * a.n_train + mb * a.batch + row];
                load_tile<NT>(a.xtrain, src, row_ok, D, lane, xs);
                if (a.mode == TRAIN_MODE_EPOCHS && a.jitter != 0.f) {
                    const long p = (long)mb * a.batch + row;
                    if (a.noise) {
                        f32x4 nz[2][NT];
                        load_tile<NT>(a.noise + (size_t)epoch * a.n_train * D, p, row_ok, D, lane, nz);
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int t = 0; t < NT; ++t) xs[c][t] = xs[c][t] + nz[c][t] * a.jitter;
                    } else {
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            f32x4 n0 = noise_normal4(a.seed, (uint64_t)p, (uint32_t)(a.epoch_offset + epoch), (uint32_t)(8 * t + 2 * g), NOISE_STREAM_JITTER);
                            f32x4 n1 = noise_normal4(a.seed, (uint64_t)p, (uint32_t)(a.epoch_offset + epoch), (uint32_t)(8 * t + 2 * g + 1), NOISE_STREAM_JITTER);
                            const int d0 = 32 * t + 8 * g;
                            if (row_ok) {
                                if (d0 + 0 < D) xs[0][t].x += n0.x * a.jitter; if (d0 + 1 < D) xs[1][t].x += n0.y * a.jitter;
                                if (d0 + 2 < D) xs[0][t].y += n0.z * a.jitter; if (d0 + 3 < D) xs[1][t].y += n0.w * a.jitter;
                                if (d0 + 4 < D) xs[0][t].z += n1.x * a.jitter; if (d0 + 5 < D) xs[1][t].z += n1.y * a.jitter;
                                if (d0 + 6 < D) xs[0][t].w += n1.z * a.jitter; if (d0 + 7 < D) xs[1][t].w += n1.w * a.jitter;
                            }
                        }
                    }
                }
                const bool affine = a.s.scale_mode == 0;
                TSTAMP(p3);
                TACC(ph[6], p3, p0);  // (diagnostic build: rows + jitter; the 'rebuild' slot is free)
                float ldp = 0.f;
                for (int b = 0; b < B; ++b) {
                    const float *wf = imgf + (size_t)b * 2 * a.s.net_floats;
                    f32x4 *sb = stash_w + (size_t)b * 2 * (L + 1) * NH * 64;
                    if (b & 1) ldp += block_forward_keep<NT, NH, L>(wf, a.s.net_floats, affine, lane, xs[0], xs[1], sb);
                    else       ldp += block_forward_keep<NT, NH, L>(wf, a.s.net_floats, affine, lane, xs[1], xs[0], sb);
                    if (blk_scale) ldp += scale_layer_tile<NT>(blk_scale[b], false, lane, xs);
                }
                ld = group_sum(ldp);
            }
            // loss = -mean(log_probs)  (trainer.py:394; networks.py:71-76)
            float lp = 0.f;
            if (tile_active) {
                float ss = 0.f;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int t = 0; t < NT; ++t) ss += base_E4(xs[c][t], a.s.base_beta);
                ss = group_sum(ss);
                lp = (row_ok && g == 0) ? (-ss + a.s.base_const * (float)D + ld) : 0.f;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) lp += __shfl_xor(lp, o);
            }
            if (lane == 0) red[wave] = tile_active ? lp : 0.f;
            __syncthreads();
            float loss = 0.f;
            for (int k = 0; k < TRAIN_WAVES; ++k) loss += red[k];
            loss = -loss / (float)M;
            epoch_loss += loss;
            // d(loss)/du = dE/du / M (= u/M for the N(0,I) base) ; d(loss)/d(logdet) = -1/M
            TSTAMP(p1);
            TACC(ph[0], p1, p0);
            const float invM = 1.0f / (float)M, gld = a.mode == TRAIN_MODE_VJP ? a.gld_in : -invM;
            if (a.mode == TRAIN_MODE_VJP) {
                if (tile_active) load_tile<NT>(a.gz, row, row_ok, D, lane, gs);
            } else {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int t = 0; t < NT; ++t) gs[c][t] = row_ok ? base_dE4(xs[c][t], a.s.base_beta) * invM : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            // Adam scalars of this step (torch/optim/adam.py _single_tensor_adam); the update itself is applied
            // per parameter by the thread that produces its gradient (weight_grad_jobs)
            AdamStep ad = {0.f, 1.f};
            if (a.mode == TRAIN_MODE_EPOCHS) {
                adam_t += 1;
                const double bc1 = 1.0 - pow(0.9, (double)adam_t), bc2 = 1.0 - pow(0.999, (double)adam_t);
                ad.step_size = (float)((double)a.lr / bc1);
                ad.inv_bc2s = (float)(1.0 / sqrt(bc2));
            }
            for (int b = B - 1; b >= 0; --b) {
                if (blk_scale) {
                    // ScaleLayer b (networks.py:318-320): y = c e^s, logdet += s.  dLoss/ds = sum_rows (gy . y) + M gld;
                    // then c = y e^{-s}, gc = gy e^s.
                    const float sb = blk_scale[b], e = __expf(sb), ei = __expf(-sb);
                    float part = 0.f;
                    if (tile_active) {
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int t = 0; t < NT; ++t) {
                                const f32x4 y = xs[c][t], gy = gs[c][t];
                                part += (gy.x * y.x + gy.y * y.y) + (gy.z * y.z + gy.w * y.w);
                                xs[c][t] = y * ei;
                                gs[c][t] = gy * e;
                            }
                        part = group_sum(part);
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) part += __shfl_xor(part, o);
                    }
                    if (lane == 0) sred[b * TRAIN_WAVES + wave] = tile_active ? part : 0.f;
                }
                if (b & 1) block_backward<NT, NH, L>(a, stg, rows_pad, b, wave, lane, tile_active, row, row_ok, xs[0], xs[1], gs[0], gs[1], gld, ad, imgf, imgb, ph, stash_w + (size_t)b * 2 * (L + 1) * NH * 64);
                else       block_backward<NT, NH, L>(a, stg, rows_pad, b, wave, lane, tile_active, row, row_ok, xs[1], xs[0], gs[1], gs[0], gld, ad, imgf, imgb, ph, stash_w + (size_t)b * 2 * (L + 1) * NH * 64);
            }
            if (blk_scale) {
                __syncthreads();
                if (threadIdx.x < B) {
                    float gsum = 0.f;
                    for (int k = 0; k < TRAIN_WAVES; ++k) gsum += sred[threadIdx.x * TRAIN_WAVES + k];
                    a.grad[a.s.nets_params() + threadIdx.x] = gsum + (float)M * gld;
                }
                __syncthreads();
            }
            if (a.mode == TRAIN_MODE_GRAD_ONLY) {
                if (threadIdx.x == 0) *a.loss_out = loss;
                __syncthreads();
                return;
            }
            if (a.mode == TRAIN_MODE_VJP) {
                if (tile_active) store_tile<NT>(a.gx, row, row_ok, D, lane, gs);
                __syncthreads();
                return;
            }
            TSTAMP(p1);
            adam_sweep(a, ad, np, imgf, imgb);
            __syncthreads();
            TSTAMP(p2);
            TACC(ph[5], p2, p1);
        }
        // ---- Trainer._validate (trainer.py:405-418): one full batch, loss / len(valid) -----------------------
        TSTAMP(p0);
        float vsum = 0.f;
        {
            const int vtiles = (a.n_valid + 15) >> 4;
            for (int tile = wave; tile < vtiles; tile += TRAIN_WAVES) {
                const int r = tile * 16 + w;
                const bool ok = r < a.n_valid;
                f32x4 xv[2][NT];
                load_tile<NT>(a.xvalid, r, ok, D, lane, xv);
                float ldv = group_sum(flow_forward_tile<NT, NH>(imgf, a.s.net_floats, B, L, lane, xv, blk_scale));
                float ss = 0.f;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int t = 0; t < NT; ++t) ss += base_E4(xv[c][t], a.s.base_beta);
                ss = group_sum(ss);
                float lpv = (ok && g == 0) ? (-ss + a.s.base_const * (float)D + ldv) : 0.f;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) lpv += __shfl_xor(lpv, o);
                vsum += lpv;
            }
        }
        __syncthreads();
        if (lane == 0) red[wave] = vsum;
        __syncthreads();
        float vtot = 0.f;
        for (int k = 0; k < TRAIN_WAVES; ++k) vtot += red[k];
        TSTAMP(p1);
        TACC(ph[7], p1, p0);
        const float valid_loss = (-vtot / (float)a.n_valid) / (float)a.n_valid;  // mean, then / len(dataset)  :418
        const float train_loss = epoch_loss / (float)a.n_train;                   // trainer.py:403
        last_train_loss = train_loss;
        epochs_run = epoch + 1;
        if (a.losses && threadIdx.x == 0) {
            a.losses[2 * epoch] = train_loss;
            a.losses[2 * epoch + 1] = valid_loss;
        }
        // early stopping bookkeeping (trainer.py:205-209, :223-232); every thread evaluates the same values
        const bool improved = valid_loss < ctlf[0];
        __syncthreads();
        if (improved) {
            for (int i = threadIdx.x; i < np; i += blockDim.x) a.best_w[i] = a.w[i];
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
    // netG.load_state_dict(best_model)  (trainer.py:241): at the end of train() or when patience ran out
    __syncthreads();
    const bool stopped = ctl[0] != 0;
    if (stopped || (a.flags & NNEST_TRAIN_FINALIZE)) {
        for (int i = threadIdx.x; i < np; i += blockDim.x) a.w[i] = a.best_w[i];
        __syncthreads();
    }
    rebuild_images(a);  // the global images (read by the inference kernels and the next chunk) follow a.w
#ifdef NNEST_STAMP
    if (threadIdx.x == 0 && a.losses)
        for (int i = 0; i < 8; ++i) a.losses[i] = (float)ph[i];
#endif
    if (threadIdx.x == 0) {
        if (a.adam_step) *a.adam_step = adam_t;
        a.result->epochs_run = a.epoch_offset + epochs_run;
        a.result->best_epoch = ctl[2];
        a.result->best_validation_loss = ctlf[0];
        a.result->last_train_loss = last_train_loss;
        a.result->counter = ctl[1];
        a.result->stopped = stopped ? 1 : 0;
    }
}

// ---- training jitter (trainer.py:168-171): 0.2 * mean over both columns of cKDTree(samples).query(samples, 2)
// = 0.2 * sum_i nn_dist(i) / (2N).  Brute force in float64: N is the live-point count (<= ~1e4).
// One wave per row i (lane l scans j = l, l + 64, ...), four rows per workgroup: N / 4 workgroups, each thread's serial scan N / 64
// rows long (the first version, one thread per row, took 6.3 ms at N = 1000, D = 50; 16 lanes per row 60 us -- which a config-2
// run waits for 392 times; this one ~20 us).  The minimum over j does not depend on the order it is taken in.
__global__ void __launch_bounds__(256) nn_distance_kernel(const double *__restrict__ X, int N, int D, double *__restrict__ out) {
    extern __shared__ double xrow[];  // [4][D]
    __shared__ double red[4];
    const int l = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + r;
    for (int k = threadIdx.x; k < 4 * D; k += 256) {
        const int row = blockIdx.x * 4 + k / D;
        xrow[k] = row < N ? X[(size_t)row * D + k % D] : 0.0;
    }
    __syncthreads();
    const double *xi = xrow + r * D;
    double best = INFINITY;
    if (i < N)
        for (int j = l; j < N; j += 64) {
            if (j == i) continue;
            const double *xj = X + (size_t)j * D;
            double s = 0.0;
            for (int d = 0; d < D; ++d) {
                double e = xi[d] - xj[d];
                s += e * e;
            }
            best = s < best ? s : best;
        }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double other = __shfl_xor(best, o);
        best = other < best ? other : best;
    }
    if (l == 0) red[r] = i < N ? sqrt(best) : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < 4; ++k) tot += red[k];
        atomicAdd(out, tot * 0.2 / (2.0 * N));
    }
}

hipError_t launch_training_jitter(const double *samples, int N, int D, double *out, hipStream_t st) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st);
    if (e != hipSuccess) return e;
    int grid = (N + 3) / 4;
    hipLaunchKernelGGL(nn_distance_kernel, dim3(grid), dim3(256), (size_t)4 * D * sizeof(double), st, samples, N, D, out);
    return hipGetLastError();
}

// ---- launchers --------------------------------------------------------------------------------------------
// workspace layout (floats): [img_bwd: image_floats][grad: num_params + 64][stash]
static size_t train_stash_floats(const FlowShape &s) { return (size_t)TRAIN_WAVES * s.B * 2 * (s.L + 1) * s.NH * 256; }
static size_t grid_workspace_floats(const FlowShape &s);
static size_t single_workspace_floats(const FlowShape &s) { return (size_t)s.image_floats + (size_t)s.num_params() + 64 + train_stash_floats(s); }
size_t train_workspace_floats(const FlowShape &s, int batch) {
    (void)batch;
    return single_workspace_floats(s) + 64 + grid_workspace_floats(s);
}

template <int NT, int NH, int L, int IMGLDS>
static hipError_t launch_train_tt(const TrainArgs &a, size_t lds, hipStream_t st) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(train_kernel<NT, NH, L, IMGLDS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((train_kernel<NT, NH, L, IMGLDS>), dim3(1), dim3(TRAIN_THREADS), lds, st, a);
    return hipGetLastError();
}

template <int NT, int NH, int L>
static hipError_t launch_train_t(const TrainArgs &a, hipStream_t st) {
    typedef StageMap<NT, NH, L> SM;
    const size_t stage = (size_t)SM::count * TRAIN_MAX_ROWS * 16 * sizeof(float);
    const size_t with_img = stage + 2 * (size_t)a.s.image_floats * sizeof(float);
    const size_t with_fwd = stage + (size_t)a.s.image_floats * sizeof(float);
    if (with_img <= 160 * 1024 - 256) return launch_train_tt<NT, NH, L, 2>(a, with_img, st);
    if (with_fwd <= 160 * 1024 - 256) return launch_train_tt<NT, NH, L, 1>(a, with_fwd, st);
    return launch_train_tt<NT, NH, L, 0>(a, stage, st);
}

static hipError_t dispatch_train(const TrainArgs &a, hipStream_t st) {
    const FlowShape &s = a.s;
#define TRY_SHAPE(nt, nh, l) if (s.NT == nt && s.NH == nh && s.L == l) return launch_train_t<nt, nh, l>(a, st)
    TRY_SHAPE(1, 1, 0); TRY_SHAPE(2, 1, 0); TRY_SHAPE(3, 1, 0); TRY_SHAPE(4, 1, 0);
    TRY_SHAPE(1, 1, 1); TRY_SHAPE(2, 1, 1); TRY_SHAPE(3, 1, 1); TRY_SHAPE(4, 1, 1);
    TRY_SHAPE(1, 1, 2); TRY_SHAPE(2, 1, 2); TRY_SHAPE(3, 1, 2); TRY_SHAPE(4, 1, 2);
    TRY_SHAPE(1, 2, 1); TRY_SHAPE(2, 2, 1); TRY_SHAPE(1, 2, 2); TRY_SHAPE(2, 2, 2);
    TRY_SHAPE(1, 4, 1);
    // round 6: the rest of what the reference's Trainer(hidden_dim, num_layers) can ask for inside the instantiated tile shapes
    // (nnest/networks.py:253-287 takes any num_layers): three hidden layers, and the wide nets without / with two of them
    TRY_SHAPE(1, 1, 3); TRY_SHAPE(2, 1, 3); TRY_SHAPE(3, 1, 3); TRY_SHAPE(4, 1, 3);
    TRY_SHAPE(1, 2, 0); TRY_SHAPE(2, 2, 0); TRY_SHAPE(1, 2, 3); TRY_SHAPE(2, 2, 3);
    TRY_SHAPE(1, 4, 0); TRY_SHAPE(1, 4, 2); TRY_SHAPE(1, 4, 3);
#undef TRY_SHAPE
    return hipErrorInvalidConfiguration;
}

#include "nnest_train_grid.h"
#include "nnest_train_rows.h"
#include "nnest_train_pipe.h"
// torch.optim.Adam (coupled weight decay), one element of the step, every operation rounded by itself (no multiply-add contraction: the choice of which product a compiler
// fuses differs from kernel to kernel, and the kernels that share this function must agree bit for bit -- maf_update_kernel steps
// the same parameters inside another loop)
#pragma clang fp contract(off)
__device__ __forceinline__ float adam_elem(float wi, float g, float &mref, float &vref, float step_size, float inv_bc2s, float wd) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const float gi = g + wd * wi;
    const float mi = mref + (gi - mref) * (1.0f - b1);
    const float vi = vref * b2 + ((1.0f - b2) * gi) * gi;
    mref = mi; vref = vi;
    return wi - step_size * (mi / (sqrtf(vi) * inv_bc2s + eps));
}
#pragma clang fp contract(fast)

#include "maf_train.h"

hipError_t launch_loss_grad(const float *packed, const FlowShape &s, const float *x, int M, float *grad, float *loss,
                            float *workspace, float *img_fwd, const int *fwd_pos, const int *bwd_pos, hipStream_t st) {
    TrainArgs a;
    memset(&a, 0, sizeof(a));
    a.w = const_cast<float *>(packed);
    a.img_fwd = img_fwd;
    a.img_bwd = workspace;
    a.stash = workspace + s.image_floats + s.num_params() + 64;
    a.fwd_pos = fwd_pos;
    a.bwd_pos = bwd_pos;
    a.grad = grad;
    a.s = s;
    a.xtrain = x;
    a.n_train = M;
    a.batch = M;
    a.loss_out = loss;
    a.mode = TRAIN_MODE_GRAD_ONLY;
    return dispatch_train(a, st);
}

hipError_t launch_vjp(const float *packed, const FlowShape &s, const float *x, const float *gz, float gld, int M, float *grad, float *gx,
                      float *workspace, float *img_fwd, const int *fwd_pos, const int *bwd_pos, hipStream_t st) {
    TrainArgs a;
    memset(&a, 0, sizeof(a));
    a.w = const_cast<float *>(packed);
    a.img_fwd = img_fwd;
    a.img_bwd = workspace;
    a.stash = workspace + s.image_floats + s.num_params() + 64;
    a.fwd_pos = fwd_pos;
    a.bwd_pos = bwd_pos;
    a.grad = grad;
    a.s = s;
    a.xtrain = x;
    a.n_train = M;
    a.batch = M;
    a.gz = gz; a.gx = gx; a.gld_in = gld;
    a.mode = TRAIN_MODE_VJP;
    return dispatch_train(a, st);
}

// torch.optim.Adam (coupled weight decay) over the packed vector from an externally supplied gradient: the optimiser step
// of a composite model whose stages are separate handles (fast/slow hierarchy)
__global__ void adam_packed_kernel(float *__restrict__ w, const float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v, int n,
                                   float step_size, float inv_bc2s, float wd) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float gi = grad[i] + wd * w[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w[i] = w[i] - step_size * (mi / (sqrtf(vi) * inv_bc2s + eps));
    }
}

// the same step with the step count on the DEVICE (the handle's counter, which the training kernels advance too): no read-back,
// no stream synchronisation between the minibatches of a host-driven epoch loop.  Bias corrections as on the host (float64 pow).
__global__ void adam_packed_dev_kernel(float *__restrict__ w, const float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v, int n,
                                       const int *__restrict__ step_dev, float lr, float wd) {
    const int step = *step_dev + 1;
    const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
    const float step_size = (float)((double)lr / bc1), inv_bc2s = (float)(1.0 / sqrt(bc2));
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float mi = m[i], vi = v[i];
        w[i] = adam_elem(w[i], grad[i], mi, vi, step_size, inv_bc2s, wd);
        m[i] = mi; v[i] = vi;
    }
}
__global__ void adam_step_advance_kernel(int *step_dev) { *step_dev += 1; }

hipError_t launch_adam_packed_dev(float *w, const float *grad, float *m, float *v, int n, int *step_dev, float lr, float wd, hipStream_t st) {
    hipLaunchKernelGGL(adam_packed_dev_kernel, dim3(64), dim3(256), 0, st, w, grad, m, v, n, step_dev, lr, wd);
    hipLaunchKernelGGL(adam_step_advance_kernel, dim3(1), dim3(1), 0, st, step_dev);
    return hipGetLastError();
}

hipError_t launch_adam_packed(float *w, const float *grad, float *m, float *v, int n, int step, float lr, float wd, hipStream_t st) {
    const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
    hipLaunchKernelGGL(adam_packed_kernel, dim3(64), dim3(256), 0, st, w, grad, m, v, n, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), wd);
    return hipGetLastError();
}

hipError_t launch_train(float *packed, float *adam_m, float *adam_v, float *best_w, float *img, int *adam_step_dev,
                        const FlowShape &s, const float *xtrain, int n_train, const float *xvalid, int n_valid,
                        const int *perm, const float *noise, uint64_t seed, float jitter, int batch, int max_epochs,
                        int patience, float lr, float wd, int epoch_offset, int flags, float *losses,
                        nnest_train_result_t *result, float *workspace, const int *fwd_pos, const int *bwd_pos,
                        hipStream_t st) {
    TrainArgs a;
    memset(&a, 0, sizeof(a));
    a.w = packed; a.m = adam_m; a.v = adam_v; a.best_w = best_w; a.img_fwd = img;
    a.img_bwd = workspace;
    a.stash = workspace + s.image_floats + s.num_params() + 64;
    a.fwd_pos = fwd_pos;
    a.bwd_pos = bwd_pos;
    a.grad = workspace + s.image_floats;
    a.adam_step = adam_step_dev;
    a.s = s;
    a.xtrain = xtrain; a.n_train = n_train; a.xvalid = xvalid; a.n_valid = n_valid;
    a.perm = perm; a.noise = noise; a.seed = seed; a.jitter = jitter; a.batch = batch;
    a.max_epochs = max_epochs; a.patience = patience; a.lr = lr; a.wd = wd;
    a.losses = losses; a.result = result;
    a.epoch_offset = epoch_offset; a.flags = flags;
    a.mode = TRAIN_MODE_EPOCHS;
    if (grid_eligible(a) && pipe_eligible(a)) return dispatch_train_pipe(a, workspace + ((single_workspace_floats(s) + 63) & ~(size_t)63), st);
#ifdef NNEST_DEV_PIPE2
    if (a.mode == TRAIN_MODE_EPOCHS) return hipErrorInvalidConfiguration;
#endif
    if (grid_eligible(a) && rows_eligible(a)) return dispatch_train_rows(a, workspace + ((single_workspace_floats(s) + 63) & ~(size_t)63), st);
    if (grid_eligible(a)) return dispatch_train_grid(a, workspace + ((single_workspace_floats(s) + 63) & ~(size_t)63), st);
    return dispatch_train(a, st);
}

}  // namespace nnest
