// maf_train.h -- gradient of the masked autoregressive flow's training loss (included at the end of nnest_train.hip, inside
// namespace nnest).  [Build-defined flow, maf_tile.h; parity against a CPU restatement of the same definition (tests/).]
//
// loss = -mean(log_probs(x)) over one minibatch (Trainer._train, trainer.py:394) and dloss/dw:
//   maf_grad_kernel    two waves per 16-row tile, one per net: forward (block inputs and hidden activations kept in LDS), loss partial,
//                      hand-written backward block by block with the tile code of the coupling stack run dense over both
//                      parity classes (mlp_fwd_keep / mlp_bwd / contract_rows with 2 NT tiles), per-tile weight-gradient
//                      tiles to a job-slot buffer -- one producer per element, no atomics;
//   maf_reduce_kernel  the tiles' partials summed in tile order into the packed gradient through a parameter -> slot map
//                      (masked parameters have no slot: gradient exactly zero), and the loss.
// The epoch loop itself is driven from the host (nnest_amd/maf.py: loss_grad + adam_step per minibatch) -- this flow's first
// version; the persistent one-launch loop of the RealNVP path (train_kernel_grid) is what it would grow into.

struct MafGradArgs {
    const float *imgf, *imgb;
    FlowShape s;
    const float *x;   // [M, D] the minibatch (jitter already applied)
    int M;
    float *gpart;     // [ntile][2 NJ 256]: weight-gradient tiles, then bias tiles
    float *lpart;     // [ntile] sum of the tile's log_probs
};

// what job J of (block, net) bn produces: lane (gq, j) register r of its result tile / of its bias vector -> packed index (-1:
// padding or masked).  Jobs per net: NT2 output-layer tiles, L hidden layers, NT2 first-layer tiles.
__host__ __device__ inline void maf_job_targets(const FlowShape &s, int J, int lane, int r, int *wt, int *bt) {
    const int NT = s.NT, NT2 = 2 * s.NT, L = s.L, D = s.D, H = s.H;
    const int NJOBS = NT2 + L + NT2;
    const int bn = J / NJOBS, b = bn >> 1;
    int q = J % NJOBS;
    const int pb0 = H * D, phid = H * D + H, pWo = H * D + H + L * (H * H + H), pbo = pWo + D * H;
    const int pbase = bn * s.net_params;
    const int gq = lane >> 4, j = lane & 15;
    *wt = -1; *bt = -1;
    if (q < NT2) {   // dWout[dim][hidden]
        const int d = maf_dim(NT, q, 4 * gq + r);
        if (d < D) {
            if (maf_deg_in(D, b, d) > maf_deg_hid(D, H, j)) *wt = pbase + pWo + d * H + j;
            if (j == 0) *bt = pbase + pbo + d;
        }
        return;
    }
    q -= NT2;
    if (q < L) {     // dW_l[out][in]
        const int ko = 4 * gq + r;
        if (maf_deg_hid(D, H, ko) >= maf_deg_hid(D, H, j)) *wt = pbase + phid + q * (H * H + H) + ko * H + j;
        if (j == 0) *bt = pbase + phid + q * (H * H + H) + H * H + ko;
        return;
    }
    q -= L;          // dW0[hidden][dim]
    const int d = maf_dim(NT, q, j), k = 4 * gq + r;
    if (d < D && maf_deg_hid(D, H, k) >= maf_deg_in(D, b, d)) *wt = pbase + k * D + d;
    if (q == 0 && j == 0) *bt = pbase + pb0 + k;
}

// packed parameter -> slot of a tile's job-result buffer ([NJ][64][4] weight tiles, then [NJ][64][4] bias tiles); -1: none
__global__ void maf_gpos_kernel(int *__restrict__ gpos, FlowShape s) {
    const int NJ = s.B * 2 * (4 * s.NT + s.L);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NJ * 256; i += gridDim.x * blockDim.x) {
        int wt, bt;
        maf_job_targets(s, i >> 8, (i >> 2) & 63, i & 3, &wt, &bt);
        if (wt >= 0) gpos[wt] = i;
        if (bt >= 0) gpos[bt] = NJ * 256 + i;
    }
}

// Two waves per 16-row tile, one per net (wave 0: scale net / tanh, wave 1: translate net / relu): both carry the tile, each runs its
// net's MLP forward, its backward and its weight-gradient contractions, and they meet in LDS where the block needs both nets --
// log s and t for the affine update, the two first-layer contributions to dL/dx.  Every value is produced by the same
// instruction sequence on the same operands as in the one-wave version (round 3's first), and the sums that combine the nets
// keep its order, (gy e^s + gm_s) + gm_t: the same bits.
template <int NT, int L>
__global__ void __launch_bounds__(128) maf_grad_kernel(MafGradArgs a) {
    constexpr int NT2 = 2 * NT, NH = 1;
    typedef StageMap<NT2, NH, L> SM;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int B = a.s.B, D = a.s.D, M = a.M;
    f32x4 *xst = reinterpret_cast<f32x4 *>(smem);                 // [B][NT2][64]     block inputs
    f32x4 *ast = xst + (size_t)B * NT2 * 64;                      // [B][2][L+1][64]  hidden activations of both nets
    f32x4 *xch = ast + (size_t)B * 2 * (L + 1) * 64;              // [2][NT2][64]     what one net's wave hands the other
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), w = lane & 15, g = lane >> 4;
    float *stg = reinterpret_cast<float *>(xch + 2 * NT2 * 64) + (size_t)wave * SM::count * 16 * 16;   // [SM::count][16 rows][16] per wave
    const int tile = blockIdx.x, row = tile * 16 + w;
    const bool row_ok = row < M;
    const int nf = a.s.net_floats;
    constexpr int NJOBS = NT2 + L + NT2;
    const int NJ = B * 2 * NJOBS;
    float *gp = a.gpart + (size_t)tile * 2 * NJ * 256;

    f32x4 xs[2][NT];
    load_tile<NT>(a.x, row, row_ok, D, lane, xs);
    f32x4 (&v)[NT2] = reinterpret_cast<f32x4 (&)[NT2]>(xs);
    float ldp = 0.f;
    for (int b = 0; b < B; ++b) {
        const float *wf = a.imgf + (size_t)b * 2 * nf;
        f32x4 act[L + 1][NH], out[NT2], ls[NT2], t[NT2];
        if (wave == 0) {
#pragma unroll
            for (int tp = 0; tp < NT2; ++tp) xst[((size_t)b * NT2 + tp) * 64 + lane] = v[tp];
            mlp_fwd_keep<NT2, NH, L, 0>(wf, lane, v, act, out);
        } else {
            mlp_fwd_keep<NT2, NH, L, 1>(wf + nf, lane, v, act, out);
        }
#pragma unroll
        for (int l = 0; l <= L; ++l) ast[(((size_t)b * 2 + wave) * (L + 1) + l) * 64 + lane] = act[l][0];
#pragma unroll
        for (int tp = 0; tp < NT2; ++tp) xch[(wave * NT2 + tp) * 64 + lane] = out[tp];
        __syncthreads();
#pragma unroll
        for (int tp = 0; tp < NT2; ++tp) {
            ls[tp] = xch[(0 * NT2 + tp) * 64 + lane];
            t[tp] = xch[(1 * NT2 + tp) * 64 + lane];
        }
        ldp += affine_update<NT2, false>(ls, t, v);
        __syncthreads();
    }
    const float ld = group_sum(ldp);
    float ss = 0.f;
#pragma unroll
    for (int tp = 0; tp < NT2; ++tp) ss += base_E4(v[tp], a.s.base_beta);
    ss = group_sum(ss);
    float lp = (row_ok && g == 0) ? (-ss + a.s.base_const * (float)D + ld) : 0.f;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) lp += __shfl_xor(lp, o);
    if (lane == 0 && wave == 0) a.lpart[tile] = lp;

    // d(loss)/du = dE/du / M ; d(loss)/d(logdet) = -1/M
    const float invM = 1.0f / (float)M, gld = -invM;
    f32x4 gy[NT2];
#pragma unroll
    for (int tp = 0; tp < NT2; ++tp) gy[tp] = row_ok ? base_dE4(v[tp], a.s.base_beta) * invM : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int b = B - 1; b >= 0; --b) {
        const float *wf = a.imgf + (size_t)b * 2 * nf, *wb = a.imgb + (size_t)b * 2 * nf;
        f32x4 x[NT2], act[L + 1][NH], ls[NT2], g_out[NT2], gm[NT2], part[NT2];
#pragma unroll
        for (int tp = 0; tp < NT2; ++tp) x[tp] = xst[((size_t)b * NT2 + tp) * 64 + lane];
#pragma unroll
        for (int l = 0; l <= L; ++l) act[l][0] = ast[(((size_t)b * 2 + wave) * (L + 1) + l) * 64 + lane];
        if (wave == 0) {
            mlp_out_layer<NT2, NH, L>(wf, lane, act[L], ls);   // log_s again from the kept activations (same accumulation order)
#pragma unroll
            for (int tp = 0; tp < NT2; ++tp) {
                const float lsv[4] = {ls[tp].x, ls[tp].y, ls[tp].z, ls[tp].w}, xv[4] = {x[tp].x, x[tp].y, x[tp].z, x[tp].w};
                const float gv[4] = {gy[tp].x, gy[tp].y, gy[tp].z, gy[tp].w};
                float o_gls[4], o_gx[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool valid = row_ok && maf_dim(NT, tp, 4 * g + r) < D;
                    const float e = __expf(lsv[r]);
                    o_gls[r] = valid ? gv[r] * (xv[r] * e) + gld : 0.f;   // z = x e^{s} + t ; logdet += s
                    o_gx[r] = gv[r] * e;                                  // direct path dz/dx
                }
                g_out[tp] = (f32x4){o_gls[0], o_gls[1], o_gls[2], o_gls[3]};
                part[tp] = (f32x4){o_gx[0], o_gx[1], o_gx[2], o_gx[3]};
            }
        } else {
#pragma unroll
            for (int tp = 0; tp < NT2; ++tp) {
                const float gv[4] = {gy[tp].x, gy[tp].y, gy[tp].z, gy[tp].w};
                float o_gt[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o_gt[r] = (row_ok && maf_dim(NT, tp, 4 * g + r) < D) ? gv[r] : 0.f;
                g_out[tp] = (f32x4){o_gt[0], o_gt[1], o_gt[2], o_gt[3]};
                part[tp] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();   // (orders each wave's LDS traffic around the reuse of its staging area)
        if (wave == 0) mlp_bwd<NT2, NH, L, 0>(wb, lane, stg, 16, w, g_out, act, gm);
        else           mlp_bwd<NT2, NH, L, 1>(wb + nf, lane, stg, 16, w, g_out, act, gm);
#pragma unroll
        for (int tp = 0; tp < NT2; ++tp) stage_tile(stg, 16, SM::m(tp), w, lane, row_ok ? x[tp] : (f32x4){0.f, 0.f, 0.f, 0.f});
        __syncthreads();
        const int J0 = (b * 2 + wave) * NJOBS;
        f32x4 bt = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < NJOBS; ++q) {
            f32x4 t;
            bt = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (q < NT2) t = contract_rows<true>(stg, 16, SM::gout(q), SM::act(L, 0), lane, bt);
            else if (q - NT2 < L) t = contract_rows<true>(stg, 16, SM::gpre(q - NT2 + 1, 0), SM::act(q - NT2, 0), lane, bt);
            else if (q - NT2 - L == 0) t = contract_rows<true>(stg, 16, SM::gpre(0, 0), SM::m(0), lane, bt);
            else t = contract_rows<false>(stg, 16, SM::gpre(0, 0), SM::m(q - NT2 - L), lane, bt);
            *reinterpret_cast<f32x4 *>(gp + ((size_t)(J0 + q) * 64 + lane) * 4) = t;
            *reinterpret_cast<f32x4 *>(gp + ((size_t)(NJ + J0 + q) * 64 + lane) * 4) = bt;
        }
        // dL/dx of the block: (gy e^s + through the scale net's first layer) + through the translate net's first layer
#pragma unroll
        for (int tp = 0; tp < NT2; ++tp) xch[(wave * NT2 + tp) * 64 + lane] = wave == 0 ? part[tp] + gm[tp] : gm[tp];
        __syncthreads();
#pragma unroll
        for (int tp = 0; tp < NT2; ++tp) gy[tp] = xch[(0 * NT2 + tp) * 64 + lane] + xch[(1 * NT2 + tp) * 64 + lane];
        __syncthreads();
    }
}

__global__ void maf_reduce_kernel(const float *__restrict__ gpart, const float *__restrict__ lpart, const int *__restrict__ gpos,
                                  int np, int ntile, int slots, int M, float *__restrict__ grad, float *__restrict__ loss_out) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < np; p += gridDim.x * blockDim.x) {
        const int sl = gpos[p];
        float g = 0.f;
        if (sl >= 0)
            for (int t = 0; t < ntile; ++t) g += gpart[(size_t)t * slots + sl];
        grad[p] = g;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss_out) {
        float s = 0.f;
        for (int t = 0; t < ntile; ++t) s += lpart[t];
        *loss_out = -s / (float)M;   // loss = -mean(log_probs)  (trainer.py:394)
    }
}

// ---- the minibatch's update in ONE kernel behind maf_grad_kernel (nnest_maf_train_epoch): per parameter the tiles' partials summed
// in tile order (as maf_reduce_kernel), one Adam step (as adam_packed_dev_kernel: the same arithmetic, the step count read from the
// device), and the new weight written to its element of the forward and of the transposed fragment image through position maps
// built once per flow (what maf_repack_kernel gathers).  The epoch's running loss and the step counter are advanced by the block
// that finishes last (every block has read the counter by then).
__global__ void maf_pos_kernel(int *__restrict__ fwd_pos, int *__restrict__ bwd_pos, FlowShape s) {   // both preset to -1
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < s.image_floats; idx += gridDim.x * blockDim.x) {
        const int f = maf_fwd_src(s, idx), b = maf_bwd_src(s, idx);
        if (f >= 0) fwd_pos[f] = idx;
        if (b >= 0) bwd_pos[b] = idx;
    }
}

hipError_t launch_maf_build_pos(int *fwd_pos, int *bwd_pos, const FlowShape &s, hipStream_t st) {
    const size_t nb = (size_t)s.num_params() * sizeof(int);
    hipError_t e = hipMemsetAsync(fwd_pos, 0xFF, nb, st);
    if (e == hipSuccess) e = hipMemsetAsync(bwd_pos, 0xFF, nb, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(maf_pos_kernel, dim3(64), dim3(256), 0, st, fwd_pos, bwd_pos, s);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) maf_update_kernel(const float *__restrict__ gpart, const float *__restrict__ lpart, const int *__restrict__ gpos,
                                                         const int *__restrict__ fwd_pos, const int *__restrict__ bwd_pos, int np, int ntile, int slots,
                                                         int M, float *__restrict__ w, float *__restrict__ m, float *__restrict__ v,
                                                         int *step_dev, float lr, float wd, float *__restrict__ imgf, float *__restrict__ imgb,
                                                         float *loss_acc, unsigned int *ticket) {
    const int step = *step_dev + 1;
    const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
    const float step_size = (float)((double)lr / bc1), inv_bc2s = (float)(1.0 / sqrt(bc2));
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < np; p += gridDim.x * blockDim.x) {
        const int sl = gpos[p];
        float g = 0.f;
        if (sl >= 0)
            for (int t = 0; t < ntile; ++t) g += gpart[(size_t)t * slots + sl];
        float mi = m[p], vi = v[p];
        const float wn = adam_elem(w[p], g, mi, vi, step_size, inv_bc2s, wd);
        m[p] = mi; v[p] = vi;
        w[p] = wn;
        const int pf = fwd_pos[p], pb = bwd_pos[p];
        if (pf >= 0) imgf[pf] = wn;
        if (pb >= 0) imgb[pb] = wn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned int t = atomicAdd(ticket, 1u);
        if (t == gridDim.x - 1) {   // the last block: every block has read the step count
            *ticket = 0u;
            *step_dev = step;
            if (loss_acc) {
                float s = 0.f;
                for (int k = 0; k < ntile; ++k) s += lpart[k];
                *loss_acc += -s / (float)M;   // loss = -mean(log_probs)  (trainer.py:394), summed over the epoch (:402)
            }
        }
    }
}

// gradient of one minibatch + the update above: two launches
hipError_t launch_maf_train_minibatch(const FlowShape &s, float *imgf, float *imgb, const int *gpos, const int *fwd_pos, const int *bwd_pos,
                                      const float *x, int M, float *w, float *m, float *v, int *step_dev, float lr, float wd,
                                      float *loss_acc, unsigned int *ticket, float *workspace, hipStream_t st);

size_t maf_workspace_floats(const FlowShape &s) {   // gpart for 8 tiles (<= 128 rows) + lpart
    const int NJ = s.B * 2 * (4 * s.NT + s.L);
    return (size_t)8 * 2 * NJ * 256 + 16;
}

hipError_t launch_maf_build_gpos(int *gpos, const FlowShape &s, hipStream_t st) {
    hipError_t e = hipMemsetAsync(gpos, 0xFF, (size_t)s.num_params() * sizeof(int), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(maf_gpos_kernel, dim3(64), dim3(256), 0, st, gpos, s);
    return hipGetLastError();
}

template <int NT, int L>
static hipError_t launch_maf_grad_t(const MafGradArgs &a, int ntile, hipStream_t st) {
    const size_t lds = ((size_t)a.s.B * 2 * NT * 64 + (size_t)a.s.B * 2 * (L + 1) * 64 + (size_t)2 * 2 * NT * 64) * sizeof(f32x4) +
                       (size_t)2 * StageMap<2 * NT, 1, L>::count * 16 * 16 * sizeof(float);
    if (lds > 160 * 1024 - 512) return hipErrorInvalidConfiguration;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(maf_grad_kernel<NT, L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((maf_grad_kernel<NT, L>), dim3(ntile), dim3(128), lds, st, a);
    return hipGetLastError();
}

// loss and dloss/dw of one minibatch x[M, D] (M <= 128); hidden_dim 16, num_layers 0..2
hipError_t launch_maf_loss_grad(const FlowShape &s, const float *imgf, const float *imgb, const int *gpos, const float *x, int M,
                                float *grad, float *loss, float *workspace, hipStream_t st) {
    if (s.NH != 1 || s.L > 2 || M < 1 || M > 128) return hipErrorInvalidConfiguration;
    const int ntile = (M + 15) / 16;
    const int NJ = s.B * 2 * (4 * s.NT + s.L);
    MafGradArgs a;
    a.imgf = imgf; a.imgb = imgb; a.s = s; a.x = x; a.M = M;
    a.gpart = workspace;
    a.lpart = workspace + (size_t)8 * 2 * NJ * 256;
    hipError_t e = hipErrorInvalidConfiguration;
#define MAF_GRAD(nt, l) if (s.NT == nt && s.L == l) e = launch_maf_grad_t<nt, l>(a, ntile, st)
    MAF_GRAD(1, 0); MAF_GRAD(2, 0); MAF_GRAD(3, 0); MAF_GRAD(4, 0);
    MAF_GRAD(1, 1); MAF_GRAD(2, 1); MAF_GRAD(3, 1); MAF_GRAD(4, 1);
    MAF_GRAD(1, 2); MAF_GRAD(2, 2); MAF_GRAD(3, 2); MAF_GRAD(4, 2);
#undef MAF_GRAD
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(maf_reduce_kernel, dim3(64), dim3(256), 0, st, a.gpart, a.lpart, gpos, s.num_params(), ntile, 2 * NJ * 256, M, grad, loss);
    return hipGetLastError();
}

hipError_t launch_maf_train_minibatch(const FlowShape &s, float *imgf, float *imgb, const int *gpos, const int *fwd_pos, const int *bwd_pos,
                                      const float *x, int M, float *w, float *m, float *v, int *step_dev, float lr, float wd,
                                      float *loss_acc, unsigned int *ticket, float *workspace, hipStream_t st) {
    if (s.NH != 1 || s.L > 2 || M < 1 || M > 128) return hipErrorInvalidConfiguration;
    const int ntile = (M + 15) / 16;
    const int NJ = s.B * 2 * (4 * s.NT + s.L);
    MafGradArgs a;
    a.imgf = imgf; a.imgb = imgb; a.s = s; a.x = x; a.M = M;
    a.gpart = workspace;
    a.lpart = workspace + (size_t)8 * 2 * NJ * 256;
    hipError_t e = hipErrorInvalidConfiguration;
#define MAF_GRAD(nt, l) if (s.NT == nt && s.L == l) e = launch_maf_grad_t<nt, l>(a, ntile, st)
    MAF_GRAD(1, 0); MAF_GRAD(2, 0); MAF_GRAD(3, 0); MAF_GRAD(4, 0);
    MAF_GRAD(1, 1); MAF_GRAD(2, 1); MAF_GRAD(3, 1); MAF_GRAD(4, 1);
    MAF_GRAD(1, 2); MAF_GRAD(2, 2); MAF_GRAD(3, 2); MAF_GRAD(4, 2);
#undef MAF_GRAD
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(maf_update_kernel, dim3(64), dim3(256), 0, st, a.gpart, a.lpart, gpos, fwd_pos, bwd_pos, s.num_params(), ntile,
                       2 * NJ * 256, M, w, m, v, step_dev, lr, wd, imgf, imgb, loss_acc, ticket);
    return hipGetLastError();
}
