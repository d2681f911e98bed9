// nnest_train_grid.h -- K5 over EIGHT compute units (included at the end of nnest_train.hip, inside namespace nnest).
//
// train_kernel runs a minibatch on one workgroup: 7 row tiles share 4 SIMDs, and the weight-gradient contractions (30 tile
// jobs at the default shape) run 5 at a time between workgroup barriers -- 38 us per minibatch, 72 % of a config-2 run, on
// 1 of 256 CUs.  train_kernel_grid gives every 16-row tile of the minibatch its own workgroup (G = 8 workgroups, one per CU):
//   F+B  waves 0 and 1 of workgroup g (scale net / translate net of every block, exchanging their outputs through LDS as K4's
//        team form does): rows 16g..16g+15 forward and backward through all blocks with no grid-level step in between (the
//        per-row gradients G and activations go to a GLOBAL staging area, one region per (block, net), instead of one LDS
//        region reused under workgroup barriers);
//   grid barrier;
//   W    the 30 weight-gradient jobs, one per wave across the 64 waves of the grid, each contracting over all rows exactly
//        as the single-workgroup kernel does (same operand order: the same bits), result tile to a job-private slot;
//   grid barrier;
//   A    EVERY workgroup applies Adam to the whole parameter vector (identical values; no third barrier) and refreshes its
//        own LDS copies of the fragment images.
// Validation: tile-sets per workgroup as the single-workgroup kernel gives them to its waves, partial sums exchanged at one
// more grid barrier per epoch; summation orders are those of train_kernel, so the two kernels agree to the last bit or two
// (hipcc fuses a few multiply-adds differently in the two bodies) and each is bitwise reproducible run to run
// (tests/test_gpu_train.py::test_grid_training_vs_single_workgroup).
//
// Cross-CU visibility (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility"): every byte
// that crosses workgroups is written by `sc1` stores (write-through) as whole 1-KB rows of 16-byte stores by one wave
// instruction, read by `sc1` dword loads (relaxed agent-scope atomic loads), and ordered by: storing waves' s_waitcnt
// vmcnt(0) -> workgroup barrier -> one lane's agent-scope atomic add on a counter -> sc1 poll of the counter by one lane
// -> workgroup barrier -> loads.  Polls are bounded (a counter that never fills sets the error word and the kernel ends).

enum { GRID_WG = 8, GRID_MAX_POLLS = 1 << 22, NNEST_TRAIN_POS_LDS = 1 << 20 /* internal flag bit, never part of the ABI */ };

// (the s_nop: a VMEM store of more than 8 bytes reads its data registers for up to two cycles after issue, and the compiler's
// hazard recognizer cannot see inside the asm -- without it the next VALU write corrupted some lanes of some stores)
__device__ __forceinline__ void st_sc1_f32x4(float *p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float ld_sc1(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all G workgroups have finished what precedes; `phase` counts barriers (same value in every thread of the grid)
__device__ __forceinline__ bool grid_barrier(unsigned int *ctr, int &phase, int G, int *err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores (incl. the asm ones the compiler does not count)
#ifdef GRID_FENCE
    __threadfence();
#endif
    __syncthreads();
    phase += 1;
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int want = (unsigned int)G * (unsigned int)phase;
        int polls = 0, good = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            if (++polls > GRID_MAX_POLLS) { good = 0; *err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        ok = good;
    }
    __syncthreads();
#ifdef GRID_FENCE
    __threadfence();
#endif
    return ok != 0;
}

// what one weight-gradient job produces: lane (gq, j) register r of its result tile / of its bias vector -> packed
// parameter index (or -1).  Job J = (block, net, q) with q enumerated as in weight_grad_jobs.
template <int NT, int NH, int L>
__host__ __device__ inline void grid_job_targets(const FlowShape &s, int J, int lane, int r, int *wt, int *bt) {
    constexpr int J_W3 = NT * NH, J_W2 = L * NH * NH, NJOBS = J_W3 + J_W2 + NH * NT;
    const int D = s.D, H = s.H;
    const int bn = J / NJOBS, b = bn >> 1;
    int q = J % NJOBS;
    const int pc = (b + 1) & 1, pt = b & 1;
    const int pb0 = H * D, phid = H * D + H, pWo = H * D + H + L * (H * H + H), pbo = pWo + D * H;
    const int pbase = bn * s.net_params;
    const int gq = lane >> 4, j = lane & 15;
    *wt = -1; *bt = -1;
    if (q < J_W3) {
        const int tau = q / NH, ht = q % NH, d = 2 * (16 * tau + 4 * gq + r) + pt;
        if (d < D) {
            *wt = pbase + pWo + d * H + 16 * ht + j;
            if (ht == 0 && j == 0) *bt = pbase + pbo + d;
        }
        return;
    }
    q -= J_W3;
    if (q < J_W2) {
        const int l = q / (NH * NH) + 1, hto = (q / NH) % NH, hti = q % NH;
        *wt = pbase + phid + (l - 1) * (H * H + H) + (16 * hto + 4 * gq + r) * H + 16 * hti + j;
        if (hti == 0 && j == 0) *bt = pbase + phid + (l - 1) * (H * H + H) + H * H + 16 * hto + 4 * gq + r;
        return;
    }
    q -= J_W2;
    const int ht = q / NT, tau = q % NT, d = 2 * (16 * tau + j) + pc;
    if (d < D) *wt = pbase + (16 * ht + 4 * gq + r) * D + d;
    if (tau == 0 && j == 0) *bt = pbase + pb0 + 16 * ht + 4 * gq + r;
}

// packed parameter -> slot of the job-result buffer ([jobs][64 lanes][4] tiles, then the same for the bias vectors); -1 where
// no job produces a gradient (the parameters the mask never reaches)
template <int NT, int NH, int L>
__global__ void grid_gpos_kernel(int *__restrict__ gpos, FlowShape s) {
    constexpr int NJOBS = NT * NH + L * NH * NH + NH * NT;
    const int NJ = s.B * 2 * NJOBS;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NJ * 256; i += gridDim.x * blockDim.x) {
        const int J = i >> 8, lane = (i >> 2) & 63, r = i & 3;
        int wt, bt;
        grid_job_targets<NT, NH, L>(s, J, lane, r, &wt, &bt);
        if (wt >= 0) gpos[wt] = i;
        if (bt >= 0) gpos[bt] = NJ * 256 + i;
    }
}

// contract_rows over the global staging area (row stride 128, sc1 loads): the same operand order as contract_rows
template <bool WITH_BIAS>
__device__ __forceinline__ f32x4 contract_rows_grid(const float *stg, int rows_pad, int ct_g, int ct_a, int lane, f32x4 &bias) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 a0 = zero4, a1 = zero4, a2 = zero4, a3 = zero4;
    float bs = 0.f;
    const float *G = stg + (size_t)ct_g * TRAIN_MAX_ROWS * 16 + (lane >> 4) * 16 + (lane & 15);
    const float *A = stg + (size_t)ct_a * TRAIN_MAX_ROWS * 16 + (lane >> 4) * 16 + (lane & 15);
    // every operand of the job is requested before the first MFMA: an sc1 load is a round trip to the memory side (~1.5 us), and
    // with the loads inside the row loop the job paid one round trip per 16 rows (20 k cycles per job phase, measured).  Rows
    // beyond rows_pad hold an earlier minibatch's data: their loads are redirected to tile 0 and their products skipped.
    constexpr int NIT = TRAIN_MAX_ROWS / 16;
    float gv[NIT][4], ev[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int r = it * 16 < rows_pad ? it * 16 : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            gv[it][k] = ld_sc1(G + (r + 4 * k) * 16);
            ev[it][k] = ld_sc1(A + (r + 4 * k) * 16);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (it * 16 < rows_pad) {
            a0 = mfma4(gv[it][0], ev[it][0], a0);
            a1 = mfma4(gv[it][1], ev[it][1], a1);
            a2 = mfma4(gv[it][2], ev[it][2], a2);
            a3 = mfma4(gv[it][3], ev[it][3], a3);
            if (WITH_BIAS) bs += (gv[it][0] + gv[it][1]) + (gv[it][2] + gv[it][3]);
        }
    }
    if (WITH_BIAS) {
        bs += __shfl_xor(bs, 16);
        bs += __shfl_xor(bs, 32);
        const int q4 = (lane >> 4) * 4;
        bias = (f32x4){__shfl(bs, q4 + 0), __shfl(bs, q4 + 1), __shfl(bs, q4 + 2), __shfl(bs, q4 + 3)};
    }
    return (a0 + a1) + (a2 + a3);
}

// staging into the global area: [ct][128 rows][16], one 16-byte sc1 store per lane = one whole 1-KB tile row set per instruction
__device__ __forceinline__ void stage_tile_grid(float *stg, int ct, int row, int lane, f32x4 v) {
    st_sc1_f32x4(stg + ((size_t)ct * TRAIN_MAX_ROWS + row) * 16 + (lane >> 4) * 4, v);
}

// mlp_bwd with the global staging (same arithmetic; `row` = row inside the minibatch)
template <int NT, int NH, int L, int ACT>
__device__ __forceinline__ void mlp_bwd_grid(const float *__restrict__ bn, int lane, float *stg, int row, const f32x4 (&g_out)[NT],
                                             const f32x4 (&acts)[L + 1][NH], f32x4 (&g_m)[NT]) {
    typedef StageMap<NT, NH, L> SM;
    const float *B3 = bn + frag_off_L1() + 4 * lane;
    const float *B2 = bn + frag_off_L2(NT, NH) + 4 * lane;
    const float *B1 = bn + frag_off_L3(NT, NH, L) + 4 * lane;
    f32x4 gh[NH];
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) stage_tile_grid(stg, SM::gout(tau), row, lane, g_out[tau]);
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
            stage_tile_grid(stg, SM::gpre(l, ht), row, lane, gpre[ht]);
            stage_tile_grid(stg, SM::act(l, ht), row, lane, acts[l][ht]);
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
        stage_tile_grid(stg, SM::gpre(0, ht), row, lane, gpre0[ht]);
        stage_tile_grid(stg, SM::act(0, ht), row, lane, acts[0][ht]);
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

// block_backward on one tile with no barrier: both nets' staging goes to their own global regions (affine couplings only)
template <int NT, int NH, int L>
__device__ __forceinline__ void block_backward_grid(const TrainArgs &a, int b, int lane, int row, bool row_ok, const f32x4 (&cond)[NT],
                                                    f32x4 (&ytrans)[NT], f32x4 (&gcond)[NT], f32x4 (&gtrans)[NT], float gld,
                                                    const float *imgf, const float *imgb, const f32x4 *__restrict__ stash) {
    typedef StageMap<NT, NH, L> SM;
    const int pt = b & 1;
    const float *wf = imgf + (size_t)b * 2 * a.s.net_floats;
    const float *wb = imgb + (size_t)b * 2 * a.s.net_floats;
    float *stg_s = a.gstage + (size_t)(b * 2 + 0) * SM::count * TRAIN_MAX_ROWS * 16;
    float *stg_t = a.gstage + (size_t)(b * 2 + 1) * SM::count * TRAIN_MAX_ROWS * 16;
    f32x4 as[L + 1][NH], at[L + 1][NH], ls[NT], t[NT], g_ls[NT], g_t[NT], gm_s[NT], gm_t[NT];
    const int g = lane >> 4;
#pragma unroll
    for (int l = 0; l <= L; ++l)
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
            as[l][ht] = stash[((0 * (L + 1) + l) * NH + ht) * 64 + lane];
            at[l][ht] = stash[((1 * (L + 1) + l) * NH + ht) * 64 + lane];
        }
    mlp_out_layer<NT, NH, L>(wf, lane, as[L], ls);
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
            float ymt = yv[r] - tv[r];
            o_gls[r] = valid ? gv[r] * ymt + gld : 0.f;
            o_gt[r] = valid ? gv[r] : 0.f;
            o_x[r] = ymt * __expf(-lsv[r]);
            o_gx[r] = gv[r] * __expf(lsv[r]);
        }
        g_ls[tau] = (f32x4){o_gls[0], o_gls[1], o_gls[2], o_gls[3]};
        g_t[tau] = (f32x4){o_gt[0], o_gt[1], o_gt[2], o_gt[3]};
        ytrans[tau] = (f32x4){o_x[0], o_x[1], o_x[2], o_x[3]};
        gtrans[tau] = (f32x4){o_gx[0], o_gx[1], o_gx[2], o_gx[3]};
    }
    mlp_bwd_grid<NT, NH, L, 0>(wb, lane, stg_s, row, g_ls, as, gm_s);
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        f32x4 mv = cond[tau];
        if (!row_ok) mv = (f32x4){0.f, 0.f, 0.f, 0.f};
        stage_tile_grid(stg_s, SM::m(tau), row, lane, mv);   // the conditioning input is shared by the two nets:
        stage_tile_grid(stg_t, SM::m(tau), row, lane, mv);   // staged once per region
    }
    mlp_bwd_grid<NT, NH, L, 1>(wb + a.s.net_floats, lane, stg_t, row, g_t, at, gm_t);
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) gcond[tau] = gcond[tau] + gm_s[tau] + gm_t[tau];
}

// ---- F+B of one tile on TWO waves: wave 0 the scale net, wave 1 the translate net of every block (as K4's team form) ----
// Workgroup barrier for data exchanged through LDS only: __syncthreads() would also drain the sc1 staging stores in flight.
__device__ __forceinline__ void lds_barrier_t() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the two net waves swap NT tiles through one LDS buffer [2 roles][NT][64] (two barriers: publish, then release)
template <int NT>
__device__ __forceinline__ void team_swap(f32x4 *xch, int role, int lane, const f32x4 (&mine)[NT], f32x4 (&other)[NT]) {
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) xch[(role * NT + tau) * 64 + lane] = mine[tau];
    lds_barrier_t();
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) other[tau] = xch[((1 - role) * NT + tau) * 64 + lane];
    lds_barrier_t();
}
enum { TEAM_BARRIERS_FWD = 2, TEAM_BARRIERS_BWD = 4 };  // per block

// block_forward_keep with the two nets on two waves: same arithmetic, same stash layout
template <int NT, int NH, int L>
__device__ __forceinline__ float team_block_forward(const float *__restrict__ wf, int net_floats, int role, int lane,
                                                    const f32x4 (&cond)[NT], f32x4 (&trans)[NT], f32x4 *__restrict__ stash, f32x4 *xch) {
    f32x4 acts[L + 1][NH], mine[NT], other[NT];
    if (role == 0) mlp_fwd_keep<NT, NH, L, 0>(wf, lane, cond, acts, mine);
    else           mlp_fwd_keep<NT, NH, L, 1>(wf + net_floats, lane, cond, acts, mine);
#pragma unroll
    for (int l = 0; l <= L; ++l)
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) stash[((role * (L + 1) + l) * NH + ht) * 64 + lane] = acts[l][ht];
    team_swap<NT>(xch, role, lane, mine, other);
    float ld = 0.f;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        const f32x4 ls = role == 0 ? mine[tau] : other[tau], t = role == 0 ? other[tau] : mine[tau];
        trans[tau].x = trans[tau].x * __expf(ls.x) + t.x;
        trans[tau].y = trans[tau].y * __expf(ls.y) + t.y;
        trans[tau].z = trans[tau].z * __expf(ls.z) + t.z;
        trans[tau].w = trans[tau].w * __expf(ls.w) + t.w;
        ld += (ls.x + ls.y) + (ls.z + ls.w);
    }
    return ld;
}

// block_backward_grid with the two nets on two waves
template <int NT, int NH, int L>
__device__ __forceinline__ void team_block_backward(const TrainArgs &a, int b, int role, int lane, int row, bool row_ok,
                                                    const f32x4 (&cond)[NT], f32x4 (&ytrans)[NT], f32x4 (&gcond)[NT], f32x4 (&gtrans)[NT],
                                                    float gld, const float *imgf, const float *imgb, const f32x4 *__restrict__ stash,
                                                    f32x4 *xch) {
    typedef StageMap<NT, NH, L> SM;
    const int pt = b & 1;
    const float *wf = imgf + ((size_t)b * 2 + role) * a.s.net_floats;
    const float *wb = imgb + ((size_t)b * 2 + role) * a.s.net_floats;
    float *stg = a.gstage + (size_t)(b * 2 + role) * SM::count * TRAIN_MAX_ROWS * 16;
    f32x4 acts[L + 1][NH], mine[NT], other[NT], g_mine[NT], gm[NT], gm_other[NT];
    const int g = lane >> 4;
#pragma unroll
    for (int l = 0; l <= L; ++l)
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) acts[l][ht] = stash[((role * (L + 1) + l) * NH + ht) * 64 + lane];
    mlp_out_layer<NT, NH, L>(wf, lane, acts[L], mine);
    team_swap<NT>(xch, role, lane, mine, other);
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        const f32x4 ls = role == 0 ? mine[tau] : other[tau], t = role == 0 ? other[tau] : mine[tau];
        float lsv[4] = {ls.x, ls.y, ls.z, ls.w};
        float tv[4] = {t.x, t.y, t.z, t.w};
        float yv[4] = {ytrans[tau].x, ytrans[tau].y, ytrans[tau].z, ytrans[tau].w};
        float gv[4] = {gtrans[tau].x, gtrans[tau].y, gtrans[tau].z, gtrans[tau].w};
        float o_gls[4], o_gt[4], o_x[4], o_gx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 2 * (16 * tau + 4 * g + r) + pt;
            const bool valid = row_ok && d < a.s.D;
            float ymt = yv[r] - tv[r];
            o_gls[r] = valid ? gv[r] * ymt + gld : 0.f;
            o_gt[r] = valid ? gv[r] : 0.f;
            o_x[r] = ymt * __expf(-lsv[r]);
            o_gx[r] = gv[r] * __expf(lsv[r]);
        }
        g_mine[tau] = role == 0 ? (f32x4){o_gls[0], o_gls[1], o_gls[2], o_gls[3]} : (f32x4){o_gt[0], o_gt[1], o_gt[2], o_gt[3]};
        ytrans[tau] = (f32x4){o_x[0], o_x[1], o_x[2], o_x[3]};
        gtrans[tau] = (f32x4){o_gx[0], o_gx[1], o_gx[2], o_gx[3]};
    }
    if (role == 0) mlp_bwd_grid<NT, NH, L, 0>(wb, lane, stg, row, g_mine, acts, gm);
    else           mlp_bwd_grid<NT, NH, L, 1>(wb, lane, stg, row, g_mine, acts, gm);
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        f32x4 mv = cond[tau];
        if (!row_ok) mv = (f32x4){0.f, 0.f, 0.f, 0.f};
        stage_tile_grid(stg, SM::m(tau), row, lane, mv);
    }
    team_swap<NT>(xch, role, lane, gm, gm_other);
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        const f32x4 gm_s = role == 0 ? gm[tau] : gm_other[tau], gm_t = role == 0 ? gm_other[tau] : gm[tau];
        gcond[tau] = gcond[tau] + gm_s + gm_t;
    }
}

// Adam over the whole packed vector (as adam_sweep: float4 groups, U groups in flight per thread) with the gradient gathered
// from the job results through gpos (sc1 loads: other workgroups wrote them).  Every workgroup steps its OWN replica of
// (w, exp_avg, exp_avg_sq): a shared copy would be read-modify-written by eight workgroups at different times (a slow one
// would step values a fast one had already stepped).
__device__ __forceinline__ void adam_sweep_grid(const TrainArgs &a, const AdamStep &ad, int np, float *imgf, float *imgb, float *rw,
                                                float *rm, float *rv, const int *fpos, const int *bpos) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    constexpr int U = 6;  // 6 x 512 float4 groups cover the default flow's 11 628 parameters in one pass
    const int n4 = np >> 2;
    for (int i0 = threadIdx.x; i0 < n4; i0 += U * blockDim.x) {
        f32x4 w4[U], m4[U], v4[U];
        float g[U][4];
        i32x4 gp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {   // the job slots first (the gathers below depend on them) ...
            const int i = i0 + u * blockDim.x;
            gp[u] = i < n4 ? reinterpret_cast<const i32x4 *>(a.gpos)[i] : (i32x4){-1, -1, -1, -1};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {   // ... then every other operand, all in flight together
            const int i = i0 + u * blockDim.x;
            if (i < n4) {
                g[u][0] = ld_sc1(a.gtile + max(gp[u].x, 0));   // unconditional (no divergent branch per element);
                g[u][1] = ld_sc1(a.gtile + max(gp[u].y, 0));
                g[u][2] = ld_sc1(a.gtile + max(gp[u].z, 0));
                g[u][3] = ld_sc1(a.gtile + max(gp[u].w, 0));
                w4[u] = reinterpret_cast<const f32x4 *>(rw)[i];
                m4[u] = reinterpret_cast<const f32x4 *>(rm)[i];
                v4[u] = reinterpret_cast<const f32x4 *>(rv)[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < n4) {
                float w[4] = {w4[u].x, w4[u].y, w4[u].z, w4[u].w};
                float m[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w}, v[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
                // where the stepped weight sits in the two fragment images: read here (from the LDS copy of the maps when it
                // fits), not held across the long-latency loads above
                const i32x4 fp = reinterpret_cast<const i32x4 *>(fpos)[i], bp = reinterpret_cast<const i32x4 *>(bpos)[i];
                const int f[4] = {fp.x, fp.y, fp.z, fp.w}, bq[4] = {bp.x, bp.y, bp.z, bp.w};
                const int gq[4] = {gp[u].x, gp[u].y, gp[u].z, gp[u].w};   // < 0: no job produces this gradient: exactly zero
#pragma unroll
                for (int k = 0; k < 4; ++k) adam_one(a, ad, w[k], gq[k] >= 0 ? g[u][k] : 0.f, m[k], v[k], f[k], bq[k], imgf, imgb);
                reinterpret_cast<f32x4 *>(rw)[i] = (f32x4){w[0], w[1], w[2], w[3]};
                reinterpret_cast<f32x4 *>(rm)[i] = (f32x4){m[0], m[1], m[2], m[3]};
                reinterpret_cast<f32x4 *>(rv)[i] = (f32x4){v[0], v[1], v[2], v[3]};
            }
        }
    }
    for (int p = 4 * n4 + threadIdx.x; p < np; p += blockDim.x) {  // tail (np not a multiple of 4)
        const int gp = a.gpos[p];
        float w = rw[p], m = rm[p], v = rv[p];
        adam_one(a, ad, w, gp >= 0 ? ld_sc1(a.gtile + gp) : 0.f, m, v, fpos[p], bpos[p], imgf, imgb);
        rw[p] = w; rm[p] = m; rv[p] = v;
    }
}

template <int NT, int NH, int L>
__global__ void __launch_bounds__(TRAIN_THREADS) train_kernel_grid(TrainArgs a) {
    typedef StageMap<NT, NH, L> SM;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *imgf = smem;
    float *imgb = smem + a.s.image_floats;
    // parameter -> image element maps: LDS copies behind the images when they fit (a.flags bit NNEST_TRAIN_POS_LDS, set by the launcher)
    const bool pos_lds = (a.flags & NNEST_TRAIN_POS_LDS) != 0;
    int *lfp = reinterpret_cast<int *>(smem + 2 * a.s.image_floats), *lbp = lfp + ((a.s.num_params() + 3) & ~3);
    const int *fpos = pos_lds ? lfp : a.fwd_pos, *bpos = pos_lds ? lbp : a.bwd_pos;
    __shared__ f32x4 xch[2 * NT * 64];   // the two net waves' exchange buffer
    __shared__ int ctl[4];      // [0] stop flag, [1] counter, [2] best epoch
    __shared__ float ctlf[2];   // [0] best validation loss
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = lane & 15, g = lane >> 4;
    const int wg = blockIdx.x, G = gridDim.x;
    const int D = a.s.D, B = a.s.B;
    const int np = a.s.num_params();
    constexpr int NJOBS = NT * NH + L * NH * NH + NH * NT;
    const int NJ = B * 2 * NJOBS;
    f32x4 *stash_w = reinterpret_cast<f32x4 *>(a.stash) + (size_t)wg * B * 2 * (L + 1) * NH * 64;
    const int npad = (np + 3) & ~3;
    float *rw = a.grep + (size_t)wg * 3 * npad, *rm = rw + npad, *rv = rm + npad;  // this workgroup's replica of w, exp_avg, exp_avg_sq
    int phase = 0;

    rebuild_images_to(a, imgf, imgb);
    for (int i = threadIdx.x; i < np; i += blockDim.x) {
        rw[i] = a.w[i]; rm[i] = a.m[i]; rv[i] = a.v[i];
        if (pos_lds) { lfp[i] = a.fwd_pos[i]; lbp[i] = a.bwd_pos[i]; }
    }
    const bool resume = (a.flags & NNEST_TRAIN_RESUME) != 0;
    if (threadIdx.x == 0) {
        ctl[0] = 0;
        ctl[1] = resume ? a.result->counter : 0;
        ctl[2] = resume ? a.result->best_epoch : 0;
        ctlf[0] = resume ? a.result->best_validation_loss : INFINITY;
    }
    if (wg == 0 && !resume)
        for (int i = threadIdx.x; i < np; i += blockDim.x) a.best_w[i] = a.w[i];  // best_model = deepcopy(netG)  trainer.py:194
    __syncthreads();

    const int n_mb = (a.n_train + a.batch - 1) / a.batch;
    int adam_t = a.adam_step ? *a.adam_step : 0;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0;
    (void)ph; (void)q0; (void)q1; (void)q2; (void)q3; (void)q4; (void)q5; (void)q6;
    int epochs_run = 0, mbcount = 0;
    float last_train_loss = 0.f;
    bool alive = true;

    for (int epoch = 0; epoch < a.max_epochs && alive; ++epoch) {
        float epoch_loss = 0.f;
        for (int mb = 0; mb < n_mb && alive; ++mb, ++mbcount) {
            const int M = min(a.batch, a.n_train - mb * a.batch);
            const int ntile = (M + 15) >> 4;
            const int rows_pad = ntile * 16;
            const int row = wg * 16 + w;
            const bool row_ok = wg < ntile && row < M;
            float *part = a.gpart + (mbcount & 1) * 16;
            AdamStep ad;
            {
                adam_t += 1;
                const double bc1 = 1.0 - pow(0.9, (double)adam_t), bc2 = 1.0 - pow(0.999, (double)adam_t);
                ad.step_size = (float)((double)a.lr / bc1);
                ad.inv_bc2s = (float)(1.0 / sqrt(bc2));
            }
            TSTAMP(q0);
            const bool wg_active = wg < ntile;   // this workgroup owns a tile of this minibatch
            if (wg_active && wave >= 2) {        // the other waves keep the net waves' LDS barriers company
                for (int k = 0; k < B * (TEAM_BARRIERS_FWD + TEAM_BARRIERS_BWD); ++k) lds_barrier_t();
            }
            if (wave < 2) {
                const int role = wave;           // 0: scale net, 1: translate net; both carry the tile and take the same steps
                float lp = 0.f;
                if (wg_active) {
                    f32x4 xs[2][NT], gs[2][NT];
                    // data = X[perm] + jitter * randn  (trainer.py:392)
                    long src = 0;
                    if (row_ok) src = a.perm[(size_t)epoch * a.n_train + mb * a.batch + row];
                    load_tile<NT>(a.xtrain, src, row_ok, D, lane, xs);
                    if (a.jitter != 0.f) {
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
                    float ldp = 0.f;
                    for (int b = 0; b < B; ++b) {
                        const float *wf = imgf + (size_t)b * 2 * a.s.net_floats;
                        f32x4 *sb = stash_w + (size_t)b * 2 * (L + 1) * NH * 64;
                        if (b & 1) ldp += team_block_forward<NT, NH, L>(wf, a.s.net_floats, role, lane, xs[0], xs[1], sb, xch);
                        else       ldp += team_block_forward<NT, NH, L>(wf, a.s.net_floats, role, lane, xs[1], xs[0], sb, xch);
                    }
                    const float ld = group_sum(ldp);
                    float ss = 0.f;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int t = 0; t < NT; ++t) ss += base_E4(xs[c][t], a.s.base_beta);
                    ss = group_sum(ss);
                    lp = (row_ok && g == 0) ? (-ss + a.s.base_const * (float)D + ld) : 0.f;
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) lp += __shfl_xor(lp, o);
                    TSTAMP(q1);
                    // d(loss)/du = dE/du / M ; d(loss)/d(logdet) = -1/M
                    const float invM = 1.0f / (float)M, gld = -invM;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int t = 0; t < NT; ++t) gs[c][t] = row_ok ? base_dE4(xs[c][t], a.s.base_beta) * invM : (f32x4){0.f, 0.f, 0.f, 0.f};
                    for (int b = B - 1; b >= 0; --b) {
                        const f32x4 *sb = stash_w + (size_t)b * 2 * (L + 1) * NH * 64;
                        if (b & 1) team_block_backward<NT, NH, L>(a, b, role, lane, row, row_ok, xs[0], xs[1], gs[0], gs[1], gld, imgf, imgb, sb, xch);
                        else       team_block_backward<NT, NH, L>(a, b, role, lane, row, row_ok, xs[1], xs[0], gs[1], gs[0], gld, imgf, imgb, sb, xch);
                    }
                }
                if (wave == 0 && lane == 0) __hip_atomic_store(part + wg, lp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            TSTAMP(q2);
            alive = grid_barrier(a.gsync, phase, G, a.gerr);
            if (!alive) break;
            TSTAMP(q3);
            // ---- W: one weight-gradient job per wave of the grid ----
            for (int J = wg * TRAIN_WAVES + wave; J < NJ; J += G * TRAIN_WAVES) {
                const int bn = J / NJOBS;
                int q = J % NJOBS;
                const float *stg = a.gstage + (size_t)bn * SM::count * TRAIN_MAX_ROWS * 16;
                f32x4 bt = {0.f, 0.f, 0.f, 0.f}, t;
                constexpr int J_W3 = NT * NH, J_W2 = L * NH * NH;
                if (q < J_W3) {
                    const int tau = q / NH, ht = q % NH;
                    t = ht == 0 ? contract_rows_grid<true>(stg, rows_pad, SM::gout(tau), SM::act(L, ht), lane, bt)
                                : contract_rows_grid<false>(stg, rows_pad, SM::gout(tau), SM::act(L, ht), lane, bt);
                } else if (q - J_W3 < J_W2) {
                    q -= J_W3;
                    const int l = q / (NH * NH) + 1, hto = (q / NH) % NH, hti = q % NH;
                    t = hti == 0 ? contract_rows_grid<true>(stg, rows_pad, SM::gpre(l, hto), SM::act(l - 1, hti), lane, bt)
                                 : contract_rows_grid<false>(stg, rows_pad, SM::gpre(l, hto), SM::act(l - 1, hti), lane, bt);
                } else {
                    q -= J_W3 + J_W2;
                    const int ht = q / NT, tau = q % NT;
                    t = tau == 0 ? contract_rows_grid<true>(stg, rows_pad, SM::gpre(0, ht), SM::m(tau), lane, bt)
                                 : contract_rows_grid<false>(stg, rows_pad, SM::gpre(0, ht), SM::m(tau), lane, bt);
                }
                st_sc1_f32x4(a.gtile + ((size_t)J * 64 + lane) * 4, t);
                st_sc1_f32x4(a.gtile + ((size_t)NJ * 64 + (size_t)J * 64 + lane) * 4, bt);
            }
            TSTAMP(q4);
            alive = grid_barrier(a.gsync, phase, G, a.gerr);
            if (!alive) break;
            TSTAMP(q5);
            // loss = -mean(log_probs)  (trainer.py:394): tile partials in tile order, as train_kernel sums its waves (requested
            // before Adam's loads, summed after: one round trip to the memory side instead of two)
            float lpart[GRID_WG];
#pragma unroll
            for (int k = 0; k < GRID_WG; ++k) lpart[k] = ld_sc1(part + k);
            adam_sweep_grid(a, ad, np, imgf, imgb, rw, rm, rv, fpos, bpos);
            float loss = 0.f;
#pragma unroll
            for (int k = 0; k < GRID_WG; ++k) loss += k < G ? lpart[k] : 0.f;
            loss = -loss / (float)M;
            epoch_loss += loss;
            __syncthreads();
            TSTAMP(q6);
            if (wave == 0 && q1 > q0) { TACC(ph[0], q1, q0); TACC(ph[1], q2, q1); }
            TACC(ph[2], q3, q2); TACC(ph[3], q4, q3); TACC(ph[4], q5, q4); TACC(ph[5], q6, q5);
        }
        if (!alive) break;
        // ---- Trainer._validate (trainer.py:405-418): workgroup g takes the tiles wave g of train_kernel takes ----
        float vsum = 0.f;
        if (wave == 0) {
            const int vtiles = (a.n_valid + 15) >> 4;
            for (int tile = wg; tile < vtiles; tile += TRAIN_WAVES) {
                const int r = tile * 16 + w;
                const bool ok = r < a.n_valid;
                f32x4 xv[2][NT];
                load_tile<NT>(a.xvalid, r, ok, D, lane, xv);
                float ldv = group_sum(flow_forward_tile<NT, NH>(imgf, a.s.net_floats, B, L, lane, xv, nullptr));
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
            if (lane == 0) __hip_atomic_store(a.gpart + 32 + (epoch & 1) * 16 + wg, vsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        alive = grid_barrier(a.gsync, phase, G, a.gerr);
        if (!alive) break;
        float vtot = 0.f;
        for (int k = 0; k < TRAIN_WAVES; ++k) vtot += k < G ? ld_sc1(a.gpart + 32 + (epoch & 1) * 16 + k) : 0.f;
        const float valid_loss = (-vtot / (float)a.n_valid) / (float)a.n_valid;  // mean, then / len(dataset)  :418
        const float train_loss = epoch_loss / (float)a.n_train;                   // trainer.py:403
        last_train_loss = train_loss;
        epochs_run = epoch + 1;
        if (a.losses && wg == 0 && threadIdx.x == 0) {
            a.losses[2 * epoch] = train_loss;
            a.losses[2 * epoch + 1] = valid_loss;
        }
        // early stopping bookkeeping (trainer.py:205-209, :223-232); every thread of the grid evaluates the same values
        const bool improved = valid_loss < ctlf[0];
        __syncthreads();
        if (improved) {
            if (wg == 0)
                for (int i = threadIdx.x; i < np; i += blockDim.x) a.best_w[i] = rw[i];
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
    // workgroup 0 writes its replica back (nobody reads a.w / a.m / a.v after the start of the launch)
    if (wg != 0) return;
    __syncthreads();
    const bool stopped = ctl[0] != 0;
    const bool restore = stopped || (a.flags & NNEST_TRAIN_FINALIZE);   // netG.load_state_dict(best_model)  (trainer.py:241)
    for (int i = threadIdx.x; i < np; i += blockDim.x) {
        a.w[i] = restore ? a.best_w[i] : rw[i];
        a.m[i] = rm[i];
        a.v[i] = rv[i];
    }
    __syncthreads();
    rebuild_images(a);
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

// grid buffers behind the single-workgroup workspace
template <int NT, int NH, int L> struct GridSizes {
    static constexpr int NJOBS = NT * NH + L * NH * NH + NH * NT;
    static size_t stage(const FlowShape &s) { return (size_t)s.B * 2 * StageMap<NT, NH, L>::count * TRAIN_MAX_ROWS * 16; }
    static size_t tiles(const FlowShape &s) { return (size_t)2 * s.B * 2 * NJOBS * 256; }
};

template <int NT, int NH, int L>
static hipError_t launch_train_grid_t(TrainArgs a, float *gridws, hipStream_t st) {
    typedef GridSizes<NT, NH, L> GS;
    a.gstage = gridws;
    a.gtile = a.gstage + GS::stage(a.s);
    a.gpos = reinterpret_cast<int *>(a.gtile + GS::tiles(a.s));
    a.gpart = reinterpret_cast<float *>(a.gpos + a.s.num_params());
    a.gsync = reinterpret_cast<unsigned int *>(a.gpart + 64);
    a.gerr = reinterpret_cast<int *>(a.gsync + 8);
    a.grep = a.gpart + 64 + 16 + ((4 - ((a.s.num_params() + 64 + 16) & 3)) & 3);   // 16-byte aligned behind the small words
    hipError_t e = hipMemsetAsync(a.gpos, 0xFF, (size_t)a.s.num_params() * sizeof(int), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.gpart, 0, (64 + 16) * sizeof(float), st);  // partial sums, the barrier counter, the error word
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((grid_gpos_kernel<NT, NH, L>), dim3(32), dim3(256), 0, st, a.gpos, a.s);
    size_t lds = 2 * (size_t)a.s.image_floats * sizeof(float);
    const size_t pos = 2 * (size_t)((a.s.num_params() + 3) & ~3) * sizeof(int);
    if (lds + pos + 2 * NT * 1024 <= 160 * 1024 - 512) {   // the two position maps beside the images (and the exchange buffer)
        lds += pos;
        a.flags |= NNEST_TRAIN_POS_LDS;
    }
    if (lds > 64 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(train_kernel_grid<NT, NH, L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((train_kernel_grid<NT, NH, L>), dim3(GRID_WG), dim3(TRAIN_THREADS), lds, st, a);
    return hipGetLastError();
}

// grid form: the epoch loop of an affine flow whose two fragment images fit one CU's LDS; everything else runs train_kernel
static bool grid_eligible(const TrainArgs &a) {
    return a.mode == TRAIN_MODE_EPOCHS && a.s.scale_mode == 0 && !(a.flags & NNEST_TRAIN_ONE_CU) &&
           2 * (size_t)a.s.image_floats * sizeof(float) <= 160 * 1024 - 1024 && a.s.NH == 1 && a.s.L <= 2;
}

static size_t grid_workspace_floats(const FlowShape &s) {
    const int CT = 2 * s.NT + 2 * (s.L + 1) * s.NH, NJOBS = 2 * s.NT * s.NH + s.L * s.NH * s.NH;
    return (size_t)s.B * 2 * CT * TRAIN_MAX_ROWS * 16 + (size_t)2 * s.B * 2 * NJOBS * 256 + (size_t)s.num_params() + 64 + 16 + 64 +
           (size_t)GRID_WG * 3 * ((s.num_params() + 3) & ~3);
}

static hipError_t dispatch_train_grid(const TrainArgs &a, float *gridws, hipStream_t st) {
    const FlowShape &s = a.s;
#define TRY_GRID(nt, l) if (s.NT == nt && s.NH == 1 && s.L == l) return launch_train_grid_t<nt, 1, l>(a, gridws, st)
    TRY_GRID(1, 0); TRY_GRID(2, 0); TRY_GRID(3, 0); TRY_GRID(4, 0);
    TRY_GRID(1, 1); TRY_GRID(2, 1); TRY_GRID(3, 1); TRY_GRID(4, 1);
    TRY_GRID(1, 2); TRY_GRID(2, 2); TRY_GRID(3, 2); TRY_GRID(4, 2);
#undef TRY_GRID
    return hipErrorInvalidConfiguration;
}
