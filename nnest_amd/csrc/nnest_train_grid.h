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
//   W+A  the 30 weight-gradient jobs, one per wave across the 64 waves of the grid (job J on wave J / 8 of workgroup J % 8), each contracting over all rows exactly
//        as the single-workgroup kernel does (same operand order: the same bits) -- and the wave that produced a gradient tile
//        OWNS that tile's parameters: their weights and Adam moments live in its registers for the whole launch (the job ->
//        wave map is static), it applies Adam right there and publishes the NEW WEIGHTS (round 3; round 2 published the
//        gradients and had every workgroup step a private replica of the whole vector: 17.7 k of a minibatch's 57 k cycles).
//        The waves without a job step the parameters no job reaches (zero gradient: weight decay only) in global memory;
//   grid barrier;
//   R    every workgroup scatters the published weights into its LDS copies of the two fragment images.
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

enum { GRID_WG = 8, GRID_MAX_POLLS = 1 << 22 };

// (the s_nop: a VMEM store of more than 8 bytes reads its data registers for up to two cycles after issue, and the compiler's
// hazard recognizer cannot see inside the asm -- without it the next VALU write corrupted some lanes of some stores)
__device__ __forceinline__ void st_sc1_f32x4(float *p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float ld_sc1(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a 16-byte sc1 load, ISSUED only: hipcc does not know it is a load, the caller drains (ld_drain) before using the value
__device__ __forceinline__ f32x4 ld_sc1_x4_issue(const float *p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc1(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all G workgroups have finished what precedes; `phase` counts barriers (same value in every thread of the grid).
// first_poll_sleep (units of 64 cycles): a poll is a round trip to the memory side (~3 k cycles, measured), and one issued right
// behind the arrival sees only the workgroups that arrived before it -- behind a phase whose workgroups finish ~1 k cycles apart
// the early ones miss and pay a second round trip (rows kernel: 2.0 polls per barrier); holding the first poll back by the
// expected spread makes it the only one.
__device__ __forceinline__ bool grid_barrier(unsigned int *ctr, int &phase, int G, int *err, int first_poll_sleep = 0) {
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
        if (first_poll_sleep >= 256) {   // 256 + n: n units of 256 cycles (the instruction takes an immediate)
            for (int i = 256; i < first_poll_sleep; ++i) __builtin_amdgcn_s_sleep(4);
        } else if (first_poll_sleep > 8) __builtin_amdgcn_s_sleep(16);
        else if (first_poll_sleep > 0) __builtin_amdgcn_s_sleep(8);
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

// Where job J's result tile sits in the two fragment images.  The tile a job produces -- lane (gq, j) register r = element (row
// 4 gq + r, column j) of a 16 x 16 block of a weight matrix -- IS the A-fragment of the BACKWARD image for that block (same lane /
// register mapping: bwd_image_src), and its transpose is the A-fragment of the FORWARD image (fwd_image_src).  So an owner
// publishes its new weights straight into two global images laid out like the LDS ones, and the refresh phase is a copy.
// Offsets in floats of the tile's [64 lanes][4]; bias: offset of the 16 floats its bias vector fills in the forward image, or -1.
template <int NT, int NH, int L>
__host__ __device__ inline void grid_job_image_offsets(const FlowShape &s, int J, int *off_f, int *off_b, int *off_bias) {
    constexpr int J_W3 = NT * NH, J_W2 = L * NH * NH, NJOBS = J_W3 + J_W2 + NH * NT;
    const int bn = J / NJOBS, base = bn * s.net_floats;
    int q = J % NJOBS;
    if (q < J_W3) {          // Wout block (tau, ht)
        const int tau = q / NH, ht = q % NH;
        *off_b = base + frag_off_L1() + (ht * NT + tau) * 256;
        *off_f = base + frag_off_L3(NT, NH, L) + (tau * NH + ht) * 256;
        *off_bias = ht == 0 ? base + frag_off_b3(NT, NH, L) + 16 * tau : -1;
        return;
    }
    q -= J_W3;
    if (q < J_W2) {          // hidden layer l, block (hto, hti)
        const int l = q / (NH * NH), hto = (q / NH) % NH, hti = q % NH;
        *off_b = base + frag_off_L2(NT, NH) + ((l * NH + hti) * NH + hto) * 256;
        *off_f = base + frag_off_L2(NT, NH) + ((l * NH + hto) * NH + hti) * 256;
        *off_bias = hti == 0 ? base + frag_off_b2(NT, NH, L) + l * 16 * NH + 16 * hto : -1;
        return;
    }
    q -= J_W2;               // W0 block (ht, tau)
    const int ht = q / NT, tau = q % NT;
    *off_b = base + frag_off_L3(NT, NH, L) + (tau * NH + ht) * 256;
    *off_f = base + frag_off_L1() + (ht * NT + tau) * 256;
    *off_bias = tau == 0 ? base + frag_off_b1(NT, NH, L) + 16 * ht : -1;
}
// the parameters no job produces a gradient for (the ones the alternating mask never reaches): a list, any order
__global__ void grid_dead_kernel(const int *__restrict__ gpos, int np, int *__restrict__ dead, int *__restrict__ count) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < np; p += gridDim.x * blockDim.x)
        if (gpos[p] < 0) dead[atomicAdd(count, 1)] = p;
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

// What a job's wave owns: the weight, exp_avg and exp_avg_sq of its result tile (lane (gq, j) register r) and of its bias vector,
// and their indices in the packed vector.  OREG (<= 2 tiles per class): in registers for the whole launch.  Otherwise (3-4 tiles per
// class: the row tile's forward / backward state alone takes the register file) in a private 128-byte record per lane in global
// memory, read at the head of the W phase beside the contraction's operands and written back behind it.
struct OwnState {
    float tw[4], tm[4], tv[4], bw[4], bm[4], bv[4];
    int wt[4], bt[4];
};
__device__ __forceinline__ void own_load(OwnState &o, const float *rec) {
    const f32x4 *q = reinterpret_cast<const f32x4 *>(rec);
    const f32x4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4], a5 = q[5], a6 = q[6], a7 = q[7];
    o.tw[0] = a0.x; o.tw[1] = a0.y; o.tw[2] = a0.z; o.tw[3] = a0.w;
    o.tm[0] = a1.x; o.tm[1] = a1.y; o.tm[2] = a1.z; o.tm[3] = a1.w;
    o.tv[0] = a2.x; o.tv[1] = a2.y; o.tv[2] = a2.z; o.tv[3] = a2.w;
    o.bw[0] = a3.x; o.bw[1] = a3.y; o.bw[2] = a3.z; o.bw[3] = a3.w;
    o.bm[0] = a4.x; o.bm[1] = a4.y; o.bm[2] = a4.z; o.bm[3] = a4.w;
    o.bv[0] = a5.x; o.bv[1] = a5.y; o.bv[2] = a5.z; o.bv[3] = a5.w;
    o.wt[0] = __float_as_int(a6.x); o.wt[1] = __float_as_int(a6.y); o.wt[2] = __float_as_int(a6.z); o.wt[3] = __float_as_int(a6.w);
    o.bt[0] = __float_as_int(a7.x); o.bt[1] = __float_as_int(a7.y); o.bt[2] = __float_as_int(a7.z); o.bt[3] = __float_as_int(a7.w);
}
__device__ __forceinline__ void own_store(const OwnState &o, float *rec, bool with_targets) {
    f32x4 *q = reinterpret_cast<f32x4 *>(rec);
    q[0] = (f32x4){o.tw[0], o.tw[1], o.tw[2], o.tw[3]};
    q[1] = (f32x4){o.tm[0], o.tm[1], o.tm[2], o.tm[3]};
    q[2] = (f32x4){o.tv[0], o.tv[1], o.tv[2], o.tv[3]};
    q[3] = (f32x4){o.bw[0], o.bw[1], o.bw[2], o.bw[3]};
    q[4] = (f32x4){o.bm[0], o.bm[1], o.bm[2], o.bm[3]};
    q[5] = (f32x4){o.bv[0], o.bv[1], o.bv[2], o.bv[3]};
    if (with_targets) {
        q[6] = (f32x4){__int_as_float(o.wt[0]), __int_as_float(o.wt[1]), __int_as_float(o.wt[2]), __int_as_float(o.wt[3])};
        q[7] = (f32x4){__int_as_float(o.bt[0]), __int_as_float(o.bt[1]), __int_as_float(o.bt[2]), __int_as_float(o.bt[3])};
    }
}

// data = X[perm] + jitter * randn (trainer.py:392) for the 16-row tile `wg` of minibatch (epoch, mb), in PIECES of four dimensions
// per lane -- piece q = (tile q / 2, half q % 2) is dims 32 t + 8 g + 4 hf + [0, 4) of the lane's row: one row load and one
// Philox block -- so that the free waves of a workgroup can prepare a minibatch's rows side by side.  out[c][k] (c = parity
// class, k = 0, 1) are components 2 hf + k of xs[c][t].  Padded rows / dims hold 0.
__device__ __forceinline__ void grid_rows_piece(const TrainArgs &a, int epoch, int mb, int wg, int lane, int q, float (&out)[2][2]) {
    const int w = lane & 15, g = lane >> 4, D = a.s.D, t = q >> 1, hf = q & 1;
    const int M = min(a.batch, a.n_train - mb * a.batch);
    const int row = wg * 16 + w;
    const bool row_ok = wg < ((M + 15) >> 4) && row < M;
    long src = 0;
    if (row_ok) src = a.perm[(size_t)epoch * a.n_train + mb * a.batch + row];
    const int d0 = 32 * t + 8 * g + 4 * hf;
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
            const f32x4 nn = noise_normal4(a.seed, (uint64_t)p, (uint32_t)(a.epoch_offset + epoch), (uint32_t)(8 * t + 2 * g + hf), NOISE_STREAM_JITTER);
            n[0] = nn.x; n[1] = nn.y; n[2] = nn.z; n[3] = nn.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (d0 + j < D) v[j] += n[j] * a.jitter;
    }
    out[0][0] = v[0]; out[1][0] = v[1]; out[0][1] = v[2]; out[1][1] = v[3];
}
// all pieces of a tile's rows, into registers (the workgroups whose waves all carry jobs)
template <int NT>
__device__ __forceinline__ void grid_rows(const TrainArgs &a, int epoch, int mb, int wg, int lane, f32x4 (&xs)[2][NT]) {
#pragma unroll
    for (int q = 0; q < 2 * NT; ++q) {
        float o[2][2];
        grid_rows_piece(a, epoch, mb, wg, lane, q, o);
        const int t = q >> 1;
        if (q & 1) { xs[0][t].z = o[0][0]; xs[0][t].w = o[0][1]; xs[1][t].z = o[1][0]; xs[1][t].w = o[1][1]; }
        else       { xs[0][t].x = o[0][0]; xs[0][t].y = o[0][1]; xs[1][t].x = o[1][0]; xs[1][t].y = o[1][1]; }
    }
}
// the pieces q = k, k + nfree, ... of minibatch (epoch, mb) into the LDS buffer [c][t][lane] of f32x4 (free wave k of nfree)
template <int NT>
__device__ __forceinline__ void grid_rows_to_lds(const TrainArgs &a, int epoch, int mb, int wg, int lane, int k, int nfree, f32x4 *buf) {
    for (int q = k; q < 2 * NT; q += nfree) {
        float o[2][2];
        grid_rows_piece(a, epoch, mb, wg, lane, q, o);
        const int t = q >> 1, hf = q & 1;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float *dst = reinterpret_cast<float *>(buf + (c * NT + t) * 64 + lane) + 2 * hf;
            dst[0] = o[c][0]; dst[1] = o[c][1];
        }
    }
}

// adam_one without the scatter into the images (the refresh phase does that from the published weights)
__device__ __forceinline__ void adam_reg(const TrainArgs &a, const AdamStep &ad, float &w, float g, float &m, float &v) {
    adam_one(a, ad, w, g, m, v, -1, -1, nullptr, nullptr);
}

template <int NT, int NH, int L>
__global__ void __launch_bounds__(TRAIN_THREADS) train_kernel_grid(TrainArgs a) {
    typedef StageMap<NT, NH, L> SM;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *imgf = smem;
    float *imgb = smem + a.s.image_floats;
    __shared__ f32x4 xch[2 * NT * 64];   // the two net waves' exchange buffer
    // the next minibatch's rows (permutation gather + Philox jitter: 6 k cycles of a 13 k forward pass when the net waves did it
    // themselves) are prepared by this workgroup's eight waves behind their W-phase work, a piece (four dimensions per lane) each
    __shared__ f32x4 xpre[2][2 * NT * 64];
    __shared__ int ctl[4];      // [0] stop flag, [1] counter, [2] best epoch
    __shared__ float ctlf[2];   // [0] best validation loss
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = lane & 15, g = lane >> 4;
    const int wg = blockIdx.x, G = gridDim.x;
    const int D = a.s.D, B = a.s.B;
    constexpr int NJOBS = NT * NH + L * NH * NH + NH * NT;
    const int NJ = B * 2 * NJOBS;
    f32x4 *stash_w = reinterpret_cast<f32x4 *>(a.stash) + (size_t)wg * B * 2 * (L + 1) * NH * 64;
    int phase = 0;

    rebuild_images_to(a, imgf, imgb);
    // ---- who owns what.  Wave (wg, wave) runs job Jmine = 8 wg + wave of every minibatch (NJ <= 64 jobs: grid_eligible) and keeps
    // that tile's parameters -- weight, exp_avg, exp_avg_sq of lane (gq, j) register r, and of the bias vector -- in registers
    // from here to the end of the launch.  The waves beyond the jobs share the parameters no job reaches.
    const int Jmine = wave * G + wg;   // jobs fill the low waves of EVERY workgroup: the high waves stay free (dead parameters, row prefetch)
    const bool owner = Jmine < NJ;
    constexpr bool OREG = NT <= 2;
    float *own_rec = a.gown + ((size_t)Jmine * 64 + lane) * 32;
    int off_f = 0, off_b = 0, off_bias = -1;
    if (owner) grid_job_image_offsets<NT, NH, L>(a.s, Jmine, &off_f, &off_b, &off_bias);
    __shared__ __attribute__((aligned(16))) float trs[TRAIN_WAVES][16 * 20];   // per wave: a 16 x 16 tile, rows 80 bytes apart
    OwnState os;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        os.wt[r] = os.bt[r] = -1;
        os.tw[r] = os.tm[r] = os.tv[r] = os.bw[r] = os.bm[r] = os.bv[r] = 0.f;
    }
    if (owner) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            grid_job_targets<NT, NH, L>(a.s, Jmine, lane, r, &os.wt[r], &os.bt[r]);
            if (os.wt[r] >= 0) { os.tw[r] = a.w[os.wt[r]]; os.tm[r] = a.m[os.wt[r]]; os.tv[r] = a.v[os.wt[r]]; }
            if (os.bt[r] >= 0) { os.bw[r] = a.w[os.bt[r]]; os.bm[r] = a.m[os.bt[r]]; os.bv[r] = a.v[os.bt[r]]; }
        }
        if (!OREG) own_store(os, own_rec, true);
    }
    // The parameters no job reaches, shared out over ALL waves of the grid (behind their job) in chunks of whole 256-byte rows of a COMPACT private
    // array (gdst: [w | exp_avg | exp_avg_sq][ndead_pad], indexed by position in the list): a wave's entries share no cache line
    // with another workgroup's -- the XCDs' L2s are not coherent with each other, and packed neighbours a.w[p], a.w[p + 1] may
    // belong to waves on different XCDs.  The packed vectors are written once, at the end, with write-through stores.
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
    __syncthreads();

    const int n_mb = (a.n_train + a.batch - 1) / a.batch;
    if (a.max_epochs > 0) grid_rows_to_lds<NT>(a, 0, 0, wg, lane, TRAIN_WAVES - 1 - wave, TRAIN_WAVES, xpre[0]);   // the first minibatch's rows
    __syncthreads();
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
                    // data = X[perm] + jitter * randn  (trainer.py:392): prepared a phase ahead by the last wave, or here
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int t = 0; t < NT; ++t) xs[c][t] = xpre[mbcount & 1][(c * NT + t) * 64 + lane];
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
            // ---- W + A: one weight-gradient job per wave of the grid; Adam on the tile's parameters in this wave's registers ----
            if (owner) {
                const int J = Jmine;
                if (!OREG) own_load(os, own_rec);   // (requested beside the contraction's operands)
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
                const float gt[4] = {t.x, t.y, t.z, t.w}, gb[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (os.wt[r] >= 0) adam_reg(a, ad, os.tw[r], gt[r], os.tm[r], os.tv[r]);
                    if (os.bt[r] >= 0) adam_reg(a, ad, os.bw[r], gb[r], os.bm[r], os.bv[r]);
                }
                // the NEW weights, one whole 1-KB fragment tile per store instruction (sc1: the other workgroups read them): as they
                // stand into the published backward image, transposed (through this wave's LDS scratch) into the forward image
                st_sc1_f32x4(a.gimgb + off_b + lane * 4, (f32x4){os.tw[0], os.tw[1], os.tw[2], os.tw[3]});
                {
                    float *tr = trs[wave];
                    const int gq = lane >> 4, j = lane & 15;
#pragma unroll
                    for (int r = 0; r < 4; ++r) tr[(4 * gq + r) * 20 + j] = os.tw[r];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its own LDS traffic, in order)
                    const f32x4 tt = *reinterpret_cast<const f32x4 *>(tr + j * 20 + 4 * gq);   // lane (g, i): T[i][4 g .. 4 g + 3]
                    st_sc1_f32x4(a.gimgf + off_f + lane * 4, tt);
                }
                if (off_bias >= 0 && (lane & 15) == 0)
                    st_sc1_f32x4(a.gimgf + off_bias + (lane >> 4) * 4, (f32x4){os.bw[0], os.bw[1], os.bw[2], os.bw[3]});
                if (!OREG) own_store(os, own_rec, false);
            }
            // every wave: its share of the parameters no job reaches -- zero gradient, weight decay only (they take their Adam step
            // too, as in torch) -- ...
            for (int k = dead0 + lane; k < dead1; k += 64) {
                float w_ = dw[k], m_ = dm[k], v_ = dv[k];
                adam_reg(a, ad, w_, 0.f, m_, v_);
                dw[k] = w_; dm[k] = m_; dv[k] = v_;
            }
            {   // ... and its pieces of the NEXT minibatch's rows (the job-less high waves first)
                int e2 = epoch, m2 = mb + 1;
                if (m2 == n_mb) { m2 = 0; e2 = epoch + 1; }
                if (e2 < a.max_epochs) grid_rows_to_lds<NT>(a, e2, m2, wg, lane, TRAIN_WAVES - 1 - wave, TRAIN_WAVES, xpre[(mbcount + 1) & 1]);
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
            // ---- R: the published weights into this workgroup's LDS images.  Every load -- the tile values (sc1) and the slot maps
            // -- is requested before the first LDS write: one round trip to the memory side for the whole phase
            {
                // both published images are laid out like the LDS ones: a straight copy, 16 bytes per lane and load, every load
                // requested before the first LDS write (one round trip to the memory side for the whole phase)
                constexpr int RU = 4;
                const int n4 = a.s.image_floats >> 2;
                for (int i0 = threadIdx.x; i0 < n4; i0 += RU * blockDim.x) {
                    f32x4 vf[RU], vb[RU];
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int i = min(i0 + u * (int)blockDim.x, n4 - 1);
                        vf[u] = ld_sc1_x4_issue(a.gimgf + 4 * (size_t)i);
                        vb[u] = ld_sc1_x4_issue(a.gimgb + 4 * (size_t)i);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(vb[0]), "+v"(vb[1]), "+v"(vb[2]), "+v"(vb[3]) : : "memory");
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int i = i0 + u * blockDim.x;
                        if (i < n4) {
                            reinterpret_cast<f32x4 *>(imgf)[i] = vf[u];
                            reinterpret_cast<f32x4 *>(imgb)[i] = vb[u];
                        }
                    }
                }
            }
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
        if (improved) {   // best_model = deepcopy(netG)  (trainer.py:208): every wave snapshots what it owns
            if (owner) {
                if (!OREG) own_load(os, own_rec);
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
    // every wave writes what it owns back -- write-through stores, one writer per entry (nobody reads the packed vectors between
    // the start of the launch and the kernel that follows: launch_repack rebuilds the forward image from a.w)
    __syncthreads();
    const bool stopped = ctl[0] != 0;
    const bool restore = stopped || (a.flags & NNEST_TRAIN_FINALIZE);   // netG.load_state_dict(best_model)  (trainer.py:241)
    if (owner) {
        if (!OREG) own_load(os, own_rec);
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
    a.gndead = reinterpret_cast<int *>(a.gsync + 12);
    a.gdead = reinterpret_cast<int *>(a.gpart + 64 + 16);
    a.gown = reinterpret_cast<float *>(a.gdead + a.s.num_params());
    a.gown += (64 - ((size_t)(a.gown - gridws) & 63)) & 63;   // 256-byte aligned records
    a.gdst = a.gown + (size_t)GRID_WG * TRAIN_WAVES * 64 * 32;
    a.gimgf = a.gdst + (size_t)3 * ((a.s.num_params() + 63) & ~63) + 64;
    a.gimgf += (64 - ((size_t)(a.gimgf - gridws) & 63)) & 63;
    a.gimgb = a.gimgf + ((a.s.image_floats + 63) & ~63);
    hipError_t e = hipMemsetAsync(a.gpos, 0xFF, (size_t)a.s.num_params() * sizeof(int), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.gimgf, 0, (size_t)2 * ((a.s.image_floats + 63) & ~63) * sizeof(float), st);   // (the backward image's unused bias area)
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.gpart, 0, (64 + 16) * sizeof(float), st);  // partial sums, the barrier counter, the error word, the dead count
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((grid_gpos_kernel<NT, NH, L>), dim3(32), dim3(256), 0, st, a.gpos, a.s);
    hipLaunchKernelGGL(grid_dead_kernel, dim3(32), dim3(256), 0, st, a.gpos, a.s.num_params(), a.gdead, a.gndead);
    size_t lds = 2 * (size_t)a.s.image_floats * sizeof(float);
    if (lds > 64 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(train_kernel_grid<NT, NH, L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((train_kernel_grid<NT, NH, L>), dim3(GRID_WG), dim3(TRAIN_THREADS), lds, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_repack(a.w, a.img_fwd, a.s, st);   // the inference kernels' forward image from the weights the launch leaves
}

// grid form: the epoch loop of an affine flow whose two fragment images fit one CU's LDS; everything else runs train_kernel
static bool grid_eligible(const TrainArgs &a) {
    return a.mode == TRAIN_MODE_EPOCHS && a.s.scale_mode == 0 && !(a.flags & NNEST_TRAIN_ONE_CU) &&
           2 * (size_t)a.s.image_floats * sizeof(float) <= 160 * 1024 - 1024 && a.s.NH == 1 && a.s.L <= 2 &&
           a.s.B * 2 * (2 * a.s.NT + a.s.L) <= GRID_WG * TRAIN_WAVES;   // one job per wave: the owner of its parameters
}

static size_t grid_workspace_floats(const FlowShape &s) {
    const int CT = 2 * s.NT + 2 * (s.L + 1) * s.NH, NJOBS = 2 * s.NT * s.NH + s.L * s.NH * s.NH;
    return (size_t)s.B * 2 * CT * TRAIN_MAX_ROWS * 16 + (size_t)2 * s.B * 2 * NJOBS * 256 + (size_t)s.num_params() + 64 + 16 + 64 +
           (size_t)s.num_params() /* dead list */ +
           (size_t)GRID_WG * TRAIN_WAVES * 64 * 32 + 64 /* owners' records */ + (size_t)3 * (s.num_params() + 64) + 128 /* dead state */ +
           (size_t)2 * (s.image_floats + 64) + 64 /* the published images */ +
           (size_t)2 * s.B * 2 * CT * TRAIN_MAX_ROWS * 16 + 192 /* train_kernel_pipe's tagged staging area */;
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
