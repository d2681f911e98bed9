// nnest_train_pipe.h -- K5, the rows form PIPELINED PER COUPLING BLOCK (round 5; included behind nnest_train_rows.h, inside
// namespace nnest).
//
// train_kernel_rows runs a minibatch as  F+B (all row waves) -> grid barrier -> weight-gradient jobs + Adam + publish -> refresh
// of the two LDS images:  5.7 k cycles of arithmetic and 19.9 k cycles of cross-CU hand-offs, all of them exposed, because the one
// barrier sits behind the WHOLE backward pass and every wave of the grid takes part in every phase (profiles/r04/k5_stamps.txt).
// But the backward pass leaves block 2 first, then 1, then 0, and a block's weights depend on that block's gradients only
// (trainer.py:396-398: one optimizer step over independent tensors).  Here a workgroup's waves have fixed roles and nothing in the
// minibatch loop is a workgroup barrier or a grid barrier:
//   row waves (0..3)      forward and backward of one row each (rows_block_forward / rows_block_backward, unchanged arithmetic);
//                         behind a block's backward stores they ARRIVE on that block's counter (the wave's counted s_waitcnt, an
//                         LDS count of the workgroup's four waves, the last one adds to the grid's counter) and go on;
//   service waves (4..7)  prepare the next minibatch's rows; the owner of a weight-gradient job waits for ITS block's counter,
//                         contracts, applies Adam in its registers and publishes the new tile as {weight, tag} granules; then
//                         every service wave re-lays the published tiles into the workgroup's two solo images BLOCK BY BLOCK
//                         (2, 1, 0 -- the order the backward pass frees them) and counts the block refreshed in LDS;
//   the next forward pass waits, block by block, for that LDS count.
// So blocks 2 and 1 are contracted, stepped, published and refreshed while the row waves are still in blocks 1 and 0; only block 0's
// chain -- a third of the bytes -- stays on the critical path.  The forward-pass quantities a job needs (activations, the
// conditioning input) are staged by the FORWARD pass, which halves the stores in front of the last arrival.
// Same jobs, same operand order, same Adam, same publish format as train_kernel_rows; the one difference in value is that the staged
// conditioning input is the forward pass's own (train_kernel_rows stages the one its backward pass recovered by inverting the
// blocks above: equal to rounding).  Every reduction keeps a fixed order: bitwise reproducible run to run.
//
// Cross-CU visibility (MI355X_MICROARCH.md, "Valid forms"): all handed-off bytes are sc1 stores and sc1 loads to registers; a
// counter add comes after the s_waitcnt of every wave it signals for (a wave's stores complete in order: only stores are in flight
// in a row wave); the wave that polled a counter loads only after its poll matched; published weights are tagged granules (R2).
// Every wait is bounded; one that runs out sets the error word and the launch ends with result.stopped = 2.

// diagnostic build: absolute time stamps of one minibatch (workgroup 0: all its waves read the same clock), a.losses[16 + k]
#ifdef NNEST_STAMP
#define TLINE(k) do { if (mbcount == 20 && wg == 0 && lane == 0) { __builtin_amdgcn_sched_barrier(0); tline[k] = (float)(__builtin_amdgcn_s_memtime() & 0xffffffull); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define TLINE(k) do { } while (0)
#endif

enum { PIPE_SVC = 4, SF_REF = 0, SF_X = 3, SF_ARR = 4, SF_ABORT = 7, SF_N = 8 };

static __device__ __forceinline__ int pipe_lds_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// spin on a word of the workgroup's LDS until it reaches `want`; false: aborted or ran out
static __device__ __forceinline__ bool pipe_wait_lds(const int *flag, int want, const int *abort_w) {
    bool ok = true;
    for (int polls = 0; pipe_lds_ld(flag) < want; ++polls) {
        if (polls > GRID_MAX_POLLS || pipe_lds_ld(abort_w)) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    return ok;
}
// spin on a counter of the grid (sc1 loads)
static __device__ __forceinline__ bool pipe_wait_ctr(const unsigned int *ctr, unsigned int want, const int *abort_w, const int *gerr) {
    bool ok = true;
    for (int polls = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++polls) {
        if (polls > GRID_MAX_POLLS || ((polls & 255) == 255 && (pipe_lds_ld(abort_w) || __hip_atomic_load(gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))) {
            ok = false;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    return ok;
}

// hipcc hoists loop-invariant ADDRESS arithmetic out of the epoch loop and then spills it (a register pair per load of the refresh,
// per operand base of a job, per owned parameter): each such load then waits for a scratch reload first.  Passing the base through an
// empty asm inside the loop keeps the arithmetic where it is used.
template <class T> static __device__ __forceinline__ T *pipe_opaque(T *p) { asm volatile("" : "+s"(p)); return p; }
static __device__ __forceinline__ int pipe_opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// what the forward pass stages for the block's weight-gradient jobs: the two hidden activations and the conditioning input
template <int U>
static __device__ __forceinline__ void rows_stage_forward(const RowsTagged &t, bool row_ok, const float (&cond)[U], float h1, float h2, bool stager) {
    typedef StageMap<U, 1, 1> SM;
    rows_stage_t(t, SM::act(1, 0), h2, stager);
    rows_stage_t(t, SM::act(0, 0), h1, stager);
    float cm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cm[u] = row_ok ? cond[u] : 0.f;
    rows_stage_slots_t<U>(t, SM::m(0), cm, stager);
}

// contract_rows_grid over the TAGGED staging area: every operand as an 8-byte {value, tag} granule (sc1 buffer loads, all requested
// before the first use); `fresh` comes back false when a granule of a row below rows_live (the rows some wave runs) did not carry tag `want` yet (the caller
// asks again).  Same operand order as contract_rows_grid: the same bits.
template <bool WITH_BIAS>
__device__ __forceinline__ f32x4 contract_rows_tagged(__amdgpu_buffer_rsrc_t stgr, int base_g, int base_a, int rows_pad, int rows_live, int lane, int want,
                                                      f32x4 &bias, bool &fresh) {
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 a0 = zero4, a1 = zero4, a2 = zero4, a3 = zero4;
    float bs = 0.f;
    constexpr int NIT = TRAIN_MAX_ROWS / 16;
    const int lo = (lane >> 4) * 256 + (lane & 15) * 16;   // (gq, j) inside a row group of granules (rows_tagged_row_off)
    u32x4_ gv[NIT][2], ev[NIT][2];   // [it][m]: {value of MFMA 2 m, tag, value of MFMA 2 m + 1, tag}
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int r = it * 16 < rows_pad ? it * 16 : 0;   // (rows beyond rows_pad hold an earlier minibatch's data: tile 0 again, products skipped)
        // (one vector register of offsets for all 56 requests: the row group's base goes in the instruction's scalar offset, the MFMA pair in its immediate)
        const int sg = __builtin_amdgcn_readfirstlane(base_g + r * 128), sa = __builtin_amdgcn_readfirstlane(base_a + r * 128);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            gv[it][m] = __builtin_amdgcn_raw_buffer_load_b128(stgr, lo + 1024 * m, sg, 16 /* sc1 */);
            ev[it][m] = __builtin_amdgcn_raw_buffer_load_b128(stgr, lo + 1024 * m, sa, 16);
        }
    }
    bool ok = true;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (it * 16 < rows_pad) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {   // (rows of the last tile that no wave runs -- the grid has ceil(batch / 4) workgroups -- are never written: zeros without a tag)
                ok = ok && (it * 16 + 8 * m + (lane >> 4) >= rows_live || ((int)gv[it][m].y == want && (int)ev[it][m].y == want));
                ok = ok && (it * 16 + 8 * m + 4 + (lane >> 4) >= rows_live || ((int)gv[it][m].w == want && (int)ev[it][m].w == want));
            }
            const float g0 = __uint_as_float(gv[it][0].x), g1 = __uint_as_float(gv[it][0].z), g2 = __uint_as_float(gv[it][1].x), g3 = __uint_as_float(gv[it][1].z);
            a0 = mfma4(g0, __uint_as_float(ev[it][0].x), a0);
            a1 = mfma4(g1, __uint_as_float(ev[it][0].z), a1);
            a2 = mfma4(g2, __uint_as_float(ev[it][1].x), a2);
            a3 = mfma4(g3, __uint_as_float(ev[it][1].z), a3);
            if (WITH_BIAS) bs += (g0 + g1) + (g2 + g3);
        }
    }
    fresh = __all(ok);
    if (WITH_BIAS) {
        bs += __shfl_xor(bs, 16);
        bs += __shfl_xor(bs, 32);
        const int q4 = (lane >> 4) * 4;
        bias = (f32x4){__shfl(bs, q4 + 0), __shfl(bs, q4 + 1), __shfl(bs, q4 + 2), __shfl(bs, q4 + 3)};
    }
    return (a0 + a1) + (a2 + a3);
}

// the refresh map of train_kernel_pipe: for element i of the published backward fragment image (weights as bwd_image_src lays them
// out, biases where the forward image keeps them) its two destinations in a workgroup's LDS as FLOAT offsets into smem, packed in
// one word: bits 0..14 the forward solo image, bits 15..29 the transposed one (IMG + offset), bit 31 = stored times the tanh
// prescale.  An element without a destination on a side points at one of 64 dummy floats behind the images (2 IMG + lane-distinct
// index), so that the re-lay is two unconditional ds_write_b32 per element.
template <int U>
__global__ void pipe_maps_kernel(unsigned int *__restrict__ maps, FlowShape s) {
    constexpr int IMG = ROWS_B * SOLO4_NF * 64;
    static_assert(2 * IMG + 64 <= (1 << 15), "refresh map packing");
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < s.image_floats; i += gridDim.x * blockDim.x) {
        const unsigned int dummy = 2 * IMG + ((i >> 2) & 63);
        unsigned int mf = dummy, mb = dummy, sc_bit = 0;
        const int o = i % s.net_floats;
        const int p = o >= frag_off_b1(U, 1, 1) ? fwd_image_src(s, i) : bwd_image_src(s, i);
        if (p >= 0) {
            const int bn = p / s.net_params;
            int df, db; bool sc;
            rows_param_dest<U>(s.D, bn >> 1, bn & 1, p - bn * s.net_params, df, db, sc);
            if (df >= 0) { mf = df; sc_bit = sc ? 0x80000000u : 0u; }
            if (db >= 0) mb = IMG + db;
        }
        maps[i] = mf | (mb << 15) | sc_bit;
    }
}

// The two kinds of wave run two FUNCTIONS (pipe_body<U, 0> the row waves, pipe_body<U, 1> the service waves), not two branches of
// one: as one function hipcc's register allocation sees 11 k instructions with the service waves' 112 operand registers and the row
// waves' kept activations in one interference graph and spills in both (the row waves' activations went to scratch with 100 of 256
// registers in use); as two, each gets an allocation of its own.  The kernel's arguments are read from the kernarg segment where
// they are needed, everything the waves share sits in the workgroup's dynamic LDS: the two images, the re-lay's dummy floats, the
// prepared rows, the control words.
template <int U> struct PipeLds {
    static constexpr int IMG = ROWS_B * SOLO4_NF * 64;
    static constexpr int XPRE = 2 * IMG + 64;                       // float offsets into the dynamic LDS
    static constexpr int CTL = XPRE + 2 * ROWS_PER_WG * 32 * U;     // int [4]: [0] stop flag, [1] counter, [2] best epoch
    static constexpr int CTLF = CTL + 4;                            // float [2]: [0] best validation loss, [1] the epoch's training loss (workgroup 0)
    static constexpr int SFLAG = CTLF + 4;                          // int [SF_N]
    static constexpr int TLINE = SFLAG + 8;                         // float [24] (diagnostic build)
    static constexpr int FLOATS = TLINE + 24;
};

template <int U, int ROLE>
__device__ __attribute__((noinline)) void pipe_body() {
    TrainArgs a;   // the kernel's one argument, word by word from the kernarg segment (scalar loads where they are used)
    {
        static_assert(sizeof(TrainArgs) % 4 == 0, "TrainArgs in words");
        // (a callable function is handed the IMPLICIT-argument pointer, not the kernarg segment's: the hidden arguments start at
        // the first 8-byte boundary behind the kernel's explicit ones, i.e. behind its one TrainArgs)
        const __attribute__((address_space(4))) unsigned int *kw = (const __attribute__((address_space(4))) unsigned int *)(
            (const __attribute__((address_space(4))) char *)__builtin_amdgcn_implicitarg_ptr() - ((sizeof(TrainArgs) + 7) & ~(size_t)7));
        unsigned int *aw = reinterpret_cast<unsigned int *>(&a);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(TrainArgs) / 4); ++i) aw[i] = kw[i];
    }
    typedef StageMap<U, 1, 1> SM;
    typedef PipeLds<U> PL;
    constexpr int B = ROWS_B, IMG = ROWS_B * SOLO4_NF * 64;
    constexpr int NPW = (ROWS_PER_WG * 8 * U + 63) / 64;   // service waves that prepare rows
    constexpr int IMGF_ = ROWS_B * 2 * (2 * U * 256 + 256 + 16 + 16 + 16 * U);   // image_floats of the shape
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *imgf = smem;          // forward solo image  [B][SOLO4_NF / 4][64][4]
    float *imgb = smem + IMG;    // transposed solo image
    float (*xpre)[ROWS_PER_WG * 32 * U] = reinterpret_cast<float (*)[ROWS_PER_WG * 32 * U]>(smem + PL::XPRE);   // the minibatch's rows, flat, a minibatch ahead
    int *ctl = reinterpret_cast<int *>(smem + PL::CTL);
    float *ctlf = smem + PL::CTLF;
#ifdef NNEST_STAMP
    float *tline = smem + PL::TLINE;
#endif
    int *sflag = reinterpret_cast<int *>(smem + PL::SFLAG);   // [SF_REF + b] waves that have refreshed block b (monotonic), [SF_X] row preparations done,
                                                              // [SF_ARR + b] row waves arrived behind block b's backward stores, [SF_ABORT]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wg = blockIdx.x, G = gridDim.x;
    const int D = a.s.D;
    constexpr int NJOBS = 2 * U + 1;
    constexpr bool row_wave = ROLE == 0;
    int phase = 0;
    // Who re-lays the published tiles into the LDS images: the four ROW waves (behind their last arrival they have nothing to do
    // until the images are refreshed), the service waves without a job, and the owners of block 2's jobs (the first block the
    // backward pass frees: its owners are done before the other blocks' tiles exist).  The owners of block 1's and block 0's jobs
    // do nothing else -- block 0's job is the critical path of a minibatch.  Jobs are numbered block 0 first and dealt
    // service-wave-major (job J on service wave J / G of workgroup J % G), so at least two service waves of a workgroup re-lay.
    const int NJ = B * 2 * NJOBS;
    auto relays = [&](int sv) { const int J = sv * G + wg; return J >= NJ || J / (2 * NJOBS) == B - 1; };
    int NREL = ROWS_PER_WG, ridx0 = row_wave ? wave : ROWS_PER_WG;   // re-laying waves of this workgroup; this wave's rank among them
    for (int sv = 0; sv < PIPE_SVC; ++sv) {
        if (relays(sv)) { if (!row_wave && sv < wave - ROWS_PER_WG) ridx0 += 1; NREL += 1; }
    }
    const bool relayer = row_wave || relays(wave - ROWS_PER_WG);

    // ---- the two solo images from the packed weights ----
    for (int i = threadIdx.x; i < 2 * IMG; i += blockDim.x) smem[i] = 0.f;
    if (threadIdx.x < SF_N) sflag[threadIdx.x] = threadIdx.x == SF_X ? NPW : 0;
    __syncthreads();
    for (int p = threadIdx.x; p < a.s.nets_params(); p += blockDim.x) {
        const int bn = p / a.s.net_params, o = p - bn * a.s.net_params;
        int df, db; bool sc;
        rows_param_dest<U>(D, bn >> 1, bn & 1, o, df, db, sc);
        const float v = a.w[p];
        if (df >= 0) imgf[df] = sc ? SOLO_TANH_PRESCALE * v : v;
        if (db >= 0) imgb[db] = v;
    }
    const bool resume = (a.flags & NNEST_TRAIN_RESUME) != 0;
    if (threadIdx.x == 0) {
        ctl[0] = 0;
        ctl[1] = resume ? a.result->counter : 0;
        ctl[2] = resume ? a.result->best_epoch : 0;
        ctlf[0] = resume ? a.result->best_validation_loss : INFINITY;
        ctlf[1] = 0.f;
    }
    const int n_mb = (a.n_train + a.batch - 1) / a.batch;
    if (a.max_epochs > 0) rows_prepare<U>(a, 0, 0, wg, xpre[0]);
    __syncthreads();

    float *part_base = a.gtile;   // [3][128] {log p of a minibatch's row, tag} granules, then [2][128] validation sums of the row waves (sc1 words, one writer each)
    float *valid_base = a.gtile + 3 * 2 * TRAIN_MAX_ROWS;
    unsigned int *arrival = reinterpret_cast<unsigned int *>(a.gtile + 1024);   // [b] at + 32 b: workgroups whose rows are behind block b's backward stores
    const int *abort_w = &sflag[SF_ABORT];
    // the sum of the 4 G row waves' words in a fixed order: lane l takes words l and l + 64, then a butterfly over the lanes
    auto sum_rows = [&](const float *words) {
        float v = (lane < ROWS_PER_WG * G ? ld_sc1(words + lane) : 0.f) + (lane + 64 < ROWS_PER_WG * G ? ld_sc1(words + lane + 64) : 0.f);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
        return v;
    };
    unsigned long long ph[4] = {0, 0, 0, 0}, q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0;
    (void)ph; (void)q0; (void)q1; (void)q2; (void)q3; (void)q4;
    int epochs_run = 0, mbcount = 0;
    float last_train_loss = 0.f;
    bool alive = true;
    // a wait ran out (or another wave's did): every wave of the workgroup leaves its loops, the launch ends.  The error word says
    // which wait (code), of which wave of which workgroup, in which minibatch (the first one to fail writes it)
    auto fail = [&](int code) {
        alive = false;
        if (lane == 0) {
            sflag[SF_ABORT] = 1;
            atomicCAS(a.gerr, 0, code | (wave << 4) | (wg << 8) | ((mbcount & 0xffff) << 16));
        }
    };
    // What both kinds of wave do at the end of an epoch, barrier for barrier (the two kinds run different code: a workgroup barrier
    // counts waves, not program counters).  validate(): the role's share of Trainer._validate; snapshot(): the role's share of
    // best_model = deepcopy(netG).  Returns false when the epoch loop ends.
    auto epoch_end = [&](int epoch, auto &&validate, auto &&snapshot) -> bool {
        __syncthreads();
        alive = alive && pipe_lds_ld(abort_w) == 0;
        if (alive) validate();
        __syncthreads();
        alive = alive && pipe_lds_ld(abort_w) == 0;
        if (!alive) return false;
        alive = grid_barrier(a.gsync, phase, G, a.gerr);
        if (!alive) return false;
        const float vtot = sum_rows(valid_base + (epoch & 1) * TRAIN_MAX_ROWS);
        const float valid_loss = (-vtot / (float)a.n_valid) / (float)a.n_valid;  // mean, then / len(dataset)  :418
        const float train_loss = ctlf[1];
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
        if (improved) {
            snapshot();
            if (threadIdx.x == 0) { ctlf[0] = valid_loss; ctl[2] = a.epoch_offset + epoch + 1; ctl[1] = 0; }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            ctl[1] += 1;
            if (ctl[1] > a.patience) ctl[0] = 1;
        }
        __syncthreads();
        return ctl[0] == 0;
    };

    // ---- the re-lay.  The owners publish their new weights as tiles of the backward FRAGMENT image (+ the biases in its forward-image
    // bias area): element i of that image is parameter src(i), whose places in the two solo images pipe_maps_kernel has packed into
    // one word.  Block b's part of the image is elements [2 b net_floats, 2 (b + 1) net_floats): QB quads of four; the workgroup's
    // NREL re-laying waves share them out, thread ridx taking quads ridx + 64 NREL u, whose map words it keeps in registers.
    constexpr int NETF = 2 * U * 256 + 256 + 16 + 16 + 16 * U;   // net_floats of the shape (flow_tile.h frag_net_floats)
    constexpr int QB = NETF / 2;
    constexpr int RU = (QB + 64 * (ROWS_PER_WG + 2) - 1) / (64 * (ROWS_PER_WG + 2));   // rounds of a thread per block with six re-laying waves
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int NRT = 64 * NREL, ridx = 64 * ridx0 + lane;
    const int sl_r = (int)a.gld_in & 255, sl_j = ((int)a.gld_in >> 8) & 255;   // hold-backs of block 0's data poll behind its hint / of a job's operand loads behind its block's arrivals, units of 64 cycles (launch_train_pipe_t)
    float *pub = a.gimgf;                    // the published weights: {weight, tag} pairs, 2 x image_floats floats
    unsigned int *hint = arrival + 32 * B;   // [b] at + 32 b: tiles of block b whose publish stores have been ISSUED (a hint, not an order: the tags decide)
    const __amdgpu_buffer_rsrc_t pubr = __builtin_amdgcn_make_buffer_rsrc(pub, 0, 2 * IMGF_ * (int)sizeof(float), 0x00020000);
    u32x4 mapw[B][RU];
    if (relayer) {
        const u32x4 *gmap = reinterpret_cast<const u32x4 *>(a.gown);
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int u = 0; u < RU; ++u) mapw[b][u] = gmap[b * QB + min(ridx + NRT * u, QB - 1)];
    }
    // a block's granules are requested (issue), later checked and written into the two images (finish); the three blocks' requests
    // overlap.  (Buffer loads with the sc1 bit: loads the compiler SEES -- it counts them in its own s_waitcnt and never copies a
    // destination register before the data has landed, which an asm load that is waited for elsewhere does not rule out; the offset
    // goes through an empty asm, so no two requests are the same load to the compiler and none leaves its loop.)
    auto issue = [&](int b, f32x4 (&va_)[RU], f32x4 (&vc_)[RU]) {
        const int rx = pipe_opaque(ridx);
#pragma unroll
        for (int u = 0; u < RU; ++u) {   // (rounds beyond the thread's last ask for the block's last quad again: every register is defined)
            const int qd = b * QB + min(rx + NRT * u, QB - 1);
            va_[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pubr, 32 * qd, 0, 16 /* sc1 */));
            vc_[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pubr, 32 * qd + 16, 0, 16));
        }
    };
    // true: every lane of the wave found its granules of the block fresh (tag `want`), and they are in the images now
    auto finish = [&](const u32x4 (&mp_)[RU], int want, const f32x4 (&va_)[RU], const f32x4 (&vc_)[RU]) {
        bool fresh = true;
#pragma unroll
        for (int u = 0; u < RU; ++u)
            fresh = fresh && __float_as_int(va_[u].y) == want && __float_as_int(va_[u].w) == want && __float_as_int(vc_[u].y) == want &&
                    __float_as_int(vc_[u].w) == want;
        if (!__all(fresh)) return false;
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            if (ridx + NRT * u < QB) {
                const float e4[4] = {va_[u].x, va_[u].z, vc_[u].x, vc_[u].z};
                const unsigned int m4[4] = {mp_[u].x, mp_[u].y, mp_[u].z, mp_[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned int mw = (unsigned int)pipe_opaque((int)m4[e]);   // (unpacked here, not hoisted into 2 registers + a mask per element)
                    const float es = (int)mw < 0 ? SOLO_TANH_PRESCALE * e4[e] : e4[e];
                    smem[mw & 0x7fff] = es;
                    smem[(mw >> 15) & 0x7fff] = e4[e];
                }
            }
        }
        return true;
    };
    auto complete = [&](int b, const u32x4 (&mp_)[RU], int want, f32x4 (&va_)[RU], f32x4 (&vc_)[RU]) {
        for (int polls = 0; !finish(mp_, want, va_, vc_); ++polls) {
            if (polls > GRID_MAX_POLLS / 64 || pipe_lds_ld(abort_w)) { fail(6); return; }
            __builtin_amdgcn_s_sleep(2);
            issue(b, va_, vc_);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&sflag[SF_REF + b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // The blocks in the order the backward pass frees them.  The cheap counter says when a block's tiles are on their way (every
    // owner adds to it right behind its publish stores, unordered: a hint); the tags say when they are there.  Block 1's request goes
    // out before block 2 is written, block 0's before block 1 is: the round trips overlap the writing.
    auto relay_all = [&](int want) {
        const unsigned int published = (unsigned int)(2 * NJOBS) * (unsigned int)want;
        f32x4 va[B][RU], vc[B][RU];
        if (!pipe_wait_ctr(hint + 32 * 2, published, abort_w, a.gerr)) { fail(5); return; }
        if (wave == 7) TLINE(12);
        issue(2, va[2], vc[2]);
        if (!pipe_wait_ctr(hint + 32 * 1, published, abort_w, a.gerr)) { fail(5); return; }
        if (wave == 7) TLINE(13);
        issue(1, va[1], vc[1]);
        complete(2, mapw[2], want, va[2], vc[2]);
        if (wave == 7) TLINE(15);
        if (!alive || !pipe_wait_ctr(hint + 32 * 0, published, abort_w, a.gerr)) { fail(5); return; }
        if (wave == 7) TLINE(14);
        for (int i = 0; i < sl_r; ++i) __builtin_amdgcn_s_sleep(1);
        issue(0, va[0], vc[0]);
        complete(1, mapw[1], want, va[1], vc[1]);
        if (wave == 7) TLINE(16);
        if (alive) complete(0, mapw[0], want, va[0], vc[0]);
    };

    if constexpr (ROLE == 0) {
        // =====================================================================================================================
        // a row of every minibatch: forward, backward, arrivals
        // =====================================================================================================================
        const int pos = lane & 15;
        const bool h1 = (lane & 16) != 0, translate_half = lane >= 32, stager = (lane & 16) == 0;
        const unsigned sel = translate_half ? 0xffffffffu : 0u;
        const int row = wg * ROWS_PER_WG + wave;
        constexpr size_t RS = (size_t)2 * 2 * SM::count * TRAIN_MAX_ROWS * 16;   // floats of a block's two staging regions (granules: 2 floats per value)
        const int voff_net = (translate_half ? SM::count * ROWS_CTB : 0) + rows_tagged_row_off(row);
        const int voff_pos = voff_net + 16 * pos, voff_slot = voff_net + ((U * pos) >> 4) * ROWS_CTB + 16 * ((U * pos) & 15);
        // the stores of block b's backward pass have been ISSUED -> count the wave in; the workgroup's last wave counts the workgroup
        // in.  Nothing waits for the stores: the count is a hint for the block's owners (when to look), the tags decide
        auto arrive = [&](int b) {
            if (lane == 0) {
                const int old = __hip_atomic_fetch_add(&sflag[SF_ARR + b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((old & (ROWS_PER_WG - 1)) == ROWS_PER_WG - 1) __hip_atomic_fetch_add(arrival + 32 * b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
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
        for (int epoch = 0; epoch < a.max_epochs; ++epoch) {
            for (int mb = 0; mb < n_mb && alive; ++mb, ++mbcount) {
                const int M = min(a.batch, a.n_train - mb * a.batch);
                const int rows_pad = ((M + 15) >> 4) * 16;
                float *part = part_base + (mbcount % 3) * 2 * TRAIN_MAX_ROWS;
                const float tg = __int_as_float(mbcount + 1);
                float *stg0 = pipe_opaque(a.gstage);
                const RowsTagged t0 = {stg0, voff_pos, voff_slot, tg}, t1 = {stg0 + RS, voff_pos, voff_slot, tg}, t2 = {stg0 + 2 * RS, voff_pos, voff_slot, tg};
                const bool row_ok = row < M;
                TSTAMP(q0);
                if (!pipe_wait_lds(&sflag[SF_X], NPW * (mbcount + 1), abort_w)) { fail(1); break; }
                if (row < rows_pad) {   // (rows M .. rows_pad - 1 run on zeros with row_ok = false: their staged gradients must read 0)
                    float xs[2][U], gs[2][U];
                    const float *xr = xpre[mbcount & 1] + wave * 32 * U + 2 * U * pos;
#pragma unroll
                    for (int u = 0; u < U; ++u) { xs[0][u] = xr[2 * u]; xs[1][u] = xr[2 * u + 1]; }
                    RowsKeep<U> kp[B];
                    // NormalizingFlow.forward (networks.py:24-32); a block's images must be the ones the last minibatch's weights were re-laid into
                    bool okw = pipe_wait_lds(&sflag[SF_REF + 0], NREL * mbcount, abort_w);
                    TSTAMP(q1);
                    if (wave == 0) TLINE(0);
                    float ld_lane = rows_block_forward<U>(Solo4Lds{imgf, lane}, sel, h1, xs[1], xs[0], kp[0]);
                    rows_stage_forward<U>(t0, row_ok, xs[1], kp[0].h1, kp[0].h2, stager);
                    okw = okw && pipe_wait_lds(&sflag[SF_REF + 1], NREL * mbcount, abort_w);
                    ld_lane += rows_block_forward<U>(Solo4Lds{imgf + SOLO4_NF * 64, lane}, sel, h1, xs[0], xs[1], kp[1]);
                    rows_stage_forward<U>(t1, row_ok, xs[0], kp[1].h1, kp[1].h2, stager);
                    okw = okw && pipe_wait_lds(&sflag[SF_REF + 2], NREL * mbcount, abort_w);
                    ld_lane += rows_block_forward<U>(Solo4Lds{imgf + 2 * SOLO4_NF * 64, lane}, sel, h1, xs[1], xs[0], kp[2]);
                    rows_stage_forward<U>(t2, row_ok, xs[1], kp[2].h1, kp[2].h2, stager);
                    if (!okw) { fail(2); break; }
                    const float lp = row_ok ? log_prob(xs, ld_lane) : 0.f;
                    if (lane == 0) st_sc1_x2(part + 2 * row, lp, tg);
                    TSTAMP(q2);
                    if (wave == 0) TLINE(1);
                    // d(loss)/du = dE/du / M ; d(loss)/d(logdet) = -1/M
                    const float invM = 1.0f / (float)M, gld = -invM;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int u = 0; u < U; ++u) gs[c][u] = row_ok ? base_dE(xs[c][u], a.s.base_beta) * invM : 0.f;
                    rows_block_backward<U, false, true>(Solo4Lds{imgb + 2 * SOLO4_NF * 64, lane}, sel, h1, translate_half, pos, D, 0, row_ok, gld, xs[1], xs[0], gs[1], gs[0], kp[2], nullptr, row, stager, t2);
                    arrive(2);
                    if (wave == 0) TLINE(2);
                    rows_block_backward<U, false, true>(Solo4Lds{imgb + SOLO4_NF * 64, lane}, sel, h1, translate_half, pos, D, 1, row_ok, gld, xs[0], xs[1], gs[0], gs[1], kp[1], nullptr, row, stager, t1);
                    arrive(1);
                    TSTAMP(q3);
                    if (wave == 0) TLINE(3);
                    rows_block_backward<U, false, true>(Solo4Lds{imgb, lane}, sel, h1, translate_half, pos, D, 0, row_ok, gld, xs[1], xs[0], gs[1], gs[0], kp[0], nullptr, row, stager, t0);
                    arrive(0);
                    TSTAMP(q4);
                    if (wave == 0) TLINE(4);
                    if (wave == 0) { TACC(ph[0], q1, q0); TACC(ph[1], q2, q1); TACC(ph[2], q3, q2); TACC(ph[3], q4, q3); }
                } else {
                    // a wave without a row arrives for its workgroup all the same -- but only behind the last minibatch's refresh, like a
                    // wave with one: an arrival counted early would complete the LAST minibatch's count for an owner still waiting on it
                    bool okw = true;
#pragma unroll
                    for (int b = 0; b < B; ++b) okw = okw && pipe_wait_lds(&sflag[SF_REF + b], NREL * mbcount, abort_w);
                    if (!okw) { fail(2); break; }
                    if (lane == 0) st_sc1_x2(part + 2 * row, 0.f, tg);
                    arrive(2); arrive(1); arrive(0);
                }
                // ---- R: nothing to do until the images are refreshed -- so the row waves do their share of it
                relay_all(mbcount + 1);
                if (wave == 0) { TLINE(20); }
                if (!alive) break;
            }
            // ---- Trainer._validate (trainer.py:405-418): validation row r on wave r % 4 of workgroup (r / 4) % G ----
            const bool go_on = epoch_end(epoch, [&]() {
                float vsum = 0.f;
                bool okw = true;
#pragma unroll
                for (int b = 0; b < B; ++b) okw = okw && pipe_wait_lds(&sflag[SF_REF + b], NREL * mbcount, abort_w);
                if (!okw) { fail(2); return; }
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
                    float ld_lane = rows_block_forward<U>(Solo4Lds{imgf, lane}, sel, h1, xs[1], xs[0], kp[0]);
                    ld_lane += rows_block_forward<U>(Solo4Lds{imgf + SOLO4_NF * 64, lane}, sel, h1, xs[0], xs[1], kp[1]);
                    ld_lane += rows_block_forward<U>(Solo4Lds{imgf + 2 * SOLO4_NF * 64, lane}, sel, h1, xs[1], xs[0], kp[2]);
                    vsum += log_prob(xs, ld_lane);
                }
                if (lane == 0) st_sc1(valid_base + (epoch & 1) * TRAIN_MAX_ROWS + row, vsum);
            }, [&]() {});
            if (!go_on) break;
        }
        __syncthreads();
#ifdef NNEST_STAMP
        // diagnostic build, cycles summed over the minibatches (workgroup 0, row wave 0): wait for the rows and block 0's images,
        // forward, backward up to the last store, drain in front of the last arrival
        if (a.losses && wg == 0 && threadIdx.x == 0) for (int i = 0; i < 4; ++i) a.losses[i] = (float)ph[i];
#endif
    } else {
        // =====================================================================================================================
        // service wave: the next minibatch's rows, this wave's weight-gradient job, the refresh block by block
        // =====================================================================================================================
        const int svc = wave - ROWS_PER_WG;                      // 0..3
        // ---- who owns what: service wave (wg, svc) runs job Jmine of every minibatch and keeps that tile's parameters and moments
        const int Jmine = svc * G + wg;
        const bool owner = Jmine < NJ;
        const int jblock = owner ? Jmine / (2 * NJOBS) : 0;
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
        // the parameters no job reaches, shared out over the service waves of the grid in whole 256-byte rows of a compact private array
        const int n_svc = G * PIPE_SVC, ndead = *a.gndead;
        const int ndead_pad = (ndead + 63) & ~63;
        const int dead_per = ((ndead + n_svc - 1) / n_svc + 63) & ~63;
        const int dead0 = min(ndead, Jmine * dead_per), dead1 = min(ndead, dead0 + dead_per);
        float *dw = a.gdst, *dm = a.gdst + ndead_pad, *dv = a.gdst + 2 * ndead_pad;
        for (int k = dead0 + lane; k < dead1; k += 64) {
            const int pidx = a.gdead[k];
            dw[k] = a.w[pidx]; dm[k] = a.m[pidx]; dv[k] = a.v[pidx];
        }
        auto snapshot = [&]() {   // best_model = deepcopy(netG)  (trainer.py:194, :208): every owner snapshots what it owns (one writer per entry)
            if (owner) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int wt = pipe_opaque(os.wt[r]), bt = pipe_opaque(os.bt[r]);
                    if (wt >= 0) st_sc1(a.best_w + wt, os.tw[r]);
                    if (bt >= 0) st_sc1(a.best_w + bt, os.bw[r]);
                }
            }
            for (int k = dead0 + lane; k < dead1; k += 64) st_sc1(a.best_w + a.gdead[k], dw[k]);
        };
        if (!resume) snapshot();
        const __amdgpu_buffer_rsrc_t partr = __builtin_amdgcn_make_buffer_rsrc(part_base, 0, 3 * 2 * TRAIN_MAX_ROWS * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t stgr = __builtin_amdgcn_make_buffer_rsrc(a.gstage, 0, B * 2 * SM::count * TRAIN_MAX_ROWS * 16 * 8, 0x00020000);
        int adam_t = a.adam_step ? *a.adam_step : 0;
        for (int epoch = 0; epoch < a.max_epochs; ++epoch) {
            float epoch_loss = 0.f;
            for (int mb = 0; mb < n_mb && alive; ++mb, ++mbcount) {
                const int M = min(a.batch, a.n_train - mb * a.batch);
                const int rows_pad = ((M + 15) >> 4) * 16;
                AdamStep ad;
                {
                    adam_t += 1;
                    const double bc1 = 1.0 - pow(0.9, (double)adam_t), bc2 = 1.0 - pow(0.999, (double)adam_t);
                    ad.step_size = (float)((double)a.lr / bc1);
                    ad.inv_bc2s = (float)(1.0 / sqrt(bc2));
                }
                TSTAMP(q0);
                if (svc == 1) TLINE(18);
                if (svc == PIPE_SVC - 1) TLINE(19);
                if (svc < NPW) {   // the NEXT minibatch's rows, beside this one's pass (xpre[(m + 1) & 1] was last read in minibatch m - 1's forward pass)
                    int e2 = epoch, m2 = mb + 1;
                    if (m2 == n_mb) { m2 = 0; e2 = epoch + 1; }
                    if (e2 < a.max_epochs) rows_prepare<U>(a, e2, m2, wg, xpre[(mbcount + 1) & 1]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_fetch_add(&sflag[SF_X], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                TSTAMP(q1);
                const unsigned int arrived = (unsigned int)G * (unsigned int)(mbcount + 1);
                const int want = mbcount + 1;
                if (owner) {
                    // ---- W + A: this wave's weight-gradient job; Adam on the tile's parameters in this wave's registers ----
                    if (!pipe_wait_ctr(arrival + 32 * jblock, arrived, abort_w, a.gerr)) { fail(3); break; }
                    TSTAMP(q2);
                    if (svc == 0) TLINE(8);
                    if (svc == 1) TLINE(5);
                    const int bn = Jmine / NJOBS;
                    const int q = Jmine % NJOBS;
                    const int lane_o = pipe_opaque(lane);
                    f32x4 bt = {0.f, 0.f, 0.f, 0.f}, t;
                    constexpr int J_W3 = U, J_W2 = 1;
                    constexpr int CTB = TRAIN_MAX_ROWS * 16 * 8;   // bytes of one column tile of granules
                    const int rb = pipe_opaque(bn * SM::count * CTB);   // this job's (block, net) region
                    // (the arrivals were counted when the rows' stores were issued, not when they landed: an operand without this
                    // minibatch's tag sends the wave round again)
                    // the job's two operands (weight_grad_jobs' enumeration): gradient tile x activation tile of the (block, net) region
                    const int ct_g = q < J_W3 ? SM::gout(q) : (q - J_W3 < J_W2 ? SM::gpre(1, 0) : SM::gpre(0, 0));
                    const int ct_a = q < J_W3 ? SM::act(1, 0) : (q - J_W3 < J_W2 ? SM::act(0, 0) : SM::m(q - J_W3 - J_W2));
                    for (int i = 0; i < sl_j; ++i) __builtin_amdgcn_s_sleep(1);   // (the count runs ahead of the rows' stores by their way to memory)
                    for (int polls = 0; ; ++polls) {
                        bool fresh;
                        t = contract_rows_tagged<true>(stgr, rb + ct_g * CTB, rb + ct_a * CTB, rows_pad, min(rows_pad, ROWS_PER_WG * G), lane_o, want, bt, fresh);   // (the bias sum of a job without a bias goes nowhere: os.bt = -1)
                        if (fresh) break;
                        if (polls > GRID_MAX_POLLS / 64 || pipe_lds_ld(abort_w)) { fail(4); break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (!alive) break;
                    const float gt[4] = {t.x, t.y, t.z, t.w}, gb[4] = {bt.x, bt.y, bt.z, bt.w};
                    if (svc == 0) TLINE(9);
                    if (svc == 1) TLINE(6);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (os.wt[r] >= 0) adam_reg(a, ad, os.tw[r], gt[r], os.tm[r], os.tv[r]);
                        if (os.bt[r] >= 0) adam_reg(a, ad, os.bw[r], gb[r], os.bm[r], os.bv[r]);
                    }
                    // the NEW weights, every one with the minibatch's TAG beside it ({w, tag} granules of 8 bytes inside 16-byte
                    // stores: element i of the backward fragment image at pub[2 i], its tag at pub[2 i + 1])
                    const float tg = __int_as_float(want);
                    float *pb = pipe_opaque(pub);
                    st_sc1_f32x4(pb + 2 * (off_b + lane_o * 4), (f32x4){os.tw[0], tg, os.tw[1], tg});
                    st_sc1_f32x4(pb + 2 * (off_b + lane_o * 4) + 4, (f32x4){os.tw[2], tg, os.tw[3], tg});
                    if (off_bias >= 0 && (lane & 15) == 0) {
                        st_sc1_f32x4(pb + 2 * (off_bias + (lane_o >> 4) * 4), (f32x4){os.bw[0], tg, os.bw[1], tg});
                        st_sc1_f32x4(pb + 2 * (off_bias + (lane_o >> 4) * 4) + 4, (f32x4){os.bw[2], tg, os.bw[3], tg});
                    }
                    if (lane == 0) __hip_atomic_fetch_add(hint + 32 * jblock, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    TSTAMP(q3);
                    if (svc == 0) TLINE(10);
                    if (svc == 1) TLINE(7);
                } else {
                    TSTAMP(q2); TSTAMP(q3);
                }
                for (int k = dead0 + lane; k < dead1; k += 64) {   // the parameters no job reaches: zero gradient, weight decay only
                    float w_ = dw[k], m_ = dm[k], v_ = dv[k];
                    adam_reg(a, ad, w_, 0.f, m_, v_);
                    dw[k] = w_; dm[k] = m_; dv[k] = v_;
                }
                // ---- R: this wave's share of the refresh (the owners of block 1's and block 0's jobs have none)
                if (relayer) {
                    relay_all(want);
                    if (svc == PIPE_SVC - 1) TLINE(17);
                }
                if (!alive) break;
                TSTAMP(q4);
                if (svc == 0) { TACC(ph[0], q1, q0); TACC(ph[1], q2, q1); TACC(ph[2], q3, q2); TACC(ph[3], q4, q3); }
                // loss = -mean(log_probs)  (trainer.py:394) over the rows' words: complete behind block 0's arrivals; only the
                // wave that reports the epoch losses needs it (the words are triple-buffered: nothing waits for this sum)
                if (wg == 0 && svc == PIPE_SVC - 1) {
                    float lsum = 0.f;
                    for (int polls = 0; ; ++polls) {   // (the words are granules too: {log p, tag})
                        typedef float f32x2_ __attribute__((ext_vector_type(2)));
                        const f32x2_ w0 = __builtin_bit_cast(f32x2_, __builtin_amdgcn_raw_buffer_load_b64(partr, pipe_opaque((mbcount % 3) * 2 * TRAIN_MAX_ROWS * 4 + 8 * min(lane, ROWS_PER_WG * G - 1)), 0, 16));
                        const f32x2_ w1 = __builtin_bit_cast(f32x2_, __builtin_amdgcn_raw_buffer_load_b64(partr, pipe_opaque((mbcount % 3) * 2 * TRAIN_MAX_ROWS * 4 + 8 * min(lane + 64, ROWS_PER_WG * G - 1)), 0, 16));
                        const bool ok = __float_as_int(w0.y) == want && __float_as_int(w1.y) == want;
                        if (__all(ok)) {
                            float v = (lane < ROWS_PER_WG * G ? w0.x : 0.f) + (lane + 64 < ROWS_PER_WG * G ? w1.x : 0.f);
#pragma unroll
                            for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
                            lsum = v;
                            break;
                        }
                        if (polls > GRID_MAX_POLLS / 64 || pipe_lds_ld(abort_w)) { fail(7); break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    epoch_loss += -lsum / (float)M;
                }
            }
            const bool go_on = epoch_end(epoch, [&]() {
                if (wg == 0 && svc == PIPE_SVC - 1 && lane == 0) ctlf[1] = epoch_loss / (float)a.n_train;   // trainer.py:403
            }, snapshot);
            if (!go_on) break;
        }
        // every owner writes what it owns back -- write-through stores, one writer per entry (launch_repack rebuilds the inference
        // kernels' forward image from a.w behind this launch)
        __syncthreads();
        const bool restore = ctl[0] != 0 || (a.flags & NNEST_TRAIN_FINALIZE);   // netG.load_state_dict(best_model)  (trainer.py:241)
        if (owner) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int wt = pipe_opaque(os.wt[r]), bt = pipe_opaque(os.bt[r]);
                if (wt >= 0) {
                    st_sc1(a.w + wt, restore ? ld_sc1(a.best_w + wt) : os.tw[r]);
                    st_sc1(a.m + wt, os.tm[r]); st_sc1(a.v + wt, os.tv[r]);
                }
                if (bt >= 0) {
                    st_sc1(a.w + bt, restore ? ld_sc1(a.best_w + bt) : os.bw[r]);
                    st_sc1(a.m + bt, os.bm[r]); st_sc1(a.v + bt, os.bv[r]);
                }
            }
        }
        for (int k = dead0 + lane; k < dead1; k += 64) {
            const int pidx = a.gdead[k];
            st_sc1(a.w + pidx, restore ? ld_sc1(a.best_w + pidx) : dw[k]);
            st_sc1(a.m + pidx, dm[k]); st_sc1(a.v + pidx, dv[k]);
        }
#ifdef NNEST_STAMP
        // service wave 0 of workgroup 0: row preparation, wait for its block's arrivals, job + Adam + publish, refresh
        if (a.losses && wg == 0 && svc == 0 && lane == 0) for (int i = 0; i < 4; ++i) a.losses[4 + i] = (float)ph[i];
#endif
        if (wg == 0 && svc == 0 && lane == 0 && a.adam_step) *a.adam_step = adam_t;   // (a service wave: it has counted the Adam steps)
    }
#ifdef NNEST_STAMP
    __syncthreads();
    if (wg == 0 && threadIdx.x < 24 && a.losses) a.losses[16 + threadIdx.x] = tline[threadIdx.x];
#endif
    if (wg == 0 && threadIdx.x == 0) {
        a.result->epochs_run = a.epoch_offset + epochs_run;
        a.result->best_epoch = ctl[2];
        a.result->best_validation_loss = ctlf[0];
        a.result->last_train_loss = last_train_loss;
        a.result->counter = *a.gerr ? *a.gerr : ctl[1];   // (a wait ran out: which one, fail())
        a.result->stopped = *a.gerr ? 2 : (ctl[0] != 0 ? 1 : 0);   // 2: a wait ran out (include/nnest_hip.h)
    }
}

template <int U>
__global__ void __launch_bounds__(TRAIN_THREADS) train_kernel_pipe(TrainArgs a) {
    (void)a;   // (read from the kernarg segment by the two bodies)
    if ((threadIdx.x >> 6) < ROWS_PER_WG) pipe_body<U, 0>();
    else pipe_body<U, 1>();
}

// The pipelined form is OPT-IN (NNEST_TRAIN_FORM=pipe in the environment): measured at 12.1-12.7 us per minibatch at config 2 against
// train_kernel_rows' 11.4 (profiles/r05/k5_pipe_timeline.txt; DESIGN.md 3.2 has the timeline and why pipelining per block does not
// shorten a chain of memory-side round trips).  Kept, tested (tests/test_gpu_train.py runs both forms), not the default.
static bool pipe_eligible(const TrainArgs &a) {
    static const bool on = [] { const char *e = getenv("NNEST_TRAIN_FORM"); return e && !strcmp(e, "pipe"); }();
    return on && rows_eligible(a);
}

template <int U>
static hipError_t launch_train_pipe_t(TrainArgs a, float *gridws, hipStream_t st) {
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
    static_assert(GS::NJOBS * 2 * ROWS_B * 2 * 256 >= 2048, "the rows' words and the arrival counters live in the tile area");
    a.gstage = a.gimgb + ((a.s.image_floats + 63) & ~63) + 64;   // the TAGGED staging area: 2 floats per value (pipe_extra_workspace_floats)
    a.gstage += (64 - ((size_t)(a.gstage - gridws) & 63)) & 63;
    hipError_t e = hipMemsetAsync(a.gstage, 0, 2 * GS::stage(a.s) * sizeof(float), st);   // tag 0: no minibatch's
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.gtile, 0, 2048 * sizeof(float), st);   // the rows' words, the arrival counters
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.gpos, 0xFF, (size_t)a.s.num_params() * sizeof(int), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.gimgf, 0, (size_t)2 * ((a.s.image_floats + 63) & ~63) * sizeof(float), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.gpart, 0, (64 + 16) * sizeof(float), st);  // the barrier counter, the error word, the dead count
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((grid_gpos_kernel<U, 1, 1>), dim3(32), dim3(256), 0, st, a.gpos, a.s);
    hipLaunchKernelGGL(grid_dead_kernel, dim3(32), dim3(256), 0, st, a.gpos, a.s.num_params(), a.gdead, a.gndead);
    hipLaunchKernelGGL((pipe_maps_kernel<U>), dim3(32), dim3(256), 0, st, reinterpret_cast<unsigned int *>(a.gown), a.s);   // (the owners' record area of train_kernel_grid: unused here)
    {   // how long a refresh's first data poll is held back behind its block's arrivals (units of 64 cycles; NNEST_K5_SLEEP overrides)
        int r = 0, j = 0;
        if (const char *ev = getenv("NNEST_K5_SLEEP")) sscanf(ev, "%d,%d", &r, &j);
        a.gld_in = (float)((r & 255) + 256 * (j & 255));   // (the VJP's scalar: unused by the training loop)
    }
    const int NJ = ROWS_B * 2 * (2 * U + 1);
    const int G = max((a.batch + ROWS_PER_WG - 1) / ROWS_PER_WG, (NJ + PIPE_SVC - 1) / PIPE_SVC);
    const size_t lds = (size_t)PipeLds<U>::FLOATS * sizeof(float);   // the two images + the re-lay's dummy floats + the prepared rows + the control words
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(train_kernel_pipe<U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((train_kernel_pipe<U>), dim3(G), dim3(TRAIN_THREADS), lds, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_repack(a.w, a.img_fwd, a.s, st);
}

static hipError_t dispatch_train_pipe(const TrainArgs &a, float *gridws, hipStream_t st) {
    switch (a.s.NT) {
        case 2: return launch_train_pipe_t<2>(a, gridws, st);
#ifndef NNEST_DEV_PIPE2
        case 1: return launch_train_pipe_t<1>(a, gridws, st);
        case 3: return launch_train_pipe_t<3>(a, gridws, st);
        case 4: return launch_train_pipe_t<4>(a, gridws, st);
#endif
    }
    return hipErrorInvalidConfiguration;
}
