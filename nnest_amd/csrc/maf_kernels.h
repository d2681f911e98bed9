// maf_kernels.h -- the masked autoregressive flow's kernels (included at the end of nnest_kernels.hip, inside namespace nnest):
//   maf_repack_kernel   packed weights -> masked forward / backward fragment images + the group table (maf_tile.h)
//   maf_pass_kernel     K1 forward / K2 inverse / log_probs / K3 fused inverse + box prior + likelihood
//   maf_mh_kernel       K4: the persistent constrained-Metropolis loop (mh_body) with the grouped sequential inverse
// [Build-defined flow: the reference has none to compare with; parity is against a CPU restatement of the same definition.]

__global__ void maf_repack_kernel(const float *__restrict__ packed, float *__restrict__ imgf, float *__restrict__ imgb, FlowShape s) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < s.image_floats; idx += gridDim.x * blockDim.x) {
        const int f = maf_fwd_src(s, idx);
        imgf[idx] = f >= 0 ? packed[f] : (f <= -2 ? (float)(-2 - f) : 0.f);
        if (imgb) {
            const int b = maf_bwd_src(s, idx);
            imgb[idx] = b >= 0 ? packed[b] : 0.f;
        }
    }
}

hipError_t launch_maf_repack(const float *packed, float *imgf, float *imgb, const FlowShape &s, hipStream_t st) {
    int block = 256, grid = (s.image_floats + block - 1) / block;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(maf_repack_kernel, dim3(grid), dim3(block), 0, st, packed, imgf, imgb, s);
    return hipGetLastError();
}

template <int NT, int NH>
__global__ void __launch_bounds__(256) maf_pass_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_img[];
    stage_image(lds_img, a.img, a.s.image_floats);
    const float *img = lds_img;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = a.waves_active;
    const int ntiles = (a.N + 15) >> 4;
    const int w = lane & 15, g = lane >> 4;
    if (wave >= wpb) return;
    for (int tile = blockIdx.x * wpb + wave; tile < ntiles; tile += gridDim.x * wpb) {
        const int row = tile * 16 + w;
        const bool ok = row < a.N;
        f32x4 xs[2][NT];
        load_tile<NT>(a.in, row, ok, a.s.D, lane, xs);
        float ld;
        if (a.mode == PASS_FORWARD || a.mode == PASS_LOGPROB) ld = maf_forward_tile<NT, NH>(img, a.s, lane, xs);
        else ld = maf_inverse_tile<NT, NH>(img, a.s, lane, xs);
        ld = group_sum(ld);
        if (a.mode == PASS_LOGPROB) {
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int tau = 0; tau < NT; ++tau) ss += base_E4(xs[c][tau], a.s.base_beta);
            ss = group_sum(ss);
            if (ok && g == 0) a.out[row] = -ss + a.s.base_const * (float)a.s.D + ld;
            continue;
        }
        if (a.out) store_tile<NT>(a.out, row, ok, a.s.D, lane, xs);
        if (a.logdet && ok && g == 0) a.logdet[row] = ld;
        if (a.mode == PASS_INVERSE_LOGLIKE) {
            int inb = inbox_tile<NT>(xs, lane);
            double ll = loglike_tile<NT>(a.like, a.s.D, lane, xs);
            if (ok && g == 0) {
                a.logl[row] = ll;
                if (a.inbox) a.inbox[row] = inb;
            }
        }
    }
}

template <int NT, int NH>
struct MafInverse {
    const float *img;
    FlowShape s;
    int lane;
#ifdef NNEST_STAMP
    unsigned long long t_mlp = 0, t_xch = 0, t_upd = 0;
#endif
    __device__ __forceinline__ float operator()(f32x4 (&xs)[2][NT]) const { return maf_inverse_tile<NT, NH>(img, s, lane, xs); }
};

// four waves per workgroup share one LDS copy of the image; one wave per SIMD (the grouped inverse holds the vector, both nets'
// outputs and the proposal state: more registers than two waves per SIMD leave)
template <int NT, int NH, bool DBG>
__global__ void __launch_bounds__(256, 1) maf_mh_kernel(MhArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_img[];
    stage_image(lds_img, a.img, a.s.image_floats);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int tile = blockIdx.x * wpb + wave;
    if (tile >= ((a.C + 15) >> 4)) return;
    MafInverse<NT, NH> inv = {lds_img, a.s, lane};
    XoshiroNoise<NT> noise;
    noise.init(a.seed, a.walker_offset + (uint64_t)(tile * 16 + (lane & 15)), lane >> 4, a.s.D);
    mh_body<NT, DBG>(a, tile, lane, inv, noise, true);
}

template <int NT, int NH>
static hipError_t launch_maf_pass_t(const PassArgs &a_in, int num_cu, hipStream_t st) {
    const int ntiles = (a_in.N + 15) / 16;
    int block, grid;
    pick_geometry(ntiles, num_cu, 4, &block, &grid);
    PassArgs a = a_in;
    a.waves_active = block / 64;
    if (grid > 8 * num_cu) grid = 8 * num_cu;
    const size_t img_bytes = (size_t)a.s.image_floats * 4;
    hipError_t e = allow_lds(maf_pass_kernel<NT, NH>, img_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((maf_pass_kernel<NT, NH>), dim3(grid), dim3(256), img_bytes, st, a);
    return hipGetLastError();
}

template <int NT, int NH>
static hipError_t launch_maf_mh_t(const MhArgs &a, int num_cu, hipStream_t st) {
    const int ntiles = (a.C + 15) / 16;
    const int form = mh_flag_form(a.flags);
    if (form != MH_FORM_AUTO && form != MH_FORM_IMAGE) return hipErrorInvalidConfiguration;
    const bool batch = (a.flags & NNEST_MH_DYNAMIC_BATCH) != 0;
    if (batch && !a.sync) return hipErrorInvalidValue;
    if (batch && mh_flag_lag(a.flags) > 0 && mh_flag_warm(a.flags) > 0) return hipErrorInvalidConfiguration;   // exact warm-up steps: the solo form only (nnest_hip.h)
    int block, grid;
    pick_geometry(ntiles, num_cu, 4, &block, &grid);
    if (batch && grid > num_cu) return hipErrorInvalidConfiguration;  // one workgroup per CU is what is certainly resident
    const size_t img_bytes = (size_t)a.s.image_floats * 4;
    if (a.noise_dz || a.hist_x || a.hist_logl) {
        hipError_t e = allow_lds(maf_mh_kernel<NT, NH, true>, img_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((maf_mh_kernel<NT, NH, true>), dim3(grid), dim3(block), img_bytes, st, a);
    } else {
        hipError_t e = allow_lds(maf_mh_kernel<NT, NH, false>, img_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((maf_mh_kernel<NT, NH, false>), dim3(grid), dim3(block), img_bytes, st, a);
    }
    return hipGetLastError();
}

bool maf_shape_supported(const FlowShape &s) {   // the image (fragments of 2 NT tiles per net + group table) has to fit one CU's LDS
    return s.NH == 1 && s.NT >= 1 && s.NT <= 4 && (size_t)s.image_floats * 4 <= (size_t)LDS_IMAGE_LIMIT;
}

#define DISPATCH_MAF(FN, s, ...)                      \
    do {                                              \
        switch ((s).NT) {                             \
            case 1: return FN<1, 1>(__VA_ARGS__);     \
            case 2: return FN<2, 1>(__VA_ARGS__);     \
            case 3: return FN<3, 1>(__VA_ARGS__);     \
            case 4: return FN<4, 1>(__VA_ARGS__);     \
        }                                             \
        return hipErrorInvalidConfiguration;          \
    } while (0)

static hipError_t launch_maf_pass(const PassArgs &a, int num_cu, hipStream_t st) { DISPATCH_MAF(launch_maf_pass_t, a.s, a, num_cu, st); }
static hipError_t launch_maf_mh(const MhArgs &a, int num_cu, hipStream_t st) { DISPATCH_MAF(launch_maf_mh_t, a.s, a, num_cu, st); }
