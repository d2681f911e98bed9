// nnest_abi.hip -- the extern "C" surface of libnnest_hip.so (include/nnest_hip.h): handle management,
// argument checks, error reporting.  No torch types, no exceptions across the boundary.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdarg.h>

#include "nnest_internal.h"
#include "mh_common.h"

using namespace nnest;

struct nnest_nvp {
    FlowShape s;
    int device;
    int num_cu;
    int num_params;
    float *w;        // packed weights (state_dict order)
    float *adam_m;   // exp_avg
    float *adam_v;   // exp_avg_sq
    float *best_w;   // best-validation snapshot (Trainer.train's deepcopy, trainer.py:194, :208)
    float *img;      // MFMA fragment image of w
    int *adam_step;  // device int: torch.optim.Adam state['step']
    float *train_ws; // training workspace
    int *fwd_pos;    // packed parameter -> element of the forward fragment image (-1: absent)
    int *bwd_pos;    // packed parameter -> element of the backward fragment image
    size_t train_ws_floats;
    float *img_bwd;  // MAF: the transposed fragment image (the RealNVP training kernels keep theirs in the workspace / LDS)
    int *gpos;       // MAF: packed parameter -> slot of a tile's weight-gradient buffer
    unsigned int *ticket; // MAF: block counter of the fused update kernel (nnest_maf_train_epoch)
};

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace nnest {
void set_last_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }
}  // namespace nnest

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) return fail(NNEST_E_HIP, "%s: %s", #expr, hipGetErrorString(e__));       \
    } while (0)

// the fragment image(s) of the handle's flow from its packed weights
static hipError_t refresh_images(nnest_nvp *h, hipStream_t st) {
    if (h->s.kind == FLOW_KIND_MAF) return launch_maf_repack(h->w, h->img, h->img_bwd, h->s, st);
    if (h->s.scale_mode != NNEST_SCALE_AFFINE) {
        hipError_t e = launch_zero_scale_nets(h->w, h->s, st);  // unused slots stay 0
        if (e != hipSuccess) return e;
    }
    return launch_repack(h->w, h->img, h->s, st);
}

extern "C" {

int nnest_hip_version(void) { return NNEST_HIP_ABI_VERSION; }

const char *nnest_hip_last_error(void) { return g_err; }

int nnest_hip_device_info(int *num_cu, int *clock_khz, char *name, int name_len) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (num_cu) *num_cu = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    return NNEST_OK;
}

int nnest_nvp_create(int D, int H, int B, int L, nnest_nvp_t **out) {
    return nnest_nvp_create_scaled(D, H, B, L, NNEST_SCALE_AFFINE, out);
}

int nnest_nvp_create_scaled(int D, int H, int B, int L, int scale_mode, nnest_nvp_t **out) {
    if (!out) return fail(NNEST_E_ARG, "out is NULL");
    *out = nullptr;
    if (D < 1 || H < 1 || B < 1 || L < 0) return fail(NNEST_E_ARG, "bad shape D=%d H=%d B=%d L=%d", D, H, B, L);
    if (scale_mode < NNEST_SCALE_AFFINE || scale_mode > NNEST_SCALE_CONSTANT)
        return fail(NNEST_E_ARG, "scale_mode=%d (0 '', 1 'translate', 2 'constant')", scale_mode);
    if (scale_mode == NNEST_SCALE_CONSTANT && B > 8)
        return fail(NNEST_E_UNSUPPORTED, "scale='constant' with num_blocks=%d > 8", B);
    if (H % 16 != 0)
        return fail(NNEST_E_UNSUPPORTED, "hidden_dim=%d: the gfx950 kernels tile the hidden layer by 16 (MFMA 16x16x4)", H);
    FlowShape s;
    s.D = D; s.H = H; s.B = B; s.L = L;
    s.NT = ((D + 1) / 2 + 15) / 16;
    s.NH = H / 16;
    s.net_floats = frag_net_floats(s.NT, s.NH, L);
    s.image_floats = B * 2 * s.net_floats;
    s.net_params = H * D + H + L * (H * H + H) + D * H + D;
    s.scale_mode = scale_mode;
    s.kind = FLOW_KIND_NVP;
    s.G = 0;
    s.base_beta = 0.f;
    s.base_const = -0.91893853320467274f;  // -log(2 pi) / 2
    if (!shape_supported(s))
        return fail(NNEST_E_UNSUPPORTED, "x_dim=%d hidden_dim=%d not instantiated (x_dim<=128 at H=16, <=64 at H=32, <=32 at H=64)", D, H);
    nnest_nvp *h = new nnest_nvp();
    memset(h, 0, sizeof(*h));
    h->s = s;
    h->num_params = s.num_params();
    if (hipGetDevice(&h->device) != hipSuccess) { delete h; return fail(NNEST_E_HIP, "hipGetDevice failed (no GPU?)"); }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, h->device) != hipSuccess) { delete h; return fail(NNEST_E_HIP, "hipGetDeviceProperties failed"); }
    h->num_cu = p.multiProcessorCount;
    size_t nb = (size_t)h->num_params * sizeof(float);
    h->train_ws_floats = train_workspace_floats(s, 128);
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc((void **)&h->w, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->adam_m, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->adam_v, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->best_w, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->img, (size_t)s.image_total() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->adam_step, sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->train_ws, h->train_ws_floats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->fwd_pos, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->bwd_pos, nb);
    if (e == hipSuccess) e = launch_build_pos(h->fwd_pos, h->bwd_pos, s, 0);
    if (e == hipSuccess) e = hipStreamSynchronize(0);
    if (e == hipSuccess) e = hipMemset(h->w, 0, nb);
    if (e == hipSuccess) e = hipMemset(h->adam_m, 0, nb);
    if (e == hipSuccess) e = hipMemset(h->adam_v, 0, nb);
    if (e == hipSuccess) e = hipMemset(h->adam_step, 0, sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->img, 0, (size_t)s.image_total() * sizeof(float));
    if (e != hipSuccess) {
        nnest_nvp_destroy(h);
        return fail(NNEST_E_HIP, "device allocation failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return NNEST_OK;
}

int nnest_maf_create(int D, int H, int B, int L, nnest_nvp_t **out) {
    if (!out) return fail(NNEST_E_ARG, "out is NULL");
    *out = nullptr;
    if (D < 2 || H < 1 || B < 1 || L < 0) return fail(NNEST_E_ARG, "bad shape D=%d H=%d B=%d L=%d (a MAF needs x_dim >= 2)", D, H, B, L);
    if (H != 16) return fail(NNEST_E_UNSUPPORTED, "maf: hidden_dim=%d, the kernels are instantiated for 16", H);
    FlowShape s;
    s.D = D; s.H = H; s.B = B; s.L = L;
    s.NT = ((D + 1) / 2 + 15) / 16;
    s.NH = H / 16;
    s.net_floats = frag_net_floats(2 * s.NT, s.NH, L);                 // fragments over both parity classes: 2 NT tiles
    s.image_floats = B * 2 * s.net_floats + B * 16 * 2 * s.NT;         // + the group of every slot, per block
    s.net_params = H * D + H + L * (H * H + H) + D * H + D;
    s.scale_mode = NNEST_SCALE_AFFINE;
    s.kind = FLOW_KIND_MAF;
    s.G = maf_num_groups(D, H);
    s.base_beta = 0.f;
    s.base_const = -0.91893853320467274f;
    if (s.NT > 4 || !maf_shape_supported(s))
        return fail(NNEST_E_UNSUPPORTED, "maf: x_dim=%d num_blocks=%d num_layers=%d: the fragment image (%d floats) has to fit one CU's LDS and x_dim <= 128", D, B, L, s.image_floats);
    nnest_nvp *h = new nnest_nvp();
    memset(h, 0, sizeof(*h));
    h->s = s;
    h->num_params = s.num_params();
    if (hipGetDevice(&h->device) != hipSuccess) { delete h; return fail(NNEST_E_HIP, "hipGetDevice failed (no GPU?)"); }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, h->device) != hipSuccess) { delete h; return fail(NNEST_E_HIP, "hipGetDeviceProperties failed"); }
    h->num_cu = p.multiProcessorCount;
    const size_t nb = (size_t)h->num_params * sizeof(float), ib = (size_t)s.image_floats * sizeof(float);
    h->train_ws_floats = maf_workspace_floats(s);
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc((void **)&h->w, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->adam_m, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->adam_v, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->best_w, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->img, ib);
    if (e == hipSuccess) e = hipMalloc((void **)&h->img_bwd, ib);
    if (e == hipSuccess) e = hipMalloc((void **)&h->adam_step, sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->train_ws, h->train_ws_floats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->gpos, nb);
    if (e == hipSuccess) e = launch_maf_build_gpos(h->gpos, s, 0);
    if (e == hipSuccess) e = hipMalloc((void **)&h->fwd_pos, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->bwd_pos, nb);
    if (e == hipSuccess) e = launch_maf_build_pos(h->fwd_pos, h->bwd_pos, s, 0);
    if (e == hipSuccess) e = hipMalloc((void **)&h->ticket, sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemset(h->ticket, 0, sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemset(h->w, 0, nb);
    if (e == hipSuccess) e = hipMemset(h->adam_m, 0, nb);
    if (e == hipSuccess) e = hipMemset(h->adam_v, 0, nb);
    if (e == hipSuccess) e = hipMemset(h->adam_step, 0, sizeof(int));
    if (e == hipSuccess) e = refresh_images(h, 0);
    if (e == hipSuccess) e = hipStreamSynchronize(0);
    if (e != hipSuccess) {
        nnest_nvp_destroy(h);
        return fail(NNEST_E_HIP, "device allocation failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return NNEST_OK;
}

int nnest_maf_num_groups(const nnest_nvp_t *h) { return (h && h->s.kind == FLOW_KIND_MAF) ? h->s.G : -1; }

int nnest_nvp_destroy(nnest_nvp_t *h) {
    if (!h) return NNEST_OK;
    (void)hipFree(h->w); (void)hipFree(h->adam_m); (void)hipFree(h->adam_v); (void)hipFree(h->best_w); (void)hipFree(h->img);
    (void)hipFree(h->adam_step); (void)hipFree(h->train_ws); (void)hipFree(h->fwd_pos); (void)hipFree(h->bwd_pos);
    (void)hipFree(h->img_bwd); (void)hipFree(h->gpos); (void)hipFree(h->ticket);
    delete h;
    return NNEST_OK;
}

int nnest_nvp_num_params(const nnest_nvp_t *h) { return h ? h->num_params : -1; }

int nnest_nvp_set_base(nnest_nvp_t *h, float beta) {
    if (!h) return fail(NNEST_E_ARG, "NULL handle");
    if (!(beta >= 0.f)) return fail(NNEST_E_ARG, "beta=%g: 0 selects N(0, I), beta > 0 GeneralisedNormal(0, 1, beta)", (double)beta);
    h->s.base_beta = beta;
    h->s.base_const = beta == 0.f ? -0.91893853320467274f : (float)(log((double)beta) - log(2.0) - lgamma(1.0 / (double)beta));
    return NNEST_OK;
}

int nnest_nvp_load_weights(nnest_nvp_t *h, const float *packed_host, void *stream) {
    if (!h || !packed_host) return fail(NNEST_E_ARG, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(h->w, packed_host, (size_t)h->num_params * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(refresh_images(h, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NNEST_OK;
}

int nnest_nvp_store_weights(nnest_nvp_t *h, float *packed_host, void *stream) {
    if (!h || !packed_host) return fail(NNEST_E_ARG, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(packed_host, h->w, (size_t)h->num_params * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NNEST_OK;
}

int nnest_nvp_device_ptrs(nnest_nvp_t *h, float **w_dev, float **m_dev, float **v_dev) {
    if (!h) return fail(NNEST_E_ARG, "NULL handle");
    if (w_dev) *w_dev = h->w;
    if (m_dev) *m_dev = h->adam_m;
    if (v_dev) *v_dev = h->adam_v;
    return NNEST_OK;
}

int nnest_nvp_store_adam(nnest_nvp_t *h, float *exp_avg_host, float *exp_avg_sq_host, void *stream) {
    if (!h || !exp_avg_host || !exp_avg_sq_host) return fail(NNEST_E_ARG, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    size_t nb = (size_t)h->num_params * sizeof(float);
    HIP_TRY(hipMemcpyAsync(exp_avg_host, h->adam_m, nb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(exp_avg_sq_host, h->adam_v, nb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NNEST_OK;
}

int nnest_nvp_load_adam(nnest_nvp_t *h, const float *exp_avg_host, const float *exp_avg_sq_host, void *stream) {
    if (!h || !exp_avg_host || !exp_avg_sq_host) return fail(NNEST_E_ARG, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    size_t nb = (size_t)h->num_params * sizeof(float);
    HIP_TRY(hipMemcpyAsync(h->adam_m, exp_avg_host, nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->adam_v, exp_avg_sq_host, nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NNEST_OK;
}

int nnest_nvp_adam_state(nnest_nvp_t *h, int *step_count, int set_step, int reset_moments, void *stream) {
    if (!h) return fail(NNEST_E_ARG, "NULL handle");
    hipStream_t st = (hipStream_t)stream;
    if (reset_moments) {
        HIP_TRY(hipMemsetAsync(h->adam_m, 0, (size_t)h->num_params * sizeof(float), st));
        HIP_TRY(hipMemsetAsync(h->adam_v, 0, (size_t)h->num_params * sizeof(float), st));
    }
    if (set_step >= 0) HIP_TRY(hipMemcpyAsync(h->adam_step, &set_step, sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (step_count) {
        HIP_TRY(hipMemcpyAsync(step_count, h->adam_step, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return NNEST_OK;
}

static int check_rows(const nnest_nvp_t *h, const void *a, const void *b, int N) {
    if (!h) return fail(NNEST_E_ARG, "NULL handle");
    if (N < 0) return fail(NNEST_E_ARG, "N=%d < 0", N);
    if (N > 0 && (!a || !b)) return fail(NNEST_E_ARG, "NULL device buffer");
    return NNEST_OK;
}

int nnest_nvp_forward(nnest_nvp_t *h, const float *x_dev, float *z_dev, float *logdet_dev, int N, void *stream) {
    int rc = check_rows(h, x_dev, z_dev, N);
    if (rc) return rc;
    HIP_TRY(launch_pass(h->img, h->s, PASS_FORWARD, x_dev, z_dev, logdet_dev, nullptr, nullptr, N, LikeSpec(), h->num_cu,
                        (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_nvp_inverse(nnest_nvp_t *h, const float *z_dev, float *x_dev, float *logdet_dev, int N, void *stream) {
    int rc = check_rows(h, z_dev, x_dev, N);
    if (rc) return rc;
    HIP_TRY(launch_pass(h->img, h->s, PASS_INVERSE, z_dev, x_dev, logdet_dev, nullptr, nullptr, N, LikeSpec(), h->num_cu,
                        (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_nvp_log_probs(nnest_nvp_t *h, const float *x_dev, float *logp_dev, int N, void *stream) {
    int rc = check_rows(h, x_dev, logp_dev, N);
    if (rc) return rc;
    HIP_TRY(launch_pass(h->img, h->s, PASS_LOGPROB, x_dev, logp_dev, nullptr, nullptr, nullptr, N, LikeSpec(), h->num_cu,
                        (hipStream_t)stream));
    return NNEST_OK;
}

static int check_like(const nnest_like_t *like, int D, LikeSpec *out) {
    if (!like) return fail(NNEST_E_ARG, "like is NULL");
    if (like->id < 0 || like->id >= NNEST_LIKE_COUNT) return fail(NNEST_E_ARG, "unknown likelihood id %d", like->id);
    if (like->id == NNEST_LIKE_EGGBOX && D != 2) return fail(NNEST_E_ARG, "Eggbox is defined for x_dim = 2 (likelihoods.py:97-102)");
    if (like->id == NNEST_LIKE_GAUSSMIX && D < 2) return fail(NNEST_E_ARG, "GaussianMix needs x_dim >= 2");
    out->id = like->id;
    out->scale = like->scale;
    for (int i = 0; i < 6; ++i) out->p[i] = like->params[i];
    return NNEST_OK;
}

int nnest_nvp_inverse_loglike(nnest_nvp_t *h, const nnest_like_t *like, const float *z_dev, float *x_dev,
                              float *logdet_dev, double *logl_dev, int *inbox_dev, int N, void *stream) {
    int rc = check_rows(h, z_dev, logl_dev, N);
    if (rc) return rc;
    LikeSpec lk;
    if ((rc = check_like(like, h->s.D, &lk))) return rc;
    HIP_TRY(launch_pass(h->img, h->s, PASS_INVERSE_LOGLIKE, z_dev, x_dev, logdet_dev, logl_dev, inbox_dev, N, lk,
                        h->num_cu, (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_loglike(const nnest_like_t *like, const float *x_unit_dev, double *logl_dev, int N, int D, void *stream) {
    if (N < 0 || D < 1) return fail(NNEST_E_ARG, "bad N=%d D=%d", N, D);
    LikeSpec lk;
    int rc = check_like(like, D, &lk);
    if (rc) return rc;
    if (D > 128) return fail(NNEST_E_UNSUPPORTED, "x_dim=%d > 128", D);
    if (N > 0 && (!x_unit_dev || !logl_dev)) return fail(NNEST_E_ARG, "NULL device buffer");
    int dev = 0, num_cu = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    HIP_TRY(launch_loglike(lk, x_unit_dev, logl_dev, N, D, num_cu, (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_mh_constrained_steps(nnest_nvp_t *h, const nnest_like_t *like, float *z_dev, float *x_dev,
                               double *logl_dev, double loglstar, float step_size, int steps, int C, int flags,
                               const float *noise_dz_dev, const float *noise_u_dev, uint64_t seed,
                               uint64_t walker_offset, float *hist_x_dev, double *hist_logl_dev, int *n_accept_dev,
                               int *n_call_dev, float *scale_out_dev, void *sync_dev, void *stream) {
    int rc = check_rows(h, z_dev, logl_dev, C);
    if (rc) return rc;
    LikeSpec lk;
    if ((rc = check_like(like, h->s.D, &lk))) return rc;
    if (steps < 0) return fail(NNEST_E_ARG, "steps=%d < 0", steps);
    if ((noise_dz_dev == nullptr) != (noise_u_dev == nullptr))
        return fail(NNEST_E_ARG, "noise_dz_dev and noise_u_dev must both be given or both be NULL");
    if ((flags & NNEST_MH_DYNAMIC_BATCH) && !sync_dev) return fail(NNEST_E_ARG, "NNEST_MH_DYNAMIC_BATCH needs sync_dev");
    hipError_t e = launch_mh(h->img, h->s, lk, z_dev, x_dev, logl_dev, loglstar, step_size, steps, C, flags,
                             noise_dz_dev, noise_u_dev, seed, walker_offset, hist_x_dev, hist_logl_dev, n_accept_dev, n_call_dev,
                             scale_out_dev, h->w, (unsigned long long *)sync_dev, h->num_cu, (hipStream_t)stream);
    if (e == hipErrorInvalidConfiguration)
        return fail(NNEST_E_UNSUPPORTED, "no kernel form for C=%d walkers with flags 0x%x (pinned form not applicable to this "
                                         "shape / population, or the batch-wide step rule on a grid that may not be resident)", C, flags);
    if (e != hipSuccess) return fail(NNEST_E_HIP, "launch_mh: %s", hipGetErrorString(e));
    return NNEST_OK;
}

int nnest_slice_steps(nnest_nvp_t *h, const nnest_like_t *like, float *z_dev, float *x_dev, double *logl_dev, double loglstar,
                      float width, int steps, int C, int max_stepout, int max_shrink, const float *noise_dz_dev, uint64_t seed,
                      uint64_t walker_offset, float *hist_x_dev, int *n_call_dev, int *n_move_dev, int *n_eval_dev, void *stream) {
    int rc = check_rows(h, z_dev, logl_dev, C);
    if (rc) return rc;
    LikeSpec lk;
    if ((rc = check_like(like, h->s.D, &lk))) return rc;
    if (steps < 0 || max_stepout < 0 || max_shrink < 1 || max_shrink > 60 || !(width > 0.f))
        return fail(NNEST_E_ARG, "steps=%d max_stepout=%d max_shrink=%d (1..60) width=%g", steps, max_stepout, max_shrink, (double)width);
    if (!slice_form_eligible(h->s))
        return fail(NNEST_E_UNSUPPORTED, "slice proposal: hidden 16, 3 blocks, 1 layer, scale '' (the one-walker-per-wave layout), x_dim <= 128");
    hipError_t e = launch_slice_solo(h->s, h->w, lk, z_dev, x_dev, logl_dev, loglstar, width, steps, C, max_stepout, max_shrink, seed,
                                     walker_offset, noise_dz_dev, hist_x_dev, n_call_dev, n_move_dev, n_eval_dev, (hipStream_t)stream);
    if (e != hipSuccess) return fail(NNEST_E_HIP, "launch_slice_solo: %s", hipGetErrorString(e));
    return NNEST_OK;
}

int nnest_slice_fill_noise(float *dz_dev, int steps, int C, int D, uint64_t seed, uint64_t walker_offset, void *stream) {
    if (!dz_dev || steps < 0 || C < 0 || D < 1) return fail(NNEST_E_ARG, "bad argument");
    HIP_TRY(launch_slice_fill_noise(dz_dev, steps, C, D, seed, walker_offset, (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_mh_form_for(const nnest_nvp_t *h, int C, int flags) {
    if (!h || C < 1) return -1;
    return mh_form_for(h->s, C, flags, h->num_cu);
}

int nnest_mh_sync_words(int steps) { return steps < 0 ? -1 : (int)mh_sync_words(steps) + 1; }

int nnest_mh_num_groups(const nnest_nvp_t *h, int C) {
    (void)h;
    return mh_num_groups(C);
}

int nnest_mh_fill_noise(float *dz_dev, float *u_dev, int steps, int C, int D, uint64_t seed, uint64_t walker_offset,
                        void *stream) {
    if (!dz_dev) return fail(NNEST_E_ARG, "NULL dz_dev");
    HIP_TRY(launch_fill_noise(dz_dev, u_dev, steps, C, D, seed, walker_offset, (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_nvp_train(nnest_nvp_t *h, const float *xtrain_dev, int n_train, const float *xvalid_dev, int n_valid,
                    const int *perm_dev, const float *noise_dev, uint64_t seed, float jitter, int batch,
                    int max_epochs, int patience, float lr, float weight_decay, int epoch_offset, int flags,
                    float *losses_dev, nnest_train_result_t *result_dev, void *stream) {
    if (!h) return fail(NNEST_E_ARG, "NULL handle");
    if (!xtrain_dev || !xvalid_dev || !perm_dev || !result_dev) return fail(NNEST_E_ARG, "NULL device buffer");
    if (n_train < 1 || n_valid < 1 || batch < 1 || max_epochs < 0)
        return fail(NNEST_E_ARG, "bad sizes n_train=%d n_valid=%d batch=%d max_epochs=%d", n_train, n_valid, batch, max_epochs);
    if (batch > 128) return fail(NNEST_E_UNSUPPORTED, "batch_size=%d > 128 (one workgroup holds a minibatch)", batch);
    if (h->s.kind == FLOW_KIND_MAF)
        return fail(NNEST_E_UNSUPPORTED, "maf: the epoch loop is driven from the host (nnest_nvp_loss_grad + nnest_nvp_adam_step per minibatch)");
    HIP_TRY(launch_train(h->w, h->adam_m, h->adam_v, h->best_w, h->img, h->adam_step, h->s, xtrain_dev, n_train, xvalid_dev,
                         n_valid, perm_dev, noise_dev, seed, jitter, batch, max_epochs, patience, lr, weight_decay,
                         epoch_offset, flags, losses_dev, result_dev, h->train_ws, h->fwd_pos, h->bwd_pos, (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_nvp_loss_grad(nnest_nvp_t *h, const float *x_dev, int M, float *grad_dev, float *loss_dev, void *stream) {
    if (!h || !x_dev || !grad_dev || !loss_dev) return fail(NNEST_E_ARG, "NULL argument");
    if (M < 1 || M > 128) return fail(NNEST_E_UNSUPPORTED, "M=%d outside [1,128]", M);
    if (h->s.kind == FLOW_KIND_MAF) {
        if (h->s.L > 2) return fail(NNEST_E_UNSUPPORTED, "maf: num_layers=%d > 2 has no training kernel", h->s.L);
        HIP_TRY(launch_maf_loss_grad(h->s, h->img, h->img_bwd, h->gpos, x_dev, M, grad_dev, loss_dev, h->train_ws, (hipStream_t)stream));
        return NNEST_OK;
    }
    HIP_TRY(launch_loss_grad(h->w, h->s, x_dev, M, grad_dev, loss_dev, h->train_ws, h->img, h->fwd_pos, h->bwd_pos,
                             (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_nvp_vjp(nnest_nvp_t *h, const float *x_dev, const float *gz_dev, float gld, int M, float *grad_dev, float *gx_dev, void *stream) {
    if (!h || !x_dev || !gz_dev || !grad_dev || !gx_dev) return fail(NNEST_E_ARG, "NULL argument");
    if (M < 1 || M > 128) return fail(NNEST_E_UNSUPPORTED, "M=%d outside [1,128]", M);
    if (h->s.kind == FLOW_KIND_MAF) return fail(NNEST_E_UNSUPPORTED, "maf: no vector-Jacobian product (it is not a stage of a fast/slow hierarchy)");
    HIP_TRY(launch_vjp(h->w, h->s, x_dev, gz_dev, gld, M, grad_dev, gx_dev, h->train_ws, h->img, h->fwd_pos, h->bwd_pos, (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_nvp_adam_step(nnest_nvp_t *h, const float *grad_dev, float lr, float weight_decay, void *stream) {
    if (!h || !grad_dev) return fail(NNEST_E_ARG, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(launch_adam_packed_dev(h->w, grad_dev, h->adam_m, h->adam_v, h->num_params, h->adam_step, lr, weight_decay, st));
    HIP_TRY(refresh_images(h, st));
    return NNEST_OK;
}

int nnest_maf_train_epoch(nnest_nvp_t *h, const float *rows_dev, int n_train, int batch, float lr, float weight_decay, float *loss_sum_dev,
                          void *stream) {
    if (!h || !rows_dev || !loss_sum_dev) return fail(NNEST_E_ARG, "NULL argument");
    if (h->s.kind != FLOW_KIND_MAF) return fail(NNEST_E_ARG, "not a MAF handle (nnest_maf_create)");
    if (n_train < 1 || batch < 1 || batch > 128) return fail(NNEST_E_UNSUPPORTED, "batch=%d outside [1,128]", batch);
    if (h->s.L > 2) return fail(NNEST_E_UNSUPPORTED, "maf: num_layers=%d > 2 has no training kernel", h->s.L);
    hipStream_t st = (hipStream_t)stream;
    for (int b0 = 0; b0 < n_train; b0 += batch) {   // Trainer._train's loop over the loader (trainer.py:387-403), queued back to back:
        const int M = n_train - b0 < batch ? n_train - b0 : batch;   // two launches per minibatch (gradient; reduce + Adam + images)
        HIP_TRY(launch_maf_train_minibatch(h->s, h->img, h->img_bwd, h->gpos, h->fwd_pos, h->bwd_pos, rows_dev + (size_t)b0 * h->s.D, M, h->w,
                                           h->adam_m, h->adam_v, h->adam_step, lr, weight_decay, loss_sum_dev, h->ticket, h->train_ws, st));
    }
    return NNEST_OK;
}

int nnest_training_jitter(const double *samples_dev, int N, int D, double *out_dev, void *stream) {
    if (!samples_dev || !out_dev || N < 2 || D < 1) return fail(NNEST_E_ARG, "bad arguments");
    HIP_TRY(launch_training_jitter(samples_dev, N, D, out_dev, (hipStream_t)stream));
    return NNEST_OK;
}

}  // extern "C"
