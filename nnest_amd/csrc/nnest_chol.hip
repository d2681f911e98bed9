// nnest_chol.hip -- the reference's 'choleksy' flow (SingleSpeedCholeksy, nnest/networks.py:162-239): ONE linear map
// y = L x + b with L lower triangular, diag(L) = softplus(unconstrained_diag) + eps, log|det| = sum log diag.
// D^2 / 2 parameters and D^2 / 2 MACs per row: plain row-parallel kernels (no matrix cores needed at x_dim <= 128).
// Packed weights = state_dict order: bias[D], lower_entries[D(D-1)/2] (np.tril_indices(D, -1) order: row-major), unconstrained_diag[D].
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "nnest_internal.h"

using namespace nnest;

struct nnest_chol {
    int D, num_params, num_cu;
    float eps;
    float base_beta, base_const;
    float *w, *adam_m, *adam_v, *ws;  // ws: y / gy rows of a minibatch [2][128][D]
    int adam_step;
};

static int cfail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    nnest::set_last_error(buf);
    return code;
}
#define CHIP_TRY(expr)                                                                              \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess) return cfail(NNEST_E_HIP, "%s: %s", #expr, hipGetErrorString(e__));  \
    } while (0)

namespace {

__device__ __forceinline__ float c_softplus(float v) { return v > 20.f ? v : log1pf(expf(v)); }
__device__ __forceinline__ float c_base_E(float u, float beta) { return beta == 0.f ? 0.5f * u * u : powf(fabsf(u), beta); }
__device__ __forceinline__ float c_base_dE(float u, float beta) {
    if (beta == 0.f) return u;
    return u == 0.f ? 0.f : beta * powf(fabsf(u), beta) / u;
}

enum { CHOL_FORWARD = 0, CHOL_INVERSE = 1, CHOL_LOGPROB = 2 };

// one thread per row (networks.py:202-214)
__global__ void chol_pass_kernel(const float *__restrict__ w, int D, float eps, int mode, const float *__restrict__ in, float *__restrict__ out,
                                 float *__restrict__ ld, int N, float beta, float cst) {
    const float *bias = w, *lower = w + D, *ud = lower + D * (D - 1) / 2;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float logdet = 0.f;
    for (int i = 0; i < D; ++i) logdet += logf(c_softplus(ud[i]) + eps);
    const float *x = in + (size_t)n * D;
    if (mode == CHOL_INVERSE) {
        float *o = out + (size_t)n * D;
        for (int i = 0; i < D; ++i) {  // forward substitution (torch.triangular_solve, networks.py:210)
            float acc = x[i] - bias[i];
            const float *Li = lower + i * (i - 1) / 2;
            for (int j = 0; j < i; ++j) acc -= Li[j] * o[j];
            o[i] = acc / (c_softplus(ud[i]) + eps);
        }
        if (ld) ld[n] = -logdet;
        return;
    }
    float e = 0.f;
    for (int i = 0; i < D; ++i) {
        const float *Li = lower + i * (i - 1) / 2;
        float acc = bias[i] + (c_softplus(ud[i]) + eps) * x[i];
        for (int j = 0; j < i; ++j) acc += Li[j] * x[j];
        if (mode == CHOL_FORWARD) out[(size_t)n * D + i] = acc;
        else e += c_base_E(acc, beta);
    }
    if (mode == CHOL_FORWARD) { if (ld) ld[n] = logdet; }
    else out[n] = -e + cst * (float)D + logdet;
}

// minibatch rows: y = L x + b and gy = dE/dy / M; also the sum of log_probs
__global__ void chol_rows_kernel(const float *__restrict__ w, int D, float eps, const float *__restrict__ x, int M, float beta, float cst,
                                 float *__restrict__ gy, float *__restrict__ loss) {
    __shared__ float red[128];
    const float *bias = w, *lower = w + D, *ud = lower + D * (D - 1) / 2;
    const int n = threadIdx.x;
    float lp = 0.f;
    if (n < M) {
        float logdet = 0.f, e = 0.f;
        for (int i = 0; i < D; ++i) logdet += logf(c_softplus(ud[i]) + eps);
        for (int i = 0; i < D; ++i) {
            const float *Li = lower + i * (i - 1) / 2;
            float acc = bias[i] + (c_softplus(ud[i]) + eps) * x[(size_t)n * D + i];
            for (int j = 0; j < i; ++j) acc += Li[j] * x[(size_t)n * D + j];
            e += c_base_E(acc, beta);
            gy[(size_t)n * D + i] = c_base_dE(acc, beta) / (float)M;
        }
        lp = -e + cst * (float)D + logdet;
    }
    red[n] = lp;
    __syncthreads();
    if (n == 0) {
        float tot = 0.f;
        for (int k = 0; k < M; ++k) tot += red[k];
        *loss = -tot / (float)M;
    }
}

// one thread per parameter: contraction over the rows
__global__ void chol_grad_kernel(const float *__restrict__ w, int D, float eps, const float *__restrict__ x, const float *__restrict__ gy, int M,
                                 float *__restrict__ grad) {
    const int nl = D * (D - 1) / 2, np = 2 * D + nl;
    const float *ud = w + D + nl;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < np; p += gridDim.x * blockDim.x) {
        float acc = 0.f;
        if (p < D) {  // bias
            for (int r = 0; r < M; ++r) acc += gy[(size_t)r * D + p];
        } else if (p < D + nl) {  // lower[i][j], j < i
            const int q = p - D;
            int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)q)) * 0.5f);
            while (i * (i - 1) / 2 > q) --i;
            while ((i + 1) * i / 2 <= q) ++i;
            const int j = q - i * (i - 1) / 2;
            for (int r = 0; r < M; ++r) acc += gy[(size_t)r * D + i] * x[(size_t)r * D + j];
        } else {  // unconstrained_diag: d/d diag = sum gy_i x_i - 1 / diag_i ; d diag / d u = sigmoid(u)
            const int i = p - D - nl;
            for (int r = 0; r < M; ++r) acc += gy[(size_t)r * D + i] * x[(size_t)r * D + i];
            const float u = ud[i];
            acc = (acc - 1.0f / (c_softplus(u) + eps)) * (1.0f / (1.0f + expf(-u)));
        }
        grad[p] = acc;
    }
}

}  // namespace

extern "C" {

int nnest_chol_create(int D, nnest_chol_t **out) {
    if (!out) return cfail(NNEST_E_ARG, "out is NULL");
    *out = nullptr;
    if (D < 1 || D > 128) return cfail(NNEST_E_UNSUPPORTED, "choleksy flow: x_dim=%d outside [1, 128]", D);
    nnest_chol *h = new nnest_chol();
    memset(h, 0, sizeof(*h));
    h->D = D;
    h->eps = 1e-3f;
    h->num_params = 2 * D + D * (D - 1) / 2;
    h->base_const = -0.91893853320467274f;
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) { delete h; return cfail(NNEST_E_HIP, "no GPU"); }
    h->num_cu = p.multiProcessorCount;
    const size_t nb = (size_t)h->num_params * sizeof(float);
    hipError_t e = hipMalloc((void **)&h->w, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->adam_m, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->adam_v, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->ws, (size_t)128 * D * sizeof(float) + 16);
    if (e == hipSuccess) e = hipMemset(h->w, 0, nb);
    if (e == hipSuccess) e = hipMemset(h->adam_m, 0, nb);
    if (e == hipSuccess) e = hipMemset(h->adam_v, 0, nb);
    if (e != hipSuccess) { nnest_chol_destroy(h); return cfail(NNEST_E_HIP, "device allocation failed"); }
    *out = h;
    return NNEST_OK;
}

int nnest_chol_destroy(nnest_chol_t *h) {
    if (!h) return NNEST_OK;
    (void)hipFree(h->w); (void)hipFree(h->adam_m); (void)hipFree(h->adam_v); (void)hipFree(h->ws);
    delete h;
    return NNEST_OK;
}

int nnest_chol_num_params(const nnest_chol_t *h) { return h ? h->num_params : -1; }

int nnest_chol_set_base(nnest_chol_t *h, float beta) {
    if (!h || !(beta >= 0.f)) return cfail(NNEST_E_ARG, "bad argument");
    h->base_beta = beta;
    h->base_const = beta == 0.f ? -0.91893853320467274f : (float)(log((double)beta) - log(2.0) - lgamma(1.0 / (double)beta));
    return NNEST_OK;
}

int nnest_chol_load_weights(nnest_chol_t *h, const float *packed_host, void *stream) {
    if (!h || !packed_host) return cfail(NNEST_E_ARG, "NULL argument");
    CHIP_TRY(hipMemcpyAsync(h->w, packed_host, (size_t)h->num_params * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
    CHIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return NNEST_OK;
}

int nnest_chol_store_weights(nnest_chol_t *h, float *packed_host, void *stream) {
    if (!h || !packed_host) return cfail(NNEST_E_ARG, "NULL argument");
    CHIP_TRY(hipMemcpyAsync(packed_host, h->w, (size_t)h->num_params * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    CHIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return NNEST_OK;
}

static int chol_pass(nnest_chol_t *h, int mode, const float *in, float *out, float *ld, int N, void *stream) {
    if (!h) return cfail(NNEST_E_ARG, "NULL handle");
    if (N < 0 || (N > 0 && (!in || !out))) return cfail(NNEST_E_ARG, "bad buffers");
    if (N == 0) return NNEST_OK;
    hipLaunchKernelGGL(chol_pass_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->w, h->D, h->eps, mode, in, out, ld, N,
                       h->base_beta, h->base_const);
    CHIP_TRY(hipGetLastError());
    return NNEST_OK;
}

int nnest_chol_forward(nnest_chol_t *h, const float *x_dev, float *z_dev, float *logdet_dev, int N, void *stream) {
    return chol_pass(h, CHOL_FORWARD, x_dev, z_dev, logdet_dev, N, stream);
}
int nnest_chol_inverse(nnest_chol_t *h, const float *z_dev, float *x_dev, float *logdet_dev, int N, void *stream) {
    return chol_pass(h, CHOL_INVERSE, z_dev, x_dev, logdet_dev, N, stream);
}
int nnest_chol_log_probs(nnest_chol_t *h, const float *x_dev, float *logp_dev, int N, void *stream) {
    return chol_pass(h, CHOL_LOGPROB, x_dev, logp_dev, nullptr, N, stream);
}

int nnest_chol_loss_grad(nnest_chol_t *h, const float *x_dev, int M, float *grad_dev, float *loss_dev, void *stream) {
    if (!h || !x_dev || !grad_dev || !loss_dev) return cfail(NNEST_E_ARG, "NULL argument");
    if (M < 1 || M > 128) return cfail(NNEST_E_UNSUPPORTED, "M=%d outside [1,128]", M);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(chol_rows_kernel, dim3(1), dim3(128), 0, st, h->w, h->D, h->eps, x_dev, M, h->base_beta, h->base_const, h->ws, loss_dev);
    hipLaunchKernelGGL(chol_grad_kernel, dim3(32), dim3(256), 0, st, h->w, h->D, h->eps, x_dev, h->ws, M, grad_dev);
    CHIP_TRY(hipGetLastError());
    return NNEST_OK;
}

int nnest_chol_adam_step(nnest_chol_t *h, const float *grad_dev, float lr, float weight_decay, void *stream) {
    if (!h || !grad_dev) return cfail(NNEST_E_ARG, "NULL argument");
    h->adam_step += 1;
    CHIP_TRY(launch_adam_packed(h->w, grad_dev, h->adam_m, h->adam_v, h->num_params, h->adam_step, lr, weight_decay, (hipStream_t)stream));
    return NNEST_OK;
}

}  // extern "C"
