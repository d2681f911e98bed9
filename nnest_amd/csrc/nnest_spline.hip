// nnest_spline.hip -- host side of the neural-spline flow (reference nnest/networks.py:393-715): the handle, the
// weight-image builder and the extern "C" entry points (include/nnest_hip.h, "spline" section).
//
// Image build (host, float64, once per weight update; x_dim <= 128 so the matrices are tiny): per block the ActNorm
// vectors (networks.py:661-705) and the 1x1 convolution W = P (tril(L,-1)+I) (triu(U,1)+diag(S)) (networks.py:640-645)
// are folded into one affine map per direction,
//     forward  z = x A_f + b_f,  A_f = diag(e^s) W,        b_f = t W
//     inverse  x = z A_b + b_b,  A_b = W^-1 diag(e^-s),     b_b = -t e^-s
// with log|det| = sum(s) + sum(log|S|) kept as a per-block constant, and every matrix is laid out as MFMA A-operand
// fragments in the slot numbering of spline_tile.h.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "nnest_internal.h"

using namespace nnest;

#include "spline_host.h"

static thread_local char g_serr[512] = "";
int nnest::spline_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_serr, sizeof(g_serr), fmt, ap);
    va_end(ap);
    nnest::set_last_error(g_serr);  // read back through nnest_hip_last_error()
    return code;
}


int nnest::spline_mlp_params(int nin, int nout, int H) { return H * nin + H + 2 * (H * H + H) + nout * H + nout; }

static SplineShape make_shape(int D, int H, int B, int K, float tail) {
    SplineShape s;
    memset(&s, 0, sizeof(s));
    s.D = D; s.H = H; s.B = B; s.K = K; s.tail = tail;
    s.nu = D / 2; s.nl = D - s.nu;
    s.NTh = (s.nl + 15) / 16;
    s.NH = H / 16;
    s.SL = (s.nl + 3) / 4; s.SU = (s.nu + 3) / 4;
    s.aff_floats = (2 * s.NTh) * (2 * s.NTh) * 256 + 2 * s.NTh * 16;
    s.f1_floats = spl_cond_floats(s.NTh, s.NH, s.SU);
    s.f2_floats = spl_cond_floats(s.NTh, s.NH, s.SL);
    s.blk_floats = 2 * s.aff_floats + s.f1_floats + s.f2_floats + 4;
    s.image_floats = B * s.blk_floats;
    const int P = 3 * K - 1;
    s.blk_params = 2 * D + (2 * D * D + D) + spline_mlp_params(s.nl, P * s.nu, H) + spline_mlp_params(s.nu, P * s.nl, H);
    s.num_params = B * s.blk_params;
    s.base_beta = 0.f;
    s.base_const = -0.91893853320467274f;
    return s;
}

// slot -> dimension of the full vector (or -1): half hf, tile t, k-step / register r, lane group g
static inline int slot_dim(const SplineShape &s, int hf, int t, int r, int g) {
    const int j = 16 * t + 4 * r + g, n = hf ? s.nu : s.nl;
    return j < n ? (hf ? s.nl : 0) + j : -1;
}

// out = in M + b: weights [to][ti][r][lane], bias [to][g][r]
static void build_affine(const SplineShape &s, const std::vector<double> &M, const std::vector<double> &b, float *img) {
    const int NTh = s.NTh, T2 = 2 * NTh, D = s.D;
    for (int to = 0; to < T2; ++to)
        for (int ti = 0; ti < T2; ++ti)
            for (int r = 0; r < 4; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    const int g = lane >> 4, i = lane & 15;
                    const int di = slot_dim(s, ti / NTh, ti % NTh, r, g);
                    const int dout = slot_dim(s, to / NTh, to % NTh, i & 3, i >> 2);
                    img[(size_t)((to * T2 + ti) * 4 + r) * 64 + lane] = (di >= 0 && dout >= 0) ? (float)M[(size_t)di * D + dout] : 0.f;
                }
    float *bias = img + (size_t)T2 * T2 * 256;
    for (int to = 0; to < T2; ++to)
        for (int g = 0; g < 4; ++g)
            for (int r = 0; r < 4; ++r) {
                const int dout = slot_dim(s, to / NTh, to % NTh, r, g);
                bias[(to * 4 + g) * 4 + r] = dout >= 0 ? (float)b[dout] : 0.f;
            }
}

// conditioner MLP(nin -> H -> H -> H -> 23 nout)  (networks.py:393-409; NSF_CL :570-574)
static void build_cond(const SplineShape &s, const float *p, int nin, int nout, float *img) {
    const int H = s.H, NH = s.NH, NTh = s.NTh, S = (nout + 3) / 4, P = SPL_P;
    const float *W0 = p, *b0 = W0 + H * nin, *W1 = b0 + H, *b1 = W1 + H * H, *W2 = b1 + H, *b2 = W2 + H * H, *W3 = b2 + H,
                *b3 = W3 + (size_t)P * nout * H;
    float *L1 = img, *L2 = L1 + NH * NTh * 256, *L3 = L2 + NH * NH * 256, *bb = L3 + NH * NH * 256, *L4 = bb + 3 * 16 * NH,
          *B4 = L4 + (size_t)S * SPL_QT * NH * 256;
    for (int ht = 0; ht < NH; ++ht)
        for (int t = 0; t < NTh; ++t)
            for (int r = 0; r < 4; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    const int g = lane >> 4, i = lane & 15, j = 16 * t + 4 * r + g;
                    L1[(size_t)((ht * NTh + t) * 4 + r) * 64 + lane] = j < nin ? W0[(16 * ht + i) * nin + j] : 0.f;
                }
    for (int l = 0; l < 2; ++l) {
        const float *W = l ? W2 : W1;
        float *Lw = l ? L3 : L2;
        for (int hto = 0; hto < NH; ++hto)
            for (int hti = 0; hti < NH; ++hti)
                for (int r = 0; r < 4; ++r)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int g = lane >> 4, i = lane & 15;
                        Lw[(size_t)((hto * NH + hti) * 4 + r) * 64 + lane] = W[(16 * hto + i) * H + 16 * hti + 4 * g + r];
                    }
    }
    for (int j = 0; j < H; ++j) { bb[j] = b0[j]; bb[H + j] = b1[j]; bb[2 * H + j] = b2[j]; }
    for (int sidx = 0; sidx < S; ++sidx)
        for (int q = 0; q < SPL_QT; ++q) {
            for (int hti = 0; hti < NH; ++hti)
                for (int r = 0; r < 4; ++r)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int g = lane >> 4, i = lane & 15;
                        const int jo = 4 * sidx + (i >> 2), pp = 4 * q + (i & 3);
                        const bool ok = jo < nout && pp < P;
                        L4[(size_t)(((sidx * SPL_QT + q) * NH + hti) * 4 + r) * 64 + lane] =
                            ok ? W3[(size_t)(jo * P + pp) * H + 16 * hti + 4 * g + r] : 0.f;
                    }
            for (int g = 0; g < 4; ++g)
                for (int r = 0; r < 4; ++r) {
                    const int jo = 4 * sidx + g, pp = 4 * q + r;
                    B4[((sidx * SPL_QT + q) * 4 + g) * 4 + r] = (jo < nout && pp < P) ? b3[jo * P + pp] : 0.f;
                }
        }
}

static bool invert(std::vector<double> &A, int D, std::vector<double> &Ai) {  // Gauss-Jordan, partial pivoting
    Ai.assign((size_t)D * D, 0.0);
    for (int i = 0; i < D; ++i) Ai[(size_t)i * D + i] = 1.0;
    for (int c = 0; c < D; ++c) {
        int piv = c;
        for (int r = c + 1; r < D; ++r)
            if (fabs(A[(size_t)r * D + c]) > fabs(A[(size_t)piv * D + c])) piv = r;
        if (A[(size_t)piv * D + c] == 0.0) return false;
        if (piv != c)
            for (int j = 0; j < D; ++j) {
                std::swap(A[(size_t)c * D + j], A[(size_t)piv * D + j]);
                std::swap(Ai[(size_t)c * D + j], Ai[(size_t)piv * D + j]);
            }
        const double inv = 1.0 / A[(size_t)c * D + c];
        for (int j = 0; j < D; ++j) { A[(size_t)c * D + j] *= inv; Ai[(size_t)c * D + j] *= inv; }
        for (int r = 0; r < D; ++r) {
            if (r == c) continue;
            const double f = A[(size_t)r * D + c];
            if (f == 0.0) continue;
            for (int j = 0; j < D; ++j) { A[(size_t)r * D + j] -= f * A[(size_t)c * D + j]; Ai[(size_t)r * D + j] -= f * Ai[(size_t)c * D + j]; }
        }
    }
    return true;
}

int nnest::spline_build_image(nnest_spline *h) {
    const SplineShape &s = h->s;
    const int D = s.D;
    h->img_host.assign((size_t)s.image_floats, 0.f);
    std::vector<double> Lm((size_t)D * D), PL((size_t)D * D), W((size_t)D * D), Wc, Wi, M((size_t)D * D), bvec(D);
    for (int b = 0; b < s.B; ++b) {
        const float *pb = h->w.data() + (size_t)b * s.blk_params;
        const float *sv = pb, *tv = pb + D, *Lp = pb + 2 * D, *Sp = Lp + D * D, *Up = Sp + D, *f1 = Up + D * D;
        const float *f2 = f1 + spline_mlp_params(s.nl, SPL_P * s.nu, s.H);
        const float *Pm = h->perm.data() + (size_t)b * D * D;
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) Lm[(size_t)i * D + j] = j < i ? (double)Lp[i * D + j] : (i == j ? 1.0 : 0.0);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                double acc = 0;
                for (int k = 0; k < D; ++k) acc += (double)Pm[i * D + k] * Lm[(size_t)k * D + j];
                PL[(size_t)i * D + j] = acc;
            }
        double ldc = 0;
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                double acc = 0;
                for (int k = 0; k <= j; ++k) acc += PL[(size_t)i * D + k] * (k < j ? (double)Up[k * D + j] : (double)Sp[k]);
                W[(size_t)i * D + j] = acc;
            }
        for (int d = 0; d < D; ++d) ldc += (double)sv[d] + log(fabs((double)Sp[d]));
        Wc = W;
        if (!invert(Wc, D, Wi)) return spline_fail(NNEST_E_ARG, "block %d: the 1x1 convolution matrix is singular", b);
        float *blk = h->img_host.data() + (size_t)b * s.blk_floats;
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) M[(size_t)i * D + j] = exp((double)sv[i]) * W[(size_t)i * D + j];
        for (int j = 0; j < D; ++j) {
            double acc = 0;
            for (int i = 0; i < D; ++i) acc += (double)tv[i] * W[(size_t)i * D + j];
            bvec[j] = acc;
        }
        build_affine(s, M, bvec, blk);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) M[(size_t)i * D + j] = Wi[(size_t)i * D + j] * exp(-(double)sv[j]);
        for (int j = 0; j < D; ++j) bvec[j] = -(double)tv[j] * exp(-(double)sv[j]);
        build_affine(s, M, bvec, blk + s.aff_floats);
        build_cond(s, f1, s.nl, s.nu, blk + 2 * s.aff_floats);
        build_cond(s, f2, s.nu, s.nl, blk + 2 * s.aff_floats + s.f1_floats);
        blk[2 * s.aff_floats + s.f1_floats + s.f2_floats] = (float)ldc;
    }
    return NNEST_OK;
}

extern "C" {

int nnest_spline_create(int D, int H, int B, int K, float tail_bound, nnest_spline_t **out) {
    if (!out) return spline_fail(NNEST_E_ARG, "out is NULL");
    *out = nullptr;
    if (D < 2 || H < 1 || B < 1 || K < 1 || !(tail_bound > 0)) return spline_fail(NNEST_E_ARG, "bad shape D=%d H=%d B=%d K=%d", D, H, B, K);
    if (H % 16 != 0) return spline_fail(NNEST_E_UNSUPPORTED, "hidden_dim=%d: the gfx950 kernels tile the hidden layers by 16", H);
    SplineShape s = make_shape(D, H, B, K, tail_bound);
    if (!spline_shape_supported(s))
        return spline_fail(NNEST_E_UNSUPPORTED, "spline flow: x_dim=%d hidden_dim=%d num_bins=%d not instantiated (num_bins 8; x_dim <= 128 at "
                     "hidden_dim 16, <= 64 at 32)", D, H, K);
    nnest_spline *h = new nnest_spline();
    h->s = s;
    h->img = nullptr;
    if (hipGetDevice(&h->device) != hipSuccess) { delete h; return spline_fail(NNEST_E_HIP, "hipGetDevice failed (no GPU?)"); }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, h->device) != hipSuccess) { delete h; return spline_fail(NNEST_E_HIP, "hipGetDeviceProperties failed"); }
    h->num_cu = p.multiProcessorCount;
    h->w.assign((size_t)s.num_params, 0.f);
    h->perm.assign((size_t)B * D * D, 0.f);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < D; ++i) h->perm[((size_t)b * D + i) * D + i] = 1.f;
    if (hipMalloc((void **)&h->img, (size_t)s.image_floats * sizeof(float)) != hipSuccess) {
        delete h;
        return spline_fail(NNEST_E_HIP, "device allocation failed");
    }
    *out = h;
    return NNEST_OK;
}

int nnest_spline_destroy(nnest_spline_t *h) {
    if (!h) return NNEST_OK;
    (void)hipFree(h->img);
    (void)hipFree(h->w_dev); (void)hipFree(h->adam_m); (void)hipFree(h->adam_v); (void)hipFree(h->best_w); (void)hipFree(h->pi_dev); (void)hipFree(h->pos_dev);
    (void)hipFree(h->wmat); (void)hipFree(h->timg); (void)hipFree(h->partial); (void)hipFree(h->grad); (void)hipFree(h->gwsum);
    (void)hipFree(h->stash); (void)hipFree(h->gbuf); (void)hipFree(h->hbuf); (void)hipFree(h->keep); (void)hipFree(h->losses_dev); (void)hipFree(h->ctl_dev); (void)hipFree(h->epoch_losses_dev);
    if (h->ctl_host) (void)hipHostFree(h->ctl_host);
    spline_rows_free(h);
    delete h;
    return NNEST_OK;
}

int nnest_spline_num_params(const nnest_spline_t *h) { return h ? h->s.num_params : -1; }

int nnest_spline_set_base(nnest_spline_t *h, float beta) {
    if (!h) return spline_fail(NNEST_E_ARG, "NULL handle");
    if (!(beta >= 0.f)) return spline_fail(NNEST_E_ARG, "beta=%g: 0 selects N(0, I), beta > 0 GeneralisedNormal(0, 1, beta)", (double)beta);
    h->s.base_beta = beta;
    h->s.base_const = beta == 0.f ? -0.91893853320467274f : (float)(log((double)beta) - log(2.0) - lgamma(1.0 / (double)beta));
    return NNEST_OK;
}

int nnest_spline_load_weights(nnest_spline_t *h, const float *packed_host, const float *perm_host, void *stream) {
    if (!h || !packed_host) return spline_fail(NNEST_E_ARG, "NULL argument");
    memcpy(h->w.data(), packed_host, h->w.size() * sizeof(float));
    if (perm_host) memcpy(h->perm.data(), perm_host, h->perm.size() * sizeof(float));
    h->w_dev_current = false;  // the training copy on the device is refreshed on its next use
    int rc = spline_build_image(h);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    SHIP_TRY(hipMemcpyAsync(h->img, h->img_host.data(), h->img_host.size() * sizeof(float), hipMemcpyHostToDevice, st));
    SHIP_TRY(hipStreamSynchronize(st));
    return NNEST_OK;
}

int nnest_spline_store_weights(nnest_spline_t *h, float *packed_host, float *perm_host, void *stream) {
    (void)stream;
    if (!h) return spline_fail(NNEST_E_ARG, "NULL handle");
    if (packed_host) memcpy(packed_host, h->w.data(), h->w.size() * sizeof(float));
    if (perm_host) memcpy(perm_host, h->perm.data(), h->perm.size() * sizeof(float));
    return NNEST_OK;
}

static int scheck_rows(const nnest_spline_t *h, const void *a, const void *b, int N) {
    if (!h) return spline_fail(NNEST_E_ARG, "NULL handle");
    if (N < 0) return spline_fail(NNEST_E_ARG, "N=%d < 0", N);
    if (N > 0 && (!a || !b)) return spline_fail(NNEST_E_ARG, "NULL device buffer");
    return NNEST_OK;
}

static int scheck_like(const nnest_like_t *like, int D, LikeSpec *out) {
    if (!like) return spline_fail(NNEST_E_ARG, "like is NULL");
    if (like->id < 0 || like->id >= NNEST_LIKE_COUNT) return spline_fail(NNEST_E_ARG, "unknown likelihood id %d", like->id);
    if (like->id == NNEST_LIKE_EGGBOX && D != 2) return spline_fail(NNEST_E_ARG, "Eggbox is defined for x_dim = 2 (likelihoods.py:97-102)");
    if (like->id == NNEST_LIKE_GAUSSMIX && D < 2) return spline_fail(NNEST_E_ARG, "GaussianMix needs x_dim >= 2");
    out->id = like->id;
    out->scale = like->scale;
    for (int i = 0; i < 6; ++i) out->p[i] = like->params[i];
    return NNEST_OK;
}

int nnest_spline_forward(nnest_spline_t *h, const float *x_dev, float *z_dev, float *logdet_dev, int N, void *stream) {
    int rc = scheck_rows(h, x_dev, z_dev, N);
    if (rc) return rc;
    SHIP_TRY(launch_spline_pass(h->img, h->s, PASS_FORWARD, x_dev, z_dev, logdet_dev, nullptr, nullptr, N, LikeSpec(), h->num_cu,
                                (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_spline_inverse(nnest_spline_t *h, const float *z_dev, float *x_dev, float *logdet_dev, int N, void *stream) {
    int rc = scheck_rows(h, z_dev, x_dev, N);
    if (rc) return rc;
    SHIP_TRY(launch_spline_pass(h->img, h->s, PASS_INVERSE, z_dev, x_dev, logdet_dev, nullptr, nullptr, N, LikeSpec(), h->num_cu,
                                (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_spline_log_probs(nnest_spline_t *h, const float *x_dev, float *logp_dev, int N, void *stream) {
    int rc = scheck_rows(h, x_dev, logp_dev, N);
    if (rc) return rc;
    SHIP_TRY(launch_spline_pass(h->img, h->s, PASS_LOGPROB, x_dev, logp_dev, nullptr, nullptr, nullptr, N, LikeSpec(), h->num_cu,
                                (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_spline_inverse_loglike(nnest_spline_t *h, const nnest_like_t *like, const float *z_dev, float *x_dev,
                                 float *logdet_dev, double *logl_dev, int *inbox_dev, int N, void *stream) {
    int rc = scheck_rows(h, z_dev, logl_dev, N);
    if (rc) return rc;
    LikeSpec lk;
    if ((rc = scheck_like(like, h->s.D, &lk))) return rc;
    SHIP_TRY(launch_spline_pass(h->img, h->s, PASS_INVERSE_LOGLIKE, z_dev, x_dev, logdet_dev, logl_dev, inbox_dev, N, lk, h->num_cu,
                                (hipStream_t)stream));
    return NNEST_OK;
}

int nnest_spline_mh_form_for(const nnest_spline_t *h, int C, int flags) {
    if (!h || C <= 0) return -1;
    return spline_mh_form(h->s, C, flags, h->num_cu);
}

int nnest_spline_mh_constrained_steps(nnest_spline_t *h, const nnest_like_t *like, float *z_dev, float *x_dev, double *logl_dev,
                                      double loglstar, float step_size, int steps, int C, int flags, const float *noise_dz_dev,
                                      const float *noise_u_dev, uint64_t seed, uint64_t walker_offset, float *hist_x_dev,
                                      double *hist_logl_dev, int *n_accept_dev, int *n_call_dev, float *scale_out_dev,
                                      void *sync_dev, void *stream) {
    int rc = scheck_rows(h, z_dev, logl_dev, C);
    if (rc) return rc;
    LikeSpec lk;
    if ((rc = scheck_like(like, h->s.D, &lk))) return rc;
    if (steps < 0) return spline_fail(NNEST_E_ARG, "steps=%d < 0", steps);
    if ((noise_dz_dev == nullptr) != (noise_u_dev == nullptr))
        return spline_fail(NNEST_E_ARG, "noise_dz_dev and noise_u_dev must both be given or both be NULL");
    if ((flags & NNEST_MH_DYNAMIC_BATCH) && !sync_dev) return spline_fail(NNEST_E_ARG, "NNEST_MH_DYNAMIC_BATCH needs sync_dev");
    hipError_t e = launch_spline_mh(h->img, h->s, lk, z_dev, x_dev, logl_dev, loglstar, step_size, steps, C, flags, noise_dz_dev,
                                    noise_u_dev, seed, walker_offset, hist_x_dev, hist_logl_dev, n_accept_dev, n_call_dev,
                                    scale_out_dev, (unsigned long long *)sync_dev, h->num_cu, (hipStream_t)stream);
    if (e == hipErrorInvalidConfiguration)
        return spline_fail(NNEST_E_UNSUPPORTED, "batch-wide step rule: C=%d walkers do not fit a resident grid", C);
    if (e != hipSuccess) return spline_fail(NNEST_E_HIP, "launch_spline_mh: %s", hipGetErrorString(e));
    return NNEST_OK;
}

}  // extern "C"
