/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C) of the nnest hot path, one function per reference
 * function, each citing the reference file:line it follows (paths relative to
 * /root/reference, adammoss/nnest v0.4.2).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this.
 *
 * This header is included twice by nnest_oracle.c with
 *     REAL = float   / FN(x) = orc32_##x      (the reference's own precision)
 *     REAL = double  / FN(x) = orc64_##x      (rounding-noise yardstick)
 *
 * Parity status: PINNED.  Checked by tests/test_oracle_golden.py against fixtures
 * produced by running the reference itself (oracle/gen_golden.py -> tests/golden/).
 *
 * Packed weight layout = torch state_dict order: for block b,
 *   scale_net:     W0[H,D] b0[H] (W[H,H] b[H])xL  Wout[D,H] bout[D]
 *   translate_net: same shapes
 * nn.Linear weight is [out,in] row-major.
 *
 * scale variants of SingleSpeedNVP (networks.py:328-347), selected with orc_set_scale_mode():
 *   0  scale=''           full affine coupling
 *   1  scale='translate'  CouplingLayer(translate_only=True): y = x + t, logdet 0 (networks.py:293-294, :304-305)
 *   2  scale='constant'   translate-only couplings, each followed by a ScaleLayer (networks.py:312-325):
 *                         y = x e^s for every dimension, logdet += s  (the scalar itself, not D s)
 * In modes 1 and 2 the packed vector keeps the scale_net slots (unused; zero, zero gradient) so that one layout
 * serves every mode; in mode 2 the B ScaleLayer scalars follow the B blocks.
 */

#ifndef REAL
#error "include from nnest_oracle.c"
#endif

/* per-net parameter count */
static int FN(net_size)(int D, int H, int L) { return H * D + H + L * (H * H + H) + D * H + D; }

/* orc_flow_kind = 1: the masked autoregressive flow of maf_oracle_impl.h (same parameter layout); the four entry points
 * below hand over to it, so everything built on them (training steps, the Metropolis loop) runs either flow */
void FN(maf_forward)(const float *w, int D, int H, int B, int L, const REAL *x, int N, REAL *z, REAL *logdet);
void FN(maf_inverse)(const float *w, int D, int H, int B, int L, const REAL *z, int N, REAL *x, REAL *logdet);
void FN(maf_log_probs)(const float *w, int D, int H, int B, int L, const REAL *x, int N, REAL *lp);
double FN(maf_loss_grad)(const float *w, int D, int H, int B, int L, const REAL *X, int M, REAL *grad);

/* y[out] = W[out,in] x[in] + b   (nn.Linear, networks.py:271-282) */
static void FN(linear)(const float *W, const float *b, int out, int in, const REAL *x, REAL *y) {
    for (int o = 0; o < out; ++o) {
        REAL acc = (REAL)b[o];
        const float *w = W + (size_t)o * in;
        for (int i = 0; i < in; ++i) acc += (REAL)w[i] * x[i];
        y[o] = acc;
    }
}

static REAL FN(tanhr)(REAL v) { return sizeof(REAL) == 4 ? (REAL)tanhf((float)v) : (REAL)tanh((double)v); }

/* One MLP of a coupling layer (networks.py:271-282): Linear(D,H) act [Linear(H,H) act]xL Linear(H,D).
 * act: 0 = tanh (scale_net), 1 = relu (translate_net).  `acts` (optional) receives the post-activation
 * hidden vectors, (L+1) x H, for the backward pass. */
static void FN(mlp)(const float *p, int D, int H, int L, int act, const REAL *m, REAL *out, REAL *acts) {
    REAL h[256], h2[256];
    FN(linear)(p, p + H * D, H, D, m, h2);
    for (int j = 0; j < H; ++j) h[j] = act == 0 ? FN(tanhr)(h2[j]) : (h2[j] > 0 ? h2[j] : (REAL)0);
    if (acts) memcpy(acts, h, sizeof(REAL) * H);
    p += H * D + H;
    for (int l = 0; l < L; ++l) {
        FN(linear)(p, p + H * H, H, H, h, h2);
        for (int j = 0; j < H; ++j) h[j] = act == 0 ? FN(tanhr)(h2[j]) : (h2[j] > 0 ? h2[j] : (REAL)0);
        if (acts) memcpy(acts + (size_t)(l + 1) * H, h, sizeof(REAL) * H);
        p += H * H + H;
    }
    FN(linear)(p, p + D * H, D, H, h, out);
}

static REAL FN(expr)(REAL v) { return sizeof(REAL) == 4 ? (REAL)expf((float)v) : (REAL)exp((double)v); }

/* CouplingLayer.forward for one row (networks.py:289-298).
 * mask (networks.py:333-334, :346): block b conditions on dims with (d + b) odd. */
static REAL FN(coupling_fwd_row)(const float *pb, int D, int H, int L, int b, REAL *x /* in/out */, REAL *ls_out,
                                 REAL *acts_s, REAL *acts_t) {
    REAL m[512], ls[512], t[512];
    int ns = FN(net_size)(D, H, L);
    for (int d = 0; d < D; ++d) m[d] = ((d + b) & 1) ? x[d] : (REAL)0; /* inputs * mask */
    if (orc_scale_mode == 0) FN(mlp)(pb, D, H, L, 0, m, ls, acts_s);
    else for (int d = 0; d < D; ++d) ls[d] = 0; /* translate_only: y = x + t, logdet 0 */
    FN(mlp)(pb + ns, D, H, L, 1, m, t, acts_t);
    REAL ld = 0;
    for (int d = 0; d < D; ++d) {
        if ((d + b) & 1) { /* (1 - mask) = 0: log_s = t = 0, x passes through bit-exactly */
            if (ls_out) ls_out[d] = 0;
            continue;
        }
        x[d] = x[d] * FN(expr)(ls[d]) + t[d];
        ld += ls[d];
        if (ls_out) ls_out[d] = ls[d];
    }
    return ld;
}

/* CouplingLayer.inverse for one row (networks.py:300-309) */
static REAL FN(coupling_inv_row)(const float *pb, int D, int H, int L, int b, REAL *x) {
    REAL m[512], ls[512], t[512];
    int ns = FN(net_size)(D, H, L);
    for (int d = 0; d < D; ++d) m[d] = ((d + b) & 1) ? x[d] : (REAL)0;
    if (orc_scale_mode == 0) FN(mlp)(pb, D, H, L, 0, m, ls, NULL);
    else for (int d = 0; d < D; ++d) ls[d] = 0;
    FN(mlp)(pb + ns, D, H, L, 1, m, t, NULL);
    REAL ld = 0;
    for (int d = 0; d < D; ++d) {
        if ((d + b) & 1) continue;
        x[d] = (x[d] - t[d]) * FN(expr)(-ls[d]);
        ld -= ls[d];
    }
    return ld;
}

/* ScaleLayer.forward / .inverse (networks.py:318-325) after / before coupling b; s_b = w[B*bs + b] */
static REAL FN(scale_fwd_row)(const float *w, int D, int B, int bs, int b, REAL *x) {
    if (orc_scale_mode != 2) return 0;
    REAL s = (REAL)w[(size_t)B * bs + b], e = FN(expr)(s);
    for (int d = 0; d < D; ++d) x[d] *= e;
    return s;
}
static REAL FN(scale_inv_row)(const float *w, int D, int B, int bs, int b, REAL *x) {
    if (orc_scale_mode != 2) return 0;
    REAL s = (REAL)w[(size_t)B * bs + b], e = FN(expr)(-s);
    for (int d = 0; d < D; ++d) x[d] *= e;
    return -s;
}

/* NormalizingFlow.forward (networks.py:24-32): blocks 0..B-1, log_det accumulated */
void FN(nvp_forward)(const float *w, int D, int H, int B, int L, const REAL *x, int N, REAL *z, REAL *logdet) {
    if (orc_flow_kind == 1) { FN(maf_forward)(w, D, H, B, L, x, N, z, logdet); return; }
    int bs = 2 * FN(net_size)(D, H, L);
    for (int n = 0; n < N; ++n) {
        REAL r[512];
        for (int d = 0; d < D; ++d) r[d] = x[(size_t)n * D + d];
        REAL ld = 0;
        for (int b = 0; b < B; ++b) {
            ld += FN(coupling_fwd_row)(w + (size_t)b * bs, D, H, L, b, r, NULL, NULL, NULL);
            ld += FN(scale_fwd_row)(w, D, B, bs, b, r);
        }
        for (int d = 0; d < D; ++d) z[(size_t)n * D + d] = r[d];
        logdet[n] = ld;
    }
}

/* NormalizingFlow.inverse (networks.py:34-42): blocks reversed */
void FN(nvp_inverse)(const float *w, int D, int H, int B, int L, const REAL *z, int N, REAL *x, REAL *logdet) {
    if (orc_flow_kind == 1) { FN(maf_inverse)(w, D, H, B, L, z, N, x, logdet); return; }
    int bs = 2 * FN(net_size)(D, H, L);
    for (int n = 0; n < N; ++n) {
        REAL r[512];
        for (int d = 0; d < D; ++d) r[d] = z[(size_t)n * D + d];
        REAL ld = 0;
        for (int b = B - 1; b >= 0; --b) {
            ld += FN(scale_inv_row)(w, D, B, bs, b, r);
            ld += FN(coupling_inv_row)(w + (size_t)b * bs, D, H, L, b, r);
        }
        for (int d = 0; d < D; ++d) x[(size_t)n * D + d] = r[d];
        logdet[n] = ld;
    }
}


/* log density of the base distribution at u[D] and (optionally) d(-log density)/du */
static REAL FN(base_logp)(const REAL *u, int D, REAL *gneg) {
    if (orc_base_beta == 0.0) {
        const double half_log_2pi = 0.91893853320467274178;
        REAL ss = 0;
        for (int d = 0; d < D; ++d) { ss += u[d] * u[d]; if (gneg) gneg[d] = u[d]; }
        return (REAL)(-0.5) * ss - (REAL)(D * half_log_2pi);
    }
    const REAL beta = (REAL)orc_base_beta;
    const REAL cst = (REAL)(log(orc_base_beta) - log(2.0) - lgamma(1.0 / orc_base_beta));
    REAL acc = 0;
    for (int d = 0; d < D; ++d) {
        REAL a = u[d] < 0 ? -u[d] : u[d];
        REAL p = sizeof(REAL) == 4 ? (REAL)powf((float)a, (float)beta) : (REAL)pow((double)a, (double)beta);
        acc += -p + cst;
        if (gneg) gneg[d] = a == 0 ? (REAL)0 : beta * p / u[d]; /* beta |u|^(beta-1) sign(u) */
    }
    return acc;
}

/* NormalizingFlowModel.log_probs (networks.py:71-76) with the N(0,I) base (networks.py:51-57):
 * MVN(0,I).log_prob(u) = -0.5*|u|^2 - (D/2) log(2 pi) */
void FN(nvp_log_probs)(const float *w, int D, int H, int B, int L, const REAL *x, int N, REAL *lp) {
    if (orc_flow_kind == 1) { FN(maf_log_probs)(w, D, H, B, L, x, N, lp); return; }
    int bs = 2 * FN(net_size)(D, H, L);
    for (int n = 0; n < N; ++n) {
        REAL r[512];
        for (int d = 0; d < D; ++d) r[d] = x[(size_t)n * D + d];
        REAL ld = 0;
        for (int b = 0; b < B; ++b) {
            ld += FN(coupling_fwd_row)(w + (size_t)b * bs, D, H, L, b, r, NULL, NULL, NULL);
            ld += FN(scale_fwd_row)(w, D, B, bs, b, r);
        }
        lp[n] = FN(base_logp)(r, D, NULL) + ld;
    }
}

/* ---------------------------------------------------------------------------------------------
 * Training: Trainer._train (trainer.py:384-403) for ONE minibatch, gradient of
 *   loss = -mean_i log_probs(x_i)        (trainer.py:394)
 * by hand-written reverse mode through the coupling stack, then torch.optim.Adam with coupled
 * weight decay (trainer.py:121-122; torch/optim/adam.py _single_tensor_adam).
 * ------------------------------------------------------------------------------------------- */

/* backward through one MLP: given g_out[D] -> accumulate parameter grads into gp, return g_m[D] (+=) */
static void FN(mlp_bwd)(const float *p, float *gp_unused, REAL *gp, int D, int H, int L, int act, const REAL *m,
                        const REAL *acts, const REAL *g_out, REAL *g_m) {
    (void)gp_unused;
    /* parameter offsets */
    int off_out = H * D + H + L * (H * H + H);
    const float *Wout = p + off_out;
    REAL *gWout = gp + off_out, *gbout = gp + off_out + D * H;
    const REAL *hlast = acts + (size_t)L * H;
    REAL gh[256], gpre[256];
    for (int j = 0; j < H; ++j) gh[j] = 0;
    for (int d = 0; d < D; ++d) {
        REAL g = g_out[d];
        gbout[d] += g;
        for (int j = 0; j < H; ++j) {
            gWout[(size_t)d * H + j] += g * hlast[j];
            gh[j] += g * (REAL)Wout[(size_t)d * H + j];
        }
    }
    for (int l = L; l >= 1; --l) { /* hidden layer l: h_l = act(W_l h_{l-1} + b_l) */
        const REAL *hl = acts + (size_t)l * H, *hp = acts + (size_t)(l - 1) * H;
        int off = H * D + H + (l - 1) * (H * H + H);
        const float *W = p + off;
        REAL *gW = gp + off, *gb = gp + off + H * H;
        for (int j = 0; j < H; ++j) gpre[j] = act == 0 ? gh[j] * ((REAL)1 - hl[j] * hl[j]) : (hl[j] > 0 ? gh[j] : (REAL)0);
        REAL ghp[256];
        for (int j = 0; j < H; ++j) ghp[j] = 0;
        for (int o = 0; o < H; ++o) {
            gb[o] += gpre[o];
            for (int i = 0; i < H; ++i) {
                gW[(size_t)o * H + i] += gpre[o] * hp[i];
                ghp[i] += gpre[o] * (REAL)W[(size_t)o * H + i];
            }
        }
        for (int j = 0; j < H; ++j) gh[j] = ghp[j];
    }
    { /* first layer */
        const REAL *h0 = acts;
        const float *W = p;
        REAL *gW = gp, *gb = gp + H * D;
        for (int j = 0; j < H; ++j) gpre[j] = act == 0 ? gh[j] * ((REAL)1 - h0[j] * h0[j]) : (h0[j] > 0 ? gh[j] : (REAL)0);
        for (int o = 0; o < H; ++o) {
            gb[o] += gpre[o];
            for (int i = 0; i < D; ++i) {
                gW[(size_t)o * D + i] += gpre[o] * m[i];
                g_m[i] += gpre[o] * (REAL)W[(size_t)o * D + i];
            }
        }
    }
}

/* loss and dloss/dw for a minibatch X[M,D].  grad has num_params entries (zeroed here). Returns loss. */
double FN(nvp_loss_grad)(const float *w, int D, int H, int B, int L, const REAL *X, int M, REAL *grad) {
    if (orc_flow_kind == 1) return FN(maf_loss_grad)(w, D, H, B, L, X, M, grad);
    int ns = FN(net_size)(D, H, L), bs = 2 * ns, np_ = B * bs + (orc_scale_mode == 2 ? B : 0);
    for (int i = 0; i < np_; ++i) grad[i] = 0;
    REAL *xin = (REAL *)malloc(sizeof(REAL) * (size_t)B * D);
    REAL *lss = (REAL *)malloc(sizeof(REAL) * (size_t)B * D);
    REAL *as = (REAL *)malloc(sizeof(REAL) * (size_t)B * (L + 1) * H);
    REAL *at = (REAL *)malloc(sizeof(REAL) * (size_t)B * (L + 1) * H);
    double loss = 0;
    for (int n = 0; n < M; ++n) {
        REAL r[512];
        for (int d = 0; d < D; ++d) r[d] = X[(size_t)n * D + d];
        REAL ld = 0;
        for (int b = 0; b < B; ++b) {
            memcpy(xin + (size_t)b * D, r, sizeof(REAL) * D);
            ld += FN(coupling_fwd_row)(w + (size_t)b * bs, D, H, L, b, r, lss + (size_t)b * D,
                                       as + (size_t)b * (L + 1) * H, at + (size_t)b * (L + 1) * H);
            ld += FN(scale_fwd_row)(w, D, B, bs, b, r);
        }
        REAL gy[512], gld = (REAL)(-1.0 / M);
        REAL lp = FN(base_logp)(r, D, gy) + ld;
        loss += -(double)lp / M;
        /* d(-lp/M)/du = -dlog p(u)/du / M (= u/M for the N(0,I) base) ; d/d(logdet) = -1/M */
        for (int d = 0; d < D; ++d) gy[d] = gy[d] / (REAL)M;
        for (int b = B - 1; b >= 0; --b) {
            const REAL *x = xin + (size_t)b * D, *ls = lss + (size_t)b * D;
            REAL m[512], gls[512], gt[512], gm[512];
            if (orc_scale_mode == 2) { /* ScaleLayer: y = c e^s, logdet += s;  r holds y on entry */
                REAL e = FN(expr)((REAL)w[(size_t)B * bs + b]), gs = gld;
                for (int d = 0; d < D; ++d) { gs += gy[d] * r[d]; gy[d] *= e; }
                grad[(size_t)B * bs + b] += gs;
            }
            for (int d = 0; d < D; ++d) {
                int cond = (d + b) & 1;
                m[d] = cond ? x[d] : (REAL)0;
                gm[d] = 0;
                if (cond) { gls[d] = 0; gt[d] = 0; }
                else {
                    REAL e = FN(expr)(ls[d]);
                    gls[d] = gy[d] * x[d] * e + gld; /* y = x e^{ls} + t ; logdet += ls */
                    gt[d] = gy[d];
                    gy[d] = gy[d] * e; /* direct path dy/dx */
                }
            }
            if (orc_scale_mode == 0)
                FN(mlp_bwd)(w + (size_t)b * bs, NULL, grad + (size_t)b * bs, D, H, L, 0, m, as + (size_t)b * (L + 1) * H, gls, gm);
            FN(mlp_bwd)(w + (size_t)b * bs + ns, NULL, grad + (size_t)b * bs + ns, D, H, L, 1, m,
                        at + (size_t)b * (L + 1) * H, gt, gm);
            for (int d = 0; d < D; ++d)
                if ((d + b) & 1) gy[d] += gm[d]; /* masked_inputs = inputs * mask */
            memcpy(r, x, sizeof(REAL) * D); /* r = this block's input = the previous ScaleLayer's output */
        }
    }
    free(xin); free(lss); free(as); free(at);
    return loss;
}

#include "maf_oracle_impl.h"
