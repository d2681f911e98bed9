/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C) of the reference's neural-spline flow, SingleSpeedSpline
 * (nnest/networks.py:393-715, adammoss/nnest v0.4.2; paths relative to /root/reference):
 *     per block:  ActNorm (:661-705)  ->  Invertible1x1Conv (:625-658)  ->  NSF_CL (:559-622)
 * NSF_CL = two rational-quadratic-spline couplings (Durkan et al. 2019) over the contiguous halves of the vector,
 * each conditioned by a 4-layer LeakyReLU(0.2) MLP (:393-409).
 *
 * Included twice by nnest_oracle.c:  REAL = float / FN(x) = spl32_##x  (the reference's precision)
 *                                    REAL = double / FN(x) = spl64_##x (rounding-noise yardstick)
 *
 * Parity status: PINNED for forward / inverse / log_probs / loss / ActNorm data-dependent initialisation against
 * fixtures produced by running the reference (oracle/gen_golden.py spline -> tests/golden/spline_*.npz, checked in
 * tests/test_oracle_golden.py).  Gradients are not restated here: the fixtures carry the reference's autograd
 * gradients, and oracle.py offers a float64 finite-difference check of this file's loss.
 *
 * Packed weight layout = torch state_dict order.  For block b (flows 3b, 3b+1, 3b+2 of the ModuleList, :711-714):
 *     ActNorm   s[D] t[D]
 *     Conv      L[D,D] S[D] U[D,D]             (P is a plain attribute, NOT in the state_dict: passed separately)
 *     NSF_CL    f1: W0[H,n1] b0[H] W1[H,H] b1[H] W2[H,H] b2[H] W3[o1,H] b3[o1]     n1 = #lower, o1 = (3K-1) #upper
 *               f2: W0[H,n2] ...                                       W3[o2,H] b3[o2]     n2 = #upper, o2 = (3K-1) #lower
 *     #lower = D/2 (+1 if D is odd), #upper = D/2            (:569-574, :578-581)
 * P: B matrices [D,D] (float), the fixed permutation of lu_unpack (:634-635).
 */

#ifndef REAL
#error "include from nnest_oracle.c"
#endif

#ifndef SPL_COMMON
#define SPL_COMMON
#define SPL_MIN_BIN 1e-3
#define SPL_MIN_DERIV 1e-3
static int spl_nlower(int D) { return D / 2 + (D & 1); }
static int spl_nupper(int D) { return D / 2; }
static int spl_mlp_size(int nin, int nout, int H) { return H * nin + H + 2 * (H * H + H) + nout * H + nout; }
static int spl_block_size(int D, int H, int K) {
    int nl = spl_nlower(D), nu = spl_nupper(D), P = 3 * K - 1;
    return 2 * D + (2 * D * D + D) + spl_mlp_size(nl, P * nu, H) + spl_mlp_size(nu, P * nl, H);
}
int orc_spline_num_params(int D, int H, int B, int K) { return B * spl_block_size(D, H, K); }
#endif

static REAL FN(exp)(REAL v) { return sizeof(REAL) == 4 ? (REAL)expf((float)v) : (REAL)exp((double)v); }
static REAL FN(log)(REAL v) { return sizeof(REAL) == 4 ? (REAL)logf((float)v) : (REAL)log((double)v); }
static REAL FN(sqrt)(REAL v) { return sizeof(REAL) == 4 ? (REAL)sqrtf((float)v) : (REAL)sqrt((double)v); }
static REAL FN(log1p)(REAL v) { return sizeof(REAL) == 4 ? (REAL)log1pf((float)v) : (REAL)log1p((double)v); }
static REAL FN(fabs)(REAL v) { return v < 0 ? -v : v; }

/* F.softplus (beta 1, threshold 20) */
static REAL FN(softplus)(REAL v) { return v > (REAL)20 ? v : FN(log1p)(FN(exp)(v)); }

/* torch.softmax over K entries */
static void FN(softmax)(const REAL *in, int K, REAL *out) {
    REAL mx = in[0], s = 0;
    for (int k = 1; k < K; ++k) if (in[k] > mx) mx = in[k];
    for (int k = 0; k < K; ++k) { out[k] = FN(exp)(in[k] - mx); s += out[k]; }
    for (int k = 0; k < K; ++k) out[k] = out[k] / s;
}

/* MLP (networks.py:393-409): Linear(nin,H) LReLU(.2) Linear(H,H) LReLU Linear(H,H) LReLU Linear(H,nout) */
static void FN(mlp4)(const float *p, int nin, int nout, int H, const REAL *x, REAL *out) {
    REAL h[256], h2[256];
    const float *W = p, *b = p + H * nin;
    for (int o = 0; o < H; ++o) {
        REAL acc = (REAL)b[o];
        for (int i = 0; i < nin; ++i) acc += (REAL)W[o * nin + i] * x[i];
        h[o] = acc > 0 ? acc : (REAL)0.2 * acc;
    }
    p += H * nin + H;
    for (int l = 0; l < 2; ++l) {
        W = p; b = p + H * H;
        for (int o = 0; o < H; ++o) {
            REAL acc = (REAL)b[o];
            for (int i = 0; i < H; ++i) acc += (REAL)W[o * H + i] * h[i];
            h2[o] = acc > 0 ? acc : (REAL)0.2 * acc;
        }
        memcpy(h, h2, sizeof(REAL) * H);
        p += H * H + H;
    }
    W = p; b = p + (size_t)nout * H;
    for (int o = 0; o < nout; ++o) {
        REAL acc = (REAL)b[o];
        for (int i = 0; i < H; ++i) acc += (REAL)W[(size_t)o * H + i] * h[i];
        out[o] = acc;
    }
}

/* NSF_CL's treatment of one conditioner output row (networks.py:583-587) followed by unconstrained_RQS + RQS
 * (:425-556) for ONE scalar input.  raw = the 3K-1 conditioner outputs of this dimension.  Returns the output and adds
 * the log|derivative| to *ld.  Quirks restated: softmax is applied twice to widths/heights (once in NSF_CL, scaled
 * by 2B, once in RQS) and softplus twice to the inner derivatives. */
static REAL FN(rqs_scalar)(const REAL *raw, int K, REAL tail, REAL x, int inverse, REAL *ld) {
    if (!(x >= -tail && x <= tail)) return x; /* outside the interval: identity, logabsdet 0 (:441-442) */
    REAL w1[64], h1[64], uw[64], uh[64], ud[65], cw[65], ch[65], wd[64], ht[64], dv[65];
    FN(softmax)(raw, K, w1);
    FN(softmax)(raw + K, K, h1);
    for (int k = 0; k < K; ++k) { uw[k] = 2 * tail * w1[k]; uh[k] = 2 * tail * h1[k]; } /* :585 */
    const REAL constant = FN(log)(FN(exp)((REAL)(1 - SPL_MIN_DERIV)) - 1);               /* :437 */
    ud[0] = constant; ud[K] = constant;
    for (int k = 0; k < K - 1; ++k) ud[k + 1] = FN(softplus)(raw[2 * K + k]);            /* :586 */
    /* RQS (:477-495) */
    FN(softmax)(uw, K, wd);
    cw[0] = 0;
    { REAL c = 0; for (int k = 0; k < K; ++k) { wd[k] = (REAL)SPL_MIN_BIN + (1 - (REAL)SPL_MIN_BIN * K) * wd[k]; c += wd[k]; cw[k + 1] = c; } }
    for (int k = 0; k <= K; ++k) cw[k] = (2 * tail) * cw[k] + (-tail);
    cw[0] = -tail; cw[K] = tail;
    for (int k = 0; k < K; ++k) wd[k] = cw[k + 1] - cw[k];
    for (int k = 0; k <= K; ++k) dv[k] = (REAL)SPL_MIN_DERIV + FN(softplus)(ud[k]);
    FN(softmax)(uh, K, ht);
    ch[0] = 0;
    { REAL c = 0; for (int k = 0; k < K; ++k) { ht[k] = (REAL)SPL_MIN_BIN + (1 - (REAL)SPL_MIN_BIN * K) * ht[k]; c += ht[k]; ch[k + 1] = c; } }
    for (int k = 0; k <= K; ++k) ch[k] = (2 * tail) * ch[k] + (-tail);
    ch[0] = -tail; ch[K] = tail;
    for (int k = 0; k < K; ++k) ht[k] = ch[k + 1] - ch[k];
    /* searchsorted (:417-422): last edge + eps, count of edges <= x, minus 1 */
    const REAL *edges = inverse ? ch : cw;
    int bin = -1;
    for (int k = 0; k <= K; ++k) {
        REAL e = edges[k];
        if (k == K) e += (REAL)1e-6;
        if (x >= e) ++bin;
    }
    if (bin < 0) bin = 0;
    if (bin > K - 1) bin = K - 1;
    const REAL icw = cw[bin], ibw = wd[bin], ich = ch[bin], ih = ht[bin];
    const REAL delta = ht[bin] / wd[bin], d0 = dv[bin], d1 = dv[bin + 1];
    if (inverse) { /* :515-539 */
        REAL a = (x - ich) * (d0 + d1 - 2 * delta) + ih * (delta - d0);
        REAL b = ih * d0 - (x - ich) * (d0 + d1 - 2 * delta);
        REAL c = -delta * (x - ich);
        REAL disc = b * b - 4 * a * c;
        REAL root = (2 * c) / (-b - FN(sqrt)(disc));
        REAL out = root * ibw + icw;
        REAL tomt = root * (1 - root);
        REAL den = delta + (d0 + d1 - 2 * delta) * tomt;
        REAL num = delta * delta * (d1 * root * root + 2 * delta * tomt + d0 * (1 - root) * (1 - root));
        *ld += -(FN(log)(num) - 2 * FN(log)(den));
        return out;
    } else { /* :541-556 */
        REAL theta = (x - icw) / ibw;
        REAL tomt = theta * (1 - theta);
        REAL numer = ih * (delta * theta * theta + d0 * tomt);
        REAL den = delta + (d0 + d1 - 2 * delta) * tomt;
        REAL out = ich + numer / den;
        REAL num = delta * delta * (d1 * theta * theta + 2 * delta * tomt + d0 * (1 - theta) * (1 - theta));
        *ld += FN(log)(num) - 2 * FN(log)(den);
        return out;
    }
}

/* NSF_CL.forward / .inverse for one row (networks.py:576-622) */
static REAL FN(nsf_row)(const float *p, int D, int H, int K, REAL tail, REAL *x, int inverse) {
    const int nl = spl_nlower(D), nu = spl_nupper(D), P = 3 * K - 1;
    const float *f1 = p, *f2 = p + spl_mlp_size(nl, P * nu, H);
    REAL out[23 * 256 + 64], ld = 0;
    REAL *lower = x, *upper = x + nl;
    if (!inverse) {
        FN(mlp4)(f1, nl, P * nu, H, lower, out);
        for (int j = 0; j < nu; ++j) upper[j] = FN(rqs_scalar)(out + j * P, K, tail, upper[j], 0, &ld);
        FN(mlp4)(f2, nu, P * nl, H, upper, out);
        for (int j = 0; j < nl; ++j) lower[j] = FN(rqs_scalar)(out + j * P, K, tail, lower[j], 0, &ld);
    } else {
        FN(mlp4)(f2, nu, P * nl, H, upper, out);
        for (int j = 0; j < nl; ++j) lower[j] = FN(rqs_scalar)(out + j * P, K, tail, lower[j], 1, &ld);
        FN(mlp4)(f1, nl, P * nu, H, lower, out);
        for (int j = 0; j < nu; ++j) upper[j] = FN(rqs_scalar)(out + j * P, K, tail, upper[j], 1, &ld);
    }
    return ld;
}

/* Invertible1x1Conv._assemble_W (networks.py:640-645): W = P (tril(L,-1) + I) (triu(U,1) + diag(S)) */
static void FN(assemble_W)(const float *Lp, const float *Sp, const float *Up, const float *Pm, int D, REAL *W) {
    REAL *Lm = (REAL *)malloc(sizeof(REAL) * D * D), *PL = (REAL *)malloc(sizeof(REAL) * D * D);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) Lm[i * D + j] = j < i ? (REAL)Lp[i * D + j] : (i == j ? (REAL)1 : (REAL)0);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            REAL acc = 0;
            for (int k = 0; k < D; ++k) acc += (REAL)Pm[i * D + k] * Lm[k * D + j];
            PL[i * D + j] = acc;
        }
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            REAL acc = 0;
            for (int k = 0; k < D; ++k) {
                REAL u = k < j ? (REAL)Up[k * D + j] : (k == j ? (REAL)Sp[k] : (REAL)0);
                acc += PL[i * D + k] * u;
            }
            W[i * D + j] = acc;
        }
    free(Lm); free(PL);
}

/* torch.inverse(W) (networks.py:655): Gauss-Jordan with partial pivoting */
static void FN(invert)(const REAL *W, int D, REAL *Wi) {
    REAL *A = (REAL *)malloc(sizeof(REAL) * D * 2 * D);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) { A[i * 2 * D + j] = W[i * D + j]; A[i * 2 * D + D + j] = i == j ? (REAL)1 : (REAL)0; }
    for (int c = 0; c < D; ++c) {
        int piv = c;
        for (int r = c + 1; r < D; ++r) if (FN(fabs)(A[r * 2 * D + c]) > FN(fabs)(A[piv * 2 * D + c])) piv = r;
        if (piv != c) for (int j = 0; j < 2 * D; ++j) { REAL t = A[c * 2 * D + j]; A[c * 2 * D + j] = A[piv * 2 * D + j]; A[piv * 2 * D + j] = t; }
        REAL inv = 1 / A[c * 2 * D + c];
        for (int j = 0; j < 2 * D; ++j) A[c * 2 * D + j] *= inv;
        for (int r = 0; r < D; ++r) {
            if (r == c) continue;
            REAL f = A[r * 2 * D + c];
            if (f == 0) continue;
            for (int j = 0; j < 2 * D; ++j) A[r * 2 * D + j] -= f * A[c * 2 * D + j];
        }
    }
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) Wi[i * D + j] = A[i * 2 * D + D + j];
    free(A);
}

/* NormalizingFlow.forward over [ActNorm, Conv, NSF_CL] x B (networks.py:24-32, :708-715) on a batch.
 * data_init != 0: ActNorm's data-dependent initialisation (:698-705) -- as on the first forward of a fresh
 * model, each block's s, t are set from the batch that reaches it (s = -log std (unbiased), t = -mean(x e^s)) and
 * WRITTEN BACK into w.  z, logdet may alias nothing; x is not modified. */
void FN(spline_forward)(float *w, const float *Pm, int D, int H, int B, int K, REAL tail, const REAL *x, int N, REAL *z,
                        REAL *logdet, int data_init) {
    const int bs = spl_block_size(D, H, K);
    REAL *W = (REAL *)malloc(sizeof(REAL) * D * D), *row = (REAL *)malloc(sizeof(REAL) * D);
    for (size_t i = 0; i < (size_t)N * D; ++i) z[i] = x[i];
    for (int n = 0; n < N; ++n) logdet[n] = 0;
    for (int b = 0; b < B; ++b) {
        float *pb = w + (size_t)b * bs;
        float *s = pb, *t = pb + D;
        if (data_init) {
            for (int d = 0; d < D; ++d) {
                REAL mean = 0;
                for (int n = 0; n < N; ++n) mean += z[(size_t)n * D + d];
                mean /= (REAL)N;
                REAL var = 0;
                for (int n = 0; n < N; ++n) { REAL c = z[(size_t)n * D + d] - mean; var += c * c; }
                var /= (REAL)(N - 1);
                REAL sv = -FN(log)(FN(sqrt)(var));
                s[d] = (float)sv;
                REAL es = FN(exp)((REAL)s[d]), m2 = 0;
                for (int n = 0; n < N; ++n) m2 += z[(size_t)n * D + d] * es;
                t[d] = (float)(-(m2 / (REAL)N));
            }
        }
        REAL lds = 0;
        for (int d = 0; d < D; ++d) lds += (REAL)s[d];
        const float *Lp = pb + 2 * D, *Sp = Lp + D * D, *Up = Sp + D;
        FN(assemble_W)(Lp, Sp, Up, Pm + (size_t)b * D * D, D, W);
        REAL ldc = 0;
        for (int d = 0; d < D; ++d) ldc += FN(log)(FN(fabs)((REAL)Sp[d]));
        const float *pn = Up + D * D;
        for (int n = 0; n < N; ++n) {
            REAL *r = z + (size_t)n * D;
            for (int d = 0; d < D; ++d) r[d] = r[d] * FN(exp)((REAL)s[d]) + (REAL)t[d]; /* AffineConstantFlow.forward :672-677 */
            for (int j = 0; j < D; ++j) {                                            /* z = x @ W :649 */
                REAL acc = 0;
                for (int i = 0; i < D; ++i) acc += r[i] * W[i * D + j];
                row[j] = acc;
            }
            memcpy(r, row, sizeof(REAL) * D);
            logdet[n] += lds;
            logdet[n] += ldc;
            logdet[n] += FN(nsf_row)(pn, D, H, K, tail, r, 0);
        }
    }
    free(W); free(row);
}

/* NormalizingFlow.inverse (networks.py:34-42): flows reversed */
void FN(spline_inverse)(const float *w, const float *Pm, int D, int H, int B, int K, REAL tail, const REAL *z, int N, REAL *x,
                        REAL *logdet) {
    const int bs = spl_block_size(D, H, K);
    REAL *W = (REAL *)malloc(sizeof(REAL) * D * D), *Wi = (REAL *)malloc(sizeof(REAL) * D * D), *row = (REAL *)malloc(sizeof(REAL) * D);
    for (size_t i = 0; i < (size_t)N * D; ++i) x[i] = z[i];
    for (int n = 0; n < N; ++n) logdet[n] = 0;
    for (int b = B - 1; b >= 0; --b) {
        const float *pb = w + (size_t)b * bs;
        const float *s = pb, *t = pb + D;
        const float *Lp = pb + 2 * D, *Sp = Lp + D * D, *Up = Sp + D, *pn = Up + D * D;
        FN(assemble_W)(Lp, Sp, Up, Pm + (size_t)b * D * D, D, W);
        FN(invert)(W, D, Wi);
        REAL lds = 0, ldc = 0;
        for (int d = 0; d < D; ++d) { lds += -(REAL)s[d]; ldc += FN(log)(FN(fabs)((REAL)Sp[d])); }
        for (int n = 0; n < N; ++n) {
            REAL *r = x + (size_t)n * D;
            logdet[n] += FN(nsf_row)(pn, D, H, K, tail, r, 1);
            for (int j = 0; j < D; ++j) {                                            /* x = z @ W^-1 :653-658 */
                REAL acc = 0;
                for (int i = 0; i < D; ++i) acc += r[i] * Wi[i * D + j];
                row[j] = acc;
            }
            memcpy(r, row, sizeof(REAL) * D);
            logdet[n] += -ldc;
            for (int d = 0; d < D; ++d) r[d] = (r[d] - (REAL)t[d]) * FN(exp)(-(REAL)s[d]); /* :679-684 */
            logdet[n] += lds;
        }
    }
    free(W); free(Wi); free(row);
}

/* log density of the base distribution (N(0,I), or GeneralisedNormal(0,1,beta) when orc_base_beta > 0) */
static REAL FN(sbase_logp)(const REAL *u, int D) {
    if (orc_base_beta == 0.0) {
        REAL ss = 0;
        for (int d = 0; d < D; ++d) ss += u[d] * u[d];
        return (REAL)(-0.5) * ss - (REAL)(D * 0.91893853320467274178);
    }
    const REAL cst = (REAL)(log(orc_base_beta) - log(2.0) - lgamma(1.0 / orc_base_beta));
    REAL acc = 0;
    for (int d = 0; d < D; ++d) {
        REAL a = u[d] < 0 ? -u[d] : u[d];
        acc += -(sizeof(REAL) == 4 ? (REAL)powf((float)a, (float)orc_base_beta) : (REAL)pow((double)a, orc_base_beta)) + cst;
    }
    return acc;
}

/* NormalizingFlowModel.log_probs (networks.py:71-76) with the N(0,I) base; returns -mean (the training loss,
 * trainer.py:394) and fills lp[N] */
double FN(spline_log_probs)(float *w, const float *Pm, int D, int H, int B, int K, REAL tail, const REAL *x, int N, REAL *lp,
                            int data_init) {
    REAL *z = (REAL *)malloc(sizeof(REAL) * (size_t)N * D);
    FN(spline_forward)(w, Pm, D, H, B, K, tail, x, N, z, lp, data_init);
    double tot = 0;
    for (int n = 0; n < N; ++n) {
        lp[n] = FN(sbase_logp)(z + (size_t)n * D, D) + lp[n];
        tot += (double)lp[n];
    }
    free(z);
    return -tot / N;
}
