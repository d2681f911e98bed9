/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Masked autoregressive flow (MAF).  [UNPINNED: absent from the reference -- nnest/trainer.py:83-100
 * accepts only 'choleksy' / 'nvp' / 'spline'; nothing in /root/reference to check against.  Build-defined; validated by
 * self-consistency: round trip, log-det antisymmetry, log-det against a brute-force Jacobian (the reference's own check of
 * its flows, nnest/trainer.py:373-382 / tests/test_flows.py:27-30), staged inverse == one-dimension-at-a-time inverse,
 * analytic gradient == finite differences.]
 *
 * Definition.  B blocks; block b holds two MADE nets (Germain et al. 2015) with the shapes of the reference's coupling
 * nets (networks.py:271-282): scale net Linear(D,H) Tanh [Linear(H,H) Tanh]xL Linear(H,D), translate net the same with
 * ReLU -- so the packed parameter vector has the RealNVP layout and size (nnest_oracle_impl.h).  Degrees: input / output
 * dimension d has degree d + 1 in even blocks and D - d in odd blocks (the order is reversed between blocks, as in
 * Papamakarios et al. 2017); hidden unit k has degree 1 + floor(k (D - 1) / H), i.e. the H units spread evenly over
 * 1 .. D-1 (with H < D - 1 the usual 1 + k mod (D - 1) would leave every dimension beyond the H-th unconditioned).
 * Masks: first layer W[k,d] lives iff deg(k) >= deg(d); hidden W[k',k] iff deg(k') >= deg(k); last layer W[d,k] iff
 * deg(d) > deg(k); masked entries are zero in effect and receive no gradient (they still take Adam's weight-decay step,
 * like the parameters RealNVP's mask never reaches).
 *
 * Orientation (MAF proper): the DENSITY direction is the single pass --
 *     forward  x -> z:  z_d = x_d exp(s_d(x)) + t_d(x),   logdet = +sum_d s_d(x)        (training, log_probs)
 *     inverse  z -> x:  x_d = (z_d - t_d(x)) exp(-s_d(x)), logdet = -sum_d s_d(x)        (sampling, MCMC proposals)
 * with s_d, t_d functions of the x of strictly lower degree.  The inverse is sequential in the degrees -- but with H hidden
 * units there are at most H distinct hidden degrees, so the dimensions fall into G <= H + 1 GROUPS (group of d = number of
 * distinct hidden degrees below deg(d)) whose members depend only on earlier groups: the inverse is G passes of the nets,
 * not D.  maf_inverse does it group by group; maf_inverse_seq one dimension at a time in degree order (the textbook
 * algorithm) -- tests hold the two equal.
 */
#ifndef REAL
#error "include from nnest_oracle.c"
#endif

#ifndef MAF_ORACLE_COMMON
#define MAF_ORACLE_COMMON
static int maf_deg_in(int D, int b, int d) { return (b & 1) ? D - d : d + 1; }
static int maf_deg_hid(int D, int H, int k) { return D < 2 ? 1 : 1 + (int)(((long)k * (D - 1)) / H); }
/* group of dimension d in block b: the number of distinct hidden degrees strictly below its own */
static int maf_group(int D, int H, int b, int d) {
    int din = maf_deg_in(D, b, d), n = 0, prev = 0;
    for (int k = 0; k < H; ++k) {
        int dk = maf_deg_hid(D, H, k);
        if (dk != prev) { if (dk < din) ++n; prev = dk; }
    }
    return n;
}
static int maf_num_groups(int D, int H) {
    int n = 0, prev = 0;
    for (int k = 0; k < H; ++k) { int dk = maf_deg_hid(D, H, k); if (dk != prev) { ++n; prev = dk; } }
    return n + 1;
}
/* is parameter `idx` of one net of block b (state_dict order: W0[H,D] b0[H] (W[H,H] b[H])xL Wout[D,H] bout[D]) unmasked? */
static int maf_param_live(int D, int H, int L, int b, int idx) {
    if (idx < H * D) return maf_deg_hid(D, H, idx / D) >= maf_deg_in(D, b, idx % D);
    idx -= H * D;
    if (idx < H) return 1;
    idx -= H;
    for (int l = 0; l < L; ++l) {
        if (idx < H * H) return maf_deg_hid(D, H, idx / H) >= maf_deg_hid(D, H, idx % H);
        idx -= H * H;
        if (idx < H) return 1;
        idx -= H;
    }
    if (idx < D * H) return maf_deg_in(D, b, idx / H) > maf_deg_hid(D, H, idx % H);
    return 1;
}
int orc_maf_num_groups(int D, int H) { return maf_num_groups(D, H); }
int orc_maf_group(int D, int H, int b, int d) { return maf_group(D, H, b, d); }
int orc_maf_param_live(int D, int H, int L, int b, int idx) { return maf_param_live(D, H, L, b, idx); }
/* the packed weights with the masks applied (caller frees) */
static float *maf_masked(const float *w, int D, int H, int B, int L) {
    int ns = H * D + H + L * (H * H + H) + D * H + D;
    float *wm = (float *)malloc(sizeof(float) * (size_t)B * 2 * ns);
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < 2; ++n)
            for (int i = 0; i < ns; ++i) {
                size_t p = ((size_t)b * 2 + n) * ns + i;
                wm[p] = maf_param_live(D, H, L, b, i) ? w[p] : 0.f;
            }
    return wm;
}
#endif

/* one block, density direction, one row: x <- x exp(s(x)) + t(x); returns sum s */
static REAL FN(maf_block_fwd_row)(const float *wmb, int D, int H, int L, REAL *x, REAL *ls_out, REAL *acts_s, REAL *acts_t) {
    REAL ls[512], t[512];
    int ns = FN(net_size)(D, H, L);
    FN(mlp)(wmb, D, H, L, 0, x, ls, acts_s);
    FN(mlp)(wmb + ns, D, H, L, 1, x, t, acts_t);
    REAL ld = 0;
    for (int d = 0; d < D; ++d) {
        x[d] = x[d] * FN(expr)(ls[d]) + t[d];
        ld += ls[d];
        if (ls_out) ls_out[d] = ls[d];
    }
    return ld;
}

/* one block, sampling direction, one row, group by group */
static REAL FN(maf_block_inv_row)(const float *wmb, int D, int H, int L, int b, REAL *v) {
    REAL ls[512], t[512];
    int ns = FN(net_size)(D, H, L), G = maf_num_groups(D, H);
    REAL ld = 0;
    for (int g = 0; g < G; ++g) {
        FN(mlp)(wmb, D, H, L, 0, v, ls, NULL);
        FN(mlp)(wmb + ns, D, H, L, 1, v, t, NULL);
        for (int d = 0; d < D; ++d)
            if (maf_group(D, H, b, d) == g) {
                v[d] = (v[d] - t[d]) * FN(expr)(-ls[d]);
                ld -= ls[d];
            }
    }
    return ld;
}

/* the same one dimension at a time in degree order (textbook MAF sampling): D passes of the nets */
static REAL FN(maf_block_inv_row_seq)(const float *wmb, int D, int H, int L, int b, REAL *v) {
    REAL ls[512], t[512];
    int ns = FN(net_size)(D, H, L);
    REAL ld = 0;
    for (int deg = 1; deg <= D; ++deg) {
        int d = (b & 1) ? D - deg : deg - 1;
        FN(mlp)(wmb, D, H, L, 0, v, ls, NULL);
        FN(mlp)(wmb + ns, D, H, L, 1, v, t, NULL);
        v[d] = (v[d] - t[d]) * FN(expr)(-ls[d]);
        ld -= ls[d];
    }
    return ld;
}

void FN(maf_forward)(const float *w, int D, int H, int B, int L, const REAL *x, int N, REAL *z, REAL *logdet) {
    int bs = 2 * FN(net_size)(D, H, L);
    float *wm = maf_masked(w, D, H, B, L);
    for (int n = 0; n < N; ++n) {
        REAL r[512];
        for (int d = 0; d < D; ++d) r[d] = x[(size_t)n * D + d];
        REAL ld = 0;
        for (int b = 0; b < B; ++b) ld += FN(maf_block_fwd_row)(wm + (size_t)b * bs, D, H, L, r, NULL, NULL, NULL);
        for (int d = 0; d < D; ++d) z[(size_t)n * D + d] = r[d];
        logdet[n] = ld;
    }
    free(wm);
}

static void FN(maf_inverse_any)(const float *w, int D, int H, int B, int L, const REAL *z, int N, REAL *x, REAL *logdet, int seq) {
    int bs = 2 * FN(net_size)(D, H, L);
    float *wm = maf_masked(w, D, H, B, L);
    for (int n = 0; n < N; ++n) {
        REAL r[512];
        for (int d = 0; d < D; ++d) r[d] = z[(size_t)n * D + d];
        REAL ld = 0;
        for (int b = B - 1; b >= 0; --b)
            ld += seq ? FN(maf_block_inv_row_seq)(wm + (size_t)b * bs, D, H, L, b, r) : FN(maf_block_inv_row)(wm + (size_t)b * bs, D, H, L, b, r);
        for (int d = 0; d < D; ++d) x[(size_t)n * D + d] = r[d];
        logdet[n] = ld;
    }
    free(wm);
}
void FN(maf_inverse)(const float *w, int D, int H, int B, int L, const REAL *z, int N, REAL *x, REAL *logdet) {
    FN(maf_inverse_any)(w, D, H, B, L, z, N, x, logdet, 0);
}
void FN(maf_inverse_seq)(const float *w, int D, int H, int B, int L, const REAL *z, int N, REAL *x, REAL *logdet) {
    FN(maf_inverse_any)(w, D, H, B, L, z, N, x, logdet, 1);
}

void FN(maf_log_probs)(const float *w, int D, int H, int B, int L, const REAL *x, int N, REAL *lp) {
    int bs = 2 * FN(net_size)(D, H, L);
    float *wm = maf_masked(w, D, H, B, L);
    for (int n = 0; n < N; ++n) {
        REAL r[512];
        for (int d = 0; d < D; ++d) r[d] = x[(size_t)n * D + d];
        REAL ld = 0;
        for (int b = 0; b < B; ++b) ld += FN(maf_block_fwd_row)(wm + (size_t)b * bs, D, H, L, r, NULL, NULL, NULL);
        lp[n] = FN(base_logp)(r, D, NULL) + ld;
    }
    free(wm);
}

/* loss = -mean_i log_probs(x_i) and its gradient wrt the packed weights (masked entries: exactly zero) */
double FN(maf_loss_grad)(const float *w, int D, int H, int B, int L, const REAL *X, int M, REAL *grad) {
    int ns = FN(net_size)(D, H, L), bs = 2 * ns, np_ = B * bs;
    for (int i = 0; i < np_; ++i) grad[i] = 0;
    float *wm = maf_masked(w, D, H, B, L);
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
            ld += FN(maf_block_fwd_row)(wm + (size_t)b * bs, D, H, L, r, lss + (size_t)b * D, as + (size_t)b * (L + 1) * H,
                                        at + (size_t)b * (L + 1) * H);
        }
        REAL gy[512], gld = (REAL)(-1.0 / M);
        REAL lp = FN(base_logp)(r, D, gy) + ld;
        loss += -(double)lp / M;
        for (int d = 0; d < D; ++d) gy[d] = gy[d] / (REAL)M;
        for (int b = B - 1; b >= 0; --b) {
            const REAL *x = xin + (size_t)b * D, *ls = lss + (size_t)b * D;
            REAL gls[512], gt[512], gm[512];
            for (int d = 0; d < D; ++d) {
                REAL e = FN(expr)(ls[d]);
                gls[d] = gy[d] * x[d] * e + gld; /* z = x e^{s} + t ; logdet += s */
                gt[d] = gy[d];
                gy[d] = gy[d] * e;               /* direct path dz/dx */
                gm[d] = 0;
            }
            FN(mlp_bwd)(wm + (size_t)b * bs, NULL, grad + (size_t)b * bs, D, H, L, 0, x, as + (size_t)b * (L + 1) * H, gls, gm);
            FN(mlp_bwd)(wm + (size_t)b * bs + ns, NULL, grad + (size_t)b * bs + ns, D, H, L, 1, x, at + (size_t)b * (L + 1) * H, gt, gm);
            for (int d = 0; d < D; ++d) gy[d] += gm[d]; /* through the nets' inputs (masked weights: strictly lower degrees) */
        }
    }
    for (int b = 0; b < B; ++b)
        for (int nn = 0; nn < 2; ++nn)
            for (int i = 0; i < ns; ++i)
                if (!maf_param_live(D, H, L, b, i)) grad[((size_t)b * 2 + nn) * ns + i] = 0;
    free(wm); free(xin); free(lss); free(as); free(at);
    return loss;
}
