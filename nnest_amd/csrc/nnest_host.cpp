// nnest_host.cpp -- host-side helpers of libnnest_hip.so that are not kernels.
//
// nnest_format_rows_e5: the text format of the reference's chain files (Sampler._save_samples, nnest/sampler.py:494-511:
// np.savetxt(..., fmt='%.5E'): one row per sample, "weight minusloglike params...", '%.5E' numbers separated by one space,
// '\n' after every row).  np.savetxt formats row by row in Python (9 us per 52-column row: 1.8 s for a config-2 chain of 2e5
// rows, 15 % of the run); here blocks of rows are formatted with snprintf("%.5E") -- the C library's correctly rounded
// conversion, the same digits Python's '%' produces -- on a few threads.
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/nnest_hip.h"

extern "C" long nnest_format_rows_e5(const double *rows, long n_rows, int n_cols, char *out, long out_cap, int threads) {
    if (!rows || !out || n_rows < 0 || n_cols < 1) return -1;
    const long per_row = (long)n_cols * 14;   // "-1.23457E+123 " is 14 characters
    if (out_cap < n_rows * per_row + 1) return -1;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (n_rows < 4096) threads = 1;
    else if ((long)threads * 2048 > n_rows) threads = (int)(n_rows / 2048);   // (a thread is worth starting for a few thousand rows)
    std::vector<long> used(threads, 0);
    const long chunk = (n_rows + threads - 1) / threads;
    auto work = [&](int t) {
        const long r0 = t * chunk, r1 = r0 + chunk < n_rows ? r0 + chunk : n_rows;
        char *p = out + r0 * per_row;   // every thread writes into its own slice of the worst-case layout
        for (long r = r0; r < r1; ++r) {
            const double *x = rows + r * n_cols;
            for (int c = 0; c < n_cols; ++c) {
                p += snprintf(p, 16, "%.5E", x[c]);
                *p++ = c + 1 < n_cols ? ' ' : '\n';
            }
        }
        used[t] = r0 < r1 ? p - (out + r0 * per_row) : 0;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    long total = used[0];   // compact the slices
    for (int t = 1; t < threads; ++t) {
        if (used[t] > 0) memmove(out + total, out + (long)t * chunk * per_row, (size_t)used[t]);
        total += used[t];
    }
    return total;
}

// nnest_format_scalar_rows: "<tag>,<step>,<repr(value)>\n".  repr(float) = the shortest decimal digits that read back to the
// same double (std::to_chars gives exactly those), laid out as CPython's float_repr does: with decpt = the position of the
// decimal point relative to the digit string, exponent form "d[.ddd]e+XX" if decpt <= -4 or decpt > 16, else positional with at
// least one digit either side of the point (".0" after an integral value); "nan", "inf", "-inf".
static char *repr_double(char *p, double v) {
    if (std::isnan(v)) { memcpy(p, "nan", 3); return p + 3; }
    if (std::isinf(v)) { if (v < 0) *p++ = '-'; memcpy(p, "inf", 3); return p + 3; }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof sci - 1, v, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX
    *r.ptr = 0;
    const char *q = sci;
    if (*q == '-') { *p++ = '-'; ++q; }
    char digits[24] = {0};
    int nd = 0;
    for (; q < r.ptr && *q != 'e'; ++q)
        if (*q != '.') digits[nd++] = *q;
    const int decpt = (int)strtol(q + 1, nullptr, 10) + 1;
    if (decpt <= -4 || decpt > 16) {
        *p++ = digits[0];
        if (nd > 1) { *p++ = '.'; memcpy(p, digits + 1, nd - 1); p += nd - 1; }
        return p + snprintf(p, 8, "e%c%02d", decpt - 1 < 0 ? '-' : '+', decpt - 1 < 0 ? 1 - decpt : decpt - 1);
    }
    if (decpt <= 0) {
        *p++ = '0'; *p++ = '.';
        for (int k = 0; k < -decpt; ++k) *p++ = '0';
        memcpy(p, digits, nd);
        return p + nd;
    }
    if (decpt >= nd) {
        memcpy(p, digits, nd); p += nd;
        for (int k = nd; k < decpt; ++k) *p++ = '0';
        *p++ = '.'; *p++ = '0';
        return p;
    }
    memcpy(p, digits, decpt); p += decpt;
    *p++ = '.';
    memcpy(p, digits + decpt, nd - decpt);
    return p + (nd - decpt);
}

extern "C" long nnest_format_scalar_rows(const char *tag, const long long *steps, const double *values, long n, char *out, long out_cap) {
    if (!tag || !steps || !values || !out || n < 0) return -1;
    const size_t tl = strlen(tag);
    if (out_cap < n * (long)(tl + 48) + 1) return -1;
    char *p = out;
    for (long i = 0; i < n; ++i) {
        memcpy(p, tag, tl);
        p += tl;
        *p++ = ',';
        p = std::to_chars(p, p + 21, steps[i]).ptr;
        *p++ = ',';
        p = repr_double(p, values[i]);
        *p++ = '\n';
    }
    return p - out;
}

// ---- the nested-sampling loop's per-iteration body (include/nnest_hip.h; nnest/nested.py:269-293, :429-437, :458-471) ----
// numpy's logaddexp (npy_logaddexp): x == y: x + log 2; else the larger + log1p(exp(-|x - y|)); libm exp / log1p
static inline double host_logaddexp(double x, double y) {
    if (x == y) return x + 0.693147180559945309417232121458176568;
    const double t = x - y;
    if (t > 0) return x + log1p(exp(-t));
    if (t <= 0) return y + log1p(exp(t));
    return t;   // NaN
}

extern "C" int nnest_host_mcmc_consume(nnest_host_state_t *st, int N, int D, int nd, double *active_u, double *active_v,
                                       double *active_logl, double *active_derived, const double *end_u, const double *end_v,
                                       const double *end_logl, const unsigned char *moved, const double *end_derived, int C,
                                       double *dead_v, double *dead_logl, double *dead_logwt, double *dead_logz_prev,
                                       long long dead_cap, double dlogz, long long max_iters, long long update_interval,
                                       long long log_interval) {
    const int W = D + nd;
    int resume = st->resume;
    st->resume = NNEST_HOST_TOP;
    // np.argmin(active_logl) (nested.py:272: the FIRST smallest) per iteration is a scan of N values -- 2.7 s of a config-5 run
    // (8000 live points, 10^6 iterations).  One value changes per iteration, so the minimum is kept in a tournament tree over the
    // indices: node = the better of its children, the left (lower indices) on a tie; rebuilt at every entry (the caller owns the
    // array between calls), one leaf-to-root path per replaced point.
    int P2 = 1;
    while (P2 < N) P2 <<= 1;
    static thread_local std::vector<int> tree;
    tree.assign(2 * (size_t)P2, -1);
    auto better = [&](int a, int b) { return b < 0 ? a : a < 0 ? b : (active_logl[b] < active_logl[a] ? b : a); };
    if (N > 64) {
        for (int i = 0; i < N; ++i) tree[P2 + i] = i;
        for (int k = P2 - 1; k >= 1; --k) tree[k] = better(tree[2 * k], tree[2 * k + 1]);
    }
    for (;;) {
        if (resume == NNEST_HOST_TOP) {
            if (!(st->fraction_remain > dlogz && st->it <= max_iters)) return NNEST_HOST_FINISHED;   // nested.py:269
            if (st->accept_point && st->n_dead >= dead_cap) return NNEST_HOST_DEAD_FULL;
            int worst = 0;
            if (N > 64) {
                worst = tree[1];
            } else {
                for (int i = 1; i < N; ++i)
                    if (active_logl[i] < active_logl[worst]) worst = i;
            }
            st->worst = worst;
            st->loglstar = active_logl[worst];
            if (st->accept_point) {                                  // nested.py:280-293
                const double logwt = st->logvol + active_logl[worst];
                const long long k = st->n_dead;
                dead_logz_prev[k] = st->logz;
                st->logz = host_logaddexp(st->logz, logwt);
                memcpy(dead_v + k * W, active_v + (size_t)worst * D, sizeof(double) * D);
                if (nd > 0) memcpy(dead_v + k * W + D, active_derived + (size_t)worst * nd, sizeof(double) * nd);
                dead_logwt[k] = logwt;
                dead_logl[k] = st->loglstar;
                st->n_dead = k + 1;
                st->accept_point = 0;
            }
            if (st->first_time || st->it % update_interval == 0) return NNEST_HOST_RETRAIN;   // nested.py:311-314
        }
        if (resume == NNEST_HOST_TOP || resume == NNEST_HOST_AFTER_TRAIN) {
            if (st->nb >= C) return NNEST_HOST_NEED_SAMPLES;         // nested.py:399
        }
        if (resume != NNEST_HOST_AFTER_LOG) {
            const double loglstar = st->loglstar;
            const int worst = st->worst;
            while (st->nb < C) {                                     // nested.py:429-437
                const int cand = st->nb++;
                if (moved[cand] && end_logl[cand] > loglstar) {
                    memcpy(active_u + (size_t)worst * D, end_u + (size_t)cand * D, sizeof(double) * D);
                    memcpy(active_v + (size_t)worst * D, end_v + (size_t)cand * D, sizeof(double) * D);
                    active_logl[worst] = end_logl[cand];
                    if (N > 64)
                        for (int k = (P2 + worst) >> 1; k >= 1; k >>= 1) tree[k] = better(tree[2 * k], tree[2 * k + 1]);
                    if (end_logl[cand] > st->max_logl) st->max_logl = end_logl[cand];
                    if (nd > 0) memcpy(active_derived + (size_t)worst * nd, end_derived + (size_t)cand * nd, sizeof(double) * nd);
                    st->accept_point = 1;
                    break;
                }
            }
            if (st->accept_point && st->it > 0 && st->it % log_interval == 0) return NNEST_HOST_LOG;   // nested.py:439-456
        }
        resume = NNEST_HOST_TOP;
        if (st->accept_point) {                                      // nested.py:458-471
            st->logvol -= 1.0 / (double)N;
            const double logz_remain = st->max_logl - (double)st->it / (double)N;
            st->fraction_remain = host_logaddexp(st->logz, logz_remain) - st->logz;
            st->it += 1;
            if (st->it > 0 && st->it % log_interval == 0) return NNEST_HOST_CHECKPOINT;   // nested.py:473-485
        }
    }
}

// ---- the same loop while 'rejection_prior' is the strategy in force (nnest/nested.py:322-334, :362-373; Sampler._rejection_prior_sample,
// nnest/sampler.py:529-543) -------------------------------------------------------------------------------------------------
// The reference draws one prior sample per likelihood call until one lies above loglstar.  The draws are independent, so the host
// driver evaluates them a block per launch of the likelihood kernel (nnest_amd/sampler.py::_rejection_prior_sample) and walks the
// block: candidates are examined in order, each once; the first above the threshold is this iteration's new live point and the
// iteration's call count is the number of candidates examined up to and including it.  What this function is handed is the
// block's candidate list -- the (sorted) indices that were above the threshold when the block was made, with their float32-input
// and float64 likelihoods, their rows and their transformed rows: the threshold only rises, so every later hit is among them.
extern "C" int nnest_host_prior_consume(nnest_host_state_t *st, nnest_host_prior_t *pr, int N, int D, int nd, double *active_u,
                                        double *active_v, double *active_logl, double *active_derived, const long long *cand_idx,
                                        const double *cand_logl32, const double *cand_logl64, const double *cand_u, const double *cand_v,
                                        const double *cand_derived, double *dead_v, double *dead_logl, double *dead_logwt,
                                        double *dead_logz_prev, long long dead_cap, double dlogz, long long max_iters,
                                        long long log_interval, double volume_switch, double mcmc_steps, int mcmc_valid) {
    const int W = D + nd;
    int resume = st->resume;
    st->resume = NNEST_HOST_TOP;
    for (;;) {
        if (resume == NNEST_HOST_TOP) {
            if (!(st->fraction_remain > dlogz && st->it <= max_iters)) return NNEST_HOST_FINISHED;   // nested.py:269
            if (pr->expired) return NNEST_HOST_EXPIRED;              // the next pass takes the next strategy (nested.py:300-306)
            if (st->accept_point && st->n_dead >= dead_cap) return NNEST_HOST_DEAD_FULL;
            int worst = 0;                                           // np.argmin: the first smallest (nested.py:272)
            for (int i = 1; i < N; ++i)
                if (active_logl[i] < active_logl[worst]) worst = i;
            st->worst = worst;
            st->loglstar = active_logl[worst];
            if (st->accept_point) {                                  // nested.py:280-293
                const double logwt = st->logvol + active_logl[worst];
                const long long k = st->n_dead;
                dead_logz_prev[k] = st->logz;
                st->logz = host_logaddexp(st->logz, logwt);
                memcpy(dead_v + k * W, active_v + (size_t)worst * D, sizeof(double) * D);
                if (nd > 0) memcpy(dead_v + k * W + D, active_derived + (size_t)worst * nd, sizeof(double) * nd);
                dead_logwt[k] = logwt;
                dead_logl[k] = st->loglstar;
                st->n_dead = k + 1;
                st->accept_point = 0;
            }
        }
        if (resume == NNEST_HOST_TOP || resume == NNEST_HOST_AFTER_SAMPLES) {
            // Sampler._rejection_prior_sample: the next candidate of the block above the threshold
            const double loglstar = st->loglstar;
            if (pr->pos >= pr->n) return NNEST_HOST_NEED_SAMPLES;    // no block, or used up: a new one, resume = NNEST_HOST_AFTER_SAMPLES
            long long k = pr->k, j = -1;
            for (; k < pr->n_cand; ++k) {
                const long long c = cand_idx[k];
                if (c >= pr->pos && cand_logl32[k] > loglstar && cand_logl64[k] > loglstar) { j = c; break; }
            }
            if (j < 0) {                                             // the rest of the block holds nothing above the threshold
                pr->total_calls += pr->n - pr->pos;
                pr->pending_calls += pr->n - pr->pos;
                if (pr->hits == 0) { const long long b = 4 * pr->n; pr->block_next = b < 65536 ? b : 65536; }
                pr->pos = pr->n; pr->k = pr->n_cand;
                return NNEST_HOST_NEED_SAMPLES;
            }
            const long long examined = j + 1 - pr->pos;
            pr->total_calls += examined;
            const double nc = (double)(pr->pending_calls + examined);   // candidates examined up to and including the accepted one
            pr->pending_calls = 0;
            pr->pos = j + 1; pr->k = k + 1; pr->hits += 1;
            {   // the next block: ~ 16 acceptances' worth of candidates at the rate seen, bounded
                double b = 16.0 * (double)(j + 1) / (double)pr->hits;
                if (b < 256.0) b = 256.0;
                if (b > 65536.0) b = 65536.0;
                pr->block_next = (long long)b;
            }
            // ncs.append(nc); mean_calls = np.mean(ncs[-20:]) if len(ncs) > 20 else 0   (nested.py:325-326; the counts are integers:
            // their sum is exact in any order)
            pr->ncs[pr->ncs_len % 20] = nc;
            pr->ncs_len += 1;
            double mean_calls = 0.0;
            if (pr->ncs_len > 20) {
                double sum = 0.0;
                for (int q = 0; q < 20; ++q) sum += pr->ncs[q];
                mean_calls = sum / 20.0;
            }
            pr->mean_calls = mean_calls;
            // nested.py:328-334: np.exp(-it / N) < volume_switch >= 0 or (volume_switch < 0 and mean_calls > mcmc_steps and mcmc_valid)
            const bool expire = (volume_switch >= 0.0 && exp(-(double)st->it / (double)N) < volume_switch) ||
                                (volume_switch < 0.0 && mean_calls > mcmc_steps && mcmc_valid);
            if (expire) { pr->expired = 1; pr->ncs_len = 0; }
            // nested.py:362-373: the candidate replaces the worst live point
            const int worst = st->worst;
            const long long kk = k;
            memcpy(active_u + (size_t)worst * D, cand_u + (size_t)kk * D, sizeof(double) * D);
            memcpy(active_v + (size_t)worst * D, cand_v + (size_t)kk * D, sizeof(double) * D);
            active_logl[worst] = cand_logl64[kk];
            if (cand_logl64[kk] > st->max_logl) st->max_logl = cand_logl64[kk];
            if (nd > 0) memcpy(active_derived + (size_t)worst * nd, cand_derived + (size_t)kk * nd, sizeof(double) * nd);
            st->accept_point = 1;
            if (st->it > 0 && (st->it + 1) % log_interval == 0) return NNEST_HOST_LOG;   // nested.py:374-378 (before `it` advances)
        }
        resume = NNEST_HOST_TOP;
        if (st->accept_point) {                                      // nested.py:458-471
            st->logvol -= 1.0 / (double)N;
            const double logz_remain = st->max_logl - (double)st->it / (double)N;
            st->fraction_remain = host_logaddexp(st->logz, logz_remain) - st->logz;
            st->it += 1;
            if (st->it > 0 && st->it % log_interval == 0) return NNEST_HOST_CHECKPOINT;   // nested.py:473-485
        }
    }
}

extern "C" double nnest_host_h_update(double h, const double *e1, const double *e2, const double *logl, const double *logz_prev,
                                      const double *total, long long n) {
    for (long long k = 0; k < n; ++k) {
        const double a = e1[k] * logl[k];
        const double b = e2[k] * (h + logz_prev[k]);
        h = (a + b) - total[k];
    }
    return h;
}
