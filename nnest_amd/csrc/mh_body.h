// mh_body.h -- the step loop of K4, shared by every 16-walker-tile form of the proposal kernel (nnest_kernels.hip: image / register /
// team forms; maf_kernels.h; spline_kernels.h) -- a header of its own so that a developer probe can instantiate one form without
// the rest of the translation unit (tools/spline_mh_probe.hip).  Included inside namespace nnest.
#pragma once

// ------------------------------------------------------------------------------------------------
// K4: persistent constrained Metropolis (Sampler._mcmc_sample hard-constraint branch, sampler.py:229-463)
// One wave = 16 walkers = one step-size adaptation group.  State (z, x, logdet, logl) stays in registers
// for all `steps`; the only global traffic is the start/end state (plus optional recorded noise / history).
// ------------------------------------------------------------------------------------------------
// The step loop, shared by the two kernel forms below; `inv(xs)` inverts the coupling stack on a tile and
// returns the lane's log-det partial.  DBG = true adds the test/diagnostic I/O (recorded noise replay, per-step
// history); the production instantiation carries none of those pointers through the step loop.
// GW = walkers per tile: 16, or 8 with the walkers held TWICE (lanes w and w ^ 8 carry walker w & 7: same state, same draws, same
// decisions) so that a flow whose per-lane work is per (walker, dimension) can give the two copies different dimensions
// (spline_kernels.h: spline_mh_kernel_pair).  Counts and stores take the low copy only.
template <int NT, bool DBG, class Inv, class Noise, int GW = 16>
__device__ __forceinline__ void mh_body(const MhArgs &a, int tile, int lane, const Inv &inv, Noise &noise, bool writer) {
    static_assert(GW == 16 || GW == 8, "walkers per tile");
    const int w = lane & 15, g = lane >> 4;
    const int row = tile * GW + (w & (GW - 1));
    const bool first_copy = GW == 16 || w < 8;
    const bool ok = row < a.C;
    const int D = a.s.D, S = a.steps;
    const int nvalid = min((int)GW, a.C - tile * GW);  // walkers in this adaptation group
    const LikeSpec like = a.like;
    const double loglstar = a.loglstar;
    const bool dynamic = (a.flags & (NNEST_MH_DYNAMIC_STEP | NNEST_MH_DYNAMIC_BATCH)) != 0;
    const bool batch = (a.flags & NNEST_MH_DYNAMIC_BATCH) != 0;
    const int lag = mh_flag_lag(a.flags);
    const int ntiles = (a.C + GW - 1) / GW;
    const bool free_mode = (a.flags & NNEST_MH_UNCONSTRAINED) != 0;

    f32x4 z[2][NT], x[2][NT];
    load_tile<NT>(a.z, row, ok, D, lane, z);
    // x = f^-1(z), log_det_J  (sampler.py:266, :295; the per-step re-inversion of the current z is
    // value-identical and therefore carried instead)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t) x[c][t] = z[c][t];
    float ld = group_sum(inv(x));
    double logl = ok ? a.logl[row] : 0.0;
    double scale = (double)a.step_size;  // python float in the reference (sampler.py:255, :428-431)
    int accept = 0, reject = 0, n_acc = 0, n_call = 0;

    // the chain's first x stays in the x output buffer for the launch: the reference counts a chain only if EVERY coordinate of its
    // last x differs from its first (nested.py:432), tested at the end (no register is held for it)
    // the chain's first x goes to a side buffer: the reference counts a chain only if EVERY coordinate of its last x differs from its first
    // (nested.py:432), tested by mh_all_moved_kernel behind this launch.  (NOT in this kernel: the compare at the end of the body --
    // sixteen loads, a ballot -- took the spline team kernel, whose 20 k-instruction step loop hipcc schedules precariously, from
    // 7.9 to 11.4 ms per launch although the loop's own instruction count moved by 2 %; profiles/r05/spline_all_moved_regression.txt)
    if (writer && a.x0) store_tile<NT>(a.x0, row, ok && first_copy, D, lane, x);
    if (DBG && writer) {
        if (a.hist_x) store_tile<NT>(a.hist_x, (long)row * (S + 1), ok, D, lane, x);
        if (a.hist_logl && ok && g == 0) a.hist_logl[(size_t)row * (S + 1)] = logl;
    }

    // the draws for step it+1 are requested before step it's coupling stack (they do not depend on it)
    float nz[NT][8];
    float u_next = 0.f;
    const bool recorded = DBG && a.noise_dz;
    if (!recorded) noise.next(nz, u_next);

    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, a_noise = 0, a_inv = 0, a_post = 0, a_tot = 0;
    (void)st0; (void)st1; (void)st2; (void)st3; (void)st4; (void)a_noise; (void)a_inv; (void)a_post; (void)a_tot;
    for (int it = 1; it <= S; ++it) {
        STAMP(st4);
        // batch-wide step rule: the counts of step it - lag are requested now and consumed at the end of the step
        unsigned long long early = 0;
        const bool have_total = batch && dynamic && it - lag >= 1;
        if (have_total && lag > 0 && !noise.relays()) early = mh_sync_read(a.sync, it - lag);
        // proposal z' = z + randn * scale  (sampler.py:310, :316); float32 like torch
        const float fs = (float)scale;
        f32x4 zp[2][NT], xp[2][NT];
        float u;
        if (recorded) {
            f32x4 dz[2][NT];
            load_tile<NT>(a.noise_dz + (size_t)(it - 1) * a.C * D, row, ok, D, lane, dz);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < NT; ++t) zp[c][t] = z[c][t] + dz[c][t] * fs;
            u = ok ? a.noise_u[(size_t)(it - 1) * a.C + row] : 1.f;
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                zp[0][t].x = z[0][t].x + nz[t][0] * fs; zp[1][t].x = z[1][t].x + nz[t][1] * fs;
                zp[0][t].y = z[0][t].y + nz[t][2] * fs; zp[1][t].y = z[1][t].y + nz[t][3] * fs;
                zp[0][t].z = z[0][t].z + nz[t][4] * fs; zp[1][t].z = z[1][t].z + nz[t][5] * fs;
                zp[0][t].w = z[0][t].w + nz[t][6] * fs; zp[1][t].w = z[1][t].w + nz[t][7] * fs;
            }
            u = u_next;
            STAMP(st0);
            noise.next(nz, u_next);
            STAMP(st1);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < NT; ++t) xp[c][t] = zp[c][t];
        STAMP(st2);
        float ldp = group_sum(inv(xp));  // sampler.py:321
        STAMP(st3);

        // log_ratio = log_det_J' - log_det_J, -inf outside the prior box  (sampler.py:326-331)
#ifdef NNEST_ABL_NOPRIOR
        const int inb = 1;
#else
        const int inb = inbox_tile<NT>(xp, lane);
#endif
        float log_ratio = inb ? (ldp - ld) : -INFINITY;
        float ratio = fminf(__expf(log_ratio), 1.0f);  // exp().clamp(max=1)  :335
        if (log_ratio != log_ratio) ratio = log_ratio;  // NaN stays NaN (u < NaN is false, as in torch)
        const bool pre = ok && (u < ratio);             // :336

        // likelihood of the proposal (the reference evaluates it only for `pre` rows, :358-360; here it is
        // evaluated for every row -- the lanes run in lock step anyway -- and only counted for `pre` rows)
#ifdef NNEST_ABL_NOLIKE
        double lp = (double)xp[0][0].x;
#else
        double lp = loglike_tile<NT>(like, D, lane, xp);
#endif
        bool acc = pre && (lp > loglstar);  // finite is guaranteed by the -1e100 clamp  :361
        if (free_mode) {  // sampler.py:396-410: float32 log-det difference + float64 likelihood difference, box prior
            const double lr = inb ? (double)(ldp - ld) + (lp - logl) : -INFINITY;
            const double rt = fmin(exp(lr), 1.0);
            acc = ok && ((double)u < rt);
        }
        n_call += (free_mode ? ok : pre) ? 1 : 0;
        n_acc += acc ? 1 : 0;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                z[c][t].x = acc ? zp[c][t].x : z[c][t].x; z[c][t].y = acc ? zp[c][t].y : z[c][t].y;
                z[c][t].z = acc ? zp[c][t].z : z[c][t].z; z[c][t].w = acc ? zp[c][t].w : z[c][t].w;
                x[c][t].x = acc ? xp[c][t].x : x[c][t].x; x[c][t].y = acc ? xp[c][t].y : x[c][t].y;
                x[c][t].z = acc ? xp[c][t].z : x[c][t].z; x[c][t].w = acc ? xp[c][t].w : x[c][t].w;
            }
        ld = acc ? ldp : ld;
        logl = acc ? lp : logl;
        if (dynamic) {  // sampler.py:422-431: per adaptation group (one wave), or over the whole batch with the
            //                 counts of step it - lag (NNEST_MH_DYNAMIC_BATCH; mh_common.h)
            const int tile_accepted = __popcll(__ballot(acc && g == 0 && first_copy));
            int num_accepted = tile_accepted, num_total = nvalid;
            bool apply = true;
            if (batch) {
                apply = have_total;
                if (noise.relays()) {   // team form, lag >= 2: the noise wave relays count and total through LDS
                    num_accepted = noise.relayed_total();
                    if (writer && lane == 0) noise.relay_count(it, tile_accepted);
                } else {
                    // lag >= 1: consume the counts requested at the top of the step BEFORE posting this step's -- memory
                    // operations retire in order, and an atomic stays outstanding for 600-3000 cycles (MI355X_MICROARCH.md)
                    if (apply && lag > 0) num_accepted = mh_sync_total(a.sync, it - lag, ntiles, early, a.sync_err);
                    if (writer && lane == 0) mh_sync_post(a.sync, it, tile, tile_accepted);
                    if (apply && lag == 0) num_accepted = mh_sync_total(a.sync, it, ntiles, mh_sync_read(a.sync, it), a.sync_err);
                }
                num_total = a.C;
            }
            if (apply) {
                if (2 * num_accepted > num_total) accept += 1; else reject += 1;
                if (accept > reject) scale *= exp(1.0 / (1 + accept));
                if (accept < reject) scale /= exp(1.0 / (1 + reject));
            }
        }
#ifdef NNEST_STAMP
        { unsigned long long e; STAMP(e); a_noise += st1 - st0; a_inv += st3 - st2; a_post += e - st3; a_tot += e - st4; }
#endif
        if (DBG && writer) {
            if (a.hist_x) store_tile<NT>(a.hist_x, (long)row * (S + 1) + it, ok, D, lane, x);
            if (a.hist_logl && ok && g == 0) a.hist_logl[(size_t)row * (S + 1) + it] = logl;
        }
    }
#ifdef NNEST_STAMP
    if (a.scale_out && lane == 0 && tile == 0) {  // diagnostic build only: cycles per segment, summed over steps
        float *o = a.scale_out + (writer ? 0 : 8);
        o[0] = (float)a_tot; o[1] = (float)a_noise; o[2] = (float)a_inv; o[3] = (float)a_post;
        o[4] = (float)inv.t_mlp; o[5] = (float)inv.t_xch; o[6] = (float)inv.t_upd;
    }
    if (!writer) return;
#else
    if (!writer) return;
#endif
    store_tile<NT>(a.z, row, ok && first_copy, D, lane, z);
    if (a.x) store_tile<NT>(a.x, row, ok && first_copy, D, lane, x);
    if (ok && g == 0 && first_copy) {
        a.logl[row] = logl;
        if (a.n_accept) a.n_accept[row] = n_acc | ((!a.x0 && n_acc > 0) ? NNEST_MH_ALL_MOVED : 0);   // (no side buffer: the accept count stands in)
        if (a.n_call) a.n_call[row] = n_call;
    }
#ifndef NNEST_STAMP
    // (one entry per 16 walkers whatever the tile: with 8-walker tiles -- fixed step or the batch rule only, every tile has the
    // same scale -- the even tiles report)
    if (a.scale_out && lane == 0 && (GW == 16 || (tile & 1) == 0)) a.scale_out[GW == 16 ? tile : tile >> 1] = (float)scale;
#endif
}

