// mh_common.h -- what the forms of the persistent constrained-Metropolis kernel (K4) share: the launch arguments, the
// in-kernel proposal-noise generator and the batch-wide step-size counters.
#pragma once
#include "flow_tile.h"
#include "../../include/nnest_hip.h"

namespace nnest {

// NNEST_STAMP: diagnostic build only (tools/stamp_run.py): s_memtime stamps around the segments of the MH step
#ifdef NNEST_STAMP
#define STAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(v) do { } while (0)
#endif

// flags word of nnest_mh_constrained_steps: low bits NNEST_MH_*, bits 8..11 the lag of the batch-wide step rule,
// bits 16..19 the kernel form (0 = chosen by population)
enum { MH_FORM_AUTO = 0, MH_FORM_IMAGE = 1, MH_FORM_REG = 2, MH_FORM_TEAM = 3, MH_FORM_QUAD = 4, MH_FORM_QUAD1 = 5, MH_FORM_SOLO = 6 };
// n_accept_dev words: the accept count, and NNEST_MH_ALL_MOVED set when every coordinate of the chain's last x differs from its
// first x = f^-1(z_0) (what nnest/nested.py:432 asks of a chain before its end may replace a live point)
enum { NNEST_MH_ALL_MOVED = 1 << 30 };
// flags bit 28 (diagnostic, set by the launcher from NNEST_SOLO_VOTE=window): the solo form's exact steps wait for the publisher's window word
// instead of reading the tiles' counters themselves
enum { NNEST_MH_WINDOW_VOTES_ONLY = 1 << 28 };
// flags bits 29 / 30 (include/nnest_hip.h): sync_dev is one half of a double buffer, and the launch ZEROES the other half -- the
// one behind it (bit 29) or in front of it (bit 30), nnest_mh_sync_words(steps) words away -- for the next launch, on the spare
// waves of the workgroup that publishes the batch totals: the caller's fill launch in front of every K4 launch goes away (solo form)
// (NNEST_MH_SYNC_ZERO_NEXT / NNEST_MH_SYNC_ZERO_PREV: include/nnest_hip.h)
__host__ __device__ inline int mh_flag_lag(int flags) { return (flags >> 8) & 15; }
__host__ __device__ inline int mh_flag_form(int flags) { return (flags >> 16) & 15; }
__host__ __device__ inline int mh_flag_warm(int flags) { return (flags >> 20) & 255; }   // NNEST_MH_WARM

// Batch-wide step-size adaptation (sampler.py:422-431 over ALL walkers of the launch).  One 64-bit word per (step, shard):
// every workgroup adds (1 << 32 | accepted walkers of its tile) once per step; a step's total is complete when the
// arrival halves sum to the number of workgroups.  8 shards per step, 64 bytes apart (same-address atomics serialise at
// ~12 ns each, MI355X_MICROARCH.md "fanin").
// Reading a step's total is the expensive half -- every workgroup reading all 8 shards puts (workgroups x 8) reads per step on
// 8 addresses, which at 250 workgroups cost 1.5 us per 2.3 us step (measured, profiles/r02) -- so the quad form has ONE wave
// (the noise wave of workgroup 0) sum the shards and publish the total in 8 replicas of a result word (bit 63 = ready); a
// workgroup then reads one word per step.  Layout: counters [steps + 2][8][8], results [steps + 2][8][8], error word.
enum { MH_SYNC_SHARDS = 8, MH_SYNC_STRIDE = 8 /* uint64 per shard slot */, MH_SYNC_MAX_POLLS = 1 << 20 };
__host__ __device__ inline size_t mh_sync_counter_words(int steps) { return (size_t)(steps + 2) * MH_SYNC_SHARDS * MH_SYNC_STRIDE; }
// behind the counters and the per-step results: one "window" word per 32 steps (8 replicas): high half = P, the number of steps
// published so far (all steps <= P), low half = the up/down votes (2 * total > walkers) of the window's steps.  ONE fresh load
// of the current window tells a reader every decision taken so far, however early or late it asks (solo form).
__host__ __device__ inline size_t mh_sync_window_words(int steps) { return (size_t)((steps + 2) / 32 + 1) * MH_SYNC_SHARDS * MH_SYNC_STRIDE; }
__host__ __device__ inline size_t mh_sync_words(int steps) { return 2 * mh_sync_counter_words(steps) + mh_sync_window_words(steps); }

struct MhArgs {
    const float *img;
    FlowShape s;
    float *z;
    float *x;
    double *logl;
    double loglstar;
    float step_size;
    int steps;
    int C;
    int flags;
    LikeSpec like;
    const float *noise_dz;
    const float *noise_u;
    uint64_t seed;
    uint64_t walker_offset;
    float *hist_x;
    double *hist_logl;
    int *n_accept;
    int *n_call;
    float *scale_out;
    float *x0;                  // the 16-walker-tile forms: where the chains' FIRST x goes (mh_first_x_buffer), for the usable-chain test behind the launch; or NULL
    const float *packed;        // packed weights (state_dict order): the quad form gathers its fragments from these
    unsigned long long *sync;   // batch-wide step rule: [steps + 2][MH_SYNC_SHARDS][MH_SYNC_STRIDE], zeroed before the launch
    int *sync_err;              // set to 1 if a bounded poll of `sync` ran out (a workgroup was not resident)
};

// the usable-chain test of the 16-walker-tile forms behind their launch (nnest_kernels.hip)
float *mh_first_x_buffer(size_t floats, hipStream_t st);
hipError_t launch_mh_all_moved(const MhArgs &a, hipStream_t st);
hipError_t launch_mh_zero_other_sync(const MhArgs &a, hipStream_t st);

// in-wave proposal streams: normals per (walker, lane group); the accept uniform per walker (identical in its 4
// lanes).  Padded dims get exactly 0 (their weight fragments are 0, but 0 * inf would poison the accumulators).
template <int NT>
struct XoshiroNoise {
    Xoshiro128 rn, ru;
    unsigned valid_mask;
    __device__ __forceinline__ void init(uint64_t seed, uint64_t walker, int g, int D) {
        rn = xoshiro_seed(seed, walker, (uint32_t)g, NOISE_STREAM_DZ);
        ru = xoshiro_seed(seed, walker, 0xffffffffu, NOISE_STREAM_U);
        valid_mask = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (32 * t + 8 * g + j < D) valid_mask |= 1u << (8 * t + j);
    }
    // the batch-wide step rule relayed by a noise wave (LdsNoise of the team form): not here
    __device__ __forceinline__ bool relays() const { return false; }
    __device__ __forceinline__ int relayed_total() const { return 0; }
    __device__ __forceinline__ void relay_count(int, int) const {}
    __device__ __forceinline__ void next(float (&nz)[NT][8], float &u) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            xoshiro_normal8(rn, nz[t]);  // dims 32t + 8g + [0,8) = (c0r0, c1r0, c0r1, c1r1, c0r2, c1r2, c0r3, c1r3)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (!((valid_mask >> (8 * t + j)) & 1u)) nz[t][j] = 0.f;
        }
        u = xoshiro_uniform(ru);
    }
};

// ---- batch-wide step rule: the counters ----------------------------------------------------------------------------
// post this workgroup's accepted count of step `it` (one lane)
__device__ __forceinline__ void mh_sync_post(unsigned long long *sync, int it, int wg, int accepted) {
    unsigned long long *w = sync + ((size_t)it * MH_SYNC_SHARDS + (wg & (MH_SYNC_SHARDS - 1))) * MH_SYNC_STRIDE;
    __hip_atomic_fetch_add(w, (1ull << 32) | (unsigned long long)(unsigned)accepted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the 8 shard words of step `it`, summed (every lane reads all of them: same addresses, one request each)
__device__ __forceinline__ unsigned long long mh_sync_read(const unsigned long long *sync, int it) {
    const unsigned long long *w = sync + (size_t)it * MH_SYNC_SHARDS * MH_SYNC_STRIDE;
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < MH_SYNC_SHARDS; ++k)
        s += __hip_atomic_load(w + k * MH_SYNC_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return s;
}
// A wait that runs out (a workgroup of the launch is not resident: the GPU is shared) raises the launch's error word; every
// later wait of the launch sees it on its first miss and gives up at once, so a failed launch ends after ONE timeout instead
// of one per remaining step (the host raises from the error word, HipNVP.check_sync).
__device__ __forceinline__ bool mh_sync_failed(const int *err) {
    return err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
// (the value says which wait it was: 1 counter total, 2 publisher, 3 published result, 4 window vote -- HipNVP.check_sync reports it)
__device__ __forceinline__ void mh_sync_fail(int *err, int which = 1) {
    if (err) __hip_atomic_store(err, which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// accepted walkers of step `it` over the whole batch; waits (bounded) until all `nwg` workgroups have posted
__device__ __forceinline__ int mh_sync_total(const unsigned long long *sync, int it, int nwg, unsigned long long first, int *err) {
    unsigned long long s = first;
    int polls = 0;
    while ((int)(s >> 32) != nwg) {
        if ((polls & 255) == 0 && mh_sync_failed(err)) break;
        if (++polls > MH_SYNC_MAX_POLLS) {
            mh_sync_fail(err);
            break;
        }
        __builtin_amdgcn_s_sleep(2);
        s = mh_sync_read(sync, it);
    }
    return (int)(s & 0xffffffffull);
}

// --- aggregated form (quad kernel): one wave publishes each step's total, everyone else reads one word ---
// The publishing wave keeps EIGHT steps in flight (lane group g = lane >> 3 polls step t0 + g, one shard per lane): a poll is a
// round trip to the memory side (~1.5 us, about one step), so a wave that handled one step per round trip would throttle the
// whole grid to its own pace (measured: +0.08 ms per 250 steps at any lag).  Steps are published in order.
__device__ __forceinline__ void mh_sync_publisher(unsigned long long *sync, int steps, int last_step, int nwg, int nwalkers, int lane, int *err) {
    const int g = lane >> 3, sh = lane & 7;
    unsigned long long *results = sync + mh_sync_counter_words(steps);
    unsigned long long *windows = sync + 2 * mh_sync_counter_words(steps);
    unsigned int wbits = 0;   // votes of the window being filled
    int t0 = 1, polls = 0;
    while (t0 <= last_step) {
        const int t = t0 + g;
        const bool live = t <= last_step;
        unsigned long long v = 0;
        if (live) v = __hip_atomic_load(sync + ((size_t)t * MH_SYNC_SHARDS + sh) * MH_SYNC_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        const bool done = live && (int)(v >> 32) == nwg;
        const unsigned long long bal = __ballot(done);
        int ndone = 0;
        while (ndone < 8 && ((bal >> (8 * ndone)) & 1ull)) ++ndone;  // complete steps, consecutive from t0
        if (g < ndone)
            __hip_atomic_store(results + ((size_t)t * MH_SYNC_SHARDS + sh) * MH_SYNC_STRIDE, (1ull << 63) | (v & 0xffffffffull),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ndone > 0) {   // the window words: steps t0 .. t0 + ndone - 1 (step s is bit (s - 1) & 31 of window (s - 1) >> 5)
            const unsigned long long up = __ballot(done && 2 * (int)(v & 0xffffffffull) > nwalkers);   // lanes 8 j: step t0 + j
            for (int j = 0; j < ndone; ++j) {
                const int s_ = t0 + j, wi = (s_ - 1) >> 5, bi = (s_ - 1) & 31;
                wbits |= (unsigned int)((up >> (8 * j)) & 1ull) << bi;
                if (bi == 31 || j == ndone - 1) {   // window complete, or the last step published this round
                    if (lane < MH_SYNC_SHARDS)
                        __hip_atomic_store(windows + ((size_t)wi * MH_SYNC_SHARDS + lane) * MH_SYNC_STRIDE,
                                           ((unsigned long long)(unsigned int)s_ << 32) | wbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (bi == 31) wbits = 0;
                }
            }
        }
        t0 += ndone;
        if (ndone == 0) {
            if (++polls > MH_SYNC_MAX_POLLS || ((polls & 255) == 0 && mh_sync_failed(err))) {
                if (polls > MH_SYNC_MAX_POLLS) mh_sync_fail(err, 2);
                // unblock the readers: publish what there is
                for (int tt = t0; tt <= last_step; ++tt)
                    if (lane < MH_SYNC_SHARDS) {
                        __hip_atomic_store(results + ((size_t)tt * MH_SYNC_SHARDS + lane) * MH_SYNC_STRIDE, 1ull << 63, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(windows + ((size_t)((tt - 1) >> 5) * MH_SYNC_SHARDS + lane) * MH_SYNC_STRIDE,
                                           (unsigned long long)(unsigned int)last_step << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
}
__device__ __forceinline__ unsigned long long mh_result_load(const unsigned long long *sync, int steps, int it, int wg) {
    const unsigned long long *r = sync + mh_sync_counter_words(steps) + ((size_t)it * MH_SYNC_SHARDS + (wg & (MH_SYNC_SHARDS - 1))) * MH_SYNC_STRIDE;
    return __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int mh_result_wait(const unsigned long long *sync, int steps, int it, int wg, unsigned long long first, int *err) {
    unsigned long long v = first;
    int polls = 0;
    while (!(v >> 63)) {
        if ((polls & 255) == 0 && mh_sync_failed(err)) break;
        if (++polls > MH_SYNC_MAX_POLLS) {
            mh_sync_fail(err, 3);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        v = mh_result_load(sync, steps, it, wg);
    }
    return (int)(v & 0xffffffffull);
}

// the window word that holds step `it` (this workgroup's replica)
__device__ __forceinline__ unsigned long long mh_window_load(const unsigned long long *sync, int steps, int it, int wg) {
    const unsigned long long *w = sync + 2 * mh_sync_counter_words(steps) + ((size_t)((it - 1) >> 5) * MH_SYNC_SHARDS + (wg & (MH_SYNC_SHARDS - 1))) * MH_SYNC_STRIDE;
    return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// did step `it` vote up?  `first`: a window word loaded earlier; polls (bounded) until the word covers the step
__device__ __forceinline__ bool mh_window_vote(const unsigned long long *sync, int steps, int it, int wg, unsigned long long first, int *err) {
    unsigned long long v = first;
    int polls = 0;
    while ((int)(v >> 32) < it) {
        if ((polls & 255) == 0 && mh_sync_failed(err)) break;
        if (++polls > MH_SYNC_MAX_POLLS) {
            mh_sync_fail(err, 4 | (it << 8) | ((wg & 1023) << 20));   /* (+ the step and the workgroup of the wait that ran out) */
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        v = mh_window_load(sync, steps, it, wg);
    }
    return ((v >> ((it - 1) & 31)) & 1ull) != 0;
}

}  // namespace nnest
