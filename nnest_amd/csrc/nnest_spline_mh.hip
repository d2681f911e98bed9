// nnest_spline_mh.hip -- the spline proposal kernel's small-population forms -- the PAIR form (round 5) and the TEAM form -- a
// translation unit of its own because it is compiled with machine-level loop-invariant code motion OFF (Makefile: -mllvm
// -disable-machine-licm).
//
// 8 walkers per workgroup, held in BOTH halves of the 16 matrix-core columns (mh_body<..., 8>); the halves run the spline stage on
// different dimensions (spl_coupling_halves, spline_train_tile.h), so a wave evaluates one spline per coupling where the team form
// (spline_kernels.h) evaluates two, and 1000 walkers are 125 workgroups instead of 63.  Same walkers, same streams, the same
// arithmetic per walker as the team form (the per-16-walker step rule stays with the team form: its adaptation group is the tile).
// Two more things the team form does not do: the folded ActNorm + 1x1 conv of a block is dealt out over the four waves (one output
// tile each, exchanged through LDS: every wave of the team form repeats all 64 matrix instructions), and the conditioners' hidden
// layers -- the part of the image all four waves read -- sit in LDS for the launch (26 KB at x_dim 50).
// Per 1000 x 250 launch at x_dim 50: team form 7.93 ms; pair form 6.72; + dealt affine 5.98; + trunks in LDS 5.85.
//
// Why the flag: the step loop runs 250 times around ~20 000 instructions that derive dozens of masks, offsets and addresses from
// the lane index.  hipcc's MachineLICM hoists them all out of the loop, runs out of registers and parks them in accumulation
// registers / SGPR spill lanes, and the loop then moves them back one by one: with the hoisting this form runs 10.2-11.3 ms per
// 1000 x 250 launch (the team form 7.9), without it 6.6 ms (tools/spline_mh_probe.hip compiles just these two kernels in seconds;
// tools/spline_inv_probe.hip times the inverse alone: 23.1 us against the team form's 28.4).  The team form lost 2 % under the flag
// and lived in nnest_kernels.hip at first; when the knot construction went to packed operands (spl_knots2: -4 % on the pair form)
// the team form's schedule flipped the same way there -- 7.9 -> 11.3 ms -- and it moved here, where it runs 7.5.
#include <stdlib.h>
#include <string.h>

#include "flow_tile.h"
#include "mh_common.h"
#include "nnest_internal.h"
#include "spline_train_tile.h"

namespace nnest {

#include "mh_body.h"

template <int NT, int NH>
struct SplineInverseHalves {
    const float *img;
    SplineShape sp;
    float *buf;     // this wave's 16 x (D+1) layout-exchange buffer
    f32x4 *xch;     // 2 x [4][NT][64]: the exchanges alternate between the two (one barrier each, round 6)
    float *ldred;   // 2 x [4][16], likewise
    const float *trunks;   // the conditioners' hidden parts in LDS (spline_stage_trunks)
    int lane, wv;
    mutable int xsel = 0, lsel = 0;
#ifdef NNEST_STAMP
    unsigned long long t_mlp = 0, t_xch = 0, t_upd = 0;
#endif
    __device__ __forceinline__ float operator()(f32x4 (&xs)[2][NT]) const {
        f32x4 t[2][NT];
        spl_from_parity<NT>(buf, sp.D, sp.nl, lane, xs, t);
        float ld = group_sum(spline_inverse_tile_halves<NT, NH>(img, sp, lane, t, wv, xch, xsel, trunks));
        float *lr = ldred + ((lsel & 1) ? 64 : 0);
        lsel ^= 1;
        if (lane < 16) lr[wv * 16 + lane] = ld;
        spl_team_barrier();
        const int w = lane & 15;
        ld = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) ld += lr[k * 16 + w];
        spl_to_parity<NT>(buf, sp.D, sp.nl, lane, t, xs);
        return 0.25f * ld;  // the caller sums the four lanes of a walker
    }
};

template <int NT, int NH, bool DBG>
__global__ void __launch_bounds__(256) spline_mh_kernel_pair(MhArgs a, SplArgs q) {
    extern __shared__ __attribute__((aligned(16))) float lds_buf[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x;
    float *bufs = lds_buf;                                                        // 4 x 16 x (D+1)
    f32x4 *xch = reinterpret_cast<f32x4 *>(lds_buf + ((4 * 16 * (q.sp.D + 1) + 3) & ~3));  // 2 x 4 x NT x 64 f32x4
    float *ldred = reinterpret_cast<float *>(xch + 2 * 4 * NT * 64);              // 2 x 4 x 16
    float *trunks = ldred + 2 * 4 * 16;                                           // B x 2 x spl_cond_hidden_floats
    spline_stage_trunks<NT, NH>(q.img, q.sp, trunks, threadIdx.x, 256);
    __syncthreads();
    SplineInverseHalves<NT, NH> inv = {q.img, q.sp, bufs + (size_t)wv * 16 * (q.sp.D + 1), xch, ldred, trunks, lane, wv};
    XoshiroNoise<NT> noise;
    noise.init(a.seed, a.walker_offset + (uint64_t)(tile * 8 + (lane & 7)), lane >> 4, q.sp.D);
    mh_body<NT, DBG, SplineInverseHalves<NT, NH>, XoshiroNoise<NT>, 8>(a, tile, lane, inv, noise, wv == 0);
}

// Team form for small populations (fewer walker tiles than CUs): one workgroup of four waves per tile.  All four carry the
// same proposal state (same noise streams, same decisions); only the spline evaluations of the flow inverse are divided
// (spl_coupling TEAM = 4), and the log-det partials are summed through LDS.  Wave 0 writes the results.
template <int NT, int NH, int TEAM>
struct SplineInverseTeam {
    const float *img;
    SplineShape sp;
    float *buf;     // this wave's 16 x (D+1) layout-exchange buffer
    f32x4 *xch;     // [TEAM][NT][64]
    float *ldred;   // [TEAM][16]
    int lane, wv;
#ifdef NNEST_STAMP
    unsigned long long t_mlp = 0, t_xch = 0, t_upd = 0;
#endif
    __device__ __forceinline__ float operator()(f32x4 (&xs)[2][NT]) const {
        f32x4 t[2][NT];
        spl_from_parity<NT>(buf, sp.D, sp.nl, lane, xs, t);
        float ld = group_sum(spline_inverse_tile<NT, NH, TEAM>(img, sp, lane, t, wv, xch));
        if (lane < 16) ldred[wv * 16 + lane] = ld;
        spl_team_barrier();
        const int w = lane & 15;
        ld = 0.f;
#pragma unroll
        for (int k = 0; k < TEAM; ++k) ld += ldred[k * 16 + w];
        spl_team_barrier();
        spl_to_parity<NT>(buf, sp.D, sp.nl, lane, t, xs);
        return 0.25f * ld;  // the caller sums the four lanes of a walker
    }
};

template <int NT, int NH, int TEAM, bool DBG>
__global__ void __launch_bounds__(64 * TEAM) spline_mh_kernel_team(MhArgs a, SplArgs q) {
    extern __shared__ __attribute__((aligned(16))) float lds_buf[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x;
    float *bufs = lds_buf;                                                        // TEAM x 16 x (D+1)
    f32x4 *xch = reinterpret_cast<f32x4 *>(lds_buf + ((TEAM * 16 * (q.sp.D + 1) + 3) & ~3));  // TEAM x NT x 64 f32x4
    float *ldred = reinterpret_cast<float *>(xch + TEAM * NT * 64);               // TEAM x 16
    SplineInverseTeam<NT, NH, TEAM> inv = {q.img, q.sp, bufs + (size_t)wv * 16 * (q.sp.D + 1), xch, ldred, lane, wv};
    XoshiroNoise<NT> noise;
    noise.init(a.seed, a.walker_offset + (uint64_t)(tile * 16 + (lane & 15)), lane >> 4, q.sp.D);
    mh_body<NT, DBG>(a, tile, lane, inv, noise, wv == 0);
}

template <int NT, int NH>
static hipError_t launch_team_t(const MhArgs &a, const SplArgs &q, bool dbg, hipStream_t st) {
    const int ntiles = (a.C + 15) / 16;
    const size_t ldsb = (size_t)(((4 * 16 * (q.sp.D + 1) + 3) & ~3) + 4 * NT * 64 * 4 + 4 * 16) * sizeof(float);
    if (dbg) hipLaunchKernelGGL((spline_mh_kernel_team<NT, NH, 4, true>), dim3(ntiles), dim3(256), ldsb, st, a, q);
    else hipLaunchKernelGGL((spline_mh_kernel_team<NT, NH, 4, false>), dim3(ntiles), dim3(256), ldsb, st, a, q);
    return hipGetLastError();
}

hipError_t launch_spline_mh_team(const MhArgs &a, const SplArgs &q, bool dbg, hipStream_t st) {
    switch (q.sp.NTh * 10 + q.sp.NH) {
        case 11: return launch_team_t<1, 1>(a, q, dbg, st);
        case 21: return launch_team_t<2, 1>(a, q, dbg, st);
        case 31: return launch_team_t<3, 1>(a, q, dbg, st);
        case 41: return launch_team_t<4, 1>(a, q, dbg, st);
        case 12: return launch_team_t<1, 2>(a, q, dbg, st);
        case 22: return launch_team_t<2, 2>(a, q, dbg, st);
        default: return hipErrorInvalidConfiguration;
    }
}

template <int NT, int NH>
static hipError_t launch_pair_t(const MhArgs &a, const SplArgs &q, bool dbg, hipStream_t st) {
    const int ntiles8 = (a.C + 7) / 8;
    const size_t ldsb = (size_t)(((4 * 16 * (q.sp.D + 1) + 3) & ~3) + 2 * 4 * NT * 64 * 4 + 2 * 4 * 16 + q.sp.B * 2 * spl_cond_hidden_floats(NT, NH)) * sizeof(float);
    if (ldsb > 64 * 1024) {   // (above 64 KB of dynamic LDS a kernel has to be told so once)
        hipError_t e = dbg ? hipFuncSetAttribute(reinterpret_cast<const void *>(spline_mh_kernel_pair<NT, NH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)
                           : hipFuncSetAttribute(reinterpret_cast<const void *>(spline_mh_kernel_pair<NT, NH, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        if (e != hipSuccess) return e;
    }
    if (dbg) hipLaunchKernelGGL((spline_mh_kernel_pair<NT, NH, true>), dim3(ntiles8), dim3(256), ldsb, st, a, q);
    else hipLaunchKernelGGL((spline_mh_kernel_pair<NT, NH, false>), dim3(ntiles8), dim3(256), ldsb, st, a, q);
    return hipGetLastError();
}

// x_dim > 32 only (two or more 16-slot tiles per half: below that a wave of the team form has one super-tile per coupling already)
hipError_t launch_spline_mh_pair(const MhArgs &a, const SplArgs &q, bool dbg, hipStream_t st) {
    switch (q.sp.NTh * 10 + q.sp.NH) {
        case 21: return launch_pair_t<2, 1>(a, q, dbg, st);
        case 31: return launch_pair_t<3, 1>(a, q, dbg, st);
        case 41: return launch_pair_t<4, 1>(a, q, dbg, st);
        case 22: return launch_pair_t<2, 2>(a, q, dbg, st);
        default: return hipErrorInvalidConfiguration;
    }
}

}  // namespace nnest
