// flow_tile.h -- device-side building blocks of the RealNVP coupling stack on gfx950 (CDNA4).
//
// Data layout (DESIGN.md "Data layout").  One wave64 owns a tile of 16 walkers.  The D coordinates
// of a walker are split by parity class (class 0 = even dims, 1 = odd dims: the two sides of the
// reference's alternating mask, nnest/networks.py:333-334, :346) into "slots": dim d = 2*slot + class.
// A class is padded to 16*NT slots.  A [16 slots x 16 walkers] sub-tile lives in the accumulator
// (C/D) layout of v_mfma_f32_16x16x4_f32:
//       lane = 16*g + w  (g = 0..3, w = walker 0..15),  register r = 0..3   <->   slot 16*tau + 4*g + r
// so lane (g,w) holds dims [32*tau + 8*g, 32*tau + 8*g + 8) of walker w: 8 consecutive floats of its row.
//
// Every layer is computed transposed, Out^T[feature][walker] = W[feature][k] * In^T[k][walker], with the
// weights as the MFMA A operand and the activations as the B operand.  The C/D layout of one layer's
// output is then directly the B operand of the next layer's k-steps (k-step r takes register r; lane
// group g supplies k = 4g + r), with the weight fragments permuted to match at repack time.  No LDS
// traffic, no cross-lane movement between layers.
//
// Weight fragment image (built by repack_fragments_kernel, one image per flow), per (block, net):
//   L1   [ht][tau][r][64 lanes]          A-fragments of Linear(D,H) restricted to the conditioning class
//   L2   [l][hto][hti][r][64 lanes]      A-fragments of the L hidden Linear(H,H)
//   L3   [tau][ht][r][64 lanes]          A-fragments of Linear(H,D) restricted to the transformed class
//   b1 [16*NH]  b2 [l][16*NH]  b3 [16*NT]   biases (b3 in slot order, zero on padded slots)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nnest {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FlowShape {
    int D, H, B, L;
    int NT;  // 16-slot tiles per parity class: ceil(ceil(D/2)/16)
    int NH;  // H / 16
    int net_floats;    // floats per (block, net) in the fragment image
    int image_floats;  // B * 2 * net_floats
    int net_params;    // packed parameter count per net (state_dict layout)
};

__host__ __device__ inline int frag_off_L1() { return 0; }
__host__ __device__ inline int frag_off_L2(int NT, int NH) { return NH * NT * 4 * 64; }
__host__ __device__ inline int frag_off_L3(int NT, int NH, int L) { return frag_off_L2(NT, NH) + L * NH * NH * 4 * 64; }
__host__ __device__ inline int frag_off_b1(int NT, int NH, int L) { return frag_off_L3(NT, NH, L) + NT * NH * 4 * 64; }
__host__ __device__ inline int frag_off_b2(int NT, int NH, int L) { return frag_off_b1(NT, NH, L) + 16 * NH; }
__host__ __device__ inline int frag_off_b3(int NT, int NH, int L) { return frag_off_b2(NT, NH, L) + L * 16 * NH; }
__host__ __device__ inline int frag_net_floats(int NT, int NH, int L) { return frag_off_b3(NT, NH, L) + 16 * NT; }

// ---- elementwise helpers ------------------------------------------------------------------------
// tanh(x) = 1 - 2/(e^{2x}+1): v_exp_f32 + v_rcp_f32; abs error ~1e-7 (inputs to a linear layer, so
// absolute accuracy is what matters).  Saturates correctly for |x| large; NaN propagates.
__device__ __forceinline__ float fast_tanh(float x) {
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

template <int ACT>  // 0 = tanh (scale_net, networks.py:271-276), 1 = relu (translate_net, networks.py:278-282)
__device__ __forceinline__ f32x4 activate(f32x4 v) {
    f32x4 o;
    if (ACT == 0) {
        o.x = fast_tanh(v.x); o.y = fast_tanh(v.y); o.z = fast_tanh(v.z); o.w = fast_tanh(v.w);
    } else {
        o.x = fmaxf(v.x, 0.f); o.y = fmaxf(v.y, 0.f); o.z = fmaxf(v.z, 0.f); o.w = fmaxf(v.w, 0.f);
    }
    return o;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One MLP of a coupling layer (networks.py:271-282) on a 16-walker tile.
//   wn   : base of this (block, net) in the fragment image (LDS or global)
//   in   : conditioning-class tiles; out: transformed-class tiles (ls or t), both in C/D layout
template <int NT, int NH, int ACT>
__device__ __forceinline__ void mlp_tile(const float *__restrict__ wn, int L, int lane, const f32x4 (&in)[NT],
                                         f32x4 (&out)[NT]) {
    const int g4 = (lane >> 4) * 4;
    const float *fL1 = wn + frag_off_L1() + lane;
    const float *fL2 = wn + frag_off_L2(NT, NH) + lane;
    const float *fL3 = wn + frag_off_L3(NT, NH, L) + lane;
    const float *b1 = wn + frag_off_b1(NT, NH, L) + g4;
    const float *b2 = wn + frag_off_b2(NT, NH, L) + g4;
    const float *b3 = wn + frag_off_b3(NT, NH, L) + g4;
    f32x4 h[NH];
#pragma unroll
    for (int ht = 0; ht < NH; ++ht) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(b1 + 16 * ht);
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            const float *f = fL1 + ((ht * NT + tau) * 4) * 64;
            acc = mfma4(f[0], in[tau].x, acc);
            acc = mfma4(f[64], in[tau].y, acc);
            acc = mfma4(f[128], in[tau].z, acc);
            acc = mfma4(f[192], in[tau].w, acc);
        }
        h[ht] = activate<ACT>(acc);
    }
    for (int l = 0; l < L; ++l) {
        f32x4 h2[NH];
#pragma unroll
        for (int hto = 0; hto < NH; ++hto) {
            f32x4 acc = *reinterpret_cast<const f32x4 *>(b2 + (l * NH + hto) * 16);
#pragma unroll
            for (int hti = 0; hti < NH; ++hti) {
                const float *f = fL2 + (((l * NH + hto) * NH + hti) * 4) * 64;
                acc = mfma4(f[0], h[hti].x, acc);
                acc = mfma4(f[64], h[hti].y, acc);
                acc = mfma4(f[128], h[hti].z, acc);
                acc = mfma4(f[192], h[hti].w, acc);
            }
            h2[hto] = activate<ACT>(acc);
        }
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) h[ht] = h2[ht];
    }
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(b3 + 16 * tau);
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
            const float *f = fL3 + ((tau * NH + ht) * 4) * 64;
            acc = mfma4(f[0], h[ht].x, acc);
            acc = mfma4(f[64], h[ht].y, acc);
            acc = mfma4(f[128], h[ht].z, acc);
            acc = mfma4(f[192], h[ht].w, acc);
        }
        out[tau] = acc;
    }
}

// CouplingLayer.forward / inverse on a tile (networks.py:289-309).  `cond` passes through untouched
// (bit-exact, which is what makes the inverse exact); `trans` is updated in place.  Returns this
// lane's partial of the block's log-det (sum over its 4*NT slots; padded slots contribute exactly 0).
template <int NT, int NH, bool INVERSE>
__device__ __forceinline__ float coupling_tile(const float *__restrict__ wblk, int net_floats, int L, int lane,
                                               const f32x4 (&cond)[NT], f32x4 (&trans)[NT]) {
    f32x4 ls[NT], t[NT];
    mlp_tile<NT, NH, 0>(wblk, L, lane, cond, ls);
    mlp_tile<NT, NH, 1>(wblk + net_floats, L, lane, cond, t);
    float ld = 0.f;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        if (!INVERSE) {  // inputs * exp(log_s) + t ; +sum(log_s)      networks.py:296-298
            trans[tau].x = trans[tau].x * __expf(ls[tau].x) + t[tau].x;
            trans[tau].y = trans[tau].y * __expf(ls[tau].y) + t[tau].y;
            trans[tau].z = trans[tau].z * __expf(ls[tau].z) + t[tau].z;
            trans[tau].w = trans[tau].w * __expf(ls[tau].w) + t[tau].w;
            ld += (ls[tau].x + ls[tau].y) + (ls[tau].z + ls[tau].w);
        } else {  // (inputs - t) * exp(-log_s) ; -sum(log_s)             networks.py:307-309
            trans[tau].x = (trans[tau].x - t[tau].x) * __expf(-ls[tau].x);
            trans[tau].y = (trans[tau].y - t[tau].y) * __expf(-ls[tau].y);
            trans[tau].z = (trans[tau].z - t[tau].z) * __expf(-ls[tau].z);
            trans[tau].w = (trans[tau].w - t[tau].w) * __expf(-ls[tau].w);
            ld -= (ls[tau].x + ls[tau].y) + (ls[tau].z + ls[tau].w);
        }
    }
    return ld;
}

// sum a per-lane partial over the 4 lane groups of a walker (lanes w, w+16, w+32, w+48); every lane
// ends with the walker total.
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ double group_sum(double v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ int group_and(int v) {
    v &= __shfl_xor(v, 16);
    v &= __shfl_xor(v, 32);
    return v;
}

// NormalizingFlow.forward (networks.py:24-32): blocks 0..B-1.  xs[c][tau] = class c tiles.
// Block b conditions on class (b+1)&1 and transforms class b&1 (mask = arange(D)%2 flipped per block).
template <int NT, int NH>
__device__ __forceinline__ float flow_forward_tile(const float *__restrict__ img, int net_floats, int B, int L,
                                                   int lane, f32x4 (&xs)[2][NT]) {
    float ld = 0.f;
    for (int b = 0; b < B; ++b) {
        const float *wblk = img + (size_t)b * 2 * net_floats;
        if (b & 1) ld += coupling_tile<NT, NH, false>(wblk, net_floats, L, lane, xs[0], xs[1]);
        else       ld += coupling_tile<NT, NH, false>(wblk, net_floats, L, lane, xs[1], xs[0]);
    }
    return ld;
}

// NormalizingFlow.inverse (networks.py:34-42): blocks reversed.
template <int NT, int NH>
__device__ __forceinline__ float flow_inverse_tile(const float *__restrict__ img, int net_floats, int B, int L,
                                                   int lane, f32x4 (&xs)[2][NT]) {
    float ld = 0.f;
    for (int b = B - 1; b >= 0; --b) {
        const float *wblk = img + (size_t)b * 2 * net_floats;
        if (b & 1) ld += coupling_tile<NT, NH, true>(wblk, net_floats, L, lane, xs[0], xs[1]);
        else       ld += coupling_tile<NT, NH, true>(wblk, net_floats, L, lane, xs[1], xs[0]);
    }
    return ld;
}

// ---- global <-> tile ------------------------------------------------------------------------------
// lane (g,w) owns dims [32*tau + 8*g, +8) of row (row0 + w): xs[c][tau][r] = dim 32*tau + 8*g + 2*r + c.
template <int NT>
__device__ __forceinline__ void load_tile(const float *__restrict__ rows, long row, bool row_ok, int D, int lane,
                                          f32x4 (&xs)[2][NT]) {
    const int g = lane >> 4;
    const float *p = rows + (size_t)row * D;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int d = 32 * tau + 8 * g + j;
            v[j] = (row_ok && d < D) ? p[d] : 0.f;
        }
        xs[0][tau].x = v[0]; xs[1][tau].x = v[1];
        xs[0][tau].y = v[2]; xs[1][tau].y = v[3];
        xs[0][tau].z = v[4]; xs[1][tau].z = v[5];
        xs[0][tau].w = v[6]; xs[1][tau].w = v[7];
    }
}

template <int NT>
__device__ __forceinline__ void store_tile(float *__restrict__ rows, long row, bool row_ok, int D, int lane,
                                           const f32x4 (&xs)[2][NT]) {
    if (!row_ok) return;
    const int g = lane >> 4;
    float *p = rows + (size_t)row * D;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        float v[8] = {xs[0][tau].x, xs[1][tau].x, xs[0][tau].y, xs[1][tau].y,
                      xs[0][tau].z, xs[1][tau].z, xs[0][tau].w, xs[1][tau].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int d = 32 * tau + 8 * g + j;
            if (d < D) p[d] = v[j];
        }
    }
}

// ---- box prior ------------------------------------------------------------------------------------
// UniformPrior(D,-1,1).__call__ (nnest/priors.py:39-43): out of box iff any(x < -1) or any(x > 1);
// NaN compares false, i.e. counts as inside, exactly like the reference.  Padded dims hold 0.
template <int NT>
__device__ __forceinline__ int inbox_tile(const f32x4 (&xs)[2][NT]) {
    int ok = 1;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            f32x4 v = xs[c][tau];
            ok &= !(v.x < -1.f || v.x > 1.f) & !(v.y < -1.f || v.y > 1.f) & !(v.z < -1.f || v.z > 1.f) &
                  !(v.w < -1.f || v.w > 1.f);
        }
    return group_and(ok);
}

// ---- likelihoods (nnest/likelihoods.py) through safe_loglike (nnest/sampler.py:110-133) --------------
// theta = like_scale * x in float32 (transform = lambda x: s*x on a float32 array, examples/nested/run.py:25-42);
// per-term arithmetic in float32 with the reference's operation order and no FMA contraction; the sum
// over terms is accumulated in float64 (the reference sums in float32; DESIGN.md "Precision").
#pragma clang fp contract(off)
template <int NT>
__device__ __forceinline__ double loglike_tile(int like_id, float scale, int D, int lane, const f32x4 (&xs)[2][NT]) {
    const int g = lane >> 4;
    double acc = 0.0;
    if (like_id == 0) {
        // Rosenbrock (likelihoods.py:51): -sum_i 100*(x[i+1]-x[i]^2)^2 + (1-x[i])^2, i = 0..D-2
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            float th[9];
            th[0] = scale * xs[0][tau].x; th[1] = scale * xs[1][tau].x;
            th[2] = scale * xs[0][tau].y; th[3] = scale * xs[1][tau].y;
            th[4] = scale * xs[0][tau].z; th[5] = scale * xs[1][tau].z;
            th[6] = scale * xs[0][tau].w; th[7] = scale * xs[1][tau].w;
            // first dim of the next 8-block: lane group g+1 of this tile, or group 0 of the next tile
            float nxt_same = __shfl(th[0], (lane + 16) & 63);
            float nxt_tile = 0.f;
            if (tau + 1 < NT) nxt_tile = __shfl(scale * xs[0][tau + 1 < NT ? tau + 1 : tau].x, (lane + 16) & 63);
            th[8] = (g < 3) ? nxt_same : nxt_tile;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int i = 32 * tau + 8 * g + j;
                float a = th[j] * th[j];
                float b = th[j + 1] - a;
                float c = b * b;
                float e = 100.0f * c;
                float f = 1.0f - th[j];
                float q = f * f;
                float term = e + q;
                if (i + 1 < D) acc += (double)term;
            }
        }
        acc = -group_sum(acc);
    } else if (like_id == 1) {
        // GaussianMix (likelihoods.py:165-189): logsumexp_k[ log w_k - |theta - mu_k|^2/2 - (D/2) log 2pi ]
        double base = 0.0;
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            float th[8];
            th[0] = scale * xs[0][tau].x; th[1] = scale * xs[1][tau].x;
            th[2] = scale * xs[0][tau].y; th[3] = scale * xs[1][tau].y;
            th[4] = scale * xs[0][tau].z; th[5] = scale * xs[1][tau].z;
            th[6] = scale * xs[0][tau].w; th[7] = scale * xs[1][tau].w;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int d = 32 * tau + 8 * g + j;
                float sq = th[j] * th[j];
                if (d >= 2 && d < D) base += (double)sq;
            }
        }
        base = group_sum(base);
        const int w = lane & 15;
        float t0 = __shfl(scale * xs[0][0].x, w);  // theta[0], theta[1] live in lane group 0
        float t1 = __shfl(scale * xs[1][0].x, w);
        const float mu0[4] = {0.f, 0.f, 4.f, -4.f}, mu1[4] = {4.f, -4.f, 0.f, 0.f};
        const double lw[4] = {-0.916290731874155, -1.203972804325936, -1.6094379124341003, -2.302585092994046};
        double l[4], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = t0 - mu0[k], b = t1 - mu1[k];
            double s = base + (double)(a * a) + (D > 1 ? (double)(b * b) : 0.0);
            l[k] = -(s * 0.5) - 0.9189385332046727 * (double)D + lw[k];
            mx = l[k] > mx ? l[k] : mx;
        }
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) se += __expf((float)(l[k] - mx));
        acc = mx + (double)__logf(se);
    } else {
        // Himmelblau (likelihoods.py:70) summed over consecutive pairs (x[2i], x[2i+1]) (= reference at D=2)
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            float e[4] = {xs[0][tau].x, xs[0][tau].y, xs[0][tau].z, xs[0][tau].w};
            float o[4] = {xs[1][tau].x, xs[1][tau].y, xs[1][tau].z, xs[1][tau].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int d1 = 2 * (16 * tau + 4 * g + r) + 1;
                float x0 = scale * e[r], x1 = scale * o[r];
                float a = x0 * x0 + x1 - 11.f;
                float b = x0 + x1 * x1 - 7.f;
                float v = -(a * a) - b * b;
                if (d1 < D) acc += (double)v;
            }
        }
        acc = group_sum(acc);
    }
    if (!(fabs(acc) <= 1.79769313486231570e308)) acc = -1e100;  // logl[~isfinite] = -1e100   sampler.py:128
    return acc;
}
#pragma clang fp contract(fast)

// ---- counter-based noise: Philox4x32-10 + Box-Muller (build-defined; DESIGN.md "Proposal noise") ----
struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

enum { NOISE_STREAM_DZ = 0, NOISE_STREAM_U = 1, NOISE_STREAM_JITTER = 2 };

// four N(0,1) draws for dims 4q..4q+3 of (walker, step)
__device__ __forceinline__ f32x4 noise_normal4(uint64_t seed, uint64_t walker, uint32_t step, uint32_t q, uint32_t stream) {
    u32x4 c;
    c.x = q;
    c.y = (uint32_t)walker;
    c.z = step;
    c.w = ((uint32_t)(walker >> 32) & 0x0fffffffu) | (stream << 28);
    u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float two_m32 = 2.3283064365386963e-10f, two_m33 = 1.1641532182693481e-10f;
    float u1 = fmaf((float)r.x, two_m32, two_m33), a1 = (float)r.y * two_m32;
    float u2 = fmaf((float)r.z, two_m32, two_m33), a2 = (float)r.w * two_m32;
    float rad1 = __builtin_sqrtf(-2.0f * __logf(u1)), rad2 = __builtin_sqrtf(-2.0f * __logf(u2));
    f32x4 o;  // v_sin_f32 / v_cos_f32 take revolutions
    o.x = rad1 * __builtin_amdgcn_cosf(a1);
    o.y = rad1 * __builtin_amdgcn_sinf(a1);
    o.z = rad2 * __builtin_amdgcn_cosf(a2);
    o.w = rad2 * __builtin_amdgcn_sinf(a2);
    return o;
}

// one U[0,1) draw (24-bit, like torch.rand on float32) for (walker, step)
__device__ __forceinline__ float noise_uniform(uint64_t seed, uint64_t walker, uint32_t step) {
    u32x4 c;
    c.x = 0;
    c.y = (uint32_t)walker;
    c.z = step;
    c.w = ((uint32_t)(walker >> 32) & 0x0fffffffu) | ((uint32_t)NOISE_STREAM_U << 28);
    u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (float)(r.x >> 8) * 5.9604644775390625e-08f;
}

}  // namespace nnest
