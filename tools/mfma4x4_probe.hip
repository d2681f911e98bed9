// Probe for the 4-walker ("quad") tile design: layout and timing of v_mfma_f32_4x4x1_16B_f32, DPP row rotations and
// v_permlane{16,32}_swap on gfx950.  Developer tool: hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_probe.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(float *out) {
    const int lane = threadIdx.x;
    // A value encodes (block, i): 100*block + i + 1 ; B encodes (block, j): 1000*block + 10*(j+1)... use distinct primes instead
    float a = (float)(lane + 1);            // lane l supplies A for (block l/4, i = l%4)?
    float b = (float)(64 + lane + 1) ;      // lane l supplies B for (block l/4, j = l%4)?
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    out[lane * 4 + 0] = c.x; out[lane * 4 + 1] = c.y; out[lane * 4 + 2] = c.z; out[lane * 4 + 3] = c.w;
    // DPP row_ror:4 and permlane swaps
    int v = lane;
    int ror4 = __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);
    int ror8 = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);
    out[256 + lane] = (float)ror4;
    out[320 + lane] = (float)ror8;
    unsigned x = 1000 + lane, y = 2000 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[384 + lane] = (float)r[0];
    out[448 + lane] = (float)r[1];
    auto q = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    out[512 + lane] = (float)q[0];
    out[576 + lane] = (float)q[1];
}

#define STAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ float fast_tanh(float x) {
    float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// reduce-scatter of a 4-register accumulator over the four 16-lane rows: row k ends with register k summed over rows
__device__ __forceinline__ float reduce_rows(f32x4 p) {
    auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p.x), __float_as_uint(p.z), false, false);
    auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p.y), __float_as_uint(p.w), false, false);
    float a = __uint_as_float(s0[0]) + __uint_as_float(s0[1]);   // [R0 | R2]
    float b = __uint_as_float(s1[0]) + __uint_as_float(s1[1]);   // [R1 | R3]
    auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

__global__ void timing_kernel(float *out, unsigned long long *cyc, int iters) {
    const int lane = threadIdx.x;
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = 0.01f * (float)((lane * 7 + i * 13) % 17 - 8);
    unsigned long long t0, t1, t2, t3, t4;
    // (1) dependent chain of 4x4x1 MFMAs on one accumulator
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    float bv = 0.5f + 0.001f * lane;
    STAMP(t0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i], bv, c, 0, 0, 0);
    }
    STAMP(t1);
    // (2) four independent accumulators
    f32x4 c0 = c, c1 = c, c2 = c, c3 = c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i], bv, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i + 1], bv, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i + 2], bv, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i + 3], bv, c3, 0, 0, 0);
        }
    }
    STAMP(t2);
    // (3) one hidden layer of the quad design, dependent from iteration to iteration:
    //     3 DPP movs + 4 MFMAs + row reduce-scatter + tanh
    float h = bv;
    for (int it = 0; it < iters; ++it) {
        float h1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, h), 0x124, 0xf, 0xf, false));
        float h2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, h), 0x128, 0xf, 0xf, false));
        float h3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, h), 0x12c, 0xf, 0xf, false));
        f32x4 p = {0.1f, 0.f, 0.f, 0.f};
        p = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0], h, p, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f32_4x4x1f32(w[1], h1, p, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f32_4x4x1f32(w[2], h2, p, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f32_4x4x1f32(w[3], h3, p, 0, 0, 0);
        h = fast_tanh(reduce_rows(p));
    }
    STAMP(t3);
    // (4) the same with two independent nets interleaved (scale: tanh, translate: relu)
    float hs = h, ht = bv;
    for (int it = 0; it < iters; ++it) {
        float s1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, hs), 0x124, 0xf, 0xf, false));
        float s2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, hs), 0x128, 0xf, 0xf, false));
        float s3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, hs), 0x12c, 0xf, 0xf, false));
        float u1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ht), 0x124, 0xf, 0xf, false));
        float u2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ht), 0x128, 0xf, 0xf, false));
        float u3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ht), 0x12c, 0xf, 0xf, false));
        f32x4 p = {0.1f, 0.f, 0.f, 0.f}, q = {0.2f, 0.f, 0.f, 0.f};
        p = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0], hs, p, 0, 0, 0);
        q = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4], ht, q, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f32_4x4x1f32(w[1], s1, p, 0, 0, 0);
        q = __builtin_amdgcn_mfma_f32_4x4x1f32(w[5], u1, q, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f32_4x4x1f32(w[2], s2, p, 0, 0, 0);
        q = __builtin_amdgcn_mfma_f32_4x4x1f32(w[6], u2, q, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f32_4x4x1f32(w[3], s3, p, 0, 0, 0);
        q = __builtin_amdgcn_mfma_f32_4x4x1f32(w[7], u3, q, 0, 0, 0);
        hs = fast_tanh(reduce_rows(p));
        ht = fmaxf(reduce_rows(q), 0.f);
    }
    STAMP(t4);
    out[lane] = c.x + c0.x + c1.y + c2.z + c3.w + h + hs + ht;
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}

int main() {
    float *d; unsigned long long *dc;
    hipMalloc(&d, 1024 * sizeof(float));
    hipMalloc(&dc, 8 * sizeof(unsigned long long));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(1024);
    hipMemcpy(h.data(), d, 640 * sizeof(float), hipMemcpyDeviceToHost);
    // hypothesis: D[r] at lane l = A(lane 4*(l/4) + r) * B(lane l)
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            float want = (float)(4 * (l / 4) + r + 1) * (float)(64 + l + 1);
            if (h[l * 4 + r] != want) ++bad;
        }
    printf("mfma_f32_4x4x1 layout hypothesis D[r]@lane l = A@lane(4*(l/4)+r) * B@lane l : %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    printf("row_ror:4  lane0<-%g lane5<-%g lane20<-%g\n", h[256 + 0], h[256 + 5], h[256 + 20]);
    printf("row_ror:8  lane0<-%g lane5<-%g lane20<-%g\n", h[320 + 0], h[320 + 5], h[320 + 20]);
    printf("permlane32_swap(x=1000+l, y=2000+l): r0 lanes 0,31,32,63 = %g %g %g %g ; r1 = %g %g %g %g\n", h[384], h[384+31], h[384+32], h[384+63], h[448], h[448+31], h[448+32], h[448+63]);
    printf("permlane16_swap: r0 lanes 0,16,32,48 = %g %g %g %g ; r1 = %g %g %g %g\n", h[512], h[512+16], h[512+32], h[512+48], h[576], h[576+16], h[576+32], h[576+48]);
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(timing_kernel, dim3(1), dim3(64), 0, 0, d, dc, iters);
    unsigned long long c[4];
    hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    printf("dependent 4x4x1 chain: %.1f cycles per MFMA\n", (double)c[0] / (iters * 16));
    printf("4 independent accumulators: %.1f cycles per MFMA\n", (double)c[1] / (iters * 16));
    printf("one quad hidden layer (3 dpp + 4 mfma + reduce-scatter + tanh), dependent: %.1f cycles\n", (double)c[2] / iters);
    printf("two nets interleaved (tanh + relu): %.1f cycles per layer pair\n", (double)c[3] / iters);
    return 0;
}
