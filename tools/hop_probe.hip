// tools/hop_probe.hip -- what one cross-CU hand-off costs on gfx950, by placement and by store / load flavour (developer tool).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/hop_probe.hip -o tools/bin/hop_probe && tools/bin/hop_probe
// Two workgroups ping-pong a word: A writes tag i to slot 0, B polls slot 0 for i and writes tag i to slot 1, A polls slot 1.
// One round = two one-way hops.  Workgroups carry 100 KB of LDS so that each has a compute unit of its own; `partner` picks
// which block answers block 0: 8 = the same XCD under round-robin placement, 1 = the next XCD.  Every poll is bounded.
// Also: a fan-in / fan-out round of G workgroups (all arrive on a counter, all poll it), the shape of K5's grid barrier.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { ST_SC1 = 0, ST_PLAIN = 1, ST_SC0SC1 = 2, ST_ATOMIC_AGENT = 3, ST_ATOMIC_WG = 4 };
enum { LD_SC1 = 0, LD_SC0SC1 = 1, LD_ATOMIC_AGENT = 2 };

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

template <int ST> __device__ __forceinline__ void put(unsigned *p, unsigned v) {
    if (ST == ST_SC1) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    if (ST == ST_PLAIN) asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
    if (ST == ST_SC0SC1) asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
    if (ST == ST_ATOMIC_AGENT) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ST == ST_ATOMIC_WG) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int LD> __device__ __forceinline__ unsigned get(const unsigned *p) {
    unsigned v;
    if (LD == LD_SC1) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == LD_SC0SC1) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == LD_ATOMIC_AGENT) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

// out[0] = ticks (100 MHz) of block 0 for `iters` rounds, out[1] = failures, out[2 + b] = XCC id of block b
template <int ST, int LD>
__global__ void pingpong(unsigned *slots, int partner, int iters, unsigned long long *out) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) { lds[0] = 0.f; out[2 + blockIdx.x] = xcc_id(); }
    if ((int)blockIdx.x != 0 && (int)blockIdx.x != partner) return;
    if (threadIdx.x != 0) return;
    const bool a = blockIdx.x == 0;
    unsigned *mine = slots + (a ? 0 : 64), *theirs = slots + (a ? 64 : 0);   // (256 bytes apart: lines of their own)
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned fails = 0;
    for (int i = 1; i <= iters; ++i) {
        if (a) put<ST>(mine, (unsigned)i);
        int polls = 0;
        while (get<LD>(theirs) < (unsigned)i) {
            if (++polls > 200000) { fails++; break; }
        }
        if (!a) put<ST>(mine, (unsigned)i);
        if (fails) break;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (a) { out[0] = t1 - t0; out[1] = fails; }
    else if (fails) out[1] = fails;
}

// fan-in / fan-out: the participating blocks (blockIdx % stride == 0, G of them) add to one counter and poll it, `iters` times
template <int ST, int LD>
__global__ void fanin(unsigned *ctr, int stride, int G, int iters, unsigned long long *out) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) lds[0] = 0.f;
    if ((int)blockIdx.x % stride != 0 || (int)blockIdx.x / stride >= G) return;
    if (threadIdx.x != 0) return;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned fails = 0;
    for (int i = 1; i <= iters; ++i) {
        put<ST>(ctr, 0u);
        int polls = 0;
        while (get<LD>(ctr) < (unsigned)(i * G)) {
            if (++polls > 200000) { fails++; break; }
        }
        if (fails) break;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0) { out[0] = t1 - t0; out[1] = fails; }
}

// payload hand-off: block 0 stores `n16` x 16 bytes per lane of one wave (tagged in every word), the partner polls the LAST
// word and then loads all of it; the partner answers with one word.  Measures hop + payload, one producer wave.
template <bool PLAIN>
__global__ void payload(float *buf, unsigned *slots, int partner, int n16, int iters, unsigned long long *out) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) lds[0] = 0.f;
    if ((int)blockIdx.x != 0 && (int)blockIdx.x != partner) return;
    if (threadIdx.x >= 64) return;
    const bool a = blockIdx.x == 0;
    const int lane = threadIdx.x;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned fails = 0;
    float sink = 0.f;
    for (int i = 1; i <= iters; ++i) {
        const float tg = __uint_as_float((unsigned)i);
        if (a) {
            for (int k = 0; k < n16; ++k) {
                float *p = buf + ((size_t)k * 64 + lane) * 4;
                f32x4 v = {tg, tg, tg, tg};
                if (PLAIN) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
            }
            int polls = 0;
            while (get<LD_SC1>(slots + 64) < (unsigned)i) if (++polls > 200000) { fails++; break; }
        } else {
            // poll the data itself: all loads in flight, again until every tag matches
            int polls = 0;
            bool fresh = false;
            while (!fresh) {
                fresh = true;
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float *p = buf + ((size_t)(k < n16 ? k : 0) * 64 + lane) * 4;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k]) : "v"(p) : "memory");
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[k]) : : "memory");
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    fresh = fresh && __float_as_uint(v[k].x) == (unsigned)i && __float_as_uint(v[k].w) == (unsigned)i;
                    sink += v[k].y;
                }
                fresh = __all(fresh);
                if (++polls > 200000) { fails++; break; }
            }
            if (lane == 0) put<ST_SC1>(slots + 64, (unsigned)i);
        }
        if (fails) break;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (a && lane == 0) { out[0] = t1 - t0; out[1] = fails; }
    if (sink == 12345.f) out[40] = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int iters = 2000, NB = 64;
    const size_t lds = 100 * 1024;
    unsigned *slots;
    unsigned long long *out, hout[2 + NB + 64];
    float *buf;
    CK(hipMalloc(&slots, 4096));
    CK(hipMalloc(&buf, 1 << 20));
    CK(hipMalloc(&out, sizeof(hout)));
#define SETLDS(k) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
#define RUN_PP(ST, LD, partner, label) do { \
        SETLDS((pingpong<ST, LD>)); \
        CK(hipMemset(slots, 0, 4096)); CK(hipMemset(out, 0, sizeof(hout))); \
        hipLaunchKernelGGL((pingpong<ST, LD>), dim3(NB), dim3(64), lds, 0, slots, partner, iters, out); \
        CK(hipDeviceSynchronize()); CK(hipMemcpy(hout, out, sizeof(hout), hipMemcpyDeviceToHost)); \
        printf("pingpong %-34s partner %2d (xcc %llu vs %llu): %7.1f ns per one-way hop, fails %llu\n", label, partner, hout[2], hout[2 + partner], \
               hout[0] * 10.0 / iters / 2, hout[1]); } while (0)
    for (int partner : {8, 1, 16, 3}) {
        RUN_PP(ST_SC1, LD_SC1, partner, "st sc1 / ld sc1");
        RUN_PP(ST_PLAIN, LD_SC1, partner, "st plain / ld sc1");
        RUN_PP(ST_SC0SC1, LD_SC0SC1, partner, "st sc0 sc1 / ld sc0 sc1");
        RUN_PP(ST_SC1, LD_ATOMIC_AGENT, partner, "st sc1 / atomic-load agent");
        RUN_PP(ST_ATOMIC_AGENT, LD_SC1, partner, "atomic add agent / ld sc1");
        RUN_PP(ST_ATOMIC_WG, LD_SC1, partner, "atomic add wg-scope / ld sc1");
    }
    printf("xcc of blocks 0..23:");
    for (int b = 0; b < 24; ++b) printf(" %llu", hout[2 + b]);
    printf("\n");
#define RUN_FI(ST, LD, stride, G, label) do { \
        SETLDS((fanin<ST, LD>)); \
        CK(hipMemset(slots, 0, 4096)); CK(hipMemset(out, 0, sizeof(hout))); \
        hipLaunchKernelGGL((fanin<ST, LD>), dim3(stride * G), dim3(64), lds, 0, slots, stride, G, iters, out); \
        CK(hipDeviceSynchronize()); CK(hipMemcpy(hout, out, sizeof(hout), hipMemcpyDeviceToHost)); \
        printf("fanin    %-34s stride %d G %2d: %7.1f ns per barrier, fails %llu\n", label, stride, G, hout[0] * 10.0 / iters, hout[1]); } while (0)
    RUN_FI(ST_ATOMIC_AGENT, LD_SC1, 1, 25, "atomic agent / ld sc1");
    RUN_FI(ST_ATOMIC_AGENT, LD_SC1, 8, 25, "atomic agent / ld sc1");
    RUN_FI(ST_ATOMIC_WG, LD_SC1, 8, 25, "atomic wg-scope / ld sc1");
    RUN_FI(ST_ATOMIC_WG, LD_SC1, 1, 25, "atomic wg-scope / ld sc1 (expected to fail)");
    RUN_FI(ST_ATOMIC_AGENT, LD_SC1, 1, 8, "atomic agent / ld sc1");
    RUN_FI(ST_ATOMIC_AGENT, LD_SC1, 8, 8, "atomic agent / ld sc1");
    RUN_FI(ST_ATOMIC_WG, LD_SC1, 8, 8, "atomic wg-scope / ld sc1");
#define RUN_PL(PLAIN, partner, n16, label) do { \
        SETLDS((payload<PLAIN>)); \
        CK(hipMemset(slots, 0, 4096)); CK(hipMemset(out, 0, sizeof(hout))); CK(hipMemset(buf, 0, 1 << 20)); \
        hipLaunchKernelGGL((payload<PLAIN>), dim3(NB), dim3(64), lds, 0, buf, slots, partner, n16, iters, out); \
        CK(hipDeviceSynchronize()); CK(hipMemcpy(hout, out, sizeof(hout), hipMemcpyDeviceToHost)); \
        printf("payload  %-20s partner %2d, %2d x 1 KB: %7.1f ns per round (data out + word back), fails %llu\n", label, partner, n16, \
               hout[0] * 10.0 / iters, hout[1]); } while (0)
    for (int n16 : {1, 4, 8}) {
        RUN_PL(false, 1, n16, "st sc1 x4");
        RUN_PL(false, 8, n16, "st sc1 x4");
        RUN_PL(true, 8, n16, "st plain x4");
    }
    return 0;
}
