// probe: does global_load_dwordx4 work from a dword-aligned (not 16-byte-aligned) address on this box?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *p, float *out) {
    const float *q = p + 1 + 5 * threadIdx.x;   // dword-aligned only
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
    out[threadIdx.x] = v.x + 10.f * v.y + 100.f * v.z + 1000.f * v.w;
}
int main() {
    float *p, *o; float h[1024], r[64];
    for (int i = 0; i < 1024; ++i) h[i] = (float)(i % 7);
    hipMalloc(&p, sizeof(h)); hipMalloc(&o, sizeof(r)); hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) { const float *q = h + 1 + 5 * t; float w = q[0] + 10.f * q[1] + 100.f * q[2] + 1000.f * q[3]; if (w != r[t]) ++bad; }
    printf("unaligned dwordx4: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return bad != 0;
}
