// nnest_train.hip -- training-side kernels (placeholder while the inference path is brought up)
#include "nnest_internal.h"

namespace nnest {

// training jitter (trainer.py:168-171): 0.2 * mean over both columns of cKDTree(samples).query(samples, 2)
// = 0.2 * sum_i nn_dist(i) / (2N).  Brute force in float64: N is the live-point count (<= ~1e4).
__global__ void __launch_bounds__(256) nn_distance_kernel(const double *__restrict__ X, int N, int D, double *__restrict__ out) {
    __shared__ double red[256];
    double local = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const double *xi = X + (size_t)i * D;
        double best = INFINITY;
        for (int j = 0; j < N; ++j) {
            if (j == i) continue;
            const double *xj = X + (size_t)j * D;
            double s = 0.0;
            for (int d = 0; d < D; ++d) {
                double e = xi[d] - xj[d];
                s += e * e;
            }
            best = s < best ? s : best;
        }
        local += sqrt(best);
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, red[0] * 0.2 / (2.0 * N));
}

hipError_t launch_training_jitter(const double *samples, int N, int D, double *out, hipStream_t st) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st);
    if (e != hipSuccess) return e;
    int grid = (N + 255) / 256;
    hipLaunchKernelGGL(nn_distance_kernel, dim3(grid), dim3(256), 0, st, samples, N, D, out);
    return hipGetLastError();
}

size_t train_workspace_floats(const FlowShape &s, int batch) { (void)s; (void)batch; return 1024; }

hipError_t launch_loss_grad(const float *, const FlowShape &, const float *, int, float *, float *, float *, hipStream_t) {
    return hipErrorNotSupported;
}

hipError_t launch_train(float *, float *, float *, float *, float *, int *, const FlowShape &, const float *, int,
                        const float *, int, const int *, const float *, uint64_t, float, int, int, int, float, float,
                        float *, nnest_train_result_t *, float *, hipStream_t) {
    return hipErrorNotSupported;
}

}  // namespace nnest
