// Start of a pretraining step as ONE launch: zero the [flat gradient | statistics] buffer (optimizer.zero_grad, BasicTrainer.py:79)
// and the step's zero-initialised scratch arena, and gather the time index of node 0 (GPTST.py:256-257) — three tiny launches before.
#include "common.h"

__global__ __launch_bounds__(256) void step_begin_kernel(float* __restrict__ z0, long n0, float* __restrict__ z1, long n1,
                                                         const float* __restrict__ src, float* __restrict__ tidx, int BT, int N, int lda,
                                                         int base) {
    const long q0 = n0 / 4, q1 = n1 / 4, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < q0 + q1; i += stride) {
        if (i < q0) st4(z0 + 4 * i, f4zero()); else st4(z1 + 4 * (i - q0), f4zero());
    }
    if (blockIdx.x == 0) {
        for (long i = 4 * q0 + threadIdx.x; i < n0; i += 256) z0[i] = 0.f;          // tails that are not a multiple of 4 floats
        for (long i = 4 * q1 + threadIdx.x; i < n1; i += 256) z1[i] = 0.f;
    }
    if (tidx != nullptr)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 2L * BT; i += stride)
            tidx[i] = src[(size_t)(i / 2) * N * lda + base + (i & 1)];
}

// z0 / z1: buffers to zero (16-byte aligned; z1 may be NULL with n1 = 0); src (BT, N, lda) -> tidx (BT, 2) = src[:, 0, base:base+2] (may be NULL)
extern "C" int gptst_step_begin(float* z0, long n0, float* z1, long n1, const float* src, float* tidx, int BT, int N, int lda, int base,
                                void* stream) {
    if (!z0 || n0 <= 0 || n1 < 0 || (n1 > 0 && !z1) || (tidx && (!src || BT <= 0 || N <= 0 || lda < base + 2))) return GPTST_EARG;
    long q = (n0 + n1) / 4;
    int nb = (int)((q + 255) / 256); if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
    hipLaunchKernelGGL(step_begin_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, z0, n0, z1, n1, src, tidx, BT, N, lda, base);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
