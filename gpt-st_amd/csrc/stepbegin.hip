// Start of a pretraining step as ONE launch: zero the [flat gradient | statistics] buffer (optimizer.zero_grad, BasicTrainer.py:79)
// and the step's zero-initialised scratch arena, gather the time index of node 0 (GPTST.py:256-257), and draw the step's mask noise
// (torch.rand_like of GPTST.py:316,367,391: any uniform [0,1) stream serves; inside a captured hipGraph torch's own generator costs three
// extra launches per replay — two to advance its Philox state and the fill itself — so the noise is drawn here with a counter-based
// Philox4x32-10 keyed by (seed, step counter): reproducible, and identical on every rank of a data-parallel job).
#include "common.h"

// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1) -> four 32-bit words
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__global__ __launch_bounds__(256) void step_begin_kernel(float* __restrict__ z0, long n0, float* __restrict__ z1, long n1,
                                                         const float* __restrict__ src, float* __restrict__ tidx, int BT, int N, int lda,
                                                         int base, float* __restrict__ noise, long n_noise, const int* __restrict__ rng) {
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
    if (noise != nullptr) {                         // four uniforms in [0, 1) per counter value: u = (word >> 8) * 2^-24
        const unsigned seed = (unsigned)rng[0], step = (unsigned)rng[1];
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; 4 * i < n_noise; i += stride) {
            unsigned w[4];
            philox4x32_10((unsigned)i, (unsigned)(i >> 32), step, 0u, seed, 0x5EEDu, w);
            const float4 u = make_float4((w[0] >> 8) * 5.9604644775390625e-8f, (w[1] >> 8) * 5.9604644775390625e-8f,
                                         (w[2] >> 8) * 5.9604644775390625e-8f, (w[3] >> 8) * 5.9604644775390625e-8f);
            if (4 * i + 3 < n_noise) st4(noise + 4 * i, u);
            else { const float uv[4] = {u.x, u.y, u.z, u.w}; for (int k = 0; 4 * i + k < n_noise; ++k) noise[4 * i + k] = uv[k]; }
        }
    }
}

// z0 / z1: buffers to zero (16-byte aligned; z1 may be NULL with n1 = 0); src (BT, N, lda) -> tidx (BT, 2) = src[:, 0, base:base+2] (may be NULL);
// noise (n_noise floats, 16-byte aligned; may be NULL) <- uniform [0,1) from Philox4x32-10 keyed by the device words rng[0] = seed, rng[1] = step
extern "C" int gptst_step_begin(float* z0, long n0, float* z1, long n1, const float* src, float* tidx, int BT, int N, int lda, int base,
                                float* noise, long n_noise, const int* rng, void* stream) {
    if (!z0 || n0 <= 0 || n1 < 0 || (n1 > 0 && !z1) || (tidx && (!src || BT <= 0 || N <= 0 || lda < base + 2))) return GPTST_EARG;
    if (noise && (n_noise <= 0 || !rng)) return GPTST_EARG;
    long q = (n0 + n1) / 4;
    int nb = (int)((q + 255) / 256); if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
    hipLaunchKernelGGL(step_begin_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, z0, n0, z1, n1, src, tidx, BT, N, lda, base, noise,
                       noise ? n_noise : 0L, rng);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
