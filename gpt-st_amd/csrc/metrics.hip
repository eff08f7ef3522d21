// Evaluation metrics of Trainer.test (reference model/BasicTrainer.py:209-248, lib/metrics.py:11-18,38-43,52-86,206-228), accumulated on
// the device over the batches of the evaluation loader: no (samples, T, N, D) tensor of predictions is ever concatenated.
//   per horizon t:        [n1, sum|e|, sum e^2, n2, sum|e/y|]        n1: cells with y > mae_thresh (all if none), n2: y > mape_thresh
//   per (t, node n):      [K, sum p, sum y, sum p^2, sum y^2, sum p y]  over (batch, channel) — the moments CORR needs
// p / y are formed on the fly as the reference does in pretrain mode (:229-235): y = inverse(label * m), p = inverse(out * m), where m
// marks the masked cells.  Sums are doubles (atomicAdd f64).
#include "common.h"

__global__ __launch_bounds__(256) void metrics_accum_kernel(const float* __restrict__ out, const float* __restrict__ src, int lda,
                                                            const float* __restrict__ vis, float sigma, float mu, int has_mae_thresh,
                                                            float mae_thresh, float mape_thresh, int T, int N, int D,
                                                            double* __restrict__ sums_t, double* __restrict__ sums_tn) {
    __shared__ double red[4][5];
    const int bt = blockIdx.x, t = bt % T;
    double a[5] = {0, 0, 0, 0, 0};
    for (int n = threadIdx.x; n < N; n += 256) {
        double m[6] = {0, 0, 0, 0, 0, 0};
        for (int d = 0; d < D; ++d) {
            const size_t cell = ((size_t)bt * N + n) * D + d;
            const float msk = vis != nullptr ? 1.f - vis[cell] : 1.f;
            const float p = (out[cell] * msk) * sigma + mu;
            const float y = (src[((size_t)bt * N + n) * lda + d] * msk) * sigma + mu;
            const float e = y - p;
            if (!has_mae_thresh || y > mae_thresh) { a[0] += 1.0; a[1] += fabsf(e); a[2] += (double)e * e; }
            if (y > mape_thresh) { a[3] += 1.0; a[4] += fabsf(e / y); }
            m[0] += 1.0; m[1] += p; m[2] += y; m[3] += (double)p * p; m[4] += (double)y * y; m[5] += (double)p * y;
        }
        double* o = sums_tn + ((size_t)t * N + n) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(o + k, m[k]);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double v = a[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) atomicAdd(sums_t + (size_t)t * 5 + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out (B*T*N, D) model output, src (B*T*N, lda) normalised input whose first D channels are the label, vis (B*T*N*D) visibility mask
// (1 = visible; NULL -> no masking: plain prediction metrics).  sums_t (T,5) and sums_tn (T,N,6) doubles are ACCUMULATED.
extern "C" int gptst_metrics_accum(const float* out, const float* src, int lda, const float* vis, float sigma, float mu, int has_mae_thresh,
                                   float mae_thresh, float mape_thresh, int B, int T, int N, int D, double* sums_t, double* sums_tn,
                                   void* stream) {
    if (!out || !src || !sums_t || !sums_tn || B <= 0 || T <= 0 || N <= 0 || D <= 0) return GPTST_EARG;
    hipLaunchKernelGGL(metrics_accum_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, out, src, lda, vis, sigma, mu, has_mae_thresh,
                       mae_thresh, mape_thresh, T, N, D, sums_t, sums_tn);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
