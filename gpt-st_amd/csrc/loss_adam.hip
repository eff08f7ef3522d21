// Loss and optimiser tail of the pretraining step.
//   mae   : scaler_mae_loss (reference Run.py:92-100) + MAE_torch (lib/metrics.py:11-18) + inverse_transform
//           (lib/normalization.py:23-27):  p=(out*s+m)*M, y=(label*s+m)*M, keep = y > thresh, loss = mean_keep |y-p|
//   kl    : 0.1 * KLDivLoss(sum)(log prob, eb) (Run.py:132, BasicTrainer.py:85) fused with the softmax backward of MLP_RL
//   adam  : clip_grad_norm_(5) + Adam (BasicTrainer.py:95-97, Run.py:134) as ONE pass over a flat parameter buffer
// stats (device float[8]): [0] sum |y-p| over kept cells, [1] kept count, [2] sum eb*(log eb - log prob), [3] extra sum g^2 terms in,
// [4] total sum g^2 (scaled) out, [6] [7] tickets of the ordered folds in tails.hip
// Everything stays on the device: the reference syncs on loss.item() and masked_select every step.
#include "common.h"

// mask: 1 = visible, 0 = masked (reconstruction target);  the reference multiplies with (1 - mask)
__global__ __launch_bounds__(256) void mae_fwd_kernel(const float* __restrict__ out, const float* __restrict__ src, int lda,
                                                      const float* __restrict__ mask, float sigma, float mu, float thresh, int rows,
                                                      int J, float* __restrict__ stats) {
    __shared__ float red[2][4];
    float ls = 0.f, cnt = 0.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)rows * J; e += (size_t)gridDim.x * 256) {
        const size_t i = e / J; const int j = (int)(e % J);
        const float M = 1.f - mask[e];
        const float p = (out[e] * sigma + mu) * M;
        const float y = (src[i * lda + j] * sigma + mu) * M;
        if (y > thresh) { ls += fabsf(y - p); cnt += 1.f; }
    }
    ls = group_sum<64>(ls); cnt = group_sum<64>(cnt);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ls; red[1][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(stats + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(stats + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// dOut = d loss / d out;  normalize: divide by the kept count in stats[1] (single GPU);  otherwise gradient of the SUM
// (data-parallel: the count is all-reduced with the gradients and applied in the optimiser)
__global__ __launch_bounds__(256) void mae_bwd_kernel(const float* __restrict__ out, const float* __restrict__ src, int lda,
                                                      const float* __restrict__ mask, float sigma, float mu, float thresh, int rows,
                                                      int J, const float* __restrict__ stats, int normalize, float* __restrict__ dOut) {
    const float inv = normalize ? 1.f / fmaxf(stats[1], 1.f) : 1.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)rows * J; e += (size_t)gridDim.x * 256) {
        const size_t i = e / J; const int j = (int)(e % J);
        const float M = 1.f - mask[e];
        const float p = (out[e] * sigma + mu) * M;
        const float y = (src[i * lda + j] * sigma + mu) * M;
        float g = 0.f;
        if (y > thresh) {
            const float d = p - y;
            g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * M * sigma * inv;
        }
        dOut[e] = g;
    }
}

// prob (rows, HS) row-major;  eb = cap1 soft assignment c laid out (BT, HS, N): eb[i=(bt,n)][h] = c[(bt*HS + h)*N + n].
// stats[2] += sum eb*(log eb - log prob);   dlogit[i][h] = w * (prob[i][h] * sum_h eb - eb[i][h])   (softmax + KL backward)
__global__ __launch_bounds__(256) void kl_kernel(const float* __restrict__ prob, const float* __restrict__ c, int rows, int N, int HS,
                                                 float w, float* __restrict__ dlogit, float* __restrict__ stats) {
    __shared__ float red[4];
    float kl = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)rows; i += (size_t)gridDim.x * 256) {
        const size_t bt = i / N, n = i % N;
        const float* cp = c + bt * HS * N + n;
        float se = 0.f;
        for (int h = 0; h < HS; ++h) {
            const float e = cp[(size_t)h * N], p = prob[i * HS + h];
            se += e;
            if (e > 0.f) kl += e * (logf(e) - logf(p));
        }
        if (dlogit != nullptr)
            for (int h = 0; h < HS; ++h) dlogit[i * HS + h] = w * (prob[i * HS + h] * se - cp[(size_t)h * N]);
    }
    kl = group_sum<64>(kl);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = kl;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(stats + 2, red[0] + red[1] + red[2] + red[3]);
}

// hyper (device float[16]), refreshed by the host before every step:
//  [0] step_size_A = lr/(1-b1^tA)  [1] bc2sqrt_A = sqrt(1-b2^tA)  [2] step_size_B  [3] bc2sqrt_B  [4] beta1  [5] beta2  [6] eps
//  [7] max_norm (<=0: no clipping)  [8] active_B (0/1)  [9] gscale_A mode: 0 -> 1, 1 -> 1/max(stats[1],1)   [10] gscale_B
// segment A = parameters on the reconstruction-loss path, B = MLP_RL / teb4mask / neb4mask (KL path; no gradient - hence no
// Adam state, as in torch - until epoch > change_epoch);  parameters after nA+nB never receive gradients.
__device__ __forceinline__ float seg_scale(const float* hyper, const float* stats, bool segA) {
    if (segA) return hyper[9] != 0.f ? 1.f / fmaxf(stats[1], 1.f) : 1.f;
    return hyper[10];
}

#define GN_NB 256      // workgroups of gradnorm_kernel = partial sums in ws

// ws[2*blk], ws[2*blk+1] = this workgroup's partial of sum g^2 over segment A / B, UNSCALED — adam_kernel applies the squared segment scales and
// folds the GN_NB partials in a fixed order (no atomics).  sws != NULL: workgroup 0 also folds the step's loss statistics (what
// gptst_stats_fold does: stats[0..2] += column sums of sws (sws_rows, 4) in a fixed order) — the scales depend on stats[1], which is why the
// squares stay unscaled here and one launch per step goes away (r03).
__global__ __launch_bounds__(256) void gradnorm_kernel(const float* __restrict__ g, long nA, long nB, const float* __restrict__ hyper,
                                                       float* __restrict__ stats, float* __restrict__ ws, const float* __restrict__ sws,
                                                       int sws_rows) {
    // float4 loads, 4 in flight per thread (tensors are padded to 16 bytes, so nA % 4 == 0 and a float4 never straddles the
    // segment boundary); the first version walked scalars with one load in flight: 15 us for 4 MB whatever the grid size.
    __shared__ float red[2][4];
    __shared__ float reds[3][4];
    const bool actB = hyper[8] != 0.f;
    const long n4 = (nA + (actB ? nB : 0)) / 4, nA4 = nA / 4;
    const long stride = (long)gridDim.x * 256;
    float sA = 0.f, sB = 0.f;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u * stride;
            v[u] = i < n4 ? ld4(g + 4 * i) : f4zero();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float q = fmaf(v[u].x, v[u].x, fmaf(v[u].y, v[u].y, fmaf(v[u].z, v[u].z, v[u].w * v[u].w)));
            if ((i0 + u * stride) < nA4) sA += q; else sB += q;
        }
    }
    sA = group_sum<64>(sA); sB = group_sum<64>(sB);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sA; red[1][threadIdx.x >> 6] = sB; }
    __syncthreads();
    if (threadIdx.x < 2) ws[2 * blockIdx.x + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    if (sws != nullptr && blockIdx.x == 0) {                     // == stats_fold_kernel (tails.hip), same order of additions
        float s[3] = {0.f, 0.f, 0.f};
        for (int r = threadIdx.x; r < sws_rows; r += 256) {
            const float4 v = ld4(sws + 4 * (size_t)r);
            s[0] += v.x; s[1] += v.y; s[2] += v.z;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            s[k] = group_sum<64>(s[k]);
            if ((threadIdx.x & 63) == 0) reds[k][threadIdx.x >> 6] = s[k];
        }
        __syncthreads();
        if (threadIdx.x < 3) stats[threadIdx.x] += (reds[threadIdx.x][0] + reds[threadIdx.x][1]) + (reds[threadIdx.x][2] + reds[threadIdx.x][3]);
    }
}

// updates skipped by the guard below since the last gptst_handoff_reset(): goes out in stats_out[6], so that a host that enqueued SEVERAL steps before it
// looked (bench loops, step groups that fell back to single steps) knows how many optimiser steps to take back
__device__ unsigned g_adam_skipped = 0u;
GPTST_INTERNAL int gptst_adam_skipped_clear(void) {
    const unsigned z = 0u;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_adam_skipped), &z, sizeof(unsigned)) == hipSuccess ? 0 : 1;
}

// stats[3] on entry: squared-norm contributions that are not in g on this rank (node-sharded runs add the other ranks' node-local
// parts there; otherwise 0); on exit (written by workgroup 0): the total squared gradient norm.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long nA, long nB, const float* __restrict__ hyper,
                                                   float* __restrict__ stats, const float* __restrict__ ws, int nws, float* __restrict__ stats_out,
                                                   const unsigned* __restrict__ lost0, const unsigned* __restrict__ lost1, const unsigned* __restrict__ lost2) {
    __shared__ float red[4];
    __shared__ float redB[4];
    // guard (r05): a bounded in-launch hand-off that expired somewhere in this step poisoned a gradient with NaN (gptst_wait_ge).  The update is
    // then SKIPPED — weights and moments stay as they were — and the count goes out in stats_out[5]; the host re-runs the step on the launches
    // without hand-offs (step.py) instead of losing the run.  The counters stay up (every later step is skipped too) until gptst_handoff_reset().
    // (stats[5]: the count gptst_stats_fold put in front of a gradient all-reduce — the sum over the ranks, so that all of them skip together)
    const unsigned nlost = (lost0 != nullptr ? *lost0 : 0u) + (lost1 != nullptr ? *lost1 : 0u) + (lost2 != nullptr ? *lost2 : 0u) +
                           ((lost0 != nullptr && stats[5] > 0.f) ? (unsigned)stats[5] : 0u);
    {   // every workgroup folds the gradient-norm partials (per segment, unscaled) in the same fixed order
        float sA = (int)threadIdx.x < nws ? ws[2 * threadIdx.x] : 0.f, sB = (int)threadIdx.x < nws ? ws[2 * threadIdx.x + 1] : 0.f;
        sA = group_sum<64>(sA); sB = group_sum<64>(sB);
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = sA; redB[threadIdx.x >> 6] = sB; }
        __syncthreads();
    }
    const float ua = seg_scale(hyper, stats, true), ub = seg_scale(hyper, stats, false);
    const float gsq = stats[3] + (ua * ua * ((red[0] + red[1]) + (red[2] + red[3])) + ub * ub * ((redB[0] + redB[1]) + (redB[2] + redB[3])));
    const bool actB = hyper[8] != 0.f;
    const long n = nA + (actB ? nB : 0);
    const float b1 = hyper[4], b2 = hyper[5], eps = hyper[6], maxn = hyper[7];
    // 1 - beta as the HOST rounds it (hyper[11], hyper[12]): torch.optim.Adam passes python's 1 - beta2 (0.001 rounded to fp32); 1.f - 0.999f is
    // 0.00099998713 — 1.3e-5 low, i.e. every step 6.4e-6 too long (r04: found by the fp64 trajectory test, tests/test_gpu_step.py)
    const float omb1 = hyper[11] != 0.f ? hyper[11] : 1.f - b1, omb2 = hyper[12] != 0.f ? hyper[12] : 1.f - b2;
    float clip = 1.f;
    if (maxn > 0.f) clip = fminf(1.f, maxn / (sqrtf(gsq) + 1e-6f));               // clip_grad_norm_
    const float sa = seg_scale(hyper, stats, true) * clip, sb = seg_scale(hyper, stats, false) * clip;
    // four consecutive elements per thread (tensors are padded to 16 bytes: nA, nB are multiples of 4, so a float4 never straddles the segments; r06:
    // scalar loads ran this 29 MB pass at 3.4 TB/s)
    for (long i = 4 * ((long)blockIdx.x * 256 + threadIdx.x); i < n && nlost == 0u; i += 4 * (long)gridDim.x * 256) {
        const bool A = i < nA;
        const float sc = A ? sa : sb, step = A ? hyper[0] : hyper[2], bc2 = A ? hyper[1] : hyper[3];
        const float4 g4 = ld4(g + i), p4 = ld4(p + i);
        float4 m4 = ld4(m + i), v4 = ld4(v + i);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, pv[4] = {p4.x, p4.y, p4.z, p4.w};
        float mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, po[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gv[e] * sc;
            float mi = mv[e], vi = vv[e];
            mi = mi + (gr - mi) * omb1;                        // exp_avg.lerp_(grad, 1-beta1)
            vi = vi * b2 + omb2 * gr * gr;                     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
            const float denom = sqrtf(vi) / bc2 + eps;
            po[e] = pv[e] - step * (mi / denom);               // param.addcdiv_(exp_avg, denom, -step_size)
            mv[e] = mi; vv[e] = vi;
        }
        st4(p + i, make_float4(po[0], po[1], po[2], po[3]));
        st4(m + i, make_float4(mv[0], mv[1], mv[2], mv[3]));
        st4(v + i, make_float4(vv[0], vv[1], vv[2], vv[3]));
    }
    // (stats[3] is rewritten only after every workgroup has read it: the caller's next kernel boundary orders that — here the
    // total goes to stats[4], which nobody reads inside this launch)
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[4] = gsq;
    if (blockIdx.x == 0) {
        __shared__ unsigned s_skipped;
        if (threadIdx.x == 0) s_skipped = nlost != 0u ? (g_adam_skipped += 1u) : g_adam_skipped;      // (only workgroup 0 touches the word)
        __syncthreads();
        if (stats_out != nullptr && threadIdx.x < 8)
            stats_out[threadIdx.x] = threadIdx.x == 4 ? gsq : threadIdx.x == 5 ? (float)nlost : threadIdx.x == 6 ? (float)s_skipped : stats[threadIdx.x];
    }
}

extern "C" int gptst_mae_fwd(const float* out, const float* src, int lda, const float* mask, float sigma, float mu, float thresh,
                             int rows, int J, float* stats, void* stream) {
    if (!out || !src || !mask || !stats) return GPTST_EARG;
    int nb = (int)(((size_t)rows * J + 255) / 256); if (nb > 64) nb = 64;      // one same-address atomic pair per workgroup: keep them few
    hipLaunchKernelGGL(mae_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, out, src, lda, mask, sigma, mu, thresh, rows, J, stats);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_mae_bwd(const float* out, const float* src, int lda, const float* mask, float sigma, float mu, float thresh,
                             int rows, int J, const float* stats, int normalize, float* dOut, void* stream) {
    if (!out || !src || !mask || !stats || !dOut) return GPTST_EARG;
    int nb = (int)(((size_t)rows * J + 255) / 256); if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(mae_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, out, src, lda, mask, sigma, mu, thresh, rows, J, stats,
                       normalize, dOut);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_kl(const float* prob, const float* c, int rows, int N, int HS, float w, float* dlogit, float* stats, void* stream) {
    if (!prob || !c || !stats) return GPTST_EARG;
    int nb = (rows + 255) / 256;
    hipLaunchKernelGGL(kl_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, prob, c, rows, N, HS, w, dlogit, stats);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_clip_adam_ws_floats(void) { return 2 * GN_NB; }

// ws: gptst_clip_adam_ws_floats() floats of scratch.  stats[3]: extra squared-norm terms (0 unless node-sharded), stats[4] <- the
// total squared gradient norm (after scaling, before clipping).  No atomics: the norm is folded in a fixed order.
// sws (may be NULL) / sws_rows: the step's per-workgroup loss statistics, folded into stats[0..2] (+=) by the first launch — replaces a
// separate gptst_stats_fold when nothing (a gradient all-reduce) has to see the folded statistics in between.
GPTST_INTERNAL const unsigned* gptst_handoff_word_capmfma(void);
GPTST_INTERNAL const unsigned* gptst_handoff_word_hypertem(void);
GPTST_INTERNAL const unsigned* gptst_handoff_word_masksel(void);
static int g_handoff_guard = 1;
// 1 (default): gptst_clip_adam skips the update while a hand-off expiry is on record (see adam_kernel);  0: the update always runs (a lost
// hand-off then shows as NaN weights, the pre-r05 behaviour).  Process-wide.
extern "C" int gptst_set_handoff_guard(int on) { g_handoff_guard = on ? 1 : 0; return GPTST_OK; }

extern "C" int gptst_clip_adam(float* p, const float* g, float* m, float* v, long nA, long nB, const float* hyper, float* stats,
                               float* ws, float* stats_out, const float* sws, int sws_rows, void* stream) {
    if (!p || !g || !m || !v || !hyper || !stats || !ws || (sws && sws_rows <= 0)) return GPTST_EARG;
    if ((nA & 3) || (nB & 3) || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)) return GPTST_EARG;      // 16-byte segments (the flat layout pads every tensor)
    long n = nA + nB;
    int nb = (int)((n / 4 + 255) / 256); if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
    const int nbn = nb > GN_NB ? GN_NB : nb;
    hipLaunchKernelGGL(gradnorm_kernel, dim3(nbn), dim3(256), 0, (hipStream_t)stream, g, nA, nB, hyper, stats, ws, sws, sws_rows);
    static const unsigned* w0 = gptst_handoff_word_capmfma();         // (device addresses: looked up once, outside any graph capture — the steppers warm up first)
    static const unsigned* w1 = gptst_handoff_word_hypertem();
    static const unsigned* w2 = gptst_handoff_word_masksel();
    hipLaunchKernelGGL(adam_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, nA, nB, hyper, stats, (const float*)ws, nbn, stats_out,
                       g_handoff_guard ? w0 : nullptr, g_handoff_guard ? w1 : nullptr, g_handoff_guard ? w2 : nullptr);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
