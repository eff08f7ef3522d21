// Guide classifier MLP_RL: input projection + the node-conditioned layer on the low-rank structure of the input (round 4; the same idea as
// encin.hip).  The classifier starts with Linear(base -> C) on the RAW flow (reference GPTST.py:21-27): for base = 1
//     h0[row,:] = s w1 + b1            (s = the flow of the row, a scalar)
//     h1[row,:] = LReLU(h0 W_n + b_n)  = LReLU(s u_n + c_n),     u_n = w1 W_n,  c_n = b1 W_n + b_n      (two C-vectors per node)
// so the first generated-weight layer is an elementwise pass, and its backward needs only two C-vectors per node:
//     p_n = sum_rows s dPre,  q_n = sum_rows dPre        (dPre = dOut * lrelu'(h1): chain form; rows = the B*T rows of node n)
//     dW_n = w1^T (x) p_n + b1^T (x) q_n,  db_n = q_n,   d w1 = sum_n W_n p_n,  d b1 = sum_n W_n q_n
// forward: replaces gptst_lin_in + gptst_apply(MODE_NODE) (6.6 + 12.2 us, 3 x 16.7 MB) by one 16.7 MB write; backward: replaces
// gptst_apply_wgrad(MODE_NODE) + gptst_rowouter_part (19.3 + 10.7 us) by one pass over dPre.  base = 1, C in {64, 128}.
#include "common.h"

#define GI_CHUNKS 3       // row chunks per node in the forward (510 workgroups at N = 170)

template <int C>
__global__ __launch_bounds__(256) void guide_in_fwd_kernel(const float* __restrict__ src, int lda, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ Wn, const float* __restrict__ bn,
                                                           float* __restrict__ h1, int BT, int N) {
    constexpr int LPR = C / 4, RPP = 256 / LPR, NQ = 256 / C, U = 4;
    __shared__ __attribute__((aligned(16))) float part[NQ * 2 * C];
    __shared__ __attribute__((aligned(16))) float vec[2 * C];
    const int n = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    {
        const int q = tid / C, o = tid % C;
        const float* W = Wn + (size_t)n * C * C;
        float au = 0.f, ac = 0.f;
#pragma unroll 8
        for (int i = q * (C / NQ); i < (q + 1) * (C / NQ); ++i) {
            const float x = W[(size_t)i * C + o];
            au = fmaf(w1[i], x, au);
            ac = fmaf(b1[i], x, ac);
        }
        part[(q * 2 + 0) * C + o] = au;
        part[(q * 2 + 1) * C + o] = ac;
    }
    __syncthreads();
    if (tid < C) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { s0 += part[(q * 2 + 0) * C + tid]; s1 += part[(q * 2 + 1) * C + tid]; }
        vec[tid] = s0; vec[C + tid] = s1 + bn[(size_t)n * C + tid];
    }
    __syncthreads();
    const int slot = tid / LPR, c4 = tid % LPR;
    const float4 u4 = ld4(vec + 4 * c4), k4 = ld4(vec + C + 4 * c4);
    const int per = (BT + GI_CHUNKS - 1) / GI_CHUNKS, r0 = chunk * per, r1 = min(BT, r0 + per);
    for (int rr = r0 + slot; rr < r1; rr += RPP * U) {
        float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = src[((size_t)min(rr + u * RPP, r1 - 1) * N + n) * lda];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = rr + u * RPP;
            if (r < r1) {
                float4 y = f4fma(s[u], u4, k4);
                y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
                st4(h1 + ((size_t)r * N + n) * C + 4 * c4, y);
            }
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void guide_in_bwd_kernel(const float* __restrict__ dPre, const float* __restrict__ src, int lda,
                                                           const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ Wn,
                                                           float* __restrict__ dWb, float* __restrict__ dinp, int BT, int N) {
    constexpr int LPR = C / 4, RPP = 256 / LPR, U = 6, PP = 256 / C;
    __shared__ __attribute__((aligned(16))) float red[RPP * 2 * C];
    __shared__ __attribute__((aligned(16))) float vec[2 * C];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int slot = tid / LPR, c4 = tid % LPR;
    float4 P = f4zero(), Q = f4zero();
    for (int rr = slot; rr < BT; rr += RPP * U) {             // U rows of loads in flight per thread (rows of node n: stride N*C floats)
        float4 d[U];
        float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = (size_t)min(rr + u * RPP, BT - 1) * N + n;
            d[u] = ld4(dPre + r * C + 4 * c4);
            s[u] = src[r * lda];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (rr + u * RPP < BT) { P = f4fma(s[u], d[u], P); Q = f4add(Q, d[u]); }
    }
    st4(red + (slot * 2 + 0) * C + 4 * c4, P); st4(red + (slot * 2 + 1) * C + 4 * c4, Q);
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) {                 // fold the slots in order
        float s = 0.f;
#pragma unroll 4
        for (int sl = 0; sl < RPP; ++sl) s += red[(sl * 2 + i / C) * C + i % C];
        vec[i] = s;
    }
    __syncthreads();
    float* row = dWb + (size_t)n * (C * C + C);
    for (int f = tid; f < C * C / 4; f += 256) {             // dW_n = w1^T (x) p + b1^T (x) q
        const int i = f / LPR, o4 = f % LPR;
        const float wi = w1[i], bb = b1[i];
        const float4 p4 = ld4(vec + 4 * o4), q4 = ld4(vec + C + 4 * o4);
        st4(row + (size_t)i * C + 4 * o4, make_float4(fmaf(wi, p4.x, bb * q4.x), fmaf(wi, p4.y, bb * q4.y), fmaf(wi, p4.z, bb * q4.z), fmaf(wi, p4.w, bb * q4.w)));
    }
    if (tid < C) row[C * C + tid] = vec[C + tid];            // db_n
    {   // [W_n p | W_n q]: thread (input channel i, part of the output channels)
        const int i = tid / PP, pq = tid % PP;
        const float* Wr = Wn + (size_t)n * C * C + (size_t)i * C + pq * (C / PP);
        float r1 = 0.f, r2 = 0.f;
#pragma unroll
        for (int k = 0; k < C / PP / 4; ++k) {
            const float4 x = ld4(Wr + 4 * k);
            r1 += f4dot(x, ld4(vec + pq * (C / PP) + 4 * k));
            r2 += f4dot(x, ld4(vec + C + pq * (C / PP) + 4 * k));
        }
        r1 = group_sum<PP>(r1); r2 = group_sum<PP>(r2);
        if (pq == 0) { dinp[(size_t)n * 2 * C + i] = r1; dinp[(size_t)n * 2 * C + C + i] = r2; }
    }
}

// h1 (BT*N, C) = LReLU((s w1 + b1) W_n + b_n):  src rows (BT*N, lda) with the flow in column 0, w1 = MLP_RL.ln1.weight (C,1), b1 = its bias,
// Wn (N,C,C) / bn (N,C) = the node-conditioned weights generated from neb4mask
extern "C" int gptst_guide_in_fwd(const float* src, int lda, const float* w1, const float* b1, const float* Wn, const float* bn, float* h1,
                                  int BT, int N, int C, void* stream) {
    if (!src || !w1 || !b1 || !Wn || !bn || !h1 || BT <= 0 || N <= 0 || lda <= 0) return GPTST_EARG;
    if (C == 64) hipLaunchKernelGGL((guide_in_fwd_kernel<64>), dim3(N, GI_CHUNKS), dim3(256), 0, (hipStream_t)stream, src, lda, w1, b1, Wn, bn, h1, BT, N);
    else if (C == 128) hipLaunchKernelGGL((guide_in_fwd_kernel<128>), dim3(N, GI_CHUNKS), dim3(256), 0, (hipStream_t)stream, src, lda, w1, b1, Wn, bn, h1, BT, N);
    else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// dPre (BT*N, C) = dOut * lrelu'(h1) (chain form) -> dWb (N, C*C + C) rows [dW_n | db_n], dinp (N, 2C) partials of [d ln1.weight | d ln1.bias]
extern "C" int gptst_guide_in_bwd(const float* dPre, const float* src, int lda, const float* w1, const float* b1, const float* Wn, float* dWb,
                                  float* dinp, int BT, int N, int C, void* stream) {
    if (!dPre || !src || !w1 || !b1 || !Wn || !dWb || !dinp || BT <= 0 || N <= 0 || lda <= 0) return GPTST_EARG;
    if (C == 64) hipLaunchKernelGGL((guide_in_bwd_kernel<64>), dim3(N), dim3(256), 0, (hipStream_t)stream, dPre, src, lda, w1, b1, Wn, dWb, dinp, BT, N);
    else if (C == 128) hipLaunchKernelGGL((guide_in_bwd_kernel<128>), dim3(N), dim3(256), 0, (hipStream_t)stream, dPre, src, lda, w1, b1, Wn, dWb, dinp, BT, N);
    else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
