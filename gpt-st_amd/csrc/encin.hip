// Encoder input projection + the encoder's FIRST hyperTem layer on the low-rank structure of the input (round 4).
//
// The encoder embeds the masked flow with Linear(base -> C) (reference GPTST.py:415-418): for base = 1 every row of the first activation is
//     x0[b,t,n,:] = m * w + bi,      m = mask ? flow : scaler_zeros   (a scalar),  w = dim_in_flow.weight[:,0],  bi = dim_in_flow.bias
// — a rank-1 term plus a constant.  hyperTem1 (:154-163) is linear up to its LeakyReLU, so it never needs the (B,T,N,C) tensor x0 nor a GEMM:
//     R_t[n,:]  = sum_u G_n[t,u] x0_u[n,:]            = alpha w + beta bi          alpha = sum_u G_n[t,u] m_u[n],  beta = sum_u G_n[t,u]
//     out_t[n,:] = LReLU(R_t W_bt + b_bt + x0_t)       = LReLU(alpha (w W_bt) + beta (bi W_bt) + b_bt + m w + bi)
// i.e. two scalars per (b,t,n) and two C-vectors per (b,t).  The backward collapses the same way (dPre = dOut * lrelu'(out) comes in, chain form):
//     dW_bt = sum_n R^T dPre     = w^T (x) A + bi^T (x) Bv        A = sum_n alpha dPre,  Bv = sum_n beta dPre      (rank 2)
//     db_bt = Cv = sum_n dPre
//     dG_n[t,u] = sum_c dR_t[n,c] x0_u[n,c] = m_u (dPre . (w W_bt)) + dPre . (bi W_bt)                              (dR = dPre W_bt^T)
//     d w  = sum_rows m dX0 = sum_bt [ sum_n m_t dPre + A W_bt^T ],   d bi = sum_bt [ Cv + Bv W_bt^T ]             (dX0 = dPre + G^T dR)
// One workgroup per (b,t) in both directions, no MFMA, no x0 / R / dX0 round trips: lin_in + hypertem_fwd (6.5 + 22.5 us, 3 x 16.7 MB written)
// become one 16.7 MB write; hypertem_bwd_wgrad + rowouter_part (30 + 11 us, ~130 MB) become one pass over dPre.
// Serves base = 1, C = 64 or 128, T = 12 (GPTST_ESHAPE otherwise: lin_in + the generic layer).
#include "common.h"

#define EI_T 12
GPTST_STAMP_TABLES(encin)

template <int C>
__global__ __launch_bounds__(256) void encin_fwd_kernel(const float* __restrict__ src, int lda, const float* __restrict__ mask, float fill,
                                                        const float* __restrict__ w, const float* __restrict__ bi, const float* __restrict__ G,
                                                        const float* __restrict__ Wbt, const float* __restrict__ bbt, float* __restrict__ out,
                                                        float* __restrict__ ab, float* __restrict__ wv, int N) {
    constexpr int LPR = C / 4, RPP = 256 / LPR, NQ = 256 / C;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* part = smem;                    // [NQ][2][C]
    float* vec = part + NQ * 2 * C;        // [3][C]   w W_bt | bi W_bt | b_bt + bi
    float* al = vec + 3 * C;               // [N] alpha, [N] beta, [N] m_t
    float* be = al + N;
    float* mm = be + N;
    const int bt = blockIdx.x, b = bt / EI_T, t = bt % EI_T, tid = threadIdx.x;
    GPTST_WG_BEGIN(); GPTST_STAMP(0);
    {   // w W_bt and bi W_bt: thread (quarter q of the input channels, output channel o)
        const int q = tid / C, o = tid % C;
        const float* W = Wbt + (size_t)bt * C * C;
        float aw = 0.f, abv = 0.f;
#pragma unroll 8
        for (int i = q * (C / NQ); i < (q + 1) * (C / NQ); ++i) {
            const float x = W[(size_t)i * C + o];
            aw = fmaf(w[i], x, aw);
            abv = fmaf(bi[i], x, abv);
        }
        part[(q * 2 + 0) * C + o] = aw;
        part[(q * 2 + 1) * C + o] = abv;
    }
    GPTST_STAMP(1);
    for (int n = tid; n < N; n += 256) {   // alpha, beta, m_t of node n
        float g[EI_T];
        const float* gr = G + (size_t)n * EI_T * EI_T + t * EI_T;
#pragma unroll
        for (int k = 0; k < EI_T / 4; ++k) { const float4 v = ld4(gr + 4 * k); g[4 * k] = v.x; g[4 * k + 1] = v.y; g[4 * k + 2] = v.z; g[4 * k + 3] = v.w; }
        float sv[EI_T], mk[EI_T];
#pragma unroll
        for (int u = 0; u < EI_T; ++u) {
            const size_t r = ((size_t)b * EI_T + u) * N + n;
            sv[u] = src[r * lda];
            mk[u] = mask ? mask[r] : 1.f;
        }
        float a_ = 0.f, b_ = 0.f, mt = 0.f;
#pragma unroll
        for (int u = 0; u < EI_T; ++u) {
            const float m = mk[u] != 0.f ? sv[u] : fill;
            a_ = fmaf(g[u], m, a_);
            b_ += g[u];
            if (u == t) mt = m;
        }
        al[n] = a_; be[n] = b_; mm[n] = mt;
        ab[((size_t)bt * N + n) * 2] = a_;
        ab[((size_t)bt * N + n) * 2 + 1] = b_;
    }
    __syncthreads();
    GPTST_STAMP(2);
    if (tid < C) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { s0 += part[(q * 2 + 0) * C + tid]; s1 += part[(q * 2 + 1) * C + tid]; }
        vec[tid] = s0; vec[C + tid] = s1; vec[2 * C + tid] = bbt[(size_t)bt * C + tid] + bi[tid];
        wv[(size_t)bt * 2 * C + tid] = s0;
        wv[(size_t)bt * 2 * C + C + tid] = s1;
    }
    __syncthreads();
    const int slot = tid / LPR, c4 = tid % LPR;
    const float4 wW = ld4(vec + 4 * c4), bW = ld4(vec + C + 4 * c4), cst = ld4(vec + 2 * C + 4 * c4), w4 = ld4(w + 4 * c4);
    for (int n = slot; n < N; n += RPP) {
        const float a_ = al[n], b_ = be[n], m = mm[n];
        float4 y = f4fma(a_, wW, f4fma(b_, bW, f4fma(m, w4, cst)));
        y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
        st4(out + ((size_t)bt * N + n) * C + 4 * c4, y);
    }
    GPTST_STAMP(3); GPTST_WG_END();
}

template <int C>
__global__ __launch_bounds__(256) void encin_bwd_kernel(const float* __restrict__ dPre, const float* __restrict__ src, int lda,
                                                        const float* __restrict__ mask, float fill, const float* __restrict__ w,
                                                        const float* __restrict__ bi, const float* __restrict__ Wbt, const float* __restrict__ ab,
                                                        const float* __restrict__ wv, float* __restrict__ dWb, float* __restrict__ dG,
                                                        float* __restrict__ dinp, int N, int B) {
    constexpr int LPR = C / 4, RPP = 256 / LPR, U = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem;                     // [RPP][4][C]  per-slot partials of A, Bv, Cv, Mv
    float* vec = red + RPP * 4 * C;        // [6][C]       A | Bv | Cv | Mv | w W_bt | bi W_bt
    float* as_ = vec + 6 * C;              // [N] dPre . (w W_bt),  [N] dPre . (bi W_bt)
    float* cs_ = as_ + N;
    const int bt = blockIdx.x, b = bt / EI_T, t = bt % EI_T, tid = threadIdx.x;
    GPTST_WG_BEGIN(); GPTST_STAMP(0);
    if (tid < 2 * C) vec[4 * C + tid] = wv[(size_t)bt * 2 * C + tid];
    __syncthreads();
    const int slot = tid / LPR, c4 = tid % LPR;
    const float4 wW = ld4(vec + 4 * C + 4 * c4), bW = ld4(vec + 5 * C + 4 * c4);
    float4 A = f4zero(), Bv = f4zero(), Cv = f4zero(), Mv = f4zero();
    for (int n0 = slot; n0 < N; n0 += RPP * U) {           // U rows of loads in flight per thread
        float4 d[U];
        float a_[U], b_[U], m_[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = min(n0 + u * RPP, N - 1);
            const size_t r = (size_t)bt * N + n;
            d[u] = ld4(dPre + r * C + 4 * c4);
            a_[u] = ab[r * 2]; b_[u] = ab[r * 2 + 1];
            m_[u] = (mask ? mask[r] : 1.f) != 0.f ? src[r * lda] : fill;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = n0 + u * RPP;
            const bool ok = n < N;                           // (uniform over the LPR lanes of a row)
            const float4 dd = ok ? d[u] : f4zero();
            A = f4fma(a_[u], dd, A); Bv = f4fma(b_[u], dd, Bv); Cv = f4add(Cv, dd); Mv = f4fma(m_[u], dd, Mv);
            const float pa = group_sum<LPR>(f4dot(dd, wW)), pc = group_sum<LPR>(f4dot(dd, bW));
            if (ok && c4 == 0) { as_[n] = pa; cs_[n] = pc; }
        }
    }
    GPTST_STAMP(1);
    st4(red + (slot * 4 + 0) * C + 4 * c4, A); st4(red + (slot * 4 + 1) * C + 4 * c4, Bv);
    st4(red + (slot * 4 + 2) * C + 4 * c4, Cv); st4(red + (slot * 4 + 3) * C + 4 * c4, Mv);
    __syncthreads();
    for (int i = tid; i < 4 * C; i += 256) {                 // fold the slots in order
        float s = 0.f;
#pragma unroll 4
        for (int sl = 0; sl < RPP; ++sl) s += red[(sl * 4 + i / C) * C + i % C];
        vec[i] = s;
    }
    __syncthreads();
    GPTST_STAMP(2);
    float* row = dWb + (size_t)bt * (C * C + C);
    for (int f = tid; f < C * C / 4; f += 256) {             // dW_bt = w^T (x) A + bi^T (x) Bv
        const int i = f / LPR, o4 = f % LPR;
        const float wi = w[i], bb = bi[i];
        const float4 a4 = ld4(vec + 4 * o4), b4 = ld4(vec + C + 4 * o4);
        st4(row + (size_t)i * C + 4 * o4, make_float4(fmaf(wi, a4.x, bb * b4.x), fmaf(wi, a4.y, bb * b4.y), fmaf(wi, a4.z, bb * b4.z), fmaf(wi, a4.w, bb * b4.w)));
    }
    if (tid < C) row[C * C + tid] = vec[2 * C + tid];        // db_bt
    {   // d(dim_in_flow): [sum_n m_t dPre + A W_bt^T | Cv + Bv W_bt^T];  thread (input channel i, quarter of the output channels)
        constexpr int PP = 256 / C;                          // parts per row: 4 (C = 64) / 2 (C = 128)
        const int i = tid / PP, pq = tid % PP;
        const float* Wr = Wbt + (size_t)bt * C * C + (size_t)i * C + pq * (C / PP);
        float p1 = 0.f, p2 = 0.f;
#pragma unroll
        for (int k = 0; k < C / PP / 4; ++k) {
            const float4 x = ld4(Wr + 4 * k);
            p1 += f4dot(x, ld4(vec + pq * (C / PP) + 4 * k));
            p2 += f4dot(x, ld4(vec + C + pq * (C / PP) + 4 * k));
        }
        p1 = group_sum<PP>(p1); p2 = group_sum<PP>(p2);
        if (pq == 0) {
            dinp[(size_t)bt * 2 * C + i] = vec[3 * C + i] + p1;
            dinp[(size_t)bt * 2 * C + C + i] = vec[2 * C + i] + p2;
        }
    }
    GPTST_STAMP(3);
    for (int n = tid; n < N; n += 256) {                     // dG_n[t, u] = m_u a + c: the sample's partial, row t
        const float a = as_[n], c = cs_[n];
        float v[EI_T];
#pragma unroll
        for (int u = 0; u < EI_T; ++u) {
            const size_t r = ((size_t)b * EI_T + u) * N + n;
            const float m = (mask ? mask[r] : 1.f) != 0.f ? src[r * lda] : fill;
            v[u] = fmaf(m, a, c);
        }
        float* o = dG + ((size_t)b * N + n) * EI_T * EI_T + t * EI_T;
#pragma unroll
        for (int k = 0; k < EI_T / 4; ++k) st4(o + 4 * k, make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]));
    }
    GPTST_STAMP(4); GPTST_WG_END();
}

// out (B,T,N,C) = LReLU(hyperTem1(Linear(base = 1 -> C)(masked flow)));  ab (B*T*N, 2): (alpha, beta) per row;  wv (B*T, 2C): (w W_bt | bi W_bt)
// — both kept for the backward.  src: (B*T*N, lda) rows whose column 0 is the flow;  mask (B*T*N) fp32 1 = visible, or NULL;  w = dim_in_flow.weight
// (C,1), bi = its bias;  G (N,T,T), Wbt (B*T,C,C), bbt (B*T,C).
extern "C" int gptst_encin_ht1_fwd(const float* src, int lda, const float* mask, float fill, const float* w, const float* bi, const float* G,
                                   const float* Wbt, const float* bbt, float* out, float* ab, float* wv, int B, int T, int N, int C, void* stream) {
    if (!src || !w || !bi || !G || !Wbt || !bbt || !out || !ab || !wv || B <= 0 || N <= 0 || lda <= 0) return GPTST_EARG;
    if (T != EI_T || (C != 64 && C != 128)) return GPTST_ESHAPE;
    const size_t smem = ((size_t)(256 / C) * 2 * C + 3 * C + 3 * (size_t)N) * sizeof(float);
    if (smem > 64 * 1024) return GPTST_ESHAPE;
    if (C == 64) hipLaunchKernelGGL((encin_fwd_kernel<64>), dim3(B * T), dim3(256), smem, (hipStream_t)stream, src, lda, mask, fill, w, bi, G, Wbt, bbt, out, ab, wv, N);
    else hipLaunchKernelGGL((encin_fwd_kernel<128>), dim3(B * T), dim3(256), smem, (hipStream_t)stream, src, lda, mask, fill, w, bi, G, Wbt, bbt, out, ab, wv, N);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// backward of the above given dPre = dOut * lrelu'(out) (chain form):  dWb (B*T, C*C + C) rows [dW_bt | db_bt] (as gptst_hypertem_bwd_wgrad, one split),
// dG (B, N, T, T) per-sample partials of the temporal-graph gradient, dinp (B*T, 2C) partials of [d dim_in_flow.weight | d dim_in_flow.bias]
// (the caller sums the rows).  No input gradient exists (the input is data).
extern "C" int gptst_encin_ht1_bwd(const float* dPre, const float* src, int lda, const float* mask, float fill, const float* w, const float* bi,
                                   const float* Wbt, const float* ab, const float* wv, float* dWb, float* dG, float* dinp, int B, int T, int N,
                                   int C, void* stream) {
    if (!dPre || !src || !w || !bi || !Wbt || !ab || !wv || !dWb || !dG || !dinp || B <= 0 || N <= 0 || lda <= 0) return GPTST_EARG;
    if (T != EI_T || (C != 64 && C != 128)) return GPTST_ESHAPE;
    const size_t smem = ((size_t)(256 / (C / 4)) * 4 * C + 6 * C + 2 * (size_t)N) * sizeof(float);
    if (smem > 64 * 1024) return GPTST_ESHAPE;
    if (C == 64) hipLaunchKernelGGL((encin_bwd_kernel<64>), dim3(B * T), dim3(256), smem, (hipStream_t)stream, dPre, src, lda, mask, fill, w, bi, Wbt, ab, wv, dWb, dG, dinp, N, B);
    else hipLaunchKernelGGL((encin_bwd_kernel<128>), dim3(B * T), dim3(256), smem, (hipStream_t)stream, dPre, src, lda, mask, fill, w, bi, Wbt, ab, wv, dWb, dG, dinp, N, B);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
