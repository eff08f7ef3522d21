// Guide classifier MLP_RL: input projection + the node-conditioned layer on the low-rank structure of the input (round 4; the same idea as
// encin.hip).  The classifier starts with Linear(base -> C) on the RAW flow (reference GPTST.py:21-27): for base = 1
//     h0[row,:] = s w1 + b1            (s = the flow of the row, a scalar)
//     h1[row,:] = LReLU(h0 W_n + b_n)  = LReLU(s u_n + c_n),     u_n = w1 W_n,  c_n = b1 W_n + b_n      (two C-vectors per node)
// so the first generated-weight layer is an elementwise pass, and its backward needs only two C-vectors per node:
//     p_n = sum_rows s dPre,  q_n = sum_rows dPre        (dPre = dOut * lrelu'(h1): chain form; rows = the B*T rows of node n)
//     dW_n = w1^T (x) p_n + b1^T (x) q_n,  db_n = q_n,   d w1 = sum_n W_n p_n,  d b1 = sum_n W_n q_n
// forward: replaces gptst_lin_in + gptst_apply(MODE_NODE) (6.6 + 12.2 us, 3 x 16.7 MB) by one 16.7 MB write; backward: replaces
// gptst_apply_wgrad(MODE_NODE) + gptst_rowouter_part (19.3 + 10.7 us) by one pass over dPre.  base = 1, C in {64, 128}.
#include "mfma_tile.h"

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

// ---- r06: the whole classifier forward in TWO launches ---------------------------------------------------------------------------------------------
// guide_in_fwd (8.8 us) -> apply64 TIME layer (14.6) -> rowdot_mfma (8.4) read and write h1 / h2 between them (4 x 16.7 MB) and pay three launch
// ramps on the critical path in front of the mask.  Everything behind the node vectors u_n, c_n is local to a row, and the time-conditioned layer is
// grouped by (b,t) exactly like the class head walks its rows, so:
//   guide_uc_kernel      one workgroup per node: u_n = w1 W_n, c_n = b1 W_n + b_n (the prologue of guide_in_fwd_kernel, same summation order)
//   guide_head_fwd       (b,t) groups as apply64 (TIME): A fragments h1 = LReLU(s u_n + c_n) built in registers (and written out for the backward),
//                        h2 = LReLU(h1 W_bt + b_bt) on the 64 register-operand MFMAs of apply64, then — through a wave-private LDS tile into the
//                        A-operand layout — the 16 MFMAs of the class head, the softmax and the first-maximum label of rowdot_mfma.
// Same arithmetic in the same order as the three kernels it replaces: h1, h2, prob and label are BIT-IDENTICAL (tests/test_gpu_kernels.py).
template <int C>
__global__ __launch_bounds__(256) void guide_uc_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ Wn,
                                                       const float* __restrict__ bn, float* __restrict__ uc) {
    constexpr int NQ = 256 / C;
    __shared__ __attribute__((aligned(16))) float part[NQ * 2 * C];
    const int n = blockIdx.x, tid = threadIdx.x;
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
    __syncthreads();
    if (tid < C) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) { s0 += part[(qq * 2 + 0) * C + tid]; s1 += part[(qq * 2 + 1) * C + tid]; }
        uc[(size_t)n * 2 * C + tid] = s0;
        uc[(size_t)n * 2 * C + C + tid] = s1 + bn[(size_t)n * C + tid];
    }
}

#define GH_P 68
extern thread_local int g_apply_tpw;
__global__ __launch_bounds__(256, 2) void guide_head_fwd_kernel(const float* __restrict__ src, int lda, const float* __restrict__ uc,
                                                                const float* __restrict__ Wbt, const float* __restrict__ bbt,
                                                                const float* __restrict__ W3, const float* __restrict__ b3,
                                                                float* __restrict__ h1, float* __restrict__ h2, float* __restrict__ prob,
                                                                int* __restrict__ label, int N, int J, int tiles_per_wave) {
    constexpr int C = 64;
    __shared__ __attribute__((aligned(16))) float Wl[C * C];
    __shared__ __attribute__((aligned(16))) float tl[4][16 * GH_P];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int g = blockIdx.x;
    const int ntiles = (N + 15) / 16;
    const int t0 = (blockIdx.y * 4 + wave) * tiles_per_wave, t1 = min(ntiles, t0 + tiles_per_wave);
    const float4 bias4 = ld4(bbt + (size_t)g * C + 4 * j);
    float4 bw[4];                                                 // class head: B = W3[class j][16q + 4kk ..] (zero for j >= J)
#pragma unroll
    for (int q = 0; q < 4; ++q) bw[q] = j < J ? ld4(W3 + (size_t)j * C + 16 * q + 4 * kk) : f4zero();
    const float bj = j < J ? b3[j] : 0.f;
    float sn;
    float4 un[4], cn[4];
    auto fetch = [&](int t) {                                     // the row's flow and its node's two vectors (L2-resident: N x 2C floats)
        const int m = min(t * 16 + j, N - 1);
        sn = src[((size_t)g * N + m) * lda];
#pragma unroll
        for (int q = 0; q < 4; ++q) { un[q] = ld4(uc + (size_t)m * 2 * C + 16 * q + 4 * kk); cn[q] = ld4(uc + (size_t)m * 2 * C + C + 16 * q + 4 * kk); }
    };
    if (t0 < t1) fetch(t0);
    load_w_lds<C, 256>(Wl, Wbt + (size_t)g * C * C, 0, threadIdx.x);
    __syncthreads();
    float4 bv[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[q][e] = ld4(Wl + (16 * q + 4 * kk + e) * C + 4 * j);
    float* tile = tl[wave];
    for (int t = t0; t < t1; ++t) {
        float4 a[4];
        const int mrow = t * 16 + j;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 y = f4fma(sn, un[q], cn[q]);
            y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
            if (mrow < N) st4(h1 + ((size_t)g * N + mrow) * C + 16 * q + 4 * kk, y); else y = f4zero();
            a[q] = y;
        }
        if (t + 1 < t1) fetch(t + 1);
        SB();
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].w, acc[3], 0, 0, 0);
            }
        }
        SB();
        // h2 rows kk*4 + r, channels 4j .. 4j+3: out to global, and into the wave's LDS tile for the class head's operand layout
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = t * 16 + kk * 4 + r;
            float4 y = f4add(make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), bias4);
            y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
            if (m < N) st4(h2 + ((size_t)g * N + m) * C + 4 * j, y);
            st4(tile + (kk * 4 + r) * GH_P + 4 * j, y);
        }
        f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {                             // (wave-private tile: the wave's own writes are visible to it in program order)
            const float4 x = ld4(tile + j * GH_P + 16 * q + 4 * kk);
            z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, bw[q].x, z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, bw[q].y, z1, 0, 0, 0);
            z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, bw[q].z, z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, bw[q].w, z1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = t * 16 + kk * 4 + r;
            const size_t i = (size_t)g * N + m;
            float v = z0[r] + z1[r] + bj;
            const float mx = group_max<16>(j < J ? v : -3.0e38f);
            const float e = j < J ? expf(v - mx) : 0.f;
            v = e / group_sum<16>(e);
            if (m < N && j < J) prob[i * J + j] = v;
            const float vv = j < J ? v : -3.0e38f;
            const float gm = group_max<16>(vv);
            const int first = (int)(-group_max<16>(vv == gm ? -(float)j : -3.0e38f));
            if (m < N && j == 0) label[i] = first;
        }
    }
}

extern "C" int gptst_guide_uc_floats(int N, int C) { return N * 2 * C; }

// src rows (B*T*N, lda) with the flow in column 0; w1 / b1 = MLP_RL.ln1; Wn (N,C,C) / bn (N,C), Wbt (B*T,C,C) / bbt (B*T,C) the generated node- and
// time-conditioned parameters; W3 (J,C) / b3 (J) = MLP_RL.ln3, J <= 16.  uc: gptst_guide_uc_floats scratch.  -> h1, h2 (B*T*N, C), prob (B*T*N, J) =
// softmax of the logits, label (B*T*N) int32 = first maximum.  C = 64 (else GPTST_ESHAPE: the three launches).
extern "C" int gptst_guide_head_fwd(const float* src, int lda, const float* w1, const float* b1, const float* Wn, const float* bn, const float* Wbt,
                                    const float* bbt, const float* W3, const float* b3, float* uc, float* h1, float* h2, float* prob, int* label,
                                    int BT, int N, int C, int J, void* stream) {
    if (!src || !w1 || !b1 || !Wn || !bn || !Wbt || !bbt || !W3 || !b3 || !uc || !h1 || !h2 || !prob || !label || BT <= 0 || N <= 0 || lda <= 0 || J <= 0)
        return GPTST_EARG;
    if (C != 64 || J > 16) return GPTST_ESHAPE;
    hipLaunchKernelGGL((guide_uc_kernel<64>), dim3(N), dim3(256), 0, (hipStream_t)stream, w1, b1, Wn, bn, uc);
    const int ntiles = (N + 15) / 16;
    // ONE tile per wave (1152 workgroups at the bench shape): the tile is a chain — operands, 64 MFMAs, LDS turn, 16 MFMAs, softmax — that a second
    // tile in the same wave only lengthens (stand-alone incl. guide_uc: 25.5 us at one tile per wave, 28.0 at two; the three launches: 30.8).
    // Ablation at one tile per wave (r06, us): full 26.3, without the h1 store 21.3, without the class head 19.6, without the h2 store 23.9.
    int tpw = 1;
    if (g_apply_tpw > 0) tpw = g_apply_tpw;                      // gptst_tune(4, n)
    if (tpw < 1) tpw = 1;
    const int gy = (ntiles + 4 * tpw - 1) / (4 * tpw);
    hipLaunchKernelGGL(guide_head_fwd_kernel, dim3(BT, gy), dim3(256), 0, (hipStream_t)stream, src, lda, uc, Wbt, bbt, W3, b3, h1, h2, prob, label, N, J, tpw);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
