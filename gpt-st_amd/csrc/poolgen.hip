// Embedding-conditioned parameter generation:  out[r, :] = sum_k emb[r,k] * pool[k, :]   (K = embed dim <= 32)
//
// This is the reference's  einsum('btd,dio->btio') / einsum('nd,dio->nio') / matmul(emb, bias_pool)
// (GPTST.py:24-25,29-30,137-138,160-161), einsum('btd,dhn->bthn') (:104), einsum('bd,dhk->bhk') (:129) and
// einsum('nk,kht->nht') (:156): a skinny GEMM (R rows <= a few hundred, K <= 16, up to C*C columns).  It is HBM/L2
// bound on the output, so it runs on the VALU with coalesced float4 columns; the three gradient reductions are here too.
// Every kernel takes an optional second (pool2, cols2) problem sharing emb (weights_pool + bias_pool in one launch).
#include "common.h"

#define PG_MAXK 16
#define PG_ROWS 8

// V = 4: float4 columns (cols % 4 == 0, rows 16-byte aligned);  V = 1: scalar columns (e.g. HS*N = 2070 for METR_LA)
template <int V> __device__ __forceinline__ float4 ldv(const float* p) { return ld4(p); }
template <> __device__ __forceinline__ float4 ldv<1>(const float* p) { return make_float4(*p, 0.f, 0.f, 0.f); }
template <int V> __device__ __forceinline__ void stv(float* p, float4 v) { st4(p, v); }
template <> __device__ __forceinline__ void stv<1>(float* p, float4 v) { *p = v.x; }

// grid: (ceil(cols/4/256) + ceil(cols2/4/256), ceil(R/PG_ROWS))
template <int V>
__global__ __launch_bounds__(256) void poolgen_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ pool,
                                                          float* __restrict__ out, int cols, const float* __restrict__ pool2,
                                                          float* __restrict__ out2, int cols2, int R, int K, int nblk1) {
    int bx = blockIdx.x;
    if (bx >= nblk1) { bx -= nblk1; pool = pool2; out = out2; cols = cols2; }
    const int c4 = bx * 256 + threadIdx.x;
    if (V * c4 >= cols) return;
    float4 p[PG_MAXK];
#pragma unroll
    for (int k = 0; k < PG_MAXK; ++k) p[k] = (k < K) ? ldv<V>(pool + (size_t)k * cols + V * c4) : f4zero();
    const int r0 = blockIdx.y * PG_ROWS;
#pragma unroll 1
    for (int r = r0; r < min(R, r0 + PG_ROWS); ++r) {
        float4 acc = f4zero();
#pragma unroll
        for (int k = 0; k < PG_MAXK; ++k)
            if (k < K) acc = f4fma(emb[(size_t)r * K + k], p[k], acc);
        stv<V>(out + (size_t)r * cols + V * c4, acc);
    }
}

// dpool[k, c] += sum_rr emb[rr % R, k] * dW[rr, c]   (rr < R*nsplit; splits come from wgrad's K-splitting)
// block = 64 column-lanes x 4 row-phases; grid: (ceil(cols/4/64)+..., row chunks)
template <int V>
__global__ __launch_bounds__(256) void poolgen_bwd_pool_kernel(const float* __restrict__ emb, const float* __restrict__ dW,
                                                               float* __restrict__ dpool, int cols, const float* __restrict__ dW2,
                                                               float* __restrict__ dpool2, int cols2, int R, int RR, int K,
                                                               int nblk1, int rows_per_block) {
    __shared__ float4 red[3][PG_MAXK][64];
    int bx = blockIdx.x;
    if (bx >= nblk1) { bx -= nblk1; dW = dW2; dpool = dpool2; cols = cols2; RR = R; }   // splits apply to problem 1 only
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c4 = bx * 64 + lane;
    const bool ok = V * c4 < cols;
    float4 acc[PG_MAXK];
#pragma unroll
    for (int k = 0; k < PG_MAXK; ++k) acc[k] = f4zero();
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(RR, r0 + rows_per_block);
    if (ok) {
        for (int rr = r0 + ph; rr < r1; rr += 4) {
            const float4 v = ldv<V>(dW + (size_t)rr * cols + V * c4);
            const float* e = emb + (size_t)(rr % R) * K;
#pragma unroll
            for (int k = 0; k < PG_MAXK; ++k)
                if (k < K) acc[k] = f4fma(e[k], v, acc[k]);
        }
    }
    if (ph > 0) {
#pragma unroll
        for (int k = 0; k < PG_MAXK; ++k)
            if (k < K) red[ph - 1][k][lane] = acc[k];
    }
    __syncthreads();
    if (ph == 0 && ok) {
#pragma unroll
        for (int k = 0; k < PG_MAXK; ++k)
            if (k < K) {
                float4 s = f4add(f4add(acc[k], red[0][k][lane]), f4add(red[1][k][lane], red[2][k][lane]));
                float* o = dpool + (size_t)k * cols + V * c4;
                atomicAdd(o + 0, s.x);
                if (V == 4) { atomicAdd(o + 1, s.y); atomicAdd(o + 2, s.z); atomicAdd(o + 3, s.w); }
            }
    }
}

// demb[r, k] += sum_split sum_c dW[split*R + r, c] * pool[k, c]  (+ second problem).  one block per row r.
template <int V>
__global__ __launch_bounds__(256) void poolgen_bwd_emb_kernel(const float* __restrict__ dW, const float* __restrict__ pool, int cols,
                                                              const float* __restrict__ dW2, const float* __restrict__ pool2,
                                                              int cols2, float* __restrict__ demb, int R, int nsplit, int K) {
    __shared__ float red[4][PG_MAXK];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc[PG_MAXK];
#pragma unroll
    for (int k = 0; k < PG_MAXK; ++k) acc[k] = 0.f;
    for (int pb = 0; pb < 2; ++pb) {
        const float* w = pb ? dW2 : dW;
        const float* p = pb ? pool2 : pool;
        const int cc = pb ? cols2 : cols;
        const int ns = pb ? 1 : nsplit;              // splits apply to problem 1 only
        if (w == nullptr) continue;
        for (int c4 = tid; V * c4 < cc; c4 += 256) {
            float4 v = f4zero();
            for (int s = 0; s < ns; ++s) v = f4add(v, ldv<V>(w + ((size_t)s * R + r) * cc + V * c4));
#pragma unroll
            for (int k = 0; k < PG_MAXK; ++k)
                if (k < K) acc[k] += f4dot(v, ldv<V>(p + (size_t)k * cc + V * c4));
        }
    }
#pragma unroll
    for (int k = 0; k < PG_MAXK; ++k) {
        const float s = group_sum<64>(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (tid < K) demb[(size_t)r * K + tid] += red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

extern "C" int gptst_poolgen_fwd(const float* emb, const float* pool, float* out, int cols, const float* pool2, float* out2,
                                 int cols2, int R, int K, void* stream) {
    if (!emb || !pool || !out || K > PG_MAXK || K <= 0) return GPTST_EARG;
    if (!pool2) cols2 = 0;
    const int V = ((cols & 3) || (cols2 & 3)) ? 1 : 4;
    const int nb1 = ((cols + V - 1) / V + 255) / 256, nb2 = pool2 ? ((cols2 + V - 1) / V + 255) / 256 : 0;
    dim3 grid(nb1 + nb2, (R + PG_ROWS - 1) / PG_ROWS);
    if (V == 4) hipLaunchKernelGGL(poolgen_fwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, emb, pool, out, cols, pool2, out2, cols2, R, K, nb1);
    else hipLaunchKernelGGL(poolgen_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, emb, pool, out, cols, pool2, out2, cols2, R, K, nb1);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_poolgen_bwd_pool(const float* emb, const float* dW, float* dpool, int cols, const float* dW2, float* dpool2,
                                      int cols2, int R, int nsplit, int K, void* stream) {
    if (!emb || !dW || !dpool || K > PG_MAXK || K <= 0) return GPTST_EARG;
    if (!dW2) cols2 = 0;
    const int V = ((cols & 3) || (cols2 & 3)) ? 1 : 4;
    const int RR = R * nsplit;
    const int nb1 = ((cols + V - 1) / V + 63) / 64, nb2 = dW2 ? ((cols2 + V - 1) / V + 63) / 64 : 0;
    int chunks = 1024 / (nb1 + nb2); if (chunks < 1) chunks = 1;
    int rpb = (RR + chunks - 1) / chunks; if (rpb < 8) rpb = 8;
    dim3 grid(nb1 + nb2, (RR + rpb - 1) / rpb);
    if (V == 4) hipLaunchKernelGGL(poolgen_bwd_pool_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, emb, dW, dpool, cols, dW2, dpool2, cols2, R, RR, K, nb1, rpb);
    else hipLaunchKernelGGL(poolgen_bwd_pool_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, emb, dW, dpool, cols, dW2, dpool2, cols2, R, RR, K, nb1, rpb);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_poolgen_bwd_emb(const float* dW, const float* pool, int cols, const float* dW2, const float* pool2, int cols2,
                                     float* demb, int R, int nsplit, int K, void* stream) {
    if (!dW || !pool || !demb || K > PG_MAXK || K <= 0) return GPTST_EARG;
    if (!dW2) cols2 = 0;
    const int V = ((cols & 3) || (cols2 & 3)) ? 1 : 4;
    if (V == 4) hipLaunchKernelGGL(poolgen_bwd_emb_kernel<4>, dim3(R), dim3(256), 0, (hipStream_t)stream, dW, pool, cols, dW2, pool2, cols2, demb, R, nsplit, K);
    else hipLaunchKernelGGL(poolgen_bwd_emb_kernel<1>, dim3(R), dim3(256), 0, (hipStream_t)stream, dW, pool, cols, dW2, pool2, cols2, demb, R, nsplit, K);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
