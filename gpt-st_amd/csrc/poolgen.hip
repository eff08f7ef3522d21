// Embedding-conditioned parameter generation:  out[r, :] = sum_k emb[r,k] * pool[k, :]   (K = embed dim <= 16)
//
// This is the reference's  einsum('btd,dio->btio') / einsum('nd,dio->nio') / matmul(emb, bias_pool)
// (GPTST.py:24-25,29-30,137-138,160-161), einsum('btd,dhn->bthn') (:104), einsum('bd,dhk->bhk') (:129) and
// einsum('nk,kht->nht') (:156): a skinny GEMM (R rows <= a few hundred, K <= 16, up to C*C columns).  It is HBM/L2
// bound on the R x cols operand, so it runs on the VALU with coalesced float4 columns; the two gradient reductions are here
// too.  Every kernel takes an optional second (pool2, cols2) problem sharing emb (weights_pool + bias_pool in one launch).
//
// Gradient kernels are written around ONE rule learnt from the first profile (profiles/r01a): global atomics are only cheap
// when few land on the same address (a same-address atomic serialises at ~80 ns) and when there are few of them overall —
// so each output element is owned by one thread (or a handful of row chunks), never by hundreds of workgroups.
#include "common.h"

#define PG_MAXK 16
#define PG_ROWS 4

// V = 4: float4 columns (cols % 4 == 0, rows 16-byte aligned);  V = 1: scalar columns (e.g. HS*N = 2070 for METR_LA)
template <int V> __device__ __forceinline__ float4 ldv(const float* p) { return ld4(p); }
template <> __device__ __forceinline__ float4 ldv<1>(const float* p) { return make_float4(*p, 0.f, 0.f, 0.f); }
template <int V> __device__ __forceinline__ void stv(float* p, float4 v) { st4(p, v); }
template <> __device__ __forceinline__ void stv<1>(float* p, float4 v) { *p = v.x; }

// grid: (ceil(cols/V/256) + ceil(cols2/V/256), ceil(R/PG_ROWS))
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

// dpool[k, c] += sum_rr emb[rr % R, k] * dW[rr, c]   (rr < R*nsplit for problem 1 — wgrad's K-splits —, rr < R for problem 2)
// on fp32 MFMA 16x16x4:  D[i = k][j] += A[i][kk] B[kk][j],  A = emb[row][k],  B = dW[row][col];  lane (kk = l>>4, j = l&15).
// V = 4: a lane fetches the float4 dW[row+kk][c0 + 4j ..] (the four lane groups read four consecutive rows, 256 B each) and
// component e feeds column tile e (columns c0 + 4j + e), so one load drives 4 MFMAs and a wave owns a 64-column slab.
// The reduction over rows happens inside the MFMA; a workgroup's 4 waves take 4 row chunks of the slab and every output
// gets g_pg_nchunk atomics in total.   grid: (slabs of problem 1 + slabs of problem 2, ceil(nchunk / 4))
int g_pg_nchunk = 4;
template <int V>
__global__ __launch_bounds__(256) void poolgen_bwd_pool_kernel(const float* __restrict__ emb, const float* __restrict__ dW,
                                                               float* __restrict__ dpool, int cols, const float* __restrict__ dW2,
                                                               float* __restrict__ dpool2, int cols2, int R, int RR, int K,
                                                               int nblk1, int nchunk) {
    constexpr int SLAB = 16 * V;
    int bx = blockIdx.x;
    if (bx >= nblk1) { bx -= nblk1; dW = dW2; dpool = dpool2; cols = cols2; RR = R; }   // splits apply to problem 1 only
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int chunk = blockIdx.y * 4 + wave;
    if (chunk >= nchunk) return;
    int per = (RR + nchunk - 1) / nchunk;
    per = (per + 3) & ~3;
    const int r0 = chunk * per, r1 = min(RR, r0 + per);
    const int c = bx * SLAB + V * j;
    const bool cok = c < cols;
    f32x4 acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int rb = r0; rb < r1; rb += 16) {
        float a[4];
        float4 b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = rb + 4 * u + kk;
            a[u] = 0.f; b[u] = f4zero();
            if (row < r1) {
                if (j < K) a[u] = emb[(size_t)(row % R) * K + j];
                if (cok) b[u] = ldv<V>(dW + (size_t)row * cols + c);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u].x, acc[0], 0, 0, 0);
            if (V == 4) {
                acc[V > 1 ? 1 : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u].y, acc[V > 1 ? 1 : 0], 0, 0, 0);
                acc[V > 2 ? 2 : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u].z, acc[V > 2 ? 2 : 0], 0, 0, 0);
                acc[V > 3 ? 3 : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u].w, acc[V > 3 ? 3 : 0], 0, 0, 0);
            }
        }
    }
    if (cok) {
#pragma unroll
        for (int e = 0; e < V; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = kk * 4 + r;                  // D reg r: row (l>>4)*4 + r = k, col l&15 = j
                if (k < K) atomicAdd(dpool + (size_t)k * cols + c + e, acc[e][r]);
            }
    }
}

// demb[r, k] += sum_split sum_c dW[split*R + r, c] * pool[k, c]  (+ second problem, no splits)  on fp32 MFMA 16x16x4:
// D[i = row][j = k] += A[i][kk] B[kk][j] with A = dW[row0+i][c], B = pool[j][c]; lane (kk = l>>4, i = l&15) fetches
// float4s at c + 4kk so one load pair feeds four MFMA steps (the usual k-permutation); the MFMA does the reduction over the
// columns that a VALU version would have to do with cross-lane shuffles.  A wave owns (16-row tile, column chunk).
// grid: (ceil(R/16), column chunks / 4); 4 waves = 4 column chunks per workgroup.
template <int V>
__global__ __launch_bounds__(256) void poolgen_bwd_emb_kernel(const float* __restrict__ dW, const float* __restrict__ pool, int cols,
                                                              const float* __restrict__ dW2, const float* __restrict__ pool2,
                                                              int cols2, float* __restrict__ demb, int R, int nsplit, int K,
                                                              int chunk_cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int row = blockIdx.x * 16 + i;
    const int ch = blockIdx.y * 4 + wave;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int pb = 0; pb < 2; ++pb) {
        const float* w = pb ? dW2 : dW;
        const float* p = pb ? pool2 : pool;
        const int cc = pb ? cols2 : cols;
        const int ns = pb ? 1 : nsplit;              // splits apply to problem 1 only
        if (w == nullptr) continue;                  // uniform
        const int cbeg = pb ? (ch == 0 ? 0 : cc) : ch * chunk_cols;            // the small second problem goes to chunk 0
        const int cend = pb ? cc : min(cc, cbeg + chunk_cols);
        if (V == 4) {
            for (int c = cbeg + 4 * kk; c < cend + 4 * kk; c += 16) {           // same trip count for the 4 lane groups
                float4 a = f4zero(), b = f4zero();
                if (c < cend) {
                    if (row < R) {
                        a = ld4(w + (size_t)row * cc + c);
                        for (int s = 1; s < ns; ++s) a = f4add(a, ld4(w + ((size_t)s * R + row) * cc + c));
                    }
                    if (i < K) b = ld4(p + (size_t)i * cc + c);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
            }
        } else {
            for (int c = cbeg + kk; c < cend + kk; c += 4) {
                float a = 0.f, b = 0.f;
                if (c < cend) {
                    if (row < R) for (int s = 0; s < ns; ++s) a += w[((size_t)s * R + row) * cc + c];
                    if (i < K) b = p[(size_t)i * cc + c];
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int orow = blockIdx.x * 16 + kk * 4 + r;      // D reg r: row (l>>4)*4 + r, col l&15
        if (orow < R && i < K) atomicAdd(demb + (size_t)orow * K + i, acc[r]);
    }
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

extern "C" int gptst_tune(int id, int value) {
    if (id == 1) g_pg_nchunk = value;
    return GPTST_OK;
}

extern "C" int gptst_poolgen_bwd_pool(const float* emb, const float* dW, float* dpool, int cols, const float* dW2, float* dpool2,
                                      int cols2, int R, int nsplit, int K, void* stream) {
    if (!emb || !dW || !dpool || K > PG_MAXK || K <= 0) return GPTST_EARG;
    if (!dW2) cols2 = 0;
    const int V = ((cols & 3) || (cols2 & 3)) ? 1 : 4;
    const int RR = R * nsplit;
    const int slab = 16 * V;
    const int nb1 = (cols + slab - 1) / slab, nb2 = dW2 ? (cols2 + slab - 1) / slab : 0;
    int nchunk = g_pg_nchunk;                            // row chunks = atomics per output element
    if (nchunk * 16 > RR) nchunk = (RR + 15) / 16;
    if (nchunk < 1) nchunk = 1;
    dim3 grid(nb1 + nb2, (nchunk + 3) / 4);
    if (V == 4) hipLaunchKernelGGL(poolgen_bwd_pool_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, emb, dW, dpool, cols, dW2, dpool2, cols2, R, RR, K, nb1, nchunk);
    else hipLaunchKernelGGL(poolgen_bwd_pool_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, emb, dW, dpool, cols, dW2, dpool2, cols2, R, RR, K, nb1, nchunk);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_poolgen_bwd_emb(const float* dW, const float* pool, int cols, const float* dW2, const float* pool2, int cols2,
                                     float* demb, int R, int nsplit, int K, void* stream) {
    if (!dW || !pool || !demb || K > PG_MAXK || K <= 0) return GPTST_EARG;
    if (!dW2) cols2 = 0;
    const int V = ((cols & 3) || (cols2 & 3)) ? 1 : 4;
    int chunk = 256;                                     // columns per wave: 16 MFMA load pairs
    int nch = (cols + chunk - 1) / chunk;
    nch = (nch + 3) / 4 * 4;
    dim3 grid((R + 15) / 16, nch / 4);
    if (V == 4) hipLaunchKernelGGL(poolgen_bwd_emb_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, dW, pool, cols, dW2, pool2, cols2, demb, R, nsplit, K, chunk);
    else hipLaunchKernelGGL(poolgen_bwd_emb_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, dW, pool, cols, dW2, pool2, cols2, demb, R, nsplit, K, chunk);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
