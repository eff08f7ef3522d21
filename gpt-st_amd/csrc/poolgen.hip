// Embedding-conditioned parameter generation:  out[r, :] = sum_k emb[r,k] * pool[k, :]   (K = embed dim <= 16)
//
// This is the reference's  einsum('btd,dio->btio') / einsum('nd,dio->nio') / matmul(emb, bias_pool)
// (GPTST.py:24-25,29-30,137-138,160-161), einsum('btd,dhn->bthn') (:104), einsum('bd,dhk->bhk') (:129) and
// einsum('nk,kht->nht') (:156): a skinny GEMM (R rows <= a few hundred, K <= 16, up to C*C columns).  It is HBM/L2
// bound on the R x cols operand, so it runs on the VALU with coalesced float4 columns; the two gradient reductions are here
// too.  Every kernel takes up to PG_MAXP problems that share emb (weights_pool + bias_pool of several layers in ONE launch:
// each launch has a ~4-5 us latency floor on this GPU, so per-layer launches are batched per STHCN).
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

#define PG_MAXP 8
struct PgFwd { const float* pool[PG_MAXP]; float* out[PG_MAXP]; int cols[PG_MAXP]; int blk0[PG_MAXP + 1]; int n; };
struct PgBwd {
    const float* dW[PG_MAXP]; const float* pool[PG_MAXP]; float* dpool[PG_MAXP];
    int cols[PG_MAXP]; int nsplit[PG_MAXP]; int blk0[PG_MAXP + 1]; int n;
};
template <class A> __device__ __forceinline__ int pg_find(const A& a, int bx) {
    int p = 0;
#pragma unroll
    for (int q = 1; q < PG_MAXP; ++q) if (q < a.n && bx >= a.blk0[q]) p = q;
    return p;
}

// out_p[r, :] = sum_k emb[r,k] pool_p[k, :].   grid: (sum_p ceil(cols_p/V/256), ceil(R/PG_ROWS))
template <int V>
__global__ __launch_bounds__(256) void poolgen_fwd_kernel(const float* __restrict__ emb, PgFwd a, int R, int K) {
    const int p = pg_find(a, blockIdx.x);
    const float* __restrict__ pool = a.pool[p];
    float* __restrict__ out = a.out[p];
    const int cols = a.cols[p];
    const int c4 = (blockIdx.x - a.blk0[p]) * 256 + threadIdx.x;
    if (V * c4 >= cols) return;
    float4 pv[PG_MAXK];
#pragma unroll
    for (int k = 0; k < PG_MAXK; ++k) pv[k] = (k < K) ? ldv<V>(pool + (size_t)k * cols + V * c4) : f4zero();
    const int r0 = blockIdx.y * PG_ROWS;
#pragma unroll 1
    for (int r = r0; r < min(R, r0 + PG_ROWS); ++r) {
        float4 acc = f4zero();
#pragma unroll
        for (int k = 0; k < PG_MAXK; ++k)
            if (k < K) acc = f4fma(emb[(size_t)r * K + k], pv[k], acc);
        stv<V>(out + (size_t)r * cols + V * c4, acc);
    }
}

// dpool_p[k, c] += sum_rr emb[rr % R, k] * dW_p[rr, c]   (rr < R*nsplit_p: wgrad's K-splits are summed here)
// on fp32 MFMA 16x16x4:  D[i = k][j] += A[i][kk] B[kk][j],  A = emb[row][k],  B = dW[row][col];  lane (kk = l>>4, j = l&15).
// V = 4: a lane fetches the float4 dW[row+kk][c0 + 4j ..] (the four lane groups read four consecutive rows, 256 B each) and
// component e feeds column tile e (columns c0 + 4j + e), so one load drives 4 MFMAs and a wave owns a 64-column slab.
// The reduction over rows happens inside the MFMA; a workgroup's 4 waves take 4 row chunks of the slab and every output
// gets g_pg_nchunk atomics in total.   grid: (sum_p slabs_p, ceil(nchunk / 4))
int g_pg_nchunk = 4;
template <int V>
__global__ __launch_bounds__(256) void poolgen_bwd_pool_kernel(const float* __restrict__ emb, PgBwd a, int R, int K, int nchunk) {
    constexpr int SLAB = 16 * V;
    const int p = pg_find(a, blockIdx.x);
    const float* __restrict__ dW = a.dW[p];
    float* __restrict__ dpool = a.dpool[p];
    const int cols = a.cols[p], RR = R * a.nsplit[p];
    const int bx = blockIdx.x - a.blk0[p];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int chunk = blockIdx.y * 4 + wave;
    int per = (RR + nchunk - 1) / nchunk;
    per = (per + 3) & ~3;
    const int r0 = chunk < nchunk ? chunk * per : RR, r1 = min(RR, r0 + per);       // an idle wave gets an empty row range
    const int c = bx * SLAB + V * j;
    const bool cok = c < cols;
    f32x4 acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // NU k-steps (4 rows each) per batch: all dW / emb loads of a batch are issued before the first MFMA (row and column are
    // clamped instead of predicated; a row beyond the chunk contributes through a zero emb operand).  The per-16-row loop of
    // the first version paid one memory round trip per 16 rows with 4 waves per CU: 23 us for 25 MB.
    constexpr int NU = 4;
    const int cl = cok ? c : 0;
    for (int rb = r0; rb < r1; rb += 4 * NU) {
        float av[NU];
        float4 b[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int row = min(rb + 4 * u + kk, r1 - 1);
            av[u] = emb[(size_t)(row % R) * K + min(j, K - 1)];
            b[u] = ldv<V>(dW + (size_t)row * cols + cl);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const float a_ = (rb + 4 * u + kk < r1 && j < K) ? av[u] : 0.f;
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b[u].x, acc[0], 0, 0, 0);
            if (V == 4) {
                acc[V > 1 ? 1 : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b[u].y, acc[V > 1 ? 1 : 0], 0, 0, 0);
                acc[V > 2 ? 2 : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b[u].z, acc[V > 2 ? 2 : 0], 0, 0, 0);
                acc[V > 3 ? 3 : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b[u].w, acc[V > 3 ? 3 : 0], 0, 0, 0);
            }
        }
    }
    // fold the 4 waves (= 4 row chunks of the same slab) through LDS: one atomic per output and workgroup instead of four
    // (the atomics, not the loads, bounded this kernel: 4 x 16 x 4160 x 4 problems = 1 M atomics per launch).
    __shared__ float fold[4][PG_MAXK][SLAB + 1];
#pragma unroll
    for (int e = 0; e < V; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) fold[wave][kk * 4 + r][V * j + e] = acc[e][r];      // D reg r: row (l>>4)*4 + r = k, col l&15 = j
    __syncthreads();
    for (int o = threadIdx.x; o < PG_MAXK * SLAB; o += 256) {
        const int k = o / SLAB, cc = o % SLAB, col = bx * SLAB + cc;
        if (k < K && col < cols) atomicAdd(dpool + (size_t)k * cols + col, fold[0][k][cc] + fold[1][k][cc] + fold[2][k][cc] + fold[3][k][cc]);
    }
}

// demb[r, k] += sum_p sum_split sum_c dW_p[split*R + r, c] * pool_p[k, c]   on fp32 MFMA 16x16x4:
// D[i = row][j = k] += A[i][kk] B[kk][j] with A = dW[row0+i][c], B = pool[j][c]; lane (kk = l>>4, i = l&15) fetches
// float4s at c + 4kk so one load pair feeds four MFMA steps (the usual k-permutation); the MFMA does the reduction over the
// columns that a VALU version would have to do with cross-lane shuffles.  A wave owns (16-row tile, column chunk of one problem).
// grid: (ceil(R/16), ceil(total chunks / 4)); blk0[] counts chunks here.
template <int V>
__global__ __launch_bounds__(256) void poolgen_bwd_emb_kernel(PgBwd a, float* __restrict__ demb, int R, int K, int chunk_cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int row = blockIdx.x * 16 + i;
    const int gch = blockIdx.y * 4 + wave;
    if (gch >= a.blk0[a.n]) return;
    const int p = pg_find(a, gch);
    const float* __restrict__ w = a.dW[p];
    const float* __restrict__ pl = a.pool[p];
    const int cc = a.cols[p], ns = a.nsplit[p];
    const int cbeg = (gch - a.blk0[p]) * chunk_cols, cend = min(cc, cbeg + chunk_cols);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (V == 4) {
        // UC column steps (16 columns each) per batch, all loads issued before the MFMAs (clamped, not predicated): the
        // one-step loop paid one L2 round trip per 16 columns (32 per chunk -> 22 us).
        constexpr int UC = 8;
        const int rowc = min(row, R - 1), ic = min(i, K - 1);
        const bool rok = row < R, kok = i < K;
        for (int c0 = cbeg; c0 < cend; c0 += 16 * UC) {
            float4 av[UC], b[UC];
#pragma unroll
            for (int u = 0; u < UC; ++u) {
                const int c = min(c0 + 16 * u + 4 * kk, cc - 4);
                av[u] = ld4(w + (size_t)rowc * cc + c);
                b[u] = ld4(pl + (size_t)ic * cc + c);
            }
            for (int s = 1; s < ns; ++s) {
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    const int c = min(c0 + 16 * u + 4 * kk, cc - 4);
                    av[u] = f4add(av[u], ld4(w + ((size_t)s * R + rowc) * cc + c));
                }
            }
#pragma unroll
            for (int u = 0; u < UC; ++u) {
                const bool ok = c0 + 16 * u + 4 * kk < cend;
                const float4 a_ = (ok && rok) ? av[u] : f4zero();
                const float4 b_ = (ok && kok) ? b[u] : f4zero();
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.x, b_.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.y, b_.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.z, b_.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.w, b_.w, acc, 0, 0, 0);
            }
        }
    } else {
        for (int c = cbeg + kk; c < cend + kk; c += 4) {
            float av = 0.f, b = 0.f;
            if (c < cend) {
                if (row < R) for (int s = 0; s < ns; ++s) av += w[((size_t)s * R + row) * cc + c];
                if (i < K) b = pl[(size_t)i * cc + c];
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int orow = blockIdx.x * 16 + kk * 4 + r;      // D reg r: row (l>>4)*4 + r, col l&15
        if (orow < R && i < K) atomicAdd(demb + (size_t)orow * K + i, acc[r]);
    }
}

extern int g_wgrad_ns_override;
extern int g_wgrad_ns0_override;
extern int g_apply_v1;
extern int g_apply_tpw;
extern "C" int gptst_tune(int id, int value) {
    if (id == 1) g_pg_nchunk = value;
    if (id == 2) g_wgrad_ns_override = value;
    if (id == 5) g_wgrad_ns0_override = value;
    if (id == 3) g_apply_v1 = value;
    if (id == 4) g_apply_tpw = value;
    return GPTST_OK;
}

static int pg_vec(int n, const int* cols) {
    for (int p = 0; p < n; ++p) if (cols[p] & 3) return 1;
    return 4;
}

extern "C" int gptst_poolgen_fwd_multi(const float* emb, int nprob, const void* pools, const void* outs, const int* cols, int R, int K,
                                       void* stream) {
    if (!emb || !pools || !outs || !cols || nprob <= 0 || nprob > PG_MAXP || K > PG_MAXK || K <= 0) return GPTST_EARG;
    const int V = pg_vec(nprob, cols);
    PgFwd a; a.n = nprob; a.blk0[0] = 0;
    for (int p = 0; p < nprob; ++p) {
        a.pool[p] = ((const float* const*)pools)[p]; a.out[p] = ((float* const*)outs)[p]; a.cols[p] = cols[p];
        if (!a.pool[p] || !a.out[p] || cols[p] <= 0) return GPTST_EARG;
        a.blk0[p + 1] = a.blk0[p] + ((cols[p] + V - 1) / V + 255) / 256;
    }
    dim3 grid(a.blk0[nprob], (R + PG_ROWS - 1) / PG_ROWS);
    if (V == 4) hipLaunchKernelGGL(poolgen_fwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, emb, a, R, K);
    else hipLaunchKernelGGL(poolgen_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, emb, a, R, K);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

static int pg_fill_bwd(PgBwd& a, int nprob, const void* dWs, const void* pools, const void* dpools, const int* cols, const int* nsplit) {
    a.n = nprob;
    for (int p = 0; p < nprob; ++p) {
        a.dW[p] = ((const float* const*)dWs)[p];
        a.pool[p] = pools ? ((const float* const*)pools)[p] : nullptr;
        a.dpool[p] = dpools ? ((float* const*)dpools)[p] : nullptr;
        a.cols[p] = cols[p]; a.nsplit[p] = nsplit ? nsplit[p] : 1;
        if (!a.dW[p] || cols[p] <= 0 || a.nsplit[p] <= 0) return GPTST_EARG;
    }
    return GPTST_OK;
}

extern "C" int gptst_poolgen_bwd_pool_multi(const float* emb, int nprob, const void* dWs, const void* dpools, const int* cols,
                                            const int* nsplit, int R, int K, void* stream) {
    if (!emb || !dWs || !dpools || !cols || nprob <= 0 || nprob > PG_MAXP || K > PG_MAXK || K <= 0) return GPTST_EARG;
    const int V = pg_vec(nprob, cols);
    PgBwd a;
    if (pg_fill_bwd(a, nprob, dWs, nullptr, dpools, cols, nsplit)) return GPTST_EARG;
    a.blk0[0] = 0;
    int minrr = 1 << 30;
    for (int p = 0; p < nprob; ++p) {
        if (!a.dpool[p]) return GPTST_EARG;
        a.blk0[p + 1] = a.blk0[p] + (cols[p] + 16 * V - 1) / (16 * V);
        if (R * a.nsplit[p] < minrr) minrr = R * a.nsplit[p];
    }
    int nchunk = g_pg_nchunk;                            // row chunks = atomics per output element
    if (nchunk * 16 > minrr) nchunk = (minrr + 15) / 16;
    if (nchunk < 1) nchunk = 1;
    dim3 grid(a.blk0[nprob], (nchunk + 3) / 4);
    if (V == 4) hipLaunchKernelGGL(poolgen_bwd_pool_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, emb, a, R, K, nchunk);
    else hipLaunchKernelGGL(poolgen_bwd_pool_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, emb, a, R, K, nchunk);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_poolgen_bwd_emb_multi(int nprob, const void* dWs, const void* pools, const int* cols, const int* nsplit, float* demb,
                                           int R, int K, void* stream) {
    if (!dWs || !pools || !cols || !demb || nprob <= 0 || nprob > PG_MAXP || K > PG_MAXK || K <= 0) return GPTST_EARG;
    const int V = pg_vec(nprob, cols);
    PgBwd a;
    if (pg_fill_bwd(a, nprob, dWs, pools, nullptr, cols, nsplit)) return GPTST_EARG;
    const int chunk = 256;                               // columns per wave: 16 MFMA load pairs
    a.blk0[0] = 0;
    for (int p = 0; p < nprob; ++p) {
        if (!a.pool[p]) return GPTST_EARG;
        a.blk0[p + 1] = a.blk0[p] + (cols[p] + chunk - 1) / chunk;
    }
    dim3 grid((R + 15) / 16, (a.blk0[nprob] + 3) / 4);
    if (V == 4) hipLaunchKernelGGL(poolgen_bwd_emb_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a, demb, R, K, chunk);
    else hipLaunchKernelGGL(poolgen_bwd_emb_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a, demb, R, K, chunk);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// ---- one / two problem convenience forms (thin wrappers) -----------------------------------------------------------
extern "C" int gptst_poolgen_fwd(const float* emb, const float* pool, float* out, int cols, const float* pool2, float* out2,
                                 int cols2, int R, int K, void* stream) {
    const float* pools[2] = {pool, pool2}; float* outs[2] = {out, out2}; int cc[2] = {cols, cols2};
    return gptst_poolgen_fwd_multi(emb, pool2 ? 2 : 1, pools, outs, cc, R, K, stream);
}

extern "C" int gptst_poolgen_bwd_pool(const float* emb, const float* dW, float* dpool, int cols, const float* dW2, float* dpool2,
                                      int cols2, int R, int nsplit, int K, void* stream) {
    const float* dWs[2] = {dW, dW2}; float* dps[2] = {dpool, dpool2}; int cc[2] = {cols, cols2}; int ns[2] = {nsplit, 1};
    return gptst_poolgen_bwd_pool_multi(emb, dW2 ? 2 : 1, dWs, dps, cc, ns, R, K, stream);
}

extern "C" int gptst_poolgen_bwd_emb(const float* dW, const float* pool, int cols, const float* dW2, const float* pool2, int cols2,
                                     float* demb, int R, int nsplit, int K, void* stream) {
    const float* dWs[2] = {dW, dW2}; const float* pls[2] = {pool, pool2}; int cc[2] = {cols, cols2}; int ns[2] = {nsplit, 1};
    return gptst_poolgen_bwd_emb_multi(dW2 ? 2 : 1, dWs, pls, cc, ns, demb, R, K, stream);
}
