// Device-side pieces of the job-table kernels (poolgen.hip) that another launch embeds: the cooperative mask launch of masksel.hip runs the step's
// forward generation jobs on the CUs its 64 workgroups leave idle (r05).
#pragma once
#include "common.h"

#define PG_MAXK 16
#define PG_MFMA_ROWS 64 // rows per block of the MFMA forward (four 16-row tiles per wave and B-fragment load)
#define PJ_MAX 112     // 112 x 64 B = 7 KB of kernel arguments (the AQL kernarg segment is not limited to 4 KB); a pretraining step queues ~100 reductions -> 1 launch
enum { PJ_FWD = 0, PJ_BWD_POOL = 1, PJ_BWD_EMB = 2, PJ_GRAM = 3 };
struct PJob {
    const float* emb;      // FWD / BWD_POOL: (R, K)
    const float* x;        // BWD_POOL / BWD_EMB: dW (R * nsplit, cols)
    const float* pool;     // FWD / BWD_EMB: (K, cols)
    float* out;            // FWD: (R, cols);  BWD_POOL: dpool (K, cols) +=;  BWD_EMB: demb (R, K) +=;  GRAM: (R, T, T)
    int R, K, cols, nsplit;
    int blk0, kind, nbx, ldx;   // ldx: row stride of x (>= cols: x may be a column window of a wider matrix)
};
struct PJobs { PJob j[PJ_MAX]; int n; };

// MFMA forward of one job block: 256 threads `tid` = 4 waves x 64 output columns, `rows` rows from by * rows.  No LDS, no barrier: any 256-thread
// quarter of a larger workgroup can run it.
__device__ __forceinline__ void pj_fwd_mfma(const PJob& a, int bx, int by, int rows, int tid) {
    const float* __restrict__ emb = a.emb;
    const float* __restrict__ pool = a.pool;
    float* __restrict__ out = a.out;
    const int cols = a.cols, K = a.K, R = a.R;
    const int lane = tid & 63, wave = tid >> 6, j = lane & 15, kk = lane >> 4;
    const int c = bx * 256 + wave * 64 + 4 * j;
    if (bx * 256 + wave * 64 >= cols) return;                      // whole wave beyond the last column
    const bool cok = c < cols;
    const int nks = (K + 3) >> 2;                                  // k-steps (K <= 16)
    float4 bf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bf[s] = (cok && 4 * s + kk < K) ? ld4(pool + (size_t)(4 * s + kk) * cols + c) : f4zero();
    const int r0 = by * rows, r1 = min(R, r0 + rows);
    for (int rt = r0; rt < r1; rt += 16) {
        float av[4];
        const int row = rt + j;
#pragma unroll
        for (int s = 0; s < 4; ++s) av[s] = (row < r1 && 4 * s + kk < K) ? emb[(size_t)row * K + 4 * s + kk] : 0.f;
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < nks) {                                             // uniform
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bf[s].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bf[s].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bf[s].z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bf[s].w, acc[3], 0, 0, 0);
            }
        }
        if (cok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = rt + 4 * kk + r;
                if (orow < r1) st4(out + (size_t)orow * cols + c, make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]));
            }
        }
    }
}

// rows of a temporal-graph (kind 3) job one workgroup takes, and the job itself on `nt` threads (scr: PJ_GRAM_SCR floats of LDS; two barriers)
#define PJ_GRAM_SCR (4 * PG_MAXK * 65)
__host__ __device__ inline int pj_gram_rows(int K, int cols) {
    const int nb = (PJ_GRAM_SCR - K * cols) / cols;
    return nb > 16 ? 16 : nb;                      // <= 0: the shape does not fit (EARG)
}
__device__ __forceinline__ void pj_gram(const PJob& a, int bx, float* __restrict__ scr, int nt) {
    const int K = a.K, cols = a.cols, Hm = cols / 12, NB = pj_gram_rows(K, cols);
    float* pl = scr;                               // [K][cols]
    float* As = scr + K * cols;                    // [NB][cols]
    const int r0 = bx * NB, nr = min(NB, a.R - r0);
    for (int i = threadIdx.x; i < K * cols; i += nt) pl[i] = a.pool[i];
    __syncthreads();
    for (int i = threadIdx.x; i < nr * cols; i += nt) {
        const int r = i / cols, c = i % cols;
        const float* __restrict__ e = a.emb + (size_t)(r0 + r) * K;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(e[k], pl[k * cols + c], acc);
        As[i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nr * 144; i += nt) {
        const int r = i / 144, t = (i % 144) / 12, u = i % 12;
        const float* ar = As + r * cols;
        float s = 0.f;
        for (int h = 0; h < Hm; ++h) s = fmaf(ar[h * 12 + t], ar[h * 12 + u], s);
        a.out[(size_t)(r0 + r) * 144 + i % 144] = s;
    }
}

// ---- the two gradient reductions (kinds 1 and 2), 256 threads `tid` each: also run as role workgroups of a backward launch that leaves CUs and HBM
// bandwidth idle (cap_mfma.hip, r05) ----
#ifndef PJ_UC
#define PJ_UC 4
#endif
#ifndef PJ_NU
#define PJ_NU 4
#endif
// V = 4: float4 columns (cols % 4 == 0, rows 16-byte aligned);  V = 1: scalar columns (e.g. HS*N = 2070 for METR_LA)
template <int V> __device__ __forceinline__ float4 ldv(const float* p) { return ld4(p); }
template <> __device__ __forceinline__ float4 ldv<1>(const float* p) { return make_float4(*p, 0.f, 0.f, 0.f); }
template <int V> __device__ __forceinline__ void stv(float* p, float4 v) { st4(p, v); }
template <> __device__ __forceinline__ void stv<1>(float* p, float4 v) { *p = v.x; }

template <int V>
__device__ __forceinline__ void pj_bwd_pool(const PJob& a, int bx, float (*fold)[PG_MAXK][65], int tid) {
    constexpr int SLAB = 16 * V;
    const float* __restrict__ emb = a.emb;
    const float* __restrict__ dW = a.x;
    float* __restrict__ dpool = a.out;
    const int cols = a.cols, R = a.R, K = a.K, RR = R * a.nsplit;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    int nchunk = 4;
    if (nchunk * 16 > RR) nchunk = (RR + 15) / 16;
    int per = (RR + nchunk - 1) / nchunk;
    per = (per + 3) & ~3;
    const int r0 = wave < nchunk ? wave * per : RR, r1 = min(RR, r0 + per);       // an idle wave gets an empty row range
    const int c = bx * SLAB + V * j;
    const bool cok = c < cols;
    f32x4 acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // NU k-steps (4 rows each) per batch: all dW / emb loads of a batch are issued before the first MFMA (row and column are
    // clamped instead of predicated; a row beyond the chunk contributes through a zero emb operand).
    constexpr int NU = PJ_NU;
    const int cl = cok ? c : 0;
    for (int rb = r0; rb < r1; rb += 4 * NU) {
        float av[NU];
        float4 b[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int row = min(rb + 4 * u + kk, r1 - 1);
            av[u] = emb[(size_t)(row % R) * K + min(j, K - 1)];
            b[u] = ldv<V>(dW + (size_t)row * a.ldx + cl);
        }
        SB();
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
#pragma unroll
    for (int e = 0; e < V; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) fold[wave][kk * 4 + r][V * j + e] = acc[e][r];      // D reg r: row (l>>4)*4 + r = k, col l&15 = j
    __syncthreads();
    for (int o = tid; o < PG_MAXK * SLAB; o += 256) {
        const int k = o / SLAB, cc = o % SLAB, col = bx * SLAB + cc;
        if (k < K && col < cols) {
            float* d = dpool + (size_t)k * cols + col;
            *d += (fold[0][k][cc] + fold[1][k][cc]) + (fold[2][k][cc] + fold[3][k][cc]);      // this workgroup owns the element
        }
    }
}

// demb[r, k] += sum_split sum_c dW[split*R + r, c] * pool[k, c]   on fp32 MFMA 16x16x4:
// D[i = row][j = k] += A[i][kk] B[kk][j] with A = dW[row0+i][c], B = pool[j][c]; lane (kk = l>>4, i = l&15) fetches
// float4s at c + 4kk so one load pair feeds four MFMA steps (the usual k-permutation); the MFMA does the reduction over the
// columns that a VALU version would have to do with cross-lane shuffles.  A wave owns (16-row tile, 256-column chunk); several
// jobs (and chunks) add into one demb, so the final add is an atomic.   blocks: (ceil(R/16), ceil(chunks/4))
// accumulate one 256-column chunk of one job into acc (D[i = row][j = k])
template <int V>
__device__ __forceinline__ void pj_emb_accum(const PJob& a, int bx, int chunk, f32x4& acc, int tid) {
    constexpr int chunk_cols = 256;
    const int lane = tid & 63;
    const int i = lane & 15, kk = lane >> 4;
    const int R = a.R, K = a.K;
    const int row = bx * 16 + i;
    const float* __restrict__ w = a.x;
    const float* __restrict__ pl = a.pool;
    const int cc = a.cols, ns = a.nsplit, ldx = a.ldx;
    const int cbeg = chunk * chunk_cols, cend = min(cc, cbeg + chunk_cols);
    if (cbeg >= cc) return;
    if (V == 4) {
        // UC column steps (16 columns each) per batch, all loads issued before the MFMAs (clamped, not predicated)
        constexpr int UC = PJ_UC;
        const int rowc = min(row, R - 1), ic = min(i, K - 1);
        const bool rok = row < R, kok = i < K;
        for (int c0 = cbeg; c0 < cend; c0 += 16 * UC) {
            float4 av[UC], b[UC];
#pragma unroll
            for (int u = 0; u < UC; ++u) {
                const int c = min(c0 + 16 * u + 4 * kk, cc - 4);
                av[u] = ld4(w + (size_t)rowc * ldx + c);
                b[u] = ld4(pl + (size_t)ic * cc + c);
            }
            for (int s = 1; s < ns; ++s) {
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    const int c = min(c0 + 16 * u + 4 * kk, cc - 4);
                    av[u] = f4add(av[u], ld4(w + ((size_t)s * R + rowc) * ldx + c));
                }
            }
            SB();
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
                if (row < R) for (int s = 0; s < ns; ++s) av += w[((size_t)s * R + row) * ldx + c];
                if (i < K) b = pl[(size_t)i * cc + c];
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc, 0, 0, 0);
        }
    }
}

template <int V>
__device__ __forceinline__ void pj_bwd_emb(const PJob& a, int bx, int by, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kk = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    pj_emb_accum<V>(a, bx, by * 4 + wave, acc, tid);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int orow = bx * 16 + kk * 4 + r;      // D reg r: row (l>>4)*4 + r, col l&15
        if (orow < a.R && i < a.K && (by * 4 + wave) * 256 < a.cols) atomicAdd(a.out + (size_t)orow * a.K + i, acc[r]);
    }
}
