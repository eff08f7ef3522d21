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
