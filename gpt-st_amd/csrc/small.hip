// Thin projections at the edges of the network — all HBM bound, VALU only, float4 rows:
//   lin_in   : Y[i,:]  = sum_{j<J} a[i,j] W(:,j) + b      J = input_base_dim (1..2) or HS; optional masking of a
//              (dim_in_flow with mask / scaler_zeros, GPTST.py:416-418; MLP_RL.ln1 :22; data-gradients of the C->J Linears)
//   rowdot   : Z[i,j]  = X[i,:] . W[j,:] + b[j], optional softmax over j
//              (dim_flow_out :455; MLP_RL.ln3 + softmax :33,:332/:343)
//   rowouter : out(j,c) += sum_i a[i,j] X[i,c]  (+ column sums)          weight / bias gradients of the above
#include "common.h"

#define SM_MAXJ 64
#define SM_MAXGRID 8192                     // workgroups of the row-streaming kernels (grid-stride beyond)
static inline unsigned sm_grid(int rows, int rpb) { const long g = ((long)rows + rpb - 1) / rpb; return (unsigned)(g < SM_MAXGRID ? g : SM_MAXGRID); }

// a'[i,j] = mask ? (mask[i*J+j] != 0 ? a : fill) : a          a has row stride lda
__device__ __forceinline__ float masked_a(const float* __restrict__ a, const float* __restrict__ mask, float fill, size_t i, int j,
                                          int lda, int J) {
    const float v = a[i * lda + j];
    if (mask == nullptr) return v;
    return mask[i * J + j] != 0.f ? v : fill;
}

// grid: ceil(rows / (256/(C/4))), block 256.  wlayout 0: W[c*J + j] (nn.Linear weight (C,J));  1: W[j*C + c].
template <int C>
__global__ __launch_bounds__(256) void lin_in_kernel(const float* __restrict__ a, int lda, const float* __restrict__ mask, float fill,
                                                     const float* __restrict__ W, int wlayout, const float* __restrict__ b,
                                                     float* __restrict__ Y, int rows, int J) {
    constexpr int LPR = C / 4, RPB = 256 / LPR;
    __shared__ float Ws[SM_MAXJ * C];           // stored [j][c]
    for (int i = threadIdx.x; i < J * C; i += 256) {
        const int j = i / C, c = i % C;
        Ws[i] = wlayout ? W[i] : W[c * J + j];
    }
    __syncthreads();
    const int c4 = threadIdx.x % LPR;
    const float4 b4 = b ? ld4(b + 4 * c4) : f4zero();
    // grid-stride over the rows: the launcher caps the grid (SM_MAXGRID), so that at N = 4096 (1.5 M rows) the weight is staged a few
    // thousand times instead of once per 8 rows
    for (size_t i = (size_t)blockIdx.x * RPB + threadIdx.x / LPR; i < (size_t)rows; i += (size_t)gridDim.x * RPB) {
        float4 acc = b4;
        for (int j = 0; j < J; ++j) acc = f4fma(masked_a(a, mask, fill, i, j, lda, J), ld4(Ws + j * C + 4 * c4), acc);
        st4(Y + i * C + 4 * c4, acc);
    }
}

// J <= 2 (the input projections of the step: J = input_base_dim): the weight columns live in registers (no LDS staging, no barrier), a
// workgroup takes 4 x (256 / (C/4)) rows and issues every a / mask load before the first store (one round trip per workgroup instead of
// weight round trip -> barrier -> operand round trip -> store for 16 rows).  Same arithmetic order as lin_in_kernel.
template <int C, int JJ>
__global__ __launch_bounds__(256) void lin_in_small_kernel(const float* __restrict__ a, int lda, const float* __restrict__ mask, float fill,
                                                           const float* __restrict__ W, int wlayout, const float* __restrict__ b,
                                                           float* __restrict__ Y, int rows) {
    constexpr int LPR = C / 4, RPB = 256 / LPR, U = 4;
    const int c4 = threadIdx.x % LPR, r = threadIdx.x / LPR;
    float4 w[JJ];
#pragma unroll
    for (int j = 0; j < JJ; ++j)
        w[j] = wlayout ? ld4(W + j * C + 4 * c4)
                       : make_float4(W[(4 * c4) * JJ + j], W[(4 * c4 + 1) * JJ + j], W[(4 * c4 + 2) * JJ + j], W[(4 * c4 + 3) * JJ + j]);
    const float4 b4 = b ? ld4(b + 4 * c4) : f4zero();
    const size_t i0 = (size_t)blockIdx.x * RPB * U + r;
    float av[U][JJ], mv[U][JJ];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = min(i0 + (size_t)u * RPB, (size_t)rows - 1);
#pragma unroll
        for (int j = 0; j < JJ; ++j) { av[u][j] = a[i * lda + j]; mv[u][j] = mask ? mask[i * JJ + j] : 1.f; }
    }
    SB();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = i0 + (size_t)u * RPB;
        if (i >= (size_t)rows) break;
        float4 acc = b4;
#pragma unroll
        for (int j = 0; j < JJ; ++j) acc = f4fma(mv[u][j] != 0.f ? av[u][j] : fill, w[j], acc);
        st4(Y + i * C + 4 * c4, acc);
    }
}

// Z[i,j] = X[i,:].W[j,:] + b[j];  softmax over j when do_softmax.  C/4 lanes per row, butterfly reduce.
template <int C, bool LOOP>
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ b,
                                                     float* __restrict__ Z, int rows, int J, int do_softmax, int* __restrict__ label) {
    constexpr int LPR = C / 4, RPB = 256 / LPR;
    __shared__ float Ws[SM_MAXJ * C];
    for (int i = threadIdx.x; i < J * C / 4; i += 256) st4(Ws + 4 * i, ld4(W + 4 * i));
    __syncthreads();
    const int c4 = threadIdx.x % LPR;
    // LOOP: grid-stride over row blocks (workgroup-uniform trip count: cross-lane ops inside) when rows > SM_MAXGRID * RPB; the one-trip
    // instantiation keeps the straight-line code (the loop form cost 7 us per launch at the bench shape)
#pragma unroll 1
    for (size_t i0 = (size_t)blockIdx.x * RPB; i0 < (LOOP ? (size_t)rows : (size_t)blockIdx.x * RPB + 1); i0 += (size_t)gridDim.x * RPB) {
    const size_t i = i0 + threadIdx.x / LPR;
    const bool valid = i < (size_t)rows;
    const float4 x = valid ? ld4(X + i * C + 4 * c4) : f4zero();
    float mine[(SM_MAXJ + LPR - 1) / LPR];      // logits owned by this lane: j = c4 + q*LPR
    float m = -3.0e38f;
#pragma unroll
    for (int q = 0; q < (SM_MAXJ + LPR - 1) / LPR; ++q) {
        mine[q] = 0.f;
        for (int jj = 0; jj < LPR; ++jj) {
            const int j = q * LPR + jj;
            if (j < J) {                                       // uniform
                const float u = group_sum<LPR>(f4dot(x, ld4(Ws + j * C + 4 * c4))) + (b ? b[j] : 0.f);
                if (jj == c4) mine[q] = u;
            }
        }
        if (q * LPR + c4 < J) m = fmaxf(m, mine[q]);
    }
    if (do_softmax) {
        m = group_max<LPR>(m);
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < (SM_MAXJ + LPR - 1) / LPR; ++q)
            if (q * LPR + c4 < J) { mine[q] = expf(mine[q] - m); s += mine[q]; }
        s = group_sum<LPR>(s);
        const float inv = 1.f / s;
#pragma unroll
        for (int q = 0; q < (SM_MAXJ + LPR - 1) / LPR; ++q) mine[q] *= inv;
    }
    if (valid) {
#pragma unroll
        for (int q = 0; q < (SM_MAXJ + LPR - 1) / LPR; ++q)
            if (q * LPR + c4 < J) Z[i * J + q * LPR + c4] = mine[q];
    }
    if (label != nullptr) {       // label[i] = argmax_j Z[i,j], FIRST maximum (GPTST.py:344-345) — on the values just stored
        float bv = -3.0e38f;
        int bj = 1 << 20;
#pragma unroll
        for (int q = 0; q < (SM_MAXJ + LPR - 1) / LPR; ++q)
            if (q * LPR + c4 < J && mine[q] > bv) { bv = mine[q]; bj = q * LPR + c4; }
        const float gm = group_max<LPR>(bv);
        const float cand = bv == gm ? -(float)bj : -3.0e38f;        // smallest index among the lanes that hold the maximum
        const int first = (int)(-group_max<LPR>(cand));
        if (valid && c4 == 0) label[i] = first;
    }
    }
}

// rowdot for C = 64, J <= 16 on MFMA 16x16x4 (r03): Z tile (16 rows x 16 classes) = X tile (16 x 64) . W^T, both operands in registers —
// lane (j, kk): A = X[row tile*16 + j][16q + 4kk ..] straight from global (float4), B = W[class j][16q + 4kk ..] (zero for j >= J) — so
// a row costs 1/16 of 16 MFMAs instead of J dot products of 4 FMAs + a 4-step DPP reduction each (the VALU version: 18.8 us for
// 65 280 rows x 10 classes).  D reg r = Z[row kk*4 + r][class j]: the softmax / arg-max over the classes runs across the 16 lanes of a DPP row.
template <int C>
__global__ __launch_bounds__(256) void rowdot_mfma_kernel(const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ b,
                                                            float* __restrict__ Z, int rows, int J, int do_softmax, int* __restrict__ label,
                                                            int tiles_per_wave) {
    constexpr int Q = C / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    float4 bw[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) bw[q] = j < J ? ld4(W + (size_t)j * C + 16 * q + 4 * kk) : f4zero();
    const float bj = (b != nullptr && j < J) ? b[j] : 0.f;
    const int ntiles = (rows + 15) / 16;
    const int t0 = (blockIdx.x * 4 + wave) * tiles_per_wave, t1 = min(ntiles, t0 + tiles_per_wave);
    for (int tb = t0; tb < t1; tb += 2) {                           // two tiles' loads in flight
        float4 a[2][Q];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float* row = X + (size_t)min((tb + u) * 16 + j, rows - 1) * C + 4 * kk;
#pragma unroll
            for (int q = 0; q < Q; ++q) a[u][q] = ld4(row + 16 * q);
        }
        SB();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (tb + u >= t1) break;                                 // wave-uniform
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][q].x, bw[q].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][q].y, bw[q].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][q].z, bw[q].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][q].w, bw[q].w, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = (tb + u) * 16 + kk * 4 + r;
                float v = acc0[r] + acc1[r] + bj;
                if (do_softmax) {
                    const float m = group_max<16>(j < J ? v : -3.0e38f);
                    const float e = j < J ? expf(v - m) : 0.f;
                    v = e / group_sum<16>(e);
                }
                if (i < rows && j < J) Z[(size_t)i * J + j] = v;
                if (label != nullptr) {                              // first maximum (GPTST.py:344-345) on the values just stored
                    const float vv = j < J ? v : -3.0e38f;
                    const float gm = group_max<16>(vv);
                    const int first = (int)(-group_max<16>(vv == gm ? -(float)j : -3.0e38f));
                    if (i < rows && j == 0) label[i] = first;
                }
            }
        }
    }
}

// out(j,c) += sum_i a'[i,j] X[i,c]   (olayout 0: out[c*J+j], 1: out[j*C+c]);  csum[c] += sum_i X[i,c];  asum[j] += sum_i a'[i,j]
// Two launches: (1) RO_NB workgroups reduce their row chunk to a partial [J*C | C | J] in scratch (4 independent rows in flight per
// thread, slots folded through LDS), (2) a fold kernel sums the RO_NB partials per output and accumulates into the gradients.
// (The single-launch version needed one atomic per output and workgroup: 160..510 same-address atomics = 45..110 us.)
#define RO_NB 512                         // row chunks (= partials) up to 512 K rows (256: 12.1 us, 512: 7.1 us, 1024: 7.8 us at 65 280 rows) ...
#define RO_NB_MAX 2048                    // ... then chunks of 1024 rows, at most this many (N = 4096: 1.5 M rows)
static int ro_rpb(int rows) {
    int nb = RO_NB;
    if (rows > RO_NB * 1024) { nb = (rows + 1023) / 1024; if (nb > RO_NB_MAX) nb = RO_NB_MAX; }
    int rpb = (rows + nb - 1) / nb;
    return rpb < 16 ? 16 : rpb;
}
template <int C>
__global__ __launch_bounds__(256) void rowouter_part_kernel(const float* __restrict__ a, int lda, const float* __restrict__ mask, float fill,
                                                            const float* __restrict__ X, float* __restrict__ part, int rows, int J,
                                                            int rows_per_block, int want_asum) {
    constexpr int LPR = C / 4, RPB = 256 / LPR;
    __shared__ float4 red[RPB][LPR];
    const int c4 = threadIdx.x % LPR, slot = threadIdx.x / LPR;
    const size_t r0 = (size_t)blockIdx.x * rows_per_block;
    const size_t r1 = min((size_t)rows, r0 + rows_per_block);
    float* mine = part + (size_t)blockIdx.x * (J * C + C + J);
    for (int jb = 0; jb < J || (jb == 0 && J == 0); jb += 8) {
        float4 acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = f4zero();
        float4 cacc = f4zero();
        for (size_t i0 = r0 + slot; i0 < r1; i0 += 4 * RPB) {            // 4 independent rows in flight per thread
            float4 x[4];
            float av[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t i = i0 + (size_t)q * RPB;
                x[q] = i < r1 ? ld4(X + i * C + 4 * c4) : f4zero();
#pragma unroll
                for (int u = 0; u < 8; ++u) av[q][u] = (jb + u < J && i < r1) ? masked_a(a, mask, fill, i, jb + u, lda, J) : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (jb == 0) cacc = f4add(cacc, x[q]);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (jb + u < J) acc[u] = f4fma(av[q][u], x[q], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const bool is_c = (u == 8);
            if (is_c ? (jb != 0) : (jb + u >= J)) continue;       // uniform
            red[slot][c4] = is_c ? cacc : acc[u < 8 ? u : 0];
            __syncthreads();
            if (slot == 0) {
                float4 s = red[0][c4];
                for (int q = 1; q < RPB; ++q) s = f4add(s, red[q][c4]);
                st4(mine + (is_c ? J * C : (jb + u) * C) + 4 * c4, s);
            }
            __syncthreads();
        }
    }
    if (want_asum) {
        for (int j = 0; j < J; ++j) {
            float s = 0.f;
            for (size_t i = r0 + threadIdx.x; i < r1; i += 256) s += masked_a(a, mask, fill, i, j, lda, J);
            s = group_sum<64>(s);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) ((float*)red)[threadIdx.x >> 6] = s;
            __syncthreads();
            if (threadIdx.x == 0) mine[J * C + C + j] = ((float*)red)[0] + ((float*)red)[1] + ((float*)red)[2] + ((float*)red)[3];
        }
    }
}

// 16 lanes per output: each sums nb/16 partials (independent loads), then a 16-lane DPP sum — a serial 256-deep load chain
// per output took 60 us.
template <int C>
__global__ __launch_bounds__(256) void rowouter_fold_kernel(const float* __restrict__ part, int nb, float* __restrict__ out, int olayout,
                                                            float* __restrict__ csum, float* __restrict__ asum, int J) {
    const int tot = J * C + C + J;
    const int e = blockIdx.x * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15;
    float s = 0.f;
    if (e < tot)
        for (int b = q; b < nb; b += 16) s += part[(size_t)b * tot + e];
    s = group_sum<16>(s);
    if (e >= tot || q != 0) return;
    if (e < J * C) {
        const int j = e / C, c = e % C;
        out[olayout ? (size_t)j * C + c : (size_t)c * J + j] += s;
    } else if (e < J * C + C) {
        if (csum) csum[e - J * C] += s;
    } else if (asum) {
        asum[e - J * C - C] += s;
    }
}

extern "C" int gptst_rowouter_ws_floats(int J, int C) { return RO_NB_MAX * (J * C + C + J); }

extern "C" int gptst_lin_in(const float* a, int lda, const float* mask, float fill, const float* W, int wlayout, const float* b,
                            float* Y, int rows, int J, int C, void* stream) {
    if (!a || !W || !Y || J <= 0 || J > SM_MAXJ) return GPTST_EARG;
    hipStream_t st = (hipStream_t)stream;
    if ((C == 64 || C == 128) && J <= 2) {
        const int rpb = 4 * 256 / (C / 4);
        const dim3 g((rows + rpb - 1) / rpb);
        if (C == 64 && J == 1) hipLaunchKernelGGL((lin_in_small_kernel<64, 1>), g, dim3(256), 0, st, a, lda, mask, fill, W, wlayout, b, Y, rows);
        else if (C == 64) hipLaunchKernelGGL((lin_in_small_kernel<64, 2>), g, dim3(256), 0, st, a, lda, mask, fill, W, wlayout, b, Y, rows);
        else if (J == 1) hipLaunchKernelGGL((lin_in_small_kernel<128, 1>), g, dim3(256), 0, st, a, lda, mask, fill, W, wlayout, b, Y, rows);
        else hipLaunchKernelGGL((lin_in_small_kernel<128, 2>), g, dim3(256), 0, st, a, lda, mask, fill, W, wlayout, b, Y, rows);
    }
    else if (C == 64) hipLaunchKernelGGL((lin_in_kernel<64>), dim3(sm_grid(rows, 16)), dim3(256), 0, st, a, lda, mask, fill, W, wlayout, b, Y, rows, J);
    else if (C == 128) hipLaunchKernelGGL((lin_in_kernel<128>), dim3(sm_grid(rows, 8)), dim3(256), 0, st, a, lda, mask, fill, W, wlayout, b, Y, rows, J);
    else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_rowdot(const float* X, const float* W, const float* b, float* Z, int rows, int J, int C, int do_softmax,
                            int* label, void* stream) {
    if (!X || !W || !Z || J <= 0 || J > SM_MAXJ) return GPTST_EARG;
    hipStream_t st = (hipStream_t)stream;
    const bool loop = (long)rows > (long)SM_MAXGRID * (C == 64 ? 16 : 8);
    if ((C == 64 || C == 128) && J <= 16) {                         // r05: C = 128 too (the VALU form ran 3.5x its memory floor at N = 512 / 4096)
        const int ntiles = (rows + 15) / 16;
        int tpw = (ntiles + 4 * 1024 - 1) / (4 * 1024);            // ~1024 workgroups of 4 waves; an even number of tiles per wave
        tpw = (tpw + 1) & ~1;
        const dim3 grid((ntiles + 4 * tpw - 1) / (4 * tpw));
        if (C == 64) hipLaunchKernelGGL(rowdot_mfma_kernel<64>, grid, dim3(256), 0, st, X, W, b, Z, rows, J, do_softmax, label, tpw);
        else hipLaunchKernelGGL(rowdot_mfma_kernel<128>, grid, dim3(256), 0, st, X, W, b, Z, rows, J, do_softmax, label, tpw);
    }
    else if (C == 64 && !loop) hipLaunchKernelGGL((rowdot_kernel<64, false>), dim3(sm_grid(rows, 16)), dim3(256), 0, st, X, W, b, Z, rows, J, do_softmax, label);
    else if (C == 64) hipLaunchKernelGGL((rowdot_kernel<64, true>), dim3(sm_grid(rows, 16)), dim3(256), 0, st, X, W, b, Z, rows, J, do_softmax, label);
    else if (C == 128 && !loop) hipLaunchKernelGGL((rowdot_kernel<128, false>), dim3(sm_grid(rows, 8)), dim3(256), 0, st, X, W, b, Z, rows, J, do_softmax, label);
    else if (C == 128) hipLaunchKernelGGL((rowdot_kernel<128, true>), dim3(sm_grid(rows, 8)), dim3(256), 0, st, X, W, b, Z, rows, J, do_softmax, label);
    else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// first stage only: part (gptst_rowouter_nparts(rows), J*C + C + J) row-chunk partials [sum a'^T X | column sums of X | sums of a'];
// the caller folds them (e.g. one kind-1 pool job per target, next to the other reductions of the step)
extern "C" int gptst_rowouter_nparts(int rows) {
    const int rpb = ro_rpb(rows);
    return (rows + rpb - 1) / rpb;
}

extern "C" int gptst_rowouter_part(const float* a, int lda, const float* mask, float fill, const float* X, float* part, int want_asum,
                                   int rows, int J, int C, void* stream) {
    if (!X || !part || J < 0 || J > SM_MAXJ || (J > 0 && !a)) return GPTST_EARG;
    const int rpb = ro_rpb(rows);
    const int nb = (rows + rpb - 1) / rpb;
    hipStream_t st = (hipStream_t)stream;
    if (C == 64) hipLaunchKernelGGL((rowouter_part_kernel<64>), dim3(nb), dim3(256), 0, st, a, lda, mask, fill, X, part, rows, J, rpb, want_asum);
    else if (C == 128) hipLaunchKernelGGL((rowouter_part_kernel<128>), dim3(nb), dim3(256), 0, st, a, lda, mask, fill, X, part, rows, J, rpb, want_asum);
    else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// ws: device scratch of gptst_rowouter_ws_floats(J, C) floats
extern "C" int gptst_rowouter(const float* a, int lda, const float* mask, float fill, const float* X, float* out, int olayout,
                              float* csum, float* asum, float* ws, int rows, int J, int C, void* stream) {
    if (!X || !ws || J < 0 || J > SM_MAXJ || (J > 0 && (!a || !out))) return GPTST_EARG;
    const int rpb = ro_rpb(rows);
    const int nb = (rows + rpb - 1) / rpb;
    const int tot = J * C + C + J;
    hipStream_t st = (hipStream_t)stream;
    if (C == 64) {
        hipLaunchKernelGGL((rowouter_part_kernel<64>), dim3(nb), dim3(256), 0, st, a, lda, mask, fill, X, ws, rows, J, rpb, asum != nullptr);
        hipLaunchKernelGGL((rowouter_fold_kernel<64>), dim3((tot + 15) / 16), dim3(256), 0, st, (const float*)ws, nb, out, olayout, csum, asum, J);
    } else if (C == 128) {
        hipLaunchKernelGGL((rowouter_part_kernel<128>), dim3(nb), dim3(256), 0, st, a, lda, mask, fill, X, ws, rows, J, rpb, asum != nullptr);
        hipLaunchKernelGGL((rowouter_fold_kernel<128>), dim3((tot + 15) / 16), dim3(256), 0, st, (const float*)ws, nb, out, olayout, csum, asum, J);
    } else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
