// hyperTem forward, fused (reference GPTST.py:154-163):   out = LReLU( (G_n X) W_bt + b_bt + X )
//   one workgroup = (sample b, 16 consecutive nodes): the 12 x 16 x C slab of X is loaded ONCE into LDS; each of the 4 waves takes
//   time steps t = w, w+4, w+8:  (1) R_t[n,:] = sum_u G_n[t,u] X_u[n,:]  on the VALU from LDS (the per-node temporal hypergraph,
//   no nonlinearity between gather and scatter => one 12x12 matrix per node), written to HBM for the weight gradient;
//   (2) R_t (16 x C) @ W_bt (C x C) on fp32 MFMA 16x16x4 — the time-conditioned weight is read straight from L2 as the B operand
//   (it is shared by the 11 node tiles of the sample, so no LDS copy and no barrier); (3) bias + residual (from the LDS slab) +
//   LeakyReLU, stored as whole rows.  Replaces tmix_kernel + apply_kernel<TIME> (11 + 20 us -> one launch) and the R round trip.
#include "mfma_tile.h"

#define HT_T 12
__device__ long long g_ht_ts[64];
int g_ht_dbg = 0;
extern "C" int gptst_ht_dbg(int v) { g_ht_dbg = v; return 0; }
extern "C" int gptst_ht_ts(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ht_ts), sizeof(long long) * 64); }

// XCD-aware work map: workgroup L runs on XCD L % 8 (observed dispatch order), and all node tiles of one sample should share an
// XCD so that the sample's twelve W_bt matrices (192 KB) are fetched into ONE L2 instead of eight (PMC: 67 MB -> expected ~23 MB
// of fabric reads per launch).  L -> xcd = L % 8, slot = L / 8;  sample = xcd + 8 * (slot / ntiles), tile = slot % ntiles.
__device__ __forceinline__ bool ht_work(int ntiles, int B, int& b, int& tile) {
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    b = xcd + 8 * (slot / ntiles);
    tile = slot % ntiles;
    return b < B;
}

template <int C>
__global__ __launch_bounds__(256, 2) void hypertem_fwd_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                              const float* __restrict__ Wbt, const float* __restrict__ bbt,
                                                              float* __restrict__ R_out, float* __restrict__ out, int N, int B, int dbg) {
    constexpr int P = C + 4, LPR = C / 4;
    int tsi = 0;
#define TS() do { if (dbg && blockIdx.x == 59 && threadIdx.x == 0) g_ht_ts[tsi] = clock64(); ++tsi; } while (0)
    TS();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                               // [12][16][P]
    float* Gs = Xs + HT_T * 16 * P;                 // [16][144]
    float* Rt = Gs + 16 * 144;                      // [4 waves][16][P]
    int b, tile;
    if (!ht_work((N + 15) / 16, B, b, tile)) return;
    const int n0 = tile * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < HT_T * 16 * LPR; i += 256) {
        const int t = i / (16 * LPR), rem = i % (16 * LPR), nl = rem / LPR, c4 = rem % LPR;
        const int n = n0 + nl;
        st4(Xs + (t * 16 + nl) * P + 4 * c4, n < N ? ld4(X + (((size_t)b * HT_T + t) * N + n) * C + 4 * c4) : f4zero());
    }
    for (int i = tid; i < 16 * 144; i += 256) Gs[i] = (n0 + i / 144 < N) ? G[(size_t)n0 * 144 + i] : 0.f;
    __syncthreads();
    TS();
    float* rt = Rt + wave * 16 * P;
    const int c4 = lane % LPR, nb = lane / LPR;              // VALU / epilogue mapping: LPR lanes per row
    const int j = lane & 15, kk = lane >> 4;                 // MFMA mapping
    constexpr int ROWS_PER_PASS = 64 / LPR;
    for (int t = wave; t < HT_T; t += 4) {
        const size_t g = (size_t)b * HT_T + t;
        // W_bt fragments for the whole time step are requested from L2 FIRST so their latency hides behind the temporal mix
        const float* W = Wbt + g * C * C;
        float bv[C / 16][4][C / 16];
#pragma unroll
        for (int q = 0; q < C / 16; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < C / 16; ++ct) bv[q][e][ct] = W[(size_t)(16 * q + 4 * kk + e) * C + 16 * ct + j];
        TS();
        // ---- (1) temporal mix ----
#pragma unroll
        for (int i = 0; i < 16 / ROWS_PER_PASS; ++i) {
            const int nl = nb + ROWS_PER_PASS * i;
            float4 acc = f4zero();
            const float* gr = Gs + nl * 144 + t * HT_T;
#pragma unroll
            for (int u = 0; u < HT_T; ++u) acc = f4fma(gr[u], ld4(Xs + (u * 16 + nl) * P + 4 * c4), acc);
            st4(rt + nl * P + 4 * c4, acc);
            if (n0 + nl < N) st4(R_out + (g * N + n0 + nl) * C + 4 * c4, acc);
        }
        TS();
        // ---- (2) R_t @ W_bt on MFMA 16x16x4: A = R_t (LDS), B = W_bt fragments (registers) ----
        f32x4 acc[C / 16];
#pragma unroll
        for (int ct = 0; ct < C / 16; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float4 a4[C / 16];
#pragma unroll
        for (int q = 0; q < C / 16; ++q) a4[q] = ld4(rt + j * P + 16 * q + 4 * kk);
#pragma unroll
        for (int q = 0; q < C / 16; ++q) {
            const float av[4] = {a4[q].x, a4[q].y, a4[q].z, a4[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < C / 16; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e][ct], acc[ct], 0, 0, 0);
        }
        TS();
        // ---- (3) epilogue through the wave's tile: rows of float4 ----
#pragma unroll
        for (int ct = 0; ct < C / 16; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) rt[(kk * 4 + r) * P + 16 * ct + j] = acc[ct][r];
        const float4 b4 = ld4(bbt + g * C + 4 * c4);
#pragma unroll
        for (int i = 0; i < 16 / ROWS_PER_PASS; ++i) {
            const int nl = nb + ROWS_PER_PASS * i;
            if (n0 + nl < N) {
                float4 y = f4add(f4add(ld4(rt + nl * P + 4 * c4), b4), ld4(Xs + (t * 16 + nl) * P + 4 * c4));
                y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
                st4(out + (g * N + n0 + nl) * C + 4 * c4, y);
            }
        }
        TS();
    }
}

extern "C" int gptst_hypertem_fwd(const float* X, const float* G, const float* Wbt, const float* bbt, float* R_out, float* out, int B,
                                  int T, int N, int C, void* stream) {
    if (!X || !G || !Wbt || !bbt || !R_out || !out || T != HT_T) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    const size_t smem = ((size_t)HT_T * 16 * (C + 4) + 16 * 144 + 4 * 16 * (C + 4)) * sizeof(float);
    static int done = 0;
    if (!done) { hipFuncSetAttribute((const void*)hypertem_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); done = 1; }
    hipLaunchKernelGGL((hypertem_fwd_kernel<64>), dim3(8 * ((B + 7) / 8) * ((N + 15) / 16)), dim3(256), smem, (hipStream_t)stream, X, G, Wbt, bbt, R_out, out, N, B, g_ht_dbg);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// =====================================================================================================================
// hyperTem backward, fused (everything except the weight gradient, which is grouped by (b,t) and stays in wgrad64_kernel):
//   dPre = dOut * lrelu'(out)                               (LDS slab, 12 x 16 x C per workgroup = (sample, 16 nodes))
//   dbias[b,t,:] += sum_n dPre                              (column sums of the slab, 11 atomics per address)
//   dR_t = dPre_t W_bt^T                                    MFMA 16x16x4 as  dR_t^T = W_bt dPre_t^T  so that W_bt (L2) is the
//                                                           A operand with coalesced float4 rows; result overwrites the slab
//   dX_u = dPre_u + sum_t G_n[t,u] dR_t                     VALU from the slab (dPre re-read from L2)
//   dG_n[t,u] += sum_c dR_t[n,c] X_u[n,c]                   MFMA 16x16x4 per node, 32 atomics per address (one per sample)
// Replaces apply_kernel<TIME, dPre> + tmix_kernel<bwd> + tmix_dgraph_kernel (19 + 14 + 11 us, and the dR round trip).
// =====================================================================================================================
template <int C>
__global__ __launch_bounds__(256, 2) void hypertem_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                              const float* __restrict__ X, const float* __restrict__ G,
                                                              const float* __restrict__ Wbt, float* __restrict__ dX,
                                                              float* __restrict__ dbias, float* __restrict__ dG, int N, int B) {
    constexpr int P = C + 4, LPR = C / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ds = smem;                               // [12][16][P]  dPre, then dR
    float* Gs = Ds + HT_T * 16 * P;                 // [16][144]
    int b, tile;
    if (!ht_work((N + 15) / 16, B, b, tile)) return;
    const int n0 = tile * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < HT_T * 16 * LPR; i += 256) {
        const int t = i / (16 * LPR), rem = i % (16 * LPR), nl = rem / LPR, c4 = rem % LPR;
        const int n = n0 + nl;
        float4 v = f4zero();
        if (n < N) {
            const size_t off = (((size_t)b * HT_T + t) * N + n) * C + 4 * c4;
            const float4 d = ld4(dOut + off), y = ld4(Y + off);
            v = make_float4(d.x * lrelu_grad_from_out(y.x), d.y * lrelu_grad_from_out(y.y), d.z * lrelu_grad_from_out(y.z),
                            d.w * lrelu_grad_from_out(y.w));
        }
        st4(Ds + (t * 16 + nl) * P + 4 * c4, v);
    }
    for (int i = tid; i < 16 * 144; i += 256) Gs[i] = (n0 + i / 144 < N) ? G[(size_t)n0 * 144 + i] : 0.f;
    __syncthreads();
    const int j = lane & 15, kk = lane >> 4;
    for (int t = wave; t < HT_T; t += 4) {
        const size_t g = (size_t)b * HT_T + t;
        float* dt = Ds + t * 16 * P;
        // bias gradient: column sums of dPre_t over the 16 nodes
        for (int c = lane; c < C; c += 64) {
            float s = 0.f;
#pragma unroll
            for (int nl = 0; nl < 16; ++nl) s += dt[nl * P + c];
            atomicAdd(dbias + g * C + c, s);
        }
        // dR_t^T (C x 16) = W_bt (C x C) dPre_t^T:  A[i][kk=o] = W[i][o] (global float4 rows), B[kk=o][j=n] = dPre_t[n][o] (LDS)
        const float* W = Wbt + g * C * C;
        float4 bq[C / 16];
#pragma unroll
        for (int q = 0; q < C / 16; ++q) bq[q] = ld4(dt + j * P + 16 * q + 4 * kk);
        f32x4 acc[C / 16];
#pragma unroll
        for (int it = 0; it < C / 16; ++it) {
            acc[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float4 aq[C / 16];
#pragma unroll
            for (int q = 0; q < C / 16; ++q) aq[q] = ld4(W + (size_t)(it * 16 + j) * C + 16 * q + 4 * kk);
#pragma unroll
            for (int q = 0; q < C / 16; ++q) {
                acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q].x, bq[q].x, acc[it], 0, 0, 0);
                acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q].y, bq[q].y, acc[it], 0, 0, 0);
                acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q].z, bq[q].z, acc[it], 0, 0, 0);
                acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q].w, bq[q].w, acc[it], 0, 0, 0);
            }
        }
        // D reg r: row i = it*16 + kk*4 + r (input channel), col j = node  ->  dR_t[node][channel] over the slab
#pragma unroll
        for (int it = 0; it < C / 16; ++it)
            st4(dt + j * P + it * 16 + kk * 4, make_float4(acc[it][0], acc[it][1], acc[it][2], acc[it][3]));
    }
    __syncthreads();
    // ---- dX_u[n,:] = dPre_u[n,:] + sum_t G_n[t,u] dR_t[n,:] ----
    {
        const int c4 = tid % LPR, nb = tid / LPR;              // 256 threads = 16 rows x LPR lanes (C = 64)
        for (int nl = nb; nl < 16; nl += 256 / LPR) {
            const int n = n0 + nl;
            if (n < N) {
                float4 dr[HT_T];
#pragma unroll
                for (int t = 0; t < HT_T; ++t) dr[t] = ld4(Ds + (t * 16 + nl) * P + 4 * c4);
#pragma unroll
                for (int u = 0; u < HT_T; ++u) {
                    const size_t off = (((size_t)b * HT_T + u) * N + n) * C + 4 * c4;
                    const float4 d = ld4(dOut + off), y = ld4(Y + off);
                    float4 acc = make_float4(d.x * lrelu_grad_from_out(y.x), d.y * lrelu_grad_from_out(y.y),
                                             d.z * lrelu_grad_from_out(y.z), d.w * lrelu_grad_from_out(y.w));
#pragma unroll
                    for (int t = 0; t < HT_T; ++t) acc = f4fma(Gs[nl * 144 + t * HT_T + u], dr[t], acc);
                    st4(dX + off, acc);
                }
            }
        }
    }
    // ---- dG_n[t,u] += sum_c dR_t[n,c] X_u[n,c]:  A[i=t][kk=c] = dR (LDS), B[kk=c][j=u] = X (global) ----
    for (int nl = wave; nl < 16; nl += 4) {
        const int n = n0 + nl;
        if (n >= N) continue;                                  // wave-uniform
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < C / 16; ++q) {
            float4 a = f4zero(), x = f4zero();
            if (j < HT_T) {
                a = ld4(Ds + (j * 16 + nl) * P + 16 * q + 4 * kk);
                x = ld4(X + (((size_t)b * HT_T + j) * N + n) * C + 16 * q + 4 * kk);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x.w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = kk * 4 + r, u = j;
            if (t < HT_T && u < HT_T) atomicAdd(dG + (size_t)n * 144 + t * HT_T + u, acc[r]);
        }
    }
}

// dbias (B*T, C) and dG (N, T, T) are ACCUMULATED (+=): zero them first.
extern "C" int gptst_hypertem_bwd(const float* dOut, const float* Y, const float* X, const float* G, const float* Wbt, float* dX,
                                  float* dbias, float* dG, int B, int T, int N, int C, void* stream) {
    if (!dOut || !Y || !X || !G || !Wbt || !dX || !dbias || !dG || T != HT_T) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    const size_t smem = ((size_t)HT_T * 16 * (C + 4) + 16 * 144) * sizeof(float);
    static int done = 0;
    if (!done) { hipFuncSetAttribute((const void*)hypertem_bwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); done = 1; }
    hipLaunchKernelGGL((hypertem_bwd_kernel<64>), dim3(8 * ((B + 7) / 8) * ((N + 15) / 16)), dim3(256), smem, (hipStream_t)stream, dOut, Y, X, G, Wbt, dX, dbias, dG, N, B);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
