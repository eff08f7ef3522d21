// Per-node adaptive temporal hypergraph of hyperTem (reference GPTST.py:156-158):
//     hyper = einsum('htn,btnd->bhnd', A, X);   ret = einsum('thn,bhnd->btnd', A^T, hyper)
// There is no nonlinearity between the gather (T -> Hm hyperedges) and the scatter (Hm -> T), so per node
//     ret[b,:,n,:] = G_n  X[b,:,n,:],     G_n = A_n^T A_n   (T x T, symmetric),  A_n = (E_node . adj)[n]  (Hm x T)
// Kernels:  gram_fwd  (A -> G),  tmix (G (*) X, also the data-gradient because G is symmetric),
//           tmix_dgraph (dG_n = sum_b dR[b,:,n,:] X[b,:,n,:]^T on fp32 MFMA 16x16x4),  gram_bwd (dG -> dA).
// T is fixed at 12 by the reference (GPTST.py:97,208-209).
#include "common.h"

#define TT 12
#define TT2 144

// ---- G[n] = A[n]^T A[n];  A: (N, Hm, T) ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_fwd_kernel(const float* __restrict__ A, float* __restrict__ G, int N, int Hm) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * TT2) return;
    const int n = idx / TT2, t = (idx % TT2) / TT, u = idx % TT;
    const float* a = A + (size_t)n * Hm * TT;
    float s = 0.f;
    for (int h = 0; h < Hm; ++h) s = fmaf(a[h * TT + t], a[h * TT + u], s);
    G[idx] = s;
}

// ---- dA[n,h,t] = sum_u A[n,h,u] (dG[n,t,u] + dG[n,u,t]) -----------------------------------------------------
// one workgroup per (layer, node): the nsplit partial graphs dG[l][s][n] are summed in a fixed order into LDS first
__global__ __launch_bounds__(256) void gram_bwd_kernel(const float* __restrict__ A, const float* __restrict__ dG,
                                                       float* __restrict__ dA, int N, int Hm, int nsplit) {
    __shared__ float gs[TT2];
    const int gn = blockIdx.x, l = gn / N, n = gn % N;
    if (threadIdx.x < TT2) {
        const float* g = dG + ((size_t)l * nsplit * N + n) * TT2 + threadIdx.x;
        // 8 independent loads in flight, summed in a FIXED order (partial k holds splits k, k+8, ...): a plain loop is a chain of
        // nsplit dependent round trips (17.8 us for 32 per-sample partials)
        float part[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) part[k] = 0.f;
        for (int sp0 = 0; sp0 < nsplit; sp0 += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = sp0 + k < nsplit ? g[(size_t)(sp0 + k) * N * TT2] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) part[k] += v[k];
        }
        gs[threadIdx.x] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    }
    __syncthreads();
    if (threadIdx.x >= Hm * TT) return;
    const int h = threadIdx.x / TT, t = threadIdx.x % TT;
    const float* a = A + ((size_t)gn * Hm + h) * TT;
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < TT; ++u) s = fmaf(a[u], gs[t * TT + u] + gs[u * TT + t], s);
    dA[((size_t)gn * Hm + h) * TT + t] = s;
}

// ---- out[b,t,n,:] = sum_u G[n,t,u] X[b,u,n,:]  (+ dOut * lrelu'(Y) when ADD_DPRE: the residual branch of backward) --
// block = (256 / (C/4)) nodes x (C/4) float4 lanes, one batch sample; every thread keeps its 12 x float4 column in registers.
template <int C, bool ADD_DPRE>
__global__ __launch_bounds__(256) void tmix_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                   const float* __restrict__ dOut, const float* __restrict__ Y,
                                                   float* __restrict__ out, int N) {
    constexpr int LPR = C / 4;              // lanes per row
    constexpr int NPB = 256 / LPR;          // nodes per block
    __shared__ float Gs[NPB][TT2];
    const int b = blockIdx.y, n0 = blockIdx.x * NPB;
    for (int i = threadIdx.x; i < NPB * TT2; i += 256) {
        const int nl = i / TT2;
        Gs[nl][i % TT2] = (n0 + nl < N) ? G[(size_t)(n0 + nl) * TT2 + i % TT2] : 0.f;
    }
    __syncthreads();
    const int nl = threadIdx.x / LPR, c4 = threadIdx.x % LPR;
    const int n = n0 + nl;
    if (n >= N) return;
    const size_t base = ((size_t)b * TT * N + n) * C + 4 * c4;     // element (b, 0, n, 4*c4)
    const size_t tstride = (size_t)N * C;
    float4 x[TT];
#pragma unroll
    for (int u = 0; u < TT; ++u) x[u] = ld4(X + base + u * tstride);
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        float4 acc = f4zero();
        if (ADD_DPRE) {
            const float4 d = ld4(dOut + base + t * tstride), y = ld4(Y + base + t * tstride);
            acc = make_float4(d.x * lrelu_grad_from_out(y.x), d.y * lrelu_grad_from_out(y.y),
                              d.z * lrelu_grad_from_out(y.z), d.w * lrelu_grad_from_out(y.w));
        }
#pragma unroll
        for (int u = 0; u < TT; ++u) acc = f4fma(Gs[nl][t * TT + u], x[u], acc);
        st4(out + base + t * tstride, acc);
    }
}

// ---- dG[n,t,u] = sum_{b,c} dR[b,t,n,c] X[b,u,n,c] --------------------------------------------------------------
// MFMA 16x16x4 f32: A-op lane l holds A[i=l&15][k=l>>4], B-op B[k=l>>4][j=l&15]; D reg r: row (l>>4)*4+r, col l&15.
// i = t, j = u (12 of 16 used), k walks the channels: lane group kk = l>>4 loads float4 c = 16q + 4kk .. +3 and feeds
// MFMA steps 4q..4q+3 (the same k-permutation on both operands).  One block per node, waves split the batch.
template <int C>
__global__ __launch_bounds__(256) void tmix_dgraph_kernel(const float* __restrict__ dR, const float* __restrict__ X,
                                                          float* __restrict__ dG, int B, int N) {
    __shared__ float red[4][TT2];
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t tstride = (size_t)N * C;
    for (int b = wave; b < B; b += 4) {
        const size_t base = ((size_t)b * TT * N + n) * C + 4 * kk;
        float4 a[C / 16], x[C / 16];
#pragma unroll
        for (int q = 0; q < C / 16; ++q) {
            a[q] = f4zero(); x[q] = f4zero();
            if (i < TT) {
                a[q] = ld4(dR + base + i * tstride + 16 * q);
                x[q] = ld4(X + base + i * tstride + 16 * q);
            }
        }
#pragma unroll
        for (int q = 0; q < C / 16; ++q) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, x[q].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, x[q].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, x[q].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, x[q].w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = kk * 4 + r, u = i;
        if (t < TT && u < TT) red[wave][t * TT + u] = acc[r];
    }
    __syncthreads();
    if (threadIdx.x < TT2) {
        const int e = threadIdx.x;
        dG[(size_t)n * TT2 + e] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    }
}

// ---- backward of the temporal mixing in ONE pass over dR (C = 128 path; the C = 64 path has it inside hypertem_bwd) ---------------
//   dX[b,u,n,:] = dOut[b,u,n,:] * lrelu'(Y[b,u,n,:]) + sum_t G[n,t,u] dR[b,t,n,:]        (tmix_kernel<C, true>)
//   dG[n,t,u]   = sum_{b,c} dR[b,t,n,c] X[b,u,n,c]                                         (tmix_dgraph_kernel<C>)
// One workgroup per node, the waves split the batch.  Both are MFMA 16x16x4 products on 12-of-16 padded time indices:
//   dX^T-free form  D[i=u][col=channel] = sum_t G[t,u] dR[t,channel]:  A lane (i,kk) = G[t=4kk+s][u=i] (k-step s), B lane (j,kk) =
//   dR[t=4kk+s][64hf+4j+e] — rows 4kk+s of the node's (T, C) block as 256-byte coalesced float4 loads, the same rows 4kk+r the lane's
//   accumulators stand for, so dOut / Y of the residual branch are loaded with the same addressing and the epilogue is register-local;
//   dG  D[i=t][j=u] = sum_c dR[t,c] X[u,c]:  lane (i,kk) holds row t=i (u=i), channels 16q+4kk..+3 of dR and of X.
// dR and X change layout through a wave-private LDS tile: HBM traffic dR + X + dOut + Y + dX instead of 2 dR + X + dOut + Y + dX
// in two launches (N = 4096, C = 128, B = 32: 595 + 482 us -> one launch).
// CH (r05, the dPre chain at C = 128): 0 = dOut with the layer's output Y for the sign; 1 = the incoming gradient already is dPre (Y is not read:
// one activation-sized read less); 2 = as 1, and dX is returned multiplied by lrelu'(X) — X is in registers for the graph gradient anyway.
template <int C, int CH = 0>
__global__ __launch_bounds__(256, 2) void tmix_bwd_dgraph_kernel(const float* __restrict__ dR, const float* __restrict__ X,
                                                                 const float* __restrict__ G, const float* __restrict__ dOut,
                                                                 const float* __restrict__ Y, float* __restrict__ dX,
                                                                 float* __restrict__ dG, int B, int N) {
    constexpr int Q = C / 16, H2 = C / 64, P = C + 4;
    __shared__ float red[4][TT2];
    __shared__ __attribute__((aligned(16))) float tile[4][2][TT][P];     // per wave: the (T, C) blocks of dR and X, to change operand layout
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const size_t tstride = (size_t)N * C;
    float ga[4];                                              // A operand of the dX product: G[n][t = 4kk+s][u = j]
#pragma unroll
    for (int s = 0; s < 4; ++s) ga[s] = (j < TT && 4 * kk + s < TT) ? G[(size_t)n * TT2 + (4 * kk + s) * TT + j] : 0.f;
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, g2 = g0, g3 = g0;     // dG accumulators (four chains), summed over this wave's samples
    for (int b = wave; b < B; b += 4) {
        const size_t base = ((size_t)b * TT * N + n) * C;
        // every global load in column form (time steps 4kk+s, channels 64hf+4j..: whole 256-byte row pieces); the row form the dG
        // product needs (time step j, channels 16q+4kk..) is read back from the wave's LDS tile (row-form loads straight from global
        // memory are 64-byte pieces of 12 rows 2 MB apart: 930 us per launch at N = 4096, B = 32 with them)
        float4 drc[4][H2], xc[4][H2], doc[4][H2], yc[4][H2];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int hf = 0; hf < H2; ++hf) {
                drc[s][hf] = f4zero(); xc[s][hf] = f4zero(); doc[s][hf] = f4zero(); yc[s][hf] = f4zero();
                if (kk < 3) {
                    const size_t o = base + (4 * kk + s) * tstride + 64 * hf + 4 * j;
                    drc[s][hf] = ld4(dR + o); xc[s][hf] = ld4(X + o); doc[s][hf] = ld4(dOut + o);
                    if (CH == 0) yc[s][hf] = ld4(Y + o);
                }
            }
        SB();
        if (kk < 3) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int hf = 0; hf < H2; ++hf) {
                    st4(&tile[wave][0][4 * kk + s][64 * hf + 4 * j], drc[s][hf]);
                    st4(&tile[wave][1][4 * kk + s][64 * hf + 4 * j], xc[s][hf]);
                }
        }
        SB();
        float4 drr[Q], xr[Q];                                 // row form: time step j, channels 16q+4kk..  (wave-private tile: program order suffices)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            drr[q] = f4zero(); xr[q] = f4zero();
            if (j < TT) { drr[q] = ld4(&tile[wave][0][j][16 * q + 4 * kk]); xr[q] = ld4(&tile[wave][1][j][16 * q + 4 * kk]); }
        }
        SB();
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(drr[q].x, xr[q].x, g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(drr[q].y, xr[q].y, g1, 0, 0, 0);
            g2 = __builtin_amdgcn_mfma_f32_16x16x4f32(drr[q].z, xr[q].z, g2, 0, 0, 0);
            g3 = __builtin_amdgcn_mfma_f32_16x16x4f32(drr[q].w, xr[q].w, g3, 0, 0, 0);
        }
        f32x4 acc[4 * H2];
#pragma unroll
        for (int ct = 0; ct < 4 * H2; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int hf = 0; hf < H2; ++hf) {
                acc[4 * hf + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], drc[s][hf].x, acc[4 * hf + 0], 0, 0, 0);
                acc[4 * hf + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], drc[s][hf].y, acc[4 * hf + 1], 0, 0, 0);
                acc[4 * hf + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], drc[s][hf].z, acc[4 * hf + 2], 0, 0, 0);
                acc[4 * hf + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], drc[s][hf].w, acc[4 * hf + 3], 0, 0, 0);
            }
        SB();
        if (kk < 3) {                                         // rows u = 4kk+r of dX: residual branch from registers, coalesced stores
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int hf = 0; hf < H2; ++hf) {
                    const float4 d = doc[r][hf];
                    float4 o4;
                    if (CH == 0) {
                        const float4 y = yc[r][hf];
                        o4 = make_float4(fmaf(d.x, lrelu_grad_from_out(y.x), acc[4 * hf + 0][r]), fmaf(d.y, lrelu_grad_from_out(y.y), acc[4 * hf + 1][r]),
                                         fmaf(d.z, lrelu_grad_from_out(y.z), acc[4 * hf + 2][r]), fmaf(d.w, lrelu_grad_from_out(y.w), acc[4 * hf + 3][r]));
                    } else {
                        o4 = make_float4(d.x + acc[4 * hf + 0][r], d.y + acc[4 * hf + 1][r], d.z + acc[4 * hf + 2][r], d.w + acc[4 * hf + 3][r]);
                        if (CH == 2) {
                            const float4 x = xc[r][hf];
                            o4.x *= lrelu_grad_from_out(x.x); o4.y *= lrelu_grad_from_out(x.y); o4.z *= lrelu_grad_from_out(x.z); o4.w *= lrelu_grad_from_out(x.w);
                        }
                    }
                    st4(dX + base + (4 * kk + r) * tstride + 64 * hf + 4 * j, o4);
                }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = kk * 4 + r, u = j;
        if (t < TT && u < TT) red[wave][t * TT + u] = (g0[r] + g1[r]) + (g2[r] + g3[r]);
    }
    __syncthreads();
    if (threadIdx.x < TT2) {
        const int e = threadIdx.x;
        dG[(size_t)n * TT2 + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    }
}

extern "C" int gptst_gram_fwd(const float* A, float* G, int N, int Hm, void* stream) {
    if (!A || !G) return GPTST_EARG;
    hipLaunchKernelGGL(gram_fwd_kernel, dim3((N * TT2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, A, G, N, Hm);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// A (L*N, Hm, T), dG (L, nsplit, N, T, T) partial graph gradients (summed here in a fixed order), dA (L*N, Hm, T)
extern "C" int gptst_gram_bwd(const float* A, const float* dG, float* dA, int L, int N, int Hm, int nsplit, void* stream) {
    if (!A || !dG || !dA || L <= 0 || N <= 0 || nsplit <= 0 || Hm * TT > 256) return GPTST_EARG;
    hipLaunchKernelGGL(gram_bwd_kernel, dim3(L * N), dim3(256), 0, (hipStream_t)stream, A, dG, dA, N, Hm, nsplit);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// the dPre-chain form (r05): dX = (dPre + G (*) dR) [* lrelu'(X) when premul], dG as above; the layer's output is not read
extern "C" int gptst_tmix_bwd_chain(const float* dR, const float* X, const float* G, const float* dPre, int premul, float* dX, float* dG,
                                    int B, int T, int N, int C, void* stream) {
    if (!dR || !X || !G || !dPre || !dX || !dG || T != TT || B < 1 || N < 1) return GPTST_EARG;
    hipStream_t st = (hipStream_t)stream;
    if (C == 64 && premul) hipLaunchKernelGGL((tmix_bwd_dgraph_kernel<64, 2>), dim3(N), dim3(256), 0, st, dR, X, G, dPre, nullptr, dX, dG, B, N);
    else if (C == 64) hipLaunchKernelGGL((tmix_bwd_dgraph_kernel<64, 1>), dim3(N), dim3(256), 0, st, dR, X, G, dPre, nullptr, dX, dG, B, N);
    else if (C == 128 && premul) hipLaunchKernelGGL((tmix_bwd_dgraph_kernel<128, 2>), dim3(N), dim3(256), 0, st, dR, X, G, dPre, nullptr, dX, dG, B, N);
    else if (C == 128) hipLaunchKernelGGL((tmix_bwd_dgraph_kernel<128, 1>), dim3(N), dim3(256), 0, st, dR, X, G, dPre, nullptr, dX, dG, B, N);
    else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// out = G (*) X  [+ dOut * lrelu'(Y) if dOut != NULL]
extern "C" int gptst_tmix(const float* X, const float* G, const float* dOut, const float* Y, float* out, int B, int T, int N,
                          int C, void* stream) {
    if (!X || !G || !out || T != TT) return GPTST_EARG;
    if (dOut && !Y) return GPTST_EARG;
    hipStream_t st = (hipStream_t)stream;
    if (C == 64) {
        dim3 grid((N + 15) / 16, B);
        if (dOut) hipLaunchKernelGGL((tmix_kernel<64, true>), grid, dim3(256), 0, st, X, G, dOut, Y, out, N);
        else hipLaunchKernelGGL((tmix_kernel<64, false>), grid, dim3(256), 0, st, X, G, dOut, Y, out, N);
    } else if (C == 128) {
        dim3 grid((N + 7) / 8, B);
        if (dOut) hipLaunchKernelGGL((tmix_kernel<128, true>), grid, dim3(256), 0, st, X, G, dOut, Y, out, N);
        else hipLaunchKernelGGL((tmix_kernel<128, false>), grid, dim3(256), 0, st, X, G, dOut, Y, out, N);
    } else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_tmix_dgraph(const float* dR, const float* X, float* dG, int B, int T, int N, int C, void* stream) {
    if (!dR || !X || !dG || T != TT) return GPTST_EARG;
    hipStream_t st = (hipStream_t)stream;
    if (C == 64) hipLaunchKernelGGL((tmix_dgraph_kernel<64>), dim3(N), dim3(256), 0, st, dR, X, dG, B, N);
    else if (C == 128) hipLaunchKernelGGL((tmix_dgraph_kernel<128>), dim3(N), dim3(256), 0, st, dR, X, dG, B, N);
    else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// dX (B,T,N,C) = dOut * lrelu'(Y) + G^T (*) dR  and  dG (N,T,T) = sum_b dR X^T  in one pass (overwrites both)
extern "C" int gptst_tmix_bwd(const float* dR, const float* X, const float* G, const float* dOut, const float* Y, float* dX, float* dG,
                              int B, int T, int N, int C, void* stream) {
    if (!dR || !X || !G || !dOut || !Y || !dX || !dG || T != TT || B < 1 || N < 1) return GPTST_EARG;
    hipStream_t st = (hipStream_t)stream;
    if (C == 64) hipLaunchKernelGGL((tmix_bwd_dgraph_kernel<64>), dim3(N), dim3(256), 0, st, dR, X, G, dOut, Y, dX, dG, B, N);
    else if (C == 128) hipLaunchKernelGGL((tmix_bwd_dgraph_kernel<128>), dim3(N), dim3(256), 0, st, dR, X, G, dOut, Y, dX, dG, B, N);
    else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
