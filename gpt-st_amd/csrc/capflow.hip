// cap for node counts whose (b,t) capsule matrix does not fit LDS (BASELINE configs[4]: N = 4096, C = 128) — second generation of the
// streaming path (cap_big.hip is the first: VALU, one wave per row, one kernel per algebraic step, 9 passes over the capsule matrix P
// per forward).  Same algebra (reference GPTST.py:102-123,135):
//     P = squash(X Wp^T + bp);  c0 = softmax_h(dadj);  v0 = squash(c0 P)
//     b = 0;  R x { [b += v P^T];  cs = softmax_h(b);  v = squash(v0 (.) cs P) };  b += v P^T;  c = softmax_h(b + dadj);  s = c P
// Every routing iteration is ONE pass over P: for a 16-node tile the wave computes the logits update  rows . V^T  (fp32 MFMA 16x16x4,
// contraction over channels), the softmax over the clusters (the 16 lanes of a DPP row), and the tile's contribution  cs^T . rows  to
// the cluster sums (MFMA, contraction over the tile's nodes) — the three kernels type2 / softmax / type1 of cap_big.hip, with P read
// once instead of twice and the (BT,HS,N) coefficient matrix never written for the inner iterations.  The first iteration (b = 0: uniform
// coefficients, i.e. column sums of P) is folded into the squash pass.  Per forward: Y written + read once, P written once and read R times.
// Sums over nodes leave a kernel as PARTIALS per node chunk (64 - 256 nodes), folded in index order by cf_post_kernel (no atomics, no zero fill);
// a node-sharded run folds, all-reduces the (BT,HS,C) sums across ranks and resumes with the post step.
//
// Operand layouts of a 16-row tile (lane = (j, kk), j = lane & 15, kk = lane >> 4):
//   "row form"     lane holds row j, channels 16q+4kk .. +3 (q < C/16):  A operand of a contraction over channels; D of a product whose
//                  output rows are channels (rec^T, dP^T) lands in the same form, so row-wise epilogues run from registers;
//   "column form"  lane holds rows 4kk+r (r < 4), channels 64hf+4j .. +3 (hf < C/64):  B operand of a contraction over the tile's rows
//                  (k-step r <-> row 4kk+r), and 256-byte coalesced loads / stores;
//   "node form" of a (BT,HS,N) matrix: lane holds cluster j, nodes 4kk .. 4kk+3 of the tile — one float4 along N; it is the D layout
//                  of  rows . V^T  and the A layout of  cs^T . rows.
// HS <= 16 at C in {64, 128} and HS <= 64 at C = 64 (NHT = ceil(HS/16) cluster tiles: cluster h = 16*ht + j; BASELINE configs[3] sweeps
// HS up to 40); other shapes stay on cap_big.hip.
#include "common.h"

// 16-row tiles per wave: 4 for long node ranges (N = 4096: 16 partials per (b,t)), fewer when a (b,t) has only a few tiles, so that the
// tile loop of a wave — one dependent chain of loads, MFMAs and cross-lane steps per tile — stays short and more workgroups share the work
static inline int cf_tpw(int N) { const int nt = (N + 15) / 16; return nt >= 128 ? 4 : (nt >= 32 ? 2 : 1); }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 fzero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }

// four consecutive node entries n .. n+3 of one (.., N) row (`row` points at node 0); `al` = rows are 16-byte aligned (N % 4 == 0)
__device__ __forceinline__ float4 ldn4(const float* __restrict__ row, int n, int N, bool al) {
    float4 v = f4zero();
    if (al) { if (n < N) v = ld4(row + n); }
    else {
        if (n < N) v.x = row[n];
        if (n + 1 < N) v.y = row[n + 1];
        if (n + 2 < N) v.z = row[n + 2];
        if (n + 3 < N) v.w = row[n + 3];
    }
    return v;
}
__device__ __forceinline__ void stn4(float* __restrict__ row, int n, int N, bool al, float4 v) {
    if (al) { if (n < N) st4(row + n, v); }
    else {
        if (n < N) row[n] = v.x;
        if (n + 1 < N) row[n + 1] = v.y;
        if (n + 2 < N) row[n + 2] = v.z;
        if (n + 3 < N) row[n + 3] = v.w;
    }
}

// softmax over the clusters h = 16*ht + j < HS (the lanes of a DPP row x the NHT cluster tiles) of the four node columns of a lane;
// invalid nodes -> 0
template <int NHT>
__device__ __forceinline__ void softmax_h4(const float4 (&x)[NHT], float (&cs)[NHT][4], int j, int HS, int n, int N) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float xv[NHT], m = -3.0e38f;
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) {
            const float v = r == 0 ? x[ht].x : (r == 1 ? x[ht].y : (r == 2 ? x[ht].z : x[ht].w));
            xv[ht] = 16 * ht + j < HS ? v : -3.0e38f;
            m = fmaxf(m, xv[ht]);
        }
        m = group_max<16>(m);
        float e[NHT], sum = 0.f;
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) { e[ht] = 16 * ht + j < HS ? __expf(xv[ht] - m) : 0.f; sum += e[ht]; }
        const float inv = 1.f / group_sum<16>(sum);
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) cs[ht][r] = (n + r < N) ? e[ht] * inv : 0.f;
    }
}

// fold the per-wave cluster sums (acc[ht][4hf+e][r'] = S[h = 16ht+4kk+r'][channel 64hf+4j+e]) through LDS, one cluster tile at a time, and
// store the workgroup's partial (rows h < HS; XR: plus row HS from the extra LDS row 16 — the column sums)
template <int C, int NHT, int XR>
__device__ __forceinline__ void store_partial(float4 (*red)[16 + XR][C / 4], const f32x4 (&acc)[NHT][C / 16], float* __restrict__ dst, int HS,
                                              int wave, int j, int kk) {
#pragma unroll
    for (int ht = 0; ht < NHT; ++ht) {
        if (ht) __syncthreads();                                      // the previous tile has been read
#pragma unroll
        for (int hf = 0; hf < C / 64; ++hf)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[wave][4 * kk + r][16 * hf + j] =
                    make_float4(acc[ht][4 * hf + 0][r], acc[ht][4 * hf + 1][r], acc[ht][4 * hf + 2][r], acc[ht][4 * hf + 3][r]);
        __syncthreads();
        const int nrow = (XR && ht == 0) ? 17 : 16;
        for (int o = threadIdx.x; o < nrow * (C / 4); o += 256) {
            const int row = o / (C / 4), c4 = o % (C / 4);
            const int h = row == 16 ? HS : 16 * ht + row;
            if (row < 16 && h >= HS) continue;
            const float4 s = f4add(f4add(red[0][row][c4], red[1][row][c4]), f4add(red[2][row][c4], red[3][row][c4]));
            st4(dst + (size_t)h * C + 4 * c4, s);
        }
    }
}

// ---- pass 0:  P = squash(Y) (row-wise), c0 = softmax_h(dadj), partial [c0^T P ; colsum P] ------------------------------------------
// part: (BT, nparts, HS+1, C) — rows h < HS: sum_n c0[h,n] P[n,:];  row HS: sum_n P[n,:]  (the first routing iteration: uniform coefficients)
template <int C, int NHT>
__global__ __launch_bounds__(256) void cf_squash_kernel(const float* __restrict__ Y, const float* __restrict__ dadj, float* __restrict__ P,
                                                        float* __restrict__ part, int HS, int N, int nparts, int tpw) {
    constexpr int H2 = C / 64;
    __shared__ float4 red[4][17][C / 4];
    const int bt = blockIdx.y, chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    const bool al = (N & 3) == 0;
    f32x4 acc[NHT][C / 16];
#pragma unroll
    for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
        for (int i = 0; i < C / 16; ++i) acc[ht][i] = fzero4();
    float4 csum[H2];
#pragma unroll
    for (int hf = 0; hf < H2; ++hf) csum[hf] = f4zero();
#pragma unroll 1
    for (int it = 0; it < tpw; ++it) {
        const int n0 = (chunk * tpw + it) * 64 + wave * 16;
        if (n0 >= N) break;                                           // wave-uniform
        const int n = n0 + 4 * kk;
        float4 y[4][H2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int hf = 0; hf < H2; ++hf) y[r][hf] = ld4(Y + ((size_t)bt * N + min(n + r, N - 1)) * C + 64 * hf + 4 * j);
        float4 lg[NHT];
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) lg[ht] = ldn4(dadj + ((size_t)bt * HS + min(16 * ht + j, HS - 1)) * N, n, N, al);
        SB();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float q = 0.f;
#pragma unroll
            for (int hf = 0; hf < H2; ++hf) q += f4dot(y[r][hf], y[r][hf]);
            const float sc = (n + r < N) ? squash_scale(group_sum<16>(q)) : 0.f;
#pragma unroll
            for (int hf = 0; hf < H2; ++hf) {
                y[r][hf] = make_float4(y[r][hf].x * sc, y[r][hf].y * sc, y[r][hf].z * sc, y[r][hf].w * sc);
                csum[hf] = f4add(csum[hf], y[r][hf]);
                if (n + r < N) st4(P + ((size_t)bt * N + n + r) * C + 64 * hf + 4 * j, y[r][hf]);
            }
        }
        float cs[NHT][4];
        softmax_h4<NHT>(lg, cs, j, HS, n, N);
        SB();
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int hf = 0; hf < H2; ++hf) {
                    acc[ht][4 * hf + 0] = mfma4(cs[ht][r], y[r][hf].x, acc[ht][4 * hf + 0]);
                    acc[ht][4 * hf + 1] = mfma4(cs[ht][r], y[r][hf].y, acc[ht][4 * hf + 1]);
                    acc[ht][4 * hf + 2] = mfma4(cs[ht][r], y[r][hf].z, acc[ht][4 * hf + 2]);
                    acc[ht][4 * hf + 3] = mfma4(cs[ht][r], y[r][hf].w, acc[ht][4 * hf + 3]);
                }
    }
#pragma unroll
    for (int hf = 0; hf < H2; ++hf) {                                 // column sums: fold the four row groups of the wave
        float4 s = csum[hf];
        s.x += __shfl_xor(s.x, 16, 64); s.y += __shfl_xor(s.y, 16, 64); s.z += __shfl_xor(s.z, 16, 64); s.w += __shfl_xor(s.w, 16, 64);
        s.x += __shfl_xor(s.x, 32, 64); s.y += __shfl_xor(s.y, 32, 64); s.z += __shfl_xor(s.z, 32, 64); s.w += __shfl_xor(s.w, 32, 64);
        if (kk == 0) red[wave][16][16 * hf + j] = s;
    }
    store_partial<C, NHT, 1>(red, acc, part + ((size_t)bt * nparts + chunk) * (HS + 1) * C, HS, wave, j, kk);
}

// ---- one pass over the rows of a (BT,N,C) matrix ------------------------------------------------------------------------------------
//   L = V ? rows . V^T : 0;   b = L + (bl_in ? bl_in : 0);   bl_out <- b (if given)
//   cs = c_in ? c_in : softmax_h(b + (l0 ? l0 : 0));   c_out <- cs (if given);   part <- partial of cs^T . rows
// routing iteration r >= 1:  (P, v, b -> b);   last step:  (P, v, b, l0 = dadj -> c);   backward of rec = c^T v:  (drec, v -> dc1 = bl_out; c_in = c -> dv)
#ifndef CF_ROUTE_OCC
#define CF_ROUTE_OCC 2      // 3 (168 VGPRs, 34 spilled at C = 128) measured slower: configs[4] 23.2 -> 22.9, N = 512 share 154.5 -> 151.0
#endif
template <int C, int NHT>
__global__ __launch_bounds__(256, (NHT == 1 ? CF_ROUTE_OCC : 2)) void cf_route_kernel(const float* __restrict__ rows, const float* __restrict__ V,
                                                                         const float* __restrict__ bl_in, float* __restrict__ bl_out,
                                                                         const float* __restrict__ l0, const float* __restrict__ c_in,
                                                                         float* __restrict__ c_out, float* __restrict__ part, int HS, int N,
                                                                         int nparts, int tpw) {
    constexpr int Q = C / 16, H2 = C / 64;
    // per wave: the 16-row tile, to change operand layout; the end-of-kernel fold of the partials reuses the same LDS (r05: 66 -> 34 KB at C = 128, so that
    // LDS no longer caps the kernel at two workgroups per CU; the register budget still does — CF_ROUTE_OCC = 3 spills, measured slower)
    __shared__ __attribute__((aligned(16))) float rt[4][16][C + 4];
    float4 (*red)[16][C / 4] = reinterpret_cast<float4 (*)[16][C / 4]>(&rt[0][0][0]);
    const int bt = blockIdx.y, chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    const bool al = (N & 3) == 0;
    f32x4 acc[NHT][C / 16];
#pragma unroll
    for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
        for (int i = 0; i < C / 16; ++i) acc[ht][i] = fzero4();
    float4 mb[NHT][Q];                                                // V[h = 16ht + j][16q+4kk ..]: B operand of rows . V^T
    size_t hrow[NHT];
#pragma unroll
    for (int ht = 0; ht < NHT; ++ht) {
        const int h = 16 * ht + j;
        hrow[ht] = ((size_t)bt * HS + min(h, HS - 1)) * N;
#pragma unroll
        for (int q = 0; q < Q; ++q) mb[ht][q] = (V != nullptr && h < HS) ? ld4(V + ((size_t)bt * HS + h) * C + 16 * q + 4 * kk) : f4zero();
    }
#pragma unroll 1
    for (int it = 0; it < tpw; ++it) {
        const int n0 = (chunk * tpw + it) * 64 + wave * 16;
        if (n0 >= N) break;
        const int n = n0 + 4 * kk;
        float4 a1[Q], a2[4][H2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int hf = 0; hf < H2; ++hf) a2[r][hf] = ld4(rows + ((size_t)bt * N + min(n + r, N - 1)) * C + 64 * hf + 4 * j);
        float4 b4[NHT], l4[NHT], c4[NHT];
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) {
            b4[ht] = f4zero(); l4[ht] = f4zero(); c4[ht] = f4zero();
            if (bl_in != nullptr) b4[ht] = ldn4(bl_in + hrow[ht], n, N, al);
            if (l0 != nullptr) l4[ht] = ldn4(l0 + hrow[ht], n, N, al);
            if (c_in != nullptr) c4[ht] = ldn4(c_in + hrow[ht], n, N, al);
        }
        SB();
        if (V != nullptr) {
            // row form of the same tile (row j, channels 16q+4kk..) through the wave's LDS tile: the rows come from HBM once, coalesced
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int hf = 0; hf < H2; ++hf) st4(&rt[wave][4 * kk + r][64 * hf + 4 * j], a2[r][hf]);
            SB();
#pragma unroll
            for (int q = 0; q < Q; ++q) a1[q] = ld4(&rt[wave][j][16 * q + 4 * kk]);
            SB();
#pragma unroll
            for (int ht = 0; ht < NHT; ++ht) {
                f32x4 L0 = fzero4(), L1 = fzero4(), L2 = fzero4(), L3 = fzero4();      // four independent accumulation chains
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    L0 = mfma4(a1[q].x, mb[ht][q].x, L0);
                    L1 = mfma4(a1[q].y, mb[ht][q].y, L1);
                    L2 = mfma4(a1[q].z, mb[ht][q].z, L2);
                    L3 = mfma4(a1[q].w, mb[ht][q].w, L3);
                }
                b4[ht].x += (L0[0] + L1[0]) + (L2[0] + L3[0]);
                b4[ht].y += (L0[1] + L1[1]) + (L2[1] + L3[1]);
                b4[ht].z += (L0[2] + L1[2]) + (L2[2] + L3[2]);
                b4[ht].w += (L0[3] + L1[3]) + (L2[3] + L3[3]);
            }
        }
        float cs[NHT][4];
        if (c_in != nullptr) {
#pragma unroll
            for (int ht = 0; ht < NHT; ++ht) {
                const bool hok = 16 * ht + j < HS;
                if (bl_out != nullptr && hok) stn4(bl_out + hrow[ht], n, N, al, b4[ht]);
                cs[ht][0] = (hok && n < N) ? c4[ht].x : 0.f; cs[ht][1] = (hok && n + 1 < N) ? c4[ht].y : 0.f;
                cs[ht][2] = (hok && n + 2 < N) ? c4[ht].z : 0.f; cs[ht][3] = (hok && n + 3 < N) ? c4[ht].w : 0.f;
            }
        } else {
            float4 x[NHT];
#pragma unroll
            for (int ht = 0; ht < NHT; ++ht) {
                if (bl_out != nullptr && 16 * ht + j < HS) stn4(bl_out + hrow[ht], n, N, al, b4[ht]);
                x[ht] = f4add(b4[ht], l4[ht]);
            }
            softmax_h4<NHT>(x, cs, j, HS, n, N);
            if (c_out != nullptr) {
#pragma unroll
                for (int ht = 0; ht < NHT; ++ht)
                    if (16 * ht + j < HS) stn4(c_out + hrow[ht], n, N, al, make_float4(cs[ht][0], cs[ht][1], cs[ht][2], cs[ht][3]));
            }
        }
        SB();
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int hf = 0; hf < H2; ++hf) {
                    acc[ht][4 * hf + 0] = mfma4(cs[ht][r], a2[r][hf].x, acc[ht][4 * hf + 0]);
                    acc[ht][4 * hf + 1] = mfma4(cs[ht][r], a2[r][hf].y, acc[ht][4 * hf + 1]);
                    acc[ht][4 * hf + 2] = mfma4(cs[ht][r], a2[r][hf].z, acc[ht][4 * hf + 2]);
                    acc[ht][4 * hf + 3] = mfma4(cs[ht][r], a2[r][hf].w, acc[ht][4 * hf + 3]);
                }
    }
    __syncthreads();                                                  // every wave is done with its row tile: the fold takes the LDS over
    store_partial<C, NHT, 0>(red, acc, part + ((size_t)bt * nparts + chunk) * HS * C, HS, wave, j, kk);
}

// ---- rec[bt,n,:] = sum_h c[bt,h,n] v[bt,h,:]  (cluster -> node scatter, GPTST.py:135) as rec^T = v^T c^T on MFMA --------------------------
template <int C, int NHT>
__global__ __launch_bounds__(256) void cf_rec_fwd_kernel(const float* __restrict__ c, const float* __restrict__ v, float* __restrict__ rec,
                                                         int HS, int N, int tpw) {
    constexpr int Q = C / 16;
    const int bt = blockIdx.y, chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    float va[NHT][Q][4];                                              // A[i = channel 16q+j][k-step (ht, s): h = 16ht+4kk+s]
#pragma unroll
    for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int h = 16 * ht + 4 * kk + s;
                va[ht][q][s] = h < HS ? v[((size_t)bt * HS + h) * C + 16 * q + j] : 0.f;
            }
#pragma unroll 1
    for (int it = 0; it < tpw; ++it) {
        const int n0 = (chunk * tpw + it) * 64 + wave * 16;
        if (n0 >= N) break;
        const bool ok = n0 + j < N;
        float ct[NHT][4];                                             // B[k: h = 16ht+4kk+s][row j]
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int h = 16 * ht + 4 * kk + s;
                ct[ht][s] = (ok && h < HS) ? c[((size_t)bt * HS + h) * N + n0 + j] : 0.f;
            }
        f32x4 acc[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = fzero4();
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[q] = mfma4(va[ht][q][s], ct[ht][s], acc[q]);
        if (ok) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
                st4(rec + ((size_t)bt * N + n0 + j) * C + 16 * q + 4 * kk, make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]));
        }
    }
}

// ---- backward through s = c P, c = softmax_h(b + dadj), P = squash(Y) for the rows of Y (cb_route_bwd_rows_kernel on MFMA) ------------
//   U[n,h] = dS[h,:].P[n,:];  dc = dc1 + U;  dlogit[h] = c[h] (dc[h] - sum_h c dc);  dP = sum_h c[h] dS[h,:];  dY = g dP + Y 2 g'(q) (Y.dP)
template <int C, int NHT>
__global__ __launch_bounds__(256) void cf_route_bwd_kernel(const float* __restrict__ Y, const float* __restrict__ c, const float* __restrict__ dc1,
                                                           const float* __restrict__ dS, float* __restrict__ dY, float* __restrict__ dlogit,
                                                           int HS, int N, int tpw) {
    constexpr int Q = C / 16;
    const int bt = blockIdx.y, chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    const bool al = (N & 3) == 0;
    float4 mb[NHT][Q];                                                // dS[h = 16ht+j][16q+4kk ..]: B operand of Y . dS^T
    float da[NHT][Q][4];                                              // dS[h = 16ht+4kk+s][16q+j]:  A operand of dP^T = dS^T c^T
    size_t hrow[NHT];
#pragma unroll
    for (int ht = 0; ht < NHT; ++ht) {
        hrow[ht] = ((size_t)bt * HS + min(16 * ht + j, HS - 1)) * N;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            mb[ht][q] = 16 * ht + j < HS ? ld4(dS + ((size_t)bt * HS + 16 * ht + j) * C + 16 * q + 4 * kk) : f4zero();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int h = 16 * ht + 4 * kk + s;
                da[ht][q][s] = h < HS ? dS[((size_t)bt * HS + h) * C + 16 * q + j] : 0.f;
            }
        }
    }
#pragma unroll 1
    for (int it = 0; it < tpw; ++it) {
        const int n0 = (chunk * tpw + it) * 64 + wave * 16;
        if (n0 >= N) break;
        const int n = n0 + 4 * kk;
        const bool ok = n0 + j < N;
        float4 a1[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) a1[q] = ld4(Y + ((size_t)bt * N + min(n0 + j, N - 1)) * C + 16 * q + 4 * kk);
        float4 c1[NHT], d1[NHT];
        float ct[NHT][4];
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) {
            c1[ht] = ldn4(c + hrow[ht], n, N, al); d1[ht] = ldn4(dc1 + hrow[ht], n, N, al);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int h = 16 * ht + 4 * kk + s;
                ct[ht][s] = (ok && h < HS) ? c[((size_t)bt * HS + h) * N + n0 + j] : 0.f;
            }
        }
        SB();
        float qn = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) qn += f4dot(a1[q], a1[q]);
        qn += __shfl_xor(qn, 16, 64); qn += __shfl_xor(qn, 32, 64);      // |Y[row j]|^2 in every lane of column j
        const float rt = sqrtf(qn), den = (1.f + qn) * (rt + 1e-8f);
        const float g = qn / den;
        float gp = 0.f;
        if (rt > 0.f) gp = (den - qn * ((rt + 1e-8f) + (1.f + qn) * 0.5f / rt)) / (den * den);
        float gr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gr[r] = __shfl(g, 4 * kk + r, 64);                 // squash gain of row 4kk+r
        // node form: rows 4kk+r, cluster 16ht+j
        float dch[NHT][4], w[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) {
            f32x4 L0 = fzero4(), L1 = fzero4(), L2 = fzero4(), L3 = fzero4();
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                L0 = mfma4(a1[q].x, mb[ht][q].x, L0);
                L1 = mfma4(a1[q].y, mb[ht][q].y, L1);
                L2 = mfma4(a1[q].z, mb[ht][q].z, L2);
                L3 = mfma4(a1[q].w, mb[ht][q].w, L3);
            }
            const float cv[4] = {c1[ht].x, c1[ht].y, c1[ht].z, c1[ht].w}, dv[4] = {d1[ht].x, d1[ht].y, d1[ht].z, d1[ht].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dch[ht][r] = dv[r] + gr[r] * ((L0[r] + L1[r]) + (L2[r] + L3[r]));
                w[r] += 16 * ht + j < HS ? cv[r] * dch[ht][r] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = group_sum<16>(w[r]);
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht)
            if (16 * ht + j < HS)
                stn4(dlogit + hrow[ht], n, N, al, make_float4(c1[ht].x * (dch[ht][0] - w[0]), c1[ht].y * (dch[ht][1] - w[1]),
                                                                c1[ht].z * (dch[ht][2] - w[2]), c1[ht].w * (dch[ht][3] - w[3])));
        // row form: dP[row j][16q+4kk+r'] = acc[q][r']
        f32x4 acc[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = fzero4();
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[q] = mfma4(da[ht][q][s], ct[ht][s], acc[q]);
        float ydp = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) ydp += a1[q].x * acc[q][0] + a1[q].y * acc[q][1] + a1[q].z * acc[q][2] + a1[q].w * acc[q][3];
        ydp += __shfl_xor(ydp, 16, 64); ydp += __shfl_xor(ydp, 32, 64);
        const float k2 = 2.f * gp * ydp;
        if (ok) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
                st4(dY + ((size_t)bt * N + n0 + j) * C + 16 * q + 4 * kk,
                    make_float4(fmaf(k2, a1[q].x, g * acc[q][0]), fmaf(k2, a1[q].y, g * acc[q][1]), fmaf(k2, a1[q].z, g * acc[q][2]),
                                fmaf(k2, a1[q].w, g * acc[q][3])));
        }
    }
}

// ---- ordered fold of the chunk partials + the cluster-level post step; one workgroup per (b,t) -----------------------------------------
//   part (BT, nparts, prow, C);  mode 0: prow = HS+1: v0 = squash(S0) -> V0, v = squash(v0 (.) colsum/HS) -> Vout (if given)
//   mode 1: v = squash(V0 (.) S) -> Vout;   mode 2: S -> Vout;   mode 3: all prow rows -> Vout (fold only: a node-sharded run all-reduces it)
template <int C>
__global__ __launch_bounds__(256) void cf_post_kernel(const float* __restrict__ part, int nparts, int prow, float* __restrict__ V0,
                                                      float* __restrict__ Vout, int mode, int HS) {
    constexpr int L = C / 4, SL = 256 / L;
    const int bt = blockIdx.x, c4 = threadIdx.x % L, slot = threadIdx.x / L;
    const float* pb = part + (size_t)bt * nparts * prow * C;
    float4 cs4 = f4zero();
    if (mode == 0) {
        for (int p = 0; p < nparts; ++p) cs4 = f4add(cs4, ld4(pb + ((size_t)p * prow + HS) * C + 4 * c4));
        const float ih = 1.f / (float)HS;
        cs4 = make_float4(cs4.x * ih, cs4.y * ih, cs4.z * ih, cs4.w * ih);
    }
    const int nrow = mode == 3 ? prow : HS;
    for (int h = slot; h < nrow; h += SL) {
        float4 s = f4zero();
        for (int p = 0; p < nparts; ++p) s = f4add(s, ld4(pb + ((size_t)p * prow + h) * C + 4 * c4));
        const size_t o = ((size_t)bt * nrow + h) * C + 4 * c4;
        if (mode >= 2) { st4(Vout + o, s); continue; }
        if (mode == 0) {
            const float sc = squash_scale(group_sum<L>(f4dot(s, s)));
            s = make_float4(s.x * sc, s.y * sc, s.z * sc, s.w * sc);
            st4(V0 + o, s);
            if (Vout == nullptr) continue;
            s = make_float4(s.x * cs4.x, s.y * cs4.y, s.z * cs4.z, s.w * cs4.w);
        } else {
            const float4 v0 = ld4(V0 + o);
            s = make_float4(s.x * v0.x, s.y * v0.y, s.z * v0.z, s.w * v0.w);
        }
        const float sc = squash_scale(group_sum<L>(f4dot(s, s)));
        st4(Vout + o, make_float4(s.x * sc, s.y * sc, s.z * sc, s.w * sc));
    }
}

// ---- C ABI ---------------------------------------------------------------------------------------------------------------------------
#define CF_SHAPE_OK(HS, C) ((HS) >= 1 && (((HS) <= 16 && ((C) == 64 || (C) == 128)) || ((HS) <= 64 && (C) == 64)))
#define CF_LAUNCH(KERNEL, GRID, ...)                                                                                   \
    do {                                                                                                               \
        const int nht_ = (HS + 15) / 16;                                                                               \
        hipStream_t st_ = (hipStream_t)stream;                                                                         \
        if (C == 128) hipLaunchKernelGGL((KERNEL<128, 1>), GRID, dim3(256), 0, st_, __VA_ARGS__);                      \
        else if (nht_ == 1) hipLaunchKernelGGL((KERNEL<64, 1>), GRID, dim3(256), 0, st_, __VA_ARGS__);                 \
        else if (nht_ == 2) hipLaunchKernelGGL((KERNEL<64, 2>), GRID, dim3(256), 0, st_, __VA_ARGS__);                 \
        else if (nht_ == 3) hipLaunchKernelGGL((KERNEL<64, 3>), GRID, dim3(256), 0, st_, __VA_ARGS__);                 \
        else hipLaunchKernelGGL((KERNEL<64, 4>), GRID, dim3(256), 0, st_, __VA_ARGS__);                                \
        GPTST_CHECK_LAUNCH();                                                                                          \
        return GPTST_OK;                                                                                               \
    } while (0)

extern "C" int gptst_capflow_supported(int HS, int C) { return CF_SHAPE_OK(HS, C) ? 1 : 0; }
extern "C" int gptst_capflow_nparts(int N) { const int rows = 64 * cf_tpw(N); return (N + rows - 1) / rows; }

extern "C" int gptst_capflow_squash(const float* Y, const float* dadj, float* P, float* part, int BT, int HS, int N, int C, void* stream) {
    if (!Y || !dadj || !P || !part || BT < 1 || N < 1) return GPTST_EARG;
    if (!CF_SHAPE_OK(HS, C)) return GPTST_ESHAPE;
    const int np = gptst_capflow_nparts(N);
    CF_LAUNCH(cf_squash_kernel, dim3(np, BT), Y, dadj, P, part, HS, N, np, cf_tpw(N));
}

extern "C" int gptst_capflow_route(const float* rows, const float* V, const float* bl_in, float* bl_out, const float* l0, const float* c_in,
                                   float* c_out, float* part, int BT, int HS, int N, int C, void* stream) {
    if (!rows || !part || BT < 1 || N < 1) return GPTST_EARG;
    if (!CF_SHAPE_OK(HS, C)) return GPTST_ESHAPE;
    const int np = gptst_capflow_nparts(N);
    CF_LAUNCH(cf_route_kernel, dim3(np, BT), rows, V, bl_in, bl_out, l0, c_in, c_out, part, HS, N, np, cf_tpw(N));
}

extern "C" int gptst_capflow_post(const float* part, int nparts, int prow, float* V0, float* Vout, int mode, int BT, int HS, int C, void* stream) {
    if (!part || nparts < 1 || mode < 0 || mode > 3 || (mode <= 1 && !V0) || (mode >= 1 && !Vout)) return GPTST_EARG;
    if (prow != (mode == 0 ? HS + 1 : HS) && mode != 3) return GPTST_EARG;
    if (!CF_SHAPE_OK(HS, C)) return GPTST_ESHAPE;
    if (C == 64) hipLaunchKernelGGL((cf_post_kernel<64>), dim3(BT), dim3(256), 0, (hipStream_t)stream, part, nparts, prow, V0, Vout, mode, HS);
    else hipLaunchKernelGGL((cf_post_kernel<128>), dim3(BT), dim3(256), 0, (hipStream_t)stream, part, nparts, prow, V0, Vout, mode, HS);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_capflow_rec_fwd(const float* c, const float* v, float* rec, int BT, int HS, int N, int C, void* stream) {
    if (!c || !v || !rec || BT < 1 || N < 1) return GPTST_EARG;
    if (!CF_SHAPE_OK(HS, C)) return GPTST_ESHAPE;
    CF_LAUNCH(cf_rec_fwd_kernel, dim3(gptst_capflow_nparts(N), BT), c, v, rec, HS, N, cf_tpw(N));
}

extern "C" int gptst_capflow_route_bwd(const float* Y, const float* c, const float* dc1, const float* dS, float* dY, float* dlogit, int BT,
                                       int HS, int N, int C, void* stream) {
    if (!Y || !c || !dc1 || !dS || !dY || !dlogit || BT < 1 || N < 1) return GPTST_EARG;
    if (!CF_SHAPE_OK(HS, C)) return GPTST_ESHAPE;
    CF_LAUNCH(cf_route_bwd_kernel, dim3(gptst_capflow_nparts(N), BT), Y, c, dc1, dS, dY, dlogit, HS, N, cf_tpw(N));
}
