// cap soft assignment + routing, forward — third generation (round 4; reference GPTST.py:102-123).
//
// cap_route_fwd2_kernel (cap_mfma.hip) keeps the N x C capsule matrix of a (b,t) in LDS and walks 16 barrier-separated phases, each a
// short latency chain (profiles/r03_cap_route_fwd2_phases.txt: 20 us for ONE workgroup alone on a CU against 2.7 us of MFMA issue).
// Here a (b,t) workgroup has ONE WAVE PER 16-NODE TILE and the tile's capsule rows never leave the wave's registers:
//   P tile   = squash(X_tile Wp^T + bp)                 64 MFMA 16x16x4, operands straight from global (A) / staged Wp (B); kept in BOTH
//                                                       operand layouts (yD: accumulator layout = B operand of c.P;  yA: row layout = B
//                                                       operand of V.P^T — one trip through a wave-private LDS tile)
//   bl[h,n] += sum_c V[h,c] P[n,c]                      16 MFMA, result = 4 registers per lane (cluster 4kk+r, node j)
//   c        = softmax_h(bl (+ dadj))                   in registers: 4 values per lane, 2 cross-row shuffles per reduction
//   S_tile   = c . P_tile                               16 MFMA (c changes to the A-operand layout through 1 KB of wave-private LDS)
// so a routing iteration is wave-local; only the sum of the tile contributions over the waves and the cluster-level step
// (squash, v0 (.) S) cross the workgroup: ONE fold per iteration = 2 barriers (partials -> [barrier] -> 16 HS threads fold + squash ->
// [barrier]).  R = 2: 6 barriers instead of 16, and no phase in which most waves idle.  The (B,T,HS,N,C) tensor of the reference
// (:106-107) is never formed:  s[h,:] = v0[h,:] (.) sum_n c[h,n] P[n,:];  the first iteration has c = 1/HS (b = 0, :112), i.e. its sum is
// the column sum of P, which rides on the first fold.
// Serves C = 64, N <= 256 (<= 16 waves), HS <= 16, 64 * waves >= 16 * HS; other shapes -> cap_route_fwd2_kernel / cap.hip.
#include "common.h"

#define CR3_P 68          // pitch of the [16][64] LDS tiles
#ifdef CR3_STAMPS          // per-phase stamps of one workgroup (tools/experiments/cap_route3_phases.hip); not in the product build
__device__ long long g_cr3_ts[4][32];
#define CR3_TS(i) do { if (lane == 0 && (wave == 0 || wave == NW - 1) && (blockIdx.x == 5 || blockIdx.x == 300)) \
    g_cr3_ts[(blockIdx.x == 300) * 2 + (wave != 0)][i] = wall_clock64(); } while (0)
#else
#define CR3_TS(i) do { } while (0)
#endif
#ifdef CR3_STAMPS
__device__ long long g_cr4_wg[1024][4];       // per workgroup: start, end (wall clock, 100 MHz), HW_ID, XCC_ID
#define CR4_WG_BEGIN() do { if (threadIdx.x == 0) { g_cr4_wg[blockIdx.x][0] = wall_clock64(); \
    g_cr4_wg[blockIdx.x][2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
    g_cr4_wg[blockIdx.x][3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); } } while (0)
#define CR4_WG_END() do { if (threadIdx.x == 0) g_cr4_wg[blockIdx.x][1] = wall_clock64(); } while (0)
__device__ long long g_cr4_ts[6][32];       // [workgroup 5 / 200 / 261][wave 0 / 7][stamp]
#define CR4_TS(i) do { if (lane == 0 && (wave == 0 || wave == CR4_NW - 1) && (blockIdx.x == 5 || blockIdx.x == 200 || blockIdx.x == 261)) \
    g_cr4_ts[(blockIdx.x == 200 ? 2 : blockIdx.x == 261 ? 4 : 0) + (wave != 0)][i] = wall_clock64(); } while (0)
#else
#define CR4_TS(i) do { } while (0)
#define CR4_WG_BEGIN() do { } while (0)
#define CR4_WG_END() do { } while (0)
#endif

__device__ __forceinline__ void cr3_softmax(const float (&x)[4], float (&c)[4], int kk, int HS, bool ok) {
    float m = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 4; ++r) if (4 * kk + r < HS) m = fmaxf(m, x[r]);
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float e[4], sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { e[r] = (4 * kk + r < HS) ? __expf(x[r] - m) : 0.f; sum += e[r]; }     // v_exp_f32 path, as cm_softmax
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = ok ? e[r] * inv : 0.f;
}

// S_tile[h][ch] = sum_{n in tile} c[h][n] P[n][ch]  ->  the wave's partial in scr as [h][64] (h < HS); c in the (cluster 4kk+r, node j) layout.
// MFMA k-slot (s, kk) <-> node 4kk + s of the tile: B = yD[s] (row 4kk+s, channels 4j..4j+3 <-> column j of tiles 0..3), A = c[h = j][4kk + s].
__device__ __forceinline__ void cr3_type1(const float (&c)[4], const float4 (&yD)[4], float* __restrict__ scr, int j, int kk, int HS) {
#pragma unroll
    for (int r = 0; r < 4; ++r) scr[(4 * kk + r) * 17 + j] = c[r];
    float cA[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) cA[s] = scr[j * 17 + 4 * kk + s];
    // two column tiles at a time: 8 accumulator registers live instead of 16 (the <= 80 VGPR variant runs two workgroups per CU)
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cA[s], yD[s].x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cA[s], yD[s].y, a1, 0, 0, 0);
    }
    f32x4 a2 = {0.f, 0.f, 0.f, 0.f}, a3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(cA[s], yD[s].z, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(cA[s], yD[s].w, a3, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (4 * kk + r < HS) st4(scr + (4 * kk + r) * 64 + 4 * j, make_float4(a0[r], a1[r], a2[r], a3[r]));
}

// bl[h][n] += sum_ch V[h][ch] P[n][ch]:  A = V[h = j][16q + 4kk + e] (LDS, rows >= HS zero), B = yA[q].e (node j, the same channel)
__device__ __forceinline__ void cr3_type2(float (&bl)[4], const float4 (&yA)[4], const float* __restrict__ Vs, int j, int kk) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
        const float4 v0 = ld4(Vs + j * CR3_P + 16 * q + 4 * kk), v1 = ld4(Vs + j * CR3_P + 16 * (q + 1) + 4 * kk);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.x, yA[q].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.y, yA[q].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.z, yA[q].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.w, yA[q].w, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.x, yA[q + 1].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.y, yA[q + 1].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.z, yA[q + 1].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.w, yA[q + 1].w, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bl[r] += acc0[r] + acc1[r];
}

// fold of the waves' partials in index order: row h = tid / 16, channels 4 (tid % 16) ..
__device__ __forceinline__ float4 cr3_fold(const float* __restrict__ scr0, int NW, int row, int c4) {
    float4 s = f4zero();
    for (int w0 = 0; w0 < NW; w0 += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld4(scr0 + min(w0 + u, NW - 1) * 16 * CR3_P + row * 64 + 4 * c4);
#pragma unroll
        for (int u = 0; u < 4; ++u) if (w0 + u < NW) s = f4add(s, v[u]);
    }
    return s;
}

// ---- fourth variant: 8 waves, capsule tiles in LDS, TWO workgroups per CU -------------------------------------------------------------------
// cap_route_fwd3_kernel (the third variant; r05: moved to tools/experiments/cap_route_fwd3_kernel.hip) is one 11-wave workgroup per CU at a time (94 VGPRs; 80 would spill): its 15 us dependency chain
// (tools/experiments/cap_route3_phases.hip: stage 2.0, tile GEMM 3.5-5.8 at three waves per SIMD, three routing passes 1-2 us each, folds
// 0.6-1.4) runs TWICE on the CUs that get two of the 384 (b,t) — 33 us, no better than the second generation.  Here the same wave-local
// passes run from 8 waves with the capsule tiles in LDS (wave w owns tiles w and w + 8: its rows are read back in either operand layout as
// the passes need them, nothing but the logits stays in registers), the weight fragments come straight from global memory / L1 (no staging
// barrier), a wave's tile contributions add up in its accumulators (8 partials instead of 11) and the cluster-level fold runs on the waves
// that own ONE tile.  75 KB of LDS and <= 128 VGPRs: two workgroups per CU, so all 384 are resident at once and one's latency phases
// overlap the other's MFMA phases.
#define CR4_NW 8
template <int TPW>
__global__ __launch_bounds__(64 * CR4_NW, 4) void cap_route_fwd4_kernel(const float* __restrict__ X, const float* __restrict__ Wp,
                                                                        const float* __restrict__ bp, const float* __restrict__ dadj,
                                                                        float* __restrict__ c_out, float* __restrict__ s_out, int N, int HS, int R,
                                                                        int redw) {
    constexpr int C = 64, P = CR3_P, NTH = 64 * CR4_NW;
    // (A start lag for the second-resident workgroups of a CU — blocks >= 256, to put one's capsule GEMM under the other's routing passes — was
    //  measured in r04 here and in the hyperTem pair launch: no gain at any lag, removed.)
    CR4_WG_BEGIN();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ntiles = (N + 15) >> 4;
    float* Ps = smem;                          // [16 ntiles][P]   capsule rows
    float* red0 = Ps + ntiles * 16 * P;        // CR4_NW x redw:   the wave's c-transposition scratch [16][17], then its partial sums [HS + 1][64]
    float* Vs = red0 + CR4_NW * redw;          // [16][P]          v of the running iteration (rows >= HS stay zero)
    const int bt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    float* scr = red0 + wave * redw;
    const float* Xbt = X + (size_t)bt * N * C;
    const float* l0g = dadj + (size_t)bt * HS * N;

    // ---- every global load of the prologue in one batch: the tiles' rows (A operand), the weight fragments (B operand, Wp[4j + ct][16q + 4kk ..]),
    //      the assignment logits of the first tile ----
    CR4_TS(0);
    float4 a[4], bv[4][4];
    float l0[TPW][4], bl[TPW][4];
    {
        const float* row = Xbt + (size_t)min(16 * wave + j, N - 1) * C + 4 * kk;
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = ld4(row + 16 * q);
    }
    // Weight fragments: fragment (ct, q) of lane (j, kk) is Wp[4j + ct][16q + 4kk ..] — 16 rows x 64 B per wave load.  Every wave needs all 16
    // fragments; loaded by each wave from global memory they were 128 of these strided loads per workgroup (prologue 5.6 us alone on a CU, 13 us
    // for a second resident: tools/experiments/cap_route3_phases.hip).  The workgroup fetches each fragment ONCE (two loads per thread) into LDS
    // in fragment order (the partial-sum region, free until the first pass) and every wave reads its copy back conflict-free.
    float4 wf[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int f = tid + u * NTH, fr = f >> 6, fl = f & 63;          // fragment fr = 4 ct + q, lane fl
        wf[u] = ld4(Wp + (size_t)(4 * (fl & 15) + (fr >> 2)) * C + 16 * (fr & 3) + 4 * (fl >> 4));
    }
    const float4 b4 = ld4(bp + 4 * j);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int ncol = 16 * (wave + CR4_NW * i) + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            l0[i][r] = (4 * kk + r < HS && ncol < N) ? l0g[(size_t)(4 * kk + r) * N + ncol] : 0.f;
            bl[i][r] = 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) st4(red0 + 4 * (tid + u * NTH), wf[u]);
    for (int i = tid; i < 16 * P; i += NTH) Vs[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[ct][q] = ld4(red0 + 4 * ((4 * ct + q) * 64 + lane));
    __syncthreads();                                       // the fragment image is dead: the region turns into the waves' scratch
    SB();
    CR4_TS(1);
    // ---- capsule tiles P = squash(X_tile Wp^T + bp) -> LDS (rows of a tile are only ever read by the wave that wrote them) ----
    float4 csum = f4zero();
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave + CR4_NW * i;
        if (tile < ntiles) {                               // wave-uniform
            float4 an[4];                                  // the next tile's rows: in flight during this tile's MFMAs
            if (i + 1 < TPW) {
                const float* row = Xbt + (size_t)min(16 * (tile + CR4_NW) + j, N - 1) * C + 4 * kk;
#pragma unroll
                for (int q = 0; q < 4; ++q) an[q] = ld4(row + 16 * q);
            }
            f32x4 acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
                const float b0[4] = {bv[0][q].x, bv[0][q].y, bv[0][q].z, bv[0][q].w}, b1[4] = {bv[1][q].x, bv[1][q].y, bv[1][q].z, bv[1][q].w};
                const float b2[4] = {bv[2][q].x, bv[2][q].y, bv[2][q].z, bv[2][q].w}, b3[4] = {bv[3][q].x, bv[3][q].y, bv[3][q].z, bv[3][q].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b0[e], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b1[e], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b2[e], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b3[e], acc[3], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * tile + 4 * kk + r;
                float4 v = f4add(make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), b4);
                if (n >= N) v = f4zero();
                const float sc = squash_scale(group_sum<16>(f4dot(v, v)));
                v = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
                csum = f4add(csum, v);
                st4(Ps + (16 * tile + 4 * kk + r) * P + 4 * j, v);
            }
            if (i + 1 < TPW) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = an[q];
            }
        }
    }
    csum.x += __shfl_xor(csum.x, 16, 64); csum.y += __shfl_xor(csum.y, 16, 64); csum.z += __shfl_xor(csum.z, 16, 64); csum.w += __shfl_xor(csum.w, 16, 64);
    csum.x += __shfl_xor(csum.x, 32, 64); csum.y += __shfl_xor(csum.y, 32, 64); csum.z += __shfl_xor(csum.z, 32, 64); csum.w += __shfl_xor(csum.w, 32, 64);
    SB();
    CR4_TS(2);
    // one routing pass over this wave's tiles: [b += v . P^T] -> c = softmax_h(b [+ dadj]) [-> c_out] -> the wave's partial of c . P
    auto pass = [&](bool upd, bool use_bl, bool use_l0, bool out) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
        float4 vA[4];
        if (upd) {
#pragma unroll
            for (int q = 0; q < 4; ++q) vA[q] = ld4(Vs + j * P + 16 * q + 4 * kk);          // A = V[h = j][16q + 4kk ..] (rows >= HS zero)
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int tile = wave + CR4_NW * i;
            if (tile < ntiles) {
                const int ncol = 16 * tile + j;
                if (upd) {                                                                   // B = P[node j][16q + 4kk ..]
                    f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = u0;
                    float4 pr[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) pr[q] = ld4(Ps + (16 * tile + j) * P + 16 * q + 4 * kk);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        u0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vA[q].x, pr[q].x, u0, 0, 0, 0);
                        u1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vA[q].y, pr[q].y, u1, 0, 0, 0);
                        u0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vA[q].z, pr[q].z, u0, 0, 0, 0);
                        u1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vA[q].w, pr[q].w, u1, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) bl[i][r] += u0[r] + u1[r];
                }
                float x[4], c[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = (use_bl ? bl[i][r] : 0.f) + (use_l0 ? l0[i][r] : 0.f);
                cr3_softmax(x, c, kk, HS, ncol < N);
                if (out && ncol < N) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * kk + r < HS) c_out[((size_t)bt * HS + 4 * kk + r) * N + ncol] = c[r];
                }
                // c -> A-operand layout (k-slot (s, kk) <-> node 4kk + s of the tile), B = P rows 4kk + s as float4 (channel 4j + ct <-> column j of tile ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[(4 * kk + r) * 17 + j] = c[r];
                float cA[4];
                float4 pd[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) { cA[s] = scr[j * 17 + 4 * kk + s]; pd[s] = ld4(Ps + (16 * tile + 4 * kk + s) * P + 4 * j); }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cA[s], pd[s].x, s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cA[s], pd[s].y, s1, 0, 0, 0);
                    s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(cA[s], pd[s].z, s2, 0, 0, 0);
                    s3 = __builtin_amdgcn_mfma_f32_16x16x4f32(cA[s], pd[s].w, s3, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * kk + r < HS) st4(scr + (4 * kk + r) * 64 + 4 * j, make_float4(s0[r], s1[r], s2[r], s3[r]));
    };
    // fold of the 8 partials in index order
    auto fold = [&](int row, int c4) {
        float4 v[CR4_NW];
#pragma unroll
        for (int w = 0; w < CR4_NW; ++w) v[w] = ld4(red0 + w * redw + row * 64 + 4 * c4);
        float4 s = v[0];
#pragma unroll
        for (int w = 1; w < CR4_NW; ++w) s = f4add(s, v[w]);
        return s;
    };
    // ---- c0 = softmax_h(dadj): partial of c0 . P and of the column sums of P (:105; first routing pass c = 1/HS) ----
    pass(false, false, true, false);
    if (kk == 0) st4(scr + HS * 64 + 4 * j, csum);
    CR4_TS(3);
    __syncthreads();
    SB();
    CR4_TS(4);
    // cluster-level steps: 16 lanes per cluster row, on the LAST waves (they own one tile each)
    const int pid = NTH - 1 - tid, prow = pid >> 4, pc4 = pid & 15;
    const bool poster = pid < 16 * HS, pwave = (NTH - 64 * (wave + 1)) < 16 * HS;          // pwave: wave-uniform
    float4 v0 = f4zero();
    if (pwave) {                                       // v0 = squash(c0 . P) (:105-106), v = squash(v0 (.) mean-over-classes of the column sums)
        float4 S0 = f4zero(), u0 = f4zero();
        if (poster) { S0 = fold(prow, pc4); if (R > 0) u0 = fold(HS, pc4); }
        const float sc = squash_scale(group_sum<16>(f4dot(S0, S0)));
        v0 = make_float4(S0.x * sc, S0.y * sc, S0.z * sc, S0.w * sc);
        if (R > 0) {
            const float inv = 1.f / (float)HS;
            const float4 t = make_float4(v0.x * (u0.x * inv), v0.y * (u0.y * inv), v0.z * (u0.z * inv), v0.w * (u0.w * inv));
            const float s2 = squash_scale(group_sum<16>(f4dot(t, t)));
            if (poster) st4(Vs + prow * P + 4 * pc4, make_float4(t.x * s2, t.y * s2, t.z * s2, t.w * s2));
        }
    }
    CR4_TS(5);
    for (int it = 1; it < R; ++it) {                   // routing iterations 1 .. R-1 (no grad, :113-118)
        __syncthreads();
        SB();
        CR4_TS(6);
        pass(true, true, false, false);                // b += v . P^T;  c = softmax_h(b);  partial of c . P
        CR4_TS(7);
        __syncthreads();
        SB();
        CR4_TS(8);
        if (pwave) {                                   // v = squash(v0 (.) c . P)
            float4 S = f4zero();
            if (poster) S = fold(prow, pc4);
            const float4 t = make_float4(v0.x * S.x, v0.y * S.y, v0.z * S.z, v0.w * S.w);
            const float s2 = squash_scale(group_sum<16>(f4dot(t, t)));
            if (poster) st4(Vs + prow * P + 4 * pc4, make_float4(t.x * s2, t.y * s2, t.z * s2, t.w * s2));
        }
    }
    CR4_TS(9);
    __syncthreads();
    SB();
    CR4_TS(10);
    pass(R > 0, true, true, true);                     // b += v . P^T;  c = softmax_h(b + dadj) -> c_out (:120);  partial of s = c . P (:123)
    CR4_TS(11);
    __syncthreads();
    SB();
    CR4_TS(12);
    if (poster) st4(s_out + ((size_t)bt * HS + prow) * C + 4 * pc4, fold(prow, pc4));
    CR4_TS(13);
    CR4_WG_END();
}

thread_local int g_cap_route_v2 = 0;          // gptst_tune(20, 1): cap_route_fwd2_kernel (second generation) also where this kernel serves

// GPTST_ESHAPE: not served here (the caller falls back to the LDS-resident second generation)
GPTST_INTERNAL int gptst_cap_route_fwd3(const float* X, const float* Wp, const float* bp, const float* dadj, float* c_out, float* s_out, int BT,
                                        int N, int C, int HS, int R, void* stream) {
    if (g_cap_route_v2 || C != 64 || N > 256 || HS > 16 || HS <= 0) return GPTST_ESHAPE;
    const int NW = (N + 15) / 16;
    {
        const int redw = (HS + 1) * 64 > 512 ? (HS + 1) * 64 : 512;          // >= the c scratch [16][17]; 8 of them >= the 16 KB fragment image
        const size_t smem4 = ((size_t)NW * 16 * CR3_P + (size_t)CR4_NW * redw + 16 * CR3_P) * sizeof(float);
        static size_t cur4[2] = {0, 0};
        if (NW <= CR4_NW) {
            if (smem4 > cur4[0]) { (void)hipFuncSetAttribute((const void*)cap_route_fwd4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4); cur4[0] = smem4; }
            hipLaunchKernelGGL((cap_route_fwd4_kernel<1>), dim3(BT), dim3(64 * CR4_NW), smem4, (hipStream_t)stream, X, Wp, bp, dadj, c_out, s_out, N, HS, R, redw);
        } else {
            if (smem4 > cur4[1]) { (void)hipFuncSetAttribute((const void*)cap_route_fwd4_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4); cur4[1] = smem4; }
            hipLaunchKernelGGL((cap_route_fwd4_kernel<2>), dim3(BT), dim3(64 * CR4_NW), smem4, (hipStream_t)stream, X, Wp, bp, dadj, c_out, s_out, N, HS, R, redw);
        }
        GPTST_CHECK_LAUNCH();
        return GPTST_OK;
    }
}
