// cap soft assignment + routing, forward (reference GPTST.py:102-123) — one workgroup (512 threads = 8 waves) per (b,t).
// The N x C capsule matrix P = squash(X Wp^T + bp) is produced by fp32 MFMA 32x32x2 straight into LDS and stays there; the
// (B,T,HS,N,C) tensor of the reference (:106-107) is never formed:  s[h,:] = v0[h,:] (.) sum_n c[h,n] P[n,:].
// The two dense node-embedding x cluster-centroid contractions run on fp32 MFMA 16x16x4 (clusters padded to tiles of 16):
//   type 1   S[h, c]  = sum_n cs[h, n] P[n, c]    wave w owns 16 columns (c = 16w..16w+15) for ALL nodes: no cross-wave fold
//   type 2   bl[h, n] += sum_c V[h, c] P[n, c]    waves take 16-node tiles
// and the softmax over clusters uses a quad of lanes per node (2 DPP steps).
// History (profiles/r01b, in-kernel s_memtime stamps): the VALU version and a first MFMA version with LDS float atomics
// both took ~90 us: every phase was a serial latency chain (ds_add_f32 with 8-way bank conflicts, one thread per node doing
// 30 dependent LDS round trips, 2 waves per SIMD).  LDS at N=170, HS=10, C=64: 78.7 KB -> two workgroups per CU.
#include "mfma_tile.h"
#include "poolgen_dev.h"
#include <vector>
#ifdef GPTST_DEBUG
__device__ long long g_cap_ts[64];     // per-phase s_memtime stamps of workgroup 5 (enabled by gptst_tune2(99))
static thread_local int g_cap_dbg = 0;
extern "C" int gptst_tune2(int v) { g_cap_dbg = v; return 0; }
extern "C" int gptst_cap_ts(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cap_ts), sizeof(long long) * 64); }
#else
static constexpr int g_cap_dbg = 0;
#endif

GPTST_STAMP_TABLES(capmfma)
GPTST_HANDOFF_COUNTER(capmfma)

#define CM_NT 512
#define CM_NW (CM_NT / 64)

__host__ __device__ inline int cm_rows(int N) { return (N + 15) / 16 * 16; }          // stored capsule rows
__host__ __device__ inline int cm_np(int N) { return ((N + 3) / 4 * 4) | 1; }         // pitch of the [h][n] arrays (>= N+3, odd)
__host__ __device__ inline int cm_hsp(int HS) { return (HS + 15) / 16 * 16; }

// type 1: S[h][c] = sum_n cs[h][n] P[n][c].  A = cs[h][n] (lane i=h, kk=n offset), B = P[n][16*ct + j].
template <int C>
__device__ __forceinline__ void cm_type1(const float* __restrict__ Ps, const float* __restrict__ cs, float* __restrict__ S, int N,
                                         int NP, int HSP) {
    constexpr int P = Tile<C>::PITCH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int nsteps = (N + 3) / 4;
    for (int job = wave; job < (HSP / 16) * (C / 16); job += CM_NW) {
        const int ht = job / (C / 16), ct = job % (C / 16);
        const float* arow = cs + (ht * 16 + j) * NP + kk;
        const float* brow = Ps + kk * P + ct * 16 + j;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 3 < nsteps; s += 4) {                          // 8 LDS reads in flight, two accumulator chains
            const float a0 = arow[4 * s], a1 = arow[4 * s + 4], a2 = arow[4 * s + 8], a3 = arow[4 * s + 12];
            const float b0 = brow[4 * s * P], b1 = brow[(4 * s + 4) * P], b2 = brow[(4 * s + 8) * P], b3 = brow[(4 * s + 12) * P];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc1, 0, 0, 0);
        }
        for (; s < nsteps; ++s) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * s], brow[4 * s * P], acc0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(ht * 16 + kk * 4 + r) * C + ct * 16 + j] = acc0[r] + acc1[r];
    }
}

// type 2: bl[h][n] += sum_c V[h][c] P[n][c]
template <int C>
__device__ __forceinline__ void cm_type2(const float* __restrict__ Ps, const float* __restrict__ Vs, float* __restrict__ bl, int N,
                                         int NP, int HS, int HSP) {
    constexpr int P = Tile<C>::PITCH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int ntiles = (N + 15) / 16;
    for (int job = wave; job < (HSP / 16) * ntiles; job += CM_NW) {
        const int ht = job / ntiles, nt = job % ntiles;
        const float* arow = Vs + (ht * 16 + j) * P + 4 * kk;
        const float* brow = Ps + (nt * 16 + j) * P + 4 * kk;
        float4 a[C / 16], b[C / 16];
#pragma unroll
        for (int q = 0; q < C / 16; ++q) { a[q] = ld4(arow + 16 * q); b[q] = ld4(brow + 16 * q); }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < C / 16; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b[q].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b[q].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b[q].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b[q].w, acc1, 0, 0, 0);
        }
        const int n = nt * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = ht * 16 + kk * 4 + r;
            if (h < HS && n < N) bl[h * NP + n] += acc0[r] + acc1[r];
        }
    }
}

// cs[h][n] = softmax_h( use_bl*bl + use_l0*(teb . adj) ): L lanes per node, lane q takes h = q, q+L, ... (at most UB of them);
// optional copy to global c_out[h*N + n].  The kernel picks (L, UB) = (2, 8) when N <= 256 and HS <= 16 — all nodes in ONE pass of
// the 512 threads — and the generic (4, 16) otherwise (HS <= 64, several passes).
// l0r (may be NULL): the thread's logits l0g[h][n] of the FIRST node pass, preloaded into registers by cm_softmax_preload (the global loads
// inside the loop were eight serialised L2 round trips per call; the kernel calls this twice with use_l0)
template <int L, int UB>
__device__ __forceinline__ void cm_softmax_preload(float (&l0r)[UB], const float* __restrict__ l0g, int N, int HS) {
    const int q = threadIdx.x & (L - 1), n = threadIdx.x / L;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const int h = q + L * u;
        l0r[u] = (L * u < HS && h < HS && n < N) ? l0g[(size_t)h * N + n] : 0.f;
    }
}
template <int L, int UB>
__device__ __forceinline__ void cm_softmax(const float* __restrict__ bl, float* __restrict__ cs, const float* __restrict__ l0g,
                                           float* __restrict__ c_out, int N, int NP, int HS, bool use_bl, bool use_l0,
                                           const float* l0r = nullptr) {
    const int q = threadIdx.x & (L - 1);
    for (int n0 = 0; n0 < N; n0 += CM_NT / L) {
        const int n = n0 + threadIdx.x / L;
        const bool valid = n < N;
        float l[UB];
        float m = -3.0e38f;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            l[u] = -3.0e38f;
            if (L * u < HS) {                                    // uniform
                const int h = q + L * u;
                if (h < HS && valid) {
                    float v = use_bl ? bl[h * NP + n] : 0.f;
                    if (use_l0) v += (l0r != nullptr && n0 == 0) ? l0r[u] : l0g[(size_t)h * N + n];
                    l[u] = v;
                    m = fmaxf(m, v);
                }
            }
        }
        m = group_max<L>(m);
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int h = q + L * u;
            if (L * u < HS && h < HS && valid) { l[u] = __expf(l[u] - m); sum += l[u]; }     // v_exp_f32 path: rel. error ~2e-6 for |x| < 50
        }
        sum = group_sum<L>(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int h = q + L * u;
            if (L * u < HS && h < HS && valid) {
                const float c = l[u] * inv;
                cs[h * NP + n] = c;
                if (c_out != nullptr) c_out[(size_t)h * N + n] = c;
            }
        }
    }
}

// consume S: post 0 -> dst_global[h*C+c] = S;  1 -> Vdst = squash(S);  2 -> Vdst = squash(V0 (.) S)
// sreg != NULL: the thread's S values come from a register (first routing pass, see cap_route_fwd2_kernel; needs HS * C / 4 <= CM_NT)
template <int C>
__device__ __forceinline__ void cm_post(const float* __restrict__ S, const float* __restrict__ V0s, float* __restrict__ Vdst,
                                        float* __restrict__ gdst, int HS, int post, const float4* sreg = nullptr) {
    constexpr int P = Tile<C>::PITCH, LPR = C / 4;
    for (int base = 0; base < HS * LPR; base += CM_NT) {
        const int pair = base + threadIdx.x;
        const bool valid = pair < HS * LPR;
        const int h = valid ? pair / LPR : 0, c4 = pair % LPR;
        float4 v = f4zero();
        if (valid) {
            v = sreg ? *sreg : ld4(S + h * C + 4 * c4);
            if (post == 2) { const float4 m = ld4(V0s + h * P + 4 * c4); v = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w); }
        }
        if (post != 0) {
            const float sc = squash_scale(group_sum<LPR>(f4dot(v, v)));
            v = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
        }
        if (valid) {
            if (post == 0) st4(gdst + h * C + 4 * c4, v);
            else st4(Vdst + h * P + 4 * c4, v);
        }
    }
}

// C = 64: one 16-row tile of  Y = X Wp^T + bp  with register-resident operands (same scheme as apply64_kernel): A fragments are
// float4s straight from global (lane (j,kk): row tile*16+j, channels 16q+4kk..+3), B fragments bv[q][e] come from the staged
// Wl[k][col] (components = column tile ct <-> channel 4j+ct).  On return y[r] holds row tile*16 + kk*4 + r, channels 4j..4j+3.
__device__ __forceinline__ void cm_fetch_a16(float4 (&a)[4], const float* __restrict__ Xbt, int tile, int N, int j, int kk) {
    const float* row = Xbt + (size_t)min(tile * 16 + j, N - 1) * 64 + 4 * kk;
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = ld4(row + 16 * q);
}
__device__ __forceinline__ void cm_load_bfrag(float4 (&bv)[4][4], const float* __restrict__ Wl, int j, int kk) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[q][e] = ld4(Wl + (16 * q + 4 * kk + e) * 64 + 4 * j);
}
__device__ __forceinline__ void cm_tile16(const float4 (&a)[4], const float4 (&bv)[4][4], float4 b4, float4 (&y)[4]) {
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].w, acc[3], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = f4add(make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), b4);
}

// one batch of float4 loads for the 32 x C row tile t of a (b,t) slice; rows beyond N are clamped (and zeroed when staged)
template <int C>
__device__ __forceinline__ void cm_fetch_x(float4 (&xv)[Tile<C>::F4_PER_LANE], const float* __restrict__ Xbt, int t, int N, int lane) {
#pragma unroll
    for (int it = 0; it < Tile<C>::F4_PER_LANE; ++it) {
        const int f = it * 64 + lane;
        const int r = f / Tile<C>::F4_PER_ROW, c4 = f % Tile<C>::F4_PER_ROW;
        xv[it] = ld4(Xbt + (size_t)min(t * 32 + r, N - 1) * C + 4 * c4);
    }
}

template <int C, int SL, int SU>
__global__ __launch_bounds__(CM_NT, 4) void cap_route_fwd2_kernel(const float* __restrict__ X, const float* __restrict__ Wp,
                                                               const float* __restrict__ bp, const float* __restrict__ dadj,
                                                               float* __restrict__ c_out,
                                                               float* __restrict__ s_out, int N, int HS, int R, int region2, int dbg) {
    using T = Tile<C>;
    constexpr int P = T::PITCH, LPR = C / 4, RPP = CM_NT / LPR;
    int tsi = 0;
#ifdef GPTST_DEBUG
#define TS() do { if (dbg == 99 && blockIdx.x == 5 && threadIdx.x == 0) g_cap_ts[tsi] = clock64(); ++tsi; } while (0)
#else
#define TS() SB()
#endif
    TS();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NR = cm_rows(N), NP = cm_np(N), HSP = cm_hsp(HS);
    float* Ps = smem;                       // NR * P
    float* Wl = Ps + NR * P;                // region2 = max(C*C, HS*NP + HSP*NP)
    float* bl = Wl;                         // HS * NP
    float* cs = bl + HS * NP;               // HSP * NP
    float* Vs = Wl + region2;               // HSP * P
    float* V0s = Vs + HSP * P;              // HSP * P
    float* S = V0s + HSP * P;               // HSP * C
    const int bt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* Xbt = X + (size_t)bt * N * C;
    const float* l0g = dadj + (size_t)bt * HS * N;
    float l0r[SU];
    cm_softmax_preload<SL, SU>(l0r, l0g, N, HS);                   // in flight during the capsule GEMM
    // First routing pass: b = 0 (GPTST.py:112), so c = softmax_h(0) = 1/HS for every node and s[h] = v0[h] (.) (1/HS) sum_n P[n]: no
    // softmax, no contraction — the column sums of P fall out of the capsule GEMM's epilogue.
    const bool uni = C == 64 && R > 0 && HS * LPR <= CM_NT;
    float4 u0 = f4zero();

    if constexpr (C == 64) {
        // ---- P = squash(X Wp^T + bp): 16-row tiles, register operands, squash fused into the MFMA epilogue (row norm = 16-lane
        //      DPP sum), P written to LDS once.  Replaces: X tile -> LDS -> MFMA 32x32x2 -> LDS -> separate squash pass + barrier. ----
        const int j = lane & 15, kk = lane >> 4;
        float4 a[4];
        cm_fetch_a16(a, Xbt, wave, N, j, kk);                      // in flight while the weight is staged
        const float4 b4 = ld4(bp + 4 * j);
        load_w_lds<C, CM_NT>(Wl, Wp, 1, tid);
        for (int i = tid; i < 2 * HSP * P + HSP * C; i += CM_NT) Vs[i] = 0.f;      // Vs, V0s, S
        __syncthreads(); TS();
        float4 bv[4][4];
        cm_load_bfrag(bv, Wl, j, kk);
        float4 csum = f4zero();                                    // column sums of this wave's rows of P (first routing pass, below)
        for (int tile = wave; tile < NR / 16; tile += CM_NW) {
            if (tile != wave) cm_fetch_a16(a, Xbt, tile, N, j, kk);
            float4 y[4];
            cm_tile16(a, bv, b4, y);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = tile * 16 + kk * 4 + r;
                float4 v = n < N ? y[r] : f4zero();
                const float sc = squash_scale(group_sum<16>(f4dot(v, v)));
                v = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
                st4(Ps + n * P + 4 * j, v);
                csum = f4add(csum, v);
            }
        }
        if (uni) {                                                 // S rows 0..7 <- the waves' partial column sums (S is first written by c0 . P)
            csum.x += __shfl_xor(csum.x, 16); csum.y += __shfl_xor(csum.y, 16); csum.z += __shfl_xor(csum.z, 16); csum.w += __shfl_xor(csum.w, 16);
            csum.x += __shfl_xor(csum.x, 32); csum.y += __shfl_xor(csum.y, 32); csum.z += __shfl_xor(csum.z, 32); csum.w += __shfl_xor(csum.w, 32);
            if (kk == 0) st4(S + wave * 64 + 4 * j, csum);
        }
        __syncthreads(); TS();                                     // all B fragments are in registers: Wl may be recycled
        // b = 0 and the padding of c (rows >= HS, columns >= N) — the cells the softmax below writes are left alone, so that it needs no
        // barrier in between
        for (int i = tid; i < (HS + HSP) * NP; i += CM_NT) {
            const int hc = i / NP - HS, n = i % NP;
            if (hc < 0 || hc >= HS || n >= N) bl[i] = 0.f;
        }
        if (uni && tid < HS * LPR) {                               // this thread's (h, 4 channels) of the first routing pass: (1/HS) sum_n P[n]
            const int c4 = tid % LPR;
#pragma unroll
            for (int w = 0; w < CM_NW; ++w) u0 = f4add(u0, ld4(S + w * 64 + 4 * c4));       // fixed order
            const float inv = 1.f / (float)HS;
            u0 = make_float4(u0.x * inv, u0.y * inv, u0.z * inv, u0.w * inv);
        }
        TS();
    } else {
    float4 xv[T::F4_PER_LANE];              // this wave's X tile: requested before the weight is staged (one batch, clamped rows)
    cm_fetch_x<C>(xv, Xbt, wave, N, lane);
    load_w_lds<C, CM_NT>(Wl, Wp, 1, tid);
    for (int i = tid; i < 2 * HSP * P + HSP * C; i += CM_NT) Vs[i] = 0.f;      // Vs, V0s, S
    __syncthreads(); TS();
    // ---- Y = X Wp^T by 32-row MFMA tiles (one per wave); rows beyond the stored range are dropped ----------------
    for (int t = wave; t < (NR + 31) / 32; t += CM_NW) {
        float* tile = Ps + t * 32 * P;
        const int rows_here = min(32, NR - t * 32);
        if (t != wave) cm_fetch_x<C>(xv, Xbt, t, N, lane);
#pragma unroll
        for (int it = 0; it < T::F4_PER_LANE; ++it) {
            const int f = it * 64 + lane;
            const int r = f / T::F4_PER_ROW, c4 = f % T::F4_PER_ROW;
            const int n = t * 32 + r;
            if (r < rows_here) st4(tile + r * P + 4 * c4, n < N ? xv[it] : f4zero());
        }
        f32x16 acc[T::NCT];
        if (rows_here == 32) {
            mfma_tile<C>(tile, Wl, acc, lane);
            acc_to_tile<C>(tile, acc, lane);
        } else {                                        // 16-row tail: operand rows 16..31 alias rows 0..15 (results discarded)
            const int i = lane & 31, h = lane >> 5;
#pragma unroll
            for (int ct = 0; ct < T::NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
            const float* arow = tile + (i & 15) * P + 4 * h;
            const float* wcol = Wl + 4 * h * C + i;
#pragma unroll 2
            for (int q = 0; q < C / 8; ++q) {
                const float4 a4 = ld4(arow + 8 * q);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int ct = 0; ct < T::NCT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], wcol[(8 * q + jj) * C + ct * 32], acc[ct], 0, 0, 0);
            }
#pragma unroll
            for (int ct = 0; ct < T::NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < 16) tile[row * P + ct * 32 + i] = acc[ct][r];
                }
        }
    }
    __syncthreads(); TS();
    // ---- P = squash(Y + bp) in place;  bl = 0, cs = 0 (incl. padding rows / columns) ---------------------------
    {
        const int slot = tid / LPR, jj = tid % LPR;
        const float4 b4 = ld4(bp + 4 * jj);
        for (int n0 = 0; n0 < NR; n0 += RPP) {
            const int n = n0 + slot;
            float4 y = f4zero();
            if (n < N) y = f4add(ld4(Ps + n * P + 4 * jj), b4);
            const float sc = squash_scale(group_sum<LPR>(f4dot(y, y)));
            if (n < NR) st4(Ps + n * P + 4 * jj, make_float4(y.x * sc, y.y * sc, y.z * sc, y.w * sc));
        }
        for (int i = tid; i < (HS + HSP) * NP; i += CM_NT) bl[i] = 0.f;
    }
    __syncthreads(); TS();
    }
    cm_softmax<SL, SU>(bl, cs, l0g, nullptr, N, NP, HS, false, true, l0r);  // c0 = softmax_h(dadj)      :105
    __syncthreads(); TS();
    cm_type1<C>(Ps, cs, S, N, NP, HSP);
    __syncthreads(); TS();
    cm_post<C>(S, nullptr, V0s, nullptr, HS, 1);                                        // v0 = squash(c0 . P)       :105-106
    __syncthreads(); TS();
    for (int r = 0; r < R; ++r) {                                                       // routing (no grad)          :113-118
        if (r > 0) { cm_type2<C>(Ps, Vs, bl, N, NP, HS, HSP); __syncthreads(); TS(); }        // b += v . P^T
        if (r == 0 && uni) {
            cm_post<C>(S, V0s, Vs, nullptr, HS, 2, &u0);                                // v = squash(v0 (.) mean-over-classes . P)
            __syncthreads(); TS();
            continue;
        }
        cm_softmax<SL, SU>(bl, cs, l0g, nullptr, N, NP, HS, true, false);   // c = softmax_h(b)
        __syncthreads(); TS();
        cm_type1<C>(Ps, cs, S, N, NP, HSP);
        __syncthreads(); TS();
        cm_post<C>(S, V0s, Vs, nullptr, HS, 2);                                         // v = squash(v0 (.) c.P)
        __syncthreads(); TS();
    }
    if (R > 0) { cm_type2<C>(Ps, Vs, bl, N, NP, HS, HSP); __syncthreads(); TS(); }
    cm_softmax<SL, SU>(bl, cs, l0g, c_out + (size_t)bt * HS * N, N, NP, HS, true, true, l0r);   // c = softmax_h(b + dadj)  :120
    __syncthreads(); TS();
    cm_type1<C>(Ps, cs, S, N, NP, HSP);
    __syncthreads(); TS();
    cm_post<C>(S, nullptr, nullptr, s_out + (size_t)bt * HS * C, HS, 0);                // s = c . P                 :123
}

template <int C>
static int launch_route_fwd2(const float* X, const float* Wp, const float* bp, const float* dadj, float* c_out,
                             float* s_out, int BT, int N, int HS, int R, hipStream_t st) {
    if (HS > 64) return GPTST_ESHAPE;
    const int NR = cm_rows(N), NP = cm_np(N), HSP = cm_hsp(HS);
    size_t r2 = (size_t)C * C, need = (size_t)(HS + HSP) * NP;
    if (need > r2) r2 = need;
    r2 = (r2 + 3) & ~(size_t)3;
    const size_t smem = ((size_t)NR * Tile<C>::PITCH + r2 + 2 * (size_t)HSP * Tile<C>::PITCH + (size_t)HSP * C) * sizeof(float);
    if (smem > 160 * 1024) return GPTST_ESHAPE;
    static size_t cur[2] = {0, 0};
    const bool one_pass = N <= CM_NT / 2 && HS <= 16;              // softmax: 2 lanes per node, all nodes in one pass
    if (one_pass) {
        if (smem > cur[0]) { hipFuncSetAttribute((const void*)cap_route_fwd2_kernel<C, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); cur[0] = smem; }
        hipLaunchKernelGGL((cap_route_fwd2_kernel<C, 2, 8>), dim3(BT), dim3(CM_NT), smem, st, X, Wp, bp, dadj, c_out, s_out, N, HS, R, (int)r2, g_cap_dbg);
    } else {
        if (smem > cur[1]) { hipFuncSetAttribute((const void*)cap_route_fwd2_kernel<C, 4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); cur[1] = smem; }
        hipLaunchKernelGGL((cap_route_fwd2_kernel<C, 4, 16>), dim3(BT), dim3(CM_NT), smem, st, X, Wp, bp, dadj, c_out, s_out, N, HS, R, (int)r2, g_cap_dbg);
    }
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

GPTST_INTERNAL int gptst_cap_route_fwd_v1(const float* X, const float* Wp, const float* bp, const float* dadj, float* c_out, float* s_out,
                                      int BT, int N, int C, int HS, int R, void* stream);
GPTST_INTERNAL int gptst_cap_route_fwd3(const float* X, const float* Wp, const float* bp, const float* dadj, float* c_out, float* s_out,
                                        int BT, int N, int C, int HS, int R, void* stream);      // cap_route3.hip (one wave per node tile)

extern "C" int gptst_cap_route_fwd(const float* X, const float* Wp, const float* bp, const float* dadj, float* c_out, float* s_out,
                                   int BT, int N, int C, int HS, int R, void* stream) {
    if (!X || !Wp || !bp || !dadj || !c_out || !s_out || HS <= 0 || R < 0) return GPTST_EARG;
    int rc = gptst_cap_route_fwd3(X, Wp, bp, dadj, c_out, s_out, BT, N, C, HS, R, stream);
    if (rc != GPTST_ESHAPE) return rc;
    if (C == 64) rc = launch_route_fwd2<64>(X, Wp, bp, dadj, c_out, s_out, BT, N, HS, R, (hipStream_t)stream);
    else if (C == 128) rc = launch_route_fwd2<128>(X, Wp, bp, dadj, c_out, s_out, BT, N, HS, R, (hipStream_t)stream);
    if (rc == GPTST_ESHAPE) return gptst_cap_route_fwd_v1(X, Wp, bp, dadj, c_out, s_out, BT, N, C, HS, R, stream);
    return rc;
}

#ifdef GPTST_DEBUG
// resident workgroups per CU of the forward kernel at a given shape
extern "C" int gptst_cap_occupancy(int N, int HS) {
    constexpr int C = 64;
    const int NR = cm_rows(N), NP = cm_np(N), HSP = cm_hsp(HS);
    size_t r2 = (size_t)C * C, need = (size_t)(HS + HSP) * NP;
    if (need > r2) r2 = need;
    r2 = (r2 + 3) & ~(size_t)3;
    const size_t smem = ((size_t)NR * Tile<C>::PITCH + r2 + 2 * (size_t)HSP * Tile<C>::PITCH + (size_t)HSP * C) * sizeof(float);
    int n = -1;
    hipFuncSetAttribute((const void*)cap_route_fwd2_kernel<C, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)cap_route_fwd2_kernel<C, 2, 8>, CM_NT, smem);
    return n * 1000000 + (int)smem;
}
#endif

// =====================================================================================================================
// backward through s = c.P, c = softmax_h(b + dadj), P = squash(X Wp^T + bp)   (routing logits b are detached, GPTST.py:108-109)
//   in : X, Wp, bp, c (BT,HS,N), dc1 (BT,HS,N: grad of c from the cluster->node scatter), dS (BT,HS,C: total grad of s)
//   out: dY (BT*N, C) grad of the pre-squash Linear output, dlogit (BT,HS,N) grad of dadj
// One 512-thread workgroup per (b,t).  Y = X Wp^T + bp is rebuilt by MFMA into LDS (cheaper than a 16.7 MB round trip); then
// every wave owns 16-node tiles end to end, with no further workgroup barrier:
//   U[h,n] = dS[h,:].P[n,:]  (MFMA 16x16x4, P = g(n) Y)  ->  dc = dc1 + U,  dlogit = c (dc - sum_h c dc)
//   dP[n,:] = sum_h c[h,n] dS[h,:]  (MFMA 16x16x4, K = clusters)  ->  squash backward  dY = g dP + Y (2 g'(q) (Y.dP))
// =====================================================================================================================
// =====================================================================================================================
// backward of the cluster -> node scatter rec[n,:] = sum_h c[h,n] v[h,:]  (GPTST.py:135):
//   dc1[h,n] = drec[n,:] . v[h,:]   (type 2, "V" = v)        dv[h,:] = sum_n c[h,n] drec[n,:]   (type 1, "P" = drec)
// drec rows of one (b,t) are staged in LDS once; both contractions on MFMA 16x16x4.
// =====================================================================================================================
// PUB: dc1 / dv are consumed by OTHER workgroups of the same launch (the roles of cap_route_bwd2_kernel): write-through (agent-scope) stores.
template <int C, bool PUB>
__device__ __forceinline__ void cap_rec_bwd2_body(const float* __restrict__ drec, const float* __restrict__ c, const float* __restrict__ v,
                                                  float* __restrict__ dc1, float* __restrict__ dv, int N, int HS, int bt, float* __restrict__ smem) {
    constexpr int P = Tile<C>::PITCH, LPR = C / 4;
    const int NR = cm_rows(N), NP = cm_np(N), HSP = cm_hsp(HS);
    float* Ds = smem;                       // NR * P   drec rows (rows >= N zero)
    float* cs = Ds + NR * P;                // HSP * NP
    float* bl = cs + HSP * NP;              // HS * NP  dc1 accumulator
    float* Vs = bl + HS * NP;               // HSP * P  v
    float* S = Vs + HSP * P;                // HSP * C  dv
    const int tid = threadIdx.x;
    // Loads are issued in batches ahead of the LDS stores (a copy loop costs one serialised L2 round trip per trip).
    const int ncn = HS * N, nv = HS * LPR, nD = N * LPR;
    const float* cb = c + (size_t)bt * ncn;
    float cr[4];
    float4 vr;
#pragma unroll
    for (int u = 0; u < 4; ++u) cr[u] = cb[min(u * CM_NT + tid, ncn - 1)];
    vr = ld4(v + (size_t)bt * HS * C + 4 * (size_t)min(tid, nv - 1));
#ifndef CRB_DREC_BATCH
#define CRB_DREC_BATCH 6          // float4 loads per thread and round trip of the drec staging (N = 170, C = 64: 5.5 per thread -> ONE trip; 4: 892.0 -> 6: 894.4 steps/s, r06)
#endif
    for (int i0 = 0; i0 < NR * LPR; i0 += CRB_DREC_BATCH * CM_NT) {
        float4 d4[CRB_DREC_BATCH];
#pragma unroll
        for (int u = 0; u < CRB_DREC_BATCH; ++u) d4[u] = ld4(drec + (size_t)bt * N * C + 4 * (size_t)min(i0 + u * CM_NT + tid, nD - 1));
        SB();
#pragma unroll
        for (int u = 0; u < CRB_DREC_BATCH; ++u) {
            const int i = i0 + u * CM_NT + tid;
            if (i < NR * LPR) st4(Ds + (i / LPR) * P + 4 * (i % LPR), i < nD ? d4[u] : f4zero());
        }
    }
    for (int i = tid; i < HSP * NP + HS * NP + HSP * P; i += CM_NT) cs[i] = 0.f;      // cs, bl, Vs
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = u * CM_NT + tid;
        if (i < ncn) cs[(i / N) * NP + i % N] = cr[u];
    }
    for (int i = 4 * CM_NT + tid; i < ncn; i += CM_NT) cs[(i / N) * NP + i % N] = cb[i];
    if (tid < nv) st4(Vs + (tid / LPR) * P + 4 * (tid % LPR), vr);
    for (int i = CM_NT + tid; i < nv; i += CM_NT) st4(Vs + (i / LPR) * P + 4 * (i % LPR), ld4(v + (size_t)bt * HS * C + 4 * i));
    __syncthreads();
    cm_type2<C>(Ds, Vs, bl, N, NP, HS, HSP);
    cm_type1<C>(Ds, cs, S, N, NP, HSP);
    __syncthreads();
    if constexpr (PUB) {
        for (int i = tid; i < HS * N; i += CM_NT) st_agent(dc1 + (size_t)bt * HS * N + i, bl[(i / N) * NP + i % N]);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dv + (size_t)bt * HS * C, 0, HS * C * 4, 0x00020000);
        for (int i = tid; i < HS * LPR; i += CM_NT) st4_sc1(rs, 16 * i, ld4(S + (i / LPR) * C + 4 * (i % LPR)));
    } else {
        for (int i = tid; i < HS * N; i += CM_NT) dc1[(size_t)bt * HS * N + i] = bl[(i / N) * NP + i % N];
        for (int i = tid; i < HS * LPR; i += CM_NT) st4(dv + (size_t)bt * HS * C + 4 * i, ld4(S + (i / LPR) * C + 4 * (i % LPR)));
    }
}

template <int C>
__global__ __launch_bounds__(CM_NT, 4) void cap_rec_bwd2_kernel(const float* __restrict__ drec, const float* __restrict__ c,
                                                                const float* __restrict__ v, float* __restrict__ dc1,
                                                                float* __restrict__ dv, int N, int HS) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    cap_rec_bwd2_body<C, false>(drec, c, v, dc1, dv, N, HS, blockIdx.x, smem);
}


#define CX_SPLIT 4        // cross-time role: workgroups per sample (each owns 12 / CX_SPLIT time steps; the whole-sample part is repeated by them)
struct CrossBwdArgs { const float* dv; const float* s; const float* Rt; const float* Ht; const float* dyn; const float* tmpl; float* ddyn; int T, HT;
                      float* dSg; unsigned* flags; int nB; };    // roles (r04): nB cross-time workgroups publish dS (B*T, HS, C) + their flags

// r05: the backward of the cap's ENTRY Linear + the layer's residual branch (GPTST.py:102,139-141; gptst_linear_bwd) folded into the routing backward,
// which holds everything it needs: the X tile (it rebuilds Y from it), dY (in LDS, never written out) and — fetched as fragments — Wp.
//   dX = dY Wp + dPre             (out == NULL: the incoming gradient already is dPre;  premul: the sum is multiplied by lrelu'(X))
//   dX = dY Wp + dOut lrelu'(out) (out given)
//   dWp[bt] = dY^T X ([out][in]), dbp[bt] = colsum(dY): ONE partial per (b,t) workgroup (B*T rows for the reduction job; 510 row splits before)
// dX == NULL: not folded (dY goes to global memory for gptst_linear_bwd).
struct LinArgs { const float* dPre; const float* out; float* dX; float* dWp; float* dbp; int premul; };

// ---- backward of the cross-time block (cap_cross_bwd_kernel, cap_cross.hip) as a PROLOGUE of the (b,t) workgroups below (r03) -----------
// cap_cross_bwd runs on B workgroups between two (b,t)-grouped kernels.  Folded in, every (b,t) workgroup repeats the part that needs the
// whole sample — du / dRpre of all T*HS tokens and dHpre (HT x C over the tokens), ~0.3 MFLOP — and then produces only what belongs to its
// own HS tokens: dS rows (straight into the LDS tile the routing backward reads: no global round trip) and the ddyn columns.
// Scratch: the Ys / Wl regions of the kernel below (free until its first phase).  Same arithmetic as cap_cross_bwd_kernel.
// GLOBAL (r04, the cross-time ROLE of the kernel below): the workgroup owns the tokens of time steps t0 .. t0 + nt - 1 of the sample (nt = 1 and
// dS into the LDS tile Vs in the prologue form) and writes their dS rows THROUGH to dSg (sc1 stores: the sample's routing workgroups read them in
// this launch).  The part that needs the whole sample (du / dRpre of every token, dHpre) is repeated by the workgroups of a sample either way.
template <int C, bool GLOBAL, bool ACQ = false>     // ACQ: dv was produced in THIS launch (three-role form): agent-scope loads
__device__ __forceinline__ void cap_cross_bwd_prologue(const float* __restrict__ dv, const float* __restrict__ s, const float* __restrict__ Rt,
                                                       const float* __restrict__ Ht, const float* __restrict__ dyn,
                                                       const float* __restrict__ tmpl, float* __restrict__ ddyn, float* __restrict__ scratch,
                                                       float* __restrict__ Vs, int b, int t0, int nt, int T, int HS, int HT,
                                                       float* __restrict__ dSg = nullptr) {
    constexpr int P = Tile<C>::PITCH, LPR = C / 4;
    const int KK = T * HS, tid = threadIdx.x;
    const int k0 = t0 * HS, nown = nt * HS;    // own tokens: k0 .. k0 + nown - 1
    float* Gs = scratch;                       // KK * P   dRpre
    float* Hs = Gs + KK * P;                   // HT * P   Ht
    float* dHs = Hs + HT * P;                  // HT * P   dHpre
    float* Zo = dHs + HT * P;                  // nown * P Z rows of the own tokens
    float* Uo = Zo + nown * P;                 // nown * P du rows of the own tokens (GLOBAL; the prologue form keeps them in Vs)
    float* dyns = Uo + (GLOBAL ? nown * P : 0);   // HT * KK
    // rows k: u = Rt + s;  du = squash_bwd(u, dv);  dRpre = du * lrelu'(Rt);  own rows: du -> Vs[h] / Uo, Z -> Zo.
    // All global loads of a batch are issued before the first LDS store / use (a copy loop is one serialised L2 round trip per trip).
    {
        const int nz = KK * LPR, nd = HT * KK / 4, nh = HT * LPR;
        const float* db = dyn + (size_t)b * HT * KK;
        const float* hb = Ht + (size_t)b * HT * C;
        const __amdgpu_buffer_rsrc_t rs_dv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dv) + (size_t)b * KK * C, 0, KK * C * 4, 0x00020000);
        for (int i0 = 0; i0 < nz; i0 += 4 * CM_NT) {
            float4 sv4[4], rt4[4], g4[4], dv1, hv1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int il = min(i0 + u * CM_NT + tid, nz - 1);
                const size_t off = (size_t)b * KK * C + 4 * (size_t)il;
                sv4[u] = ld4(s + off); rt4[u] = ld4(Rt + off); g4[u] = ACQ ? ld4_sc1(rs_dv, 16 * il) : ld4(dv + off);
            }
            if (i0 == 0) { dv1 = ld4(db + 4 * (size_t)min(tid, nd - 1)); hv1 = ld4(hb + 4 * (size_t)min(tid, nh - 1)); }
            SB();
            if (i0 == 0) {
                if (tid < nd) st4(dyns + 4 * tid, dv1);
                if (tid < nh) st4(Hs + (tid / LPR) * P + 4 * (tid % LPR), hv1);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * CM_NT + tid;
                const bool valid = i < nz;
                const int k = valid ? i / LPR : 0, c4 = i % LPR;
                const float4 sv = valid ? sv4[u] : f4zero(), rt = valid ? rt4[u] : f4zero(), g = valid ? g4[u] : f4zero();
                const float4 u_ = f4add(rt, sv);
                const float q = group_sum<LPR>(f4dot(u_, u_));
                const float udg = group_sum<LPR>(f4dot(u_, g));
                const float r = sqrtf(q), den = (1.f + q) * (r + 1e-8f);
                const float gq = q / den;
                float gp = 0.f;
                if (r > 0.f) gp = (den - q * ((r + 1e-8f) + (1.f + q) * 0.5f / r)) / (den * den);
                const float k2 = 2.f * gp * udg;
                const float4 du = make_float4(fmaf(k2, u_.x, gq * g.x), fmaf(k2, u_.y, gq * g.y), fmaf(k2, u_.z, gq * g.z), fmaf(k2, u_.w, gq * g.w));
                if (valid) {
                    st4(Gs + k * P + 4 * c4, make_float4(du.x * lrelu_grad_from_out(rt.x), du.y * lrelu_grad_from_out(rt.y),
                                                         du.z * lrelu_grad_from_out(rt.z), du.w * lrelu_grad_from_out(rt.w)));
                    if (k >= k0 && k < k0 + nown) {
                        const float tm = tmpl[k / HS];
                        st4((GLOBAL ? Uo : Vs) + (k - k0) * P + 4 * c4, du);
                        st4(Zo + (k - k0) * P + 4 * c4, make_float4(sv.x + tm, sv.y + tm, sv.z + tm, sv.w + tm));
                    }
                }
            }
        }
        for (int i = CM_NT + tid; i < nd; i += CM_NT) st4(dyns + 4 * i, ld4(db + 4 * (size_t)i));             // (beyond the bench shape)
        for (int i = CM_NT + tid; i < nh; i += CM_NT) st4(Hs + (i / LPR) * P + 4 * (i % LPR), ld4(hb + 4 * (size_t)i));
    }
    __syncthreads();
    // dHpre[j] = lrelu'(Ht[j]) * sum_k dyn[j][k] dRpre[k]   on MFMA 16x16x4: waves 0-3 = column tiles, KK/4 dependent steps (the fmaf chain over k
    // in order: the same numbers as the VALU loop of cap_cross_bwd_kernel), six steps' LDS operands in flight
    {
        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kk = lane >> 4;
        if (wave < C / 16) {
            for (int jt = 0; jt < HT; jt += 16) {
                const int ja = jt + li;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* ar = dyns + min(ja, HT - 1) * KK + kk;
                const float* br = Gs + kk * P + 16 * wave + li;
                const int ns4 = KK / 4;
                int s4 = 0;
                for (; s4 + 6 <= ns4; s4 += 6) {
                    float av[6], bv[6];
#pragma unroll
                    for (int u = 0; u < 6; ++u) { av[u] = ja < HT ? ar[4 * (s4 + u)] : 0.f; bv[u] = br[4 * (s4 + u) * P]; }
#pragma unroll
                    for (int u = 0; u < 6; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
                }
                for (; s4 < ns4; ++s4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ja < HT ? ar[4 * s4] : 0.f, br[4 * s4 * P], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jj = jt + 4 * kk + r;
                    if (jj < HT) dHs[jj * P + 16 * wave + li] = acc[r] * lrelu_grad_from_out(Hs[jj * P + 16 * wave + li]);
                }
            }
        }
    }
    __syncthreads();
    // own tokens: ddyn[j][k] = Ht[j].dRpre[k] + dHpre[j].Z[k]
    for (int i = tid; i < HT * nown; i += CM_NT) {
        const int j = i / nown, h = i % nown, k = k0 + h;
        float acc = 0.f;
#pragma unroll 4
        for (int c4 = 0; c4 < LPR; ++c4) {
            acc += f4dot(ld4(Hs + j * P + 4 * c4), ld4(Gs + k * P + 4 * c4));
            acc += f4dot(ld4(dHs + j * P + 4 * c4), ld4(Zo + h * P + 4 * c4));
        }
        ddyn[((size_t)b * HT + j) * KK + k] = acc;
    }
    // own tokens: dS[k] = du[k] + sum_j dyn[j][k] dHpre[j]
    for (int i = tid; i < nown * LPR; i += CM_NT) {
        const int h = i / LPR, c4 = i % LPR, k = k0 + h;
        float4 acc = f4zero();
#pragma unroll 8
        for (int j = 0; j < HT; ++j) acc = f4fma(dyns[j * KK + k], ld4(dHs + j * P + 4 * c4), acc);
        if (GLOBAL) {
            // write-through (sc1): the routing workgroups of the sample read it in this launch (MI355X_MICROARCH.md, inter-workgroup visibility)
            typedef int i32x4_ __attribute__((ext_vector_type(4)));
            const float4 r4 = f4add(ld4(Uo + h * P + 4 * c4), acc);
            const f32x4 v4 = {r4.x, r4.y, r4.z, r4.w};
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dSg + ((size_t)b * KK + k0) * C, 0, nown * C * 4, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_, v4), rs, (h * C + 4 * c4) * 4, 0, 16);
        } else {
            st4(Vs + h * P + 4 * c4, f4add(ld4(Vs + h * P + 4 * c4), acc));
        }
    }
    __syncthreads();
}

// r05, late: REDUCTION JOBS as a further role.  The weight-gradient reductions of the layers ALREADY behind the step's backward (gptst_pool_jobs kinds 1 / 2: their
// inputs were written by earlier launches, so nothing is waited for) ride behind the routing workgroups: this launch runs 384 routing workgroups on 256 CUs at
// ~2.3 TB/s, the CUs whose cross-time role has finished and the spare HBM bandwidth take the jobs, and the 69-us reduction launch at the end of the step shrinks
// by what was carried.  A job workgroup = two 256-thread job blocks of the SAME kind (a pool-gradient block has one barrier: both halves reach it).
template <bool JOBS> struct RouteJobs { };
template <> struct RouteJobs<true> { PJobs t; int npool, npb, neb, first; };      // jobs [0, npool): kind 1 (npb blocks), the rest kind 2 (neb blocks); first job workgroup

template <int C, int ROLES, bool LIN = false, bool JOBS = false>      // ROLES 0: one role (dS given, or the cross-time backward as a prologue); 1: + cross-time role
                                                   // (r05: the three-role form — + the rec backward — measured slower and left the library: profiles/r04_roles3_stamps.txt)
__global__ __launch_bounds__(CM_NT, 4) void cap_route_bwd2_kernel(const float* __restrict__ X, const float* __restrict__ Wp,
                                                                  const float* __restrict__ bp, const float* __restrict__ c,
                                                                  const float* __restrict__ dc1, const float* __restrict__ dS,
                                                                  float* __restrict__ dY, float* __restrict__ dlogit, int N, int HS,
                                                                  int region2, CrossBwdArgs cx, LinArgs lin, RouteJobs<JOBS> jr, int nfull, int nhr) {
    if constexpr (JOBS) {
        if ((int)blockIdx.x >= jr.first) {
            extern __shared__ __attribute__((aligned(16))) float jsmem[];
            const int half = (int)(threadIdx.x >> 8), tid_ = (int)(threadIdx.x & 255);
            float (*fold)[PG_MAXK][65] = reinterpret_cast<float (*)[PG_MAXK][65]>(jsmem + half * (4 * PG_MAXK * 65));
            const int w = (int)blockIdx.x - jr.first, wp = (jr.npb + 1) >> 1;
            if (w < wp) {                                            // two pool-gradient blocks (or one and a half that only keeps the barrier)
                const int vb = __builtin_amdgcn_readfirstlane(2 * w + half);
                if (vb < jr.npb) {
                    int p = 0;
                    for (int q = 1; q < jr.npool; ++q) if (vb >= jr.t.j[q].blk0) p = q;
                    const PJob& a = jr.t.j[p];
                    if (((a.cols | a.ldx) & 3) == 0) pj_bwd_pool<4>(a, vb - a.blk0, fold, tid_); else pj_bwd_pool<1>(a, vb - a.blk0, fold, tid_);
                } else {
                    __syncthreads();
                }
            } else {
                const int vb = __builtin_amdgcn_readfirstlane(2 * (w - wp) + half);
                if (vb < jr.neb) {
                    int p = jr.npool;
                    for (int q = jr.npool + 1; q < jr.t.n; ++q) if (vb >= jr.t.j[q].blk0) p = q;
                    const PJob& a = jr.t.j[p];
                    const int rel = vb - a.blk0;
                    if (((a.cols | a.ldx) & 3) == 0) pj_bwd_emb<4>(a, rel % a.nbx, rel / a.nbx, tid_); else pj_bwd_emb<1>(a, rel % a.nbx, rel / a.nbx, tid_);
                }
            }
            return;
        }
    }
    using T = Tile<C>;
    constexpr int P = T::PITCH, LPR = C / 4, RPP = CM_NT / LPR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NR = cm_rows(N), NP = cm_np(N), HSP = cm_hsp(HS);
    float* Ys = smem;                       // NR * P     Y = X Wp^T + bp
    float* Wl = Ys + NR * P;                // region2 = max(C*C, 2*HSP*NP)
    float* cs = Wl;                         // HSP * NP   (rows >= HS zero)
    float* dcs = cs + HSP * NP;             // HSP * NP   dc1, then dc
    float* Vs = Wl + region2;               // HSP * P    dS (rows >= HS zero)
    float* gq = Vs + HSP * P;               // NR         squash factor g(n)
    float* qq = gq + NR;                    // NR         squared norm q(n)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- ROLES (r04): the backward of the cross-time block needs the whole sample, and as a prologue of every (b,t) workgroup it was repeated
    // T times and sat IN FRONT of each workgroup's chain (8-13 us of its 27-31, tools/phase_stamps.py).  In the role form the first nB workgroups
    // do it once per sample and publish dS (write-through stores, then one flag per sample), while the B*T routing workgroups rebuild their
    // capsule tile first — that GEMM does not depend on dS — and pick dS up behind it.  All nB + B*T workgroups are resident together (two per
    // CU, checked by the launcher); the wait is bounded (2 s wall clock, gptst_wait_ge): on expiry dS is poisoned with NaN, so a lost hand-off ends the run loudly (NaN
    // loss / gradient norm) instead of hanging the GPU or training on a wrong gradient.  A separate instantiation: with both forms in one
    // kernel the register allocator spilled 100 registers.
    unsigned* s_ok = reinterpret_cast<unsigned*>(qq + NR);
    GPTST_WG_BEGIN(); GPTST_STAMP(0);
    const int blk = (int)blockIdx.x;
    int ru;                                              // routing work unit of this workgroup
    if (ROLES && blk < cx.nB) {                          // cross-time role: workgroup r = (sample r / CX_SPLIT, time steps of part r % CX_SPLIT)
        constexpr int TS = 12 / CX_SPLIT;
        cap_cross_bwd_prologue<C, true>(cx.dv, cx.s, cx.Rt, cx.Ht, cx.dyn, cx.tmpl, cx.ddyn, smem, nullptr, blk / CX_SPLIT,
                                                    (blk % CX_SPLIT) * TS, TS, cx.T, HS, cx.HT, cx.dSg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its write-through stores ...
        __syncthreads();
        if (tid == 0) { gptst_publish_fence(); __hip_atomic_store(cx.flags + blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }     // ... before the flag goes up
        GPTST_STAMP(1);
        if (blk >= nhr) { GPTST_WG_END(); return; }
        // r06: the first nhr role workgroups go on with a node HALF (below) — a role is a 12-us latency chain that leaves the CU's matrix pipe idle, a half
        // is what the CU lacks for an even load; with the halves behind the roles' slots instead, the last of them start when the roles END (measured:
        // profiles/r06_node_halves.txt).  Their own role is done before they wait for anybody's flag.
        ru = nfull + blk;
        __syncthreads();
    } else {
        // r06 NODE HALVES: routing work unit ru < nfull is the whole (b,t) = ru; the units behind are the two node halves (tiles [0, th) / [th, ntiles)) of
        // (b,t) = nfull + (ru - nfull) / 2 — see gptst_cap_split_units().  Everything below is per node tile, so a half needs no exchange with the other
        // one: only its weight / bias gradient partial is a row of its own (row ru of dWp / dbp: B*T + nsplit rows).
        const int r = ROLES ? blk - cx.nB : blk;
        ru = r < nfull ? r : r + nhr;                  // (halves 0 .. nhr-1 ride behind the roles)
    }
    int bt = ru, tlo = 0, thi = NR / 16;
    if (ru >= nfull) {
        const int k = ru - nfull, th = (NR / 16 + 1) >> 1;
        bt = nfull + (k >> 1);
        if (k & 1) tlo = th; else thi = th;
    }
    const float* Xbt = X + (size_t)bt * N * C;
    const bool fold = !ROLES && cx.dv != nullptr;                  // dS out of the cross-time backward computed HERE (uniform)
    if constexpr (!ROLES) {
        if (fold) {
            for (int i = tid; i < HSP * P; i += CM_NT) Vs[i] = 0.f;
            __syncthreads();
            cap_cross_bwd_prologue<C, false>(cx.dv, cx.s, cx.Rt, cx.Ht, cx.dyn, cx.tmpl, cx.ddyn, smem, Vs, bt / cx.T, bt % cx.T, 1, cx.T, HS, cx.HT);
        }
    }
    GPTST_STAMP(1);

    if constexpr (C == 64) {
        // ---- Y = X Wp^T + bp, q = |Y|^2, g = squash factor: fused MFMA epilogue as in the forward ----
        const int j = lane & 15, kk = lane >> 4;
        float4 a[4];
        cm_fetch_a16(a, Xbt, tlo + wave, N, j, kk);
        const float4 b4 = ld4(bp + 4 * j);
        load_w_lds<C, CM_NT>(Wl, Wp, 1, tid);
        if (!fold) for (int i = tid; i < HSP * P; i += CM_NT) Vs[i] = 0.f;
        __syncthreads();
        GPTST_STAMP(2);
        float4 bv[4][4];
        cm_load_bfrag(bv, Wl, j, kk);
        for (int tile = tlo + wave; tile < thi; tile += CM_NW) {
            if (tile != tlo + wave) cm_fetch_a16(a, Xbt, tile, N, j, kk);
            float4 y[4];
            cm_tile16(a, bv, b4, y);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = tile * 16 + kk * 4 + r;
                const float4 v = n < N ? y[r] : f4zero();
                const float q = group_sum<16>(f4dot(v, v));
                st4(Ys + n * P + 4 * j, v);
                if (j == 0) { qq[n] = q; gq[n] = q / ((1.f + q) * (sqrtf(q) + 1e-8f)); }
            }
        }
        __syncthreads();
        GPTST_STAMP(3);
        for (int i = tid; i < 2 * HSP * NP; i += CM_NT) cs[i] = 0.f;
        __syncthreads();
    } else {
    float4 xv[T::F4_PER_LANE];
    cm_fetch_x<C>(xv, Xbt, wave, N, lane);
    load_w_lds<C, CM_NT>(Wl, Wp, 1, tid);
    if (!fold) for (int i = tid; i < HSP * P; i += CM_NT) Vs[i] = 0.f;
    __syncthreads();
    for (int t = wave; t < (NR + 31) / 32; t += CM_NW) {           // same tiling as the forward
        float* tile = Ys + t * 32 * P;
        const int rows_here = min(32, NR - t * 32);
        if (t != wave) cm_fetch_x<C>(xv, Xbt, t, N, lane);
#pragma unroll
        for (int it = 0; it < T::F4_PER_LANE; ++it) {
            const int f = it * 64 + lane;
            const int r = f / T::F4_PER_ROW, c4 = f % T::F4_PER_ROW;
            const int n = t * 32 + r;
            if (r < rows_here) st4(tile + r * P + 4 * c4, n < N ? xv[it] : f4zero());
        }
        f32x16 acc[T::NCT];
        if (rows_here == 32) {
            mfma_tile<C>(tile, Wl, acc, lane);
            acc_to_tile<C>(tile, acc, lane);
        } else {
            const int i = lane & 31, h = lane >> 5;
#pragma unroll
            for (int ct = 0; ct < T::NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
            const float* arow = tile + (i & 15) * P + 4 * h;
            const float* wcol = Wl + 4 * h * C + i;
#pragma unroll 2
            for (int q = 0; q < C / 8; ++q) {
                const float4 a4 = ld4(arow + 8 * q);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int ct = 0; ct < T::NCT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], wcol[(8 * q + jj) * C + ct * 32], acc[ct], 0, 0, 0);
            }
#pragma unroll
            for (int ct = 0; ct < T::NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < 16) tile[row * P + ct * 32 + i] = acc[ct][r];
                }
        }
    }
    __syncthreads();
    {   // Y += bp, per-row q and g;  stage c, dc1, dS
        const int slot = tid / LPR, jj = tid % LPR;
        const float4 b4 = ld4(bp + 4 * jj);
        for (int n0 = 0; n0 < NR; n0 += RPP) {
            const int n = n0 + slot;
            float4 y = f4zero();
            if (n < N) y = f4add(ld4(Ys + n * P + 4 * jj), b4);
            const float q = group_sum<LPR>(f4dot(y, y));
            if (n < NR) {
                st4(Ys + n * P + 4 * jj, y);
                if (jj == 0) { qq[n] = q; gq[n] = q / ((1.f + q) * (sqrtf(q) + 1e-8f)); }
            }
        }
        for (int i = tid; i < 2 * HSP * NP; i += CM_NT) cs[i] = 0.f;
    }
    __syncthreads();
    }
    auto poll = [&]() {                                              // ONE lane polls the flags it depends on, relaxed, bounded
        if (tid == 0) {
            bool got = gptst_wait_ge(cx.flags + (bt / cx.T) * CX_SPLIT + (bt % cx.T) / (12 / CX_SPLIT), 1u, &g_handoff_lost_capmfma);
            *s_ok = got ? 1u : 0u;
        }
    };
    // (r06: requesting c / dc1 ahead of the capsule GEMM — 8 more live registers across it — measured 889.1 vs 892.0 steps/s: not kept)
    for (int i0 = 0; i0 < HS * N; i0 += 4 * CM_NT) {             // c, dc1: batches of 4 + 4 loads per thread
        float cv[4], dv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = min(i0 + k * CM_NT + tid, HS * N - 1);
            cv[k] = c[(size_t)bt * HS * N + i];
            dv[k] = dc1[(size_t)bt * HS * N + i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * CM_NT + tid;
            if (i < HS * N) { cs[(i / N) * NP + i % N] = cv[k]; dcs[(i / N) * NP + i % N] = dv[k]; }
        }
    }
    if (!fold && !ROLES) for (int i = tid; i < HS * LPR; i += CM_NT) st4(Vs + (i / LPR) * P + 4 * (i % LPR), ld4(dS + (size_t)bt * HS * C + 4 * i));
    if constexpr (ROLES != 0) {
        poll();
        __syncthreads();
        const bool ok = *s_ok != 0u;
        typedef int i32x4_ __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cx.dSg) + (size_t)bt * HS * C, 0, HS * C * 4, 0x00020000);
        for (int i = tid; i < HS * LPR; i += CM_NT) {                // dS rows of this (b,t): sc1 loads (written through by the cross-time role)
            const i32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 16);
            const f32x4 f4 = __builtin_bit_cast(f32x4, v);
            const float nan = __int_as_float(0x7fc00000);
            st4(Vs + (i / LPR) * P + 4 * (i % LPR), ok ? make_float4(f4[0], f4[1], f4[2], f4[3]) : make_float4(nan, nan, nan, nan));
        }
    }
    __syncthreads();
    GPTST_STAMP(4);

    const int j = lane & 15, kk = lane >> 4;
    float csum[C / 16];
#pragma unroll
    for (int ct = 0; ct < C / 16; ++ct) csum[ct] = 0.f;
    for (int nt = tlo + wave; nt < thi; nt += CM_NW) {
        const int n = nt * 16 + j;                          // node of this lane as MFMA column (type 2) / row (dP)
        const float gn = gq[n];
        // ---- (a) dc = dc1 + dS.P^T on this node tile, wsum[n] = sum_h c dc ----
        float wsum = 0.f;
        for (int ht = 0; ht < HSP / 16; ++ht) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < C / 16; ++q) {
                const float4 a = ld4(Vs + (ht * 16 + j) * P + 16 * q + 4 * kk);
                const float4 b = ld4(Ys + n * P + 16 * q + 4 * kk);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, gn * b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, gn * b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, gn * b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, gn * b.w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = ht * 16 + kk * 4 + r;           // D reg r: row h, col n
                if (n < N) {                                  // (n >= N would alias the next row of the [h][NP] arrays)
                    const float dch = dcs[h * NP + n] + acc[r];
                    dcs[h * NP + n] = dch;
                    wsum = fmaf(cs[h * NP + n], dch, wsum);   // c is zero for padded h
                }
            }
        }
        wsum += __shfl_xor(wsum, 16, 64);
        wsum += __shfl_xor(wsum, 32, 64);
        if (n < N)
            for (int ht = 0; ht < HSP / 16; ++ht)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int h = ht * 16 + kk * 4 + r;
                    if (h < HS) dlogit[((size_t)bt * HS + h) * N + n] = cs[h * NP + n] * (dcs[h * NP + n] - wsum);
                }
        // ---- (b) dP[n,:] = sum_h c[h,n] dS[h,:];  D[i = node][j = column] ----
        f32x4 dp[C / 16];
#pragma unroll
        for (int ct = 0; ct < C / 16; ++ct) dp[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < HSP / 4; ++s) {
            const float a = n < N ? cs[(4 * s + kk) * NP + n] : 0.f;
#pragma unroll
            for (int ct = 0; ct < C / 16; ++ct)
                dp[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Vs[(4 * s + kk) * P + 16 * ct + j], dp[ct], 0, 0, 0);
        }
        // squash backward in the D layout: reg r <-> node nt*16 + kk*4 + r, column 16ct + j
        float yv[C / 16][4], ydp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = 0.f;
#pragma unroll
            for (int ct = 0; ct < C / 16; ++ct) {
                yv[ct][r] = Ys[(nt * 16 + kk * 4 + r) * P + 16 * ct + j];
                s = fmaf(yv[ct][r], dp[ct][r], s);
            }
            ydp[r] = group_sum<16>(s);                        // the 16 lanes of a DPP row share kk, i.e. the same 4 nodes
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nn = nt * 16 + kk * 4 + r;
            const float q = qq[nn], g = gq[nn];
            const float rt = sqrtf(q), den = (1.f + q) * (rt + 1e-8f);
            float gp = 0.f;
            if (rt > 0.f) gp = (den - q * ((rt + 1e-8f) + (1.f + q) * 0.5f / rt)) / (den * den);
            const float k2 = 2.f * gp * ydp[r];
#pragma unroll
            for (int ct = 0; ct < C / 16; ++ct) {
                const float dy = fmaf(k2, yv[ct][r], g * dp[ct][r]);
                Ys[nn * P + 16 * ct + j] = dy;                                                                    // dY over Y (own tile)
                if constexpr (LIN) csum[ct] += dy;                // column sums of dY (the entry Linear's bias gradient): rows beyond N are zero
            }
        }
        // coalesced rows out (LIN: dY stays in LDS — its only consumer is the Linear backward below)
        if constexpr (!LIN) {
#pragma unroll
            for (int i = 0; i < 16 * LPR / 64; ++i) {
                const int f = i * 64 + lane, nl = f / LPR, c4 = f % LPR;
                const int nn = nt * 16 + nl;
                if (nn < N) st4(dY + ((size_t)bt * N + nn) * C + 4 * c4, ld4(Ys + nn * P + 4 * c4));
            }
        }
    }
    GPTST_STAMP(5);
    if constexpr (LIN && C == 64) {
        // (lane-derived indices re-derived from an OPAQUE copy of the thread index: otherwise the compiler hoists this block's address arithmetic to the
        //  top of the kernel and the capsule GEMM above — 112 live registers of fragments — spills under the 128-register budget)
        int tid_ = threadIdx.x;
        asm volatile("" : "+v"(tid_));
        const int lane_ = tid_ & 63, wave_ = __builtin_amdgcn_readfirstlane(tid_ >> 6), j_ = lane_ & 15, kk_ = lane_ >> 4;
        // ---- (c) dX tile = dY tile . Wp + residual branch.  Same operand scheme as applywg64_kernel<1>: A = dY rows from LDS (lane (j_,kk_): row j_,
        //      channels 16q+4kk..), B = Wp fragments ([out][in] as stored: row 16q+4kk+e, columns 4j..4j+3 <-> column tile = component), accumulator
        //      tile ct / register r = row 4kk+r, channel 4j+ct.  Wp is staged in LDS over the cs / dcs region (dead once every wave_ has left (a)/(b):
        //      one barrier, which also completes dY for (d)); holding the 64 fragment registers next to the residual operands spilled at 128 VGPRs.
        const float* sgn_src = lin.out != nullptr ? lin.out : X;      // the sign operand: the layer's output, or X with premul
        float4 rv[4], sv[4];
#define CRL_LOAD_RES(nt_) _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                   \
            const size_t off_ = ((size_t)bt * N + min((nt_) * 16 + kk_ * 4 + r, N - 1)) * 64 + 4 * j_;                         \
            sv[r] = ld4(sgn_src + off_); rv[r] = ld4(lin.dPre + off_); }
        CRL_LOAD_RES(tlo + wave_);                                // first tile's residual operands: in flight across the barrier and the staging
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { csum[ct] += __shfl_xor(csum[ct], 16, 64); csum[ct] += __shfl_xor(csum[ct], 32, 64); }
        __syncthreads();
        load_w_lds<C, CM_NT>(Wl, Wp, 0, tid_);
        if (kk_ == 0) {                                          // per-wave column sums of dY -> Vs (free since the barrier), [wave][channel 16ct + j]
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) Vs[wave_ * 64 + 16 * ct + j_] = csum[ct];
        }
        __syncthreads();
        if (tid_ < 64) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < CM_NW; ++w) t += Vs[w * 64 + tid_];
            lin.dbp[(size_t)ru * 64 + tid_] = t;
        }
        GPTST_STAMP(6);
        for (int nt = tlo + wave_; nt < thi; nt += CM_NW) {
            if (nt != tlo + wave_) { CRL_LOAD_RES(nt); }
            SB();
            unsigned sgn = 0u;                                   // bit 4r+e: the sign operand of row r / channel 4j+e is positive
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sgn |= ((sv[r].x > 0.f ? 1u : 0u) | (sv[r].y > 0.f ? 2u : 0u) | (sv[r].z > 0.f ? 4u : 0u) | (sv[r].w > 0.f ? 8u : 0u)) << (4 * r);
            float4 ap[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) ap[q] = ld4(Ys + (nt * 16 + j_) * P + 16 * q + 4 * kk_);
            f32x4 acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 bw[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) bw[e] = ld4(Wl + (16 * q + 4 * kk_ + e) * 64 + 4 * j_);
                const float av[4] = {ap[q].x, ap[q].y, ap[q].z, ap[q].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e].x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e].y, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e].z, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e].w, acc[3], 0, 0, 0);
                }
            }
            SB();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = nt * 16 + kk_ * 4 + r;
                float4 o4 = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
                const float g0 = (sgn >> (4 * r)) & 1u ? 1.f : LRELU_SLOPE, g1 = (sgn >> (4 * r)) & 2u ? 1.f : LRELU_SLOPE;
                const float g2 = (sgn >> (4 * r)) & 4u ? 1.f : LRELU_SLOPE, g3 = (sgn >> (4 * r)) & 8u ? 1.f : LRELU_SLOPE;
                if (lin.out != nullptr) {                        // dX = dY Wp + dOut lrelu'(out)
                    o4.x = fmaf(rv[r].x, g0, o4.x); o4.y = fmaf(rv[r].y, g1, o4.y); o4.z = fmaf(rv[r].z, g2, o4.z); o4.w = fmaf(rv[r].w, g3, o4.w);
                } else {
                    o4 = f4add(o4, rv[r]);                       // dX = dY Wp + dPre ...
                    if (lin.premul) { o4.x *= g0; o4.y *= g1; o4.z *= g2; o4.w *= g3; }       // ... times lrelu'(X)
                }
                if (nn < N) st4(lin.dX + ((size_t)bt * N + nn) * 64 + 4 * j_, o4);
            }
        }
#undef CRL_LOAD_RES
        GPTST_STAMP(7);
        // ---- (d) dWp[o][i] = sum_n dY[n][o] X[n][i], dbp[o] = sum_n dY[n][o]: wave_ w owns output tile (o-tile w/2, i-tiles 2(w%2), 2(w%2)+1) for ALL
        //      nodes — no cross-wave_ fold, no further barrier (dY is complete since the barrier in front of (c)).  A = dY^T from LDS (lane (j_ = o, kk_ = node of the k-step)), B = X rows from global memory (L1 / L2: this
        //      workgroup read them at its start).  Padded rows of Ys are zero.  (dbp: column sums gathered in loop (b), folded over the waves above.)
        {
            const int ot = wave_ >> 1, it0 = 2 * (wave_ & 1);
            f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
            const float* xcol = Xbt + 16 * it0 + j_;
            const float* ycol = Ys + 16 * ot + j_;
            constexpr int UB = 11;                               // k-steps per batch of loads; the next batch is requested before this one's MFMAs
            const int sbeg = 4 * tlo, nsteps = 4 * thi;              // k-steps (4 nodes each) of this unit's node tiles
            float b0[UB], b1[UB], n0[UB], n1[UB];
#define CRL_LOADX(d0, d1, s0_) _Pragma("unroll") for (int u = 0; u < UB; ++u) {                          \
                const int n_ = min(4 * ((s0_) + u) + kk_, N - 1);                                          \
                d0[u] = xcol[(size_t)n_ * 64]; d1[u] = xcol[(size_t)n_ * 64 + 16]; }
            CRL_LOADX(b0, b1, sbeg);
#pragma unroll 1
            for (int s0 = sbeg; s0 < nsteps; s0 += UB) {
                if (s0 + UB < nsteps) { CRL_LOADX(n0, n1, s0 + UB); }
                float av[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) av[u] = (s0 + u < nsteps) ? ycol[(4 * (s0 + u) + kk_) * P] : 0.f;
                SB();
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    w0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b0[u], w0, 0, 0, 0);
                    w1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b1[u], w1, 0, 0, 0);
                }
                SB();
#pragma unroll
                for (int u = 0; u < UB; ++u) { b0[u] = n0[u]; b1[u] = n1[u]; }
            }
#undef CRL_LOADX
            float* dw = lin.dWp + (size_t)ru * 64 * 64;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * ot + 4 * kk_ + r;             // D reg r: row o, col i
                dw[o * 64 + 16 * it0 + j_] = w0[r];
                dw[o * 64 + 16 * it0 + 16 + j_] = w1[r];
            }
        }
        GPTST_STAMP(8);
    }
    GPTST_WG_END();
}

thread_local int g_cap_split_roles = 1;     // gptst_tune(26, 0): node halves never ride behind the cross-time roles
thread_local int g_cap_bwd_noroles = 0;       // gptst_tune(23, 1): the cross-time backward as a replicated prologue (r03) also where the role form serves

template <int C>
static int launch_route_bwd2(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1, const float* dS,
                             float* dY, float* dlogit, int BT, int N, int HS, hipStream_t st, CrossBwdArgs cx = CrossBwdArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0},
                             LinArgs lin = LinArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0}, RouteJobs<true>* jobs = nullptr, int nsplit = 0) {
    if (HS > 64) return GPTST_ESHAPE;
    if (nsplit < 0 || nsplit > BT || (nsplit > 0 && (C != 64 || lin.dX == nullptr))) return GPTST_EARG;      // node halves: the folded C = 64 form only
    const int nfull = BT - nsplit;                            // routing work units: nfull whole (b,t) + 2 nsplit node halves
    if (lin.dX != nullptr && C != 64) return GPTST_ESHAPE;      // the folded Linear backward: C = 64
    const int NR = cm_rows(N), NP = cm_np(N), HSP = cm_hsp(HS);
    size_t r2 = (size_t)C * C, need = (size_t)2 * HSP * NP;
    if (need > r2) r2 = need;
    r2 = (r2 + 3) & ~(size_t)3;
    const size_t smem = ((size_t)NR * Tile<C>::PITCH + r2 + (size_t)HSP * Tile<C>::PITCH + 2 * (size_t)NR + 4) * sizeof(float);
    if (smem > 160 * 1024) return GPTST_ESHAPE;
    if (cx.dv) {                                     // the prologue's scratch must fit the Ys + Wl regions
        const size_t need_cx = (size_t)(cx.T * HS + 2 * cx.HT + HS) * Tile<C>::PITCH + (size_t)cx.HT * cx.T * HS;
        if (need_cx > (size_t)NR * Tile<C>::PITCH + r2 || (cx.HT * cx.T * HS) % 4 != 0) return GPTST_ESHAPE;
    }
    if (cx.nB > 0) {                                     // roles: every workgroup resident (two per CU), T = 12, the role's scratch within the LDS tile regions
        const int no = (12 / CX_SPLIT) * HS;
        const size_t need_r = (size_t)(cx.T * HS + 2 * cx.HT + 2 * no) * Tile<C>::PITCH + (size_t)cx.HT * cx.T * HS;
        if (smem > 80 * 1024 || BT + cx.nB > 512 || cx.T != 12 || need_r > (size_t)NR * Tile<C>::PITCH + r2) cx.nB = 0;
    }
    // halves carried by role workgroups (gptst_tune(26, 0): none): the launch then has nB + nfull + (2 nsplit - nhr) workgroups ahead of the jobs
    const int nhr = (cx.nB > 0 && g_cap_split_roles) ? (2 * nsplit < cx.nB ? 2 * nsplit : cx.nB) : 0;
    const int NU = BT + nsplit - nhr;                         // routing workgroups behind the roles
    if (jobs != nullptr && !(cx.nB > 0 && lin.dX != nullptr && smem >= 2 * (4 * PG_MAXK * 65) * sizeof(float))) return GPTST_ESHAPE;   // (the caller then runs the jobs on their own)
    if (jobs != nullptr) {
        static size_t curJ = 0;
        if (smem > curJ) { (void)hipFuncSetAttribute((const void*)cap_route_bwd2_kernel<C, 1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); curJ = smem; }
        jobs->first = NU + cx.nB;
        const int njw = ((jobs->npb + 1) >> 1) + ((jobs->neb + 1) >> 1);
        hipLaunchKernelGGL((cap_route_bwd2_kernel<C, 1, true, true>), dim3(NU + cx.nB + njw), dim3(CM_NT), smem, st, X, Wp, bp, c, dc1, dS, dY, dlogit, N, HS, (int)r2, cx, lin, *jobs, nfull, nhr);
        GPTST_CHECK_LAUNCH();
        return GPTST_OK;
    }
    if (cx.nB > 0 && lin.dX != nullptr) {
        static size_t curRL = 0;
        if (smem > curRL) { (void)hipFuncSetAttribute((const void*)cap_route_bwd2_kernel<C, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); curRL = smem; }
        hipLaunchKernelGGL((cap_route_bwd2_kernel<C, 1, true>), dim3(NU + cx.nB), dim3(CM_NT), smem, st, X, Wp, bp, c, dc1, dS, dY, dlogit, N, HS, (int)r2, cx, lin, RouteJobs<false>{}, nfull, nhr);
        GPTST_CHECK_LAUNCH();
        return GPTST_OK;
    }
    if (cx.nB > 0) {
        static size_t curR = 0;
        if (smem > curR) { (void)hipFuncSetAttribute((const void*)cap_route_bwd2_kernel<C, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); curR = smem; }
        hipLaunchKernelGGL((cap_route_bwd2_kernel<C, 1>), dim3(BT + cx.nB), dim3(CM_NT), smem, st, X, Wp, bp, c, dc1, dS, dY, dlogit, N, HS, (int)r2, cx, lin, RouteJobs<false>{}, BT, 0);
        GPTST_CHECK_LAUNCH();
        return GPTST_OK;
    }
    if (lin.dX != nullptr) {
        static size_t curL = 0;
        if (smem > curL) { (void)hipFuncSetAttribute((const void*)cap_route_bwd2_kernel<C, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); curL = smem; }
        hipLaunchKernelGGL((cap_route_bwd2_kernel<C, 0, true>), dim3(NU), dim3(CM_NT), smem, st, X, Wp, bp, c, dc1, dS, dY, dlogit, N, HS, (int)r2, cx, lin, RouteJobs<false>{}, nfull, nhr);
        GPTST_CHECK_LAUNCH();
        return GPTST_OK;
    }
    static size_t cur = 0;
    if (smem > cur) { (void)hipFuncSetAttribute((const void*)cap_route_bwd2_kernel<C, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); cur = smem; }
    hipLaunchKernelGGL((cap_route_bwd2_kernel<C, 0>), dim3(BT), dim3(CM_NT), smem, st, X, Wp, bp, c, dc1, dS, dY, dlogit, N, HS, (int)r2, cx, lin, RouteJobs<false>{}, BT, 0);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

GPTST_INTERNAL int gptst_cap_route_bwd_v1(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1,
                                      const float* dS, float* dY, float* dlogit, int BT, int N, int C, int HS, void* stream);

extern "C" int gptst_cap_route_bwd(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1,
                                   const float* dS, float* dY, float* dlogit, int BT, int N, int C, int HS, void* stream) {
    if (!X || !Wp || !bp || !c || !dc1 || !dS || !dY || !dlogit) return GPTST_EARG;
    int rc = GPTST_ESHAPE;
    if (C == 64) rc = launch_route_bwd2<64>(X, Wp, bp, c, dc1, dS, dY, dlogit, BT, N, HS, (hipStream_t)stream);
    if (rc == GPTST_ESHAPE) return gptst_cap_route_bwd_v1(X, Wp, bp, c, dc1, dS, dY, dlogit, BT, N, C, HS, stream);
    return rc;
}

// gptst_cap_cross_bwd + gptst_cap_route_bwd in ONE launch (the cross-time backward as a prologue of every (b,t) workgroup; dS never leaves
// LDS).  dv (B, T*HS, C) gradient of the cross-time block's output v; -> dY, dlogit, ddyn (B, HT, T*HS).  C = 64 and T*HS tokens within the
// kernel's LDS scratch, else GPTST_ESHAPE (use the two launches).
// dS_ws (B*T, HS, C) + flags (4 B 32-bit words, ZERO on entry; e.g. a slice of the step's zeroed scratch): both given -> the cross-time backward runs as a
// ROLE of the launch (B extra workgroups, once per sample, overlapped with the routing workgroups' capsule GEMM); NULL -> every (b,t) workgroup repeats it.
extern "C" int gptst_cap_cross_route_bwd(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1, const float* dv,
                                         const float* s, const float* Rt, const float* Ht, const float* dyn, const float* tmpl, float* dY,
                                         float* dlogit, float* ddyn, float* dS_ws, void* flags, int B, int T, int N, int C, int HS, int HT, void* stream) {
    if (!X || !Wp || !bp || !c || !dc1 || !dv || !s || !Rt || !Ht || !dyn || !tmpl || !dY || !dlogit || !ddyn || B <= 0 || T <= 0) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    const bool roles = dS_ws != nullptr && flags != nullptr && !g_cap_bwd_noroles;
    return launch_route_bwd2<64>(X, Wp, bp, c, dc1, nullptr, dY, dlogit, B * T, N, HS, (hipStream_t)stream,
                                 CrossBwdArgs{dv, s, Rt, Ht, dyn, tmpl, ddyn, T, HT, roles ? dS_ws : nullptr, roles ? (unsigned*)flags : nullptr, roles ? B * CX_SPLIT : 0});
}

// gptst_cap_cross_route_bwd + gptst_linear_bwd in ONE launch (r05): the (b,t) workgroup goes on from its dY tile (kept in LDS) to the backward of the
// cap's entry Linear and the layer's residual branch.  dPre (B*T*N, C): the cap layer's output gradient — already dPre when out == NULL (then premul
// multiplies dX by lrelu'(X)), else dOut with out (B*T*N, C) the layer's output.  -> dX (B*T*N, C), dWp (B*T, C*C) / dbp (B*T, C): ONE partial per
// (b,t) (the caller's reduction sums B*T rows), dlogit, ddyn.  No dY tensor exists.  dS_ws / flags as gptst_cap_cross_route_bwd.  C = 64, else GPTST_ESHAPE.
extern "C" int gptst_cap_cross_route_lin_bwd(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1, const float* dv,
                                             const float* s, const float* Rt, const float* Ht, const float* dyn, const float* tmpl,
                                             const float* dPre, const float* out, int premul, float* dX, float* dWp, float* dbp, float* dlogit,
                                             float* ddyn, float* dS_ws, void* flags, int B, int T, int N, int C, int HS, int HT, void* stream) {
    if (!X || !Wp || !bp || !c || !dc1 || !dv || !s || !Rt || !Ht || !dyn || !tmpl || !dPre || !dX || !dWp || !dbp || !dlogit || !ddyn || B <= 0 ||
        T <= 0 || (out && premul)) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    const bool roles = dS_ws != nullptr && flags != nullptr && !g_cap_bwd_noroles;
    return launch_route_bwd2<64>(X, Wp, bp, c, dc1, nullptr, nullptr, dlogit, B * T, N, HS, (hipStream_t)stream,
                                 CrossBwdArgs{dv, s, Rt, Ht, dyn, tmpl, ddyn, T, HT, roles ? dS_ws : nullptr, roles ? (unsigned*)flags : nullptr, roles ? B * CX_SPLIT : 0},
                                 LinArgs{dPre, out, dX, dWp, dbp, premul});
}

// =====================================================================================================================
// backward of the cluster -> node scatter rec[n,:] = sum_h c[h,n] v[h,:]  (GPTST.py:135):
//   dc1[h,n] = drec[n,:] . v[h,:]   (type 2, "V" = v)        dv[h,:] = sum_n c[h,n] drec[n,:]   (type 1, "P" = drec)
// drec rows of one (b,t) are staged in LDS once; both contractions on MFMA 16x16x4.
// =====================================================================================================================
// (cap_rec_bwd2_kernel: defined above cap_route_bwd2_kernel, which runs the same body as its first ROLE)
GPTST_INTERNAL int gptst_cap_rec_bwd_v1(const float* drec, const float* c, const float* v, float* dc1, float* dv, int BT, int N, int C,
                                    int HS, void* stream);

// r05, late: gptst_cap_cross_route_lin_bwd + gptst_pool_jobs(njobs gradient-reduction jobs, kinds 1 / 2) — the jobs as role workgroups of the same launch where
// the role form serves (see RouteJobs above), else as those two calls.  The jobs' inputs must be complete when this call is enqueued (earlier launches of the
// stream), and nothing in this launch may read their outputs.
GPTST_INTERNAL int gptst_pj_reduce_table(PJobs* t, int njobs, const int* kind, const void* const* emb, const void* const* x, const void* const* pool,
                                         const void* const* out, const int* R, const int* K, const int* cols, const int* nsplit, const int* ldx,
                                         int* npool, int* npb, int* neb);
extern "C" int gptst_pool_jobs(int njobs, const int* kind, const void* const* emb, const void* const* x, const void* const* pool,
                               const void* const* out, const int* R, const int* K, const int* cols, const int* nsplit, const int* ldx, void* stream);
// 1: gptst_cap_cross_route_lin_bwd at this shape runs the cross-time backward as a ROLE (dS_ws / flags given) — the form that also carries reduction jobs
// (gptst_cap_cross_route_lin_bwd_jobs); 0: a caller with jobs to place keeps them for its reduction launch (the _jobs entry would run them as a launch of their own).
extern "C" int gptst_cap_route_roles_ok(int B, int T, int N, int C, int HS, int HT) {
    if (C != 64 || HS > 64 || HS <= 0 || B <= 0 || N <= 0 || T != 12 || g_cap_bwd_noroles) return 0;
    const int NR = cm_rows(N), NP = cm_np(N), HSP = cm_hsp(HS), BT = B * T, nB = B * CX_SPLIT;
    size_t r2 = (size_t)C * C, need = (size_t)2 * HSP * NP;
    if (need > r2) r2 = need;
    r2 = (r2 + 3) & ~(size_t)3;
    const size_t smem = ((size_t)NR * Tile<64>::PITCH + r2 + (size_t)HSP * Tile<64>::PITCH + 2 * (size_t)NR + 4) * sizeof(float);
    const int no = (12 / CX_SPLIT) * HS;
    const size_t need_r = (size_t)(T * HS + 2 * HT + 2 * no) * Tile<64>::PITCH + (size_t)HT * T * HS;
    if (smem > 80 * 1024 || BT + nB > 512 || need_r > (size_t)NR * Tile<64>::PITCH + r2 || (HT * T * HS) % 4 != 0) return 0;
    return smem >= 2 * (4 * PG_MAXK * 65) * sizeof(float) ? 1 : 0;
}

// r06 NODE HALVES (MEASURED AND NOT ADOPTED: off by default, profiles/r06_node_halves.txt).  B*T = 384 routing workgroups on 256 CUs at
// two per CU: half the CUs run two (b,t), the other half one.  Everything in the routing backward is per node tile, so the LAST nsplit (b,t) can be cut into
// two node halves of their own workgroup each: B*T - nsplit whole units + 2 nsplit halves — at B*T = 1.5 CUs every CU gets one whole unit and one half (the
// first 4 B halves ride behind the cross-time role workgroups, so that all 512 workgroups are resident from the start).  The price: a half writes its own
// dWp / dbp partial row (B*T + nsplit rows for the reduction).  What the stamps say: a HALF lasts 33-35 us next to a whole unit's 35-40 (a whole unit next to
// another: 41) — the workgroup is a chain of dependent phases whose length hardly depends on its tile count, the launch stays at 43 us, and the carried
// reduction jobs lose the slots the 12-us roles used to free (857 vs 887 steps/s).  The 384-on-256 quantisation is not what bounds this launch.
thread_local int g_cap_split = 0;             // gptst_tune(25, v): 0 = off (default), -1 = by the CU count (below), v > 0 = that many (b,t) in halves
static int cap_cu_count() {
    static thread_local int dev_cached = -1, ncu = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev != dev_cached) {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 0;
        dev_cached = dev;
    }
    return ncu;
}
// number of (b,t) units — the last ones — that the C = 64 routing kernels of this device cut into node halves (0: none)
extern "C" int gptst_cap_split_units(int BT, int N, int C, int HS) {
    if (C != 64 || HS <= 0 || HS > 64 || N < 32 || BT <= 0) return 0;
    if (g_cap_split >= 0) return g_cap_split < BT ? g_cap_split : BT;
    const int ncu = cap_cu_count();
    if (ncu <= 0 || BT <= ncu || 2 * BT > 3 * ncu) return 0;      // one unit per CU already / more than 1.5: the extra halves would only queue
    return BT - ncu;
}

static int cap_cross_route_lin_bwd_any(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1, const float* dv,
                                       const float* s, const float* Rt, const float* Ht, const float* dyn, const float* tmpl,
                                       const float* dPre, const float* out, int premul, float* dX, float* dWp, float* dbp, float* dlogit,
                                       float* ddyn, float* dS_ws, void* flags, int B, int T, int N, int C, int HS, int HT, int nsplit,
                                       int njobs, const int* jkind, const void* const* jemb, const void* const* jx, const void* const* jpool,
                                       const void* const* jout, const int* jR, const int* jK, const int* jcols, const int* jnsplit, const int* jldx,
                                       void* stream) {
    if (!X || !Wp || !bp || !c || !dc1 || !dv || !s || !Rt || !Ht || !dyn || !tmpl || !dPre || !dX || !dWp || !dbp || !dlogit || !ddyn || B <= 0 ||
        T <= 0 || (out && premul) || nsplit < 0 || nsplit > B * T) return GPTST_EARG;
    if (njobs < 0 || (njobs && (!jkind || !jemb || !jx || !jpool || !jout || !jR || !jK || !jcols || !jnsplit))) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    if (nsplit > 0 && N < 32) return GPTST_EARG;               // a half needs a node tile of its own
    const bool roles = dS_ws != nullptr && flags != nullptr && !g_cap_bwd_noroles;
    const CrossBwdArgs cx{dv, s, Rt, Ht, dyn, tmpl, ddyn, T, HT, roles ? dS_ws : nullptr, roles ? (unsigned*)flags : nullptr, roles ? B * CX_SPLIT : 0};
    const LinArgs lin{dPre, out, dX, dWp, dbp, premul};
    if (njobs > 0 && roles) {
        RouteJobs<true> jr;
        if (gptst_pj_reduce_table(&jr.t, njobs, jkind, jemb, jx, jpool, jout, jR, jK, jcols, jnsplit, jldx, &jr.npool, &jr.npb, &jr.neb) == GPTST_OK) {
            const int rc = launch_route_bwd2<64>(X, Wp, bp, c, dc1, nullptr, nullptr, dlogit, B * T, N, HS, (hipStream_t)stream, cx, lin, &jr, nsplit);
            if (rc != GPTST_ESHAPE) return rc;
        }
    }
    const int rc = launch_route_bwd2<64>(X, Wp, bp, c, dc1, nullptr, nullptr, dlogit, B * T, N, HS, (hipStream_t)stream, cx, lin, nullptr, nsplit);
    if (rc || njobs == 0) return rc;
    return gptst_pool_jobs(njobs, jkind, jemb, jx, jpool, jout, jR, jK, jcols, jnsplit, jldx, stream);
}

extern "C" int gptst_cap_cross_route_lin_bwd_jobs(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1, const float* dv,
                                                  const float* s, const float* Rt, const float* Ht, const float* dyn, const float* tmpl,
                                                  const float* dPre, const float* out, int premul, float* dX, float* dWp, float* dbp, float* dlogit,
                                                  float* ddyn, float* dS_ws, void* flags, int B, int T, int N, int C, int HS, int HT,
                                                  int njobs, const int* jkind, const void* const* jemb, const void* const* jx, const void* const* jpool,
                                                  const void* const* jout, const int* jR, const int* jK, const int* jcols, const int* jnsplit, const int* jldx,
                                                  void* stream) {
    return cap_cross_route_lin_bwd_any(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl, dPre, out, premul, dX, dWp, dbp, dlogit, ddyn, dS_ws, flags, B, T, N, C, HS,
                                       HT, 0, njobs, jkind, jemb, jx, jpool, jout, jR, jK, jcols, jnsplit, jldx, stream);
}

// ... with the last nsplit (b,t) as node halves (gptst_cap_split_units): dWp (B*T + nsplit, C*C), dbp (B*T + nsplit, C) partial rows
extern "C" int gptst_cap_cross_route_lin_bwd_split(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1, const float* dv,
                                                   const float* s, const float* Rt, const float* Ht, const float* dyn, const float* tmpl,
                                                   const float* dPre, const float* out, int premul, float* dX, float* dWp, float* dbp, float* dlogit,
                                                   float* ddyn, float* dS_ws, void* flags, int B, int T, int N, int C, int HS, int HT, int nsplit,
                                                   int njobs, const int* jkind, const void* const* jemb, const void* const* jx, const void* const* jpool,
                                                   const void* const* jout, const int* jR, const int* jK, const int* jcols, const int* jnsplit, const int* jldx,
                                                   void* stream) {
    return cap_cross_route_lin_bwd_any(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl, dPre, out, premul, dX, dWp, dbp, dlogit, ddyn, dS_ws, flags, B, T, N, C, HS,
                                       HT, nsplit, njobs, jkind, jemb, jx, jpool, jout, jR, jK, jcols, jnsplit, jldx, stream);
}

extern "C" int gptst_cap_rec_bwd(const float* drec, const float* c, const float* v, float* dc1, float* dv, int BT, int N, int C, int HS,
                                 void* stream) {
    if (!drec || !c || !v || !dc1 || !dv) return GPTST_EARG;
    if (C == 64 && HS <= 64) {
        const int NR = cm_rows(N), NP = cm_np(N), HSP = cm_hsp(HS);
        const size_t smem = ((size_t)NR * Tile<64>::PITCH + (size_t)(HSP + HS) * NP + (size_t)HSP * Tile<64>::PITCH + (size_t)HSP * 64) * sizeof(float);
        if (smem <= 160 * 1024) {
            static size_t cur = 0;
            if (smem > cur) { hipFuncSetAttribute((const void*)cap_rec_bwd2_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); cur = smem; }
            hipLaunchKernelGGL((cap_rec_bwd2_kernel<64>), dim3(BT), dim3(CM_NT), smem, (hipStream_t)stream, drec, c, v, dc1, dv, N, HS);
            GPTST_CHECK_LAUNCH();
            return GPTST_OK;
        }
    }
    return gptst_cap_rec_bwd_v1(drec, c, v, dc1, dv, BT, N, C, HS, stream);
}
