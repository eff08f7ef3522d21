// hyperTem forward, fused (reference GPTST.py:154-163):   out = LReLU( (G_n X) W_bt + b_bt + X )
//   one workgroup = (sample b, 16 consecutive nodes): the 12 x 16 x C slab of X is loaded ONCE into LDS; each of the 4 waves takes
//   time steps t = w, w+4, w+8:  (1) R_t[n,:] = sum_u G_n[t,u] X_u[n,:]  on the VALU from LDS (the per-node temporal hypergraph,
//   no nonlinearity between gather and scatter => one 12x12 matrix per node), written to HBM for the weight gradient;
//   (2) R_t (16 x C) @ W_bt (C x C) on fp32 MFMA 16x16x4 — the time-conditioned weight is read straight from L2 as the B operand
//   (it is shared by the 11 node tiles of the sample, so no LDS copy and no barrier); (3) bias + residual (from the LDS slab) +
//   LeakyReLU, stored as whole rows.  Replaces tmix_kernel + apply_kernel<TIME> (11 + 20 us -> one launch) and the R round trip.
#include "mfma_tile.h"
#include "wgrad64.h"
GPTST_STAMP_TABLES(hypertem)
GPTST_HANDOFF_COUNTER(hypertem)      // -DGPTST_STAMPS: per-phase / per-workgroup wall-clock stamps of the pair launch (tools/phase_stamps.py ht_bwd_pair)

#define HT_T 12
#ifndef HT_OCC
#define HT_OCC 2        // waves per SIMD the slab kernels leave room for (2 <-> 256 VGPRs; r05 experiment: 3 / 4, tools/kernel_regs.py)
#endif
#ifdef GPTST_DEBUG
__device__ long long g_ht_ts[64];
static thread_local int g_ht_dbg = 0, g_ht_nt_override = 0;
extern "C" int gptst_ht_dbg(int v) { g_ht_dbg = v; return 0; }
extern "C" int gptst_ht_variant(int fwd, int bwd) { g_ht_nt_override = fwd; (void)bwd; return 0; }
extern "C" int gptst_ht_ts(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ht_ts), sizeof(long long) * 64); }
#define HT_DBG(d) (d)
#else
#define HT_DBG(d) 0
static constexpr int g_ht_dbg = 0, g_ht_nt_override = 0;
#endif

// XCD-aware work map: workgroup L runs on XCD L % 8 (observed dispatch order), and all node tiles of one sample should share an
// XCD so that the sample's twelve W_bt matrices (192 KB) are fetched into ONE L2 instead of eight (PMC: 67 MB -> expected ~23 MB
// of fabric reads per launch).  L -> xcd = L % 8, slot = L / 8;  sample = xcd + 8 * (slot / ntiles), tile = slot % ntiles.
__device__ __forceinline__ bool ht_work_at(int L, int ntiles, int B, int& b, int& tile) {
    const int xcd = L & 7, slot = L >> 3;
    b = xcd + 8 * (slot / ntiles);
    tile = slot % ntiles;
    return b < B;
}
__device__ __forceinline__ bool ht_work(int ntiles, int B, int& b, int& tile) { return ht_work_at(blockIdx.x, ntiles, B, b, tile); }

// NT (rows of the 16-row MFMA tile that are real nodes) is a run-time parameter for experiments: at (B, N) = (32, 170) the time
// is flat for NT = 11..16 (352..512 workgroups) and 35 % worse for NT <= 10 — the MFMA / fragment work per tile does not shrink.
static int ht_pick_nt(int B, int N) {
    if (g_ht_nt_override >= 4 && g_ht_nt_override <= 16) return g_ht_nt_override;
    (void)B; (void)N;
    return 16;     // measured flat for NT = 11..16 at (32, 170): the per-tile MFMA / fragment cost does not shrink with NT
}

__global__ __launch_bounds__(256, HT_OCC) void hypertem_fwd_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                              const float* __restrict__ Wbt, const float* __restrict__ bbt,
                                                              float* __restrict__ R_out, float* __restrict__ out, int N, int B, int NT,
                                                              int dbg) {
    constexpr int C = 64, P = C + 4, GP = 145;
    int tsi = 0;
#ifdef GPTST_DEBUG
#define TS() do { if ((dbg & 1) && blockIdx.x == 59 && threadIdx.x == 0) g_ht_ts[tsi] = clock64(); ++tsi; } while (0)
#else
// Not a no-op: the phase boundaries must stay where they are written.  Without a fence here the machine scheduler interleaves
// the phases (next step's W_bt loads behind this step's output stores, mix FMAs into the MFMA chain): measured 35.6 vs 25.2 us per
// launch (r02j A/B on one box: the -DGPTST_DEBUG build, whose stamps fenced the phases by accident, was 5 % faster end to end).
#define TS() __builtin_amdgcn_sched_barrier(0)
#endif
    TS();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                               // [12][NT][P]
    float* Gs = Xs + HT_T * NT * P;                 // [NT][GP]  (pitch 145: the 16 rows of a mix step hit 16 different banks)
    int b, tile;
    if (!ht_work((N + NT - 1) / NT, B, b, tile)) return;
    const int n0 = tile * NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // slab + graph staging: ALL global loads are issued before the first LDS store (a plain `for (i = tid; ...) lds[i] = glb[i]`
        // loop compiles to load -> s_waitcnt vmcnt(0) -> ds_write per trip: 21 serialised L2 round trips, 1/3 of the kernel)
        const int nl = tid >> 4, c4 = tid & 15;     // thread = (row, float4 column) of every time slice
        const int n = min(n0 + nl, N - 1);
        float4 v[HT_T];
        float gv[9];
#pragma unroll
        for (int t = 0; t < HT_T; ++t) v[t] = ld4(X + (((size_t)b * HT_T + t) * N + n) * C + 4 * c4);
#pragma unroll
        for (int k = 0; k < 9; ++k) gv[k] = G[min(n0 * 144 + tid + k * 256, N * 144 - 1)];
        if (nl < NT) {
#pragma unroll
            for (int t = 0; t < HT_T; ++t) st4(Xs + (t * NT + nl) * P + 4 * c4, n0 + nl < N ? v[t] : f4zero());
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int i = tid + k * 256;
            if (i < NT * 144) Gs[(i / 144) * GP + i % 144] = (n0 + i / 144 < N) ? gv[k] : 0.f;
        }
    }
    __syncthreads();
    TS();
    const int j = lane & 15, kk = lane >> 4;                 // MFMA mapping: A row / D column j, k-slice kk
    const int ja = j < NT ? j : 0;                           // A-operand row (rows >= NT of the 16-row MFMA tile are don't-care)
    // B fragments as float4: MFMA column tile ct, column j  <->  output channel 4j + ct, so one 16-byte load per (k) feeds the
    // four column tiles (the dword version spent ~1000 TA cycles per time step on 64 loads) and the accumulators of a lane are
    // four CONSECUTIVE channels of a row: the epilogue runs from registers.
    // The fragments of the NEXT time step are requested right after the MFMAs of the current one, BEFORE its output stores:
    // vmcnt retires in order on gfx9, so a load issued behind stores waits for their write acknowledgements.
    float4 bv[C / 16][4];
    float4 b4;
#define HT_LOAD_W(tt) do {                                                                                         \
        const float* W_ = Wbt + ((size_t)b * HT_T + (tt)) * C * C;                                                 \
        _Pragma("unroll") for (int q = 0; q < C / 16; ++q)                                                         \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) bv[q][e] = ld4(W_ + (size_t)(16 * q + 4 * kk + e) * C + 4 * j); \
        b4 = ld4(bbt + ((size_t)b * HT_T + (tt)) * C + 4 * j);                                                     \
    } while (0)
    HT_LOAD_W(wave);
    for (int t = wave; t < HT_T; t += 4) {
        const size_t g = (size_t)b * HT_T + t;
        TS();
        // ---- (1) temporal mix, computed directly in the MFMA A-operand layout: lane (j, kk) owns R_t[row j][16q + 4kk .. +3] ----
        float4 a4[C / 16];
#pragma unroll
        for (int q = 0; q < C / 16; ++q) a4[q] = f4zero();
        {
            const float* gr = Gs + ja * GP + t * HT_T;
            const float* xr = Xs + ja * P + 4 * kk;
#pragma unroll
            for (int u = 0; u < HT_T; ++u) {
                const float gu = gr[u];
#pragma unroll
                for (int q = 0; q < C / 16; ++q) a4[q] = f4fma(gu, ld4(xr + u * NT * P + 16 * q), a4[q]);
            }
        }
        TS();
        // ---- (2) R_t @ W_bt on MFMA 16x16x4: A = R_t (registers), B = W_bt fragments (registers) ----
        f32x4 acc[C / 16];
#pragma unroll
        for (int ct = 0; ct < C / 16; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < C / 16; ++q) {
            const float av[4] = {a4[q].x, a4[q].y, a4[q].z, a4[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].w, acc[3], 0, 0, 0);
            }
        }
        TS();
        const float4 bias = b4;
        if (t + 4 < HT_T) HT_LOAD_W(t + 4);
        TS();
        // R_t (kept for the weight gradient) is stored only now, together with the output rows: every store of a time step is
        // issued BEHIND the next step's W_bt loads (vmcnt retires in order)
        if (R_out != nullptr && j < NT && n0 + j < N) {        // (NULL: the backward rebuilds R from X, see wgrad64_mix_body)
#pragma unroll
            for (int q = 0; q < C / 16; ++q) st4_far(R_out + (g * N + n0 + j) * C + 16 * q + 4 * kk, a4[q]);
        }
        // ---- (3) epilogue from registers: lane (j, kk) owns rows kk*4 + r, channels 4j .. 4j+3 ----
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nl = kk * 4 + r;
            if (nl < NT && n0 + nl < N) {
                float4 y = f4add(f4add(make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), bias), ld4(Xs + (t * NT + nl) * P + 4 * j));
                y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
                st4(out + (g * N + n0 + nl) * C + 4 * j, y);
            }
        }
        TS();
    }
}

static size_t ht_smem(int NT) { return ((size_t)HT_T * NT * 68 + NT * 145) * sizeof(float); }

extern "C" int gptst_hypertem_fwd(const float* X, const float* G, const float* Wbt, const float* bbt, float* R_out, float* out, int B,
                                  int T, int N, int C, void* stream) {
    if (!X || !G || !Wbt || !bbt || !out || T != HT_T) return GPTST_EARG;       // R_out may be NULL
    if (C != 64) return GPTST_ESHAPE;
    static int done = 0;
    if (!done) { hipFuncSetAttribute((const void*)hypertem_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ht_smem(16)); done = 1; }
    const int NT = ht_pick_nt(B, N);
    hipLaunchKernelGGL(hypertem_fwd_kernel, dim3(8 * ((B + 7) / 8) * ((N + NT - 1) / NT)), dim3(256), ht_smem(NT), (hipStream_t)stream, X, G, Wbt, bbt, R_out, out, N, B, NT, g_ht_dbg);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// =====================================================================================================================
// hyperTem backward, fused (everything except the weight gradient, which is grouped by (b,t): wgrad role below):
//   dPre = dOut * lrelu'(out)   [Y given]   or   dPre = dOut  [Y == NULL: the producer already multiplied, see PREMUL]
//   dbias[tile][b,t,:] = sum_{n in tile} dPre               (column sums of the slab: one PARTIAL per node tile, plain stores)
//   dR_t = dPre_t W_bt^T                                    MFMA 16x16x4 as  dR_t^T = W_bt dPre_t^T  so that W_bt (L2) is the
//                                                           A operand with coalesced float4 rows; result overwrites the slab
//   dX_u = dPre_u + sum_t G_n[t,u] dR_t                     VALU from the slab (dPre kept in registers)
//   PREMUL: dX_u *= lrelu'(X_u)                             X is the OUTPUT of the layer below (a LeakyReLU), so what leaves this kernel is
//                                                           already that layer's dPre: no backward kernel has to read its own output
//                                                           only for the sign (one 16.7 MB read per layer and role, r03).  The signs
//                                                           are taken from X in the first load batch and kept as 48 bits per thread.
//   dG[b][n][t,u] = sum_c dR_t[n,c] X_u[n,c]                MFMA 16x16x4 per node: one PARTIAL per sample (gram_bwd sums them)
// No atomics: every output element is written by exactly one lane, so the result does not depend on scheduling.
// Replaces apply_kernel<TIME, dPre> + tmix_kernel<bwd> + tmix_dgraph_kernel (19 + 14 + 11 us, and the dR round trip).
// =====================================================================================================================
// Scheduling notes (measured): all global loads of a phase are issued as one batch into registers (a copy loop
// compiles to one L2 round trip per trip); dPre stays in registers for the dX phase instead of being re-read; W_bt fragments of
// the next time step and the X operands of the dG phase are requested before the stores / atomics of the current phase.
// PAIR (r04, hypertem_bwd_pair_kernel): two consecutive layers on the slab.  1 = the upper layer: its input gradient (already multiplied by
// lrelu'(input) = the lower layer's dPre) stays in the thread's dp[] registers for stage 2 AND goes to dX with write-through stores (the lower
// layer's weight-gradient role reads it in the same launch); 2 = the lower layer: dPre comes in dp[], nothing is read from dOut.
template <bool HASY, bool PREMUL, int PAIR>
__device__ __forceinline__ void hypertem_bwd_stage(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                   const float* __restrict__ X, const float* __restrict__ G,
                                                   const float* __restrict__ Wbt, float* __restrict__ dX,
                                                   float* __restrict__ dbias, float* __restrict__ dG, int N, int B, int dbg, int b, int tile,
                                                   float* __restrict__ smem, float4 (&dp)[HT_T]) {
    constexpr int C = 64, P = C + 4, GP = 145, NT = 16;
    float* Ds = smem;                               // [12][16][P]  dPre, then dR
    float* Gs = Ds + HT_T * NT * P;                 // [16][GP]
    const int n0 = tile * NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = tid >> 4, c4 = tid & 15;         // thread = (row, float4 column) of every time slice
    const bool valid = n0 + nl < N;
    const size_t rowoff = ((size_t)b * HT_T * N + min(n0 + nl, N - 1)) * C + 4 * c4;     // + t * N * C
    const int j = lane & 15, kk = lane >> 4;
    // dR_t^T (C x 16) = W_bt (C x C) dPre_t^T:  A[i][kk=o] = W[i][o] (global float4 rows), B[kk=o][j=n] = dPre_t[n][o] (LDS)
    float4 aq[C / 16][C / 16];
#define HT_LOAD_WT(tt) do {                                                                                         \
        const float* W_ = Wbt + ((size_t)b * HT_T + (tt)) * C * C;                                                  \
        _Pragma("unroll") for (int it = 0; it < C / 16; ++it)                                                       \
            _Pragma("unroll") for (int q = 0; q < C / 16; ++q) aq[it][q] = ld4(W_ + (size_t)(it * 16 + j) * C + 16 * q + 4 * kk); \
    } while (0)
    // All global loads of the slab are issued up front IN THE ORDER THEY ARE CONSUMED (vmcnt retires in order): time steps 0-3, the
    // node graphs, this wave's first W_bt, then time steps 4-11.  The slab is staged in three groups of four time steps with a barrier
    // each, and wave w runs its time step 4*group + w right after its group has landed — the dR phase of the first groups overlaps
    // with the arrival of the later ones instead of waiting for the whole 33 MB burst (r02: loads 12 us + time-step loop 12 us were
    // strictly serial; measured gain 30.0 -> 29.3 us: the kernel is bound by the per-workgroup dependency chain, not by this overlap).
    // yv: the layer's output (HASY: sign of dPre) or its input X (PREMUL: sign of the outgoing gradient) — never both.
    static_assert(!(HASY && PREMUL), "one sign operand per launch");
    const float* __restrict__ Sg = HASY ? Y : X;
    float4 yv[HT_T];
    float gv[9];
    unsigned sb0 = 0u, sb1 = 0u;                    // PREMUL: bit 4*t + e of (sb0 | sb1 << 32) = (X_t[row][4*c4 + e] > 0)
#pragma unroll
    for (int t = 0; t < 4; ++t) { if (PAIR != 2) dp[t] = ld4(dOut + rowoff + (size_t)t * N * C); if (HASY || PREMUL) yv[t] = ld4(Sg + rowoff + (size_t)t * N * C); }
#pragma unroll
    for (int k = 0; k < 9; ++k) gv[k] = G[min(n0 * 144 + tid + k * 256, N * 144 - 1)];
    HT_LOAD_WT(wave);
#pragma unroll
    for (int t = 4; t < HT_T; ++t) { if (PAIR != 2) dp[t] = ld4(dOut + rowoff + (size_t)t * N * C); if (HASY || PREMUL) yv[t] = ld4(Sg + rowoff + (size_t)t * N * C); }
    SB();
#pragma unroll
    for (int grp = 0; grp < HT_T / 4; ++grp) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int t = 4 * grp + tt;
            float4 v = dp[t];
            if (HASY) v = make_float4(dp[t].x * lrelu_grad_from_out(yv[t].x), dp[t].y * lrelu_grad_from_out(yv[t].y),
                                      dp[t].z * lrelu_grad_from_out(yv[t].z), dp[t].w * lrelu_grad_from_out(yv[t].w));
            if (PREMUL) {
                const unsigned m4 = (yv[t].x > 0.f ? 1u : 0u) | (yv[t].y > 0.f ? 2u : 0u) | (yv[t].z > 0.f ? 4u : 0u) | (yv[t].w > 0.f ? 8u : 0u);
                if (t < 8) sb0 |= m4 << (4 * t); else sb1 |= m4 << (4 * (t - 8));
            }
            if (!valid) v = f4zero();
            dp[t] = v;
            st4(Ds + (t * NT + nl) * P + 4 * c4, v);
        }
        if (grp == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int i = tid + k * 256;
                Gs[(i / 144) * GP + i % 144] = (n0 + i / 144 < N) ? gv[k] : 0.f;
            }
        }
        __syncthreads();
        SB();
        if (PAIR != 0 && grp == 0) GPTST_STAMP(PAIR == 2 ? 6 : 1);
        const int t = 4 * grp + wave;
        if (HT_DBG(dbg) & 32) continue;
        const size_t g = (size_t)b * HT_T + t;
        float* dt = Ds + t * NT * P;
        // bias gradient: column sums of dPre_t over the 16 nodes (lane = channel)
        float s = 0.f;
        if (dbias != nullptr) {
#pragma unroll
            for (int r = 0; r < NT; ++r) s += dt[r * P + lane];
        }
        float4 bq[C / 16];
#pragma unroll
        for (int q = 0; q < C / 16; ++q) bq[q] = ld4(dt + j * P + 16 * q + 4 * kk);
        f32x4 acc[C / 16];
#pragma unroll
        for (int it = 0; it < C / 16; ++it) acc[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the four output tiles are interleaved so that consecutive MFMAs never depend on each other
        if (!(HT_DBG(dbg) & 64))
#pragma unroll
        for (int q = 0; q < C / 16; ++q) {
#pragma unroll
            for (int it = 0; it < C / 16; ++it) acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[it][q].x, bq[q].x, acc[it], 0, 0, 0);
#pragma unroll
            for (int it = 0; it < C / 16; ++it) acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[it][q].y, bq[q].y, acc[it], 0, 0, 0);
#pragma unroll
            for (int it = 0; it < C / 16; ++it) acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[it][q].z, bq[q].z, acc[it], 0, 0, 0);
#pragma unroll
            for (int it = 0; it < C / 16; ++it) acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[it][q].w, bq[q].w, acc[it], 0, 0, 0);
        }
        SB();
        if (t + 4 < HT_T && !(HT_DBG(dbg) & 128)) HT_LOAD_WT(t + 4);
        SB();
        if (dbias != nullptr && !(HT_DBG(dbg) & 4)) dbias[((size_t)tile * B * HT_T + g) * C + lane] = s;   // partial of this node tile
        // D reg r: row i = it*16 + kk*4 + r (input channel), col j = node  ->  dR_t[node][channel] over the slab
#pragma unroll
        for (int it = 0; it < C / 16; ++it)
            st4(dt + j * P + it * 16 + kk * 4, make_float4(acc[it][0], acc[it][1], acc[it][2], acc[it][3]));
    }
    __syncthreads();
    if (PAIR != 0) GPTST_STAMP(PAIR == 2 ? 7 : 2);
    // X operands of the dG phase (nodes wave, wave+4, ...): requested now, consumed after the dX phase
    float4 xg[NT / 4][C / 16];
#pragma unroll
    for (int i = 0; i < NT / 4; ++i) {
        const int n = min(n0 + wave + 4 * i, N - 1);
#pragma unroll
        for (int q = 0; q < C / 16; ++q)
            xg[i][q] = j < HT_T ? ld4(X + (((size_t)b * HT_T + j) * N + n) * C + 16 * q + 4 * kk) : f4zero();
    }
    SB();
    // ---- dX_u[n,:] = dPre_u[n,:] + sum_t G_n[t,u] dR_t[n,:]   (thread = (row nl, float4 column c4), dPre still in registers) ----
    if (valid && !(HT_DBG(dbg) & 16)) {
        float4 dr[HT_T];
#pragma unroll
        for (int t = 0; t < HT_T; ++t) dr[t] = ld4(Ds + (t * NT + nl) * P + 4 * c4);
        const float* gr = Gs + nl * GP;
#pragma unroll
        for (int u = 0; u < HT_T; ++u) {
            float4 acc = dp[u];
#pragma unroll
            for (int t = 0; t < HT_T; ++t) acc = f4fma(gr[t * HT_T + u], dr[t], acc);
            if (PREMUL) {
                const unsigned m4 = (u < 8 ? sb0 >> (4 * u) : sb1 >> (4 * (u - 8))) & 15u;
                acc.x *= (m4 & 1u) ? 1.f : LRELU_SLOPE; acc.y *= (m4 & 2u) ? 1.f : LRELU_SLOPE;
                acc.z *= (m4 & 4u) ? 1.f : LRELU_SLOPE; acc.w *= (m4 & 8u) ? 1.f : LRELU_SLOPE;
            }
            if (PAIR == 1) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dX + (size_t)b * HT_T * N * C, 0, HT_T * N * C * 4, 0x00020000);
                st4_sc1(rs, (int)(((size_t)u * N + n0 + nl) * C + 4 * c4) * 4, acc);
                dp[u] = acc;
            } else if (!(HT_DBG(dbg) & 8)) st4(dX + rowoff + (size_t)u * N * C, acc);
            SB();
        }
    }
    SB();
    if (PAIR != 0) GPTST_STAMP(PAIR == 2 ? 8 : 3);
    // ---- dG_n[t,u] += sum_c dR_t[n,c] X_u[n,c]:  A[i=t][kk=c] = dR (LDS), B[kk=c][j=u] = X (registers) ----
#pragma unroll
    for (int i = 0; i < NT / 4; ++i) {
        const int nn = wave + 4 * i, n = n0 + nn;
        if (n >= N) continue;                                  // wave-uniform
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < C / 16; ++q) {
            const float4 a = j < HT_T ? ld4(Ds + (j * NT + nn) * P + 16 * q + 4 * kk) : f4zero();
            const float4 x = xg[i][q];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x.w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = kk * 4 + r, u = j;
            if (t < HT_T && u < HT_T && !(HT_DBG(dbg) & 2)) dG[((size_t)b * N + n) * 144 + t * HT_T + u] = acc[r];   // partial of sample b
        }
    }
    if (PAIR != 0) GPTST_STAMP(PAIR == 2 ? 9 : 4);
}

template <bool HASY, bool PREMUL>
__device__ __forceinline__ void hypertem_bwd_body(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                  const float* __restrict__ X, const float* __restrict__ G,
                                                  const float* __restrict__ Wbt, float* __restrict__ dX,
                                                  float* __restrict__ dbias, float* __restrict__ dG, int N, int B, int dbg, int L,
                                                  float* __restrict__ smem) {
    int b, tile;
    if (!ht_work_at(L, (N + 15) / 16, B, b, tile)) return;
    float4 dp[HT_T];
    hypertem_bwd_stage<HASY, PREMUL, 0>(dOut, Y, X, G, Wbt, dX, dbias, dG, N, B, dbg, b, tile, smem, dp);
}

template <bool HASY, bool PREMUL>
__global__ __launch_bounds__(256, HT_OCC) void hypertem_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                              const float* __restrict__ X, const float* __restrict__ G,
                                                              const float* __restrict__ Wbt, float* __restrict__ dX,
                                                              float* __restrict__ dbias, float* __restrict__ dG, int N, int B, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    hypertem_bwd_body<HASY, PREMUL>(dOut, Y, X, G, Wbt, dX, dbias, dG, N, B, dbg, blockIdx.x, smem);
}

// ---- weight-gradient role WITHOUT a saved R (r03) -----------------------------------------------------------------------------------------
// dW_bt = sum_n R_t[n,:]^T dPre_t[n,:] needs R_t[n,:] = sum_u G_n[t,u] X_u[n,:].  Keeping R costs a 16.7 MB write in the forward and a 16.7 MB
// read here; instead the (b,t) workgroup rebuilds its R rows from the sample's X — 12 float4 per (node, 4 channels), straight into the
// A-operand layout of wgrad64_body — in the summation order of the forward (bit-identical R).  X[b] (522 KB) is read by the slab workgroups of
// the same launch on the SAME XCD (work map below), so these 12x re-reads are L2 hits, not HBM traffic.  One row split only (N <= ~700).
template <bool HASY>
__device__ __forceinline__ void wgrad64_mix_body(const float* __restrict__ X, const float* __restrict__ G, const float* __restrict__ D,
                                                 const float* __restrict__ D2, float* __restrict__ dW, int N, int b, int t,
                                                 float* __restrict__ smem) {
    constexpr int C = 64, U = 2;
    float (*red)[C * C] = reinterpret_cast<float (*)[C * C]>(smem);
    float (*csred)[C] = reinterpret_cast<float (*)[C]>(smem + 4 * C * C);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int q = (N + 3) / 4;
    const int mbeg = wave * q, mend = min(N, mbeg + q);
    const size_t g = (size_t)b * HT_T + t;
    f32x4 acc[4][4];
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[ca][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 sa = f4zero();
    for (int m0 = mbeg; m0 < mend; m0 += 4 * U) {
        float4 x[U][HT_T], gq[U][3], d[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = min(m0 + 4 * u + kk, mend - 1);                            // clamped: out-of-range rows are zeroed below
#pragma unroll
            for (int k = 0; k < 3; ++k) gq[u][k] = ld4(G + (size_t)m * 144 + t * HT_T + 4 * k);
#pragma unroll
            for (int uu = 0; uu < HT_T; ++uu) x[u][uu] = ld4(X + (((size_t)b * HT_T + uu) * N + m) * C + 4 * j);
            const size_t off = (g * N + m) * C + 4 * j;
            d[u] = ld4(D + off);
            if (HASY) y[u] = ld4(D2 + off);
        }
        SB();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float gs[HT_T] = {gq[u][0].x, gq[u][0].y, gq[u][0].z, gq[u][0].w, gq[u][1].x, gq[u][1].y, gq[u][1].z, gq[u][1].w,
                                    gq[u][2].x, gq[u][2].y, gq[u][2].z, gq[u][2].w};
            float4 a = f4zero();
#pragma unroll
            for (int uu = 0; uu < HT_T; ++uu) a = f4fma(gs[uu], x[u][uu], a);            // same order as hypertem_fwd_kernel's mix
            if (m0 + 4 * u + kk >= mend) a = f4zero();
            if (HASY) {
                d[u].x = __fmul_rn(d[u].x, lrelu_grad_from_out(y[u].x)); d[u].y = __fmul_rn(d[u].y, lrelu_grad_from_out(y[u].y));
                d[u].z = __fmul_rn(d[u].z, lrelu_grad_from_out(y[u].z)); d[u].w = __fmul_rn(d[u].w, lrelu_grad_from_out(y[u].w));
            }
            if (m0 + 4 * u + kk < mend) sa = make_float4(__fadd_rn(sa.x, d[u].x), __fadd_rn(sa.y, d[u].y), __fadd_rn(sa.z, d[u].z), __fadd_rn(sa.w, d[u].w));
            const float av[4] = {a.x, a.y, a.z, a.w}, dv[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
#pragma unroll
            for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ca], dv[cb], acc[ca][cb], 0, 0, 0);
        }
        SB();
    }
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            st4(&red[wave][(4 * (kk * 4 + r) + ca) * C + 4 * j], make_float4(acc[ca][0][r], acc[ca][1][r], acc[ca][2][r], acc[ca][3][r]));
    sa.x += __shfl_xor(sa.x, 16, 64); sa.y += __shfl_xor(sa.y, 16, 64); sa.z += __shfl_xor(sa.z, 16, 64); sa.w += __shfl_xor(sa.w, 16, 64);
    sa.x += __shfl_xor(sa.x, 32, 64); sa.y += __shfl_xor(sa.y, 32, 64); sa.z += __shfl_xor(sa.z, 32, 64); sa.w += __shfl_xor(sa.w, 32, 64);
    if (kk == 0) st4(&csred[wave][4 * j], sa);
    __syncthreads();
    float* o = dW + g * (size_t)(C * C + C);
    if (threadIdx.x < C) o[C * C + threadIdx.x] = csred[0][threadIdx.x] + csred[1][threadIdx.x] + csred[2][threadIdx.x] + csred[3][threadIdx.x];
#pragma unroll
    for (int k = 0; k < C * C / 4 / 256; ++k) {
        const int f = threadIdx.x + k * 256;
        const float4 s = f4add(f4add(ld4(&red[0][4 * f]), ld4(&red[1][4 * f])), f4add(ld4(&red[2][4 * f]), ld4(&red[3][4 * f])));
        st4(o + 4 * f, s);
    }
}

// ---- hyperTem backward AND its weight gradient side by side in ONE launch ---------------------------------------------------------------
// Both consume (dOut, out) of the layer and are independent of each other: dR / dX / dG by the slab workgroups above, dW_bt (+ db_bt) by the
// grouped weight-gradient workgroups of wgrad64.h.  As two launches they cost their fixed dependency chains one after the other
// (tools/mb_scaling.py: ~10 us of every launch does not scale with the work); here workgroups 0 .. nH-1 take the hyperTem role (they are the
// longer ones and keep the XCD-aware index map), the rest the weight-gradient role, and the chains overlap.
// MIX: the weight-gradient role rebuilds R from X (R == NULL) and its workgroups follow the same sample -> XCD map as the slab role.
template <int U, bool HASY, bool PREMUL, bool MIX>
__global__ __launch_bounds__(256, HT_OCC) void hypertem_bwd_wgrad_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                                    const float* __restrict__ X, const float* __restrict__ G,
                                                                    const float* __restrict__ Wbt, const float* __restrict__ R,
                                                                    float* __restrict__ dX, float* __restrict__ dG, float* __restrict__ dWb,
                                                                    int N, int B, int nH, RowMap rm, int rows_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < nH) {
        hypertem_bwd_body<HASY, PREMUL>(dOut, Y, X, G, Wbt, dX, nullptr, dG, N, B, 0, blockIdx.x, smem);
    } else if (MIX) {
        int b, t;
        if (!ht_work_at(blockIdx.x - nH, HT_T, B, b, t)) return;      // nH is a multiple of 8: the XCD of this workgroup is (blockIdx.x - nH) % 8
        wgrad64_mix_body<HASY>(X, G, dOut, Y, dWb, N, b, t, smem);
    } else {
        const int w = blockIdx.x - nH;
        wgrad64_body<HASY ? PRO_DPRE : PRO_NONE, U>(R, dOut, Y, dWb, rm, rows_per_split, 64 * 64 + 64, 2, w % rm.G, w / rm.G, smem);
    }
}

// dbias: (gptst_hypertem_ntiles(N) * B*T, C) node-tile partials;  dG: (B * N, T, T) per-sample partials — both fully written here.
extern "C" int gptst_hypertem_ntiles(int N) { return (N + 15) / 16; }

// Y == NULL: dOut already is dPre.  premul != 0: dX is multiplied by lrelu'(X) (X = output of the LeakyReLU layer below) — not together with Y.
extern "C" int gptst_hypertem_bwd(const float* dOut, const float* Y, const float* X, const float* G, const float* Wbt, float* dX,
                                  float* dbias, float* dG, int premul, int B, int T, int N, int C, void* stream) {
    if (!dOut || !X || !G || !Wbt || !dX || !dG || T != HT_T || (Y && premul)) return GPTST_EARG;      // dbias may be NULL (bias gradient from gptst_wgrad_colsum)
    if (C != 64) return GPTST_ESHAPE;
    const size_t smem = ht_smem(16);
    static int done = 0;
    if (!done) {
        hipFuncSetAttribute((const void*)hypertem_bwd_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute((const void*)hypertem_bwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute((const void*)hypertem_bwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        done = 1;
    }
    const dim3 grid(8 * ((B + 7) / 8) * ((N + 15) / 16));
    if (Y) hipLaunchKernelGGL((hypertem_bwd_kernel<true, false>), grid, dim3(256), smem, (hipStream_t)stream, dOut, Y, X, G, Wbt, dX, dbias, dG, N, B, g_ht_dbg);
    else if (premul) hipLaunchKernelGGL((hypertem_bwd_kernel<false, true>), grid, dim3(256), smem, (hipStream_t)stream, dOut, Y, X, G, Wbt, dX, dbias, dG, N, B, g_ht_dbg);
    else hipLaunchKernelGGL((hypertem_bwd_kernel<false, false>), grid, dim3(256), smem, (hipStream_t)stream, dOut, Y, X, G, Wbt, dX, dbias, dG, N, B, g_ht_dbg);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// One launch for the whole backward of a hyperTem layer (C = 64): gptst_hypertem_bwd (without dbias) + gptst_wgrad_colsum(mode 0, pro 1,
// which 2) on (R, dOut, Y).  dWb: (nsplit * B*T, C*C + C) rows [dW_bt | db_bt] with nsplit = gptst_wgrad_nsplit(0, B*T, N, 64).
// Y == NULL / premul: as gptst_hypertem_bwd.  R == NULL: the weight-gradient role rebuilds R from X (needs nsplit == 1, else GPTST_ESHAPE).
extern "C" int gptst_wgrad_nsplit(int mode, int BT, int N, int C);
template <bool HASY, bool PREMUL>
static void ht_bwd_wgrad_launch(const float* dOut, const float* Y, const float* X, const float* G, const float* Wbt, const float* R, float* dX,
                                float* dG, float* dWb, int B, int N, bool u6, int nH, int nW, RowMap rm, int rps, size_t smem, hipStream_t st) {
    static int done = 0;
    if (!done) {
        hipFuncSetAttribute((const void*)hypertem_bwd_wgrad_kernel<4, HASY, PREMUL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute((const void*)hypertem_bwd_wgrad_kernel<6, HASY, PREMUL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute((const void*)hypertem_bwd_wgrad_kernel<4, HASY, PREMUL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        done = 1;
    }
    if (R == nullptr) hipLaunchKernelGGL((hypertem_bwd_wgrad_kernel<4, HASY, PREMUL, true>), dim3(nH + 8 * ((B + 7) / 8) * HT_T), dim3(256), smem, st, dOut, Y, X, G, Wbt, R, dX, dG, dWb, N, B, nH, rm, rps);
    else if (u6) hipLaunchKernelGGL((hypertem_bwd_wgrad_kernel<6, HASY, PREMUL, false>), dim3(nH + nW), dim3(256), smem, st, dOut, Y, X, G, Wbt, R, dX, dG, dWb, N, B, nH, rm, rps);
    else hipLaunchKernelGGL((hypertem_bwd_wgrad_kernel<4, HASY, PREMUL, false>), dim3(nH + nW), dim3(256), smem, st, dOut, Y, X, G, Wbt, R, dX, dG, dWb, N, B, nH, rm, rps);
}

extern "C" int gptst_hypertem_bwd_wgrad(const float* dOut, const float* Y, const float* X, const float* G, const float* Wbt, const float* R,
                                        float* dX, float* dG, float* dWb, int premul, int B, int T, int N, int C, void* stream) {
    if (!dOut || !X || !G || !Wbt || !dX || !dG || !dWb || T != HT_T || (Y && premul)) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    const size_t smem_h = ht_smem(16), smem_w = WGRAD64_SMEM_FLOATS * sizeof(float), smem = smem_h > smem_w ? smem_h : smem_w;
    const RowMap rm = make_rowmap(0, B * T, N);
    const int ns = gptst_wgrad_nsplit(0, B * T, N, 64);
    if (!R && ns != 1) return GPTST_ESHAPE;
    int rps = (rm.M + ns - 1) / ns;
    rps = (rps + 1) & ~1;
    const int rows = rps < rm.M ? rps : rm.M, steps = ((rows + 3) / 4 + 3) / 4;          // k-steps per wave (as wgrad_impl, apply.hip)
    const bool u6 = (steps + 5) / 6 * 6 <= (steps + 3) / 4 * 4;
    const int nH = 8 * ((B + 7) / 8) * ((N + 15) / 16), nW = rm.G * ns;
    if (Y) ht_bwd_wgrad_launch<true, false>(dOut, Y, X, G, Wbt, R, dX, dG, dWb, B, N, u6, nH, nW, rm, rps, smem, (hipStream_t)stream);
    else if (premul) ht_bwd_wgrad_launch<false, true>(dOut, Y, X, G, Wbt, R, dX, dG, dWb, B, N, u6, nH, nW, rm, rps, smem, (hipStream_t)stream);
    else ht_bwd_wgrad_launch<false, false>(dOut, Y, X, G, Wbt, R, dX, dG, dWb, B, N, u6, nH, nW, rm, rps, smem, (hipStream_t)stream);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// ---- TWO consecutive hyperTem layers' backward in ONE launch (r04) --------------------------------------------------------------------------
// Layers L+1 ("1", the upper one: its dPre is the input) and L ("0") of a chain hyperTem(L) -> hyperTem(L+1) with nothing in between
// (GPTST.py:267-268 hyperTem2 -> hyperTem3; :271 / :454 the encoder's hyperTem4 -> the decoder's hyperTem1).  The input gradient of layer L+1,
// multiplied by lrelu'(its input), IS dPre of layer L for the same (b, 16-node, all-t) slab: the slab workgroup keeps it in registers / LDS and
// runs layer L right away — no launch boundary, no 16.7 MB read-back in front of the second dependency chain.  The weight gradient of layer L
// needs dPre_L of ALL nodes of a (b,t): its workgroups sit at the END of the grid [slab | wgrad L+1 | wgrad L] and wait for the sample's slab
// workgroups to have published stage 1 (write-through stores + one counter per sample): every wait points to a lower block index, so with
// blocks dispatched in index order nothing waits on a workgroup that is not yet on the chip; the wait is bounded and ends in NaN rows of dWb0.
struct HtPairArgs {
    const float* dOut1; const float* X1; const float* G1; const float* Wbt1; const float* R1;
    const float* X0; const float* G0; const float* Wbt0; const float* R0;
    float* dXmid; float* dX0; float* dG1; float* dG0; float* dWb1; float* dWb0; unsigned* cnt;
};

template <int U>
__global__ __launch_bounds__(256, HT_OCC) void hypertem_bwd_pair_kernel(HtPairArgs a, int N, int B, int nH, RowMap rm, int rows_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ntiles = (N + 15) / 16;
    if ((int)blockIdx.x < nH) {
        int b, tile;
        if (!ht_work_at(blockIdx.x, ntiles, B, b, tile)) return;
        float4 dp[HT_T];
        GPTST_WG_BEGIN(); GPTST_STAMP(0);
        hypertem_bwd_stage<false, true, 1>(a.dOut1, nullptr, a.X1, a.G1, a.Wbt1, a.dXmid, nullptr, a.dG1, N, B, 0, b, tile, smem, dp);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's write-through stores of dPre_L have left ...
        __syncthreads();                                              // ... (all waves; and the dG phase is done with the LDS slab)
        if (threadIdx.x == 0) { gptst_publish_fence(); __hip_atomic_fetch_add(a.cnt + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        GPTST_STAMP(5);
        hypertem_bwd_stage<false, true, 2>(nullptr, nullptr, a.X0, a.G0, a.Wbt0, a.dX0, nullptr, a.dG0, N, B, 0, b, tile, smem, dp);
        GPTST_WG_END();
        return;
    }
    GPTST_WG_BEGIN();
    const int nW = rm.G * ((rm.M + rows_per_split - 1) / rows_per_split);
    int w = blockIdx.x - nH;
    if (w < nW) {
        wgrad64_body<PRO_NONE, U>(a.R1, a.dOut1, nullptr, a.dWb1, rm, rows_per_split, 64 * 64 + 64, 2, w % rm.G, w / rm.G, smem);
        GPTST_WG_END();
        return;
    }
    w -= nW;
    const int g = w % rm.G, sp = w / rm.G;
    unsigned* s_ok = reinterpret_cast<unsigned*>(smem);
    if (threadIdx.x == 0) *s_ok = gptst_wait_ge(a.cnt + g / HT_T, (unsigned)ntiles, &g_handoff_lost_hypertem) ? 1u : 0u;
    __syncthreads();
    const bool ok = *s_ok != 0u;
    __syncthreads();
    wgrad64_body<PRO_NONE, U, true>(a.R0, a.dXmid, nullptr, a.dWb0, rm, rows_per_split, 64 * 64 + 64, 2, g, sp, smem);
    if (!ok) a.dWb0[((size_t)sp * rm.G + g) * (size_t)(64 * 64 + 64) + threadIdx.x] = __int_as_float(0x7fc00000);    // lost hand-off: loud
    GPTST_WG_END();
}

// dOut1 = dPre of layer L+1; X1 / X0 the layers' inputs (X1 = output of layer L), R1 / R0 their saved mixes; -> dXmid (= dPre of layer L),
// dX0 (multiplied by lrelu'(X0)), dG1 / dG0 (B*N, T, T) partials, dWb1 / dWb0 (nsplit * B*T, C*C + C) rows [dW_bt | db_bt].
// cnt: B 32-bit words, ZERO on entry.
extern "C" int gptst_hypertem_bwd_pair(const float* dOut1, const float* X1, const float* G1, const float* Wbt1, const float* R1, const float* X0,
                                       const float* G0, const float* Wbt0, const float* R0, float* dXmid, float* dX0, float* dG1, float* dG0,
                                       float* dWb1, float* dWb0, void* cnt, int B, int T, int N, int C, void* stream) {
    if (!dOut1 || !X1 || !G1 || !Wbt1 || !R1 || !X0 || !G0 || !Wbt0 || !R0 || !dXmid || !dX0 || !dG1 || !dG0 || !dWb1 || !dWb0 || !cnt || T != HT_T)
        return GPTST_EARG;
    if (C != 64 || (size_t)HT_T * N * C * 4 >= ((size_t)1 << 31)) return GPTST_ESHAPE;
    const size_t smem_h = ht_smem(16), smem_w = WGRAD64_SMEM_FLOATS * sizeof(float), smem = smem_h > smem_w ? smem_h : smem_w;
    const RowMap rm = make_rowmap(0, B * T, N);
    const int ns = gptst_wgrad_nsplit(0, B * T, N, 64);
    int rps = (rm.M + ns - 1) / ns;
    rps = (rps + 1) & ~1;
    if ((rm.M + rps - 1) / rps != ns || (size_t)B * T * N * C * 4 >= ((size_t)1 << 31)) return GPTST_ESHAPE;
    const int rows = rps < rm.M ? rps : rm.M, steps = ((rows + 3) / 4 + 3) / 4;
    const bool u6 = (steps + 5) / 6 * 6 <= (steps + 3) / 4 * 4;
    const int nH = 8 * ((B + 7) / 8) * ((N + 15) / 16), nW = rm.G * ns;
    static int done = 0;
    if (!done) {
        hipFuncSetAttribute((const void*)hypertem_bwd_pair_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute((const void*)hypertem_bwd_pair_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        done = 1;
    }
    const HtPairArgs a{dOut1, X1, G1, Wbt1, R1, X0, G0, Wbt0, R0, dXmid, dX0, dG1, dG0, dWb1, dWb0, (unsigned*)cnt};
    // (all of a wave's k-steps as ONE batch of loads — the launch allocates 256 VGPRs for the slab role anyway — measured 832.6 vs 834.0 steps/s
    // for the batches of 6: no gain, not kept)
    if (u6) hipLaunchKernelGGL((hypertem_bwd_pair_kernel<6>), dim3(nH + 2 * nW), dim3(256), smem, (hipStream_t)stream, a, N, B, nH, rm, rps);
    else hipLaunchKernelGGL((hypertem_bwd_pair_kernel<4>), dim3(nH + 2 * nW), dim3(256), smem, (hipStream_t)stream, a, N, B, nH, rm, rps);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// =====================================================================================================================
// hyperTem forward CHAIN (round 4): up to three consecutive hyperTem layers in ONE launch on the (b, 16-node) slab.
//
// Every one of these layers is node-local: a workgroup that owns (sample b, 16 nodes, all 12 time steps) needs nothing from any other
// workgroup until the next cap (whose sums over the nodes of a (b,t) are the only coupling in an STHCN).  As separate launches each layer pays
// its own load phase (the 16.7 MB activation comes back from L2 / HBM while every workgroup of the one-round grid waits: 7.6 us of the 22 us,
// DESIGN.md section 7) and its own fill / drain; here the layer output goes from the accumulators straight back into the LDS slab (and to HBM
// once, for the backward) and the next layer starts from it.  Chains of the step: [hyperTem2, hyperTem3], [hyperTem4, the decoder's hyperTem1]
// (GPTST.py:267-271, :454).  (r04 also ran the cap's node-conditioned layer as a first stage: measured slower than the node-grouped apply64
// and removed in r05 — tools/experiments/hypertem_chain_node_layer.hip.)
struct HtStage { const float* G; const float* Wbt; const float* bbt; float* R_out; float* out; };
struct HtChain {
    int nstage;                                                   // hyperTem layers: 1 .. 3
    const float* X;                                               // slab source
    HtStage st[3];
};

// NSTAGE is compile-time (the run-time form — one kernel, stage loop over ch.nstage — spilled 250-400 registers: the allocator saw
// the fragments of every path live around the loop)
// HTC_NW waves per workgroup share the 12 time steps: 4 (three steps per wave, two waves per SIMD at the LDS-imposed two workgroups per CU).  r05 experiment,
// -DHTC_NW=6: two steps per wave, 12 waves per CU = three waves per SIMD within 168 VGPRs (15 spilled address registers outside the loops) — the resident-wave
// lever of VERDICT r04 item 2 without any exchange between workgroups — measured 876 -> 849 steps/s (profiles/r05_ab_chain_six_waves.txt): not adopted.
#ifndef HTC_NW
#define HTC_NW 4
#endif
template <int NSTAGE>
__global__ __launch_bounds__(64 * HTC_NW, HTC_NW == 6 ? 3 : HT_OCC) void hypertem_chain_fwd_kernel(HtChain ch, int N, int B) {
    constexpr int C = 64, P = C + 4, GP = 145, NT = 16, NW = HTC_NW, NTH = 64 * NW, TPW = HT_T / NW;      // TPW time steps per wave
    static_assert(HT_T % NW == 0 && (HT_T * NT * (C / 4)) % NTH == 0 && (NT * 144) % NTH == 0, "waves must divide the steps and the staging loops");
    constexpr int GK = NT * 144 / NTH, XK = HT_T * NT * (C / 4) / NTH;                                     // staging trips per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                               // [12][NT][P]
    float* Gs = Xs + HT_T * NT * P;                 // [NT][GP]
    int b, tile;
    if (!ht_work((N + NT - 1) / NT, B, b, tile)) return;
    const int n0 = tile * NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    float gv[GK];
#define HTC_LOAD_G(Gp) _Pragma("unroll") for (int k = 0; k < GK; ++k) gv[k] = (Gp)[min(n0 * 144 + tid + k * NTH, N * 144 - 1)]
#define HTC_STORE_G() _Pragma("unroll") for (int k = 0; k < GK; ++k) { const int i = tid + k * NTH; \
        Gs[(i / 144) * GP + i % 144] = (n0 + i / 144 < N) ? gv[k] : 0.f; }
    float4 bv[C / 16][4];
    float4 b4;
#define HTC_LOAD_W(W0, b0, tt) do {                                                                                \
        const float* W_ = (W0) + ((size_t)b * HT_T + (tt)) * C * C;                                                \
        _Pragma("unroll") for (int q = 0; q < C / 16; ++q)                                                         \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) bv[q][e] = ld4(W_ + (size_t)(16 * q + 4 * kk + e) * C + 4 * j); \
        b4 = ld4((b0) + ((size_t)b * HT_T + (tt)) * C + 4 * j);                                                    \
    } while (0)

    {
        // ---- slab + graph staging: ALL global loads before the first LDS store (as hypertem_fwd_kernel) ----
        // trip k of thread tid: float4 i = tid + k * NTH of the (12, NT, 16) slab -> (time step i / 256, row (i % 256) / 16, column i % 16)
        float4 v[XK];
#pragma unroll
        for (int k = 0; k < XK; ++k) {
            const int i = tid + k * NTH, t = i >> 8, nl = (i & 255) >> 4, c4 = i & 15;
            v[k] = ld4(ch.X + (((size_t)b * HT_T + t) * N + min(n0 + nl, N - 1)) * C + 4 * c4);
        }
        HTC_LOAD_G(ch.st[0].G);
#pragma unroll
        for (int k = 0; k < XK; ++k) {
            const int i = tid + k * NTH, t = i >> 8, nl = (i & 255) >> 4, c4 = i & 15;
            st4(Xs + (t * NT + nl) * P + 4 * c4, n0 + nl < N ? v[k] : f4zero());
        }
        HTC_STORE_G();
    }
    // W_bt fragments are never live across the mix phase of a non-final layer (64 registers on top of the three kept mixes): every layer
    // requests its first fragments itself — the final one at its start, a non-final one right behind its mixes
    __syncthreads();
    SB();

#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) {
        const HtStage S = ch.st[s];
        const HtStage S1 = ch.st[s + 1 < 3 ? s + 1 : 2];
        if (s + 1 < NSTAGE) {
            // ---- a layer whose output is the next layer's slab: all mixes first (they read every time step of the slab), then — behind a
            //      barrier — MFMA + epilogue, writing the output rows IN PLACE (time step t of the slab is only touched by its own wave now) ----
            float4 a4a[C / 16], a4b[C / 16], a4c[C / 16];          // named arrays: an [3][4] array indexed by the rolled loop below went to scratch (a4c: TPW = 3 only)
#pragma unroll
            for (int q = 0; q < C / 16; ++q) { a4a[q] = f4zero(); a4b[q] = f4zero(); a4c[q] = f4zero(); }
            {   // the three mixes of this wave share every slab operand: each X_u row piece is read ONCE and feeds the three time steps
                const float* gr = Gs + j * GP + wave * HT_T;
                const float* xr = Xs + j * P + 4 * kk;
#pragma unroll
                for (int u = 0; u < HT_T; ++u) {
                    float4 x[C / 16];
#pragma unroll
                    for (int q = 0; q < C / 16; ++q) x[q] = ld4(xr + u * NT * P + 16 * q);
                    const float g0 = gr[u], g1 = gr[NW * HT_T + u], g2 = TPW == 3 ? gr[2 * NW * HT_T + u] : 0.f;
#pragma unroll
                    for (int q = 0; q < C / 16; ++q) {
                        a4a[q] = f4fma(g0, x[q], a4a[q]);
                        a4b[q] = f4fma(g1, x[q], a4b[q]);
                        if (TPW == 3) a4c[q] = f4fma(g2, x[q], a4c[q]);
                    }
                    if (TPW == 3 ? u % 3 == 2 : u % 2 == 1) SB();   // at most three (two) time steps' operands (48 / 32 registers) in flight
                }
            }
            HTC_LOAD_W(S.Wbt, S.bbt, wave);
            HTC_LOAD_G(S1.G);                                     // next layer's temporal graphs: in flight during the MFMA phase
            __syncthreads();
            SB();
#pragma nounroll
            for (int ti = 0; ti < TPW; ++ti) {                     // rolled: unrolled, the addresses of all three steps' stores / loads were live at once
                const int t = wave + NW * ti;
                const size_t g = (size_t)b * HT_T + t;
                float4 m4[C / 16];                                 // this step's mix; the kept ones rotate down (register moves: a select by ti
#pragma unroll                                                     //  turned the three arrays into one scratch array)
                for (int q = 0; q < C / 16; ++q) { m4[q] = a4a[q]; a4a[q] = a4b[q]; a4b[q] = a4c[q]; }
                f32x4 acc[C / 16];
#pragma unroll
                for (int ct = 0; ct < C / 16; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < C / 16; ++q) {
                    const float av[4] = {m4[q].x, m4[q].y, m4[q].z, m4[q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].x, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].y, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].z, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].w, acc[3], 0, 0, 0);
                    }
                }
                SB();
                const float4 bias = b4;
                if (ti < TPW - 1) HTC_LOAD_W(S.Wbt, S.bbt, t + NW);
                SB();
                if (S.R_out != nullptr && n0 + j < N) {
#pragma unroll
                    for (int q = 0; q < C / 16; ++q) st4_far(S.R_out + (g * N + n0 + j) * C + 16 * q + 4 * kk, m4[q]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nl = kk * 4 + r;
                    float* xs = Xs + (t * NT + nl) * P + 4 * j;
                    float4 y = f4add(f4add(make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), bias), ld4(xs));
                    y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
                    if (n0 + nl < N) st4(S.out + (g * N + n0 + nl) * C + 4 * j, y);
                    else y = f4zero();
                    st4(xs, y);
                }
                SB();
            }
            HTC_STORE_G();
            __syncthreads();
            SB();
        } else {
            // ---- last layer of the chain: per time step mix -> MFMA -> epilogue, as hypertem_fwd_kernel ----
            HTC_LOAD_W(S.Wbt, S.bbt, wave);
            for (int t = wave; t < HT_T; t += NW) {
                const size_t g = (size_t)b * HT_T + t;
                SB();
                float4 a4[C / 16];
#pragma unroll
                for (int q = 0; q < C / 16; ++q) a4[q] = f4zero();
                {
                    const float* gr = Gs + j * GP + t * HT_T;
                    const float* xr = Xs + j * P + 4 * kk;
#pragma unroll
                    for (int u = 0; u < HT_T; ++u) {
                        const float gu = gr[u];
#pragma unroll
                        for (int q = 0; q < C / 16; ++q) a4[q] = f4fma(gu, ld4(xr + u * NT * P + 16 * q), a4[q]);
                    }
                }
                SB();
                f32x4 acc[C / 16];
#pragma unroll
                for (int ct = 0; ct < C / 16; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < C / 16; ++q) {
                    const float av[4] = {a4[q].x, a4[q].y, a4[q].z, a4[q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].x, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].y, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].z, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].w, acc[3], 0, 0, 0);
                    }
                }
                SB();
                const float4 bias = b4;
                if (t + NW < HT_T) HTC_LOAD_W(S.Wbt, S.bbt, t + NW);
                SB();
                if (S.R_out != nullptr && n0 + j < N) {
#pragma unroll
                    for (int q = 0; q < C / 16; ++q) st4_far(S.R_out + (g * N + n0 + j) * C + 16 * q + 4 * kk, a4[q]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nl = kk * 4 + r;
                    if (n0 + nl < N) {
                        float4 y = f4add(f4add(make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), bias), ld4(Xs + (t * NT + nl) * P + 4 * j));
                        y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
                        st4(S.out + (g * N + n0 + nl) * C + 4 * j, y);
                    }
                }
            }
        }
    }
#undef HTC_LOAD_G
#undef HTC_STORE_G
#undef HTC_LOAD_W
}

// X: input of the first hyperTem layer;  Gs .. outs: HOST arrays of nstage device pointers (G (N,T,T), Wbt (BT,C,C), bbt (BT,C), R_out or NULL,
// out), read at call time.  C = 64 only (GPTST_ESHAPE otherwise: use the per-layer entry points).
extern "C" int gptst_hypertem_chain_fwd(const float* X, int nstage, const void* Gs, const void* Wbts, const void* bbts, const void* Rs,
                                        const void* outs, int B, int T, int N, int C, void* stream) {
    if (!X || nstage < 1 || nstage > 3 || !Gs || !Wbts || !bbts || !Rs || !outs || T != HT_T || B <= 0 || N <= 0) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    HtChain ch;
    ch.nstage = nstage; ch.X = X;
    for (int s = 0; s < 3; ++s) {
        const int k = s < nstage ? s : nstage - 1;
        ch.st[s] = HtStage{((const float* const*)Gs)[k], ((const float* const*)Wbts)[k], ((const float* const*)bbts)[k],
                           ((float* const*)Rs)[k], ((float* const*)outs)[k]};
        if (!ch.st[s].G || !ch.st[s].Wbt || !ch.st[s].bbt || !ch.st[s].out) return GPTST_EARG;
    }
    const dim3 grid(8 * ((B + 7) / 8) * ((N + 15) / 16));
    const int smem = (int)ht_smem(16);
    hipStream_t st = (hipStream_t)stream;
#define HTC_LAUNCH(NS_) do {                                                                                                          \
        static int done_ = 0;                                                                                                          \
        if (!done_) { (void)hipFuncSetAttribute((const void*)hypertem_chain_fwd_kernel<NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); done_ = 1; } \
        hipLaunchKernelGGL((hypertem_chain_fwd_kernel<NS_>), grid, dim3(64 * HTC_NW), smem, st, ch, N, B);                                     \
    } while (0)
    if (nstage == 1) HTC_LAUNCH(1); else if (nstage == 2) HTC_LAUNCH(2); else HTC_LAUNCH(3);
#undef HTC_LAUNCH
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
