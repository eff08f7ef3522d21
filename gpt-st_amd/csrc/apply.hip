// "Customised parameter" contractions of GPT-ST on fp32 MFMA (gfx950).
//
//   apply :  out[g,m,:] = epi( pro(A)[g,m,:] @ W[g] (+ bias[g]) (+ resid[g,m,:]) )
//   wgrad :  dW[g]      = sum_m A[g,m,:]^T  pro(D)[g,m,:]
//
// replacing the reference's  einsum('btni,btio->btno') / einsum('btni,nio->btno')  + bias + residual + LeakyReLU
// (GPTST.py:26-27,31-32,139-141,162-163), nn.Linear C->C (GPTST.py:102) and their autograd backward.
// Activations are (B*T*N, C) row-major.  A "group" g shares one C x C weight:
//   mode 0 (TIME)   g = (b,t), rows m = n      : row = g*N + m          (time-conditioned weights, hyperTem / MLP_RL)
//   mode 1 (NODE)   g = n,     rows m = (b,t)  : row = m*N + g          (node-conditioned weights, cap / MLP_RL)
//   mode 2 (SHARED) one group, rows = all                                (nn.Linear)
#include "mfma_tile.h"
#include "wgrad64.h"
#ifdef GPTST_DEBUG
__device__ long long g_ap_ts[64];     // per-phase s_memtime stamps (enabled by gptst_ap_dbg(1))
static thread_local int g_ap_dbg = 0;
extern "C" int gptst_ap_dbg(int v) { g_ap_dbg = v; return 0; }
extern "C" int gptst_ap_ts(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ap_ts), sizeof(long long) * 64); }
#else
static constexpr int g_ap_dbg = 0;
#endif
// launch-geometry knobs of gptst_tune: thread-local (ranks emulated by threads must not see each other's experiments)
thread_local int g_apply_tpw = 0;                      // gptst_tune(4, n) forces tiles per wave of apply64 / apply128
thread_local int g_apply128_minwg = 768;                // gptst_tune(24, n): apply128_geometry halves tiles per wave until n workgroups exist
thread_local int g_apply128_v1 = 0;                    // gptst_tune(8, 1) selects the first-generation apply_kernel for C = 128

template <int C, int PRO, int EPI>
__global__ __launch_bounds__(256) void apply_kernel(const float* __restrict__ A, const float* __restrict__ A2,
                                                    const float* __restrict__ W, long w_gstride, int transw,
                                                    const float* __restrict__ bias, const float* __restrict__ resid,
                                                    const float* __restrict__ resid2, float* __restrict__ out, float* __restrict__ colsum, RowMap rm, int dbg) {
    using T = Tile<C>;
    int tsi = 0;
#ifdef GPTST_DEBUG
#define TS() do { if (dbg && blockIdx.x == 100 && blockIdx.y == 0 && threadIdx.x == 0) g_ap_ts[tsi] = clock64(); ++tsi; } while (0)
#else
#define TS() do { } while (0)
#endif
    TS();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                   // C*C
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* tile = smem + C * C + wave * T::TILE_FLOATS;
    const int g = blockIdx.x;
    const int ntiles = (rm.M + 31) / 32;
    float cs[C / 64];
#pragma unroll
    for (int u = 0; u < C / 64; ++u) cs[u] = 0.f;
    const int tstride = gridDim.y * 4;
    int t = blockIdx.y * 4 + wave;

    // The A tile of the first (usually only) tile is requested from HBM BEFORE the weight is staged, so the two round trips
    // overlap; likewise the residual operands of the epilogue are requested before the MFMA phase.
    float4 av[T::F4_PER_LANE], ov[T::F4_PER_LANE];
    auto fetch_a = [&](int tt) {
        const int m0 = tt * 32;
#pragma unroll
        for (int it = 0; it < T::F4_PER_LANE; ++it) {
            const int f = it * 64 + lane;
            const int r = f / T::F4_PER_ROW, c4 = f % T::F4_PER_ROW;
            av[it] = f4zero(); ov[it] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (tt < ntiles && m0 + r < rm.M) {
                const size_t off = ((size_t)g * rm.rs_g + (size_t)(m0 + r) * rm.rs_m) * C + 4 * c4;
                av[it] = ld4(A + off);
                if (PRO == PRO_DPRE) ov[it] = ld4(A2 + off);
            }
        }
    };
    fetch_a(t);
    TS();
    load_w_lds<C, 256>(Wl, W + (size_t)g * w_gstride, transw, tid);
    __syncthreads();
    TS();

    for (; t < ntiles; t += tstride) {
        const int m0 = t * 32;
        // ---- stage the prefetched A tile ----
#pragma unroll
        for (int it = 0; it < T::F4_PER_LANE; ++it) {
            const int f = it * 64 + lane;
            const int r = f / T::F4_PER_ROW, c4 = f % T::F4_PER_ROW;
            float4 v = av[it];
            if (PRO == PRO_DPRE) {
                const float4 o = ov[it];
                v.x *= lrelu_grad_from_out(o.x); v.y *= lrelu_grad_from_out(o.y);
                v.z *= lrelu_grad_from_out(o.z); v.w *= lrelu_grad_from_out(o.w);
            }
            st4(tile + r * T::PITCH + 4 * c4, v);
        }
        // ---- request the epilogue operands now (they land while the MFMAs run) ----
        float4 rv[T::F4_PER_LANE], rv2[T::F4_PER_LANE];
#pragma unroll
        for (int it = 0; it < T::F4_PER_LANE; ++it) {
            const int f = it * 64 + lane;
            const int r = f / T::F4_PER_ROW, c4 = f % T::F4_PER_ROW;
            rv[it] = f4zero(); rv2[it] = f4zero();
            if ((EPI == EPI_RES_LRELU || EPI == EPI_ADD_DPRE) && m0 + r < rm.M) {
                const size_t off = ((size_t)g * rm.rs_g + (size_t)(m0 + r) * rm.rs_m) * C + 4 * c4;
                rv[it] = ld4(resid + off);
                if (EPI == EPI_ADD_DPRE) rv2[it] = ld4(resid2 + off);
            }
        }
        TS();
        if (colsum != nullptr) {      // column sums of the staged tile (bias gradient)
#pragma unroll
            for (int u = 0; u < C / 64; ++u) {
                float s = 0.f;
#pragma unroll 8
                for (int r = 0; r < 32; ++r) s += tile[r * T::PITCH + u * 64 + lane];
                cs[u] += s;
            }
        }
        f32x16 acc[T::NCT];
        TS();
        mfma_tile<C>(tile, Wl, acc, lane);
        TS();
        acc_to_tile<C>(tile, acc, lane);
        TS();
        if (t + tstride < ntiles) fetch_a(t + tstride);      // next tile of a persistent wave
        // ---- epilogue: row-major float4 ----
#pragma unroll
        for (int it = 0; it < T::F4_PER_LANE; ++it) {
            const int f = it * 64 + lane;
            const int r = f / T::F4_PER_ROW, c4 = f % T::F4_PER_ROW;
            if (m0 + r < rm.M) {
                const size_t off = ((size_t)g * rm.rs_g + (size_t)(m0 + r) * rm.rs_m) * C + 4 * c4;
                float4 y = ld4(tile + r * T::PITCH + 4 * c4);
                if (bias != nullptr) y = f4add(y, ld4(bias + (size_t)g * C + 4 * c4));
                if (EPI == EPI_RES_LRELU) {
                    y = f4add(y, rv[it]);
                    y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
                }
                if (EPI == EPI_LRELU) { y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w); }
                if (EPI == EPI_ADD_DPRE) {
                    const float4 d = rv[it], o = rv2[it];
                    y.x = fmaf(d.x, lrelu_grad_from_out(o.x), y.x); y.y = fmaf(d.y, lrelu_grad_from_out(o.y), y.y);
                    y.z = fmaf(d.z, lrelu_grad_from_out(o.z), y.z); y.w = fmaf(d.w, lrelu_grad_from_out(o.w), y.w);
                }
                st4(out + off, y);
            }
        }
        TS();
    }
    if (colsum != nullptr) {          // fold the 4 waves in LDS: one partial per column and workgroup
        __syncthreads();
#pragma unroll
        for (int u = 0; u < C / 64; ++u) tile[u * 64 + lane] = cs[u];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int u = 0; u < C / 64; ++u) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) s += smem[C * C + w * T::TILE_FLOATS + u * 64 + lane];
                colsum[((size_t)blockIdx.y * rm.G + g) * C + u * 64 + lane] = s;      // partial of this row split (plain store)
            }
        }
    }
}

// C = 64 apply with register-resident operands, one barrier (second generation; apply_kernel above stays for C = 128):
//   a wave keeps the whole C x C weight of its group as MFMA B fragments in registers (16 float4 = 64 VGPRs, staged once per
//   workgroup through LDS) and walks `tiles_per_wave` 16-row tiles; the A fragment of a lane is row (tile*16 + j), channels 16q+4kk..+3, loaded straight
//   from global as float4; accumulator tile ct, column j stands for output channel 4j+ct, so a lane owns four CONSECUTIVE channels
//   of rows kk*4+r and the epilogue (bias / residual / LReLU / dPre) runs from registers with float4 loads and stores.
//   The next tile's operands are requested before the current tile's stores (vmcnt retires in order).
template <int PRO, int EPI>
__global__ __launch_bounds__(256, 2) void apply64_kernel(const float* __restrict__ A, const float* __restrict__ A2,
                                                         const float* __restrict__ W, long w_gstride, int transw,
                                                         const float* __restrict__ bias, const float* __restrict__ resid,
                                                         const float* __restrict__ resid2, float* __restrict__ out,
                                                         float* __restrict__ colsum, RowMap rm, int tiles_per_wave) {
    constexpr int C = 64;
    __shared__ __attribute__((aligned(16))) float Wl[C * C];
    __shared__ __attribute__((aligned(16))) float csl[4][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int g = blockIdx.x;
    const int ntiles = (rm.M + 15) / 16;
    const int t0 = (blockIdx.y * 4 + wave) * tiles_per_wave, t1 = min(ntiles, t0 + tiles_per_wave);
    const float* Wg = W + (size_t)g * w_gstride;
    float4 bias4 = f4zero();
    if (bias != nullptr) bias4 = ld4(bias + (size_t)g * C + 4 * j);
    float4 cs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cs[q] = f4zero();
    float4 an[4], on[4];
    auto fetch = [&](int t) {
        const int m = min(t * 16 + j, rm.M - 1);
        const size_t off = ((size_t)g * rm.rs_g + (size_t)m * rm.rs_m) * C + 4 * kk;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            an[q] = ld4(A + off + 16 * q);
            if (PRO == PRO_DPRE) on[q] = ld4(A2 + off + 16 * q);
        }
    };
    if (t0 < t1) fetch(t0);                          // in flight while the weight is staged
    // the group's weight goes through LDS once per workgroup (coalesced, transposed on the way if needed) and from there into
    // the B fragments of every wave: bv[q][e], components = column tile ct  (reading fragments straight from L2 cost 4x the traffic)
    load_w_lds<C, 256>(Wl, Wg, transw, threadIdx.x);
    __syncthreads();
    float4 bv[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[q][e] = ld4(Wl + (16 * q + 4 * kk + e) * C + 4 * j);
    for (int t = t0; t < t1; ++t) {
        float4 a[4];
        const bool rowok = t * 16 + j < rm.M;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = an[q];
            if (PRO == PRO_DPRE) {
                v.x *= lrelu_grad_from_out(on[q].x); v.y *= lrelu_grad_from_out(on[q].y);
                v.z *= lrelu_grad_from_out(on[q].z); v.w *= lrelu_grad_from_out(on[q].w);
            }
            if (!rowok) v = f4zero();
            a[q] = v;
            cs[q] = f4add(cs[q], v);
        }
        // epilogue operands of this tile, then the next tile's A fragments: all in flight while the MFMAs run
        float4 rv[4], rv2[4];
        size_t orow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = min(t * 16 + kk * 4 + r, rm.M - 1);
            orow[r] = ((size_t)g * rm.rs_g + (size_t)m * rm.rs_m) * C + 4 * j;
            if (EPI == EPI_RES_LRELU || EPI == EPI_ADD_DPRE) rv[r] = ld4(resid + orow[r]);
            if (EPI == EPI_ADD_DPRE) rv2[r] = ld4(resid2 + orow[r]);
        }
        if (t + 1 < t1) fetch(t + 1);
        SB();
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
        SB();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (t * 16 + kk * 4 + r < rm.M) {
                float4 y = f4add(make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), bias4);
                if (EPI == EPI_RES_LRELU) {
                    y = f4add(y, rv[r]);
                    y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
                }
                if (EPI == EPI_LRELU) { y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w); }
                if (EPI == EPI_ADD_DPRE) {
                    const float4 d = rv[r], o = rv2[r];
                    y.x = fmaf(d.x, lrelu_grad_from_out(o.x), y.x); y.y = fmaf(d.y, lrelu_grad_from_out(o.y), y.y);
                    y.z = fmaf(d.z, lrelu_grad_from_out(o.z), y.z); y.w = fmaf(d.w, lrelu_grad_from_out(o.w), y.w);
                }
                st4(out + orow[r], y);
            }
        }
    }
    if (colsum != nullptr) {          // column sums of (pro-applied) A: 16-lane row reduction, 4 waves folded in LDS in fixed order, and
        // stored as this workgroup's PARTIAL [blockIdx.y][g][:] — no atomics: the consumer sums the gridDim.y row splits
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cs[q].x = group_sum<16>(cs[q].x); cs[q].y = group_sum<16>(cs[q].y);
            cs[q].z = group_sum<16>(cs[q].z); cs[q].w = group_sum<16>(cs[q].w);
            if (j == 0) st4(&csl[wave][16 * q + 4 * kk], cs[q]);
        }
        __syncthreads();
        if (wave == 0) colsum[((size_t)blockIdx.y * rm.G + g) * C + lane] = (csl[0][lane] + csl[1][lane]) + (csl[2][lane] + csl[3][lane]);
    }
}

// launch geometry of apply64: tiles per wave and row splits (gridDim.y = number of colsum partials per group)
static void apply64_geometry(const RowMap& rm, bool has_colsum, int& tpw, int& gy) {
    const int ntiles = (rm.M + 15) / 16;
    long tot = (long)rm.G * ntiles;
    tpw = (int)((tot + 4 * 512 - 1) / (4 * 512));              // ~512 workgroups of 4 waves: one round at 2 per CU ...
    if (tpw > 2) tpw = 2;                                      // ... but never more than 2 tiles per wave (measured: 3+ is 15-40 % slower)
    if (has_colsum && rm.G == 1) { const int lim = (ntiles + 4 * 128 - 1) / (4 * 128); if (tpw < lim) tpw = lim; }   // <= 128 partials
    if (g_apply_tpw > 0) tpw = g_apply_tpw;
    if (tpw < 1) tpw = 1;
    gy = (ntiles + 4 * tpw - 1) / (4 * tpw);
}

template <int PRO, int EPI>
static void launch_apply64_t(const float* A, const float* A2, const float* W, long gs, int transw, const float* bias, const float* resid,
                             const float* resid2, float* out, float* colsum, RowMap rm, hipStream_t st) {
    int tpw, gy;
    apply64_geometry(rm, colsum != nullptr, tpw, gy);
    hipLaunchKernelGGL((apply64_kernel<PRO, EPI>), dim3(rm.G, gy), dim3(256), 0, st, A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, tpw);
}

static int launch_apply64(const float* A, const float* A2, const float* W, long gs, int transw, const float* bias, const float* resid,
                          const float* resid2, float* out, float* colsum, RowMap rm, int pro, int epi, hipStream_t st) {
    if (pro == PRO_NONE && epi == EPI_PLAIN) launch_apply64_t<PRO_NONE, EPI_PLAIN>(A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, st);
    else if (pro == PRO_NONE && epi == EPI_RES_LRELU) launch_apply64_t<PRO_NONE, EPI_RES_LRELU>(A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, st);
    else if (pro == PRO_DPRE && epi == EPI_PLAIN) launch_apply64_t<PRO_DPRE, EPI_PLAIN>(A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, st);
    else if (pro == PRO_NONE && epi == EPI_ADD_DPRE) launch_apply64_t<PRO_NONE, EPI_ADD_DPRE>(A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, st);
    else if (pro == PRO_NONE && epi == EPI_LRELU) launch_apply64_t<PRO_NONE, EPI_LRELU>(A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, st);
    else return GPTST_EARG;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// C = 64 weight gradient on fp32 MFMA 16x16x4, operands straight from global memory as float4:
//   lane (j, kk) of k-step s loads row m = m0 + 4s + kk, channels 4j..4j+3 of A and of D: one instruction covers four whole
//   256-byte rows (the 32x32x2 version needed dword loads: 4x the load instructions, 128-byte pieces).  Component ca of the
//   A fragment and cb of the D fragment feed accumulator tile (ca, cb), whose MFMA index i / j stands for channel 4i+ca / 4j+cb:
//   the permutation is undone for free when the tile is written (a lane owns four CONSECUTIVE output columns of a row).
// The row range of a group is split over the 4 waves, folded through LDS, and stored as coalesced float4 rows.
template <int PRO, int U>
__global__ __launch_bounds__(256, 2) void wgrad64_kernel(const float* __restrict__ A, const float* __restrict__ D,
                                                         const float* __restrict__ D2, float* __restrict__ dW, RowMap rm,
                                                         int rows_per_split, int ostride, int csa) {
    __shared__ __attribute__((aligned(16))) float smem[WGRAD64_SMEM_FLOATS];
    wgrad64_body<PRO, U>(A, D, D2, dW, rm, rows_per_split, ostride, csa, blockIdx.x, blockIdx.y, smem);
}

// =====================================================================================================================
// Backward of one generated-weight layer in ONE pass (C = 64):   out = lrelu(S W_g + b_g [+ x])   with saved input S
//   dPre = dOut * lrelu'(out);   dS = dPre W_g^T   (data gradient);   dW_g = S^T dPre  (per row split);   db_g = colsum(dPre)
// replacing apply64<PRO_DPRE> + wgrad64<PRO_DPRE>, which both read dOut and out (26 us -> one launch).  A wave walks its 16-row tiles:
// dOut / out / S are loaded ONCE in the weight-gradient operand layout (lane (j,kk): rows 4s+kk, channels 4j..4j+3), dPre feeds the 64
// weight-gradient MFMAs from registers, then goes through a wave-private LDS tile to change to the data-gradient operand layout (lane
// (j,kk): row j, channels 16q+4kk..) for the 64 MFMAs against the register-resident W^T fragments.  The weight-gradient tiles of the 4
// waves fold through LDS at the end (the fold buffer aliases the weight staging area and the transposition tiles: 64.3 KB, 2 per CU).
// =====================================================================================================================
// KIND 0: generated-weight layer (above).  KIND 1: the shared Linear at the entry of `cap` (P = squash(X Wp^T + bp), GPTST.py:102) together
// with the residual branch of the layer:  dX = dY Wp + dOut*lrelu'(out),  dWp = dY^T X,  dbp = colsum(dY)  — here "dOut" carries dY (no
// activation), S = X, W = Wp ([out][in], used untransposed), and resid / resid2 = the layer's output gradient and output.
// CHAIN (the "dPre chain" of include/gptst_hip.h): 0 = the incoming gradient is dOut and the layer's output Y (KIND 1: resid2) gives the sign;
// 1 = the incoming gradient already is dPre (Y / the layer output are never read);  2 = as 1, and the result is multiplied by lrelu'(S)
// (S = the layer's input, the output of the LeakyReLU layer below — KIND 1 reads it from resid2 = S in the accumulator layout: an L1/L2 hit).
template <int KIND, int CHAIN>
__global__ __launch_bounds__(256, 2) void applywg64_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                            const float* __restrict__ S, const float* __restrict__ W, long w_gstride,
                                                            const float* __restrict__ resid, const float* __restrict__ resid2,
                                                            float* __restrict__ dS, float* __restrict__ dW, float* __restrict__ colsum,
                                                            RowMap rm, int tiles_per_wave) {
    constexpr int C = 64, TP = C + 4;
    __shared__ __attribute__((aligned(16))) float smem[4 * C * C];            // fold [4][C*C]; first: Wl [C*C] | 4 tiles [16][TP]
    __shared__ __attribute__((aligned(16))) float csl[4][C];
    float* Wl = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tile = smem + C * C + wave * 16 * TP;
    const int j = lane & 15, kk = lane >> 4;
    const int g = blockIdx.x;
    const int ntiles = (rm.M + 15) / 16;
    const int t0 = (blockIdx.y * 4 + wave) * tiles_per_wave, t1 = min(ntiles, t0 + tiles_per_wave);
    float4 d[4], y[4], a[4];
    auto fetch = [&](int t) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int m = min(t * 16 + 4 * s4 + kk, rm.M - 1);
            const size_t off = ((size_t)g * rm.rs_g + (size_t)m * rm.rs_m) * C + 4 * j;
            d[s4] = ld4(dOut + off); a[s4] = ld4(S + off);
            if (KIND == 0 && CHAIN == 0) y[s4] = ld4(Y + off);
        }
    };
    if (t0 < t1) fetch(t0);                          // in flight while the weight is staged
    load_w_lds<C, 256>(Wl, W + (size_t)g * w_gstride, KIND == 0 ? 1 : 0, threadIdx.x);     // KIND 0: W_g^T (dS = dPre W_g^T); 1: Wp as stored
    __syncthreads();
    float4 bv[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[q][e] = ld4(Wl + (16 * q + 4 * kk + e) * C + 4 * j);
    f32x4 accw[4][4];
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) accw[ca][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 cs = f4zero();
    for (int t = t0; t < t1; ++t) {
        if (KIND == 1 && t != t0) fetch(t);          // (KIND 1 keeps no prefetch: its epilogue operands need the registers)
        SB();
        // ---- dPre in the weight-gradient layout; rows beyond M contribute nothing ----
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float4 v = d[s4];
            if (KIND == 0 && CHAIN == 0) v = make_float4(d[s4].x * lrelu_grad_from_out(y[s4].x), d[s4].y * lrelu_grad_from_out(y[s4].y),
                                                         d[s4].z * lrelu_grad_from_out(y[s4].z), d[s4].w * lrelu_grad_from_out(y[s4].w));
            if (t * 16 + 4 * s4 + kk >= rm.M) v = f4zero();
            d[s4] = v;
            cs = f4add(cs, v);
            st4(tile + (4 * s4 + kk) * TP + 4 * j, v);
        }
        // ---- dW += S^T dPre: component ca of S / cb of dPre feed accumulator tile (ca, cb) (as wgrad64_kernel) ----
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            // KIND 0: dW = S^T dPre ([in][out]);  KIND 1: dWp = dY^T X ([out][in]) — the roles of the two operands swap
            const float sv[4] = {a[s4].x, a[s4].y, a[s4].z, a[s4].w}, dv[4] = {d[s4].x, d[s4].y, d[s4].z, d[s4].w};
#pragma unroll
            for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    accw[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(KIND == 0 ? sv[ca] : dv[ca], KIND == 0 ? dv[cb] : sv[cb], accw[ca][cb], 0, 0, 0);
        }
        SB();
        // ---- dPre tile back in the data-gradient operand layout (wave-private tile: no barrier) ----
        float4 ap[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ap[q] = ld4(tile + j * TP + 16 * q + 4 * kk);
        const int tcur = t;
        if (KIND == 0 && t + 1 < t1) fetch(t + 1);   // next tile's operands: in flight during the 64 MFMAs below
        float4 rv[4], rv2[4];
        if (KIND == 1 || CHAIN == 2) {               // residual branch operands / sign operand of the epilogue, in the D layout
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = min(tcur * 16 + kk * 4 + r, rm.M - 1);
                const size_t off = ((size_t)g * rm.rs_g + (size_t)m * rm.rs_m) * C + 4 * j;
                if (KIND == 1) rv[r] = ld4(resid + off);
                if (KIND == 1 && CHAIN != 1) rv2[r] = ld4(resid2 + off);
                if (KIND == 0 && CHAIN == 2) rv2[r] = ld4(S + off);
            }
        }
        SB();
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float av[4] = {ap[q].x, ap[q].y, ap[q].z, ap[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].w, acc[3], 0, 0, 0);
            }
        }
        SB();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = tcur * 16 + kk * 4 + r;
            float4 o4 = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
            if (KIND == 1 && CHAIN == 0) {
                o4.x = fmaf(rv[r].x, lrelu_grad_from_out(rv2[r].x), o4.x); o4.y = fmaf(rv[r].y, lrelu_grad_from_out(rv2[r].y), o4.y);
                o4.z = fmaf(rv[r].z, lrelu_grad_from_out(rv2[r].z), o4.z); o4.w = fmaf(rv[r].w, lrelu_grad_from_out(rv2[r].w), o4.w);
            }
            if (KIND == 1 && CHAIN != 0) o4 = f4add(o4, rv[r]);
            if (CHAIN == 2) {
                o4.x *= lrelu_grad_from_out(rv2[r].x); o4.y *= lrelu_grad_from_out(rv2[r].y);
                o4.z *= lrelu_grad_from_out(rv2[r].z); o4.w *= lrelu_grad_from_out(rv2[r].w);
            }
            if (m < rm.M) st4(dS + ((size_t)g * rm.rs_g + (size_t)m * rm.rs_m) * C + 4 * j, o4);
        }
    }
    // ---- column sums of dPre (bias gradient partial of this row split) and the weight-gradient fold ----
    cs.x += __shfl_xor(cs.x, 16, 64); cs.y += __shfl_xor(cs.y, 16, 64); cs.z += __shfl_xor(cs.z, 16, 64); cs.w += __shfl_xor(cs.w, 16, 64);
    cs.x += __shfl_xor(cs.x, 32, 64); cs.y += __shfl_xor(cs.y, 32, 64); cs.z += __shfl_xor(cs.z, 32, 64); cs.w += __shfl_xor(cs.w, 32, 64);
    if (kk == 0) st4(&csl[wave][4 * j], cs);
    __syncthreads();                                 // every wave is done with Wl and its tile: smem becomes the fold buffer
    float* red = smem + wave * C * C;
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            st4(&red[(4 * (kk * 4 + r) + ca) * C + 4 * j], make_float4(accw[ca][0][r], accw[ca][1][r], accw[ca][2][r], accw[ca][3][r]));
    __syncthreads();
    if (colsum != nullptr && threadIdx.x < C)
        colsum[((size_t)blockIdx.y * rm.G + g) * C + threadIdx.x] = (csl[0][threadIdx.x] + csl[1][threadIdx.x]) + (csl[2][threadIdx.x] + csl[3][threadIdx.x]);
    float* o = dW + ((size_t)blockIdx.y * rm.G + g) * (size_t)(C * C);
#pragma unroll
    for (int k = 0; k < C * C / 4 / 256; ++k) {
        const int f = threadIdx.x + k * 256;
        const float4 s4 = f4add(f4add(ld4(smem + 4 * f), ld4(smem + C * C + 4 * f)), f4add(ld4(smem + 2 * C * C + 4 * f), ld4(smem + 3 * C * C + 4 * f)));
        st4(o + 4 * f, s4);
    }
}

// row splits (= weight-gradient and bias-gradient partials per group) of gptst_apply_wgrad at this shape
extern "C" int gptst_apply_wgrad_nsplit(int mode, int BT, int N) {
    RowMap rm = make_rowmap(mode, BT, N);
    int tpw, gy;
    apply64_geometry(rm, false, tpw, gy);
    return gy;
}

// dS (rows, C), dW (nsplit*G, C, C), colsum (nsplit*G, C) or NULL; W (G, C, C) row-major [in][out] as in the forward.  C = 64.
// Y == NULL: dOut already is dPre;  premul: dS is multiplied by lrelu'(S) (dPre-chain convention, include/gptst_hip.h) — not with Y.
extern "C" int gptst_apply_wgrad(const float* dOut, const float* Y, const float* S, const float* W, float* dS, float* dW, float* colsum,
                                 int premul, int mode, int BT, int N, int C, void* stream) {
    if (!dOut || !S || !W || !dS || !dW || BT <= 0 || N <= 0 || mode < 0 || mode > 1 || (Y && premul)) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    RowMap rm = make_rowmap(mode, BT, N);
    int tpw, gy;
    apply64_geometry(rm, false, tpw, gy);
    const dim3 grid(rm.G, gy);
    hipStream_t st = (hipStream_t)stream;
    if (Y) hipLaunchKernelGGL((applywg64_kernel<0, 0>), grid, dim3(256), 0, st, dOut, Y, S, W, (long)C * C, nullptr, nullptr, dS, dW, colsum, rm, tpw);
    else if (!premul) hipLaunchKernelGGL((applywg64_kernel<0, 1>), grid, dim3(256), 0, st, dOut, Y, S, W, (long)C * C, nullptr, nullptr, dS, dW, colsum, rm, tpw);
    else hipLaunchKernelGGL((applywg64_kernel<0, 2>), grid, dim3(256), 0, st, dOut, Y, S, W, (long)C * C, nullptr, nullptr, dS, dW, colsum, rm, tpw);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// Backward through the shared Linear at the entry of cap plus the residual branch of the layer (GPTST.py:102,139-141), one pass:
//   dX = dY Wp + dOut*lrelu'(out);  per row split s < gptst_linear_bwd_nsplit: dWp[s] = dY^T X ([out][in]), dbp[s] = colsum(dY).
//   out == NULL: dOut already is dPre ->  dX = dY Wp + dPre, and with premul  dX = (dY Wp + dPre) * lrelu'(X).
// Replaces gptst_apply(mode 2, epi 2) + gptst_wgrad_colsum(mode 2).  rows = BT*N.  C = 64.
extern "C" int gptst_linear_bwd_nsplit(int rows) {
    RowMap rm = make_rowmap(2, rows, 1);
    int tpw, gy;
    apply64_geometry(rm, false, tpw, gy);
    return gy;
}

extern "C" int gptst_linear_bwd(const float* dY, const float* X, const float* Wp, const float* dOut, const float* out, float* dX, float* dWp,
                                float* dbp, int premul, int rows, int C, void* stream) {
    if (!dY || !X || !Wp || !dOut || !dX || !dWp || !dbp || rows <= 0 || (out && premul)) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    RowMap rm = make_rowmap(2, rows, 1);
    int tpw, gy;
    apply64_geometry(rm, false, tpw, gy);
    hipStream_t st = (hipStream_t)stream;
    if (out) hipLaunchKernelGGL((applywg64_kernel<1, 0>), dim3(1, gy), dim3(256), 0, st, dY, nullptr, X, Wp, 0L, dOut, out, dX, dWp, dbp, rm, tpw);
    else if (!premul) hipLaunchKernelGGL((applywg64_kernel<1, 1>), dim3(1, gy), dim3(256), 0, st, dY, nullptr, X, Wp, 0L, dOut, nullptr, dX, dWp, dbp, rm, tpw);
    else hipLaunchKernelGGL((applywg64_kernel<1, 2>), dim3(1, gy), dim3(256), 0, st, dY, nullptr, X, Wp, 0L, dOut, X, dX, dWp, dbp, rm, tpw);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// dW[g] (C x C, [i][o]) = sum_m A[row(g,m)][i] * D[row(g,m)][o];  both operands come straight from global memory:
// for MFMA step s the half-wave h reads row m = 2s+h, 32 consecutive floats (128 B) of each operand.
template <int C, int PRO>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ A, const float* __restrict__ D,
                                                    const float* __restrict__ D2, float* __restrict__ dW,
                                                    RowMap rm, int rows_per_split) {
    constexpr int NCT = C / 32;
    constexpr int TPW = NCT * NCT / 4;                 // output tiles per wave (C=64: 1, C=128: 4)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int g = blockIdx.x, sp = blockIdx.y;
    const int mbeg = sp * rows_per_split;
    const int mend = min(rm.M, mbeg + rows_per_split);
    const int it = (wave * TPW) / NCT;                 // A column tile (input channel block) of this wave
    f32x16 acc[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    for (int m = mbeg + h; m < mend + h; m += 2) {     // both halves run the same trip count
        float a = 0.f;
        float d[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) d[u] = 0.f;
        if (m < mend) {
            const size_t off = ((size_t)g * rm.rs_g + (size_t)m * rm.rs_m) * C;
            a = A[off + it * 32 + j];
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                const int jt = (wave * TPW + u) % NCT;
                float dv = D[off + jt * 32 + j];
                if (PRO == PRO_DPRE) dv *= lrelu_grad_from_out(D2[off + jt * 32 + j]);
                d[u] = dv;
            }
        }
#pragma unroll
        for (int u = 0; u < TPW; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d[u], acc[u], 0, 0, 0);
    }
    float* o = dW + ((size_t)sp * rm.G + g) * C * C;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int jt = (wave * TPW + u) % NCT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            o[(size_t)(it * 32 + row) * C + jt * 32 + j] = acc[u][r];
        }
    }
}

// C = 128 apply, second generation (apply_kernel<128> above keeps the weight AND a 32-row A tile per wave in LDS: 132 KB, one workgroup of
// 4 waves per CU, i.e. one wave per SIMD with load, MFMA and store phases strictly serial — 25-42 % of the fp32 MFMA roof, which bounds
// this shape).  Here only the group's weight lives in LDS (64 KB, staged once per workgroup of 8 waves, two workgroups per CU = 4 waves per
// SIMD whose phases overlap); a wave walks 16-row tiles with the A fragments straight from global memory (row j, channels 16q+4kk..+3, as
// apply64) and reads the B fragments of k-row 16q+4kk+e as two float4 (channels 4j..+3 and 64+4j..+3) from the row-major LDS image —
// conflict-free for ds_read_b128 without padding (every 16-lane service group covers the 16 slots of a 256-byte bank row).  Accumulator
// tile ct, column j stands for channel 4j+ct (ct < 4) / 64+4j+ct-4, so a lane owns two float4 of output rows 4kk+r and the epilogue
// (bias / residual / LReLU / dPre) runs from registers with 256-byte coalesced loads and stores.
// Measured at N = 4096, B = 32 (TIME mode, 51.5 GFLOP, 2.4 GB): 587-608 us = 85 TFLOP/s; with the MFMAs compiled out 454-493 us (the memory
// path alone: 4.9 TB/s), with the global traffic compiled out ~400 us (matrix pipe alone at the 2.1 GHz the chip holds here) — the
// kernel sits 20 % above its memory floor.  Tried and dropped: distinct s_setprio levels per wave (no change), a 2-waves-per-SIMD
// variant that requests the next tile and the epilogue operands before the MFMA loop and double-buffers the B fragments (685-725 us).
#define AP128_NW 8
template <int PRO, int EPI, bool CS>
__global__ __launch_bounds__(64 * AP128_NW, 4) void apply128_kernel(const float* __restrict__ A, const float* __restrict__ A2,
                                                                    const float* __restrict__ W, long w_gstride, int transw,
                                                                    const float* __restrict__ bias, const float* __restrict__ resid,
                                                                    const float* __restrict__ resid2, float* __restrict__ out,
                                                                    float* __restrict__ colsum, RowMap rm, int tiles_per_wave) {
    constexpr int C = 128, Q = C / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                   // [C][C]
    float* csl = smem + C * C;                          // [AP128_NW][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int g = blockIdx.x;
    const int ntiles = (rm.M + 15) / 16;
    const int t0 = (blockIdx.y * AP128_NW + wave) * tiles_per_wave, t1 = min(ntiles, t0 + tiles_per_wave);
    float4 bias0 = f4zero(), bias1 = f4zero();
    if (bias != nullptr) { bias0 = ld4(bias + (size_t)g * C + 4 * j); bias1 = ld4(bias + (size_t)g * C + 64 + 4 * j); }
    float4 a[Q], on[Q];
    auto fetch = [&](int t) {
        const int m = min(t * 16 + j, rm.M - 1);
        const size_t off = ((size_t)g * rm.rs_g + (size_t)m * rm.rs_m) * C + 4 * kk;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            a[q] = ld4(A + off + 16 * q);
            if (PRO == PRO_DPRE) on[q] = ld4(A2 + off + 16 * q);
        }
    };
    if (t0 < t1) fetch(t0);                              // in flight while the weight is staged
    load_w_lds<C, 64 * AP128_NW>(Wl, W + (size_t)g * w_gstride, transw, threadIdx.x);
    if (CS) { csl[wave * C + lane] = 0.f; csl[wave * C + 64 + lane] = 0.f; }    // column sums accumulate in the wave's own LDS row (no registers held across the MFMAs)
    __syncthreads();
    const float* wb = Wl + 4 * kk * C + 4 * j;
    for (int t = t0; t < t1; ++t) {
        const bool rowok = t * 16 + j < rm.M;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float4 v = a[q];
            if (PRO == PRO_DPRE) {
                v.x *= lrelu_grad_from_out(on[q].x); v.y *= lrelu_grad_from_out(on[q].y);
                v.z *= lrelu_grad_from_out(on[q].z); v.w *= lrelu_grad_from_out(on[q].w);
            }
            if (!rowok) v = f4zero();
            a[q] = v;
            if (CS) {                                  // column sums of the tile: 16-lane row reduction, lane j = 0 adds it to the wave's LDS row
                float4 c = make_float4(group_sum<16>(v.x), group_sum<16>(v.y), group_sum<16>(v.z), group_sum<16>(v.w));
                if (j == 0) { float* p = csl + wave * C + 16 * q + 4 * kk; st4(p, f4add(ld4(p), c)); }
            }
        }
        SB();
        f32x4 acc[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < Q; ++q) {                 // one k-block of 16 in batches of EB k-rows: B fragments, fence, MFMAs, fence (bounds the
            constexpr int EB = 4;                     // LDS prefetch depth: the kernel has to fit 128 VGPRs for 4 waves per SIMD)
            const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
            for (int e0 = 0; e0 < 4; e0 += EB) {
                float4 b0[EB], b1[EB];
#pragma unroll
                for (int e = 0; e < EB; ++e) { b0[e] = ld4(wb + (16 * q + e0 + e) * C); b1[e] = ld4(wb + (16 * q + e0 + e) * C + 64); }
                SB();
#pragma unroll
                for (int e = 0; e < EB; ++e) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e0 + e], b0[e].x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e0 + e], b0[e].y, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e0 + e], b0[e].z, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e0 + e], b0[e].w, acc[3], 0, 0, 0);
                    acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e0 + e], b1[e].x, acc[4], 0, 0, 0);
                    acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e0 + e], b1[e].y, acc[5], 0, 0, 0);
                    acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e0 + e], b1[e].z, acc[6], 0, 0, 0);
                    acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e0 + e], b1[e].w, acc[7], 0, 0, 0);
                }
                SB();
            }
        }
        // the epilogue operands and the next tile's A fragments are requested once the A registers are free
        float4 rv[4][2], rv2[4][2];
        size_t orow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = min(t * 16 + kk * 4 + r, rm.M - 1);
            orow[r] = ((size_t)g * rm.rs_g + (size_t)m * rm.rs_m) * C + 4 * j;
            if (EPI == EPI_RES_LRELU || EPI == EPI_ADD_DPRE || EPI == EPI_ADD_PREMUL) { rv[r][0] = ld4(resid + orow[r]); rv[r][1] = ld4(resid + orow[r] + 64); }
            if (EPI == EPI_ADD_DPRE || EPI == EPI_ADD_PREMUL || EPI == EPI_PREMUL) { rv2[r][0] = ld4(resid2 + orow[r]); rv2[r][1] = ld4(resid2 + orow[r] + 64); }
        }
        constexpr bool EARLY = (EPI == EPI_PLAIN || EPI == EPI_LRELU) && PRO == PRO_NONE;   // no residual / dPre operands: room for the next tile's fragments now
        if (EARLY && t + 1 < t1) fetch(t + 1);
        SB();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (t * 16 + kk * 4 + r < rm.M) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float4 y = f4add(make_float4(acc[4 * hf + 0][r], acc[4 * hf + 1][r], acc[4 * hf + 2][r], acc[4 * hf + 3][r]), hf ? bias1 : bias0);
                    if (EPI == EPI_RES_LRELU) {
                        y = f4add(y, rv[r][hf]);
                        y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
                    }
                    if (EPI == EPI_LRELU) { y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w); }
                    if (EPI == EPI_ADD_DPRE) {
                        const float4 d = rv[r][hf], o = rv2[r][hf];
                        y.x = fmaf(d.x, lrelu_grad_from_out(o.x), y.x); y.y = fmaf(d.y, lrelu_grad_from_out(o.y), y.y);
                        y.z = fmaf(d.z, lrelu_grad_from_out(o.z), y.z); y.w = fmaf(d.w, lrelu_grad_from_out(o.w), y.w);
                    }
                    if (EPI == EPI_ADD_PREMUL) {                     // dPre chain: (dY Wp + dPre) * lrelu'(X)
                        const float4 d = rv[r][hf], o = rv2[r][hf];
                        y.x = (y.x + d.x) * lrelu_grad_from_out(o.x); y.y = (y.y + d.y) * lrelu_grad_from_out(o.y);
                        y.z = (y.z + d.z) * lrelu_grad_from_out(o.z); y.w = (y.w + d.w) * lrelu_grad_from_out(o.w);
                    }
                    if (EPI == EPI_PREMUL) {                         // dPre chain, no residual branch: (dPre W^T) * lrelu'(X)
                        const float4 o = rv2[r][hf];
                        y.x *= lrelu_grad_from_out(o.x); y.y *= lrelu_grad_from_out(o.y); y.z *= lrelu_grad_from_out(o.z); y.w *= lrelu_grad_from_out(o.w);
                    }
                    st4(out + orow[r] + 64 * hf, y);
                }
            }
        }
        SB();
        if (!EARLY && t + 1 < t1) fetch(t + 1);
    }
    if (CS) {                         // fold the waves' column sums in fixed order: this workgroup's partial [blockIdx.y][g][:]
        __syncthreads();
        if (threadIdx.x < C) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < AP128_NW; ++w) s += csl[w * C + threadIdx.x];
            colsum[((size_t)blockIdx.y * rm.G + g) * C + threadIdx.x] = s;
        }
    }
}

static void apply128_geometry(const RowMap& rm, bool has_colsum, int& tpw, int& gy) {
    const int ntiles = (rm.M + 15) / 16;
    tpw = rm.G == 1 ? 8 : 4;                                     // shared weight: fewer re-stagings of W; else >= ~2 workgroups per CU and group
    if (has_colsum && rm.G == 1) { const int lim = (ntiles + AP128_NW * 128 - 1) / (AP128_NW * 128); if (tpw < lim) tpw = lim; }   // <= 128 partials
    if (tpw * AP128_NW > ntiles) tpw = (ntiles + AP128_NW - 1) / AP128_NW;
    // few, short groups (one rank's 512-node share of configs[4]: 384 groups x 32 tiles): tiles per wave halved until ~3 workgroups per CU exist
    // (r05: 8 -> 4 -> 2 keeps every wave of a workgroup busy; the 1024-workgroup rule of round 4 ended at one tile per wave, i.e. a 64 KB weight
    // staged per 8 tiles: 148 -> 152.5 steps/s on that share)
    while (tpw > 1 && !(has_colsum && rm.G == 1) && (long)rm.G * ((ntiles + AP128_NW * tpw - 1) / (AP128_NW * tpw)) < g_apply128_minwg) tpw = (tpw + 1) >> 1;
    if (g_apply_tpw > 0) tpw = g_apply_tpw;
    if (tpw < 1) tpw = 1;
    gy = (ntiles + AP128_NW * tpw - 1) / (AP128_NW * tpw);
}

template <int PRO, int EPI, bool CS>
static void launch_apply128_t(const float* A, const float* A2, const float* W, long gs, int transw, const float* bias, const float* resid,
                              const float* resid2, float* out, float* colsum, RowMap rm, hipStream_t st) {
    constexpr size_t smem = (size_t)(128 * 128 + AP128_NW * 128) * sizeof(float);
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)apply128_kernel<PRO, EPI, CS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr = true; }
    int tpw, gy;
    apply128_geometry(rm, CS, tpw, gy);
    hipLaunchKernelGGL((apply128_kernel<PRO, EPI, CS>), dim3(rm.G, gy), dim3(64 * AP128_NW), smem, st, A, A2, W, gs, transw, bias, resid, resid2, out,
                       colsum, rm, tpw);
}

static int launch_apply128(const float* A, const float* A2, const float* W, long gs, int transw, const float* bias, const float* resid,
                           const float* resid2, float* out, float* colsum, RowMap rm, int pro, int epi, hipStream_t st) {
#define AP128(P, E, S) launch_apply128_t<P, E, S>(A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, st)
    if (colsum != nullptr) {                            // column sums ride on the two variants whose callers want them (bias gradients)
        if (pro == PRO_DPRE && epi == EPI_PLAIN) AP128(PRO_DPRE, EPI_PLAIN, true);
        else if (pro == PRO_NONE && epi == EPI_ADD_DPRE) AP128(PRO_NONE, EPI_ADD_DPRE, true);
        else if (pro == PRO_NONE && epi == EPI_PLAIN) AP128(PRO_NONE, EPI_PLAIN, true);
        else return GPTST_ESHAPE;
    }
    else if (pro == PRO_NONE && epi == EPI_PLAIN) AP128(PRO_NONE, EPI_PLAIN, false);
    else if (pro == PRO_NONE && epi == EPI_RES_LRELU) AP128(PRO_NONE, EPI_RES_LRELU, false);
    else if (pro == PRO_DPRE && epi == EPI_PLAIN) AP128(PRO_DPRE, EPI_PLAIN, false);
    else if (pro == PRO_NONE && epi == EPI_ADD_DPRE) AP128(PRO_NONE, EPI_ADD_DPRE, false);
    else if (pro == PRO_NONE && epi == EPI_LRELU) AP128(PRO_NONE, EPI_LRELU, false);
    else if (pro == PRO_NONE && epi == EPI_ADD_PREMUL) AP128(PRO_NONE, EPI_ADD_PREMUL, false);
    else if (pro == PRO_NONE && epi == EPI_PREMUL) AP128(PRO_NONE, EPI_PREMUL, false);
    else return GPTST_EARG;
#undef AP128
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// C = 128 weight gradient, second generation (wgrad_kernel<128> above: dword operand loads, one k-step per trip — latency-bound at
// ~25 % of the fp32 MFMA roof, which is the roof of this shape: 2*128*128 flop per 2*128*4 operand bytes).  Same operand scheme as
// wgrad64_kernel (16x16x4, lane (j,kk): rows 4s+kk, four consecutive channels of A and of D as one float4 each, accumulator tile
// (ca, cb) <-> channels 4i+ca / 4j+cb), with the 128 x 128 output split into four 64 x 64 quadrants, one per wave: every wave walks
// ALL rows of the split (no fold through LDS), 16 MFMAs per pair of float4 loads, and the quadrant is stored straight from the
// accumulators as coalesced float4 rows.  Row splits (gridDim.y) give the partial sums the pool jobs fold.
template <int PRO, int U>
__global__ __launch_bounds__(256, 2) void wgrad128_kernel(const float* __restrict__ A, const float* __restrict__ D,
                                                          const float* __restrict__ D2, float* __restrict__ dW, RowMap rm,
                                                          int rows_per_split, int ostride, int csa) {
    constexpr int C = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4, wa = wave >> 1, wb = wave & 1;
    const int g = blockIdx.x, sp = blockIdx.y;
    const int mbeg = sp * rows_per_split, mend = min(rm.M, mbeg + rows_per_split);
    f32x4 acc[4][4];
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[ca][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 sa = f4zero();
    const float* Ab = A + (size_t)g * rm.rs_g * C + 64 * wa + 4 * j;
    const float* Db = D + (size_t)g * rm.rs_g * C + 64 * wb + 4 * j;
    const float* Yb = D2 + (size_t)g * rm.rs_g * C + 64 * wb + 4 * j;
    for (int m0 = mbeg; m0 < mend; m0 += 4 * U) {
        float4 a[U], d[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t off = (size_t)min(m0 + 4 * u + kk, mend - 1) * rm.rs_m * C;      // clamped: out-of-range rows are zeroed below
            a[u] = ld4(Ab + off); d[u] = ld4(Db + off);
            if (PRO == PRO_DPRE) y[u] = ld4(Yb + off);
        }
        SB();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = m0 + 4 * u + kk < mend;
            if (!ok) a[u] = f4zero();
            if (PRO == PRO_DPRE) {
                d[u].x *= lrelu_grad_from_out(y[u].x); d[u].y *= lrelu_grad_from_out(y[u].y);
                d[u].z *= lrelu_grad_from_out(y[u].z); d[u].w *= lrelu_grad_from_out(y[u].w);
            }
            if (csa == 2) { if (ok) sa = f4add(sa, d[u]); }
            else sa = f4add(sa, a[u]);
            const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, dv[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
#pragma unroll
            for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ca], dv[cb], acc[ca][cb], 0, 0, 0);
        }
    }
    // D reg r of tile (ca, cb): dW row 64wa + 4*(kk*4 + r) + ca, columns 64wb + 4j + cb
    float* o = dW + ((size_t)sp * rm.G + g) * (size_t)ostride;
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            st4(o + (size_t)(64 * wa + 4 * (kk * 4 + r) + ca) * C + 64 * wb + 4 * j,
                make_float4(acc[ca][0][r], acc[ca][1][r], acc[ca][2][r], acc[ca][3][r]));
    if (csa) {                                   // column sums: of A (which = 1) by the waves of column half wb = 0, of pro(D) by wa = 0
        sa.x += __shfl_xor(sa.x, 16, 64); sa.y += __shfl_xor(sa.y, 16, 64); sa.z += __shfl_xor(sa.z, 16, 64); sa.w += __shfl_xor(sa.w, 16, 64);
        sa.x += __shfl_xor(sa.x, 32, 64); sa.y += __shfl_xor(sa.y, 32, 64); sa.z += __shfl_xor(sa.z, 32, 64); sa.w += __shfl_xor(sa.w, 32, 64);
        if (kk == 0) {
            if (csa == 2 && wa == 0) st4(o + C * C + 64 * wb + 4 * j, sa);
            if (csa != 2 && wb == 0) st4(o + C * C + 64 * wa + 4 * j, sa);
        }
    }
}

static int apply_v1_gy(const RowMap& rm, bool has_colsum) {
    const int ntiles = (rm.M + 31) / 32;
    int gy = (ntiles + 3) / 4;
    if (rm.G == 1) gy = min(gy, has_colsum ? 128 : 1024);  // shared weight: with colsum keep the partial count low
    else gy = min(gy, max(2, (768 + rm.G - 1) / rm.G));      // >= ~768 workgroups when the groups are few and long (N = 4096: 96 groups x 128 tiles)
    return gy < 1 ? 1 : gy;
}

template <int C>
static int launch_apply(const float* A, const float* A2, const float* W, long w_gstride, int transw, const float* bias,
                        const float* resid, const float* resid2, float* out, float* colsum, RowMap rm, int pro, int epi, hipStream_t st) {
    const int gy = apply_v1_gy(rm, colsum != nullptr);
    dim3 grid(rm.G, gy), block(256);
    const size_t smem = (size_t)(C * C + 4 * Tile<C>::TILE_FLOATS) * sizeof(float);
#define LAUNCH(P, E)                                                                                              \
    hipLaunchKernelGGL((apply_kernel<C, P, E>), grid, block, smem, st, A, A2, W, w_gstride, transw, bias, resid, \
                       resid2, out, colsum, rm, g_ap_dbg)
    if (pro == PRO_NONE && epi == EPI_PLAIN) LAUNCH(PRO_NONE, EPI_PLAIN);
    else if (pro == PRO_NONE && epi == EPI_RES_LRELU) LAUNCH(PRO_NONE, EPI_RES_LRELU);
    else if (pro == PRO_DPRE && epi == EPI_PLAIN) LAUNCH(PRO_DPRE, EPI_PLAIN);
    else if (pro == PRO_NONE && epi == EPI_ADD_DPRE) LAUNCH(PRO_NONE, EPI_ADD_DPRE);
    else if (pro == PRO_NONE && epi == EPI_LRELU) LAUNCH(PRO_NONE, EPI_LRELU);
    else return GPTST_EARG;
#undef LAUNCH
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

static int g_smem_attr_done = 0;
thread_local int g_apply_v1 = 0;                                    // experiments: gptst_tune(3, 1) selects the LDS-staged apply_kernel for C = 64
template <int C>
static void raise_smem_limits() {
    const int smem = (int)((C * C + 4 * Tile<C>::TILE_FLOATS) * sizeof(float));
    hipFuncSetAttribute((const void*)apply_kernel<C, PRO_NONE, EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute((const void*)apply_kernel<C, PRO_NONE, EPI_RES_LRELU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute((const void*)apply_kernel<C, PRO_DPRE, EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute((const void*)apply_kernel<C, PRO_NONE, EPI_ADD_DPRE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute((const void*)apply_kernel<C, PRO_NONE, EPI_LRELU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
}

// number of row-split partials gptst_apply writes per group into `colsum` (shape [nsplit][G][C]) at this shape
extern "C" int gptst_apply_nsplit(int mode, int BT, int N, int C) {
    RowMap rm = make_rowmap(mode, BT, N);
    if (C == 64 && !g_apply_v1) { int tpw, gy; apply64_geometry(rm, true, tpw, gy); return gy; }
    if (C == 128 && !g_apply128_v1) { int tpw, gy; apply128_geometry(rm, true, tpw, gy); return gy; }
    return apply_v1_gy(rm, true);
}

extern "C" int gptst_apply(const float* A, const float* A2, const float* W, int w_per_group, int transw,
                           const float* bias, const float* resid, const float* resid2, float* out, float* colsum,
                           int mode, int pro, int epi, int BT, int N, int C, void* stream) {
    if (!A || !W || !out || BT <= 0 || N <= 0) return GPTST_EARG;
    if (pro == PRO_DPRE && !A2) return GPTST_EARG;
    if (epi == EPI_RES_LRELU && !resid) return GPTST_EARG;
    if ((epi == EPI_ADD_DPRE || epi == EPI_ADD_PREMUL) && (!resid || !resid2)) return GPTST_EARG;
    if (epi == EPI_PREMUL && !resid2) return GPTST_EARG;
    if ((epi == EPI_ADD_PREMUL || epi == EPI_PREMUL) && (C != 128 || g_apply128_v1)) return GPTST_ESHAPE;      // (C = 64 has the fused gptst_linear_bwd for this)
    if (!g_smem_attr_done) { raise_smem_limits<64>(); raise_smem_limits<128>(); g_smem_attr_done = 1; }
    RowMap rm = make_rowmap(mode, BT, N);
    const long gs = w_per_group ? (long)C * C : 0;
    hipStream_t st = (hipStream_t)stream;
    if (C == 64 && !g_apply_v1) return launch_apply64(A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, pro, epi, st);
    if (C == 64) return launch_apply<64>(A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, pro, epi, st);
    if (C == 128 && !g_apply128_v1) return launch_apply128(A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, pro, epi, st);
    if (C == 128) return launch_apply<128>(A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, pro, epi, st);
    return GPTST_ESHAPE;
}

// dW has room for nsplit * G matrices; returns nsplit through *nsplit_out (consumers sum the splits).
thread_local int g_wgrad_ns0_override = 0;
thread_local int g_wgrad_ns_override = 0;                           // experiments: gptst_tune(2, ns) forces the NODE-mode split
thread_local int g_wgrad_v1 = 0;                                    // experiments: gptst_tune(7, 1) selects the first-generation wgrad_kernel for C = 128
extern "C" int gptst_wgrad_nsplit(int mode, int BT, int N, int C) {
    RowMap rm = make_rowmap(mode, BT, N);
    if (mode == 2 && C == 128 && !g_wgrad_v1) {        // shared weight, 64 KB partials: <= 1024 of them (N = 4096: 1536-row chunks, 67 MB instead of 400 MB)
        const int ns = (rm.M + 255) / 256;
        return ns < 1024 ? ns : 1024;
    }
    if (mode == 2) return (rm.M + 255) / 256;          // shared weight: 256-row chunks
    if (mode == 1 && g_wgrad_ns_override > 0) return g_wgrad_ns_override;
    if (mode == 0 && g_wgrad_ns0_override > 0) return g_wgrad_ns0_override;
    if (C == 128 && !g_wgrad_v1) {                     // MFMA-bound: ~3 equal workgroups per CU, splits of >= 128 rows; every split is a 64 KB
        // partial per group that the reduction jobs read back (384 groups: 2 splits 126.7, 3 splits 123.6 steps/s at N = 512; equal at N = 4096)
        // r05: rounded DOWN — 512 node groups run one split each (two: 150.8, one: 152.5 steps/s on the N = 512 share; half the partial bytes)
        int want = 768 / rm.G, maxs = (rm.M + 127) / 128;
        if (want > maxs) want = maxs;
        return want < 1 ? 1 : want;
    }
    if (rm.G >= 256) return 1;
    int want = 512 / rm.G;                             // largest split that still fits ONE round of 2 workgroups per CU
    if (want < 1) want = 1;                            // (G = 170: 3 x 170 = 510 workgroups 12.8 us; 4 x 170 = 680 -> 16.4 us)
    int maxs = (rm.M + 63) / 64;
    return want < maxs ? want : maxs;
}

static int wgrad_impl(const float* A, const float* D, const float* D2, float* dW, int mode, int pro, int BT, int N, int C, int csa,
                      void* stream);

extern "C" int gptst_wgrad(const float* A, const float* D, const float* D2, float* dW, int mode, int pro, int BT, int N,
                           int C, void* stream) {
    return wgrad_impl(A, D, D2, dW, mode, pro, BT, N, C, 0, stream);
}

// as gptst_wgrad, plus column sums appended to every split: dW rows are C*C + C floats, [dW | sum_m A[m,:]] (which = 1: the bias
// gradient of a Linear whose OUTPUT gradient is A) or [dW | sum_m pro(D)[m,:]] (which = 2: the bias gradient next to a weight
// gradient dW = A^T pro(D), e.g. hyperTem's b_bt).  C = 64.
extern "C" int gptst_wgrad_colsum(const float* A, const float* D, const float* D2, float* dW, int mode, int pro, int which, int BT, int N,
                                  int C, void* stream) {
    if (C != 64 && C != 128) return GPTST_ESHAPE;
    if (which != 1 && which != 2) return GPTST_EARG;
    return wgrad_impl(A, D, D2, dW, mode, pro, BT, N, C, which, stream);
}

static int wgrad_impl(const float* A, const float* D, const float* D2, float* dW, int mode, int pro, int BT, int N, int C, int csa,
                      void* stream) {
    if (!A || !D || !dW) return GPTST_EARG;
    if (pro == PRO_DPRE && !D2) return GPTST_EARG;
    RowMap rm = make_rowmap(mode, BT, N);
    const int ns = gptst_wgrad_nsplit(mode, BT, N, C);
    int rps = (rm.M + ns - 1) / ns;
    rps = (rps + 1) & ~1;                               // even, so a k-step never straddles a split
    dim3 grid(rm.G, ns), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (C == 64) {
        const int rows = rps < rm.M ? rps : rm.M, steps = ((rows + 3) / 4 + 3) / 4;      // k-steps per wave
        const bool u6 = (steps + 5) / 6 * 6 <= (steps + 3) / 4 * 4;
        const int os = csa ? C * C + C : C * C;
#define WG64(P, UU) hipLaunchKernelGGL((wgrad64_kernel<P, UU>), grid, block, 0, st, A, D, D2, dW, rm, rps, os, csa)
        if (pro == PRO_DPRE) { if (u6) WG64(PRO_DPRE, 6); else WG64(PRO_DPRE, 4); }
        else { if (u6) WG64(PRO_NONE, 6); else WG64(PRO_NONE, 4); }
#undef WG64
    } else if (C == 128 && !g_wgrad_v1) {
        rps = (rps + 3) & ~3;                           // whole k-steps of 4 rows per split (a trailing split may be empty: it stores zeros)
        const int os = csa ? C * C + C : C * C;
        if (pro == PRO_DPRE) hipLaunchKernelGGL((wgrad128_kernel<PRO_DPRE, 4>), grid, block, 0, st, A, D, D2, dW, rm, rps, os, csa);
        else hipLaunchKernelGGL((wgrad128_kernel<PRO_NONE, 4>), grid, block, 0, st, A, D, D2 ? D2 : D, dW, rm, rps, os, csa);
    } else if (C == 128) {
        if (csa) return GPTST_ESHAPE;
        if (pro == PRO_DPRE) hipLaunchKernelGGL((wgrad_kernel<128, PRO_DPRE>), grid, block, 0, st, A, D, D2, dW, rm, rps);
        else hipLaunchKernelGGL((wgrad_kernel<128, PRO_NONE>), grid, block, 0, st, A, D, D2, dW, rm, rps);
    } else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
