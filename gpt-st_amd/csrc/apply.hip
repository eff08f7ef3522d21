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
__device__ long long g_ap_ts[64];     // debug: per-phase s_memtime stamps (enabled by gptst_ap_dbg(1))
int g_ap_dbg = 0;
extern "C" int gptst_ap_dbg(int v) { g_ap_dbg = v; return 0; }
extern "C" int gptst_ap_ts(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ap_ts), sizeof(long long) * 64); }

enum { PRO_NONE = 0, PRO_DPRE = 1 };          // PRO_DPRE: a = A * lrelu'(A2)   (A = dOut, A2 = layer output)
enum { EPI_PLAIN = 0, EPI_RES_LRELU = 1, EPI_ADD_DPRE = 2, EPI_LRELU = 3 };   // 1: lrelu(acc+bias+resid)  2: acc + resid*lrelu'(resid2)  3: lrelu(acc+bias)

struct RowMap {
    int G, M;
    long rs_g, rs_m;
};

__host__ __device__ inline RowMap make_rowmap(int mode, int BT, int N) {
    RowMap r;
    if (mode == 0) { r.G = BT; r.M = N; r.rs_g = N; r.rs_m = 1; }
    else if (mode == 1) { r.G = N; r.M = BT; r.rs_g = 1; r.rs_m = N; }
    else { r.G = 1; r.M = BT * N; r.rs_g = 0; r.rs_m = 1; }
    return r;
}

template <int C, int PRO, int EPI>
__global__ __launch_bounds__(256) void apply_kernel(const float* __restrict__ A, const float* __restrict__ A2,
                                                    const float* __restrict__ W, long w_gstride, int transw,
                                                    const float* __restrict__ bias, const float* __restrict__ resid,
                                                    const float* __restrict__ resid2, float* __restrict__ out, float* __restrict__ colsum, RowMap rm, int dbg) {
    using T = Tile<C>;
    int tsi = 0;
#define TS() do { if (dbg && blockIdx.x == 100 && blockIdx.y == 0 && threadIdx.x == 0) g_ap_ts[tsi] = clock64(); ++tsi; } while (0)
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
    load_w_lds<C>(Wl, W + (size_t)g * w_gstride, transw, tid, 256);
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
    if (colsum != nullptr) {          // fold the 4 waves in LDS first: one atomic per column and workgroup
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
                atomicAdd(colsum + (size_t)g * C + u * 64 + lane, s);
            }
        }
    }
}

// C = 64 weight gradient with the row range split over the 4 waves: every wave accumulates all four 32x32 output tiles
// over its quarter of the rows (4 operand loads per 4 MFMAs, k-loop unrolled so ~16 loads are in flight), then the waves
// are folded through LDS and the 64x64 result is stored with coalesced float4 rows.
template <int PRO>
__global__ __launch_bounds__(256) void wgrad64_kernel(const float* __restrict__ A, const float* __restrict__ D,
                                                      const float* __restrict__ D2, float* __restrict__ dW, RowMap rm,
                                                      int rows_per_split, int ostride, int csa) {
    constexpr int C = 64;
    __shared__ float red[4][C * C];
    __shared__ float csred[4][C];
    float sa0 = 0.f, sa1 = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int g = blockIdx.x, sp = blockIdx.y;
    const int mbeg0 = sp * rows_per_split;
    const int mend0 = min(rm.M, mbeg0 + rows_per_split);
    int q = (mend0 - mbeg0 + 3) / 4;
    q = (q + 1) & ~1;
    const int mbeg = mbeg0 + wave * q, mend = min(mend0, mbeg + q);
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    constexpr int U = 8;                               // 8 k-steps = 48 operand loads in flight per wave (memory-latency bound otherwise)
    for (int m0 = mbeg; m0 < mend; m0 += 2 * U) {
        float a0[U], a1[U], d0[U], d1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + 2 * u + h;
            a0[u] = a1[u] = d0[u] = d1[u] = 0.f;
            if (m < mend) {
                const size_t off = ((size_t)g * rm.rs_g + (size_t)m * rm.rs_m) * C;
                a0[u] = A[off + j]; a1[u] = A[off + 32 + j];
                d0[u] = D[off + j]; d1[u] = D[off + 32 + j];
                if (PRO == PRO_DPRE) { d0[u] *= lrelu_grad_from_out(D2[off + j]); d1[u] *= lrelu_grad_from_out(D2[off + 32 + j]); }
            }
            sa0 += a0[u]; sa1 += a1[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], d0[u], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], d1[u], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], d0[u], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], d1[u], acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                red[wave][(a * 32 + row) * C + b * 32 + j] = acc[a][b][r];
            }
    if (csa) {                                   // column sums of A (bias gradient of a shared Linear), folded over halves and waves
        sa0 += __shfl_xor(sa0, 32, 64); sa1 += __shfl_xor(sa1, 32, 64);
        if (h == 0) { csred[wave][j] = sa0; csred[wave][32 + j] = sa1; }
    }
    __syncthreads();
    float* o = dW + ((size_t)sp * rm.G + g) * (size_t)ostride;
    if (csa && threadIdx.x < C) o[C * C + threadIdx.x] = csred[0][threadIdx.x] + csred[1][threadIdx.x] + csred[2][threadIdx.x] + csred[3][threadIdx.x];
    for (int f = threadIdx.x; f < C * C / 4; f += 256) {
        const float4 s = f4add(f4add(ld4(&red[0][4 * f]), ld4(&red[1][4 * f])), f4add(ld4(&red[2][4 * f]), ld4(&red[3][4 * f])));
        st4(o + 4 * f, s);
    }
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

template <int C>
static int launch_apply(const float* A, const float* A2, const float* W, long w_gstride, int transw, const float* bias,
                        const float* resid, const float* resid2, float* out, float* colsum, RowMap rm, int pro, int epi, hipStream_t st) {
    const int ntiles = (rm.M + 31) / 32;
    int gy = (ntiles + 3) / 4;
    if (rm.G == 1) gy = min(gy, colsum ? 128 : 1024);  // shared weight: with colsum keep the atomics per address low
    else gy = min(gy, 2);
    if (gy < 1) gy = 1;
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
template <int C>
static void raise_smem_limits() {
    const int smem = (int)((C * C + 4 * Tile<C>::TILE_FLOATS) * sizeof(float));
    hipFuncSetAttribute((const void*)apply_kernel<C, PRO_NONE, EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute((const void*)apply_kernel<C, PRO_NONE, EPI_RES_LRELU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute((const void*)apply_kernel<C, PRO_DPRE, EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute((const void*)apply_kernel<C, PRO_NONE, EPI_ADD_DPRE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute((const void*)apply_kernel<C, PRO_NONE, EPI_LRELU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
}

extern "C" int gptst_apply(const float* A, const float* A2, const float* W, int w_per_group, int transw,
                           const float* bias, const float* resid, const float* resid2, float* out, float* colsum,
                           int mode, int pro, int epi, int BT, int N, int C, void* stream) {
    if (!A || !W || !out || BT <= 0 || N <= 0) return GPTST_EARG;
    if (pro == PRO_DPRE && !A2) return GPTST_EARG;
    if (epi == EPI_RES_LRELU && !resid) return GPTST_EARG;
    if (epi == EPI_ADD_DPRE && (!resid || !resid2)) return GPTST_EARG;
    if (!g_smem_attr_done) { raise_smem_limits<64>(); raise_smem_limits<128>(); g_smem_attr_done = 1; }
    RowMap rm = make_rowmap(mode, BT, N);
    const long gs = w_per_group ? (long)C * C : 0;
    hipStream_t st = (hipStream_t)stream;
    if (C == 64) return launch_apply<64>(A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, pro, epi, st);
    if (C == 128) return launch_apply<128>(A, A2, W, gs, transw, bias, resid, resid2, out, colsum, rm, pro, epi, st);
    return GPTST_ESHAPE;
}

// dW has room for nsplit * G matrices; returns nsplit through *nsplit_out (consumers sum the splits).
extern "C" int gptst_wgrad_nsplit(int mode, int BT, int N) {
    RowMap rm = make_rowmap(mode, BT, N);
    if (mode == 2) return (rm.M + 255) / 256;          // shared weight: 256-row chunks
    if (rm.G >= 256) return 1;
    int want = (512 + rm.G - 1) / rm.G;                // aim for >= 512 workgroups
    int maxs = (rm.M + 63) / 64;
    return want < maxs ? want : maxs;
}

static int wgrad_impl(const float* A, const float* D, const float* D2, float* dW, int mode, int pro, int BT, int N, int C, int csa,
                      void* stream);

extern "C" int gptst_wgrad(const float* A, const float* D, const float* D2, float* dW, int mode, int pro, int BT, int N,
                           int C, void* stream) {
    return wgrad_impl(A, D, D2, dW, mode, pro, BT, N, C, 0, stream);
}

// as gptst_wgrad, plus the column sums of A appended to every split: dW rows are C*C + C floats ([dW | sum_m A[m,:]])
extern "C" int gptst_wgrad_colsum(const float* A, const float* D, const float* D2, float* dW, int mode, int pro, int BT, int N,
                                  int C, void* stream) {
    if (C != 64) return GPTST_ESHAPE;
    return wgrad_impl(A, D, D2, dW, mode, pro, BT, N, C, 1, stream);
}

static int wgrad_impl(const float* A, const float* D, const float* D2, float* dW, int mode, int pro, int BT, int N, int C, int csa,
                      void* stream) {
    if (!A || !D || !dW) return GPTST_EARG;
    if (pro == PRO_DPRE && !D2) return GPTST_EARG;
    RowMap rm = make_rowmap(mode, BT, N);
    const int ns = gptst_wgrad_nsplit(mode, BT, N);
    int rps = (rm.M + ns - 1) / ns;
    rps = (rps + 1) & ~1;                               // even, so a k-step never straddles a split
    dim3 grid(rm.G, ns), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (C == 64) {
        if (pro == PRO_DPRE) hipLaunchKernelGGL((wgrad64_kernel<PRO_DPRE>), grid, block, 0, st, A, D, D2, dW, rm, rps, csa ? C * C + C : C * C, csa);
        else hipLaunchKernelGGL((wgrad64_kernel<PRO_NONE>), grid, block, 0, st, A, D, D2, dW, rm, rps, csa ? C * C + C : C * C, csa);
    } else if (C == 128) {
        if (pro == PRO_DPRE) hipLaunchKernelGGL((wgrad_kernel<128, PRO_DPRE>), grid, block, 0, st, A, D, D2, dW, rm, rps);
        else hipLaunchKernelGGL((wgrad_kernel<128, PRO_NONE>), grid, block, 0, st, A, D, D2, dW, rm, rps);
    } else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
