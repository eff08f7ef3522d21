// hypertem_fwd12_kernel (one time step per wave, 768 threads) and hypertem_fwd128_kernel (slab-fused forward at C = 128): built, parity-tested,
// measured slower than what they were meant to replace (DESIGN.md sections 7 / 8: 21.8 -> 21.4 us at B = 32; 1086 us against 285 + 587 us at
// N = 4096), and removed from the product library in r05.  Kept as a record: they compile inside gpt-st_amd/csrc/hypertem.hip
// behind hypertem_fwd_kernel (same helpers / macros).

// ---- 12-wave variant (r03): one time step per wave ------------------------------------------------------------------------------------------
// The 4-wave kernel above runs three time steps per wave one after the other (mix -> 64 MFMAs -> epilogue, ~2 us each) behind the slab load,
// and at B = 32 the 352 workgroups make 1.4 rounds of two co-resident workgroups.  Here a workgroup has 12 waves (3 per SIMD = the register
// budget of 168), each owning ONE time step: the per-workgroup chain is load -> one step -> store, a CU holds one workgroup at a time, and
// the three waves of a SIMD interleave mix (VALU) and MFMA phases.  NT = 16 only.
__global__ __launch_bounds__(768, 3) void hypertem_fwd12_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                                 const float* __restrict__ Wbt, const float* __restrict__ bbt,
                                                                 float* __restrict__ R_out, float* __restrict__ out, int N, int B) {
    constexpr int C = 64, P = C + 4, GP = 145, NT = 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                               // [12][NT][P]
    float* Gs = Xs + HT_T * NT * P;                 // [NT][GP]
    int b, tile;
#ifdef GPTST_DEBUG
#define TS12(i) do { if (blockIdx.x == 59 && threadIdx.x == 64 * 5) g_ht_ts[i] = __builtin_readcyclecounter(); if (blockIdx.x == 200 && threadIdx.x == 64 * 5) g_ht_ts[16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define TS12(i) do { } while (0)
#endif
    TS12(0);
    if (!ht_work((N + NT - 1) / NT, B, b, tile)) return;
    const int n0 = tile * NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int t = wave;
    const size_t g = (size_t)b * HT_T + t;
    // this wave's W_bt fragments and bias: requested first, consumed after the mix
    float4 bv[C / 16][4];
    {
        const float* W_ = Wbt + g * C * C;
#pragma unroll
        for (int q = 0; q < C / 16; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[q][e] = ld4(W_ + (size_t)(16 * q + 4 * kk + e) * C + 4 * j);
    }
    const float4 b4 = ld4(bbt + g * C + 4 * j);
    {   // slab staging: thread = (time third, row, float4 column): 4 time slices each
        const int th = tid >> 8, r = tid & 255, nl = r >> 4, c4 = r & 15;
        const int n = min(n0 + nl, N - 1);
        float4 v[4];
        float gv[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ld4(X + (((size_t)b * HT_T + 4 * th + i) * N + n) * C + 4 * c4);
#pragma unroll
        for (int k = 0; k < 3; ++k) gv[k] = G[min(n0 * 144 + tid + k * 768, N * 144 - 1)];
#pragma unroll
        for (int i = 0; i < 4; ++i) st4(Xs + ((4 * th + i) * NT + nl) * P + 4 * c4, n0 + nl < N ? v[i] : f4zero());
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = tid + k * 768;
            Gs[(i / 144) * GP + i % 144] = (n0 + i / 144 < N) ? gv[k] : 0.f;
        }
    }
    TS12(1);
    __syncthreads();
    SB();
    TS12(2);
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
    TS12(3);
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
    TS12(4);
    if (R_out != nullptr && n0 + j < N) {
#pragma unroll
        for (int q = 0; q < C / 16; ++q) st4(R_out + (g * N + n0 + j) * C + 16 * q + 4 * kk, a4[q]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int nl = kk * 4 + r;
        if (n0 + nl < N) {
            float4 y = f4add(f4add(make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), b4), ld4(Xs + (t * NT + nl) * P + 4 * j));
            y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
            st4(out + (g * N + n0 + nl) * C + 4 * j, y);
        }
    }
    TS12(5);
#ifdef GPTST_DEBUG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    TS12(6);
}

// ---- C = 128 (BASELINE configs[4]) ---------------------------------------------------------------------------------------------------------
// Same fusion at C = 128: out = LReLU((G_n X) W_bt + b_bt + X) with the 12 x 16 x 128 slab of X in LDS (101 KB: one workgroup per CU) and
// R = G_n X written once for the weight gradient — replaces tmix_kernel + apply128_kernel<TIME> (the R round trip through HBM and a second
// pass over X for the residual).  W_bt is 64 KB per (b, t): a lane cannot hold a whole matrix as at C = 64, so the unit of work is
// (time step, half of the output channels): 24 items over 8 waves (two per SIMD: one's W_bt fragment loads from L2 run under the other's
// MFMAs), each item = mix (A operand, VALU from LDS) -> 128 MFMAs against its 128 x 64 block of W_bt (32 float4 fragments in registers,
// the next item's requested right behind the MFMAs) -> bias + residual + LeakyReLU from registers.  The two waves that share a time step
// both compute its mix; the one with the lower half writes R.  Work map: all node tiles of a sample on one XCD (ht_work), so the sample's
// 768 KB of W_bt are fetched into one L2.
// MEASURED (N = 4096, B = 32, r03): 1086 us per launch against 285 + 587 us for the two kernels it replaces — with one workgroup per CU the slab
// staging, the (duplicated) mixes and the MFMA phases of a tile run one after the other (34 us per tile, MFMA alone 11.7) and nothing of the
// next tile overlaps them; the two-kernel form keeps 2-4 workgroups per CU.  Kept for parity coverage and as the starting point of a
// pipelined version; the engine uses it only with GPTST_HT128_FUSED=1.
#define HT128_NT 16
static size_t ht128_smem() { return ((size_t)HT_T * HT128_NT * 132 + HT128_NT * 145) * sizeof(float); }
__global__ __launch_bounds__(512, 1) void hypertem_fwd128_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                                 const float* __restrict__ Wbt, const float* __restrict__ bbt,
                                                                 float* __restrict__ R_out, float* __restrict__ out, int N, int B) {
    constexpr int C = 128, P = C + 4, GP = 145, NT = HT128_NT, NQ = C / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                               // [12][NT][P]
    float* Gs = Xs + HT_T * NT * P;                 // [NT][GP]
    int b, tile;
    if (!ht_work((N + NT - 1) / NT, B, b, tile)) return;
    const int n0 = tile * NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    float4 bv[NQ][4];
    float4 b4;
#define HT128_LOAD_W(item) do {                                                                                    \
        const size_t g_ = (size_t)b * HT_T + ((item) >> 1);                                                        \
        const float* W_ = Wbt + g_ * C * C + 64 * ((item) & 1);                                                    \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                                             \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) bv[q][e] = ld4(W_ + (size_t)(16 * q + 4 * kk + e) * C + 4 * j); \
        b4 = ld4(bbt + g_ * C + 64 * ((item) & 1) + 4 * j);                                                        \
    } while (0)
    HT128_LOAD_W(wave);                             // in flight during the slab staging
    {   // slab + graph staging: every global load before the first LDS store
        const int nl = tid >> 5, c4 = tid & 31;     // thread = (row, float4 column) of every time slice
        const int n = min(n0 + nl, N - 1);
        float4 v[HT_T];
        float gv[5];
#pragma unroll
        for (int t = 0; t < HT_T; ++t) v[t] = ld4(X + (((size_t)b * HT_T + t) * N + n) * C + 4 * c4);
#pragma unroll
        for (int k = 0; k < 5; ++k) gv[k] = G[min(n0 * 144 + tid + k * 512, N * 144 - 1)];
#pragma unroll
        for (int t = 0; t < HT_T; ++t) st4(Xs + (t * NT + nl) * P + 4 * c4, n0 + nl < N ? v[t] : f4zero());
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = tid + k * 512;
            if (i < NT * 144) Gs[(i / 144) * GP + i % 144] = (n0 + i / 144 < N) ? gv[k] : 0.f;
        }
    }
    __syncthreads();
    for (int item = wave; item < 2 * HT_T; item += 8) {
        const int t = item >> 1, half = item & 1;
        const size_t g = (size_t)b * HT_T + t;
        SB();
        // ---- (1) temporal mix in the MFMA A-operand layout: lane (j, kk) owns R_t[row j][16q + 4kk .. +3] ----
        float4 a4[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) a4[q] = f4zero();
        {
            const float* gr = Gs + j * GP + t * HT_T;
            const float* xr = Xs + j * P + 4 * kk;
#pragma unroll
            for (int u = 0; u < HT_T; ++u) {
                const float gu = gr[u];
#pragma unroll
                for (int q = 0; q < NQ; ++q) a4[q] = f4fma(gu, ld4(xr + u * NT * P + 16 * q), a4[q]);
            }
        }
        SB();
        // ---- (2) R_t @ W_bt[:, 64 half ..]: column tile ct, column j <-> output channel 64 half + 4j + ct ----
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
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
        if (item + 8 < 2 * HT_T) HT128_LOAD_W(item + 8);         // requested before this item's stores (vmcnt retires in order)
        SB();
        if (half == 0 && R_out != nullptr && n0 + j < N) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) st4(R_out + (g * N + n0 + j) * C + 16 * q + 4 * kk, a4[q]);
        }
        // ---- (3) epilogue from registers: lane (j, kk) owns rows kk*4 + r, channels 64 half + 4j .. +3 ----
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nl = kk * 4 + r;
            if (n0 + nl < N) {
                float4 y = f4add(f4add(make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), bias),
                                 ld4(Xs + (t * NT + nl) * P + 64 * half + 4 * j));
                y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
                st4(out + (g * N + n0 + nl) * C + 64 * half + 4 * j, y);
            }
        }
    }
#undef HT128_LOAD_W
}

