// cap: node x cluster soft assignment with dynamic routing, intra-cluster aggregation, cluster -> node scatter
// (reference GPTST.py:100-141).  One workgroup per (b,t): the N x C capsule matrix P = squash(X Wp^T + bp) is built by
// fp32 MFMA straight into LDS and stays there for the whole routing loop; the (B,T,HS,N,C) tensor of the reference
// (:106-107) is never formed:   s[h,:] = v0[h,:] (.) sum_n c[h,n] P[n,:].
//
// Two thread mappings alternate (separated by barriers):
//   map A  (row-parallel):   C/4 lanes own one node row n as float4s; dot products against the HS cluster vectors are
//                            reduced with a wave-level segmented (C/4-lane) butterfly, then the lanes split the HS logits
//                            for the softmax over clusters;
//   map B  (cluster-parallel): thread (h, c4) accumulates sum_n c[h,n] P[n, 4c4..] over the LDS-resident rows.
// Soft assignments / logits are kept as [h][n] (the layout of the `adj` parameter (ds,HS,N) and of the returned HS1).
#include "mfma_tile.h"

template <int C>
struct CapCfg {
    static constexpr int LPR = C / 4;             // lanes per row
    static constexpr int RPP = 256 / LPR;         // rows per pass of map A
    static constexpr int PITCH = Tile<C>::PITCH;
};

__host__ __device__ inline int cap_npad(int N) { return (N + 31) / 32 * 32; }
__host__ __device__ inline int cap_np(int N) { return N | 1; }     // odd LDS pitch of the [h][n] arrays

// ---- map A: optional routing-logit update, then softmax over clusters ----------------------------------------------------
//   do_dots : bl[h][n] += V[h,:] . P[n,:]          (V in LDS, HS x C)
//   logits  = (use_bl ? bl : 0) + (use_l0 ? L0 : 0);  cs[h][n] = softmax_h(logits);  optional copy to global c_out[h*N+n]
template <int C>
__device__ __forceinline__ void cap_pass_a(const float* __restrict__ Ps, const float* __restrict__ V, float* __restrict__ bl,
                                           const float* __restrict__ L0, float* __restrict__ cs, float* __restrict__ c_out,
                                           int N, int HS, int NP, bool do_dots, bool use_bl, bool use_l0) {
    using K = CapCfg<C>;
    const int slot = threadIdx.x / K::LPR, j = threadIdx.x % K::LPR;
    for (int n0 = 0; n0 < N; n0 += K::RPP) {
        const int n = n0 + slot;
        const bool valid = n < N;
        if (do_dots) {
            const float4 p4 = valid ? ld4(Ps + n * K::PITCH + 4 * j) : f4zero();
            for (int h = 0; h < HS; ++h) {
                const float u = group_sum<K::LPR>(f4dot(ld4(V + h * C + 4 * j), p4));
                if (valid && j == (h % K::LPR)) bl[h * NP + n] += u;
            }
        }
        float m = -3.0e38f;
        for (int h = j; h < HS; h += K::LPR) {
            float l = 0.f;
            if (valid) l = (use_bl ? bl[h * NP + n] : 0.f) + (use_l0 ? L0[h * NP + n] : 0.f);
            m = fmaxf(m, l);
        }
        m = group_max<K::LPR>(m);
        float sum = 0.f;
        for (int h = j; h < HS; h += K::LPR) {
            float l = 0.f;
            if (valid) l = (use_bl ? bl[h * NP + n] : 0.f) + (use_l0 ? L0[h * NP + n] : 0.f);
            const float e = expf(l - m);
            sum += e;
            if (valid) cs[h * NP + n] = e;
        }
        sum = group_sum<K::LPR>(sum);
        const float inv = 1.f / sum;
        for (int h = j; h < HS; h += K::LPR) {
            if (valid) {
                const float c = cs[h * NP + n] * inv;
                cs[h * NP + n] = c;
                if (c_out != nullptr) c_out[(size_t)h * N + n] = c;
            }
        }
    }
}

// ---- map B: acc[h][c4] = sum_n cs[h][n] * P[n][4c4..];  post: 0 -> dst = acc, 1 -> dst = squash(acc), 2 -> dst = squash(mul (.) acc)
template <int C>
__device__ __forceinline__ void cap_pass_b(const float* __restrict__ Ps, const float* __restrict__ cs, const float* __restrict__ mul,
                                           float* __restrict__ dst, int N, int HS, int NP, int post) {
    using K = CapCfg<C>;
    for (int base = 0; base < HS * K::LPR; base += 256) {
        const int pair = base + threadIdx.x;
        const bool valid = pair < HS * K::LPR;
        const int h = valid ? pair / K::LPR : 0, c4 = pair % K::LPR;
        float4 acc = f4zero();
        if (valid) {
            const float* crow = cs + h * NP;
            const float* pcol = Ps + 4 * c4;
#pragma unroll 4
            for (int n = 0; n < N; ++n) acc = f4fma(crow[n], ld4(pcol + n * K::PITCH), acc);
            if (post == 2) {
                const float4 m = ld4(mul + h * C + 4 * c4);
                acc = make_float4(acc.x * m.x, acc.y * m.y, acc.z * m.z, acc.w * m.w);
            }
        }
        if (post != 0) {
            const float sc = squash_scale(group_sum<K::LPR>(f4dot(acc, acc)));
            acc = make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc);
        }
        if (valid) st4(dst + h * C + 4 * c4, acc);
    }
}

// Y = X[bt] Wp^T (MFMA, into Ps rows), rows >= N zero.  Wl must already hold Wp^T as [k=in][j=out].
template <int C>
__device__ __forceinline__ void cap_linear_to_lds(const float* __restrict__ Xbt, const float* __restrict__ Wl, float* __restrict__ Ps,
                                                  int N, int NPAD) {
    using T = Tile<C>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < NPAD / 32; t += 4) {
        float* tile = Ps + t * 32 * T::PITCH;
#pragma unroll
        for (int it = 0; it < T::F4_PER_LANE; ++it) {
            const int f = it * 64 + lane;
            const int r = f / T::F4_PER_ROW, c4 = f % T::F4_PER_ROW;
            const int n = t * 32 + r;
            st4(tile + r * T::PITCH + 4 * c4, n < N ? ld4(Xbt + (size_t)n * C + 4 * c4) : f4zero());
        }
        f32x16 acc[T::NCT];
        mfma_tile<C>(tile, Wl, acc, lane);
        acc_to_tile<C>(tile, acc, lane);
    }
}

// forward: c (BT,HS,N) final soft assignment, s (BT,HS,C) intra-cluster aggregate  (GPTST.py:102-123)
template <int C>
__global__ __launch_bounds__(256) void cap_route_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Wp,
                                                            const float* __restrict__ bp, const float* __restrict__ dadj,
                                                            float* __restrict__ c_out,
                                                            float* __restrict__ s_out, int N, int HS, int R, int region2) {
    using K = CapCfg<C>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NPAD = cap_npad(N), NP = cap_np(N);
    float* Ps = smem;
    float* Wl = Ps + NPAD * K::PITCH;
    float* L0 = Wl;
    float* bl = L0 + HS * NP;
    float* cs = bl + HS * NP;
    float* v0s = Wl + region2;
    float* vs = v0s + HS * C;
    const int bt = blockIdx.x, tid = threadIdx.x;

    load_w_lds<C, 256>(Wl, Wp, 1, tid);
    __syncthreads();
    cap_linear_to_lds<C>(X + (size_t)bt * N * C, Wl, Ps, N, NPAD);
    __syncthreads();
    {   // P = squash(Y + bp) in place (map A rows); logits L0 = teb . adj; bl = 0
        const int slot = tid / K::LPR, j = tid % K::LPR;
        const float4 b4 = ld4(bp + 4 * j);
        for (int n0 = 0; n0 < NPAD; n0 += K::RPP) {
            const int n = n0 + slot;
            float4 y = f4zero();
            if (n < N) y = f4add(ld4(Ps + n * K::PITCH + 4 * j), b4);
            const float sc = squash_scale(group_sum<K::LPR>(f4dot(y, y)));
            if (n < NPAD) st4(Ps + n * K::PITCH + 4 * j, make_float4(y.x * sc, y.y * sc, y.z * sc, y.w * sc));
        }
        for (int i = tid; i < HS * N; i += 256) {
            const int h = i / N, n = i % N;
            const float l = dadj[(size_t)bt * HS * N + i];
            L0[h * NP + n] = l;
            bl[h * NP + n] = 0.f;
        }
    }
    __syncthreads();
    cap_pass_a<C>(Ps, nullptr, bl, L0, cs, nullptr, N, HS, NP, false, false, true);        // c0 = softmax_h(dadj)        :105
    __syncthreads();
    cap_pass_b<C>(Ps, cs, nullptr, v0s, N, HS, NP, 1);                                     // v0 = squash(c0 . P)        :105-106
    __syncthreads();
    for (int r = 0; r < R; ++r) {                                                          // routing (no grad)          :113-118
        cap_pass_a<C>(Ps, vs, bl, L0, cs, nullptr, N, HS, NP, r > 0, true, false);         // b += v.P^T (prev iter); c = softmax(b)
        __syncthreads();
        cap_pass_b<C>(Ps, cs, v0s, vs, N, HS, NP, 2);                                      // v = squash(v0 (.) c.P)
        __syncthreads();
    }
    cap_pass_a<C>(Ps, vs, bl, L0, cs, c_out + (size_t)bt * HS * N, N, HS, NP, R > 0, true, true);   // c = softmax(b + dadj) :120
    __syncthreads();
    cap_pass_b<C>(Ps, cs, nullptr, s_out + (size_t)bt * HS * C, N, HS, NP, 0);             // s = c . P                  :123
}

// ---- cluster -> node scatter: rec[bt,n,:] = sum_h c[bt,h,n] v[bt,h,:]   (GPTST.py:135) --------------------------------------
template <int C>
__global__ __launch_bounds__(256) void cap_rec_fwd_kernel(const float* __restrict__ c, const float* __restrict__ v,
                                                          float* __restrict__ rec, int N, int HS) {
    using K = CapCfg<C>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* vs = smem;                    // HS*C
    float* cs = smem + HS * C;           // HS*N
    const int bt = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < HS * C / 4; i += 256) st4(vs + 4 * i, ld4(v + (size_t)bt * HS * C + 4 * i));
    for (int i = tid; i < HS * N; i += 256) cs[i] = c[(size_t)bt * HS * N + i];
    __syncthreads();
    const int slot = tid / K::LPR, j = tid % K::LPR;
    for (int n = slot; n < N; n += K::RPP) {
        float4 acc = f4zero();
        for (int h = 0; h < HS; ++h) acc = f4fma(cs[h * N + n], ld4(vs + h * C + 4 * j), acc);
        st4(rec + ((size_t)bt * N + n) * C + 4 * j, acc);
    }
}

// backward of the scatter:  dc1[bt,h,n] = drec[bt,n,:] . v[bt,h,:];   dv[bt,h,:] = sum_n c[bt,h,n] drec[bt,n,:]
template <int C>
__global__ __launch_bounds__(256) void cap_rec_bwd_kernel(const float* __restrict__ drec, const float* __restrict__ c,
                                                          const float* __restrict__ v, float* __restrict__ dc1,
                                                          float* __restrict__ dv, int N, int HS) {
    using K = CapCfg<C>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NP = cap_np(N);
    float* Ds = smem;                          // N * PITCH  (drec rows)
    float* vs = Ds + N * K::PITCH;             // HS*C
    float* cs = vs + HS * C;                   // HS*NP
    const int bt = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < N * K::LPR; i += 256) {
        const int n = i / K::LPR, c4 = i % K::LPR;
        st4(Ds + n * K::PITCH + 4 * c4, ld4(drec + ((size_t)bt * N + n) * C + 4 * c4));
    }
    for (int i = tid; i < HS * C / 4; i += 256) st4(vs + 4 * i, ld4(v + (size_t)bt * HS * C + 4 * i));
    for (int i = tid; i < HS * N; i += 256) cs[(i / N) * NP + i % N] = c[(size_t)bt * HS * N + i];
    __syncthreads();
    const int slot = tid / K::LPR, j = tid % K::LPR;
    for (int n0 = 0; n0 < N; n0 += K::RPP) {
        const int n = n0 + slot;
        const bool valid = n < N;
        const float4 d4 = valid ? ld4(Ds + n * K::PITCH + 4 * j) : f4zero();
        for (int h = 0; h < HS; ++h) {
            const float u = group_sum<K::LPR>(f4dot(ld4(vs + h * C + 4 * j), d4));
            if (valid && j == (h % K::LPR)) dc1[((size_t)bt * HS + h) * N + n] = u;
        }
    }
    cap_pass_b<C>(Ds, cs, nullptr, dv + (size_t)bt * HS * C, N, HS, NP, 0);
}

// ---- backward through s = c.P, c = softmax_h(b + dadj), P = squash(X Wp^T + bp)   (routing itself is detached) ------------
//   in : X, Wp, bp, c (BT,HS,N), dc1 (BT,HS,N: grad of c from the scatter), dS (BT,HS,C: total grad of s)
//   out: dY (BT*N, C) grad of the pre-squash Linear output, dlogit (BT,HS,N) grad of dadj
template <int C>
__global__ __launch_bounds__(256) void cap_route_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Wp,
                                                            const float* __restrict__ bp, const float* __restrict__ c,
                                                            const float* __restrict__ dc1, const float* __restrict__ dS,
                                                            float* __restrict__ dY, float* __restrict__ dlogit, int N, int HS,
                                                            int region2) {
    using K = CapCfg<C>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NPAD = cap_npad(N), NP = cap_np(N);
    float* Ys = smem;                          // NPAD*PITCH: Y = X Wp^T (bias added on the fly)
    float* Wl = Ys + NPAD * K::PITCH;
    float* cs = Wl;                            // HS*NP  (after the MFMA phase)
    float* dcs = cs + HS * NP;                 // HS*NP
    float* dss = Wl + region2;                 // HS*C
    const int bt = blockIdx.x, tid = threadIdx.x;
    load_w_lds<C, 256>(Wl, Wp, 1, tid);
    __syncthreads();
    cap_linear_to_lds<C>(X + (size_t)bt * N * C, Wl, Ys, N, NPAD);
    __syncthreads();
    for (int i = tid; i < HS * N; i += 256) {
        cs[(i / N) * NP + i % N] = c[(size_t)bt * HS * N + i];
        dcs[(i / N) * NP + i % N] = dc1[(size_t)bt * HS * N + i];
    }
    for (int i = tid; i < HS * C / 4; i += 256) st4(dss + 4 * i, ld4(dS + (size_t)bt * HS * C + 4 * i));
    __syncthreads();
    const int slot = tid / K::LPR, j = tid % K::LPR;
    const float4 b4 = ld4(bp + 4 * j);
    for (int n0 = 0; n0 < N; n0 += K::RPP) {
        const int n = n0 + slot;
        const bool valid = n < N;
        float4 y = f4zero();
        if (valid) y = f4add(ld4(Ys + n * K::PITCH + 4 * j), b4);
        const float q = group_sum<K::LPR>(f4dot(y, y));
        const float r = sqrtf(q), den = (1.f + q) * (r + 1e-8f);
        const float g = q / den;                                   // P = g * Y
        const float4 p4 = make_float4(g * y.x, g * y.y, g * y.z, g * y.w);
        // dc[h] = dc1[h] + dS[h,:].P[n,:];   dP = sum_h c[h] dS[h,:]
        float4 dp = f4zero();
        float mine = 0.f;        // this lane's dc for its clusters h = j, j+LPR, ... (softmax-weighted sum needs all of them)
        float wsum = 0.f;
        for (int h = 0; h < HS; ++h) {
            const float4 s4 = ld4(dss + h * C + 4 * j);
            const float u = group_sum<K::LPR>(f4dot(s4, p4));
            const float ch = valid ? cs[h * NP + n] : 0.f;
            const float dch = (valid ? dcs[h * NP + n] : 0.f) + u;
            dp = f4fma(ch, s4, dp);
            wsum = fmaf(ch, dch, wsum);                            // identical on all lanes of the row
            if (valid && j == (h % K::LPR)) dcs[h * NP + n] = dch;
        }
        (void)mine;
        for (int h = j; h < HS; h += K::LPR)
            if (valid) dlogit[((size_t)bt * HS + h) * N + n] = cs[h * NP + n] * (dcs[h * NP + n] - wsum);
        // squash backward: dY = g dP + Y * (2 g'(q) (Y.dP))
        const float ydp = group_sum<K::LPR>(f4dot(y, dp));
        float gp = 0.f;
        if (r > 0.f) gp = ((1.f + q) * (r + 1e-8f) - q * ((r + 1e-8f) + (1.f + q) * 0.5f / r)) / (den * den);
        const float k2 = 2.f * gp * ydp;
        if (valid)
            st4(dY + ((size_t)bt * N + n) * C + 4 * j,
                make_float4(fmaf(k2, y.x, g * dp.x), fmaf(k2, y.y, g * dp.y), fmaf(k2, y.z, g * dp.z), fmaf(k2, y.w, g * dp.w)));
    }
}

static size_t cap_region2(int C, int N, int HS) {
    size_t a = (size_t)C * C, b = (size_t)3 * HS * cap_np(N);
    return ((a > b ? a : b) + 3) & ~(size_t)3;
}

template <int C>
static int launch_route_fwd(const float* X, const float* Wp, const float* bp, const float* dadj, float* c_out, float* s_out, int BT,
                            int N, int HS, int R, hipStream_t st) {
    const size_t r2 = cap_region2(C, N, HS);
    const size_t smem = ((size_t)cap_npad(N) * CapCfg<C>::PITCH + r2 + 2 * (size_t)HS * C) * sizeof(float);
    if (smem > 160 * 1024) return GPTST_ESHAPE;
    static size_t cur = 0;
    if (smem > cur) { hipFuncSetAttribute((const void*)cap_route_fwd_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); cur = smem; }
    hipLaunchKernelGGL((cap_route_fwd_kernel<C>), dim3(BT), dim3(256), smem, st, X, Wp, bp, dadj, c_out, s_out, N, HS, R, (int)r2);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// first-generation (VALU contractions) forward: kept as the fallback for shapes the MFMA version rejects (HS > 64)
GPTST_INTERNAL int gptst_cap_route_fwd_v1(const float* X, const float* Wp, const float* bp, const float* dadj, float* c_out, float* s_out,
                                      int BT, int N, int C, int HS, int R, void* stream) {
    if (!X || !Wp || !bp || !dadj || !c_out || !s_out || HS <= 0 || R < 0) return GPTST_EARG;
    if (C == 64) return launch_route_fwd<64>(X, Wp, bp, dadj, c_out, s_out, BT, N, HS, R, (hipStream_t)stream);
    if (C == 128) return launch_route_fwd<128>(X, Wp, bp, dadj, c_out, s_out, BT, N, HS, R, (hipStream_t)stream);
    return GPTST_ESHAPE;
}

template <int C>
static int launch_rec(const float* c, const float* v, float* rec, const float* drec, float* dc1, float* dv, int BT, int N, int HS,
                      hipStream_t st) {
    if (rec) {
        const size_t smem = ((size_t)HS * C + (size_t)HS * N) * sizeof(float);
        if (smem > 160 * 1024) return GPTST_ESHAPE;
        static size_t cur = 0;
        if (smem > cur) { hipFuncSetAttribute((const void*)cap_rec_fwd_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); cur = smem; }
        hipLaunchKernelGGL((cap_rec_fwd_kernel<C>), dim3(BT), dim3(256), smem, st, c, v, rec, N, HS);
    } else {
        const size_t smem = ((size_t)N * CapCfg<C>::PITCH + (size_t)HS * C + (size_t)HS * cap_np(N)) * sizeof(float);
        if (smem > 160 * 1024) return GPTST_ESHAPE;
        static size_t cur = 0;
        if (smem > cur) { hipFuncSetAttribute((const void*)cap_rec_bwd_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); cur = smem; }
        hipLaunchKernelGGL((cap_rec_bwd_kernel<C>), dim3(BT), dim3(256), smem, st, drec, c, v, dc1, dv, N, HS);
    }
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_cap_rec_fwd(const float* c, const float* v, float* rec, int BT, int N, int C, int HS, void* stream) {
    if (!c || !v || !rec) return GPTST_EARG;
    if (C == 64) return launch_rec<64>(c, v, rec, nullptr, nullptr, nullptr, BT, N, HS, (hipStream_t)stream);
    if (C == 128) return launch_rec<128>(c, v, rec, nullptr, nullptr, nullptr, BT, N, HS, (hipStream_t)stream);
    return GPTST_ESHAPE;
}

GPTST_INTERNAL int gptst_cap_rec_bwd_v1(const float* drec, const float* c, const float* v, float* dc1, float* dv, int BT, int N, int C,
                                 int HS, void* stream) {
    if (!drec || !c || !v || !dc1 || !dv) return GPTST_EARG;
    if (C == 64) return launch_rec<64>(c, v, nullptr, drec, dc1, dv, BT, N, HS, (hipStream_t)stream);
    if (C == 128) return launch_rec<128>(c, v, nullptr, drec, dc1, dv, BT, N, HS, (hipStream_t)stream);
    return GPTST_ESHAPE;
}

template <int C>
static int launch_route_bwd(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1, const float* dS,
                            float* dY, float* dlogit, int BT, int N, int HS, hipStream_t st) {
    size_t a = (size_t)C * C, b = (size_t)2 * HS * cap_np(N);
    const size_t r2 = ((a > b ? a : b) + 3) & ~(size_t)3;
    const size_t smem = ((size_t)cap_npad(N) * CapCfg<C>::PITCH + r2 + (size_t)HS * C) * sizeof(float);
    if (smem > 160 * 1024) return GPTST_ESHAPE;
    static size_t cur = 0;
    if (smem > cur) { hipFuncSetAttribute((const void*)cap_route_bwd_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); cur = smem; }
    hipLaunchKernelGGL((cap_route_bwd_kernel<C>), dim3(BT), dim3(256), smem, st, X, Wp, bp, c, dc1, dS, dY, dlogit, N, HS, (int)r2);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

GPTST_INTERNAL int gptst_cap_route_bwd_v1(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1,
                                   const float* dS, float* dY, float* dlogit, int BT, int N, int C, int HS, void* stream) {
    if (!X || !Wp || !bp || !c || !dc1 || !dS || !dY || !dlogit) return GPTST_EARG;
    if (C == 64) return launch_route_bwd<64>(X, Wp, bp, c, dc1, dS, dY, dlogit, BT, N, HS, (hipStream_t)stream);
    if (C == 128) return launch_route_bwd<128>(X, Wp, bp, c, dc1, dS, dY, dlogit, BT, N, HS, (hipStream_t)stream);
    return GPTST_ESHAPE;
}
