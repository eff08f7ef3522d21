// cap, inter-cluster / cross-time hyperedges on the T*HS cluster tokens of one sample (reference GPTST.py:125-134):
//     Z   = s + (t+1)/12                                   (B, KK = T*HS, C)
//     Ht  = LReLU(dyn Z)        dyn = time_eb_spg . t_adj  (B, HT, KK), produced by poolgen
//     Rt  = LReLU(dyn^T Ht)
//     v   = squash(Rt + s)
// Tiny (0.5 MFLOP per sample) and latency bound: one workgroup per sample, everything LDS-resident, VALU only.
#include "common.h"

#define CX_NT 1024     // threads per sample: every phase is a short latency chain, so use all 16 waves of a CU

template <int C>
struct CrossCfg {
    static constexpr int LPR = C / 4;
    static constexpr int PITCH = C + 4;
};

template <int C>
__global__ __launch_bounds__(CX_NT) void cap_cross_fwd_kernel(const float* __restrict__ s, const float* __restrict__ dyn,
                                                            const float* __restrict__ tmpl, float* __restrict__ v,
                                                            float* __restrict__ Ht_out, float* __restrict__ Rt_out, int T, int HS,
                                                            int HT) {
    using K = CrossCfg<C>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KK = T * HS;
    float* Zs = smem;                         // KK * PITCH
    float* Hs = Zs + KK * K::PITCH;           // HT * C
    float* dyns = Hs + HT * C;                // HT * KK
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < KK * K::LPR; i += CX_NT) {
        const int k = i / K::LPR, c4 = i % K::LPR;
        const float tm = tmpl[k / HS];
        const float4 x = ld4(s + ((size_t)b * KK + k) * C + 4 * c4);
        st4(Zs + k * K::PITCH + 4 * c4, make_float4(x.x + tm, x.y + tm, x.z + tm, x.w + tm));
    }
    for (int i = tid; i < HT * KK; i += CX_NT) dyns[i] = dyn[(size_t)b * HT * KK + i];
    __syncthreads();
    for (int i = tid; i < HT * K::LPR; i += CX_NT) {
        const int j = i / K::LPR, c4 = i % K::LPR;
        float4 acc = f4zero();
#pragma unroll 8
        for (int k = 0; k < KK; ++k) acc = f4fma(dyns[j * KK + k], ld4(Zs + k * K::PITCH + 4 * c4), acc);
        acc = make_float4(lrelu(acc.x), lrelu(acc.y), lrelu(acc.z), lrelu(acc.w));
        st4(Hs + j * C + 4 * c4, acc);
        st4(Ht_out + ((size_t)b * HT + j) * C + 4 * c4, acc);
    }
    __syncthreads();
    for (int base = 0; base < KK * K::LPR; base += CX_NT) {
        const int i = base + tid;
        const bool valid = i < KK * K::LPR;
        const int k = valid ? i / K::LPR : 0, c4 = i % K::LPR;
        float4 acc = f4zero();
#pragma unroll 8
        for (int j = 0; j < HT; ++j) acc = f4fma(dyns[j * KK + k], ld4(Hs + j * C + 4 * c4), acc);
        const float4 rt = make_float4(lrelu(acc.x), lrelu(acc.y), lrelu(acc.z), lrelu(acc.w));
        const float tm = tmpl[k / HS];
        const float4 z = ld4(Zs + k * K::PITCH + 4 * c4);
        const float4 u = make_float4(rt.x + (z.x - tm), rt.y + (z.y - tm), rt.z + (z.z - tm), rt.w + (z.w - tm));
        const float sc = squash_scale(group_sum<K::LPR>(f4dot(u, u)));
        if (valid) {
            st4(Rt_out + ((size_t)b * KK + k) * C + 4 * c4, rt);
            st4(v + ((size_t)b * KK + k) * C + 4 * c4, make_float4(u.x * sc, u.y * sc, u.z * sc, u.w * sc));
        }
    }
}

// ---- cross-time block + cluster -> node scatter in ONE launch (r03) -------------------------------------------------------------------
// cap_cross_fwd runs on B workgroups (12 % of the CUs at B = 32) between two (b,t)-grouped kernels, and every launch has a fixed cost of
// several microseconds.  Here each (b,t) workgroup of the scatter  rec[bt,n,:] = sum_h c[bt,h,n] v[bt,h,:]  (GPTST.py:135)  first rebuilds what
// it needs of the sample's cross-time block: Ht = LReLU(dyn Z) for the whole sample (16 x C outputs over T*HS tokens — the 12 workgroups of
// a sample repeat it, 0.25 MFLOP each), then Rt and v for its OWN HS tokens only.  Same arithmetic and summation order as
// cap_cross_fwd_kernel, so v / Ht / Rt are bit-identical; Rt and v rows are written by their owner, Ht by the t = 0 workgroup.
template <int C>
__global__ __launch_bounds__(256) void cap_cross_rec_fwd_kernel(const float* __restrict__ s, const float* __restrict__ dyn,
                                                                const float* __restrict__ tmpl, const float* __restrict__ c,
                                                                float* __restrict__ v, float* __restrict__ Ht_out,
                                                                float* __restrict__ Rt_out, float* __restrict__ rec, int T, int HS, int HT,
                                                                int N) {
    using K = CrossCfg<C>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KK = T * HS;
    float* Zs = smem;                         // KK * PITCH
    float* Hs = Zs + KK * K::PITCH;           // HT * C
    float* dyns = Hs + HT * C;                // HT * KK
    float* vs = dyns + HT * KK;               // HS * C      (offset is a multiple of 4 floats: HT*KK = HT*T*HS with T = 12)
    float* cs = vs + HS * C;                  // HS * N
    const int bt = blockIdx.x, b = bt / T, t = bt % T, tid = threadIdx.x;
    {   // staging: every global load of a batch is issued before the first LDS store (a copy loop `lds[i] = glb[i]` compiles to one
        // serialised L2 round trip per trip — 21 of them here, most of the former 8 us of this block)
        const float* sb = s + (size_t)b * KK * C;
        const float* db = dyn + (size_t)b * HT * KK;
        const float* cb = c + (size_t)bt * HS * N;
        const int nz = KK * K::LPR, nd = HT * KK / 4, nc = HS * N;
        const bool c4ok = (nc & 3) == 0;
        for (int i0 = 0; i0 < nz; i0 += 8 * 256) {
            float4 zv[8], dv[2], cv[2];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * 256 + tid; zv[u] = ld4(sb + 4 * (size_t)min(i, nz - 1)); }
            if (i0 == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    dv[u] = ld4(db + 4 * (size_t)min(u * 256 + tid, nd - 1));
                    cv[u] = c4ok ? ld4(cb + 4 * (size_t)min(u * 256 + tid, nc / 4 - 1)) : f4zero();
                }
            }
            SB();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256 + tid;
                if (i < nz) {
                    const int k = i / K::LPR, c4 = i % K::LPR;
                    const float tm = tmpl[k / HS];
                    st4(Zs + k * K::PITCH + 4 * c4, make_float4(zv[u].x + tm, zv[u].y + tm, zv[u].z + tm, zv[u].w + tm));
                }
            }
            if (i0 == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = u * 256 + tid;
                    if (i < nd) st4(dyns + 4 * i, dv[u]);
                    if (c4ok && i < nc / 4) st4(cs + 4 * i, cv[u]);
                }
            }
        }
        for (int i = 2 * 256 + tid; i < nd; i += 256) st4(dyns + 4 * i, ld4(db + 4 * (size_t)i));          // (beyond the bench shape)
        for (int i = (c4ok ? 2 * 256 * 4 : 0) + tid; i < nc; i += 256) cs[i] = cb[i];
    }
    __syncthreads();
    // Both products on MFMA 16x16x4 (an fp32 MFMA is the fmaf chain over its four k values in order, and the k-steps run in order: the
    // results are bit-identical to the VALU loops of cap_cross_fwd_kernel, tests/test_gpu_kernels.py::test_cap_cross_folded...).
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kk = lane >> 4;
    for (int jt = 0; jt < HT; jt += 16) {               // Ht[j][c] = LReLU(sum_k dyn[j][k] Z[k][c]): wave = column tile, 30 dependent steps
        const int ja = jt + li;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* ar = dyns + min(ja, HT - 1) * KK + kk;
        const float* br = Zs + kk * K::PITCH + 16 * wave + li;
        const int ns4 = KK / 4;
        int s4 = 0;
        for (; s4 + 6 <= ns4; s4 += 6) {                    // six steps' LDS operands in flight ahead of the dependent MFMA chain
            float av[6], bv[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) { av[u] = ja < HT ? ar[4 * (s4 + u)] : 0.f; bv[u] = br[4 * (s4 + u) * K::PITCH]; }
#pragma unroll
            for (int u = 0; u < 6; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        }
        for (; s4 < ns4; ++s4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ja < HT ? ar[4 * s4] : 0.f, br[4 * s4 * K::PITCH], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jj = jt + 4 * kk + r;
            if (jj < HT) {
                const float h = lrelu(acc[r]);
                Hs[jj * C + 16 * wave + li] = h;
                if (t == 0) Ht_out[((size_t)b * HT + jj) * C + 16 * wave + li] = h;
            }
        }
    }
    __syncthreads();
    if (wave == 0) {                                    // own tokens k = t*HS + h:  Rt = LReLU(dyn^T Ht), v = squash(Rt + s)
        for (int ht = 0; ht < HS; ht += 16) {
            const int h = ht + li;
            f32x4 acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int s4 = 0; s4 < (HT + 3) / 4; ++s4) {
                const int jj = 4 * s4 + kk;
                const float a = (h < HS && jj < HT) ? dyns[jj * KK + t * HS + h] : 0.f;
                const float4 bq = jj < HT ? ld4(Hs + jj * C + 4 * li) : f4zero();      // channel 4*li + ct <-> column li of tile ct
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq.w, acc[3], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hh = ht + 4 * kk + r, k = t * HS + min(hh, HS - 1);
                const float4 rt = make_float4(lrelu(acc[0][r]), lrelu(acc[1][r]), lrelu(acc[2][r]), lrelu(acc[3][r]));
                const float tm = tmpl[k / HS];
                const float4 z = ld4(Zs + k * K::PITCH + 4 * li);
                const float4 u = make_float4(rt.x + (z.x - tm), rt.y + (z.y - tm), rt.z + (z.z - tm), rt.w + (z.w - tm));
                const float sc = squash_scale(group_sum<16>(f4dot(u, u)));
                if (hh < HS) {
                    const float4 vv = make_float4(u.x * sc, u.y * sc, u.z * sc, u.w * sc);
                    st4(Rt_out + ((size_t)b * KK + k) * C + 4 * li, rt);
                    st4(v + ((size_t)b * KK + k) * C + 4 * li, vv);
                    st4(vs + hh * C + 4 * li, vv);
                }
            }
        }
    }
    __syncthreads();
    constexpr int RPP = 256 / K::LPR;
    const int slot = tid / K::LPR, j = tid % K::LPR;
#ifndef CXR_UNROLL
#define CXR_UNROLL 2          // (1: 896.1, 2: 898.8, 4: 893.7 steps/s, same box)
#endif
    // (r06: CXR_UNROLL nodes per trip — the v rows are read once per trip instead of once per node and the trips' LDS reads overlap; same fmaf chain
    //  over h per node, bit-identical)
    for (int n0 = slot; n0 < N; n0 += RPP * CXR_UNROLL) {            // as cap_rec_fwd_kernel (cap.hip)
        float4 acc[CXR_UNROLL];
#pragma unroll
        for (int u = 0; u < CXR_UNROLL; ++u) acc[u] = f4zero();
        for (int h = 0; h < HS; ++h) {
            const float4 vh = ld4(vs + h * C + 4 * j);
#pragma unroll
            for (int u = 0; u < CXR_UNROLL; ++u) acc[u] = f4fma(cs[h * N + min(n0 + u * RPP, N - 1)], vh, acc[u]);
        }
#pragma unroll
        for (int u = 0; u < CXR_UNROLL; ++u) {
            const int n = n0 + u * RPP;
            if (n < N) st4(rec + ((size_t)bt * N + n) * C + 4 * j, acc[u]);
        }
    }
}

// (s, dyn, tmpl, c) -> v, Ht, Rt, rec.  GPTST_ESHAPE when the sample's tokens + the (b,t) soft assignment do not fit LDS (use the two launches).
extern "C" int gptst_cap_cross_rec_fwd(const float* s, const float* dyn, const float* tmpl, const float* c, float* v, float* Ht, float* Rt,
                                       float* rec, int B, int T, int N, int C, int HS, int HT, void* stream) {
    if (!s || !dyn || !tmpl || !c || !v || !Ht || !Rt || !rec || B <= 0 || T <= 0 || N <= 0 || HS <= 0 || HT <= 0) return GPTST_EARG;
    if (C != 64 || (T * HS * HT) % 4 != 0) return GPTST_ESHAPE;
    using K = CrossCfg<64>;
    const int KK = T * HS;
    const size_t smem = ((size_t)KK * K::PITCH + (size_t)HT * 64 + (size_t)HT * KK + (size_t)HS * 64 + (size_t)HS * N) * sizeof(float);
    if (smem > 80 * 1024) return GPTST_ESHAPE;            // two workgroups per CU, as the scatter alone
    static size_t cur = 0;
    if (smem > cur) { hipFuncSetAttribute((const void*)cap_cross_rec_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); cur = smem; }
    hipLaunchKernelGGL((cap_cross_rec_fwd_kernel<64>), dim3(B * T), dim3(256), smem, (hipStream_t)stream, s, dyn, tmpl, c, v, Ht, Rt, rec, T, HS, HT, N);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// backward: dS (total grad of s) and ddyn from dv
template <int C>
__global__ __launch_bounds__(CX_NT) void cap_cross_bwd_kernel(const float* __restrict__ dv, const float* __restrict__ s,
                                                            const float* __restrict__ Rt, const float* __restrict__ Ht,
                                                            const float* __restrict__ dyn, const float* __restrict__ tmpl,
                                                            float* __restrict__ dS, float* __restrict__ ddyn, int T, int HS, int HT) {
    using K = CrossCfg<C>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KK = T * HS;
    float* Zs = smem;                         // KK * PITCH
    float* Gs = Zs + KK * K::PITCH;           // KK * PITCH   dRpre
    float* Hs = Gs + KK * K::PITCH;           // HT * PITCH   Ht
    float* dHs = Hs + HT * K::PITCH;          // HT * PITCH   dHpre
    float* dyns = dHs + HT * K::PITCH;        // HT * KK
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < HT * KK; i += CX_NT) dyns[i] = dyn[(size_t)b * HT * KK + i];
    for (int i = tid; i < HT * K::LPR; i += CX_NT)
        st4(Hs + (i / K::LPR) * K::PITCH + 4 * (i % K::LPR), ld4(Ht + ((size_t)b * HT) * C + 4 * i));
    // rows k: u = Rt + s;  du = squash_bwd(u, dv);  dRpre = du * lrelu'(Rt)
    for (int base = 0; base < KK * K::LPR; base += CX_NT) {
        const int i = base + tid;
        const bool valid = i < KK * K::LPR;
        const int k = valid ? i / K::LPR : 0, c4 = i % K::LPR;
        const size_t off = ((size_t)b * KK + k) * C + 4 * c4;
        float4 sv = f4zero(), rt = f4zero(), g = f4zero();
        if (valid) { sv = ld4(s + off); rt = ld4(Rt + off); g = ld4(dv + off); }
        const float4 u = f4add(rt, sv);
        const float q = group_sum<K::LPR>(f4dot(u, u));
        const float udg = group_sum<K::LPR>(f4dot(u, g));
        const float r = sqrtf(q), den = (1.f + q) * (r + 1e-8f);
        const float gq = q / den;
        float gp = 0.f;
        if (r > 0.f) gp = (den - q * ((r + 1e-8f) + (1.f + q) * 0.5f / r)) / (den * den);
        const float k2 = 2.f * gp * udg;
        const float4 du = make_float4(fmaf(k2, u.x, gq * g.x), fmaf(k2, u.y, gq * g.y), fmaf(k2, u.z, gq * g.z), fmaf(k2, u.w, gq * g.w));
        if (valid) {
            const float tm = tmpl[k / HS];
            st4(dS + off, du);
            st4(Zs + k * K::PITCH + 4 * c4, make_float4(sv.x + tm, sv.y + tm, sv.z + tm, sv.w + tm));
            st4(Gs + k * K::PITCH + 4 * c4, make_float4(du.x * lrelu_grad_from_out(rt.x), du.y * lrelu_grad_from_out(rt.y),
                                                        du.z * lrelu_grad_from_out(rt.z), du.w * lrelu_grad_from_out(rt.w)));
        }
    }
    __syncthreads();
    // dHpre[j] = lrelu'(Ht[j]) * sum_k dyn[j][k] dRpre[k]
    for (int i = tid; i < HT * K::LPR; i += CX_NT) {
        const int j = i / K::LPR, c4 = i % K::LPR;
        float4 acc = f4zero();
#pragma unroll 8
        for (int k = 0; k < KK; ++k) acc = f4fma(dyns[j * KK + k], ld4(Gs + k * K::PITCH + 4 * c4), acc);
        const float4 h = ld4(Hs + j * K::PITCH + 4 * c4);
        st4(dHs + j * K::PITCH + 4 * c4, make_float4(acc.x * lrelu_grad_from_out(h.x), acc.y * lrelu_grad_from_out(h.y),
                                                     acc.z * lrelu_grad_from_out(h.z), acc.w * lrelu_grad_from_out(h.w)));
    }
    __syncthreads();
    // ddyn[j][k] = Ht[j].dRpre[k] + dHpre[j].Z[k]
    for (int i = tid; i < HT * KK; i += CX_NT) {
        const int j = i / KK, k = i % KK;
        float acc = 0.f;
#pragma unroll 4
        for (int c4 = 0; c4 < K::LPR; ++c4) {
            acc += f4dot(ld4(Hs + j * K::PITCH + 4 * c4), ld4(Gs + k * K::PITCH + 4 * c4));
            acc += f4dot(ld4(dHs + j * K::PITCH + 4 * c4), ld4(Zs + k * K::PITCH + 4 * c4));
        }
        ddyn[(size_t)b * HT * KK + i] = acc;
    }
    // dS[k] += sum_j dyn[j][k] dHpre[j]
    for (int i = tid; i < KK * K::LPR; i += CX_NT) {
        const int k = i / K::LPR, c4 = i % K::LPR;
        float4 acc = f4zero();
#pragma unroll 8
        for (int j = 0; j < HT; ++j) acc = f4fma(dyns[j * KK + k], ld4(dHs + j * K::PITCH + 4 * c4), acc);
        const size_t off = ((size_t)b * KK + k) * C + 4 * c4;
        st4(dS + off, f4add(ld4(dS + off), acc));
    }
}

template <int C>
static int launch_cross(bool fwd, const float* a0, const float* a1, const float* a2, const float* a3, const float* a4, const float* a5,
                        float* o0, float* o1, float* o2, int B, int T, int HS, int HT, hipStream_t st) {
    const int KK = T * HS;
    using K = CrossCfg<C>;
    if (fwd) {
        const size_t smem = ((size_t)KK * K::PITCH + (size_t)HT * C + (size_t)HT * KK) * sizeof(float);
        if (smem > 160 * 1024) return GPTST_ESHAPE;
        static size_t cur = 0;
        if (smem > cur) { hipFuncSetAttribute((const void*)cap_cross_fwd_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); cur = smem; }
        hipLaunchKernelGGL((cap_cross_fwd_kernel<C>), dim3(B), dim3(CX_NT), smem, st, a0, a1, a2, o0, o1, o2, T, HS, HT);
    } else {
        const size_t smem = ((size_t)2 * KK * K::PITCH + (size_t)2 * HT * K::PITCH + (size_t)HT * KK) * sizeof(float);
        if (smem > 160 * 1024) return GPTST_ESHAPE;
        static size_t cur = 0;
        if (smem > cur) { hipFuncSetAttribute((const void*)cap_cross_bwd_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); cur = smem; }
        hipLaunchKernelGGL((cap_cross_bwd_kernel<C>), dim3(B), dim3(CX_NT), smem, st, a0, a1, a2, a3, a4, a5, o0, o1, T, HS, HT);
    }
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// ---- large token counts (T*HS*C no longer fits LDS, e.g. HS = 40 -> 480 tokens): plain global-memory kernels, same algebra -------
// Correctness path for BASELINE config 4's HS sweep; every kernel is a thread / wave per output with a short loop.
template <int C>
__global__ void cx_big_ht_kernel(const float* __restrict__ s, const float* __restrict__ dyn, const float* __restrict__ tmpl,
                                 float* __restrict__ Ht, int KK, int HS, int HT) {           // Ht[b,j,c] = LReLU(sum_k dyn[b,j,k] (s[b,k,c] + tm[k]))
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HT * C) return;
    const int j = i / C, c = i % C;
    float acc = 0.f;
#pragma unroll 8                      // eight tokens' loads in flight per trip (the plain loop is one dependent round trip per token)
    for (int k = 0; k < KK; ++k) acc = fmaf(dyn[((size_t)b * HT + j) * KK + k], s[((size_t)b * KK + k) * C + c] + tmpl[k / HS], acc);
    Ht[((size_t)b * HT + j) * C + c] = lrelu(acc);
}
template <int C>
__global__ void cx_big_v_kernel(const float* __restrict__ s, const float* __restrict__ dyn, const float* __restrict__ Ht,
                                float* __restrict__ v, float* __restrict__ Rt, int KK, int HT) {   // one wave per token row
    constexpr int E = C / 64;
    const int lane = threadIdx.x & 63, k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), b = blockIdx.y;
    if (k >= KK) return;
    float u[E], sq = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int c = lane + 64 * e;
        float acc = 0.f;
        for (int j = 0; j < HT; ++j) acc = fmaf(dyn[((size_t)b * HT + j) * KK + k], Ht[((size_t)b * HT + j) * C + c], acc);
        const float rt = lrelu(acc);
        Rt[((size_t)b * KK + k) * C + c] = rt;
        u[e] = rt + s[((size_t)b * KK + k) * C + c];
        sq = fmaf(u[e], u[e], sq);
    }
    const float sc = squash_scale(group_sum<64>(sq));
#pragma unroll
    for (int e = 0; e < E; ++e) v[((size_t)b * KK + k) * C + lane + 64 * e] = u[e] * sc;
}
template <int C>
__global__ void cx_big_du_kernel(const float* __restrict__ dv, const float* __restrict__ s, const float* __restrict__ Rt,
                                 float* __restrict__ dS, int KK) {                                 // dS = du = squash_bwd(Rt + s, dv)
    constexpr int E = C / 64;
    const int lane = threadIdx.x & 63, k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), b = blockIdx.y;
    if (k >= KK) return;
    float u[E], g[E], q = 0.f, udg = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const size_t off = ((size_t)b * KK + k) * C + lane + 64 * e;
        u[e] = Rt[off] + s[off]; g[e] = dv[off];
        q = fmaf(u[e], u[e], q); udg = fmaf(u[e], g[e], udg);
    }
    q = group_sum<64>(q); udg = group_sum<64>(udg);
    const float r = sqrtf(q), den = (1.f + q) * (r + 1e-8f);
    const float gq = q / den;
    float gp = 0.f;
    if (r > 0.f) gp = (den - q * ((r + 1e-8f) + (1.f + q) * 0.5f / r)) / (den * den);
    const float k2 = 2.f * gp * udg;
#pragma unroll
    for (int e = 0; e < E; ++e) dS[((size_t)b * KK + k) * C + lane + 64 * e] = fmaf(k2, u[e], gq * g[e]);
}
template <int C>
__global__ void cx_big_dh_kernel(const float* __restrict__ dS, const float* __restrict__ Rt, const float* __restrict__ Ht,
                                 const float* __restrict__ dyn, float* __restrict__ dH, int KK, int HT) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;       // dHpre[j,c] = lrelu'(Ht) sum_k dyn[j,k] du[k,c] lrelu'(Rt[k,c])
    if (i >= HT * C) return;
    const int j = i / C, c = i % C;
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < KK; ++k) {
        const size_t off = ((size_t)b * KK + k) * C + c;
        acc = fmaf(dyn[((size_t)b * HT + j) * KK + k], dS[off] * lrelu_grad_from_out(Rt[off]), acc);
    }
    dH[((size_t)b * HT + j) * C + c] = acc * lrelu_grad_from_out(Ht[((size_t)b * HT + j) * C + c]);
}
template <int C>
__global__ void cx_big_ddyn_kernel(const float* __restrict__ dS, const float* __restrict__ Rt, const float* __restrict__ Ht,
                                   const float* __restrict__ dH, const float* __restrict__ s, const float* __restrict__ tmpl,
                                   float* __restrict__ ddyn, int KK, int HS, int HT) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;       // ddyn[j,k] = Ht[j].dRpre[k] + dHpre[j].Z[k]
    if (i >= HT * KK) return;
    const int j = i / KK, k = i % KK;
    const float tm = tmpl[k / HS];
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
        const size_t off = ((size_t)b * KK + k) * C + c;
        acc = fmaf(Ht[((size_t)b * HT + j) * C + c], dS[off] * lrelu_grad_from_out(Rt[off]), acc);
        acc = fmaf(dH[((size_t)b * HT + j) * C + c], s[off] + tm, acc);
    }
    ddyn[(size_t)b * HT * KK + i] = acc;
}
template <int C>
__global__ void cx_big_ds_kernel(const float* __restrict__ dyn, const float* __restrict__ dH, float* __restrict__ dS, int KK, int HT) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;       // dS[k,c] += sum_j dyn[j,k] dHpre[j,c]   (LAST: dS held du until here)
    if (i >= KK * C) return;
    const int k = i / C, c = i % C;
    float acc = 0.f;
    for (int j = 0; j < HT; ++j) acc = fmaf(dyn[((size_t)b * HT + j) * KK + k], dH[((size_t)b * HT + j) * C + c], acc);
    dS[((size_t)b * KK + k) * C + c] += acc;
}

template <int C>
static int launch_cross_big(bool fwd, const float* a0, const float* a1, const float* a2, const float* a3, const float* a4, const float* a5,
                            float* o0, float* o1, float* o2, float* ws, int B, int T, int HS, int HT, hipStream_t st) {
    const int KK = T * HS;
    if (fwd) {       // (s, dyn, tmpl) -> v = o0, Ht = o1, Rt = o2
        hipLaunchKernelGGL((cx_big_ht_kernel<C>), dim3((HT * C + 255) / 256, B), dim3(256), 0, st, a0, a1, a2, o1, KK, HS, HT);
        hipLaunchKernelGGL((cx_big_v_kernel<C>), dim3((KK + 3) / 4, B), dim3(256), 0, st, a0, a1, (const float*)o1, o0, o2, KK, HT);
    } else {         // (dv, s, Rt, Ht, dyn, tmpl) -> dS = o0, ddyn = o1; ws = dHpre (B*HT*C)
        if (!ws) return GPTST_EARG;
        hipLaunchKernelGGL((cx_big_du_kernel<C>), dim3((KK + 3) / 4, B), dim3(256), 0, st, a0, a1, a2, o0, KK);
        hipLaunchKernelGGL((cx_big_dh_kernel<C>), dim3((HT * C + 255) / 256, B), dim3(256), 0, st, (const float*)o0, a2, a3, a4, ws, KK, HT);
        hipLaunchKernelGGL((cx_big_ddyn_kernel<C>), dim3((HT * KK + 255) / 256, B), dim3(256), 0, st, (const float*)o0, a2, a3, (const float*)ws, a1, a5, o1, KK, HS, HT);
        hipLaunchKernelGGL((cx_big_ds_kernel<C>), dim3((KK * C + 255) / 256, B), dim3(256), 0, st, a4, (const float*)ws, o0, KK, HT);
    }
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// floats of device scratch gptst_cap_cross_bwd needs at this shape (0 while the tokens fit LDS)
extern "C" int gptst_cap_cross_ws_floats(int B, int T, int C, int HS, int HT) {
    const size_t smem = ((size_t)2 * T * HS * (C + 4) + (size_t)2 * HT * (C + 4) + (size_t)HT * T * HS) * sizeof(float);
    return smem > 160 * 1024 ? B * HT * C : 0;
}

extern "C" int gptst_cap_cross_fwd(const float* s, const float* dyn, const float* tmpl, float* v, float* Ht, float* Rt, int B, int T,
                                   int C, int HS, int HT, void* stream) {
    if (!s || !dyn || !tmpl || !v || !Ht || !Rt) return GPTST_EARG;
    int rc = GPTST_ESHAPE;
    if (C == 64) rc = launch_cross<64>(true, s, dyn, tmpl, 0, 0, 0, v, Ht, Rt, B, T, HS, HT, (hipStream_t)stream);
    if (C == 128) rc = launch_cross<128>(true, s, dyn, tmpl, 0, 0, 0, v, Ht, Rt, B, T, HS, HT, (hipStream_t)stream);
    if (rc == GPTST_ESHAPE && C == 64) rc = launch_cross_big<64>(true, s, dyn, tmpl, 0, 0, 0, v, Ht, Rt, 0, B, T, HS, HT, (hipStream_t)stream);
    if (rc == GPTST_ESHAPE && C == 128) rc = launch_cross_big<128>(true, s, dyn, tmpl, 0, 0, 0, v, Ht, Rt, 0, B, T, HS, HT, (hipStream_t)stream);
    return rc;
}

extern "C" int gptst_cap_cross_bwd(const float* dv, const float* s, const float* Rt, const float* Ht, const float* dyn,
                                   const float* tmpl, float* dS, float* ddyn, float* ws, int B, int T, int C, int HS, int HT, void* stream) {
    if (!dv || !s || !Rt || !Ht || !dyn || !tmpl || !dS || !ddyn) return GPTST_EARG;
    int rc = GPTST_ESHAPE;
    if (C == 64) rc = launch_cross<64>(false, dv, s, Rt, Ht, dyn, tmpl, dS, ddyn, 0, B, T, HS, HT, (hipStream_t)stream);
    if (C == 128) rc = launch_cross<128>(false, dv, s, Rt, Ht, dyn, tmpl, dS, ddyn, 0, B, T, HS, HT, (hipStream_t)stream);
    if (rc == GPTST_ESHAPE && C == 64) rc = launch_cross_big<64>(false, dv, s, Rt, Ht, dyn, tmpl, dS, ddyn, 0, ws, B, T, HS, HT, (hipStream_t)stream);
    if (rc == GPTST_ESHAPE && C == 128) rc = launch_cross_big<128>(false, dv, s, Rt, Ht, dyn, tmpl, dS, ddyn, 0, ws, B, T, HS, HT, (hipStream_t)stream);
    return rc;
}
