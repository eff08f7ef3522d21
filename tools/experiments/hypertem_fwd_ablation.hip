// ablation of the hyperTem forward chain: which phase costs what (standalone, hipEvents, graph-free back-to-back launches)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define HT_T 12
#define SB() __builtin_amdgcn_sched_barrier(0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4_wt(float* p, float4 v) {        // 16-byte write-through (sc1) store
    f32x4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
    f32x4 x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 c) { return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w)); }
__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : 0.01f * x; }
struct HtFrag { float4 bv[4][4]; float4 b4; };
template <int F>
__device__ __forceinline__ void ht_load_frag(HtFrag& f, const float* __restrict__ Wbt, const float* __restrict__ bbt, size_t g, int j, int kk) {
    const float* W_ = Wbt + g * 64 * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) f.bv[q][e] = (F & 1) ? ld4(W_ + (size_t)(16 * q + 4 * kk + e) * 64 + 4 * j) : make_float4(0.01f * q, 0.02f * e, 0.03f, 0.04f);
    f.b4 = (F & 1) ? ld4(bbt + g * 64 + 4 * j) : f4zero();
}
// F bits: 1 = W loads, 2 = slab load from global, 4 = mix, 8 = MFMA, 16 = out stores, 32 = R stores
template <int F>
__global__ __launch_bounds__(256, 2) void fwd(const float* __restrict__ X, const float* __restrict__ G, const float* __restrict__ Wbt,
                                               const float* __restrict__ bbt, float* __restrict__ R_out, float* __restrict__ out, int N, int B) {
    constexpr int C = 64, P = C + 4, GP = 145, NT = 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* Gs = Xs + HT_T * NT * P;
    const int ntiles = (N + NT - 1) / NT;
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    const int b = xcd + 8 * (slot / ntiles), tile = slot % ntiles;
    if (b >= B) return;
    const int n0 = tile * NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    HtFrag f0, f1;
    ht_load_frag<F>(f0, Wbt, bbt, (size_t)b * HT_T + wave, j, kk);
    {
        const int nl = tid >> 4, c4 = tid & 15;
        const int n = min(n0 + nl, N - 1);
        float4 v[HT_T];
        float gv[9];
#pragma unroll
        for (int t = 0; t < HT_T; ++t) v[t] = (F & 2) ? ld4(X + (((size_t)b * HT_T + t) * N + n) * C + 4 * c4) : make_float4(0.1f * t, 0.2f, 0.3f, 0.4f * c4);
#pragma unroll
        for (int k = 0; k < 9; ++k) gv[k] = (F & 2) ? G[min(n0 * 144 + tid + k * 256, N * 144 - 1)] : 0.01f * k;
        SB();
#pragma unroll
        for (int t = 0; t < HT_T; ++t) st4(Xs + (t * NT + nl) * P + 4 * c4, n0 + nl < N ? v[t] : f4zero());
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int i = tid + k * 256;
            Gs[(i / 144) * GP + i % 144] = (n0 + i / 144 < N) ? gv[k] : 0.f;
        }
    }
    __syncthreads();
    SB();
    auto step = [&](int t, const HtFrag& fc, HtFrag& fn, bool more) {
        const size_t g = (size_t)b * HT_T + t;
        float4 a4[C / 16];
#pragma unroll
        for (int q = 0; q < C / 16; ++q) a4[q] = f4zero();
        if (F & 4) {
            const float* gr = Gs + j * GP + t * HT_T;
            const float* xr = Xs + j * P + 4 * kk;
#pragma unroll
            for (int u = 0; u < HT_T; ++u) {
                const float gu = gr[u];
#pragma unroll
                for (int q = 0; q < C / 16; ++q) a4[q] = f4fma(gu, ld4(xr + u * NT * P + 16 * q), a4[q]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < C / 16; ++q) a4[q] = ld4(Xs + (t * NT + j) * P + 16 * q + 4 * kk);
        }
        SB();
        if (more) ht_load_frag<F>(fn, Wbt, bbt, g + 4, j, kk);
        SB();
        f32x4 acc[C / 16];
#pragma unroll
        for (int ct = 0; ct < C / 16; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (F & 8) {
#pragma unroll
            for (int q = 0; q < C / 16; ++q) {
                const float av[4] = {a4[q].x, a4[q].y, a4[q].z, a4[q].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], fc.bv[q][e].x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], fc.bv[q][e].y, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], fc.bv[q][e].z, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], fc.bv[q][e].w, acc[3], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < C / 16; ++q) { acc[q][0] = a4[q].x + fc.bv[q][0].x; acc[q][1] = a4[q].y + fc.bv[q][1].y; acc[q][2] = a4[q].z + fc.bv[q][2].z; acc[q][3] = a4[q].w + fc.bv[q][3].w; }
        }
        SB();
        if ((F & 32) && n0 + j < N) {
#pragma unroll
            for (int q = 0; q < C / 16; ++q) { float* p_ = R_out + (g * N + n0 + j) * C + 16 * q + 4 * kk; if (F & 64) st4_wt(p_, a4[q]); else if (F & 128) st4_nt(p_, a4[q]); else st4(p_, a4[q]); }
        }
        float4 keep = f4zero();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nl = kk * 4 + r;
            if (n0 + nl < N) {
                float4 y = f4add(f4add(make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), fc.b4), ld4(Xs + (t * NT + nl) * P + 4 * j));
                y.x = lrelu(y.x); y.y = lrelu(y.y); y.z = lrelu(y.z); y.w = lrelu(y.w);
                if (F & 16) { float* p_ = out + (g * N + n0 + nl) * C + 4 * j; if (F & 64) st4_wt(p_, y); else if (F & 128) st4_nt(p_, y); else st4(p_, y); } else keep = f4add(keep, y);
            }
        }
        if (!(F & 16) && keep.x == 123.456f) st4(out + g * 64, keep);      // keep the arithmetic alive
        SB();
    };
    step(wave, f0, f1, true);
    step(wave + 4, f1, f0, true);
    step(wave + 8, f0, f1, false);
}
template <int F>
float run(const float* X, const float* G, const float* W, const float* bb, float* R, float* out, int N, int B) {
    const size_t smem = (12 * 16 * 68 + 16 * 145) * 4;
    hipFuncSetAttribute((const void*)fwd<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int grid = 8 * ((B + 7) / 8) * ((N + 15) / 16);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fwd<F>, dim3(grid), dim3(256), smem, 0, X, G, W, bb, R, out, N, B);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(fwd<F>, dim3(grid), dim3(256), smem, 0, X, G, W, bb, R, out, N, B);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1000.f / 50;
}
int main() {
    const int N = 170, C = 64;
    for (int B : {8, 32}) {
        const size_t n = (size_t)B * 12 * N * C;
        float *X, *G, *W, *bb, *R, *out;
        hipMalloc(&X, n * 4); hipMalloc(&R, n * 4); hipMalloc(&out, n * 4); hipMalloc(&G, N * 144 * 4); hipMalloc(&W, (size_t)B * 12 * 4096 * 4); hipMalloc(&bb, B * 12 * 64 * 4);
        hipMemset(X, 0, n * 4); hipMemset(G, 0, N * 144 * 4); hipMemset(W, 0, (size_t)B * 12 * 4096 * 4); hipMemset(bb, 0, B * 12 * 64 * 4);
        printf("B=%d: full plain %.1f | sc1 write-through %.1f | nt %.1f | noR plain %.1f | noR sc1 %.1f | noR nt %.1f\n", B, run<63>(X, G, W, bb, R, out, N, B), run<63 + 64>(X, G, W, bb, R, out, N, B),
               run<63 + 128>(X, G, W, bb, R, out, N, B), run<31>(X, G, W, bb, R, out, N, B), run<31 + 64>(X, G, W, bb, R, out, N, B), run<31 + 128>(X, G, W, bb, R, out, N, B));
        printf("B=%d: full %.1f | noR %.1f | no stores %.1f | no W loads %.1f | no slab load %.1f | no mix %.1f | no MFMA %.1f | only loads+LDS (no mix/mfma/stores) %.1f | nothing from memory, compute only %.1f | empty %.1f\n", B,
               run<63>(X, G, W, bb, R, out, N, B), run<31>(X, G, W, bb, R, out, N, B), run<15>(X, G, W, bb, R, out, N, B), run<62>(X, G, W, bb, R, out, N, B),
               run<61>(X, G, W, bb, R, out, N, B), run<59>(X, G, W, bb, R, out, N, B), run<55>(X, G, W, bb, R, out, N, B), run<3>(X, G, W, bb, R, out, N, B),
               run<12>(X, G, W, bb, R, out, N, B), run<0>(X, G, W, bb, R, out, N, B));
        hipFree(X); hipFree(R); hipFree(out); hipFree(G); hipFree(W); hipFree(bb);
    }
    return 0;
}
