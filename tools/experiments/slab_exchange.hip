// Micro-benchmark for the per-sample exchange of the slab design (round 4): a sample's NTILE workgroups ((b, node tile) slabs, all on one
// XCD by the dispatch-order work map) sum their partial cluster aggregates S (T*HS*C floats ~ 30 KB) inside ONE launch.
//   variant 0  all-read:        sc1 partial stores -> counter -> every workgroup reads all NTILE partials (sc1 loads) and sums in index order
//   variant 1  reduce-scatter:  sc1 partial stores -> counter -> workgroup i sums slice i of the NTILE partials -> sc1 store -> counter 2 ->
//                               every workgroup reads the reduced vector
//   variant 2  as 0 with plain stores + agent release fence / agent acquire fence + plain loads
// Between exchanges every workgroup does `work` dependent MFMAs per wave and streams `stream_kb` KB of plain stores (the saved activations
// of the real kernel), so that the hand-off is measured under load.  Every received word is checked.
//   hipcc --offload-arch=gfx950 -O3 -o slab_exchange slab_exchange.hip && ./slab_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define NTH 256
#define PER_THREAD 8                       // float4 per thread -> 8192 floats = 32 KB per partial
#define NF4 (NTH * PER_THREAD)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, int byteoff) {
    i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byteoff, 0, 16);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, int byteoff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, byteoff, 0, 16);
}
__device__ __forceinline__ void wait_count(unsigned* cnt, unsigned want, unsigned* tmo) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ float pval(int b, int tile, int ph, int i) { return (float)((b * 31 + tile * 7 + ph * 3 + i) % 61) * 0.25f; }

__global__ __launch_bounds__(NTH, 2) void exch_kernel(float* part, float* red, unsigned* cnt, unsigned* tmo, float* sink, long long* ts,
                                                      unsigned* bad, int B, int ntile, int phases, int variant, int work, int stream_f4) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    const int b = xcd + 8 * (slot / ntile), tile = slot % ntile;
    if (b >= B) return;
    const int tid = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned nbad = 0;
    for (int ph = 0; ph < phases; ++ph) {
        if (L == 0 && tid == 0) ts[3 * ph] = wall_clock64();
        // ---- "layer work": dependent MFMAs + streaming plain stores ----
        float x = tid * 0.001f, y = 1.f + ph;
        for (int i = 0; i < work; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);
        for (int i = tid; i < stream_f4; i += NTH)
            reinterpret_cast<f32x4*>(sink)[((size_t)L * phases + ph) * stream_f4 + i] = acc;
        if (L == 0 && tid == 0) ts[3 * ph + 1] = wall_clock64();
        // ---- publish the partial ----
        float* mypart = part + ((size_t)(ph * B + b) * ntile + tile) * NF4 * 4;
        const float* spart = part + (size_t)(ph * B + b) * ntile * NF4 * 4;
        unsigned* c1 = cnt + (ph * B + b) * 2, *c2 = c1 + 1;
        if (variant == 2) {
#pragma unroll
            for (int k = 0; k < PER_THREAD; ++k) {
                const int f = tid + k * NTH;
                f32x4 v = {pval(b, tile, ph, 4 * f), pval(b, tile, ph, 4 * f + 1), pval(b, tile, ph, 4 * f + 2), pval(b, tile, ph, 4 * f + 3)};
                reinterpret_cast<f32x4*>(mypart)[f] = v;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            __amdgpu_buffer_rsrc_t r = rsrc(mypart);
#pragma unroll
            for (int k = 0; k < PER_THREAD; ++k) {
                const int f = tid + k * NTH;
                f32x4 v = {pval(b, tile, ph, 4 * f), pval(b, tile, ph, 4 * f + 1), pval(b, tile, ph, 4 * f + 2), pval(b, tile, ph, 4 * f + 3)};
                st_sc1(r, f * 16, v);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        wait_count(c1, ntile, tmo);
        f32x4 s[PER_THREAD];
        if (variant == 0 || variant == 2) {
            if (variant == 2) { if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __syncthreads(); }
#pragma unroll
            for (int k = 0; k < PER_THREAD; ++k) s[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int t2 = 0; t2 < ntile; t2 += 2) {                   // 16 loads in flight
                f32x4 v[2][PER_THREAD];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int tt = min(t2 + u, ntile - 1);
                    if (variant == 2) {
#pragma unroll
                        for (int k = 0; k < PER_THREAD; ++k) v[u][k] = reinterpret_cast<const f32x4*>(spart + (size_t)tt * NF4 * 4)[tid + k * NTH];
                    } else {
                        __amdgpu_buffer_rsrc_t r = rsrc(spart + (size_t)tt * NF4 * 4);
#pragma unroll
                        for (int k = 0; k < PER_THREAD; ++k) v[u][k] = ld_sc1(r, (tid + k * NTH) * 16);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (t2 + u < ntile)
#pragma unroll
                        for (int k = 0; k < PER_THREAD; ++k) s[k] += v[u][k];
            }
        } else {
            // reduce-scatter: slice = NF4 / ntile float4 (ntile 16 -> 128), thread f < slice sums the ntile partials
            const int slice = NF4 / ntile;
            if (tid < slice) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                f32x4 v[16];
#pragma unroll
                for (int t2 = 0; t2 < 16; ++t2) {
                    __amdgpu_buffer_rsrc_t r = rsrc(spart + (size_t)min(t2, ntile - 1) * NF4 * 4);
                    v[t2] = ld_sc1(r, (tile * slice + tid) * 16);
                }
#pragma unroll
                for (int t2 = 0; t2 < 16; ++t2) if (t2 < ntile) a += v[t2];
                __amdgpu_buffer_rsrc_t rr = rsrc(red + (size_t)(ph * B + b) * NF4 * 4);
                st_sc1(rr, (tile * slice + tid) * 16, a);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(c2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wait_count(c2, ntile, tmo);
            __amdgpu_buffer_rsrc_t rr = rsrc(red + (size_t)(ph * B + b) * NF4 * 4);
#pragma unroll
            for (int k = 0; k < PER_THREAD; ++k) s[k] = ld_sc1(rr, (tid + k * NTH) * 16);
        }
        // ---- check ----
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) {
            const int f = tid + k * NTH;
            for (int e = 0; e < 4; ++e) {
                float want = 0.f;
                for (int t2 = 0; t2 < ntile; ++t2) want += pval(b, t2, ph, 4 * f + e);
                if (s[k][e] != want) ++nbad;
            }
            acc += s[k];
        }
        if (L == 0 && tid == 0) ts[3 * ph + 2] = wall_clock64();
    }
    smem[tid] = acc[0];
    if (nbad) atomicAdd(bad, nbad);
    if (acc[0] == 12345.678f) sink[0] = smem[(tid + 1) % NTH];
}

int main(int argc, char** argv) {
    const int B = 32, phases = 6;
    int ntile = argc > 1 ? atoi(argv[1]) : 16;
    const int nwg = 8 * ((B + 7) / 8) * ntile;
    float *part, *red, *sink; unsigned *cnt, *tmo, *bad; long long* ts;
    const size_t stream_max = 1024;                       // float4 per workgroup and phase (16 KB) at most 4096 -> 64 KB
    CK(hipMalloc(&part, (size_t)phases * B * ntile * NF4 * 16));
    CK(hipMalloc(&red, (size_t)phases * B * NF4 * 16));
    CK(hipMalloc(&sink, (size_t)nwg * phases * 4096 * 16));
    CK(hipMalloc(&cnt, phases * B * 2 * 4)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&ts, 3 * phases * 8));
    CK(hipFuncSetAttribute((const void*)exch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)exch_kernel, NTH, 70 * 1024));
    printf("ntile %d, %d workgroups, occupancy API %d per CU\n", ntile, nwg, occ);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    (void)stream_max;
    for (int variant = 0; variant < 3; ++variant)
        for (int work : {0, 400})
            for (int skb : {0, 48}) {
                const int stream_f4 = skb * 1024 / 16;
                float best = 1e9f, sum = 0.f; unsigned hb = 0, ht = 0;
                long long hts[3 * phases];
                const int reps = 12;
                for (int it = 0; it < reps; ++it) {
                    CK(hipMemsetAsync(cnt, 0, phases * B * 2 * 4)); CK(hipMemsetAsync(tmo, 0, 4)); CK(hipMemsetAsync(bad, 0, 4));
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(exch_kernel, dim3(nwg), dim3(NTH), 70 * 1024, 0, part, red, cnt, tmo, sink, ts, bad, B, ntile, phases, variant, work, stream_f4);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (it >= 2) { best = ms < best ? ms : best; sum += ms; }
                    unsigned b1, t1; CK(hipMemcpy(&b1, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&t1, tmo, 4, hipMemcpyDeviceToHost));
                    hb += b1; ht += t1;
                }
                CK(hipMemcpy(hts, ts, sizeof(hts), hipMemcpyDeviceToHost));
                printf("variant %d work %3d stream %2d KB: %7.1f us best, %7.1f avg for %d phases | wg0 phase (work, exchange) x100MHz ticks:", variant, work, skb,
                       best * 1e3f, sum / (reps - 2) * 1e3f, phases);
                for (int ph = 0; ph < phases; ++ph) printf(" (%lld,%lld)", hts[3 * ph + 1] - hts[3 * ph], hts[3 * ph + 2] - hts[3 * ph + 1]);
                printf(" | bad %u timeouts %u\n", hb, ht);
            }
    return 0;
}
