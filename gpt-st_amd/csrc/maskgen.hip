// Mask generation of the masked autoencoder (reference GPTST.py:314-323 random phase, :344-413 adaptive phase).
//
// The reference zeroes the k largest noise values with  sort(descending) -> idx[:k] -> scatter_(0)  (one global sort per
// selection) and picks whole cluster classes in a host `while` loop with a device->host sync per iteration (:366-369).
// Here every selection is a radix SELECT (k-th largest by exact float bits, 4 x 8-bit digits, histograms in LDS) followed by
// an index-ordered pass, so the result is the same SET as the sort-based one whenever the k-th and (k+1)-th values differ;
// ties straddling rank k go to the lowest indices (torch's sort is unstable there).  Class selection runs on the
// device from a per-class histogram: nothing on this path reads back to the host, so it is hipGraph-capturable.
// Noise is non-negative (uniform [0,1)), so uint32 bit order == float order.
// Masks are fp32 {0,1} arrays (1 = visible, 0 = masked) of B*T*N*base cells — the consumers multiply with them.
#include "common.h"

#define MG_THREADS 1024
#define MG_WAVES (MG_THREADS / 64)

struct MgShared {
    unsigned hist[256];
    unsigned scan[MG_THREADS];
    unsigned prefix, remaining;
};

// Zero, in `mask`, the k largest values val(i), i < M (ties at the boundary: lowest index first).  All threads must call.
template <class F>
__device__ void block_drop_topk(F val, int M, int k, float* __restrict__ mask, MgShared& sh) {
    if (k <= 0) return;                     // uniform
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) { sh.prefix = 0u; sh.remaining = (unsigned)k; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 256; i += MG_THREADS) sh.hist[i] = 0u;
        __syncthreads();
        const unsigned prefix = sh.prefix;
        for (int i0 = 0; i0 < M; i0 += MG_THREADS) {
            const int i = i0 + tid;
            bool act = i < M;
            unsigned key = 0u;
            if (act) {
                key = __float_as_uint(val(i));
                if (pass > 0) act = (key >> (shift + 8)) == (prefix >> (shift + 8));
            }
            const unsigned bin = (key >> shift) & 255u;
            if (pass == 0) {
                // the top digit of uniform noise is extremely skewed (half of [0,1) shares one exponent): aggregate per wave
                unsigned long long todo = __ballot(act);
                while (todo) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const unsigned b = __shfl(bin, leader, 64);
                    const unsigned long long m = __ballot(act && bin == b);
                    if (lane == leader) atomicAdd(&sh.hist[b], (unsigned)__popcll(m));
                    todo &= ~m;
                }
            } else if (act) {
                atomicAdd(&sh.hist[bin], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) {
            unsigned rem = sh.remaining, cum = 0u;
            int b = 255;
            for (; b > 0; --b) {
                if (cum + sh.hist[b] >= rem) break;
                cum += sh.hist[b];
            }
            sh.prefix = prefix | ((unsigned)b << shift);
            sh.remaining = rem - cum;
        }
        __syncthreads();
    }
    const unsigned thr = sh.prefix;
    const unsigned need = sh.remaining;          // how many elements equal to thr are dropped (lowest index first)
    const int chunk = (M + MG_THREADS - 1) / MG_THREADS;
    const int lo = min(M, tid * chunk), hi = min(M, lo + chunk);
    unsigned cnt = 0u;
    for (int i = lo; i < hi; ++i) cnt += (__float_as_uint(val(i)) == thr) ? 1u : 0u;
    sh.scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < MG_THREADS; off <<= 1) {          // Hillis-Steele inclusive scan
        const unsigned v = (tid >= off) ? sh.scan[tid - off] : 0u;
        __syncthreads();
        sh.scan[tid] += v;
        __syncthreads();
    }
    unsigned rank = sh.scan[tid] - cnt;
    for (int i = lo; i < hi; ++i) {
        const unsigned key = __float_as_uint(val(i));
        if (key > thr) mask[i] = 0.f;
        else if (key == thr) { if (rank < need) mask[i] = 0.f; ++rank; }
    }
    __syncthreads();
}

// random phase: mask = ones with the k largest noise cells zeroed (M = B*T*N*base)
__global__ __launch_bounds__(MG_THREADS) void mask_random_kernel(const float* __restrict__ noise, int M, int k,
                                                                 float* __restrict__ mask) {
    __shared__ MgShared sh;
    for (int i = threadIdx.x; i < M; i += MG_THREADS) mask[i] = 1.f;
    __syncthreads();
    block_drop_topk([&](int i) { return noise[i]; }, M, k, mask, sh);
}

// label[i] = argmax_h prob[i, h] (first maximum), counts[h] += 1      (GPTST.py:344-345)
__global__ __launch_bounds__(256) void mask_labels_kernel(const float* __restrict__ prob, int rows, int HS,
                                                          int* __restrict__ label, int* __restrict__ counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    const float* p = prob + (size_t)i * HS;
    float best = p[0];
    int bi = 0;
    for (int h = 1; h < HS; ++h) { const float v = p[h]; if (v > best) { best = v; bi = h; } }
    label[i] = bi;
    atomicAdd(counts + bi, 1);
}

// adaptive phase (GPTST.py:356-413).  nums = {adaptive_mask_num, random_mask_num} (device ints, they change with the epoch);
// list_c = shuffled class order.  Writes mask (M cells x base channels, repeated over base) and, if given, the two partial masks.
__global__ __launch_bounds__(MG_THREADS) void mask_adaptive_kernel(const int* __restrict__ label, const int* __restrict__ counts,
                                                                   const int* __restrict__ list_c, const int* __restrict__ nums,
                                                                   const float* __restrict__ noise_a, const float* __restrict__ noise_r,
                                                                   int ada_all, int M, int HS, int base, float* __restrict__ m_ada,
                                                                   float* __restrict__ m_rnd, float* __restrict__ mask) {
    __shared__ MgShared sh;
    __shared__ unsigned char cls_d[256], cls_f[256];
    __shared__ int s_ka;
    const int tid = threadIdx.x;
    for (int h = tid; h < 256; h += MG_THREADS) { cls_d[h] = 0; cls_f[h] = 0; }
    __syncthreads();
    const int ada_num = nums[0], rnd_num = nums[1];
    if (tid == 0) {
        int num = 0, i = 0;
        while (num < ada_num && i < HS) { num += counts[list_c[i]]; ++i; }      // :366-369 / :379-382
        int dnum = 0;
        if (ada_all && i >= 2) {                                                // :370-374
            for (int k = 0; k < i - 1; ++k) { cls_d[list_c[k]] = 1; dnum += counts[list_c[k]]; }
            cls_f[list_c[i - 1]] = 1;
        } else {                                                                // :375-377 / :383-384
            for (int k = 0; k < i; ++k) cls_f[list_c[k]] = 1;
        }
        s_ka = ada_num - dnum;                                                  // :393
    }
    for (int i = tid; i < M; i += MG_THREADS) m_ada[i] = 1.f;
    __syncthreads();
    block_drop_topk([&](int i) { return cls_f[label[i]] ? noise_a[i] : 0.f; }, M, s_ka, m_ada, sh);      // :390-396
    for (int i = tid; i < M; i += MG_THREADS) {
        if (cls_d[label[i]]) m_ada[i] = 0.f;                                    // :397
        m_rnd[i] = 1.f;
    }
    __syncthreads();
    block_drop_topk([&](int i) { return m_ada[i] != 0.f ? noise_r[i] : 0.f; }, M, rnd_num, m_rnd, sh);    // :401-406
    for (int i = tid; i < M; i += MG_THREADS) {
        const float f = m_ada[i] * m_rnd[i];                                    // :411
        for (int j = 0; j < base; ++j) mask[(size_t)i * base + j] = f;          // :412-413
    }
}

extern "C" int gptst_mask_random(const float* noise, int M, int k, float* mask, void* stream) {
    if (!noise || !mask || M <= 0 || k < 0 || k > M) return GPTST_EARG;
    hipLaunchKernelGGL(mask_random_kernel, dim3(1), dim3(MG_THREADS), 0, (hipStream_t)stream, noise, M, k, mask);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_mask_labels(const float* prob, int rows, int HS, int* label, int* counts, void* stream) {
    if (!prob || !label || !counts || HS <= 0 || HS > 256) return GPTST_EARG;
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * HS, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mask_labels_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, prob, rows, HS, label, counts);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_mask_adaptive(const int* label, const int* counts, const int* list_c, const int* nums, const float* noise_a,
                                   const float* noise_r, int ada_all, int M, int HS, int base, float* m_ada, float* m_rnd,
                                   float* mask, void* stream) {
    if (!label || !counts || !list_c || !nums || !noise_a || !noise_r || !m_ada || !m_rnd || !mask || HS > 256) return GPTST_EARG;
    hipLaunchKernelGGL(mask_adaptive_kernel, dim3(1), dim3(MG_THREADS), 0, (hipStream_t)stream, label, counts, list_c, nums, noise_a,
                       noise_r, ada_all, M, HS, base, m_ada, m_rnd, mask);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
