// Pieces of apply.hip shared with hypertem.hip: the row mapping of the three grouping modes and the body of the C = 64 weight-gradient
// kernel (so that it can run side by side with the hyperTem backward inside ONE launch, hypertem.hip).
#pragma once
#include "common.h"

enum { PRO_NONE = 0, PRO_DPRE = 1 };          // PRO_DPRE: a = A * lrelu'(A2)   (A = dOut, A2 = layer output)
enum { EPI_PLAIN = 0, EPI_RES_LRELU = 1, EPI_ADD_DPRE = 2, EPI_LRELU = 3, EPI_ADD_PREMUL = 4, EPI_PREMUL = 5 };   // 1: lrelu(acc+bias+resid)  2: acc + resid*lrelu'(resid2)  3: lrelu(acc+bias)  4 / 5 (apply128, r05): (acc + resid)*lrelu'(resid2) / acc*lrelu'(resid2)

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


#define WGRAD64_SMEM_FLOATS (4 * 64 * 64 + 4 * 64)

// C = 64 weight gradient on fp32 MFMA 16x16x4, operands straight from global memory as float4 (see apply.hip for the layout notes).
// Body of wgrad64_kernel for group g, row split sp (of nsplit_groups = rm.G groups per split); smem: 4*C*C + 4*C floats.
// ACQ: D was produced by other workgroups of THIS launch (write-through stores + a flag the caller has waited for): agent-scope loads.
template <int PRO, int U, bool ACQ = false>
__device__ __forceinline__ void wgrad64_body(const float* __restrict__ A, const float* __restrict__ D, const float* __restrict__ D2,
                                             float* __restrict__ dW, RowMap rm, int rows_per_split, int ostride, int csa, int g, int sp,
                                             float* __restrict__ smem) {
    constexpr int C = 64;
    float (*red)[C * C] = reinterpret_cast<float (*)[C * C]>(smem);
    float (*csred)[C] = reinterpret_cast<float (*)[C]>(smem + 4 * C * C);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int mbeg0 = sp * rows_per_split;
    const int mend0 = min(rm.M, mbeg0 + rows_per_split);
    const int q = (mend0 - mbeg0 + 3) / 4;
    const int mbeg = mbeg0 + wave * q, mend = min(mend0, mbeg + q);
    f32x4 acc[4][4];
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[ca][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 sa = f4zero();
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(D), 0, ACQ ? (int)((size_t)rm.G * rm.M * C * 4) : 0, 0x00020000);
    // U = k-steps (of 4 rows) per batch of loads (2-3 float4 per step in flight per lane); the launcher picks 4 or 6 so that the
    // wave's step count divides evenly (measured: TIME 11 steps 13.5 us either way, NODE 6 steps 13.7 vs 16.8, SHARED 16 steps 10.5 vs 13.0).
    // The scheduler fence keeps the whole batch of loads ahead of the MFMA block (18.0 -> 15.1 us on the TIME weight gradient); a
    // ping-pong pair of register buffers (loads of batch i+1 behind the MFMAs of batch i) was slower again: 251 VGPRs, 15.9 us.
    for (int m0 = mbeg; m0 < mend; m0 += 4 * U) {
        float4 a[U], d[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = min(m0 + 4 * u + kk, mend - 1);                        // clamped: out-of-range rows are zeroed below
            const size_t off = ((size_t)g * rm.rs_g + (size_t)m * rm.rs_m) * C + 4 * j;
            a[u] = ld4(A + off); d[u] = ACQ ? ld4_sc1(rsD, (int)(off * 4)) : ld4(D + off);
            if (PRO == PRO_DPRE) y[u] = ld4(D2 + off);
        }
        SB();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (m0 + 4 * u + kk >= mend) a[u] = f4zero();
            if (PRO == PRO_DPRE) {
                // explicitly rounded products / sums: this body is compiled into two kernels (apply.hip, hypertem.hip) and must not depend on
                // where the compiler chooses to contract a multiply into the column-sum add (the results are compared bit for bit)
                d[u].x = __fmul_rn(d[u].x, lrelu_grad_from_out(y[u].x)); d[u].y = __fmul_rn(d[u].y, lrelu_grad_from_out(y[u].y));
                d[u].z = __fmul_rn(d[u].z, lrelu_grad_from_out(y[u].z)); d[u].w = __fmul_rn(d[u].w, lrelu_grad_from_out(y[u].w));
            }
            if (csa == 2) {                                                          // column sums of pro(D): the bias gradient
                if (m0 + 4 * u + kk < mend) sa = make_float4(__fadd_rn(sa.x, d[u].x), __fadd_rn(sa.y, d[u].y), __fadd_rn(sa.z, d[u].z), __fadd_rn(sa.w, d[u].w));
            }
            else sa = f4add(sa, a[u]);
            const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, dv[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
#pragma unroll
            for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ca], dv[cb], acc[ca][cb], 0, 0, 0);
        }
    }
    // D reg r of tile (ca, cb): dW row 4*(kk*4 + r) + ca, columns 4j + cb
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            st4(&red[wave][(4 * (kk * 4 + r) + ca) * C + 4 * j], make_float4(acc[ca][0][r], acc[ca][1][r], acc[ca][2][r], acc[ca][3][r]));
    if (csa) {                                   // column sums of A (bias gradient of a shared Linear), folded over row slots and waves
        sa.x += __shfl_xor(sa.x, 16, 64); sa.y += __shfl_xor(sa.y, 16, 64); sa.z += __shfl_xor(sa.z, 16, 64); sa.w += __shfl_xor(sa.w, 16, 64);
        sa.x += __shfl_xor(sa.x, 32, 64); sa.y += __shfl_xor(sa.y, 32, 64); sa.z += __shfl_xor(sa.z, 32, 64); sa.w += __shfl_xor(sa.w, 32, 64);
        if (kk == 0) st4(&csred[wave][4 * j], sa);
    }
    __syncthreads();
    float* o = dW + ((size_t)sp * rm.G + g) * (size_t)ostride;
    if (csa && threadIdx.x < C) o[C * C + threadIdx.x] = csred[0][threadIdx.x] + csred[1][threadIdx.x] + csred[2][threadIdx.x] + csred[3][threadIdx.x];
#pragma unroll
    for (int k = 0; k < C * C / 4 / 256; ++k) {
        const int f = threadIdx.x + k * 256;
        const float4 s = f4add(f4add(ld4(&red[0][4 * f]), ld4(&red[1][4 * f])), f4add(ld4(&red[2][4 * f]), ld4(&red[3][4 * f])));
        st4(o + 4 * f, s);
    }
}

