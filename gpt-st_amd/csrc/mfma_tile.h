// fp32 MFMA tile primitives (v_mfma_f32_32x32x2_f32) for the C x C "customised parameter" contractions.
//
// One wave computes a 32-row x C-col output tile  D = A(32 x C) * W(C x C):
//   * the A tile lives in a wave-private LDS region, row-major with pitch C+4 floats (conflict-free
//     ds_read_b128 for the operand fetch, 16-byte aligned rows for coalesced float4 staging);
//   * W lives in LDS as [k][C] (shared by the workgroup's waves);
//   * K is walked in a permuted order so that each lane fetches its A operands as one float4 per four
//     MFMAs: MFMA step s = 4q+jj uses k = 8q + 4h + jj for lane-half h = lane>>5 (A and B agree).
// Operand / result maps (cdna_hip_programming.md §3):  A: lane l holds A[i=l&31][k=l>>5];
//   B: lane l holds B[k=l>>5][j=l&31];  D reg r: col j = l&31, row i = (r&3) + 8*(r>>2) + 4*(l>>5).
#pragma once
#include "common.h"

template <int C>
struct Tile {
    static constexpr int PITCH = C + 4;
    static constexpr int NCT = C / 32;           // 32-wide column tiles
    static constexpr int F4_PER_ROW = C / 4;
    static constexpr int TILE_FLOATS = 32 * PITCH;
    static constexpr int F4_PER_LANE = 32 * F4_PER_ROW / 64;   // float4 slots each lane stages
};

// D(32 x C) = tile(32 x C) * Wl(C x C); acc[ct] holds column tile ct.
template <int C>
__device__ __forceinline__ void mfma_tile(const float* __restrict__ tile, const float* __restrict__ Wl,
                                          f32x16 (&acc)[Tile<C>::NCT], int lane) {
    constexpr int P = Tile<C>::PITCH;
    constexpr int NCT = Tile<C>::NCT;
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    const float* arow = tile + i * P + 4 * h;
    const float* wcol = Wl + 4 * h * C + i;
    // Software pipeline: the operands of step q+1 are fetched from LDS before the MFMAs of step q issue — the first
    // version (read -> s_waitcnt -> mfma per step) was LDS-latency bound at ~3.5x the MFMA time (profiles/r01b).
    float4 a_cur = ld4(arow);
    float b_cur[4][NCT];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) b_cur[jj][ct] = wcol[jj * C + ct * 32];
#pragma unroll
    for (int q = 0; q < C / 8; ++q) {
        float4 a_nxt = a_cur;
        float b_nxt[4][NCT];
        if (q + 1 < C / 8) {
            a_nxt = ld4(arow + 8 * (q + 1));
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) b_nxt[jj][ct] = wcol[(8 * (q + 1) + jj) * C + ct * 32];
        }
        const float av[4] = {a_cur.x, a_cur.y, a_cur.z, a_cur.w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], b_cur[jj][ct], acc[ct], 0, 0, 0);
        if (q + 1 < C / 8) {
            a_cur = a_nxt;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) b_cur[jj][ct] = b_nxt[jj][ct];
        }
    }
}

// write the accumulators back into the (wave-private) tile region as row-major [32][PITCH]
template <int C>
__device__ __forceinline__ void acc_to_tile(float* __restrict__ tile, const f32x16 (&acc)[Tile<C>::NCT], int lane) {
    constexpr int P = Tile<C>::PITCH;
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ct = 0; ct < Tile<C>::NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            tile[row * P + ct * 32 + j] = acc[ct][r];
        }
}

// Cooperative load of a C x C matrix into LDS as Wl[k][j].
//   trans == 0: source is [k][j] row-major;  trans == 1: source is [j][k] (nn.Linear weight, or W^T for backward).
// The float4 loads are issued in batches of up to 8 per thread BEFORE the LDS stores: the obvious copy loop compiles to
// load -> s_waitcnt vmcnt(0) -> ds_write per trip, i.e. one serialised L2 round trip per 4 KB (measured 4 x ~0.4 us per launch).
template <int C, int NTH>
__device__ __forceinline__ void load_w_lds(float* __restrict__ Wl, const float* __restrict__ W, int trans, int tid) {
    constexpr int NF4 = C * C / 4;
    static_assert(NF4 % NTH == 0, "weight must divide evenly over the workgroup");
    constexpr int K = NF4 / NTH, KB = K < 8 ? K : 8;
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += KB) {
        float4 v[KB];
        if (!trans) {
#pragma unroll
            for (int k = 0; k < KB; ++k) v[k] = ld4(W + 4 * (tid + (k0 + k) * NTH));
#pragma unroll
            for (int k = 0; k < KB; ++k) st4(Wl + 4 * (tid + (k0 + k) * NTH), v[k]);
        } else {
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int f = tid + (k0 + k) * NTH, k4 = f / C, j = f % C;      // lanes walk j: conflict-free LDS writes
                v[k] = ld4(W + (size_t)j * C + 4 * k4);
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int f = tid + (k0 + k) * NTH, k4 = f / C, j = f % C;
                Wl[(4 * k4 + 0) * C + j] = v[k].x;
                Wl[(4 * k4 + 1) * C + j] = v[k].y;
                Wl[(4 * k4 + 2) * C + j] = v[k].z;
                Wl[(4 * k4 + 3) * C + j] = v[k].w;
            }
        }
    }
}
