// The gate of the downstream front end (reference model/Model.py:5-18 `Fusion`, :106 `lin_test`) — SURVEY.md section 8(f), the caller behind the
// pretrained encoder:
//     x_t  = flow . W_t^T + b_t                      (lin_test: input_base_dim -> C, a rank-`base` lift of the raw flow)
//     z    = sigmoid(F . W_s^T + b_s + x_t . W_h^T + b_h)        (HS_fc on the frozen encoder's embedding F, HT_fc on the lift)
//     Hm   = z * F + (1 - z) * x_t
//     out  = Hm . W_o^T + b_o                        (output_fc)
// As torch modules this is three rocBLAS GEMMs and seven elementwise launches over (B,T,N,C) per downstream batch, each a round trip through
// HBM (~10 A of traffic for 2 A of data).  Here the forward is ONE launch: a wave owns 16-row tiles; the lift is formed per lane straight in
// the MFMA A-operand layout (no x_t tensor), the two gate GEMMs accumulate into the same tile, the blend runs in the accumulator layout, the
// blended tile goes through a wave-private LDS tile into the A layout of the output GEMM.  fp32 MFMA 16x16x4, weights in LDS (pitch 68).
// The backward's data path is one launch too (dHm = dOut W_o, gate derivative, the operands of the weight gradients); the weight gradients
// themselves reuse gptst_wgrad_colsum / gptst_apply (gpt-st_amd/fusion.py).  C = 64, base <= 4.
#include "common.h"
#include "gptst_hip.h"

#define FU_P 68
#define FU_MAXB 4

struct FuW { const float* Ws; const float* bs; const float* Wh; const float* bh; const float* Wo; const float* bo; const float* Wt; const float* bt; };

__device__ __forceinline__ float fu_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// x_t[c] for the four channels c0..c0+3 of a row whose `base` flow values are s[]
__device__ __forceinline__ float4 fu_lift4(const float* __restrict__ Wtl, const float* __restrict__ btl, const float (&s)[FU_MAXB], int base, int c0) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a = btl[c0 + e];
        for (int b = 0; b < base; ++b) a = fmaf(s[b], Wtl[(c0 + e) * FU_MAXB + b], a);
        v[e] = a;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

// stage a (64 x 64) nn.Linear weight W[o][i] into LDS rows of pitch FU_P, as stored (T = false) or transposed (T = true: rows = i)
template <bool T>
__device__ __forceinline__ void fu_stage_w(float* __restrict__ dst, const float* __restrict__ W, int tid, int nth) {
    for (int f = tid; f < 64 * 16; f += nth) {
        const int r = f >> 4, c4 = f & 15;
        const float4 v = ld4(W + r * 64 + 4 * c4);
        if (!T) st4(dst + r * FU_P + 4 * c4, v);
        else { dst[(4 * c4 + 0) * FU_P + r] = v.x; dst[(4 * c4 + 1) * FU_P + r] = v.y; dst[(4 * c4 + 2) * FU_P + r] = v.z; dst[(4 * c4 + 3) * FU_P + r] = v.w; }
    }
}

// acc[ct] += A (16 rows x 64, per lane four float4: row j, channels 16q + 4kk ..) . Wl^T   with Wl[o][c] in LDS: D reg r = out[row kk*4 + r][16 ct + j]
__device__ __forceinline__ void fu_gemm(f32x4 (&acc)[4], const float4 (&a)[4], const float* __restrict__ Wl, int j, int kk) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const float4 b = ld4(Wl + (16 * ct + j) * FU_P + 16 * q + 4 * kk);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], b.x, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], b.y, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], b.z, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], b.w, acc[ct], 0, 0, 0);
        }
    }
}

// a 16 x 64 tile held in the accumulator layout (v[ct][r] = element (row kk*4 + r, column 16 ct + j)) -> rows of `out` (coalesced float4 stores)
__device__ __forceinline__ void fu_store_tile(float* __restrict__ tile, const float (&v)[4][4], float* __restrict__ out, int row0, int rows, int lane, int j, int kk) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(kk * 4 + r) * FU_P + 16 * ct + j] = v[ct][r];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = i * 64 + lane, rl = f >> 4, c4 = f & 15;
        if (row0 + rl < rows) st4(out + (size_t)(row0 + rl) * 64 + 4 * c4, ld4(tile + rl * FU_P + 4 * c4));
    }
}

__global__ __launch_bounds__(256, 2) void fusion_gate_fwd_kernel(const float* __restrict__ F, const float* __restrict__ src, int lda, int base, FuW w,
                                                                 float* __restrict__ out, float* __restrict__ zout, int rows, int tiles_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wsl = smem;                      // [64][FU_P] x 3
    float* Whl = Wsl + 64 * FU_P;
    float* Wol = Whl + 64 * FU_P;
    float* Wtl = Wol + 64 * FU_P;           // [64][FU_MAXB]
    float* btl = Wtl + 64 * FU_MAXB;        // [64]
    float* bgl = btl + 64;                  // [64]  b_s + b_h
    float* bol = bgl + 64;                  // [64]
    float* tiles = bol + 64;                // [4 waves][16][FU_P]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kk = lane >> 4;
    fu_stage_w<false>(Wsl, w.Ws, tid, 256); fu_stage_w<false>(Whl, w.Wh, tid, 256); fu_stage_w<false>(Wol, w.Wo, tid, 256);
    if (tid < 64) {
        for (int b = 0; b < FU_MAXB; ++b) Wtl[tid * FU_MAXB + b] = b < base ? w.Wt[tid * base + b] : 0.f;
        btl[tid] = w.bt[tid]; bgl[tid] = w.bs[tid] + w.bh[tid]; bol[tid] = w.bo[tid];
    }
    __syncthreads();
    float* tile = tiles + wave * 16 * FU_P;
    const int ntiles = (rows + 15) / 16;
    const int t0 = (blockIdx.x * 4 + wave) * tiles_per_wave, t1 = min(ntiles, t0 + tiles_per_wave);
    for (int t = t0; t < t1; ++t) {
        const int row0 = t * 16;
        // ---- operands in the A layout: lane (j, kk) = row j, channels 16q + 4kk .. ----
        const int ra = min(row0 + j, rows - 1);
        float4 af[4], ax[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) af[q] = ld4(F + (size_t)ra * 64 + 16 * q + 4 * kk);
        float sa[FU_MAXB];
#pragma unroll
        for (int b = 0; b < FU_MAXB; ++b) sa[b] = b < base ? src[(size_t)ra * lda + b] : 0.f;
        // the same operands in the accumulator layout (rows kk*4 + r, column 16 ct + j): the blend runs there
        float fd[4][4], xd[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = min(row0 + kk * 4 + r, rows - 1);
            float sd[FU_MAXB];
#pragma unroll
            for (int b = 0; b < FU_MAXB; ++b) sd[b] = b < base ? src[(size_t)rr * lda + b] : 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                fd[ct][r] = F[(size_t)rr * 64 + 16 * ct + j];
                float a = btl[16 * ct + j];
                for (int b = 0; b < base; ++b) a = fmaf(sd[b], Wtl[(16 * ct + j) * FU_MAXB + b], a);
                xd[ct][r] = a;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ax[q] = fu_lift4(Wtl, btl, sa, base, 16 * q + 4 * kk);
        SB();
        // ---- gate: z = sigmoid(F W_s^T + x_t W_h^T + b_s + b_h) ----
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        fu_gemm(acc, af, Wsl, j, kk);
        fu_gemm(acc, ax, Whl, j, kk);
        float zv[4][4], hv[4][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = fu_sigmoid(acc[ct][r] + bgl[16 * ct + j]);
                zv[ct][r] = z;
                hv[ct][r] = fmaf(z, fd[ct][r] - xd[ct][r], xd[ct][r]);          // z F + (1 - z) x_t
            }
        if (zout != nullptr) fu_store_tile(tile, zv, zout, row0, rows, lane, j, kk);
        // ---- Hm -> A layout through the wave's tile, out = Hm W_o^T + b_o ----
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[(kk * 4 + r) * FU_P + 16 * ct + j] = hv[ct][r];
        float4 ah[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ah[q] = ld4(tile + j * FU_P + 16 * q + 4 * kk);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        fu_gemm(acc, ah, Wol, j, kk);
        float ov[4][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[ct][r] = acc[ct][r] + bol[16 * ct + j];
        fu_store_tile(tile, ov, out, row0, rows, lane, j, kk);
    }
}

// backward data path: dHm = dOut W_o;  dz = dHm (F - x_t);  dpre = dz z (1 - z)   (gradient of both gate pre-activations);
// dxd = dHm (1 - z)   (the blend's direct path into x_t);  Hm, x_t re-formed as operands of the weight gradients
__global__ __launch_bounds__(256, 2) void fusion_gate_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ F, const float* __restrict__ z,
                                                                 const float* __restrict__ src, int lda, int base, FuW w, float* __restrict__ dpre,
                                                                 float* __restrict__ dxd, float* __restrict__ Hm, float* __restrict__ xt, int rows,
                                                                 int tiles_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* WoT = smem;                      // [64 (c)][FU_P] : W_o transposed, so that dHm[row][c] = sum_o dOut[row][o] W_o[o][c] is the same GEMM form
    float* Wtl = WoT + 64 * FU_P;
    float* btl = Wtl + 64 * FU_MAXB;
    float* tiles = btl + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kk = lane >> 4;
    fu_stage_w<true>(WoT, w.Wo, tid, 256);
    if (tid < 64) {
        for (int b = 0; b < FU_MAXB; ++b) Wtl[tid * FU_MAXB + b] = b < base ? w.Wt[tid * base + b] : 0.f;
        btl[tid] = w.bt[tid];
    }
    __syncthreads();
    float* tile = tiles + wave * 16 * FU_P;
    const int ntiles = (rows + 15) / 16;
    const int t0 = (blockIdx.x * 4 + wave) * tiles_per_wave, t1 = min(ntiles, t0 + tiles_per_wave);
    for (int t = t0; t < t1; ++t) {
        const int row0 = t * 16;
        const int ra = min(row0 + j, rows - 1);
        float4 ad[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ad[q] = ld4(dOut + (size_t)ra * 64 + 16 * q + 4 * kk);
        float fd[4][4], xd[4][4], zd[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = min(row0 + kk * 4 + r, rows - 1);
            float sd[FU_MAXB];
#pragma unroll
            for (int b = 0; b < FU_MAXB; ++b) sd[b] = b < base ? src[(size_t)rr * lda + b] : 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                fd[ct][r] = F[(size_t)rr * 64 + 16 * ct + j];
                zd[ct][r] = z[(size_t)rr * 64 + 16 * ct + j];
                float a = btl[16 * ct + j];
                for (int b = 0; b < base; ++b) a = fmaf(sd[b], Wtl[(16 * ct + j) * FU_MAXB + b], a);
                xd[ct][r] = a;
            }
        }
        SB();
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        fu_gemm(acc, ad, WoT, j, kk);
        float v0[4][4], v1[4][4], v2[4][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dh = acc[ct][r], zz = zd[ct][r];
                v0[ct][r] = dh * (fd[ct][r] - xd[ct][r]) * (zz * (1.f - zz));     // dpre
                v1[ct][r] = dh * (1.f - zz);                                     // dxd
                v2[ct][r] = fmaf(zz, fd[ct][r] - xd[ct][r], xd[ct][r]);          // Hm
            }
        fu_store_tile(tile, v0, dpre, row0, rows, lane, j, kk);
        fu_store_tile(tile, v1, dxd, row0, rows, lane, j, kk);
        fu_store_tile(tile, v2, Hm, row0, rows, lane, j, kk);
        fu_store_tile(tile, xd, xt, row0, rows, lane, j, kk);
    }
}

static int fu_tiles_per_wave(int rows) {
    const int ntiles = (rows + 15) / 16;
    int tpw = (ntiles + 4 * 512 - 1) / (4 * 512);                    // ~512 workgroups of 4 waves (two per CU: 70 KB of LDS each)
    return tpw < 1 ? 1 : tpw;
}

// F (rows, 64): the encoder's embedding;  src: the raw input rows, `base` flow values at stride lda;  -> out (rows, 64), z (rows, 64; NULL: not kept)
extern "C" int gptst_fusion_gate_fwd(const float* F, const float* src, int lda, int base, const float* Ws, const float* bs, const float* Wh,
                                     const float* bh, const float* Wo, const float* bo, const float* Wt, const float* bt, float* out, float* z,
                                     int rows, int C, void* stream) {
    if (!F || !src || !Ws || !bs || !Wh || !bh || !Wo || !bo || !Wt || !bt || !out || rows <= 0 || base <= 0 || lda < base) return GPTST_EARG;
    if (C != 64 || base > FU_MAXB) return GPTST_ESHAPE;
    const size_t smem = (3 * 64 * FU_P + 64 * FU_MAXB + 3 * 64 + 4 * 16 * FU_P) * sizeof(float);
    static int done = 0;
    if (!done) { (void)hipFuncSetAttribute((const void*)fusion_gate_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); done = 1; }
    const int tpw = fu_tiles_per_wave(rows), ntiles = (rows + 15) / 16;
    hipLaunchKernelGGL(fusion_gate_fwd_kernel, dim3((ntiles + 4 * tpw - 1) / (4 * tpw)), dim3(256), smem, (hipStream_t)stream, F, src, lda, base,
                       FuW{Ws, bs, Wh, bh, Wo, bo, Wt, bt}, out, z, rows, tpw);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// -> dpre (gradient of the gate pre-activations), dxd (direct gradient of x_t through the blend), Hm, xt (operands of the weight gradients), all (rows, 64)
extern "C" int gptst_fusion_gate_bwd(const float* dOut, const float* F, const float* z, const float* src, int lda, int base, const float* Wo,
                                     const float* Wt, const float* bt, float* dpre, float* dxd, float* Hm, float* xt, int rows, int C, void* stream) {
    if (!dOut || !F || !z || !src || !Wo || !Wt || !bt || !dpre || !dxd || !Hm || !xt || rows <= 0 || base <= 0 || lda < base) return GPTST_EARG;
    if (C != 64 || base > FU_MAXB) return GPTST_ESHAPE;
    const size_t smem = (64 * FU_P + 64 * FU_MAXB + 64 + 4 * 16 * FU_P) * sizeof(float);
    const int tpw = fu_tiles_per_wave(rows), ntiles = (rows + 15) / 16;
    hipLaunchKernelGGL(fusion_gate_bwd_kernel, dim3((ntiles + 4 * tpw - 1) / (4 * tpw)), dim3(256), smem, (hipStream_t)stream, dOut, F, z, src, lda, base,
                       FuW{nullptr, nullptr, nullptr, nullptr, Wo, nullptr, Wt, bt}, dpre, dxd, Hm, xt, rows, tpw);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
