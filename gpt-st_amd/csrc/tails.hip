// Fused "thin head" kernels at the two places where the loss meets the network — one pass over the C-wide activation each,
// replacing chains of 4-6 small launches:
//
//   mae tail  (reference GPTST.py:455 dim_flow_out, Run.py:92-100 scaler_mae_loss, lib/metrics.py:11-18 and their autograd backward)
//       out = dec W^T + b;  p = (out s + m) M, y = (label s + m) M, keep = y > thresh;  stats += (sum_keep |y-p|, #keep);
//       a = sign(p-y) M s [keep]  (gradient of the SUM loss: the 1/#keep of the mean is applied by the optimiser, hyper[9] = 1);
//       d_dec = a W;  partial (gW, gb) = (a^T dec, sum a)                    was rowdot, mae_fwd, mae_bwd, lin_in, rowouter x2
//   kl head   (reference BasicTrainer.py:85 0.1 KLDivLoss(sum)(log prob, eb), softmax + MLP_RL.ln3 GPTST.py:33 backward)
//       a = w (prob sum_h eb - eb);  stats[2] += sum eb (log eb - log prob);  d_h2 = a W3;  partial (gW3, gb3) = (a^T h2, sum a)
//                                                                          was kl, lin_in, rowouter x2
// Layout as small.hip: C/4 lanes per row (float4 columns), 16 row slots per workgroup; TL_NB persistent workgroups walk contiguous
// row chunks, keep the weight-gradient partials in registers and fold them through LDS once at the end: part[blk][J*C + J]
// (the caller sums the TL_NB partials of [gW | gb] with one bwd_pool job) and write their loss statistics to sws[blk][4]
// ([0] sum |y-p|, [1] kept count: mae tail; [2] KL sum: kl head), which gptst_stats_fold sums in index order into stats: no float
// atomics, the result does not depend on scheduling.  (Measured: 510 same-address atomics at the end of a launch — the stats
// themselves, or a "last workgroup folds" ticket — cost ~8 us; one extra 1-workgroup launch costs 2.5.)
// C in {64, 128}, J <= TL_MAXJ; other shapes use the unfused ops.
#include "common.h"

#define TL_NB 512
#define TL_MAXJ 16

struct TailArgs {
    const float* X; const float* W; const float* b; float* dX; float* part; float* sws;
    int rows, J, rows_per_block;
    // mae tail
    const float* src; const float* mask; float* out; int lda; float sigma, mu, thresh;
    // kl head
    const float* prob; const float* c; int N; float w;
    int premul;          // dPre chain: dX is multiplied by lrelu'(X) (X = the output of a LeakyReLU layer), include/gptst_hip.h
};

template <int KIND, int C>      // KIND 0: mae tail, 1: kl head
__global__ __launch_bounds__(256) void tail_kernel(TailArgs t) {
    constexpr int LPR = C / 4, RPB = 256 / LPR;
    __shared__ __attribute__((aligned(16))) float Ws[TL_MAXJ * C];
    __shared__ float4 red[RPB][LPR];
    __shared__ float redb[RPB];
    __shared__ float reds[3][4];
    const int J = t.J;
    for (int i = threadIdx.x; i < J * C / 4; i += 256) st4(Ws + 4 * i, ld4(t.W + 4 * i));
    __syncthreads();
    const int c4 = threadIdx.x % LPR, slot = threadIdx.x / LPR;
    const size_t r0 = (size_t)blockIdx.x * t.rows_per_block;
    const size_t r1 = min((size_t)t.rows, r0 + t.rows_per_block);
    float4 accW[TL_MAXJ];
    float accb[TL_MAXJ];
#pragma unroll
    for (int j = 0; j < TL_MAXJ; ++j) { accW[j] = f4zero(); accb[j] = 0.f; }
    float s0 = 0.f, s1 = 0.f;                                         // mae: sum |y-p|, count;  kl: sum, -
    constexpr int UR = 4;                                             // independent rows in flight per thread
    for (size_t i0 = r0 + slot; i0 < r1; i0 += UR * RPB) {
      float4 xs[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) { const size_t i = i0 + (size_t)u * RPB; xs[u] = i < r1 ? ld4(t.X + i * C + 4 * c4) : f4zero(); }
      SB();
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const size_t i = i0 + (size_t)u * RPB;
        if (i >= r1) continue;                                        // uniform per 16-lane row group; no barrier inside
        const float4 x = xs[u];
        // per-row scalars: lane j of the row's C/4-lane group loads / computes entry j (J <= 16) and the group shares the results —
        // every lane loading all J entries itself cost 16x the load instructions (42 us for the KL head at the bench shape)
        float a[TL_MAXJ];
        float a_own = 0.f;
        const int gl = (threadIdx.x & 63) & ~(LPR - 1);                // first lane of this row group within the wave
        if (KIND == 0) {
            float o_all[TL_MAXJ];
#pragma unroll
            for (int j = 0; j < TL_MAXJ; ++j) {
                o_all[j] = 0.f;
                if (j < J) o_all[j] = group_sum<LPR>(f4dot(x, ld4(Ws + j * C + 4 * c4)));      // uniform branch
            }
            float o = 0.f;
#pragma unroll
            for (int j = 0; j < TL_MAXJ; ++j) if (j == c4) o = o_all[j];
            if (c4 < J) {
                o += t.b ? t.b[c4] : 0.f;
                const size_t e = i * J + c4;
                const float M = 1.f - t.mask[e];
                const float p = (o * t.sigma + t.mu) * M;
                const float y = (t.src[i * t.lda + c4] * t.sigma + t.mu) * M;
                if (y > t.thresh) {
                    const float d = p - y;
                    s0 += fabsf(d); s1 += 1.f;
                    a_own = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * M * t.sigma;
                }
                t.out[e] = o;
            }
        } else {
            const size_t bt = i / t.N, n = i % t.N;
            float e_ = 0.f, p_ = 1.f;
            if (c4 < J) { e_ = t.c[(bt * J + c4) * t.N + n]; p_ = t.prob[i * J + c4]; }
            const float se = group_sum<LPR>(e_);
            if (c4 < J && e_ > 0.f) s0 += e_ * (logf(e_) - logf(p_));
            a_own = c4 < J ? t.w * (p_ * se - e_) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < TL_MAXJ; ++j) a[j] = j < J ? __shfl(a_own, gl + j, 64) : 0.f;
        float4 dx = f4zero();
#pragma unroll
        for (int j = 0; j < TL_MAXJ; ++j)
            if (j < J) {
                dx = f4fma(a[j], ld4(Ws + j * C + 4 * c4), dx);
                accW[j] = f4fma(a[j], x, accW[j]);
                accb[j] += a[j];
            }
        if (t.premul) {
            dx.x *= lrelu_grad_from_out(x.x); dx.y *= lrelu_grad_from_out(x.y); dx.z *= lrelu_grad_from_out(x.z); dx.w *= lrelu_grad_from_out(x.w);
        }
        st4(t.dX + i * C + 4 * c4, dx);
      }
    }
    // ---- fold the 16 row slots of the workgroup, then the loss statistics ----
    float* mine = t.part + (size_t)blockIdx.x * (J * C + J);
#pragma unroll
    for (int j = 0; j < TL_MAXJ; ++j) {
        if (j >= J) continue;                                         // uniform
        red[slot][c4] = accW[j];
        if (c4 == 0) redb[slot] = accb[j];
        __syncthreads();
        if (slot == 0) {
            float4 s = red[0][c4];
            for (int q = 1; q < RPB; ++q) s = f4add(s, red[q][c4]);
            st4(mine + j * C + 4 * c4, s);
            if (c4 == 0) {
                float sb = redb[0];
                for (int q = 1; q < RPB; ++q) sb += redb[q];
                mine[J * C + j] = sb;
            }
        }
        __syncthreads();
    }
    s0 = group_sum<64>(s0); s1 = group_sum<64>(s1);
    if ((threadIdx.x & 63) == 0) { reds[0][threadIdx.x >> 6] = s0; reds[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* w = t.sws + 4 * (size_t)blockIdx.x;
        const float v0 = (reds[0][0] + reds[0][1]) + (reds[0][2] + reds[0][3]), v1 = (reds[1][0] + reds[1][1]) + (reds[1][2] + reds[1][3]);
        if (KIND == 0) { w[0] = v0; w[1] = v1; } else { w[2] = v0; }
    }
}

// ---- the same two heads on MFMA 16x16x4 (C = 64, r03) ------------------------------------------------------------------------------------
// tail_kernel spends its time in J dot products per row (4 FMAs + a 4-step DPP reduction each), J rank-1 updates of the data gradient and J
// of the weight gradient per row.  Here a wave takes 16-row tiles and the three products are matrix products with register operands:
//   Z  (16 rows x 16 classes) = X . W^T          A = X rows straight from global (lane (j,kk): row j, channels 16q+4kk..), B = W[class j][..]
//   dX (16 rows x 64)         = a . W            A = a through a wave-private LDS tile (D layout -> A layout), B = W[class][4j+ct] (float4)
//   gW (16 classes x 64)     += a^T . X          A = a in the D layout as it is (step s <-> row 4kk+s), B = X rows in the D layout (float4)
// D layout of Z / a: lane (j = class, kk), register r <-> row 4kk + r, so the per-row softmax terms run across the 16 lanes of a DPP row.
// Same outputs as tail_kernel (out, dX, part[blk][J*C + J], sws[blk][4]); sums are accumulated in a different order (tolerance-checked).
template <int KIND, int C>      // C = 128 (r05): the D-layout operands come in two 64-channel halves hf (channel 64 hf + 4j + ct)
__global__ __launch_bounds__(256) void tail_mfma_kernel(TailArgs t) {
    constexpr int Q = C / 16, HF = C / 64;
    __shared__ float at[4][16][17];                     // per wave: a[row][class] (D layout -> A layout)
    __shared__ __attribute__((aligned(16))) float fold[4][16 * C + 16];
    __shared__ float reds[2][4];
    const int J = t.J;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    // W as B operand of Z (class j, channels 16q + 4kk ..) and of dX (class 4s + kk, channels 4j .. 4j+3)
    float4 bz[Q], bd[HF][4];
#pragma unroll
    for (int q = 0; q < Q; ++q) bz[q] = (KIND == 0 && j < J) ? ld4(t.W + (size_t)j * C + 16 * q + 4 * kk) : f4zero();
#pragma unroll
    for (int hf = 0; hf < HF; ++hf)
#pragma unroll
        for (int q = 0; q < 4; ++q) bd[hf][q] = (4 * q + kk < J) ? ld4(t.W + (size_t)(4 * q + kk) * C + 64 * hf + 4 * j) : f4zero();
    const float bj = (KIND == 0 && t.b != nullptr && j < J) ? t.b[j] : 0.f;
    const int nks = (J + 3) / 4;                        // k-steps of the dX product
    f32x4 gw[HF][4];
#pragma unroll
    for (int hf = 0; hf < HF; ++hf)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) gw[hf][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float gb = 0.f, s0 = 0.f, s1 = 0.f;
    const size_t r0 = (size_t)blockIdx.x * t.rows_per_block;
    const size_t r1 = min((size_t)t.rows, r0 + t.rows_per_block);
    for (size_t tb = r0 + 16 * wave; tb < r1; tb += 64) {
        // X tile in both layouts (the second read hits L1): A layout for Z, D layout for gW / the LeakyReLU sign
        float4 xa[Q], xd[HF][4];
        if (KIND == 0) {
#pragma unroll
            for (int q = 0; q < Q; ++q) xa[q] = ld4(t.X + min(tb + j, r1 - 1) * C + 16 * q + 4 * kk);
        }
#pragma unroll
        for (int hf = 0; hf < HF; ++hf)
#pragma unroll
            for (int q = 0; q < 4; ++q) xd[hf][q] = ld4(t.X + min(tb + 4 * kk + q, r1 - 1) * C + 64 * hf + 4 * j);
        // the epilogue's per-row operands travel with the X tile (after the MFMAs they were a second, dependent round trip per tile)
        float o0[4], o1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t i = min(tb + 4 * kk + r, r1 - 1);
            const int jc = min(j, J - 1);
            if (KIND == 0) { o0[r] = t.mask[i * J + jc]; o1[r] = t.src[i * t.lda + jc]; }
            else { const size_t bt = i / t.N, n = i % t.N; o0[r] = t.c[(bt * J + jc) * t.N + n]; o1[r] = t.prob[i * J + jc]; }
        }
        SB();
        float a[4];
        if (KIND == 0) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[q].x, bz[q].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[q].y, bz[q].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[q].z, bz[q].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[q].w, bz[q].w, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t i = tb + 4 * kk + r;
                a[r] = 0.f;
                if (i < r1 && j < J) {
                    const float o = acc0[r] + acc1[r] + bj;
                    const size_t e = i * J + j;
                    const float M = 1.f - o0[r];
                    const float p = (o * t.sigma + t.mu) * M;
                    const float y = (o1[r] * t.sigma + t.mu) * M;
                    if (y > t.thresh) {
                        const float d = p - y;
                        s0 += fabsf(d); s1 += 1.f;
                        a[r] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * M * t.sigma;
                    }
                    t.out[e] = o;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t i = tb + 4 * kk + r;
                const bool ok = i < r1 && j < J;
                float e_ = 0.f, p_ = 1.f;
                if (ok) { e_ = o0[r]; p_ = o1[r]; }
                const float se = group_sum<16>(e_);
                if (ok && e_ > 0.f) s0 += e_ * (logf(e_) - logf(p_));
                a[r] = ok ? t.w * (p_ * se - e_) : 0.f;
            }
        }
        // ---- gW += a^T X (step s <-> row 4kk + s on both operands), gb += column sums of a ----
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gb += a[r];
#pragma unroll
            for (int hf = 0; hf < HF; ++hf) {
                gw[hf][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], xd[hf][r].x, gw[hf][0], 0, 0, 0);
                gw[hf][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], xd[hf][r].y, gw[hf][1], 0, 0, 0);
                gw[hf][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], xd[hf][r].z, gw[hf][2], 0, 0, 0);
                gw[hf][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], xd[hf][r].w, gw[hf][3], 0, 0, 0);
            }
        }
        // ---- dX = a W: a from the D layout into the A layout (lane (i = row, kk): classes 4s + kk) through the wave's tile ----
#pragma unroll
        for (int r = 0; r < 4; ++r) at[wave][4 * kk + r][j] = a[r];
        f32x4 dx[HF][4];
#pragma unroll
        for (int hf = 0; hf < HF; ++hf)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) dx[hf][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < nks) {                                           // uniform
                const float as = at[wave][j][4 * s + kk];
#pragma unroll
                for (int hf = 0; hf < HF; ++hf) {
                    dx[hf][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(as, bd[hf][s].x, dx[hf][0], 0, 0, 0);
                    dx[hf][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(as, bd[hf][s].y, dx[hf][1], 0, 0, 0);
                    dx[hf][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(as, bd[hf][s].z, dx[hf][2], 0, 0, 0);
                    dx[hf][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(as, bd[hf][s].w, dx[hf][3], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t i = tb + 4 * kk + r;
#pragma unroll
            for (int hf = 0; hf < HF; ++hf) {
                float4 v = make_float4(dx[hf][0][r], dx[hf][1][r], dx[hf][2][r], dx[hf][3][r]);
                if (t.premul) {
                    v.x *= lrelu_grad_from_out(xd[hf][r].x); v.y *= lrelu_grad_from_out(xd[hf][r].y);
                    v.z *= lrelu_grad_from_out(xd[hf][r].z); v.w *= lrelu_grad_from_out(xd[hf][r].w);
                }
                if (i < r1) st4(t.dX + i * C + 64 * hf + 4 * j, v);
            }
        }
    }
    // ---- fold the four waves: gW (D reg r of tile ct: class 4kk + r, channel 4j + ct), gb, loss statistics ----
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int hf = 0; hf < HF; ++hf) st4(&fold[wave][(4 * kk + r) * C + 64 * hf + 4 * j], make_float4(gw[hf][0][r], gw[hf][1][r], gw[hf][2][r], gw[hf][3][r]));
    gb += __shfl_xor(gb, 16, 64); gb += __shfl_xor(gb, 32, 64);
    if (kk == 0) fold[wave][16 * C + j] = gb;
    s0 = group_sum<64>(s0); s1 = group_sum<64>(s1);
    if (lane == 0) { reds[0][wave] = s0; reds[1][wave] = s1; }
    __syncthreads();
    float* mine = t.part + (size_t)blockIdx.x * (J * C + J);
    for (int o = threadIdx.x; o < J * C; o += 256) mine[o] = (fold[0][o] + fold[1][o]) + (fold[2][o] + fold[3][o]);
    if ((int)threadIdx.x < J) mine[J * C + threadIdx.x] = (fold[0][16 * C + threadIdx.x] + fold[1][16 * C + threadIdx.x]) + (fold[2][16 * C + threadIdx.x] + fold[3][16 * C + threadIdx.x]);
    if (threadIdx.x == 0) {
        float* w = t.sws + 4 * (size_t)blockIdx.x;
        const float v0 = (reds[0][0] + reds[0][1]) + (reds[0][2] + reds[0][3]), v1 = (reds[1][0] + reds[1][1]) + (reds[1][2] + reds[1][3]);
        if (KIND == 0) { w[0] = v0; w[1] = v1; } else { w[2] = v0; }
    }
}

// stats[k] += sum over rows of sws[row][k], k < 3, in a fixed order (thread t: rows t, t+256, ...; then a fixed tree)
// lost0 / lost1 (may be NULL): the hand-off expiry counters — their sum goes to stats[5], so that a gradient all-reduce carries it to every rank and
// all of them skip the update together (gptst_clip_adam's guard)
__global__ __launch_bounds__(256) void stats_fold_kernel(const float* __restrict__ sws, int rows, float* __restrict__ stats,
                                                         const unsigned* __restrict__ lost0, const unsigned* __restrict__ lost1,
                                                         const unsigned* __restrict__ lost2) {
    __shared__ float red[3][4];
    if (threadIdx.x == 255) stats[5] = (float)((lost0 != nullptr ? *lost0 : 0u) + (lost1 != nullptr ? *lost1 : 0u) + (lost2 != nullptr ? *lost2 : 0u));
    float s[3] = {0.f, 0.f, 0.f};
    for (int r = threadIdx.x; r < rows; r += 256) {
        const float4 v = ld4(sws + 4 * (size_t)r);
        s[0] += v.x; s[1] += v.y; s[2] += v.z;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s[k] = group_sum<64>(s[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) stats[threadIdx.x] += (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

thread_local int g_tl_nb = TL_NB;       // experiments: gptst_tune(6, nb)
thread_local int g_tl_mfma = 1;         // gptst_tune(15, 0): the VALU loss heads (tail_kernel) instead of the MFMA ones
static void tl_geometry(int rows, int& nb, int& rpb) {
    int want = g_tl_nb;
    if (g_tl_nb == TL_NB && rows > TL_NB * 512) { want = rows / 512; if (want > 4096) want = 4096; }      // N = 4096: 1.5 M rows -> 3072 chunks
    rpb = (rows + want - 1) / want; if (rpb < 16) rpb = 16;
    nb = (rows + rpb - 1) / rpb;
}

// number of row-chunk partials that gptst_tail_mae / gptst_tail_kl write: part must hold that many x (J*C + J) floats
extern "C" int gptst_tail_parts(int rows) { int nb, rpb; tl_geometry(rows, nb, rpb); return nb; }

GPTST_INTERNAL const unsigned* gptst_handoff_word_capmfma(void);
GPTST_INTERNAL const unsigned* gptst_handoff_word_hypertem(void);
GPTST_INTERNAL const unsigned* gptst_handoff_word_masksel(void);
// sum the per-workgroup loss statistics sws (rows, 4) of the tail kernels into stats[0..2] (+=), in a fixed order; stats[5] <- hand-off expiries on record
extern "C" int gptst_stats_fold(const float* sws, int rows, float* stats, void* stream) {
    if (!sws || !stats || rows <= 0) return GPTST_EARG;
    static const unsigned* w0 = gptst_handoff_word_capmfma();         // looked up once (the steppers warm up before any capture)
    static const unsigned* w1 = gptst_handoff_word_hypertem();
    static const unsigned* w2 = gptst_handoff_word_masksel();
    hipLaunchKernelGGL(stats_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sws, rows, stats, w0, w1, w2);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_tail_mae(const float* dec, const float* W, const float* b, const float* src, int lda, const float* mask, float sigma,
                              float mu, float thresh, float* out, float* d_dec, float* part, float* sws, int premul, int rows, int J, int C,
                              void* stream) {
    if (!dec || !W || !src || !mask || !out || !d_dec || !part || !sws || rows <= 0 || J <= 0) return GPTST_EARG;
    if ((C != 64 && C != 128) || J > TL_MAXJ) return GPTST_ESHAPE;
    TailArgs t{};
    t.X = dec; t.W = W; t.b = b; t.dX = d_dec; t.part = part; t.sws = sws; t.rows = rows; t.J = J;
    t.src = src; t.mask = mask; t.out = out; t.lda = lda; t.sigma = sigma; t.mu = mu; t.thresh = thresh; t.premul = premul;
    int nb; tl_geometry(rows, nb, t.rows_per_block);
    if (C == 64 && g_tl_mfma) hipLaunchKernelGGL((tail_mfma_kernel<0, 64>), dim3(nb), dim3(256), 0, (hipStream_t)stream, t);
    else if (C == 64) hipLaunchKernelGGL((tail_kernel<0, 64>), dim3(nb), dim3(256), 0, (hipStream_t)stream, t);
    else if (g_tl_mfma) hipLaunchKernelGGL((tail_mfma_kernel<0, 128>), dim3(nb), dim3(256), 0, (hipStream_t)stream, t);
    else hipLaunchKernelGGL((tail_kernel<0, 128>), dim3(nb), dim3(256), 0, (hipStream_t)stream, t);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern "C" int gptst_tail_kl(const float* h2, const float* W3, const float* prob, const float* c, float w, float* d_h2, float* part,
                             float* sws, int premul, int rows, int N, int HS, int C, void* stream) {
    if (!h2 || !W3 || !prob || !c || !d_h2 || !part || !sws || rows <= 0 || HS <= 0 || N <= 0) return GPTST_EARG;
    if ((C != 64 && C != 128) || HS > TL_MAXJ) return GPTST_ESHAPE;
    TailArgs t{};
    t.X = h2; t.W = W3; t.dX = d_h2; t.part = part; t.sws = sws; t.rows = rows; t.J = HS;
    t.prob = prob; t.c = c; t.N = N; t.w = w; t.premul = premul;
    int nb; tl_geometry(rows, nb, t.rows_per_block);
    if (C == 64 && g_tl_mfma) hipLaunchKernelGGL((tail_mfma_kernel<1, 64>), dim3(nb), dim3(256), 0, (hipStream_t)stream, t);
    else if (C == 64) hipLaunchKernelGGL((tail_kernel<1, 64>), dim3(nb), dim3(256), 0, (hipStream_t)stream, t);
    else if (g_tl_mfma) hipLaunchKernelGGL((tail_mfma_kernel<1, 128>), dim3(nb), dim3(256), 0, (hipStream_t)stream, t);
    else hipLaunchKernelGGL((tail_kernel<1, 128>), dim3(nb), dim3(256), 0, (hipStream_t)stream, t);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
