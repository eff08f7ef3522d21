// Time-index embeddings (reference GPTST.py:187-219): time_feature (per (b,t): Linear(1->e) on day / week index) and
// time_feature_spg (per b: Linear(12->e) over the window), each followed by  ln(relu(ln2(relu(ln1(.))))).
// <= 384 rows x e <= 32: pure latency.  One thread per row; backward recomputes the activations, keeps them in LDS and reduces the
// weight gradients cooperatively (+= into the gradient buffers).  Feature k of row r is tidx[(r*K + k)*2 + {0: day, 1: week}].
#include "common.h"

#define TF_ROWS 64      // rows per workgroup: the backward is a serial chain per workgroup, so use many small ones

struct TfParams {      // nn.Linear tensors: weight (out,in) row-major, bias (out)
    const float *wd, *bd, *ww, *bw, *w1, *b1, *w2, *b2, *w3, *b3;
};
struct TfGrads {
    float *wd, *bd, *ww, *bw, *w1, *b1, *w2, *b2, *w3, *b3;
};

template <int E>
__device__ __forceinline__ void tf_forward_row(const TfParams& p, const float* __restrict__ tidx, int r, int K, float (&h0)[E],
                                               float (&h1)[E], float (&h2)[E], float (&o)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        float s = p.bd[e] + p.bw[e];
        for (int k = 0; k < K; ++k) {
            s = fmaf(p.wd[e * K + k], tidx[((size_t)r * K + k) * 2 + 0], s);
            s = fmaf(p.ww[e * K + k], tidx[((size_t)r * K + k) * 2 + 1], s);
        }
        h0[e] = s;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        float s = p.b1[e];
#pragma unroll
        for (int i = 0; i < E; ++i) s = fmaf(p.w1[e * E + i], h0[i], s);
        h1[e] = fmaxf(s, 0.f);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        float s = p.b2[e];
#pragma unroll
        for (int i = 0; i < E; ++i) s = fmaf(p.w2[e * E + i], h1[i], s);
        h2[e] = fmaxf(s, 0.f);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        float s = p.b3[e];
#pragma unroll
        for (int i = 0; i < E; ++i) s = fmaf(p.w3[e * E + i], h2[i], s);
        o[e] = s;
    }
}

template <int E>
__global__ __launch_bounds__(TF_ROWS) void timefeat_fwd_kernel(TfParams p, const float* __restrict__ tidx, float* __restrict__ out,
                                                               int rows, int K) {
    const int r = blockIdx.x * TF_ROWS + threadIdx.x;
    if (r >= rows) return;
    float h0[E], h1[E], h2[E], o[E];
    tf_forward_row<E>(p, tidx, r, K, h0, h1, h2, o);
#pragma unroll
    for (int e = 0; e < E; ++e) out[(size_t)r * E + e] = o[e];
}

template <int E>
__global__ __launch_bounds__(256) void timefeat_bwd_kernel(TfParams p, TfGrads g, const float* __restrict__ tidx,
                                                               const float* __restrict__ dout, int rows, int K) {
    // LDS: per-row vectors [row][E+1] (pad) for h0,h1,h2,do,dz2,dz1,dz0
    __shared__ float sh[7][TF_ROWS][E + 1];
    const int tid = threadIdx.x;
    const int r = blockIdx.x * TF_ROWS + tid;
    const bool valid = tid < TF_ROWS && r < rows;        // 256 threads: the first TF_ROWS own a row, all of them reduce
    float h0[E], h1[E], h2[E], o[E], d3[E], z2[E], z1[E], z0[E];
    if (valid) {
        tf_forward_row<E>(p, tidx, r, K, h0, h1, h2, o);
#pragma unroll
        for (int e = 0; e < E; ++e) d3[e] = dout[(size_t)r * E + e];
#pragma unroll
        for (int i = 0; i < E; ++i) {       // dz2 = relu'(h2) * W3^T d3
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) s = fmaf(p.w3[e * E + i], d3[e], s);
            z2[i] = h2[i] > 0.f ? s : 0.f;
        }
#pragma unroll
        for (int i = 0; i < E; ++i) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) s = fmaf(p.w2[e * E + i], z2[e], s);
            z1[i] = h1[i] > 0.f ? s : 0.f;
        }
#pragma unroll
        for (int i = 0; i < E; ++i) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) s = fmaf(p.w1[e * E + i], z1[e], s);
            z0[i] = s;
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) { h0[e] = h1[e] = h2[e] = d3[e] = z2[e] = z1[e] = z0[e] = 0.f; }
    }
    if (tid < TF_ROWS) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        sh[0][tid][e] = h0[e]; sh[1][tid][e] = h1[e]; sh[2][tid][e] = h2[e]; sh[3][tid][e] = d3[e];
        sh[4][tid][e] = z2[e]; sh[5][tid][e] = z1[e]; sh[6][tid][e] = z0[e];
    }
    }
    __syncthreads();
    const int nrow = min(TF_ROWS, rows - blockIdx.x * TF_ROWS);
    // weight grads: dW3[e][i] = sum_r d3[e] h2[i]; dW2 = z2 (x) h1; dW1 = z1 (x) h0; biases = column sums of d3, z2, z1, z0
    for (int idx = tid; idx < 3 * E * E; idx += 256) {
        const int which = idx / (E * E), e = (idx / E) % E, i = idx % E;
        const int ga = which == 0 ? 3 : (which == 1 ? 4 : 5), gb = which == 0 ? 2 : (which == 1 ? 1 : 0);
        float s = 0.f;
        for (int rr = 0; rr < nrow; ++rr) s = fmaf(sh[ga][rr][e], sh[gb][rr][i], s);
        float* dst = which == 0 ? g.w3 : (which == 1 ? g.w2 : g.w1);
        atomicAdd(dst + e * E + i, s);
    }
    for (int idx = tid; idx < 4 * E; idx += 256) {
        const int which = idx / E, e = idx % E;
        const int ga = which == 0 ? 3 : (which == 1 ? 4 : (which == 2 ? 5 : 6));
        float s = 0.f;
        for (int rr = 0; rr < nrow; ++rr) s += sh[ga][rr][e];
        if (which == 0) atomicAdd(g.b3 + e, s);
        else if (which == 1) atomicAdd(g.b2 + e, s);
        else if (which == 2) atomicAdd(g.b1 + e, s);
        else { atomicAdd(g.bd + e, s); atomicAdd(g.bw + e, s); }
    }
    // input Linears: dWd[e][k] = sum_r z0[e] * day[r,k]; dWw likewise
    for (int idx = tid; idx < 2 * E * K; idx += 256) {
        const int ch = idx / (E * K), e = (idx / K) % E, k = idx % K;
        float s = 0.f;
        for (int rr = 0; rr < nrow; ++rr)
            s = fmaf(sh[6][rr][e], tidx[((size_t)(blockIdx.x * TF_ROWS + rr) * K + k) * 2 + ch], s);
        atomicAdd((ch == 0 ? g.wd : g.ww) + e * K + k, s);
    }
}

#define TF_DISPATCH(E_, CALL)                  \
    switch (E_) {                              \
        case 2: { constexpr int EE = 2; CALL; } break;   \
        case 3: { constexpr int EE = 3; CALL; } break;   \
        case 4: { constexpr int EE = 4; CALL; } break;   \
        case 8: { constexpr int EE = 8; CALL; } break;   \
        case 16: { constexpr int EE = 16; CALL; } break; \
        default: return GPTST_ESHAPE;          \
    }

// params / grads: the 10 nn.Linear tensors in module order: ln_day.{weight,bias}, ln_week.{..}, ln1, ln2, ln.
extern "C" int gptst_timefeat_fwd(const float* wd, const float* bd, const float* ww, const float* bw, const float* w1, const float* b1,
                                  const float* w2, const float* b2, const float* w3, const float* b3, const float* tidx, float* out,
                                  int rows, int K, int E, void* stream) {
    if (!wd || !tidx || !out || rows <= 0 || K <= 0) return GPTST_EARG;
    TfParams p{wd, bd, ww, bw, w1, b1, w2, b2, w3, b3};
    dim3 grid((rows + TF_ROWS - 1) / TF_ROWS);
    TF_DISPATCH(E, hipLaunchKernelGGL((timefeat_fwd_kernel<EE>), grid, dim3(TF_ROWS), 0, (hipStream_t)stream, p, tidx, out, rows, K));
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// gradients are ACCUMULATED (+=) into gwd..gb3
extern "C" int gptst_timefeat_bwd(const float* wd, const float* bd, const float* ww, const float* bw, const float* w1, const float* b1,
                                  const float* w2, const float* b2, const float* w3, const float* b3, float* gwd, float* gbd, float* gww,
                                  float* gbw, float* gw1, float* gb1, float* gw2, float* gb2, float* gw3, float* gb3, const float* tidx,
                                  const float* dout, int rows, int K, int E, void* stream) {
    if (!wd || !tidx || !dout || !gwd || rows <= 0 || K <= 0) return GPTST_EARG;
    TfParams p{wd, bd, ww, bw, w1, b1, w2, b2, w3, b3};
    TfGrads g{gwd, gbd, gww, gbw, gw1, gb1, gw2, gb2, gw3, gb3};
    dim3 grid((rows + TF_ROWS - 1) / TF_ROWS);
    TF_DISPATCH(E, hipLaunchKernelGGL((timefeat_bwd_kernel<EE>), grid, dim3(256), 0, (hipStream_t)stream, p, g, tidx, dout, rows, K));
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
