// Time-index embeddings (reference GPTST.py:187-219): time_feature (per (b,t): Linear(1->e) on day / week index) and
// time_feature_spg (per b: Linear(12->e) over the window), each followed by  ln(relu(ln2(relu(ln1(.))))).
// <= 384 rows x e <= 16: pure latency, so the work of one row is spread over E lanes (lane e owns output e of every layer; the
// E x E weight rows / columns it needs live in its registers; activations are exchanged through a per-row LDS line, which is
// separated by workgroup barriers).  The backward recomputes the activations, and the weight gradients are
// reduced per workgroup from the LDS lines (+= with one atomic per output and workgroup).
// Feature k of row r is tidx[(r*K + k)*2 + {0: day, 1: week}].
#include "common.h"

struct TfParams {      // nn.Linear tensors: weight (out,in) row-major, bias (out)
    const float *wd, *bd, *ww, *bw, *w1, *b1, *w2, *b2, *w3, *b3;
};
struct TfGrads {
    float *wd, *bd, *ww, *bw, *w1, *b1, *w2, *b2, *w3, *b3;
};

template <int E, int NT = 256> struct TfCfg { static constexpr int ROWS = NT / E; };

// y[e] = b[e] + sum_i W[e][i] x[i]  with x in the row's LDS line; lane e keeps W[e][:] in registers
template <int E>
__device__ __forceinline__ float tf_matvec(const float (&wrow)[E], float b, const float* __restrict__ xline) {
    float s = b;
#pragma unroll
    for (int i = 0; i < E; ++i) s = fmaf(wrow[i], xline[i], s);
    return s;
}

template <int E>
__device__ __forceinline__ float tf_input(const TfParams& p, const float* __restrict__ tidx, int r, int e, int K) {
    float s = p.bd[e] + p.bw[e];
    for (int k = 0; k < K; ++k) {
        s = fmaf(p.wd[e * K + k], tidx[((size_t)r * K + k) * 2 + 0], s);
        s = fmaf(p.ww[e * K + k], tidx[((size_t)r * K + k) * 2 + 1], s);
    }
    return s;
}

#define TF_SH_FLOATS (7 * 128 * 3)      // largest footprint: E = 2 -> 7 x [128][3]

template <int E>
__device__ __forceinline__ void tf_fwd_body(const TfParams& p, const float* __restrict__ tidx, float* __restrict__ out,
                                            int rows, int K, int blk, float* __restrict__ shraw) {
    constexpr int ROWS = TfCfg<E>::ROWS;
    float (*line)[E + 1] = reinterpret_cast<float (*)[E + 1]>(shraw);
    const int e = threadIdx.x % E, rl = threadIdx.x / E;
    const int r = blk * ROWS + rl;
    const bool valid = r < rows;
    float w1[E], w2[E], w3[E];
#pragma unroll
    for (int i = 0; i < E; ++i) { w1[i] = p.w1[e * E + i]; w2[i] = p.w2[e * E + i]; w3[i] = p.w3[e * E + i]; }
    line[rl][e] = valid ? tf_input<E>(p, tidx, r, e, K) : 0.f;
    __syncthreads();
    float h = fmaxf(tf_matvec<E>(w1, p.b1[e], line[rl]), 0.f);
    __syncthreads();
    line[rl][e] = h;
    __syncthreads();
    h = fmaxf(tf_matvec<E>(w2, p.b2[e], line[rl]), 0.f);
    __syncthreads();
    line[rl][e] = h;
    __syncthreads();
    h = tf_matvec<E>(w3, p.b3[e], line[rl]);
    if (valid) out[(size_t)r * E + e] = h;
}

template <bool DET> __device__ __forceinline__ void tf_add(float* p, float v) { if (DET) *p += v; else atomicAdd(p, v); }

// DET: this workgroup is the only one that touches the job's gradients (it walks all row blocks itself): plain += in a fixed order.
// NT = threads of the workgroup (NT / E rows per block; the deterministic launch runs 1024 threads: a quarter of the passes)
template <int E, bool DET, int NT>
__device__ __forceinline__ void tf_bwd_body(const TfParams& p, const TfGrads& g, const float* __restrict__ tidx,
                                            const float* __restrict__ dout, int rows, int K, int blk, float* __restrict__ shraw) {
    constexpr int ROWS = TfCfg<E, NT>::ROWS;
    float (*sh)[ROWS][E + 1] = reinterpret_cast<float (*)[ROWS][E + 1]>(shraw);      // h0, h1, h2, d3, z2, z1, z0 per row
    const int tid = threadIdx.x, e = tid % E, rl = tid / E;
    const int r = blk * ROWS + rl;
    const bool valid = r < rows;
    float w1[E], w2[E], w3[E], c1[E], c2[E], c3[E];   // rows (forward) and columns (backward) of the three E x E weights
#pragma unroll
    for (int i = 0; i < E; ++i) {
        w1[i] = p.w1[e * E + i]; w2[i] = p.w2[e * E + i]; w3[i] = p.w3[e * E + i];
        c1[i] = p.w1[i * E + e]; c2[i] = p.w2[i * E + e]; c3[i] = p.w3[i * E + e];
    }
    const float h0 = valid ? tf_input<E>(p, tidx, r, e, K) : 0.f;
    sh[0][rl][e] = h0;
    __syncthreads();
    const float h1 = fmaxf(tf_matvec<E>(w1, p.b1[e], sh[0][rl]), 0.f);
    sh[1][rl][e] = h1;
    __syncthreads();
    const float h2 = fmaxf(tf_matvec<E>(w2, p.b2[e], sh[1][rl]), 0.f);
    sh[2][rl][e] = h2;
    const float d3 = valid ? dout[(size_t)r * E + e] : 0.f;
    sh[3][rl][e] = d3;
    __syncthreads();
    const float z2 = h2 > 0.f ? tf_matvec<E>(c3, 0.f, sh[3][rl]) : 0.f;      // relu'(h2) * W3^T d3
    sh[4][rl][e] = z2;
    __syncthreads();
    const float z1 = h1 > 0.f ? tf_matvec<E>(c2, 0.f, sh[4][rl]) : 0.f;
    sh[5][rl][e] = z1;
    __syncthreads();
    const float z0 = tf_matvec<E>(c1, 0.f, sh[5][rl]);
    sh[6][rl][e] = z0;
    __syncthreads();
    const int nrow = min(ROWS, rows - blk * ROWS);
    // dW3[e][i] = sum_r d3[e] h2[i]; dW2 = z2 (x) h1; dW1 = z1 (x) h0; biases = column sums of d3, z2, z1, z0
    for (int idx = tid; idx < 3 * E * E; idx += NT) {
        const int which = idx / (E * E), eo = (idx / E) % E, i = idx % E;
        const int ga = which == 0 ? 3 : (which == 1 ? 4 : 5), gb = which == 0 ? 2 : (which == 1 ? 1 : 0);
        float s = 0.f;
        for (int rr = 0; rr < nrow; ++rr) s = fmaf(sh[ga][rr][eo], sh[gb][rr][i], s);
        float* dst = which == 0 ? g.w3 : (which == 1 ? g.w2 : g.w1);
        tf_add<DET>(dst + eo * E + i, s);
    }
    for (int idx = tid; idx < 4 * E; idx += NT) {
        const int which = idx / E, eo = idx % E;
        const int ga = which == 0 ? 3 : (which == 1 ? 4 : (which == 2 ? 5 : 6));
        float s = 0.f;
        for (int rr = 0; rr < nrow; ++rr) s += sh[ga][rr][eo];
        if (which == 0) tf_add<DET>(g.b3 + eo, s);
        else if (which == 1) tf_add<DET>(g.b2 + eo, s);
        else if (which == 2) tf_add<DET>(g.b1 + eo, s);
        else { tf_add<DET>(g.bd + eo, s); tf_add<DET>(g.bw + eo, s); }
    }
    // input Linears: dWd[e][k] = sum_r z0[e] * day[r,k]; dWw likewise
    for (int idx = tid; idx < 2 * E * K; idx += NT) {
        const int ch = idx / (E * K), eo = (idx / K) % E, k = idx % K;
        float s = 0.f;
        for (int rr = 0; rr < nrow; ++rr)
            s = fmaf(sh[6][rr][eo], tidx[((size_t)(blk * ROWS + rr) * K + k) * 2 + ch], s);
        tf_add<DET>((ch == 0 ? g.wd : g.ww) + eo * K + k, s);
    }
}

// One launch for up to TF_MAXJ time-feature instances (a step has seven: three per STHCN + teb4mask): block range per job.
#define TF_MAXJ 16
struct TfJob { TfParams p; TfGrads g; const float* tidx; float* out; const float* dout; int rows, K, E, blk0; };
struct TfJobs { TfJob j[TF_MAXJ]; int n; };

// BWD: 0 forward, 1 backward (workgroup per row block, atomics), 2 deterministic backward (ONE workgroup of 1024 threads per job walks its blocks)
#define TF_DET_T 1024
template <int BWD>
__global__ __launch_bounds__(BWD == 2 ? TF_DET_T : 256) void timefeat_jobs_kernel(TfJobs t) {
    constexpr int NT = BWD == 2 ? TF_DET_T : 256;
    __shared__ float shraw[TF_SH_FLOATS * (NT / 256)];
    int q = 0;
    if (BWD == 2) q = blockIdx.x;
    else for (int i = 1; i < t.n; ++i) if ((int)blockIdx.x >= t.j[i].blk0) q = i;
    const TfJob& a = t.j[q];
    const int blk = blockIdx.x - a.blk0;
    const int nblk = (a.rows + NT / a.E - 1) / (NT / a.E);
#define TF_CASE(EE)                                                                              \
    case EE:                                                                                     \
        if (BWD == 2) { for (int b2 = 0; b2 < nblk; ++b2) { tf_bwd_body<EE, true, NT>(a.p, a.g, a.tidx, a.dout, a.rows, a.K, b2, shraw); __syncthreads(); } } \
        else if (BWD == 1) tf_bwd_body<EE, false, NT>(a.p, a.g, a.tidx, a.dout, a.rows, a.K, blk, shraw); \
        else tf_fwd_body<EE>(a.p, a.tidx, a.out, a.rows, a.K, blk, shraw);                       \
        break;
    switch (a.E) { TF_CASE(2) TF_CASE(4) TF_CASE(8) TF_CASE(16) default: break; }
#undef TF_CASE
}

static int tf_launch(TfJobs& t, int bwd, hipStream_t st) {
    int nb = 0;
    for (int q = 0; q < t.n; ++q) {
        TfJob& a = t.j[q];
        if (!a.p.wd || !a.tidx || a.rows <= 0 || a.K <= 0 || (bwd ? (!a.dout || !a.g.wd) : !a.out)) return GPTST_EARG;
        if (a.E != 2 && a.E != 4 && a.E != 8 && a.E != 16) return GPTST_ESHAPE;
        const int rows_per = 256 / a.E;
        a.blk0 = nb;
        nb += (a.rows + rows_per - 1) / rows_per;
    }
    if (bwd && g_deterministic) hipLaunchKernelGGL(timefeat_jobs_kernel<2>, dim3(t.n), dim3(TF_DET_T), 0, st, t);
    else if (bwd) hipLaunchKernelGGL(timefeat_jobs_kernel<1>, dim3(nb), dim3(256), 0, st, t);
    else hipLaunchKernelGGL(timefeat_jobs_kernel<0>, dim3(nb), dim3(256), 0, st, t);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// params / grads: the 10 nn.Linear tensors in module order: ln_day.{weight,bias}, ln_week.{..}, ln1, ln2, ln.
extern "C" int gptst_timefeat_fwd(const float* wd, const float* bd, const float* ww, const float* bw, const float* w1, const float* b1,
                                  const float* w2, const float* b2, const float* w3, const float* b3, const float* tidx, float* out,
                                  int rows, int K, int E, void* stream) {
    TfJobs t; t.n = 1;
    t.j[0] = TfJob{TfParams{wd, bd, ww, bw, w1, b1, w2, b2, w3, b3}, TfGrads{}, tidx, out, nullptr, rows, K, E, 0};
    return tf_launch(t, 0, (hipStream_t)stream);
}

// gradients are ACCUMULATED (+=) into gwd..gb3
extern "C" int gptst_timefeat_bwd(const float* wd, const float* bd, const float* ww, const float* bw, const float* w1, const float* b1,
                                  const float* w2, const float* b2, const float* w3, const float* b3, float* gwd, float* gbd, float* gww,
                                  float* gbw, float* gw1, float* gb1, float* gw2, float* gb2, float* gw3, float* gb3, const float* tidx,
                                  const float* dout, int rows, int K, int E, void* stream) {
    TfJobs t; t.n = 1;
    t.j[0] = TfJob{TfParams{wd, bd, ww, bw, w1, b1, w2, b2, w3, b3}, TfGrads{gwd, gbd, gww, gbw, gw1, gb1, gw2, gb2, gw3, gb3}, tidx,
                   nullptr, dout, rows, K, E, 0};
    return tf_launch(t, 1, (hipStream_t)stream);
}

// njobs instances in ONE launch.  params: njobs x 10 pointers (module order, as above); grads: njobs x 10 pointers (bwd only, +=);
// io: njobs outputs (fwd) or output gradients (bwd); rows / K / E per job; all jobs read the same tidx (B,T,2).
extern "C" int gptst_timefeat_jobs(int njobs, int bwd, const void* const* params, const void* const* grads, const float* tidx,
                                   const void* const* io, const int* rows, const int* K, const int* E, void* stream) {
    if (njobs <= 0 || njobs > TF_MAXJ || !params || !io || !rows || !K || !E || (bwd && !grads)) return GPTST_EARG;
    TfJobs t; t.n = njobs;
    for (int q = 0; q < njobs; ++q) {
        const float* const* pp = (const float* const*)params + 10 * q;
        TfGrads g{};
        if (bwd) { float* const* gg = (float* const*)grads + 10 * q; g = TfGrads{gg[0], gg[1], gg[2], gg[3], gg[4], gg[5], gg[6], gg[7], gg[8], gg[9]}; }
        t.j[q] = TfJob{TfParams{pp[0], pp[1], pp[2], pp[3], pp[4], pp[5], pp[6], pp[7], pp[8], pp[9]}, g, tidx,
                       bwd ? nullptr : (float*)io[q], bwd ? (const float*)io[q] : nullptr, rows[q], K[q], E[q], 0};
    }
    return tf_launch(t, bwd, (hipStream_t)stream);
}
