// cap for node counts whose (b,t) capsule matrix does not fit LDS (N > ~480 at C = 64, > ~230 at C = 128; BASELINE config 5: N = 4096,
// C = 128) — the same algebra as cap_mfma.hip (reference GPTST.py:102-123,135) as a sequence of small streaming kernels over
// global memory.  The host (ops.py) orchestrates them:
//     P = squash(X Wp^T + bp)                      gptst_apply (shared weight) + capbig_squash_rows
//     c0 = softmax_h(dadj); S = c0 P; v0 = squash(S)
//     R x { [b += v P^T]; c = softmax_h(b); S = c P; v = squash(v0 (.) S) };  b += v P^T;  c = softmax_h(b + dadj);  s = c P
// Every sum over nodes is ONE kernel (capbig_type1), so a node-sharded run only has to all-reduce its (BT,HS,C) result
// (SURVEY.md §8e row 2).  Correctness path: VALU, wave per row, not tuned.
#include "common.h"

template <int C>
__global__ void cb_squash_rows_kernel(float* __restrict__ Y, long rows) {           // in place, one wave per row
    constexpr int E = C / 64;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float y[E], sq = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { y[e] = Y[row * C + lane + 64 * e]; sq = fmaf(y[e], y[e], sq); }
    const float sc = squash_scale(group_sum<64>(sq));
#pragma unroll
    for (int e = 0; e < E; ++e) Y[row * C + lane + 64 * e] = y[e] * sc;
}

// cs[bt,h,n] = softmax_h( use_bl*bl[bt,h,n] + use_l0*l0[bt,h,n] ), one thread per (bt, n)
__global__ void cb_softmax_kernel(const float* __restrict__ bl, const float* __restrict__ l0, float* __restrict__ cs, int BT, int HS,
                                  int N, int use_bl, int use_l0) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)BT * N) return;
    const long bt = i / N, n = i % N;
    const long base = bt * HS * N + n;
    float m = -3.0e38f;
    for (int h = 0; h < HS; ++h) {
        float v = use_bl ? bl[base + (long)h * N] : 0.f;
        if (use_l0) v += l0[base + (long)h * N];
        m = fmaxf(m, v);
    }
    float sum = 0.f;
    for (int h = 0; h < HS; ++h) {
        float v = use_bl ? bl[base + (long)h * N] : 0.f;
        if (use_l0) v += l0[base + (long)h * N];
        sum += __expf(v - m);
    }
    const float inv = 1.f / sum;
    for (int h = 0; h < HS; ++h) {
        float v = use_bl ? bl[base + (long)h * N] : 0.f;
        if (use_l0) v += l0[base + (long)h * N];
        cs[base + (long)h * N] = __expf(v - m) * inv;
    }
}

// S[bt,h,c] += sum_{n in chunk} cs[bt,h,n] P[bt,n,c];  grid (BT, node chunks, ceil(HS/16)).  A thread owns four channels (one float4
// of P per node, 4 nodes in flight) and 256/(C/4) node slots share a chunk; slots fold in LDS, one atomic per output and workgroup.
#define CB_CHUNK 256
template <int C>
__global__ __launch_bounds__(256) void cb_type1_kernel(const float* __restrict__ cs, const float* __restrict__ P, float* __restrict__ S,
                                                       int HS, int N) {
    constexpr int L = C / 4, SL = 256 / L;              // lanes per node row, node slots per workgroup
    __shared__ float4 red[SL][16][L];
    const int bt = blockIdx.x, n0 = blockIdx.y * CB_CHUNK, h0 = blockIdx.z * 16;
    const int c4 = threadIdx.x % L, sl = threadIdx.x / L;
    const int nh = min(16, HS - h0), nend = min(N, n0 + CB_CHUNK);
    float4 acc[16];
#pragma unroll
    for (int h = 0; h < 16; ++h) acc[h] = f4zero();
    const float* csb = cs + ((long)bt * HS + h0) * N;
    for (int n = n0 + sl; n < nend; n += 4 * SL) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int nn = min(n + u * SL, nend - 1);
            p[u] = ld4(P + ((long)bt * N + nn) * C + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int nn = n + u * SL;
            if (nn < nend) {
#pragma unroll
                for (int h = 0; h < 16; ++h)
                    if (h < nh) acc[h] = f4fma(csb[(long)h * N + nn], p[u], acc[h]);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 16; ++h) red[sl][h][c4] = acc[h];
    __syncthreads();
    for (int o = threadIdx.x; o < nh * L; o += 256) {
        const int h = o / L, cc = o % L;
        float4 s = red[0][h][cc];
#pragma unroll
        for (int q = 1; q < SL; ++q) s = f4add(s, red[q][h][cc]);
        float* dst = S + ((long)bt * HS + h0 + h) * C + 4 * cc;
        atomicAdd(dst + 0, s.x); atomicAdd(dst + 1, s.y); atomicAdd(dst + 2, s.z); atomicAdd(dst + 3, s.w);
    }
}
__global__ void cb_zero_kernel(float* __restrict__ p, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// rows of (BT*HS): post 0 -> out = S;  1 -> out = squash(S);  2 -> out = squash(V0 (.) S)
template <int C>
__global__ void cb_post_kernel(const float* __restrict__ S, const float* __restrict__ V0, float* __restrict__ out, long rows, int post) {
    constexpr int E = C / 64;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[E], sq = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        v[e] = S[row * C + lane + 64 * e];
        if (post == 2) v[e] *= V0[row * C + lane + 64 * e];
        sq = fmaf(v[e], v[e], sq);
    }
    const float sc = post == 0 ? 1.f : squash_scale(group_sum<64>(sq));
#pragma unroll
    for (int e = 0; e < E; ++e) out[row * C + lane + 64 * e] = v[e] * sc;
}

// bl[bt,h,n] += V[bt,h,:] . P[bt,n,:]:  C/4 lanes per (bt, n) row, each with one float4 of P; the HS dot products are reduced over
// those lanes with DPP (16 lanes) plus one shuffle for C = 128
template <int C>
__global__ void cb_type2_kernel(const float* __restrict__ V, const float* __restrict__ P, float* __restrict__ bl, int BT, int HS, int N) {
    constexpr int L = C / 4, RPB = 256 / L;
    const long row = (long)blockIdx.x * RPB + threadIdx.x / L;
    const int c4 = threadIdx.x % L;
    const bool ok = row < (long)BT * N;
    const long r = ok ? row : 0;
    const long bt = r / N, n = r % N;
    const float4 p = ld4(P + r * C + 4 * c4);
    for (int h = 0; h < HS; ++h) {
        float d = f4dot(ld4(V + (bt * HS + h) * C + 4 * c4), p);
        d = group_sum<L>(d);
        if (c4 == 0 && ok) bl[(bt * HS + h) * N + n] += d;
    }
}

// rec[bt,n,:] = sum_h c[bt,h,n] v[bt,h,:]   (cluster -> node scatter, GPTST.py:135), one wave per row
template <int C>
__global__ void cb_rec_fwd_kernel(const float* __restrict__ c, const float* __restrict__ v, float* __restrict__ rec, int BT, int HS, int N) {
    constexpr int E = C / 64;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long)BT * N) return;
    const long bt = row / N, n = row % N;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    for (int h = 0; h < HS; ++h) {
        const float ch = c[(bt * HS + h) * N + n];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = fmaf(ch, v[(bt * HS + h) * C + lane + 64 * e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) rec[row * C + lane + 64 * e] = acc[e];
}

// dc1[bt,h,n] = drec[bt,n,:] . v[bt,h,:], one wave per row   (dv = type1(c, drec))
template <int C>
__global__ void cb_rec_bwd_dc_kernel(const float* __restrict__ drec, const float* __restrict__ v, float* __restrict__ dc1, int BT, int HS, int N) {
    constexpr int E = C / 64;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long)BT * N) return;
    const long bt = row / N, n = row % N;
    float d[E];
#pragma unroll
    for (int e = 0; e < E; ++e) d[e] = drec[row * C + lane + 64 * e];
    for (int h = 0; h < HS; ++h) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) s = fmaf(v[(bt * HS + h) * C + lane + 64 * e], d[e], s);
        s = group_sum<64>(s);
        if (lane == 0) dc1[(bt * HS + h) * N + n] = s;
    }
}

// backward through s = c P, c = softmax_h(b + dadj), P = squash(Y), given Y = X Wp^T + bp (rows of BT*N), one wave per row:
//   U[h] = dS[bt,h,:].P[n,:];  dc = dc1 + U;  dlogit[h] = c[h] (dc[h] - sum_h c dc);  dP = sum_h c[h] dS[h,:];  dY = g dP + Y 2 g'(q) (Y.dP)
template <int C>
__global__ void cb_route_bwd_rows_kernel(const float* __restrict__ Y, const float* __restrict__ c, const float* __restrict__ dc1,
                                         const float* __restrict__ dS, float* __restrict__ dY, float* __restrict__ dlogit, int BT, int HS, int N) {
    constexpr int E = C / 64;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long)BT * N) return;
    const long bt = row / N, n = row % N;
    float y[E], q = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { y[e] = Y[row * C + lane + 64 * e]; q = fmaf(y[e], y[e], q); }
    q = group_sum<64>(q);
    const float rt = sqrtf(q), den = (1.f + q) * (rt + 1e-8f);
    const float g = q / den;
    float gp = 0.f;
    if (rt > 0.f) gp = (den - q * ((rt + 1e-8f) + (1.f + q) * 0.5f / rt)) / (den * den);
    float dp[E], wsum = 0.f, my_c = 0.f, my_dc = 0.f;            // lane h keeps c[h], dc[h]  (HS <= 64)
#pragma unroll
    for (int e = 0; e < E; ++e) dp[e] = 0.f;
    for (int h = 0; h < HS; ++h) {
        float u = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) u = fmaf(dS[(bt * HS + h) * C + lane + 64 * e], g * y[e], u);
        u = group_sum<64>(u);
        const float ch = c[(bt * HS + h) * N + n];
        const float dch = dc1[(bt * HS + h) * N + n] + u;
        wsum = fmaf(ch, dch, wsum);
        if (lane == h) { my_c = ch; my_dc = dch; }
#pragma unroll
        for (int e = 0; e < E; ++e) dp[e] = fmaf(ch, dS[(bt * HS + h) * C + lane + 64 * e], dp[e]);
    }
    if (lane < HS) dlogit[(bt * HS + lane) * N + n] = my_c * (my_dc - wsum);
    float ydp = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) ydp = fmaf(y[e], dp[e], ydp);
    ydp = group_sum<64>(ydp);
    const float k2 = 2.f * gp * ydp;
#pragma unroll
    for (int e = 0; e < E; ++e) dY[row * C + lane + 64 * e] = fmaf(k2, y[e], g * dp[e]);
}

// ---- C ABI ---------------------------------------------------------------------------------------------------------------------
#define CB_ROWS_GRID(rows) dim3((unsigned)(((rows) + 3) / 4)), dim3(256)
#define CB_DISPATCH(KERNEL, GRID, ...)                                                        \
    do {                                                                                      \
        if (C == 64) hipLaunchKernelGGL((KERNEL<64>), GRID, 0, (hipStream_t)stream, __VA_ARGS__);        \
        else if (C == 128) hipLaunchKernelGGL((KERNEL<128>), GRID, 0, (hipStream_t)stream, __VA_ARGS__); \
        else return GPTST_ESHAPE;                                                             \
        GPTST_CHECK_LAUNCH();                                                                 \
        return GPTST_OK;                                                                      \
    } while (0)

// 1 when the one-workgroup-per-(b,t) LDS kernels of cap_mfma.hip / cap.hip hold this shape, 0 when the capbig path is needed
extern "C" int gptst_cap_fits_lds(int N, int C, int HS) {
    const int NR = (N + 15) / 16 * 16, NP = ((N + 3) / 4 * 4) | 1, HSP = (HS + 15) / 16 * 16;
    size_t r2 = (size_t)C * C, need = (size_t)(HS + HSP) * NP, need2 = (size_t)2 * HSP * NP;
    if (need > r2) r2 = need;
    if (need2 > r2) r2 = need2;
    const size_t smem = ((size_t)NR * (C + 4) + r2 + 2 * (size_t)HSP * (C + 4) + (size_t)HSP * C + 2 * (size_t)NR) * sizeof(float);
    return smem <= 160 * 1024 && HS <= 64;
}
extern "C" int gptst_capbig_squash_rows(float* Y, long rows, int C, void* stream) {
    if (!Y) return GPTST_EARG;
    CB_DISPATCH(cb_squash_rows_kernel, CB_ROWS_GRID(rows), Y, rows);
}
extern "C" int gptst_capbig_softmax(const float* bl, const float* l0, float* cs, int BT, int HS, int N, int use_bl, int use_l0, void* stream) {
    if (!cs || (use_bl && !bl) || (use_l0 && !l0)) return GPTST_EARG;
    const long tot = (long)BT * N;
    hipLaunchKernelGGL(cb_softmax_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bl, l0, cs, BT, HS, N, use_bl, use_l0);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
// S (BT,HS,C) = cs (BT,HS,N) . P (BT,N,C)   [overwrites S]
extern "C" int gptst_capbig_type1(const float* cs, const float* P, float* S, int BT, int HS, int N, int C, void* stream) {
    if (!cs || !P || !S) return GPTST_EARG;
    const long tot = (long)BT * HS * C;
    hipLaunchKernelGGL(cb_zero_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S, tot);
    const dim3 grid(BT, (N + CB_CHUNK - 1) / CB_CHUNK, (HS + 15) / 16);
    if (C == 64) hipLaunchKernelGGL((cb_type1_kernel<64>), grid, dim3(256), 0, (hipStream_t)stream, cs, P, S, HS, N);
    else if (C == 128) hipLaunchKernelGGL((cb_type1_kernel<128>), grid, dim3(256), 0, (hipStream_t)stream, cs, P, S, HS, N);
    else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
extern "C" int gptst_capbig_post(const float* S, const float* V0, float* out, long rows, int C, int post, void* stream) {
    if (!S || !out || (post == 2 && !V0)) return GPTST_EARG;
    CB_DISPATCH(cb_post_kernel, CB_ROWS_GRID(rows), S, V0, out, rows, post);
}
extern "C" int gptst_capbig_type2(const float* V, const float* P, float* bl, int BT, int HS, int N, int C, void* stream) {
    if (!V || !P || !bl) return GPTST_EARG;
    const long rows = (long)BT * N;
    if (C == 64) hipLaunchKernelGGL((cb_type2_kernel<64>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, V, P, bl, BT, HS, N);
    else if (C == 128) hipLaunchKernelGGL((cb_type2_kernel<128>), dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, V, P, bl, BT, HS, N);
    else return GPTST_ESHAPE;
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}
extern "C" int gptst_capbig_rec_fwd(const float* c, const float* v, float* rec, int BT, int HS, int N, int C, void* stream) {
    if (!c || !v || !rec) return GPTST_EARG;
    CB_DISPATCH(cb_rec_fwd_kernel, CB_ROWS_GRID((long)BT * N), c, v, rec, BT, HS, N);
}
extern "C" int gptst_capbig_rec_bwd_dc(const float* drec, const float* v, float* dc1, int BT, int HS, int N, int C, void* stream) {
    if (!drec || !v || !dc1) return GPTST_EARG;
    CB_DISPATCH(cb_rec_bwd_dc_kernel, CB_ROWS_GRID((long)BT * N), drec, v, dc1, BT, HS, N);
}
extern "C" int gptst_capbig_route_bwd_rows(const float* Y, const float* c, const float* dc1, const float* dS, float* dY, float* dlogit,
                                           int BT, int HS, int N, int C, void* stream) {
    if (!Y || !c || !dc1 || !dS || !dY || !dlogit || HS > 64) return GPTST_EARG;
    CB_DISPATCH(cb_route_bwd_rows_kernel, CB_ROWS_GRID((long)BT * N), Y, c, dc1, dS, dY, dlogit, BT, HS, N);
}
