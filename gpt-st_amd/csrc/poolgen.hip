// Embedding-conditioned parameter generation:  out[r, :] = sum_k emb[r,k] * pool[k, :]   (K = embed dim <= 16)
//
// This is the reference's  einsum('btd,dio->btio') / einsum('nd,dio->nio') / matmul(emb, bias_pool)
// (GPTST.py:24-25,29-30,137-138,160-161), einsum('btd,dhn->bthn') (:104), einsum('bd,dhk->bhk') (:129) and
// einsum('nk,kht->nht') (:156): a skinny GEMM (R rows <= a few hundred, K <= 16, up to C*C columns).  It is HBM/L2
// bound on the R x cols operand, so it runs on the VALU with coalesced float4 columns; the two gradient reductions are here
// too (fp32 MFMA 16x16x4: the reduction over rows / columns happens inside the matrix core).
//
// Gradient kernels are written around ONE rule learnt from the first profile (profiles/r01a): global atomics are only cheap
// when few land on the same address (a same-address atomic serialises at ~80 ns) and when there are few of them overall —
// so each output element is owned by one thread (or a handful of row chunks), never by hundreds of workgroups.
#include "common.h"
#include "poolgen_dev.h"
#include <algorithm>
#include <vector>

#define PG_ROWS 16     // rows per block of the VALU forward: the K pool rows a thread keeps in registers are re-read from L2 once per block

// V = 4: float4 columns (cols % 4 == 0, rows 16-byte aligned);  V = 1: scalar columns (e.g. HS*N = 2070 for METR_LA)

// ---- job table ------------------------------------------------------------------------------------------------------------
// One launch serves up to PJ_MAX independent problems ("jobs"), each with its OWN embedding, shapes and kind.  A pretraining step has
// ~50 generated-parameter problems in the forward and ~90 gradient reductions in the backward, every one of them a few microseconds
// of work: as one launch per embedding they were ~60 launches of 4-20 us each at the head and the tail of the step (25 % of it,
// profiles/r02b timeline); as job tables they are a handful.  The table travels in the kernel arguments (<= 4 KB).
// Occupancy of the job kernel (r05): one kernel holds every job kind, so its register count is the maximum over the kinds — 152 VGPRs with
// UC = 8 column steps per batch in the embedding-gradient jobs, i.e. 3 waves per SIMD, and the reductions stream at 3.8-4.0 TB/s with 48 KB in
// flight per CU (tools/mb_pooljobs.py).  PJ_OCC = minimum waves per SIMD the compiler must leave room for, PJ_UC / PJ_NU = loads per batch.
#ifndef PJ_OCC
#define PJ_OCC 6       // r05: 76 VGPRs with PJ_UC = 4 (was 152 / 3 waves per SIMD): the forward table 32.8 -> 28.0 us, the reduction table unchanged at the
#endif                 // ~4 TB/s a streaming read reaches on this chip (torch.sum: 3.7-4.9 TB/s), profiles/r05_pooljobs_occupancy.txt
// (PJ_MAX, PJob, PJobs, pj_fwd_mfma: poolgen_dev.h — the cooperative mask launch of masksel.hip carries forward jobs too, r05)

// out[r, :] = sum_k emb[r,k] pool[k, :].   blocks: (ceil(cols/V/256), ceil(R/rows)), rows <= PG_MAXROWS
// The block's emb rows are staged in LDS (zero-padded to PG_MAXK) BEFORE the store loop: read per row from global memory they
// became vector loads whose s_waitcnt vmcnt(0) also waited for the previous row's stores (vmcnt retires in order) — one HBM
// write round trip per row, 30 us for 50 MB where a memset takes 9.
#define PG_MAXROWS 64
template <int V>
__device__ __forceinline__ void pj_fwd(const PJob& a, int bx, int by, int rows, float* __restrict__ embs /* [PG_MAXROWS][PG_MAXK] */) {
    const float* __restrict__ emb = a.emb;
    const float* __restrict__ pool = a.pool;
    float* __restrict__ out = a.out;
    const int cols = a.cols, K = a.K, R = a.R;
    const int r0 = by * rows, nr = min(R, r0 + rows) - r0;
    for (int i = threadIdx.x; i < nr * PG_MAXK; i += 256) {
        const int r = i / PG_MAXK, k = i % PG_MAXK;
        embs[i] = k < K ? emb[(size_t)(r0 + r) * K + k] : 0.f;
    }
    const int c4 = bx * 256 + threadIdx.x;
    const bool ok = V * c4 < cols;
    float4 pv[PG_MAXK];
#pragma unroll
    for (int k = 0; k < PG_MAXK; ++k) pv[k] = (k < K && ok) ? ldv<V>(pool + (size_t)k * cols + V * c4) : f4zero();
    __syncthreads();
    if (!ok) return;
#pragma unroll 2
    for (int r = 0; r < nr; ++r) {
        float4 acc = f4zero();
#pragma unroll
        for (int k = 0; k < PG_MAXK; k += 4) {
            const float4 e = ld4(embs + r * PG_MAXK + k);           // same address in every lane: LDS broadcast
            acc = f4fma(e.x, pv[k], acc); acc = f4fma(e.y, pv[k + 1], acc); acc = f4fma(e.z, pv[k + 2], acc); acc = f4fma(e.w, pv[k + 3], acc);
        }
        stv<V>(out + (size_t)(r0 + r) * cols + V * c4, acc);
    }
}

// The same forward on fp32 MFMA 16x16x4 (cols % 4 == 0): a wave owns a 64-column slab — its K x 64 block of the pool is four float4 B
// fragments per lane (k-step s: pool row 4s+kk, columns c0+4j..+3, component = column tile), read ONCE and kept for every row of the
// block — and walks 16-row tiles: A = emb[row j][4s+kk] (four scalar loads per tile), 16 MFMAs, and accumulator tile ct / register r is
// out[row 4kk+r][c0+4j+ct], i.e. one float4 store per row.  No VALU arithmetic, the pool block is not re-read per 16 rows (the VALU
// version above reads as many bytes from L2 as it writes), and the result is BIT-IDENTICAL to it: an fp32 MFMA is the fmaf chain over
// its four k values in order (MI355X_MICROARCH.md), and the k-steps run in order (tests/test_gpu_kernels.py::test_poolgen_mfma_bitwise).
// blocks: (ceil(cols/256), ceil(R/rows)), rows a multiple of 16.
// pj_fwd_mfma(a, bx, by, rows, tid): poolgen_dev.h

// GRAM: out[r] = A_r^T A_r (T x T, T = 12) of the per-row matrix A_r = (emb . pool)[r] viewed as (cols / 12, 12) — hyperTem's per-node
// temporal graph G_n = A_n^T A_n (GPTST.py:156-158, tmix.hip) straight from the node embedding and the hyperedge pool, so that it does not
// have to wait for (and be launched behind) the job that materialises A.  A workgroup takes pj_gram_rows() rows: the pool is staged in LDS,
// the rows' A are rebuilt there with the forward job's fmaf order over k (bit-identical to reading them from its output), then reduced over
// the hyperedges in gram_fwd's order.   blocks: ceil(R / pj_gram_rows())
// pj_gram_rows(K, cols), pj_gram(a, bx, scr, nt): poolgen_dev.h

// dpool[k, c] += sum_rr emb[rr % R, k] * dW[rr, c]   (rr < R*nsplit: wgrad's K-splits are summed here)
// on fp32 MFMA 16x16x4:  D[i = k][j] += A[i][kk] B[kk][j],  A = emb[row][k],  B = dW[row][col];  lane (kk = l>>4, j = l&15).
// V = 4: a lane fetches the float4 dW[row+kk][c0 + 4j ..] (the four lane groups read four consecutive rows, 256 B each) and
// component e feeds column tile e (columns c0 + 4j + e), so one load drives 4 MFMAs and a wave owns a 64-column slab.
// The reduction over rows happens inside the MFMA; the workgroup's 4 waves take 4 row chunks of the slab and fold through LDS:
// ONE read-modify-write per output element by ONE workgroup (no atomics: the result does not depend on scheduling).
// blocks: (ceil(cols / (16 V)), 1)
// pj_bwd_pool<V>(a, bx, fold, tid), pj_emb_accum<V>(a, bx, chunk, acc, tid), pj_bwd_emb<V>(a, bx, by, tid): poolgen_dev.h

// Deterministic form (gptst_set_deterministic), r05: ONE launch for every embedding gradient of the call.  The table holds the kind-2 jobs grouped
// by their demb (a group = the jobs that add into one demb, in call order; blk0 / nbx of a job = first workgroup / number of 16-row tiles of its
// GROUP); one workgroup of 16 waves owns a 16-row tile of a demb, walks every 256-column chunk of every job of the group (chunks dealt round-robin
// to its waves, each accumulating in order), folds the waves in a fixed order and updates demb with a plain read-modify-write.  (Round 4 ran one
// 4-wave launch per demb, 11 - 24 workgroups each: 12 launches of 18 - 47 us per step.)
#define PJ_DET_T 1024
__global__ __launch_bounds__(PJ_DET_T) void pool_emb_det_kernel(PJobs t) {
    __shared__ float fold[PJ_DET_T / 64][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    int q0 = 0;
    for (int q = 1; q < t.n; ++q) if ((int)blockIdx.x >= t.j[q].blk0) q0 = q;      // last job of the group (uniform scan of the scalar table)
    const int g0 = t.j[q0].blk0, tile = blockIdx.x - g0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int chunk_id = 0, first = q0;
    for (int q = 0; q <= q0; ++q) {
        const PJob& a = t.j[q];
        if (a.blk0 != g0) continue;
        if (q < first) first = q;
        const bool v4 = ((a.cols | a.ldx) & 3) == 0;
        const int nch = (a.cols + 255) / 256;
        for (int ch = 0; ch < nch; ++ch, ++chunk_id) {
            if ((chunk_id & (PJ_DET_T / 64 - 1)) != wave) continue;
            if (v4) pj_emb_accum<4>(a, tile, ch, acc, threadIdx.x); else pj_emb_accum<1>(a, tile, ch, acc, threadIdx.x);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) fold[wave][kk * 4 + r][i] = acc[r];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int R = t.j[first].R, K = t.j[first].K;
        const int rr = threadIdx.x >> 4, k = threadIdx.x & 15, orow = tile * 16 + rr;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < PJ_DET_T / 64; ++w) s += fold[w][rr][k];
        if (orow < R && k < K) t.j[first].out[(size_t)orow * K + k] += s;
    }
}

__global__ __launch_bounds__(256, PJ_OCC) void pool_jobs_kernel(PJobs t, int fwd_rows, int fwd_mfma) {
    __shared__ __attribute__((aligned(16))) float fold[4][PG_MAXK][65];
    int p = 0;
    for (int q = 1; q < t.n; ++q) if ((int)blockIdx.x >= t.j[q].blk0) p = q;      // uniform scan of the (scalar) table
    const PJob& a = t.j[p];
    const int rel = blockIdx.x - a.blk0, bx = rel % a.nbx, by = rel / a.nbx;
    const bool v4 = ((a.cols | a.ldx) & 3) == 0;
    if (a.kind == PJ_FWD) {
        if (v4 && fwd_mfma) pj_fwd_mfma(a, bx, by, fwd_mfma, threadIdx.x);
        else if (v4) pj_fwd<4>(a, bx, by, fwd_rows, &fold[0][0][0]);
        else pj_fwd<1>(a, bx, by, fwd_rows, &fold[0][0][0]);
    }
    else if (a.kind == PJ_GRAM) pj_gram(a, rel, &fold[0][0][0], 256);
    else if (a.kind == PJ_BWD_POOL) { if (v4) pj_bwd_pool<4>(a, bx, fold, threadIdx.x); else pj_bwd_pool<1>(a, bx, fold, threadIdx.x); }
    else { if (v4) pj_bwd_emb<4>(a, bx, by, threadIdx.x); else pj_bwd_emb<1>(a, bx, by, threadIdx.x); }
}

// Host side: fill one job and its block range; returns the number of blocks.
thread_local int g_pg_rows = PG_ROWS;            // experiments: gptst_tune(1, rows)
thread_local int g_pg_sort = 1;                  // gptst_tune(19, 0): keep the caller's job order
thread_local int g_pg_mfma = 1;                  // gptst_tune(10, 0): VALU forward instead of the MFMA one
static int pj_blocks(PJob& j) {
    const int V = ((j.cols | j.ldx) & 3) ? 1 : 4;
    if (j.kind == PJ_FWD && V == 4 && g_pg_mfma) { j.nbx = (j.cols + 255) / 256; return j.nbx * ((j.R + PG_MFMA_ROWS - 1) / PG_MFMA_ROWS); }
    if (j.kind == PJ_FWD) { j.nbx = ((j.cols + V - 1) / V + 255) / 256; return j.nbx * ((j.R + g_pg_rows - 1) / g_pg_rows); }
    if (j.kind == PJ_GRAM) { const int nb = pj_gram_rows(j.K, j.cols); j.nbx = (j.R + nb - 1) / nb; return j.nbx; }
    if (j.kind == PJ_BWD_POOL) { j.nbx = (j.cols + 16 * V - 1) / (16 * V); return j.nbx; }
    j.nbx = (j.R + 15) / 16;
    return j.nbx * (((j.cols + 255) / 256 + 3) / 4);
}

// rows of a GRAM job one workgroup takes for (K, cols) — <= 0: the pool + one row do not fit the kernel's LDS scratch (e.g. more than 20 hyperedges
// at embed_dim 16): the caller then runs gptst_gram_fwd on the forward job's output instead (ops.PoolJobs.gram does)
extern "C" int gptst_pool_jobs_gram_rows(int K, int cols) { return (K <= 0 || K > PG_MAXK || cols <= 0 || cols % 12) ? 0 : pj_gram_rows(K, cols); }

static int pj_launch(PJobs& t, hipStream_t st) {
    int nb = 0;
    for (int p = 0; p < t.n; ++p) {
        PJob& j = t.j[p];
        if (j.R <= 0 || j.K <= 0 || j.K > PG_MAXK || j.cols <= 0 || j.nsplit <= 0 || !j.out) return GPTST_EARG;
        if ((j.kind == PJ_FWD || j.kind == PJ_BWD_POOL || j.kind == PJ_GRAM) && !j.emb) return GPTST_EARG;
        if (j.kind == PJ_GRAM && (!j.pool || j.cols % 12 || pj_gram_rows(j.K, j.cols) <= 0)) return GPTST_EARG;
        if ((j.kind == PJ_BWD_POOL || j.kind == PJ_BWD_EMB) && !j.x) return GPTST_EARG;
        if ((j.kind == PJ_FWD || j.kind == PJ_BWD_EMB) && !j.pool) return GPTST_EARG;
        j.blk0 = nb;
        if (j.ldx <= 0) j.ldx = j.cols;
        if (j.ldx < j.cols) return GPTST_EARG;
        nb += pj_blocks(j);
    }
    if (nb == 0) return GPTST_OK;
    hipLaunchKernelGGL(pool_jobs_kernel, dim3(nb), dim3(256), 0, st, t, g_pg_rows, g_pg_mfma ? PG_MFMA_ROWS : 0);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// Table of forward (MFMA form) and temporal-graph jobs for a launch that embeds them (masksel.hip, gptst_mask_u24_fwd_jobs): forward jobs first, numbered in
// 256-thread blocks (*nvb of them), then the kind-3 jobs, numbered in whole workgroups (*ngw); *nf = number of forward jobs.  GPTST_ESHAPE when a job needs
// another form / the table does not hold them (the caller then runs gptst_pool_jobs).
GPTST_INTERNAL int gptst_pj_embed_table(PJobs* t, int njobs, const int* kind, const void* const* emb, const void* const* pool, const void* const* out,
                                        const int* R, const int* K, const int* cols, int* nf, int* nvb, int* ngw) {
    if (njobs > PJ_MAX || !g_pg_mfma) return GPTST_ESHAPE;
    t->n = 0;
    *nf = *nvb = *ngw = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int p = 0; p < njobs; ++p) {
            const int kd = kind ? kind[p] : (int)PJ_FWD;
            if (kd != PJ_FWD && kd != PJ_GRAM) return GPTST_ESHAPE;
            if ((kd == PJ_GRAM) != (pass == 1)) continue;
            PJob j{(const float*)emb[p], nullptr, (const float*)pool[p], (float*)out[p], R[p], K[p], cols[p], 1, 0, kd, 0, cols[p]};
            if (!j.emb || !j.pool || !j.out || j.R <= 0 || j.K <= 0 || j.K > PG_MAXK || j.cols <= 0) return GPTST_EARG;
            if (kd == PJ_FWD) {
                if (j.cols & 3) return GPTST_ESHAPE;
                j.blk0 = *nvb; *nvb += pj_blocks(j); ++*nf;
            } else {
                if (j.cols % 12 || pj_gram_rows(j.K, j.cols) <= 0) return GPTST_EARG;
                j.blk0 = *ngw; *ngw += pj_blocks(j);
            }
            t->j[t->n++] = j;
        }
    }
    return GPTST_OK;
}

// Table of gradient-reduction jobs (kinds 1 and 2) for a backward launch that carries them as a role (cap_mfma.hip, gptst_cap_cross_route_lin_bwd_jobs): the kind-1
// jobs first, each kind numbered in its own space of 256-thread blocks.  GPTST_ESHAPE: the caller runs gptst_pool_jobs instead (deterministic mode: the kind-2
// jobs have their single-owner launch; another kind; too many jobs).
GPTST_INTERNAL int gptst_pj_reduce_table(PJobs* t, int njobs, const int* kind, const void* const* emb, const void* const* x, const void* const* pool,
                                         const void* const* out, const int* R, const int* K, const int* cols, const int* nsplit, const int* ldx,
                                         int* npool, int* npb, int* neb) {
    if (njobs > PJ_MAX || g_deterministic) return GPTST_ESHAPE;
    t->n = 0;
    *npool = *npb = *neb = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int p = 0; p < njobs; ++p) {
            if (kind[p] != PJ_BWD_POOL && kind[p] != PJ_BWD_EMB) return GPTST_ESHAPE;
            if ((kind[p] == PJ_BWD_EMB) != (pass == 1)) continue;
            PJob j{(const float*)emb[p], (const float*)x[p], (const float*)pool[p], (float*)out[p], R[p], K[p], cols[p], nsplit[p], 0, kind[p], 0, ldx ? ldx[p] : 0};
            if (j.R <= 0 || j.K <= 0 || j.K > PG_MAXK || j.cols <= 0 || j.nsplit <= 0 || !j.out || !j.x) return GPTST_EARG;
            if (kind[p] == PJ_BWD_POOL ? !j.emb : !j.pool) return GPTST_EARG;
            if (j.ldx <= 0) j.ldx = j.cols;
            if (j.ldx < j.cols) return GPTST_EARG;
            if (pass == 0) { j.blk0 = *npb; *npb += pj_blocks(j); ++*npool; }
            else { j.blk0 = *neb; *neb += pj_blocks(j); }
            t->j[t->n++] = j;
        }
    }
    return GPTST_OK;
}

// Generic entry: njobs problems of any kind in as few launches as the 4 KB argument block allows (PJ_MAX jobs each).
// A BWD_EMB job must not share a launch with a job that produces its input; callers order dependent work into separate calls.
extern "C" int gptst_pool_jobs(int njobs, const int* kind, const void* const* emb, const void* const* x, const void* const* pool,
                               const void* const* out, const int* R, const int* K, const int* cols, const int* nsplit, const int* ldx,
                               void* stream) {
    if (njobs < 0 || (njobs && (!kind || !emb || !x || !pool || !out || !R || !K || !cols || !nsplit))) return GPTST_EARG;
    for (int p = 0; p < njobs; ++p) if (kind[p] < PJ_FWD || kind[p] > PJ_GRAM) return GPTST_EARG;
    auto job = [&](int p) {
        return PJob{(const float*)emb[p], (const float*)x[p], (const float*)pool[p], (float*)out[p], R[p], K[p], cols[p], nsplit[p],
                    0, kind[p], 0, ldx ? ldx[p] : 0};
    };
    const bool det = g_deterministic != 0;
    // Workgroups are dispatched in table order: the jobs whose workgroups stream the most bytes go first, so that the launch does not end
    // with a few long workgroups on an otherwise idle chip (the jobs of a call are independent: any order gives the same result).
    std::vector<int> ord(njobs);
    std::vector<long> cost(njobs);
    for (int p = 0; p < njobs; ++p) {
        ord[p] = p;
        const long rr = (long)R[p] * nsplit[p], cc = cols[p];
        cost[p] = kind[p] == PJ_BWD_POOL ? rr * (cc < 64 ? cc : 64) : kind[p] == PJ_BWD_EMB ? 16L * nsplit[p] * (cc < 1024 ? cc : 1024)
                : kind[p] == PJ_FWD ? 64L * (cc < 256 ? cc : 256) : 1L;
    }
    if (g_pg_sort) std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    PJobs t; t.n = 0;
    for (int o = 0; o < njobs; ++o) {                            // table launches (in deterministic mode: everything but the kind-2 jobs)
        const int p = ord[o];
        if (det && kind[p] == PJ_BWD_EMB) continue;
        t.j[t.n++] = job(p);
        if (t.n == PJ_MAX) { const int rc = pj_launch(t, (hipStream_t)stream); if (rc) return rc; t.n = 0; }
    }
    if (t.n) { const int rc = pj_launch(t, (hipStream_t)stream); if (rc) return rc; }
    if (!det) return GPTST_OK;
    // deterministic embedding gradients: the jobs grouped by demb (call order inside a group), as many whole groups per launch as the table holds
    PJobs g; g.n = 0;
    int nb = 0;
    auto flush = [&]() {
        if (g.n) { hipLaunchKernelGGL(pool_emb_det_kernel, dim3(nb), dim3(PJ_DET_T), 0, (hipStream_t)stream, g); g.n = 0; nb = 0; }
    };
    for (int p = 0; p < njobs; ++p) {
        if (kind[p] != PJ_BWD_EMB) continue;
        bool seen = false;
        for (int q = 0; q < p; ++q) if (kind[q] == PJ_BWD_EMB && out[q] == out[p]) seen = true;
        if (seen) continue;
        int members = 0;
        for (int q = p; q < njobs; ++q) if (kind[q] == PJ_BWD_EMB && out[q] == out[p]) ++members;
        if (members > PJ_MAX) return GPTST_EARG;
        if (g.n + members > PJ_MAX) flush();
        const int tiles = (R[p] + 15) / 16;
        for (int q = p; q < njobs; ++q) {
            if (kind[q] != PJ_BWD_EMB || out[q] != out[p]) continue;
            if (R[q] != R[p] || K[q] != K[p]) return GPTST_EARG;
            PJob j = job(q);
            if (!j.x || !j.pool || !j.out || j.R <= 0 || j.K <= 0 || j.K > PG_MAXK || j.cols <= 0 || j.nsplit <= 0) return GPTST_EARG;
            if (j.ldx <= 0) j.ldx = j.cols;
            if (j.ldx < j.cols) return GPTST_EARG;
            j.blk0 = nb; j.nbx = tiles;
            g.j[g.n++] = j;
        }
        nb += tiles;
    }
    flush();
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

extern thread_local int g_wgrad_ns_override;
extern thread_local int g_wgrad_ns0_override;
extern thread_local int g_apply_v1;
extern thread_local int g_apply_tpw;
extern thread_local int g_tl_nb;
extern thread_local int g_wgrad_v1;
extern thread_local int g_apply128_v1;
extern thread_local int g_apply128_minwg;
extern thread_local int g_tl_mfma;
extern thread_local int g_cap_route_v2;
extern thread_local int g_cap_bwd_noroles;
extern thread_local int g_cap_split;
extern thread_local int g_cap_split_roles;
extern "C" int gptst_tune(int id, int value) {
    if (id == 15) g_tl_mfma = value;
    if (id == 20) g_cap_route_v2 = value;
    if (id == 23) g_cap_bwd_noroles = value;
    if (id == 25) g_cap_split = value;
    if (id == 26) g_cap_split_roles = value;
    if (id == 19) g_pg_sort = value;
    if (id == 1 && value > 0 && value <= PG_MAXROWS) g_pg_rows = value;
    if (id == 6 && value > 0) g_tl_nb = value;
    if (id == 2) g_wgrad_ns_override = value;
    if (id == 5) g_wgrad_ns0_override = value;
    if (id == 3) g_apply_v1 = value;
    if (id == 4) g_apply_tpw = value;
    if (id == 7) g_wgrad_v1 = value;
    if (id == 8) g_apply128_v1 = value;
    if (id == 24 && value > 0) g_apply128_minwg = value;
    if (id == 10) g_pg_mfma = value;
    return GPTST_OK;
}

extern "C" int gptst_poolgen_fwd_multi(const float* emb, int nprob, const void* pools, const void* outs, const int* cols, int R, int K,
                                       void* stream) {
    if (!emb || !pools || !outs || !cols || nprob <= 0 || nprob > PJ_MAX || K > PG_MAXK || K <= 0) return GPTST_EARG;
    PJobs t; t.n = nprob;
    for (int p = 0; p < nprob; ++p)
        t.j[p] = PJob{emb, nullptr, ((const float* const*)pools)[p], ((float* const*)outs)[p], R, K, cols[p], 1, 0, PJ_FWD, 0, 0};
    return pj_launch(t, (hipStream_t)stream);
}

extern "C" int gptst_poolgen_bwd_pool_multi(const float* emb, int nprob, const void* dWs, const void* dpools, const int* cols,
                                            const int* nsplit, int R, int K, void* stream) {
    if (!emb || !dWs || !dpools || !cols || nprob <= 0 || nprob > PJ_MAX || K > PG_MAXK || K <= 0) return GPTST_EARG;
    PJobs t; t.n = nprob;
    for (int p = 0; p < nprob; ++p)
        t.j[p] = PJob{emb, ((const float* const*)dWs)[p], nullptr, ((float* const*)dpools)[p], R, K, cols[p], nsplit ? nsplit[p] : 1, 0,
                      PJ_BWD_POOL, 0, 0};
    return pj_launch(t, (hipStream_t)stream);
}

extern "C" int gptst_poolgen_bwd_emb_multi(int nprob, const void* dWs, const void* pools, const int* cols, const int* nsplit, float* demb,
                                           int R, int K, void* stream) {
    if (!dWs || !pools || !cols || !demb || nprob <= 0 || nprob > PJ_MAX || K > PG_MAXK || K <= 0) return GPTST_EARG;
    PJobs t; t.n = nprob;
    for (int p = 0; p < nprob; ++p)
        t.j[p] = PJob{nullptr, ((const float* const*)dWs)[p], ((const float* const*)pools)[p], demb, R, K, cols[p], nsplit ? nsplit[p] : 1, 0,
                      PJ_BWD_EMB, 0, 0};
    return pj_launch(t, (hipStream_t)stream);
}

// ---- one / two problem convenience forms (thin wrappers) -----------------------------------------------------------
extern "C" int gptst_poolgen_fwd(const float* emb, const float* pool, float* out, int cols, const float* pool2, float* out2,
                                 int cols2, int R, int K, void* stream) {
    const float* pools[2] = {pool, pool2}; float* outs[2] = {out, out2}; int cc[2] = {cols, cols2};
    return gptst_poolgen_fwd_multi(emb, pool2 ? 2 : 1, pools, outs, cc, R, K, stream);
}

extern "C" int gptst_poolgen_bwd_pool(const float* emb, const float* dW, float* dpool, int cols, const float* dW2, float* dpool2,
                                      int cols2, int R, int nsplit, int K, void* stream) {
    const float* dWs[2] = {dW, dW2}; float* dps[2] = {dpool, dpool2}; int cc[2] = {cols, cols2}; int ns[2] = {nsplit, 1};
    return gptst_poolgen_bwd_pool_multi(emb, dW2 ? 2 : 1, dWs, dps, cc, ns, R, K, stream);
}

extern "C" int gptst_poolgen_bwd_emb(const float* dW, const float* pool, int cols, const float* dW2, const float* pool2, int cols2,
                                     float* demb, int R, int nsplit, int K, void* stream) {
    const float* dWs[2] = {dW, dW2}; const float* pls[2] = {pool, pool2}; int cc[2] = {cols, cols2}; int ns[2] = {nsplit, 1};
    return gptst_poolgen_bwd_emb_multi(dW2 ? 2 : 1, dWs, pls, cc, ns, demb, R, K, stream);
}
