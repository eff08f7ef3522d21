// Mask generation of the masked autoencoder (reference GPTST.py:314-323 random phase, :344-413 adaptive phase).
//
// The reference zeroes the k largest noise values with  sort(descending) -> idx[:k] -> scatter_(0)  (one global sort per
// selection) and picks whole cluster classes in a host `while` loop with a device->host sync per iteration (:366-369).
// Here every selection is a multi-workgroup radix SELECT on the exact float bits (3 digits of 11/11/10 bits): each digit
// pass is one launch in which every workgroup histograms its slice in LDS and merges into a global histogram; the next
// launch re-derives the running prefix from the finished histograms (256-2048 bins, redundantly per workgroup — cheaper
// than a grid barrier), and a final launch writes the mask.  The result is the same SET as the sort-based one whenever
// the k-th and (k+1)-th values differ; ties straddling rank k go to the lowest indices (torch's sort is unstable there,
// SURVEY.md §7).  Class selection runs on the device from a per-class histogram.  Nothing on this path reads back to
// the host, so it is hipGraph-capturable.  Noise is non-negative (uniform [0,1)): uint32 bit order == float order.
// Masks are fp32 {0,1} arrays (1 = visible, 0 = masked) of B*T*N*base cells — the consumers multiply with them.
#include "common.h"
#include "poolgen_dev.h"
#include <vector>

#define MS_BLOCKS 64                     // workgroups of a selection launch up to 2^16 cells; more cells: one per 1024 cells, up to MS_BLOCKS_MAX
#define MS_BLOCKS_MAX 512
#define MS_THREADS 256
#define MS_BINS 4096
#define MS_BLK (3 * MS_BINS + 16)      // words of one selection's block: three digit histograms + the prefix memo (9 words) + [15] the lattice flag
// float keys: digit d covers bits [SH(d), SH(d)+W(d)):  d0 = 31..21, d1 = 20..10, d2 = 9..0
// u24 (r04): the noise is on the 2^-24 lattice (Philox / torch.rand): keys are the integers k = noise * 2^24 and TWO uniform 12-bit digits serve —
// one launch less per selection.  The lattice is checked where a key is formed; a violation sets word 15 of the block's memo and the mask write
// poisons its output with NaN (see gptst_mask_random_u24).
__device__ __forceinline__ int ms_ndig(int u24) { return u24 ? 2 : 3; }
__device__ __forceinline__ int ms_shift(int d, int u24 = 0) { return u24 ? (d == 0 ? 12 : 0) : (d == 0 ? 21 : (d == 1 ? 10 : 0)); }
__device__ __forceinline__ int ms_bins(int d, int u24 = 0) { return u24 ? 4096 : (d == 2 ? 1024 : 2048); }
__device__ __forceinline__ unsigned ms_fkey(float v, int u24, unsigned& bad) {
    if (!u24) return __float_as_uint(v);
    const unsigned k = (unsigned)(v * 16777216.f);
    bad |= (!(v >= 0.f) || k >= (1u << 24) || (float)k * (1.f / 16777216.f) != v) ? 1u : 0u;
    return k & 0xFFFFFFu;
}

// workspace layout (uint32): hist[3][2048] for selection A/R or the random selection, followed by nothing else.
struct MsPlan {            // what a selection works on
    const int* label;      // adaptive: cluster label per cell (else null)
    const int* counts;     // adaptive: cells per class
    const int* list_c;     // adaptive: shuffled class order
    const int* nums;       // adaptive: {adaptive_mask_num, random_mask_num}
    const float* noise;    // noise of THIS selection
    const float* gate;     // selection R: m_ada (cells with gate == 0 are not eligible); else null
    int mode;              // 0 random phase, 1 adaptive selection A (class gated), 2 adaptive selection R (m_ada gated)
    int ada_all, HS, M, k_const;
    int u24;               // keys are noise * 2^24 (two digits) instead of the float bits (three)
};

struct MsClass {           // class roles of the adaptive phase, derived per workgroup (cheap: <= HS steps)
    unsigned char d[256], f[256];
    int ka;
};

__device__ void ms_classes(const MsPlan& p, MsClass& c) {       // GPTST.py:357-384,393
    for (int h = threadIdx.x; h < 256; h += MS_THREADS) { c.d[h] = 0; c.f[h] = 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int ada_num = p.nums[0];
        int num = 0, i = 0;
        while (num < ada_num && i < p.HS) { num += p.counts[p.list_c[i]]; ++i; }      // :366-369 / :379-382
        int dnum = 0;
        if (p.ada_all && i >= 2) {                                                     // :370-374
            for (int k = 0; k < i - 1; ++k) { c.d[p.list_c[k]] = 1; dnum += p.counts[p.list_c[k]]; }
            c.f[p.list_c[i - 1]] = 1;
        } else {                                                                       // :375-377 / :383-384
            for (int k = 0; k < i; ++k) c.f[p.list_c[k]] = 1;
        }
        c.ka = ada_num - dnum;                                                         // :393
    }
    __syncthreads();
}

__device__ __forceinline__ unsigned ms_key(const MsPlan& p, const MsClass& c, int i, unsigned& bad) {
    const unsigned k = ms_fkey(p.noise[i], p.u24, bad);                                  // (every value is checked, eligible or not)
    if (p.mode == 1) return c.f[p.label[i]] ? k : 0u;                                    // :390
    if (p.mode == 2) return p.gate[i] != 0.f ? k : 0u;                                   // :401
    return k;                                                                            // :316-317
}
__device__ __forceinline__ unsigned ms_key(const MsPlan& p, const MsClass& c, int i) { unsigned bad = 0u; return ms_key(p, c, i, bad); }

__device__ __forceinline__ int ms_rank(const MsPlan& p, const MsClass& c) {
    return p.mode == 0 ? p.k_const : (p.mode == 1 ? c.ka : p.nums[1]);
}

// (prefix, remaining, cnt_eq) after the finished histograms of digits 0..ndig-1.  Every workgroup scans the NEWEST digit only (a block-
// parallel suffix scan: 2048 bins x 64 workgroups as a thread-0 loop adds up); the state after the earlier digits comes from `memo`
// (3 words per digit behind the histograms), written by workgroup 0 of the launch that scanned that digit.
__device__ void ms_prefix(const unsigned* __restrict__ hist, int ndig, int k, unsigned* sh /* >= MS_BINS+8 */, unsigned& prefix,
                          unsigned& remaining, unsigned& cnt_eq, int u24 = 0) {
    prefix = 0u; remaining = (unsigned)k; cnt_eq = 0u;
    if (ndig <= 0) return;
    unsigned* memo = const_cast<unsigned*>(hist) + 3 * MS_BINS;
    if (ndig >= 2) { prefix = memo[3 * (ndig - 2)]; remaining = memo[3 * (ndig - 2) + 1]; cnt_eq = memo[3 * (ndig - 2) + 2]; }
    for (int d = ndig - 1; d < ndig; ++d) {
        const int nb = ms_bins(d, u24);
        const unsigned* h = hist + d * MS_BINS;
        // each thread owns nb/256 consecutive bins (descending order = ascending "rank from the top")
        const int per = nb / MS_THREADS;
        unsigned loc = 0u;
        for (int j = 0; j < per; ++j) loc += h[nb - 1 - (threadIdx.x * per + j)];
        // exclusive prefix over the 256 threads: shuffles inside a wave + the four wave totals through LDS (two barriers; the Hillis-Steele
        // scan over LDS this replaces took sixteen — ~1.5 us in each of a step's six selection launches)
        unsigned inc = loc;
        {
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const unsigned v = __shfl_up(inc, off, 64); if (lane >= off) inc += v; }
            __syncthreads();                                         // (sh of an earlier use has been read)
            if (lane == 63) sh[threadIdx.x >> 6] = inc;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < MS_THREADS / 64; ++w) if (w < (int)(threadIdx.x >> 6)) inc += sh[w];
        }
        const unsigned before = inc - loc;                       // keys in bins above this thread's range
        if (before < remaining && before + loc >= remaining) {   // the threshold bin is in this thread's range (exactly one thread)
            unsigned cum = before;
            for (int j = 0; j < per; ++j) {
                const int b = nb - 1 - (threadIdx.x * per + j);
                if (cum + h[b] >= remaining) { sh[MS_THREADS] = (unsigned)b; sh[MS_THREADS + 1] = remaining - cum; sh[MS_THREADS + 2] = h[b]; break; }
                cum += h[b];
            }
        }
        __syncthreads();
        prefix |= sh[MS_THREADS] << ms_shift(d, u24);
        remaining = sh[MS_THREADS + 1];
        cnt_eq = sh[MS_THREADS + 2];
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { memo[3 * (ndig - 1)] = prefix; memo[3 * (ndig - 1) + 1] = remaining; memo[3 * (ndig - 1) + 2] = cnt_eq; }
}

// one digit pass: histogram of digit `dig` over the keys that match the prefix of the earlier digits
__global__ __launch_bounds__(MS_THREADS) void ms_hist_kernel(MsPlan p, unsigned* __restrict__ hist, int dig) {
    __shared__ unsigned lh[MS_BINS];
    __shared__ unsigned sc[MS_THREADS + 8];
    __shared__ MsClass cls;
    if (p.mode == 1) ms_classes(p, cls);
    const int k = ms_rank(p, cls);
    if (k <= 0) return;                                          // uniform: nothing to select
    unsigned prefix, remaining, cnt_eq, bad = 0u;
    ms_prefix(hist, dig, k, sc, prefix, remaining, cnt_eq, p.u24);
    const int sh = ms_shift(dig, p.u24), nb = ms_bins(dig, p.u24);
    for (int b = threadIdx.x; b < nb; b += MS_THREADS) lh[b] = 0u;
    __syncthreads();
    const unsigned hi_mask = dig == 0 ? 0u : (p.u24 ? 0xFFFFF000u : (0xFFFFFFFFu << (sh + (dig == 1 ? 11 : 10))));
    // four cells per trip: their (noise, label / gate) loads are issued together (one cell per trip = one serialised L2 round trip per trip)
    for (int i0 = blockIdx.x * MS_THREADS + threadIdx.x; i0 < p.M; i0 += 4 * gridDim.x * MS_THREADS) {
        unsigned key[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * gridDim.x * MS_THREADS; key[u] = i < p.M ? ms_key(p, cls, i, bad) : 0u; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * gridDim.x * MS_THREADS < p.M && (key[u] & hi_mask) == (prefix & hi_mask)) atomicAdd(&lh[(key[u] >> sh) & (unsigned)(nb - 1)], 1u);
    }
    __syncthreads();
    unsigned* gh = hist + dig * MS_BINS;
    for (int b = threadIdx.x; b < nb; b += MS_THREADS) if (lh[b]) atomicAdd(gh + b, lh[b]);
    if (bad) atomicOr(hist + 3 * MS_BINS + 15, 1u);                  // a noise value off the lattice: the mask write poisons its output
}

// final launch of a selection: write the {0,1} mask of this selection (and the combined mask for selection R)
//   mode 0: out = mask (M = B*T*N*base cells)
//   mode 1: out = m_ada  (also zeroes the fully masked classes, :397)
//   mode 2: out = m_rnd, final[i*base + j] = gate[i] * m_rnd[i]   (:411-413)
//   next_noise != NULL (mode 1 only): digit 0 of selection R — whose keys are next_noise gated by THIS launch's output — is histogrammed
//   here into next_hist (zeroed), so selection R starts at digit 1: one launch less.
__global__ __launch_bounds__(MS_THREADS) void ms_apply_kernel(MsPlan p, const unsigned* __restrict__ hist, float* __restrict__ out,
                                                              float* __restrict__ final_mask, int base, const float* __restrict__ next_noise,
                                                              unsigned* __restrict__ next_hist) {
    __shared__ unsigned sc[MS_THREADS + 8];
    __shared__ MsClass cls;
    __shared__ unsigned s_base;
    __shared__ unsigned lh[MS_BINS];
    const bool nxt = next_noise != nullptr && p.nums[1] > 0;        // uniform
    const int nb0 = ms_bins(0, p.u24);
    if (nxt) { for (int b = threadIdx.x; b < nb0; b += MS_THREADS) lh[b] = 0u; }
    if (p.mode == 1) ms_classes(p, cls);
    const int k = ms_rank(p, cls);
    unsigned thr = 0xFFFFFFFFu, need = 0u, cnt_eq = 0u, bad = 0u, bad2 = 0u;
    if (k > 0) ms_prefix(hist, ms_ndig(p.u24), k, sc, thr, need, cnt_eq, p.u24);
    const bool ties = k > 0 && need != cnt_eq;                   // uniform
    // u24: a value off the lattice was seen by a digit pass of this selection (or, selection R: by selection A, whose block sits MS_BLK words
    // below) -> every output of this launch is NaN
    const bool poison = p.u24 && (hist[3 * MS_BINS + 15] != 0u || (p.mode == 2 && (hist - MS_BLK)[3 * MS_BINS + 15] != 0u));
    const float one = poison ? __int_as_float(0x7fc00000) : 1.f, zero = poison ? __int_as_float(0x7fc00000) : 0.f;
    for (int i0 = blockIdx.x * MS_THREADS + threadIdx.x; i0 < p.M; i0 += 4 * gridDim.x * MS_THREADS) {       // four cells' loads in flight
        unsigned keys[4], nk[4];
        int lab[4];
        float gat[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * (int)gridDim.x * MS_THREADS, p.M - 1);
            keys[u] = ms_key(p, cls, i, bad);
            lab[u] = p.mode == 1 ? p.label[i] : 0;
            gat[u] = p.mode == 2 ? p.gate[i] : 0.f;
            nk[u] = nxt ? ms_fkey(next_noise[i], p.u24, bad2) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * gridDim.x * MS_THREADS;
            if (i >= p.M) break;
            const unsigned key = keys[u];
            // a tie straddling rank k: threshold-equal cells belong to workgroup 0 alone (below) — nobody else writes them, so the
            // result does not depend on the order in which workgroups finish
            if (ties && key == thr) continue;
            float vis = 1.f;
            if (k > 0 && (key > thr || (key == thr && !ties))) vis = 0.f;
            if (p.mode == 1 && cls.d[lab[u]]) vis = 0.f;
            out[i] = vis != 0.f ? one : zero;
            if (p.mode == 2) { const float f = gat[u] * (vis != 0.f ? one : zero); for (int j = 0; j < base; ++j) final_mask[(size_t)i * base + j] = f; }
            if (nxt) atomicAdd(&lh[(vis != 0.f ? nk[u] : 0u) >> ms_shift(0, p.u24)], 1u);
        }
    }
    const bool tie_wg = ties && blockIdx.x == 0;
    if (p.u24 && bad) atomicOr(const_cast<unsigned*>(hist) + 3 * MS_BINS + 15, 1u);      // (seen only here: the flag serves the launches behind this one)
    if (p.u24 && bad2 && next_hist) atomicOr(next_hist + 3 * MS_BINS + 15, 1u);
    if (!tie_wg) {
        if (nxt) { __syncthreads(); for (int b = threadIdx.x; b < nb0; b += MS_THREADS) if (lh[b]) atomicAdd(next_hist + b, lh[b]); }
        return;
    }
    // rare: a tie straddles rank k.  Workgroup 0 hands the `need` threshold-equal slots out in index order and writes BOTH
    // values (0 for the first `need` of them, 1 for the rest).
    // (A thread takes TPT consecutive cells, so one block scan orders TPT * 256 of them: the scan over all M cells by this one workgroup
    //  is what the path costs — 1 ms at 2.6e5 cells with one cell per thread, which matters once a data-parallel job selects over
    //  world x B*T*N cells with 24-bit noise: a straddling tie every ~30 steps at 5e5 cells.)
    constexpr int TPT = 16;
    if (threadIdx.x == 0) s_base = 0u;
    __syncthreads();
    for (int i0 = 0; i0 < p.M; i0 += TPT * MS_THREADS) {
        const int ib = i0 + TPT * threadIdx.x;
        unsigned eqm = 0u;                                           // bit u: cell ib + u is threshold-equal
#pragma unroll
        for (int u = 0; u < TPT; ++u) if (ib + u < p.M && ms_key(p, cls, ib + u) == thr) eqm |= 1u << u;
        const unsigned cnt = __popc(eqm);
        sc[threadIdx.x] = cnt;
        __syncthreads();
        for (int off = 1; off < MS_THREADS; off <<= 1) {
            const unsigned v = threadIdx.x >= off ? sc[threadIdx.x - off] : 0u;
            __syncthreads();
            sc[threadIdx.x] += v;
            __syncthreads();
        }
        unsigned rank = s_base + sc[threadIdx.x] - cnt;
        for (unsigned m = eqm; m; m &= m - 1u, ++rank) {
            const int i = ib + (__ffs(m) - 1);
            float vis = rank < need ? 0.f : 1.f;
            if (p.mode == 1 && cls.d[p.label[i]]) vis = 0.f;
            out[i] = vis != 0.f ? one : zero;
            if (p.mode == 2) { const float f = p.gate[i] * (vis != 0.f ? one : zero); for (int j = 0; j < base; ++j) final_mask[(size_t)i * base + j] = f; }
            if (nxt) { unsigned b3 = 0u; atomicAdd(&lh[(vis != 0.f ? ms_fkey(next_noise[i], p.u24, b3) : 0u) >> ms_shift(0, p.u24)], 1u); }
        }
        __syncthreads();
        if (threadIdx.x == MS_THREADS - 1) s_base += sc[threadIdx.x];
        __syncthreads();
    }
    if (nxt) { __syncthreads(); for (int b = threadIdx.x; b < nb0; b += MS_THREADS) if (lh[b]) atomicAdd(next_hist + b, lh[b]); }
}

// ======================================================================================================================
// Small problems (M <= MSS_MAXM = 8192 cells): the WHOLE mask generation of a step — class histogram, class roles, both radix
// selects and the mask writes — runs as ONE launch of ONE 1024-thread workgroup instead of 12 launches.  Measured at the bench
// shape (65 280 cells): 234 us against 64 us for the multi-launch path — one CU re-reading the keys nine times at ~60 GB/s and
// the skewed float-bit digits (half of all keys in four bins of the top digit, every ineligible cell in bin 0) serialising the LDS
// atomics — so the single launch is only used where the data is a few KB.
// ======================================================================================================================
#define MSS_T 1024
#define MSS_MAXM (1 << 13)

struct MssShared {
    unsigned hist[MS_BINS];
    unsigned sc[MSS_T + 8];
    int counts[256];
    unsigned s_base;
    MsClass cls;
};

// threshold of rank k (from the top) among the keys of plan p: -> thr, need (how many threshold-equal keys are selected), cnt_eq
__device__ void mss_select(const MsPlan& p, MssShared& sh, int k, unsigned& thr, unsigned& need, unsigned& cnt_eq) {
    unsigned prefix = 0u, remaining = (unsigned)k;
    cnt_eq = 0u;
    for (int d = 0; d < 3; ++d) {
        const int shf = ms_shift(d), nb = ms_bins(d);
        const unsigned hi_mask = d == 0 ? 0u : (0xFFFFFFFFu << (shf + (d == 1 ? 11 : 10)));
        for (int b = threadIdx.x; b < MS_BINS; b += MSS_T) sh.hist[b] = 0u;
        __syncthreads();
        for (int i0 = threadIdx.x; i0 < p.M; i0 += 8 * MSS_T) {
            unsigned key[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * MSS_T; key[u] = i < p.M ? ms_key(p, sh.cls, i) : 0u; }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * MSS_T < p.M && (key[u] & hi_mask) == (prefix & hi_mask)) atomicAdd(&sh.hist[(key[u] >> shf) & (unsigned)(nb - 1)], 1u);
        }
        __syncthreads();
        // suffix scan over the bins (descending key order): thread t owns bins nb-1 - (t*per + j)
        const int per = (nb + MSS_T - 1) / MSS_T;
        unsigned loc = 0u;
        for (int j = 0; j < per; ++j) { const int b = nb - 1 - (threadIdx.x * per + j); if (b >= 0) loc += sh.hist[b]; }
        sh.sc[threadIdx.x] = loc;
        __syncthreads();
        for (int off = 1; off < MSS_T; off <<= 1) {
            const unsigned v = threadIdx.x >= off ? sh.sc[threadIdx.x - off] : 0u;
            __syncthreads();
            sh.sc[threadIdx.x] += v;
            __syncthreads();
        }
        const unsigned before = sh.sc[threadIdx.x] - loc;
        if (before < remaining && before + loc >= remaining) {        // exactly one thread
            unsigned cum = before;
            for (int j = 0; j < per; ++j) {
                const int b = nb - 1 - (threadIdx.x * per + j);
                if (b < 0) break;
                if (cum + sh.hist[b] >= remaining) { sh.sc[MSS_T] = (unsigned)b; sh.sc[MSS_T + 1] = remaining - cum; sh.sc[MSS_T + 2] = sh.hist[b]; break; }
                cum += sh.hist[b];
            }
        }
        __syncthreads();
        prefix |= sh.sc[MSS_T] << shf;
        remaining = sh.sc[MSS_T + 1];
        cnt_eq = sh.sc[MSS_T + 2];
        __syncthreads();
    }
    thr = prefix; need = remaining;
}

// write the {0,1} result of one selection (same semantics as ms_apply_kernel, one workgroup: ties are resolved in index order here)
__device__ void mss_apply(const MsPlan& p, MssShared& sh, int k, unsigned thr, unsigned need, unsigned cnt_eq, float* __restrict__ out,
                          float* __restrict__ final_mask, int base) {
    const bool ties = k > 0 && need != cnt_eq;
    for (int i = threadIdx.x; i < p.M; i += MSS_T) {
        const unsigned key = ms_key(p, sh.cls, i);
        float vis = 1.f;
        if (k > 0 && (key > thr || (key == thr && !ties))) vis = 0.f;
        if (p.mode == 1 && sh.cls.d[p.label[i]]) vis = 0.f;
        out[i] = vis;
        if (p.mode == 2) { const float f = p.gate[i] * vis; for (int j = 0; j < base; ++j) final_mask[(size_t)i * base + j] = f; }
    }
    if (!ties) return;                                               // uniform
    __syncthreads();
    if (threadIdx.x == 0) sh.s_base = 0u;
    __syncthreads();
    for (int i0 = 0; i0 < p.M; i0 += MSS_T) {
        const int i = i0 + threadIdx.x;
        const bool eq = i < p.M && ms_key(p, sh.cls, i) == thr;
        sh.sc[threadIdx.x] = eq ? 1u : 0u;
        __syncthreads();
        for (int off = 1; off < MSS_T; off <<= 1) {
            const unsigned v = threadIdx.x >= off ? sh.sc[threadIdx.x - off] : 0u;
            __syncthreads();
            sh.sc[threadIdx.x] += v;
            __syncthreads();
        }
        const unsigned rank = sh.s_base + sh.sc[threadIdx.x] - (eq ? 1u : 0u);
        if (eq && rank < need) {
            out[i] = 0.f;
            if (p.mode == 2) for (int j = 0; j < base; ++j) final_mask[(size_t)i * base + j] = 0.f;
        }
        __syncthreads();
        if (threadIdx.x == MSS_T - 1) sh.s_base += sh.sc[threadIdx.x];
        __syncthreads();
        if (sh.s_base >= need) break;                                // uniform
    }
}

__global__ __launch_bounds__(MSS_T) void mss_random_kernel(MsPlan p, float* __restrict__ mask) {
    __shared__ MssShared sh;
    unsigned thr = 0xFFFFFFFFu, need = 0u, cnt_eq = 0u;
    if (p.k_const > 0) mss_select(p, sh, p.k_const, thr, need, cnt_eq);
    mss_apply(p, sh, p.k_const, thr, need, cnt_eq, mask, nullptr, 1);
}

// counts_in may be null: the class histogram is then taken from the labels here (what mask_labels_kernel would have produced)
__global__ __launch_bounds__(MSS_T) void mss_adaptive_kernel(MsPlan a, const float* __restrict__ noise_r, const int* __restrict__ counts_in,
                                                             float* __restrict__ m_ada, float* __restrict__ m_rnd,
                                                             float* __restrict__ mask, int base) {
    __shared__ MssShared sh;
    for (int h = threadIdx.x; h < 256; h += MSS_T) sh.counts[h] = (counts_in && h < a.HS) ? counts_in[h] : 0;
    __syncthreads();
    if (!counts_in) {
        for (int i = threadIdx.x; i < a.M; i += MSS_T) atomicAdd(&sh.counts[a.label[i]], 1);
        __syncthreads();
    }
    a.counts = sh.counts;
    // class roles (ms_classes with this launch's thread count)
    for (int h = threadIdx.x; h < 256; h += MSS_T) { sh.cls.d[h] = 0; sh.cls.f[h] = 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int ada_num = a.nums[0];
        int num = 0, i = 0;
        while (num < ada_num && i < a.HS) { num += sh.counts[a.list_c[i]]; ++i; }        // GPTST.py:366-369 / :379-382
        int dnum = 0;
        if (a.ada_all && i >= 2) {                                                        // :370-374
            for (int k = 0; k < i - 1; ++k) { sh.cls.d[a.list_c[k]] = 1; dnum += sh.counts[a.list_c[k]]; }
            sh.cls.f[a.list_c[i - 1]] = 1;
        } else {                                                                          // :375-377 / :383-384
            for (int k = 0; k < i; ++k) sh.cls.f[a.list_c[k]] = 1;
        }
        sh.cls.ka = ada_num - dnum;                                                       // :393
    }
    __syncthreads();
    unsigned thr = 0xFFFFFFFFu, need = 0u, cnt_eq = 0u;
    const int ka = sh.cls.ka;
    if (ka > 0) mss_select(a, sh, ka, thr, need, cnt_eq);                                 // :386-397
    mss_apply(a, sh, ka, thr, need, cnt_eq, m_ada, nullptr, base);
    __syncthreads();                                   // m_ada (global) is read back by other threads of this workgroup below
    MsPlan r = a;
    r.noise = noise_r; r.gate = m_ada; r.mode = 2;
    const int kr = a.nums[1];
    thr = 0xFFFFFFFFu; need = 0u; cnt_eq = 0u;
    if (kr > 0) mss_select(r, sh, kr, thr, need, cnt_eq);                                 // :399-413
    mss_apply(r, sh, kr, thr, need, cnt_eq, m_rnd, mask, base);
}

// label[i] = argmax_h prob[i, h] (first maximum), counts[h] += 1      (GPTST.py:344-345)
// counts: per-workgroup LDS histogram, then one global atomic per class and workgroup (same-address global atomics serialise).
__global__ __launch_bounds__(256) void mask_labels_kernel(const float* __restrict__ prob, int rows, int HS,
                                                          int* __restrict__ label, int* __restrict__ counts) {
    __shared__ int hist[256];
    for (int h = threadIdx.x; h < HS; h += 256) hist[h] = 0;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) {
        const float* p = prob + (size_t)i * HS;
        float best = p[0];
        int bi = 0;
        for (int h = 1; h < HS; ++h) { const float v = p[h]; if (v > best) { best = v; bi = h; } }
        label[i] = bi;
        atomicAdd(&hist[bi], 1);
    }
    __syncthreads();
    for (int h = threadIdx.x; h < HS; h += 256) if (hist[h]) atomicAdd(counts + h, hist[h]);
}

// zeroing by kernel rather than hipMemsetAsync: memset nodes captured in a hipGraph were observed not to re-execute
// reliably on replay (ROCm 7.0 runtime bundled with torch), which silently doubled the histograms.
__global__ __launch_bounds__(256) void ms_zero_kernel(unsigned* __restrict__ p, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = 0u;
}

// one selection on a ZEROED histogram block (digit passes d0..2 + the mask write); next_noise / next_hist: see ms_apply_kernel
static int ms_select(const MsPlan& p, unsigned* hist, float* out, float* final_mask, int base, hipStream_t st, int d0 = 0,
                     const float* next_noise = nullptr, unsigned* next_hist = nullptr) {
    // one trip of four cells per thread up to MS_BLOCKS_MAX workgroups: a data-parallel job selects over world x B*T*N cells on every rank,
    // and with a fixed 64 workgroups the generation grew from 50 to 112 us at 8 x 65280 cells (tools/experiments/mb_mask_scaling.py)
    int nbk = (p.M + 4 * MS_THREADS - 1) / (4 * MS_THREADS);
    nbk = nbk < MS_BLOCKS ? MS_BLOCKS : (nbk > MS_BLOCKS_MAX ? MS_BLOCKS_MAX : nbk);
    for (int d = d0; d < (p.u24 ? 2 : 3); ++d) hipLaunchKernelGGL(ms_hist_kernel, dim3(nbk), dim3(MS_THREADS), 0, st, p, hist, d);
    hipLaunchKernelGGL(ms_apply_kernel, dim3(nbk), dim3(MS_THREADS), 0, st, p, (const unsigned*)hist, out, final_mask, base, next_noise,
                       next_hist);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

// counts[h] += number of cells with label h (LDS histogram per workgroup, one global atomic per class and workgroup); counts zeroed before
__global__ __launch_bounds__(256) void ms_count_kernel(const int* __restrict__ label, int M, int HS, int* __restrict__ counts) {
    __shared__ int hist[256];
    for (int h = threadIdx.x; h < HS; h += 256) hist[h] = 0;
    __syncthreads();
    for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < M; i0 += 8 * gridDim.x * 256) {          // eight labels' loads in flight per trip
        int lab[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) lab[u] = label[min(i0 + u * (int)gridDim.x * 256, M - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + u * (int)gridDim.x * 256 < M) atomicAdd(&hist[lab[u]], 1);
    }
    __syncthreads();
    for (int h = threadIdx.x; h < HS; h += 256) if (hist[h]) atomicAdd(counts + h, hist[h]);
}

// ws: device scratch of gptst_mask_ws_bytes() bytes: [block of selection A / the random selection | block of selection R |
// class counts (256)] — ONE zeroing launch per mask generation covers all of it (none when the caller hands it over zeroed)
#define MS_WS_WORDS (2 * MS_BLK + 256)
extern "C" int gptst_mask_ws_bytes(void) { return (int)(sizeof(unsigned) * MS_WS_WORDS); }

int g_ms_force_multi = 0;       // tests: 1 = take the multi-launch path for every size

#ifndef MS_COOP_DEFAULT
#define MS_COOP_DEFAULT 1
#endif
extern thread_local int g_ms_coop;
// 0 default; 1: multi-launch path for every size; 2 (u24): the one-workgroup lattice kernel up to 65536 cells; 3 (u24): the multi-launch path where
// the cooperative launch (r05) would serve, sizes up to 8192 cells still on the one-workgroup kernel
extern "C" int gptst_mask_force_multi(int on) { g_ms_force_multi = on; return GPTST_OK; }
extern "C" int gptst_mask_cooperative(int on) { g_ms_coop = on < 0 ? MS_COOP_DEFAULT : (on != 0); return GPTST_OK; }
extern "C" int gptst_mask_cooperative_state(void) { return g_ms_coop; }

static int ms_random_impl(const float* noise, int M, int k, float* mask, void* ws, int ws_zeroed, void* stream, int u24) {
    if (!noise || !mask || !ws || M <= 0 || k < 0 || k > M) return GPTST_EARG;
    MsPlan p{nullptr, nullptr, nullptr, nullptr, noise, nullptr, 0, 0, 0, M, k, u24};
    if (M <= MSS_MAXM && !g_ms_force_multi && !u24) {
        hipLaunchKernelGGL(mss_random_kernel, dim3(1), dim3(MSS_T), 0, (hipStream_t)stream, p, mask);
        GPTST_CHECK_LAUNCH();
        return GPTST_OK;
    }
    if (!ws_zeroed) hipLaunchKernelGGL(ms_zero_kernel, dim3(12), dim3(256), 0, (hipStream_t)stream, (unsigned*)ws, MS_BLK);
    return ms_select(p, (unsigned*)ws, mask, nullptr, 1, (hipStream_t)stream);
}
extern "C" int gptst_mask_random(const float* noise, int M, int k, float* mask, void* ws, int ws_zeroed, void* stream) {
    return ms_random_impl(noise, M, k, mask, ws, ws_zeroed, stream, 0);
}

extern "C" int gptst_mask_labels(const float* prob, int rows, int HS, int* label, int* counts, void* stream) {
    if (!prob || !label || !counts || HS <= 0 || HS > 256) return GPTST_EARG;
    hipLaunchKernelGGL(ms_zero_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (unsigned*)counts, HS);
    int nb = (rows + 255) / 256; if (nb > 64) nb = 64;
    hipLaunchKernelGGL(mask_labels_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, prob, rows, HS, label, counts);
    GPTST_CHECK_LAUNCH();
    return GPTST_OK;
}

static int ms_adaptive_impl(const int* label, const int* counts, const int* list_c, const int* nums, const float* noise_a,
                            const float* noise_r, int ada_all, int M, int HS, int base, float* m_ada, float* m_rnd,
                            float* mask, void* ws, int ws_zeroed, void* stream, int u24) {
    if (!label || !list_c || !nums || !noise_a || !noise_r || !m_ada || !m_rnd || !mask || !ws || HS > 256) return GPTST_EARG;
    MsPlan a{label, counts, list_c, nums, noise_a, nullptr, 1, ada_all, HS, M, 0, u24};
    if (M <= MSS_MAXM && !g_ms_force_multi && !u24) {        // counts may be NULL here: the single-workgroup kernel histograms the labels itself
        hipLaunchKernelGGL(mss_adaptive_kernel, dim3(1), dim3(MSS_T), 0, (hipStream_t)stream, a, noise_r, counts, m_ada, m_rnd, mask, base);
        GPTST_CHECK_LAUNCH();
        return GPTST_OK;
    }
    unsigned* w = (unsigned*)ws;
    if (!ws_zeroed) hipLaunchKernelGGL(ms_zero_kernel, dim3(24), dim3(256), 0, (hipStream_t)stream, w, MS_WS_WORDS);   // both blocks + counts
    if (!counts) {                                         // class histogram from the labels (what gptst_mask_labels would have produced)
        int nb = (M + 8 * 256 - 1) / (8 * 256); if (nb > 128) nb = 128;                      // <= 128 same-address atomics per class
        int* cw = (int*)(w + 2 * MS_BLK);
        hipLaunchKernelGGL(ms_count_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, label, M, HS, cw);
        a.counts = cw;
    }
    // selection A's mask write also histograms digit 0 of selection R (whose keys it gates): R starts at digit 1
    int rc = ms_select(a, w, m_ada, nullptr, base, (hipStream_t)stream, 0, noise_r, w + MS_BLK);           // :386-397
    if (rc) return rc;
    MsPlan r{label, a.counts, list_c, nums, noise_r, m_ada, 2, ada_all, HS, M, 0, u24};
    return ms_select(r, w + MS_BLK, m_rnd, mask, base, (hipStream_t)stream, 1);                            // :399-413
}
extern "C" int gptst_mask_adaptive(const int* label, const int* counts, const int* list_c, const int* nums, const float* noise_a,
                                   const float* noise_r, int ada_all, int M, int HS, int base, float* m_ada, float* m_rnd,
                                   float* mask, void* ws, int ws_zeroed, void* stream) {
    return ms_adaptive_impl(label, counts, list_c, nums, noise_a, noise_r, ada_all, M, HS, base, m_ada, m_rnd, mask, ws, ws_zeroed, stream, 0);
}

// ======================================================================================================================
// 24-bit noise (r04): the WHOLE mask generation of a step as ONE launch of ONE 1024-thread workgroup, keys in REGISTERS.
//
// The step's own noise is Philox (u32 >> 8) * 2^-24 and torch.rand draws the same lattice: every value is k * 2^-24 with an integer k < 2^24,
// and ordering the floats = ordering the integers.  On the integers the radix select needs TWO uniform 12-bit digits instead of three skewed
// float-bit digits (half of all keys share four bins of the float's top digit — the LDS atomics of the r02 single-workgroup attempt serialised
// on them: 234 us at 65 280 cells), and a thread keeps its 64 cells' keys in registers, so nothing is re-read between the passes.  The
// multi-launch path above costs 8 launches of ~5.5 us in the adaptive phase (4 in the random phase) — latency, not work; this is one.
//   precondition: noise values on the 2^-24 lattice in [0, 1).  Checked on the device: any other value poisons the whole mask with NaN (the loss
//   turns NaN at once) instead of selecting on rounded keys.  Callers with arbitrary float noise use gptst_mask_random / gptst_mask_adaptive.
//   M <= 65536 cells (64 per thread); ties at the threshold are resolved in index order exactly as above.
// ======================================================================================================================
#define MU_T 1024
#define MU_CPT 64
#define MU_BINS 4096
typedef unsigned long long mu_u64;

// LDS: the TOP 12-bit digit of every cell's key as 16 bits (bit 15: the key is exactly 0 — ineligible cell or zero noise) = 128 KB for 65 536
// cells; the low digit is needed only for the handful of cells that share the threshold's top digit, and those re-read their noise.  (A first
// form kept the 64 keys of a thread in registers: fully unrolled loops over a 64-entry register array at 128 VGPRs spilled 140-500 registers.)
struct MuShared {
    unsigned short top[MU_T * MU_CPT];
    unsigned hist[MU_BINS];         // (first use: the class histogram per wave, [16][256] ints — one address per class would serialise the
                                    //  whole workgroup's LDS atomics)
    unsigned wsum[MU_T / 64 + 1];
    unsigned res[4];                // threshold bin, remaining, count in the bin
    int counts[256];
    unsigned bad, base_rank;
    MsClass cls;
};

__device__ __forceinline__ unsigned mu_key(float v, unsigned& bad) {
    const unsigned k = (unsigned)(v * 16777216.f);
    bad |= (!(v >= 0.f) || k >= (1u << 24) || (float)k * (1.f / 16777216.f) != v) ? 1u : 0u;
    return k & 0xFFFFFFu;
}

__device__ __forceinline__ unsigned mu_wave_sum(unsigned v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// exclusive prefix of `loc` over the 1024 threads in thread order (wave shuffles + one LDS hop); -> before, and the block total in `total`
__device__ __forceinline__ unsigned mu_scan(unsigned loc, MuShared& sh, unsigned& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = loc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned v = __shfl_up(inc, off, 64); if (lane >= off) inc += v; }
    __syncthreads();                                        // (wsum of the previous scan has been read by everyone)
    if (lane == 63) sh.wsum[wave] = inc;
    __syncthreads();
    unsigned before = inc - loc, tot = 0u;
#pragma unroll
    for (int w = 0; w < MU_T / 64; ++w) { const unsigned s = sh.wsum[w]; if (w < wave) before += s; tot += s; }
    total = tot;
    return before;
}

// the bin of sh.hist (4096 bins, filled) that holds rank `remaining` from the top -> sh.res[] = {bin, remaining inside the bin, count of the bin}
__device__ __forceinline__ void mu_find(unsigned remaining, MuShared& sh) {
    unsigned h[4], loc = 0u;                                 // thread t owns bins 4095 - (4t + j): descending keys = ascending rank from the top
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = sh.hist[MU_BINS - 1 - (4 * threadIdx.x + j)]; loc += h[j]; }
    unsigned total;
    const unsigned before = mu_scan(loc, sh, total);
    if (before < remaining && before + loc >= remaining) {          // exactly one thread
        unsigned cum = before;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (cum + h[j] >= remaining) { sh.res[0] = MU_BINS - 1 - (4 * threadIdx.x + j); sh.res[1] = remaining - cum; sh.res[2] = h[j]; break; }
            cum += h[j];
        }
    }
    __syncthreads();
}

// One selection of rank k (1 <= k <= M) over the keys  key(i) = elig(i) ? k24(noise[i]) : 0,  elig(i) = (gate bit of the owning thread) — the
// caller has filled sh.top[] — -> bit u of the result: cell u * 1024 + t is MASKED (key > threshold, or threshold-equal and among the first
// `need` of them in index order).  `elig`: bit u = cell u * 1024 + t is eligible.
__device__ __forceinline__ mu_u64 mu_select_mark(const float* __restrict__ noise, mu_u64 elig, int M, int k, MuShared& sh, unsigned& bad) {
    const int t = threadIdx.x;
    // ---- digit 1: histogram of the top digits (exact zeros counted in a register: every ineligible cell sits there) ----
    for (int b = t; b < MU_BINS; b += MU_T) sh.hist[b] = 0u;
    __syncthreads();
    unsigned zeros = 0u;
#pragma unroll 8
    for (int u = 0; u < MU_CPT; ++u) {
        const int i = u * MU_T + t;
        if (i < M) {
            const unsigned tp = sh.top[i];
            if (tp & 0x8000u) ++zeros; else atomicAdd(&sh.hist[tp], 1u);
        }
    }
    zeros = mu_wave_sum(zeros);
    if ((t & 63) == 0 && zeros) atomicAdd(&sh.hist[0], zeros);
    __syncthreads();
    mu_find((unsigned)k, sh);
    const unsigned d1 = sh.res[0], rem = sh.res[1];
    __syncthreads();
    // ---- digit 2: the low digits of the cells whose top digit is d1 (they re-read their noise) ----
    for (int b = t; b < MU_BINS; b += MU_T) sh.hist[b] = 0u;
    __syncthreads();
    zeros = 0u;
#pragma unroll 4
    for (int u = 0; u < MU_CPT; ++u) {
        const int i = u * MU_T + t;
        if (i < M) {
            const unsigned tp = sh.top[i];
            if ((tp & 0xFFFu) == d1) {
                if (tp & 0x8000u) ++zeros; else atomicAdd(&sh.hist[mu_key(noise[i], bad) & 0xFFFu], 1u);
            }
        }
    }
    zeros = mu_wave_sum(zeros);
    if ((t & 63) == 0 && zeros) atomicAdd(&sh.hist[0], zeros);
    __syncthreads();
    mu_find(rem, sh);
    const unsigned d2 = sh.res[0], need = sh.res[1], cnt_eq = sh.res[2];
    __syncthreads();
    const bool ties = need != cnt_eq;                                // uniform
    mu_u64 m = 0ull, eqm = 0ull;
#pragma unroll 4
    for (int u = 0; u < MU_CPT; ++u) {
        const int i = u * MU_T + t;
        if (i < M) {
            const unsigned tp = sh.top[i], tt = tp & 0xFFFu;
            if (tt > d1) m |= 1ull << u;
            else if (tt == d1) {
                const unsigned lo = (tp & 0x8000u) ? 0u : (mu_key(noise[i], bad) & 0xFFFu);
                if (lo > d2) m |= 1ull << u;
                else if (lo == d2) eqm |= 1ull << u;
            }
        }
    }
    (void)elig;
    if (!ties) return m | eqm;
    // rare: a tie straddles rank k — the threshold-equal cells take the `need` slots in index order (cell = u * 1024 + t: row u, then thread t)
    if (t == 0) sh.base_rank = 0u;
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < MU_CPT; ++u) {
        const unsigned eq = (unsigned)(eqm >> u) & 1u;
        unsigned total;
        const unsigned before = mu_scan(eq, sh, total);
        const unsigned base_rank = sh.base_rank;
        if (eq && base_rank + before < need) m |= 1ull << u;
        __syncthreads();
        if (t == 0) sh.base_rank = base_rank + total;
        __syncthreads();
        if (base_rank + total >= need) break;                        // uniform
    }
    return m;
}

template <bool ADAPTIVE>
__global__ __launch_bounds__(MU_T) void mu_mask_kernel(const int* __restrict__ label, const int* __restrict__ list_c, const int* __restrict__ nums,
                                                       const float* __restrict__ noise_a, const float* __restrict__ noise_r, int ada_all, int M,
                                                       int HS, int base, int k_const, float* __restrict__ m_ada, float* __restrict__ m_rnd,
                                                       float* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mu_smem[];
    MuShared& sh = *reinterpret_cast<MuShared*>(mu_smem);
    const int t = threadIdx.x, wave = t >> 6;
    unsigned bad = 0u;
    if (t == 0) sh.bad = 0u;
    mu_u64 ma = 0ull;                                                // selection A: masked cells (incl. the fully masked classes)
    if constexpr (ADAPTIVE) {
        for (int h = t; h < 256; h += MU_T) { sh.counts[h] = 0; sh.cls.d[h] = 0; sh.cls.f[h] = 0; }
        int (*wcounts)[256] = reinterpret_cast<int (*)[256]>(sh.hist);
        static_assert(MU_BINS == (MU_T / 64) * 256, "the per-wave class histogram aliases the digit histogram");
        for (int h = t; h < MU_BINS; h += MU_T) sh.hist[h] = 0u;
        __syncthreads();
        // class histogram (GPTST.py:344-345 bincount)
#pragma unroll 8
        for (int u = 0; u < MU_CPT; ++u) {
            const int i = u * MU_T + t;
            if (i < M) atomicAdd(&wcounts[wave][label[i] & 255], 1);
        }
        __syncthreads();
        for (int h = t; h < HS; h += MU_T) {
            int c = 0;
            for (int w = 0; w < MU_T / 64; ++w) c += wcounts[w][h];
            sh.counts[h] = c;
        }
        __syncthreads();
        if (t == 0) {                                                // class roles, GPTST.py:357-384,393 (as ms_classes)
            const int ada_num = nums[0];
            int num = 0, i = 0;
            while (num < ada_num && i < HS) { num += sh.counts[list_c[i]]; ++i; }
            int dnum = 0;
            if (ada_all && i >= 2) {
                for (int k = 0; k < i - 1; ++k) { sh.cls.d[list_c[k]] = 1; dnum += sh.counts[list_c[k]]; }
                sh.cls.f[list_c[i - 1]] = 1;
            } else {
                for (int k = 0; k < i; ++k) sh.cls.f[list_c[k]] = 1;
            }
            sh.cls.ka = ada_num - dnum;
        }
        __syncthreads();
        const int ka = sh.cls.ka;
        mu_u64 da = 0ull;                                            // cells of the fully masked classes (:397)
#pragma unroll 8
        for (int u = 0; u < MU_CPT; ++u) {                           // selection A keys: noise_a gated by the class (:390)
            const int i = u * MU_T + t;
            if (i < M) {
                const int l = label[i] & 255;
                const unsigned k24 = mu_key(noise_a[i], bad);              // (every value is checked, eligible or not)
                const unsigned kk = sh.cls.f[l] ? k24 : 0u;
                sh.top[i] = (unsigned short)((kk >> 12) | (kk == 0u ? 0x8000u : 0u));
                if (sh.cls.d[l]) da |= 1ull << u;
            }
        }
        __syncthreads();
        // (re-read noise of an INELIGIBLE cell would give it a low digit: the zero flag in top[] stands for the whole key there)
        if (ka > 0) ma = mu_select_mark(noise_a, 0ull, M, ka, sh, bad);       // :386-397
        ma |= da;
        __syncthreads();
    }
    // selection R (adaptive: noise_r gated by m_ada, :399-413) / the random selection (:316-321)
    const float* __restrict__ nz = ADAPTIVE ? noise_r : noise_a;
#pragma unroll 8
    for (int u = 0; u < MU_CPT; ++u) {
        const int i = u * MU_T + t;
        if (i < M) {
            const unsigned k24 = mu_key(nz[i], bad);
            const unsigned kk = (ADAPTIVE && ((ma >> u) & 1ull)) ? 0u : k24;
            sh.top[i] = (unsigned short)((kk >> 12) | (kk == 0u ? 0x8000u : 0u));
        }
    }
    __syncthreads();
    const int kr = ADAPTIVE ? nums[1] : k_const;
    mu_u64 mr = 0ull;
    if (kr > 0) mr = mu_select_mark(nz, 0ull, M, kr, sh, bad);
    if (bad) atomicOr(&sh.bad, 1u);
    __syncthreads();
    const bool poison = sh.bad != 0u;
    const float nan = __int_as_float(0x7fc00000);
#pragma unroll 4
    for (int u = 0; u < MU_CPT; ++u) {
        const int i = u * MU_T + t;
        if (i >= M) break;
        const float va = ((ma >> u) & 1ull) ? 0.f : 1.f;
        const float vr = ((mr >> u) & 1ull) ? 0.f : 1.f;
        if (ADAPTIVE) {
            if (m_ada) m_ada[i] = poison ? nan : va;
            if (m_rnd) m_rnd[i] = poison ? nan : vr;
            const float f = poison ? nan : va * vr;
            for (int j = 0; j < base; ++j) mask[(size_t)i * base + j] = f;
        } else {
            mask[i] = poison ? nan : vr;
        }
    }
}

// ======================================================================================================================
// r05: the whole mask generation as ONE COOPERATIVE launch — <= MC_MAXWG = 128 workgroups of 1024 threads (64 at the bench shape), ONE CELL PER THREAD up to 131 072
// cells (two or four beyond, r06; the keys live in registers), a grid barrier where the multi-launch path has a kernel boundary.  The six launches of the adaptive phase are ~7.5 us each of launch
// + two dependent memory round trips on 64 workgroups (45 us per step, three quarters of the chip idle); a barrier of 64 arrivals on one counter is
// 2-3 us.  Phases: [class histogram] | A digit 1 | A digit 2 | (ties: per-workgroup counts) | mark A, R digit 1 | R digit 2 | (ties) | write.
// Histograms: per-workgroup in LDS (exact zeros — every ineligible cell — counted with a ballot, not with 1000 same-address atomics), non-empty
// bins merged into the global histogram with atomics; behind the barrier every workgroup reads the 4096 bins back (agent-scope loads) and finds
// the threshold bin itself.  Same SET as the multi-launch path and the oracle: ties at rank k go to the lowest cell indices (per-workgroup counts
// of threshold-equal cells + one more barrier, only when a tie straddles the rank).
// All workgroups are resident by construction (<= 128 workgroups of 16 waves, at most half of what the device holds: mc_fits); the barrier wait is bounded like every in-launch wait
// (gptst_wait_ge, 2 s): on expiry the outputs are NaN and the expiry is on record (gptst_handoff_timeouts; the optimiser's guard skips the step).
// ws words (zeroed): [0, 16384) four histograms (selection s, digit d) | 16384 class counts (256) | 16768 barrier | 16769 bad | 16800 tie counts (2 x MC_MAXWG)
// ======================================================================================================================
GPTST_HANDOFF_COUNTER(masksel)
#define MC_T 1024
#define MC_MAXWG 128     // workgroups of the mask role; 1 / 2 / 4 / 8 cells per thread: 131 072 / 262 144 / 524 288 / 1 048 576 cells (r06: the global batch of eight ranks at the PEMS08, METR_LA and NYC_TAXI shapes)
#define MC_MAXM (8 * MC_T * MC_MAXWG)
static inline int mc_cpt(int M) { return M <= MC_T * MC_MAXWG ? 1 : M <= 2 * MC_T * MC_MAXWG ? 2 : M <= 4 * MC_T * MC_MAXWG ? 4 : 8; }      // cells per thread
static inline int mc_nwg(int M) { const int c = mc_cpt(M) * MC_T; return (M + c - 1) / c; }
static_assert(16800 + 2 * MC_MAXWG <= MS_WS_WORDS, "tie counts beyond the mask workspace");
struct McShared {
    unsigned hist[MU_BINS];
    unsigned wsum[MC_T / 64 + 1];
    unsigned res[4];
    int counts[256];
    unsigned ok, base_rank;
    MsClass cls;
};

__device__ __forceinline__ unsigned mc_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// grid barrier number `phase` (1, 2, ..): every thread's global atomics / stores of the phase are out before the workgroup arrives
__device__ __forceinline__ bool mc_barrier(unsigned* bar, unsigned nwg, unsigned phase, McShared& sh) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh.ok = gptst_wait_ge(bar, nwg * phase, &g_handoff_lost_masksel) ? 1u : 0u;
    }
    __syncthreads();
    return sh.ok != 0u;
}

__device__ __forceinline__ unsigned mc_scan(unsigned loc, McShared& sh, unsigned& total) {      // exclusive prefix over the 1024 threads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = loc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned v = __shfl_up(inc, off, 64); if (lane >= off) inc += v; }
    __syncthreads();
    if (lane == 63) sh.wsum[wave] = inc;
    __syncthreads();
    unsigned before = inc - loc, tot = 0u;
#pragma unroll
    for (int w = 0; w < MC_T / 64; ++w) { const unsigned s_ = sh.wsum[w]; if (w < wave) before += s_; tot += s_; }
    total = tot;
    return before;
}

// merge this workgroup's histogram of `dig` (only keys with match) into gh, zeros through a ballot.  CPT cells per thread (r06).
template <int CPT>
__device__ __forceinline__ void mc_hist(const unsigned (&bin)[CPT], const bool (&take)[CPT], unsigned* __restrict__ gh, McShared& sh) {
    for (int b = threadIdx.x; b < MU_BINS; b += MC_T) sh.hist[b] = 0u;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const bool z = take[u] && bin[u] == 0u;
        const unsigned long long zb = __ballot(z);
        if (take[u] && bin[u] != 0u) atomicAdd(&sh.hist[bin[u]], 1u);
        if ((threadIdx.x & 63) == 0 && zb) atomicAdd(&sh.hist[0], (unsigned)__popcll(zb));
    }
    __syncthreads();
    for (int b = threadIdx.x; b < MU_BINS; b += MC_T) { const unsigned v = sh.hist[b]; if (v) atomicAdd(gh + b, v); }
}

// the bin of the (complete) global histogram gh that holds rank `remaining` from the top -> sh.res = {bin, remaining inside it, its count}
__device__ __forceinline__ void mc_find(const unsigned* __restrict__ gh, unsigned remaining, McShared& sh) {
    unsigned h[4], loc = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = mc_ld(gh + MU_BINS - 1 - (4 * threadIdx.x + j)); loc += h[j]; }
    unsigned total;
    const unsigned before = mc_scan(loc, sh, total);
    if (before < remaining && before + loc >= remaining) {
        unsigned cum = before;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (cum + h[j] >= remaining) { sh.res[0] = MU_BINS - 1 - (4 * threadIdx.x + j); sh.res[1] = remaining - cum; sh.res[2] = h[j]; break; }
            cum += h[j];
        }
    }
    __syncthreads();
}

// one selection of rank k over this grid's keys (keys of this thread's CPT consecutive cells; 0 = ineligible): -> m[u] = the cell is masked.
// `ph`: barrier phases used so far.
template <int CPT>
__device__ __forceinline__ void mc_select(const unsigned (&key)[CPT], const bool (&valid)[CPT], int k, unsigned* __restrict__ ws, int s, unsigned nwg,
                                          unsigned& ph, McShared& sh, bool& ok, bool (&m)[CPT]) {
#pragma unroll
    for (int u = 0; u < CPT; ++u) m[u] = false;
    if (k <= 0) return;                                              // uniform over the grid
    unsigned* H0 = ws + (2 * s) * MU_BINS, *H1 = H0 + MU_BINS;
    unsigned* bar = ws + 16768;
    unsigned bin[CPT];
    bool take[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) { bin[u] = key[u] >> 12; take[u] = valid[u]; }
    mc_hist<CPT>(bin, take, H0, sh);
    ok = mc_barrier(bar, nwg, ++ph, sh) && ok;
    mc_find(H0, (unsigned)k, sh);
    const unsigned d1 = sh.res[0], rem = sh.res[1];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CPT; ++u) { take[u] = valid[u] && (key[u] >> 12) == d1; bin[u] = key[u] & 0xFFFu; }
    mc_hist<CPT>(bin, take, H1, sh);
    ok = mc_barrier(bar, nwg, ++ph, sh) && ok;
    mc_find(H1, rem, sh);
    const unsigned thr = (d1 << 12) | sh.res[0], need = sh.res[1], cnt_eq = sh.res[2];
    __syncthreads();
    bool eq[CPT];
    unsigned neq = 0u;
#pragma unroll
    for (int u = 0; u < CPT; ++u) { m[u] = valid[u] && key[u] > thr; eq[u] = valid[u] && key[u] == thr; neq += eq[u] ? 1u : 0u; }
    if (need == cnt_eq) {                                            // uniform: no tie straddles the rank
#pragma unroll
        for (int u = 0; u < CPT; ++u) m[u] = m[u] || eq[u];
        return;
    }
    // a tie straddles rank k: threshold-equal cells take the `need` slots in CELL-INDEX order = (workgroup, thread, cell of the thread) order
    unsigned total;
    const unsigned before = mc_scan(neq, sh, total);
    unsigned* ec = ws + 16800 + MC_MAXWG * s;
    if (threadIdx.x == 0) __hip_atomic_store(ec + blockIdx.x, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = mc_barrier(bar, nwg, ++ph, sh) && ok;
    if (threadIdx.x == 0) {
        unsigned b = 0u;
        for (unsigned w = 0; w < blockIdx.x; ++w) b += mc_ld(ec + w);
        sh.base_rank = b;
    }
    __syncthreads();
    unsigned r = sh.base_rank + before;
#pragma unroll
    for (int u = 0; u < CPT; ++u)
        if (eq[u]) { if (r < need) m[u] = true; ++r; }
    __syncthreads();
}

struct McArgs {
    const int* label; const int* list_c; const int* nums; const float* noise_a; const float* noise_r;
    int ada_all, M, HS, base, k_const;
    float* m_ada; float* m_rnd; float* mask; unsigned* ws;
};

// the mask role: workgroups 0 .. nwg-1 of the launch.  CPT consecutive cells per thread: 1 up to 131 072 cells (the single-rank shapes and two
// data-parallel ranks), 2 / 4 / 8 beyond — the global batch of up to EIGHT ranks (522 240 cells at the bench shape, 817 152 at NYC_TAXI's) still runs on <= 128 workgroups, i.e. with the
// barrier cost and the residency margin of the small grid (r06; the multi-launch select took 119 us there against ~34 for this launch at one rank,
// gpurun_out/r06w8.txt)
template <bool ADAPTIVE, int CPT>
__device__ __forceinline__ void mc_mask_body(const McArgs& g, unsigned nwg) {
    const int* __restrict__ label = g.label; const int* __restrict__ list_c = g.list_c; const int* __restrict__ nums = g.nums;
    const float* __restrict__ noise_a = g.noise_a; const float* __restrict__ noise_r = g.noise_r;
    const int ada_all = g.ada_all, M = g.M, HS = g.HS, base = g.base, k_const = g.k_const;
    float* __restrict__ m_ada = g.m_ada; float* __restrict__ m_rnd = g.m_rnd; float* __restrict__ mask = g.mask;
    unsigned* __restrict__ ws = g.ws;
    __shared__ McShared sh;
    const int t = threadIdx.x, i0 = (blockIdx.x * MC_T + t) * CPT;
    unsigned ph = 0u, bad = 0u;
    bool ok = true;
    unsigned* bar = ws + 16768;
    bool valid[CPT], ma[CPT], mr[CPT];
    unsigned ka24[CPT], kr24[CPT];
    int lab[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        valid[u] = i0 + u < M;
        const int ic = valid[u] ? i0 + u : M - 1;
        const float na = noise_a[ic];
        const float nr = ADAPTIVE ? noise_r[ic] : 0.f;
        lab[u] = ADAPTIVE ? (label[ic] & 255) : 0;
        ka24[u] = mu_key(na, bad); kr24[u] = ADAPTIVE ? mu_key(nr, bad) : 0u;              // (every value is checked, eligible or not)
        ma[u] = false; mr[u] = false;
    }
    if (bad) atomicOr(ws + 16769, 1u);
    if constexpr (ADAPTIVE) {
        // class histogram (GPTST.py:344-345 bincount): LDS per workgroup, one global atomic per class and workgroup
        for (int h = t; h < 256; h += MC_T) sh.counts[h] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < CPT; ++u) if (valid[u]) atomicAdd(&sh.counts[lab[u]], 1);
        __syncthreads();
        int* gc = reinterpret_cast<int*>(ws + 16384);
        for (int h = t; h < HS; h += MC_T) if (sh.counts[h]) atomicAdd(gc + h, sh.counts[h]);
        ok = mc_barrier(bar, nwg, ++ph, sh) && ok;
        for (int h = t; h < 256; h += MC_T) { sh.counts[h] = h < HS ? (int)mc_ld(ws + 16384 + h) : 0; sh.cls.d[h] = 0; sh.cls.f[h] = 0; }
        __syncthreads();
        if (t == 0) {                                                // class roles, GPTST.py:357-384,393 (as ms_classes)
            const int ada_num = nums[0];
            int num = 0, c = 0;
            while (num < ada_num && c < HS) { num += sh.counts[list_c[c]]; ++c; }
            int dnum = 0;
            if (ada_all && c >= 2) {
                for (int k = 0; k < c - 1; ++k) { sh.cls.d[list_c[k]] = 1; dnum += sh.counts[list_c[k]]; }
                sh.cls.f[list_c[c - 1]] = 1;
            } else {
                for (int k = 0; k < c; ++k) sh.cls.f[list_c[k]] = 1;
            }
            sh.cls.ka = ada_num - dnum;
        }
        __syncthreads();
        const int ka = sh.cls.ka;
        bool da[CPT];
        unsigned keyA[CPT], keyR[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            da[u] = valid[u] && sh.cls.d[lab[u]] != 0;
            keyA[u] = (valid[u] && sh.cls.f[lab[u]]) ? ka24[u] : 0u;                      // :390
        }
        __syncthreads();
        mc_select<CPT>(keyA, valid, ka, ws, 0, nwg, ph, sh, ok, ma);                       // :386-397
#pragma unroll
        for (int u = 0; u < CPT; ++u) { ma[u] = ma[u] || da[u]; keyR[u] = ma[u] ? 0u : kr24[u]; }      // :401
        mc_select<CPT>(keyR, valid, nums[1], ws, 1, nwg, ph, sh, ok, mr);                  // :399-413
    } else {
        mc_select<CPT>(ka24, valid, k_const, ws, 0, nwg, ph, sh, ok, mr);                  // :316-321
    }
    // the lattice flag: every workgroup raised it before its first barrier; a launch without any barrier (k = 0 everywhere) reads what is there
    const bool poison = !ok || mc_ld(ws + 16769) != 0u;
    const float nan = __int_as_float(0x7fc00000);
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        if (!valid[u]) continue;
        const int i = i0 + u;
        const float va = ma[u] ? 0.f : 1.f, vr = mr[u] ? 0.f : 1.f;
        if (ADAPTIVE) {
            if (m_ada) m_ada[i] = poison ? nan : va;
            if (m_rnd) m_rnd[i] = poison ? nan : vr;
            const float f = poison ? nan : va * vr;
            for (int j = 0; j < base; ++j) mask[(size_t)i * base + j] = f;
        } else {
            mask[i] = poison ? nan : vr;
        }
    }
}

template <bool ADAPTIVE, int CPT>
__global__ __launch_bounds__(MC_T) void mc_mask_kernel(McArgs g) { mc_mask_body<ADAPTIVE, CPT>(g, gridDim.x); }

// r05: the same launch also runs a table of forward generation jobs (poolgen_dev.h) on workgroups nmask, nmask+1, ..: each 256-thread quarter of a
// workgroup takes one 256-thread job block.  The jobs do not depend on the mask nor the mask on them; the 64 mask workgroups are latency-bound
// (five grid barriers) on 64 CUs, the jobs are write-bound on all of them.  The mask workgroups come FIRST in dispatch order: all resident at once.
template <bool ADAPTIVE, int CPT>
__global__ __launch_bounds__(MC_T) void mc_mask_jobs_kernel(McArgs g, unsigned nmask, PJobs t, int nf, int nvb, int fwd_rows) {
    if (blockIdx.x >= nmask) {
        const int nfw = (nvb + MC_T / 256 - 1) / (MC_T / 256);              // workgroups of the forward jobs; behind them one workgroup per graph-job block
        const int wrel = (int)(blockIdx.x - nmask);
        if (wrel >= nfw) {                                                 // kind 3 (temporal graphs): the whole workgroup runs one block
            __shared__ float scr[PJ_GRAM_SCR];
            const int gb = wrel - nfw;
            int p = nf;
            for (int q = nf + 1; q < t.n; ++q) if (gb >= t.j[q].blk0) p = q;
            pj_gram(t.j[p], gb - t.j[p].blk0, scr, MC_T);
            return;
        }
        const int vb = __builtin_amdgcn_readfirstlane(wrel * (MC_T / 256) + (int)(threadIdx.x >> 8));
        if (vb >= nvb) return;
        int p = 0;
        for (int q = 1; q < nf; ++q) if (vb >= t.j[q].blk0) p = q;
        const PJob& a = t.j[p];
        const int rel = vb - a.blk0;
        pj_fwd_mfma(a, rel % a.nbx, rel / a.nbx, fwd_rows, threadIdx.x & 255);
        return;
    }
    mc_mask_body<ADAPTIVE, CPT>(g, nmask);
}

// The grid barriers need every mask workgroup resident at once: workgroups the device can hold = CUs x occupancy of the kernel (queried once per kernel
// AND device; a partitioned or CU-masked device simply takes the multi-launch path).  The query knows nothing about what else is on the chip — a forked RCCL
// all-reduce under the data-parallel bucket overlap, another process — so the launch is taken only with HALF the slots to spare: at most
// MC_MAXWG = 128 mask workgroups against >= 256 slots on an MI355X.  (The job workgroups of mc_mask_jobs_kernel sit behind the mask workgroups in block
// order and take no part in the barriers.)  A mask workgroup that is not resident after all ends in the bounded wait -> NaN -> the steppers' safe mode.
static bool mc_fits_ptr(const void* kernel, int nwg) {
    // (capacity cached per kernel AND device: the instantiations share one function type, so a per-type static would mix them up)
    struct Ent { const void* k; int dev, cap; };
    static thread_local Ent tab[16];
    static thread_local int ntab = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    int cap = -1;
    for (int i = 0; i < ntab; ++i) if (tab[i].k == kernel && tab[i].dev == dev) cap = tab[i].cap;
    if (cap < 0) {
        int ncu = 0, per = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, MC_T, 0) != hipSuccess) { ncu = 0; per = 0; }
        cap = ncu * per;
        if (ntab < 16) tab[ntab++] = Ent{kernel, dev, cap};
    }
    return 2 * nwg <= cap;
}
// ... of the instantiation that M cells would launch
template <bool ADAPTIVE, bool JOBS>
static bool mc_fits(int M) {
    const int nwg = mc_nwg(M);
#define MC_PTR(CP) (JOBS ? (const void*)mc_mask_jobs_kernel<ADAPTIVE, CP> : (const void*)mc_mask_kernel<ADAPTIVE, CP>)
    switch (mc_cpt(M)) {
        case 1: return mc_fits_ptr(MC_PTR(1), nwg);
        case 2: return mc_fits_ptr(MC_PTR(2), nwg);
        case 4: return mc_fits_ptr(MC_PTR(4), nwg);
        default: return mc_fits_ptr(MC_PTR(8), nwg);
    }
#undef MC_PTR
}

thread_local int g_ms_coop = MS_COOP_DEFAULT;              // (thread-local like the other launch-mode knobs: ranks emulated by threads) gptst_mask_cooperative(0): the multi-launch path instead of the cooperative launch (the steppers' fallback after a lost hand-off; A/B; tests)

static int mu_prepare() {
    static int done = 0;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)mu_mask_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MuShared));
        (void)hipFuncSetAttribute((const void*)mu_mask_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MuShared));
        done = 1;
    }
    return (int)sizeof(MuShared);
}

// Noise on the 2^-24 lattice (see above).  Up to MSS_MAXM cells: ONE launch of one workgroup (top digits in LDS); beyond: the multi-workgroup
// radix select above on the integer keys — TWO digit passes per selection instead of three (adaptive phase: 6 launches instead of 8, random
// phase 3 instead of 4).  gptst_mask_force_multi(2) (tests / A-B): the one-workgroup form up to 65536 cells (measured ~120 us at 65 280 cells
// against ~35 us: one CU's bandwidth and LDS atomics).  ws / ws_zeroed as gptst_mask_random.
extern "C" int gptst_mask_random_u24(const float* noise, int M, int k, float* mask, void* ws, int ws_zeroed, void* stream) {
    if (!noise || !mask || M <= 0 || k < 0 || k > M) return GPTST_EARG;
    if ((M <= MSS_MAXM && g_ms_force_multi != 1) || (g_ms_force_multi == 2 && M <= MU_T * MU_CPT)) {
        hipLaunchKernelGGL((mu_mask_kernel<false>), dim3(1), dim3(MU_T), mu_prepare(), (hipStream_t)stream, (const int*)nullptr, (const int*)nullptr,
                           (const int*)nullptr, noise, (const float*)nullptr, 0, M, 0, 1, k, (float*)nullptr, (float*)nullptr, mask);
        GPTST_CHECK_LAUNCH();
        return GPTST_OK;
    }
    if (g_ms_coop && g_ms_force_multi == 0 && ws && M <= MC_MAXM && mc_fits<false, false>(M)) {   // r05: one cooperative launch
        if (!ws_zeroed) hipLaunchKernelGGL(ms_zero_kernel, dim3(24), dim3(256), 0, (hipStream_t)stream, (unsigned*)ws, MS_WS_WORDS);
        const McArgs g{nullptr, nullptr, nullptr, noise, nullptr, 0, M, 0, 1, k, nullptr, nullptr, mask, (unsigned*)ws};
        const int cpt = mc_cpt(M);
        if (cpt == 1) hipLaunchKernelGGL((mc_mask_kernel<false, 1>), dim3(mc_nwg(M)), dim3(MC_T), 0, (hipStream_t)stream, g);
        else if (cpt == 2) hipLaunchKernelGGL((mc_mask_kernel<false, 2>), dim3(mc_nwg(M)), dim3(MC_T), 0, (hipStream_t)stream, g);
        else if (cpt == 4) hipLaunchKernelGGL((mc_mask_kernel<false, 4>), dim3(mc_nwg(M)), dim3(MC_T), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((mc_mask_kernel<false, 8>), dim3(mc_nwg(M)), dim3(MC_T), 0, (hipStream_t)stream, g);
        GPTST_CHECK_LAUNCH();
        return GPTST_OK;
    }
    return ms_random_impl(noise, M, k, mask, ws, ws_zeroed, stream, 1);
}

// as gptst_mask_adaptive (counts may be NULL: the class histogram is taken from the labels); m_ada / m_rnd may be NULL in the one-workgroup form only.
extern "C" int gptst_mask_adaptive_u24(const int* label, const int* counts, const int* list_c, const int* nums, const float* noise_a,
                                       const float* noise_r, int ada_all, int M, int HS, int base, float* m_ada, float* m_rnd, float* mask,
                                       void* ws, int ws_zeroed, void* stream) {
    if (!label || !list_c || !nums || !noise_a || !noise_r || !mask || HS <= 0 || HS > 256 || M <= 0 || base <= 0) return GPTST_EARG;
    if ((M <= MSS_MAXM && g_ms_force_multi != 1) || (g_ms_force_multi == 2 && M <= MU_T * MU_CPT)) {
        hipLaunchKernelGGL((mu_mask_kernel<true>), dim3(1), dim3(MU_T), mu_prepare(), (hipStream_t)stream, label, list_c, nums, noise_a, noise_r, ada_all,
                           M, HS, base, 0, m_ada, m_rnd, mask);
        GPTST_CHECK_LAUNCH();
        return GPTST_OK;
    }
    if (g_ms_coop && g_ms_force_multi == 0 && ws && M <= MC_MAXM && mc_fits<true, false>(M)) {    // r05: one cooperative launch (the class histogram is taken inside)
        if (!ws_zeroed) hipLaunchKernelGGL(ms_zero_kernel, dim3(24), dim3(256), 0, (hipStream_t)stream, (unsigned*)ws, MS_WS_WORDS);
        const McArgs g{label, list_c, nums, noise_a, noise_r, ada_all, M, HS, base, 0, m_ada, m_rnd, mask, (unsigned*)ws};
        const int cpt = mc_cpt(M);
        if (cpt == 1) hipLaunchKernelGGL((mc_mask_kernel<true, 1>), dim3(mc_nwg(M)), dim3(MC_T), 0, (hipStream_t)stream, g);
        else if (cpt == 2) hipLaunchKernelGGL((mc_mask_kernel<true, 2>), dim3(mc_nwg(M)), dim3(MC_T), 0, (hipStream_t)stream, g);
        else if (cpt == 4) hipLaunchKernelGGL((mc_mask_kernel<true, 4>), dim3(mc_nwg(M)), dim3(MC_T), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((mc_mask_kernel<true, 8>), dim3(mc_nwg(M)), dim3(MC_T), 0, (hipStream_t)stream, g);
        GPTST_CHECK_LAUNCH();
        return GPTST_OK;
    }
    return ms_adaptive_impl(label, counts, list_c, nums, noise_a, noise_r, ada_all, M, HS, base, m_ada, m_rnd, mask, ws, ws_zeroed, stream, 1);
}

// r05: gptst_pool_jobs (njobs generation jobs — kind 0 forward: out_j (R_j, cols_j) = emb_j (R_j, K_j) @ pool_j (K_j, cols_j), kind 3 temporal graph;
// kind NULL: all forward) followed by gptst_mask_random_u24
// (adaptive == 0: noise_a, k) or gptst_mask_adaptive_u24 (adaptive != 0) — as ONE launch when the mask takes the cooperative form and every job the MFMA
// form (cols % 4 == 0), else as those two calls.  The jobs must not read the mask's outputs nor write its inputs (they are independent work that fills
// the CUs the 64 latency-bound mask workgroups leave idle).
GPTST_INTERNAL int gptst_pj_embed_table(PJobs* t, int njobs, const int* kind, const void* const* emb, const void* const* pool, const void* const* out,
                                        const int* R, const int* K, const int* cols, int* nf, int* nvb, int* ngw);
extern "C" int gptst_pool_jobs(int njobs, const int* kind, const void* const* emb, const void* const* x, const void* const* pool,
                               const void* const* out, const int* R, const int* K, const int* cols, const int* nsplit, const int* ldx, void* stream);
extern "C" int gptst_mask_u24_fwd_jobs(int adaptive, const int* label, const int* counts, const int* list_c, const int* nums, const float* noise_a,
                                       const float* noise_r, int ada_all, int M, int HS, int base, int k, float* m_ada, float* m_rnd, float* mask,
                                       void* ws, int ws_zeroed, int njobs, const int* kind, const void* const* emb, const void* const* pool,
                                       const void* const* out, const int* R, const int* K, const int* cols, void* stream) {
    if (njobs < 0 || (njobs && (!emb || !pool || !out || !R || !K || !cols))) return GPTST_EARG;
    if (adaptive ? (!label || !list_c || !nums || !noise_a || !noise_r || !mask || HS <= 0 || HS > 256 || M <= 0 || base <= 0)
                 : (!noise_a || !mask || M <= 0 || k < 0 || k > M)) return GPTST_EARG;
    PJobs t;
    int nf = 0, nvb = 0, ngw = 0;
    const bool coop = g_ms_coop && g_ms_force_multi == 0 && ws && M > MSS_MAXM && M <= MC_MAXM &&
                      (adaptive ? mc_fits<true, true>(M) : mc_fits<false, true>(M));
    if (coop && njobs > 0 && gptst_pj_embed_table(&t, njobs, kind, emb, pool, out, R, K, cols, &nf, &nvb, &ngw) == GPTST_OK) {
        if (!ws_zeroed) hipLaunchKernelGGL(ms_zero_kernel, dim3(24), dim3(256), 0, (hipStream_t)stream, (unsigned*)ws, MS_WS_WORDS);
        const unsigned nmask = (unsigned)mc_nwg(M);
        const int cpt = mc_cpt(M);
        const dim3 grid(nmask + (unsigned)((nvb + MC_T / 256 - 1) / (MC_T / 256)) + (unsigned)ngw);
#define MC_JOBS_LAUNCH(AD, CP) hipLaunchKernelGGL((mc_mask_jobs_kernel<AD, CP>), grid, dim3(MC_T), 0, (hipStream_t)stream, g, nmask, t, nf, nvb, PG_MFMA_ROWS)
        if (adaptive) {
            const McArgs g{label, list_c, nums, noise_a, noise_r, ada_all, M, HS, base, 0, m_ada, m_rnd, mask, (unsigned*)ws};
            if (cpt == 1) MC_JOBS_LAUNCH(true, 1); else if (cpt == 2) MC_JOBS_LAUNCH(true, 2); else if (cpt == 4) MC_JOBS_LAUNCH(true, 4); else MC_JOBS_LAUNCH(true, 8);
        } else {
            const McArgs g{nullptr, nullptr, nullptr, noise_a, nullptr, 0, M, 0, 1, k, nullptr, nullptr, mask, (unsigned*)ws};
            if (cpt == 1) MC_JOBS_LAUNCH(false, 1); else if (cpt == 2) MC_JOBS_LAUNCH(false, 2); else if (cpt == 4) MC_JOBS_LAUNCH(false, 4); else MC_JOBS_LAUNCH(false, 8);
        }
#undef MC_JOBS_LAUNCH
        GPTST_CHECK_LAUNCH();
        return GPTST_OK;
    }
    if (njobs > 0) {
        std::vector<int> kd(njobs, (int)PJ_FWD), one(njobs, 1);
        std::vector<const void*> nul(njobs, nullptr);
        const int rc = gptst_pool_jobs(njobs, kind ? kind : kd.data(), emb, nul.data(), pool, out, R, K, cols, one.data(), nullptr, stream);
        if (rc) return rc;
    }
    return adaptive ? gptst_mask_adaptive_u24(label, counts, list_c, nums, noise_a, noise_r, ada_all, M, HS, base, m_ada, m_rnd, mask, ws, ws_zeroed, stream)
                    : gptst_mask_random_u24(noise_a, M, k, mask, ws, ws_zeroed, stream);
}
