// Layer-level entry points of the C ABI (SURVEY.md 8b "minimum set"): ONE call per reference layer — hyperTem (GPTST.py:154-163), cap
// (:100-141), MLP_RL (:21-34) — forward and backward, for consumers that do not want to compose the kernel-level entry points themselves
// (the Python step, engine.py, batches parameter generation and gradient reductions across layers and keeps using the kernels directly).
// Host code only: every function is a fixed sequence of the kernel entry points of this library on the caller's stream, working in
// caller-owned memory: `saved` (forward -> backward) and `scratch` (backward only) are carved by gptst_layer_bytes().  No allocation, no
// synchronisation, no host reads.  Gradients of parameters / embeddings are ACCUMULATED (+=), data gradients are written.
// C = 64 and shapes whose (b,t) capsule matrix fits LDS (gptst_cap_fits_lds): anything else returns GPTST_ESHAPE (compose the streaming
// kernels, as ops.py does).
#include "common.h"
#include "gptst_hip.h"

namespace {
struct Carve {
    char* p; long left;
    float* take(long nfloats) {
        const long b = ((nfloats * 4 + 255) / 256) * 256;
        if (b > left) { left = -1; return nullptr; }
        float* r = (float*)p; p += b; left -= b;
        return r;
    }
};
inline long pad(long nfloats) { return ((nfloats * 4 + 255) / 256) * 256; }

__global__ void layer_fill_kernel(float* p, int n, float v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

struct Jobs {                       // a small table for gptst_pool_jobs
    int n = 0;
    int kind[24], R[24], K[24], cols[24], ns[24], ldx[24];
    const void *emb[24], *x[24], *pool[24], *out[24];
    void add(int k, const float* e, const float* xx, const float* pl, float* o, int r, int kk, int c, int s = 1, int l = 0) {
        kind[n] = k; emb[n] = e; x[n] = xx; pool[n] = pl; out[n] = o; R[n] = r; K[n] = kk; cols[n] = c; ns[n] = s; ldx[n] = l; ++n;
    }
    int launch(void* st) { return n ? gptst_pool_jobs(n, kind, emb, x, pool, out, R, K, cols, ns, ldx, st) : GPTST_OK; }
};
enum { FWD = 0, BWD_POOL = 1, BWD_EMB = 2 };
#define TRY(e) do { const int rc__ = (e); if (rc__) return rc__; } while (0)
}  // namespace

// kind 0 hyperTem (d, Hm used), 1 cap (d, ds, HS, HT used), 2 MLP_RL (d, HS, base used).  -> bytes of the `saved` and `scratch` regions.
extern "C" int gptst_layer_bytes(int kind, int B, int T, int N, int C, int d, int Hm, int ds, int HS, int HT, int base, long* saved_bytes,
                                 long* scratch_bytes) {
    if (!saved_bytes || !scratch_bytes || B <= 0 || T != 12 || N <= 0) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    const long BT = (long)B * T, rows = BT * N, CC = (long)C * C;
    long sv = 0, sc = 0;
    if (kind == 0) {
        sv = pad((long)N * Hm * T) + pad((long)N * T * T) + pad(BT * CC) + pad(BT * C) + pad(rows * C);
        const long ns = gptst_wgrad_nsplit(0, (int)BT, N, C);
        sc = pad(ns * BT * (CC + C)) + pad((long)B * N * T * T) + pad((long)N * Hm * T);
    } else if (kind == 1) {
        sv = pad(BT * HS * N) + pad(BT * HS * C) * 3 + pad((long)B * HT * C) + pad(rows * C) + pad((long)N * CC) + pad((long)N * C);
        const long ns = gptst_apply_wgrad_nsplit(1, (int)BT, N), ns2 = gptst_linear_bwd_nsplit((int)rows);
        sc = pad(rows * C) * 2 + pad(ns * N * CC) + pad(ns * N * C) + pad(BT * HS * N) * 2 + pad(BT * HS * C) * 2 + pad((long)B * HT * T * HS) +
             pad(ns2 * CC) + pad(ns2 * C) + pad(ns2) + pad((long)gptst_cap_cross_ws_floats(B, T, C, HS, HT));
    } else if (kind == 2) {
        sv = pad(rows * C) * 3 + pad((long)N * CC) + pad((long)N * C) + pad(BT * CC) + pad(BT * C);
        const long nsn = gptst_apply_wgrad_nsplit(1, (int)BT, N), nst = gptst_apply_wgrad_nsplit(0, (int)BT, N);
        sc = pad(rows * C) * 3 + pad(nsn * N * (CC + C)) + pad(nst * BT * (CC + C)) + pad((long)gptst_rowouter_ws_floats(HS > base ? HS : base, C));
    } else return GPTST_EARG;
    *saved_bytes = sv; *scratch_bytes = sc;
    return GPTST_OK;
}

// ---- hyperTem (GPTST.py:154-163) -------------------------------------------------------------------------------------------------------
// x (B,T,N,C), node_emb (N,d), time_eb (B*T,d), adj (d,Hm,T), wpool (d,C,C), bpool (d,C) -> out (B,T,N,C).
extern "C" int gptst_hypertem_layer_fwd(const float* x, const float* node_emb, const float* time_eb, const float* adj, const float* wpool,
                                        const float* bpool, float* out, void* saved, long saved_bytes, int B, int T, int N, int C, int d,
                                        int Hm, void* stream) {
    if (!x || !node_emb || !time_eb || !adj || !wpool || !bpool || !out || !saved) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    const long BT = (long)B * T, CC = (long)C * C;
    Carve w{(char*)saved, saved_bytes};
    float *A = w.take((long)N * Hm * T), *G = w.take((long)N * T * T), *Wbt = w.take(BT * CC), *bbt = w.take(BT * C), *R = w.take(BT * N * C);
    if (w.left < 0) return GPTST_EWS;
    Jobs j;
    j.add(FWD, node_emb, nullptr, adj, A, N, d, Hm * T);                       // :156
    j.add(FWD, time_eb, nullptr, wpool, Wbt, (int)BT, d, (int)CC);             // :160
    j.add(FWD, time_eb, nullptr, bpool, bbt, (int)BT, d, C);                   // :161
    TRY(j.launch(stream));
    TRY(gptst_gram_fwd(A, G, N, Hm, stream));                                  // :157-158 as one T x T matrix per node
    return gptst_hypertem_fwd(x, G, Wbt, bbt, R, out, B, T, N, C, stream);     // :157-163
}

// dout: gradient of `out`.  -> dx (written); d_node_emb, d_time_eb, d_adj, d_wpool, d_bpool (+=).
extern "C" int gptst_hypertem_layer_bwd(const float* dout, const float* x, const float* out, const float* node_emb, const float* time_eb,
                                        const float* adj, const float* wpool, const float* bpool, const void* saved, long saved_bytes,
                                        float* dx, float* d_node_emb, float* d_time_eb, float* d_adj, float* d_wpool, float* d_bpool,
                                        void* scratch, long scratch_bytes, int B, int T, int N, int C, int d, int Hm, void* stream) {
    if (!dout || !x || !out || !node_emb || !time_eb || !adj || !wpool || !bpool || !saved || !dx || !d_node_emb || !d_time_eb || !d_adj ||
        !d_wpool || !d_bpool || !scratch) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    const long BT = (long)B * T, CC = (long)C * C;
    Carve w{(char*)saved, saved_bytes};
    float *A = w.take((long)N * Hm * T), *G = w.take((long)N * T * T), *Wbt = w.take(BT * CC);
    w.take(BT * C);
    float* R = w.take(BT * N * C);
    const int ns = gptst_wgrad_nsplit(0, (int)BT, N, C);
    Carve s{(char*)scratch, scratch_bytes};
    float *dWb = s.take((long)ns * BT * (CC + C)), *dG = s.take((long)B * N * T * T), *dA = s.take((long)N * Hm * T);
    if (w.left < 0 || s.left < 0) return GPTST_EWS;
    TRY(gptst_hypertem_bwd_wgrad(dout, out, x, G, Wbt, R, dx, dG, dWb, 0, B, T, N, C, stream));
    TRY(gptst_gram_bwd(A, dG, dA, 1, N, Hm, B, stream));
    const int ld = (int)(CC + C);
    Jobs j;
    j.add(BWD_POOL, time_eb, dWb, nullptr, d_wpool, (int)BT, d, (int)CC, ns, ld);
    j.add(BWD_POOL, time_eb, dWb + CC, nullptr, d_bpool, (int)BT, d, C, ns, ld);
    j.add(BWD_EMB, nullptr, dWb, wpool, d_time_eb, (int)BT, d, (int)CC, ns, ld);
    j.add(BWD_EMB, nullptr, dWb + CC, bpool, d_time_eb, (int)BT, d, C, ns, ld);
    j.add(BWD_POOL, node_emb, dA, nullptr, d_adj, N, d, Hm * T);
    j.add(BWD_EMB, nullptr, dA, adj, d_node_emb, N, d, Hm * T);
    return j.launch(stream);
}

// ---- cap (GPTST.py:100-141) -------------------------------------------------------------------------------------------------------------
// x (B,T,N,C), node_emb_spg (N,d), time_eb_spg (B,ds), teb (B*T,ds), ln_p (C,C)+(C), adj (ds,HS,N), t_adj (ds,HT,T*HS), wspa (d,C,C),
// bspa (d,C), mask_template (T)  ->  out (B,T,N,C), c_out (B*T,HS,N) soft assignment, dyn_out (B,HT,T*HS) cross-time hyperedges.
extern "C" int gptst_cap_layer_fwd(const float* x, const float* node_emb_spg, const float* time_eb_spg, const float* teb, const float* ln_p_w,
                                   const float* ln_p_b, const float* adj, const float* t_adj, const float* wspa, const float* bspa,
                                   const float* mask_template, float* out, float* c_out, float* dyn_out, void* saved, long saved_bytes,
                                   int B, int T, int N, int C, int d, int ds, int HS, int HT, int R, void* stream) {
    if (!x || !node_emb_spg || !time_eb_spg || !teb || !ln_p_w || !ln_p_b || !adj || !t_adj || !wspa || !bspa || !mask_template || !out ||
        !c_out || !dyn_out || !saved) return GPTST_EARG;
    if (C != 64 || !gptst_cap_fits_lds(N, C, HS)) return GPTST_ESHAPE;
    const long BT = (long)B * T, CC = (long)C * C;
    Carve w{(char*)saved, saved_bytes};
    float *dadj = w.take(BT * HS * N), *s = w.take(BT * HS * C), *v = w.take(BT * HS * C), *Rt = w.take(BT * HS * C), *Ht = w.take((long)B * HT * C),
          *rec = w.take(BT * N * C), *Wn = w.take((long)N * CC), *bn = w.take((long)N * C);
    if (w.left < 0) return GPTST_EWS;
    Jobs j;
    j.add(FWD, teb, nullptr, adj, dadj, (int)BT, ds, HS * N);                  // :104
    j.add(FWD, time_eb_spg, nullptr, t_adj, dyn_out, B, ds, HT * T * HS);      // :129
    j.add(FWD, node_emb_spg, nullptr, wspa, Wn, N, d, (int)CC);                // :137
    j.add(FWD, node_emb_spg, nullptr, bspa, bn, N, d, C);                      // :138
    TRY(j.launch(stream));
    TRY(gptst_cap_route_fwd(x, ln_p_w, ln_p_b, dadj, c_out, s, (int)BT, N, C, HS, R, stream));                          // :102-123
    int rc = gptst_cap_cross_rec_fwd(s, dyn_out, mask_template, c_out, v, Ht, Rt, rec, B, T, N, C, HS, HT, stream);      // :125-135
    if (rc == GPTST_ESHAPE) {
        TRY(gptst_cap_cross_fwd(s, dyn_out, mask_template, v, Ht, Rt, B, T, C, HS, HT, stream));
        rc = gptst_cap_rec_fwd(c_out, v, rec, (int)BT, N, C, HS, stream);
    }
    TRY(rc);
    return gptst_apply(rec, nullptr, Wn, 1, 0, bn, x, nullptr, out, nullptr, 1, 0, 1, (int)BT, N, C, stream);           // :139-141
}

// dout: gradient of `out` (c_out / dyn_out are detached in the reference).  -> dx; d_node_emb_spg, d_time_eb_spg, d_teb, d_ln_p_w,
// d_ln_p_b, d_adj, d_t_adj, d_wspa, d_bspa (+=).
extern "C" int gptst_cap_layer_bwd(const float* dout, const float* x, const float* out, const float* c, const float* dyn,
                                   const float* node_emb_spg, const float* time_eb_spg, const float* teb, const float* ln_p_w,
                                   const float* ln_p_b, const float* adj, const float* t_adj, const float* wspa, const float* bspa,
                                   const float* mask_template, const void* saved, long saved_bytes, float* dx, float* d_node_emb_spg,
                                   float* d_time_eb_spg, float* d_teb, float* d_ln_p_w, float* d_ln_p_b, float* d_adj, float* d_t_adj,
                                   float* d_wspa, float* d_bspa, void* scratch, long scratch_bytes, int B, int T, int N, int C, int d, int ds,
                                   int HS, int HT, void* stream) {
    if (!dout || !x || !out || !c || !dyn || !node_emb_spg || !time_eb_spg || !teb || !ln_p_w || !ln_p_b || !adj || !t_adj || !wspa || !bspa ||
        !mask_template || !saved || !dx || !d_node_emb_spg || !d_time_eb_spg || !d_teb || !d_ln_p_w || !d_ln_p_b || !d_adj || !d_t_adj ||
        !d_wspa || !d_bspa || !scratch) return GPTST_EARG;
    if (C != 64 || !gptst_cap_fits_lds(N, C, HS)) return GPTST_ESHAPE;
    const long BT = (long)B * T, rows = BT * N, CC = (long)C * C;
    Carve w{(char*)saved, saved_bytes};
    w.take(BT * HS * N);
    float *s = w.take(BT * HS * C), *v = w.take(BT * HS * C), *Rt = w.take(BT * HS * C), *Ht = w.take((long)B * HT * C), *rec = w.take(rows * C),
          *Wn = w.take((long)N * CC);
    const int ns = gptst_apply_wgrad_nsplit(1, (int)BT, N), ns2 = gptst_linear_bwd_nsplit((int)rows);
    Carve q{(char*)scratch, scratch_bytes};
    float *drec = q.take(rows * C), *dY = q.take(rows * C), *dWn = q.take((long)ns * N * CC), *dbn = q.take((long)ns * N * C),
          *dc1 = q.take(BT * HS * N), *dlogit = q.take(BT * HS * N), *dv = q.take(BT * HS * C), *dS = q.take(BT * HS * C),
          *ddyn = q.take((long)B * HT * T * HS), *dWp = q.take((long)ns2 * CC), *dbp = q.take((long)ns2 * C), *ones = q.take(ns2),
          *cws = q.take(gptst_cap_cross_ws_floats(B, T, C, HS, HT));
    if (w.left < 0 || q.left < 0) return GPTST_EWS;
    TRY(gptst_apply_wgrad(dout, out, rec, Wn, drec, dWn, dbn, 0, 1, (int)BT, N, C, stream));                             // :139-141 backward
    TRY(gptst_cap_rec_bwd(drec, c, v, dc1, dv, (int)BT, N, C, HS, stream));                                              // :135
    int rc = gptst_cap_cross_route_bwd(x, ln_p_w, ln_p_b, c, dc1, dv, s, Rt, Ht, dyn, mask_template, dY, dlogit, ddyn, nullptr, nullptr, B, T, N, C, HS, HT, stream);   // (no zeroed flag words in the caller-owned scratch: the replicated-prologue form)
    if (rc == GPTST_ESHAPE) {
        TRY(gptst_cap_cross_bwd(dv, s, Rt, Ht, dyn, mask_template, dS, ddyn, cws, B, T, C, HS, HT, stream));             // :125-134
        rc = gptst_cap_route_bwd(x, ln_p_w, ln_p_b, c, dc1, dS, dY, dlogit, (int)BT, N, C, HS, stream);                  // :102-123
    }
    TRY(rc);
    TRY(gptst_linear_bwd(dY, x, ln_p_w, dout, out, dx, dWp, dbp, 0, (int)rows, C, stream));                              // :102 + residual branch
    hipLaunchKernelGGL(layer_fill_kernel, dim3((ns2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, ones, ns2, 1.0f);
    GPTST_CHECK_LAUNCH();
    Jobs j;
    j.add(BWD_POOL, node_emb_spg, dWn, nullptr, d_wspa, N, d, (int)CC, ns);
    j.add(BWD_POOL, node_emb_spg, dbn, nullptr, d_bspa, N, d, C, ns);
    j.add(BWD_EMB, nullptr, dWn, wspa, d_node_emb_spg, N, d, (int)CC, ns);
    j.add(BWD_EMB, nullptr, dbn, bspa, d_node_emb_spg, N, d, C, ns);
    j.add(BWD_POOL, time_eb_spg, ddyn, nullptr, d_t_adj, B, ds, HT * T * HS);
    j.add(BWD_EMB, nullptr, ddyn, t_adj, d_time_eb_spg, B, ds, HT * T * HS);
    j.add(BWD_POOL, teb, dlogit, nullptr, d_adj, (int)BT, ds, HS * N);
    j.add(BWD_EMB, nullptr, dlogit, adj, d_teb, (int)BT, ds, HS * N);
    j.add(BWD_POOL, ones, dWp, nullptr, d_ln_p_w, ns2, 1, (int)CC);
    j.add(BWD_POOL, ones, dbp, nullptr, d_ln_p_b, ns2, 1, C);
    return j.launch(stream);
}

// ---- MLP_RL (GPTST.py:21-34): the guide classifier of the adaptive mask -----------------------------------------------------------------
// a (rows = B*T*N, lda): the first `base` columns of a row are the raw flow; time_eb (B*T,d), node_emb (N,d); ln1 (C,base)+(C), spatial / temporal
// pools (d,C,C)+(d,C), ln3 (HS,C)+(HS)  ->  logits (rows, HS)  (the reference applies softmax outside, :332 / :343).
extern "C" int gptst_mlprl_layer_fwd(const float* a, int lda, const float* time_eb, const float* node_emb, const float* ln1_w,
                                     const float* ln1_b, const float* wpool_spa, const float* bpool_spa, const float* wpool_tem,
                                     const float* bpool_tem, const float* ln3_w, const float* ln3_b, float* logits, void* saved,
                                     long saved_bytes, int B, int T, int N, int C, int d, int base, int HS, void* stream) {
    if (!a || !time_eb || !node_emb || !ln1_w || !ln1_b || !wpool_spa || !bpool_spa || !wpool_tem || !bpool_tem || !ln3_w || !ln3_b || !logits ||
        !saved) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    const long BT = (long)B * T, rows = BT * N, CC = (long)C * C;
    Carve w{(char*)saved, saved_bytes};
    float *h0 = w.take(rows * C), *h1 = w.take(rows * C), *h2 = w.take(rows * C), *Wn = w.take((long)N * CC), *bn = w.take((long)N * C),
          *Wt = w.take(BT * CC), *bt = w.take(BT * C);
    if (w.left < 0) return GPTST_EWS;
    Jobs j;
    j.add(FWD, node_emb, nullptr, wpool_spa, Wn, N, d, (int)CC);               // :24
    j.add(FWD, node_emb, nullptr, bpool_spa, bn, N, d, C);                     // :25
    j.add(FWD, time_eb, nullptr, wpool_tem, Wt, (int)BT, d, (int)CC);          // :29
    j.add(FWD, time_eb, nullptr, bpool_tem, bt, (int)BT, d, C);                // :30
    TRY(j.launch(stream));
    TRY(gptst_lin_in(a, lda, nullptr, 0.f, ln1_w, 0, ln1_b, h0, (int)rows, base, C, stream));                           // :22
    TRY(gptst_apply(h0, nullptr, Wn, 1, 0, bn, nullptr, nullptr, h1, nullptr, 1, 0, 3, (int)BT, N, C, stream));         // :26-27
    TRY(gptst_apply(h1, nullptr, Wt, 1, 0, bt, nullptr, nullptr, h2, nullptr, 0, 0, 3, (int)BT, N, C, stream));         // :31-32
    return gptst_rowdot(h2, ln3_w, ln3_b, logits, (int)rows, HS, C, 0, nullptr, stream);                                // :33
}

// dlogits (rows, HS) -> d_time_eb, d_node_emb, d_ln1_w/b, d_wpool_spa / d_bpool_spa, d_wpool_tem / d_bpool_tem, d_ln3_w/b (+=).  (The input is data: no dx.)
extern "C" int gptst_mlprl_layer_bwd(const float* dlogits, const float* a, int lda, const float* time_eb, const float* node_emb,
                                     const float* ln1_w, const float* wpool_spa, const float* bpool_spa, const float* wpool_tem,
                                     const float* bpool_tem, const float* ln3_w, const void* saved, long saved_bytes, float* d_time_eb,
                                     float* d_node_emb, float* d_ln1_w, float* d_ln1_b, float* d_wpool_spa, float* d_bpool_spa,
                                     float* d_wpool_tem, float* d_bpool_tem, float* d_ln3_w, float* d_ln3_b, void* scratch,
                                     long scratch_bytes, int B, int T, int N, int C, int d, int base, int HS, void* stream) {
    if (!dlogits || !a || !time_eb || !node_emb || !ln1_w || !wpool_spa || !bpool_spa || !wpool_tem || !bpool_tem || !ln3_w || !saved ||
        !d_time_eb || !d_node_emb || !d_ln1_w || !d_ln1_b || !d_wpool_spa || !d_bpool_spa || !d_wpool_tem || !d_bpool_tem || !d_ln3_w ||
        !d_ln3_b || !scratch) return GPTST_EARG;
    if (C != 64) return GPTST_ESHAPE;
    const long BT = (long)B * T, rows = BT * N, CC = (long)C * C;
    Carve w{(char*)saved, saved_bytes};
    float *h0 = w.take(rows * C), *h1 = w.take(rows * C), *h2 = w.take(rows * C), *Wn = w.take((long)N * CC);
    w.take((long)N * C);
    float* Wt = w.take(BT * CC);
    const int nsn = gptst_apply_wgrad_nsplit(1, (int)BT, N), nst = gptst_apply_wgrad_nsplit(0, (int)BT, N);
    Carve q{(char*)scratch, scratch_bytes};
    float *dh2 = q.take(rows * C), *dh1 = q.take(rows * C), *dh0 = q.take(rows * C), *dWn = q.take((long)nsn * N * (CC + C)),
          *dWt = q.take((long)nst * BT * (CC + C)), *ro = q.take(gptst_rowouter_ws_floats(HS > base ? HS : base, C));
    if (w.left < 0 || q.left < 0) return GPTST_EWS;
    float *dbn = dWn + (long)nsn * N * CC, *dbt = dWt + (long)nst * BT * CC;
    TRY(gptst_lin_in(dlogits, HS, nullptr, 0.f, ln3_w, 1, nullptr, dh2, (int)rows, HS, C, stream));                     // d h2 = dlogits W3
    TRY(gptst_rowouter(dlogits, HS, nullptr, 0.f, h2, d_ln3_w, 1, nullptr, d_ln3_b, ro, (int)rows, HS, C, stream));     // ln3 weight / bias
    TRY(gptst_apply_wgrad(dh2, h2, h1, Wt, dh1, dWt, dbt, 0, 0, (int)BT, N, C, stream));                                // :31-32 backward
    TRY(gptst_apply_wgrad(dh1, h1, h0, Wn, dh0, dWn, dbn, 0, 1, (int)BT, N, C, stream));                                // :26-27 backward
    TRY(gptst_rowouter(a, lda, nullptr, 0.f, dh0, d_ln1_w, 0, d_ln1_b, nullptr, ro, (int)rows, base, C, stream));       // ln1 weight / bias
    Jobs j;
    j.add(BWD_POOL, time_eb, dWt, nullptr, d_wpool_tem, (int)BT, d, (int)CC, nst);
    j.add(BWD_POOL, time_eb, dbt, nullptr, d_bpool_tem, (int)BT, d, C, nst);
    j.add(BWD_EMB, nullptr, dWt, wpool_tem, d_time_eb, (int)BT, d, (int)CC, nst);
    j.add(BWD_EMB, nullptr, dbt, bpool_tem, d_time_eb, (int)BT, d, C, nst);
    j.add(BWD_POOL, node_emb, dWn, nullptr, d_wpool_spa, N, d, (int)CC, nsn);
    j.add(BWD_POOL, node_emb, dbn, nullptr, d_bpool_spa, N, d, C, nsn);
    j.add(BWD_EMB, nullptr, dWn, wpool_spa, d_node_emb, N, d, (int)CC, nsn);
    j.add(BWD_EMB, nullptr, dbn, bpool_spa, d_node_emb, N, d, C, nsn);
    return j.launch(stream);
}
