/* gptst_hip.h — C ABI of the MI355X (gfx950) GPT-ST pretraining kernels.
 *
 * Drop-in boundary: the reference has no FFI — its hot path sits behind the Python nn.Module
 * GPTST_Model (model/Pretrain_model/GPTST.py:459-493).  gpt-st_amd/model.py mirrors that interface and
 * calls these entry points through ctypes; each entry replaces the chain of ATen ops cited with it.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer into caller-owned memory (a torch tensor) that stays valid until
 *     the stream has executed the call; fp32 unless the name says otherwise; tensors are dense row-major;
 *   - activations are (B, T, N, C) = (B*T*N rows, C); T is 12 (reference GPTST.py:97,208-209);
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); calls only enqueue — no
 *     allocation, no synchronisation, no host reads — so a whole step can be captured in a hipGraph;
 *   - return 0 on success, a negative GPTST_E* code for bad arguments / unsupported shapes, or a positive
 *     hipError_t from the launch.  Never throws, never exits.
 *   - "+=" in a description means the kernel ACCUMULATES into the output (caller zeroes it once per step).
 */
#ifndef GPTST_HIP_H
#define GPTST_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: only what is declared between this push and the pop below is exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define GPTST_ABI_VERSION 15  /* 15 (r06): + gptst_cap_split_units, gptst_cap_cross_route_lin_bwd_split (the last (b,t) units of the routing backward as two node halves: 384 units on 256 CUs -> one whole unit + one half per CU; measured, opt-in), gptst_clip_adam: stats_out[6] = updates skipped since gptst_handoff_reset; + gptst_mask_cooperative_state (the switch is thread-local now); + gptst_guide_head_fwd, gptst_guide_uc_floats (the guide classifier's forward in two launches); 14 (r05, late): + gptst_cap_cross_route_lin_bwd_jobs (gradient-reduction jobs as a role of the routing backward); 13 (r05, late): + gptst_mask_u24_fwd_jobs (forward generation jobs inside the cooperative mask launch); 12 (r05, late): + gptst_mask_cooperative (gptst_mask_*_u24 with 8192 < M <= 65536 cells and a workspace: ONE cooperative launch); 11 (r05): + gptst_cap_cross_route_lin_bwd, gptst_comm_available, gptst_handoff_reset, gptst_set_handoff_guard (gptst_clip_adam: stats_out[5] = expiries on record); - gptst_cap_rec_cross_route_bwd (three-role form, measured slower); 9, 10 (r04, late): + gptst_hypertem_bwd_pair, gptst_cap_rec_cross_route_bwd, gptst_mask_*_u24, gptst_pool_jobs_gram_rows, gptst_handoff_timeouts; gptst_fusion_gate_fwd/bwd */
int gptst_abi_version(void);
/* 1: bit-reproducible steps — the two reductions that end in float atomics by default (embedding gradients of gptst_pool_jobs kind 2,
 * weight gradients of gptst_timefeat_jobs) run as single-owner kernels with a fixed summation order (slower).  Everything else is
 * order-fixed by construction.  Thread-local. */
int gptst_set_deterministic(int on);
/* Number of bounded in-launch hand-off waits that expired since the library was loaded (the roles of gptst_cap_cross_route_bwd /
 * gptst_cap_cross_route_lin_bwd, the lower weight-gradient role of gptst_hypertem_bwd_pair, the grid barriers of the cooperative mask launch
 * (gptst_mask_*_u24): a consumer workgroup waits at most 2 s of wall clock for its producer and poisons its output with NaN on expiry).  0 in a healthy run; lets a NaN loss be told from numerical trouble.  Synchronises. */
int gptst_handoff_timeouts(int* out);
/* r05: while an expiry is on record gptst_clip_adam SKIPS its update (weights and moments untouched; the count goes out in stats_out[5], and — r06 —
 * the number of updates skipped since the last gptst_handoff_reset() in stats_out[6]: a caller that enqueued several steps before it looked knows how many to take back) — the
 * caller re-runs the step without in-launch hand-offs and then clears the record with gptst_handoff_reset() (synchronises).
 * gptst_set_handoff_guard(0) turns the skip off (process-wide). */
int gptst_handoff_reset(void);
/* r05: gptst_mask_random_u24 / gptst_mask_adaptive_u24 with a workspace and 8192 < M <= 1048576 cells run as ONE launch of at most 128 workgroups of
 * 1024 threads (one cell per thread up to 131072 cells, two / four / eight beyond — r06: the global batch of up to eight bench-shape data-parallel ranks; digit histograms in the workspace, grid barriers with the bounded wait above; every workgroup is resident: at most
 * 128 on 256 CUs; the launcher takes this form only while the grid is at most HALF of what the device holds).  0: the multi-launch radix select instead (what a stepper falls back to after a lost hand-off); 1: on; < 0: the build's default
 * (on).  Thread-local (r06), like the other launch-mode switches.  Same masks bit for bit either way.  gptst_mask_cooperative_state(): the calling thread's setting (0 / 1) —
 * a scope that switches the form off restores what it found. */
int gptst_mask_cooperative(int on);
int gptst_mask_cooperative_state(void);
int gptst_set_handoff_guard(int on);

/* ---- embedding-conditioned parameter generation (poolgen.hip) -----------------------------------------
 * out[r,:] = sum_k emb[r,k] * pool[k,:]   r < R, k < K <= 16.  Optional second problem (pool2/out2/cols2) shares emb.
 * Replaces einsum('btd,dio->btio'), einsum('nd,dio->nio'), matmul(emb, bias_pool) (GPTST.py:24-25,29-30,137-138,
 * 160-161), einsum('btd,dhn->bthn') (:104), einsum('bd,dhk->bhk') (:129), einsum('nk,kht->nht') (:156). */
int gptst_poolgen_fwd(const float* emb, const float* pool, float* out, int cols, const float* pool2, float* out2,
                      int cols2, int R, int K, void* stream);
/* dpool[k,:] += sum_rr emb[rr % R, k] * dW[rr,:],  rr < R*nsplit (problem 1) / R (problem 2). */
int gptst_poolgen_bwd_pool(const float* emb, const float* dW, float* dpool, int cols, const float* dW2, float* dpool2,
                           int cols2, int R, int nsplit, int K, void* stream);
/* demb[r,k] += sum_s sum_c dW[s*R + r, c] pool[k,c]  (+ problem 2 without splits). */
int gptst_poolgen_bwd_emb(const float* dW, const float* pool, int cols, const float* dW2, const float* pool2, int cols2,
                          float* demb, int R, int nsplit, int K, void* stream);

/* multi-problem forms (<= 112 problems sharing emb; host arrays of device pointers, read at call time): one launch generates /
 * reduces the generated parameters of several layers (every launch has a ~4-5 us floor on MI355X). */
int gptst_poolgen_fwd_multi(const float* emb, int nprob, const void* pools, const void* outs, const int* cols, int R, int K,
                            void* stream);
int gptst_poolgen_bwd_pool_multi(const float* emb, int nprob, const void* dWs, const void* dpools, const int* cols, const int* nsplit,
                                 int R, int K, void* stream);
int gptst_poolgen_bwd_emb_multi(int nprob, const void* dWs, const void* pools, const int* cols, const int* nsplit, float* demb, int R,
                                int K, void* stream);

/* Job table: njobs independent problems of ANY kind, each with its own embedding and shapes, in ceil(njobs / 112) launches
 * (host arrays, read at call time).  kind 0: out (R,cols) = emb (R,K) . pool (K,cols);  kind 1: out = dpool (K,cols) += sum_rr
 * emb[rr % R,:]^T x[rr,:], rr < R*nsplit — owned by ONE workgroup per element (no atomics: two kind-1 jobs of one call must
 * not share `out`);  kind 2: out = demb (R,K) += (sum_s x[s*R + r,:]) . pool^T (atomic: several jobs may add into one demb).
 * kind 3: out (R,12,12) = A_r^T A_r with A_r = (emb . pool)[r] viewed as (cols/12, 12): hyperTem's per-node temporal graph (what
 * gptst_gram_fwd computes from a materialised A; bit-identical to it) without a launch of its own.
 * ldx[j]: row stride of x in floats (0 = cols) — a job may read a column window of a wider matrix.
 * Unused pointers of a kind may be NULL.  A whole pretraining step needs 3 calls (forward generation, two backward reductions)
 * where per-embedding launches needed ~50. */
int gptst_pool_jobs(int njobs, const int* kind, const void* const* emb, const void* const* x, const void* const* pool,
                    const void* const* out, const int* R, const int* K, const int* cols, const int* nsplit, const int* ldx, void* stream);
/* rows per workgroup of a kind-3 (temporal graph) job for (K, cols); <= 0: the shape does not fit the job kernel — use gptst_gram_fwd on the
 * forward job's output (more than 20 hyperedges at embed_dim 16). */
int gptst_pool_jobs_gram_rows(int K, int cols);

/* ---- C x C contractions on fp32 MFMA (apply.hip) ------------------------------------------------------
 * out[g,m,:] = epi( pro(A)[g,m,:] @ W[g] (+bias[g]) (+resid) ).   mode: 0 TIME (g=(b,t), rows n), 1 NODE (g=n, rows
 * (b,t)), 2 SHARED (one weight).  w_per_group: W is (G,C,C) else (C,C).  transw: W[g] stored [out][in].
 * pro: 0 none, 1 A*lrelu'(A2) (A=dOut, A2=layer output).  epi: 0 plain, 1 lrelu(acc+bias+resid),
 * 2 acc + resid*lrelu'(resid2) (adds the residual branch of a layer's backward), 3 lrelu(acc+bias),
 * 4 (C = 128, r05) (acc + resid)*lrelu'(resid2): the dPre-chain form of 2 (resid already is dPre, resid2 = the layer's INPUT);  5 (C = 128)
 * acc*lrelu'(resid2): the same without a residual branch.
 * colsum (optional): row-split PARTIALS of the bias gradient, colsum[s][g][:] = sum over the rows m of split s of pro(A)[g,m,:],
 *   s < gptst_apply_nsplit(mode, BT, N, C); plain stores in a fixed order (no atomics): the consumer sums the splits.
 * Replaces einsum('btni,btio->btno') / einsum('btni,nio->btno') + bias + residual + LeakyReLU
 * (GPTST.py:26-27,31-32,139-141,162-163), nn.Linear C->C (:102) and their backward w.r.t. the data. */
int gptst_apply_nsplit(int mode, int BT, int N, int C);
int gptst_apply(const float* A, const float* A2, const float* W, int w_per_group, int transw, const float* bias,
                const float* resid, const float* resid2, float* out, float* colsum, int mode, int pro, int epi, int BT,
                int N, int C, void* stream);
/* dW[s*G + g] = sum_{m in split s} A[g,m,:]^T pro(D)[g,m,:];  nsplit = gptst_wgrad_nsplit(mode,BT,N,C) partial sums
 * that the consumer (gptst_poolgen_bwd_*) adds up.  dW must hold nsplit*G*C*C floats. */
int gptst_wgrad_nsplit(int mode, int BT, int N, int C);
int gptst_wgrad(const float* A, const float* D, const float* D2, float* dW, int mode, int pro, int BT, int N, int C,
                void* stream);
/* same, with column sums appended to every split (rows of C*C + C floats): which = 1: [dW | sum_m A[m,:]] (weight AND bias gradient
 * of a shared Linear whose output gradient is A); which = 2: [dW | sum_m pro(D)[m,:]] (weight and bias gradient of a generated
 * layer, e.g. hyperTem's W_bt / b_bt — replaces a separate, atomically accumulated bias-gradient pass).  C = 64 or 128. */
int gptst_wgrad_colsum(const float* A, const float* D, const float* D2, float* dW, int mode, int pro, int which, int BT, int N, int C,
                       void* stream);

/* Backward of one generated-weight layer out = lrelu(S W_g + b_g [+ x]) in ONE pass (C = 64; mode 0 TIME / 1 NODE):
 * dPre = dOut*lrelu'(Y); dS = dPre W_g^T; per row split s < gptst_apply_wgrad_nsplit: dW[s][g] = S^T dPre, colsum[s][g] = column sums
 * of dPre (may be NULL).  Replaces gptst_apply(pro=1, transw=1, colsum) + gptst_wgrad(pro=1), which both read dOut and Y.
 * W (G,C,C) as in the forward ([in][out]).  Backward of einsum('btni,nio->btno') / ('btni,btio->btno') + bias + LeakyReLU
 * (GPTST.py:26-27,31-32,139-141).  dPre chain (see gptst_hypertem_bwd): Y == NULL -> dOut already is dPre; premul != 0 -> dS is
 * multiplied by lrelu'(S) (S = output of the LeakyReLU layer below). */
int gptst_apply_wgrad_nsplit(int mode, int BT, int N);
int gptst_apply_wgrad(const float* dOut, const float* Y, const float* S, const float* W, float* dS, float* dW, float* colsum, int premul,
                      int mode, int BT, int N, int C, void* stream);

/* Backward through the shared Linear at the entry of cap (P = squash(X Wp^T + bp), GPTST.py:102) plus the residual branch of the layer
 * (:139-141) in ONE pass: dX = dY Wp + dOut*lrelu'(out); per row split s < gptst_linear_bwd_nsplit(rows): dWp[s] = dY^T X ([out][in]),
 * dbp[s] = colsum(dY).  Replaces gptst_apply(mode 2, epi 2) + gptst_wgrad_colsum(mode 2).  C = 64.
 * dPre chain: out == NULL -> dOut already is dPre: dX = dY Wp + dPre, and with premul != 0  dX = (dY Wp + dPre) * lrelu'(X). */
int gptst_linear_bwd_nsplit(int rows);
int gptst_linear_bwd(const float* dY, const float* X, const float* Wp, const float* dOut, const float* out, float* dX, float* dWp, float* dbp,
                     int premul, int rows, int C, void* stream);

/* ---- per-node temporal hypergraph of hyperTem (tmix.hip), GPTST.py:156-158 -----------------------------
 * A (N,Hm,T) = node_emb . adj (via poolgen);  G[n] = A[n]^T A[n] (T x T);  ret[b,:,n,:] = G[n] X[b,:,n,:]. */
int gptst_gram_fwd(const float* A, float* G, int N, int Hm, void* stream);
/* gram_bwd: A (L*N,Hm,T) of L layers, dG (L, nsplit, N, T, T) PARTIAL graph gradients (e.g. one per sample from gptst_hypertem_bwd),
 * summed in a fixed order -> dA (L*N,Hm,T). */
int gptst_gram_bwd(const float* A, const float* dG, float* dA, int L, int N, int Hm, int nsplit, void* stream);
/* out = G (*) X  [+ dOut*lrelu'(Y) when dOut != NULL: fuses the residual branch of hyperTem's backward]. */
int gptst_tmix(const float* X, const float* G, const float* dOut, const float* Y, float* out, int B, int T, int N, int C,
               void* stream);
/* dG[n,t,u] = sum_{b,c} dR[b,t,n,c] X[b,u,n,c]   (fp32 MFMA 16x16x4). */
int gptst_tmix_dgraph(const float* dR, const float* X, float* dG, int B, int T, int N, int C, void* stream);
/* both of the above in one pass over dR (the unfused hyperTem backward, C = 128): dX = dOut*lrelu'(Y) + G (*) dR, dG = sum_b dR X^T */
int gptst_tmix_bwd(const float* dR, const float* X, const float* G, const float* dOut, const float* Y, float* dX, float* dG, int B, int T,
                   int N, int C, void* stream);
/* r05, the dPre-chain form of gptst_tmix_bwd (C = 128 path): dPre is the layer's pre-activation gradient (its output is not read);
 * dX = (dPre + G (*) dR), times lrelu'(X) when premul;  dG as above. */
int gptst_tmix_bwd_chain(const float* dR, const float* X, const float* G, const float* dPre, int premul, float* dX, float* dG, int B, int T,
                         int N, int C, void* stream);

/* fused hyperTem forward (hypertem.hip): R = G (*) X, out = LReLU(R W_bt + b_bt + X); one workgroup per (sample, 16 nodes), MFMA 16x16x4
 * with W_bt read from L2.  R_out: R kept for the weight gradient, or NULL (the backward then rebuilds it from X).  C = 64.
 * Replaces GPTST.py:157-158 + :162-163. */
int gptst_hypertem_fwd(const float* X, const float* G, const float* Wbt, const float* bbt, float* R_out, float* out, int B, int T,
                       int N, int C, void* stream);
/* hyperTem forward CHAIN (r04): nstage (1..3) consecutive hyperTem layers in ONE launch on the (sample, 16-node) slab — every one of them is
 * node-local, so a layer's output goes from the accumulators back into the LDS slab (and to HBM once, for the backward) and the next layer
 * starts from it: no load phase, no launch boundary in between.  X: the first layer's input.  Gs, Wbts, bbts, Rs, outs: HOST arrays of nstage
 * device pointers (G (N,T,T), W_bt (B*T,C,C), b_bt (B*T,C), R_out or NULL, out), read at call time.  Bit-identical to the per-layer calls.
 * C = 64 (GPTST_ESHAPE otherwise).  Replaces nstage x (GPTST.py:157-163) per launch.  (r05: the optional node-conditioned first stage of r04 —
 * measured slower than gptst_apply — left the signature.) */
int gptst_hypertem_chain_fwd(const float* X, int nstage, const void* Gs, const void* Wbts, const void* bbts, const void* Rs, const void* outs,
                             int B, int T, int N, int C, void* stream);

/* Encoder input projection + the encoder's first hyperTem layer on the low-rank structure of the input (encin.hip, r04).  For base = 1 the
 * first activation is x0 = m w + bi (m = mask ? flow : fill, a scalar per row), so hyperTem1 needs neither x0 nor a GEMM:
 * out = LReLU(alpha (w W_bt) + beta (bi W_bt) + b_bt + m w + bi) with alpha = sum_u G_n[t,u] m_u, beta = sum_u G_n[t,u]; the backward collapses
 * the same way (rank-2 weight gradient, graph gradient from two dot products per row, input-projection gradient from per-(b,t) vectors).
 * fwd: src rows (B*T*N, lda) with the flow in column 0, mask (B*T*N) 1 = visible or NULL, w = dim_in_flow.weight (C,1), bi = its bias ->
 *      out (B,T,N,C), ab (B*T*N, 2) = (alpha, beta), wv (B*T, 2C) = (w W_bt | bi W_bt), both kept for the backward.
 * bwd: dPre = dOut * lrelu'(out) (chain form) -> dWb (B*T, C*C + C) rows [dW_bt | db_bt], dG (B,N,T,T) per-sample partials,
 *      dinp (B*T, 2C) partials of [d weight | d bias] of dim_in_flow (rows summed by the caller).
 * Replaces GPTST.py:418 + :154-163 of encoder.STHCN_encode.hyperTem1 and their backward.  base = 1, T = 12, C in {64, 128}; else GPTST_ESHAPE. */
int gptst_encin_ht1_fwd(const float* src, int lda, const float* mask, float fill, const float* w, const float* bi, const float* G,
                        const float* Wbt, const float* bbt, float* out, float* ab, float* wv, int B, int T, int N, int C, void* stream);
int gptst_encin_ht1_bwd(const float* dPre, const float* src, int lda, const float* mask, float fill, const float* w, const float* bi,
                        const float* Wbt, const float* ab, const float* wv, float* dWb, float* dG, float* dinp, int B, int T, int N, int C,
                        void* stream);

/* Guide classifier MLP_RL, input projection + node-conditioned layer on the low-rank structure of the input (guidein.hip, r04): for base = 1
 * h1 = LReLU((s w1 + b1) W_n + b_n) = LReLU(s u_n + c_n) with u_n = w1 W_n, c_n = b1 W_n + b_n — an elementwise pass; the backward needs
 * p_n = sum_rows s dPre and q_n = sum_rows dPre only: dW_n = w1^T (x) p_n + b1^T (x) q_n, db_n = q_n, d w1 = sum_n W_n p_n, d b1 = sum_n W_n q_n.
 * fwd: src rows (BT*N, lda) with the flow in column 0 -> h1 (BT*N, C).   bwd: dPre = dOut * lrelu'(h1) -> dWb (N, C*C + C) rows [dW_n | db_n],
 * dinp (N, 2C) partials of [d ln1.weight | d ln1.bias].  Replaces GPTST.py:22-27 and its backward.  base = 1, C in {64, 128}. */
int gptst_guide_in_fwd(const float* src, int lda, const float* w1, const float* b1, const float* Wn, const float* bn, float* h1, int BT, int N,
                       int C, void* stream);
int gptst_guide_in_bwd(const float* dPre, const float* src, int lda, const float* w1, const float* b1, const float* Wn, float* dWb, float* dinp,
                       int BT, int N, int C, void* stream);
/* r06: the classifier's whole forward (GPTST.py:21-34 + the softmax of :332 / :343 and the arg-max of :344-345) in two launches: the node vectors
 * u_n, c_n (N workgroups), then ONE pass grouped by (b,t): h1 = LReLU(s u_n + c_n) built in registers, h2 = LReLU(h1 W_bt + b_bt), logits = h2 W3^T + b3,
 * softmax, first-maximum label.  Replaces gptst_guide_in_fwd + gptst_apply(mode TIME, LReLU) + gptst_rowdot(softmax, label): same arithmetic in the
 * same order (bit-identical h1, h2, prob, label), three activation-sized round trips less.  uc: gptst_guide_uc_floats(N, C) floats of scratch.
 * Wbt (B*T, C, C) [in][out] / bbt (B*T, C): the time-conditioned parameters; W3 (J, C) / b3 (J): MLP_RL.ln3, J <= 16.  C = 64, else GPTST_ESHAPE. */
int gptst_guide_uc_floats(int N, int C);
int gptst_guide_head_fwd(const float* src, int lda, const float* w1, const float* b1, const float* Wn, const float* bn, const float* Wbt,
                         const float* bbt, const float* W3, const float* b3, float* uc, float* h1, float* h2, float* prob, int* label,
                         int BT, int N, int C, int J, void* stream);

/* "dPre chain" convention of the backward kernels (r03): the gradient that travels down the layer chain may be handed over ALREADY multiplied
 * by the LeakyReLU derivative of the activation it belongs to (dPre = dOut * lrelu'(out)).  A consumer is told so by Y == NULL (it then
 * takes its incoming gradient as dPre and never reads its own output), and a producer is told to emit that form by premul != 0: it
 * multiplies its input gradient by lrelu'(X), where X — its input, which it reads anyway — is the output of the LeakyReLU layer below.
 * Y != NULL together with premul != 0 is rejected (GPTST_EARG).
 *
 * fused hyperTem backward w.r.t. data and graph: dX = (dPre + G (*) (dPre W_bt^T)) [* lrelu'(X) if premul], dPre = dOut*lrelu'(Y) or dOut,
 * and PARTIALS of the bias and graph gradients, plain stores in a fixed order (no atomics): dbias (gptst_hypertem_ntiles(N) * B*T, C) — one
 * partial per 16-node tile — and dG (B*N, T, T) — one per sample; the consumers sum them (pool jobs nsplit, gptst_gram_bwd nsplit).
 * (The weight gradient stays in gptst_wgrad.)  C = 64.  Replaces the backward of GPTST.py:157-163. */
int gptst_hypertem_ntiles(int N);
int gptst_hypertem_bwd(const float* dOut, const float* Y, const float* X, const float* G, const float* Wbt, float* dX, float* dbias,
                       float* dG, int premul, int B, int T, int N, int C, void* stream);
/* gptst_hypertem_bwd (without dbias) AND the layer's weight + bias gradient (gptst_wgrad_colsum mode 0, pro 1, which 2 on R, dOut, Y) side
 * by side in ONE launch: the two are independent, and as separate launches their fixed dependency chains add up.
 * dWb: (gptst_wgrad_nsplit(0, B*T, N, 64) * B*T, C*C + C) rows [dW_bt | db_bt].  R == NULL: the weight-gradient workgroups rebuild
 * R_t[n,:] = sum_u G_n[t,u] X_u[n,:] from the sample's X (L2 hits: the slab workgroups of the same sample run on the same XCD) — needs
 * gptst_wgrad_nsplit(...) == 1, else GPTST_ESHAPE.  C = 64. */
int gptst_hypertem_bwd_wgrad(const float* dOut, const float* Y, const float* X, const float* G, const float* Wbt, const float* R, float* dX,
                             float* dG, float* dWb, int premul, int B, int T, int N, int C, void* stream);
/* The backward of TWO consecutive hyperTem layers (L+1 "1" above L "0", nothing in between: GPTST.py:267-268, :271 -> :454) in ONE launch (r04):
 * the slab workgroup turns dPre of layer L+1 into the input gradient times lrelu'(input) — dPre of layer L — and runs layer L on it at once;
 * grid [slab | weight gradient L+1 | weight gradient L], the last role waits (bounded, NaN on expiry) for the sample's slabs to publish dXmid.
 * dOut1 = dPre of layer L+1; X1 / X0 the layers' inputs (B,T,N,C), R1 / R0 their saved temporal mixes; -> dXmid (dPre of layer L), dX0 (times
 * lrelu'(X0)), dG1 / dG0 (B*N,T,T) per-sample partials, dWb1 / dWb0 (nsplit * B*T, C*C + C).  cnt: B 32-bit words, ZERO on entry.  C = 64. */
int gptst_hypertem_bwd_pair(const float* dOut1, const float* X1, const float* G1, const float* Wbt1, const float* R1, const float* X0,
                            const float* G0, const float* Wbt0, const float* R0, float* dXmid, float* dX0, float* dG1, float* dG0,
                            float* dWb1, float* dWb0, void* cnt, int B, int T, int N, int C, void* stream);

/* ---- cap: node x cluster soft assignment + routing + aggregation (cap.hip, cap_cross.hip), GPTST.py:100-141 ----
 * route_fwd, one workgroup per (b,t):  P = squash(X Wp^T + bp) by MFMA into LDS; dadj (BT,HS,N) = teb.adj (from gptst_poolgen_fwd);
 *   v0 = squash(softmax_h(dadj) P);  R x { c = softmax_h(b); v = squash(v0 (.) c P); b += v P^T }  (both contractions on MFMA 16x16x4);
 *   c = softmax_h(b + dadj) -> c_out (BT,HS,N);  s = c P -> (BT,HS,C).   Wp/bp = ln_p.weight ([out][in]) / bias. */
int gptst_cap_route_fwd(const float* X, const float* Wp, const float* bp, const float* dadj, float* c_out, float* s_out, int BT, int N,
                        int C, int HS, int R, void* stream);
/* cross-time hyperedges per sample (GPTST.py:125-134): v = squash(LReLU(dyn^T LReLU(dyn (s + (t+1)/12))) + s);
 * dyn (B,HT,T*HS) = time_eb_spg . t_adj (poolgen); saves Ht (B,HT,C), Rt (B,T*HS,C) for backward. */
int gptst_cap_cross_fwd(const float* s, const float* dyn, const float* tmpl, float* v, float* Ht, float* Rt, int B, int T, int C,
                        int HS, int HT, void* stream);
/* ws: device scratch of gptst_cap_cross_ws_floats() floats — 0 (ws may be NULL) while the T*HS cluster tokens of a sample fit LDS;
 * beyond that (e.g. HS = 40: 480 tokens) both calls switch to plain global-memory kernels and the backward needs B*HT*C floats. */
int gptst_cap_cross_ws_floats(int B, int T, int C, int HS, int HT);
int gptst_cap_cross_bwd(const float* dv, const float* s, const float* Rt, const float* Ht, const float* dyn, const float* tmpl,
                        float* dS, float* ddyn, float* ws, int B, int T, int C, int HS, int HT, void* stream);
/* cluster -> node scatter (GPTST.py:135): rec[bt,n,:] = sum_h c[bt,h,n] v[bt,h,:]; and its backward (dc1, dv). */
int gptst_cap_rec_fwd(const float* c, const float* v, float* rec, int BT, int N, int C, int HS, void* stream);
/* gptst_cap_cross_fwd + gptst_cap_rec_fwd in ONE launch of B*T workgroups: every (b,t) workgroup rebuilds the sample's Ht (repeated by
 * the T workgroups of a sample) and Rt / v of its own HS tokens, then scatters rec[bt] = c[bt]^T v[bt].  Bit-identical to the two launches.
 * C = 64 and the sample's tokens + c[bt] within 80 KB of LDS, else GPTST_ESHAPE.  GPTST.py:125-135. */
int gptst_cap_cross_rec_fwd(const float* s, const float* dyn, const float* tmpl, const float* c, float* v, float* Ht, float* Rt, float* rec,
                            int B, int T, int N, int C, int HS, int HT, void* stream);
int gptst_cap_rec_bwd(const float* drec, const float* c, const float* v, float* dc1, float* dv, int BT, int N, int C, int HS,
                      void* stream);
/* backward through s = c P, c = softmax_h(b + dadj) (routing logits b are detached, GPTST.py:108-109) and the squash:
 * dY (BT*N,C) = grad of X Wp^T + bp;  dlogit (BT,HS,N) = grad of dadj. */
int gptst_cap_route_bwd(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1, const float* dS,
                        float* dY, float* dlogit, int BT, int N, int C, int HS, void* stream);
/* gptst_cap_cross_bwd + gptst_cap_route_bwd in ONE launch of B*T workgroups: the backward of the cross-time block (GPTST.py:125-134) runs as a
 * prologue of every (b,t) workgroup (the part that needs the whole sample is repeated by its T workgroups), dS stays in LDS.
 * dv (B, T*HS, C): gradient of v;  -> dY, dlogit as gptst_cap_route_bwd, ddyn (B, HT, T*HS).  C = 64, else GPTST_ESHAPE.
 * r04 — dS_ws (B*T, HS, C) scratch + flags (4 B 32-bit words, ZERO on entry): both given -> the cross-time backward is a ROLE of the launch: 4 B extra
 * workgroups do it (four per sample, three time steps each) and publish dS (write-through stores + one flag per sample) while the B*T routing workgroups rebuild their
 * capsule tile, which does not depend on dS, and pick dS up behind it (a bounded wait; on expiry dS is poisoned with NaN: a lost hand-off ends the run loudly).
 * Either NULL: every (b,t) workgroup repeats the cross-time backward as a prologue (r03). */
int gptst_cap_cross_route_bwd(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1, const float* dv,
                              const float* s, const float* Rt, const float* Ht, const float* dyn, const float* tmpl, float* dY,
                              float* dlogit, float* ddyn, float* dS_ws, void* flags, int B, int T, int N, int C, int HS, int HT, void* stream);
/* r05 — gptst_cap_cross_route_bwd + gptst_linear_bwd in ONE launch: each (b,t) workgroup goes on from its dY tile (kept in LDS; no dY tensor exists) to
 * the backward of cap's entry Linear (GPTST.py:102) and the layer's residual branch (:139-141):
 *   out == NULL: dX = dY Wp + dPre (the incoming gradient already is dPre), times lrelu'(X) when premul;   out given: dX = dY Wp + dPre*lrelu'(out).
 *   dWp (B*T, C*C) [out][in] and dbp (B*T, C): ONE partial per (b,t) — the caller's reduction sums B*T rows.
 * dlogit, ddyn, dS_ws, flags as gptst_cap_cross_route_bwd.  C = 64, else GPTST_ESHAPE (use the two calls). */
int gptst_cap_cross_route_lin_bwd(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1, const float* dv,
                                  const float* s, const float* Rt, const float* Ht, const float* dyn, const float* tmpl,
                                  const float* dPre, const float* out, int premul, float* dX, float* dWp, float* dbp, float* dlogit,
                                  float* ddyn, float* dS_ws, void* flags, int B, int T, int N, int C, int HS, int HT, void* stream);
/* r05, late — gptst_cap_cross_route_lin_bwd followed by gptst_pool_jobs(njobs, ...) with gradient-reduction jobs (kinds 1 and 2 only), in ONE launch where the
 * role form serves (dS_ws and flags given, not the deterministic mode): the jobs run as further role workgroups behind the routing workgroups, on the CUs and the
 * HBM bandwidth this launch leaves idle (384 routing workgroups on 256 CUs at ~2.3 TB/s).  The jobs' inputs must have been written by EARLIER launches of the
 * stream and nothing in this launch reads their outputs (the steppers: the decoder's weight-gradient reductions under the encoder's routing backward).
 * Otherwise exactly the two calls.  Same results either way (kind-2 outputs are float atomics as in gptst_pool_jobs). */
/* 1 when gptst_cap_cross_route_lin_bwd at this shape takes the role form that also carries jobs (every workgroup resident: B*T + 4 B <= 512, <= 80 KB of LDS) */
int gptst_cap_route_roles_ok(int B, int T, int N, int C, int HS, int HT);
int gptst_cap_cross_route_lin_bwd_jobs(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1, const float* dv,
                                       const float* s, const float* Rt, const float* Ht, const float* dyn, const float* tmpl,
                                       const float* dPre, const float* out, int premul, float* dX, float* dWp, float* dbp, float* dlogit,
                                       float* ddyn, float* dS_ws, void* flags, int B, int T, int N, int C, int HS, int HT,
                                       int njobs, const int* jkind, const void* const* jemb, const void* const* jx, const void* const* jpool,
                                       const void* const* jout, const int* jR, const int* jK, const int* jcols, const int* jnsplit, const int* jldx,
                                       void* stream);
/* r06 NODE HALVES.  The (b,t)-grouped cap kernels run B*T = 384 workgroups on the 256 CUs of an MI355X at two per CU: half the CUs get two units, the rest
 * one, and the launch lasts as long as the former (the reference has no such notion: GPTST.py:100-141 is a chain of whole-tensor einsums).  Every step of
 * the routing backward is local to a 16-node tile, so the LAST nsplit (b,t) can be cut into two node halves with a workgroup each (B*T - nsplit whole
 * units + 2 nsplit halves: at B*T = 1.5 x CUs one whole unit and one half per CU).
 * MEASURED AND NOT ADOPTED (profiles/r06_node_halves.txt: a half lasts as long as a whole unit — the workgroup is a latency chain, not a throughput item):
 * gptst_cap_split_units returns 0 unless a test / benchmark turns the policy on (gptst_tune(25, -1): B*T - CUs where CUs < B*T <= 1.5 CUs, C = 64, N >= 32) — what a caller passes to
 * gptst_cap_cross_route_lin_bwd_split, which is gptst_cap_cross_route_lin_bwd_jobs (njobs may be 0) with dWp (B*T + nsplit, C*C) and dbp (B*T + nsplit, C):
 * a half writes a partial row of its own (rows B*T - nsplit + 2k, + 2k + 1 are the halves of (b,t) = B*T - nsplit + k).  Any 0 <= nsplit <= B*T is valid;
 * the results do not depend on it beyond the summation order of the caller's reduction over the partial rows. */
int gptst_cap_split_units(int BT, int N, int C, int HS);
int gptst_cap_cross_route_lin_bwd_split(const float* X, const float* Wp, const float* bp, const float* c, const float* dc1, const float* dv,
                                        const float* s, const float* Rt, const float* Ht, const float* dyn, const float* tmpl,
                                        const float* dPre, const float* out, int premul, float* dX, float* dWp, float* dbp, float* dlogit,
                                        float* ddyn, float* dS_ws, void* flags, int B, int T, int N, int C, int HS, int HT, int nsplit,
                                        int njobs, const int* jkind, const void* const* jemb, const void* const* jx, const void* const* jpool,
                                        const void* const* jout, const int* jR, const int* jK, const int* jcols, const int* jnsplit, const int* jldx,
                                        void* stream);

/* ---- cap for node counts whose (b,t) capsule matrix does not fit LDS (cap_big.hip; BASELINE config 5: N = 4096, C = 128) ----
 * gptst_cap_fits_lds() == 0 -> the host composes the same algebra from these streaming kernels (ops.py: cap_route_fwd/bwd, cap_rec_*):
 * P = squash(X Wp^T + bp) via gptst_apply + capbig_squash_rows;  cs = softmax_h(use_bl*bl + use_l0*l0)  (all (BT,HS,N));
 * type1: S (BT,HS,C) = cs . P  — the ONLY sum over nodes, i.e. the all-reduce point of a node-sharded run;
 * post: 0 out = S, 1 out = squash(S), 2 out = squash(V0 (.) S);  type2: bl += V . P^T;  rec_fwd / rec_bwd_dc: cluster -> node scatter
 * and the c-gradient of it (dv = type1(c, drec));  route_bwd_rows: backward of gptst_cap_route_bwd given Y = X Wp^T + bp. */
int gptst_cap_fits_lds(int N, int C, int HS);
int gptst_capbig_squash_rows(float* Y, long rows, int C, void* stream);
int gptst_capbig_softmax(const float* bl, const float* l0, float* cs, int BT, int HS, int N, int use_bl, int use_l0, void* stream);
int gptst_capbig_type1(const float* cs, const float* P, float* S, int BT, int HS, int N, int C, void* stream);
int gptst_capbig_post(const float* S, const float* V0, float* out, long rows, int C, int post, void* stream);
int gptst_capbig_type2(const float* V, const float* P, float* bl, int BT, int HS, int N, int C, void* stream);
int gptst_capbig_rec_fwd(const float* c, const float* v, float* rec, int BT, int HS, int N, int C, void* stream);
int gptst_capbig_rec_bwd_dc(const float* drec, const float* v, float* dc1, int BT, int HS, int N, int C, void* stream);
int gptst_capbig_route_bwd_rows(const float* Y, const float* c, const float* dc1, const float* dS, float* dY, float* dlogit, int BT,
                                int HS, int N, int C, void* stream);

/* ---- streaming cap, second generation (capflow.hip): one fused MFMA pass over the capsule matrix per routing iteration ----
 * Same algebra as above for HS <= 16, C in {64,128} (gptst_capflow_supported); sums over nodes leave the kernels as partials per
 * gptst_capflow_nparts(N) node chunks and are folded in index order by gptst_capflow_post (no atomics):
 * squash:  P = squash(Y) row-wise; c0 = softmax_h(dadj); part (BT,nparts,HS+1,C) = [c0^T P ; colsum P]  (colsum = the first routing
 *          iteration, whose coefficients are uniform);
 * route:   L = V ? rows . V^T : 0;  b = L + bl_in;  bl_out <- b;  cs = c_in ? c_in : softmax_h(b + l0);  c_out <- cs;
 *          part (BT,nparts,HS,C) = cs^T . rows   (NULL = operand absent).  Routing iteration: (P, v, b);  last step: (P, v, b, l0 = dadj,
 *          c_out = c);  backward of rec = c^T v: rows = drec, V = v, bl_out = dc1, c_in = c, fold of part = dv;
 * post:    part (BT,nparts,prow,C) -> mode 0 (prow = HS+1): V0 = squash(S0), Vout = squash(V0 (.) colsum/HS) (if given);
 *          mode 1: Vout = squash(V0 (.) S);  mode 2: Vout = S;  mode 3: Vout (BT,prow,C) = plain fold (a node-sharded run all-reduces
 *          it and calls post again with nparts = 1);
 * rec_fwd: rec (BT*N,C) = c^T v;  route_bwd: as gptst_capbig_route_bwd_rows. */
int gptst_capflow_supported(int HS, int C);
int gptst_capflow_nparts(int N);
int gptst_capflow_squash(const float* Y, const float* dadj, float* P, float* part, int BT, int HS, int N, int C, void* stream);
int gptst_capflow_route(const float* rows, const float* V, const float* bl_in, float* bl_out, const float* l0, const float* c_in,
                        float* c_out, float* part, int BT, int HS, int N, int C, void* stream);
int gptst_capflow_post(const float* part, int nparts, int prow, float* V0, float* Vout, int mode, int BT, int HS, int C, void* stream);
int gptst_capflow_rec_fwd(const float* c, const float* v, float* rec, int BT, int HS, int N, int C, void* stream);
int gptst_capflow_route_bwd(const float* Y, const float* c, const float* dc1, const float* dS, float* dY, float* dlogit, int BT,
                            int HS, int N, int C, void* stream);

/* ---- evaluation metrics of Trainer.test (metrics.hip), reference model/BasicTrainer.py:209-248 + lib/metrics.py:11-18,38-43,52-86 ----
 * Accumulates, over one batch, the sums the per-horizon MAE / RMSE / MAPE / CORR need (doubles; caller zeroes them once per evaluation):
 * sums_t (T,5) = [n1, sum|e|, sum e^2, n2, sum|e/y|], sums_tn (T,N,6) = [K, sum p, sum y, sum p^2, sum y^2, sum p y];
 * p = (out*m)*sigma+mu, y = (label*m)*sigma+mu with m = 1 - vis (pretrain mode, :229-235); vis == NULL -> m = 1. */
int gptst_metrics_accum(const float* out, const float* src, int lda, const float* vis, float sigma, float mu, int has_mae_thresh,
                        float mae_thresh, float mape_thresh, int B, int T, int N, int D, double* sums_t, double* sums_tn, void* stream);

/* ---- mask generation, integer work, bit-exact given noise/labels/class order (masksel.hip), GPTST.py:314-323,344-413 ----
 * Masks are fp32 {0,1} arrays, 1 = visible, 0 = masked.  Top-k = radix select on the float bits, 11/11/10-bit digits (ties at
 * rank k -> lowest index).  Up to 2^13 cells the whole generation (class histogram and roles, both selections, mask writes) is ONE
 * launch of one 1024-thread workgroup; beyond, one launch per digit + one to write the mask, 64 workgroups each (the adaptive phase's
 * first mask write also histograms the first digit of the second selection). */
int gptst_mask_ws_bytes(void);   /* device scratch (ws) needed by the two selections below */
/* ws_zeroed != 0: the caller hands ws over all-zero (e.g. as part of a scratch region it clears once per step) — saves the zeroing launch.
 * ws is left dirty. */
int gptst_mask_random(const float* noise, int M, int k, float* mask, void* ws, int ws_zeroed, void* stream);
/* label[i] = argmax_h prob[i,h] (int32), counts[h] (int32, zeroed here) — replaces sort(..)[..., 0] (:344-345). */
int gptst_mask_labels(const float* prob, int rows, int HS, int* label, int* counts, void* stream);
/* adaptive phase: device-side class selection (:356-384) + two selections (:386-407) + product (:410-413).
 * list_c: shuffled class order (int32[HS]); nums: {adaptive_mask_num, random_mask_num} int32[2] on the device;
 * m_ada / m_rnd (M) are the partial masks, mask (M*base) the final one.
 * counts may be NULL: the class histogram is then taken from the labels here (inside the single launch for M <= 2^13, by one more
 * small launch beyond). */
int gptst_mask_adaptive(const int* label, const int* counts, const int* list_c, const int* nums, const float* noise_a,
                        const float* noise_r, int ada_all, int M, int HS, int base, float* m_ada, float* m_rnd, float* mask,
                        void* ws, int ws_zeroed, void* stream);

/* r04 — mask generation for noise on the 2^-24 lattice (every value k * 2^-24, 0 <= k < 2^24: what the step's Philox draws and torch.rand
 * produce): the radix select runs on the integers k with TWO uniform 12-bit digits instead of three float-bit digits — one launch less per
 * selection (adaptive phase 6 launches instead of 8); up to 8192 cells the whole generation is ONE launch of one workgroup.  Same results as
 * gptst_mask_random / gptst_mask_adaptive bit for bit (ties at the threshold in index order).  The lattice is CHECKED on the device: any other
 * noise value poisons the outputs with NaN.  ws / ws_zeroed / counts as in the calls above; m_ada / m_rnd must be given beyond 8192 cells. */
int gptst_mask_random_u24(const float* noise, int M, int k, float* mask, void* ws, int ws_zeroed, void* stream);
int gptst_mask_adaptive_u24(const int* label, const int* counts, const int* list_c, const int* nums, const float* noise_a, const float* noise_r,
                            int ada_all, int M, int HS, int base, float* m_ada, float* m_rnd, float* mask, void* ws, int ws_zeroed, void* stream);
/* r05 — gptst_pool_jobs with njobs generation jobs (kind[j] 0: out_j (R_j, cols_j) = emb_j (R_j, K_j) @ pool_j (K_j, cols_j), or 3: the temporal graphs of
 * gptst_pool_jobs; kind NULL: all 0; no gradient kinds) followed by gptst_mask_random_u24
 * (adaptive == 0: noise_a is the noise, k the number of cells to drop) or gptst_mask_adaptive_u24 (adaptive != 0; k unused) — in ONE launch when the mask
 * takes the cooperative form (gptst_mask_cooperative, 8192 < M <= 1048576, ws given) and every forward job has cols % 4 == 0: the (at most 128) mask workgroups are
 * latency-bound on as many CUs, the jobs are write-bound on all of them, and neither depends on the other (the steppers: the generated parameters of the
 * two STHCNs next to the mask that the guide's output selects).  Otherwise exactly those two calls.  Same results bit for bit either way. */
int gptst_mask_u24_fwd_jobs(int adaptive, const int* label, const int* counts, const int* list_c, const int* nums, const float* noise_a,
                            const float* noise_r, int ada_all, int M, int HS, int base, int k, float* m_ada, float* m_rnd, float* mask, void* ws,
                            int ws_zeroed, int njobs, const int* kind, const void* const* emb, const void* const* pool, const void* const* out,
                            const int* R, const int* K, const int* cols, void* stream);

/* ---- thin projections (small.hip) ------------------------------------------------------------------------
 * lin_in: Y[i,:] = sum_j a'[i,j] W(:,j) + b, a' = mask ? (mask[i,j] ? a[i*lda+j] : fill) : a;  wlayout 0: W[c*J+j], 1: W[j*C+c].
 *   (dim_in_flow on the masked source, GPTST.py:416-418; MLP_RL.ln1 :22; data-gradients of rowdot)
 * rowdot: Z[i,j] = X[i,:].W[j,:] + b[j] (+ softmax over j)   (dim_flow_out :455; MLP_RL.ln3 + softmax :33,:332)
 * rowouter: out(j,c) += sum_i a'[i,j] X[i,c] (olayout 0: out[c*J+j], 1: out[j*C+c]); csum[c] += sum_i X[i,c];
 *   asum[j] += sum_i a'[i,j]    (weight / bias gradients of both) */
int gptst_lin_in(const float* a, int lda, const float* mask, float fill, const float* W, int wlayout, const float* b, float* Y,
                 int rows, int J, int C, void* stream);
/* label (optional, int32 per row): argmax_j Z[i,j], first maximum — the cluster label of the adaptive mask (GPTST.py:344-345) */
int gptst_rowdot(const float* X, const float* W, const float* b, float* Z, int rows, int J, int C, int do_softmax, int* label,
                 void* stream);
/* r04 — the gate of the downstream front end (reference model/Model.py:5-18 Fusion, :106 lin_test) in one launch:
 *   x_t = flow . Wt^T + bt  (flow: `base` values per row of src at stride lda);  z = sigmoid(F Ws^T + bs + x_t Wh^T + bh);
 *   out = (z F + (1 - z) x_t) Wo^T + bo.      F, out, z: (rows, C); weights as nn.Linear stores them ([out][in]); z may be NULL (inference).
 * gptst_fusion_gate_bwd: the backward's data path — dpre = dHm (F - x_t) z (1 - z) with dHm = dOut Wo (gradient of both gate pre-activations),
 * dxd = dHm (1 - z), and Hm, xt re-formed as operands of the weight gradients (gptst_wgrad_colsum / gptst_apply do those: gpt-st_amd/fusion.py).
 * C = 64 and base <= 4, else GPTST_ESHAPE. */
int gptst_fusion_gate_fwd(const float* F, const float* src, int lda, int base, const float* Ws, const float* bs, const float* Wh,
                          const float* bh, const float* Wo, const float* bo, const float* Wt, const float* bt, float* out, float* z,
                          int rows, int C, void* stream);
int gptst_fusion_gate_bwd(const float* dOut, const float* F, const float* z, const float* src, int lda, int base, const float* Wo,
                          const float* Wt, const float* bt, float* dpre, float* dxd, float* Hm, float* xt, int rows, int C, void* stream);
int gptst_rowouter_ws_floats(int J, int C);   /* scratch (ws) size of gptst_rowouter */
/* first stage of gptst_rowouter alone: part (gptst_rowouter_nparts(rows), J*C + C + J) = row-chunk partials [sum a'^T X (j,c) | column
 * sums of X | sums of a'] for the caller to fold (one kind-1 pool job next to the other reductions of a step). */
int gptst_rowouter_nparts(int rows);
int gptst_rowouter_part(const float* a, int lda, const float* mask, float fill, const float* X, float* part, int want_asum, int rows,
                        int J, int C, void* stream);
int gptst_rowouter(const float* a, int lda, const float* mask, float fill, const float* X, float* out, int olayout, float* csum,
                   float* asum, float* ws, int rows, int J, int C, void* stream);

/* ---- time-index embeddings (timefeat.hip), GPTST.py:187-219 ------------------------------------------------
 * rows x K day/week features (K=1: time_feature, rows=B*T; K=12: time_feature_spg, rows=B) -> (rows, E).
 * The ten tensors are ln_day.{weight,bias}, ln_week.{..}, ln1.{..}, ln2.{..}, ln.{..}; bwd ACCUMULATES into g*. */
int gptst_timefeat_fwd(const float* wd, const float* bd, const float* ww, const float* bw, const float* w1, const float* b1,
                       const float* w2, const float* b2, const float* w3, const float* b3, const float* tidx, float* out, int rows,
                       int K, int E, void* stream);
int gptst_timefeat_bwd(const float* wd, const float* bd, const float* ww, const float* bw, const float* w1, const float* b1,
                       const float* w2, const float* b2, const float* w3, const float* b3, float* gwd, float* gbd, float* gww,
                       float* gbw, float* gw1, float* gb1, float* gw2, float* gb2, float* gw3, float* gb3, const float* tidx,
                       const float* dout, int rows, int K, int E, void* stream);

/* ---- fused thin heads (tails.hip): one pass over the C-wide activation where the loss meets the network --------------------
 * tail_mae: out = dec W^T + b (GPTST.py:455), masked-MAE statistics (Run.py:92-100, lib/metrics.py:11-18), gradient of the SUM loss
 *   w.r.t. dec -> d_dec (the 1/#kept of the mean is applied by gptst_clip_adam, hyper[9] = 1), per-workgroup partials
 *   part[blk][J*C + J] of (gW, gb) — gptst_tail_parts(rows) rows, summed by the caller (one kind-1 pool job) — and per-workgroup loss
 *   statistics sws[blk][4] = (sum |y-p|, kept count, -, -).
 * tail_kl: the backward through softmax + MLP_RL.ln3 (BasicTrainer.py:85, GPTST.py:33): d_h2, the partials of (gW3, gb3), and the
 *   KL sum of the workgroup in sws[blk][2].  prob (rows,HS) row-major, c (BT,HS,N).
 * stats_fold: stats[0..2] += the column sums of sws (rows, 4) in a fixed order (no float atomics anywhere on this path); rows the
 *   tail kernels did not write must be zero.  C = 64 and J / HS <= 16, else GPTST_ESHAPE (use the unfused ops).
 * premul != 0 (dPre chain, see gptst_hypertem_bwd): d_dec / d_h2 are multiplied by lrelu'(dec) / lrelu'(h2). */
int gptst_tail_parts(int rows);
int gptst_tail_mae(const float* dec, const float* W, const float* b, const float* src, int lda, const float* mask, float sigma, float mu,
                   float thresh, float* out, float* d_dec, float* part, float* sws, int premul, int rows, int J, int C, void* stream);
int gptst_tail_kl(const float* h2, const float* W3, const float* prob, const float* c, float w, float* d_h2, float* part, float* sws,
                  int premul, int rows, int N, int HS, int C, void* stream);
int gptst_stats_fold(const float* sws, int rows, float* stats, void* stream);

/* njobs (<= 16) time-feature instances in ONE launch (a step has seven).  params: njobs x 10 device pointers in the module order
 * above; grads likewise (bwd != 0 only, +=); io[q]: output (fwd) or output gradient (bwd) of job q; all jobs read one tidx. */
int gptst_timefeat_jobs(int njobs, int bwd, const void* const* params, const void* const* grads, const float* tidx,
                        const void* const* io, const int* rows, const int* K, const int* E, void* stream);

/* start of a step in one launch (stepbegin.hip): zero z0[0..n0) (the [flat gradient | statistics] buffer: optimizer.zero_grad,
 * BasicTrainer.py:79) and z1[0..n1) (the step's zero-initialised scratch; may be NULL), and gather tidx (BT,2) = src[:, 0, base:base+2]
 * from src (BT, N, lda) (GPTST.py:256-257; tidx may be NULL); noise[0..n_noise) (may be NULL) <- the step's mask noise, uniform [0,1)
 * (torch.rand_like of GPTST.py:316,367,391) from Philox4x32-10 keyed by the DEVICE words rng[0] = seed, rng[1] = step counter. */
int gptst_step_begin(float* z0, long n0, float* z1, long n1, const float* src, float* tidx, int BT, int N, int lda, int base,
                     float* noise, long n_noise, const int* rng, void* stream);

/* ---- loss + optimiser (loss_adam.hip) ---------------------------------------------------------------------
 * stats: device float[8] zeroed once per step: [0] sum|y-p| [1] kept count [2] KL sum [3] extra sum g^2 terms (in; node-sharded
 * runs) [4] total sum g^2 (out) [6] [7] fold tickets of tails.hip.
 * mae: Run.py:92-100 + lib/metrics.py:11-18 + lib/normalization.py:23-27;  kl: Run.py:132 + BasicTrainer.py:85 (w = 0.1),
 * also emits the gradient w.r.t. the MLP_RL logits;  clip_adam: BasicTrainer.py:95-97 + Run.py:134 over flat buffers
 * (hyper layout documented in loss_adam.hip). */
int gptst_mae_fwd(const float* out, const float* src, int lda, const float* mask, float sigma, float mu, float thresh, int rows,
                  int J, float* stats, void* stream);
int gptst_mae_bwd(const float* out, const float* src, int lda, const float* mask, float sigma, float mu, float thresh, int rows,
                  int J, const float* stats, int normalize, float* dOut, void* stream);
int gptst_kl(const float* prob, const float* c, int rows, int N, int HS, float w, float* dlogit, float* stats, void* stream);
int gptst_clip_adam_ws_floats(void);   /* scratch floats (ws) of gptst_clip_adam: one gradient-norm partial per workgroup, folded in order */
int gptst_clip_adam(float* p, const float* g, float* m, float* v, long nA, long nB, const float* hyper, float* stats, float* ws,
                    float* stats_out, const float* sws, int sws_rows, void* stream);   /* stats_out (optional float[8]): copy of the final statistics block */

/* ---- layer-level entry points (layers.hip): ONE call per reference layer, forward and backward ---------------------------------------
 * SURVEY.md 8(b) "minimum set".  Host-side compositions of the kernel entry points above on the caller's stream; `saved` (forward ->
 * backward) and `scratch` (backward only) are caller-owned device regions of gptst_layer_bytes() bytes; no allocation / sync / host read.
 * Parameter and embedding gradients are ACCUMULATED (+=: zero them once), data gradients are written.  C = 64 and a (b,t) capsule matrix
 * that fits LDS (gptst_cap_fits_lds), else GPTST_ESHAPE; GPTST_EWS (-3) when a region is too small.
 * kind: 0 hyperTem (GPTST.py:154-163; uses d, Hm), 1 cap (:100-141; d, ds, HS, HT), 2 MLP_RL (:21-34; d, HS, base). */
int gptst_layer_bytes(int kind, int B, int T, int N, int C, int d, int Hm, int ds, int HS, int HT, int base, long* saved_bytes,
                      long* scratch_bytes);
/* hyperTem.forward(x, node_embeddings, time_eb) with parameters adj (d,Hm,T), weights_pool (d,C,C), bias_pool (d,C); time_eb (B*T,d). */
int gptst_hypertem_layer_fwd(const float* x, const float* node_emb, const float* time_eb, const float* adj, const float* wpool,
                             const float* bpool, float* out, void* saved, long saved_bytes, int B, int T, int N, int C, int d, int Hm,
                             void* stream);
int gptst_hypertem_layer_bwd(const float* dout, const float* x, const float* out, const float* node_emb, const float* time_eb,
                             const float* adj, const float* wpool, const float* bpool, const void* saved, long saved_bytes, float* dx,
                             float* d_node_emb, float* d_time_eb, float* d_adj, float* d_wpool, float* d_bpool, void* scratch,
                             long scratch_bytes, int B, int T, int N, int C, int d, int Hm, void* stream);
/* cap.forward(x, node_embeddings_spg, time_eb_spg (B,ds), teb (B*T,ds)) with ln_p, adj (ds,HS,N), t_adj (ds,HT,T*HS), weights_spa (d,C,C),
 * bias_spa (d,C), mask_template (T) -> out, c_out (B*T,HS,N) = the returned soft assignment, dyn_out (B,HT,T*HS); R = num_route. */
int gptst_cap_layer_fwd(const float* x, const float* node_emb_spg, const float* time_eb_spg, const float* teb, const float* ln_p_w,
                        const float* ln_p_b, const float* adj, const float* t_adj, const float* wspa, const float* bspa,
                        const float* mask_template, float* out, float* c_out, float* dyn_out, void* saved, long saved_bytes, int B, int T,
                        int N, int C, int d, int ds, int HS, int HT, int R, void* stream);
int gptst_cap_layer_bwd(const float* dout, const float* x, const float* out, const float* c, const float* dyn, const float* node_emb_spg,
                        const float* time_eb_spg, const float* teb, const float* ln_p_w, const float* ln_p_b, const float* adj,
                        const float* t_adj, const float* wspa, const float* bspa, const float* mask_template, const void* saved,
                        long saved_bytes, float* dx, float* d_node_emb_spg, float* d_time_eb_spg, float* d_teb, float* d_ln_p_w,
                        float* d_ln_p_b, float* d_adj, float* d_t_adj, float* d_wspa, float* d_bspa, void* scratch, long scratch_bytes, int B,
                        int T, int N, int C, int d, int ds, int HS, int HT, void* stream);
/* MLP_RL.forward(eb = a[:, :base] (rows = B*T*N, row stride lda), time_eb (B*T,d), node_eb (N,d)) -> logits (rows, HS); the input is data: no dx. */
int gptst_mlprl_layer_fwd(const float* a, int lda, const float* time_eb, const float* node_emb, const float* ln1_w, const float* ln1_b,
                          const float* wpool_spa, const float* bpool_spa, const float* wpool_tem, const float* bpool_tem, const float* ln3_w,
                          const float* ln3_b, float* logits, void* saved, long saved_bytes, int B, int T, int N, int C, int d, int base,
                          int HS, void* stream);
int gptst_mlprl_layer_bwd(const float* dlogits, const float* a, int lda, const float* time_eb, const float* node_emb, const float* ln1_w,
                          const float* wpool_spa, const float* bpool_spa, const float* wpool_tem, const float* bpool_tem, const float* ln3_w,
                          const void* saved, long saved_bytes, float* d_time_eb, float* d_node_emb, float* d_ln1_w, float* d_ln1_b,
                          float* d_wpool_spa, float* d_bpool_spa, float* d_wpool_tem, float* d_bpool_tem, float* d_ln3_w, float* d_ln3_b,
                          void* scratch, long scratch_bytes, int B, int T, int N, int C, int d, int base, int HS, void* stream);

/* ---- communication (comm.hip): RCCL over xGMI with an explicit stream — a collective can sit inside a captured hipGraph -------------
 * The reference has no distributed code; these carry the data-parallel gradient exchange (one all-reduce of [flat gradient | statistics])
 * and the node-sharded cluster aggregations.  RCCL is bound at run time (dlopen): -4 (GPTST_ECOMM) when it is not available or no
 * communicator is given, 1000 + ncclResult_t on an RCCL error.  Communicators are HANDLES (r04; a process-global one before): a process may
 * hold several — the row and the column of a data-parallel x node-shard mesh (SURVEY 8(e) "Combination") — each created by gptst_comm_init
 * on the ranks that form it.  unique_id: 128 bytes (ncclUniqueId) created on one rank of the communicator and distributed by the caller. */
int gptst_comm_available(void);                                /* RCCL can be bound here (no bootstrap, no socket): the probe before a communicator is formed */
int gptst_comm_unique_id(void* out128);
int gptst_comm_init(int rank, int world, const void* unique_id, void** comm_out);    /* *comm_out: the handle the calls below take */
int gptst_allreduce_f32(void* comm, float* buf, long n, void* stream);      /* in-place sum over the ranks */
/* recv[r*n .. (r+1)*n) <- rank r's send[0..n), 32-bit words (the cluster labels of a data-parallel global batch); send may be recv + rank*n */
int gptst_allgather_i32(void* comm, const int* send, int* recv, long n, void* stream);
int gptst_comm_count(void* comm, int* out);                    /* ncclCommCount of the communicator */
int gptst_comm_destroy(void* comm);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GPTST_HIP_H */
