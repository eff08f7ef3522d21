// The cap's node-conditioned layer as the FIRST stage of hypertem_chain_fwd_kernel (template parameter NODE, r04; GPTST_CHAIN_NODE=1): per node one
// 16-row MFMA tile whose rows are the 12 time steps of the sample.  Measured slower than the node-grouped apply64 it replaces (18 vs 12.6 us: it
// cannot share W_n over the 384 (b,t) rows of a node) and removed from the product library in r05.  This is the block that sat
// in `if constexpr (NODE) { ... }` at the top of the kernel in gpt-st_amd/csrc/hypertem.hip (struct HtChain carried rec / Wn / bn / xres / out0).

    if constexpr (NODE) {
        // ---- node layer: x[t][n][:] = LReLU(rec[b,t,n,:] W_n + b_n + xres[b,t,n,:]) -> slab + out0 ----
        HTC_LOAD_G(ch.st[0].G);
        const int tr = min(j, HT_T - 1);                              // A-operand row = time step (rows >= 12: don't-care)
        float4 a[4], xr[4], bn4;
#define HTC_LOAD_NODE(nn) do {                                                                                     \
            const int n_ = (nn);                                                                                   \
            const float* W_ = ch.Wn + (size_t)n_ * C * C;                                                          \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) a[q] = ld4(ch.rec + (((size_t)b * HT_T + tr) * N + n_) * C + 16 * q + 4 * kk); \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                          \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) bv[q][e] = ld4(W_ + (size_t)(16 * q + 4 * kk + e) * C + 4 * j); \
            bn4 = ld4(ch.bn + (size_t)n_ * C + 4 * j);                                                             \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                          \
                xr[r] = ld4(ch.xres + (((size_t)b * HT_T + min(4 * kk + r, HT_T - 1)) * N + n_) * C + 4 * j);      \
        } while (0)
        if (n0 + wave < N) HTC_LOAD_NODE(n0 + wave);
        for (int nl = wave; nl < NT; nl += 4) {
            const int n = n0 + nl;
            if (n < N) {                                              // wave-uniform
                SB();
                f32x4 acc[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].x, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].y, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].z, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[q][e].w, acc[3], 0, 0, 0);
                    }
                }
                SB();
                float4 y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y[r] = f4add(f4add(make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), bn4), xr[r]);
                    y[r].x = lrelu(y[r].x); y[r].y = lrelu(y[r].y); y[r].z = lrelu(y[r].z); y[r].w = lrelu(y[r].w);
                }
                if (nl + 4 < NT && n + 4 < N) HTC_LOAD_NODE(n + 4);   // next node's operands: requested before this node's stores
                SB();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 4 * kk + r;
                    if (t < HT_T) {
                        st4(ch.out0 + (((size_t)b * HT_T + t) * N + n) * C + 4 * j, y[r]);
                        st4(Xs + (t * NT + nl) * P + 4 * j, y[r]);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * kk + r < HT_T) st4(Xs + ((4 * kk + r) * NT + nl) * P + 4 * j, f4zero());
            }
        }
#undef HTC_LOAD_NODE
        HTC_STORE_G();
