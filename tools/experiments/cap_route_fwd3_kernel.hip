// cap_route_fwd3_kernel: the one-wave-per-tile routing forward (r04), superseded by cap_route_fwd4_kernel (gpt-st_amd/csrc/cap_route3.hip) and
// removed from the product library in r05.  Measured: 33 us against 27 us (profiles/r04_cap_route3_phases.txt; the <= 80 VGPR
// variant spills).  Kept as a record: it compiles inside cap_route3.hip in front of the fourth variant (it uses that file's cr3_* helpers).

template <int MAXW, int OCC>
__global__ __launch_bounds__(64 * MAXW, OCC) void cap_route_fwd3_kernel(const float* __restrict__ X, const float* __restrict__ Wp,
                                                                   const float* __restrict__ bp, const float* __restrict__ dadj,
                                                                   float* __restrict__ c_out, float* __restrict__ s_out, int N, int HS, int R) {
    constexpr int C = 64, P = CR3_P;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NTH = blockDim.x, NW = NTH >> 6;
    float* Wl = smem;                          // [64][64]   Wl[k][col] = Wp[col][k]
    float* scr0 = Wl + C * C;                  // NW x [16][P]: the wave's transposition scratch, then its partial sums [HS + 1][64]
    float* Vs = scr0 + NW * 16 * P;            // [16][P]    v of the running iteration (rows >= HS stay zero)
    const int bt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    float* scr = scr0 + wave * 16 * P;
    const float* Xbt = X + (size_t)bt * N * C;
    const float* l0g = dadj + (size_t)bt * HS * N;
    const int ncol = 16 * wave + j;            // this lane's node in the (cluster, node) layouts
    const bool ncol_ok = ncol < N;

    CR3_TS(0);
    // ---- staging: every global load is issued before the first LDS store ----
    float4 a[4];
    {
        const float* row = Xbt + (size_t)min(16 * wave + j, N - 1) * C + 4 * kk;
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = ld4(row + 16 * q);
    }
    float l0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) l0[r] = (4 * kk + r < HS && ncol_ok) ? l0g[(size_t)(4 * kk + r) * N + ncol] : 0.f;
    const float4 b4 = ld4(bp + 4 * j);
    for (int f0 = 0; f0 < C * C / 4; f0 += 2 * NTH) {
        float4 wv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f = min(f0 + u * NTH + tid, C * C / 4 - 1), k4 = f / C, col = f % C;
            wv[u] = ld4(Wp + (size_t)col * C + 4 * k4);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f = f0 + u * NTH + tid, k4 = f / C, col = f % C;
            if (f < C * C / 4) {
                Wl[(4 * k4 + 0) * C + col] = wv[u].x; Wl[(4 * k4 + 1) * C + col] = wv[u].y;
                Wl[(4 * k4 + 2) * C + col] = wv[u].z; Wl[(4 * k4 + 3) * C + col] = wv[u].w;
            }
        }
    }
    for (int i = tid; i < 16 * P; i += NTH) Vs[i] = 0.f;
    __syncthreads();
    SB();
    CR3_TS(1);

    // ---- P tile = squash(X_tile Wp^T + bp):  yD[r] = row 16 wave + 4kk + r, channels 4j..4j+3 ----
    float4 yD[4], yA[4];
    float4 csum = f4zero();
    {
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 bq[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bq[e] = ld4(Wl + (16 * q + 4 * kk + e) * C + 4 * j);
            const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bq[e].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bq[e].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bq[e].z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bq[e].w, acc[3], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 16 * wave + 4 * kk + r;
            float4 v = f4add(make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), b4);
            if (n >= N) v = f4zero();
            const float sc = squash_scale(group_sum<16>(f4dot(v, v)));
            yD[r] = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
            csum = f4add(csum, yD[r]);
            st4(scr + (4 * kk + r) * P + 4 * j, yD[r]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) yA[q] = ld4(scr + j * P + 16 * q + 4 * kk);      // node j of the tile, channels 16q + 4kk ..
        csum.x += __shfl_xor(csum.x, 16, 64); csum.y += __shfl_xor(csum.y, 16, 64); csum.z += __shfl_xor(csum.z, 16, 64); csum.w += __shfl_xor(csum.w, 16, 64);
        csum.x += __shfl_xor(csum.x, 32, 64); csum.y += __shfl_xor(csum.y, 32, 64); csum.z += __shfl_xor(csum.z, 32, 64); csum.w += __shfl_xor(csum.w, 32, 64);
    }
    CR3_TS(2);
    SB();
    // ---- c0 = softmax_h(dadj), partial of c0 . P and of the column sums (:105; first routing pass) ----
    float c[4], bl[4] = {0.f, 0.f, 0.f, 0.f};
    cr3_softmax(l0, c, kk, HS, ncol_ok);
    cr3_type1(c, yD, scr, j, kk, HS);
    if (kk == 0) st4(scr + HS * 64 + 4 * j, csum);
    CR3_TS(3);
    __syncthreads();
    SB();
    CR3_TS(4);
    // ---- v0 = squash(c0 . P) (:105-106), v = squash(v0 (.) mean-over-classes of the column sums) (:113-117, c = 1/HS) ----
    const int prow = tid >> 4, pc4 = tid & 15;
    const bool poster = tid < 16 * HS;
    float4 v0 = f4zero();
    if (wave * 64 < 16 * HS) {                         // wave-uniform: the waves that hold cluster rows
        float4 S0 = f4zero(), u0 = f4zero();
        if (poster) { S0 = cr3_fold(scr0, NW, prow, pc4); if (R > 0) u0 = cr3_fold(scr0, NW, HS, pc4); }
        const float sc = squash_scale(group_sum<16>(f4dot(S0, S0)));
        v0 = make_float4(S0.x * sc, S0.y * sc, S0.z * sc, S0.w * sc);
        if (R > 0) {
            const float inv = 1.f / (float)HS;
            const float4 t = make_float4(v0.x * (u0.x * inv), v0.y * (u0.y * inv), v0.z * (u0.z * inv), v0.w * (u0.w * inv));
            const float s2 = squash_scale(group_sum<16>(f4dot(t, t)));
            if (poster) st4(Vs + prow * P + 4 * pc4, make_float4(t.x * s2, t.y * s2, t.z * s2, t.w * s2));
        }
    }
    CR3_TS(5);
    for (int it = 1; it < R; ++it) {                   // routing iterations 1 .. R-1 (no grad, :113-118)
        __syncthreads();
        SB();
        CR3_TS(6);
        cr3_type2(bl, yA, Vs, j, kk);                  // b += v . P^T
        cr3_softmax(bl, c, kk, HS, ncol_ok);           // c = softmax_h(b)
        cr3_type1(c, yD, scr, j, kk, HS);
        CR3_TS(7);
        __syncthreads();
        SB();
        CR3_TS(8);
        if (wave * 64 < 16 * HS) {                     // v = squash(v0 (.) c . P)
            float4 S = f4zero();
            if (poster) S = cr3_fold(scr0, NW, prow, pc4);
            const float4 t = make_float4(v0.x * S.x, v0.y * S.y, v0.z * S.z, v0.w * S.w);
            const float s2 = squash_scale(group_sum<16>(f4dot(t, t)));
            if (poster) st4(Vs + prow * P + 4 * pc4, make_float4(t.x * s2, t.y * s2, t.z * s2, t.w * s2));
        }
    }
    CR3_TS(9);
    __syncthreads();
    SB();
    CR3_TS(10);
    if (R > 0) cr3_type2(bl, yA, Vs, j, kk);
    {
        float x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = bl[r] + l0[r];
        cr3_softmax(x, c, kk, HS, ncol_ok);            // c = softmax_h(b + dadj)      :120
    }
    if (ncol_ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * kk + r < HS) c_out[((size_t)bt * HS + 4 * kk + r) * N + ncol] = c[r];
    }
    cr3_type1(c, yD, scr, j, kk, HS);
    CR3_TS(11);
    __syncthreads();
    SB();
    CR3_TS(12);
    if (poster) st4(s_out + ((size_t)bt * HS + prow) * C + 4 * pc4, cr3_fold(scr0, NW, prow, pc4));       // s = c . P    :123
    CR3_TS(13);
}

