// Per-phase wall-clock stamps (100 MHz) of cap_route_fwd3_kernel at the bench shape, workgroups 5 and 300, first and last wave; and the
// launch time back to back.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCR3_STAMPS -I gpt-st_amd/csrc -I include -o scratch/cr3_phases tools/experiments/cap_route3_phases.hip
#include "../../gpt-st_amd/csrc/cap_route3.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, T = 12, N = 170, C = 64, HS = 10, R = 2, BT = B * T;
    std::vector<float> hx((size_t)BT * N * C), hw(C * C), hb(C), hd((size_t)BT * HS * N);
    srand(1);
    auto rn = [] { return (rand() / (float)RAND_MAX - 0.5f); };
    for (auto& v : hx) v = rn(); for (auto& v : hw) v = 0.3f * rn(); for (auto& v : hb) v = 0.3f * rn(); for (auto& v : hd) v = 2.f * rn();
    float *X, *Wp, *bp, *dadj, *c, *s;
    CK(hipMalloc(&X, hx.size() * 4)); CK(hipMalloc(&Wp, hw.size() * 4)); CK(hipMalloc(&bp, hb.size() * 4)); CK(hipMalloc(&dadj, hd.size() * 4));
    CK(hipMalloc(&c, hd.size() * 4)); CK(hipMalloc(&s, (size_t)BT * HS * C * 4));
    CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Wp, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bp, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dadj, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {   // cap_route_fwd4_kernel: schedule of the workgroups (start / end per workgroup, CU it ran on)
        g_cap_route_occ6 = 0;
        for (int lag = 0; lag <= 0; ++lag) {      // (the start-lag variant of r04 is gone from the kernel: profiles/r04_cap_route3_phases.txt holds its runs)
            for (int i = 0; i < 5; ++i) gptst_cap_route_fwd3(X, Wp, bp, dadj, c, s, BT, N, C, HS, R, nullptr);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 50; ++i) gptst_cap_route_fwd3(X, Wp, bp, dadj, c, s, BT, N, C, HS, R, nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            static long long wg[1024][4];
            CK(hipMemcpyFromSymbol(wg, HIP_SYMBOL(g_cr4_wg), sizeof(wg)));
            long long t0 = wg[0][0];
            for (int i = 0; i < BT; ++i) t0 = wg[i][0] < t0 ? wg[i][0] : t0;
            printf("fwd4 B = %d lag %d: %.2f us per launch; workgroup: start end (us from the first start) hw_id xcc\n", B, lag, ms * 1e3f / 50);
            for (int i = 0; i < BT; i += (i < 16 || (i >= 248 && i < 272) || i >= BT - 8) ? 1 : 16)
                printf("   wg %3d: %6.2f %6.2f  cu-id %03llx se %llu xcc %llu\n", i, (wg[i][0] - t0) * 0.01, (wg[i][1] - t0) * 0.01, (wg[i][2] >> 8) & 0xf, (wg[i][2] >> 13) & 0x7, wg[i][3] & 0xf);
            double se = 0; long long last = 0;
            for (int i = 0; i < BT; ++i) { se += (wg[i][1] - wg[i][0]) * 0.01; last = wg[i][1] > last ? wg[i][1] : last; }
            printf("   mean workgroup duration %.2f us, last end %.2f us\n", se / BT, (last - t0) * 0.01);
            static long long t4[6][32];
            CK(hipMemcpyFromSymbol(t4, HIP_SYMBOL(g_cr4_ts), sizeof(t4)));
            const char* n4[] = {"prologue loads landed (X rows, Wp fragments, logits)", "capsule tiles: GEMM + squash -> LDS", "c0 pass (softmax, c0.P) + partial", "barrier",
                                "fold 0: v0, v", "barrier", "pass r1: b += v.P^T, softmax, c.P", "barrier", "fold: v", "barrier", "last pass (+dadj) -> c_out, c.P", "barrier", "fold: s_out"};
            for (int k = 0; k < 6; ++k) {
                if (BT <= 261 && k >= 4) break;
                printf("  workgroup %d (%s), wave %d: start %.2f total %.2f us\n", k < 2 ? 5 : k < 4 ? 200 : 261, k < 2 ? "shares its CU with 261" : k < 4 ? "alone on its CU" : "second resident",
                       (k & 1) ? 7 : 0, (t4[k][0] - t0) * 0.01, (t4[k][13] - t4[k][0]) * 0.01);
                for (int i = 1; i <= 13; ++i) printf("     %-52s %6.2f us\n", n4[i - 1], (t4[k][i] - t4[k][i - 1]) * 0.01);
            }
        }
    }
    for (int occ6 = 1; occ6 < 3; ++occ6) {
        g_cap_route_occ6 = occ6;
        for (int i = 0; i < 5; ++i) if (gptst_cap_route_fwd3(X, Wp, bp, dadj, c, s, BT, N, C, HS, R, nullptr)) { printf("launch failed\n"); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 50; ++i) gptst_cap_route_fwd3(X, Wp, bp, dadj, c, s, BT, N, C, HS, R, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long ts[4][32];
        CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_cr3_ts), sizeof(ts)));
        printf("B = %d occ6 = %d: %.2f us per launch (50 back to back)\n", B, occ6, ms * 1e3f / 50);
        const char* nm[] = {"stage Wl (+ X tile, l0 requested)", "P tile GEMM + squash + layout change", "c0 softmax + c0.P + partial", "barrier", "fold 0: v0, v",
                            "barrier", "r1: b += v.P^T, softmax, c.P", "barrier", "fold: v", "barrier", "b += v.P^T, softmax(+dadj), c_out, c.P", "barrier", "fold: s_out"};
        for (int k = 0; k < 4; ++k) {
            if (B * T <= 300 && k >= 2) break;
            printf("  workgroup %d, %s wave: total %.2f us\n", k < 2 ? 5 : 300, (k & 1) ? "last" : "first", (ts[k][13] - ts[k][0]) * 0.01);
            for (int i = 1; i <= 13; ++i) printf("     %-48s %6.2f us\n", nm[i - 1], (ts[k][i] - ts[k][i - 1]) * 0.01);
        }
    }
    return 0;
}
