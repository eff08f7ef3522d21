#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k16(float* out, long long* ts, int n) {
    f32x4 a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
    long long t0 = __builtin_readcyclecounter(); long long w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter(); long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) { ts[0] = t1 - t0; ts[1] = w1 - w0; }
}
__global__ void k32(float* out, long long* ts, int n) {
    f32x16 a0, a1; for (int i = 0; i < 16; ++i) { a0[i] = 0; a1[i] = 0; }
    float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
    long long t0 = __builtin_readcyclecounter(); long long w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter(); long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) { ts[0] = t1 - t0; ts[1] = w1 - w0; }
}
int main() {
    float* out; long long* ts; hipMalloc(&out, 1 << 24); hipMalloc(&ts, 64);
    long long h[2];
    for (int waves = 1; waves <= 2; ++waves)
    for (int blocks : {1, 256, 512}) {
        const int n = 4096;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k16, dim3(blocks), dim3(256 * waves), 0, 0, out, ts, n);
        hipDeviceSynchronize(); hipMemcpy(h, ts, 16, hipMemcpyDeviceToHost);
        printf("16x16x4 : %d blocks x %d waves/SIMD: %.1f cycles/MFMA (shader clock), %.1f ns/MFMA -> %.2f GHz\n", blocks, waves, (double)h[0] / (4.0 * n), 10.0 * h[1] / (4.0 * n), (double)h[0] / (10.0 * h[1]));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256 * waves), 0, 0, out, ts, n);
        hipDeviceSynchronize(); hipMemcpy(h, ts, 16, hipMemcpyDeviceToHost);
        printf("32x32x2 : %d blocks x %d waves/SIMD: %.1f cycles/MFMA (shader clock), %.1f ns/MFMA -> %.2f GHz\n", blocks, waves, (double)h[0] / (2.0 * n), 10.0 * h[1] / (2.0 * n), (double)h[0] / (10.0 * h[1]));
    }
    return 0;
}
