// PMC calibration: kernels that move a KNOWN number of bytes in the access patterns of the step's hot kernels, run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE   and   --pmc WRITE_SIZE   (separate passes, tools/pmc_calibrate.sh)
// so that bytes-per-counter-unit can be read off per pattern.  MI355X_MICROARCH.md states the factor only for wide coalesced streaming reads
// (FETCH_SIZE reports half of a 16 B/lane read); profiles/pmc_traffic.json applied it to every kernel.
//   hipcc --offload-arch=gfx950 -O3 -o pmc_calibration tools/experiments/pmc_calibration.hip
// Every kernel touches a 1 GiB buffer once (>> the 256 MiB Infinity Cache and the 32 MiB of L2) unless its name says otherwise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void read_stream(const T* __restrict__ p, size_t n, float* __restrict__ sink) {      // n elements of T, grid-stride
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = p[i];
        acc += reinterpret_cast<const float*>(&v)[0];
    }
    if (acc == 123.456f) *sink = acc;
}

// rows of `rowbytes` bytes; every wave reads a 256-byte segment of 4 consecutive rows per instruction (lane (kk, j): row 4i + kk, bytes 16 j ..):
// the operand pattern of the weight-gradient reductions (pool jobs) and of the MFMA fragment loads
__global__ __launch_bounds__(256) void read_seg256(const float4* __restrict__ p, size_t rows, size_t row_f4, int segs, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    float acc = 0.f;
    const size_t seg = blockIdx.x % segs;                         // column slab of 16 float4 = 256 B
    for (size_t r = ((size_t)(blockIdx.x / segs) * 4 + wave) * 4; r < rows; r += (size_t)(gridDim.x / segs) * 16) {
        const float4 v = p[(r + kk) * row_f4 + seg * 16 + j];
        acc += v.x;
    }
    if (acc == 123.456f) *sink = acc;
}

// re-reads ONE 16 KB matrix per workgroup `reps` times (L2 / L1 hits after the first touch): fragments of a staged weight
__global__ __launch_bounds__(256) void read_l2_resident(const float4* __restrict__ p, int reps, float* __restrict__ sink) {
    float acc = 0.f;
    const float4* m = p + (size_t)(blockIdx.x % 64) * 1024;      // 64 distinct matrices: 1 MiB in all
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < 1024; i += 256) { const float4 v = m[(i + r) & 1023]; acc += v.x; }
    if (acc == 123.456f) *sink = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void write_stream(T* __restrict__ p, size_t n) {
    T v;
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) reinterpret_cast<float*>(&v)[k] = 1.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}

// 64-byte segments (16 lanes x 4 B) at a 256-byte row pitch: the store pattern of a D-layout accumulator tile written as scalars
__global__ __launch_bounds__(256) void write_seg64(float* __restrict__ p, size_t rows) {
    const int lane = threadIdx.x & 63, j = lane & 15, kk = lane >> 4;
    for (size_t r = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; r < rows; r += (size_t)gridDim.x * 16) p[(r + kk) * 64 + j] = 1.f;
}

int main() {
    const size_t BYTES = (size_t)1 << 30;
    float *buf, *sink;
    CHECK(hipMalloc(&buf, BYTES));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 0, BYTES));
    CHECK(hipDeviceSynchronize());
    const int G = 256 * 16;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_stream<float4>, dim3(G), dim3(256), 0, 0, (const float4*)buf, BYTES / 16, sink);       // 1 GiB, 16 B / lane
        hipLaunchKernelGGL(read_stream<float2>, dim3(G), dim3(256), 0, 0, (const float2*)buf, BYTES / 8, sink);        // 1 GiB,  8 B / lane
        hipLaunchKernelGGL(read_stream<float>, dim3(G), dim3(256), 0, 0, (const float*)buf, BYTES / 4, sink);          // 1 GiB,  4 B / lane
        // 64 K rows of 16 KB (a (B*T, C*C) weight-gradient matrix is 384 x 16.6 KB): 64 column slabs of 256 B -> every byte once: 1 GiB
        hipLaunchKernelGGL(read_seg256, dim3(64 * 64), dim3(256), 0, 0, (const float4*)buf, (size_t)65536, (size_t)1024, 64, sink);
        hipLaunchKernelGGL(read_l2_resident, dim3(G), dim3(256), 0, 0, (const float4*)buf, 64, sink);                  // 1 MiB distinct, 4 GiB of loads
        hipLaunchKernelGGL(write_stream<float4>, dim3(G), dim3(256), 0, 0, (float4*)buf, BYTES / 16);                  // 1 GiB, 16 B / lane
        hipLaunchKernelGGL(write_stream<float>, dim3(G), dim3(256), 0, 0, buf, BYTES / 4);                             // 1 GiB,  4 B / lane
        hipLaunchKernelGGL(write_seg64, dim3(G), dim3(256), 0, 0, buf, BYTES / 256);                                   // 256 MiB written: 64 of every 256 B
    }
    CHECK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
