// MFMA issue-rate microbenchmark: waves per SIMD x chain length x shape, random fp16 operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

// ORDER 0: four MFMAs in a row on each accumulator; 1: two accumulators alternate; 2: all accumulators in turn
template <int ORDER>
__global__ __launch_bounds__(512) void k32(const half8_t* __restrict__ in, float* __restrict__ out, int iters) {
    half8_t a[4], b[4];
    for (int t = 0; t < 4; t++) { a[t] = in[(threadIdx.x * 8 + t) % 4096]; b[t] = in[(threadIdx.x * 8 + 4 + t) % 4096]; }
    f16v acc[8];
    for (int c = 0; c < 8; c++) for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
    for (int it = 0; it < iters; it++) {
        if (ORDER == 0) {
#pragma unroll
            for (int c = 0; c < 8; c++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], b[t], acc[c], 0, 0, 0);
        } else if (ORDER == 1) {
#pragma unroll
            for (int c = 0; c < 8; c += 2)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], b[t], acc[c], 0, 0, 0);
                    acc[c + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], b[t], acc[c + 1], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int c = 0; c < 8; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], b[t], acc[c], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int c = 0; c < 8; c++) for (int i = 0; i < 16; i++) s += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ORDER>
void run32(const char* name, int threads, const half8_t* in, float* out) {
    const int iters = 2000, grid = 256;
    k32<ORDER><<<grid, threads>>>(in, out, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k32<ORDER><<<grid, threads>>>(in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 32 * (threads / 64) * grid;
    printf("%-28s threads=%d: %.3f ms  %.0f TFLOP/s\n", name, threads, ms, mfmas * 32768.0 / ms / 1e9);
}

template <int SHAPE, int NACC>
__global__ __launch_bounds__(512) void k(const half8_t* __restrict__ in, float* __restrict__ out, int iters) {
    half8_t a[4], b[4];
    for (int t = 0; t < 4; t++) { a[t] = in[(threadIdx.x * 8 + t) % 4096]; b[t] = in[(threadIdx.x * 8 + 4 + t) % 4096]; }
    if (SHAPE == 32) {
        f16v acc[NACC];
        for (int c = 0; c < NACC; c++) for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int c = 0; c < NACC; c++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], b[t], acc[c], 0, 0, 0);
        }
        float s = 0;
        for (int c = 0; c < NACC; c++) for (int i = 0; i < 16; i++) s += acc[c][i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f4v acc[NACC];
        for (int c = 0; c < NACC; c++) for (int i = 0; i < 4; i++) acc[c][i] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int c = 0; c < NACC; c++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b[t], acc[c], 0, 0, 0);
        }
        float s = 0;
        for (int c = 0; c < NACC; c++) for (int i = 0; i < 4; i++) s += acc[c][i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int SHAPE, int NACC>
void run(const char* name, int threads, const half8_t* in, float* out) {
    const int iters = 2000, grid = 256;
    k<SHAPE, NACC><<<grid, threads>>>(in, out, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<SHAPE, NACC><<<grid, threads>>>(in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * NACC * 4 * (threads / 64) * grid;
    const double flops = mfmas * (SHAPE == 32 ? 32768.0 : 16384.0);
    printf("%-28s threads=%d: %.3f ms  %.0f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", name, threads, ms, flops / ms / 1e9,
           ms * 1e6 / (mfmas / (grid * 4)));
}

int main() {
    std::vector<_Float16> h(4096 * 8);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2 - 1);
    half8_t* in; float* out;
    hipMalloc(&in, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run32<0>("32x32 order: chains of 4", 512, in, out);
    run32<1>("32x32 order: 2 alternate", 512, in, out);
    run32<2>("32x32 order: 8 in turn", 512, in, out);
    run32<0>("32x32 order: chains of 4", 512, in, out);
    run32<1>("32x32 order: 2 alternate", 512, in, out);
    run32<2>("32x32 order: 8 in turn", 512, in, out);
    run<32, 8>("32x32x16 8 acc (chain 4)", 256, in, out);
    run<32, 8>("32x32x16 8 acc (chain 4)", 512, in, out);
    run<32, 1>("32x32x16 1 acc", 256, in, out);
    run<32, 1>("32x32x16 1 acc", 512, in, out);
    run<16, 8>("16x16x32 8 acc (chain 4)", 256, in, out);
    run<16, 8>("16x16x32 8 acc (chain 4)", 512, in, out);
    run<16, 32>("16x16x32 32 acc (chain 4)", 512, in, out);
    return 0;
}
