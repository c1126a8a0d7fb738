// Micro-benchmark (round 4): matrix-pipe time of one fp16+8 k-slab's MFMA sequence with nothing else in the way — per wave 16 x
// v_mfma_f32_32x32x16_f16 + 8 x v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 operands) on 8 accumulator blocks, 1 or 2 waves per SIMD.
// Nominal: 16 x 32 + 8 x 64 = 1024 cycles per wave, 2048 per SIMD with two waves.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_slab.hip -o tools/_build/mfma_slab && tools/_build/mfma_slab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>      // 0: the slab mix (f16 k-step 0, f16 k-step 1, scaled)  1: 24 f16 MFMAs  2: 12 scaled MFMAs  3: 8 scaled only  4 / 5: mix with fp4- / fp6-typed scaled operands  6: 8 fp6-typed scaled only
__global__ void __launch_bounds__(512) k(int iters, const float* seed, float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
    for (int b = 0; b < 8; ++b)
        for (int q = 0; q < 16; ++q) acc[b][q] = seed[(lane + q + b) & 63];
    f16x8 a[2][4], bh[2][2];
    i32x8 a8[4], b8[2];
    for (int i = 0; i < 2; ++i)
        for (int f = 0; f < 4; ++f)
            for (int q = 0; q < 8; ++q) a[i][f][q] = (_Float16)seed[(lane * 3 + f + q + i) & 63];
    for (int i = 0; i < 2; ++i)
        for (int f = 0; f < 2; ++f)
            for (int q = 0; q < 8; ++q) bh[i][f][q] = (_Float16)seed[(lane * 5 + f + q + i) & 63];
    for (int f = 0; f < 4; ++f)
        for (int q = 0; q < 8; ++q) a8[f][q] = __float_as_int(seed[(lane + f + q) & 63]) & 0x3f3f3f3f;
    for (int f = 0; f < 2; ++f)
        for (int q = 0; q < 8; ++q) b8[f][q] = __float_as_int(seed[(lane * 7 + f + q) & 63]) & 0x3f3f3f3f;
    const int sa = 120 + (lane & 3), sb = 121;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 1 || MODE == 4 || MODE == 5) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int fa = 0; fa < 4; ++fa)
#pragma unroll
                    for (int fb = 0; fb < 2; ++fb)
                        acc[fa * 2 + fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k2][fa], bh[k2][fb], acc[fa * 2 + fb], 0, 0, 0);
        }
        if (MODE == 1) {
#pragma unroll
            for (int fa = 0; fa < 4; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
                    acc[fa * 2 + fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][fa], bh[1][fb], acc[fa * 2 + fb], 0, 0, 0);
        }
        if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int fa = 0; fa < 4; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
                    acc[fa * 2 + fb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[fa], b8[fb], acc[fa * 2 + fb], 0, 0, 0, sa, 0, sb);
        }
        if (MODE == 2) {
#pragma unroll
            for (int fa = 0; fa < 2; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
                    acc[fa * 2 + fb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[fa + 2], b8[fb], acc[fa * 2 + fb], 0, 0, 0, sa, 0, sb);
        }
        if (MODE == 5 || MODE == 6) {
#pragma unroll
            for (int fa = 0; fa < 4; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
                    acc[fa * 2 + fb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[fa], b8[fb], acc[fa * 2 + fb], 2, 2, 0, sa, 0, sb);   // fp6 (e2m3) x fp6
        }
        if (MODE == 4) {
#pragma unroll
            for (int fa = 0; fa < 4; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
                    acc[fa * 2 + fb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[fa], b8[fb], acc[fa * 2 + fb], 4, 4, 0, sa, 0, sb);   // fp4 x fp4 (same registers: timing only)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int b = 0; b < 8; ++b)
        for (int q = 0; q < 16; ++q) s += acc[b][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int nominal, int waves, int ncu, const float* seed, float* out, unsigned long long* dcyc) {
    const int iters = 4000;
    k<MODE><<<ncu, waves * 64>>>(100, seed, out, dcyc);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<MODE><<<ncu, waves * 64>>>(iters, seed, out, dcyc);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(ncu);
    hipMemcpy(h.data(), dcyc, ncu * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double c = 0;
    for (auto v : h) c += (double)v;
    c /= ncu * (double)iters;
    printf("%-44s %d waves/CU (%d per SIMD): %7.1f cycles per iteration per wave-pair slot, nominal %d per wave -> pipe busy %.3f  (%.3f ms, %.2f GHz)\n",
           name, waves, waves / 4, c, nominal, nominal * (waves / 4.0) / c, ms, c * iters / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    const bool zeros = argc > 1 && argv[1][0] == 'z';
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    float *seed, *out;
    unsigned long long* dcyc;
    hipMalloc(&seed, 64 * 4); hipMalloc(&out, ncu * 512 * 4); hipMalloc(&dcyc, ncu * 8);
    std::vector<float> hs(64);
    for (int i = 0; i < 64; ++i) hs[i] = zeros ? 0.f : 0.01f * (float)((i * 37) % 64 - 32);
    printf(zeros ? "zero operands\n" : "non-zero operands\n");
    hipMemcpy(seed, hs.data(), 256, hipMemcpyHostToDevice);
    for (int waves : {4, 8}) {
        run<1>("24 x v_mfma_f32_32x32x16_f16", 24 * 32, waves, ncu, seed, out, dcyc);
        run<3>("8 x v_mfma_scale_f32_32x32x64_f8f6f4 (fp8)", 8 * 64, waves, ncu, seed, out, dcyc);
        run<2>("12 x v_mfma_scale_f32_32x32x64_f8f6f4 (fp8)", 12 * 64, waves, ncu, seed, out, dcyc);
        run<0>("slab mix: 16 f16 + 8 scaled fp8", 16 * 32 + 8 * 64, waves, ncu, seed, out, dcyc);
        run<4>("16 f16 + 8 scaled fp4 x fp4", 16 * 32 + 8 * 32, waves, ncu, seed, out, dcyc);
        run<5>("16 f16 + 8 scaled fp6 x fp6", 16 * 32 + 8 * 32, waves, ncu, seed, out, dcyc);
        run<6>("8 scaled fp6 x fp6", 8 * 32, waves, ncu, seed, out, dcyc);
    }
    return 0;
}
