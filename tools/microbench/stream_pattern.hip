// Micro-benchmark (round 4): does the access pattern of the blocked storages' streaming kernels cost HBM rate?  Every kernel copies the
// same 1 GiB with two 16-byte loads and two 16-byte stores per lane:
//   half  : lane l touches bytes [16 s, 16 s + 16) and [64 + 16 s, 64 + 16 s + 16) of line l / 4 (s = l % 4) — what ldraw<hm> / straw<hm> do:
//           each wave instruction covers HALF of 16 consecutive 128-byte lines, the second instruction the other half
//   full  : lane l touches [16 l, 16 l + 16) of two consecutive KiB — each wave instruction covers 8 whole lines
// one trip per thread (like affine_act_kernel at the bench's sizes) or a grid-stride loop with `trips` trips.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/stream_pattern.hip -o tools/_build/stream_pattern && tools/_build/stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool HALF>
__global__ void __launch_bounds__(256) copy_k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t wave_chunks, int trips) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned l = threadIdx.x & 63;
    const size_t stride = (size_t)gridDim.x * 4;
    size_t w = wave;
    for (int t = 0; t < trips && w < wave_chunks; ++t, w += stride) {
        const size_t base = w * 2048;
        const size_t o0 = HALF ? base + (l >> 2) * 128 + (l & 3) * 16 : base + l * 16;
        const size_t o1 = HALF ? o0 + 64 : o0 + 1024;
        u32x4 a = *reinterpret_cast<const u32x4*>(src + o0), b = *reinterpret_cast<const u32x4*>(src + o1);
        a[0] += 1u; b[1] ^= 3u;
        *reinterpret_cast<u32x4*>(dst + o0) = a;
        *reinterpret_cast<u32x4*>(dst + o1) = b;
    }
}

template <bool HALF>
static void run(const char* name, const unsigned char* s, unsigned char* d, size_t bytes, int trips) {
    const size_t wave_chunks = bytes / 2048;
    const int grid = (int)((wave_chunks / 4 + trips - 1) / trips);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) copy_k<HALF><<<grid, 256>>>(s, d, wave_chunks, trips);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) copy_k<HALF><<<grid, 256>>>(s, d, wave_chunks, trips);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-6s trips %2d: %7.3f ms per GiB copied, %6.0f GB/s (read + write)\n", name, trips, ms / reps, 2.0 * bytes / (ms / reps * 1e-3) / 1e9);
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    unsigned char *s, *d;
    hipMalloc(&s, bytes); hipMalloc(&d, bytes);
    hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
    for (int rep = 0; rep < 2; ++rep)
        for (int trips : {1, 4, 16}) {
            run<true>("half", s, d, bytes, trips);
            run<false>("full", s, d, bytes, trips);
        }
    return 0;
}
