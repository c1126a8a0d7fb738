// Micro-benchmark (round 6): which shape of a streaming kernel reaches the rate of torch's elementwise add on this chip (6.3 TB/s read + write against the 5.1 TB/s of
// GroupNorm apply and the 5.8 TB/s of stream_pattern.hip)?  Every kernel reads 1 GiB and writes 1 GiB.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/stream_variants.hip -o tools/_build/stream_variants && tools/_build/stream_variants
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// V loads of 16 bytes per lane, lane-contiguous KiB segments; NT: non-temporal loads / stores; block size B
template <int V, bool NT, int B>
__global__ void __launch_bounds__(B) k_lin(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t base = (size_t)blockIdx.x * (B * V) + threadIdx.x;
    u32x4 a[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const size_t o = base + (size_t)i * B;
        if (o < n16) a[i] = NT ? __builtin_nontemporal_load(src + o) : src[o];
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const size_t o = base + (size_t)i * B;
        a[i][0] += 1u;
        if (o < n16) { if (NT) __builtin_nontemporal_store(a[i], dst + o); else dst[o] = a[i]; }
    }
}
// the blocked storages' pattern (ldraw<hm> / straw<hm>): lane l touches bytes [16 s, +16) and [64 + 16 s, +16) of line l / 4; TRIPS trips with a grid stride
template <bool NT>
__global__ void __launch_bounds__(256) k_half(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t wave_chunks, int trips) {
    const unsigned l = threadIdx.x & 63;
    const size_t stride = (size_t)gridDim.x * 4;
    size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int t = 0; t < trips && w < wave_chunks; ++t, w += stride) {
        const size_t o0 = w * 2048 + (l >> 2) * 128 + (l & 3) * 16, o1 = o0 + 64;
        u32x4 a = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + o0)) : *reinterpret_cast<const u32x4*>(src + o0);
        u32x4 b = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + o1)) : *reinterpret_cast<const u32x4*>(src + o1);
        a[0] += 1u; b[1] ^= 3u;
        if (NT) { __builtin_nontemporal_store(a, reinterpret_cast<u32x4*>(dst + o0)); __builtin_nontemporal_store(b, reinterpret_cast<u32x4*>(dst + o1)); }
        else { *reinterpret_cast<u32x4*>(dst + o0) = a; *reinterpret_cast<u32x4*>(dst + o1) = b; }
    }
}
template <typename F>
static void timeit(const char* name, size_t bytes, F&& launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-58s %7.3f ms  %6.0f GB/s (read + write)\n", name, ms / reps, 2.0 * bytes / (ms / reps * 1e-3) / 1e9);
}
template <int V, bool NT, int B>
static void run_lin(const u32x4* s, u32x4* d, size_t bytes) {
    const size_t n16 = bytes / 16;
    const unsigned grid = (unsigned)((n16 + (size_t)B * V - 1) / ((size_t)B * V));
    char name[96];
    snprintf(name, sizeof name, "linear %d x 16 B per lane, block %4d%s", V, B, NT ? ", non-temporal" : "");
    timeit(name, bytes, [&] { k_lin<V, NT, B><<<grid, B>>>(s, d, n16); });
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    unsigned char *s, *d;
    hipMalloc(&s, bytes); hipMalloc(&d, bytes);
    hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
    const u32x4* s4 = (const u32x4*)s; u32x4* d4 = (u32x4*)d;
    for (int rep = 0; rep < 2; ++rep) {
        run_lin<1, false, 256>(s4, d4, bytes); run_lin<2, false, 256>(s4, d4, bytes); run_lin<4, false, 256>(s4, d4, bytes); run_lin<8, false, 256>(s4, d4, bytes);
        run_lin<1, false, 512>(s4, d4, bytes); run_lin<1, false, 1024>(s4, d4, bytes); run_lin<4, false, 1024>(s4, d4, bytes); run_lin<1, false, 64>(s4, d4, bytes);
        run_lin<1, true, 256>(s4, d4, bytes); run_lin<4, true, 256>(s4, d4, bytes);
        const size_t wc = bytes / 2048;
        for (int trips : {1, 8}) {
            const int grid = (int)((wc / 4 + trips - 1) / trips);
            char name[96];
            snprintf(name, sizeof name, "half-line pattern (fp16+8 blocks), %d trip(s)", trips);
            timeit(name, bytes, [&] { k_half<false><<<grid, 256>>>(s, d, wc, trips); });
            snprintf(name, sizeof name, "half-line pattern (fp16+8 blocks), %d trip(s), non-temporal", trips);
            timeit(name, bytes, [&] { k_half<true><<<grid, 256>>>(s, d, wc, trips); });
        }
    }
    return 0;
}
