// Micro-benchmark (round 4): what does the L2 -> LDS DMA path (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction) deliver per CU
// when nothing else runs?  The three conv kernels all sit at 17-19 B/clk/CU of L2->LDS traffic — is that a ceiling of
// the path or of the kernels' schedule?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_dma_peak.hip -o tools/_build/lds_dma_peak && tools/_build/lds_dma_peak
// Modes: waves per CU issuing (1, 2, 4, 8, 16), pieces in flight per wave before `s_waitcnt vmcnt(0)` (4, 8, 16), source window per CU
// (16 KiB: L1-resident; 1 MiB shared by the XCD: L2-resident), access pattern = the conv kernels' (8 lanes fetch one 128-byte row).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;

template <int INFLIGHT>
__global__ void __launch_bounds__(1024) dma_kernel(const unsigned char* src, unsigned window_bytes, int active_waves, int iters, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave >= active_waves) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)window_bytes, 0x00020000);
    // lane -> (row = lane / 8, 16-byte chunk = lane % 8) of an [8 rows][128 B] piece; rows 2 KiB apart in the source (a 512-channel fp16+8 pixel row)
    unsigned off = (unsigned)((lane >> 3) * 2048 + (lane & 7) * 16) + (unsigned)(blockIdx.x * 7919 % 64) * 16384u + (unsigned)wave * 128u;
    const unsigned mask = window_bytes - 1;
    unsigned char* dst = smem + wave * (INFLIGHT * 1024);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + j * 1024), 16, off & mask, 0, 0, 0);
            off += 16384u + 1024u;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && wave == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int INFLIGHT>
static void run(const unsigned char* src, unsigned window, int waves, int iters, unsigned long long* dcyc, int ncu) {
    const int lds = waves * INFLIGHT * 1024;      // <= 128 KiB in every mode below
    hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<INFLIGHT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    dma_kernel<INFLIGHT><<<ncu, 1024, lds>>>(src, window, waves, 50, dcyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    dma_kernel<INFLIGHT><<<ncu, 1024, lds>>>(src, window, waves, iters, dcyc);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(ncu);
    hipMemcpy(h.data(), dcyc, ncu * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto v : h) cyc += (double)v;
    cyc /= ncu;
    const double bytes_per_cu = (double)waves * iters * INFLIGHT * 1024.0;
    printf("window %7u KiB  waves/CU %2d  in flight/wave %2d : %7.1f B/clk/CU (shader cycles)  %7.1f GB/s/CU  chip %6.2f TB/s  (%.3f ms, %.0f cycles per piece per CU)\n",
           window >> 10, waves, INFLIGHT, bytes_per_cu / cyc, bytes_per_cu / (ms * 1e-3) / 1e9, bytes_per_cu * ncu / (ms * 1e-3) / 1e12, ms,
           cyc / ((double)waves * iters * INFLIGHT));
}

int main() {
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned char* src;
    const size_t total = 64u << 20;
    hipMalloc(&src, total);
    hipMemset(src, 1, total);
    unsigned long long* dcyc;
    hipMalloc(&dcyc, 1024 * sizeof(unsigned long long));
    for (unsigned window : {1u << 14, 1u << 20, 1u << 25}) {
        for (int waves : {1, 2, 4, 8, 16}) {
            run<8>(src, window, waves, 2000, dcyc, ncu);
        }
        run<4>(src, window, 8, 4000, dcyc, ncu);
        run<16>(src, window, 8, 1000, dcyc, ncu);
    }
    return 0;
}
