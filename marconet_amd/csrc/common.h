// Shared device helpers for the gfx950 kernels of libmarconet_hip.so.  gfx950 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/marconet_hip.h"

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // one raw 16-byte chunk
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// host-side error plumbing (defined in api.hip)
int mnet_fail(int code, const char* fmt, ...);

#define MNET_CHECK_ARG(cond, ...) do { if (!(cond)) return mnet_fail(MNET_E_ARG, __VA_ARGS__); } while (0)
#define MNET_CHECK_ALIGN(cond, ...) do { if (!(cond)) return mnet_fail(MNET_E_ALIGN, __VA_ARGS__); } while (0)
#define MNET_LAUNCH_CHECK(what) do { hipError_t e__ = hipGetLastError(); \
    if (e__ != hipSuccess) return mnet_fail(MNET_E_LAUNCH, "%s: %s", what, hipGetErrorString(e__)); } while (0)

// One-time per-device work of a kernel instantiation (hipFuncSetAttribute is per device): `static thread_local DeviceOnce once;`
// then `if (!once.done()) { ...; once.mark(); }` — keyed by the CURRENT device, so a process that drives several GPUs raises the
// attribute on each of them.
struct DeviceOnce {
    unsigned long long mask[4] = {0, 0, 0, 0};
    static int dev() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0; return d & 255; }
    bool done() const { const int d = dev(); return (mask[d >> 6] >> (d & 63)) & 1ull; }
    void mark() { const int d = dev(); mask[d >> 6] |= 1ull << (d & 63); }
};

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename To, typename From>
__device__ __forceinline__ To bitcast(const From& v) { return __builtin_bit_cast(To, v); }

__device__ __forceinline__ u32x4 ldg16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void stg16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// number of elements in a 16-byte chunk
template <typename T> struct ChunkOf { static constexpr int N = 16 / sizeof(T); };

// unpack a 16-byte chunk to fp32 lanes and back (f16: 8 values, f32: 4 values)
template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(u32x4 raw, float* o) {
        f32x4 v = bitcast<f32x4>(raw); o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
    static __device__ __forceinline__ u32x4 pack(const float* o) {
        f32x4 v = {o[0], o[1], o[2], o[3]}; return bitcast<u32x4>(v);
    }
};
template <> struct Vec<f16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(u32x4 raw, float* o) {
        f16x8 v = bitcast<f16x8>(raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
    }
    static __device__ __forceinline__ u32x4 pack(const float* o) {
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (f16)o[j];
        return bitcast<u32x4>(v);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// MNET_F16X2 — "split half" storage of the fp16x3 precision mode.  A logical element is a PAIR of halves (hi, lo) with
// value = float(hi) + float(lo), hi = f16(v), lo = f16(v - float(hi)): ~22 significant bits, the range of fp16.
// Layout of an NHWC tensor [.., C] (C % 32 == 0, base 128-byte aligned): per pixel 4*C bytes, in blocks of 32 channels —
//     bytes [128 b, 128 b + 64)      hi of channels 32 b .. 32 b + 31
//     bytes [128 b + 64, 128 b + 128) lo of the same channels
// so that ONE 128-byte k-slab of the implicit-GEMM kernels is exactly one 32-channel block: 16-byte chunks 0-3 are the MFMA
// operand of the hi part, chunks 4-7 of the lo part, and x*w is evaluated as hi*hi + hi*lo + lo*hi (three fp16 MFMAs into one
// fp32 accumulator) from a single LDS image of the slab.  Conv weights of this mode hold hi/lo of 256*W (exponent offset: the
// lo parts of typical |W| ~ 1e-2 stay normal numbers); the conv epilogue multiplies the accumulator by 2^-8.
// `hs` is the 4-byte element type tag; pointer arithmetic in units of hs gives NOMINAL addresses (chunk j of a pixel at
// +32 j bytes) which ldraw / straw map to the real hi / lo locations (chunk j = block j/4, sub-chunk j%4: hi at
// 128 (j/4) + 16 (j%4), lo 64 bytes further).
struct hs { unsigned short hi_bits, lo_bits; };
static_assert(sizeof(hs) == 4, "hs is 4 bytes");
#define MNET_SPLIT_WSCALE 256.0f
#define MNET_SPLIT_WSCALE_INV 0.00390625f

template <> struct Vec<hs> { static constexpr int N = 8; };

// raw storage of one chunk of Vec<T>::N consecutive channels
template <typename T> struct Raw { u32x4 v; };
template <> struct Raw<hs> { u32x4 hi, lo; };

template <typename T> __device__ __forceinline__ Raw<T> ldraw(const T* p) { Raw<T> r; r.v = ldg16(p); return r; }
template <> __device__ __forceinline__ Raw<hs> ldraw<hs>(const hs* p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const unsigned char* q = reinterpret_cast<const unsigned char*>(a - ((a >> 5) & 3u) * 16u);
    Raw<hs> r; r.hi = ldg16(q); r.lo = ldg16(q + 64); return r;
}
template <typename T> __device__ __forceinline__ void straw(T* p, const Raw<T>& r) { stg16(p, r.v); }
template <> __device__ __forceinline__ void straw<hs>(hs* p, const Raw<hs>& r) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    unsigned char* q = reinterpret_cast<unsigned char*>(a - ((a >> 5) & 3u) * 16u);
    stg16(q, r.hi); stg16(q + 64, r.lo);
}
template <typename T> __device__ __forceinline__ Raw<T> zero_raw() { Raw<T> r; r.v = u32x4{0u, 0u, 0u, 0u}; return r; }
template <> __device__ __forceinline__ Raw<hs> zero_raw<hs>() { Raw<hs> r; r.hi = u32x4{0u, 0u, 0u, 0u}; r.lo = r.hi; return r; }

template <typename T> __device__ __forceinline__ void unpackr(const Raw<T>& r, float* o) { Vec<T>::unpack(r.v, o); }
template <> __device__ __forceinline__ void unpackr<hs>(const Raw<hs>& r, float* o) {
    const f16x8 h = bitcast<f16x8>(r.hi), l = bitcast<f16x8>(r.lo);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)h[j] + (float)l[j];
}
// hi = f16(v), lo = f16(v - hi)  (both round-to-nearest-even; v - hi is exact in fp32)
__device__ __forceinline__ void split8(const float* o, u32x4& hi, u32x4& lo) {
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = (f16)o[j]; l[j] = (f16)(o[j] - (float)h[j]); }
    hi = bitcast<u32x4>(h); lo = bitcast<u32x4>(l);
}
template <typename T> __device__ __forceinline__ Raw<T> packr(const float* o) { Raw<T> r; r.v = Vec<T>::pack(o); return r; }
template <> __device__ __forceinline__ Raw<hs> packr<hs>(const float* o) { Raw<hs> r; split8(o, r.hi, r.lo); return r; }

// single element c of the pixel whose first channel is at `px` (layout kernels; not on a hot path)
template <typename T> __device__ __forceinline__ float ld_elem(const T* px, int c) { return (float)px[c]; }
template <> __device__ __forceinline__ float ld_elem<hs>(const hs* px, int c) {
    const f16* q = reinterpret_cast<const f16*>(px) + (c >> 5) * 64 + (c & 31);
    return (float)q[0] + (float)q[32];
}
template <typename T> __device__ __forceinline__ void st_elem(T* px, int c, float v) { px[c] = (T)v; }
template <> __device__ __forceinline__ void st_elem<hs>(hs* px, int c, float v) {
    f16* q = reinterpret_cast<f16*>(px) + (c >> 5) * 64 + (c & 31);
    const f16 h = (f16)v;
    q[0] = h; q[32] = (f16)(v - (float)h);
}

static inline bool aligned128(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 127u) == 0; }

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case MNET_ACT_RELU: return fmaxf(v, 0.f);
        case MNET_ACT_LRELU: return v > 0.f ? v : v * 0.2f;
        case MNET_ACT_LRELU_SQRT2: return (v > 0.f ? v : v * 0.2f) * 1.41421356237309515f;
        case MNET_ACT_TANH: return tanhf(v);
        case MNET_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        case MNET_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// Activation over N register values with ONE wave-uniform branch per activation kind.  (A per-element `switch (act)` in an
// unrolled epilogue compiles to a scalar branch chain per value — ~1000 branches and ~100 KB of code in the 256x256 conv
// kernel, measured at 10 us per workgroup.)  The cheap family is branch-free: (max(v,0) + min(v,0)*slope) * post, which
// reproduces act_apply bit for bit (one of the two terms is zero; the products round exactly as there).
template <int N, bool CHEAP_ONLY = false>
__device__ __forceinline__ void act_apply_vec(float* v, int act) {
    if (act == MNET_ACT_NONE) return;
    if (CHEAP_ONLY && act > MNET_ACT_LRELU_SQRT2) return;
    if (act <= MNET_ACT_LRELU_SQRT2) {
        const float slope = act == MNET_ACT_RELU ? 0.f : 0.2f;
        const float post = act == MNET_ACT_LRELU_SQRT2 ? 1.41421356237309515f : 1.f;
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = (fmaxf(v[q], 0.f) + fminf(v[q], 0.f) * slope) * post;
    } else if (act == MNET_ACT_TANH) {
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = tanhf(v[q]);
    } else if (act == MNET_ACT_GELU) {
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = 0.5f * v[q] * (1.f + erff(v[q] * 0.70710678118654752f));
    } else {
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = 1.f / (1.f + expf(-v[q]));
    }
}

__device__ __forceinline__ float swish_f(float v) { return v / (1.f + expf(-v)); }

// 64-lane butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
