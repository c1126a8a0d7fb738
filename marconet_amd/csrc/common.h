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
// 16-byte store of the HBM-bound streaming kernels (straw: up-sample, GroupNorm apply, AdaIN, scatter, converters).  A/B build
// EXTRA_HIPCC_FLAGS=-DMNET_NT_STRAW=1 marks them non-temporal — measured on one box (bench.py, B = 256): fp16x2 203.1 -> 185.2
// images/s (the 8-byte lo pieces of a block become partial-line writes), fp16x3 188.6 -> 188.1: off.  (Non-temporal stores in
// the CONV epilogues: fp16x2 197.7 -> 168.5 images/s.)
#ifndef MNET_NT_STRAW
#define MNET_NT_STRAW 0
#endif
// the streaming kernels' 16-byte load (ldraw).  A/B build -DMNET_NT_LDRAW=1: non-temporal (read-once streams)
#ifndef MNET_NT_LDRAW
#define MNET_NT_LDRAW 0
#endif
__device__ __forceinline__ u32x4 ldg16s(const void* p) {
#if MNET_NT_LDRAW
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#else
    return *reinterpret_cast<const u32x4*>(p);
#endif
}
__device__ __forceinline__ void stg16s(void* p, u32x4 v) {
#if MNET_NT_STRAW
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
#else
    *reinterpret_cast<u32x4*>(p) = v;
#endif
}
__device__ __forceinline__ void stg8s(void* p, u32x2 v) {
#if MNET_NT_STRAW
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
#else
    *reinterpret_cast<u32x2*>(p) = v;
#endif
}

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

template <typename T> __device__ __forceinline__ Raw<T> ldraw(const T* p) { Raw<T> r; r.v = ldg16s(p); return r; }
template <> __device__ __forceinline__ Raw<hs> ldraw<hs>(const hs* p) {
    // (the pointer itself is offset — through an integer round trip hipcc loses the address space and emits FLAT loads / stores, which wait on both counters)
    const unsigned char* q = reinterpret_cast<const unsigned char*>(p) - (size_t)(((unsigned)reinterpret_cast<uintptr_t>(p) >> 5) & 3u) * 16u;
    Raw<hs> r; r.hi = ldg16s(q); r.lo = ldg16s(q + 64); return r;
}
template <typename T> __device__ __forceinline__ void straw(T* p, const Raw<T>& r) { stg16s(p, r.v); }
template <> __device__ __forceinline__ void straw<hs>(hs* p, const Raw<hs>& r) {
    unsigned char* q = reinterpret_cast<unsigned char*>(p) - (size_t)(((unsigned)reinterpret_cast<uintptr_t>(p) >> 5) & 3u) * 16u;
    stg16s(q, r.hi); stg16s(q + 64, r.lo);
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

// ---------------------------------------------------------------------------------------------------------------------
// MNET_F16M — "fp16+8" storage of the fp16x2 precision mode (layout: include/marconet_hip.h).  Same nominal addressing as hs
// (chunk j of a pixel at +32 j bytes).  A chunk of 8 channels = 8 hi halves + 8 lo bytes + the E8M0 exponent of ITS BLOCK's scale,
// which is shared by the 4 chunks of the block: packr<hm> therefore reduces over the 4 lanes of a quad — every kernel that stores
// hm chunks keeps chunk j of a pixel in lane j (mod 4) of a fully active quad (linear thread -> chunk maps, C % 32 == 0).
#ifndef MNET_HM_SIMPLE_LOAD
#define MNET_HM_SIMPLE_LOAD 0
#endif
struct hm { unsigned int bits; };
static_assert(sizeof(hm) == 4, "hm is 4 bytes");
template <> struct Vec<hm> { static constexpr int N = 8; };
template <> struct Raw<hm> { u32x4 hi; u32x2 lo8; int e8; };

// 8-byte slot of the lo bytes of chunk s (channels 8 s .. 8 s + 7) inside the block's 32 lo bytes (order 0-7,16-23,8-15,24-31)
__device__ __forceinline__ int hm_lo_slot(int s) { return ((s & 1) << 1) | (s >> 1); }

// (quad-cooperative like straw<hm>: the 4 lanes of a quad load the 4 chunks of one block.)  Two 16-byte loads per lane — its hi
// halves and one quarter of the block's second half (lanes 0 / 1: the lo bytes of chunks (0, 2) / (1, 3), lane 2: the scale byte) —
// then five DPP moves hand every lane its own 8 lo bytes and the block's exponent.
template <> __device__ __forceinline__ Raw<hm> ldraw<hm>(const hm* p) {
    const unsigned s = ((unsigned)reinterpret_cast<uintptr_t>(p) >> 5) & 3u;
    const unsigned char* blk = reinterpret_cast<const unsigned char*>(p) - (size_t)(s * 32u);      // (pointer arithmetic, not integer: see ldraw<hs>)
    Raw<hm> r;
    r.hi = ldg16s(blk + s * 16u);
#if MNET_HM_SIMPLE_LOAD      // A/B build: three independent loads per lane (hi 16 B, lo 8 B, scale byte)
    r.lo8 = *reinterpret_cast<const u32x2*>(blk + 64 + hm_lo_slot((int)s) * 8);
    r.e8 = blk[96];
    return r;
#endif
    const u32x4 q = ldg16s(blk + 64 + (s < 3u ? s : 2u) * 16u);           // (lane 3 re-reads lane 2's chunk: no divergence, same line)
    const unsigned a0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)q[0], 0x44, 0xf, 0xf, true);     // from lane (s & 1)
    const unsigned a1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)q[1], 0x44, 0xf, 0xf, true);
    const unsigned a2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)q[2], 0x44, 0xf, 0xf, true);
    const unsigned a3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)q[3], 0x44, 0xf, 0xf, true);
    r.lo8 = s < 2u ? u32x2{a0, a1} : u32x2{a2, a3};
    r.e8 = __builtin_amdgcn_update_dpp(0, (int)q[0], 0xAA, 0xf, 0xf, true) & 0xff;                     // lane 2's first byte
    return r;
}
// (called with the 4 lanes of a quad active on the 4 chunks of ONE block, like packr<hm>.)  Every lane issues exactly two 16-byte
// stores — its hi halves and one quarter of the block's second half: lanes 0 / 1 gather the lo bytes of chunks (0, 2) / (1, 3) —
// the storage order — from their partner lane ^ 2 with two DPP moves, lane 2 writes the scale byte, lane 3 the padding; the whole
// 128-byte line is written (no partial-line write-back, deterministic padding).  (8-byte lo stores + predicated scale / padding
// stores measured 24 % slower than the split-half kernels on the up-sample and GroupNorm-apply passes.)
template <> __device__ __forceinline__ void straw<hm>(hm* p, const Raw<hm>& r) {
    const unsigned s = ((unsigned)reinterpret_cast<uintptr_t>(p) >> 5) & 3u;
    unsigned char* blk = reinterpret_cast<unsigned char*>(p) - (size_t)(s * 32u);
    stg16s(blk + s * 16u, r.hi);
    const unsigned p0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r.lo8[0], 0x4E, 0xf, 0xf, true);     // partner (lane ^ 2) lo bytes
    const unsigned p1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r.lo8[1], 0x4E, 0xf, 0xf, true);
    u32x4 v;
    v[0] = s < 2u ? r.lo8[0] : (s == 2u ? (unsigned)r.e8 : 0u);
    v[1] = s < 2u ? r.lo8[1] : 0u;
    v[2] = s < 2u ? p0 : 0u;
    v[3] = s < 2u ? p1 : 0u;
    stg16s(blk + 64 + s * 16u, v);
}
template <> __device__ __forceinline__ Raw<hm> zero_raw<hm>() { Raw<hm> r; r.hi = u32x4{0u, 0u, 0u, 0u}; r.lo8 = u32x2{0u, 0u}; r.e8 = 0; return r; }

// lo scale 2^(E - 127 - 11) as a float (0 for blocks too small to matter)
__device__ __forceinline__ float hm_lo_scale(int e8) { return e8 >= 12 ? __builtin_bit_cast(float, (unsigned)(e8 - 11) << 23) : 0.f; }
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void hm_decode_lo(u32x2 lo8, float sl, float* o) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo8[d], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo8[d], true);
        o[4 * d] = a[0] * sl; o[4 * d + 1] = a[1] * sl; o[4 * d + 2] = b[0] * sl; o[4 * d + 3] = b[1] * sl;
    }
}
// Round 5 — the fp16+8 encode / decode on the mixed-precision VALU forms (MNET_HM_FAST, default on; -DMNET_HM_FAST=0 is the A/B build).
// hipcc turns `(float)h + l * s` into v_cvt_f32_f16 + v_mul + v_add and `(v - (float)f16(v)) * inv` into v_cvt_f16_f32 + v_cvt_f32_f16 + v_sub +
// v_mul NEXT TO the packed v_cvt_pk_f16_f32 it also emits for the stored halves (ISA of the 256x256 tile's epilogue): 5 VALU
// instructions per encoded value, 3.5 per decoded one — and every streaming kernel of the fp16+8 mode decodes and encodes every element.
//   v_fma_mix_f32 d, a, b, c   reads any of its operands as the low / high HALF of a register in place: decode = 1 instruction per value
//                              (l * s + h), the lo residual v - h = h * (-1) + v = 1 instruction (exact, like the subtraction it replaces);
//   v_cvt_scalef32_pk_fp8_f32  divides by a power of two (the exponent field of its scale operand) while it converts: the `* inv` disappears.
// Same values, same roundings: l * s and v - h are exact, so the fused forms round exactly once where the separate ones did.
#ifndef MNET_HM_FAST
#define MNET_HM_FAST 1
#endif
// a * b + half(packed, HI)
template <int HI> __device__ __forceinline__ float fma_mix_c16(float a, float b, unsigned packed) {
    float d;
    if constexpr (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(packed));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(packed));
    return d;
}
// c - half(packed, HI)   (= half * -1.0 + c)
template <int HI> __device__ __forceinline__ float sub_mix_a16(float c, unsigned packed) {
    float d;
    if constexpr (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(c));
    return d;
}
// o[j] (+)= (float)hi[j] + lo8[j] * sl for the 8 channels of a chunk (hi: 8 packed halves, lo8: 8 e4m3 bytes)
template <bool ACC = false>
__device__ __forceinline__ void hm_decode8(u32x4 hi, u32x2 lo8, float sl, float* o) {
#if MNET_HM_FAST
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo8[d], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo8[d], true);
        const float t0 = fma_mix_c16<0>(a[0], sl, hi[2 * d]), t1 = fma_mix_c16<1>(a[1], sl, hi[2 * d]);
        const float t2 = fma_mix_c16<0>(b[0], sl, hi[2 * d + 1]), t3 = fma_mix_c16<1>(b[1], sl, hi[2 * d + 1]);
        if constexpr (ACC) { o[4 * d] += t0; o[4 * d + 1] += t1; o[4 * d + 2] += t2; o[4 * d + 3] += t3; }
        else { o[4 * d] = t0; o[4 * d + 1] = t1; o[4 * d + 2] = t2; o[4 * d + 3] = t3; }
    }
#else
    const f16x8 h = bitcast<f16x8>(hi);
    float l[8];
    hm_decode_lo(lo8, sl, l);
#pragma unroll
    for (int j = 0; j < 8; ++j) { if constexpr (ACC) o[j] += (float)h[j] + l[j]; else o[j] = (float)h[j] + l[j]; }
#endif
}
template <> __device__ __forceinline__ void unpackr<hm>(const Raw<hm>& r, float* o) { hm_decode8(r.hi, r.lo8, hm_lo_scale(r.e8), o); }
// E8M0 byte of the block scale from the block's max |hi| (as a float): 2^(floor(log2 m) - 7)
// (raw form: the weight packer clamps it to its own range)
__device__ __forceinline__ int hm_e8_raw(float m) { return max(0, (int)((__builtin_bit_cast(unsigned, m) >> 23) & 255u) - 7); }
// Activation blocks: floored at 2^(-15 - 7) for a non-zero block (round 6).  Below 2^-14 the hi halves are fp16 SUBNORMALS: their rounding error stops shrinking with the
// block (2^-25 at most), so lo * 2^11 / s grows as the block shrinks and leaves e4m3's range for max |hi| < 2^-15 — where v_cvt_pk_fp8_f32 / v_cvt_scalef32_pk_fp8_f32
// return NaN (0x7f), not the saturated byte (measured: tests/test_round6_gpu.py).  With the floor the scaled residual is at most 2^8; nothing is lost: the
// lo byte's resolution at the floor, 2^-36, is far below the hi half's own 2^-25.  An all-zero block keeps exponent byte 0 (lo bytes 0).
__device__ __forceinline__ int hm_e8_of(float m) {
    const int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 255u) - 7;
    return e > 0 ? max(e, 105) : 0;
}
// 8 lo bytes: e4m3((v - hi) * 2^11 / s), s = 2^(e8 - 127)
__device__ __forceinline__ u32x2 hm_encode_lo_ref(const float* o, const f16x8& h, int e8) {
    const float inv = e8 >= 11 ? __builtin_bit_cast(float, (unsigned)(265 - e8) << 23) : 0.f;      // 2^(11 + 127 - e8)
    u32x2 r;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32((o[4 * d] - (float)h[4 * d]) * inv, (o[4 * d + 1] - (float)h[4 * d + 1]) * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32((o[4 * d + 2] - (float)h[4 * d + 2]) * inv, (o[4 * d + 3] - (float)h[4 * d + 3]) * inv, w, true);
        r[d] = (unsigned)w;
    }
    return r;
}
__device__ __forceinline__ u32x2 hm_encode_lo(const float* o, const f16x8& h, int e8) {
#if MNET_HM_FAST
    // (v - hi) / 2^(e8 - 138): the conversion divides by the power of two whose exponent field its scale operand carries — field e8 - 11.  hi is a
    // HALF: a block's largest |hi| is 0 or >= 2^-24, so e8 is 0 (an all-zero block) or >= 96 — the fields 1 ... 10, and the field 0 that e8 = 11
    // would ask for, never occur; the all-zero block divides by 2^127: zero bytes (the reference form multiplies by 0 there)
    const float sc = __builtin_bit_cast(float, (unsigned)(e8 >= 12 ? e8 - 11 : 254) << 23);
    const u32x4 hp = bitcast<u32x4>(h);
    u32x2 r;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        s16x2 w = {0, 0};
        w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, sub_mix_a16<0>(o[4 * d], hp[2 * d]), sub_mix_a16<1>(o[4 * d + 1], hp[2 * d]), sc, false);
        w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, sub_mix_a16<0>(o[4 * d + 2], hp[2 * d + 1]), sub_mix_a16<1>(o[4 * d + 3], hp[2 * d + 1]), sc, true);
        r[d] = bitcast<unsigned>(w);
    }
    return r;
#else
    return hm_encode_lo_ref(o, h, e8);
#endif
}
__device__ __forceinline__ float quad_max(float m) {      // max over the 4 lanes of a quad (DPP quad_perm: xor 1, xor 2)
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xf, 0xf, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xf, 0xf, true)));
    return m;
}
template <> __device__ __forceinline__ Raw<hm> packr<hm>(const float* o) {
    Raw<hm> r;
    f16x8 h;
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = (f16)o[j]; m = fmaxf(m, fabsf((float)h[j])); }
    r.e8 = hm_e8_of(quad_max(m));
    r.hi = bitcast<u32x4>(h);
    r.lo8 = hm_encode_lo(o, h, r.e8);
    return r;
}
template <> __device__ __forceinline__ float ld_elem<hm>(const hm* px, int c) {
    const unsigned char* blk = reinterpret_cast<const unsigned char*>(px) + (c >> 5) * 128;
    const int ci = c & 31;
    const float hi = (float)reinterpret_cast<const f16*>(blk)[ci];
    const int b = blk[64 + hm_lo_slot(ci >> 3) * 8 + (ci & 7)];
    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(b, false);
    return hi + lo[0] * hm_lo_scale(blk[96]);
}
// one lane per channel, the 32 channels of a block in 32 consecutive lanes (lane % 32 == c % 32), all active
__device__ __forceinline__ void st_block32_hm(hm* px, int c, float v) {
    unsigned char* blk = reinterpret_cast<unsigned char*>(px) + (c >> 5) * 128;
    const int ci = c & 31;
    const f16 h = (f16)v;
    float m = fabsf((float)h);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const int e8 = hm_e8_of(m);
    const float inv = e8 >= 11 ? __builtin_bit_cast(float, (unsigned)(265 - e8) << 23) : 0.f;
    reinterpret_cast<f16*>(blk)[ci] = h;
    blk[64 + hm_lo_slot(ci >> 3) * 8 + (ci & 7)] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32((v - (float)h) * inv, 0.f, 0, false) & 0xff);
    if (ci == 0) blk[96] = (unsigned char)e8;
    if (ci >= 1) blk[96 + ci] = 0;
}

static inline bool aligned128(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 127u) == 0; }

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case MNET_ACT_RELU: return v < 0.f ? v * 0.f : v;         // (not fmaxf: a NaN must stay a NaN, like torch.relu; -inf becomes NaN, see act_apply_vec)
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
// kernel, measured at 10 us per workgroup.)  The cheap family is branch-free: (v < 0 ? v * slope : v) * post (compare + select; ReLU is
// slope 0, so relu(-inf) = NaN: an infinity only ever comes from a half-precision overflow and must stay visible),
// which reproduces act_apply bit for bit AND keeps a NaN a NaN (the earlier max/min form, fmaxf(v,0) + fminf(v,0)*slope, turned NaN
// into 0 — it swallowed the evidence of a half-precision overflow two layers after it happened).
template <int N, bool CHEAP_ONLY = false>
__device__ __forceinline__ void act_apply_vec(float* v, int act) {
    if (act == MNET_ACT_NONE) return;
    if (CHEAP_ONLY && act > MNET_ACT_LRELU_SQRT2) return;
    if (act <= MNET_ACT_LRELU_SQRT2) {
        const float slope = act == MNET_ACT_RELU ? 0.f : 0.2f;
        const float post = act == MNET_ACT_LRELU_SQRT2 ? 1.41421356237309515f : 1.f;
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = (v[q] < 0.f ? v[q] * slope : v[q]) * post;
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
