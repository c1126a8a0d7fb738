// 3x3 / stride 1 / pad 1 convolution, LDS-DMA "strip" variant (f16 storage, fp32 accumulate).
//
// conv_igemm_dma.hip fetches one 64-channel activation slab per filter TAP: the three taps of one filter row re-fetch
// the same pixels shifted by one — 2/3 of the activation DMA (and half of all L2→LDS bytes of a 256x256 tile) is
// redundant, and this kernel is power/clock limited, i.e. limited by the bytes it moves.  Here one
// activation STRIP per (64-channel slice, filter row) — the tile's pixels plus one halo pixel on either side of every image
// row — is fetched once and the MFMA B-fragments of the three taps are read from it at row offsets +0 / +1 / +2:
//     L2→LDS bytes per filter row of a 256x256 tile:  3 x (32 + 32) KiB  →  3 x 32 + 36 KiB   (-31 %)
//     of a 64x512 tile:                               3 x (8 + 64)  KiB  →  3 x 8  + 68 KiB   (-57 %)
// Tiles stay runs of BP consecutive pixels (same epilogue, same output addressing as conv_igemm_dma.hip):
//     W >= BP (W % BP == 0): a tile is BP pixels of one image row; strip = BP + 2 rows  (left / right neighbour or zero)
//     W <  BP (BP % W == 0): a tile is BP/W whole image rows;      strip = (BP/W) x (W + 2) rows (zero column at both ends)
// A strip row is a 128-byte line (64 channels of one pixel), XOR-swizzled by its row index like every other LDS image here,
// so a fragment read of 16 consecutive strip rows is conflict-free for every tap offset.
// Zero padding (image border, ragged valid_w, rows above/below the image) is an out-of-range DMA offset → zeros in LDS.
//
// Stream structure (persistent, one workgroup per CU, like conv_igemm_dma.hip): per tile the k-slabs run
//     for 64-channel slice:  for filter row r:  [strip(slice, r)]  for tap s = 0..2:  [weights(slice, r, s)]
// with a 2-stage ring for the weight slabs (issued one slab ahead) and a 2-stage ring for the strips (issued one filter row
// ahead: 3 slabs to land).  The two DMA cursors run independently and both cross into the workgroup's next tile early.
// k is walked in the same order as conv_igemm_dma.hip (slice outer, tap inner) with the same MFMA → identical bits.
#include "conv_dma_common.h"
#ifndef MNET_STRIP_EPI_IN_STAGE
#define MNET_STRIP_EPI_IN_STAGE 1
#endif

template <int BP> struct StripCap { static constexpr int MINW = BP == 256 ? 16 : 32; static constexpr int ROWS = BP + 2 * (BP / MINW); };

// X3: split-half launch (MNET_F16X2): the tensors are walked as f16 with twice the channels — a strip row / weight row of 128 bytes is
//     one 32-channel block, hi halves in chunks 0-3, lo halves in chunks 4-7 — and every (slab, tap) is multiplied three times
//     (hi*hi, hi*lo, lo*hi), exactly like conv_dma_kernel<…, X3>.
// MX (with X3): fp16+8 launch (MNET_F16M) — the slab arithmetic of conv_dma_kernel<…, MX> (two v_mfma_f32_32x32x16_f16 + one block-scaled
//     v_mfma_scale_f32_32x32x64_f8f6f4 per 32x32 output block and slab, same order) with the activation fragments read from the strip.
template <int BC, int BP, int WC, int WP, bool X3 = false, bool MX = false>
__global__ void __launch_bounds__(WC * WP * 64, WC * WP / 4) conv_strip_kernel(const ConvArgs p) {
    constexpr int NW = WC * WP;
    constexpr int FC = BC / WC / 16, FP = BP / WP / 16;
    constexpr int WJ = BC / (8 * NW);                          // weight DMA instructions per wave per slab
    constexpr int SCAP = StripCap<BP>::ROWS;                   // strip capacity in 128-byte rows
    constexpr int XS = (SCAP / 8 + NW - 1) / NW;               // strip DMA instructions per wave per strip (the last may be partial)
    constexpr int WBYTES = BC * 128, SBYTES = SCAP * 128;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(NW == 8 || NW == 16, "8 or 16 waves");
    static_assert(WJ >= 1 && WJ * 8 * NW == BC, "tile / wave-count mismatch");
    static_assert((BC / WC) % 64 == 0 && (BP / WP) % 32 == 0 && SCAP % 8 == 0, "wave tile shape");
    static_assert(!MX || X3, "MX is a 4-byte storage mode");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [W0][W1][S0][S1][epilogue scratch of the fp16+8 tiles]
    constexpr int XB = dma_mx_xpose_bytes<NW, MX>(2 * WBYTES + 2 * SBYTES);   // per wave: the epilogue's stores go through it (dma_epilogue_mx)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int l16 = lane & 15, g = lane >> 4;
    const int rg = lane >> 3, pc = lane & 7;
    const int G = gridDim.x, ntiles = p.ntiles, nk = p.ktiles;
    const int W = p.w, H = p.h;
    const int rows_per_tile = W < BP ? BP / W : 1;             // image rows covered by one tile
    const int SW = W < BP ? W + 2 : BP + 2;                    // strip rows per image row (pixels + 2 halo)
    const int SR = rows_per_tile * SW;                         // strip rows in use

    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    auto tile_coords = [&](int v, int& co0, int& pix0) __attribute__((always_inline)) {
        const int xcd = v & 7;
        const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
        co0 = (t % p.tilesC) * BC;
        pix0 = (t / p.tilesC) * BP;
    };
    auto uni64 = [](unsigned long long v) __attribute__((always_inline)) -> unsigned long long {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };

    // ================================================================== weight cursor (one slab ahead of the multiply)
    unsigned long long bW = 0; int nW = 0;
    unsigned woff[WJ];
    int w_v = blockIdx.x, w_kt = 0, w_stage = 0;               // tile, slab inside the tile, LDS stage
    int w_tap = 0, w_c = 0;                                    // filter tap (r*3+s) and channel offset of slab w_kt
    bool w_live = true;
    auto setup_w = [&](int v) __attribute__((always_inline)) {
        int co0, pix0;
        tile_coords(v, co0, pix0);
        const long long wbytes = (long long)(p.cout - co0) * p.K * 2;
        bW = (unsigned long long)(reinterpret_cast<const f16*>(p.wgt) + (size_t)co0 * p.K);
        nW = (int)(wbytes < 0x7fffffffLL ? wbytes : 0x7fffffffLL);
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int row = (wave + NW * j) * 8 + rg;
            const int ch = MX ? dma_weight_channel_mx(row) : dma_weight_channel<16>(row);
            const int lc = pc ^ ((row >> 1) & 7);
            woff[j] = (co0 + ch < p.cout) ? (unsigned)(ch * p.K * 2 + lc * 16) : OOB;
        }
        w_tap = 0; w_c = 0;
    };
    auto issue_w_hot = [&]() __attribute__((always_inline)) {             // next weight slab of the SAME tile
        unsigned char* sw_ = smem + w_stage * WBYTES;
        const unsigned kb = (unsigned)(w_tap * p.cin + w_c) * 2u;
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)uni64(bW), 0, __builtin_amdgcn_readfirstlane(nW), 0x00020000);
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const unsigned vo = woff[j] == OOB ? OOB : woff[j] + kb;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_void*)(sw_ + (wave + NW * j) * 1024), 16, vo, 0, 0, 0);
        }
        if (++w_tap == 9) { w_tap = 0; w_c += 64; }
        ++w_kt; w_stage ^= 1;
    };
    auto issue_w = [&]() __attribute__((always_inline)) {                 // next weight slab of the stream (may cross into the next tile)
        if (w_kt == nk) {
            w_kt = 0; w_v += G;
            w_live = w_v < ntiles;
            if (w_live) setup_w(w_v);
        }
        if (w_live) issue_w_hot();
    };

    // ================================================================== strip cursor (one filter row ahead of the multiply)
    unsigned long long bX = 0; int nX = 0;
    unsigned sb[XS];                                           // byte offset of this lane's chunk at filter row 0, channel 0
    unsigned sm[XS];                                           // bit r: the pixel exists for filter row r
    int s_v = blockIdx.x, s_gi = 0, s_stage = 0;               // tile, strip index inside the tile (slice*3 + r), LDS stage
    const int ngroups = nk / 3;
    bool s_live = true;
    const long long img_bytes = (long long)H * W * p.c0 * 2;
    auto setup_s = [&](int v) __attribute__((always_inline)) {
        int co0, pix0;
        tile_coords(v, co0, pix0);
        const int n = pix0 / p.howo, rem = pix0 - n * p.howo;
        const int y0 = rem / W, x0 = rem - y0 * W;            // first pixel of the tile (x0 == 0 when W < BP)
        bX = (unsigned long long)(reinterpret_cast<const char*>(p.x0) + (size_t)n * img_bytes);
        nX = (int)img_bytes;
        const int vw = p.valid_w ? min(p.valid_w[n], W) : W;
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            const int R = (wave + NW * j) * 8 + rg;            // strip row filled by this lane
            const int jr = R / SW, xx = R - jr * SW - 1;       // image row inside the tile, column offset (-1 = left halo)
            const int y = y0 + jr, x = x0 + xx;
            const int lcb = (pc ^ ((R >> 1) & 7)) * 16;
            unsigned m = 0;
            if (R < SR && (unsigned)x < (unsigned)vw) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    if ((unsigned)(y + r - 1) < (unsigned)H) m |= 1u << r;
            }
            sm[j] = m;
            sb[j] = (unsigned)((((y - 1) * W + x) * p.c0) * 2 + lcb);   // filter row 0 reads image row y-1 (may wrap: masked)
        }
    };
    auto issue_s_hot = [&]() __attribute__((always_inline)) {             // next strip of the SAME tile
        unsigned char* ss_ = smem + 2 * WBYTES + s_stage * SBYTES;
        const int cs = s_gi / 3, r = s_gi - cs * 3;
        const unsigned uni = (unsigned)((r * W * p.c0 + cs * 64) * 2);
        const unsigned rbit = 1u << r;
        const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)uni64(bX), 0, __builtin_amdgcn_readfirstlane(nX), 0x00020000);
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            if ((wave + NW * j) * 8 < SR) {                    // wave-uniform: rows beyond the strip in use are never read
                const unsigned vo = (sm[j] & rbit) ? sb[j] + uni : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void*)(ss_ + (wave + NW * j) * 1024), 16, vo, 0, 0, 0);
            }
        }
        ++s_gi; s_stage ^= 1;
    };
    auto issue_s = [&]() __attribute__((always_inline)) {                 // next strip of the stream (may cross into the next tile)
        if (s_gi == ngroups) {
            s_gi = 0; s_v += G;
            s_live = s_v < ntiles;
            if (s_live) setup_s(s_v);
        }
        if (s_live) issue_s_hot();
    };

    // ================================================================== compute side
    f32x4 acc[FC][FP];
    f32x16 acc32_unused[1][1];
    int srow[FP];                                              // strip row of this lane's pixel of fragment f, tap s = 0
#pragma unroll
    for (int f = 0; f < FP; ++f) {
        const int pq = wp * (BP / WP) + f * 16 + l16;          // pixel inside the tile
        const int jr = W < BP ? pq / W : 0;
        srow[f] = jr * SW + (pq - jr * W);                     // + s selects the tap (strip column 0 is x = -1)
    }
    // fp16+8: 32x32 fragments; lane = (row lane & 31, k half h = lane >> 5)
    f32x16 acc32[MX ? FC / 2 : 1][MX ? FP / 2 : 1];
    int srow32[MX ? FP / 2 : 1];
    int mx_sa[MX ? FC / 2 : 1];
    if constexpr (MX) {
#pragma unroll
        for (int f = 0; f < FP / 2; ++f) {
            const int pq = wp * (BP / WP) + f * 32 + (lane & 31);
            const int jr = W < BP ? pq / W : 0;
            srow32[f] = jr * SW + (pq - jr * W);
        }
    }
    auto compute_mx = [&](int wst, int sst, int s) __attribute__((always_inline)) {
        const unsigned char* sw_ = smem + wst * WBYTES;
        const unsigned char* ss_ = smem + 2 * WBYTES + sst * SBYTES;
        const int l32 = lane & 31, h = lane >> 5;
        constexpr int FA = FC / 2, FB = FP / 2;
        u32x4 a[2][FA], bh[2][FB];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
            for (int f = 0; f < FA; ++f) a[k2][f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 32 + l32, 2 * k2 + h));
#pragma unroll
            for (int f = 0; f < FB; ++f) bh[k2][f] = *reinterpret_cast<const u32x4*>(ss_ + swz_dma(srow32[f] + s, 2 * k2 + h));
        }
        i32x8 b8[FB], a8[FA];
        int eb[FB];
#pragma unroll
        for (int f = 0; f < FB; ++f) {
            const int R = srow32[f] + s;
            const unsigned char* row = ss_ + R * 128;
            const int sw3 = (R >> 1) & 7;
            eb[f] = *(row + ((6 ^ sw3) << 4));
            const u32x4 lo8 = *reinterpret_cast<const u32x4*>(row + (((4 + h) ^ sw3) << 4));
            b8[f][4] = (int)lo8[0]; b8[f][5] = (int)lo8[1]; b8[f][6] = (int)lo8[2]; b8[f][7] = (int)lo8[3];
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int fa = 0; fa < FA; ++fa)
#pragma unroll
                for (int fb = 0; fb < FB; ++fb)
                    acc32[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bitcast<f16x8>(a[k2][fa]), bitcast<f16x8>(bh[k2][fb]), acc32[fa][fb], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < FB; ++f) {
            const float sc = __builtin_bit_cast(float, (unsigned)eb[f] << 23);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    s16x2 r = {0, 0};
                    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, bitcast<f16x2>(bh[k2][f][2 * d]), sc, false);
                    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, bitcast<f16x2>(bh[k2][f][2 * d + 1]), sc, true);
                    b8[f][2 * k2 + d] = bitcast<int>(r);
                }
        }
#pragma unroll
        for (int f = 0; f < FA; ++f) {
            const u32x4 lo8 = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 32 + l32, 4 + 2 * h));
            const u32x4 hi8 = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 32 + l32, 5 + 2 * h));
            a8[f] = i32x8{(int)lo8[0], (int)lo8[1], (int)lo8[2], (int)lo8[3], (int)hi8[0], (int)hi8[1], (int)hi8[2], (int)hi8[3]};
        }
#pragma unroll
        for (int fa = 0; fa < FA; ++fa)
#pragma unroll
            for (int fb = 0; fb < FB; ++fb)
                acc32[fa][fb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[fa], b8[fb], acc32[fa][fb], 0, 0, 0, mx_sa[fa], 0, eb[fb]);
    };

    auto compute_half = [&](int wst, int sst, int s, int ks) __attribute__((always_inline)) {
        const unsigned char* sw_ = smem + wst * WBYTES;
        const unsigned char* ss_ = smem + 2 * WBYTES + sst * SBYTES;
        const int chunk = ks * 4 + g;
        u32x4 a[FC], b[FP];
#pragma unroll
        for (int f = 0; f < FC; ++f) a[f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 16 + l16, chunk));
#pragma unroll
        for (int f = 0; f < FP; ++f) b[f] = *reinterpret_cast<const u32x4*>(ss_ + swz_dma(srow[f] + s, chunk));
#pragma unroll
        for (int fa = 0; fa < FC; ++fa)
#pragma unroll
            for (int fb = 0; fb < FP; ++fb)
                acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a[fa]), bitcast<f16x8>(b[fb]), acc[fa][fb], 0, 0, 0);
    };

    auto compute_x3 = [&](int wst, int sst, int s) __attribute__((always_inline)) {
        const unsigned char* sw_ = smem + wst * WBYTES;
        const unsigned char* ss_ = smem + 2 * WBYTES + sst * SBYTES;
        u32x4 a[FC], bh[FP], bl[FP];
#pragma unroll
        for (int f = 0; f < FC; ++f) a[f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 16 + l16, g));
#pragma unroll
        for (int f = 0; f < FP; ++f) bh[f] = *reinterpret_cast<const u32x4*>(ss_ + swz_dma(srow[f] + s, g));
#pragma unroll
        for (int fa = 0; fa < FC; ++fa)
#pragma unroll
            for (int fb = 0; fb < FP; ++fb)
                acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a[fa]), bitcast<f16x8>(bh[fb]), acc[fa][fb], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < FP; ++f) bl[f] = *reinterpret_cast<const u32x4*>(ss_ + swz_dma(srow[f] + s, 4 + g));
#pragma unroll
        for (int fa = 0; fa < FC; ++fa)
#pragma unroll
            for (int fb = 0; fb < FP; ++fb)
                acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a[fa]), bitcast<f16x8>(bl[fb]), acc[fa][fb], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < FC; ++f) a[f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 16 + l16, 4 + g));
#pragma unroll
        for (int fa = 0; fa < FC; ++fa)
#pragma unroll
            for (int fb = 0; fb < FP; ++fb)
                acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a[fa]), bitcast<f16x8>(bh[fb]), acc[fa][fb], 0, 0, 0);
    };

    // ---- prime: strip 0 and weight slab 0 of the first tile
    setup_w(w_v);
    setup_s(s_v);
    issue_s();
    issue_w();
    int c_wst = 0, c_sst = 0;

    for (int c_v = blockIdx.x; c_v < ntiles; c_v += G) {
#pragma unroll
        for (int a = 0; a < FC; ++a)
#pragma unroll
            for (int b = 0; b < FP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (MX) {
#pragma unroll
            for (int a = 0; a < FC / 2; ++a)
#pragma unroll
                for (int b = 0; b < FP / 2; ++b)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc32[a][b][q] = 0.f;
            int co0_, pix0_;
            tile_coords(c_v, co0_, pix0_);
            const unsigned char* wexp = reinterpret_cast<const unsigned char*>(p.wgt) + (size_t)p.cout * p.K * 2;
#pragma unroll
            for (int f = 0; f < FC / 2; ++f) {
                const int ch = co0_ + dma_weight_channel_mx(wc * (BC / WC) + f * 32 + (lane & 31));
                mx_sa[f] = ch < p.cout ? (int)wexp[ch] : 0;      // (every slab waits vmcnt(0): these loads need no extra fence)
            }
        }
        int s = 0;                                             // tap column of slab kt (kt % 3)
        // hot iterations: both cursors stay inside this tile (the weight cursor crosses at kt = nk-1, the strip cursor at nk-3)
        for (int kt = 0; kt < nk - 3; ++kt) {
            VMCNT(0);                                          // everything this wave issued has landed ...
            __builtin_amdgcn_s_barrier();                      // ... everyone's; the stages refilled next are no longer read
            asm volatile("" ::: "memory");
            issue_w_hot();
            if (s == 0) issue_s_hot();
            if constexpr (MX) compute_mx(c_wst, c_sst, s);
            else if constexpr (X3) compute_x3(c_wst, c_sst, s);
            else { compute_half(c_wst, c_sst, s, 0); compute_half(c_wst, c_sst, s, 1); }
            c_wst ^= 1;
            if (++s == 3) { s = 0; c_sst ^= 1; }
        }
        // last filter row: the strip cursor, then the weight cursor, move on to this workgroup's next tile
#pragma unroll 1
        for (int t = 0; t < 3; ++t) {
            VMCNT(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_w();
            if (t == 0) issue_s();
            if constexpr (MX) compute_mx(c_wst, c_sst, t);
            else if constexpr (X3) compute_x3(c_wst, c_sst, t);
            else { compute_half(c_wst, c_sst, t, 0); compute_half(c_wst, c_sst, t, 1); }
            c_wst ^= 1;
        }
        c_sst ^= 1;
        int co0, pix0;
        tile_coords(c_v, co0, pix0);
        if constexpr (MX) {
            if constexpr (MNET_STRIP_EPI_IN_STAGE && XB == 1024 && SBYTES >= NW * 4096) {
                // round 6: the 64x512 tile has 1 KiB of scratch per wave left behind its stages — the epilogue's 64-lane transposition round became four 16-lane rounds (one
                // LDS round trip and one 256-byte store instruction each).  The strip stage the tile's last filter row was read from is free from here until the next tile's
                // second strip is requested (after that tile's first slab barrier): 4 KiB per wave of it are the scratch of the full-size round.  (c_sst already points at the
                // NEXT tile's first strip, which may be landing: the other stage is the consumed one.)
                __syncthreads();                                   // everyone has read the last strip
                dma_epilogue_mx<BC, BP, WC, WP, FC, FP, 64>(p, acc32, co0, pix0, wc, wp, lane, smem + 2 * WBYTES + (c_sst ^ 1) * SBYTES + wave * 4096);
            } else
            dma_epilogue_mx<BC, BP, WC, WP, FC, FP, (XB == 1024 ? 16 : 64)>(p, acc32, co0, pix0, wc, wp, lane, XB ? smem + 2 * WBYTES + 2 * SBYTES + wave * XB : nullptr);
        }
        else dma_epilogue<BC, BP, WC, WP, 16, 0, FC, FP, X3>(p, acc, acc32_unused, co0, pix0, wc, wp, lane);
    }
}

template <int BC, int BP, int WC, int WP, bool X3 = false, bool MX = false>
static int launch_strip_cfg(const ConvArgs& a, hipStream_t st) {
    constexpr int LDS0 = 2 * BC * 128 + 2 * StripCap<BP>::ROWS * 128;
    constexpr int LDS = LDS0 + WC * WP * dma_mx_xpose_bytes<WC * WP, MX>(LDS0);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv_strip_kernel<BC, BP, WC, WP, X3, MX>;
    static thread_local DeviceOnce attr_once;      // per instantiation, per thread, per device
    if (!attr_once.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return mnet_fail(MNET_E_LAUNCH, "hipFuncSetAttribute(strip): %s", hipGetErrorString(e));
        attr_once.mark();
    }
    ConvArgs b = a;
    b.tilesC = (a.cout + BC - 1) / BC;
    b.ntiles = b.tilesC * (a.npix / BP);
    int grid = b.ntiles;
    const int lim = dma_grid_limit();
    if (grid > lim && !a.one_tile_per_wg) grid = lim & ~7;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WC * WP * 64), LDS, st, b);
    MNET_LAUNCH_CHECK("conv_strip_kernel");
    return MNET_OK;
}

// tile configuration the strip kernel would use for this launch (0: 256x256, 16 waves; 1: 64x512, 8 waves), or -1 when
// the launch is not eligible: 3x3 / stride 1 / pad 1, one source tensor, whole-row (or whole-segment) tiles that never
// straddle an image, and everything conv_dma_eligible already requires
int conv_strip_pick(const ConvArgs& a, int dtype, bool explicit_request) {
    // 256x256 tiles: measured neutral (89.8 + 5.1 vs 94.5 ms per bench step; 128-VGPR budget of its 16 waves is exhausted,
    // 20 spills) — AUTO keeps the per-tap kernel there unless MNET_STRIP_256=1; the 64x512 tile (8 waves) gains 20 %.
    static const bool auto256 = [] { const char* e = getenv("MNET_STRIP_256"); return e && atoi(e) != 0; }();
    if ((dtype != MNET_F16 && dtype != MNET_F16X2 && dtype != MNET_F16M) || !conv_dma_eligible(a, dtype)) return -1;
    // fp16+8, cout >= 256: the 8-wave 256x256 strip tile (round 4) — explicit request or MNET_MX_STRIP256=1 (A/B knob)
    static const bool mx256 = [] { const char* e = getenv("MNET_MX_STRIP256"); return e && atoi(e) != 0; }();
    const bool mx_big = dtype == MNET_F16M && a.cout >= 256 && a.cout % 256 == 0 && (explicit_request || mx256);
    if (dtype != MNET_F16 && a.cout >= 128 && !mx_big) return -1;        // split-half: the 64x512 tile only (the big tiles take the 8-wave per-tap forms)
    if (a.cout >= 256 && !explicit_request && !auto256 && !mx_big) return -1;
    if (a.kh != 3 || a.kw != 3 || a.sh != 1 || a.sw != 1 || a.ph != 1 || a.pw != 1 || a.c1 != 0 || a.x1) return -1;
    if (a.ho != a.h || a.wo != a.w) return -1;
    int cfg, bp;
    if (a.cout >= 256) { cfg = 0; bp = 256; }
    else if (a.cout < 128) { cfg = 1; bp = 512; }
    else {                                                     // cout 128: a 512-pixel strip pair + weights exceed the LDS → 128x256
        static const bool s128 = [] { const char* e = getenv("MNET_STRIP_128"); return e && atoi(e) != 0; }();   // A/B knob
        if (!explicit_request && !s128) return -1;
        cfg = 2; bp = 256;
    }
    if (a.npix < 256 * 256 || a.npix % bp != 0) return -1;
    const int minw = bp == 256 ? 16 : 32;
    if (a.w < minw || a.w % 16 != 0) return -1;
    if (a.w < bp ? (bp % a.w != 0 || a.howo % bp != 0) : (a.w % bp != 0)) return -1;
    if ((long long)a.h * a.w * a.c0 * 2 >= 0x7fffffffLL) return -1;       // one image per buffer descriptor
    return cfg;
}

int launch_conv_strip(const ConvArgs& a, hipStream_t st, int cfg) {
    if (a.split == 2) {
        if (cfg == 1) return launch_strip_cfg<64, 512, 1, 8, true, true>(a, st);
        if (cfg == 0) return launch_strip_cfg<256, 256, 2, 4, true, true>(a, st);      // 8 waves, 128x64 per wave (the per-tap id 6 / 11 tile shape)
        return mnet_fail(MNET_E_ARG, "conv: strip configuration %d has no fp16+8 form", cfg);
    }
    if (a.split) {
        if (cfg == 1) return launch_strip_cfg<64, 512, 1, 8, true>(a, st);
        return mnet_fail(MNET_E_ARG, "conv: strip configuration %d has no split-half form", cfg);
    }
    if (cfg == 0) return launch_strip_cfg<256, 256, 4, 4>(a, st);
    if (cfg == 1) return launch_strip_cfg<64, 512, 1, 8>(a, st);
    if (cfg == 2) return launch_strip_cfg<128, 256, 2, 4>(a, st);
    return mnet_fail(MNET_E_ARG, "conv: unknown strip configuration %d", cfg);
}
