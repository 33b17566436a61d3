// Fused Inception-ResNet-B block ("Block17", facenet/models/inception_resnet_v1.py:153-204) for the BN-folded inference /
// mining forward: ONE launch per block instead of five convolution launches.
//
//     t0  = relu(conv1x1(x,   896 -> 128) + b)        t1a = relu(conv1x1(x, 896 -> 128) + b)
//     t1b = relu(conv1x7(t1a, 128 -> 128) + b)        t1c = relu(conv7x1(t1b, 128 -> 128) + b)
//     out = act(x + scale * (conv1x1(concat(t0, t1c), 256 -> 896) + bias))
//
// Inference has no BatchNorm statistics barrier between the layers (the statistics are folded into weights and biases,
// facenet/tfutils.py:244-250), so the whole block of one image is a chain of four GEMM stages whose intermediate
// activations (8 x 8 pixels x 128 .. 256 channels, <= 32 KB) never leave LDS:
//   * a workgroup (4 waves) owns ONE image: 64 pixels = four 16-row MFMA fragments; wave w owns a quarter of every stage's
//     output columns, so each A fragment read from LDS feeds NREP MFMAs;
//   * activations live in LDS as 32-channel slices of 64-byte pixel rows, the 16-byte slot XOR-ed by ((row >> 2) & 1) << 1
//     (the halo kernel's layout: any run of 16 pixels is conflict free for ds_read_b128).  t1a is kept with a 3-pixel zero
//     halo left and right, t1b with a 3-row halo above and below: the seven taps of the 1x7 / 7x1 layers are seven shifted
//     views of the same LDS image, no gather, no bounds test;
//   * only weights stream: per k step (32 input channels of one tap) a [columns][32] slab goes global -> registers -> LDS
//     while the previous slab is multiplied (issue early / write late); the trunk tile of stage 1 streams the same way;
//   * the epilogue of stages 1-3 adds the folded bias, applies ReLU and writes f16/bf16 straight into the next stage's LDS
//     image; stage 4 goes through an fp32 C tile for 16-byte coalesced stores, adds the trunk (re-read from L2) and the bias
//     in fp32 exactly like the un-fused epilogue (resid + scale * (acc + bias), one rounding).
// Work per image: 88 MFLOP and 1.38 MB of weights; the weight stream (L2 -> CU, ~64 B/clk) and the MFMA time are about equal.
#include "common.h"
#include "../../include/facenet_hip.h"
#include <cstdlib>

namespace fn {

struct Block17Args {
    const unsigned short* x;       // [N, 8, 8, 896] trunk
    unsigned short* y;             // [N, 8, 8, 896] block output
    const unsigned short* w_t0;    // [128][896]
    const unsigned short* w_t1a;   // [128][896]
    const unsigned short* w_t1b;   // [128][7][128]   (1x7: taps along x)
    const unsigned short* w_t1c;   // [128][7][128]   (7x1: taps along y)
    const unsigned short* w_up;    // [896][256]
    const float* b_t0;
    const float* b_t1a;
    const float* b_t1b;
    const float* b_t1c;
    const float* b_up;
    float scale;
    int relu;
    int N;
    const unsigned char* warm;     // optional: bytes the NEXT launch streams (its weight packs), touched by extra workgroups
    long warm_bytes;
};

// Workgroups beyond the images of a fused-block launch (blockIdx.x >= N; the launch adds WARM_WGS of them when asked to) read a
// byte range into the L2 of their XCD and leave.  One image per workgroup uses 180 of the 256 CUs, so they run on idle CUs next to
// the block's own work; workgroup ids go round the eight XCDs, so the j-th extra workgroup sits on XCD (N + j) % 8 and is that
// XCD's (j / 8)-th: it takes eighth j / 8 of the range, every XCD ends up with all of it.  What it buys: the next Block17 streams
// 1.38 MB of weights per workgroup through a dependent chain of k steps; from memory that chain costs 50 us per launch, from a warm
// L2 39 us (measured by pointing all ten blocks at one block's weights).
enum { WARM_WGS = 64 };
__device__ __forceinline__ void warm_range(const unsigned char* p, long bytes, int j, int nthreads) {
    const long per = ((bytes + 8 * 16 - 1) / (8 * 16)) * 16;     // eighths in whole 16-byte chunks
    const long lo = (long)(j >> 3) * per, hi = lo + per < bytes ? lo + per : bytes;
    unsigned acc = 0u;
    for (long o = lo + (long)threadIdx.x * 16; o + 16 <= hi; o += (long)nthreads * 16 * 4) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long oo = o + (long)u * nthreads * 16;
            v[u] = oo + 16 <= hi ? *reinterpret_cast<const u32x4*>(p + oo) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u][0];
    }
    asm volatile("" ::"v"(acc));      // the loads are the effect: keep them
}

__device__ __forceinline__ int swz64(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 2) & 1) << 1)) << 4); }

// Software pipeline shared by the four stages: k tiles kt+1 .. kt+D-1 are in flight in a ring of D register sets (a workgroup is
// alone on its CU -- the LDS images take 130 KB -- so nothing but its own prefetch depth hides the L2 latency of the weight
// stream), two LDS staging buffers, ONE barrier per k tile:
//     iteration kt:  registers of tile kt+1 -> LDS buffer (kt+1)&1 ; issue the loads of tile kt+D ; multiply buffer kt&1 ; barrier
// LOAD(kt, d) / STORE(buf, d) / COMPUTE(kt, buf, d) are statement macros of the stage; d (ring slot) and buf are COMPILE-TIME
// inside the D-unrolled body (D is even and tile loops start at multiples of D), so buffer selection folds into instruction
// offsets.  With one wave per SIMD every instruction issues back to back: all per-lane addresses are computed ONCE per stage
// and the k loop carries no address arithmetic (the first version of this kernel spent ~1000 of its 1400 cycles per k tile on
// it); global operands are buffer loads whose k offset travels in an SGPR.
#define FN_RING_PIPELINE(D_, NT_, LOAD, STORE, COMPUTE)                                   \
    {                                                                                       \
        _Pragma("unroll") for (int d_ = 0; d_ < (D_); ++d_) { LOAD(min(d_, (NT_) - 1), d_) } \
        STORE(0, 0)                                                                         \
        LOAD(min((D_), (NT_) - 1), 0)                                                       \
        __syncthreads();                                                                    \
        for (int kt0_ = 0; kt0_ < (NT_); kt0_ += (D_)) {                                    \
            _Pragma("unroll") for (int d_ = 0; d_ < (D_); ++d_) {                           \
                const int kt_ = kt0_ + d_;                                                  \
                if (kt_ < (NT_)) {                                                          \
                    STORE((d_ + 1) & 1, (d_ + 1) % (D_))                                    \
                    LOAD(min(kt_ + 1 + (D_), (NT_) - 1), (d_ + 1) % (D_))                   \
                    COMPUTE(kt_, d_ & 1, d_)                                                \
                    __syncthreads();                                                        \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
    }

// LDS-DMA form of the weight stream of stages 2-4 (template DMA): the slabs go global -> LDS directly (buffer_load ... lds, 1 KiB per
// wave instruction, the chunk XOR of the LDS layout applied to the per-lane SOURCE address), three LDS buffers, a counted
// s_waitcnt vmcnt(2) + a raw s_barrier per k tile -- no staging registers, no ds_write, the next two tiles stay in flight across
// the barrier (cdna_hip_programming.md section 5, "Pipelining across barriers").  NB buffers: tile kt lives in buffer kt % 3.
#define FN_DMA_PIPELINE(U_, NT_, ISSUE, COMPUTE)                                          \
    {                                                                                       \
        ISSUE(0, 0)                                                                         \
        ISSUE(min(1, (NT_) - 1), 1)                                                         \
        for (int kt0_ = 0; kt0_ < (NT_); kt0_ += (U_)) {                                    \
            _Pragma("unroll") for (int d_ = 0; d_ < (U_); ++d_) {                           \
                const int kt_ = kt0_ + d_;                                                  \
                if (kt_ < (NT_)) {                                                          \
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   /* this wave's share of tile kt has landed */ \
                    __builtin_amdgcn_s_barrier();                      /* everybody's has; buffer (kt + 2) % 3 is free */ \
                    ISSUE(min(kt_ + 2, (NT_) - 1), (d_ + 2) % 3)                            \
                    COMPUTE(kt_, d_ % 3, d_)                                                \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    \
        __builtin_amdgcn_s_barrier();                                                       \
    }

template <typename T, bool DMA>
__global__ __launch_bounds__(512) void block17_infer_kernel(const Block17Args a) {
    constexpr int C = 896, CT = 128, NPIX = 64, D = 4;
    constexpr int DMA_BUF = 16 * 1024;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr int PATCH_PIX = 112;                        // 8 x 14 (t1a) or 14 x 8 (t1b)
    constexpr int MIXED_BYTES = 8 * NPIX * 64;            // 8 slices: t0 (4) | t1c (4)
    constexpr int PATCH_BYTES = 4 * PATCH_PIX * 64;       // 4 slices
    constexpr int STAGE_BYTES = 20 * 1024;                // one staging buffer: stage 1 = trunk slice 4 KB + slab 16 KB, stages 2-4 = slab 16 KB
    typedef typename LP<T>::vec8 vec8;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sMixed = smem;
    unsigned char* sT1a = sMixed + MIXED_BYTES;           // [4 slices][8 x 14 pixels][64 B]
    unsigned char* sT1b = sT1a + PATCH_BYTES;             // [4 slices][14 x 8 pixels][64 B]
    unsigned char* sStage = sT1b + PATCH_BYTES;           // [2][STAGE_BYTES]
    float* sC = reinterpret_cast<float*>(sT1a);           // stage 4: fp32 C tile [64][132] over the (dead) patches

    // 8 waves = 2 per SIMD (a wave alone on its SIMD cannot overlap its LDS / barrier waits with anything): wave = (row half wh,
    // column quarter wq); every stage gives a wave 2 row fragments (32 pixels) x its quarter of the stage's columns
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wq = wave & 3, wh = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;
    const int img = blockIdx.x;
    if (img >= a.N) {         // warm-ahead workgroup (uniform per workgroup: no barrier is skipped by part of one)
        warm_range(a.warm, a.warm_bytes, img - a.N, 512);
        return;
    }
    const unsigned short* xin = a.x + (long)img * NPIX * C;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xin), 0, NPIX * C * 2, 0x00020000);

    // zero the halo patches once (the interiors are overwritten by stages 1 and 2)
    for (int i = tid; i < 2 * PATCH_BYTES / 16; i += 512) reinterpret_cast<u32x4*>(sT1a)[i] = u32x4{0u, 0u, 0u, 0u};

    // ---------------- stage 1: [t0 | t1a] = relu(x[64 x 896] * W[896 x 256] + b), wave w owns columns 64 w .. 64 w + 63 ----------------
    // k tile = 32 input channels: trunk slice [64 pixels][64 B] + slab [256 columns][64 B] (64-byte rows, swz64)
    {
        const int lrow = tid >> 2, lch = tid & 3;          // 128 rows: columns lrow (t0) and lrow (t1a); trunk pixel lrow & 63
        const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w_t0), 0, CT * C * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w_t1a), 0, CT * C * 2, 0x00020000);
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 ra[D], rb[D][2];
        const int voff = (lrow * C + lch * 8) * 2;                  // row lrow of either weight matrix
        const int voff_x = ((lrow & 63) * C + lch * 8) * 2;         // trunk pixel (both thread halves load it, the lower half stores it)
        const int st_a = swz64(lrow & 63, lch);                     // LDS slots of this thread's chunks (buffer 0)
        int st_b[2], fa_off[2], fb_off[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) st_b[q] = 4096 + swz64(lrow + 128 * q, lch);
#pragma unroll
        for (int i = 0; i < 2; ++i) fa_off[i] = swz64(wh * 32 + i * 16 + fr, fq);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb_off[j] = 4096 + swz64(wq * 64 + j * 16 + fr, fq);
#define S1_LOAD(kt, d)                                                                                    \
        ra[d] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff_x, (kt) * 64, 0);                         \
        rb[d][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_w0, voff, (kt) * 64, 0);                       \
        rb[d][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_w1, voff, (kt) * 64, 0);
#define S1_STORE(buf, d)                                                                                   \
        if (wh == 0) *reinterpret_cast<u32x4*>(sStage + (buf) * STAGE_BYTES + st_a) = ra[d];                 \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) *reinterpret_cast<u32x4*>(sStage + (buf) * STAGE_BYTES + st_b[q]) = rb[d][q];
#define S1_COMPUTE(kt, buf, d)                                                                                           \
        {                                                                                                                  \
            vec8 fa[2], fb[4];                                                                                             \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const vec8*>(sStage + (buf) * STAGE_BYTES + fa_off[i]); \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const vec8*>(sStage + (buf) * STAGE_BYTES + fb_off[j]); \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                  \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = LP<T>::mfma(fa[i], fb[j], acc[i][j]);            \
        }
        FN_RING_PIPELINE(D, C / 32, S1_LOAD, S1_STORE, S1_COMPUTE)
#undef S1_LOAD
#undef S1_STORE
#undef S1_COMPUTE
        // epilogue: column quarters 0,1 hold t0 (-> mixed slices 0..3), quarters 2,3 hold t1a (-> the 8 x 14 patch, interior columns 3..10)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = (wq & 1) * 64 + j * 16 + fr;            // channel within t0 / t1a
            const float bias = wq < 2 ? a.b_t0[col] : a.b_t1a[col];
            const int slice = col >> 5, cc = col & 31;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int p = wh * 32 + i * 16 + fq * 4 + r;   // pixel (y, x) = (p >> 3, p & 7)
                    const unsigned short h = LP<T>::from_f32(fmaxf(acc[i][j][r] + bias, 0.f));
                    if (wq < 2) {
                        *reinterpret_cast<unsigned short*>(sMixed + slice * (NPIX * 64) + swz64(p, cc >> 3) + (cc & 7) * 2) = h;
                    } else {
                        const int pidx = (p >> 3) * 14 + (p & 7) + 3;
                        *reinterpret_cast<unsigned short*>(sT1a + slice * (PATCH_PIX * 64) + swz64(pidx, cc >> 3) + (cc & 7) * 2) = h;
                    }
                }
        }
        __syncthreads();
    }

    // Stages 2-4 stream slabs of [128 columns][64 input channels]: 128-byte rows, 16-byte chunk XOR (row & 7) (the implicit-GEMM
    // kernel's B layout); a k tile is two MFMA k steps.  Loader role: row = tid >> 3 (+32 q), chunk = tid & 7.
    const int srow = tid >> 3, sch = tid & 7;        // 64 rows (+64 q)
    int sl_off[2], fb_off[2][2];          // LDS slots of this thread's four slab chunks / of this lane's B fragments [k step][column fragment]
#pragma unroll
    for (int q = 0; q < 2; ++q) sl_off[q] = (srow + 64 * q) * 128 + ((sch ^ ((srow + 64 * q) & 7)) << 4);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rw = wq * 32 + j * 16 + fr;
            fb_off[h][j] = rw * 128 + (((h * 4 + fq) ^ (rw & 7)) << 4);
        }
#define SLAB_STORE(buf, d) \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) *reinterpret_cast<u32x4*>(sStage + (buf) * STAGE_BYTES + sl_off[q]) = rb[d][q];

    // ---------------- stages 2 and 3: 1x7 then 7x1, 128 -> 128, wave w owns columns 32 w .. 32 w + 31 ----------------
#pragma unroll 1
    for (int stage = 2; stage <= 3; ++stage) {
        const unsigned short* wgt = stage == 2 ? a.w_t1b : a.w_t1c;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wgt), 0, CT * 7 * CT * 2, 0x00020000);
        const float* bias_p = stage == 2 ? a.b_t1b : a.b_t1c;
        const unsigned char* src = stage == 2 ? sT1a : sT1b;
        // tap t reads patch pixel base + t * tstep: 1x7 walks along x in the 8 x 14 patch, 7x1 along y in the 14 x 8 patch
        const int tstep = stage == 2 ? 1 : 8;
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int fa_off[7][2];      // LDS offset (slice 0) of this lane's A fragment of row fragment i at tap t: 14 registers, no arithmetic in the loop
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = wh * 32 + i * 16 + fr;
                const int base = stage == 2 ? (p >> 3) * 14 + (p & 7) : p;      // 7x1: pixel (y, x) at tap t is row y + t of the 14 x 8 patch
                fa_off[t][i] = swz64(base + t * tstep, fq);
            }
        u32x4 rb[D][2];
        const int voff = (srow * (7 * CT) + sch * 8) * 2;
        // k tile kt = (tap, half): input channels 64 * half .. + 63 of tap kt >> 1.  The tile loop is unrolled by D = 4 from
        // multiples of 4, so tap = 2 * (kt0 / 4 ... ) is only known at run time: the fragment offsets are selected with a
        // uniform switch on the tap PAIR (kt >> 2 picks taps 2g, 2g+1; d picks which), all register indices stay static.
#define S23_LOAD(kt, d)                                                                                                         \
        _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                            \
            rb[d][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff + 64 * q * (7 * CT) * 2, (((kt) >> 1) * CT + ((kt) & 1) * 64) * 2, 0);
#define S23_TAP(TAP, buf, half)                                                                                                  \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                          \
            vec8 fa[2], fb[2];                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                        \
                fa[i] = *reinterpret_cast<const vec8*>(src + ((half) * 2 + h) * (PATCH_PIX * 64) + fa_off[TAP][i]);              \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const vec8*>(sStage + (buf) * STAGE_BYTES + fb_off[h][j]); \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                        \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = LP<T>::mfma(fa[i], fb[j], acc[i][j]);                  \
        }
#define S23_COMPUTE(kt, buf, d)                                                                                                  \
        {                                                                                                                        \
            const int g = (kt) >> 2;               /* uniform; kt = 4 g + d: tap = 2 g + (d >> 1), half = d & 1 */              \
            if ((d) < 2) {                                                                                                       \
                if (g == 0) { S23_TAP(0, buf, (d) & 1) } else if (g == 1) { S23_TAP(2, buf, (d) & 1) }                           \
                else if (g == 2) { S23_TAP(4, buf, (d) & 1) } else { S23_TAP(6, buf, (d) & 1) }                                  \
            } else {                                                                                                             \
                if (g == 0) { S23_TAP(1, buf, (d) & 1) } else if (g == 1) { S23_TAP(3, buf, (d) & 1) } else { S23_TAP(5, buf, (d) & 1) } \
            }                                                                                                                    \
        }
        if constexpr (DMA) {
            // unrolled by 6 from multiples of 6: buffer = d % 3, half = d & 1, tap = 3 (kt0 / 6) + (d >> 1)
#define S23_DMA_ISSUE(kt, buf)                                                                                               \
            _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                        \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(sStage + (buf) * DMA_BUF + (wave * 16 + 8 * q) * 128), 16,   \
                                                         dma_voff23[q], (((kt) >> 1) * CT + ((kt) & 1) * 64) * 2, 0, 0);
#define S23_DMA_FB(buf, h, j) *reinterpret_cast<const vec8*>(sStage + (buf) * DMA_BUF + fb_off[h][j])
#define S23_DMA_TAP(TAP, buf, half)                                                                                              \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                      \
                vec8 fa[2], fb[2];                                                                                               \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                    \
                    fa[i] = *reinterpret_cast<const vec8*>(src + ((half) * 2 + h) * (PATCH_PIX * 64) + fa_off[TAP][i]);          \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) fb[j] = S23_DMA_FB(buf, h, j);                                     \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                    \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = LP<T>::mfma(fa[i], fb[j], acc[i][j]);              \
            }
#define S23_DMA_COMPUTE(kt, buf, d)                                                                                              \
            {                                                                                                                    \
                const int g = (kt) / 6;                /* uniform */                                                             \
                if (((d) >> 1) == 0) { if (g == 0) { S23_DMA_TAP(0, buf, (d) & 1) } else if (g == 1) { S23_DMA_TAP(3, buf, (d) & 1) } else { S23_DMA_TAP(6, buf, (d) & 1) } } \
                else if (((d) >> 1) == 1) { if (g == 0) { S23_DMA_TAP(1, buf, (d) & 1) } else { S23_DMA_TAP(4, buf, (d) & 1) } } \
                else { if (g == 0) { S23_DMA_TAP(2, buf, (d) & 1) } else { S23_DMA_TAP(5, buf, (d) & 1) } }                      \
            }
            int dma_voff23[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = wave * 16 + 8 * q + (lane >> 3), cs = (lane & 7) ^ (r & 7);
                dma_voff23[q] = (r * (7 * CT) + cs * 8) * 2;
            }
            FN_DMA_PIPELINE(6, 14, S23_DMA_ISSUE, S23_DMA_COMPUTE)
#undef S23_DMA_ISSUE
#undef S23_DMA_FB
#undef S23_DMA_TAP
#undef S23_DMA_COMPUTE
        } else {
            FN_RING_PIPELINE(D, 14, S23_LOAD, SLAB_STORE, S23_COMPUTE)
        }
#undef S23_LOAD
#undef S23_TAP
#undef S23_COMPUTE
        // 1x7 output -> 14 x 8 patch (rows 3..10); 7x1 output -> mixed slices 4..7
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wq * 32 + j * 16 + fr;
            const float bias = bias_p[col];
            const int slice = col >> 5, cc = col & 31;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int p = wh * 32 + i * 16 + fq * 4 + r;
                    const unsigned short h = LP<T>::from_f32(fmaxf(acc[i][j][r] + bias, 0.f));
                    if (stage == 2) *reinterpret_cast<unsigned short*>(sT1b + slice * (PATCH_PIX * 64) + swz64(p + 24, cc >> 3) + (cc & 7) * 2) = h;
                    else *reinterpret_cast<unsigned short*>(sMixed + (4 + slice) * (NPIX * 64) + swz64(p, cc >> 3) + (cc & 7) * 2) = h;
                }
        }
        __syncthreads();
    }

    // ---------------- stage 4: out = act(x + scale * (mixed[64 x 256] * Wup[256 x 896] + bias)), 7 passes of 128 columns ----------------
    // ONE pipeline over the 7 x 4 k tiles (the weight stream does not restart per pass); the pass epilogue runs inside the tile loop
    {
        constexpr int CLD = 132;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w_up), 0, C * 256 * 2, 0x00020000);
        u32x4 rb[D][2];
        const int voff = (srow * 256 + sch * 8) * 2;
        int fa_off[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa_off[i] = swz64(wh * 32 + i * 16 + fr, fq);
        f32x4 acc[2][2];
        // k tile kt = (pass, kk): columns 128 pass .. + 127, input channels 64 kk .. + 63 of mixed (slices 2 kk, 2 kk + 1); kk == d
#define S4_LOAD(kt, d)                                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                            \
            rb[d][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff + 64 * q * 256 * 2, ((((kt) >> 2) * 128) * 256 + ((kt) & 3) * 64) * 2, 0);
#define S4_COMPUTE(kt, buf, d)                                                                                                   \
        {                                                                                                                        \
            if ((d) == 0) {                                                                                                      \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                    \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};                         \
            }                                                                                                                    \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                      \
                vec8 fa[2], fb[2];                                                                                               \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                    \
                    fa[i] = *reinterpret_cast<const vec8*>(sMixed + ((d) * 2 + h) * (NPIX * 64) + fa_off[i]);                    \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const vec8*>(sStage + (buf) * STAGE_BYTES + fb_off[h][j]); \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                    \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = LP<T>::mfma(fa[i], fb[j], acc[i][j]);              \
            }                                                                                                                    \
            if ((d) == 3) {   /* pass complete: fp32 C tile (over the dead patches) -> coalesced residual epilogue */              \
                const int pass = (kt) >> 2;                                                                                      \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                    \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
                        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                            \
                            sC[(wh * 32 + i * 16 + fq * 4 + r) * CLD + wq * 32 + j * 16 + fr] = acc[i][j][r];                            \
                __syncthreads();                                                                                                 \
                _Pragma("unroll") for (int q = 0; q < 2; ++q) {   /* 64 pixels x 16 column groups of 8 = 1024 chunks */           \
                    const int idx = tid + 512 * q, p = idx >> 4, cg = idx & 15;                                                  \
                    const int col = pass * 128 + cg * 8;                                                                         \
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(&sC[p * CLD + cg * 8]);                                     \
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(&sC[p * CLD + cg * 8 + 4]);                                 \
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.b_up + col), b1 = *reinterpret_cast<const f32x4*>(a.b_up + col + 4); \
                    float rv[8], v[8];                                                                                           \
                    unpack8<T>(load_global_b128(xin, (long)p * C + col), rv);                                                    \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                              \
                        v[e] = rv[e] + a.scale * (c0[e] + b0[e]);                                                                \
                        v[4 + e] = rv[4 + e] + a.scale * (c1[e] + b1[e]);                                                        \
                    }                                                                                                            \
                    if (a.relu) {                                                                                                \
                        _Pragma("unroll") for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);                                   \
                    }                                                                                                            \
                    *reinterpret_cast<u32x4*>(a.y + ((long)img * NPIX + p) * C + col) = pack8<T>(v);                             \
                }                                                                                                                \
            }                                                                                                                    \
        }
        if constexpr (DMA) {
            // unrolled by 12 from multiples of 12: buffer = d % 3, kk = d & 3
            int dma_voff4[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = wave * 16 + 8 * q + (lane >> 3), cs = (lane & 7) ^ (r & 7);
                dma_voff4[q] = (r * 256 + cs * 8) * 2;
            }
#define S4_DMA_ISSUE(kt, buf)                                                                                                    \
            _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                        \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(sStage + (buf) * DMA_BUF + (wave * 16 + 8 * q) * 128), 16,   \
                                                         dma_voff4[q], ((((kt) >> 2) * 128) * 256 + ((kt) & 3) * 64) * 2, 0, 0);
#define S4_DMA_COMPUTE(kt, buf, d)                                                                                               \
            {                                                                                                                    \
                if (((d) & 3) == 0) {                                                                                            \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};                     \
                }                                                                                                                \
                _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                  \
                    vec8 fa[2], fb[2];                                                                                           \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
                        fa[i] = *reinterpret_cast<const vec8*>(sMixed + (((d) & 3) * 2 + h) * (NPIX * 64) + fa_off[i]);          \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const vec8*>(sStage + (buf) * DMA_BUF + fb_off[h][j]); \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = LP<T>::mfma(fa[i], fb[j], acc[i][j]);          \
                }                                                                                                                \
                if (((d) & 3) == 3) {                                                                                            \
                    const int pass = (kt) >> 2;                                                                                  \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
                            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                        \
                                sC[(wh * 32 + i * 16 + fq * 4 + r) * CLD + wq * 32 + j * 16 + fr] = acc[i][j][r];                \
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
                    __builtin_amdgcn_s_barrier();                                                                                \
                    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                              \
                        const int idx = tid + 512 * q, p = idx >> 4, cg = idx & 15;                                              \
                        const int col = pass * 128 + cg * 8;                                                                     \
                        const f32x4 c0 = *reinterpret_cast<const f32x4*>(&sC[p * CLD + cg * 8]);                                 \
                        const f32x4 c1 = *reinterpret_cast<const f32x4*>(&sC[p * CLD + cg * 8 + 4]);                             \
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.b_up + col), b1 = *reinterpret_cast<const f32x4*>(a.b_up + col + 4); \
                        float rv[8], v[8];                                                                                       \
                        unpack8<T>(load_global_b128(xin, (long)p * C + col), rv);                                                \
                        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                          \
                            v[e] = rv[e] + a.scale * (c0[e] + b0[e]);                                                            \
                            v[4 + e] = rv[4 + e] + a.scale * (c1[e] + b1[e]);                                                    \
                        }                                                                                                        \
                        if (a.relu) {                                                                                            \
                            _Pragma("unroll") for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);                               \
                        }                                                                                                        \
                        *reinterpret_cast<u32x4*>(a.y + ((long)img * NPIX + p) * C + col) = pack8<T>(v);                         \
                    }                                                                                                            \
                }                                                                                                                \
            }
            FN_DMA_PIPELINE(12, 28, S4_DMA_ISSUE, S4_DMA_COMPUTE)
#undef S4_DMA_ISSUE
#undef S4_DMA_COMPUTE
        } else {
            FN_RING_PIPELINE(D, 28, S4_LOAD, SLAB_STORE, S4_COMPUTE)
        }
#undef S4_LOAD
#undef S4_COMPUTE
    }
#undef SLAB_STORE
}

// ------------------------------------------------------------------------------------------------------------------------
// Fused Inception-ResNet-A block ("Block35", inception_resnet_v1.py:83-150), BN-folded inference / mining forward:
//     t0 = relu(1x1(x)), t1 = relu(3x3(relu(1x1(x)))), t2 = relu(3x3(relu(3x3(relu(1x1(x))))))        (256 -> 32 each, 32 -> 32)
//     out = act(x + scale * (1x1(concat(t0, t1, t2), 96 -> 256) + bias))
// The layer-by-layer plan moves the 17 x 17 x 256 trunk through HBM five times per block (three 1x1 readers, the residual, the
// output) for 44 MFLOP per image; here the trunk is read twice (stage 1 stream, residual) and written once, everything else
// stays in LDS.  One workgroup (8 waves) per image: 289 pixels = 19 row fragments, wave w owns fragments w, w+8, w+16 and ALL
// columns of a stage (the stages are only 32 .. 96 columns wide).  Activations: 64-byte pixel rows per 32-channel slice
// (swz64); the 3x3 inputs are kept as 19 x 19 patches with a 1-pixel zero halo, so the nine taps are nine shifted views.
// The weights of a stage (<= 48 KB) are staged in LDS in one piece; only the trunk of stage 1 streams (ring pipeline).
// ------------------------------------------------------------------------------------------------------------------------
struct Block35Args {
    const unsigned short* x;       // [N, 17, 17, 256]
    unsigned short* y;
    const unsigned short* w_1x1[3];   // tower_conv0/1x1, tower_conv1/0a, tower_conv2/0a: [32][256]
    const unsigned short* w_3x3[3];   // tower_conv1/0b, tower_conv2/0b, tower_conv2/0c: [32][9][32]
    const unsigned short* w_up;       // [256][96]
    const float* b_1x1[3];
    const float* b_3x3[3];
    const float* b_up;
    float scale;
    int relu;
    int N;
    const unsigned char* warm;     // optional warm-ahead range (see Block17Args / warm_range)
    long warm_bytes;
};

template <typename T>
__global__ __launch_bounds__(512) void block35_infer_kernel(const Block35Args a) {
    constexpr int C = 256, NPIX = 289, MROWS = 304, D = 4;          // 19 row fragments
    constexpr int SLICE_BYTES = MROWS * 64;                          // one 32-channel slice of the image (padded rows)
    constexpr int PATCH_PIX = 19 * 19, PATCH_BYTES = (PATCH_PIX * 64 + 1023) / 1024 * 1024 + 1024;   // reads of padded rows stay inside
    constexpr int MIXED_BYTES = 3 * SLICE_BYTES;                     // t0 | t1 | t2
    constexpr int STAGE_BYTES = SLICE_BYTES + 96 * 64;               // stage 1: trunk slice + slab [96 columns][64 B]
    typedef typename LP<T>::vec8 vec8;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sMixed = smem;
    unsigned char* sP1 = sMixed + MIXED_BYTES;
    unsigned char* sP2 = sP1 + PATCH_BYTES;
    unsigned char* sStage = sP2 + PATCH_BYTES;            // [2][STAGE_BYTES]; later: the weights of a stage in one piece
    float* sC = reinterpret_cast<float*>(sP1);            // stage 4: fp32 C tile [304][36] over the (dead) patches

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int img = blockIdx.x;
    if (img >= a.N) {         // warm-ahead workgroup
        warm_range(a.warm, a.warm_bytes, img - a.N, 512);
        return;
    }
    const unsigned short* xin = a.x + (long)img * NPIX * C;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xin), 0, NPIX * C * 2, 0x00020000);
    constexpr unsigned OOB = 0x60000000u;

    for (int i = tid; i < 2 * PATCH_BYTES / 16; i += 512) reinterpret_cast<u32x4*>(sP1)[i] = u32x4{0u, 0u, 0u, 0u};

    // row fragments of this wave: f = wave + 8 i, i < 3, valid while f < 19 (wave-uniform)
    const int nfr = wave < 3 ? 3 : 2;
    int prow[3];          // first pixel of each fragment's lane (row = lane & 15)
#pragma unroll
    for (int i = 0; i < 3; ++i) prow[i] = (wave + 8 * i) * 16 + fr;

    // cooperative copy of a [rows][32 channels] weight block (64-byte rows in global, row stride `gstride` elements) into LDS
    auto stage_weights = [&](unsigned char* dst, const unsigned short* w, int rows, int gstride, int goff) {
        for (int i = tid; i < rows * 4; i += 512) {
            const int r = i >> 2, ch = i & 3;
            *reinterpret_cast<u32x4*>(dst + swz64(r, ch)) = load_global_b128(w, (long)r * gstride + goff + ch * 8);
        }
    };
    // relu(acc + bias) -> LDS image: pixel p at index pidx(p) of `region` (slice base), channel c (0..31); C layout of the 16x16 MFMA
    auto store_frag = [&](const f32x4& v, float bias, unsigned char* region, int frag, int col, bool patch) {
        const int cc = col & 31;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = frag * 16 + fq * 4 + r;
            if (p < NPIX) {
                const int y = p / 17, x = p - y * 17;
                const int pidx = patch ? (y + 1) * 19 + x + 1 : p;
                *reinterpret_cast<unsigned short*>(region + swz64(pidx, cc >> 3) + (cc & 7) * 2) = LP<T>::from_f32(fmaxf(v[r] + bias, 0.f));
            }
        }
    };

    // ---------------- stage 1: [t0 | t1a | t2a] = relu(x[289 x 256] * W[256 x 96] + b): trunk slices stream, 8 k tiles of 32 channels ----------------
    {
        f32x4 acc[3][6];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 ra[D][3], rb[D];
        int voff_a[3], st_a[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = tid + 512 * q, p = idx >> 2, ch = idx & 3;
            voff_a[q] = p < NPIX ? (p * C + ch * 8) * 2 : (int)OOB;
            st_a[q] = p < MROWS ? swz64(p, ch) : -1;
        }
        const int wr = tid >> 2, wch = tid & 3;                     // slab row (column of the stage) 0..95 for tid < 384
        const unsigned short* wrow = a.w_1x1[min(wr >> 5, 2)] + (long)(wr & 31) * C + wch * 8;
        const int st_b = wr < 96 ? SLICE_BYTES + swz64(wr, wch) : -1;
#define S1_LOAD(kt, d)                                                                                     \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) ra[d][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff_a[q], (kt) * 64, 0); \
        rb[d] = load_global_b128(wrow, (kt) * 32);
#define S1_STORE(buf, d)                                                                                   \
        _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                        \
            if (st_a[q] >= 0) *reinterpret_cast<u32x4*>(sStage + (buf) * STAGE_BYTES + st_a[q]) = ra[d][q];  \
        if (st_b >= 0) *reinterpret_cast<u32x4*>(sStage + (buf) * STAGE_BYTES + st_b) = rb[d];
#define S1_COMPUTE(kt, buf, d)                                                                                           \
        {                                                                                                                  \
            const unsigned char* pa = sStage + (buf) * STAGE_BYTES;                                                        \
            vec8 fb[6];                                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 6; ++j) fb[j] = *reinterpret_cast<const vec8*>(pa + SLICE_BYTES + swz64(j * 16 + fr, fq)); \
            _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                                  \
                if (i < nfr) {                                                                                             \
                    const vec8 fa = *reinterpret_cast<const vec8*>(pa + swz64(prow[i], fq));                               \
                    _Pragma("unroll") for (int j = 0; j < 6; ++j) acc[i][j] = LP<T>::mfma(fa, fb[j], acc[i][j]);           \
                }                                                                                                          \
        }
        FN_RING_PIPELINE(D, C / 32, S1_LOAD, S1_STORE, S1_COMPUTE)
#undef S1_LOAD
#undef S1_STORE
#undef S1_COMPUTE
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int col = j * 16 + fr, t = col >> 5;               // tower 0 -> mixed slice 0, towers 1, 2 -> patches P1, P2
            const float bias = a.b_1x1[t][col & 31];
            unsigned char* region = t == 0 ? sMixed : (t == 1 ? sP1 : sP2);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < nfr) store_frag(acc[i][j], bias, region, wave + 8 * i, col, t != 0);
        }
        __syncthreads();
    }

    // ---------------- stages 2a, 2b, 3: 3x3 32 -> 32 from a 19 x 19 patch; weights [32][9][32] staged whole (18 KB) ----------------
    int fa_tap[9][3];          // LDS offset of this lane's A fragment i at tap t
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int p = min(prow[i], NPIX - 1);                    // padding rows read a real pixel; their results are dropped
            const int y = p / 17, x = p - y * 17;
            fa_tap[t][i] = swz64((y + t / 3) * 19 + x + t % 3, fq);
        }
#pragma unroll 1
    for (int c3 = 0; c3 < 3; ++c3) {
        // c3 = 0: t1b = 3x3(P1) -> mixed slice 1; c3 = 1: t2b = 3x3(P2) -> P1 (its reader is done); c3 = 2: t2c = 3x3(P1) -> mixed slice 2
        const unsigned char* src = c3 == 1 ? sP2 : sP1;
        for (int i = tid; i < 9 * 32 * 4; i += 512) {                // slab rows = tap * 32 + column
            const int r = i >> 2, ch = i & 3, tap = r >> 5, co = r & 31;
            *reinterpret_cast<u32x4*>(sStage + swz64(r, ch)) = load_global_b128(a.w_3x3[c3], (long)co * 288 + tap * 32 + ch * 8);
        }
        __syncthreads();
        f32x4 acc[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            vec8 fb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const vec8*>(sStage + swz64(t * 32 + j * 16 + fr, fq));
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < nfr) {
                    const vec8 fa = *reinterpret_cast<const vec8*>(src + fa_tap[t][i]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = LP<T>::mfma(fa, fb[j], acc[i][j]);
                }
        }
        __syncthreads();                                             // everybody is done reading src before P1 is overwritten (c3 = 1)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 16 + fr;
            const float bias = a.b_3x3[c3][col];
            unsigned char* region = c3 == 0 ? sMixed + SLICE_BYTES : (c3 == 1 ? sP1 : sMixed + 2 * SLICE_BYTES);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < nfr) store_frag(acc[i][j], bias, region, wave + 8 * i, col, c3 == 1);
        }
        __syncthreads();
    }

    // ---------------- stage 4: out = act(x + scale * (mixed[289 x 96] * Wup[96 x 256] + bias)); weights staged whole (48 KB), 8 passes of 32 columns ----------------
    {
        constexpr int CLD = 36;
        for (int i = tid; i < 3 * 256 * 4; i += 512) {               // slab rows = slice * 256 + column
            const int r = i >> 2, ch = i & 3, sl = r >> 8, co = r & 255;
            *reinterpret_cast<u32x4*>(sStage + swz64(r, ch)) = load_global_b128(a.w_up, (long)co * 96 + sl * 32 + ch * 8);
        }
        __syncthreads();
#pragma unroll 1
        for (int pass = 0; pass < 8; ++pass) {
            f32x4 acc[3][2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
                vec8 fb[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const vec8*>(sStage + swz64(sl * 256 + pass * 32 + j * 16 + fr, fq));
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (i < nfr) {
                        const vec8 fa = *reinterpret_cast<const vec8*>(sMixed + sl * SLICE_BYTES + swz64(prow[i], fq));
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = LP<T>::mfma(fa, fb[j], acc[i][j]);
                    }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < nfr) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sC[((wave + 8 * i) * 16 + fq * 4 + r) * CLD + j * 16 + fr] = acc[i][j][r];
                }
            __syncthreads();
            for (int idx = tid; idx < NPIX * 4; idx += 512) {        // 289 pixels x 4 column groups of 8
                const int p = idx >> 2, cg = idx & 3;
                const int col = pass * 32 + cg * 8;
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(&sC[p * CLD + cg * 8]);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(&sC[p * CLD + cg * 8 + 4]);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.b_up + col), b1 = *reinterpret_cast<const f32x4*>(a.b_up + col + 4);
                float rv[8], v[8];
                unpack8<T>(load_global_b128(xin, (long)p * C + col), rv);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = rv[e] + a.scale * (c0[e] + b0[e]);
                    v[4 + e] = rv[4 + e] + a.scale * (c1[e] + b1[e]);
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *reinterpret_cast<u32x4*>(a.y + ((long)img * NPIX + p) * C + col) = pack8<T>(v);
            }
            __syncthreads();
        }
    }
}

}  // namespace fn

using namespace fn;

// One Block17 of the BN-folded inference network in one launch (see the kernel).  Weights are the inference packs of the five
// layers ([Cout][taps][Cin], BN folded), biases the folded BN shifts of the four tower layers and the `up` bias.
extern "C" int fn_block17_infer(const void* x, void* y, int N, const void* w_t0, const void* w_t1a, const void* w_t1b, const void* w_t1c,
                                const void* w_up, const float* b_t0, const float* b_t1a, const float* b_t1b, const float* b_t1c, const float* b_up,
                                float scale, int relu, int dtype, void* stream) {
    return fn_block17_infer_warm(x, y, N, w_t0, w_t1a, w_t1b, w_t1c, w_up, b_t0, b_t1a, b_t1b, b_t1c, b_up, scale, relu, nullptr, 0, dtype, stream);
}

// The same launch plus WARM_WGS workgroups that read [warm, warm + warm_bytes) -- what the NEXT launch will stream, normally the
// weight packs of the next block -- into every XCD's L2 (see warm_range).  warm == nullptr: exactly fn_block17_infer.
extern "C" int fn_block17_infer_warm(const void* x, void* y, int N, const void* w_t0, const void* w_t1a, const void* w_t1b, const void* w_t1c,
                                     const void* w_up, const float* b_t0, const float* b_t1a, const float* b_t1b, const float* b_t1c,
                                     const float* b_up, float scale, int relu, const void* warm, int64_t warm_bytes, int dtype, void* stream) {
    FN_REQUIRE(dtype == FN_BF16 || dtype == FN_F16, "dtype %d unsupported", dtype);
    FN_REQUIRE((warm == nullptr) == (warm_bytes == 0) && warm_bytes >= 0 && warm_bytes < (1L << 31) && ((uintptr_t)warm & 15) == 0,
               "block17_infer: warm range must be 16-byte aligned, below 2 GiB, and given with its size");
    FN_REQUIRE(x && y && x != y && N > 0 && w_t0 && w_t1a && w_t1b && w_t1c && w_up && b_t0 && b_t1a && b_t1b && b_t1c && b_up,
               "block17_infer: bad arguments");
    Block17Args a{(const unsigned short*)x, (unsigned short*)y, (const unsigned short*)w_t0, (const unsigned short*)w_t1a,
                  (const unsigned short*)w_t1b, (const unsigned short*)w_t1c, (const unsigned short*)w_up, b_t0, b_t1a, b_t1b, b_t1c, b_up,
                  scale, relu, N, (const unsigned char*)warm, (long)warm_bytes};
    const int grid = N + (warm ? WARM_WGS : 0);
    constexpr size_t smem = 8 * 64 * 64 + 2 * 4 * 112 * 64 + 3 * 16 * 1024;      // staging: 2 x 20 KB (register ring) or 3 x 16 KB (LDS-DMA)
    static const int use_dma = getenv("FN_B17_DMA") ? atoi(getenv("FN_B17_DMA")) : 1;   // measured: stages 2+3 12.4 -> 9.6 us per block (tools/dev_block17.py)
    static LdsOptIn ok[4];
    const void* kerns[4] = {reinterpret_cast<const void*>(block17_infer_kernel<__bf16, false>), reinterpret_cast<const void*>(block17_infer_kernel<_Float16, false>),
                            reinterpret_cast<const void*>(block17_infer_kernel<__bf16, true>), reinterpret_cast<const void*>(block17_infer_kernel<_Float16, true>)};
    const int which = (use_dma ? 2 : 0) + (dtype == FN_BF16 ? 0 : 1);
    if (int rc = allow_big_lds(kerns[which], ok[which], "block17_infer")) return rc;
    if (use_dma) {
        if (dtype == FN_BF16) hipLaunchKernelGGL((block17_infer_kernel<__bf16, true>), dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((block17_infer_kernel<_Float16, true>), dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
    } else {
        if (dtype == FN_BF16) hipLaunchKernelGGL((block17_infer_kernel<__bf16, false>), dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((block17_infer_kernel<_Float16, false>), dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
    }
    return check_launch("block17_infer");
}

// One Block35 of the BN-folded inference network in one launch (see the kernel).  w_1x1 / b_1x1: the three tower-entry 1x1 layers
// (tower_conv0/Conv2d_1x1, tower_conv1/Conv2d_0a_1x1, tower_conv2/Conv2d_0a_1x1); w_3x3 / b_3x3: tower_conv1/Conv2d_0b_3x3,
// tower_conv2/Conv2d_0b_3x3, tower_conv2/Conv2d_0c_3x3; all inference packs [Cout][taps][Cin] with the folded BN shifts.
extern "C" int fn_block35_infer(const void* x, void* y, int N, const void* const* w_1x1, const void* const* w_3x3, const void* w_up,
                                const float* const* b_1x1, const float* const* b_3x3, const float* b_up, float scale, int relu, int dtype,
                                void* stream) {
    return fn_block35_infer_warm(x, y, N, w_1x1, w_3x3, w_up, b_1x1, b_3x3, b_up, scale, relu, nullptr, 0, dtype, stream);
}

// fn_block35_infer plus WARM_WGS workgroups that read [warm, warm + warm_bytes) into every XCD's L2 (see fn_block17_infer_warm).
extern "C" int fn_block35_infer_warm(const void* x, void* y, int N, const void* const* w_1x1, const void* const* w_3x3, const void* w_up,
                                     const float* const* b_1x1, const float* const* b_3x3, const float* b_up, float scale, int relu,
                                     const void* warm, int64_t warm_bytes, int dtype, void* stream) {
    FN_REQUIRE(dtype == FN_BF16 || dtype == FN_F16, "dtype %d unsupported", dtype);
    FN_REQUIRE((warm == nullptr) == (warm_bytes == 0) && warm_bytes >= 0 && warm_bytes < (1L << 31) && ((uintptr_t)warm & 15) == 0,
               "block35_infer: warm range must be 16-byte aligned, below 2 GiB, and given with its size");
    FN_REQUIRE(x && y && x != y && N > 0 && w_1x1 && w_3x3 && w_up && b_1x1 && b_3x3 && b_up, "block35_infer: bad arguments");
    Block35Args a{};
    a.x = (const unsigned short*)x; a.y = (unsigned short*)y; a.w_up = (const unsigned short*)w_up; a.b_up = b_up;
    for (int i = 0; i < 3; ++i) {
        FN_REQUIRE(w_1x1[i] && w_3x3[i] && b_1x1[i] && b_3x3[i], "block35_infer: null layer %d", i);
        a.w_1x1[i] = (const unsigned short*)w_1x1[i]; a.w_3x3[i] = (const unsigned short*)w_3x3[i];
        a.b_1x1[i] = b_1x1[i]; a.b_3x3[i] = b_3x3[i];
    }
    a.scale = scale; a.relu = relu; a.N = N;
    a.warm = (const unsigned char*)warm; a.warm_bytes = (long)warm_bytes;
    const int grid = N + (warm ? WARM_WGS : 0);
    constexpr int SLICE = 304 * 64, PATCH = (19 * 19 * 64 + 1023) / 1024 * 1024 + 1024;
    constexpr size_t smem = 3 * SLICE + 2 * PATCH + 2 * (SLICE + 96 * 64);
    static_assert(smem <= 160 * 1024 && 3 * 256 * 64 <= 2 * (SLICE + 96 * 64) && 304 * 36 * 4 <= 2 * PATCH, "block35 LDS plan");
    static LdsOptIn ok[2];
    const int which = dtype == FN_BF16 ? 0 : 1;
    if (int rc = allow_big_lds(which == 0 ? reinterpret_cast<const void*>(block35_infer_kernel<__bf16>) : reinterpret_cast<const void*>(block35_infer_kernel<_Float16>),
                               ok[which], "block35_infer")) return rc;
    if (dtype == FN_BF16) hipLaunchKernelGGL(block35_infer_kernel<__bf16>, dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(block35_infer_kernel<_Float16>, dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
    return check_launch("block35_infer");
}
