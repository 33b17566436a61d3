// Tap-sharing weight gradient for the k x k layers of Inception-ResNet-v1 (3x3 stride 1 / 2, 1x7, 7x1) on gfx950.
//
// dW[co][ky][kx][ci] = sum over output pixels  dY[n, oy, ox, co] * X[n, oy*s - pad + ky, ox*s - pad + kx, ci]
// (the gradient Keras computes for every Conv2D declared at facenet/models/inception_resnet_v1.py:90-430).
//
// conv_wgrad_kernel (conv_igemm.hip) treats the KH*KW*Cin columns of dW as one flat GEMM dimension: a workgroup owns 64
// couts x 64 columns and fetches, for every 64 pixels, a dY tile and an X tile gathered at ONE tap -- so X crosses the
// L2 -> CU path once per tap and dY once per column tile (nine and up to 27 times for a 3x3 layer; 128 B of operands per
// MFMA clock against the ~29 B/clk a CU draws from L2, MI355X_MICROARCH.md "Indexed rows").  Here a workgroup owns a cout
// tile x a 32-channel slice of Cin x ALL taps.  Per stage it loads one 2-D tile of <= 64 output pixels of dY and the
// source PATCH those pixels touch ((TH-1)s+KH) x ((TW-1)s+KW) pixels, once; the B fragment of tap (ky,kx) is a shifted
// view of the patch in LDS.  Both operands are pixel-major in memory, so fragments come through ds_read_b64_tr_b16
// (hardware transpose) exactly as in conv_wgrad_kernel; patch pixels are padded to 96 B (stride 1) / 80 B (stride 2)
// so that the eight pixel rows a 32-lane half reads are conflict free.  26 B of operands per MFMA clock.
//
// The result is stored, never atomically added: a workgroup whose layer is not split over pixels writes dW itself, split
// z of a split layer writes slab z and wgrad_reduce_kernel adds the slabs in order (deterministic; see wgrad_taps.h).
#include "wgrad_taps.h"
#include <cstdio>
#include <cstdlib>

namespace fn {

struct WgradTapArgs {
    WgradOut out;   // first member (wgrad_reduce_kernel)
    const unsigned short* x;
    const unsigned short* dy;
    int H, W, OH, OW, Cin, stride, pad_h, pad_w, ld_x, ld_y;
    int TH, TW, PWt, npix, RS;       // tile rows / columns, patch width, patch pixels, LDS bytes per patch pixel
    int pbytes;                      // bytes of one patch buffer (16-byte multiple)
    int tiles_x, tiles_img, ntiles;  // tiles per image row, per image, in all
    int chunk;                       // tiles per pixel split
    int gx, gy;                      // Cin slices, cout tiles
    int ntaps;                       // KH*KW (<= 9)
    int wpair;                       // patch store: lanes 4..7 of an 8-lane group hold the pixel `wpair` after the one in lanes 0..3
    int tapoff[9];                   // LDS byte offset of tap t relative to a tile pixel's own patch position
    int x_bytes, dy_bytes;
    float inv_img, inv_tx, inv_tw, inv_pw;
};

template <typename T, int BMW, int TAPS, int NPL>
__device__ __forceinline__ void conv_wgrad_taps_body(const WgradTapArgs& a, const int bx, const int by, const int bz) {
    constexpr int CI = 32;                       // input channels per workgroup
    constexpr int RSA = BMW * 2 + 32;            // dY tile: bytes per pixel row (160 / 96: conflict-free transposed reads)
    constexpr int A_BYTES = 64 * RSA;
    constexpr int CGA = BMW / 8;                 // 16-byte chunks per dY row
    constexpr int AP = 64 * CGA / 256;           // dY chunks per thread (2 / 1)
    constexpr int TM = BMW / 2;                  // waves: 2 (cout) x 2 (16 input channels each)
    constexpr int MREP = TM / 16;
    typedef typename LP<T>::vec8 vec8;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                    // [2][64][RSA]   dY tile, rows = tile pixels
    unsigned char* sP = smem + 2 * A_BYTES;      // [2][npix][RS]  source patch, rows = patch pixels, 64 B of channels each

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ci0 = bx * CI, co0 = by * BMW;
    const int tbeg = bz * a.chunk, tend = min(a.ntiles, tbeg + a.chunk);
    const int nst = tend - tbeg;
    if (nst <= 0) return;

    constexpr unsigned OOB = 0x60000000u;
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.dy), 0, a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, a.x_bytes, 0x00020000);

    // ---- per-thread load slots (fixed for the whole kernel) ----
    int a_ty[AP], a_tx[AP], a_rel[AP], a_lds[AP];
    bool a_ok[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int c = tid + 256 * i, p = c / CGA, cg = c - p * CGA;
        int ty, tx;
        fast_divmod(p, a.TW, a.inv_tw, ty, tx);
        a_ty[i] = ty;
        a_tx[i] = tx;
        a_ok[i] = ty < a.TH && co0 + cg * 8 < a.out.Cout;
        a_rel[i] = ((ty * a.OW + tx) * a.ld_y + co0 + cg * 8) * 2;
        a_lds[i] = p * RSA + cg * 16;
    }
    int p_r[NPL], p_c[NPL], p_rel[NPL], p_lds[NPL];
    bool p_ok[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        // A ds_write_b128 is served in groups of 8 lanes = 128 B = all 32 banks once when the group's two pixels lie 64 B (mod
        // 128) apart: with 96-byte pixels that is pixel p beside p + 2, with 80-byte pixels p beside p + 4 (`wpair`).  Neighbours
        // (p, p + 1) collide on a quarter / half of the banks (rocprofv3: 34 % of the LDS cycles were bank conflicts).
        const int c = tid + 256 * j, k = c >> 2, ch = c & 3;
        const int pp = (k & ~(2 * a.wpair - 1)) + ((k & (2 * a.wpair - 1)) >> 1) + (k & 1) * a.wpair;
        int pr, pc;
        fast_divmod(pp, a.PWt, a.inv_pw, pr, pc);
        p_r[j] = pr;
        p_c[j] = pc;
        p_ok[j] = pp < a.npix && ci0 + ch * 8 < a.Cin;
        p_rel[j] = ((pr * a.W + pc) * a.ld_x + ci0 + ch * 8) * 2;
        p_lds[j] = pp < a.npix ? pp * a.RS + ch * 16 : -1;
    }

    constexpr int DEPTH = 2;
    u32x4 ra[DEPTH][AP], rp[DEPTH][NPL];
    auto load_tile = [&](int t, u32x4 (&ra)[AP], u32x4 (&rp)[NPL]) {
        int n, rem, tyi, txi;
        fast_divmod(t, a.tiles_img, a.inv_img, n, rem);
        fast_divmod(rem, a.tiles_x, a.inv_tx, tyi, txi);
        const int oy0 = tyi * a.TH, ox0 = txi * a.TW;
        const int ybase = ((n * a.OH + oy0) * a.OW + ox0) * a.ld_y * 2;
        const int sy0 = oy0 * a.stride - a.pad_h, sx0 = ox0 * a.stride - a.pad_w;
        const int xbase = ((n * a.H + sy0) * a.W + sx0) * a.ld_x * 2;     // may be negative; only used where the pixel is inside
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const bool ok = a_ok[i] && oy0 + a_ty[i] < a.OH && ox0 + a_tx[i] < a.OW;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, ok ? ybase + a_rel[i] : (int)OOB, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const bool ok = p_ok[j] && (unsigned)(sy0 + p_r[j]) < (unsigned)a.H && (unsigned)(sx0 + p_c[j]) < (unsigned)a.W;
            rp[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? xbase + p_rel[j] : (int)OOB, 0, 0);
        }
    };
    auto store_tile = [&](int buf, const u32x4 (&ra)[AP], const u32x4 (&rp)[NPL]) {
#pragma unroll
        for (int i = 0; i < AP; ++i) *reinterpret_cast<u32x4*>(sA + buf * A_BYTES + a_lds[i]) = ra[i];
#pragma unroll
        for (int j = 0; j < NPL; ++j)
            if (p_lds[j] >= 0) *reinterpret_cast<u32x4*>(sP + buf * a.pbytes + p_lds[j]) = rp[j];
    };

    // ---- fragment addressing: lane -> (g, q, p4); k index (pixel) rho = q + 4(g&1) + 16(g>>1) + 8h + 32ks ----
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;
    const int rho0 = q + 4 * (g & 1) + 16 * (g >> 1);
    int aoff[2][2], boff[2][2];          // [ks][h]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rho = rho0 + 8 * h + 32 * ks;
            int ty, tx;
            fast_divmod(rho, a.TW, a.inv_tw, ty, tx);
            if (ty >= a.TH) { ty = 0; tx = 0; }          // not a tile pixel: its dY row is zero, any in-range patch address will do
            aoff[ks][h] = rho * RSA + (wm * TM + 4 * p4) * 2;
            boff[ks][h] = ((ty * a.stride) * a.PWt + tx * a.stride) * a.RS + (wn * 16 + 4 * p4) * 2;
        }

    f32x4 acc[TAPS][MREP];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < MREP; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    // ---- pipeline: DEPTH register stages, two LDS buffers, one barrier per tile; branch-free steady state (clamped index) ----
    // Within an iteration the global loads of tile st+2 and the LDS stores of tile st+1 are spread between the fragment groups of
    // the MFMA work on tile st (one load / one store per slot) instead of forming a load phase and a store phase.  What the
    // round-3 ablations of the step's launch showed (FN_WGT_DBG switches, since removed): loads alone 87 us, multiply alone 144 us,
    // stores 37 us, prologue + epilogue 44 us -- and the whole kernel 290 us, their SUM.  Neither this interleaving (260 vs
    // 261-273 us), nor a deeper register pipeline (DEPTH 3), nor fragment prefetch, nor staggered workgroup starts changed that;
    // what does is the number of workgroups a CU holds (513 / 336 / 260 us at 1 / 2 / 3).  MFMA utilisation ~35 %.
    static_assert(AP + NPL <= 9, "side-task slots");
    auto iteration = [&](const int buf, const bool do_compute, const int t_next, u32x4 (&ra_n)[AP], u32x4 (&rp_n)[NPL], const u32x4 (&ra_o)[AP],
                         const u32x4 (&rp_o)[NPL]) {
        int n, rem, tyi, txi;
        fast_divmod(t_next, a.tiles_img, a.inv_img, n, rem);
        fast_divmod(rem, a.tiles_x, a.inv_tx, tyi, txi);
        const int oy0 = tyi * a.TH, ox0 = txi * a.TW;
        const int ybase = ((n * a.OH + oy0) * a.OW + ox0) * a.ld_y * 2;
        const int sy0 = oy0 * a.stride - a.pad_h, sx0 = ox0 * a.stride - a.pad_w;
        const int xbase = ((n * a.H + sy0) * a.W + sx0) * a.ld_x * 2;     // may be negative; only used where the pixel is inside
        const unsigned char* pa = sA + buf * A_BYTES;
        const unsigned char* pb = sP + buf * a.pbytes;
        unsigned char* wa = sA + (buf ^ 1) * A_BYTES;
        unsigned char* wb = sP + (buf ^ 1) * a.pbytes;
        const bool nine = a.ntaps > 7;
        // One B fragment per tap and k step (2 transposed reads -> MREP MFMAs), through one register set.  A ring that keeps 1-8
        // fragments in flight ahead of the MFMAs (counted lgkmcnt) was measured: no gain at equal occupancy (338 us for every
        // distance at two waves per SIMD) and it costs the third wave per SIMD (260 us) -- the kernel's time follows the waves per
        // CU (513 / 336 / 260 us at one / two / three workgroups per CU), not the LDS latency of a single wave.
        constexpr int NFR = 2 * TAPS;
        vec8 fa[MREP];
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
            if (f % 2 == 0 && f / 2 < AP + NPL) {           // one global load of tile st+2 (slots 0, 2, 4, ...)
                const int k = f / 2;
                if (k < AP) {
                    const int i = k < AP ? k : 0;
                    const bool ok = a_ok[i] && oy0 + a_ty[i] < a.OH && ox0 + a_tx[i] < a.OW;
                    ra_n[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, ok ? ybase + a_rel[i] : (int)OOB, 0, 0);
                } else {
                    const int j = k >= AP ? k - AP : 0;
                    const bool ok = p_ok[j] && (unsigned)(sy0 + p_r[j]) < (unsigned)a.H && (unsigned)(sx0 + p_c[j]) < (unsigned)a.W;
                    rp_n[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? xbase + p_rel[j] : (int)OOB, 0, 0);
                }
            }
            if (f >= TAPS - 1 && (f - (TAPS - 1)) % 2 == 0 && (f - (TAPS - 1)) / 2 < AP + NPL) {   // one LDS store of tile st+1 (slots 8, 10, ...)
                const int k = (f - (TAPS - 1)) / 2;
                if (k < AP) {
                    const int i = k < AP ? k : 0;
                    *reinterpret_cast<u32x4*>(wa + a_lds[i]) = ra_o[i];
                } else {
                    const int j = k >= AP ? k - AP : 0;
                    if (p_lds[j] >= 0) *reinterpret_cast<u32x4*>(wb + p_lds[j]) = rp_o[j];
                }
            }
            if (do_compute && f % TAPS == 0) {
#pragma unroll
                for (int i = 0; i < MREP; ++i) {
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa + aoff[f / TAPS][0] + i * 32));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa + aoff[f / TAPS][1] + i * 32));
                    fa[i] = __builtin_bit_cast(vec8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            }
            if (do_compute && (f % TAPS < 7 || nine)) {     // 1x7 / 7x1 layers have seven taps (uniform test, compile-time indices)
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pb + boff[f / TAPS][0] + a.tapoff[f % TAPS]));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pb + boff[f / TAPS][1] + a.tapoff[f % TAPS]));
                const vec8 fb = __builtin_bit_cast(vec8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int i = 0; i < MREP; ++i) acc[f % TAPS][i] = LP<T>::mfma(fa[i], fb, acc[f % TAPS][i]);
            }
        }
    };
    const int last = tend - 1;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_tile(min(tbeg + d, last), ra[d], rp[d]);
    store_tile(0, ra[0], rp[0]);
    __syncthreads();
    for (int s0 = 0; s0 < nst; s0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int st = s0 + d;
            iteration(st & 1, st < nst, min(tbeg + st + DEPTH, last), ra[d], rp[d], ra[(d + 1) % DEPTH], rp[(d + 1) % DEPTH]);
            __syncthreads();
        }
    }

    // ---- C layout: col = lane & 15 (input channel), row = (lane >> 4) * 4 + r (cout) ----
    float* const dst = a.out.ws ? a.out.ws + (long)bz * a.out.Cout * a.out.KTOT : a.out.dw;
    const int ci = ci0 + wn * 16 + (lane & 15);
    if (ci < a.Cin) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            if (t < a.ntaps) {
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = co0 + wm * TM + i * 16 + g * 4 + r;
                        if (co < a.out.Cout) dst[(long)co * a.out.KTOT + t * a.Cin + ci] = acc[t][i][r];
                    }
            }
        }
    }
}

template <typename T, int BMW, int TAPS, int NPL>
__global__ __launch_bounds__(256) void conv_wgrad_taps_kernel(const unsigned char* __restrict__ args, int stride, const int* __restrict__ prefix, int n) {
    const int bid = blockIdx.x;
    int lo = 0, hi = n;                    // largest layer index with prefix[lo] <= bid
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= bid) lo = mid; else hi = mid;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    // records are `stride` bytes apart (fn_conv2d_wgrad_arg_bytes(): one record size for both weight-gradient kernels)
    const WgradTapArgs a = *reinterpret_cast<const WgradTapArgs*>(args + (long)lo * stride);
    // workgroups that share a pixel split (all gx*gy tiles of it) are consecutive in the layer's own order and get one XCD:
    // the split's dY tiles and patches are fetched into one L2
    const int gxy = a.gx * a.gy;
    const int local = xcd_remap(bid - prefix[lo], gxy * a.out.splits);
    const int bz = local / gxy, r = local - bz * gxy;
    conv_wgrad_taps_body<T, BMW, TAPS, NPL>(a, r % a.gx, r / a.gx, bz);
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static int taps_of(const fn_conv_desc* d) { return d->KH * d->KW; }

// ONE instantiation per dtype serves every layer this kernel takes (3x3 stride 1 / 2, 1x7, 7x1; 64-cout tiles, up to nine taps,
// patches of up to 160 / 192 pixels): all of them go into a single grouped launch, which is what fills the chip -- per variant the
// launches held 160-370 workgroups with 90-stage chains each and ran one after the other (measured: 363 us for the four).
enum { TAPS_BMW = 64, TAPS_MAX = 9, TAPS_NPL = 3, TAPS_PATCH_BYTES = 15360 };   // patch buffer: 160 px x 96 B (stride 1) / 192 px x 80 B (stride 2)
int wgrad_taps_variant(const fn_conv_desc* d) {
    static const int enabled = getenv("FN_WGRAD_TAPS") ? atoi(getenv("FN_WGRAD_TAPS")) : 1;
    static const int min_pix = getenv("FN_WGRAD_TAPS_MINPIX") ? atoi(getenv("FN_WGRAD_TAPS_MINPIX")) : 32;
    if (!enabled || d->nrm_stats) return 0;
    const bool shape = (d->KH == 3 && d->KW == 3 && (d->stride == 1 || d->stride == 2)) ||
                       (d->stride == 1 && ((d->KH == 1 && d->KW == 7) || (d->KH == 7 && d->KW == 1)));
    // 3x3 maps: a 64-pixel tile would be mostly empty; Conv2d_1a (3 -> 8 padded input channels): a quarter of one 32-channel slice
    if (!shape || d->OH * d->OW < min_pix || d->Cin < 32) return 0;
    return WGRAD_TAPS_VARIANT + TAPS_BMW * 1000 + TAPS_MAX * 10;
}

size_t wgrad_taps_arg_bytes() { return sizeof(WgradTapArgs); }

// tile of TH x TW <= 64 output pixels: minimise (stages per image) x (stage cost), stage cost = the larger of its MFMA time and
// its operand bytes at the ~28 B/clk a CU draws from L2, stretched by the bank conflicts of the patch reads: a 32-lane half of
// ds_read_b64_tr_b16 reads eight CONSECUTIVE tile pixels (32 B each); they are conflict free when their patch addresses fall into
// eight different 32-byte groups mod 256, which holds inside a tile row (s * RS = 32 mod 64) but not across a row wrap -- narrow
// tiles (TW = 3, 4, 7) wrapped inside every group of eight (rocprofv3: a quarter of the LDS cycles were conflicts).
static double patch_read_conflicts(int th, int tw, int s, int pwt, int rs) {
    double cycles = 0;
    for (int base = 0; base < 64; base += 8) {
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, worst = 1;
        for (int i = 0; i < 8; ++i) {
            int ty = (base + i) / tw, tx = (base + i) % tw;
            if (ty >= th) { ty = 0; tx = 0; }
            const int grp = ((((ty * s) * pwt + tx * s) * rs) & 255) >> 5;
            worst = std::max(worst, ++cnt[grp]);
        }
        cycles += worst;
    }
    return cycles / 8.0;      // 1.0 = conflict free
}

static bool pick_tile(const fn_conv_desc* d, int bmw, int rs, int maxpix, int& TH, int& TW) {
    double best = 1e30;
    bool found = false;
    const int s = d->stride;
    for (int tw = 1; tw <= 64 && tw <= d->OW; ++tw) {
        const int th = std::min(64 / tw, d->OH);
        if (th < 1) continue;
        const int ph = (th - 1) * s + d->KH, pw = (tw - 1) * s + d->KW;
        if (ph * pw > maxpix) continue;
        const double mfma = 64.0 * bmw * 32 * d->KH * d->KW / 2048.0;
        const double bytes = (64.0 * bmw * 2 + ph * pw * 64.0) / 28.0;
        const double lds = 0.5 * mfma * (patch_read_conflicts(th, tw, s, pw, rs) - 1.0);     // B reads are ~half of a stage's LDS cycles
        const double cost = (double)cdiv(d->OH, th) * cdiv(d->OW, tw) * (std::max(mfma, bytes) + lds);
        if (cost < best - 1e-9) { best = cost; TH = th; TW = tw; found = true; }
    }
    return found;
}

long wgrad_taps_plan(const fn_conv_desc* d, int variant, void* rec, float* ws, long* ws_used) {
    FN_REQUIRE(d->x && d->y && d->dw, "conv_wgrad(taps): null x/dy/dw");
    FN_REQUIRE(wgrad_taps_variant(d) == variant, "wgrad_group_build: the descriptor dispatches to %d, group is %d", wgrad_taps_variant(d), variant);
    FN_REQUIRE(d->Cin % 8 == 0 && d->ld_x % 8 == 0 && d->ld_y % 8 == 0 && d->ld_y >= d->Cout, "conv_wgrad(taps): channels / strides must be multiples of 8");
    const int bmw = TAPS_BMW;
    WgradTapArgs a{};
    a.ntaps = d->KH * d->KW;
    a.out.dw = d->dw;
    a.out.Cout = d->Cout;
    a.out.KTOT = d->KH * d->KW * d->Cin;
    a.out.store = 1;
    a.x = (const unsigned short*)d->x;
    a.dy = (const unsigned short*)d->y;
    a.H = d->H; a.W = d->W; a.OH = d->OH; a.OW = d->OW; a.Cin = d->Cin; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
    a.ld_x = d->ld_x; a.ld_y = d->ld_y;
    FN_REQUIRE((long)d->N * d->H * d->W * d->ld_x * 2 < (1L << 30) && (long)d->N * d->OH * d->OW * d->ld_y * 2 < (1L << 30),
               "conv_wgrad(taps): x or dy exceeds the 1 GiB range of 32-bit buffer offsets");
    a.x_bytes = d->N * d->H * d->W * d->ld_x * 2;
    a.dy_bytes = d->N * d->OH * d->OW * d->ld_y * 2;
    a.RS = d->stride == 1 ? 96 : 80;        // s * RS = 32 (mod 64): eight consecutive tile pixels hit eight different 32-byte bank groups
    FN_REQUIRE(pick_tile(d, bmw, a.RS, TAPS_PATCH_BYTES / a.RS, a.TH, a.TW), "conv_wgrad(taps): no pixel tile fits");
    const int PHt = (a.TH - 1) * d->stride + d->KH;
    a.PWt = (a.TW - 1) * d->stride + d->KW;
    a.npix = PHt * a.PWt;
    a.wpair = d->stride == 1 ? 2 : 4;       // wpair * RS = 64 (mod 128)
    a.pbytes = (a.npix * a.RS + 15) / 16 * 16;
    FN_REQUIRE(a.pbytes <= TAPS_PATCH_BYTES && a.npix * 4 <= 256 * TAPS_NPL, "conv_wgrad(taps): patch of %d pixels does not fit", a.npix);
    const int tiles_y = cdiv(d->OH, a.TH);
    a.tiles_x = cdiv(d->OW, a.TW);
    a.tiles_img = tiles_y * a.tiles_x;
    const long ntiles = (long)d->N * a.tiles_img;
    FN_REQUIRE(ntiles < (1L << 24), "conv_wgrad(taps): too many pixel tiles");
    a.ntiles = (int)ntiles;
    a.gx = cdiv(d->Cin, 32);
    a.gy = cdiv(d->Cout, bmw);
    for (int t = 0; t < d->KH * d->KW; ++t) a.tapoff[t] = ((t / d->KW) * a.PWt + (t % d->KW)) * a.RS;
    a.inv_img = 1.0f / (float)a.tiles_img; a.inv_tx = 1.0f / (float)a.tiles_x; a.inv_tw = 1.0f / (float)a.TW; a.inv_pw = 1.0f / (float)a.PWt;
    // pixel splits.  The grouped launch holds the workgroups of every k x k layer, so a layer need not fill the chip alone; what a
    // split costs is one more fp32 copy of the layer's dW written and read again (slab), what it buys is a shorter serial chain
    // of stages (~0.5 us each): chains of about `chain` stages, slabs of at most `slab_mb` MB per layer.
    static const int chain = getenv("FN_WGT_CHAIN") ? atoi(getenv("FN_WGT_CHAIN")) : 64;
    static const int slab_mb = getenv("FN_WGT_SLAB_MB") ? atoi(getenv("FN_WGT_SLAB_MB")) : 16;
    const long numel = (long)a.out.Cout * a.out.KTOT;
    int splits = d->splits > 0 ? d->splits : std::max(1, std::min(cdiv(ntiles, chain), (int)(((long)slab_mb << 20) / (numel * 4))));
    a.chunk = cdiv(ntiles, splits);
    splits = cdiv(ntiles, a.chunk);
    a.out.splits = splits;
    a.out.ws = nullptr;
    if (splits > 1) {
        a.out.ws = ws ? ws + *ws_used : reinterpret_cast<float*>(16);
        *ws_used += (long)splits * a.out.Cout * a.out.KTOT;
    }
    static const int debug = getenv("FN_WGT_DEBUG") ? atoi(getenv("FN_WGT_DEBUG")) : 0;
    if (debug && ws == nullptr)
        fprintf(stderr, "wgrad_taps: %dx%dx%d->%d k%dx%d s%d: tile %dx%d patch %d px, %d tiles, %d x %d x %d workgroups\n", d->H, d->W, d->Cin, d->Cout,
                d->KH, d->KW, d->stride, a.TH, a.TW, a.npix, a.ntiles, a.gx, a.gy, splits);
    *reinterpret_cast<WgradTapArgs*>(rec) = a;
    return (long)a.gx * a.gy * splits;
}

template <typename T> static int launch_taps(const unsigned char* a, const int32_t* prefix, int n, int total, int variant, hipStream_t st) {
    if (variant != WGRAD_TAPS_VARIANT + TAPS_BMW * 1000 + TAPS_MAX * 10) {
        set_error("wgrad_taps: unknown variant %d", variant);
        return FN_EINVAL;
    }
    const int stride = fn_conv2d_wgrad_arg_bytes();
    constexpr size_t smem = 2 * 64 * (TAPS_BMW * 2 + 32) + 2 * TAPS_PATCH_BYTES;      // 20 KB of dY tiles + two patch buffers = 50 KB: three workgroups per CU
    auto kern = conv_wgrad_taps_kernel<T, TAPS_BMW, TAPS_MAX, TAPS_NPL>;
    hipLaunchKernelGGL(kern, dim3(total), dim3(256), smem, st, a, stride, prefix, n);
    return check_launch("conv_wgrad_taps");
}

int wgrad_taps_launch(const void* dev_args, const int32_t* dev_prefix, int n, int total, int variant, int dtype, hipStream_t st) {
    const unsigned char* a = reinterpret_cast<const unsigned char*>(dev_args);
    return dtype == FN_BF16 ? launch_taps<__bf16>(a, dev_prefix, n, total, variant, st) : launch_taps<_Float16>(a, dev_prefix, n, total, variant, st);
}

}  // namespace fn
