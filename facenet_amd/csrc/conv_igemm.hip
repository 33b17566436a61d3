// Implicit-GEMM convolution for gfx950 (MI355X): forward, data-gradient and weight-gradient.
//
// Replaces tf.keras.layers.Conv2D / Dense and the ops Keras wraps around them in
// facenet/models/inception_resnet_v1.py (Conv2D :90-138,:160-193,:215-248,:269-299,:316-367,
// :387-430; tf.concat :141,196,251,304,372; residual scale-add :145-148,199-202,254-257; Dense :462)
// and apps/train_softmax.py:57-63.
//
// Design (MI355X-first, not a cuDNN-style translation):
//   * no im2col buffer: the A operand is gathered straight from the NHWC activation tensor, 16 B
//     (8 channels) per lane, k = (tap, channel) decoded through a small LDS table;
//   * 64-lane waves, v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate; 4 waves per workgroup;
//   * tiles staged global -> VGPR -> LDS (issue-early / write-late, double-buffered LDS, one barrier
//     per 64-deep K step); LDS rows are 128 B with an XOR-8 chunk swizzle so every ds_read_b128
//     fragment read is bank-conflict free;
//   * dgrad is the SAME kernel with a transposed-conv gather (so/sk/div parameters) reading the
//     [Cin][tap][Cout] weight pack;
//   * wgrad reduces over pixels (K = N*OH*OW): both operands are k-strided in memory, so fragments
//     come from LDS through ds_read_b64_tr_b16 (hardware transpose read); split-K over pixels with
//     fp32 global atomics into dW;
//   * epilogue through LDS (fp32 C tile) so global stores are full 16-B coalesced rows; fused there:
//     bias, residual scale-add, ReLU, accumulate, BatchNorm batch statistics (sum / sum of squares),
//     and channel-slice output (ld_out) which makes tf.concat free.
#include "common.h"
#include "../../include/facenet_hip.h"
#include "wgrad_taps.h"
#include <cstdio>
#include <type_traits>
#include <cstdlib>

#ifndef FN_IG_DBG
#define FN_IG_DBG 0     // developer builds only (-DFN_IG_DBG=n, tools/dev_stemtiles.py / dev_phases.py): ablations 1 no loads, 2 no multiply, 4 no LDS stores,
                        // 8 no epilogue, 16 no first tile (results are wrong by construction); 32 per-workgroup phase clocks (results unchanged)
#endif

#if FN_IG_DBG & 32      // per-workgroup phase clocks (100 MHz wall clock), summed per kernel class: tools/dev_phases.py
__device__ unsigned long long fn_ig_phase[256 * 576 * 16];     // 256 replicas: same-address atomics would serialise the whole chip
extern "C" int fn_debug_phases(unsigned long long* out, int reset) {
    static unsigned long long host[256 * 576 * 16];
    if (out) {
        if (hipMemcpyFromSymbol(host, HIP_SYMBOL(fn_ig_phase), sizeof(host)) != hipSuccess) return -1;
        for (int i = 0; i < 576 * 16; ++i) {
            out[i] = 0;
            for (int r = 0; r < 256; ++r) out[i] += host[r * 576 * 16 + i];
        }
    }
    if (reset) {
        for (auto& v : host) v = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(fn_ig_phase), host, sizeof(host)) != hipSuccess) return -1;
    }
    return 0;
}
#define FN_PHASE_CLOCK(v) const unsigned long long v = wall_clock64()
#define FN_EP_PARAM , unsigned long long* ep
#define FN_EP_ARG , ep
#define FN_EP_STAMP(i) ep[i] = wall_clock64()
#else
#define FN_PHASE_CLOCK(v)
#define FN_EP_PARAM
#define FN_EP_ARG
#define FN_EP_STAMP(i)
#endif

namespace fn {

struct ConvArgs {
    const unsigned short* src;  // gathered activation operand
    const unsigned short* wp;   // packed weights [NOUT][KTOT]
    void* out;
    const float* bias;
    acc_t* stats;               // BatchNorm batch statistics (fixed point, ACC_STAT): stats[rep*stride + c] += sum y, [.. + sq_off + c] += sum y^2
    const unsigned short* resid;
    int M, PH, PW;   // output pixels = N*PH*PW
    int SH, SW;      // source spatial dims
    int CS;          // source channels per tap
    int NOUT, KTOT, KH, KW;
    int so, sk, offy, offx, dshift;  // t = p*so + k*sk + off ; src = t >> dshift, valid iff t>=0, (t & ((1<<dshift)-1))==0, src < S
    int ld_src, ld_out, ld_res;
    int relu, accumulate, out_f32;
    float scale;
    int tiles_m, tiles_n;
    int stats_sq_off, stats_replicas, stats_rep_stride;
    int plain;  // 1x1 / stride 1 / no padding: source pixel == output pixel, k == channel
    // dgrad epilogue: reduction of the BatchNorm backward of the layer whose output gradient this launch produces
    const unsigned short* bn_y;   // raw forward output of that layer (same pixels / channel slice as `out`)
    const float* bn_scale;
    const float* bn_shift;
    const float* bn_beta;
    acc_t* bn_acc;                // fixed point (ACC_GRAD): acc[rep*stride + c] += sum dyh ; acc[rep*stride + sq_off + c] += sum dyh*xhat
    int ld_bn_y, bn_sq_off, bn_replicas, bn_rep_stride, bn_relu;
    // stride-2 dgrad: output pixels are split into 4 parity classes ((iy+pad)&1, (ix+pad)&1); a class only sees the taps
    // of matching parity, so each class is its own GEMM (M = its pixels, K = its taps) inside one launch.
    int s2;                 // 1 = class mode
    int cp1, cp2, cp3;      // first tile of classes 1..3 (class 0 starts at 0)
    int total_tiles;
    int src_bytes, w_bytes;   // extents for the buffer resource descriptors (< 2^30)
    int tile;                 // 0 = heuristic, BM*1000+BN = caller's choice (fn_conv_desc.tile_fwd / tile_dgrad)
    int nocheck;              // forward, no padding: taps never leave the source, the per-chunk bounds test is skipped
    // 1x1 data gradient of SIBLING layers that read the same input: dX = sum_s dY_s * Wt_s as ONE GEMM whose K runs through
    // the sources (k tiles [0,t1) source 1, [t1,t2) source 2, [t2,nt_total) source 3); nt_total == 0: single source
    const unsigned short* src2; const unsigned short* wp2;
    const unsigned short* src3; const unsigned short* wp3;
    int K2, ld2, K3, ld3, t1, t2, nt_total, src2_bytes, w2_bytes, src3_bytes, w3_bytes;
    // normalise-on-load (forward only): src is the raw output of a BN(center)+ReLU layer, see fn_conv_desc.nrm_*
    const acc_t* nrm_stats;
    const float* nrm_beta;
    int nrm_sq_off, nrm_replicas, nrm_rep_stride, nrm_count;
    float nrm_eps;
    unsigned short* nrm_z;    // optional: the normalised operand is also written here (geometry of src), see fn_conv_desc.nrm_z
    // dgrad epilogue: fused residual backward (fn_conv_desc.rb_*).  `resid` (scale 1) carries rb_prev, `out` is rb_dtrunk.
    int halo_ty, halo_tx;         // halo kernel: 8x16-pixel output tiles per image (rows, columns)
    const unsigned short* mask;   // rows of the block's forward output: values <= 0 zero the gradient
    unsigned short* out2;         // scale2 * (masked gradient), geometry of out
    acc_t* colsum;                // fixed point (ACC_GRAD): += column sums of what goes to out2
    float scale2;
    const float* prelu;           // forward: per-output-channel PReLU slope applied to conv + bias
};

// q = m / d, r = m % d through the hardware reciprocal (0 <= m < 2^24, d > 0): integer division is a ~40-instruction
// sequence on this ISA and the prologue of every workgroup needs several
__device__ __forceinline__ void rcp_divmod(int m, int d, int& q, int& r) { fast_divmod(m, d, __builtin_amdgcn_rcpf((float)d), q, r); }

// class mode: taps ky in {qy, qy+2, ..}, kx in {qx, qx+2, ..}
__device__ __forceinline__ int ktab_entry_s2(int kgroup, int CS, int KH, int KW, int qy, int qx) {
    const int nky = (KH - qy + 1) >> 1, nkx = (KW - qx + 1) >> 1;
    const int k = kgroup * 8;
    if (k >= nky * nkx * CS) return -1;
    int tap, c, ty, tx;
    rcp_divmod(k, CS, tap, c);
    rcp_divmod(tap, nkx, ty, tx);
    return ((qy + 2 * ty) << 24) | ((qx + 2 * tx) << 16) | c;
}

__device__ __forceinline__ int ktab_entry(int kgroup, int KTOT, int CS, int KW) {
    const int k = kgroup * 8;
    if (k >= KTOT) return -1;
    int tap, c, ky, kx;
    rcp_divmod(k, CS, tap, c);
    rcp_divmod(tap, KW, ky, kx);
    return (ky << 24) | (kx << 16) | c;
}

// Epilogue of every forward / data-gradient convolution kernel: the fp32 accumulators go through LDS (C tile) so that global
// stores are full 16-byte rows, and everything that touches the output once is fused here: bias, residual scale-add, ReLU
// (or the ReLU mask of a fused residual backward), accumulate, BatchNorm batch statistics, the BatchNorm-backward reduction
// of the producing layer, channel-slice output.  Tile rows map to output pixels either linearly (m0 + row < a.M) or through
// the row table sRow (-1 = no pixel): the parity classes of a stride-2 data gradient and the 2-D tiles of the halo kernel.
// `active`: threads 0..255 of the group that owns the accumulators; every thread of the workgroup must call (barriers).
template <typename T, int BM, int BN, int WM, int WN, bool MULTI = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], unsigned char* smem, float* sRed,
                                              const int* sRow, const bool rowtab, const int m0, const int n0, const int tm, const bool active FN_EP_PARAM) {
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MREP = TM / 16, NREP = TN / 16;
    constexpr int CLD = BN + 4;
    float* sC = reinterpret_cast<float*>(smem);
    constexpr int GT = 64 * WM * WN;      // threads of the group that owns the accumulators (256, or 512 for the 8-wave tiles)
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid % GT) >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    if (a.stats && active) {  // BatchNorm batch statistics from the fp32 accumulators (rows >= M are exact zeros)
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[i][j][r];
                    s += v;
                    q += v * v;
                }
            s += __shfl_xor(s, 16);
            q += __shfl_xor(q, 16);
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            if (lane < 16) {      // slot [wm]: one writer per (row wave, column) -- no LDS atomics, the WM partials are added in order below
                sRed[wm * 2 * BN + wn * TN + j * 16 + lane] = s;
                sRed[wm * 2 * BN + BN + wn * TN + j * 16 + lane] = q;
            }
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sC[(wm * TM + i * 16 + fq * 4 + r) * CLD + wn * TN + j * 16 + fr] = acc[i][j][r];
    }
    __syncthreads();
    FN_EP_STAMP(0);

    constexpr int CG = BN / 8, RP = GT / CG;
    const int cg = tid % CG, rr = tid / CG;
    const int col = n0 + cg * 8;
    float bq1[8], bq2[8];   // fused BN-backward partial sums of this thread's 8 columns
#pragma unroll
    for (int e = 0; e < 8; ++e) { bq1[e] = 0.f; bq2[e] = 0.f; }
    if (active && col < a.NOUT) {
        // What a layer's epilogue does beyond bias + ReLU decides its code path (uniform branch).  Each path keeps only its own
        // operands in registers -- the epilogue is where these kernels peak in VGPRs, and one path carrying every option costs the
        // main loop a wave per SIMD -- and each requests everything it READS from global memory for a chunk of row passes
        // before the chunk's first store: with one pass at a time, the wait for a pass's loads also waited for the previous pass's
        // store (vmcnt counts both on this ISA), one full write round trip per pass, 6-9 us of a 128-row tile's epilogue
        // (tools/dev_phases.py).  Buffer descriptors make the accesses unconditional: a tile row without a pixel gets the
        // offset OOB, its loads return zeros and its stores are dropped without touching memory (byte offsets < 2^31).
        enum { EP_PLAIN = 0, EP_RESID = 1, EP_BNBWD = 2, EP_RESBWD = 3, EP_GENERIC = 4, EP_ACC = 5 };
        const bool full = (col + 8 <= a.NOUT);
        constexpr int NP = (BM + RP - 1) / RP;
        constexpr int CHMAX = BM * BN == 128 * 64 ? 2 : 4;    // row passes in flight together; 8 192-element tiles sit one register below an occupancy step
        constexpr unsigned OOB = 0x80000000u;
        auto rsrc = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, ptr ? 0x7fffffff : 0, 0x00020000); };
        auto row_pixel = [&](const int row) {
            if (row >= BM) return -1;
            if (rowtab) return sRow[row];
            return m0 + row < a.M ? m0 + row : -1;
        };
        auto byte_off = [&](const int m, const int ld) { return (int)(m >= 0 ? (unsigned)(m * ld + col) * 2u : OOB); };
        float bias[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias[e] = (a.bias && col + e < a.NOUT) ? a.bias[col + e] : 0.f;
        auto tile_row = [&](const int row, float (&v)[8]) {      // C tile row + bias
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(&sC[row * CLD + cg * 8]);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(&sC[row * CLD + cg * 8 + 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = c0[e] + bias[e];
                v[4 + e] = c1[e] + bias[4 + e];
            }
        };
        auto relu8 = [&](float (&v)[8]) {
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
        };
        const int kind = (a.out_f32 || !full || a.prelu) ? EP_GENERIC
                         : a.accumulate                   ? ((a.mask || a.out2 || a.bn_y || a.resid) ? EP_GENERIC : EP_ACC)
                         : (a.mask || a.out2)             ? ((a.mask && a.out2 && !a.bn_y) ? EP_RESBWD : EP_GENERIC)
                         : a.bn_y                         ? (a.resid ? EP_GENERIC : EP_BNBWD)
                         : a.resid                        ? EP_RESID
                                                          : EP_PLAIN;
#if FN_IG_DBG & 32
        ep[2] = (unsigned long long)kind;
#endif
        if (kind == EP_PLAIN) {            // bias + ReLU: nothing to wait for
            const __amdgpu_buffer_rsrc_t rs_out = rsrc(a.out);
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                const int row = ps * RP + rr;
                float v[8];
                tile_row(row < BM ? row : 0, v);
                relu8(v);
                __builtin_amdgcn_raw_buffer_store_b128(pack8<T>(v), rs_out, byte_off(row_pixel(row), a.ld_out), 0, STORE_WT);
            }
        } else if (kind == EP_RESID || kind == EP_ACC) {     // residual scale-add of the block `up` layers / a data gradient added to an earlier one
            constexpr int CH = NP < CHMAX ? NP : CHMAX;
            const bool acc_mode = kind == EP_ACC;
            const __amdgpu_buffer_rsrc_t rs_out = rsrc(a.out), rs_res = rsrc(acc_mode ? (const void*)a.out : (const void*)a.resid);
            const int ld_q = acc_mode ? a.ld_out : a.ld_res;
            const float vs = acc_mode ? 1.f : a.scale;
#pragma unroll 1
            for (int p0 = 0; p0 < NP; p0 += CH) {
                int mm[CH];
                u32x4 q[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    mm[c] = row_pixel((p0 + c) * RP + rr);
                    q[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, byte_off(mm[c], ld_q), 0, 0);
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int row = (p0 + c) * RP + rr;
                    float v[8], rv[8];
                    tile_row(row < BM ? row : 0, v);
                    unpack8<T>(q[c], rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = rv[e] + vs * v[e];      // accumulate: vs = 1, the same bits as v + rv
                    relu8(v);
                    q[c] = pack8<T>(v);
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) __builtin_amdgcn_raw_buffer_store_b128(q[c], rs_out, byte_off(mm[c], a.ld_out), 0, STORE_WT);
            }
        } else if (kind == EP_BNBWD) {     // data gradient + the BatchNorm-backward sums of the layer that produced its input
            constexpr int CH = NP < CHMAX ? NP : CHMAX;
            const __amdgpu_buffer_rsrc_t rs_out = rsrc(a.out), rs_bny = rsrc(a.bn_y);
            float bnsc[8], bnsf[8], bnbt[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { bnsc[e] = a.bn_scale[col + e]; bnsf[e] = a.bn_shift[col + e]; bnbt[e] = a.bn_beta[col + e]; }
#pragma unroll 1
            for (int p0 = 0; p0 < NP; p0 += CH) {
                int mm[CH];
                u32x4 q[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    mm[c] = row_pixel((p0 + c) * RP + rr);
                    q[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_bny, byte_off(mm[c], a.ld_bn_y), 0, 0);
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int row = (p0 + c) * RP + rr;
                    float v[8], yy[8];
                    tile_row(row < BM ? row : 0, v);
                    unpack8<T>(q[c], yy);
                    if (mm[c] >= 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float zf = fmaf(yy[e], bnsc[e], bnsf[e]);
                            const float gg = (!a.bn_relu || zf > 0.f) ? v[e] : 0.f;
                            bq1[e] += gg;
                            bq2[e] += gg * (zf - bnbt[e]);
                        }
                    }
                    relu8(v);
                    q[c] = pack8<T>(v);
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) __builtin_amdgcn_raw_buffer_store_b128(q[c], rs_out, byte_off(mm[c], a.ld_out), 0, STORE_WT);
            }
        } else if (kind == EP_RESBWD) {    // fused residual backward: [carried gradient +] ReLU mask of the block output, scaled copy for the `up` branch
            // two operands + two results per pass: four passes in flight cost the single-source 128x128 kernels a wave per SIMD;
            // the sibling-source kernels (where this path runs) have the registers
            constexpr int CH = MULTI ? (NP < 4 ? NP : 4) : (NP < 2 ? NP : 2);
            const __amdgpu_buffer_rsrc_t rs_out = rsrc(a.out), rs_res = rsrc(a.resid), rs_mask = rsrc(a.mask), rs_out2 = rsrc(a.out2);
#pragma unroll 1
            for (int p0 = 0; p0 < NP; p0 += CH) {
                int mm[CH];
                u32x4 qr[CH], qm[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    mm[c] = row_pixel((p0 + c) * RP + rr);
                    qr[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, byte_off(mm[c], a.ld_res), 0, 0);     // zeros without a carried gradient
                    qm[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_mask, byte_off(mm[c], a.ld_out), 0, 0);
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int row = (p0 + c) * RP + rr;
                    float v[8], rv[8], mk[8], u[8];
                    tile_row(row < BM ? row : 0, v);
                    if (a.resid) {
                        unpack8<T>(qr[c], rv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = rv[e] + a.scale * v[e];
                    }
                    unpack8<T>(qm[c], mk);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = mk[e] > 0.f ? v[e] : 0.f;
                        u[e] = a.scale2 * v[e];
                        if (mm[c] >= 0) bq1[e] += u[e];
                    }
                    relu8(v);
                    qr[c] = pack8<T>(u);
                    qm[c] = pack8<T>(v);
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    __builtin_amdgcn_raw_buffer_store_b128(qr[c], rs_out2, byte_off(mm[c], a.ld_out), 0, STORE_WT);
                    __builtin_amdgcn_raw_buffer_store_b128(qm[c], rs_out, byte_off(mm[c], a.ld_out), 0, STORE_WT);
                }
            }
        } else {
            // everything else, one row pass at a time: fp32 output (logits), ragged column groups, PReLU (MTCNN), accumulating
            // data gradients, and the combinations the paths above do not name
            float bnsc[8], bnsf[8], bnbt[8];
            if (a.bn_y) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { bnsc[e] = a.bn_scale[col + e]; bnsf[e] = a.bn_shift[col + e]; bnbt[e] = a.bn_beta[col + e]; }
            }
            float slope[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) slope[e] = (a.prelu && col + e < a.NOUT) ? a.prelu[col + e] : 1.f;
#pragma unroll 1
            for (int ps = 0; ps < NP; ++ps) {
                const int row = ps * RP + rr;
                const int m = row_pixel(row);
                if (m < 0) continue;
                float v[8];
                tile_row(row, v);
                if (a.prelu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : slope[e] * v[e];
                }
                if (a.resid) {
                    float rv[8];
                    unpack8<T>(*reinterpret_cast<const u32x4*>(a.resid + (long)m * a.ld_res + col), rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = rv[e] + a.scale * v[e];
                }
                const long o = (long)m * a.ld_out + col;
                if (a.mask) {
                    float mk[8];
                    unpack8<T>(*reinterpret_cast<const u32x4*>(a.mask + o), mk);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
                }
                if (a.out2) {
                    float u[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { u[e] = a.scale2 * v[e]; bq1[e] += u[e]; }
                    *reinterpret_cast<u32x4*>(a.out2 + o) = pack8<T>(u);
                }
                if (a.bn_y) {   // dgrad only (no resid/relu/f32 here): v is the complete gradient unless accumulating
                    float yy[8];
                    unpack8<T>(*reinterpret_cast<const u32x4*>(a.bn_y + (long)m * a.ld_bn_y + col), yy);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float zf = fmaf(yy[e], bnsc[e], bnsf[e]);
                        const float gg = (!a.bn_relu || zf > 0.f) ? v[e] : 0.f;
                        bq1[e] += gg;
                        bq2[e] += gg * (zf - bnbt[e]);
                    }
                }
                if (a.out_f32) {
                    float* op = reinterpret_cast<float*>(a.out) + o;
                    if (full && !a.accumulate) {
                        relu8(v);
                        *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    } else {
                        for (int e = 0; e < 8 && col + e < a.NOUT; ++e) {
                            float x = v[e] + (a.accumulate ? op[e] : 0.f);
                            op[e] = a.relu ? fmaxf(x, 0.f) : x;
                        }
                    }
                } else {
                    unsigned short* op = reinterpret_cast<unsigned short*>(a.out) + o;
                    if (full) {
                        if (a.accumulate) {
                            float pv[8];
                            unpack8<T>(*reinterpret_cast<const u32x4*>(op), pv);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += pv[e];
                        }
                        relu8(v);
                        *reinterpret_cast<u32x4*>(op) = pack8<T>(v);
                    } else {
                        for (int e = 0; e < 8 && col + e < a.NOUT; ++e) {
                            float x = v[e] + (a.accumulate ? LP<T>::to_f32(op[e]) : 0.f);
                            op[e] = LP<T>::from_f32(a.relu ? fmaxf(x, 0.f) : x);
                        }
                    }
                }
            }
        }
    }
    FN_EP_STAMP(1);
    if (a.bn_y || a.out2) {   // fold the RP row lanes through LDS (the C tile is no longer needed), one atomic per column per block
        __syncthreads();
        float* sP = reinterpret_cast<float*>(smem);   // [RP][2*BN]
        if (active) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sP[rr * 2 * BN + cg * 8 + e] = bq1[e];
                sP[rr * 2 * BN + BN + cg * 8 + e] = bq2[e];
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int c = tid < BN ? tid : tid - BN;
            if (n0 + c < a.NOUT) {
                float sum = 0.f;
#pragma unroll 8
                for (int t = 0; t < RP; ++t) sum += sP[t * 2 * BN + tid];
                if (a.out2) {
                    if (tid < BN) acc_add<ACC_GRAD>(&a.colsum[n0 + c], sum);
                } else {
                    acc_t* ap = a.bn_acc + (long)(tm % a.bn_replicas) * a.bn_rep_stride + (tid < BN ? 0 : a.bn_sq_off);
                    acc_add<ACC_GRAD>(&ap[n0 + c], sum);
                }
            }
        }
    }
    if (a.stats && tid < BN && n0 + tid < a.NOUT) {
        // global float atomics to one address serialise at the memory side: spread the row tiles over replicas
        // (integer atomics to one address serialise at the memory side like float ones: the replicas stay)
        acc_t* sp = a.stats + (long)(tm % a.stats_replicas) * a.stats_rep_stride;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s1 += sRed[w * 2 * BN + tid]; s2 += sRed[w * 2 * BN + BN + tid]; }
        acc_add<ACC_STAT>(&sp[n0 + tid], s1);
        acc_add<ACC_STAT>(&sp[a.stats_sq_off + n0 + tid], s2);
    }
}

// KS > 1: in-launch split-K.  The workgroup has KS groups of 256 threads; group g multiplies k tiles g, g+KS, .. with its own
// LDS staging buffers, the partial accumulators are summed through LDS into group 0, which runs the epilogue.  The layers
// on the 17x17 / 8x8 / 3x3 maps have a few hundred workgroups and 9-36 k tiles each: their run time is the length of that
// serial chain, and a CU has wave slots to spare.
// MODE: 0 = ordinary, 1 = normalise-on-load operand (NORM), 2 = sibling sources for a 1x1 data gradient (MULTI).  The extra
// descriptors of mode 2 cost 16 SGPRs: as a run-time option they pushed every 1x1 kernel into SGPR spills.
template <typename T, int BM, int BN, int WM, int WN, int DEPTH, int KS, bool PLAIN, int MODE>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a, const int bid) {
    constexpr bool NORM = MODE == 1, MULTI = MODE == 2;
    static_assert(KS == 1 || MODE == 0, "normalise-on-load / sibling sources are built for KS == 1 only");
    static_assert(!MULTI || PLAIN, "sibling sources are 1x1 layers");
    constexpr int GT = 64 * WM * WN;      // threads per k group: 256 (4 waves) or 512 (8 waves: the 256-row tiles)
    constexpr int RPP = GT / 8;           // tile rows one pass of the group's loads covers
    constexpr int NT = GT * KS;
    constexpr int BK = 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MREP = TM / 16, NREP = TN / 16;
    constexpr int AP = BM / RPP, BP = BN / RPP;
    static_assert(AP >= 1 && BP >= 1 && (KS == 1 || GT == 256), "tile too small for the thread count / split-K is built for 4-wave groups");
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE_BYTES = 2 * (A_BYTES + B_BYTES);
    constexpr int CLD = BN + 4;
    constexpr int C_BYTES = BM * CLD * 4;
    constexpr int MAIN_BYTES = KS * STAGE_BYTES > C_BYTES ? KS * STAGE_BYTES : C_BYTES;
    typedef typename LP<T>::vec8 vec8;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FN_PHASE_CLOCK(ph0);
    {   // The argument block (kernel arguments, or this member's record of a grouped launch) spans eight 64-byte lines of the
        // scalar cache and its fields are read where they are first used: a chain of cold misses, one per line, spread over
        // the prologue (and, for normalise-on-load, in front of the statistics round trip).  One dword of every line is requested
        // here, all misses in flight together.
        const int* ap = reinterpret_cast<const int*>(&a);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(ConvArgs) / 64); ++k) {
            const int t = ap[k * 16];
            asm volatile("" ::"s"(t));
        }
    }
    const int grp = KS == 1 ? 0 : (int)(threadIdx.x / GT);   // split-K group of this thread
    unsigned char* sA = smem + grp * STAGE_BYTES;
    unsigned char* sB = sA + 2 * A_BYTES;
    float* sC = reinterpret_cast<float*>(smem);
    float* sRed = reinterpret_cast<float*>(smem + MAIN_BYTES);         // [WM <= 4][2*BN]: per-row-wave statistic partials
    int4* sT = reinterpret_cast<int4*>(smem + MAIN_BYTES + 8 * BN * 4);  // tap table [ceil(KTOT/64)*8] (general convolutions)

    const int tid = threadIdx.x, gtid = tid % GT, lane = tid & 63, wave = gtid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int cls = 0, qy = 0, qx = 0, cls_M = a.M, ktot = a.KTOT, tm, tn, cny = 0, cnx = 0;
    if (!PLAIN && a.s2) {
        // static indices only: in the grouped kernel `a` lives in registers and a dynamic index would send it to scratch
        const int t = xcd_remap(bid, a.total_tiles);
        int cp = 0;
        if (t >= a.cp1) { cls = 1; cp = a.cp1; }
        if (t >= a.cp2) { cls = 2; cp = a.cp2; }
        if (t >= a.cp3) { cls = 3; cp = a.cp3; }
        const int lt = t - cp;
        rcp_divmod(lt, a.tiles_n, tm, tn);
        qy = cls >> 1;
        qx = cls & 1;
        // pixels per image column / row of the class, recomputed (plan_tiles has the same formula).  As eight argument fields selected
        // by the class they became an indexed load from a STACK copy of the arguments: 40 bytes of scratch in every general
        // (non-1x1) variant of this kernel, stride-2 data gradient or not, and a scratch set-up at every wave launch.
        const int y0 = (qy - a.offy) & 1, x0 = (qx - a.offx) & 1;       // first pixel of the class
        cny = a.PH > y0 ? (a.PH - y0 + 1) >> 1 : 0;
        cnx = a.PW > x0 ? (a.PW - x0 + 1) >> 1 : 0;
        cls_M = (a.M / (a.PH * a.PW)) * cny * cnx;
        ktot = ((a.KH - qy + 1) >> 1) * ((a.KW - qx + 1) >> 1) * a.CS;
    } else {
        const int tile = xcd_remap(bid, a.tiles_m * a.tiles_n);
        rcp_divmod(tile, a.tiles_n, tm, tn);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int ntiles_k = MULTI ? a.nt_total : (ktot + BK - 1) / BK;
    int* sRow = reinterpret_cast<int*>(sT + (PLAIN ? 0 : ((a.KTOT + BK - 1) / BK) * 8));   // [BM] output pixel of every tile row (class mode)
    float* sNs = reinterpret_cast<float*>(sRow + BM);   // NORM: [CS] scale, [CS] shift of the source channels
    float* sNh = sNs + a.CS;

    // Operand addressing.  Both operands are read with BUFFER loads (32-bit byte offsets against a resource descriptor):
    // an offset at or beyond num_records returns zeros without touching memory, so padding, ragged rows / columns and the
    // K tail cost no select and no mask, and the loads are unconditional (the compiler can count what is in flight and
    // keeps DEPTH stages outstanding instead of draining the queue at every stage).  OOB + anything this kernel adds to it
    // stays >= 2^30 > num_records (the host checks the extents).
    //   A row i of this thread : ry/rx = source coordinates of tap (0,0), rbyte = byte offset of that pixel (or OOB)
    //   tap-table entry (kt,kg): x = dy<<24 | dx<<16 | c, y = byte delta of the tap (or OOB), z = byte offset of the weight column
    // Class mode (stride-2 dgrad): coordinates are in units of 2 source pixels; parity is guaranteed by the class.
    constexpr unsigned OOB = 0x60000000u;
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.src), 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wp), 0, a.w_bytes, 0x00020000);
    // NORM with nrm_z: the workgroups of the first column tile also write the normalised operand (the activated tensor)
    const bool side_write = NORM && a.nrm_z != nullptr && tn == 0;
    // The weights of this group's first k tile are requested NOW, before the tap table and the row arithmetic: they only need the
    // column tile, and they are the colder operand (the activations were just written by the previous kernel, the weight pack was
    // last touched a step ago).  Column c of the pack at k index k sits at byte (c * KTOT + k) * 2 for the linear tap order; the
    // parity classes of a stride-2 data gradient and the sibling sources order their taps differently and load with the tile.
    const int kg = gtid & 7, r0 = gtid >> 3;
    u32x4 ra[DEPTH][AP], rb[DEPTH][BP];   // tile t lives in register stage t % DEPTH
    unsigned rmask[DEPTH];                // NORM only: A rows of a stage that hold real pixels (padding must stay zero)
    unsigned wbyte[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        const int co = n0 + r0 + RPP * j;
        wbyte[j] = co < a.NOUT ? (unsigned)co * (unsigned)a.KTOT * 2u : OOB;
    }
    const bool early_b = !MULTI && (PLAIN || !a.s2) && !(FN_IG_DBG & 16);
    {
        const int kk0 = (KS == 1 ? 0 : grp) * BK + kg * 8;
        const unsigned kb0 = (early_b && kk0 < a.KTOT) ? (unsigned)kk0 * 2u : OOB;
#pragma unroll
        for (int j = 0; j < BP; ++j) rb[0][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wbyte[j] + kb0), 0, 0);
    }
    if constexpr (!PLAIN) {
        for (int i = tid; i < ntiles_k * 8; i += NT) {
            const int e = a.s2 ? ktab_entry_s2(i, a.CS, a.KH, a.KW, qy, qx) : ktab_entry(i, a.KTOT, a.CS, a.KW);
            int4 t = make_int4(0, (int)OOB, (int)OOB, 0);
            if (e >= 0) {
                const int ky = (e >> 24) & 0xff, kx = (e >> 16) & 0xff, c = e & 0xffff;
                const int dy = a.s2 ? -((ky - qy) >> 1) : ky * a.sk, dx = a.s2 ? -((kx - qx) >> 1) : kx * a.sk;
                t.x = (int)(((unsigned)(dy & 0xff) << 24) | ((unsigned)(dx & 0xff) << 16) | (unsigned)c);
                t.y = ((dy * a.SW + dx) * a.ld_src + c) * 2;
                t.z = ((ky * a.KW + kx) * a.CS + c) * 2;
                t.w = (!a.s2 && ky + a.offy == 0 && kx + a.offx == 0) ? 1 : 0;   // centre tap: source pixel == output pixel
            }
            sT[i] = t;
        }
    }
    int ry[AP], rx[AP];
    unsigned rbyte[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + r0 + RPP * i;
        ry[i] = 0;
        rx[i] = 0;
        rbyte[i] = OOB;
        if constexpr (PLAIN) {
            if (m < a.M) rbyte[i] = (unsigned)m * (unsigned)a.ld_src * 2u;
        } else if (a.s2) {
            if (m < cls_M) {
                const int ny = cny, nx = cnx;
                int n, rem, j, ii;
                rcp_divmod(m, ny * nx, n, rem);
                rcp_divmod(rem, nx, j, ii);
                const int py = 2 * j + ((qy - a.offy) & 1), px = 2 * ii + ((qx - a.offx) & 1);   // (py + pad) & 1 == qy
                ry[i] = (py + a.offy - qy) >> 1;     // taps of the class sit at ry - (ky - qy)/2
                rx[i] = (px + a.offx - qx) >> 1;
                rbyte[i] = (unsigned)(((n * a.SH + ry[i]) * a.SW + rx[i]) * a.ld_src) * 2u;
                if (kg == 0 && grp == 0) sRow[r0 + RPP * i] = (n * a.PH + py) * a.PW + px;
            } else if (kg == 0 && grp == 0) {
                sRow[r0 + RPP * i] = -1;
            }
        } else if (m < a.M) {
            int n, rem, py, px;
            rcp_divmod(m, a.PH * a.PW, n, rem);
            rcp_divmod(rem, a.PW, py, px);
            ry[i] = py * a.so + a.offy;
            rx[i] = px * a.so + a.offx;
            rbyte[i] = (unsigned)(((n * a.SH + ry[i]) * a.SW + rx[i]) * a.ld_src) * 2u;
        }
    }
    __syncthreads();

    // DEPTH register stages: while tile kt is multiplied out of LDS, tiles kt+1 .. kt+DEPTH are loaded or in flight.
    // Most layers of this network run at <= 1-2 workgroups per CU with cold per-XCD L2s at every kernel start, so
    // nothing else hides the (MALL/HBM) load latency; small tiles have the registers to spare, large grids use DEPTH 1.
    auto load_tile = [&](int kt, u32x4 (&ra)[AP], u32x4 (&rb)[BP], unsigned& msk, const bool have_b = false) {
        unsigned mk = 0u;
        if constexpr (PLAIN) {
            if constexpr (MULTI) {   // pick the source of this k tile (uniform), same two loads per row as below
                const int s = kt >= a.t2 ? 2 : (kt >= a.t1 ? 1 : 0);
                const int tb = s == 2 ? a.t2 : (s == 1 ? a.t1 : 0);
                const int Ks = s == 2 ? a.K3 : (s == 1 ? a.K2 : a.KTOT);
                const unsigned lds = (unsigned)(s == 2 ? a.ld3 : (s == 1 ? a.ld2 : a.ld_src));
                // descriptors are rebuilt per tile from the selected pointer (scalar work): six live descriptors spill SGPRs
                const unsigned short* ps = s == 2 ? a.src3 : (s == 1 ? a.src2 : a.src);
                const unsigned short* pw = s == 2 ? a.wp3 : (s == 1 ? a.wp2 : a.wp);
                const int bs = s == 2 ? a.src3_bytes : (s == 1 ? a.src2_bytes : a.src_bytes);
                const int bw = s == 2 ? a.w3_bytes : (s == 1 ? a.w2_bytes : a.w_bytes);
                const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(ps), 0, bs, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(pw), 0, bw, 0x00020000);
                const int kk = (kt - tb) * BK + kg * 8;
                const unsigned kb = kk < Ks ? (unsigned)kk * 2u : OOB;
#pragma unroll
                for (int i = 0; i < AP; ++i) {
                    const unsigned m2 = (unsigned)(m0 + r0 + RPP * i) * 2u;
                    const unsigned off = rbyte[i] < 0x40000000u ? m2 * lds + kb : OOB;
                    ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsa, (int)off, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < BP; ++j) {
                    const unsigned co2 = (unsigned)(n0 + r0 + RPP * j) * 2u;
                    const unsigned off = wbyte[j] < 0x40000000u ? co2 * (unsigned)Ks + kb : OOB;
                    rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)off, 0, 0);
                }
            } else {
            const int kk = kt * BK + kg * 8;
            const unsigned kb = kk < a.KTOT ? (unsigned)kk * 2u : OOB;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, (int)(rbyte[i] + kb), 0, 0);
                if constexpr (NORM) mk |= ((rbyte[i] + kb) < 0x40000000u ? 1u : 0u) << i;
            }
            if (!have_b) {
#pragma unroll
                for (int j = 0; j < BP; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wbyte[j] + kb), 0, 0);
            }
            }
        } else {
            const int4 t = sT[kt * 8 + kg];
            if (a.nocheck) {   // VALID forward convolution (uniform): every tap of a real row is inside the source, one add per chunk
#pragma unroll
                for (int i = 0; i < AP; ++i) {
                    const unsigned off = rbyte[i] + (unsigned)t.y;
                    ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, (int)off, 0, 0);
                    if constexpr (NORM) mk |= (off < 0x40000000u ? 1u : 0u) << i;
                }
            } else {
            const int dy = t.x >> 24, dx = (int)((unsigned)t.x << 8) >> 24;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const bool ok = (unsigned)(ry[i] + dy) < (unsigned)a.SH && (unsigned)(rx[i] + dx) < (unsigned)a.SW;
                const unsigned off = ok ? rbyte[i] + (unsigned)t.y : OOB;
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, (int)off, 0, 0);
                if constexpr (NORM) mk |= (off < 0x40000000u ? 1u : 0u) << i;
            }
            }
            if (!have_b) {
#pragma unroll
                for (int j = 0; j < BP; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wbyte[j] + (unsigned)t.z), 0, 0);
            }
        }
        msk = mk;
    };
    auto store_tile = [&](int buf, const u32x4 (&ra)[AP], const u32x4 (&rb)[BP], const unsigned msk, const int kt) {
        float nsc[8], nsh[8];
        unsigned zoff = 0u;       // NORM side write: byte offset of this thread's chunk relative to its row's pixel
        bool zw = false;
        if constexpr (NORM) {   // the 8 source channels of this thread's chunk of tile kt
            int c = kt * BK + kg * 8;
            if constexpr (!PLAIN) {
                const int4 te = sT[kt * 8 + kg];
                c = te.x & 0xffff;
                zoff = (unsigned)te.y;
                zw = side_write && te.w != 0;
            } else {
                zoff = (unsigned)c * 2u;
                zw = side_write;
            }
            if (msk) {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sNs + c), s1 = *reinterpret_cast<const f32x4*>(sNs + c + 4);
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(sNh + c), h1 = *reinterpret_cast<const f32x4*>(sNh + c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { nsc[e] = s0[e]; nsc[4 + e] = s1[e]; nsh[e] = h0[e]; nsh[4 + e] = h1[e]; }
            }
        }
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int r = r0 + RPP * i;
            u32x4 v = ra[i];
            if constexpr (NORM) {
                if (msk & (1u << i)) {
                    float f[8];
                    unpack8<T>(v, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], nsc[e], nsh[e]), 0.f);
                    v = pack8<T>(f);
                    // a real (in-range) chunk of the centre tap is element (output pixel, channel) of the source itself.
                    // The clamped tail iterations restage the last tile: they store the same bytes again.
                    if (zw) *(FN_GLOBAL u32x4*)((FN_GLOBAL unsigned char*)a.nrm_z + (size_t)(rbyte[i] + zoff)) = v;
                }
            }
            *reinterpret_cast<u32x4*>(sA + buf * A_BYTES + r * 128 + ((kg ^ (r & 7)) << 4)) = v;
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int r = r0 + RPP * j;
            *reinterpret_cast<u32x4*>(sB + buf * B_BYTES + r * 128 + ((kg ^ (r & 7)) << 4)) = rb[j];
        }
    };

    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            vec8 fa[MREP], fb[NREP];
            const int chunk = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < MREP; ++i) {
                const int r = wm * TM + i * 16 + fr;
                fa[i] = *reinterpret_cast<const vec8*>(sA + buf * A_BYTES + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NREP; ++j) {
                const int r = wn * TN + j * 16 + fr;
                fb[j] = *reinterpret_cast<const vec8*>(sB + buf * B_BYTES + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) acc[i][j] = LP<T>::mfma(fa[i], fb[j], acc[i][j]);
        }
    };

    // Steady state without branches around loads (same reason as above: a conditional load makes the compiler drain the
    // queue).  Tile indices are clamped to the last tile, so the tail re-loads it (L2 hits) and up to DEPTH-1 trailing
    // iterations only move data; nothing reads what they stage.
#if FN_IG_DBG & 32
    unsigned long long ph1 = wall_clock64();
    const unsigned long long phi = ph1;     // index / tap-table prologue done, nothing loaded yet
#endif
    if (ntiles_k > 0) {
        const int last = ntiles_k - 1;
        const int n_iter = (ntiles_k + KS - 1) / KS;           // same trip count for every group (barriers are block-wide)
        auto tile_of = [&](int it) { return grp + KS * it; };   // it-th k tile of this group
#pragma unroll
        for (int d = 0; d < (FN_IG_DBG & 16 ? 0 : DEPTH); ++d) load_tile(min(tile_of(d), last), ra[d], rb[d], rmask[d], d == 0 && early_b);
        if constexpr (NORM) {
            // scale / shift of the source channels from the producer's statistic replicas -- AFTER the first operand loads have
            // been issued, so the two memory round trips overlap; (channel, replica quarter) pairs spread over all threads
            acc_t* sPart = reinterpret_cast<acc_t*>(smem + grp * STAGE_BYTES);     // staging LDS is still free: [4][CS] (<= 16 KB)
            const int CS = a.CS;
            // Every load of a trip is in flight at once (2 (channel, half) pairs x 2 replicas x 2 sums per thread; replica
            // indices are clamped, not branched on): with one replica per loop iteration the prologue was a chain of
            // nrm_replicas / 2 dependent memory round trips, ~1.2 us each, in front of every normalise-on-load launch.
            const int R = a.nrm_replicas;
            for (int i0 = tid; i0 < 2 * CS; i0 += 2 * NT) {
                acc_t s1[2] = {0, 0}, s2[2] = {0, 0};         // fixed-point replicas: integer sums, exact in any order
                for (int r0 = 0; r0 < R; r0 += 4) {
                    acc_t t1[2][2], t2[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int i = min(i0 + u * NT, 2 * CS - 1);
                        const int q = i >= CS ? 1 : 0, c = i - q * CS;
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const long rp = min(r0 + q + 2 * k, R - 1);
                            t1[u][k] = a.nrm_stats[rp * a.nrm_rep_stride + c];
                            t2[u][k] = a.nrm_stats[rp * a.nrm_rep_stride + a.nrm_sq_off + c];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int q = min(i0 + u * NT, 2 * CS - 1) >= CS ? 1 : 0;
#pragma unroll
                        for (int k = 0; k < 2; ++k)
                            if (r0 + q + 2 * k < R) { s1[u] += t1[u][k]; s2[u] += t2[u][k]; }
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = i0 + u * NT;
                    if (i < 2 * CS) {
                        const int q = i >= CS ? 1 : 0, c = i - q * CS;
                        sPart[q * CS + c] = s1[u];
                        sPart[(2 + q) * CS + c] = s2[u];
                    }
                }
            }
            const float beta0 = tid < CS ? a.nrm_beta[tid] : 0.f;      // requested before the barrier: off the chain as well
            __syncthreads();
            for (int c = tid; c < CS; c += NT) {      // same arithmetic as bn_batch_affine / bn_relu_fwd_kernel: same bits
                const acc_t* sp = sPart;
                const float s1 = acc_get<ACC_STAT>(sp[c] + sp[CS + c]);
                const float s2 = acc_get<ACC_STAT>(sp[2 * CS + c] + sp[3 * CS + c]);
                float sc, sh, mean, var;
                bn_affine_from_sums(s1, s2, a.nrm_count, a.nrm_eps, c == tid ? beta0 : a.nrm_beta[c], sc, sh, mean, var);
                sNs[c] = sc;
                sNh[c] = sh;
            }
            __syncthreads();
        }
        if (!(FN_IG_DBG & 16)) store_tile(0, ra[0], rb[0], rmask[0], min(tile_of(0), last));
        __syncthreads();
#if FN_IG_DBG & 32
        ph1 = wall_clock64();
#endif
        for (int it0 = 0; it0 < n_iter; it0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int it = it0 + d;
                // stage d held this group's tile `it`, already copied to LDS: refill it with tile it+DEPTH
                if (!(FN_IG_DBG & 1)) load_tile(min(tile_of(it + DEPTH), last), ra[d], rb[d], rmask[d]);
                if (!(FN_IG_DBG & 2)) if (tile_of(it) < ntiles_k) compute(it & 1);
                if (!(FN_IG_DBG & 4)) store_tile((it + 1) & 1, ra[(d + 1) % DEPTH], rb[(d + 1) % DEPTH], rmask[(d + 1) % DEPTH], min(tile_of(it + 1), last));
                __syncthreads();
            }
        }
    }
    if constexpr (KS > 1) {   // partial accumulators of groups 1.. -> group 0 (same thread position, conflict-free b128 rows)
        f32x4* sX = reinterpret_cast<f32x4*>(smem);
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) sX[((grp - 1) * MREP * NREP + i * NREP + j) * GT + gtid] = acc[i][j];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int g = 1; g < KS; ++g)
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int j = 0; j < NREP; ++j) acc[i][j] += sX[((g - 1) * MREP * NREP + i * NREP + j) * GT + gtid];
        }
        __syncthreads();
    }
    const bool active = grp == 0;   // group 0 owns the epilogue; the others only keep the barriers company

    // ---- epilogue (shared with the halo kernel) -----------------------------------------------------
    if (FN_IG_DBG & 8) return;
    FN_PHASE_CLOCK(ph2);
#if FN_IG_DBG & 32
    unsigned long long ep[4] = {0, 0, 0, 0};
#endif
    conv_epilogue<T, BM, BN, WM, WN, MULTI>(a, acc, smem, sRed, sRow, !PLAIN && a.s2, m0, n0, tm, active FN_EP_ARG);
#if FN_IG_DBG & 32
    const unsigned long long ph3a = wall_clock64();
    __builtin_amdgcn_s_waitcnt(0);          // stores and atomics of the epilogue acknowledged
    const unsigned long long ph3 = wall_clock64();
    if (tid == 0) {
        constexpr int tile = (BM == 128 ? 3 : BM == 64 ? 2 : 1) * 4 + (BN == 128 ? 3 : BN == 64 ? 2 : 1);
        const int kind = (int)ep[2] + (a.stats ? 6 : 0);      // epilogue path (EP_*), +6: with BatchNorm statistics
        unsigned long long* p = fn_ig_phase + ((bid & 255) * 576 + ((tile & 15) * 12 + kind) * 3 + MODE) * 16;
        atomicAdd(p + 0, 1ull);
        atomicAdd(p + 1, phi - ph0);
        atomicAdd(p + 2, ph1 - phi);
        atomicAdd(p + 3, ph2 - ph1);
        atomicAdd(p + 4, ph3 - ph2);
        atomicAdd(p + 5, (unsigned long long)ntiles_k);
        atomicAdd(p + 6, (unsigned long long)(PLAIN ? 1 : 0));
        // epilogue split: C tile in LDS | row passes | folds + atomics | acknowledgement of the stores
        atomicAdd(p + 7, ep[0] - ph2);
        atomicAdd(p + 8, ep[1] - ep[0]);
        atomicAdd(p + 9, ph3a - ep[1]);
        atomicAdd(p + 10, ph3 - ph3a);
        atomicAdd(p + 11, (unsigned long long)((a.out_f32 || a.prelu) ? 1 : 0));      // what sent workgroups down the generic path
        atomicAdd(p + 12, (unsigned long long)(a.accumulate ? 1 : 0));
        atomicAdd(p + 13, (unsigned long long)((a.mask || a.out2) ? 1 : 0));
        atomicAdd(p + 14, (unsigned long long)(a.bn_y ? 1 : 0));
        atomicAdd(p + 15, (unsigned long long)(a.resid ? 1 : 0));
    }
#endif
}

template <typename T, int BM, int BN, int WM, int WN, int DEPTH, int KS, bool PLAIN, int MODE>
__global__ __launch_bounds__(64 * WM * WN * KS) void conv_igemm_kernel(const ConvArgs a) {
    conv_igemm_body<T, BM, BN, WM, WN, DEPTH, KS, PLAIN, MODE>(a, blockIdx.x);
}

// Grouped form: one launch runs several INDEPENDENT convolutions of one tile variant (sibling inception towers, the
// same dependency level of the launch list): args[g] is layer g, prefix[g] .. prefix[g+1] its workgroups.
template <typename T, int BM, int BN, int WM, int WN, int DEPTH, int KS, bool PLAIN, int MODE>
__global__ __launch_bounds__(64 * WM * WN * KS) void conv_igemm_grouped_kernel(const ConvArgs* __restrict__ args, const int* __restrict__ prefix, int n) {
    const int bid = blockIdx.x;
    int g = 0;
    while (g + 1 < n && prefix[g + 1] <= bid) ++g;     // n is small (<= 8)
    g = __builtin_amdgcn_readfirstlane(g);             // provably wave-uniform: args[g] is fetched with scalar loads into SGPRs
    const ConvArgs a = args[g];
    conv_igemm_body<T, BM, BN, WM, WN, DEPTH, KS, PLAIN, MODE>(a, bid - prefix[g]);
}

// ------------------------------------------------------------------------------------------------------------------------
// Halo-tile convolution: stride-1 k x k layers on the large maps (stem 3x3 layers, forward and data gradient).
//
// The implicit-GEMM kernel above gathers every tap of every output pixel from global memory: a 3x3 layer moves each input
// element nine times from L2 to the CU, and on the 35^2 .. 79^2 maps that traffic (~64 B/clk/CU of L2 bandwidth), not the
// MFMA pipe, sets the time (Conv2d_2a/2b: 240-370 TFLOP/s).  Here a workgroup owns an 8 x 16 pixel output tile of one image
// and BN output channels: per 32-channel slice of the input it loads the (8+KH-1) x (16+KW-1) pixel source patch ONCE
// (64 B per pixel, hardware zero fill outside the map) and the BN x taps x 32 weights into LDS, and forms all KH*KW taps from
// LDS: the A fragment of tile row `py` at tap (dy, dx) is the 1 KiB run of 16 consecutive patch pixels starting at
// (py + dy) * PW + dx.  Rows are 64 B with the 16-byte slot XOR-ed by ((row >> 2) & 1) << 1, which makes every such run --
// at ANY start pixel -- conflict free for ds_read_b128's lane groups (brute-forced against MI355X_MICROARCH.md's table).
// One MFMA 16x16x32 consumes a whole 32-channel slice per tap; the next slice's loads are issued before the current slice
// is multiplied (issue early / write late).  The data gradient is the same kernel with the taps mirrored (sk = -1) and the
// transposed weight pack.  The epilogue is conv_epilogue (row table: tile pixel -> output pixel).
// ------------------------------------------------------------------------------------------------------------------------
template <typename T, int BN, int WM, int WN, int KH, int KW>
__global__ __launch_bounds__(256) void conv_halo_kernel(const ConvArgs a) {
    constexpr int TH = 8, TW = 16, BM = TH * TW;
    constexpr int TM = BM / WM, TN = BN / WN, MREP = TM / 16, NREP = TN / 16;
    constexpr int PHt = TH + KH - 1, PWt = TW + KW - 1, NPIX = PHt * PWt, TAPS = KH * KW;
    constexpr int PATCH_BYTES = (NPIX * 64 + 255) / 256 * 256;
    constexpr int WROWS = TAPS * BN;
    constexpr int W_BYTES = WROWS * 64;
    constexpr int STAGE_BYTES = PATCH_BYTES + W_BYTES;
    constexpr int C_BYTES = BM * (BN + 4) * 4;
    constexpr int MAIN_BYTES = STAGE_BYTES > C_BYTES ? STAGE_BYTES : C_BYTES;
    constexpr int NPL = (NPIX * 4 + 255) / 256;      // 16-byte patch chunks per thread
    constexpr int NWL = (WROWS * 4 + 255) / 256;     // 16-byte weight chunks per thread
    typedef typename LP<T>::vec8 vec8;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;
    unsigned char* sW = smem + PATCH_BYTES;
    float* sRed = reinterpret_cast<float*>(smem + MAIN_BYTES);         // [WM <= 4][2*BN]
    int* sRow = reinterpret_cast<int*>(smem + MAIN_BYTES + 8 * BN * 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;

    // tile -> (image, tile row, tile column, column tile); the column tile varies fastest: workgroups that share a patch are neighbours
    const int tiles_n = a.tiles_n;
    const int t = xcd_remap(blockIdx.x, a.total_tiles);
    int sp, tn, n, rem, ty, tx;
    rcp_divmod(t, tiles_n, sp, tn);
    rcp_divmod(sp, a.halo_ty * a.halo_tx, n, rem);
    rcp_divmod(rem, a.halo_tx, ty, tx);
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = tn * BN;
    // source coordinates of patch pixel (0,0): forward taps run down/right (sk = +1), mirrored for the data gradient
    const int sy0 = oy0 + a.offy - (a.sk > 0 ? 0 : KH - 1), sx0 = ox0 + a.offx - (a.sk > 0 ? 0 : KW - 1);

    if (tid < BM) {
        const int py = tid >> 4, px = tid & 15;
        sRow[tid] = (oy0 + py < a.PH && ox0 + px < a.PW) ? (n * a.PH + oy0 + py) * a.PW + ox0 + px : -1;
    }

    constexpr unsigned OOB = 0x60000000u;
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.src), 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wp), 0, a.w_bytes, 0x00020000);
    // per-thread load slots: byte offset at channel 0 (or OOB) and the LDS byte address, fixed for the whole kernel
    unsigned poff[NPL], woff[NWL];
    int plds[NPL], wlds[NWL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int idx = tid + 256 * i, p = idx >> 2, ch = idx & 3;
        const int pr = p / PWt, pc = p - pr * PWt;
        const int sy = sy0 + pr, sx = sx0 + pc;
        const bool ok = p < NPIX && (unsigned)sy < (unsigned)a.SH && (unsigned)sx < (unsigned)a.SW;
        poff[i] = ok ? (unsigned)(((n * a.SH + sy) * a.SW + sx) * a.ld_src + ch * 8) * 2u : OOB;
        plds[i] = p < NPIX ? p * 64 + ((ch ^ (((p >> 2) & 1) << 1)) << 4) : -1;
    }
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
        const int idx = tid + 256 * j, rw = idx >> 2, ch = idx & 3;
        const int tap = rw / BN, co = rw - tap * BN;
        const bool ok = rw < WROWS && n0 + co < a.NOUT;
        woff[j] = ok ? (unsigned)((n0 + co) * a.KTOT + tap * a.CS + ch * 8) * 2u : OOB;
        wlds[j] = rw < WROWS ? rw * 64 + ((ch ^ (((rw >> 2) & 1) << 1)) << 4) : -1;
    }

    u32x4 rp[NPL], rw_[NWL];
    auto load_slice = [&](int c0) {      // 32 input channels starting at c0; channel groups beyond CS read zeros
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int ch = (tid + 256 * i) & 3;
            const unsigned off = (c0 + ch * 8 < a.CS) ? poff[i] + (unsigned)c0 * 2u : OOB;
            rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, (int)off, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const int ch = (tid + 256 * j) & 3;
            const unsigned off = (c0 + ch * 8 < a.CS) ? woff[j] + (unsigned)c0 * 2u : OOB;
            rw_[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)off, 0, 0);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < NPL; ++i)
            if (plds[i] >= 0) *reinterpret_cast<u32x4*>(sP + plds[i]) = rp[i];
#pragma unroll
        for (int j = 0; j < NWL; ++j)
            if (wlds[j] >= 0) *reinterpret_cast<u32x4*>(sW + wlds[j]) = rw_[j];
    };

    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nslices = (a.CS + 31) >> 5;
    load_slice(0);
    for (int sl = 0; sl < nslices; ++sl) {
        __syncthreads();                     // everybody is done reading the previous slice
        store_slice();
        __syncthreads();
        if (nslices > 1) load_slice(min(sl + 1, nslices - 1) * 32);     // uniform condition; within a multi-slice loop unconditional (clamped)
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KW, kx = tap - ky * KW;
            const int pdy = a.sk > 0 ? ky : KH - 1 - ky, pdx = a.sk > 0 ? kx : KW - 1 - kx;
            vec8 fa[MREP], fb[NREP];
#pragma unroll
            for (int i = 0; i < MREP; ++i) {
                const int pix = (wm * (TM / 16) + i + pdy) * PWt + fr + pdx;       // tile row wm*TM/16 + i, 16 pixels of it
                fa[i] = *reinterpret_cast<const vec8*>(sP + pix * 64 + ((fq ^ (((pix >> 2) & 1) << 1)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NREP; ++j) {
                const int rw = tap * BN + wn * TN + j * 16 + fr;
                fb[j] = *reinterpret_cast<const vec8*>(sW + rw * 64 + ((fq ^ (((rw >> 2) & 1) << 1)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) acc[i][j] = LP<T>::mfma(fa[i], fb[j], acc[i][j]);
        }
    }
    // tile pixels outside the output map were computed from real source pixels: zero them, the statistics in the epilogue sum
    // whole accumulator columns (C layout: row = 4 * (lane >> 4) + register = pixel column, tile row = row fragment)
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const bool row_ok = oy0 + wm * MREP + i < a.PH;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = row_ok && ox0 + fq * 4 + r < a.PW;
#pragma unroll
            for (int j = 0; j < NREP; ++j) acc[i][j][r] = ok ? acc[i][j][r] : 0.f;
        }
    }
    __syncthreads();        // the C tile overlays the staging buffers
#if FN_IG_DBG & 32
    unsigned long long ep[4];
#endif
    conv_epilogue<T, BM, BN, WM, WN>(a, acc, smem, sRed, sRow, true, 0, n0, sp, true FN_EP_ARG);
}

static size_t halo_smem_bytes(int BN, int KH, int KW) {
    const int npix = (8 + KH - 1) * (16 + KW - 1);
    const int stage = (npix * 64 + 255) / 256 * 256 + KH * KW * BN * 64;
    const int cb = 128 * (BN + 4) * 4;
    return (size_t)(stage > cb ? stage : cb) + 8 * BN * 4 + 128 * 4;
}

// which layers go to the halo kernel: stride-1 3x3 (forward or data gradient) on maps of at least 30 x 30 output pixels with at
// most FN_CONV_HALO_MAXC (64) source channels, plain operand (no normalise-on-load / sibling sources).  Measured on MI355X
// (tools/dev_convbench.py halo, batch 90 / 180): Conv2d_2a 53 -> 35 / 89 -> 55 us, Conv2d_2b 61 -> 54 / 108 -> 93 us; Conv2d_4a
// (80 -> 192 channels on 35 x 35: 36 % of the 8x16 tile slots fall outside the map and the weights are re-staged per slice) is
// slower here (79 vs 65 us) and stays on the implicit-GEMM kernel.  FN_CONV_HALO=0 switches the kernel off (A/B measurements).
enum { TILE_HALO = 9000000 };   // fn_conv_desc.tile_fwd / tile_dgrad: an explicit request for the halo-tile kernel
static bool halo_capable(const ConvArgs& a) {   // what the kernel can compute at all
    return !a.s2 && !a.plain && a.so == 1 && a.dshift == 0 && a.KH == 3 && a.KW == 3 && a.CS % 8 == 0 && !a.nrm_stats && a.nt_total == 0;
}
static bool halo_eligible(const ConvArgs& a) {
    if (a.tile == TILE_HALO) return halo_capable(a);   // kernel tests cover it beyond the sizes where it wins
    static const int enabled = getenv("FN_CONV_HALO") ? atoi(getenv("FN_CONV_HALO")) : 1;
    static const int maxc = getenv("FN_CONV_HALO_MAXC") ? atoi(getenv("FN_CONV_HALO_MAXC")) : 64;
    return enabled && halo_capable(a) && a.CS <= maxc && a.PH >= 30 && a.PW >= 30 &&
           a.tile == 0;   // an explicit tile (fn_conv_desc.tile_*) asks for the implicit-GEMM kernel
}
static int halo_bn(const ConvArgs& a) { return a.NOUT <= 32 ? 32 : (a.NOUT <= 48 || (a.NOUT > 64 && a.NOUT <= 96) ? 32 : 64); }

template <typename T, int BN, int WM, int WN, int KH, int KW> static int launch_halo_p(ConvArgs a, hipStream_t st) {
    a.halo_ty = cdiv(a.PH, 8);
    a.halo_tx = cdiv(a.PW, 16);
    a.tiles_n = cdiv(a.NOUT, BN);
    a.tiles_m = 0;
    const long total = (long)(a.M / (a.PH * a.PW)) * a.halo_ty * a.halo_tx * a.tiles_n;
    FN_REQUIRE(total < (1L << 24), "conv_halo: too many tiles");
    a.total_tiles = (int)total;
    const size_t smem = halo_smem_bytes(BN, KH, KW);
    auto kern = conv_halo_kernel<T, BN, WM, WN, KH, KW>;
    static LdsOptIn lds_ok;   // one per instantiation
    if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), lds_ok, "conv_halo")) return rc;
    hipLaunchKernelGGL(kern, dim3(a.total_tiles), dim3(256), smem, st, a);
    return check_launch("conv_halo");
}

template <typename T> static int launch_halo(const ConvArgs& a, hipStream_t st) {
    return halo_bn(a) == 32 ? launch_halo_p<T, 32, 4, 1, 3, 3>(a, st) : launch_halo_p<T, 64, 2, 2, 3, 3>(a, st);
}

// tiles of a launch; in class mode (stride-2 dgrad) every parity class has its own row tiles
static void plan_tiles(ConvArgs& a, int BM, int BN) {
    a.tiles_n = cdiv(a.NOUT, BN);
    if (!a.s2) {
        a.tiles_m = cdiv(a.M, BM);
        a.total_tiles = a.tiles_m * a.tiles_n;
        return;
    }
    const int nimg = a.M / (a.PH * a.PW);
    int ny[4], nx[4], pre[5];
    int t = 0;
    for (int c = 0; c < 4; ++c) {
        const int qy = c >> 1, qx = c & 1;
        const int y0 = (qy - a.offy) & 1, x0 = (qx - a.offx) & 1;       // first pixel of the class
        ny[c] = a.PH > y0 ? (a.PH - y0 + 1) / 2 : 0;
        nx[c] = a.PW > x0 ? (a.PW - x0 + 1) / 2 : 0;
        pre[c] = t;   // a class without taps still owns pixels: they receive zeros (K = 0) and must be written
        t += cdiv(nimg * ny[c] * nx[c], BM) * a.tiles_n;
    }
    a.cp1 = pre[1]; a.cp2 = pre[2]; a.cp3 = pre[3];
    a.tiles_m = 0;
    a.total_tiles = t;
}

static size_t conv_smem_bytes(int BM, int BN, int KTOT, int plain, int norm_channels, int ks) {
    const int STAGE = ks * 2 * (BM * 128 + BN * 128);
    const int CB = BM * (BN + 4) * 4;
    return (size_t)(STAGE > CB ? STAGE : CB) + 8 * BN * 4 + (plain ? 0 : (size_t)cdiv(KTOT, 64) * 8 * 16) + (size_t)BM * 4 +
           (size_t)norm_channels * 8;
}

template <typename T, int BM, int BN, int WM, int WN, int DEPTH, int KS, bool PLAIN, int MODE>
static int launch_conv_grouped_p(const ConvArgs* dev_args, const int32_t* dev_prefix, int n, int total, size_t smem, hipStream_t st) {
    auto kern = conv_igemm_grouped_kernel<T, BM, BN, WM, WN, DEPTH, KS, PLAIN, MODE>;
    static LdsOptIn lds_ok;   // one per instantiation
    if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), lds_ok, "conv_igemm_grouped")) return rc;
    hipLaunchKernelGGL(kern, dim3(total), dim3(64 * WM * WN * KS), smem, st, dev_args, dev_prefix, n);
    return check_launch("conv_igemm_grouped");
}

template <typename T, int BM, int BN, int WM, int WN, int DEPTH, int KS, bool PLAIN, int MODE>
static int launch_conv_p(const ConvArgs& a0, hipStream_t st) {
    ConvArgs a = a0;
    plan_tiles(a, BM, BN);
    const size_t smem = conv_smem_bytes(BM, BN, a.KTOT, PLAIN ? 1 : 0, MODE == 1 ? a.CS : 0, KS);
    if (smem > 160 * 1024) {
        set_error("conv: K=%d needs %zu B of LDS (>160 KiB)", a.KTOT, smem);
        return FN_EUNSUPPORTED;
    }
    auto kern = conv_igemm_kernel<T, BM, BN, WM, WN, DEPTH, KS, PLAIN, MODE>;
    static LdsOptIn lds_ok;   // one per instantiation
    if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), lds_ok, "conv_igemm")) return rc;
    hipLaunchKernelGGL(kern, dim3(a.total_tiles), dim3(64 * WM * WN * KS), smem, st, a);
    return check_launch("conv_igemm");
}

template <typename T, int BM, int BN, int WM, int WN, int DEPTH, int KS>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
    if constexpr (KS == 1) {
        if (a.nt_total > 0) return launch_conv_p<T, BM, BN, WM, WN, DEPTH, 1, true, 2>(a, st);
        if (a.nrm_stats)
            return a.plain ? launch_conv_p<T, BM, BN, WM, WN, DEPTH, 1, true, 1>(a, st) : launch_conv_p<T, BM, BN, WM, WN, DEPTH, 1, false, 1>(a, st);
    }
    return a.plain ? launch_conv_p<T, BM, BN, WM, WN, DEPTH, KS, true, 0>(a, st) : launch_conv_p<T, BM, BN, WM, WN, DEPTH, KS, false, 0>(a, st);
}

// Tile choice.  BN: smallest padded width, ties -> larger tile.  BM: the largest of {128, 64, 32} that still gives
// >= 2 workgroups per CU (these launches are latency-bound: occupancy first), else the smallest (most workgroups).
static void choose_conv_tile_auto(int M, int NOUT, int& bm, int& bn) {
    bn = 128;
    long best = (long)cdiv(NOUT, 128) * 128;
    for (int c : {64, 32}) {
        const long w = (long)cdiv(NOUT, c) * c;
        if (w < best) { best = w; bn = c; }
    }
    static const int minb = getenv("FN_CONV_MINBLOCKS") ? atoi(getenv("FN_CONV_MINBLOCKS")) : 384;   // tuning aid
    bm = 32;
    for (int c : {128, 64}) {
        if ((long)cdiv(M, c) * cdiv(NOUT, bn) >= minb) { bm = c; break; }
    }
    // narrow the N tile as well when even 32-row tiles leave most CUs idle
    while (bm == 32 && bn > 32 && (long)cdiv(M, 32) * cdiv(NOUT, bn) < minb) bn >>= 1;
}

static void choose_conv_tile(int M, int NOUT, int forced, int& bm, int& bn) {
    if (forced > 0) { bm = forced / 1000; bn = forced % 1000; return; }   // validated by check_tile
    choose_conv_tile_auto(M, NOUT, bm, bn);
}

static bool valid_tile(int t) {
    if (t == 0 || t == TILE_HALO) return true;
    const int bm = t / 1000, bn = t % 1000;
    return (bm == 128 || bm == 64 || bm == 32) && (bn == 128 || bn == 64 || bn == 32);
}

// In-launch split-K factor for the 32-row tiles: long k chains on few workgroups (see conv_igemm_body).
static int choose_conv_ks(int M, int NOUT, int KTOT, int bm, int bn) {
    static const int force = getenv("FN_CONV_KS") ? atoi(getenv("FN_CONV_KS")) : 0;   // tuning aid
    if (bm == 64) {   // 64-row tiles on grids of at most one workgroup per CU (block17 1x7 / 7x1, block35 3x3 at batch 90: 180 workgroups, 9-14 k
                      // tiles): a second wave per SIMD halves the k chain -- 13.6 -> 10.2 us per launch, step 7.06 -> 7.01 ms; larger grids
                      // (FN_CONV_KS64=400 / 640) and 128-wide tiles measured no gain
        static const int ks64 = getenv("FN_CONV_KS64") ? atoi(getenv("FN_CONV_KS64")) : 256;   // largest grid that splits (0: never)
        // four groups (1024 threads): k chains of >= 12 tiles on at most one workgroup per CU
        static const int ks64_4 = getenv("FN_CONV_KS64_4") ? atoi(getenv("FN_CONV_KS64_4")) : 0;   // largest grid that splits four ways (0: never)
        const long grid = (long)cdiv(M, 64) * cdiv(NOUT, bn);
        if (force != 1 && force != 2 && bn <= 64 && cdiv(KTOT, 64) >= 12 && grid <= ks64_4) return 4;
        return (force != 1 && bn <= 64 && cdiv(KTOT, 64) >= 8 && grid <= ks64) ? 2 : 1;
    }
    if (bm != 32) return 1;
    const long blocks = (long)cdiv(M, 32) * cdiv(NOUT, bn);
    const int ntk = cdiv(KTOT, 64);
    // measured (tools/dev_ksweep.py): the chain costs ~0.16 us per k tile only while a CU holds one workgroup; with 3+
    // workgroups per CU the loop is issue-bound and splitting K just adds the reduction
    static const int lim2 = getenv("FN_CONV_KS32_2") ? atoi(getenv("FN_CONV_KS32_2")) : 128;   // tuning aids: largest grids that split
    static const int lim4 = getenv("FN_CONV_KS32_4") ? atoi(getenv("FN_CONV_KS32_4")) : 64;
    int ks = 1;
    if (ntk >= 8 && blocks <= lim2) ks = 2;
    if (ntk >= 16 && blocks <= lim4 && bn <= 64) ks = 4;
    if (force == 1) ks = 1;
    if (force == 2 && ks > 2) ks = 2;
    return ks;
}

// tile variants: BM, BN, waves (M x N), register stages, split-K groups.  The body is written for 64 * WM * WN threads per k group;
// an 8-wave 256x128 variant (512 threads, 48 B of operands per MFMA clock instead of 64-96) was built and measured in round 3
// (tools/dev_stemtiles.py, 180 images f16): Conv2d_4a 156 us against 129 (128x64), Conv2d_4b 84.8 against 83.7 (128x128),
// ReductionA 3x3 78 against 68 -- one workgroup per CU runs its load / store / multiply phases strictly one after the other
// (~4 000 cycles per k tile for 1 024 cycles of MFMA work), so the tuner never picked it and the variant was removed again.
#define FN_CONV_VARIANTS(X) X(128, 128, 2, 2, 1, 1) X(128, 64, 2, 2, 2, 1) X(128, 32, 4, 1, 2, 1) X(64, 128, 1, 4, 2, 1) X(64, 64, 2, 2, 2, 1) \
    X(64, 32, 2, 2, 2, 1) X(32, 128, 1, 4, 4, 1) X(32, 64, 1, 4, 4, 1) X(32, 32, 2, 2, 4, 1)                                           \
    X(32, 128, 1, 4, 4, 2) X(32, 64, 1, 4, 4, 2) X(32, 64, 1, 4, 4, 4) X(32, 32, 2, 2, 4, 2) X(32, 32, 2, 2, 4, 4)                       \
    X(64, 64, 2, 2, 2, 2) X(64, 32, 2, 2, 2, 2) X(64, 64, 2, 2, 2, 4) X(64, 32, 2, 2, 2, 4)

// variant code: BM*1000 + BN (+ KS*1000000 when KS > 1)
static int variant_code(int bm, int bn, int ks) { return bm * 1000 + bn + (ks > 1 ? ks * 1000000 : 0); }

template <typename T> static int dispatch_conv(const ConvArgs& a, hipStream_t st) {
    if (halo_eligible(a)) return launch_halo<T>(a, st);
    if (a.tile == TILE_HALO) {
        set_error("conv: the halo-tile kernel was requested (tile %d) for a layer it cannot run (needs 3x3, stride 1, channels %% 8 == 0)", a.tile);
        return FN_EUNSUPPORTED;
    }
    int bm, bn;
    choose_conv_tile(a.M, a.NOUT, a.tile, bm, bn);
    if (const char* f = getenv("FN_CONV_TILE")) {   // tuning aid: "BMxBN"
        int fm = 0, fnn = 0;
        if (sscanf(f, "%dx%d", &fm, &fnn) == 2) { bm = fm; bn = fnn; }
    }
    const int ks = (a.nrm_stats || a.nt_total > 0) ? 1 : choose_conv_ks(a.M, a.NOUT, a.KTOT, bm, bn);
#define FN_X(BM_, BN_, WM_, WN_, D_, KS_) \
    if (bm == BM_ && bn == BN_ && ks == KS_) return launch_conv<T, BM_, BN_, WM_, WN_, D_, KS_>(a, st);
    FN_CONV_VARIANTS(FN_X)
#undef FN_X
    set_error("conv: no tile variant %dx%d ks=%d", bm, bn, ks);
    return FN_EUNSUPPORTED;
}

template <typename T>
static int dispatch_conv_grouped(const ConvArgs* dev_args, const int32_t* dev_prefix, int n, int total, int bm, int bn, int ks, int plain,
                                 size_t smem, hipStream_t st) {
    const bool norm = (plain & 2) != 0;   // bit 1 of `plain`: every member normalises on load
    plain &= 1;
#define FN_X(BM_, BN_, WM_, WN_, D_, KS_)                                                                                                       \
    if (bm == BM_ && bn == BN_ && ks == KS_) {                                                                                                  \
        if constexpr (KS_ == 1) {                                                                                                               \
            if (norm)                                                                                                                           \
                return plain ? launch_conv_grouped_p<T, BM_, BN_, WM_, WN_, D_, 1, true, 1>(dev_args, dev_prefix, n, total, smem, st)         \
                             : launch_conv_grouped_p<T, BM_, BN_, WM_, WN_, D_, 1, false, 1>(dev_args, dev_prefix, n, total, smem, st);       \
        }                                                                                                                                       \
        return plain ? launch_conv_grouped_p<T, BM_, BN_, WM_, WN_, D_, KS_, true, 0>(dev_args, dev_prefix, n, total, smem, st)              \
                     : launch_conv_grouped_p<T, BM_, BN_, WM_, WN_, D_, KS_, false, 0>(dev_args, dev_prefix, n, total, smem, st);            \
    }
    FN_CONV_VARIANTS(FN_X)
#undef FN_X
    set_error("conv_grouped: no tile variant %dx%d ks=%d", bm, bn, ks);
    return FN_EUNSUPPORTED;
}

static int check_desc(const fn_conv_desc* d) {
    FN_REQUIRE(d != nullptr, "conv: null descriptor");
    FN_REQUIRE(d->dtype == FN_BF16 || d->dtype == FN_F16, "conv: dtype %d unsupported", d->dtype);
    FN_REQUIRE(d->Cin % 8 == 0, "conv: Cin=%d must be a multiple of 8 (pad the input channels)", d->Cin);
    FN_REQUIRE(d->ld_x % 8 == 0 && d->ld_x >= d->Cin, "conv: ld_x=%d invalid for Cin=%d", d->ld_x, d->Cin);
    FN_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride %d unsupported", d->stride);
    FN_REQUIRE(d->KH >= 1 && d->KW >= 1 && d->KH < 256 && d->KW < 256, "conv: kernel %dx%d unsupported", d->KH, d->KW);
    FN_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->OH > 0 && d->OW > 0 && d->Cout > 0, "conv: empty geometry");
    FN_REQUIRE((d->H + 2 * d->pad_h - d->KH) / d->stride + 1 == d->OH && (d->W + 2 * d->pad_w - d->KW) / d->stride + 1 == d->OW,
               "conv: OHxOW=%dx%d inconsistent with HxW=%dx%d k=%dx%d s=%d pad=%d,%d", d->OH, d->OW, d->H, d->W, d->KH, d->KW,
               d->stride, d->pad_h, d->pad_w);
    FN_REQUIRE((long)d->N * d->H * d->W < (1L << 31) / 8 && (long)d->N * d->OH * d->OW < (1L << 24) * 8L,
               "conv: tensor too large for 32-bit pixel indexing");
    return FN_OK;
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[co][kcol] += sum_m dY[m][co] * X[m @ tap(kcol)][ci(kcol)]
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    WgradOut out;   // first member: where the result goes (grouped launches; wgrad_reduce_kernel reads it through a byte stride)
    const unsigned short* x;
    const unsigned short* dy;
    float* dw;
    int M, OH, OW, H, W, Cin, Cout, KTOT, KW;
    int stride, pad_h, pad_w;
    int ld_x, ld_y;
    int chunk;  // pixels per split (multiple of 64)
    int gx, gy, splits;  // grid of this layer inside a grouped launch
    int plain;  // 1x1 stride-1: source pixel == output pixel
    float inv_ow, inv_ohw;
    int x_bytes, dy_bytes;   // extents for the buffer resource descriptors (< 2^30)
    // normalise-on-load of x (see fn_conv_desc.nrm_*)
    const acc_t* nrm_stats;
    const float* nrm_beta;
    int nrm_sq_off, nrm_replicas, nrm_rep_stride, nrm_count;
    float nrm_eps;
    // Grouped launches are DETERMINISTIC and atomic-free (out.store = 1): a layer with one split stores its tiles straight into
    // dw; a layer with several splits stores split z into slab z of out.ws and wgrad_reduce_kernel adds the slabs in order.
    // (Global float atomics run at ~1.3 TB/s at the memory side, plain stores at ~6 TB/s, and the order of atomic adds -- hence
    // the rounding of dW -- changed from run to run.)  out.store = 0: legacy single launch, atomic accumulation into dw.
};

// k-step pixel permutation shared by both operands: tile row of MFMA k index (g = lane>>4, h = half, q)
//   rho = q + 4*(g&1) + 8*h + 16*(g>>1)   -> the 8 rows a 32-lane half reads per ds_read_b64_tr_b16
//   are distinct mod 8, which with row strides of 160 B / 288 B makes the transposed reads conflict free.
template <typename T, int BMW, int BNW, bool NORM>
__device__ __forceinline__ void conv_wgrad_body(const WgradArgs& a, const int bx, const int by, const int bz) {
    constexpr int BK = 64;                   // pixels per stage
    constexpr int DEPTH = (BMW * BNW <= 64 * 64) ? 3 : (BMW * BNW <= 64 * 128 ? 2 : 1);   // register stages in flight
    constexpr int RSA = BMW * 2 + 32;        // LDS row strides in bytes (160 for 64, 288 for 128, 96 for 32)
    constexpr int RSB = BNW * 2 + 32;
    constexpr int A_BYTES = BK * RSA, B_BYTES = BK * RSB;
    constexpr int CGA = BMW / 8, CGB = BNW / 8;  // 16-B chunks per row
    constexpr int AP = BK * CGA / 256, BP = BK * CGB / 256;
    constexpr int WMW = (BMW >= 64) ? 2 : 1, WNW = 4 / WMW;
    constexpr int TM = BMW / WMW, TN = BNW / WNW;
    constexpr int MREP = TM / 16, NREP = TN / 16;
    static_assert(AP >= 1 && BP >= 1 && MREP >= 1 && NREP >= 1, "tile too small");
    typedef typename LP<T>::vec8 vec8;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                 // [2][BK][RSA]  dY tile  (rows = pixels, cols = co)
    unsigned char* sB = smem + 2 * A_BYTES;   // [2][BK][RSB]  X  tile  (rows = pixels, cols = kcol)
    float* sNs = reinterpret_cast<float*>(smem + 2 * (A_BYTES + B_BYTES));   // NORM: [BNW] scale, [BNW] shift of this tile's columns
    float* sNh = sNs + BNW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WNW, wn = wave % WNW;
    const int n0 = bx * BNW;  // kcol tile
    const int c0 = by * BMW;  // cout tile
    const int mbeg = bz * a.chunk;
    const int mend = min(a.M, mbeg + a.chunk);
    const int nst = (mend - mbeg + BK - 1) / BK;
    if (nst <= 0) return;
    if constexpr (NORM) {
        if (tid < BNW) {
            const int e = ktab_entry((n0 >> 3) + (tid >> 3), a.KTOT, a.Cin, a.KW);
            float sc = 0.f, sh = 0.f, mean, var;
            if (e >= 0) {
                const int c = (e & 0xffff) + (tid & 7);
                bn_batch_affine(a.nrm_stats, c, a.nrm_sq_off, a.nrm_replicas, a.nrm_rep_stride, a.nrm_count, a.nrm_eps, a.nrm_beta[c], sc, sh,
                                mean, var);
            }
            sNs[tid] = sc;
            sNh[tid] = sh;
        }
        __syncthreads();
    }

    // B-operand columns handled by this thread (fixed for the whole kernel)
    int bcol_c[BP], bcol_dy[BP], bcol_dx[BP], brow[BP];
    bool bcol_ok[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        const int cidx = tid + 256 * j;
        brow[j] = cidx / CGB;
        const int e = ktab_entry((n0 >> 3) + (cidx % CGB), a.KTOT, a.Cin, a.KW);
        bcol_ok[j] = e >= 0;
        bcol_dy[j] = ((e >> 24) & 0xff) - a.pad_h;
        bcol_dx[j] = ((e >> 16) & 0xff) - a.pad_w;
        bcol_c[j] = e & 0xffff;
    }
    int arow[AP], acol[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int cidx = tid + 256 * i;
        arow[i] = cidx / CGA;
        acol[i] = c0 + (cidx % CGA) * 8;
    }

    u32x4 ra[DEPTH][AP], rb[DEPTH][BP];
    unsigned bmask[DEPTH];   // NORM only: X chunks of a stage that hold real pixels (bits 8..)
    // buffer loads with hardware zero fill (see conv_igemm_body): ragged rows / columns and padding need no select
    constexpr unsigned OOB = 0x60000000u;
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.dy), 0, a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, a.x_bytes, 0x00020000);
    unsigned acolb[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) acolb[i] = acol[i] < a.Cout ? (unsigned)acol[i] * 2u : OOB;
    auto load_tile = [&](int stg, u32x4 (&ra)[AP], u32x4 (&rb)[BP], unsigned& msk) {
        unsigned mk = 0u;
        const int mb = mbeg + stg * BK;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int m = mb + arow[i];
            const unsigned off = m < mend ? (unsigned)m * (unsigned)a.ld_y * 2u + acolb[i] : OOB;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)off, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int m = mb + brow[j];
            bool ok = m < mend && bcol_ok[j];
            int pix = m;
            if (!a.plain) {
                int n, rem, oy, ox;
                fast_divmod(m, a.OH * a.OW, a.inv_ohw, n, rem);
                fast_divmod(rem, a.OW, a.inv_ow, oy, ox);
                const int iy = oy * a.stride + bcol_dy[j], ix = ox * a.stride + bcol_dx[j];
                ok = ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                pix = (n * a.H + iy) * a.W + ix;
            }
            const unsigned off = ok ? ((unsigned)pix * (unsigned)a.ld_x + (unsigned)bcol_c[j]) * 2u : OOB;
            rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 0);
            if constexpr (NORM) mk |= (ok ? 1u : 0u) << (8 + j);
        }
        msk = mk;
    };
    auto store_tile = [&](int buf, const u32x4 (&ra)[AP], const u32x4 (&rb)[BP], const unsigned msk) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int cidx = tid + 256 * i;
            *reinterpret_cast<u32x4*>(sA + buf * A_BYTES + arow[i] * RSA + (cidx % CGA) * 16) = ra[i];
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int cidx = tid + 256 * j;
            u32x4 v = rb[j];
            if constexpr (NORM) {
                if (msk & (1u << (8 + j))) {
                    const int col = (cidx % CGB) * 8;
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sNs + col), s1 = *reinterpret_cast<const f32x4*>(sNs + col + 4);
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(sNh + col), h1 = *reinterpret_cast<const f32x4*>(sNh + col + 4);
                    float f[8];
                    unpack8<T>(v, f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f[e] = fmaxf(fmaf(f[e], s0[e], h0[e]), 0.f);
                        f[4 + e] = fmaxf(fmaf(f[4 + e], s1[e], h1[e]), 0.f);
                    }
                    v = pack8<T>(f);
                }
            }
            *reinterpret_cast<u32x4*>(sB + buf * B_BYTES + brow[j] * RSB + (cidx % CGB) * 16) = v;
        }
    };

    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read addressing: lane -> (g, q, p); supplies the address of row rho(g,h,q), columns 4p..4p+3
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int rho0 = q + 4 * (g & 1) + 16 * (g >> 1);  // + 8*h + 32*ks
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            vec8 fa[MREP], fb[NREP];
#pragma unroll
            for (int i = 0; i < MREP; ++i) {
                const unsigned char* base = sA + buf * A_BYTES + (wm * TM + i * 16 + 4 * p) * 2;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + (rho0 + 32 * ks) * RSA));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + (rho0 + 8 + 32 * ks) * RSA));
                fa[i] = __builtin_bit_cast(vec8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < NREP; ++j) {
                const unsigned char* base = sB + buf * B_BYTES + (wn * TN + j * 16 + 4 * p) * 2;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + (rho0 + 32 * ks) * RSB));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + (rho0 + 8 + 32 * ks) * RSB));
                fb[j] = __builtin_bit_cast(vec8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) acc[i][j] = LP<T>::mfma(fa[i], fb[j], acc[i][j]);
        }
    };

    const int last = nst - 1;   // branch-free steady state, clamped stage index (see conv_igemm_body)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_tile(min(d, last), ra[d], rb[d], bmask[d]);
    store_tile(0, ra[0], rb[0], bmask[0]);
    __syncthreads();
    for (int s0 = 0; s0 < nst; s0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int stg = s0 + d;
            load_tile(min(stg + DEPTH, last), ra[d], rb[d], bmask[d]);
            if (stg < nst) compute(stg & 1);
            store_tile((stg + 1) & 1, ra[(d + 1) % DEPTH], rb[(d + 1) % DEPTH], bmask[(d + 1) % DEPTH]);
            __syncthreads();
        }
    }

    // C layout: col = lane&15 (kcol), row = (lane>>4)*4 + r (cout)
    float* const dst = a.out.ws ? a.out.ws + (long)bz * a.Cout * a.KTOT : a.dw;
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            const int kc = n0 + wn * TN + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c0 + wm * TM + i * 16 + g * 4 + r;
                if (co < a.Cout && kc < a.KTOT) {
                    if (a.out.store) dst[(long)co * a.KTOT + kc] = acc[i][j][r];
                    else unsafeAtomicAdd(&a.dw[(long)co * a.KTOT + kc], acc[i][j][r]);
                }
            }
        }
}

// Second stage of the grouped weight gradients: dw = slab 0 + slab 1 + ... in that order (blockIdx.y = layer; the per-layer
// records of either weight-gradient kernel start with a WgradOut and are `stride` bytes apart).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const unsigned char* __restrict__ args, int stride) {
    const WgradOut a = *reinterpret_cast<const WgradOut*>(args + (long)blockIdx.y * stride);
    if (a.ws == nullptr) return;
    const long n4 = (long)a.Cout * a.KTOT / 4;          // layer sizes are multiples of 4
    const f32x4* ws = reinterpret_cast<const f32x4*>(a.ws);
    f32x4* dw = reinterpret_cast<f32x4*>(a.dw);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 s = ws[i];
        int z = 1;
        for (; z + 8 <= a.splits; z += 8) {       // eight slab reads in flight, added in slab order
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = ws[(long)(z + k) * n4 + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; z < a.splits; ++z) s += ws[(long)z * n4 + i];
        dw[i] = s;
    }
}

template <typename T, int BMW, int BNW, bool NORM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    conv_wgrad_body<T, BMW, BNW, NORM>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Grouped form: ONE launch computes the weight gradients of many layers.  Weight gradients have no consumer before the
// optimiser, so the engine defers them to the end of backward and issues them per tile configuration: thousands of
// workgroups per launch instead of 133 launches that each fill a fraction of the 256 CUs.
// args[g] describes layer g; prefix[g] .. prefix[g+1] are its workgroups (gx * gy * splits).
template <typename T, int BMW, int BNW, bool NORM>
__global__ __launch_bounds__(256) void conv_wgrad_grouped_kernel(const unsigned char* __restrict__ args_raw, int stride, const int* __restrict__ prefix, int n) {
    const int bid = blockIdx.x;
    int lo = 0, hi = n;                    // largest g with prefix[g] <= bid
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= bid) lo = mid; else hi = mid;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);           // wave-uniform: scalar loads of the record
    const WgradArgs a = *reinterpret_cast<const WgradArgs*>(args_raw + (long)lo * stride);
    // Workgroups that share a pixel chunk (same split, all gx*gy tiles) are consecutive in the layer's logical order: inside
    // the layer give every XCD a contiguous run of it, so a chunk of X / dY is fetched into one L2 instead of all eight
    // (per layer, not per launch: whole layers on one XCD would unbalance the chip).
    const int gxy = a.gx * a.gy;
    const int local = xcd_remap(bid - prefix[lo], gxy * a.splits);
    const int bz = local / gxy, r = local - bz * gxy;
    conv_wgrad_body<T, BMW, BNW, NORM>(a, r % a.gx, r / a.gx, bz);
}

template <typename T, int BMW, int BNW> static int launch_wgrad(const WgradArgs& a, int splits, hipStream_t st) {
    constexpr int RSA = BMW * 2 + 32, RSB = BNW * 2 + 32;
    // 64x64 stages are exactly 40 KiB: four workgroups per CU.  Only normalise-on-load launches pay for the affine table.
    const size_t smem = 2 * 64 * (RSA + RSB) + (a.nrm_stats ? 2 * BNW * 4 : 0);
    dim3 grid(cdiv(a.KTOT, BNW), cdiv(a.Cout, BMW), splits);
    if (a.nrm_stats) hipLaunchKernelGGL((conv_wgrad_kernel<T, BMW, BNW, true>), grid, dim3(256), smem, st, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<T, BMW, BNW, false>), grid, dim3(256), smem, st, a);
    return check_launch("conv_wgrad");
}

static void choose_wgrad_tile(int Cout, int KTOT, int& bmw, int& bnw) {
    bmw = Cout <= 32 ? 32 : (Cout <= 64 || Cout % 128 != 0 ? 64 : 128);
    bnw = (KTOT <= 64 || (cdiv(KTOT, 128) * 128 - KTOT) > 32) ? 64 : 128;
}

// Split-K factor over pixels.  Every split adds one fp32 copy of dW through global atomics (~1.3 TB/s chip-wide,
// MI355X_MICROARCH.md) while fewer splits mean a longer serial stage chain per workgroup (~0.5 us per 64-pixel stage at
// the occupancy these launches get).  Minimise  stages(s)*0.5us + s*bytes(dW)/1.3TB/s  subject to filling the chip.
static int choose_wgrad_splits(int M, int Cout, int KTOT, int bmw, int bnw) {
    const long tiles = (long)cdiv(KTOT, bnw) * cdiv(Cout, bmw);
    const int stages = cdiv(M, 64);
    static const int stem_wgs = getenv("FN_WG_STEMWGS") ? atoi(getenv("FN_WG_STEMWGS")) : 1024;   // tuning aid
    if (stages >= 1024) {   // long chains (stem): ~1024 workgroups in total, at least 4 stages each
        int s = (int)((stem_wgs + tiles - 1) / tiles);
        if (s > stages / 4) s = stages / 4;
        return s < 1 ? 1 : s;
    }
    const double atom_us = (double)Cout * KTOT * 4.0 / 1.3e6;   // one fp32 copy of dW
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= stages && s <= 512; s = (s < 8 ? s + 1 : s + s / 4)) {
        const double waves = (double)(tiles * s) / 512.0;        // ~2 workgroups per CU resident
        const double t = cdiv(stages, s) * 0.5 * (waves > 1.0 ? waves : 1.0) + s * atom_us;
        if (t < best_t) { best_t = t; best = s; }
    }
    return best;
}

static void final_wgrad_tile(int Cout, int KTOT, int& bmw, int& bnw) {
    choose_wgrad_tile(Cout, KTOT, bmw, bnw);
    static const int big = getenv("FN_WGRAD_BIG") ? atoi(getenv("FN_WGRAD_BIG")) : 0;   // tuning aid
    if (big == 1) return;
    // small problems: prefer 64-wide tiles so that enough workgroups exist without a deep split
    if ((long)cdiv(KTOT, bnw) * cdiv(Cout, bmw) < (big == 2 ? 16 : 64)) {
        if (bmw == 128) bmw = 64;
        if (bnw == 128 && KTOT > 64) bnw = 64;
    }
}

static int plan_wgrad(WgradArgs& a, int want_splits, int bmw, int bnw, bool grouped) {
    int splits = want_splits > 0 ? want_splits : choose_wgrad_splits(a.M, a.Cout, a.KTOT, bmw, bnw);
    if (grouped && want_splits <= 0) {
        // inside a grouped launch the chip is full anyway: fewer, longer splits.  Every split adds one fp32 copy of dW through
        // global atomics, and that traffic -- not the MFMA work -- is what the launch is made of once X / dY come from L2:
        // measured 606 / 533 / 510 / 502 / 504 / 591 us for >= 8 / 16 / 32 / 48 / 64 / 96 stages per split
        static const int min_stages = getenv("FN_WG_MINSTAGES") ? atoi(getenv("FN_WG_MINSTAGES")) : 48;   // tuning aid
        const int cap = cdiv(cdiv(a.M, 64), min_stages);
        if (splits > cap) splits = cap < 1 ? 1 : cap;
    }
    a.chunk = cdiv(cdiv(a.M, splits), 64) * 64;
    splits = cdiv(a.M, a.chunk);
    a.gx = cdiv(a.KTOT, bnw);
    a.gy = cdiv(a.Cout, bmw);
    a.splits = splits;
    return splits;
}

template <typename T> static int dispatch_wgrad(WgradArgs& a, int want_splits, hipStream_t st) {
    int bmw, bnw;
    final_wgrad_tile(a.Cout, a.KTOT, bmw, bnw);
    const int splits = plan_wgrad(a, want_splits, bmw, bnw, false);
    if (bmw == 32) return bnw == 64 ? launch_wgrad<T, 32, 64>(a, splits, st) : launch_wgrad<T, 32, 128>(a, splits, st);
    if (bmw == 64) return bnw == 64 ? launch_wgrad<T, 64, 64>(a, splits, st) : launch_wgrad<T, 64, 128>(a, splits, st);
    return bnw == 64 ? launch_wgrad<T, 128, 64>(a, splits, st) : launch_wgrad<T, 128, 128>(a, splits, st);
}

}  // namespace fn

using namespace fn;

static int make_fwd_args(const fn_conv_desc* d, ConvArgs& a) {
    if (int rc = check_desc(d)) return rc;
    FN_REQUIRE(d->x && d->w && d->y, "conv_fwd: null x/w/y");
    FN_REQUIRE(d->ld_y >= d->Cout && (d->out_f32 ? d->ld_y % 4 == 0 : d->ld_y % 8 == 0), "conv_fwd: ld_y=%d invalid", d->ld_y);
    FN_REQUIRE(!d->resid || d->ld_res % 8 == 0, "conv_fwd: ld_res=%d invalid", d->ld_res);
    a = ConvArgs{};
    a.src = (const unsigned short*)d->x; a.wp = (const unsigned short*)d->w; a.out = d->y;
    a.bias = d->bias; a.stats = d->stats; a.resid = (const unsigned short*)d->resid;
    FN_REQUIRE(!d->prelu || (!d->resid && !d->relu && !d->accumulate && !d->stats), "conv_fwd: prelu excludes resid / relu / accumulate / stats");
    a.prelu = d->prelu;
    a.M = d->N * d->OH * d->OW; a.PH = d->OH; a.PW = d->OW; a.SH = d->H; a.SW = d->W; a.CS = d->Cin;
    a.NOUT = d->Cout; a.KTOT = d->KH * d->KW * d->Cin; a.KH = d->KH; a.KW = d->KW;
    a.so = d->stride; a.sk = 1; a.offy = -d->pad_h; a.offx = -d->pad_w; a.dshift = 0;
    a.ld_src = d->ld_x; a.ld_out = d->ld_y; a.ld_res = d->ld_res;
    a.relu = d->relu; a.accumulate = d->accumulate; a.out_f32 = d->out_f32; a.scale = d->scale;
    a.stats_sq_off = d->stats_sq_off;
    a.plain = (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0) ? 1 : 0;
    a.nocheck = (d->pad_h == 0 && d->pad_w == 0) ? 1 : 0;   // check_desc guarantees (OH-1)*stride + KH <= H
    a.stats_replicas = d->stats_replicas > 0 ? d->stats_replicas : 1;
    a.stats_rep_stride = d->stats_rep_stride;
    FN_REQUIRE((long)d->N * d->H * d->W * d->ld_x * 2 < (1L << 30) && (long)d->Cout * a.KTOT * 2 < (1L << 30),
               "conv_fwd: x or w exceeds the 1 GiB range of 32-bit buffer offsets");
    // the epilogue addresses y and the residual with 32-bit byte offsets as well (buffer stores / loads, below 2 GiB)
    FN_REQUIRE((long)a.M * d->ld_y * (d->out_f32 ? 4 : 2) < (1L << 31) && (!d->resid || (long)a.M * d->ld_res * 2 < (1L << 31)),
               "conv_fwd: y or the residual exceeds the 2 GiB range of 32-bit byte offsets");
    a.src_bytes = d->N * d->H * d->W * d->ld_x * 2;
    a.w_bytes = d->Cout * a.KTOT * 2;
    FN_REQUIRE(valid_tile(d->tile_fwd), "conv_fwd: tile_fwd=%d is not one of {128,64,32}x{128,64,32} or 9000000 (halo)", d->tile_fwd);
    a.tile = d->tile_fwd;
    if (d->nrm_stats) {
        FN_REQUIRE(d->nrm_beta && d->Cin <= 512 && d->nrm_count > 0 && d->nrm_eps > 0.f, "conv_fwd: normalise-on-load needs beta, Cin <= 512, count, eps");
        a.nrm_stats = d->nrm_stats; a.nrm_beta = d->nrm_beta; a.nrm_sq_off = d->nrm_sq_off;
        a.nrm_replicas = d->nrm_replicas > 0 ? d->nrm_replicas : 1; a.nrm_rep_stride = d->nrm_rep_stride;
        a.nrm_count = d->nrm_count; a.nrm_eps = d->nrm_eps;
        if (d->nrm_z) {
            FN_REQUIRE(d->stride == 1 && d->OH == d->H && d->OW == d->W, "conv_fwd: nrm_z needs stride 1 and an output map of the input's size");
            a.nrm_z = (unsigned short*)d->nrm_z;
        }
    } else {
        FN_REQUIRE(!d->nrm_z, "conv_fwd: nrm_z without nrm_stats");
    }
    return FN_OK;
}

// dX[n,iy,ix,ci] = sum_{ky,kx,co} dY[n,(iy+pad-ky)/s,(ix+pad-kx)/s,co] * Wt[ci][ky,kx][co]
static int make_dgrad_args(const fn_conv_desc* d, ConvArgs& a) {
    if (int rc = check_desc(d)) return rc;
    FN_REQUIRE(d->y && d->w && (d->dx || d->rb_dup), "conv_dgrad: null dy/wt/dx");
    FN_REQUIRE(d->Cout % 8 == 0 && d->ld_y % 8 == 0 && d->ld_y >= d->Cout, "conv_dgrad: Cout=%d ld_y=%d must be multiples of 8", d->Cout,
               d->ld_y);
    a = ConvArgs{};
    a.src = (const unsigned short*)d->y; a.wp = (const unsigned short*)d->w; a.out = d->dx;
    a.M = d->N * d->H * d->W; a.PH = d->H; a.PW = d->W; a.SH = d->OH; a.SW = d->OW; a.CS = d->Cout;
    a.NOUT = d->Cin; a.KTOT = d->KH * d->KW * d->Cout; a.KH = d->KH; a.KW = d->KW;
    a.so = 1; a.sk = -1; a.offy = d->pad_h; a.offx = d->pad_w; a.dshift = d->stride == 2 ? 1 : 0;
    a.s2 = d->stride == 2 ? 1 : 0;
    a.ld_src = d->ld_y; a.ld_out = d->ld_x; a.ld_res = 0;
    a.relu = 0; a.accumulate = d->accumulate; a.out_f32 = d->out_f32; a.scale = 1.f;
    a.plain = (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0) ? 1 : 0;
    a.stats_replicas = 1;
    FN_REQUIRE((long)d->N * d->OH * d->OW * d->ld_y * 2 < (1L << 30) && (long)d->Cin * a.KTOT * 2 < (1L << 30),
               "conv_dgrad: dy or wt exceeds the 1 GiB range of 32-bit buffer offsets");
    // the epilogue addresses dx (and everything of dx's geometry: the fused residual backward's tensors) and the forward output of
    // the fused BatchNorm backward with 32-bit byte offsets (buffer stores / loads, below 2 GiB)
    FN_REQUIRE((long)d->N * d->H * d->W * d->ld_x * (d->out_f32 ? 4 : 2) < (1L << 31) &&
                   (!d->bn_y || (long)d->N * d->H * d->W * d->ld_bn_y * 2 < (1L << 31)),
               "conv_dgrad: dx or bn_y exceeds the 2 GiB range of 32-bit byte offsets");
    a.src_bytes = d->N * d->OH * d->OW * d->ld_y * 2;
    a.w_bytes = d->Cin * a.KTOT * 2;
    FN_REQUIRE(valid_tile(d->tile_dgrad), "conv_dgrad: tile_dgrad=%d is not one of {128,64,32}x{128,64,32} or 9000000 (halo)", d->tile_dgrad);
    a.tile = d->tile_dgrad;
    if (d->dy2) {   // sibling sources
        FN_REQUIRE(a.plain && d->w2 && d->Cout2 > 0 && d->Cout2 % 8 == 0 && d->ld_y2 % 8 == 0 && d->ld_y2 >= d->Cout2 && !d->bn_y,
                   "conv_dgrad: sibling sources need a 1x1 stride-1 layer, w2 and Cout2/ld_y2 multiples of 8");
        FN_REQUIRE(!d->dy3 || (d->w3 && d->Cout3 > 0 && d->Cout3 % 8 == 0 && d->ld_y3 % 8 == 0 && d->ld_y3 >= d->Cout3),
                   "conv_dgrad: third source needs w3 and Cout3/ld_y3 multiples of 8");
        const long pix = (long)d->N * d->OH * d->OW;
        FN_REQUIRE(pix * d->ld_y2 * 2 < (1L << 30) && (!d->dy3 || pix * d->ld_y3 * 2 < (1L << 30)), "conv_dgrad: sibling dy exceeds 1 GiB");
        a.src2 = (const unsigned short*)d->dy2; a.wp2 = (const unsigned short*)d->w2; a.K2 = d->Cout2; a.ld2 = d->ld_y2;
        a.src2_bytes = (int)(pix * d->ld_y2 * 2); a.w2_bytes = d->Cin * d->Cout2 * 2;
        a.t1 = cdiv(a.KTOT, 64);
        a.t2 = a.t1 + cdiv(a.K2, 64);
        a.nt_total = a.t2;
        if (d->dy3) {
            a.src3 = (const unsigned short*)d->dy3; a.wp3 = (const unsigned short*)d->w3; a.K3 = d->Cout3; a.ld3 = d->ld_y3;
            a.src3_bytes = (int)(pix * d->ld_y3 * 2); a.w3_bytes = d->Cin * d->Cout3 * 2;
            a.nt_total = a.t2 + cdiv(a.K3, 64);
        }
    }
    if (d->rb_dup) {
        FN_REQUIRE(d->rb_dtrunk && d->rb_dbias && !d->bn_y && !d->out_f32 && d->Cin % 8 == 0 && d->ld_x % 8 == 0,
                   "conv_dgrad: fused residual backward needs rb_dtrunk, rb_dbias, a low-precision dx geometry and no fused BN reduction");
        a.out = d->rb_dtrunk; a.accumulate = d->rb_accumulate;
        a.resid = (const unsigned short*)d->rb_prev; a.ld_res = d->ld_x; a.scale = 1.f;
        a.mask = (const unsigned short*)d->rb_out; a.out2 = (unsigned short*)d->rb_dup; a.colsum = d->rb_dbias; a.scale2 = d->rb_scale;
    }
    if (d->bn_y) {
        FN_REQUIRE(!d->accumulate && !d->out_f32 && d->bn_scale && d->bn_shift && d->bn_beta && d->bn_acc && d->ld_bn_y % 8 == 0,
                   "conv_dgrad: fused BN reduction needs a sole-writer low-precision dx and all bn_* pointers");
        a.bn_y = (const unsigned short*)d->bn_y; a.bn_scale = d->bn_scale; a.bn_shift = d->bn_shift; a.bn_beta = d->bn_beta;
        a.bn_acc = d->bn_acc; a.ld_bn_y = d->ld_bn_y; a.bn_sq_off = d->bn_sq_off;
        a.bn_replicas = d->bn_replicas > 0 ? d->bn_replicas : 1; a.bn_rep_stride = d->bn_rep_stride; a.bn_relu = d->bn_relu;
    }
    return FN_OK;
}

extern "C" int fn_conv2d_fwd(const fn_conv_desc* d, void* stream) {
    ConvArgs a;
    if (int rc = make_fwd_args(d, a)) return rc;
    return d->dtype == FN_BF16 ? dispatch_conv<__bf16>(a, (hipStream_t)stream) : dispatch_conv<_Float16>(a, (hipStream_t)stream);
}

extern "C" int fn_conv2d_dgrad(const fn_conv_desc* d, void* stream) {
    ConvArgs a;
    if (int rc = make_dgrad_args(d, a)) return rc;
    return d->dtype == FN_BF16 ? dispatch_conv<__bf16>(a, (hipStream_t)stream) : dispatch_conv<_Float16>(a, (hipStream_t)stream);
}

// ---- grouped forward / data-gradient convolutions -----------------------------------------------------------------------
extern "C" int fn_conv2d_arg_bytes(void) { return (int)sizeof(ConvArgs); }

// Host-side planning for n INDEPENDENT descriptors (op 0 = fwd, 1 = dgrad) that share tile `variant`
// (= fn_conv2d_variant(desc, op)) and 1x1-ness: fills host_args / host_prefix, *smem_bytes; returns total workgroups.
extern "C" int fn_conv2d_group_build(const fn_conv_desc* descs, int n, int op, int variant, void* host_args, int32_t* host_prefix,
                                     int32_t* smem_bytes) {
    FN_REQUIRE(descs && host_args && host_prefix && smem_bytes && n > 0 && (op == 0 || op == 1), "conv_group_build: bad arguments");
    const int ks = variant >= 1000000 ? variant / 1000000 : 1;
    const int bm = variant % 1000000 / 1000, bn = variant % 1000;
    ConvArgs* out = reinterpret_cast<ConvArgs*>(host_args);
    long total = 0;
    size_t smem = 0;
    int plain0 = -1;
    for (int i = 0; i < n; ++i) {
        ConvArgs a;
        if (int rc = (op == 0 ? make_fwd_args(&descs[i], a) : make_dgrad_args(&descs[i], a))) return rc;
        int m, k;
        choose_conv_tile(a.M, a.NOUT, a.tile, m, k);
        const int ksi = a.nrm_stats ? 1 : choose_conv_ks(a.M, a.NOUT, a.KTOT, m, k);
        FN_REQUIRE(m == bm && k == bn && ksi == ks, "conv_group_build: descriptor %d dispatches to %dx%d ks=%d, group is %dx%d ks=%d", i, m, k,
                   ksi, bm, bn, ks);
        FN_REQUIRE(descs[i].dtype == descs[0].dtype, "conv_group_build: mixed dtypes");
        FN_REQUIRE(a.nt_total == 0, "conv_group_build: a data gradient with sibling sources is a launch of its own");
        const int pl = a.plain | (a.nrm_stats ? 2 : 0);
        if (plain0 < 0) plain0 = pl;
        FN_REQUIRE(pl == plain0, "conv_group_build: 1x1 / general / normalise-on-load convolutions cannot share a group");
        plan_tiles(a, bm, bn);
        host_prefix[i] = (int32_t)total;
        total += (long)a.total_tiles;
        const size_t sm = conv_smem_bytes(bm, bn, a.KTOT, a.plain, a.nrm_stats ? a.CS : 0, ks);
        if (sm > smem) smem = sm;
        out[i] = a;
    }
    FN_REQUIRE(total < (1L << 30) && smem <= 160 * 1024, "conv_group_build: group too large");
    host_prefix[n] = (int32_t)total;
    *smem_bytes = (int32_t)smem;
    return (int)total;
}

extern "C" int fn_conv2d_grouped(const void* dev_args, const int32_t* dev_prefix, int n, int total_blocks, int variant, int plain,
                                 int smem_bytes, int dtype, void* stream) {
    FN_REQUIRE(dev_args && dev_prefix && n > 0 && n <= 64 && total_blocks > 0, "conv_grouped: bad arguments");
    FN_REQUIRE(dtype == FN_BF16 || dtype == FN_F16, "dtype %d unsupported", dtype);
    const ConvArgs* a = reinterpret_cast<const ConvArgs*>(dev_args);
    const int ks = variant >= 1000000 ? variant / 1000000 : 1, bm = variant % 1000000 / 1000, bn = variant % 1000;
    return dtype == FN_BF16
               ? dispatch_conv_grouped<__bf16>(a, dev_prefix, n, total_blocks, bm, bn, ks, plain, (size_t)smem_bytes, (hipStream_t)stream)
               : dispatch_conv_grouped<_Float16>(a, dev_prefix, n, total_blocks, bm, bn, ks, plain, (size_t)smem_bytes, (hipStream_t)stream);
}

static int make_wgrad_args(const fn_conv_desc* d, WgradArgs& a) {
    if (int rc = check_desc(d)) return rc;
    FN_REQUIRE(d->x && d->y && d->dw, "conv_wgrad: null x/dy/dw");
    FN_REQUIRE(d->ld_y % 8 == 0 && d->ld_y >= d->Cout, "conv_wgrad: ld_y=%d invalid", d->ld_y);
    FN_REQUIRE((long)d->N * d->OH * d->OW < (1L << 24), "conv_wgrad: N*OH*OW must be < 2^24");
    a = WgradArgs{};
    a.x = (const unsigned short*)d->x; a.dy = (const unsigned short*)d->y; a.dw = d->dw;
    a.M = d->N * d->OH * d->OW; a.OH = d->OH; a.OW = d->OW; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
    a.KTOT = d->KH * d->KW * d->Cin; a.KW = d->KW; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
    a.ld_x = d->ld_x; a.ld_y = d->ld_y;
    a.plain = (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0) ? 1 : 0;
    a.inv_ow = 1.0f / (float)d->OW; a.inv_ohw = 1.0f / (float)(d->OH * d->OW);
    FN_REQUIRE((long)d->N * d->H * d->W * d->ld_x * 2 < (1L << 30) && (long)a.M * d->ld_y * 2 < (1L << 30),
               "conv_wgrad: x or dy exceeds the 1 GiB range of 32-bit buffer offsets");
    a.x_bytes = d->N * d->H * d->W * d->ld_x * 2;
    a.dy_bytes = a.M * d->ld_y * 2;
    if (d->nrm_stats) {
        FN_REQUIRE(d->nrm_beta && d->nrm_count > 0 && d->nrm_eps > 0.f, "conv_wgrad: normalise-on-load needs beta, count, eps");
        a.nrm_stats = d->nrm_stats; a.nrm_beta = d->nrm_beta; a.nrm_sq_off = d->nrm_sq_off;
        a.nrm_replicas = d->nrm_replicas > 0 ? d->nrm_replicas : 1; a.nrm_rep_stride = d->nrm_rep_stride;
        a.nrm_count = d->nrm_count; a.nrm_eps = d->nrm_eps;
    }
    return FN_OK;
}

extern "C" int fn_conv2d_wgrad(const fn_conv_desc* d, void* stream) {
    WgradArgs a;
    if (int rc = make_wgrad_args(d, a)) return rc;
    return d->dtype == FN_BF16 ? dispatch_wgrad<__bf16>(a, d->splits, (hipStream_t)stream)
                               : dispatch_wgrad<_Float16>(a, d->splits, (hipStream_t)stream);
}

// ---- grouped weight gradients ------------------------------------------------------------------------------------
// one record size for both weight-gradient kernels (groups of either kind use the same host / device buffers)
extern "C" int fn_conv2d_wgrad_arg_bytes(void) {
    const size_t a = sizeof(WgradArgs), b = wgrad_taps_arg_bytes();
    return (int)(((a > b ? a : b) + 15) / 16 * 16);
}

// Host-side planning: fills host_args[n * fn_conv2d_wgrad_arg_bytes()] and host_prefix[n+1] for n descriptors that all
// dispatch to `variant` (= fn_conv2d_variant(desc, 2)); returns the total number of workgroups (or a negative status).
extern "C" int fn_conv2d_wgrad_group_build(const fn_conv_desc* descs, int n, int variant, void* host_args, int32_t* host_prefix, float* ws,
                                           int64_t* ws_elems) {
    FN_REQUIRE(descs && host_args && host_prefix && ws_elems && n > 0, "wgrad_group_build: bad arguments");
    long ws_used = 0;
    const size_t rec_bytes = (size_t)fn_conv2d_wgrad_arg_bytes();
    if (variant >= WGRAD_TAPS_VARIANT) {      // tap-sharing kernel (conv_wgrad_taps.hip)
        long total = 0;
        for (int i = 0; i < n; ++i) {
            if (int rc = check_desc(&descs[i])) return rc;
            FN_REQUIRE(descs[i].dtype == descs[0].dtype, "wgrad_group_build: mixed dtypes");
            host_prefix[i] = (int32_t)total;
            const long wgs = wgrad_taps_plan(&descs[i], variant, reinterpret_cast<unsigned char*>(host_args) + i * rec_bytes, ws, &ws_used);
            if (wgs < 0) return (int)wgs;
            total += wgs;
        }
        FN_REQUIRE(total < (1L << 30), "wgrad_group_build: too many workgroups");
        host_prefix[n] = (int32_t)total;
        *ws_elems = ws_used;
        return (int)total;
    }
    const bool norm = variant >= 1000000;
    variant %= 1000000;
    const int bmw = variant / 1000, bnw = variant % 1000;
    long total = 0;
    for (int i = 0; i < n; ++i) {
        WgradArgs a;
        if (int rc = make_wgrad_args(&descs[i], a)) return rc;
        int m, k;
        final_wgrad_tile(a.Cout, a.KTOT, m, k);
        FN_REQUIRE(m == bmw && k == bnw, "wgrad_group_build: descriptor %d dispatches to %dx%d, group is %dx%d", i, m, k, bmw, bnw);
        FN_REQUIRE(descs[i].dtype == descs[0].dtype, "wgrad_group_build: mixed dtypes");
        FN_REQUIRE((a.nrm_stats != nullptr) == norm, "wgrad_group_build: descriptor %d: normalise-on-load members need a group of their own (variant + 1000000)", i);
        const int splits = plan_wgrad(a, descs[i].splits, bmw, bnw, true);
        FN_REQUIRE(((long)a.Cout * a.KTOT) % 4 == 0, "wgrad_group_build: descriptor %d: Cout*K must be a multiple of 4", i);
        a.out = WgradOut{a.dw, nullptr, a.Cout, a.KTOT, splits, 1};
        if (splits > 1) {            // slabs of this layer: [splits][Cout*KTOT]; ws == NULL on the sizing call
            a.out.ws = ws ? ws + ws_used : reinterpret_cast<float*>(16);
            ws_used += (long)splits * a.Cout * a.KTOT;
        }
        host_prefix[i] = (int32_t)total;
        total += (long)a.gx * a.gy * splits;
        *reinterpret_cast<WgradArgs*>(reinterpret_cast<unsigned char*>(host_args) + i * rec_bytes) = a;
    }
    FN_REQUIRE(total < (1L << 30), "wgrad_group_build: too many workgroups");
    host_prefix[n] = (int32_t)total;
    *ws_elems = ws_used;
    return (int)total;
}

extern "C" int fn_conv2d_wgrad_reduce(const void* dev_args, int n, void* stream) {
    FN_REQUIRE(dev_args && n > 0 && n < 65536, "wgrad_reduce: bad arguments");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const unsigned char*>(dev_args),
                       fn_conv2d_wgrad_arg_bytes());
    return check_launch("wgrad_reduce");
}

template <typename T> static int launch_wgrad_grouped(const void* args, const int32_t* prefix, int n, int total, int variant, hipStream_t st) {
    const bool norm = variant >= 1000000;     // +1000000: every member normalises x on load
    variant %= 1000000;
    const int bmw = variant / 1000, bnw = variant % 1000;
    const unsigned char* a = reinterpret_cast<const unsigned char*>(args);
    const int stride_ = fn_conv2d_wgrad_arg_bytes();
#define FN_WG(BM_, BN_)                                                                                                     \
    if (bmw == BM_ && bnw == BN_) {                                                                                         \
        const size_t sm_ = 2 * 64 * ((BM_) * 2 + 32 + (BN_) * 2 + 32) + (norm ? 2 * (BN_) * 4 : 0);                            \
        if (norm) hipLaunchKernelGGL((conv_wgrad_grouped_kernel<T, BM_, BN_, true>), dim3(total), dim3(256), sm_, st, a, stride_, prefix, n); \
        else hipLaunchKernelGGL((conv_wgrad_grouped_kernel<T, BM_, BN_, false>), dim3(total), dim3(256), sm_, st, a, stride_, prefix, n);    \
        return check_launch("conv_wgrad_grouped");                                                                          \
    }
    FN_WG(32, 64) FN_WG(32, 128) FN_WG(64, 64) FN_WG(64, 128) FN_WG(128, 64) FN_WG(128, 128)
#undef FN_WG
    set_error("wgrad_grouped: unknown variant %d", variant);
    return FN_EINVAL;
}

extern "C" int fn_conv2d_wgrad_grouped(const void* dev_args, const int32_t* dev_prefix, int n, int total_blocks, int variant, int dtype,
                                       void* stream) {
    FN_REQUIRE(dev_args && dev_prefix && n > 0 && total_blocks > 0, "wgrad_grouped: bad arguments");
    FN_REQUIRE(dtype == FN_BF16 || dtype == FN_F16, "dtype %d unsupported", dtype);
    if (variant >= WGRAD_TAPS_VARIANT) return wgrad_taps_launch(dev_args, dev_prefix, n, total_blocks, variant, dtype, (hipStream_t)stream);
    return dtype == FN_BF16 ? launch_wgrad_grouped<__bf16>(dev_args, dev_prefix, n, total_blocks, variant, (hipStream_t)stream)
                            : launch_wgrad_grouped<_Float16>(dev_args, dev_prefix, n, total_blocks, variant, (hipStream_t)stream);
}

// Which kernel instantiation a descriptor dispatches to: returns BM*1000 + BN (op 0 = fwd, 1 = dgrad) or
// BMW*1000 + BNW (op 2 = wgrad).  Lets bench.py attribute HIP-event timings to the kernel names rocprofv3 reports.
extern "C" int fn_conv2d_variant(const fn_conv_desc* d, int op) {
    if (!d || op < 0 || op > 2) return FN_EINVAL;
    int a, b;
    if (op < 2) {   // halo-tile kernel: 9000000 + BN (never grouped, never re-tiled)
        ConvArgs ca;
        if ((op == 0 ? make_fwd_args(d, ca) : make_dgrad_args(d, ca)) == FN_OK && halo_eligible(ca)) return 9000000 + halo_bn(ca);
        if ((op == 0 ? d->tile_fwd : d->tile_dgrad) == TILE_HALO) return FN_EUNSUPPORTED;   // requested, but not a layer the kernel runs
    }
    if (op == 0) {
        choose_conv_tile(d->N * d->OH * d->OW, d->Cout, valid_tile(d->tile_fwd) ? d->tile_fwd : 0, a, b);
        return variant_code(a, b, d->nrm_stats ? 1 : choose_conv_ks(d->N * d->OH * d->OW, d->Cout, d->KH * d->KW * d->Cin, a, b));
    }
    if (op == 1) {
        choose_conv_tile(d->N * d->H * d->W, d->Cin, valid_tile(d->tile_dgrad) ? d->tile_dgrad : 0, a, b);
        return variant_code(a, b, d->dy2 ? 1 : choose_conv_ks(d->N * d->H * d->W, d->Cin, d->KH * d->KW * d->Cout, a, b));
    }
    if (const int tv = wgrad_taps_variant(d)) return tv;     // k x k layers on maps of >= 32 pixels: the tap-sharing kernel
    final_wgrad_tile(d->Cout, d->KH * d->KW * d->Cin, a, b);
    return a * 1000 + b;
}
