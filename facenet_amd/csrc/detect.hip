// Device side of the MTCNN face detector behind the reference's detectors/face_detector.py:63-78 (a wrapper around the PyPI
// `mtcnn` package, whose arithmetic is restated in oracle/mtcnn_oracle.py): everything of the cascade that touches pixels or
// feature maps and is not a convolution.  The convolutions / dense layers themselves are fn_conv2d_fwd launches with the PReLU
// epilogue (fn_conv_desc.prelu).
//   * fn_area_resize_crop   zero-padded crop of the uint8 frame -> cv2.resize(..., INTER_AREA) -> (x - 127.5) / 128 ->
//                           transposed [x][y] low-precision NHWC with 8 channels (3 + zero padding): the image pyramid of
//                           stage 1 (one "crop" = the whole frame) and the 24x24 / 48x48 candidate crops of stages 2 and 3
//   * fn_maxpool2d_fwd      MaxPooling2D(k, strides=s, 'valid' | 'same') with windows clipped at the map border
//   * fn_mtcnn_candidates   softmax over the two class logits of the P-Net map, threshold, compaction of (cell, score, reg)
// HBM-bound byte / elementwise work: coalesced 16-byte accesses, no LDS needed.
#include "../../include/facenet_hip.h"
#include "common.h"

namespace fn {

// One entry list of cv2's computeResizeAreaTab for ONE destination index d (imgproc/resize.cpp): source cells [s_first, ..]
// with weights: an optional leading partial cell, full cells, an optional trailing partial cell.
struct AreaSpan {
    int lead;       // source index of the leading partial cell or -1
    float a_lead;
    int s1, s2;     // full cells s1 .. s2-1
    float a_full;
    int trail;      // source index of the trailing partial cell or -1
    float a_trail;
};

__device__ __forceinline__ AreaSpan area_span(int d, int ssize, double scale) {
#pragma clang fp contract(off)
    AreaSpan t;
    const double fsx1 = d * scale;
    const double fsx2 = fsx1 + scale;
    const double cell = fmin(scale, (double)ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = min(sx2, ssize - 1);
    sx1 = min(sx1, sx2);
    t.lead = -1; t.trail = -1; t.a_lead = 0.f; t.a_trail = 0.f;
    if (sx1 - fsx1 > 1e-3) { t.lead = sx1 - 1; t.a_lead = (float)((sx1 - fsx1) / cell); }
    t.s1 = sx1; t.s2 = sx2;
    t.a_full = (float)(1.0 / cell);
    if (fsx2 - sx2 > 1e-3) { t.trail = sx2; t.a_trail = (float)(fmin(fmin(fsx2 - sx2, 1.0), cell) / cell); }
    return t;
}

// cv2's "area mode" bilinear coordinates (used by INTER_AREA whenever one axis is enlarged)
// horizontal: the last source column takes weight 1 (fx = 0); vertical: the row pair is only clipped (resizeGeneric_Invoker)
__device__ __forceinline__ void area_linear_coord(int d, int ssize, double scale, double inv_scale, bool horizontal, int& s, float& f) {
#pragma clang fp contract(off)
    s = (int)floor(d * scale);
    f = (float)((d + 1) - (s + 1) * inv_scale);
    f = f <= 0.f ? 0.f : f - floorf(f);
    if (horizontal) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    } else {
        s = min(max(s, 0), ssize - 1);
    }
}

template <typename WT> struct Px3 { WT c[3]; };

template <typename WT>
__device__ __forceinline__ Px3<WT> frame_px(const uint8_t* __restrict__ frame, int H, int W, int y, int x) {
    Px3<WT> p;
    if (y >= 0 && y < H && x >= 0 && x < W) {
        const uint8_t* s = frame + ((long)y * W + x) * 3;
        p.c[0] = (WT)s[0]; p.c[1] = (WT)s[1]; p.c[2] = (WT)s[2];
    } else {
        p.c[0] = p.c[1] = p.c[2] = (WT)0;
    }
    return p;
}

// boxes[k] = (ox, oy, cw, ch): crop of cw x ch pixels whose pixel (0,0) is frame pixel (ox, oy); outside the frame = 0.
// WT = float + round_u8 for a uint8 source (cv2 on uint8: float accumulators, saturate_cast<uchar> = round half to even);
// WT = double for the float64 crops of stages 2 / 3 (no rounding).
template <typename T, typename WT>
__global__ __launch_bounds__(256) void area_resize_crop_kernel(const uint8_t* __restrict__ frame, int H, int W, const int32_t* __restrict__ boxes,
                                                               int n, int OH, int OW, int round_u8, unsigned short* __restrict__ out) {
#pragma clang fp contract(off)
    const long total = (long)n * OH * OW;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        // consecutive threads -> consecutive dy of one dx: the transposed store is coalesced
        const int dy = (int)(t % OH);
        const int dx = (int)((t / OH) % OW);
        const int k = (int)(t / ((long)OH * OW));
        const int ox = boxes[4 * k], oy = boxes[4 * k + 1], cw = boxes[4 * k + 2], ch = boxes[4 * k + 3];
        WT r[3] = {(WT)0, (WT)0, (WT)0};
        if (cw > 0 && ch > 0) {
            const double inv_sx = (double)OW / cw, inv_sy = (double)OH / ch;
            const double scale_x = 1.0 / inv_sx, scale_y = 1.0 / inv_sy;
            if (scale_x >= 1.0 && scale_y >= 1.0) {   // true area resampling
                const AreaSpan xs = area_span(dx, cw, scale_x), ys = area_span(dy, ch, scale_y);
                bool first = true;
                const int ny = (ys.lead >= 0) + (ys.s2 - ys.s1) + (ys.trail >= 0);
                for (int j = 0; j < ny; ++j) {
                    int sy; float beta;
                    if (ys.lead >= 0 && j == 0) { sy = ys.lead; beta = ys.a_lead; }
                    else {
                        const int jj = j - (ys.lead >= 0);
                        if (jj < ys.s2 - ys.s1) { sy = ys.s1 + jj; beta = ys.a_full; }
                        else { sy = ys.trail; beta = ys.a_trail; }
                    }
                    WT buf[3] = {(WT)0, (WT)0, (WT)0};
                    if (xs.lead >= 0) {
                        const Px3<WT> p = frame_px<WT>(frame, H, W, oy + sy, ox + xs.lead);
#pragma unroll
                        for (int c = 0; c < 3; ++c) buf[c] = buf[c] + p.c[c] * (WT)xs.a_lead;
                    }
                    for (int sx = xs.s1; sx < xs.s2; ++sx) {
                        const Px3<WT> p = frame_px<WT>(frame, H, W, oy + sy, ox + sx);
#pragma unroll
                        for (int c = 0; c < 3; ++c) buf[c] = buf[c] + p.c[c] * (WT)xs.a_full;
                    }
                    if (xs.trail >= 0) {
                        const Px3<WT> p = frame_px<WT>(frame, H, W, oy + sy, ox + xs.trail);
#pragma unroll
                        for (int c = 0; c < 3; ++c) buf[c] = buf[c] + p.c[c] * (WT)xs.a_trail;
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) r[c] = first ? (WT)beta * buf[c] : r[c] + (WT)beta * buf[c];
                    first = false;
                }
            } else {   // one axis enlarged: bilinear with the area-mode coordinates
                int sx, sy; float fx, fy;
                area_linear_coord(dx, cw, scale_x, inv_sx, true, sx, fx);
                area_linear_coord(dy, ch, scale_y, inv_sy, false, sy, fy);
                const int sx1 = min(sx + 1, cw - 1), sy1 = min(sy + 1, ch - 1);
                const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
                const Px3<WT> p00 = frame_px<WT>(frame, H, W, oy + sy, ox + sx), p01 = frame_px<WT>(frame, H, W, oy + sy, ox + sx1);
                const Px3<WT> p10 = frame_px<WT>(frame, H, W, oy + sy1, ox + sx), p11 = frame_px<WT>(frame, H, W, oy + sy1, ox + sx1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const WT h0 = p00.c[c] * (WT)a0 + p01.c[c] * (WT)a1;
                    const WT h1 = p10.c[c] * (WT)a0 + p11.c[c] * (WT)a1;
                    r[c] = h0 * (WT)b0 + h1 * (WT)b1;
                }
            }
        }
        float v[8];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            WT q = r[c];
            if (round_u8) q = (WT)fminf(fmaxf(rintf((float)q), 0.f), 255.f);
            v[c] = (float)((q - (WT)127.5) * (WT)0.0078125);
        }
#pragma unroll
        for (int c = 3; c < 8; ++c) v[c] = 0.f;
        *reinterpret_cast<u32x4*>(out + (((long)k * OW + dx) * OH + dy) * 8) = pack8<T>(v);
    }
}

// Whole-frame pyramid level in two passes -- the order cv2's resizeArea_ itself works in: every source row is first reduced
// horizontally (buf[dx] += S[sx] * alpha, table order), then the rows of one destination row are combined (sum = beta * buf,
// sum += beta * buf).  Same arithmetic, bit for bit, as area_resize_crop_kernel<T, float>, but H x OW and OH x OW threads with
// loops of <= scale entries instead of OH x OW threads with loops of scale^2 pixels (the small levels of a 1280x720 pyramid
// reduce 52 x 52 pixels per output: 130 us per level in the one-pass kernel).
__global__ __launch_bounds__(256) void area_rows_kernel(const uint8_t* __restrict__ frame, int H, int W, int OW, float* __restrict__ rows) {
#pragma clang fp contract(off)
    const long total = (long)H * OW;
    const double scale_x = 1.0 / ((double)OW / W);
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int dx = (int)(t % OW), sy = (int)(t / OW);
        const AreaSpan xs = area_span(dx, W, scale_x);
        const uint8_t* S = frame + (long)sy * W * 3;
        float buf[3] = {0.f, 0.f, 0.f};
        if (xs.lead >= 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) buf[c] = buf[c] + (float)S[xs.lead * 3 + c] * xs.a_lead;
        }
        for (int sx = xs.s1; sx < xs.s2; ++sx) {
#pragma unroll
            for (int c = 0; c < 3; ++c) buf[c] = buf[c] + (float)S[sx * 3 + c] * xs.a_full;
        }
        if (xs.trail >= 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) buf[c] = buf[c] + (float)S[xs.trail * 3 + c] * xs.a_trail;
        }
        float* o = rows + t * 3;
        o[0] = buf[0]; o[1] = buf[1]; o[2] = buf[2];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void area_cols_kernel(const float* __restrict__ rows, int H, int OH, int OW, unsigned short* __restrict__ out) {
#pragma clang fp contract(off)
    const long total = (long)OH * OW;
    const double scale_y = 1.0 / ((double)OH / H);
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int dx = (int)(t % OW), dy = (int)(t / OW);     // consecutive threads read consecutive row entries
        const AreaSpan ys = area_span(dy, H, scale_y);
        float r[3] = {0.f, 0.f, 0.f};
        bool first = true;
        const int ny = (ys.lead >= 0) + (ys.s2 - ys.s1) + (ys.trail >= 0);
        for (int j = 0; j < ny; ++j) {
            int sy; float beta;
            if (ys.lead >= 0 && j == 0) { sy = ys.lead; beta = ys.a_lead; }
            else {
                const int jj = j - (ys.lead >= 0);
                if (jj < ys.s2 - ys.s1) { sy = ys.s1 + jj; beta = ys.a_full; }
                else { sy = ys.trail; beta = ys.a_trail; }
            }
            const float* b = rows + ((long)sy * OW + dx) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) r[c] = first ? beta * b[c] : r[c] + beta * b[c];
            first = false;
        }
        float v[8];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (fminf(fmaxf(rintf(r[c]), 0.f), 255.f) - 127.5f) * 0.0078125f;
#pragma unroll
        for (int c = 3; c < 8; ++c) v[c] = 0.f;
        *reinterpret_cast<u32x4*>(out + ((long)dx * OH + dy) * 8) = pack8<T>(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2d_fwd_kernel(const unsigned short* __restrict__ x, int ld_x, unsigned short* __restrict__ y, int ld_y,
                                                            int N, int H, int W, int C, int k, int stride, int pad_h, int pad_w, int OH, int OW) {
    const int CG = C >> 3;
    const long total = (long)N * OH * OW * CG;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int cg = (int)(t % CG);
        long pix = t / CG;
        const int ox = (int)(pix % OW); pix /= OW;
        const int oy = (int)(pix % OH);
        const int n = (int)(pix / OH);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -3.0e38f;
        const int y0 = max(oy * stride - pad_h, 0), y1 = min(oy * stride - pad_h + k, H);
        const int x0 = max(ox * stride - pad_w, 0), x1 = min(ox * stride - pad_w + k, W);
        for (int iy = y0; iy < y1; ++iy)
            for (int ix = x0; ix < x1; ++ix) {
                float v[8];
                unpack8<T>(*reinterpret_cast<const u32x4*>(x + ((long)(n * H + iy) * W + ix) * ld_x + cg * 8), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        *reinterpret_cast<u32x4*>(y + ((long)(n * OH + oy) * OW + ox) * ld_y + cg * 8) = pack8<T>(m);
    }
}

// P-Net output map, one row of `ld` floats per cell: [logit0, logit1, reg0..reg3, ...].  Keras Softmax(axis=3) in fp32:
// e = exp(l - max(l)); p1 = e1 / (e0 + e1).  Cells with p1 >= threshold are appended (any order) as 8-float records
// (cell index, tag, p1, reg0..reg3, 0); `tag` names the pyramid level, so all levels of a frame share one buffer and one
// counter.  *counter counts every hit, also those beyond max_cand (the host then reruns with more room).
__global__ __launch_bounds__(256) void mtcnn_candidates_kernel(const float* __restrict__ map, long ncell, int ld, float threshold,
                                                               float* __restrict__ cand, int* __restrict__ counter, int max_cand, int tag) {
#pragma clang fp contract(off)
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < ncell; t += (long)gridDim.x * 256) {
        const float* r = map + t * ld;
        const float2 l = *reinterpret_cast<const float2*>(r);
        const float mx = fmaxf(l.x, l.y);
        const float e0 = expf(l.x - mx), e1 = expf(l.y - mx);
        const float p1 = e1 / (e0 + e1);
        if (p1 >= threshold) {
            const int slot = atomicAdd(counter, 1);
            if (slot < max_cand) {
                f32x4* o = reinterpret_cast<f32x4*>(cand + (long)slot * 8);
                o[0] = f32x4{__int_as_float((int)t), __int_as_float(tag), p1, r[2]};
                o[1] = f32x4{r[3], r[4], r[5], 0.f};
            }
        }
    }
}

// ---- greedy non-maximum suppression (the package's __nms) ----------------------------------------------------------------
// Candidates are visited in rank order (rank r = box order[n-1-r]: the host's np.argsort of the scores, best first); a visited
// box that is still alive is kept and removes every later box whose overlap ratio o fails `o <= threshold`.  The O(n^2) part --
// all pairwise ratios, in float64 with the package's operation order (no contraction) -- is one bit matrix; the inherently
// serial part is a scan that resolves 64 ranks at a time inside one wave and ORs the kept rows into the `removed` bit set.
struct NmsJobs {          // up to 16 independent jobs per launch; boxes / order / keep are concatenated in job order
    int n[16], off[16], by_min[16];
    long ws_off[16];      // in 8-byte words
    double thr[16];
};

__global__ __launch_bounds__(64) void nms_mask_kernel(const double* __restrict__ boxes_all, int ld, const int32_t* __restrict__ order_all, NmsJobs jobs,
                                                      unsigned long long* __restrict__ ws) {
#pragma clang fp contract(off)
    const int job = blockIdx.z;
    const int n = jobs.n[job], W = (n + 63) / 64;
    const int rb = blockIdx.y, cb = blockIdx.x, t = threadIdx.x;
    if (cb >= W || rb >= W || cb < rb) return;
    const double* boxes = boxes_all + (long)jobs.off[job] * ld;
    const int32_t* order = order_all + jobs.off[job];
    unsigned long long* mask = ws + jobs.ws_off[job];
    const double threshold = jobs.thr[job];
    const int by_min = jobs.by_min[job];
    __shared__ double sx1[64], sy1[64], sx2[64], sy2[64], sar[64];
    const int cj = cb * 64 + t;
    if (cj < n) {
        const double* b = boxes + (long)order[n - 1 - cj] * ld;
        sx1[t] = b[0]; sy1[t] = b[1]; sx2[t] = b[2]; sy2[t] = b[3];
        sar[t] = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
    }
    __syncthreads();
    const int i = rb * 64 + t;
    if (i >= n) return;
    const double* b = boxes + (long)order[n - 1 - i] * ld;
    const double x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
    const double area = (x2 - x1 + 1) * (y2 - y1 + 1);
    unsigned long long bits = 0;
    const int jn = min(64, n - cb * 64);
    for (int j = 0; j < jn; ++j) {
        if (cb * 64 + j <= i) continue;
        const double w = fmax(0.0, fmin(x2, sx2[j]) - fmax(x1, sx1[j]) + 1);
        const double h = fmax(0.0, fmin(y2, sy2[j]) - fmax(y1, sy1[j]) + 1);
        const double inter = w * h;
        const double o = by_min ? inter / fmin(area, sar[j]) : inter / (area + sar[j] - inter);
        if (!(o <= threshold)) bits |= 1ull << j;
    }
    mask[(long)i * W + cb] = bits;
}

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// One workgroup per job.  Ranks are resolved 64 at a time: lane l of wave 0 holds the diagonal word of rank b*64+l (which later
// ranks OF THE SAME BLOCK it removes); the 64-step dependency chain runs on the scalar unit (v_readlane with constant lanes,
// SGPR bit sets), then all 256 threads OR the kept rows into the `removed` words of the later blocks.
__global__ __launch_bounds__(256) void nms_scan_kernel(const unsigned long long* __restrict__ ws, const int32_t* __restrict__ order_all, NmsJobs jobs,
                                                       int32_t* __restrict__ keep_all, int32_t* __restrict__ n_keep) {
    extern __shared__ unsigned long long removed[];
    __shared__ unsigned long long s_kept;
    __shared__ int s_list[64];
    const int job = blockIdx.x;
    const int n = jobs.n[job], W = (n + 63) / 64;
    const unsigned long long* mask = ws + jobs.ws_off[job];
    const int32_t* order = order_all + jobs.off[job];
    int32_t* keep = keep_all + jobs.off[job];
    const int tid = threadIdx.x;
    for (int w = tid; w < W; w += 256) removed[w] = 0;
    int count = 0;   // meaningful in wave 0
    unsigned long long diag = tid < 64 && tid < n ? mask[(long)tid * W] : 0ull;      // diagonal word of block 0
    __syncthreads();
    for (int b = 0; b < W; ++b) {
        if (tid < 64) {
            const int row = b * 64 + tid;
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
            // the next block's diagonal does not depend on this block's outcome: request it now, use it after the barriers
            const int nrow = row + 64;
            if (b + 1 < W) diag = nrow < n ? mask[(long)nrow * W + b + 1] : 0ull;
            unsigned long long rem = uniform64(removed[b]);
            if (n - b * 64 < 64) rem |= ~0ull << (n - b * 64);       // ranks beyond n do not exist
            unsigned long long kept = 0;
            if (rem != ~0ull) {
#pragma unroll
                for (int t = 0; t < 64; ++t) {
                    const unsigned long long d = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dhi, t) << 32) |
                                                 (unsigned)__builtin_amdgcn_readlane((int)dlo, t);
                    if (!((rem >> t) & 1ull)) { kept |= 1ull << t; rem |= d; }
                }
            }
            if ((kept >> tid) & 1ull) {
                const int pos = __popcll(kept & ((1ull << tid) - 1ull));
                keep[count + pos] = order[n - 1 - row];
                s_list[pos] = tid;
            }
            count += __popcll(kept);
            if (tid == 0) s_kept = kept;
        }
        __syncthreads();
        // every (kept rank, later word) pair is one independent load + LDS atomic: one memory round trip per block of 64 ranks
        const int nk = __popcll(s_kept), wrem = W - b - 1;
        for (int idx = tid; idx < nk * wrem; idx += 256) {
            const int ki = idx / wrem, w = b + 1 + (idx - ki * wrem);
            const unsigned long long m = mask[(long)(b * 64 + s_list[ki]) * W + w];
            if (m) atomicOr(&removed[w], m);
        }
        __syncthreads();
    }
    if (tid == 0) n_keep[job] = count;
}

static inline int grid_of(long items, int cap = 8192) {
    long b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace fn

using namespace fn;

extern "C" int fn_area_resize_crop(const uint8_t* frame, int H, int W, const int32_t* boxes, int n, int OH, int OW, int source_is_u8, void* out,
                                   int dtype, void* stream) {
    FN_REQUIRE(frame && boxes && out && H > 0 && W > 0 && n > 0 && OH > 0 && OW > 0, "area_resize_crop: bad arguments");
    FN_REQUIRE(dtype == DT_BF16 || dtype == DT_F16, "area_resize_crop: dtype %d", dtype);
    const int grid = grid_of((long)n * OH * OW);
    hipStream_t st = (hipStream_t)stream;
    if (source_is_u8) {
        if (dtype == DT_F16) hipLaunchKernelGGL((area_resize_crop_kernel<_Float16, float>), dim3(grid), dim3(256), 0, st, frame, H, W, boxes, n, OH, OW, 1, (unsigned short*)out);
        else hipLaunchKernelGGL((area_resize_crop_kernel<__bf16, float>), dim3(grid), dim3(256), 0, st, frame, H, W, boxes, n, OH, OW, 1, (unsigned short*)out);
    } else {
        if (dtype == DT_F16) hipLaunchKernelGGL((area_resize_crop_kernel<_Float16, double>), dim3(grid), dim3(256), 0, st, frame, H, W, boxes, n, OH, OW, 0, (unsigned short*)out);
        else hipLaunchKernelGGL((area_resize_crop_kernel<__bf16, double>), dim3(grid), dim3(256), 0, st, frame, H, W, boxes, n, OH, OW, 0, (unsigned short*)out);
    }
    return check_launch("area_resize_crop");
}

extern "C" int fn_maxpool2d_fwd(const void* x, int ld_x, void* y, int ld_y, int N, int H, int W, int C, int k, int stride, int pad_h, int pad_w,
                                int OH, int OW, int dtype, void* stream) {
    FN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && ld_x % 8 == 0 && ld_y % 8 == 0 && ld_x >= C && ld_y >= C,
               "maxpool2d: bad arguments");
    FN_REQUIRE(k >= 1 && stride >= 1 && pad_h >= 0 && pad_w >= 0 && pad_h < k && pad_w < k && OH > 0 && OW > 0, "maxpool2d: bad window");
    // every window must hold at least one real element
    FN_REQUIRE((OH - 1) * stride - pad_h < H && (OW - 1) * stride - pad_w < W, "maxpool2d: output %dx%d exceeds the input %dx%d", OH, OW, H, W);
    FN_REQUIRE(dtype == DT_BF16 || dtype == DT_F16, "maxpool2d: dtype %d", dtype);
    const int grid = grid_of((long)N * OH * OW * (C / 8));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DT_F16) hipLaunchKernelGGL((maxpool2d_fwd_kernel<_Float16>), dim3(grid), dim3(256), 0, st, (const unsigned short*)x, ld_x, (unsigned short*)y, ld_y, N, H, W, C, k, stride, pad_h, pad_w, OH, OW);
    else hipLaunchKernelGGL((maxpool2d_fwd_kernel<__bf16>), dim3(grid), dim3(256), 0, st, (const unsigned short*)x, ld_x, (unsigned short*)y, ld_y, N, H, W, C, k, stride, pad_h, pad_w, OH, OW);
    return check_launch("maxpool2d");
}

extern "C" int fn_mtcnn_candidates(const float* map, long ncell, int ld, float threshold, float* cand, int32_t* counter, int max_cand, int tag,
                                   int reset_counter, void* stream) {
    FN_REQUIRE(map && cand && counter && ncell > 0 && ld >= 6 && ld % 2 == 0 && max_cand > 0, "mtcnn_candidates: bad arguments");
    FN_REQUIRE(((uintptr_t)cand & 15) == 0, "mtcnn_candidates: cand must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (reset_counter) fill_words((unsigned*)counter, 0u, 0u, 1, st);
    hipLaunchKernelGGL(mtcnn_candidates_kernel, dim3(grid_of(ncell)), dim3(256), 0, st, map, ncell, ld, threshold, cand, (int*)counter, max_cand, tag);
    return check_launch("mtcnn_candidates");
}

extern "C" int fn_nms_greedy_batch(const double* boxes, int ld, const int32_t* order, const int32_t* sizes, const double* thresholds, const int32_t* by_min,
                                   int njobs, void* workspace, long workspace_bytes, int32_t* keep, int32_t* n_keep, void* stream) {
    FN_REQUIRE(boxes && order && sizes && thresholds && by_min && workspace && keep && n_keep && njobs > 0 && ld >= 4, "nms_greedy: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    long off = 0;
    for (int j0 = 0; j0 < njobs; j0 += 16) {
        NmsJobs jobs = {};
        const int nj = njobs - j0 < 16 ? njobs - j0 : 16;
        long words = 0;
        int wmax = 0;
        for (int j = 0; j < nj; ++j) {
            const int n = sizes[j0 + j];
            FN_REQUIRE(n > 0 && n <= (1 << 19), "nms_greedy: job %d has %d boxes", j0 + j, n);
            const int W = (n + 63) / 64;
            jobs.n[j] = n; jobs.off[j] = (int)off; jobs.by_min[j] = by_min[j0 + j]; jobs.thr[j] = thresholds[j0 + j]; jobs.ws_off[j] = words;
            words += (long)n * W;
            off += n;
            wmax = W > wmax ? W : wmax;
        }
        FN_REQUIRE(words * 8 <= workspace_bytes, "nms_greedy: workspace of %ld bytes is too small (%ld needed)", workspace_bytes, words * 8);
        FN_REQUIRE(wmax <= 65535, "nms_greedy: too many boxes in one job");
        hipLaunchKernelGGL(nms_mask_kernel, dim3(wmax, wmax, nj), dim3(64), 0, st, boxes, ld, order, jobs, (unsigned long long*)workspace);
        hipLaunchKernelGGL(nms_scan_kernel, dim3(nj), dim3(256), (size_t)wmax * 8, st, (const unsigned long long*)workspace, order, jobs, keep, n_keep + j0);
    }
    return check_launch("nms_greedy");
}

extern "C" int fn_nms_greedy(const double* boxes, int ld, const int32_t* order, int n, double threshold, int by_min, void* workspace,
                             long workspace_bytes, int32_t* keep, int32_t* n_keep, void* stream) {
    return fn_nms_greedy_batch(boxes, ld, order, &n, &threshold, &by_min, 1, workspace, workspace_bytes, keep, n_keep, stream);
}

extern "C" int fn_area_resize_frame(const uint8_t* frame, int H, int W, int OH, int OW, float* rows, void* out, int dtype, void* stream) {
    FN_REQUIRE(frame && rows && out && H > 0 && W > 0 && OH > 0 && OW > 0, "area_resize_frame: bad arguments");
    FN_REQUIRE(OH <= H && OW <= W, "area_resize_frame: %dx%d -> %dx%d enlarges the uint8 frame (cv2's fixed-point bilinear path is not built)", H, W, OH, OW);
    FN_REQUIRE(dtype == DT_BF16 || dtype == DT_F16, "area_resize_frame: dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(area_rows_kernel, dim3(grid_of((long)H * OW)), dim3(256), 0, st, frame, H, W, OW, rows);
    if (dtype == DT_F16) hipLaunchKernelGGL((area_cols_kernel<_Float16>), dim3(grid_of((long)OH * OW)), dim3(256), 0, st, (const float*)rows, H, OH, OW, (unsigned short*)out);
    else hipLaunchKernelGGL((area_cols_kernel<__bf16>), dim3(grid_of((long)OH * OW)), dim3(256), 0, st, (const float*)rows, H, OH, OW, (unsigned short*)out);
    return check_launch("area_resize_frame");
}
