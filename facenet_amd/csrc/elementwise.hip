// HBM-bound kernels around the convolutions: input normalisation, BatchNorm(+ReLU) train forward /
// backward, pooling, residual backward, the fp32 embedding head.  All NHWC, 16 B (8 channels) per lane,
// channel-slice aware (ld = channel stride of the enclosing concat buffer).
#include "common.h"
#include "../../include/facenet_hip.h"

namespace fn {

// ------------------------------------------------------------------------------------------------
// ImageProcessing.call  (facenet/facenet.py:67-86)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2ord(float f) {  // order-preserving float -> uint
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// work per image (8 words): [0]=ord(max) [1]=ord(-min) [2..3]=sum [4..5]=sumsq as fixed-point acc_t (ACC_STAT: order-independent;
// sums of u8 pixels are integers and exact)   (zeroed by a kernel node first; ord(x) > 0 always)
// SRC = uint8_t (the reference's 'input' node dtype, facenet/__init__.py:16-20) or float (facenet.py:69 casts anyway)
template <typename SRC>
__global__ __launch_bounds__(256) void img_stats_kernel(const SRC* __restrict__ img, unsigned* __restrict__ work, int count) {
    const int n = blockIdx.y;
    const SRC* p = img + (long)n * count;
    float mx = -3e38f, mn = 3e38f, s = 0.f, q = 0.f;
    constexpr int PER = 16 / sizeof(SRC);
    const int nvec = count / PER;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(p + (long)i * PER);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if constexpr (sizeof(SRC) == 1) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float x = (float)((v[w] >> (8 * b)) & 0xffu);
                    mx = fmaxf(mx, x); mn = fminf(mn, x); s += x; q += x * x;
                }
            } else {
                const float x = __uint_as_float(v[w]);
                mx = fmaxf(mx, x); mn = fminf(mn, x); s += x; q += x * x;
            }
        }
    }
    if (blockIdx.x == 0)
        for (int i = nvec * PER + threadIdx.x; i < count; i += 256) {
            const float x = (float)p[i];
            mx = fmaxf(mx, x); mn = fminf(mn, x); s += x; q += x * x;
        }
    mx = wave_max(mx); mn = -wave_max(-mn); s = wave_sum(s); q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&work[8 * n + 0], f2ord(mx));
        atomicMax(&work[8 * n + 1], f2ord(-mn));
        acc_add<ACC_STAT>(reinterpret_cast<acc_t*>(&work[8 * n + 2]), s);
        acc_add<ACC_STAT>(reinterpret_cast<acc_t*>(&work[8 * n + 4]), q);
    }
}

template <typename T, typename SRC>
__global__ __launch_bounds__(256) void img_apply_kernel(const SRC* __restrict__ img, unsigned short* __restrict__ out,
                                                        const unsigned* __restrict__ work, int HW, int mode) {
    const int n = blockIdx.y;
    // same operation order as the reference so that exact cases (constant images) stay exact:
    //   mode 0: (2x - (min+max)) / max(max-min, 1e-3)                      facenet.py:72-77
    //   mode 1: (x - mean) / max(std, 1/sqrt(numel))  per_image_standardization   facenet.py:79-80
    float mul, sub, den;
    if (mode == 0) {
        const float mx = ord2f(work[8 * n + 0]), mn = -ord2f(work[8 * n + 1]);
        mul = 2.f; sub = mn + mx; den = fmaxf(mx - mn, 1e-3f);
    } else {
        const float cnt = (float)HW * 3.f;
        const float mean = acc_get<ACC_STAT>(*reinterpret_cast<const acc_t*>(&work[8 * n + 2])) / cnt;
        const float var = fmaxf(acc_get<ACC_STAT>(*reinterpret_cast<const acc_t*>(&work[8 * n + 4])) / cnt - mean * mean, 0.f);
        mul = 1.f; sub = mean; den = fmaxf(sqrtf(var), rsqrtf(cnt));
    }
    const SRC* p = img + (long)n * HW * 3;
    unsigned short* o = out + (long)n * HW * 8;
    for (int px = blockIdx.x * 256 + threadIdx.x; px < HW; px += gridDim.x * 256) {
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (mul * (float)p[px * 3 + c] - sub) / den;
        *reinterpret_cast<u32x4*>(o + (long)px * 8) = pack8<T>(v);
    }
}

// tf.image.resize(images, [size, size]) as called at facenet/facenet.py:70: bilinear, half-pixel centres, no antialias
// (TF2 defaults): src = (dst + 0.5) * in/out - 0.5, lower = max(floor(src), 0), upper = min(ceil(src), in-1).
template <typename SRC>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const SRC* __restrict__ img, float* __restrict__ out, int N, int H, int W,
                                                              int OH, int OW) {
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    const long total = (long)N * OH * OW;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int ox = (int)(t % OW);
        const int oy = (int)((t / OW) % OH);
        const int n = (int)(t / ((long)OW * OH));
        const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
        const float fy0 = floorf(fy), fx0 = floorf(fx);
        const int y0 = max((int)fy0, 0), y1 = min((int)ceilf(fy), H - 1);
        const int x0 = max((int)fx0, 0), x1 = min((int)ceilf(fx), W - 1);
        const float ly = fy - fy0, lx = fx - fx0;
        const SRC* p = img + (long)n * H * W * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v00 = (float)p[((long)y0 * W + x0) * 3 + c], v01 = (float)p[((long)y0 * W + x1) * 3 + c];
            const float v10 = (float)p[((long)y1 * W + x0) * 3 + c], v11 = (float)p[((long)y1 * W + x1) * 3 + c];
            const float top = v00 + (v01 - v00) * lx, bot = v10 + (v11 - v10) * lx;
            out[t * 3 + c] = top + (bot - top) * ly;
        }
    }
}

__global__ __launch_bounds__(256) void gather_images_kernel(const uint8_t* __restrict__ pool, const int32_t* __restrict__ idx,
                                                            uint8_t* __restrict__ out, int bytes) {
    const int i = blockIdx.y;
    const uint8_t* s = pool + (long)idx[i] * bytes;
    uint8_t* d = out + (long)i * bytes;
    const int nvec = bytes >> 4;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < nvec; k += gridDim.x * 256)
        reinterpret_cast<u32x4*>(d)[k] = reinterpret_cast<const u32x4*>(s)[k];
    if (blockIdx.x == 0)
        for (int k = (nvec << 4) + threadIdx.x; k < bytes; k += 256) d[k] = s[k];
}

// ------------------------------------------------------------------------------------------------
// BatchNormalization(center only) + ReLU, training mode.  y (raw conv output) is KEPT: the backward
// needs xhat for every element, including the ones ReLU clamps.
// ------------------------------------------------------------------------------------------------
// Grid = (row chunks, 64-channel stripes): a workgroup finalises the statistics of ITS stripe only (sum over the
// accumulator replicas the convolution epilogue spread its row tiles over), so the replica count can be high
// (same-address float atomics serialise) without every workgroup re-reading every channel.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const unsigned short* __restrict__ y, int ld_y, unsigned short* __restrict__ z,
                                                          int ld_z, int M, int C, const acc_t* __restrict__ stats, int sq_off, int reps, int rep_stride,
                                                          const float* __restrict__ beta, float* __restrict__ save_scale,
                                                          float* __restrict__ save_shift, float* __restrict__ mm, float* __restrict__ mv,
                                                          float momentum, float eps, int relu, int rows_per_block) {
    __shared__ float s_scale[64], s_shift[64];
    __shared__ acc_t s_part[4][2][64];
    const int lin_ = xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);   // rows -> XCD like the convolutions
    const int bx_ = lin_ / gridDim.y, by_ = lin_ - bx_ * gridDim.y;
    const int c0 = by_ * 64;
    // The first chunk of every thread is requested BEFORE the statistics prologue: most launches give a thread one or two
    // 16-byte chunks, so the kernel is two dependent memory round trips (statistics, then data) unless they overlap.
    const int ncg = min(8, (C - c0) >> 3);
    const int TX = ncg <= 1 ? 1 : (ncg <= 2 ? 2 : (ncg <= 4 ? 4 : 8));   // narrow stripes keep all 256 threads busy
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int r0 = bx_ * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const int c = c0 + tx * 8;
    u32x4 pre = {0u, 0u, 0u, 0u};
    if (tx < ncg && r0 + ty < r1) pre = *reinterpret_cast<const u32x4*>(y + (long)(r0 + ty) * ld_y + c);
    {   // replica sums: 4 thread groups x 64 channels, independent loads in flight (a rolled serial loop would expose
        // one memory latency per replica)
        // (the replicas are fixed-point integers: their sum is exact in any order)
        const int cc = threadIdx.x & 63, q = threadIdx.x >> 6;
        acc_t s1 = 0, s2v = 0;
        if (c0 + cc < C) {
#pragma unroll 4
            for (int rp = q; rp < reps; rp += 4) {
                s1 += stats[(long)rp * rep_stride + c0 + cc];
                s2v += stats[(long)rp * rep_stride + sq_off + c0 + cc];
            }
        }
        s_part[q][0][cc] = s1;
        s_part[q][1][cc] = s2v;
    }
    __syncthreads();
    if (threadIdx.x < 64 && c0 + threadIdx.x < C) {
        const int c = c0 + threadIdx.x;
        const float s1 = acc_get<ACC_STAT>(s_part[0][0][threadIdx.x] + s_part[1][0][threadIdx.x] + s_part[2][0][threadIdx.x] + s_part[3][0][threadIdx.x]);
        const float s2v = acc_get<ACC_STAT>(s_part[0][1][threadIdx.x] + s_part[1][1][threadIdx.x] + s_part[2][1][threadIdx.x] + s_part[3][1][threadIdx.x]);
        float rstd, shf, mean, var;
        bn_affine_from_sums(s1, s2v, M, eps, beta[c], rstd, shf, mean, var);
        s_scale[threadIdx.x] = rstd;
        s_shift[threadIdx.x] = shf;
        if (bx_ == 0) {
            save_scale[c] = rstd;
            save_shift[c] = shf;
            if (mm) {
                mm[c] = bn_moving_update(mm[c], mean, momentum);
                mv[c] = bn_moving_update(mv[c], var, momentum);  // biased variance (hazard 3)
            }
        }
    }
    __syncthreads();
    if (tx >= ncg) return;
    const __amdgpu_buffer_rsrc_t rs_z = wt_rsrc(z);
    float sc[8], sf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = s_scale[tx * 8 + e]; sf[e] = s_shift[tx * 8 + e]; }
    // four rows per trip: the loads of a trip are all in flight before the first is used (the big stem maps give a thread 8+ rows;
    // one row per trip is one exposed memory latency per row)
    for (int r = r0 + ty; r < r1; r += 4 * TY) {
        u32x4 in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * TY;
            if (rr < r1) in[u] = (u == 0 && r == r0 + ty) ? pre : *reinterpret_cast<const u32x4*>(y + (long)rr * ld_y + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * TY;
            if (rr >= r1) break;
            float v[8];
            unpack8<T>(in[u], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float zf = fmaf(v[e], sc[e], sf[e]);
                v[e] = relu ? fmaxf(zf, 0.f) : zf;
            }
            store_wt(rs_z, ((long)rr * ld_z + c) * 2, pack8<T>(v));
        }
    }
}

// pass 1: dbeta[c] += sum dyh, s2[c] += sum dyh*xhat, with zf = y*scale+shift, dyh = dz*(zf>0), xhat = zf - beta.
// Grid = (row chunks, 64-channel stripes).  Global float atomics to ONE address serialise (~0.09 TB/s, MI355X_MICROARCH
// "Global float atomics"), so the host bounds row_chunks * 2C to ~64K adds and the stripes supply the parallelism.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const unsigned short* __restrict__ dz, int ld_d,
                                                                 const unsigned short* __restrict__ y, int ld_y, int M, int C,
                                                                 const float* __restrict__ beta, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, acc_t* __restrict__ dbeta,
                                                                 acc_t* __restrict__ s2, int relu, int rows_per_block) {
    // (dbeta, s2) here are the two halves of the fixed-point accumulator workspace: acc[c] and acc[sq_off + c]
    __shared__ float red[32][2 * 64 + 1];
    const int lin_ = xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);   // rows -> XCD like the convolutions
    const int bx_ = lin_ / gridDim.y, by_ = lin_ - bx_ * gridDim.y;
    const int c0 = by_ * 64;
    const int ncg = min(8, (C - c0) >> 3);          // 16-B channel groups in this stripe
    const int TX = ncg <= 1 ? 1 : (ncg <= 2 ? 2 : (ncg <= 4 ? 4 : 8));
    const int TY = 256 / TX;                        // 32 .. 256 row lanes
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int r0 = bx_ * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float a1[8], a2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    if (tx < ncg) {
        const int c = c0 + tx * 8;
        float b[8], sc[8], sf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { b[e] = beta[c + e]; sc[e] = scale[c + e]; sf[e] = shift[c + e]; }
        for (int r = r0 + ty; r < r1; r += 4 * TY) {
            u32x4 ig[4], iy[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + u * TY;
                if (rr < r1) {
                    ig[u] = *reinterpret_cast<const u32x4*>(dz + (long)rr * ld_d + c);
                    iy[u] = *reinterpret_cast<const u32x4*>(y + (long)rr * ld_y + c);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (r + u * TY >= r1) break;
                float g[8], yy[8];
                unpack8<T>(ig[u], g);
                unpack8<T>(iy[u], yy);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float zf = fmaf(yy[e], sc[e], sf[e]);
                    const float gg = (!relu || zf > 0.f) ? g[e] : 0.f;
                    a1[e] += gg;
                    a2[e] += gg * (zf - b[e]);
                }
            }
        }
    }
    // fold the TY row lanes down to 32 with shuffles-free LDS passes: lanes ty and ty+32k share red[ty & 31]
    for (int pass = 0; pass < TY / 32; ++pass) {
        if (ty / 32 == pass && tx < ncg) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (pass == 0) { red[ty & 31][tx * 8 + e] = a1[e]; red[ty & 31][64 + tx * 8 + e] = a2[e]; }
                else { red[ty & 31][tx * 8 + e] += a1[e]; red[ty & 31][64 + tx * 8 + e] += a2[e]; }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 128) {
        const int col = threadIdx.x;   // 0..63 -> dbeta, 64..127 -> s2
        const int c = c0 + (col & 63);
        if (c < C && (col & 63) < ncg * 8) {
            float s = 0.f;
#pragma unroll 8
            for (int t = 0; t < 32; ++t) s += red[t][col];
            acc_add<ACC_GRAD>(col < 64 ? &dbeta[c] : &s2[c], s);
        }
    }
}

// pass 2: dy = rstd * (dyh - S1/M - xhat * S2/M) for EVERY element (masked ones included), in place over dz.
// Grid = (row chunks, 64-channel stripes); S1, S2 = accumulator replicas summed in the prologue; dbeta += S1.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(unsigned short* __restrict__ dz, int ld_d,
                                                                const unsigned short* __restrict__ y, int ld_y, int M, int C,
                                                                const float* __restrict__ beta, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, float* __restrict__ dbeta,
                                                                const acc_t* __restrict__ acc, int sq_off, int reps, int rep_stride,
                                                                int relu, int rows_per_block) {
    __shared__ float s_k1[64], s_k2[64];
    __shared__ acc_t s_part[4][2][64];
    const int lin_ = xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);   // rows -> XCD like the convolutions
    const int bx_ = lin_ / gridDim.y, by_ = lin_ - bx_ * gridDim.y;
    const int c0 = by_ * 64;
    // first chunk of dz / y requested before the reduction prologue (see bn_relu_fwd_kernel)
    const int ncg = min(8, (C - c0) >> 3);
    const int TX = ncg <= 1 ? 1 : (ncg <= 2 ? 2 : (ncg <= 4 ? 4 : 8));
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = c0 + tx * 8;
    const int r0 = bx_ * rows_per_block, r1 = min(M, r0 + rows_per_block);
    u32x4 pre_g = {0u, 0u, 0u, 0u}, pre_y = {0u, 0u, 0u, 0u};
    if (tx < ncg && r0 + ty < r1) {
        pre_g = *reinterpret_cast<const u32x4*>(dz + (long)(r0 + ty) * ld_d + c);
        pre_y = *reinterpret_cast<const u32x4*>(y + (long)(r0 + ty) * ld_y + c);
    }
    {
        const int cc = threadIdx.x & 63, q = threadIdx.x >> 6;
        acc_t s1 = 0, s2v = 0;          // fixed-point replicas: exact integer sums
        if (c0 + cc < C) {
#pragma unroll 4
            for (int rp = q; rp < reps; rp += 4) {
                s1 += acc[(long)rp * rep_stride + c0 + cc];
                s2v += acc[(long)rp * rep_stride + sq_off + c0 + cc];
            }
        }
        s_part[q][0][cc] = s1;
        s_part[q][1][cc] = s2v;
    }
    __syncthreads();
    if (threadIdx.x < 64 && c0 + threadIdx.x < C) {
        const float s1 = acc_get<ACC_GRAD>(s_part[0][0][threadIdx.x] + s_part[1][0][threadIdx.x] + s_part[2][0][threadIdx.x] + s_part[3][0][threadIdx.x]);
        const float s2v = acc_get<ACC_GRAD>(s_part[0][1][threadIdx.x] + s_part[1][1][threadIdx.x] + s_part[2][1][threadIdx.x] + s_part[3][1][threadIdx.x]);
        const float invM = 1.f / (float)M;
        s_k1[threadIdx.x] = s1 * invM;
        s_k2[threadIdx.x] = s2v * invM;
        if (bx_ == 0) dbeta[c0 + threadIdx.x] += s1;
    }
    __syncthreads();
    if (tx >= ncg) return;
    const __amdgpu_buffer_rsrc_t rs_dz = wt_rsrc(dz);
    float k1[8], k2[8], sc[8], sf[8], bt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        k1[e] = s_k1[tx * 8 + e]; k2[e] = s_k2[tx * 8 + e]; sc[e] = scale[c + e]; sf[e] = shift[c + e]; bt[e] = beta[c + e];
    }
    for (int r = r0 + ty; r < r1; r += 4 * TY) {      // four rows per trip, all loads first (see bn_relu_fwd_kernel)
        u32x4 ig[4], iy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * TY;
            if (rr < r1) {
                const bool first = u == 0 && r == r0 + ty;
                ig[u] = first ? pre_g : *reinterpret_cast<const u32x4*>(dz + (long)rr * ld_d + c);
                iy[u] = first ? pre_y : *reinterpret_cast<const u32x4*>(y + (long)rr * ld_y + c);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * TY;
            if (rr >= r1) break;
            float g[8], yy[8];
            unpack8<T>(ig[u], g);
            unpack8<T>(iy[u], yy);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float zf = fmaf(yy[e], sc[e], sf[e]);
                const float gg = (!relu || zf > 0.f) ? g[e] : 0.f;
                g[e] = sc[e] * (gg - k1[e] - (zf - bt[e]) * k2[e]);
            }
            store_wt(rs_dz, ((long)rr * ld_d + c) * 2, pack8<T>(g));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MaxPool2D(3, strides=2, 'valid')
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const unsigned short* __restrict__ x, int ld_x, unsigned short* __restrict__ y,
                                                          int ld_y, int N, int H, int W, int C, int OH, int OW, uint8_t* __restrict__ amax) {
    const int CG = C >> 3;
    const long total = (long)N * OH * OW * CG;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int cg = (int)(t % CG);
        long pix = t / CG;
        const int ox = (int)(pix % OW); pix /= OW;
        const int oy = (int)(pix % OH);
        const int n = (int)(pix / OH);
        float m[8];
        unsigned am[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { m[e] = -3.0e38f; am[e] = 0; }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float v[8];
                unpack8<T>(*reinterpret_cast<const u32x4*>(x + ((long)(n * H + oy * 2 + ky) * W + ox * 2 + kx) * ld_x + cg * 8), v);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (v[e] > m[e]) { m[e] = v[e]; am[e] = ky * 3 + kx; }   // strict >: the FIRST maximum wins (tf / torch CPU tie rule)
            }
        *reinterpret_cast<u32x4*>(y + ((long)(n * OH + oy) * OW + ox) * ld_y + cg * 8) = pack8<T>(m);
        if (amax) {
            const unsigned lo = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24), hi = am[4] | (am[5] << 8) | (am[6] << 16) | (am[7] << 24);
            *reinterpret_cast<uint2*>(amax + ((long)(n * OH + oy) * OW + ox) * C + cg * 8) = make_uint2(lo, hi);
        }
    }
}

// gather form (deterministic, no atomics): input element (iy,ix) receives dy of every covering window whose
// FIRST maximum (scan order ky, kx) sits at (iy,ix) -- the tie rule of tf/torch CPU max-pool gradients.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned short* __restrict__ x, int ld_x,
                                                          const unsigned short* __restrict__ dy, int ld_dy,
                                                          unsigned short* __restrict__ dx, int ld_dx, int N, int H, int W, int C, int OH,
                                                          int OW, int accumulate) {
    const int CG = C >> 3;
    const long total = (long)N * H * W * CG;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int cg = (int)(t % CG);
        long pix = t / CG;
        const int ix = (int)(pix % W); pix /= W;
        const int iy = (int)(pix % H);
        const int n = (int)(pix / H);
        float mine[8], g[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(x + ((long)(n * H + iy) * W + ix) * ld_x + cg * 8), mine);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.f;
        const int oy_lo = max(0, (iy - 1) >> 1), oy_hi = min(OH - 1, iy >> 1);
        const int ox_lo = max(0, (ix - 1) >> 1), ox_hi = min(OW - 1, ix >> 1);
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const int my = (iy - 2 * oy) * 3 + (ix - 2 * ox);  // my scan position in this window
                bool win[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) win[e] = true;
                // only the EARLIER positions need loading for the tie rule once the window maximum is known:
                // I win iff x == max(window) and no earlier element equals it.  max(window) = max over later ones too,
                // so later elements are still compared, but with <= (no tie break) -- identical result, same loads;
                // the saving is the early exit for windows where this element is clearly not the maximum.
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    if (k == my) continue;
                    float v[8];
                    unpack8<T>(*reinterpret_cast<const u32x4*>(x + ((long)(n * H + 2 * oy + k / 3) * W + 2 * ox + k % 3) * ld_x + cg * 8), v);
                    bool any = false;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        win[e] = win[e] && (k < my ? (v[e] < mine[e]) : (v[e] <= mine[e]));
                        any = any || win[e];
                    }
                    if (!any) break;   // nobody in this channel group can still be the first maximum
                }
                bool any = false;
#pragma unroll
                for (int e = 0; e < 8; ++e) any = any || win[e];
                if (!any) continue;
                float d[8];
                unpack8<T>(*reinterpret_cast<const u32x4*>(dy + ((long)(n * OH + oy) * OW + ox) * ld_dy + cg * 8), d);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] += win[e] ? d[e] : 0.f;
            }
        unsigned short* o = dx + ((long)(n * H + iy) * W + ix) * ld_dx + cg * 8;
        if (accumulate) {
            float pv[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(o), pv);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] += pv[e];
        }
        *reinterpret_cast<u32x4*>(o) = pack8<T>(g);
    }
}

// AvgPool2D over the whole HW map (3x3 -> 1x1 at 160x160, :460) + Flatten
// argmax form: 4 x (8 B argmax + 16 B dy) per thread instead of re-deriving the maximum from up to 36 window loads
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_amax_kernel(const uint8_t* __restrict__ amax, const unsigned short* __restrict__ dy, int ld_dy,
                                                               unsigned short* __restrict__ dx, int ld_dx, int N, int H, int W, int C, int OH,
                                                               int OW, int accumulate) {
    const int CG = C >> 3;
    const long total = (long)N * H * W * CG;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int cg = (int)(t % CG);
        long pix = t / CG;
        const int ix = (int)(pix % W); pix /= W;
        const int iy = (int)(pix % H);
        const int n = (int)(pix / H);
        float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int oy_lo = max(0, (iy - 1) >> 1), oy_hi = min(OH - 1, iy >> 1);
        const int ox_lo = max(0, (ix - 1) >> 1), ox_hi = min(OW - 1, ix >> 1);
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const unsigned my = (iy - 2 * oy) * 3 + (ix - 2 * ox);
                const long o = (long)(n * OH + oy) * OW + ox;
                const uint2 a = *reinterpret_cast<const uint2*>(amax + o * C + cg * 8);
                float d[8];
                unpack8<T>(*reinterpret_cast<const u32x4*>(dy + o * ld_dy + cg * 8), d);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned am = ((e < 4 ? a.x : a.y) >> (8 * (e & 3))) & 0xffu;
                    g[e] += (am == my) ? d[e] : 0.f;
                }
            }
        unsigned short* op = dx + ((long)(n * H + iy) * W + ix) * ld_dx + cg * 8;
        if (accumulate) {
            float pv[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(op), pv);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] += pv[e];
        }
        *reinterpret_cast<u32x4*>(op) = pack8<T>(g);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int N, int HW,
                                                          int C) {
    const int CG = C >> 3;
    const int total = N * CG;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const int n = t / CG, cg = t - n * CG;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = 0; p < HW; ++p) {
            float v[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(x + ((long)n * HW + p) * C + cg * 8), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v[e];
        }
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] *= inv;
        *reinterpret_cast<u32x4*>(y + (long)n * C + cg * 8) = pack8<T>(s);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx, int N, int HW,
                                                          int C) {
    const int CG = C >> 3;
    const int total = N * CG;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const int n = t / CG, cg = t - n * CG;
        float v[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(dy + (long)n * C + cg * 8), v);
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= inv;
        const u32x4 pk = pack8<T>(v);
        for (int p = 0; p < HW; ++p) *reinterpret_cast<u32x4*>(dx + ((long)n * HW + p) * C + cg * 8) = pk;
    }
}

// ------------------------------------------------------------------------------------------------
// residual backward:  out = act(trunk + scale*(up + bias))
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void residual_bwd_kernel(const unsigned short* __restrict__ dout, const unsigned short* __restrict__ out,
                                                           unsigned short* __restrict__ dtrunk, unsigned short* __restrict__ dup,
                                                           acc_t* __restrict__ dbias, int M, int C, float scale, int relu, int accumulate,
                                                           int rows_per_block) {
    __shared__ float red[32][64 + 1];
    const int lin_ = xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);   // rows -> XCD like the convolutions
    const int bx_ = lin_ / gridDim.y, by_ = lin_ - bx_ * gridDim.y;
    const int c0 = by_ * 64;
    const int ncg = min(8, (C - c0) >> 3);
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int r0 = bx_ * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float a1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (tx < ncg) {
        for (int r = r0 + ty; r < r1; r += 32) {
            const long o = (long)r * C + c0 + tx * 8;
            float g[8], zz[8], u[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(dout + o), g);
            if (relu) {
                unpack8<T>(*reinterpret_cast<const u32x4*>(out + o), zz);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { u[e] = scale * g[e]; a1[e] += u[e]; }
            *reinterpret_cast<u32x4*>(dup + o) = pack8<T>(u);
            if (accumulate) {
                float pv[8];
                unpack8<T>(*reinterpret_cast<const u32x4*>(dtrunk + o), pv);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] += pv[e];
            }
            *reinterpret_cast<u32x4*>(dtrunk + o) = pack8<T>(g);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = a1[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
#pragma unroll 8
        for (int t = 0; t < 32; ++t) s += red[t][threadIdx.x];
        const int c = c0 + threadIdx.x;
        if (c < C) acc_add<ACC_GRAD>(&dbias[c], s);
    }
}

// ------------------------------------------------------------------------------------------------
// embedding head (fp32 [N,E]): BatchNorm without ReLU, l2_normalize
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void head_bn_fwd_kernel(const float* __restrict__ y, float* __restrict__ out, int N, int E,
                                                         const float* __restrict__ beta, float* __restrict__ mm, float* __restrict__ mv,
                                                         float* __restrict__ save_mean, float* __restrict__ save_rstd, int training,
                                                         float momentum, float eps) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= E) return;
    float mean, var;
    if (training) {
        float s = 0.f;
#pragma unroll 8
        for (int n = 0; n < N; ++n) s += y[(long)n * E + c];
        mean = s / (float)N;
        float q = 0.f;
#pragma unroll 8
        for (int n = 0; n < N; ++n) { const float d = y[(long)n * E + c] - mean; q += d * d; }
        var = q / (float)N;
        mm[c] = bn_moving_update(mm[c], mean, momentum);
        mv[c] = bn_moving_update(mv[c], var, momentum);
    } else {
        mean = mm[c];
        var = mv[c];
    }
    const float rstd = rsqrtf(var + eps);
    if (save_mean) { save_mean[c] = mean; save_rstd[c] = rstd; }
    const float b = beta[c];
#pragma unroll 8
    for (int n = 0; n < N; ++n) out[(long)n * E + c] = (y[(long)n * E + c] - mean) * rstd + b;
}

template <typename T>
__global__ __launch_bounds__(64) void head_bn_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                         const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                         float* __restrict__ dbeta, unsigned short* __restrict__ dy, int N, int E) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= E) return;
    const float mean = save_mean[c], rstd = save_rstd[c];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) {
        const float g = dout[(long)n * E + c];
        s1 += g;
        s2 += g * (y[(long)n * E + c] - mean) * rstd;
    }
    dbeta[c] += s1;
    const float k1 = s1 / (float)N, k2 = s2 / (float)N;
#pragma unroll 8
    for (int n = 0; n < N; ++n) {
        const float xh = (y[(long)n * E + c] - mean) * rstd;
        dy[(long)n * E + c] = LP<T>::from_f32(rstd * (dout[(long)n * E + c] - k1 - xh * k2));
    }
}

// one wave per row: out = x * rsqrt(max(sum x^2, eps))   (tf.nn.l2_normalize, :491-492)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int E, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    float s = 0.f;
    for (int k = lane; k < E; k += 64) { const float v = x[(long)row * E + k]; s += v * v; }
    s = wave_sum(s);
    const float r = rsqrtf(fmaxf(s, eps));
    for (int k = lane; k < E; k += 64) out[(long)row * E + k] = x[(long)row * E + k] * r;
}
// dx = r*(dout - xn * <dout, xn>) with xn = x*r (the eps clamp branch has zero derivative through r)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout, float* __restrict__ dx,
                                                         int N, int E, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    float s = 0.f, d = 0.f;
    for (int k = lane; k < E; k += 64) {
        const float v = x[(long)row * E + k];
        s += v * v;
        d += v * dout[(long)row * E + k];
    }
    s = wave_sum(s);
    d = wave_sum(d);
    const bool clamped = s < eps;
    const float r = rsqrtf(fmaxf(s, eps));
    for (int k = lane; k < E; k += 64) {
        const float v = x[(long)row * E + k];
        dx[(long)row * E + k] = clamped ? r * dout[(long)row * E + k] : r * (dout[(long)row * E + k] - v * r * r * d);
    }
}

template <typename T> __global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = LP<T>::from_f32(x[i]);
}

// column reductions: row chunks such that chunks * 2C global atomics stay ~<= 64K and every chunk has >= 32 rows
static inline int reduce_rows_per_block(int M, int C) {
    int chunks = 32768 / (C > 0 ? C : 1);
    if (chunks < 8) chunks = 8;
    if (chunks > 1024) chunks = 1024;
    int rpb = cdiv(M, chunks);
    if (rpb < 32) rpb = 32;
    return rpb;
}

static inline int grid_for(long work_items, int per_block = 256, int cap = 4096) {
    long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace fn

using namespace fn;
#define DT_CHECK(dt) FN_REQUIRE((dt) == FN_BF16 || (dt) == FN_F16, "dtype %d unsupported", (dt))
#define LAUNCH_T(dt, KERN, grid, block, smem, st, ...)                                             \
    do {                                                                                             \
        if ((dt) == FN_BF16) hipLaunchKernelGGL(KERN<__bf16>, grid, block, smem, st, __VA_ARGS__);   \
        else hipLaunchKernelGGL(KERN<_Float16>, grid, block, smem, st, __VA_ARGS__);                 \
    } while (0)

template <typename SRC>
static int image_normalize_impl(const SRC* img, void* out, float* work, int N, int HW, int mode, int dtype, void* stream) {
    DT_CHECK(dtype);
    FN_REQUIRE(img && out && work && N > 0 && HW > 0, "image_normalize: bad arguments");
    FN_REQUIRE(mode == 0 || mode == 1, "Invalid image normalization algorithm");  // facenet.py:82
    FN_REQUIRE(((uintptr_t)img & 15) == 0 && ((long)HW * 3 * sizeof(SRC)) % 16 == 0, "image_normalize: images must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    fill_words(work, 0u, 0u, 8 * N, st);
    hipLaunchKernelGGL(img_stats_kernel<SRC>, dim3(8, N), dim3(256), 0, st, img, (unsigned*)work, HW * 3);
    const int gx = grid_for(HW, 256, 32);
    if (dtype == FN_BF16) hipLaunchKernelGGL((img_apply_kernel<__bf16, SRC>), dim3(gx, N), dim3(256), 0, st, img, (unsigned short*)out, (const unsigned*)work, HW, mode);
    else hipLaunchKernelGGL((img_apply_kernel<_Float16, SRC>), dim3(gx, N), dim3(256), 0, st, img, (unsigned short*)out, (const unsigned*)work, HW, mode);
    return check_launch("image_normalize");
}
extern "C" int fn_image_normalize(const uint8_t* img, void* out, float* work, int N, int HW, int mode, int dtype, void* stream) {
    return image_normalize_impl<uint8_t>(img, out, work, N, HW, mode, dtype, stream);
}
extern "C" int fn_image_normalize_f32(const float* img, void* out, float* work, int N, int HW, int mode, int dtype, void* stream) {
    return image_normalize_impl<float>(img, out, work, N, HW, mode, dtype, stream);
}

extern "C" int fn_image_resize_bilinear(const void* img, int src_is_f32, float* out, int N, int H, int W, int OH, int OW, void* stream) {
    FN_REQUIRE(img && out && N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "image_resize: bad arguments");
    const int grid = grid_for((long)N * OH * OW);
    if (src_is_f32) hipLaunchKernelGGL(resize_bilinear_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)img, out, N, H, W, OH, OW);
    else hipLaunchKernelGGL(resize_bilinear_kernel<uint8_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img, out, N, H, W, OH, OW);
    return check_launch("image_resize");
}

// tf.image.resize_with_crop_or_pad (ImageLoader, facenet.py:45-54): a ragged batch of decoded HWC u8 images, packed back to
// back in `src` (image n starts at byte off[n], is hw[2n] x hw[2n+1] x 3), centre-cropped / zero-padded to S x S.
// One thread per 4 output bytes (one 32-bit store); source rows are unaligned byte runs, read through L1.
__global__ __launch_bounds__(256) void crop_or_pad_kernel(const uint8_t* __restrict__ src, const long long* __restrict__ off,
                                                          const int* __restrict__ hw, uint8_t* __restrict__ dst, int S) {
    const int n = blockIdx.y;
    const int h = hw[2 * n], w = hw[2 * n + 1];
    const int cy = max((h - S) / 2, 0), cx = max((w - S) / 2, 0);     // offset_crop = max(-diff // 2, 0)
    const int py = max((S - h) / 2, 0), px = max((S - w) / 2, 0);     // offset_pad  = max( diff // 2, 0)
    const int ch = min(h, S), cw = min(w, S);
    const uint8_t* im = src + off[n];
    const int words = S * S * 3 / 4;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < words; t += gridDim.x * 256) {
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int e = t * 4 + b;
            const int c = e % 3, pix = e / 3;
            const int x = pix % S - px, y = pix / S - py;
            if (x >= 0 && x < cw && y >= 0 && y < ch) v |= (unsigned)im[((long)(y + cy) * w + (x + cx)) * 3 + c] << (8 * b);
        }
        reinterpret_cast<unsigned*>(dst + (long)n * S * S * 3)[t] = v;
    }
}

extern "C" int fn_crop_or_pad_u8(const uint8_t* src, const long long* offsets, const int32_t* hw, uint8_t* dst, int N, int S, void* stream) {
    FN_REQUIRE(src && offsets && hw && dst && N > 0 && S > 0 && (S * S * 3) % 4 == 0, "crop_or_pad: bad arguments");
    hipLaunchKernelGGL(crop_or_pad_kernel, dim3(cdiv(S * S * 3 / 4, 256 * 4), N), dim3(256), 0, (hipStream_t)stream, src, offsets, hw, dst, S);
    return check_launch("crop_or_pad");
}

extern "C" int fn_gather_images(const uint8_t* pool, const int32_t* idx, uint8_t* out, int n_out, int bytes, void* stream) {
    FN_REQUIRE(pool && idx && out && n_out > 0 && bytes > 0 && bytes % 16 == 0, "gather_images: bad arguments");
    hipLaunchKernelGGL(gather_images_kernel, dim3(8, n_out), dim3(256), 0, (hipStream_t)stream, pool, idx, out, bytes);
    return check_launch("gather_images");
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const acc_t* __restrict__ stats, int sq_off, int rep_stride, const int* __restrict__ reps,
                                                          const int* __restrict__ count, const float* __restrict__ beta,
                                                          float* __restrict__ save_scale, float* __restrict__ save_shift,
                                                          float* __restrict__ mm, float* __restrict__ mv, float momentum, float eps, int CB) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= CB || reps[c] <= 0) return;
    float sc, sh, mean, var;
    bn_batch_affine(stats, c, sq_off, reps[c], rep_stride, count[c], eps, beta[c], sc, sh, mean, var);
    save_scale[c] = sc;
    save_shift[c] = sh;
    if (mm) {
        mm[c] = bn_moving_update(mm[c], mean, momentum);
        mv[c] = bn_moving_update(mv[c], var, momentum);
    }
}

extern "C" int fn_bn_finalize(const fn_acc_t* stats, int sq_off, int rep_stride, const int32_t* reps, const int32_t* count, const float* beta,
                              float* save_scale, float* save_shift, float* moving_mean, float* moving_var, float momentum, float eps, int CB,
                              void* stream) {
    FN_REQUIRE(stats && reps && count && beta && save_scale && save_shift && CB > 0 && (!moving_mean == !moving_var), "bn_finalize: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(CB, 256)), dim3(256), 0, (hipStream_t)stream, (const acc_t*)stats, sq_off, rep_stride, reps, count, beta,
                       save_scale, save_shift, moving_mean, moving_var, momentum, eps, CB);
    return check_launch("bn_finalize");
}

extern "C" int fn_bn_relu_train_fwd(const void* y, int ld_y, void* z, int ld_z, int M, int C, const fn_acc_t* stats, int stats_sq_off,
                                    int stats_replicas, int stats_rep_stride, const float* beta,
                                    float* save_scale, float* save_shift, float* moving_mean, float* moving_var, float momentum, float eps,
                                    int relu, int dtype, void* stream) {
    DT_CHECK(dtype);
    FN_REQUIRE(y && z && stats && beta && save_scale && save_shift && M > 0 && C > 0 && C % 8 == 0 && ld_y % 8 == 0 && ld_z % 8 == 0 &&
                   ld_y >= C && ld_z >= C && C <= 8192, "bn_fwd: bad arguments");
    FN_REQUIRE((long)M * ld_z * 2 < (1L << 31), "bn_fwd: z exceeds the 2 GiB range of the 32-bit byte offsets its stores use");
    const int stripes = cdiv(C, 64);
    int rpb = cdiv((long)M * stripes, 2048);
    if (rpb < 32) rpb = 32;
    LAUNCH_T(dtype, bn_relu_fwd_kernel, dim3(cdiv(M, rpb), stripes), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)y, ld_y, (unsigned short*)z, ld_z, M, C, (const acc_t*)stats, stats_sq_off, stats_replicas > 0 ? stats_replicas : 1, stats_rep_stride, beta, save_scale, save_shift, moving_mean, moving_var, momentum, eps, relu, rpb);
    return check_launch("bn_relu_fwd");
}

extern "C" int fn_bn_relu_train_bwd(void* dz, int ld_d, const void* y, int ld_y, int M, int C, const float* beta, const float* save_scale,
                                    const float* save_shift, float* dbeta, fn_acc_t* acc_, int acc_sq_off, int acc_replicas, int acc_rep_stride,
                                    int reduced, int relu, int dtype, void* stream) {
    acc_t* acc = reinterpret_cast<acc_t*>(acc_);
    DT_CHECK(dtype);
    FN_REQUIRE(dz && y && beta && save_scale && save_shift && dbeta && acc && M > 0 && C > 0 && C % 8 == 0 && ld_d % 8 == 0 &&
                   ld_y % 8 == 0 && C <= 4096, "bn_bwd: bad arguments");
    FN_REQUIRE((long)M * ld_d * 2 < (1L << 31), "bn_bwd: dz exceeds the 2 GiB range of the 32-bit byte offsets its stores use");
    hipStream_t st = (hipStream_t)stream;
    const int reps = acc_replicas > 0 ? acc_replicas : 1;
    if (!reduced) {
        const int rpb = reduce_rows_per_block(M, C);
        const dim3 g1(cdiv(M, rpb), cdiv(C, 64));
        LAUNCH_T(dtype, bn_relu_bwd_reduce_kernel, g1, dim3(256), 0, st, (const unsigned short*)dz, ld_d, (const unsigned short*)y, ld_y, M, C, beta, save_scale, save_shift, acc, acc + acc_sq_off, relu, rpb);
    }
    const int stripes = cdiv(C, 64);
    int rpb2 = cdiv((long)M * stripes, 2048);
    if (rpb2 < 32) rpb2 = 32;
    LAUNCH_T(dtype, bn_relu_bwd_apply_kernel, dim3(cdiv(M, rpb2), stripes), dim3(256), 0, st, (unsigned short*)dz, ld_d, (const unsigned short*)y, ld_y, M, C, beta, save_scale, save_shift, dbeta, acc, acc_sq_off, reps, acc_rep_stride, relu, rpb2);
    return check_launch("bn_relu_bwd");
}

extern "C" int fn_maxpool3x3s2_fwd(const void* x, int ld_x, void* y, int ld_y, int N, int H, int W, int C, uint8_t* argmax, int dtype,
                                   void* stream) {
    DT_CHECK(dtype);
    FN_REQUIRE(x && y && N > 0 && H >= 3 && W >= 3 && C % 8 == 0 && ld_x % 8 == 0 && ld_y % 8 == 0, "maxpool_fwd: bad arguments");
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
    const int grid = grid_for((long)N * OH * OW * (C / 8));
    LAUNCH_T(dtype, maxpool_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, ld_x, (unsigned short*)y, ld_y, N, H, W, C, OH, OW, argmax);
    return check_launch("maxpool_fwd");
}

extern "C" int fn_maxpool3x3s2_bwd(const void* x, int ld_x, const void* dy, int ld_dy, void* dx, int ld_dx, int N, int H, int W, int C,
                                   const uint8_t* argmax, int accumulate, int dtype, void* stream) {
    DT_CHECK(dtype);
    if (argmax) {
        FN_REQUIRE(dy && dx && N > 0 && H >= 3 && W >= 3 && C % 8 == 0 && ld_dy % 8 == 0 && ld_dx % 8 == 0, "maxpool_bwd: bad arguments");
        const int OH2 = (H - 3) / 2 + 1, OW2 = (W - 3) / 2 + 1;
        LAUNCH_T(dtype, maxpool_bwd_amax_kernel, dim3(grid_for((long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, argmax, (const unsigned short*)dy, ld_dy, (unsigned short*)dx, ld_dx, N, H, W, C, OH2, OW2, accumulate);
        return check_launch("maxpool_bwd");
    }
    FN_REQUIRE(x && dy && dx && N > 0 && H >= 3 && W >= 3 && C % 8 == 0 && ld_x % 8 == 0 && ld_dy % 8 == 0 && ld_dx % 8 == 0,
               "maxpool_bwd: bad arguments");
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
    const int grid = grid_for((long)N * H * W * (C / 8));
    LAUNCH_T(dtype, maxpool_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, ld_x, (const unsigned short*)dy, ld_dy, (unsigned short*)dx, ld_dx, N, H, W, C, OH, OW, accumulate);
    return check_launch("maxpool_bwd");
}

extern "C" int fn_avgpool_fwd(const void* x, void* y, int N, int HW, int C, int dtype, void* stream) {
    DT_CHECK(dtype);
    FN_REQUIRE(x && y && N > 0 && HW > 0 && C % 8 == 0, "avgpool_fwd: bad arguments");
    LAUNCH_T(dtype, avgpool_fwd_kernel, dim3(grid_for((long)N * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (unsigned short*)y, N, HW, C);
    return check_launch("avgpool_fwd");
}
extern "C" int fn_avgpool_bwd(const void* dy, void* dx, int N, int HW, int C, int dtype, void* stream) {
    DT_CHECK(dtype);
    FN_REQUIRE(dy && dx && N > 0 && HW > 0 && C % 8 == 0, "avgpool_bwd: bad arguments");
    LAUNCH_T(dtype, avgpool_bwd_kernel, dim3(grid_for((long)N * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)dy, (unsigned short*)dx, N, HW, C);
    return check_launch("avgpool_bwd");
}

extern "C" int fn_residual_bwd(const void* dout, const void* out, void* dtrunk, void* dup, fn_acc_t* dbias_, int M, int C, float scale, int relu,
                               int accumulate, int dtype, void* stream) {
    DT_CHECK(dtype);
    acc_t* dbias = reinterpret_cast<acc_t*>(dbias_);
    FN_REQUIRE(dout && dtrunk && dup && dbias && (out || !relu) && M > 0 && C > 0 && C % 8 == 0 && C <= 8192, "residual_bwd: bad arguments");
    const int rpb = reduce_rows_per_block(M, C);
    LAUNCH_T(dtype, residual_bwd_kernel, dim3(cdiv(M, rpb), cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)dout, (const unsigned short*)out, (unsigned short*)dtrunk, (unsigned short*)dup, dbias, M, C, scale, relu, accumulate, rpb);
    return check_launch("residual_bwd");
}

extern "C" int fn_head_bn_fwd(const float* y, float* out, int N, int E, const float* beta, float* moving_mean, float* moving_var,
                              float* save_mean, float* save_rstd, int training, float momentum, float eps, void* stream) {
    FN_REQUIRE(y && out && beta && moving_mean && moving_var && N > 0 && E > 0, "head_bn_fwd: bad arguments");
    hipLaunchKernelGGL(head_bn_fwd_kernel, dim3(cdiv(E, 64)), dim3(64), 0, (hipStream_t)stream, y, out, N, E, beta, moving_mean, moving_var,
                       save_mean, save_rstd, training, momentum, eps);
    return check_launch("head_bn_fwd");
}
extern "C" int fn_head_bn_bwd(const float* dout, const float* y, const float* save_mean, const float* save_rstd, float* dbeta, void* dy_lp,
                              int N, int E, int dtype, void* stream) {
    DT_CHECK(dtype);
    FN_REQUIRE(dout && y && save_mean && save_rstd && dbeta && dy_lp && N > 0 && E > 0, "head_bn_bwd: bad arguments");
    LAUNCH_T(dtype, head_bn_bwd_kernel, dim3(cdiv(E, 64)), dim3(64), 0, (hipStream_t)stream, dout, y, save_mean, save_rstd, dbeta, (unsigned short*)dy_lp, N, E);
    return check_launch("head_bn_bwd");
}
extern "C" int fn_l2norm_fwd(const float* x, float* out, int N, int E, float eps, void* stream) {
    FN_REQUIRE(x && out && N > 0 && E > 0, "l2norm_fwd: bad arguments");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, x, out, N, E, eps);
    return check_launch("l2norm_fwd");
}
extern "C" int fn_l2norm_bwd(const float* x, const float* dout, float* dx, int N, int E, float eps, void* stream) {
    FN_REQUIRE(x && dout && dx && N > 0 && E > 0, "l2norm_bwd: bad arguments");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, x, dout, dx, N, E, eps);
    return check_launch("l2norm_bwd");
}
extern "C" int fn_cast_f32_to_lp(const float* x, void* y, long n, int dtype, void* stream) {
    DT_CHECK(dtype);
    FN_REQUIRE(x && y && n > 0, "cast: bad arguments");
    LAUNCH_T(dtype, cast_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)y, n);
    return check_launch("cast");
}

// fixed-point accumulators -> fp32 (bias gradients into the flat gradient buffer; fn_acc_t in facenet_hip.h)
namespace fn {
__global__ __launch_bounds__(256) void acc_to_float_kernel(const acc_t* __restrict__ src, float* __restrict__ dst, long n, double scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (float)((double)src[i] * scale);
}
}  // namespace fn
extern "C" int fn_acc_to_float(const fn_acc_t* src, float* dst, long n, int bits, void* stream) {
    FN_REQUIRE(src && dst && n > 0 && bits >= 0 && bits < 63, "acc_to_float: bad arguments");
    hipLaunchKernelGGL(acc_to_float_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const acc_t*>(src), dst, n,
                       1.0 / (double)(1ll << bits));
    return check_launch("acc_to_float");
}
