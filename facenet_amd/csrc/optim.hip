// Optimiser and weight packs.
//  * fn_adam_keras: tf.keras.optimizers.Adam(epsilon=0.1) exactly as Keras applies it
//    (apps/train_softmax.py:92; SURVEY.md hazard 8): lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
//    w -= lr_t*m/(sqrt(v)+eps), with the Keras L2(5e-4) kernel regulariser (inception_resnet_v1.py:65)
//    folded in as a coupled term g += 2*l2*w on the first n_decay elements.  One fused multi-tensor
//    pass over the flat parameter buffer also emits the low-precision weight copy the MFMA kernels read.
//  * fn_pack_transpose: [Cout][tap][Cin] -> [Cin][tap][Cout] (operand of the dgrad implicit GEMM).
//  * fn_fold_bn: inference weights with BatchNorm folded in, the formula of facenet/tfutils.py:244-250.
#include "common.h"
#include "../../include/facenet_hip.h"

namespace fn {

// hyper: [0]=lr [1]=beta1^t [2]=beta2^t [3]=grad_scale [4]=t (int32 bits) ; powers are those AFTER this step's tick
template <typename T>
__global__ __launch_bounds__(256) void adam_keras_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, unsigned short* __restrict__ w_lp, long n_lp, long n, long n_decay,
                                                         const float* __restrict__ hyper, float beta1, float beta2, float eps, float l2) {
    const float lr = hyper[0], b1t = hyper[1], b2t = hyper[2], gs = hyper[3];
    const float lr_t = lr * sqrtf(1.f - b2t) / (1.f - b1t);
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 wv = reinterpret_cast<f32x4*>(w)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float gg = gv[e] * gs;
            if (i * 4 + e < n_decay) gg += 2.f * l2 * wv[e];
            mv[e] = beta1 * mv[e] + (1.f - beta1) * gg;
            vv[e] = beta2 * vv[e] + (1.f - beta2) * gg * gg;
            wv[e] -= lr_t * mv[e] / (sqrtf(vv[e]) + eps);
        }
        // streamed: nothing reads the parameters or the moments before the next step (step 6.488 -> 6.473 ms against plain stores)
        __builtin_nontemporal_store(wv, reinterpret_cast<f32x4*>(w) + i);
        __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m) + i);
        __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
        if (w_lp && i * 4 < n_lp) {
            const unsigned lo = (unsigned)LP<T>::from_f32(wv[0]) | ((unsigned)LP<T>::from_f32(wv[1]) << 16);
            const unsigned hi = (unsigned)LP<T>::from_f32(wv[2]) | ((unsigned)LP<T>::from_f32(wv[3]) << 16);
            reinterpret_cast<uint2*>(w_lp)[i] = make_uint2(lo, hi);
        }
    }
}

// Adam's step count is an integer (word 4 of `hyper`); the beta powers are re-derived from it on every tick.  A running
// product beta1^t in fp32 reaches the denormal floor after ~970 steps and stops changing, so t could not be recovered from it
// (checkpoints: Adam/iter) -- and Keras computes pow(beta, t) from its integer `iterations` as well.
__global__ void adam_tick_kernel(float* hyper, float beta1, float beta2) {
    int* it = reinterpret_cast<int*>(hyper) + 4;
    const int t = *it + 1;
    *it = t;
    hyper[1] = (float)pow((double)beta1, (double)t);
    hyper[2] = (float)pow((double)beta2, (double)t);
}

// one workgroup column per layer (blockIdx.y); table row = {w_off, cout, ktot, taps, cin, bn_off, fold_bias_off, 0}.
// Tiled through LDS: for every tap a [cout][cin] -> [cin][cout] transpose in 32x32 tiles, both sides coalesced
// (the naive destination-linear gather fetched ~10x the bytes: profiles/r01_pmc_hbm_traffic.json).
template <typename T>
__global__ __launch_bounds__(256) void pack_transpose_kernel(const unsigned short* __restrict__ w, unsigned short* __restrict__ wt,
                                                             const int* __restrict__ table) {
    __shared__ __attribute__((aligned(16))) unsigned short tile64[64][72];
    unsigned short (*tile)[33] = reinterpret_cast<unsigned short (*)[33]>(&tile64[0][0]);
    const int* row = table + 8 * blockIdx.y;
    const long off = row[0];
    const int cout = row[1], taps = row[3], cin = row[4];
    if ((cout & 7) == 0) {
        // 64 x 64 tiles moved in 16-byte pieces on both sides (cin and cout are multiples of 8 for every layer but an odd-sized
        // classifier): 8 ci per load along a [co] row, 8 co gathered from LDS per store along a [ci] row
        const int tco = (cout + 63) >> 6, tci = (cin + 63) >> 6;
        const int ntiles = taps * tco * tci;
        for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
            const int tap = t / (tco * tci), r = t - tap * tco * tci;
            const int co0 = (r / tci) << 6, ci0 = (r % tci) << 6;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = threadIdx.x + 256 * i, rr = idx >> 3, ch = idx & 7;
                const int co = co0 + rr, ci = ci0 + ch * 8;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (co < cout && ci < cin) v = *reinterpret_cast<const u32x4*>(w + off + ((long)co * taps + tap) * cin + ci);
                *reinterpret_cast<u32x4*>(&tile64[rr][ch * 8]) = v;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = threadIdx.x + 256 * i, rr = idx >> 3, ch = idx & 7;     // rr: ci row, ch: group of 8 co
                const int ci = ci0 + rr, co = co0 + ch * 8;
                if (ci < cin && co < cout) {
                    u32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = (unsigned)tile64[ch * 8 + 2 * e][rr] | ((unsigned)tile64[ch * 8 + 2 * e + 1][rr] << 16);
                    *reinterpret_cast<u32x4*>(wt + off + ((long)ci * taps + tap) * cout + co) = v;
                }
            }
            __syncthreads();
        }
        return;
    }
    const int tco = (cout + 31) >> 5, tci = (cin + 31) >> 5;
    const int ntiles = taps * tco * tci;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tap = t / (tco * tci), r = t - tap * tco * tci;
        const int co0 = (r / tci) << 5, ci0 = (r % tci) << 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + ty + 8 * i, ci = ci0 + tx;
            tile[ty + 8 * i][tx] = (co < cout && ci < cin) ? w[off + ((long)co * taps + tap) * cin + ci] : (unsigned short)0;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = ci0 + ty + 8 * i, co = co0 + tx;
            if (ci < cin && co < cout) wt[off + ((long)ci * taps + tap) * cout + co] = tile[tx][ty + 8 * i];
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fold_bn_kernel(const float* __restrict__ w, unsigned short* __restrict__ wf, float* __restrict__ fold_bias,
                                                      const float* __restrict__ beta, const float* __restrict__ mm,
                                                      const float* __restrict__ mv, const int* __restrict__ table, float eps) {
    const int* row = table + 8 * blockIdx.y;
    const long off = row[0];
    const int cout = row[1], ktot = row[2], bn = row[5], fb = row[6];
    // 8 weights per thread (ktot and the layer offsets are multiples of 8: channels are padded to 8): two 16-byte loads, one
    // 16-byte store and ONE integer division per 8 elements
    const int total8 = (int)(((long)cout * ktot) >> 3);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total8; i += gridDim.x * 256) {
        const int co = (i << 3) / ktot;
        const float s = bn >= 0 ? rsqrtf(mv[bn + co] + eps) : 1.f;
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + off + ((long)i << 3)), b = *reinterpret_cast<const f32x4*>(w + off + ((long)i << 3) + 4);
        const float v[8] = {a[0] * s, a[1] * s, a[2] * s, a[3] * s, b[0] * s, b[1] * s, b[2] * s, b[3] * s};
        *reinterpret_cast<u32x4*>(wf + off + ((long)i << 3)) = pack8<T>(v);
    }
    if (bn >= 0 && fb >= 0 && blockIdx.x == 0)
        for (int co = threadIdx.x; co < cout; co += 256) {
            const float s = rsqrtf(mv[bn + co] + eps);
            fold_bias[fb + co] = beta[bn + co] - mm[bn + co] * s;
        }
}

}  // namespace fn
using namespace fn;

extern "C" int fn_adam_keras(float* w, const float* g, float* m, float* v, void* w_lp, long n_lp, long n, long n_decay, float* hyper, float beta1,
                             float beta2, float eps, float l2, int dtype, void* stream) {
    FN_REQUIRE(dtype == FN_BF16 || dtype == FN_F16, "dtype %d unsupported", dtype);
    FN_REQUIRE(w && g && m && v && hyper && n > 0 && n % 4 == 0 && n_decay >= 0 && n_decay <= n && n_lp % 4 == 0 && n_lp <= n,
               "adam_keras: bad arguments (n %% 4 == 0)");
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == FN_BF16)
        hipLaunchKernelGGL(adam_keras_kernel<__bf16>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, w, g, m, v, (unsigned short*)w_lp, n_lp, n, n_decay, hyper, beta1, beta2, eps, l2);
    else
        hipLaunchKernelGGL(adam_keras_kernel<_Float16>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, w, g, m, v, (unsigned short*)w_lp, n_lp, n, n_decay, hyper, beta1, beta2, eps, l2);
    return check_launch("adam_keras");
}

extern "C" int fn_adam_tick(float* hyper, float beta1, float beta2, void* stream) {
    FN_REQUIRE(hyper, "adam_tick: null hyper");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper, beta1, beta2);
    return check_launch("adam_tick");
}

extern "C" int fn_pack_transpose(const void* w_lp, void* wt_lp, const int32_t* table, int n_layers, int max_layer_elems, int dtype, void* stream) {
    FN_REQUIRE(dtype == FN_BF16 || dtype == FN_F16, "dtype %d unsupported", dtype);
    FN_REQUIRE(w_lp && wt_lp && table && n_layers > 0 && max_layer_elems > 0, "pack_transpose: bad arguments");
    int gx = cdiv(max_layer_elems, 1024 * 4);
    if (gx > 128) gx = 128;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(pack_transpose_kernel<__bf16>, dim3(gx, n_layers), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)w_lp,
                       (unsigned short*)wt_lp, table);  // pure 16-bit moves: one instantiation serves both dtypes
    return check_launch("pack_transpose");
}

extern "C" int fn_fold_bn(const float* w, void* wf_lp, float* fold_bias, const float* beta, const float* moving_mean, const float* moving_var,
                          const int32_t* table, int n_layers, int max_layer_elems, float eps, int dtype, void* stream) {
    FN_REQUIRE(dtype == FN_BF16 || dtype == FN_F16, "dtype %d unsupported", dtype);
    FN_REQUIRE(w && wf_lp && fold_bias && beta && moving_mean && moving_var && table && n_layers > 0, "fold_bn: bad arguments");
    int gx = cdiv(max_layer_elems, 256 * 8);
    if (gx > 256) gx = 256;
    if (dtype == FN_BF16)
        hipLaunchKernelGGL(fold_bn_kernel<__bf16>, dim3(gx, n_layers), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)wf_lp, fold_bias, beta, moving_mean, moving_var, table, eps);
    else
        hipLaunchKernelGGL(fold_bn_kernel<_Float16>, dim3(gx, n_layers), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)wf_lp, fold_bias, beta, moving_mean, moving_var, table, eps);
    return check_launch("fold_bn");
}
