// Face-to-face validation statistics on device: the class-pair confusion counts of
// facenet/statistics.py:111-138 (ConfidenceMatrix.__init__) with the class-balanced weights of
// SimilarityCalculator.evaluate (:92-103).  The reference walks all C(C+1)/2 class pairs in Python, calls
// pairwise_similarities (:22-57) for each and loops over 100 thresholds with np.count_nonzero: 700-1 550 s per
// validation of 26 489 embeddings in its own logs (models/20200724-231357/logs/report.txt:47,647).
//
// Here: one workgroup per class pair (i >= k).  Distances are computed in 32x32 image tiles from LDS-staged fp32
// embedding chunks (fp32 FMA: thresholds are compared exactly, so no low-precision MFMA here), every distance is
// binned once by upper_bound over the ascending thresholds into an LDS histogram, a prefix sum turns the histogram into
// "count(sims < threshold[n])" for all n at once, and the weighted tp/fn or fp/tn contributions go out as fp64 atomics.
#include "common.h"
#include "../../include/facenet_hip.h"

namespace fn {

__device__ __forceinline__ int f2ord_i2(float f) {
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}

constexpr int VT = 32;        // images per tile side
constexpr int VE = 64;        // embedding chunk
constexpr int VMAXT = 256;    // thresholds

__global__ __launch_bounds__(256) void confidence_kernel(const float* __restrict__ emb, const int* __restrict__ cls_start, int C, int E,
                                                         const float* __restrict__ thr, int T, int metric, double* __restrict__ out,
                                                         int* __restrict__ range) {
    __shared__ float sA[VT][VE + 1], sB[VT][VE + 1];
    __shared__ float sThr[VMAXT];
    __shared__ int sHist[VMAXT + 1];
    const int tid = threadIdx.x;
    // class pair (i >= k) from the linear block id
    const long b = blockIdx.x;
    int i = (int)((sqrtf(8.f * (float)b + 1.f) - 1.f) * 0.5f);
    while ((long)i * (i + 1) / 2 > b) --i;
    while ((long)(i + 1) * (i + 2) / 2 <= b) ++i;
    const int k = (int)(b - (long)i * (i + 1) / 2);
    const int a0 = cls_start[i], na = cls_start[i + 1] - a0;
    const int b0 = cls_start[k], nb = cls_start[k + 1] - b0;
    const long P = (i == k) ? (long)na * (na - 1) / 2 : (long)na * nb;
    if (P < 1) return;                                   // statistics.py:126-127
    for (int t = tid; t < T; t += 256) sThr[t] = thr[t];
    for (int t = tid; t <= T; t += 256) sHist[t] = 0;
    __syncthreads();
    const int ar = tid >> 3, bc = (tid & 7) * 4;         // thread -> row ar of the A tile, 4 consecutive rows of the B tile
    float lo = 3e38f, hi = -3e38f;
    for (int ta = 0; ta < na; ta += VT)
        for (int tb = 0; tb < nb; tb += VT) {
            if (i == k && tb + VT - 1 <= ta) continue;   // tile entirely on/below the diagonal: no pair with b > a
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int e0 = 0; e0 < E; e0 += VE) {
                __syncthreads();
                for (int t = tid; t < VT * VE; t += 256) {
                    const int r = t / VE, c = t - r * VE;
                    sA[r][c] = (ta + r < na && e0 + c < E) ? emb[(long)(a0 + ta + r) * E + e0 + c] : 0.f;
                    sB[r][c] = (tb + r < nb && e0 + c < E) ? emb[(long)(b0 + tb + r) * E + e0 + c] : 0.f;
                }
                __syncthreads();
#pragma unroll 8
                for (int c = 0; c < VE; ++c) {
                    const float av = sA[ar][c];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(av, sB[bc + j][c], acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ia = ta + ar, ib = tb + bc + j;
                if (ia >= na || ib >= nb || (i == k && ib <= ia)) continue;   // strict upper triangle (:32-34)
                const float s = acc[j];
                lo = fminf(lo, s);
                hi = fmaxf(hi, s);
                const float sc = fminf(fmaxf(s, -1.f), 1.f);                  // :45-46
                const float d = (metric == 0) ? 2.f * (1.f - sc) : acosf(sc); // :48-53
                int l = 0, h = T;                                             // first n with thr[n] > d  (d < thr[n] counts)
                while (l < h) {
                    const int m = (l + h) >> 1;
                    if (sThr[m] > d) h = m; else l = m + 1;
                }
                atomicAdd(&sHist[l], 1);
            }
        }
    lo = -wave_max(-lo);
    hi = wave_max(hi);
    if ((tid & 63) == 0 && range) {
        atomicMin(&range[0], f2ord_i2(lo));
        atomicMax(&range[1], f2ord_i2(hi));
    }
    __syncthreads();
    if (tid == 0) {                                       // prefix: sHist[n] = count(sims < thr[n])
        int run = 0;
        for (int n = 0; n < T; ++n) { run += sHist[n]; sHist[n] = run; }
    }
    __syncthreads();
    const double w = (double)P * ((i == k) ? (double)C : (double)C * (C - 1) * 0.5);   // :93-101
    for (int n = tid; n < T; n += 256) {
        const double c = (double)sHist[n];
        if (i == k) {
            atomicAdd(&out[0 * T + n], c / w);                    // tp
            atomicAdd(&out[3 * T + n], ((double)P - c) / w);      // fn
        } else {
            atomicAdd(&out[2 * T + n], c / w);                    // fp
            atomicAdd(&out[1 * T + n], ((double)P - c) / w);      // tn
        }
    }
}

}  // namespace fn
using namespace fn;

extern "C" int fn_confidence_counts(const float* emb, const int32_t* cls_start, int C, int E, const float* thresholds, int T, int metric,
                                    double* out, int32_t* range, void* stream) {
    FN_REQUIRE(emb && cls_start && thresholds && out && C > 0 && E > 0 && T > 0 && T <= VMAXT, "confidence_counts: bad arguments (T <= 256)");
    FN_REQUIRE(metric == 0 || metric == 1, "Undefined similarity metric %d", metric);   // statistics.py:258-260
    hipStream_t st = (hipStream_t)stream;
    fill_words(out, 0u, 0u, 2 * 4 * T, st);                              // 4*T doubles; kernel nodes, see fill_words
    if (range) fill_words(range, 0x7f7fffffu, 0x80800000u, 2, st);
    const long pairs = (long)C * (C + 1) / 2;
    FN_REQUIRE(pairs < (1L << 31), "confidence_counts: too many classes");
    hipLaunchKernelGGL(confidence_kernel, dim3((unsigned)pairs), dim3(256), 0, st, emb, cls_start, C, E, thresholds, T, metric, out, (int*)range);
    return check_launch("confidence_counts");
}
