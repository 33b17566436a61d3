// Distance matrix, online triplet selection, triplet loss, softmax cross-entropy.
//  * fn_pairwise_sqdist restates facenet/statistics.py:22-57 (pairwise_similarities) on device:
//    one wave per (row i, block of columns), dot products by wavefront reduction.
//  * triplet selection / loss are build-defined (arXiv 1503.03832 sec. 3); the
//    counter-based hash below is shared bit-for-bit with oracle/facenet_oracle.py:hash_u32 so the
//    selected indices can be compared exactly.
//  * softmax cross-entropy = SparseCategoricalCrossentropy(from_logits=True), apps/train_softmax.py:91.
#include "common.h"
#include "../../include/facenet_hip.h"

namespace fn {

__device__ __forceinline__ int f2ord_i(float f) {  // order-preserving float -> int
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}

// out[i][j] for i<n, j<m ; range[0]=ord(min dot) via atomicMin, range[1]=ord(max dot) via atomicMax
__global__ __launch_bounds__(256) void pairwise_kernel(const float* __restrict__ xa, const float* __restrict__ xb, float* __restrict__ out,
                                                       int* __restrict__ range, int n, int m, int E, int metric) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.y;
    const int jb = (blockIdx.x * 4 + wave) * 16;
    if (jb >= m) return;
    // this lane's slice of row i stays in registers (E <= 64*8).  Every sum below is an explicit fma chain in ONE fixed order,
    // and a row's squared norm is computed by the same code whether the row is an `a` or a `b`: out[i][j] and out[j][i] are
    // then the same bits (|a|^2 + |b|^2 commutes, the dot product's terms commute), which the triplet selection relies on
    // when it reads columns instead of rows.  Left to the compiler's contraction choices the two norms differed in the last bit.
    float a[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) a[t] = (lane + 64 * t < E) ? xa[(long)i * E + lane + 64 * t] : 0.f;
    float asq = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) asq = __fmaf_rn(a[t], a[t], asq);
    asq = wave_sum(asq);
    float lo = 3e38f, hi = -3e38f;
    for (int j = jb; j < min(m, jb + 16); ++j) {
        float d = 0.f, bsq = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float b = (lane + 64 * t < E) ? xb[(long)j * E + lane + 64 * t] : 0.f;
            d = __fmaf_rn(a[t], b, d);
            bsq = __fmaf_rn(b, b, bsq);
        }
        d = wave_sum(d);
        float r;
        if (metric == 2) {
            bsq = wave_sum(bsq);
            r = fmaxf(__fmaf_rn(-2.f, d, asq + bsq), 0.f);
        } else {
            lo = fminf(lo, d);
            hi = fmaxf(hi, d);
            const float s = fminf(fmaxf(d, -1.f), 1.f);   // statistics.py:45-46
            r = (metric == 0) ? 2.f * (1.f - s) : acosf(s);  // :48-53
        }
        if (lane == 0) out[(long)i * m + j] = r;
    }
    if (range && lane == 0 && metric != 2) {
        atomicMin(&range[0], f2ord_i(lo));
        atomicMax(&range[1], f2ord_i(hi));
    }
}

__device__ __forceinline__ unsigned hash_mix(unsigned h, unsigned v) {
    h ^= v;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ unsigned hash_u32(unsigned a, unsigned b, unsigned c) {
    return hash_mix(hash_mix(hash_mix(0x9E3779B9u, a), b), c);
}

// The distance matrix is bitwise symmetric (same products, same order), so thread a reads COLUMN a
// (dist[j*n + a]: consecutive threads -> consecutive addresses) instead of its own row.
// n <= 1024.  scratch (int32, global): per pair q: [a, p, neg, key(u32), cls] at scratch[8 + 5*q].
// info[0] = #pairs, info[1] = #valid (candidate found), info[2] = 1 if fewer pairs than requested triplets,
// info[3] = call counter: the effective seed is seed + info[3], so a HIP-graph replay (frozen kernel arguments)
// still draws fresh negatives every step.  The caller zeroes info once.
// Three launches: (A) one workgroup enumerates the anchor-positive pairs, (B) one WAVE per pair -- spread over the chip -- picks
// the negative, (C) one workgroup ranks the pairs and writes the triplets.  As a single workgroup phase B ran 17 rounds of
// dependent loads (~70 us of the 85 us kernel).
__global__ __launch_bounds__(1024) void select_pairs_kernel(const int* __restrict__ labels, int n, int* __restrict__ info) {
    __shared__ int s_lab[1024];
    __shared__ int s_off[1025];
    const int tid = threadIdx.x;
    int* rec = info + 8;
    for (int i = tid; i < n; i += 1024) s_lab[i] = labels[i];
    __syncthreads();
    // pairs per anchor (a < p, same label), exclusive prefix -> pair ids in row-major order
    int cnt = 0;
    if (tid < n)
        for (int p = tid + 1; p < n; ++p) cnt += (s_lab[p] == s_lab[tid]);
    // inclusive prefix of the per-anchor counts: wave scan + carry of the wave totals (a serial loop of thread 0 over the n entries
    // was ~7 us of dependent LDS round trips)
    {
        __shared__ int s_wtot[16];
        const int lane = tid & 63, wv = tid >> 6;
        int v = tid < n ? cnt : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_up(v, d);
            if (lane >= d) v += u;
        }
        if (lane == 63) s_wtot[wv] = v;
        __syncthreads();
        int carry = 0;
        for (int w = 0; w < wv; ++w) carry += s_wtot[w];
        if (tid < n) s_off[tid + 1] = v + carry;
        if (tid == 0) s_off[0] = 0;
    }
    __syncthreads();
    if (tid < n) {
        int q = s_off[tid];
        for (int p = tid + 1; p < n; ++p)
            if (s_lab[p] == s_lab[tid]) { rec[5 * q + 0] = tid; rec[5 * q + 1] = p; ++q; }
    }
    if (tid == 0) {
        info[0] = s_off[n];
        info[1] = 0;
    }
}

__global__ __launch_bounds__(256) void select_negatives_kernel(const float* __restrict__ dist, const int* __restrict__ labels, int n,
                                                               float alpha, unsigned seed0, int semi_hard, int* __restrict__ info) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= info[0]) return;
    const unsigned seed = seed0 + (unsigned)info[3];
    int* rec = info + 8;
    const int a = rec[5 * q + 0], p = rec[5 * q + 1];
    const int la = labels[a];
    const float dap = dist[(long)p * n + a];
    int c = 0, others = 0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const bool oth = (j < n) && (labels[j] != la);
        const float daj = oth ? dist[(long)j * n + a] : 0.f;
        const bool cand = oth && (daj - dap < alpha) && (!semi_hard || daj > dap);
        c += __popcll(__ballot(cand));
        others += __popcll(__ballot(oth));
    }
    const bool use_cand = c > 0;
    int want = use_cand ? (int)(hash_u32(seed, (unsigned)q, 0u) % (unsigned)c)
                        : (others > 0 ? (int)(hash_u32(seed, (unsigned)q, 2u) % (unsigned)others) : -1);
    int neg = -1;
    for (int j0 = 0; j0 < n && want >= 0; j0 += 64) {
        const int j = j0 + lane;
        const bool oth = (j < n) && (labels[j] != la);
        const float daj = oth ? dist[(long)j * n + a] : 0.f;
        const bool hit = use_cand ? (oth && (daj - dap < alpha) && (!semi_hard || daj > dap)) : oth;
        const unsigned long long m = __ballot(hit);
        const int cnt = __popcll(m);
        if (want < cnt) {
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            const unsigned long long sel = __ballot(hit && rank == want);
            neg = j0 + (int)__ffsll((long long)sel) - 1;
            want = -1;
        } else {
            want -= cnt;
        }
    }
    if (lane == 0) {
        rec[5 * q + 2] = neg;
        rec[5 * q + 3] = (int)hash_u32(seed, (unsigned)q, 1u);
        rec[5 * q + 4] = use_cand ? 0 : 1;
        if (use_cand) atomicAdd(&info[1], 1);
    }
}

__global__ __launch_bounds__(1024) void select_rank_kernel(int T, int* __restrict__ triplets, int* __restrict__ info) {
    const int tid = threadIdx.x;
    const int* rec = info + 8;
    const int Q = info[0];
    // (cls, key) of the first 4096 pairs staged in LDS as one 64-bit sort key: the O(Q^2) comparison loop then reads LDS, not global
    __shared__ unsigned long long s_key[4096];
    for (int q = tid; q < min(Q, 4096); q += 1024)
        s_key[q] = ((unsigned long long)(unsigned)rec[5 * q + 4] << 32) | (unsigned)rec[5 * q + 3];
    __syncthreads();
    // rank by (cls, key, q); rank < T wins slot `rank`
    for (int q = tid; q < Q; q += 1024) {
        const int cls = rec[5 * q + 4];
        const unsigned key = (unsigned)rec[5 * q + 3];
        const unsigned long long mine = ((unsigned long long)(unsigned)cls << 32) | key;
        int rank = 0;
        const int QL = min(Q, 4096);
        for (int r = 0; r < QL; ++r) {
            const unsigned long long k2 = s_key[r];
            rank += (k2 < mine || (k2 == mine && r < q)) ? 1 : 0;
        }
        for (int r = QL; r < Q; ++r) {
            const int c2 = rec[5 * r + 4];
            const unsigned k2 = (unsigned)rec[5 * r + 3];
            const bool before = (c2 < cls) || (c2 == cls && (k2 < key || (k2 == key && r < q)));
            rank += before ? 1 : 0;
        }
        if (rank < T && rec[5 * q + 2] >= 0) {
            triplets[3 * rank + 0] = rec[5 * q + 0];
            triplets[3 * rank + 1] = rec[5 * q + 1];
            triplets[3 * rank + 2] = rec[5 * q + 2];
        }
    }
    if (tid == 0) {
        info[2] = (Q < T) ? 1 : 0;
        info[3] = info[3] + 1;
    }
}

// emb rows (a0,p0,n0,a1,...), fp32 [3T,E].  One wave per triplet.
__global__ __launch_bounds__(256) void triplet_loss_kernel(const float* __restrict__ emb, float* __restrict__ demb, acc_t* __restrict__ loss,
                                                           int T, int E, float alpha) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    const float* a = emb + (long)(3 * t) * E;
    const float* p = a + E;
    const float* ng = p + E;
    float pos = 0.f, neg = 0.f;
    for (int k = lane; k < E; k += 64) {
        const float dp = a[k] - p[k], dn = a[k] - ng[k];
        pos += dp * dp;
        neg += dn * dn;
    }
    pos = wave_sum(pos);
    neg = wave_sum(neg);
    const float l = pos - neg + alpha;
    const float on = l > 0.f ? 1.f : 0.f;
    if (lane == 0) acc_add<ACC_GRAD>(loss, fmaxf(l, 0.f) / (float)T);
    if (demb) {
        const float s = 2.f * on / (float)T;
        float* da = demb + (long)(3 * t) * E;
        for (int k = lane; k < E; k += 64) {
            da[k] = s * (ng[k] - p[k]);
            da[E + k] = s * (p[k] - a[k]);
            da[2 * E + k] = s * (a[k] - ng[k]);
        }
    }
}

// one workgroup per row: loss += (lse - logit[label])/N ; dlogits = (softmax - onehot) * grad_scale (low precision, padded cols = 0)
template <typename T>
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ logits, int ld, const int* __restrict__ labels,
                                                           acc_t* __restrict__ loss, unsigned short* __restrict__ dlogits, int ld_d,
                                                           acc_t* __restrict__ dbias, int N, int C, float grad_scale) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (long)row * ld;
    float mx = -3e38f;
    for (int c = tid; c < C; c += 256) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = tid; c < C; c += 256) s += __expf(x[c] - mx);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = red[0] + red[1] + red[2] + red[3];
    // an out-of-range class index never indexes the logits: the loss becomes NaN (what TF's GPU kernel returns; its CPU kernel
    // raises, which Trainer.set_images does on the host) and the row contributes no one-hot term
    const int lab = labels[row];
    const bool lab_ok = lab >= 0 && lab < C;
    // (a NaN cannot live in the fixed-point sum: an invalid label sets the flag word, loss_finish_kernel then reports NaN)
    if (tid == 0) {
        if (lab_ok) acc_add<ACC_GRAD>(loss, (logf(s) + mx - x[lab]) / (float)N);
        else reinterpret_cast<volatile unsigned*>(loss)[-1] = 1u;
    }
    if (dlogits) {
        const float inv = 1.f / s;
        unsigned short* d = dlogits + (long)row * ld_d;
        for (int c = tid; c < ld_d; c += 256) {
            float g = 0.f;
            if (c < C) {
                g = (__expf(x[c] - mx) * inv - (c == lab ? 1.f : 0.f)) * grad_scale;
                if (dbias) acc_add<ACC_GRAD>(&dbias[c], g);
            }
            d[c] = LP<T>::from_f32(g);
        }
    }
}

// loss words: [0] = the loss (fp32), [1] = invalid-input flag, [2..3] = fixed-point accumulator (ACC_GRAD) the rows add into
__global__ void loss_finish_kernel(float* loss) {
    const float v = acc_get<ACC_GRAD>(*reinterpret_cast<const acc_t*>(loss + 2));
    loss[0] = reinterpret_cast<const unsigned*>(loss)[1] ? __builtin_nanf("") : v;
}

}  // namespace fn
using namespace fn;

extern "C" int fn_pairwise_sqdist(const float* xa, const float* xb, float* out, float* range, int n, int m, int E, int metric, void* stream) {
    FN_REQUIRE(xa && xb && out && n > 0 && m > 0 && E > 0 && E <= 512, "pairwise_sqdist: bad arguments (E must be <= 512)");
    FN_REQUIRE(metric >= 0 && metric <= 2, "Undefined similarity metric %d", metric);  // statistics.py:55
    hipStream_t st = (hipStream_t)stream;
    if (range) {
        fill_words(range, 0x7f7fffffu, 0x80800000u, 2, st);   // ord(+FLT_MAX), ord(-FLT_MAX); a kernel node, see fill_words
    }
    hipLaunchKernelGGL(pairwise_kernel, dim3(cdiv(m, 64), n), dim3(256), 0, st, xa, xb, out, (int*)range, n, m, E, metric);
    return check_launch("pairwise_sqdist");
}

extern "C" int fn_select_triplets(const float* dist, const int32_t* labels, int n, float alpha, int nrof_triplets, uint32_t seed,
                                  int semi_hard, int32_t* triplets, int32_t* info, void* stream) {
    FN_REQUIRE(dist && labels && triplets && info && n > 1 && n <= 1024 && nrof_triplets > 0, "select_triplets: bad arguments (n <= 1024)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(select_pairs_kernel, dim3(1), dim3(1024), 0, st, labels, n, info);
    const int max_pairs = n * (n - 1) / 2;      // upper bound; waves beyond info[0] leave at once
    hipLaunchKernelGGL(select_negatives_kernel, dim3(cdiv(max_pairs, 4)), dim3(256), 0, st, dist, labels, n, alpha, seed, semi_hard, info);
    hipLaunchKernelGGL(select_rank_kernel, dim3(1), dim3(1024), 0, st, nrof_triplets, triplets, info);
    return check_launch("select_triplets");
}

extern "C" int fn_triplet_loss_fwd_bwd(const float* emb, float* demb, float* loss, int T, int E, float alpha, void* stream) {
    FN_REQUIRE(emb && loss && T > 0 && E > 0, "triplet_loss: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    FN_REQUIRE(((uintptr_t)loss & 7) == 0, "triplet_loss: loss must be an 8-byte aligned fp32[4]");
    fill_words(loss, 0u, 0u, 4, st);
    hipLaunchKernelGGL(triplet_loss_kernel, dim3(cdiv(T, 4)), dim3(256), 0, st, emb, demb, reinterpret_cast<acc_t*>(loss + 2), T, E, alpha);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1), 0, st, loss);
    return check_launch("triplet_loss");
}

extern "C" int fn_softmax_xent_fwd_bwd(const float* logits, int ld, const int32_t* labels, float* loss, void* dlogits_lp, int ld_d, fn_acc_t* dbias_,
                                       int N, int C, float grad_scale, int dtype, void* stream) {
    acc_t* dbias = reinterpret_cast<acc_t*>(dbias_);
    FN_REQUIRE(((uintptr_t)loss & 7) == 0, "softmax_xent: loss must be an 8-byte aligned fp32[4]");
    FN_REQUIRE(dtype == FN_BF16 || dtype == FN_F16, "dtype %d unsupported", dtype);
    FN_REQUIRE(logits && labels && loss && N > 0 && C > 0 && ld >= C && (!dlogits_lp || ld_d >= C), "softmax_xent: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    fill_words(loss, 0u, 0u, 4, st);
    acc_t* lacc = reinterpret_cast<acc_t*>(loss + 2);
    if (dtype == FN_BF16)
        hipLaunchKernelGGL(softmax_xent_kernel<__bf16>, dim3(N), dim3(256), 0, st, logits, ld, labels, lacc, (unsigned short*)dlogits_lp, ld_d, dbias, N, C, grad_scale);
    else
        hipLaunchKernelGGL(softmax_xent_kernel<_Float16>, dim3(N), dim3(256), 0, st, logits, ld, labels, lacc, (unsigned short*)dlogits_lp, ld_d, dbias, N, C, grad_scale);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1), 0, st, loss);
    return check_launch("softmax_xent");
}
