// Shared device/host helpers for the gfx950 (MI355X, CDNA4) FaceNet hot path.
// Wave = 64 lanes everywhere; nothing here is portable to other targets on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace fn {

// ---- status / error plumbing (C ABI returns int, fn_last_error() gives text) ----
enum { FN_OK = 0, FN_EINVAL = -1, FN_ELAUNCH = -2, FN_EUNSUPPORTED = -3 };
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define FN_REQUIRE(cond, ...)                  \
    do {                                       \
        if (!(cond)) {                         \
            fn::set_error(__VA_ARGS__);        \
            return fn::FN_EINVAL;              \
        }                                      \
    } while (0)

// ---- low-precision storage types: bf16 (training) and f16 (embedding path) ----
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

enum { DT_BF16 = 0, DT_F16 = 1 };

// Pointers that a kernel reads out of a descriptor table in memory (grouped launches) have no known address space, and the
// compiler falls back to FLAT loads.  Those count on lgkmcnt as well as vmcnt, so every wait for an LDS read also drains
// the global loads in flight.  Hot-loop loads go through explicitly global pointers.
#define FN_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ u32x4 load_global_b128(const unsigned short* base, long elem_off) {
    return *(const FN_GLOBAL u32x4*)((const FN_GLOBAL unsigned short*)base + elem_off);
}

template <typename T> struct LP;  // low-precision traits
template <> struct LP<__bf16> {
    typedef bf16x8 vec8;
    typedef bf16x4 vec4;
    static __device__ __forceinline__ float to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
    static __device__ __forceinline__ unsigned short from_f32(float f) {
        __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
        return __builtin_bit_cast(unsigned short, h);
    }
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct LP<_Float16> {
    typedef f16x8 vec8;
    typedef f16x4 vec4;
    static __device__ __forceinline__ float to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
    static __device__ __forceinline__ unsigned short from_f32(float f) {
        _Float16 h = (_Float16)f;
        return __builtin_bit_cast(unsigned short, h);
    }
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// 8 packed 16-bit values <-> 8 floats
template <typename T> __device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = LP<T>::to_f32((unsigned short)(v[i] & 0xffffu));
        f[2 * i + 1] = LP<T>::to_f32((unsigned short)(v[i] >> 16));
    }
}
template <typename T> __device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        v[i] = (unsigned)LP<T>::from_f32(f[2 * i]) | ((unsigned)LP<T>::from_f32(f[2 * i + 1]) << 16);
    return v;
}

// q = m / d, r = m % d for 0 <= m < 2^24 with inv = 1.0f/d (host checks the range).
__device__ __forceinline__ void fast_divmod(int m, int d, float inv, int& q, int& r) {
    q = (int)((float)m * inv);
    r = m - q * d;
    const int lo = r < 0 ? 1 : 0, hi = r >= d ? 1 : 0;   // branch-free fix-up (a branch here ends up around the loads that follow)
    q += hi - lo;
    r += (lo - hi) * d;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- order-independent accumulation ----------------------------------------------------------------------------------------
// Sums that many workgroups contribute to (BatchNorm batch statistics, the BatchNorm-backward sums, bias gradients, losses) are
// accumulated as 64-bit FIXED-POINT integers: every contribution is an fp32 partial sum computed in a fixed order, converted once
// (round to nearest) and added with an integer atomic.  Integer addition is associative, so the total has the same bits whatever
// order the workgroups arrive in -- fp32 atomics gave training results that differed from run to run (and two data-parallel
// replicas that were only equal "within the atomics' noise").  Two scales: forward statistics (sums of activations and their
// squares over up to 5e5 pixels: |total| < 2^43 = 8.8e12, resolution 2^-20) and gradient sums (|total| < 2^23 = 8.4e6, resolution
// 2^-40 = 9e-13).  fn_acc_t in the C ABI.
typedef int64_t acc_t;      // == fn_acc_t of the C ABI
enum { ACC_STAT = 20, ACC_GRAD = 40 };
template <int FRAC> __device__ __forceinline__ void acc_add(acc_t* p, float v) {
    atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double2ll_rn((double)v * (double)(1ll << FRAC)));
}
template <int FRAC> __device__ __forceinline__ float acc_get(acc_t a) { return (float)((double)a * (1.0 / (double)(1ll << FRAC))); }

// BatchNorm batch statistics -> affine (center only): scale = rstd, shift = beta - mean*rstd.  One definition for the
// materialising kernel, the normalise-on-load operand path and fn_bn_finalize (the replicas hold fixed-point integers: their sum
// is exact in any order, so all three produce the same bits).
// (sum, sum of squares, count) -> mean, biased variance, scale = rstd, shift = beta - mean * rstd.  ONE definition with every
// rounding spelled out (no fused multiply-add left to the compiler's contraction choices): the materialising kernel, the
// normalise-on-load prologue and fn_bn_finalize all call it and publish the same bits.
__device__ __forceinline__ void bn_affine_from_sums(float s1, float s2, int count, float eps, float beta, float& scale, float& shift,
                                                    float& mean, float& var) {
#pragma clang fp contract(off)      // (HIP's __fmul_rn & co. are plain operators: only the pragma keeps fused multiply-adds out)
    const float inv = 1.f / (float)count;
    mean = s1 * inv;
    const float ex2 = s2 * inv, m2 = mean * mean;
    var = fmaxf(ex2 - m2, 0.f);
    scale = rsqrtf(var + eps);
    const float ms = mean * scale;
    shift = beta - ms;
}

// moving statistic update (Keras: moving * momentum + batch * (1 - momentum)), every rounding spelled out for the same reason
__device__ __forceinline__ float bn_moving_update(float moving, float batch, float momentum) {
#pragma clang fp contract(off)
    const float a = moving * momentum, w = 1.f - momentum;
    const float b = batch * w;
    return a + b;
}

__device__ __forceinline__ void bn_batch_affine(const acc_t* __restrict__ stats, int c, int sq_off, int reps, int rep_stride, int count,
                                                float eps, float beta, float& scale, float& shift, float& mean, float& var) {
    acc_t s = 0, q = 0;       // replica sums are integer: exact, any order
    for (int rp = 0; rp < reps; ++rp) {
        s += stats[(long)rp * rep_stride + c];
        q += stats[(long)rp * rep_stride + sq_off + c];
    }
    bn_affine_from_sums(acc_get<ACC_STAT>(s), acc_get<ACC_STAT>(q), count, eps, beta, scale, shift, mean, var);
}

// Write-through stores (sc0 sc1) for tensors the NEXT kernel reads.  The eight XCDs' L2s are not coherent with each other: what a
// kernel leaves dirty in an L2 is written back at the kernel boundary, on the critical path of the next launch.  With write-through
// the bytes leave for memory while the kernel still runs (measured on the convolution epilogues alone: step 6.785 -> 6.738 ms; the
// streaming hint `nt` instead: 6.86).  Byte offsets are 32-bit against a descriptor over the tensor's base pointer.
enum { STORE_WT = 17 };      // aux / cache-policy operand of the buffer-store builtins: sc0 | sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void store_wt(const __amdgpu_buffer_rsrc_t rs, long byte_offset, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)byte_offset, 0, STORE_WT);
}

// Bijective XCD-aware remap: blocks b and b+8 share an XCD (and its 4 MiB L2).  Give each XCD a contiguous run of the
// work-item space; every kernel of the step uses the SAME rows->XCD partition (row fraction x/8 .. (x+1)/8 on XCD x), so
// what one kernel wrote is still in the L2 of the XCD that reads it in the next kernel (per-XCD L2s are not coherent:
// anything written on another XCD comes from the memory side).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Kernels that ask for more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised first.  The
// attribute is PER DEVICE: one flag per (launch site, device), set once, and a failing call is an error here rather than an
// unrelated launch failure later.
struct LdsOptIn {
    std::atomic<bool> done[32];
};
static inline int allow_big_lds(const void* kernel, LdsOptIn& state, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) {
        set_error("%s: no current HIP device (or more than 32 devices)", what);
        return FN_ELAUNCH;
    }
    if (state.done[dev].load(std::memory_order_relaxed)) return FN_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
        set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize, 160 KiB) failed on device %d: %s", what, dev, hipGetErrorString(e));
        return FN_ELAUNCH;
    }
    state.done[dev].store(true, std::memory_order_relaxed);
    return FN_OK;
}

// Re-initialisation of small per-call state words (statistics, losses, min/max) inside launch functions that may be captured
// into a HIP graph.  A kernel, not hipMemsetAsync: memset NODES of a replayed graph were observed (ROCm 7.2, MI355X) to be
// skipped or reordered depending on where the caching allocator had placed the words (tools/dev_minergraph3.py: the image
// min/max of the previous replay leaked into the next one), while kernel nodes keep stream order.
static __global__ void fill_words_kernel(unsigned* __restrict__ p, unsigned v0, unsigned v1, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = (i & 1) ? v1 : v0;
}
static inline void fill_words(void* p, unsigned v0, unsigned v1, int n, hipStream_t st) {
    hipLaunchKernelGGL(fill_words_kernel, dim3(n > 4096 ? 16 : 1), dim3(256), 0, st, (unsigned*)p, v0, v1, n);
}

}  // namespace fn
