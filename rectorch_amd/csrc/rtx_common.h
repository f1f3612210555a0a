// rtx_common.h -- shared device/host helpers for librectorch_hip (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

typedef uint16_t bf16_t;  // raw bfloat16 bits; arithmetic type of the throughput mode
struct fp8_t { uint8_t v; };  // raw OCP e4m3 bits (gfx950); only the EASE Gram matrix uses it

// ---------------------------------------------------------------------------------------------
// error handling: every C-ABI entry point returns 0 or a negative RTX_E* code, message in a
// thread-local string (rtx_last_error()).
// ---------------------------------------------------------------------------------------------
#define RTX_OK 0
#define RTX_EINVAL (-1)
#define RTX_EHIP (-2)
#define RTX_ENOMEM (-3)
#define RTX_ESTATE (-4)

void rtx_set_error(const char* fmt, ...);

#define RTX_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            rtx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return RTX_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

#define RTX_CHECK(cond, code, ...)  \
    do {                            \
        if (!(cond)) {              \
            rtx_set_error(__VA_ARGS__); \
            return (code);          \
        }                           \
    } while (0)

#define RTX_TRY(expr)           \
    do {                        \
        int _rc = (expr);       \
        if (_rc != RTX_OK) return _rc; \
    } while (0)

// ---------------------------------------------------------------------------------------------
// padding rule: every GEMM operand dimension d is stored padded to PAD(d) = roundup(d + 1, 128).
// The "+1" guarantees a spare row/column at index d: the transposed activation buffers keep a row of
// ones there, so the weight-gradient GEMM emits the bias gradient as one extra output column.
// ---------------------------------------------------------------------------------------------
static inline int rtx_pad(int d) { return ((d + 1 + 127) / 128) * 128; }
static inline int rtx_pad_batch(int b) { return ((b + 127) / 128) * 128; }

// ---------------------------------------------------------------------------------------------
// element traits
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t h)
{
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)h) << 16;
    return c.f;
}

__host__ __device__ __forceinline__ bf16_t f32_to_bf16(float f)
{
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                          // round to nearest even
    return (bf16_t)(u >> 16);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    __host__ __device__ static __forceinline__ float from(float v) { return v; }
    __host__ __device__ static __forceinline__ float to(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    __host__ __device__ static __forceinline__ bf16_t from(float v) { return f32_to_bf16(v); }
    __host__ __device__ static __forceinline__ float to(bf16_t v) { return bf16_to_f32(v); }
};

#ifdef __HIPCC__
// four consecutive elements of T from four floats: one 16-byte (f32) or 8-byte (bf16) store
template <typename T> __device__ __forceinline__ void store4(T* dst, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* dst, float a, float b, float c, float d)
{
    *(float4*)dst = make_float4(a, b, c, d);
}
// two floats -> two bf16 in one dword: gfx950's v_cvt_pk_bf16_f32 (round to nearest even, as f32_to_bf16 above)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    typedef __attribute__((ext_vector_type(2))) __bf16 rtx_bf16x2;
    rtx_bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* dst, float a, float b, float c, float d)
{
    uint2 u;
    u.x = pack_bf16x2(a, b);
    u.y = pack_bf16x2(c, d);
    *(uint2*)dst = u;
}

// Consumer side of a cross-stream dependency folded into a kernel's prologue (rtx_gemm.h: RtxGemm::wait_word).  The guide's hand-off
// form (MI355X_MICROARCH.md, "Consumer, always"): ONE lane polls the word with relaxed agent-scope loads, then ONE agent-scope
// acquire -- unconditionally: the ordering must not rest on WHEN the number was stored relative to this kernel's dispatch -- then a
// workgroup barrier, then plain loads (the fence invalidates this CU's L1, which is what every wave of the workgroup reads through).
// Bounded: a producer that never arrives traps after 20 s instead of hanging the device.
__device__ __forceinline__ void rtx_fold_wait(const unsigned* word, unsigned seq)
{
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000000ull) __builtin_trap();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
#endif

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (dropout keep-mask and the reparameterisation noise).
// One call -> 4 x 32 random bits as a pure function of (seed, offset, index).
// ---------------------------------------------------------------------------------------------
struct Philox4 {
    uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ uint32_t rtx_mulhi32(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t offset, uint64_t index)
{
    uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32);
    uint32_t c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = rtx_mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = rtx_mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 o = {c0, c1, c2, c3};
    return o;
}

// uniform in [0,1) with 24 bits
__host__ __device__ __forceinline__ float rtx_u01(uint32_t bits) { return (float)(bits >> 8) * (1.0f / 16777216.0f); }

// dropout decision for element `index` of the call identified by (seed, offset): keep iff u >= p
__host__ __device__ __forceinline__ bool rtx_dropout_keep(uint64_t seed, uint64_t offset, uint64_t index, float p)
{
    return rtx_u01(philox4x32_10(seed, offset, index).x) >= p;
}

// standard normal for element `index` (Box-Muller on two of the four words)
__host__ __device__ __forceinline__ float rtx_normal(uint64_t seed, uint64_t offset, uint64_t index)
{
    const Philox4 r = philox4x32_10(seed, offset ^ 0x5851F42D4C957F2DULL, index);
    const float u1 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
    const float u2 = rtx_u01(r.y);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}
