// Common device helpers for the msseg HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/msseg.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define MSSEG_DEVFN __device__ __forceinline__

// ---- error plumbing -------------------------------------------------------------
void msseg_set_error(const char* fmt, ...);
#define MSSEG_FAIL(code, ...)           \
    do {                                \
        msseg_set_error(__VA_ARGS__);   \
        return (code);                  \
    } while (0)
#define MSSEG_CHECK_LAUNCH(name)                                              \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess) MSSEG_FAIL(MSSEG_ELAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// ---- optional event pair around ONE kernel launch (ktimer.hip; bench.py's per-kernel roofline) -------
bool msseg_ktimer_on();
int msseg_ktimer_begin(const char* name, hipStream_t stream);
void msseg_ktimer_end(int slot, hipStream_t stream);
#define MSSEG_KTIMED(name, stream, launch_stmt)                               \
    do {                                                                      \
        const int kt__ = msseg_ktimer_on() ? msseg_ktimer_begin(name, stream) : -1; \
        launch_stmt;                                                          \
        if (kt__ >= 0) msseg_ktimer_end(kt__, stream);                        \
    } while (0)

// ---- dynamic LDS above 64 KB ---------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a (kernel, device) pair.  One of these lives as a
// function-local static next to each kernel instantiation: a bit per device ordinal, set after the first successful
// call on that device; two threads racing on the first call both make the same (idempotent) setting.
#include <atomic>
struct msseg_lds_attr_once {
    std::atomic<unsigned long long> done{0};
    bool ensure(const void* kern, int bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        const bool tracked = dev >= 0 && dev < 64;
        if (tracked && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return true;
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
        if (tracked) done.fetch_or(1ull << dev, std::memory_order_release);
        return true;
    }
};

// ---- per-dtype traits -------------------------------------------------------------
// A "chunk" is 16 bytes of consecutive channels: 8 bf16 or 4 f32.  One MFMA k-group
// (4 lane-quarters x 1 chunk) therefore spans CB = 4*EPC channels: 32 (bf16) / 16 (f32).
template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int EPC = 4;
    static constexpr int CODE = MSSEG_F32;
    static MSSEG_DEVFN float ld(const float* p) { return *p; }
    static MSSEG_DEVFN void st(float* p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
    static constexpr int EPC = 8;
    static constexpr int CODE = MSSEG_BF16;
    static MSSEG_DEVFN float ld(const bf16_t* p) { return (float)*p; }
    static MSSEG_DEVFN void st(bf16_t* p, float v) { *p = (bf16_t)v; }
};

// D[row = A-row (l>>4)*4+reg][col = B-col l&15] += A[row][k] * B[k][col] over one 16-byte chunk
// per lane-quarter: 32 k (bf16, one 16x16x32 MFMA) or 16 k (f32, four 16x16x4 MFMAs whose
// k-order is permuted identically on both operands).
template <typename T> MSSEG_DEVFN void mma_chunk(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b);
template <> MSSEG_DEVFN void mma_chunk<bf16_t>(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                  acc, 0, 0, 0);
}
template <> MSSEG_DEVFN void mma_chunk<float>(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
    f32x4_t af = __builtin_bit_cast(f32x4_t, a), bf = __builtin_bit_cast(f32x4_t, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], acc, 0, 0, 0);
}

// store 4 consecutive channel values
template <typename T> MSSEG_DEVFN void store4(T* p, const f32x4_t& v);
template <> MSSEG_DEVFN void store4<float>(float* p, const f32x4_t& v) { *(f32x4_t*)p = v; }
template <> MSSEG_DEVFN void store4<bf16_t>(bf16_t* p, const f32x4_t& v) {
    bf16x4_t o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *(bf16x4_t*)p = o;
}

MSSEG_DEVFN float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- deterministic grid-wide reductions: "last block finalises" -----------------------------------
// Every block stores its partial results with plain stores, then calls this.  Exactly one block (the
// last to arrive) gets `true` and may read all partials with plain loads.  Placement-independent
// agent-scope release/acquire (cdna_hip_programming.md Guideline 16); the counter is left at zero.
MSSEG_DEVFN bool grid_last_block(unsigned int* counter, unsigned int total_blocks, int* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (prev == total_blocks - 1u) ? 1 : 0;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *lds_flag = last;
    }
    __syncthreads();
    return *lds_flag != 0;
}

// Fixed-order sum of R partial rows of L floats (L % 4 == 0, L <= 4 * NT) by one block of NT threads: float4-wide,
// every thread with up to 8 independent loads in flight, row slots then combined through LDS in a fixed order
// (bit-reproducible).  The L totals end up in lds[NT * 4 ..]; lds: >= NT * 4 + L floats, 16-byte aligned.
template <int NT>
MSSEG_DEVFN void block_rows_sum(const float* rows, int R, int L, float* lds, long long stride = 0) {
    // stride: distance between rows in floats (a multiple of 4; 0 = L): sums an L-wide column slice of wider rows
    const int tid = threadIdx.x, q = L >> 2;
    const int SL = NT / q;
    const int c4 = tid % q, slot = tid / q;
    const long long q_stride = (stride ? stride : (long long)L) >> 2;
    f32x4_t* l4 = (f32x4_t*)lds;
    if (slot < SL) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const f32x4_t* src = (const f32x4_t*)rows + c4;
#pragma unroll 8
        for (int x = slot; x < R; x += SL) acc += src[(long long)x * q_stride];
        l4[slot * q + c4] = acc;
    }
    __syncthreads();
    if (tid < q) {
        f32x4_t t = {0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < SL; ++sl) t += l4[sl * q + tid];
        l4[NT + tid] = t;
    }
    __syncthreads();
}

#define MSSEG_SCRATCH_COUNTER_BYTES 256
#define MSSEG_STATS_NMAX 8

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }
