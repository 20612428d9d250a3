// Depthwise Conv3d 3x3x3 (stride 1, pad 1, groups = channels) on channels-last tensors, gfx950.
//
// Replaces nn.Conv3d(C, C, 3, padding=1, groups=C) of the SwinDepth MLP (/root/reference/models/backbones/swindepth.py:36-41,
// 56-65: fc1 -> GELU -> 3 x (depthwise conv -> BatchNorm3d -> GELU) -> fc2) and its two gradients.  A depthwise conv
// has 27 MACs per output element -- no matrix shape to put on MFMA; the three kernels are HBM / cache streaming work:
//
//   forward / input gradient : one thread per (voxel, 16-byte channel chunk); the 27 neighbour chunks come through the
//                              vector cache (consecutive lanes = consecutive chunks of one voxel, consecutive waves =
//                              neighbouring voxels), weights from a tap-major fp32 table [27][C]; the input gradient is
//                              the same kernel with the taps mirrored.
//   weight + bias gradient   : lane = (channel chunk, kd plane), wave = voxel phase; every lane keeps 9 taps x chunk
//                              accumulators in registers over its voxels; waves are added in a fixed order through
//                              LDS, workgroups leave partial rows [28][C] that a second kernel adds in row order
//                              (deterministic, no atomics).
#include "common.h"

namespace {

template <typename T> struct Chunk;
template <> struct Chunk<bf16_t> {
    static constexpr int E = 8;
    static MSSEG_DEVFN void load(const bf16_t* p, float* f) {
        const bf16x8_t v = *(const bf16x8_t*)p;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
    }
    static MSSEG_DEVFN void store(bf16_t* p, const float* f) {
        bf16x8_t v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16_t)f[e];
        *(bf16x8_t*)p = v;
    }
};
template <> struct Chunk<float> {
    static constexpr int E = 4;
    static MSSEG_DEVFN void load(const float* p, float* f) {
        const f32x4_t v = *(const f32x4_t*)p;
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = v[e];
    }
    static MSSEG_DEVFN void store(float* p, const float* f) { *(f32x4_t*)p = f32x4_t{f[0], f[1], f[2], f[3]}; }
};

struct DwParams {
    const void* x; long long ldx;
    const void* w;         // [27][C] tap-major, compute dtype (one 16-byte chunk per tap and thread)
    const float* bias;     // [C] or null
    void* y; long long ldy;
    int N, D, H, W, C;
    int flip;              // 1: taps mirrored (input gradient)
    float out_scale;       // w == null: every tap is 1 and the fp32 sum is scaled by this (AvgPool3d(3, 1, 1): 1 / 27)
};

template <typename T>
__global__ __launch_bounds__(256) void dwconv3_kernel(const DwParams p) {
    constexpr int E = Chunk<T>::E;
    const int nch = p.C / E;
    const long long total = (long long)p.N * p.D * p.H * p.W * nch;
    const T* __restrict__ xg = (const T*)p.x;
    T* __restrict__ yg = (T*)p.y;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long v = i / nch;
        const int w0 = (int)(v % p.W); long long t = v / p.W;
        const int h0 = (int)(t % p.H); t /= p.H;
        const int d0 = (int)(t % p.D);
        const int c0 = ch * E;
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = p.bias ? p.bias[c0 + e] : 0.f;
        // branch-free: out-of-volume taps read the centre voxel and are multiplied by zero, so that all 27 + 27 loads of a
        // thread can be in flight together
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const bool okd = (unsigned)(d0 + kd - 1) < (unsigned)p.D;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const bool okh = okd && (unsigned)(h0 + kh - 1) < (unsigned)p.H;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const bool ok = okh && (unsigned)(w0 + kw - 1) < (unsigned)p.W;
                    const int tap = (kd * 3 + kh) * 3 + kw;
                    float wt[E];
                    if (p.w) {
                        Chunk<T>::load((const T*)p.w + (long long)(p.flip ? 26 - tap : tap) * p.C + c0, wt);
                    } else {
#pragma unroll
                        for (int e = 0; e < E; ++e) wt[e] = 1.f;
                    }
                    const long long nv = ok ? v + ((long long)(kd - 1) * p.H + (kh - 1)) * p.W + (kw - 1) : v;
                    float xv[E];
                    Chunk<T>::load(xg + nv * p.ldx + c0, xv);
                    const float m = ok ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] = fmaf(xv[e] * m, wt[e], acc[e]);
                }
            }
        }
        if (!p.w) {
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] *= p.out_scale;
        }
        Chunk<T>::store(yg + v * p.ldy + c0, acc);
    }
}

struct DwWgParams {
    const void* x; long long ldx;
    const void* dy; long long lddy;
    float* ws;             // [rows][28][C]: taps 0..26, row 27 = bias gradient
    int N, D, H, W, C;
    long long vox_per_row; // voxels each workgroup row covers
};

template <typename T>
__global__ __launch_bounds__(256) void dwconv3_wgrad_kernel(const DwWgParams p) {
    constexpr int E = Chunk<T>::E;
    __shared__ float red[3][64][E];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = p.C / E;
    const int pair = blockIdx.y * 64 + lane;           // pair = kd * nch + chunk: consecutive lanes -> consecutive chunks
    const bool live = pair < 3 * nch;
    const int kd = live ? pair / nch : 0, ch = live ? pair - kd * nch : 0;
    const int c0 = ch * E;
    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ gg = (const T*)p.dy;
    float acc[9][E], bacc[E];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < E; ++e) acc[k][e] = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) bacc[e] = 0.f;
    const long long NV = (long long)p.N * p.D * p.H * p.W;
    const long long v_begin = (long long)blockIdx.x * p.vox_per_row;
    long long v_end = v_begin + p.vox_per_row;
    if (v_end > NV) v_end = NV;
    if (live) {
        // two voxels per iteration: twenty independent loads in flight per lane instead of ten (the loop is latency-bound)
        for (long long vv = v_begin + wave; vv < v_end; vv += 8) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long long v = vv + 4 * u;
                if (v >= v_end) continue;
                const int w0 = (int)(v % p.W); long long t = v / p.W;
                const int h0 = (int)(t % p.H); t /= p.H;
                const int d0 = (int)(t % p.D);
                float g[E];
                Chunk<T>::load(gg + v * p.lddy + c0, g);
                if (kd == 1) {
#pragma unroll
                    for (int e = 0; e < E; ++e) bacc[e] += g[e];
                }
                const bool okd = (unsigned)(d0 + kd - 1) < (unsigned)p.D;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const bool okh = okd && (unsigned)(h0 + kh - 1) < (unsigned)p.H;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const bool ok = okh && (unsigned)(w0 + kw - 1) < (unsigned)p.W;
                        const long long nv = ok ? v + ((long long)(kd - 1) * p.H + (kh - 1)) * p.W + (kw - 1) : v;
                        float xv[E];
                        Chunk<T>::load(xg + nv * p.ldx + c0, xv);
                        const float m = ok ? 1.f : 0.f;
#pragma unroll
                        for (int e = 0; e < E; ++e) acc[kh * 3 + kw][e] = fmaf(g[e] * m, xv[e], acc[kh * 3 + kw][e]);
                    }
                }
            }
        }
    }
    // waves 1..3 -> wave 0 in a fixed order, one group of E accumulators at a time
    float* row = p.ws + (long long)blockIdx.x * 28 * p.C;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        float* a = k < 9 ? acc[k] : bacc;
        if (wave > 0) {
#pragma unroll
            for (int e = 0; e < E; ++e) red[wave - 1][lane][e] = a[e];
        }
        __syncthreads();
        if (wave == 0 && live) {
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] = ((a[e] + red[0][lane][e]) + red[1][lane][e]) + red[2][lane][e];
            if (k < 9) {
#pragma unroll
                for (int e = 0; e < E; ++e) row[(long long)(kd * 9 + k) * p.C + c0 + e] = a[e];
            } else if (kd == 1) {
#pragma unroll
                for (int e = 0; e < E; ++e) row[(long long)27 * p.C + c0 + e] = a[e];
            }
        }
        __syncthreads();
    }
}

// dw[c][tap] (torch layout [C, 1, 3, 3, 3]) and db[c] (+)= sum over rows: 64 columns x 4 row slots per block (slot s adds
// rows s, s + 4, ... in order), slots combined in a fixed order -- bit-reproducible
__global__ __launch_bounds__(256) void dwconv3_wgrad_finalize_kernel(const float* ws, int rows, int C, float* dw, float* db,
                                                                     int acc_w, int acc_b) {
    __shared__ float part[4][64];
    const int col = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;               // i = t * C + c
    const int L = 28 * C;
    float s = 0.f;
    if (i < L) {
#pragma unroll 8
        for (int r = slot; r < rows; r += 4) s += ws[(long long)r * L + i];
    }
    part[slot][col] = s;
    __syncthreads();
    if (slot != 0 || i >= L) return;
    s = ((part[0][col] + part[1][col]) + part[2][col]) + part[3][col];
    const int t = i / C, c = i - t * C;
    if (t < 27) {
        if (dw) dw[(long long)c * 27 + t] = acc_w ? dw[(long long)c * 27 + t] + s : s;
    } else if (db) {
        db[c] = acc_b ? db[c] + s : s;
    }
}

int check(const void* x, long long ldx, const void* y, long long ldy, int N, int D, int H, int W, int C, int dtype, const char* what) {
    if (!x || !y) MSSEG_FAIL(MSSEG_EINVAL, "%s: null pointer", what);
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "%s: bad dtype", what);
    const int esz = dtype == MSSEG_F32 ? 4 : 2, epc = 16 / esz;
    if (N < 1 || D < 1 || H < 1 || W < 1 || C < 1 || C % epc) MSSEG_FAIL(MSSEG_EINVAL, "%s: channels must be a multiple of %d", what, epc);
    if (ldx < C || ldy < C || ldx % epc || ldy % epc || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
        MSSEG_FAIL(MSSEG_EINVAL, "%s: tensors must be 16-byte aligned with voxel strides that are multiples of %d", what, epc);
    return MSSEG_OK;
}

int launch_dw(const DwParams& p, int dtype, msseg_stream_t stream) {
    const int N = p.N, D = p.D, H = p.H, W = p.W, C = p.C;
    const long long total = (long long)N * D * H * W * (C / (dtype == MSSEG_F32 ? 4 : 8));
    long long gx = (total + 255) / 256;
    const long long cap = (long long)msseg_num_cus() * 16;
    if (gx > cap) gx = cap;
    if (dtype == MSSEG_F32) hipLaunchKernelGGL(dwconv3_kernel<float>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(dwconv3_kernel<bf16_t>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, p);
    MSSEG_CHECK_LAUNCH("dwconv3d_k3_fwd");
    return MSSEG_OK;
}

}  // namespace

extern "C" {

int msseg_dwconv3d_k3_fwd(const void* x, long long ldx, const void* w_taps, const float* bias, void* y, long long ldy, int N,
                          int D, int H, int W, int C, int flip, int dtype, msseg_stream_t stream) {
    int rc = check(x, ldx, y, ldy, N, D, H, W, C, dtype, "dwconv3d_k3_fwd");
    if (rc) return rc;
    if (!w_taps || ((uintptr_t)w_taps & 15)) MSSEG_FAIL(MSSEG_EINVAL, "dwconv3d_k3_fwd: weight table must be 16-byte aligned");
    return launch_dw(DwParams{x, ldx, w_taps, bias, y, ldy, N, D, H, W, C, flip ? 1 : 0, 1.f}, dtype, stream);
}

int msseg_avgpool3d_k3(const void* x, long long ldx, void* y, long long ldy, int N, int D, int H, int W, int C, int dtype,
                       msseg_stream_t stream) {
    int rc = check(x, ldx, y, ldy, N, D, H, W, C, dtype, "avgpool3d_k3");
    if (rc) return rc;
    return launch_dw(DwParams{x, ldx, nullptr, nullptr, y, ldy, N, D, H, W, C, 0, 1.f / 27.f}, dtype, stream);
}

int msseg_dwconv3d_k3_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw, float* dbias,
                            int accumulate_w, int accumulate_b, int N, int D, int H, int W, int C, void* scratch,
                            size_t scratch_bytes, int dtype, msseg_stream_t stream) {
    int rc = check(x, ldx, dy, lddy, N, D, H, W, C, dtype, "dwconv3d_k3_wgrad");
    if (rc) return rc;
    if (!dw && !dbias) MSSEG_FAIL(MSSEG_EINVAL, "dwconv3d_k3_wgrad: nothing to compute");
    if (!scratch || ((uintptr_t)scratch & 255) || scratch_bytes < msseg_reduce_scratch_bytes())
        MSSEG_FAIL(MSSEG_EWORKSPACE, "dwconv3d_k3_wgrad: needs the reduce scratch of %zu bytes", msseg_reduce_scratch_bytes());
    float* ws = (float*)((unsigned char*)scratch + MSSEG_SCRATCH_COUNTER_BYTES);
    const long long NV = (long long)N * D * H * W;
    const size_t row_bytes = (size_t)28 * C * sizeof(float);
    long long rows = (long long)((scratch_bytes - MSSEG_SCRATCH_COUNTER_BYTES) / row_bytes);
    if (rows < 1) MSSEG_FAIL(MSSEG_EINVAL, "dwconv3d_k3_wgrad: %d channels exceed the reduce scratch", C);
    const int epc = dtype == MSSEG_F32 ? 4 : 8;
    const int gy = ceil_div(3 * (C / epc), 64);
    long long want = (long long)msseg_num_cus() * 8 / gy;      // several workgroups per CU in total: the voxel loop is latency-bound
    if (want < 1) want = 1;
    if (rows > want) rows = want;
    if (rows > (NV + 3) / 4) rows = (NV + 3) / 4;
    DwWgParams p{x, ldx, dy, lddy, ws, N, D, H, W, C, ceil_div_ll(NV, rows)};
    rows = ceil_div_ll(NV, p.vox_per_row);
    if (dtype == MSSEG_F32)
        hipLaunchKernelGGL(dwconv3_wgrad_kernel<float>, dim3((unsigned)rows, gy), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(dwconv3_wgrad_kernel<bf16_t>, dim3((unsigned)rows, gy), dim3(256), 0, (hipStream_t)stream, p);
    MSSEG_CHECK_LAUNCH("dwconv3d_k3_wgrad");
    hipLaunchKernelGGL(dwconv3_wgrad_finalize_kernel, dim3(ceil_div(28 * C, 64)), dim3(256), 0, (hipStream_t)stream, ws,
                       (int)rows, C, dw, dbias, accumulate_w, accumulate_b);
    MSSEG_CHECK_LAUNCH("dwconv3d_k3_wgrad_finalize");
    return MSSEG_OK;
}

}  // extern "C"
