// Layout kernels of the vendored MONAI Swin-UNETR encoder (/root/reference/models/segmentors/swin_unetr_official.py): what that
// file does with F.pad, slicing and torch.cat on the token volume, each as ONE pass over 16-byte channel chunks.
//
//   box_copy      dst[n, d, h, w, :] = src[n, d, h, w, :] inside the common box, zero elsewhere in dst:
//                 * F.pad(x, (0, 0, 0, pw, 0, ph, 0, pd)) of the window partition (`:244-250`, 48^3 -> 49^3 for window 7) and of
//                   the odd-size patch merging (`:699-703`),
//                 * the crop x[:, :d, :h, :w] after the window reverse (`:268-270`),
//                 and each is the other's adjoint.
//   merge_gather  PatchMerging (`:699-712`): out[n, i, j, k, s * C + c] = x[n, 2i + a_s, 2j + b_s, 2k + c_s, c] for the eight
//                 sub-grids in the reference's order INCLUDING its duplicates (x5 == x2, x6 == x3), zero beyond an odd grid; the
//                 adjoint sums, per fine voxel, the slots whose offset equals the voxel's parity in slot order (deterministic).
// torch ran these as a fill + strided copy per pad / crop and eight strided copies + eight index-adds per merging (7.8 % of the
// GPU time of the official-variant step).  HBM-bound; one thread per (destination voxel, chunk).
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void box_copy_kernel(const T* __restrict__ src, long long lds, int SD, int SH, int SW,
                                                       T* __restrict__ dst, long long ldd, int DD, int DH, int DW, int N, int C) {
    constexpr int EPC = DT<T>::EPC;
    const unsigned cpv = (unsigned)(C / EPC);
    const unsigned total = (unsigned)N * DD * DH * DW * cpv;   // < 2^31: checked by the host wrapper
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        unsigned t = i / cpv;
        const unsigned g = i - t * cpv;
        const unsigned w = t % (unsigned)DW; t /= (unsigned)DW;
        const unsigned h = t % (unsigned)DH; t /= (unsigned)DH;
        const unsigned d = t % (unsigned)DD, n = t / (unsigned)DD;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (d < (unsigned)SD && h < (unsigned)SH && w < (unsigned)SW)
            v = *(const u32x4_t*)(src + ((((long long)n * SD + d) * SH + h) * SW + w) * lds + g * EPC);
        *(u32x4_t*)(dst + ((((long long)n * DD + d) * DH + h) * DW + w) * ldd + g * EPC) = v;
    }
}

// subs: 3 bits per slot (bit 0: offset along D, bit 1: along H, bit 2: along W), slot s at bits 3s..3s+2
template <typename T>
__global__ __launch_bounds__(256) void merge_gather_fwd_kernel(const T* __restrict__ x, long long ldx, int D, int H, int W,
                                                               T* __restrict__ out, long long ldo, int N, int C, unsigned subs) {
    constexpr int EPC = DT<T>::EPC;
    const int OD = (D + 1) / 2, OH = (H + 1) / 2, OW = (W + 1) / 2;
    const unsigned cpv = (unsigned)(C / EPC);
    const unsigned total = (unsigned)N * OD * OH * OW * 8u * cpv;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        unsigned t = i / cpv;
        const unsigned g = i - t * cpv;
        const unsigned s = t & 7u; t >>= 3;
        const unsigned k = t % (unsigned)OW; t /= (unsigned)OW;
        const unsigned j = t % (unsigned)OH; t /= (unsigned)OH;
        const unsigned ii = t % (unsigned)OD, n = t / (unsigned)OD;
        const unsigned o = (subs >> (3 * s)) & 7u;
        const unsigned d = 2 * ii + (o & 1u), h = 2 * j + ((o >> 1) & 1u), w = 2 * k + ((o >> 2) & 1u);
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (d < (unsigned)D && h < (unsigned)H && w < (unsigned)W)
            v = *(const u32x4_t*)(x + ((((long long)n * D + d) * H + h) * W + w) * ldx + g * EPC);
        *(u32x4_t*)(out + ((((long long)n * OD + ii) * OH + j) * OW + k) * ldo + (long long)s * C + g * EPC) = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void merge_gather_bwd_kernel(const T* __restrict__ dy, long long lddy, T* __restrict__ dx,
                                                               long long lddx, int N, int D, int H, int W, int C, unsigned subs) {
    constexpr int EPC = DT<T>::EPC;
    const int OD = (D + 1) / 2, OH = (H + 1) / 2, OW = (W + 1) / 2;
    const unsigned cpv = (unsigned)(C / EPC);
    const unsigned total = (unsigned)N * D * H * W * cpv;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        unsigned t = i / cpv;
        const unsigned g = i - t * cpv;
        const unsigned w = t % (unsigned)W; t /= (unsigned)W;
        const unsigned h = t % (unsigned)H; t /= (unsigned)H;
        const unsigned d = t % (unsigned)D, n = t / (unsigned)D;
        const unsigned par = (d & 1u) | ((h & 1u) << 1) | ((w & 1u) << 2);
        const T* row = dy + ((((long long)n * OD + (d >> 1)) * OH + (h >> 1)) * OW + (w >> 1)) * lddy + g * EPC;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        int hits = 0;
        u32x4_t only = {0u, 0u, 0u, 0u};
#pragma unroll
        for (unsigned s = 0; s < 8; ++s) {
            if (((subs >> (3 * s)) & 7u) == par) {
                const u32x4_t v = *(const u32x4_t*)(row + (long long)s * C);
                only = v;
                ++hits;
                if constexpr (sizeof(T) == 4) {
                    const f32x4_t f = __builtin_bit_cast(f32x4_t, v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += f[e];
                } else {
                    const bf16x8_t f = __builtin_bit_cast(bf16x8_t, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += (float)f[e];
                }
            }
        }
        u32x4_t o;
        if (hits == 1) {
            o = only;                                   // a single slot: the gradient passes through unrounded
        } else if constexpr (sizeof(T) == 4) {
            o = __builtin_bit_cast(u32x4_t, f32x4_t{acc[0], acc[1], acc[2], acc[3]});
        } else {
            bf16x8_t f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (bf16_t)acc[e];
            o = __builtin_bit_cast(u32x4_t, f);
        }
        *(u32x4_t*)(dx + ((((long long)n * D + d) * H + h) * W + w) * lddx + g * EPC) = o;
    }
}

int grid_for(long long total) {
    long long b = (total + 255) / 256;
    const long long cap = (long long)msseg_num_cus() * 16;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

int check(const void* a, long long lda, const void* b, long long ldb, int C, int dtype, long long total, const char* who) {
    if (!a || !b) MSSEG_FAIL(MSSEG_EINVAL, "%s: null pointer", who);
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "%s: bad dtype", who);
    const int epc = dtype == MSSEG_F32 ? 4 : 8;
    if (C < epc || C % epc || lda % epc || ldb % epc || (((uintptr_t)a | (uintptr_t)b) & 15))
        MSSEG_FAIL(MSSEG_EINVAL, "%s: channels and strides must be multiples of %d, pointers 16-byte aligned", who, epc);
    if (total < 1 || total > 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "%s: bad element count", who);
    return MSSEG_OK;
}

}  // namespace

extern "C" {

int msseg_box_copy(const void* src, long long lds, int SD, int SH, int SW, void* dst, long long ldd, int DD, int DH, int DW, int N,
                   int C, int dtype, msseg_stream_t stream) {
    const long long total = (long long)N * DD * DH * DW * (C / (dtype == MSSEG_F32 ? 4 : 8));
    if (int rc = check(src, lds, dst, ldd, C, dtype, total, "box_copy")) return rc;
    if (SD < 1 || SH < 1 || SW < 1 || lds < C || ldd < C || (long long)N * SD * SH * SW > 0x7fffffffLL)
        MSSEG_FAIL(MSSEG_EINVAL, "box_copy: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MSSEG_F32)
        hipLaunchKernelGGL(box_copy_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)src, lds, SD, SH, SW, (float*)dst, ldd, DD, DH, DW, N, C);
    else
        hipLaunchKernelGGL(box_copy_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)src, lds, SD, SH, SW, (bf16_t*)dst, ldd, DD, DH, DW, N, C);
    MSSEG_CHECK_LAUNCH("box_copy");
    return MSSEG_OK;
}

int msseg_merge_gather_fwd(const void* x, long long ldx, int D, int H, int W, void* out, long long ldo, int N, int C, unsigned subs,
                           int dtype, msseg_stream_t stream) {
    const long long OV = (long long)N * ((D + 1) / 2) * ((H + 1) / 2) * ((W + 1) / 2);
    const long long total = OV * 8 * (C / (dtype == MSSEG_F32 ? 4 : 8));
    if (int rc = check(x, ldx, out, ldo, C, dtype, total, "merge_gather_fwd")) return rc;
    if (D < 1 || H < 1 || W < 1 || ldx < C || ldo < 8LL * C) MSSEG_FAIL(MSSEG_EINVAL, "merge_gather_fwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MSSEG_F32)
        hipLaunchKernelGGL(merge_gather_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, ldx, D, H, W, (float*)out, ldo, N, C, subs);
    else
        hipLaunchKernelGGL(merge_gather_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)x, ldx, D, H, W, (bf16_t*)out, ldo, N, C, subs);
    MSSEG_CHECK_LAUNCH("merge_gather_fwd");
    return MSSEG_OK;
}

int msseg_merge_gather_bwd(const void* dy, long long lddy, void* dx, long long lddx, int N, int D, int H, int W, int C, unsigned subs,
                           int dtype, msseg_stream_t stream) {
    const long long total = (long long)N * D * H * W * (C / (dtype == MSSEG_F32 ? 4 : 8));
    if (int rc = check(dy, lddy, dx, lddx, C, dtype, total, "merge_gather_bwd")) return rc;
    if (D < 1 || H < 1 || W < 1 || lddx < C || lddy < 8LL * C) MSSEG_FAIL(MSSEG_EINVAL, "merge_gather_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MSSEG_F32)
        hipLaunchKernelGGL(merge_gather_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)dy, lddy, (float*)dx, lddx, N, D, H, W, C, subs);
    else
        hipLaunchKernelGGL(merge_gather_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, N, D, H, W, C, subs);
    MSSEG_CHECK_LAUNCH("merge_gather_bwd");
    return MSSEG_OK;
}

}  // extern "C"
