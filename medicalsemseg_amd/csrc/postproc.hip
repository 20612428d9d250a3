// Post-inference byte work on the device: class map from the blended logits, nearest-neighbour
// resampling of the label map to the original grid, majority vote over fold predictions.
//
//   argmax_u8       /root/reference/engine/test.py:140-141   softmax(outputs, 1) -> np.argmax(axis=1).astype(uint8)
//   resample_nearest /root/reference/utils/misc.py:420-425   scipy.ndimage.zoom(img, target/shape, order=0, prefilter=False)
//   majority_vote   /root/reference/majority_vote.py:23-37   votes of the foreground classes, background starts with one
//                                                            vote, np.argmax (first maximum wins)
// All three are HBM-bound gathers over uint8 / fp32 volumes; one thread per output voxel, 16-byte stores where the
// output is contiguous.
#include "common.h"

namespace {

// first maximum over the class axis of NCDHW fp32 logits (softmax is monotonic: the arg max of the probabilities is
// the arg max of the logits; the reference's fp32 softmax can only differ where two logits round to equal probabilities)
__global__ __launch_bounds__(256) void argmax_u8_kernel(const float* __restrict__ logits, int C, long long V,
                                                        unsigned char* __restrict__ out) {
    for (long long v4 = (blockIdx.x * 256LL + threadIdx.x) * 4; v4 < V; v4 += (long long)gridDim.x * 1024) {
        unsigned char o[4] = {0, 0, 0, 0};
        if (v4 + 4 <= V && (((uintptr_t)(logits + v4)) & 15) == 0 && (V & 3) == 0) {
            f32x4_t best = *(const f32x4_t*)(logits + v4);
            for (int c = 1; c < C; ++c) {
                const f32x4_t x = *(const f32x4_t*)(logits + (long long)c * V + v4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (x[e] > best[e]) { best[e] = x[e]; o[e] = (unsigned char)c; }
            }
            if ((((uintptr_t)(out + v4)) & 3) == 0) {
                *(uint32_t*)(out + v4) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) out[v4 + e] = o[e];
            }
        } else {
            for (int e = 0; e < 4 && v4 + e < V; ++e) {
                float best = logits[v4 + e];
                unsigned char a = 0;
                for (int c = 1; c < C; ++c) {
                    const float x = logits[(long long)c * V + v4 + e];
                    if (x > best) { best = x; a = (unsigned char)c; }
                }
                out[v4 + e] = a;
            }
        }
    }
}

// scipy's order-0 zoom: input coordinate = o * (in - 1) / (out - 1) in double (the ratio is formed first, as scipy
// does), index = floor(coordinate + 0.5)
__global__ __launch_bounds__(256) void resample_nearest_u8_kernel(const unsigned char* __restrict__ src, int SD, int SH,
                                                                  int SW, unsigned char* __restrict__ dst, int TD, int TH,
                                                                  int TW, double rd, double rh, double rw) {
    const long long total = (long long)TD * TH * TW;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int w = (int)(i % TW), h = (int)((i / TW) % TH), d = (int)(i / ((long long)TW * TH));
        int sd = (int)floor((double)d * rd + 0.5), sh = (int)floor((double)h * rh + 0.5), sw = (int)floor((double)w * rw + 0.5);
        sd = sd < 0 ? 0 : (sd >= SD ? SD - 1 : sd);
        sh = sh < 0 ? 0 : (sh >= SH ? SH - 1 : sh);
        sw = sw < 0 ? 0 : (sw >= SW ? SW - 1 : sw);
        dst[i] = src[((long long)sd * SH + sh) * SW + sw];
    }
}

// labels [F][V] uint8 -> out [V]: votes[c] = #folds predicting c (c >= 1), votes[0] = 1, first maximum
template <int CMAX>
__global__ __launch_bounds__(256) void majority_vote_u8_kernel(const unsigned char* __restrict__ labels, int F, long long V,
                                                               int C, unsigned char* __restrict__ out) {
    for (long long v = blockIdx.x * 256LL + threadIdx.x; v < V; v += (long long)gridDim.x * 256) {
        unsigned char votes[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) votes[c] = c == 0 ? 1 : 0;
        for (int f = 0; f < F; ++f) {
            const int l = labels[(long long)f * V + v];
#pragma unroll
            for (int c = 1; c < CMAX; ++c) votes[c] += (l == c && c < C) ? 1 : 0;
        }
        int best = 0;
#pragma unroll
        for (int c = 1; c < CMAX; ++c)
            if (c < C && votes[c] > votes[best]) best = c;
        out[v] = (unsigned char)best;
    }
}

inline int pp_grid(long long total, int per_thread) {
    long long b = ceil_div_ll(total, 256LL * per_thread);
    const long long cap = (long long)msseg_num_cus() * 16;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

int msseg_argmax_u8(const float* logits, int C, long long V, uint8_t* out, msseg_stream_t stream) {
    if (!logits || !out || C < 1 || C > 255 || V < 1) MSSEG_FAIL(MSSEG_EINVAL, "argmax_u8: bad args");
    hipLaunchKernelGGL(argmax_u8_kernel, dim3(pp_grid(V, 4)), dim3(256), 0, (hipStream_t)stream, logits, C, V, out);
    MSSEG_CHECK_LAUNCH("argmax_u8");
    return MSSEG_OK;
}

int msseg_resample_nearest_u8(const uint8_t* src, int SD, int SH, int SW, uint8_t* dst, int TD, int TH, int TW,
                              msseg_stream_t stream) {
    if (!src || !dst || SD < 1 || SH < 1 || SW < 1 || TD < 1 || TH < 1 || TW < 1)
        MSSEG_FAIL(MSSEG_EINVAL, "resample_nearest_u8: bad args");
    const double rd = TD > 1 ? (double)(SD - 1) / (double)(TD - 1) : 0.0;
    const double rh = TH > 1 ? (double)(SH - 1) / (double)(TH - 1) : 0.0;
    const double rw = TW > 1 ? (double)(SW - 1) / (double)(TW - 1) : 0.0;
    hipLaunchKernelGGL(resample_nearest_u8_kernel, dim3(pp_grid((long long)TD * TH * TW, 1)), dim3(256), 0,
                       (hipStream_t)stream, src, SD, SH, SW, dst, TD, TH, TW, rd, rh, rw);
    MSSEG_CHECK_LAUNCH("resample_nearest_u8");
    return MSSEG_OK;
}

int msseg_majority_vote_u8(const uint8_t* labels, int F, long long V, int C, uint8_t* out, msseg_stream_t stream) {
    if (!labels || !out || F < 1 || F > 254 || V < 1 || C < 1 || C > 16)
        MSSEG_FAIL(MSSEG_EINVAL, "majority_vote_u8: 1 <= folds <= 254, 1 <= classes <= 16");
    const int g = pp_grid(V, 1);
    if (C <= 4) hipLaunchKernelGGL(majority_vote_u8_kernel<4>, dim3(g), dim3(256), 0, (hipStream_t)stream, labels, F, V, C, out);
    else if (C <= 8) hipLaunchKernelGGL(majority_vote_u8_kernel<8>, dim3(g), dim3(256), 0, (hipStream_t)stream, labels, F, V, C, out);
    else hipLaunchKernelGGL(majority_vote_u8_kernel<16>, dim3(g), dim3(256), 0, (hipStream_t)stream, labels, F, V, C, out);
    MSSEG_CHECK_LAUNCH("majority_vote_u8");
    return MSSEG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Device-side training crop + augmentation on a cached volume: replaces the CPU chain
// RandCropByPosNegLabeld -> RandFlipd x3 -> RandRotate90d -> RandShiftIntensityd -> RandScaleIntensityd
// (/root/reference/data/dataset_builder.py:108-193, data/transforms.py:264-419) for volumes resident in HBM.
// The random draws are made on the host (a few numbers per patch) and arrive as a parameter table; the kernel is a
// pure gather: out[b][c][o] = (img[c][src(o)] + shift) * scale for the image channels, lab[src(o)] for the label,
// with src(o) = crop start + the inverse of (flip d, flip h, flip w, rot90^k in the (d, h) plane) applied to o.
// ---------------------------------------------------------------------------------------------------------
namespace {

struct AugRow { int z0, y0, x0, flips, rotk, pad0; float shift, scale; };   // 32 bytes, mirrors msseg_aug_row

template <typename TO>
__global__ __launch_bounds__(256) void aug_crop_kernel(const float* __restrict__ img, const unsigned char* __restrict__ lab,
                                                       int C, int VD, int VH, int VW, const AugRow* __restrict__ table,
                                                       TO* __restrict__ out_img, float* __restrict__ out_lab, int R) {
#pragma clang fp contract(off)   // (v + shift) * scale with two roundings, as the two numpy transforms apply them
    const AugRow row = table[blockIdx.y];
    const long long R3 = (long long)R * R * R, V = (long long)VD * VH * VW;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < R3; i += (long long)gridDim.x * 256) {
        int x = (int)(i % R), y = (int)((i / R) % R), z = (int)(i / ((long long)R * R));
        // undo rot90^k in the (z, y) plane: out[i, j] = m[j, n-1-i] (k = 1), m[n-1-i, n-1-j] (2), m[n-1-j, i] (3)
        int sz = z, sy = y;
        if (row.rotk == 1) { sz = y; sy = R - 1 - z; }
        else if (row.rotk == 2) { sz = R - 1 - z; sy = R - 1 - y; }
        else if (row.rotk == 3) { sz = R - 1 - y; sy = z; }
        int sx = x;
        if (row.flips & 1) sz = R - 1 - sz;
        if (row.flips & 2) sy = R - 1 - sy;
        if (row.flips & 4) sx = R - 1 - sx;
        const long long v = ((long long)(row.z0 + sz) * VH + (row.y0 + sy)) * VW + (row.x0 + sx);
        for (int c = 0; c < C; ++c) {
            const float a = img[(long long)c * V + v] + row.shift;
            const float b = a * row.scale;
            if constexpr (sizeof(TO) == 4) out_img[((long long)blockIdx.y * C + c) * R3 + i] = b;
            else out_img[((long long)blockIdx.y * C + c) * R3 + i] = (TO)b;
        }
        if (lab) out_lab[(long long)blockIdx.y * R3 + i] = (float)lab[v];
    }
}

}  // namespace

extern "C" int msseg_aug_crop_batch(const float* img, const uint8_t* lab, int C, int VD, int VH, int VW,
                                    const void* table, int npatch, void* out_img, int out_dtype, float* out_lab, int R,
                                    msseg_stream_t stream) {
    if (!img || !table || !out_img || C < 1 || npatch < 1 || npatch > 65535 || R < 1 || R > VD || R > VH || R > VW)
        MSSEG_FAIL(MSSEG_EINVAL, "aug_crop_batch: bad args (cubic roi %d inside volume %dx%dx%d)", R, VD, VH, VW);
    if ((lab == nullptr) != (out_lab == nullptr)) MSSEG_FAIL(MSSEG_EINVAL, "aug_crop_batch: lab and out_lab go together");
    long long gx = ceil_div_ll((long long)R * R * R, 256LL * 4);
    if (gx > 4096) gx = 4096;
    dim3 grid((unsigned)gx, npatch);
    if (out_dtype == MSSEG_F32)
        hipLaunchKernelGGL(aug_crop_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, img, lab, C, VD, VH, VW,
                           (const AugRow*)table, (float*)out_img, out_lab, R);
    else if (out_dtype == MSSEG_BF16)
        hipLaunchKernelGGL(aug_crop_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, img, lab, C, VD, VH, VW,
                           (const AugRow*)table, (bf16_t*)out_img, out_lab, R);
    else MSSEG_FAIL(MSSEG_EINVAL, "aug_crop_batch: bad dtype");
    MSSEG_CHECK_LAUNCH("aug_crop_batch");
    return MSSEG_OK;
}
