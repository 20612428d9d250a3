// Dice + cross-entropy loss (single pass over logits/labels forward, single pass backward), hard Dice
// metric counts, and the sliding-window gather / blend / normalise kernels.
#include "common.h"

namespace {

enum { LAB_F32 = 0, LAB_BF16 = 1, LAB_U8 = 2, LAB_I64 = 3 };

MSSEG_DEVFN int load_label(const void* labels, int label_dtype, long long idx) {
    switch (label_dtype) {
        case LAB_F32: return (int)((const float*)labels)[idx];
        case LAB_BF16: return (int)(float)((const bf16_t*)labels)[idx];
        case LAB_U8: return (int)((const uint8_t*)labels)[idx];
        default: return (int)((const long long*)labels)[idx];
    }
}

// a 16-byte logits row (channels-last, ld * sizeof(T) == 16, what the networks' logits buffers are): ONE vector load per
// voxel instead of one 2-byte load per class
template <typename T, int CMAX>
MSSEG_DEVFN void unpack_row(const u32x4_t raw, int C, float* x) {
    constexpr int EPC = DT<T>::EPC;
    float v[EPC];
    if constexpr (sizeof(T) == 2) {
        const bf16x8_t h = __builtin_bit_cast(bf16x8_t, raw);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = (float)h[e];
    } else {
        const f32x4_t f = __builtin_bit_cast(f32x4_t, raw);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = f[e];
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) x[c] = (c < C && c < EPC) ? v[c < EPC ? c : 0] : -INFINITY;
}

template <typename T, int CMAX>
MSSEG_DEVFN void load_logits(const T* base, long long ld, long long S, int C, long long n, long long s, float* x) {
    // ld > 0: channels-last [N][S][ld]; ld == 0: NCDHW [N][C][S]
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        if (c < C) x[c] = ld > 0 ? DT<T>::ld(base + (n * S + s) * ld + c) : DT<T>::ld(base + (n * C + c) * S + s);
        else x[c] = -INFINITY;
    }
}

template <int CMAX> MSSEG_DEVFN void softmax_inplace(float* x, int C, float& lse) {
    float mx = x[0];
#pragma unroll
    for (int c = 1; c < CMAX; ++c) mx = fmaxf(mx, x[c]);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        x[c] = (c < C) ? expf(x[c] - mx) : 0.f;
        sum += x[c];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) x[c] *= inv;
    lse = mx + logf(sum);
}

// partial[n][c][0..3] += (sum p*t, sum p^2, sum t, -sum t*log p); hard[n][c][0..2] += (|P&T|, |P|, |T|)
template <typename T, int CMAX>
__global__ __launch_bounds__(256) void dice_ce_partials_kernel(const T* __restrict__ logits, long long ld,
                                                               const void* __restrict__ labels, int label_dtype,
                                                               float* partial, float* hard, long long S, int C,
                                                               long long vox_per_block, float* rows) {
    __shared__ float red[4][CMAX * 7];
    const long long n = blockIdx.y;
    const long long s0 = (long long)blockIdx.x * vox_per_block;
    long long s1 = s0 + vox_per_block;
    if (s1 > S) s1 = S;
    float acc[CMAX][7];
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[c][k] = 0.f;
    // four voxels per trip, their rows and labels requested before the first is used (a loop of one voxel per trip kept a
    // single 2-byte load in flight per class: 1.3 TB/s); the per-thread order of the voxels, and with it every sum, is unchanged
    const bool vec = ld > 0 && ld * (long long)sizeof(T) == 16 && C <= DT<T>::EPC && (((uintptr_t)logits) & 15) == 0;
    auto one = [&](float* x, int lab) {
        int am = 0;
        float best = x[0];
#pragma unroll
        for (int c = 1; c < CMAX; ++c)
            if (c < C && x[c] > best) { best = x[c]; am = c; }
        float raw[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) raw[c] = x[c];
        float lse;
        softmax_inplace<CMAX>(x, C, lse);
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C) {
                const float t = (lab == c) ? 1.f : 0.f;
                acc[c][0] += x[c] * t;
                acc[c][1] += x[c] * x[c];
                acc[c][2] += t;
                acc[c][3] += t * (lse - raw[c]);
                const float pm = (am == c) ? 1.f : 0.f;
                acc[c][4] += pm * t;
                acc[c][5] += pm;
                acc[c][6] += t;
            }
        }
    };
    for (long long s = s0 + threadIdx.x; s < s1; s += 1024) {
        if (vec) {
            u32x4_t rows4[4];
            int lab4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long sj = s + 256 * j;
                rows4[j] = u32x4_t{0u, 0u, 0u, 0u};
                lab4[j] = 0;
                if (sj < s1) {
                    rows4[j] = *(const u32x4_t*)(logits + (n * S + sj) * ld);
                    lab4[j] = load_label(labels, label_dtype, n * S + sj);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (s + 256 * j < s1) {
                    float x[CMAX];
                    unpack_row<T, CMAX>(rows4[j], C, x);
                    one(x, lab4[j]);
                }
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                const long long sj = s + 256 * j;
                if (sj < s1) {
                    float x[CMAX];
                    load_logits<T, CMAX>(logits, ld, S, C, n, sj, x);
                    one(x, load_label(labels, label_dtype, n * S + sj));
                }
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float v = wave_sum(acc[c][k]);
            if (lane == 0) red[wave][c * 7 + k] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 7; i += 256) {
        const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        const int c = i / 7, k = i % 7;
        if (rows) {   // deterministic mode: one row per block, summed in a fixed order by dice_ce_rows_finalize_kernel
            rows[((long long)n * gridDim.x + blockIdx.x) * (C * 7) + i] = v;
            continue;
        }
        if (k < 4) {
            if (partial) atomicAdd(&partial[(n * C + c) * 4 + k], v);
        } else if (hard) {
            atomicAdd(&hard[(n * C + c) * 3 + (k - 4)], v);
        }
    }
}

__global__ void dice_ce_finalize_kernel(const float* partial, float* loss, int N, long long S, int C, float snr,
                                        float sdr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float dice = 0.f, ce = 0.f;
    for (int i = 0; i < N * C; ++i) {
        const float I = partial[i * 4 + 0], p2 = partial[i * 4 + 1], t = partial[i * 4 + 2];
        dice += 1.f - (2.f * I + snr) / (p2 + t + sdr);
        ce += partial[i * 4 + 3];
    }
    dice /= (float)(N * C);
    ce /= (float)((double)N * (double)S);
    loss[0] = dice + ce;
    loss[1] = dice;
    loss[2] = ce;
}

// rows [N][nblk][C*7] -> partial[N][C][4], hard[N][C][3] (optional) and the loss triple, one block, fixed order
__global__ __launch_bounds__(256) void dice_ce_rows_finalize_kernel(const float* rows, int nblk, float* partial, float* hard,
                                                                    float* loss, int N, long long S, int C, float snr,
                                                                    float sdr) {
    __shared__ float tot[8 * 16 * 7];
    const int L = C * 7;
    // thread = (column o, row slot): 256 / L' slots per column add every slot-th row, then the slots are added in order
    for (int n = 0; n < N; ++n) {
        const float* base = rows + (long long)n * nblk * L;
        const int cols = L < 256 ? L : 256, G = 256 / cols;
        __shared__ float fin[256];
        for (int c0 = 0; c0 < L; c0 += cols) {
            const int col = threadIdx.x % cols, rg = threadIdx.x / cols, o = c0 + col;
            float s = 0.f;
            if (o < L && rg < G) {
                // eight rows in flight per thread (the adds stay in row order: same sum); a loop of one load per trip paid the
                // memory latency 36 times over: 20 us for 432 rows per sample
                int b = rg;
                for (; b + 7 * G < nblk; b += 8 * G) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = base[(long long)(b + j * G) * L + o];
#pragma unroll
                    for (int j = 0; j < 8; ++j) s += v[j];
                }
                for (; b < nblk; b += G) s += base[(long long)b * L + o];
            }
            __syncthreads();
            fin[threadIdx.x] = s;
            __syncthreads();
            if (o < L && rg == 0) {
                float t = 0.f;
                for (int g = 0; g < G; ++g) t += fin[g * cols + col];
                tot[n * L + o] = t;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N * L; i += 256) {
        const int n = i / L, o = i % L, c = o / 7, k = o % 7;
        if (k < 4) partial[(n * C + c) * 4 + k] = tot[i];
        else if (hard) hard[(n * C + c) * 3 + (k - 4)] = tot[i];
    }
    if (threadIdx.x == 0 && loss) {
        float dice = 0.f, ce = 0.f;
        for (int i = 0; i < N * C; ++i) {
            const float I = tot[i * 7 + 0], p2 = tot[i * 7 + 1], t = tot[i * 7 + 2];
            dice += 1.f - (2.f * I + snr) / (p2 + t + sdr);
            ce += tot[i * 7 + 3];
        }
        dice /= (float)(N * C);
        ce /= (float)((double)N * (double)S);
        loss[0] = dice + ce;
        loss[1] = dice;
        loss[2] = ce;
    }
}

template <typename T, int CMAX>
__global__ __launch_bounds__(256) void dice_ce_bwd_kernel(const T* __restrict__ logits, long long ld,
                                                          const void* __restrict__ labels, int label_dtype,
                                                          const float* __restrict__ partial, const float* gscale,
                                                          T* __restrict__ dlogits, long long ldd, int N, long long S,
                                                          int C, float snr, float sdr) {
    const long long n = blockIdx.y;
    // per (n, c) constants of d dice / d p
    float ka[CMAX], kb[CMAX];
    const float gs = gscale ? gscale[0] : 1.f;
    const float wd = gs / (float)(N * C);
    const float wc = gs / (float)((double)N * (double)S);
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        ka[c] = kb[c] = 0.f;
        if (c < C) {
            const float I = partial[(n * C + c) * 4 + 0], p2 = partial[(n * C + c) * 4 + 1], t = partial[(n * C + c) * 4 + 2];
            const float den = p2 + t + sdr, num = 2.f * I + snr;
            ka[c] = -2.f / den * wd;              // multiplies t
            kb[c] = 2.f * num / (den * den) * wd;  // multiplies p
        }
    }
    const bool vec = ld > 0 && ld * (long long)sizeof(T) == 16 && ldd == ld && C <= DT<T>::EPC &&
                     ((((uintptr_t)logits) | ((uintptr_t)dlogits)) & 15) == 0;
    auto grad = [&](float* x, int lab, float* d) {   // x: logits in, probabilities out; d[c] = d loss / d logit c
        float lse;
        softmax_inplace<CMAX>(x, C, lse);
        float g[CMAX], dot = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            const float t = (lab == c) ? 1.f : 0.f;
            g[c] = (c < C) ? (ka[c] * t + kb[c] * x[c]) : 0.f;
            dot += g[c] * x[c];
        }
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            const float t = (lab == c) ? 1.f : 0.f;
            d[c] = (c < C) ? x[c] * (g[c] - dot) + wc * (x[c] - t) : 0.f;
        }
    };
    const long long stride = (long long)gridDim.x * 256;
    if (vec) {
        // 16-byte rows in and out: one vector load and ONE vector store per voxel (the padding channels are written as zeros
        // in the same store), four voxels per trip with all loads first
        constexpr int EPC = DT<T>::EPC;
        for (long long s = blockIdx.x * 256LL + threadIdx.x; s < S; s += 4 * stride) {
            u32x4_t rows4[4];
            int lab4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long sj = s + j * stride;
                rows4[j] = u32x4_t{0u, 0u, 0u, 0u};
                lab4[j] = 0;
                if (sj < S) {
                    rows4[j] = *(const u32x4_t*)(logits + (n * S + sj) * ld);
                    lab4[j] = load_label(labels, label_dtype, n * S + sj);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long sj = s + j * stride;
                if (sj < S) {
                    float x[CMAX], d[CMAX];
                    unpack_row<T, CMAX>(rows4[j], C, x);
                    grad(x, lab4[j], d);
                    u32x4_t o;
                    if constexpr (sizeof(T) == 2) {
                        bf16x8_t h;
#pragma unroll
                        for (int e = 0; e < EPC; ++e) h[e] = (bf16_t)((e < CMAX && e < C) ? d[e < CMAX ? e : 0] : 0.f);
                        o = __builtin_bit_cast(u32x4_t, h);
                    } else {
                        f32x4_t f;
#pragma unroll
                        for (int e = 0; e < EPC; ++e) f[e] = (e < CMAX && e < C) ? d[e < CMAX ? e : 0] : 0.f;
                        o = __builtin_bit_cast(u32x4_t, f);
                    }
                    *(u32x4_t*)(dlogits + (n * S + sj) * ldd) = o;
                }
            }
        }
        return;
    }
    for (long long s = blockIdx.x * 256LL + threadIdx.x; s < S; s += stride) {
        float x[CMAX], d[CMAX];
        load_logits<T, CMAX>(logits, ld, S, C, n, s, x);
        const int lab = load_label(labels, label_dtype, n * S + s);
        grad(x, lab, d);
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C) {
                if (ldd > 0) DT<T>::st(dlogits + (n * S + s) * ldd + c, d[c]);
                else DT<T>::st(dlogits + (n * C + c) * S + s, d[c]);
            }
        }
        if (ldd > C) {
            for (int c = C; c < ldd; ++c) DT<T>::st(dlogits + (n * S + s) * ldd + c, 0.f);
        }
    }
}

// ---------------- sliding window ----------------
template <typename T>
__global__ void sw_blend_kernel(const T* __restrict__ win, long long ld, const float* __restrict__ imp,
                                float* __restrict__ out, float* __restrict__ cnt, int C, int VD, int VH, int VW,
                                int RD, int RH, int RW, int z0, int y0, int x0) {
#pragma clang fp contract(off)  // mul then add with two roundings, bit-identical to torch's `out[idx] += imp * seg`
    const long long R = (long long)RD * RH * RW, V = (long long)VD * VH * VW;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < R; i += (long long)gridDim.x * 256) {
        const int rx = (int)(i % RW), ry = (int)((i / RW) % RH), rz = (int)(i / ((long long)RW * RH));
        const long long v = ((long long)(z0 + rz) * VH + (y0 + ry)) * VW + (x0 + rx);
        const float w = imp[i];
        for (int c = 0; c < C; ++c) {
            const float val = ld > 0 ? DT<T>::ld(win + i * ld + c) : DT<T>::ld(win + c * R + i);
            const float prod = w * val;
            out[c * V + v] = out[c * V + v] + prod;
        }
        cnt[v] += w;
    }
}

__global__ void sw_normalize_kernel(float* out, const float* cnt, int C, long long V) {
    const long long total = (long long)C * V;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        out[i] = out[i] / cnt[i % V];
}

template <typename T>
__global__ void sw_gather_kernel(const float* __restrict__ vol, T* __restrict__ win, int C, int VD, int VH, int VW,
                                 int RD, int RH, int RW, int z0, int y0, int x0, float cval) {
    const long long R = (long long)RD * RH * RW, V = (long long)VD * VH * VW;
    const long long total = R * C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long ri = i % R;
        const int c = (int)(i / R);
        const int rx = (int)(ri % RW), ry = (int)((ri / RW) % RH), rz = (int)(ri / ((long long)RW * RH));
        const int z = z0 + rz, y = y0 + ry, x = x0 + rx;
        float v = cval;
        if ((unsigned)z < (unsigned)VD && (unsigned)y < (unsigned)VH && (unsigned)x < (unsigned)VW)
            v = vol[c * V + ((long long)z * VH + y) * VW + x];
        DT<T>::st(win + i, v);
    }
}


// ---- batched forms: one launch per window batch, window starts in a device-resident table ----------------
// table[j] = (b, z0, y0, x0); b < 0 marks an unused slot (short last batch).
template <typename T>
__global__ void sw_gather_batch_kernel(const float* __restrict__ vol, long long vol_bstride, T* __restrict__ win,
                                       long long ldw, const int* __restrict__ table, int nwin, int C, int VD, int VH,
                                       int VW, int RD, int RH, int RW, float cval) {
    const long long R = (long long)RD * RH * RW, V = (long long)VD * VH * VW;
    const int j = blockIdx.y;
    const int b = table[j * 4 + 0];
    if (b < 0) return;
    const int z0 = table[j * 4 + 1], y0 = table[j * 4 + 2], x0 = table[j * 4 + 3];
    const float* vb = vol + (long long)b * vol_bstride;
    for (long long ri = blockIdx.x * 256LL + threadIdx.x; ri < R; ri += (long long)gridDim.x * 256) {
        const int rx = (int)(ri % RW), ry = (int)((ri / RW) % RH), rz = (int)(ri / ((long long)RW * RH));
        const int z = z0 + rz, y = y0 + ry, x = x0 + rx;
        const bool in = (unsigned)z < (unsigned)VD && (unsigned)y < (unsigned)VH && (unsigned)x < (unsigned)VW;
        const long long v = ((long long)z * VH + y) * VW + x;
        for (int c = 0; c < C; ++c) {
            const float val = in ? vb[c * V + v] : cval;
            if (ldw > 0) DT<T>::st(win + ((long long)j * R + ri) * ldw + c, val);
            else DT<T>::st(win + ((long long)j * C + c) * R + ri, val);
        }
    }
}

// Every output voxel touched by the batch is owned by exactly one thread -- the thread of the FIRST window (in table
// order) that covers it -- which then adds the contributions of all windows of the batch covering that voxel, in
// table order, with the reference's two roundings per window (mul, then add): the same fp32 sequence as
// `out[idx] += imp * seg` executed window after window (/root/reference/engine/utils.py:146-148).
template <typename T, int CMAX>
__global__ __launch_bounds__(256) void sw_blend_batch_kernel(const T* __restrict__ win, long long ldw,
                                                             const float* __restrict__ imp, float* __restrict__ out,
                                                             long long out_bstride, float* __restrict__ cnt,
                                                             long long cnt_bstride, const int* __restrict__ table,
                                                             int nwin, int C, int VD, int VH, int VW, int RD, int RH,
                                                             int RW) {
#pragma clang fp contract(off)
    __shared__ int tab[256 * 4];
    for (int i = threadIdx.x; i < nwin * 4; i += 256) tab[i] = table[i];
    __syncthreads();
    const long long R = (long long)RD * RH * RW, V = (long long)VD * VH * VW;
    const bool vecw = ldw > 0 && ldw * (long long)sizeof(T) == 16 && C <= DT<T>::EPC && (((uintptr_t)win) & 15) == 0;
    const int j = blockIdx.y;
    const int b = tab[j * 4 + 0];
    if (b < 0) return;
    const int z0 = tab[j * 4 + 1], y0 = tab[j * 4 + 2], x0 = tab[j * 4 + 3];
    for (long long ri = blockIdx.x * 256LL + threadIdx.x; ri < R; ri += (long long)gridDim.x * 256) {
        const int rx = (int)(ri % RW), ry = (int)((ri / RW) % RH), rz = (int)(ri / ((long long)RW * RH));
        const int z = z0 + rz, y = y0 + ry, x = x0 + rx;
        bool owner = true;
        for (int k = 0; k < j; ++k) {
            if (tab[k * 4] != b) continue;
            const unsigned dz = (unsigned)(z - tab[k * 4 + 1]), dy = (unsigned)(y - tab[k * 4 + 2]),
                           dx = (unsigned)(x - tab[k * 4 + 3]);
            if (dz < (unsigned)RD && dy < (unsigned)RH && dx < (unsigned)RW) { owner = false; break; }
        }
        if (!owner) continue;
        const long long v = ((long long)z * VH + y) * VW + x;
        float* ob = out + (long long)b * out_bstride;
        float* cb = cnt + (long long)b * cnt_bstride;
        float o[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) o[c] = c < C ? ob[c * V + v] : 0.f;
        float cv = cb[v];
        for (int k = j; k < nwin; ++k) {
            if (tab[k * 4] != b) continue;
            const unsigned dz = (unsigned)(z - tab[k * 4 + 1]), dy = (unsigned)(y - tab[k * 4 + 2]),
                           dx = (unsigned)(x - tab[k * 4 + 3]);
            if (dz >= (unsigned)RD || dy >= (unsigned)RH || dx >= (unsigned)RW) continue;
            const long long pi = ((long long)dz * RH + dy) * RW + dx;
            const float w = imp[pi];
            float vals[CMAX];
            if (vecw) {   // 16-byte logits rows (the networks' channels-last logits): one load per (voxel, window)
                unpack_row<T, CMAX>(*(const u32x4_t*)(win + ((long long)k * R + pi) * ldw), C, vals);
            } else {
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    vals[c] = c < C ? (ldw > 0 ? DT<T>::ld(win + ((long long)k * R + pi) * ldw + c)
                                               : DT<T>::ld(win + ((long long)k * C + c) * R + pi)) : 0.f;
            }
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                if (c < C) {
                    const float prod = w * vals[c];
                    o[c] = o[c] + prod;
                }
            }
            cv = cv + w;
        }
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) ob[c * V + v] = o[c];
        cb[v] = cv;
    }
}

template <typename T, int CMAX>
int launch_blend_batch(const void* win, long long ldw, const float* imp, float* out, long long obs, float* cnt,
                              long long cbs, const int* table, int nwin, int C, int VD, int VH, int VW, int RD, int RH,
                              int RW, hipStream_t st) {
    const long long R = (long long)RD * RH * RW;
    long long gx = ceil_div_ll(R, 256);
    if (gx > 65535) gx = 65535;
    hipLaunchKernelGGL((sw_blend_batch_kernel<T, CMAX>), dim3((unsigned)gx, nwin), dim3(256), 0, st, (const T*)win, ldw,
                       imp, out, obs, cnt, cbs, table, nwin, C, VD, VH, VW, RD, RH, RW);
    MSSEG_CHECK_LAUNCH("sw_blend_batch");
    return MSSEG_OK;
}

inline int grid_for(long long total, int per_thread = 4) {
    long long b = ceil_div_ll(total, 256LL * per_thread);
    const long long cap = (long long)msseg_num_cus() * 16;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

template <typename T, int CMAX>
int launch_partials(const void* logits, long long ld, const void* labels, int label_dtype, float* partial, float* hard,
                    int N, long long S, int C, hipStream_t st, float* rows = nullptr, int* nblk_out = nullptr) {
    long long blocks = ceil_div_ll(S, 256LL * 8);
    const long long cap = (long long)msseg_num_cus() * 8 / N + 1;
    if (blocks > cap) blocks = cap;
    const long long vpb = ceil_div_ll(S, blocks);
    blocks = ceil_div_ll(S, vpb);
    if (nblk_out) *nblk_out = (int)blocks;
    hipLaunchKernelGGL((dice_ce_partials_kernel<T, CMAX>), dim3((unsigned)blocks, N), dim3(256), 0, st, (const T*)logits,
                       ld, labels, label_dtype, partial, hard, S, C, vpb, rows);
    MSSEG_CHECK_LAUNCH("dice_ce_partials");
    return MSSEG_OK;
}

template <typename T, int CMAX>
int launch_fwd(const void* logits, long long ld, const void* labels, int label_dtype, float* partial, float* hard,
               float* loss, int N, long long S, int C, float snr, float sdr, float* rows, hipStream_t st) {
    int nblk = 0;
    const int rc = launch_partials<T, CMAX>(logits, ld, labels, label_dtype, partial, hard, N, S, C, st, rows, &nblk);
    if (rc) return rc;
    hipLaunchKernelGGL(dice_ce_rows_finalize_kernel, dim3(1), dim3(256), 0, st, rows, nblk, partial, hard, loss, N, S, C,
                       snr, sdr);
    MSSEG_CHECK_LAUNCH("dice_ce_rows_finalize");
    return MSSEG_OK;
}

template <typename T, int CMAX>
int launch_bwd(const void* logits, long long ld, const void* labels, int label_dtype, const float* partial,
               const float* gscale, void* dlogits, long long ldd, int N, long long S, int C, float snr, float sdr,
               hipStream_t st) {
    long long blocks = ceil_div_ll(S, 256LL * 4);
    const long long cap = (long long)msseg_num_cus() * 16 / N + 1;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((dice_ce_bwd_kernel<T, CMAX>), dim3((unsigned)blocks, N), dim3(256), 0, st, (const T*)logits, ld,
                       labels, label_dtype, partial, gscale, (T*)dlogits, ldd, N, S, C, snr, sdr);
    MSSEG_CHECK_LAUNCH("dice_ce_bwd");
    return MSSEG_OK;
}

}  // namespace

#define DISPATCH_TC(dtype, C, FN, ...)                                                        \
    do {                                                                                      \
        if ((dtype) == MSSEG_F32) {                                                           \
            if ((C) <= 4) return FN<float, 4>(__VA_ARGS__);                                   \
            if ((C) <= 8) return FN<float, 8>(__VA_ARGS__);                                   \
            return FN<float, 16>(__VA_ARGS__);                                                \
        } else if ((dtype) == MSSEG_BF16) {                                                   \
            if ((C) <= 4) return FN<bf16_t, 4>(__VA_ARGS__);                                  \
            if ((C) <= 8) return FN<bf16_t, 8>(__VA_ARGS__);                                  \
            return FN<bf16_t, 16>(__VA_ARGS__);                                               \
        }                                                                                     \
        MSSEG_FAIL(MSSEG_EINVAL, "bad dtype %d", (int)(dtype));                               \
    } while (0)

extern "C" {

int msseg_dice_ce_partials(const void* logits, long long ld, int dtype, const void* labels, int label_dtype,
                           float* partial, float* hard, int N, long long S, int C, msseg_stream_t stream) {
    if (!logits || !labels || (!partial && !hard) || N < 1 || S < 1 || C < 1 || C > 16 || (ld != 0 && ld < C) ||
        label_dtype < 0 || label_dtype > 3)
        MSSEG_FAIL(MSSEG_EINVAL, "dice_ce_partials: bad args (C=%d must be 1..16)", C);
    DISPATCH_TC(dtype, C, launch_partials, logits, ld, labels, label_dtype, partial, hard, N, S, C, (hipStream_t)stream);
}

int msseg_dice_ce_fwd(const void* logits, long long ld, int dtype, const void* labels, int label_dtype, float* partial,
                      float* hard, float* loss, int N, long long S, int C, float smooth_nr, float smooth_dr, void* scratch,
                      size_t scratch_bytes, msseg_stream_t stream) {
    if (!logits || !labels || !partial || !loss || N < 1 || N > 8 || S < 1 || C < 1 || C > 16 || (ld != 0 && ld < C) ||
        label_dtype < 0 || label_dtype > 3)
        MSSEG_FAIL(MSSEG_EINVAL, "dice_ce_fwd: bad args (1 <= N <= 8, 1 <= C <= 16)");
    const size_t need = MSSEG_SCRATCH_COUNTER_BYTES + (size_t)N * ((size_t)msseg_num_cus() * 8 / N + 2) * C * 7 * 4;
    if (!scratch || ((uintptr_t)scratch & 255) || scratch_bytes < need)
        MSSEG_FAIL(MSSEG_EWORKSPACE, "dice_ce_fwd: scratch of %zu bytes needed", need);
    float* rows = (float*)((unsigned char*)scratch + MSSEG_SCRATCH_COUNTER_BYTES);
    DISPATCH_TC(dtype, C, launch_fwd, logits, ld, labels, label_dtype, partial, hard, loss, N, S, C, smooth_nr, smooth_dr,
                rows, (hipStream_t)stream);
}

int msseg_dice_ce_finalize(const float* partial, float* loss, int N, long long S, int C, float smooth_nr,
                           float smooth_dr, msseg_stream_t stream) {
    if (!partial || !loss || N < 1 || S < 1 || C < 1) MSSEG_FAIL(MSSEG_EINVAL, "dice_ce_finalize: bad args");
    hipLaunchKernelGGL(dice_ce_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, loss, N, S, C,
                       smooth_nr, smooth_dr);
    MSSEG_CHECK_LAUNCH("dice_ce_finalize");
    return MSSEG_OK;
}

int msseg_dice_ce_bwd(const void* logits, long long ld, int dtype, const void* labels, int label_dtype,
                      const float* partial, const float* gscale, void* dlogits, long long ldd, int N, long long S, int C,
                      float smooth_nr, float smooth_dr, msseg_stream_t stream) {
    if (!logits || !labels || !partial || !dlogits || N < 1 || S < 1 || C < 1 || C > 16 || (ld != 0 && ld < C) ||
        (ldd != 0 && ldd < C) || label_dtype < 0 || label_dtype > 3)
        MSSEG_FAIL(MSSEG_EINVAL, "dice_ce_bwd: bad args");
    DISPATCH_TC(dtype, C, launch_bwd, logits, ld, labels, label_dtype, partial, gscale, dlogits, ldd, N, S, C, smooth_nr,
                smooth_dr, (hipStream_t)stream);
}

int msseg_sw_blend(const void* win, long long ld, int dtype, const float* imp, float* out, float* cnt, int C, int VD,
                   int VH, int VW, int RD, int RH, int RW, int z0, int y0, int x0, msseg_stream_t stream) {
    if (!win || !imp || !out || !cnt || C < 1) MSSEG_FAIL(MSSEG_EINVAL, "sw_blend: bad args");
    if (z0 < 0 || y0 < 0 || x0 < 0 || z0 + RD > VD || y0 + RH > VH || x0 + RW > VW)
        MSSEG_FAIL(MSSEG_EINVAL, "sw_blend: window (%d,%d,%d)+(%d,%d,%d) outside volume (%d,%d,%d)", z0, y0, x0, RD, RH,
                   RW, VD, VH, VW);
    const int g = grid_for((long long)RD * RH * RW, 1);
    if (dtype == MSSEG_F32)
        hipLaunchKernelGGL(sw_blend_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)win, ld, imp,
                           out, cnt, C, VD, VH, VW, RD, RH, RW, z0, y0, x0);
    else if (dtype == MSSEG_BF16)
        hipLaunchKernelGGL(sw_blend_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)win, ld,
                           imp, out, cnt, C, VD, VH, VW, RD, RH, RW, z0, y0, x0);
    else MSSEG_FAIL(MSSEG_EINVAL, "sw_blend: bad dtype");
    MSSEG_CHECK_LAUNCH("sw_blend");
    return MSSEG_OK;
}

int msseg_sw_normalize(float* out, const float* cnt, int C, long long V, msseg_stream_t stream) {
    if (!out || !cnt || C < 1 || V < 1) MSSEG_FAIL(MSSEG_EINVAL, "sw_normalize: bad args");
    hipLaunchKernelGGL(sw_normalize_kernel, dim3(grid_for((long long)C * V)), dim3(256), 0, (hipStream_t)stream, out, cnt,
                       C, V);
    MSSEG_CHECK_LAUNCH("sw_normalize");
    return MSSEG_OK;
}

int msseg_sw_gather(const float* vol, void* win, int dtype, int C, int VD, int VH, int VW, int RD, int RH, int RW, int z0,
                    int y0, int x0, float cval, msseg_stream_t stream) {
    if (!vol || !win || C < 1) MSSEG_FAIL(MSSEG_EINVAL, "sw_gather: bad args");
    const int g = grid_for((long long)C * RD * RH * RW);
    if (dtype == MSSEG_F32)
        hipLaunchKernelGGL(sw_gather_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, vol, (float*)win, C, VD, VH,
                           VW, RD, RH, RW, z0, y0, x0, cval);
    else if (dtype == MSSEG_BF16)
        hipLaunchKernelGGL(sw_gather_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, vol, (bf16_t*)win, C, VD,
                           VH, VW, RD, RH, RW, z0, y0, x0, cval);
    else MSSEG_FAIL(MSSEG_EINVAL, "sw_gather: bad dtype");
    MSSEG_CHECK_LAUNCH("sw_gather");
    return MSSEG_OK;
}

int msseg_sw_blend_batch(const void* win, long long ldw, int dtype, const float* imp, float* out, long long out_bstride,
                         float* cnt, long long cnt_bstride, const int* table, int nwin, int C, int VD, int VH, int VW,
                         int RD, int RH, int RW, msseg_stream_t stream) {
    if (!win || !imp || !out || !cnt || !table) MSSEG_FAIL(MSSEG_EINVAL, "sw_blend_batch: null pointer");
    if (nwin < 1 || nwin > 256) MSSEG_FAIL(MSSEG_EINVAL, "sw_blend_batch: 1 <= nwin <= 256 windows per launch (got %d)", nwin);
    if (C < 1 || C > 16) MSSEG_FAIL(MSSEG_EINVAL, "sw_blend_batch: 1 <= C <= 16 classes");
    if (RD < 1 || RH < 1 || RW < 1 || RD > VD || RH > VH || RW > VW)
        MSSEG_FAIL(MSSEG_EINVAL, "sw_blend_batch: roi (%d,%d,%d) does not fit the volume (%d,%d,%d)", RD, RH, RW, VD, VH, VW);
    DISPATCH_TC(dtype, C, launch_blend_batch, win, ldw, imp, out, out_bstride, cnt, cnt_bstride, table, nwin, C, VD, VH,
                VW, RD, RH, RW, (hipStream_t)stream);
}

int msseg_sw_gather_batch(const float* vol, long long vol_bstride, void* win, long long ldw, int dtype, const int* table,
                          int nwin, int C, int VD, int VH, int VW, int RD, int RH, int RW, float cval,
                          msseg_stream_t stream) {
    if (!vol || !win || !table) MSSEG_FAIL(MSSEG_EINVAL, "sw_gather_batch: null pointer");
    if (nwin < 1 || nwin > 65535 || C < 1) MSSEG_FAIL(MSSEG_EINVAL, "sw_gather_batch: bad window / channel count");
    if (ldw != 0 && ldw < C) MSSEG_FAIL(MSSEG_EINVAL, "sw_gather_batch: ldw smaller than the channel count");
    const long long R = (long long)RD * RH * RW;
    long long gx = ceil_div_ll(R, 256LL * 2);
    if (gx > 65535) gx = 65535;
    if (dtype == MSSEG_F32)
        hipLaunchKernelGGL(sw_gather_batch_kernel<float>, dim3((unsigned)gx, nwin), dim3(256), 0, (hipStream_t)stream, vol,
                           vol_bstride, (float*)win, ldw, table, nwin, C, VD, VH, VW, RD, RH, RW, cval);
    else if (dtype == MSSEG_BF16)
        hipLaunchKernelGGL(sw_gather_batch_kernel<bf16_t>, dim3((unsigned)gx, nwin), dim3(256), 0, (hipStream_t)stream,
                           vol, vol_bstride, (bf16_t*)win, ldw, table, nwin, C, VD, VH, VW, RD, RH, RW, cval);
    else MSSEG_FAIL(MSSEG_EINVAL, "sw_gather_batch: bad dtype");
    MSSEG_CHECK_LAUNCH("sw_gather_batch");
    return MSSEG_OK;
}

}  // extern "C"
