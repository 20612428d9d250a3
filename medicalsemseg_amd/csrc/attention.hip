// Swin 3-D shifted-window attention (forward + backward), LayerNorm over channels, GELU.
//
// Window attention replaces swin_nnformer.py:235-289 + :128-196 of the reference between the qkv Linear and the
// proj Linear:  pad -> roll(-shift) -> window_partition -> softmax(q k^T * scale + bias[+mask]) v -> window_reverse
// -> roll(+shift) -> crop.  Nothing is materialised: the kernel addresses tokens of the [B,S,H,W,3C] qkv tensor
// through the (shift, window, position) map, synthesises padded tokens (qkv = qkv bias), derives the -100 region
// mask from coordinates, and streams keys through LDS with an online softmax, so the [B_,h,N,N] score tensor
// never exists.  This is the exact-fp32-math version used for both dtypes (bf16 I/O, fp32 accumulate); one
// workgroup = one window, one thread = one query (forward / dQ) or one key (dK, dV).
#include "attention_common.h"

#include <stdlib.h>

using namespace msseg_attn;

namespace {

constexpr int HD_MAX = 32;

// ---------------------------------------------------------------------------------------------------------
// forward: grid (nW, B), block 256.  Loops heads; per head K,V of the window live in LDS (fp32).
// ---------------------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(256) void win_attn_fwd_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* kS = (float*)smem;            // [N][HD]
    float* vS = kS + p.N * HD;           // [N][HD]
    float* tabS = vS + p.N * HD;         // [M3] bias table of the current head
    int* tok = (int*)(tabS + p.M3);      // [N] voxel index or -1
    int* regS = tok + p.N;               // [N]
    int* codeS = regS + p.N;             // [N]
    const int nW = p.nWs * p.nWh * p.nWw;
    const int m = 2 * p.bws - 1;
    const int off = ((p.bws - 1) * m + (p.bws - 1)) * m + (p.bws - 1);
    for (int wb = blockIdx.x; wb < p.nwin_total; wb += gridDim.x) {
        const int w = wb % nW, b = wb / nW;
        const int wx = w % p.nWw, wy = (w / p.nWw) % p.nWh, wz = w / (p.nWw * p.nWh);
        const T* qkv = (const T*)p.qkv + (long long)b * p.S * p.H * p.W * 3 * p.C;
        T* out = (T*)p.out + (long long)b * p.S * p.H * p.W * p.C;
        __syncthreads();
        for (int i = threadIdx.x; i < p.N; i += 256) {
            int rg, cd;
            tok[i] = window_token(p, wz, wy, wx, i, rg, cd);
            regS[i] = rg;
            codeS[i] = cd;
        }
        for (int h = blockIdx.y; h < p.heads; h += gridDim.y) {
            __syncthreads();
            for (int i = threadIdx.x; i < p.M3; i += 256) tabS[i] = p.table[(long long)i * p.heads + h];
            for (int i = threadIdx.x; i < p.N * HD; i += 256) {
                const int j = i / HD, e = i % HD;
                const int t = tok[j];
                const int ck = p.C + h * HD + e, cv = 2 * p.C + h * HD + e;
                float kv, vv;
                if (t >= 0) {
                    kv = DT<T>::ld(qkv + (long long)t * 3 * p.C + ck);
                    vv = DT<T>::ld(qkv + (long long)t * 3 * p.C + cv);
                } else {
                    kv = p.qkv_bias ? p.qkv_bias[ck] : 0.f;
                    vv = p.qkv_bias ? p.qkv_bias[cv] : 0.f;
                    if (sizeof(T) == 2) { kv = (float)(bf16_t)kv; vv = (float)(bf16_t)vv; }
                }
                kS[i] = kv;
                vS[i] = vv;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < p.N; i += 256) {
                const int t = tok[i];
                float q[HD], o[HD];
#pragma unroll
                for (int e = 0; e < HD; ++e) {
                    float qv;
                    if (t >= 0) qv = DT<T>::ld(qkv + (long long)t * 3 * p.C + h * HD + e);
                    else {
                        qv = p.qkv_bias ? p.qkv_bias[h * HD + e] : 0.f;
                        if (sizeof(T) == 2) qv = (float)(bf16_t)qv;
                    }
                    q[e] = qv * p.scale;
                    o[e] = 0.f;
                }
                const int ri = regS[i], ci = codeS[i] + off;
                float mx = -INFINITY, l = 0.f;
                for (int j = 0; j < p.N; ++j) {
                    float sc = 0.f;
#pragma unroll
                    for (int e = 0; e < HD; ++e) sc += q[e] * kS[j * HD + e];
                    sc += tabS[ci - codeS[j]];
                    if (p.use_mask && regS[j] != ri) sc += -100.f;
                    const float mn = fmaxf(mx, sc);
                    const float a = expf(mx - mn), pe = expf(sc - mn);
                    l = l * a + pe;
#pragma unroll
                    for (int e = 0; e < HD; ++e) o[e] = o[e] * a + pe * vS[j * HD + e];
                    mx = mn;
                }
                const float inv = 1.f / l;
                p.lse[((long long)wb * p.heads + h) * p.N + i] = mx + logf(l);
                if (t >= 0) {
#pragma unroll
                    for (int e = 0; e < HD; ++e) DT<T>::st(out + (long long)t * p.C + h * HD + e, o[e] * inv);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward: grid (nW, B), block 256.  Per head: Q, K, V, dO rows, lse, delta in LDS; phase A thread = query
// (dQ, dbias), phase B thread = key (dK, dV).  Gradients of padded tokens are dropped (they are constants).
// ---------------------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(256) void win_attn_bwd_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* qS = (float*)smem;          // [N][HD]  (already scaled)
    float* kS = qS + p.N * HD;
    float* vS = kS + p.N * HD;
    float* dS = vS + p.N * HD;         // dO [N][HD]
    float* lseS = dS + p.N * HD;       // [N]
    float* delS = lseS + p.N;          // [N]  delta_i = dO_i . O_i
    float* tabS = delS + p.N;          // [M3]
    int* tok = (int*)(tabS + p.M3);
    int* regS = tok + p.N;
    int* codeS = regS + p.N;
    float* dtabS = (float*)(codeS + p.N);  // [M3] or [heads][M3]
    const int nW = p.nWs * p.nWh * p.nWw;
    const int m = 2 * p.bws - 1;
    const int off = ((p.bws - 1) * m + (p.bws - 1)) * m + (p.bws - 1);
    const long long vol = (long long)p.S * p.H * p.W;
    const int ndt = p.dtable ? (p.dtab_all_heads ? p.heads * p.M3 : p.M3) : 0;
    for (int i = threadIdx.x; i < ndt; i += 256) dtabS[i] = 0.f;
    for (int wb = blockIdx.x; wb < p.nwin_total; wb += gridDim.x) {
        const int w = wb % nW, b = wb / nW;
        const int wx = w % p.nWw, wy = (w / p.nWw) % p.nWh, wz = w / (p.nWw * p.nWh);
        const T* qkv = (const T*)p.qkv + b * vol * 3 * p.C;
        const T* outp = (const T*)p.out + b * vol * p.C;
        const T* dout = (const T*)p.dout + b * vol * p.C;
        T* dqkv = (T*)p.dqkv + b * vol * 3 * p.C;
        __syncthreads();
        for (int i = threadIdx.x; i < p.N; i += 256) {
            int rg, cd;
            tok[i] = window_token(p, wz, wy, wx, i, rg, cd);
            regS[i] = rg;
            codeS[i] = cd;
        }
        for (int h = blockIdx.y; h < p.heads; h += gridDim.y) {
            __syncthreads();
            float* dtab = p.dtab_all_heads ? dtabS + h * p.M3 : dtabS;
            for (int i = threadIdx.x; i < p.M3; i += 256) tabS[i] = p.table[(long long)i * p.heads + h];
            for (int i = threadIdx.x; i < p.N * HD; i += 256) {
                const int j = i / HD, e = i % HD;
                const int t = tok[j];
                const int c = h * HD + e;
                float qv, kv, vv, dov = 0.f;
                if (t >= 0) {
                    const T* row = qkv + (long long)t * 3 * p.C;
                    qv = DT<T>::ld(row + c); kv = DT<T>::ld(row + p.C + c); vv = DT<T>::ld(row + 2 * p.C + c);
                    dov = DT<T>::ld(dout + (long long)t * p.C + c);
                } else {
                    qv = p.qkv_bias ? p.qkv_bias[c] : 0.f;
                    kv = p.qkv_bias ? p.qkv_bias[p.C + c] : 0.f;
                    vv = p.qkv_bias ? p.qkv_bias[2 * p.C + c] : 0.f;
                    if (sizeof(T) == 2) { qv = (float)(bf16_t)qv; kv = (float)(bf16_t)kv; vv = (float)(bf16_t)vv; }
                }
                qS[i] = qv * p.scale; kS[i] = kv; vS[i] = vv; dS[i] = dov;
            }
            for (int i = threadIdx.x; i < p.N; i += 256) {
                lseS[i] = p.lse[((long long)wb * p.heads + h) * p.N + i];
                const int t = tok[i];
                float d = 0.f;
                if (t >= 0) {
#pragma unroll
                    for (int e = 0; e < HD; ++e)
                        d += DT<T>::ld(dout + (long long)t * p.C + h * HD + e) * DT<T>::ld(outp + (long long)t * p.C + h * HD + e);
                }
                delS[i] = d;
            }
            __syncthreads();
            // phase A: thread = query i -> dQ_i ; dtable[rel(i,j)] += dS_ij
            for (int i = threadIdx.x; i < p.N; i += 256) {
                float q[HD], dq[HD], dO[HD];
#pragma unroll
                for (int e = 0; e < HD; ++e) { q[e] = qS[i * HD + e]; dO[e] = dS[i * HD + e]; dq[e] = 0.f; }
                const float lse = lseS[i], del = delS[i];
                const int ri = regS[i], ci = codeS[i] + off;
                const bool live = tok[i] >= 0;
                for (int j = 0; j < p.N; ++j) {
                    float sc = 0.f, dp = 0.f;
#pragma unroll
                    for (int e = 0; e < HD; ++e) { sc += q[e] * kS[j * HD + e]; dp += dO[e] * vS[j * HD + e]; }
                    const int ti = ci - codeS[j];
                    sc += tabS[ti];
                    if (p.use_mask && regS[j] != ri) sc += -100.f;
                    const float pr = expf(sc - lse);
                    const float ds = pr * (dp - del);
#pragma unroll
                    for (int e = 0; e < HD; ++e) dq[e] += ds * kS[j * HD + e];
                    if (p.dtable && live) atomicAdd(&dtab[ti], ds);
                }
                const int t = tok[i];
                if (t >= 0) {
#pragma unroll
                    for (int e = 0; e < HD; ++e) DT<T>::st(dqkv + (long long)t * 3 * p.C + h * HD + e, dq[e] * p.scale);
                }
            }
            // phase B: thread = key j -> dK_j, dV_j
            for (int j = threadIdx.x; j < p.N; j += 256) {
                float k[HD], v[HD], dk[HD], dv[HD];
#pragma unroll
                for (int e = 0; e < HD; ++e) { k[e] = kS[j * HD + e]; v[e] = vS[j * HD + e]; dk[e] = dv[e] = 0.f; }
                const int rj = regS[j], cj = off - codeS[j];
                for (int i = 0; i < p.N; ++i) {
                    if (tok[i] < 0) continue;  // padded queries produce no output, hence no gradient
                    float sc = 0.f, dp = 0.f;
#pragma unroll
                    for (int e = 0; e < HD; ++e) { sc += qS[i * HD + e] * k[e]; dp += dS[i * HD + e] * v[e]; }
                    sc += tabS[codeS[i] + cj];
                    if (p.use_mask && regS[i] != rj) sc += -100.f;
                    const float pr = expf(sc - lseS[i]);
                    const float ds = pr * (dp - delS[i]);
#pragma unroll
                    for (int e = 0; e < HD; ++e) { dv[e] += pr * dS[i * HD + e]; dk[e] += ds * qS[i * HD + e]; }
                }
                const int t = tok[j];
                if (t >= 0) {
                    T* row = dqkv + (long long)t * 3 * p.C;
#pragma unroll
                    for (int e = 0; e < HD; ++e) {
                        DT<T>::st(row + p.C + h * HD + e, dk[e]);   // qS already carries the scale
                        DT<T>::st(row + 2 * p.C + h * HD + e, dv[e]);
                    }
                }
            }
            if (p.dtable && !p.dtab_all_heads) {
                __syncthreads();
                for (int i = threadIdx.x; i < p.M3; i += 256) {
                    const float v = dtabS[i];
                    if (v != 0.f) atomicAdd(&p.dtable[(long long)i * p.heads + h], v);
                    dtabS[i] = 0.f;
                }
            }
        }
    }
    if (p.dtable && p.dtab_all_heads) {
        __syncthreads();
        for (int i = threadIdx.x; i < p.heads * p.M3; i += 256) {
            const float v = dtabS[i];
            const int h = i / p.M3, e = i % p.M3;
            if (v != 0.f) atomicAdd(&p.dtable[(long long)e * p.heads + h], v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm over channels (one thread per token) and GELU
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, long long ldx, const float* g,
                                                            const float* bta, T* __restrict__ y, long long ldy,
                                                            float* mean, float* rstd, long long rows, int C, float eps) {
    for (long long r = blockIdx.x * 256LL + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
        const T* xr = x + r * ldx;
        float s = 0.f, s2 = 0.f;
        for (int c = 0; c < C; ++c) { const float v = DT<T>::ld(xr + c); s += v; s2 += v * v; }
        const float mu = s / C;
        float var = s2 / C - mu * mu;
        var = var > 0.f ? var : 0.f;
        const float rs = rsqrtf(var + eps);
        if (mean) { mean[r] = mu; rstd[r] = rs; }
        T* yr = y + r * ldy;
        for (int c = 0; c < C; ++c)
            DT<T>::st(yr + c, (DT<T>::ld(xr + c) - mu) * rs * (g ? g[c] : 1.f) + (bta ? bta[c] : 0.f));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, long long ldx, const float* g,
                                                            const float* mean, const float* rstd,
                                                            const T* __restrict__ dy, long long lddy, T* __restrict__ dx,
                                                            long long lddx, float* dg, float* db, long long rows, int C) {
    for (long long r = blockIdx.x * 256LL + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
        const T* xr = x + r * ldx;
        const T* dr = dy + r * lddy;
        const float mu = mean[r], rs = rstd[r];
        float a = 0.f, b = 0.f;
        for (int c = 0; c < C; ++c) {
            const float dyg = DT<T>::ld(dr + c) * (g ? g[c] : 1.f);
            const float xh = (DT<T>::ld(xr + c) - mu) * rs;
            a += dyg; b += dyg * xh;
        }
        a /= C; b /= C;
        T* dxr = dx + r * lddx;
        for (int c = 0; c < C; ++c) {
            const float d = DT<T>::ld(dr + c);
            const float xh = (DT<T>::ld(xr + c) - mu) * rs;
            DT<T>::st(dxr + c, rs * (d * (g ? g[c] : 1.f) - a - xh * b));
        }
    }
}

// Vector forms: LPR lanes (a power of two <= 64) share one token row, 16-byte chunks, the row stays in registers between
// the statistics and the normalisation (one pass over HBM); reductions by xor-shuffles inside the lane group.
struct LnVecParams {
    const void* x; long long ldx; const float* g; const float* b; void* y; long long ldy;
    float* mean; float* rstd; const void* dy; long long lddy; long long rows; int C, lpr, nch; float eps;
    const void* add; long long ldadd;   // backward: dx = T(layernorm backward) + add (the gradient of a residual branch that left x)
};

template <typename T> MSSEG_DEVFN void ln_load(const T* p, float (&v)[DT<T>::EPC]) {
    const u32x4_t raw = *(const u32x4_t*)p;
    if constexpr (sizeof(T) == 2) {
        const bf16x8_t h = __builtin_bit_cast(bf16x8_t, raw);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
    } else {
        const f32x4_t f = __builtin_bit_cast(f32x4_t, raw);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = f[e];
    }
}
template <typename T> MSSEG_DEVFN void ln_store(T* p, const float (&v)[DT<T>::EPC]) {
    if constexpr (sizeof(T) == 2) {
        bf16x8_t h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (bf16_t)v[e];
        *(u32x4_t*)p = __builtin_bit_cast(u32x4_t, h);
    } else {
        *(f32x4_t*)p = f32x4_t{v[0], v[1], v[2], v[3]};
    }
}

template <typename T, int MAXCH, bool BWD>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const LnVecParams p) {
    constexpr int EPC = DT<T>::EPC;
    const int lane = threadIdx.x & 63, sub = lane & (p.lpr - 1), rows_per_wave = 64 / p.lpr;
    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * 4;
    const float invC = 1.0f / (float)p.C;
    // per-chunk affine parameters of this lane
    float gv[MAXCH][EPC], bv[MAXCH][EPC];
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        const int ch = sub + k * p.lpr;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            gv[k][e] = (ch < p.nch && p.g) ? p.g[ch * EPC + e] : 1.f;
            bv[k][e] = (!BWD && ch < p.nch && p.b) ? p.b[ch * EPC + e] : 0.f;
        }
    }
    for (long long r0 = wave_id * rows_per_wave; r0 < p.rows; r0 += nwaves * rows_per_wave) {
        const long long r = r0 + lane / p.lpr;
        const bool rok = r < p.rows;
        float xv[MAXCH][EPC], dv[MAXCH][EPC];
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int ch = sub + k * p.lpr;
            const bool ok = rok && ch < p.nch;
#pragma unroll
            for (int e = 0; e < EPC; ++e) xv[k][e] = dv[k][e] = 0.f;
            if (ok) {
                ln_load<T>((const T*)p.x + r * p.ldx + ch * EPC, xv[k]);
                if constexpr (BWD) ln_load<T>((const T*)p.dy + r * p.lddy + ch * EPC, dv[k]);
            }
        }
        if constexpr (!BWD) {
            // the row is in registers: mean first, then the centred sum of squares (E[x^2] - mean^2 loses
            // eps * mean^2 / var, which shows on wide rows with a large mean)
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < MAXCH; ++k)
#pragma unroll
                for (int e = 0; e < EPC; ++e) s += xv[k][e];
            for (int o = p.lpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float mu = s * invC;
            float s2 = 0.f;
#pragma unroll
            for (int k = 0; k < MAXCH; ++k) {
                if (sub + k * p.lpr < p.nch) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) { const float d = xv[k][e] - mu; s2 += d * d; }
                }
            }
            for (int o = p.lpr >> 1; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
            const float var = s2 * invC;
            const float rs = rsqrtf(var + p.eps);
            if (rok && sub == 0 && p.mean) { p.mean[r] = mu; p.rstd[r] = rs; }
#pragma unroll
            for (int k = 0; k < MAXCH; ++k) {
                const int ch = sub + k * p.lpr;
                if (rok && ch < p.nch) {
                    float o[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) o[e] = (xv[k][e] - mu) * rs * gv[k][e] + bv[k][e];
                    ln_store<T>((T*)p.y + r * p.ldy + ch * EPC, o);
                }
            }
        } else {
            const float mu = rok ? p.mean[r] : 0.f, rs = rok ? p.rstd[r] : 0.f;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int k = 0; k < MAXCH; ++k)
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float dyg = dv[k][e] * gv[k][e];
                    a += dyg;
                    b += dyg * ((xv[k][e] - mu) * rs);
                }
            for (int o = p.lpr >> 1; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
            a *= invC; b *= invC;
#pragma unroll
            for (int k = 0; k < MAXCH; ++k) {
                const int ch = sub + k * p.lpr;
                if (rok && ch < p.nch) {
                    float o[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) o[e] = rs * (dv[k][e] * gv[k][e] - a - (xv[k][e] - mu) * rs * b);
                    if (p.add) {   // the sum autograd forms for a tensor with two consumers: both terms rounded to T, then added
                        float r2[EPC];
                        ln_load<T>((const T*)p.add + r * p.ldadd + ch * EPC, r2);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) o[e] = (float)(T)o[e] + r2[e];
                    }
                    ln_store<T>((T*)p.y + r * p.ldy + ch * EPC, o);
                }
            }
        }
    }
}

// returns false when the vector form does not apply (unaligned rows / odd channel counts)
template <typename T, bool BWD> bool launch_ln_vec(LnVecParams p, hipStream_t stream) {
    constexpr int EPC = DT<T>::EPC;
    const size_t esz = sizeof(T);
    if (p.C % EPC || ((uintptr_t)p.x & 15) || ((uintptr_t)p.y & 15) || (p.ldx * esz) % 16 || (p.ldy * esz) % 16) return false;
    if (BWD && (((uintptr_t)p.dy & 15) || (p.lddy * esz) % 16)) return false;
    p.nch = p.C / EPC;
    int lpr = 1;
    while (lpr < p.nch && lpr < 64) lpr <<= 1;
    p.lpr = lpr;
    const int maxch = (p.nch + lpr - 1) / lpr;
    // (rows wider than 8 x 64 chunks -- 4096 bf16 channels -- fall back to the scalar kernel: one thread per row.  The 3072-wide
    //  LayerNorm of the MONAI variant's last patch merging ran there at 1 ms per pass for 54 rows.)
    if (maxch > 8) return false;
    const long long waves = (p.rows + (64 / lpr) - 1) / (64 / lpr);
    long long blocks = (waves + 3) / 4;
    const long long cap = (long long)msseg_num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks);
    switch (maxch) {
        case 1: hipLaunchKernelGGL((layernorm_vec_kernel<T, 1, BWD>), grid, dim3(256), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((layernorm_vec_kernel<T, 2, BWD>), grid, dim3(256), 0, stream, p); break;
        case 3: hipLaunchKernelGGL((layernorm_vec_kernel<T, 3, BWD>), grid, dim3(256), 0, stream, p); break;
        case 4: hipLaunchKernelGGL((layernorm_vec_kernel<T, 4, BWD>), grid, dim3(256), 0, stream, p); break;
        case 5: case 6: hipLaunchKernelGGL((layernorm_vec_kernel<T, 6, BWD>), grid, dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL((layernorm_vec_kernel<T, 8, BWD>), grid, dim3(256), 0, stream, p); break;
    }
    return true;
}

// exact-erf GELU (nn.GELU of the patch merging, swin_nnformer.py:306) and its gradient; 16-byte chunks where the tensors are
// aligned (one element per thread and trip ran at a fraction of the memory rate), the scalar form for the rest
template <typename T, bool BWD>
__global__ void gelu_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, long long n) {
    constexpr int EPC = DT<T>::EPC;
    const bool vec = ((((uintptr_t)x) | ((uintptr_t)out) | (BWD ? (uintptr_t)dy : 0)) & 15) == 0;
    const long long nvec = vec ? n / EPC : 0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        float v[EPC], g[EPC], o[EPC];
        ln_load<T>(x + i * EPC, v);
        if constexpr (BWD) ln_load<T>(dy + i * EPC, g);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float cdf = 0.5f * (1.f + erff(v[e] * 0.70710678118654752f));
            if constexpr (!BWD) o[e] = v[e] * cdf;
            else o[e] = g[e] * (cdf + v[e] * 0.3989422804014327f * expf(-0.5f * v[e] * v[e]));
        }
        ln_store<T>(out + i * EPC, o);
    }
    for (long long i = nvec * EPC + blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = DT<T>::ld(x + i);
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
        if constexpr (!BWD) DT<T>::st(out + i, v * cdf);
        else DT<T>::st(out + i, DT<T>::ld(dy + i) * (cdf + v * 0.3989422804014327f * expf(-0.5f * v * v)));
    }
}

inline int grid_for(long long total, int per_thread = 4) {
    long long b = ceil_div_ll(total, 256LL * per_thread);
    const long long cap = (long long)msseg_num_cus() * 16;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

int fill_attn(AttnParams& p, int B, int S, int H, int W, int C, int heads, int ws, int shift, int bias_ws = 0) {
    if (B < 1 || S < 1 || H < 1 || W < 1 || heads < 1 || C % heads || ws < 1 || shift < 0 || shift >= ws)
        MSSEG_FAIL(MSSEG_EINVAL, "window_attention: bad shape");
    if (bias_ws == 0) bias_ws = ws;
    if (bias_ws < ws) MSSEG_FAIL(MSSEG_EINVAL, "window_attention: bias window %d smaller than the window %d", bias_ws, ws);
    p.bws = bias_ws;
    p.B = B; p.S = S; p.H = H; p.W = W; p.C = C; p.heads = heads; p.hd = C / heads; p.ws = ws; p.shift = shift;
    p.nWs = ceil_div(S, ws); p.nWh = ceil_div(H, ws); p.nWw = ceil_div(W, ws);
    p.Sp = p.nWs * ws; p.Hp = p.nWh * ws; p.Wp = p.nWw * ws;
    p.N = ws * ws * ws;
    p.M3 = (2 * bias_ws - 1) * (2 * bias_ws - 1) * (2 * bias_ws - 1);
    p.nwin_total = p.nWs * p.nWh * p.nWw * B;
    p.scale = 1.0f / sqrtf((float)p.hd);
    p.use_mask = shift > 0;
    if (p.hd != 8 && p.hd != 16 && p.hd != 32) MSSEG_FAIL(MSSEG_EINVAL, "window_attention: head_dim %d not in {8,16,32}", p.hd);
    if (p.N > 512) MSSEG_FAIL(MSSEG_EINVAL, "window_attention: window of %d tokens too large", p.N);
    return MSSEG_OK;
}

#define ATTN_LAUNCH(KERN, T_, SMEM)                                                                      \
    do {                                                                                                 \
        int gx__ = p.nwin_total < msseg_num_cus() * 2 ? p.nwin_total : msseg_num_cus() * 2;              \
        int gy__ = (msseg_num_cus() * 2 + gx__ - 1) / gx__;                                              \
        if (gy__ > p.heads) gy__ = p.heads;                                                              \
        if (p.dtab_all_heads) gy__ = 1;                                                                  \
        dim3 grid(gx__, gy__);                                                                            \
        if (p.hd == 8) { if (int rc__ = attn_set_lds((const void*)KERN<T_, 8>, SMEM)) return rc__;          \
            hipLaunchKernelGGL((KERN<T_, 8>), grid, dim3(256), SMEM, (hipStream_t)stream, p); }              \
        else if (p.hd == 16) { if (int rc__ = attn_set_lds((const void*)KERN<T_, 16>, SMEM)) return rc__;   \
            hipLaunchKernelGGL((KERN<T_, 16>), grid, dim3(256), SMEM, (hipStream_t)stream, p); }             \
        else { if (int rc__ = attn_set_lds((const void*)KERN<T_, 32>, SMEM)) return rc__;                   \
            hipLaunchKernelGGL((KERN<T_, 32>), grid, dim3(256), SMEM, (hipStream_t)stream, p); }             \
    } while (0)

}  // namespace

extern "C" {

static int attn_set_lds(const void* kern, size_t smem) {
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        MSSEG_FAIL(MSSEG_ELAUNCH, "window_attention: cannot set %zu bytes of dynamic LDS", smem);
    return MSSEG_OK;
}

int msseg_window_attention_fwd(const void* qkv, const float* qkv_bias, const float* table, void* out, float* lse, int B,
                               int S, int H, int W, int C, int heads, int ws, int shift, int dtype,
                               msseg_stream_t stream) {
    return msseg_window_attention_fwd2(qkv, qkv_bias, table, out, lse, B, S, H, W, C, heads, ws, shift, ws, dtype, stream);
}

int msseg_window_attention_fwd2(const void* qkv, const float* qkv_bias, const float* table, void* out, float* lse, int B,
                                int S, int H, int W, int C, int heads, int ws, int shift, int bias_ws, int dtype,
                                msseg_stream_t stream) {
    if (!qkv || !table || !out || !lse) MSSEG_FAIL(MSSEG_EINVAL, "window_attention_fwd: null pointer");
    AttnParams p{};
    if (int rc = fill_attn(p, B, S, H, W, C, heads, ws, shift, bias_ws)) return rc;
    p.qkv = qkv; p.qkv_bias = qkv_bias; p.table = table; p.out = out; p.lse = lse;
    const size_t smem = (size_t)p.N * p.hd * 2 * 4 + (size_t)p.M3 * 4 + (size_t)p.N * 3 * 4;
    if (smem > 160 * 1024) MSSEG_FAIL(MSSEG_EINVAL, "window_attention_fwd: window too large for LDS");
    if (dtype == MSSEG_BF16 && (p.hd == 16 || p.hd == 32) && p.N <= 352 && p.M3 <= 4095 && (C % 8) == 0 &&
        !getenv("MSSEG_ATTN_NO_MFMA")) {
        // bf16: QK^T and PV on the matrix cores (attention_mfma.hip)
        return msseg_window_attention_fwd_mfma(p, (hipStream_t)stream);
    }
    if (dtype == MSSEG_F32) ATTN_LAUNCH(win_attn_fwd_kernel, float, smem);
    else if (dtype == MSSEG_BF16) ATTN_LAUNCH(win_attn_fwd_kernel, bf16_t, smem);
    else MSSEG_FAIL(MSSEG_EINVAL, "window_attention_fwd: bad dtype");
    MSSEG_CHECK_LAUNCH("window_attention_fwd");
    return MSSEG_OK;
}

static bool attn_bwd_on_mfma(const AttnParams& p, int C, int dtype) {
    {   // LDS image of the MFMA backward (attention_mfma.hip launch_bwd): head dim 32 at 343 tokens does not fit
        const int nkt = (p.N + 31) / 32;
        const size_t np = (size_t)(nkt == 1 ? 1 : (nkt == 2 ? 2 : (nkt <= 4 ? 4 : (nkt <= 7 ? 7 : 11)))) * 32;
        if (7 * np * p.hd * 2 + 2 * np * 4 + (size_t)2 * p.M3 * 4 + np * 8 + 16 > 160 * 1024) return false;
    }
    return dtype == MSSEG_BF16 && (p.hd == 16 || p.hd == 32) && p.N <= 352 && p.M3 <= 4095 && (C % 8) == 0 &&
           !getenv("MSSEG_ATTN_NO_MFMA") && !getenv("MSSEG_ATTN_BWD_NO_MFMA");
}

size_t msseg_window_attention_bwd_workspace_bytes(int B, int S, int H, int W, int C, int heads, int ws, int shift, int dtype) {
    AttnParams p{};
    if (fill_attn(p, B, S, H, W, C, heads, ws, shift) != MSSEG_OK) return 0;
    if (!attn_bwd_on_mfma(p, C, dtype) || getenv("MSSEG_ATTN_BWD_NO_WS")) return 0;
    return msseg_window_attention_bwd_mfma_ws_bytes(p);
}

int msseg_window_attention_bwd_ws(const void* qkv, const float* qkv_bias, const float* table, const void* out,
                                  const float* lse, const void* dout, void* dqkv, float* dtable, int B, int S, int H, int W,
                                  int C, int heads, int ws, int shift, int dtype, void* workspace, size_t workspace_bytes,
                                  msseg_stream_t stream) {
    return msseg_window_attention_bwd2(qkv, qkv_bias, table, out, lse, dout, dqkv, dtable, B, S, H, W, C, heads, ws, shift, ws,
                                       dtype, workspace, workspace_bytes, stream);
}

int msseg_window_attention_bwd2(const void* qkv, const float* qkv_bias, const float* table, const void* out,
                                const float* lse, const void* dout, void* dqkv, float* dtable, int B, int S, int H, int W,
                                int C, int heads, int ws, int shift, int bias_ws, int dtype, void* workspace,
                                size_t workspace_bytes, msseg_stream_t stream) {
    if (!qkv || !table || !out || !lse || !dout || !dqkv) MSSEG_FAIL(MSSEG_EINVAL, "window_attention_bwd: null pointer");
    AttnParams p{};
    if (int rc = fill_attn(p, B, S, H, W, C, heads, ws, shift, bias_ws)) return rc;
    p.qkv = qkv; p.qkv_bias = qkv_bias; p.table = table; p.out = (void*)out; p.lse = (float*)lse; p.dout = dout;
    p.dqkv = dqkv; p.dtable = dtable;
    if (attn_bwd_on_mfma(p, C, dtype)) {
        // bf16: all five contractions of the backward on the matrix cores (attention_mfma.hip); with a workspace the
        // table gradient is a window sum + gather instead of LDS float atomics
        if (workspace != nullptr && dtable != nullptr) {
            if (workspace_bytes < msseg_window_attention_bwd_mfma_ws_bytes(p))
                MSSEG_FAIL(MSSEG_EINVAL, "window_attention_bwd: workspace too small (%zu < %zu bytes)", workspace_bytes,
                           msseg_window_attention_bwd_mfma_ws_bytes(p));
            msseg_window_attention_bwd_mfma_carve(p, workspace);
        }
        return msseg_window_attention_bwd_mfma(p, (hipStream_t)stream);
    }
    const size_t base = (size_t)p.N * p.hd * 4 * 4 + (size_t)p.N * 5 * 4 + (size_t)p.M3 * 4;
    p.dtab_all_heads = (base + (size_t)heads * p.M3 * 4 <= 96 * 1024) ? 1 : 0;
    const size_t smem = base + (size_t)(p.dtab_all_heads ? heads : 1) * p.M3 * 4;
    if (smem > 160 * 1024) MSSEG_FAIL(MSSEG_EINVAL, "window_attention_bwd: window too large for LDS (%zu bytes)", smem);
    if (dtype == MSSEG_F32) ATTN_LAUNCH(win_attn_bwd_kernel, float, smem);
    else if (dtype == MSSEG_BF16) ATTN_LAUNCH(win_attn_bwd_kernel, bf16_t, smem);
    else MSSEG_FAIL(MSSEG_EINVAL, "window_attention_bwd: bad dtype");
    MSSEG_CHECK_LAUNCH("window_attention_bwd");
    return MSSEG_OK;
}

int msseg_window_attention_bwd(const void* qkv, const float* qkv_bias, const float* table, const void* out,
                               const float* lse, const void* dout, void* dqkv, float* dtable, int B, int S, int H, int W,
                               int C, int heads, int ws, int shift, int dtype, msseg_stream_t stream) {
    return msseg_window_attention_bwd_ws(qkv, qkv_bias, table, out, lse, dout, dqkv, dtable, B, S, H, W, C, heads, ws, shift,
                                         dtype, nullptr, 0, stream);
}

int msseg_layernorm_fwd(const void* x, long long ldx, const float* gamma, const float* beta, void* y, long long ldy,
                        float* mean, float* rstd, long long rows, int C, float eps, int dtype, msseg_stream_t stream) {
    if (!x || !y || rows < 1 || C < 1) MSSEG_FAIL(MSSEG_EINVAL, "layernorm_fwd: bad args");
    if (dtype == MSSEG_F32 || dtype == MSSEG_BF16) {
        LnVecParams v{};
        v.x = x; v.ldx = ldx; v.g = gamma; v.b = beta; v.y = y; v.ldy = ldy; v.mean = mean; v.rstd = rstd;
        v.rows = rows; v.C = C; v.eps = eps;
        const bool done = dtype == MSSEG_F32 ? launch_ln_vec<float, false>(v, (hipStream_t)stream)
                                             : launch_ln_vec<bf16_t, false>(v, (hipStream_t)stream);
        if (done) { MSSEG_CHECK_LAUNCH("layernorm_fwd"); return MSSEG_OK; }
    }
    const int g = grid_for(rows, 1);
    if (dtype == MSSEG_F32)
        hipLaunchKernelGGL(layernorm_fwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx,
                           gamma, beta, (float*)y, ldy, mean, rstd, rows, C, eps);
    else if (dtype == MSSEG_BF16)
        hipLaunchKernelGGL(layernorm_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                           gamma, beta, (bf16_t*)y, ldy, mean, rstd, rows, C, eps);
    else MSSEG_FAIL(MSSEG_EINVAL, "layernorm_fwd: bad dtype");
    MSSEG_CHECK_LAUNCH("layernorm_fwd");
    return MSSEG_OK;
}

int msseg_layernorm_bwd(const void* x, long long ldx, const float* gamma, const float* mean, const float* rstd,
                        const void* dy, long long lddy, void* dx, long long lddx, long long rows, int C, int dtype,
                        msseg_stream_t stream) {
    if (!x || !mean || !rstd || !dy || !dx || rows < 1 || C < 1) MSSEG_FAIL(MSSEG_EINVAL, "layernorm_bwd: bad args");
    if (dtype == MSSEG_F32 || dtype == MSSEG_BF16) {
        LnVecParams v{};
        v.x = x; v.ldx = ldx; v.g = gamma; v.y = dx; v.ldy = lddx; v.mean = (float*)mean; v.rstd = (float*)rstd;
        v.dy = dy; v.lddy = lddy; v.rows = rows; v.C = C;
        const bool done = dtype == MSSEG_F32 ? launch_ln_vec<float, true>(v, (hipStream_t)stream)
                                             : launch_ln_vec<bf16_t, true>(v, (hipStream_t)stream);
        if (done) { MSSEG_CHECK_LAUNCH("layernorm_bwd"); return MSSEG_OK; }
    }
    float* dgamma = nullptr;
    float* dbeta = nullptr;
    const int g = grid_for(rows, 1);
    const size_t smem = 0;
    if (dtype == MSSEG_F32)
        hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(g), dim3(256), smem, (hipStream_t)stream, (const float*)x, ldx,
                           gamma, mean, rstd, (const float*)dy, lddy, (float*)dx, lddx, dgamma, dbeta, rows, C);
    else if (dtype == MSSEG_BF16)
        hipLaunchKernelGGL(layernorm_bwd_kernel<bf16_t>, dim3(g), dim3(256), smem, (hipStream_t)stream, (const bf16_t*)x,
                           ldx, gamma, mean, rstd, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, dgamma, dbeta, rows, C);
    else MSSEG_FAIL(MSSEG_EINVAL, "layernorm_bwd: bad dtype");
    MSSEG_CHECK_LAUNCH("layernorm_bwd");
    return MSSEG_OK;
}

int msseg_layernorm_bwd_add(const void* x, long long ldx, const float* gamma, const float* mean, const float* rstd,
                            const void* dy, long long lddy, const void* add, long long ldadd, void* dx, long long lddx,
                            long long rows, int C, int dtype, msseg_stream_t stream) {
    if (!x || !mean || !rstd || !dy || !dx || !add || rows < 1 || C < 1) MSSEG_FAIL(MSSEG_EINVAL, "layernorm_bwd_add: bad args");
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "layernorm_bwd_add: bad dtype");
    const size_t esz = dtype == MSSEG_F32 ? 4 : 2;
    if (((uintptr_t)add & 15) || (ldadd * esz) % 16) MSSEG_FAIL(MSSEG_EINVAL, "layernorm_bwd_add: `add` must have 16-byte aligned rows");
    LnVecParams v{};
    v.x = x; v.ldx = ldx; v.g = gamma; v.y = dx; v.ldy = lddx; v.mean = (float*)mean; v.rstd = (float*)rstd;
    v.dy = dy; v.lddy = lddy; v.rows = rows; v.C = C; v.add = add; v.ldadd = ldadd;
    const bool done = dtype == MSSEG_F32 ? launch_ln_vec<float, true>(v, (hipStream_t)stream)
                                         : launch_ln_vec<bf16_t, true>(v, (hipStream_t)stream);
    if (!done) MSSEG_FAIL(MSSEG_EINVAL, "layernorm_bwd_add: rows of %d channels do not take the vector kernel (16-byte chunks, <= 4096 channels)", C);
    MSSEG_CHECK_LAUNCH("layernorm_bwd_add");
    return MSSEG_OK;
}

int msseg_gelu_fwd(const void* x, void* y, long long n, int dtype, msseg_stream_t stream) {
    if (!x || !y || n < 1) MSSEG_FAIL(MSSEG_EINVAL, "gelu_fwd: bad args");
    if (dtype == MSSEG_F32)
        hipLaunchKernelGGL((gelu_kernel<float, false>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (const float*)nullptr, (float*)y, n);
    else if (dtype == MSSEG_BF16)
        hipLaunchKernelGGL((gelu_kernel<bf16_t, false>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, (const bf16_t*)nullptr, (bf16_t*)y, n);
    else MSSEG_FAIL(MSSEG_EINVAL, "gelu_fwd: bad dtype");
    MSSEG_CHECK_LAUNCH("gelu_fwd");
    return MSSEG_OK;
}

int msseg_gelu_bwd(const void* x, const void* dy, void* dx, long long n, int dtype, msseg_stream_t stream) {
    if (!x || !dy || !dx || n < 1) MSSEG_FAIL(MSSEG_EINVAL, "gelu_bwd: bad args");
    if (dtype == MSSEG_F32)
        hipLaunchKernelGGL((gelu_kernel<float, true>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                           (const float*)dy, (float*)dx, n);
    else if (dtype == MSSEG_BF16)
        hipLaunchKernelGGL((gelu_kernel<bf16_t, true>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n);
    else MSSEG_FAIL(MSSEG_EINVAL, "gelu_bwd: bad dtype");
    MSSEG_CHECK_LAUNCH("gelu_bwd");
    return MSSEG_OK;
}

}  // extern "C"
