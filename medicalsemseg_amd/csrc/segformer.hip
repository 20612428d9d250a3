// Kernels of the SegFormer3D family (/root/reference/models/backbones/segformer_backbone.py, segmentors/segformer_head*.py)
// that the Swin / UNet paths do not already provide, gfx950.  All are streaming (HBM / cache bound) kernels on
// channels-last tensors; reductions are fixed-order (no atomics).
//
//   interp_trilinear fwd/bwd : F.interpolate(mode='trilinear', align_corners=False) and its adjoint in gather form
//                              (every input voxel collects the output voxels that read it: deterministic).
//   kv_attention fwd         : softmax(q k^T * scale) v with FEW keys -- the spatial-reduction attention of
//                              segformer_backbone.py:96-117 (27 keys at 96^3): one thread per (head, query), keys and
//                              values of the (batch, head) in LDS, online softmax, no score tensor.
//   kv_attention bwd         : (1) per query: dq and the rows P, dS (fp32 [B, heads, N, M]); (2) per (batch, head, key):
//                              dk = sum_q dS q, dv = sum_q P dO, lanes = channels, fixed-order sum over query lanes.
//   scale_channels           : y[n, v, c] = x[n, v, c] * s[n, c] -- Dropout3d (channel dropout) with a given keep mask.
#include "common.h"

namespace {

template <typename T> struct Ch;
template <> struct Ch<bf16_t> {
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
template <> struct Ch<float> {
    static constexpr int E = 4;
    static MSSEG_DEVFN void load(const float* p, float* f) {
        const f32x4_t v = *(const f32x4_t*)p;
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = v[e];
    }
    static MSSEG_DEVFN void store(float* p, const float* f) { *(f32x4_t*)p = f32x4_t{f[0], f[1], f[2], f[3]}; }
};

// ---- trilinear ------------------------------------------------------------------------------------------------
struct Lin { int i0, i1; float l0, l1; };
// torch's area_pixel_compute_source_index (align_corners = False): src = max(scale * (o + 0.5) - 0.5, 0)
MSSEG_DEVFN Lin lin_of(int o, float scale, int in) {
    float src = scale * ((float)o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    Lin L;
    L.i0 = min((int)src, in - 1);
    L.i1 = min(L.i0 + 1, in - 1);
    L.l1 = src - (float)L.i0;
    L.l0 = 1.f - L.l1;
    return L;
}

struct InterpParams {
    const void* x; long long ldx;
    void* y; long long ldy;
    int N, ID, IH, IW, OD, OH, OW, C;
    float sd, sh, sw;      // in / out per dim
};

template <typename T>
__global__ __launch_bounds__(256) void interp_fwd_kernel(const InterpParams p) {
    constexpr int E = Ch<T>::E;
    const int nch = p.C / E;
    const long long total = (long long)p.N * p.OD * p.OH * p.OW * nch;
    const T* __restrict__ xg = (const T*)p.x;
    T* __restrict__ yg = (T*)p.y;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long v = i / nch;
        const int ow = (int)(v % p.OW); long long t = v / p.OW;
        const int oh = (int)(t % p.OH); t /= p.OH;
        const int od = (int)(t % p.OD); const int n = (int)(t / p.OD);
        const Lin Ld = lin_of(od, p.sd, p.ID), Lh = lin_of(oh, p.sh, p.IH), Lw = lin_of(ow, p.sw, p.IW);
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int d = a ? Ld.i1 : Ld.i0, h = b ? Lh.i1 : Lh.i0, w = c ? Lw.i1 : Lw.i0;
                    const float wt = (a ? Ld.l1 : Ld.l0) * (b ? Lh.l1 : Lh.l0) * (c ? Lw.l1 : Lw.l0);
                    float xv[E];
                    Ch<T>::load(xg + ((((long long)n * p.ID + d) * p.IH + h) * p.IW + w) * p.ldx + ch * E, xv);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] = fmaf(wt, xv[e], acc[e]);
                }
        Ch<T>::store(yg + v * p.ldy + ch * E, acc);
    }
}

// weight with which output index o reads input index i (0 when it does not)
MSSEG_DEVFN float wt_of(int o, int i, float scale, int in) {
    const Lin L = lin_of(o, scale, in);
    return (L.i0 == i ? L.l0 : 0.f) + (L.i1 == i ? L.l1 : 0.f);
}
MSSEG_DEVFN void range_of(int i, float scale, int out, int* lo, int* hi) {
    // outputs whose source coordinate lies in (i - 1, i + 1), widened by one on both sides against rounding
    int a = (int)floorf(((float)i - 0.5f) / scale - 0.5f) - 1;
    int b = (int)ceilf(((float)i + 1.5f) / scale - 0.5f) + 1;
    *lo = a < 0 ? 0 : a;
    *hi = b > out - 1 ? out - 1 : b;
}

// dx[n, i] = sum over outputs o of w(o, i) * dy[n, o]   (x = the low-resolution side; p.x = dx, p.y = dy)
template <typename T>
__global__ __launch_bounds__(256) void interp_bwd_kernel(const InterpParams p) {
    constexpr int E = Ch<T>::E;
    const int nch = p.C / E;
    const long long total = (long long)p.N * p.ID * p.IH * p.IW * nch;
    T* __restrict__ dxg = (T*)p.x;
    const T* __restrict__ dyg = (const T*)p.y;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long v = i / nch;
        const int iw = (int)(v % p.IW); long long t = v / p.IW;
        const int ih = (int)(t % p.IH); t /= p.IH;
        const int id = (int)(t % p.ID); const int n = (int)(t / p.ID);
        int d0, d1, h0, h1, w0, w1;
        range_of(id, p.sd, p.OD, &d0, &d1);
        range_of(ih, p.sh, p.OH, &h0, &h1);
        range_of(iw, p.sw, p.OW, &w0, &w1);
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        for (int od = d0; od <= d1; ++od) {
            const float wd = wt_of(od, id, p.sd, p.ID);
            if (wd == 0.f) continue;
            for (int oh = h0; oh <= h1; ++oh) {
                const float wh = wd * wt_of(oh, ih, p.sh, p.IH);
                if (wh == 0.f) continue;
                const T* row = dyg + (((long long)n * p.OD + od) * p.OH + oh) * p.OW * p.ldy + ch * E;
                for (int ow = w0; ow <= w1; ++ow) {
                    const float wt = wh * wt_of(ow, iw, p.sw, p.IW);
                    if (wt == 0.f) continue;
                    float g[E];
                    Ch<T>::load(row + (long long)ow * p.ldy, g);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] = fmaf(wt, g[e], acc[e]);
                }
            }
        }
        Ch<T>::store(dxg + v * p.ldx + ch * E, acc);
    }
}

// ---- attention with few keys ------------------------------------------------------------------------------------
constexpr int KT = 64;       // keys per LDS tile

struct KvParams {
    const void* q;           // [B, N, C]
    const void* kv;          // [B, M, 2C]: k = [..., :C], v = [..., C:], channel = head * hd + c
    void* o;                 // [B, N, C]
    float* lse;              // [B, heads, N]
    const void* dout;        // [B, N, C]
    void* dq;                // [B, N, C]
    float* P; float* dS;     // [B, heads, N, M]
    void* dkv;               // [B, M, 2C]
    int B, N, M, heads, hd;
    float scale;
};

template <typename T> MSSEG_DEVFN float ldf(const T* p) { return (float)*p; }

// one thread per query; grid = (ceil(N / 256), heads, B)
template <typename T, int HD, bool BWD>
__global__ __launch_bounds__(256) void kv_attn_kernel(const KvParams p) {
    __shared__ float ks[KT][HD + 1];
    __shared__ float vs[KT][HD + 1];
    const int h = blockIdx.y, b = blockIdx.z;
    const int C = p.heads * HD;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const bool live = qi < p.N;
    const long long qoff = ((long long)b * p.N + (live ? qi : 0)) * C + h * HD;
    const T* kvg = (const T*)p.kv + (long long)b * p.M * 2 * C + h * HD;
    float q[HD], acc[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) { q[c] = live ? ldf((const T*)p.q + qoff + c) : 0.f; acc[c] = 0.f; }
    float dO[BWD ? HD : 1];
    float Dsum = 0.f, lse = 0.f;
    if constexpr (BWD) {
#pragma unroll
        for (int c = 0; c < HD; ++c) {
            dO[c] = live ? ldf((const T*)p.dout + qoff + c) : 0.f;
            Dsum += dO[c] * (live ? ldf((const T*)p.o + qoff + c) : 0.f);
        }
        lse = live ? p.lse[((long long)b * p.heads + h) * p.N + qi] : 0.f;
    }
    float mx = -3.0e38f, l = 0.f;
    for (int j0 = 0; j0 < p.M; j0 += KT) {
        const int nk = min(KT, p.M - j0);
        __syncthreads();
        for (int i = threadIdx.x; i < nk * HD; i += 256) {
            const int j = i / HD, c = i - j * HD;
            ks[j][c] = ldf(kvg + (long long)(j0 + j) * 2 * C + c);
            vs[j][c] = ldf(kvg + (long long)(j0 + j) * 2 * C + C + c);
        }
        __syncthreads();
        for (int j = 0; j < nk; ++j) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < HD; ++c) s = fmaf(q[c], ks[j][c], s);
            s *= p.scale;
            if constexpr (!BWD) {
                const float mn = fmaxf(mx, s);
                const float corr = __expf(mx - mn), e = __expf(s - mn);
                l = l * corr + e;
#pragma unroll
                for (int c = 0; c < HD; ++c) acc[c] = fmaf(e, vs[j][c], acc[c] * corr);
                mx = mn;
            } else {
                const float pj = __expf(s - lse);
                float dp = 0.f;
#pragma unroll
                for (int c = 0; c < HD; ++c) dp = fmaf(dO[c], vs[j][c], dp);
                const float ds = pj * (dp - Dsum) * p.scale;
#pragma unroll
                for (int c = 0; c < HD; ++c) acc[c] = fmaf(ds, ks[j][c], acc[c]);
                if (live) {
                    const long long r = (((long long)b * p.heads + h) * p.N + qi) * p.M + j0 + j;
                    p.P[r] = pj;
                    p.dS[r] = ds;
                }
            }
        }
    }
    if (!live) return;
    if constexpr (!BWD) {
        const float inv = 1.f / l;
#pragma unroll
        for (int c = 0; c < HD; ++c) ((T*)p.o)[qoff + c] = (T)(acc[c] * inv);
        p.lse[((long long)b * p.heads + h) * p.N + qi] = mx + __logf(l);
    } else {
#pragma unroll
        for (int c = 0; c < HD; ++c) ((T*)p.dq)[qoff + c] = (T)acc[c];
    }
}

// grid = (M, heads, B); thread = (channel c, query lane): dk[j][c] = sum_q dS[q][j] q[q][c], dv[j][c] = sum_q P[q][j] dO[q][c]
template <typename T>
__global__ __launch_bounds__(256) void kv_attn_bwd_kv_kernel(const KvParams p) {
    __shared__ float red[2][256];
    const int j = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int HD = p.hd, C = p.heads * HD;
    const int lanes = 256 / HD;                   // query lanes
    const int c = threadIdx.x % HD, ql = threadIdx.x / HD;
    float ak = 0.f, av = 0.f;
    if (ql < lanes) {
        const T* qg = (const T*)p.q + (long long)b * p.N * C + h * HD + c;
        const T* og = (const T*)p.dout + (long long)b * p.N * C + h * HD + c;
        const float* Pg = p.P + ((long long)b * p.heads + h) * p.N * p.M + j;
        const float* Sg = p.dS + ((long long)b * p.heads + h) * p.N * p.M + j;
        for (int qi = ql; qi < p.N; qi += lanes) {
            ak = fmaf(Sg[(long long)qi * p.M], ldf(qg + (long long)qi * C), ak);
            av = fmaf(Pg[(long long)qi * p.M], ldf(og + (long long)qi * C), av);
        }
    }
    red[0][threadIdx.x] = ak;
    red[1][threadIdx.x] = av;
    __syncthreads();
    if (threadIdx.x < HD) {
        float sk = 0.f, sv = 0.f;
        for (int l2 = 0; l2 < lanes; ++l2) { sk += red[0][l2 * HD + c]; sv += red[1][l2 * HD + c]; }
        T* dg = (T*)p.dkv + ((long long)b * p.M + j) * 2 * C + h * HD + c;
        dg[0] = (T)sk;
        dg[C] = (T)sv;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_channels_kernel(const T* x, const float* s, T* y, long long S, int C, long long total) {
    constexpr int E = Ch<T>::E;
    const int nch = C / E;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long long v = i / nch;
        const long long n = v / S;
        float f[E];
        Ch<T>::load(x + v * C + ch * E, f);
#pragma unroll
        for (int e = 0; e < E; ++e) f[e] *= s[n * C + ch * E + e];
        Ch<T>::store(y + v * C + ch * E, f);
    }
}

int grid_of(long long total) {
    long long gx = (total + 255) / 256;
    const long long cap = (long long)msseg_num_cus() * 16;
    if (gx > cap) gx = cap;
    return (int)(gx < 1 ? 1 : gx);
}

int interp_check(const void* lo, long long ldlo, const void* hi, long long ldhi, int N, int ID, int IH, int IW, int OD, int OH,
                 int OW, int C, int dtype, const char* what) {
    if (!lo || !hi) MSSEG_FAIL(MSSEG_EINVAL, "%s: null pointer", what);
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "%s: bad dtype", what);
    const int epc = dtype == MSSEG_F32 ? 4 : 8;
    if (N < 1 || ID < 1 || IH < 1 || IW < 1 || OD < 1 || OH < 1 || OW < 1 || C < 1 || C % epc)
        MSSEG_FAIL(MSSEG_EINVAL, "%s: bad shape (channels must be a multiple of %d)", what, epc);
    if (ldlo < C || ldhi < C || ldlo % epc || ldhi % epc || ((uintptr_t)lo & 15) || ((uintptr_t)hi & 15))
        MSSEG_FAIL(MSSEG_EINVAL, "%s: tensors must be 16-byte aligned with voxel strides that are multiples of %d", what, epc);
    return MSSEG_OK;
}

}  // namespace

extern "C" {

int msseg_interp_trilinear_fwd(const void* x, long long ldx, void* y, long long ldy, int N, int ID, int IH, int IW, int OD,
                               int OH, int OW, int C, int dtype, msseg_stream_t stream) {
    int rc = interp_check(x, ldx, y, ldy, N, ID, IH, IW, OD, OH, OW, C, dtype, "interp_trilinear_fwd");
    if (rc) return rc;
    InterpParams p{x, ldx, y, ldy, N, ID, IH, IW, OD, OH, OW, C, (float)ID / OD, (float)IH / OH, (float)IW / OW};
    const long long total = (long long)N * OD * OH * OW * (C / (dtype == MSSEG_F32 ? 4 : 8));
    if (dtype == MSSEG_F32) hipLaunchKernelGGL(interp_fwd_kernel<float>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(interp_fwd_kernel<bf16_t>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, p);
    MSSEG_CHECK_LAUNCH("interp_trilinear_fwd");
    return MSSEG_OK;
}

int msseg_interp_trilinear_bwd(const void* dy, long long lddy, void* dx, long long lddx, int N, int ID, int IH, int IW, int OD,
                               int OH, int OW, int C, int dtype, msseg_stream_t stream) {
    int rc = interp_check(dx, lddx, dy, lddy, N, ID, IH, IW, OD, OH, OW, C, dtype, "interp_trilinear_bwd");
    if (rc) return rc;
    InterpParams p{dx, lddx, (void*)dy, lddy, N, ID, IH, IW, OD, OH, OW, C, (float)ID / OD, (float)IH / OH, (float)IW / OW};
    const long long total = (long long)N * ID * IH * IW * (C / (dtype == MSSEG_F32 ? 4 : 8));
    if (dtype == MSSEG_F32) hipLaunchKernelGGL(interp_bwd_kernel<float>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(interp_bwd_kernel<bf16_t>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, p);
    MSSEG_CHECK_LAUNCH("interp_trilinear_bwd");
    return MSSEG_OK;
}

static int kv_check(const KvParams& p, int dtype, const char* what) {
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "%s: bad dtype", what);
    if (p.B < 1 || p.N < 1 || p.M < 1 || p.heads < 1) MSSEG_FAIL(MSSEG_EINVAL, "%s: bad shape", what);
    if (p.hd != 16 && p.hd != 32 && p.hd != 48 && p.hd != 64) MSSEG_FAIL(MSSEG_EINVAL, "%s: head_dim %d not in {16, 32, 48, 64}", what, p.hd);
    if (p.heads > 65535 || p.B > 65535) MSSEG_FAIL(MSSEG_EINVAL, "%s: too many heads / samples", what);
    return MSSEG_OK;
}

#define KV_DISPATCH(BWDFLAG)                                                                                            \
    do {                                                                                                                \
        const dim3 g((p.N + 255) / 256, p.heads, p.B);                                                                  \
        if (dtype == MSSEG_F32) {                                                                                       \
            switch (p.hd) {                                                                                             \
                case 16: hipLaunchKernelGGL((kv_attn_kernel<float, 16, BWDFLAG>), g, dim3(256), 0, (hipStream_t)stream, p); break; \
                case 32: hipLaunchKernelGGL((kv_attn_kernel<float, 32, BWDFLAG>), g, dim3(256), 0, (hipStream_t)stream, p); break; \
                case 48: hipLaunchKernelGGL((kv_attn_kernel<float, 48, BWDFLAG>), g, dim3(256), 0, (hipStream_t)stream, p); break; \
                default: hipLaunchKernelGGL((kv_attn_kernel<float, 64, BWDFLAG>), g, dim3(256), 0, (hipStream_t)stream, p); break; \
            }                                                                                                           \
        } else {                                                                                                        \
            switch (p.hd) {                                                                                             \
                case 16: hipLaunchKernelGGL((kv_attn_kernel<bf16_t, 16, BWDFLAG>), g, dim3(256), 0, (hipStream_t)stream, p); break; \
                case 32: hipLaunchKernelGGL((kv_attn_kernel<bf16_t, 32, BWDFLAG>), g, dim3(256), 0, (hipStream_t)stream, p); break; \
                case 48: hipLaunchKernelGGL((kv_attn_kernel<bf16_t, 48, BWDFLAG>), g, dim3(256), 0, (hipStream_t)stream, p); break; \
                default: hipLaunchKernelGGL((kv_attn_kernel<bf16_t, 64, BWDFLAG>), g, dim3(256), 0, (hipStream_t)stream, p); break; \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)

int msseg_kv_attention_fwd(const void* q, const void* kv, void* o, float* lse, int B, int N, int M, int heads, int head_dim,
                           float scale, int dtype, msseg_stream_t stream) {
    if (!q || !kv || !o || !lse) MSSEG_FAIL(MSSEG_EINVAL, "kv_attention_fwd: null pointer");
    KvParams p{};
    p.q = q; p.kv = kv; p.o = o; p.lse = lse; p.B = B; p.N = N; p.M = M; p.heads = heads; p.hd = head_dim; p.scale = scale;
    int rc = kv_check(p, dtype, "kv_attention_fwd");
    if (rc) return rc;
    KV_DISPATCH(false);
    MSSEG_CHECK_LAUNCH("kv_attention_fwd");
    return MSSEG_OK;
}

size_t msseg_kv_attention_bwd_workspace_bytes(int B, int N, int M, int heads) {
    return (size_t)2 * B * heads * N * M * sizeof(float);
}

int msseg_kv_attention_bwd(const void* q, const void* kv, const void* o, const float* lse, const void* dout, void* dq, void* dkv,
                           int B, int N, int M, int heads, int head_dim, float scale, void* workspace, size_t workspace_bytes,
                           int dtype, msseg_stream_t stream) {
    if (!q || !kv || !o || !lse || !dout || !dq || !dkv) MSSEG_FAIL(MSSEG_EINVAL, "kv_attention_bwd: null pointer");
    KvParams p{};
    p.q = q; p.kv = kv; p.o = (void*)o; p.lse = (float*)lse; p.dout = dout; p.dq = dq; p.dkv = dkv;
    p.B = B; p.N = N; p.M = M; p.heads = heads; p.hd = head_dim; p.scale = scale;
    int rc = kv_check(p, dtype, "kv_attention_bwd");
    if (rc) return rc;
    const size_t need = msseg_kv_attention_bwd_workspace_bytes(B, N, M, heads);
    if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < need)
        MSSEG_FAIL(MSSEG_EWORKSPACE, "kv_attention_bwd: needs a workspace of %zu bytes", need);
    p.P = (float*)workspace;
    p.dS = p.P + (size_t)B * heads * N * M;
    KV_DISPATCH(true);
    MSSEG_CHECK_LAUNCH("kv_attention_bwd (queries)");
    const dim3 g2(M, heads, B);
    if (dtype == MSSEG_F32) hipLaunchKernelGGL(kv_attn_bwd_kv_kernel<float>, g2, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(kv_attn_bwd_kv_kernel<bf16_t>, g2, dim3(256), 0, (hipStream_t)stream, p);
    MSSEG_CHECK_LAUNCH("kv_attention_bwd (keys)");
    return MSSEG_OK;
}

int msseg_scale_channels(const void* x, const float* scale, void* y, int N, long long S, int C, int dtype, msseg_stream_t stream) {
    if (!x || !scale || !y) MSSEG_FAIL(MSSEG_EINVAL, "scale_channels: null pointer");
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "scale_channels: bad dtype");
    const int epc = dtype == MSSEG_F32 ? 4 : 8;
    if (N < 1 || S < 1 || C < 1 || C % epc || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
        MSSEG_FAIL(MSSEG_EINVAL, "scale_channels: dense 16-byte aligned tensors with C %% %d == 0", epc);
    const long long total = (long long)N * S * (C / epc);
    if (dtype == MSSEG_F32)
        hipLaunchKernelGGL(scale_channels_kernel<float>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, scale, (float*)y, S, C, total);
    else
        hipLaunchKernelGGL(scale_channels_kernel<bf16_t>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scale, (bf16_t*)y, S, C, total);
    MSSEG_CHECK_LAUNCH("scale_channels");
    return MSSEG_OK;
}

}  // extern "C"
