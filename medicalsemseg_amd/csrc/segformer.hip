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

// Adjoint of the interpolation, one axis at a time (trilinear weights are a product of three 1-D weights, so the adjoint is
// the composition of three 1-D adjoints): dx[outer, i, inner] = sum over outputs o of w(o, i) * dy[outer, o, inner].
// Gather form -- every input index collects the outputs that read it -- hence deterministic.  Four channels per thread;
// intermediates are fp32.
struct Interp1Params {
    const void* dy; long long lddy;    // [outer, Lo, inner, C] rows of lddy elements
    void* dx; long long lddx;          // [outer, Li, inner, C]
    long long outer, inner;
    int Lo, Li, C;
    float scale;                       // Li / Lo
};

template <typename T> MSSEG_DEVFN void ld4(const T* p, float* f);
template <> MSSEG_DEVFN void ld4<float>(const float* p, float* f) {
    const f32x4_t v = *(const f32x4_t*)p;
    f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
}
template <> MSSEG_DEVFN void ld4<bf16_t>(const bf16_t* p, float* f) {
    const bf16x4_t v = *(const bf16x4_t*)p;
    f[0] = (float)v[0]; f[1] = (float)v[1]; f[2] = (float)v[2]; f[3] = (float)v[3];
}
template <typename T> MSSEG_DEVFN void st4(T* p, const float* f);
template <> MSSEG_DEVFN void st4<float>(float* p, const float* f) { *(f32x4_t*)p = f32x4_t{f[0], f[1], f[2], f[3]}; }
template <> MSSEG_DEVFN void st4<bf16_t>(bf16_t* p, const float* f) {
    *(bf16x4_t*)p = bf16x4_t{(bf16_t)f[0], (bf16_t)f[1], (bf16_t)f[2], (bf16_t)f[3]};
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void interp1_bwd_kernel(const Interp1Params p) {
    const int nch = p.C / 4;
    const long long total = p.outer * p.Li * p.inner * nch;
    const TI* __restrict__ gy = (const TI*)p.dy;
    TO* __restrict__ gx = (TO*)p.dx;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int ch = (int)(t % nch);
        long long v = t / nch;
        const long long in = v % p.inner; v /= p.inner;
        const int i = (int)(v % p.Li);
        const long long ou = v / p.Li;
        int lo, hi;
        range_of(i, p.scale, p.Lo, &lo, &hi);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int o = lo; o <= hi; ++o) {
            const float wt = wt_of(o, i, p.scale, p.Li);
            if (wt == 0.f) continue;
            float g[4];
            ld4<TI>(gy + ((ou * p.Lo + o) * p.inner + in) * p.lddy + ch * 4, g);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(wt, g[e], acc[e]);
        }
        st4<TO>(gx + ((ou * p.Li + i) * p.inner + in) * p.lddx + ch * 4, acc);
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

// dk[j][c] = sum_q dS[q][j] q[q][c], dv[j][c] = sum_q P[q][j] dO[q][c].
// Step 1, grid = (query chunks, heads, B): thread = (channel c, query lane) keeps the sums of up to KT keys in registers
// over the queries of its chunk (P / dS rows of the chunk staged in LDS), lanes are added in a fixed order, the workgroup
// leaves a partial [M][2][hd].  Step 2 adds the chunks in order.  Deterministic.
constexpr int QCH = 128;     // queries per workgroup

template <typename T>
__global__ __launch_bounds__(256) void kv_attn_bwd_kv_kernel(const KvParams p, float* part) {
    __shared__ float sP[QCH][KT + 1];
    __shared__ float sS[QCH][KT + 1];
    __shared__ float red[2][256];
    const int qc = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int HD = p.hd, C = p.heads * HD;
    const int lanes = 256 / HD;
    const int c = threadIdx.x % HD, ql = threadIdx.x / HD;
    const int q0 = qc * QCH, nq = min(QCH, p.N - q0);
    const T* qg = (const T*)p.q + ((long long)b * p.N + q0) * C + h * HD + c;
    const T* og = (const T*)p.dout + ((long long)b * p.N + q0) * C + h * HD + c;
    const long long rbase = (((long long)b * p.heads + h) * p.N + q0) * p.M;
    float* out = part + ((((long long)qc * gridDim.z + b) * p.heads + h) * p.M) * 2 * HD;
    for (int j0 = 0; j0 < p.M; j0 += KT) {
        const int nk = min(KT, p.M - j0);
        __syncthreads();
        for (int i = threadIdx.x; i < nq * nk; i += 256) {
            const int qi = i / nk, j = i - qi * nk;
            sP[qi][j] = p.P[rbase + (long long)qi * p.M + j0 + j];
            sS[qi][j] = p.dS[rbase + (long long)qi * p.M + j0 + j];
        }
        __syncthreads();
        float ak[KT], av[KT];
#pragma unroll
        for (int j = 0; j < KT; ++j) ak[j] = av[j] = 0.f;
        if (ql < lanes) {
            for (int qi = ql; qi < nq; qi += lanes) {
                const float qv = ldf(qg + (long long)qi * C), ov = ldf(og + (long long)qi * C);
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    if (j < nk) {
                        ak[j] = fmaf(sS[qi][j], qv, ak[j]);
                        av[j] = fmaf(sP[qi][j], ov, av[j]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            if (j < nk) {                    // nk is uniform over the workgroup
                __syncthreads();
                red[0][threadIdx.x] = ak[j];
                red[1][threadIdx.x] = av[j];
                __syncthreads();
                if (threadIdx.x < HD) {
                    float sk = 0.f, sv = 0.f;
                    for (int l2 = 0; l2 < lanes; ++l2) { sk += red[0][l2 * HD + c]; sv += red[1][l2 * HD + c]; }
                    out[((long long)(j0 + j) * 2 + 0) * HD + c] = sk;
                    out[((long long)(j0 + j) * 2 + 1) * HD + c] = sv;
                }
            }
        }
    }
}

// grid = (M, heads, B), HD threads: dkv[b][j][h*hd + c] (k half, v half) = sum over query chunks, in order
template <typename T>
__global__ void kv_attn_bwd_kv_finalize_kernel(const KvParams p, const float* part, int nqc) {
    const int j = blockIdx.x, h = blockIdx.y, b = blockIdx.z, c = threadIdx.x;
    const int HD = p.hd, C = p.heads * HD;
    if (c >= HD) return;
    float sk = 0.f, sv = 0.f;
    for (int qc = 0; qc < nqc; ++qc) {
        const float* src = part + (((((long long)qc * gridDim.z + b) * p.heads + h) * p.M + j) * 2) * HD;
        sk += src[c];
        sv += src[HD + c];
    }
    T* dg = (T*)p.dkv + ((long long)b * p.M + j) * 2 * C + h * HD + c;
    dg[0] = (T)sk;
    dg[C] = (T)sv;
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

template <typename TI, typename TO>
int interp1_launch(const void* dy, long long lddy, void* dx, long long lddx, long long outer, int Lo, int Li, long long inner,
                          int C, hipStream_t stream) {
    Interp1Params p{dy, lddy, dx, lddx, outer, inner, Lo, Li, C, (float)Li / Lo};
    hipLaunchKernelGGL((interp1_bwd_kernel<TI, TO>), dim3(grid_of(outer * Li * inner * (C / 4))), dim3(256), 0, stream, p);
    MSSEG_CHECK_LAUNCH("interp_trilinear_bwd");
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

size_t msseg_interp_trilinear_bwd_workspace_bytes(int N, int ID, int IH, int IW, int OD, int OH, int OW, int C) {
    (void)ID;
    return ((size_t)N * OD * OH * IW + (size_t)N * OD * IH * IW) * C * sizeof(float);
}

int msseg_interp_trilinear_bwd(const void* dy, long long lddy, void* dx, long long lddx, int N, int ID, int IH, int IW, int OD,
                               int OH, int OW, int C, void* workspace, size_t workspace_bytes, int dtype, msseg_stream_t stream) {
    int rc = interp_check(dx, lddx, dy, lddy, N, ID, IH, IW, OD, OH, OW, C, dtype, "interp_trilinear_bwd");
    if (rc) return rc;
    const size_t need = msseg_interp_trilinear_bwd_workspace_bytes(N, ID, IH, IW, OD, OH, OW, C);
    if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < need)
        MSSEG_FAIL(MSSEG_EWORKSPACE, "interp_trilinear_bwd: needs a 16-byte aligned workspace of %zu bytes", need);
    float* t1 = (float*)workspace;                               // [N, OD, OH, IW, C]
    float* t2 = t1 + (size_t)N * OD * OH * IW * C;               // [N, OD, IH, IW, C]
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MSSEG_F32) {
        rc = interp1_launch<float, float>(dy, lddy, t1, C, (long long)N * OD * OH, OW, IW, 1, C, st);
        if (!rc) rc = interp1_launch<float, float>(t1, C, t2, C, (long long)N * OD, OH, IH, IW, C, st);
        if (!rc) rc = interp1_launch<float, float>(t2, C, dx, lddx, N, OD, ID, (long long)IH * IW, C, st);
    } else {
        rc = interp1_launch<bf16_t, float>(dy, lddy, t1, C, (long long)N * OD * OH, OW, IW, 1, C, st);
        if (!rc) rc = interp1_launch<float, float>(t1, C, t2, C, (long long)N * OD, OH, IH, IW, C, st);
        if (!rc) rc = interp1_launch<float, bf16_t>(t2, C, dx, lddx, N, OD, ID, (long long)IH * IW, C, st);
    }
    return rc;
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

size_t msseg_kv_attention_bwd_workspace_bytes(int B, int N, int M, int heads, int head_dim) {
    const size_t nqc = (size_t)(N + QCH - 1) / QCH;
    return ((size_t)2 * B * heads * N * M + nqc * B * heads * M * 2 * head_dim) * sizeof(float);
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
    const size_t need = msseg_kv_attention_bwd_workspace_bytes(B, N, M, heads, head_dim);
    if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < need)
        MSSEG_FAIL(MSSEG_EWORKSPACE, "kv_attention_bwd: needs a workspace of %zu bytes", need);
    p.P = (float*)workspace;
    p.dS = p.P + (size_t)B * heads * N * M;
    KV_DISPATCH(true);
    MSSEG_CHECK_LAUNCH("kv_attention_bwd (queries)");
    const int nqc = (N + QCH - 1) / QCH;
    float* part = p.dS + (size_t)B * heads * N * M;
    const dim3 g2(nqc, heads, B), g3(M, heads, B);
    if (dtype == MSSEG_F32) {
        hipLaunchKernelGGL(kv_attn_bwd_kv_kernel<float>, g2, dim3(256), 0, (hipStream_t)stream, p, part);
        hipLaunchKernelGGL(kv_attn_bwd_kv_finalize_kernel<float>, g3, dim3(64), 0, (hipStream_t)stream, p, (const float*)part, nqc);
    } else {
        hipLaunchKernelGGL(kv_attn_bwd_kv_kernel<bf16_t>, g2, dim3(256), 0, (hipStream_t)stream, p, part);
        hipLaunchKernelGGL(kv_attn_bwd_kv_finalize_kernel<bf16_t>, g3, dim3(64), 0, (hipStream_t)stream, p, (const float*)part, nqc);
    }
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
