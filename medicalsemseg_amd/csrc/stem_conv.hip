// Stem convolution: conv3d 3x3x3 (stride 1, pad 1) with ONE input channel, bf16, and its weight gradient.
//
// BasicUNet's first layer (conv_0.conv_0: 1 -> 32 channels at full resolution; MONAI TwoConv -- BASELINE.json configs 1-3,
// SURVEY.md row A15; MONAI is not vendored by the reference) and the first conv of Swin-UNETR's encoder1 UnetResBlock
// (/root/reference/models/segmentors/swin_unetr.py:73-81: 1 -> 48) have K = 27: it writes 113 MB per 96^3 batch-2 step for 1.5 GFLOP,
// i.e. it is bound by its output stream, and its weight gradient by reading the output gradient once.  On the
// generic gather path every operand element was a separate global load (224 us + 98 us); here the one-channel halo
// lives in LDS and the taps are gathered from it.
//
// forward : D[cout][voxel] = W[cout][tap] * P[tap][voxel], one v_mfma_f32_16x16x32_bf16 per (16 couts, 16 voxels); the
//           B operand lane (voxel, tap group g) gathers its 8 taps with ds_read_u16 from the halo image; the packed
//           weight image of msseg_pack_weights (K = 27) is the A operand as it stands.  Fused bias / bf16 / InstanceNorm
//           statistics epilogue as in conv3d_k3_pp.hip.
// wgrad   : dW[cout][tap] = sum_v dy[v][cout] * x[v + tap], one v_mfma_f32_32x32x16_bf16 per 16-voxel row (all 32 couts
//           x 27 taps); dy tile filled by LDS-DMA and read with the transposing LDS read, the patch operand
//           (8 consecutive voxels of one tap per lane) is ONE ds_read_b128 from one of three copies of the halo
//           shifted by kw = 0, 1, 2 (so that every read is 16-byte aligned).
#include "k3pp.h"

namespace {

constexpr int TD = 4, TH = 4, TW = 16;
constexpr int PD = TD + 2, PH = TH + 2, PW = TW + 2;
constexpr int HV = PD * PH * PW;   // 648

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct TileCo { int n, d0, h0, w0; };

struct Sched {
    int tiles_w, tiles_h, tiles_d, t_first, t_step, n_my;
    MSSEG_DEVFN void init(int N, int D, int H, int W) {
        tiles_w = (W + TW - 1) / TW; tiles_h = (H + TH - 1) / TH; tiles_d = (D + TD - 1) / TD;
        const int ntiles = N * tiles_d * tiles_h * tiles_w;
        int t_end;
        if ((gridDim.x & 7) == 0) {   // XCD-contiguous walk (conv3d_k3_pp.hip)
            const int chunk = (ntiles + 7) >> 3, xcd = blockIdx.x & 7;
            t_first = xcd * chunk + (blockIdx.x >> 3);
            t_step = gridDim.x >> 3;
            t_end = min(ntiles, (xcd + 1) * chunk);
        } else {
            t_first = blockIdx.x; t_step = gridDim.x; t_end = ntiles;
        }
        n_my = t_first < t_end ? (t_end - t_first + t_step - 1) / t_step : 0;
    }
    MSSEG_DEVFN TileCo tile(int k) const {
        int t = t_first + k * t_step;
        TileCo tc;
        tc.w0 = (t % tiles_w) * TW; t /= tiles_w;
        tc.h0 = (t % tiles_h) * TH; t /= tiles_h;
        tc.d0 = (t % tiles_d) * TD; t /= tiles_d;
        tc.n = t;
        return tc;
    }
};

// ------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------
constexpr int F_THREADS = 512;
// JT = 16-wide cout tiles per cout block: 2 (32-wide blocks, BasicUNet's 1 -> 32 stem) or 3 (48-wide: the 1 -> 48 first conv of
// Swin-UNETR's encoder1 UnetResBlock, /root/reference/models/segmentors/swin_unetr.py:73-81)
// NORM: the output is normalised with given statistics and activated before it is stored (no raw output, no statistics) -- with
// a statistics-only launch in front (y == nullptr) this is conv + InstanceNorm + LeakyReLU of an inference forward without the
// raw tensor's write and the normalisation pass's read + write: the one-channel conv is cheap enough to run twice
template <int STATS, int JT, int NORM = 0>
__global__ __launch_bounds__(F_THREADS) void stem_fwd_kernel(const StemParams p) {
    constexpr int CBW = JT * 16;
    constexpr int F_STAT_FLOATS = 8 * MSSEG_STATS_NMAX * CBW * 2;
    __shared__ __attribute__((aligned(16))) unsigned short halo[2][HV + 8];
    __shared__ float ldsS[STATS ? F_STAT_FLOATS : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int coutblk = blockIdx.y;
    const unsigned short* __restrict__ xg = (const unsigned short*)p.x;
    bf16_t* __restrict__ yg = (bf16_t*)p.y;
    Sched sc;
    sc.init(p.N, p.D, p.H, p.W);

    // weights: A operand fragments straight from the packed image [q][cout 32][16 B] of this cout block
    u32x4_t wf[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
        wf[jt] = *(const u32x4_t*)((const unsigned char*)p.wp + (long long)coutblk * (4 * CBW * 16) + (q * CBW + jt * 16 + r) * 16);
    f32x4_t bv[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        bv[jt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv[jt] = *(const f32x4_t*)(p.bias + coutblk * CBW + jt * 16 + q * 4);
    }
    // taps 8q .. 8q+7 of this lane's k group: element offsets in the halo image (tap >= 27: masked)
    const int ntaps = p.taps == 1 ? 1 : 27;   // 1: the 1x1x1 conv (UnetResBlock's conv3 on the one-channel input): centre voxel only
    int toff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = q * 8 + j;
        toff[j] = ntaps == 1 ? (PH + 1) * PW + 1 : ((k < 27) ? ((k / 9) * PH + ((k / 3) % 3)) * PW + (k % 3) : 0);   // k >= taps: any valid element
    }
    if (STATS) {
        for (int i = tid; i < F_STAT_FLOATS; i += F_THREADS) ldsS[i] = 0.f;
    }
    float nsc[NORM ? JT : 1][4], nsh[NORM ? JT : 1][4];
    int n_n = -1;
    auto load_norm = [&](int n) {   // scale / shift of sample n: instnorm_kernel's arithmetic (elementwise.hip mean_rstd)
        if constexpr (NORM != 0) {
            const float inv = 1.0f / (float)((long long)p.D * p.H * p.W);
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = coutblk * CBW + jt * 16 + q * 4 + e;
                    const float su = p.nstats[((long long)n * p.M + c) * 2 + 0], su2 = p.nstats[((long long)n * p.M + c) * 2 + 1];
                    const float mean = su * inv;
                    float var = su2 * inv - mean * mean;
                    var = var > 0.f ? var : 0.f;
                    const float rstd = rsqrtf(var + p.eps);
                    nsc[jt][e] = rstd * (p.gamma ? p.gamma[c] : 1.f);
                    nsh[jt][e] = (p.beta ? p.beta[c] : 0.f) - mean * nsc[jt][e];
                }
        }
    };
    float s1[JT][4], s2[JT][4];
    int s_n = -1;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[jt][e] = s2[jt][e] = 0.f;
    auto flush_stats = [&]() {
        if (s_n < 0) return;
        float* slot = ldsS + ((wave * MSSEG_STATS_NMAX + s_n) * CBW) * 2;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = s1[jt][e], b = s2[jt][e];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    a += __shfl_xor(a, o);
                    b += __shfl_xor(b, o);
                }
                if (r == 0) {
                    float* sp = slot + (jt * 16 + q * 4 + e) * 2;
                    sp[0] += a;
                    sp[1] += b;
                }
                s1[jt][e] = s2[jt][e] = 0.f;
            }
    };

    // halo staging: two elements per thread, register-staged one tile ahead
    unsigned short st[2];
    auto fetch = [&](const TileCo& tc) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int hv = tid + it * F_THREADS;
            const int hd = hv / (PH * PW), rem = hv - hd * (PH * PW), hh = rem / PW, hw = rem - hh * PW;
            const int d = tc.d0 - 1 + hd, h = tc.h0 - 1 + hh, w = tc.w0 - 1 + hw;
            const bool inb = hv < HV && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            st[it] = inb ? xg[((((long long)tc.n * p.D + d) * p.H + h) * p.W + w) * p.ldx] : (unsigned short)0;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int hv = tid + it * F_THREADS;
            if (hv < HV) halo[buf][hv] = st[it];
        }
    };

    if (sc.n_my > 0) {
        fetch(sc.tile(0));
        commit(0);
    }
    __syncthreads();
    const int dw = wave >> 1, h2 = (wave & 1) * 2;   // this wave: depth slice dw, rows h2, h2 + 1
    for (int k = 0; k < sc.n_my; ++k) {
        const TileCo tc = sc.tile(k);
        if (k + 1 < sc.n_my) fetch(sc.tile(k + 1));
        const unsigned short* hb = halo[k & 1];
        if constexpr (STATS != 0) {
            if (tc.n != s_n) { flush_stats(); s_n = tc.n; }
        }
        if constexpr (NORM != 0) {
            if (tc.n != n_n) { load_norm(tc.n); n_n = tc.n; }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int hrow = h2 + m;
            const int base = (dw * PH + hrow) * PW + r;
            unsigned short v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = hb[base + toff[j]];
            // k-slots past the last tap read a valid halo element (toff = that of tap 0) against a ZERO weight: the packed image
            // is zero-filled beyond K, so no masking of the operand is needed (a non-finite input there already reaches this
            // voxel through tap 0)
            u32x4_t xf;
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[j] = (unsigned)v[2 * j] | ((unsigned)v[2 * j + 1] << 16);
            const int d = tc.d0 + dw, h = tc.h0 + hrow, w = tc.w0 + r;
            const bool ok = d < p.D && h < p.H && w < p.W;
            const long long vox = (((long long)tc.n * p.D + d) * p.H + h) * p.W + w;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                mma_chunk<bf16_t>(acc, wf[jt], xf);
                const f32x4_t o = acc + bv[jt];
                const bf16x4_t ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                if constexpr (NORM != 0) {   // from the value as the unfused chain stores it (bf16), in that chain's arithmetic
                    bf16x4_t oa;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z = (float)ob[e] * nsc[jt][e] + nsh[jt][e];
                        oa[e] = (bf16_t)(z > 0.f ? z : z * p.slope);
                    }
                    if (ok) *(bf16x4_t*)(yg + vox * p.ldy + coutblk * CBW + jt * 16 + q * 4) = oa;
                } else {
                    if (ok && yg != nullptr) *(bf16x4_t*)(yg + vox * p.ldy + coutblk * CBW + jt * 16 + q * 4) = ob;
                }
                if constexpr (STATS != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float rv = ok ? (float)ob[e] : 0.f;
                        s1[jt][e] += rv;
                        s2[jt][e] += rv * rv;
                    }
                }
            }
        }
        if (k + 1 < sc.n_my) commit((k + 1) & 1);
        __syncthreads();
    }
    if constexpr (STATS != 0) {
        flush_stats();
        __syncthreads();
        const int PN = p.N * CBW * 2;
        float* wsp = p.stats_ws + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * PN;
        for (int i = tid; i < PN; i += F_THREADS) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) s += ldsS[wv * MSSEG_STATS_NMAX * CBW * 2 + i];
            wsp[i] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------------------
constexpr int W_THREADS = 256;
constexpr int XP = 24;                              // row pitch (elements) of the shifted halo copies: 16-byte aligned rows
constexpr int XS_ELEMS = PD * PH * XP;              // one shifted copy
constexpr int P_BYTES = TD * TH * TW * 64;          // dy tile, 32 channels

__device__ u32x4_t g_stem_zero_chunk;

MSSEG_DEVFN void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
MSSEG_DEVFN bf16x4_t lds_tr(lds_u8* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)p);
}

__global__ __launch_bounds__(W_THREADS) void stem_wgrad_kernel(const StemWgParams p) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    // [2 buffers] x { dy tile 16 KB | 3 shifted x copies }
    constexpr int BUF_BYTES = P_BYTES + 3 * XS_ELEMS * 2 + 64;
    lds_u8* smem3 = (lds_u8*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mblk = blockIdx.y;
    const unsigned char* dyg = (const unsigned char*)p.dy + mblk * 64;
    const unsigned short* __restrict__ xg = (const unsigned short*)p.x;
    Sched sc;
    sc.init(p.N, p.D, p.H, p.W);

    // dy tile DMA: 16 wave-instructions of 16 rows x 64 B; wave w issues instructions w, w+4, ...
    unsigned p_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tv = (wave + 4 * j) * 16 + (lane >> 2);
        const int td = tv / (TH * TW), th = (tv / TW) % TH, tw = tv % TW;
        p_off[j] = (unsigned)((((long long)td * p.H + th) * p.W + tw) * p.lddy * 2 + (lane & 3) * 16);
    }
    unsigned short st[3];
    auto load_tile = [&](const TileCo& tc, int buf) {
        unsigned char* base = smem + buf * BUF_BYTES;
        const long long pvox = (((long long)tc.n * p.D + tc.d0) * p.H + tc.h0) * p.W + tc.w0;
        const unsigned char* pb = dyg + pvox * p.lddy * 2;
        const bool full = tc.d0 + TD <= p.D && tc.h0 + TH <= p.H && tc.w0 + TW <= p.W;
        const unsigned char* zsrc = (const unsigned char*)&g_stem_zero_chunk;
        const bool ch_ok = mblk * 32 + (lane & 3) * 8 < p.M;   // 48 couts: the second block's upper half is padding
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned char* src = pb + p_off[j];
            if (!ch_ok) src = zsrc;
            if (!full) {
                const int tv = (wave + 4 * j) * 16 + (lane >> 2);
                const int td = tv / (TH * TW), th = (tv / TW) % TH, tw = tv % TW;
                if (!(tc.d0 + td < p.D && tc.h0 + th < p.H && tc.w0 + tw < p.W)) src = zsrc;
            }
            glds16(src, base + (wave + 4 * j) * 1024);
        }
        // x halo -> registers (written to the three shifted copies by commit_x)
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int hv = tid + it * W_THREADS;
            const int hd = hv / (PH * PW), rem = hv - hd * (PH * PW), hh = rem / PW, hw = rem - hh * PW;
            const int d = tc.d0 - 1 + hd, h = tc.h0 - 1 + hh, w = tc.w0 - 1 + hw;
            const bool inb = hv < HV && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            st[it] = inb ? xg[((((long long)tc.n * p.D + d) * p.H + h) * p.W + w) * p.ldx] : (unsigned short)0;
        }
    };
    auto commit_x = [&](int buf) {
        unsigned short* xs = (unsigned short*)(smem + buf * BUF_BYTES + P_BYTES);
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int hv = tid + it * W_THREADS;
            if (hv < HV) {
                const int hd = hv / (PH * PW), rem = hv - hd * (PH * PW), hh = rem / PW, hw = rem - hh * PW;
#pragma unroll
                for (int c = 0; c < 3; ++c) {   // copy c holds x[.., hw] at slot hw - c
                    const int slot = hw - c;
                    if (slot >= 0 && slot < TW) xs[c * XS_ELEMS + (hd * PH + hh) * XP + slot] = st[it];
                }
            }
        }
    };

    // MFMA operands (v_mfma_f32_32x32x16_bf16): A = dy^T: lane l holds voxels 8*(l>>5).. of cout l & 31 (transposing
    // reads as in conv3d_k3_wgrad_pp.hip); B: lane l holds voxels 8*(l>>5).. of tap l & 31 = one 16-byte read.
    const int G = lane >> 4, qr = (lane >> 2) & 3, pc = lane & 3;
    unsigned a_rd[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_rd[i] = (8 * (G >> 1) + 4 * i + qr) * 64 + ((G & 1) << 5) + pc * 8;
    const int tap = lane & 31;
    const bool tap_ok = tap < 27;
    const int tkd = tap / 9, tkh = (tap / 3) % 3, tkw = tap % 3;
    const unsigned b_rd = tap_ok ? (unsigned)(P_BYTES + (tkw * XS_ELEMS + (tkd * PH + tkh) * XP + 8 * (lane >> 5)) * 2) : 0u;

    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    if (sc.n_my > 0) {
        load_tile(sc.tile(0), 0);
        commit_x(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < sc.n_my; ++k) {
        const int buf = k & 1;
        if (k + 1 < sc.n_my) load_tile(sc.tile(k + 1), buf ^ 1);
        lds_u8* bb = smem3 + buf * BUF_BYTES;
        // 16 tile rows, 4 per wave: row = (td, th)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const int row = wave * 4 + i4;
            const int td = row >> 2, th = row & 3;
            const bf16x4_t lo = lds_tr(bb + a_rd[0] + row * 1024), hi = lds_tr(bb + a_rd[1] + row * 1024);
            const bf16x8_t af = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            u32x4_t bfr = *(const u32x4_t*)(smem + buf * BUF_BYTES + b_rd + ((td * PH + th) * XP) * 2);
            if (!tap_ok) bfr = u32x4_t{0u, 0u, 0u, 0u};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8_t, bfr), acc, 0, 0, 0);
        }
        if (k + 1 < sc.n_my) commit_x(buf ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // ---- 4 waves -> one slab [cout 32][tap 32] through LDS (fixed order)
    float* xch = (float*)smem;
    const int col = lane & 31, rb = 4 * (lane >> 5);
#pragma unroll
    for (int e = 0; e < 16; ++e) xch[(wave * 32 + (e & 3) + 8 * (e >> 2) + rb) * 32 + col] = acc[e];
    __syncthreads();
    float* slab = p.slabs + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 1024;
    for (int i = tid; i < 1024; i += W_THREADS) slab[i] = xch[i] + xch[1024 + i] + xch[2048 + i] + xch[3072 + i];
}

}  // namespace

bool msseg_stem_eligible(int dtype, int Cin, int Cout, int k, int s, int pd, long long ldx, long long ldy, const void* y) {
    static const bool off = getenv("MSSEG_NO_STEM") != nullptr;
    return !off && dtype == MSSEG_BF16 && Cin == 1 && ((k == 3 && pd == 1) || (k == 1 && pd == 0)) && s == 1 &&
           (Cout % 32 == 0 || Cout % 48 == 0) && Cout <= 256 &&
           ldx >= 1 && (ldy % 4) == 0 && (((uintptr_t)y) & 7) == 0;
}

int msseg_stem_fwd_launch(const StemParams& p, hipStream_t stream) {
    const int cbw = (p.M % 32 == 0) ? 32 : 48;   // = msseg_cout_block(M): the block width of the packed image
    const int ncb = p.M / cbw;
    const int tiles = p.N * ceil_div(p.D, TD) * ceil_div(p.H, TH) * ceil_div(p.W, TW);
    int gx = msseg_num_cus() * 2 / ncb;
    gx &= ~7;
    if (gx < 8) gx = 8;
    if (gx > tiles) gx = tiles;
    if (p.nstats != nullptr && p.stats != nullptr) MSSEG_FAIL(MSSEG_EINVAL, "stem_fwd: the normalised form emits no statistics");
    if (p.y == nullptr && p.stats == nullptr) MSSEG_FAIL(MSSEG_EINVAL, "stem_fwd: nothing to produce");
    if (cbw == 32) {
        if (p.nstats) hipLaunchKernelGGL((stem_fwd_kernel<0, 2, 1>), dim3(gx, ncb), dim3(F_THREADS), 0, stream, p);
        else if (p.stats) hipLaunchKernelGGL((stem_fwd_kernel<1, 2>), dim3(gx, ncb), dim3(F_THREADS), 0, stream, p);
        else hipLaunchKernelGGL((stem_fwd_kernel<0, 2>), dim3(gx, ncb), dim3(F_THREADS), 0, stream, p);
    } else {
        if (p.nstats) hipLaunchKernelGGL((stem_fwd_kernel<0, 3, 1>), dim3(gx, ncb), dim3(F_THREADS), 0, stream, p);
        else if (p.stats) hipLaunchKernelGGL((stem_fwd_kernel<1, 3>), dim3(gx, ncb), dim3(F_THREADS), 0, stream, p);
        else hipLaunchKernelGGL((stem_fwd_kernel<0, 3>), dim3(gx, ncb), dim3(F_THREADS), 0, stream, p);
    }
    MSSEG_CHECK_LAUNCH("stem_fwd");
    if (p.stats) {
        K3FinParams f{};
        f.ws = p.stats_ws; f.R = gx; f.N = p.N; f.coutb = cbw; f.M = p.M; f.stats = p.stats;
        return msseg_k3_stats_finalize(f, ncb, stream);
    }
    return MSSEG_OK;
}

int msseg_stem_wgrad_grid(const StemWgParams& p) {
    const int tiles = p.N * ceil_div(p.D, TD) * ceil_div(p.H, TH) * ceil_div(p.W, TW);
    int gx = msseg_num_cus() * 2 / ceil_div(p.M, 32);
    gx &= ~7;
    if (gx < 8) gx = 8;
    if (gx > tiles) gx = tiles;
    return gx;
}

int msseg_stem_wgrad_launch(const StemWgParams& p, int gx, hipStream_t stream) {
    const int lds = 2 * (P_BYTES + 3 * XS_ELEMS * 2 + 64);
    static msseg_lds_attr_once attr;
    if (!attr.ensure((const void*)stem_wgrad_kernel, lds)) MSSEG_FAIL(MSSEG_ELAUNCH, "stem_wgrad: cannot set dynamic LDS size %d", lds);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(gx, ceil_div(p.M, 32)), dim3(W_THREADS), lds, stream, p);
    MSSEG_CHECK_LAUNCH("stem_wgrad");
    return MSSEG_OK;
}
