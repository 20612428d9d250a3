// Implicit-GEMM forward-shaped convolutions on MFMA (gfx950).
//
//   D[cout][voxel] = sum_{tap, c} W[cout][tap][c] * X[voxel + tap][c]
//
// One persistent workgroup walks output tiles.  Per tile and per block of CB = 4 chunks of input
// channels (32 bf16 / 16 f32) it stages
//   A: the input halo tile, channel-chunk-planar  [quarter q][halo voxel][16 B]
//   B: the packed weights for that channel block  [tap][q][cout][16 B]   (resident across tiles when
//      the layer has a single channel block)
// in LDS, then every wave runs NTAPS x MT x NT MFMA k-groups with both operands read as ds_read_b128.
// The planar A image keeps the 16 voxel rows of an MFMA operand on 16 consecutive 16-byte slots for
// every tap shift, and plane strides are multiples of 256 B, so the reads are bank-conflict free.
// Weights are the MFMA "A" operand (rows = cout) so that each lane ends up with 4 consecutive output
// channels of one voxel and the epilogue stores 8 B (bf16) / 16 B (f32) per lane.
//
// Source modes (how the A tile is filled) and epilogues cover: conv k3 s1 p1 (+ its input gradient with
// flipped weights), conv k1, few-channel convs gathered im2col-style (stem conv, patch embedding),
// ConvTranspose k2 s2 forward (1x1 GEMM + pixel-shuffle scatter) and its input gradient (gather).
#include "common.h"

#include <stdlib.h>

namespace {

enum { SRC_DIRECT = 0, SRC_GATHER = 1, SRC_DECONV_BWD = 2 };
enum { EPI_STORE = 0, EPI_DECONV = 1 };

struct IgemmParams {
    const void* x;
    long long ldx;
    const void* wp;
    const float* bias;
    void* y;
    long long ldy;
    int N, D, H, W;      // tiled output grid (flat mode: N = D = H = 1, W = number of output voxels)
    int ID, IH, IW;      // source spatial dims (gather / deconv modes)
    int OD, OH, OW;      // output spatial dims of one sample (flat modes that need coordinates)
    int K, M, NKB;       // logical input channels, logical output channels, channel blocks
    int tiles_d, tiles_h, tiles_w;
    int ntiles;
    int cin, k, s, p;    // gather: real Cin, kernel, stride, pad
    int creal;           // deconv modes: real channel count of the fine tensor
    int vec_store;       // y / ldy allow 4-element vector stores
    int rel32_ok;        // halo-relative element offsets fit 32 bits (precomputed-offset staging path)
    int cout_block;      // 0 = msseg_cout_block(M)
    float* stats;        // optional fused per-(n, cout) (sum, sum of squares) of the stored output
    float* stats_ws;     // scratch partials
    unsigned int* counter;
    // optional fused InstanceNorm-BACKWARD reductions: this launch produces da (gradient w.r.t. the activation
    // a = lrelu(IN(yraw))); the epilogue then accumulates red[n][c] = (sum dz, sum dz*xhat), dz = da*lrelu'(a), into
    // `stats`, and the finalising block also emits dbeta / dgamma.
    const void* nb_y; long long nb_ldy;
    const void* nb_a; long long nb_lda;
    const float* nb_stats;   // [N][M][2] forward statistics (sum, sum of squares) of yraw
    float nb_slope, nb_eps; long long nb_S;
    float* nb_dgamma; float* nb_dbeta; int nb_acc;
};

template <typename T, int NTAPS, int SRC, int EPI, int TD, int TH, int TW, int WAVES, int NT, int STRIDE = 1, int NSL = 1>
struct IgemmCfg {
    static constexpr int EPC = DT<T>::EPC;
    static constexpr int CB = 4 * EPC;
    static constexpr int PAD = (NTAPS == 27) ? 1 : 0;
    // input halo of an output tile: (T-1)*STRIDE + kernel extent
    static constexpr int PD = (TD - 1) * STRIDE + 1 + 2 * PAD, PH = (TH - 1) * STRIDE + 1 + 2 * PAD,
                         PW = (TW - 1) * STRIDE + 1 + 2 * PAD;
    static constexpr int HV = PD * PH * PW;
    static constexpr int TV = TD * TH * TW;
    static constexpr int MT = TV / 16 / WAVES;
    static constexpr int NTHREADS = WAVES * 64;
    static constexpr int COUTB = NT * 16;
    static constexpr int PLANE = ((HV * 16 + 255) / 256) * 256;
    static constexpr int A_BYTES = ((4 * PLANE + 64 + 255) / 256) * 256;
    // NSL > 1: the weight image is staged in NSL tap slices (one kd plane each) instead of all 27 taps, which
    // brings the workgroup under 80 KB of LDS so that two workgroups share a CU and fill each other's bubbles
    static constexpr int BT = NTAPS / NSL;
    static constexpr int B_BYTES = BT * 4 * COUTB * 16;
    static_assert(NTAPS % NSL == 0, "tap slices must divide the taps");
    static constexpr int STAT_FLOATS = MSSEG_STATS_NMAX * COUTB * 2;
    static constexpr int STAT_BYTES = (EPI == EPI_STORE) ? (STAT_FLOATS + WAVES * COUTB * 2) * 4 + 256 : 0;
    static constexpr int LDS_BYTES = A_BYTES + B_BYTES + STAT_BYTES;
    static_assert(TV % (16 * WAVES) == 0, "tile must split into 16-voxel MFMA tiles per wave");
};

MSSEG_DEVFN int aoff(int q, int plane) { return q * plane + (q >> 1) * 32; }

template <typename T, int NTAPS, int SRC, int EPI, int TD, int TH, int TW, int WAVES, int NT, int STRIDE = 1, int NSL = 1>
__global__ __launch_bounds__(WAVES * 64, (NSL > 1 ? 2 : 1)) void igemm_fwd_kernel(const IgemmParams p) {
    using C = IgemmCfg<T, NTAPS, SRC, EPI, TD, TH, TW, WAVES, NT, STRIDE, NSL>;
    constexpr int BT = C::BT;
    constexpr int EPC = C::EPC, CB = C::CB, PAD = C::PAD, PH = C::PH, PW = C::PW, HV = C::HV, MT = C::MT;
    constexpr int NTHREADS = C::NTHREADS, COUTB = C::COUTB, PLANE = C::PLANE;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    unsigned char* ldsA = smem;
    unsigned char* ldsB = smem + C::A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int coutblk = blockIdx.y;
    const T* __restrict__ xg = (const T*)p.x;
    T* __restrict__ yg = (T*)p.y;

    // per-lane LDS offsets of the voxel rows this lane feeds into the MFMA B operand
    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int v = (wave * MT + m) * 16 + r;
        const int td = v / (TH * TW), th = (v / TW) % TH, tw = v % TW;
        abase[m] = aoff(q, PLANE) + ((td * STRIDE * PH + th * STRIDE) * PW + tw * STRIDE) * 16;
    }
    const int bbase = (q * COUTB + r) * 16;
    float* spart = (float*)(smem + C::A_BYTES + C::B_BYTES);
    const bool do_stats = (EPI == EPI_STORE) && p.stats != nullptr;
    float* wpart = spart + C::STAT_FLOATS;  // [WAVES][COUTB][2] staging for the per-sample flush
    if (do_stats) {
        for (int i = tid; i < C::STAT_FLOATS; i += NTHREADS) spart[i] = 0.f;
    }
    float st[NT][4], st2[NT][4];  // this lane's running (sum, sum of squares) of the current sample
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) st[j][e] = st2[j][e] = 0.f;
    int cur_n = -1;
    // Flush = fixed-order reduction lanes -> wave -> workgroup, so the statistics are bit-reproducible.
    auto flush_stats = [&](int nn) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = st[j][e], b = st2[j][e];
#pragma unroll
                for (int o2 = 1; o2 < 16; o2 <<= 1) {
                    a += __shfl_xor(a, o2);
                    b += __shfl_xor(b, o2);
                }
                if (r == 0) {
                    const int cl = j * 16 + q * 4 + e;
                    wpart[(wave * COUTB + cl) * 2 + 0] = a;
                    wpart[(wave * COUTB + cl) * 2 + 1] = b;
                }
                st[j][e] = st2[j][e] = 0.f;
            }
        __syncthreads();
        if (tid < COUTB * 2) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < WAVES; ++wv) s += wpart[wv * COUTB * 2 + tid];
            spart[nn * COUTB * 2 + tid] += s;
        }
        __syncthreads();
    };

    // ---------------------------------------------------------------------------------------------
    // Software pipeline over stages (tile, channel block): the global loads of stage s+1 are issued right
    // after stage s's LDS image is complete and stay in flight under stage s's MFMAs; they are written to
    // LDS after the barrier that ends stage s (issue-early / write-late register staging).
    // ---------------------------------------------------------------------------------------------
    constexpr int NIT_A = (HV * 4 + NTHREADS - 1) / NTHREADS;
    constexpr int NIT_B = (C::B_BYTES / 16 + NTHREADS - 1) / NTHREADS;
    struct TileCo { int n, d0, h0, w0; };
    auto decode = [&](int tile) {
        TileCo tc;
        int t = tile;
        tc.w0 = (t % p.tiles_w) * TW; t /= p.tiles_w;
        tc.h0 = (t % p.tiles_h) * TH; t /= p.tiles_h;
        tc.d0 = (t % p.tiles_d) * TD; t /= p.tiles_d;
        tc.n = t;
        return tc;
    };
    auto load_a_chunk = [&](const TileCo& tc, int kb, int i) -> u32x4_t {
        const int cq = i & 3, hv = i >> 2;
        const int c = kb * CB + cq * EPC;
        u32x4_t val = {0u, 0u, 0u, 0u};
        if constexpr (SRC == SRC_DIRECT) {
            const int hw = hv % PW, t2 = hv / PW, hh = t2 % PH, hd = t2 / PH;
            const int d = tc.d0 * STRIDE - PAD + hd, h = tc.h0 * STRIDE - PAD + hh, w = tc.w0 * STRIDE - PAD + hw;
            const int XD = STRIDE == 1 ? p.D : p.ID, XH = STRIDE == 1 ? p.H : p.IH, XW = STRIDE == 1 ? p.W : p.IW;
            if (c < p.K && (unsigned)d < (unsigned)XD && (unsigned)h < (unsigned)XH && (unsigned)w < (unsigned)XW) {
                const long long vox = (((long long)tc.n * XD + d) * XH + h) * XW + w;
                val = *(const u32x4_t*)(xg + vox * p.ldx + c);
            }
        } else if constexpr (SRC == SRC_GATHER) {
            const int ov = tc.w0 + hv;  // flat output voxel
            if (ov < p.W && c < p.K) {
                int tt = ov;
                const int ow = tt % p.OW; tt /= p.OW;
                const int oh = tt % p.OH; tt /= p.OH;
                const int od = tt % p.OD; const int nn = tt / p.OD;
                alignas(16) T tmp[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const int vc = c + e;
                    float fv = 0.f;
                    if (vc < p.K) {
                        const int tap = vc / p.cin, ci = vc - tap * p.cin;
                        const int kw = tap % p.k, kh = (tap / p.k) % p.k, kd = tap / (p.k * p.k);
                        const int id = od * p.s - p.p + kd, ih = oh * p.s - p.p + kh, iw = ow * p.s - p.p + kw;
                        if ((unsigned)id < (unsigned)p.ID && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW) {
                            const long long vox = (((long long)nn * p.ID + id) * p.IH + ih) * p.IW + iw;
                            fv = DT<T>::ld(xg + vox * p.ldx + ci);
                        }
                    }
                    DT<T>::st(&tmp[e], fv);
                }
                val = *(const u32x4_t*)tmp;
            }
        } else {  // SRC_DECONV_BWD: coarse voxel gathers its 8 fine children
            const int cv = tc.w0 + hv;
            if (cv < p.W && c < p.K) {
                int tt = cv;
                const int cw = tt % p.OW; tt /= p.OW;
                const int ch = tt % p.OH; tt /= p.OH;
                const int cd = tt % p.OD; const int nn = tt / p.OD;
                const int abc = c / p.creal, co = c - abc * p.creal;
                const int fd = 2 * cd + (abc >> 2), fh = 2 * ch + ((abc >> 1) & 1), fw = 2 * cw + (abc & 1);
                const long long vox = (((long long)nn * (2 * p.OD) + fd) * (2 * p.OH) + fh) * (2 * p.OW) + fw;
                val = *(const u32x4_t*)(xg + vox * p.ldx + co);
            }
        }
        return val;
    };
    u32x4_t pa[NIT_A], pb[NIT_B];
    // DIRECT source: the (halo voxel, chunk) a thread stages is the same for every tile, so its element offset
    // relative to the tile origin and its halo coordinates are computed once (VALU work per tile drops from ~40 to
    // ~3 instructions per chunk; interior tiles skip the bounds checks altogether).
    int a_rel[NIT_A], a_pk[NIT_A];
    const int XD = STRIDE == 1 ? p.D : p.ID, XH = STRIDE == 1 ? p.H : p.IH, XW = STRIDE == 1 ? p.W : p.IW;
    if constexpr (SRC == SRC_DIRECT) {
#pragma unroll
        for (int it = 0; it < NIT_A; ++it) {
            const int i = tid + it * NTHREADS;
            const int cq = i & 3, hv = i >> 2;
            const int hw = hv % PW, t2 = hv / PW, hh = t2 % PH, hd = t2 / PH;
            a_rel[it] = (int)((((long long)hd * XH + hh) * XW + hw) * p.ldx) + cq * EPC;
            a_pk[it] = (i < HV * 4) ? (hd | (hh << 8) | (hw << 16)) : -1;
        }
    }
    auto fetch = [&](const TileCo& tc, int kb, int ks, bool with_a, bool with_b) {
        if (with_a) {
        if constexpr (SRC == SRC_DIRECT) {
            if (p.rel32_ok) {
                const int dB = tc.d0 * STRIDE - PAD, hB = tc.h0 * STRIDE - PAD, wB = tc.w0 * STRIDE - PAD;
                const long long basev = (((long long)tc.n * XD + dB) * XH + hB) * XW + wB;
                const T* bp = xg + basev * p.ldx + kb * CB;
                const bool interior = dB >= 0 && dB + C::PD <= XD && hB >= 0 && hB + PH <= XH && wB >= 0 && wB + PW <= XW;
                const int nchunk = (p.K - kb * CB + EPC - 1) / EPC;   // valid 16-byte chunks of this channel block
                const bool cok = (tid & 3) < nchunk;
                if (interior) {
#pragma unroll
                    for (int it = 0; it < NIT_A; ++it) {
                        const bool valid = (NSL > 1) ? (tid + it * NTHREADS < HV * 4) : (a_pk[it] >= 0);
                        pa[it] = (cok && valid) ? *(const u32x4_t*)(bp + a_rel[it]) : u32x4_t{0u, 0u, 0u, 0u};
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < NIT_A; ++it) {
                        int pk = a_pk[it];
                        if constexpr (NSL > 1) {   // register-lean variant: recompute the halo coordinates
                            const int i = tid + it * NTHREADS;
                            const int hv = i >> 2;
                            const int hw = hv % PW, t2 = hv / PW;
                            pk = (i < HV * 4) ? ((t2 / PH) | ((t2 % PH) << 8) | (hw << 16)) : -1;
                        }
                        const int d = dB + (pk & 255), h = hB + ((pk >> 8) & 255), w = wB + ((pk >> 16) & 255);
                        const bool ok = cok && pk >= 0 && (unsigned)d < (unsigned)XD && (unsigned)h < (unsigned)XH &&
                                        (unsigned)w < (unsigned)XW;
                        pa[it] = ok ? *(const u32x4_t*)(bp + a_rel[it]) : u32x4_t{0u, 0u, 0u, 0u};
                    }
                }
            } else {
#pragma unroll
                for (int it = 0; it < NIT_A; ++it) {
                    const int i = tid + it * NTHREADS;
                    pa[it] = (i < HV * 4) ? load_a_chunk(tc, kb, i) : u32x4_t{0u, 0u, 0u, 0u};
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < NIT_A; ++it) {
                const int i = tid + it * NTHREADS;
                pa[it] = (i < HV * 4) ? load_a_chunk(tc, kb, i) : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
        }
        if (with_b) {
            const u32x4_t* src = (const u32x4_t*)((const unsigned char*)p.wp +
                                                  (((long long)coutblk * p.NKB + kb) * NSL + ks) * C::B_BYTES);
#pragma unroll
            for (int it = 0; it < NIT_B; ++it) {
                const int i = tid + it * NTHREADS;
                pb[it] = (i < C::B_BYTES / 16) ? src[i] : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
    };
    auto commit = [&](bool with_a, bool with_b) {
        if (with_a) {
#pragma unroll
            for (int it = 0; it < NIT_A; ++it) {
                const int i = tid + it * NTHREADS;
                if (i < HV * 4) *(u32x4_t*)(ldsA + aoff(i & 3, PLANE) + (i >> 2) * 16) = pa[it];
            }
        }
        if (with_b) {
#pragma unroll
            for (int it = 0; it < NIT_B; ++it) {
                const int i = tid + it * NTHREADS;
                if (i < C::B_BYTES / 16) ((u32x4_t*)ldsB)[i] = pb[it];
            }
        }
    };

    f32x4_t acc[MT][NT];
    int tile = blockIdx.x, kb = 0, ks = 0;
    TileCo tc = decode(tile < p.ntiles ? tile : 0);
    bool a_pending = true, b_pending = true;
    if (tile < p.ntiles) fetch(tc, 0, 0, true, true);
    while (tile < p.ntiles) {
        __syncthreads();  // everyone finished reading the previous stage's LDS image
        commit(a_pending, b_pending);
        __syncthreads();
        // next stage: (tile, kb, ks) advance ks fastest
        int ntile = tile, nkb = kb, nks = ks + 1;
        if (nks == NSL) {
            nks = 0;
            nkb = kb + 1;
            if (nkb == p.NKB) { nkb = 0; ntile = tile + gridDim.x; }
        }
        const TileCo ntc = decode(ntile < p.ntiles ? ntile : 0);
        a_pending = nks == 0;
        b_pending = NSL > 1 || p.NKB > 1;
        if (ntile < p.ntiles) fetch(ntc, nkb, nks, a_pending, b_pending);

        const int n = tc.n, d0 = tc.d0, h0 = tc.h0, w0 = tc.w0;
        if (kb == 0 && ks == 0) {
            if constexpr (EPI == EPI_STORE) {
                if (do_stats && n != cur_n) {
                    if (cur_n >= 0) flush_stats(cur_n);
                    cur_n = n;
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[m][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        // ---------------- MFMA ----------------
        // Operand fragments are double-buffered in registers: the LDS reads of tap t+1 are issued before the
        // MFMAs of tap t, so the matrix pipe runs under one full LDS latency instead of waiting for it per pair.
        {
            u32x4_t af[2][MT], bf[2][NT];
            const int ksoff = (NSL > 1) ? ks * (PH * PW * 16) : 0;   // slice ks = kd plane ks of the halo
            auto load_frags = [&](int t, int buf) {
                const int tap = (NSL > 1) ? t : t;   // local tap within the slice; (kh, kw) from t when sliced by kd
                const int kd = (NSL > 1) ? 0 : tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                const int toff = (NTAPS == 27) ? ((kd * PH + kh) * PW + kw) * 16 : 0;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bf[buf][j] = *(const u32x4_t*)(ldsB + bbase + t * (4 * COUTB * 16) + j * 256);
#pragma unroll
                for (int m = 0; m < MT; ++m) af[buf][m] = *(const u32x4_t*)(ldsA + abase[m] + ksoff + toff);
            };
            if constexpr (NSL == 1) {
                load_frags(0, 0);
#pragma unroll
                for (int t = 0; t < BT; ++t) {
                    const int cur = t & 1;
                    if (t + 1 < BT) load_frags(t + 1, cur ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int j = 0; j < NT; ++j) mma_chunk<T>(acc[m][j], bf[cur][j], af[cur][m]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // two workgroups share the CU in this variant: the other workgroup's waves cover LDS latency, so the
                // fragments are single-buffered to stay within 256 VGPRs (2 waves per SIMD)
#pragma unroll
                for (int t = 0; t < BT; ++t) {
                    load_frags(t, 0);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int j = 0; j < NT; ++j) mma_chunk<T>(acc[m][j], bf[0][j], af[0][m]);
                }
            }
        }
        if (kb == p.NKB - 1 && ks == NSL - 1) {
            // ---------------- epilogue ----------------
    #pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int v = (wave * MT + m) * 16 + r;
                const int td = v / (TH * TW), th = (v / TW) % TH, tw = v % TW;
                const int d = d0 + td, h = h0 + th, w = w0 + tw;
                if (d >= p.D || h >= p.H || w >= p.W) continue;
                long long vox;
                int dn = 0, dd = 0, dh = 0, dw = 0;
                if constexpr (EPI == EPI_STORE) {
                    vox = (((long long)n * p.D + d) * p.H + h) * p.W + w;
                } else {
                    int tt = w;
                    dw = tt % p.OW; tt /= p.OW;
                    dh = tt % p.OH; tt /= p.OH;
                    dd = tt % p.OD; dn = tt / p.OD;
                    vox = 0;
                }
    #pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int co = coutblk * COUTB + j * 16 + q * 4;
                    if (co >= p.M) continue;
                    f32x4_t o = acc[m][j];
                    T* dst;
                    int cbase;
                    if constexpr (EPI == EPI_STORE) {
                        cbase = co;
                        dst = yg + vox * p.ldy + co;
                    } else {
                        const int abc = co / p.creal;
                        cbase = co - abc * p.creal;
                        const int fd = 2 * dd + (abc >> 2), fh = 2 * dh + ((abc >> 1) & 1), fw = 2 * dw + (abc & 1);
                        const long long fv = (((long long)dn * (2 * p.OD) + fd) * (2 * p.OH) + fh) * (2 * p.OW) + fw;
                        dst = yg + fv * p.ldy + cbase;
                    }
                    if (p.vec_store && co + 4 <= p.M) {
                        if (p.bias) {
    #pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += p.bias[cbase + e];
                        }
                        store4<T>(dst, o);
                    } else {
                        for (int e = 0; e < 4 && co + e < p.M; ++e) {
                            o[e] += (p.bias ? p.bias[cbase + e] : 0.f);
                            DT<T>::st(dst + e, o[e]);
                        }
                    }
                    if constexpr (EPI == EPI_STORE) {
                        if (do_stats) {
                            if (p.nb_y == nullptr) {
    #pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float rv = (float)(T)o[e];  // statistics of the tensor as stored
                                    st[j][e] += rv;
                                    st2[j][e] += rv * rv;
                                }
                            } else if (co + 4 <= p.M) {
                                // dz = da * lrelu'(a);  accumulate (sum dz, sum dz * yraw); xhat is formed at the end:
                                // sum dz*xhat = rstd * (sum dz*yraw - mean * sum dz)
                                f32x4_t yv, av;
                                if constexpr (sizeof(T) == 2) {
                                    const bf16x4_t y4 = *(const bf16x4_t*)((const T*)p.nb_y + vox * p.nb_ldy + co);
                                    const bf16x4_t a4 = *(const bf16x4_t*)((const T*)p.nb_a + vox * p.nb_lda + co);
    #pragma unroll
                                    for (int e = 0; e < 4; ++e) { yv[e] = (float)y4[e]; av[e] = (float)a4[e]; }
                                } else {
                                    yv = *(const f32x4_t*)((const T*)p.nb_y + vox * p.nb_ldy + co);
                                    av = *(const f32x4_t*)((const T*)p.nb_a + vox * p.nb_lda + co);
                                }
    #pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float da = (float)(T)o[e];
                                    const float dz = av[e] > 0.f ? da : da * p.nb_slope;
                                    st[j][e] += dz;
                                    st2[j][e] += dz * yv[e];
                                }
                            }
                        }
                    }
                }
            }
        }
        tile = ntile; kb = nkb; ks = nks; tc = ntc;
    }
    if constexpr (EPI == EPI_STORE) {
        if (do_stats) {
            if (cur_n >= 0) flush_stats(cur_n);
            __syncthreads();
            const int PN = p.N * COUTB * 2;
            float* wsp = p.stats_ws + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * PN;
            for (int i = tid; i < PN; i += NTHREADS) wsp[i] = spart[i];
            int* flag = (int*)(wpart + WAVES * COUTB * 2);
            if (grid_last_block(p.counter, gridDim.x * gridDim.y, flag)) {
                // 16 lanes share one (cout) output pair, every lane keeps up to 16 loads in flight; fixed order
                constexpr int PARTS = 16;
                const int nch = gridDim.y * COUTB;
                const int sub = tid % PARTS;
                for (int base = 0; base < nch; base += NTHREADS / PARTS) {
                    const int o = base + tid / PARTS;
                    const bool ok = o < nch;
                    const int y = ok ? o / COUTB : 0, cl = ok ? o % COUTB : 0;
                    const int cg = y * COUTB + cl;
                    float g0 = 0.f, g1 = 0.f;
                    for (int nn = 0; nn < p.N; ++nn) {
                        float s0 = 0.f, s1 = 0.f;
                        if (ok) {
                            const float* src = p.stats_ws + (long long)y * gridDim.x * PN + (nn * COUTB + cl) * 2;
#pragma unroll 8
                            for (int x = sub; x < (int)gridDim.x; x += PARTS) {
                                s0 += src[(long long)x * PN];
                                s1 += src[(long long)x * PN + 1];
                            }
                        }
#pragma unroll
                        for (int o2 = 1; o2 < PARTS; o2 <<= 1) {
                            s0 += __shfl_xor(s0, o2);
                            s1 += __shfl_xor(s1, o2);
                        }
                        if (ok && sub == 0 && cg < p.M) {
                            if (p.nb_y != nullptr) {
                                const float inv = 1.0f / (float)p.nb_S;
                                const float fs = p.nb_stats[((long long)nn * p.M + cg) * 2], fs2 = p.nb_stats[((long long)nn * p.M + cg) * 2 + 1];
                                const float mean = fs * inv;
                                float var = fs2 * inv - mean * mean;
                                var = var > 0.f ? var : 0.f;
                                s1 = rsqrtf(var + p.nb_eps) * (s1 - mean * s0);
                                g0 += s0;
                                g1 += s1;
                            }
                            p.stats[((long long)nn * p.M + cg) * 2 + 0] = s0;
                            p.stats[((long long)nn * p.M + cg) * 2 + 1] = s1;
                        }
                    }
                    if (ok && sub == 0 && cg < p.M && p.nb_y != nullptr && p.nb_dgamma != nullptr) {
                        p.nb_dbeta[cg] = p.nb_acc ? p.nb_dbeta[cg] + g0 : g0;
                        p.nb_dgamma[cg] = p.nb_acc ? p.nb_dgamma[cg] + g1 : g1;
                    }
                }
            }
        }
    }
}

template <typename T, int NTAPS, int SRC, int EPI, int TD, int TH, int TW, int WAVES, int NT, int STRIDE = 1, int NSL = 1>
int launch_cfg(IgemmParams& p, hipStream_t stream) {
    using C = IgemmCfg<T, NTAPS, SRC, EPI, TD, TH, TW, WAVES, NT, STRIDE, NSL>;
    p.tiles_d = ceil_div(p.D, TD);
    p.tiles_h = ceil_div(p.H, TH);
    p.tiles_w = ceil_div(p.W, TW);
    const long long nt = (long long)p.N * p.tiles_d * p.tiles_h * p.tiles_w;
    if (nt > 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "igemm: too many tiles");
    p.ntiles = (int)nt;
    p.NKB = ceil_div(p.K, C::CB);
    p.vec_store = ((((uintptr_t)p.y) % (4 * sizeof(T))) == 0 && (p.ldy % 4) == 0) ? 1 : 0;
    {
        const long long xh = STRIDE == 1 ? p.H : p.IH, xw = STRIDE == 1 ? p.W : p.IW;
        p.rel32_ok = ((long long)(C::PD + 1) * xh * xw * p.ldx < 0x7fffffffLL) ? 1 : 0;
    }
    auto kern = igemm_fwd_kernel<T, NTAPS, SRC, EPI, TD, TH, TW, WAVES, NT, STRIDE, NSL>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) !=
            hipSuccess)
            MSSEG_FAIL(MSSEG_ELAUNCH, "igemm: cannot set dynamic LDS size %d", C::LDS_BYTES);
        attr_set = true;
    }
    const int ncb = ceil_div(p.M, C::COUTB);
    const int wg_per_cu = (C::LDS_BYTES > 80 * 1024) ? 1 : ((C::LDS_BYTES > 40 * 1024) ? 2 : 4);
    int gx = msseg_num_cus() * wg_per_cu / (ncb > 1 ? 1 : 1);
    if (gx > p.ntiles) gx = p.ntiles;
    if (gx < 1) gx = 1;
    dim3 grid(gx, ncb, 1);
    hipLaunchKernelGGL(kern, grid, dim3(C::NTHREADS), C::LDS_BYTES, stream, p);
    MSSEG_CHECK_LAUNCH("igemm_fwd");
    return MSSEG_OK;
}

template <typename T, int NTAPS, int SRC, int EPI, int TD, int TH, int TW, int WAVES, int STRIDE = 1, int NSL = 1>
int launch_nt(IgemmParams& p, hipStream_t stream) {
    const int cb = p.cout_block ? p.cout_block : msseg_cout_block(p.M);
    switch (cb) {
        case 16: return launch_cfg<T, NTAPS, SRC, EPI, TD, TH, TW, WAVES, 1, STRIDE, NSL>(p, stream);
        case 32: return launch_cfg<T, NTAPS, SRC, EPI, TD, TH, TW, WAVES, 2, STRIDE, NSL>(p, stream);
        case 48: return launch_cfg<T, NTAPS, SRC, EPI, TD, TH, TW, WAVES, 3, STRIDE, NSL>(p, stream);
    }
    MSSEG_FAIL(MSSEG_EINVAL, "igemm: bad cout block %d", cb);
}

// Tile / cout-block choice for a conv k3 problem: the largest tile that still yields >= ~3/4 of a chip of
// workgroups; tiny grids (6^3 .. 12^3 with many channels) drop to 16-wide cout blocks to expose more parallelism
// (those layers are latency-bound chains of weight-block loads, not MFMA-bound).  cfg: 0 big, 1 mid, 2 small.
static void k3_plan(int N, int D, int H, int W, int M, int* cfg, int* cb) {
    const int mn = D < H ? (D < W ? D : W) : (H < W ? H : W);
    const int std_cb = msseg_cout_block(M);
    const long long want = (long long)msseg_num_cus() * 3 / 4;
    auto wgs = [&](int td, int th, int tw, int c) {
        return (long long)N * ceil_div(D, td) * ceil_div(H, th) * ceil_div(W, tw) * ceil_div(M, c);
    };
    if (mn >= 32 && wgs(4, 8, 16, std_cb) >= want) { *cfg = 0; *cb = std_cb; return; }
    if (mn >= 12 && wgs(4, 4, 8, std_cb) >= want) { *cfg = 1; *cb = std_cb; return; }
    *cfg = 2;
    *cb = (wgs(2, 4, 8, std_cb) >= want || std_cb == 16) ? std_cb : 16;
}

template <typename T> int launch_k3(IgemmParams& p, hipStream_t stream) {
    int cfg, cb;
    k3_plan(p.N, p.D, p.H, p.W, p.M, &cfg, &cb);
    p.cout_block = cb;
    if (cfg == 0) {
        // A/B switch.  Measured on MI355X (round 1): the tap-sliced 2-WG/CU variant is SLOWER (32->32 @96^3: 144 us
        // vs 110 us) -- weight-slice refetch + 3x barriers cost more than the second workgroup hides.
        static const bool old_big = getenv("MSSEG_K3_SLICED") == nullptr;
        if (old_big || cb == 48) return launch_nt<T, 27, SRC_DIRECT, EPI_STORE, 4, 8, 16, 8>(p, stream);  // 48-wide: sliced variant spills
        return launch_nt<T, 27, SRC_DIRECT, EPI_STORE, 2, 8, 16, 4, 1, 3>(p, stream);
    }
    if (cfg == 1) return launch_nt<T, 27, SRC_DIRECT, EPI_STORE, 4, 4, 8, 4>(p, stream);
    return launch_nt<T, 27, SRC_DIRECT, EPI_STORE, 2, 4, 8, 4>(p, stream);
}

template <typename T, int SRC, int EPI> int launch_flat(IgemmParams& p, hipStream_t stream) {
    return launch_nt<T, 1, SRC, EPI, 1, 1, 256, 4>(p, stream);
}

int check_common(const void* x, long long ldx, const void* wp, const void* y, long long ldy, int dtype, int esz) {
    if (!x || !wp || !y) MSSEG_FAIL(MSSEG_EINVAL, "igemm: null pointer");
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "igemm: bad dtype %d", dtype);
    if (((uintptr_t)x | (uintptr_t)wp) & 15) MSSEG_FAIL(MSSEG_EINVAL, "igemm: x/wp must be 16-byte aligned");
    if ((ldx * esz) % 16) MSSEG_FAIL(MSSEG_EINVAL, "igemm: ldx*elem must be a multiple of 16 bytes");
    (void)ldy;
    return MSSEG_OK;
}

}  // namespace

extern "C" {

int msseg_cout_block(int M) {
    if (M <= 16) return 16;
    if (M % 32 == 0) return 32;
    if (M % 48 == 0) return 48;
    return 32;
}

int msseg_conv3d_k3_cout_block(int N, int D, int H, int W, int Cout) {
    int cfg, cb;
    k3_plan(N, D, H, W, Cout, &cfg, &cb);
    return cb;
}

int msseg_conv3d_k3_variant(int N, int D, int H, int W, int Cout) {
    int cfg, cb;
    k3_plan(N, D, H, W, Cout, &cfg, &cb);
    return cfg;
}

static int k3_fwd_impl(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy, int N,
                       int D, int H, int W, int Cin, int Cout, float* stats, void* scratch, size_t scratch_bytes,
                       const void* nb_y, long long nb_ldy, const void* nb_a, long long nb_lda, const float* nb_stats,
                       float nb_slope, float nb_eps, float* nb_dgamma, float* nb_dbeta, int nb_acc, int dtype,
                       msseg_stream_t stream);

int msseg_conv3d_k3_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                        int N, int D, int H, int W, int Cin, int Cout, float* stats, void* scratch,
                        size_t scratch_bytes, int dtype, msseg_stream_t stream) {
    return k3_fwd_impl(x, ldx, wp, bias, y, ldy, N, D, H, W, Cin, Cout, stats, scratch, scratch_bytes, nullptr, 0, nullptr,
                       0, nullptr, 0.f, 0.f, nullptr, nullptr, 0, dtype, stream);
}

int msseg_conv3d_k3_dgrad_inbwd(const void* dy, long long lddy, const void* wp, void* da, long long ldda, int N, int D,
                                int H, int W, int Cin, int Cout, const void* yraw, long long ldyraw, const void* act,
                                long long ldact, const float* fwd_stats, float slope, float eps, float* red,
                                float* dgamma, float* dbeta, int accumulate, void* scratch, size_t scratch_bytes,
                                int dtype, msseg_stream_t stream) {
    if (!yraw || !act || !fwd_stats || !red) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_dgrad_inbwd: null pointer");
    if ((dgamma == nullptr) != (dbeta == nullptr)) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_dgrad_inbwd: dgamma/dbeta go together");
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    if (Cout % 4 || (ldyraw % 4) || (ldact % 4) || ((uintptr_t)yraw % (4 * esz)) || ((uintptr_t)act % (4 * esz)))
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_dgrad_inbwd: channel count / strides must be multiples of 4");
    return k3_fwd_impl(dy, lddy, wp, nullptr, da, ldda, N, D, H, W, Cin, Cout, red, scratch, scratch_bytes, yraw, ldyraw,
                       act, ldact, fwd_stats, slope, eps, dgamma, dbeta, accumulate, dtype, stream);
}

static int k3_fwd_impl(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy, int N,
                       int D, int H, int W, int Cin, int Cout, float* stats, void* scratch, size_t scratch_bytes,
                       const void* nb_y, long long nb_ldy, const void* nb_a, long long nb_lda, const float* nb_stats,
                       float nb_slope, float nb_eps, float* nb_dgamma, float* nb_dbeta, int nb_acc, int dtype,
                       msseg_stream_t stream) {
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    int rc = check_common(x, ldx, wp, y, ldy, dtype, esz);
    if (rc) return rc;
    if (N < 1 || D < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3: bad shape");
    if (Cin % (16 / esz)) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3: Cin=%d must be a multiple of %d (use conv3d_gather)", Cin, 16 / esz);
    if (ldx < Cin || ldy < Cout) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3: ld smaller than channels");
    IgemmParams p{};
    p.x = x; p.ldx = ldx; p.wp = wp; p.bias = bias; p.y = y; p.ldy = ldy;
    p.N = N; p.D = D; p.H = H; p.W = W; p.K = Cin; p.M = Cout;
    if (stats) {
        if (N > MSSEG_STATS_NMAX) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3: fused statistics need N <= %d", MSSEG_STATS_NMAX);
        if (!scratch || ((uintptr_t)scratch & 255) || scratch_bytes < msseg_reduce_scratch_bytes())
            MSSEG_FAIL(MSSEG_EWORKSPACE, "conv3d_k3: fused statistics need a zero-initialised scratch of %zu bytes",
                       msseg_reduce_scratch_bytes());
        p.stats = stats;
        p.counter = (unsigned int*)scratch;
        p.stats_ws = (float*)((unsigned char*)scratch + MSSEG_SCRATCH_COUNTER_BYTES);
        p.nb_y = nb_y; p.nb_ldy = nb_ldy; p.nb_a = nb_a; p.nb_lda = nb_lda; p.nb_stats = nb_stats;
        p.nb_slope = nb_slope; p.nb_eps = nb_eps; p.nb_S = (long long)D * H * W;
        p.nb_dgamma = nb_dgamma; p.nb_dbeta = nb_dbeta; p.nb_acc = nb_acc;
    }
    return dtype == MSSEG_F32 ? launch_k3<float>(p, (hipStream_t)stream) : launch_k3<bf16_t>(p, (hipStream_t)stream);
}

int msseg_conv3d_k3s2_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                          int N, int ID, int IH, int IW, int Cin, int Cout, int dtype, msseg_stream_t stream) {
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    int rc = check_common(x, ldx, wp, y, ldy, dtype, esz);
    if (rc) return rc;
    if (N < 1 || ID < 1 || IH < 1 || IW < 1 || Cin < 1 || Cout < 1) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3s2: bad shape");
    if (Cin % (16 / esz)) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3s2: Cin=%d must be a multiple of %d", Cin, 16 / esz);
    IgemmParams p{};
    p.x = x; p.ldx = ldx; p.wp = wp; p.bias = bias; p.y = y; p.ldy = ldy;
    p.N = N; p.ID = ID; p.IH = IH; p.IW = IW;
    p.D = (ID - 1) / 2 + 1; p.H = (IH - 1) / 2 + 1; p.W = (IW - 1) / 2 + 1;
    p.K = Cin; p.M = Cout;
    return dtype == MSSEG_F32 ? launch_nt<float, 27, SRC_DIRECT, EPI_STORE, 2, 4, 8, 4, 2>(p, (hipStream_t)stream)
                              : launch_nt<bf16_t, 27, SRC_DIRECT, EPI_STORE, 2, 4, 8, 4, 2>(p, (hipStream_t)stream);
}

int msseg_conv3d_k1_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                        long long NV, int Cin, int Cout, int dtype, msseg_stream_t stream) {
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    int rc = check_common(x, ldx, wp, y, ldy, dtype, esz);
    if (rc) return rc;
    if (NV < 1 || NV > 0x7fffffffLL || Cin < 1 || Cout < 1) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1: bad shape");
    if (Cin % (16 / esz)) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1: Cin=%d must be a multiple of %d", Cin, 16 / esz);
    IgemmParams p{};
    p.x = x; p.ldx = ldx; p.wp = wp; p.bias = bias; p.y = y; p.ldy = ldy;
    p.N = 1; p.D = 1; p.H = 1; p.W = (int)NV; p.K = Cin; p.M = Cout;
    return dtype == MSSEG_F32 ? launch_flat<float, SRC_DIRECT, EPI_STORE>(p, (hipStream_t)stream)
                              : launch_flat<bf16_t, SRC_DIRECT, EPI_STORE>(p, (hipStream_t)stream);
}

int msseg_conv3d_gather_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                            int N, int ID, int IH, int IW, int Cin, int Cout, int k, int s, int pd, int dtype,
                            msseg_stream_t stream) {
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    if (!x || !wp || !y) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_gather: null pointer");
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_gather: bad dtype");
    if (k < 1 || s < 1 || pd < 0 || Cin < 1) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_gather: bad kernel geometry");
    const int OD = (ID + 2 * pd - k) / s + 1, OH = (IH + 2 * pd - k) / s + 1, OW = (IW + 2 * pd - k) / s + 1;
    if (OD < 1 || OH < 1 || OW < 1) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_gather: empty output");
    const long long NV = (long long)N * OD * OH * OW;
    if (NV > 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_gather: too many voxels");
    IgemmParams p{};
    p.x = x; p.ldx = ldx; p.wp = wp; p.bias = bias; p.y = y; p.ldy = ldy;
    p.N = 1; p.D = 1; p.H = 1; p.W = (int)NV; p.K = Cin * k * k * k; p.M = Cout;
    p.ID = ID; p.IH = IH; p.IW = IW; p.OD = OD; p.OH = OH; p.OW = OW;
    p.cin = Cin; p.k = k; p.s = s; p.p = pd;
    return dtype == MSSEG_F32 ? launch_flat<float, SRC_GATHER, EPI_STORE>(p, (hipStream_t)stream)
                              : launch_flat<bf16_t, SRC_GATHER, EPI_STORE>(p, (hipStream_t)stream);
}

int msseg_deconv_k2s2_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                          int N, int D, int H, int W, int Cin, int Cout, int dtype, msseg_stream_t stream) {
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    int rc = check_common(x, ldx, wp, y, ldy, dtype, esz);
    if (rc) return rc;
    if (Cin % (16 / esz) || Cout % 4) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2: Cin %% %d and Cout %% 4 must be 0", 16 / esz);
    const long long NV = (long long)N * D * H * W;
    if (NV < 1 || NV > 0x7fffffffLL / 8) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2: bad voxel count");
    IgemmParams p{};
    p.x = x; p.ldx = ldx; p.wp = wp; p.bias = bias; p.y = y; p.ldy = ldy;
    p.N = 1; p.D = 1; p.H = 1; p.W = (int)NV; p.K = Cin; p.M = 8 * Cout;
    p.OD = D; p.OH = H; p.OW = W; p.creal = Cout;
    return dtype == MSSEG_F32 ? launch_flat<float, SRC_DIRECT, EPI_DECONV>(p, (hipStream_t)stream)
                              : launch_flat<bf16_t, SRC_DIRECT, EPI_DECONV>(p, (hipStream_t)stream);
}

int msseg_deconv_k2s2_bwd_data(const void* dy, long long lddy, const void* wp, void* dx, long long lddx,
                               int N, int D, int H, int W, int Cin, int Cout, int dtype, msseg_stream_t stream) {
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    int rc = check_common(dy, lddy, wp, dx, lddx, dtype, esz);
    if (rc) return rc;
    if (Cout % (16 / esz)) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2_bwd_data: Cout %% %d must be 0", 16 / esz);
    const long long NV = (long long)N * D * H * W;
    if (NV < 1 || NV > 0x7fffffffLL / 8) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2_bwd_data: bad voxel count");
    IgemmParams p{};
    p.x = dy; p.ldx = lddy; p.wp = wp; p.bias = nullptr; p.y = dx; p.ldy = lddx;
    p.N = 1; p.D = 1; p.H = 1; p.W = (int)NV; p.K = 8 * Cout; p.M = Cin;
    p.OD = D; p.OH = H; p.OW = W; p.creal = Cout;
    return dtype == MSSEG_F32 ? launch_flat<float, SRC_DECONV_BWD, EPI_STORE>(p, (hipStream_t)stream)
                              : launch_flat<bf16_t, SRC_DECONV_BWD, EPI_STORE>(p, (hipStream_t)stream);
}

}  // extern "C"
