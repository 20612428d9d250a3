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
#include "k3pp.h"

#include <stdlib.h>

#include <type_traits>

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

// DIAG (timing-only ablations, wrong results): 1 = no MFMA, 2 = no LDS fragment reads, 3 = no epilogue stores,
// 4 = no global prefetch loads
// DIAG == 5: per-phase cycle counters of workgroup 0 / wave 0 (s_memtime), read back by msseg_debug_phase_cycles()
__device__ unsigned long long g_phase_cycles[16];

template <typename T, int NTAPS, int SRC, int EPI, int TD, int TH, int TW, int WAVES, int NT, int STRIDE = 1, int NSL = 1, int DIAG = 0>
__global__ __launch_bounds__(WAVES * 64, (NSL > 1 ? 2 : 1)) void igemm_fwd_kernel(const IgemmParams p) {
    using C = IgemmCfg<T, NTAPS, SRC, EPI, TD, TH, TW, WAVES, NT, STRIDE, NSL>;
    constexpr int BT = C::BT;
    constexpr int EPC = C::EPC, CB = C::CB, PAD = C::PAD, PH = C::PH, PW = C::PW, HV = C::HV, MT = C::MT;
    constexpr int NTHREADS = C::NTHREADS, COUTB = C::COUTB, PLANE = C::PLANE;
    constexpr bool HREUSE = NTAPS == 27 && STRIDE == 1 && NSL == 1 && TW == 16 && TH % MT == 0 && DIAG != 1 && DIAG != 2;
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
                int tt = tc.n * p.W + cv;   // flat index over all samples (p.N == 1: tc.n == 0)
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
    int a_rel[NIT_A];
    // small tiles also keep the packed halo coordinates (hd | hh << 8 | hw << 16): on the 24^3 ... 6^3 grids nearly every
    // tile is an edge tile and the bounds test runs inside the MFMA loop on one wave per SIMD, where recomputing the
    // coordinates (two divisions by constants per chunk) is exposed latency; the large tiles (NIT_A up to 17) recompute
    constexpr bool KEEP_PK = SRC == SRC_DIRECT && NIT_A <= 8;
    int a_pk[KEEP_PK ? NIT_A : 1];
    const int XD = STRIDE == 1 ? p.D : p.ID, XH = STRIDE == 1 ? p.H : p.IH, XW = STRIDE == 1 ? p.W : p.IW;
    if constexpr (SRC == SRC_DIRECT) {
#pragma unroll
        for (int it = 0; it < NIT_A; ++it) {
            const int i = tid + it * NTHREADS;
            const int cq = i & 3, hv = i >> 2;
            const int hw = hv % PW, t2 = hv / PW, hh = t2 % PH, hd = t2 / PH;
            a_rel[it] = (int)((((long long)hd * XH + hh) * XW + hw) * p.ldx) + cq * EPC;
            if constexpr (KEEP_PK) a_pk[it] = hd | (hh << 8) | (hw << 16);
        }
    }
    // The global loads of a stage are issued one at a time (fetch_a / fetch_b) from slots spread over the MFMA loop
    // of the previous stage, after fetch_setup() has fixed the stage's base pointers.
    const T* f_bp = xg;
    const u32x4_t* f_bsrc = nullptr;
    bool f_interior = false, f_cok = false;
    int f_dB = 0, f_hB = 0, f_wB = 0, f_kb = 0;
    TileCo f_tc{0, 0, 0, 0};
    auto fetch_setup = [&](const TileCo& tc, int kb, int ks) {
        f_tc = tc;
        f_kb = kb;
        if constexpr (SRC == SRC_DIRECT) {
            f_dB = tc.d0 * STRIDE - PAD; f_hB = tc.h0 * STRIDE - PAD; f_wB = tc.w0 * STRIDE - PAD;
            const long long basev = (((long long)tc.n * XD + f_dB) * XH + f_hB) * XW + f_wB;
            f_bp = xg + basev * p.ldx + kb * CB;
            f_interior = f_dB >= 0 && f_dB + C::PD <= XD && f_hB >= 0 && f_hB + PH <= XH && f_wB >= 0 && f_wB + PW <= XW;
            const int nchunk = (p.K - kb * CB + EPC - 1) / EPC;   // valid 16-byte chunks of this channel block
            f_cok = (tid & 3) < nchunk;
        }
        f_bsrc = (const u32x4_t*)((const unsigned char*)p.wp + (((long long)coutblk * p.NKB + kb) * NSL + ks) * C::B_BYTES);
    };
    auto fetch_a = [&](int it) {
        if constexpr (DIAG == 4) { pa[it] = u32x4_t{1u, 2u, 3u, 4u}; return; }
        const int i = tid + it * NTHREADS;
        if constexpr (SRC == SRC_DIRECT) {
            if (p.rel32_ok) {
                bool ok = f_cok && i < HV * 4;
                if (!f_interior) {   // edge tile
                    int d, h, w;
                    if constexpr (KEEP_PK) {
                        const int pk = a_pk[it];
                        d = f_dB + (pk & 255); h = f_hB + ((pk >> 8) & 255); w = f_wB + (pk >> 16);
                    } else {         // recompute this chunk's halo coordinates (kept out of registers)
                        const int hv = i >> 2;
                        const int hw = hv % PW, t2 = hv / PW;
                        d = f_dB + t2 / PH; h = f_hB + t2 % PH; w = f_wB + hw;
                    }
                    ok = ok && (unsigned)d < (unsigned)XD && (unsigned)h < (unsigned)XH && (unsigned)w < (unsigned)XW;
                }
                pa[it] = ok ? *(const u32x4_t*)(f_bp + a_rel[it]) : u32x4_t{0u, 0u, 0u, 0u};
                return;
            }
        }
        pa[it] = (i < HV * 4) ? load_a_chunk(f_tc, f_kb, i) : u32x4_t{0u, 0u, 0u, 0u};
    };
    auto fetch_b = [&](int it) {
        if constexpr (DIAG == 4) { pb[it] = u32x4_t{1u, 2u, 3u, 4u}; return; }
        const int i = tid + it * NTHREADS;
        pb[it] = (i < C::B_BYTES / 16) ? f_bsrc[i] : u32x4_t{0u, 0u, 0u, 0u};
    };
    auto fetch = [&](const TileCo& tc, int kb, int ks, bool with_a, bool with_b) {   // whole stage at once (prologue)
        fetch_setup(tc, kb, ks);
        if (with_a) {
#pragma unroll
            for (int it = 0; it < NIT_A; ++it) fetch_a(it);
        }
        if (with_b) {
#pragma unroll
            for (int it = 0; it < NIT_B; ++it) fetch_b(it);
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
    // Deferred epilogue stores: a finished tile's outputs are kept packed in registers and written at the START of
    // the next stage, BEFORE that stage's prefetch loads are issued.  The wait in front of the LDS commit
    // (s_waitcnt vmcnt(0) at the loop head) then only sees operations that had a whole MFMA phase to complete,
    // instead of stalling every tile on the write latency of stores issued just before it.
    using PendT = typename std::conditional<sizeof(T) == 2, bf16x4_t, f32x4_t>::type;
    PendT pend[MT][NT];
    TileCo ptc{0, 0, 0, 0};
    bool pend_valid = false;
    auto store_one = [&](int m, int j) {
        const int v = (wave * MT + m) * 16 + r;
        const int td = v / (TH * TW), th = (v / TW) % TH, tw = v % TW;
        const int d = ptc.d0 + td, h = ptc.h0 + th, w = ptc.w0 + tw;
        if (d >= p.D || h >= p.H || w >= p.W) return;
        const int co = coutblk * COUTB + j * 16 + q * 4;
        if (co + 4 > p.M) return;
        T* dst;
        if constexpr (EPI == EPI_STORE) {
            const long long vox = (((long long)ptc.n * p.D + d) * p.H + h) * p.W + w;
            dst = yg + vox * p.ldy + co;
        } else {
            int tt = w;
            const int dw = tt % p.OW; tt /= p.OW;
            const int dh = tt % p.OH; tt /= p.OH;
            const int dd = tt % p.OD, dn = tt / p.OD;
            const int abc = co / p.creal;
            const int cbase = co - abc * p.creal;
            const int fd = 2 * dd + (abc >> 2), fh = 2 * dh + ((abc >> 1) & 1), fw = 2 * dw + (abc & 1);
            const long long fv = (((long long)dn * (2 * p.OD) + fd) * (2 * p.OH) + fh) * (2 * p.OW) + fw;
            dst = yg + fv * p.ldy + cbase;
        }
        *(PendT*)dst = pend[m][j];
    };
    auto store_pending = [&]() {
        if (!pend_valid) return;
        pend_valid = false;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j) store_one(m, j);
    };
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto mark = [&](int k) {
        if constexpr (DIAG == 5) {
            const unsigned long long now = __builtin_readcyclecounter();
            ph[k] += now - tprev;
            tprev = now;
        }
    };
    if constexpr (DIAG == 5) tprev = __builtin_readcyclecounter();
    int tile = blockIdx.x, kb = 0, ks = 0;
    TileCo tc = decode(tile < p.ntiles ? tile : 0);
    bool a_pending = true, b_pending = true;
    if (tile < p.ntiles) fetch(tc, 0, 0, true, true);
    while (tile < p.ntiles) {
        __syncthreads();  // everyone finished reading the previous stage's LDS image
        mark(0);
        if constexpr (DIAG == 5) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); mark(1); }
        commit(a_pending, b_pending);
        if constexpr (DIAG == 5) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); mark(2); }
        __syncthreads();
        mark(3);
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
        // Side work of this stage, issued from slots spread over the MFMA loop so that it overlaps the matrix pipe
        // instead of serialising all waves in front of it: the previous tile's output stores (held in pend[][]),
        // then the next stage's halo and weight loads.
        const bool st_now = pend_valid, ld_now = ntile < p.ntiles;
        pend_valid = false;
        fetch_setup(ntc, nkb, nks);
        constexpr int NI_ST = MT * NT, NI = NI_ST + NIT_A + NIT_B;
        auto side_item = [&](int i) {
            if (i < NI_ST) {
                if (st_now) store_one(i / NT, i % NT);
            } else if (i < NI_ST + NIT_A) {
                if (ld_now && a_pending) fetch_a(i - NI_ST);
            } else {
                if (ld_now && b_pending) fetch_b(i - NI_ST - NIT_A);
            }
        };
        if constexpr (NSL > 1) {
#pragma unroll
            for (int i = 0; i < NI; ++i) side_item(i);
        }
        mark(4);

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
                if constexpr (DIAG == 2) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) bf[buf][j] = u32x4_t{(unsigned)(t + j), 1u, 2u, 3u};
#pragma unroll
                    for (int m = 0; m < MT; ++m) af[buf][m] = u32x4_t{(unsigned)(t + m), 5u, 6u, 7u};
                    return;
                }
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bf[buf][j] = *(const u32x4_t*)(ldsB + bbase + t * (4 * COUTB * 16) + j * 256);
#pragma unroll
                for (int m = 0; m < MT; ++m) af[buf][m] = *(const u32x4_t*)(ldsA + abase[m] + ksoff + toff);
            };
            if constexpr (HREUSE) {
                // Row-reuse order (k3, 16-wide tiles whose MT voxel rows per wave are consecutive in h): for a fixed
                // (kd, kw) an activation row of the halo feeds up to three output rows (kh = 0..2), so it is read from
                // LDS once instead of three times and the weights of the three kh taps stay in registers while the
                // MT+2 halo rows stream past: 108 fragment reads per stage instead of 162 for the same 216 MFMAs.
                u32x4_t wf[3][NT], xf[2];
                const int xb = abase[0];
                auto ldw = [&](int g, int kh) {
                    const int tap = (g / 3) * 9 + kh * 3 + (g % 3);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        wf[kh][j] = *(const u32x4_t*)(ldsB + bbase + tap * (4 * COUTB * 16) + j * 256);
                };
                auto ldx = [&](int g, int hr, int buf) {
                    xf[buf] = *(const u32x4_t*)(ldsA + xb + (((g / 3) * PH + hr) * PW + (g % 3)) * 16);
                };
                constexpr int NSTEP = 9 * (MT + 2);
                ldw(0, 0); ldw(0, 1); ldw(0, 2);
                ldx(0, 0, 0);
#pragma unroll
                for (int g = 0; g < 9; ++g) {
#pragma unroll
                    for (int hr = 0; hr < MT + 2; ++hr) {
                        const int step = g * (MT + 2) + hr, cur = step & 1;
                        if (hr + 1 < MT + 2) ldx(g, hr + 1, cur ^ 1);
                        else if (g + 1 < 9) ldx(g + 1, 0, cur ^ 1);
#pragma unroll
                        for (int i = 0; i < NI; ++i)
                            if (i * NSTEP / NI == step) side_item(i);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int kh = 0; kh < 3; ++kh) {
                            const int m = hr - kh;
                            if (m >= 0 && m < MT) {
#pragma unroll
                                for (int j = 0; j < NT; ++j) mma_chunk<T>(acc[m][j], wf[kh][j], xf[cur]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        // a kh tap's weights are free once its last output row is issued: refill for the next (kd, kw)
                        if (g + 1 < 9) {
                            if (hr == MT - 1) ldw(g + 1, 0);
                            if (hr == MT) ldw(g + 1, 1);
                            if (hr == MT + 1) ldw(g + 1, 2);
                        }
                    }
                }
            } else if constexpr (NSL == 1) {
                load_frags(0, 0);
#pragma unroll
                for (int t = 0; t < BT; ++t) {
                    const int cur = t & 1;
                    if (t + 1 < BT) load_frags(t + 1, cur ^ 1);
#pragma unroll
                    for (int i = 0; i < NI; ++i)
                        if (i * BT / NI == t) side_item(i);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            if constexpr (DIAG == 1) {
                                asm volatile("" ::"v"(bf[cur][j]), "v"(af[cur][m]));   // keep the reads alive, no MFMA
                            } else {
                                mma_chunk<T>(acc[m][j], bf[cur][j], af[cur][m]);
                            }
                        }
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
        // the held outputs are written by now: give them a definite (dead) state so they free their registers
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j) pend[m][j] = PendT{};
        mark(5);
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
                    if constexpr (DIAG == 3) {
                        asm volatile("" ::"v"(o));
                        continue;
                    }
                    if (p.vec_store && co + 4 <= p.M) {
                        if (p.bias) {
    #pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += p.bias[cbase + e];
                        }
                        if constexpr (sizeof(T) == 2) {
                            pend[m][j] = bf16x4_t{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                        } else {
                            pend[m][j] = o;
                        }
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
        if (kb == p.NKB - 1 && ks == NSL - 1 && p.vec_store) { ptc = tc; pend_valid = true; }
        mark(6);
        tile = ntile; kb = nkb; ks = nks; tc = ntc;
    }
    store_pending();
    if constexpr (DIAG == 5) {
        if (blockIdx.x == 0 && tid == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) g_phase_cycles[k] = ph[k];
        }
    }
    if constexpr (EPI == EPI_STORE) {
        if (do_stats) {
            if (cur_n >= 0) flush_stats(cur_n);
            __syncthreads();
            const int PN = p.N * COUTB * 2;
            float* wsp = p.stats_ws + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * PN;
            for (int i = tid; i < PN; i += NTHREADS) wsp[i] = spart[i];
        }
    }
}

// second step of the fused epilogue reductions: see K3FinParams (k3pp.h)
__global__ __launch_bounds__(256) void k3_stats_finalize_kernel(const K3FinParams f) {
    __shared__ __attribute__((aligned(16))) float fin[256 * 4 + 8 * 48 * 2];
    const int tid = threadIdx.x;
    const int L = f.N * f.coutb * 2;
    const int cbase = blockIdx.x * f.coutb;
    if (f.nb_stats == nullptr) {
        // forward statistics: one block per (cout block, sample) sums that sample's column slice of the partial rows
        // (with one block per cout block a thread walked 64 ... 128 rows: 5 us at N = 2, 8 us at N = 8)
        const int n = blockIdx.y, Ls = f.coutb * 2;
        block_rows_sum<256>(f.ws + (long long)blockIdx.x * f.R * L + n * Ls, f.R, Ls, fin, L);
        const float* tot = fin + 256 * 4;   // [cl][2]
        for (int i = tid; i < Ls; i += 256) {
            const int cl = i >> 1, k = i & 1;
            if (cbase + cl < f.M) f.stats[((long long)n * f.M + cbase + cl) * 2 + k] = tot[i];
        }
        return;
    }
    block_rows_sum<256>(f.ws + (long long)blockIdx.x * f.R * L, f.R, L, fin);
    const float* tot = fin + 256 * 4;   // [n][cl][2]
    if (tid < f.coutb && cbase + tid < f.M) {
        // sum dz*xhat = rstd * (sum dz*yraw - mean * sum dz); dbeta / dgamma = sums over the samples
        const int cg = cbase + tid;
        float g0 = 0.f, g1 = 0.f;
        const float inv = 1.0f / (float)f.nb_S;
        for (int nn = 0; nn < f.N; ++nn) {
            const float a = tot[(nn * f.coutb + tid) * 2], b0 = tot[(nn * f.coutb + tid) * 2 + 1];
            const float fs = f.nb_stats[((long long)nn * f.M + cg) * 2], fs2 = f.nb_stats[((long long)nn * f.M + cg) * 2 + 1];
            const float mean = fs * inv;
            float var = fs2 * inv - mean * mean;
            var = var > 0.f ? var : 0.f;
            const float b2 = rsqrtf(var + f.nb_eps) * (b0 - mean * a);
            g0 += a;
            g1 += b2;
            f.stats[((long long)nn * f.M + cg) * 2 + 0] = a;
            f.stats[((long long)nn * f.M + cg) * 2 + 1] = b2;
        }
        if (f.nb_dgamma != nullptr) {
            f.nb_dbeta[cg] = f.nb_acc ? f.nb_dbeta[cg] + g0 : g0;
            f.nb_dgamma[cg] = f.nb_acc ? f.nb_dgamma[cg] + g1 : g1;
        }
    }
}

template <typename T, int NTAPS, int SRC, int EPI, int TD, int TH, int TW, int WAVES, int NT, int STRIDE = 1, int NSL = 1, int DIAG = 0>
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
    auto kern = igemm_fwd_kernel<T, NTAPS, SRC, EPI, TD, TH, TW, WAVES, NT, STRIDE, NSL, DIAG>;
    static msseg_lds_attr_once attr;
    if (!attr.ensure((const void*)kern, C::LDS_BYTES)) MSSEG_FAIL(MSSEG_ELAUNCH, "igemm: cannot set dynamic LDS size %d", C::LDS_BYTES);
    const int ncb = ceil_div(p.M, C::COUTB);
    const int wg_per_cu = (C::LDS_BYTES > 80 * 1024) ? 1 : ((C::LDS_BYTES > 40 * 1024) ? 2 : 4);
    int gx = msseg_num_cus() * wg_per_cu / (ncb > 1 ? 1 : 1);
    if (gx > p.ntiles) gx = p.ntiles;
    if (gx < 1) gx = 1;
    dim3 grid(gx, ncb, 1);
    static const char* const kt_name = NTAPS != 27 ? "igemm_fwd_kernel<flat>"
                                       : (TD * TH * TW == 512 ? "igemm_fwd_kernel<27,4x8x16>"
                                          : (TD * TH * TW == 128 ? "igemm_fwd_kernel<27,4x4x8>" : "igemm_fwd_kernel<27,2x4x8>"));
    MSSEG_KTIMED(kt_name, stream, hipLaunchKernelGGL(kern, grid, dim3(C::NTHREADS), C::LDS_BYTES, stream, p));
    MSSEG_CHECK_LAUNCH("igemm_fwd");
    if (EPI == EPI_STORE && p.stats != nullptr) {
        K3FinParams f{};
        f.ws = p.stats_ws; f.R = gx; f.N = p.N; f.coutb = C::COUTB; f.M = p.M; f.stats = p.stats;
        f.nb_stats = p.nb_y ? p.nb_stats : nullptr; f.nb_eps = p.nb_eps; f.nb_S = p.nb_S;
        f.nb_dgamma = p.nb_dgamma; f.nb_dbeta = p.nb_dbeta; f.nb_acc = p.nb_acc;
        return msseg_k3_stats_finalize(f, ncb, stream);
    }
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
    {   // tuning override for grids below 32 voxels per axis: MSSEG_K3_FORCE="<cfg 1|2>,<cout block 16|32>"
        static const char* force = getenv("MSSEG_K3_FORCE");
        if (force && mn < 32 && std_cb == 32) {
            *cfg = force[0] - '0';
            *cb = atoi(force + 2);
            return;
        }
    }
    const long long want = (long long)msseg_num_cus() * 3 / 4;
    auto wgs = [&](int td, int th, int tw, int c) {
        return (long long)N * ceil_div(D, td) * ceil_div(H, th) * ceil_div(W, tw) * ceil_div(M, c);
    };
    if (mn >= 32 && wgs(4, 8, 16, std_cb) >= want) {
        *cfg = 0;
        // the 48-wide (NT = 3) instantiation of the big tile spills (142 VGPRs): two 32-wide blocks, the second half
        // empty, measured faster at 96^3 (48->48: 488 vs 633 us, 96->48: 731 vs 948 us; 16-wide blocks: 491 / 778 us)
        *cb = (std_cb == 48) ? 32 : std_cb;
        return;
    }
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
        if constexpr (sizeof(T) == 2) {
            static const char* diag = getenv("MSSEG_DIAG");   // timing-only ablations of the dominant kernel
            if (diag && cb == 32) {
                p.cout_block = 32;
                switch (diag[0]) {
                    case '1': return launch_cfg<T, 27, SRC_DIRECT, EPI_STORE, 4, 8, 16, 8, 2, 1, 1, 1>(p, stream);
                    case '2': return launch_cfg<T, 27, SRC_DIRECT, EPI_STORE, 4, 8, 16, 8, 2, 1, 1, 2>(p, stream);
                    case '3': return launch_cfg<T, 27, SRC_DIRECT, EPI_STORE, 4, 8, 16, 8, 2, 1, 1, 3>(p, stream);
                    case '4': return launch_cfg<T, 27, SRC_DIRECT, EPI_STORE, 4, 8, 16, 8, 2, 1, 1, 4>(p, stream);
                    case '5': return launch_cfg<T, 27, SRC_DIRECT, EPI_STORE, 4, 8, 16, 8, 2, 1, 1, 5>(p, stream);
                }
            }
        }
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

int msseg_k3_stats_finalize(const K3FinParams& f, int ncb, hipStream_t stream) {
    hipLaunchKernelGGL(k3_stats_finalize_kernel, dim3(ncb, f.nb_stats ? 1 : f.N), dim3(256), 0, stream, f);
    MSSEG_CHECK_LAUNCH("k3_stats_finalize");
    return MSSEG_OK;
}

// tools-only (not part of include/msseg.h): cycle counters written by the MSSEG_DIAG=5 build of the big-tile kernel
extern "C" int msseg_debug_phase_cycles(unsigned long long* out8) {
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_phase_cycles), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}

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

int msseg_conv3d_k3_kernel(int N, int D, int H, int W, int Cin, int Cout, int dtype) {
    int cfg, cb;
    if (dtype == MSSEG_BF16 && Cin == 48 && msseg_k3c48_shape_ok(N, D, H, W, Cout)) return 4;
    k3_plan(N, D, H, W, Cout, &cfg, &cb);
    if (dtype == MSSEG_BF16 && cb == 32) {
        K3ppParams pp{};
        pp.x = (const void*)256; pp.y = (void*)256; pp.ldx = Cin; pp.ldy = Cout;
        pp.N = N; pp.D = D; pp.H = H; pp.W = W; pp.K = Cin; pp.M = Cout;
        if ((Cin % 8) == 0 && (Cout % 4) == 0 && msseg_k3pp_eligible(pp)) return 3;
    }
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

int msseg_conv3d_k3_fwd_accumulate(const void* x, long long ldx, const void* wp, void* y, long long ldy, int N, int D, int H,
                                   int W, int Cin, int Cout, float* stats, void* scratch, size_t scratch_bytes, int dtype,
                                   msseg_stream_t stream) {
    if (!x || !wp || !y || !stats) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_fwd_accumulate: null pointer");
    if (dtype != MSSEG_BF16 || N < 1 || N > MSSEG_STATS_NMAX) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_fwd_accumulate: bf16, N <= %d", MSSEG_STATS_NMAX);
    if (!scratch || ((uintptr_t)scratch & 255) || scratch_bytes < msseg_reduce_scratch_bytes())
        MSSEG_FAIL(MSSEG_EWORKSPACE, "conv3d_k3_fwd_accumulate: needs the reduce scratch of %zu bytes", msseg_reduce_scratch_bytes());
    K3ppParams pp{};
    pp.x = x; pp.ldx = ldx; pp.wp = wp; pp.y = y; pp.ldy = ldy;
    pp.N = N; pp.D = D; pp.H = H; pp.W = W; pp.K = Cin; pp.M = Cout;
    pp.stats = stats; pp.counter = (unsigned int*)scratch;
    pp.stats_ws = (float*)((unsigned char*)scratch + MSSEG_SCRATCH_COUNTER_BYTES);
    pp.accumulate = 1;
    if (Cin == 48) {
        if (!msseg_k3c48_eligible(pp))
            MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_fwd_accumulate: 48 input channels need a shape and operands the 48-channel "
                                     "kernel takes (msseg_conv3d_k3_kernel() == 4, 16-byte aligned x, N <= 4)");
        return msseg_k3c48_launch(pp, (hipStream_t)stream);
    }
    if (!msseg_k3pp_eligible(pp))
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_fwd_accumulate: only shapes of the ping-pong kernel (32 input channels, Cout %% 32 == 0, "
                                 "large grids; msseg_conv3d_k3_kernel() == 3)");
    return msseg_k3pp_launch(pp, (hipStream_t)stream);
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
        if (Cout > 64 * 16) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3: fused statistics need Cout <= 1024 (one counter per cout block)");
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
    if (dtype == MSSEG_BF16) {
        // the 32-input-channel layers (the 96^3 / 48^3 levels) run on the ping-pong kernel when the grid is large enough
        K3ppParams pp{};
        pp.x = x; pp.ldx = ldx; pp.wp = wp; pp.bias = bias; pp.y = y; pp.ldy = ldy;
        pp.N = N; pp.D = D; pp.H = H; pp.W = W; pp.K = Cin; pp.M = Cout;
        pp.stats = p.stats; pp.stats_ws = p.stats_ws; pp.counter = p.counter;
        pp.nb_y = p.nb_y; pp.nb_ldy = p.nb_ldy; pp.nb_a = p.nb_a; pp.nb_lda = p.nb_lda; pp.nb_stats = p.nb_stats;
        pp.nb_slope = p.nb_slope; pp.nb_eps = p.nb_eps; pp.nb_S = p.nb_S;
        pp.nb_dgamma = p.nb_dgamma; pp.nb_dbeta = p.nb_dbeta; pp.nb_acc = p.nb_acc;
        if (Cin == 48 && msseg_k3c48_shape_ok(N, D, H, W, Cout)) {
            // the packed image is the four-part one of the 48-channel kernel (msseg_conv3d_k3_kernel() == 4): no fallback
            if (!msseg_k3c48_eligible(pp))
                MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3: 48-input-channel layer on a large grid needs 16-byte aligned x, 8-byte "
                                         "aligned y and ldx <= 256");
            return msseg_k3c48_launch(pp, (hipStream_t)stream);
        }
        int cfg, cb;
        k3_plan(N, D, H, W, Cout, &cfg, &cb);   // the packed weight image must be the 32-wide one
        if (cb == 32 && msseg_k3pp_eligible(pp)) return msseg_k3pp_launch(pp, (hipStream_t)stream);
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

int msseg_conv3d_stem_norm_fwd(const void* x, long long ldx, const void* wp, const float* bias, const float* stats,
                               const float* gamma, const float* beta, float eps, float slope, void* y, long long ldy, int N,
                               int D, int H, int W, int Cout, int dtype, msseg_stream_t stream) {
    if (!x || !wp || !y || !stats || N < 1 || D < 1 || H < 1 || W < 1) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_stem_norm: bad args");
    if (!msseg_stem_eligible(dtype, 1, Cout, 3, 1, 1, ldx, ldy, y))
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_stem_norm: bf16, Cout a multiple of 32 or 48 (<= 256), 8-byte aligned output rows only");
    StemParams sp{};
    sp.x = x; sp.ldx = ldx; sp.wp = wp; sp.bias = bias; sp.y = y; sp.ldy = ldy;
    sp.N = N; sp.D = D; sp.H = H; sp.W = W; sp.M = Cout;
    sp.nstats = stats; sp.gamma = gamma; sp.beta = beta; sp.eps = eps; sp.slope = slope;
    return msseg_stem_fwd_launch(sp, (hipStream_t)stream);
}

int msseg_conv3d_k1_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                        long long NV, int Cin, int Cout, int dtype, msseg_stream_t stream) {
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    int rc = check_common(x, ldx, wp, y, ldy, dtype, esz);
    if (rc) return rc;
    if (NV < 1 || NV > 0x7fffffffLL || Cin < 1 || Cout < 1) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1: bad shape");
    if (Cin % (16 / esz)) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1: Cin=%d must be a multiple of %d", Cin, 16 / esz);
    // many tokens, few channels (the first Swin stage's Linear layers): register-resident-weight streaming kernel
    if (msseg_linear_regw_eligible(dtype, NV, Cin, Cout, x, ldx, y, ldy, bias))
        return msseg_linear_regw_launch(x, ldx, wp, bias, y, ldy, NV, Cin, Cout, (hipStream_t)stream);
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
    if (msseg_stem_eligible(dtype, Cin, Cout, k, s, pd, ldx, ldy, y)) {
        StemParams sp{};
        sp.x = x; sp.ldx = ldx; sp.wp = wp; sp.bias = bias; sp.y = y; sp.ldy = ldy;
        sp.N = N; sp.D = ID; sp.H = IH; sp.W = IW; sp.M = Cout; sp.taps = k == 1 ? 1 : 27;
        return msseg_stem_fwd_launch(sp, (hipStream_t)stream);
    }
    IgemmParams p{};
    p.x = x; p.ldx = ldx; p.wp = wp; p.bias = bias; p.y = y; p.ldy = ldy;
    p.N = 1; p.D = 1; p.H = 1; p.W = (int)NV; p.K = Cin * k * k * k; p.M = Cout;
    p.ID = ID; p.IH = IH; p.IW = IW; p.OD = OD; p.OH = OH; p.OW = OW;
    p.cin = Cin; p.k = k; p.s = s; p.p = pd;
    return dtype == MSSEG_F32 ? launch_flat<float, SRC_GATHER, EPI_STORE>(p, (hipStream_t)stream)
                              : launch_flat<bf16_t, SRC_GATHER, EPI_STORE>(p, (hipStream_t)stream);
}

static int stem_fwd_taps(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy, int N, int D,
                         int H, int W, int Cout, float* stats, void* scratch, size_t scratch_bytes, int dtype, int taps,
                         msseg_stream_t stream);

int msseg_conv3d_stem_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                          int N, int D, int H, int W, int Cout, float* stats, void* scratch, size_t scratch_bytes,
                          int dtype, msseg_stream_t stream) {
    return stem_fwd_taps(x, ldx, wp, bias, y, ldy, N, D, H, W, Cout, stats, scratch, scratch_bytes, dtype, 27, stream);
}

int msseg_conv3d_stem_k1_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                             int N, int D, int H, int W, int Cout, float* stats, void* scratch, size_t scratch_bytes,
                             int dtype, msseg_stream_t stream) {
    return stem_fwd_taps(x, ldx, wp, bias, y, ldy, N, D, H, W, Cout, stats, scratch, scratch_bytes, dtype, 1, stream);
}

static int stem_fwd_taps(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy, int N, int D,
                         int H, int W, int Cout, float* stats, void* scratch, size_t scratch_bytes, int dtype, int taps,
                         msseg_stream_t stream) {
    if (!x || !wp || (!y && !stats) || N < 1 || D < 1 || H < 1 || W < 1) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_stem: bad args");
    if (!msseg_stem_eligible(dtype, 1, Cout, 3, 1, 1, ldx, y ? ldy : 4, y))   // y == NULL: statistics only
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_stem: bf16, Cout a multiple of 32 or 48 (<= 256), 8-byte aligned output rows only");
    StemParams sp{};
    sp.x = x; sp.ldx = ldx; sp.wp = wp; sp.bias = bias; sp.y = y; sp.ldy = ldy;
    sp.N = N; sp.D = D; sp.H = H; sp.W = W; sp.M = Cout; sp.taps = taps;
    if (stats) {
        if (N > MSSEG_STATS_NMAX) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_stem: fused statistics need N <= %d", MSSEG_STATS_NMAX);
        if (!scratch || ((uintptr_t)scratch & 255) || scratch_bytes < msseg_reduce_scratch_bytes())
            MSSEG_FAIL(MSSEG_EWORKSPACE, "conv3d_stem: fused statistics need a scratch of %zu bytes", msseg_reduce_scratch_bytes());
        sp.stats = stats;
        sp.stats_ws = (float*)((unsigned char*)scratch + MSSEG_SCRATCH_COUNTER_BYTES);
    }
    return msseg_stem_fwd_launch(sp, (hipStream_t)stream);
}

int msseg_deconv_k2s2_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                          int N, int D, int H, int W, int Cin, int Cout, int dtype, msseg_stream_t stream) {
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    int rc = check_common(x, ldx, wp, y, ldy, dtype, esz);
    if (rc) return rc;
    if (Cin % (16 / esz) || Cout % 4) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2: Cin %% %d and Cout %% 4 must be 0", 16 / esz);
    const long long NV = (long long)N * D * H * W;
    if (NV < 1 || NV > 0x7fffffffLL / 8) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2: bad voxel count");
    if (msseg_deconv2_fast_eligible(dtype, Cin, Cout, x, ldx, y, ldy, bias))
        return msseg_deconv2_fwd_launch(x, ldx, wp, bias, y, ldy, N, D, H, W, Cin, Cout, (hipStream_t)stream);
    if (msseg_deconv2g_fwd_eligible(dtype, Cin, Cout, x, ldx, y, ldy, bias))
        return msseg_deconv2g_fwd_launch(x, ldx, wp, bias, y, ldy, N, D, H, W, Cin, Cout, (hipStream_t)stream);
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
    if (msseg_deconv2_fast_eligible(dtype, Cin, Cout, dx, lddx, dy, lddy, nullptr))
        return msseg_deconv2_bwd_launch(dy, lddy, wp, dx, lddx, N, D, H, W, Cin, Cout, nullptr, 0, nullptr, 0, nullptr, 0.f, 0.f,
                                        nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, (hipStream_t)stream);
    if (msseg_deconv2g_bwd_eligible(dtype, Cin, Cout, dx, lddx, dy, lddy))
        return msseg_deconv2g_bwd_launch(dy, lddy, wp, dx, lddx, nullptr, N, D, H, W, Cin, Cout, (hipStream_t)stream);
    IgemmParams p{};
    p.x = dy; p.ldx = lddy; p.wp = wp; p.bias = nullptr; p.y = dx; p.ldy = lddx;
    p.N = 1; p.D = 1; p.H = 1; p.W = (int)NV; p.K = 8 * Cout; p.M = Cin;
    p.OD = D; p.OH = H; p.OW = W; p.creal = Cout;
    return dtype == MSSEG_F32 ? launch_flat<float, SRC_DECONV_BWD, EPI_STORE>(p, (hipStream_t)stream)
                              : launch_flat<bf16_t, SRC_DECONV_BWD, EPI_STORE>(p, (hipStream_t)stream);
}

// ---- flat (1x1x1-tap) input-gradient kernels with the InstanceNorm-backward sums of the receiving layer fused into the
// epilogue, as msseg_conv3d_k3_dgrad_inbwd: the grid is tiled per sample (N samples of S voxels) so that a tile never
// straddles two samples.
static int flat_inbwd_common(IgemmParams& p, int N, long long S, int Cout, const void* yraw, long long ldyraw,
                             const void* act, long long ldact, const float* fwd_stats, float slope, float eps, float* red,
                             float* dgamma, float* dbeta, int accumulate, void* scratch, size_t scratch_bytes, int esz,
                             const char* who) {
    if (!yraw || !act || !fwd_stats || !red) MSSEG_FAIL(MSSEG_EINVAL, "%s: null pointer", who);
    if ((dgamma == nullptr) != (dbeta == nullptr)) MSSEG_FAIL(MSSEG_EINVAL, "%s: dgamma/dbeta go together", who);
    if (N < 1 || N > MSSEG_STATS_NMAX || S < 1 || S > 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "%s: 1 <= N <= %d samples", who, MSSEG_STATS_NMAX);
    if (Cout % 4 || (ldyraw % 4) || (ldact % 4) || ((uintptr_t)yraw % (4 * esz)) || ((uintptr_t)act % (4 * esz)))
        MSSEG_FAIL(MSSEG_EINVAL, "%s: channel count / strides must be multiples of 4", who);
    if (Cout > 64 * 16) MSSEG_FAIL(MSSEG_EINVAL, "%s: at most 1024 output channels", who);
    if (!scratch || ((uintptr_t)scratch & 255) || scratch_bytes < msseg_reduce_scratch_bytes())
        MSSEG_FAIL(MSSEG_EWORKSPACE, "%s: needs a scratch of %zu bytes", who, msseg_reduce_scratch_bytes());
    p.N = N; p.D = 1; p.H = 1; p.W = (int)S;
    p.stats = red;
    p.counter = (unsigned int*)scratch;
    p.stats_ws = (float*)((unsigned char*)scratch + MSSEG_SCRATCH_COUNTER_BYTES);
    p.nb_y = yraw; p.nb_ldy = ldyraw; p.nb_a = act; p.nb_lda = ldact; p.nb_stats = fwd_stats;
    p.nb_slope = slope; p.nb_eps = eps; p.nb_S = S;
    p.nb_dgamma = dgamma; p.nb_dbeta = dbeta; p.nb_acc = accumulate;
    return MSSEG_OK;
}

int msseg_conv3d_k1_dgrad_inbwd(const void* dy, long long lddy, const void* wp, void* da, long long ldda, int N,
                                long long S, int Cin, int Cout, const void* yraw, long long ldyraw, const void* act,
                                long long ldact, const float* fwd_stats, float slope, float eps, float* red,
                                float* dgamma, float* dbeta, int accumulate, void* scratch, size_t scratch_bytes,
                                int dtype, msseg_stream_t stream) {
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    int rc = check_common(dy, lddy, wp, da, ldda, dtype, esz);
    if (rc) return rc;
    if (Cin < 1 || Cout < 1 || Cin % (16 / esz)) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_dgrad_inbwd: bad channel counts");
    IgemmParams p{};
    p.x = dy; p.ldx = lddy; p.wp = wp; p.bias = nullptr; p.y = da; p.ldy = ldda;
    p.K = Cin; p.M = Cout;
    rc = flat_inbwd_common(p, N, S, Cout, yraw, ldyraw, act, ldact, fwd_stats, slope, eps, red, dgamma, dbeta, accumulate,
                           scratch, scratch_bytes, esz, "conv3d_k1_dgrad_inbwd");
    if (rc) return rc;
    return dtype == MSSEG_F32 ? launch_flat<float, SRC_DIRECT, EPI_STORE>(p, (hipStream_t)stream)
                              : launch_flat<bf16_t, SRC_DIRECT, EPI_STORE>(p, (hipStream_t)stream);
}

int msseg_deconv_k2s2_bwd_fused(const void* dy, long long lddy, const void* wp, void* dx, long long lddx, int N,
                                int D, int H, int W, int Cin, int Cout, const void* yraw, long long ldyraw,
                                const void* act, long long ldact, const float* fwd_stats, float slope, float eps,
                                float* red, float* dgamma, float* dbeta, int accumulate, float* dbias, int dbias_accumulate,
                                void* scratch, size_t scratch_bytes, int dtype, msseg_stream_t stream) {
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    int rc = check_common(dy, lddy, wp, dx, lddx, dtype, esz);
    if (rc) return rc;
    if (Cout % (16 / esz)) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2_bwd_fused: Cout %% %d must be 0", 16 / esz);
    const long long S = (long long)D * H * W;
    if ((long long)N * S > 0x7fffffffLL / 8) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2_bwd_fused: bad voxel count");
    if (yraw) {
        if (!act || !fwd_stats || !red) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2_bwd_fused: yraw needs act, fwd_stats and red");
        if (N > MSSEG_STATS_NMAX) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2_bwd_fused: fused sums need N <= %d", MSSEG_STATS_NMAX);
    }
    if (msseg_deconv2_fast_eligible(dtype, Cin, Cout, dx, lddx, dy, lddy, nullptr) &&
        (!yraw || ((ldyraw % 4) == 0 && (ldact % 4) == 0 && ((uintptr_t)yraw & 7) == 0 && ((uintptr_t)act & 7) == 0)))
        return msseg_deconv2_bwd_launch(dy, lddy, wp, dx, lddx, N, D, H, W, Cin, Cout, yraw, ldyraw, act, ldact, fwd_stats,
                                        slope, eps, red, dgamma, dbeta, accumulate, dbias, dbias_accumulate, scratch,
                                        scratch_bytes, (hipStream_t)stream);
    // generic path: the implicit-GEMM input gradient (with or without the fused sums) + a channel-sum pass for the bias
    if (yraw) {
        IgemmParams p{};
        p.x = dy; p.ldx = lddy; p.wp = wp; p.bias = nullptr; p.y = dx; p.ldy = lddx;
        p.K = 8 * Cout; p.M = Cin;
        p.OD = D; p.OH = H; p.OW = W; p.creal = Cout;
        rc = flat_inbwd_common(p, N, S, Cin, yraw, ldyraw, act, ldact, fwd_stats, slope, eps, red, dgamma, dbeta, accumulate,
                               scratch, scratch_bytes, esz, "deconv_k2s2_bwd_fused");
        if (rc) return rc;
        rc = dtype == MSSEG_F32 ? launch_flat<float, SRC_DECONV_BWD, EPI_STORE>(p, (hipStream_t)stream)
                                : launch_flat<bf16_t, SRC_DECONV_BWD, EPI_STORE>(p, (hipStream_t)stream);
    } else {
        rc = msseg_deconv_k2s2_bwd_data(dy, lddy, wp, dx, lddx, N, D, H, W, Cin, Cout, dtype, stream);
    }
    if (rc) return rc;
    if (dbias)
        return msseg_channel_sum(dy, lddy, dbias, 8 * (long long)N * S, Cout, dbias_accumulate, scratch, scratch_bytes, dtype,
                                 stream);
    return MSSEG_OK;
}

int msseg_deconv_k2s2_bwd_data_inbwd(const void* dy, long long lddy, const void* wp, void* dx, long long lddx, int N,
                                     int D, int H, int W, int Cin, int Cout, const void* yraw, long long ldyraw,
                                     const void* act, long long ldact, const float* fwd_stats, float slope, float eps,
                                     float* red, float* dgamma, float* dbeta, int accumulate, void* scratch,
                                     size_t scratch_bytes, int dtype, msseg_stream_t stream) {
    if (!yraw || !act || !fwd_stats || !red) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2_bwd_data_inbwd: null pointer");
    return msseg_deconv_k2s2_bwd_fused(dy, lddy, wp, dx, lddx, N, D, H, W, Cin, Cout, yraw, ldyraw, act, ldact, fwd_stats,
                                       slope, eps, red, dgamma, dbeta, accumulate, nullptr, 0, scratch, scratch_bytes, dtype,
                                       stream);
}

}  // extern "C"
