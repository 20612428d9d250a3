// ConvTranspose3d k = s = 2 (bf16) for the channel counts the register-resident kernels of deconv_k2s2.hip do not hold:
// the transposed convolutions of Swin-UNETR's UnetrUpBlocks (/root/reference/models/segmentors/swin_unetr.py:93-128:
// 768 -> 384 @3^3, 384 -> 192 @6^3, 192 -> 96 @12^3, 96 -> 48 @24^3, 48 -> 48 @48^3) and the deep UpCat levels of MONAI
// BasicUNet (256 -> 128 @6^3, 128 -> 64 @12^3).  The generic implicit-GEMM path ran them as flat GEMMs through LDS-staged
// 256-voxel tiles, one 32-channel stage after the other (forward 52 us, input gradient 55-97 us per launch for 0.1-2 GFLOP;
// five + five launches = 0.64 ms of the Swin-UNETR step).
//
// Same scheme as deconv_k2s2.hip / linear_regw.hip -- a wave takes 16 coarse voxels, its B operand comes straight from
// global memory (a channels-last row IS the operand layout), the weights it needs live in registers -- with the output
// sliced over grid.y so that any channel count fits the register file:
//   forward   slice = (fine row pair ab, cout slice): the two children (ab, c = 0 / 1) of 16 coarse voxels form ONE fine row
//             segment of 32 voxels, transposed through a wave-private LDS tile into coalesced 16-byte stores.
//   backward  dx[v][ci] = sum over the 8 children and Cout: K = 8 * Cout is split over the FOUR WAVES of a workgroup (every
//             wave gathers its k-steps from the child rows -- a 16-byte chunk never straddles two children), the partial
//             tiles meet in LDS in a fixed order (deterministic); slice = Cin tiles.  Output: bf16 dx, or the fp32 block
//             layout of the small-grid finish kernels (conv3d_k3_small.hip: [Cin / 4][voxel][4]), whose k3s_bwd_unit_kernel
//             then runs the receiving conv + InstanceNorm + LeakyReLU unit's whole backward in one launch.
// Bound: HBM on the 96^3 / 48^3 outputs (8x the input voxels written / read once), launch latency below.
#include "k3pp.h"

#include <stdlib.h>

namespace {

constexpr int DG_THREADS = 256;

struct DcgParams {
    const void* x; long long ldx;      // coarse [N, D, H, W, Cin]   (forward input / backward output dx)
    const void* wp;                    // packed image (forward: M = 8*Cout, K = Cin; backward: M = Cin, K = 8*Cout)
    const float* bias;
    void* y; long long ldy;            // fine [N, 2D, 2H, 2W, Cout] (forward output / backward input dy)
    int N, D, H, W, Cin, Cout;
    int cb;                            // cout block width of the image
    float* part;                       // backward: fp32 [Cin / 4][N*D*H*W][4] instead of bf16 dx
};

MSSEG_DEVFN u32x4_t ldg16(const void* p) { return *(const u32x4_t*)p; }

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <int KS, int NH, int CH>   // KS = ceil(Cin / 32) k-steps, NH cout tiles per slice, CH children per slice (2 or 1)
__global__ __launch_bounds__(DG_THREADS, 2) void dcg_fwd_kernel(const DcgParams p) {
    constexpr int CS = NH * 16;                         // channels of a slice
    constexpr int RSB = CS * 2 + 16;                    // LDS bytes per fine voxel (16-byte pad: fewer write conflicts)
    constexpr int FV = 16 * CH;                         // fine voxels of a segment that this slice writes
    constexpr int TILE_B = FV * RSB;
    constexpr int CPV = CS * 2 / 16;                    // 16-byte chunks per fine voxel
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * TILE_B];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    unsigned char* tile = lds + wave * TILE_B;
    const bf16_t* __restrict__ xg = (const bf16_t*)p.x;
    const int nsl = p.Cout / CS;
    const int cg = blockIdx.y / nsl, js = blockIdx.y - cg * nsl;
    const int abc0 = cg * CH;                           // first child of the slice; CH == 2: (abc0, abc0 + 1) share a fine row
    const int cbase = js * CS;
    bf16_t* __restrict__ yg = (bf16_t*)p.y + cbase;

    u32x4_t af[CH][NH][KS];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < NH; ++j)
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int m0 = (abc0 + c) * p.Cout + cbase + j * 16;
                const int blk = m0 / p.cb, row = m0 - blk * p.cb + r;
                af[c][j][k] = ldg16((const unsigned char*)p.wp + ((((long long)blk * KS + k) * 4 + q) * p.cb + row) * 16);
            }
    f32x4_t bv[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        bv[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv[j] = *(const f32x4_t*)(p.bias + cbase + j * 16 + q * 4);
    }
    bool kok[KS];                                       // this lane's chunk of k-step k lies inside the row
#pragma unroll
    for (int k = 0; k < KS; ++k) kok[k] = k * 32 + q * 8 < p.Cin;

    const int GW = (p.W + 15) >> 4;
    const long long groups = (long long)p.N * p.D * p.H * GW;
    const long long wstride = (long long)gridDim.x * 4;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += wstride) {
        const int gw = (int)(g % GW);
        long long t = g / GW;
        const int h = (int)(t % p.H); t /= p.H;
        const int d = (int)(t % p.D);
        const int n = (int)(t / p.D);
        const int w0 = gw * 16, w = w0 + r;
        const bool valid = w < p.W;
        const long long cvox = (((long long)n * p.D + d) * p.H + h) * p.W + w;
        u32x4_t bx[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k)
            bx[k] = (valid && kok[k]) ? ldg16(xg + cvox * p.ldx + k * 32 + q * 8) : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int j = 0; j < NH; ++j) {
                f32x4_t acc = bv[j];
#pragma unroll
                for (int k = 0; k < KS; ++k) mma_chunk<bf16_t>(acc, af[c][j][k], bx[k]);
                const bf16x4_t o = {(bf16_t)acc[0], (bf16_t)acc[1], (bf16_t)acc[2], (bf16_t)acc[3]};
                *(bf16x4_t*)(tile + (CH * r + c) * RSB + (j * 16 + q * 4) * 2) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: LDS ops of one wave execute in order
        const int ncv = (p.W - w0) < 16 ? (p.W - w0) : 16;   // valid coarse voxels of this segment
        const long long frow = (((long long)n * 2 * p.D + 2 * d + (abc0 >> 2)) * 2 * p.H + 2 * h + ((abc0 >> 1) & 1)) * 2 * p.W + 2 * w0;
#pragma unroll
        for (int it = 0; it < (FV * CPV + 63) / 64; ++it) {
            const int ch = it * 64 + lane;
            const int fv = ch / CPV, part = ch - fv * CPV;
            if (ch < FV * CPV && fv < CH * ncv) {
                const u32x4_t v = *(const u32x4_t*)(tile + fv * RSB + part * 16);
                const long long fw = CH == 2 ? fv : 2 * fv + (abc0 & 1);
                *(u32x4_t*)(yg + (frow + fw) * p.ldy + part * 8) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next group's writes
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward-data: the four waves split K = 8 * Cout
// ---------------------------------------------------------------------------------------------------------
template <int KSW, int NH, bool PART>   // KSW = Cout / 16 k-steps per wave, NH Cin tiles per slice
__global__ __launch_bounds__(DG_THREADS, 2) void dcg_bwd_kernel(const DcgParams p) {
    __shared__ __attribute__((aligned(16))) float xch[3][NH][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int mbase = blockIdx.y * NH * 16;
    const int kstot = 4 * KSW;                          // k-steps of the packed image (K = 8 * Cout = 128 * KSW)
    const int k0 = wave * KSW;
    const bf16_t* __restrict__ dyg = (const bf16_t*)p.y;
    const long long NV = (long long)p.N * p.D * p.H * p.W;

    u32x4_t af[NH][KSW];
#pragma unroll
    for (int j = 0; j < NH; ++j)
#pragma unroll
        for (int k = 0; k < KSW; ++k) {
            const int m0 = mbase + j * 16;
            const int blk = m0 / p.cb, row = m0 - blk * p.cb + r;
            af[j][k] = ldg16((const unsigned char*)p.wp + ((((long long)blk * kstot + k0 + k) * 4 + q) * p.cb + row) * 16);
        }
    int off[KSW];                                       // this lane's chunk of k-step k0 + k, relative to fine voxel (2d, 2h, 2w)
#pragma unroll
    for (int k = 0; k < KSW; ++k) {
        const int c0 = (k0 + k) * 32 + q * 8;
        const int abc = c0 / p.Cout, co = c0 - abc * p.Cout;
        off[k] = (int)((((long long)(abc >> 2) * 2 * p.H + ((abc >> 1) & 1)) * 2 * p.W + (abc & 1)) * p.ldy) + co;
    }
    const long long groups = (NV + 15) >> 4;
    for (long long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long long v = g * 16 + r;
        const bool valid = v < NV;
        long long t = valid ? v : 0;
        const int w = (int)(t % p.W); t /= p.W;
        const int h = (int)(t % p.H); t /= p.H;
        const int d = (int)(t % p.D);
        const long long n = t / p.D;
        const long long fbase = ((((n * 2 * p.D + 2 * d) * 2 * p.H + 2 * h) * 2 * p.W) + 2 * w) * p.ldy;
        u32x4_t bx[KSW];
#pragma unroll
        for (int k = 0; k < KSW; ++k) bx[k] = valid ? ldg16(dyg + fbase + off[k]) : u32x4_t{0u, 0u, 0u, 0u};
        f32x4_t acc[NH];
#pragma unroll
        for (int j = 0; j < NH; ++j) {
            acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KSW; ++k) mma_chunk<bf16_t>(acc[j], af[j][k], bx[k]);
        }
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < NH; ++j) *(f32x4_t*)xch[wave - 1][j][lane] = acc[j];
        }
        __syncthreads();
        if (wave == 0 && valid) {
#pragma unroll
            for (int j = 0; j < NH; ++j) {
                const f32x4_t a1 = *(const f32x4_t*)xch[0][j][lane], a2 = *(const f32x4_t*)xch[1][j][lane], a3 = *(const f32x4_t*)xch[2][j][lane];
                const f32x4_t o = (acc[j] + a1) + (a2 + a3);
                const int m = mbase + j * 16 + q * 4;
                if constexpr (PART) {
                    *(f32x4_t*)(p.part + ((long long)(m >> 2) * NV + v) * 4) = o;
                } else {
                    const bf16x4_t ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                    *(bf16x4_t*)((bf16_t*)p.x + v * p.ldx + m) = ob;
                }
            }
        }
        __syncthreads();                                // wave 0 has read the exchange tiles before the next group's writes
    }
}

// forward instantiation for a shape: k-steps, cout tiles per slice, children per slice (weight fragments per lane <= 24)
struct FwdCfg { int ks, nh, ch; };
bool fwd_cfg(int Cin, int Cout, FwdCfg* c) {
    const int ks = (Cin + 31) / 32;
    static const FwdCfg table[] = {{2, 3, 2}, {3, 3, 2}, {4, 2, 2}, {6, 2, 2}, {8, 1, 2}, {12, 1, 2}, {24, 1, 1}};
    for (const FwdCfg& t : table)
        if (t.ks == ks && Cout % (t.nh * 16) == 0) { *c = t; return true; }
    return false;
}

// backward instantiation: k-steps per wave (Cout / 16), Cin tiles per slice
struct BwdCfg { int ksw, nh; };
bool bwd_cfg(int Cin, int Cout, BwdCfg* c) {
    if (Cout % 16 || Cin % 16) return false;
    static const BwdCfg table[] = {{3, 3}, {6, 4}, {12, 2}, {24, 1}, {8, 2}, {4, 4}};
    for (const BwdCfg& t : table)
        if (t.ksw == Cout / 16 && (Cin / 16) % t.nh == 0) { *c = t; return true; }
    return false;
}

bool common_ok(int dtype, const void* coarse, long long ldc, const void* fine, long long ldf, int Cin, int Cout) {
    static const bool off = getenv("MSSEG_NO_DECONV_GEN") != nullptr;   // A/B switch
    if (off || dtype != MSSEG_BF16) return false;
    if ((ldc % 8) || (ldf % 8) || ((uintptr_t)coarse & 15) || ((uintptr_t)fine & 15) || ldc < Cin || ldf < Cout) return false;
    return true;
}

}  // namespace

bool msseg_deconv2g_fwd_eligible(int dtype, int Cin, int Cout, const void* coarse, long long ldc, const void* fine,
                                 long long ldf, const float* bias) {
    FwdCfg c;
    if (!common_ok(dtype, coarse, ldc, fine, ldf, Cin, Cout) || Cin % 8 || Cout % 16) return false;
    if (bias && ((uintptr_t)bias & 15)) return false;
    return fwd_cfg(Cin, Cout, &c);
}

int msseg_deconv2g_fwd_launch(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy, int N,
                              int D, int H, int W, int Cin, int Cout, hipStream_t stream) {
    FwdCfg c;
    if (!fwd_cfg(Cin, Cout, &c)) MSSEG_FAIL(MSSEG_EINVAL, "deconv2g_fwd: shape %d -> %d has no instantiation", Cin, Cout);
    DcgParams p{};
    p.x = x; p.ldx = ldx; p.wp = wp; p.bias = bias; p.y = y; p.ldy = ldy;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.cb = msseg_cout_block(8 * Cout);
    const int slices = (8 / c.ch) * (Cout / (c.nh * 16));
    const long long groups = (long long)N * D * H * ((W + 15) / 16);
    long long gx = (groups + 3) / 4;
    long long cap = (long long)msseg_num_cus() * 4 / slices;
    if (cap < 1) cap = 1;
    if (gx > cap) gx = cap;
    dim3 grid((unsigned)gx, (unsigned)slices);
#define DCG_FWD(KS_, NH_, CH_)                                                                                         \
    if (c.ks == KS_ && c.nh == NH_ && c.ch == CH_) {                                                                   \
        MSSEG_KTIMED("dcg_fwd_kernel", stream,                                                                          \
                     hipLaunchKernelGGL((dcg_fwd_kernel<KS_, NH_, CH_>), grid, dim3(DG_THREADS), 0, stream, p));        \
    } else
    DCG_FWD(2, 3, 2) DCG_FWD(3, 3, 2) DCG_FWD(4, 2, 2) DCG_FWD(6, 2, 2) DCG_FWD(8, 1, 2) DCG_FWD(12, 1, 2) DCG_FWD(24, 1, 1) {}
#undef DCG_FWD
    MSSEG_CHECK_LAUNCH("deconv2g_fwd");
    return MSSEG_OK;
}

bool msseg_deconv2g_bwd_eligible(int dtype, int Cin, int Cout, const void* coarse, long long ldc, const void* fine,
                                 long long ldf) {
    BwdCfg c;
    if (coarse == nullptr) {   // partial-sum form: no coarse tensor
        static const bool off = getenv("MSSEG_NO_DECONV_GEN") != nullptr;
        if (off || dtype != MSSEG_BF16 || (ldf % 8) || ((uintptr_t)fine & 15) || ldf < Cout) return false;
    } else if (!common_ok(dtype, coarse, ldc, fine, ldf, Cin, Cout) || (ldc % 4)) {
        return false;
    }
    return bwd_cfg(Cin, Cout, &c);
}

// dx (bf16, part == nullptr) or part[Cin / 4][N*D*H*W][4] (fp32) = the input gradient of ConvTranspose3d k2 s2
int msseg_deconv2g_bwd_launch(const void* dy, long long lddy, const void* wp, void* dx, long long lddx, float* part, int N,
                              int D, int H, int W, int Cin, int Cout, hipStream_t stream) {
    BwdCfg c;
    if (!bwd_cfg(Cin, Cout, &c)) MSSEG_FAIL(MSSEG_EINVAL, "deconv2g_bwd: shape %d -> %d has no instantiation", Cin, Cout);
    if ((long long)(2 * H) * (2 * W) * 2 * lddy > 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "deconv2g_bwd: fine plane too large");
    DcgParams p{};
    p.x = dx; p.ldx = lddx; p.wp = wp; p.y = (void*)dy; p.ldy = lddy; p.part = part;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.cb = msseg_cout_block(Cin);
    const int slices = Cin / (c.nh * 16);
    const long long groups = ((long long)N * D * H * W + 15) / 16;
    long long gx = groups;
    long long cap = (long long)msseg_num_cus() * 8 / slices;   // persistent above two rounds of two workgroups per CU
    if (cap < 1) cap = 1;
    if (gx > cap) gx = cap;
    dim3 grid((unsigned)gx, (unsigned)slices);
#define DCG_BWD(KSW_, NH_)                                                                                                     \
    if (c.ksw == KSW_ && c.nh == NH_) {                                                                                        \
        if (part) {                                                                                                            \
            MSSEG_KTIMED("dcg_bwd_kernel", stream,                                                                              \
                         hipLaunchKernelGGL((dcg_bwd_kernel<KSW_, NH_, true>), grid, dim3(DG_THREADS), 0, stream, p));         \
        } else {                                                                                                               \
            MSSEG_KTIMED("dcg_bwd_kernel", stream,                                                                              \
                         hipLaunchKernelGGL((dcg_bwd_kernel<KSW_, NH_, false>), grid, dim3(DG_THREADS), 0, stream, p));        \
        }                                                                                                                      \
    } else
    DCG_BWD(3, 3) DCG_BWD(6, 4) DCG_BWD(12, 2) DCG_BWD(24, 1) DCG_BWD(8, 2) DCG_BWD(4, 4) {}
#undef DCG_BWD
    MSSEG_CHECK_LAUNCH("deconv2g_bwd");
    return MSSEG_OK;
}

extern "C" {

/* 1 when msseg_deconv_k2s2_bwd_partials takes the shape (bf16, channel counts with an instantiation) */
int msseg_deconv_k2s2_bwd_partials_ok(int Cin, int Cout, int dtype) {
    BwdCfg c;
    static const bool off = getenv("MSSEG_NO_DECONV_GEN") != nullptr;
    return (!off && dtype == MSSEG_BF16 && bwd_cfg(Cin, Cout, &c) && Cin % 32 == 0) ? 1 : 0;
}

/* input gradient of ConvTranspose3d k2 s2 as ONE fp32 stage group in the layout of msseg_conv3d_k3_small_partials
 * (part[Cin / 4][N*D*H*W][4]): msseg_conv3d_k3_small_bwd_finish(part, 1, ...) then stores it, or runs the whole backward
 * of the conv + InstanceNorm + LeakyReLU unit whose activation the transposed conv read.  dy: fine [N, 2D, 2H, 2W, Cout];
 * wp: the backward image of msseg_pack_weights (M = Cin, K = 8 * Cout). */
int msseg_deconv_k2s2_bwd_partials(const void* dy, long long lddy, const void* wp, float* part, size_t part_bytes, int N,
                                   int D, int H, int W, int Cin, int Cout, int dtype, msseg_stream_t stream) {
    if (!dy || !wp || !part) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2_bwd_partials: null pointer");
    if (!msseg_deconv_k2s2_bwd_partials_ok(Cin, Cout, dtype) || !msseg_deconv2g_bwd_eligible(dtype, Cin, Cout, nullptr, 0, dy, lddy))
        MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2_bwd_partials: shape %d -> %d / alignment not supported", Cin, Cout);
    const long long NV = (long long)N * D * H * W;
    if (N < 1 || D < 1 || H < 1 || W < 1 || NV > 0x7fffffffLL / 8) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2_bwd_partials: bad voxel count");
    if (((uintptr_t)wp & 15) || ((uintptr_t)part & 15)) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2_bwd_partials: 16-byte alignment");
    if (part_bytes < (size_t)NV * Cin * sizeof(float))
        MSSEG_FAIL(MSSEG_EWORKSPACE, "deconv_k2s2_bwd_partials: workspace %zu B < %zu B", part_bytes, (size_t)NV * Cin * sizeof(float));
    return msseg_deconv2g_bwd_launch(dy, lddy, wp, nullptr, 0, part, N, D, H, W, Cin, Cout, (hipStream_t)stream);
}

}  // extern "C"
