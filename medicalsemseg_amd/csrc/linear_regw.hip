// Linear / 1x1x1 convolution on many tokens with few channels (bf16): register-resident-weight streaming kernel for gfx950.
//
//   y[v][m] = sum_k W[m][k] * x[v][k] + b[m]         v over N*D*H*W voxels (tokens), channels-last rows
//
// Replaces nn.Linear in the Swin stages -- qkv / proj / fc1 / fc2 of
// /root/reference/models/backbones/swin_nnformer.py:24-42,128-196: 221 k tokens x 48 ... 192 channels in the first stage,
// 27 k x 96 ... 384, 3.4 k x 192 ... 768, 432 x 384 below -- and its input gradient (the same GEMM on the transposed weight
// image).  The first stage is pure bandwidth (85-106 MB in + out for 1-2 GFLOP), the deeper ones are launch-sized; the
// generic implicit-GEMM kernel (igemm_fwd.hip: LDS-staged 256-voxel tiles, one 32-channel stage after the other on one
// wave per SIMD) ran the former at about 1 TB/s and the latter at a 30 us floor.  Same scheme as deconv_k2s2.hip: the
// weights of a workgroup's output slice (<= 24 MFMA A fragments per lane) live in registers for the life of the kernel;
// a wave takes 16 consecutive tokens, loads its B operand straight from global memory (the channels-last row IS the
// operand layout: 8 channels of token r per lane quarter), issues NH x KS MFMAs, transposes the 16 x M outputs through a
// wave-private LDS tile and writes the rows as coalesced 16-byte stores.  grid.y slices the output channels: workgroup
// (x, s) computes channels s * NH * 16 .. of its tokens, so that a wide layer's weights still fit the register file and
// a short token list still fills the chip.
//
// Fused epilogues for the MLP of a Swin block (fc1 -> GELU -> fc2, /root/reference/models/backbones/swin_nnformer.py:24-42),
// applied to the 16-byte output chunks on their way from the LDS tile to memory:
//   EPI_GELU      y = the pre-activation, y2 = gelu(y as stored): fc1 + GELU forward without the separate pass that re-reads it
//   EPI_GELU_BWD  y = (dy W) as stored * gelu'(aux): the input gradient of fc2 straight into the gradient of fc1's output
//   EPI_ADD       y = aux + (x W + b) as stored: the residual add behind the attention projection / fc2 of a Swin block
//                 (swin_nnformer.py:243-262) without the separate add pass
// with the arithmetic of gelu_kernel (attention.hip) on the bf16-rounded operands, i.e. bit-identical to the unfused chain.
#include "k3pp.h"

#include <stdlib.h>

namespace {

constexpr int LR_THREADS = 256;

struct LinParams {
    const void* x; long long ldx;
    const void* wp;                  // msseg_pack_weights image, T = 1: [cout block][k block][quarter][cout][16 B]
    const float* bias;
    void* y; long long ldy;
    long long NV;
    int K, cb;                       // logical input channels, cout block width of the image
    void* y2; long long ldy2;        // EPI_GELU: the activation
    const void* aux; long long ldaux;   // EPI_GELU_BWD: the pre-activation of the layer that receives this gradient
};

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_GELU_BWD = 2, EPI_ADD = 3 };

MSSEG_DEVFN float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
MSSEG_DEVFN float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
MSSEG_DEVFN float gelu_cdf(float v) { return 0.5f * (1.f + erff(v * 0.70710678118654752f)); }

MSSEG_DEVFN u32x4_t ldg16(const void* p) { return *(const u32x4_t*)p; }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
MSSEG_DEVFN unsigned pack_bf16x2(float a, float b) {
    const bf16x2_t v = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(unsigned, v);
}

template <int KS, int NH, int EPI>   // KS = ceil(K / 32) k-steps, NH = M / 16 output tiles
__global__ __launch_bounds__(LR_THREADS, 2) void linear_regw_kernel(const LinParams p) {
    constexpr int M = NH * 16;
    constexpr int RSB = M * 2 + 16;                    // LDS bytes per token row (16-byte pad: fewer write conflicts)
    constexpr int CPV = M * 2 / 16;                    // 16-byte chunks per output row
    constexpr int TILE_B = 16 * RSB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * TILE_B];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    unsigned char* tile = lds + wave * TILE_B;
    const int mbase = blockIdx.y * M;
    const bf16_t* __restrict__ xg = (const bf16_t*)p.x;
    bf16_t* __restrict__ yg = (bf16_t*)p.y + mbase;

    u32x4_t af[NH][KS];
#pragma unroll
    for (int j = 0; j < NH; ++j)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int m0 = mbase + j * 16;
            const int blk = m0 / p.cb, row = m0 - blk * p.cb + r;
            af[j][k] = ldg16((const unsigned char*)p.wp + ((((long long)blk * KS + k) * 4 + q) * p.cb + row) * 16);
        }
    f32x4_t bv[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        bv[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv[j] = *(const f32x4_t*)(p.bias + mbase + j * 16 + q * 4);
    }
    bool kok[KS];                                       // this lane's chunk of k-step k lies inside the row
#pragma unroll
    for (int k = 0; k < KS; ++k) kok[k] = k * 32 + q * 8 < p.K;

    const long long groups = (p.NV + 15) >> 4;
    const long long wstride = (long long)gridDim.x * 4;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += wstride) {
        const long long v0 = g * 16, v = v0 + r;
        const bool valid = v < p.NV;
        u32x4_t bx[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k)
            bx[k] = (valid && kok[k]) ? ldg16(xg + v * p.ldx + k * 32 + q * 8) : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < NH; ++j) {
            f32x4_t acc = bv[j];
#pragma unroll
            for (int k = 0; k < KS; ++k) mma_chunk<bf16_t>(acc, af[j][k], bx[k]);
            const u32x2_t o = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3])};
            *(u32x2_t*)(tile + r * RSB + (j * 16 + q * 4) * 2) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: LDS ops of one wave execute in order
        const int nv = (p.NV - v0) < 16 ? (int)(p.NV - v0) : 16;
#pragma unroll
        for (int it = 0; it < (16 * CPV + 63) / 64; ++it) {
            const int ch = it * 64 + lane;
            const int vv = ch / CPV, part = ch - vv * CPV;
            if (ch < 16 * CPV && vv < nv) {
                u32x4_t o = *(const u32x4_t*)(tile + vv * RSB + part * 16);
                if constexpr (EPI == EPI_GELU_BWD) {
                    const u32x4_t a = *(const u32x4_t*)((const bf16_t*)p.aux + mbase + (v0 + vv) * p.ldaux + part * 8);
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float a0 = bf16_lo(a[w]), a1 = bf16_hi(a[w]);
                        const float g0 = bf16_lo(o[w]) * (gelu_cdf(a0) + a0 * 0.3989422804014327f * expf(-0.5f * a0 * a0));
                        const float g1 = bf16_hi(o[w]) * (gelu_cdf(a1) + a1 * 0.3989422804014327f * expf(-0.5f * a1 * a1));
                        o[w] = pack_bf16x2(g0, g1);
                    }
                }
                if constexpr (EPI == EPI_ADD) {   // the unfused chain's arithmetic: the Linear output rounded to bf16, then the add
                    const u32x4_t a = *(const u32x4_t*)((const bf16_t*)p.aux + mbase + (v0 + vv) * p.ldaux + part * 8);
#pragma unroll
                    for (int w = 0; w < 4; ++w) o[w] = pack_bf16x2(bf16_lo(a[w]) + bf16_lo(o[w]), bf16_hi(a[w]) + bf16_hi(o[w]));
                }
                *(u32x4_t*)(yg + (v0 + vv) * p.ldy + part * 8) = o;
                if constexpr (EPI == EPI_GELU) {
                    u32x4_t h;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float a0 = bf16_lo(o[w]), a1 = bf16_hi(o[w]);
                        h[w] = pack_bf16x2(a0 * gelu_cdf(a0), a1 * gelu_cdf(a1));
                    }
                    *(u32x4_t*)((bf16_t*)p.y2 + mbase + (v0 + vv) * p.ldy2 + part * 8) = h;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next group's writes
    }
}

template <int KS, int NH, int EPI = EPI_NONE> int launch(const LinParams& p, int slices, hipStream_t stream) {
    const long long groups = (p.NV + 15) >> 4;
    long long gx = (groups + 3) / 4;
    long long cap = (long long)msseg_num_cus() * 2 / slices;
    if (cap < 1) cap = 1;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL((linear_regw_kernel<KS, NH, EPI>), dim3((unsigned)gx, (unsigned)slices), dim3(LR_THREADS), 0, stream, p);
    MSSEG_CHECK_LAUNCH("linear_regw");
    return MSSEG_OK;
}

// ---- few tokens, deep K (the last Swin stage: 432 tokens x 1152 ... 1536 input channels; MONAI variant: 54 x 3072) ----
// The streaming kernel above cannot hold K / 32 > 24 weight fragments per output tile, and the generic flat kernel walks K in 32-
// channel stages behind a load -> LDS -> barrier round trip each on two dozen workgroups: 48-64 us for 0.5 GFLOP.  Here the four
// waves of a workgroup split K (KSW k-steps each) for the same 16 tokens x NH output tiles, every wave holds its NH x KSW weight
// fragments and its KSW token chunks in registers at once (all loads in flight together), and the four partial tiles meet in
// LDS (fixed order, deterministic).  Grid = token groups x output slices: hundreds of workgroups for a one-round launch.
template <int KSW, int NH>
__global__ __launch_bounds__(LR_THREADS, 2) void linear_ksplit_kernel(const LinParams p) {
    __shared__ __attribute__((aligned(16))) float xch[3][NH][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int mbase = blockIdx.y * NH * 16;
    const int kstot = (p.K + 31) / 32;              // k-steps of the packed image
    const int k0 = wave * KSW;
    const long long v = (long long)blockIdx.x * 16 + r;
    const bool valid = v < p.NV;
    const bf16_t* __restrict__ xg = (const bf16_t*)p.x;
    u32x4_t bx[KSW];
#pragma unroll
    for (int k = 0; k < KSW; ++k)
        bx[k] = valid ? ldg16(xg + v * p.ldx + (k0 + k) * 32 + q * 8) : u32x4_t{0u, 0u, 0u, 0u};
    f32x4_t acc[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int m0 = mbase + j * 16;
        const int blk = m0 / p.cb, row = m0 - blk * p.cb + r;
        u32x4_t af[KSW];
#pragma unroll
        for (int k = 0; k < KSW; ++k)
            af[k] = ldg16((const unsigned char*)p.wp + ((((long long)blk * kstot + k0 + k) * 4 + q) * p.cb + row) * 16);
        acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (wave == 0 && p.bias) acc[j] = *(const f32x4_t*)(p.bias + m0 + q * 4);
#pragma unroll
        for (int k = 0; k < KSW; ++k) mma_chunk<bf16_t>(acc[j], af[k], bx[k]);
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < NH; ++j) *(f32x4_t*)xch[wave - 1][j][lane] = acc[j];
    }
    __syncthreads();
    if (wave == 0 && valid) {
        bf16_t* __restrict__ yg = (bf16_t*)p.y + v * p.ldy + mbase + q * 4;
#pragma unroll
        for (int j = 0; j < NH; ++j) {
            const f32x4_t a1 = *(const f32x4_t*)xch[0][j][lane], a2 = *(const f32x4_t*)xch[1][j][lane], a3 = *(const f32x4_t*)xch[2][j][lane];
            const f32x4_t o = (acc[j] + a1) + (a2 + a3);
            *(u32x2_t*)(yg + j * 16) = u32x2_t{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        }
    }
}

// k-steps per wave with an instantiation: K = 128 * KSW input channels
static bool ksplit_ok(int Cin, int Cout, long long NV) {
    if (Cin % 128 || Cout % 16 || NV > 4096) return false;
    const int ksw = Cin / 128;
    return ksw == 9 || ksw == 12 || ksw == 24;
}

template <int KSW> int launch_ksplit(const LinParams& p, int Cout, hipStream_t stream) {
    const unsigned gx = (unsigned)((p.NV + 15) / 16);
    if (KSW <= 12 && Cout % 32 == 0) {
        hipLaunchKernelGGL((linear_ksplit_kernel<KSW, 2>), dim3(gx, (unsigned)(Cout / 32)), dim3(LR_THREADS), 0, stream, p);
    } else {
        hipLaunchKernelGGL((linear_ksplit_kernel<KSW, 1>), dim3(gx, (unsigned)(Cout / 16)), dim3(LR_THREADS), 0, stream, p);
    }
    MSSEG_CHECK_LAUNCH("linear_ksplit");
    return MSSEG_OK;
}

constexpr int MAX_FRAGS = 24;   // weight fragments per lane (96 VGPRs)

// output tiles per workgroup: the widest of {12, 9, 6, 4, 3, 2, 1} that divides the layer and fits the register budget
int pick_nh(int ks, int Cout) {
    static const int cand[] = {12, 9, 6, 4, 3, 2, 1};
    for (int nh : cand)
        if (Cout % (nh * 16) == 0 && ks * nh <= MAX_FRAGS) return nh;
    return 0;
}

template <int KS, int EPI = EPI_NONE> int launch_ks(const LinParams& p, int nh, int slices, hipStream_t stream) {
    switch (nh) {
        case 1: return launch<KS, 1, EPI>(p, slices, stream);
        case 2: if constexpr (KS * 2 <= MAX_FRAGS) return launch<KS, 2, EPI>(p, slices, stream); break;
        case 3: if constexpr (KS * 3 <= MAX_FRAGS) return launch<KS, 3, EPI>(p, slices, stream); break;
        case 4: if constexpr (KS * 4 <= MAX_FRAGS) return launch<KS, 4, EPI>(p, slices, stream); break;
        case 6: if constexpr (KS * 6 <= MAX_FRAGS) return launch<KS, 6, EPI>(p, slices, stream); break;
        case 9: if constexpr (KS * 9 <= MAX_FRAGS) return launch<KS, 9, EPI>(p, slices, stream); break;
        case 12: if constexpr (KS * 12 <= MAX_FRAGS) return launch<KS, 12, EPI>(p, slices, stream); break;
    }
    MSSEG_FAIL(MSSEG_EINVAL, "linear_regw: no instantiation for %d k-steps x %d output tiles", KS, nh);
}

}  // namespace

// k-step counts with an instantiation (input widths 33 ... 64, 65 ... 96, 129 ... 192, 257 ... 288, 353 ... 384, 545 ... 576,
// 737 ... 768 and the 144-wide qkv gradient): the Swin stages at width 48 and its multiples; everything else stays generic
static bool lr_ks_ok(int ks) { return ks == 2 || ks == 3 || ks == 5 || ks == 6 || ks == 9 || ks == 12 || ks == 18 || ks == 24; }

bool msseg_linear_regw_eligible(int dtype, long long NV, int Cin, int Cout, const void* x, long long ldx, const void* y,
                                long long ldy, const float* bias) {
    static const bool off = getenv("MSSEG_NO_LINEAR_REGW") != nullptr;   // A/B switch
    if (off || dtype != MSSEG_BF16 || NV < 1 || Cout % 16 || Cin % 8) return false;
    const int ks = (Cin + 31) / 32;
    if ((!lr_ks_ok(ks) || pick_nh(ks, Cout) == 0) && !ksplit_ok(Cin, Cout, NV)) return false;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || (ldx % 8) || (ldy % 8) || ldx < Cin || ldy < Cout) return false;
    if (bias && ((uintptr_t)bias & 15)) return false;
    return true;
}

int msseg_linear_regw_launch(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                             long long NV, int Cin, int Cout, hipStream_t stream) {
    LinParams p{};
    p.x = x; p.ldx = ldx; p.wp = wp; p.bias = bias; p.y = y; p.ldy = ldy; p.NV = NV; p.K = Cin;
    p.cb = msseg_cout_block(Cout);
    const int ks = (Cin + 31) / 32, nh = pick_nh(ks, Cout);
    if (!lr_ks_ok(ks) || nh == 0) {
        if (!ksplit_ok(Cin, Cout, NV)) MSSEG_FAIL(MSSEG_EINVAL, "linear_regw: shape %d -> %d has no instantiation", Cin, Cout);
        switch (Cin / 128) {
            case 9: return launch_ksplit<9>(p, Cout, stream);
            case 12: return launch_ksplit<12>(p, Cout, stream);
            default: return launch_ksplit<24>(p, Cout, stream);
        }
    }
    const int slices = Cout / (nh * 16);
    switch (ks) {
        case 2: return launch_ks<2>(p, nh, slices, stream);
        case 3: return launch_ks<3>(p, nh, slices, stream);
        case 5: return launch_ks<5>(p, nh, slices, stream);
        case 6: return launch_ks<6>(p, nh, slices, stream);
        case 9: return launch_ks<9>(p, nh, slices, stream);
        case 12: return launch_ks<12>(p, nh, slices, stream);
        case 18: return launch_ks<18>(p, nh, slices, stream);
        default: return launch_ks<24>(p, nh, slices, stream);
    }
}

// ---- fused GELU epilogues: the (k-steps, output tiles) pairs of fc1 / fc2's input gradient at the Swin widths
// (C -> 4C for C = 48, 96, 192, 384); everything else takes the unfused chain (msseg_linear_gelu_ok() == 0)
static bool lr_gelu_shape(int Cin, int Cout, int* ks_out, int* nh_out) {
    const int ks = (Cin + 31) / 32;
    if (!lr_ks_ok(ks)) return false;
    const int nh = pick_nh(ks, Cout);
    const bool ok = (ks == 2 && nh == 12) || (ks == 3 && nh == 6) || (ks == 6 && nh == 4) || (ks == 12 && nh == 2);
    if (ks_out) { *ks_out = ks; *nh_out = nh; }
    return ok;
}

template <int EPI> static int launch_gelu(const LinParams& p, int ks, int nh, int Cout, hipStream_t stream) {
    const int slices = Cout / (nh * 16);
    if (ks == 2) return launch<2, 12, EPI>(p, slices, stream);
    if (ks == 3) return launch<3, 6, EPI>(p, slices, stream);
    if (ks == 6) return launch<6, 4, EPI>(p, slices, stream);
    return launch<12, 2, EPI>(p, slices, stream);
}

extern "C" {

int msseg_linear_gelu_ok(long long NV, int Cin, int Cout, int dtype) {
    static const bool off = getenv("MSSEG_NO_LINEAR_GELU") != nullptr || getenv("MSSEG_NO_LINEAR_REGW") != nullptr;   // A/B switch
    if (off || dtype != MSSEG_BF16 || NV < 1 || Cout % 16 || Cin % 8) return 0;
    return lr_gelu_shape(Cin, Cout, nullptr, nullptr) ? 1 : 0;
}

static int lin_check(const void* x, long long ldx, const void* wp, const void* y, long long ldy, const void* z, long long ldz,
                     const float* bias, long long NV, int Cin, int Cout, int dtype, const char* who) {
    if (!x || !wp || !y || !z) MSSEG_FAIL(MSSEG_EINVAL, "%s: null pointer", who);
    if (!msseg_linear_gelu_ok(NV, Cin, Cout, dtype)) MSSEG_FAIL(MSSEG_EINVAL, "%s: shape %d -> %d not supported (msseg_linear_gelu_ok)", who, Cin, Cout);
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)z & 15) || ((uintptr_t)wp & 15) || (ldx % 8) || (ldy % 8) || (ldz % 8) ||
        ldx < Cin || ldy < Cout || ldz < Cout || (bias && ((uintptr_t)bias & 15)))
        MSSEG_FAIL(MSSEG_EINVAL, "%s: operands must be 16-byte aligned with strides that are multiples of 8 elements", who);
    return MSSEG_OK;
}

int msseg_linear_gelu_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* pre, long long ldpre, void* act,
                          long long ldact, long long NV, int Cin, int Cout, int dtype, msseg_stream_t stream) {
    if (int rc = lin_check(x, ldx, wp, pre, ldpre, act, ldact, bias, NV, Cin, Cout, dtype, "linear_gelu_fwd")) return rc;
    LinParams p{};
    p.x = x; p.ldx = ldx; p.wp = wp; p.bias = bias; p.y = pre; p.ldy = ldpre; p.y2 = act; p.ldy2 = ldact; p.NV = NV; p.K = Cin;
    p.cb = msseg_cout_block(Cout);
    int ks, nh;
    lr_gelu_shape(Cin, Cout, &ks, &nh);
    return launch_gelu<EPI_GELU>(p, ks, nh, Cout, (hipStream_t)stream);
}

/* y = res + (x W^T + b): nn.Linear with the residual add of a Swin block in its epilogue (the sum of the bf16-rounded Linear
 * output and res, as the unfused chain forms it).  msseg_linear_add_ok() == 1 for the widths of the register-resident-weight
 * kernel (Cin <= 768 at the Swin widths); callers keep Linear + add otherwise. */
int msseg_linear_add_ok(long long NV, int Cin, int Cout, int dtype) {
    static const bool off = getenv("MSSEG_NO_LINEAR_ADD") != nullptr || getenv("MSSEG_NO_LINEAR_REGW") != nullptr;   // A/B switch
    if (off || dtype != MSSEG_BF16 || NV < 1 || Cout % 16 || Cin % 8) return 0;
    const int ks = (Cin + 31) / 32;
    return (lr_ks_ok(ks) && pick_nh(ks, Cout) != 0) ? 1 : 0;
}

int msseg_linear_add_fwd(const void* x, long long ldx, const void* wp, const float* bias, const void* res, long long ldres, void* y,
                         long long ldy, long long NV, int Cin, int Cout, int dtype, msseg_stream_t stream) {
    if (!x || !wp || !y || !res) MSSEG_FAIL(MSSEG_EINVAL, "linear_add_fwd: null pointer");
    if (!msseg_linear_add_ok(NV, Cin, Cout, dtype)) MSSEG_FAIL(MSSEG_EINVAL, "linear_add_fwd: shape %d -> %d not supported (msseg_linear_add_ok)", Cin, Cout);
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)res & 15) || ((uintptr_t)wp & 15) || (ldx % 8) || (ldy % 8) || (ldres % 8) ||
        ldx < Cin || ldy < Cout || ldres < Cout || (bias && ((uintptr_t)bias & 15)))
        MSSEG_FAIL(MSSEG_EINVAL, "linear_add_fwd: operands must be 16-byte aligned with strides that are multiples of 8 elements");
    LinParams p{};
    p.x = x; p.ldx = ldx; p.wp = wp; p.bias = bias; p.y = y; p.ldy = ldy; p.aux = res; p.ldaux = ldres; p.NV = NV; p.K = Cin;
    p.cb = msseg_cout_block(Cout);
    const int ks = (Cin + 31) / 32, nh = pick_nh(ks, Cout), slices = Cout / (nh * 16);
    hipStream_t st = (hipStream_t)stream;
    switch (ks) {
        case 2: return launch_ks<2, EPI_ADD>(p, nh, slices, st);
        case 3: return launch_ks<3, EPI_ADD>(p, nh, slices, st);
        case 5: return launch_ks<5, EPI_ADD>(p, nh, slices, st);
        case 6: return launch_ks<6, EPI_ADD>(p, nh, slices, st);
        case 9: return launch_ks<9, EPI_ADD>(p, nh, slices, st);
        case 12: return launch_ks<12, EPI_ADD>(p, nh, slices, st);
        case 18: return launch_ks<18, EPI_ADD>(p, nh, slices, st);
        default: return launch_ks<24, EPI_ADD>(p, nh, slices, st);
    }
}

int msseg_linear_gelu_bwd(const void* dy, long long lddy, const void* wp, const void* pre, long long ldpre, void* dpre,
                          long long lddpre, long long NV, int Cin, int Cout, int dtype, msseg_stream_t stream) {
    if (int rc = lin_check(dy, lddy, wp, dpre, lddpre, pre, ldpre, nullptr, NV, Cin, Cout, dtype, "linear_gelu_bwd")) return rc;
    LinParams p{};
    p.x = dy; p.ldx = lddy; p.wp = wp; p.y = dpre; p.ldy = lddpre; p.aux = pre; p.ldaux = ldpre; p.NV = NV; p.K = Cin;
    p.cb = msseg_cout_block(Cout);
    int ks, nh;
    lr_gelu_shape(Cin, Cout, &ks, &nh);
    return launch_gelu<EPI_GELU_BWD>(p, ks, nh, Cout, (hipStream_t)stream);
}

}  // extern "C"
