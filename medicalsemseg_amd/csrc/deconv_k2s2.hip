// ConvTranspose3d k = s = 2 (bf16) as register-resident-weight streaming kernels for gfx950.
//
// Replaces the transposed convolution of MONAI's UpCat / UnetrUpBlock (BasicUNet `upcat_i.upsample.deconv`,
// /root/reference/models/segmentors/swin_unetr.py:93-101 `transp_conv`) for the high-resolution levels, where the
// layer is pure bandwidth: 2 * voxels * Cin * 8 * Cout FLOPs against a write of 8x the input voxels.
//
// Non-overlapping 2x2x2 transposed convolution = one 1x1 GEMM per coarse voxel: out[child abc][co] = W[ci][co][abc] . x[ci].
// The whole weight tensor (8 * Cout * Cin <= 16 K elements) is held as MFMA A-operand fragments in registers for the
// life of the kernel; a wave takes 16 consecutive coarse voxels of one W-row, loads its B operand straight from global
// memory (the operand layout "8 channels of voxel r per lane quarter" IS the channels-last row), issues 8 * Cout/16 *
// Cin/32 MFMAs, and
//   forward : transposes the 8 children through a wave-private LDS tile into the four fine rows they form
//             (32 consecutive fine voxels each) and writes them as fully coalesced 16-byte stores;
//   backward: reads the 8 children of every coarse voxel (the B operand again, no staging), writes dx, and accumulates
//             in registers the InstanceNorm-backward sums of the layer that receives dx and the bias gradient
//             (sum of dy), so the separate statistics / channel-sum passes over the fine tensor disappear.
// The generic implicit-GEMM path (igemm_fwd.hip, 1x1 GEMM + pixel-shuffle scatter: 64-byte segments at 128-byte
// stride, input re-read once per child) stays for fp32 and for shapes whose weights do not fit the register file.
#include "k3pp.h"

namespace {

constexpr int DC_THREADS = 256;

struct Dc2Params {
    const void* x; long long ldx;      // coarse [N, D, H, W, Cin]   (forward input / backward output dx)
    const void* wp;                    // packed image (forward: M = 8*Cout, K = Cin; backward: M = Cin, K = 8*Cout)
    const float* bias;
    void* y; long long ldy;            // fine [N, 2D, 2H, 2W, Cout] (forward output / backward input dy)
    int N, D, H, W, Cin, Cout;
    // backward extras
    const void* nb_y; long long nb_ldy;    // raw conv output of the layer whose activation is the coarse tensor
    const void* nb_a; long long nb_lda;    // its stored activation (sign of the pre-activation)
    float nb_slope;
    float* stats_ws;                   // partial rows for msseg_k3_stats_finalize: [cin block][gx * N][N * 32 * 2]
    float* bias_ws;                    // partial rows [gx * N][Cout]
};

MSSEG_DEVFN u32x4_t ldg16(const void* p) { return *(const u32x4_t*)p; }

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <int KS, int NH>   // KS = Cin / 32 k-steps, NH = Cout / 16 cout tiles
__global__ __launch_bounds__(DC_THREADS, 2) void deconv2_fwd_kernel(const Dc2Params p) {
    constexpr int COUT = NH * 16;
    constexpr int RSB = COUT * 2 + 16;                  // LDS bytes per fine voxel (16-byte pad: fewer write conflicts)
    constexpr int TILE_B = 4 * 32 * RSB;                // 4 fine rows x 32 fine voxels
    constexpr int CPV = COUT * 2 / 16;                  // 16-byte chunks per fine voxel
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * TILE_B];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    unsigned char* tile = lds + wave * TILE_B;
    const bf16_t* __restrict__ xg = (const bf16_t*)p.x;
    bf16_t* __restrict__ yg = (bf16_t*)p.y;

    // weights -> registers: fragment (abc, j, k) = rows abc*COUT + j*16 .. +16 of the [8*COUT][Cin] matrix, k-step k
    u32x4_t af[8][NH][KS];
#pragma unroll
    for (int abc = 0; abc < 8; ++abc)
#pragma unroll
        for (int j = 0; j < NH; ++j)
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int m0 = abc * COUT + j * 16;
                const int cb = m0 >> 5, row = (m0 & 31) + r;
                af[abc][j][k] = ldg16((const unsigned char*)p.wp + ((((long long)cb * KS + k) * 4 + q) * 32 + row) * 16);
            }
    f32x4_t bv[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        bv[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv[j] = *(const f32x4_t*)(p.bias + j * 16 + q * 4);
    }
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
            bx[k] = valid ? ldg16(xg + cvox * p.ldx + k * 32 + q * 8) : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
        for (int abc = 0; abc < 8; ++abc) {
#pragma unroll
            for (int j = 0; j < NH; ++j) {
                f32x4_t acc = bv[j];
#pragma unroll
                for (int k = 0; k < KS; ++k) mma_chunk<bf16_t>(acc, af[abc][j][k], bx[k]);
                const bf16x4_t o = {(bf16_t)acc[0], (bf16_t)acc[1], (bf16_t)acc[2], (bf16_t)acc[3]};
                *(bf16x4_t*)(tile + ((abc >> 1) * 32 + 2 * r + (abc & 1)) * RSB + (j * 16 + q * 4) * 2) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: LDS ops of one wave execute in order
        const int nfv = 2 * ((p.W - w0) < 16 ? (p.W - w0) : 16);   // valid fine voxels of this segment
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            const long long frow = (((long long)n * 2 * p.D + 2 * d + (ab >> 1)) * 2 * p.H + 2 * h + (ab & 1)) * 2 * p.W + 2 * w0;
#pragma unroll
            for (int it = 0; it < (32 * CPV) / 64; ++it) {
                const int ch = it * 64 + lane;
                const int fv = ch / CPV, part = ch % CPV;
                const u32x4_t v = *(const u32x4_t*)(tile + (ab * 32 + fv) * RSB + part * 16);
                if (fv < nfv) *(u32x4_t*)(yg + (frow + fv) * p.ldy + part * 8) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next group's writes
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward-data (+ InstanceNorm-backward sums of the receiving layer, + bias gradient)
// ---------------------------------------------------------------------------------------------------------
template <int KC, int NH, bool INBWD, bool DBIAS>   // KC = Cout / 32, NH = Cin / 16
__global__ __launch_bounds__(DC_THREADS, 2) void deconv2_bwd_kernel(const Dc2Params p) {
    constexpr int CIN = NH * 16, COUT = KC * 32, NKB = 8 * KC;
    __shared__ float red[4][CIN * 2 + COUT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int n = blockIdx.y;
    const bf16_t* __restrict__ dyg = (const bf16_t*)p.y;
    bf16_t* __restrict__ dxg = (bf16_t*)p.x;

    // weights -> registers: fragment (jt, kb) = rows jt*16.. of the [Cin][8*COUT] matrix, k-step kb = abc * KC + kc
    u32x4_t af[NH][NKB];
#pragma unroll
    for (int jt = 0; jt < NH; ++jt)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int m0 = jt * 16;
            const int cb = m0 >> 5, row = (m0 & 31) + r;
            af[jt][kb] = ldg16((const unsigned char*)p.wp + ((((long long)cb * NKB + kb) * 4 + q) * 32 + row) * 16);
        }
    float s1[NH][4], s2[NH][4], bs[KC][8];
#pragma unroll
    for (int jt = 0; jt < NH; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[jt][e] = s2[jt][e] = 0.f;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[kc][e] = 0.f;

    const int GW = (p.W + 15) >> 4;
    const long long groups = (long long)p.D * p.H * GW;      // of this sample
    const long long wstride = (long long)gridDim.x * 4;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += wstride) {
        const int gw = (int)(g % GW);
        long long t = g / GW;
        const int h = (int)(t % p.H);
        const int d = (int)(t / p.H);
        const int w = gw * 16 + r;
        const bool valid = w < p.W;
        f32x4_t acc[NH];
#pragma unroll
        for (int jt = 0; jt < NH; ++jt) acc[jt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int abc = 0; abc < 8; ++abc) {
            const long long fvox = (((long long)n * 2 * p.D + 2 * d + (abc >> 2)) * 2 * p.H + 2 * h + ((abc >> 1) & 1)) * 2 * p.W +
                                   2 * w + (abc & 1);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const u32x4_t b = valid ? ldg16(dyg + fvox * p.ldy + kc * 32 + q * 8) : u32x4_t{0u, 0u, 0u, 0u};
                if constexpr (DBIAS) {
                    const bf16x8_t b8 = __builtin_bit_cast(bf16x8_t, b);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bs[kc][e] += (float)b8[e];
                }
#pragma unroll
                for (int jt = 0; jt < NH; ++jt) mma_chunk<bf16_t>(acc[jt], af[jt][abc * KC + kc], b);
            }
        }
        const long long cvox = (((long long)n * p.D + d) * p.H + h) * p.W + w;
#pragma unroll
        for (int jt = 0; jt < NH; ++jt) {
            const bf16x4_t o = {(bf16_t)acc[jt][0], (bf16_t)acc[jt][1], (bf16_t)acc[jt][2], (bf16_t)acc[jt][3]};
            if (valid) {
                *(bf16x4_t*)(dxg + cvox * p.ldx + jt * 16 + q * 4) = o;
                if constexpr (INBWD) {
                    const bf16x4_t y4 = *(const bf16x4_t*)((const bf16_t*)p.nb_y + cvox * p.nb_ldy + jt * 16 + q * 4);
                    const bf16x4_t a4 = *(const bf16x4_t*)((const bf16_t*)p.nb_a + cvox * p.nb_lda + jt * 16 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float da = (float)o[e];
                        const float dz = (float)a4[e] > 0.f ? da : da * p.nb_slope;
                        s1[jt][e] += dz;
                        s2[jt][e] += dz * (float)y4[e];
                    }
                }
            }
        }
    }
    // ---- reductions: lanes (16 voxel columns, fixed butterfly) -> wave slot -> workgroup row (fixed order) ----
    if constexpr (INBWD) {
#pragma unroll
        for (int jt = 0; jt < NH; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = s1[jt][e], b = s2[jt][e];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                if (r == 0) {
                    red[wave][(jt * 16 + q * 4 + e) * 2 + 0] = a;
                    red[wave][(jt * 16 + q * 4 + e) * 2 + 1] = b;
                }
            }
    }
    if constexpr (DBIAS) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = bs[kc][e];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) a += __shfl_xor(a, o);
                if (r == 0) red[wave][CIN * 2 + kc * 32 + q * 8 + e] = a;
            }
    }
    __syncthreads();
    const int R = gridDim.x * p.N, row = n * gridDim.x + blockIdx.x;
    if constexpr (INBWD) {
        // rows in the layout msseg_k3_stats_finalize sums: [cin block][R][N][32][2]; only this sample's slice is non-zero
        const int L = p.N * 64;
        for (int i = threadIdx.x; i < (CIN / 32) * L; i += DC_THREADS) {
            const int b = i / L, rem = i % L, nn = rem / 64, cl2 = rem % 64;
            float v = 0.f;
            if (nn == n) {
                const int idx = (b * 32 + (cl2 >> 1)) * 2 + (cl2 & 1);
                v = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
            }
            p.stats_ws[((long long)b * R + row) * L + rem] = v;
        }
    }
    if constexpr (DBIAS) {
        for (int c = threadIdx.x; c < COUT; c += DC_THREADS)
            p.bias_ws[(long long)row * COUT + c] = (red[0][CIN * 2 + c] + red[1][CIN * 2 + c]) + (red[2][CIN * 2 + c] + red[3][CIN * 2 + c]);
    }
}

// out[c] (+)= sum of R rows (fixed order): float4 columns x row slots, slots added through LDS
__global__ __launch_bounds__(256) void dc2_bias_finalize_kernel(const float* rows, int R, int C, float* out, int accumulate) {
    __shared__ __attribute__((aligned(16))) float fin[256 * 4 + 256];
    block_rows_sum<256>(rows, R, C, fin);
    const float* tot = fin + 256 * 4;
    for (int c = threadIdx.x; c < C; c += 256) out[c] = accumulate ? out[c] + tot[c] : tot[c];
}

int grid_x(long long groups) {
    long long gx = (groups + 3) / 4;                        // 4 waves per workgroup, one group per wave at a time
    const long long cap = (long long)msseg_num_cus() * 4;   // persistent: a few workgroups per CU, weights loaded once each
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    return (int)gx;
}

}  // namespace

// ---- host interface (igemm_fwd.hip dispatches here when eligible) ----------------------------------------
bool msseg_deconv2_fast_eligible(int dtype, int Cin, int Cout, const void* coarse, long long ldc, const void* fine,
                                 long long ldf, const float* bias) {
    static const bool off = getenv("MSSEG_NO_DECONV_FAST") != nullptr;
    if (off || dtype != MSSEG_BF16) return false;
    if (!((Cin == 32 || Cin == 64) && Cout == 32)) return false;   // weight fragments must fit the register file
    if ((ldc % 8) || (ldf % 8) || ((uintptr_t)coarse & 15) || ((uintptr_t)fine & 15)) return false;
    if (bias && ((uintptr_t)bias & 15)) return false;
    return true;
}

int msseg_deconv2_fwd_launch(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy, int N,
                             int D, int H, int W, int Cin, int Cout, hipStream_t stream) {
    Dc2Params p{};
    p.x = x; p.ldx = ldx; p.wp = wp; p.bias = bias; p.y = y; p.ldy = ldy;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    const long long groups = (long long)N * D * H * ((W + 15) / 16);
    const int gx = grid_x(groups);
    if (Cin == 32) hipLaunchKernelGGL((deconv2_fwd_kernel<1, 2>), dim3(gx), dim3(DC_THREADS), 0, stream, p);
    else hipLaunchKernelGGL((deconv2_fwd_kernel<2, 2>), dim3(gx), dim3(DC_THREADS), 0, stream, p);
    MSSEG_CHECK_LAUNCH("deconv2_fwd");
    return MSSEG_OK;
}

// dx (+ red[N][Cin][2] InstanceNorm-backward sums when yraw != null, + dbias[Cout] when dbias != null).
// scratch: the zero-initialised reduce scratch (msseg_reduce_scratch_bytes()).
int msseg_deconv2_bwd_launch(const void* dy, long long lddy, const void* wp, void* dx, long long lddx, int N, int D, int H,
                             int W, int Cin, int Cout, const void* yraw, long long ldyraw, const void* act, long long ldact,
                             const float* fwd_stats, float slope, float eps, float* red, float* dgamma, float* dbeta,
                             int accumulate, float* dbias, int dbias_accumulate, void* scratch, size_t scratch_bytes,
                             hipStream_t stream) {
    Dc2Params p{};
    p.x = dx; p.ldx = lddx; p.wp = wp; p.y = (void*)dy; p.ldy = lddy;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.nb_y = yraw; p.nb_ldy = ldyraw; p.nb_a = act; p.nb_lda = ldact; p.nb_slope = slope;
    const long long groups = (long long)D * H * ((W + 15) / 16);
    long long gxl = (groups + 3) / 4;
    long long cap = (long long)msseg_num_cus() * 4 / N;
    if (cap < 1) cap = 1;
    if (gxl > cap) gxl = cap;
    const int gx = (int)(gxl < 1 ? 1 : gxl);
    const bool inbwd = yraw != nullptr, dbg = dbias != nullptr;
    const int R = gx * N, ncb = Cin / 32;
    const size_t need = MSSEG_SCRATCH_COUNTER_BYTES + ((size_t)ncb * R * N * 64 + (size_t)R * Cout) * 4;
    if ((inbwd || dbg) && (!scratch || scratch_bytes < need))
        MSSEG_FAIL(MSSEG_EWORKSPACE, "deconv2_bwd: scratch of %zu bytes needed", need);
    p.stats_ws = (float*)((unsigned char*)scratch + MSSEG_SCRATCH_COUNTER_BYTES);
    p.bias_ws = p.stats_ws + (size_t)ncb * R * N * 64;
    dim3 grid(gx, N);
#define DC2B(NH_) do {                                                                                                   \
        if (inbwd && dbg) hipLaunchKernelGGL((deconv2_bwd_kernel<1, NH_, true, true>), grid, dim3(DC_THREADS), 0, stream, p);   \
        else if (inbwd) hipLaunchKernelGGL((deconv2_bwd_kernel<1, NH_, true, false>), grid, dim3(DC_THREADS), 0, stream, p);    \
        else if (dbg) hipLaunchKernelGGL((deconv2_bwd_kernel<1, NH_, false, true>), grid, dim3(DC_THREADS), 0, stream, p);      \
        else hipLaunchKernelGGL((deconv2_bwd_kernel<1, NH_, false, false>), grid, dim3(DC_THREADS), 0, stream, p);              \
    } while (0)
    if (Cin == 32) DC2B(2); else DC2B(4);
#undef DC2B
    MSSEG_CHECK_LAUNCH("deconv2_bwd");
    if (dbg) {
        hipLaunchKernelGGL(dc2_bias_finalize_kernel, dim3(1), dim3(256), 0, stream, p.bias_ws, R, Cout, dbias, dbias_accumulate);
        MSSEG_CHECK_LAUNCH("deconv2_bias_finalize");
    }
    if (inbwd) {
        K3FinParams f{};
        f.ws = p.stats_ws; f.R = R; f.N = N; f.coutb = 32; f.M = Cin; f.stats = red;
        f.nb_stats = fwd_stats; f.nb_eps = eps; f.nb_S = (long long)D * H * W;
        f.nb_dgamma = dgamma; f.nb_dbeta = dbeta; f.nb_acc = accumulate;
        return msseg_k3_stats_finalize(f, ncb, stream);
    }
    return MSSEG_OK;
}
