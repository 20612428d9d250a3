// conv3d 3x3x3 (stride 1, pad 1) on channels-last bf16 for SMALL grids (the 24^3 .. 6^3 levels of the UNet):
// implicit GEMM with both MFMA operands fed straight from global memory / the caches -- no LDS image, no barrier.
//
// Replaces torch's Conv3d inside MONAI's TwoConv on the deep levels (BasicUNet down_2..down_4 / upcat_4..upcat_3,
// SURVEY.md A15) and their input gradients.  On those levels a layer is a small-M / large-K GEMM (M = 432 .. 27 648
// voxels, K = 27 * Cin = 1 728 .. 6 912): the tile kernels (igemm_fwd.hip) spend their time in
// "load stage -> LDS -> barrier -> 54 MFMAs" chains of a few hundred workgroups; this kernel instead gives every WAVE
// an independent output block of MT x 16 voxels by NT x 16 output channels and lets it stream its operands:
//   B (activations): lane (r, q) loads the 16 bytes [voxel r + tap][k-block, chunk q] -- the channels-last row IS the
//                    MFMA operand layout; out-of-volume taps are a per-lane predicate (27-bit mask, built once);
//   A (weights)    : lane (r, q) loads the 16 bytes [cout r][tap][k-block, chunk q] from the packed image
//                    (msseg_pack_weights) -- served by the vector cache / L2, shared by all waves of the cout tile.
// The whole input of these levels (<= 3.5 MB) lives in L2, so the 27-fold tap re-reads never reach HBM; with thousands
// of independent waves and no synchronisation the latency of one wave's loads hides under the others.
// Epilogue as in conv3d_k3_pp.hip: bias, bf16 store, InstanceNorm statistics or InstanceNorm-backward sums as partial
// rows for msseg_k3_stats_finalize (deterministic: fixed lane butterfly, waves added through LDS in order).
#include "k3pp.h"

#include <stdlib.h>

namespace {

constexpr int KD_WAVES = 4;                   // waves per workgroup

MSSEG_DEVFN u32x4_t ldg16(const void* p) { return *(const u32x4_t*)p; }

// STATS: 0 none, 1 forward statistics, 2 InstanceNorm-backward sums
template <int MT, int NT, int CH, int STATS>
__global__ __launch_bounds__(KD_WAVES * 64) void k3direct_kernel(const K3ppParams p, int cb, int groups_per_sample,
                                                                 int wg_per_sample) {
    __shared__ float red[KD_WAVES][NT * 16 * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int n = blockIdx.z;
    const int ct0 = blockIdx.y * NT;                       // first 16-wide cout tile of this workgroup
    const int S = p.D * p.H * p.W, HW = p.H * p.W;
    const int NKB = p.K >> 5;
    const bf16_t* __restrict__ xg = (const bf16_t*)p.x + (long long)n * S * p.ldx;
    bf16_t* __restrict__ yg = (bf16_t*)p.y + (long long)n * S * p.ldy;
    const unsigned char* __restrict__ wimg = (const unsigned char*)p.wp;

    // weight fragment base of cout tile j: [cout block][k block][tap][quarter][cout in block][16 B]
    long long wbase[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int m0 = (ct0 + j) * 16;
        const int cbk = m0 / cb, row = m0 % cb + r;
        wbase[j] = ((long long)cbk * NKB * 27 * 4 + q) * cb * 16 + row * 16;   // + ((kb * 27 + tap) * 4) * cb * 16
    }
    const long long wstep = (long long)4 * cb * 16;        // one (kb, tap) step
    f32x4_t bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bv[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv[j] = *(const f32x4_t*)(p.bias + (ct0 + j) * 16 + q * 4);
    }
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[j][e] = s2[j][e] = 0.f;

    // this wave's voxel groups: MT consecutive groups of 16 flat voxels, walked with stride (waves of the sample)
    const int wid = blockIdx.x * KD_WAVES + wave, nwaves = wg_per_sample * KD_WAVES;
    for (int g0 = wid * MT; g0 < groups_per_sample; g0 += nwaves * MT) {
        int vox[MT];
        unsigned mask[MT];                                  // bit t: tap t of this lane's voxel is inside the volume
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int v = (g0 + m) * 16 + r;
            const bool ok = (g0 + m) < groups_per_sample && v < S;
            const int vv = ok ? v : 0;
            const int d = vv / HW, rem = vv - d * HW, h = rem / p.W, w = rem - h * p.W;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const int dd = d + t / 9 - 1, hh = h + (t / 3) % 3 - 1, ww = w + t % 3 - 1;
                if ((unsigned)dd < (unsigned)p.D && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W) mk |= 1u << t;
            }
            vox[m] = vv;
            mask[m] = ok ? mk : 0u;
        }
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[m][j] = bv[j];
        // steps s = tap * NKB + kb in chunks of CH (CH divides NKB: a chunk stays inside one tap); the operands of chunk
        // c + 1 are in flight while chunk c feeds the matrix pipe (two register buffers, counted vmcnt waits)
        const int nchunks = 27 * NKB / CH;
        auto load_chunk = [&](int c, u32x4_t (&bfb)[CH][MT], u32x4_t (&afb)[CH][NT]) {
            const int s0 = c * CH;
            const int t = s0 / NKB, kb0 = s0 - t * NKB;
            const int toff = ((t / 9 - 1) * HW + ((t / 3) % 3 - 1) * p.W + (t % 3 - 1));
#pragma unroll
            for (int i = 0; i < CH; ++i) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    bfb[i][m] = ((mask[m] >> t) & 1u) ? ldg16(xg + (long long)(vox[m] + toff) * p.ldx + (kb0 + i) * 32 + q * 8)
                                                      : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
                for (int j = 0; j < NT; ++j) afb[i][j] = ldg16(wimg + wbase[j] + (long long)((kb0 + i) * 27 + t) * wstep);
            }
        };
        auto compute = [&](u32x4_t (&bfb)[CH][MT], u32x4_t (&afb)[CH][NT]) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int j = 0; j < NT; ++j) mma_chunk<bf16_t>(acc[m][j], afb[i][j], bfb[i][m]);
        };
        u32x4_t b0[CH][MT], a0[CH][NT], b1[CH][MT], a1[CH][NT];
        load_chunk(0, b0, a0);
        for (int c = 0; c < nchunks; c += 2) {
            if (c + 1 < nchunks) load_chunk(c + 1, b1, a1);
            compute(b0, a0);
            if (c + 2 < nchunks) load_chunk(c + 2, b0, a0);
            if (c + 1 < nchunks) compute(b1, a1);
        }
        // ---- epilogue ----
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int v = (g0 + m) * 16 + r;
            const bool ok = (g0 + m) < groups_per_sample && v < S;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = (ct0 + j) * 16 + q * 4;
                const bf16x4_t ob = {(bf16_t)acc[m][j][0], (bf16_t)acc[m][j][1], (bf16_t)acc[m][j][2], (bf16_t)acc[m][j][3]};
                if (ok) *(bf16x4_t*)(yg + (long long)v * p.ldy + co) = ob;
                if constexpr (STATS == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float rv = ok ? (float)ob[e] : 0.f;      // statistics of the tensor as stored
                        s1[j][e] += rv;
                        s2[j][e] += rv * rv;
                    }
                } else if constexpr (STATS == 2) {
                    if (ok) {
                        const long long gv = (long long)n * S + v;
                        const bf16x4_t y4 = *(const bf16x4_t*)((const bf16_t*)p.nb_y + gv * p.nb_ldy + co);
                        const bf16x4_t a4 = *(const bf16x4_t*)((const bf16_t*)p.nb_a + gv * p.nb_lda + co);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float da = (float)ob[e];
                            const float dz = (float)a4[e] > 0.f ? da : da * p.nb_slope;
                            s1[j][e] += dz;
                            s2[j][e] += dz * (float)y4[e];
                        }
                    }
                }
            }
        }
    }
    if constexpr (STATS != 0) {
        // lanes (16 voxel columns, fixed butterfly) -> wave slot -> workgroup row (waves added in order)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = s1[j][e], b = s2[j][e];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                if (r == 0) {
                    red[wave][(j * 16 + q * 4 + e) * 2 + 0] = a;
                    red[wave][(j * 16 + q * 4 + e) * 2 + 1] = b;
                }
            }
        __syncthreads();
        // row layout of msseg_k3_stats_finalize: [cout block = blockIdx.y][R = wg_per_sample * N][N][NT*16][2]; only this
        // sample's slice is non-zero
        constexpr int CBW = NT * 16;
        const int L = p.N * CBW * 2, R = wg_per_sample * p.N, row = n * wg_per_sample + blockIdx.x;
        float* dst = p.stats_ws + ((long long)blockIdx.y * R + row) * L;
        for (int i = threadIdx.x; i < L; i += KD_WAVES * 64) {
            const int nn = i / (CBW * 2), idx = i % (CBW * 2);
            float v = 0.f;
            if (nn == n) v = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
            dst[i] = v;
        }
    }
}

template <int MT, int NT, int CH>
int launch_ch(const K3ppParams& p, int cb, hipStream_t stream) {
    const int S = p.D * p.H * p.W;
    const int groups = (S + 15) / 16;
    int wg = (groups + MT * KD_WAVES - 1) / (MT * KD_WAVES);
    // persistent cap: beyond ~8 workgroups per CU in total the extra rows only lengthen the finalising step
    const int ncb = p.M / (NT * 16);
    const long long cap = (long long)msseg_num_cus() * 8 / ((long long)ncb * p.N) + 1;
    if (wg > cap) wg = (int)cap;
    dim3 grid(wg, ncb, p.N), block(KD_WAVES * 64);
    if (p.stats == nullptr) hipLaunchKernelGGL((k3direct_kernel<MT, NT, CH, 0>), grid, block, 0, stream, p, cb, groups, wg);
    else if (p.nb_y == nullptr) hipLaunchKernelGGL((k3direct_kernel<MT, NT, CH, 1>), grid, block, 0, stream, p, cb, groups, wg);
    else hipLaunchKernelGGL((k3direct_kernel<MT, NT, CH, 2>), grid, block, 0, stream, p, cb, groups, wg);
    MSSEG_CHECK_LAUNCH("conv3d_k3_direct");
    if (p.stats != nullptr) {
        K3FinParams f{};
        f.ws = p.stats_ws; f.R = wg * p.N; f.N = p.N; f.coutb = NT * 16; f.M = p.M; f.stats = p.stats;
        f.nb_stats = p.nb_y ? p.nb_stats : nullptr; f.nb_eps = p.nb_eps; f.nb_S = p.nb_S;
        f.nb_dgamma = p.nb_dgamma; f.nb_dbeta = p.nb_dbeta; f.nb_acc = p.nb_acc;
        return msseg_k3_stats_finalize(f, ncb, stream);
    }
    return MSSEG_OK;
}

template <int MT, int NT>
int launch_mn(const K3ppParams& p, int cb, hipStream_t stream) {
    const int nkb = p.K / 32;
    if (nkb % 4 == 0) return launch_ch<MT, NT, 4>(p, cb, stream);
    if (nkb % 2 == 0) return launch_ch<MT, NT, 2>(p, cb, stream);
    return launch_ch<MT, NT, 1>(p, cb, stream);
}

}  // namespace

// voxel-count window in which the direct kernel is used (per launch, all samples): below ~64 K voxels the tile kernels
// are chains of latency-bound stages; above it their LDS reuse wins.  MSSEG_K3DIRECT_MAX overrides (0 disables).
static long long k3direct_max_voxels() {
    static const long long v = [] {
        const char* e = getenv("MSSEG_K3DIRECT_MAX");
        return e ? atoll(e) : 8192LL;
    }();
    return v;
}

bool msseg_k3direct_eligible(const K3ppParams& p) {
    const long long nv = (long long)p.N * p.D * p.H * p.W;
    if (nv > k3direct_max_voxels()) return false;
    if ((p.K % 32) || (p.M % 16) || p.M < 16) return false;
    if ((p.ldx % 8) || (p.ldy % 4) || ((uintptr_t)p.x & 15) || ((uintptr_t)p.y & 7)) return false;
    if (p.bias && ((uintptr_t)p.bias & 15)) return false;
    if (p.stats && (p.N > MSSEG_STATS_NMAX || p.M > 1024)) return false;
    if (p.nb_y && ((p.nb_ldy % 4) || (p.nb_lda % 4) || ((uintptr_t)p.nb_y & 7) || ((uintptr_t)p.nb_a & 7))) return false;
    if ((long long)p.D * p.H * p.W * (p.ldx > p.ldy ? p.ldx : p.ldy) >= 0x7fffffffLL) return false;
    return p.N <= 65535;
}

// cb: cout-block width of the packed weight image (16 or 32)
int msseg_k3direct_launch(const K3ppParams& p, int cb, hipStream_t stream) {
    if (cb != 16 && cb != 32 && cb != 48) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_direct: bad cout block %d", cb);
    const long long nv = (long long)p.N * p.D * p.H * p.W;
    const long long tiles = p.M / 16;
    // block shape by available parallelism: one 16 x 16 block per wave on the tiniest grids, 32 x 32 when there are
    // enough (voxel group, cout tile) pairs to fill the chip several times over
    const long long blocks16 = ((nv + 15) / 16) * tiles;
    const long long want = (long long)msseg_num_cus() * 4 * 2;
    if ((tiles % 2) == 0 && blocks16 / 4 >= want) return launch_mn<2, 2>(p, cb, stream);
    if ((tiles % 2) == 0 && blocks16 / 2 >= want) return launch_mn<1, 2>(p, cb, stream);
    return launch_mn<1, 1>(p, cb, stream);
}
