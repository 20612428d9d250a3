// conv3d 3x3x3 (stride 1, pad 1), bf16, on SMALL grids: the 12^3 and 6^3 levels of the UNet (64 ... 256 channels, 432 ...
// 3456 voxels per batch).  Replaces torch Conv3d + InstanceNorm3d + LeakyReLU (+ MaxPool3d) of MONAI BasicUNet's TwoConv /
// Down blocks at those levels (BASELINE.json configs 1-3) and their backward.
//
// Why a kernel of its own.  On these grids a layer is a few GFLOP over tensors that live in L2 / Infinity Cache, and the
// tile kernels (igemm_fwd.hip) run it as a CHAIN per workgroup: for each 32-channel stage load 55 KB of weights + the halo,
// commit to LDS, barrier, <= 108 MFMAs per wave -- 2.3 us per stage whatever the tile, one wave per SIMD, 14 ... 38 us per
// launch for 1 ... 6 GFLOP (profiles/README.md, round 2).  Here the channel stages are split over WORKGROUPS instead
// (split-K): grid = (3x6x6-voxel tiles) x (32-wide cout blocks) x (groups of 32-channel stages), a workgroup does ONE
// stage (two on the widest layers, so that the grid stays within one round) of ONE tile -- one load round trip, 108 MFMAs
// per wave, one fp32 partial block out; 76 KB of LDS lets two workgroups share a CU, one loading while the other
// multiplies -- and the whole layer is a single round of workgroups over the chip.  The partial blocks are combined by the
// follow-up kernel that the layer needs anyway:
//
//   k3s_fwd_finish   one workgroup per (sample, 4 channels): sums the stages in a fixed order, adds the bias, rounds to
//                    bf16 (the raw conv output the backward needs), forms the InstanceNorm statistics of the whole
//                    (sample, channel) slab INSIDE the workgroup (no partial rows, no finalize launch), normalises,
//                    applies LeakyReLU, writes the activation (into a concat buffer slice if asked) and its 2x2x2
//                    max-pool.  conv + finish = 2 launches for what took 3-4 (conv, statistics finalize, normalise[, pool]).
//   k3s_bwd_finish   input-gradient form: sums the stages; either stores the gradient (plain), or -- when the conv's input
//                    is the activation of a conv + InstanceNorm + LeakyReLU unit -- goes on to that unit's backward in the
//                    same workgroup: dz = da * lrelu'(.), the two slab sums, dy = gamma * rstd * (dz - mean(dz) - xhat *
//                    mean(dz * xhat)), dgamma / dbeta.  dgrad + finish = 2 launches for 3 (dgrad with fused sums, finalize,
//                    apply).
//
// Layouts: activations channels-last bf16 [N][D][H][W][ld]; weights = the packed MFMA images of msseg_pack_weights with
// cout block 32 ([cout block][channel stage][tap][quarter][cout][16 B]: one (block, stage) slice is 55,296 contiguous
// bytes); partials fp32 [stage group][M / 4][N * D * H * W][4] (channel-group major: the finish kernels read contiguous rows).  D a multiple of 3, H and W of 6; Cin, Cout multiples of 32.
// Bound: latency (one round of workgroups); the MFMA phase is LDS-read bound (one fragment read per MFMA, the 6-wide tile
// rows on an 8-wide halo cost a 2-way bank conflict on part of the voxel-operand reads).
#include "common.h"

#include <stdlib.h>

namespace {

constexpr int TS = 6;                       // tile edge (h, w)
constexpr int TSD = 3;                      // tile depth
constexpr int TV = TSD * TS * TS;           // 108 voxels
constexpr int NFRAG = (TV + 15) / 16;       // 7 voxel fragments (the last holds 12)
constexpr int HS = TS + 2;                  // halo edge (h, w)
constexpr int HSD = TSD + 2;
constexpr int HV = HSD * HS * HS;           // 320 halo voxels
constexpr int W_BYTES = 27 * 4 * 32 * 16;   // 55,296: one (cout block, stage) slice of the packed image
constexpr int A_PLANE = HV * 16;            // 5,120: one channel quarter of the halo (a multiple of 256)
constexpr int K3S_LDS = W_BYTES + 4 * A_PLANE;   // 75,776: two workgroups per CU
constexpr int A_ITERS = (HV * 4 + 255) / 256;

struct K3sParams {
    const bf16_t* x; long long ldx;
    const bf16_t* wp;
    float* part;
    int N, D, H, W, K, M;
    int td, th, tw;       // tiles per axis
    int kpw;              // 32-channel stages per workgroup
};

__global__ __launch_bounds__(256, 2) void k3s_kernel(const K3sParams p) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    unsigned char* ldsW = smem;
    unsigned char* ldsA = smem + W_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    int t = blockIdx.x;
    const int tw0 = (t % p.tw) * TS; t /= p.tw;
    const int th0 = (t % p.th) * TS; t /= p.th;
    const int td0 = (t % p.td) * TSD;
    const int n = t / p.td;
    const int cb = blockIdx.y, kg = blockIdx.z, nks = p.K >> 5;

    // this wave's voxel fragments: 7 fragments of 16 tile voxels, waves take 2, 2, 2, 1
    const int f0 = wave * 2;
    const int nf = wave < 3 ? 2 : 1;
    int abase[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        int v = (f0 + f) * 16 + r;
        v = v < TV ? v : TV - 1;                     // padding rows read a valid voxel, their results are dropped
        const int vw = v % TS, vh = (v / TS) % TS, vd = v / (TS * TS);
        abase[f] = q * A_PLANE + ((vd * HS + vh) * HS + vw) * 16;
    }
    const int bbase = (q * 32 + r) * 16;
    f32x4_t acc[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f) acc[f][0] = acc[f][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int si = 0; si < p.kpw; ++si) {
        const int ks = kg * p.kpw + si;
        if (ks >= nks) break;
        // ---- stage the weight slice and the halo (zero outside the volume); all loads of a thread in flight together
        {
            const u32x4_t* wsrc = (const u32x4_t*)(p.wp + ((long long)(cb * nks + ks)) * (W_BYTES / 2));
            u32x4_t wreg[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                const int idx = tid + i * 256;
                wreg[i] = idx < W_BYTES / 16 ? wsrc[idx] : u32x4_t{0, 0, 0, 0};
            }
            u32x4_t areg[A_ITERS];
#pragma unroll
            for (int i = 0; i < A_ITERS; ++i) {
                const int idx = tid + i * 256;           // [halo voxel][quarter]
                const int hq = idx & 3, hv = idx >> 2;
                const int hw = hv & 7, hh = (hv >> 3) & 7, hd = hv >> 6;
                const int d = td0 - 1 + hd, h = th0 - 1 + hh, w = tw0 - 1 + hw;
                const bool ok = hv < HV && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                const long long v = (((long long)n * p.D + (ok ? d : 0)) * p.H + (ok ? h : 0)) * p.W + (ok ? w : 0);
                const u32x4_t val = *(const u32x4_t*)(p.x + v * p.ldx + ks * 32 + hq * 8);
                areg[i] = ok ? val : u32x4_t{0, 0, 0, 0};
            }
            if (si) __syncthreads();                      // the previous stage's fragment reads are done
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                const int idx = tid + i * 256;
                if (idx < W_BYTES / 16) *(u32x4_t*)(ldsW + idx * 16) = wreg[i];
            }
#pragma unroll
            for (int i = 0; i < A_ITERS; ++i) {
                const int idx = tid + i * 256;
                if ((idx >> 2) < HV) *(u32x4_t*)(ldsA + (idx & 3) * A_PLANE + (idx >> 2) * 16) = areg[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const int toff = ((kd * HS + kh) * HS + kw) * 16;
            const u32x4_t w0 = *(const u32x4_t*)(ldsW + tap * (4 * 32 * 16) + bbase);
            const u32x4_t w1 = *(const u32x4_t*)(ldsW + tap * (4 * 32 * 16) + bbase + 256);
            const u32x4_t a0 = *(const u32x4_t*)(ldsA + abase[0] + toff);
            mma_chunk<bf16_t>(acc[0][0], w0, a0);
            mma_chunk<bf16_t>(acc[0][1], w1, a0);
            if (nf > 1) {
                const u32x4_t a1 = *(const u32x4_t*)(ldsA + abase[1] + toff);
                mma_chunk<bf16_t>(acc[1][0], w0, a1);
                mma_chunk<bf16_t>(acc[1][1], w1, a1);
            }
        }
    }
    // ---- partial block out, channel-group major: part[kg][channel group of 4][voxel][4]; a lane holds one group
    // (cb * 8 + j * 4 + q) of its voxel, so the finish kernels read whole voxel runs of one group as contiguous 16-byte rows
    const long long NV = (long long)p.N * p.D * p.H * p.W;
    float* po = p.part + ((long long)kg * (p.M >> 2) + cb * 8 + q) * NV * 4;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        if (f < nf) {
            const int v = (f0 + f) * 16 + r;
            if (v < TV) {
                const int vw = v % TS, vh = (v / TS) % TS, vd = v / (TS * TS);
                if (td0 + vd < p.D && th0 + vh < p.H && tw0 + vw < p.W) {      // edge tiles of grids that are not tile multiples
                    const long long gv = (((long long)n * p.D + td0 + vd) * p.H + th0 + vh) * p.W + tw0 + vw;
                    *(f32x4_t*)(po + gv * 4) = acc[f][0];
                    *(f32x4_t*)(po + (4 * NV + gv) * 4) = acc[f][1];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
struct K3sFinParams {
    const float* part; int nks; long long NV;   // NV = N * S
    int N, D, H, W, M;
    const float* bias;
    // forward
    const float* gamma; const float* beta; float eps, slope;
    bf16_t* yraw; long long ldy;
    bf16_t* act; long long lda;
    const bf16_t* res; long long ldr;            // optional residual added before the LeakyReLU (UnetResBlock's second conv)
    bf16_t* pooled; long long ldp;
    float* stats;                                // [N][M][2] (sum, sum of squares) of the stored raw output
    // backward
    bf16_t* dx; long long lddx;                  // plain: the summed gradient; unit mode: dy of the receiving unit
    const bf16_t* uy; long long lduy;            // receiving unit: raw conv output, forward statistics, affine
    const float* ustats; const float* ugamma; const float* ubeta;
    float* dgamma; float* dbeta; int acc;
    int dbg;                                     // timing experiments only (MSSEG_K3S_DBG): 1 = no output stores, 2 = no partial loads
};

constexpr int FCH = 4;        // channels of a finish workgroup (one float4 per stage group and voxel)
constexpr int NKG_MAX = 8;    // stage groups

MSSEG_DEVFN void store4_bf16(bf16_t* p, const float* v) {
    bf16x4_t o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *(bf16x4_t*)p = o;
}

// fixed-order sum of NV per-thread values over the BS threads of the block: lanes by xor-shuffles, then the per-wave rows
// are added in wave order by the first NV threads and broadcast through LDS.  lds: >= (BS / 64 + 1) * NV floats.
template <int BS, int NV>
MSSEG_DEVFN void block_sum(float* v, float* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] = wave_sum(v[e]);
    __syncthreads();                                  // lds may still be read from a previous call
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < NV; ++e) lds[wave * NV + e] = v[e];
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float t = lds[threadIdx.x];
#pragma unroll
        for (int w = 1; w < BS / 64; ++w) t += lds[w * NV + threadIdx.x];
        lds[(BS / 64) * NV + threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] = lds[(BS / 64) * NV + e];
}

// sum over the stage groups of the partial rows of one (channel group, item): every load of the thread's NI items is issued
// before the first use (NKG = groups the kernel is built for; absent groups add exact zeros, the order is fixed)
// NIC: items whose loads are in flight together (register-tight kernels take their items in chunks)
template <int NI, int NKG, int NIC = NI>
MSSEG_DEVFN void sum_partials(const K3sFinParams& p, int cg, const long long (&row)[NI], const bool (&ok)[NI], float (&a)[NI][FCH]) {
    static_assert(NI % NIC == 0, "chunks must divide the items");
#pragma unroll
    for (int c = 0; c < NI; c += NIC) {
        f32x4_t t[NIC][NKG];
#pragma unroll
        for (int i = 0; i < NIC; ++i) {
            const float* src = p.part + ((long long)cg * p.NV + (ok[c + i] ? row[c + i] : 0)) * 4;
#pragma unroll
            for (int k = 0; k < NKG; ++k)
                t[i][k] = (k < p.nks && ok[c + i] && !(p.dbg & 2)) ? *(const f32x4_t*)(src + (long long)k * p.NV * p.M) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NIC; ++i) {
            f32x4_t acc = t[i][0];
#pragma unroll
            for (int k = 1; k < NKG; ++k) acc += t[i][k];
#pragma unroll
            for (int e = 0; e < FCH; ++e) a[c + i][e] = acc[e];
        }
        if (c + NIC < NI) asm volatile("" ::: "memory");   // keep the chunks' loads from being hoisted together
    }
}

// BS threads, VPT voxels per thread (S <= BS * VPT): 256 x 1 for 6^3, 1024 x 2 for 12^3
template <int BS, int VPT, int NKG>
__global__ __launch_bounds__(BS) void k3s_fwd_finish_kernel(const K3sFinParams p) {
    __shared__ float red[(BS / 64 + 1) * 8];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];   // pooled only: the slab's activation [S][4] bf16
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * FCH, n = blockIdx.y;
    const int S = p.D * p.H * p.W;
    long long row[VPT];
    bool ok[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) { ok[i] = tid + i * BS < S; row[i] = (long long)n * S + tid + i * BS; }
    float y[VPT][FCH];
    sum_partials<VPT, NKG>(p, blockIdx.x, row, ok, y);
    float bias[FCH];
#pragma unroll
    for (int e = 0; e < FCH; ++e) bias[e] = p.bias ? p.bias[c0 + e] : 0.f;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        if (ok[i]) {
            bf16x4_t o;
#pragma unroll
            for (int e = 0; e < FCH; ++e) o[e] = (bf16_t)(y[i][e] + bias[e]);
            if (!(p.dbg & 1)) *(bf16x4_t*)(p.yraw + row[i] * p.ldy + c0) = o;
#pragma unroll
            for (int e = 0; e < FCH; ++e) {
                y[i][e] = (float)o[e];              // statistics of the STORED values, as every conv epilogue takes them
                s[e] += y[i][e];
                s[4 + e] += y[i][e] * y[i][e];
            }
        }
    }
    block_sum<BS, 8>(s, red);
    if (tid < FCH) {
        p.stats[((long long)n * p.M + c0 + tid) * 2 + 0] = s[tid];
        p.stats[((long long)n * p.M + c0 + tid) * 2 + 1] = s[4 + tid];
    }
    float sc[FCH], sh[FCH];
    const float inv = 1.0f / (float)S;
#pragma unroll
    for (int e = 0; e < FCH; ++e) {
        const float mean = s[e] * inv;
        float var = s[4 + e] * inv - mean * mean;
        var = var > 0.f ? var : 0.f;
        const float rstd = rsqrtf(var + p.eps);
        sc[e] = rstd * (p.gamma ? p.gamma[c0 + e] : 1.f);
        sh[e] = (p.beta ? p.beta[c0 + e] : 0.f) - mean * sc[e];
    }
    bf16x4_t* slab = (bf16x4_t*)dyn;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        if (ok[i]) {
            bf16x4_t o;
            bf16x4_t rv = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            if (p.res) rv = *(const bf16x4_t*)(p.res + row[i] * p.ldr + c0);
#pragma unroll
            for (int e = 0; e < FCH; ++e) {
                const float z = y[i][e] * sc[e] + sh[e] + (float)rv[e];      // as msseg_instnorm_act_fwd: one rounding at the end
                o[e] = (bf16_t)(z > 0.f ? z : z * p.slope);
            }
            if (!(p.dbg & 1)) *(bf16x4_t*)(p.act + row[i] * p.lda + c0) = o;
            if (p.pooled) slab[tid + i * BS] = o;
        }
    }
    if (p.pooled) {
        __syncthreads();
        const int PD = p.D >> 1, PH = p.H >> 1, PW = p.W >> 1;
        for (int pv = tid; pv < PD * PH * PW; pv += BS) {
            const int pw = pv % PW, ph = (pv / PW) % PH, pd = pv / (PW * PH);
            float m[FCH];
#pragma unroll
            for (int e = 0; e < FCH; ++e) m[e] = -3.0e38f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int v = ((2 * pd + (j >> 2)) * p.H + 2 * ph + ((j >> 1) & 1)) * p.W + 2 * pw + (j & 1);
                const bf16x4_t a = slab[v];
#pragma unroll
                for (int e = 0; e < FCH; ++e) m[e] = fmaxf(m[e], (float)a[e]);
            }
            store4_bf16(p.pooled + ((long long)n * PD * PH * PW + pv) * p.ldp + c0, m);
        }
    }
}

// input-gradient finish, plain: dx = sum of the stage groups (one workgroup per (sample, 4 channels))
template <int BS, int VPT, int NKG>
__global__ __launch_bounds__(BS) void k3s_bwd_plain_kernel(const K3sFinParams p) {
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * FCH, n = blockIdx.y;
    const int S = p.D * p.H * p.W;
    long long row[VPT];
    bool ok[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) { ok[i] = tid + i * BS < S; row[i] = (long long)n * S + tid + i * BS; }
    float a[VPT][FCH];
    sum_partials<VPT, NKG>(p, blockIdx.x, row, ok, a);
#pragma unroll
    for (int i = 0; i < VPT; ++i)
        if (ok[i]) store4_bf16(p.dx + row[i] * p.lddx + c0, a[i]);
}

// input-gradient finish, unit mode: the conv's input was the activation of a conv + InstanceNorm + LeakyReLU unit (raw
// output uy, statistics ustats): dx = that unit's dy.  One workgroup per 4 channels; it takes the samples TWO at a time
// (items = (sample, voxel) pairs over both, 2 * VPT per thread), so that a batch of 2 is one round of loads, and keeps
// the per-channel sums of all samples: dgamma / dbeta are complete inside it.
template <int BS, int VPT, int NKG>
__global__ __launch_bounds__(BS) void k3s_bwd_unit_kernel(const K3sFinParams p) {
    constexpr int NI = 2 * VPT;
    __shared__ float red[(BS / 64 + 1) * 16];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * FCH;
    const int S = p.D * p.H * p.W;
    const float inv = 1.0f / (float)S;
    float g0[FCH], g1[FCH];
#pragma unroll
    for (int e = 0; e < FCH; ++e) g0[e] = g1[e] = 0.f;
    for (int nb = 0; nb < p.N; nb += 2) {
        const bool two = nb + 1 < p.N;
        float mean[2][FCH], rstd[2][FCH], ga[FCH], be[FCH];
#pragma unroll
        for (int e = 0; e < FCH; ++e) {
            ga[e] = p.ugamma ? p.ugamma[c0 + e] : 1.f;
            be[e] = p.ubeta ? p.ubeta[c0 + e] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = two || j == 0 ? nb + j : nb;
#pragma unroll
            for (int e = 0; e < FCH; ++e) {
                const float s0 = p.ustats[((long long)n * p.M + c0 + e) * 2 + 0], s1 = p.ustats[((long long)n * p.M + c0 + e) * 2 + 1];
                mean[j][e] = s0 * inv;
                float var = s1 * inv - mean[j][e] * mean[j][e];
                var = var > 0.f ? var : 0.f;
                rstd[j][e] = rsqrtf(var + p.eps);
            }
        }
        // item i of the thread: sample nb + (i >= VPT), voxel tid + (i % VPT) * BS
        long long row[NI];
        bool ok[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = i / VPT, v = tid + (i % VPT) * BS;
            ok[i] = v < S && (j == 0 || two);
            row[i] = (long long)(nb + j) * S + v;
        }
        float a[NI][FCH];
        bf16x4_t yb[NI];
        sum_partials<NI, NKG, (BS >= 1024 && NKG >= 8) ? NI / 4 : ((BS >= 1024 && NKG >= 4) ? NI / 2 : NI)>(p, blockIdx.x, row, ok, a);
        float s[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = i / VPT;
            if (ok[i]) {
                yb[i] = *(const bf16x4_t*)(p.uy + row[i] * p.lduy + c0);
#pragma unroll
                for (int e = 0; e < FCH; ++e) {
                    // the gradient of the activation as the unfused path stores it (bf16), then the unit's backward with the
                    // scale / shift formed exactly as the forward formed them (the sign of the pre-activation must agree)
                    const float da = (float)(bf16_t)a[i][e];
                    const float yv = (float)yb[i][e];
                    const float sc = rstd[j][e] * ga[e];
                    const float z = yv * sc + (be[e] - mean[j][e] * sc);
                    a[i][e] = z > 0.f ? da : da * p.slope;              // dz
                    const float xh = (yv - mean[j][e]) * rstd[j][e];
                    s[j * 8 + e] += a[i][e];
                    s[j * 8 + 4 + e] += a[i][e] * xh;
                }
            }
        }
        block_sum<BS, 16>(s, red);
#pragma unroll
        for (int e = 0; e < FCH; ++e) {        // samples in order: the same sums whatever the pairing
            g0[e] += s[e]; g1[e] += s[4 + e];
            if (two) { g0[e] += s[8 + e]; g1[e] += s[12 + e]; }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = i / VPT;
            if (ok[i]) {
                float o[FCH];
#pragma unroll
                for (int e = 0; e < FCH; ++e) {
                    const float xh = ((float)yb[i][e] - mean[j][e]) * rstd[j][e];
                    o[e] = rstd[j][e] * ga[e] * (a[i][e] - s[j * 8 + e] * inv - xh * (s[j * 8 + 4 + e] * inv));
                }
                store4_bf16(p.dx + row[i] * p.lddx + c0, o);
            }
        }
    }
    if (p.dgamma != nullptr && tid < FCH) {
        p.dbeta[c0 + tid] = p.acc ? p.dbeta[c0 + tid] + g0[tid] : g0[tid];
        p.dgamma[c0 + tid] = p.acc ? p.dgamma[c0 + tid] + g1[tid] : g1[tid];
    }
}

// kernel instantiations by (block, voxels per thread) x stage groups (2, 4, 8)
#define K3S_LAUNCH(KERN, grid, lds, s, p, S, nks)                                                         \
    do {                                                                                                  \
        if ((S) <= 256) {                                                                                 \
            if ((nks) <= 2) hipLaunchKernelGGL((KERN<256, 1, 2>), grid, dim3(256), lds, s, p);            \
            else if ((nks) <= 4) hipLaunchKernelGGL((KERN<256, 1, 4>), grid, dim3(256), lds, s, p);       \
            else hipLaunchKernelGGL((KERN<256, 1, 8>), grid, dim3(256), lds, s, p);                       \
        } else {                                                                                          \
            if ((nks) <= 2) hipLaunchKernelGGL((KERN<1024, 2, 2>), grid, dim3(1024), lds, s, p);          \
            else if ((nks) <= 4) hipLaunchKernelGGL((KERN<1024, 2, 4>), grid, dim3(1024), lds, s, p);     \
            else hipLaunchKernelGGL((KERN<1024, 2, 8>), grid, dim3(1024), lds, s, p);                     \
        }                                                                                                 \
    } while (0)

int k3s_check(const void* x, long long ldx, const void* wp, const void* part, int N, int D, int H, int W, int K, int M,
              const char* who) {
    if (!x || !wp || !part) MSSEG_FAIL(MSSEG_EINVAL, "%s: null pointer", who);
    if (N < 1 || D < 1 || H < 1 || W < 1) MSSEG_FAIL(MSSEG_EINVAL, "%s: bad spatial dims %dx%dx%d", who, D, H, W);
    if (K < 32 || M < 32 || K % 32 || M % 32) MSSEG_FAIL(MSSEG_EINVAL, "%s: channels %d -> %d must be multiples of 32", who, K, M);
    if (ldx < K || ldx % 8 || ((uintptr_t)x & 15) || ((uintptr_t)wp & 15) || ((uintptr_t)part & 15))
        MSSEG_FAIL(MSSEG_EINVAL, "%s: 16-byte aligned tensors with a voxel stride that is a multiple of 8", who);
    return MSSEG_OK;
}

}  // namespace

extern "C" {

int msseg_conv3d_k3_small_ok(int N, int D, int H, int W, int Cin, int Cout, int dtype) {
    if (dtype != MSSEG_BF16 || N < 1 || N > MSSEG_STATS_NMAX) return 0;
    // whole tiles (the 12^3 / 6^3 levels), or the 3^3 grid of Swin-UNETR's bottleneck (one quarter-filled tile per sample:
    // the layer is 32 MB of weights for 54 voxels, what counts is that every workgroup streams a different slice of them)
    const bool whole = D >= TSD && H >= TS && W >= TS && D % TSD == 0 && H % TS == 0 && W % TS == 0;
    if (!whole && !(D == 3 && H == 3 && W == 3)) return 0;
    if ((long long)D * H * W > 1024 * 2) return 0;               // the finish kernels hold a (sample, 4 channel) slab per workgroup
    return (Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0 && Cin / 32 <= 64) ? 1 : 0;
}

/* stages per workgroup: the fewest that keep the grid within one round of two workgroups per CU and the stage groups
 * within what a finish thread sums at once */
static int k3s_kpw(int N, int D, int H, int W, int Cin, int Cout) {
    const long long units = (long long)N * ceil_div(D, TSD) * ceil_div(H, TS) * ceil_div(W, TS) * (Cout / 32);
    const int nks = Cin / 32;
    int kpw = 1;
    while (kpw < nks && (units * ceil_div(nks, kpw) > 2LL * msseg_num_cus() || ceil_div(nks, kpw) > NKG_MAX)) kpw *= 2;
    return kpw;
}

int msseg_conv3d_k3_small_stage_groups(int N, int D, int H, int W, int Cin, int Cout) {
    return ceil_div(Cin / 32, k3s_kpw(N, D, H, W, Cin, Cout));
}

size_t msseg_conv3d_k3_small_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout) {
    return (size_t)msseg_conv3d_k3_small_stage_groups(N, D, H, W, Cin, Cout) * (size_t)N * D * H * W * Cout * sizeof(float);
}

/* partial sums of conv(x, w): part[stage][voxel][Cout]; wp = msseg_pack_weights image with cout block 32 (forward image,
 * or the flipped / transposed one for the input gradient) */
int msseg_conv3d_k3_small_partials(const void* x, long long ldx, const void* wp, float* part, size_t part_bytes, int N,
                                   int D, int H, int W, int Cin, int Cout, msseg_stream_t stream) {
    int rc = k3s_check(x, ldx, wp, part, N, D, H, W, Cin, Cout, "conv3d_k3_small_partials");
    if (rc) return rc;
    if (part_bytes < msseg_conv3d_k3_small_workspace_bytes(N, D, H, W, Cin, Cout))
        MSSEG_FAIL(MSSEG_EWORKSPACE, "conv3d_k3_small_partials: workspace %zu B < %zu B", part_bytes,
                   msseg_conv3d_k3_small_workspace_bytes(N, D, H, W, Cin, Cout));
    const int kpw = k3s_kpw(N, D, H, W, Cin, Cout);
    K3sParams p{(const bf16_t*)x, ldx, (const bf16_t*)wp, part, N, D, H, W, Cin, Cout, ceil_div(D, TSD), ceil_div(H, TS), ceil_div(W, TS), kpw};
    static msseg_lds_attr_once attr;
    if (!attr.ensure((const void*)k3s_kernel, K3S_LDS)) MSSEG_FAIL(MSSEG_ELAUNCH, "conv3d_k3_small: cannot set dynamic LDS size %d", K3S_LDS);
    const long long tiles = (long long)N * p.td * p.th * p.tw;
    if (tiles > 0x7fffffffLL || Cout / 32 > 65535) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_small: grid too large");
    MSSEG_KTIMED("k3s_kernel", (hipStream_t)stream,
                 hipLaunchKernelGGL(k3s_kernel, dim3((unsigned)tiles, Cout / 32, ceil_div(Cin / 32, kpw)), dim3(256), K3S_LDS,
                                    (hipStream_t)stream, p));
    MSSEG_CHECK_LAUNCH("conv3d_k3_small_partials");
    return MSSEG_OK;
}

/* y = sum of the stages + bias (bf16, voxel stride ldy); stats[N][Cout][2] of y; act = lrelu(instance_norm(y) * gamma + beta)
 * (voxel stride lda: may be a channel slice of a concat buffer); pooled (optional) = max_pool3d(act, 2) */
int msseg_conv3d_k3_small_fwd_finish(const float* part, int nstages, const float* bias, const float* gamma,
                                     const float* beta, float eps, float slope, void* yraw, long long ldy, void* act,
                                     long long lda, void* pooled, long long ldp, float* stats, int N, int D, int H, int W,
                                     int Cout, msseg_stream_t stream) {
    return msseg_conv3d_k3_small_fwd_finish_res(part, nstages, bias, gamma, beta, eps, slope, yraw, ldy, act, lda, nullptr, 0,
                                                pooled, ldp, stats, N, D, H, W, Cout, stream);
}

int msseg_conv3d_k3_small_fwd_finish_res(const float* part, int nstages, const float* bias, const float* gamma,
                                         const float* beta, float eps, float slope, void* yraw, long long ldy, void* act,
                                         long long lda, const void* residual, long long ldr, void* pooled, long long ldp,
                                         float* stats, int N, int D, int H, int W, int Cout, msseg_stream_t stream) {
    if (residual && (ldr % 4 || ((uintptr_t)residual & 7)))
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_small_fwd_finish: the residual must be 8-byte aligned with a stride that is a multiple of 4");
    if (!part || !yraw || !act || !stats || nstages < 1 || nstages > NKG_MAX)
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_small_fwd_finish: bad args (1 ... %d stage groups)", NKG_MAX);
    const int S = D * H * W;
    if (S > 1024 * 2 || Cout % 4 || ldy % 4 || lda % 4 || (pooled && (ldp % 4 || D % 2 || H % 2 || W % 2)))
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_small_fwd_finish: unsupported shape %dx%dx%d x %d", D, H, W, Cout);
    if (((uintptr_t)yraw & 7) || ((uintptr_t)act & 7) || ((uintptr_t)pooled & 7))
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_small_fwd_finish: tensors must be 8-byte aligned");
    K3sFinParams p{};
    p.part = part; p.nks = nstages; p.NV = (long long)N * S; p.N = N; p.D = D; p.H = H; p.W = W; p.M = Cout; p.bias = bias;
    p.gamma = gamma; p.beta = beta; p.eps = eps; p.slope = slope;
    p.yraw = (bf16_t*)yraw; p.ldy = ldy; p.act = (bf16_t*)act; p.lda = lda; p.pooled = (bf16_t*)pooled; p.ldp = ldp; p.stats = stats;
    p.res = (const bf16_t*)residual; p.ldr = ldr;
    static const int dbg = getenv("MSSEG_K3S_DBG") ? atoi(getenv("MSSEG_K3S_DBG")) : 0;
    p.dbg = dbg;
    const int lds = pooled ? S * 8 : 0;
    dim3 grid(Cout / FCH, N);
    K3S_LAUNCH(k3s_fwd_finish_kernel, grid, lds, (hipStream_t)stream, p, S, nstages);
    MSSEG_CHECK_LAUNCH("conv3d_k3_small_fwd_finish");
    return MSSEG_OK;
}

/* input-gradient finish.  unit_yraw == NULL: dx = sum of the stages.  Otherwise dx = dy of the conv + InstanceNorm +
 * LeakyReLU unit whose activation was the conv's input (raw output unit_yraw, forward statistics unit_stats, affine
 * unit_gamma / unit_beta), and dgamma / dbeta (optional) receive that unit's affine gradients. */
int msseg_conv3d_k3_small_bwd_finish(const float* part, int nstages, void* dx, long long lddx, const void* unit_yraw,
                                     long long lduy, const float* unit_stats, const float* unit_gamma,
                                     const float* unit_beta, float eps, float slope, float* dgamma, float* dbeta,
                                     int accumulate, int N, int D, int H, int W, int Cin, msseg_stream_t stream) {
    if (!part || !dx || nstages < 1 || nstages > NKG_MAX)
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_small_bwd_finish: bad args (1 ... %d stage groups)", NKG_MAX);
    const int S = D * H * W;
    if (S > 1024 * 2 || Cin % 4 || lddx % 4 || ((uintptr_t)dx & 7) || (unit_yraw && (lduy % 4 || ((uintptr_t)unit_yraw & 7) || !unit_stats)))
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_small_bwd_finish: unsupported shape / alignment");
    if ((dgamma == nullptr) != (dbeta == nullptr)) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_small_bwd_finish: dgamma and dbeta go together");
    K3sFinParams p{};
    p.part = part; p.nks = nstages; p.NV = (long long)N * S; p.N = N; p.D = D; p.H = H; p.W = W; p.M = Cin;
    p.eps = eps; p.slope = slope; p.dx = (bf16_t*)dx; p.lddx = lddx;
    p.uy = (const bf16_t*)unit_yraw; p.lduy = lduy; p.ustats = unit_stats; p.ugamma = unit_gamma; p.ubeta = unit_beta;
    p.dgamma = dgamma; p.dbeta = dbeta; p.acc = accumulate;
    hipStream_t s = (hipStream_t)stream;
    if (unit_yraw) {
        dim3 grid(Cin / FCH, 1);
        K3S_LAUNCH(k3s_bwd_unit_kernel, grid, 0, s, p, S, nstages);
    } else {
        dim3 grid(Cin / FCH, N);
        K3S_LAUNCH(k3s_bwd_plain_kernel, grid, 0, s, p, S, nstages);
    }
    MSSEG_CHECK_LAUNCH("conv3d_k3_small_bwd_finish");
    return MSSEG_OK;
}

}  // extern "C"
