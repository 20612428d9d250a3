// Weight gradient of conv3d 3x3x3 (stride 1, pad 1), channels-last bf16 -- "ping-pong" kernel for gfx950.
//
//   dW[co][tap][ci] = sum_v dy[v][co] * x[v + tap][ci]
//
// (torch.nn.Conv3d weight gradient of MONAI's Convolution blocks: BasicUNet's TwoConv -- BASELINE.json configs 1-3, SURVEY.md row A15;
// MONAI is not vendored by the reference -- and the UnetResBlock convs built at /root/reference/models/segmentors/swin_unetr.py:73-128.)
// Same execution scheme as conv3d_k3_pp.hip: one persistent workgroup of 8 waves per CU in two groups of 4 that
// alternate roles per phase -- one group runs MFMAs on the tile staged in its LDS buffers, the other issues the
// LDS-DMA loads (global_load_lds_dwordx4) of its next tile.  One workgroup grid column per (32-cout, 32-cin) block pair.
//
// The contraction index is the voxel, HBM holds channels-last rows, so both MFMA operands are read with the
// transposing LDS read ds_read_b64_tr_b16 from row-major [voxel][64 B] images.  An LDS-DMA image is lane-linear
// (no row padding possible) and one transposing read touches rows r..r+3 and r+8..r+11, i.e. two rows on every bank:
// the two 32-byte halves of a row are therefore swapped on rows with bit 3 of the row index set -- applied on the
// SOURCE address of the DMA and undone in the (precomputed, per-lane) read offsets, which makes the reads conflict-free
// for any tap shift.  The x halo keeps a pitch of 24 rows per 18-voxel line so that every k-step starts on a
// multiple of 16 rows and the swizzle does not depend on the k-step.
//
// Tile = 4 x 4 x 16 voxels (8 k-steps of 32 voxels); wave w of a group owns taps 7w .. 7w+6 (7/7/7/6) for the whole
// 32 x 32 block: 28 MFMAs per k-step, dy fragments shared by the wave's taps.  Partial sums stay in registers across
// all tiles of the workgroup; at the end the two groups are added through LDS and ONE fp32 slab per workgroup is
// written for the deterministic slab reduction in igemm_wgrad.hip.
#include "k3pp.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int TD = 4, TH = 4, TW = 16;
constexpr int PD = TD + 2, PH = TH + 2, PW = TW + 2;
constexpr int PWP = PW;                   // LDS row pitch of one halo line
constexpr int HV = PD * PH * PWP;         // 864 LDS rows of the x halo image
constexpr int TV = TD * TH * TW;          // 256 tile voxels
constexpr int Q_BYTES = HV * 64;
constexpr int P_BYTES = TV * 64;
constexpr int GRP_BYTES = Q_BYTES + P_BYTES;
constexpr int NIQ = (HV + 15) / 16;       // DMA wave-instructions per image (16 rows of 64 B each)
constexpr int NIP = TV / 16;
constexpr int NIQ_W = (NIQ + 3) / 4, NIP_W = NIP / 4;   // per wave of a group
constexpr int NKS = TV / 16;                 // k-steps: one 16-voxel tile row each
constexpr int TAPW = 7;
constexpr int NTHREADS = 512;
constexpr int SLAB_FLOATS = 27 * 32 * 32;

__device__ u32x4_t g_wg_zero_chunk;
__device__ unsigned long long g_k3wg_cycles[8];

MSSEG_DEVFN void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// the pointer stays an LDS-address-space pointer up to the builtin so that constant offsets fold into the
// instruction's 16-bit offset field (one address VGPR per lane-dependent base instead of one per read)
MSSEG_DEVFN bf16x4_t lds_tr(lds_u8* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)p);
}

MSSEG_DEVFN u32x4_t tr_frag(lds_u8* r0, lds_u8* r1) {
    const bf16x4_t lo = lds_tr(r0), hi = lds_tr(r1);
    const bf16x8_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(u32x4_t, f);
}

struct TileCo { int n, d0, h0, w0; };

// LDS halo row of the first voxel of tile row ks (td = ks / TH, th = ks % TH)
constexpr int ks_row(int ks) { return ((ks / TH) * PH + ks % TH) * PWP; }

template <int TIMING>
__global__ __launch_bounds__(NTHREADS, 1) void k3wg_pp_kernel(const K3WgParams p) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    lds_u8* smem3 = (lds_u8*)smem;
    unsigned char* ldsQ = smem + grp * GRP_BYTES;
    unsigned char* ldsP = ldsQ + Q_BYTES;
    const int mblk = blockIdx.y / p.kblks, kblk = blockIdx.y % p.kblks;
    const unsigned char* pg = (const unsigned char*)p.pten + mblk * 64;   // dy, this cout block
    const unsigned char* qg = (const unsigned char*)p.qten + kblk * 64;   // x, this cin block

    // ---- tile schedule (XCD-contiguous, as in conv3d_k3_pp.hip)
    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int ntiles = p.N * tiles_d * tiles_h * tiles_w;
    int t_first, t_step, t_end;
    if ((gridDim.x & 7) == 0) {
        const int chunk = (ntiles + 7) >> 3, xcd = blockIdx.x & 7;
        t_first = xcd * chunk + (blockIdx.x >> 3);
        t_step = gridDim.x >> 3;
        t_end = min(ntiles, (xcd + 1) * chunk);
    } else {
        t_first = blockIdx.x; t_step = gridDim.x; t_end = ntiles;
    }
    const int n_my = t_first < t_end ? (t_end - t_first + t_step - 1) / t_step : 0;
    auto tile_of = [&](int k) {
        int t = t_first + k * t_step;
        TileCo tc;
        tc.w0 = (t % tiles_w) * TW; t /= tiles_w;
        tc.h0 = (t % tiles_h) * TH; t /= tiles_h;
        tc.d0 = (t % tiles_d) * TD; t /= tiles_d;
        tc.n = t;
        return tc;
    };

    // ---- memory role: per-lane byte offsets of the DMA sources, relative to the tile's halo / tile origin.
    // Lane l of wave-instruction `it` fills row it*16 + (l >> 2), physical 16-byte slot l & 3, with the logical chunk
    // (l & 3) ^ 2*bit3(row); bit 3 of the row is bit 5 of the lane for every `it`.
    const int lrow = lane >> 2;
    const int lchunk = lane & 3;
    constexpr unsigned SKIP = 0xffffffffu;   // q_off of an unused LDS row
    unsigned q_off[NIQ_W], p_off[NIP_W];
#pragma unroll
    for (int j = 0; j < NIQ_W; ++j) {
        const int hv = (wq + 4 * j) * 16 + lrow;
        const int hd = hv / (PH * PWP), rem = hv - hd * (PH * PWP), hh = rem / PWP, hw = rem - hh * PWP;
        q_off[j] = (hv < HV && hw < PW) ? (unsigned)((((long long)hd * p.H + hh) * p.W + hw) * p.ldq * 2 + lchunk * 16) : SKIP;
    }
#pragma unroll
    for (int j = 0; j < NIP_W; ++j) {
        const int tv = (wq + 4 * j) * 16 + lrow;
        const int td = tv / (TH * TW), th = (tv / TW) % TH, tw = tv % TW;
        p_off[j] = (unsigned)((((long long)td * p.H + th) * p.W + tw) * p.ldp * 2 + lchunk * 16);
    }
    // channel chunks beyond the tensor's channel count (last block of a count that is not a multiple of 32) read zeros
    const bool p_cok = mblk * 32 + lchunk * 8 < p.M, q_cok = kblk * 32 + lchunk * 8 < p.K;
    const bool chunks_full = (mblk + 1) * 32 <= p.M && (kblk + 1) * 32 <= p.K;
    auto load_tile = [&](const TileCo& tc) {
        const int dB = tc.d0 - 1, hB = tc.h0 - 1, wB = tc.w0 - 1;
        const long long qvox = (((long long)tc.n * p.D + dB) * p.H + hB) * p.W + wB;
        const long long pvox = (((long long)tc.n * p.D + tc.d0) * p.H + tc.h0) * p.W + tc.w0;
        const unsigned char* qb = qg + qvox * p.ldq * 2;
        const unsigned char* pb = pg + pvox * p.ldp * 2;
        const bool interior = chunks_full && dB >= 0 && dB + PD <= p.D && hB >= 0 && hB + PH <= p.H && wB >= 0 && wB + PW <= p.W;
        if (interior) {
#pragma unroll
            for (int j = 0; j < NIQ_W; ++j) {
                if (q_off[j] != SKIP) glds16(qb + q_off[j], ldsQ + (wq + 4 * j) * 1024);
            }
#pragma unroll
            for (int j = 0; j < NIP_W; ++j) glds16(pb + p_off[j], ldsP + (wq + 4 * j) * 1024);
        } else {
            const unsigned char* zsrc = (const unsigned char*)&g_wg_zero_chunk;
#pragma unroll
            for (int j = 0; j < NIQ_W; ++j) {
                const int it = wq + 4 * j;
                const int hv = it * 16 + lrow;
                const int hd = hv / (PH * PWP), rem = hv - hd * (PH * PWP), hh = rem / PWP, hw = rem - hh * PWP;
                const bool inb = q_cok && (unsigned)(dB + hd) < (unsigned)p.D && (unsigned)(hB + hh) < (unsigned)p.H &&
                                 (unsigned)(wB + hw) < (unsigned)p.W;
                const unsigned char* src = inb ? qb + q_off[j] : zsrc;
                if (q_off[j] != SKIP) glds16(src, ldsQ + it * 1024);
            }
#pragma unroll
            for (int j = 0; j < NIP_W; ++j) {
                const int tv = (wq + 4 * j) * 16 + lrow;
                const int td = tv / (TH * TW), th = (tv / TW) % TH, tw = tv % TW;
                const bool inb = p_cok && tc.d0 + td < p.D && tc.h0 + th < p.H && tc.w0 + tw < p.W;
                const unsigned char* src = inb ? pb + p_off[j] : zsrc;
                glds16(src, ldsP + (wq + 4 * j) * 1024);
            }
        }
    };

    // ---- MFMA role (v_mfma_f32_32x32x16_bf16: one instruction = one tap x 16 voxels x the whole 32 x 32 block).
    // Operand layout: lane l holds 8 consecutive voxels (k = 8*(l>>5) ..) of channel l & 31.  With the transposing
    // read, 16-lane group G = l >> 4 supplies rows (voxels) 8*(G>>1) + 4*i + qr of channel tile G & 1 at channels
    // 4*pc..; the 32 lanes of one half-wave then read 4 consecutive full 64-byte rows: conflict-free without swizzle.
    // A 16-cycle-issue budget per 32-cycle MFMA leaves room for the fragment reads (the 16x16x32 form, which holds the
    // vector issue port for 8 of its 16 cycles, ran this loop at 22 cycles per MFMA).
    const int G = lane >> 4, qr = (lane >> 2) & 3, pc = lane & 3;
    const unsigned qbase = grp * GRP_BYTES, pbase = qbase + Q_BYTES;
    unsigned p_rd[2], q_rd[TAPW][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lrow = 8 * (G >> 1) + 4 * i + qr;
        p_rd[i] = pbase + lrow * 64 + ((G & 1) << 5) + pc * 8;
#pragma unroll
        for (int tt = 0; tt < TAPW; ++tt) {
            int tap = wq * TAPW + tt;
            if (tap > 26) tap = 26;             // wave 3 owns 6 taps: its 7th slot repeats tap 26 and is discarded
            const int row = ((tap / 9) * PH + ((tap / 3) % 3)) * PWP + (tap % 3) + lrow;
            q_rd[tt][i] = qbase + row * 64 + ((G & 1) << 5) + pc * 8;
        }
    }

    f32x16_t acc[TAPW];
#pragma unroll
    for (int t = 0; t < TAPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    auto compute = [&]() {
        constexpr int NSTEP = NKS * TAPW;
        constexpr int QA = 2;                 // x fragments in flight ahead of their MFMA
        u32x4_t pf[2], qf[QA + 1];
        auto ldp = [&](int ks) { pf[ks & 1] = tr_frag(smem3 + p_rd[0] + ks * 1024, smem3 + p_rd[1] + ks * 1024); };
        auto ldq = [&](int s) {     // s = ks * TAPW + tt
            const int ks = s / TAPW, tt = s % TAPW;
            qf[s % (QA + 1)] = tr_frag(smem3 + q_rd[tt][0] + ks_row(ks) * 64, smem3 + q_rd[tt][1] + ks_row(ks) * 64);
        };
        ldp(0);
#pragma unroll
        for (int s = 0; s < QA; ++s) ldq(s);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int tt = 0; tt < TAPW; ++tt) {
                const int s = ks * TAPW + tt;
                if (s + QA < NSTEP) ldq(s + QA);
                if (tt == 0 && ks + 1 < NKS) ldp(ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, pf[ks & 1]),
                                                                  __builtin_bit_cast(bf16x8_t, qf[s % (QA + 1)]),
                                                                  acc[tt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- prologue
    if (grp == 0 && n_my > 0) load_tile(tile_of(0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    unsigned long long tcyc[4] = {0, 0, 0, 0};
    for (int ph = 0; ph < n_my; ++ph) {
        unsigned long long t0 = 0;
        if constexpr (TIMING) t0 = __builtin_readcyclecounter();
        if ((ph & 1) == grp) {
            compute();
            if constexpr (TIMING) tcyc[0] += __builtin_readcyclecounter() - t0;
        } else {
            if (ph + 1 < n_my && !(TIMING && p.dbg_noload)) load_tile(tile_of(ph + 1));
            if constexpr (TIMING) tcyc[1] += __builtin_readcyclecounter() - t0;
        }
        if constexpr (TIMING) t0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (TIMING) { tcyc[2] += __builtin_readcyclecounter() - t0; t0 = __builtin_readcyclecounter(); }
        __syncthreads();
        if constexpr (TIMING) tcyc[3] += __builtin_readcyclecounter() - t0;
    }
    if constexpr (TIMING) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && wave == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) g_k3wg_cycles[k] = tcyc[k];
        }
    }

    // ---- group 1 -> LDS, group 0 adds and writes the workgroup's slab
    // 32x32 accumulator layout: register e of lane l = row (cout) (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), column (cin) l & 31
    float* xch = (float*)smem;
    const int col = lane & 31, rb = 4 * (lane >> 5);
    const int tap0 = wq * TAPW;
    if (grp == 1) {
#pragma unroll
        for (int tt = 0; tt < TAPW; ++tt) {
            if (tap0 + tt < 27) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    xch[((tap0 + tt) * 32 + (e & 3) + 8 * (e >> 2) + rb) * 32 + col] = acc[tt][e];
            }
        }
    }
    __syncthreads();
    if (grp == 0) {
        float* slab = p.slabs + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * SLAB_FLOATS;
#pragma unroll
        for (int tt = 0; tt < TAPW; ++tt) {
            if (tap0 + tt < 27) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int idx = ((tap0 + tt) * 32 + (e & 3) + 8 * (e >> 2) + rb) * 32 + col;
                    slab[idx] = acc[tt][e] + xch[idx];
                }
            }
        }
    }
}

}  // namespace

bool msseg_k3wg_pp_eligible(const K3WgParams& p) {
    static const bool off = getenv("MSSEG_NO_K3PP") != nullptr;
    if (off) return false;
    if (p.M % 8 || p.K % 8) return false;   // 16-byte channel chunks; partial 32-blocks are zero-filled
    if ((p.ldp % 8) || (p.ldq % 8) || ((uintptr_t)p.pten & 15) || ((uintptr_t)p.qten & 15)) return false;
    const long long ldm = p.ldp > p.ldq ? p.ldp : p.ldq;
    if ((long long)(PD + 1) * p.H * p.W * ldm * 2 >= 0x7fffffffLL) return false;   // 32-bit tile-relative offsets
    const long long tiles = (long long)p.N * ceil_div(p.D, TD) * ceil_div(p.H, TH) * ceil_div(p.W, TW);
    if (tiles > 0x7fffffffLL) return false;
    return tiles * ceil_div(p.M, 32) * ceil_div(p.K, 32) >= 2LL * msseg_num_cus();
}

int msseg_k3wg_pp_grid(const K3WgParams& p) {
    const int pairs = ceil_div(p.M, 32) * ceil_div(p.K, 32);
    const int tiles = p.N * ceil_div(p.D, TD) * ceil_div(p.H, TH) * ceil_div(p.W, TW);
    int gx = msseg_num_cus() / pairs;
    gx &= ~7;
    if (gx < 8) gx = 8;
    if (gx > tiles) gx = tiles;
    return gx;
}

int msseg_k3wg_pp_launch(const K3WgParams& p, int gx, hipStream_t stream) {
    static const bool timing = getenv("MSSEG_K3PP_TIMING") != nullptr;
    static const bool noload = getenv("MSSEG_K3PP_NOLOAD") != nullptr;   // timing experiments only (wrong results)
    K3WgParams pl = p;
    pl.dbg_noload = noload ? 1 : 0;
    const int lds = 2 * GRP_BYTES;
    static msseg_lds_attr_once attr[2];
    if (!attr[0].ensure((const void*)k3wg_pp_kernel<0>, lds) || !attr[1].ensure((const void*)k3wg_pp_kernel<1>, lds))
        MSSEG_FAIL(MSSEG_ELAUNCH, "conv3d_k3_wgrad_pp: cannot set dynamic LDS size %d", lds);
    const int pairs = ceil_div(p.M, 32) * ceil_div(p.K, 32);
    if (timing) hipLaunchKernelGGL(k3wg_pp_kernel<1>, dim3(gx, pairs, 1), dim3(NTHREADS), lds, stream, pl);
    else MSSEG_KTIMED("k3wg_pp_kernel", stream, hipLaunchKernelGGL(k3wg_pp_kernel<0>, dim3(gx, pairs, 1), dim3(NTHREADS), lds, stream, pl));
    MSSEG_CHECK_LAUNCH("conv3d_k3_wgrad_pp");
    return MSSEG_OK;
}

// tools-only: {MFMA role, memory role, vmcnt wait, barrier wait} ticks of workgroup 0 / wave 0 (MSSEG_K3PP_TIMING)
extern "C" int msseg_debug_k3wg_cycles(unsigned long long* out4) {
    return hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_k3wg_cycles), 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
