// conv3d 3x3x3 (stride 1, pad 1) on channels-last bf16, 48 input channels per launch -- ping-pong implicit GEMM for gfx950.
//
// The 3x3x3 convolutions of MONAI's UnetResBlock in the Swin-UNETR decoder / encoder blocks at embedding width 48
// (/root/reference/models/segmentors/swin_unetr.py:73-128: 48 -> 48 and cat(48, 48) -> 48 at 96^3 and 48^3) and their
// input gradients.  The 32-channel kernel (conv3d_k3_pp.hip) does not fit them and the generic kernel (igemm_fwd.hip) pads
// 48 input channels to two 32-channel stages and 48 output channels to two 32-wide blocks: 44 % of its MFMAs multiply
// zeros.  This kernel spends 41.5 k-steps of 32 where 40.5 are needed and no padding on the output side.
//
// Same execution scheme as conv3d_k3_pp.hip: one persistent workgroup of 8 waves per CU in two groups of 4 waves that
// alternate roles every phase -- one group issues MFMAs on the tile whose halo sits in its LDS buffer, the other issues
// the LDS-DMA loads of its next halo and stores the tile it finished.  Differences:
//
//   * K = 48 does not divide into 32-wide MFMA k-steps per tap.  The contraction is reordered instead: channels 0..31 of
//     every tap are 27 ordinary k-steps ("main"); channels 32..47 are packed TWO TAPS per k-step ("rest"): lane quarters
//     0, 1 carry channels 32..39 / 40..47 of tap A, quarters 2, 3 the same channels of tap B.  The B operand is read from
//     LDS per lane anyway, so the two halves of a wave simply read at different voxel shifts -- the shift is folded into
//     a per-lane base address, the reads stay `ds_read_b128 base + immediate`.  Tap pairs are (kw 0, kw 1) of every
//     (kd, kh), (kd 0, kd 1) at kw = 2, and (kd 2, kw 2) alone (half a k-step of zeros): 14 k-step groups of 3 (kh)
//     instead of 13.5.  Pairs never mix kh, so the row reuse of the 32-channel kernel survives: one activation fragment per
//     (group, halo row) feeds up to three output rows.
//   * one workgroup computes a 16-wide cout block (grid.y = Cout / 16): 48 = 3 blocks, no 32 + 16 split.  The halo holds
//     6 channel-chunk planes (61.5 KB per group); what is left of the 160 KB takes 30 of the block's 42 weight
//     fragments (1 KB each), the other 12 stay in registers for the life of the workgroup.
//
// Weight image = four msseg_pack_weights images back to back (hip.pack_conv_k3_c48): main [blk][27 taps][q][16][16 B],
// rest1 [blk][kd * 3 + kh][..] (kw pair), rest2 [blk][kh][..] (kd pair at kw = 2), rest3 [blk][kh][..] (kd 2, kw 2).
#include "k3pp.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int TD = 4, TH = 4, TW = 16;
constexpr int PD = TD + 2, PH = TH + 2, PW = TW + 2;
constexpr int HV = PD * PH * PW;                        // 648 halo voxels
constexpr int VB = 96;                                  // bytes of one halo voxel in LDS: its 48 channels, as in memory
constexpr int ROWC = PW * 6;                            // 16-byte chunks of one halo line (18 voxels)
constexpr int NCHUNK = HV * 6;                          // 3888
constexpr int NPIECE = (NCHUNK + 63) / 64;              // 61 LDS-DMA instructions (1 KB each) per halo
constexpr int HALO_BYTES = NPIECE * 1024;               // 62,464: the image (62,208 B) + the pad the last piece's spare lanes hit
constexpr int NPW = (NPIECE + 3) / 4;                   // 16 per wave of a group
constexpr int NGRP = 14;                                // k-step groups: 9 main (kd, kw), 5 rest
constexpr int NGRP_LDS = 10;                            // groups whose weight fragments live in LDS
constexpr int NRES = (NGRP - NGRP_LDS) * 3;             // register-resident fragments
constexpr int W_BYTES = NGRP_LDS * 3 * 1024;
constexpr int NMAX = 4;                                 // samples with fused statistics
constexpr int STAT_FLOATS = 8 * NMAX * 16 * 2;
constexpr int NTHREADS = 512;
constexpr int LDS_BYTES = W_BYTES + 2 * HALO_BYTES;
static_assert(LDS_BYTES + STAT_FLOATS * 4 <= 160 * 1024, "LDS budget");

__device__ u32x4_t g_c48_zero_chunk;                    // source of padding voxels
__device__ unsigned long long g_c48_cycles[8];          // MSSEG_K3C48_TIMING build: role cycle counters of workgroup 0

MSSEG_DEVFN void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

struct TileCo { int n, d0, h0, w0; };

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
MSSEG_DEVFN unsigned pack_bf16x2(float a, float b) {
    const bf16x2_t v = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(unsigned, v);
}
MSSEG_DEVFN float bf16_of_pair(unsigned u, int hi) {
    return __builtin_bit_cast(float, hi ? (u & 0xffff0000u) : (u << 16));
}
MSSEG_DEVFN void acc_add(float& s, float v) { asm("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(v)); }
MSSEG_DEVFN void acc_fma(float& s, float a, float b) { asm("v_fmac_f32 %0, %1, %2" : "+v"(s) : "v"(a), "v"(b)); }

// 8-byte global load that the compiler's s_waitcnt insertion does not see: a tracked load would be completed with vmcnt(0) at
// its first use, i.e. together with the LDS-DMA pieces issued after it (the pass does not count across the two kinds).
// The value is NOT valid until wait_loads<>() has passed it through; nothing may touch it in between.
MSSEG_DEVFN void gload8_untracked(u32x2_t& v, const void* ptr) {
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
}
// wait until at most NLATER vector-memory operations issued after the four loads are outstanding (loads return in order)
template <int NLATER> MSSEG_DEVFN void wait_loads(u32x2_t (&a)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(NLATER) : "memory");
}

// offset of weight fragment (group g, kh) of cout block b inside the four-part image (bytes)
MSSEG_DEVFN long long wfrag_off(int g, int kh, int b, int ncb) {
    if (g < 9) return ((long long)b * 27 + (g / 3) * 9 + kh * 3 + (g % 3)) * 1024;
    const long long r1 = (long long)ncb * 27 * 1024, r2 = r1 + (long long)ncb * 9 * 1024, r3 = r2 + (long long)ncb * 3 * 1024;
    if (g < 12) return r1 + ((long long)b * 9 + (g - 9) * 3 + kh) * 1024;
    if (g == 12) return r2 + ((long long)b * 3 + kh) * 1024;
    return r3 + ((long long)b * 3 + kh) * 1024;
}

template <int STATS, int TIMING>
__global__ __launch_bounds__(NTHREADS, 1) void k3c48_kernel(const K3ppParams p) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    unsigned char* ldsW = smem;
    unsigned char* ldsH = smem + W_BYTES;
    float* ldsS = (float*)(smem + W_BYTES + 2 * HALO_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    const int coutblk = blockIdx.y, ncb = gridDim.y;
    const bf16_t* __restrict__ xg = (const bf16_t*)p.x;
    bf16_t* __restrict__ yg = (bf16_t*)p.y;

    // ---- tile schedule (as conv3d_k3_pp.hip): each XCD walks one contiguous eighth of the tile list; the cout blocks of
    // one grid column share the XCD, so that the halo a tile's three (or more) workgroups fetch comes out of one L2
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

    // ---- halo fill.  The LDS image is voxel-major -- [halo voxel][96 B], the voxel's 48 channels as they lie in memory --
    // so that the 64 lanes of one LDS-DMA instruction (1 KB of consecutive LDS) read consecutive 16-byte chunks of a halo
    // line: 9-10 cache lines per instruction where a chunk-planar image (conv3d_k3_pp.hip) touches 48 at this voxel
    // pitch.  96 B = 24 banks: the 16 lanes of a ds_read_b128 group (voxels r .. at chunk q) still hit 64 different banks.
    // Wave wq of a group issues pieces wq, wq + 4, ...: exactly NPW instructions each (the waves that have one piece less
    // repeat their last one; lanes beyond the last chunk load a duplicate into the image's pad), so that the memory role
    // can wait for the loads it issued BEFORE them with a constant vmcnt.
    unsigned h_off[NPW], h_code[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        int P = wq + 4 * k;
        if (P >= NPIECE) P -= 4;
        int i = P * 64 + lane;
        i = i < NCHUNK ? i : NCHUNK - 1;
        const int row = i / ROWC, pos = i - row * ROWC, hw = pos / 6, c = pos - hw * 6;
        const int hd = row / PH, hh = row - hd * PH;
        h_off[k] = (unsigned)((((long long)hd * p.H + hh) * p.W + hw) * p.ldx * 2 + c * 16);
        h_code[k] = (unsigned)(hd | (hh << 8) | (hw << 16));
    }
    // address of the zero chunk in a VGPR pair the compiler cannot rematerialise (it would re-load it from the GOT, with an
    // lgkmcnt wait, before every piece of a boundary tile)
    unsigned long long zaddr = (unsigned long long)(uintptr_t)&g_c48_zero_chunk;
    asm volatile("" : "+v"(zaddr));
    auto load_halo = [&](const TileCo& tc) {
        unsigned char* dst = ldsH + grp * HALO_BYTES;
        const int dB = tc.d0 - 1, hB = tc.h0 - 1, wB = tc.w0 - 1;
        const long long vox = (((long long)tc.n * p.D + dB) * p.H + hB) * p.W + wB;
        const unsigned char* hbase = (const unsigned char*)xg + vox * p.ldx * 2;
        const bool interior = dB >= 0 && dB + PD <= p.D && hB >= 0 && hB + PH <= p.H && wB >= 0 && wB + PW <= p.W;
        if (interior) {
#pragma unroll
            for (int k = 0; k < NPW; ++k) {
                const int P = (wq + 4 * k >= NPIECE) ? wq + 4 * k - 4 : wq + 4 * k;
                glds16(hbase + h_off[k], dst + P * 1024);
            }
        } else {
            const unsigned char* zsrc = (const unsigned char*)(uintptr_t)zaddr;
#pragma unroll
            for (int k = 0; k < NPW; ++k) {
                const int P = (wq + 4 * k >= NPIECE) ? wq + 4 * k - 4 : wq + 4 * k;
                const int hd = h_code[k] & 255, hh = (h_code[k] >> 8) & 255, hw = h_code[k] >> 16;
                const bool inb = (unsigned)(dB + hd) < (unsigned)p.D && (unsigned)(hB + hh) < (unsigned)p.H &&
                                 (unsigned)(wB + hw) < (unsigned)p.W;
                glds16(inb ? hbase + h_off[k] : zsrc, dst + P * 1024);
            }
        }
    };

    // ---- prologue: weights (LDS part by all waves, register part per lane), first halo (group 0), statistics slots
    u32x4_t wres[NRES];
    {
        const unsigned char* wsrc = (const unsigned char*)p.wp;
        for (int s = wave; s < NGRP_LDS * 3; s += 8)
            glds16(wsrc + wfrag_off(s / 3, s % 3, coutblk, ncb) + lane * 16, ldsW + s * 1024);
#pragma unroll
        for (int i = 0; i < NRES; ++i)
            wres[i] = *(const u32x4_t*)(wsrc + wfrag_off(NGRP_LDS + i / 3, i % 3, coutblk, ncb) + lane * 16);
        if (STATS != 0) {
            for (int i = tid; i < STAT_FLOATS; i += NTHREADS) ldsS[i] = 0.f;
        }
        if (grp == 0 && n_my > 0) load_halo(tile_of(0));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4_t acc[TH];
    f32x4_t bv = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *(const f32x4_t*)(p.bias + coutblk * 16 + q * 4);
    // everything loaded so far becomes a plain register value for the compiler HERE: a load still pending in its books at
    // the loop header would turn the first wait inside the MFMA role into vmcnt(0), i.e. a wait for the epilogue operands
    // prefetched at the top of that role
    asm volatile("" : "+v"(bv));
#pragma unroll
    for (int i = 0; i < NRES; ++i) asm volatile("" : "+v"(wres[i]));

    // ---- the MFMA role ------------------------------------------------------------------------------
    auto compute = [&]() {
#pragma unroll
        for (int m = 0; m < TH; ++m) acc[m] = bv;
        const unsigned char* hal = ldsH + grp * HALO_BYTES;
        const int vrow = (wq * PH) * PW + r;
        // per-lane bases: main (plane q), rest with the second tap one voxel / one depth plane further, rest alone
        const unsigned char* xb0 = hal + vrow * VB + q * 16;
        const unsigned char* xb1 = hal + (vrow + (q >> 1)) * VB + (4 + (q & 1)) * 16;
        const unsigned char* xb2 = hal + (vrow + (q >> 1) * PH * PW) * VB + (4 + (q & 1)) * 16;
        const unsigned char* xb3 = hal + vrow * VB + (4 + (q & 1)) * 16;
        const unsigned char* wb = ldsW + lane * 16;
        constexpr int NSTEP = NGRP * PH;    // group x halo row
        constexpr int XAHEAD = 5;           // activation fragments in flight ahead of their MFMAs
        u32x4_t wf[2][3], xf[XAHEAD + 1];
        auto ldw = [&](int g, int kh) {
            if (g < NGRP_LDS) wf[g & 1][kh] = *(const u32x4_t*)(wb + (g * 3 + kh) * 1024);
            else wf[g & 1][kh] = wres[(g - NGRP_LDS) * 3 + kh];
        };
        auto ldx = [&](int s) {             // s = g * PH + hr
            const int g = s / PH, hr = s % PH;
            const unsigned char* a;
            if (g < 9) a = xb0 + (((g / 3) * PH + hr) * PW + (g % 3)) * VB;
            else if (g < 12) a = xb1 + (((g - 9) * PH + hr) * PW) * VB;
            else if (g == 12) a = xb2 + (hr * PW + 2) * VB;
            else a = xb3 + ((2 * PH + hr) * PW + 2) * VB;
            xf[s % (XAHEAD + 1)] = *(const u32x4_t*)a;
        };
#pragma unroll
        for (int i = 0; i < 3; ++i) ldw(0, i);
#pragma unroll
        for (int s = 0; s < XAHEAD; ++s) ldx(s);
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
#pragma unroll
            for (int hr = 0; hr < PH; ++hr) {
                const int s = g * PH + hr;
                if (s + XAHEAD < NSTEP) ldx(s + XAHEAD);
                if (g + 1 < NGRP && hr < 3) ldw(g + 1, hr);   // next group's three weight fragments
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int m = hr - kh;
                    if (m >= 0 && m < TH) mma_chunk<bf16_t>(acc[m], wf[g & 1][kh], xf[s % (XAHEAD + 1)]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- the memory role: accumulators of a finished tile -> bf16, global (+ fused reductions) ------------
    unsigned o_off[TH], ny_off[TH], na_off[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) {
        const long long rel = ((long long)wq * p.H + m) * p.W + r;
        o_off[m] = (unsigned)(rel * p.ldy * 2 + q * 8);
        ny_off[m] = (unsigned)(rel * p.nb_ldy * 2 + q * 8);
        na_off[m] = (unsigned)(rel * p.nb_lda * 2 + q * 8);
    }
    float s1[4], s2[4];
    int s_n = -1;
#pragma unroll
    for (int e = 0; e < 4; ++e) s1[e] = s2[e] = 0.f;
    auto flush_stats = [&]() {
        if (s_n < 0) return;
        float* slot = ldsS + ((wave * NMAX + s_n) * 16) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = s1[e], b = s2[e];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                a += __shfl_xor(a, o);
                b += __shfl_xor(b, o);
            }
            if (r == 0) {
                float* sp = slot + (q * 4 + e) * 2;
                sp[0] += a;
                sp[1] += b;
            }
            s1[e] = s2[e] = 0.f;
        }
    };
    // One memory-role phase.  Order: the 8-byte loads the fused epilogues need for the finished tile `tc` (in-backward sums:
    // yraw and the activation of the receiving layer; accumulate mode: the stored partial sums), THEN the LDS-DMA of the
    // next tile `tn` (the slow part: 150-200 cycles per instruction while the other group's MFMAs and fragment reads run),
    // then a wait for the former only (vmcnt(NPW): loads complete in order), conversion / sums / stores while the halo
    // lands.  Every memory phase issues exactly NPW DMA instructions -- the last one of a group re-loads the tile it just
    // finished into its own, now unused, buffer -- so that the wait's count is a constant.
    auto memory_phase = [&](bool has_prev, const TileCo& tc, bool has_next, const TileCo& tn) {
        if (!has_prev) {
            if (has_next) load_halo(tn);
            return;
        }
        if constexpr (STATS != 0) {
            if (tc.n != s_n) { flush_stats(); s_n = tc.n; }
        }
        const long long vox = (((long long)tc.n * p.D + tc.d0) * p.H + tc.h0) * p.W + tc.w0;
        unsigned char* ybase = (unsigned char*)yg + (vox * p.ldy + coutblk * 16) * 2;
        const bool full = tc.d0 + TD <= p.D && tc.h0 + TH <= p.H && tc.w0 + TW <= p.W;
        auto body = [&](auto fullc) {
            constexpr bool FULL = decltype(fullc)::value;
            const bool okdw = FULL || (tc.d0 + wq < p.D && tc.w0 + r < p.W);
            u32x2_t pre_y[TH], pre_a[TH];
            if constexpr (STATS == 2 || STATS == 3) {
                const unsigned char* nyb = (STATS == 2) ? (const unsigned char*)p.nb_y + (vox * p.nb_ldy + coutblk * 16) * 2 : ybase;
                const unsigned char* nab = (const unsigned char*)p.nb_a + (vox * p.nb_lda + coutblk * 16) * 2;
#pragma unroll
                for (int m = 0; m < TH; ++m) {
                    // voxels outside the volume read the tensor's first element instead: unconditional loads, no select on
                    // a value that has not landed; the sums and stores below mask those voxels
                    const bool ok = FULL || (okdw && tc.h0 + m < p.H);
                    gload8_untracked(pre_y[m], ok ? nyb + (STATS == 2 ? ny_off[m] : o_off[m])
                                                  : (STATS == 2 ? (const unsigned char*)p.nb_y : (const unsigned char*)yg));
                    if constexpr (STATS == 2) gload8_untracked(pre_a[m], ok ? nab + na_off[m] : (const unsigned char*)p.nb_a);
                }
            }
            load_halo(tn);   // the caller passes tn = tc when there is no next tile
            if constexpr (STATS == 2 || STATS == 3) wait_loads<NPW>(pre_y);
            if constexpr (STATS == 2) wait_loads<NPW>(pre_a);
            if constexpr (STATS == 3) {
#pragma unroll
                for (int m = 0; m < TH; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[m][e] += bf16_of_pair(pre_y[m][e >> 1], e & 1);
            }
            u32x2_t ob[TH];
#pragma unroll
            for (int m = 0; m < TH; ++m) {
                const bool ok = FULL || (okdw && tc.h0 + m < p.H);
                const f32x4_t o = acc[m];
                ob[m] = u32x2_t{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                if (ok) *(u32x2_t*)(ybase + o_off[m]) = ob[m];
            }
            if constexpr (STATS != 0) {
#pragma unroll
                for (int m = 0; m < TH; ++m) {
                    const bool ok = FULL || (okdw && tc.h0 + m < p.H);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = bf16_of_pair(ob[m][e >> 1], e & 1);   // the value as stored
                        if (!FULL) v = ok ? v : 0.f;
                        if constexpr (STATS == 1 || STATS == 3) {
                            acc_add(s1[e], v);
                            acc_fma(s2[e], v, v);
                        } else {
                            // dz = da * lrelu'(a); accumulate (sum dz, sum dz * yraw); xhat is formed by the finalising block
                            const float av = bf16_of_pair(pre_a[m][e >> 1], e & 1);
                            const float dz = v * (av > 0.f ? 1.0f : p.nb_slope);
                            acc_add(s1[e], dz);
                            acc_fma(s2[e], dz, bf16_of_pair(pre_y[m][e >> 1], e & 1));
                        }
                    }
                }
            }
        };
        if (full) body(std::true_type{});
        else body(std::false_type{});
    };

    // ---- phases ----------------------------------------------------------------------------------------
    unsigned long long tcyc[6] = {0, 0, 0, 0, 0, 0};
    for (int ph = 0; ph <= n_my; ++ph) {
        unsigned long long t0 = 0;
        if constexpr (TIMING) t0 = __builtin_readcyclecounter();
        if ((ph & 1) == grp) {
            if (ph < n_my) compute();
            if constexpr (TIMING) tcyc[0] += __builtin_readcyclecounter() - t0;
        } else {
            memory_phase(ph >= 1, tile_of(ph >= 1 ? ph - 1 : 0), ph + 1 < n_my, tile_of(ph + 1 < n_my ? ph + 1 : (ph >= 1 ? ph - 1 : 0)));
            if constexpr (TIMING) tcyc[2] += __builtin_readcyclecounter() - t0;
        }
        if constexpr (TIMING) t0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (TIMING) { tcyc[3] += __builtin_readcyclecounter() - t0; t0 = __builtin_readcyclecounter(); }
        __syncthreads();
        if constexpr (TIMING) tcyc[4] += __builtin_readcyclecounter() - t0;
    }
    if constexpr (TIMING) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && wave == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) g_c48_cycles[k] = tcyc[k];
            g_c48_cycles[5] = (unsigned long long)n_my;
        }
    }

    // ---- fused reductions: waves -> workgroup partial row; rows are added by msseg_k3_stats_finalize (fixed order) ----
    if constexpr (STATS != 0) {
        flush_stats();
        __syncthreads();
        const int PN = p.N * 16 * 2;
        float* wsp = p.stats_ws + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * PN;
        for (int i = tid; i < PN; i += NTHREADS) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) s += ldsS[wv * NMAX * 32 + i];
            wsp[i] = s;
        }
    }
}

int grid_x(const K3ppParams& p) {
    const int ncb = p.M / 16;
    const int tiles = p.N * ceil_div(p.D, TD) * ceil_div(p.H, TH) * ceil_div(p.W, TW);
    int gx = msseg_num_cus() / ncb;
    gx &= ~7;
    if (gx < 8) gx = 8;
    if (gx > tiles) gx = tiles;
    return gx;
}

template <int STATS, int TIMING = 0> int launch(const K3ppParams& p, hipStream_t stream) {
    const int lds = LDS_BYTES + (STATS ? STAT_FLOATS * 4 : 0);
    auto kern = k3c48_kernel<STATS, TIMING>;
    static msseg_lds_attr_once attr;
    if (!attr.ensure((const void*)kern, lds)) MSSEG_FAIL(MSSEG_ELAUNCH, "conv3d_k3_c48: cannot set dynamic LDS size %d", lds);
    const int ncb = p.M / 16, gx = grid_x(p);
    MSSEG_KTIMED("k3c48_kernel", stream, hipLaunchKernelGGL(kern, dim3(gx, ncb, 1), dim3(NTHREADS), lds, stream, p));
    MSSEG_CHECK_LAUNCH("conv3d_k3_c48");
    if (STATS != 0) {
        K3FinParams f{};
        f.ws = p.stats_ws; f.R = gx; f.N = p.N; f.coutb = 16; f.M = p.M; f.stats = p.stats;
        f.nb_stats = (STATS == 2) ? p.nb_stats : nullptr; f.nb_eps = p.nb_eps; f.nb_S = p.nb_S;
        f.nb_dgamma = p.nb_dgamma; f.nb_dbeta = p.nb_dbeta; f.nb_acc = p.nb_acc;
        return msseg_k3_stats_finalize(f, ncb, stream);
    }
    return MSSEG_OK;
}

}  // namespace

bool msseg_k3c48_shape_ok(int N, int D, int H, int W, int M) {
    static const bool off = getenv("MSSEG_NO_K3C48") != nullptr || getenv("MSSEG_NO_K3PP") != nullptr;
    if (off) return false;
    if (M % 16 || M < 16 || M > 256 || N < 1 || N > NMAX) return false;
    const long long tiles = (long long)N * ceil_div(D, TD) * ceil_div(H, TH) * ceil_div(W, TW);
    if (tiles > 0x7fffffffLL) return false;
    if ((long long)(PD + 1) * H * W * 256 * 2 >= 0x7fffffffLL) return false;   // 32-bit halo-relative offsets (ldx <= 256)
    // two tiles per workgroup are the minimum for the two groups to overlap at all
    return tiles * (M / 16) >= 2LL * msseg_num_cus();
}

bool msseg_k3c48_eligible(const K3ppParams& p) {
    if (p.K != 48 || !msseg_k3c48_shape_ok(p.N, p.D, p.H, p.W, p.M)) return false;
    if ((p.ldx % 8) || p.ldx > 256 || (p.ldy % 4) || ((uintptr_t)p.x & 15) || ((uintptr_t)p.y & 7)) return false;
    if (p.bias && ((uintptr_t)p.bias & 15)) return false;
    if (p.nb_y && ((p.nb_ldy % 4) || (p.nb_lda % 4) || ((uintptr_t)p.nb_y & 7) || ((uintptr_t)p.nb_a & 7))) return false;
    return true;
}

int msseg_k3c48_launch(const K3ppParams& p, hipStream_t stream) {
    if (p.accumulate) {
        if (p.stats == nullptr || p.nb_y != nullptr) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_c48: accumulate mode comes with forward statistics");
        return launch<3>(p, stream);
    }
    static const bool timing = getenv("MSSEG_K3C48_TIMING") != nullptr;   // tools/bench_c48.py
    if (p.stats == nullptr) return timing ? launch<0, 1>(p, stream) : launch<0>(p, stream);
    if (p.nb_y == nullptr) return launch<1>(p, stream);
    return timing ? launch<2, 1>(p, stream) : launch<2>(p, stream);
}

// tools-only: cycle counters of the MSSEG_K3C48_TIMING build (workgroup 0, wave 0), summed over its phases:
// {MFMA role, -, memory role (loads, halo DMA issue, conversion, stores), final vmcnt wait, barrier wait, tiles}
extern "C" int msseg_debug_k3c48_cycles(unsigned long long* out8) {
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_c48_cycles), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
