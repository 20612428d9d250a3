// conv3d 3x3x3 (stride 1, pad 1) on channels-last bf16 -- "ping-pong" implicit GEMM for gfx950.
//
// Replaces torch's Conv3d inside MONAI's Convolution block: MONAI BasicUNet's TwoConv (BASELINE.json configs 1-3; the
// reference ships no UNet, SURVEY.md row A15 -- here medicalsemseg_amd/models/unet.py) and the 3x3x3 convolutions of
// UnetResBlock in the reference's UNETR decoder (/root/reference/models/segmentors/swin_unetr.py:73-128), for the layers
// with 32 input channels per stage: the 96^3 / 48^3 levels that hold ~85 % of the UNet's FLOPs.
//
// One persistent workgroup of 8 waves per CU, split into two groups of 4 waves (one wave per SIMD each).  The groups
// work on different output tiles and alternate roles every phase:
//
//     phase p     group (p & 1)      : 216 MFMAs per wave on the tile whose halo landed in its LDS buffer
//                 the other group    : issues the LDS-DMA (global_load_lds_dwordx4) loads of ITS next tile's halo,
//                                      then converts / stores the accumulators of the tile it finished in phase p-1
//     s_waitcnt vmcnt(0) ; s_barrier
//
// so that the matrix pipe always has one wave per SIMD feeding it while all memory-side work (halo fill, epilogue
// stores, statistics) runs on the other wave of the SIMD.  No staging registers, no LDS write pass, one barrier per
// tile.  LDS: 27 x 32 x 32 bf16 weights (54 KB, loaded once per workgroup) + one 6x6x18-voxel x 32-channel halo
// image per group (2 x 41 KB) + optional statistics slots.
//
// Tile = 4 x 4 x 16 output voxels per group; wave w of a group owns depth slice w: 4 rows (h) of 16 voxels (w).
// For a fixed (kd, kw) one halo row feeds up to three output rows (kh = 0..2): each activation fragment is read from
// LDS once per (kd, kw) and the three kh weight fragments stay in registers, i.e. 54 + 54 ds_read_b128 per 216 MFMAs.
//
// MFMA operand roles: A = weights (rows = cout), B = activations (cols = voxels), so a lane holds 4 consecutive
// couts of one voxel: 8-byte channels-last stores.
//
// Halo image (round 3): voxel-major [halo voxel][64 B] -- the 64 lanes of one LDS-DMA instruction fetch 16 voxels x 4 chunks,
// i.e. 1 KB of consecutive memory on a dense 32-channel tensor (8 cache lines; the chunk-planar image of rounds 1-2 had a
// lane per voxel: 32 lines per instruction, and the instruction's issue cost follows the lines).  A 64-byte voxel pitch
// alone puts voxels v and v + 4 on the same banks; chunk q of voxel v therefore sits in slot q ^ 2*bit2(v), applied on the
// DMA's SOURCE chunk (the LDS side of a DMA is lane-linear): the 16 lanes of a ds_read_b128 group (voxels b .. b+15 at two
// adjacent q) then cover the 16 slots of a 256-byte bank window exactly once for every b.  bit2(v) of a read at
// lane voxel + constant offset depends on (offset mod 8): eight per-lane base addresses, the offset stays an immediate.
#include "k3pp.h"

#include <type_traits>

namespace {

constexpr int TD = 4, TH = 4, TW = 16;
constexpr int PD = TD + 2, PH = TH + 2, PW = TW + 2;
constexpr int HV = PD * PH * PW;                        // 648 halo voxels
constexpr int PLANE = ((HV * 16 + 255) / 256) * 256;    // one 16-byte channel chunk of every halo voxel
constexpr int HALO_BYTES = 4 * PLANE;
constexpr int W_BYTES = 27 * 4 * 32 * 16;
constexpr int STAT_FLOATS = 8 * MSSEG_STATS_NMAX * 32 * 2;
constexpr int NTHREADS = 512;

__device__ u32x4_t g_zero_chunk;                        // source of padding voxels
__device__ unsigned long long g_k3pp_cycles[8];

MSSEG_DEVFN void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

struct TileCo { int n, d0, h0, w0; };

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
MSSEG_DEVFN unsigned pack_bf16x2(float a, float b) {            // one v_cvt_pk_bf16_f32 (round to nearest even)
    const bf16x2_t v = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(unsigned, v);
}
MSSEG_DEVFN float bf16_of_pair(unsigned u, int hi) {            // element `hi` of a packed pair as f32: one shift / mask
    return __builtin_bit_cast(float, hi ? (u & 0xffff0000u) : (u << 16));
}
MSSEG_DEVFN void acc_add(float& s, float v) { asm("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(v)); }
MSSEG_DEVFN void acc_fma(float& s, float a, float b) { asm("v_fmac_f32 %0, %1, %2" : "+v"(s) : "v"(a), "v"(b)); }

template <int STATS, int TIMING>
__global__ __launch_bounds__(NTHREADS, 1) void k3pp_kernel(const K3ppParams p) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    unsigned char* ldsW = smem;
    unsigned char* ldsH = smem + W_BYTES;
    float* ldsS = (float*)(smem + W_BYTES + 2 * HALO_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    const int coutblk = blockIdx.y;
    const bf16_t* __restrict__ xg = (const bf16_t*)p.x;
    bf16_t* __restrict__ yg = (bf16_t*)p.y;

    // ---- tile schedule: each XCD (workgroup id mod 8) walks one contiguous eighth of the tile list, so that
    // neighbouring tiles, which share halo voxels, are fetched through the same L2 at about the same time
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

    // ---- halo fill: wave wq of a group issues DMA pieces wq, wq + 4, ... (1 KB each = 16 halo voxels x 4 slots).
    // The memory role shares its SIMD with a wave that issues MFMAs back to back, so its VALU instructions get an
    // issue slot only every few cycles: everything per-lane is precomputed (byte offsets relative to the tile's halo
    // origin), the per-tile part is scalar, and an interior tile costs no vector ALU work at all per load.
    constexpr int NPIECE = (HV + 15) / 16;              // 41
    constexpr int NIT_P = (NPIECE + 3) / 4;             // 11 per wave
    unsigned h_off[NIT_P];
#pragma unroll
    for (int it = 0; it < NIT_P; ++it) {
        const int hv = (wq + 4 * it) * 16 + (lane >> 2);
        const int hd = hv / (PH * PW), rem = hv - hd * (PH * PW), hh = rem / PW, hw = rem - hh * PW;
        const int chunk = (lane & 3) ^ (((hv >> 2) & 1) << 1);
        h_off[it] = (unsigned)((((long long)hd * p.H + hh) * p.W + hw) * p.ldx * 2 + chunk * 16);
    }
    auto piece_ok = [&](int it) {                       // piece exists / lane's voxel inside the halo
        return (wq + 4 * it) * 16 + (lane >> 2) < HV;
    };
    auto load_halo = [&](const TileCo& tc) {
        unsigned char* dst = ldsH + grp * HALO_BYTES + wq * 1024;
        const int dB = tc.d0 - 1, hB = tc.h0 - 1, wB = tc.w0 - 1;
        const long long vox = (((long long)tc.n * p.D + dB) * p.H + hB) * p.W + wB;
        const unsigned char* hbase = (const unsigned char*)xg + vox * p.ldx * 2;
        const bool interior = dB >= 0 && dB + PD <= p.D && hB >= 0 && hB + PH <= p.H && wB >= 0 && wB + PW <= p.W;
        if (interior) {
#pragma unroll
            for (int it = 0; it < NIT_P; ++it) {
                if (piece_ok(it)) glds16(hbase + h_off[it], dst + it * 4096);
            }
        } else {
            const unsigned char* zsrc = (const unsigned char*)&g_zero_chunk;
#pragma unroll
            for (int it = 0; it < NIT_P; ++it) {
                const int hv = (wq + 4 * it) * 16 + (lane >> 2);
                const int hd = hv / (PH * PW), rem = hv - hd * (PH * PW), hh = rem / PW, hw = rem - hh * PW;
                const bool inb = (unsigned)(dB + hd) < (unsigned)p.D && (unsigned)(hB + hh) < (unsigned)p.H &&
                                 (unsigned)(wB + hw) < (unsigned)p.W;
                const unsigned char* src = inb ? hbase + h_off[it] : zsrc;
                if (piece_ok(it)) glds16(src, dst + it * 4096);
            }
        }
    };

    // ---- prologue: weights (all waves), first halo (group 0), statistics slots
    {
        const unsigned char* wsrc = (const unsigned char*)p.wp + (long long)coutblk * W_BYTES;
        for (int it = wave; it < W_BYTES / 1024; it += 8) glds16(wsrc + (it * 64 + lane) * 16, ldsW + it * 1024);
        if (STATS != 0) {
            for (int i = tid; i < STAT_FLOATS; i += NTHREADS) ldsS[i] = 0.f;
        }
        if (grp == 0 && n_my > 0) load_halo(tile_of(0));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    unsigned long long tcyc[6] = {0, 0, 0, 0, 0, 0};
    f32x4_t acc[TH][2];
    // the bias is the accumulators' initial value: the epilogue (memory role, which shares its SIMD's vector issue
    // with the other group's MFMAs) has no add left to do
    f32x4_t bv[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = *(const f32x4_t*)(p.bias + coutblk * 32 + j * 16 + q * 4);
    }

    // ---- the MFMA role ------------------------------------------------------------------------------
    auto compute = [&]() {
#pragma unroll
        for (int m = 0; m < TH; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[m][j] = bv[j];
        // lane voxel (wq * PH) * PW + r of the image; xb[k] = its address for a read at a voxel offset == k (mod 8)
        const int v0 = (wq * PH) * PW + r;
        const unsigned char* xb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            xb[k] = ldsH + grp * HALO_BYTES + v0 * 64 + ((q ^ ((((v0 + k) >> 2) & 1) << 1)) * 16);
        const unsigned char* wb = ldsW + (q * 32 + r) * 16;
        constexpr int NSTEP = 9 * PH;       // (kd, kw) x halo row
        constexpr int XAHEAD = 3;           // activation fragments in flight ahead of their MFMAs
        u32x4_t wf[2][3][2], xf[XAHEAD + 1];
        auto ldw = [&](int g, int i) {      // i = kh * 2 + j of (kd, kw) group g
            const int kh = i >> 1, j = i & 1;
            const int tap = (g / 3) * 9 + kh * 3 + (g % 3);
            wf[g & 1][kh][j] = *(const u32x4_t*)(wb + tap * 2048 + j * 256);
        };
        auto ldx = [&](int s) {             // s = g * PH + hr
            const int g = s / PH, hr = s % PH;
            const int off = ((g / 3) * PH + hr) * PW + (g % 3);
            xf[s % (XAHEAD + 1)] = *(const u32x4_t*)(xb[off & 7] + off * 64);
        };
#pragma unroll
        for (int i = 0; i < 6; ++i) ldw(0, i);
#pragma unroll
        for (int s = 0; s < XAHEAD; ++s) ldx(s);
#pragma unroll
        for (int g = 0; g < 9; ++g) {
#pragma unroll
            for (int hr = 0; hr < PH; ++hr) {
                const int s = g * PH + hr;
                if (s + XAHEAD < NSTEP) ldx(s + XAHEAD);
                if (g + 1 < 9) ldw(g + 1, hr);          // next group's six weight fragments, one per step
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int m = hr - kh;
                    if (m >= 0 && m < TH) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            mma_chunk<bf16_t>(acc[m][j], wf[g & 1][kh][j], xf[s % (XAHEAD + 1)]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- the memory role: accumulators of a finished tile -> bias, bf16, global (+ fused reductions) ------------
    // Same rule as the halo fill: per-lane byte offsets relative to the tile's first output voxel are precomputed,
    // the tile's base pointers are scalar, full tiles run without any masking.
    unsigned o_off[TH], ny_off[TH], na_off[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) {
        const long long rel = ((long long)wq * p.H + m) * p.W + r;
        o_off[m] = (unsigned)(rel * p.ldy * 2 + q * 8);
        ny_off[m] = (unsigned)(rel * p.nb_ldy * 2 + q * 8);
        na_off[m] = (unsigned)(rel * p.nb_lda * 2 + q * 8);
    }
    // running (sum, sum2) of sample s_n: per lane across tiles, reduced over the 16 voxel lanes (fixed butterfly
    // order) into this wave's private LDS slot when the sample changes and at the end: deterministic, no atomics
    float s1[2][4], s2[2][4];
    int s_n = -1;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[j][e] = s2[j][e] = 0.f;
    auto flush_stats = [&]() {
        if (s_n < 0) return;
        float* slot = ldsS + ((wave * MSSEG_STATS_NMAX + s_n) * 32) * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = s1[j][e], b = s2[j][e];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    a += __shfl_xor(a, o);
                    b += __shfl_xor(b, o);
                }
                if (r == 0) {
                    float* sp = slot + (j * 16 + q * 4 + e) * 2;
                    sp[0] += a;
                    sp[1] += b;
                }
                s1[j][e] = s2[j][e] = 0.f;
            }
    };
    auto epilogue = [&](const TileCo& tc) {
        if constexpr (STATS != 0) {
            if (tc.n != s_n) { flush_stats(); s_n = tc.n; }
        }
        const long long vox = (((long long)tc.n * p.D + tc.d0) * p.H + tc.h0) * p.W + tc.w0;
        unsigned char* ybase = (unsigned char*)yg + (vox * p.ldy + coutblk * 32) * 2;
        const unsigned char* nyb = (const unsigned char*)p.nb_y + (vox * p.nb_ldy + coutblk * 32) * 2;
        const unsigned char* nab = (const unsigned char*)p.nb_a + (vox * p.nb_lda + coutblk * 32) * 2;
        const bool full = tc.d0 + TD <= p.D && tc.h0 + TH <= p.H && tc.w0 + TW <= p.W;
        auto body = [&](auto fullc) {
            constexpr bool FULL = decltype(fullc)::value;
            const bool okdw = FULL || (tc.d0 + wq < p.D && tc.w0 + r < p.W);
            u32x2_t y4[TH][2], a4[TH][2];
            if constexpr (STATS == 2) {
#pragma unroll
                for (int m = 0; m < TH; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bool ok = FULL || (okdw && tc.h0 + m < p.H);
                        y4[m][j] = a4[m][j] = u32x2_t{0u, 0u};
                        if (ok) {
                            y4[m][j] = *(const u32x2_t*)(nyb + ny_off[m] + j * 32);
                            a4[m][j] = *(const u32x2_t*)(nab + na_off[m] + j * 32);
                        }
                    }
            }
            // Everything below is vector-ALU work of a wave whose SIMD is issuing the other group's MFMAs: a 16x16x32
            // MFMA leaves two 4-cycle issue slots per gap, and packed-f32 instructions cost several slots each
            // (MI355X_MICROARCH.md, cycle constants).  So: ONE v_cvt_pk_bf16_f32 per output pair, the rounded values
            // recovered from the packed word by a shift / a mask, and plain v_add_f32 / v_fmac_f32 accumulation (inline
            // asm: the compiler would pair them into v_pk_add_f32).
            // all stores first (they are what the phase's closing s_waitcnt vmcnt(0) waits for), the sums afterwards
            u32x2_t ob[TH][2];
            if constexpr (STATS == 3) {
                // accumulate mode: this launch's sums are added to the values already stored in y (the first K half of a
                // 64-input-channel layer run as two 32-channel launches); all loads first
                u32x2_t yo[TH][2];
#pragma unroll
                for (int m = 0; m < TH; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bool ok = FULL || (okdw && tc.h0 + m < p.H);
                        yo[m][j] = ok ? *(const u32x2_t*)(ybase + o_off[m] + j * 32) : u32x2_t{0u, 0u};
                    }
#pragma unroll
                for (int m = 0; m < TH; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[m][j][e] += bf16_of_pair(yo[m][j][e >> 1], e & 1);
            }
#pragma unroll
            for (int m = 0; m < TH; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool ok = FULL || (okdw && tc.h0 + m < p.H);
                    const f32x4_t o = acc[m][j];
                    ob[m][j] = u32x2_t{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    if (ok) *(u32x2_t*)(ybase + o_off[m] + j * 32) = ob[m][j];
                }
            if constexpr (STATS != 0) {
#pragma unroll
                for (int m = 0; m < TH; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bool ok = FULL || (okdw && tc.h0 + m < p.H);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = bf16_of_pair(ob[m][j][e >> 1], e & 1);   // the value as stored
                            if (!FULL) v = ok ? v : 0.f;
                            if constexpr (STATS == 1 || STATS == 3) {
                                acc_add(s1[j][e], v);
                                acc_fma(s2[j][e], v, v);
                            } else {
                                // dz = da * lrelu'(a); accumulate (sum dz, sum dz * yraw); xhat is formed by the finalising block
                                const float av = bf16_of_pair(a4[m][j][e >> 1], e & 1);
                                const float dz = v * (av > 0.f ? 1.0f : p.nb_slope);
                                acc_add(s1[j][e], dz);
                                acc_fma(s2[j][e], dz, bf16_of_pair(y4[m][j][e >> 1], e & 1));
                            }
                        }
                    }
            }
        };
        if (full) body(std::true_type{});
        else body(std::false_type{});
    };

    // ---- phases ----------------------------------------------------------------------------------------
    for (int ph = 0; ph <= n_my; ++ph) {
        unsigned long long t0 = 0;
        if constexpr (TIMING) t0 = __builtin_readcyclecounter();
        if ((ph & 1) == grp) {
            if (ph < n_my) compute();
            if constexpr (TIMING) tcyc[0] += __builtin_readcyclecounter() - t0;
        } else {
            if (ph + 1 < n_my) load_halo(tile_of(ph + 1));
            if constexpr (TIMING) {
                tcyc[4] += __builtin_readcyclecounter() - t0; t0 = __builtin_readcyclecounter();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                tcyc[5] += __builtin_readcyclecounter() - t0; t0 = __builtin_readcyclecounter();
            }
            if (ph >= 1) epilogue(tile_of(ph - 1));
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
            for (int k = 0; k < 6; ++k) g_k3pp_cycles[k] = tcyc[k];
        }
    }

    // ---- fused reductions: waves -> workgroup partial row; rows are added by msseg_k3_stats_finalize (fixed order) ----
    if constexpr (STATS != 0) {
        flush_stats();
        __syncthreads();
        const int PN = p.N * 32 * 2;
        float* wsp = p.stats_ws + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * PN;
        for (int i = tid; i < PN; i += NTHREADS) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) s += ldsS[wv * MSSEG_STATS_NMAX * 64 + i];
            wsp[i] = s;
        }
    }
}

template <int STATS, int TIMING> int launch(const K3ppParams& p, hipStream_t stream) {
    const int lds = W_BYTES + 2 * HALO_BYTES + (STATS ? STAT_FLOATS * 4 : 0);
    auto kern = k3pp_kernel<STATS, TIMING>;
    static msseg_lds_attr_once attr;
    if (!attr.ensure((const void*)kern, lds)) MSSEG_FAIL(MSSEG_ELAUNCH, "conv3d_k3_pp: cannot set dynamic LDS size %d", lds);
    const int ncb = p.M / 32;
    const int tiles = p.N * ceil_div(p.D, TD) * ceil_div(p.H, TH) * ceil_div(p.W, TW);
    int gx = msseg_num_cus() / ncb;
    gx &= ~7;
    if (gx < 8) gx = 8;
    if (gx > tiles) gx = tiles;
    MSSEG_KTIMED("k3pp_kernel", stream, hipLaunchKernelGGL(kern, dim3(gx, ncb, 1), dim3(NTHREADS), lds, stream, p));
    MSSEG_CHECK_LAUNCH("conv3d_k3_pp");
    if (STATS != 0) {
        K3FinParams f{};
        f.ws = p.stats_ws; f.R = gx; f.N = p.N; f.coutb = 32; f.M = p.M; f.stats = p.stats;
        f.nb_stats = (STATS == 2) ? p.nb_stats : nullptr; f.nb_eps = p.nb_eps; f.nb_S = p.nb_S;
        f.nb_dgamma = p.nb_dgamma; f.nb_dbeta = p.nb_dbeta; f.nb_acc = p.nb_acc;
        return msseg_k3_stats_finalize(f, ncb, stream);
    }
    return MSSEG_OK;
}

}  // namespace

bool msseg_k3pp_eligible(const K3ppParams& p) {
    static const bool off = getenv("MSSEG_NO_K3PP") != nullptr;
    if (off) return false;
    if (p.K != 32 || p.M % 32 || p.M > 256) return false;
    if ((p.ldx % 8) || (p.ldy % 4) || ((uintptr_t)p.x & 15) || ((uintptr_t)p.y & 7)) return false;
    if (p.bias && ((uintptr_t)p.bias & 15)) return false;
    if (p.stats && p.N > MSSEG_STATS_NMAX) return false;
    if (p.nb_y && ((p.nb_ldy % 4) || (p.nb_lda % 4) || ((uintptr_t)p.nb_y & 7) || ((uintptr_t)p.nb_a & 7))) return false;
    const long long tiles = (long long)p.N * ceil_div(p.D, TD) * ceil_div(p.H, TH) * ceil_div(p.W, TW);
    if (tiles > 0x7fffffffLL) return false;
    // two tiles per workgroup are the minimum for the two groups to overlap at all
    return tiles * (p.M / 32) >= 2LL * msseg_num_cus();
}

int msseg_k3pp_launch(const K3ppParams& p, hipStream_t stream) {
    static const bool timing = getenv("MSSEG_K3PP_TIMING") != nullptr;
    if (p.accumulate) {
        if (p.stats == nullptr || p.nb_y != nullptr) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_pp: accumulate mode comes with forward statistics");
        return launch<3, 0>(p, stream);
    }
    if (p.stats == nullptr) return timing ? launch<0, 1>(p, stream) : launch<0, 0>(p, stream);
    if (p.nb_y == nullptr) return timing ? launch<1, 1>(p, stream) : launch<1, 0>(p, stream);
    return launch<2, 0>(p, stream);
}

// tools-only: cycle counters of the MSSEG_K3PP_TIMING build (workgroup 0, wave 0):
// {MFMA role, epilogue, final vmcnt wait, barrier wait, halo load issue, halo load wait}
extern "C" int msseg_debug_k3pp_cycles(unsigned long long* out8) {
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_k3pp_cycles), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
