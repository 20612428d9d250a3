// Weight-gradient implicit GEMMs on MFMA (gfx950).
//
//   dW[m][tap][k] = sum_v P[v][m] * Q[v + tap][k]        (P = output-gradient role, Q = input role)
//
// The contraction index is the voxel, so both MFMA operands need "8 consecutive voxels of one channel"
// per lane while HBM holds "all channels of one voxel".  Tiles are staged row-major [voxel][channel] in
// LDS and read with ds_read_b64_tr_b16 (the CDNA4 transposing LDS read) for bf16; the f32 path needs one
// element per lane and reads plain dwords.  One persistent workgroup walks voxel tiles and keeps its
// partial dW for a (m-block, k-block) pair in registers (27 taps split over 4 waves: 7/7/7/6), then writes
// ONE fp32 slab; a second kernel sums the slabs in a fixed order (deterministic) straight into the
// torch-layout gradient.
#include "common.h"
#include "k3pp.h"

#include <stdlib.h>

namespace {

enum { Q_DIRECT = 0, Q_GATHER = 1, Q_DECONV = 2 };

struct WgradParams {
    const void* pten;  // P tensor (channels M), channels-last
    long long ldp;
    const void* qten;  // Q tensor (channels K or source of the gathered virtual channels)
    long long ldq;
    float* slabs;
    int N, D, H, W;    // tiled voxel grid of P (flat: N=D=H=1, W = voxels)
    int ID, IH, IW;    // gather source dims
    int OD, OH, OW;    // per-sample voxel grid dims (flat modes)
    int M, K;          // logical channels
    int mblks, kblks;
    int tiles_d, tiles_h, tiles_w, ntiles;
    int cin, k, s, p;
    int creal;
    int rel32_ok;      // tile-relative element offsets of P and Q fit 32 bits
};

template <typename T, int NTAPS, int TD, int TH, int TW> struct WgCfg {
    static constexpr int ESZ = sizeof(T);
    static constexpr int CBW = 64 / ESZ;       // channels per block: 32 bf16 / 16 f32
    static constexpr int CT = CBW / 16;        // 16-wide MFMA tiles per block
    static constexpr int PAD = (NTAPS == 27) ? 1 : 0;
    static constexpr int PD = TD + 2 * PAD, PH = TH + 2 * PAD, PW = TW + 2 * PAD;
    static constexpr int HV = PD * PH * PW;
    static constexpr int TV = TD * TH * TW;
    // LDS row stride in bytes.  bf16 rows are padded to 96 B (conflict-free transposing reads) when the tile then
    // still fits; the 2x8x16 tile keeps 64-B rows so that TWO workgroups share a CU (62 KB each).
    static constexpr int RS = 64 + ((ESZ == 2 && !(TD == 2 && TH == 8 && TW == 16)) ? 32 : 0);
    static constexpr int P_BYTES = ((TV * RS + 255) / 256) * 256;
    static constexpr int Q_BYTES = ((HV * RS + 255) / 256) * 256;
    static constexpr int LDS_BYTES = P_BYTES + Q_BYTES;
    static constexpr int WAVES = 4;
    static constexpr int NTHREADS = 256;
    static constexpr int TAPW = (NTAPS + WAVES - 1) / WAVES;  // taps per wave (27 -> 7)
    static constexpr int KSTEP = (ESZ == 2) ? 32 : 4;         // voxels per MFMA k-step
    static constexpr int NKS = TV / KSTEP;
    static constexpr int SLAB_FLOATS = NTAPS * CBW * CBW;
    static_assert(TV % KSTEP == 0, "tile voxels must be a multiple of the k-step");
};

MSSEG_DEVFN bf16x4_t lds_tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4_t*)(uintptr_t)(uint32_t)(uintptr_t)p);
}

template <typename T, int NTAPS, int QSRC, int TD, int TH, int TW>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const WgradParams p) {
    using C = WgCfg<T, NTAPS, TD, TH, TW>;
    constexpr int CBW = C::CBW, CT = C::CT, PAD = C::PAD, PH = C::PH, PW = C::PW, HV = C::HV, TV = C::TV;
    constexpr int RS = C::RS, TAPW = C::TAPW, NKS = C::NKS, EPC = DT<T>::EPC;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    unsigned char* ldsP = smem;
    unsigned char* ldsQ = smem + C::P_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mblk = blockIdx.y / p.kblks, kblk = blockIdx.y % p.kblks;
    const T* __restrict__ pg = (const T*)p.pten;
    const T* __restrict__ qg = (const T*)p.qten;
    const int tap0 = (NTAPS == 27) ? wave * TAPW : 0;

    f32x4_t acc[TAPW][CT][CT];
#pragma unroll
    for (int t = 0; t < TAPW; ++t)
#pragma unroll
        for (int a = 0; a < CT; ++a)
#pragma unroll
            for (int b = 0; b < CT; ++b) acc[t][a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // software pipeline: the next tile's P/Q chunks are loaded into registers while this tile's MFMAs run
    constexpr int NIT_P = (TV * 4 + 255) / 256, NIT_Q = (HV * 4 + 255) / 256;
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
    auto load_p = [&](const TileCo& tc, int i) -> u32x4_t {
        const int cq = i & 3, tv = i >> 2;
        const int c = mblk * CBW + cq * EPC;
        const int td = tv / (TH * TW), th = (tv / TW) % TH, tw = tv % TW;
        const int d = tc.d0 + td, h = tc.h0 + th, w = tc.w0 + tw;
        u32x4_t val = {0u, 0u, 0u, 0u};
        if (d < p.D && h < p.H && w < p.W && c < p.M) {
            const long long vox = (((long long)tc.n * p.D + d) * p.H + h) * p.W + w;
            const T* src = pg + vox * p.ldp + c;
            if (c + EPC <= p.M) {
                val = *(const u32x4_t*)src;
            } else {
                alignas(16) T tmp[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) DT<T>::st(&tmp[e], (c + e < p.M) ? DT<T>::ld(src + e) : 0.f);
                val = *(const u32x4_t*)tmp;
            }
        }
        return val;
    };
    auto load_q = [&](const TileCo& tc, int i) -> u32x4_t {
        const int cq = i & 3, hv = i >> 2;
        const int c = kblk * CBW + cq * EPC;
        u32x4_t val = {0u, 0u, 0u, 0u};
        if constexpr (QSRC == Q_DIRECT) {
            const int hw = hv % PW, t2 = hv / PW, hh = t2 % PH, hd = t2 / PH;
            const int d = tc.d0 - PAD + hd, h = tc.h0 - PAD + hh, w = tc.w0 - PAD + hw;
            if (c < p.K && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W) {
                const long long vox = (((long long)tc.n * p.D + d) * p.H + h) * p.W + w;
                val = *(const u32x4_t*)(qg + vox * p.ldq + c);
            }
        } else if constexpr (QSRC == Q_GATHER) {
            const int ov = tc.w0 + hv;
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
                            fv = DT<T>::ld(qg + vox * p.ldq + ci);
                        }
                    }
                    DT<T>::st(&tmp[e], fv);
                }
                val = *(const u32x4_t*)tmp;
            }
        } else {  // Q_DECONV: virtual channel (abc, co) of coarse voxel = fine child abc, channel co
            const int cv = tc.w0 + hv;
            if (cv < p.W && c < p.K) {
                int tt = cv;
                const int cw = tt % p.OW; tt /= p.OW;
                const int ch = tt % p.OH; tt /= p.OH;
                const int cd = tt % p.OD; const int nn = tt / p.OD;
                const int abc = c / p.creal, co = c - abc * p.creal;
                const int fd = 2 * cd + (abc >> 2), fh = 2 * ch + ((abc >> 1) & 1), fw = 2 * cw + (abc & 1);
                const long long vox = (((long long)nn * (2 * p.OD) + fd) * (2 * p.OH) + fh) * (2 * p.OW) + fw;
                val = *(const u32x4_t*)(qg + vox * p.ldq + co);
            }
        }
        return val;
    };
    u32x4_t pp[NIT_P], pq[NIT_Q];
    // tile-invariant per-thread staging offsets (element offset relative to the tile origin + packed coordinates):
    // interior tiles then need one add per chunk instead of ~40 VALU instructions
    int p_rel[NIT_P], p_pk[NIT_P], q_rel[NIT_Q], q_pk[NIT_Q];
    const bool fast = (QSRC == Q_DIRECT) && p.rel32_ok && (p.M % EPC == 0);
    if (fast) {
#pragma unroll
        for (int it = 0; it < NIT_P; ++it) {
            const int i = tid + it * 256;
            const int tv = i >> 2;
            const int td = tv / (TH * TW), th = (tv / TW) % TH, tw = tv % TW;
            p_rel[it] = (int)((((long long)td * p.H + th) * p.W + tw) * p.ldp) + (i & 3) * EPC;
            p_pk[it] = (i < TV * 4) ? (td | (th << 8) | (tw << 16)) : -1;
        }
#pragma unroll
        for (int it = 0; it < NIT_Q; ++it) {
            const int i = tid + it * 256;
            const int hv = i >> 2;
            const int hw = hv % PW, t2 = hv / PW, hh = t2 % PH, hd = t2 / PH;
            q_rel[it] = (int)((((long long)hd * p.H + hh) * p.W + hw) * p.ldq) + (i & 3) * EPC;
            q_pk[it] = (i < HV * 4) ? (hd | (hh << 8) | (hw << 16)) : -1;
        }
    }
    auto fetch = [&](const TileCo& tc) {
        if (fast) {
            const long long pbase = (((long long)tc.n * p.D + tc.d0) * p.H + tc.h0) * p.W + tc.w0;
            const long long qbase = (((long long)tc.n * p.D + tc.d0 - PAD) * p.H + tc.h0 - PAD) * p.W + tc.w0 - PAD;
            const T* pb = pg + pbase * p.ldp + mblk * CBW;
            const T* qb = qg + qbase * p.ldq + kblk * CBW;
            const bool pcok = mblk * CBW + (tid & 3) * EPC < p.M, qcok = kblk * CBW + (tid & 3) * EPC < p.K;
            const bool interior = tc.d0 - PAD >= 0 && tc.d0 + TD + PAD <= p.D && tc.h0 - PAD >= 0 &&
                                  tc.h0 + TH + PAD <= p.H && tc.w0 - PAD >= 0 && tc.w0 + TW + PAD <= p.W;
            if (interior) {
#pragma unroll
                for (int it = 0; it < NIT_P; ++it)
                    pp[it] = (pcok && p_pk[it] >= 0) ? *(const u32x4_t*)(pb + p_rel[it]) : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
                for (int it = 0; it < NIT_Q; ++it)
                    pq[it] = (qcok && q_pk[it] >= 0) ? *(const u32x4_t*)(qb + q_rel[it]) : u32x4_t{0u, 0u, 0u, 0u};
            } else {
#pragma unroll
                for (int it = 0; it < NIT_P; ++it) {
                    const int pk = p_pk[it];
                    const int d = tc.d0 + (pk & 255), h = tc.h0 + ((pk >> 8) & 255), w = tc.w0 + ((pk >> 16) & 255);
                    const bool ok = pcok && pk >= 0 && d < p.D && h < p.H && w < p.W;
                    pp[it] = ok ? *(const u32x4_t*)(pb + p_rel[it]) : u32x4_t{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int it = 0; it < NIT_Q; ++it) {
                    const int pk = q_pk[it];
                    const int d = tc.d0 - PAD + (pk & 255), h = tc.h0 - PAD + ((pk >> 8) & 255), w = tc.w0 - PAD + ((pk >> 16) & 255);
                    const bool ok = qcok && pk >= 0 && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H &&
                                    (unsigned)w < (unsigned)p.W;
                    pq[it] = ok ? *(const u32x4_t*)(qb + q_rel[it]) : u32x4_t{0u, 0u, 0u, 0u};
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NIT_P; ++it) {
            const int i = tid + it * 256;
            pp[it] = (i < TV * 4) ? load_p(tc, i) : u32x4_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int it = 0; it < NIT_Q; ++it) {
            const int i = tid + it * 256;
            pq[it] = (i < HV * 4) ? load_q(tc, i) : u32x4_t{0u, 0u, 0u, 0u};
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < NIT_P; ++it) {
            const int i = tid + it * 256;
            if (i < TV * 4) *(u32x4_t*)(ldsP + (i >> 2) * RS + (i & 3) * 16) = pp[it];
        }
#pragma unroll
        for (int it = 0; it < NIT_Q; ++it) {
            const int i = tid + it * 256;
            if (i < HV * 4) *(u32x4_t*)(ldsQ + (i >> 2) * RS + (i & 3) * 16) = pq[it];
        }
    };

    if ((int)blockIdx.x < p.ntiles) fetch(decode(blockIdx.x));
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        __syncthreads();
        commit();
        __syncthreads();
        if (tile + (int)gridDim.x < p.ntiles) fetch(decode(tile + gridDim.x));
        // ---- MFMA over the tile's voxels ----
        const int ks_begin = (NTAPS == 27) ? 0 : wave;
        const int ks_step = (NTAPS == 27) ? 1 : C::WAVES;
        for (int ks = ks_begin; ks < NKS; ks += ks_step) {
            if constexpr (sizeof(T) == 2) {
                // lane = 16*g + 4*qr + pc : group g covers voxels ks*32 + 8g + {0..7}; this lane supplies the
                // address of block row qr (voxel) and the 4 channels 4*pc.. of the 16-channel tile.
                // Fragments are double-buffered: the transposing reads of tap t+1 are in flight under tap t's MFMAs.
                const int g = lane >> 4, qr = (lane >> 2) & 3, pc = lane & 3;
                int prow[2], qrow[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int tv = ks * 32 + 8 * g + 4 * i + qr;
                    const int td = tv / (TH * TW), th = (tv / TW) % TH, tw = tv % TW;
                    prow[i] = tv * RS + pc * 8;
                    qrow[i] = ((td * PH + th) * PW + tw) * RS + pc * 8;
                }
                auto tr_frag = [&](const unsigned char* base, int r0, int r1) -> u32x4_t {
                    bf16x4_t lo = lds_tr_read(base + r0);
                    bf16x4_t hi = lds_tr_read(base + r1);
                    bf16x8_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    return __builtin_bit_cast(u32x4_t, f);
                };
                auto tap_off = [&](int tap) -> int {
                    if constexpr (NTAPS == 27) {
                        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                        return ((kd * PH + kh) * PW + kw) * RS;
                    } else {
                        return 0;
                    }
                };
                u32x4_t pf[CT], qf[2][CT];
#pragma unroll
                for (int a = 0; a < CT; ++a) pf[a] = tr_frag(ldsP, prow[0] + a * 32, prow[1] + a * 32);
                {
                    const int toff = tap_off(tap0);
#pragma unroll
                    for (int b = 0; b < CT; ++b) qf[0][b] = tr_frag(ldsQ, qrow[0] + toff + b * 32, qrow[1] + toff + b * 32);
                }
#pragma unroll
                for (int tt = 0; tt < TAPW; ++tt) {
                    const int tap = tap0 + tt;
                    if (tap >= NTAPS) break;
                    const int cur = tt & 1;
                    if (tt + 1 < TAPW && tap + 1 < NTAPS) {
                        const int toff = tap_off(tap + 1);
#pragma unroll
                        for (int b = 0; b < CT; ++b)
                            qf[cur ^ 1][b] = tr_frag(ldsQ, qrow[0] + toff + b * 32, qrow[1] + toff + b * 32);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < CT; ++b)
#pragma unroll
                        for (int a = 0; a < CT; ++a) mma_chunk<bf16_t>(acc[tt][a][b], pf[a], qf[cur][b]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                const int r = lane & 15, q = lane >> 4;
                const int tv = ks * 4 + q;
                const int td = tv / (TH * TW), th = (tv / TW) % TH, tw = tv % TW;
                const float pv = *(const float*)(ldsP + tv * RS + r * 4);
                const int qrow = ((td * PH + th) * PW + tw) * RS + r * 4;
#pragma unroll
                for (int tt = 0; tt < TAPW; ++tt) {
                    const int tap = tap0 + tt;
                    if (tap >= NTAPS) break;
                    int toff = 0;
                    if constexpr (NTAPS == 27) {
                        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                        toff = ((kd * PH + kh) * PW + kw) * RS;
                    }
                    const float qv = *(const float*)(ldsQ + qrow + toff);
                    acc[tt][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, qv, acc[tt][0][0], 0, 0, 0);
                }
            }
        }
    }
    // ---- write this workgroup's partial slab ----
    // 27 taps: every wave owns different taps of ONE slab.  1 tap (flat kernels): the four waves hold partial sums of the
    // same CBW x CBW block over different voxels; they are added through LDS (fixed order) so that one slab per workgroup
    // goes to memory instead of four.
    float* slab = p.slabs + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * C::SLAB_FLOATS;
    const int r = lane & 15, q = lane >> 4;
    if constexpr (NTAPS == 27) {
#pragma unroll
        for (int tt = 0; tt < TAPW; ++tt) {
            const int tap = tap0 + tt;
            if (tap >= NTAPS) break;
#pragma unroll
            for (int a = 0; a < CT; ++a)
#pragma unroll
                for (int b = 0; b < CT; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        slab[(tap * CBW + a * 16 + q * 4 + e) * CBW + b * 16 + r] = acc[tt][a][b][e];
        }
    } else {
        __syncthreads();   // operand images are dead
        float* xch = (float*)smem;
#pragma unroll
        for (int a = 0; a < CT; ++a)
#pragma unroll
            for (int b = 0; b < CT; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    xch[wave * (CBW * CBW) + (a * 16 + q * 4 + e) * CBW + b * 16 + r] = acc[0][a][b][e];
        __syncthreads();
        for (int i = tid; i < CBW * CBW; i += 256)
            slab[i] = xch[i] + xch[CBW * CBW + i] + xch[2 * CBW * CBW + i] + xch[3 * CBW * CBW + i];
    }
}

struct ReduceParams {
    const float* slabs;
    float* dw;
    int M, M0, T, K, K0;
    long long s_m1, s_m0, s_t, s_k1, s_k0;
    int mblks, kblks, nslots, cbw, accumulate;
};

// 256 threads = 32 outputs x 8 slot groups: each thread sums every 8th slab, LDS adds the 8 partial sums in a
// fixed order (deterministic), one thread per output writes the torch-layout gradient.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const ReduceParams p) {
    __shared__ float part[8][33];
    const unsigned total = (unsigned)p.M * (unsigned)p.T * (unsigned)p.K;   // < 2^31 (checked by the host)
    const unsigned K = (unsigned)p.K, T = (unsigned)p.T, cbw = (unsigned)p.cbw, KT = K * T;
    const int ol = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const long long slabf = (long long)p.T * p.cbw * p.cbw;
    for (unsigned base = blockIdx.x * 32u; base < total; base += gridDim.x * 32u) {
        const unsigned i = base + ol;
        float s = 0.f;
        unsigned k = 0, t = 0, m = 0;
        if (i < total) {
            m = i / KT;
            const unsigned rem = i - m * KT;
            t = rem / K;
            k = rem - t * K;
            const unsigned mb = m / cbw, ml = m - mb * cbw, kb = k / cbw, kl = k - kb * cbw;
            const float* src = p.slabs + ((long long)(mb * (unsigned)p.kblks + kb) * p.nslots) * slabf + (t * cbw + ml) * cbw + kl;
#pragma unroll 4
            for (int sl = sg; sl < p.nslots; sl += 8) s += src[(long long)sl * slabf];
        }
        part[sg][ol] = s;
        __syncthreads();
        if (sg == 0 && i < total) {
            float tot = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) tot += part[g][ol];
            const unsigned m1 = m / (unsigned)p.M0, m0 = m - m1 * (unsigned)p.M0, k1 = k / (unsigned)p.K0, k0 = k - k1 * (unsigned)p.K0;
            const long long di = (long long)m1 * p.s_m1 + (long long)m0 * p.s_m0 + (long long)t * p.s_t + (long long)k1 * p.s_k1 +
                                 (long long)k0 * p.s_k0;
            p.dw[di] = p.accumulate ? p.dw[di] + tot : tot;
        }
        __syncthreads();
    }
}

// few slabs (small-spatial, many-channel layers): one thread per output, no LDS
__global__ __launch_bounds__(256) void wgrad_reduce_small_kernel(const ReduceParams p) {
    // 32-bit index arithmetic (the host checks M*T*K < 2^31): the 64-bit divisions dominated this kernel
    const unsigned total = (unsigned)p.M * (unsigned)p.T * (unsigned)p.K;
    const unsigned K = (unsigned)p.K, T = (unsigned)p.T, cbw = (unsigned)p.cbw, KT = K * T;
    const long long slabf = (long long)p.T * p.cbw * p.cbw;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned m = i / KT, rem = i - m * KT;
        const unsigned t = rem / K, k = rem - t * K;
        const unsigned mb = m / cbw, ml = m - mb * cbw, kb = k / cbw, kl = k - kb * cbw;
        const float* src = p.slabs + ((long long)(mb * (unsigned)p.kblks + kb) * p.nslots) * slabf + (t * cbw + ml) * cbw + kl;
        float s = 0.f;
#pragma unroll 8
        for (int sl = 0; sl < p.nslots; ++sl) s += src[(long long)sl * slabf];
        const unsigned m1 = m / (unsigned)p.M0, m0 = m - m1 * (unsigned)p.M0, k1 = k / (unsigned)p.K0, k0 = k - k1 * (unsigned)p.K0;
        const long long di = (long long)m1 * p.s_m1 + (long long)m0 * p.s_m0 + (long long)t * p.s_t + (long long)k1 * p.s_k1 +
                             (long long)k0 * p.s_k0;
        p.dw[di] = p.accumulate ? p.dw[di] + s : s;
    }
}

// The same sum for conv-layout gradients (dw[m][k][tap] with the taps contiguous, T <= 32): one block per (m, 32-wide k
// block).  The slab rows [tap][m][k 0..31] are read coalesced (128 B per row and slab), summed over the slabs in slab
// order (the order of the kernel above: bit-identical results), transposed through LDS and written as ONE contiguous
// run of 32 x T floats -- the thread-per-output form above writes with a stride of T floats, i.e. one cache line per
// lane (36 us for the 64 MB gradient of a 768 -> 768 convolution, 14 us for 256 -> 256).
__global__ __launch_bounds__(256) void wgrad_reduce_rows_kernel(const ReduceParams p) {
    __shared__ float tile[32][33];
    const int m = blockIdx.x, kb = blockIdx.y;
    const int cbw = p.cbw, T = p.T;
    const int mb = m / cbw, ml = m - mb * cbw;
    const long long slabf = (long long)T * cbw * cbw;
    const int kl = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    const int kbs = kb * 32 / cbw, klo = kb * 32 - kbs * cbw;   // slab k block and offset inside it (cbw = 16: two rows of 16)
    const int kk = klo + kl, kbb = kbs + kk / cbw, kin = kk % cbw;
    const float* base = p.slabs + ((long long)(mb * p.kblks + kbb) * p.nslots) * slabf + (long long)ml * cbw + kin;
    const bool kok = kb * 32 + kl < p.K;
    for (int t = r0; t < T; t += 8) {
        float s = 0.f;
        if (kok) {
            const float* src = base + (long long)t * cbw * cbw;
#pragma unroll 8
            for (int sl = 0; sl < p.nslots; ++sl) s += src[(long long)sl * slabf];
        }
        tile[t][kl] = s;
    }
    __syncthreads();
    const int nk = min(32, p.K - kb * 32);
    float* dst = p.dw + (long long)m * p.s_m0 + (long long)kb * 32 * T;
    for (int o = threadIdx.x; o < nk * T; o += 256) {
        const int k = o / T, t = o - k * T;
        dst[o] = p.accumulate ? dst[o] + tile[t][k] : tile[t][k];
    }
}

int launch_reduce(const ReduceParams& rp, hipStream_t stream) {
    const long long total = (long long)rp.M * rp.T * rp.K;
    if (total >= 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "wgrad: weight tensor too large");
    // conv layout: dw[m][k][tap], taps contiguous, no two-level channel index
    const bool conv_layout = rp.s_t == 1 && rp.s_k0 == rp.T && rp.K0 == rp.K && rp.M0 == rp.M && rp.T > 1 && rp.T <= 32 &&
                             rp.s_m0 == (long long)rp.K * rp.T && (rp.cbw == 32 || rp.cbw == 16);
    static const bool no_rows = getenv("MSSEG_NO_REDUCE_ROWS") != nullptr;   // A/B switch
    if (rp.nslots <= 32 && conv_layout && !no_rows && ceil_div(rp.K, 32) <= 65535) {
        hipLaunchKernelGGL(wgrad_reduce_rows_kernel, dim3(rp.M, ceil_div(rp.K, 32)), dim3(256), 0, stream, rp);
    } else if (rp.nslots <= 32) {
        int rb = (int)((total + 255) / 256);
        if (rb > 8192) rb = 8192;
        hipLaunchKernelGGL(wgrad_reduce_small_kernel, dim3(rb), dim3(256), 0, stream, rp);
    } else {
        int rb = (int)((total + 31) / 32);
        if (rb > 4096) rb = 4096;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb), dim3(256), 0, stream, rp);
    }
    MSSEG_CHECK_LAUNCH("wgrad_reduce");
    return MSSEG_OK;
}

template <typename T, int NTAPS, int QSRC, int TD, int TH, int TW>
int launch_wg(WgradParams& p, ReduceParams& rp, void* workspace, size_t ws_bytes, hipStream_t stream) {
    using C = WgCfg<T, NTAPS, TD, TH, TW>;
    p.tiles_d = ceil_div(p.D, TD);
    p.tiles_h = ceil_div(p.H, TH);
    p.tiles_w = ceil_div(p.W, TW);
    const long long nt = (long long)p.N * p.tiles_d * p.tiles_h * p.tiles_w;
    if (nt > 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "wgrad: too many tiles");
    p.ntiles = (int)nt;
    p.mblks = ceil_div(p.M, C::CBW);
    p.kblks = ceil_div(p.K, C::CBW);
    {
        const long long ldm = p.ldp > p.ldq ? p.ldp : p.ldq;
        p.rel32_ok = ((long long)(C::PD + 1) * p.H * p.W * ldm < 0x7fffffffLL) ? 1 : 0;
    }
    const int pairs = p.mblks * p.kblks;
    const int wave_slots = 1;   // one slab per workgroup (the flat kernels add their four waves through LDS)
    const size_t slab_bytes = (size_t)C::SLAB_FLOATS * 4;
    long long gx = msseg_num_cus() * ((C::LDS_BYTES > 80 * 1024) ? 1 : 2);
    // flat (1-tap) problems: a tile is 8 MFMAs per wave behind a load -> LDS -> barrier round trip and a slab is 4 KB, so
    // resident workgroups are what hides the latency: as many as the 48 KB of LDS per workgroup allow (MSSEG_WG_FLAT_PER_CU)
    static const int flat_per_cu = getenv("MSSEG_WG_FLAT_PER_CU") ? atoi(getenv("MSSEG_WG_FLAT_PER_CU")) : 3;
    if (NTAPS == 1 && C::LDS_BYTES * flat_per_cu <= 160 * 1024) gx = (long long)msseg_num_cus() * flat_per_cu;
    // Small grids (the 24^3 ... 6^3 levels): every workgroup writes a 110 KB slab whatever it computed, so a full chip
    // of workgroups moves 56 MB of slabs (written, then read by the reduction) for a few GFLOP.  Half a workgroup per CU
    // measured fastest there (sweep 64 ... 512 in profiles/README.md): 30-34 -> 21-25 us per layer.
    static const int wgs_env = getenv("MSSEG_WG_TOTAL") ? atoi(getenv("MSSEG_WG_TOTAL")) : 0;   // A/B: total workgroups
    if (NTAPS == 27 && (long long)p.ntiles * pairs <= 4LL * msseg_num_cus()) gx = wgs_env > 0 ? wgs_env : msseg_num_cus() / 2;
    if (pairs > 1) gx = (gx + pairs - 1) / pairs;
    if (gx > p.ntiles) gx = p.ntiles;
    const long long fit = (long long)(ws_bytes / (slab_bytes * pairs * wave_slots));
    if (fit < 1) MSSEG_FAIL(MSSEG_EWORKSPACE, "wgrad: workspace %zu B too small (need >= %zu)", ws_bytes,
                            slab_bytes * pairs * wave_slots);
    if (gx > fit) gx = fit;
    if (gx < 1) gx = 1;
    p.slabs = (float*)workspace;
    auto kern = igemm_wgrad_kernel<T, NTAPS, QSRC, TD, TH, TW>;
    static msseg_lds_attr_once attr;
    if (!attr.ensure((const void*)kern, C::LDS_BYTES)) MSSEG_FAIL(MSSEG_ELAUNCH, "wgrad: cannot set dynamic LDS size %d", C::LDS_BYTES);
    MSSEG_KTIMED(NTAPS == 27 ? "igemm_wgrad_kernel<27>" : "igemm_wgrad_kernel<flat>", stream,
                 hipLaunchKernelGGL(kern, dim3((unsigned)gx, pairs, 1), dim3(256), C::LDS_BYTES, stream, p));
    MSSEG_CHECK_LAUNCH("igemm_wgrad");
    rp.slabs = p.slabs;
    rp.mblks = p.mblks; rp.kblks = p.kblks; rp.nslots = (int)gx * wave_slots; rp.cbw = C::CBW;
    return launch_reduce(rp, stream);
}

template <typename T> int launch_wg_k3(WgradParams& p, ReduceParams& rp, void* ws, size_t wsb, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        // large grids with 32-multiple channel counts: the ping-pong kernel (conv3d_k3_wgrad_pp.hip)
        K3WgParams pp{};
        pp.pten = p.pten; pp.ldp = p.ldp; pp.qten = p.qten; pp.ldq = p.ldq;
        pp.N = p.N; pp.D = p.D; pp.H = p.H; pp.W = p.W; pp.M = p.M; pp.K = p.K; pp.kblks = ceil_div(p.K, 32);
        // grids below 8 voxels per axis fill a quarter of the ping-pong kernel's 4x4x16 tiles: generic kernel (2x4x8 tiles)
        static const int pp_min_dim = getenv("MSSEG_K3WG_MINDIM") ? atoi(getenv("MSSEG_K3WG_MINDIM")) : 8;   // A/B
        const int mnd = p.D < p.H ? (p.D < p.W ? p.D : p.W) : (p.H < p.W ? p.H : p.W);
        if (mnd >= pp_min_dim && msseg_k3wg_pp_eligible(pp)) {
            const int pairs = ceil_div(p.M, 32) * ceil_div(p.K, 32);
            int gx = msseg_k3wg_pp_grid(pp);
            const long long fit = (long long)(wsb / ((size_t)27 * 32 * 32 * 4 * pairs));
            if (fit >= 8) {
                if (gx > fit) gx = (int)(fit & ~7LL);
                pp.slabs = (float*)ws;
                const int rc = msseg_k3wg_pp_launch(pp, gx, st);
                if (rc) return rc;
                rp.slabs = pp.slabs;
                rp.mblks = ceil_div(p.M, 32); rp.kblks = ceil_div(p.K, 32); rp.nslots = gx; rp.cbw = 32;
                return launch_reduce(rp, st);
            }
        }
    }
    const int mn = p.D < p.H ? (p.D < p.W ? p.D : p.W) : (p.H < p.W ? p.H : p.W);
    if (mn >= 32) {
        static const bool two_wg = getenv("MSSEG_WGRAD_2WG") != nullptr;   // A/B switch
        if (two_wg) return launch_wg<T, 27, Q_DIRECT, 2, 8, 16>(p, rp, ws, wsb, st);
        return launch_wg<T, 27, Q_DIRECT, 4, 8, 16>(p, rp, ws, wsb, st);
    }
    if (mn >= 12) return launch_wg<T, 27, Q_DIRECT, 4, 4, 8>(p, rp, ws, wsb, st);
    return launch_wg<T, 27, Q_DIRECT, 2, 4, 8>(p, rp, ws, wsb, st);
}

int wg_check(const void* a, long long lda, const void* b, long long ldb, const void* dw, const void* ws, int dtype) {
    if (!a || !b || !dw || !ws) MSSEG_FAIL(MSSEG_EINVAL, "wgrad: null pointer");
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "wgrad: bad dtype");
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)ws) & 15) || (lda * esz) % 16 || (ldb * esz) % 16)
        MSSEG_FAIL(MSSEG_EINVAL, "wgrad: operands must be 16-byte aligned with 16-byte row strides");
    return MSSEG_OK;
}

}  // namespace

// one-pass weight gradient of ConvTranspose3d k2 s2 (linear_wgrad.hip)
bool msseg_lwg_deconv_ok(int dtype, long long NV, int Cin, int Cout, const void* x, long long ldx, const void* dy, long long lddy);
int msseg_lwg_deconv_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw, int N, int D, int H, int W,
                           int Cin, int Cout, int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream);

extern "C" {

size_t msseg_wgrad_workspace_bytes(int M, int T, int K) {
    // worst case over dtypes: f32 blocks are 16 wide, bf16 32 wide; sized for a full-chip persistent grid
    const size_t padM = (size_t)((M + 31) / 32) * 32, padK = (size_t)((K + 31) / 32) * 32;
    const size_t per_slot = padM * padK * (size_t)T * 4;
    size_t slots = 256;
    // large-channel layers live on small grids: bound the scratch at ~256 MiB
    while (slots > 8 && per_slot * slots * (T == 1 ? 4 : 1) > ((size_t)256 << 20)) slots /= 2;
    return per_slot * slots * (T == 1 ? 4 : 1);
}

int msseg_conv3d_k3_wgrad_kernel(int N, int D, int H, int W, int Cin, int Cout, int dtype) {
    if (dtype != MSSEG_BF16) return 0;
    K3WgParams pp{};
    pp.pten = (const void*)256; pp.ldp = Cout; pp.qten = (const void*)256; pp.ldq = Cin;
    pp.N = N; pp.D = D; pp.H = H; pp.W = W; pp.M = Cout; pp.K = Cin; pp.kblks = ceil_div(Cin, 32);
    return msseg_k3wg_pp_eligible(pp) ? 3 : 0;
}

int msseg_conv3d_k3_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw, int N, int D, int H,
                          int W, int Cin, int Cout, int accumulate, void* workspace, size_t workspace_bytes, int dtype,
                          msseg_stream_t stream) {
    int rc = wg_check(x, ldx, dy, lddy, dw, workspace, dtype);
    if (rc) return rc;
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    if (Cin % (16 / esz)) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k3_wgrad: Cin %% %d != 0", 16 / esz);
    WgradParams p{};
    p.pten = dy; p.ldp = lddy; p.qten = x; p.ldq = ldx;
    p.N = N; p.D = D; p.H = H; p.W = W; p.M = Cout; p.K = Cin;
    ReduceParams rp{};
    rp.dw = dw; rp.M = Cout; rp.M0 = Cout; rp.T = 27; rp.K = Cin; rp.K0 = Cin;
    rp.s_m1 = 0; rp.s_m0 = (long long)Cin * 27; rp.s_t = 1; rp.s_k1 = 0; rp.s_k0 = 27; rp.accumulate = accumulate;
    return dtype == MSSEG_F32 ? launch_wg_k3<float>(p, rp, workspace, workspace_bytes, (hipStream_t)stream)
                              : launch_wg_k3<bf16_t>(p, rp, workspace, workspace_bytes, (hipStream_t)stream);
}

int msseg_conv3d_k1_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw, long long NV, int Cin,
                          int Cout, int accumulate, void* workspace, size_t workspace_bytes, int dtype,
                          msseg_stream_t stream) {
    int rc = wg_check(x, ldx, dy, lddy, dw, workspace, dtype);
    if (rc) return rc;
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    if (Cin % (16 / esz) || NV < 1 || NV > 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_wgrad: bad shape");
    WgradParams p{};
    p.pten = dy; p.ldp = lddy; p.qten = x; p.ldq = ldx;
    p.N = 1; p.D = 1; p.H = 1; p.W = (int)NV; p.M = Cout; p.K = Cin;
    ReduceParams rp{};
    rp.dw = dw; rp.M = Cout; rp.M0 = Cout; rp.T = 1; rp.K = Cin; rp.K0 = Cin;
    rp.s_m0 = Cin; rp.s_k0 = 1; rp.accumulate = accumulate;
    return dtype == MSSEG_F32
               ? launch_wg<float, 1, Q_DIRECT, 1, 1, 256>(p, rp, workspace, workspace_bytes, (hipStream_t)stream)
               : launch_wg<bf16_t, 1, Q_DIRECT, 1, 1, 256>(p, rp, workspace, workspace_bytes, (hipStream_t)stream);
}

int msseg_conv3d_gather_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw, int N, int ID,
                              int IH, int IW, int Cin, int Cout, int k, int s, int pd, int accumulate, void* workspace,
                              size_t workspace_bytes, int dtype, msseg_stream_t stream) {
    if (!x || !dy || !dw || !workspace) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_gather_wgrad: null pointer");
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_gather_wgrad: bad dtype");
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    if (((uintptr_t)dy & 15) || (lddy * esz) % 16) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_gather_wgrad: dy alignment");
    if (k < 1 || s < 1 || pd < 0) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_gather_wgrad: bad kernel");
    const int OD = (ID + 2 * pd - k) / s + 1, OH = (IH + 2 * pd - k) / s + 1, OW = (IW + 2 * pd - k) / s + 1;
    const long long NV = (long long)N * OD * OH * OW;
    if (OD < 1 || OH < 1 || OW < 1 || NV > 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_gather_wgrad: bad shape");
    const int KT = k * k * k;
    if (msseg_stem_eligible(dtype, Cin, Cout, k, s, pd, ldx, lddy, dy) && (lddy % 8) == 0 &&
        workspace_bytes >= (size_t)8 * 4096 * ceil_div(Cout, 32)) {
        StemWgParams sp{};
        sp.x = x; sp.ldx = ldx; sp.dy = dy; sp.lddy = lddy; sp.N = N; sp.D = ID; sp.H = IH; sp.W = IW; sp.M = Cout;
        int gx = msseg_stem_wgrad_grid(sp);
        const long long fit = (long long)(workspace_bytes / ((size_t)4096 * ceil_div(Cout, 32)));
        if (gx > fit) gx = (int)(fit & ~7LL);
        sp.slabs = (float*)workspace;
        int rc = msseg_stem_wgrad_launch(sp, gx, (hipStream_t)stream);
        if (rc) return rc;
        ReduceParams rq{};
        rq.dw = dw; rq.M = Cout; rq.M0 = Cout; rq.T = 1; rq.K = KT; rq.K0 = 1;
        rq.s_m0 = KT; rq.s_k1 = 1; rq.s_k0 = KT; rq.accumulate = accumulate;
        // k == 1 (the 1x1x1 conv of a one-channel volume): the slabs hold all 27 taps, the gradient is the centre one
        rq.slabs = sp.slabs + (k == 1 ? 13 : 0); rq.mblks = ceil_div(Cout, 32); rq.kblks = 1; rq.nslots = gx; rq.cbw = 32;
        return launch_reduce(rq, (hipStream_t)stream);
    }
    WgradParams p{};
    p.pten = dy; p.ldp = lddy; p.qten = x; p.ldq = ldx;
    p.N = 1; p.D = 1; p.H = 1; p.W = (int)NV; p.M = Cout; p.K = Cin * KT;
    p.ID = ID; p.IH = IH; p.IW = IW; p.OD = OD; p.OH = OH; p.OW = OW; p.cin = Cin; p.k = k; p.s = s; p.p = pd;
    ReduceParams rp{};
    // logical k = tap*Cin + ci  ->  torch [Cout][Cin][KT]
    rp.dw = dw; rp.M = Cout; rp.M0 = Cout; rp.T = 1; rp.K = Cin * KT; rp.K0 = Cin;
    rp.s_m0 = (long long)Cin * KT; rp.s_k1 = 1; rp.s_k0 = KT; rp.accumulate = accumulate;
    return dtype == MSSEG_F32
               ? launch_wg<float, 1, Q_GATHER, 1, 1, 256>(p, rp, workspace, workspace_bytes, (hipStream_t)stream)
               : launch_wg<bf16_t, 1, Q_GATHER, 1, 1, 256>(p, rp, workspace, workspace_bytes, (hipStream_t)stream);
}

int msseg_deconv_k2s2_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw, int N, int D, int H,
                            int W, int Cin, int Cout, int accumulate, void* workspace, size_t workspace_bytes,
                            int dtype, msseg_stream_t stream) {
    int rc = wg_check(x, ldx, dy, lddy, dw, workspace, dtype);
    if (rc) return rc;
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    const long long NV = (long long)N * D * H * W;
    if (Cout % (16 / esz) || NV < 1 || NV > 0x7fffffffLL / 8) MSSEG_FAIL(MSSEG_EINVAL, "deconv_k2s2_wgrad: bad shape");
    if (msseg_lwg_deconv_ok(dtype, NV, Cin, Cout, x, ldx, dy, lddy))   // one pass over whole rows (linear_wgrad.hip)
        return msseg_lwg_deconv_wgrad(x, ldx, dy, lddy, dw, N, D, H, W, Cin, Cout, accumulate, workspace, workspace_bytes,
                                      (hipStream_t)stream);
    WgradParams p{};
    p.pten = x; p.ldp = ldx; p.qten = dy; p.ldq = lddy;
    p.N = 1; p.D = 1; p.H = 1; p.W = (int)NV; p.M = Cin; p.K = 8 * Cout;
    p.OD = D; p.OH = H; p.OW = W; p.creal = Cout;
    ReduceParams rp{};
    // logical [m = ci][k = abc*Cout + co] -> torch ConvTranspose3d weight [Cin][Cout][2][2][2]
    rp.dw = dw; rp.M = Cin; rp.M0 = Cin; rp.T = 1; rp.K = 8 * Cout; rp.K0 = Cout;
    rp.s_m0 = (long long)Cout * 8; rp.s_k1 = 1; rp.s_k0 = 8; rp.accumulate = accumulate;
    return dtype == MSSEG_F32
               ? launch_wg<float, 1, Q_DECONV, 1, 1, 256>(p, rp, workspace, workspace_bytes, (hipStream_t)stream)
               : launch_wg<bf16_t, 1, Q_DECONV, 1, 1, 256>(p, rp, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
