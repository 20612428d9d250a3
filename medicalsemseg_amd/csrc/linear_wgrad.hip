// Weight and bias gradient of a Linear layer on many tokens (bf16) in one pass over the tokens, for gfx950.
//
//   dW[m][k] = sum_t dy[t][m] * x[t][k]        db[m] = sum_t dy[t][m]
//
// Replaces, for nn.Linear in the Swin stages (qkv / proj / fc1 / fc2 of /root/reference/models/backbones/swin_nnformer.py:
// 24-42,128-196; 221 k tokens x 48 ... 192 channels in the first stage), autograd's weight.grad = dy^T x and
// bias.grad = dy.sum(0).  The generic flat kernel (igemm_wgrad.hip) gives every (32 x 32) output block pair its own
// workgroup column, each of which walks ALL tokens reading its 64-byte slices of the rows -- ten columns for 48 -> 144,
// 87 us where the tensors are 85 MB (10.6 us at 8 TB/s) -- and the bias gradient was one more pass over dy
// (msseg_channel_sum, 28 us).  Here a workgroup reads WHOLE token rows (its slice of the output: NTP x NTQ 16-wide
// tiles, up to 192 x 48 / 48 x 192 / 96 x 96, usually the whole layer), keeps its partial dW in registers across all
// its token chunks, gets db from one more MFMA per row tile against a fragment of ones, and writes ONE partial block;
// a second small kernel adds the workgroups' blocks in a fixed order (deterministic) into the torch-layout gradients.
//
// Per 128-token chunk: 256 threads load the rows (16-byte pieces, consecutive lanes on consecutive pieces of a row) into
// registers while the previous chunk is computed, then store them as [32-channel block][token][96 B] images (64 B of data,
// 32 B pad: the pitch that makes the transposing reads conflict-free, as in igemm_wgrad.hip); wave w takes tokens
// 32 w .. 32 w + 31 as its MFMA k-step: both operands are read with ds_read_b64_tr_b16 (contraction index = token,
// memory holds token rows), NTP + NTQ fragments for NTP x NTQ (+ NTP) MFMAs.
#include "common.h"

#include <stdlib.h>

namespace {

constexpr int TT = 128;            // tokens per chunk
constexpr int RS = 96;             // LDS row pitch of a 32-channel block image
constexpr int BLK_BYTES = TT * RS;
constexpr int LW_THREADS = 256;

struct LwgParams {
    const void* x; long long ldx;      // [NV][K]
    const void* dy; long long lddy;    // [NV][M]
    float* part;                       // [slice][workgroup][ [k: NTQ*16][m: NTP*16] + bias [NTP*16] ] fp32
    long long NV;
    int M, K, mslices, kslices, nchunks;
    // DCV (weight gradient of ConvTranspose3d k2 s2): dy is the FINE tensor [N, 2D, 2H, 2W, cout]; the row of coarse token t is
    // the 8 child rows (m = abc * cout + co), gathered while staging
    int D, H, W, cout;
};

MSSEG_DEVFN bf16x4_t lds_tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4_t*)(uintptr_t)(uint32_t)(uintptr_t)p);
}
MSSEG_DEVFN u32x4_t tr_frag(const unsigned char* base, int r0, int r1) {
    const bf16x4_t lo = lds_tr_read(base + r0), hi = lds_tr_read(base + r1);
    const bf16x8_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(u32x4_t, f);
}

template <int NTP, int NTQ, bool DCV = false>
__global__ __launch_bounds__(LW_THREADS, 1) void lwg_kernel(const LwgParams p) {
    constexpr int NBP = (NTP + 1) / 2, NBQ = (NTQ + 1) / 2;       // 32-channel block images
    constexpr int CHP = NTP * 2, CHQ = NTQ * 2;                   // 16-byte pieces per row of the slice
    constexpr int NLP = (TT * CHP + LW_THREADS - 1) / LW_THREADS; // staged pieces per thread
    constexpr int NLQ = (TT * CHQ + LW_THREADS - 1) / LW_THREADS;
    constexpr int PART = NTP * 16 * NTQ * 16 + NTP * 16;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    unsigned char* ldsP = smem;
    unsigned char* ldsQ = smem + NBP * BLK_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ms = blockIdx.y / p.kslices, ks = blockIdx.y - ms * p.kslices;
    const bool with_bias = ks == 0;
    const unsigned char* pg = (const unsigned char*)p.dy + (long long)ms * NTP * 32;
    const unsigned char* qg = (const unsigned char*)p.x + (long long)ks * NTQ * 32;

    // ---- staging: piece i of the chunk = (row i / CH, 16-byte piece i % CH of the slice's part of the row); the offsets are
    // recomputed per chunk (constant divisions) instead of held in 3 registers per piece: the accumulators need the room
    u32x4_t sp[NLP], sq[NLQ];
    __shared__ unsigned fbase[2][DCV ? TT : 1];   // DCV: fine voxel (2d, 2h, 2w) of the chunk's tokens (fine voxel count < 2^31)
    auto fill_table = [&](int chunk, int sel) {
        if constexpr (DCV) {
            if (tid < TT) {
                unsigned t = (unsigned)chunk * (unsigned)TT + (unsigned)tid;
                if ((long long)t >= p.NV) t = 0;
                const unsigned w = t % (unsigned)p.W; t /= (unsigned)p.W;
                const unsigned h = t % (unsigned)p.H; t /= (unsigned)p.H;
                const unsigned d = t % (unsigned)p.D, n = t / (unsigned)p.D;
                fbase[sel][tid] = ((n * 2u * p.D + 2u * d) * 2u * p.H + 2u * h) * 2u * p.W + 2u * w;
            }
        }
    };
    auto fetch = [&](int chunk, int tsel) {
        const long long t0 = (long long)chunk * TT;
        const int rows = (p.NV - t0) < TT ? (int)(p.NV - t0) : TT;
        const unsigned char* pb = pg + t0 * p.lddy * 2;
        const unsigned char* qb = qg + t0 * p.ldx * 2;
#pragma unroll
        for (int it = 0; it < NLP; ++it) {
            const int i = tid + it * LW_THREADS, row = i / CHP, c = i - row * CHP;
            if constexpr (DCV) {
                // piece c of the slice = channels (ms * NTP * 2 + c) * 8 .. of the gathered row: child abc, channel co; the
                // row's fine base voxel comes from the table the first 128 threads filled for this chunk
                const unsigned ch = (unsigned)(ms * NTP * 2 + c) * 8u;
                const unsigned abc = ch / (unsigned)p.cout, co = ch - abc * (unsigned)p.cout;
                const unsigned fv = fbase[tsel][row < TT ? row : 0] + ((abc >> 2) * 2u * p.H + ((abc >> 1) & 1u)) * 2u * p.W + (abc & 1u);
                sp[it] = row < rows ? *(const u32x4_t*)((const unsigned char*)p.dy + ((unsigned long long)fv * (unsigned)p.lddy + co) * 2)
                                    : u32x4_t{0u, 0u, 0u, 0u};
            } else {
                sp[it] = row < rows ? *(const u32x4_t*)(pb + (unsigned)(row * (int)p.lddy * 2 + c * 16)) : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int it = 0; it < NLQ; ++it) {
            const int i = tid + it * LW_THREADS, row = i / CHQ, c = i - row * CHQ;
            sq[it] = row < rows ? *(const u32x4_t*)(qb + (unsigned)(row * (int)p.ldx * 2 + c * 16)) : u32x4_t{0u, 0u, 0u, 0u};
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < NLP; ++it) {
            const int i = tid + it * LW_THREADS, row = i / CHP, c = i - row * CHP;
            if (row < TT) *(u32x4_t*)(ldsP + (c >> 2) * BLK_BYTES + row * RS + (c & 3) * 16) = sp[it];
        }
#pragma unroll
        for (int it = 0; it < NLQ; ++it) {
            const int i = tid + it * LW_THREADS, row = i / CHQ, c = i - row * CHQ;
            if (row < TT) *(u32x4_t*)(ldsQ + (c >> 2) * BLK_BYTES + row * RS + (c & 3) * 16) = sq[it];
        }
    };

    // ---- transposing fragment reads: lane = 16 g + 4 qr + pc supplies row 8 g + 4 i + qr of the wave's 32 tokens, channels
    // 4 pc .. of a 16-channel tile (cdna_hip_programming.md T10)
    const int g = lane >> 4, qr = (lane >> 2) & 3, pc = lane & 3;
    int trow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) trow[i] = (wave * 32 + 8 * g + 4 * i + qr) * RS + pc * 8;

    f32x4_t acc[NTP][NTQ], accb[NTP];
#pragma unroll
    for (int a = 0; a < NTP; ++a) {
        accb[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NTQ; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const u32x4_t ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};   // bf16 1.0 x 8

    if constexpr (DCV) {
        fill_table(blockIdx.x, 0);
        __syncthreads();
    }
    if ((int)blockIdx.x < p.nchunks) fetch(blockIdx.x, 0);
    int tsel = 0;
    for (int chunk = blockIdx.x; chunk < p.nchunks; chunk += gridDim.x) {
        __syncthreads();                 // the previous chunk's fragment reads are done
        commit();
        tsel ^= 1;
        fill_table(chunk + gridDim.x, tsel);   // read by the fetch behind the next barrier; the other half is still the table of
                                               // the pieces in flight ... which were fetched before this point
        __syncthreads();
        if (chunk + (int)gridDim.x < p.nchunks) fetch(chunk + gridDim.x, tsel);
        u32x4_t pf[NTP], qf[NTQ];
#pragma unroll
        for (int a = 0; a < NTP; ++a) pf[a] = tr_frag(ldsP + (a >> 1) * BLK_BYTES, trow[0] + (a & 1) * 32, trow[1] + (a & 1) * 32);
#pragma unroll
        for (int b = 0; b < NTQ; ++b) qf[b] = tr_frag(ldsQ + (b >> 1) * BLK_BYTES, trow[0] + (b & 1) * 32, trow[1] + (b & 1) * 32);
#pragma unroll
        for (int a = 0; a < NTP; ++a) {
#pragma unroll
            for (int b = 0; b < NTQ; ++b) mma_chunk<bf16_t>(acc[a][b], pf[a], qf[b]);
            if (with_bias) mma_chunk<bf16_t>(accb[a], pf[a], ones);
        }
    }

    // ---- the four waves' partial sums (different tokens) meet in LDS, [wave][k][m] so that a lane's four consecutive m are one
    // 16-byte store, and are added in a fixed order: ONE block [k][m] (+ the bias sums) per workgroup goes to memory
    __syncthreads();                     // the images are dead
    float* xch = (float*)smem;
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int a = 0; a < NTP; ++a) {
#pragma unroll
        for (int b = 0; b < NTQ; ++b)
            *(f32x4_t*)(xch + wave * PART + (b * 16 + r) * (NTP * 16) + a * 16 + q * 4) = acc[a][b];
        if (r == 0) *(f32x4_t*)(xch + wave * PART + NTQ * 16 * NTP * 16 + a * 16 + q * 4) = accb[a];
    }
    __syncthreads();
    float* out = p.part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * PART;
    for (int i = tid * 4; i < PART; i += LW_THREADS * 4) {
        const f32x4_t v0 = *(const f32x4_t*)(xch + i), v1 = *(const f32x4_t*)(xch + PART + i),
                      v2 = *(const f32x4_t*)(xch + 2 * PART + i), v3 = *(const f32x4_t*)(xch + 3 * PART + i);
        *(f32x4_t*)(out + i) = (v0 + v1) + (v2 + v3);
    }
}

struct LwgRedParams {
    const float* part;
    float* dw; float* db;
    int M, K, ntp16, ntq16, mslices, kslices, nwg, acc_w, acc_b;
    int dcv_cout;   // > 0: m = abc * cout + co of a transposed conv: dw in the torch layout [K = Cin][cout][8]
};

// 256 threads = 32 outputs x 8 groups: a thread adds every 8th workgroup's partial (8 loads in flight), the 8 group sums are
// added through LDS in a fixed order (deterministic)
__global__ __launch_bounds__(256) void lwg_reduce_kernel(const LwgRedParams p) {
    __shared__ float gs[8][33];
    const int total = p.M * p.K + (p.db ? p.M : 0);
    const int part = p.ntp16 * p.ntq16 + p.ntp16;
    const int ol = threadIdx.x & 31, sg = threadIdx.x >> 5;
    for (int base = blockIdx.x * 32; base < total; base += gridDim.x * 32) {
        const int i = base + ol;
        float s = 0.f;
        bool bias = false;
        int m = 0;
        if (i < total) {
            bias = i >= p.M * p.K;
            m = bias ? i - p.M * p.K : i / p.K;
            const int k = bias ? 0 : i - m * p.K;
            const int ms = m / p.ntp16, ml = m - ms * p.ntp16, ks = k / p.ntq16, kl = k - ks * p.ntq16;
            const float* src = p.part + (long long)(ms * p.kslices + ks) * p.nwg * part + (bias ? p.ntp16 * p.ntq16 + ml : kl * p.ntp16 + ml);
            int w = sg;
            for (; w + 56 < p.nwg; w += 64) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = src[(long long)(w + 8 * j) * part];
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[j];
            }
            for (; w < p.nwg; w += 8) s += src[(long long)w * part];
        }
        gs[sg][ol] = s;
        __syncthreads();
        if (sg == 0 && i < total) {
            float tot = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) tot += gs[j][ol];
            if (bias) {
                p.db[m] = p.acc_b ? p.db[m] + tot : tot;
            } else if (p.dcv_cout > 0) {
                const int k = i - m * p.K, abc = m / p.dcv_cout, co = m - abc * p.dcv_cout;
                float* o = p.dw + ((long long)k * p.dcv_cout + co) * 8 + abc;
                *o = p.acc_w ? *o + tot : tot;
            } else {
                p.dw[i] = p.acc_w ? p.dw[i] + tot : tot;
            }
        }
        __syncthreads();
    }
}

// slice shapes with an instantiation, in order of preference (most output tiles per workgroup first)
struct Shape { int ntp, ntq; };
constexpr Shape kShapes[] = {{12, 3}, {3, 12}, {6, 6}, {9, 3}, {3, 9}, {6, 3}, {3, 6}, {3, 3}};
// transposed convs: 8 * Cout / 16 is a multiple of 4, so a multiple of 3 is one of 12.  Measured (tools/bench_deconv.py, B = 2):
// 48 -> 48 on 221 k coarse voxels 96 -> 63 us against the generic flat kernel, but 64 -> 32 on 27 k voxels 18 -> 28 us and
// 128 -> 64 on 3.5 k 14 -> 21 us (power-of-two shapes {8, 4} / {8, 2}, since removed): with a few chunks per workgroup the
// partial blocks and their reduction cost more than the flat kernel's re-reads.  Only the 48-multiples on large grids come here.
constexpr Shape kShapesDcv[] = {{12, 3}};

bool pick_shape_dcv(int M, int K, Shape* out) {
    if (M % 16 || K % 16) return false;
    const int tm = M / 16, tk = K / 16;
    for (const Shape& s : kShapesDcv)
        if (tm % s.ntp == 0 && tk % s.ntq == 0) { *out = s; return true; }
    return false;
}

bool pick_shape(int M, int K, Shape* out) {
    if (M % 16 || K % 16) return false;
    const int tm = M / 16, tk = K / 16;
    for (const Shape& s : kShapes)
        if (tm % s.ntp == 0 && tk % s.ntq == 0) { *out = s; return true; }
    return false;
}

template <int NTP, int NTQ, bool DCV = false> int launch_lwg(const LwgParams& p, int gx, hipStream_t stream) {
    constexpr int NBP = (NTP + 1) / 2, NBQ = (NTQ + 1) / 2;
    constexpr int PART = NTP * 16 * NTQ * 16 + NTP * 16;
    constexpr int lds = (NBP + NBQ) * BLK_BYTES > 4 * PART * 4 ? (NBP + NBQ) * BLK_BYTES : 4 * PART * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = lwg_kernel<NTP, NTQ, DCV>;
    static msseg_lds_attr_once attr;
    if (!attr.ensure((const void*)kern, lds)) MSSEG_FAIL(MSSEG_ELAUNCH, "linear_wgrad: cannot set dynamic LDS size %d", lds);
    MSSEG_KTIMED(DCV ? "lwg_kernel<deconv>" : "lwg_kernel", stream,
                 hipLaunchKernelGGL(kern, dim3(gx, p.mslices * p.kslices), dim3(LW_THREADS), lds, stream, p));
    MSSEG_CHECK_LAUNCH("linear_wgrad");
    return MSSEG_OK;
}

}  // namespace

// ---- weight gradient of ConvTranspose3d k2 s2 on the one-pass kernel (called by msseg_deconv_k2s2_wgrad, igemm_wgrad.hip):
// dw[Cin][Cout][2][2][2] (+)= sum over coarse voxels of x[v][ci] * dy[child abc of v][co].  The generic flat kernel gave every
// (32 x 32) block pair of the [Cin][8 * Cout] matrix its own workgroup column, each walking all voxels (34 us per BasicUNet
// layer, 64 us per Swin-UNETR layer); here a workgroup reads whole rows -- the coarse row and the 8 child rows it gathers.
bool msseg_lwg_deconv_ok(int dtype, long long NV, int Cin, int Cout, const void* x, long long ldx, const void* dy, long long lddy) {
    static const bool off = getenv("MSSEG_NO_LINEAR_WGRAD") != nullptr || getenv("MSSEG_NO_DECONV_LWG") != nullptr;   // A/B switch
    Shape s;
    if (off || dtype != MSSEG_BF16 || NV < 100000 || NV > 0x7fffffffLL / 8 || Cout % 8) return false;
    if ((((uintptr_t)x | (uintptr_t)dy) & 15) || (ldx % 8) || (lddy % 8) || ldx < Cin || lddy < Cout) return false;
    return pick_shape_dcv(8 * Cout, Cin, &s);
}

int msseg_lwg_deconv_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw, int N, int D, int H, int W,
                           int Cin, int Cout, int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    Shape s;
    if (!pick_shape_dcv(8 * Cout, Cin, &s)) MSSEG_FAIL(MSSEG_EINVAL, "deconv wgrad: shape %d -> %d has no instantiation", Cin, Cout);
    if ((uintptr_t)workspace & 15) MSSEG_FAIL(MSSEG_EINVAL, "deconv wgrad: workspace alignment");
    LwgParams p{};
    p.x = x; p.ldx = ldx; p.dy = dy; p.lddy = lddy; p.NV = (long long)N * D * H * W; p.M = 8 * Cout; p.K = Cin;
    p.D = D; p.H = H; p.W = W; p.cout = Cout;
    p.mslices = p.M / (s.ntp * 16); p.kslices = Cin / (s.ntq * 16);
    p.nchunks = (int)((p.NV + TT - 1) / TT);
    const int slices = p.mslices * p.kslices;
    const size_t part_bytes = (size_t)(s.ntp * 16 * s.ntq * 16 + s.ntp * 16) * 4;
    int gx = msseg_num_cus() / slices;
    if (gx < 1) gx = 1;
    if (gx > (p.nchunks + 1) / 2) gx = (p.nchunks + 1) / 2;
    if (gx < 1) gx = 1;
    const size_t fit = workspace_bytes / (part_bytes * slices);
    if (fit < 1) MSSEG_FAIL(MSSEG_EWORKSPACE, "deconv wgrad: workspace %zu B too small (need >= %zu)", workspace_bytes, part_bytes * slices);
    if ((size_t)gx > fit) gx = (int)fit;
    p.part = (float*)workspace;
    int rc = MSSEG_EINVAL;
    switch (s.ntp * 100 + s.ntq) {
        case 1203: rc = launch_lwg<12, 3, true>(p, gx, stream); break;
    }
    if (rc) return rc;
    LwgRedParams r{};
    r.part = p.part; r.dw = dw; r.db = nullptr; r.M = p.M; r.K = Cin; r.ntp16 = s.ntp * 16; r.ntq16 = s.ntq * 16;
    r.mslices = p.mslices; r.kslices = p.kslices; r.nwg = gx; r.acc_w = accumulate; r.acc_b = 0; r.dcv_cout = Cout;
    int rb = (p.M * Cin + 31) / 32;
    if (rb > 4096) rb = 4096;
    hipLaunchKernelGGL(lwg_reduce_kernel, dim3(rb), dim3(256), 0, stream, r);
    MSSEG_CHECK_LAUNCH("deconv_wgrad_reduce");
    return MSSEG_OK;
}

extern "C" {

int msseg_linear_wgrad_ok(long long NV, int Cin, int Cout, int dtype) {
    static const bool off = getenv("MSSEG_NO_LINEAR_WGRAD") != nullptr;   // A/B switch
    Shape s;
    if (off || dtype != MSSEG_BF16 || NV < 1 || NV > 0x7fffffffLL || Cin > 4096 || Cout > 4096) return 0;
    // a few hundred tokens against a large weight (last Swin stage): the partial blocks dominate, the generic kernel + channel
    // sum measured 3 us faster per layer (tools/bench_linear.py: 432 tokens, 384 -> 1152 / 1536, 1536 -> 384)
    if (NV < 1024 && (long long)Cin * Cout > 200000) return 0;
    return pick_shape(Cout, Cin, &s) ? 1 : 0;
}

int msseg_linear_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw, float* dbias, long long NV,
                       int Cin, int Cout, int accumulate_w, int accumulate_b, void* workspace, size_t workspace_bytes,
                       int dtype, msseg_stream_t stream) {
    if (!x || !dy || !dw || !workspace) MSSEG_FAIL(MSSEG_EINVAL, "linear_wgrad: null pointer");
    if (!msseg_linear_wgrad_ok(NV, Cin, Cout, dtype)) MSSEG_FAIL(MSSEG_EINVAL, "linear_wgrad: shape %d -> %d not supported (msseg_linear_wgrad_ok)", Cin, Cout);
    if ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)workspace) & 15) || (ldx % 8) || (lddy % 8) || ldx < Cin || lddy < Cout)
        MSSEG_FAIL(MSSEG_EINVAL, "linear_wgrad: operands must be 16-byte aligned with 16-byte row strides");
    if ((long long)TT * (ldx > lddy ? ldx : lddy) * 2 >= 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "linear_wgrad: row stride too large");
    Shape s;
    pick_shape(Cout, Cin, &s);
    LwgParams p{};
    p.x = x; p.ldx = ldx; p.dy = dy; p.lddy = lddy; p.NV = NV; p.M = Cout; p.K = Cin;
    p.mslices = Cout / (s.ntp * 16); p.kslices = Cin / (s.ntq * 16);
    p.nchunks = (int)((NV + TT - 1) / TT);
    const int slices = p.mslices * p.kslices;
    const size_t part_bytes = (size_t)(s.ntp * 16 * s.ntq * 16 + s.ntp * 16) * 4;
    // workgroups per slice: the chip divided by the slices, at least two chunks each when there are that many tokens (every
    // workgroup writes a partial block whatever it summed), bounded by the workspace
    int gx = msseg_num_cus() / slices;
    if (gx < 1) gx = 1;
    if (gx > (p.nchunks + 1) / 2) gx = (p.nchunks + 1) / 2;
    if (gx < 1) gx = 1;
    const size_t fit = workspace_bytes / (part_bytes * slices);
    if (fit < 1) MSSEG_FAIL(MSSEG_EWORKSPACE, "linear_wgrad: workspace %zu B too small (need >= %zu)", workspace_bytes, part_bytes * slices);
    if ((size_t)gx > fit) gx = (int)fit;
    p.part = (float*)workspace;
    int rc = MSSEG_EINVAL;
    switch (s.ntp * 100 + s.ntq) {
        case 1203: rc = launch_lwg<12, 3>(p, gx, (hipStream_t)stream); break;
        case 312: rc = launch_lwg<3, 12>(p, gx, (hipStream_t)stream); break;
        case 606: rc = launch_lwg<6, 6>(p, gx, (hipStream_t)stream); break;
        case 903: rc = launch_lwg<9, 3>(p, gx, (hipStream_t)stream); break;
        case 309: rc = launch_lwg<3, 9>(p, gx, (hipStream_t)stream); break;
        case 603: rc = launch_lwg<6, 3>(p, gx, (hipStream_t)stream); break;
        case 306: rc = launch_lwg<3, 6>(p, gx, (hipStream_t)stream); break;
        case 303: rc = launch_lwg<3, 3>(p, gx, (hipStream_t)stream); break;
    }
    if (rc) return rc;
    LwgRedParams r{};
    r.part = p.part; r.dw = dw; r.db = dbias; r.M = Cout; r.K = Cin; r.ntp16 = s.ntp * 16; r.ntq16 = s.ntq * 16;
    r.mslices = p.mslices; r.kslices = p.kslices; r.nwg = gx; r.acc_w = accumulate_w; r.acc_b = accumulate_b;
    const int total = Cout * Cin + (dbias ? Cout : 0);
    int rb = (total + 31) / 32;
    if (rb > 4096) rb = 4096;
    hipLaunchKernelGGL(lwg_reduce_kernel, dim3(rb), dim3(256), 0, (hipStream_t)stream, r);
    MSSEG_CHECK_LAUNCH("linear_wgrad_reduce");
    return MSSEG_OK;
}

}  // extern "C"
