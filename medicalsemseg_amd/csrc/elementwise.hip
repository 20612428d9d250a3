// HBM-bound passes: statistics, InstanceNorm+LeakyReLU (fwd/bwd), MaxPool3d(2), layout changes, channel sums,
// weight packing, AdamW.  All tensors channels-last with an explicit voxel stride; 16-byte vector accesses when
// the channel count allows (C % (16/sizeof(T)) == 0), scalar fallback otherwise.
#include "common.h"

#include <stdarg.h>
#include <string.h>

// ---------------------------------------------------------------------------------------------------------
// error plumbing / device info
// ---------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void msseg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
int msseg_abi_version(void) { return MSSEG_ABI_VERSION; }
const char* msseg_last_error(void) { return g_err; }
int msseg_num_cus(void) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}
}

namespace {

template <typename T> struct Chunk {
    static constexpr int EPC = DT<T>::EPC;
    float v[EPC];
    MSSEG_DEVFN void load(const T* p) {
        if constexpr (sizeof(T) == 4) {
            f32x4_t t = *(const f32x4_t*)p;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = t[e];
        } else {
            bf16x8_t t = *(const bf16x8_t*)p;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
        }
    }
    MSSEG_DEVFN void store(T* p) const {
        if constexpr (sizeof(T) == 4) {
            f32x4_t t = {v[0], v[1], v[2], v[3]};
            *(f32x4_t*)p = t;
        } else {
            bf16x8_t t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
            *(bf16x8_t*)p = t;
        }
    }
};

inline bool vec_ok(const void* p, long long ld, int C, int esz) {
    const int epc = 16 / esz;
    return (C % epc) == 0 && (ld % epc) == 0 && (((uintptr_t)p) & 15) == 0;
}

// Work decomposition shared by the per-(n, voxel, channel-group) passes: a block owns `rows_per_block`
// consecutive voxels of sample blockIdx.y; thread -> (row lane, channel group); channel group fixed per thread.
struct RowMap {
    int groups;       // channel groups per voxel
    int rows_par;     // voxel rows processed in parallel by a block
};
inline RowMap row_map(int C, int width, int nthr = 256) {
    RowMap m;
    m.groups = ceil_div(C, width);
    if (m.groups > nthr) m.groups = nthr;
    m.rows_par = nthr / m.groups;
    if (m.rows_par < 1) m.rows_par = 1;
    return m;
}

// ---------------------------------------------------------------------------------------------------------
// channel statistics: stats[n][c][0..1] += (sum x, sum x^2)
// ---------------------------------------------------------------------------------------------------------
// Finalise per-channel reductions in the last block: `nper` values per (n, c); partials laid out
// ws[(n * nblk + b) * C * nper + c * nper + k].  MODE_OUT 0: out[n][c][k] = sum_b (statistics);
// 1: out[c] (+)= sum_{n,b} (channel sums); 2: red[n][c][k] = sum_b and dparam_k[c] (+)= sum_n red[n][c][k].
MSSEG_DEVFN void finalize_channels(const float* ws, int N, int nblk, int C, int nper, int mode, float* out,
                                   float* dp0, float* dp1, int accumulate) {
    // partial matrix: rows (n, b), L = C*nper floats per row.  Threads tile [row group][column]; every thread keeps
    // UNR loads in flight; row groups are then added in a fixed order through LDS (bit-reproducible).
    __shared__ float fin[256];
    const int L = C * nper;
    const int cols = L < 256 ? L : 256;
    const int G = 256 / cols;               // row groups
    const int col = threadIdx.x % cols, rg = threadIdx.x / cols;
    const bool in256 = threadIdx.x < 256;   // blocks may be larger than the 256 threads this routine tiles
    // blocks split the columns (and, where the samples stay separate -- mode 0 -- the samples over grid.y): with one block
    // a 768-channel statistics row (L = 1536) took six sequential passes per sample
    const int n_step = mode == 0 ? (int)gridDim.y : 1, n_first = mode == 0 ? (int)blockIdx.y : 0;
    for (int c0 = blockIdx.x * cols; c0 < L; c0 += gridDim.x * cols) {
        const int o = c0 + col;
        const bool ok = in256 && o < L && rg < G;
        float tot = 0.f;
        for (int n = n_first; n < N; n += n_step) {
            float s = 0.f;
            if (ok) {
                const float* src = ws + (long long)n * nblk * L + o;
#pragma unroll 16
                for (int bb = rg; bb < nblk; bb += G) s += src[(long long)bb * L];
            }
            __syncthreads();
            if (in256) fin[threadIdx.x] = s;
            __syncthreads();
            if (ok && rg == 0) {
                float t = 0.f;
                for (int g = 0; g < G; ++g) t += fin[g * cols + col];
                if (mode != 1) out[(long long)n * L + o] = t;
                tot += t;
            }
        }
        if (ok && rg == 0) {
            if (mode == 1) {
                out[o] = accumulate ? out[o] + tot : tot;
            } else if (mode == 2) {
                const int c = o / nper, k = o % nper;
                float* dp = (k == 0) ? dp0 : dp1;
                if (dp) dp[c] = accumulate ? dp[c] + tot : tot;
            }
        }
    }
}

// Second step of every channel reduction: 256-thread blocks (one per 256 columns, and per sample for statistics) add the
// per-block partial rows in a fixed order.  A
// separate launch -- the kernel boundary makes the rows visible, where an in-kernel "last block finalises" pays an
// agent-scope release (L2 write-back) per block plus an acquire: ~10-15 us per reduction on these sizes.
struct FinalizeArgs {
    const float* ws; int N, nblk, C, nper, mode; float* out; float* dp0; float* dp1; int accumulate;
};
__global__ __launch_bounds__(256) void channels_finalize_kernel(const FinalizeArgs a) {
    finalize_channels(a.ws, a.N, a.nblk, a.C, a.nper, a.mode, a.out, a.dp0, a.dp1, a.accumulate);
}

static inline dim3 channels_finalize_grid(const FinalizeArgs& a) {
    const int L = a.C * a.nper;
    int gx = (L + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    return dim3((unsigned)gx, a.mode == 0 ? (unsigned)(a.N < 1 ? 1 : a.N) : 1u);
}

// Reduction kernels run 512-thread blocks (8 waves per CU, one block per CU, 12+ loads in flight per thread): enough loads in flight to stream HBM
// while the number of partial rows the finalising block has to add stays at one per CU.
constexpr int RED_THREADS = 512;
constexpr int RED_STAGE2 = 16;

// two-stage fixed-order sum over the rows_par row slots of red[(k * groups + g) * W + e][2]; result in slot k = 0
template <int W>
MSSEG_DEVFN void block_rows_reduce(float* red, int groups, int rows_par, int g, int rl, bool act) {
    if (act && rl < RED_STAGE2 && rl < rows_par) {
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float a = 0.f, b = 0.f;
            for (int k = rl; k < rows_par; k += RED_STAGE2) {
                a += red[((k * groups + g) * W + e) * 2 + 0];
                b += red[((k * groups + g) * W + e) * 2 + 1];
            }
            red[((rl * groups + g) * W + e) * 2 + 0] = a;
            red[((rl * groups + g) * W + e) * 2 + 1] = b;
        }
    }
    __syncthreads();
}

template <typename T, bool VEC>
__global__ __launch_bounds__(RED_THREADS) void channel_stats_kernel(const T* __restrict__ x, long long ldx, float* out,
                                                            long long S, int C, int groups, int rows_par,
                                                            long long rows_per_block, int nacc, int accumulate,
                                                            float* ws, unsigned int* counter) {
    constexpr int W = VEC ? DT<T>::EPC : 1;
    __shared__ float red[RED_THREADS * 2 * (VEC ? DT<T>::EPC : 1)];
    __shared__ int lflag;
    const int n = blockIdx.y;
    const int g = threadIdx.x % groups, rl = threadIdx.x / groups;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > S) r1 = S;
    const T* xn = x + (long long)n * S * ldx;
    float* wsb = ws + ((long long)n * gridDim.x + blockIdx.x) * C * nacc;
    for (int gbase = 0; gbase * W < C; gbase += groups) {   // uniform trip count: barriers inside
        const int gg = gbase + g;
        const bool act = gg * W < C;
        float s[W], s2[W];
#pragma unroll
        for (int e = 0; e < W; ++e) s[e] = s2[e] = 0.f;
        if (act && rl < rows_par) {
#pragma unroll 8
            for (long long r = r0 + rl; r < r1; r += rows_par) {
                if constexpr (VEC) {
                    Chunk<T> c;
                    c.load(xn + r * ldx + gg * W);
#pragma unroll
                    for (int e = 0; e < W; ++e) { s[e] += c.v[e]; s2[e] += c.v[e] * c.v[e]; }
                } else {
                    const float v = DT<T>::ld(xn + r * ldx + gg);
                    s[0] += v; s2[0] += v * v;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < W; ++e) {
            red[(threadIdx.x * W + e) * 2 + 0] = s[e];
            red[(threadIdx.x * W + e) * 2 + 1] = s2[e];
        }
        __syncthreads();
        block_rows_reduce<W>(red, groups, rows_par, g, rl, act);
        if (act && rl == 0) {
            const int kmax = rows_par < RED_STAGE2 ? rows_par : RED_STAGE2;
#pragma unroll
            for (int e = 0; e < W; ++e) {
                float a = 0.f, b = 0.f;
                for (int k = 0; k < kmax; ++k) {
                    a += red[((k * groups + g) * W + e) * 2 + 0];
                    b += red[((k * groups + g) * W + e) * 2 + 1];
                }
                const int c = gg * W + e;
                wsb[c * nacc + 0] = a;
                if (nacc == 2) wsb[c * nacc + 1] = b;
            }
        }
    }
}

// LayerNorm parameter gradients: dgamma[c] = sum_rows dy*xhat, dbeta[c] = sum_rows dy  (xhat from per-row mean/rstd)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const T* __restrict__ x, long long ldx,
                                                            const T* __restrict__ dy, long long lddy,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            long long rows, int C, int groups, int rows_par,
                                                            long long rows_per_block, float* ws, unsigned int* counter,
                                                            float* dgamma, float* dbeta, int accumulate) {
    constexpr int W = VEC ? DT<T>::EPC : 1;
    __shared__ float red[256 * 2 * (VEC ? DT<T>::EPC : 1)];
    __shared__ int lflag;
    const int g = threadIdx.x % groups, rl = threadIdx.x / groups;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float* wsb = ws + (long long)blockIdx.x * C * 2;
    for (int gbase = 0; gbase * W < C; gbase += groups) {
        const int gg = gbase + g;
        const bool act = gg * W < C;
        float s[W], s2[W];
#pragma unroll
        for (int e = 0; e < W; ++e) s[e] = s2[e] = 0.f;
        if (act && rl < rows_par) {
#pragma unroll 4
            for (long long r = r0 + rl; r < r1; r += rows_par) {
                const float mu = mean[r], rs = rstd[r];
                if constexpr (VEC) {
                    Chunk<T> cx, cd;
                    cx.load(x + r * ldx + gg * W);
                    cd.load(dy + r * lddy + gg * W);
#pragma unroll
                    for (int e = 0; e < W; ++e) { s[e] += cd.v[e] * (cx.v[e] - mu) * rs; s2[e] += cd.v[e]; }
                } else {
                    const float d = DT<T>::ld(dy + r * lddy + gg);
                    s[0] += d * (DT<T>::ld(x + r * ldx + gg) - mu) * rs;
                    s2[0] += d;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < W; ++e) {
            red[(threadIdx.x * W + e) * 2 + 0] = s[e];
            red[(threadIdx.x * W + e) * 2 + 1] = s2[e];
        }
        __syncthreads();
        if (act && rl == 0) {
#pragma unroll
            for (int e = 0; e < W; ++e) {
                float a = 0.f, b = 0.f;
                for (int k = 0; k < rows_par; ++k) {
                    a += red[((k * groups + g) * W + e) * 2 + 0];
                    b += red[((k * groups + g) * W + e) * 2 + 1];
                }
                wsb[(gg * W + e) * 2 + 0] = a;
                wsb[(gg * W + e) * 2 + 1] = b;
            }
        }
    }
}

inline long long reduce_blocks(long long S, int rows_par, int N, int C, int nper);

template <typename T>
int launch_ln_param_grad(const void* x, long long ldx, const void* dy, long long lddy, const float* mean, const float* rstd,
                         float* dgamma, float* dbeta, int accumulate, long long rows, int C, void* scratch, hipStream_t st) {
    const bool vec = vec_ok(x, ldx, C, sizeof(T)) && vec_ok(dy, lddy, C, sizeof(T));
    const RowMap m = row_map(C, vec ? DT<T>::EPC : 1);
    long long blocks = reduce_blocks(rows, m.rows_par, 1, C, 2);
    const long long rpb = ceil_div_ll(rows, blocks);
    blocks = ceil_div_ll(rows, rpb);
    unsigned int* counter = (unsigned int*)scratch;
    float* ws = (float*)((unsigned char*)scratch + MSSEG_SCRATCH_COUNTER_BYTES);
    if (vec)
        hipLaunchKernelGGL((ln_param_grad_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, ldx,
                           (const T*)dy, lddy, mean, rstd, rows, C, m.groups, m.rows_par, rpb, ws, counter, dgamma, dbeta, accumulate);
    else
        hipLaunchKernelGGL((ln_param_grad_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, ldx,
                           (const T*)dy, lddy, mean, rstd, rows, C, m.groups, m.rows_par, rpb, ws, counter, dgamma, dbeta, accumulate);
    MSSEG_CHECK_LAUNCH("ln_param_grad");
    {
        FinalizeArgs a{ws, 1, (int)blocks, C, 2, 2, ws + blocks * C * 2, dgamma, dbeta, accumulate};
        hipLaunchKernelGGL(channels_finalize_kernel, channels_finalize_grid(a), dim3(256), 0, st, a);
        MSSEG_CHECK_LAUNCH("channels_finalize");
    }
    return MSSEG_OK;
}

inline long long reduce_blocks(long long S, int rows_par, int N, int C, int nper) {
    // one block per CU in total (each keeps ~32 KB of loads in flight) keeps the finalising block's job small
    long long blocks = ceil_div_ll(S, (long long)rows_par * 8);
    long long cap = (long long)msseg_num_cus() / (N > 0 ? N : 1);
    if (cap < 1) cap = 1;
    const long long fit = (long long)(((size_t)16 << 20) / ((size_t)N * C * nper * 4));
    if (cap > fit) cap = fit;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return blocks;
}

template <typename T>
int launch_stats(const void* x, long long ldx, float* out, int N, long long S, int C, int nacc, int accumulate,
                 void* scratch, hipStream_t st) {
    const bool vec = vec_ok(x, ldx, C, sizeof(T));
    const RowMap m = row_map(C, vec ? DT<T>::EPC : 1, RED_THREADS);
    long long blocks = reduce_blocks(S, m.rows_par, N, C, nacc);
    const long long rpb = ceil_div_ll(S, blocks);
    blocks = ceil_div_ll(S, rpb);
    dim3 grid((unsigned)blocks, N);
    unsigned int* counter = (unsigned int*)scratch;
    float* ws = (float*)((unsigned char*)scratch + MSSEG_SCRATCH_COUNTER_BYTES);
    if (vec)
        hipLaunchKernelGGL((channel_stats_kernel<T, true>), grid, dim3(RED_THREADS), 0, st, (const T*)x, ldx, out, S, C,
                           m.groups, m.rows_par, rpb, nacc, accumulate, ws, counter);
    else
        hipLaunchKernelGGL((channel_stats_kernel<T, false>), grid, dim3(RED_THREADS), 0, st, (const T*)x, ldx, out, S, C,
                           m.groups, m.rows_par, rpb, nacc, accumulate, ws, counter);
    MSSEG_CHECK_LAUNCH("channel_stats");
    {
        FinalizeArgs a{ws, N, (int)blocks, C, nacc, nacc == 2 ? 0 : 1, out, nullptr, nullptr, accumulate};
        hipLaunchKernelGGL(channels_finalize_kernel, channels_finalize_grid(a), dim3(256), 0, st, a);
        MSSEG_CHECK_LAUNCH("channels_finalize");
    }
    return MSSEG_OK;
}

// ---------------------------------------------------------------------------------------------------------
// InstanceNorm + LeakyReLU
// ---------------------------------------------------------------------------------------------------------
struct NormParams {
    const void* x; long long ldx;
    const float* stats; const float* gamma; const float* beta;
    const void* res; long long ldr;
    void* y; long long ldy;
    const void* dy; long long lddy;
    float* red;
    void* dx; long long lddx;
    void* dres; long long lddres;
    long long S; int C; float eps, slope;
    long long rows_per_block; int groups, rows_par;
    float* ws; unsigned int* counter; float* dgamma; float* dbeta; int accumulate;
};

MSSEG_DEVFN void mean_rstd(const float* stats, int n, int C, int c, long long S, float eps, float& mean, float& rstd) {
    const float inv = 1.0f / (float)S;
    const float s = stats[((long long)n * C + c) * 2 + 0], s2 = stats[((long long)n * C + c) * 2 + 1];
    mean = s * inv;
    float var = s2 * inv - mean * mean;
    var = var > 0.f ? var : 0.f;
    rstd = rsqrtf(var + eps);
}

// MODE 0: forward   MODE 1: backward reduce   MODE 2: backward apply
template <typename T, bool VEC, int MODE>
__global__ __launch_bounds__(MODE == 1 ? RED_THREADS : 256) void instnorm_kernel(const NormParams p) {
    constexpr int W = VEC ? DT<T>::EPC : 1;
    __shared__ float red[(MODE == 1) ? RED_THREADS * 2 * W : 1];
    __shared__ int lflag;
    const int n = blockIdx.y;
    const int g = threadIdx.x % p.groups, rl = threadIdx.x / p.groups;
    const long long r0 = (long long)blockIdx.x * p.rows_per_block;
    long long r1 = r0 + p.rows_per_block;
    if (r1 > p.S) r1 = p.S;
    const T* xn = (const T*)p.x + (long long)n * p.S * p.ldx;
    const T* yn = (MODE != 0 && p.y != nullptr) ? (const T*)p.y + (long long)n * p.S * p.ldy : nullptr;
    T* yo = (MODE == 0) ? (T*)p.y + (long long)n * p.S * p.ldy : nullptr;
    const T* dyn = (MODE != 0) ? (const T*)p.dy + (long long)n * p.S * p.lddy : nullptr;
    const T* resn = (MODE == 0 && p.res) ? (const T*)p.res + (long long)n * p.S * p.ldr : nullptr;
    T* dxn = (MODE == 2) ? (T*)p.dx + (long long)n * p.S * p.lddx : nullptr;
    T* drn = (MODE == 2 && p.dres) ? (T*)p.dres + (long long)n * p.S * p.lddres : nullptr;
    const float invS = 1.0f / (float)p.S;

    for (int gbase = 0; gbase * W < p.C; gbase += p.groups) {   // uniform trip count: barriers inside
        const int gg = gbase + g;
        const bool act = gg * W < p.C;
        float mean[W], rstd[W], sc[W], sh[W], k0[W], k1[W], a0[W], a1[W];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const int c = act ? gg * W + e : 0;
            mean_rstd(p.stats, n, p.C, c, p.S, p.eps, mean[e], rstd[e]);
            const float ga = p.gamma ? p.gamma[c] : 1.f;
            sc[e] = rstd[e] * ga;
            sh[e] = (p.beta ? p.beta[c] : 0.f) - mean[e] * sc[e];
            a0[e] = a1[e] = 0.f;
            if constexpr (MODE == 2) {
                k0[e] = p.red[((long long)n * p.C + c) * 2 + 0] * invS;
                k1[e] = p.red[((long long)n * p.C + c) * 2 + 1] * invS;
            }
        }
        if (act && rl < p.rows_par) {
#pragma unroll 4
            for (long long r = r0 + rl; r < r1; r += p.rows_par) {
                float xv[W], o[W], yv[W], dv[W];
                if constexpr (VEC) {
                    Chunk<T> c; c.load(xn + r * p.ldx + gg * W);
#pragma unroll
                    for (int e = 0; e < W; ++e) xv[e] = c.v[e];
                } else {
                    xv[0] = DT<T>::ld(xn + r * p.ldx + gg);
                }
                if constexpr (MODE == 0) {
                    float rv[W];
#pragma unroll
                    for (int e = 0; e < W; ++e) rv[e] = 0.f;
                    if (resn) {
                        if constexpr (VEC) {
                            Chunk<T> c; c.load(resn + r * p.ldr + gg * W);
#pragma unroll
                            for (int e = 0; e < W; ++e) rv[e] = c.v[e];
                        } else {
                            rv[0] = DT<T>::ld(resn + r * p.ldr + gg);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < W; ++e) {
                        const float z = xv[e] * sc[e] + sh[e] + rv[e];
                        o[e] = z > 0.f ? z : z * p.slope;
                    }
                    if constexpr (VEC) {
                        Chunk<T> c;
#pragma unroll
                        for (int e = 0; e < W; ++e) c.v[e] = o[e];
                        c.store(yo + r * p.ldy + gg * W);
                    } else {
                        DT<T>::st(yo + r * p.ldy + gg, o[0]);
                    }
                } else {
                    if constexpr (VEC) {
                        Chunk<T> d; d.load(dyn + r * p.lddy + gg * W);
#pragma unroll
                        for (int e = 0; e < W; ++e) dv[e] = d.v[e];
                        if (yn != nullptr) {
                            Chunk<T> c; c.load(yn + r * p.ldy + gg * W);
#pragma unroll
                            for (int e = 0; e < W; ++e) yv[e] = c.v[e];
                        } else {   // sign of the pre-activation, recomputed exactly as the forward formed it
#pragma unroll
                            for (int e = 0; e < W; ++e) yv[e] = xv[e] * sc[e] + sh[e];
                        }
                    } else {
                        dv[0] = DT<T>::ld(dyn + r * p.lddy + gg);
                        yv[0] = (yn != nullptr) ? DT<T>::ld(yn + r * p.ldy + gg) : xv[0] * sc[0] + sh[0];
                    }
#pragma unroll
                    for (int e = 0; e < W; ++e) {
                        const float dz = yv[e] > 0.f ? dv[e] : dv[e] * p.slope;
                        const float xh = (xv[e] - mean[e]) * rstd[e];
                        if constexpr (MODE == 1) {
                            a0[e] += dz;
                            a1[e] += dz * xh;
                        } else {
                            o[e] = sc[e] * (dz - k0[e] - xh * k1[e]);
                            dv[e] = dz;
                        }
                    }
                    if constexpr (MODE == 2) {
                        if constexpr (VEC) {
                            Chunk<T> c;
#pragma unroll
                            for (int e = 0; e < W; ++e) c.v[e] = o[e];
                            c.store(dxn + r * p.lddx + gg * W);
                            if (drn) {
#pragma unroll
                                for (int e = 0; e < W; ++e) c.v[e] = dv[e];
                                c.store(drn + r * p.lddres + gg * W);
                            }
                        } else {
                            DT<T>::st(dxn + r * p.lddx + gg, o[0]);
                            if (drn) DT<T>::st(drn + r * p.lddres + gg, dv[0]);
                        }
                    }
                }
            }
        }
        if constexpr (MODE == 1) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < W; ++e) {
                red[(threadIdx.x * W + e) * 2 + 0] = a0[e];
                red[(threadIdx.x * W + e) * 2 + 1] = a1[e];
            }
            __syncthreads();
            block_rows_reduce<W>(red, p.groups, p.rows_par, g, rl, act);
            if (act && rl == 0) {
                const int kmax = p.rows_par < RED_STAGE2 ? p.rows_par : RED_STAGE2;
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    float a = 0.f, b = 0.f;
                    for (int k = 0; k < kmax; ++k) {
                        a += red[((k * p.groups + g) * W + e) * 2 + 0];
                        b += red[((k * p.groups + g) * W + e) * 2 + 1];
                    }
                    const int c = gg * W + e;
                    float* wsb = p.ws + ((long long)n * gridDim.x + blockIdx.x) * p.C * 2;
                    wsb[c * 2 + 0] = a;
                    wsb[c * 2 + 1] = b;
                }
            }
        }
    }
}

// InstanceNorm + LeakyReLU forward fused with the MaxPool3d(2) that follows it in an encoder level: one thread owns a
// 2x2x2 cell x one 16-byte channel chunk, writes the eight activated voxels and their maximum.  The maximum is taken
// over the values as stored (rounded to T), so the pooled tensor equals maxpool2_kernel applied to the stored activation
// bit for bit (same NaN propagation, and the backward's arg-max recomputation sees the same numbers).
struct NormPoolParams {
    const void* x; long long ldx;
    const float* stats; const float* gamma; const float* beta;
    void* y; long long ldy;
    void* pooled; long long ldp;
    int N, D, H, W, C; float eps, slope;
};

template <typename T>
__global__ __launch_bounds__(256) void instnorm_pool_fwd_kernel(const NormPoolParams p) {
    constexpr int WD = DT<T>::EPC;
    const int OD = p.D / 2, OH = p.H / 2, OW = p.W / 2;
    const unsigned ug = (unsigned)(p.C / WD);
    const unsigned total = (unsigned)((long long)p.N * OD * OH * OW * ug);   // < 2^31: checked by the host wrapper
    const long long S = (long long)p.D * p.H * p.W;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        unsigned t = i / ug;
        const int g = (int)(i - t * ug);
        unsigned t2 = t / (unsigned)OW;
        const int ow = (int)(t - t2 * (unsigned)OW);
        t = t2 / (unsigned)OH;
        const int oh = (int)(t2 - t * (unsigned)OH);
        const int n = (int)(t / (unsigned)OD);
        const int od = (int)(t - (unsigned)n * (unsigned)OD);
        float sc[WD], sh[WD], best[WD];
#pragma unroll
        for (int e = 0; e < WD; ++e) {
            const int c = g * WD + e;
            float mean, rstd;
            mean_rstd(p.stats, n, p.C, c, S, p.eps, mean, rstd);
            sc[e] = rstd * (p.gamma ? p.gamma[c] : 1.f);
            sh[e] = (p.beta ? p.beta[c] : 0.f) - mean * sc[e];
            best[e] = -INFINITY;
        }
        Chunk<T> in[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long vox = (((long long)n * p.D + 2 * od + (k >> 2)) * p.H + 2 * oh + ((k >> 1) & 1)) * p.W + 2 * ow + (k & 1);
            in[k].load((const T*)p.x + vox * p.ldx + g * WD);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long vox = (((long long)n * p.D + 2 * od + (k >> 2)) * p.H + 2 * oh + ((k >> 1) & 1)) * p.W + 2 * ow + (k & 1);
            Chunk<T> o;
#pragma unroll
            for (int e = 0; e < WD; ++e) {
                const float z = in[k].v[e] * sc[e] + sh[e];
                const float a = (float)(T)(z > 0.f ? z : z * p.slope);
                o.v[e] = a;
                if (a > best[e] || a != a) best[e] = a;
            }
            o.store((T*)p.y + vox * p.ldy + g * WD);
        }
        Chunk<T> b;
#pragma unroll
        for (int e = 0; e < WD; ++e) b.v[e] = best[e];
        const long long ovox = (((long long)n * OD + od) * OH + oh) * OW + ow;
        b.store((T*)p.pooled + ovox * p.ldp + g * WD);
    }
}

// Backward counterpart for an encoder level: the gradient that reaches the level's activation a = lrelu(IN(x)) is
// skip + maxpool_bwd(a, g) (the decoder's skip gradient plus the pooled path from the level below).  One thread owns a
// 2x2x2 cell x one channel chunk: it recomputes the eight activations from x exactly as the forward stored them (the
// arg-max of the cell, first maximum wins, NaN propagates: maxpool2_kernel's rule), writes
// da = T(skip + [voxel is the arg-max] * g) to a dense tensor and accumulates the InstanceNorm-backward sums
// (sum dz, sum dz * xhat), dz = da * lrelu'(pre-activation) -- what maxpool2_bwd followed by instnorm_kernel<MODE 1> do.
struct NormPoolBwdParams {
    const void* x; long long ldx;
    const float* stats; const float* gamma; const float* beta;
    const void* skip; long long lds;
    const void* g; long long ldg;
    void* da; long long ldda;
    int N, D, H, W, C; float eps, slope;
    int groups, rows_par; long long cells_per_block;
    float* ws;
};

template <typename T>
__global__ __launch_bounds__(RED_THREADS) void instnorm_poolbwd_reduce_kernel(const NormPoolBwdParams p) {
    constexpr int WD = DT<T>::EPC;
    __shared__ float red[RED_THREADS * 2 * WD];
    const int n = blockIdx.y;
    const int g = threadIdx.x % p.groups, rl = threadIdx.x / p.groups;
    const int OD = p.D / 2, OH = p.H / 2, OW = p.W / 2;
    const long long cells = (long long)OD * OH * OW, S = (long long)p.D * p.H * p.W;
    const long long c0 = (long long)blockIdx.x * p.cells_per_block;
    long long c1 = c0 + p.cells_per_block;
    if (c1 > cells) c1 = cells;
    const T* xn = (const T*)p.x + (long long)n * S * p.ldx;
    const T* sn = (const T*)p.skip + (long long)n * S * p.lds;
    const T* gn = (const T*)p.g + (long long)n * cells * p.ldg;
    T* dn = (T*)p.da + (long long)n * S * p.ldda;
    for (int gbase = 0; gbase * WD < p.C; gbase += p.groups) {   // uniform trip count: barriers inside
        const int gg = gbase + g;
        const bool act = gg * WD < p.C;
        float mean[WD], rstd[WD], sc[WD], sh[WD], a0[WD], a1[WD];
#pragma unroll
        for (int e = 0; e < WD; ++e) {
            const int c = act ? gg * WD + e : 0;
            mean_rstd(p.stats, n, p.C, c, S, p.eps, mean[e], rstd[e]);
            sc[e] = rstd[e] * (p.gamma ? p.gamma[c] : 1.f);
            sh[e] = (p.beta ? p.beta[c] : 0.f) - mean[e] * sc[e];
            a0[e] = a1[e] = 0.f;
        }
        if (act && rl < p.rows_par) {
            for (long long cell = c0 + rl; cell < c1; cell += p.rows_par) {
                const int ow = (int)(cell % OW), oh = (int)((cell / OW) % OH), od = (int)(cell / ((long long)OW * OH));
                Chunk<T> xin[8], gv;
                Chunk<T> skin[8];
                long long vox[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    vox[k] = ((long long)(2 * od + (k >> 2)) * p.H + 2 * oh + ((k >> 1) & 1)) * p.W + 2 * ow + (k & 1);
                    xin[k].load(xn + vox[k] * p.ldx + gg * WD);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) skin[k].load(sn + vox[k] * p.lds + gg * WD);   // all 17 loads in flight together
                gv.load(gn + cell * p.ldg + gg * WD);
                float best[WD];
                int arg[WD];
#pragma unroll
                for (int e = 0; e < WD; ++e) { best[e] = -INFINITY; arg[e] = 0; }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
#pragma unroll
                    for (int e = 0; e < WD; ++e) {
                        const float z = xin[k].v[e] * sc[e] + sh[e];
                        const float a = (float)(T)(z > 0.f ? z : z * p.slope);
                        if (a > best[e] || a != a) { best[e] = a; arg[e] = k; }
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    Chunk<T> o;
#pragma unroll
                    for (int e = 0; e < WD; ++e) {
                        const float dv = (float)(T)(skin[k].v[e] + (arg[e] == k ? gv.v[e] : 0.f));
                        o.v[e] = dv;
                        const float z = xin[k].v[e] * sc[e] + sh[e];
                        const float dz = z > 0.f ? dv : dv * p.slope;
                        a0[e] += dz;
                        a1[e] += dz * ((xin[k].v[e] - mean[e]) * rstd[e]);
                    }
                    o.store(dn + vox[k] * p.ldda + gg * WD);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < WD; ++e) {
            red[(threadIdx.x * WD + e) * 2 + 0] = a0[e];
            red[(threadIdx.x * WD + e) * 2 + 1] = a1[e];
        }
        __syncthreads();
        block_rows_reduce<WD>(red, p.groups, p.rows_par, g, rl, act);
        if (act && rl == 0) {
            const int kmax = p.rows_par < RED_STAGE2 ? p.rows_par : RED_STAGE2;
#pragma unroll
            for (int e = 0; e < WD; ++e) {
                float a = 0.f, b = 0.f;
                for (int k = 0; k < kmax; ++k) {
                    a += red[((k * p.groups + g) * WD + e) * 2 + 0];
                    b += red[((k * p.groups + g) * WD + e) * 2 + 1];
                }
                float* wsb = p.ws + ((long long)n * gridDim.x + blockIdx.x) * p.C * 2;
                wsb[(gg * WD + e) * 2 + 0] = a;
                wsb[(gg * WD + e) * 2 + 1] = b;
            }
        }
    }
}

// Input gradient of the segmentation head (1x1x1 conv, <= 4 classes) fused with the InstanceNorm-backward sums of the
// layer whose activation a = lrelu(IN(x)) feeds the head: da[v][c] = T(sum_k dy[v][k] * w[k][c]) and
// (sum dz, sum dz * xhat), dz = da * lrelu'(pre-activation recomputed from x).  A streaming pass (thread = voxel x
// 16-byte channel chunk) with the reduction scheme of instnorm_kernel<MODE 1>; replaces a 16-wide MFMA column block fed
// with three real rows, and reads neither the packed weights nor the stored activation.
struct HeadBwdParams {
    const void* dy; long long lddy;
    const float* w;                      // [Cout][C] fp32
    const void* x; long long ldx;        // raw conv output of the receiving layer
    const float* stats; const float* gamma; const float* beta;
    void* da; long long ldda;
    long long S; int C, Cout; float eps, slope;
    int groups, rows_par; long long rows_per_block;
    float* ws;
    int want_dw;                         // also accumulate dW[k][c] = sum_v dy[v][k] * a[v][c], a = T(lrelu(IN(x))) recomputed
};

// second step of the head backward: rows ws[(n * nblk + b)][C][nper] (nper = 2, or 6 with the weight gradient) ->
// red[n][c][0..1], dbeta / dgamma (+)= sum_n, dW[k][c] (+)= sum_{n, b} -- one block, fixed order (bit-reproducible)
struct HeadFinArgs {
    const float* ws; int N, nblk, C, nper, Cout; float* red; float* dbeta; float* dgamma; float* dw; int acc_norm, acc_dw;
};
__global__ __launch_bounds__(256) void head_finalize_kernel(const HeadFinArgs a) {
    __shared__ __attribute__((aligned(16))) float fin[256 * 4 + 64 * 6];
    __shared__ float tot_n[64 * 6];
    const int L = a.C * a.nper;      // <= 64 * 6 floats, a multiple of 4
    for (int o = threadIdx.x; o < L; o += 256) tot_n[o] = 0.f;
    for (int n = 0; n < a.N; ++n) {
        block_rows_sum<256>(a.ws + (long long)n * a.nblk * L, a.nblk, L, fin);   // barriers inside
        const float* t = fin + 256 * 4;
        for (int o = threadIdx.x; o < L; o += 256) {
            const int c = o / a.nper, k = o % a.nper;
            if (k < 2) a.red[((long long)n * a.C + c) * 2 + k] = t[o];
            tot_n[o] += t[o];
        }
        __syncthreads();
    }
    for (int o = threadIdx.x; o < L; o += 256) {
        const int c = o / a.nper, k = o % a.nper;
        const float tot = tot_n[o];
        if (k == 0) { if (a.dbeta) a.dbeta[c] = a.acc_norm ? a.dbeta[c] + tot : tot; }
        else if (k == 1) { if (a.dgamma) a.dgamma[c] = a.acc_norm ? a.dgamma[c] + tot : tot; }
        else if (k - 2 < a.Cout && a.dw) a.dw[(k - 2) * a.C + c] = a.acc_dw ? a.dw[(k - 2) * a.C + c] + tot : tot;
    }
}

template <typename T, bool DW>
__global__ __launch_bounds__(RED_THREADS) void head_dgrad_inbwd_kernel(const HeadBwdParams p) {
    constexpr int WD = DT<T>::EPC;
    __shared__ float red[RED_THREADS * 2 * WD];
    const int n = blockIdx.y;
    const int g = threadIdx.x % p.groups, rl = threadIdx.x / p.groups;
    const long long r0 = (long long)blockIdx.x * p.rows_per_block;
    long long r1 = r0 + p.rows_per_block;
    if (r1 > p.S) r1 = p.S;
    const T* xn = (const T*)p.x + (long long)n * p.S * p.ldx;
    const T* dyn = (const T*)p.dy + (long long)n * p.S * p.lddy;
    T* dan = (T*)p.da + (long long)n * p.S * p.ldda;
    for (int gbase = 0; gbase * WD < p.C; gbase += p.groups) {   // uniform trip count: barriers inside
        const int gg = gbase + g;
        const bool act = gg * WD < p.C;
        float mean[WD], rstd[WD], sc[WD], sh[WD], a0[WD], a1[WD], wv[4][WD];
        float gw[DW ? 4 : 1][WD];
#pragma unroll
        for (int e = 0; e < WD; ++e) {
            const int c = act ? gg * WD + e : 0;
            mean_rstd(p.stats, n, p.C, c, p.S, p.eps, mean[e], rstd[e]);
            sc[e] = rstd[e] * (p.gamma ? p.gamma[c] : 1.f);
            sh[e] = (p.beta ? p.beta[c] : 0.f) - mean[e] * sc[e];
            a0[e] = a1[e] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) wv[k][e] = k < p.Cout ? (float)(T)p.w[k * p.C + c] : 0.f;
#pragma unroll
            for (int k = 0; k < (DW ? 4 : 1); ++k) gw[k][e] = 0.f;
        }
        if (act && rl < p.rows_par) {
            constexpr int UNR = DW ? 2 : 4;   // the weight-gradient variant carries 32 more accumulators: stay under 256 VGPRs
#pragma unroll UNR
            for (long long r = r0 + rl; r < r1; r += p.rows_par) {
                Chunk<T> xc, dc, o;
                xc.load(xn + r * p.ldx + gg * WD);
                dc.load(dyn + r * p.lddy);            // the first EPC (>= 4) channels of the class gradient
#pragma unroll
                for (int e = 0; e < WD; ++e) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc += dc.v[k] * wv[k][e];
                    const float da = (float)(T)acc;
                    o.v[e] = da;
                    const float z = xc.v[e] * sc[e] + sh[e];
                    const float dz = z > 0.f ? da : da * p.slope;
                    a0[e] += dz;
                    a1[e] += dz * ((xc.v[e] - mean[e]) * rstd[e]);
                    if constexpr (DW) {
                        const float av = (float)(T)(z > 0.f ? z : z * p.slope);   // the activation as the forward stored it
#pragma unroll
                        for (int k = 0; k < 4; ++k) gw[k][e] += dc.v[k] * av;
                    }
                }
                o.store(dan + r * p.ldda + gg * WD);
            }
        }
        constexpr int NPER = DW ? 6 : 2;
        float* wsb = p.ws + ((long long)n * gridDim.x + blockIdx.x) * p.C * NPER;
#pragma unroll
        for (int round = 0; round < NPER / 2; ++round) {   // the two-value block reduction, once per value pair
            __syncthreads();
#pragma unroll
            for (int e = 0; e < WD; ++e) {
                float v0, v1;
                if (round == 0) { v0 = a0[e]; v1 = a1[e]; }
                else { v0 = gw[DW ? 2 * round - 2 : 0][e]; v1 = gw[DW ? 2 * round - 1 : 0][e]; }
                red[(threadIdx.x * WD + e) * 2 + 0] = v0;
                red[(threadIdx.x * WD + e) * 2 + 1] = v1;
            }
            __syncthreads();
            block_rows_reduce<WD>(red, p.groups, p.rows_par, g, rl, act);
            if (act && rl == 0) {
                const int kmax = p.rows_par < RED_STAGE2 ? p.rows_par : RED_STAGE2;
#pragma unroll
                for (int e = 0; e < WD; ++e) {
                    float a = 0.f, b = 0.f;
                    for (int k = 0; k < kmax; ++k) {
                        a += red[((k * p.groups + g) * WD + e) * 2 + 0];
                        b += red[((k * p.groups + g) * WD + e) * 2 + 1];
                    }
                    wsb[(gg * WD + e) * NPER + 2 * round + 0] = a;
                    wsb[(gg * WD + e) * NPER + 2 * round + 1] = b;
                }
            }
        }
    }
}

template <typename T, int MODE> int launch_norm(NormParams& p, int N, bool vec, hipStream_t st) {
    constexpr int NTHR = MODE == 1 ? RED_THREADS : 256;
    const RowMap m = row_map(p.C, vec ? DT<T>::EPC : 1, NTHR);
    p.groups = m.groups; p.rows_par = m.rows_par;
    long long blocks;
    if (MODE == 1) {
        blocks = reduce_blocks(p.S, m.rows_par, N, p.C, 2);
    } else {
        // pure streaming passes: 8 blocks of 4 waves per CU, each thread with 4+ independent 16-byte loads in flight
        blocks = ceil_div_ll(p.S, (long long)m.rows_par * 4);
        const long long cap = (long long)msseg_num_cus() * 8 / (N > 0 ? N : 1);
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
    }
    p.rows_per_block = ceil_div_ll(p.S, blocks);
    blocks = ceil_div_ll(p.S, p.rows_per_block);
    dim3 grid((unsigned)blocks, N);
    if (vec) hipLaunchKernelGGL((instnorm_kernel<T, true, MODE>), grid, dim3(NTHR), 0, st, p);
    else hipLaunchKernelGGL((instnorm_kernel<T, false, MODE>), grid, dim3(NTHR), 0, st, p);
    MSSEG_CHECK_LAUNCH("instnorm");
    if (MODE == 1) {
        // red[n][c] = (sum dz, sum dz*xhat); dbeta = sum_n red0, dgamma = sum_n red1
        FinalizeArgs a{p.ws, N, (int)blocks, p.C, 2, 2, p.red, p.dbeta, p.dgamma, p.accumulate};
        hipLaunchKernelGGL(channels_finalize_kernel, channels_finalize_grid(a), dim3(256), 0, st, a);
        MSSEG_CHECK_LAUNCH("channels_finalize");
    }
    return MSSEG_OK;
}

// ---------------------------------------------------------------------------------------------------------
// MaxPool3d(2)
// ---------------------------------------------------------------------------------------------------------
template <typename T, bool VEC, bool BWD>
__global__ __launch_bounds__(256) void maxpool2_kernel(const T* __restrict__ x, long long ldx, const T* __restrict__ dy,
                                                       long long lddy, T* __restrict__ out, long long ldo, int N, int D,
                                                       int H, int W, int C, int accumulate) {
    constexpr int WD = VEC ? DT<T>::EPC : 1;
    const int OD = D / 2, OH = H / 2, OW = W / 2;
    const int groups = (C + WD - 1) / WD;
    const unsigned total = (unsigned)((long long)N * OD * OH * OW * groups);   // < 2^31: checked by the host wrappers
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        // 32-bit index arithmetic: five 64-bit divisions per thread cost more than the eight loads
        const unsigned ug = (unsigned)groups;
        unsigned t = i / ug;
        const int g = (int)(i - t * ug);
        unsigned t2 = t / (unsigned)OW;
        const int ow = (int)(t - t2 * (unsigned)OW);
        t = t2 / (unsigned)OH;
        const int oh = (int)(t2 - t * (unsigned)OH);
        const int n = (int)(t / (unsigned)OD);
        const int od = (int)(t - (unsigned)n * (unsigned)OD);
        float best[WD];
        int arg[WD];
#pragma unroll
        for (int e = 0; e < WD; ++e) { best[e] = -INFINITY; arg[e] = 0; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long vox = (((long long)n * D + 2 * od + (k >> 2)) * H + 2 * oh + ((k >> 1) & 1)) * W + 2 * ow + (k & 1);
            float v[WD];
            if constexpr (VEC) {
                Chunk<T> c; c.load(x + vox * ldx + g * WD);
#pragma unroll
                for (int e = 0; e < WD; ++e) v[e] = c.v[e];
            } else {
                v[0] = DT<T>::ld(x + vox * ldx + g);
            }
#pragma unroll
            for (int e = 0; e < WD; ++e)
                if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; arg[e] = k; }
        }
        const long long ovox = (((long long)n * OD + od) * OH + oh) * OW + ow;
        if constexpr (!BWD) {
            if constexpr (VEC) {
                Chunk<T> c;
#pragma unroll
                for (int e = 0; e < WD; ++e) c.v[e] = best[e];
                c.store(out + ovox * ldo + g * WD);
            } else {
                DT<T>::st(out + ovox * ldo + g, best[0]);
            }
        } else {
            float gv[WD];
            if constexpr (VEC) {
                Chunk<T> c; c.load(dy + ovox * lddy + g * WD);
#pragma unroll
                for (int e = 0; e < WD; ++e) gv[e] = c.v[e];
            } else {
                gv[0] = DT<T>::ld(dy + ovox * lddy + g);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const long long vox = (((long long)n * D + 2 * od + (k >> 2)) * H + 2 * oh + ((k >> 1) & 1)) * W + 2 * ow + (k & 1);
                T* dst = out + vox * ldo + g * WD;
                if constexpr (VEC) {
                    Chunk<T> c;
                    if (accumulate) c.load(dst);
                    else {
#pragma unroll
                        for (int e = 0; e < WD; ++e) c.v[e] = 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < WD; ++e) if (arg[e] == k) c.v[e] += gv[e];
                    c.store(dst);
                } else {
                    float base = accumulate ? DT<T>::ld(dst) : 0.f;
                    if (arg[0] == k) base += gv[0];
                    DT<T>::st(dst, base);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// layout / misc
// ---------------------------------------------------------------------------------------------------------
template <typename TS, typename TD_>
__global__ void ncdhw_to_ndhwc_kernel(const TS* __restrict__ src, TD_* __restrict__ dst, long long ldd, int N, int C,
                                      long long S) {
    const long long total = (long long)N * S * C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long s = (i / C) % S;
        const long long n = i / ((long long)C * S);
        DT<TD_>::st(dst + (n * S + s) * ldd + c, DT<TS>::ld(src + (n * C + c) * S + s));
    }
}
template <typename TS, typename TD_>
__global__ void ndhwc_to_ncdhw_kernel(const TS* __restrict__ src, long long lds, TD_* __restrict__ dst, int N, int C,
                                      long long S) {
    const long long total = (long long)N * S * C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long s = i % S;
        const int c = (int)((i / S) % C);
        const long long n = i / ((long long)C * S);
        DT<TD_>::st(dst + (n * C + c) * S + s, DT<TS>::ld(src + (n * S + s) * lds + c));
    }
}

// out[n][2o] = dy[n][o], zero elsewhere: turns the input gradient / weight gradient of a stride-2 convolution
// into stride-1 problems on the input grid.
template <typename T>
__global__ void zero_stuff2_kernel(const T* __restrict__ dy, long long lddy, T* __restrict__ out, long long ldo, int N,
                                   int OD, int OH, int OW, int ID, int IH, int IW, int C) {
    const long long total = (long long)N * ID * IH * IW * C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int w = (int)(t % IW); t /= IW;
        const int h = (int)(t % IH); t /= IH;
        const int d = (int)(t % ID);
        const int n = (int)(t / ID);
        float v = 0.f;
        if (!(d & 1) && !(h & 1) && !(w & 1) && d / 2 < OD && h / 2 < OH && w / 2 < OW)
            v = DT<T>::ld(dy + ((((long long)n * OD + d / 2) * OH + h / 2) * OW + w / 2) * lddy + c);
        DT<T>::st(out + ((((long long)n * ID + d) * IH + h) * IW + w) * ldo + c, v);
    }
}

template <typename T>
__global__ void add_kernel(const T* a, long long lda, const T* b, long long ldb, T* y, long long ldy, long long rows,
                           int C) {
    const long long total = rows * C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long r = i / C;
        DT<T>::st(y + r * ldy + c, DT<T>::ld(a + r * lda + c) + DT<T>::ld(b + r * ldb + c));
    }
}

// the same on 16-byte chunks (rows of C channels inside wider buffers: ld >= C, everything a multiple of a chunk)
template <typename T>
__global__ __launch_bounds__(256) void add_vec_kernel(const T* __restrict__ a, long long lda, const T* __restrict__ b,
                                                      long long ldb, T* __restrict__ y, long long ldy, long long rows, int cpr) {
    constexpr int EPC = DT<T>::EPC;
    const long long total = rows * cpr;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cpr;
        const int c = (int)(i - r * cpr) * EPC;
        Chunk<T> ca, cb, o;
        ca.load(a + r * lda + c);
        cb.load(b + r * ldb + c);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.v[e] = ca.v[e] + cb.v[e];
        o.store(y + r * ldy + c);
    }
}

// y[n][r][:] = a[n][r][:] + s[n] * b[n][r][:]   (a nullable: y = s * b; s nullable: s = 1) -- the residual add of the Swin
// blocks with the per-sample stochastic-depth scale mask[n] / keep folded in (models/layers/drop_path.py:15-45), and its
// backward (branch gradient = s[n] * dy).  16-byte chunks, rows_per_sample rows per sample.
template <typename T>
__global__ __launch_bounds__(256) void axpy_rows_vec_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                            const float* __restrict__ s, T* __restrict__ y,
                                                            long long chunks_per_sample, long long total_chunks) {
    constexpr int EPC = DT<T>::EPC;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total_chunks; i += (long long)gridDim.x * 256) {
        const float sc = s ? s[i / chunks_per_sample] : 1.f;
        Chunk<T> cb, ca, o;
        cb.load(b + i * EPC);
        if (a) ca.load(a + i * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.v[e] = (a ? ca.v[e] : 0.f) + sc * cb.v[e];
        o.store(y + i * EPC);
    }
}

template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ src, T* __restrict__ dst, int M, int M0, int Tt, int K,
                                    int K0, long long s_m1, long long s_m0, long long s_t, long long s_k1,
                                    long long s_k0, int flip, int coutb, int nkb, long long total) {
    constexpr int EPC = DT<T>::EPC;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long t = i;
        const int e = (int)(t % EPC); t /= EPC;
        const int col = (int)(t % coutb); t /= coutb;
        const int q = (int)(t % 4); t /= 4;
        const int tap = (int)(t % Tt); t /= Tt;
        const int kb = (int)(t % nkb);
        const int cb = (int)(t / nkb);
        const int m = cb * coutb + col;
        const int k = kb * 4 * EPC + q * EPC + e;
        float v = 0.f;
        if (m < M && k < K) {
            const int tt = flip ? (Tt - 1 - tap) : tap;
            v = src[(long long)(m / M0) * s_m1 + (long long)(m % M0) * s_m0 + (long long)tt * s_t +
                    (long long)(k / K0) * s_k1 + (long long)(k % K0) * s_k0];
        }
        DT<T>::st(dst + i, v);
    }
}

// All images of a network in one launch.  Flat grid: block b belongs to the job whose run of ceil(chunks / PACK_CPB) blocks
// contains b (the first wave finds it with a prefix sum over the table), so a table of many small and a few large images
// fills the chip evenly (a grid of [blocks of the largest image] x [jobs] left most blocks empty and the largest image on ~100 of
// them: 56 us for the UNet's 23 MB, 0.8 TB/s).  One thread packs PACK_CPT 16-byte chunks, all their source loads in flight together.
constexpr unsigned PACK_CPT = 4, PACK_CPB = 256 * PACK_CPT;

template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const msseg_pack_job* __restrict__ jobs, int njobs) {
    constexpr unsigned EPC = DT<T>::EPC;
    __shared__ int sel[2];
    if (threadIdx.x < 64) {
        const unsigned lane = threadIdx.x;
        if (lane == 0) sel[0] = -1;
        unsigned before = 0;
        for (int base = 0; base < njobs; base += 64) {   // wave-uniform trip count and exit
            const int jj = base + (int)lane;
            const unsigned nb = jj < njobs ? (unsigned)((jobs[jj].total / EPC + PACK_CPB - 1) / PACK_CPB) : 0u;
            unsigned incl = nb;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = __shfl_up(incl, o);
                if ((int)lane >= o) incl += v;
            }
            const unsigned lo = before + incl - nb;
            const bool mine = blockIdx.x >= lo && blockIdx.x < lo + nb;
            if (mine) { sel[0] = jj; sel[1] = (int)(blockIdx.x - lo); }
            if (__ballot(mine)) break;
            before += __shfl(incl, 63);
        }
    }
    __syncthreads();
    if (sel[0] < 0) return;            // a block past the last job's run (the host rounds the grid up)
    const msseg_pack_job j = jobs[sel[0]];
    T* dst = (T*)j.dst;
    // 32-bit index arithmetic (images are far below 2^31 elements)
    const unsigned nchunks = (unsigned)(j.total / EPC), cb_w = (unsigned)j.cout_block, Tt = (unsigned)j.T, nkb = (unsigned)j.nkb;
    // work order: eight neighbouring columns fastest (eight lanes fill one 128-byte line of the image), then the taps (neighbouring
    // source elements of a conv weight, [m][k][tap] in torch layout), then the other columns; the destination chunk index is
    // recomputed from the coordinates
    const unsigned cb8 = cb_w >> 3;
    float v[PACK_CPT][EPC];
    unsigned ch[PACK_CPT];
#pragma unroll
    for (unsigned i = 0; i < PACK_CPT; ++i) {
        const unsigned wi = (unsigned)sel[1] * PACK_CPB + i * 256u + threadIdx.x;
        ch[i] = 0xffffffffu;
#pragma unroll
        for (unsigned e = 0; e < EPC; ++e) v[i][e] = 0.f;
        if (wi >= nchunks) continue;
        unsigned t = wi;
        const unsigned col_lo = t & 7u; t >>= 3;
        const unsigned tap = t % Tt; t /= Tt;
        const unsigned col = (t % cb8) * 8u + col_lo; t /= cb8;
        const unsigned q = t & 3u; t >>= 2;
        const unsigned kb = t % nkb;
        const unsigned cb = t / nkb;
        ch[i] = (((cb * nkb + kb) * Tt + tap) * 4u + q) * cb_w + col;
        const int m = (int)(cb * cb_w + col);
        if (m >= j.M) continue;
        const unsigned tt = j.flip ? (Tt - 1 - tap) : tap;
        const long long mbase = (long long)(m / j.M0) * j.s_m1 + (long long)(m % j.M0) * j.s_m0 + (long long)tt * j.s_t;
        // k = k1 * K0 + k0 of the chunk's first element by one division, the other seven by carry (16 runtime divisions per chunk
        // made this pass instruction-bound)
        const int kfirst = (int)(kb * 4 * EPC + q * EPC);
        int k1 = kfirst / j.K0, k0 = kfirst - k1 * j.K0;
#pragma unroll
        for (unsigned e = 0; e < EPC; ++e) {
            if (kfirst + (int)e < j.K) v[i][e] = j.src[mbase + (long long)k1 * j.s_k1 + (long long)k0 * j.s_k0];
            if (++k0 == j.K0) { k0 = 0; ++k1; }
        }
    }
#pragma unroll
    for (unsigned i = 0; i < PACK_CPT; ++i) {
        if (ch[i] == 0xffffffffu) continue;
        alignas(16) T out[EPC];
#pragma unroll
        for (unsigned e = 0; e < EPC; ++e) DT<T>::st(&out[e], v[i][e]);
        *(u32x4_t*)(dst + (size_t)ch[i] * EPC) = *(const u32x4_t*)out;
    }
}

// 1x1x1 convolution with a handful of output channels (the segmentation head: 32 -> 2..8 classes).  The implicit-GEMM
// kernels spend a 16-wide MFMA column block on three real outputs; this is a streaming pass instead: one thread per
// voxel reads its Cin channels (16-byte chunks), the weights sit in LDS (same address for every lane: broadcast reads),
// fp32 accumulation, weights rounded to T first so the arithmetic equals the MFMA path's up to summation order.
template <typename T, int COUT>
__global__ __launch_bounds__(256) void conv1x1_head_kernel(const T* __restrict__ x, long long ldx, const float* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ y, long long ldy,
                                                           long long NV, int Cin) {
    constexpr int WD = DT<T>::EPC;
    __shared__ float wS[COUT * 64];
    for (int i = threadIdx.x; i < COUT * Cin; i += 256) wS[i] = (float)(T)w[i];
    __syncthreads();
    float b[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) b[c] = bias ? bias[c] : 0.f;
    for (long long v = blockIdx.x * 256LL + threadIdx.x; v < NV; v += (long long)gridDim.x * 256) {
        float acc[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = b[c];
        const T* xr = x + v * ldx;
        // the row in pieces of up to four chunks with all loads of a piece issued before the first multiply (a loop of one load
        // per trip kept 16 bytes in flight per thread: 1.7 TB/s on the 48-channel rows of Swin-UNETR's output block)
        for (int k0 = 0; k0 < Cin; k0 += 4 * WD) {
            Chunk<T> xc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j * WD < Cin) xc[j].load(xr + k0 + j * WD);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j * WD < Cin) {
#pragma unroll
                    for (int c = 0; c < COUT; ++c)
#pragma unroll
                        for (int e = 0; e < WD; ++e) acc[c] += xc[j].v[e] * wS[c * Cin + k0 + j * WD + e];
                }
        }
        T* yr = y + v * ldy;
#pragma unroll
        for (int c = 0; c < COUT; ++c) DT<T>::st(yr + c, acc[c]);
    }
}

// The same head reading the RAW conv output of the last conv+norm unit: a = T(lrelu(IN(x))) is formed in registers, so
// the normalised activation of that unit is never written (the backward recomputes it as well, head_dgrad_inbwd_kernel).
template <typename T, int COUT>
__global__ __launch_bounds__(256) void conv1x1_head_norm_kernel(const T* __restrict__ x, long long ldx,
                                                                const float* __restrict__ stats,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float slope, float eps, const float* __restrict__ w,
                                                                const float* __restrict__ bias, T* __restrict__ y,
                                                                long long ldy, long long S, int Cin) {
    constexpr int WD = DT<T>::EPC;
    __shared__ float wS[COUT * 64];
    __shared__ float scS[64], shS[64];
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < COUT * Cin; i += 256) wS[i] = (float)(T)w[i];
    for (int c = threadIdx.x; c < Cin; c += 256) {
        float mean, rstd;
        mean_rstd(stats, n, Cin, c, S, eps, mean, rstd);
        const float sc = rstd * (gamma ? gamma[c] : 1.f);
        scS[c] = sc;
        shS[c] = (beta ? beta[c] : 0.f) - mean * sc;
    }
    __syncthreads();
    float b[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) b[c] = bias ? bias[c] : 0.f;
    const T* xn = x + (long long)n * S * ldx;
    T* yn = y + (long long)n * S * ldy;
    for (long long v = blockIdx.x * 256LL + threadIdx.x; v < S; v += (long long)gridDim.x * 256) {
        float acc[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = b[c];
        const T* xr = xn + v * ldx;
        // the row in pieces of up to four chunks, all loads of a piece issued before the first use (as conv1x1_head_kernel)
        for (int k0 = 0; k0 < Cin; k0 += 4 * WD) {
            Chunk<T> xc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j * WD < Cin) xc[j].load(xr + k0 + j * WD);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j * WD < Cin) {
#pragma unroll
                    for (int e = 0; e < WD; ++e) {
                        const int k = k0 + j * WD + e;
                        const float z = xc[j].v[e] * scS[k] + shS[k];
                        const float a = (float)(T)(z > 0.f ? z : z * slope);
#pragma unroll
                        for (int c = 0; c < COUT; ++c) acc[c] += a * wS[c * Cin + k];
                    }
                }
        }
        T* yr = yn + v * ldy;
#pragma unroll
        for (int c = 0; c < COUT; ++c) DT<T>::st(yr + c, acc[c]);
    }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, const uint8_t* __restrict__ decay, long long n, float lr, float b1,
                             float b2, float eps, float wd, float bc1, float bc2_sqrt, const float* gscale,
                             const float* hyper) {
    const float gs = gscale ? gscale[0] : 1.f;
    if (hyper) {  // device-resident (lr, step): lets a captured hipGraph replay with a changing schedule
        lr = hyper[0];
        const float st = hyper[1];
        bc1 = 1.f - powf(b1, st);
        bc2_sqrt = sqrtf(1.f - powf(b2, st));
    }
    // four parameters per thread (16-byte accesses) when the buffers allow it; scalar tail / unaligned fallback
    const bool vec = (n % 4 == 0) && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) &&
                     (!decay || (((uintptr_t)decay & 3) == 0));
    const long long nv = vec ? n / 4 : 0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
        f32x4_t pi = ((const f32x4_t*)p)[i], mi = ((const f32x4_t*)m)[i], vi = ((const f32x4_t*)v)[i];
        const f32x4_t gi = ((const f32x4_t*)g)[i];
        const uint32_t dm = decay ? ((const uint32_t*)decay)[i] : 0x01010101u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gi[e] * gs;
            float pe = pi[e];
            if ((dm >> (8 * e)) & 0xffu) pe *= (1.f - lr * wd);
            const float me = b1 * mi[e] + (1.f - b1) * ge;
            const float ve = b2 * vi[e] + (1.f - b2) * ge * ge;
            mi[e] = me; vi[e] = ve;
            const float denom = sqrtf(ve) / bc2_sqrt + eps;
            pi[e] = pe - (lr / bc1) * (me / denom);
        }
        ((f32x4_t*)m)[i] = mi; ((f32x4_t*)v)[i] = vi; ((f32x4_t*)p)[i] = pi;
    }
    for (long long i = nv * 4 + blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * gs;
        float pi = p[i];
        if (!decay || decay[i]) pi *= (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

// Sum of squares (the gradient norm of clip_grad_norm_) in a fixed order: per-block partials, then one block adds them.  The float
// atomicAdd this replaces made the clipping scale -- and with it every parameter of a clipped step -- differ in the last bit from
// run to run.  The partial rows live in a buffer of the caller (nothing here is shared between calls).
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_finalize_kernel(const float* __restrict__ part, int nblk, float* out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

inline int grid_for(long long total, int per_thread = 4) {
    long long b = ceil_div_ll(total, 256LL * per_thread);
    const long long cap = (long long)msseg_num_cus() * 16;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16)                                   \
    do {                                                                         \
        if ((dtype) == MSSEG_F32) { CALL_F32; }                                  \
        else if ((dtype) == MSSEG_BF16) { CALL_BF16; }                           \
        else MSSEG_FAIL(MSSEG_EINVAL, "bad dtype %d", (int)(dtype));             \
    } while (0)

extern "C" {

size_t msseg_packed_weight_bytes(int M, int T, int K, int cout_block, int dtype) {
    const int epc = dtype == MSSEG_F32 ? 4 : 8;
    const size_t ncb = (size_t)ceil_div(M, cout_block), nkb = (size_t)ceil_div(K, 4 * epc);
    return ncb * nkb * (size_t)T * 4 * (size_t)cout_block * 16;
}

int msseg_pack_weights(const float* src, void* dst, int dtype, int M, int M0, int T, int K, int K0, long long s_m1,
                       long long s_m0, long long s_t, long long s_k1, long long s_k0, int flip, int cout_block,
                       msseg_stream_t stream) {
    if (!src || !dst || M < 1 || T < 1 || K < 1 || M0 < 1 || K0 < 1) MSSEG_FAIL(MSSEG_EINVAL, "pack_weights: bad args");
    if (cout_block != 16 && cout_block != 32 && cout_block != 48) MSSEG_FAIL(MSSEG_EINVAL, "pack_weights: bad cout block");
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    const long long total = (long long)(msseg_packed_weight_bytes(M, T, K, cout_block, dtype) / esz);
    const int nkb = ceil_div(K, 64 / esz);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                  src, (float*)dst, M, M0, T, K, K0, s_m1, s_m0, s_t, s_k1, s_k0, flip, cout_block, nkb, total),
               hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                  src, (bf16_t*)dst, M, M0, T, K, K0, s_m1, s_m0, s_t, s_k1, s_k0, flip, cout_block, nkb, total));
    MSSEG_CHECK_LAUNCH("pack_weights");
    return MSSEG_OK;
}

int msseg_pack_weights_batch(const msseg_pack_job* jobs_dev, int njobs, long long max_total, long long sum_total, int dtype,
                             msseg_stream_t stream) {
    if (!jobs_dev || njobs < 1 || njobs > 65535 || max_total < 1 || sum_total < max_total)
        MSSEG_FAIL(MSSEG_EINVAL, "pack_weights_batch: bad args");
    const long long epc = dtype == MSSEG_F32 ? 4 : 8;
    if (max_total / epc >= 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "pack_weights_batch: image too large");
    // sum over the jobs of ceil(chunks / PACK_CPB) <= sum of chunks / PACK_CPB + njobs: the blocks past the end find no job and exit
    const long long gx = sum_total / epc / PACK_CPB + njobs + 1;
    if (gx >= 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "pack_weights_batch: too many blocks");
    dim3 grid((unsigned)gx);
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weights_batch_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs),
               hipLaunchKernelGGL(pack_weights_batch_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs));
    MSSEG_CHECK_LAUNCH("pack_weights_batch");
    return MSSEG_OK;
}

size_t msseg_reduce_scratch_bytes(void) { return ((size_t)16 << 20) + MSSEG_SCRATCH_COUNTER_BYTES; }

static int scratch_ok(const void* scratch, size_t bytes, const char* who) {
    if (!scratch || ((uintptr_t)scratch & 255) || bytes < msseg_reduce_scratch_bytes())
        MSSEG_FAIL(MSSEG_EWORKSPACE, "%s: needs a zero-initialised, 256-byte aligned scratch of %zu bytes", who,
                   msseg_reduce_scratch_bytes());
    return MSSEG_OK;
}

int msseg_layernorm_param_grad(const void* x, long long ldx, const float* mean, const float* rstd, const void* dy,
                               long long lddy, float* dgamma, float* dbeta, int accumulate, long long rows, int C,
                               void* scratch, size_t scratch_bytes, int dtype, msseg_stream_t stream) {
    if (!x || !mean || !rstd || !dy || !dgamma || !dbeta || rows < 1 || C < 1)
        MSSEG_FAIL(MSSEG_EINVAL, "layernorm_param_grad: bad args");
    if (int rc = scratch_ok(scratch, scratch_bytes, "layernorm_param_grad")) return rc;
    DISPATCH_T(dtype,
               return launch_ln_param_grad<float>(x, ldx, dy, lddy, mean, rstd, dgamma, dbeta, accumulate, rows, C, scratch, (hipStream_t)stream),
               return launch_ln_param_grad<bf16_t>(x, ldx, dy, lddy, mean, rstd, dgamma, dbeta, accumulate, rows, C, scratch, (hipStream_t)stream));
}

int msseg_channel_stats(const void* x, long long ldx, float* stats, int N, long long S, int C, void* scratch,
                        size_t scratch_bytes, int dtype, msseg_stream_t stream) {
    if (!x || !stats || N < 1 || S < 1 || C < 1 || ldx < C) MSSEG_FAIL(MSSEG_EINVAL, "channel_stats: bad args");
    if (int rc = scratch_ok(scratch, scratch_bytes, "channel_stats")) return rc;
    DISPATCH_T(dtype, return launch_stats<float>(x, ldx, stats, N, S, C, 2, 0, scratch, (hipStream_t)stream),
               return launch_stats<bf16_t>(x, ldx, stats, N, S, C, 2, 0, scratch, (hipStream_t)stream));
}

int msseg_channel_sum(const void* x, long long ldx, float* out, long long rows, int C, int accumulate, void* scratch,
                      size_t scratch_bytes, int dtype, msseg_stream_t stream) {
    if (!x || !out || rows < 1 || C < 1 || ldx < C) MSSEG_FAIL(MSSEG_EINVAL, "channel_sum: bad args");
    if (int rc = scratch_ok(scratch, scratch_bytes, "channel_sum")) return rc;
    DISPATCH_T(dtype, return launch_stats<float>(x, ldx, out, 1, rows, C, 1, accumulate, scratch, (hipStream_t)stream),
               return launch_stats<bf16_t>(x, ldx, out, 1, rows, C, 1, accumulate, scratch, (hipStream_t)stream));
}

int msseg_instnorm_act_fwd(const void* x, long long ldx, const float* stats, const float* gamma, const float* beta,
                           const void* residual, long long ldr, void* y, long long ldy, int N, long long S, int C,
                           float eps, float slope, int dtype, msseg_stream_t stream) {
    if (!x || !stats || !y || N < 1 || S < 1 || C < 1) MSSEG_FAIL(MSSEG_EINVAL, "instnorm_act_fwd: bad args");
    NormParams p{};
    p.x = x; p.ldx = ldx; p.stats = stats; p.gamma = gamma; p.beta = beta; p.res = residual; p.ldr = ldr;
    p.y = y; p.ldy = ldy; p.S = S; p.C = C; p.eps = eps; p.slope = slope;
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    const bool vec = vec_ok(x, ldx, C, esz) && vec_ok(y, ldy, C, esz) && (!residual || vec_ok(residual, ldr, C, esz));
    DISPATCH_T(dtype, return (launch_norm<float, 0>(p, N, vec, (hipStream_t)stream)),
               return (launch_norm<bf16_t, 0>(p, N, vec, (hipStream_t)stream)));
}

int msseg_conv3d_k1_head_fwd(const void* x, long long ldx, const float* w, const float* bias, void* y, long long ldy,
                             long long NV, int Cin, int Cout, int dtype, msseg_stream_t stream) {
    if (!x || !w || !y || NV < 1) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_head: bad args");
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_head: bad dtype");
    if (Cout < 1 || Cout > 4 || Cin < 1 || Cin > 64 || !vec_ok(x, ldx, Cin, esz) || ldy < Cout)
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_head: needs 1 <= Cout <= 4, Cin <= 64 in 16-byte chunks (got %d -> %d)", Cin, Cout);
    const int g = grid_for(NV, 1);
#define HEAD(T_, C_) hipLaunchKernelGGL((conv1x1_head_kernel<T_, C_>), dim3(g), dim3(256), 0, (hipStream_t)stream, \
                                        (const T_*)x, ldx, w, bias, (T_*)y, ldy, NV, Cin)
#define HEAD_C(T_) do { if (Cout == 1) HEAD(T_, 1); else if (Cout == 2) HEAD(T_, 2); else if (Cout == 3) HEAD(T_, 3); else HEAD(T_, 4); } while (0)
    DISPATCH_T(dtype, HEAD_C(float), HEAD_C(bf16_t));
#undef HEAD_C
#undef HEAD
    MSSEG_CHECK_LAUNCH("conv3d_k1_head");
    return MSSEG_OK;
}

static int head_bwd_impl(const void* dy, long long lddy, const float* w, void* da, long long ldda, int N, long long S,
                         int C, int Cout, const void* yraw, long long ldyraw, const float* fwd_stats, const float* gamma,
                         const float* beta, float slope, float eps, float* red, float* dgamma, float* dbeta, int accumulate,
                         float* dw, int dw_accumulate, void* scratch, size_t scratch_bytes, int dtype,
                         msseg_stream_t stream) {
    if (!dy || !w || !da || !yraw || !fwd_stats || !red || N < 1 || S < 1)
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_head_dgrad_inbwd: bad args");
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_head_dgrad_inbwd: bad dtype");
    if (int rc = scratch_ok(scratch, scratch_bytes, "conv3d_k1_head_dgrad_inbwd")) return rc;
    const int esz = dtype == MSSEG_F32 ? 4 : 2, epc = 16 / esz;
    if (Cout < 1 || Cout > 4 || C < 1 || C > 64 || !vec_ok(yraw, ldyraw, C, esz) || !vec_ok(da, ldda, C, esz) ||
        (lddy % epc) || ((uintptr_t)dy & 15))
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_head_dgrad_inbwd: needs <= 4 classes in 16-byte aligned gradient rows and "
                                 "16-byte channel chunks (C=%d, Cout=%d, lddy=%lld)", C, Cout, lddy);
    HeadBwdParams p{dy, lddy, w, yraw, ldyraw, fwd_stats, gamma, beta, da, ldda, S, C, Cout, eps, slope, 0, 0, 0, nullptr, 0};
    p.ws = (float*)((unsigned char*)scratch + MSSEG_SCRATCH_COUNTER_BYTES);
    p.want_dw = dw != nullptr;
    const int nper = dw ? 6 : 2;
    const RowMap m = row_map(C, epc, RED_THREADS);
    p.groups = m.groups; p.rows_par = m.rows_par;
    long long blocks = reduce_blocks(S, m.rows_par, N, C, nper);
    p.rows_per_block = ceil_div_ll(S, blocks);
    blocks = ceil_div_ll(S, p.rows_per_block);
    dim3 grid((unsigned)blocks, N);
    if (dw) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((head_dgrad_inbwd_kernel<float, true>), grid, dim3(RED_THREADS), 0, (hipStream_t)stream, p),
                   hipLaunchKernelGGL((head_dgrad_inbwd_kernel<bf16_t, true>), grid, dim3(RED_THREADS), 0, (hipStream_t)stream, p));
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((head_dgrad_inbwd_kernel<float, false>), grid, dim3(RED_THREADS), 0, (hipStream_t)stream, p),
                   hipLaunchKernelGGL((head_dgrad_inbwd_kernel<bf16_t, false>), grid, dim3(RED_THREADS), 0, (hipStream_t)stream, p));
    }
    MSSEG_CHECK_LAUNCH("conv3d_k1_head_dgrad_inbwd");
    HeadFinArgs a{p.ws, N, (int)blocks, C, nper, Cout, red, dbeta, dgamma, dw, accumulate, dw_accumulate};
    hipLaunchKernelGGL(head_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    MSSEG_CHECK_LAUNCH("head_finalize");
    return MSSEG_OK;
}

int msseg_conv3d_k1_head_dgrad_inbwd(const void* dy, long long lddy, const float* w, void* da, long long ldda, int N,
                                     long long S, int C, int Cout, const void* yraw, long long ldyraw,
                                     const float* fwd_stats, const float* gamma, const float* beta, float slope, float eps,
                                     float* red, float* dgamma, float* dbeta, int accumulate, void* scratch,
                                     size_t scratch_bytes, int dtype, msseg_stream_t stream) {
    return head_bwd_impl(dy, lddy, w, da, ldda, N, S, C, Cout, yraw, ldyraw, fwd_stats, gamma, beta, slope, eps, red, dgamma,
                         dbeta, accumulate, nullptr, 0, scratch, scratch_bytes, dtype, stream);
}

int msseg_conv3d_k1_head_bwd_fused(const void* dy, long long lddy, const float* w, void* da, long long ldda, int N,
                                   long long S, int C, int Cout, const void* yraw, long long ldyraw, const float* fwd_stats,
                                   const float* gamma, const float* beta, float slope, float eps, float* red, float* dgamma,
                                   float* dbeta, int accumulate, float* dw, int dw_accumulate, void* scratch,
                                   size_t scratch_bytes, int dtype, msseg_stream_t stream) {
    if (!dw) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_head_bwd_fused: dw is required");
    return head_bwd_impl(dy, lddy, w, da, ldda, N, S, C, Cout, yraw, ldyraw, fwd_stats, gamma, beta, slope, eps, red, dgamma,
                         dbeta, accumulate, dw, dw_accumulate, scratch, scratch_bytes, dtype, stream);
}

int msseg_conv3d_k1_head_norm_fwd(const void* x, long long ldx, const float* stats, const float* gamma, const float* beta,
                                  float slope, float eps, const float* w, const float* bias, void* y, long long ldy, int N,
                                  long long S, int Cin, int Cout, int dtype, msseg_stream_t stream) {
    if (!x || !stats || !w || !y || N < 1 || S < 1) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_head_norm: bad args");
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_head_norm: bad dtype");
    if ((gamma == nullptr) != (beta == nullptr)) MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_head_norm: gamma/beta go together");
    if (Cout < 1 || Cout > 4 || Cin < 1 || Cin > 64 || !vec_ok(x, ldx, Cin, esz) || ldy < Cout)
        MSSEG_FAIL(MSSEG_EINVAL, "conv3d_k1_head_norm: needs 1 <= Cout <= 4, Cin <= 64 in 16-byte chunks (got %d -> %d)", Cin, Cout);
    long long gx = ceil_div_ll(S, 256);
    const long long cap = (long long)msseg_num_cus() * 16 / N + 1;
    if (gx > cap) gx = cap;
    dim3 grid((unsigned)gx, N);
#define HEADN(T_, C_) hipLaunchKernelGGL((conv1x1_head_norm_kernel<T_, C_>), grid, dim3(256), 0, (hipStream_t)stream, \
                                         (const T_*)x, ldx, stats, gamma, beta, slope, eps, w, bias, (T_*)y, ldy, S, Cin)
#define HEADN_C(T_) do { if (Cout == 1) HEADN(T_, 1); else if (Cout == 2) HEADN(T_, 2); else if (Cout == 3) HEADN(T_, 3); else HEADN(T_, 4); } while (0)
    DISPATCH_T(dtype, HEADN_C(float), HEADN_C(bf16_t));
#undef HEADN_C
#undef HEADN
    MSSEG_CHECK_LAUNCH("conv3d_k1_head_norm");
    return MSSEG_OK;
}

int msseg_instnorm_act_pool_fwd(const void* x, long long ldx, const float* stats, const float* gamma, const float* beta,
                                void* y, long long ldy, void* pooled, long long ldp, int N, int D, int H, int W, int C,
                                float eps, float slope, int dtype, msseg_stream_t stream) {
    if (!x || !stats || !y || !pooled || N < 1 || D < 2 || H < 2 || W < 2 || C < 1)
        MSSEG_FAIL(MSSEG_EINVAL, "instnorm_act_pool_fwd: bad args");
    if ((D | H | W) & 1) MSSEG_FAIL(MSSEG_EINVAL, "instnorm_act_pool_fwd: odd spatial size %dx%dx%d", D, H, W);
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "instnorm_act_pool_fwd: bad dtype");
    if (!(vec_ok(x, ldx, C, esz) && vec_ok(y, ldy, C, esz) && vec_ok(pooled, ldp, C, esz)))
        MSSEG_FAIL(MSSEG_EINVAL, "instnorm_act_pool_fwd: needs 16-byte aligned rows and a channel count multiple of %d", 16 / esz);
    const long long total = (long long)N * (D / 2) * (H / 2) * (W / 2) * (C / (16 / esz));
    if (total >= 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "instnorm_act_pool_fwd: too many elements");
    NormPoolParams p{x, ldx, stats, gamma, beta, y, ldy, pooled, ldp, N, D, H, W, C, eps, slope};
    const int g = grid_for(total, 1);
    DISPATCH_T(dtype, hipLaunchKernelGGL(instnorm_pool_fwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, p),
               hipLaunchKernelGGL(instnorm_pool_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, p));
    MSSEG_CHECK_LAUNCH("instnorm_act_pool_fwd");
    return MSSEG_OK;
}

int msseg_instnorm_act_poolbwd_reduce(const void* x, long long ldx, const float* stats, const float* gamma,
                                      const float* beta, const void* skip, long long lds, const void* g, long long ldg,
                                      void* da, long long ldda, float* red, float* dgamma, float* dbeta, int accumulate,
                                      int N, int D, int H, int W, int C, float eps, float slope, void* scratch,
                                      size_t scratch_bytes, int dtype, msseg_stream_t stream) {
    if (!x || !stats || !skip || !g || !da || !red || N < 1 || D < 2 || H < 2 || W < 2 || C < 1)
        MSSEG_FAIL(MSSEG_EINVAL, "instnorm_act_poolbwd_reduce: bad args");
    if ((D | H | W) & 1) MSSEG_FAIL(MSSEG_EINVAL, "instnorm_act_poolbwd_reduce: odd spatial size %dx%dx%d", D, H, W);
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "instnorm_act_poolbwd_reduce: bad dtype");
    if (int rc = scratch_ok(scratch, scratch_bytes, "instnorm_act_poolbwd_reduce")) return rc;
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    if (!(vec_ok(x, ldx, C, esz) && vec_ok(skip, lds, C, esz) && vec_ok(g, ldg, C, esz) && vec_ok(da, ldda, C, esz)))
        MSSEG_FAIL(MSSEG_EINVAL, "instnorm_act_poolbwd_reduce: needs 16-byte aligned rows and a channel count multiple of %d",
                   16 / esz);
    NormPoolBwdParams p{x, ldx, stats, gamma, beta, skip, lds, g, ldg, da, ldda, N, D, H, W, C, eps, slope, 0, 0, 0, nullptr};
    p.ws = (float*)((unsigned char*)scratch + MSSEG_SCRATCH_COUNTER_BYTES);
    const RowMap m = row_map(C, 16 / esz, RED_THREADS);
    p.groups = m.groups; p.rows_par = m.rows_par;
    const long long cells = (long long)(D / 2) * (H / 2) * (W / 2);
    // a cell is eight voxel rows of work: one cell per thread while the grid stays within one block per CU
    long long blocks = reduce_blocks(cells * 8, m.rows_par, N, C, 2);
    p.cells_per_block = ceil_div_ll(cells, blocks);
    blocks = ceil_div_ll(cells, p.cells_per_block);
    dim3 grid((unsigned)blocks, N);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(instnorm_poolbwd_reduce_kernel<float>, grid, dim3(RED_THREADS), 0, (hipStream_t)stream, p),
               hipLaunchKernelGGL(instnorm_poolbwd_reduce_kernel<bf16_t>, grid, dim3(RED_THREADS), 0, (hipStream_t)stream, p));
    MSSEG_CHECK_LAUNCH("instnorm_act_poolbwd_reduce");
    FinalizeArgs a{p.ws, N, (int)blocks, C, 2, 2, red, dbeta, dgamma, accumulate};
    hipLaunchKernelGGL(channels_finalize_kernel, channels_finalize_grid(a), dim3(256), 0, (hipStream_t)stream, a);
    MSSEG_CHECK_LAUNCH("channels_finalize");
    return MSSEG_OK;
}

int msseg_instnorm_act_bwd_reduce(const void* x, long long ldx, const float* stats, const float* gamma,
                                  const float* beta, const void* y, long long ldy,
                                  const void* dy, long long lddy, float* red, float* dgamma, float* dbeta,
                                  int accumulate, int N, long long S, int C, float eps, float slope, void* scratch,
                                  size_t scratch_bytes, int dtype, msseg_stream_t stream) {
    if (!x || !stats || !dy || !red) MSSEG_FAIL(MSSEG_EINVAL, "instnorm_act_bwd_reduce: null pointer");
    if (int rc = scratch_ok(scratch, scratch_bytes, "instnorm_act_bwd_reduce")) return rc;
    NormParams p{};
    p.counter = (unsigned int*)scratch;
    p.ws = (float*)((unsigned char*)scratch + MSSEG_SCRATCH_COUNTER_BYTES);
    p.dgamma = dgamma; p.dbeta = dbeta; p.accumulate = accumulate;
    p.x = x; p.ldx = ldx; p.stats = stats; p.gamma = gamma; p.beta = beta; p.y = (void*)y; p.ldy = ldy; p.dy = dy; p.lddy = lddy; p.red = red;
    p.S = S; p.C = C; p.eps = eps; p.slope = slope;
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    const bool vec = vec_ok(x, ldx, C, esz) && vec_ok(y, ldy, C, esz) && vec_ok(dy, lddy, C, esz);
    DISPATCH_T(dtype, return (launch_norm<float, 1>(p, N, vec, (hipStream_t)stream)),
               return (launch_norm<bf16_t, 1>(p, N, vec, (hipStream_t)stream)));
}

int msseg_instnorm_act_bwd_apply(const void* x, long long ldx, const float* stats, const float* gamma,
                                 const float* beta, const void* y, long long ldy, const void* dy, long long lddy, const float* red, void* dx,
                                 long long lddx, void* dres, long long lddres, int N, long long S, int C, float eps,
                                 float slope, int dtype, msseg_stream_t stream) {
    if (!x || !stats || !dy || !red || !dx) MSSEG_FAIL(MSSEG_EINVAL, "instnorm_act_bwd_apply: null pointer");
    NormParams p{};
    p.x = x; p.ldx = ldx; p.stats = stats; p.gamma = gamma; p.beta = beta; p.y = (void*)y; p.ldy = ldy; p.dy = dy; p.lddy = lddy;
    p.red = (float*)red; p.dx = dx; p.lddx = lddx; p.dres = dres; p.lddres = lddres;
    p.S = S; p.C = C; p.eps = eps; p.slope = slope;
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    const bool vec = vec_ok(x, ldx, C, esz) && vec_ok(y, ldy, C, esz) && vec_ok(dy, lddy, C, esz) &&
                     vec_ok(dx, lddx, C, esz) && (!dres || vec_ok(dres, lddres, C, esz));
    DISPATCH_T(dtype, return (launch_norm<float, 2>(p, N, vec, (hipStream_t)stream)),
               return (launch_norm<bf16_t, 2>(p, N, vec, (hipStream_t)stream)));
}

int msseg_maxpool2_fwd(const void* x, long long ldx, void* y, long long ldy, int N, int D, int H, int W, int C,
                       int dtype, msseg_stream_t stream) {
    if (!x || !y || D < 2 || H < 2 || W < 2 || C < 1) MSSEG_FAIL(MSSEG_EINVAL, "maxpool2_fwd: bad args");
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    const bool vec = vec_ok(x, ldx, C, esz) && vec_ok(y, ldy, C, esz);
    const long long total = (long long)N * (D / 2) * (H / 2) * (W / 2) * ceil_div(C, vec ? 16 / esz : 1);
    if (total >= 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "maxpool2_fwd: too many elements");
    const int g = grid_for(total, 1);
#define MP_F(T_, V_) hipLaunchKernelGGL((maxpool2_kernel<T_, V_, false>), dim3(g), dim3(256), 0, (hipStream_t)stream, \
                                        (const T_*)x, ldx, (const T_*)nullptr, 0LL, (T_*)y, ldy, N, D, H, W, C, 0)
    DISPATCH_T(dtype, if (vec) MP_F(float, true); else MP_F(float, false),
               if (vec) MP_F(bf16_t, true); else MP_F(bf16_t, false));
#undef MP_F
    MSSEG_CHECK_LAUNCH("maxpool2_fwd");
    return MSSEG_OK;
}

int msseg_maxpool2_bwd(const void* x, long long ldx, const void* dy, long long lddy, void* dx, long long lddx, int N,
                       int D, int H, int W, int C, int accumulate, int dtype, msseg_stream_t stream) {
    if (!x || !dy || !dx || D < 2 || H < 2 || W < 2 || C < 1) MSSEG_FAIL(MSSEG_EINVAL, "maxpool2_bwd: bad args");
    const int esz = dtype == MSSEG_F32 ? 4 : 2;
    const bool vec = vec_ok(x, ldx, C, esz) && vec_ok(dy, lddy, C, esz) && vec_ok(dx, lddx, C, esz);
    const long long total = (long long)N * (D / 2) * (H / 2) * (W / 2) * ceil_div(C, vec ? 16 / esz : 1);
    if (total >= 0x7fffffffLL) MSSEG_FAIL(MSSEG_EINVAL, "maxpool2_bwd: too many elements");
    const int g = grid_for(total, 1);
#define MP_B(T_, V_) hipLaunchKernelGGL((maxpool2_kernel<T_, V_, true>), dim3(g), dim3(256), 0, (hipStream_t)stream, \
                                        (const T_*)x, ldx, (const T_*)dy, lddy, (T_*)dx, lddx, N, D, H, W, C, accumulate)
    DISPATCH_T(dtype, if (vec) MP_B(float, true); else MP_B(float, false),
               if (vec) MP_B(bf16_t, true); else MP_B(bf16_t, false));
#undef MP_B
    MSSEG_CHECK_LAUNCH("maxpool2_bwd");
    return MSSEG_OK;
}

int msseg_ncdhw_to_ndhwc(const void* src, int src_dtype, void* dst, long long ldd, int dst_dtype, int N, int C,
                         long long S, msseg_stream_t stream) {
    if (!src || !dst || N < 1 || C < 1 || S < 1 || ldd < C) MSSEG_FAIL(MSSEG_EINVAL, "ncdhw_to_ndhwc: bad args");
    const int g = grid_for((long long)N * C * S);
    hipStream_t st = (hipStream_t)stream;
    if (src_dtype == MSSEG_F32 && dst_dtype == MSSEG_F32)
        hipLaunchKernelGGL((ncdhw_to_ndhwc_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)src, (float*)dst, ldd, N, C, S);
    else if (src_dtype == MSSEG_F32 && dst_dtype == MSSEG_BF16)
        hipLaunchKernelGGL((ncdhw_to_ndhwc_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, ldd, N, C, S);
    else if (src_dtype == MSSEG_BF16 && dst_dtype == MSSEG_BF16)
        hipLaunchKernelGGL((ncdhw_to_ndhwc_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, ldd, N, C, S);
    else if (src_dtype == MSSEG_BF16 && dst_dtype == MSSEG_F32)
        hipLaunchKernelGGL((ncdhw_to_ndhwc_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, ldd, N, C, S);
    else MSSEG_FAIL(MSSEG_EINVAL, "ncdhw_to_ndhwc: bad dtypes");
    MSSEG_CHECK_LAUNCH("ncdhw_to_ndhwc");
    return MSSEG_OK;
}

int msseg_ndhwc_to_ncdhw(const void* src, long long lds, int src_dtype, void* dst, int dst_dtype, int N, int C,
                         long long S, msseg_stream_t stream) {
    if (!src || !dst || N < 1 || C < 1 || S < 1 || lds < C) MSSEG_FAIL(MSSEG_EINVAL, "ndhwc_to_ncdhw: bad args");
    const int g = grid_for((long long)N * C * S);
    hipStream_t st = (hipStream_t)stream;
    if (src_dtype == MSSEG_F32 && dst_dtype == MSSEG_F32)
        hipLaunchKernelGGL((ndhwc_to_ncdhw_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)src, lds, (float*)dst, N, C, S);
    else if (src_dtype == MSSEG_BF16 && dst_dtype == MSSEG_F32)
        hipLaunchKernelGGL((ndhwc_to_ncdhw_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, lds, (float*)dst, N, C, S);
    else if (src_dtype == MSSEG_BF16 && dst_dtype == MSSEG_BF16)
        hipLaunchKernelGGL((ndhwc_to_ncdhw_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, lds, (bf16_t*)dst, N, C, S);
    else if (src_dtype == MSSEG_F32 && dst_dtype == MSSEG_BF16)
        hipLaunchKernelGGL((ndhwc_to_ncdhw_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)src, lds, (bf16_t*)dst, N, C, S);
    else MSSEG_FAIL(MSSEG_EINVAL, "ndhwc_to_ncdhw: bad dtypes");
    MSSEG_CHECK_LAUNCH("ndhwc_to_ncdhw");
    return MSSEG_OK;
}

int msseg_zero_stuff2(const void* dy, long long lddy, void* out, long long ldo, int N, int OD, int OH, int OW, int ID,
                      int IH, int IW, int C, int dtype, msseg_stream_t stream) {
    if (!dy || !out || N < 1 || C < 1) MSSEG_FAIL(MSSEG_EINVAL, "zero_stuff2: bad args");
    const int g = grid_for((long long)N * ID * IH * IW * C);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(zero_stuff2_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                                  lddy, (float*)out, ldo, N, OD, OH, OW, ID, IH, IW, C),
               hipLaunchKernelGGL(zero_stuff2_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream,
                                  (const bf16_t*)dy, lddy, (bf16_t*)out, ldo, N, OD, OH, OW, ID, IH, IW, C));
    MSSEG_CHECK_LAUNCH("zero_stuff2");
    return MSSEG_OK;
}

int msseg_add(const void* a, long long lda, const void* b, long long ldb, void* y, long long ldy, long long rows, int C,
              int dtype, msseg_stream_t stream) {
    if (!a || !b || !y || rows < 1 || C < 1) MSSEG_FAIL(MSSEG_EINVAL, "add: bad args");
    {
        const int esz = dtype == MSSEG_F32 ? 4 : 2, epc = 16 / esz;
        if (C % epc == 0 && lda % epc == 0 && ldb % epc == 0 && ldy % epc == 0 &&
            ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)y)) & 15) == 0) {
            const int cpr = C / epc;
            const int gv = grid_for(rows * cpr, 2);
            DISPATCH_T(dtype,
                       hipLaunchKernelGGL(add_vec_kernel<float>, dim3(gv), dim3(256), 0, (hipStream_t)stream, (const float*)a,
                                          lda, (const float*)b, ldb, (float*)y, ldy, rows, cpr),
                       hipLaunchKernelGGL(add_vec_kernel<bf16_t>, dim3(gv), dim3(256), 0, (hipStream_t)stream,
                                          (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)y, ldy, rows, cpr));
            MSSEG_CHECK_LAUNCH("add");
            return MSSEG_OK;
        }
    }
    const int g = grid_for(rows * C);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(add_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)a, lda,
                                  (const float*)b, ldb, (float*)y, ldy, rows, C),
               hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda,
                                  (const bf16_t*)b, ldb, (bf16_t*)y, ldy, rows, C));
    MSSEG_CHECK_LAUNCH("add");
    return MSSEG_OK;
}

int msseg_axpy_rows(const void* a, const void* b, const float* scale, void* y, int N, long long elems_per_sample, int dtype,
                    msseg_stream_t stream) {
    if (!b || !y || N < 1 || elems_per_sample < 1) MSSEG_FAIL(MSSEG_EINVAL, "axpy_rows: bad args");
    const int esz = dtype == MSSEG_F32 ? 4 : 2, epc = 16 / esz;
    if (dtype != MSSEG_F32 && dtype != MSSEG_BF16) MSSEG_FAIL(MSSEG_EINVAL, "axpy_rows: bad dtype");
    if ((elems_per_sample % epc) || ((uintptr_t)b & 15) || ((uintptr_t)y & 15) || (a && ((uintptr_t)a & 15)))
        MSSEG_FAIL(MSSEG_EINVAL, "axpy_rows: dense 16-byte aligned tensors with elems_per_sample %% %d == 0 expected", epc);
    const long long cps = elems_per_sample / epc, total = cps * N;
    const int g = grid_for(total, 2);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(axpy_rows_vec_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)a,
                                  (const float*)b, scale, (float*)y, cps, total),
               hipLaunchKernelGGL(axpy_rows_vec_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                                  (const bf16_t*)b, scale, (bf16_t*)y, cps, total));
    MSSEG_CHECK_LAUNCH("axpy_rows");
    return MSSEG_OK;
}

int msseg_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* decay_mask,
                     long long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                     const float* grad_scale, const float* dev_hyper, msseg_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1 || (step < 1 && !dev_hyper))
        MSSEG_FAIL(MSSEG_EINVAL, "adamw_step: bad args");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, decay_mask, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, dev_hyper);
    MSSEG_CHECK_LAUNCH("adamw_step");
    return MSSEG_OK;
}

int msseg_sumsq(const float* x, long long n, float* out, float* partials, int n_partials, msseg_stream_t stream) {
    if (!x || !out || !partials || n < 1 || n_partials < 1) MSSEG_FAIL(MSSEG_EINVAL, "sumsq: bad args");
    int nblk = grid_for(n, 16);
    if (nblk > n_partials) nblk = n_partials;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, n, partials);
    MSSEG_CHECK_LAUNCH("sumsq");
    hipLaunchKernelGGL(sumsq_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partials, nblk, out);
    MSSEG_CHECK_LAUNCH("sumsq_finalize");
    return MSSEG_OK;
}

}  // extern "C"
