// Hausdorff-95 of `eval_model` on the device (/root/reference/engine/test.py:31,48-51,64: MONAI
// HausdorffDistanceMetric(include_background=True, percentile=95)).  Byte / integer work, exact:
//
//   hd_edges        surface voxels of BOTH label maps for ALL classes in one pass: a voxel of class c is on c's surface
//                   iff one of its 6 neighbours is not of class c (outside the volume counts as "not c") -- scipy's
//                   binary_erosion(mask) ^ mask with the default cross structure and border value 0.  Edge maps hold the
//                   class on surface voxels and 0xFF elsewhere; per class the bounding box of both surfaces and the two
//                   surface sizes are formed with integer atomics (exact, order-free).
//   hd_directed     exact squared Euclidean distance from every surface voxel of map A (class c) to the nearest surface
//                   voxel of map B (class c) inside the class's bounding box: the separable minimum
//                   min_z' min_y' min_x' (x-x')^2 + (y-y')^2 + (z-z')^2 as three line passes (x: two-sided scan, y and z:
//                   outward search that stops once k^2 reaches the best value) on integers, then a histogram of the
//                   squared distances (integer atomics).  The host reads the order statistics the percentile needs from
//                   the histogram and takes the square roots in double, as scipy's distance_transform_edt returns them.
//
// HBM / cache bound; nothing here is MFMA-shaped.
#include "common.h"

namespace {

constexpr int HD_INF16 = 0xFFFF;
constexpr int HD_INF32 = 0x3FFFFFFF;

struct HdBox { int z0, y0, x0, z1, y1, x1; };   // half-open

__global__ __launch_bounds__(256) void hd_edges_kernel(const unsigned char* __restrict__ pred,
                                                       const unsigned char* __restrict__ gt, int D, int H, int W, int C,
                                                       unsigned char* __restrict__ ep, unsigned char* __restrict__ eg,
                                                       int* __restrict__ stats /* [C][8]: min z,y,x  max z,y,x  #ep  #eg */) {
    const long long V = (long long)D * H * W;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < V; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H), z = (int)(i / ((long long)W * H));
        const bool border = x == 0 || y == 0 || z == 0 || x == W - 1 || y == H - 1 || z == D - 1;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const unsigned char* src = m ? gt : pred;
            const unsigned char c = src[i];
            bool edge = border;
            if (!edge)
                edge = src[i - 1] != c || src[i + 1] != c || src[i - W] != c || src[i + W] != c ||
                       src[i - (long long)W * H] != c || src[i + (long long)W * H] != c;
            (m ? eg : ep)[i] = edge ? c : (unsigned char)0xFF;
            if (edge && c < C) {
                int* s = stats + 8 * c;
                atomicMin(s + 0, z); atomicMin(s + 1, y); atomicMin(s + 2, x);
                atomicMax(s + 3, z); atomicMax(s + 4, y); atomicMax(s + 5, x);
                atomicAdd(s + 6 + m, 1);
            }
        }
    }
}

__global__ void hd_stats_init_kernel(int* stats, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C * 8) {
        const int f = i & 7;
        stats[i] = f < 3 ? 0x7FFFFFFF : (f < 6 ? -1 : 0);
    }
}

// pass x: one thread per (z, y) line of the box; gx[z][y][x] = distance along the line to the nearest surface voxel of
// class `cls` in `src` (0xFFFF: none on this line).  Box-local output layout [bz][by][bx].
__global__ __launch_bounds__(256) void hd_pass_x_kernel(const unsigned char* __restrict__ src, int cls, int H, int W,
                                                        HdBox b, unsigned short* __restrict__ gx) {
    const int by = b.y1 - b.y0, bx = b.x1 - b.x0, bz = b.z1 - b.z0;
    const long long lines = (long long)bz * by;
    for (long long l = blockIdx.x * 256LL + threadIdx.x; l < lines; l += (long long)gridDim.x * 256) {
        const int y = (int)(l % by), z = (int)(l / by);
        const unsigned char* row = src + ((long long)(b.z0 + z) * H + (b.y0 + y)) * W + b.x0;
        unsigned short* out = gx + l * bx;
        int last = -HD_INF16;
        for (int x = 0; x < bx; ++x) {
            if (row[x] == cls) last = x;
            const int d = x - last;
            out[x] = (unsigned short)(d < HD_INF16 ? d : HD_INF16);
        }
        last = 2 * HD_INF16;
        for (int x = bx - 1; x >= 0; --x) {
            if (row[x] == cls) last = x;
            const int d = last - x;
            if (d < (int)out[x]) out[x] = (unsigned short)d;
        }
    }
}

// pass y: h[z][y][x] = min over y' of gx[z][y'][x]^2 + (y - y')^2, searched outward from y until k^2 >= best
__global__ __launch_bounds__(256) void hd_pass_y_kernel(const unsigned short* __restrict__ gx, HdBox b, int* __restrict__ h) {
    const int by = b.y1 - b.y0, bx = b.x1 - b.x0, bz = b.z1 - b.z0;
    const long long V = (long long)bz * by * bx;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < V; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % bx), y = (int)((i / bx) % by);
        const long long base = i - (long long)y * bx;          // (z, 0, x)
        int best = HD_INF32;
        const int kmax = y > by - 1 - y ? y : by - 1 - y;
        for (int k = 0; k <= kmax; ++k) {
            const int k2 = k * k;
            if (k2 >= best) break;
            if (y - k >= 0) {
                const int g = gx[base + (long long)(y - k) * bx];
                if (g != HD_INF16) { const int v = g * g + k2; best = v < best ? v : best; }
            }
            if (k && y + k < by) {
                const int g = gx[base + (long long)(y + k) * bx];
                if (g != HD_INF16) { const int v = g * g + k2; best = v < best ? v : best; }
            }
        }
        h[i] = best;
    }
}

// pass z at the surface voxels of class `cls` in `tgt` only + histogram of the squared distances
__global__ __launch_bounds__(256) void hd_pass_z_hist_kernel(const int* __restrict__ h, const unsigned char* __restrict__ tgt,
                                                             int cls, int H, int W, HdBox b, int* __restrict__ hist,
                                                             int nbins) {
    const int by = b.y1 - b.y0, bx = b.x1 - b.x0, bz = b.z1 - b.z0;
    const long long V = (long long)bz * by * bx, plane = (long long)by * bx;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < V; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % bx), y = (int)((i / bx) % by), z = (int)(i / plane);
        if (tgt[((long long)(b.z0 + z) * H + (b.y0 + y)) * W + b.x0 + x] != cls) continue;
        int best = HD_INF32;
        const int kmax = z > bz - 1 - z ? z : bz - 1 - z;
        for (int k = 0; k <= kmax; ++k) {
            const int k2 = k * k;
            if (k2 >= best) break;
            if (z - k >= 0) {
                const int g = h[i - (long long)k * plane];
                if (g != HD_INF32) { const int v = g + k2; best = v < best ? v : best; }
            }
            if (k && z + k < bz) {
                const int g = h[i + (long long)k * plane];
                if (g != HD_INF32) { const int v = g + k2; best = v < best ? v : best; }
            }
        }
        atomicAdd(hist + (best < nbins - 1 ? best : nbins - 1), 1);   // last bin: no surface voxel of the other map (inf)
    }
}

inline unsigned grid_for(long long n, int per = 256) {
    long long g = ceil_div_ll(n, per);
    const long long cap = (long long)msseg_num_cus() * 16;
    if (g > cap) g = cap;
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" {

int msseg_hd_edges(const uint8_t* pred, const uint8_t* gt, int D, int H, int W, int C, uint8_t* edges_pred,
                   uint8_t* edges_gt, int* stats, msseg_stream_t stream) {
    if (!pred || !gt || !edges_pred || !edges_gt || !stats || D < 1 || H < 1 || W < 1 || C < 1 || C > 255)
        MSSEG_FAIL(MSSEG_EINVAL, "hd_edges: bad args (%dx%dx%d, %d classes)", D, H, W, C);
    if (D > 16000 || H > 16000 || W > 16000) MSSEG_FAIL(MSSEG_EINVAL, "hd_edges: axis longer than 16000 voxels");
    hipLaunchKernelGGL(hd_stats_init_kernel, dim3(ceil_div(C * 8, 256)), dim3(256), 0, (hipStream_t)stream, stats, C);
    hipLaunchKernelGGL(hd_edges_kernel, dim3(grid_for((long long)D * H * W)), dim3(256), 0, (hipStream_t)stream, pred, gt, D,
                       H, W, C, edges_pred, edges_gt, stats);
    MSSEG_CHECK_LAUNCH("hd_edges");
    return MSSEG_OK;
}

size_t msseg_hd_directed_workspace_bytes(int bz, int by, int bx) {
    const size_t v = (size_t)bz * by * bx;
    return ((v * 2 + 255) / 256) * 256 + v * 4;
}

int msseg_hd_directed_hist(const uint8_t* edges_src, const uint8_t* edges_tgt, int cls, int D, int H, int W,
                           const int* box6, void* workspace, size_t workspace_bytes, int* hist, int nbins,
                           msseg_stream_t stream) {
    if (!edges_src || !edges_tgt || !box6 || !workspace || !hist || cls < 0 || cls > 254)
        MSSEG_FAIL(MSSEG_EINVAL, "hd_directed_hist: bad args");
    HdBox b{box6[0], box6[1], box6[2], box6[3], box6[4], box6[5]};
    if (b.z0 < 0 || b.y0 < 0 || b.x0 < 0 || b.z1 > D || b.y1 > H || b.x1 > W || b.z0 >= b.z1 || b.y0 >= b.y1 || b.x0 >= b.x1)
        MSSEG_FAIL(MSSEG_EINVAL, "hd_directed_hist: box [%d,%d)x[%d,%d)x[%d,%d) outside %dx%dx%d", b.z0, b.z1, b.y0, b.y1,
                   b.x0, b.x1, D, H, W);
    const int bz = b.z1 - b.z0, by = b.y1 - b.y0, bx = b.x1 - b.x0;
    const long long need_bins = (long long)(bz - 1) * (bz - 1) + (long long)(by - 1) * (by - 1) + (long long)(bx - 1) * (bx - 1) + 2;
    if (nbins < need_bins) MSSEG_FAIL(MSSEG_EINVAL, "hd_directed_hist: %d bins, the box needs %lld", nbins, need_bins);
    if (bx >= HD_INF16) MSSEG_FAIL(MSSEG_EINVAL, "hd_directed_hist: box too wide");
    if (workspace_bytes < msseg_hd_directed_workspace_bytes(bz, by, bx))
        MSSEG_FAIL(MSSEG_EWORKSPACE, "hd_directed_hist: workspace %zu B < %zu B", workspace_bytes,
                   msseg_hd_directed_workspace_bytes(bz, by, bx));
    const size_t v = (size_t)bz * by * bx;
    unsigned short* gx = (unsigned short*)workspace;
    int* h = (int*)((char*)workspace + ((v * 2 + 255) / 256) * 256);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(hist, 0, (size_t)nbins * sizeof(int), s) != hipSuccess)
        MSSEG_FAIL(MSSEG_ELAUNCH, "hd_directed_hist: memset failed");
    hipLaunchKernelGGL(hd_pass_x_kernel, dim3(grid_for((long long)bz * by, 64)), dim3(256), 0, s, edges_src, cls, H, W, b, gx);
    hipLaunchKernelGGL(hd_pass_y_kernel, dim3(grid_for((long long)v)), dim3(256), 0, s, gx, b, h);
    hipLaunchKernelGGL(hd_pass_z_hist_kernel, dim3(grid_for((long long)v)), dim3(256), 0, s, h, edges_tgt, cls, H, W, b, hist,
                       nbins);
    MSSEG_CHECK_LAUNCH("hd_directed_hist");
    return MSSEG_OK;
}

}  // extern "C"
