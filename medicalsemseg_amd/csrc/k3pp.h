// Ping-pong conv3d k3 kernel (bf16, 32 input channels per stage, 32-wide cout blocks): host-side interface.
#pragma once
#include "common.h"

struct K3ppParams {
    const void* x;
    long long ldx;
    const void* wp;      // msseg_pack_weights image for cout block 32: [coutblk][kblk][27][4][32][16 B]
    const float* bias;   // optional
    void* y;
    long long ldy;
    int N, D, H, W, K, M;
    // optional fused reductions (same meaning as IgemmParams in igemm_fwd.hip)
    float* stats;
    float* stats_ws;
    unsigned int* counter;
    const void* nb_y; long long nb_ldy;
    const void* nb_a; long long nb_lda;
    const float* nb_stats;
    float nb_slope, nb_eps; long long nb_S;
    float* nb_dgamma; float* nb_dbeta; int nb_acc;
    int accumulate;      // 1: y += conv(x) (the stored bf16 values are read back in the epilogue), statistics of the sums
};

// true when the problem can run on the ping-pong kernel (otherwise the generic igemm kernel is used)
bool msseg_k3pp_eligible(const K3ppParams& p);
int msseg_k3pp_launch(const K3ppParams& p, hipStream_t stream);

// ---- 48 input channels per launch, 16-wide cout blocks (conv3d_k3_c48.hip); weight image = hip.pack_conv_k3_c48 ----
bool msseg_k3c48_shape_ok(int N, int D, int H, int W, int M);   // grid / channel conditions (what the packer must know)
bool msseg_k3c48_eligible(const K3ppParams& p);                 // ... and operand alignment
int msseg_k3c48_launch(const K3ppParams& p, hipStream_t stream);

// ---- second step of the fused conv-epilogue reductions (igemm_fwd.hip): one small block per cout block adds the
// per-workgroup partial rows ws[(y * R + x) * L ..] in a fixed order and writes the per-(n, channel) results.  A
// separate launch: the kernel boundary makes the partial rows visible without release/acquire fences (which would
// write back / invalidate whole L2s at the end of every conv launch).
struct K3FinParams {
    const float* ws;
    int R, N, coutb, M;        // partial rows per cout block, samples, cout block width, logical output channels
    float* stats;              // [N][M][2]
    const float* nb_stats;     // non-null: InstanceNorm-backward mode (see IgemmParams)
    float nb_eps; long long nb_S;
    float* nb_dgamma; float* nb_dbeta; int nb_acc;
};
int msseg_k3_stats_finalize(const K3FinParams& f, int ncb, hipStream_t stream);

// ---- weight gradient (conv3d_k3_wgrad_pp.hip) ----
struct K3WgParams {
    const void* pten; long long ldp;   // dy (channels M)
    const void* qten; long long ldq;   // x  (channels K)
    float* slabs;                      // [pair = mblk * kblks + kblk][workgroup][27][32][32] fp32 partial sums
    int N, D, H, W, M, K, kblks;
    int dbg_noload;                    // timing experiments only
};
bool msseg_k3wg_pp_eligible(const K3WgParams& p);
int msseg_k3wg_pp_grid(const K3WgParams& p);      // workgroups per block pair (= slabs per pair)
int msseg_k3wg_pp_launch(const K3WgParams& p, int gx, hipStream_t stream);

// ---- one-input-channel stem convolution (stem_conv.hip) ----
struct StemParams {
    const void* x; long long ldx;     // [N, D, H, W, 1] bf16
    const void* wp;                   // packed image, K = 27, cout block 32
    const float* bias;
    void* y; long long ldy;
    int N, D, H, W, M;
    float* stats; float* stats_ws;    // optional fused statistics (partial rows -> msseg_k3_stats_finalize)
    int taps;                         // 27 (or 0): conv k3 p1; 1: the 1x1x1 conv of a one-channel volume (image with K = 1)
    // y == nullptr (with stats): statistics only, nothing is stored.  nstats != nullptr: y = lrelu(instance_norm(conv) * gamma +
    // beta) with the statistics nstats[N][M][2] of a previous statistics-only launch (inference: the raw output never exists)
    const float* nstats; const float* gamma; const float* beta; float eps, slope;
};
struct StemWgParams {
    const void* x; long long ldx;
    const void* dy; long long lddy;
    float* slabs;                     // [cout block][workgroup][32][32] fp32
    int N, D, H, W, M;
};
bool msseg_stem_eligible(int dtype, int Cin, int Cout, int k, int s, int pd, long long ldx, long long ldy, const void* y);
int msseg_stem_fwd_launch(const StemParams& p, hipStream_t stream);
int msseg_stem_wgrad_grid(const StemWgParams& p);
int msseg_stem_wgrad_launch(const StemWgParams& p, int gx, hipStream_t stream);

// ---- ConvTranspose3d k2 s2, register-resident weights (deconv_k2s2.hip) ----
bool msseg_deconv2_fast_eligible(int dtype, int Cin, int Cout, const void* coarse, long long ldc, const void* fine,
                                 long long ldf, const float* bias);
int msseg_deconv2_fwd_launch(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy, int N,
                             int D, int H, int W, int Cin, int Cout, hipStream_t stream);
int msseg_deconv2_bwd_launch(const void* dy, long long lddy, const void* wp, void* dx, long long lddx, int N, int D, int H,
                             int W, int Cin, int Cout, const void* yraw, long long ldyraw, const void* act, long long ldact,
                             const float* fwd_stats, float slope, float eps, float* red, float* dgamma, float* dbeta,
                             int accumulate, float* dbias, int dbias_accumulate, void* scratch, size_t scratch_bytes,
                             hipStream_t stream);

// ---- ConvTranspose3d k2 s2 for any channel count with an instantiation, output sliced over grid.y (deconv_k2s2_gen.hip) ----
bool msseg_deconv2g_fwd_eligible(int dtype, int Cin, int Cout, const void* coarse, long long ldc, const void* fine,
                                 long long ldf, const float* bias);
int msseg_deconv2g_fwd_launch(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy, int N,
                              int D, int H, int W, int Cin, int Cout, hipStream_t stream);
bool msseg_deconv2g_bwd_eligible(int dtype, int Cin, int Cout, const void* coarse, long long ldc, const void* fine,
                                 long long ldf);
int msseg_deconv2g_bwd_launch(const void* dy, long long lddy, const void* wp, void* dx, long long lddx, float* part, int N,
                              int D, int H, int W, int Cin, int Cout, hipStream_t stream);
// ---- Linear / 1x1x1 conv on many tokens with few channels, register-resident weights (linear_regw.hip) ----
bool msseg_linear_regw_eligible(int dtype, long long NV, int Cin, int Cout, const void* x, long long ldx, const void* y,
                                long long ldy, const float* bias);
int msseg_linear_regw_launch(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                             long long NV, int Cin, int Cout, hipStream_t stream);
