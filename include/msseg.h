/* msseg.h -- C ABI of libmsseg_hip.so: the MI355X (gfx950) kernels behind the 3-D segmentation
 * hot path (UNet / Swin-UNETR forward+backward, Dice+CE loss, sliding-window blend).
 *
 * The reference (zouyunkai/MedicalSemSeg) has no FFI layer: its hot path is torch/ATen + MONAI calls
 * issued from Python.  Each entry point below names the reference call site(s) whose arithmetic it
 * replaces (paths relative to /root/reference).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - All tensors are dense "channels-last" volumes [N, D, H, W, C]; `ld*` is the element distance
 *    between consecutive voxels (>= C), so a kernel can read/write a channel slice of a wider
 *    (concatenated) buffer.  Pointers must be 16-byte aligned and ld*sizeof(elem) a multiple of 16.
 *  - dtype: MSSEG_F32 (exact fp32 MFMA path, used for parity) or MSSEG_BF16 (bf16 storage, fp32
 *    accumulate).  Reductions, statistics, weight gradients and losses are always fp32.
 *  - Every call only ENQUEUES work on `stream` (a hipStream_t); it never allocates, frees, syncs or
 *    keeps caller pointers.  Scratch comes from the caller (`workspace`), sized by the *_workspace_bytes
 *    queries.  Safe to capture into a hipGraph.
 *  - Return value: MSSEG_OK or a negative MSSEG_E* code; msseg_last_error() gives a thread-local message.
 */
#ifndef MSSEG_H_
#define MSSEG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSSEG_ABI_VERSION 1

#define MSSEG_OK 0
#define MSSEG_EINVAL (-1)   /* bad argument / unsupported shape */
#define MSSEG_ELAUNCH (-2)  /* HIP launch error */
#define MSSEG_EWORKSPACE (-3) /* workspace too small */

#define MSSEG_F32 0
#define MSSEG_BF16 1

typedef void* msseg_stream_t; /* hipStream_t */

int msseg_abi_version(void);
const char* msseg_last_error(void);
/* number of compute units of the current device (grid sizing of the persistent kernels) */
int msseg_num_cus(void);

/* Measurement aid (no reference counterpart): while enabled, the launches of the main convolution kernels (k3pp_kernel,
 * k3wg_pp_kernel, igemm_fwd_kernel, igemm_wgrad_kernel) are bracketed by a hipEvent pair on their stream; `_get` returns
 * the kernel's name and the elapsed time of record i (it waits for that launch).  Process-wide, not for use while a
 * stream is capturing.  bench.py derives `roofline.achieved` of the dominant kernel from these. */
int msseg_ktimer_enable(int on);
int msseg_ktimer_reset(void);
int msseg_ktimer_count(void);
int msseg_ktimer_get(int i, char* name, int cap, float* ms);

/* ---------------------------------------------------------------------------------------------
 * Weight packing.  Source: fp32 parameter in torch layout.  Destination: the MFMA operand image
 * [cout_block][k_block][tap][quarter][cout_in_block][16 bytes] consumed by the igemm kernels.
 * Logical matrix W[m][t][k], m = m1*M0+m0, k = k1*K0+k0, read from
 *   src[m1*s_m1 + m0*s_m0 + t'*s_t + k1*s_k1 + k0*s_k0],  t' = flip ? T-1-t : t.
 * ------------------------------------------------------------------------------------------- */
size_t msseg_packed_weight_bytes(int M, int T, int K, int cout_block, int dtype);
int msseg_pack_weights(const float* src, void* dst, int dtype, int M, int M0, int T, int K, int K0,
                       long long s_m1, long long s_m0, long long s_t, long long s_k1, long long s_k0,
                       int flip, int cout_block, msseg_stream_t stream);
/* Batched form: one launch repacks every job of a device-resident table (all jobs share `dtype`).  Used once per
 * optimizer step to refresh all packed images of a network (reference equivalent: none -- torch.nn.Conv3d reads the
 * fp32 parameter directly, /root/reference/run_training.py:92-93 updates it in place). */
typedef struct msseg_pack_job {
    const float* src;
    void* dst;
    long long s_m1, s_m0, s_t, s_k1, s_k0;
    long long total;          /* elements of the destination image = msseg_packed_weight_bytes / element size */
    int M, M0, T, K, K0, flip, cout_block, nkb;   /* nkb = ceil(K / (64 / element size)) */
} msseg_pack_job;
/* max_total / sum_total: largest and summed `total` of the table's jobs (the table itself is device memory: the grid is sized
 * from these) */
int msseg_pack_weights_batch(const msseg_pack_job* jobs_dev, int njobs, long long max_total, long long sum_total, int dtype,
                             msseg_stream_t stream);
/* cout block (16/32/48) the igemm kernels use for a layer with M logical output channels */
int msseg_cout_block(int M);
/* cout block msseg_conv3d_k3_fwd will use for this problem (pack the weights with it) */
int msseg_conv3d_k3_cout_block(int N, int D, int H, int W, int Cout);
/* tile variant it will use: 0 = 4x8x16-voxel tiles / 8 waves (the MFMA-bound large layers), 1 = 4x4x8, 2 = 2x4x8 */
int msseg_conv3d_k3_variant(int N, int D, int H, int W, int Cout);
/* kernel msseg_conv3d_k3_fwd / _dgrad_inbwd will run for this problem (16-byte aligned, dense operands):
 * 0/1/2 = generic implicit-GEMM tile configurations (as msseg_conv3d_k3_variant), 3 = the LDS-DMA ping-pong kernel
 * (bf16, 32 input channels per stage, large grids), 4 = the 48-input-channel ping-pong kernel (bf16, Cin == 48,
 * Cout % 16 == 0, N <= 4, large grids: Swin-UNETR's 48-wide UnetResBlock convolutions, models/segmentors/swin_unetr.py:
 * 73-128).  Variant 4 reads a DIFFERENT weight image -- four msseg_pack_weights images back to back, see
 * medicalsemseg_amd/hip.py pack_conv_k3_c48 and csrc/conv3d_k3_c48.hip -- so the packer must ask this function. */
int msseg_conv3d_k3_kernel(int N, int D, int H, int W, int Cin, int Cout, int dtype);

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM convolutions (forward-shaped).  y = conv(x, W) + bias.
 * ------------------------------------------------------------------------------------------- */
/* Conv3d k=3 s=1 p=1.  Replaces nn.Conv3d 3x3x3 inside MONAI UnetResBlock / BasicUNet TwoConv
 * (models/segmentors/swin_unetr.py:73-128) and, with dgrad-packed weights, its input gradient.
 * Requires Cin % (16/sizeof(elem)) == 0. */
int msseg_conv3d_k3_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                        int N, int D, int H, int W, int Cin, int Cout, float* stats /* nullable: fused InstanceNorm
                        statistics stats[n][cout][2] = (sum, sum of squares) of y as stored; needs N <= 8 */,
                        void* scratch, size_t scratch_bytes, int dtype, msseg_stream_t stream);
/* Input gradient of a conv k3 (dgrad-packed weights) FUSED with the InstanceNorm-backward reductions of the layer
 * whose activation receives that gradient: da = conv(dy, W'); red[n][c] = (sum dz, sum dz*xhat) with
 * dz = da*lrelu'(act), xhat from (yraw, fwd_stats); dbeta/dgamma (nullable) = sum_n red (written or accumulated).
 * Saves the separate msseg_instnorm_act_bwd_reduce pass (3 tensor reads).  N <= 8, Cout % 4 == 0. */
/* y += conv3d k3 (x, w) without bias: the stored bf16 values of y are read back and the sums stored; stats = InstanceNorm
 * statistics of the sums.  Lets a 64-input-channel layer over cat([a, b]) run as two 32-channel launches on the ping-pong
 * kernel without a concat buffer (inference forward of MONAI BasicUNet's UpCat convs).  Only for shapes the
 * ping-pong kernel takes (bf16, Cin == 32, Cout % 32 == 0, large grids: msseg_conv3d_k3_kernel() == 3), and with
 * Cin == 48 for the shapes of the 48-channel kernel (== 4): a 96-input-channel layer over cat([up, skip]) as two launches
 * on the channel halves of the concat buffer (ldx = 96), Swin-UNETR's decoder blocks. */
int msseg_conv3d_k3_fwd_accumulate(const void* x, long long ldx, const void* wp, void* y, long long ldy, int N, int D, int H,
                                   int W, int Cin, int Cout, float* stats, void* scratch, size_t scratch_bytes, int dtype,
                                   msseg_stream_t stream);
int msseg_conv3d_k3_dgrad_inbwd(const void* dy, long long lddy, const void* wp, void* da, long long ldda, int N, int D,
                                int H, int W, int Cin, int Cout, const void* yraw, long long ldyraw, const void* act,
                                long long ldact, const float* fwd_stats, float slope, float eps, float* red,
                                float* dgamma, float* dbeta, int accumulate, void* scratch, size_t scratch_bytes,
                                int dtype, msseg_stream_t stream);
/* Conv3d k=3 s=2 p=1 (PatchMerging.reduction, models/backbones/swin_nnformer.py:297): x [N,ID,IH,IW,Cin] ->
 * y [N,(ID-1)/2+1,...,Cout].  Its input / weight gradients run as stride-1 problems on msseg_zero_stuff2(dy). */
int msseg_conv3d_k3s2_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                          int N, int ID, int IH, int IW, int Cin, int Cout, int dtype, msseg_stream_t stream);
int msseg_zero_stuff2(const void* dy, long long lddy, void* out, long long ldo, int N, int OD, int OH, int OW, int ID,
                      int IH, int IW, int C, int dtype, msseg_stream_t stream);
/* Conv3d k=3 s=1 p=1 with ONE input channel (BasicUNet conv_0.conv_0, the stem): dedicated kernel, bf16, Cout % 32 == 0 or
 * % 48 == 0 (Swin-UNETR's 1 -> 48 first conv); y == NULL with stats != NULL: statistics only, nothing is stored;
 * wp = the msseg_pack_weights image used by msseg_conv3d_gather_fwd (K = 27); optional fused statistics as
 * msseg_conv3d_k3_fwd.  msseg_conv3d_gather_fwd / _wgrad route eligible problems here by themselves; this entry
 * point adds the statistics.  Returns MSSEG_EINVAL for problems the stem kernel does not cover. */
int msseg_conv3d_stem_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                          int N, int D, int H, int W, int Cout, float* stats, void* scratch, size_t scratch_bytes,
                          int dtype, msseg_stream_t stream);
/* ... the 1x1x1 convolution of a one-channel volume (UnetResBlock conv3 of Swin-UNETR's encoder1; wp = the gather image with
 * K = 1) on the same kernel: centre tap only, same optional statistics */
int msseg_conv3d_stem_k1_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                             int N, int D, int H, int W, int Cout, float* stats, void* scratch, size_t scratch_bytes,
                             int dtype, msseg_stream_t stream);
/* Inference form of the stem unit: y = lrelu(instance_norm(conv(x) + bias) * gamma + beta) with the statistics stats[N][Cout][2]
 * of a statistics-only msseg_conv3d_stem_fwd call in front -- the one-channel conv is cheap enough to run twice, and the raw
 * output's write and the normalisation pass over it disappear (sliding-window inference of BasicUNet at 96^3 windows). */
int msseg_conv3d_stem_norm_fwd(const void* x, long long ldx, const void* wp, const float* bias, const float* stats,
                               const float* gamma, const float* beta, float eps, float slope, void* y, long long ldy, int N,
                               int D, int H, int W, int Cout, int dtype, msseg_stream_t stream);
/* Conv3d k=1 (UnetResBlock.conv3, UnetOutBlock models/segmentors/swin_unetr.py:130, BasicUNet final_conv). */
int msseg_conv3d_k1_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                        long long NV, int Cin, int Cout, int dtype, msseg_stream_t stream);
/* The same for a segmentation head with 1..4 output channels (UnetOutBlock / final_conv with 2-4 classes): a streaming
 * kernel, w = the layer's fp32 weight [Cout][Cin] as it is (no packed image), Cin <= 64 in 16-byte chunks.
 * MSSEG_EINVAL for other shapes (use msseg_conv3d_k1_fwd). */
int msseg_conv3d_k1_head_fwd(const void* x, long long ldx, const float* w, const float* bias, void* y, long long ldy,
                             long long NV, int Cin, int Cout, int dtype, msseg_stream_t stream);
/* Its input gradient fused with the InstanceNorm-backward sums of the layer that feeds the head (as
 * msseg_conv3d_k1_dgrad_inbwd, same streaming form): da[v][c] = sum_k dy[v][k] * w[k][c] for the C channels of that
 * layer, red[n][c] = (sum dz, sum dz*xhat) with the LeakyReLU sign recomputed from yraw (gamma, beta nullable),
 * dgamma/dbeta written or accumulated.  dy rows 16-byte aligned with >= 4 readable channels (classes zero-padded). */
int msseg_conv3d_k1_head_dgrad_inbwd(const void* dy, long long lddy, const float* w, void* da, long long ldda, int N,
                                     long long S, int C, int Cout, const void* yraw, long long ldyraw,
                                     const float* fwd_stats, const float* gamma, const float* beta, float slope, float eps,
                                     float* red, float* dgamma, float* dbeta, int accumulate, void* scratch,
                                     size_t scratch_bytes, int dtype, msseg_stream_t stream);
/* The head reading the RAW conv output `x` of the last conv + InstanceNorm + LeakyReLU unit (N samples of S voxels):
 * y = conv1x1(T(lrelu(IN(x) * gamma + beta))) with the normalisation applied in registers -- the unit's activation is
 * never written.  stats: [N][Cin][2] (sum, sum of squares) of x.  Replaces InstanceNorm3d + LeakyReLU + final Conv3d(k=1)
 * of MONAI BasicUNet (TwoConv tail + final_conv). */
int msseg_conv3d_k1_head_norm_fwd(const void* x, long long ldx, const float* stats, const float* gamma, const float* beta,
                                  float slope, float eps, const float* w, const float* bias, void* y, long long ldy, int N,
                                  long long S, int Cin, int Cout, int dtype, msseg_stream_t stream);
/* msseg_conv3d_k1_head_dgrad_inbwd plus the head's weight gradient dw[Cout][C] (+)= sum_v dy[v][k] * a[v][c] with the
 * activation a recomputed from yraw (as the fused forward does): one streaming pass for da, the InstanceNorm-backward sums
 * and dw. */
int msseg_conv3d_k1_head_bwd_fused(const void* dy, long long lddy, const float* w, void* da, long long ldda, int N,
                                   long long S, int C, int Cout, const void* yraw, long long ldyraw, const float* fwd_stats,
                                   const float* gamma, const float* beta, float slope, float eps, float* red, float* dgamma,
                                   float* dbeta, int accumulate, float* dw, int dw_accumulate, void* scratch,
                                   size_t scratch_bytes, int dtype, msseg_stream_t stream);

/* Conv3d with few input channels (Cin*k^3 <= 128), kernel k, stride s, pad p, gathered im2col-style:
 * the 1->C stem convs and PatchEmbed3D.proj (models/blocks/patch_embeddings.py:109). */
int msseg_conv3d_gather_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                            int N, int ID, int IH, int IW, int Cin, int Cout, int k, int s, int p, int dtype,
                            msseg_stream_t stream);
/* ConvTranspose3d k=s=2 (UnetrUpBlock.transp_conv, BasicUNet UpCat): x [N,D,H,W,Cin] -> y [N,2D,2H,2W,Cout]. */
int msseg_deconv_k2s2_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* y, long long ldy,
                          int N, int D, int H, int W, int Cin, int Cout, int dtype, msseg_stream_t stream);
/* input gradient of the above: dy [N,2D,2H,2W,Cout] -> dx [N,D,H,W,Cin]. */
int msseg_deconv_k2s2_bwd_data(const void* dy, long long lddy, const void* wp, void* dx, long long lddx,
                               int N, int D, int H, int W, int Cin, int Cout, int dtype, msseg_stream_t stream);
/* The two flat input-gradient kernels with the InstanceNorm-backward sums of the RECEIVING layer fused into the epilogue
 * (arguments as msseg_conv3d_k3_dgrad_inbwd): the gradient that reaches the second conv+norm unit of a UNet level comes
 * from the 1x1x1 output conv (level 0) or from a transposed conv (other decoder levels / the bottleneck).
 * k1: dy [N*S, Cin] -> da [N*S, Cout], S voxels per sample.  deconv: dy [N,2D,2H,2W,Cout] -> dx [N,D,H,W,Cin]. */
int msseg_conv3d_k1_dgrad_inbwd(const void* dy, long long lddy, const void* wp, void* da, long long ldda, int N,
                                long long S, int Cin, int Cout, const void* yraw, long long ldyraw, const void* act,
                                long long ldact, const float* fwd_stats, float slope, float eps, float* red,
                                float* dgamma, float* dbeta, int accumulate, void* scratch, size_t scratch_bytes,
                                int dtype, msseg_stream_t stream);
int msseg_deconv_k2s2_bwd_data_inbwd(const void* dy, long long lddy, const void* wp, void* dx, long long lddx, int N,
                                     int D, int H, int W, int Cin, int Cout, const void* yraw, long long ldyraw,
                                     const void* act, long long ldact, const float* fwd_stats, float slope, float eps,
                                     float* red, float* dgamma, float* dbeta, int accumulate, void* scratch,
                                     size_t scratch_bytes, int dtype, msseg_stream_t stream);
/* Input gradient of ConvTranspose3d k2 s2 with everything the pass over dy can produce: dx, optionally (yraw != NULL) the
 * InstanceNorm-backward sums red[N][Cin][2] (+ dgamma / dbeta) of the layer that receives dx, and optionally
 * (dbias != NULL) the bias gradient dbias[Cout] (+)= sum over all fine voxels of dy.  bf16 layers with Cin in {32, 64},
 * Cout = 32 run on the register-resident-weight kernel (deconv_k2s2.hip): one read of dy, no statistics / channel-sum
 * passes; other shapes fall back to the implicit-GEMM kernel plus a channel-sum pass. */
int msseg_deconv_k2s2_bwd_fused(const void* dy, long long lddy, const void* wp, void* dx, long long lddx, int N,
                                int D, int H, int W, int Cin, int Cout, const void* yraw, long long ldyraw,
                                const void* act, long long ldact, const float* fwd_stats, float slope, float eps,
                                float* red, float* dgamma, float* dbeta, int accumulate, float* dbias, int dbias_accumulate,
                                void* scratch, size_t scratch_bytes, int dtype, msseg_stream_t stream);

/* Input gradient of ConvTranspose3d k2 s2 as ONE fp32 stage group in the layout of msseg_conv3d_k3_small_partials
 * (part[Cin / 4][N*D*H*W][4], csrc/deconv_k2s2_gen.hip): msseg_conv3d_k3_small_bwd_finish(part, 1, ...) then stores dx, or
 * runs the whole backward of the conv + InstanceNorm + LeakyReLU unit whose activation the transposed conv read (the deep
 * UpCat levels of MONAI BasicUNet; /root/reference/models/segmentors/swin_unetr.py:93-128 transp_conv).  bf16; dy: fine
 * [N, 2D, 2H, 2W, Cout]; wp: the backward image of msseg_pack_weights (M = Cin, K = 8 * Cout); part_bytes >= N*D*H*W*Cin*4.
 * msseg_deconv_k2s2_bwd_partials_ok() == 1 for the channel counts with an instantiation (Cin % 32 == 0, Cout / 16 in
 * {3, 4, 6, 8, 12, 24}). */
int msseg_deconv_k2s2_bwd_partials_ok(int Cin, int Cout, int dtype);
int msseg_deconv_k2s2_bwd_partials(const void* dy, long long lddy, const void* wp, float* part, size_t part_bytes, int N,
                                   int D, int H, int W, int Cin, int Cout, int dtype, msseg_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * Weight gradients (fp32 output in the torch parameter layout; deterministic two-stage reduction).
 * `accumulate` != 0 adds into dw instead of overwriting.
 * ------------------------------------------------------------------------------------------- */
size_t msseg_wgrad_workspace_bytes(int M, int T, int K);
/* which kernel msseg_conv3d_k3_wgrad selects for dense (ld == channels), aligned tensors of this shape:
 * 3 = LDS-DMA ping-pong kernel (conv3d_k3_wgrad_pp.hip), 0 = generic igemm_wgrad kernel (tests assert the selection). */
int msseg_conv3d_k3_wgrad_kernel(int N, int D, int H, int W, int Cin, int Cout, int dtype);
int msseg_conv3d_k3_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw,
                          int N, int D, int H, int W, int Cin, int Cout, int accumulate,
                          void* workspace, size_t workspace_bytes, int dtype, msseg_stream_t stream);
int msseg_conv3d_k1_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw,
                          long long NV, int Cin, int Cout, int accumulate,
                          void* workspace, size_t workspace_bytes, int dtype, msseg_stream_t stream);
/* Weight AND bias gradient of nn.Linear / a 1x1x1 convolution in one pass over the tokens: dw[Cout][Cin] (+)= dy^T x,
 * dbias[Cout] (+)= column sums of dy (dbias nullable).  Replaces autograd's two reductions for the Swin stages' Linear layers
 * (models/backbones/swin_nnformer.py:24-42,128-196).  bf16; channel counts that split into the kernel's output slices
 * (msseg_linear_wgrad_ok() == 1: multiples of 48 on both sides cover every Swin width); deterministic; workspace as
 * msseg_wgrad_workspace_bytes(Cout, 1, Cin).  Callers keep msseg_conv3d_k1_wgrad + msseg_channel_sum otherwise. */
int msseg_linear_wgrad_ok(long long NV, int Cin, int Cout, int dtype);
int msseg_linear_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw, float* dbias, long long NV,
                       int Cin, int Cout, int accumulate_w, int accumulate_b, void* workspace, size_t workspace_bytes,
                       int dtype, msseg_stream_t stream);
int msseg_conv3d_gather_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw,
                              int N, int ID, int IH, int IW, int Cin, int Cout, int k, int s, int p, int accumulate,
                              void* workspace, size_t workspace_bytes, int dtype, msseg_stream_t stream);
int msseg_deconv_k2s2_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw,
                            int N, int D, int H, int W, int Cin, int Cout, int accumulate,
                            void* workspace, size_t workspace_bytes, int dtype, msseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * conv3d k3 s1 p1 on SMALL grids (12^3 / 6^3 levels of the UNet; csrc/conv3d_k3_small.hip), bf16: the 32-channel stages of a
 * layer are split over workgroups (split-K), the fp32 partial blocks part[stage group][Cout/4][N*D*H*W][4] are combined by a finish
 * kernel that also does what follows the conv -- forward: bias, bf16 raw output, InstanceNorm statistics, normalise +
 * LeakyReLU, optional 2x2x2 max-pool (MONAI TwoConv / Down); input gradient: either the plain sum, or the
 * whole backward of the conv + InstanceNorm + LeakyReLU unit whose activation was the conv's input (dy of that unit, its
 * dgamma / dbeta).  wp: msseg_pack_weights image with cout block 32 (forward image, or the flipped / transposed input-
 * gradient image).  D a multiple of 6 (3 for the tiles, even for the pool), H, W multiples of 6, D*H*W <= 2048; channels
 * multiples of 32; N <= 8.
 * ------------------------------------------------------------------------------------------- */
int msseg_conv3d_k3_small_ok(int N, int D, int H, int W, int Cin, int Cout, int dtype);
/* number of partial blocks per output element the partials call writes (= `nstages` of the finish calls) */
int msseg_conv3d_k3_small_stage_groups(int N, int D, int H, int W, int Cin, int Cout);
size_t msseg_conv3d_k3_small_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout);
int msseg_conv3d_k3_small_partials(const void* x, long long ldx, const void* wp, float* part, size_t part_bytes, int N,
                                   int D, int H, int W, int Cin, int Cout, msseg_stream_t stream);
int msseg_conv3d_k3_small_fwd_finish(const float* part, int nstages, const float* bias, const float* gamma,
                                     const float* beta, float eps, float slope, void* yraw, long long ldy, void* act,
                                     long long lda, void* pooled, long long ldp, float* stats, int N, int D, int H, int W,
                                     int Cout, msseg_stream_t stream);
/* ... with an optional residual added before the LeakyReLU: act = lrelu(instance_norm(y) * gamma + beta + residual) -- the
 * second convolution of MONAI's UnetResBlock (models/segmentors/swin_unetr.py:73-128 of the reference) */
int msseg_conv3d_k3_small_fwd_finish_res(const float* part, int nstages, const float* bias, const float* gamma,
                                         const float* beta, float eps, float slope, void* yraw, long long ldy, void* act,
                                         long long lda, const void* residual, long long ldr, void* pooled, long long ldp,
                                         float* stats, int N, int D, int H, int W, int Cout, msseg_stream_t stream);
int msseg_conv3d_k3_small_bwd_finish(const float* part, int nstages, void* dx, long long lddx, const void* unit_yraw,
                                     long long lduy, const float* unit_stats, const float* unit_gamma,
                                     const float* unit_beta, float eps, float slope, float* dgamma, float* dbeta,
                                     int accumulate, int N, int D, int H, int W, int Cin, msseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Depthwise Conv3d k3 s1 p1 (groups = channels) of the SwinDepth MLP (models/backbones/swindepth.py:36-41,56-65).
 * x, y channels-last [N, D, H, W, C] (C % (16 / sizeof(elem)) == 0); w_taps = the weight [C, 1, 3, 3, 3] transposed to
 * tap-major [27][C] in the tensors' dtype; flip = 1 mirrors the taps (the input gradient: dx = dwconv(dy, flip)).
 * wgrad: dw (torch layout [C, 1, 3, 3, 3], fp32) and dbias [C] written or accumulated; partial rows in `scratch`
 * (msseg_reduce_scratch_bytes()), added in row order by a second kernel -- deterministic.
 * ------------------------------------------------------------------------------------------- */
int msseg_dwconv3d_k3_fwd(const void* x, long long ldx, const void* w_taps, const float* bias, void* y, long long ldy, int N,
                          int D, int H, int W, int C, int flip, int dtype, msseg_stream_t stream);
/* nn.AvgPool3d(kernel_size=3, stride=1, padding=1), count_include_pad (models/backbones/swinception.py:113-116): the same
 * kernel with unit taps, the fp32 sum scaled by 1/27 before the one rounding to the tensors' dtype; self-adjoint. */
int msseg_avgpool3d_k3(const void* x, long long ldx, void* y, long long ldy, int N, int D, int H, int W, int C, int dtype,
                       msseg_stream_t stream);
int msseg_dwconv3d_k3_wgrad(const void* x, long long ldx, const void* dy, long long lddy, float* dw, float* dbias,
                            int accumulate_w, int accumulate_b, int N, int D, int H, int W, int C, void* scratch,
                            size_t scratch_bytes, int dtype, msseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SegFormer3D pieces (models/backbones/segformer_backbone.py:51-117, models/segmentors/segformer_head_official.py:65-90).
 * interp_trilinear: F.interpolate(mode='trilinear', align_corners=False) between channels-last volumes
 *   x [N, ID, IH, IW, C] and y [N, OD, OH, OW, C] (C % (16 / sizeof(elem)) == 0); bwd = its adjoint in gather form
 *   (deterministic, one axis per pass, fp32 intermediates in `workspace`): dx from dy; C % 4 == 0 suffices here.
 * kv_attention: o = softmax(q k^T * scale) v per head with few keys (the spatial-reduction attention): q, o [B, N, C],
 *   kv [B, M, 2C] (k = first C channels, v = last C; channel = head * head_dim + c), head_dim in {16, 32, 48, 64};
 *   lse [B, heads, N] fp32 (log-sum-exp of the scaled scores, kept for the backward).  bwd: dq [B, N, C] and
 *   dkv [B, M, 2C] (both fully written); workspace of msseg_kv_attention_bwd_workspace_bytes() holds P, dS and the
 *   per-query-chunk partial sums of dk / dv.
 * scale_channels: y[n, v, c] = x[n, v, c] * scale[n][c] (Dropout3d with a given mask / (1 - p)); dense [N, S, C].
 * ------------------------------------------------------------------------------------------- */
int msseg_interp_trilinear_fwd(const void* x, long long ldx, void* y, long long ldy, int N, int ID, int IH, int IW, int OD,
                               int OH, int OW, int C, int dtype, msseg_stream_t stream);
size_t msseg_interp_trilinear_bwd_workspace_bytes(int N, int ID, int IH, int IW, int OD, int OH, int OW, int C);
int msseg_interp_trilinear_bwd(const void* dy, long long lddy, void* dx, long long lddx, int N, int ID, int IH, int IW, int OD,
                               int OH, int OW, int C, void* workspace, size_t workspace_bytes, int dtype, msseg_stream_t stream);
int msseg_kv_attention_fwd(const void* q, const void* kv, void* o, float* lse, int B, int N, int M, int heads, int head_dim,
                           float scale, int dtype, msseg_stream_t stream);
size_t msseg_kv_attention_bwd_workspace_bytes(int B, int N, int M, int heads, int head_dim);
int msseg_kv_attention_bwd(const void* q, const void* kv, const void* o, const float* lse, const void* dout, void* dq, void* dkv,
                           int B, int N, int M, int heads, int head_dim, float scale, void* workspace, size_t workspace_bytes,
                           int dtype, msseg_stream_t stream);
int msseg_scale_channels(const void* x, const float* scale, void* y, int N, long long S, int C, int dtype, msseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Normalisation / activation / pooling / layout (HBM-bound, one pass each).
 * ------------------------------------------------------------------------------------------- */
/* Reductions are two-stage and deterministic: blocks write partials into `scratch`, the last block to arrive
 * (agent-scope release/acquire on a counter at scratch[0]) sums them in a fixed order.  `scratch` must be
 * 256-byte aligned, at least msseg_reduce_scratch_bytes() long and ZERO-INITIALISED ONCE by the caller (kernels
 * leave the counter at zero); kernels sharing one scratch must be ordered on one stream. */
size_t msseg_reduce_scratch_bytes(void);
/* stats[n][c][0..1] = (sum, sum of squares) over the S voxels of sample n. */
int msseg_channel_stats(const void* x, long long ldx, float* stats, int N, long long S, int C, void* scratch,
                        size_t scratch_bytes, int dtype, msseg_stream_t stream);
/* InstanceNorm3d(eps) [+affine] [+residual] + LeakyReLU(slope) in one pass (slope 1.0 = identity):
 * y = lrelu((x-mean)*rstd*gamma+beta + residual).  nn.InstanceNorm3d + nn.LeakyReLU of UnetResBlock /
 * TwoConv. */
int msseg_instnorm_act_fwd(const void* x, long long ldx, const float* stats, const float* gamma, const float* beta,
                           const void* residual, long long ldr, void* y, long long ldy, int N, long long S, int C,
                           float eps, float slope, int dtype, msseg_stream_t stream);
/* The same fused with the MaxPool3d(2) that follows it in an encoder level (MONAI BasicUNet Down = MaxPool3d(2) -> TwoConv):
 * y as above (no residual) and pooled[n][d/2][h/2][w/2][c] = max over the 2x2x2 cell of y as stored; equals
 * msseg_maxpool2_fwd(y) bit for bit.  D, H, W even; rows 16-byte aligned, C a multiple of 16 / sizeof(element). */
int msseg_instnorm_act_pool_fwd(const void* x, long long ldx, const float* stats, const float* gamma, const float* beta,
                                void* y, long long ldy, void* pooled, long long ldp, int N, int D, int H, int W, int C,
                                float eps, float slope, int dtype, msseg_stream_t stream);
/* Backward of that pair for an encoder level, first pass: da = skip + maxpool2_bwd(y, g) written densely (skip: the
 * decoder's gradient of the level's output, g: gradient of the pooled tensor; y is recomputed from x, arg-max rule as
 * msseg_maxpool2_bwd) and red[n][c] = (sum dz, sum dz*xhat) of da as msseg_instnorm_act_bwd_reduce(y == NULL) gives it,
 * dgamma/dbeta likewise.  Follow with msseg_instnorm_act_bwd_apply(dy = da, red). */
int msseg_instnorm_act_poolbwd_reduce(const void* x, long long ldx, const float* stats, const float* gamma,
                                      const float* beta, const void* skip, long long lds, const void* g, long long ldg,
                                      void* da, long long ldda, float* red, float* dgamma, float* dbeta, int accumulate,
                                      int N, int D, int H, int W, int C, float eps, float slope, void* scratch,
                                      size_t scratch_bytes, int dtype, msseg_stream_t stream);
/* backward, pass 1: red[n][c] = (sum dz, sum dz*xhat), dz = dy * lrelu'(z): the sign of the pre-activation z is taken
 * from the forward output y, or -- y == NULL, layers without residual -- recomputed as x*rstd*gamma + beta - mean*...,
 * which saves one tensor read (gamma/beta = the forward's affine parameters, nullable);
 * optional affine gradients dbeta[c] (+)= sum_n red[n][c][0], dgamma[c] (+)= sum_n red[n][c][1]. */
int msseg_instnorm_act_bwd_reduce(const void* x, long long ldx, const float* stats, const float* gamma,
                                  const float* beta, const void* y, long long ldy,
                                  const void* dy, long long lddy, float* red, float* dgamma, float* dbeta,
                                  int accumulate, int N, long long S, int C, float eps, float slope, void* scratch,
                                  size_t scratch_bytes, int dtype, msseg_stream_t stream);
/* backward, pass 2: dx = rstd*gamma*(dz - red0/S - xhat*red1/S); dres (optional) = dz.  y == NULL as above. */
int msseg_instnorm_act_bwd_apply(const void* x, long long ldx, const float* stats, const float* gamma,
                                 const float* beta, const void* y, long long ldy, const void* dy, long long lddy, const float* red, void* dx,
                                 long long lddx, void* dres, long long lddres, int N, long long S, int C, float eps,
                                 float slope, int dtype, msseg_stream_t stream);
/* MaxPool3d(2) forward / backward (first maximum in scan order receives the gradient, as ATen). */
int msseg_maxpool2_fwd(const void* x, long long ldx, void* y, long long ldy, int N, int D, int H, int W, int C,
                       int dtype, msseg_stream_t stream);
int msseg_maxpool2_bwd(const void* x, long long ldx, const void* dy, long long lddy, void* dx, long long lddx,
                       int N, int D, int H, int W, int C, int accumulate, int dtype, msseg_stream_t stream);
/* NCDHW (src_dtype) <-> NDHWC (dst_dtype) */
int msseg_ncdhw_to_ndhwc(const void* src, int src_dtype, void* dst, long long ldd, int dst_dtype, int N, int C,
                         long long S, msseg_stream_t stream);
int msseg_ndhwc_to_ncdhw(const void* src, long long lds, int src_dtype, void* dst, int dst_dtype, int N, int C,
                         long long S, msseg_stream_t stream);
/* out[c] (+)= sum over rows of x[row][c] (bias gradients). */
int msseg_channel_sum(const void* x, long long ldx, float* out, long long rows, int C, int accumulate, void* scratch,
                      size_t scratch_bytes, int dtype, msseg_stream_t stream);
/* y = a + b (elementwise over rows x C with strides) */
int msseg_add(const void* a, long long lda, const void* b, long long ldb, void* y, long long ldy, long long rows,
              int C, int dtype, msseg_stream_t stream);
/* y[n] = a[n] + scale[n] * b[n] over N samples of elems_per_sample dense elements (a NULL: y = scale * b; scale NULL: 1):
 * residual add with the per-sample stochastic-depth factor mask[n] / keep folded in (models/layers/drop_path.py:15-45,
 * swin_nnformer.py:286-287) and the branch gradient of its backward. */
int msseg_axpy_rows(const void* a, const void* b, const float* scale, void* y, int N, long long elems_per_sample, int dtype,
                    msseg_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * Swin transformer pieces (models/backbones/swin_nnformer.py).  Linear layers are msseg_conv3d_k1_* on tokens.
 * ------------------------------------------------------------------------------------------- */
/* Shifted-window attention between the qkv and proj Linears (swin_nnformer.py:128-196 inside :235-289):
 * qkv [B,S,H,W,3C] (channel = which*C + head*hd + e) -> out [B,S,H,W,C].  Window partition, cyclic shift, zero
 * padding to a window multiple (padded tokens carry qkv_bias), the relative-position bias table
 * [(2ws-1)^3][heads] and the -100 region mask are all applied through addressing; lse [B*nW][heads][ws^3] is
 * saved for the backward.  head_dim in {8,16,32}. */
int msseg_window_attention_fwd(const void* qkv, const float* qkv_bias, const float* table, void* out, float* lse, int B,
                               int S, int H, int W, int C, int heads, int ws, int shift, int dtype,
                               msseg_stream_t stream);
/* dqkv fully written for every token; dtable (nullable) ACCUMULATED (caller zero-fills when needed). */
int msseg_window_attention_bwd(const void* qkv, const float* qkv_bias, const float* table, const void* out,
                               const float* lse, const void* dout, void* dqkv, float* dtable, int B, int S, int H, int W,
                               int C, int heads, int ws, int shift, int dtype, msseg_stream_t stream);
/* Same with a caller workspace (bytes from msseg_window_attention_bwd_workspace_bytes; 0 = this shape/dtype has no use
 * for one).  With it the bias-table gradient (swin_nnformer.py:147-155 in reverse) is computed without atomics: dS tiles
 * in bf16 -> sum over windows -> gather per table entry, deterministic.  workspace == NULL behaves as the call above. */
size_t msseg_window_attention_bwd_workspace_bytes(int B, int S, int H, int W, int C, int heads, int ws, int shift, int dtype);
/* Forms with a separate BIAS window: the table has (2*bias_ws-1)^3 rows and token i of a window takes the relative
 * position of index position decode_{bias_ws}(i) -- MONAI SwinUNETR builds table and index for window 7 and slices the
 * index [:n, :n] when the window is clamped to a smaller grid (swin_unetr_official.py:375-385, 477-480).  bias_ws == ws is
 * the plain case (bf16 then runs on the MFMA kernels); bias_ws > ws runs on the exact-fp32-math kernels. */
int msseg_window_attention_fwd2(const void* qkv, const float* qkv_bias, const float* table, void* out, float* lse, int B,
                                int S, int H, int W, int C, int heads, int ws, int shift, int bias_ws, int dtype,
                                msseg_stream_t stream);
int msseg_window_attention_bwd2(const void* qkv, const float* qkv_bias, const float* table, const void* out,
                                const float* lse, const void* dout, void* dqkv, float* dtable, int B, int S, int H, int W,
                                int C, int heads, int ws, int shift, int bias_ws, int dtype, void* workspace,
                                size_t workspace_bytes, msseg_stream_t stream);
int msseg_window_attention_bwd_ws(const void* qkv, const float* qkv_bias, const float* table, const void* out,
                                  const float* lse, const void* dout, void* dqkv, float* dtable, int B, int S, int H, int W,
                                  int C, int heads, int ws, int shift, int dtype, void* workspace, size_t workspace_bytes,
                                  msseg_stream_t stream);
/* LayerNorm over the channel dim of rows x C (nn.LayerNorm, eps 1e-5); mean/rstd [rows] saved for backward. */
int msseg_layernorm_fwd(const void* x, long long ldx, const float* gamma, const float* beta, void* y, long long ldy,
                        float* mean, float* rstd, long long rows, int C, float eps, int dtype, msseg_stream_t stream);
/* input gradient */
int msseg_layernorm_bwd(const void* x, long long ldx, const float* gamma, const float* mean, const float* rstd,
                        const void* dy, long long lddy, void* dx, long long lddx, long long rows, int C, int dtype,
                        msseg_stream_t stream);
/* ... + add: dx = T(layernorm backward) + add, the sum autograd forms when the normalised tensor also feeds a residual branch
 * (x -> LayerNorm and x -> x + f(LayerNorm(x)) in every Swin block, /root/reference/models/backbones/swin_nnformer.py:243-262):
 * one pass instead of the backward + a separate add.  Vector kernel only (C a multiple of the 16-byte chunk, <= 4096). */
int msseg_layernorm_bwd_add(const void* x, long long ldx, const float* gamma, const float* mean, const float* rstd,
                            const void* dy, long long lddy, const void* add, long long ldadd, void* dx, long long lddx,
                            long long rows, int C, int dtype, msseg_stream_t stream);
/* parameter gradients dgamma[c] (+)= sum_rows dy*xhat, dbeta[c] (+)= sum_rows dy (deterministic two-stage reduction) */
int msseg_layernorm_param_grad(const void* x, long long ldx, const float* mean, const float* rstd, const void* dy,
                               long long lddy, float* dgamma, float* dbeta, int accumulate, long long rows, int C,
                               void* scratch, size_t scratch_bytes, int dtype, msseg_stream_t stream);
/* exact (erf) GELU */
int msseg_gelu_fwd(const void* x, void* y, long long n, int dtype, msseg_stream_t stream);
int msseg_gelu_bwd(const void* x, const void* dy, void* dx, long long n, int dtype, msseg_stream_t stream);
/* Linear + GELU of a Swin block's MLP in one pass each way (models/backbones/swin_nnformer.py:24-42 of the reference: fc1,
 * act, fc2; north_star "GELU fusion").  fwd: pre = x W^T + b and act = gelu(pre as stored) from ONE launch (the separate
 * GELU pass re-read pre).  bwd: dpre = (dy W2) * gelu'(pre) -- fc2's input gradient written as fc1's output gradient (the
 * separate pass wrote and re-read the intermediate).  Bit-identical to msseg_conv3d_k1_fwd + msseg_gelu_fwd / _bwd.
 * wp: msseg_pack_weights image (T = 1) of the layer (bwd: of fc2's input-gradient form).  bf16, the register-resident-weight
 * kernel's shapes C -> 4C at C = 48, 96, 192, 384 (msseg_linear_gelu_ok() == 1); callers keep the two-kernel chain otherwise. */
int msseg_linear_gelu_ok(long long NV, int Cin, int Cout, int dtype);
int msseg_linear_gelu_fwd(const void* x, long long ldx, const void* wp, const float* bias, void* pre, long long ldpre, void* act,
                          long long ldact, long long NV, int Cin, int Cout, int dtype, msseg_stream_t stream);
int msseg_linear_gelu_bwd(const void* dy, long long lddy, const void* wp, const void* pre, long long ldpre, void* dpre,
                          long long lddpre, long long NV, int Cin, int Cout, int dtype, msseg_stream_t stream);
/* nn.Linear with the residual add of a Swin block in its epilogue: y = res + (x W^T + b), the sum formed from the bf16-rounded
 * Linear output exactly as Linear followed by an add pass forms it (/root/reference/models/backbones/swin_nnformer.py:243-262:
 * x = shortcut + proj(attn), x = x + mlp(norm2(x))).  msseg_linear_add_ok() == 1 for the shapes the register-resident-weight
 * kernel takes. */
int msseg_linear_add_ok(long long NV, int Cin, int Cout, int dtype);
int msseg_linear_add_fwd(const void* x, long long ldx, const void* wp, const float* bias, const void* res, long long ldres,
                         void* y, long long ldy, long long NV, int Cin, int Cout, int dtype, msseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Dice + cross-entropy loss (MONAI DiceCELoss(to_onehot_y, softmax, squared_pred) as built at
 * run_training.py:103-105, called engine/train.py:62) and the hard Dice metric (engine/train.py:89-111).
 * logits: channels-last [N, S, ld] with ld >= C the voxel stride, or NCDHW when ld == 0; labels: one value
 * per voxel; C <= 16.
 * ------------------------------------------------------------------------------------------- */
/* partial[n][c][0..3] += (sum p*t, sum p^2, sum t, -sum t*log p); hard[n][c][0..2] += (|P&T|, |P|, |T|)
 * (either pointer may be NULL).  label_dtype: 0 = f32, 1 = bf16, 2 = u8, 3 = i64. */
int msseg_dice_ce_partials(const void* logits, long long ld, int dtype, const void* labels, int label_dtype,
                           float* partial, float* hard, int N, long long S, int C, msseg_stream_t stream);
/* loss[0] = mean_{n,c}(1 - (2I+snr)/(den+sdr)) + CE ; loss[1] = dice term, loss[2] = ce term. */
/* Deterministic one-call form of the forward (N <= 8): per-block partial rows in `scratch` (the reduce scratch,
 * msseg_reduce_scratch_bytes()), added in a fixed order by a one-block second step that also writes partial[N][C][4],
 * hard[N][C][3] (nullable) and loss[3] = (dice + ce, dice, ce).  No zero-initialised outputs, no atomics: two runs on the
 * same inputs give the same bits. */
int msseg_dice_ce_fwd(const void* logits, long long ld, int dtype, const void* labels, int label_dtype, float* partial,
                      float* hard, float* loss, int N, long long S, int C, float smooth_nr, float smooth_dr, void* scratch,
                      size_t scratch_bytes, msseg_stream_t stream);
int msseg_dice_ce_finalize(const float* partial, float* loss, int N, long long S, int C, float smooth_nr,
                           float smooth_dr, msseg_stream_t stream);
/* dlogits = gscale[0] * dLoss/dlogits (same layout/dtype as logits). */
int msseg_dice_ce_bwd(const void* logits, long long ld, int dtype, const void* labels, int label_dtype,
                      const float* partial, const float* gscale, void* dlogits, long long ldd, int N, long long S,
                      int C, float smooth_nr, float smooth_dr, msseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser: fused AdamW over one flat fp32 buffer (torch.optim.AdamW(betas=(0.9,0.95), eps=1e-6) of
 * run_training.py:92-93; decay applies where decay_mask[i] != 0 -- timm add_weight_decay semantics).
 * grad_scale[0] multiplies the gradient first (clip coefficient / loss-scale inverse).
 * ------------------------------------------------------------------------------------------- */
int msseg_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* decay_mask,
                     long long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                     const float* grad_scale, const float* dev_hyper /* nullable: device [lr, step] overriding the
                     host values, so a captured hipGraph can be replayed while the schedule advances */,
                     msseg_stream_t stream);
/* out[0] = sum of squares of x[0..n), added in a fixed order (bit-identical from run to run): per-block sums into
   partials[0..n_partials) -- a buffer of the caller, at most n_partials blocks are launched (4096 is plenty) -- then one
   block adds them and overwrites out[0]. */
int msseg_sumsq(const float* x, long long n, float* out, float* partials, int n_partials, msseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Sliding-window inference (engine/utils.py:120-151): gather windows, weighted blend, normalise.
 * ------------------------------------------------------------------------------------------- */
/* out[c][vol] += imp[roi] * win[c][roi] placed at start (z,y,x); cnt[vol] += imp (one channel).
 * win: channels-last [roi][ld] when ld > 0, NCDHW when ld == 0. */
int msseg_sw_blend(const void* win, long long ld, int dtype, const float* imp, float* out, float* cnt,
                   int C, int VD, int VH, int VW, int RD, int RH, int RW, int z0, int y0, int x0,
                   msseg_stream_t stream);
/* win[c][roi] (NCDHW, dtype) = vol[c][window at (z0,y0,x0)] with `cval` outside the volume. */
int msseg_sw_gather(const float* vol, void* win, int dtype, int C, int VD, int VH, int VW, int RD, int RH, int RW,
                    int z0, int y0, int x0, float cval, msseg_stream_t stream);
int msseg_sw_normalize(float* out, const float* cnt, int C, long long V, msseg_stream_t stream);
/* Batched forms (one launch per window batch; the reference's loop body engine/utils.py:120-148 for `sw_batch_size`
 * windows at once).  table: device int32 [nwin][4] = (sample b, z0, y0, x0); b < 0 = unused slot.
 * gather: win[j] = window j of vol[b] (fp32 [B][C][VD][VH][VW], sample stride vol_bstride elements), `cval` outside;
 *         win layout NCDHW [nwin][C][roi] when ldw == 0, channels-last [nwin][roi][ldw] otherwise.
 * blend : out[b][c][vol] += imp * win[j][c], cnt[b][vol] += imp for every window of the table, each output voxel
 *         accumulated in table order with the reference's roundings (bit-identical to nwin sequential msseg_sw_blend
 *         calls); windows of one batch may overlap.  nwin <= 256, C <= 16. */
int msseg_sw_gather_batch(const float* vol, long long vol_bstride, void* win, long long ldw, int dtype, const int* table,
                          int nwin, int C, int VD, int VH, int VW, int RD, int RH, int RW, float cval,
                          msseg_stream_t stream);
int msseg_sw_blend_batch(const void* win, long long ldw, int dtype, const float* imp, float* out, long long out_bstride,
                         float* cnt, long long cnt_bstride, const int* table, int nwin, int C, int VD, int VH, int VW,
                         int RD, int RH, int RW, msseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Layout passes of the vendored MONAI Swin-UNETR encoder (csrc/layout.hip; /root/reference/models/segmentors/
 * swin_unetr_official.py:244-270,699-712), channels-last volumes, 16-byte channel chunks (C % 8 == 0 bf16, % 4 fp32).
 * msseg_box_copy: dst[n, d, h, w, :C] = src[n, d, h, w, :C] inside the common box, zero elsewhere in dst -- F.pad with zeros at
 *   the high end when dst is larger, the crop x[:, :d, :h, :w] when it is smaller (each is the other's adjoint).
 * msseg_merge_gather_fwd: out[n, i, j, k, s*C + c] = x[n, 2i + a_s, 2j + b_s, 2k + c_s, c] (zero beyond the grid) for eight
 *   sub-grids; `subs` holds 3 bits per slot s at bits 3s..3s+2 (bit 0: offset along D, bit 1: H, bit 2: W); duplicates allowed
 *   (PatchMerging's x5 == x2, x6 == x3).  out: [N, ceil(D/2), ceil(H/2), ceil(W/2), 8C].  _bwd: the adjoint (per fine voxel the
 *   sum, in slot order, of the slots whose offset equals the voxel's parity; deterministic).
 * ------------------------------------------------------------------------------------------- */
int msseg_box_copy(const void* src, long long lds, int SD, int SH, int SW, void* dst, long long ldd, int DD, int DH, int DW,
                   int N, int C, int dtype, msseg_stream_t stream);
int msseg_merge_gather_fwd(const void* x, long long ldx, int D, int H, int W, void* out, long long ldo, int N, int C,
                           unsigned subs, int dtype, msseg_stream_t stream);
int msseg_merge_gather_bwd(const void* dy, long long lddy, void* dx, long long lddx, int N, int D, int H, int W, int C,
                           unsigned subs, int dtype, msseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Post-inference (the step right after the path).
 * argmax_u8: out[v] = first arg max over c of logits[c][v] (NCDHW fp32, one sample) -- engine/test.py:140-141
 *            (softmax is monotonic, so it is skipped).
 * resample_nearest_u8: scipy.ndimage.zoom(order=0, prefilter=False) of a uint8 label map to (TD, TH, TW) --
 *            utils/misc.py:420-425; coordinates in double exactly as scipy forms them.
 * majority_vote_u8: labels [F][V] -> out[V]; votes[c] = #folds with label c for c >= 1, votes[0] = 1, first maximum --
 *            majority_vote.py:23-37.
 * ------------------------------------------------------------------------------------------- */
int msseg_argmax_u8(const float* logits, int C, long long V, uint8_t* out, msseg_stream_t stream);
int msseg_resample_nearest_u8(const uint8_t* src, int SD, int SH, int SW, uint8_t* dst, int TD, int TH, int TW,
                              msseg_stream_t stream);
int msseg_majority_vote_u8(const uint8_t* labels, int F, long long V, int C, uint8_t* out, msseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Hausdorff-95 of eval_model (engine/test.py:31,48-51,64: MONAI HausdorffDistanceMetric(include_background=True,
 * percentile=95, reduction="mean", get_not_nans=True)).  Exact integer work on uint8 label maps [D][H][W]:
 * hd_edges: surface voxels (6-neighbour erosion XOR, outside = background) of both maps for all classes: edge maps hold
 *            the class on surface voxels and 0xFF elsewhere; stats[c][8] = {min z, y, x, max z, y, x (inclusive) of class
 *            c's surface voxels of BOTH maps, #surface voxels of pred, of gt}.
 * hd_directed_hist: hist[d2] = number of class-`cls` surface voxels of `edges_tgt` whose squared Euclidean distance (voxel
 *            units) to the nearest class-`cls` surface voxel of `edges_src` is d2, searched inside box6 = {z0, y0, x0, z1,
 *            y1, x1} (half-open; must contain both surfaces: hd_edges' box); bin nbins-1 counts voxels with no source voxel.
 *            nbins >= (bz-1)^2 + (by-1)^2 + (bx-1)^2 + 2.  The caller takes the percentile's order statistics from the
 *            histogram and the square roots in double (scipy.ndimage.distance_transform_edt's values).
 * ------------------------------------------------------------------------------------------- */
int msseg_hd_edges(const uint8_t* pred, const uint8_t* gt, int D, int H, int W, int C, uint8_t* edges_pred,
                   uint8_t* edges_gt, int* stats, msseg_stream_t stream);
size_t msseg_hd_directed_workspace_bytes(int bz, int by, int bx);
int msseg_hd_directed_hist(const uint8_t* edges_src, const uint8_t* edges_tgt, int cls, int D, int H, int W,
                           const int* box6, void* workspace, size_t workspace_bytes, int* hist, int nbins,
                           msseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Device-side training crop + augmentation (the step right before the path): one gather launch per
 * batch of patches from a volume cached in HBM.  Replaces RandCropByPosNegLabeld + RandFlipd x3 + RandRotate90d +
 * RandShiftIntensityd + RandScaleIntensityd (data/dataset_builder.py:108-193); the random draws arrive in `table`.
 * img: fp32 [C][VD][VH][VW]; lab: uint8 [VD][VH][VW] or NULL; out_img: [npatch][C][R][R][R] (out_dtype), out_lab: fp32
 * [npatch][1][R][R][R].  Row: crop start (z0, y0, x0), flips bit0/1/2 = axis d/h/w, rotk = quarter turns in the (d, h)
 * plane (np.rot90 sense), image value = (v + shift) * scale.
 * ------------------------------------------------------------------------------------------- */
typedef struct msseg_aug_row { int32_t z0, y0, x0, flips, rotk, pad0; float shift, scale; } msseg_aug_row;
int msseg_aug_crop_batch(const float* img, const uint8_t* lab, int C, int VD, int VH, int VW, const void* table,
                         int npatch, void* out_img, int out_dtype, float* out_lab, int R, msseg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MSSEG_H_ */
