// shared by attention.hip (exact-fp32-math kernels) and attention_mfma.hip (bf16 MFMA forward)
#pragma once
#include "common.h"

namespace msseg_attn {

struct AttnParams {
    const void* qkv;      // [B, S, H, W, 3C]  channel = which*C + head*hd + e
    const float* qkv_bias;  // [3C] or null (value of padded tokens)
    const float* table;   // relative position bias table [(2ws-1)^3][heads]
    void* out;            // [B, S, H, W, C]
    float* lse;           // [B, nW, heads, N]   log-sum-exp per query (saved for backward)
    const void* dout;     // backward: [B,S,H,W,C]
    void* dqkv;           // backward: [B,S,H,W,3C]
    float* dtable;        // backward: [(2ws-1)^3][heads] fp32, accumulated (atomics from per-workgroup LDS sums)
    int dtab_all_heads;   // the workgroup keeps dtable partial sums for all heads in LDS across its windows
    // MFMA backward with a caller workspace: dS of every (window, head) is written in bf16 MFMA-fragment order, summed
    // over the windows in `ds_groups` groups, and the table gradient is gathered from the sums (no atomics)
    void* ds_ws;          // bf16 [nwin_total][heads][NKT*NKT][64 lanes][16]
    float* ds_psum;       // fp32 [ds_groups][heads][NKT*NKT][64][16]
    int ds_groups;
    int B, S, H, W, C, heads, hd, ws, shift;
    int bws;              // window edge the bias table / index were BUILT for (>= ws; MONAI SwinUNETR slices the 7^3 index
                          // [:n, :n] when the window is clamped to a smaller grid: token i then takes the bias of position
                          // decode_bws(i), swin_unetr_official.py:477-480)
    int Sp, Hp, Wp, nWs, nWh, nWw, N, M3, nwin_total;
    float scale;
    int use_mask;
};

MSSEG_DEVFN int region_id(int z, int Lp, int ws, int shift) { return z < Lp - ws ? 0 : (z < Lp - shift ? 1 : 2); }

// token of window (wz,wy,wx) position p: returns linear voxel index in [0, S*H*W) or -1 for a padded token;
// reg = region id triple packed (only meaningful when shift > 0)
MSSEG_DEVFN int window_token(const AttnParams& p, int wz, int wy, int wx, int pos, int& reg, int& code) {
    const int ws = p.ws;
    const int pz = pos / (ws * ws), py = (pos / ws) % ws, px = pos % ws;
    {   // rel_index(i, j) = code_i - code_j + off, positions decoded on the grid the index was built for
        const int b = p.bws, m = 2 * b - 1;
        const int bz = pos / (b * b), by = (pos / b) % b, bx = pos % b;
        code = (bz * m + by) * m + bx;
    }
    const int sz = wz * ws + pz, sy = wy * ws + py, sx = wx * ws + px;  // coordinates in the shifted, padded grid
    reg = region_id(sz, p.Sp, ws, p.shift) * 9 + region_id(sy, p.Hp, ws, p.shift) * 3 + region_id(sx, p.Wp, ws, p.shift);
    int z = sz + p.shift, y = sy + p.shift, x = sx + p.shift;      // shifted[i] = x[(i + shift) mod Lp]
    if (z >= p.Sp) z -= p.Sp;
    if (y >= p.Hp) y -= p.Hp;
    if (x >= p.Wp) x -= p.Wp;
    if (z >= p.S || y >= p.H || x >= p.W) return -1;
    return (z * p.H + y) * p.W + x;
}


}  // namespace msseg_attn

int msseg_window_attention_fwd_mfma(const msseg_attn::AttnParams& p, hipStream_t stream);  // attention_mfma.hip
int msseg_window_attention_bwd_mfma(const msseg_attn::AttnParams& p, hipStream_t stream);
// bytes of workspace the MFMA backward wants for its atomics-free table gradient; carve() points p.ds_* into it
size_t msseg_window_attention_bwd_mfma_ws_bytes(const msseg_attn::AttnParams& p);
void msseg_window_attention_bwd_mfma_carve(msseg_attn::AttnParams& p, void* workspace);
