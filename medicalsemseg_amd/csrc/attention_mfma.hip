// Swin 3-D window attention forward on the matrix cores (bf16): S^T = K Q^T and O^T = V^T P with
// v_mfma_f32_32x32x16_bf16, softmax in registers.
//
// One workgroup (4 waves) = one window; heads are looped; a wave owns 32-query tiles.  For a query tile the wave
// computes the TRANSPOSED score tile X_j = K_j Q_i^T for every 32-key tile j (A = K rows, B = Q rows, both read from
// row-major LDS images with ds_read_b128).  An accumulator of that MFMA has the query on the lane and the keys in
// its 16 registers, so (1) the softmax over keys is a per-lane reduction over registers plus one exchange with lane
// l^32, and (2) the probabilities, converted pairwise to bf16, ARE the B operand of the next MFMA O^T += V^T_j P_j
// with no lane movement (cdna_hip_programming.md section 3, "an accumulator tile as the next MFMA's operand";
// the A operand V^T is read from a transposed LDS image with the matching permuted key order).
// Relative-position bias comes from the per-head table in LDS via code_q - code_k, the shifted-window -100 mask from
// region ids, padded keys (N..32*NKT) get -inf.  Same addressing / padded-token semantics as attention.hip.
#include "attention_common.h"

using namespace msseg_attn;

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int HD, int NKT>
__global__ __launch_bounds__(256) void win_attn_fwd_mfma_kernel(const AttnParams p) {
    constexpr int NP = NKT * 32;          // padded token count
    constexpr int KS = HD / 16;           // 32x32x16 k-steps over the head dim
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* qS = (bf16_t*)smem;                    // [NP][HD]
    bf16_t* kS = qS + NP * HD;                     // [NP][HD]
    bf16_t* vT = kS + NP * HD;                     // [HD][NP]   (transposed)
    float* tabS = (float*)(vT + HD * NP);          // [M3]
    int* tok = (int*)(tabS + p.M3);                // [NP]
    unsigned int* kinfo = (unsigned int*)(tok + NP);  // [NP]  code | region << 12 ; 0xFFFFFFFF = padded key
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int nW = p.nWs * p.nWh * p.nWw;
    const int m = 2 * p.bws - 1;
    const int off = ((p.bws - 1) * m + (p.bws - 1)) * m + (p.bws - 1);
    const int C3 = 3 * p.C;

    for (int wb = blockIdx.x; wb < p.nwin_total; wb += gridDim.x) {
        const int w = wb % nW, b = wb / nW;
        const int wx = w % p.nWw, wy = (w / p.nWw) % p.nWh, wz = w / (p.nWw * p.nWh);
        const bf16_t* qkv = (const bf16_t*)p.qkv + (long long)b * p.S * p.H * p.W * C3;
        bf16_t* out = (bf16_t*)p.out + (long long)b * p.S * p.H * p.W * p.C;
        __syncthreads();
        for (int i = tid; i < NP; i += 256) {
            if (i < p.N) {
                int rg, cd;
                tok[i] = window_token(p, wz, wy, wx, i, rg, cd);
                kinfo[i] = (unsigned int)(cd | (rg << 12));
            } else {
                tok[i] = -2;            // padded row of the MFMA tile (not a token)
                kinfo[i] = 0xFFFFFFFFu;
            }
        }
        for (int h = blockIdx.y; h < p.heads; h += gridDim.y) {   // heads split over grid.y when there are few windows
            __syncthreads();
            for (int i = tid; i < p.M3; i += 256) tabS[i] = p.table[(long long)i * p.heads + h];
            // stage Q, K (row-major) and V (transposed): one 16-byte chunk (8 channels) per thread-iteration
            constexpr int CPT = HD / 8;   // chunks per token per matrix
            for (int i = tid; i < NP * CPT * 3; i += 256) {
                const int which = i / (NP * CPT), rem = i % (NP * CPT);
                const int t = rem / CPT, ch = rem % CPT;
                const int tk = tok[t];
                bf16x8_t v;
                if (tk >= 0) {
                    v = *(const bf16x8_t*)(qkv + (long long)tk * C3 + which * p.C + h * HD + ch * 8);
                } else if (tk == -1 && p.qkv_bias) {   // spatially padded token: Linear(0) = bias
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)p.qkv_bias[which * p.C + h * HD + ch * 8 + e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
                }
                if (which == 0) *(bf16x8_t*)(qS + t * HD + ch * 8) = v;
                else if (which == 1) *(bf16x8_t*)(kS + t * HD + ch * 8) = v;
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) vT[(ch * 8 + e) * NP + t] = v[e];
                }
            }
            __syncthreads();
            for (int qt = wave; qt < NKT; qt += 4) {
                const int qi = qt * 32 + r;              // this lane's query (column of every X tile)
                const unsigned int qinfo = kinfo[qi];
                const int qcode = (qinfo == 0xFFFFFFFFu ? 0 : (qinfo & 4095)) + off, qreg = qinfo >> 12;  // padded query rows: any valid code
                // ---- X_j = K_j Q_i^T ----
                bf16x8_t qf[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) qf[s] = *(const bf16x8_t*)(qS + qi * HD + s * 16 + hh * 8);
                // keys in chunks of JC tiles (one chunk up to 7 tiles; 343-token windows take two, with the running max /
                // sum rescale of an online softmax -- the query sits on the lane, so the rescale is a per-lane scalar)
                constexpr int JC = NKT <= 7 ? NKT : 4;
                float mx = -INFINITY, l = 0.f;
                f32x16_t Y = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const int vrow = r & (HD - 1);   // rows >= HD of the 32-row A tile are don't-care
                constexpr int UNR_CHUNKS = JC < NKT ? 1 : 2;   // several chunks: keep ONE chunk's scores live
#pragma unroll UNR_CHUNKS
                for (int j0 = 0; j0 < NKT; j0 += JC) {
                    f32x16_t X[JC];
#pragma unroll
                    for (int jj = 0; jj < JC; ++jj) {
                        const int j = j0 + jj;
                        f32x16_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        if (j < NKT) {
#pragma unroll
                            for (int s = 0; s < KS; ++s) {
                                const bf16x8_t kf = *(const bf16x8_t*)(kS + (j * 32 + r) * HD + s * 16 + hh * 8);
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], acc, 0, 0, 0);
                            }
                        }
                        X[jj] = acc;
                    }
                    // ---- scores: scale, relative-position bias, shift mask, key padding; chunk max ----
                    float mc = -INFINITY;
#pragma unroll
                    for (int jj = 0; jj < JC; ++jj)
#pragma unroll
                        for (int g = 0; g < 16; ++g) {
                            const int j = j0 + jj;
                            const int key = j * 32 + (g & 3) + 8 * (g >> 2) + 4 * hh;
                            const unsigned int ki = j < NKT ? kinfo[key] : 0xFFFFFFFFu;
                            float sc;
                            if (ki == 0xFFFFFFFFu) sc = -INFINITY;
                            else {
                                sc = X[jj][g] * p.scale + tabS[qcode - (ki & 4095)];
                                if (p.use_mask && (ki >> 12) != qreg) sc += -100.f;
                            }
                            X[jj][g] = sc;
                            mc = fmaxf(mc, sc);
                        }
                    mc = fmaxf(mc, __shfl_xor(mc, 32));
                    if constexpr (JC < NKT) {
                        const float mnew = fmaxf(mx, mc);
                        const float alpha = __expf(mx - mnew);     // 0 for the first chunk (mx = -inf)
                        l *= alpha;
#pragma unroll
                        for (int g = 0; g < 16; ++g) Y[g] *= alpha;
                        mx = mnew;
                    } else {
                        mx = mc;
                    }
                    float lc = 0.f;
#pragma unroll
                    for (int jj = 0; jj < JC; ++jj)
#pragma unroll
                        for (int g = 0; g < 16; ++g) {
                            const float pe = __expf(X[jj][g] - mx);
                            X[jj][g] = pe;
                            lc += pe;
                        }
                    lc += __shfl_xor(lc, 32);
                    l += lc;
                    // ---- O^T += V^T_j P_j  (P_j straight from the accumulator registers) ----
#pragma unroll
                    for (int jj = 0; jj < JC; ++jj) {
                        const int j = j0 + jj;
                        if (j < NKT) {
#pragma unroll
                            for (int s = 0; s < 2; ++s) {
                                bf16x8_t pf;
#pragma unroll
                                for (int e = 0; e < 8; ++e) pf[e] = (bf16_t)X[jj][8 * s + e];
                                // element e of lane-half hh of P's fragment is key row 16s + 8(e>>2) + 4hh + (e&3) of tile j
                                const bf16_t* vp = vT + vrow * NP + j * 32 + 16 * s + 4 * hh;
                                const bf16x4_t v0 = *(const bf16x4_t*)vp;
                                const bf16x4_t v1 = *(const bf16x4_t*)(vp + 8);
                                const bf16x8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                                Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, Y, 0, 0, 0);
                            }
                        }
                    }
                }
                // ---- store: lane = query, registers = head-dim rows (reg&3) + 8*(reg>>2) + 4*hh ----
                const int tk = tok[qi];
                const float inv = 1.f / l;
                if (qi < p.N) {
                    if (hh == 0) p.lse[((long long)wb * p.heads + h) * p.N + qi] = mx + __logf(l);
                    if (tk >= 0) {
                        bf16_t* orow = out + (long long)tk * p.C + h * HD;
#pragma unroll
                        for (int g4 = 0; g4 < HD / 8; ++g4) {
                            bf16x4_t o = {(bf16_t)(Y[4 * g4 + 0] * inv), (bf16_t)(Y[4 * g4 + 1] * inv),
                                          (bf16_t)(Y[4 * g4 + 2] * inv), (bf16_t)(Y[4 * g4 + 3] * inv)};
                            *(bf16x4_t*)(orow + 8 * g4 + 4 * hh) = o;
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward (bf16): dQ, dK, dV and the relative-position-bias gradient.
//
// Two passes per (window, head), both on v_mfma_f32_32x32x16_bf16 with the same "accumulator = next B operand" trick
// as the forward:
//   pass 1, a wave per 32-query tile i, keys j streamed:  S^T = K_j Q_i^T and dP^T = V_j dO_i^T have the query on the lane
//           (lse_i, delta_i are per-lane scalars), dS^T = P^T (dP^T - delta_i) in registers IS the B operand of
//           dQ_i^T += K_j^T dS^T; dtable[code_q - code_k] += dS goes to LDS bins with float atomics (as the vector kernel).
//   pass 2, a wave per 32-key tile j, queries i streamed:  S = Q_i K_j^T and dP = dO_i V_j^T have the key on the lane and
//           the queries in the registers; P and dS feed dV_j^T += dO_i^T P and dK_j^T += Q_i^T dS.
// Scores are recomputed in each pass (2 + 2 MFMAs per tile pair at head dim 16) instead of transposing a 32 x 32
// accumulator through LDS.  Semantics (padded tokens carry the qkv bias, padded queries get no gradient, padded rows
// of the MFMA tiles are masked) follow win_attn_bwd_kernel in attention.hip.
template <int HD, int NKT, bool DSWS>
__global__ __launch_bounds__(256) void win_attn_bwd_mfma_kernel(const AttnParams p) {
    constexpr int NP = NKT * 32;
    constexpr int KS = HD / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* qS = (bf16_t*)smem;                    // row-major [NP][HD]
    bf16_t* kS = qS + NP * HD;
    bf16_t* vS = kS + NP * HD;
    bf16_t* oS = vS + NP * HD;                     // dO
    bf16_t* qT = oS + NP * HD;                     // transposed [HD][NP]
    bf16_t* kT = qT + HD * NP;
    bf16_t* oT = kT + HD * NP;
    float* lseS = (float*)(oT + HD * NP);          // [NP]
    float* delS = lseS + NP;                       // [NP]
    float* tabS = delS + NP;                       // [M3]
    float* dtabS = tabS + p.M3;                    // [M3]
    int* tok = (int*)(dtabS + p.M3);               // [NP]
    unsigned int* kinfo = (unsigned int*)(tok + NP);  // [NP] code | region << 12 ; 0xFFFFFFFF = padded row of the tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int nW = p.nWs * p.nWh * p.nWw;
    const int m = 2 * p.bws - 1;
    const int off = ((p.bws - 1) * m + (p.bws - 1)) * m + (p.bws - 1);
    const int C3 = 3 * p.C;
    const int vrow = r & (HD - 1);

    for (int wb = blockIdx.x; wb < p.nwin_total; wb += gridDim.x) {
        const int w = wb % nW, b = wb / nW;
        const int wx = w % p.nWw, wy = (w / p.nWw) % p.nWh, wz = w / (p.nWw * p.nWh);
        const long long vol = (long long)p.S * p.H * p.W;
        const bf16_t* qkv = (const bf16_t*)p.qkv + b * vol * C3;
        const bf16_t* outp = (const bf16_t*)p.out + b * vol * p.C;
        const bf16_t* dout = (const bf16_t*)p.dout + b * vol * p.C;
        bf16_t* dqkv = (bf16_t*)p.dqkv + b * vol * C3;
        __syncthreads();
        for (int i = tid; i < NP; i += 256) {
            if (i < p.N) {
                int rg, cd;
                tok[i] = window_token(p, wz, wy, wx, i, rg, cd);
                kinfo[i] = (unsigned int)(cd | (rg << 12));
            } else {
                tok[i] = -2;
                kinfo[i] = 0xFFFFFFFFu;
            }
        }
        for (int h = blockIdx.y; h < p.heads; h += gridDim.y) {   // heads split over grid.y when there are few windows
            __syncthreads();
            for (int i = tid; i < p.M3; i += 256) { tabS[i] = p.table[(long long)i * p.heads + h]; dtabS[i] = 0.f; }
            bf16_t* dsw = DSWS ? (bf16_t*)p.ds_ws + ((long long)wb * p.heads + h) * (NKT * NKT * 1024) + lane * 16 : nullptr;
            constexpr int CPT = HD / 8;
            for (int i = tid; i < NP * CPT * 4; i += 256) {   // q, k, v, dO
                const int which = i / (NP * CPT), rem = i % (NP * CPT);
                const int t = rem / CPT, ch = rem % CPT;
                const int tk = tok[t];
                bf16x8_t v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
                if (tk >= 0) {
                    v = (which < 3) ? *(const bf16x8_t*)(qkv + (long long)tk * C3 + which * p.C + h * HD + ch * 8)
                                    : *(const bf16x8_t*)(dout + (long long)tk * p.C + h * HD + ch * 8);
                } else if (tk == -1 && p.qkv_bias && which < 3) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)p.qkv_bias[which * p.C + h * HD + ch * 8 + e];
                }
                bf16_t* rowm = which == 0 ? qS : (which == 1 ? kS : (which == 2 ? vS : oS));
                *(bf16x8_t*)(rowm + t * HD + ch * 8) = v;
                bf16_t* colm = which == 0 ? qT : (which == 1 ? kT : (which == 3 ? oT : nullptr));
                if (colm != nullptr) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) colm[(ch * 8 + e) * NP + t] = v[e];
                }
            }
            for (int i = tid; i < NP; i += 256) {
                float d = 0.f, ls = 0.f;
                if (i < p.N) {
                    ls = p.lse[((long long)wb * p.heads + h) * p.N + i];
                    const int t = tok[i];
                    if (t >= 0) {
#pragma unroll
                        for (int c8 = 0; c8 < HD / 8; ++c8) {
                            const bf16x8_t a = *(const bf16x8_t*)(dout + (long long)t * p.C + h * HD + c8 * 8);
                            const bf16x8_t o = *(const bf16x8_t*)(outp + (long long)t * p.C + h * HD + c8 * 8);
#pragma unroll
                            for (int e = 0; e < 8; ++e) d += (float)a[e] * (float)o[e];
                        }
                    }
                }
                lseS[i] = ls;
                delS[i] = d;
            }
            __syncthreads();
            // ------------------------------ pass 1: dQ (+ dtable) ------------------------------
            for (int qt = wave; qt < NKT; qt += 4) {
                const int qi = qt * 32 + r;
                const unsigned int qinfo = kinfo[qi];
                const bool qlive = tok[qi] >= 0;
                const int qcode = (qinfo == 0xFFFFFFFFu ? 0 : (qinfo & 4095)) + off, qreg = qinfo >> 12;
                const float lq = lseS[qi], dq_del = delS[qi];
                bf16x8_t qf[KS], of[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    qf[s] = *(const bf16x8_t*)(qS + qi * HD + s * 16 + hh * 8);
                    of[s] = *(const bf16x8_t*)(oS + qi * HD + s * 16 + hh * 8);
                }
                f32x16_t Y;
#pragma unroll
                for (int g = 0; g < 16; ++g) Y[g] = 0.f;
#pragma unroll 1
                for (int j = 0; j < NKT; ++j) {
                    f32x16_t X, DP;
#pragma unroll
                    for (int g = 0; g < 16; ++g) X[g] = DP[g] = 0.f;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const bf16x8_t kf = *(const bf16x8_t*)(kS + (j * 32 + r) * HD + s * 16 + hh * 8);
                        const bf16x8_t vf = *(const bf16x8_t*)(vS + (j * 32 + r) * HD + s * 16 + hh * 8);
                        X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], X, 0, 0, 0);
                        DP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, of[s], DP, 0, 0, 0);
                    }
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int key = j * 32 + (g & 3) + 8 * (g >> 2) + 4 * hh;
                        const unsigned int ki = kinfo[key];
                        float ds = 0.f;
                        // (a padded query's dQ row is never written, so with DSWS its dS may be zeroed here: the stored
                        // tile then holds exactly the table-gradient contributions)
                        if (ki != 0xFFFFFFFFu && qinfo != 0xFFFFFFFFu && (!DSWS || qlive)) {
                            const int ti = qcode - (ki & 4095);
                            float sc = X[g] * p.scale + tabS[ti];
                            if (p.use_mask && (ki >> 12) != qreg) sc += -100.f;
                            const float pr = __expf(sc - lq);
                            ds = pr * (DP[g] - dq_del);
                            if (!DSWS && p.dtable && qlive) atomicAdd(&dtabS[ti], ds);
                        }
                        X[g] = ds;
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        bf16x8_t df;
#pragma unroll
                        for (int e = 0; e < 8; ++e) df[e] = (bf16_t)X[8 * s + e];
                        if (DSWS) *(bf16x8_t*)(dsw + (qt * NKT + j) * 1024 + 8 * s) = df;   // 2 KB per wave and tile pair
                        const bf16_t* kp = kT + vrow * NP + j * 32 + 16 * s + 4 * hh;
                        const bf16x4_t k0 = *(const bf16x4_t*)kp;
                        const bf16x4_t k1 = *(const bf16x4_t*)(kp + 8);
                        const bf16x8_t kf = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
                        Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, df, Y, 0, 0, 0);
                    }
                }
                const int tk = tok[qi];
                if (qi < p.N && tk >= 0) {
                    bf16_t* drow = dqkv + (long long)tk * C3 + h * HD;
#pragma unroll
                    for (int g4 = 0; g4 < HD / 8; ++g4) {
                        const bf16x4_t o = {(bf16_t)(Y[4 * g4 + 0] * p.scale), (bf16_t)(Y[4 * g4 + 1] * p.scale),
                                            (bf16_t)(Y[4 * g4 + 2] * p.scale), (bf16_t)(Y[4 * g4 + 3] * p.scale)};
                        *(bf16x4_t*)(drow + 8 * g4 + 4 * hh) = o;
                    }
                }
            }
            // ------------------------------ pass 2: dK, dV ------------------------------
            for (int jt = wave; jt < NKT; jt += 4) {
                const int kj = jt * 32 + r;
                const unsigned int kinf = kinfo[kj];
                const int kcode = (kinf == 0xFFFFFFFFu) ? 0 : (kinf & 4095), kreg = kinf >> 12;
                bf16x8_t kf[KS], vf[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    kf[s] = *(const bf16x8_t*)(kS + kj * HD + s * 16 + hh * 8);
                    vf[s] = *(const bf16x8_t*)(vS + kj * HD + s * 16 + hh * 8);
                }
                f32x16_t YK, YV;
#pragma unroll
                for (int g = 0; g < 16; ++g) YK[g] = YV[g] = 0.f;
#pragma unroll 1
                for (int i = 0; i < NKT; ++i) {
                    f32x16_t X, DP;
#pragma unroll
                    for (int g = 0; g < 16; ++g) X[g] = DP[g] = 0.f;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const bf16x8_t qf = *(const bf16x8_t*)(qS + (i * 32 + r) * HD + s * 16 + hh * 8);
                        const bf16x8_t of = *(const bf16x8_t*)(oS + (i * 32 + r) * HD + s * 16 + hh * 8);
                        X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[s], X, 0, 0, 0);
                        DP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of, vf[s], DP, 0, 0, 0);
                    }
                    f32x16_t PR;
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int qrow = i * 32 + (g & 3) + 8 * (g >> 2) + 4 * hh;
                        const unsigned int qi2 = kinfo[qrow];
                        float pr = 0.f, ds = 0.f;
                        if (qi2 != 0xFFFFFFFFu && kinf != 0xFFFFFFFFu && tok[qrow] >= 0) {   // padded queries get no gradient
                            float sc = X[g] * p.scale + tabS[(qi2 & 4095) + off - kcode];
                            if (p.use_mask && (qi2 >> 12) != kreg) sc += -100.f;
                            pr = __expf(sc - lseS[qrow]);
                            ds = pr * (DP[g] - delS[qrow]);
                        }
                        PR[g] = pr;
                        X[g] = ds;
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        bf16x8_t df, pf;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { df[e] = (bf16_t)X[8 * s + e]; pf[e] = (bf16_t)PR[8 * s + e]; }
                        const bf16_t* qp = qT + vrow * NP + i * 32 + 16 * s + 4 * hh;
                        const bf16_t* op = oT + vrow * NP + i * 32 + 16 * s + 4 * hh;
                        const bf16x4_t q0 = *(const bf16x4_t*)qp, q1 = *(const bf16x4_t*)(qp + 8);
                        const bf16x4_t o0 = *(const bf16x4_t*)op, o1 = *(const bf16x4_t*)(op + 8);
                        const bf16x8_t qf2 = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
                        const bf16x8_t of2 = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
                        YK = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf2, df, YK, 0, 0, 0);
                        YV = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of2, pf, YV, 0, 0, 0);
                    }
                }
                const int tk = tok[kj];
                if (kj < p.N && tk >= 0) {
                    bf16_t* drow = dqkv + (long long)tk * C3 + h * HD;
#pragma unroll
                    for (int g4 = 0; g4 < HD / 8; ++g4) {
                        const bf16x4_t dk = {(bf16_t)(YK[4 * g4 + 0] * p.scale), (bf16_t)(YK[4 * g4 + 1] * p.scale),
                                             (bf16_t)(YK[4 * g4 + 2] * p.scale), (bf16_t)(YK[4 * g4 + 3] * p.scale)};
                        const bf16x4_t dv = {(bf16_t)YV[4 * g4 + 0], (bf16_t)YV[4 * g4 + 1], (bf16_t)YV[4 * g4 + 2],
                                             (bf16_t)YV[4 * g4 + 3]};
                        *(bf16x4_t*)(drow + p.C + 8 * g4 + 4 * hh) = dk;
                        *(bf16x4_t*)(drow + 2 * p.C + 8 * g4 + 4 * hh) = dv;
                    }
                }
            }
            if (!DSWS && p.dtable) {
                __syncthreads();
                for (int i = tid; i < p.M3; i += 256) {
                    const float v = dtabS[i];
                    if (v != 0.f) atomicAdd(&p.dtable[(long long)i * p.heads + h], v);
                }
            }
        }
    }
}

// ---- table gradient without atomics (DSWS) ----
// step 1: sum the bf16 dS tiles over the windows of one group; element order is the MFMA fragment order the backward
// kernel wrote, identical for every window, so this is a plain strided sum (16 B per thread and window)
__global__ __launch_bounds__(256) void attn_ds_window_sum_kernel(const bf16_t* __restrict__ ws, float* __restrict__ psum,
                                                                 int nwin, int heads, int E, int chunk) {
    const int e8 = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y, g = blockIdx.z;
    if (e8 * 8 >= E) return;
    const int w0 = g * chunk, w1 = min(nwin, w0 + chunk);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const long long stride = (long long)heads * E;
    const bf16_t* src = ws + ((long long)w0 * heads + h) * E + e8 * 8;
    int w = w0;
    for (; w + 4 <= w1; w += 4) {
        bf16x8_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8_t*)(src + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[u][e];
        src += 4 * stride;
    }
    for (; w < w1; ++w) {
        const bf16x8_t v = *(const bf16x8_t*)src;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        src += stride;
    }
    float* dst = psum + ((long long)g * heads + h) * E + e8 * 8;
    *(float4*)dst = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *(float4*)(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// step 2: one workgroup per (table entry, head).  The entry's relative offset (dz, dy, dx) fixes the key of every query:
// thread i reads the summed dS of (query i, key i - offset) from each group and the block adds them up in a fixed order.
__global__ __launch_bounds__(256) void attn_dtable_gather_kernel(const float* __restrict__ psum, float* __restrict__ dtable,
                                                                 int groups, int heads, int nkt, int wsz, int N) {
    __shared__ float red[256];
    const int ti = blockIdx.x, h = blockIdx.y, i = threadIdx.x;
    const int m = 2 * wsz - 1;
    const int dz = ti / (m * m) - (wsz - 1), dy = (ti / m) % m - (wsz - 1), dx = ti % m - (wsz - 1);
    float v = 0.f;
    for (int qi = i; qi < N; qi += 256) {     // windows of up to 352 tokens: a thread may own two queries
        const int jz = qi / (wsz * wsz) - dz, jy = (qi / wsz) % wsz - dy, jx = qi % wsz - dx;
        const int j = (jz * wsz + jy) * wsz + jx;
        if (jz >= 0 && jz < wsz && jy >= 0 && jy < wsz && jx >= 0 && jx < wsz && j < N) {
            // fragment order of the backward kernel's pass 1: tile (i / 32, j / 32), lane = (i % 32) + 32 * hh,
            // register g with key offset (g & 3) + 8 * (g >> 2) + 4 * hh
            const int kk = j & 31, hh = (kk >> 2) & 1, g = (kk & 3) + 4 * (kk >> 3);
            const long long e = ((long long)((qi >> 5) * nkt + (j >> 5)) * 64 + (qi & 31) + 32 * hh) * 16 + g;
            const long long E = (long long)nkt * nkt * 1024;
            for (int q = 0; q < groups; ++q) v += psum[((long long)q * heads + h) * E + e];
        }
    }
    red[i] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (i < s) red[i] += red[i + s];
        __syncthreads();
    }
    if (i == 0) dtable[(long long)ti * heads + h] += red[0];
}

constexpr int DS_GROUPS = 8;

template <int HD, int NKT> int launch_bwd(const AttnParams& p, hipStream_t stream) {
    constexpr int NP = NKT * 32;
    const size_t smem = (size_t)7 * NP * HD * 2 + (size_t)2 * NP * 4 + (size_t)2 * p.M3 * 4 + (size_t)NP * 4 + (size_t)NP * 4 + 16;
    const bool dsws = p.ds_ws != nullptr && p.dtable != nullptr;
    auto kern = dsws ? win_attn_bwd_mfma_kernel<HD, NKT, true> : win_attn_bwd_mfma_kernel<HD, NKT, false>;
    static msseg_lds_attr_once attr[2];
    if (!attr[dsws].ensure((const void*)kern, 160 * 1024))
        MSSEG_FAIL(MSSEG_ELAUNCH, "window_attention_bwd_mfma: cannot set dynamic LDS size");
    if (smem > 160 * 1024) MSSEG_FAIL(MSSEG_EINVAL, "window_attention_bwd_mfma: window too large for LDS (%zu bytes)", smem);
    int gx = p.nwin_total < msseg_num_cus() * 2 ? p.nwin_total : msseg_num_cus() * 2;
    int gy = (msseg_num_cus() * 2 + gx - 1) / gx;      // few windows (deep stages): spread the heads over workgroups too
    if (gy > p.heads) gy = p.heads;
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), smem, stream, p);
    MSSEG_CHECK_LAUNCH("window_attention_bwd_mfma");
    if (dsws) {
        const int E = NKT * NKT * 1024;
        const int groups = p.ds_groups;
        const int chunk = (p.nwin_total + groups - 1) / groups;
        hipLaunchKernelGGL(attn_ds_window_sum_kernel, dim3((E / 8 + 255) / 256, p.heads, groups), dim3(256), 0, stream,
                           (const bf16_t*)p.ds_ws, p.ds_psum, p.nwin_total, p.heads, E, chunk);
        MSSEG_CHECK_LAUNCH("attn_ds_window_sum");
        hipLaunchKernelGGL(attn_dtable_gather_kernel, dim3(p.M3, p.heads), dim3(256), 0, stream, (const float*)p.ds_psum,
                           p.dtable, groups, p.heads, NKT, p.bws, p.N);
        MSSEG_CHECK_LAUNCH("attn_dtable_gather");
    }
    return MSSEG_OK;
}

template <int HD> int launch_bwd_hd(const AttnParams& p, hipStream_t stream) {
    const int nkt = (p.N + 31) / 32;
    if (nkt == 1) return launch_bwd<HD, 1>(p, stream);
    if (nkt == 2) return launch_bwd<HD, 2>(p, stream);
    if (nkt <= 4) return launch_bwd<HD, 4>(p, stream);
    if (nkt <= 7) return launch_bwd<HD, 7>(p, stream);
    return launch_bwd<HD, 11>(p, stream);
}

template <int HD, int NKT> int launch(const AttnParams& p, hipStream_t stream) {
    constexpr int NP = NKT * 32;
    const size_t smem = (size_t)3 * NP * HD * 2 + (size_t)p.M3 * 4 + (size_t)NP * 4 + (size_t)NP * 4 + 16;
    int gx = p.nwin_total < msseg_num_cus() * 4 ? p.nwin_total : msseg_num_cus() * 4;
    if (smem > 160 * 1024) MSSEG_FAIL(MSSEG_EINVAL, "window_attention_fwd_mfma: window too large for LDS (%zu bytes)", smem);
    if (smem > 64 * 1024) {
        static msseg_lds_attr_once attr;
        if (!attr.ensure((const void*)win_attn_fwd_mfma_kernel<HD, NKT>, 160 * 1024))
            MSSEG_FAIL(MSSEG_ELAUNCH, "window_attention_fwd_mfma: cannot set dynamic LDS size");
    }
    int gy = (msseg_num_cus() * 4 + gx - 1) / gx;
    if (gy > p.heads) gy = p.heads;
    hipLaunchKernelGGL((win_attn_fwd_mfma_kernel<HD, NKT>), dim3(gx, gy), dim3(256), smem, stream, p);
    MSSEG_CHECK_LAUNCH("window_attention_fwd_mfma");
    return MSSEG_OK;
}

template <int HD> int launch_hd(const AttnParams& p, hipStream_t stream) {
    const int nkt = (p.N + 31) / 32;
    if (nkt == 1) return launch<HD, 1>(p, stream);
    if (nkt == 2) return launch<HD, 2>(p, stream);
    if (nkt <= 4) return launch<HD, 4>(p, stream);
    if (nkt <= 7) return launch<HD, 7>(p, stream);
    return launch<HD, 11>(p, stream);     // window 7: 343 tokens (MONAI Swin-UNETR)
}

}  // namespace

static int bwd_nkt(const AttnParams& p) {   // the NKT instantiation launch_bwd_hd picks
    const int nkt = (p.N + 31) / 32;
    return nkt == 1 ? 1 : (nkt == 2 ? 2 : (nkt <= 4 ? 4 : (nkt <= 7 ? 7 : 11)));
}

static int ds_groups_for(const AttnParams& p) { return p.nwin_total < DS_GROUPS ? p.nwin_total : DS_GROUPS; }

size_t msseg_window_attention_bwd_mfma_ws_bytes(const AttnParams& p) {
    const size_t E = (size_t)bwd_nkt(p) * bwd_nkt(p) * 1024;
    return (size_t)p.nwin_total * p.heads * E * 2 + (size_t)ds_groups_for(p) * p.heads * E * 4;
}

void msseg_window_attention_bwd_mfma_carve(AttnParams& p, void* workspace) {
    const size_t E = (size_t)bwd_nkt(p) * bwd_nkt(p) * 1024;
    p.ds_ws = workspace;
    p.ds_psum = (float*)((char*)workspace + (size_t)p.nwin_total * p.heads * E * 2);
    p.ds_groups = ds_groups_for(p);
}

int msseg_window_attention_bwd_mfma(const AttnParams& p, hipStream_t stream) {
    if (p.M3 > 4095 || p.N > 352) MSSEG_FAIL(MSSEG_EINVAL, "window_attention_bwd_mfma: window too large");
    return p.hd == 16 ? launch_bwd_hd<16>(p, stream) : launch_bwd_hd<32>(p, stream);
}

int msseg_window_attention_fwd_mfma(const AttnParams& p, hipStream_t stream) {
    if (p.M3 > 4095 || p.N > 352) MSSEG_FAIL(MSSEG_EINVAL, "window_attention_fwd_mfma: window too large");
    return p.hd == 16 ? launch_hd<16>(p, stream) : launch_hd<32>(p, stream);
}
