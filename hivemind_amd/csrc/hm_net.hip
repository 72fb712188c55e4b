// hm_net.hip — RISEv3 forward as ONE gfx950 kernel launch (the evaluator's hot op).
//
// Replaces the TensorRT FP16 plan of nn/engine.cc:290-401,577-650 for the network of
// src/architectures/rise_mobile_v3.py:105-230 (deployed form: BatchNorm folded, ECA reduced to its
// centre tap).  Design for MI355X, small-batch regime of the search (a few hundred rows):
//   * one workgroup (4 waves) owns one position for the whole network: its 8x8xC activation tile
//     lives in LDS from the stem to the heads, so the only HBM traffic is the 9.5 KB of input
//     planes, the 18.7 KB of policy logits and the (L2-resident) weights;
//   * every convolution is an MFMA GEMM with M = the 64 squares: 1x1 convs directly, 3x3 convs as an
//     implicit im2col over the LDS tile (a zero row stands in for off-board taps); weights are
//     pre-packed on the host in v_mfma_f32_32x32x16_f16 fragment order, so a wave fetches a B (or
//     transposed A) fragment with one coalesced 1 KiB load and never stages weights through LDS;
//   * the mobile block is fused end to end: 1x1 expand (+bias, ReLU) -> LDS [ch][sq] -> depthwise
//     kxk on the VALU (+bias, ReLU) -> LDS [sq][ch] -> 1x1 project accumulated in registers across
//     64-channel chunks -> + bias + (ECA-gated) residual -> back into the LDS tile.
// Launch overhead and the ~100 small kernels of the library path disappear; the kernel is bound by
// MFMA issue + LDS fragment reads (M = 64 rows per workgroup), not by HBM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/hivemind_amd.h"
#include "hm_queue.hpp"
#include "hm_policy.hpp"

int hm_fail(int code, const std::string& msg);

#include "hm_net_device.hpp"

namespace hmn {

// CT = C / 64 accumulator tiles per wave; K5 = some block uses a 5x5 depthwise; OCC = workgroups the register budget must
// leave room for per CU.  The search hands over a few hundred rows at a time — at most one workgroup per CU — so the
// narrow nets also come in an OCC = 1 form that may use the whole register file (the OCC = 2 form of <2,*> spills ~100
// VGPRs per lane to scratch, 57 MB of scratch writes per launch in the round-1 PMC profile); large batches keep OCC = 2.
template <int CT, bool K5, int OCC>
__global__ __launch_bounds__(256, OCC) void rise_forward_kernel(const NetDesc* __restrict__ ndp, const h16* __restrict__ wh, const float* __restrict__ wf,
                                                           const h16* __restrict__ planes, int n, int stageBytes,
                                                           h16* __restrict__ value, h16* __restrict__ piA, h16* __restrict__ piB,
                                                           h16* __restrict__ wdl, h16* __restrict__ ml,
                                                           const int* __restrict__ groupRows, int group,
                                                           unsigned long long* __restrict__ dbg, unsigned long long* __restrict__ clk) {
    if (clk && threadIdx.x == 0) atomicMin(clk, (unsigned long long)__builtin_amdgcn_s_memrealtime());   // leg clock: (start, end) pair
    // diagnostic stamps (dbg != nullptr only from hm_net_profile): cycle counter at phase boundaries of block 0
    int dbgN = 0;
#define HM_STAMP() do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0 && dbgN < 256) dbg[dbgN++] = __builtin_amdgcn_s_memtime(); } while (0)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const NetDesc& nd = *ndp;                 // descriptor stays in (scalar-loadable) global memory
    const int C = nd.C;
    const int ldx = C + 8;
    h16* Xs = reinterpret_cast<h16*>(smem);                       // [65][ldx]   activation tile (+ zero row)
    h16* Ss = Xs + 65 * ldx;                                      // [65][ldx]   policy trunk / input staging
    h16* Y1 = Ss + 65 * ldx;                                      // [64][66]    expand output [ch][sq]
    h16* Y2 = Y1 + 64 * 66;                                       // [64][72]    depthwise output [sq][ch]
    float* Mv = reinterpret_cast<float*>(Y2 + 64 * 72);           // [C] channel means / scratch
    float* Gv = Mv + C;                                           // [C] gates / scratch
    // optional per-block parameter stage (biases + depthwise weights): one coalesced copy per block
    // instead of latency-exposed global loads in every chunk phase
    unsigned char* Pst = reinterpret_cast<unsigned char*>(Gv + C);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int sIdx = blockIdx.x; sIdx < n; sIdx += gridDim.x) {
        // rows beyond a group's live count carry no position this step (uniform per workgroup)
        if (groupRows && (sIdx % group) >= groupRows[sIdx / group]) continue;
        HM_STAMP();   // 0: start
        // ---- input planes: NCHW [74][64] fp16 -> Ss as [sq][cin_pad] (+ zero row 64)
        const int ldi = nd.cin_pad + 8;
        for (int i = tid; i < 65 * ldi; i += 256) Ss[i] = (h16)0.0f;
        for (int i = tid; i < ldx; i += 256) Xs[64 * ldx + i] = (h16)0.0f;
        __syncthreads();
        const h16* pin = planes + (size_t)sIdx * HM_PLANE_VALUES;
        for (int i = tid; i < HM_PLANE_VALUES; i += 256) Ss[(i & 63) * ldi + (i >> 6)] = pin[i];
        __syncthreads();
        HM_STAMP();   // 1: input staged
        // ---- stem: 3x3 conv cin -> C, + bias, ReLU
        {
            floatx16 acc[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) acc[i] = zero16();
            gemm_rows<CT>(acc, Ss, ldi, 64, true, nd.cin_pad, 9 * nd.cin_pad / 16, wh + nd.stem_w, C, wave, lane);
            const float* b = wf + nd.stem_b;
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int co = ((wave >> 1) + 2 * i) * 32 + (lane & 31);
                const float bias = b[co];
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int sq = (wave & 1) * 32 + drow(rg, lane);
                    Xs[sq * ldx + co] = (h16)fmaxf(acc[i][rg] + bias, 0.0f);
                }
            }
        }
        __syncthreads();
        HM_STAMP();   // 2: stem done
        // ---- mobile bottleneck blocks
        for (int bi = 0; bi < nd.nblocks; ++bi) {
            HM_STAMP();   // block start
            const BlockDesc bd = nd.blk[bi];
            if (bd.eca) {   // x = x * hardsigmoid(W_eca . mean_sq(x) + b)   (builder_util.py:49-80, centre tap)
                for (int c = tid; c < C; c += 256) {
                    float s = 0.0f;
                    for (int sq = 0; sq < 64; ++sq) s += (float)Xs[sq * ldx + c];
                    Mv[c] = s * (1.0f / 64.0f);
                }
                __syncthreads();
                const h16* we = wh + bd.ecaw;                     // [ci][co]
                for (int co = tid; co < C; co += 256) {
                    float s = wf[bd.ecab + co];
                    for (int ci0 = 0; ci0 < C; ci0 += 16) {            // 16 independent (coalesced) loads in flight
                        h16 wv[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) wv[u] = we[(size_t)(ci0 + u) * C + co];
#pragma unroll
                        for (int u = 0; u < 16; ++u) s += (float)wv[u] * Mv[ci0 + u];
                    }
                    Gv[co] = fminf(fmaxf(s * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f);
                }
                __syncthreads();
                for (int i = tid; i < 64 * C; i += 256) {
                    const int sq = i / C, c = i - sq * C;
                    Xs[sq * ldx + c] = (h16)((float)Xs[sq * ldx + c] * Gv[c]);
                }
                __syncthreads();
            }
            const int cop = bd.cop, kk = bd.k;
            const float* pb1 = wf + bd.b1;
            const float* pb2 = wf + bd.b2;
            const float* pb3 = wf + bd.b3;
            const h16* pdw = wh + bd.dw;
            if (stageBytes > 0) {
                float* sb1 = reinterpret_cast<float*>(Pst);
                float* sb2 = sb1 + cop;
                float* sb3 = sb2 + cop;
                h16* sdw = reinterpret_cast<h16*>(sb3 + C);
                for (int i = tid; i < cop; i += 256) { sb1[i] = pb1[i]; sb2[i] = pb2[i]; }
                for (int i = tid; i < C; i += 256) sb3[i] = pb3[i];
                for (int i = tid; i < cop * kk * kk; i += 256) sdw[i] = pdw[i];
                pb1 = sb1; pb2 = sb2; pb3 = sb3; pdw = sdw;
                __syncthreads();
            }
            HM_STAMP();   // eca + param stage done
            floatx16 acc[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) acc[i] = zero16();
            const int copTiles = cop >> 5;
            for (int c0 = 0; c0 < cop; c0 += 64) {
                const int cw = cop - c0 < 64 ? cop - c0 : 64;       // 64 or 32 channels in this chunk
                // 1x1 expand, transposed: Y1[ch][sq] = relu(W1^T x^T + b1)
                if ((wave >> 1) * 32 < cw) {
                    const int wt = (c0 >> 5) + (wave >> 1);
                    floatx16 e = gemm_cols(Xs, ldx, 64, false, C, C / 16, wh + bd.w1, copTiles, wt, wave, lane);
                    const int sq = (wave & 1) * 32 + (lane & 31);
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int ch = (wave >> 1) * 32 + drow(rg, lane);
                        Y1[ch * 66 + sq] = (h16)fmaxf(e[rg] + pb1[c0 + ch], 0.0f);
                    }
                }
                __syncthreads();
                HM_STAMP();   // expand done
                // depthwise kxk (+bias, ReLU): thread = (channel, 2 board rows); output transposed to [sq][ch]
                {
                    const int ch = tid & 63, g = tid >> 6;
                    if (ch < cw) {
                        const h16* wd = pdw + (size_t)(c0 + ch) * kk * kk;
                        const float bias = pb2[c0 + ch];
                        if (!K5 || kk == 3) depthwise_rows<3>(Y1 + ch * 66, Y2, ch, g, wd, bias);
                        else depthwise_rows<5>(Y1 + ch * 66, Y2, ch, g, wd, bias);
                    }
                }
                __syncthreads();
                HM_STAMP();   // depthwise done
                // 1x1 project: acc[sq][co] += Y2[sq][chunk] . W2[chunk][co]
                if (cw == 64) gemm_rows<CT>(acc, Y2, 72, 0, false, 64, 4, wh + bd.w2 + (size_t)(c0 >> 4) * (C >> 5) * 512, C, wave, lane);
                else          gemm_rows<CT>(acc, Y2, 72, 0, false, 64, 2, wh + bd.w2 + (size_t)(c0 >> 4) * (C >> 5) * 512, C, wave, lane);
                __syncthreads();
                HM_STAMP();   // project done
            }
            // residual: x = x + (acc + b3); every wave owns disjoint (sq, co) elements
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int co = ((wave >> 1) + 2 * i) * 32 + (lane & 31);
                const float bias = pb3[co];
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int sq = (wave & 1) * 32 + drow(rg, lane);
                    Xs[sq * ldx + co] = (h16)((float)Xs[sq * ldx + co] + acc[i][rg] + bias);
                }
            }
            __syncthreads();
        }
        HM_STAMP();   // blocks done
        // ---- value head: 1x1 conv C -> cv (+bias, ReLU), NCHW flatten, linear -> (wdl x3, plys)
        {
            const int cv = nd.cv;                                  // <= 32
            if (wave < 2) {                                        // one 32-channel tile, two square tiles
                floatx16 e = gemm_cols(Xs, ldx, 64, false, C, C / 16, wh + nd.v_w, 1, 0, wave, lane);
                const int sq = (wave & 1) * 32 + (lane & 31);
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int ch = drow(rg, lane);
                    if (ch < cv) Y1[ch * 66 + sq] = (h16)fmaxf(e[rg] + wf[nd.v_b + ch], 0.0f);
                }
            }
            __syncthreads();
            float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const h16* wl = wh + nd.vl_w;                          // [4][cv*64], NCHW flatten order
            for (int i = tid; i < cv * 64; i += 256) {
                const float v = (float)Y1[(i >> 6) * 66 + (i & 63)];
#pragma unroll
                for (int o = 0; o < 4; ++o) part[o] += v * (float)wl[(size_t)o * cv * 64 + i];
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float v = part[o];
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                if (lane == 0) Mv[wave * 4 + o] = v;
            }
            __syncthreads();
            if (tid == 0) {
                float lo[4];
                for (int o = 0; o < 4; ++o) lo[o] = Mv[o] + Mv[4 + o] + Mv[8 + o] + Mv[12 + o] + wf[nd.vl_b + o];
                const float mx = fmaxf(lo[0], fmaxf(lo[1], lo[2]));
                const float e0 = __expf(lo[0] - mx), e1 = __expf(lo[1] - mx), e2 = __expf(lo[2] - mx);
                const float inv = 1.0f / (e0 + e1 + e2);
                value[sIdx] = (h16)((e2 - e0) * inv);              // win - loss (builder_util.py:315-324)
                wdl[(size_t)sIdx * 3 + 0] = (h16)lo[0]; wdl[(size_t)sIdx * 3 + 1] = (h16)lo[1]; wdl[(size_t)sIdx * 3 + 2] = (h16)lo[2];
                ml[sIdx] = (h16)(1.0f / (1.0f + __expf(-lo[3])));
            }
        }
        HM_STAMP();   // value head done
        // ---- policy heads: shared 3x3 conv C -> C (+bias, ReLU) into Ss, then 3x3 C -> 146 (two boards)
        for (int i = tid; i < ldx; i += 256) Ss[64 * ldx + i] = (h16)0.0f;
        {
            floatx16 acc[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) acc[i] = zero16();
            gemm_rows<CT>(acc, Xs, ldx, 64, true, C, 9 * C / 16, wh + nd.ps_w, C, wave, lane);
            const float* b = wf + nd.ps_b;
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int co = ((wave >> 1) + 2 * i) * 32 + (lane & 31);
                const float bias = b[co];
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int sq = (wave & 1) * 32 + drow(rg, lane);
                    Ss[sq * ldx + co] = (h16)fmaxf(acc[i][rg] + bias, 0.0f);
                }
            }
        }
        __syncthreads();
        HM_STAMP();   // policy trunk done
        {
            // 146 output planes padded to 160 = 5 channel tiles x 2 square tiles = 10 tiles over 4 waves
            for (int t = wave; t < 10; t += 4) {
                const int ctile = t >> 1, stile = t & 1;
                floatx16 e = gemm_cols(Ss, ldx, 64, true, C, 9 * C / 16, wh + nd.pp_w, 5, ctile, stile, lane);
                const int sq = stile * 32 + (lane & 31);
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int ch = ctile * 32 + drow(rg, lane);
                    if (ch < 73) piA[(size_t)sIdx * HM_POLICY_VALUES + ch * 64 + sq] = (h16)e[rg];
                    else if (ch < 146) piB[(size_t)sIdx * HM_POLICY_VALUES + (ch - 73) * 64 + sq] = (h16)e[rg];
                }
            }
        }
        __syncthreads();
        HM_STAMP();   // projection done
    }
    if (clk && threadIdx.x == 0) atomicMax(clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#undef HM_STAMP
}


template <int CTILES, bool K5>
__global__ __launch_bounds__(512, 1) void rise_forward_narrow(const NetDesc* __restrict__ ndp, const h16* __restrict__ wh, const float* __restrict__ wf,
                                                              const h16* __restrict__ planes, int n, int copMax, int uHalfs,
                                                              h16* __restrict__ value, h16* __restrict__ piA, h16* __restrict__ piB,
                                                              h16* __restrict__ wdl, h16* __restrict__ ml,
                                                              const int* __restrict__ groupRows, int group,
                                                              unsigned long long* __restrict__ dbg, unsigned long long* __restrict__ clk) {
    if (clk && threadIdx.x == 0) atomicMin(clk, (unsigned long long)__builtin_amdgcn_s_memrealtime());   // leg clock: (start, end) pair
    int dbgN = 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const NetDesc& nd = *ndp;
    for (int sIdx = blockIdx.x; sIdx < n; sIdx += gridDim.x) {
        if (groupRows && (sIdx % group) >= groupRows[sIdx / group]) continue;
        narrow_position<CTILES, K5>(nd, wh, wf, planes + (size_t)sIdx * HM_PLANE_VALUES, (size_t)sIdx, copMax, uHalfs, smem, value, piA, piB, wdl, ml, dbg, dbgN);
    }
    if (clk && threadIdx.x == 0) atomicMax(clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

// The 4-wave form of the same launch (narrow_position4): 256-thread workgroups, every weight fragment fetched once per workgroup.
template <int CTILES, bool K5>
__global__ __launch_bounds__(256, 1) void rise_forward_narrow4(const NetDesc* __restrict__ ndp, const h16* __restrict__ wh, const float* __restrict__ wf,
                                                               const h16* __restrict__ planes, int n, int copMax, int uHalfs,
                                                               h16* __restrict__ value, h16* __restrict__ piA, h16* __restrict__ piB,
                                                               h16* __restrict__ wdl, h16* __restrict__ ml,
                                                               const int* __restrict__ groupRows, int group,
                                                               unsigned long long* __restrict__ dbg, unsigned long long* __restrict__ clk) {
    if (clk && threadIdx.x == 0) atomicMin(clk, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    int dbgN = 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const NetDesc& nd = *ndp;
    for (int sIdx = blockIdx.x; sIdx < n; sIdx += gridDim.x) {
        if (groupRows && (sIdx % group) >= groupRows[sIdx / group]) continue;
        narrow_position4<CTILES, K5>(nd, wh, wf, planes + (size_t)sIdx * HM_PLANE_VALUES, (size_t)sIdx, copMax, uHalfs, smem, value, piA, piB, wdl, ml, dbg, dbgN);
    }
    if (clk && threadIdx.x == 0) atomicMax(clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

}  // namespace hmn

// One loaded network: descriptor on the device, launch geometry resolved once.  Re-entrant per handle; no global state.
struct hm_net {
    hmn::NetDesc nd;
    hmn::NetDesc* d_nd = nullptr;
    const void* d_wh = nullptr;
    const void* d_wf = nullptr;
    void* owned[2] = {nullptr, nullptr};     // parameter buffers this handle allocated (hm_net_create_host), else caller-owned
    size_t lds = 0, stage = 0;
    bool k5 = false;
    bool narrow = false;                     // C <= 128: the 8-wave latency-first kernel (rise_forward_narrow)
    int waves = 8;                           // workgroup size of the narrow form: 8 (rise_forward_narrow) or 4 (rise_forward_narrow4); HM_NET_WAVES
    int copMax = 0, uHalfs = 0;
    size_t ldsNarrow = 0;
    int maxBlocks = 1024;
    int wideRows = 256;                      // batches up to this many rows (one workgroup per CU) run the OCC = 1 instantiation
};

template <typename K>
static hipError_t launch_forward(K kern, const hm_net* net, int grid, hipStream_t st, const void* d_planes, int n, void* d_value, void* d_pi_a,
                                 void* d_pi_b, void* d_wdl, void* d_moves_left, const int32_t* d_group_rows, int group, unsigned long long* d_dbg, unsigned long long* d_clk) {
    using namespace hmn;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), net->lds, st, net->d_nd, static_cast<const h16*>(net->d_wh), static_cast<const float*>(net->d_wf),
                       static_cast<const h16*>(d_planes), n, (int)net->stage, static_cast<h16*>(d_value), static_cast<h16*>(d_pi_a),
                       static_cast<h16*>(d_pi_b), static_cast<h16*>(d_wdl), static_cast<h16*>(d_moves_left),
                       reinterpret_cast<const int*>(d_group_rows), group, d_dbg, d_clk);
    return hipGetLastError();
}
// applies `f` to the kernel instantiation this network runs on
template <typename K>
static hipError_t launch_narrow(K kern, const hm_net* net, int grid, hipStream_t st, const void* d_planes, int n, void* d_value, void* d_pi_a,
                                void* d_pi_b, void* d_wdl, void* d_moves_left, const int32_t* d_group_rows, int group, unsigned long long* d_dbg, unsigned long long* d_clk) {
    using namespace hmn;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * net->waves), net->ldsNarrow, st, net->d_nd, static_cast<const h16*>(net->d_wh), static_cast<const float*>(net->d_wf),
                       static_cast<const h16*>(d_planes), n, net->copMax, net->uHalfs, static_cast<h16*>(d_value), static_cast<h16*>(d_pi_a),
                       static_cast<h16*>(d_pi_b), static_cast<h16*>(d_wdl), static_cast<h16*>(d_moves_left),
                       reinterpret_cast<const int*>(d_group_rows), group, d_dbg, d_clk);
    return hipGetLastError();
}
template <typename F>
static hipError_t with_narrow(const hm_net* net, F f) {
    using namespace hmn;
    if (net->waves == 4) {
        if (net->nd.C == 64) return net->k5 ? f(rise_forward_narrow4<2, true>) : f(rise_forward_narrow4<2, false>);
        if (net->nd.C == 384) return net->k5 ? f(rise_forward_narrow4<12, true>) : f(rise_forward_narrow4<12, false>);
        return net->k5 ? f(rise_forward_narrow4<4, true>) : f(rise_forward_narrow4<4, false>);
    }
    if (net->nd.C == 64) return net->k5 ? f(rise_forward_narrow<2, true>) : f(rise_forward_narrow<2, false>);
    if (net->nd.C == 384) return net->k5 ? f(rise_forward_narrow<12, true>) : f(rise_forward_narrow<12, false>);
    return net->k5 ? f(rise_forward_narrow<4, true>) : f(rise_forward_narrow<4, false>);
}
int hm_net_can_serve(const hm_net* net) {
    if (!net || !net->narrow) return 0;
    // narrow_position's policy staging + prior scratch: inside the union region behind Ss, or else in Xs
    const int ldx = net->nd.C + 8, polOff = (65 * std::max(ldx, 88) + 7) & ~7, need = 160 * 64 + hmn::PRIOR_SCRATCH_HALFS;
    return (polOff + need <= net->uHalfs || need <= 65 * ldx) ? 1 : 0;
}
int hm_net_serve_is_slow(const hm_net* net) { return net && net->nd.C >= 384 ? 1 : 0; }
int hm_net_serve_info_get(const hm_net* net, hm_net_serve_info* out) {
    if (!hm_net_can_serve(net) || !out) return hm_fail(HM_ERR_INVALID, "this network cannot be the evaluator of the single-launch search");
    out->d_nd = net->d_nd; out->d_wh = net->d_wh; out->d_wf = net->d_wf;
    out->C = net->nd.C; out->k5 = net->k5 ? 1 : 0; out->copMax = net->copMax; out->uHalfs = net->uHalfs; out->ldsBytes = net->ldsNarrow;
    return 0;
}

template <typename F>
static hipError_t with_kernel(const hm_net* net, bool wide, F f) {
    using namespace hmn;
    switch (net->nd.C / 64) {
        case 1: return wide ? (net->k5 ? f(rise_forward_kernel<1, true, 1>) : f(rise_forward_kernel<1, false, 1>))
                            : (net->k5 ? f(rise_forward_kernel<1, true, 2>) : f(rise_forward_kernel<1, false, 2>));
        case 2: return wide ? (net->k5 ? f(rise_forward_kernel<2, true, 1>) : f(rise_forward_kernel<2, false, 1>))
                            : (net->k5 ? f(rise_forward_kernel<2, true, 2>) : f(rise_forward_kernel<2, false, 2>));
        case 4: return net->k5 ? f(rise_forward_kernel<4, true, 1>) : f(rise_forward_kernel<4, false, 1>);
        case 6: return net->k5 ? f(rise_forward_kernel<6, true, 1>) : f(rise_forward_kernel<6, false, 1>);
    }
    return hipErrorInvalidValue;
}

extern "C" {

// desc: hmn::NetDesc as a flat int32 array (see hivemind_amd/net.py FusedNet); d_wh / d_wf: packed fp16 / fp32 parameter
// buffers on the device, caller-owned, must outlive the handle.
int hm_net_create(const int32_t* desc, size_t desc_ints, const void* d_wh, const void* d_wf, hm_net** out) {
    using namespace hmn;
    if (!desc || !d_wh || !d_wf || !out) return hm_fail(HM_ERR_INVALID, "null argument");
    if (desc_ints * 4 != sizeof(NetDesc)) return hm_fail(HM_ERR_INVALID, "bad network descriptor size");
    hm_net* net = new hm_net();
    memcpy(&net->nd, desc, sizeof(NetDesc));
    const NetDesc& nd = net->nd;
    auto bad = [&](const char* m) { delete net; return hm_fail(HM_ERR_INVALID, m); };
    if (nd.C % 64 || nd.nblocks > MAXB || nd.nblocks < 0 || nd.cv > 32) return bad("unsupported network geometry");
    if (nd.C != 64 && nd.C != 128 && nd.C != 256 && nd.C != 384) return bad("trunk width must be 64, 128, 256 or 384");
    const size_t ldx = nd.C + 8;
    size_t lds = (2 * 65 * ldx + 64 * 66 + 64 * 72) * 2 + 2 * nd.C * 4;
    if (lds > 160 * 1024) return bad("network too wide for one LDS tile");
    size_t stage = 0;
    for (int i = 0; i < nd.nblocks; ++i) {
        if (nd.blk[i].k != 3 && nd.blk[i].k != 5) return bad("depthwise kernel must be 3 or 5");
        net->k5 |= nd.blk[i].k == 5;
        const size_t need = (size_t)nd.blk[i].cop * 8 + (size_t)nd.C * 4 + (size_t)nd.blk[i].cop * nd.blk[i].k * nd.blk[i].k * 2;
        if (need > stage) stage = need;
    }
    stage = (stage + 15) & ~(size_t)15;
    if (lds + stage <= 80 * 1024) lds += stage; else stage = 0;     // only while two workgroups still fit per CU
    net->lds = lds; net->stage = stage;
    net->d_wh = d_wh; net->d_wf = d_wf;
    if (const char* e = std::getenv("HM_NET_MAX_BLOCKS")) net->maxBlocks = std::atoi(e) > 0 ? std::atoi(e) : 1024;
    if (const char* e = std::getenv("HM_NET_WIDE_ROWS")) net->wideRows = std::atoi(e);
    if (hipMalloc(&net->d_nd, sizeof(NetDesc)) != hipSuccess) { delete net; return hm_fail(HM_ERR_NO_DEVICE, "hipMalloc failed"); }
    hipError_t e = hipMemcpy(net->d_nd, &net->nd, sizeof(NetDesc), hipMemcpyHostToDevice);
    for (int wide = 0; wide < 2 && e == hipSuccess; ++wide)
        e = with_kernel(net, wide != 0, [&](auto kern) { return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    if (e == hipSuccess && (nd.C == 64 || nd.C == 128 || nd.C == 384) && nd.cin_pad == 80 && !std::getenv("HM_NET_NO_NARROW")) {
        // the 8-wave kernel keeps `copMax` expanded channels of a block in LDS at a time; blocks wider than that (the deployed
        // 384-channel net: up to 1152) are walked in chunks of copMax
        int copMax = 32;
        for (int i = 0; i < nd.nblocks; ++i) copMax = std::max(copMax, nd.blk[i].cop);
        const size_t ldxN = nd.C + 8, ldi = nd.cin_pad + 8;
        auto bytes_for = [&](int cm) {
            size_t u = std::max((size_t)cm * 66 + (size_t)64 * (cm + 8), (size_t)65 * std::max(ldxN, ldi));
            u = (u + 7) & ~(size_t)7;
            return std::make_pair(u, (65 * ldxN + u) * 2 + (2 * (size_t)cm + nd.C) * 4 + ((((size_t)cm * 25 + 7) & ~(size_t)7) * 2) + (5 * (size_t)nd.C + 64) * 4);
        };
        if (const char* ce = std::getenv("HM_NET_COP_CHUNK")) copMax = std::max(64, std::atoi(ce) & ~63);
        while (copMax > 64 && bytes_for(copMax).second > 160 * 1024) copMax -= 64;      // chunk size: a multiple of 64 channels that fits
        const auto ub = bytes_for(copMax);
        if (ub.second <= 160 * 1024) {
            net->narrow = true; net->copMax = copMax; net->uHalfs = (int)ub.first; net->ldsNarrow = ub.second;
            if (const char* we = std::getenv("HM_NET_WAVES")) net->waves = std::atoi(we) == 4 ? 4 : 8;
            e = with_narrow(net, [&](auto kern) { return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ub.second); });
        }
    }
    if (e != hipSuccess) { (void)hipFree(net->d_nd); delete net; return hm_fail(HM_ERR_NO_DEVICE, std::string("hm_net_create: ") + hipGetErrorString(e)); }
    *out = net;
    return 0;
}
// As hm_net_create with the packed parameters in host memory: the handle uploads and owns device copies.
int hm_net_create_host(const int32_t* desc, size_t desc_ints, const void* h_wh, size_t wh_bytes, const void* h_wf, size_t wf_bytes, hm_net** out) {
    if (!h_wh || !h_wf || !wh_bytes || !wf_bytes || !out) return hm_fail(HM_ERR_INVALID, "null argument");
    void *dh = nullptr, *df = nullptr;
    if (hipMalloc(&dh, wh_bytes) != hipSuccess || hipMalloc(&df, wf_bytes) != hipSuccess) { if (dh) (void)hipFree(dh); return hm_fail(HM_ERR_NO_DEVICE, "hipMalloc failed"); }
    if (hipMemcpy(dh, h_wh, wh_bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(df, h_wf, wf_bytes, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(dh); (void)hipFree(df);
        return hm_fail(HM_ERR_NO_DEVICE, "parameter upload failed");
    }
    if (int rc = hm_net_create(desc, desc_ints, dh, df, out)) { (void)hipFree(dh); (void)hipFree(df); return rc; }
    (*out)->owned[0] = dh; (*out)->owned[1] = df;
    return 0;
}
int hm_net_destroy(hm_net* net) {
    if (!net) return 0;
    if (net->d_nd) (void)hipFree(net->d_nd);
    for (void* p : net->owned) if (p) (void)hipFree(p);
    delete net;
    return 0;
}

static int net_forward_impl(const hm_net* net, const void* d_planes, int n, const int32_t* d_group_rows, int group, void* d_value, void* d_pi_a,
                            void* d_pi_b, void* d_wdl, void* d_moves_left, void* stream, unsigned long long* d_dbg, unsigned long long* d_clk = nullptr) {
    if (!net || !d_planes || !d_value || !d_pi_a || !d_pi_b || !d_wdl || !d_moves_left) return hm_fail(HM_ERR_INVALID, "null argument");
    if (n <= 0) return 0;
    if (d_group_rows && group <= 0) return hm_fail(HM_ERR_INVALID, "group size must be positive");
    const int grid = n < net->maxBlocks ? n : net->maxBlocks;
    const bool wide = n <= net->wideRows;           // small batches: one workgroup per CU, full register file
    const hipError_t e = net->narrow
        ? with_narrow(net, [&](auto kern) {
              return launch_narrow(kern, net, grid, static_cast<hipStream_t>(stream), d_planes, n, d_value, d_pi_a, d_pi_b, d_wdl, d_moves_left, d_group_rows, group, d_dbg, d_clk);
          })
        : with_kernel(net, wide, [&](auto kern) {
              return launch_forward(kern, net, grid, static_cast<hipStream_t>(stream), d_planes, n, d_value, d_pi_a, d_pi_b, d_wdl, d_moves_left, d_group_rows, group, d_dbg, d_clk);
          });
    if (e != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, std::string("rise_forward_kernel: ") + hipGetErrorString(e));
    return 0;
}

int hm_net_forward(const hm_net* net, const void* d_planes, int n,
                   void* d_value, void* d_pi_a, void* d_pi_b, void* d_wdl, void* d_moves_left, void* stream) {
    return net_forward_impl(net, d_planes, n, nullptr, 0, d_value, d_pi_a, d_pi_b, d_wdl, d_moves_left, stream, nullptr);
}
// Ragged batches: rows are grouped (`group` consecutive rows per game slot) and only the first d_group_rows[g]
// rows of group g hold a position; the other rows are skipped and their head outputs are left untouched
// (the reference hands TensorRT exactly batchSize rows per search thread, searchthread.cc:474-484).
int hm_net_forward_groups(const hm_net* net, const void* d_planes, int n, const int32_t* d_group_rows, int group,
                          void* d_value, void* d_pi_a, void* d_pi_b, void* d_wdl, void* d_moves_left, void* stream) {
    return net_forward_impl(net, d_planes, n, d_group_rows, group, d_value, d_pi_a, d_pi_b, d_wdl, d_moves_left, stream, nullptr);
}
// As hm_net_forward_groups; additionally the launch min-/max-es its start / end (constant 100 MHz device clock) into
// d_interval[0] / d_interval[1] — the forward's slot of the search engine's leg clock (hm_sp_leg_clock_net).
int hm_net_forward_groups_timed(const hm_net* net, const void* d_planes, int n, const int32_t* d_group_rows, int group,
                                void* d_value, void* d_pi_a, void* d_pi_b, void* d_wdl, void* d_moves_left, void* stream, uint64_t* d_interval) {
    return net_forward_impl(net, d_planes, n, d_group_rows, group, d_value, d_pi_a, d_pi_b, d_wdl, d_moves_left, stream, nullptr,
                            reinterpret_cast<unsigned long long*>(d_interval));
}
// diagnostic build of the same launch: d_stamps[256] receives s_memtime at the phase boundaries of workgroup 0
int hm_net_profile(const hm_net* net, const void* d_planes, int n,
                   void* d_value, void* d_pi_a, void* d_pi_b, void* d_wdl, void* d_moves_left, void* stream, uint64_t* d_stamps) {
    return net_forward_impl(net, d_planes, n, nullptr, 0, d_value, d_pi_a, d_pi_b, d_wdl, d_moves_left, stream,
                            reinterpret_cast<unsigned long long*>(d_stamps));
}

}  // extern "C"
