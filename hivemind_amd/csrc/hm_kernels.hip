// hm_kernels.hip — gfx950 kernels + C-ABI of the Bughouse rollout engine (board side).
// Reference interfaces replaced: see include/hivemind_amd.h (each entry point cites file:line).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <vector>

#include "hm_device.hpp"
#include "hm_host.hpp"

using namespace hmd;

// =====================================================================================
// plane encoder — board_to_planes (environment/planes.cc:213-265), batched.
//
// One wavefront per position (grid-stride).  The 74 planes all have the shape
// "value where mask bit set, else 0" (bitboard planes: value 1; scalar planes: mask = ~0), so
// a wave first derives 74 (mask, value) descriptors into LDS (lane p -> plane p), then streams
// the 4736 outputs as 16-byte stores, lane i -> chunk i, i+64, ...: every store instruction
// writes 1 KiB contiguous.  Algorithmic HBM traffic per position: 208 B read + 4736*sizeof(T)
// written (9472 B for f16).  Bound: HBM write bandwidth.
// =====================================================================================
__device__ PlaneConsts g_plane_consts;

template <int DT> struct ElemT;
template <> struct ElemT<HM_DT_F16> { typedef uint16_t type; static constexpr int per16 = 8; };
template <> struct ElemT<HM_DT_F32> { typedef uint32_t type; static constexpr int per16 = 4; };
template <> struct ElemT<HM_DT_U8>  { typedef uint8_t  type; static constexpr int per16 = 16; };

constexpr int PLANES_WAVES_PER_BLOCK = 4;

template <int DT>
__global__ __launch_bounds__(64 * PLANES_WAVES_PER_BLOCK) void encode_planes_kernel(
    const hm_board* __restrict__ boards, size_t n, void* __restrict__ out) {
    __shared__ u64 s_board[PLANES_WAVES_PER_BLOCK][26];
    __shared__ u64 s_mask[PLANES_WAVES_PER_BLOCK][HM_NB_PLANES + 6];
    __shared__ uint32_t s_val[PLANES_WAVES_PER_BLOCK][HM_NB_PLANES + 6];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave0 = (size_t)blockIdx.x * PLANES_WAVES_PER_BLOCK + wv;
    const size_t nwaves = (size_t)gridDim.x * PLANES_WAVES_PER_BLOCK;
    const uint32_t ONE = g_plane_consts.one[DT];
    for (size_t i = wave0; i < n; i += nwaves) {
        if (lane < 26) s_board[wv][lane] = reinterpret_cast<const u64*>(boards + i)[lane];
        __builtin_amdgcn_wave_barrier();
        const u64* bw = s_board[wv];
        const u64 tail = bw[25];                       // last_move[1] is in bw[24] hi; see below
        // hm_board layout in u64 words: pos0 = 0..11, pos1 = 12..23, word 24 = last_move[0] |
        // last_move[1]<<32, word 25 = rep_count[0], rep_count[1], team, time_adv, reserved
        const int team = (int)((tail >> 16) & 0xff);
        const int adv = (int)((tail >> 24) & 0xff);
        for (int p = lane; p < HM_NB_PLANES; p += 64) {
            const int b = p >= HM_NB_PLANES_PER_BOARD ? 1 : 0;
            const int j = p - b * HM_NB_PLANES_PER_BOARD;
            const u64* pw = bw + 12 * b;
            const bool flip = b == 0 ? team == 1 : team == 0;      // planes.cc:90-93
            const int first = b == 0 ? team : team ^ 1;            // planes.cc:99
            const u64 t1 = pw[11];
            const int castling = (int)((t1 >> 16) & 0xff), ep = (int)((t1 >> 24) & 0xff);
            const int stm = (int)((t1 >> 32) & 0xff), r50 = (int)((t1 >> 40) & 0xff);
            u64 mask = ~0ULL;
            uint32_t val = ONE;
            bool orient = false;
            if (j < 12) {                                           // pieces: own P..K, opp P..K
                const int c = j < 6 ? first : first ^ 1;
                mask = pw[j < 6 ? j : j - 6] & pw[6 + c];
                orient = true;
            } else if (j < 22) {                                    // pockets /16
                const int c = j < 17 ? first : first ^ 1;
                const int k = j < 17 ? j - 12 : j - 17;
                const int byteIdx = 80 + c * 5 + k;                 // hand[c][k] within hm_pos
                const int cnt = (int)((pw[byteIdx >> 3] >> (8 * (byteIdx & 7))) & 0xff);
                val = g_plane_consts.pocket[DT][cnt & 63];
            } else if (j < 24) {                                    // promoted own / opp
                const int c = j == 22 ? first : first ^ 1;
                mask = pw[8] & pw[6 + c];
                orient = true;
            } else if (j == 24) {                                   // en passant square
                mask = ep < 64 ? bit(ep) : 0;
                orient = true;
            } else if (j == 25) {                                   // on turn (planes.cc:153-159)
                val = stm == first ? ONE : 0;
            } else if (j == 26) {                                   // ones
            } else if (j < 31) {                                    // castling own-K, own-Q, opp-K, opp-Q
                const int c = j < 29 ? first : first ^ 1;
                const int right = ((j - 27) & 1) ? (c == 0 ? 2 : 8) : (c == 0 ? 1 : 4);
                val = (castling & right) ? ONE : 0;
            } else if (j == 31) {
                val = adv ? ONE : 0;
            } else if (j < 34) {                                    // last move from / to (planes.cc:182-198)
                const uint32_t lm = (uint32_t)(bw[24] >> (32 * b));
                mask = 0;
                if (lm != 0) {
                    const bool drop = (lm & (15u << 12)) == HM_MT_DROP;
                    int sq = j == 32 ? (int)((lm >> 6) & 63) : (int)(lm & 63);
                    if (flip) sq ^= 56;
                    if (!(j == 32 && drop)) mask = bit(sq);
                }
            } else if (j == 34) {
                val = g_plane_consts.r50[DT][r50 > 50 ? 50 : r50];
            } else {                                                // repetition >=2, >=3
                const int rc = (int)((tail >> (8 * b)) & 0xff);
                val = rc >= (j == 35 ? 2 : 3) ? ONE : 0;
            }
            if (orient && flip) mask = __builtin_bswap64(mask);     // flip_vertical (utils.h:111-113)
            s_mask[wv][p] = mask;
            s_val[wv][p] = val;
        }
        __builtin_amdgcn_wave_barrier();
        constexpr int PER = ElemT<DT>::per16;                       // squares per 16-byte chunk
        constexpr int CHUNKS = HM_PLANE_VALUES / PER;
        uint4* dst = reinterpret_cast<uint4*>(static_cast<char*>(out) + i * (size_t)HM_PLANE_VALUES * (16 / PER));
        for (int c = lane; c < CHUNKS; c += 64) {
            const int sq0 = c * PER;                                // first square index in the flat [74][64] array
            const int p = sq0 >> 6;
            const u64 m = s_mask[wv][p] >> (sq0 & 63);
            const uint32_t v = s_val[wv][p];
            uint4 o;
            if (DT == HM_DT_F16) {
                const uint32_t b = (uint32_t)m & 0xff;
                o.x = ((b & 1) ? v : 0) | ((b & 2) ? v << 16 : 0);
                o.y = ((b & 4) ? v : 0) | ((b & 8) ? v << 16 : 0);
                o.z = ((b & 16) ? v : 0) | ((b & 32) ? v << 16 : 0);
                o.w = ((b & 64) ? v : 0) | ((b & 128) ? v << 16 : 0);
            } else if (DT == HM_DT_F32) {
                const uint32_t b = (uint32_t)m & 0xf;
                o.x = (b & 1) ? v : 0; o.y = (b & 2) ? v : 0; o.z = (b & 4) ? v : 0; o.w = (b & 8) ? v : 0;
            } else {
                const uint32_t b = (uint32_t)m & 0xffff;
                uint32_t w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t nib = (b >> (4 * q)) & 0xf;
                    // spread 4 bits to 4 bytes of 0x01, then scale by v (v <= 255)
                    const uint32_t sp = (nib | (nib << 7) | (nib << 14) | (nib << 21)) & 0x01010101u;
                    w[q] = sp * v;
                }
                o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
            }
            dst[c] = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// =====================================================================================
// batched movegen / make — one position per lane, tables in LDS
// =====================================================================================
__device__ DeviceTables* g_tables_ptr;   // set by hm_init

__global__ __launch_bounds__(256) void legal_moves_kernel(const DeviceTables* __restrict__ tab,
                                                          const hm_pos* __restrict__ pos, size_t n,
                                                          u32* __restrict__ moves, u32* __restrict__ counts) {
    __shared__ AttackTab s_att;
    stage_table(&s_att, &tab->att);
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        P p;
        load_pos(p, pos + i);
        counts[i] = (u32)gen_legal_to(s_att, p, moves + i * HM_MAX_MOVES);
    }
}

// Test hook for gen_legal_wave (the wave-cooperative generator of the search kernels): one wave per position.
__global__ __launch_bounds__(64) void legal_moves_wave_kernel(const DeviceTables* __restrict__ tab,
                                                              const hm_pos* __restrict__ pos, size_t n,
                                                              u32* __restrict__ moves, u32* __restrict__ counts) {
    __shared__ AttackTab s_att;
    __shared__ u32 s_list[HM_MAX_MOVES];
    stage_table(&s_att, &tab->att);
    __syncthreads();
    for (size_t i = blockIdx.x; i < n; i += gridDim.x) {
        P p;
        load_pos(p, pos + i);
        const int cnt = gen_legal_wave(s_att, p, s_list);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 64) moves[i * HM_MAX_MOVES + k] = s_list[k];
        if (threadIdx.x == 0) counts[i] = (u32)cnt;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void count_moves_kernel(const DeviceTables* __restrict__ tab,
                                                          const hm_pos* __restrict__ pos, size_t n,
                                                          u32* __restrict__ counts) {
    __shared__ AttackTab s_att;
    stage_table(&s_att, &tab->att);
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        P p;
        load_pos(p, pos + i);
        counts[i] = (u32)count_legal(s_att, p);
    }
}

__global__ __launch_bounds__(256) void make_moves_kernel(const DeviceTables* __restrict__ tab,
                                                         const hm_board* __restrict__ in, const u32* __restrict__ ma,
                                                         const u32* __restrict__ mb, size_t n, hm_board* __restrict__ out) {
    __shared__ AttackTab s_att;
    __shared__ ZobristTab s_zob;
    stage_table(&s_att, &tab->att);
    stage_table(&s_zob, &tab->zob);
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        P A, B;
        load_pos(A, &in[i].pos[0]);
        load_pos(B, &in[i].pos[1]);
        const u32 a = ma[i], b = mb[i];
        make_joint(s_att, s_zob, A, B, a, b);
        store_pos(&out[i].pos[0], A);
        store_pos(&out[i].pos[1], B);
        out[i].last_move[0] = a ? a : in[i].last_move[0];
        out[i].last_move[1] = b ? b : in[i].last_move[1];
        out[i].rep_count[0] = in[i].rep_count[0];
        out[i].rep_count[1] = in[i].rep_count[1];
        out[i].team = in[i].team;
        out[i].time_adv = in[i].time_adv;
        out[i].reserved = 0;
    }
}

// =====================================================================================
// joint perft (tools/benchmark.cc:59-76)
//   level expansion: count kernel -> exclusive scan -> one thread per child
//   leaves: one thread per depth-(d-2) position, loops over its (a,b) children doing
//   copy-make + two bulk counts (the reference's depth-1 shortcut |A|*|B|).
// Joint positions travel as 2 x hm_pos (192 B).
// =====================================================================================
struct JointPos { hm_pos a, b; };

__global__ __launch_bounds__(256) void perft_count_children_kernel(const DeviceTables* __restrict__ tab,
                                                                   const JointPos* __restrict__ front, size_t n,
                                                                   u32* __restrict__ na, u32* __restrict__ nb,
                                                                   unsigned long long* __restrict__ nchild) {
    __shared__ AttackTab s_att;
    stage_table(&s_att, &tab->att);
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        P A, B;
        load_pos(A, &front[i].a);
        load_pos(B, &front[i].b);
        const u32 ca = (u32)count_legal(s_att, A), cb = (u32)count_legal(s_att, B);
        na[i] = ca; nb[i] = cb;
        nchild[i] = (unsigned long long)ca * cb;
    }
}

// move lists of every frontier position: lists[(i*2+board)*HM_MAX_MOVES + k]
__global__ __launch_bounds__(256) void perft_lists_kernel(const DeviceTables* __restrict__ tab,
                                                          const JointPos* __restrict__ front, size_t n,
                                                          u32* __restrict__ lists) {
    __shared__ AttackTab s_att;
    stage_table(&s_att, &tab->att);
    __syncthreads();
    // one thread per (position, board)
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < 2 * n; t += (size_t)gridDim.x * blockDim.x) {
        P p;
        load_pos(p, (t & 1) ? &front[t >> 1].b : &front[t >> 1].a);
        gen_legal_to(s_att, p, lists + t * HM_MAX_MOVES);
    }
}

__global__ __launch_bounds__(256) void perft_expand_kernel(const DeviceTables* __restrict__ tab,
                                                           const JointPos* __restrict__ front, size_t n,
                                                           const u32* __restrict__ nb, const unsigned long long* __restrict__ offs,
                                                           const u32* __restrict__ lists, unsigned long long total,
                                                           JointPos* __restrict__ out) {
    __shared__ AttackTab s_att;
    __shared__ ZobristTab s_zob;
    stage_table(&s_att, &tab->att);
    stage_table(&s_zob, &tab->zob);
    __syncthreads();
    for (unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; j < total;
         j += (unsigned long long)gridDim.x * blockDim.x) {
        // parent = last i with offs[i] <= j
        size_t lo = 0, hi = n;
        while (hi - lo > 1) {
            size_t mid = (lo + hi) >> 1;
            if (offs[mid] <= j) lo = mid; else hi = mid;
        }
        const unsigned long long r = j - offs[lo];
        const u32 cb = nb[lo];
        const u32 ia = (u32)(r / cb), ib = (u32)(r % cb);
        const u32 ma = lists[(lo * 2) * HM_MAX_MOVES + ia], mb = lists[(lo * 2 + 1) * HM_MAX_MOVES + ib];
        P A, B;
        load_pos(A, &front[lo].a);
        load_pos(B, &front[lo].b);
        make_joint(s_att, s_zob, A, B, ma, mb);
        store_pos(&out[j].a, A);
        store_pos(&out[j].b, B);
    }
}

// perft(2) below each frontier position; each resident thread owns 2*HM_MAX_MOVES words of a
// global scratch for its two move lists (only the first ~40 words of each are ever touched, so
// the live footprint stays cache-resident).
__global__ __launch_bounds__(256) void perft_leaf2_kernel(const DeviceTables* __restrict__ tab,
                                                          const JointPos* __restrict__ front, size_t n,
                                                          u32* __restrict__ scratch,
                                                          unsigned long long* __restrict__ partial) {
    __shared__ AttackTab s_att;
    __shared__ ZobristTab s_zob;
    stage_table(&s_att, &tab->att);
    stage_table(&s_zob, &tab->zob);
    __syncthreads();
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long sum = 0;
    // per-thread private lists (contiguous per thread; gen_legal compacts in place)
    u32* la = scratch + tid * (2 * HM_MAX_MOVES);
    u32* lb = la + HM_MAX_MOVES;
    for (size_t i = tid; i < n; i += nthreads) {
        P A, B;
        load_pos(A, &front[i].a);
        load_pos(B, &front[i].b);
        const int ca = gen_legal_to(s_att, A, la);
        const int cb = gen_legal_to(s_att, B, lb);
        for (int x = 0; x < ca; ++x) {
            P A1 = A;
            const u32 ma = la[x];
            const int ha = do_move(s_att, s_zob, A1, ma);
            for (int y = 0; y < cb; ++y) {
                P A2 = A1, B2 = B;
                if (ha) add_to_hand(s_zob, B2, ha);
                const int hb = do_move(s_att, s_zob, B2, lb[y]);
                if (hb) add_to_hand(s_zob, A2, hb);
                sum += (unsigned long long)count_legal(s_att, A2) * (unsigned long long)count_legal(s_att, B2);
            }
        }
    }
    // block reduce
    typedef hipcub::BlockReduce<unsigned long long, 256> BR;
    __shared__ typename BR::TempStorage tmp;
    unsigned long long tot = BR(tmp).Sum(sum);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// perft(1) of each frontier position: |A|*|B|
__global__ __launch_bounds__(256) void perft_leaf1_kernel(const DeviceTables* __restrict__ tab,
                                                          const JointPos* __restrict__ front, size_t n,
                                                          unsigned long long* __restrict__ partial) {
    __shared__ AttackTab s_att;
    stage_table(&s_att, &tab->att);
    __syncthreads();
    unsigned long long sum = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        P A, B;
        load_pos(A, &front[i].a);
        load_pos(B, &front[i].b);
        sum += (unsigned long long)count_legal(s_att, A) * (unsigned long long)count_legal(s_att, B);
    }
    typedef hipcub::BlockReduce<unsigned long long, 256> BR;
    __shared__ typename BR::TempStorage tmp;
    unsigned long long tot = BR(tmp).Sum(sum);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// =====================================================================================
// host side: C ABI
// =====================================================================================
namespace {

thread_local std::string t_err;
int fail(int code, const std::string& msg) { t_err = msg; return code; }

struct Lib {
    std::mutex mu;
    bool ready = false;
    int device = -1;
    DeviceTables* d_tab = nullptr;
    HostTables host;
} g;

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(HM_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int ensure_ready() {
    if (!g.ready) return fail(HM_ERR_STATE, "hm_init() has not been called (or failed)");
    return 0;
}

int grid_for(size_t n, int block, int maxBlocks = 2048) {
    size_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > (size_t)maxBlocks) b = maxBlocks;
    return (int)b;
}

}  // namespace

const HostTables& hm_host_tables() { return g.host; }
int hm_fail(int code, const std::string& msg) { return fail(code, msg); }

extern "C" {

void hm_search_config_default(hm_search_config* c) {   // search_params.h:26-273
    if (!c) return;
    c->cpuct_init = 2.5f; c->cpuct_base = 19652.0f; c->fpu_reduction = 1.0f; c->draw_contempt = 0.0f;
    c->wdl_value_weight = 0.25f; c->moves_left_discount = 0.005f;
    c->pw_coefficient = 2.0f; c->root_pw_coefficient = 4.0f; c->pw_exponent = 0.4f;
    c->enable_transpositions = 1; c->enable_dynamic_fpu = 1; c->enable_wdl_eval = 1;
}

int hm_abi_version(void) { return 1; }
const char* hm_last_error(void) { return t_err.c_str(); }

int hm_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}

int hm_init(int device) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.ready && g.device == device) return 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(HM_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= n) return fail(HM_ERR_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    build_host_tables(g.host);
    if (g.d_tab) { (void)hipFree(g.d_tab); g.d_tab = nullptr; }
    HIPCHK(hipMalloc(&g.d_tab, sizeof(DeviceTables)));
    HIPCHK(hipMemcpy(g.d_tab, &g.host.dev, sizeof(DeviceTables), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_plane_consts), &g.host.plane_consts, sizeof(PlaneConsts)));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_tables_ptr), &g.d_tab, sizeof(DeviceTables*)));
    HIPCHK(hipDeviceSynchronize());
    g.device = device;
    g.ready = true;
    return 0;
}

int hm_board_startpos(hm_board* out) {
    if (!out) return fail(HM_ERR_INVALID, "null out");
    HostTables& h = g.host;
    if (!h.built) build_host_tables(h);
    startpos(h, out);
    return 0;
}

int hm_policy_index(hm_move m, int stm) {
    HostTables& h = g.host;
    if (!h.built) build_host_tables(h);
    return host_policy_index(h, m, stm);
}

int hm_encode_planes(const hm_board* d_boards, size_t n, int dtype, void* d_out, void* stream) {
    if (int rc = ensure_ready()) return rc;
    if (n == 0) return 0;
    if (!d_boards || !d_out) return fail(HM_ERR_INVALID, "null device pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // enough waves to fill the chip several times over; grid-stride over positions
    size_t waves = n;
    int blocks = (int)std::min<size_t>((waves + PLANES_WAVES_PER_BLOCK - 1) / PLANES_WAVES_PER_BLOCK, 256 * 16);
    dim3 grid(blocks), block(64 * PLANES_WAVES_PER_BLOCK);
    switch (dtype) {
        case HM_DT_F16: hipLaunchKernelGGL(encode_planes_kernel<HM_DT_F16>, grid, block, 0, st, d_boards, n, d_out); break;
        case HM_DT_F32: hipLaunchKernelGGL(encode_planes_kernel<HM_DT_F32>, grid, block, 0, st, d_boards, n, d_out); break;
        case HM_DT_U8:  hipLaunchKernelGGL(encode_planes_kernel<HM_DT_U8>, grid, block, 0, st, d_boards, n, d_out); break;
        default: return fail(HM_ERR_INVALID, "unknown dtype");
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int hm_legal_moves(const hm_pos* d_pos, size_t n, hm_move* d_moves, uint32_t* d_counts, void* stream) {
    if (int rc = ensure_ready()) return rc;
    if (n == 0) return 0;
    if (!d_pos || !d_moves || !d_counts) return fail(HM_ERR_INVALID, "null device pointer");
    hipLaunchKernelGGL(legal_moves_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g.d_tab, d_pos, n, d_moves, d_counts);
    HIPCHK(hipGetLastError());
    return 0;
}

int hm_legal_moves_wave(const hm_pos* d_pos, size_t n, hm_move* d_moves, uint32_t* d_counts, void* stream) {
    if (int rc = ensure_ready()) return rc;
    if (n == 0) return 0;
    if (!d_pos || !d_moves || !d_counts) return fail(HM_ERR_INVALID, "null device pointer");
    hipLaunchKernelGGL(legal_moves_wave_kernel, dim3((unsigned)std::min<size_t>(n, 8192)), dim3(64), 0, static_cast<hipStream_t>(stream),
                       g.d_tab, d_pos, n, d_moves, d_counts);
    HIPCHK(hipGetLastError());
    return 0;
}

int hm_count_moves(const hm_pos* d_pos, size_t n, uint32_t* d_counts, void* stream) {
    if (int rc = ensure_ready()) return rc;
    if (n == 0) return 0;
    if (!d_pos || !d_counts) return fail(HM_ERR_INVALID, "null device pointer");
    hipLaunchKernelGGL(count_moves_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g.d_tab, d_pos, n, d_counts);
    HIPCHK(hipGetLastError());
    return 0;
}

int hm_make_moves(const hm_board* d_boards, const hm_move* d_move_a, const hm_move* d_move_b, size_t n,
                  hm_board* d_out, void* stream) {
    if (int rc = ensure_ready()) return rc;
    if (n == 0) return 0;
    if (!d_boards || !d_move_a || !d_move_b || !d_out) return fail(HM_ERR_INVALID, "null device pointer");
    hipLaunchKernelGGL(make_moves_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g.d_tab, d_boards, d_move_a, d_move_b, n, d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"

// ---- perft -------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <typename T> T* as() { return static_cast<T*>(p); }
};

static int sum_partials(unsigned long long* d_partial, int blocks, unsigned long long* out) {
    std::vector<unsigned long long> h(blocks);
    HIPCHK(hipMemcpy(h.data(), d_partial, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
    unsigned long long s = 0;
    for (auto v : h) s += v;
    *out = s;
    return 0;
}

// expands `front` (n joint positions) one ply; on return `next` holds the children.
static int expand_level(DevBuf& front, size_t n, DevBuf& next, size_t* nNext) {
    DevBuf na, nb, nchild, offs, lists, tmp;
    HIPCHK(na.alloc(n * 4)); HIPCHK(nb.alloc(n * 4));
    HIPCHK(nchild.alloc((n + 1) * 8)); HIPCHK(offs.alloc((n + 1) * 8));
    HIPCHK(hipMemset(nchild.p, 0, (n + 1) * 8));
    hipLaunchKernelGGL(perft_count_children_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, g.d_tab,
                       front.as<JointPos>(), n, na.as<u32>(), nb.as<u32>(), nchild.as<unsigned long long>());
    HIPCHK(hipGetLastError());
    size_t tmpBytes = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, nchild.as<unsigned long long>(), offs.as<unsigned long long>(), (int)(n + 1)));
    HIPCHK(tmp.alloc(tmpBytes));
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmpBytes, nchild.as<unsigned long long>(), offs.as<unsigned long long>(), (int)(n + 1)));
    unsigned long long total = 0;
    HIPCHK(hipMemcpy(&total, offs.as<unsigned long long>() + n, 8, hipMemcpyDeviceToHost));
    HIPCHK(lists.alloc(n * 2 * HM_MAX_MOVES * 4));
    hipLaunchKernelGGL(perft_lists_kernel, dim3(grid_for(2 * n, 256)), dim3(256), 0, 0, g.d_tab, front.as<JointPos>(), n, lists.as<u32>());
    HIPCHK(hipGetLastError());
    HIPCHK(next.alloc((size_t)total * sizeof(JointPos)));
    if (total) {
        hipLaunchKernelGGL(perft_expand_kernel, dim3(grid_for((size_t)total, 256, 8192)), dim3(256), 0, 0, g.d_tab,
                           front.as<JointPos>(), n, nb.as<u32>(), offs.as<unsigned long long>(), lists.as<u32>(), total,
                           next.as<JointPos>());
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipDeviceSynchronize());
    *nNext = (size_t)total;
    return 0;
}

extern "C" {

int hm_perft(const hm_board* root, int depth, int shard, int nshards, uint64_t* nodes, double* seconds) {
    if (int rc = ensure_ready()) return rc;
    if (!root || !nodes || depth < 0 || nshards < 1 || shard < 0 || shard >= nshards)
        return fail(HM_ERR_INVALID, "bad perft arguments");
    auto t0 = std::chrono::steady_clock::now();
    if (depth == 0) { *nodes = shard == 0 ? 1 : 0; if (seconds) *seconds = 0; return 0; }
    DevBuf front;
    size_t n = 1;
    HIPCHK(front.alloc(sizeof(JointPos)));
    JointPos r; r.a = root->pos[0]; r.b = root->pos[1];
    HIPCHK(hipMemcpy(front.p, &r, sizeof r, hipMemcpyHostToDevice));
    // expand down to the frontier the leaf kernel starts from: depth-2 plies above the leaves
    // (or depth-1 when depth < 3); the shard split is applied at ply 2 (or at the leaf frontier
    // when the tree is shallower).
    const int leafDepth = depth >= 3 ? 2 : 1;          // plies handled by the leaf kernel
    const int expandPlies = depth - leafDepth;
    bool sharded = false;
    for (int ply = 0; ply < expandPlies; ++ply) {
        DevBuf next;
        size_t nn = 0;
        if (int rc = expand_level(front, n, next, &nn)) return rc;
        std::swap(front.p, next.p);
        n = nn;
        if (!sharded && (ply + 1 == 2 || ply + 1 == expandPlies) && nshards > 1) {
            // contiguous stripe of the frontier for this shard
            size_t lo = n * (size_t)shard / nshards, hi = n * (size_t)(shard + 1) / nshards;
            DevBuf sub;
            HIPCHK(sub.alloc((hi - lo) * sizeof(JointPos)));
            if (hi > lo) HIPCHK(hipMemcpy(sub.p, front.as<JointPos>() + lo, (hi - lo) * sizeof(JointPos), hipMemcpyDeviceToDevice));
            std::swap(front.p, sub.p);
            n = hi - lo;
            sharded = true;
        }
    }
    if (!sharded && nshards > 1) {
        size_t lo = n * (size_t)shard / nshards, hi = n * (size_t)(shard + 1) / nshards;
        DevBuf sub;
        HIPCHK(sub.alloc((hi - lo) * sizeof(JointPos)));
        if (hi > lo) HIPCHK(hipMemcpy(sub.p, front.as<JointPos>() + lo, (hi - lo) * sizeof(JointPos), hipMemcpyDeviceToDevice));
        std::swap(front.p, sub.p);
        n = hi - lo;
    }
    unsigned long long total = 0;
    if (n) {
        const int blocks = grid_for(n, 256, 256 * 8);
        DevBuf partial;
        HIPCHK(partial.alloc(sizeof(unsigned long long) * blocks));
        if (leafDepth == 1) {
            hipLaunchKernelGGL(perft_leaf1_kernel, dim3(blocks), dim3(256), 0, 0, g.d_tab, front.as<JointPos>(), n,
                               partial.as<unsigned long long>());
        } else {
            DevBuf scratch;
            HIPCHK(scratch.alloc((size_t)blocks * 256 * 2 * HM_MAX_MOVES * 4));
            hipLaunchKernelGGL(perft_leaf2_kernel, dim3(blocks), dim3(256), 0, 0, g.d_tab, front.as<JointPos>(), n,
                               scratch.as<u32>(), partial.as<unsigned long long>());
            HIPCHK(hipGetLastError());
            HIPCHK(hipDeviceSynchronize());
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
        if (int rc = sum_partials(partial.as<unsigned long long>(), blocks, &total)) return rc;
    }
    *nodes = total;
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// Deterministic stand-in network for parity runs at full size (tests / tools only): FNV-1a of a row's 4736 fp16 words
// (xor salt) seeds a splitmix64 stream whose draws, quantised to 1e-3 and rounded to fp16, fill the five heads — the same
// function as the oracle's hash_evaluator_salted (oracle/search.hpp), so GPU self-play and the CPU restatement can be
// compared byte for byte without shipping planes to the host on every lockstep iteration.  One block per row.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long hash_draw(unsigned long long seed, unsigned long long call) {   // the (call+1)-th draw
    unsigned long long z = seed + (call + 1ULL) * 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint16_t f16_bits_rn(float f) { const _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
__global__ __launch_bounds__(256) void hash_evaluator_kernel(const uint16_t* __restrict__ planes, int rows, unsigned long long salt, uint16_t* value,
                                                             uint16_t* piA, uint16_t* piB, uint16_t* wdl, uint16_t* ml) {
    __shared__ unsigned long long s_seed;
    const int row = blockIdx.x;
    if (row >= rows) return;
    if (threadIdx.x == 0) {
        unsigned long long h = 0xcbf29ce484222325ULL ^ salt;
        const uint16_t* p = planes + (size_t)row * HM_PLANE_VALUES;
        for (int k = 0; k < HM_PLANE_VALUES; ++k) { h ^= p[k]; h *= 0x100000001b3ULL; }
        s_seed = h;
    }
    __syncthreads();
    const unsigned long long seed = s_seed;
    auto q = [&](unsigned long long call, int lo, int hi) { return (float)(lo + (int)(hash_draw(seed, call) % (unsigned long long)(hi - lo))) * 0.001f; };
    if (threadIdx.x == 0) {
        value[row] = f16_bits_rn(q(0, -900, 901));
        for (int k = 0; k < 3; ++k) wdl[(size_t)row * 3 + k] = f16_bits_rn(q(1 + k, -2000, 2001));
        ml[row] = f16_bits_rn(q(4, 0, 1001));
    }
    for (int k = threadIdx.x; k < HM_POLICY_VALUES; k += 256) {
        const unsigned long long r = hash_draw(seed, 5ULL + (unsigned long long)k);
        piA[(size_t)row * HM_POLICY_VALUES + k] = f16_bits_rn((float)((int)(r % 8001ULL) - 4000) * 0.001f);
        piB[(size_t)row * HM_POLICY_VALUES + k] = f16_bits_rn((float)((int)((r >> 32) % 8001ULL) - 4000) * 0.001f);
    }
}

extern "C" int hm_hash_evaluator(const void* d_planes, int rows, uint64_t salt, void* d_value, void* d_pi_a, void* d_pi_b, void* d_wdl,
                                 void* d_moves_left, void* stream) {
    if (!d_planes || !d_value || !d_pi_a || !d_pi_b || !d_wdl || !d_moves_left || rows < 0) return hm_fail(HM_ERR_INVALID, "null argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(hash_evaluator_kernel, dim3(rows), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const uint16_t*>(d_planes), rows,
                       (unsigned long long)salt, static_cast<uint16_t*>(d_value), static_cast<uint16_t*>(d_pi_a), static_cast<uint16_t*>(d_pi_b),
                       static_cast<uint16_t*>(d_wdl), static_cast<uint16_t*>(d_moves_left));
    if (hipGetLastError() != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hash_evaluator_kernel launch failed");
    return 0;
}
