// hm_prof.hpp — diagnostic cycle accounting of game slot 0 (build with -DHM_SEARCH_PROF, read with
// hm_sp_profile / tools/profile_search.py).  Probes accumulate in LDS and are flushed once per launch.
#pragma once
#include <hip/hip_runtime.h>
#ifdef HM_SEARCH_PROF
constexpr int HM_NPROF = 64;                 // probe slots: sums in [0, NPROF), call counts in [NPROF, 2 NPROF)
__shared__ unsigned long long s_prof[2 * HM_NPROF];
#define PROF_T(t) const unsigned long long t = __builtin_amdgcn_s_memtime()
#define PROF_ADD(slot, t) do { if (blockIdx.x == 0 && threadIdx.x == 0) { s_prof[slot] += __builtin_amdgcn_s_memtime() - (t); s_prof[HM_NPROF + (slot)]++; } } while (0)
#define PROF_ADD_T(slot, t, tid) do { if (blockIdx.x == 0 && threadIdx.x == (tid)) { s_prof[slot] += __builtin_amdgcn_s_memtime() - (t); s_prof[HM_NPROF + (slot)]++; } } while (0)
#define PROF_INIT() do { if (threadIdx.x < 2 * HM_NPROF) s_prof[threadIdx.x] = 0; __syncthreads(); } while (0)
#define PROF_FLUSH() do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x < 2 * HM_NPROF) g_prof[threadIdx.x] += s_prof[threadIdx.x]; } while (0)
#else
#define PROF_T(t) do {} while (0)
#define PROF_ADD(slot, t) do {} while (0)
#define PROF_ADD_T(slot, t, tid) do {} while (0)
#define PROF_INIT() do {} while (0)
#define PROF_FLUSH() do {} while (0)
#endif

// Event log of collect_batch for one game slot (build with -DHM_SEARCH_TRACE; read with hm_sp_trace).  Same encoding
// as the oracle's Search::ctxTrace: (collect# << 32) | (code << 24) | (path length << 8) | outcome.
#ifdef HM_SEARCH_TRACE
#define TRACE_EV(code, len, outcome) do { if ((int)blockIdx.x == g_traceGame && (threadIdx.x & 63) == 0 && threadIdx.x < 64) { \
    const unsigned int k_ = g_traceCount++; if (k_ < 65536u) g_trace[k_] = ((unsigned long long)g_traceSeq << 32) | ((unsigned long long)(code) << 24) | ((unsigned long long)((len) & 0xffff) << 8) | (unsigned long long)((outcome) & 0xff); } } while (0)
#define TRACE_SEQ() do { if ((int)blockIdx.x == g_traceGame && threadIdx.x == 0) g_traceSeq++; } while (0)
#else
#define TRACE_EV(code, len, outcome) do {} while (0)
#define TRACE_SEQ() do {} while (0)
#endif
