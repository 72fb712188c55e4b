// Probe (MI355X, 8 XCDs): are agent-scope atomic adds ever LOST when other workgroups poll the same 128-byte line with
// agent-scope (sc1) loads?  Half of the workgroups add K times to one of NCOUNT counters (all in one or two lines), the other
// half poll "their" counter like hm_queue.hpp's wait_count does.  The host then compares every counter with the adds made.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__global__ void probe(unsigned* cnt, int ncount, int adders, int K, unsigned long long* polls, int mode) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (b < adders) {
        unsigned* c = cnt + (b % ncount);
        for (int i = 0; i < K; ++i) {
            __hip_atomic_fetch_add(c, 1u, RLX);
            for (int w = 0; w < 1 + (b & 7); ++w) __builtin_amdgcn_s_sleep(1);
        }
    } else {
        unsigned* c = cnt + (b % ncount);
        unsigned long long n = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned want = (unsigned)(((adders - (b % ncount) + ncount - 1) / ncount) * K);   // adds aimed at this counter
        for (;;) {
            unsigned v;
            if (mode == 0) v = __hip_atomic_load(c, RLX);
            else v = __hip_atomic_fetch_add(c, 0u, RLX);
            ++n;
            if (v >= want) break;
            __builtin_amdgcn_s_sleep(4);
            if ((n & 255) == 0 && (unsigned long long)__builtin_amdgcn_s_memrealtime() - t0 > 200000000ULL) { n |= 1ULL << 63; break; }   // 2 s
        }
        polls[b] = n;
    }
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, rounds = argc > 2 ? atoi(argv[2]) : 200;
    const int G = 256, adders = 128, K = 2000, NC = 16;
    unsigned* d; unsigned long long* p;
    hipMalloc(&d, 4096); hipMalloc(&p, G * 8);
    int lost = 0, timeouts = 0;
    for (int r = 0; r < rounds; ++r) {
        hipMemset(d, 0, 4096); hipMemset(p, 0, G * 8);
        hipLaunchKernelGGL(probe, dim3(G), dim3(64), 0, 0, d, NC, adders, K, p, mode);
        if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\n"); return 2; }
        std::vector<unsigned> h(NC); std::vector<unsigned long long> hp(G);
        hipMemcpy(h.data(), d, NC * 4, hipMemcpyDeviceToHost); hipMemcpy(hp.data(), p, G * 8, hipMemcpyDeviceToHost);
        for (int c = 0; c < NC; ++c) {
            const unsigned want = (unsigned)(((adders - c + NC - 1) / NC) * K);
            if (h[c] != want) { if (lost < 10) printf("round %d counter %d: %u of %u\n", r, c, h[c], want); ++lost; }
        }
        for (int b = adders; b < G; ++b) if (hp[b] >> 63) ++timeouts;
    }
    printf("mode %d rounds %d: counters with lost adds %d, poller timeouts %d\n", mode, rounds, lost, timeouts);
    return lost || timeouts ? 1 : 0;
}
