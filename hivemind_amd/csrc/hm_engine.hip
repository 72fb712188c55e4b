// hm_engine.hip — the evaluator seam as an object: class Engine (nn/engine.h:43-81, nn/engine.cc:403-679) behind the C ABI.
//
// The reference's Engine owns a TensorRT plan, SearchParams::NUM_SEARCH_THREADS execution states (context, stream,
// device buffers, pinned host output buffers, a CUDA graph) and exposes an asynchronous enqueue / synchronize pair with
// one request in flight per worker.  Here the plan is a packed RISEv3 network (hm_net, one fused launch per batch), each
// worker has its own HIP stream + buffers, and the same contract holds: caller-owned fp16 observations that must stay
// untouched until the sync, engine-owned pinned outputs valid until the worker's next enqueue, bool-style failures
// (negative hm_status + hm_last_error) instead of exceptions.  Host code only.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/hivemind_amd.h"

int hm_fail(int code, const std::string& msg);

namespace {

struct Worker {          // Engine::ExecutionState (engine.h:83-103)
    hipStream_t stream = nullptr;
    void *dObs = nullptr, *dValue = nullptr, *dPolA = nullptr, *dPolB = nullptr, *dWdl = nullptr, *dMl = nullptr;
    uint16_t *hObs = nullptr, *hValue = nullptr, *hPolA = nullptr, *hPolB = nullptr, *hWdl = nullptr, *hMl = nullptr;
    bool pending = false;
};

uint16_t f32_to_f16(float f) {                      // round to nearest even (floatsToHalves, engine.cc:60-75 uses __float2half_rn)
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u | (m >> 13) : 0));
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int shift = 14 - e;
        uint32_t h = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (uint16_t)(sign | h);
}
float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31;
    uint32_t m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int s = 0; while (!(m & 0x400u)) { m <<= 1; ++s; } m &= 0x3ffu; x = sign | ((uint32_t)(113 - s) << 23) | (m << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

}  // namespace

struct hm_engine {
    int device = 0, batch = 8;
    hm_net* net = nullptr;
    Worker w[HM_ENGINE_WORKERS];
};

static int engine_free(hm_engine* e) {
    if (!e) return 0;
    for (Worker& w : e->w) {
        if (w.stream) { (void)hipStreamSynchronize(w.stream); (void)hipStreamDestroy(w.stream); }
        for (void* p : {w.dObs, w.dValue, w.dPolA, w.dPolB, w.dWdl, w.dMl}) if (p) (void)hipFree(p);
        for (void* p : {(void*)w.hObs, (void*)w.hValue, (void*)w.hPolA, (void*)w.hPolB, (void*)w.hWdl, (void*)w.hMl}) if (p) (void)hipHostFree(p);
    }
    if (e->net) hm_net_destroy(e->net);
    delete e;
    return 0;
}

extern "C" {

int hm_engine_create(int device, int batch_size, hm_engine** out) {   // Engine::Engine (engine.cc:256-270) + createExecutionResources (:403-535)
    if (!out || batch_size <= 0 || batch_size > 65536) return hm_fail(HM_ERR_INVALID, "bad hm_engine_create arguments");
    if (hipSetDevice(device) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipSetDevice failed");
    hm_engine* e = new hm_engine();
    e->device = device; e->batch = batch_size;
    const size_t B = (size_t)batch_size;
    for (Worker& w : e->w) {
        bool ok = hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipMalloc(&w.dObs, B * HM_PLANE_VALUES * 2) == hipSuccess && hipMalloc(&w.dValue, B * 2) == hipSuccess
            && hipMalloc(&w.dPolA, B * HM_POLICY_VALUES * 2) == hipSuccess && hipMalloc(&w.dPolB, B * HM_POLICY_VALUES * 2) == hipSuccess
            && hipMalloc(&w.dWdl, B * 3 * 2) == hipSuccess && hipMalloc(&w.dMl, B * 2) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&w.hObs, B * HM_PLANE_VALUES * 2, hipHostMallocDefault) == hipSuccess
            && hipHostMalloc((void**)&w.hValue, B * 2, hipHostMallocDefault) == hipSuccess
            && hipHostMalloc((void**)&w.hPolA, B * HM_POLICY_VALUES * 2, hipHostMallocDefault) == hipSuccess
            && hipHostMalloc((void**)&w.hPolB, B * HM_POLICY_VALUES * 2, hipHostMallocDefault) == hipSuccess
            && hipHostMalloc((void**)&w.hWdl, B * 3 * 2, hipHostMallocDefault) == hipSuccess
            && hipHostMalloc((void**)&w.hMl, B * 2, hipHostMallocDefault) == hipSuccess;
        if (!ok) { engine_free(e); return hm_fail(HM_ERR_NO_DEVICE, "engine buffer allocation failed"); }
    }
    *out = e;
    return 0;
}
int hm_engine_destroy(hm_engine* e) { return engine_free(e); }

int hm_engine_load_network(hm_engine* e, const int32_t* desc, size_t desc_ints, const void* h_wh, size_t wh_bytes, const void* h_wf, size_t wf_bytes) {
    if (!e) return hm_fail(HM_ERR_INVALID, "null argument");
    for (const Worker& w : e->w) if (w.pending) return hm_fail(HM_ERR_STATE, "cannot load a network while an inference is pending");
    if (hipSetDevice(e->device) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipSetDevice failed");
    hm_net* net = nullptr;
    if (int rc = hm_net_create_host(desc, desc_ints, h_wh, wh_bytes, h_wf, wf_bytes, &net)) return rc;
    if (e->net) hm_net_destroy(e->net);
    e->net = net;
    return 0;
}

int hm_net_save_file(const char* path, const int32_t* desc, size_t desc_ints, const void* h_wh, size_t wh_bytes, const void* h_wf, size_t wf_bytes) {
    if (!path || !desc || !h_wh || !h_wf) return hm_fail(HM_ERR_INVALID, "null argument");
    const std::string tmp = std::string(path) + ".tmp";
    std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
    if (!f) return hm_fail(HM_ERR_INVALID, "Unable to create " + tmp);
    const uint32_t version = 1;
    const uint64_t n[3] = {desc_ints, wh_bytes, wf_bytes};
    f.write("HMNP", 4);
    f.write(reinterpret_cast<const char*>(&version), 4);
    f.write(reinterpret_cast<const char*>(n), sizeof n);
    f.write(reinterpret_cast<const char*>(desc), (std::streamsize)(desc_ints * 4));
    f.write(static_cast<const char*>(h_wh), (std::streamsize)wh_bytes);
    f.write(static_cast<const char*>(h_wf), (std::streamsize)wf_bytes);
    f.close();
    if (!f) return hm_fail(HM_ERR_INVALID, "Failed to finalize " + tmp);
    if (std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); return hm_fail(HM_ERR_INVALID, std::string("Unable to publish ") + path); }
    return 0;
}
int hm_engine_load_network_file(hm_engine* e, const char* path) {   // Engine::loadNetwork (engine.cc:290-401): the cached-plan branch
    if (!e || !path) return hm_fail(HM_ERR_INVALID, "null argument");
    std::ifstream f(path, std::ios::binary);
    if (!f) return hm_fail(HM_ERR_INVALID, std::string("Unable to open ") + path);
    char magic[4];
    uint32_t version = 0;
    uint64_t n[3] = {0, 0, 0};
    f.read(magic, 4);
    f.read(reinterpret_cast<char*>(&version), 4);
    f.read(reinterpret_cast<char*>(n), sizeof n);
    if (!f || std::memcmp(magic, "HMNP", 4) != 0 || version != 1 || n[0] > (1u << 20) || n[1] > (1ull << 32) || n[2] > (1ull << 32))
        return hm_fail(HM_ERR_INVALID, std::string(path) + " is not a packed network file");
    std::vector<int32_t> desc(n[0]);
    std::vector<char> wh(n[1]), wf(n[2]);
    f.read(reinterpret_cast<char*>(desc.data()), (std::streamsize)(n[0] * 4));
    f.read(wh.data(), (std::streamsize)n[1]);
    f.read(wf.data(), (std::streamsize)n[2]);
    if (!f) return hm_fail(HM_ERR_INVALID, std::string(path) + " is truncated");
    return hm_engine_load_network(e, desc.data(), desc.size(), wh.data(), wh.size(), wf.data(), wf.size());
}

// Engine::enqueueInferenceHalf (engine.cc:577-650)
int hm_engine_enqueue_half(hm_engine* e, const void* obs, size_t worker) {
    if (!e || !obs || worker >= HM_ENGINE_WORKERS) return hm_fail(HM_ERR_INVALID, "bad hm_engine_enqueue_half arguments");
    if (!e->net) return hm_fail(HM_ERR_STATE, "no network loaded");
    if (hipSetDevice(e->device) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipSetDevice failed");
    Worker& w = e->w[worker];
    if (w.pending) return hm_fail(HM_ERR_STATE, "worker context " + std::to_string(worker) + " already has an inference pending");
    const size_t B = (size_t)e->batch;
    if (hipMemcpyAsync(w.dObs, obs, B * HM_PLANE_VALUES * 2, hipMemcpyDefault, w.stream) != hipSuccess)
        return hm_fail(HM_ERR_NO_DEVICE, "hipMemcpyAsync(input) failed");
    if (int rc = hm_net_forward(e->net, w.dObs, e->batch, w.dValue, w.dPolA, w.dPolB, w.dWdl, w.dMl, w.stream)) return rc;
    w.pending = true;
    const bool ok = hipMemcpyAsync(w.hValue, w.dValue, B * 2, hipMemcpyDeviceToHost, w.stream) == hipSuccess
        && hipMemcpyAsync(w.hPolA, w.dPolA, B * HM_POLICY_VALUES * 2, hipMemcpyDeviceToHost, w.stream) == hipSuccess
        && hipMemcpyAsync(w.hPolB, w.dPolB, B * HM_POLICY_VALUES * 2, hipMemcpyDeviceToHost, w.stream) == hipSuccess
        && hipMemcpyAsync(w.hWdl, w.dWdl, B * 3 * 2, hipMemcpyDeviceToHost, w.stream) == hipSuccess
        && hipMemcpyAsync(w.hMl, w.dMl, B * 2, hipMemcpyDeviceToHost, w.stream) == hipSuccess;
    if (!ok) {
        (void)hipStreamSynchronize(w.stream);
        w.pending = false;
        return hm_fail(HM_ERR_NO_DEVICE, "hipMemcpyAsync(outputs) failed");
    }
    return 0;
}
// Engine::synchronizeInferenceHalf (engine.cc:652-679)
int hm_engine_sync_half(hm_engine* e, hm_half_outputs* out, size_t worker) {
    if (out) *out = hm_half_outputs{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!e || !out || worker >= HM_ENGINE_WORKERS) return hm_fail(HM_ERR_INVALID, "bad hm_engine_sync_half arguments");
    if (hipSetDevice(e->device) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipSetDevice failed");
    Worker& w = e->w[worker];
    if (!w.pending) return hm_fail(HM_ERR_STATE, "worker context " + std::to_string(worker) + " has no inference pending");
    const hipError_t err = hipStreamSynchronize(w.stream);
    w.pending = false;
    if (err != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(err));
    out->value = w.hValue; out->policy_a = w.hPolA; out->policy_b = w.hPolB; out->wdl = w.hWdl; out->moves_left = w.hMl;
    return 0;
}
int hm_engine_run_half(hm_engine* e, const void* obs, hm_half_outputs* out, size_t worker) {   // runInferenceHalf (engine.cc:566-568, 681-686)
    if (out) *out = hm_half_outputs{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (int rc = hm_engine_enqueue_half(e, obs, worker)) return rc;
    return hm_engine_sync_half(e, out, worker);
}
// Engine::runInference (engine.cc:537-564): caller-owned f32 buffers, full batch
int hm_engine_run_f32(hm_engine* e, const float* obs, float* value, float* pi_a, float* pi_b, float* wdl, float* moves_left, size_t worker) {
    if (!e || !obs || !value || !pi_a || !pi_b || !wdl || !moves_left || worker >= HM_ENGINE_WORKERS) return hm_fail(HM_ERR_INVALID, "bad hm_engine_run_f32 arguments");
    Worker& w = e->w[worker];
    if (w.pending) return hm_fail(HM_ERR_STATE, "worker context " + std::to_string(worker) + " already has an inference pending");
    const size_t B = (size_t)e->batch;
    for (size_t i = 0; i < B * HM_PLANE_VALUES; ++i) w.hObs[i] = f32_to_f16(obs[i]);
    hm_half_outputs o;
    if (int rc = hm_engine_run_half(e, w.hObs, &o, worker)) return rc;
    for (size_t i = 0; i < B; ++i) { value[i] = f16_to_f32(o.value[i]); moves_left[i] = f16_to_f32(o.moves_left[i]); }
    for (size_t i = 0; i < B * HM_POLICY_VALUES; ++i) { pi_a[i] = f16_to_f32(o.policy_a[i]); pi_b[i] = f16_to_f32(o.policy_b[i]); }
    for (size_t i = 0; i < B * 3; ++i) wdl[i] = f16_to_f32(o.wdl[i]);
    return 0;
}
int hm_engine_batch_size(const hm_engine* e) { return e ? e->batch : 0; }
const hm_net* hm_engine_net(const hm_engine* e) { return e ? e->net : nullptr; }

}  // extern "C"
