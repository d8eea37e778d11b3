// block_cache.cpp — process-wide reuse of device / page-locked blocks and of streams across handles (engine.hpp).
#include <map>
#include <mutex>
#include <vector>

#include "engine.hpp"

namespace cba {
namespace {

constexpr size_t kMaxCachedBlock = size_t(16) << 20;   // larger blocks go straight back to the runtime
constexpr size_t kMaxCachedTotal = size_t(256) << 20;  // per kind and device

struct Pool {
    std::multimap<size_t, void*> free_blocks;  // size class -> block
    size_t bytes = 0;
};

std::mutex g_mu;
std::map<std::pair<int, int>, Pool> g_pools;  // (pinned, device)
std::map<int, std::vector<hipStream_t>> g_streams;

size_t size_class(size_t bytes) {
    size_t c = 512;
    while (c < bytes) c <<= 1;
    return c;
}

void raw_free(bool pinned, void* p) noexcept {
    if (pinned) (void)hipHostFree(p);
    else (void)hipFree(p);
}

}  // namespace

void* cache_alloc(bool pinned, size_t bytes, size_t* granted) {
    int device = 0;
    CBA_HIP(hipGetDevice(&device));
    const bool cacheable = bytes <= kMaxCachedBlock;
    const size_t want = cacheable ? size_class(bytes) : bytes;
    if (cacheable) {
        std::lock_guard<std::mutex> lock(g_mu);
        Pool& pool = g_pools[{pinned ? 1 : 0, device}];
        auto it = pool.free_blocks.find(want);
        if (it != pool.free_blocks.end()) {
            void* p = it->second;
            pool.free_blocks.erase(it);
            pool.bytes -= want;
            *granted = want;
            return p;
        }
    }
    void* p = nullptr;
    hipError_t err = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
    if (err != hipSuccess) {  // out of memory: give the cached blocks back and try once more
        (void)hipGetLastError();
        cache_trim();
        err = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
    }
    if (err != hipSuccess) throw HipError(std::string(pinned ? "hipHostMalloc: " : "hipMalloc: ") + hipGetErrorString(err));
    *granted = want;
    return p;
}

void cache_release(bool pinned, int device, void* p, size_t granted) noexcept {
    if (!p) return;
    if (granted <= kMaxCachedBlock && granted == size_class(granted)) {
        std::lock_guard<std::mutex> lock(g_mu);
        Pool& pool = g_pools[{pinned ? 1 : 0, device}];
        if (pool.bytes + granted <= kMaxCachedTotal) {
            try {
                pool.free_blocks.emplace(granted, p);
                pool.bytes += granted;
                return;
            } catch (...) {
            }
        }
    }
    raw_free(pinned, p);
}

hipStream_t cache_stream() {
    int device = 0;
    CBA_HIP(hipGetDevice(&device));
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto& v = g_streams[device];
        if (!v.empty()) {
            hipStream_t s = v.back();
            v.pop_back();
            return s;
        }
    }
    hipStream_t s = nullptr;
    CBA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}

void cache_stream_release(int device, hipStream_t s) noexcept {
    if (!s) return;
    std::lock_guard<std::mutex> lock(g_mu);
    try {
        auto& v = g_streams[device];
        if (v.size() < 16) { v.push_back(s); return; }
    } catch (...) {
    }
    (void)hipStreamDestroy(s);
}

void cache_trim() {
    std::map<std::pair<int, int>, Pool> pools;
    std::map<int, std::vector<hipStream_t>> streams;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        pools.swap(g_pools);
        streams.swap(g_streams);
    }
    int current = 0;
    (void)hipGetDevice(&current);
    for (auto& kv : pools) {
        (void)hipSetDevice(kv.first.second);
        for (auto& blk : kv.second.free_blocks) raw_free(kv.first.first != 0, blk.second);
    }
    for (auto& kv : streams) {
        (void)hipSetDevice(kv.first);
        for (hipStream_t s : kv.second) (void)hipStreamDestroy(s);
    }
    (void)hipSetDevice(current);
}

}  // namespace cba
