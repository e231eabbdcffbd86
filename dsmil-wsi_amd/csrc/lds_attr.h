// Raising a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) applies to the CURRENT device
// only, so the "already done" set is keyed by (device, kernel) and keeps the largest size asked for; guarded by a
// mutex (host threads driving different devices launch concurrently).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>

namespace dsmil_lds {

inline bool allow(const void* fn, int bytes) {
    struct Ent { int dev; const void* fn; int bytes; };
    static Ent seen[256];
    static int n = 0;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < n; ++i)
        if (seen[i].dev == dev && seen[i].fn == fn) {
            if (seen[i].bytes >= bytes) return true;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
            seen[i].bytes = bytes;
            return true;
        }
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (n < 256) seen[n++] = Ent{dev, fn, bytes};
    return true;
}

}  // namespace dsmil_lds
