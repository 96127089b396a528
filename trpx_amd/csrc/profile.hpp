// Optional per-stage timing (bench.py's roofline leg): when enabled through trpx_profile_enable,
// the launchers record a hipEvent on the launch stream before the first and after every kernel.
// Disabled (the default) it costs one branch per launch and nothing on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

namespace trpx {

struct Profiler {
    bool enabled = false;
    std::vector<hipEvent_t> pool;
    int used = 0;
    void begin() { used = 0; }
    void mark(hipStream_t st) {
        if (!enabled) return;
        if (used == (int)pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            pool.push_back(e);
        }
        (void)hipEventRecord(pool[used++], st);
    }
    // elapsed ms between consecutive marks of the last launch; returns the number of stages
    int read(float* ms, int cap) {
        if (used < 2) return 0;
        if (hipEventSynchronize(pool[used - 1]) != hipSuccess) return 0;
        int n = 0;
        for (int i = 0; i + 1 < used && n < cap; ++i, ++n)
            if (hipEventElapsedTime(&ms[n], pool[i], pool[i + 1]) != hipSuccess) ms[n] = -1.f;
        return n;
    }
};

Profiler& profiler();   // defined in api.hip (thread local)

}  // namespace trpx
