// <hipcub/hipcub.hpp> for the CPU emulation of tests/emu — TEST INFRASTRUCTURE: the one primitive the builders use, serially.
#pragma once
#include <hip/hip_runtime.h>
namespace hipcub {
struct DeviceScan {
    template <class In, class Out> static hipError_t ExclusiveSum(void* temp, size_t& temp_bytes, In in, Out out, int n, hipStream_t = nullptr)
    {
        if (!temp) { temp_bytes = 16; return hipSuccess; }
        unsigned long long run = 0;
        for (int i = 0; i < n; i++) { const auto v = in[i]; out[i] = (decltype(v))run; run += v; }
        return hipSuccess;
    }
};
} // namespace hipcub
