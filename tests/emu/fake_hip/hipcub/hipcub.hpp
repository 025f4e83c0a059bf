// <hipcub/hipcub.hpp> for the CPU emulation of tests/emu — TEST INFRASTRUCTURE: the two primitives the builders use, serially.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>
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
struct DeviceRadixSort {
    // stable, on bits [begin_bit, end_bit) of the keys
    template <class K, class V> static hipError_t SortPairs(void* temp, size_t& temp_bytes, const K* keys_in, K* keys_out, const V* vals_in, V* vals_out, int n, int begin_bit = 0,
                                                            int end_bit = (int)sizeof(K) * 8, hipStream_t = nullptr)
    {
        if (!temp) { temp_bytes = 16; return hipSuccess; }
        const K mask = (K)((end_bit - begin_bit >= (int)sizeof(K) * 8 ? ~(K)0 : (((K)1 << (end_bit - begin_bit)) - 1)) << begin_bit);
        std::vector<int> idx((size_t)n);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return (keys_in[a] & mask) < (keys_in[b] & mask); });
        for (int i = 0; i < n; i++) { keys_out[i] = keys_in[idx[(size_t)i]]; vals_out[i] = vals_in[idx[(size_t)i]]; }
        return hipSuccess;
    }
};
} // namespace hipcub
