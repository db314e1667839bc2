// cub/cub.cuh -- host restatement of the two CUB device algorithms the reference calls
// (RAS/rasterizer_impl.cu:165,186,275,301): an inclusive prefix sum and a STABLE least-significant-digit
// radix sort of (key, value) pairs on key bits [begin_bit, end_bit).  Integer-exact by definition.
// oracle/_ref build only.
#pragma once
#include "cuda_on_cpu.h"
#include <numeric>
namespace cub {
struct DeviceScan {
    template <class In, class Out>
    static cudaError_t InclusiveSum(void *temp, size_t &temp_bytes, In in, Out out, int n)
    {
        if (!temp) { temp_bytes = 128; return cudaSuccess; }
        typename std::remove_reference<decltype(out[0])>::type acc = 0;
        for (int i = 0; i < n; ++i) { acc += in[i]; out[i] = acc; }
        return cudaSuccess;
    }
};
struct DeviceRadixSort {
    template <class K, class V>
    static cudaError_t SortPairs(void *temp, size_t &temp_bytes, const K *kin, K *kout, const V *vin, V *vout, int n,
                                 int begin_bit = 0, int end_bit = sizeof(K) * 8)
    {
        if (!temp) { temp_bytes = 128; return cudaSuccess; }
        const int nb = end_bit - begin_bit;
        const K mask = nb >= (int)sizeof(K) * 8 ? ~K(0) : (((K(1) << nb) - 1) << begin_bit);
        std::vector<int> idx(n);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return (kin[a] & mask) < (kin[b] & mask); });
        for (int i = 0; i < n; ++i) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
        return cudaSuccess;
    }
};
}  // namespace cub
