// Times the hand-written radix sort in isolation: hipcc ... sort_bench.hip ../../r2_gaussian_amd/csrc/radix_sort.hip
#include "../../r2_gaussian_amd/csrc/r2_common.hpp"
#include <vector>
#include <random>
#include <algorithm>
#include <stdarg.h>
namespace r2 { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); } int g_profile_mask_on = 0; void stage_begin(int, hipStream_t) {} void stage_end(int, hipStream_t) {} }
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main()
{
    struct Case { size_t n; int bits; const char *name; } cases[] = { {300000, 32, "depth P=300k 32b"}, {1156000, 11, "tiles R=1.16M 11b"}, {4843000, 16, "voxel R=4.8M 16b"}, {12000000, 13, "R=12M 13b"} };
    for (auto c : cases) {
        std::mt19937 rng(1);
        std::vector<uint32_t> k(c.n), v(c.n);
        for (size_t i = 0; i < c.n; ++i) { k[i] = c.bits == 32 ? (0x40000000u | (rng() & 0xFFFFFF)) : (rng() & ((1u << c.bits) - 1)); v[i] = (uint32_t)i; }
        uint32_t *dk, *dv, *ok, *ov; char *tmp;
        size_t tb = r2::sort_temp_bytes(c.n);
        CHECK(hipMalloc(&dk, c.n * 4)); CHECK(hipMalloc(&dv, c.n * 4)); CHECK(hipMalloc(&ok, c.n * 4)); CHECK(hipMalloc(&ov, c.n * 4)); CHECK(hipMalloc(&tmp, tb));
        CHECK(hipMemcpy(dk, k.data(), c.n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dv, v.data(), c.n * 4, hipMemcpyHostToDevice));
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        for (int i = 0; i < 3; ++i) r2::sort_pairs_u32_u32(tmp, tb, dk, ok, dv, ov, c.n, c.bits, 0);
        CHECK(hipEventRecord(a));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) r2::sort_pairs_u32_u32(tmp, tb, dk, ok, dv, ov, c.n, c.bits, 0);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        std::vector<uint32_t> rk(c.n), rv(c.n);
        CHECK(hipMemcpy(rk.data(), ok, c.n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(rv.data(), ov, c.n * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> idx(c.n); for (size_t i = 0; i < c.n; ++i) idx[i] = (uint32_t)i;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return k[x] < k[y]; });
        bool good = true; for (size_t i = 0; i < c.n && good; ++i) good = rv[i] == idx[i] && rk[i] == k[idx[i]];
        printf("%-22s %8.1f us/sort  %s\n", c.name, ms / reps * 1e3, good ? "OK" : "WRONG");
        hipFree(dk); hipFree(dv); hipFree(ok); hipFree(ov); hipFree(tmp);
    }
    return 0;
}
