// cuda_on_cpu.h -- executes CUDA C++ kernels on the host CPU, one OS thread, blocks sequential, the threads
// of a block as cooperatively scheduled fibers (ucontext) so that block barriers keep CUDA semantics.
// TEST INFRASTRUCTURE ONLY: lets oracle/Makefile compile the REFERENCE's own .cu files (from where they lie
// under /root/reference) with g++ into oracle/_ref/*.so, to pin the hand-written oracle restatement.
// Force-included (-include) in front of every reference translation unit.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <iostream>
#include <stdexcept>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static          /* blocks run one at a time on one OS thread: a function-local static IS block-shared */
#define __launch_bounds__(...)

struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct uint2 { unsigned int x, y; };
struct uint3 { unsigned int x, y, z; };
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct dim3 {
    unsigned int x, y, z;
    dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
    dim3(uint3 v) : x(v.x), y(v.y), z(v.z) {}
    operator uint3() const { return uint3{ x, y, z }; }
};

extern uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

// CUDA's integer/float min/max overload set (crt/math_functions.hpp)
inline int min(int a, int b) { return a < b ? a : b; }
inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
inline unsigned int min(int a, unsigned int b) { return min((unsigned int)a, b); }
inline unsigned int min(unsigned int a, int b) { return min(a, (unsigned int)b); }
inline float min(float a, float b) { return fminf(a, b); }
inline double min(double a, double b) { return fmin(a, b); }
inline double min(float a, double b) { return fmin((double)a, b); }
inline double min(double a, float b) { return fmin(a, (double)b); }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
inline unsigned int max(int a, unsigned int b) { return max((unsigned int)a, b); }
inline unsigned int max(unsigned int a, int b) { return max(a, (unsigned int)b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline double max(double a, double b) { return fmax(a, b); }
inline double max(float a, double b) { return fmax((double)a, b); }
inline double max(double a, float b) { return fmax(a, (double)b); }

namespace r2ref {
void barrier();                      // __syncthreads
int barrier_count(int predicate);    // __syncthreads_count
void run_grid(dim3 grid, dim3 block, const std::function<void()> &thread_body);
template <class... P, class... A>
void launch(dim3 grid, dim3 block, void (*kernel)(P...), A &&...args)
{
    run_grid(grid, block, [&]() { kernel(args...); });
}
}  // namespace r2ref

inline void __syncthreads() { r2ref::barrier(); }
inline int __syncthreads_count(int p) { return r2ref::barrier_count(p); }
inline void __trap() { abort(); }
// threads of a block are interleaved only at barriers, blocks are sequential: a plain read-modify-write is atomic
inline float atomicAdd(float *addr, float v) { float old = *addr; *addr = old + v; return old; }
inline int atomicAdd(int *addr, int v) { int old = *addr; *addr = old + v; return old; }
inline unsigned int atomicAdd(unsigned int *addr, unsigned int v) { unsigned int old = *addr; *addr = old + v; return old; }

// ---- the slice of the CUDA runtime API the reference's host code touches
typedef int cudaError_t;
#define cudaSuccess 0
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
inline cudaError_t cudaMemcpy(void *dst, const void *src, size_t n, cudaMemcpyKind) { memcpy(dst, src, n); return cudaSuccess; }
inline cudaError_t cudaMemset(void *dst, int v, size_t n) { memset(dst, v, n); return cudaSuccess; }
inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
inline const char *cudaGetErrorString(cudaError_t) { return "no error (CPU execution)"; }
