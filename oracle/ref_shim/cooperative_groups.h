// cooperative_groups.h -- the two group handles the reference uses (this_grid().thread_rank(),
// this_thread_block().{sync,thread_rank,group_index,thread_index}) on top of cuda_on_cpu.h.  oracle/_ref build only.
#pragma once
#include "cuda_on_cpu.h"
namespace cooperative_groups {
struct grid_group {
    unsigned long long thread_rank() const
    {
        const unsigned long long b = ((unsigned long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const unsigned long long t = ((unsigned long long)threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x;
        return b * ((unsigned long long)blockDim.x * blockDim.y * blockDim.z) + t;
    }
};
struct thread_block {
    void sync() const { r2ref::barrier(); }
    unsigned int thread_rank() const { return (threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x; }
    dim3 group_index() const { return dim3(blockIdx.x, blockIdx.y, blockIdx.z); }
    dim3 thread_index() const { return dim3(threadIdx.x, threadIdx.y, threadIdx.z); }
    dim3 group_dim() const { return blockDim; }
};
inline grid_group this_grid() { return grid_group(); }
inline thread_block this_thread_block() { return thread_block(); }
}  // namespace cooperative_groups
