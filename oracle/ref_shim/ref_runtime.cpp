// ref_runtime.cpp -- CPU execution engine behind cuda_on_cpu.h: runs a CUDA grid block by block on one OS
// thread; the threads of a block are ucontext fibers resumed round-robin, each running until its next block
// barrier (or its end), which gives __syncthreads / __syncthreads_count / cg::thread_block::sync their CUDA
// meaning.  TEST INFRASTRUCTURE ONLY (oracle/_ref build).
#include "cuda_on_cpu.h"
#include <ucontext.h>
#include <sys/mman.h>

uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace r2ref {
namespace {
constexpr size_t STACK_BYTES = 128 * 1024;
struct Fiber { ucontext_t ctx; bool done; };
std::vector<Fiber> g_fibers;
char *g_stacks = nullptr;
size_t g_nstacks = 0;
ucontext_t g_main;
int g_cur = -1;
const std::function<void()> *g_body = nullptr;
int g_count_acc = 0, g_count_result = 0;

void trampoline()
{
    (*g_body)();
    g_fibers[g_cur].done = true;
    swapcontext(&g_fibers[g_cur].ctx, &g_main);
}
void set_thread(unsigned t)
{
    threadIdx.x = t % blockDim.x;
    threadIdx.y = (t / blockDim.x) % blockDim.y;
    threadIdx.z = t / (blockDim.x * blockDim.y);
}
void run_block(unsigned nthreads)
{
    if (g_nstacks < nthreads) {
        if (g_stacks) munmap(g_stacks, g_nstacks * STACK_BYTES);
        g_stacks = (char *)mmap(nullptr, (size_t)nthreads * STACK_BYTES, PROT_READ | PROT_WRITE,
                                MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == MAP_FAILED) { perror("mmap fiber stacks"); abort(); }
        g_nstacks = nthreads;
        g_fibers.resize(nthreads);
    }
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber &f = g_fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = g_stacks + (size_t)t * STACK_BYTES;
        f.ctx.uc_stack.ss_size = STACK_BYTES;
        f.ctx.uc_link = &g_main;
        makecontext(&f.ctx, trampoline, 0);
        f.done = false;
    }
    unsigned remaining = nthreads;
    g_count_acc = 0;
    while (remaining) {
        for (unsigned t = 0; t < nthreads; ++t) {
            Fiber &f = g_fibers[t];
            if (f.done) continue;
            set_thread(t);
            g_cur = (int)t;
            swapcontext(&g_main, &f.ctx);
            if (f.done) --remaining;
        }
        // every live fiber now sits at the same barrier: publish the vote, open the barrier
        g_count_result = g_count_acc;
        g_count_acc = 0;
    }
    g_cur = -1;
}
}  // namespace

void barrier()
{
    if (g_cur < 0) return;
    swapcontext(&g_fibers[g_cur].ctx, &g_main);
}
int barrier_count(int predicate)
{
    g_count_acc += predicate ? 1 : 0;
    barrier();
    return g_count_result;
}
void run_grid(dim3 grid, dim3 block, const std::function<void()> &thread_body)
{
    gridDim = grid;
    blockDim = block;
    g_body = &thread_body;
    const unsigned nthreads = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = uint3{ bx, by, bz };
                run_block(nthreads);
            }
    g_body = nullptr;
}
}  // namespace r2ref
