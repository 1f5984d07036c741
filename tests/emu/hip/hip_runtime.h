// TEST INFRASTRUCTURE ONLY — a host-side stand-in for <hip/hip_runtime.h>.
//
// Compiling halo2-lib_amd/csrc/*.hip with `clang++ -x c++ -I tests/emu` runs the *same kernel source* on
// the CPU (one fiber per GPU thread, __syncthreads = yield-to-next-fiber, blocks spread over OS threads).
// It exists so that index math, carry handling and sort/scan/segment logic can be debugged in the
// GPU-less build container before GPU minutes are spent.  It is never built by __graft_entry__.build(),
// never loaded by the product (`halo2-lib_amd/`), and the library it produces (tests/emu/libh2hip_emu.so)
// is only opened explicitly by tests/test_emu_*.py.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define H2_HIPEMU 1
// H2_EMU_TSAN (tools/emu_tsan.sh): every emulated GPU thread is a ThreadSanitizer *fiber*, i.e. a logical thread of its own.  Switching between
// fibers creates no happens-before edge; __syncthreads (block-wide), the cross-lane operations and H2_WAVE_SYNC (both wave-wide) do, as do kernel
// boundaries.  Two lanes touching the same LDS or global word without such an edge between them — a data race on the GPU — then show up as a
// ThreadSanitizer report although the fibers of a block never run at the same time.
#ifdef H2_EMU_TSAN
extern "C" {
void *__tsan_get_current_fiber(void);
void *__tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void *fiber);
void __tsan_switch_to_fiber(void *fiber, unsigned flags);
void __tsan_acquire(void *addr);
void __tsan_release(void *addr);
}
#define H2_NO_TSAN __attribute__((no_sanitize("thread")))
#define H2_TSAN(...) __VA_ARGS__
#else
#define H2_NO_TSAN
#define H2_TSAN(...)
#endif
#define __global__
#define __device__
#define __host__
#define __constant__ static const
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };

// Fiber switches: glibc's swapcontext makes a rt_sigprocmask system call per switch — with a switch per lane and barrier that was 40 % of the CPU
// suite's time.  The plain build switches stacks itself (x86-64 SysV: the callee-saved registers and the stack pointer are the whole context of
// a function call); the sanitizer builds keep ucontext, which ThreadSanitizer / AddressSanitizer know how to follow.
#if !defined(H2_EMU_UCONTEXT) && (defined(H2_EMU_TSAN) || !defined(__x86_64__))
#define H2_EMU_UCONTEXT 1
#endif
#if !defined(H2_EMU_UCONTEXT) && defined(__has_feature)
#if __has_feature(address_sanitizer)
#define H2_EMU_UCONTEXT 1
#endif
#endif
#ifndef H2_EMU_UCONTEXT
// saves the caller's callee-saved registers on its stack, stores that stack pointer to *from_sp, continues on to_sp
__attribute__((naked, noinline)) static void hipemu_ctx_switch(void ** /*from_sp: rdi*/, void * /*to_sp: rsi*/) {
    __asm__ volatile(
        "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
        "movq %rsp, (%rdi)\n\tmovq %rsi, %rsp\n\t"
        "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\tret");
}
#endif

namespace hipemu {
inline thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
enum { FIBER_READY = 0, FIBER_AT_BARRIER = 1, FIBER_DONE = 2 };
struct Worker {
#ifdef H2_EMU_UCONTEXT
    std::vector<ucontext_t> ctx;
    ucontext_t main_ctx;
#else
    std::vector<void *> sp;     // a suspended fiber's stack pointer
    void *main_sp = nullptr;
#endif
    std::vector<char *> stacks;
    std::vector<char> state;
    int cur = 0;
    unsigned nthreads = 0;
    dim3 bdim;
    std::function<void()> body;
    std::vector<char> dyn_smem;
    std::vector<int> or_calls;
    int or_val[3] = {0, 0, 0};
    // cross-lane support (__shfl / __any): double-buffered slots + per-fiber call counters, wave-wide rendezvous
    std::vector<uint32_t> xl_slot[2];   // 16 words per lane
    std::vector<uint32_t> xl_calls;
#ifdef H2_EMU_TSAN
    std::vector<void *> tfiber;
    void *tmain = nullptr;
    std::vector<uint32_t> bar_calls;
    unsigned blocks_on_fibers = 0;
    char tag_block = 0, tag_done = 0, tag_barrier[2] = {0, 0}, tag_wave[64][2] = {};
#endif
    ~Worker() {
        for (char *p : stacks) free(p);
        H2_TSAN(for (void *f : tfiber) __tsan_destroy_fiber(f);)
    }
};
inline thread_local Worker *t_worker = nullptr;
#ifdef H2_EMU_UCONTEXT
#define H2_EMU_TO_MAIN(w, me) swapcontext(&(w)->ctx[me], &(w)->main_ctx)
#define H2_EMU_FROM_MAIN(w, i) swapcontext(&(w)->main_ctx, &(w)->ctx[i])
#define H2_EMU_FIBER_TO_FIBER(w, me, j) swapcontext(&(w)->ctx[me], &(w)->ctx[j])
#else
#define H2_EMU_TO_MAIN(w, me) hipemu_ctx_switch(&(w)->sp[me], (w)->main_sp)
#define H2_EMU_FROM_MAIN(w, i) hipemu_ctx_switch(&(w)->main_sp, (w)->sp[i])
#define H2_EMU_FIBER_TO_FIBER(w, me, j) hipemu_ctx_switch(&(w)->sp[me], (w)->sp[j])
#endif
H2_NO_TSAN inline void set_thread_idx(Worker *w, unsigned i) {
    w->cur = (int)i;
    t_threadIdx = dim3(i % w->bdim.x, (i / w->bdim.x) % w->bdim.y, i / (w->bdim.x * w->bdim.y));
}
H2_NO_TSAN inline void fiber_entry() {
    Worker *w = t_worker;
    H2_TSAN(__tsan_acquire(&w->tag_block);)
    w->body();
    w->state[w->cur] = FIBER_DONE;
    H2_TSAN(__tsan_release(&w->tag_done); __tsan_switch_to_fiber(w->tmain, 1);)
    H2_EMU_TO_MAIN(w, w->cur);
    abort();   // a finished fiber is never resumed
}
constexpr size_t kStack = 256 * 1024;
H2_NO_TSAN inline void run_block(Worker &w, unsigned nthreads, dim3 bdim) {
    if (w.stacks.size() < nthreads) {
        size_t old = w.stacks.size();
#ifdef H2_EMU_UCONTEXT
        w.ctx.resize(nthreads);
#else
        w.sp.resize(nthreads, nullptr);
#endif
        w.stacks.resize(nthreads, nullptr);
        for (size_t i = old; i < nthreads; ++i) w.stacks[i] = (char *)malloc(kStack);
    }
    w.nthreads = nthreads;
    w.bdim = bdim;
    w.state.assign(nthreads, FIBER_READY);
    for (unsigned i = 0; i < nthreads; ++i) {
#ifdef H2_EMU_UCONTEXT
        getcontext(&w.ctx[i]);
        w.ctx[i].uc_stack.ss_sp = w.stacks[i];
        w.ctx[i].uc_stack.ss_size = kStack;
        w.ctx[i].uc_link = &w.main_ctx;
        makecontext(&w.ctx[i], (void (*)())fiber_entry, 0);
#else
        // a fresh fiber looks like one suspended in hipemu_ctx_switch at the first instruction of fiber_entry: six register slots, the entry
        // address as the return address, and above it the (never used) return address of fiber_entry — rsp = 8 mod 16 at entry, as after a call
        void **top = (void **)(((uintptr_t)w.stacks[i] + kStack) & ~(uintptr_t)15);
        top[-1] = nullptr;
        top[-2] = (void *)&fiber_entry;
        for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
        w.sp[i] = (void *)(top - 8);
#endif
    }
    w.or_calls.assign(nthreads, 0);
    w.or_val[0] = w.or_val[1] = w.or_val[2] = 0;
    w.xl_slot[0].assign((size_t)nthreads * 16, 0);
    w.xl_slot[1].assign((size_t)nthreads * 16, 0);
    w.xl_calls.assign(nthreads, 0);
#ifdef H2_EMU_TSAN
    w.tmain = __tsan_get_current_fiber();
    if (w.blocks_on_fibers >= 1024) {   // a fiber's last switch never returns: its shadow stack keeps a frame or two per block
        for (void *f : w.tfiber) __tsan_destroy_fiber(f);
        w.tfiber.clear();
        w.blocks_on_fibers = 0;
    }
    while (w.tfiber.size() < nthreads) w.tfiber.push_back(__tsan_create_fiber(0));
    w.blocks_on_fibers++;
    w.bar_calls.assign(nthreads, 0);
    __tsan_release(&w.tag_block);
#endif
    for (;;) {
        bool ran = false, alive = false;
        for (unsigned i = 0; i < nthreads; ++i) {
            if (w.state[i] != FIBER_READY) continue;
            ran = true;
            set_thread_idx(&w, i);
            H2_TSAN(__tsan_switch_to_fiber(w.tfiber[i], 1);)   // 1 = no synchronisation: a switch orders nothing
            H2_EMU_FROM_MAIN(&w, i);
        }
        for (unsigned i = 0; i < nthreads; ++i) alive |= (w.state[i] != FIBER_DONE);
        if (!alive) break;
        if (!ran)   // everybody still alive is waiting at the barrier: release it
            for (unsigned i = 0; i < nthreads; ++i)
                if (w.state[i] == FIBER_AT_BARRIER) w.state[i] = FIBER_READY;
    }
    H2_TSAN(__tsan_acquire(&w.tag_done);)
}
inline void *dyn_smem_ptr() { return t_worker->dyn_smem.data(); }
// wave_only: the scheduling barrier is block-wide either way (the emulator has nothing finer), the happens-before edge is not — H2_WAVE_SYNC
// orders the lanes of ONE wave on the GPU, and the race check must not credit it with more
H2_NO_TSAN inline void syncthreads(bool wave_only = false) {
    Worker *w = t_worker;
#ifdef H2_EMU_TSAN
    const uint32_t k = w->bar_calls[w->cur]++;
    char *tag = wave_only ? &w->tag_wave[(unsigned)w->cur >> 6][k & 1] : &w->tag_barrier[k & 1];
    __tsan_release(tag);
#endif
    w->state[w->cur] = FIBER_AT_BARRIER;
    H2_TSAN(__tsan_switch_to_fiber(w->tmain, 1);)
    H2_EMU_TO_MAIN(w, w->cur);
    H2_TSAN(__tsan_acquire(tag);)
}
// run fiber j (a lane this fiber is waiting for) right now; returns when somebody resumes this fiber again
H2_NO_TSAN inline void switch_to(unsigned j) {
    Worker *w = t_worker;
    int me = w->cur;
    if (w->state[j] != FIBER_READY) abort();   // a lane reached a barrier / exited before the cross-lane op: non-uniform control flow
    set_thread_idx(w, j);
    H2_TSAN(__tsan_switch_to_fiber(w->tfiber[j], 1);)
    H2_EMU_FIBER_TO_FIBER(w, me, j);
}
// wave-wide rendezvous: post `v`, wait until every live lane of this wave has posted its call #k, return the slot array
H2_NO_TSAN inline const uint32_t *crosslane_exchange(const uint32_t *v, unsigned nwords, unsigned &wave_lo, unsigned &wave_hi) {
    Worker *w = t_worker;
    const unsigned me = (unsigned)w->cur;
    const uint32_t k = w->xl_calls[me];
    for (unsigned i = 0; i < nwords; ++i) w->xl_slot[k & 1][(size_t)me * 16 + i] = v[i];
    w->xl_calls[me] = k + 1;
    // a cross-lane operation is one instruction of the wave: what its lanes did before it is ordered before what they do after it.  (Its own
    // parity of tags: barrier-type and exchange-type edges of a wave never share one.)
    H2_TSAN(char *xtag = &w->tag_wave[32 + (me >> 6)][k & 1]; __tsan_release(xtag);)
    wave_lo = me & ~63u;
    wave_hi = wave_lo + 64 < w->nthreads ? wave_lo + 64 : w->nthreads;
    for (;;) {
        bool all = true;
        for (unsigned j = wave_lo; j < wave_hi; ++j) {
            if (w->state[j] == FIBER_DONE || w->xl_calls[j] > k) continue;
            all = false;
            switch_to(j);
            break;
        }
        if (all) break;
    }
    H2_TSAN(__tsan_acquire(xtag);)
    return w->xl_slot[k & 1].data();
}
}  // namespace hipemu

#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)hipemu::dyn_smem_ptr();
#define threadIdx (hipemu::t_threadIdx)
#define blockIdx (hipemu::t_blockIdx)
#define blockDim (hipemu::t_blockDim)
#define gridDim (hipemu::t_gridDim)
inline void __syncthreads() { hipemu::syncthreads(); }
// the emulated build's stand-in for a wave-level ordering point (H2_WAVE_SYNC): every lane has to get there, but only the lanes of one wave are ordered
inline void hipemu_wave_sync() { hipemu::syncthreads(true); }
H2_NO_TSAN inline unsigned __shfl(unsigned v, int src_lane, int /*width*/ = 64) {
    unsigned lo, hi;
    const uint32_t *slots = hipemu::crosslane_exchange(&v, 1, lo, hi);
    unsigned src = lo + ((unsigned)src_lane & 63u);
    return src < hi ? slots[(size_t)src * 16] : v;
}
// emulation-speed helper: shuffle up to 16 words with ONE rendezvous (the HIP build issues one __shfl per word)
template <int N>
H2_NO_TSAN inline void hipemu_shfl_words(uint32_t (&out)[N], const uint32_t (&in)[N], unsigned src_lane) {
    static_assert(N <= 16, "at most 16 words");
    unsigned lo, hi;
    const uint32_t *slots = hipemu::crosslane_exchange(in, N, lo, hi);
    unsigned src = lo + (src_lane & 63u);
    for (int i = 0; i < N; ++i) out[i] = src < hi ? slots[(size_t)src * 16 + i] : in[i];
}
inline int __shfl(int v, int src_lane, int width = 64) { return (int)__shfl((unsigned)v, src_lane, width); }
H2_NO_TSAN inline int __any(int pred) {
    unsigned lo, hi;
    uint32_t pv = pred ? 1u : 0u;
    const uint32_t *slots = hipemu::crosslane_exchange(&pv, 1, lo, hi);
    hipemu::Worker *w = hipemu::t_worker;
    int r = 0;
    for (unsigned j = lo; j < hi; ++j)
        if (w->state[j] != hipemu::FIBER_DONE) r |= (int)slots[(size_t)j * 16];
    return r;
}
H2_NO_TSAN inline int __syncthreads_or(int pred) {
    hipemu::Worker *w = hipemu::t_worker;
    int k = w->or_calls[w->cur]++;
    w->or_val[k % 3] |= (pred != 0);
    hipemu::syncthreads();
    int r = w->or_val[k % 3];
    w->or_val[(k + 2) % 3] = 0;
    return r;
}
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }

template <class T> inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicMax(T *p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline unsigned __brev(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 32; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline int __popc(unsigned x) { return __builtin_popcount(x); }

// ---------------------------------------------------------------- runtime API subset
typedef int hipError_t;
typedef struct hipemu_stream *hipStream_t;
struct hipemu_event { std::chrono::steady_clock::time_point t; };
typedef hipemu_event *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2 };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; char gcnArchName[64]; size_t totalGlobalMem; };
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof(*p));
    p->multiProcessorCount = 8;
    strcpy(p->name, "hipemu-cpu");
    strcpy(p->gcnArchName, "emu");
    p->totalGlobalMem = (size_t)8 << 30;
    return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
enum { hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
enum { hipHostRegisterDefault = 0 };
inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemu_event(); return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}

namespace hipemu {
// ONE worker state per OS thread, shared by every kernel (a function-local thread_local inside the launch template would be one per kernel
// instantiation: a hundred sets of fiber stacks per thread)
inline Worker &thread_worker() {
    static thread_local Worker w;
    return w;
}
// Persistent worker threads: a fresh std::thread per launch meant fresh fiber stacks (page faults, mmap / munmap) for every kernel launch.
struct Pool {
    std::mutex m, launch_m;
    std::condition_variable cv, done_cv;
    std::vector<std::thread> *threads = new std::vector<std::thread>();
    std::function<void()> job;
    uint64_t gen = 0;
    unsigned want = 0, active = 0;
    bool stop = false;
    pid_t owner = getpid();
    H2_NO_TSAN void loop(unsigned idx) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || (gen != seen && idx < want); });
            if (stop) return;
            seen = gen;
            std::function<void()> f = job;
            lk.unlock();
            f();
            lk.lock();
            if (--active == 0) done_cv.notify_all();
        }
    }
    void run(unsigned nworkers, const std::function<void()> &f) {
        std::lock_guard<std::mutex> one_launch(launch_m);
        std::unique_lock<std::mutex> lk(m);
        if (owner != getpid()) {   // forked: the parent's threads do not exist here
            threads = new std::vector<std::thread>();
            owner = getpid();
        }
        while (threads->size() < nworkers) {
            const unsigned idx = (unsigned)threads->size();
            threads->emplace_back([this, idx] { loop(idx); });
        }
        job = f;
        want = active = nworkers;
        ++gen;
        cv.notify_all();
        done_cv.wait(lk, [&] { return active == 0; });
        want = 0;
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        if (owner == getpid())
            for (auto &t : *threads) t.join();
    }
};
inline Pool &pool() {
    static Pool p;
    return p;
}
}  // namespace hipemu

template <class K, class... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t /*stream*/, A... args) {
    unsigned nblocks = grid.x * grid.y * grid.z, nthreads = block.x * block.y * block.z;
    if (!nblocks || !nthreads) return;
    unsigned nworkers = std::min<unsigned>(nblocks, std::max(1u, std::thread::hardware_concurrency()));
    H2_TSAN(nworkers = std::min(nworkers, 4u);)   // every fiber costs the sanitizer a few mappings: stay far below vm.max_map_count
    std::atomic<unsigned> next{0};
    auto work = [&]() {
        hipemu::Worker &worker = hipemu::thread_worker();
        hipemu::t_worker = &worker;
        hipemu::t_blockDim = block;
        hipemu::t_gridDim = grid;
        worker.body = [&]() { kernel(args...); };
        if (worker.dyn_smem.size() < shmem + 64) worker.dyn_smem.resize(shmem + 64);
        for (;;) {
            unsigned b = next.fetch_add(1);
            if (b >= nblocks) break;
            hipemu::t_blockIdx = dim3(b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y));
            hipemu::run_block(worker, nthreads, block);
        }
        worker.body = nullptr;
    };
    if (nworkers == 1) { work(); return; }
    hipemu::pool().run(nworkers, work);
}
