// wave_emu.cpp — cooperative-fiber runtime of the wave emulator (see hip/hip_runtime.h in this directory).
// TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <execinfo.h>
#include <vector>

// Minimal x86-64 System V context switch (callee-saved registers + stack pointer): ~20 ns instead of swapcontext's signal-mask
// system call; the kernels step through ~10^5 lane switches per simulation step.
extern "C" void wave_emu_switch(void** save_sp, void* load_sp);
__asm__(
    ".text\n.globl wave_emu_switch\n.type wave_emu_switch,@function\nwave_emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size wave_emu_switch,.-wave_emu_switch\n");
#if !defined(__x86_64__)
#error "tests/native/wave_emu: x86-64 only"
#endif

namespace wave_emu {

thread_local Ctx* cur = nullptr;

namespace {
constexpr size_t kStack = 256 * 1024;
constexpr int kMaxThreads = 1024;

struct Fiber {
    void* sp;
    Ctx ctx;
    bool done;
    int wave;
    uint32_t ops;          // cross-lane operations completed (parity selects the publish buffer)
    uint64_t pub[2];
    uint64_t wide[2][4];   // 32-byte payloads (two 8-half matrix operands), double-buffered like pub
    const char* wait_what;
    int wait_line;
};
struct Barrier {
    int live, arrived;
    uint64_t gen;
    const char* what;
    int line;
    int key;   // readlane: the (wave-uniform) source lane - tells two readlanes of one source line apart
};

Fiber* fibers = nullptr;
char* stacks = nullptr;
int n_fibers = 0;
Barrier waves[kMaxThreads / 64];
Barrier block;
void* sched_sp = nullptr;
Fiber* running = nullptr;
const std::function<void()>* body = nullptr;
uint64_t progress = 0;
uint64_t n_switches = 0, n_ops = 0;
unsigned long long ticks = 0;

void yield_to_scheduler() { ++n_switches; wave_emu_switch(&running->sp, sched_sp); }

[[noreturn]] void die_divergent(const Barrier& b, const char* what, int line, int key) {
    fprintf(stderr, "wave_emu: DIVERGENT cross-lane operation: some lanes wait at %s (line %d, source lane %d), lane %u arrived at %s (line %d, source lane %d)\n",
            b.what, b.line, b.key, running->ctx.tid.x, what, line, key);
    void* bt[32];
    const int n = backtrace(bt, 32);
    backtrace_symbols_fd(bt, n, 2);   // resolve with: addr2line -i -f -C -e tests/native/libraz_emu.so <offsets>
    abort();
}

void arrive(Barrier& b, const char* what, int line, int key = -1) {
    if (b.arrived == 0) {
        b.what = what;
        b.line = line;
        b.key = key;
    } else if (b.line != line || b.what != what || b.key != key) {
        die_divergent(b, what, line, key);
    }
    running->wait_what = what;
    running->wait_line = line;
    ++n_ops;
    ++b.arrived;
    const uint64_t g = b.gen;
    if (b.arrived >= b.live) {
        b.arrived = 0;
        ++b.gen;
        ++progress;
        return;
    }
    while (b.gen == g) yield_to_scheduler();
}

void leave(Barrier& b) {   // a lane finished: it no longer counts, and may complete a rendezvous the others wait at
    --b.live;
    if (b.live > 0 && b.arrived >= b.live) {
        b.arrived = 0;
        ++b.gen;
    }
}

void trampoline() {
    (*body)();
    running->done = true;
    ++progress;
    leave(waves[running->wave]);
    leave(block);
    yield_to_scheduler();
    abort();   // a finished fiber is never resumed
}
}  // namespace

uint64_t exchange(uint64_t v, int src, const char* what, int line) {
    Fiber* f = running;
    static const bool trace = getenv("WAVE_EMU_TRACE") != nullptr;
    if (trace && (f->ctx.tid.x == 0 || f->ctx.tid.x == 1)) fprintf(stderr, "T%u op%u %s line %d v=%llx src=%d\n", f->ctx.tid.x, f->ops, what, line, (unsigned long long)v, src);
    const int k = (int)(f->ops & 1u);
    f->pub[k] = v;
    ++f->ops;
    arrive(waves[f->wave], what, line, what[0] == 'r' ? src : -1);   // "readlane" / "readfirstlane": uniform source lane
    return fibers[f->wave * 64 + src].pub[k];
}

uint64_t ballot(bool pred, const char* what, int line) {
    Fiber* f = running;
    const int k = (int)(f->ops & 1u);
    f->pub[k] = pred ? 1u : 0u;
    ++f->ops;
    arrive(waves[f->wave], what, line);
    uint64_t m = 0;
    const int base = f->wave * 64;
    for (int i = 0; i < 64 && base + i < n_fibers; ++i)
        if (!fibers[base + i].done && fibers[base + i].pub[k]) m |= 1ULL << i;
    return m;
}

void gather(uint64_t v, uint64_t* out, const char* what, int line) {
    Fiber* f = running;
    const int k = (int)(f->ops & 1u);
    f->pub[k] = v;
    ++f->ops;
    arrive(waves[f->wave], what, line);
    const int base = f->wave * 64;
    for (int i = 0; i < 64; ++i) out[i] = base + i < n_fibers ? fibers[base + i].pub[k] : 0;
}

void gather32(const uint64_t v[4], uint64_t (*out)[4], const char* what, int line) {
    Fiber* f = running;
    const int k = (int)(f->ops & 1u);
    memcpy(f->wide[k], v, 32);
    ++f->ops;
    arrive(waves[f->wave], what, line);
    const int base = f->wave * 64;
    for (int i = 0; i < 64; ++i) {
        if (base + i < n_fibers) memcpy(out[i], fibers[base + i].wide[k], 32);
        else memset(out[i], 0, 32);
    }
}

void wave_barrier(const char* what, int line) { arrive(waves[running->wave], what, line); }
void block_barrier(int line) { arrive(block, "__syncthreads", line); }
unsigned long long clock64() { return ++ticks; }

void launch(dim3 grid, dim3 blk, const std::function<void()>& fn) {
    const int nt = (int)(blk.x * blk.y * blk.z);
    if (nt <= 0 || nt > kMaxThreads || blk.y != 1 || blk.z != 1) {
        fprintf(stderr, "wave_emu: unsupported block shape\n");
        abort();
    }
    if (running) {
        fprintf(stderr, "wave_emu: nested launch\n");
        abort();
    }
    if (!fibers) {
        fibers = new Fiber[kMaxThreads];
        stacks = (char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == (char*)MAP_FAILED) abort();
    }
    body = &fn;
    Ctx* const saved = cur;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                n_fibers = nt;
                for (int w = 0; w * 64 < nt; ++w) waves[w] = Barrier{std::min(64, nt - w * 64), 0, 0, nullptr, 0, -1};
                block = Barrier{nt, 0, 0, nullptr, 0, -1};
                for (int t = 0; t < nt; ++t) {
                    Fiber& f = fibers[t];
                    f.done = false;
                    f.wave = t / 64;
                    f.ops = 0;
                    f.pub[0] = f.pub[1] = 0;
                    f.wait_what = "start";
                    f.wait_line = 0;
                    f.ctx.tid = dim3((unsigned)t, 0, 0);
                    f.ctx.bid = dim3(bx, by, bz);
                    f.ctx.bdim = blk;
                    f.ctx.gdim = grid;
                    uintptr_t top = ((uintptr_t)(stacks + (size_t)(t + 1) * kStack)) & ~(uintptr_t)15;
                    void** frame = (void**)(top - 64);   // r15 r14 r13 r12 rbx rbp | return address (16-byte aligned slot) | pad
                    for (int i = 0; i < 6; ++i) frame[i] = nullptr;
                    frame[6] = (void*)trampoline;
                    frame[7] = nullptr;
                    f.sp = frame;
                }
                int left = nt;
                while (left > 0) {
                    const uint64_t before = progress;
                    left = 0;
                    for (int t = 0; t < nt; ++t) {
                        Fiber& f = fibers[t];
                        if (f.done) continue;
                        running = &f;
                        cur = &f.ctx;
                        wave_emu_switch(&sched_sp, f.sp);
                        if (!f.done) ++left;
                    }
                    if (left > 0 && progress == before) {   // a full round in which nobody got anywhere
                        fprintf(stderr, "wave_emu: DEADLOCK in block (%u,%u): lanes wait at different cross-lane operations:\n", bx, by);
                        for (int t = 0; t < nt; ++t)
                            if (!fibers[t].done) fprintf(stderr, "  thread %d: %s line %d\n", t, fibers[t].wait_what, fibers[t].wait_line);
                        abort();
                    }
                }
                running = nullptr;
            }
    cur = saved;
    body = nullptr;
}

}  // namespace wave_emu

extern "C" void wave_emu_counters(unsigned long long* out2) {
    out2[0] = wave_emu::n_switches;
    out2[1] = wave_emu::n_ops;
}
