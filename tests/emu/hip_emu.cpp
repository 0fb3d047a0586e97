// hip_emu.cpp -- TEST INFRASTRUCTURE ONLY: fiber scheduler behind hip_emu.h.
#include "hip_emu.h"

#include <mutex>

namespace emu {

thread_local BlockCtx* g_blk = nullptr;

static const size_t kStack = 96 * 1024;

static void trampoline() {
    BlockCtx* b = g_blk;
    (*b->body)();
    Fiber& f = b->fibers[b->cur];
    f.done = true;
    b->alive--;
    WaveCtx& w = b->waves[f.wave];
    w.alive--;
    if (b->alive > 0 && b->bar_count >= b->alive) {
        b->bar_count = 0;
        b->bar_gen++;
    }
    if (w.alive > 0 && w.count >= w.alive) {
        w.count = 0;
        w.gen++;
    }
    // returning follows uc_link back to the scheduler
}

static void run_block(BlockCtx& b, std::vector<char*>& stacks) {
    g_blk = &b;
    int n = b.nthreads;
    b.fibers.resize(n);
    b.waves.assign((n + 63) / 64, WaveCtx());
    for (auto& w : b.waves) { w.count = 0; w.gen = 0; w.alive = 0; }
    b.alive = n;
    b.bar_count = 0;
    b.bar_gen = 0;
    while ((int)stacks.size() < n) stacks.push_back((char*)malloc(kStack));
    for (int i = 0; i < n; ++i) {
        Fiber& f = b.fibers[i];
        f.flat = i;
        f.lane = i & 63;
        f.wave = i >> 6;
        f.done = false;
        f.tid.x = i % b.bdim.x;
        f.tid.y = (i / b.bdim.x) % b.bdim.y;
        f.tid.z = i / (b.bdim.x * b.bdim.y);
        f.stack = stacks[i];
        b.waves[f.wave].alive++;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &b.main_ctx;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    while (b.alive > 0) {
        for (int i = 0; i < n; ++i) {
            if (b.fibers[i].done) continue;
            b.cur = i;
            swapcontext(&b.main_ctx, &b.fibers[i].ctx);
        }
    }
    g_blk = nullptr;
}

void launch(Dim3 grid, Dim3 block, const std::function<void()>& body, size_t dyn_bytes) {
    long nblocks = (long)grid.x * grid.y * grid.z;
    if (nblocks <= 0) return;
    int nthreads = block.x * block.y * block.z;
    unsigned hw = std::thread::hardware_concurrency();
    int nworkers = (int)std::min<long>(nblocks, hw ? hw : 4);
    const char* env = getenv("AERO_EMU_THREADS");
    if (env) nworkers = std::max(1, std::min(nworkers, atoi(env)));
    std::atomic<long> next(0);
    auto worker = [&]() {
        std::vector<char*> stacks;
        char* dyn = dyn_bytes ? (char*)aligned_alloc(64, (dyn_bytes + 63) / 64 * 64) : nullptr;
        for (;;) {
            long i = next.fetch_add(1);
            if (i >= nblocks) break;
            BlockCtx b;
            b.gdim = grid;
            b.bdim = block;
            b.nthreads = nthreads;
            b.bid.x = i % grid.x;
            b.bid.y = (i / grid.x) % grid.y;
            b.bid.z = i / ((long)grid.x * grid.y);
            b.body = &body;
            b.dyn_smem = dyn;
            run_block(b, stacks);
        }
        for (char* s : stacks) free(s);
        free(dyn);
    };
    if (nworkers == 1) {
        worker();
    } else {
        std::vector<std::thread> ts;
        for (int t = 0; t < nworkers; ++t) ts.emplace_back(worker);
        for (auto& t : ts) t.join();
    }
}

}  // namespace emu
