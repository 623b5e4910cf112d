// hip_emu.cpp -- fiber scheduler of the test-only SIMT emulator (see hip_emu.h).
#include "hip_emu.h"

extern "C" void nsim_emu_ctx_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.globl nsim_emu_ctx_switch
.type nsim_emu_ctx_switch,@function
nsim_emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size nsim_emu_ctx_switch,.-nsim_emu_ctx_switch
)");

namespace emu {

static State g_state;
State& st() { return g_state; }

void yield() {
  State& s = g_state;
  Fiber* f = s.cur;
  nsim_emu_ctx_switch(&f->sp, s.sched_sp);
}

static void fiber_exit_bookkeeping(Fiber* f) {
  State& s = g_state;
  f->done = true;
  s.progress++;
  Wave& w = *f->wave;
  w.alive[f->lane] = false;
  w.nlive--;
  if (w.nlive > 0 && w.arrived >= w.nlive) {
    w.arrived = 0;
    w.gen++;
  }
  s.b_nlive--;
  if (s.b_nlive > 0 && s.b_arrived >= s.b_nlive) {
    s.b_arrived = 0;
    s.b_gen++;
  }
}

static void fiber_entry() {
  State& s = g_state;
  Fiber* f = s.cur;
  (*s.body)();
  fiber_exit_bookkeeping(f);
  nsim_emu_ctx_switch(&f->sp, s.sched_sp);
  fprintf(stderr, "emu: resumed a finished fiber\n");
  abort();
}

static constexpr size_t kStack = 256 * 1024;
static std::vector<Fiber> g_fibers;
static std::vector<Wave> g_waves;
static std::vector<char> g_dyn;

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  State& s = g_state;
  const int T = (int)(block.x * block.y * block.z);
  if (T % 64 != 0) {
    fprintf(stderr, "emu: block size %d is not a multiple of 64\n", T);
    abort();
  }
  if ((int)g_fibers.size() < T) {
    size_t old = g_fibers.size();
    g_fibers.resize(T);
    for (size_t i = old; i < g_fibers.size(); ++i) {
      void* p = nullptr;
      if (posix_memalign(&p, 64, kStack) != 0) abort();
      g_fibers[i].stack = (char*)p;
    }
  }
  if ((int)g_waves.size() < T / 64) g_waves.resize(T / 64);
  if (g_dyn.size() < shmem + 64) g_dyn.resize(shmem + 64);
  s.dyn_smem = (char*)(((uintptr_t)g_dyn.data() + 63) & ~(uintptr_t)63);
  s.bDim = block;
  s.gDim = grid;
  s.body = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        s.bIdx = dim3(bx, by, bz);
        s.b_nlive = T;
        s.b_arrived = 0;
        for (int wv = 0; wv < T / 64; ++wv) {
          Wave& w = g_waves[wv];
          w.nlive = 64;
          w.arrived = 0;
          for (int l = 0; l < 64; ++l) w.alive[l] = true;
        }
        for (int t = 0; t < T; ++t) {
          Fiber& f = g_fibers[t];
          f.done = false;
          f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          f.lane = t & 63;
          f.wave = &g_waves[t >> 6];
          uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
          void** sp = (void**)top;
          *(--sp) = nullptr;              // fake return address of fiber_entry
          *(--sp) = (void*)&fiber_entry;  // popped by `ret`
          for (int r = 0; r < 6; ++r) *(--sp) = nullptr;
          f.sp = (void*)sp;
        }
        int remaining = T;
        while (remaining > 0) {
          uint64_t before = s.progress;
          remaining = 0;
          for (int t = 0; t < T; ++t) {
            Fiber& f = g_fibers[t];
            if (f.done) continue;
            s.cur = &f;
            nsim_emu_ctx_switch(&s.sched_sp, f.sp);
            if (!f.done) remaining++;
          }
          if (remaining > 0 && s.progress == before) {
            fprintf(stderr,
                    "emu: deadlock (divergent collective or barrier) in block (%u,%u,%u)\n", bx,
                    by, bz);
            abort();
          }
        }
      }
  s.cur = nullptr;
}

}  // namespace emu
