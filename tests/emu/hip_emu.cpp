// TEST INFRASTRUCTURE -- fiber scheduler behind tests/emu/hip/hip_runtime.h (see the header for scope).
#include <hip/hip_runtime.h>

namespace hipemu {

State g;

namespace {

constexpr size_t kStack = 128 * 1024;
constexpr int kMaxThreads = 1024;
constexpr int kWave = 64;

enum { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

// Context switch.  glibc's swapcontext saves / restores the signal mask with a system call on every switch, which was a
// third of the CPU tier's wall time; on x86-64 the fibers switch with a dozen instructions instead (callee-saved
// registers + stack pointer), elsewhere through ucontext.
#if defined(__x86_64__)
#define HIPEMU_ASM_SWITCH 1
struct Ctx { void* sp; };
extern "C" void hipemu_switch(Ctx* from, Ctx* to);
__asm__(
    ".text\n"
    ".globl hipemu_switch\n"
    ".type hipemu_switch,@function\n"
    "hipemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq (%rsi), %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size hipemu_switch,.-hipemu_switch\n");
#else
#define HIPEMU_ASM_SWITCH 0
struct Ctx { ucontext_t uc; };
inline void hipemu_switch(Ctx* from, Ctx* to) { swapcontext(&from->uc, &to->uc); }
#endif

struct Fiber {
  Ctx ctx;
  int state = DONE;
  dim3 tidx;
};

struct Block {
  std::vector<Fiber> fibers;
  char* stacks = nullptr;
  Ctx sched;
  int nthreads = 0, nwaves = 0, cur = -1;
  int alive = 0, arrived_block = 0;
  int wave_alive[kMaxThreads / kWave], wave_arrived[kMaxThreads / kWave], wave_parity[kMaxThreads / kWave];
  std::vector<Slot> slots;  // [wave][parity][64]
  const std::function<void()>* body = nullptr;
  std::vector<unsigned char> lds;
};

Block B;

void release_block() {
  B.arrived_block = 0;
  for (auto& f : B.fibers)
    if (f.state == WAIT_BLOCK) f.state = READY;
}
void release_wave(int w) {
  B.wave_arrived[w] = 0;
  for (int l = 0; l < kWave; ++l) {
    const int t = w * kWave + l;
    if (t < B.nthreads && B.fibers[t].state == WAIT_WAVE) B.fibers[t].state = READY;
  }
}

void yield_to_scheduler() { hipemu_switch(&B.fibers[B.cur].ctx, &B.sched); }

void trampoline() {
  (*B.body)();
  const int t = B.cur, w = t / kWave;
  B.fibers[t].state = DONE;
  --B.alive;
  --B.wave_alive[w];
  // a thread that exits may complete a rendezvous the others are waiting in
  if (B.alive > 0 && B.arrived_block == B.alive) release_block();
  if (B.wave_alive[w] > 0 && B.wave_arrived[w] == B.wave_alive[w]) release_wave(w);
  hipemu_switch(&B.fibers[t].ctx, &B.sched);
  abort();      // a finished fiber is never resumed
}

}  // namespace

int lane_id() { return B.cur % kWave; }

void block_sync() {
  Fiber& f = B.fibers[B.cur];
  if (++B.arrived_block == B.alive) {
    release_block();
    return;
  }
  f.state = WAIT_BLOCK;
  yield_to_scheduler();
}

Slot* wave_exchange(const void* mine, size_t bytes) {
  const int t = B.cur, w = t / kWave, l = t % kWave;
  const int parity = B.wave_parity[w];
  Slot* s = &B.slots[((size_t)w * 2 + parity) * kWave];
  memcpy(s[l].b, mine, bytes);
  // every lane flips its own view of the parity by counting: keep a per-fiber copy implicit in the order of
  // collectives -- all lanes of a wave execute the same sequence, so the shared counter is advanced by the
  // lane that completes the rendezvous.
  if (++B.wave_arrived[w] == B.wave_alive[w]) {
    B.wave_parity[w] ^= 1;
    release_wave(w);
    return s;
  }
  B.fibers[t].state = WAIT_WAVE;
  yield_to_scheduler();
  return s;
}

void launch(dim3 grid, dim3 block, size_t lds, const std::function<void()>& body) {
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads <= 0 || nthreads > kMaxThreads) { fprintf(stderr, "hipemu: bad block size %d\n", nthreads); abort(); }
  if ((int)B.fibers.size() < nthreads) {
    B.fibers.resize(kMaxThreads);
    B.stacks = (char*)malloc((size_t)kMaxThreads * kStack);
    B.slots.resize((size_t)(kMaxThreads / kWave) * 2 * kWave);
  }
  if (B.lds.size() < lds + 64) B.lds.resize(lds + 64);
  g.dyn_lds = (void*)(((uintptr_t)B.lds.data() + 63) & ~(uintptr_t)63);
  g.bdim = block;
  g.gdim = grid;
  B.body = &body;
  B.nthreads = nthreads;
  B.nwaves = (nthreads + kWave - 1) / kWave;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g.bidx = dim3(bx, by, bz);
        B.alive = nthreads;
        B.arrived_block = 0;
        for (int w = 0; w < B.nwaves; ++w) {
          const int n = nthreads - w * kWave;
          B.wave_alive[w] = n < kWave ? n : kWave;
          B.wave_arrived[w] = 0;
          B.wave_parity[w] = 0;
        }
        for (int t = 0; t < nthreads; ++t) {
          Fiber& f = B.fibers[t];
#if HIPEMU_ASM_SWITCH
          // fresh stack: six callee-saved register slots, then trampoline as the "return address" of the first switch
          // and a null return address above it, so that trampoline starts with the ABI's (rsp + 8) % 16 == 0
          void** top = reinterpret_cast<void**>(((uintptr_t)(B.stacks + (size_t)(t + 1) * kStack)) & ~(uintptr_t)15);
          top[-1] = nullptr;
          top[-2] = reinterpret_cast<void*>(&trampoline);
          for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
          f.ctx.sp = top - 8;
#else
          getcontext(&f.ctx.uc);
          f.ctx.uc.uc_stack.ss_sp = B.stacks + (size_t)t * kStack;
          f.ctx.uc.uc_stack.ss_size = kStack;
          f.ctx.uc.uc_link = nullptr;
          makecontext(&f.ctx.uc, trampoline, 0);
#endif
          f.state = READY;
          f.tidx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        }
        int guard = 0;
        while (B.alive > 0) {
          bool progressed = false;
          for (int t = 0; t < nthreads; ++t) {
            Fiber& f = B.fibers[t];
            if (f.state != READY) continue;
            progressed = true;
            B.cur = t;
            g.tidx = f.tidx;
            hipemu_switch(&B.sched, &f.ctx);
          }
          if (!progressed && ++guard > 2) { fprintf(stderr, "hipemu: deadlock (divergent barrier?)\n"); abort(); }
          if (progressed) guard = 0;
        }
      }
  B.body = nullptr;
}

}  // namespace hipemu
