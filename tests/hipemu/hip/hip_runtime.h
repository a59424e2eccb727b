// hipemu — a minimal HOST emulation of the HIP execution model, TEST INFRASTRUCTURE ONLY.
//
// Purpose: compile dorpatch_amd/csrc/dorpatch_hip.hip *unchanged* as plain C++ (host clang++,
// `-I tests/hipemu` so that <hip/hip_runtime.h> resolves here) into tests/hipemu/libdorpatch_emu.so,
// so that the index arithmetic, LDS tiling, wave64 shuffles and barrier structure of every kernel
// can be checked against the CPU oracle in the GPU-less build container (`pytest -m "not gpu"`).
// It is never loaded by the product (dorpatch_amd/_lib.py only ever opens libdorpatch_hip.so and
// dorpatch_amd.ops rejects non-GPU tensors); nothing here is shipped or measured.
//
// Model: blocks run one after another; the threads of a block are ucontext fibers scheduled
// cooperatively on the calling OS thread.  __syncthreads() and the wave64 shuffles are the only
// yield points.  A wave's shuffle completes when all of its not-yet-exited lanes have arrived
// (lanes of a wave execute shuffles in lockstep on the GPU too); __syncthreads() completes when
// all not-yet-exited threads of the block have arrived.  "Device pointers" are host pointers.
#pragma once

#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)   /* scheduling hint: no meaning on the host */
#define HIPEMU_HOST 1   /* lets the product TU drop gfx950 inline-asm register hints (DP_LAUNDER) */

using std::max;
using std::min;

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 {
  float x, y;
};

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorLaunchFailure = 719 };
typedef struct hipemuStream *hipStream_t;
typedef struct hipemuEvent *hipEvent_t;

namespace hipemu {

constexpr int kWave = 64;
constexpr size_t kStack = 64 * 1024;

enum State { RUNNABLE, AT_WAVE, AT_BLOCK, DONE };

struct Fiber {
  ucontext_t ctx;
  State state;
  dim3 tid;
  int flat;
  unsigned nshfl;  // shuffles executed so far (selects the exchange buffer)
};

struct Sched {
  ucontext_t main;
  std::vector<Fiber> fibers;
  std::vector<char> stacks;
  Fiber *cur = nullptr;
  dim3 block_idx, block_dim, grid_dim;
  const std::function<void()> *body = nullptr;
  std::vector<uint32_t> slots;  // [2][nthreads] shuffle exchange
  std::vector<unsigned> slot_seq;  // shuffle ordinal each slot was written at (0: never)
  std::vector<char> dyn_lds;
  hipError_t last_error = hipSuccess;
};

inline Sched &S() {
  static Sched s;
  return s;
}

inline void trampoline() {
  Sched &s = S();
  (*s.body)();
  s.cur->state = DONE;
  swapcontext(&s.cur->ctx, &s.main);
}

// Schedule fuzzing (tests/test_kernels_schedule.py): HIPEMU_ORDER=1 runs the runnable fibers of a block in DESCENDING
// thread order and the blocks of a grid in reverse order; 2 = a fixed pseudo-random permutation of both.  A kernel whose
// result depends on the order in which threads between two barriers (or blocks of a launch) execute — a missing
// __syncthreads() around an LDS tile, an inter-block dependency — then gives different results and fails its parity test.
inline int order_mode() {
  static const int mode = [] {
    const char *e = getenv("HIPEMU_ORDER");
    return e ? atoi(e) : 0;
  }();
  return mode;
}

inline unsigned permute(unsigned i, unsigned n) {  // position i of the schedule -> index (a bijection on [0, n))
  const int m = order_mode();
  if (m == 1) return n - 1 - i;
  if (m == 2) {  // multiplicative shuffle: stride coprime to n
    unsigned stride = 7919u % n;
    if (stride == 0) stride = 1;
    auto gcd = [](unsigned a, unsigned b) { while (b) { unsigned t = a % b; a = b; b = t; } return a; };
    while (gcd(stride, n) != 1) ++stride;
    return (unsigned)(((unsigned long long)i * stride + 3) % n);
  }
  return i;
}

inline void yield(State st) {
  Sched &s = S();
  s.cur->state = st;
  swapcontext(&s.cur->ctx, &s.main);
}

inline void run_block(const dim3 &bidx, const dim3 &bdim, const dim3 &gdim) {
  Sched &s = S();
  const int T = (int)(bdim.x * bdim.y * bdim.z);
  s.block_idx = bidx;
  s.block_dim = bdim;
  s.grid_dim = gdim;
  if ((int)s.fibers.size() < T) s.fibers.resize(T);
  if (s.stacks.size() < (size_t)T * kStack) s.stacks.resize((size_t)T * kStack);
  s.slots.assign((size_t)2 * T, 0u);
  s.slot_seq.assign((size_t)2 * T, 0u);
  for (int t = 0; t < T; ++t) {
    Fiber &f = s.fibers[t];
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = s.stacks.data() + (size_t)t * kStack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &s.main;
    makecontext(&f.ctx, trampoline, 0);
    f.state = RUNNABLE;
    f.flat = t;
    f.tid = dim3(t % bdim.x, (t / bdim.x) % bdim.y, t / (bdim.x * bdim.y));
    f.nshfl = 0;
  }
  int alive = T;
  while (alive > 0) {
    bool ran = false;
    for (int i = 0; i < T; ++i) {
      Fiber &f = s.fibers[permute((unsigned)i, (unsigned)T)];
      if (f.state != RUNNABLE) continue;
      s.cur = &f;
      swapcontext(&s.main, &f.ctx);
      ran = true;
      if (f.state == DONE) --alive;
    }
    // release waves whose live lanes have all arrived at a shuffle
    bool released = false;
    for (int w0 = 0; w0 < T; w0 += kWave) {
      int live = 0, at = 0;
      for (int t = w0; t < std::min(T, w0 + kWave); ++t) {
        if (s.fibers[t].state != DONE) ++live;
        if (s.fibers[t].state == AT_WAVE) ++at;
      }
      if (live > 0 && at == live) {
        for (int t = w0; t < std::min(T, w0 + kWave); ++t)
          if (s.fibers[t].state == AT_WAVE) s.fibers[t].state = RUNNABLE;
        released = true;
      }
    }
    if (!released && alive > 0) {
      int at = 0;
      for (int t = 0; t < T; ++t)
        if (s.fibers[t].state == AT_BLOCK) ++at;
      if (at == alive) {
        for (int t = 0; t < T; ++t)
          if (s.fibers[t].state == AT_BLOCK) s.fibers[t].state = RUNNABLE;
        released = true;
      }
    }
    if (!ran && !released && alive > 0) {
      fprintf(stderr, "hipemu: deadlock (divergent barrier / shuffle) in block (%u,%u,%u)\n", bidx.x, bidx.y,
              bidx.z);
      abort();
    }
  }
  s.cur = nullptr;
}

inline void launch(dim3 grid, dim3 block, size_t lds, const std::function<void()> &body) {
  Sched &s = S();
  if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 ||
      block.x * block.y * block.z > 1024 || grid.y > 65535 || grid.z > 65535 || lds > 160 * 1024) {
    s.last_error = hipErrorInvalidValue;  // what the runtime would report for an invalid configuration
    return;
  }
  s.body = &body;
  s.dyn_lds.assign(lds + 16, 0);
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x)
        run_block(dim3(permute(x, grid.x), permute(y, grid.y), permute(z, grid.z)), block, grid);
  s.body = nullptr;
}

inline void *dyn_lds_ptr() {
  uintptr_t p = reinterpret_cast<uintptr_t>(S().dyn_lds.data());
  return reinterpret_cast<void *>((p + 15) & ~(uintptr_t)15);
}

template <typename T>
inline T shuffle(T v, int src_lane_rel /* absolute lane within the wave, or -1: keep own */) {
  static_assert(sizeof(T) == 4, "32-bit shuffles only");
  Sched &s = S();
  Fiber &f = *s.cur;
  const int T_ = (int)(s.block_dim.x * s.block_dim.y * s.block_dim.z);
  const unsigned buf = f.nshfl & 1u;
  uint32_t bits;
  memcpy(&bits, &v, 4);
  s.slots[(size_t)buf * T_ + f.flat] = bits;
  const unsigned seq = ++f.nshfl;
  s.slot_seq[(size_t)buf * T_ + f.flat] = seq;
  yield(AT_WAVE);
  const int w0 = (f.flat / kWave) * kWave;
  int src = (src_lane_rel < 0 || src_lane_rel >= kWave) ? f.flat : w0 + src_lane_rel;
  // a source lane that exited before this shuffle never published a value (undefined on the GPU): own value.
  // (A lane that published and THEN ran to completion must still be readable, hence the ordinal check
  // instead of a liveness check.)
  if (src >= T_ || s.slot_seq[(size_t)buf * T_ + src] != seq) src = f.flat;
  bits = s.slots[(size_t)buf * T_ + src];
  T r;
  memcpy(&r, &bits, 4);
  return r;
}

}  // namespace hipemu

#define threadIdx (hipemu::S().cur->tid)
#define blockIdx (hipemu::S().block_idx)
#define blockDim (hipemu::S().block_dim)
#define gridDim (hipemu::S().grid_dim)

inline void __syncthreads() { hipemu::yield(hipemu::AT_BLOCK); }

template <typename T>
inline T __shfl_down(T v, unsigned off, int width = 64) {
  const int lane = hipemu::S().cur->flat % hipemu::kWave;
  const int src = lane + (int)off;
  return hipemu::shuffle(v, (src / width == lane / width) ? src : -1);
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  const int lane = hipemu::S().cur->flat % hipemu::kWave;
  const int src = lane ^ mask;
  return hipemu::shuffle(v, (src / width == lane / width) ? src : -1);
}
template <typename T>
inline T __shfl(T v, int src, int width = 64) {
  const int lane = hipemu::S().cur->flat % hipemu::kWave;
  return hipemu::shuffle(v, (lane / width) * width + (src % width));
}

// v_mfma_f32_16x16x4_f32 (one block): A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k, D[i][j] in lane j + 16 (i / 4),
// register i % 4; exact f32, an fmaf chain over k (MI355X_MICROARCH.md).  Emulated with 32 wave shuffles.
typedef float hipemu_f4 __attribute__((ext_vector_type(4)));
inline hipemu_f4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f4 c, int, int, int) {
  const int lane = hipemu::S().cur->flat % hipemu::kWave;
  const int j = lane & 15, ig = lane >> 4;
  hipemu_f4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * ig + r;
    float acc = d[r];
    for (int k = 0; k < 4; ++k) {
      const float av = hipemu::shuffle(a, i + 16 * k);
      const float bv = hipemu::shuffle(b, j + 16 * k);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  return d;
}

// v_mfma_f32_32x32x2_f32: A[i][k] in lane i + 32 k, B[k][j] in lane j + 32 k, D[i][j] in lane j + 32 ((i >> 2) & 1),
// register (i & 3) + 4 (i >> 3); exact f32, an fmaf chain over k.
typedef float hipemu_f16 __attribute__((ext_vector_type(16)));
inline hipemu_f16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipemu_f16 c, int, int, int) {
  const int lane = hipemu::S().cur->flat % hipemu::kWave;
  const int j = lane & 31, hi = lane >> 5;
  // gather the two operand matrices once (4 x 32 shuffles), then the lane's 16 results
  float av[2][32], bv[2];
  for (int k = 0; k < 2; ++k) {
    bv[k] = hipemu::shuffle(b, j + 32 * k);
    for (int i = 0; i < 32; ++i) av[k][i] = hipemu::shuffle(a, i + 32 * k);
  }
  hipemu_f16 d = c;
  for (int v = 0; v < 16; ++v) {
    const int i = (v & 3) + 8 * (v >> 2) + 4 * hi;
    d[v] = fmaf(av[1][i], bv[1], fmaf(av[0][i], bv[0], d[v]));
  }
  return d;
}

// v_mul_i32_i24: the product of the operands' low 24 bits (sign-extended); exact for the small values it is used on
inline int __mul24(int a, int b) { return ((a << 8) >> 8) * ((b << 8) >> 8); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }

// v_readlane_b32 with a wave-uniform lane index
inline int __builtin_amdgcn_readlane(int v, int lane) { return hipemu::shuffle(v, lane); }

// Raw buffer descriptor + 16-byte load with the hardware's range check: an offset at or past num_records reads zeros.
struct __amdgpu_buffer_rsrc_t {
  const char *base;
  unsigned bytes;
};
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void *p, short stride, int num_records, int flags) {
  (void)stride; (void)flags;
  return __amdgpu_buffer_rsrc_t{(const char *)p, (unsigned)num_records};
}
typedef unsigned hipemu_u4 __attribute__((ext_vector_type(4)));
inline hipemu_u4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voffset, int soffset, int aux) {
  (void)aux;
  hipemu_u4 v = {0u, 0u, 0u, 0u};
  const unsigned long o = (unsigned long)(unsigned)voffset;
  if (o + 16 <= r.bytes) memcpy(&v, r.base + o + (unsigned)soffset, 16);
  return v;
}

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...)                              \
  do {                                                                                         \
    (void)(stream);                                                                            \
    hipemu::launch((grid), (block), (lds), std::function<void()>([=]() { kernel(__VA_ARGS__); })); \
  } while (0)

#define hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ev0, ev1, flags, ...)          \
  do {                                                                                         \
    (void)(stream); (void)(ev0); (void)(ev1);                                                  \
    hipemu::launch((grid), (block), (lds), std::function<void()>([=]() { kernel(__VA_ARGS__); })); \
  } while (0)

inline hipError_t hipGetLastError() {
  const hipError_t e = hipemu::S().last_error;
  hipemu::S().last_error = hipSuccess;
  return e;
}
inline const char *hipGetErrorString(hipError_t e) {
  return e == hipSuccess ? "no error" : e == hipErrorInvalidValue ? "invalid argument" : "hipemu error";
}
inline hipError_t hipEventCreate(hipEvent_t *e) {
  *e = reinterpret_cast<hipEvent_t>(malloc(8));
  return hipSuccess;
}
inline hipError_t hipEventDestroy(hipEvent_t e) {
  free(e);
  return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
  *ms = 1.0f;  // the emulation has no clock: a fixed, non-zero placeholder (callers divide by it)
  return hipSuccess;
}
