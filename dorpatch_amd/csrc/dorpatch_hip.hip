// dorpatch_hip.hip — hand-written gfx950 (MI355X, CDNA4) kernels for the
// DorPatch EOT hot path, behind the C ABI declared in include/dorpatch_hip.h.
//
// Design notes (see DESIGN.md for the roofline of every kernel):
//  * wave = 64 lanes, blocks = 256 threads (4 waves, one per SIMD);
//  * all streaming kernels move 16 B per lane (float4) with lane-contiguous
//    addresses -> 1 KiB per wave-instruction, fully coalesced;
//  * occlusion masks are never read from memory: a mask is <= 4 axis-aligned
//    windows fetched through the scalar unit (wave-uniform index), the per-pixel
//    test is a handful of VALU compares that hide under the HBM stream;
//  * every reduction has a fixed order (no float atomics): the optimiser takes
//    sign(grad), so run-to-run reproducibility matters more than a few us;
//  * neighbour (structural / TV-like) terms stage an image tile + 1-pixel halo
//    in LDS; partial sums use wave shuffles then a 4-entry LDS exchange;
//  * no fast-math: 0 * inf must stay NaN (group-lasso "frozen cell" semantics,
//    reference attack.py:243-245) and sign(NaN) must be 0 like torch.sign.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "dorpatch_hip.h"

// One library, one file per kernel family, five translation units (round 6).  This file holds what the
// families share (macros, reductions, the debug knobs) and the list of family files below; the product build
// (dorpatch_amd/build.py) compiles it five times IN PARALLEL with -DDP_PART=1..5 — 1: apply.inc + update.inc (the DorPatch
// arithmetic), gn.inc, stem.inc; 2: conv3x3.inc (direct 3x3, stride 1); 3: conv1x1.inc; 4: conv3x3_wino.inc; 5: conv3x3s2.inc
// (stride-2 3x3 and its input gradient) — and links the objects into the one libdorpatch_hip.so; a kernel, its launcher and
// its extern "C" entry point live in the same file.  DP_PART = 0 (the default: tools/kbench's white-box include, the host
// emulation of tests/hipemu) is the whole library in one unit.
#ifndef DP_PART
#define DP_PART 0
#endif
#define DP_HAS(part) (DP_PART == 0 || DP_PART == (part))

// dp_debug_set knobs (include/dorpatch_hip.h): launch-geometry overrides for tests / A-B runs, 0 = the product's choice.
// Defined in part 1 (where dp_debug_set lives), read by the launchers of every part.
namespace dp_state {
#if DP_HAS(1)
__attribute__((visibility("hidden"))) int g_aff_samples_per_block = 0;   // DP_DEBUG_AFFINE_SAMPLES_PER_BLOCK (tools/kbench sweeps it too)
__attribute__((visibility("hidden"))) int g_update_variant = 0;          // DP_DEBUG_UPDATE_VARIANT
__attribute__((visibility("hidden"))) int g_apply_order = 0;             // DP_DEBUG_APPLY_ORDER
__attribute__((visibility("hidden"))) int g_aff_gather = 0;              // DP_DEBUG_AFFINE_GATHER
__attribute__((visibility("hidden"))) int g_conv1x1_variant = 0;         // DP_DEBUG_CONV1X1_VARIANT: bits 0-1 workgroup map, bit 2 non-temporal stores, bit 3 LDS staging in one lump, bits 4-6 forced pixel tile
__attribute__((visibility("hidden"))) int g_conv3x3_variant = 0;         // DP_DEBUG_CONV3X3_VARIANT: bits 0-1: 0 per-side default, 1 k_conv3x3_mfma wherever it applies, 2 k_conv3x3_flat everywhere; bits 4-6 forced pixel tile
#else
extern int g_aff_samples_per_block, g_update_variant, g_apply_order, g_aff_gather, g_conv1x1_variant, g_conv3x3_variant;
#endif
}  // namespace dp_state
using namespace dp_state;

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;  // threads per workgroup: 4 waves of 64

// Register-allocation hint: the compiler must forget what it knows about a lane-private value (so that it re-derives
// addresses / predicates from it instead of keeping dozens of them alive).  No semantics; empty in the host emulation.
#ifdef HIPEMU_HOST
#define DP_LAUNDER(v) ((void)0)
#else
#define DP_LAUNDER(v) asm volatile("" : "+v"(v))
#endif

// Workgroup barrier that waits for this wave's LDS traffic only (lgkmcnt), not for its global loads: __syncthreads()
// also drains vmcnt, i.e. it would wait for prefetches that are meant to stay in flight ACROSS the barrier (they land in
// the wave's own registers; the compiler still inserts the vmcnt wait in front of their first use).
#ifdef HIPEMU_HOST
#define DP_BARRIER_LDS() __syncthreads()
#else
#define DP_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

#define DP_REQUIRE(cond)                                \
  do {                                                  \
    if (!(cond)) return (int)hipErrorInvalidValue;      \
  } while (0)

inline int launch_status() { return (int)hipGetLastError(); }
inline hipStream_t as_stream(dp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ----------------------------------------------------------------------------
// reductions
// ----------------------------------------------------------------------------

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

// Sum over the 256 threads of a block in a fixed order; result valid in thread 0.
__device__ __forceinline__ float block_sum(float v, float *sm4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) sm4[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = (sm4[0] + sm4[1]) + (sm4[2] + sm4[3]);
  return r;
}

__device__ __forceinline__ float sgn(float v) {
  // torch.sign semantics: sign(NaN) == 0, sign(+-0) == 0
  return (float)((v > 0.f) - (v < 0.f));
}

}  // namespace

// The kernel families, each with its launchers and its extern "C" entry points.
#include "conv_common.inc"    // shared by the convolution families (the stem convolution in stem.inc included)
#if DP_HAS(1)
#include "apply.inc"          // clip + blend, occlusion apply (+ the affine extension), slab sums
#include "update.inc"         // CW / structural losses, mask statistics, the fused projected update, argmax
#include "gn.inc"             // GroupNorm + ReLU
#include "stem.inc"           // pad + max-pool, stem convolution and its input gradient, subsampling
#endif
#if DP_HAS(2) || DP_HAS(5)
#include "conv3x3.inc"        // direct 3x3, stride 1 (entry points in part 2 only) + the flat-image walk the stride-2 input gradient shares
#endif
#if DP_HAS(5)
#include "conv3x3s2.inc"      // direct 3x3, stride 2, and its input gradient
#endif
#if DP_HAS(3)
#include "conv1x1.inc"        // k_conv1x1_mfma
#endif
#if DP_HAS(4)
#include "conv3x3_wino.inc"   // Winograd F(2x2, 3x3) on the matrix cores
#endif

// ============================================================================
// C ABI: version, errors, debug knobs, events (every other entry point sits with its kernels)
// ============================================================================

#if DP_HAS(1)
extern "C" {

int dp_abi_version(void) { return DP_ABI_VERSION; }

const char *dp_error_string(int err) { return hipGetErrorString((hipError_t)err); }

int dp_debug_set(int knob, int value) {
  switch (knob) {
    case DP_DEBUG_AFFINE_SAMPLES_PER_BLOCK:
      DP_REQUIRE(value >= 0 && value <= 64);
      g_aff_samples_per_block = value;
      return 0;
    case DP_DEBUG_UPDATE_VARIANT:
      DP_REQUIRE(value == 0 || value == 1);
      g_update_variant = value;
      return 0;
    case DP_DEBUG_APPLY_ORDER:
      DP_REQUIRE(value == 0 || value == 1);
      g_apply_order = value;
      return 0;
    case DP_DEBUG_AFFINE_GATHER:
      DP_REQUIRE(value >= 0 && value <= 2);
      g_aff_gather = value;
      return 0;
    case DP_DEBUG_CONV1X1_VARIANT:
      DP_REQUIRE(value >= 0 && value < 512 && (value & 3) != 3 && ((value >> 4) & 7) <= 4);     // bits 4-6: forced pixel tile, 7-8: fold preference (A/B)
      g_conv1x1_variant = value;
      return 0;
    case DP_DEBUG_CONV3X3_VARIANT:
      DP_REQUIRE(value >= 0 && (value & 15) <= 2 && (value >> 4) <= 4 && (value >> 4) != 2);     // bits 4-6: forced pixel tile
      g_conv3x3_variant = value;
      return 0;
    default:
      return (int)hipErrorInvalidValue;
  }
}

/* ---- measurement support: kernel-precise timing of the dominant kernel (bench.py roofline) ---- */
int dp_event_create(dp_event_t *ev) {
  DP_REQUIRE(ev);
  hipEvent_t e;
  const hipError_t rc = hipEventCreate(&e);
  *ev = rc == hipSuccess ? (dp_event_t)e : nullptr;
  return (int)rc;
}

int dp_event_destroy(dp_event_t ev) {
  DP_REQUIRE(ev);
  return (int)hipEventDestroy((hipEvent_t)ev);
}

int dp_event_elapsed_ms(dp_event_t start, dp_event_t stop, float *ms) {
  DP_REQUIRE(start && stop && ms);
  hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
  if (rc != hipSuccess) return (int)rc;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}

}  // extern "C"
#endif   // part 1
