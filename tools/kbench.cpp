// kbench — standalone HIP micro-benchmark of every libdorpatch_hip.so entry point at
// BASELINE.json configs[1] geometry (64 images x 32 sampled double-masks @224x224, fp32),
// plus two calibration kernels (write-only fill, float4 copy) that give this box's
// achievable HBM ceilings.  No Python / torch: starts in milliseconds on the GPU box.
//
//   build:  python -m dorpatch_amd.build  &&  make -C tools        (or __graft_entry__.build())
//   run:    tools/kbench [B S H iters [name-filter]]     e.g. tools/kbench 64 32 224 2 "dp_apply_fwd (default"
//
// Output: one line per kernel: name, avg ms (hipEvent on the launch stream), algorithmic
// bytes per launch, GB/s, fraction of the 8 TB/s spec.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>
#include <array>

// White-box: the product translation unit is compiled INTO this binary so that the kernel
// variants behind dp_apply_fwd (launch_apply_fwd(variant, ...)) can be swept; every other kernel
// is still called through its extern "C" entry point.
#include "../dorpatch_amd/csrc/dorpatch_hip.hip"

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)
#define DP(x)                                                                \
  do {                                                                       \
    int e_ = (x);                                                            \
    if (e_ != 0) {                                                           \
      fprintf(stderr, "dp error %d (%s) at %s:%d\n", e_, dp_error_string(e_), __FILE__, __LINE__); \
      exit(3);                                                               \
    }                                                                        \
  } while (0)

__global__ __launch_bounds__(256) void k_fill(f4 *__restrict__ out, size_t n4, float v) {
  const f4 val = {v, v, v, v};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(val, out + i);
}

__global__ __launch_bounds__(256) void k_copy(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

// Calibration of the matrix-core convolutions' loop structure (round 5): what does the MI355X give a workgroup of 4 waves,
// 7 accumulators per wave, that does NOTHING but the 1x1 kernel's k-step walk?  MODE 0: operands stay in registers (pure
// v_mfma_f32_32x32x2_f32 issue); 1: + the 8 LDS operand reads per k-step, requested one step ahead; 2: + one workgroup
// barrier per 56 MFMAs (before the last group, like the kernel); 3: + 8 ds_write_b128 per chunk spread over the groups.
// Launched with 2 workgroups per CU (64 KB of LDS each) and with 1.
// MODE 4: + 8 global_load_dwordx4 per lane and chunk (32 KB per workgroup: the kernel's staging volume) from an L2-sized
// region, issued in ONE burst after the barrier and consumed (stored to LDS) a chunk later; 5: the same loads issued one
// after each MFMA group.
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_mfma_probe(float *__restrict__ out, int chunks, const float *__restrict__ src = nullptr) {
  constexpr int kC1Buf = C1Geom<7>::BUF, kC1In = C1Geom<7>::IN;      // the full-size (448-pixel) tile
  __shared__ __attribute__((aligned(16))) float lds[2 * kC1Buf];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31;
  for (int i = tid; i < 2 * kC1Buf; i += 256) lds[i] = 1e-3f * (float)((i * 7 + 3) & 63);
  __syncthreads();
  const int abase = kC1In + half * kC1O + (wave & 1) * 32 + l32;
  int boff[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) boff[q] = half * kC1Pix + ((wave >> 1) + 2 * q) * 32 + l32;
  f16v acc[7];
#pragma unroll
  for (int q = 0; q < 7; ++q)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[q][v] = 0.f;
  auto operands = [&](const float *cur, int t, float &a, float (&bv)[7]) {
    a = cur[abase + t * 2 * kC1O];
#pragma unroll
    for (int q = 0; q < 7; ++q) bv[q] = cur[boff[q] + t * 2 * kC1Pix];
  };
  float a0, b0[7], a1, b1[7];
  operands(lds, 0, a0, b0);
  operands(lds, 1, a1, b1);
  f4 junk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) junk[i] = f4{1.f, 2.f, 3.f, 4.f};
  // 64 MB region (16 M floats), every workgroup its own 32 KB per chunk, wrapping: mostly L2 / MALL hits like the kernel's x
  const size_t region = (size_t)16 << 20;
  size_t goff = ((size_t)blockIdx.x * 8192 + 4 * tid) % region;
  auto gload = [&](int i) {
    if (MODE >= 4) junk[i] = *reinterpret_cast<const f4 *>(src + (goff + (size_t)i * 1024) % region);
  };
  if (MODE >= 4)
#pragma unroll
    for (int i = 0; i < 8; ++i) gload(i);
  for (int chunk = 0; chunk < chunks; ++chunk) {
    const float *cur = lds + (chunk & 1) * kC1Buf;
    float *nxt = lds + ((chunk + 1) & 1) * kC1Buf;
    goff = (goff + (size_t)gridDim.x * 8192) % region;
#pragma unroll
    for (int t = 0; t < kC1Steps; t += 2) {
      if (MODE >= 1) operands(cur, t + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 7; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[q], acc[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE >= 3) *reinterpret_cast<f4 *>(nxt + 4 * (tid + t * 256)) = junk[t];
      if (MODE == 5) gload(t);
      if (t + 2 < kC1Steps) {
        if (MODE >= 1) operands(cur, t + 2, a0, b0);
      } else {
        if (MODE >= 2) DP_BARRIER_LDS();
        if (MODE >= 1) operands(nxt, 0, a0, b0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 7; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[q], acc[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE >= 3) *reinterpret_cast<f4 *>(nxt + 4 * (tid + (t + 1) * 256)) = junk[t + 1];
      if (MODE == 5) gload(t + 1);
      if (MODE == 4 && t + 2 == kC1Steps)
#pragma unroll
        for (int i = 0; i < 8; ++i) gload(i);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 7; ++q)
#pragma unroll
    for (int v = 0; v < 16; ++v) s += acc[q][v];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

// write-only fill with the gfx950 store flavours (MI355X_MICROARCH.md, "stores of each flavour"): which one gives the
// best pure-write HBM rate?  F: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 nt sc0 sc1.  One WG per 32 KiB.
template <int F>
__device__ __forceinline__ void store_flavour(f4 *p, f4 v) {
  if (F == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (F == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if (F == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (F == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (F == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  if (F == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int F>
__global__ __launch_bounds__(256) void k_fill_flavour(f4 *__restrict__ out, size_t n4, float v) {
  const f4 val = {v, v, v, v};
  const size_t base = (size_t)blockIdx.x * 2048;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const size_t i = base + (size_t)u * 256 + threadIdx.x;
    if (i < n4) store_flavour<F>(out + i, val);
  }
}

// U independent 16-B loads in flight per lane before the first store (block-contiguous chunks of U KiB per wave)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_u(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n4) {
  const size_t per_block = (size_t)256 * U;
  for (size_t base = (size_t)blockIdx.x * per_block; base < n4; base += (size_t)gridDim.x * per_block) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 256 + threadIdx.x;
      const size_t ic = i < n4 ? i : 0;
      v[u] = NT ? __builtin_nontemporal_load(in + ic) : in[ic];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 256 + threadIdx.x;
      if (i < n4) {
        if (NT) __builtin_nontemporal_store(v[u], out + i);
        else out[i] = v[u];
      }
    }
  }
}

// The access pattern of dp_project_update without its arithmetic or its halo: per pixel read x 12 + adv 12 + g 12 + lv 4 +
// pattern 12 + mask 4, write pattern 12 + mask 4 IN PLACE — what this GPU's memory system gives a 6-stream read /
// 2-stream read-modify-write kernel (the ceiling the real kernel can be held to).  One lane = 4 consecutive pixels.
__global__ __launch_bounds__(256) void k_calib_update_pattern(const float *__restrict__ x, const float *__restrict__ adv,
                                                              const float *__restrict__ g, const float *__restrict__ lv,
                                                              float *__restrict__ pattern, float *__restrict__ mask, int P) {
  const int b = blockIdx.y;
  const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (p >= P) return;
  const f4 m4 = *reinterpret_cast<const f4 *>(mask + (size_t)b * P + p);
  const f4 l4 = *reinterpret_cast<const f4 *>(lv + (size_t)b * P + p);
  f4 acc = m4 + l4;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const size_t off = ((size_t)b * 3 + c) * P + p;
    const f4 a = *reinterpret_cast<const f4 *>(x + off), d = *reinterpret_cast<const f4 *>(adv + off);
    const f4 e = *reinterpret_cast<const f4 *>(g + off), q = *reinterpret_cast<const f4 *>(pattern + off);
    const f4 r = (a - d) * e + q * 0.999f;
    acc = acc + r;
    *reinterpret_cast<f4 *>(pattern + off) = r;
  }
  *reinterpret_cast<f4 *>(mask + (size_t)b * P + p) = acc * 0.5f;
}

__global__ __launch_bounds__(256) void k_count_diff(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, size_t n,
                                                    int *__restrict__ count) {
  int local = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) local += a[i] != b[i];
  if (local) atomicAdd(count, local);
}

// max |a - b| and max |b| over two fp32 arrays (variants that sum in another order are compared by value, not by bytes)
__global__ __launch_bounds__(256) void k_max_diff(const float *__restrict__ a, const float *__restrict__ b, size_t n,
                                                  unsigned *__restrict__ out /* [2]: bits of max diff, max |b| */) {
  float d = 0.f, m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    d = fmaxf(d, fabsf(a[i] - b[i]));
    m = fmaxf(m, fabsf(b[i]));
  }
  atomicMax(out, __float_as_uint(d));       // non-negative floats order like their bit patterns
  atomicMax(out + 1, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void k_readsum(const f4 *__restrict__ in, float *__restrict__ out, size_t n4) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
    acc += __builtin_nontemporal_load(in + i);
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;  // keep the loads alive
}

// fp32 VALU issue rates: N dependent-chain-free FMAs per lane, scalar (v_fma_f32) vs packed (v_pk_fma_f32: 2 FMAs per
// instruction).  The 157.3 TFLOP/s vector spec counts the packed form; what does this GPU sustain?
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PACKED>   // 0: v_fma_f32 all-VGPR, 1: v_pk_fma_f32, 2: v_fmac_f32 with an SGPR multiplier (k_stem_dgrad's form)
__global__ __launch_bounds__(256) void k_fma_rate(float *__restrict__ out, int iters, float seed) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
  const float m = 1.0000001f, c = 1e-7f;
  for (int it = 0; it < iters; ++it) {
    if (PACKED == 4) {   // VOP2 fmac, all operands VGPRs
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    } else if (PACKED == 5) {   // VOP3 fma with an SGPR multiplier
      const float sm = __builtin_amdgcn_readfirstlane(m);
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sm), "v"(c));
    } else if (PACKED == 3) {   // multiplier broadcast from lane 3 of each row of 16 lanes of a VGPR (DPP row_newbcast)
      float mv = m;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(mv), "v"(c));
    } else if (PACKED == 2) {
      const float sm = __builtin_amdgcn_readfirstlane(m);
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sm), "v"(c));
    } else if (PACKED == 1) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f2 v = {a[i], a[i + 1]};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(f2{m, m}), "v"(f2{c, c}));
        a[i] = v.x;
        a[i + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 12345.678f) out[0] = s;
}

static const char *g_filter = nullptr;  // 5th argument: run only the entries whose name contains it

static double bench(const char *name, double bytes, int iters, hipStream_t st, const std::function<void()> &fn) {
  if (g_filter && !strstr(name, g_filter)) return 0.0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) fn();
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  const double gbs = bytes / (ms * 1e-3) / 1e9;
  printf("%-34s %9.4f ms  %12.0f B  %8.1f GB/s  %5.1f%% of 8 TB/s\n", name, ms, bytes, gbs, gbs / 80.0);
  fflush(stdout);
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms;
}

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64;
  const int S = argc > 2 ? atoi(argv[2]) : 32;
  const int H = argc > 3 ? atoi(argv[3]) : 224;
  const int iters = argc > 4 ? atoi(argv[4]) : 20;
  g_filter = argc > 5 ? argv[5] : nullptr;
  const int W = H, P = H * W, N = B * S;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s, %d CUs; B=%d S=%d H=%d (N=%d masked images, %.3f GB), iters=%d\n", prop.name,
         prop.multiProcessorCount, B, S, H, N, (double)N * 3 * P * 4 / 1e9, iters);
  hipStream_t st;
  CK(hipStreamCreate(&st));

  // ---- PatchCleanser double-mask universe (defenses/PatchCleanser.py:11-16, 51-58, 23-29)
  std::vector<int32_t> table;
  const double ratios[4] = {0.015, 0.03, 0.06, 0.12};
  for (double r : ratios) {
    const int mask_size = (int)std::floor(std::sqrt((double)H * H * r));
    const int stride = (int)std::ceil((H - mask_size + 1) / 6.0);
    const int window = mask_size + stride - 1;
    int rc[36][4];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        int *q = rc[i * 6 + j];
        q[0] = stride * i; q[1] = std::min(H, stride * i + window);
        q[2] = stride * j; q[3] = std::min(H, stride * j + window);
      }
    for (int a = 0; a < 36; ++a)
      for (int b = a + 1; b < 36; ++b) {
        table.insert(table.end(), rc[a], rc[a] + 4);
        table.insert(table.end(), rc[b], rc[b] + 4);
      }
  }
  const int n_mask = (int)table.size() / 8;
  std::mt19937 rng(1234);
  std::vector<int32_t> idx((size_t)B * S);
  for (auto &v : idx) v = (int32_t)(rng() % n_mask);

  auto dmalloc = [](size_t bytes) { void *p; CK(hipMalloc(&p, bytes)); return p; };
  const size_t img = (size_t)3 * P * 4;
  float *x = (float *)dmalloc(B * img), *pattern = (float *)dmalloc(B * img), *adv = (float *)dmalloc(B * img);
  float *g_adv = (float *)dmalloc(B * img), *best_p = (float *)dmalloc(B * img);
  float *mask = (float *)dmalloc((size_t)B * P * 4), *best_m = (float *)dmalloc((size_t)B * P * 4);
  float *lv = (float *)dmalloc((size_t)B * P * 4);
  float *big = (float *)dmalloc((size_t)N * img), *big2 = (float *)dmalloc((size_t)N * img);
  int32_t *d_table = (int32_t *)dmalloc(table.size() * 4), *d_idx = (int32_t *)dmalloc(idx.size() * 4);
  CK(hipMemcpy(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
  {  // random fill of the small state (host rng -> device)
    std::vector<float> h((size_t)B * 3 * P);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    for (float *dst : {x, pattern, g_adv}) {
      for (auto &v : h) v = U(rng);
      CK(hipMemcpy(dst, h.data(), B * img, hipMemcpyHostToDevice));
    }
    for (auto &v : h) v = U(rng);
    CK(hipMemcpy(mask, h.data(), (size_t)B * P * 4, hipMemcpyHostToDevice));
  }
  const size_t n4_big = (size_t)N * 3 * P / 4;
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (f4 *)big2, n4_big, 0.25f);

  const int nchunk = dp_sumsq_nchunk(P), ntile = dp_struct_ntile(H, W);
  float *partials = (float *)dmalloc((size_t)B * nchunk * 4), *tpart = (float *)dmalloc((size_t)B * ntile * 4);
  float *scale = (float *)dmalloc(B * 4), *l2 = (float *)dmalloc(B * 4), *sloss = (float *)dmalloc(B * 4);
  float *structured = (float *)dmalloc(B * 4), *coeff = (float *)dmalloc(B * 4), *lr = (float *)dmalloc(B * 4);
  float *gl = (float *)dmalloc(B * 4), *dens = (float *)dmalloc(B * 4);
  {
    std::vector<float> a(B, 1e-3f), b(B, 1e-5f), c(B, 1e-2f);
    CK(hipMemcpy(structured, a.data(), B * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(coeff, b.data(), B * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(lr, c.data(), B * 4, hipMemcpyHostToDevice));
  }
  const int unit = 7, win = W / 8;
  const int ncy = (H - unit) / unit + 1, nwy = (H - win) / win + 1;
  float *cell = (float *)dmalloc((size_t)B * ncy * ncy * 4), *wsum = (float *)dmalloc((size_t)B * nwy * nwy * 4);
  const int C = 1000;
  float *logits = (float *)dmalloc((size_t)N * C * 4), *dlogits = (float *)dmalloc((size_t)N * C * 4);
  float *loss = (float *)dmalloc((size_t)N * 4);
  int32_t *pred = (int32_t *)dmalloc((size_t)N * 4), *tflag = (int32_t *)dmalloc(B * 4);
  int64_t *y = (int64_t *)dmalloc(B * 8);
  CK(hipMemset(tflag, 0, B * 4));
  CK(hipMemset(y, 0, B * 8));
  hipLaunchKernelGGL(k_fill, dim3(512), dim3(256), 0, st, (f4 *)logits, (size_t)N * C / 4, 0.5f);
  const int nslab = dp_apply_bwd_nslab(B, S, P);
  float *slabs = (float *)dmalloc((size_t)nslab * B * img);
  CK(hipStreamSynchronize(st));

  dp_norm_t norm = {1, {0.5f, 0.5f, 0.5f}, {0.5f, 0.5f, 0.5f}, 0.5f};
  const double out_bytes = (double)N * img;

  if (g_filter && strstr(g_filter, "conv3x3")) {
    // VERDICT r3 item 7 / r4 items 5 + 6: the stride-1 3x3 convolutions of ResNetV2-50 at the training micro-batch (B here = N
    // of the convolution) on the matrix cores — all four 224-input shapes have the same 118 GFLOP at N = 512 — on both
    // kernels (k_conv3x3_mfma: zero rows / columns laid out in LDS; k_conv3x3_flat: flat image + masked taps), plain and with
    // the GroupNorm fold.  DP_C3_SIDES=384: the planes of a 384 x 384 input (flat kernel only).
    const int Nc = B;
    const char *sides = getenv("DP_C3_SIDES");
    const bool big = sides && strstr(sides, "384");
    const int shapes224[4][2] = {{64, 56}, {128, 28}, {256, 14}, {512, 7}};
    const int shapes384[4][2] = {{64, 96}, {128, 48}, {256, 24}, {512, 12}};
    for (int si = 0; si < 4; ++si) {
      const int Cc = big ? shapes384[si][0] : shapes224[si][0], Sc = big ? shapes384[si][1] : shapes224[si][1];
      const size_t e = (size_t)Nc * Cc * Sc * Sc;
      float *cx = (float *)dmalloc(e * 4), *cy = (float *)dmalloc(e * 4), *cw = (float *)dmalloc((size_t)Cc * Cc * 16 * 4);   // 16: room for the Winograd-domain filter
      float *cab = (float *)dmalloc((size_t)Nc * Cc * 2 * 4);
      hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (f4 *)cx, e / 4, 0.37f);
      hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, (f4 *)cw, (size_t)Cc * Cc * 16 / 4, 0.01f);
      hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, (f4 *)cab, (size_t)Nc * Cc * 2 / 4, 0.5f);
      const double flop = 2.0 * Nc * Sc * Sc * (double)Cc * Cc * 9;
      // $DP_C3_VARIANTS = values of DP_DEBUG_CONV3X3_VARIANT to run (default: 1 = k_conv3x3_mfma, 2 = k_conv3x3_flat; round 6:
      // 16 / 48 / 64 = k_conv3x3_flat with 448- / 128- / 64-pixel tiles, 0 = the launcher's own choice, 1000 = dp_conv3x3_wino_fwd)
      std::vector<int> c3v;
      if (const char *ev = getenv("DP_C3_VARIANTS")) {
        int v, n = 0;
        for (const char *p = ev; sscanf(p, "%d%n", &v, &n) == 1; p += n + (p[n] == ',')) c3v.push_back(v);
      } else {
        for (int v = big ? 2 : 1; v <= 2; ++v) c3v.push_back(v);
      }
      for (int variant : c3v) {
        if ((variant & 3) == 1 && big && variant != 1000) continue;
        const bool wino = variant == 1000;          // round 6: dp_conv3x3_wino_fwd (Winograd F(2x2, 3x3) on the matrix cores)
        if (!wino) DP(dp_debug_set(DP_DEBUG_CONV3X3_VARIANT, variant));
        for (int fold = 0; fold < 2; ++fold) {
          if (fold && Sc == 7 && !wino) continue;
          for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            auto run = [&]() {
              if (wino) DP(dp_conv3x3_wino_fwd(cx, cw, fold ? cab : nullptr, Nc, Cc, Cc, Sc, Sc, cy, st));
              else if (fold) DP(dp_conv3x3_gn_fwd(cx, cw, cab, Nc, Cc, Cc, Sc, Sc, cy, st));
              else DP(dp_conv3x3_fwd(cx, cw, Nc, Cc, Cc, Sc, Sc, cy, st));
              return 0;
            };
            if (run()) return 1;
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) if (run()) return 1;
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters;
            printf("dp_conv3x3_fwd %3d->%3d @%2dx%2d N=%d %s %s  %8.4f ms  %7.1f TFLOP/s  (%.1f%% of the 157.3 TFLOP/s f32 peak)\n",
                   Cc, Cc, Sc, Sc, Nc, variant == 1 ? "k_conv3x3_mfma" : variant == 2 ? "k_conv3x3_flat" : variant == 0 ? "auto          " :
                   variant == 1000 ? "k_conv3x3_wino" :
                   variant == 16 ? "flat 448px    " : variant == 48 ? "flat 128px    " : variant == 64 ? "flat  64px    " : "other         ",
                   fold ? "fold " : "plain", ms,
                   flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 1.573);
          }
        }
      }
      DP(dp_debug_set(DP_DEBUG_CONV3X3_VARIANT, 0));
      CK(hipFree(cx)); CK(hipFree(cy)); CK(hipFree(cw)); CK(hipFree(cab));
    }
    return 0;
  }
  if (g_filter && strstr(g_filter, "stemconv")) {
    // round 5: the stem convolution (3 -> 64, 7x7 / 2 @224) on the matrix cores; B = images
    const int Nc = B;
    const size_t ex = (size_t)Nc * 3 * 224 * 224, ey = (size_t)Nc * 64 * 112 * 112;
    float *cx = (float *)dmalloc(ex * 4), *cy = (float *)dmalloc(ey * 4), *cw = (float *)dmalloc(77 * 128 * 4);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (f4 *)cx, ex / 4, 0.37f);
    hipLaunchKernelGGL(k_fill, dim3(8), dim3(256), 0, st, (f4 *)cw, (size_t)77 * 128 / 4, 0.01f);
    const double flop = 2.0 * Nc * 112 * 112 * 147.0 * 64;
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      DP(dp_stem_conv_fwd(cx, cw, Nc, 224, 224, cy, st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) DP(dp_stem_conv_fwd(cx, cw, Nc, 224, 224, cy, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      printf("dp_stem_conv_fwd 3->64 7x7/2 @224x224 N=%d  %8.4f ms  %7.1f TFLOP/s useful (%.1f%% of the 157.3 TFLOP/s f32 peak)  %.2f TB/s (x read + y written)\n",
             Nc, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 1.573, (ex + ey) * 4.0 / (ms * 1e-3) / 1e12);
    }
    CK(hipFree(cx)); CK(hipFree(cy)); CK(hipFree(cw));
    return 0;
  }
  if (g_filter && strstr(g_filter, "conv3s2bwd")) {
    // round 5: the input gradients of the three stride-2 3x3 convolutions (four masked parity-class walks, one launch)
    const int Nc = B;
    const char *sides = getenv("DP_C3_SIDES");
    const bool big = sides && strstr(sides, "384");
    const int shapes224[3][2] = {{128, 28}, {256, 14}, {512, 7}};
    const int shapes384[3][2] = {{128, 48}, {256, 24}, {512, 12}};
    for (int si = 0; si < 3; ++si) {
      const int Cc = big ? shapes384[si][0] : shapes224[si][0], So = big ? shapes384[si][1] : shapes224[si][1];
      const size_t ey = (size_t)Nc * Cc * So * So, ex = 4 * ey;
      float *cdy = (float *)dmalloc(ey * 4), *cdx = (float *)dmalloc(ex * 4), *cw = (float *)dmalloc((size_t)Cc * Cc * 9 * 4);
      hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (f4 *)cdy, ey / 4, 0.37f);
      hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, (f4 *)cw, (size_t)Cc * Cc * 9 / 4, 0.01f);
      const double flop = 2.0 * Nc * So * So * (double)Cc * Cc * 9;
      for (int form = 0; form < 2; ++form)
      for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        DP(dp_conv3x3s2_bwd(cdy, cw, Nc, Cc, Cc, So, So, cdx, form, st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) DP(dp_conv3x3s2_bwd(cdy, cw, Nc, Cc, Cc, So, So, cdx, form, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        printf("dp_conv3x3s2_bwd %3d->%3d @%2dx%2d -> %2dx%2d N=%d %s  %8.4f ms  %7.1f TFLOP/s  (%.1f%% of the 157.3 TFLOP/s f32 peak)  %.2f TB/s written\n",
               Cc, Cc, So, So, 2 * So, 2 * So, Nc, form == DP_S2BWD_PAIRS ? "pairs  " : "classes", ms, flop / (ms * 1e-3) / 1e12,
               flop / (ms * 1e-3) / 1e12 / 1.573, ex * 4.0 / (ms * 1e-3) / 1e12);
      }
      CK(hipFree(cdy)); CK(hipFree(cdx)); CK(hipFree(cw));
    }
    return 0;
  }
  if (g_filter && strstr(g_filter, "conv3s2")) {
    // round 5: the three stride-2 3x3 convolutions (conv2 of the first bottleneck of stages 2-4), plain and with the
    // GroupNorm fold (coefficients = a constant table: the kernel's work does not depend on their values)
    const int Nc = B;
    const int shapes[3][2] = {{128, 56}, {256, 28}, {512, 14}};
    for (auto &sh : shapes) {
      const int Cc = sh[0], Sc = sh[1], So = Sc / 2;
      const size_t ex = (size_t)Nc * Cc * Sc * Sc, ey = (size_t)Nc * Cc * So * So;
      float *cx = (float *)dmalloc(ex * 4), *cy = (float *)dmalloc(ey * 4), *cw = (float *)dmalloc((size_t)Cc * Cc * 9 * 4);
      float *cab = (float *)dmalloc((size_t)Nc * Cc * 2 * 4);
      hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (f4 *)cx, ex / 4, 0.37f);
      hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, (f4 *)cw, (size_t)Cc * Cc * 9 / 4, 0.01f);
      hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, (f4 *)cab, (size_t)Nc * Cc * 2 / 4, 0.5f);
      const double flop = 2.0 * Nc * So * So * (double)Cc * Cc * 9;
      for (int fold = 0; fold < 2; ++fold) {
        for (int rep = 0; rep < 2; ++rep) {
          hipEvent_t e0, e1;
          CK(hipEventCreate(&e0));
          CK(hipEventCreate(&e1));
          DP(dp_conv3x3s2_fwd(cx, cw, fold ? cab : nullptr, Nc, Cc, Cc, Sc, Sc, cy, st));
          CK(hipEventRecord(e0, st));
          for (int i = 0; i < iters; ++i) DP(dp_conv3x3s2_fwd(cx, cw, fold ? cab : nullptr, Nc, Cc, Cc, Sc, Sc, cy, st));
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          ms /= iters;
          printf("dp_conv3x3s2_fwd %3d->%3d @%2dx%2d -> %2dx%2d N=%d %s  %8.4f ms  %7.1f TFLOP/s  (%.1f%% of the 157.3 TFLOP/s f32 peak)\n",
                 Cc, Cc, Sc, Sc, So, So, Nc, fold ? "fold " : "plain", ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 1.573);
        }
      }
      CK(hipFree(cx)); CK(hipFree(cy)); CK(hipFree(cw)); CK(hipFree(cab));
    }
    return 0;
  }
  if (g_filter && strstr(g_filter, "mfma_probe")) {
    float *po = (float *)dmalloc((size_t)2048 * 256 * 4);
    float *psrc = (float *)dmalloc((size_t)64 << 20);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (f4 *)psrc, (size_t)(16 << 20) / 4, 0.25f);
    const int chunks = 256;
    for (int wgs : {512, 256, 1024}) {
      for (int mode = 0; mode < 6; ++mode) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        auto launch = [&]() {
          if (mode == 0) hipLaunchKernelGGL(k_mfma_probe<0>, dim3(wgs), dim3(256), 0, st, po, chunks);
          else if (mode == 1) hipLaunchKernelGGL(k_mfma_probe<1>, dim3(wgs), dim3(256), 0, st, po, chunks);
          else if (mode == 2) hipLaunchKernelGGL(k_mfma_probe<2>, dim3(wgs), dim3(256), 0, st, po, chunks);
          else if (mode == 3) hipLaunchKernelGGL(k_mfma_probe<3>, dim3(wgs), dim3(256), 0, st, po, chunks);
          else if (mode == 4) hipLaunchKernelGGL(k_mfma_probe<4>, dim3(wgs), dim3(256), 0, st, po, chunks, psrc);
          else hipLaunchKernelGGL(k_mfma_probe<5>, dim3(wgs), dim3(256), 0, st, po, chunks, psrc);
        };
        launch();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        const double flop = (double)wgs * 4 * chunks * 56 * 4096.0;
        printf("k_mfma_probe mode %d (%s) %4d workgroups x 4 waves x %d chunks x 56 MFMAs  %8.4f ms  %6.1f TFLOP/s (%4.1f%% of 157.3)\n", mode,
               mode == 0 ? "registers only" : mode == 1 ? "+ LDS operand reads" : mode == 2 ? "+ barrier / chunk" : mode == 3 ? "+ 8 ds_write_b128 / chunk" :
               mode == 4 ? "+ 8 global_load_dwordx4 / chunk, one burst" : "+ 8 global_load_dwordx4 / chunk, one per MFMA group",
               wgs, chunks, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 1.573);
      }
    }
    CK(hipFree(po));
    CK(hipFree(psrc));
    return 0;
  }
  if (g_filter && strstr(g_filter, "conv1x1")) {
    // Round 5: the 1x1 convolutions of ResNetV2-50 (B here = N of the convolution) on the matrix cores, every launch
    // variant of DP_DEBUG_CONV1X1_VARIANT named in $DP_C1_VARIANTS (default "0"); $DP_C1_SHAPES = "C:O:S,..." restricts
    // the shapes (for PMC passes).  `fold` / `res` rows: the GroupNorm fold and the epilogue add on the same shape.
    const int Nc = B;
    std::vector<std::array<int, 3>> shapes = {{64, 64, 56}, {64, 256, 56}, {256, 64, 56}, {256, 128, 56}, {128, 512, 28},
        {256, 512, 28}, {512, 128, 28}, {512, 256, 28}, {256, 1024, 14}, {512, 1024, 14}, {1024, 256, 14}, {1024, 512, 14},
        {512, 2048, 7}, {1024, 2048, 7}, {2048, 512, 7}};
    if (const char *e = getenv("DP_C1_SHAPES")) {
      shapes.clear();
      int c, o, s2, n = 0;
      for (const char *p = e; sscanf(p, "%d:%d:%d%n", &c, &o, &s2, &n) == 3; p += n + (p[n] == ',')) shapes.push_back({c, o, s2});
    }
    std::vector<int> variants = {0};
    if (const char *e = getenv("DP_C1_VARIANTS")) {
      variants.clear();
      int v, n = 0;
      for (const char *p = e; sscanf(p, "%d%n", &v, &n) == 1; p += n + (p[n] == ',')) variants.push_back(v);
    }
    for (auto &sh : shapes) {
      const int Cc = sh[0], Oc = sh[1], Sc = sh[2], HWc = Sc * Sc;
      const size_t ex = (size_t)Nc * Cc * HWc, ey = (size_t)Nc * Oc * HWc;
      float *cx = (float *)dmalloc(ex * 4), *cy = (float *)dmalloc(ey * 4), *cw = (float *)dmalloc((size_t)Cc * Oc * 4);
      float *cab = (float *)dmalloc((size_t)Nc * Cc * 2 * 4), *cr = (float *)dmalloc(ey * 4);
      hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (f4 *)cx, ex / 4, 0.37f);
      hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (f4 *)cr, ey / 4, 0.11f);
      hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, (f4 *)cw, (size_t)Cc * Oc / 4, 0.01f);
      hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, (f4 *)cab, (size_t)Nc * Cc * 2 / 4, 0.5f);
      const double flop = 2.0 * Nc * HWc * (double)Cc * Oc, bytes = 4.0 * (ex + ey);
      for (int v : variants) {
        for (int mode = 0; mode < 4; ++mode) {     // 0 plain, 1 fold, 2 res, 3 fold + res ($DP_C1_MODES = "3" / "0,3": only those)
          if ((mode & 1) && (HWc & 3)) continue;
          if (mode && getenv("DP_C1_PLAIN_ONLY")) continue;
          if (const char *em = getenv("DP_C1_MODES")) {
            if (!strchr(em, '0' + mode)) continue;
          } else if (mode == 3) continue;
          DP(dp_debug_set(DP_DEBUG_CONV1X1_VARIANT, v));
          hipEvent_t e0, e1;
          CK(hipEventCreate(&e0));
          CK(hipEventCreate(&e1));
          const float *ab = (mode & 1) ? cab : nullptr, *res = (mode & 2) ? cr : nullptr;
          DP(dp_conv1x1_fwd(cx, cw, ab, res, Nc, Cc, Oc, HWc, cy, st));
          CK(hipEventRecord(e0, st));
          for (int i = 0; i < iters; ++i) DP(dp_conv1x1_fwd(cx, cw, ab, res, Nc, Cc, Oc, HWc, cy, st));
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          ms /= iters;
          printf("dp_conv1x1_fwd %4d->%4d @%2dx%2d N=%d variant %2d %-5s %8.4f ms  %6.1f TFLOP/s (%4.1f%% of 157.3)  %5.2f TB/s algorithmic\n",
                 Cc, Oc, Sc, Sc, Nc, v, mode == 0 ? "plain" : mode == 1 ? "fold" : mode == 2 ? "res" : "f+res", ms, flop / (ms * 1e-3) / 1e12,
                 flop / (ms * 1e-3) / 1e12 / 1.573, (bytes + ((mode & 2) ? 4.0 * ey : 0.0)) / (ms * 1e-3) / 1e12);
          fflush(stdout);
        }
      }
      DP(dp_debug_set(DP_DEBUG_CONV1X1_VARIANT, 0));
      CK(hipFree(cx)); CK(hipFree(cy)); CK(hipFree(cw)); CK(hipFree(cab)); CK(hipFree(cr));
    }
    return 0;
  }
  if (g_filter && strstr(g_filter, "fma_rate")) {
    const int it = 4096, blocks = 256 * 16;
    for (int packed = 0; packed < 6; ++packed) {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      auto run = [&] {
        if (packed == 1) hipLaunchKernelGGL((k_fma_rate<1>), dim3(blocks), dim3(256), 0, st, loss, it, 0.5f);
        else if (packed == 2) hipLaunchKernelGGL((k_fma_rate<2>), dim3(blocks), dim3(256), 0, st, loss, it, 0.5f);
        else if (packed == 3) hipLaunchKernelGGL((k_fma_rate<3>), dim3(blocks), dim3(256), 0, st, loss, it, 0.5f);
        else if (packed == 4) hipLaunchKernelGGL((k_fma_rate<4>), dim3(blocks), dim3(256), 0, st, loss, it, 0.5f);
        else if (packed == 5) hipLaunchKernelGGL((k_fma_rate<5>), dim3(blocks), dim3(256), 0, st, loss, it, 0.5f);
        else hipLaunchKernelGGL((k_fma_rate<0>), dim3(blocks), dim3(256), 0, st, loss, it, 0.5f);
      };
      run();
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 5; ++i) run();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 5;
      const double flop = 2.0 * 16 * it * 256.0 * blocks;   // v_fmac a += s * c: also 16 FMAs per iteration
      printf("fma_rate %-22s %8.3f ms  %7.1f TFLOP/s\n", packed == 1 ? "v_pk_fma_f32 (8/iter)" : packed == 2 ? "v_fmac_f32 v,s,v (16/iter)" : packed == 3 ? "v_fmac_f32_dpp row_newbcast" : packed == 4 ? "v_fmac_f32 v,v,v" : packed == 5 ? "v_fma_f32 v,s,v,v" : "v_fma_f32 (16/iter)", ms, flop / (ms * 1e-3) / 1e12);
    }
    return 0;
  }
  // ---- calibration: what this box's HBM does for the same footprint
  bench("calib: fill (write-only)", out_bytes, iters, st,
        [&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (f4 *)big, n4_big, 1.f); });
  {
    const unsigned gb = (unsigned)((n4_big + 2047) / 2048);
    bench("calib: fill 32KiB/WG plain", out_bytes, iters, st, [&] { hipLaunchKernelGGL((k_fill_flavour<0>), dim3(gb), dim3(256), 0, st, (f4 *)big, n4_big, 1.f); });
    bench("calib: fill 32KiB/WG nt", out_bytes, iters, st, [&] { hipLaunchKernelGGL((k_fill_flavour<1>), dim3(gb), dim3(256), 0, st, (f4 *)big, n4_big, 1.f); });
    bench("calib: fill 32KiB/WG sc1", out_bytes, iters, st, [&] { hipLaunchKernelGGL((k_fill_flavour<2>), dim3(gb), dim3(256), 0, st, (f4 *)big, n4_big, 1.f); });
    bench("calib: fill 32KiB/WG sc0 sc1", out_bytes, iters, st, [&] { hipLaunchKernelGGL((k_fill_flavour<3>), dim3(gb), dim3(256), 0, st, (f4 *)big, n4_big, 1.f); });
    bench("calib: fill 32KiB/WG sc1 nt", out_bytes, iters, st, [&] { hipLaunchKernelGGL((k_fill_flavour<4>), dim3(gb), dim3(256), 0, st, (f4 *)big, n4_big, 1.f); });
    bench("calib: fill 32KiB/WG sc0 sc1 nt", out_bytes, iters, st, [&] { hipLaunchKernelGGL((k_fill_flavour<5>), dim3(gb), dim3(256), 0, st, (f4 *)big, n4_big, 1.f); });
  }
  bench("calib: read-sum (read-only)", out_bytes, iters, st,
        [&] { hipLaunchKernelGGL(k_readsum, dim3(2048), dim3(256), 0, st, (const f4 *)big2, loss, n4_big); });
  bench("calib: copy (read+write)", 2 * out_bytes, iters, st,
        [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, st, (const f4 *)big2, (f4 *)big, n4_big); });
  // what can a copy reach on this box?  (grid size, loads in flight per lane, temporal hints)
  bench("calib: copy x4 in flight, NT", 2 * out_bytes, iters, st, [&] {
    hipLaunchKernelGGL((k_copy_u<4, true>), dim3(4096), dim3(256), 0, st, (const f4 *)big2, (f4 *)big, n4_big); });
  bench("calib: copy x4 in flight, plain", 2 * out_bytes, iters, st, [&] {
    hipLaunchKernelGGL((k_copy_u<4, false>), dim3(4096), dim3(256), 0, st, (const f4 *)big2, (f4 *)big, n4_big); });
  bench("calib: copy x8 in flight, NT", 2 * out_bytes, iters, st, [&] {
    hipLaunchKernelGGL((k_copy_u<8, true>), dim3(2048), dim3(256), 0, st, (const f4 *)big2, (f4 *)big, n4_big); });
  bench("calib: copy x8, NT, 1 WG per 32 KiB", 2 * out_bytes, iters, st, [&] {
    hipLaunchKernelGGL((k_copy_u<8, true>), dim3((unsigned)((n4_big + 2047) / 2048)), dim3(256), 0, st,
                       (const f4 *)big2, (f4 *)big, n4_big); });

  // ---- a-2
  bench("dp_sumsq_partials", (double)B * P * 28, iters, st,
        [&] { DP(dp_sumsq_partials(mask, pattern, x, B, P, partials, st)); });
  bench("dp_blend", (double)B * P * 40, iters, st,
        [&] { DP(dp_blend(mask, pattern, x, partials, 4.f, B, P, 1, adv, scale, l2, st)); });
  // ---- a-4 forward / backward
  bench("dp_apply_fwd (default variant)", out_bytes + (double)B * img, iters, st,
        [&] { DP(dp_apply_fwd(adv, d_table, 2, d_idx, nullptr, S, B, S, H, W, &norm, big, st)); });
  g_apply_order = 1;
  bench("dp_apply_fwd (XCD-aware walk, A/B)", out_bytes + (double)B * img, iters, st,
        [&] { DP(dp_apply_fwd(adv, d_table, 2, d_idx, nullptr, S, B, S, H, W, &norm, big, st)); });
  g_apply_order = 0;
  for (int variant : {1, 2, 4, 9, 10, 12, 16 + 4, 16 + 7, 16 + 8 + 4, 16 + 8 + 7, 32 + 9, 32 + 10, 32 + 1, 64 + 9, 64 + 10}) {
    char name[64];
    snprintf(name, sizeof name, "  k_apply_fwd%s<G=%d,NT=%d>%s", (variant & 16) ? "_ch" : "", variant & 7, (variant >> 3) & 1,
             (variant & 32) ? " 1 sample/WG" : (variant & 64) ? " 4 samples/WG" : "");
    bench(name, out_bytes + (double)B * img, iters, st, [&] {
      DP(launch_apply_fwd(variant, adv, d_table, 2, d_idx, nullptr, S, B, S, H, W, &norm, big, st));
    });
  }
  if (!g_filter || strstr("k_apply_fwd check", g_filter) || strstr(g_filter, "apply_fwd")) {
    // every variant must produce the default variant's bytes
    int *d_diff = (int *)dmalloc(4);
    DP(dp_apply_fwd(adv, d_table, 2, d_idx, nullptr, S, B, S, H, W, &norm, big, st));
    for (int variant : {2, 12, 16 + 4, 16 + 7, 16 + 8 + 4, 16 + 8 + 7}) {
      CK(hipMemsetAsync(d_diff, 0, 4, st));
      DP(launch_apply_fwd(variant, adv, d_table, 2, d_idx, nullptr, S, B, S, H, W, &norm, big2, st));
      hipLaunchKernelGGL(k_count_diff, dim3(2048), dim3(256), 0, st, (const uint32_t *)big, (const uint32_t *)big2,
                         (size_t)N * 3 * P, d_diff);
      int h_diff = -1;
      CK(hipMemcpyAsync(&h_diff, d_diff, 4, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      printf("  k_apply_fwd check: variant %2d vs default: %d differing words\n", variant, h_diff);
    }
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (f4 *)big2, n4_big, 0.25f);
  }
  bench("dp_apply_bwd (+dp_sum_slabs)", out_bytes + (double)B * img, iters, st, [&] {
    DP(dp_apply_bwd(big2, d_table, 2, d_idx, nullptr, S, B, S, H, W, &norm, nslab == 1 ? g_adv : slabs, st));
    if (nslab > 1) DP(dp_sum_slabs(slabs, nslab, (int64_t)B * 3 * P, g_adv, 0, st));
  });
  {  // the shape HotLoop uses per micro-batch: 8 images x 32 masks
    const int Bm = std::min(B, 8), ns = dp_apply_bwd_nslab(Bm, S, P);
    float *sl = (float *)dmalloc((size_t)ns * Bm * img);
    bench("dp_apply_bwd micro-batch 8x32", (double)Bm * S * img + (double)Bm * img, iters, st, [&] {
      DP(dp_apply_bwd(big2, d_table, 2, d_idx, nullptr, S, Bm, S, H, W, &norm, ns == 1 ? g_adv : sl, st));
      if (ns > 1) DP(dp_sum_slabs(sl, ns, (int64_t)Bm * 3 * P, g_adv, 0, st));
    });
  }
  // ---- EXTENSION: per-sample affine placement fused with the occlusion apply (dorpatch_amd/placement.py RandomAffine
  // defaults: rotation +-10 deg, scale 0.9..1.1, translation +-8 px about the image centre; theta = output -> source)
  if (!g_filter || strstr(g_filter, "affine")) {
    std::vector<float> th((size_t)N * 6), thi((size_t)N * 6);
    std::uniform_real_distribution<double> Ur(-10.0, 10.0), Us(0.9, 1.1), Ut(-8.0, 8.0);
    const double cx = (W - 1) / 2.0, cy = (H - 1) / 2.0;
    for (int n = 0; n < N; ++n) {
      const double rot = Ur(rng) * M_PI / 180.0, sc = Us(rng), tx = Ut(rng), ty = Ut(rng);
      const double c = std::cos(rot) / sc, s_ = std::sin(rot) / sc;
      const double a00 = c, a01 = s_, a10 = -s_, a11 = c;
      const double t0 = cx - (a00 * (cx + tx) + a01 * (cy + ty)), t1 = cy - (a10 * (cx + tx) + a11 * (cy + ty));
      const double det = a00 * a11 - a01 * a10;
      const double i00 = a11 / det, i01 = -a01 / det, i10 = -a10 / det, i11 = a00 / det;
      float *t = &th[(size_t)n * 6], *ti = &thi[(size_t)n * 6];
      t[0] = (float)a00; t[1] = (float)a01; t[2] = (float)t0; t[3] = (float)a10; t[4] = (float)a11; t[5] = (float)t1;
      ti[0] = (float)i00; ti[1] = (float)i01; ti[2] = (float)-(i00 * t0 + i01 * t1);
      ti[3] = (float)i10; ti[4] = (float)i11; ti[5] = (float)-(i10 * t0 + i11 * t1);
    }
    float *d_th = (float *)dmalloc(th.size() * 4), *d_thi = (float *)dmalloc(thi.size() * 4);
    CK(hipMemcpy(d_th, th.data(), th.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_thi, thi.data(), thi.size() * 4, hipMemcpyHostToDevice));
    // algorithmic bytes: the same 3*P*4 B per EOT sample as the identity-placement kernels (+ x and delta / + the result once)
    bench("dp_apply_affine_fwd", out_bytes + 2.0 * B * img, iters, st, [&] {
      DP(dp_apply_affine_fwd(x, g_adv, d_th, d_table, 2, d_idx, nullptr, S, B, S, H, W, &norm, big, st));
    });
    for (int spb : {1, 2, 4, 8, 16, 32}) {   // samples per forward workgroup (0 = the launcher's own choice, above)
      if (spb > S) break;
      g_aff_samples_per_block = spb;
      char name[64];
      snprintf(name, sizeof name, "dp_apply_affine_fwd %2d samples/workgroup", spb);
      bench(name, out_bytes + 2.0 * B * img, iters, st, [&] {
        DP(dp_apply_affine_fwd(x, g_adv, d_th, d_table, 2, d_idx, nullptr, S, B, S, H, W, &norm, big, st));
      });
    }
    g_aff_samples_per_block = 0;
    bench("dp_apply_affine_bwd (+dp_sum_slabs)", out_bytes + (double)B * img, iters, st, [&] {
      DP(dp_apply_affine_bwd(big2, d_th, d_thi, d_table, 2, d_idx, nullptr, S, B, S, H, W, &norm, nslab == 1 ? g_adv : slabs, st));
      if (nslab > 1) DP(dp_sum_slabs(slabs, nslab, (int64_t)B * 3 * P, g_adv, 0, st));
    });
    g_aff_gather = 1;
    bench("dp_apply_affine_bwd, branch-free gather (A/B)", out_bytes + (double)B * img, iters, st, [&] {
      DP(dp_apply_affine_bwd(big2, d_th, d_thi, d_table, 2, d_idx, nullptr, S, B, S, H, W, &norm, nslab == 1 ? g_adv : slabs, st));
      if (nslab > 1) DP(dp_sum_slabs(slabs, nslab, (int64_t)B * 3 * P, g_adv, 0, st));
    });
    g_aff_gather = 2;
    bench("dp_apply_affine_bwd, hit-compaction gather (round 5)", out_bytes + (double)B * img, iters, st, [&] {
      DP(dp_apply_affine_bwd(big2, d_th, d_thi, d_table, 2, d_idx, nullptr, S, B, S, H, W, &norm, nslab == 1 ? g_adv : slabs, st));
      if (nslab > 1) DP(dp_sum_slabs(slabs, nslab, (int64_t)B * 3 * P, g_adv, 0, st));
    });
    g_aff_gather = 0;
    g_aff_bwd_cap = 2048;
    bench("dp_apply_affine_bwd, 48 KiB region (3 workgroups per CU)", out_bytes + (double)B * img, iters, st, [&] {
      DP(dp_apply_affine_bwd(big2, d_th, d_thi, d_table, 2, d_idx, nullptr, S, B, S, H, W, &norm, nslab == 1 ? g_adv : slabs, st));
      if (nslab > 1) DP(dp_sum_slabs(slabs, nslab, (int64_t)B * 3 * P, g_adv, 0, st));
    });
    g_aff_bwd_cap = 0;
    {  // the shape HotLoop uses per micro-batch: 8 images x 32 masks
      const int Bm = std::min(B, 8), ns = dp_apply_bwd_nslab(Bm, S, P);
      float *sl = (float *)dmalloc((size_t)ns * Bm * img);
      bench("dp_apply_affine_bwd micro-batch 8x32", (double)Bm * S * img + (double)Bm * img, iters, st, [&] {
        DP(dp_apply_affine_bwd(big2, d_th, d_thi, d_table, 2, d_idx, nullptr, S, Bm, S, H, W, &norm, ns == 1 ? g_adv : sl, st));
        if (ns > 1) DP(dp_sum_slabs(sl, ns, (int64_t)Bm * 3 * P, g_adv, 0, st));
      });
    }
  }
  // ---- a-7
  bench("dp_cw_loss (+grad, pred)", (double)N * C * 8, iters, st,
        [&] { DP(dp_cw_loss(logits, y, tflag, N, C, S, 0.1f, 1.f / S, loss, dlogits, pred, st)); });
  // ---- a-5 / a-6
  bench("dp_local_variance", (double)B * P * 16, iters, st, [&] { DP(dp_local_variance(x, B, H, W, lv, st)); });
  bench("dp_struct_loss (+reduce_rows)", (double)B * P * 16, iters, st, [&] {
    DP(dp_struct_loss(adv, lv, B, H, W, tpart, st));
    DP(dp_reduce_rows(tpart, B, ntile, 1.f / P, sloss, st));
  });
  bench("dp_mask_stats", (double)B * P * 4, iters, st,
        [&] { DP(dp_mask_stats(mask, B, H, W, unit, win, cell, wsum, gl, dens, st)); });
  // ---- a-2 bwd + a-5/a-6 grads + a-9
  bench("dp_project_update calib: bare 6-read / 2-RMW stream", (double)B * P * 72.0, iters, st, [&] {
    hipLaunchKernelGGL(k_calib_update_pattern, dim3(cdiv(P, 1024), B), dim3(256), 0, st, x, adv, g_adv, lv, pattern, mask, P);
  });
  for (int stage = 0; stage < 2; ++stage) {
    dp_update_cfg_t cfg = {B, H, W, stage, unit, win, 1, 1e-3f, 0.f, 1.f};
    // x 12 + adv_x 12 + lv 4 + g_adv 12 + pattern 12 r + 12 w + mask 4 r (+ 4 w in stage 0)
    const double bpp = stage == 0 ? 72.0 : 68.0;
    bench(stage == 0 ? "dp_project_update stage 0" : "dp_project_update stage 1", (double)B * P * bpp, iters, st,
          [&] {
            DP(dp_project_update(&cfg, x, adv, lv, g_adv, scale, structured, coeff, lr, cell, wsum, nullptr,
                                 pattern, mask, best_p, best_m, nullptr, nullptr, st));
          });
    g_update_variant = 1;
    bench(stage == 0 ? "dp_project_update stage 0, 4-byte lanes (ABI 7)" : "dp_project_update stage 1, 4-byte lanes (ABI 7)",
          (double)B * P * bpp, iters, st, [&] {
            DP(dp_project_update(&cfg, x, adv, lv, g_adv, scale, structured, coeff, lr, cell, wsum, nullptr,
                                 pattern, mask, best_p, best_m, nullptr, nullptr, st));
          });
    g_update_variant = 0;
  }
  // ---- a-8: backbone element-wise kernels at the training micro-batch (256 samples)
  if (H == 224 && (!g_filter || strstr(g_filter, "gn_relu") || strstr(g_filter, "maxpool") || strstr(g_filter, "stem") || strstr(g_filter, "pool"))) {
    const int Nb = 256;
    struct Shape { int C, HW; const char *what; };
    const Shape shapes[] = {{64, 3136, "64ch@56x56"}, {256, 3136, "256ch@56x56"}, {512, 784, "512ch@28x28"},
                            {1024, 196, "1024ch@14x14"}, {2048, 49, "2048ch@7x7"}};
    const size_t maxe = (size_t)Nb * 256 * 3136;
    float *gx = (float *)dmalloc(maxe * 4), *gy = (float *)dmalloc(maxe * 4), *gr = (float *)dmalloc(maxe * 4);
    float *gs = (float *)dmalloc(maxe * 4);
    float *gam = (float *)dmalloc(2048 * 4), *bet = (float *)dmalloc(2048 * 4);
    float *gmean = (float *)dmalloc(Nb * 32 * 4), *grstd = (float *)dmalloc(Nb * 32 * 4);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (f4 *)gx, maxe / 4, 0.37f);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (f4 *)gr, maxe / 4, -0.11f);
    hipLaunchKernelGGL(k_fill, dim3(8), dim3(256), 0, st, (f4 *)gam, 512, 1.f);
    hipLaunchKernelGGL(k_fill, dim3(8), dim3(256), 0, st, (f4 *)bet, 512, 0.05f);
    for (const Shape &sh : shapes) {
      const double e = (double)Nb * sh.C * sh.HW;
      char name[96];
      snprintf(name, sizeof name, "dp_gn_relu_fwd %s", sh.what);
      bench(name, e * 8, iters, st,
            [&] { DP(dp_gn_relu_fwd(gx, nullptr, nullptr, gam, bet, Nb, sh.C, sh.HW, 32, 1e-5f, gy, gmean, grstd, st)); });
      snprintf(name, sizeof name, "dp_gn_relu_fwd +res %s", sh.what);
      bench(name, e * 16, iters, st,
            [&] { DP(dp_gn_relu_fwd(gx, gr, gs, gam, bet, Nb, sh.C, sh.HW, 32, 1e-5f, gy, gmean, grstd, st)); });
      snprintf(name, sizeof name, "dp_gn_relu_bwd %s", sh.what);
      bench(name, e * 12, iters, st,
            [&] { DP(dp_gn_relu_bwd(gy, nullptr, gx, gam, bet, gmean, grstd, Nb, sh.C, sh.HW, 32, gs, st)); });
      snprintf(name, sizeof name, "dp_gn_relu_bwd +dres %s", sh.what);
      bench(name, e * 16, iters, st,
            [&] { DP(dp_gn_relu_bwd(gy, gr, gx, gam, bet, gmean, grstd, Nb, sh.C, sh.HW, 32, gs, st)); });
    }
    // variant sweep of the register-resident GroupNorm kernels (bits: 1 NT, 2 LDS coefficients, 4 fwd V=7 @ 8 waves)
    if (g_filter && strstr(g_filter, "gn_relu")) {
      for (const Shape &sh : shapes) {
        const double e = (double)Nb * sh.C * sh.HW;
        for (int variant = 0; variant < 8; ++variant) {
          if ((variant & 4) && !(sh.C == 256 && sh.HW == 3136)) continue;  // MW8 only exists for the V = 7 forward
          GnArgs A;
          DP(gn_check(gx, gam, bet, Nb, sh.C, sh.HW, 32, A, 1e-5f));
          char name[96];
          snprintf(name, sizeof name, "  gn_relu_fwd v%d %s", variant, sh.what);
          bench(name, e * 8, iters, st, [&] { DP(launch_gn_fwd(variant, A, Nb, gy, gmean, grstd, st)); });
          GnArgs Ar = A;
          Ar.res = gr;
          Ar.sum_out = gs;
          snprintf(name, sizeof name, "  gn_relu_fwd+res v%d %s", variant, sh.what);
          bench(name, e * 16, iters, st, [&] { DP(launch_gn_fwd(variant, Ar, Nb, gy, gmean, grstd, st)); });
          if (variant & 4) continue;
          snprintf(name, sizeof name, "  gn_relu_bwd v%d %s", variant, sh.what);
          bench(name, e * 12, iters, st, [&] { DP(launch_gn_bwd(variant, A, Nb, gy, gmean, grstd, gs, st)); });
          GnArgs Ad = A;
          Ad.dres = gr;
          snprintf(name, sizeof name, "  gn_relu_bwd+dres v%d %s", variant, sh.what);
          bench(name, e * 16, iters, st, [&] { DP(launch_gn_bwd(variant, Ad, Nb, gy, gmean, grstd, gs, st)); });
        }
      }
    }
    // stem: pad + maxpool on (256, 64, 112, 112) and the 7x7/2 input gradient
    const int64_t NC = (int64_t)Nb * 64;
    uint8_t *code = (uint8_t *)dmalloc((size_t)NC * 56 * 56);
    const double pe = (double)NC * 112 * 112;
    bench("dp_pad_maxpool_fwd 256x64x112x112", pe * 5.25, iters, st,
          [&] { DP(dp_pad_maxpool_fwd(gx, NC, 112, 112, gy, code, st)); });
    bench("dp_pad_maxpool_bwd 256x64x112x112", pe * 5.25, iters, st,
          [&] { DP(dp_pad_maxpool_bwd(gy, code, NC, 112, 112, gs, st)); });
    {  // in-situ conditions (3.6 TB/s in every step trace vs 5.8 here): the 512-sample micro-batch of the
       // headline configuration (dy + codes = 514 MB: past the 256 MiB Infinity Cache), random dy, codes from random x
      const int Nc = 512;
      const int64_t NC5 = (int64_t)Nc * 64;
      const size_t e_in = (size_t)NC5 * 112 * 112, e_out = (size_t)NC5 * 56 * 56;
      float *px = (float *)dmalloc(e_in * 4), *pdx = (float *)dmalloc(e_in * 4);
      float *py = (float *)dmalloc(e_out * 4), *pdy = (float *)dmalloc(e_out * 4);
      uint8_t *pcode = (uint8_t *)dmalloc(e_out);
      {
        std::vector<float> h(1 << 22);
        std::normal_distribution<float> Nrm(0.f, 1.f);
        for (auto &v : h) v = Nrm(rng);
        for (size_t off = 0; off < e_in; off += h.size())
          CK(hipMemcpy(px + off, h.data(), std::min(h.size(), e_in - off) * 4, hipMemcpyHostToDevice));
        for (size_t off = 0; off < e_out; off += h.size() - 4099)      // a different phase per block
          CK(hipMemcpy(pdy + off, h.data() + (off / 977) % 4099, std::min(h.size() - 4099, e_out - off) * 4, hipMemcpyHostToDevice));
      }
      const double pe5 = (double)e_in;
      bench("dp_pad_maxpool_fwd 512x64x112x112 rnd", pe5 * 5.25, iters, st,
            [&] { DP(dp_pad_maxpool_fwd(px, NC5, 112, 112, py, pcode, st)); });
      bench("dp_pad_maxpool_bwd 512x64x112x112 rnd", pe5 * 5.25, iters, st,
            [&] { DP(dp_pad_maxpool_bwd(pdy, pcode, NC5, 112, 112, pdx, st)); });
      for (int mode = 0; mode < 6; ++mode) {   // which rows a workgroup owns (see k_pad_maxpool_fwd)
        char name[64];
        snprintf(name, sizeof name, "  pad_maxpool_fwd mode %d, 512 rnd", mode);
        if (mode != 3 && mode < 5) bench(name, pe5 * 5.25, iters, st, [&] { DP(launch_pad_maxpool_fwd(mode, px, NC5, 112, 112, py, (uint32_t *)pcode, st)); });
        snprintf(name, sizeof name, "  pad_maxpool_bwd mode %d, 512 rnd", mode);
        bench(name, pe5 * 5.25, iters, st, [&] { DP(launch_pad_maxpool_bwd(mode, pdy, pcode, NC5, 112, 112, pdx, st)); });
      }
      {  // every launch mode must produce mode 1's bytes
        int *d_diff = (int *)dmalloc(4);
        float *py2 = (float *)dmalloc(e_out * 4);
        uint8_t *pcode2 = (uint8_t *)dmalloc(e_out);
        DP(launch_pad_maxpool_fwd(1, px, NC5, 112, 112, py, (uint32_t *)pcode, st));
        for (int mode : {0, 2, 4}) {
          DP(launch_pad_maxpool_fwd(mode, px, NC5, 112, 112, py2, (uint32_t *)pcode2, st));
          CK(hipMemsetAsync(d_diff, 0, 4, st));
          hipLaunchKernelGGL(k_count_diff, dim3(2048), dim3(256), 0, st, (const uint32_t *)py, (const uint32_t *)py2, e_out, d_diff);
          hipLaunchKernelGGL(k_count_diff, dim3(2048), dim3(256), 0, st, (const uint32_t *)pcode, (const uint32_t *)pcode2, e_out / 4, d_diff);
          int h_diff = -1;
          CK(hipMemcpyAsync(&h_diff, d_diff, 4, hipMemcpyDeviceToHost, st));
          CK(hipStreamSynchronize(st));
          printf("  pad_maxpool_fwd check: mode %d vs mode 1: %d differing words\n", mode, h_diff);
        }
        CK(hipFree(py2)); CK(hipFree(pcode2));
        float *pdx2 = (float *)dmalloc(e_in * 4);
        DP(launch_pad_maxpool_bwd(1, pdy, pcode, NC5, 112, 112, pdx, st));
        for (int mode : {0, 2, 3, 4, 5}) {
          DP(launch_pad_maxpool_bwd(mode, pdy, pcode, NC5, 112, 112, pdx2, st));
          CK(hipMemsetAsync(d_diff, 0, 4, st));
          hipLaunchKernelGGL(k_count_diff, dim3(2048), dim3(256), 0, st, (const uint32_t *)pdx, (const uint32_t *)pdx2, e_in, d_diff);
          int h_diff = -1;
          CK(hipMemcpyAsync(&h_diff, d_diff, 4, hipMemcpyDeviceToHost, st));
          CK(hipStreamSynchronize(st));
          printf("  pad_maxpool_bwd check: mode %d vs mode 1: %d differing words\n", mode, h_diff);
        }
        CK(hipFree(pdx2));
      }
      CK(hipFree(px)); CK(hipFree(pdx)); CK(hipFree(py)); CK(hipFree(pdy)); CK(hipFree(pcode));
    }
    float *wst = (float *)dmalloc(64 * 147 * 4);
    hipLaunchKernelGGL(k_fill, dim3(8), dim3(256), 0, st, (f4 *)wst, 64 * 147 / 4, 0.01f);
    {  // fused stem dgrad + occlusion-masked S-reduction: 8 images x 32 samples
      const int Bf = 8, ns = dp_apply_bwd_nslab(Bf, 32, 224 * 224);
      float *sl = (float *)dmalloc((size_t)ns * Bf * 3 * 224 * 224 * 4);
      const double flopf = 2.0 * 64 * 147 * 112 * 112 * Bf * 32;
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      auto run = [&] {
        DP(dp_stem_dgrad_reduce(gx, wst, d_table, 2, d_idx, nullptr, 32, Bf, 32, 64, 112, 112, &norm, sl, st));
        if (ns > 1) DP(dp_sum_slabs(sl, ns, (int64_t)Bf * 3 * 224 * 224, g_adv, 0, st));
      };
      for (int i = 0; i < 2; ++i) run();
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) run();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      printf("%-34s %9.4f ms  %12.3e flop  %8.2f TFLOP/s  %5.1f%% of 157.3 TF fp32  (%d slabs)\n",
             "dp_stem_dgrad_reduce 8x32 samples", ms, flopf, flopf / (ms * 1e-3) / 1e12, flopf / (ms * 1e-3) / 1e12 / 1.573, ns);
    }
    {
      const double flop = 2.0 * 64 * 147 * 112 * 112 * Nb;  // 60.4 GFLOP
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int i = 0; i < 2; ++i) DP(dp_stem_dgrad(gx, wst, Nb, 64, 112, 112, gy, st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) DP(dp_stem_dgrad(gx, wst, Nb, 64, 112, 112, gy, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      printf("%-34s %9.4f ms  %12.3e flop  %8.2f TFLOP/s  %5.1f%% of 157.3 TF fp32\n", "dp_stem_dgrad 256x64x112x112", ms,
             flop, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 1.573);
      for (int variant : {0, 2}) {   // quads per thread: 2 (round 2) / 4
        for (int i = 0; i < 2; ++i) DP(launch_stem_dgrad(variant, gx, wst, Nb, 64, 112, 112, gs, st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) DP(launch_stem_dgrad(variant, gx, wst, Nb, 64, 112, 112, gs, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        printf("  stem_dgrad VALU, %d quads/thread    %9.4f ms  %8.2f TFLOP/s  %5.1f%% of 157.3 TF fp32\n", variant == 0 ? 2 : 4, ms,
               flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 1.573);
      }
      // the matrix-core formulation (k_stem_dgrad_mfma): same useful flops, 57 % of the issued MACs useful
      std::vector<float> hw(64 * 147);
      std::normal_distribution<float> Nw(0.f, 0.1f);
      for (auto &v : hw) v = Nw(rng);
      CK(hipMemcpy(wst, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
      for (int i = 0; i < 2; ++i) DP(launch_stem_dgrad(1, gx, wst, Nb, 64, 112, 112, gs, st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) DP(launch_stem_dgrad(1, gx, wst, Nb, 64, 112, 112, gs, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      printf("%-34s %9.4f ms  %12.3e flop  %8.2f TFLOP/s  %5.1f%% of 157.3 TF fp32 (useful flops)\n", "  stem_dgrad MFMA 16x16x4 f32", ms,
             flop, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 1.573);
      DP(launch_stem_dgrad(0, gx, wst, Nb, 64, 112, 112, gy, st));
      unsigned *d_md = (unsigned *)dmalloc(8);
      CK(hipMemsetAsync(d_md, 0, 8, st));
      hipLaunchKernelGGL(k_max_diff, dim3(2048), dim3(256), 0, st, gs, gy, (size_t)Nb * 3 * 224 * 224, d_md);
      unsigned h_md[2];
      CK(hipMemcpyAsync(h_md, d_md, 8, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      float fd, fm;
      memcpy(&fd, &h_md[0], 4);
      memcpy(&fm, &h_md[1], 4);
      printf("  stem_dgrad MFMA vs VALU: max |diff| %.3e of max |value| %.3e (%.2e relative)\n", fd, fm, fd / fm);
      DP(launch_stem_dgrad(2, gx, wst, Nb, 64, 112, 112, gs, st));
      CK(hipMemsetAsync(d_md, 0, 8, st));
      hipLaunchKernelGGL(k_max_diff, dim3(2048), dim3(256), 0, st, gs, gy, (size_t)Nb * 3 * 224 * 224, d_md);
      CK(hipMemcpyAsync(h_md, d_md, 8, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      memcpy(&fd, &h_md[0], 4);
      printf("  stem_dgrad 4 quads/thread vs 2: max |diff| %.3e (same summation order per output: 0 expected)\n", fd);
    }
  }
  // ---- a-8 at 384 x 384 (BASELINE configs[2], 64 samples): the large-group GroupNorm backward — register / LDS resident
  // (k_gn_relu_bwd_big) vs the round-2 streaming kernel (variant bit 8)
  if (H == 384 && (!g_filter || strstr(g_filter, "gn_relu"))) {
    const int Nb = 64;
    struct Shape { int C, HW; const char *what; };
    const Shape shapes[] = {{256, 9216, "256ch@96x96 (V=18)"}, {128, 9216, "128ch@96x96 (V=9)"}, {512, 2304, "512ch@48x48 (V=9)"}};
    const size_t maxe = (size_t)Nb * 256 * 9216;
    float *gx = (float *)dmalloc(maxe * 4), *gy = (float *)dmalloc(maxe * 4), *gr = (float *)dmalloc(maxe * 4);
    float *gs = (float *)dmalloc(maxe * 4);
    float *gam = (float *)dmalloc(2048 * 4), *bet = (float *)dmalloc(2048 * 4);
    float *gmean = (float *)dmalloc(Nb * 32 * 4), *grstd = (float *)dmalloc(Nb * 32 * 4);
    {
      std::vector<float> h(1 << 22);
      std::normal_distribution<float> Nrm(0.3f, 1.5f);
      for (auto &v : h) v = Nrm(rng);
      for (float *dst : {gx, gy, gr})
        for (size_t off = 0; off < maxe; off += h.size())
          CK(hipMemcpy(dst + off, h.data(), std::min(h.size(), maxe - off) * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_fill, dim3(8), dim3(256), 0, st, (f4 *)gam, 512, 1.f);
    hipLaunchKernelGGL(k_fill, dim3(8), dim3(256), 0, st, (f4 *)bet, 512, 0.05f);
    for (const Shape &sh : shapes) {
      const double e = (double)Nb * sh.C * sh.HW;
      GnArgs A;
      DP(gn_check(gx, gam, bet, Nb, sh.C, sh.HW, 32, A, 1e-5f));
      DP(launch_gn_fwd(kGnDefaultVariant, A, Nb, gs, gmean, grstd, st));       // real statistics for the backward
      char name[96];
      for (int big = 2; big >= 0; --big) {
        const int variant = kGnDefaultVariant | (big == 0 ? kGnStreamLarge : big == 2 ? kGnBigBatch : 0);
        snprintf(name, sizeof name, "gn_relu_bwd %s %s", big == 2 ? "on-chip b9" : big ? "on-chip b3" : "streaming ", sh.what);
        bench(name, e * 12, iters, st, [&] { DP(launch_gn_bwd(variant, A, Nb, gy, gmean, grstd, gs, st)); });
        GnArgs Ad = A;
        Ad.dres = gr;
        snprintf(name, sizeof name, "gn_relu_bwd+dres %s %s", big == 2 ? "on-chip b9" : big ? "on-chip b3" : "streaming ", sh.what);
        bench(name, e * 16, iters, st, [&] { DP(launch_gn_bwd(variant, Ad, Nb, gy, gmean, grstd, gs, st)); });
      }
    }
  }
  CK(hipStreamSynchronize(st));
  return 0;
}
