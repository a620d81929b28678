// Probe (round 4): can a loader wave decouple a CU's weight stream (HBM misses) from its
// activations (L2 hits)?  Only if the CU's vector-memory pipeline lets a later L2 hit of ONE wave return before an
// earlier HBM miss of ANOTHER wave.  Stand-alone HIP program, one 2-wave (or 8-wave) workgroup per CU.
//
//   mode 0  wave 1 alone reads `hit_kib` KiB of a small table (L2-resident, warmed by an earlier pass): time from its
//           first request to its last datum                                               -> the L2-hit baseline
//   mode 1  wave 0 first requests `miss_kib` KiB of a region no one has touched (HBM, nt loads), THEN (LDS flag) wave 1
//           reads the table as in mode 0.  In-order return per CU: wave 1's time ~ wave 0's.  Independent: ~ mode 0.
//   mode 2  the other order: wave 1's table reads first, then wave 0's misses (control: both as fast as alone)
//   mode 3  the activations of a decode GEMM: all 8 waves of the workgroup read 32 KiB each of ONE 256 KiB table
//           (what every workgroup of skinny_gemm_kernel does with X), every CU at once: the L2 -> CU rate per CU
//   mode 4  mode 3 with each wave's first 16 KiB of weights (HBM, nt) requested BEFORE its 32 KiB of the table
//           (LVLLM_GEMM_XORDER=1's order) -- time until the table part has arrived / until everything has
//   mode 5  mode 3 with the weights requested AFTER the table (the shipped order)
// Times in ns (s_memrealtime, 100 MHz), median / min / max over the workgroups, per mode.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

template <int N, bool NT>
__device__ __forceinline__ uint32_t issue_and_wait(const u32x4* p, const int stride_vec) {
  u32x4 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i)
    v[i] = NT ? __builtin_nontemporal_load(p + (size_t)i * stride_vec) : p[(size_t)i * stride_vec];
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  return acc;
}

// 2 waves.  stamps[wg][0..3] = {wave-1 start, wave-1 end, wave-0 start, wave-0 end}
template <int HIT_LOADS, int MISS_LOADS>
__global__ __launch_bounds__(128) void order_kernel(const u32x4* table, const u32x4* big, unsigned long long* stamps,
                                                    uint32_t* sink, const int mode) {
  extern __shared__ int lds[];  // (a large request keeps it to one workgroup per CU)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) lds[0] = 0;
  __syncthreads();
  uint32_t acc = 0;
  if (wave == 0) {
    if (mode == 2) {  // wait until wave 1 has issued
      while (__hip_atomic_load(&lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    }
    if (mode != 0) {
      const u32x4* p = big + (size_t)blockIdx.x * MISS_LOADS * 64 + lane;
      const unsigned long long t0 = now();
      u32x4 v[MISS_LOADS];
#pragma unroll
      for (int i = 0; i < MISS_LOADS; ++i) v[i] = __builtin_nontemporal_load(p + (size_t)i * 64);
      if (mode == 1 && lane == 0) __hip_atomic_store(&lds[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int i = 0; i < MISS_LOADS; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
      asm volatile("" ::"v"(acc));
      const unsigned long long t1 = now();
      if (lane == 0) {
        stamps[blockIdx.x * 4 + 2] = t0;
        stamps[blockIdx.x * 4 + 3] = t1;
      }
    }
  } else {
    if (mode == 1) {
      while (__hip_atomic_load(&lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    }
    const u32x4* p = table + lane;
    const unsigned long long t0 = now();
    u32x4 v[HIT_LOADS];
#pragma unroll
    for (int i = 0; i < HIT_LOADS; ++i) v[i] = p[(size_t)i * 64];
    if (mode == 2 && lane == 0) __hip_atomic_store(&lds[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int i = 0; i < HIT_LOADS; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    asm volatile("" ::"v"(acc));
    const unsigned long long t1 = now();
    if (lane == 0) {
      stamps[blockIdx.x * 4 + 0] = t0;
      stamps[blockIdx.x * 4 + 1] = t1;
    }
  }
  if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

// 8 waves: the activations of a decode GEMM.  stamps[wg][0..2] = {start, table arrived (max over waves), all arrived}
template <int mode>
__global__ __launch_bounds__(512) void xphase_kernel(const u32x4* table, const u32x4* big, unsigned long long* stamps,
                                                     uint32_t* sink) {
  extern __shared__ int lds[];
  __shared__ unsigned long long s_t[3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) { s_t[0] = ~0ull; s_t[1] = 0; s_t[2] = 0; }
  __syncthreads();
  const u32x4* x = table + (size_t)wave * 32 * 64 + lane;                       // this wave's 32 KiB of the table
  const u32x4* w = big + ((size_t)blockIdx.x * 8 + wave) * 16 * 64 + lane;      // its 16 KiB of weights
  uint32_t acc = 0;
  const unsigned long long t0 = now();
  u32x4 xv[32], wv[16];
  if constexpr (mode == 4) {
#pragma unroll
    for (int i = 0; i < 16; ++i) wv[i] = __builtin_nontemporal_load(w + (size_t)i * 64);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) xv[i] = x[(size_t)i * 64];
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (mode == 5) {
#pragma unroll
    for (int i = 0; i < 16; ++i) wv[i] = __builtin_nontemporal_load(w + (size_t)i * 64);
    __builtin_amdgcn_sched_barrier(0);
  }
  unsigned long long tx, tall;
  if constexpr (mode == 4) {  // table behind the weights: waiting for it is waiting for everything
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= wv[i].x ^ wv[i].y ^ wv[i].z ^ wv[i].w;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc ^= xv[i].x ^ xv[i].y ^ xv[i].z ^ xv[i].w;
    asm volatile("" ::"v"(acc));
    tx = tall = now();
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc ^= xv[i].x ^ xv[i].y ^ xv[i].z ^ xv[i].w;
    asm volatile("" ::"v"(acc));
    tx = now();
    if constexpr (mode == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc ^= wv[i].x ^ wv[i].y ^ wv[i].z ^ wv[i].w;
      asm volatile("" ::"v"(acc));
    }
    tall = now();
  }
  if (lane == 0) {
    atomicMin(&s_t[0], t0);
    atomicMax(&s_t[1], tx);
    atomicMax(&s_t[2], tall);
  }
  __syncthreads();
  if (threadIdx.x < 3) stamps[blockIdx.x * 4 + threadIdx.x] = s_t[threadIdx.x];
  if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

static void stats(const char* what, std::vector<double> v) {
  std::sort(v.begin(), v.end());
  printf("    %-44s median %7.0f ns   min %7.0f   max %7.0f\n", what, v[v.size() / 2], v.front(), v.back());
}

int main() {
  const int WGS = 256;
  const size_t big_bytes = (size_t)3 << 30;  // every launch reads a fresh slice: nothing is in a cache
  u32x4 *table, *big;
  unsigned long long* stamps;
  uint32_t* sink;
  CHECK(hipMalloc(&table, 1 << 20));
  CHECK(hipMemset(table, 1, 1 << 20));
  CHECK(hipMalloc(&big, big_bytes));
  CHECK(hipMemset(big, 2, big_bytes));
  CHECK(hipMalloc(&stamps, WGS * 4 * 8));
  CHECK(hipMalloc(&sink, 4096));
  std::vector<unsigned long long> h(WGS * 4);
  const int lds_bytes = 100 * 1024;
  CHECK(hipFuncSetAttribute((const void*)order_kernel<16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  CHECK(hipFuncSetAttribute((const void*)order_kernel<16, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  CHECK(hipFuncSetAttribute((const void*)xphase_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  CHECK(hipFuncSetAttribute((const void*)xphase_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  CHECK(hipFuncSetAttribute((const void*)xphase_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  size_t slice = 0;  // in 16-byte vectors
  auto next_big = [&](size_t vecs) {
    if ((slice + vecs) * 16 > big_bytes) slice = 0;
    const u32x4* p = big + slice;
    slice += vecs;
    return p;
  };
  printf("one workgroup per CU (%d workgroups); times in ns\n", WGS);
  for (int rnd = 0; rnd < 3; ++rnd) {
    printf("round %d\n", rnd);
    for (int miss = 16; miss <= 32; miss *= 2) {
      for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {  // (the first pass of a mode warms the table into this XCD's L2)
          const u32x4* b = next_big((size_t)WGS * miss * 64);
          if (miss == 16)
            hipLaunchKernelGGL((order_kernel<16, 16>), dim3(WGS), dim3(128), lds_bytes, 0, table, b, stamps, sink, mode);
          else
            hipLaunchKernelGGL((order_kernel<16, 32>), dim3(WGS), dim3(128), lds_bytes, 0, table, b, stamps, sink, mode);
          CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> hit, mis;
        for (int i = 0; i < WGS; ++i) {
          hit.push_back((double)(h[i * 4 + 1] - h[i * 4 + 0]) * 10.0);
          if (mode != 0) mis.push_back((double)(h[i * 4 + 3] - h[i * 4 + 2]) * 10.0);
        }
        printf("  mode %d (%s), wave 0 misses %d KiB, wave 1 hits 16 KiB\n", mode,
               mode == 0 ? "hits alone" : mode == 1 ? "misses requested first" : "hits requested first", miss);
        stats("wave 1: 16 KiB of L2 hits", hit);
        if (mode != 0) stats("wave 0: its HBM misses", mis);
      }
    }
    for (int mode = 3; mode <= 5; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        const u32x4* b = next_big((size_t)WGS * 8 * 16 * 64);
        if (mode == 3) hipLaunchKernelGGL(xphase_kernel<3>, dim3(WGS), dim3(512), lds_bytes, 0, table, b, stamps, sink);
        else if (mode == 4) hipLaunchKernelGGL(xphase_kernel<4>, dim3(WGS), dim3(512), lds_bytes, 0, table, b, stamps, sink);
        else hipLaunchKernelGGL(xphase_kernel<5>, dim3(WGS), dim3(512), lds_bytes, 0, table, b, stamps, sink);
        CHECK(hipDeviceSynchronize());
      }
      CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> tx, tall;
      for (int i = 0; i < WGS; ++i) {
        tx.push_back((double)(h[i * 4 + 1] - h[i * 4 + 0]) * 10.0);
        tall.push_back((double)(h[i * 4 + 2] - h[i * 4 + 0]) * 10.0);
      }
      printf("  mode %d (%s): 8 waves x 32 KiB of one 256 KiB table per workgroup\n", mode,
             mode == 3 ? "table only" : mode == 4 ? "16 KiB of weights per wave requested BEFORE the table"
                                                  : "16 KiB of weights per wave requested AFTER the table");
      stats("table (256 KiB) in registers", tx);
      if (mode != 3) stats("table and weights (128 KiB) in registers", tall);
    }
  }
  return 0;
}
