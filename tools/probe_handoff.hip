// Probe (round 4): what does it cost to merge the shares of a split decode-attention launch INSIDE
// the launch (the share that arrives last merges) instead of in a second launch?  Stand-alone HIP program, no torch.
//
// Shape of the real thing (attention_mfma.h): G = seqs x kv heads groups, NP shares per group, one 512-thread workgroup
// per share.  A workgroup streams `stream_kib` KiB of a big buffer with nt loads (stand-in for its K/V walk, so that the
// tail is measured where it happens: behind a memory-bound loop, on a loaded chip), reduces what it read to one
// number per thread, and leaves a partial result: NH x D 16-bit values + NH x {max, sum} floats, as the partition pass
// leaves in tmp_out / max_logits / exp_sums.  Then, per variant:
//   A  two launches: plain stores; a reduce kernel (one wave per (group, head) row, as paged_attn_v2_reduce_generic)
//   B  one launch, the guide's recipe R1: partials stored write-through (agent-scope atomic stores = sc1), every
//      storing wave drains (s_waitcnt vmcnt(0)), barrier, one lane adds to the group's counter (relaxed, agent);
//      the workgroup whose add returns NP-1 does ONE agent-scope acquire, then reads the partials with plain loads,
//      merges, writes `out`, and resets the counter
//   C  one launch: plain stores, barrier, one lane release fence (agent) + add; last arriver acquire + plain loads
//   D  as B without the acquire, every load of the partials an agent-scope atomic load (sc1)   [valid only under the
//      conditions of MI355X_MICROARCH.md's table; here to price the acquire]
//   E  as B without the acquire, the partials read by returning atomic adds of 0 (performed at the memory side)
// Output: microseconds per launch (or launch pair) in trains of 32 over rotating source buffers, and a checksum per
// variant (all equal).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

constexpr int NH = 4, D = 128, THREADS = 512;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Args {
  const u32x4* src;   // stream source, one region per workgroup
  int stream_vec;     // 16-byte vectors per thread
  uint32_t* tmp;      // [G][NP][NH*D/2] packed 16-bit pairs
  float* ml;          // [G][NH][NP]
  float* es;          // [G][NH][NP]
  uint32_t* out;      // [G][NH*D/2]
  int* counters;      // [G]
  int np;
};

__device__ __forceinline__ float stream_part(const Args& a, const int wg) {
  const u32x4* s = a.src + (size_t)wg * a.stream_vec * THREADS + threadIdx.x;
  uint32_t acc = 0;
  for (int i = 0; i < a.stream_vec; i += 4) {  // 4 loads in flight per thread
    u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_nontemporal_load(s + (size_t)(i + j) * THREADS);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  }
  return (float)(acc & 0xff) * (1.0f / 256.0f);
}

// the in-workgroup merge's stand-in: every thread owns one pair of 16-bit outputs? no: NH*D/2 = 256 words, threads
// 0..255 own one each (the real kernel has 512 two-byte stores; words keep the probe's stores 4 bytes wide, as B needs)
template <int V>
__global__ __launch_bounds__(THREADS) void share_kernel(const Args a) {
  const int g = blockIdx.x, part = blockIdx.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
  const float f = stream_part(a, wg);
  __shared__ int s_last;
  const int t = threadIdx.x;
  const uint32_t word = (uint32_t)(g * 7 + part * 3 + t) + (uint32_t)(f * 0.0f);  // value independent of the data read
  uint32_t* my_tmp = a.tmp + ((size_t)g * a.np + part) * (NH * D / 2);
  if (V == 0 || V == 2) {  // plain stores
    if (t < NH * D / 2) my_tmp[t] = word;
    if (t < NH) {
      a.ml[((size_t)g * NH + t) * a.np + part] = (float)part;
      a.es[((size_t)g * NH + t) * a.np + part] = 1.0f + part;
    }
    if (V == 0) return;
  } else {  // write-through
    if (t < NH * D / 2) __hip_atomic_store(my_tmp + t, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < NH) {
      __hip_atomic_store(a.ml + ((size_t)g * NH + t) * a.np + part, (float)part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.es + ((size_t)g * NH + t) * a.np + part, 1.0f + part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains
  __syncthreads();
  if (t == 0) {
    if (V == 2) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int old = __hip_atomic_fetch_add(a.counters + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == a.np - 1;
    if (last) {
      __hip_atomic_store(a.counters + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (V == 1 || V == 2) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  // the merge of paged_attn_v2_reduce_generic: weights from (max, sum) of every share, weighted sum of the partials
  if (t < NH * D / 2) {
    const int h = t / (D / 2);
    float M = -1e30f;
    float w[8];
    for (int j = 0; j < a.np; ++j) {
      float m;
      if (V == 3) m = __hip_atomic_load(a.ml + ((size_t)g * NH + h) * a.np + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (V == 4) m = __hip_atomic_fetch_add(a.ml + ((size_t)g * NH + h) * a.np + j, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else m = a.ml[((size_t)g * NH + h) * a.np + j];
      w[j] = m;
      M = fmaxf(M, m);
    }
    float tot = 0.f;
    for (int j = 0; j < a.np; ++j) {
      float e;
      if (V == 3) e = __hip_atomic_load(a.es + ((size_t)g * NH + h) * a.np + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (V == 4) e = __hip_atomic_fetch_add(a.es + ((size_t)g * NH + h) * a.np + j, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else e = a.es[((size_t)g * NH + h) * a.np + j];
      w[j] = e * __expf(w[j] - M);
      tot += w[j];
    }
    uint32_t acc = 0;
    for (int j = 0; j < a.np; ++j) {
      const uint32_t* p = a.tmp + ((size_t)g * a.np + j) * (NH * D / 2) + t;
      uint32_t x;
      if (V == 3) x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (V == 4) x = __hip_atomic_fetch_add((uint32_t*)p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else x = *p;
      acc += x * (uint32_t)(1 + (int)(w[j] / tot * 0.0f));  // keeps the weights alive, value = sum of the words
    }
    a.out[(size_t)g * (NH * D / 2) + t] = acc;
  }
}

__global__ void reduce_kernel(const Args a, const int G) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // (group, head)
  if (row >= G * NH) return;
  const int lane = threadIdx.x & 63, g = row / NH, h = row % NH;
  float M = -1e30f, w[8], tot = 0.f;
  for (int j = 0; j < a.np; ++j) M = fmaxf(M, a.ml[(size_t)row * a.np + j]);
  for (int j = 0; j < a.np; ++j) {
    w[j] = a.es[(size_t)row * a.np + j] * __expf(a.ml[(size_t)row * a.np + j] - M);
    tot += w[j];
  }
  uint32_t acc = 0;
  for (int j = 0; j < a.np; ++j)
    acc += a.tmp[((size_t)g * a.np + j) * (NH * D / 2) + h * (D / 2) + lane] * (uint32_t)(1 + (int)(w[j] / tot * 0.0f));
  a.out[(size_t)g * (NH * D / 2) + h * (D / 2) + lane] = acc;
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256;
  const int NP = argc > 2 ? atoi(argv[2]) : 2;
  const int stream_kib = argc > 3 ? atoi(argv[3]) : 256;
  const int COPIES = 6, TRAIN = 32, TRAINS = 6;
  if (NP > 8 || NP < 1) return 1;
  const int stream_vec = stream_kib * 1024 / 16 / THREADS;  // per thread, a multiple of 4
  const size_t src_bytes = (size_t)G * NP * stream_kib * 1024;
  std::vector<u32x4*> srcs(COPIES);
  for (auto& s : srcs) {
    CHECK(hipMalloc(&s, src_bytes));
    CHECK(hipMemset(s, 1, src_bytes));
  }
  Args a{};
  a.stream_vec = stream_vec & ~3;
  a.np = NP;
  CHECK(hipMalloc(&a.tmp, (size_t)G * NP * NH * D * 2));
  CHECK(hipMalloc(&a.ml, (size_t)G * NH * NP * 4));
  CHECK(hipMalloc(&a.es, (size_t)G * NH * NP * 4));
  CHECK(hipMalloc(&a.out, (size_t)G * NH * D * 2));
  CHECK(hipMalloc(&a.counters, (size_t)G * 4));
  CHECK(hipMemset(a.counters, 0, (size_t)G * 4));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::vector<uint32_t> host((size_t)G * NH * D / 2);
  const char* names[5] = {"A two launches (plain stores + reduce kernel)", "B one launch: write-through stores, acquire, plain loads",
                          "C one launch: plain stores, release fence, acquire", "D one launch: write-through stores, sc1 loads, NO acquire",
                          "E one launch: write-through stores, read by atomic add 0"};
  printf("groups %d  shares %d  %d KiB streamed per workgroup  (%.1f MB per launch)\n", G, NP, stream_kib, src_bytes / 1e6);
  for (int rnd = 0; rnd < 2; ++rnd)
    for (int v = 0; v < 5; ++v) {
      auto launch = [&](int i) {
        a.src = srcs[i % COPIES];
        dim3 grid(G, NP);
        switch (v) {
          case 0:
            hipLaunchKernelGGL(share_kernel<0>, grid, dim3(THREADS), 0, st, a);
            hipLaunchKernelGGL(reduce_kernel, dim3((G * NH + 3) / 4), dim3(256), 0, st, a, G);
            break;
          case 1: hipLaunchKernelGGL(share_kernel<1>, grid, dim3(THREADS), 0, st, a); break;
          case 2: hipLaunchKernelGGL(share_kernel<2>, grid, dim3(THREADS), 0, st, a); break;
          case 3: hipLaunchKernelGGL(share_kernel<3>, grid, dim3(THREADS), 0, st, a); break;
          default: hipLaunchKernelGGL(share_kernel<4>, grid, dim3(THREADS), 0, st, a); break;
        }
      };
      CHECK(hipMemsetAsync(a.out, 0, (size_t)G * NH * D * 2, st));
      for (int i = 0; i < TRAIN; ++i) launch(i);
      CHECK(hipStreamSynchronize(st));
      CHECK(hipMemcpy(host.data(), a.out, host.size() * 4, hipMemcpyDeviceToHost));
      unsigned long long sum = 0;
      for (auto x : host) sum += x;
      float best = 1e30f, tot = 0.f;
      for (int tr = 0; tr < TRAINS; ++tr) {
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < TRAIN; ++i) launch(i);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        tot += ms;
      }
      printf("round %d  %-62s %7.2f us avg  %7.2f us best   checksum %llu\n", rnd, names[v], tot / TRAINS / TRAIN * 1e3,
             best / TRAIN * 1e3, sum);
    }
  return 0;
}
