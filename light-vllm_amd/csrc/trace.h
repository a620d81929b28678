// Diagnosis builds only (-DLVLLM_TRACE, tools/build_variant.sh): every workgroup of an instrumented
// kernel appends (start, end, kernel id | grid | block) to a ring in device memory, 100 MHz
// wall clock.  One ring per translation unit; tools/trace_step.py reads them through the
// lvllm_trace_read_* entry points and rebuilds the timeline of a decode step across streams --
// which the profiler cannot show, because its kernel trace serialises the streams.
#pragma once
#ifdef LVLLM_TRACE
#include <hip/hip_runtime.h>
namespace lvllm {
constexpr unsigned kTraceRecords = 1u << 20;
struct TraceRing {
  unsigned long long rec[3 * kTraceRecords];
  unsigned int head;
};
static __device__ TraceRing g_trace;
__device__ __forceinline__ unsigned long long trace_begin() { return wall_clock64(); }
__device__ __forceinline__ void trace_end(const int kid, const unsigned long long t0) {
  if (threadIdx.x == 0) {
    const unsigned i = atomicAdd(&g_trace.head, 1u) & (kTraceRecords - 1);
    g_trace.rec[3 * i] = t0;
    g_trace.rec[3 * i + 1] = wall_clock64();
    g_trace.rec[3 * i + 2] = ((unsigned long long)kid << 48) | ((unsigned long long)(gridDim.x * gridDim.y) << 24) |
                             (blockIdx.y * gridDim.x + blockIdx.x);
  }
}
}  // namespace lvllm
#define LVLLM_TRACE_BEGIN() const unsigned long long lv_trace_t0 = lvllm::trace_begin()
#define LVLLM_TRACE_END(kid) lvllm::trace_end((kid), lv_trace_t0)
#define LVLLM_TRACE_READER(name)                                                                      \
  extern "C" int name(void* host_dst, unsigned* head_out) {                                           \
    lvllm::TraceRing* d = nullptr;                                                                    \
    if (hipGetSymbolAddress((void**)&d, HIP_SYMBOL(lvllm::g_trace)) != hipSuccess) return 1;          \
    if (hipMemcpy(head_out, &d->head, 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;              \
    return (int)hipMemcpy(host_dst, d->rec, sizeof(d->rec), hipMemcpyDeviceToHost);                   \
  }
#else
#define LVLLM_TRACE_BEGIN()
#define LVLLM_TRACE_END(kid)
#define LVLLM_TRACE_READER(name)
#endif
