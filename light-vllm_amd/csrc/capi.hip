// Error reporting, version string and device queries of the C-ABI (include/lvllm_hip.h).
#include "common.h"

namespace lvllm {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace lvllm

extern "C" const char* lvllm_last_error(void) { return lvllm::g_last_error.c_str(); }

extern "C" const char* lvllm_version(void) { return "lvllm_hip gfx950 abi1"; }

// csrc/cuda_utils_kernels.cu:1-35 of the reference: thin wrappers over the
// runtime's attribute query, kept so that torch.ops._C_cuda_utils.* exists.
extern "C" int64_t lvllm_get_device_attribute(int64_t attribute, int64_t device_id) {
  int device = (int)device_id, value = 0;
  if (device < 0) (void)hipGetDevice(&device);
  if (hipDeviceGetAttribute(&value, (hipDeviceAttribute_t)attribute, device) != hipSuccess) {
    lvllm::set_error("lvllm_get_device_attribute: hipDeviceGetAttribute failed");
    return -1;
  }
  return value;
}

extern "C" int64_t lvllm_get_max_shared_memory_per_block_device_attribute(int64_t device_id) {
  // gfx950: 160 KiB of LDS per workgroup (the reference asks for the opt-in maximum)
  return lvllm_get_device_attribute((int64_t)hipDeviceAttributeMaxSharedMemoryPerBlock, device_id);
}
