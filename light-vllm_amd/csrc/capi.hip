// Error reporting, version string and device queries of the C-ABI (include/lvllm_hip.h).
#include <stdlib.h>

#include "common.h"

namespace lvllm {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
Tuning& tuning() {
  static Tuning t = [] {
    Tuning v;
    if (const char* e = getenv("LVLLM_GEMM_CUS")) v.gemm_workgroups = atoi(e);
    if (const char* e = getenv("LVLLM_GEMM_CUS_WIDE")) v.gemm_workgroups_wide = atoi(e);
    if (const char* e = getenv("LVLLM_GEMM_WIDE_MIN_TILES")) v.gemm_wide_min_tiles = atoi(e);
    if (const char* e = getenv("LVLLM_ATTN_WAVES")) v.attn_waves = atoi(e);
    if (const char* e = getenv("LVLLM_ATTN_SPLITS")) v.attn_splits = atoi(e);  // read once, at load
    if (const char* e = getenv("LVLLM_PREFILL_LDS")) v.prefill_lds = atoi(e);
    if (const char* e = getenv("LVLLM_VARLEN_DENSE")) v.varlen_dense = atoi(e) != 0;
    if (const char* e = getenv("LVLLM_VARLEN_DENSE_WAVES")) v.varlen_dense_waves = atoi(e) == 4 ? 4 : atoi(e) == 8 ? 8 : 0;
    if (const char* e = getenv("LVLLM_PREFILL_MFMA32_MIN_QUERY")) v.prefill_mfma32_min_query = atoi(e);
    if (const char* e = getenv("LVLLM_PREFILL_CHUNK_MAX_QUERY")) v.prefill_chunk_max_query = atoi(e);
    return v;
  }();
  return t;
}
}  // namespace lvllm

extern "C" const char* lvllm_last_error(void) { return lvllm::g_last_error.c_str(); }

extern "C" int lvllm_set_tuning(const char* key, int value) {
  const std::string k = key ? key : "";
  if (k == "gemm_workgroups") {
    LV_CHECK(value >= 8 && value <= 1024, "gemm_workgroups must be in [8, 1024]");
    lvllm::tuning().gemm_workgroups = value;
  } else if (k == "gemm_workgroups_wide") {
    LV_CHECK(value >= 0 && value <= 1024, "gemm_workgroups_wide must be in [0, 1024]");
    lvllm::tuning().gemm_workgroups_wide = value;
  } else if (k == "gemm_wide_min_tiles") {
    LV_CHECK(value >= 1, "gemm_wide_min_tiles must be positive");
    lvllm::tuning().gemm_wide_min_tiles = value;
  } else if (k == "gemm_partials_ksplit") {
    LV_CHECK(value >= 0 && value <= 16, "gemm_partials_ksplit must be in [0, 16]");
    lvllm::tuning().gemm_partials_ksplit = value;
  } else if (k == "gemm_balance") {
    lvllm::tuning().gemm_balance = value != 0;
  } else if (k == "varlen_dense") {
    lvllm::tuning().varlen_dense = value != 0;
  } else if (k == "varlen_dense_waves") {
    LV_CHECK(value == 0 || value == 4 || value == 8, "varlen_dense_waves must be 0 (by sequence length), 4 or 8");
    lvllm::tuning().varlen_dense_waves = value;
  } else if (k == "attn_waves") {
    LV_CHECK(value == 4 || value == 8, "attn_waves must be 4 or 8");
    lvllm::tuning().attn_waves = value;
  } else if (k == "attn_splits") {
    LV_CHECK(value >= -1, "attn_splits must be -1 (512-token partitions), 0 (automatic) or a share count");
    lvllm::tuning().attn_splits = value;
  } else if (k == "swap_kernel_min_runs") {
    LV_CHECK(value >= 0, "swap_kernel_min_runs must be non-negative");
    lvllm::tuning().swap_kernel_min_runs = value;
  } else if (k == "cache_tile_min_tokens") {
    LV_CHECK(value >= 1, "cache_tile_min_tokens must be positive");
    lvllm::tuning().cache_tile_min_tokens = value;
  } else if (k == "prefill_lds") {
    lvllm::tuning().prefill_lds = value != 0;
  } else if (k == "prefill_mfma32_min_query") {
    LV_CHECK(value >= 0, "prefill_mfma32_min_query must be non-negative (0: never)");
    lvllm::tuning().prefill_mfma32_min_query = value;
  } else if (k == "prefill_chunk_max_query") {
    LV_CHECK(value >= 0, "prefill_chunk_max_query must be non-negative (0: never)");
    lvllm::tuning().prefill_chunk_max_query = value;
  } else if (k == "prefill_chunk_max_avg_x8") {
    LV_CHECK(value >= 8, "prefill_chunk_max_avg_x8 must be at least 8 (one token per sequence)");
    lvllm::tuning().prefill_chunk_max_avg_x8 = value;
  } else {
    LV_CHECK(false, "unknown tuning key '" + k + "'");
  }
  return 0;
}

extern "C" int lvllm_get_tuning(const char* key, int* value) {
  const std::string k = key ? key : "";
  LV_CHECK(value != nullptr, "null result pointer");
  const lvllm::Tuning& t = lvllm::tuning();
  if (k == "gemm_workgroups") *value = t.gemm_workgroups;
  else if (k == "gemm_workgroups_wide") *value = t.gemm_workgroups_wide;
  else if (k == "gemm_wide_min_tiles") *value = t.gemm_wide_min_tiles;
  else if (k == "gemm_partials_ksplit") *value = t.gemm_partials_ksplit;
  else if (k == "gemm_balance") *value = t.gemm_balance;
  else if (k == "varlen_dense") *value = t.varlen_dense;
  else if (k == "varlen_dense_waves") *value = t.varlen_dense_waves;
  else if (k == "attn_waves") *value = t.attn_waves;
  else if (k == "attn_splits") *value = t.attn_splits;
  else if (k == "swap_kernel_min_runs") *value = t.swap_kernel_min_runs;
  else if (k == "cache_tile_min_tokens") *value = t.cache_tile_min_tokens;
  else if (k == "prefill_lds") *value = t.prefill_lds;
  else if (k == "prefill_mfma32_min_query") *value = t.prefill_mfma32_min_query;
  else if (k == "prefill_chunk_max_query") *value = t.prefill_chunk_max_query;
  else if (k == "prefill_chunk_max_avg_x8") *value = t.prefill_chunk_max_avg_x8;
  else LV_CHECK(false, "unknown tuning key '" + k + "'");
  return 0;
}

extern "C" const char* lvllm_version(void) { return "lvllm_hip gfx950 abi2"; }

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
