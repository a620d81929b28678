// Instantiates the MFMA paged-attention ladder for bf16 (see attention_mfma.h).
#include "attention_mfma.h"

namespace lvllm {
template int launch_mfma_hs<BF16>(const AttnParams&, int, int, int, int, int, hipStream_t);
}  // namespace lvllm

LVLLM_TRACE_READER(lvllm_trace_read_attn)
