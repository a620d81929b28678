"""fp8 W8A8 linear of the reference (backends/quantization/utils/w8a8_utils.py): activation
quantisation by this package's HIP kernels, GEMM + dequantisation through `torch._scaled_mm`
(hipBLASLt fp8 MFMA on gfx950, OCP e4m3fn) -- the reference's own route on ROCm, where its CUTLASS
kernels do not exist."""
from .fp8 import (apply_fp8_linear, pack_fp8_weight, per_tensor_quantize_weight,  # noqa: F401
                  skinny_fp8_linear)
