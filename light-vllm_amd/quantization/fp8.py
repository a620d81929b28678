"""apply_fp8_linear: light_vllm/backends/quantization/utils/w8a8_utils.py:103-189, the
`cutlass_fp8_supported=False` branch (torch._scaled_mm), which is the one a ROCm build takes."""
from typing import Optional, Tuple

import torch

from .. import _custom_ops as ops

FP8_MAX = 448.0  # torch.finfo(torch.float8_e4m3fn).max


def per_tensor_quantize_weight(weight: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[N, K] weight -> (fp8 weight transposed to [K, N] as _scaled_mm wants its second operand,
    scale [1]); the dynamic per-tensor kernel does the work (fp8.py:196-213 of the reference
    quantises checkpoint weights the same way at load time)."""
    q, scale = ops.scaled_fp8_quant(weight.contiguous())
    return q.t(), scale


def apply_fp8_linear(input: torch.Tensor, weight: torch.Tensor, weight_scale: torch.Tensor,
                     input_scale: Optional[torch.Tensor] = None, input_scale_ub: Optional[torch.Tensor] = None,
                     bias: Optional[torch.Tensor] = None, cutlass_fp8_supported: bool = False,
                     use_per_token_if_dynamic: bool = False) -> torch.Tensor:
    if cutlass_fp8_supported:
        raise NotImplementedError("cutlass_scaled_mm is an NVIDIA kernel; use the torch._scaled_mm path")
    # the input is padded to 17 rows as in the reference (w8a8_utils.py:137-144)
    qinput, x_scale = ops.scaled_fp8_quant(input, input_scale, num_token_padding=17,
                                           scale_ub=input_scale_ub if use_per_token_if_dynamic else None,
                                           use_per_token_if_dynamic=use_per_token_if_dynamic)
    per_tensor_weights = weight_scale.numel() == 1
    per_tensor_activations = x_scale.numel() == 1
    if per_tensor_weights and per_tensor_activations:
        output = torch._scaled_mm(qinput, weight, out_dtype=input.dtype, scale_a=x_scale, scale_b=weight_scale,
                                  bias=bias)
        if isinstance(output, tuple):
            output = output[0]
        return torch.narrow(output, 0, 0, input.shape[0])
    # channelwise weights or per-token activations: C = s_w * s_x * (X W) + bias, unfused
    one = torch.ones(1, dtype=torch.float32, device=input.device)
    output = torch._scaled_mm(qinput, weight, out_dtype=torch.float32, scale_a=one, scale_b=one)
    if isinstance(output, tuple):
        output = output[0]
    output = torch.narrow(output, 0, 0, input.shape[0])
    x_scale = torch.narrow(x_scale, 0, 0, input.shape[0])
    output = output * x_scale * weight_scale.t()
    if bias is not None:
        output = output + bias
    return output.to(dtype=input.dtype)


def pack_fp8_weight(weight_fp8: torch.Tensor) -> torch.Tensor:
    """[N, K] fp8 weight (row-major, K % 64 == 0, N % 16 == 0) -> the order the weight-streaming
    kernel reads: lvllm_pack_weight applied to the [N, K/2] 16-bit view of the same bytes."""
    N, K = weight_fp8.shape
    assert weight_fp8.element_size() == 1 and weight_fp8.is_contiguous() and K % 64 == 0 and N % 16 == 0
    return torch.ops._C_amd.pack_weight(weight_fp8.view(torch.float16)).view(torch.uint8)


def skinny_fp8_linear(x: torch.Tensor, w_packed: torch.Tensor, w_scale: torch.Tensor, x_scale: torch.Tensor,
                      N: int, K: int, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """W8A8 projection of a decode batch (M <= 64) with a static per-tensor activation scale:
    one launch, activation quantised inside the kernel, fp8 x fp8 MFMA, scales and bias in the
    epilogue -- the fused equivalent of apply_fp8_linear's per-tensor branch."""
    return torch.ops._C_amd.skinny_linear_w8a8(x, w_packed, w_scale, x_scale, N, K, bias)
