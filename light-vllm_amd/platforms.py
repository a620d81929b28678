"""ROCm platform for the reference's platform probe.

`light_vllm/platforms/__init__.py:9-13` knows two cases: `torch.version.cuda is not None` -> CudaPlatform
(pynvml), anything else -> UnspecifiedPlatform, whose `get_device_capability()` returns None.  On
PyTorch-ROCm that None is indexed by `_check_if_gpu_supports_dtype(bf16)`
(decoding/worker/gpu_worker.py:227-238, prefill_only/worker/gpu_worker.py:83) and by the prefill-only backend
selector (prefill_only/backends/attention/selector.py:108-109): bf16 cannot start.  `RocmPlatform` has the method
set of `light_vllm/platforms/interface.py:31-106` (same names, arguments, return types) answered from the HIP
runtime through torch; `install()` puts it where the reference looks.
"""
import enum
from typing import NamedTuple, Optional, Tuple, Union

import torch


class PlatformEnum(enum.Enum):  # interface.py:7-13
    CUDA = enum.auto()
    ROCM = enum.auto()
    TPU = enum.auto()
    XPU = enum.auto()
    CPU = enum.auto()
    UNSPECIFIED = enum.auto()


class DeviceCapability(NamedTuple):  # interface.py:16-28
    major: int
    minor: int

    def as_version_str(self) -> str:
        return f"{self.major}.{self.minor}"

    def to_int(self) -> int:
        assert 0 <= self.minor < 10
        return self.major * 10 + self.minor


class RocmPlatform:
    _enum = PlatformEnum.ROCM

    def is_cuda(self) -> bool:
        return False

    def is_rocm(self) -> bool:
        return True

    def is_tpu(self) -> bool:
        return False

    def is_xpu(self) -> bool:
        return False

    def is_cpu(self) -> bool:
        return False

    def is_cuda_alike(self) -> bool:
        """CUDA or ROCm: torch.cuda is the device namespace (interface.py:49-51)."""
        return True

    @classmethod
    def get_device_capability(cls, device_id: int = 0) -> Optional[DeviceCapability]:
        """(major, minor) of the gfx target: gfx950 -> (9, 5).  Every check the reference makes is
        `capability[0] >= 8` or `to_int() >= 80/89` (bf16, fp8, flash-attn): CDNA3/4 pass them, as they should
        (bf16 and OCP fp8 MFMA are native on gfx950)."""
        major, minor = torch.cuda.get_device_capability(device_id)
        return DeviceCapability(major=major, minor=minor)

    @classmethod
    def has_device_capability(cls, capability: Union[Tuple[int, int], int], device_id: int = 0) -> bool:
        current = cls.get_device_capability(device_id=device_id)
        if current is None:
            return False
        if isinstance(capability, tuple):
            return current >= capability
        return current.to_int() >= capability

    @classmethod
    def get_device_name(cls, device_id: int = 0) -> str:
        return torch.cuda.get_device_name(device_id)

    @classmethod
    def get_device_total_memory(cls, device_id: int = 0) -> int:
        return int(torch.cuda.get_device_properties(device_id).total_memory)

    @classmethod
    def inference_mode(cls):
        return torch.inference_mode(mode=True)


def is_rocm_torch() -> bool:
    return torch.version.hip is not None


def install(platforms_module=None):
    """Makes the reference see a ROCm platform: `light_vllm.platforms.current_platform = RocmPlatform()`, and the
    same for modules that bound the name at import time (`from light_vllm.platforms import current_platform`).
    Returns the platform object.  No-op module-wise when light_vllm is not importable."""
    import sys
    plat = RocmPlatform()
    if platforms_module is None:
        platforms_module = sys.modules.get("light_vllm.platforms")
        if platforms_module is None:
            try:
                import importlib
                platforms_module = importlib.import_module("light_vllm.platforms")
            except Exception:
                return plat
    old = getattr(platforms_module, "current_platform", None)
    platforms_module.current_platform = plat
    for name, mod in list(sys.modules.items()):
        if name.startswith("light_vllm.") and getattr(mod, "current_platform", None) is old and old is not None:
            mod.current_platform = plat
    return plat
