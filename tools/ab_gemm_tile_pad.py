"""Diagnosis: do the projections of a decode step care where the 16-row tiles of a packed weight start?  Builds of
liblvllm_hip.so with -DLVLLM_GEMM_TILE_PAD=<bytes> (variants/tilepad<bytes>/) against the shipped one, through the C-ABI
(ctypes: lvllm_pack_weight, lvllm_skinny_gemm), alternating in ONE process; M = 32 rows of bf16, the four shapes of an
8B model; HIP events around trains of 32 launches over 8 rotating weight copies (>> the Infinity Cache).
usage: python tools/ab_gemm_tile_pad.py <pad bytes> [<pad bytes> ...]    (0 = light-vllm_amd/lib)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (28672, 4096), "down": (4096, 14336)}
M, COPIES, TRAIN, TRAINS = 32, 8, 32, 8


def load(pad):
    """pad: bytes (a -DLVLLM_GEMM_TILE_PAD build under variants/tilepad<bytes>/), 0 = the shipped library, or the name
    of any other diagnosis build under variants/tilepad_<name>/ (same weight layout as shipped; results may be wrong)"""
    if pad == 0:
        path = os.path.join(ROOT, "light-vllm_amd", "lib", "liblvllm_hip.so")
    elif isinstance(pad, str):
        path = os.path.join(ROOT, "variants", f"tilepad_{pad}", "liblvllm_hip.so")
    else:
        path = os.path.join(ROOT, "variants", f"tilepad{pad}", "liblvllm_hip.so")
    return ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)


def main():
    pads = [int(p) if p.lstrip('-').isdigit() else p for p in sys.argv[1:]] or [0]
    dev = "cuda:0"
    torch.manual_seed(0)
    libs = {p: load(p) for p in pads}
    vp = ctypes.c_void_p
    res = {p: {} for p in pads}
    for name, (N, K) in SHAPES.items():
        x = (torch.randn(M, K, device=dev) * 0.1).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        want = (x.float() @ w.float().T)
        for rnd in range(2):
            for p in pads:
                lib = libs[p]
                nbytes = N * K * 2 + (N // 16) * (p if isinstance(p, int) else 0)
                packed = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(COPIES)]
                stream = torch.cuda.current_stream().cuda_stream
                for b in packed:
                    assert lib.lvllm_pack_weight(vp(b.data_ptr()), vp(w.data_ptr()), N, K, 2, vp(stream)) == 0
                ws_bytes = lib.lvllm_skinny_gemm_workspace_bytes
                ws_bytes.restype = ctypes.c_int64
                nws = ws_bytes(M, N, K)
                ws = torch.empty(max(nws, 4), dtype=torch.uint8, device=dev)
                y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

                def launch(i):
                    rc = lib.lvllm_skinny_gemm(vp(y.data_ptr()), vp(x.data_ptr()), vp(packed[i % COPIES].data_ptr()), vp(0),
                                               M, N, K, ctypes.c_int64(K), 2, 1, vp(ws.data_ptr() if nws else 0),
                                               ctypes.c_int64(nws), vp(stream))
                    assert rc == 0, rc
                for i in range(TRAIN):
                    launch(i)
                torch.cuda.synchronize()
                err = float((y.float() - want).abs().max() / want.abs().max())
                assert isinstance(p, str) or err < 2e-2, (name, p, err)
                ts = []
                for _ in range(TRAINS):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for i in range(TRAIN):
                        launch(i)
                    b.record()
                    torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b) / TRAIN * 1e3)
                res[p].setdefault(name, []).append(min(ts))
                del packed
    for p in pads:
        line = "  ".join(f"{n} {'/'.join(f'{t:.2f}' for t in res[p][n])} us ({SHAPES[n][0] * SHAPES[n][1] * 2 / min(res[p][n]) / 1e6:.2f} TB/s)"
                         for n in SHAPES)
        print(f"{p!s:>10}: {line}")


if __name__ == "__main__":
    main()
