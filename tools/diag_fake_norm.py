"""Diagnosis (WRONG results): bench.py with the add + norm launch behind the output projection skipped, for pricing a
[add + norm -> projection] fusion.  With the plain library the run measures what removing the launch is worth at most;
with a -DLVLLM_GEMM_FAKE_NORM build of skinny_gemm.hip (tools/build_variant.sh fakenorm skinny_gemm.hip
-DLVLLM_GEMM_FAKE_NORM) the SwiGLU projection also pays for normalising its own activations in its prologue.
    python tools/diag_fake_norm.py [--skip post|none] [bench.py flags ...]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
skip = "post"
args = sys.argv[1:]
if args[:1] == ["--skip"]:
    skip, args = args[1], args[2:]
import light_vllm_amd  # noqa: E402,F401
from light_vllm_amd.engine import model as M  # noqa: E402

if skip == "post":
    real = M.DecoderModel._add_norm

    def patched(self, x, residual, weight):
        post = getattr(self, "_post_norm_ptrs", None)
        if post is None:
            post = self._post_norm_ptrs = {lw.post_norm.data_ptr() for lw in self.layers}
        if weight.data_ptr() in post and not isinstance(x, tuple) and x.dim() == 2:
            return x  # no add, no norm: the projection behind it reads the output projection's result as it is
        return real(self, x, residual, weight)
    M.DecoderModel._add_norm = patched
sys.argv = [os.path.join(ROOT, "bench.py")] + args
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
