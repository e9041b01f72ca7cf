"""sdfk_sqrt (the JIT prelude's MathF.Sqrt, sample_codegen.h) must be the correctly rounded root for EVERY float.
Its short path -- v_rsq_f32 and one FMA correction step -- is exact on gfx950 as a matter of fact, not of theorem, so
the fact is checked here exhaustively: the function's text is taken from a generated program (what hiprtc compiles),
wrapped in a kernel that walks all 2^32 bit patterns, built with hipcc with the JIT's options and compared with the
fp64 root rounded once (53 >= 2*24 + 2 bits: innocuous double rounding)."""
import ctypes as C
import os
import re
import shutil
import subprocess

import pytest

from sdfkit_amd import MathF, Sdf, Vec4
from sdfkit_amd import _native as N

pytestmark = pytest.mark.gpu

HARNESS = r"""
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
@PRELUDE@
__global__ void walk(unsigned long long* bad, uint32_t* first)
{
    unsigned long long nb = 0;
    for (uint64_t b = blockIdx.x * 256ull + threadIdx.x; b < (1ull << 32); b += gridDim.x * 256ull) {
        const float x = __builtin_bit_cast(float, (uint32_t)b);
        const float got = sdfk_sqrt(x);
        const float want = (float)__builtin_sqrt((double)x);
        const uint32_t g = __builtin_bit_cast(uint32_t, got), w = __builtin_bit_cast(uint32_t, want);
        const bool same = g == w || (got != got && want != want);   // any NaN for a NaN
        if (!same) { nb++; atomicMin(first, (uint32_t)b); }
    }
    if (nb) atomicAdd(bad, nb);
}
int main()
{
    unsigned long long* bad; uint32_t* first;
    if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&first, 4) != hipSuccess) return 2;
    hipMemset(bad, 0, 8); hipMemset(first, 0xff, 4);
    hipLaunchKernelGGL(walk, dim3(256 * 64), dim3(256), 0, 0, bad, first);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    unsigned long long hb; uint32_t hf;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("mismatches %llu first 0x%08x\n", hb, hf);
    return hb ? 1 : 0;
}
"""


def test_sdfk_sqrt_every_float(gpu, tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    sdf = Sdf(lambda p: Vec4(0.0, 0.0, 0.0, MathF.Sqrt(p.x)), False)
    L = N.lib()
    L.sdfk_program_source.restype = C.c_char_p
    src = L.sdfk_program_source(sdf.program()).decode()
    m = re.search(r"__device__ __forceinline__ float sdfk_sqrt\(float x\)\n\{.*?\n\}\n", src, re.S)
    assert m, "sdfk_sqrt not found in the generated source"
    (tmp_path / "walk.hip").write_text(HARNESS.replace("@PRELUDE@", m.group(0)))
    exe = tmp_path / "walk"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", str(tmp_path / "walk.hip"), "-o", str(exe)],
                          stderr=subprocess.DEVNULL)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "mismatches 0 " in r.stdout, r.stdout + r.stderr
    shutil.rmtree(tmp_path, ignore_errors=True)
