"""tools/asm_patch.py (a build step of __graft_entry__.build): the two-address LDS reads of the compiler's gfx950 assembly split into
single-address ones.  Pure text processing: checked here on hand-written lines, and on the real assembly when hipcc is present."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_patch  # noqa: E402


def test_split_forms():
    src = "\n".join([
        "kernel_a:",
        "\tds_read2_b64 v[10:13], v4 offset0:7 offset1:10",          # plain
        "\tds_read2_b64 v[2:5], v2 offset1:1",                      # the address register is in the FIRST half: that half last
        "\tds_read2_b64 v[6:9], v9 offset0:3",                      # ... in the second half: natural order
        "\tds_read2st64_b64 v[20:23], v1 offset0:1 offset1:2",      # offsets in units of 512 bytes
        "\tds_read2_b32 v[30:31], v1 offset1:1",                    # other widths are left alone
        "\tds_write2_b64 v1, v[2:3], v[4:5] offset1:1",
        "\ts_waitcnt lgkmcnt(1)",
    ])
    out, n, left = asm_patch.patch(src)
    lines = [line.strip() for line in out.split("\n")]
    assert n == 4 and left == 0
    assert lines[1:3] == ["ds_read_b64 v[10:11], v4 offset:56", "ds_read_b64 v[12:13], v4 offset:80"]
    assert lines[3:5] == ["ds_read_b64 v[4:5], v2 offset:8", "ds_read_b64 v[2:3], v2"]
    assert lines[5:7] == ["ds_read_b64 v[6:7], v9 offset:24", "ds_read_b64 v[8:9], v9"]
    assert lines[7:9] == ["ds_read_b64 v[20:21], v1 offset:512", "ds_read_b64 v[22:23], v1 offset:1024"]
    assert lines[9] == "ds_read2_b32 v[30:31], v1 offset1:1" and lines[10].startswith("ds_write2_b64") and lines[11] == "s_waitcnt lgkmcnt(1)"


def test_only_named_functions():
    src = "kernel_a: ; @kernel_a\n\tds_read2_b64 v[0:3], v8 offset1:1\nkernel_b: ; @kernel_b\n\tds_read2_b64 v[0:3], v8 offset1:1\n"
    out, n, _ = asm_patch.patch(src, ("kernel_b",))
    assert n == 1 and out.count("ds_read2_b64") == 1 and out.index("ds_read2_b64") < out.index("kernel_b")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_on_the_compiler_output_of_a_small_kernel(tmp_path):
    """every split line assembles, and no two-address 8-byte read is left"""
    hip = tmp_path / "k.hip"
    hip.write_text("#include <hip/hip_runtime.h>\n__global__ void k(double* o) { __shared__ double b[1024]; const int t = threadIdx.x;\n"
                   "  b[t] = t; b[t + 256] = 2 * t; __syncthreads(); const double* p = b + 3 * (t & 63) + 1;\n"
                   "  o[t] = p[0] * p[1] + p[2] * p[5] + p[7]; }\n")
    s = tmp_path / "k.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "--cuda-device-only", "-S", "-o", str(s), str(hip)],
                          stderr=subprocess.DEVNULL)
    out, n, left = asm_patch.patch(s.read_text())
    assert left == 0 and "ds_read2_b64" not in out and "ds_read2st64_b64" not in out
    p = tmp_path / "k_split.s"
    p.write_text(out)
    clang = "/opt/rocm/lib/llvm/bin/clang"
    subprocess.check_call([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(p), "-o", str(tmp_path / "k.o")])
