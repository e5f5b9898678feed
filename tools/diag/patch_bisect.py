#!/usr/bin/env python3
"""Diagnostic (never part of the product): builds libhipnlp variants whose ds_read2_b64 split (tools/asm_patch.py) is applied to a chosen
subset of the sites of hipnlp_pose.hip only — to find a site whose split changes behaviour.
usage: patch_bisect.py OUT.so KERNEL_SUBSTRING FIRST LAST [overlap|nonoverlap]
    sites FIRST <= i < LAST (in order of appearance inside the functions whose label contains KERNEL_SUBSTRING) are split; everything else
    in hipnlp_pose.hip keeps the compiler's ds_read2_b64; hipnlp.hip is compiled by plain hipcc."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_patch

HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-I", os.path.join(ROOT, "include")]


def main():
    out, key, first, last = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    kind = sys.argv[5] if len(sys.argv) > 5 else ""
    llvm = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(HIPCC))), "lib", "llvm", "bin")
    clang, lld, bundler = (os.path.join(llvm, t) for t in ("clang", "lld", "clang-offload-bundler"))
    tmp = tempfile.mkdtemp(prefix="hipnlp_bisect_")
    try:
        src = os.path.join(ROOT, "hippopt_amd", "csrc", "hipnlp_pose.hip")
        base = os.path.join(tmp, "pose")
        subprocess.check_call([HIPCC] + FLAGS + ["--cuda-device-only", "-S", "-o", base + ".s", src])
        lines, active, idx, n = open(base + ".s").read().split("\n"), False, 0, 0
        res = []
        for line in lines:
            if line[:1] not in (" ", "\t", ".", ";", "") and ":" in line.split(";")[0]:
                active = key in line
            m = asm_patch.PAT.match(line) if active else None
            if m:
                lo, addr = int(m.group(3)), int(m.group(5))
                ov = lo <= addr <= lo + 3
                take = first <= idx < last and (kind == "" or (kind == "overlap") == ov)
                idx += 1
                if take:
                    text, k, _ = asm_patch.patch(line)
                    res.append(text)
                    n += k
                    continue
            res.append(line)
        print("patch_bisect: %d sites in '%s', %d split" % (idx, key, n))
        open(base + "_split.s", "w").write("\n".join(res))
        subprocess.check_call([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", base + "_split.s", "-o", base + "_dev.o"])
        subprocess.check_call([lld, "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", base + ".hsaco", base + "_dev.o"])
        subprocess.check_call([bundler, "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
                               "-input=/dev/null", "-input=" + base + ".hsaco", "-output=" + base + ".hipfb"])
        subprocess.run([HIPCC] + FLAGS + ["-fPIC", "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", base + ".hipfb", "-c", src, "-o", base + "_host.o"],
                       check=True, stderr=subprocess.DEVNULL)
        plain = os.path.join(ROOT, "tools", "diag", "_build", "hipnlp_plain.o")
        if not os.path.exists(plain) or os.path.getmtime(plain) < os.path.getmtime(os.path.join(ROOT, "hippopt_amd", "csrc", "hipnlp.hip")):
            subprocess.run([HIPCC] + FLAGS + ["-fPIC", "-c", os.path.join(ROOT, "hippopt_amd", "csrc", "hipnlp.hip"), "-o", plain], check=True, stderr=subprocess.DEVNULL)
        ip = os.path.join(tmp, "ipopt.o")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "hippopt_amd", "csrc", "hipnlp_ipopt.cpp"), "-o", ip])
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, base + "_host.o", plain, ip])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
