#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel in a built libhipnlp.so, read from the code-object metadata of the library
ITSELF (what the GPU box will load), not from a fresh compile: the .hip_fatbin section holds one clang offload bundle per
translation unit; its gfx950 entries are ELF code objects whose NT_AMDGPU_METADATA note lists every kernel.
    python tools/kernel_resources.py [path/to/libhipnlp.so]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("HIPNLP_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib_path, os.path.join(tmp, "copy.so")])
    data = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and sz:
                p = os.path.join(tmp, "dev%d.co" % len(out))
                open(p, "wb").write(data[i + o:i + o + sz])
                out.append(p)
        pos = i + 1


def kernel_resources(lib_path=None):
    """{demangled-ish kernel name: {"vgpr", "sgpr", "scratch", "lds", "wg"}}"""
    lib_path = lib_path or os.path.join(ROOT, "hippopt_amd", "lib", "libhipnlp.so")
    res = {}
    with tempfile.TemporaryDirectory(prefix="hipnlp_res_") as tmp:
        for co in code_objects(lib_path, tmp):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:   # one block per kernel (.agpr_count is its first key, alphabetical)
                def num(key):
                    m = re.search(r"\." + key + r":\s+(\d+)", blk)
                    return int(m.group(1)) if m else -1
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
                short = name
                if m:
                    n0 = m.end()
                    short = name[n0:n0 + int(m.group(1))]
                    t = re.match(r"I((?:L[ib]\d+E)+)E", name[n0 + int(m.group(1)):])
                    if t:
                        args = re.findall(r"L([ib])(\d+)E", t.group(1))
                        while args and args[-1] == ("b", "0"):   # (trailing `false` defaults are left out: <0,4> is <0,4,false>)
                            args.pop()
                        short += "<" + ",".join(("true" if v == "1" else "false") if k == "b" else v for k, v in args) + ">"
                res[short] = {"vgpr": num("vgpr_count"), "agpr": int(blk.split()[0]), "sgpr": num("sgpr_count"), "scratch": num("private_segment_fixed_size"),
                              "lds": num("group_segment_fixed_size"), "wg": num("max_flat_workgroup_size")}
    return res


if __name__ == "__main__":
    r = kernel_resources(sys.argv[1] if len(sys.argv) > 1 else None)
    for k in sorted(r):
        v = r[k]
        print("%-34s VGPRs %3d (+%d AGPRs)  SGPRs %3d  scratch %4d B/lane  LDS %6d B  workgroup %4d" % (k, v["vgpr"], v["agpr"], v["sgpr"], v["scratch"], v["lds"], v["wg"]))
