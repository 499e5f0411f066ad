#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of the built library (from the code-object metadata notes): a quick way to see
that a change to a hot kernel did not spill (`.private_segment_fixed_size` must stay 0 for the GEMM kernels).

    python tools/kernel_resources.py [path/to/libtutel_amd.so] [substring filter]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(lib):
    tmp = "/tmp/tutel_amd_co"
    os.makedirs(tmp, exist_ok=True)
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--list", f"--input={lib}"], capture_output=True)
    # the fat binary section holds the gfx950 code object; roc-obj-ls / extraction via llvm-objcopy of .hip_fatbin
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    data = open(fat, "rb").read()
    out = []
    # bundles: each code object is an ELF; find ELF magics and dump notes of each
    pos = [m.start() for m in re.finditer(b"\x7fELF", data)]
    for i, a in enumerate(pos):
        b = pos[i + 1] if i + 1 < len(pos) else len(data)
        co = os.path.join(tmp, f"co{i}.elf")
        open(co, "wb").write(data[a:b])
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True)
        txt = r.stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            def g(key):
                m = re.search(r"\." + key + r":\s+(\S+)", blk)
                return m.group(1) if m else "?"
            out.append((g("name"), g("vgpr_count"), "a" + blk.split()[0], g("sgpr_count"), g("private_segment_fixed_size"),
                        g("group_segment_fixed_size")))
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(ROOT, "tutel_amd", "lib", "libtutel_amd.so")
    filt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    for name, v, a, s_, scratch, lds in sorted(resources(lib)):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if filt in dem:
            print(f"vgpr {v:>4} {a:>5} sgpr {s_:>4} scratch {scratch:>5} lds {lds:>6}  {dem[:150]}")
