#!/usr/bin/env python3
"""Static instruction histogram of one gfx950 kernel of an object file of the build:

    python tools/isa_histogram.py msm.o msm_accumulate_kernel Bls12_381 Li20   [--dump out.s]

Every simple VALU instruction issues at the same rate on gfx950 (tools/microbench/int_rates.hip),
so instruction counts of the hot loops are what the kernels are tuned against."""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True,
                   stderr=subprocess.DEVNULL)
    d = open(fat, "rb").read()
    offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", d)] + [len(d)]
    text = ""
    for k in range(len(offs) - 1):
        b, o = os.path.join(tmp, f"b{k}.bin"), os.path.join(tmp, f"co{k}.o")
        open(b, "wb").write(d[offs[k]:offs[k + 1]])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={b}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={o}"], check=True)
        text += subprocess.run([f"{LLVM}/llvm-objdump", "-d", o], capture_output=True, text=True).stdout
    return text


def main():
    argv = sys.argv[1:]
    dump = None
    if "--dump" in argv:
        i = argv.index("--dump")
        dump = argv[i + 1]
        del argv[i:i + 2]
    args = argv
    obj = args[0] if os.path.exists(args[0]) else os.path.join(ROOT, "kzg_snark_amd", "lib", args[0])
    keys = args[1:]
    text = disassemble(obj)
    blocks = re.split(r"\n(?=[0-9a-f]{16} <)", text)
    for blk in blocks:
        head = blk.split("\n", 1)[0]
        if not all(k in head for k in keys):
            continue
        cnt = collections.Counter()
        for line in blk.splitlines()[1:]:
            m = re.match(r"\s+(\w+)\s", line)
            if m:
                cnt[m.group(1)] += 1
        tot = sum(cnt.values())
        print(head.strip())
        print("total", tot, " mads", cnt["v_mad_u64_u32"], " other", tot - cnt["v_mad_u64_u32"])
        for k, v in cnt.most_common(14):
            print(f"  {k:26s}{v:6d} {100 * v / tot:5.1f}%")
        if dump:
            open(dump, "w").write(blk)


if __name__ == "__main__":
    main()
