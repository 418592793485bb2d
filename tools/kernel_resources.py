"""List VGPR / SGPR / LDS / scratch use of every gfx950 kernel in libkzg_mi355x.so (reads the code-object notes)."""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "kzg_snark_amd", "lib", "libkzg_mi355x.so")
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
d = open(fat, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
offs = [m.start() for m in re.finditer(magic, d)] + [len(d)]
for k in range(len(offs) - 1):
    b = os.path.join(tmp, f"b{k}.bin"); o = os.path.join(tmp, f"co{k}.o")
    open(b, "wb").write(d[offs[k]:offs[k + 1]])
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={b}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={o}"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", o], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count")[1:]:
        g = lambda key: (re.search(r"\." + key + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        if g("vgpr_count") in ("0", "?"): continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        m = re.search(r"(\w+_kernel)<", dem)
        short = m.group(1) if m else dem[:110]
        if "rocprim" in dem:
            m2 = re.search(r"wrapped_(\w+?)_config", dem); short = "rocprim:" + (m2.group(1) if m2 else dem[:60])
            m3 = re.search(r"(onesweep\w*|histogram\w*|block_sort\w*|lookback\w*|scan\w*)", dem)
            if m3: short += ":" + m3.group(1)
        print("%-60s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %5s wg %4s" % (
            short[:60], g("vgpr_count"), blk.split()[0].strip(":") if False else re.search(r"^:\s+(\d+)", blk).group(1) if re.search(r"^:\s+(\d+)", blk) else "?",
            g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("max_flat_workgroup_size")))
