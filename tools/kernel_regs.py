#!/usr/bin/env python3
"""Register / spill / LDS figures of every kernel in a built library (from the code-object notes):
   python tools/kernel_regs.py [cf-nerf_amd/libcfnerf_hip.so]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "cf-nerf_amd", "libcfnerf_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={lib}"], capture_output=True)
    # the fat binary sits in .hip_fatbin: carve every embedded code object
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={d}/fat.bin", lib], check=True)
    blob = open(f"{d}/fat.bin", "rb").read()
    n = 0
    pos = 0
    while True:
        i = blob.find(b"\x7fELF", pos)
        if i < 0:
            break
        j = blob.find(b"\x7fELF", i + 4)
        open(f"{d}/co{n}.elf", "wb").write(blob[i:j if j > 0 else len(blob)])
        out = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f"{d}/co{n}.elf"], capture_output=True, text=True).stdout
        for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+)\s+\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", out, re.S):
            agpr, lds, name, priv, sgpr, sspill, vgpr, spill = m.groups()
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*\)$", "", dem).replace("void cfnerf::", "")
            print(f"{dem[:60]:60s} vgpr={vgpr:>3s} agpr={agpr:>3s} vgpr_spill={spill:>3s} sgpr={sgpr:>3s} sgpr_spill={sspill:>3s} scratch={priv:>5s} lds={lds:>6s}")
        n += 1
        pos = i + 4
