#!/usr/bin/env python3
"""Every buffer store wider than 64 bits in a built library must carry its offset in the VECTOR offset with the literal 0 as soffset:

    python tools/check_wide_stores.py [cf-nerf_amd/libcfnerf_hip.so]      (exit code 1 + the offending lines otherwise)

Why (round 5): on gfx950 a buffer_store_dwordx4 whose offset sits in an SGPR reads its four data VGPRs LATE; a VALU write
into those registers a few instructions on reached memory instead (sparse garbage in g_h, different from run to run).  LLVM's hazard
recognizer inserts the wait state of the "store wider than 64 bits, then VALU write of its data" hazard only for stores WITHOUT an SGPR
offset (GCNHazardRecognizer::createsVALUHazard), so the protection of q4_store / stash_rows (csrc/cfnerf_device.h) rests on the form of
the instruction the compiler emits - which a compiler upgrade or a different addressing pattern could change silently.  This check reads the
form back from the code objects (tests/test_abi_cpu.py runs it on every build): any dwordx3 / dwordx4 buffer store whose soffset is not
the literal 0 fails it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib, d):
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={d}/fat.bin", lib], check=True)
    blob = open(f"{d}/fat.bin", "rb").read()
    out, pos, n = [], 0, 0
    while True:
        i = blob.find(b"\x7fELF", pos)
        if i < 0:
            return out
        j = blob.find(b"\x7fELF", i + 4)
        path = f"{d}/co{n}.elf"
        open(path, "wb").write(blob[i:j if j > 0 else len(blob)])
        out.append(path)
        n += 1
        pos = i + 4


def check(lib):
    bad, n_wide = [], 0
    pat = re.compile(r"\b(buffer_store_dwordx[34])\s+v\[\d+:\d+\],\s*(?:v\d+|off),\s*s\[\d+:\d+\],\s*(\S+)")
    with tempfile.TemporaryDirectory() as d:
        for co in code_objects(lib, d):
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            fn = "?"
            for line in dis.splitlines():
                if line.endswith(">:"):
                    fn = line.split("<")[-1][:-2]
                    continue
                m = pat.search(line)
                if m:
                    n_wide += 1
                    if m.group(2) != "0":
                        bad.append((fn, line.strip()))
    return n_wide, bad


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "cf-nerf_amd", "libcfnerf_hip.so")
    n, bad = check(lib)
    print(f"{n} buffer stores wider than 64 bits, {len(bad)} with an SGPR / non-zero soffset")
    for fn, line in bad[:40]:
        print(f"  {fn[:80]}: {line}")
    sys.exit(1 if bad else 0)
