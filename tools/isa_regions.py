"""Static instruction mix of one kernel in a `hipcc -S` listing, split at s_barrier: tools/isa_regions.py file.s <mangled-name-substring>
Prints per barrier-delimited region: VALU (non-MFMA), MFMA, LDS, VMEM, SALU counts and the backward-branch targets (loops) inside."""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
body = lines[start:end + 1]
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store"): return "smem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
regions, cur, labels = [], {}, {}
pos = 0
seq = []
for l in body:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", s): labels[s[:-1]] = len(seq)
        continue
    if re.match(r"^\.?[A-Za-z_0-9$]+:", s):
        labels[s.split(":")[0]] = len(seq); continue
    op = s.split()[0]
    seq.append((op, s))
# loops: backward branches
loops = []
for i, (op, s) in enumerate(seq):
    if op.startswith("s_cbranch") or op == "s_branch":
        tgt = s.split()[-1]
        if tgt in labels and labels[tgt] <= i: loops.append((labels[tgt], i))
bar = [i for i, (op, _) in enumerate(seq) if op == "s_barrier"]
edges = [0] + bar + [len(seq)]
tot = {}
for r in range(len(edges) - 1):
    a, b = edges[r], edges[r + 1]
    c = {}
    for op, _ in seq[a:b]:
        k = cls(op); c[k] = c.get(k, 0) + 1; tot[k] = tot.get(k, 0) + 1
    lp = [(x, y) for x, y in loops if a <= x and y < b]
    lpd = []
    for x, y in lp:
        cc = {}
        for op, _ in seq[x:y + 1]:
            k = cls(op); cc[k] = cc.get(k, 0) + 1
        lpd.append("loop[%d..%d] valu=%d mfma=%d lds=%d vmem=%d" % (x, y, cc.get("valu", 0), cc.get("mfma", 0), cc.get("lds", 0), cc.get("vmem", 0)))
    print("region %2d [%6d..%6d) valu=%5d mfma=%4d lds=%4d vmem=%4d salu=%5d smem=%3d wait=%3d  %s" % (
        r, a, b, c.get("valu", 0), c.get("mfma", 0), c.get("lds", 0), c.get("vmem", 0), c.get("salu", 0), c.get("smem", 0), c.get("wait", 0), "; ".join(lpd)))
print("total", tot)
# top VALU opcodes
from collections import Counter
cn = Counter(op for op, _ in seq if cls(op) == "valu")
print(cn.most_common(25))
