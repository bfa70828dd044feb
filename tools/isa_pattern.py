"""print the instruction-class sequence of the hottest loop (most MFMAs) of one kernel in a hipcc -save-temps .s file
usage: python tools/isa_pattern.py file.s mangled_kernel_substring"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ": " in l + " ")
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
# basic blocks
blocks, cur, name = [], [], "entry"
for l in body:
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1)
    else:
        t = l.strip()
        if t and not t.startswith((";", ".")):
            cur.append(t)
blocks.append((name, cur))
def cls(i):
    op = i.split()[0]
    if op.startswith("v_mfma"): return "M"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "r"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "w"
    if op.startswith(("global_load", "buffer_load")): return "G"
    if op.startswith(("global_store", "buffer_store")): return "S"
    if op.startswith("s_waitcnt"): return "|"
    if op.startswith("s_barrier"): return "B"
    if op.startswith("v_"): return "v"
    if op.startswith("s_"): return "s"
    return "?"
best = max(blocks, key=lambda b: sum(1 for i in b[1] if i.startswith("v_mfma")))
seq = "".join(cls(i) for i in best[1])
print(best[0], "instructions", len(best[1]), {c: seq.count(c) for c in "MrwGvs|B"})
for i in range(0, len(seq), 120):
    print(seq[i:i + 120])
waits = [i for i in best[1] if i.startswith("s_waitcnt")]
print(waits)
