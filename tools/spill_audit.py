"""Developer tool: global loads of a kernel that sit behind a spill reload (scratch_load + s_waitcnt vmcnt(0)
serialises every load in flight).  usage: python tools/spill_audit.py /tmp/tls.s <kernel-name-substring>"""
import sys
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().endswith(":") is False and ":" in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = [l for l in lines[start:end] if l.startswith("\t") and not l.startswith("\t;") and not l.startswith("\t.")]
marks = 0
out = []
for i, l in enumerate(body):
    if "s_memtime" in l:
        marks += 1
    if "global_load" in l or "global_store" in l:
        back = body[max(0, i - 10):i]
        flag = any("scratch_load" in b for b in back)
        wait = any("vmcnt(0)" in b for b in back)
        out.append((i, marks, l.split()[0], "SPILL" if flag else "", "wait0" if wait else ""))
print("instructions", len(body), "memtime marks", marks)
last = None
for i, m, op, f, w in out:
    print(i, "after_mark", m, op, f, w)
