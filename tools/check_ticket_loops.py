"""Developer check on the device assembly (tools/count_spills.sh leaves it in /tmp/tls.s): every LDS ticket fetch
`if (lane == 0) t = atomicAdd(counter, 1); t = readfirstlane(t)` at the head of a wave-level work loop must stay ONE
scalar loop.  hipcc once threaded it together with a `lane == 0` store at the loop's end (search-role instantiation of
the strided-row predicate): lane 0 went on to the next ticket while lanes 1..63 repeated the row for ever.  The symptom
in the assembly is a second loop header between the atomic and the use of its result."""
import re
import sys

lines = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/tls.s").read().split("\n")
name, bad, seen = None, 0, 0
for i, l in enumerate(lines):
    m = re.match(r"^(_ZN6tlsdev\S+):", l)
    if m:
        name = m.group(1)
    if name and "tls_search_kernel" in name and "ds_add_rtn_u32" in l:
        seen += 1
        window = lines[i + 1:i + 30]
        # the result is made uniform right behind the atomic; a loop header before the compare is the bad shape
        for j, w in enumerate(window):
            if "Loop Header" in w:
                print("SUSPICIOUS", name, "line", i + 1, l.strip())
                bad += 1
                break
            if "s_cmp_ge" in w or "s_cmp_lt" in w or "s_cmp_ge_u32" in w or "v_cmp" in w:
                break
print("checked %d LDS ticket atomics, %d suspicious" % (seen, bad))
sys.exit(1 if bad else 0)
