"""Lab: list the waits / barriers / first MFMA of one kernel in a hipcc -S dump (instruction index, loads issued so far)."""
import sys
t = open(sys.argv[1]).read()
key = sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
i = t.index("\n" + key)
body = t[i + 1:].split("\n")
cnt = n = nm = 0
for ln in body[1:]:
    l = ln.strip()
    if l.startswith(".Lfunc_end"):
        break
    if not l or l.startswith(";") or l.startswith("."):
        continue
    n += 1
    if "global_load" in l or "buffer_load" in l:
        cnt += 1
    if "v_mfma" in l:
        nm += 1
        if nm == 1:
            print(n, cnt, "first MFMA")
    if any(k in l for k in ("s_waitcnt vmcnt", "s_barrier", "s_memtime")) and n < limit:
        print(n, cnt, l[:80])
print("instructions:", n, "loads:", cnt, "mfma:", nm)
