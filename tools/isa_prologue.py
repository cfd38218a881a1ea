"""Lab: where the integer divisions (v_rcp_iflag sequences) of a kernel's prologue sit, and how long the prologue is."""
import sys
t = open(sys.argv[1]).read()
key = sys.argv[2]
i = t.index("\n" + key)
ins = []
for l in (x.strip() for x in t[i + 1:].split("\n")[1:]):
    if l.startswith(".Lfunc_end"):
        break
    if not l or l.startswith(";") or l.startswith("."):
        continue
    ins.append(l)
first = next(k for k, l in enumerate(ins) if "v_mfma" in l)
print("instructions before the first MFMA:", first, " total:", len(ins))
print("divisions (v_rcp_iflag) at:", [k for k, l in enumerate(ins) if "v_rcp_iflag" in l])
for cls in ("s_", "v_", "ds_", "global_", "s_cbranch"):
    print(cls, sum(l.startswith(cls) for l in ins[:first]))
