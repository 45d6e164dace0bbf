"""Instruction order of a kernel's hot loop from device assembly (no GPU needed): one letter per instruction from the first inner-loop header to
the first s_barrier behind it -- M = MFMA, r / w = LDS read / write, v = other VALU, L = buffer load, s = scalar, . = s_waitcnt, | = barrier.
    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S file.hip -o file.s ;  python scripts/isa_order.py file.s <mangled-name fragment> [loop index]"""
import sys

s = open(sys.argv[1]).read()
frag = sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 0
names = [ln.split(":")[0] for ln in s.splitlines() if ln.startswith("_Z") and ":" in ln and frag in ln.split(":")[0]]
if not names:
    raise SystemExit("no kernel label contains %r" % frag)
name = names[0]
a = s.index(name + ":")
b = s.index(".end_amdhsa_kernel", a) if ".end_amdhsa_kernel" in s[a:] else len(s)
body = s[a:b].splitlines()
idx = [i for i, l in enumerate(body) if "Inner Loop Header" in l or "Loop Header" in l]
start = idx[which] if idx else 0
seq = []
for l in body[start:]:
    t = l.strip().split()
    if not t or t[0].startswith(";") or t[0].startswith("."):
        continue
    op = t[0]
    if op.startswith("v_mfma"):
        seq.append("M")
    elif op.startswith("ds_read"):
        seq.append("r")
    elif op.startswith("ds_write"):
        seq.append("w")
    elif op.startswith("v_"):
        seq.append("v")
    elif op.startswith("buffer_load") or op.startswith("global_load"):
        seq.append("L")
    elif op.startswith("s_waitcnt"):
        seq.append(".")
    elif op.startswith("s_barrier"):
        seq.append("|")
        break
    elif op.startswith("s_"):
        seq.append("s")
print(name)
print("".join(seq))
print({k: seq.count(k) for k in "MrwvLs."})
