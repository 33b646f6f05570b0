"""instruction histogram of one kernel of a gfx950 assembly dump (how profiles/r04_experiments.md sections 8-11 counted):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 [-mllvm -amdgpu-mfma-vgpr-form] -S --cuda-device-only uc2_amd/csrc/<file>.hip -o scratch/isa/<file>.s
    python scratch/isa_hist.py scratch/isa/<file>.s <mangled-name-prefix> [top]
prints totals (all / vector without MFMA / scalar / LDS / branches), the register and scratch figures and the most frequent opcodes"""
import collections, re, sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2])
i = s.index(":", i)
j = s.index("s_endpgm", i)
ins = []
for l in s[i:j].split("\n"):
    t = l.strip()
    if not l.startswith("\t") or not t or t.startswith((".", ";")):
        continue
    ins.append(t.split()[0])
c = collections.Counter(ins)
meta = [l.strip() for l in s[j:j + 5000].split("\n") if re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy):", l)]
print("instructions %d  vector (no MFMA) %d  MFMA %d  scalar %d  LDS %d  branches %d  %s" % (
    sum(c.values()), sum(n for k, n in c.items() if k.startswith("v_") and "mfma" not in k), sum(n for k, n in c.items() if "mfma" in k),
    sum(n for k, n in c.items() if k.startswith("s_")), sum(n for k, n in c.items() if k.startswith("ds_")),
    sum(n for k, n in c.items() if k.startswith("s_cbranch")), " ".join(meta[:4])))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print("  %-32s %d" % (k, v))
