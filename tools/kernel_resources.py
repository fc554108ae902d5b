"""Developer tool: registers / scratch / occupancy of the stage kernels as the compiler reports them.
usage: python tools/kernel_resources.py <N> [filter substring ...]    (compiles stage_inst.hip with -Rpass-analysis=kernel-resource-usage)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1]
filt = sys.argv[2:]
out = "/tmp/res_n%s.txt" % n
if not os.path.exists(out) or os.environ.get("FORCE"):
    with open(out, "w") as f:
        subprocess.call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-DDFLO_STAGE_N=" + n,
                         "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/tmp/stage_n%s.o" % n, "stage_inst.hip"] + [a for a in os.environ.get("FLAGS", "").split() if a],
                        cwd=os.path.join(ROOT, "dflo_amd", "csrc"), stderr=f)
txt = open(out).read()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip().split()[0]
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = d.replace("void dflo::", "").replace("(dflo::StageArgs)", "")
    if filt and not all(f in d for f in filt):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "-1"])[1]
    print("%-46s VGPR %3s AGPR %3s scratch %4s occ %s LDS %s" % (d[:46], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
