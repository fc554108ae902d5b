#!/usr/bin/env python
"""Developer tool: variant of ONE stage-kernel unit, linked with the shipped objects of the others (seconds instead of minutes).
  python tools/build_unit_variant.py <name> <N> <element 1|2> [-DFLAG ...]   ->  scratch/variants/<name>.so
Needs dflo_amd/csrc/build/*.o of the current tree (python __graft_entry__.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
name, n, e, flags = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4:]
bdir = os.path.join(g.CSRC, "build")
odir = os.path.join(ROOT, "scratch", "variants", "obj_" + name)
os.makedirs(odir, exist_ok=True)
unit = "stage_n%d_e%d.o" % (n, e)
obj = os.path.join(odir, unit)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-w", "-c", "-o", obj,
                       "-DDFLO_STAGE_N=%d" % n, "-DDFLO_STAGE_ONLY=%d" % e] + g.SCHED.get((n, e), []) + flags + ["stage_inst.hip"], cwd=g.CSRC)
objs = [obj if f == unit else os.path.join(bdir, f) for f in sorted(os.listdir(bdir)) if f.endswith(".o")]
out = os.path.join(ROOT, "scratch", "variants", name + ".so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"], cwd=g.CSRC)
print(out)
