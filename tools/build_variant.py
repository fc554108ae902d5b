#!/usr/bin/env python
"""Developer tool: build a variant of the engine for A/B runs on one box.
  python tools/build_variant.py <name> [-DFLAG ...]   ->  scratch/variants/<name>.so
Only the translation units that see the flags are compiled again (all of them, in parallel); scratch/ travels to the
GPU box with gpurun and stays out of git."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "scratch", "variants", name + ".so")
os.makedirs(os.path.dirname(out), exist_ok=True)
g.compile_engine(out, extra_flags=flags, objdir=os.path.join(ROOT, "scratch", "variants", "obj_" + name))
print(out)
