#!/usr/bin/env python
"""Developer tool: static instruction mix of one kernel in a device assembly file, split by basic-block label ranges.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDFLO_STAGE_N=3 -S --cuda-device-only -o x.s dflo_amd/csrc/stage_inst.hip
  python tools/isa_mix.py x.s _ZN4dflo12stage_kernelILi3ELi4ELi1ELi0ELi0ELi1EEEvNS_9StageArgsE"""
import collections, re, sys
txt = open(sys.argv[1]).read()
name = sys.argv[2]
i = txt.index("\n" + name + ":")
j = txt.index(".Lfunc_end", i)
cnt = collections.Counter()
for l in txt[i:j].splitlines():
    l = l.strip()
    if not l or l.startswith((";", ".", "_Z")) or l.split()[0].endswith(":"):
        continue
    cnt[l.split()[0]] += 1
groups = collections.Counter()
for op, n in cnt.items():
    g = ("v_f64" if "f64" in op else "v_other") if op.startswith("v_") else "salu" if op.startswith("s_") else \
        "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
    groups[g] += n
print("static instructions", sum(cnt.values()), dict(groups))
for op, n in cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print("%6d %s" % (n, op))
m = re.search(r"\.name:\s+" + re.escape(name) + r"\n(.*?)\n  - ", txt[j:], re.S)
meta = txt[txt.index(".amdhsa_kernel " + name):]
for k in ("next_free_vgpr", "next_free_sgpr", "accum_offset", "group_segment_fixed_size", "private_segment_fixed_size"):
    mm = re.search(r"\.amdhsa_" + k + r"\s+(\S+)", meta)
    print(k, mm.group(1) if mm else "?")
