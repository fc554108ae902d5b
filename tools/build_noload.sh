#!/bin/bash
# variant libraries for the async-fill upper bound: pad.so (engine with DFLO_LDS_PAD), noload.so (pad + stage unit N=3 without the u(s) loads)
set -e
cd /root/repo/dflo_amd/csrc
O=/root/repo/scratch/variants/obj_noload; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -w"
/opt/rocm/bin/hipcc $F -DDFLO_LDS_PAD_ENV -c -o $O/engine.o engine.hip &
/opt/rocm/bin/hipcc $F -DDFLO_STAGE_N=3 -DDFLO_STAGE_ONLY=1 -mllvm -amdgpu-set-wave-priority -DDFLO_NOLOAD_EXPERIMENT -c -o $O/stage_n3_e1.o stage_inst.hip &
wait
objs=""; for f in build/*.o; do b=$(basename $f); [ $b = engine.o ] && continue; objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/variants/pad.so $O/engine.o $objs -ldl
objs2=""; for f in build/*.o; do b=$(basename $f); [ $b = engine.o ] && continue; [ $b = stage_n3_e1.o ] && continue; objs2="$objs2 $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/variants/noload.so $O/engine.o $O/stage_n3_e1.o $objs2 -ldl
ls -la /root/repo/scratch/variants/*.so
