import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import bench
for th in (1, 8, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    r = bench.cpu_baseline(threads=th, nx=256, steps=1)
    print(th, round(r["value"], 1), r["sample"][:60])
