"""Copy the digests of gpurun_out/prof/<tag>/ (tools/profile.sh) into profiles/<round>/ and refresh
profiles/traffic.json (HBM bytes per stage-kernel launch, read by bench.py for roofline.traffic).
usage: python tools/collect_profiles.py r01 c2_r1:c2_q2_hllc_1024 q1lxf_r1:c2_q1_lxf_1024 ..."""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
tf = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(tf)) if os.path.exists(tf) else {}
for spec in sys.argv[2:]:
    tag, key = spec.split(":")
    src = os.path.join(ROOT, "gpurun_out", "prof", tag)
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, tag + "_summary.txt"))
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, tag + "_kernel_stats.csv"))
    for line in open(os.path.join(src, "trace.log")):
        if line.startswith('{"metric"'):
            open(os.path.join(dst, tag + "_bench.json"), "w").write(line)
    t = json.load(open(os.path.join(src, "traffic.json")))
    st = {k: v for k, v in t.items() if k.startswith("stage_kernel")}
    # mean over the RK stages of one step: the first stage (MODE 0) does not read u(n)
    mode0 = [v["hbm_bytes_per_launch"] for k, v in st.items() if k.split(",")[2].strip() == "0"]
    mode1 = [v["hbm_bytes_per_launch"] for k, v in st.items() if k.split(",")[2].strip() == "1"]
    n_rk = 2 if next(iter(st)).split("<")[1].split(",")[0].strip() == "2" else 3
    rec = {"hbm_bytes_per_launch": (mode0[0] + (n_rk - 1) * mode1[0]) / n_rk, "first_stage": mode0[0], "later_stages": mode1[0],
           "source": "profiles/%s/%s_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 per the gfx950 correction)" % (rnd, tag),
           "calibration": "profiles/r02/hbm_calibration.json: at 8 B per lane FETCH_SIZE x 2.000 and WRITE_SIZE x 1.000 reproduce known byte counts", "note": "mean over the RK stages of one step: the first stage does not read u(n)"}
    lim = [v["hbm_bytes_per_launch"] for k, v in t.items() if k.startswith("limiter")]
    if lim:
        rec["limiter_kernel_bytes_per_launch"] = lim[0]
    traffic[key] = rec
    print(tag, key, rec["hbm_bytes_per_launch"])
json.dump(traffic, open(tf, "w"), indent=1)
