#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts at 8 B per lane -> gpurun_out/prof/hbm_calib/calibration.json
set -u
OUT=$PWD/gpurun_out/prof/hbm_calib; mkdir -p $OUT; export TMPDIR=/tmp
BIN=$PWD/scratch/hbm_calib
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $c -d $OUT/$c -o $c -f csv -- $BIN ) > $OUT/$c.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
V = 37748736 * 8
known = {"calib_read": (1, 0), "calib_copy": (1, 1), "calib_triad": (2, 1), "calib_triad_nt": (2, 1)}
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/" + ctr + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            if row["Counter_Name"] == ctr and name in known:
                acc[name].append(float(row["Counter_Value"]))
    for name, vals in acc.items():
        vals = vals[1:] or vals                      # drop the first (cold) launch
        kib = sum(vals) / len(vals)
        r, w = known[name]
        want = (r if ctr == "FETCH_SIZE" else w) * V
        res.setdefault(name, {})[ctr] = {"counter_KiB": kib, "known_bytes": want, "bytes_per_counted_KiB_byte": (want / (kib * 1024) if kib else None)}
json.dump({"bytes_per_vector": V, "access": "8 B per lane (global_load_dwordx2 / global_store_dwordx2), 302 MB vectors", "kernels": res},
          open(out + "/calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
