#!/usr/bin/env python3
"""profiles/hbm_traffic.json from a tools/prof_traffic.sh run: HBM bytes per launch of the two SpMV kernels.

Per MI355X_MICROARCH.md (HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
bytes of wide (16 B/lane) coalesced reads -- confirmed in the same run on a 256 MiB device copy (FETCH_SIZE 131084,
WRITE_SIZE 262144).  `traffic` = 2 * FETCH_SIZE + WRITE_SIZE, i.e. every read priced as wide: an UPPER bound here,
because the 64-byte scalar/prefetch requests of the metadata (bitmaps, offsets: ~20 % of the bytes) are tallied at
full size; pricing only the packed stream as wide gives the second figure.
    python tools/make_traffic_json.py gpurun_out/traffic_r1 c3 [metadata_bytes_key metadata_bytes_value]
"""
import csv, glob, hashlib, json, os, sys

base, cfg = sys.argv[1], sys.argv[2]
out = {}
res = {}
for tag, cname in (("F", "FETCH_SIZE"), ("W", "WRITE_SIZE")):
    f = glob.glob(f"{base}_{tag}/*/*counter_collection.csv")[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for name in ("key_spmv_kernel", "key_lean_kernel", "value_spmv_kernel", "value_lean_kernel", "decode_onepass_sb", "decode_onepass_small", "decode_onepass_leanpair", "decode_onepass_kernel"):
            if name in k and r["Counter_Name"] == cname:
                acc.setdefault(name, []).append(float(r["Counter_Value"]) * 1024)
    for name, v in acc.items():
        res.setdefault(name, {})[cname] = sum(v) / len(v)
entry = {}
for name, d in res.items():
    short = "onepass" if name.startswith("decode_onepass") else name.split("_")[0]   # (key_lean_kernel and key_spmv_kernel never run in one collection)
    entry[short] = int(2 * d["FETCH_SIZE"] + d["WRITE_SIZE"])
    entry[short + "_raw"] = {"FETCH_SIZE_bytes": int(d["FETCH_SIZE"]), "WRITE_SIZE_bytes": int(d["WRITE_SIZE"])}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "hbm_traffic.json")
allj = json.load(open(path)) if os.path.exists(path) else {}
allj[cfg] = entry
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in ("spmv.hip",):   # bench.py's kernel_source_tag(): a figure is only reported for the kernels it was measured on
    h.update(open(os.path.join(ROOT, "mustafar_amd", "csrc", f), "rb").read())
allj["kernel_source_tag"] = h.hexdigest()[:12]
allj["_note"] = ("bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB counters; gfx950 FETCH_SIZE halves wide reads, calibrated on a "
                 "256 MiB copy in the same run); upper bound: the 64-byte metadata requests are not halved by the counter")
json.dump(allj, open(path, "w"), indent=1)
print(json.dumps(allj[cfg]))
