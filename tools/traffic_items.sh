#!/bin/bash
# Round 6: where the bytes of a one-pass launch go (TCC FETCH_SIZE / WRITE_SIZE, one counter per pass: tools/prof_traffic.sh).
#   usage (GPU box, repo root): tools/traffic_items.sh <tag>     -> gpurun_out/<tag>_traffic_items.txt, profiles/hbm_traffic.json (c3, c2, b1)
# The product library at c3 / c2 / b1, and at c3 the probe build whose stream loads return zeros without touching memory
# (-DMUSTAFAR_PROBE_NOSTREAM; built on the box if absent): its FETCH_SIZE is everything BUT the packed streams, so
#   stream bytes (wide 16-byte-per-lane reads: the counter halves them)  = 2 x (FETCH_base - FETCH_nostream)
#   everything else (64-byte metadata requests, windows, q, e)           = FETCH_nostream at face value
set -e -o pipefail
TAG=${1:-r06}; R=$(pwd); O=$R/gpurun_out/${TAG}_traffic_items.txt; : > $O
[ -f mustafar_amd/lib/variants/libmustafar_hip_nostream.so ] || tools/build_variant.sh nostream -DMUSTAFAR_PROBE_NOSTREAM > /dev/null
for C in c3 c2 b1; do
  tools/prof_traffic.sh ${TAG}_$C $C > /dev/null 2> gpurun_out/traffic_${TAG}_$C.err
  echo "== $C product library" >> $O; python3 tools/make_traffic_json.py gpurun_out/traffic_${TAG}_$C $C >> $O
done
MUSTAFAR_HIP_LIB=$R/mustafar_amd/lib/variants/libmustafar_hip_nostream.so tools/prof_traffic.sh ${TAG}_c3ns c3 > /dev/null 2> gpurun_out/traffic_${TAG}_c3ns.err
echo "== c3 no-stream probe library (onepass_raw: everything but the packed streams)" >> $O
python3 - >> $O <<PY
import csv, glob
for tag, cname in (("F", "FETCH_SIZE"), ("W", "WRITE_SIZE")):
    f = glob.glob("gpurun_out/traffic_${TAG}_c3ns_%s/*/*counter_collection.csv" % tag)[0]
    v = [float(r["Counter_Value"]) * 1024 for r in csv.DictReader(open(f)) if "decode_onepass" in r["Kernel_Name"] and r["Counter_Name"] == cname]
    print(cname, int(sum(v) / len(v)), "bytes per one-pass launch, avg of", len(v))
    if cname == "FETCH_SIZE":
        f_ns = sum(v) / len(v)
import json
path = "profiles/hbm_traffic.json"
j = json.load(open(path))
raw = j["c3"]["onepass_raw"]
stream = 2 * (raw["FETCH_SIZE_bytes"] - f_ns)
j["c3"]["onepass_items"] = {"stream_reads": int(stream), "other_reads_at_face_value": int(f_ns), "writes": raw["WRITE_SIZE_bytes"],
                            "total": int(stream + f_ns + raw["WRITE_SIZE_bytes"]),
                            "how": "stream = 2 x (FETCH_SIZE - FETCH_SIZE of the no-stream probe build): only the 16-byte-per-lane stream loads are halved by the counter; "
                                   "bitmaps / offsets / q / e (64-byte requests) and the windows at face value; `onepass` above prices EVERY read as wide (upper bound)"}
json.dump(j, open(path, "w"), indent=1)
print("c3 itemised:", json.dumps(j["c3"]["onepass_items"]))
PY
rm -rf gpurun_out/traffic_${TAG}_*_F gpurun_out/traffic_${TAG}_*_W
cat $O
