#!/bin/bash
# PMC passes (each counter set in its own run, --kernel-trace only): HBM bytes and MFMA busy for the bench kernels.
TAG=${1:-x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${TAG}_$N -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-e2e > $O/pmc_${TAG}_$N.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections, json, re
out = {}
for f in sorted(glob.glob("$O/pmc_${TAG}_*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        if "conv_mfma" in k or "rowfold" in k: k += " grid=%s" % r.get("Grid_Size", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
json.dump(out, open("$O/pmc_$TAG.json", "w"), indent=1, sort_keys=True)
dom = [k for k in out if k.startswith("fav::conv3_wino4_kernel")]
f4 = bool(dom)
if not dom: dom = [k for k in out if k.startswith("fav::conv3_wino_kernel")]
if dom and all("FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k] for k in dom):
    nl = sum(out[k]["FETCH_SIZE"]["n"] for k in dom)
    f = sum(out[k]["FETCH_SIZE"]["mean"] * out[k]["FETCH_SIZE"]["n"] for k in dom) / nl
    w = sum(out[k]["WRITE_SIZE"]["mean"] * out[k]["WRITE_SIZE"]["n"] for k in dom) / nl
    # MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of
    # 16-B/lane coalesced streams (this kernel's loads are all 16 B/lane) -> doubled; WRITE_SIZE taken as is (uncalibrated)
    import hashlib
    srcf = "kernels_wino4.hip" if f4 else "kernels_wino.hip"
    sha = hashlib.sha256(open("$R/fast-artistic-videos_amd/csrc/" + srcf, "rb").read()).hexdigest()[:16]
    json.dump({"kernel": "conv3_wino4_kernel" if f4 else "conv3_wino_kernel", "source": srcf, "source_sha16": sha, "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w,
               "launches_averaged": nl, "hbm_bytes_per_launch": int((2 * f + w) * 1024),
               "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, mean over the launches of all instances in bench.py; read side doubled per the gfx950 "
                       "FETCH_SIZE calibration; algorithmic bytes per launch: ~69 MB (33.5 MB in + 33 MB out + 2.4 MB packed weights; F(2x2) kernel with pending joins: +66 MB in three of ten)"},
              open("$O/pmc_traffic_$TAG.json", "w"), indent=1)
for k in sorted(out):
    print(k[:80], {c: round(v["mean"], 1) for c, v in out[k].items()})
PY
grep -i -E "mfma|FETCH_SIZE|WRITE_SIZE" $O/counters_list.txt | head -30
