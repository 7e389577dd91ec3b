#!/bin/bash
# L2 counters of the march of one simulated rank in the PAIRED layout, with the classify pass's
# bricklets stored plainly (AVR_CLASSIFY_STREAM=0) and streamed to memory (=1): does the march find
# them in L2?   usage: share_cache_pmc.sh <outdir> <n_ranks> <rank>
set -u
OUT=$1; N=$2; RANK=$3
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf "$OUT"; mkdir -p "$OUT"
for S in 0 1; do
  AVR_CLASSIFY_STREAM=$S timeout -k 10 150 rocprofv3 --kernel-include-regex "render_runs" \
    --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pass$S" -- \
    python3 $R/tools/rank_share.py --worker $N $RANK --ownership level_pairs --overlap 2 --classify-share 12288 --frames 40 \
    > "$OUT/run$S.log" 2>&1 || echo "pass $S failed"
  python3 - "$OUT/pass$S" "$S" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_runs_kernel<false" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("classify stores %s:" % ("streamed (sc0 sc1 nt)" if sys.argv[2] == "1" else "plain"))
for k, v in sorted(agg.items()):
    print("  %-20s n=%d mean=%.5g" % (k, len(v), sum(v) / len(v)))
if agg.get("TCC_HIT_sum") and agg.get("TCC_MISS_sum"):
    h, m = sum(agg["TCC_HIT_sum"]), sum(agg["TCC_MISS_sum"])
    print("  L2 hit rate of the march %.1f %%" % (100.0 * h / (h + m)))
PY
done
rm -rf "$OUT"/pass*
