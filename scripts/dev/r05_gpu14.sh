#!/bin/bash
# vidor-size training step: where the f16 backward's extra time goes (kernel trace per call), eager and as graphs
mkdir -p gpurun_out/r05
cd /root/repo
export TMPDIR=/tmp
for fb in 1 0; do
  echo "-- graphs, VRDONE_F16_BACKWARD=$fb"; VRDONE_F16_BACKWARD=$fb timeout -k 10 300 python scripts/train_step.py --config vidor --pairs 48 --steps 8 --graphs 2>&1 | grep "^step [4567]"
done
for fb in 1 0; do
  rm -rf /tmp/tr$fb
  VRDONE_F16_BACKWARD=$fb timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr$fb -- python scripts/train_step.py --config vidor --pairs 48 --steps 4 > /tmp/tr$fb.log 2>&1
  echo "rocprof fb=$fb rc $?"
  f=$(find /tmp/tr$fb -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r05/train14_stats_fb$fb.csv
  t=$(find /tmp/tr$fb -name '*kernel_trace.csv' | head -1)
  python - "$t" gpurun_out/r05/train14_trace_fb$fb.txt <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step only: the last quarter of the launches
n = len(rows); rows = rows[3 * n // 4:]
out = open(sys.argv[2], "w")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    agg[name][0] += 1; agg[name][1] += d
    if "absmax" in name or "attn_bwd" in name:
        out.write("%-60s grid %s  %.1f us\n" % (name, r.get("Grid_Size", "?"), d))
out.write("\n== last step, per kernel\n")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    out.write("%-60s %5d calls %9.1f us\n" % (k, c, t))
out.write("total %.1f us over %d launches; span %.1f us\n" % (sum(v[1] for v in agg.values()), len(rows),
          (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3))
P
done
