# kernel stats (and the trace) of four vidor.yaml training steps (48 pairs x 512 frames) -> gpurun_out/train_prof_stats${TAG}.csv
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp && rocprofv3 --kernel-trace --stats -d /tmp/tp -o tp --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/train_step.py --config vidor --pairs 48 --steps 4 > $GRAFT_REPO_ROOT/gpurun_out/train_prof${TAG}.log 2>&1
cp $(find /tmp/tp -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/train_prof_stats${TAG}.csv
python3 - <<PY
import csv, os
src = [os.path.join(d, f) for d, _, fs in os.walk("/tmp/tp") for f in fs if f.endswith("kernel_trace.csv")][0]
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/train_prof_trace${TAG}.csv", "w") as f:
    for r in rows[len(rows) * 3 // 4:]:      # the last step
        f.write("%s,%s,%s\n" % (r["Kernel_Name"][:60].replace(",", ";"), r["Start_Timestamp"], r["End_Timestamp"]))
PY
tail -2 $GRAFT_REPO_ROOT/gpurun_out/train_prof${TAG}.log | cut -c1-300
