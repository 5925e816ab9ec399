#!/bin/bash
# round-5 GPU call 8: where the f16 backward's extra 11 ms per vidor-size training step go (kernel trace, both forms)
mkdir -p /root/repo/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
for fb in 1 0; do
  export VRDONE_F16_BACKWARD=$fb
  rm -rf /tmp/prof_fb$fb
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fb$fb -o t -- python3 /root/repo/scripts/train_step.py --config vidor --pairs 48 --steps 4 > /root/repo/gpurun_out/r05/prof_train_fb$fb.log 2>&1
  echo "fb=$fb rc $?"; grep "step \|Error\|error" /root/repo/gpurun_out/r05/prof_train_fb$fb.log | tail -5 | cut -c1-200
  find /tmp/prof_fb$fb -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/r05/train_vidor48_kernel_stats_fb$fb.csv \;
done
cd /root/repo
python - <<'PY'
import csv, os
for fb in (1, 0):
    f = f"gpurun_out/r05/train_vidor48_kernel_stats_fb{fb}.csv"
    if not os.path.exists(f): print("no stats", fb); continue
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"== F16_BACKWARD={fb}: total kernel time {tot/1e6/4:.1f} ms per step (4 steps incl. the first)")
    for r in rows[:24]:
        print(f"   {float(r['TotalDurationNs'])/1e6/4:8.3f} ms/step  {int(r['Calls'])//4:5d} calls/step  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:120]}")
PY
