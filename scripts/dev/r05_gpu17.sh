#!/bin/bash
# the vidor-size training step of round 4's tree (_r04/, built from the round-4 commit) beside this tree's, same box
cd /root/repo
for rep in 1 2; do
for g in "" "--graphs"; do
  echo "-- r04 $g"; (cd _r04 && timeout -k 10 300 python scripts/train_step.py --config vidor --pairs 48 --steps 8 $g 2>&1 | grep "^step [4567]")
  for fb in 0 1; do
  echo "-- r05 $g VRDONE_F16_BACKWARD=$fb"; VRDONE_F16_BACKWARD=$fb timeout -k 10 300 python scripts/train_step.py --config vidor --pairs 48 --steps 8 $g 2>&1 | grep "^step [4567]"
  done
done
done
