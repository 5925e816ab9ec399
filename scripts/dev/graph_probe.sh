#!/bin/bash
# dev aid: training step with and without HIP graphs
python scripts/train_step.py --steps 6 2>&1 | tail -3
python scripts/train_step.py --steps 6 --graphs 2>&1 | tail -3
