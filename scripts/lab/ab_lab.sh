#!/bin/bash
# lab aid: two builds of gemm_lab one after the other, twice (A B A B), big-kernel lines only
for i in 1 2; do for b in "$@"; do echo "== $b"; timeout -k 10 120 scripts/lab/$b 0 | grep "var 11"; done; done
