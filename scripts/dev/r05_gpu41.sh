#!/bin/bash
cd /root/repo
timeout -k 10 600 python scripts/dev/ft_small.py 2>&1 | grep -v amdgpu.ids
