#!/bin/bash
# round-5 GPU call 1: sustained MFMA roof on random data; 256x256 GEMM per-tile stamps, global vs buffer LDS-DMA, 32x32 vs 16x16 MFMA
set -o pipefail
mkdir -p gpurun_out/r05
cd /root/repo
echo "== mfma_sustain random" ; timeout -k 10 120 scripts/lab/r05/mfma_sustain 0 | tee gpurun_out/r05/mfma_sustain_random.txt
echo "== mfma_sustain zeros" ; timeout -k 10 120 scripts/lab/r05/mfma_sustain 1 | tee gpurun_out/r05/mfma_sustain_zeros.txt
for b in dma0 dma1 dma0 dma1; do
  echo "== gemm5_lab_$b f16 random"; GEMM_LAB_F16=1 timeout -k 10 120 scripts/lab/r05/gemm5_lab_$b 0 | tee -a gpurun_out/r05/gemm5_lab_$b.txt
done
echo "== gemm5_lab_dma0 M16 f16 random"; GEMM_LAB_F16=1 VRD_BIG_M16=1 timeout -k 10 120 scripts/lab/r05/gemm5_lab_dma0 0 | tee gpurun_out/r05/gemm5_lab_dma0_m16.txt
echo "== gemm5_lab_dma1 M16 f16 random"; GEMM_LAB_F16=1 VRD_BIG_M16=1 timeout -k 10 120 scripts/lab/r05/gemm5_lab_dma1 0 | tee gpurun_out/r05/gemm5_lab_dma1_m16.txt
echo "== correctness of the buffer-DMA build: GEMM tests on the alternative library"
VRDONE_HIP_LIB=$PWD/scripts/lab/libs/libvrdone_bufdma.so timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or conv" 2>&1 | tail -5 | tee gpurun_out/r05/bufdma_tests.txt
echo "== whole step A/B"
source scripts/lab/ab_lib.sh base libvrdone_bufdma.so base libvrdone_bufdma.so 2>&1 | tee gpurun_out/r05/ab_bufdma.txt
