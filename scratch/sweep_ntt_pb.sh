#!/bin/bash
for pb in 2 3 4 5 6 8 12 16 20 24 32; do
  echo "HB_NTT_PB=$pb"; HB_NTT_PB=$pb python scratch/time_ntt.py 2>&1 | grep NTT
done
