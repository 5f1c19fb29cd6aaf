#!/bin/bash
for q in 375 750 1000 1500; do for sl in 384 1536; do echo "min_quota $q slots $sl"; CLWRAP_SPLIT_SLOTS=$sl CLWRAP_SPLIT_MIN_QUOTA=$q timeout -k 10 600 python tools/tpt_check.py time ref800,hd15,c3s 48 2>&1 | cut -c1-100; done; done
