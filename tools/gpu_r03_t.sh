#!/bin/bash
for sl in 448 512 576 640; do echo "slots $sl"; CLWRAP_SPLIT_SLOTS=$sl timeout -k 10 600 python tools/tpt_check.py time ref800,hd15,c3s 48 2>&1 | cut -c1-100; done
