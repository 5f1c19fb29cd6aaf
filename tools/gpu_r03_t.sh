#!/bin/bash
timeout -k 10 900 python tools/tpt_check.py time ref800,hd15,c3s 24,32,40,48,56 2>&1 | cut -c1-100
