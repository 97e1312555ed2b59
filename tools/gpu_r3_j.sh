#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/api_profile.py 2>&1 | tail -50 | cut -c1-160
