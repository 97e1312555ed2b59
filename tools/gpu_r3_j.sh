#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/single_env_latency.py 2>&1 | tail -8
