#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6
