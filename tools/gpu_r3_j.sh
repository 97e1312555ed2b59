#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/pytest_j.log
cat gpurun_out/pytest_j.log
