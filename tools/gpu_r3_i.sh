#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/pytest_i.log
cat gpurun_out/pytest_i.log
