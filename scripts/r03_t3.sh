#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q 2>&1 | grep -v "^$" | tail -25
