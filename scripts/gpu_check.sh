#!/bin/bash
# scripts/gpu_check.sh : parity tests under both schedulers, then the scheduler sweep (static work)
mkdir -p gpurun_out
RTX_SCHED=groups timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_groups.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_groups.log; [ $rc -eq 0 ] || exit 1
RTX_SCHED=refill timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_refill.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_refill.log; [ $rc -eq 0 ] || exit 1
bash scripts/sweep_sched.sh
