#!/bin/bash
# scripts/gpu_check.sh : parity tests with the scheduler forced each way and tuned, then the scheduler sweep
mkdir -p gpurun_out
RTX_SCHED=groups timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_groups.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_groups.log; [ $rc -eq 0 ] || exit 1
RTX_SCHED=refill timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_refill.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_refill.log; [ $rc -eq 0 ] || exit 1
RTX_SCHED=split timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_split.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_split.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_tuned.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_tuned.log; [ $rc -eq 0 ] || exit 1
bash scripts/sweep_sched.sh
