#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -3
