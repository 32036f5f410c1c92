#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_checkpoint.py tests/test_algorithms_gpu.py tests/test_residency.py -m gpu -x -q 2>&1 | tail -3
