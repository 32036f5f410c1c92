#!/bin/bash
# round 5: the fused triangular solve (one balanced launch for the four group products) and loop-head alignment of gemm.hip,
# same-box A/B on the default bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_tile4096_gpu.py tests/test_chain_partition_gpu.py tests/test_algorithms_gpu.py -m gpu -x -q -k "trsm or chol or Chol or cholesky or config1 or config2 or chain" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for f in 0 1; do echo "NPW_TRSM_FUSED=$f"; NPW_TRSM_FUSED=$f python tools/trsm_run.py 2>&1 | tail -8 | tr '\n' ' '; echo; done
for i in 1 2 3; do
  for v in base fused fused_align; do
    unset NUMPYWREN_AMD_LIB; export NPW_TRSM_FUSED=1
    [ $v = base ] && export NPW_TRSM_FUSED=0
    [ $v = fused_align ] && export NUMPYWREN_AMD_LIB=$R/gpurun_tmp/libnpw_align.so
    timeout 200 python bench.py --no-cpu-baseline --no-north-star > $O/${v}_$i.json 2> $O/${v}_$i.err
    python - <<PY
import json
d=json.load(open("$O/${v}_$i.json"))
print("$v $i", d["value"], d["ms_per_step"], "median", d["ms_per_step_median"], "syrk", d["roofline"]["avg_ms"], d["roofline"]["frac"], {k: d["kernel_ms"][k] for k in ("trsm","chol","syrk_sym","trtri_complete","sum_per_step") if k in d["kernel_ms"]}, d["config"]["residual_all_tiles"])
PY
  done
done
