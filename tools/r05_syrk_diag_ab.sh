#!/bin/bash
# round 5: the symmetric trailing update's diagonal blocks riding in the pair launch (NPW_SYRK_DIAG_FUSED), same-box A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05l; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_tile4096_gpu.py tests/test_chain_partition_gpu.py tests/test_algorithms_gpu.py -m gpu -x -q -k "syrk or chol or Chol or cholesky or config1 or config2 or chain or masked or trsm" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for i in 1 2 3; do
  for v in 0 1; do
    export NPW_SYRK_DIAG_FUSED=$v
    timeout 200 python bench.py --no-cpu-baseline --no-north-star > $O/f${v}_$i.json 2> $O/f${v}_$i.err
    python - <<PY
import json
d=json.load(open("$O/f${v}_$i.json"))
print("fused=$v $i", d["value"], d["ms_per_step"], "median", d["ms_per_step_median"], "syrk", d["roofline"]["avg_ms"], d["roofline"]["frac"], {k: d["kernel_ms"][k] for k in ("trsm","chol","syrk_sym","syrk_sym@rest","sum_per_step") if k in d["kernel_ms"]}, d["config"]["residual_all_tiles"])
PY
  done
done
unset NPW_SYRK_DIAG_FUSED
timeout 400 python bench.py --no-cpu-baseline > $O/full.json 2> $O/full.err
python -c "
import json; d=json.load(open('$O/full.json')); print('full', d['value'], d['ms_per_step'], d['north_star'])"
