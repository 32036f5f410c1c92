#!/bin/bash
# The N > 1 code path on one GPU with this round's first-contact fields: a world of one over RCCL (self-exchange) at the real tile
# size, and two ranks sharing the GPU (host-staged payloads) at a small size.
out=gpurun_out/r06m; mkdir -p $out
NUMPYWREN_AMD_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29711 timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --tiles 16 --no-cpu-baseline > $out/world_of_one_65536.json 2> $out/world_of_one.err
NUMPYWREN_AMD_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 --tile 1024 --tiles 8 --no-anchor > $out/two_ranks_one_gpu.json 2> $out/two_ranks.err
python - <<'PY'
import json
for f in ("gpurun_out/r06m/world_of_one_65536.json", "gpurun_out/r06m/two_ranks_one_gpu.json"):
    try:
        l = json.loads([x for x in open(f).read().splitlines() if x.startswith("{")][-1])
        c = l["config"]
        print(f, l["value"], l["ms_per_step"], c.get("xgmi_GBps"), {k: v for k, v in (c.get("link_calibration") or {}).items() if k != "samples"}, c.get("predicted_at_measured_link"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $out/world_of_one.err $out/two_ranks.err
