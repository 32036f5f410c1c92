for pct in 0 50 100 150; do echo "== stagger $pct"; NPW_GEMM_STAGGER_PCT=$pct python tools/gemm_rank_k.py 2>&1 | grep "K   128\|K   256\|K   512"; done
