cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 5 --warmup 2 2>/dev/null | tail -c 400
