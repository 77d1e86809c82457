cd $GRAFT_REPO_ROOT
python tools/potf2_time.py
python tools/prep_time.py
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
