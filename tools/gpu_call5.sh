cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x 2>&1 | tail -12
python tools/time_steady.py mel4096 spec4096 2>&1 | grep median
python tools/time_others.py apply_filterbank 2>&1 | grep apply
