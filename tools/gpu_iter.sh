cd $GRAFT_REPO_ROOT
timeout 600 python tools/profile_eval.py 2>&1 | tail -48 | cut -c1-200
