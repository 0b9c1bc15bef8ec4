cd $GRAFT_REPO_ROOT
timeout -k 10 100 python tools/enc_profile.py 6 2>&1 | head -60
