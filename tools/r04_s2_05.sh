cd $GRAFT_REPO_ROOT
bash tools/pmc_voc.sh new 2>&1 | tail -30
