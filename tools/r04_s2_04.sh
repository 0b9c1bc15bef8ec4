cd $GRAFT_REPO_ROOT
bash tools/pmc_voc.sh old SPARKMI_CB2_MIN=0 2>&1 | tail -30
