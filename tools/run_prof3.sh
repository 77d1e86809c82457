cd $GRAFT_REPO_ROOT
bash tools/run_profile.sh r1g_H H
