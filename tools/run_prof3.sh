cd $GRAFT_REPO_ROOT
bash tools/run_profile.sh r1h_H H
bash tools/run_all.sh
