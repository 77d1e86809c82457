cd $GRAFT_REPO_ROOT
bash tools/run_profile.sh r1e_H H
bash tools/run_profile.sh r1e_C5 C5
bash tools/run_all.sh
