cd $GRAFT_REPO_ROOT
bash tools/run_profile.sh r1f_H H
bash tools/run_profile.sh r1f_H32 H32
