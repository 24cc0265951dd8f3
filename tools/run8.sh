cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/prepare_stages.py 2>&1 | grep -v Warn > gpurun_out/prepare_stages.txt; cat gpurun_out/prepare_stages.txt
python tools/step_profile.py 2>&1 | grep -v Warn | head -50 > gpurun_out/step_profile.txt; head -45 gpurun_out/step_profile.txt
uptime; nproc
