# round 5: where the optimizer leg goes (tools/probes/optim_leg.py), optimizer tests on hardware
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05l
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 600 python tools/probes/optim_leg.py $O/optim_leg.json) 2>&1 | tee $O/optim_leg.log | grep -v Warning | cut -c1-700

