export REPO=$PWD
export SRC=$PWD/scripts/micro/old_pipe/brl_amd/csrc/brl_kernels.hip
echo "== old ws 32x11, emit idle"; CFGS=32x11 DBG=1 python scripts/micro/old_pipe/timing_old.py 2>&1 | tail -12
echo "== old pipe NP=2, emit idle"; BRL_ROLLOUT_PIPE=2 NW_DUMP=13 CFGS=32x11 DBG=1 python scripts/micro/old_pipe/timing_old.py 2>&1 | tail -14
echo "== old pipe NP=2, emit idle, timeline"; BRL_ROLLOUT_PIPE=2 NW_DUMP=13 CFGS=32x11 DBG=257 python scripts/micro/old_pipe/timing_old.py 2>&1 | tail -17
echo "== old pipe NP=2 normal"; BRL_ROLLOUT_PIPE=2 NW_DUMP=13 CFGS=32x11 DBG=256 python scripts/micro/old_pipe/timing_old.py 2>&1 | tail -17
