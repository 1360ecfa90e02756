#!/bin/bash
# One box, the library rebuilt in between: k_gemm_x3s (csrc/mlp_gemm_x3.hpp) with two register sets of its operand stream (-DMGS_SETS=2: a
# chunk is requested two phases before it is split) against one — the PPO step (scripts/step_ab.py) and the large-batch layer
# (scripts/x3p_layer_probe.py with BRL_INFERENCE_PLANES=0).  Leaves the library built with MGS_SETS=2: rebuild (python -m brl_amd.build
# --force) afterwards.  usage (repo root, on the GPU box): bash scripts/ab_x3_sets.sh > gpurun_out/<tag>.txt   (profiles/r06/r06ap_*)
set -e
mkdir -p gpurun_out/r06ap
run() { timeout -k 10 200 python scripts/step_ab.py dw_gemm=lib dw_gemm=bf16x3 rounds=5 steps=256 2>/dev/null | tail -2; BRL_INFERENCE_PLANES=0 timeout -k 10 200 python scripts/x3p_layer_probe.py 100 2>/dev/null | grep "brl_mlp_gemm_x3 "; }
echo "== MGS_SETS=2 (shipped)"; run
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -DMGS_SETS=1 -c -o brl_amd/lib/obj/brl_mlp_gemm_x3.o brl_amd/csrc/brl_mlp_gemm_x3.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o brl_amd/lib/libbrl_hip.so brl_amd/lib/obj/*.o
echo "== MGS_SETS=1"; run
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -DMGS_SETS=2 -c -o brl_amd/lib/obj/brl_mlp_gemm_x3.o brl_amd/csrc/brl_mlp_gemm_x3.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o brl_amd/lib/libbrl_hip.so brl_amd/lib/obj/*.o
echo "== MGS_SETS=2 again"; run
