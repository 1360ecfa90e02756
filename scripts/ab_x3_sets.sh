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
