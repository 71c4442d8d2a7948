for mode in 0 1 2 3; do MSA_SIM_MODE=$mode python tools/sim_modes.py x 2>&1 | grep -v amdgpu.ids; done
