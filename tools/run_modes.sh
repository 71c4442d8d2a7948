python tools/sim_modes.py x 2>&1 | grep -v amdgpu.ids | cut -c1-200
MSA_SIM_MODE=64 python tools/sim_modes.py x 2>&1 | grep -v amdgpu.ids | cut -c1-200
