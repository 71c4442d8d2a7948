MSA_SIM_SERIAL=1 python tools/sim_modes.py x 2>&1 | grep -v amdgpu.ids
python tools/sim_modes.py x 2>&1 | grep -v amdgpu.ids
