python tools/sim_modes.py x 2>&1 | grep -v amdgpu.ids | cut -c1-200
