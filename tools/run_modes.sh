python tools/configs_bench.py C2 C4 C5 2>&1 | grep -v amdgpu.ids | cut -c1-400
echo "--- TCOLS=64"
MSA_SIM_TCOLS=64 python tools/configs_bench.py C5 2>&1 | grep -v amdgpu.ids | cut -c1-300
