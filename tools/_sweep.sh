k() { timeout -k 10 400 python tools/gpu.py knobs "$@" 2>&1 | grep -v amdgpu.ids; }
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "two_ended_zyz or two_samples_per_workgroup or issue_priority" 2>&1 | tail -12
k cfg2 1024 CPF_ZYZ3_M=3 CPF_ZYZ3_M=4 CPF_ZYZ3_M=5
CPF_LIB=$PWD/cpflow_amd/libcpflow_amd.base.so timeout -k 10 200 python tools/gpu.py knobs cfg2 1024 CPF_ZYZ3_M=5 2>&1 | grep -v amdgpu.ids
