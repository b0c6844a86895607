# same-process A/B of the cut (blocks of wave A) of the two-ended 3-qubit kernel over launch sizes: python tools/gpu.py knobs
set -e
k() { timeout -k 10 400 python tools/gpu.py knobs "$@" 2>&1 | grep -v amdgpu.ids; }
k cfg3 512 CPF_ZYZ3_M=2 CPF_ZYZ3_M=3 CPF_ZYZ3_M=4 CPF_ZYZ3_M=5
k cfg2 1024 CPF_ZYZ3_M=2 CPF_ZYZ3_M=3 CPF_ZYZ3_M=4
k cfg2 2048 CPF_ZYZ3_M=3 CPF_ZYZ3_M=4 CPF_ZYZ3_M=5
k cfg2 4096 CPF_ZYZ3_M=3 CPF_ZYZ3_M=4 CPF_ZYZ3_M=5
k cfg3 4096 CPF_ZYZ3_M=4 CPF_ZYZ3_M=5 CPF_ZYZ3_M=6 CPF_ZYZ3_M=7
k cfg2 256 CPF_ZYZ3_M=3 CPF_ZYZ3_M=4 CPF_ZYZ3_M=5 CPF_ZYZ3_M=6
k cfg3 128 CPF_ZYZ3_M=4 CPF_ZYZ3_M=5 CPF_ZYZ3_M=6 CPF_ZYZ3_M=8
