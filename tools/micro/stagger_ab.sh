# out-of-phase step loops (CPF_Z3_STAGGER) and the fused gradient blocks (CPF_Z3_FUSED) of cpf_zyz3.h: mean of 16 launches each, same process
export CPF_LIB=$PWD/cpflow_amd/libcpflow_amd.diag.so REPS=16
D="CPF_ZYZ3_DEFS=-DCPF_Z3_FUSED"
timeout -k 10 900 python tools/gpu.py knobs ${1:-cfg2} ${2:-1024} "$D=1 -DCPF_Z3_STAGGER=0" "$D=0 -DCPF_Z3_STAGGER=0" "$D=1 -DCPF_Z3_STAGGER=2" "$D=1 -DCPF_Z3_STAGGER=3" "$D=1 -DCPF_Z3_STAGGER=4" "$D=1 -DCPF_Z3_STAGGER=8" "$D=0 -DCPF_Z3_STAGGER=4" 2>&1 | grep -v amdgpu.ids
