# fused gradient blocks and out-of-phase step loops of cpf_zyz3.h: sustained means (150 launches, two visits), run-time builds of one library on one box
export CPF_LIB=$PWD/cpflow_amd/libcpflow_amd.diag.so
timeout -k 10 900 python tools/micro/steady.py cfg2 1024 150 "-DCPF_Z3_FUSED=0 -DCPF_Z3_STAGGER=0" "-DCPF_Z3_FUSED=1 -DCPF_Z3_STAGGER=0" "-DCPF_Z3_FUSED=1 -DCPF_Z3_STAGGER=4" "CPF_PRIO=1;-DCPF_Z3_FUSED=1 -DCPF_Z3_STAGGER=4" "-DCPF_Z3_FUSED=0 -DCPF_Z3_STAGGER=4" 2>&1 | grep -v amdgpu.ids
