export CPF_LIB=$PWD/cpflow_amd/libcpflow_amd.diag.so
D="CPF_ZYZ3_DEFS=-DCPF_Z3_PRIO_A"
timeout -k 10 900 python tools/gpu.py knobs cfg2 1024 "$D=0 -DCPF_Z3_PRIO_B=111" "$D=0 -DCPF_Z3_PRIO_B=0" "$D=100 -DCPF_Z3_PRIO_B=111" "$D=101 -DCPF_Z3_PRIO_B=111" "$D=100 -DCPF_Z3_PRIO_B=11" "$D=101 -DCPF_Z3_PRIO_B=10" "$D=1 -DCPF_Z3_PRIO_B=110" "$D=201 -DCPF_Z3_PRIO_B=111" "$D=100 -DCPF_Z3_PRIO_B=110" "$D=211 -DCPF_Z3_PRIO_B=121" 2>&1 | grep -v amdgpu.ids
