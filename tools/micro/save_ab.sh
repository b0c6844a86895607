# kept states of cpf_zyzl.h (SAVE layers): same-process A/B against the instance without, for builds with 1, 2 and 3 kept layers
for sv in 1 2 3; do
  CPF_LIB=$PWD/cpflow_amd/libcpflow_amd.sv$sv.so timeout -k 10 300 python tools/gpu.py knobs cfg4 2048 CPF_ZYZL_SAVE=0 CPF_ZYZL_SAVE=1 2>&1 | grep -v amdgpu.ids | sed "s/^/kept layers $sv: /"
done
