// cpflow_amd: ahead-of-time instances of the dense-block kernel (cpf_dense.h), compiled in their own translation unit
// (cpf_dense_aot.hip) with the alloca-promotion limits that keep 32 amplitudes per thread in registers.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

namespace cpf { struct DenseArgs; }

// Launches the instance for (qubits, log2 waves per sample, layer) when there is one: a layer-specialised instance if
// `ud` matches, else the table-dispatched instance of (nq, lw).  Returns 1 = layer-specialised, 2 = table-dispatched,
// 0 = no such (nq, lw); *err receives the launch status.
// rot: the template's rotation code (cpf_blocks.h); layer-specialised instances exist for CPF_DENSE_AOT_ROT only.
#define CPF_DENSE_AOT_ROT 8451          /* rot_pack(3, OP_RX, OP_RY, OP_RZ), CP blocks */
int cpf_dense_launch_aot(int nq, int lw, int rot, const std::vector<int>& ud, bool allow_specialised, const cpf::DenseArgs& da, long long S, size_t lds,
                         hipStream_t stream, hipError_t* err);
// flags every build of cpf_dense.h needs (the JIT passes the same ones)
#define CPF_DENSE_MLLVM_FLAGS "-mllvm", "-amdgpu-promote-alloca-to-vector-max-regs=128", "-mllvm", "-amdgpu-promote-alloca-to-vector-vgpr-ratio=1"
