// cpflow_amd: ahead-of-time instances of the ZYZ-form kernel (cpf_zyz.h), compiled in their own translation unit
// (cpf_zyz_aot.hip) with the alloca-promotion limits that keep 32 amplitudes per thread in registers (cpf_dense_aot.h).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

namespace cpf { struct ZyzArgs; struct ZyzlArgs; }

// Launches the layer-specialised instance for (qubits, log2 waves per sample, rotation code, layer) when there is one.
// Returns 1 = launched (*err receives the launch status), 0 = no such instance.  Instances exist for the rotation string
// 'xyz' with CP blocks (CPF_DENSE_AOT_ROT) on the BASELINE layers; every other shape is built at run time (cpf_jit.h).
int cpf_zyz_launch_aot(int nq, int lw, int rot, const std::vector<int>& ud, const cpf::ZyzArgs& za, long long S, size_t lds,
                       hipStream_t stream, hipError_t* err);
bool cpf_zyz_has_aot(int nq, int lw, int rot, const std::vector<int>& ud);

// The same for the ZYZ form on the gate-by-gate layout (cpf_zyzl.h; one wave per sample): config 4's chain_layer(4) in the
// plan's relabelling (0, 1, 2, 3) -> (0, 2, 3, 1).  grid / block: workgroups and threads per workgroup (64 per sample).
// nslot: items (sides + qubits) per lane, (2 K + n + 63) / 64; the instances hold one (K <= 30 at 4 qubits) - plus 16 x the number
// of layers whose blocks' input states stay in registers (cpf_zyzl.h SAVE): 0 and CPF_ZYZL_AOT_SAVE are built.
#ifndef CPF_ZYZL_AOT_SAVE
#define CPF_ZYZL_AOT_SAVE 2      // what 256 registers per lane hold beside the kernel's own ~190 (3: 123 spills, 5: 223)
#endif
#ifndef CPF_ZYZL_AOT_SAVEL
#define CPF_ZYZL_AOT_SAVEL 0     // + 256 x the layers in front of those whose blocks' input states are kept in LDS (6 KB per sample and layer of 3 blocks):
                                 // measured with one such layer: 29.09 -> 28.96 ms where two register layers alone give 28.86 -> 28.43 - not shipped
#endif
#define CPF_ZYZL_AOT_KEPT_CODE (1 + 16 * CPF_ZYZL_AOT_SAVE + 256 * CPF_ZYZL_AOT_SAVEL)
int cpf_zyzl_launch_aot(int nq, int rot, int nslot, const std::vector<int>& ud, const cpf::ZyzlArgs& za, unsigned grid, unsigned block, size_t lds,
                        hipStream_t stream, hipError_t* err);
bool cpf_zyzl_has_aot(int nq, int rot, int nslot, const std::vector<int>& ud);
