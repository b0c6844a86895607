// cpflow_amd: ahead-of-time instances of the two-ended ZYZ-form kernel of 3 qubits (cpf_zyz3.h), in a translation unit of their
// own (cpf_zyz3_aot.hip) so that they build beside the rest of the library: BASELINE configs 1 / 2 (chain_layer(3), 12 blocks) and
// config 3 (connected_layer(3), 16 blocks), rotation string 'xyz', CP blocks.  Every other shape is built at run time (cpf_jit.h)
// or shipped in cpflow_amd/jit_cache/.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

namespace cpf { struct KArgs; }

// the cut (blocks swept by wave A) the instance of this shape was built with, or 0 when there is none
int cpf_zyz3_aot_cut(int rot, int K, const std::vector<int>& ud);
// Launches the instance for (rotation code, blocks, cut, layer).  Returns 1 = launched (*err receives the launch status), 0 = no such instance.
int cpf_zyz3_launch_aot(int rot, int K, int M, const std::vector<int>& ud, const cpf::KArgs& a, long long S, size_t lds, hipStream_t stream, hipError_t* err);
