// cpflow_amd: ahead-of-time instances of the two-ended ZYZ-form kernel of 3 qubits (cpf_zyz3.h), in a translation unit of their
// own (cpf_zyz3_aot.hip) so that they build beside the rest of the library: BASELINE configs 1 / 2 (chain_layer(3), 12 blocks) and
// config 3 (connected_layer(3), 16 blocks) with the cuts and samples per workgroup of their launch regimes (cpflow_amd.hip
// zyz3_cut), rotation string 'xyz', CP blocks.  Every other shape is built at run time (cpf_jit.h) or shipped in
// cpflow_amd/jit_cache/.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

namespace cpf { struct KArgs; }

// is there an instance for (rotation code, blocks, cut = blocks swept by wave A, samples per workgroup, layer)?
bool cpf_zyz3_has_aot(int rot, int K, int M, int pairs, const std::vector<int>& ud);
// Launches the instance for (rotation code, blocks, cut, samples per workgroup, layer); lds: bytes of dynamic LDS per workgroup
// (at least what the instance needs).
// Returns 1 = launched (*err receives the launch status), 0 = no such instance.
int cpf_zyz3_launch_aot(int rot, int K, int M, int pairs, const std::vector<int>& ud, const cpf::KArgs& a, long long S, size_t lds, hipStream_t stream, hipError_t* err);
