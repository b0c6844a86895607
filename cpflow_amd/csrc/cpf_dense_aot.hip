// cpflow_amd: ahead-of-time instances of the dense-block kernel.  Built with CPF_DENSE_MLLVM_FLAGS (cpf_dense_aot.h).
#define CPF_NO_PLAIN_KERNELS
#include "cpf_dense.h"
#include "cpf_dense_aot.h"

namespace {
template <int NQ, int LW, int ROT, int... UD>
hipError_t go(const cpf::DenseArgs& da, long long S, size_t lds) {
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cpf::k_cp_dense<NQ, LW, ROT, UD...>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
template <int NQ, int LW, int ROT, int... UD>
hipError_t run(const cpf::DenseArgs& da, long long S, size_t lds, hipStream_t stream) {
    hipError_t e = go<NQ, LW, ROT, UD...>(da, S, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((cpf::k_cp_dense<NQ, LW, ROT, UD...>), dim3((unsigned)S), dim3(64 << LW), lds, stream, da);
    return hipGetLastError();
}
}  // namespace

int cpf_dense_launch_aot(int nq, int lw, int rot, const std::vector<int>& ud, bool allow_specialised, const cpf::DenseArgs& da, long long S, size_t lds,
                         hipStream_t stream, hipError_t* err) {
    // layer-specialised: the BASELINE shapes (config 5: 6 qubits on a ring; config 4: chain_layer(4)) and chain_layer(5)
    // (rot = CPF_DENSE_AOT_ROT: rotation string 'xyz' with CP blocks, the reference's default; any other string is built at
    // run time or served by the table-dispatched instances, whose walks read the kinds from the records)
    if (allow_specialised && rot == CPF_DENSE_AOT_ROT) {
        constexpr int R = CPF_DENSE_AOT_ROT;
        if (nq == 6 && lw == 1 && ud == std::vector<int>{0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0}) { *err = run<6, 1, R, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0>(da, S, lds, stream); return 1; }
        if (nq == 6 && lw == 2 && ud == std::vector<int>{0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0}) { *err = run<6, 2, R, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0>(da, S, lds, stream); return 1; }
        if (nq == 5 && lw == 0 && ud == std::vector<int>{0, 1, 1, 2, 2, 3, 3, 4}) { *err = run<5, 0, R, 0, 1, 1, 2, 2, 3, 3, 4>(da, S, lds, stream); return 1; }
        if (nq == 4 && lw == 0 && ud == std::vector<int>{0, 1, 1, 2, 2, 3}) { *err = run<4, 0, R, 0, 1, 1, 2, 2, 3>(da, S, lds, stream); return 1; }
    }
    // table-dispatched: any circuit
    if (nq == 6 && lw == 1) { *err = run<6, 1, 0>(da, S, lds, stream); return 2; }
    if (nq == 6 && lw == 2) { *err = run<6, 2, 0>(da, S, lds, stream); return 2; }
    if (nq == 5 && lw == 0) { *err = run<5, 0, 0>(da, S, lds, stream); return 2; }
    if (nq == 5 && lw == 1) { *err = run<5, 1, 0>(da, S, lds, stream); return 2; }
    if (nq == 4 && lw == 0) { *err = run<4, 0, 0>(da, S, lds, stream); return 2; }
    return 0;
}
