// cpflow_amd: ahead-of-time instances of the two-ended ZYZ-form kernel of 3 qubits (see cpf_zyz3_aot.h)
#define CPF_NO_PLAIN_KERNELS
#include "cpf_zyz3.h"
#include "cpf_zyz3_aot.h"

namespace {
constexpr int XYZ = cpf::rot_pack(3, cpf::OP_RX, cpf::OP_RY, cpf::OP_RZ);
template <int ROT, int K, int M, int... UD>
hipError_t run(const cpf::KArgs& a, long long S, size_t lds, hipStream_t stream) {
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cpf::k_cp_zyz3<ROT, K, M, UD...>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((cpf::k_cp_zyz3<ROT, K, M, UD...>), dim3((unsigned)S), dim3(128), lds, stream, a);
    return hipGetLastError();
}
const std::vector<int> CHAIN3 = {0, 1, 1, 2}, CONN3 = {0, 1, 0, 2, 1, 2};
}  // namespace

int cpf_zyz3_aot_cut(int rot, int K, const std::vector<int>& ud) {
    if (rot != XYZ) return 0;
    if (K == 12 && ud == CHAIN3) return 5;
    if (K == 16 && ud == CONN3) return 7;
    return 0;
}

int cpf_zyz3_launch_aot(int rot, int K, int M, const std::vector<int>& ud, const cpf::KArgs& a, long long S, size_t lds, hipStream_t stream, hipError_t* err) {
    if (cpf_zyz3_aot_cut(rot, K, ud) != M || M == 0) return 0;
    if (K == 12) *err = run<XYZ, 12, 5, 0, 1, 1, 2>(a, S, lds, stream);
    else *err = run<XYZ, 16, 7, 0, 1, 0, 2, 1, 2>(a, S, lds, stream);
    return 1;
}
