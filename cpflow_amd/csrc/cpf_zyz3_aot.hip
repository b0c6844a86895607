// cpflow_amd: ahead-of-time instances of the two-ended ZYZ-form kernel of 3 qubits (see cpf_zyz3_aot.h)
#define CPF_NO_PLAIN_KERNELS
#include "cpf_zyz3.h"
#include "cpf_zyz3_aot.h"

namespace {
constexpr int XYZ = cpf::rot_pack(3, cpf::OP_RX, cpf::OP_RY, cpf::OP_RZ);
// lds: bytes of dynamic LDS per workgroup (PAIRS samples share one; the host may ask for more than the kernel uses)
template <int ROT, int K, int M, int PAIRS, int... UD>
hipError_t run(const cpf::KArgs& a, long long S, size_t lds, hipStream_t stream) {
    if (lds < (size_t)PAIRS * cpf::zyz3_lds_floats(K) * sizeof(float)) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cpf::k_cp_zyz3<ROT, K, M, PAIRS, UD...>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((cpf::k_cp_zyz3<ROT, K, M, PAIRS, UD...>), dim3((unsigned)((S + PAIRS - 1) / PAIRS)), dim3(128 * PAIRS), lds, stream, a);
    return hipGetLastError();
}
const std::vector<int> CHAIN3 = {0, 1, 1, 2}, CONN3 = {0, 1, 0, 2, 1, 2};
}  // namespace

bool cpf_zyz3_has_aot(int rot, int K, int M, int pairs, const std::vector<int>& ud) {
    if (rot != XYZ) return false;
    if (K == 12 && ud == CHAIN3) return (M == 3 || M == 4 || M == 5 || M == 6) && pairs == 1;
    if (K == 16 && ud == CONN3) return (M == 4 || M == 5 || M == 7 || M == 8) && pairs == 1;
    return false;
}

int cpf_zyz3_launch_aot(int rot, int K, int M, int pairs, const std::vector<int>& ud, const cpf::KArgs& a, long long S, size_t lds, hipStream_t stream, hipError_t* err) {
    if (!cpf_zyz3_has_aot(rot, K, M, pairs, ud)) return 0;
    if (K == 12 && M == 3) *err = run<XYZ, 12, 3, 1, 0, 1, 1, 2>(a, S, lds, stream);
    else if (K == 12 && M == 4) *err = run<XYZ, 12, 4, 1, 0, 1, 1, 2>(a, S, lds, stream);
    else if (K == 12 && M == 5) *err = run<XYZ, 12, 5, 1, 0, 1, 1, 2>(a, S, lds, stream);
    else if (K == 12) *err = run<XYZ, 12, 6, 1, 0, 1, 1, 2>(a, S, lds, stream);
    else if (M == 4) *err = run<XYZ, 16, 4, 1, 0, 1, 0, 2, 1, 2>(a, S, lds, stream);
    else if (M == 5) *err = run<XYZ, 16, 5, 1, 0, 1, 0, 2, 1, 2>(a, S, lds, stream);
    else if (M == 8) *err = run<XYZ, 16, 8, 1, 0, 1, 0, 2, 1, 2>(a, S, lds, stream);
    else *err = run<XYZ, 16, 7, 1, 0, 1, 0, 2, 1, 2>(a, S, lds, stream);
    return 1;
}
