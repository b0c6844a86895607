// cpflow_amd: ahead-of-time instances of the ZYZ-form kernel.  Built with CPF_DENSE_MLLVM_FLAGS (cpf_dense_aot.h).
#define CPF_NO_PLAIN_KERNELS
#include "cpf_zyz.h"
#include "cpf_zyzl.h"
#include "cpf_dense_aot.h"
#include "cpf_zyz_aot.h"

namespace {
template <int NQ, int LW, int ROT, int... UD>
hipError_t run(const cpf::ZyzArgs& za, long long S, size_t lds, hipStream_t stream) {
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cpf::k_cp_zyz<NQ, LW, ROT, UD...>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((cpf::k_cp_zyz<NQ, LW, ROT, UD...>), dim3((unsigned)S), dim3(64 << LW), lds, stream, za);
    return hipGetLastError();
}
const std::vector<int> RING6 = {0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0}, CHAIN5 = {0, 1, 1, 2, 2, 3, 3, 4}, CHAIN4 = {0, 1, 1, 2, 2, 3};
}  // namespace

bool cpf_zyz_has_aot(int nq, int lw, int rot, const std::vector<int>& ud) {
    if (rot != CPF_DENSE_AOT_ROT) return false;
    return (nq == 6 && (lw == 1 || lw == 2) && ud == RING6) || (nq == 5 && lw == 0 && ud == CHAIN5) || (nq == 4 && lw == 0 && ud == CHAIN4);
}

int cpf_zyz_launch_aot(int nq, int lw, int rot, const std::vector<int>& ud, const cpf::ZyzArgs& za, long long S, size_t lds,
                       hipStream_t stream, hipError_t* err) {
    if (!cpf_zyz_has_aot(nq, lw, rot, ud)) return 0;
    constexpr int R = CPF_DENSE_AOT_ROT;
    if (nq == 6 && lw == 1) *err = run<6, 1, R, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0>(za, S, lds, stream);
    else if (nq == 6 && lw == 2) *err = run<6, 2, R, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0>(za, S, lds, stream);
    else if (nq == 5) *err = run<5, 0, R, 0, 1, 1, 2, 2, 3, 3, 4>(za, S, lds, stream);
    else *err = run<4, 0, R, 0, 1, 1, 2, 2, 3>(za, S, lds, stream);
    return 1;
}

namespace {
template <int NQ, int ROT, int NSLOT, int... UD>
hipError_t run_l(const cpf::ZyzlArgs& za, unsigned grid, unsigned block, size_t lds, hipStream_t stream) {
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cpf::k_cp_zyzl<NQ, 0, ROT, NSLOT, UD...>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((cpf::k_cp_zyzl<NQ, 0, ROT, NSLOT, UD...>), dim3(grid), dim3(block), lds, stream, za);
    return hipGetLastError();
}
const std::vector<int> CHAIN4_RELABELLED = {0, 2, 2, 3, 3, 1};
}  // namespace

// nslot: items per lane + 16 x (layers whose blocks' input states stay in registers: cpf_zyzl.h SAVE)
bool cpf_zyzl_has_aot(int nq, int rot, int nslot, const std::vector<int>& ud) {
    return rot == CPF_DENSE_AOT_ROT && nq == 4 && (nslot == 1 || nslot == CPF_ZYZL_AOT_KEPT_CODE) && ud == CHAIN4_RELABELLED;
}

int cpf_zyzl_launch_aot(int nq, int rot, int nslot, const std::vector<int>& ud, const cpf::ZyzlArgs& za, unsigned grid, unsigned block, size_t lds,
                        hipStream_t stream, hipError_t* err) {
    if (!cpf_zyzl_has_aot(nq, rot, nslot, ud)) return 0;
    if (nslot == 1) *err = run_l<4, CPF_DENSE_AOT_ROT, 1, 0, 2, 2, 3, 3, 1>(za, grid, block, lds, stream);
    else *err = run_l<4, CPF_DENSE_AOT_ROT, CPF_ZYZL_AOT_KEPT_CODE, 0, 2, 2, 3, 3, 1>(za, grid, block, lds, stream);
    return 1;
}
