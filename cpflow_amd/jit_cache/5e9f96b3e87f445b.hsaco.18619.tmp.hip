#include "cpf_zyzl.h"
extern "C" __global__ __launch_bounds__(256, 1) void cpf_jit_kernel(const cpf::ZyzlArgs a) {
    cpf::zyzl_body<5, 2, 8451, 1, 0, 1, 1, 2, 2, 3, 3, 4>(a);
}
