#include "cpf_zyzl.h"
extern "C" __global__ __launch_bounds__(256, 1) void cpf_jit_kernel(const cpf::ZyzlArgs a) {
    cpf::zyzl_body<5, 2, 514, 1, 0, 1, 0, 2, 0, 3, 0, 4, 1, 2, 1, 3, 1, 4, 2, 3, 2, 4, 3, 4>(a);
}
