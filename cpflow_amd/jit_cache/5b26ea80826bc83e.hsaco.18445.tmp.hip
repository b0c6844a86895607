#include "cpf_layered.h"
extern "C" __global__ __launch_bounds__(256, 1) void cpf_jit_kernel(const cpf::KArgs a) {
    cpf::layered_body<4, 0, 514, 3, 0, 0, 2, 2, 3, 3, 1>(a);
}
