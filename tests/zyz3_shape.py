"""What a launch of the two-ended ZYZ-form kernel of 3 qubits looks like on an MI355X (256 CUs, 160 KiB of LDS each) - the host's
rules (cpflow_amd/csrc/cpflow_amd.hip zyz3_cut / zyz3_pairs, cpf_zyz3.h zyz3_lds_floats) restated for the tests that pin the
variant string of a launch."""

NUM_CU, LDS_PER_CU = 256, 160 * 1024


def lds_bytes(K):
    rows, srows, up4 = 10 + 7 * K, 7 + 5 * K, lambda x: (x + 3) & ~3
    tot = 28 + 8 * K + 4 + 3 * up4(K) + 16 * K + 48 + 24
    return 4 * (up4(tot + rows + 1) + 2 * 128 + 128 + srows * 68)


def pairs(K, S):
    return 1          # two samples per workgroup: a switch only since the two-lane owner layout (cpflow_amd.hip zyz3_pairs)


def cut(K, S):
    per_cu, fit = -(-S // NUM_CU), min(16, LDS_PER_CU // lds_bytes(K))
    if per_cu <= 1:
        m = K // 2
    elif pairs(K, S) == 2:
        m = K // 2
    elif per_cu == 2:
        m = (25 * K + 50) // 100
    elif per_cu <= fit:
        m = (33 * K + 50) // 100
    else:
        m = (43 * K + 50) // 100
    return max(1, min(K - 1, m))


def tail(K, S):
    """the variant string behind 'zyz3/aot' or 'zyz3/jit(hipcc)'"""
    return f' waves=2 K cut={cut(K, S)}' + (' pairs=2' if pairs(K, S) == 2 else '')
