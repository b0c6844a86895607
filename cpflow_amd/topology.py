"""Qubit-connectivity layers and their filling into template placements
(counterpart of cpflow/topology.py:7-20, 36-38)."""


def connected_layer(num_qubits):
    """All-to-all: every pair [i, j] with i < j."""
    return [[i, j] for i in range(num_qubits) for j in range(i + 1, num_qubits)]


def chain_layer(num_qubits):
    """Nearest neighbours on a line."""
    return [[i, i + 1] for i in range(num_qubits - 1)]


def ring_layer(num_qubits):
    """Nearest neighbours on a ring (not in the reference; used by the 6-qubit benchmark config)."""
    return [[i, (i + 1) % num_qubits] for i in range(num_qubits)]


def fill_layers(layer, depth):
    """`depth` blocks: as many complete layers as fit, then the first depth % len(layer) pairs."""
    full, rest = divmod(depth, len(layer))
    return {'layers': [layer, full], 'free': layer[:rest]}


def num_qubits_from_layer(layer):
    """Largest qubit index appearing in the layer, plus one."""
    return max(q for pair in layer for q in pair) + 1
