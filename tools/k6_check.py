import sys; sys.path.insert(0,'.')
import numpy as np
from cpflow_amd import Ansatz, connected_layer, fill_layers, prng
anz = Ansatz(3, 'cz', fill_layers(connected_layer(3), 4))
t = anz.unitary(prng.uniform_angles(prng.prng_key(0), anz.num_angles))
res = anz.learn(t, num_iterations=5000, num_repeats=100, keep_history=False)
print('ratio', np.mean([r['loss'].min() < 1e-4 for r in res]))
