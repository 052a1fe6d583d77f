import numpy as np


def np_random(seed=None):
    # real gymnasium returns a np.random.Generator; the capture harness replaces
    # env.np_random by a recording proxy anyway
    rng = np.random.RandomState(seed)
    return rng, seed
