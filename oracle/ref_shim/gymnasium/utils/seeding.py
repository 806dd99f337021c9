import numpy as np


def np_random(seed=None):
    """gymnasium 0.27.1 semantics: Generator(PCG64(SeedSequence(seed))), returns (rng, entropy)."""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
        raise ValueError(f"Seed must be a non-negative integer or omitted, not {seed}")
    seed_seq = np.random.SeedSequence(seed)
    np_seed = seed_seq.entropy
    rng = np.random.Generator(np.random.PCG64(seed_seq))
    return rng, np_seed
