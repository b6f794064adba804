"""Deterministic synthetic weights and inputs:  value = f(seed, state_dict key, element index).

The reference ships no checkpoints or fixtures and torch's RNG stream depends on module construction order
and version, so every test / golden vector / bench run draws its tensors from this counter-based generator
(splitmix64 -> Box-Muller, pure NumPy integer + float64 math).  The same (seed, key, shape) gives the same
tensor in this container, on the GPU box and in any later round.

Scale rules (chosen so that every term on the path matters numerically -- the reference zero-initialises
rel-pos tables and pos_embed, which would hide indexing bugs):
  1-D "*.weight"            -> LayerNorm gain          1 + 0.1 n
  "*.bias"                  -> 0.1 n
  >=2-D "*.weight"          -> n / sqrt(fan_in)        (Embedding tables: 0.5 n)
  rel_pos_*                 -> 0.1 n
  anything else (pos_embed, queries, tokens, gaussian matrix, ...) -> per-name table below, default 0.5 n
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, key: str, n: int, stream: int = 0) -> np.ndarray:
    """n float64 values in (0, 1)."""
    base = (_fnv1a64(key) ^ ((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF) ^ (stream * 0xD1B54A32D192ED03)) \
        & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(base)) & _M64
    z = _splitmix64(idx)
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, key: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    half = (n + 1) // 2
    u1 = uniform01(seed, key, half, 0)
    u2 = uniform01(seed, key, half, 1)
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.concatenate([r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)])[:n]
    return (z * std + mean).astype(np.float32).reshape(shape)


_NAMED_STD = {
    "pos_embed": 0.1,
    "positional_encoding_gaussian_matrix": 1.0,
    "pad_token": 0.5,
    "text_type": 0.1,
    "log_temp": 0.1,
    "class_embedding": 0.5,
}


def param(seed: int, key: str, shape) -> np.ndarray:
    """Synthetic value for the parameter / buffer called `key` (reference state_dict naming)."""
    shape = tuple(int(s) for s in shape)
    leaf = key.split(".")[-1]
    if leaf.startswith("rel_pos"):
        return normal(seed, key, shape, 0.1)
    if leaf == "bias" or leaf == "in_proj_bias":
        return normal(seed, key, shape, 0.1)
    if leaf == "weight" or leaf == "in_proj_weight":
        if len(shape) == 1:
            return normal(seed, key, shape, 0.1, 1.0)
        parent = key.split(".")[-2] if "." in key else ""
        if "token" in parent or "embed" in parent and len(shape) == 2 and "patch" not in key:
            return normal(seed, key, shape, 0.5)
        fan_in = int(np.prod(shape[1:]))
        return normal(seed, key, shape, 1.0 / np.sqrt(fan_in))
    return normal(seed, key, shape, _NAMED_STD.get(leaf, 0.5))


def state_dict_like(seed: int, shapes: dict, prefix: str = "") -> dict:
    """{key: np.ndarray} for a {key: shape} table; `prefix` is prepended when hashing (so the same module placed
    under two parents gets different values)."""
    return {k: param(seed, prefix + k, s) for k, s in shapes.items()}
