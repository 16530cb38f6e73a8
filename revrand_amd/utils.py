"""Small host helpers shared by the basis and model classes (reference: revrand/utils/base.py)."""
import numpy as np


def issequence(obj):
    """True for list/tuple/generator-like containers, False for arrays, strings, scalars."""
    if isinstance(obj, (str, bytes, np.ndarray)) or np.isscalar(obj):
        return False
    try:
        iter(obj)
    except TypeError:
        return False
    return True


def atleast_list(a):
    """Wrap a non-sequence in a list; sequences become lists (utils/base.py atleast_list)."""
    return list(a) if issequence(a) else [a]


def atleast_tuple(a):
    return tuple(a) if issequence(a) else (a,)


def flatten_values(nested):
    """Depth-first concatenation of a nested sequence of scalars / arrays / [] into a 1-d array."""
    out = []

    def walk(v):
        if issequence(v):
            for u in v:
                walk(u)
        else:
            out.append(np.ravel(np.asarray(v, dtype=float)))
    walk(nested)
    return np.concatenate(out) if out else np.zeros(0)


def shapes_of(nested, shape=np.shape):
    """Nested list of shapes mirroring `nested`."""
    if issequence(nested):
        return [shapes_of(v, shape) for v in nested]
    return shape(nested)


def unflatten(x, shapes):
    """Inverse of flatten_values given the nested shapes: () -> python float, (0,) -> []."""
    pos = [0]

    def build(s):
        if isinstance(s, list):
            return [build(t) for t in s]
        if s == ():
            v = float(x[pos[0]])
            pos[0] += 1
            return v
        if s == (0,):
            return []
        k = int(np.prod(s))
        v = np.reshape(x[pos[0]:pos[0] + k], s)
        pos[0] += k
        return v
    return [build(s) for s in shapes]


def endless_permutations(N, random_state=None):
    """Indices from successive permutations of range(N), forever (utils/rand.py:7-31)."""
    from sklearn.utils import check_random_state
    generator = check_random_state(random_state)
    while True:
        for b in generator.permutation(N):
            yield b
