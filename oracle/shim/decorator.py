"""
Process-local stand-in for the PyPI ``decorator`` package, used ONLY by
``oracle/make_golden.py`` so that ``import revrand`` works in the build
container (the package is not installed and there is no network).

revrand uses it once (basis_functions.py:17,96) to keep ``inspect.signature``
intact on ``slice_transform``-wrapped methods; ``functools.wraps`` sets
``__wrapped__`` which ``inspect.signature`` follows, which is all that is needed.
This file never travels into the product or the GPU tests.
"""
import functools


def decorator(caller):
    def deco(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            return caller(func, *args, **kwargs)
        return wrapper
    return deco
