"""Identity-jit stand-in for numba (test infrastructure, fixture generation only).

The reference's only numba use is ``from numba import jit`` (utils/math_utils.py:14,31-36);
every jitted body is plain NumPy-compatible Python, so an identity decorator runs the
reference unchanged with NumPy semantics.  Used ONLY by tests/golden/make_golden.py in the
build container (the reference never travels to the GPU box).
"""


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(func):
        return func

    return deco


njit = jit
