"""Drop-in for the reference's fft_ff.py: same three functions, same signatures,
same argument meaning and error behaviour, backed by the gfx950 NTT kernels
(csrc/ntt.hip) through the C ABI.  No CPU fallback: without the shared library
and a GPU these raise kzg_snark_amd._native.NativeUnavailable.

    fft_ff(coeffs, w, F)            reference fft_ff.py:3-37
    ifft_ff(values, w, F)           reference fft_ff.py:39-58
    fft_ff_interpolation(values, g, F)   reference fft_ff.py:60-85

`F` may be a kzg_snark_amd.field.PrimeField or any object answering order()
(a Sage GF); elements may be anything int() accepts.  Outputs are F(x) elements.
A numpy uint64[n, 4] array of canonical little-endian limbs is also accepted (and
returned) by fft_ff / ifft_ff: the buffer fast path that skips per-element
Python-object conversion.
"""
import numpy as np

from . import _native
from .field import PolynomialRing, field_modulus

# scalar-field moduli the engine implements (kzg.py:26-35 curve choices)
_R_BN254 = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_R_BLS12_381 = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
_CURVE_OF_MODULUS = {_R_BN254: "bn254", _R_BLS12_381: "bls12_381"}


def _curve_for(F):
    r = field_modulus(F)
    if r not in _CURVE_OF_MODULUS:
        raise ValueError(f"fft_ff: field of size {r} is not the scalar field of bn254 or bls12_381")
    return _CURVE_OF_MODULUS[r], r


def _transform(seq, w, F, inverse):
    """Any length >= 1.  Powers of two run the tiled NTT kernels; other lengths run the
    level-by-level kernel that reproduces the reference recursion's slicing (ceil/floor halves,
    n//2 butterflies, result[n-1] = F(0) for odd n, fft_ff.py:20-35) -- the reference never checks
    the length and marlin/prover.py:439-449 relies on that when it passes list(row_A)."""
    n = len(seq)
    if n == 0:
        # fft_ff.py:16-26 with n == 0 slices two empty lists and recurses on them without end
        raise RecursionError("maximum recursion depth exceeded (fft_ff of an empty list)")
    curve, r = _curve_for(F)
    ww = _native.int_to_words(int(w) % r)
    ctx = _native.get_context(curve)
    as_buffer = isinstance(seq, np.ndarray) and seq.dtype == np.uint64 and seq.ndim == 2 and seq.shape[1] == 4
    # buffer fast path (Python-object marshalling costs far more than the
    # transform): canonical little-endian limbs in, a new array of the same shape out
    data = np.ascontiguousarray(seq).copy() if as_buffer else _native.ints_to_limbs([int(x) % r for x in seq])
    if n & (n - 1):
        ctx.fft_ff_any(data, ww, inverse)
    else:
        ctx.ntt(data, n.bit_length() - 1, ww, inverse)
    if as_buffer:
        return data
    out = _native.limbs_to_ints(data)                       # canonical: the kernels reduce below r
    bulk = getattr(F, "elements", None)                     # our PrimeField; a Sage GF is called per element
    return bulk(out) if bulk is not None else [F(v) for v in out]


def fft_ff(coeffs, w, F):
    """Evaluations of the polynomial with coefficient list `coeffs` at w^0..w^(n-1)
    (for primitive w); for any w, exactly what the reference recursion returns."""
    n = len(coeffs)
    if n == 1:
        return coeffs                  # fft_ff.py:16-17: the same list object
    return _transform(coeffs, w, F, inverse=False)


def ifft_ff(values, w, F):
    """fft_ff with w^-1, scaled by n^-1 (fft_ff.py:51-58)."""
    n = len(values)
    if n == 1:
        # fft_ff.py:54 returns the input list, :58 multiplies by F(1)^-1 = 1
        one_inv = F(1) ** (-1)
        return [x * one_inv for x in values]
    return _transform(values, w, F, inverse=True)


def fft_ff_interpolation(values, g, F):
    """Polynomial of degree < n through (g^i, values[i]) (fft_ff.py:60-85)."""
    n = len(values)
    assert (n & (n - 1)) == 0, "Length of values must be a power of 2"          # fft_ff.py:74
    order = g.multiplicative_order()                                            # fft_ff.py:77
    assert order >= n, f"Order of g ({order}) must be at least n ({n})"         # fft_ff.py:78
    coeffs = ifft_ff(values, g, F)                                              # fft_ff.py:81
    R = PolynomialRing(F, "X")                                                  # fft_ff.py:84
    return R(coeffs)                                                            # fft_ff.py:85
